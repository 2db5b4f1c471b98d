"""Cold-process cost of the drop-in call: writes the inputs of C1 (O74807 x uniprot_sprot15, reference test_data)
and of the published shape (P19930 x 541 762 synthetic sequences) as binary files, builds scripts/cold/cold_process.cpp against
libopal.so and runs it in fresh processes (5 each); also a bare HIP process (runtime initialisation only) for reference."""
import os, subprocess, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from opal_amd import synth
from opal_amd.fasta import read_fasta
from opal_amd.score_matrix import blosum50
import cases

alphabet, m = blosum50()
OUT = os.environ.get("COLD_OUT", "/tmp/cold"); os.makedirs(OUT, exist_ok=True)


def write(path, q, lens, res):
    with open(path, "wb") as f:
        f.write(struct.pack("4i", len(q), len(lens), m.shape[0], 0))
        f.write(np.ascontiguousarray(q, np.uint8).tobytes()); f.write(np.ascontiguousarray(m, np.int32).tobytes())
        f.write(np.ascontiguousarray(lens, np.int32).tobytes()); f.write(np.ascontiguousarray(res, np.uint8).tobytes())


q110 = read_fasta(os.path.join(cases.DATA, "query_O74807.fasta"), alphabet)[0]
db15 = read_fasta(os.path.join(cases.DATA, "uniprot_sprot15.fasta"), alphabet)
write(f"{OUT}/c1.bin", q110, [len(s) for s in db15], np.concatenate(db15))
q195 = read_fasta(os.path.join(cases.DATA, "query_P19930.fasta"), alphabet)[0]
n = synth.SPROT_SEQS
lens = synth.lengths(42, 0, n, "sprot")
res, offs, _ = synth.database_subset(np.arange(n, dtype=np.int64), 42, lens=lens, stride=synth.SPROT_STRIDE)
write(f"{OUT}/sprot.bin", q195, lens, res)
res2, offs2, lens2 = synth.database(100000, seed=42)
write(f"{OUT}/c2.bin", synth.query(300, seed=7), lens2, res2)
exe = f"{OUT}/cold_process"
csrc = os.path.join(ROOT, "opal_amd", "csrc")
subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "scripts", "cold", "cold_process.cpp"), "-I" + os.path.join(ROOT, "include"),
                       "-L" + csrc, "-lopal", "-Wl,-rpath," + csrc, "-o", exe])
print(f"libopal.so: {os.path.getsize(os.path.join(csrc, 'libopal.so')) / 1e6:.1f} MB", flush=True)
for name in ("c1", "c2", "sprot"):
    for st in (0,):
        for rep in range(5):
            out = subprocess.run([exe, f"{OUT}/{name}.bin", "3", str(st)], capture_output=True, text=True, env=dict(os.environ, **{k: v for k, v in (("X", "1"),)}))
            print(f"{name} searchType={st} run {rep}: {out.stdout.strip()} {out.stderr.strip()[-200:]}", flush=True)
