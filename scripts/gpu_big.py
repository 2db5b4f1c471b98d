"""Large-shard runs: C4-like (513-aa query x 625k sequences = one GPU's share of 5M) and C5-like (2000-aa query)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.fasta import read_fasta
from opal_amd.score_matrix import blosum50
n = int(os.environ.get("NSEQ", "625000"))
t0 = time.time(); res, offs, lens = synth.database(n, seed=42); print("gen s", time.time() - t0, "residues", offs[-1], flush=True)
alpha, m = blosum50()
t0 = time.time(); pdb = PreparedDb(res, offs); print("pack+upload s", time.time() - t0, flush=True)
score = torch.zeros(n, dtype=torch.int32, device="cuda")
eq = torch.zeros(n, dtype=torch.int32, device="cuda"); et = torch.zeros(n, dtype=torch.int32, device="cuda")
orc = oracle_lib.Oracle()
rng = np.random.RandomState(0)
pick = np.unique(np.concatenate([np.argsort(lens)[-10:], rng.choice(n, 60, replace=False)]))
db = [res[offs[i]:offs[i + 1]] for i in pick]
queries = {"P18080(513)": read_fasta(os.path.join(ROOT, "tests/golden/data/query_P18080.fasta"), alpha)[0],
           "synth2000": synth.query(2000, seed=7)}
for qn, q in queries.items():
    cells = float(len(q)) * float(offs[-1])
    for go in (3, 11):
        for mode, st in (("SW", 0), ("SW", 1), ("NW", 1), ("HW", 1), ("OV", 1)):
            if qn == "synth2000" and mode != "SW" and go == 11: continue
            ks = []
            for it in range(2):
                rc, kms = pdb.search_device(q, go, 1, m, st, oracle_lib.MODES[mode], score.data_ptr(), eq.data_ptr() if st else 0, et.data_ptr() if st else 0)
                ks.append(kms)
            s = score.cpu().numpy()[pick]
            orc_rc, os_, oq, ot = orc.search(q, db, go, 1, m, st, oracle_lib.MODES[mode])
            ok = (s == os_).all() and (not st or ((eq.cpu().numpy()[pick] == oq).all() and (et.cpu().numpy()[pick] == ot).all()))
            print(f"{qn} gaps {go}/1 {mode} st={st}: rc={rc} kernel_ms={min(ks):.2f} GCUPS={cells/min(ks)/1e6:.0f} sample-vs-oracle={'OK' if ok else 'MISMATCH'} max={score.max().item()}", flush=True)
