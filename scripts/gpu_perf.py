"""Throughput exploration on the GPU: uniform-length database (no tail) and the C2 database, per lanes-per-sequence."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50

n = int(os.environ.get("NSEQ", "100000"))
qlen = int(os.environ.get("QLEN", "300"))
res, offs, lens = synth.database(n, seed=42)
q = synth.query(qlen, seed=7); m = blosum50()[1]
score = torch.zeros(n, dtype=torch.int32, device="cuda")

def run(pdb, tag, cells, mode=3, st=0):
    for g in os.environ.get("GLIST", "auto,8,16,32,64").split(","):
        if g == "auto": os.environ.pop("OPAL_HIP_LANES_PER_SEQ", None)
        else: os.environ["OPAL_HIP_LANES_PER_SEQ"] = g
        ks = []
        for it in range(4):
            rc, kms = pdb.search_device(q, 11, 1, m, st, mode, score.data_ptr(), score.data_ptr() if st else 0, score.data_ptr() if st else 0)
            ks.append(kms)
        print(f"{tag} G={g}: rc={rc} kernel_ms={min(ks):.3f} GCUPS={cells / min(ks) / 1e6:.1f}", flush=True)

if os.environ.get("UNIFORM", "1") == "1":
    L = int(os.environ.get("ULEN", "347"))
    ures = res[: n * L]
    uoffs = np.arange(n + 1, dtype=np.int64) * L
    run(PreparedDb(ures, uoffs), f"uniform L={L}", float(qlen) * n * L)
if os.environ.get("C2", "1") == "1":
    run(PreparedDb(res, offs), "c2", float(qlen) * float(offs[-1]))
    # without the tail: sequences shorter than 1000
    keep = np.nonzero(lens < int(os.environ.get("CUT", "1000")))[0]
    seqs = [res[offs[i]:offs[i + 1]] for i in keep]
    run(PreparedDb.from_sequences(seqs), f"c2 L<cut ({len(keep)} seqs)", float(qlen) * float(lens[keep].sum()))
