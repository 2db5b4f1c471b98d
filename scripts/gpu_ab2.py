"""A/B timing of scheduler knobs inside ONE process on ONE GPU (kernel times vary by several percent between boxes and
runs, so variants are interleaved round-robin and compared by their medians).
usage: python scripts/gpu_ab2.py [workload args of bench-like form]   env ABSETS="name:VAR=val,VAR=val;name2:..." """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb, MODES
from opal_amd.score_matrix import blosum50

n = int(os.environ.get("NSEQ", "100000")); qlen = int(os.environ.get("QLEN", "300"))
mode = MODES[os.environ.get("MODE", "SW")]; st = int(os.environ.get("ST", "0")); go = int(os.environ.get("GO", "11"))
res, offs, lens = synth.database(n, seed=42)
q = synth.query(qlen, seed=7); m = blosum50()[1]
pdb = PreparedDb(res, offs)
cells = float(qlen) * float(offs[-1])
bufs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(3)]
sets = [("default", {})]
for part in filter(None, os.environ.get("ABSETS", "").split(";")):
    name, kv = part.split(":", 1)
    sets.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
allkeys = sorted({k for _, d in sets for k in d})
times = {name: [] for name, _ in sets}
for it in range(int(os.environ.get("ITERS", "25"))):
    for name, env in sets:
        for k in allkeys:
            os.environ.pop(k, None)
        os.environ.update(env)
        rc, kms = pdb.search_device(q, go, 1, m, st, mode, bufs[0].data_ptr(), bufs[1].data_ptr() if st else 0, bufs[2].data_ptr() if st else 0)
        assert rc == 0
        if it >= 3:
            times[name].append(kms)
for name, _ in sets:
    t = np.array(times[name])
    print(f"{name:28s} median {np.median(t):.4f} ms  min {t.min():.4f}  GCUPS(median) {cells / np.median(t) / 1e6:.0f}", flush=True)
