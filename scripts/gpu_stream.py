"""Streaming SW score kernel against the class kernels: same scores, kernel time, for several query lengths and
lanes-per-sequence settings; a sample against the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import opal_amd
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
import oracle_lib

n = int(os.environ.get("NSEQ", "20000"))
res, offs, lens = synth.database(n, seed=42)
m = blosum50()[1]
pdb = PreparedDb(res, offs)
orc = oracle_lib.Oracle()
sample = np.random.RandomState(5).choice(n, size=min(n, 120), replace=False)
sdb = [res[offs[i]:offs[i + 1]] for i in sample]
a = torch.zeros(n, dtype=torch.int32, device="cuda")
b = torch.zeros(n, dtype=torch.int32, device="cuda")
e1 = torch.zeros(n, dtype=torch.int32, device="cuda")
e2 = torch.zeros(n, dtype=torch.int32, device="cuda")
ST = int(os.environ.get("ST", "0"))  # 1: with end locations

def run(q, go, ge, env, out, iters):
    for k in ("OPAL_HIP_STREAM", "OPAL_HIP_LANES_PER_SEQ", "OPAL_HIP_NARROW"):
        opal_amd.set_option(k, None)
    for k, v in env.items():
        opal_amd.set_option(k, v)
    ks = []
    out.fill_(-7)
    for _ in range(iters):
        rc, kms = pdb.search_device(q, go, ge, m, ST, 3, out.data_ptr(), e1.data_ptr() if ST else 0, e2.data_ptr() if ST else 0)
        assert rc == 0, rc
        ks.append(kms)
    return min(ks)

cases = [(100, {"OPAL_HIP_LANES_PER_SEQ": "8"}), (100, {"OPAL_HIP_LANES_PER_SEQ": "4"}), (300, {}), (513, {}), (513, {"OPAL_HIP_LANES_PER_SEQ": "32"}),
         (1000, {}), (1300, {}), (1400, {"OPAL_HIP_NARROW": "18"}), (2000, {}), (2500, {})]
only = os.environ.get("QLIST")
iters = int(os.environ.get("ITERS", "3"))
for qlen, extra in cases:
    if only and str(qlen) not in only.split(","):
        continue
    q = synth.query(qlen, seed=7)
    for go, ge in ((11, 1), (3, 1)):
        t0 = run(q, go, ge, {"OPAL_HIP_STREAM": "0"}, a, iters)
        t1 = run(q, go, ge, dict(extra, OPAL_HIP_STREAM="1"), b, iters)
        same = bool((a == b).all().item())
        rc, os_, _, _ = orc.search(q, sdb, go, ge, m, ST, 3)
        got = b.cpu().numpy()[sample]
        ok = bool((got == os_).all())
        cells = float(qlen) * float(offs[-1])
        print(f"Q={qlen} gaps {go}/{ge} {extra}: classes {t0:.3f} ms ({cells / t0 / 1e6:.0f} GCUPS)  stream {t1:.3f} ms ({cells / t1 / 1e6:.0f} GCUPS)  "
              f"same={same} oracle_sample={ok}", flush=True)
        if not same:
            bad = (a != b).nonzero().flatten()[:8].cpu().numpy()
            print("   first mismatches:", [(int(i), int(a[i]), int(b[i]), int(lens[i])) for i in bad], flush=True)
