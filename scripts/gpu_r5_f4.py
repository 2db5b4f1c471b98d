"""sprot 541 762 x 300-aa SW: the natural plan (one launch + swapped) against its alternatives, with the scheduler's debug lines;
and the run-to-run spread of the 1000-aa global-mode searches (a 32-bit launch beside the persistent stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import opal_amd
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
n = synth.SPROT_SEQS
dev = torch.device("cuda", 0)
lens = synth.lengths(42, 0, n, "sprot")
d_res, offs = synth.residues_device(np.arange(n, dtype=np.int64), lens, 42, dev, stride=synth.SPROT_STRIDE)
torch.cuda.synchronize()
pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
m = blosum50()[1]
out = torch.zeros(3 * n, dtype=torch.int32, device=dev)
def t(Q, mode, st, opts, reps=5, debug=False):
    q = synth.query(Q, seed=7)
    for k, v in opts.items(): opal_amd.set_option(k, v)
    if debug: opal_amd.set_option("OPAL_HIP_DEBUG", "1")
    ks = []
    try:
        for r in range(reps):
            rc, k = pdb.search_device(q, 3, 1, m, st, opal_amd.MODES[mode], out.data_ptr(), out[n:].data_ptr() if st else 0, out[2 * n:].data_ptr() if st else 0)
            ks.append(round(k, 2))
            if debug: opal_amd.set_option("OPAL_HIP_DEBUG", None); debug = False
    finally:
        for k in opts: opal_amd.set_option(k, None)
        opal_amd.set_option("OPAL_HIP_DEBUG", None)
    print(f"Q={Q} {mode} st={st} {opts}: kernel ms {ks}", flush=True)
what = os.environ.get("WHAT", "f4,bimodal")
if "f4" in what:
    for Q in (300, 390):
        t(Q, "SW", 0, {}, debug=True)
        t(Q, "SW", 0, {"OPAL_HIP_NO_UBER": "1"}, debug=True)
        t(Q, "SW", 0, {"OPAL_HIP_SWAP": "0"}, debug=True)
        t(Q, "SW", 0, {"OPAL_HIP_STREAM": "1"})
if "bimodal" in what:
    for mode in ("HW", "NW", "OV"):
        for st in (0, 1):
            t(1000, mode, st, {}, reps=8)
