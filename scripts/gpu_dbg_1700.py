import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import opal_amd
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
n = 100000
dev = torch.device("cuda", 0)
lens = synth.lengths(42, 0, n, "sprot")
d_res, offs = synth.residues_device(np.arange(n, dtype=np.int64), lens, 42, dev, stride=synth.SPROT_STRIDE)
torch.cuda.synchronize()
pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
m = blosum50()[1]
q = synth.query(1700, seed=7)
out = torch.zeros(n, dtype=torch.int32, device=dev)
for name, o in (("natural", {}), ("narrow=16", {"OPAL_HIP_NARROW": "16"}), ("no_uber", {"OPAL_HIP_NO_UBER": "1"}), ("natural", {}), ("stream_prio=0", {"OPAL_HIP_STREAM_PRIO": "0"}), ("natural", {})):
    for k, v in o.items(): opal_amd.set_option(k, v)
    ks = []
    for rep in range(5):
        if rep == 1 and name != "x": opal_amd.set_option("OPAL_HIP_DEBUG", "1" if rep == 1 else None)
        rc, k = pdb.search_device(q, 3, 1, m, 0, 3, out.data_ptr())
        opal_amd.set_option("OPAL_HIP_DEBUG", None)
        ks.append(k)
    for k in o: opal_amd.set_option(k, None)
    print(name, np.round(ks, 2), flush=True)
