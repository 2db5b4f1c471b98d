"""Quick GPU sanity script: a few parity cases with mismatch printouts, then a timing of the C2 workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases, oracle_lib
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50

ours = oracle_lib.OpalABI(os.path.join(ROOT, "opal_amd", "csrc", "libopal.so"))
orc = oracle_lib.Oracle()
names = sys.argv[1:] or ["readme_example", "td_test1_test1", "td_O74807_sprot15_o3", "pin_100_same_match3", "synth_q300_db2000_o3"]
for name in names:
    case = cases.CASES[name]
    q, db, m = cases.materialise(case)
    for mode in case["modes"]:
        rc, s, eq, et, sset, _ = ours.search(q, db, case["gap_open"], case["gap_ext"], m, case["search_type"], oracle_lib.MODES[mode])
        orc_rc, os_, oq, ot = orc.search(q, db, case["gap_open"], case["gap_ext"], m, case["search_type"], oracle_lib.MODES[mode])
        bad = np.nonzero((s != os_) | (eq != oq) | (et != ot))[0]
        print(f"{name:32s} {mode} rc={rc}/{orc_rc} mismatches={len(bad)}/{len(s)}", flush=True)
        for i in bad[:5]:
            print(f"   #{i} len={len(db[i])} got=({s[i]},{eq[i]},{et[i]}) want=({os_[i]},{oq[i]},{ot[i]})")
if os.environ.get("TIMING", "1") == "1":
    n = 100000
    res, offs, lens = synth.database(n, seed=42)
    q = synth.query(300, seed=7); m = blosum50()[1]
    t0 = time.time(); pdb = PreparedDb(res, offs); print("pack+upload s:", time.time() - t0)
    import torch
    score = torch.zeros(n, dtype=torch.int32, device="cuda")
    for g in os.environ.get("GLIST", "auto").split(","):
        if g != "auto": os.environ["OPAL_HIP_LANES_PER_SEQ"] = g
        for it in range(3):
            rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
        print(f"G={g}: rc={rc} kernel_ms={kms:.3f} GCUPS={300.0 * offs[-1] / kms / 1e6:.1f}", flush=True)
