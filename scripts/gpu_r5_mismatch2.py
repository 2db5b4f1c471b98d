import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import opal_amd, oracle_lib
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
Q = 2600
q = synth.query(Q, seed=7); m = blosum50()[1]
orc = oracle_lib.Oracle()
def run(n, opts, mode="NW", st=0, nsel=None):
    res, offs, lens = synth.database(n, seed=42)
    if nsel is not None:   # keep only the nsel longest + everything (test: is it the count of items?)
        pass
    pdb = PreparedDb(res, offs)
    order = np.argsort(-lens.astype(np.int64), kind="stable")
    pick = order[:24]
    db = [res[offs[i]:offs[i + 1]] for i in pick]
    want = orc.search(q, db, 11, 1, m, st, oracle_lib.MODES[mode])
    for k, v in opts.items(): opal_amd.set_option(k, v)
    try:
        rc, s, eq, et = pdb.search(q, 11, 1, m, st, oracle_lib.MODES[mode])
    finally:
        for k in opts: opal_amd.set_option(k, None)
    bad = [int(i) for i in np.nonzero(s[pick] != want[1])[0]]
    print(f"n={n} {mode} st={st} {opts}: rc={rc} bad items (of the 24 longest) {bad}  lens {[int(lens[pick[i]]) for i in bad]}", flush=True)
for n in (3000, 6000, 12000, 20000):
    run(n, {"OPAL_HIP_NARROW": "18"})
run(20000, {"OPAL_HIP_NARROW": "18", "OPAL_HIP_NO_SIDE_PREFIX": "1"})
run(20000, {"OPAL_HIP_NARROW": "18", "OPAL_HIP_CHUNK_GEOMETRY": "32,24"})
run(20000, {"OPAL_HIP_NARROW": "18", "OPAL_HIP_CHUNK_GEOMETRY": "64,24"})
run(20000, {"OPAL_HIP_NARROW": "18", "OPAL_HIP_CHUNK_GEOMETRY": "16,32"})
run(20000, {"OPAL_HIP_NARROW": "18", "OPAL_HIP_DEBUG": "1"})
run(20000, {"OPAL_HIP_STREAM": "0"})
run(20000, {"OPAL_HIP_NO_STREAM_CHUNKS": "1"})
run(20000, {"OPAL_HIP_FORCE_WIDTH": "32"})
