"""Which of the plans disagrees with the oracle: chunked global-mode queries, natural plan against OPAL_HIP_NARROW=16 / 18."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import opal_amd, oracle_lib
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
n = int(os.environ.get("NSEQ", "20000")); Q = int(os.environ.get("QLEN", "2600"))
res, offs, lens = synth.database(n, seed=42)
pdb = PreparedDb(res, offs)
q = synth.query(Q, seed=7); m = blosum50()[1]
orc = oracle_lib.Oracle()
order = np.argsort(lens)
pick = np.unique(np.concatenate([order[-60:], order[:20], np.random.RandomState(1).choice(n, 60, replace=False)]))
db = [res[offs[i]:offs[i + 1]] for i in pick]
for mode in ("NW", "HW", "OV"):
    for st in (0, 1):
        want = orc.search(q, db, 11, 1, m, st, oracle_lib.MODES[mode])
        outs = {}
        for name, opt in (("natural", None), ("narrow=16", "16"), ("narrow=18", "18"), ("narrow=17", "17")):
            if opt: opal_amd.set_option("OPAL_HIP_NARROW", opt)
            try:
                rc, s, eq, et = pdb.search(q, 11, 1, m, st, oracle_lib.MODES[mode])
            finally:
                opal_amd.set_option("OPAL_HIP_NARROW", None)
            outs[name] = (rc, s, eq, et)
            bad = np.nonzero(s[pick] != want[1])[0]
            badq = np.nonzero(eq[pick] != want[2])[0] if st else []
            badt = np.nonzero(et[pick] != want[3])[0] if st else []
            print(f"{mode} st={st} {name:10s} rc={rc} oracle rc={want[0]}: score mismatches {len(bad)} endQ {len(badq)} endT {len(badt)}",
                  (f"first: seq {pick[bad[0]]} len {lens[pick[bad[0]]]} got {s[pick[bad[0]]]} want {want[1][bad[0]]}" if len(bad) else ""),
                  (f"endT first: seq {pick[badt[0]]} len {lens[pick[badt[0]]]} got {et[pick[badt[0]]]} want {want[3][badt[0]]}" if len(badt) else ""), flush=True)
        a = outs["natural"]
        for name in ("narrow=16", "narrow=18", "narrow=17"):
            b = outs[name]
            d = np.nonzero((a[1] != b[1]) | (a[2] != b[2]) | (a[3] != b[3]))[0]
            if len(d):
                print(f"   natural vs {name}: {len(d)} sequences differ; lengths of the first: {lens[d[:8]]}, idx {d[:8]}, natural {a[1][d[:4]]},{a[3][d[:4]]} other {b[1][d[:4]]},{b[3][d[:4]]}")
