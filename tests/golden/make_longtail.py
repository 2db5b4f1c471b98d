#!/usr/bin/env python3
"""Record the reference's answers for the long-tail targets of tests/longtail.py -> tests/golden/longtail.json.

Runs ONLY in the build container (needs oracle/_ref/libopal_ref.so = /root/reference/src/opal.cpp compiled by
oracle/build_ref.sh). For every query length of longtail.QUERY_LENGTHS and every mode, one OPAL_SEARCH_ALIGNMENT
search of the 41 planted sequences (8 000 - 35 213 residues) through the reference library:
score, end cell, start cell, length and crc32 of the operation string per sequence (the strings themselves would be
18 MB). NW is the exception: targets of this length need the reference's 32-bit NW path, which is undefined behaviour
(it returns OPAL_ERR_OVERFLOW here), so NW scores and end cells come from the scalar oracle
(oracle/opal_oracle.c = the reference's own ground truth, test.cpp:253-328) and NW has no recorded alignment.
The script asserts that the oracle agrees with the reference wherever the reference answers.

Usage: python tests/golden/make_longtail.py
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import longtail  # noqa: E402
import oracle_lib  # noqa: E402
from opal_amd.score_matrix import blosum50  # noqa: E402


def main():
    ref = oracle_lib.RefLib()
    orc = oracle_lib.Oracle()
    m = blosum50()[1]
    out = {"gap_open": longtail.GAP_OPEN, "gap_ext": longtail.GAP_EXT, "lengths": longtail.PLANT_LENGTHS + [35000], "queries": {}}
    for qlen in longtail.QUERY_LENGTHS:
        q = longtail.query(qlen)
        db = longtail.planted(q)
        per = {"input_crc32": zlib.crc32(b"".join(np.ascontiguousarray(s).tobytes() for s in [q] + db))}
        for mode in ("SW", "HW", "OV", "NW"):
            orc_rc, os_, oq, ot = orc.search(q, db, longtail.GAP_OPEN, longtail.GAP_EXT, m, 1, oracle_lib.MODES[mode])
            assert orc_rc == 0
            if mode == "NW":
                per[mode] = {"source": "oracle", "score": os_.tolist(), "endQ": oq.tolist(), "endT": ot.tolist()}
                continue
            rc, s, eq, et, sset, (arr, _) = ref.search(q, db, longtail.GAP_OPEN, longtail.GAP_EXT, m, oracle_lib.SEARCH_ALIGNMENT,
                                                       oracle_lib.MODES[mode])
            sq, st, ops = ref.alignments(arr)
            assert rc == 0 and (s == os_).all() and (eq == oq).all() and (et == ot).all(), f"oracle != reference: Q={qlen} {mode}"
            for i in range(len(db)):
                assert orc.check_alignment(q, db[i], longtail.GAP_OPEN, longtail.GAP_EXT, m, s[i], sq[i], st[i], eq[i], et[i],
                                           np.frombuffer(ops[i], np.uint8)) == 0
            per[mode] = {"source": "reference", "score": s.tolist(), "endQ": eq.tolist(), "endT": et.tolist(),
                         "startQ": sq.tolist(), "startT": st.tolist(), "opsLen": [len(o) for o in ops],
                         "opsCrc32": [zlib.crc32(o) for o in ops]}
            print(f"Q={qlen} {mode}: max score {int(s.max())}, longest alignment {max(len(o) for o in ops)}", flush=True)
        out["queries"][str(qlen)] = per
    with open(os.path.join(HERE, "longtail.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote tests/golden/longtail.json")


if __name__ == "__main__":
    main()
