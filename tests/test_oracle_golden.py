"""The CPU oracle (oracle/opal_oracle.c) against the reference's own outputs.

Pins the oracle: every golden vector recorded from the reference's SSE4.1/AVX2 library
(tests/golden/make_golden.py), the known answers of SURVEY.md 8(c), and -- when the build
container's oracle/_ref/libopal_ref.so is present -- a live differential run.
"""
import os

import numpy as np
import pytest

import cases
import golden_io
import oracle_lib
from oracle_lib import MODES


def _mask_sw_zero(mode, score):
    """SW rows with score 0 have undefined end locations in the reference."""
    return (score != 0) if mode == "SW" else np.ones_like(score, dtype=bool)


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_oracle_matches_golden(name, oracle, golden):
    case = cases.CASES[name]
    query, seqs, matrix = cases.materialise(case)
    g = golden[name]
    assert golden_io.input_sha1(query, seqs) == g["input_sha1"], "input generator drifted from the golden inputs"
    for mode in case["modes"]:
        rc, score, eq, et = oracle.search(query, seqs, case["gap_open"], case["gap_ext"], matrix,
                                          case["search_type"], MODES[mode])
        want = g["runs"][mode]
        assert rc == want["rc"]
        np.testing.assert_array_equal(score, want["score"], err_msg=f"{name} {mode} score")
        ok = _mask_sw_zero(mode, score)
        np.testing.assert_array_equal(eq[ok], want["endQ"][ok], err_msg=f"{name} {mode} endQ")
        np.testing.assert_array_equal(et[ok], want["endT"][ok], err_msg=f"{name} {mode} endT")
    if case.get("char_sw"):
        rc, score, sset = oracle.char_sw(query, seqs, case["gap_open"], case["gap_ext"], matrix)
        want = g["char_sw"]
        assert rc == want["rc"]
        np.testing.assert_array_equal(sset, want["scoreSet"])
        np.testing.assert_array_equal(score, want["score"])


def test_known_answers_survey_8c(oracle):
    """SURVEY.md 8(c) pins: O74807 x uniprot_sprot15, BLOSUM50, gaps 3/1, score(endQ,endT)."""
    q, db, m = cases.materialise(cases.CASES["td_O74807_sprot15_o3"])
    rc, s, eq, et = oracle.search(q, db, 3, 1, m, 1, MODES["SW"])
    assert rc == 0
    assert s.tolist() == [155, 178, 169, 119, 152, 68, 143, 151, 182, 94, 126, 169, 138, 181, 108]
    assert list(zip(eq.tolist(), et.tolist()))[:4] == [(106, 161), (109, 229), (107, 314), (108, 145)]
    rc, s, eq, et = oracle.search(q, db, 3, 1, m, 1, MODES["NW"])
    assert s.tolist() == [111, 95, -3, 105, 26, 52, 107, 142, 86, 74, 110, 28, 66, -354, 87]
    assert (eq == 109).all() and et.tolist() == [len(x) - 1 for x in db]
    rc, s, eq, et = oracle.search(q, db, 3, 1, m, 1, MODES["HW"])
    assert s.tolist() == [150, 176, 168, 118, 152, 53, 138, 148, 180, 75, 121, 169, 137, 177, 94]
    assert et.tolist() == [163, 229, 318, 145, 437, 58, 215, 187, 304, 73, 125, 399, 322, 930, 84]
    rc, s, eq, et = oracle.search(q, db, 3, 1, m, 1, MODES["OV"])
    assert s.tolist() == [150, 176, 168, 118, 152, 64, 140, 148, 180, 94, 124, 169, 137, 177, 108]
    assert list(zip(eq.tolist(), et.tolist()))[5:7] == [(103, 59), (108, 216)]
    # test1: ACCTCG x AACCGCTG, SW = 47 ending at (5, 7)
    q, db, m = cases.materialise(cases.CASES["td_test1_test1"])
    rc, s, eq, et = oracle.search(q, db, 3, 1, m, 1, MODES["SW"])
    assert (s[0], eq[0], et[0]) == (47, 5, 7)


def test_oracle_api_rules(oracle):
    q, db, m = cases.materialise(cases.CASES["readme_example"])
    rc, s, eq, et = oracle.search(q, db, 3, 1, m, 0, MODES["SW"])
    assert rc == 0 and (eq == -1).all() and (et == -1).all() and s.tolist() == [14, 10, 16, 12]
    rc, *_ = oracle.search(q, db, 3, 1, m, 0, 7)
    assert rc == 3  # OPAL_ERR_INVALID_MODE
    big = m.copy(); big[0, 0] = 2 ** 30
    rc, *_ = oracle.search(q, db, 3, 1, big, 0, MODES["SW"])
    assert rc == 1  # matrix entry outside (INT_MIN/2, INT_MAX/2)
    # SW score 0 -> end (-1,-1)
    rc, s, eq, et = oracle.search([0, 0, 0], [[1, 1, 1, 1]], 3, 1, cases.simple_matrix(4, 2, -1), 1, MODES["SW"])
    assert (s[0], eq[0], et[0]) == (0, -1, -1)
    # 32-bit wrap in SW is reported (three matches of 2^30-2; two still fit)
    mm = cases.simple_matrix(4, 2 ** 30 - 2, -1)
    rc, *_ = oracle.search([1, 1, 1], [[1, 1, 1]], 3, 1, mm, 0, MODES["SW"])
    assert rc == 1
    rc, s, *_ = oracle.search([1, 1], [[1, 1]], 3, 1, mm, 0, MODES["SW"])
    assert rc == 0 and s[0] == 2 ** 31 - 4


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_oracle_vs_live_reference(seed, oracle):
    """Differential run against the compiled reference on fresh random protein sets."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    ref = oracle_lib.RefLib()
    res, offs, lens = synth.database(150, seed=seed)
    db = [res[offs[i]:offs[i + 1]] for i in range(150)]
    q = synth.query(90 + 40 * (seed % 3), seed=seed)
    m = blosum50()[1]
    for mode in cases.ALL_MODES:
        rc, s, eq, et = oracle.search(q, db, 3, 1, m, 1, MODES[mode])
        rrc, rs, req, ret, _, _ = ref.search(q, db, 3, 1, m, 1, MODES[mode])
        assert rc == rrc == 0
        np.testing.assert_array_equal(s, rs)
        ok = _mask_sw_zero(mode, s)
        np.testing.assert_array_equal(eq[ok], req[ok])
        np.testing.assert_array_equal(et[ok], ret[ok])


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_int32_wrap_rule_vs_live_reference(oracle):
    """The SW 32-bit overflow rule (ulH + P > INT_MAX) agrees with the reference's wrap test."""
    ref = oracle_lib.RefLib()
    mm = cases.simple_matrix(4, 2 ** 30 - 2, -1)
    for n, want_rc in ((2, 0), (3, 1)):
        rc, s, *_ = oracle.search([1] * n, [[1] * n], 3, 1, mm, 0, MODES["SW"])
        rrc, rs, *_ = ref.search([1] * n, [[1] * n], 3, 1, mm, 0, MODES["SW"])
        assert rc == rrc == want_rc
        if not rc:
            assert s[0] == rs[0]


def test_oracle_sanitized(tmp_path):
    """The restatement is UBSan-clean: the -fsanitize=undefined build (oracle/Makefile,
    liboracle_san.so, aborting on the first report) runs a spread of golden cases, incl. the 32-bit escalation pins
    and the degenerate shapes, in a fresh process."""
    import subprocess
    import sys
    subprocess.check_call(["make", "-s", "-C", oracle_lib.ORACLE_DIR, "liboracle_san.so"])
    script = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, cases, oracle_lib
orc = oracle_lib.Oracle(%r)
for name in ("readme_example", "td_test1_test1", "pin_100_same_match3", "pin_2000_same_match30_int32", "random4_q200_db300_o0",
             "synth_q37_db700_o3", "c1_O74807_sprot15_sw_score"):
    case = cases.CASES[name]
    q, db, m = cases.materialise(case)
    for mode in case["modes"]:
        rc, s, eq, et = orc.search(q, db, case["gap_open"], case["gap_ext"], m, case["search_type"], oracle_lib.MODES[mode])
    if case.get("char_sw"):
        orc.char_sw(q, db, case["gap_open"], case["gap_ext"], m)
# degenerate shapes and extreme arguments
m = cases.simple_matrix(4, 2**30 - 2, -1)
for mode in range(4):
    orc.search([], [[1, 2, 3], []], 3, 1, cases.simple_matrix(4, 2, -1), 1, mode)
    orc.search([1, 1, 1], [[1, 1, 1], []], 2**30, 2**30, m, 1, mode)
print("sanitized ok")
""" % (os.path.dirname(oracle_lib.ORACLE_DIR), os.path.join(os.path.dirname(oracle_lib.ORACLE_DIR), "tests"),
       os.path.join(oracle_lib.ORACLE_DIR, "liboracle_san.so"))
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sanitized ok" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr
