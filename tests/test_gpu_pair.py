"""One pair over several waves (opal_amd/csrc/pair_kernel.hpp) against the oracle: the longest targets of a NW / HW / OV search
each on a workgroup whose waves share the query's rows (W x 64 lanes x 2 / 4 / 8 rows, the bottom row of a wave's rows handed to
the wave below through a boundary row, wave w + 1 running 96 steps behind wave w). Forced onto the N longest targets
(OPAL_HIP_PAIR=N) of a database with targets of 1 .. 20 000 residues -- shorter than the lag between two waves, exactly on and
around the lane-pipeline and barrier boundaries, copies of the query at the very start and end -- for queries from one wave's
worth of rows to sixteen waves' worth, four gap settings (0/0: ties everywhere), score only and with end cells; and the
planner's own choice on a database with a real length tail."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import MODES

pytestmark = pytest.mark.gpu

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opal_amd", "csrc", "libopal.so")


@pytest.fixture(scope="module")
def ours():
    return oracle_lib.OpalABI(LIB)


def _paired(ours):
    buf = (C.c_longlong * 11)()
    ours.lib.opalx_counters(buf, 11)
    return int(buf[10])


def _database(q, seed):
    from opal_amd import synth
    rng = np.random.RandomState(seed)
    res, offs, lens = synth.database(200, seed=seed)
    db = [res[offs[i]:offs[i + 1]] for i in range(200)]
    longs = [1, 2, 63, 64, 65, 95, 96, 97, 159, 160, 161, 500, 1023, 1024, 1025, 3000, 7001, 20000]
    for n, L in enumerate(longs):
        t = synth.query(L, seed=2000 + seed + n).copy()
        if n % 2 == 0 and L >= len(q) >= 8:
            cp = q.copy()
            hit = rng.rand(len(cp)) < 0.1
            cp[hit] = rng.randint(0, 20, hit.sum())
            at = (L - len(cp)) // 2 if n % 4 == 0 else 0
            t[at:at + len(cp)] = cp
        if n % 3 == 0 and L > 2 * len(q):
            t[L - len(q):] = q
        db.append(t)
    order = rng.permutation(len(db))
    return [db[i] for i in order]


@pytest.mark.parametrize("mode", ["NW", "HW", "OV"])
@pytest.mark.parametrize("qlen", [1, 33, 128, 129, 300, 513, 1000, 1537, 2000, 2560])
def test_pair_kernel_vs_oracle(qlen, mode, ours, oracle, opal_opt):
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    q = synth.query(qlen, seed=qlen)
    db = _database(q, seed=qlen + 3)
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_PAIR", "24")  # the 18 long ones and a few ordinary sequences
    for go, ge in ((11, 1), (3, 1), (0, 0), (5, 5)):
        for st in (1, 0):
            before = _paired(ours)
            got = ours.search(q, db, go, ge, m, st, MODES[mode])[:4]
            assert _paired(ours) - before == 24, "the pair kernel did not run"
            want = oracle.search(q, db, go, ge, m, st, MODES[mode])
            assert got[0] == want[0] == 0
            np.testing.assert_array_equal(got[1], want[1], err_msg=f"pair q{qlen} {mode} gaps {go}/{ge} st={st}: score")
            np.testing.assert_array_equal(got[2], want[2], err_msg=f"pair q{qlen} {mode} gaps {go}/{ge} st={st}: endQ")
            np.testing.assert_array_equal(got[3], want[3], err_msg=f"pair q{qlen} {mode} gaps {go}/{ge} st={st}: endT")


@pytest.mark.parametrize("mode", ["NW", "HW", "OV"])
def test_pair_kernel_with_an_asymmetric_matrix_and_scores_beyond_16_bits(mode, ours, oracle, opal_opt):
    """S(a, b) keeps a = query letter (an asymmetric matrix tells); a matrix scaled so that the long targets' scores leave the
    16-bit range (the kernel is the 32-bit rung for them)."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    m = blosum50()[1].astype(np.int32).copy()
    m += np.random.RandomState(5).randint(-2, 3, m.shape).astype(np.int32)
    m *= 9
    q = synth.query(700, seed=12)
    db = _database(q, seed=41)
    opal_opt.set("OPAL_HIP_PAIR", str(len(db)))  # every sequence: with this matrix most of them are beyond the 16-bit bounds, and those go together
    for st in (0, 1):
        before = _paired(ours)
        got = ours.search(q, db, 30, 9, m, st, MODES[mode])[:4]
        assert _paired(ours) - before == len(db)
        want = oracle.search(q, db, 30, 9, m, st, MODES[mode])
        assert got[0] == want[0] == 0
        assert np.abs(want[1]).max() > 40000
        for a, b, what in zip(got[1:], want[1:], ("score", "endQ", "endT")):
            np.testing.assert_array_equal(a, b, err_msg=f"{mode} st={st}: {what}")


def test_planner_sends_the_tail_of_a_global_search_to_the_pair_kernel(ours, oracle):
    """No forcing: a 1500-aa HW / NW search of 3000 sequences with four targets of 9 000 - 33 000 residues (beyond the 16-bit
    bounds: gap extension alone leaves the range) -- the planner's own choice must take them (counter), answers the oracle's."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(3000, seed=8)
    db = [res[offs[i]:offs[i + 1]] for i in range(3000)]
    for n, L in enumerate((33000, 32500, 15000, 9000)):
        db[100 + n] = synth.query(L, seed=50 + n)
    q = synth.query(1500, seed=3)
    m = blosum50()[1]
    pick = np.unique(np.concatenate([np.arange(95, 110), np.random.RandomState(4).choice(3000, 40, replace=False)]))
    sdb = [db[i] for i in pick]
    for mode in ("HW", "NW"):
        before = _paired(ours)
        got = ours.search(q, db, 3, 1, m, 1, MODES[mode])[:4]
        assert _paired(ours) - before >= 2, "the planner left the 32-bit targets off the pair kernel"
        want = oracle.search(q, sdb, 3, 1, m, 1, MODES[mode])
        assert got[0] == want[0] == 0
        for a, b, what in zip(got[1:], want[1:], ("score", "endQ", "endT")):
            np.testing.assert_array_equal(a[pick], b, err_msg=f"{mode}: {what}")
