"""The role-swapped SW kernel (opal_amd/csrc/swap_kernel.hpp) against the oracle: the longest targets of a search with a
short query run with the TARGET on the rows of a wave (row chunks of 64 lanes x 8 / 16 / 24 / 40 residues) and the query on the
columns. Local-alignment scores are symmetric under that transposition (reference recurrence src/opal.cpp:280-327; S(a, b)
keeps a = query letter), so every score must equal the oracle's -- for targets that are an exact number of row chunks, one
residue more or less, a single partial chunk, with near copies of the query straddling chunk boundaries, and with scores
beyond the 16-bit rung (re-run list)."""
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


def _swapped(ours):
    buf = (C.c_longlong * 10)()
    ours.lib.opalx_counters(buf, 10)
    return int(buf[9])


def _database(q, rows, seed):
    """300 ordinary sequences and 14 long ones whose lengths sit on and around the row-chunk size 64 x rows; some carry
    mutated copies of the query across a chunk boundary (the alignment then crosses from one chunk's lanes into the next
    chunk through the boundary row in LDS)."""
    from opal_amd import synth
    rng = np.random.RandomState(seed)
    res, offs, lens = synth.database(300, seed=seed)
    db = [res[offs[i]:offs[i + 1]] for i in range(300)]
    chunk = 64 * rows
    longs = [chunk, chunk - 1, chunk + 1, 2 * chunk, 2 * chunk + 7, 3 * chunk - 5, chunk // 2, chunk + rows, chunk + rows + 1,
             5 * chunk + 123, 7000, 9001, 3 * chunk, 4 * chunk + 1]
    for n, L in enumerate(longs):
        t = synth.query(L, seed=1000 + seed + n).copy()
        if n % 2 == 0 and len(q) >= 8:
            # a mutated copy of the query across the first chunk boundary (or in the middle of a short target)
            cp = q.copy()
            hit = rng.rand(len(cp)) < 0.15
            cp[hit] = rng.randint(0, 20, hit.sum())
            at = min(max(0, (chunk if L > chunk else L // 2) - len(cp) // 2), L - len(cp)) if L >= len(cp) else 0
            t[at:at + min(len(cp), L - at)] = cp[: min(len(cp), L - at)]
        if n % 3 == 0 and L > 2 * len(q):
            t[L - len(q):] = q  # ... and an exact copy at the very end (the last rows of the last chunk)
        db.append(t)
    order = rng.permutation(len(db))
    return [db[i] for i in order]


@pytest.mark.parametrize("rows", ["40", "24", "16", "8"])
@pytest.mark.parametrize("qlen", [1, 7, 64, 110, 195, 320, 700])
def test_swapped_kernel_vs_oracle(qlen, rows, ours, oracle, opal_opt):
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    q = synth.query(qlen, seed=qlen)
    db = _database(q, int(rows), seed=qlen + int(rows))
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_SWAP", "20")  # the 20 longest: the 14 long ones and a few ordinary ones (single partial chunk)
    opal_opt.set("OPAL_HIP_SWAP_ROWS", rows)
    for go, ge in ((11, 1), (3, 1), (0, 0), (5, 5)):
        before = _swapped(ours)
        got = ours.search(q, db, go, ge, m, 0, MODES["SW"])[:4]
        assert _swapped(ours) - before == 20, "the swapped kernel did not run"
        want = oracle.search(q, db, go, ge, m, 0, MODES["SW"])
        assert got[0] == want[0] == 0
        np.testing.assert_array_equal(got[1], want[1], err_msg=f"swapped q{qlen} rows={rows} gaps {go}/{ge}")
        assert (got[2] == -1).all() and (got[3] == -1).all()
        # ... and with end cells (the 8- and 16-row shapes; taller requests fall back to 16 rows): the smallest target index
        # holding the score, then the smallest query index -- rows first, then columns, in the swapped frame
        before = _swapped(ours)
        got = ours.search(q, db, go, ge, m, 1, MODES["SW"])[:4]
        assert _swapped(ours) - before == 20, "the swapped kernel did not run for the search with end cells"
        want = oracle.search(q, db, go, ge, m, 1, MODES["SW"])
        assert got[0] == want[0] == 0
        np.testing.assert_array_equal(got[1], want[1], err_msg=f"swapped + end q{qlen} rows={rows} gaps {go}/{ge}: score")
        live = want[1] != 0
        np.testing.assert_array_equal(got[3][live], want[3][live], err_msg=f"swapped + end q{qlen} rows={rows} gaps {go}/{ge}: endT")
        np.testing.assert_array_equal(got[2][live], want[2][live], err_msg=f"swapped + end q{qlen} rows={rows} gaps {go}/{ge}: endQ")


@pytest.mark.parametrize("rows", ["8", "16"])
@pytest.mark.parametrize("mode", ["NW", "HW", "OV"])
@pytest.mark.parametrize("qlen", [1, 7, 64, 110, 300, 513])
def test_swapped_global_kernels_vs_oracle(qlen, mode, rows, ours, oracle, opal_opt):
    """NW / HW / OV with the roles swapped (opal_swap_gl: int32, the target on the rows, boundary conditions and result rules
    of reference src/opal.cpp:671-689, :844-904 in the transposed frame): the 20 longest targets forced onto it, scores and end
    cells against the oracle -- targets of exactly one / two / three row chunks, one residue more or less, a partial chunk,
    copies of the query across a chunk boundary and at the very end, gaps 11/1, 3/1, 0/0 and 5/5."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    q = synth.query(qlen, seed=qlen)
    db = _database(q, int(rows), seed=qlen + int(rows) + 7)
    db += [synth.query(1, seed=5), synth.query(2, seed=6)]  # (targets of one and two residues stay on the class kernels)
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_SWAP", "20")
    opal_opt.set("OPAL_HIP_SWAP_ROWS", rows)
    for go, ge in ((11, 1), (3, 1), (0, 0), (5, 5)):
        for st in (1, 0):
            before = _swapped(ours)
            got = ours.search(q, db, go, ge, m, st, MODES[mode])[:4]
            assert _swapped(ours) - before == 20, "the swapped kernel did not run"
            want = oracle.search(q, db, go, ge, m, st, MODES[mode])
            assert got[0] == want[0] == 0
            what = f"swapped {mode} q{qlen} rows={rows} gaps {go}/{ge} st={st}"
            np.testing.assert_array_equal(got[1], want[1], err_msg=what + ": score")
            np.testing.assert_array_equal(got[2], want[2], err_msg=what + ": endQ")
            np.testing.assert_array_equal(got[3], want[3], err_msg=what + ": endT")


def test_swapped_global_kernels_take_the_32bit_targets(ours, oracle, opal_opt):
    """Without forcing: targets beyond what 16 bits hold in the global modes (gap extension alone exceeds the range past
    ~32 000 residues) and the long ones behind them go to the swapped kernels; the 32-bit class kernels stay idle."""
    import ctypes as C
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(20000, seed=42)
    db = [res[offs[i]:offs[i + 1]] for i in range(20000)]
    tail = [synth.query(L, seed=L) for L in (35213, 34000, 33000, 22000, 12000)]
    m = blosum50()[1]
    q = synth.query(110, seed=7)
    seqs = db + tail
    idx = list(range(len(seqs) - 5, len(seqs))) + list(range(0, len(seqs) - 5, 397))
    for mode in ("HW", "OV", "NW"):
        buf = (C.c_longlong * 10)(); ours.lib.opalx_counters(buf, 10)
        w32, sw = int(buf[5]), int(buf[9])
        got = ours.search(q, seqs, 3, 1, m, 1, MODES[mode])[:4]
        ours.lib.opalx_counters(buf, 10)
        assert int(buf[9]) - sw >= 3 and int(buf[5]) == w32, (mode, int(buf[9]) - sw, int(buf[5]) - w32)
        want = oracle.search(q, [seqs[i] for i in idx], 3, 1, m, 1, MODES[mode])
        assert got[0] == 0
        np.testing.assert_array_equal(got[1][idx], want[1], err_msg=mode)
        np.testing.assert_array_equal(got[2][idx], want[2], err_msg=mode)
        np.testing.assert_array_equal(got[3][idx], want[3], err_msg=mode)


def test_swapped_kernel_saturated_scores_rerun(ours, oracle, opal_opt):
    """Scores at the 16-bit limit leave the swapped kernel through the re-run list and come back from the 32-bit rung."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    q = synth.query(200, seed=3)
    db = _database(q, 40, seed=77)
    m = blosum50()[1] * 60
    opal_opt.set("OPAL_HIP_SWAP", "20")
    got = ours.search(q, db, 660, 60, m, 0, MODES["SW"])[:4]
    want = oracle.search(q, db, 660, 60, m, 0, MODES["SW"])
    assert got[0] == want[0] == 0 and (want[1] > 32767).sum() >= 3
    np.testing.assert_array_equal(got[1], want[1])


def test_swapped_kernel_is_what_a_long_tail_runs_on(ours, oracle, opal_opt):
    """Without forcing: a short query against a database with a few very long targets sends exactly those to the swapped
    kernel (planner.cpp plan_swap), with and without end locations; with a longer query or no tail it stays away."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(20000, seed=42)
    db = [res[offs[i]:offs[i + 1]] for i in range(20000)]
    tail = [synth.query(L, seed=L) for L in (35213, 30000, 22000, 12000)]
    m = blosum50()[1]
    q = synth.query(110, seed=7)
    opal_opt.unset("OPAL_HIP_SWAP")
    short = [s[:1000] for s in db]  # (no tail at all: nothing outlasts the search on the chain by enough to move)
    for st, seqs, expect in ((0, db + tail, True), (1, db + tail, True), (0, short, False)):
        before = _swapped(ours)
        got = ours.search(q, seqs, 3, 1, m, st, MODES["SW"])[:4]
        n = _swapped(ours) - before
        assert (n >= 3) == expect, (st, len(seqs), n)
        idx = list(range(len(seqs) - 4, len(seqs))) + list(range(0, len(seqs) - 4, 397))
        want = oracle.search(q, [seqs[i] for i in idx], 3, 1, m, st, MODES["SW"])
        assert got[0] == 0
        np.testing.assert_array_equal(got[1][idx], want[1])
        if st:
            np.testing.assert_array_equal(got[2][idx], want[2])
            np.testing.assert_array_equal(got[3][idx], want[3])
    before = _swapped(ours)
    assert ours.search(synth.query(600, seed=7), db + tail, 3, 1, m, 0, MODES["SW"])[0] == 0
    assert _swapped(ours) == before, "a 600-aa query reaches 64 lanes on the class kernels"


@pytest.mark.parametrize("rows", ["8", "16"])
@pytest.mark.parametrize("gaps", [(0, 1), (0, 0), (0, 3), (1, 1)])
def test_swapped_end_cell_with_free_gap_opening_at_the_last_query_column(rows, gaps, ours, oracle, opal_opt):
    """gapOpen = 0: a score reached in the LAST query column flows on, undiminished, down every row below it (F hands it on
    for free) and through the pad columns behind the query (E keeps it). Every lane further down the chunk then holds the
    final score too -- in the real column Q-1 already, never first in a pad column (a pad cell's value has gone one step to the
    right and then down, the cell of column Q-1 in the same row gets it by going straight down: never less) -- and the
    reduction must pick the smallest target index among them, not a cell a lane kept from an earlier row chunk. The query's last
    letter occurs once in each long target (third row chunk), so the score is first reached exactly there (reference tie
    rule: opal.h:44-45)."""
    rng = np.random.RandomState(int(rows) * 10 + gaps[1])
    A = 24
    m = np.full((A, A), -3, np.int32)
    np.fill_diagonal(m, 5)
    q = rng.randint(0, 20, 30).astype(np.uint8)
    q[-1] = 23
    chunk = 64 * int(rows)
    db = [rng.randint(0, 20, int(L)).astype(np.uint8) for L in rng.randint(50, 400, 60)]
    where = []
    for n in range(6):
        t = rng.randint(0, 20, 3 * chunk + 17 * n).astype(np.uint8)
        at = 2 * chunk + 5 + 61 * n  # third chunk, a different lane every time
        t[at] = 23
        where.append(at)
        db.append(t)
    opal_opt.set("OPAL_HIP_SWAP", "6")
    opal_opt.set("OPAL_HIP_SWAP_ROWS", rows)
    before = _swapped(ours)
    got = ours.search(q, db, gaps[0], gaps[1], m, 1, MODES["SW"])[:4]
    assert _swapped(ours) - before == 6
    want = oracle.search(q, db, gaps[0], gaps[1], m, 1, MODES["SW"])
    assert got[0] == want[0] == 0
    if gaps[0] == 0:
        assert list(want[3][-6:]) == where and (want[2][-6:] == len(q) - 1).all(), "the construction: score first reached at the rare letter"
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[3], want[3])
    np.testing.assert_array_equal(got[2], want[2])


@pytest.mark.parametrize("mode", ["SW", "HW", "NW"])
def test_swapped_kernels_in_a_call_sharded_over_devices(mode, ours, oracle, opal_opt):
    """The same long-tailed database through opalSearchDatabase sharded over two device contexts ("0,0": both on this box's
    one GPU): every shard plans its own swapped prefix on its own side stream; results as unsharded, as the oracle's."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(20000, seed=43)
    db = [res[offs[i]:offs[i + 1]] for i in range(20000)]
    tail = [synth.query(L, seed=L) for L in (35213, 34000, 30000, 22000, 12000, 9000)]
    seqs = db + tail
    m = blosum50()[1]
    q = synth.query(110, seed=9)
    opal_opt.unset("OPAL_HIP_SWAP")
    one = ours.search(q, seqs, 3, 1, m, 1, MODES[mode])[:4]
    opal_opt.set("OPAL_HIP_DEVICES", "0,0")
    before = _swapped(ours)
    two = ours.search(q, seqs, 3, 1, m, 1, MODES[mode])[:4]
    assert _swapped(ours) - before >= 4, "the shards did not use the swapped kernels"
    assert one[0] == two[0] == 0
    for a, b in zip(one[1:], two[1:]):
        np.testing.assert_array_equal(a, b)
    idx = list(range(len(seqs) - 6, len(seqs))) + list(range(0, len(seqs) - 6, 397))
    want = oracle.search(q, [seqs[i] for i in idx], 3, 1, m, 1, MODES[mode])
    for a, b in zip(two[1:], want[1:4]):
        np.testing.assert_array_equal(a[idx], b)


@pytest.mark.parametrize("rows", ["8", "16"])
@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_swapped_kernels_with_an_asymmetric_matrix(mode, rows, ours, oracle, opal_opt):
    """The swapped kernels transpose the DP; the score matrix must NOT be transposed with it: S(a, b) keeps a = query letter
    (reference src/opal.cpp:190, opal.h:115-116). Every other case of this file uses a symmetric matrix, which cannot tell
    M[a*A+b] from M[b*A+a]; this one is BLOSUM50 with a seeded, asymmetric perturbation of every entry. Scores and end cells
    of the 20 longest targets forced onto the swapped kernels, against the oracle."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    m = blosum50()[1].astype(np.int32).copy()
    rng = np.random.RandomState(91)
    m += rng.randint(-3, 4, m.shape).astype(np.int32)
    assert (m != m.T).sum() > 100
    q = synth.query(110, seed=4)
    db = _database(q, int(rows), seed=77 + int(rows))
    opal_opt.set("OPAL_HIP_SWAP", "20")
    opal_opt.set("OPAL_HIP_SWAP_ROWS", rows)
    for go, ge in ((11, 1), (3, 1)):
        for st in (0, 1):
            before = _swapped(ours)
            got = ours.search(q, db, go, ge, m, st, MODES[mode])[:4]
            assert _swapped(ours) - before == 20, "the swapped kernel did not run"
            want = oracle.search(q, db, go, ge, m, st, MODES[mode])
            assert got[0] == want[0] == 0
            np.testing.assert_array_equal(got[1], want[1], err_msg=f"asymmetric matrix, {mode} rows={rows} gaps {go}/{ge}: score")
            if st:
                live = want[1] != 0 if mode == "SW" else np.ones(len(db), bool)
                np.testing.assert_array_equal(got[2][live], want[2][live], err_msg=f"asymmetric matrix, {mode}: endQ")
                np.testing.assert_array_equal(got[3][live], want[3][live], err_msg=f"asymmetric matrix, {mode}: endT")
