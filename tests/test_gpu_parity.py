"""GPU parity tests: the HIP path, called through the C ABI (opalSearchDatabase in
opal_amd/csrc/libopal.so), against (a) the committed golden vectors recorded from the reference's
own SSE4.1/AVX2 library and (b) the CPU oracle on the same inputs. Integer results: bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import golden_io
import oracle_lib
from oracle_lib import MODES

pytestmark = pytest.mark.gpu

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opal_amd", "csrc", "libopal.so")


@pytest.fixture(scope="module")
def ours():
    assert os.path.exists(LIB), "libopal.so not built"
    return oracle_lib.OpalABI(LIB)


def _check(got, want, mode, what):
    rc, s, eq, et = got
    wrc, ws, weq, wet = want
    assert rc == wrc, f"{what}: rc {rc} != {wrc}"
    np.testing.assert_array_equal(s, ws, err_msg=f"{what} score")
    ok = (ws != 0) if mode == "SW" else np.ones_like(ws, dtype=bool)  # SW score 0: reference end is undefined
    np.testing.assert_array_equal(eq[ok], weq[ok], err_msg=f"{what} endQ")
    np.testing.assert_array_equal(et[ok], wet[ok], err_msg=f"{what} endT")


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_case_vs_golden_and_oracle(name, ours, oracle, golden):
    case = cases.CASES[name]
    query, seqs, matrix = cases.materialise(case)
    g = golden[name]
    for mode in case["modes"]:
        rc, s, eq, et, sset, _ = ours.search(query, seqs, case["gap_open"], case["gap_ext"], matrix,
                                             case["search_type"], MODES[mode], case.get("overflow", 1))
        assert (sset == 1).all()
        want = g["runs"][mode]
        _check((rc, s, eq, et), (want["rc"], want["score"], want["endQ"], want["endT"]), mode, f"{name}/{mode} vs golden")
        orc = oracle.search(query, seqs, case["gap_open"], case["gap_ext"], matrix, case["search_type"], MODES[mode])
        rc2, s2, eq2, et2 = orc
        assert rc == rc2
        np.testing.assert_array_equal(s, s2)
        np.testing.assert_array_equal(eq, eq2)  # including our defined (-1,-1) for SW score 0
        np.testing.assert_array_equal(et, et2)
    if case.get("char_sw"):
        rc, s, sset = ours.char_sw(query, seqs, case["gap_open"], case["gap_ext"], matrix)
        want = g["char_sw"]
        assert rc == want["rc"]
        np.testing.assert_array_equal(sset, want["scoreSet"])
        np.testing.assert_array_equal(s, want["score"])


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64])
@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_every_wave_geometry(lanes, mode, ours, oracle, opal_opt):
    """Force each lanes-per-sequence value: results must not depend on how rows are spread over lanes."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    qlen = {1: 33, 2: 61, 4: 97, 8: 200, 16: 300, 32: 513, 64: 700}[lanes]
    res, offs, lens = synth.database(140, seed=100 + lanes)
    db = [res[offs[i]:offs[i + 1]] for i in range(140)]
    q = synth.query(qlen, seed=lanes)
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", str(lanes))
    rc, s, eq, et, _, _ = ours.search(q, db, 3, 1, m, 1, MODES[mode])
    want = oracle.search(q, db, 3, 1, m, 1, MODES[mode])
    _check((rc, s, eq, et), want, mode, f"G={lanes} {mode}")


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64])
def test_sw_end_cells_with_ties_in_every_wave_geometry(lanes, ours, oracle, opal_opt):
    """SW end cells where ties are the rule: two-letter sequences, match +1 / mismatch -1, gaps 0/0, 1/0, 0/1 and 2/1 -- the
    final score is reached in many cells of many lanes, in the same step on different columns and in later steps on SMALLER
    columns (a lane further down the slot sits on an earlier column). The single-launch kernels settle the slot's best cell
    by an all-reduce over its lanes (dp_kernel.hpp, CKPT: DPP up to 16 lanes, the LDS crossbar beyond) and look the row up in
    the column they kept in LDS: smallest target index, then smallest query index (opal.h:44-45), for every lanes-per-sequence
    value and sequences shorter than the lane pipeline."""
    rng = np.random.RandomState(lanes)
    qlen = {1: 29, 2: 61, 4: 97, 8: 200, 16: 300, 32: 513, 64: 700}[lanes]
    A = 4
    m = np.full((A, A), -1, np.int32)
    np.fill_diagonal(m, 1)
    q = rng.randint(0, 2, qlen).astype(np.uint8)
    db = [rng.randint(0, 2, int(L)).astype(np.uint8) for L in rng.randint(1, 700, 150)]
    db += [np.tile(q[:7], 40)[: 250 + 3 * i].copy() for i in range(6)]   # periodic: the same score over and over
    db += [q.copy(), q[::-1].copy(), np.zeros(300, np.uint8), np.ones(2, np.uint8)]
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", str(lanes))
    for go, ge in ((0, 0), (1, 0), (0, 1), (2, 1)):
        got = ours.search(q, db, go, ge, m, 1, MODES["SW"])[:4]
        want = oracle.search(q, db, go, ge, m, 1, MODES["SW"])
        _check(got, want, "SW", f"ties G={lanes} gaps {go}/{ge}")


@pytest.mark.parametrize("rows", [15, 16, 22, 23, 24, 25, 33, 36, 37, 38, 39, 40])
@pytest.mark.parametrize("stream", [False, True])
def test_sw_end_cells_with_ties_in_every_row_grouping(rows, stream, ours, oracle, opal_opt):
    """The kept rows of a slot's best cell are ONE group of a lane's row blocks, and the number of groups depends on the rows
    per lane (four; two or one where four accumulators would cost a wave per SIMD: dp_kernel.hpp, opal_swend_groups): every
    grouping, in the class kernels and -- up to the 36 rows they exist for -- in the streaming ones, on the tie-ridden inputs."""
    if stream and rows > 36:
        pytest.skip("streaming kernels with end locations stop at 36 rows per lane")
    rng = np.random.RandomState(500 + rows)
    lanes = 8
    qlen = lanes * rows - 3
    A = 4
    m = np.full((A, A), -1, np.int32)
    np.fill_diagonal(m, 1)
    q = rng.randint(0, 2, qlen).astype(np.uint8)
    db = [rng.randint(0, 2, int(L)).astype(np.uint8) for L in rng.randint(1, 500, 260)]
    db += [np.tile(q[:7], 60)[: 250 + 3 * i].copy() for i in range(6)]
    db += [q.copy(), q[::-1].copy(), np.zeros(300, np.uint8), np.ones(2, np.uint8)]
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", str(lanes))
    opal_opt.set("OPAL_HIP_NARROW", "17")
    opal_opt.set("OPAL_HIP_STREAM", "1" if stream else "0")
    if stream:
        opal_opt.set("OPAL_HIP_STREAM_BLOCKS", "1")
    for go, ge in ((0, 0), (1, 0), (2, 1), (11, 1)):
        got = ours.search(q, db, go, ge, m, 1, MODES["SW"])[:4]
        want = oracle.search(q, db, go, ge, m, 1, MODES["SW"])
        _check(got, want, "SW", f"ties rows={rows} stream={stream} gaps {go}/{ge}")


@pytest.mark.parametrize("lanes,qlen,narrow", [(4, 97, "17"), (8, 200, "17"), (16, 300, "17"), (16, 520, "17"), (32, 513, "17"),
                                               (64, 700, "17"), (64, 1300, "18"), (64, 1500, "18"), (64, 2300, "18")])
def test_sw_end_cells_with_ties_in_the_streaming_kernels(lanes, qlen, narrow, ours, oracle, opal_opt):
    """The same where sequences follow each other through a slot's lanes (stream_kernel.hpp): the slot's best cell is kept per
    SEQUENCE IN FLIGHT -- the lanes above a separator already work on the next one, whose best the lanes below must learn before
    they get there, and the two kept columns sit in two LDS buffers by the parity of the sequence number. Ties in every column,
    sequences shorter than the lane pipeline (pad columns between entries), one workgroup so that every slot sees many."""
    rng = np.random.RandomState(1000 + lanes)
    A = 4
    m = np.full((A, A), -1, np.int32)
    np.fill_diagonal(m, 1)
    q = rng.randint(0, 2, qlen).astype(np.uint8)
    n = max(200, 4 * 16 * 64 // lanes)
    db = [rng.randint(0, 2, int(L)).astype(np.uint8) for L in rng.randint(1, 2 * max(lanes, 100) + 40, n)]
    db += [np.tile(q[:7], 60)[: 250 + 3 * i].copy() for i in range(6)]   # periodic: the same score over and over
    db += [q.copy(), q[::-1].copy(), np.zeros(300, np.uint8), np.ones(2, np.uint8), np.ones(1, np.uint8)]
    order = rng.permutation(len(db))
    db = [db[i] for i in order]
    opal_opt.set("OPAL_HIP_STREAM", "1")
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", str(lanes))
    opal_opt.set("OPAL_HIP_NARROW", narrow)
    cnt = (C.c_longlong * 11)()
    for blocks in ("1", "3"):
        opal_opt.set("OPAL_HIP_STREAM_BLOCKS", blocks)
        for go, ge in ((0, 0), (1, 0), (0, 1), (2, 1)):
            ours.lib.opalx_counters(cnt, 11)
            before = cnt[0]
            got = ours.search(q, db, go, ge, m, 1, MODES["SW"])[:4]
            ours.lib.opalx_counters(cnt, 11)
            assert cnt[0] - before >= 1, "not on the streaming kernels"
            want = oracle.search(q, db, go, ge, m, 1, MODES["SW"])
            _check(got, want, "SW", f"stream ties G={lanes} Q={qlen} w{narrow} gaps {go}/{ge} blocks={blocks}")


@pytest.mark.parametrize("lanes,qlen", [(64, 700), (32, 513), (16, 300)])
def test_sw_end_cells_with_three_sequences_in_flight(lanes, qlen, ours, oracle, opal_opt):
    """Round 6: a sequence shorter than G - 2 behind a long one puts THREE sequences into the G lanes of a slot -- the long one, the
    short one and, in the short one's pad columns, already the lanes' next -- and the settlement of the streaming SW kernels
    grouped lanes by the parity of their sequence number: the lanes on the third took the first one's best cell for their own
    and never raised the (lower) best of their real sequence again: a wrong end cell among equal scores, a lower score otherwise.
    Rare in the suite (which slot a sequence lands in depends on timing; the ties test above failed once in 25 runs), certain here:
    one persistent workgroup, long sequences whose score grows to their last column, then only short ones whose first best
    cell sits in lane 0. Fails on the round-5 kernel."""
    A = 4
    m = np.full((A, A), -1, np.int32)
    np.fill_diagonal(m, 1)
    q = np.zeros(qlen, np.uint8)
    slots = 16 * (64 // lanes)
    db = [np.zeros(3 * lanes + 7 * (i % 5), np.uint8) for i in range(3 * slots)]            # long: the score grows to the last column
    db += [np.zeros(1 + (i % max(2, lanes - 3)), np.uint8) for i in range(40 * slots)]      # short: 1 .. G-3 residues, every best in lane 0
    db += [np.array([0, 1, 0, 0, 1, 0], np.uint8), np.array([1], np.uint8)]
    opal_opt.set("OPAL_HIP_STREAM", "1")
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", str(lanes))
    opal_opt.set("OPAL_HIP_NARROW", "17")
    cnt = (C.c_longlong * 11)()
    for blocks in ("1", "2"):
        opal_opt.set("OPAL_HIP_STREAM_BLOCKS", blocks)
        for go, ge in ((1, 1), (0, 0), (2, 1)):
            ours.lib.opalx_counters(cnt, 11)
            before = cnt[0]
            got = ours.search(q, db, go, ge, m, 1, MODES["SW"])[:4]
            ours.lib.opalx_counters(cnt, 11)
            assert cnt[0] - before >= 1, "not on the streaming kernels"
            want = oracle.search(q, db, go, ge, m, 1, MODES["SW"])
            _check(got, want, "SW", f"three in flight G={lanes} Q={qlen} gaps {go}/{ge} blocks={blocks}")


def test_api_rules(ours, oracle):
    q, db, m = cases.materialise(cases.CASES["readme_example"])
    # score-only search: end locations -1; alignment fields post-filled (opal.cpp:1508-1515)
    rc, s, eq, et, sset, (arr, _) = ours.search(q, db, 3, 1, m, 0, MODES["SW"])
    assert rc == 0 and s.tolist() == [14, 10, 16, 12] and (eq == -1).all() and (et == -1).all()
    assert all(arr[i].alignmentLength == -1 and not arr[i].alignment and arr[i].startLocationQuery == -1 for i in range(4))
    # invalid mode: 3, nothing written
    rc, s, eq, et, sset, _ = ours.search(q, db, 3, 1, m, 0, 7)
    assert rc == 3 and (sset == 0).all()
    # matrix entry outside (INT_MIN/2, INT_MAX/2): OPAL_ERR_OVERFLOW
    big = m.copy(); big[0, 0] = 2 ** 30
    rc, *_ = ours.search(q, db, 3, 1, big, 0, MODES["SW"])
    assert rc == 1
    # 32-bit wrap in SW is detected like the reference does (opal.cpp:302-303,339-351)
    mm = cases.simple_matrix(4, 2 ** 30 - 2, -1)
    rc, s, *_ = ours.search([1, 1], [[1, 1]], 3, 1, mm, 0, MODES["SW"])
    assert rc == 0 and s[0] == 2 ** 31 - 4
    rc, *_ = ours.search([1, 1, 1], [[1, 1, 1]], 3, 1, mm, 0, MODES["SW"])
    assert rc == 1
    # SW score 0 -> end (-1,-1)
    rc, s, eq, et, *_ = ours.search([0, 0, 0], [[1, 1, 1, 1]], 3, 1, cases.simple_matrix(4, 2, -1), 1, MODES["SW"])
    assert (rc, s[0], eq[0], et[0]) == (0, 0, -1, -1)


def test_result_reuse(ours):
    """A non-empty result (with end location when one is asked for) is kept, not recomputed (opal.h:118-122)."""
    q, db, m = cases.materialise(cases.CASES["td_O74807_sprot15_o3"])
    res = ours.new_results(len(db))
    arr, ptrs = res
    arr[3].scoreSet = 1; arr[3].score = 12345; arr[3].endLocationQuery = 7; arr[3].endLocationTarget = 9
    arr[5].scoreSet = 1; arr[5].score = 777  # no end location: must be recomputed for SCORE_END
    rc, s, eq, et, sset, _ = ours.search(q, db, 3, 1, m, 1, MODES["SW"], results=res)
    assert rc == 0 and s[3] == 12345 and (eq[3], et[3]) == (7, 9)
    assert s[5] == 68 and s[0] == 155
    rc, s, *_ = ours.search(q, db, 3, 1, m, 0, MODES["NW"], results=res)  # everything already has a score
    assert rc == 0 and s[0] == 155


def test_result_reuse_in_a_large_database(ours, oracle):
    """The same on a database large enough for the device ordering to run while the result structs are looked through (opal_api.cpp,
    overlapLook: one device, >= 16 384 sequences): when some results are already set the overlapped ordering of ALL sequences is
    dropped and the call goes on with what is left -- on a cached context whose previous database it has just overwritten. Kept
    results stay, everything else is the oracle's, and the next call (everything to compute, same database) is right too."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    m = blosum50()[1]
    n = 20000
    res_, offs, lens = synth.database(n, seed=42, first=700000)
    db = [res_[offs[i]:offs[i + 1]] for i in range(n)]
    q = synth.query(150, seed=9)
    rc, s0, eq0, et0, *_ = ours.search(q, db, 11, 1, m, 1, MODES["SW"])      # the context now holds this database
    assert rc == 0
    res = ours.new_results(n)
    arr, _ = res
    kept = list(range(0, n, 97))
    for i in kept:
        arr[i].scoreSet = 1; arr[i].score = 100000 + i; arr[i].endLocationQuery = 5; arr[i].endLocationTarget = 6
    arr[11].scoreSet = 1; arr[11].score = -5    # no end location: recomputed for SCORE_END
    rc, s, eq, et, sset, _ = ours.search(q, db, 11, 1, m, 1, MODES["SW"], results=res)
    assert rc == 0 and (sset == 1).all()
    mask = np.ones(n, bool); mask[kept] = False
    np.testing.assert_array_equal(s[kept], 100000 + np.array(kept))
    assert (eq[kept] == 5).all() and (et[kept] == 6).all()
    np.testing.assert_array_equal(s[mask], s0[mask])
    np.testing.assert_array_equal(eq[mask], eq0[mask])
    np.testing.assert_array_equal(et[mask], et0[mask])
    pick = np.arange(3, n, 400)
    want = oracle.search(q, [db[i] for i in pick], 11, 1, m, 1, MODES["SW"])
    np.testing.assert_array_equal(s[pick][mask[pick]], want[1][mask[pick]])
    # alignment search with some results set (score + end cells): only those are traced, the others computed first
    res2 = ours.new_results(n)
    arr2, _ = res2
    for i in kept[:50]:
        arr2[i].scoreSet = 1; arr2[i].score = int(s0[i]); arr2[i].endLocationQuery = int(eq0[i]); arr2[i].endLocationTarget = int(et0[i])
    rc, s2, eq2, et2, *_ = ours.search(q, db, 11, 1, m, 1, MODES["SW"], results=res2)
    assert rc == 0
    np.testing.assert_array_equal(s2, s0)
    np.testing.assert_array_equal(eq2, eq0)
    # ... and everything again from scratch on the same context
    rc, s3, eq3, et3, *_ = ours.search(q, db, 11, 1, m, 1, MODES["SW"])
    assert rc == 0
    np.testing.assert_array_equal(s3, s0)
    np.testing.assert_array_equal(et3, et0)


def test_empty_and_ragged(ours, oracle):
    """Zero-length targets, one-residue targets, an empty database; values defined in oracle/opal_oracle.c."""
    from opal_amd.score_matrix import blosum50
    m = blosum50()[1]
    q = np.arange(20, dtype=np.uint8)
    db = [np.zeros(0, np.uint8), np.array([3], np.uint8), np.arange(20, dtype=np.uint8), np.zeros(0, np.uint8),
          np.array([1, 2], np.uint8)]
    for mode in cases.ALL_MODES:
        rc, s, eq, et, *_ = ours.search(q, db, 3, 1, m, 1, MODES[mode])
        want = oracle.search(q, db, 3, 1, m, 1, MODES[mode])
        _check((rc, s, eq, et), want, mode, f"ragged {mode}")
    rc, s, *_ = ours.search(q, [], 3, 1, m, 0, MODES["SW"])
    assert rc == 0 and len(s) == 0


@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_width32_paths(mode, ours, oracle):
    """Inputs that cannot use packed 16-bit math: huge matrix entries / gap penalties, long targets."""
    from opal_amd import synth
    q = (synth.splitmix64(5, 0, 90) % np.uint64(4)).astype(np.uint8)
    _, db = cases.random4(77, 10, 40, 4, 50, 300)
    m = cases.simple_matrix(4, 40000, -30000)
    rc, s, eq, et, *_ = ours.search(q, db, 50000, 1000, m, 1, MODES[mode])
    want = oracle.search(q, db, 50000, 1000, m, 1, MODES[mode])
    _check((rc, s, eq, et), want, mode, f"w32 big-values {mode}")
    # long targets with a large extension penalty: the 16-bit a-priori bound fails, 32-bit takes over
    _, dbl = cases.random4(78, 10, 12, 4, 3000, 6000)
    m2 = cases.simple_matrix(4, 3, -2)
    rc, s, eq, et, *_ = ours.search(q, dbl + db[:7], 9, 7, m2, 1, MODES[mode])
    want = oracle.search(q, dbl + db[:7], 9, 7, m2, 1, MODES[mode])
    _check((rc, s, eq, et), want, mode, f"w32 long {mode}")


ALIGN_CASES = sorted(n for n, c in cases.CASES.items() if c.get("align"))


@pytest.mark.parametrize("name", ALIGN_CASES)
def test_alignment_search(name, ours, oracle, golden):
    """OPAL_SEARCH_ALIGNMENT end to end: GPU score/end + host traceback == the reference's alignments."""
    case = cases.CASES[name]
    q, db, m = cases.materialise(case)
    for mode in case["modes"]:
        rc, s, eq, et, sset, (arr, _) = ours.search(q, db, case["gap_open"], case["gap_ext"], m, 2, MODES[mode])
        assert rc == 0
        sq, st, ops = ours.alignments(arr)
        want = golden[name]["align"][mode]
        np.testing.assert_array_equal(s, want["score"])
        for i in range(len(db)):
            if mode == "SW" and s[i] == 0:
                assert (sq[i], st[i], eq[i], et[i], ops[i]) == (-1, -1, -1, -1, b"")
                continue
            assert (eq[i], et[i]) == (want["endQ"][i], want["endT"][i])
            assert (sq[i], st[i]) == (want["startQ"][i], want["startT"][i])
            assert "".join(str(b) for b in ops[i]) == want["ops"][i]
            assert oracle.check_alignment(q, db[i], case["gap_open"], case["gap_ext"], m, s[i], sq[i], st[i], eq[i], et[i],
                                          np.frombuffer(ops[i], np.uint8)) == 0


def _device_starts(q, db, go, ge, m, mode):
    """score+end search, then the reverse pass, both on the device -> host arrays + redo list."""
    import torch
    from opal_amd.api import PreparedDb
    pdb = PreparedDb.from_sequences(db)
    n = len(db)
    dev = torch.device("cuda", 0)
    t = [torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(5)]
    rc, _ = pdb.search_device(q, go, ge, m, 1, mode, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())
    assert rc == 0
    rc, redo = pdb.search_starts_device(q, go, ge, m, mode, *[x.data_ptr() for x in t])
    assert rc == 0
    return [x.cpu().numpy() for x in t], redo


@pytest.mark.parametrize("name", ALIGN_CASES)
def test_device_start_locations_vs_golden(name, golden):
    """n4: start cells computed ON THE GPU (rev_kernel.hip) equal the reference's recorded startQ/startT."""
    case = cases.CASES[name]
    q, db, m = cases.materialise(case)
    for mode in case["modes"]:
        (s, eq, et, sq, st), redo = _device_starts(q, db, case["gap_open"], case["gap_ext"], m, MODES[mode])
        want = golden[name]["align"][mode]
        np.testing.assert_array_equal(s, want["score"])
        live = np.ones(len(db), bool)
        live[redo] = False                    # sequences handed back to the host pass (OV coincidences; none expected here)
        assert len(redo) <= max(1, len(db) // 10)
        for i in np.nonzero(live)[0]:
            if mode == "SW" and s[i] == 0:
                assert (sq[i], st[i]) == (-1, -1)
                continue
            assert (sq[i], st[i]) == (want["startQ"][i], want["startT"][i]), f"{name} {mode} #{i}: start"


@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_device_start_locations_differential(mode, ours, opal_opt):
    """n4 on a larger set: the alignment search with device start cells + box traceback gives the same start cells and
    operation strings as the same search with the host's full reverse pass (OPAL_HIP_HOST_STARTS=1)."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(700, seed=42, first=40000)
    db = [res[offs[i]:offs[i + 1]] for i in range(700)]
    q = synth.query(333, seed=9)
    m = blosum50()[1]
    rc, s, eq, et, sset, (arr, _) = ours.search(q, db, 3, 1, m, 2, MODES[mode])
    assert rc == 0
    sq, st, ops = ours.alignments(arr)
    opal_opt.set("OPAL_HIP_HOST_STARTS", "1")
    rc2, s2, eq2, et2, sset2, (arr2, _) = ours.search(q, db, 3, 1, m, 2, MODES[mode])
    assert rc2 == 0
    sq2, st2, ops2 = ours.alignments(arr2)
    np.testing.assert_array_equal(s, s2)
    np.testing.assert_array_equal(sq, sq2)
    np.testing.assert_array_equal(st, st2)
    assert ops == ops2
    # and straight from the kernel: few or no sequences handed back to the host
    (ds, deq, det, dsq, dst), redo = _device_starts(q, db, 3, 1, m, MODES[mode])
    keep = np.ones(len(db), bool); keep[redo] = False
    np.testing.assert_array_equal(dsq[keep], sq[keep])
    np.testing.assert_array_equal(dst[keep], st[keep])
    assert len(redo) == 0, f"{len(redo)} of 700 sequences needed the host pass"  # (doubtful rows: second pass on the device)


REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")


CLI = os.path.join(os.path.dirname(LIB), "opal_hip_aligner")


def _cli_lines(binary, args):
    import subprocess
    out = subprocess.run([binary] + args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
    skip = ("Cpu time of searching", "Wall time of searching", "GCUPS")
    return [ln for ln in out.stdout.splitlines() if not ln.startswith(skip)]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "ref_opal_aligner")), reason="reference CLI not built")
@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
@pytest.mark.parametrize("level", [0, 1, 2])
def test_own_cli_output_equals_reference_cli(mode, level):
    """n3: our own command-line front end (opal_cli.cpp, GPU library) prints the same lines as the reference's
    opal_aligner (its own objects, CPU) on the bundled test data: scores, start/end cells, alignment blocks."""
    assert os.path.exists(CLI), "opal_hip_aligner not built"
    for query, db in (("query_O74807.fasta", "uniprot_sprot15.fasta"), ("query_P18080.fasta", "uniprot_sprot196.fasta")):
        if level == 2 and db.endswith("196.fasta") and mode != "SW":
            continue  # keep the CPU reference's alignment runs short
        args = ["-a", mode, "-x", str(level), os.path.join(DATA, query), os.path.join(DATA, db)]
        ours_lines = _cli_lines(CLI, args)
        ref_lines = _cli_lines(os.path.join(REF_DIR, "ref_opal_aligner"), args)
        if mode == "SW" and level == 1:
            # SW score 0 with end location: the reference prints an undefined end cell; none in this data
            pass
        assert ours_lines == ref_lines


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "dropin_test")), reason="drop-in binaries not built")
@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_dropin_reference_selftest(mode):
    """The reference's own randomized self-test (src/test.cpp), compiled UNCHANGED and linked against our
    libopal.so: it prints one '#i: ...' line per mismatch against its scalar ground truth; none expected."""
    import subprocess
    out = subprocess.run([os.path.join(REF_DIR, "dropin_test"), mode], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    bad = [ln for ln in out.stdout.splitlines() if ln.startswith("#") or "Error" in ln or "Wrong" in ln or "Should be" in ln
           or "went outside" in ln or "ended at" in ln]
    assert not bad, "\n".join(bad[:10])
    assert "Maximum:" in out.stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "dropin_opal_aligner")), reason="drop-in binaries not built")
@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
@pytest.mark.parametrize("level", ["0", "1", "2"])
def test_dropin_cli_output_equals_reference_cli(mode, level):
    """The reference CLI (src/opal_aligner.cpp) linked against our library prints the same result lines as the
    reference CLI linked against the reference library (timing lines aside)."""
    import subprocess
    args = ["-a", mode, "-x", level, os.path.join(DATA, "query_P19930.fasta"), os.path.join(DATA, "uniprot_sprot196.fasta")]
    a = subprocess.run([os.path.join(REF_DIR, "dropin_opal_aligner")] + args, capture_output=True, text=True, timeout=600)
    b = subprocess.run([os.path.join(REF_DIR, "ref_opal_aligner")] + args, capture_output=True, text=True, timeout=600)
    assert a.returncode == b.returncode == 0, a.stderr[-2000:]
    strip = lambda txt: [ln for ln in txt.splitlines() if not ln.startswith(("Cpu time", "GCUPS", "Read ", "Reading"))]
    la, lb = strip(a.stdout), strip(b.stdout)
    if mode == "SW" and level == "1":
        # score-0 rows have undefined end locations in the reference; none occur for this pair
        pass
    assert la == lb


@pytest.mark.parametrize("geometry,narrow", [("2,8", None), ("8,8", None), ("1,16", None), ("64,8", None), ("4,40", None),
                                             ("2,8", "18"), ("16,8", "18"), ("1,24", "18"), ("4,32", "18"), ("2,8", "16"), ("4,40", "16")])
@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_query_chunks_forced(geometry, narrow, mode, ours, oracle, opal_opt):
    """Long-query path on short inputs: force the query to be cut into many row chunks (boundary row through
    HBM) and compare with the oracle, 16-bit (plain with 4- and 2-byte profile entries, packed pairs) and 32-bit
    kernels, score and end locations."""
    if narrow:
        opal_opt.set("OPAL_HIP_NARROW", narrow)
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(90, seed=321)
    db = [res[offs[i]:offs[i + 1]] for i in range(90)]
    q = synth.query(203, seed=11)
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_FORCE_CHUNKS", "1")
    opal_opt.set("OPAL_HIP_CHUNK_GEOMETRY", geometry)
    got = ours.search(q, db, 3, 1, m, 1, MODES[mode])[:4]
    _check(got, oracle.search(q, db, 3, 1, m, 1, MODES[mode]), mode, f"chunks {geometry} {mode}")
    # 32-bit kernels: scores beyond int16
    m32 = m * 300
    got = ours.search(q, db[:30], 900, 300, m32, 1, MODES[mode])[:4]
    _check(got, oracle.search(q, db[:30], 900, 300, m32, 1, MODES[mode]), mode, f"chunks32 {geometry} {mode}")


@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_query_longer_than_one_wave(mode, ours, oracle):
    """3000-residue query: more rows than 64 lanes x 40 registers, so it runs as two chunks."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(40, seed=77)
    db = [res[offs[i]:offs[i + 1]] for i in range(40)]
    q = synth.query(3000, seed=5)
    db.append(q[100:2900].copy())  # a long near-identical target: high scores across the chunk boundary
    m = blosum50()[1]
    got = ours.search(q, db, 3, 1, m, 1, MODES[mode])[:4]
    _check(got, oracle.search(q, db, 3, 1, m, 1, MODES[mode]), mode, f"Q=3000 {mode}")


def test_large_alphabet_profile_exceeds_lds(ours, oracle):
    """200-letter alphabet: the query profile of a 700-residue query does not fit LDS in one piece."""
    from opal_amd import synth
    A = 200
    rnd = lambda seed, n: (synth.splitmix64(seed, 0, n) % np.uint64(A)).astype(np.uint8)
    q = rnd(1, 700)
    db = [rnd(10 + i, 50 + 37 * i) for i in range(25)] + [q[50:650].copy()]
    m = ((synth.splitmix64(99, 0, A * A) % np.uint64(9)).astype(np.int32) - 5).reshape(A, A)
    np.fill_diagonal(m, 6)
    for mode in ("SW", "HW"):
        got = ours.search(q, db, 4, 1, m, 1, MODES[mode])[:4]
        _check(got, oracle.search(q, db, 4, 1, m, 1, MODES[mode]), mode, f"A=200 {mode}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("OPAL_TEST_SEEDS", "12"))))
def test_randomized_differential(seed, ours, oracle):
    """Seeded random shapes: alphabet size, matrix, gap penalties, query/target lengths, mode, search type;
    each drawn configuration is compared with the oracle (bit-exact)."""
    rng = np.random.RandomState(1000 + seed)
    for _ in range(6):
        A = int(rng.choice([2, 4, 5, 20, 24, 31, 64]))
        m = rng.randint(-9, 12, size=(A, A)).astype(np.int32)
        if rng.rand() < 0.5:
            m = ((m + m.T) // 2).astype(np.int32)
            np.fill_diagonal(m, rng.randint(1, 14))
        go = int(rng.randint(0, 16)); ge = int(rng.randint(0, go + 1)) if rng.rand() < 0.8 else int(rng.randint(0, 20))
        qlen = int(rng.choice([1, 2, 7, 39, 40, 41, 80, 81, 159, 320, 333, 641, 900]))
        n = int(rng.randint(1, 70))
        lens = np.maximum(1, (rng.lognormal(4.5, 1.0, size=n)).astype(int)); lens = np.minimum(lens, 1500)
        if rng.rand() < 0.3:
            lens[rng.randint(0, n)] = int(rng.randint(1500, 4000))
        q = rng.randint(0, A, size=qlen).astype(np.uint8)
        db = [rng.randint(0, A, size=int(L)).astype(np.uint8) for L in lens]
        if rng.rand() < 0.5:  # plant related sequences so scores are not all tiny
            for i in range(0, n, 3):
                L = len(db[i]); k = min(L, qlen)
                db[i][:k] = np.where(rng.rand(k) < 0.8, q[:k], db[i][:k])
        mode = str(rng.choice(["SW", "NW", "HW", "OV"])); st = int(rng.randint(0, 2))
        got = ours.search(q, db, go, ge, m, st, MODES[mode], int(rng.randint(0, 2)))[:4]
        want = oracle.search(q, db, go, ge, m, st, MODES[mode])
        _check(got, want, mode, f"seed {seed}: A={A} Q={qlen} n={n} gaps {go}/{ge} {mode} st={st}")
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all()  # including the defined SW score-0 ends


def test_reentrant_from_host_threads(ours, oracle):
    """The reference entry point is re-entrant (SURVEY.md 8b "Threading"); so is ours: four host threads search
    four different databases at once and every result matches the oracle."""
    import threading
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    m = blosum50()[1]
    jobs = []
    for t in range(4):
        res, offs, lens = synth.database(150 + 40 * t, seed=500 + t)
        db = [res[offs[i]:offs[i + 1]] for i in range(len(lens))]
        q = synth.query(100 + 57 * t, seed=t)
        jobs.append((q, db, ["SW", "NW", "HW", "OV"][t]))
    out = [None] * 4

    def run(i):
        q, db, mode = jobs[i]
        for _ in range(3):
            out[i] = ours.search(q, db, 3, 1, m, 1, MODES[mode])[:4]

    threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for th in threads: th.start()
    for th in threads: th.join()
    for i, (q, db, mode) in enumerate(jobs):
        _check(out[i], oracle.search(q, db, 3, 1, m, 1, MODES[mode]), mode, f"thread {i} {mode}")


@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
@pytest.mark.parametrize("qlen", [513, 1100, 2000])
def test_width18_two_byte_profile_kernels(mode, qlen, ours, oracle, opal_opt):
    """The plain-16-bit kernels with 2-byte profile entries (width 18: what 2000-aa queries run on), forced also for
    shorter queries: every rows-per-lane instantiation family (8/16/24/32/40), score + end location and score only."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    opal_opt.set("OPAL_HIP_NARROW", "18")
    res, offs, lens = synth.database(260, seed=42, first=70000 + qlen)
    db = [res[offs[i]:offs[i + 1]] for i in range(260)] + [synth.query(qlen, seed=7)[: qlen - 3].copy()]
    q = synth.query(qlen, seed=7)
    m = blosum50()[1]
    for lanes in (("", "64") if qlen > 1280 else ("", "32", "64")):
        if lanes:
            opal_opt.set("OPAL_HIP_LANES_PER_SEQ", lanes)
        got = ours.search(q, db, 3, 1, m, 1, MODES[mode])[:4]
        _check(got, oracle.search(q, db, 3, 1, m, 1, MODES[mode]), mode, f"width 18 q{qlen} {mode} lanes={lanes or 'auto'}")
    if mode == "SW":
        got0 = ours.search(q, db, 11, 1, m, 0, MODES[mode])[:4]
        _check(got0, oracle.search(q, db, 11, 1, m, 0, MODES[mode]), mode, f"width 18 q{qlen} score only")


@pytest.mark.parametrize("narrow", ["16", "17", "18"])
def test_many_saturated_scores_rerun(narrow, ours, oracle, opal_opt):
    """Most of the database exceeds the 16-bit rung: the ballot-built re-run list must hold every one of them
    exactly once (scores and end locations come from the 32-bit kernels), the rest stays at 16 bit."""
    opal_opt.set("OPAL_HIP_NARROW", narrow)
    rng = np.random.RandomState(5)
    q = rng.randint(0, 4, size=900).astype(np.uint8)
    m = cases.simple_matrix(4, 60, -40)
    db = []
    for i in range(150):
        if i % 3 == 0:
            db.append(rng.randint(0, 4, size=int(rng.randint(20, 900))).astype(np.uint8))  # unrelated: small score
        else:
            t = q[: int(rng.randint(600, 900))].copy()                                       # near copy: > 32767
            t[rng.randint(0, len(t), size=5)] = rng.randint(0, 4, size=5)
            db.append(t)
    got = ours.search(q, db, 30, 5, m, 1, MODES["SW"])[:4]
    want = oracle.search(q, db, 30, 5, m, 1, MODES["SW"])
    assert (want[1] > 32767).sum() > 80 and (want[1] < 20000).sum() > 30
    _check(got, want, "SW", f"many overflows narrow={narrow}")


@pytest.mark.parametrize("mode", ["NW", "HW", "OV"])
def test_unskewed_16bit_kernels_still_match(mode, ours, oracle, opal_opt):
    """NW/HW/OV normally run in the diagonal-skewed representation; the plain one (used when the skew would
    leave the 16-bit range) must give the same answers."""
    q, db, m = cases.materialise(cases.CASES["synth_q300_db2000_o3"])
    db = db[:400]
    opal_opt.set("OPAL_HIP_NO_SKEW", "1")
    got = ours.search(q, db, 3, 1, m, 1, MODES[mode])[:4]
    _check(got, oracle.search(q, db, 3, 1, m, 1, MODES[mode]), mode, f"no-skew {mode}")
    opal_opt.unset("OPAL_HIP_NO_SKEW")
    # large extension penalty: the skew bound fails for the long targets, holds for the short ones
    _, dbl = cases.random4(91, 10, 30, 4, 30, 2500)
    q4 = (np.arange(150) % 4).astype(np.uint8)
    m4 = cases.simple_matrix(4, 5, -4)
    got = ours.search(q4, dbl, 12, 9, m4, 1, MODES[mode])[:4]
    _check(got, oracle.search(q4, dbl, 12, 9, m4, 1, MODES[mode]), mode, f"mixed skew {mode}")


def _stream_db(qlen, n=500):
    """Synthetic sequences plus what a stream of sequences has to get right: single residues, sequences shorter than
    the lane pipeline, a 5000-residue one, near copies of the query, an odd count (a pair with an absent half)."""
    from opal_amd import synth
    res, offs, lens = synth.database(n, seed=42, first=31000 + qlen)
    q = synth.query(qlen, seed=7)
    db = [res[offs[i]:offs[i + 1]] for i in range(n)]
    rng = np.random.RandomState(qlen)
    for L in (1, 1, 2, 3, 5, 17, 33, 63, 64, 65, 127, 128, 129):
        db.append(rng.randint(0, 20, size=L).astype(np.uint8))
    db.append(synth.database(1, seed=9, first=5)[0][:30])
    db.append(rng.randint(0, 20, size=5000).astype(np.uint8))
    db.append(q.copy())
    db.append(q[: max(1, qlen - 7)].copy())
    db.append(np.concatenate([q[qlen // 3:], q[: qlen // 5]]))
    if len(db) % 2 == 0:
        db.append(q[: max(1, qlen // 2)].copy())
    order = rng.permutation(len(db))
    return q, [db[i] for i in order]


@pytest.mark.parametrize("qlen,lanes,narrow", [(100, "4", "17"), (300, "8", "17"), (513, "16", "17"), (1000, "32", "17"), (1280, "32", "17"),
                                               (1330, "64", "17"), (1100, "64", "18"), (2000, "64", "18"), (2500, "64", "18")])
def test_streaming_sw_score_kernels(qlen, lanes, narrow, ours, oracle, opal_opt):
    """The streaming SW score kernels (persistent workgroups, sequences handed out by a device counter and separated by
    two separator columns), forced for every lanes-per-sequence value and both profile widths; once more with a single
    workgroup, so that every slot works through many sequences."""
    from opal_amd.score_matrix import blosum50
    q, db = _stream_db(qlen)
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_STREAM", "1")
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", lanes)
    opal_opt.set("OPAL_HIP_NARROW", narrow)
    for blocks in ("", "1"):
        if blocks:
            opal_opt.set("OPAL_HIP_STREAM_BLOCKS", blocks)
        for go, ge in ((11, 1), (3, 1)):
            got = ours.search(q, db, go, ge, m, 0, MODES["SW"])[:4]
            want = oracle.search(q, db, go, ge, m, 0, MODES["SW"])
            assert got[0] == want[0] == 0
            np.testing.assert_array_equal(got[1], want[1], err_msg=f"stream q{qlen} G={lanes} w{narrow} gaps {go}/{ge} blocks={blocks or 'auto'}")
            # with end locations (the streaming kernels exist up to 36 rows per lane; beyond that the class kernels answer)
            got = ours.search(q, db, go, ge, m, 1, MODES["SW"])[:4]
            _check(got, oracle.search(q, db, go, ge, m, 1, MODES["SW"]), "SW", f"stream+end q{qlen} G={lanes} w{narrow} gaps {go}/{ge} blocks={blocks or 'auto'}")


@pytest.mark.parametrize("qlen,geometry", [(300, "16,16"), (700, "16,16"), (1000, "32,16"), (1290, "16,40"), (2100, "64,16"), (2000, "32,23"),
                                           (3000, None), (5200, None)])
def test_streaming_sw_row_chunks(qlen, geometry, ours, oracle, opal_opt):
    """Queries beyond one wave stream through one launch per row chunk, the chunk's bottom row (H and F, one word per column)
    going through HBM: forced on shorter queries with small chunks (2 to 5 passes, a partial first chunk), and the natural
    case; once more with a single workgroup, so that every slot carries many sequences across every chunk boundary."""
    from opal_amd.score_matrix import blosum50
    q, db = _stream_db(qlen, n=300 if qlen > 2500 else 500)
    m = blosum50()[1]
    if geometry:
        opal_opt.set("OPAL_HIP_STREAM_CHUNKS", geometry)
    else:
        opal_opt.set("OPAL_HIP_STREAM", "1")
    cnt = (C.c_longlong * 9)()
    for blocks in ("", "1"):
        if blocks:
            opal_opt.set("OPAL_HIP_STREAM_BLOCKS", blocks)
        for go, ge in ((11, 1), (3, 1)):
            ours.lib.opalx_counters(cnt, 9)
            before = (cnt[0], cnt[4])
            got = ours.search(q, db, go, ge, m, 0, MODES["SW"])[:4]
            ours.lib.opalx_counters(cnt, 9)
            assert cnt[0] - before[0] == 1 and cnt[4] - before[1] >= 2, "not on the chunked streaming path"
            want = oracle.search(q, db, go, ge, m, 0, MODES["SW"])
            assert got[0] == want[0] == 0
            np.testing.assert_array_equal(got[1], want[1], err_msg=f"stream chunks q{qlen} {geometry} gaps {go}/{ge} blocks={blocks or 'auto'}")
            # with end locations: the best cell so far travels from chunk to chunk with the score (kernels up to 32 rows per lane)
            got = ours.search(q, db, go, ge, m, 1, MODES["SW"])[:4]
            _check(got, oracle.search(q, db, go, ge, m, 1, MODES["SW"]), "SW", f"stream chunks + end q{qlen} {geometry} gaps {go}/{ge} blocks={blocks or 'auto'}")


def test_streaming_row_chunks_saturated_scores(ours, oracle, opal_opt):
    """Scores that reach the 16-bit limit in the first, a middle or the last chunk are listed for the 32-bit re-run exactly
    once and come back exact; the others keep the maximum over the chunks."""
    opal_opt.set("OPAL_HIP_STREAM_CHUNKS", "16,20")
    rng = np.random.RandomState(15)
    q = rng.randint(0, 4, size=1400).astype(np.uint8)
    m = cases.simple_matrix(4, 60, -40)
    db = []
    for i in range(181):
        if i % 3 == 0:
            db.append(rng.randint(0, 4, size=int(rng.randint(20, 900))).astype(np.uint8))
        else:
            a = int(rng.randint(0, 1000))
            t = q[a: a + int(rng.randint(100, 1400))].copy()   # a piece of the query: high scores in the chunks that hold it
            t[rng.randint(0, len(t), size=3)] = rng.randint(0, 4, size=3)
            db.append(t)
    got = ours.search(q, db, 30, 5, m, 0, MODES["SW"])[:4]
    want = oracle.search(q, db, 30, 5, m, 0, MODES["SW"])
    assert (want[1] > 32767).sum() > 40 and (want[1] < 20000).sum() > 40
    assert got[0] == 0
    np.testing.assert_array_equal(got[1], want[1])
    got = ours.search(q, db, 30, 5, m, 1, MODES["SW"])[:4]
    _check(got, oracle.search(q, db, 30, 5, m, 1, MODES["SW"]), "SW", "stream chunks + end, overflows")


@pytest.mark.parametrize("mode", ["NW", "HW", "OV"])
@pytest.mark.parametrize("qlen,geometry", [(1000, "16,24"), (1700, "32,24"), (2000, "16,37"), (3000, None), (4100, None)])
def test_streaming_global_row_chunks(mode, qlen, geometry, ours, oracle, opal_opt):
    """NW/HW/OV queries beyond one wave stream their row chunks too: the boundary row goes through HBM as true values and
    changes skewed frame on the way (NW: the last chunk is the partial one, HW/OV: the first; OV's last-column maximum
    travels from chunk to chunk). The sequences that need 32 bits or cannot be skewed (a 5000-residue one, one beyond
    31 000) take the class kernels' chunks beside the stream. Forced on shorter queries, natural at 3000 and 4100 aa;
    once more with a single workgroup."""
    from opal_amd.score_matrix import blosum50
    q, db = _stream_db(qlen, n=300)
    m = blosum50()[1]
    if geometry:
        opal_opt.set("OPAL_HIP_STREAM_CHUNKS", geometry)
    opal_opt.set("OPAL_HIP_STREAM", "1")  # (a small database: its few long sequences hold more than the share the scheduler allows)
    cnt = (C.c_longlong * 9)()
    for blocks in ("", "1"):
        if blocks:
            opal_opt.set("OPAL_HIP_STREAM_BLOCKS", blocks)
        for go, ge in ((11, 1), (3, 1)):
            ours.lib.opalx_counters(cnt, 9)
            before = (cnt[1], cnt[4])
            got = ours.search(q, db, go, ge, m, 1, MODES[mode])[:4]
            ours.lib.opalx_counters(cnt, 9)
            assert cnt[1] - before[0] == 1 and cnt[4] - before[1] >= 2, "not on the chunked streaming path"
            _check(got, oracle.search(q, db, go, ge, m, 1, MODES[mode]), mode, f"global stream chunks q{qlen} {geometry} {mode} gaps {go}/{ge} blocks={blocks or 'auto'}")
        got = ours.search(q, db, 11, 1, m, 0, MODES[mode])[:4]
        want = oracle.search(q, db, 11, 1, m, 0, MODES[mode])
        np.testing.assert_array_equal(got[1], want[1])


@pytest.mark.parametrize("qlen,lanes,head", [(100, "4", "50"), (300, "8", "64"), (513, "16", "37"), (560, "16", "128")])
def test_streaming_with_a_head_on_the_class_body(qlen, lanes, head, ours, oracle, opal_opt):
    """A long-tailed database: the longest sequences run on the class body (4 x the lanes) as the first workgroups of the
    streaming launch, the rest streams behind them -- forced here with a head of a given size (the 36 000-residue, the
    5000-residue sequence and the near copies of the query are in it), score and end locations, 16-bit overflows included."""
    from opal_amd.score_matrix import blosum50
    q, db = _stream_db(qlen)
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_STREAM", "1")
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", lanes)
    opal_opt.set("OPAL_HIP_NARROW", "17")
    opal_opt.set("OPAL_HIP_STREAM_HEAD_ITEMS", head)
    for go, ge in ((11, 1), (3, 1)):
        for st in (0, 1):
            got = ours.search(q, db, go, ge, m, st, MODES["SW"])[:4]
            want = oracle.search(q, db, go, ge, m, st, MODES["SW"])
            if st == 0:
                assert got[0] == want[0] == 0
                np.testing.assert_array_equal(got[1], want[1], err_msg=f"stream head q{qlen} G={lanes} head={head} gaps {go}/{ge}")
            else:
                _check(got, want, "SW", f"stream head + end q{qlen} G={lanes} head={head} gaps {go}/{ge}")
    # scores beyond 16 bits in the head and behind it: both parts append to the same re-run list
    m60 = blosum50()[1] * 60
    got = ours.search(q, db, 660, 60, m60, 0, MODES["SW"])[:4]
    want = oracle.search(q, db, 660, 60, m60, 0, MODES["SW"])
    assert got[0] == want[0] == 0 and (want[1] > 32767).sum() >= 2
    np.testing.assert_array_equal(got[1], want[1])


def test_streaming_is_what_long_queries_run_on(ours):
    """Without any forcing, a long query on a database large enough for the longest sequence not to dominate takes the
    streaming path (with or without end locations); a short query and a small database do not."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(30000, seed=42)
    db = [res[offs[i]:offs[i + 1]] for i in range(30000)]
    m = blosum50()[1]
    count = ours.lib.opalx_stream_launch_count
    count.restype = C.c_longlong
    for qlen, st, n, expect in ((2000, 0, 30000, 1), (300, 0, 30000, 0), (2000, 1, 30000, 1), (2000, 0, 300, 0)):
        before = count()
        rc = ours.search(synth.query(qlen, seed=7), db[:n], 11, 1, m, st, MODES["SW"])[0]
        assert rc == 0
        assert count() - before == expect, (qlen, st, n)


@pytest.mark.parametrize("narrow", ["17", "18"])
def test_streaming_saturated_scores_rerun(narrow, ours, oracle, opal_opt):
    """Scores at the 16-bit limit leave the streaming kernel through the re-run list (one atomic per sequence, written by
    the last lane of the slot) and come back from the 32-bit kernels; the others stay."""
    opal_opt.set("OPAL_HIP_STREAM", "1")
    opal_opt.set("OPAL_HIP_NARROW", narrow)
    rng = np.random.RandomState(5)
    q = rng.randint(0, 4, size=1400).astype(np.uint8)
    m = cases.simple_matrix(4, 60, -40)
    db = []
    for i in range(151):
        if i % 3 == 0:
            db.append(rng.randint(0, 4, size=int(rng.randint(20, 900))).astype(np.uint8))
        else:
            t = q[: int(rng.randint(600, 1400))].copy()
            t[rng.randint(0, len(t), size=5)] = rng.randint(0, 4, size=5)
            db.append(t)
    got = ours.search(q, db, 30, 5, m, 0, MODES["SW"])[:4]
    want = oracle.search(q, db, 30, 5, m, 0, MODES["SW"])
    assert (want[1] > 32767).sum() > 80 and (want[1] < 20000).sum() > 20
    assert got[0] == 0
    np.testing.assert_array_equal(got[1], want[1])
    got = ours.search(q, db, 30, 5, m, 1, MODES["SW"])[:4]
    _check(got, oracle.search(q, db, 30, 5, m, 1, MODES["SW"]), "SW", f"stream+end overflows narrow={narrow}")


@pytest.mark.parametrize("mode", ["NW", "HW", "OV"])
@pytest.mark.parametrize("qlen,lanes,narrow", [(100, "4", "17"), (290, "8", "17"), (513, "16", "17"), (1000, "32", "17"), (1200, "64", "17"),
                                               (1330, "64", "18"), (2000, "64", "18"), (2360, "64", "18")])
def test_streaming_global_kernels(mode, qlen, lanes, narrow, ours, oracle, opal_opt):
    """NW/HW/OV as a stream of sequences (boundary column reloaded from LDS at every sequence start, diagonal-skewed
    arithmetic with a local step per sequence), forced for every lanes-per-sequence value and both profile widths, with
    end locations and score only; once more with a single workgroup."""
    from opal_amd.score_matrix import blosum50
    q, db = _stream_db(qlen)
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_STREAM", "1")
    opal_opt.set("OPAL_HIP_LANES_PER_SEQ", lanes)
    opal_opt.set("OPAL_HIP_NARROW", narrow)
    count = ours.lib.opalx_stream_launch_count
    count.restype = C.c_longlong
    before = count()
    for blocks in ("", "1"):
        if blocks:
            opal_opt.set("OPAL_HIP_STREAM_BLOCKS", blocks)
        # (gaps 12/4: the 5000-residue sequence no longer fits the skewed arithmetic and runs on the unskewed class body inside
        # the streaming launch, for the longer queries also at the smaller extension penalties)
        for go, ge, st in ((11, 1, 1), (3, 1, 1), (3, 1, 0), (12, 4, 1)):
            got = ours.search(q, db, go, ge, m, st, MODES[mode])[:4]
            _check(got, oracle.search(q, db, go, ge, m, st, MODES[mode]), mode, f"stream {mode} q{qlen} G={lanes} w{narrow} gaps {go}/{ge} st={st} blocks={blocks or 'auto'}")
    assert count() - before == 8


@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
@pytest.mark.parametrize("qlen,go,ge", [(60, 3, 1), (333, 11, 1), (333, 1, 1), (1000, 3, 1), (1700, 5, 2), (200, 0, 0), (200, 5, 0), (200, 1, 3)])
def test_device_traceback_equals_host_traceback(mode, qlen, go, ge, ours, oracle, opal_opt):
    """The operations traced on the device (trace_kernel.hip: box DP with one flag byte per cell in HBM, walk kernel) equal
    the host traceback's, for every rows-per-lane shape of the kernel, cheap gaps (long indels), a flag budget so small
    that the work is cut into many batches and the largest boxes are left to the host; every alignment re-scores."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    n = 260 if qlen < 1000 else 90
    res, offs, lens = synth.database(n, seed=42, first=43000 + qlen)
    q = synth.query(qlen, seed=9)
    db = [res[offs[i]:offs[i + 1]] for i in range(n)] + [q[5:qlen - 3].copy(), np.concatenate([q[: qlen // 2], q[qlen // 2 + 7:]])]
    m = blosum50()[1]
    opal_opt.set("OPAL_HIP_HOST_TRACEBACK", "1")
    rc, s, eq, et, sset, (arr, _) = ours.search(q, db, go, ge, m, 2, MODES[mode])
    assert rc == 0
    sq, st, ops = ours.alignments(arr)
    opal_opt.unset("OPAL_HIP_HOST_TRACEBACK")
    for budget in ("", "300000"):
        if budget:
            opal_opt.set("OPAL_TRACEBACK_DEVICE_BYTES", budget)
        rc2, s2, eq2, et2, sset2, (arr2, _) = ours.search(q, db, go, ge, m, 2, MODES[mode])
        assert rc2 == 0
        sq2, st2, ops2 = ours.alignments(arr2)
        np.testing.assert_array_equal(s, s2)
        np.testing.assert_array_equal(sq, sq2)
        np.testing.assert_array_equal(st, st2)
        assert ops == ops2, f"{mode} q{qlen} gaps {go}/{ge} budget={budget or 'default'}"
    if go < ge:
        # An opening cheaper than an extension: the recurrence (reference opal.cpp:280-327) lets a gap be "re-opened" column
        # after column for gapOpen each, so the optimal SCORE is below what any operation string re-scores to under the
        # run-length rule (first gap position gapOpen, the others gapExt; the reference's own checkAlignment, test.cpp, prices
        # runs the same way). Scores and end cells are the recurrence's (golden cases gaps_o1_e3); here device == host only.
        return
    for i in range(0, len(db), 7):
        if mode == "SW" and s[i] == 0:
            continue
        o = np.frombuffer(ops[i], dtype=np.uint8)
        assert oracle.check_alignment(q, db[i], go, ge, m, int(s[i]), int(sq[i]), int(st[i]), int(eq[i]), int(et[i]), o) == 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("OPAL_TEST_SEEDS", "8"))))  # (OPAL_TEST_SEEDS=600: a one-off stress run)
def test_randomized_differential_streaming(seed, ours, oracle, opal_opt):
    """The randomized shapes once more with the streaming kernels forced wherever a geometry exists (other alphabets and
    matrices than BLOSUM50, gap extension 0 or larger than the opening, queries from 1 to 2400 residues, every mode and
    search type, alignment searches included)."""
    opal_opt.set("OPAL_HIP_STREAM", "1")
    rng = np.random.RandomState(7000 + seed)
    for _ in range(5):
        A = int(rng.choice([2, 4, 5, 20, 24, 31]))
        m = rng.randint(-9, 12, size=(A, A)).astype(np.int32)
        if rng.rand() < 0.5:
            m = ((m + m.T) // 2).astype(np.int32)
            np.fill_diagonal(m, rng.randint(1, 14))
        go = int(rng.randint(0, 16)); ge = int(rng.randint(0, go + 1)) if rng.rand() < 0.8 else int(rng.randint(0, 20))
        qlen = int(rng.choice([1, 2, 7, 40, 41, 159, 333, 641, 900, 1281, 1500, 2047, 2400]))
        n = int(rng.randint(1, 120))
        lens = np.maximum(1, (rng.lognormal(4.5, 1.0, size=n)).astype(int)); lens = np.minimum(lens, 1500)
        if rng.rand() < 0.3:
            lens[rng.randint(0, n)] = int(rng.randint(1500, 4000))
        q = rng.randint(0, A, size=qlen).astype(np.uint8)
        db = [rng.randint(0, A, size=int(L)).astype(np.uint8) for L in lens]
        if rng.rand() < 0.5:
            for i in range(0, n, 3):
                L = len(db[i]); k = min(L, qlen)
                db[i][:k] = np.where(rng.rand(k) < 0.8, q[:k], db[i][:k])
        mode = str(rng.choice(["SW", "NW", "HW", "OV"])); st = int(rng.randint(0, 2))
        lanes = str(rng.choice(["", "", "16", "32", "64"]))
        if lanes:
            opal_opt.set("OPAL_HIP_LANES_PER_SEQ", lanes)
        else:
            opal_opt.unset("OPAL_HIP_LANES_PER_SEQ")
        got = ours.search(q, db, go, ge, m, st, MODES[mode], int(rng.randint(0, 2)))[:4]
        want = oracle.search(q, db, go, ge, m, st, MODES[mode])
        what = f"stream seed {seed}: A={A} Q={qlen} n={n} gaps {go}/{ge} {mode} st={st} lanes={lanes or 'auto'}"
        _check(got, want, mode, what)
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all(), what


@pytest.mark.parametrize("seed", range(int(os.environ.get("OPAL_TEST_SEEDS", "10"))))  # (OPAL_TEST_SEEDS=400: a one-off stress run)
def test_randomized_differential_round4_paths(seed, ours, oracle, opal_opt):
    """Randomized shapes aimed at what round 4 changed in the class kernels and at the role-swapped kernel: matrices scaled
    so that scores and the skew frame of the global modes (row position + column, times the gap extension) come close to the
    16-bit limit -- the host's safety bound decides kernel by kernel between the skewed 16-bit, the plain 16-bit and the
    32-bit path, and a wrong decision is a silently wrong score --, gap extensions up to 25, sequences shorter than the lane
    pipeline (the fill steps run in a loop of their own), every lanes-per-sequence value, and a forced number of longest
    targets on the swapped kernels (8 or 16 rows per lane, scores and end cells, all four modes)."""
    rng = np.random.RandomState(9100 + seed)
    for _ in range(6):
        A = int(rng.choice([4, 20, 24]))
        scale = int(rng.choice([1, 1, 3, 7, 13, 40]))
        m = (rng.randint(-5, 12, size=(A, A)) * scale).astype(np.int32)
        m = ((m + m.T) // 2).astype(np.int32)
        np.fill_diagonal(m, rng.randint(4, 14) * scale)
        go = int(rng.choice([0, 1, 3, 11, 40])) * (scale if rng.rand() < 0.5 else 1)
        ge = int(rng.choice([0, 1, 1, 2, 5, 25]))
        qlen = int(rng.choice([1, 3, 30, 64, 110, 200, 300, 333, 420, 640]))
        n = int(rng.randint(5, 160))
        lens = np.maximum(1, (rng.lognormal(5.0, 1.1, size=n)).astype(int)); lens = np.minimum(lens, 2500)
        lens[rng.randint(0, n)] = int(rng.choice([1, 2, 5, 7]))            # shorter than any lane pipeline
        if rng.rand() < 0.6:
            for _k in range(int(rng.randint(1, 4))):
                lens[rng.randint(0, n)] = int(rng.choice([511, 512, 513, 1024, 1025, 3000, 6000, 9000]))  # swapped-kernel chunk sizes
        q = rng.randint(0, A, size=qlen).astype(np.uint8)
        db = [rng.randint(0, A, size=int(L)).astype(np.uint8) for L in lens]
        for i in range(0, n, 4):  # near copies of the query: high scores
            L = len(db[i]); k = min(L, qlen); at = int(rng.randint(0, L - k + 1))
            db[i][at:at + k] = np.where(rng.rand(k) < 0.85, q[:k], db[i][at:at + k])
        mode = str(rng.choice(["SW", "SW", "NW", "HW", "OV"])); st = int(rng.randint(0, 2))
        lanes = str(rng.choice(["", "", "", "2", "8", "16", "64"]))
        for name in ("OPAL_HIP_LANES_PER_SEQ", "OPAL_HIP_SWAP", "OPAL_HIP_SWAP_ROWS"):
            opal_opt.unset(name)
        swap = ""
        if qlen <= 420 and rng.rand() < 0.7 and not (mode != "SW" and lanes):  # (SW: 16-bit swapped kernels; NW / HW / OV: the int32 ones)
            swap = str(int(rng.choice([1, 4, 25])))
            opal_opt.set("OPAL_HIP_SWAP", swap)
            opal_opt.set("OPAL_HIP_SWAP_ROWS", str(int(rng.choice([8, 16]))))
        elif lanes and qlen <= 40 * int(lanes):
            opal_opt.set("OPAL_HIP_LANES_PER_SEQ", lanes)
        got = ours.search(q, db, go, ge, m, st, MODES[mode], int(rng.randint(0, 2)))[:4]
        want = oracle.search(q, db, go, ge, m, st, MODES[mode])
        what = f"round-4 seed {seed}: A={A} x{scale} Q={qlen} n={n} gaps {go}/{ge} {mode} st={st} lanes={lanes or 'auto'} swap={swap or 'auto'}"
        _check(got, want, mode, what)


@pytest.mark.parametrize("mode", ["SW", "OV"])
def test_alignment_search_sharded_over_devices(mode, ours, opal_opt):
    """OPAL_SEARCH_ALIGNMENT with the call sharded over several devices ("0,0": two contexts on this box's one GPU): the
    operations traced per shard land in the right result structs -- same start cells and operation strings as unsharded."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    res, offs, lens = synth.database(900, seed=42, first=52000)
    db = [res[offs[i]:offs[i + 1]] for i in range(900)]
    q = synth.query(250, seed=3)
    m = blosum50()[1]
    rc, s, eq, et, sset, (arr, _) = ours.search(q, db, 3, 1, m, 2, MODES[mode])
    assert rc == 0
    sq, st, ops = ours.alignments(arr)
    opal_opt.set("OPAL_HIP_DEVICES", "0,0")
    rc2, s2, eq2, et2, sset2, (arr2, _) = ours.search(q, db, 3, 1, m, 2, MODES[mode])
    assert rc2 == 0
    sq2, st2, ops2 = ours.alignments(arr2)
    np.testing.assert_array_equal(s, s2)
    np.testing.assert_array_equal(eq, eq2)
    np.testing.assert_array_equal(et, et2)
    np.testing.assert_array_equal(sq, sq2)
    np.testing.assert_array_equal(st, st2)
    assert ops == ops2


def test_reentrant_streaming_and_alignment_from_host_threads(ours, oracle, opal_opt):
    """Four host threads at once, every one on the streaming kernels (persistent workgroups of different searches share
    the GPU: each must still run to completion) or in an alignment search with the traceback on the device."""
    import threading
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    opal_opt.set("OPAL_HIP_STREAM", "1")
    m = blosum50()[1]
    jobs = []
    for t in range(4):
        res, offs, lens = synth.database(700 + 100 * t, seed=600 + t)
        db = [res[offs[i]:offs[i + 1]] for i in range(len(lens))]
        q = synth.query([1100, 700, 1500, 333][t], seed=20 + t)
        jobs.append((q, db, ["SW", "NW", "OV", "HW"][t], [0, 1, 1, 2][t]))
    out = [None] * 4
    errs = []

    def run(i):
        q, db, mode, st = jobs[i]
        try:
            for _ in range(3):
                r = ours.search(q, db, 11, 1, m, st, MODES[mode])
                if st == 2:
                    ours.alignments(r[5][0])
                out[i] = r[:4]
        except Exception as e:  # noqa
            errs.append((i, repr(e)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for th in threads: th.start()
    for th in threads: th.join(timeout=300)
    assert not errs and all(not th.is_alive() for th in threads), errs
    for i, (q, db, mode, st) in enumerate(jobs):
        _check(out[i], oracle.search(q, db, 11, 1, m, min(st, 1), MODES[mode]), mode, f"thread {i} {mode} st={st}")


def test_device_ordering_equals_stable_sort_by_length():
    """order_kernel.hip (the ordering of a cold opalSearchDatabase call, on the device): the same inputs as the host ordering's
    test (tests/test_host_logic.py::test_ordering_is_a_stable_sort_by_length) plus sizes around the kernel's range cuts --
    stable by length, longest first, zero lengths listed apart in caller order, byte offsets = running sum, the absent item
    of an odd count; a length beyond the kernel's table is refused (the host orders then)."""
    import ctypes as C
    from opal_amd import _lib, synth
    L = _lib.lib()
    L.opalx_debug_order_device.restype = C.c_int
    L.opal_hip_order_max_length.restype = C.c_int
    rng = np.random.RandomState(5)
    cases_ = [np.array([0, 0], np.int32), np.array([7], np.int32), np.array([3, 0, 3, 9, 3], np.int32),
              rng.randint(0, 50, 1000).astype(np.int32), synth.lengths(42, 0, 100001, "sprot"),
              rng.randint(1, 4, 70000).astype(np.int32), np.concatenate([synth.lengths(42, 0, 66000), [0, 0, 35213]]).astype(np.int32),
              np.concatenate([synth.lengths(42, 0, 450000, "sprot"), [0, 17]]).astype(np.int32),
              rng.randint(0, 3, 2048 * 256 + 77).astype(np.int32), np.full(4097, 5, np.int32), synth.lengths(42, 0, 1000003)]
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    for lens in cases_:
        lens = np.ascontiguousarray(lens, np.int32)
        n = len(lens)
        cap = n + 2
        seq = np.full(cap, -7, np.int32); ln = np.full(cap, -7, np.int32); off = np.zeros(cap, np.uint64)
        emp = np.full(cap, -7, np.int32); nemp = C.c_int(-1)
        items = L.opalx_debug_order_device(ip(lens), n, ip(seq), ip(ln), off.ctypes.data_as(C.POINTER(C.c_ulonglong)), cap, ip(emp), C.byref(nemp))
        nz = np.nonzero(lens > 0)[0]
        want = nz[np.argsort(-lens[nz].astype(np.int64), kind="stable")]
        m = len(want)
        assert items == (m + 1) // 2 * 2, (n, items, m)
        np.testing.assert_array_equal(seq[:m], want)
        np.testing.assert_array_equal(ln[:m], lens[want])
        np.testing.assert_array_equal(off[:m], np.concatenate([[0], np.cumsum(lens[want].astype(np.int64))])[:m].astype(np.uint64))
        if items > m:
            assert (seq[m], ln[m], off[m]) == (-1, 0, int(lens[want].astype(np.int64).sum()))
        assert nemp.value == n - m
        np.testing.assert_array_equal(emp[:n - m], np.nonzero(lens <= 0)[0])
    too_long = np.array([5, L.opal_hip_order_max_length() + 1, 9], np.int32)
    seq = np.zeros(5, np.int32); ln = np.zeros(5, np.int32); off = np.zeros(5, np.uint64); emp = np.zeros(5, np.int32); nemp = C.c_int(0)
    assert L.opalx_debug_order_device(ip(too_long), 3, ip(seq), ip(ln), off.ctypes.data_as(C.POINTER(C.c_ulonglong)), 5, ip(emp), C.byref(nemp)) == -1


@pytest.mark.parametrize("device_order", ["0", "1"])
def test_api_results_do_not_depend_on_where_the_ordering_runs(device_order, oracle, opal_opt):
    """opalSearchDatabase with the ordering forced onto the host / onto the device (OPAL_HIP_DEVICE_ORDER): the same answers,
    the oracle's, on a database with empty sequences, duplicates of one length and an odd count; twice (the second call finds
    the database of the first)."""
    import opal_amd
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    opal_opt.set("OPAL_HIP_DEVICE_ORDER", device_order)
    n = 3001
    res, offs, lens = synth.database(n, seed=11)
    db = [res[offs[i]:offs[i + 1]] for i in range(n)]
    for i in (0, 17, 18, 2999):
        db[i] = db[i][:0]
    for i in range(100, 140):
        db[i] = db[99].copy()
    q = synth.query(150, seed=3)
    m = blosum50()[1]
    for mode_name in ("SW", "NW", "HW", "OV"):
        mode = opal_amd.MODES[mode_name]
        orc, os_, oq, ot = oracle.search(q, db, 3, 1, m, 1, mode)
        for rep in range(2):
            rc, results = opal_amd.search_database(q, db, 3, 1, m, opal_amd.OPAL_SEARCH_SCORE_END, mode)
            assert rc == orc == 0
            s, eq, et = opal_amd.result_fields(results, "score", "endLocationQuery", "endLocationTarget")
            live = os_ != 0 if mode_name == "SW" else np.ones(n, bool)
            np.testing.assert_array_equal(s, os_, err_msg=mode_name)
            np.testing.assert_array_equal(eq[live], oq[live], err_msg=mode_name)
            np.testing.assert_array_equal(et[live], ot[live], err_msg=mode_name)


@pytest.mark.parametrize("mode", ["NW", "HW", "OV"])
def test_chunked_query_with_32bit_prefix_at_forced_width16(mode, oracle, opal_opt):
    """Found by the planner sweep (round 5): a query in row chunks (2600 aa) whose longest targets need the 32-bit rung, with
    the 16-bit range forced onto the packed pairs (OPAL_HIP_NARROW=16/17/18 all end there for such a query). The pairs' chunks
    index the boundary buffer by token-stream offsets, the 32-bit prefix by residue offsets: run side by side on two streams
    they overwrote each other's boundary rows (garbage scores for the first items). The prefix now runs on the same stream."""
    from opal_amd import synth
    from opal_amd.api import PreparedDb
    from opal_amd.score_matrix import blosum50
    n = 6000
    res, offs, lens = synth.database(n, seed=42)
    pdb = PreparedDb(res, offs)
    q = synth.query(2600, seed=7)
    m = blosum50()[1]
    order = np.argsort(-lens.astype(np.int64), kind="stable")
    pick = np.unique(np.concatenate([order[:30], np.random.RandomState(2).choice(n, 20, replace=False)]))
    db = [res[offs[i]:offs[i + 1]] for i in pick]
    want = oracle.search(q, db, 11, 1, m, 1, MODES[mode])
    for narrow in ("16", "18"):
        opal_opt.set("OPAL_HIP_NARROW", narrow)
        rc, s, eq, et = pdb.search(q, 11, 1, m, 1, MODES[mode])
        assert rc == want[0] == 0
        np.testing.assert_array_equal(s[pick], want[1], err_msg=f"{mode} narrow={narrow}: score")
        np.testing.assert_array_equal(eq[pick], want[2], err_msg=f"{mode} narrow={narrow}: endQ")
        np.testing.assert_array_equal(et[pick], want[3], err_msg=f"{mode} narrow={narrow}: endT")


# ---- the a-priori score bound of the 16-bit rung (ladder.cpp: every query residue against its best letter) ----
_BOUND_M = np.array([[3, 5, -4, -4],     # asymmetric on purpose: query letter 0 scores best against TARGET letter 1 (5 > 3), while
                     [-4, 7, -4, -4],    # target letter 0 never scores more than 3 -- a bound taken over the wrong index would be too small
                     [-4, -4, 11, -4],
                     [-4, -4, -4, 15]], dtype=np.int32)


def _bound_case(counts, scale, seed, lengths, fillers=260):
    """A query with the given letter counts (shuffled), whose bound is scale * (5 n0 + 7 n1 + 11 n2 + 15 n3), and a database that
    REACHES that bound: the target that pairs every query residue with its best letter, alone and inside random flanks of the
    given total lengths, near copies of it, and unrelated sequences."""
    rng = np.random.RandomState(seed)
    q = np.repeat(np.arange(4, dtype=np.uint8), counts)
    rng.shuffle(q)
    best = np.where(q == 0, 1, q).astype(np.uint8)
    m = (_BOUND_M * scale).astype(np.int32)
    bound = int(scale * (5 * counts[0] + 7 * counts[1] + 11 * counts[2] + 15 * counts[3]))
    db = [best.copy()]
    for L in lengths:
        extra = max(0, L - len(best))
        left = int(rng.randint(0, extra + 1))
        db.append(np.concatenate([rng.randint(0, 4, size=left), best, rng.randint(0, 4, size=extra - left)]).astype(np.uint8))
    for i in range(12):  # near copies: a few substitutions, one deletion
        t = best.copy()
        t[rng.randint(0, len(t), size=1 + i)] = rng.randint(0, 4, size=1 + i)
        db.append(np.delete(t, int(rng.randint(0, len(t)))) if i % 2 else t)
    for i in range(fillers):
        db.append(rng.randint(0, 4, size=int(rng.randint(1, 700))).astype(np.uint8))
    return q, [db[i] for i in rng.permutation(len(db))], m, bound


@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
@pytest.mark.parametrize("which", ["sat-1", "sat", "int16", "int16+2", "skew-edge", "far"])
def test_scores_that_reach_the_a_priori_bound(mode, which, ours, oracle, opal_opt):
    """The 16-bit rung is chosen by a bound computed from the query's own residues (every residue against its best letter),
    and the database holds a target that attains it. The bound sits one below / at the saturation limit of the SW kernels
    (32767 - the largest entry), at the top of the int16 range and two beyond it, and where the diagonal-skewed kernels stop
    being safe (bound + (rows + columns) * extension = 32000: targets on either side of it); scheduler's choice and a forced
    stream of sequences."""
    if which == "sat-1":      # bound 32736 = 32767 - 30 - 1: "cannot saturate", no re-run list is read
        counts, scale, ge, lengths = (6, 34, 100, 1000), 2, 1, (1141, 1200, 1500)
    elif which == "sat":      # bound 32738: may saturate
        counts, scale, ge, lengths = (2, 37, 100, 1000), 2, 1, (1140, 1200, 1500)
    elif which == "int16":    # bound 32766: the largest that stays on the 16-bit kernels in NW / HW / OV
        counts, scale, ge, lengths = (2, 39, 100, 1000), 2, 1, (1141, 1200, 1500)
    elif which == "int16+2":  # bound 32768: targets of 1093 residues and more need 32 bits
        counts, scale, ge, lengths = (5, 37, 100, 1000), 2, 1, (1142, 1090, 1093, 1500)
    elif which == "skew-edge":  # bound 16383, extension 4: the streamed skewed kernels are safe to 1276 columns, the class kernels' to 2182
        counts, scale, ge, lengths = (2, 39, 100, 1000), 1, 4, tuple(range(1270, 1284)) + tuple(range(2176, 2190))
    else:                     # bound 16383 with extension 1: everything skewed
        counts, scale, ge, lengths = (2, 39, 100, 1000), 1, 1, (1141, 1300, 2500, 6000)
    q, db, m, bound = _bound_case(counts, scale, 77, lengths)
    go = 3 * ge + 8
    want = oracle.search(q, db, go, ge, m, 1, MODES[mode])
    assert int(want[1].max()) == bound, (int(want[1].max()), bound)  # (the bound is attained: free end gaps, or the bare target)
    for stream in ("", "1"):
        if stream:
            opal_opt.set("OPAL_HIP_STREAM", "1")
        got = ours.search(q, db, go, ge, m, 1, MODES[mode])[:4]
        _check(got, want, mode, f"bound {which} {mode} stream={stream or 'auto'}")
    got0 = ours.search(q, db, go, ge, m, 0, MODES[mode])[:4]
    assert got0[0] == want[0]
    np.testing.assert_array_equal(got0[1], want[1], err_msg=f"bound {which} {mode} score only")
