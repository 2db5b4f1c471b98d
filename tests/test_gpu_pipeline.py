"""Queued searches on two streams (round 6; opalx_search_device_async / opalx_db_wait2, ladder.cpp: SearchPipe): a search's grid is
released when the search before it has started its last workgroup, so consecutive searches OVERLAP on the device. What they
share -- the handle's header slots, counters, scratch buffers, side streams -- must not leak from one into the other: every
queued search, whatever ran beside it, must give exactly the synchronous call's result arrays."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def _timeouts():
    """opal_hip_wait_count calls that gave up: a queued search whose wait for "the search before it has started all its
    workgroups" ran into its quarter-second patience means the count it waited for was wrong."""
    import ctypes as C
    from opal_amd import _lib
    f = _lib.lib().opal_hip_wait_timeouts
    f.restype = C.c_int
    return int(f())


def _sync(pdb, n, q, go, ge, m, st, mode):
    import torch
    out = torch.full((3 * n,), -99, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc, _ = pdb.search_device(q, go, ge, m, st, mode, out.data_ptr(), out[n:].data_ptr() if st else 0, out[2 * n:].data_ptr() if st else 0)
    assert rc == 0
    return out.cpu().numpy().copy()


def _queue(pdb, n, jobs, depth):
    """Runs the jobs through the queue `depth` deep; returns (result arrays, kernel_ms, period_ms)."""
    import torch
    outs = [torch.full((3 * n,), -99, dtype=torch.int32, device="cuda") for _ in jobs]
    torch.cuda.synchronize()
    got, kms, per = [], [], []
    queued = 0
    for (q, go, ge, m, st, mode), o in zip(jobs, outs):
        assert pdb.search_device_async(q, go, ge, m, st, mode, o.data_ptr(), o[n:].data_ptr() if st else 0, o[2 * n:].data_ptr() if st else 0) == 0
        queued += 1
        if queued == depth:
            rc, k, p = pdb.wait2()
            assert rc == 0
            kms.append(k); per.append(p); queued -= 1
    while queued:
        rc, k, p = pdb.wait2()
        assert rc == 0
        kms.append(k); per.append(p); queued -= 1
    torch.cuda.synchronize()
    for o in outs:
        got.append(o.cpu().numpy().copy())
    return got, kms, per


@pytest.mark.parametrize("pipeline", ["1", "2", "0"])
def test_queued_searches_of_every_kind_equal_the_synchronous_call(pipeline, opal_opt):
    """One-grid searches (SW score / SW + end / NW / HW / OV class grids, with and without the pair launch beside them), searches
    that are NOT one grid (a scaled matrix whose scores can saturate: re-run list and host round trip; a query in row chunks; a
    forced 32-bit rung) and streaming launches, interleaved so that every kind follows and precedes every other, two and three
    deep: identical arrays to the synchronous call. pipeline 1 = the default, 2 = two streams with full waits, 0 = one stream."""
    from opal_amd import synth
    from opal_amd.api import PreparedDb
    from opal_amd.score_matrix import blosum50
    opal_opt.set("OPAL_HIP_PIPELINE", pipeline)
    t0 = _timeouts()
    n = 24000
    res, offs, lens = synth.database(n, seed=42, first=50000)
    m = blosum50()[1]
    m3 = (m * 3).astype(np.int32)
    pdb = PreparedDb(res, offs)
    qa, qb, qc = synth.query(300, seed=7), synth.query(217, seed=8), synth.query(131, seed=9)
    qlong = synth.query(2000, seed=10)
    qchunk = synth.query(2700, seed=11)
    kinds = [
        (qa, 11, 1, m, 0, 3), (qb, 11, 1, m, 1, 3), (qa, 3, 1, m, 1, 0), (qc, 11, 1, m, 1, 1), (qb, 3, 1, m, 1, 2),   # one grid each
        (qa, 11, 1, m3, 1, 3),        # x3 matrix: 300 x 45 reaches the 16-bit limit -> the length of the re-run list is read back
        (qlong, 11, 1, m, 0, 3), (qlong, 11, 1, m, 1, 1),   # 2000 aa: streaming launches (SW score, HW)
        (qchunk, 11, 1, m, 0, 3),     # 2700 aa: row chunks through db->boundary
        (qc, 11, 1, m, 0, 0),
    ]
    want = [_sync(pdb, n, *k) for k in kinds]
    rng = np.random.RandomState(5)
    order = list(range(len(kinds))) + list(rng.permutation(len(kinds))) + list(rng.permutation(len(kinds)))
    for depth in (2, 3):
        got, kms, per = _queue(pdb, n, [kinds[i] for i in order], depth)
        for j, i in enumerate(order):
            st = kinds[i][4]
            w, g = want[i].reshape(3, n), got[j].reshape(3, n)
            np.testing.assert_array_equal(g[0], w[0], err_msg=f"pipeline={pipeline} depth={depth} job {j} (kind {i}): scores")
            if st:
                np.testing.assert_array_equal(g[1], w[1], err_msg=f"job {j} (kind {i}): endQ")
                np.testing.assert_array_equal(g[2], w[2], err_msg=f"job {j} (kind {i}): endT")
            else:
                assert (g[1:] == -99).all()
        assert all(k > 0 for k in kms) and all(p >= 0 for p in per)
    assert _timeouts() == t0, "a queued search waited in vain for the started-workgroup count of the search before it"
    pdb.close()


def test_queued_one_grid_searches_overlap_and_keep_their_own_queries(opal_opt):
    """Forty queued searches alternating between two queries and two modes on 100 000 sequences: every one must carry ITS query's
    answers (the header slots of searches in flight together are distinct) -- and with the pipeline on, the time a search costs
    the queue (the period between two ends) must be below the life of its own launch: they do overlap."""
    from opal_amd import synth
    from opal_amd.api import PreparedDb
    from opal_amd.score_matrix import blosum50
    n = 100000
    t0 = _timeouts()
    res, offs, lens = synth.database(n, seed=42)
    m = blosum50()[1]
    pdb = PreparedDb(res, offs)
    kinds = [(synth.query(300, seed=7), 11, 1, m, 0, 3), (synth.query(280, seed=17), 11, 1, m, 1, 1)]
    want = [_sync(pdb, n, *k) for k in kinds]
    jobs = [kinds[i & 1] for i in range(40)]
    ratio = {}
    for pipeline in ("1", "0"):
        opal_opt.set("OPAL_HIP_PIPELINE", pipeline)
        got, kms, per = _queue(pdb, n, jobs, 2)
        for j, g in enumerate(got):
            st = jobs[j][4]
            np.testing.assert_array_equal(g[:n], want[j & 1][:n], err_msg=f"pipeline={pipeline}: search {j} does not hold its own query's scores")
            if st:
                np.testing.assert_array_equal(g[n:], want[j & 1][n:])
        ratio[pipeline] = float(np.mean(per[5:])) / float(np.mean(kms[5:]))
    assert _timeouts() == t0
    print(f"\nperiod / own launch life: pipeline on {ratio['1']:.3f}, off {ratio['0']:.3f}")
    pdb.close()
    assert ratio["0"] > 0.99, ratio     # one stream: period = life + the gap between launches
    # overlapped: a search costs less than its launch lives. Whether two streams of one process overlap is the runtime's decision
    # (two streams on one hardware queue do not: DESIGN.md 3.13): a box on which they do not is reported, not failed -- the results
    # above are what this suite is about.
    if ratio["1"] >= 0.97:
        pytest.xfail(f"queued searches did not overlap on this box: period / launch life {ratio['1']:.3f}")
