"""BASELINE.json-sized runs on the GPU, checked through size-independent properties and a sampled
oracle comparison (the oracle cannot finish 1e10 cells in seconds)."""
import os

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    from opal_amd import synth
    from opal_amd.api import PreparedDb
    from opal_amd.score_matrix import blosum50
    n = 100000
    res, offs, lens = synth.database(n, seed=42)
    q = synth.query(300, seed=7)
    return dict(res=res, offs=offs, lens=lens, q=q, m=blosum50()[1], pdb=PreparedDb(res, offs), n=n)


def test_c2_sw_sampled_vs_oracle_and_properties(c2, oracle):
    import ctypes as C
    from opal_amd import _lib
    pdb, q, m = c2["pdb"], c2["q"], c2["m"]
    out = {}
    buf = (C.c_longlong * 10)()
    _lib.lib().opalx_counters(buf, 10)
    swapped0 = int(buf[9])
    for mode in ("SW", "OV", "HW", "NW"):
        rc, s, eq, et = pdb.search(q, 11, 1, m, 1, oracle_lib.MODES[mode])
        assert rc == 0
        out[mode] = (s, eq, et)
    # a database without a length tail (clamped at 5 000 residues) and a 300-aa query: nothing belongs on the role-swapped
    # kernels (an optimistic planner model once sent the 5 000-residue targets there: C3 NW / HW / OV 1.15 -> 1.67 ms)
    _lib.lib().opalx_counters(buf, 10)
    assert int(buf[9]) == swapped0, "the planner sent targets of a tail-less database to the swapped kernels"
    # relaxation order: every global alignment is a semi-global one, etc.
    assert (out["SW"][0] >= out["OV"][0]).all() and (out["OV"][0] >= out["HW"][0]).all() and (out["HW"][0] >= out["NW"][0]).all()
    # NW/HW end at the last query row; NW at the last target column
    assert (out["NW"][1] == 299).all() and (out["NW"][2] == c2["lens"] - 1).all() and (out["HW"][1] == 299).all()
    # end locations inside the matrix
    for mode in out:
        s, eq, et = out[mode]
        live = s != 0 if mode == "SW" else np.ones(len(s), bool)
        assert (eq[live] >= 0).all() and (eq[live] < 300).all() and (et[live] >= 0).all() and (et[live] < c2["lens"][live]).all()
    # sampled bit-exact comparison with the oracle: the 40 longest, the 40 shortest, 300 random
    order = np.argsort(c2["lens"])
    rng = np.random.RandomState(1)
    pick = np.unique(np.concatenate([order[:40], order[-40:], rng.choice(c2["n"], 300, replace=False)]))
    db = [c2["res"][c2["offs"][i]:c2["offs"][i + 1]] for i in pick]
    for mode in out:
        rc, s, eq, et = oracle.search(q, db, 11, 1, m, 1, oracle_lib.MODES[mode])
        np.testing.assert_array_equal(out[mode][0][pick], s, err_msg=mode)
        np.testing.assert_array_equal(out[mode][1][pick], eq, err_msg=mode)
        np.testing.assert_array_equal(out[mode][2][pick], et, err_msg=mode)


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_c2_c3_every_sequence_vs_reference_library(c2):
    """SURVEY.md 8d, C3 as written: score, endQ, endT of EVERY sequence of the 100 000-sequence database, all four modes,
    against the reference's own AVX2 library (oracle/_ref, opal.cpp:1435-1519) run on every host core; and C2's score-only
    search for both gap settings. SW with score 0 has no defined end cell in the reference: masked."""
    ref = oracle_lib.RefThreads()
    if not ref.ok:
        pytest.skip("oracle/_ref/libref_bench.so lacks ref_search_run")
    pdb, q, m = c2["pdb"], c2["q"], c2["m"]
    for go in (11, 3):
        rc, s, eq, et = pdb.search(q, go, 1, m, 0, oracle_lib.MODE_SW)
        rrc, rs, _, _, sec = ref.search(q, c2["res"], c2["offs"], m, go, 1, 0, oracle_lib.MODE_SW)
        assert rc == rrc == 0
        np.testing.assert_array_equal(s, rs, err_msg=f"C2 score-only, gaps {go}/1")
    for go in (11, 3):
        for mode in ("SW", "NW", "HW", "OV"):
            rc, s, eq, et = pdb.search(q, go, 1, m, 1, oracle_lib.MODES[mode])
            rrc, rs, rq, rt, sec = ref.search(q, c2["res"], c2["offs"], m, go, 1, 1, oracle_lib.MODES[mode])
            assert rc == rrc == 0, (mode, go, rc, rrc)
            np.testing.assert_array_equal(s, rs, err_msg=f"C3 {mode} gaps {go}/1: scores")
            live = rs != 0 if mode == "SW" else np.ones(len(rs), bool)
            np.testing.assert_array_equal(eq[live], rq[live], err_msg=f"C3 {mode} gaps {go}/1: endQ")
            np.testing.assert_array_equal(et[live], rt[live], err_msg=f"C3 {mode} gaps {go}/1: endT")


def test_c2_order_independence_and_gaps(c2):
    """Per-sequence results do not depend on database order or on its neighbours (shuffle = permuted results)."""
    from opal_amd.api import PreparedDb
    n = 20000
    rng = np.random.RandomState(3)
    perm = rng.permutation(n)
    seqs = [c2["res"][c2["offs"][i]:c2["offs"][i + 1]] for i in range(n)]
    a = PreparedDb.from_sequences(seqs)
    b = PreparedDb.from_sequences([seqs[i] for i in perm])
    for go in (3, 11):
        rc, s1, q1, t1 = a.search(c2["q"], go, 1, c2["m"], 1, 3)
        rc2, s2, q2, t2 = b.search(c2["q"], go, 1, c2["m"], 1, 3)
        assert rc == rc2 == 0
        np.testing.assert_array_equal(s1[perm], s2)
        np.testing.assert_array_equal(q1[perm], q2)
        np.testing.assert_array_equal(t1[perm], t2)
    # a planted copy of the query scores the sum of its self-scores
    planted = seqs[:50] + [c2["q"].copy()]
    rc, s, eq, et = PreparedDb.from_sequences(planted).search(c2["q"], 11, 1, c2["m"], 1, 3)
    assert s[-1] == int(c2["m"][c2["q"], c2["q"]].sum()) and (eq[-1], et[-1]) == (299, 299)


def _sample(lens, n_random, seed=1, n_edge=40):
    order = np.argsort(lens, kind="stable")
    rng = np.random.RandomState(seed)
    return np.unique(np.concatenate([order[:n_edge], order[-n_edge:], rng.choice(len(lens), n_random, replace=False)]))


def _every_sequence_vs_reference(label, q, host_res, offs, m, go, ge, got, search_type=1):
    """Every sequence of a full-size search against the reference's own AVX2 library on all host cores (oracle/_ref,
    ref_search_run; the reference entry point is opal.cpp:1435-1519). `got` = {mode: (score, endQ, endT)} from the HIP path.
    SW with score 0 has no defined end cell in the reference: masked. A global mode whose reference call
    returns non-zero (its 32-bit NW / HW / OV rung is undefined behaviour) is compared on the sequences the
    reference did set and the rest is reported. Returns False where oracle/_ref is not there (the sampled scalar-oracle
    checks of the callers stay as the fallback)."""
    ref = oracle_lib.RefThreads() if oracle_lib.have_ref() else None
    if ref is None or not ref.ok:
        print(f"\n{label}: oracle/_ref absent -- sampled oracle comparison only")
        return False
    n = len(offs) - 1
    for mode, (s, eq, et) in got.items():
        rrc, rs, rq, rt, sec = ref.search(q, host_res, offs, m, go, ge, search_type, oracle_lib.MODES[mode])
        have = rs != np.iinfo(np.int32).min
        if mode == "SW":
            assert rrc == 0 and have.all(), (label, mode, rrc)
        elif rrc != 0:
            print(f"\n{label} {mode}: the reference returned {rrc} (its undefined 32-bit rung); comparing the {int(have.sum())} of {n} it set")
        np.testing.assert_array_equal(s[have], rs[have], err_msg=f"{label} {mode}: scores of every sequence vs the reference library")
        if search_type:
            live = have & (rs != 0) if mode == "SW" else have
            np.testing.assert_array_equal(eq[live], rq[live], err_msg=f"{label} {mode}: endQ")
            np.testing.assert_array_equal(et[live], rt[live], err_msg=f"{label} {mode}: endT")
        print(f"\n{label} {mode}{' + end' if search_type else ' score'}: all {int(have.sum())} sequences equal the reference library's ({sec:.1f} s on the host)")
    return True


def test_c2_score_only_is_the_kernel_the_bench_times(c2, oracle):
    """BASELINE.json configs[1] exactly as stated: SW SCORE-ONLY on 100 000 sequences (the instantiation bench.py
    times, opal_dp_kernel<17,SW,0,S>), both gap settings of SURVEY.md 8d. Every score must equal the score of the
    score+end instantiation, and a sample must equal the oracle."""
    pdb, q, m = c2["pdb"], c2["q"], c2["m"]
    pick = _sample(c2["lens"], 300)
    db = [c2["res"][c2["offs"][i]:c2["offs"][i + 1]] for i in pick]
    for go in (11, 3):
        rc0, s0, eq0, et0 = pdb.search(q, go, 1, m, 0, oracle_lib.MODE_SW)
        rc1, s1, eq1, et1 = pdb.search(q, go, 1, m, 1, oracle_lib.MODE_SW)
        assert rc0 == rc1 == 0
        np.testing.assert_array_equal(s0, s1)
        assert (eq0 == -1).all() and (et0 == -1).all()  # score-only leaves end locations at -1 (opal.cpp:423-426)
        orc, so, _, _ = oracle.search(q, db, go, 1, m, 0, oracle_lib.MODE_SW)
        assert orc == 0
        np.testing.assert_array_equal(s0[pick], so)
        assert (s0 > 0).all()


def test_device_generator_matches_numpy():
    import torch
    from opal_amd import synth
    ids = np.random.RandomState(2).choice(5000000, 3000, replace=False)
    lens = synth._lengths_of(42, ids)
    res, offs, _ = synth.database_subset(ids, 42, lens=lens)
    d, o = synth.residues_device(ids, lens, 42, torch.device("cuda", 0))
    assert (o == offs).all() and (d.cpu().numpy() == res).all()


@pytest.fixture(scope="module")
def c4shard():
    """One rank's share of BASELINE.json configs[3]: 5 000 000 synthetic sequences dealt to 8 GPUs (625 000 here)."""
    import torch
    from opal_amd import shard, synth
    from opal_amd.api import PreparedDb
    from opal_amd.fasta import read_fasta
    from opal_amd.score_matrix import blosum50
    import cases
    all_lens = synth.lengths(42, 0, 5000000)
    owned = shard.deal(all_lens, 8)[3]
    lens = all_lens[owned]
    d_res, offs = synth.residues_device(owned, lens, 42, torch.device("cuda", 0))
    torch.cuda.synchronize()
    q = read_fasta(os.path.join(cases.DATA, "query_P18080.fasta"), blosum50()[0])[0]
    pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
    return dict(owned=owned, lens=lens, offs=offs, q=q, m=blosum50()[1], pdb=pdb, keep=d_res)


def test_c4_shard_all_modes(c4shard, oracle):
    """P18080 (513 aa) x a 625 000-sequence shard, all four modes with end locations + SW score-only:
    size-independent properties on everything, bit-exact oracle comparison on a sample."""
    from opal_amd import synth
    pdb, q, m, lens, owned = c4shard["pdb"], c4shard["q"], c4shard["m"], c4shard["lens"], c4shard["owned"]
    Q = len(q)
    assert Q == 513 and 600000 < len(owned) < 650000
    out = {}
    for mode in ("SW", "OV", "HW", "NW"):
        rc, s, eq, et = pdb.search(q, 3, 1, m, 1, oracle_lib.MODES[mode])
        assert rc == 0, mode
        out[mode] = (s, eq, et)
    rc, s0, _, _ = pdb.search(q, 3, 1, m, 0, oracle_lib.MODE_SW)
    assert rc == 0
    np.testing.assert_array_equal(s0, out["SW"][0])
    assert (out["SW"][0] >= out["OV"][0]).all() and (out["OV"][0] >= out["HW"][0]).all() and (out["HW"][0] >= out["NW"][0]).all()
    assert (out["NW"][1] == Q - 1).all() and (out["NW"][2] == lens - 1).all() and (out["HW"][1] == Q - 1).all()
    for mode in out:
        s, eq, et = out[mode]
        live = s != 0 if mode == "SW" else np.ones(len(s), bool)
        assert (eq[live] >= 0).all() and (eq[live] < Q).all() and (et[live] >= 0).all() and (et[live] < lens[live]).all()
    pick = _sample(lens, 200, seed=4, n_edge=30)
    res, offs, _ = synth.database_subset(owned[pick], 42, lens=lens[pick])
    db = [res[offs[i]:offs[i + 1]] for i in range(len(pick))]
    for mode in out:
        rc, s, eq, et = oracle.search(q, db, 3, 1, m, 1, oracle_lib.MODES[mode])
        np.testing.assert_array_equal(out[mode][0][pick], s, err_msg=mode)
        np.testing.assert_array_equal(out[mode][1][pick], eq, err_msg=mode)
        np.testing.assert_array_equal(out[mode][2][pick], et, err_msg=mode)
    # SURVEY.md 8d, C4 as written: EVERY sequence of the shard against the reference itself
    host_res = c4shard["keep"].cpu().numpy()
    if _every_sequence_vs_reference("C4 shard (P18080 x 625 000)", q, host_res, c4shard["offs"], m, 3, 1, out):
        _every_sequence_vs_reference("C4 shard (P18080 x 625 000)", q, host_res, c4shard["offs"], m, 3, 1, {"SW": (s0, None, None)}, search_type=0)


def test_c5_long_query_buckets(c4shard, oracle):
    """BASELINE.json configs[4] at its stated size: a 2000-aa query on one rank's share of the C4 database (625 000 of the
    5 000 000 sequences, shard 3 of the 8-way deal), OPAL_OVERFLOW_BUCKETS, all four modes with end locations + SW
    score-only: size-independent properties on everything, bit-exact oracle comparison on a sample. BLOSUM50 cannot push a
    2000-aa alignment past 16 bits, so the forced 16 -> 32 bit re-run comes from planted self-matches under
    a x3-scaled matrix, as the survey prescribes (second half of the test)."""
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    pdb, lens, owned = c4shard["pdb"], c4shard["lens"], c4shard["owned"]
    assert 600000 < len(owned) < 650000
    q = synth.query(2000, seed=7)
    m = blosum50()[1]
    Q = len(q)
    pick = _sample(lens, 60, seed=5, n_edge=10)
    sres, soffs, _ = synth.database_subset(owned[pick], 42, lens=lens[pick])
    db = [sres[soffs[i]:soffs[i + 1]] for i in range(len(pick))]
    out, full = {}, {}
    for mode in ("SW", "OV", "HW", "NW"):
        rc, s, eq, et = pdb.search(q, 3, 1, m, 1, oracle_lib.MODES[mode], overflow_method=1)
        assert rc == 0, mode
        out[mode] = s
        full[mode] = (s, eq, et)
        orc, so, oq, ot = oracle.search(q, db, 3, 1, m, 1, oracle_lib.MODES[mode])
        np.testing.assert_array_equal(s[pick], so, err_msg=mode)
        np.testing.assert_array_equal(eq[pick], oq, err_msg=mode)
        np.testing.assert_array_equal(et[pick], ot, err_msg=mode)
        live = s != 0 if mode == "SW" else np.ones(len(s), bool)
        assert (eq[live] >= 0).all() and (eq[live] < Q).all() and (et[live] >= 0).all() and (et[live] < lens[live]).all()
        if mode in ("NW", "HW"):
            assert (eq == Q - 1).all()
        if mode == "NW":
            assert (et == lens - 1).all()
    assert (out["SW"] >= out["OV"]).all() and (out["OV"] >= out["HW"]).all() and (out["HW"] >= out["NW"]).all()
    rc, s0, _, _ = pdb.search(q, 3, 1, m, 0, oracle_lib.MODE_SW, overflow_method=1)
    assert rc == 0
    np.testing.assert_array_equal(s0, out["SW"])
    # SURVEY.md 8d, C5 as written: EVERY sequence of the shard against the reference itself (streaming kernels, two
    # sequences in flight per slot: where a rare interleaving would show)
    host_res = c4shard["keep"].cpu().numpy()
    _every_sequence_vs_reference("C5 shard (2000 aa x 625 000)", q, host_res, c4shard["offs"], m, 3, 1, full)
    del host_res
    from opal_amd.api import PreparedDb
    res, offs, _ = synth.database(3000, seed=42)
    # forced 16 -> 32 bit: x3 matrix, the query itself (and two mutilated copies) planted among 3000 sequences
    m3 = (m * 3).astype(np.int32)
    seqs = [res[offs[i]:offs[i + 1]] for i in range(3000)]
    planted = {17: q.copy(), 1501: q[:1700].copy(), 2999: np.concatenate([q[300:], q[:50]])}
    for k, v in planted.items():
        seqs[k] = v
    rc, s, eq, et = PreparedDb.from_sequences(seqs).search(q, 11, 1, m3, 1, oracle_lib.MODE_SW, overflow_method=1)
    assert rc == 0
    self_score = int(m3[q, q].sum())
    assert self_score > 32767 and s[17] == self_score and (eq[17], et[17]) == (1999, 1999)
    chk = sorted(planted) + [0, 1, 2, 1000, 2000]
    orc, so, oq, ot = oracle.search(q, [seqs[i] for i in chk], 11, 1, m3, 1, oracle_lib.MODE_SW)
    np.testing.assert_array_equal(s[chk], so)
    np.testing.assert_array_equal(eq[chk], oq)
    np.testing.assert_array_equal(et[chk], ot)
    assert (s[sorted(planted)] > 32767).all()


def test_c4_eight_way_equals_unsharded(opal_opt, oracle):
    """BASELINE.json configs[3] whole: P18080 x all 5 000 000 sequences. Once unsharded on the device-resident database
    (opalx_search_device), once through opalSearchDatabase itself with the in-library sharding over eight contexts
    (OPAL_HIP_DEVICES = eight times this box's one GPU: the deal, the per-shard pipelines and the host-side merge are
    those of an 8-GPU call) -- identical score / end-cell arrays, SW score-only and HW with end locations; and both against
    the oracle on 400 sequences drawn from ALL eight shards of the deal (50 each, the longest of every shard among them)."""
    import ctypes as C
    import torch
    import zlib
    from opal_amd import _lib, synth
    from opal_amd.api import OpalSearchResult, PreparedDb
    from opal_amd.fasta import read_fasta
    from opal_amd.score_matrix import blosum50
    import cases
    n = 5000000
    dev = torch.device("cuda", 0)
    lens = synth.lengths(42, 0, n)
    d_res, offs = synth.residues_device(np.arange(n, dtype=np.int64), lens, 42, dev)
    torch.cuda.synchronize()
    q = read_fasta(os.path.join(cases.DATA, "query_P18080.fasta"), blosum50()[0])[0]
    m = np.ascontiguousarray(blosum50()[1], np.int32)
    pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
    out = torch.empty(3 * n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    want = {}
    for mode, st in (("SW", 0), ("HW", 1)):
        out.fill_(-7)
        torch.cuda.synchronize()
        rc, _ = pdb.search_device(q, 3, 1, m, st, oracle_lib.MODES[mode], out[:n].data_ptr(), out[n:2 * n].data_ptr() if st else 0,
                                  out[2 * n:].data_ptr() if st else 0)
        assert rc == 0
        want[mode] = out.cpu().numpy().reshape(3, n).copy()
    pdb.close()
    # the oracle on a sample across the eight shards of the deal the 8-way call below will make (the same deal: shard.deal
    # mirrors deal_groups, tests/test_host_logic.py)
    from opal_amd import shard
    parts = shard.deal(lens, 8)
    rng = np.random.RandomState(8)
    pick = np.sort(np.concatenate([np.concatenate([p[np.argsort(lens[p])[-2:]], rng.choice(p, 48, replace=False)]) for p in parts]))
    assert len(pick) == 400
    sres, soffs, _ = synth.database_subset(pick, 42, lens=lens[pick])
    sdb = [sres[soffs[i]:soffs[i + 1]] for i in range(len(pick))]
    for mode, st in (("SW", 0), ("HW", 1)):
        orc, so, oq, ot = oracle.search(q, sdb, 3, 1, m, st, oracle_lib.MODES[mode])
        assert orc == 0
        np.testing.assert_array_equal(want[mode][0][pick], so, err_msg=f"{mode} vs oracle, sample over all eight shards")
        if st:
            np.testing.assert_array_equal(want[mode][1][pick], oq)
            np.testing.assert_array_equal(want[mode][2][pick], ot)
    res_host = d_res.cpu().numpy()
    del d_res, out
    torch.cuda.empty_cache()
    # SURVEY.md 8d, C4 whole: all 5 000 000 sequences against the reference itself
    if _every_sequence_vs_reference("C4 whole (P18080 x 5 000 000)", q, res_host, offs, m, 3, 1, {"SW": (want["SW"][0], None, None)}, search_type=0):
        _every_sequence_vs_reference("C4 whole (P18080 x 5 000 000)", q, res_host, offs, m, 3, 1, {"HW": tuple(want["HW"])})
    L = _lib.lib()
    lens32 = np.ascontiguousarray(lens, np.int32)
    ptrs = np.ascontiguousarray(res_host.ctypes.data + np.asarray(offs[:-1], dtype=np.uint64), dtype=np.uint64)
    arr = (OpalSearchResult * n)()
    words = np.frombuffer(arr, dtype=np.int32).reshape(n, 10)
    rptrs = np.ascontiguousarray(C.addressof(arr) + 40 * np.arange(n, dtype=np.uint64), dtype=np.uint64)
    opal_opt.set("OPAL_HIP_DEVICES", "0,0,0,0,0,0,0,0")
    for mode, st in (("SW", 0), ("HW", 1)):
        words[:, 0] = 0
        words[:, 2:6] = -1
        rc = L.opalSearchDatabase(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), C.c_void_p(ptrs.ctypes.data), n,
                                  lens32.ctypes.data_as(C.POINTER(C.c_int)), 3, 1, m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]),
                                  C.c_void_p(rptrs.ctypes.data), st, oracle_lib.MODES[mode], 1)
        assert rc == 0
        assert (words[:, 0] == 1).all()
        assert zlib.crc32(np.ascontiguousarray(words[:, 1]).tobytes()) == zlib.crc32(want[mode][0].tobytes()), f"{mode}: 8-way scores differ"
        if st:
            np.testing.assert_array_equal(words[:, 3], want[mode][1], err_msg="endQ")  # struct order: endLocationTarget, then Query
            np.testing.assert_array_equal(words[:, 2], want[mode][2], err_msg="endT")
        else:
            assert (words[:, 2] == -1).all() and (words[:, 3] == -1).all()
    L.opalx_release_cached()


def test_sprot_shape_full_size(oracle):
    """The shape the reference was published on (reference aligner_comparison.md:6-34; bench.py --workload sprot): 541 762
    synthetic sequences with a Swiss-Prot-like tail to 35 213 residues, BLOSUM50, gaps 3/1, the four published queries, the
    scheduler's own choice of kernels (asserted through opalx_counters: the role-swapped kernel takes the tail of the two
    short queries, the longer ones stream behind a head on the class body). SW scores for all four; SW / HW / OV / NW with
    end locations for P19930. The oracle on ALL sequences beyond 5 000 residues and 300 random ones; size-independent
    properties on everything."""
    import ctypes as C
    import torch
    import cases
    from opal_amd import _lib, synth
    from opal_amd.api import PreparedDb
    from opal_amd.fasta import read_fasta
    from opal_amd.score_matrix import blosum50

    names = ["stream_sw", "stream_global", "one_launch", "class_launches", "chunk_launches", "width32", "device_traced", "host_traced", "reused_fills", "swapped"]

    def counters():
        buf = (C.c_longlong * 10)()
        _lib.lib().opalx_counters(buf, 10)
        return dict(zip(names, [int(v) for v in buf]))
    n = synth.SPROT_SEQS
    dev = torch.device("cuda", 0)
    lens = synth.lengths(42, 0, n, "sprot")
    assert lens.max() == synth.SPROT_MAX_LEN
    d_res, offs = synth.residues_device(np.arange(n, dtype=np.int64), lens, 42, dev, stride=synth.SPROT_STRIDE)
    torch.cuda.synchronize()
    pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
    alphabet, m = blosum50()
    long_ones = np.nonzero(lens > 5000)[0]
    assert 200 < len(long_ones) < 300
    pick = np.unique(np.concatenate([long_ones, np.random.RandomState(3).choice(n, 300, replace=False)]))
    sres, soffs, _ = synth.database_subset(pick, 42, lens=lens[pick], stride=synth.SPROT_STRIDE)
    sdb = [sres[soffs[i]:soffs[i + 1]] for i in range(len(pick))]
    sw = {}
    for name in ("O74807", "P19930", "Q3ZAI3", "P18080"):
        q = read_fasta(os.path.join(cases.DATA, f"query_{name}.fasta"), alphabet)[0]
        c0 = counters()
        rc, s, eq, et = pdb.search(q, 3, 1, m, 0, oracle_lib.MODE_SW)
        c1 = counters()
        d = {k: c1[k] - c0[k] for k in names}
        assert rc == 0 and (eq == -1).all() and (et == -1).all()
        orc, so, _, _ = oracle.search(q, sdb, 3, 1, m, 0, oracle_lib.MODE_SW)
        assert orc == 0
        np.testing.assert_array_equal(s[pick], so, err_msg=f"{name}: SW score vs oracle ({len(pick)} sequences, every one beyond 5000 residues)")
        assert (s > 0).all() and (s <= int(m.max()) * np.minimum(len(q), lens)).all()
        print(f"\n{name} (Q={len(q)}): " + ", ".join(f"{k} {v}" for k, v in d.items() if v))
        if len(q) <= 200:   # short queries: the tail on the swapped kernel, the rest on class kernels
            assert d["swapped"] >= 10 and d["stream_sw"] == 0, d
        else:               # the head (longest sequences) on the class body inside the streaming launch
            assert d["stream_sw"] == 1 and d["swapped"] == 0, d
        sw[name] = s
    q = read_fasta(os.path.join(cases.DATA, "query_P19930.fasta"), alphabet)[0]
    Q = len(q)
    out = {}
    for mode in ("SW", "OV", "HW", "NW"):
        rc, s, eq, et = pdb.search(q, 3, 1, m, 1, oracle_lib.MODES[mode])
        assert rc == 0, mode
        orc, so, oq, ot = oracle.search(q, sdb, 3, 1, m, 1, oracle_lib.MODES[mode])
        assert orc == 0
        np.testing.assert_array_equal(s[pick], so, err_msg=mode)
        np.testing.assert_array_equal(eq[pick], oq, err_msg=mode)
        np.testing.assert_array_equal(et[pick], ot, err_msg=mode)
        out[mode] = (s, eq, et)
    np.testing.assert_array_equal(out["SW"][0], sw["P19930"], err_msg="SW score-only (swapped tail) != SW with end locations (class kernels)")
    # every one of the 541 762 sequences against the reference's own library (all host cores; NW left out: targets this long
    # need the reference's 32-bit NW, which is undefined behaviour and returns OPAL_ERR_OVERFLOW here)
    ref = oracle_lib.RefThreads() if oracle_lib.have_ref() else None
    if ref is not None and ref.ok:
        host_res = d_res.cpu().numpy()
        for mode in ("SW", "HW", "OV"):
            rrc, rs, rq, rt, sec = ref.search(q, host_res, offs, m, 3, 1, 1, oracle_lib.MODES[mode])
            assert rrc == 0, (mode, rrc)
            s, eq, et = out[mode]
            np.testing.assert_array_equal(s, rs, err_msg=f"sprot shape {mode}: scores of all sequences vs the reference library")
            live = rs != 0 if mode == "SW" else np.ones(len(rs), bool)
            np.testing.assert_array_equal(eq[live], rq[live], err_msg=f"sprot shape {mode}: endQ")
            np.testing.assert_array_equal(et[live], rt[live], err_msg=f"sprot shape {mode}: endT")
            print(f"P19930 {mode} + end: all {n} sequences equal the reference library's ({sec:.2f} s on the host)")
    assert (out["SW"][0] >= out["OV"][0]).all() and (out["OV"][0] >= out["HW"][0]).all() and (out["HW"][0] >= out["NW"][0]).all()
    assert (out["NW"][1] == Q - 1).all() and (out["NW"][2] == lens - 1).all() and (out["HW"][1] == Q - 1).all()
    for mode in ("SW", "OV", "HW"):
        s, eq, et = out[mode]
        assert (eq >= 0).all() and (eq < Q).all() and (et >= 0).all() and (et < lens).all()


def _run_bench(extra, env_extra=None):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(env_extra or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_two_rank_bench_on_one_gpu():
    """The N>1 path of bench.py with the REAL device search: `--gpus 2` starts two ranks by itself (both on GPU 0 here,
    results gathered through gloo); the gathered, re-ordered scores must equal the one-rank run's (same global database)."""
    common = ["--total-seqs", "30000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]
    one = _run_bench(["--gpus", "1"] + common)
    two = _run_bench(["--gpus", "2"] + common, {"BENCH_SHARE_GPU": "1", "BENCH_BACKEND": "gloo"})
    assert one["verified"] and two["verified"]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["verify"]["score_crc32"] == two["verify"]["score_crc32"]
    assert one["roofline"]["bound"] == "valu-issue" and 0 < one["roofline"]["frac"] < 1
    # what an N > 1 line needs to be read: every rank's kernel time, the deal's imbalance, the time of the exchange
    r = two["ranks"]
    assert len(r["kernel_ms"]["per_rank"]) == 2 and 0 < r["kernel_ms"]["min"] <= r["kernel_ms"]["max"]
    assert 1.0 <= r["imbalance"] < 1.02 and r["exchange_ms"] > 0
    assert one["in_flight"] == 2 and two["in_flight"] == 2 and one["ranks"]["exchange_ms"] is None
    # ... and the drop-in call over both "GPUs" from one process (OPAL_HIP_DEVICES=0,0 here), the other rank waiting on the store
    two_e2e = _run_bench(["--gpus", "2", "--total-seqs", "30000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         {"BENCH_SHARE_GPU": "1", "BENCH_BACKEND": "gloo"})
    assert two_e2e["verified"] and two_e2e["e2e"]["scores_match"] and two_e2e["e2e"]["ms"] > 0, two_e2e.get("e2e")
    # the same with a long query: both ranks on the streaming kernels (persistent workgroups of two processes share the GPU)
    common = ["--workload", "c5", "--total-seqs", "60000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]
    one = _run_bench(["--gpus", "1"] + common)
    two = _run_bench(["--gpus", "2"] + common, {"BENCH_SHARE_GPU": "1", "BENCH_BACKEND": "gloo"})
    assert one["verified"] and two["verified"] and two["n_gpus"] == 2
    assert one["verify"]["score_crc32"] == two["verify"]["score_crc32"]


def test_default_run_carries_the_strong_scaling_legs():
    """What the driver's N = 1, 2, 4, 8 runs of `bench.py --gpus N` (default flags) print: the weak-scaled C2 headline AND, in
    `also`, north_star's strong-scaling legs -- P18080 (C4) and a 2000-aa query (C5) x one total database dealt to the N GPUs, with
    every rank's kernel time, the deal's imbalance, the exchange time and a verdict of their own. Rehearsed here with a reduced
    total (BENCH_ALSO_TOTAL) at N = 1 and, over gloo on the one GPU, N = 2 (which also times the in-library multi-GPU handle
    with both gather kinds): the same cells per step at both N (strong scaling), both verified."""
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--settle-ms", "0"]
    env = {"BENCH_ALSO_TOTAL": "150000"}
    one = _run_bench(["--gpus", "1", "--no-e2e"] + common, env)
    two = _run_bench(["--gpus", "2"] + common, dict(env, BENCH_SHARE_GPU="1", BENCH_BACKEND="gloo"))
    for line, n in ((one, 1), (two, 2)):
        assert line["verified"] and line["scaling"] == "weak" and line["n_gpus"] == n
        for leg, qlen in (("c4", 513), ("c5", 2000)):
            r = line["also"][leg]
            assert r["verified"] and r["scaling"] == "strong" and r["value"] > 0, r
            assert len(r["kernel_ms_per_rank"]) == n and min(r["kernel_ms_per_rank"]) > 0 and 1.0 <= r["imbalance"] < 1.02
            assert (r["exchange_ms"] is None) == (n == 1)
            assert f"{qlen}-aa query x 150000" in r["workload"]
    mh = two["multi_handle"]
    assert mh["peer"]["scores_match"] and mh["rccl"]["scores_match"] and mh["peer"]["ms"] > 0
    assert mh["peer"]["gather_used"] == "peer"  # (one GPU listed twice: RCCL refuses, the handle falls back)


def test_eight_rank_bench_on_one_gpu():
    """The shape of the driver's 8-GPU run, rehearsed on the one GPU there is: `--gpus 8` starts eight ranks (all on GPU 0, gloo
    instead of RCCL, which refuses several ranks on one device), each takes its deal of the C4 workload (P18080, a 1 M-sequence
    cut of the 5 M database here), the gathered and re-ordered scores equal the one-rank run's, and the line carries what an N = 8
    line must: eight kernel times, the deal's imbalance, the exchange. With end locations too (three gathered vectors)."""
    common = ["--workload", "c4", "--total-seqs", "1000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]
    one = _run_bench(["--gpus", "1"] + common)
    eight = _run_bench(["--gpus", "8"] + common, {"BENCH_SHARE_GPU": "1", "BENCH_BACKEND": "gloo"})
    assert one["verified"] and eight["verified"] and eight["n_gpus"] == 8 and eight["scaling"] == "strong"
    assert one["verify"]["score_crc32"] == eight["verify"]["score_crc32"]
    r = eight["ranks"]
    assert len(r["kernel_ms"]["per_rank"]) == 8 and min(r["kernel_ms"]["per_rank"]) > 0
    assert 1.0 <= r["imbalance"] < 1.02 and r["exchange_ms"] > 0
    assert eight["config"]["cells_per_step"] == one["config"]["cells_per_step"]
    end = ["--workload", "c3", "--mode", "OV", "--total-seqs", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]
    one = _run_bench(["--gpus", "1"] + end)
    eight = _run_bench(["--gpus", "8"] + end, {"BENCH_SHARE_GPU": "1", "BENCH_BACKEND": "gloo"})
    assert one["verified"] and eight["verified"]
    assert one["verify"] == eight["verify"] or one["verify"]["score_crc32"] == eight["verify"]["score_crc32"]


def test_bench_rccl_branch_with_one_rank():
    """The RCCL side of bench.py on a one-GPU box (RCCL refuses two ranks on one GPU, so the two-rank test above runs on
    gloo): BENCH_FORCE_DIST=1 sends ONE rank through everything N ranks go through -- `nccl` process group bound to the
    device, the deal's permutation, dist.gather into the plan's block, the one index_select, barrier, the all-reduces of the
    time and of the verdict. Same scores as the plain one-GPU run."""
    common = ["--total-seqs", "30000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]
    plain = _run_bench(["--gpus", "1"] + common)
    forced = _run_bench(["--gpus", "1"] + common, {"BENCH_FORCE_DIST": "1"})
    assert plain["verified"] and forced["verified"]
    assert plain["verify"]["score_crc32"] == forced["verify"]["score_crc32"]
    forced_end = _run_bench(["--gpus", "1", "--workload", "c3", "--mode", "HW", "--seqs", "20000", "--steps", "2", "--warmup", "1",
                             "--no-cpu-baseline", "--no-e2e"], {"BENCH_FORCE_DIST": "1"})
    assert forced_end["verified"]


def test_bench_launcher_failing_rank_on_gpu():
    """`bench.py --gpus 2` with rank 1 dying before the rendezvous while rank 0 really waits in it (gloo, shared GPU): the
    parent kills rank 0 and returns non-zero within seconds instead of sitting out the rendezvous timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FAIL_RANK="1", BENCH_SHARE_GPU="1", BENCH_BACKEND="gloo")
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--total-seqs", "30000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-e2e"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "rank 1 exited with code 7" in p.stderr
    assert time.time() - t0 < 20


def test_api_context_reuse_and_multi_device(oracle, opal_opt):
    """opalSearchDatabase keeps its device buffers between calls (cached contexts) and can shard one call over
    several GPUs (OPAL_HIP_DEVICES; "0,0" = two contexts on this box's one GPU): growing and shrinking databases,
    interleaved modes, and the sharded call must all give the single-call answers."""
    import opal_amd
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    m = blosum50()[1]
    q = synth.query(180, seed=7)
    want = {}
    for n, first in ((500, 0), (6000, 1000), (64, 9000), (3000, 20000)):
        res, offs, lens = synth.database(n, seed=42, first=first)
        db = [res[offs[i]:offs[i + 1]] for i in range(n)]
        for mode in ("SW", "NW"):
            rc, results = opal_amd.search_database(q, db, 3, 1, m, opal_amd.OPAL_SEARCH_SCORE_END, opal_amd.MODES[mode])
            assert rc == 0
            got = opal_amd.result_fields(results, "score", "endLocationQuery", "endLocationTarget")
            pick = np.arange(0, n, max(1, n // 100))
            orc, so, oq, ot = oracle.search(q, [db[i] for i in pick], 3, 1, m, 1, opal_amd.MODES[mode])
            np.testing.assert_array_equal(got[0][pick], so)
            np.testing.assert_array_equal(got[1][pick], oq)
            np.testing.assert_array_equal(got[2][pick], ot)
            want[(n, mode)] = got
    opal_opt.set("OPAL_HIP_DEVICES", "0,0")
    try:
        res, offs, lens = synth.database(6000, seed=42, first=1000)
        db = [res[offs[i]:offs[i + 1]] for i in range(6000)]
        for mode in ("SW", "NW"):
            rc, results = opal_amd.search_database(q, db, 3, 1, m, opal_amd.OPAL_SEARCH_SCORE_END, opal_amd.MODES[mode])
            assert rc == 0
            got = opal_amd.result_fields(results, "score", "endLocationQuery", "endLocationTarget")
            for a, b in zip(got, want[(6000, mode)]):
                np.testing.assert_array_equal(a, b)
    finally:
        opal_opt.unset("OPAL_HIP_DEVICES")


def test_same_database_again_uploads_only_what_changed(oracle):
    """A caller that sends query after query against ONE database through the plain opalSearchDatabase: the cached context
    recognises the database (same lengths), compares the residues while gathering them and uploads only the chunks that
    differ. Same database, another query; a few mutated sequences (same lengths); every sequence mutated; a database of
    other lengths -- every answer must be the oracle's (stale device data would show as wrong scores)."""
    import ctypes as C
    import opal_amd
    from opal_amd import _lib, synth
    from opal_amd.score_matrix import blosum50

    def reused():
        buf = (C.c_longlong * 9)()
        _lib.lib().opalx_counters(buf, 9)
        return int(buf[8])
    m = blosum50()[1]
    n = 30000
    res, offs, lens = synth.database(n, seed=42, first=200000)
    db = [res[offs[i]:offs[i + 1]].copy() for i in range(n)]
    pick = np.arange(0, n, 250)

    def check(q, mode="SW"):
        rc, results = opal_amd.search_database(q, db, 11, 1, m, opal_amd.OPAL_SEARCH_SCORE_END, opal_amd.MODES[mode])
        assert rc == 0
        got = opal_amd.result_fields(results, "score", "endLocationQuery", "endLocationTarget")
        orc, so, oq, ot = oracle.search(q, [db[i] for i in pick], 11, 1, m, 1, opal_amd.MODES[mode])
        np.testing.assert_array_equal(got[0][pick], so)
        np.testing.assert_array_equal(got[1][pick], oq)
        np.testing.assert_array_equal(got[2][pick], ot)
        return got
    q1, q2 = synth.query(150, seed=7), synth.query(220, seed=8)
    r0 = reused()
    check(q1)                      # first sight: full fill
    assert reused() == r0
    check(q2)                      # same database, another query
    check(q1, "NW")
    assert reused() == r0 + 2
    for i in (0, 250, 7750, 29750):   # a few sequences change (same lengths): their chunks go up again
        db[i][:] = synth._residues(99 + i, 0, len(db[i]))
    check(q1)
    assert reused() == r0 + 3
    for i in range(n):                # everything changes
        db[i][0] = (db[i][0] + 1) % 20
    check(q2)
    assert reused() == r0 + 4
    db[5] = db[5][:-1].copy()         # another length: a different database
    check(q2)
    assert reused() == r0 + 4


@pytest.mark.parametrize("narrow", [None, "16", "18"])
def test_same_lengths_mutated_residues_long_query(narrow, oracle, opal_opt):
    """The repeat-search path of a re-seen database at the widths a 2000-aa query runs on: the speculative search of a
    database recognised by its lengths builds the width-16 pair tokens from whatever the device holds; when the comparison
    finds changed chunks the search is repeated -- on tokens rebuilt from the new residues (they used to stay marked valid:
    stale scores). Score-only and score + end, SW and NW, by default (packed pairs for SW score / NW, width 18 for SW + end)
    and with the narrow rung forced."""
    import opal_amd
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    if narrow:
        opal_opt.set("OPAL_HIP_NARROW", narrow)
    opal_opt.set("OPAL_HIP_STREAM", "0")  # (the class kernels: the streaming ones never read pair tokens)
    m = blosum50()[1]
    n = 6000
    res, offs, lens = synth.database(n, seed=42, first=400000)
    db = [res[offs[i]:offs[i + 1]].copy() for i in range(n)]
    q = synth.query(2000, seed=11)
    pick = np.arange(0, n, 150)
    for rnd in range(3):
        for mode, st in (("SW", 0), ("SW", 1), ("NW", 1)):
            rc, results = opal_amd.search_database(q, db, 11, 1, m, st, opal_amd.MODES[mode])
            assert rc == 0
            got = opal_amd.result_fields(results, "score", "endLocationQuery", "endLocationTarget")
            orc, so, oq, ot = oracle.search(q, [db[i] for i in pick], 11, 1, m, st, opal_amd.MODES[mode])
            np.testing.assert_array_equal(got[0][pick], so, err_msg=f"round {rnd} {mode} st={st} narrow={narrow}")
            if st:
                np.testing.assert_array_equal(got[1][pick], oq)
                np.testing.assert_array_equal(got[2][pick], ot)
            # every sequence changes under the same lengths before the next search (which will speculate, then repeat)
            for i in range(n):
                db[i][:] = synth._residues(5000 + 17 * rnd + i, 0, len(db[i]))


def test_same_lengths_different_residues_across_threads_and_alignment(oracle):
    """The nasty corner of the database recognition: databases with IDENTICAL lengths and DIFFERENT residues, searched from
    three host threads that keep exchanging cached contexts -- every call finds "the same database" by its lengths,
    searches speculatively and must notice that the residues are another database's. Plus OPAL_SEARCH_ALIGNMENT twice on
    one database (start cells, operation strings identical the second time)."""
    import threading
    import opal_amd
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    m = blosum50()[1]
    n = 8000
    _, offs, lens = synth.database(n, seed=42, first=300000)
    dbs = []
    for t in range(3):  # same lengths, residues from different streams
        flat = synth._residues(1000 + t, 0, int(offs[-1]))
        dbs.append([flat[offs[i]:offs[i + 1]] for i in range(n)])
    qs = [synth.query(140 + 30 * t, seed=20 + t) for t in range(3)]
    pick = np.arange(0, n, 100)
    want = [oracle.search(qs[t], [dbs[t][i] for i in pick], 11, 1, m, 1, opal_amd.MODES["SW"]) for t in range(3)]
    errors = []

    def run(t):
        try:
            for _ in range(6):
                rc, results = opal_amd.search_database(qs[t], dbs[t], 11, 1, m, opal_amd.OPAL_SEARCH_SCORE_END, opal_amd.MODES["SW"])
                got = opal_amd.result_fields(results, "score", "endLocationQuery", "endLocationTarget")
                assert rc == 0
                for a, b in zip(got, want[t][1:]):
                    np.testing.assert_array_equal(a[pick], b)
        except Exception as e:  # noqa
            errors.append((t, repr(e)[:300]))
    threads = [threading.Thread(target=run, args=(t,)) for t in range(3)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors
    # alignment search twice on one database: the second call recognises it; identical starts and operations
    ours = oracle_lib.OpalABI(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opal_amd", "csrc", "libopal.so"))
    small = dbs[0][:600]
    first = None
    for rep in range(2):
        rc, s, eq, et, sset, (arr, _) = ours.search(qs[0], small, 11, 1, m, 2, opal_amd.MODES["SW"])
        sq, st, ops = ours.alignments(arr)
        assert rc == 0
        if first is None:
            first = (s, eq, et, sq, st, ops)
            for i in range(0, 600, 60):
                assert oracle.check_alignment(qs[0], small[i], 11, 1, m, s[i], sq[i], st[i], eq[i], et[i], np.frombuffer(ops[i], np.uint8)) == 0
        else:
            for a, b in zip(first[:5], (s, eq, et, sq, st)):
                np.testing.assert_array_equal(a, b)
            assert first[5] == ops


def test_multi_gpu_handle_device_side_gather(c2):
    """opalx_multi_*: one database dealt to several GPUs of one process (here three contexts on this box's one GPU), every
    shard's result block gathered on the root GPU by peer copies and permuted into database order there -- no result
    touches the host. All modes, score-only and with end locations, against the single-GPU handle on 40 000 sequences."""
    import torch
    from opal_amd.api import PreparedDb, PreparedMultiDb
    n = 40000
    res, offs = c2["res"][: c2["offs"][n]], c2["offs"][: n + 1]
    q, m = c2["q"], c2["m"]
    one = PreparedDb(res, offs)
    multi = PreparedMultiDb(res, offs, [0, 0, 0])
    out = torch.full((3 * n,), -12345, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for mode in ("SW", "NW", "HW", "OV"):
        for st in (0, 1):
            rc0, s0, eq0, et0 = one.search(q, 3, 1, m, st, oracle_lib.MODES[mode])
            out.fill_(-12345)
            torch.cuda.synchronize()
            rc, kms = multi.search_device(q, 3, 1, m, st, oracle_lib.MODES[mode], out[:n].data_ptr(), out[n:2 * n].data_ptr() if st else 0,
                                          out[2 * n:].data_ptr() if st else 0)
            assert rc == rc0 == 0 and kms > 0
            got = out.cpu().numpy().reshape(3, n)
            np.testing.assert_array_equal(got[0], s0, err_msg=f"{mode} st={st} score")
            if st:
                np.testing.assert_array_equal(got[1], eq0, err_msg=f"{mode} endQ")
                np.testing.assert_array_equal(got[2], et0, err_msg=f"{mode} endT")
            else:
                assert (got[1:] == -12345).all()
    # three searches QUEUED on the multi-GPU handle (different queries and modes, result buffers of their own), waited for
    # first in, first out; a fourth is refused until one has been waited for
    from opal_amd import synth
    jobs = [(synth.query(250, seed=31), "SW", 0), (q, "HW", 1), (synth.query(180, seed=32), "NW", 1)]
    outs = [torch.full((3 * n,), -12345, dtype=torch.int32, device="cuda") for _ in jobs]
    torch.cuda.synchronize()
    for (qq, mode, st), o in zip(jobs, outs):
        assert multi.search_device_async(qq, 3, 1, m, st, oracle_lib.MODES[mode], o[:n].data_ptr(), o[n:2 * n].data_ptr() if st else 0,
                                         o[2 * n:].data_ptr() if st else 0) == 0
    assert multi.search_device_async(q, 3, 1, m, 0, oracle_lib.MODE_SW, outs[0].data_ptr()) != 0, "a fourth queued search must be refused"
    for (qq, mode, st), o in zip(jobs, outs):
        rc, kms = multi.wait()
        assert rc == 0 and kms > 0
        rc0, s0, eq0, et0 = one.search(qq, 3, 1, m, st, oracle_lib.MODES[mode])
        got = o.cpu().numpy().reshape(3, n)
        np.testing.assert_array_equal(got[0], s0, err_msg=f"queued {mode} score")
        if st:
            np.testing.assert_array_equal(got[1], eq0)
            np.testing.assert_array_equal(got[2], et0)
    assert multi.wait() == (0, 0.0)  # nothing queued any more
    multi.close()


def test_multi_gpu_handle_rccl_gather(c2, opal_opt):
    """The path's one collective THROUGH RCCL inside the library (ncclCommInitAll over the device list, one
    ncclGather of the padded {score, endQ, endT} blocks per search, OPAL_HIP_MULTI_GATHER=rccl) -- with the one rank a one-GPU
    box allows (RCCL refuses a device listed twice): communicator set-up, the gather on the shard's stream behind its kernels,
    the root's block permutation, queued searches. Results against the single handle. What N > 1 ranks add can only run on a
    multi-GPU node (unmeasured on hardware)."""
    import ctypes as C
    import torch
    from opal_amd import _lib, synth
    from opal_amd.api import PreparedDb, PreparedMultiDb
    n = 30001
    res, offs = c2["res"][: c2["offs"][n]], c2["offs"][: n + 1]
    q, m = c2["q"], c2["m"]
    opal_opt.set("OPAL_HIP_MULTI_GATHER", "rccl")
    multi = PreparedMultiDb(res, offs, [0])
    _lib.lib().opalx_multi_gather_kind.restype = C.c_int
    _lib.lib().opalx_multi_gather_kind.argtypes = [C.c_void_p]
    if _lib.lib().opalx_multi_gather_kind(multi._h) != 1:
        multi.close()
        pytest.skip("librccl.so could not be loaded here: the handle fell back to peer copies")
    one = PreparedDb(res, offs)
    out = torch.full((3 * n,), -12345, dtype=torch.int32, device="cuda")
    for mode in ("SW", "OV"):
        for st in (0, 1):
            rc0, s0, eq0, et0 = one.search(q, 3, 1, m, st, oracle_lib.MODES[mode])
            out.fill_(-12345)
            torch.cuda.synchronize()
            rc, kms = multi.search_device(q, 3, 1, m, st, oracle_lib.MODES[mode], out[:n].data_ptr(), out[n:2 * n].data_ptr() if st else 0,
                                          out[2 * n:].data_ptr() if st else 0)
            assert rc == rc0 == 0 and kms > 0
            got = out.cpu().numpy().reshape(3, n)
            np.testing.assert_array_equal(got[0], s0, err_msg=f"rccl gather {mode} st={st} score")
            if st:
                np.testing.assert_array_equal(got[1], eq0)
                np.testing.assert_array_equal(got[2], et0)
            else:
                assert (got[1:] == -12345).all()
    jobs = [(synth.query(250, seed=31), "SW", 0), (q, "HW", 1)]
    outs = [torch.full((3 * n,), -12345, dtype=torch.int32, device="cuda") for _ in jobs]
    torch.cuda.synchronize()
    for (qq, mode, st), o in zip(jobs, outs):
        assert multi.search_device_async(qq, 3, 1, m, st, oracle_lib.MODES[mode], o[:n].data_ptr(), o[n:2 * n].data_ptr() if st else 0,
                                         o[2 * n:].data_ptr() if st else 0) == 0
    for (qq, mode, st), o in zip(jobs, outs):
        rc, kms = multi.wait()
        assert rc == 0
        rc0, s0, eq0, et0 = one.search(qq, 3, 1, m, st, oracle_lib.MODES[mode])
        got = o.cpu().numpy().reshape(3, n)
        np.testing.assert_array_equal(got[0], s0, err_msg=f"rccl gather, queued {mode}")
        if st:
            np.testing.assert_array_equal(got[1], eq0)
            np.testing.assert_array_equal(got[2], et0)
    multi.close()


@pytest.mark.parametrize("mode", ["SW", "NW", "HW", "OV"])
def test_one_launch_kernels_equal_separate_launches(mode, c2, oracle, opal_opt):
    """The one-launch ("uber") kernels -- all length classes of a search as ONE grid -- against the same search as
    separate launches on forked streams (OPAL_HIP_NO_UBER=1): identical scores and end cells on 30 000 sequences of the
    C2 database (three classes: the 5000-residue tail, the long ones, the bulk), score only and with end locations."""
    from opal_amd.api import PreparedDb
    n = 30000
    pdb = PreparedDb(c2["res"][: c2["offs"][n]], c2["offs"][: n + 1])
    q, m = c2["q"], c2["m"]
    for st in (0, 1):
        for go in (11, 3):
            opal_opt.unset("OPAL_HIP_NO_UBER")
            a = pdb.search(q, go, 1, m, st, oracle_lib.MODES[mode])
            opal_opt.set("OPAL_HIP_NO_UBER", "1")
            b = pdb.search(q, go, 1, m, st, oracle_lib.MODES[mode])
            assert a[0] == b[0] == 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
    opal_opt.unset("OPAL_HIP_NO_UBER")
    pick = _sample(c2["lens"][:n], 120, seed=9, n_edge=25)
    db = [c2["res"][c2["offs"][i]:c2["offs"][i + 1]] for i in pick]
    rc, s, eq, et = pdb.search(q, 3, 1, m, 1, oracle_lib.MODES[mode])
    orc, so, oq, ot = oracle.search(q, db, 3, 1, m, 1, oracle_lib.MODES[mode])
    np.testing.assert_array_equal(s[pick], so)
    np.testing.assert_array_equal(eq[pick], oq)
    np.testing.assert_array_equal(et[pick], ot)


def test_queued_async_searches_first_in_first_out():
    """opalx_search_device_async: up to three searches queue up on a handle (different queries, modes and result buffers), run
    one after the other with no host round trip in between, and every opalx_db_wait returns the oldest one's kernel time;
    a fourth is refused without disturbing the queue; results equal those of the synchronous call."""
    import torch
    from opal_amd import synth
    from opal_amd.api import PreparedDb
    from opal_amd.score_matrix import blosum50
    n = 30000
    res, offs, lens = synth.database(n, seed=42)
    m = blosum50()[1]
    pdb = PreparedDb(res, offs)
    jobs = [(synth.query(200, seed=1), 0, 3), (synth.query(300, seed=2), 1, 0), (synth.query(450, seed=3), 1, 3)]   # (query, search type, mode)
    want = []
    for q, st, mode in jobs:
        out = torch.zeros(3 * n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        rc, _ = pdb.search_device(q, 11, 1, m, st, mode, out.data_ptr(), out[n:].data_ptr() if st else 0, out[2 * n:].data_ptr() if st else 0)
        assert rc == 0
        want.append(out.cpu().numpy().copy())
    for rounds in range(3):
        outs = [torch.zeros(3 * n, dtype=torch.int32, device="cuda") for _ in jobs]
        torch.cuda.synchronize()
        for (q, st, mode), out in zip(jobs, outs):
            assert pdb.search_device_async(q, 11, 1, m, st, mode, out.data_ptr(), out[n:].data_ptr() if st else 0, out[2 * n:].data_ptr() if st else 0) == 0
        extra = torch.zeros(n, dtype=torch.int32, device="cuda")
        assert pdb.search_device_async(jobs[0][0], 11, 1, m, 0, 3, extra.data_ptr()) != 0   # the queue is full
        times = []
        for _ in jobs:
            rc, kms = pdb.wait()
            assert rc == 0 and kms > 0
            times.append(kms)
        assert times[2] > times[0]          # 450-aa NW with end locations against 200-aa SW scores: first in, first out
        assert pdb.wait() == (0, 0.0)       # nothing outstanding: waits for the stream
        for out, w in zip(outs, want):
            np.testing.assert_array_equal(out.cpu().numpy(), w)
    # a longer query than any before: the header buffers grow behind queued searches
    big = synth.query(1500, seed=9)
    a = torch.zeros(n, dtype=torch.int32, device="cuda"); b = torch.zeros(n, dtype=torch.int32, device="cuda"); c = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert pdb.search_device_async(jobs[0][0], 11, 1, m, 0, 3, a.data_ptr()) == 0
    assert pdb.search_device_async(big, 11, 1, m, 0, 3, b.data_ptr()) == 0
    assert pdb.wait()[0] == 0 and pdb.wait()[0] == 0
    assert pdb.search_device(big, 11, 1, m, 0, 3, c.data_ptr())[0] == 0
    np.testing.assert_array_equal(a.cpu().numpy(), want[0][:n])
    np.testing.assert_array_equal(b.cpu().numpy(), c.cpu().numpy())
    # a search that launches nothing (empty query) BETWEEN two real ones keeps its own place in the queue: the first wait is
    # the first search's, the second reports 0 ms, the third is the last search's
    a.zero_(); b.fill_(7); c.zero_()
    torch.cuda.synchronize()
    assert pdb.search_device_async(jobs[0][0], 11, 1, m, 0, 3, a.data_ptr()) == 0
    assert pdb.search_device_async(np.zeros(0, np.uint8), 11, 1, m, 0, 3, b.data_ptr()) == 0
    assert pdb.search_device_async(jobs[1][0], 11, 1, m, 0, 0, c.data_ptr()) == 0
    t = [pdb.wait() for _ in range(3)]
    assert [x[0] for x in t] == [0, 0, 0] and t[0][1] > 0 and t[1][1] < 0.05 and t[2][1] > 0, t
    np.testing.assert_array_equal(a.cpu().numpy(), want[0][:n])
    assert (b.cpu().numpy() == 0).all()     # SW against an empty query: score 0 everywhere
    np.testing.assert_array_equal(c.cpu().numpy(), want[1][:n])
