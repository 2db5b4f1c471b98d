"""Long-tail parity on the GPU: the C2 database (100 000 synthetic sequences, clamped at 5 000 residues) with 41 planted
targets of 8 000 - 35 213 residues (tests/longtail.py), query lengths 110 / 513 / 2000, four modes x {score, score + end,
alignment}. Real protein databases have such tails (the reference was published on Swiss-Prot, reference
aligner_comparison.md:6; its own random targets stop at 4 799 residues, src/test.cpp:47-54).

Expected values: tests/golden/longtail.json -- recorded from the reference library itself for SW/HW/OV (scores, end
cells, start cells, crc32 of the operation strings) and from the scalar oracle for NW, whose 32-bit reference path is
undefined behaviour at these lengths (tests/golden/make_longtail.py). The rest of the database is checked on a sample
against the oracle. Every search also reports which kernels it ran on (opalx_counters)."""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

import longtail
import oracle_lib
from oracle_lib import MODES

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "opal_amd", "csrc", "libopal.so")
NAMES = ["stream_sw", "stream_global", "one_launch", "class_launches", "chunk_launches", "width32", "device_traced", "host_traced", "reused_fills", "swapped", "paired"]


def counters():
    from opal_amd import _lib
    buf = (C.c_longlong * 11)()
    _lib.lib().opalx_counters(buf, 11)
    return dict(zip(NAMES, [int(v) for v in buf]))


def delta(before):
    now = counters()
    return {k: now[k] - before[k] for k in NAMES}


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "longtail.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def base():
    from opal_amd import synth
    from opal_amd.score_matrix import blosum50
    n = 100000
    res, offs, lens = synth.database(n, seed=42)
    return dict(res=res, offs=offs, lens=lens, n=n, m=blosum50()[1])


def _with_planted(base, plants, nbase):
    """The first `nbase` sequences of the C2 database with the planted ones spread through it (database order matters to
    nothing: results come back in caller order). -> (list of sequences, positions of the planted ones)"""
    seqs = [base["res"][base["offs"][i]:base["offs"][i + 1]] for i in range(nbase)]
    step = max(1, nbase // (len(plants) + 1))
    pos = []
    for k, p in enumerate(plants):
        at = (k + 1) * step + k  # (earlier insertions have shifted everything behind them by one)
        seqs.insert(at, p)
        pos.append(at)
    return seqs, np.array(pos)


@pytest.mark.parametrize("qlen", longtail.QUERY_LENGTHS)
def test_longtail_search_score_and_end(qlen, base, gold, oracle):
    """100 041 sequences, four modes, score-only and score + end: planted targets vs the recorded reference answers, a
    sample of the rest vs the oracle, size-independent properties on everything."""
    from opal_amd.api import PreparedDb
    q = longtail.query(qlen)
    plants = longtail.planted(q)
    g = gold["queries"][str(qlen)]
    assert g["input_crc32"] == zlib.crc32(b"".join(np.ascontiguousarray(s).tobytes() for s in [q] + plants))
    seqs, pos = _with_planted(base, plants, base["n"])
    lens = np.array([len(s) for s in seqs])
    pdb = PreparedDb.from_sequences(seqs)
    m = base["m"]
    rng = np.random.RandomState(qlen)
    others = np.setdiff1d(rng.choice(len(seqs), 80, replace=False), pos)
    out, paths = {}, {}
    for mode in ("SW", "OV", "HW", "NW"):
        c0 = counters()
        rc1, s1, eq1, et1 = pdb.search(q, longtail.GAP_OPEN, longtail.GAP_EXT, m, 1, MODES[mode])
        paths[mode] = delta(c0)
        rc0, s0, eq0, et0 = pdb.search(q, longtail.GAP_OPEN, longtail.GAP_EXT, m, 0, MODES[mode])
        assert rc0 == rc1 == 0, mode
        np.testing.assert_array_equal(s0, s1, err_msg=f"{mode}: score-only != score+end")
        assert (eq0 == -1).all() and (et0 == -1).all()
        want = g[mode]
        np.testing.assert_array_equal(s1[pos], want["score"], err_msg=f"Q={qlen} {mode} planted score")
        np.testing.assert_array_equal(eq1[pos], want["endQ"], err_msg=f"Q={qlen} {mode} planted endQ")
        np.testing.assert_array_equal(et1[pos], want["endT"], err_msg=f"Q={qlen} {mode} planted endT")
        orc, so, oq, ot = oracle.search(q, [seqs[i] for i in others], longtail.GAP_OPEN, longtail.GAP_EXT, m, 1, MODES[mode])
        assert orc == 0
        np.testing.assert_array_equal(s1[others], so, err_msg=mode)
        np.testing.assert_array_equal(eq1[others], oq, err_msg=mode)
        np.testing.assert_array_equal(et1[others], ot, err_msg=mode)
        out[mode] = (s1, eq1, et1)
    print(f"\nQ={qlen} kernels per score+end search: " + "; ".join(f"{k}: {v}" for k, v in paths.items()))
    assert (out["SW"][0] >= out["OV"][0]).all() and (out["OV"][0] >= out["HW"][0]).all() and (out["HW"][0] >= out["NW"][0]).all()
    assert (out["NW"][1] == qlen - 1).all() and (out["NW"][2] == lens - 1).all() and (out["HW"][1] == qlen - 1).all()
    for mode in ("SW", "OV", "HW"):
        s, eq, et = out[mode]
        assert (eq >= 0).all() and (eq < qlen).all() and (et >= 0).all() and (et < lens).all()
    assert (out["SW"][0] > 0).all()
    # which path: NW targets of more than ~32 000 residues cannot stay in 16 bits (gap extension alone exceeds the range): the
    # role-swapped int32 kernels (short queries, round 4), one pair over several waves (pair_kernel.hpp, round 5), or -- with
    # both switched off -- the 32-bit class kernels
    assert paths["NW"]["width32"] + paths["NW"]["swapped"] + paths["NW"]["paired"] >= 1
    if qlen <= 1024:
        for mode in ("NW", "HW", "OV"):
            assert paths[mode]["swapped"] + paths[mode]["paired"] >= 8 and paths[mode]["width32"] == 0, (mode, paths[mode])
    # a 2000-aa query on 100 000 sequences: the global modes stream their 16-bit part, the targets beyond the 16-bit bounds (and
    # the long ones behind them) each on a workgroup of eight waves; SW does not stream -- eight 35 213-residue targets on 64
    # lanes each outlast the rest of this small database, so plan_stream leaves it to the class kernels
    if qlen == 2000:
        assert paths["HW"]["stream_global"] == 1 and paths["HW"]["paired"] >= 8 and paths["HW"]["width32"] == 0
        assert paths["SW"]["stream_sw"] + paths["SW"]["class_launches"] + paths["SW"]["one_launch"] >= 1


@pytest.mark.parametrize("qlen", longtail.QUERY_LENGTHS)
def test_longtail_alignment(qlen, base, gold, oracle, opal_opt):
    """OPAL_SEARCH_ALIGNMENT through the C ABI on 1 500 sequences + the planted ones: start cells and operation strings of
    the planted targets equal the reference's (SW/HW/OV; crc32 of the strings), every NW alignment re-scores to its
    score (35 213-column boxes), and the device traceback equals the host traceback string by string."""
    ours = oracle_lib.OpalABI(LIB)
    q = longtail.query(qlen)
    plants = longtail.planted(q)
    g = gold["queries"][str(qlen)]
    seqs, pos = _with_planted(base, plants, 1500)
    m = base["m"]
    for mode in ("SW", "HW", "OV", "NW"):
        c0 = counters()
        rc, s, eq, et, sset, (arr, _) = ours.search(q, seqs, longtail.GAP_OPEN, longtail.GAP_EXT, m, 2, MODES[mode])
        path = delta(c0)
        assert rc == 0, mode
        sq, st, ops = ours.alignments(arr)
        want = g[mode]
        np.testing.assert_array_equal(s[pos], want["score"])
        np.testing.assert_array_equal(eq[pos], want["endQ"])
        np.testing.assert_array_equal(et[pos], want["endT"])
        if mode != "NW":
            np.testing.assert_array_equal(sq[pos], want["startQ"], err_msg=f"Q={qlen} {mode} startQ")
            np.testing.assert_array_equal(st[pos], want["startT"], err_msg=f"Q={qlen} {mode} startT")
            assert [len(ops[i]) for i in pos] == want["opsLen"]
            assert [zlib.crc32(ops[i]) for i in pos] == want["opsCrc32"], f"Q={qlen} {mode}: operation strings differ from the reference's"
        check = list(pos) + list(range(0, 1500, 50))
        for i in check:
            assert oracle.check_alignment(q, seqs[i], longtail.GAP_OPEN, longtail.GAP_EXT, m, s[i], sq[i], st[i], eq[i], et[i],
                                          np.frombuffer(ops[i], np.uint8)) == 0, f"Q={qlen} {mode} #{i} does not re-score"
        # the same call with the host traceback: identical starts and strings everywhere
        opal_opt.set("OPAL_HIP_HOST_TRACEBACK", "1")
        rc2, s2, eq2, et2, _, (arr2, _) = ours.search(q, seqs, longtail.GAP_OPEN, longtail.GAP_EXT, m, 2, MODES[mode])
        opal_opt.unset("OPAL_HIP_HOST_TRACEBACK")
        sq2, st2, ops2 = ours.alignments(arr2)
        assert rc2 == 0
        np.testing.assert_array_equal(s, s2); np.testing.assert_array_equal(sq, sq2); np.testing.assert_array_equal(st, st2)
        assert ops == ops2, f"Q={qlen} {mode}: device and host traceback differ"
        print(f"\nQ={qlen} {mode} alignment search: {path}")
        assert path["device_traced"] >= len(pos)  # (the long boxes too: 35 213 columns x up to 2000 rows of flags each)
