"""CPU-only tests of the host side of the library: the C-ABI surface, the loud no-GPU failure, and
the host traceback (opalx_find_alignments) against the reference's recorded alignments."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import cases
import oracle_lib
from oracle_lib import MODES, OpalSearchResult

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("OPAL_LIB_PATH") or os.path.join(ROOT, "opal_amd", "csrc", "libopal.so")

pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="libopal.so not built (run __graft_entry__.build())")


def _declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(opalx?_?[A-Za-z0-9_]*|opal[A-Z][A-Za-z0-9]*)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(LIB)
    names = _declared_functions("opal.h") + _declared_functions("opal_hip.h")
    assert {"opalSearchDatabase", "opalSearchDatabaseCharSW", "opalInitSearchResult", "opalSearchResultIsEmpty",
            "opalSearchResultSetScore", "opal_hip_launch", "opalx_search_device", "opalx_find_alignments"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/ but not exported by libopal.so"


def test_library_exports_nothing_but_its_c_interface():
    """a drop-in library. Its dynamic symbol table holds the five functions of opal.h and the additive C
    entries of opal_hip.h (opalx_*, opal_hip_*) -- no C++ internals of the host scheduler, no kernel launch stubs
    (opal_amd/csrc/libopal.map): nothing that could interpose with the application it is linked into."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    five = {"opalSearchDatabase", "opalSearchDatabaseCharSW", "opalInitSearchResult", "opalSearchResultIsEmpty", "opalSearchResultSetScore"}
    assert five <= set(names)
    stray = [n for n in names if n not in five and not n.startswith("opalx_") and not n.startswith("opal_hip_")]
    assert not stray, stray[:10]
    declared = set(_declared_functions("opal.h") + _declared_functions("opal_hip.h"))
    undeclared = [n for n in names if n not in declared]
    assert not undeclared, f"exported but not declared in include/: {undeclared[:10]}"


def test_struct_layout_matches_reference_abi():
    # reference src/opal.h:47-74: 40 bytes, offsets 0/4/8/12/16/20/24/32
    assert C.sizeof(OpalSearchResult) == 40
    offs = [getattr(OpalSearchResult, f).offset for f, _ in OpalSearchResult._fields_]
    assert offs == [0, 4, 8, 12, 16, 20, 24, 32]


def test_result_helpers():
    abi = oracle_lib.OpalABI(LIB)
    r = OpalSearchResult()
    r.score = 99; r.scoreSet = 1; r.alignmentLength = 7
    abi.f_init(C.byref(r))
    assert (r.scoreSet, r.endLocationQuery, r.endLocationTarget, r.startLocationQuery, r.startLocationTarget) == (0, -1, -1, -1, -1)
    assert r.alignmentLength == 0 and not r.alignment and r.score == 99  # score is left alone (opal.cpp:1549-1555)
    assert abi.f_empty(r) == 1
    abi.f_set(C.byref(r), 5)
    assert abi.f_empty(r) == 0 and r.score == 5 and r.scoreSet == 1


def test_fails_loudly_without_a_gpu(capfd):
    """No CPU fallback: without a device the search returns OPAL_ERR_NO_SIMD_SUPPORT and says so."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    abi = oracle_lib.OpalABI(LIB)
    q, db, m = cases.materialise(cases.CASES["readme_example"])
    rc, s, eq, et, sset, _ = abi.search(q, db, 3, 1, m, 0, MODES["SW"])
    assert rc == 2 and (sset == 0).all()
    assert "no usable HIP device" in capfd.readouterr().err
    rc, *_ = abi.search(q, db, 3, 1, m, 0, 9)
    assert rc == 3  # invalid mode is reported before anything else


def _traceback(lib, q, db, go, ge, m, mode, score, eq, et, starts=None):
    n = len(db)
    arr = (OpalSearchResult * n)()
    ptrs = (C.POINTER(OpalSearchResult) * n)()
    for i in range(n):
        arr[i].scoreSet = 1; arr[i].score = int(score[i])
        arr[i].endLocationQuery = int(eq[i]); arr[i].endLocationTarget = int(et[i])
        arr[i].startLocationQuery = arr[i].startLocationTarget = -1
        ptrs[i] = C.pointer(arr[i])
    keep, dptrs, lens = oracle_lib._db_pointers(db)
    qq = np.ascontiguousarray(q, np.uint8); mm = np.ascontiguousarray(m, np.int32)
    args = (qq.ctypes.data_as(C.POINTER(C.c_ubyte)), len(qq), dptrs, n, lens.ctypes.data_as(C.POINTER(C.c_int)), int(go), int(ge),
            mm.ctypes.data_as(C.POINTER(C.c_int)), mm.shape[0], ptrs, int(mode))
    if starts is None:
        rc = lib.opalx_find_alignments(*args)
    else:  # start cells already known (as the device's reverse pass delivers them): only the start..end box is traced
        sq = np.ascontiguousarray(starts[0], np.int32); st = np.ascontiguousarray(starts[1], np.int32)
        rc = lib.opalx_find_alignments_from(*args, sq.ctypes.data_as(C.POINTER(C.c_int)), st.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    return arr


ALIGN_CASES = sorted(n for n, c in cases.CASES.items() if c.get("align"))


@pytest.mark.parametrize("name", ALIGN_CASES)
def test_traceback_matches_reference_alignments(name, oracle, golden):
    """Scores/end cells from the oracle -> host traceback -> start cells and operation strings equal to the
    ones the reference produced (tests/golden), and every alignment re-scores to its score."""
    lib = C.CDLL(LIB)
    abi = oracle_lib.OpalABI(LIB)
    case = cases.CASES[name]
    q, db, m = cases.materialise(case)
    for mode in case["modes"]:
        rc, s, eq, et = oracle.search(q, db, case["gap_open"], case["gap_ext"], m, 1, MODES[mode])
        arr = _traceback(lib, q, db, case["gap_open"], case["gap_ext"], m, MODES[mode], s, eq, et)
        ends = [(arr[i].endLocationQuery, arr[i].endLocationTarget) for i in range(len(db))]
        sq, st, ops = abi.alignments(arr)
        want = golden[name]["align"][mode]
        for i in range(len(db)):
            if mode == "SW" and s[i] == 0:
                assert (sq[i], st[i], ops[i]) == (-1, -1, b"") and ends[i] == (-1, -1)
                continue
            assert "".join(str(b) for b in ops[i]) == want["ops"][i], f"{name} {mode} #{i} operations"
            assert (sq[i], st[i]) == (want["startQ"][i], want["startT"][i]), f"{name} {mode} #{i} start"
            assert oracle.check_alignment(q, db[i], case["gap_open"], case["gap_ext"], m, s[i], sq[i], st[i],
                                          eq[i], et[i], np.frombuffer(ops[i], np.uint8)) == 0


@pytest.mark.parametrize("name", ALIGN_CASES)
def test_box_traceback_from_known_starts(name, oracle, golden):
    """With the start cells given (the goldens' own), tracing only the start..end box yields the same operation
    strings as the reference; unknown entries (INT_MIN) fall back to the full pass."""
    lib = C.CDLL(LIB)
    abi = oracle_lib.OpalABI(LIB)
    case = cases.CASES[name]
    q, db, m = cases.materialise(case)
    for mode in case["modes"]:
        rc, s, eq, et = oracle.search(q, db, case["gap_open"], case["gap_ext"], m, 1, MODES[mode])
        want = golden[name]["align"][mode]
        sq0 = np.array(want["startQ"], np.int32); st0 = np.array(want["startT"], np.int32)
        sq0[::3] = np.iinfo(np.int32).min  # every third one "unknown"
        arr = _traceback(lib, q, db, case["gap_open"], case["gap_ext"], m, MODES[mode], s, eq, et, starts=(sq0, st0))
        sq, st, ops = abi.alignments(arr)
        for i in range(len(db)):
            if mode == "SW" and s[i] == 0:
                assert (sq[i], st[i], ops[i]) == (-1, -1, b"")
                continue
            assert "".join(str(b) for b in ops[i]) == want["ops"][i], f"{name} {mode} #{i} operations"
            assert (sq[i], st[i]) == (want["startQ"][i], want["startT"][i]), f"{name} {mode} #{i} start"


def _gappy_pairs(seed, n, alphabet=4):
    """Targets that equal the query with long insertions / deletions: optimal alignments with long gaps, i.e.
    paths far from the diagonal (the reference traces inside a band, opal.cpp:1057-1179; ours does not)."""
    rng = np.random.RandomState(seed)
    q = rng.randint(0, alphabet, 260).astype(np.uint8)
    db = []
    for i in range(n):
        cut = sorted(rng.choice(np.arange(20, 240), 2, replace=False))
        ins = rng.randint(0, alphabet, rng.randint(30, 400)).astype(np.uint8)
        kind = i % 3
        if kind == 0:    # long insertion in the target
            t = np.concatenate([q[:cut[0]], ins, q[cut[0]:]])
        elif kind == 1:  # long deletion (target lacks a stretch of the query)
            t = np.concatenate([q[:cut[0]], q[cut[1]:]])
        else:            # both, plus flanks
            t = np.concatenate([ins[:50], q[:cut[0]], ins, q[cut[1]:], ins[:30]])
        db.append(t.astype(np.uint8))
    return q, db


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("mode", cases.ALL_MODES)
@pytest.mark.parametrize("gaps", [(3, 1), (2, 1), (6, 1)])
def test_traceback_long_gaps_vs_live_reference(mode, gaps, oracle):
    """Long, gappy alignments (cheap gaps, insertions of up to 400 columns): start cells and operation strings of
    the unbanded host traceback equal the reference's banded one (band vs no band, HW/OV stop test).
    (Gap settings with gapOpen == gapExt or gapExt == 0 are left out: the reference's own alignment pass crashes on
    this data with them -- its band algebra, opal.cpp:1057-1179 -- so there is nothing to compare with.)"""
    lib = C.CDLL(LIB)
    abi = oracle_lib.OpalABI(LIB)
    ref = oracle_lib.RefLib()
    q, db = _gappy_pairs(77, 18)
    m = cases.simple_matrix(4, 4, -3)
    go, ge = gaps
    rc, s, eq, et, sset, (rarr, _) = ref.search(q, db, go, ge, m, 2, MODES[mode], 0)
    assert rc == 0
    rsq, rst, rops = ref.alignments(rarr)
    arr = _traceback(lib, q, db, go, ge, m, MODES[mode], s, eq, et)
    sq, st, ops = abi.alignments(arr)
    np.testing.assert_array_equal(sq, rsq)
    np.testing.assert_array_equal(st, rst)
    assert ops == rops
    assert max(len(o) for o in ops) > 300
    # and with the starts given: same strings from the box alone
    arr = _traceback(lib, q, db, go, ge, m, MODES[mode], s, eq, et, starts=(rsq, rst))
    sq2, st2, ops2 = abi.alignments(arr)
    assert ops2 == rops and (sq2 == rsq).all() and (st2 == rst).all()


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("mode", cases.ALL_MODES)
def test_traceback_vs_live_reference(mode, oracle):
    """test.cpp-shaped data (4 letters, match 3 / mismatch -1, gaps 11/1): alignments equal the reference's."""
    lib = C.CDLL(LIB)
    abi = oracle_lib.OpalABI(LIB)
    ref = oracle_lib.RefLib()
    q, db = cases.random4(4242, 300, 12, 4, 200, 900)
    m = cases.simple_matrix(4, 3, -1)
    rc, s, eq, et, sset, (rarr, _) = ref.search(q, db, 11, 1, m, 2, MODES[mode], 0)
    rsq, rst, rops = ref.alignments(rarr)
    arr = _traceback(lib, q, db, 11, 1, m, MODES[mode], s, eq, et)
    sq, st, ops = abi.alignments(arr)
    np.testing.assert_array_equal(sq, rsq)
    np.testing.assert_array_equal(st, rst)
    assert ops == rops


def test_ordering_is_a_stable_sort_by_length():
    """The packed database's item order (db_pack.cpp, spread over the host pool's threads): stable by length, longest first,
    zero lengths left out, byte offsets = running sum -- against numpy's stable argsort, on sizes below and above the
    threshold at which the ordering runs on several threads, with a long tail and with duplicate-heavy lengths."""
    import ctypes as C
    from opal_amd import synth
    L = C.CDLL(LIB)
    L.opalx_debug_order.restype = C.c_int
    rng = np.random.RandomState(5)
    cases = [np.zeros(0, np.int32), np.array([0, 0], np.int32), np.array([7], np.int32),
             rng.randint(0, 50, 1000).astype(np.int32), synth.lengths(42, 0, 100001, "sprot"),
             rng.randint(1, 4, 70000).astype(np.int32), np.concatenate([synth.lengths(42, 0, 66000), [0, 0, 35213]]).astype(np.int32),
             np.concatenate([synth.lengths(42, 0, 450000, "sprot"), [0, 17]]).astype(np.int32)]  # (>= 400 000: several threads)
    for lens in cases:
        n = len(lens)
        cap = n + 2
        seq = np.full(cap, -7, np.int32); ln = np.full(cap, -7, np.int32); off = np.zeros(cap, np.uint64)
        items = L.opalx_debug_order(lens.ctypes.data_as(C.POINTER(C.c_int)), n, seq.ctypes.data_as(C.POINTER(C.c_int)),
                                    ln.ctypes.data_as(C.POINTER(C.c_int)), off.ctypes.data_as(C.POINTER(C.c_ulonglong)), cap)
        nz = np.nonzero(lens > 0)[0]
        want = nz[np.argsort(-lens[nz].astype(np.int64), kind="stable")]
        m = len(want)
        assert items == (m + 1) // 2 * 2
        np.testing.assert_array_equal(seq[:m], want)
        np.testing.assert_array_equal(ln[:m], lens[want])
        np.testing.assert_array_equal(off[:m], np.concatenate([[0], np.cumsum(lens[want].astype(np.int64))])[:m].astype(np.uint64))
        if items > m:
            assert (seq[m], ln[m], off[m]) == (-1, 0, int(lens[want].astype(np.int64).sum()))


def test_host_pool_runs_every_task_exactly_once():
    """The library's host thread pool (host_pool.hpp): jobs of 1..300 tasks, synchronous and begin/end in turn, from one
    caller and from four callers at once; every task of every job exactly once."""
    lib = C.CDLL(LIB)
    lib.opalx_debug_pool_selftest.restype = C.c_int
    assert lib.opalx_debug_pool_selftest(3000, 1) == 0
    assert lib.opalx_debug_pool_selftest(1500, 4) == 0


def test_host_code_sanitized():
    """The host translation units rebuilt with UndefinedBehaviorSanitizer (opal_amd/csrc/Makefile: libopal_ubsan.so) run
    this module's tests -- API rules, ordering on several threads, the thread pool, the host traceback against the
    reference's alignments -- in a child process; any undefined behaviour aborts it."""
    import subprocess
    import sys
    if os.environ.get("OPAL_LIB_PATH"):
        pytest.skip("already running against an alternative library")
    csrc = os.path.join(ROOT, "opal_amd", "csrc")
    try:
        subprocess.check_call(["make", "-s", "-j", "4", "-C", csrc, "libopal_ubsan.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("sanitized build not available here")
    env = dict(os.environ, OPAL_LIB_PATH=os.path.join(csrc, "libopal_ubsan.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "not sanitized"], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_host_scheduler_under_thread_sanitizer():
    """The host scheduler's concurrency under ThreadSanitizer (tests/fake_device): the library's real host translation
    units -- thread pool, gather beside the search, chunk gates, the speculative search of a re-seen database, zero-copy
    fills, the queue of asynchronous searches, two caller threads, a call sharded over three device slots, the multi-GPU
    handle's queued searches -- compiled with -fsanitize=thread against a FAKE device whose streams are host threads and
    whose kernels are the scalar oracle. Every answer must be the oracle's and ThreadSanitizer must stay silent (a report
    makes the exit code 66). The fake device is test infrastructure: libopal.so never contains it."""
    import subprocess
    d = os.path.join(ROOT, "tests", "fake_device")
    try:
        subprocess.check_call(["make", "-s", "-j", "4", "-C", d, "host_stress"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("ThreadSanitizer build not available here")
    env = dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=1")
    for k in list(env):
        if k.startswith("OPAL_HIP_") or k.startswith("OPAL_HOST_"):
            del env[k]
    p = subprocess.run([os.path.join(d, "host_stress")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "host_stress ok" in p.stdout, p.stdout[-2000:] + p.stderr[-6000:]


def test_options_are_typed_and_validated(opal_opt):
    """opalx_set_option (options.cpp): unknown names and values that do not parse are refused and change nothing; the
    environment is not read again after the first use (a later setenv has no effect)."""
    L = C.CDLL(LIB)
    f = L.opalx_set_option
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_char_p]
    assert f(b"OPAL_HIP_NO_SUCH_OPTION", b"1") == 1
    for name, bad in ((b"OPAL_HIP_LANES_PER_SEQ", b"3"), (b"OPAL_HIP_NARROW", b"19"), (b"OPAL_HIP_STREAM", b"yes"), (b"OPAL_HIP_STREAM_CHUNKS", b"16"),
                      (b"OPAL_HIP_STREAM_CHUNKS", b"16,41"), (b"OPAL_HIP_DEVICES", b"0;1"), (b"OPAL_HIP_TAIL_FRACTION", b"-1"), (b"OPAL_HIP_SWAP_ROWS", b"12"),
                      (b"OPAL_HIP_SLICES", b"1,,2"), (b"OPAL_HIP_FORCE_WIDTH", b"16"), (b"OPAL_HOST_THREADS", b"0")):
        assert f(name, bad) == 2, (name, bad)
    for name, good in ((b"OPAL_HIP_LANES_PER_SEQ", b"16"), (b"OPAL_HIP_NARROW", b"18"), (b"OPAL_HIP_STREAM", b"1"), (b"OPAL_HIP_STREAM_CHUNKS", b"16,20"),
                       (b"OPAL_HIP_DEVICES", b"all"), (b"OPAL_HIP_DEVICES", b"0,0"), (b"OPAL_HIP_TAIL_FRACTION", b"0.5"), (b"OPAL_HIP_SLICES", b"1,3,5")):
        assert f(name, good) == 0, (name, good)
        assert f(name, None) == 0
    # the streaming plan follows the option, not the environment of the moment
    L.opalx_debug_plan_stream.restype = C.c_int
    L.opalx_debug_plan_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_int)]
    out = (C.c_int * 4)()
    os.environ["OPAL_HIP_STREAM"] = "1"
    try:
        L.opalx_debug_plan_stream(300, 24, 3, 0, 100000 * 350, 3776, 100000, out)
        assert tuple(out) == (0, 0, 0, 0), "the environment was read again"
    finally:
        del os.environ["OPAL_HIP_STREAM"]
    opal_opt.set("OPAL_HIP_STREAM", "1")
    L.opalx_debug_plan_stream(300, 24, 3, 0, 100000 * 350, 3776, 100000, out)
    assert out[0] == 17 and out[1] >= 1


def test_streaming_plan_of_the_scheduler(opal_opt):
    """planner.cpp::plan_stream on the host alone: which searches stream (and on what geometry), which stay on the class kernels."""
    import ctypes as C
    L = C.CDLL(LIB)
    L.opalx_debug_plan_stream.restype = C.c_int
    L.opalx_debug_plan_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_int)]
    for e in ("OPAL_HIP_STREAM", "OPAL_HIP_LANES_PER_SEQ", "OPAL_HIP_NARROW", "OPAL_HIP_STREAM_CHUNKS", "OPAL_HIP_NO_STREAM_CHUNKS"):
        opal_opt.unset(e)
    SW, NW = 3, 0
    def plan(q, mode=SW, end=0, nseq=100000, mean=350, longest=3776, a=24):
        out = (C.c_int * 4)()
        assert L.opalx_debug_plan_stream(q, a, mode, end, nseq * mean, longest, nseq, out) == 0
        return tuple(out)
    # a 300-aa query on 100 000 sequences: three sequences per slot -- class kernels; on a million: streams on 8 lanes
    assert plan(300) == (0, 0, 0, 0)
    w, g, rows, passes = plan(300, nseq=1000000)
    assert (w, g, passes) == (17, 8, 1) and g * rows >= 300
    # 2000 aa: 64 lanes, 2-byte profile entries, one launch
    assert plan(2000) == (18, 64, 32, 1)
    # beyond one wave: row chunks of equal size on 32 lanes, 4-byte entries (SW with and without end locations, NW)
    assert plan(3000) == (17, 32, 32, 3)
    assert plan(3000, end=1) == (17, 32, 32, 3)
    assert plan(3000, mode=NW, end=1) == (17, 32, 32, 3)
    w, g, rows, passes = plan(10000)
    assert w == 17 and passes >= 8 and passes * g * rows >= 10000 and (passes - 1) * g * rows < 10000
    # a small database: the longest sequence would outlast the whole search on its lanes -- class kernels
    assert plan(2000, nseq=300) == (0, 0, 0, 0)
    assert plan(3000, nseq=300) == (0, 0, 0, 0)
    # ... but with every lane of a wave on one sequence (64 lanes) the class kernels have nothing wider to offer the longest
    # sequence: a 2000-aa query streams from 1000 sequences on (planner sweep, profiles/r05_planner_sweep.txt)
    assert plan(2000, nseq=2000) == (18, 64, 32, 1)
    # a query in row chunks streams only with enough sequences per slot (20 000 on 8 192 slots: class-kernel chunks)
    assert plan(3000, nseq=20000) == (0, 0, 0, 0)
    # a 35 000-residue sequence among 540 000 (the published shape), 513-aa query: not as a whole ...
    assert plan(513, nseq=541762, mean=356, longest=35213) == (0, 0, 0, 0)
    # ... but what is left behind its 128 longest sequences streams on 16 lanes
    w, g, rows, passes = plan(513, nseq=541762 - 128, mean=355, longest=9000)
    assert (w, g, passes) == (17, 16, 1) and rows == 33
    # SW with end locations at 16 x 37..40 rows (the streams stop at 36): the class kernels at 16 lanes, not a 32-lane stream
    # (608 aa on the published shape: 25.5 -> 20.8 ms); one step up it streams (1216 aa at 64 x 19: 47 ms against 60)
    assert plan(608, end=1, nseq=541762, mean=360, longest=5000) == (0, 0, 0, 0)
    assert plan(560, end=1, nseq=541762, mean=360, longest=5000)[1] == 16
    assert plan(608, end=0, nseq=541762, mean=360, longest=5000)[1] == 16
    assert plan(1216, end=1, nseq=541762, mean=360, longest=5000)[1] == 64
    # an alphabet whose profile leaves no room for a streaming geometry
    assert plan(3000, a=250) == (0, 0, 0, 0)


def test_no_kernel_keeps_dp_state_in_scratch():
    """The DP kernels are built with resource remarks and the build itself fails on a spilled VGPR or a demoted register array
    (opal_amd/csrc/Makefile, inst_%.o); `make check-scratch` applies the same rule to the remarks of the objects as built --
    seconds, nothing is recompiled when the library is up to date -- and lists the kernels that park SGPRs."""
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc here")
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "opal_amd", "csrc"), "-j8", "check-scratch"], capture_output=True, text=True, timeout=3000)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "no kernel uses scratch for DP state" in p.stdout
    assert "VGPR spills" not in p.stdout


def test_score_bound_of_the_16_bit_rung():
    """ladder.cpp::query_score_bound on the host alone: every query residue against its best letter, the QUERY letter on the rows of
    the matrix (score_matrix[query * A + target], as opal.h and opal.cpp:260-268), never below 0 per residue, the largest entry
    for a residue outside the alphabet. (tests/test_gpu_parity.py::test_scores_that_reach_the_a_priori_bound holds searches whose
    scores attain it.)"""
    import ctypes as C
    from opal_amd.score_matrix import blosum50
    L = C.CDLL(LIB)
    L.opalx_debug_score_bound.restype = C.c_longlong
    L.opalx_debug_score_bound.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_int), C.c_int]

    def bound(q, m):
        q = np.ascontiguousarray(q, dtype=np.uint8); m = np.ascontiguousarray(m, dtype=np.int32)
        a = int(round(m.size ** 0.5))
        return int(L.opalx_debug_score_bound(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), m.ctypes.data_as(C.POINTER(C.c_int)), a))

    # asymmetric: query letter 0 scores 5 against target letter 1, target letter 0 never more than 3
    m = np.array([[3, 5, -4, -4], [-4, 7, -4, -4], [-4, -4, 11, -4], [-4, -4, -4, 15]], dtype=np.int32)
    assert bound([0], m) == 5 and bound([1], m) == 7 and bound([0, 1, 2, 3, 3], m) == 5 + 7 + 11 + 15 + 15
    assert bound([], m) == 0
    # a letter that scores negative against everything contributes nothing; one outside the alphabet the largest entry
    m2 = np.array([[-2, -3], [-1, 6]], dtype=np.int32)
    assert bound([0, 0, 1], m2) == 6 and bound([7], m2) == 6 and bound([0, 200, 1], m2) == 12
    # random queries under BLOSUM50 and a random asymmetric matrix: the numpy restatement; never above Q x the largest entry
    rng = np.random.RandomState(3)
    b50 = np.asarray(blosum50()[1], dtype=np.int32).reshape(24, 24)
    for mat in (b50, rng.randint(-9, 12, size=(24, 24)).astype(np.int32)):
        for qlen in (1, 17, 400, 2000):
            q = rng.randint(0, 24, size=qlen).astype(np.uint8)
            want = int(np.maximum(mat[q].max(axis=1), 0).sum())
            assert bound(q, mat) == want <= qlen * int(mat.max())
    # a 2000-aa protein query under BLOSUM50: less than half of length x 15, which is what keeps its mid-length targets on the 16-bit kernels
    from opal_amd import synth
    assert bound(synth.query(2000, seed=7), b50) == 13181


def test_long_target_plan_of_the_scheduler(opal_opt):
    """planner.cpp::plan_pair / plan_swap on the host alone: where the LONGEST sequences of a search go (the role-swapped kernel,
    the pair kernel, or nowhere: the class body). Round 6 retuned this for the global modes after a query-length scan
    (DESIGN.md 3.10): the rules below are the measured ones."""
    import ctypes as C
    from opal_amd import synth
    L = C.CDLL(LIB)
    L.opalx_debug_plan_long_targets.restype = C.c_int
    L.opalx_debug_plan_long_targets.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_int)]
    for e in ("OPAL_HIP_SWAP", "OPAL_HIP_PAIR", "OPAL_HIP_TAIL_FRACTION", "OPAL_HIP_SWAP_ROWS"):
        opal_opt.unset(e)
    NW, OV, SW = 0, 2, 3
    cache = {}

    def plan(q, shape, n, mode=NW):
        if (shape, n) not in cache:
            lens = np.sort(synth.lengths(42, 0, n, shape))[::-1].astype(np.int32)
            cache[(shape, n)] = np.ascontiguousarray(lens[lens > 0])
        lens = cache[(shape, n)]
        need32 = int((3 * 3 + (q + lens.astype(np.int64) + 2) + 5 >= 32767).sum())  # (the negative bound of ladder.cpp, gaps 3/1)
        out = (C.c_int * 5)()
        assert L.opalx_debug_plan_long_targets(q, 24, mode, lens.ctypes.data_as(C.POINTER(C.c_int)), len(lens), int(lens.sum()), need32, out) == 0
        swapped, _, paired, _, waves = out
        assert not (swapped and paired)
        return swapped, paired, need32

    # C3 (100 000 synthetic sequences clamped at 5 000 residues, 300-aa query): the longest target alone does not outlast the
    # search -- nothing on the role-swapped kernel (forty there cost 3-4 %), a handful on the pair kernel as in round 5
    for mode in (NW, OV):
        swapped, paired, _ = plan(300, "c2", 100000, mode)
        assert swapped == 0 and 0 < paired <= 16
    # ... nor anything at all on the 541 762-sequence synthetic database, whatever the query
    for q in (110, 256, 513, 1000):
        assert plan(q, "c2", 541762)[:2] == (0, 0)
    # the published shape (tail to 35 213 residues): 160-320-aa global searches send their forty longest targets to the role-swapped
    # kernel (the model alone said 10-19: 256-aa NW 7.7 ms, with forty 5.3), and the pair kernel takes none of them
    for q in (180, 256, 300):
        swapped, paired, need32 = plan(q, "sprot", 541762)
        assert swapped == 40 and paired == 0 and need32 in (9, 10)
    assert plan(110, "sprot", 541762)[0] >= 40  # (short queries: as many as the model counts, 43)
    # from 400 residues on the longest target no longer outlasts that search: the pair kernel, as before
    for q, least in ((400, 10), (513, 10), (700, 20), (1000, 30)):
        swapped, paired, need32 = plan(q, "sprot", 541762)
        assert swapped == 0 and paired >= max(least, need32)
    # the first 100 000 / 20 000 sequences of it, and 20 000 synthetic ones: the search is shorter, the tail the same -- forty swapped to
    # 513 aa (400 aa x 100 000: 7.5 -> 5.8 ms), the pair kernel beyond 600
    for shape, n in (("sprot", 100000), ("sprot", 20000), ("c2", 20000)):
        for q in (180, 256, 400, 513):
            assert plan(q, shape, n)[:2] == (40, 0), (shape, n, q)
        swapped, paired, _ = plan(700, shape, n)
        assert swapped == 0 and paired > 0
    # more than 240 candidates (a 40-aa query: every 1 000-residue sequence outlasts its share): the forty longest, not none
    assert plan(40, "c2", 100000)[:2] == (40, 0)
    # SW keeps the round-5 rule (a 30 % modelled win, at most 240): the published shape's tail at short queries only
    assert plan(110, "sprot", 541762, SW)[0] > 40 and plan(513, "sprot", 541762, SW)[:2] == (0, 0)
    assert plan(300, "c2", 100000, SW)[:2] == (0, 0)
    # the test hook still forces a count
    opal_opt.set("OPAL_HIP_SWAP", "7")
    assert plan(300, "c2", 100000)[0] == 7


def test_swend_slot_buffer_geometry():
    """The LDS buffer in which the rows of a slot's best SW cell wait (dp_kernel.hpp, opal_swend_groups): a multiple of 16 bytes
    per slot, room for the largest group of row blocks plus two header words, never more than the whole column, and the same
    number the launch asks for per slot whatever the workgroup shape."""
    import ctypes as C
    from opal_amd import _lib
    L = _lib.lib()
    L.opal_hip_swend_ckpt_bytes.restype = C.c_size_t
    L.opal_hip_swend_ckpt_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    for S in range(1, 41):
        per_slot = L.opal_hip_swend_ckpt_bytes(64, S, 1)
        assert per_slot % 16 == 0 and per_slot >= 4 * (min(S, 4) + 2), (S, per_slot)
        blocks = (S + 3) // 4
        assert per_slot <= 4 * (4 * blocks + 2 + 3), (S, per_slot)           # at most the whole column
        if 25 <= S <= 36:
            assert per_slot <= 4 * (4 * ((blocks + 3) // 4) + 2 + 3), (S, per_slot)   # a quarter where four accumulators fit
        for G, wpb in ((16, 4), (8, 8), (1, 2)):
            assert L.opal_hip_swend_ckpt_bytes(G, S, wpb) == per_slot * (wpb * 64 // G)
    assert L.opal_hip_swend_ckpt_bytes(0, 8, 4) == 0 and L.opal_hip_swend_ckpt_bytes(16, 0, 4) == 0
