"""Parity test cases shared by tests/golden/make_golden.py (which records the reference's
outputs for them) and the test modules (which check the oracle and the HIP path against
those records). A case is a plain description; `materialise` turns it into arrays.

Input kinds:
  fasta      query / db are files under tests/golden/data (the reference's bundled test_data)
  synth      opal_amd.synth protein generator: seeds 42 (db) / 7 (query)
  random4    test.cpp-style uniform 4-letter sequences (reference src/test.cpp:37-54), drawn
             from splitmix64 instead of libc rand
  literal    sequences written out in the case
  randomA    uniform random sequences over an alphabet of A letters with a seeded random symmetric matrix
             (entries -6..8, diagonal 2..10), drawn from splitmix64: alphabets of 2 .. 255 letters
"""
import os

import numpy as np

from opal_amd import synth
from opal_amd.fasta import read_fasta
from opal_amd.score_matrix import blosum50, simple_matrix

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")

SCORE, SCORE_END = 0, 1
SIMPLE, BUCKETS = 0, 1
ALL_MODES = ["NW", "HW", "OV", "SW"]

CASES = {}


def _add(name, **kw):
    kw.setdefault("modes", ALL_MODES)
    kw.setdefault("search_type", SCORE_END)
    kw.setdefault("gap_open", 3)
    kw.setdefault("gap_ext", 1)
    CASES[name] = kw


# ---- the reference's bundled test_data (BLOSUM50, CLI default gaps 3/1 and the no-int8-overflow 11/1)
_QUERIES = ["O74807", "P19930", "B8E1A7", "Q3ZAI3", "P18080", "test1"]
for q in _QUERIES:
    for go in (3, 11):
        _add(f"td_{q}_sprot15_o{go}", kind="fasta", query=f"query_{q}.fasta", db="uniprot_sprot15.fasta",
             matrix="blosum50", gap_open=go)
    _add(f"td_{q}_sprot196_o3", kind="fasta", query=f"query_{q}.fasta", db="uniprot_sprot196.fasta",
         matrix="blosum50")
_add("td_test1_test1", kind="fasta", query="query_test1.fasta", db="test1.fasta", matrix="blosum50")
# BASELINE.json configs[0]: O74807 x uniprot_sprot15, SW score-only, OPAL_OVERFLOW_BUCKETS
_add("c1_O74807_sprot15_sw_score", kind="fasta", query="query_O74807.fasta", db="uniprot_sprot15.fasta",
     matrix="blosum50", modes=["SW"], search_type=SCORE, overflow=BUCKETS, char_sw=True)
_add("c1_O74807_sprot15_o11_char", kind="fasta", query="query_O74807.fasta", db="uniprot_sprot15.fasta",
     matrix="blosum50", modes=["SW"], search_type=SCORE, gap_open=11, char_sw=True)

# ---- README example (reference README.md:34-55)
_add("readme_example", kind="literal", alphabet=4,
     matrix=[2, -1, -3, 0, -1, 4, -5, -1, -3, -5, 1, -10, 0, -1, -10, 4],
     query=[0, 1, 3, 2, 1, 0, 3, 0, 1, 1],
     db=[[1, 3, 2, 3, 0, 0, 1, 0, 2, 2, 1, 2, 3, 2], [2, 1, 1, 3, 2, 0, 0, 2, 2, 0, 2, 1],
         [0, 0, 2, 1, 0, 3, 1, 1, 2, 3, 2, 1, 0], [2, 3, 3, 3, 1, 1, 2, 2, 0]])

# ---- escalation pins: 8 -> 16 bit, and 16 -> 32 bit (SW only: the reference's
#      32-bit NW/HW/OV path is undefined behaviour)
_add("pin_100_same_match3", kind="literal", alphabet=4, matrix=("simple", 3, -1),
     query=[1] * 100, db=[[1] * 100, [1] * 50 + [2] * 50, [2] * 100])
_add("pin_2000_same_match30_int32", kind="literal", alphabet=4, matrix=("simple", 30, -1),
     query=[1] * 2000, db=[[1] * 2000, [1] * 1500, [0] * 30 + [1] * 1200], modes=["SW"], gap_open=11)

# ---- synthetic protein sets; npz storage
for go in (3, 11):
    _add(f"synth_q300_db2000_o{go}", kind="synth", qlen=300, nseq=2000, first=0, matrix="blosum50",
         gap_open=go, store="npz")
_add("synth_q513_db600_o3", kind="synth", qlen=513, nseq=600, first=5000, matrix="blosum50", store="npz")
_add("synth_q2000_db96_o3", kind="synth", qlen=2000, nseq=96, first=9000, matrix="blosum50", store="npz")
_add("synth_q37_db700_o3", kind="synth", qlen=37, nseq=700, first=20000, matrix="blosum50", store="npz")

# ---- test.cpp-style random 4-letter sets: scores 400-600 force the 8 -> 16 bit escalation in every mode
_add("random4_q1000_db24", kind="random4", qlen=1000, nseq=24, seed=1001, matrix=("simple", 3, -1),
     alphabet=4, gap_open=11, overflow=SIMPLE)
_add("random4_q200_db300_o0", kind="random4", qlen=200, nseq=300, seed=1002, matrix=("simple", 2, -3),
     alphabet=4, gap_open=5, gap_ext=2, minlen=1, maxlen=400)


# ---- gap penalties at the edges the API allows (opal.h:99-111: non-negative): free gaps, linear gaps (open == extend), an
#      opening cheaper than an extension, free extensions. (The reference's ALIGNMENT pass crashes for some of these; its
#      score / end-location search is defined for all of them.)
for _go, _ge in ((0, 0), (0, 1), (1, 0), (2, 2), (1, 3), (5, 0), (7, 7)):
    _add(f"gaps_o{_go}_e{_ge}", kind="random4", qlen=90, nseq=40, seed=2001, matrix=("simple", 3, -2), alphabet=4,
         gap_open=_go, gap_ext=_ge, minlen=20, maxlen=300)

# ---- the smallest shapes: queries of 1-3 letters against targets of 1, 2 and 5 letters
for _q in (1, 2, 3):
    for _l in (1, 2, 5):
        _add(f"tiny_q{_q}_l{_l}", kind="random4", qlen=_q, nseq=6, seed=3000 + 10 * _q + _l, matrix=("simple", 3, -2), alphabet=4,
             minlen=_l, maxlen=_l)

# ---- alphabets from 2 to 255 letters (the API takes any unsigned char alphabet; the query profile of a large one no
#      longer fits the LDS in one piece: row chunks)
for _a in (2, 64, 200, 255):
    _add(f"alphabet_{_a}", kind="randomA", alphabet=_a, qlen=150, nseq=30, seed=4000 + _a, minlen=30, maxlen=400, gap_open=5, gap_ext=2)

# ---- alignment (traceback) goldens: start locations and operation strings recorded from the reference
for _n in ("readme_example", "td_test1_test1", "td_O74807_sprot15_o3", "td_O74807_sprot15_o11", "pin_100_same_match3"):
    CASES[_n]["align"] = True
_add("random4_q120_db40_align", kind="random4", qlen=120, nseq=40, seed=1003, matrix=("simple", 3, -2),
     alphabet=4, gap_open=4, gap_ext=1, minlen=20, maxlen=260, align=True)
_add("synth_q90_db60_align", kind="synth", qlen=90, nseq=60, first=30000, matrix="blosum50", align=True)


def _matrix(spec, alphabet_length=None):
    if spec == "blosum50":
        return blosum50()[1]
    if isinstance(spec, tuple) and spec[0] == "simple":
        return simple_matrix(alphabet_length, spec[1], spec[2])
    m = np.array(spec, dtype=np.int32)
    n = int(round(len(m) ** 0.5))
    return m.reshape(n, n)


def random4(seed, qlen, nseq, alphabet=4, minlen=800, maxlen=4799):
    """Uniform random sequences over `alphabet` letters; lengths uniform in [minlen, maxlen]."""
    q = (synth.splitmix64(seed, 0, qlen) % np.uint64(alphabet)).astype(np.uint8)
    lens = (minlen + synth.splitmix64(seed + 1, 0, nseq) % np.uint64(maxlen - minlen + 1)).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])
    flat = (synth.splitmix64(seed + 2, 0, int(offs[-1])) % np.uint64(alphabet)).astype(np.uint8)
    return q, [flat[offs[i]:offs[i + 1]] for i in range(nseq)]


def random_alphabet(seed, alphabet, qlen, nseq, minlen, maxlen):
    """Uniform random sequences over `alphabet` letters and a random symmetric matrix (entries -6..8, diagonal 2..10)."""
    A = alphabet
    r = synth.splitmix64(seed, 0, A * A + A)
    m = (r[: A * A] % np.uint64(15)).astype(np.int64).reshape(A, A) - 6
    m = np.minimum(m, m.T)
    m[np.arange(A), np.arange(A)] = 2 + (r[A * A:] % np.uint64(9)).astype(np.int64)
    q = (synth.splitmix64(seed + 1, 0, qlen) % np.uint64(A)).astype(np.uint8)
    lens = (minlen + synth.splitmix64(seed + 2, 0, nseq) % np.uint64(maxlen - minlen + 1)).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])
    flat = (synth.splitmix64(seed + 3, 0, int(offs[-1])) % np.uint64(A)).astype(np.uint8)
    return q, [flat[offs[i]:offs[i + 1]] for i in range(nseq)], m.astype(np.int32)


def materialise(case):
    """-> (query uint8[], list of uint8[] targets, int32[A,A] matrix)."""
    kind = case["kind"]
    if kind == "fasta":
        alphabet = blosum50()[0]
        q = read_fasta(os.path.join(DATA, case["query"]), alphabet)[0]
        db = read_fasta(os.path.join(DATA, case["db"]), alphabet)
        return q, db, _matrix(case["matrix"])
    if kind == "synth":
        res, offs, lens = synth.database(case["nseq"], seed=42, first=case["first"])
        db = [res[offs[i]:offs[i + 1]] for i in range(case["nseq"])]
        return synth.query(case["qlen"], seed=7), db, _matrix(case["matrix"])
    if kind == "random4":
        q, db = random4(case["seed"], case["qlen"], case["nseq"], case["alphabet"],
                        case.get("minlen", 800), case.get("maxlen", 4799))
        return q, db, _matrix(case["matrix"], case["alphabet"])
    if kind == "randomA":
        return random_alphabet(case["seed"], case["alphabet"], case["qlen"], case["nseq"], case["minlen"], case["maxlen"])
    if kind == "literal":
        q = np.array(case["query"], dtype=np.uint8)
        db = [np.array(s, dtype=np.uint8) for s in case["db"]]
        return q, db, _matrix(case["matrix"], case["alphabet"])
    raise ValueError(kind)
