"""Score matrices for the harness side (bench, tests, CLI).

BLOSUM50 over the 24-letter alphabet ``ARNDCQEGHILKMFPSTWYVBZX*`` -- the matrix
and letter order the reference's CLI uses by default
(reference src/ScoreMatrix.cpp:51-53, 57-85). The values are the published
BLOSUM50 table, kept here as its lower triangle (the matrix is symmetric).
``*`` is the catch-all letter: every character not in the alphabet maps to it
(reference src/opal_aligner.cpp:250-258).
"""
import numpy as np

BLOSUM50_ALPHABET = "ARNDCQEGHILKMFPSTWYVBZX*"

_BLOSUM50_LOWER = """
5
-2 7
-1 -1 7
-2 -2 2 8
-1 -4 -2 -4 13
-1 1 0 0 -3 7
-1 0 0 2 -3 2 6
0 -3 0 -1 -3 -2 -3 8
-2 0 1 -1 -3 1 0 -2 10
-1 -4 -3 -4 -2 -3 -4 -4 -4 5
-2 -3 -4 -4 -2 -2 -3 -4 -3 2 5
-1 3 0 -1 -3 2 1 -2 0 -3 -3 6
-1 -2 -2 -4 -2 0 -2 -3 -1 2 3 -2 7
-3 -3 -4 -5 -2 -4 -3 -4 -1 0 1 -4 0 8
-1 -3 -2 -1 -4 -1 -1 -2 -2 -3 -4 -1 -3 -4 10
1 -1 1 0 -1 0 -1 0 -1 -3 -3 0 -2 -3 -1 5
0 -1 0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1 2 5
-3 -3 -4 -5 -5 -1 -3 -3 -3 -3 -2 -3 -1 1 -4 -4 -3 15
-2 -1 -2 -3 -3 -1 -2 -3 2 -1 -1 -2 0 4 -3 -2 -2 2 8
0 -3 -3 -4 -1 -3 -3 -4 -4 4 1 -3 1 -1 -3 -2 0 -3 -1 5
-2 -1 5 6 -3 0 1 -1 0 -4 -4 0 -3 -4 -2 0 0 -5 -3 -3 6
-1 0 0 1 -3 4 5 -2 0 -3 -3 1 -1 -4 -1 0 -1 -2 -2 -3 1 5
-1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1
-5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 1
"""


def blosum50():
    """Return (alphabet:str, matrix:int32[24,24]); matrix[q, t] scores query letter q vs target letter t."""
    rows = [[int(x) for x in line.split()] for line in _BLOSUM50_LOWER.strip().splitlines()]
    n = len(rows)
    m = np.zeros((n, n), dtype=np.int32)
    for i, row in enumerate(rows):
        assert len(row) == i + 1
        for j, v in enumerate(row):
            m[i, j] = m[j, i] = v
    return BLOSUM50_ALPHABET, m


def simple_matrix(alphabet_length, match, mismatch):
    """match on the diagonal, mismatch elsewhere (reference src/test.cpp:190-197)."""
    m = np.full((alphabet_length, alphabet_length), mismatch, dtype=np.int32)
    np.fill_diagonal(m, match)
    return m


def letter_index(alphabet):
    """256-entry letter -> alphabet index table; unknown letters go to '*' when the alphabet has one
    (reference src/opal_aligner.cpp:250-258), else to 0."""
    idx = np.zeros(256, dtype=np.uint8)
    if "*" in alphabet:
        idx[:] = alphabet.index("*")
    for i, ch in enumerate(alphabet):
        idx[ord(ch)] = i
    return idx
