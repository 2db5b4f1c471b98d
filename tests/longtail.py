"""Long-tail targets for the parity tests: real protein databases have 10 000 - 35 000 residue tails (the reference
was published on Swiss-Prot, reference aligner_comparison.md:6; its own random targets stop at 4 799, src/test.cpp:47-54),
the synthetic C2-C5 generator clamps at 5 000. These 41 planted sequences take the scheduler (class
boundaries, stream plans, skew bounds, the 16 -> 32 bit prefix of NW/HW/OV) and the traceback (flag budget, box sizes)
where that distribution never does.

  12 x 8 000, 12 x 12 000, 10 x 22 000, 6 x 35 213 residues from the synthetic background distribution
  (seed 4242), and one 35 000-residue concatenation of mutated copies of the query (every 17th residue replaced),
  the only target with high-scoring diagonals over its whole length.

Shared by tests/golden/make_longtail.py (which records the reference's answers) and tests/test_gpu_longtail.py.
"""
import numpy as np

from opal_amd import synth

PLANT_LENGTHS = [8000] * 12 + [12000] * 12 + [22000] * 10 + [35213] * 6
QUERY_LENGTHS = (110, 513, 2000)
GAP_OPEN, GAP_EXT = 11, 1
SEED = 4242


def planted(query):
    """-> list of 41 uint8 arrays (alphabet indices, BLOSUM50 order) for this query."""
    out = []
    start = 0
    for L in PLANT_LENGTHS:
        out.append(synth._residues(SEED, start, L))
        start += L
    reps = (35000 + len(query) - 1) // len(query)
    cat = np.tile(np.asarray(query, dtype=np.uint8), reps)[:35000].copy()
    noise = synth._residues(SEED + 1, 0, len(cat))
    cat[::17] = noise[::17]
    out.append(cat)
    return out


def query(qlen):
    return synth.query(qlen, seed=7)
