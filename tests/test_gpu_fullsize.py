"""BASELINE.json-sized runs on the GPU, checked through size-independent properties and a sampled
oracle comparison (the oracle cannot finish 1e10 cells in seconds)."""
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
    pdb, q, m = c2["pdb"], c2["q"], c2["m"]
    out = {}
    for mode in ("SW", "OV", "HW", "NW"):
        rc, s, eq, et = pdb.search(q, 11, 1, m, 1, oracle_lib.MODES[mode])
        assert rc == 0
        out[mode] = (s, eq, et)
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
