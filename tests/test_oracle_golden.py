"""Pins the oracle (oracle/) against the golden vectors captured from the reference itself
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import som_oracle as o
from tests import golden_inputs as gi


def r_atol(X):
    """absolute error bound of r = xx - 2 x.w + yy near r = 0 (summation-order noise)"""
    return 1e-12 * float(np.max(np.einsum("ij,ij->i", X, X, dtype=np.float64)))


def _epochs(g):
    return [int(e) for e in g["epochs_full"]]


@pytest.mark.parametrize("name", gi.FIT_CASES)
def test_hot_path_per_epoch(name):
    g = gi.load(name)
    X, _ = gi.case_X(name)
    for e in _epochs(g):
        W = g[f"e{e}_weights_in"]
        M = W.shape[0]
        gw, gd = g[f"e{e}_winners"], g[f"e{e}_distances"]
        for fn in (o.bmu_chain, o.bmu_blas):
            d, i = fn(X, W, 1)
            assert np.array_equal(i, gw), (name, e, fn.__name__)  # BMU indices bit-exact
            # squared distances carry the cancellation error of the expanded form: compare r
            # (float32 samples AND prototypes -- epoch 0 of a float32 fit: distances come back rounded through
            #  float32; BLAS order and chain order may round to neighbouring float32 values, 1.2e-7 apart, which
            #  doubles in the square)
            np.testing.assert_allclose(d * d, gd * gd, rtol=3e-7 if X.dtype == np.float32 and
                                       W.dtype == np.float32 else 1e-9, atol=r_atol(X))
        d, i = o.bmu_chain(X, W, 1)
        kw = o.exp_similarity(gd, X.dtype.type(g[f"e{e}_total_variance"]))
        np.testing.assert_allclose(kw, g[f"e{e}_sample_weights"], rtol=1e-13, atol=1e-15)
        kw = g[f"e{e}_sample_weights"]
        S, K, a, E = o.accumulate(X, gw, kw, gd, M)
        assert np.array_equal(a, g[f"e{e}_activations"])
        if f"e{e}_errors" in g:
            np.testing.assert_allclose(E, g[f"e{e}_errors"], rtol=1e-12)
        C = o.voronoi_centers(S, K, a, "compact")  # quirk Q1
        np.testing.assert_allclose(C, g[f"e{e}_centers_compact"], rtol=1e-11, atol=1e-12)
        S2, K2, a2, E2 = o.accumulate_numpy(X, gw, kw, gd, M)
        np.testing.assert_allclose(S2, S, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(K2, K, rtol=1e-12)
        h = o.gaussian_neighborhood(g[f"e{e}_hop_distance"], float(g[f"e{e}_sigma"]))
        for sm in (o.smooth_matmul, o.smooth_broadcast):
            Wn = sm(h, a, C)
            np.testing.assert_allclose(Wn, g[f"e{e}_weights_out"], rtol=1e-10, atol=1e-11,
                                       equal_nan=True)
        ct = o.change_total(W, g[f"e{e}_weights_out"])
        assert (ct < 1e-5) == bool(g[f"e{e}_converged"]) or bool(g[f"e{e}_converged"])


def test_reference_splits_exact_ties_between_duplicate_prototypes():
    """grow_dup_f64 (tools/make_golden.py): at epoch 43 of this fit of the reference two dead neurons at the same hop
    distance from every live one leave the smoothing (BaseSom.py:509-515) as bit-identical prototype rows; at epoch 44 every sample
    nearest to them is an exact tie, and the reference's BLAS gives some of them to the one and some to the other
    (the summation order of a GEMM column depends on where it lies in the blocking).  The build's rule -- the
    lowest index wins -- differs from the recorded winners ONLY between these bit-identical rows; everything else of
    the epoch, and every epoch before it, is the reference's."""
    g = gi.load(gi.DUP_CASE)
    X, _ = gi.case_X(gi.DUP_CASE)
    e, (a, b) = gi.DUP_EPOCH, gi.DUP_ROWS
    # how the duplicates come about: both dead at epoch 43, and their hop distances agree wherever a neuron is ALIVE
    # (they differ towards each other and one more dead neuron): the rows of h * a are the same numbers
    hop = g["e43_hop_distance"]
    alive = g["e43_activations"] > 0
    assert not alive[a] and not alive[b] and not np.array_equal(hop[a], hop[b])
    assert np.array_equal(hop[a, alive], hop[b, alive])
    assert np.array_equal(g["e43_weights_out"][a], g["e43_weights_out"][b])
    oo = o.epoch(X, g["e43_weights_in"], hop, float(g["e43_sigma"]), g["e43_total_variance"], "compact", "chain")
    assert np.array_equal(oo.winners, g["e43_winners"])
    assert np.array_equal(oo.new_weights[a], oo.new_weights[b])          # the oracle makes the same duplicates
    W = g[f"e{e}_weights_in"]
    assert np.array_equal(W[a], W[b])
    gw, gd = g[f"e{e}_winners"], g[f"e{e}_distances"]
    for fn in (o.bmu_chain, o.bmu_blas):
        d, i = fn(X, W, 1)
        np.testing.assert_allclose(d * d, gd * gd, rtol=1e-9, atol=r_atol(X))
        # identical once the two names of the one prototype are merged
        assert np.array_equal(np.where(i == b, a, i), np.where(gw == b, a, gw)), fn.__name__
    d, i = o.bmu_chain(X, W, 1)
    tied = i == a
    assert not (i == b).any() and tied.sum() > 100                       # fixed order: lowest index, always
    assert (gw[tied] == a).any() and (gw[tied] == b).any()               # the reference: both, by BLAS rounding
    assert np.array_equal(i[~tied], gw[~tied])


def test_q1_compaction_differs_from_aligned():
    g = gi.load("blobs_dead")
    X, _ = gi.case_X("blobs_dead")
    hit = False
    for e in _epochs(g):
        a = g[f"e{e}_activations"]
        if (a == 0).any() and (a[np.argmax(a == 0):] > 0).any():
            S, K, a2, _ = o.accumulate(X, g[f"e{e}_winners"], g[f"e{e}_sample_weights"],
                                       g[f"e{e}_distances"], a.size)
            Cc = o.voronoi_centers(S, K, a2, "compact")
            Ca = o.voronoi_centers(S, K, a2, "aligned")
            assert not np.allclose(Cc, Ca)
            np.testing.assert_allclose(Cc, g[f"e{e}_centers_compact"], rtol=1e-11, atol=1e-12)
            hit = True
    assert hit, "fixture must contain a dead neuron below a live one"


@pytest.mark.parametrize("name", gi.FIT_CASES)
def test_bmu_k2_on_fitted_map(name):
    g = gi.load(name)
    X, _ = gi.case_X(name)
    W = g["final_weights"]
    for fn in (o.bmu_chain, o.bmu_blas):
        d, i = fn(X, W, 2)
        gd, gidx = g["final_bmu2_dist"], g["final_bmu2_idx"]
        np.testing.assert_allclose(d * d, gd * gd, rtol=1e-9, atol=r_atol(X))
        same = (i == gidx).all(axis=1)
        # rows that differ may only be exact ties between the two reported neighbours
        bad = ~same
        if bad.any():
            assert np.array_equal(np.sort(i[bad], axis=1), np.sort(gidx[bad], axis=1)) or \
                np.allclose(d[bad, 0], d[bad, 1], rtol=1e-12)
    d1, i1 = o.bmu_chain(X, W, 1)
    assert np.array_equal(i1, g["final_labels"]) or name in gi.CLF_CASES


@pytest.mark.parametrize("name", gi.FROZEN_CASES)
def test_frozen_epoch(name):
    g = gi.load(name)
    X, _ = gi.case_X(name)
    W, rows, cols = gi.frozen_W(name, X)
    hop = gi.lattice_hops(rows, cols)
    out = o.epoch(X, W, hop, float(g["sigma"]), X.dtype.type(g["total_variance"]), "compact", "chain")
    assert np.array_equal(out.winners, g["winners"])
    # W rows are rows of X: those samples sit at r ~ 0 where the expanded form is pure
    # summation-order noise (|r| <~ 1e-12 |x|^2), in the reference as much as here
    np.testing.assert_allclose(out.distances ** 2, g["distances"] ** 2, rtol=1e-10,
                               atol=r_atol(X))
    np.testing.assert_allclose(out.sample_weights, g["sample_weights"], rtol=1e-9, atol=1e-7)
    assert np.array_equal(out.activations, g["activations"])
    np.testing.assert_allclose(out.errors, g["errors"], rtol=1e-9, atol=5e-6)
    np.testing.assert_allclose(out.new_weights.sum(axis=1), g["weights_out_sum_rows"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(out.new_weights[:8], g["weights_out_head"], rtol=1e-8, atol=1e-10)
    if "weights_out" in g:
        # north_star tolerance for prototype weights is 1e-5 rel; the f64 oracle does far better
        np.testing.assert_allclose(out.new_weights, g["weights_out"], rtol=1e-8, atol=1e-10)
    d2, i2 = o.bmu_chain(X, W, 2)
    assert np.array_equal(i2, g["bmu2_idx"])
    np.testing.assert_allclose(d2 ** 2, g["bmu2_dist"] ** 2, rtol=1e-10, atol=r_atol(X))


def test_chain_matches_blas_and_sklearn_on_random():
    rng = np.random.default_rng(0)
    for (N, d, M, dt) in [(500, 17, 9, np.float32), (300, 64, 40, np.float64), (64, 5, 4, np.float32)]:
        X = rng.normal(size=(N, d)).astype(dt)
        W = rng.normal(size=(M, d))
        d1, i1 = o.bmu_chain(X, W, 1)
        d2, i2 = o.bmu_blas(X, W, 1)
        d3, i3 = o.bmu_sklearn(X, W, 1)
        assert np.array_equal(i1, i2) and np.array_equal(i1, i3)
        np.testing.assert_allclose(d1, d2, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(d1, d3, rtol=1e-6 if dt == np.float32 and d <= 15 else 1e-11,
                                   atol=1e-6 if d <= 15 else 1e-12)


def test_ties_resolve_to_lowest_index():
    X = np.array([[1.0, 2.0, 3.0], [0.0, 0.0, 0.0]], dtype=np.float64)
    W = np.array([[5.0, 5.0, 5.0], [1.0, 2.0, 3.0], [1.0, 2.0, 3.0], [0.0, 0.0, 0.0]])
    d, i = o.bmu_chain(X, W, 2)
    assert i.tolist() == [[1, 2], [3, 1]]
    assert d[0].tolist() == [0.0, 0.0]
    _, ib = o.bmu_blas(X, W, 2)
    assert ib.tolist() == i.tolist()
