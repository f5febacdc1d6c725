"""GPU parity tests: the HIP path (through the C ABI) against the oracle on identical inputs.
Run on the MI355X box with `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest

from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from dbgsom_amd.backend import HipBackend

    return HipBackend()


@pytest.fixture(scope="module")
def o():
    from oracle import som_oracle

    return som_oracle


def _rand(N, d, M, dt, seed):
    rng = np.random.default_rng(seed)
    X = (rng.normal(size=(N, d)) * rng.uniform(0.5, 3.0, size=d)).astype(dt)
    W = rng.normal(size=(M, d)) * 1.5
    return X, W


SHAPES = [
    # N, d, M, dtype
    (1, 1, 1, np.float32),
    (5, 3, 2, np.float64),
    (127, 15, 7, np.float32),
    (128, 16, 128, np.float32),
    (129, 17, 129, np.float64),
    (1000, 64, 25, np.float64),
    (777, 100, 300, np.float32),
    (2049, 130, 257, np.float32),
    (513, 784, 140, np.float32),
    (300, 33, 1030, np.float64),
    # small maps on the LDS-DMA kernel: 32- and 64-wide prototype chunks
    (1500, 64, 4, np.float32),
    (1500, 64, 31, np.float32),
    (1000, 784, 33, np.float32),
    (1000, 48, 64, np.float32),
    (700, 128, 65, np.float32),
]


@pytest.mark.parametrize("N,d,M,dt", SHAPES)
@pytest.mark.parametrize("k", [1, 2])
def test_bmu_bit_exact(hip, o, N, d, M, dt, k):
    if M < k:
        pytest.skip("k > M")
    X, W = _rand(N, d, M, dt, seed=N * 31 + d)
    hip.load(X)
    dist, idx = hip.bmu(W, k)
    rd, ri = o.bmu_chain(X, W, k)
    assert np.array_equal(idx, ri)          # BMU indices: bit-exact
    assert np.array_equal(dist, rd)         # same fma chain + correctly rounded sqrt: bit-exact
    # and against the BLAS statement (summation order differs): indices equal, distances close
    bd, bi = o.bmu_blas(X, W, k)
    assert np.array_equal(idx, bi)
    np.testing.assert_allclose(dist ** 2, bd ** 2, rtol=1e-10, atol=1e-9)


def test_bmu_query_path_and_f32_pair_rounding(hip, o):
    X, W = _rand(400, 20, 9, np.float32, 5)
    hip.load(X)
    W32 = W.astype(np.float32)
    dist, idx = hip.bmu(W32, 1)
    rd, ri = o.bmu_chain(X, W32, 1)
    assert np.array_equal(idx, ri) and np.array_equal(dist, rd)
    assert np.array_equal(dist, dist.astype(np.float32).astype(np.float64))
    Xq, _ = _rand(77, 20, 9, np.float64, 6)
    dq, iq = hip.bmu(W, 2, X=Xq)
    rq, rqi = o.bmu_chain(Xq, W, 2)
    assert np.array_equal(iq, rqi) and np.array_equal(dq, rq)


def test_ties_lowest_index(hip, o):
    X = np.array([[1.0, 2.0, 3.0], [0.0, 0.0, 0.0], [2.0, 2.0, 2.0]], dtype=np.float64)
    W = np.array([[5.0, 5.0, 5.0], [1.0, 2.0, 3.0], [1.0, 2.0, 3.0], [0.0, 0.0, 0.0]])
    hip.load(X)
    d, i = hip.bmu(W, 2)
    assert i[:2].tolist() == [[1, 2], [3, 1]]
    assert d[0].tolist() == [0.0, 0.0]
    # many duplicated prototypes spread over tiles / lanes / wavefronts
    rng = np.random.default_rng(3)
    base = rng.integers(0, 5, size=(6, 12)).astype(np.float64)
    W2 = np.tile(base, (60, 1))  # 360 prototypes, each row repeated 60 times
    X2 = rng.integers(0, 5, size=(500, 12)).astype(np.float64)
    hip.load(X2)
    d2, i2 = hip.bmu(W2, 2)
    assert (i2[:, 0] < 6).all()                 # lowest copy wins
    rd, ri = o.bmu_chain(X2, W2, 2)             # exact integer arithmetic: every tie is a true tie
    assert np.array_equal(i2, ri) and np.array_equal(d2, rd)
    assert (d2[:, 0] == d2[:, 1]).all() and (i2[:, 1] > i2[:, 0]).all()


@pytest.mark.parametrize("name", gi.FIT_CASES)
def test_golden_epochs(hip, o, name):
    """Every recorded epoch of the reference's own fits: winners bit-exact, the rest within
    float64 noise (north_star: weights within 1e-5 rel)."""
    g = gi.load(name)
    X, _ = gi.case_X(name)
    hip.load(X)
    for e in [int(v) for v in g["epochs_full"]]:
        W = g[f"e{e}_weights_in"]
        hop = g[f"e{e}_hop_distance"]
        gamma = float(X.dtype.type(g[f"e{e}_total_variance"]) ** -1)
        res = hip.epoch(W, hop, float(g[f"e{e}_sigma"]), gamma, "compact", True)
        assert np.array_equal(res.winners, g[f"e{e}_winners"]), (name, e)
        xn = float(np.max(np.einsum("ij,ij->i", X, X, dtype=np.float64)))
        np.testing.assert_allclose(res.distances ** 2, g[f"e{e}_distances"] ** 2,
                                   rtol=3e-7 if W.dtype == np.float32 else 1e-9, atol=1e-12 * xn)
        assert np.array_equal(res.activations, g[f"e{e}_activations"])
        if f"e{e}_errors" in g:
            np.testing.assert_allclose(res.errors, g[f"e{e}_errors"], rtol=1e-9, atol=1e-6)
        # (float32 prototypes -- epoch 0 of a float32 fit: the reference returns distances rounded through float32,
        #  and of 20 000 samples a few sit on a rounding boundary that BLAS order and chain order resolve
        #  differently; their sample weights move by 1e-7 and the 4 start prototypes by 1.5e-8.  north_star: 1e-5)
        f32w = W.dtype == np.float32
        np.testing.assert_allclose(res.new_weights, g[f"e{e}_weights_out"], rtol=1e-6 if f32w else 1e-7,
                                   atol=1e-7 if f32w else 1e-9, equal_nan=True)
        # against the oracle on the same inputs: tight
        oo = o.epoch(X, W, hop, float(g[f"e{e}_sigma"]), X.dtype.type(g[f"e{e}_total_variance"]),
                     "compact", "chain")
        assert np.array_equal(res.winners, oo.winners)
        assert np.array_equal(res.distances, oo.distances)
        np.testing.assert_allclose(res.new_weights, oo.new_weights, rtol=1e-11, atol=1e-12,
                                   equal_nan=True)
        np.testing.assert_allclose(res.errors, oo.errors, rtol=1e-12)
        np.testing.assert_allclose(res.change_total, oo.change_total, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("algo", ["exact", "filtered", "prune"])
def test_reference_tie_between_duplicate_prototypes(o, algo):
    """grow_dup_f64, epoch 44 (see tests/test_oracle_golden.py): two bit-identical prototype rows (175, 194), 389
    samples nearest to them.  Every form of the search gives them to the LOWER index, with the chain's distances;
    the reference's BLAS splits them between the two.  Everything else equals the recorded epoch."""
    from dbgsom_amd.backend import HipBackend

    g = gi.load(gi.DUP_CASE)
    X, _ = gi.case_X(gi.DUP_CASE)
    e, (a, b) = gi.DUP_EPOCH, gi.DUP_ROWS
    W, hop = g[f"e{e}_weights_in"], g[f"e{e}_hop_distance"]
    assert np.array_equal(W[a], W[b]) and W.shape[0] >= HipBackend.FILTER_MIN_PROTOTYPES
    be = HipBackend(0, algorithm="filtered" if algo == "prune" else algo).load(X)
    if algo == "prune":
        be.sweep_planes = 4
    gamma = float(g[f"e{e}_total_variance"] ** -1)
    res = be.epoch(W, hop, float(g[f"e{e}_sigma"]), gamma, "compact", True)
    if algo != "exact":
        assert be.filter_log[-1][0] == "filtered"
    gw = g[f"e{e}_winners"]
    assert not (res.winners == b).any()
    assert np.array_equal(res.winners, np.where(gw == b, a, gw))
    oo = o.epoch(X, W, hop, float(g[f"e{e}_sigma"]), g[f"e{e}_total_variance"], "compact", "chain")
    assert np.array_equal(res.winners, oo.winners) and np.array_equal(res.distances, oo.distances)
    np.testing.assert_allclose(res.new_weights, oo.new_weights, rtol=1e-11, atol=1e-12, equal_nan=True)
    d2, i2 = be.bmu(W, 2)
    tied = res.winners == a
    assert (i2[tied] == [a, b]).all() and (d2[tied, 0] == d2[tied, 1]).all()
    # the epochs before the tie are the reference's, winners bit for bit
    for e0 in (0, 30, 43):
        r0 = be.epoch(g[f"e{e0}_weights_in"], g[f"e{e0}_hop_distance"], float(g[f"e{e0}_sigma"]), gamma, "compact", True)
        assert np.array_equal(r0.winners, g[f"e{e0}_winners"]), e0
        np.testing.assert_allclose(r0.new_weights, g[f"e{e0}_weights_out"], rtol=1e-7, atol=1e-9, equal_nan=True)
    be.release()


@pytest.mark.parametrize("name", gi.FROZEN_CASES)
def test_golden_frozen(hip, o, name):
    g = gi.load(name)
    X, _ = gi.case_X(name)
    W, rows, cols = gi.frozen_W(name, X)
    hop = gi.lattice_hops(rows, cols)
    hip.load(X)
    gamma = float(X.dtype.type(g["total_variance"]) ** -1)
    res = hip.epoch(W, hop, float(g["sigma"]), gamma, "compact", True)
    assert np.array_equal(res.winners, g["winners"])
    assert np.array_equal(res.activations, g["activations"])
    np.testing.assert_allclose(res.errors, g["errors"], rtol=1e-9, atol=5e-6)
    np.testing.assert_allclose(res.new_weights[:8], g["weights_out_head"], rtol=1e-5)  # north_star
    np.testing.assert_allclose(res.new_weights[:8], g["weights_out_head"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(res.new_weights.sum(axis=1), g["weights_out_sum_rows"], rtol=1e-7,
                               atol=1e-8)
    d2, i2 = hip.bmu(W, 2)
    assert np.array_equal(i2, g["bmu2_idx"])


def test_accumulate_and_smooth_vs_oracle_with_dead_neurons(hip, o):
    rng = np.random.default_rng(9)
    N, d, M = 5000, 37, 50
    X = rng.normal(size=(N, d)).astype(np.float32)
    W = rng.normal(size=(M, d))
    W[[3, 4, 17, 49]] += 50.0  # never win -> dead neurons (quirk Q1 territory)
    hip.load(X)
    dist, win = hip.bmu(W, 1)
    gamma = 1.0 / float(np.var(X, axis=0).sum())
    kw = hip.exp_similarity(dist, gamma)
    np.testing.assert_allclose(kw, o.exp_similarity_gamma(dist, gamma), rtol=1e-13, atol=1e-16)
    rows, cols = 5, 10
    hop = gi.lattice_hops(rows, cols)
    for layout in ("compact", "aligned"):
        Wn, chg, E, a = hip.update(W, hop, 1.3, kw, win, dist, layout)
        S, K, a_o, E_o = o.accumulate(X, win, kw, dist, M)
        C = o.voronoi_centers(S, K, a_o, layout)
        Wo = o.smooth_matmul(o.gaussian_neighborhood(hop, 1.3), a_o, C)
        assert (a_o == 0).sum() >= 4
        assert np.array_equal(a, a_o)
        np.testing.assert_allclose(E, E_o, rtol=1e-12)
        np.testing.assert_allclose(Wn, Wo, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(chg, o.change_total(W, Wo), rtol=1e-10)


def test_disconnected_lattice_and_nan_rows(hip, o):
    """inf hop distance -> h = 0; a row whose every h*a underflows -> 0/0 = NaN like the reference"""
    rng = np.random.default_rng(2)
    X = rng.normal(size=(300, 6))
    W = rng.normal(size=(4, 6))
    W[3] += 100.0  # dead
    hop = np.array([[0, 1, np.inf, np.inf], [1, 0, np.inf, np.inf],
                    [np.inf, np.inf, 0, 1], [np.inf, np.inf, 1, 0]], dtype=np.float64)
    hip.load(X)
    res = hip.epoch(W, hop, 0.8, 0.1, "aligned", True)
    oo = o.epoch(X, W, hop, 0.8, np.float64(10.0), "aligned", "chain")
    np.testing.assert_allclose(res.new_weights, oo.new_weights, rtol=1e-11, equal_nan=True)


def test_determinism_bitwise(hip):
    X, W = _rand(20000, 64, 100, np.float32, 42)
    hip.load(X)
    hop = gi.lattice_hops(10, 10)
    r1 = hip.epoch(W, hop, 2.0, 0.01, "compact", True)
    r2 = hip.epoch(W, hop, 2.0, 0.01, "compact", True)
    assert np.array_equal(r1.new_weights, r2.new_weights)  # no float atomics anywhere
    assert np.array_equal(r1.errors, r2.errors) and r1.change_total == r2.change_total


@pytest.mark.parametrize("N,d,rows,cols", [(3000, 24, 4, 5), (9000, 50, 12, 13), (7000, 100, 16, 17)])
def test_ctx_api_numpy_only(o, N, d, rows, cols):
    """The context-level ABI a NumPy caller (the reference) would bind: host pointers only, no
    torch.  The larger maps (M >= 129) go through the filtered search behind the same calls, the
    feature counts are not multiples of 16 (padding happens behind the ABI too)."""
    from dbgsom_amd import _native as nat

    lib = nat.load()
    rng = np.random.default_rng(1)
    M = rows * cols
    X = (rng.normal(size=(N, d)) + 3 * rng.integers(0, 5, size=(N, 1))).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    ctx = ctypes.c_void_p()
    nat.call("dbgsom_ctx_create", 0, ctypes.byref(ctx))
    try:
        nat.call("dbgsom_ctx_load", ctx, X.ctypes.data, nat.F32, N, d, nat.F32)
        nat.call("dbgsom_ctx_set_topology", ctx, hop.ctypes.data, M)
        idx = np.empty((N, 2), np.int64)
        dist = np.empty((N, 2), np.float64)
        nat.call("dbgsom_ctx_bmu", ctx, W.ctypes.data, M, 2, 0, idx.ctypes.data, dist.ctypes.data)
        rd, ri = o.bmu_chain(X, W, 2)
        assert np.array_equal(idx, ri) and np.array_equal(dist, rd)
        Wn = np.empty((M, d)); chg = np.empty(1); E = np.empty(M); a = np.empty(M)
        i1 = np.empty(N, np.int64); d1 = np.empty(N)
        gamma = float(np.var(X, axis=0).sum() ** -1)
        nat.call("dbgsom_ctx_epoch", ctx, W.ctypes.data, M, 0, gamma, 0.9, nat.CENTRES_COMPACT, 0,
                 Wn.ctypes.data, chg.ctypes.data, E.ctypes.data, a.ctypes.data, i1.ctypes.data,
                 d1.ctypes.data)
        info = (ctypes.c_double * 8)()
        nat.call("dbgsom_ctx_epoch_info", ctx, info)
        assert bool(info[0]) == (M >= 129)          # the filtered search ran behind the ABI
        oo = o.epoch(X, W, hop, 0.9, np.var(X, axis=0).sum(), "compact", "chain")
        assert np.array_equal(i1, oo.winners) and np.array_equal(d1, oo.distances)
        np.testing.assert_allclose(Wn, oo.new_weights, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(E, oo.errors, rtol=1e-12)
        assert np.array_equal(a, oo.activations)
        # the next epoch from the prototypes the first one left in HBM (W_host = NULL), seeded by
        # its winners; new prototypes stay in HBM too (W_new_host = NULL) and are fetched after
        nat.call("dbgsom_ctx_epoch", ctx, None, M, 0, gamma, 0.8, nat.CENTRES_COMPACT, 0, None,
                 chg.ctypes.data, E.ctypes.data, a.ctypes.data, i1.ctypes.data, d1.ctypes.data)
        nat.call("dbgsom_ctx_epoch_info", ctx, info)
        assert bool(info[3]) == (M >= 129)          # previous winners were the seeds
        o2 = o.epoch(X, oo.new_weights, hop, 0.8, np.var(X, axis=0).sum(), "compact", "chain")
        W2 = np.empty((M, d))
        nat.call("dbgsom_ctx_get_weights", ctx, 0, W2.ctypes.data, M)
        Wprev = np.empty((M, d))
        nat.call("dbgsom_ctx_get_weights", ctx, 1, Wprev.ctypes.data, M)
        assert np.array_equal(Wprev, Wn)            # what the second epoch consumed
        np.testing.assert_allclose(W2, o2.new_weights, rtol=1e-9, atol=1e-12)
        assert np.array_equal(a, o2.activations)
        Xq = rng.normal(size=(50, d))
        iq = np.empty((50, 1), np.int64); dq = np.empty((50, 1))
        nat.call("dbgsom_ctx_bmu_query", ctx, Xq.ctypes.data, nat.F64, 50, d, W.ctypes.data, M, 1,
                 0, iq.ctypes.data, dq.ctypes.data)
        rd, ri = o.bmu_chain(Xq, W, 1)
        assert np.array_equal(iq.ravel(), ri) and np.array_equal(dq.ravel(), rd)
        # error behaviour: status codes + message, nothing thrown across the ABI
        rc = lib.dbgsom_ctx_bmu(ctx, W.ctypes.data, M, 3, 0, idx.ctypes.data, dist.ctypes.data)
        assert rc == -1 and b"bad arguments" in lib.dbgsom_last_error()
        rc = lib.dbgsom_ctx_epoch(ctx, W.ctypes.data, M - 1, 0, gamma, 0.9, 0, 0, Wn.ctypes.data,
                                  chg.ctypes.data, E.ctypes.data, a.ctypes.data, None, None)
        assert rc == -4
    finally:
        nat.call("dbgsom_ctx_destroy", ctx)


def test_ctx_api_bf16_storage_and_allreduce_callback(o):
    """bfloat16 storage behind dbgsom_ctx_load, and the all-reduce seam: a callback that leaves
    the buffer alone (one rank) is called once per epoch with M * (d_padded + 3) + 1 values."""
    import torch

    from dbgsom_amd import _native as nat

    rng = np.random.default_rng(4)
    N, d, rows, cols = 5000, 72, 12, 12
    M = rows * cols
    X = (rng.normal(size=(N, d)) + 2 * rng.integers(0, 6, size=(N, 1))).astype(np.float32)
    Xr = _bf16_round(X)
    W = Xr[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    seen = []

    def cb(_user, ptr, count, stream):
        seen.append(int(count))
        return 0

    fn = nat.ALLREDUCE_FN(cb)
    ctx = ctypes.c_void_p()
    nat.call("dbgsom_ctx_create", 0, ctypes.byref(ctx))
    try:
        nat.call("dbgsom_ctx_load", ctx, X.ctypes.data, nat.F32, N, d, nat.BF16)
        nat.call("dbgsom_ctx_set_topology", ctx, hop.ctypes.data, M)
        nat.call("dbgsom_ctx_set_allreduce", ctx, fn, None)
        Wn = np.empty((M, d)); chg = np.empty(1); E = np.empty(M); a = np.empty(M)
        i1 = np.empty(N, np.int64); d1 = np.empty(N)
        nat.call("dbgsom_ctx_epoch", ctx, W.ctypes.data, M, 0, 1e-3, 1.1, nat.CENTRES_COMPACT, 0,
                 Wn.ctypes.data, chg.ctypes.data, E.ctypes.data, a.ctypes.data, i1.ctypes.data,
                 d1.ctypes.data)
        assert seen == [M * (80 + 3) + 1]
        oo = o.epoch(Xr, W, hop, 1.1, np.float64(1e3), "compact", "chain")
        assert np.array_equal(i1, oo.winners) and np.array_equal(d1, oo.distances)
        np.testing.assert_allclose(Wn, oo.new_weights, rtol=1e-11, atol=1e-13)
        # a failing callback comes back as a status code
        bad = nat.ALLREDUCE_FN(lambda u, p, c, s: 7)
        nat.call("dbgsom_ctx_set_allreduce", ctx, bad, None)
        rc = nat.load().dbgsom_ctx_epoch(ctx, W.ctypes.data, M, 0, 1e-3, 1.1, 0, 0, Wn.ctypes.data,
                                         chg.ctypes.data, E.ctypes.data, a.ctypes.data, None, None)
        assert rc == -6
    finally:
        nat.call("dbgsom_ctx_destroy", ctx)
    del torch


def test_ctx_rccl_inside_the_library_and_the_callback_seam():
    """The multi-GPU route of a NumPy caller: RCCL driven by the library itself
    (dbgsom_rccl_unique_id / dbgsom_rccl_comm_init / dbgsom_ctx_set_rccl: librccl resolved at run time, no
    callback, no torch) and, for other transports, a callback plugged into dbgsom_ctx_set_allreduce -- here
    ncclAllReduce on the same communicator through ctypes.  One rank (a test box has one GPU): the
    communicator comes up, the collective runs on the context's stream once per epoch, results are the
    single-rank ones either way.  Runs in a fresh process so that exactly one HIP runtime is loaded."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, ROOT)
from dbgsom_amd import _native as nat
from oracle import som_oracle as o
nat.load()                                                    # the HIP runtime first (RTLD_GLOBAL)
rng = np.random.default_rng(2)
N, d, rows, cols = 4000, 40, 12, 12
M = rows * cols
X = (rng.normal(size=(N, d)) + 3 * rng.integers(0, 5, size=(N, 1))).astype(np.float32)
W = X[rng.choice(N, M, replace=False)].astype(np.float64)
ii, jj = np.divmod(np.arange(M), cols)
hop = (np.abs(ii[:, None] - ii[None]) + np.abs(jj[:, None] - jj[None])).astype(np.float64)
ctx = ctypes.c_void_p()
nat.call("dbgsom_ctx_create", 0, ctypes.byref(ctx))
nat.call("dbgsom_ctx_load", ctx, X.ctypes.data, nat.F32, N, d, nat.F32)
nat.call("dbgsom_ctx_set_topology", ctx, hop.ctypes.data, M)
uid, comm = ctypes.create_string_buffer(128), ctypes.c_void_p()
nat.call("dbgsom_rccl_unique_id", uid)
nat.call("dbgsom_rccl_comm_init", uid, 1, 0, ctypes.byref(comm))
assert comm.value
nat.call("dbgsom_ctx_set_rccl", ctx, comm)
v = np.arange(5, dtype=np.float64)
nat.call("dbgsom_ctx_allreduce_host", ctx, v.ctypes.data, v.size)
assert np.array_equal(v, np.arange(5.0))
oo = o.epoch(X, W, hop, 1.4, np.float64(1e3), "compact", "chain")
def epoch():
    Wn = np.empty((M, d)); chg = np.empty(1); E = np.empty(M); a = np.empty(M)
    i1 = np.empty(N, np.int64); d1 = np.empty(N)
    nat.call("dbgsom_ctx_epoch", ctx, W.ctypes.data, M, 0, 1e-3, 1.4, nat.CENTRES_COMPACT, 0, Wn.ctypes.data,
             chg.ctypes.data, E.ctypes.data, a.ctypes.data, i1.ctypes.data, d1.ctypes.data)
    assert np.array_equal(i1, oo.winners) and np.array_equal(d1, oo.distances)
    np.testing.assert_allclose(Wn, oo.new_weights, rtol=1e-11, atol=1e-13)
    return Wn
W_lib = epoch()                                               # RCCL issued by the library
# ... and the smoothing in its sharded form on the same communicator: ncclReduceScatter of the (one) column block
# of the sums and ncclAllGather of the new prototypes, in place, issued by the library (forced: it is off for
# one rank and for maps this small)
nat.call("dbgsom_ctx_set_option", ctx, b"shard_smooth", 1)
W_shard = epoch()
n_shard = ctypes.c_int64(0)
nat.call("dbgsom_ctx_get_option", ctx, b"shard_epochs", ctypes.byref(n_shard))
assert n_shard.value == 1 and np.array_equal(W_shard, W_lib)
nat.call("dbgsom_ctx_set_option", ctx, b"shard_smooth", 2)
rccl = ctypes.CDLL(os.environ["DBGSOM_RCCL_LIB"])             # (the librccl of the HIP runtime in use)
rccl.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p]
calls = []
def allreduce(user, buf, count, stream):                      # ncclDouble = 8, ncclSum = 0
    calls.append(count)
    return rccl.ncclAllReduce(buf, buf, count, 8, 0, comm, stream)
cb = nat.ALLREDUCE_FN(allreduce)
nat.call("dbgsom_ctx_set_allreduce", ctx, cb, None)           # replaces the library's own
W_cb = epoch()
assert calls == [M * (48 + 3) + 1], calls
assert np.array_equal(W_lib, W_cb)
nat.call("dbgsom_ctx_destroy", ctx)
nat.call("dbgsom_rccl_comm_destroy", comm)
print("rccl ok")
'''.replace("ROOT", repr(root))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "rccl ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_full_size_properties(hip, o):
    """BASELINE config C4 shape (N=1e6, d=784, M=1024): size-independent properties + an oracle
    spot check on a sample subset."""
    import torch

    N, d, rows, cols = 1_000_000, 784, 32, 32
    M = rows * cols
    gen = torch.Generator(device="cuda").manual_seed(1004)
    centers = torch.randn(32, d, device="cuda", generator=gen) * 4
    lab = torch.randint(0, 32, (N,), device="cuda", generator=gen)
    X = (centers[lab] + torch.randn(N, d, device="cuda", generator=gen)).float()
    del lab
    hip.load_device(X)
    sel = torch.randperm(N, device="cuda", generator=gen)[:M]
    W = X[sel].double().cpu().numpy()
    hop = gi.lattice_hops(rows, cols)
    gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
    res = hip.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", True)
    win, dist = res.winners, res.distances
    # (1) oracle spot check: 3000 random samples, bit-exact
    pick = np.random.default_rng(0).choice(N, 3000, replace=False)
    Xs = X[torch.from_numpy(pick).cuda()].cpu().numpy()
    rd, ri = o.bmu_chain(Xs, W, 1)
    assert np.array_equal(win[pick], ri) and np.array_equal(dist[pick], rd)
    # (2) the samples that ARE prototypes find themselves (or an identical earlier row) at r ~ 0
    assert (dist[sel.cpu().numpy()] < 1e-4).all()
    # (3) conservation: hits sum to N, E sums to sum(dist), S sums to sum(kw * x)
    assert res.activations.sum() == N
    np.testing.assert_allclose(res.errors.sum(), dist.sum(), rtol=1e-10)
    kw = 1 - np.sqrt(1 - np.exp(-gamma * dist ** 2))
    S = hip.read_sums(M)[: M * d].reshape(M, d)
    col_ref = (torch.from_numpy(kw).cuda()[None, :] @ X.double()).view(-1)
    np.testing.assert_allclose(S.sum(axis=0), col_ref.cpu().numpy(), rtol=1e-9)
    # (4) permutation equivariance of the BMU step
    perm = torch.randperm(N, device="cuda", generator=gen)
    hip.load_device(X[perm].contiguous())
    d2, i2 = hip.bmu(W, 1)
    p = perm.cpu().numpy()
    assert np.array_equal(i2, win[p]) and np.array_equal(d2, dist[p])
    hip.release()
    # (5) the filtered search against the all-pairs search on the FULL arrays (k = 1 and k = 2), and the
    # whole epoch's new prototypes
    from dbgsom_amd.backend import HipBackend

    ex = HipBackend(algorithm="exact")
    ex.load_device(X)
    fi = HipBackend(algorithm="filtered")
    fi.load_device(X)
    re_, rf = (b.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", True) for b in (ex, fi))
    assert fi.filter_log[-1][0] == "filtered" and not ex.filter_log   # (what ran)
    assert np.array_equal(re_.winners, rf.winners) and np.array_equal(re_.distances, rf.distances)
    assert np.array_equal(re_.winners, win) and np.array_equal(re_.distances, dist)
    assert np.array_equal(re_.new_weights, rf.new_weights)
    de, ie = ex.bmu(W, 2)
    df, if_ = fi.bmu(W, 2)
    assert np.array_equal(ie, if_) and np.array_equal(de, df)
    ex.release()
    fi.release()


def test_device_reductions_f2_f3(hip, o):
    """QE / TE / hit counts / densities / class histogram as device
    reductions against the host statements."""
    rng = np.random.default_rng(17)
    N, d, rows, cols, C = 7001, 24, 6, 7, 5
    M = rows * cols
    X = rng.normal(size=(N, d)).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    W[5] += 40.0  # a dead neuron
    y = rng.integers(0, C, size=N).astype(np.int32)
    hip.load(X)
    hip.set_labels(y)
    dist, win = o.bmu_chain(X, W, 1)
    _, idx2 = o.bmu_chain(X, W, 2)
    np.testing.assert_allclose(hip.quantization_error(W), dist.mean(), rtol=1e-13)
    coords = [(i, j) for i in range(rows) for j in range(cols)]
    pos = np.asarray(coords, dtype=np.float64)
    te = int((np.linalg.norm(pos[idx2[:, 0]] - pos[idx2[:, 1]], axis=1) > 1.5).sum())
    assert hip.topographic_error_count(W, coords) == te
    sigma = 1.7
    hits, dens = hip.node_statistics(W, sigma)
    assert np.array_equal(hits, np.bincount(win, minlength=M))
    terms = np.exp(-(dist ** 2) / (2 * sigma ** 2)) / (sigma * np.sqrt(2 * np.pi))
    np.testing.assert_allclose(dens, np.bincount(win, weights=terms, minlength=M), rtol=1e-12,
                               atol=1e-300)
    ref = np.zeros((M, C), dtype=np.int64)
    np.add.at(ref, (win, y), 1)
    assert np.array_equal(hip.class_histogram(win, C, M), ref)
    res = hip.epoch(W, gi.lattice_hops(rows, cols), 1.0, 0.01, "compact", n_classes=C)
    assert np.array_equal(res.class_hist, ref)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("N,d", [(50_000, 100), (1797, 64), (33, 784), (200_000, 7)])
def test_column_moments_are_numpys_var_and_std_bit_for_bit(hip, N, d, dt):
    """np.var(X, 0) and np.std(X, 0, ddof=1) -- total variance and the "se" growing threshold of
    BaseSom._initialize_som (BaseSom.py:363, 380) -- from the resident samples."""
    rng = np.random.default_rng(N + d)
    X = (rng.normal(size=(N, d)) * rng.uniform(0.1, 30.0, size=d) + rng.normal(size=d) * 5).astype(dt)
    hip.load(X)
    s1, s2, n = hip.column_moments()
    assert n == N and s1.dtype == dt and s2.dtype == dt
    assert np.array_equal(s1, np.sum(X, axis=0))
    assert np.array_equal(np.true_divide(s2, N), np.var(X, axis=0))
    assert np.array_equal(np.sqrt(np.true_divide(s2, N - 1)), np.std(X, axis=0, ddof=1))


def _bf16_round(X):
    import torch

    return torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("N,d,M", [(3000, 256, 130), (1025, 2048, 260), (700, 13, 9)])
def test_bf16_storage_matches_oracle_on_rounded_samples(hip, o, N, d, M):
    """BASELINE config C5 shape (d=2048, bf16 samples), scaled: parity is against the oracle fed
    the same bf16-rounded X (the reference has no bf16)."""
    X, W = _rand(N, d, M, np.float32, 77 + d)
    Xr = _bf16_round(X)
    hip.load(X, storage="bf16")
    rows = cols = int(np.ceil(np.sqrt(M)))
    hop = gi.lattice_hops(rows, cols)[:M, :M]
    res = hip.epoch(W, hop, 1.5, 1e-3, "compact", True)
    oo = o.epoch(Xr, W, hop, 1.5, np.float64(1e3), "compact", "chain")
    assert np.array_equal(res.winners, oo.winners)
    assert np.array_equal(res.distances, oo.distances)
    np.testing.assert_allclose(res.new_weights, oo.new_weights, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(res.errors, oo.errors, rtol=1e-12)
    d2, i2 = hip.bmu(W, 2)
    rd, ri = o.bmu_chain(Xr, W, 2)
    assert np.array_equal(i2, ri) and np.array_equal(d2, rd)


def test_bf16_samples_filtered_search_is_identical_to_exact(o):
    """bfloat16-resident samples through the filtered search (float32 widened copy for the BMU
    kernels, bfloat16 rows for the accumulate step) = the all-pairs search = the oracle on the
    rounded samples."""
    from dbgsom_amd.backend import HipBackend

    N, d, rows, cols = 7000, 512, 20, 20
    M = rows * cols
    X, _ = gi.blobs_f32(N, d, 99)
    Xr = _bf16_round(X)
    W = Xr[np.random.default_rng(2).choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    ex = HipBackend(algorithm="exact").load(X, storage="bf16")
    fi = HipBackend(algorithm="filtered").load(X, storage="bf16")
    re_ = ex.epoch(W, hop, 3.0, 1e-3, "compact", True)
    rf = fi.epoch(W, hop, 3.0, 1e-3, "compact", True)
    assert fi.planes_cached and fi._x_np_dtype == "bf16"
    assert np.array_equal(re_.winners, rf.winners) and np.array_equal(re_.distances, rf.distances)
    assert np.array_equal(re_.new_weights, rf.new_weights)
    pick = np.random.default_rng(0).choice(N, 1500, replace=False)
    rd, ri = o.bmu_chain(Xr[pick], W, 1)
    assert np.array_equal(rf.winners[pick], ri) and np.array_equal(rf.distances[pick], rd)


@pytest.mark.parametrize("N,d,rows,cols", [(20000, 784, 16, 16), (5000, 64, 18, 19),
                                            (12345, 128, 17, 17), (9000, 208, 30, 30),
                                            (6000, 48, 23, 23), (8000, 96, 10, 13)])
def test_filtered_search_is_identical_to_exact(o, N, d, rows, cols):
    """The int8-MFMA filter + exact re-evaluation must reproduce the all-pairs float64 search bit
    for bit, over several epochs of a moving map (the filter uses the previous epoch's winners)."""
    from dbgsom_amd.backend import HipBackend

    M = rows * cols
    X, _ = gi.blobs_f32(N, d, 4242 + d)
    W = X[np.random.default_rng(1).choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    gamma = float(1.0 / np.var(X, axis=0).sum())
    exact = HipBackend(algorithm="exact").load(X)
    filt = HipBackend(algorithm="filtered_hint").load(X)   # pre-pass at epoch 0, hints afterwards
    stateless = HipBackend(algorithm="filtered").load(X)   # pre-pass every epoch
    three = HipBackend(algorithm="filtered").load(X)       # the six-product (3 digit planes) sweep
    three.sweep_planes = 3
    sigma = 0.2 * np.sqrt(M)
    We, Wf = W, W
    for e in range(5):
        re_ = exact.epoch(We, hop, sigma, gamma, "compact", True)
        rf = filt.epoch(Wf, hop, sigma, gamma, "compact", True)
        rs = stateless.epoch(We, hop, sigma, gamma, "compact", True)
        r3 = three.epoch(We, hop, sigma, gamma, "compact", True)
        for r in (rf, rs, r3):
            assert np.array_equal(re_.winners, r.winners), f"epoch {e}"
            assert np.array_equal(re_.distances, r.distances), f"epoch {e}"
            assert np.array_equal(re_.new_weights, r.new_weights), f"epoch {e}"
        We, Wf = re_.new_weights, rf.new_weights
        counts = filt.filter_counts()
        assert counts.min() >= 1 and counts.max() <= M
        sigma *= 0.6
    assert filt.planes_cached and stateless.planes_cached
    # oracle spot check of the last filtered epoch
    pick = np.random.default_rng(0).choice(N, 1500, replace=False)
    W_last_in = We if False else None  # (weights fed to the last epoch are not kept; check below)
    d1, i1 = filt.bmu(rf.new_weights, 1)
    rd, ri = o.bmu_chain(X[pick], rf.new_weights, 1)
    assert np.array_equal(i1[pick], ri) and np.array_equal(d1[pick], rd)


@pytest.mark.parametrize("N,d,rows,cols,algo", [(20_000, 100, 12, 13, "filtered"), (3000, 13, 3, 3, "auto"),
                                                 (9000, 50, 16, 17, "auto"), (2000, 17, 2, 2, "exact")])
def test_feature_counts_that_are_not_multiples_of_16_take_the_fast_paths_unchanged(o, N, d, rows, cols, algo):
    """Samples and prototypes are zero-padded to 16-feature multiples on their way to the device
    (LDS-DMA kernels, filtered search): zeros change no fma chain, results stay those of the
    oracle on the unpadded data, bit for bit; shapes that come back are the caller's."""
    from dbgsom_amd.backend import HipBackend

    M = rows * cols
    X, _ = gi.blobs_f32(N, d, 1000 + d)
    W = X[np.random.default_rng(4).choice(N, M, replace=False)].astype(np.float64) + 0.001
    hop = gi.lattice_hops(rows, cols)
    be = HipBackend(algorithm=algo).load(X)
    assert be.padded_features % 16 == 0 and be._d == d
    res = be.epoch(W, hop, 1.5, 1e-3, "compact", True)
    oo = o.epoch(X, W, hop, 1.5, np.float64(1e3), "compact", "chain")
    assert res.new_weights.shape == (M, d)
    assert np.array_equal(res.winners, oo.winners) and np.array_equal(res.distances, oo.distances)
    np.testing.assert_allclose(res.new_weights, oo.new_weights, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(res.errors, oo.errors, rtol=1e-12)
    assert abs(res.change_total - oo.change_total) <= 1e-10 * max(1.0, abs(oo.change_total))
    # device-resident chaining hands out (M, d) views and takes them back
    r1 = be.epoch(W, hop, 1.5, 1e-3, "compact", False, keep_on_device=True)
    assert r1.new_weights is None and r1.new_weights_dev is not None
    assert be.get_weights(0).shape == (M, d)
    r2 = be.epoch(r1.new_weights_dev, hop, 1.2, 1e-3, "compact", False)
    o2 = o.epoch(X, oo.new_weights, hop, 1.2, np.float64(1e3), "compact", "chain")
    np.testing.assert_allclose(r2.new_weights, o2.new_weights, rtol=1e-9, atol=1e-12)
    # queries on other samples, k = 1 and 2
    Xq = X[:777] * 1.25
    for k in (1, 2):
        dq, iq = be.bmu(W, k, Xq)
        rd, ri = o.bmu_chain(Xq, W, k)
        assert np.array_equal(iq, ri) and np.array_equal(dq, rd)
    s1, s2, n = be.column_moments()
    assert s1.shape == (d,) and np.array_equal(np.true_divide(s2, N), np.var(X, axis=0))


def test_large_queries_take_the_filtered_search_and_agree_with_the_all_pairs_kernel(o):
    """predict-style k = 1 queries on samples that are not resident (HipBackend.bmu(W, 1, X)) and
    post-fit queries on the resident ones go through the filtered search when they are large."""
    from dbgsom_amd.backend import HipBackend

    N, d, rows, cols = 40_000, 96, 19, 19
    M = rows * cols
    X, _ = gi.blobs_f32(N, d, 31)
    W = X[np.random.default_rng(3).choice(N, M, replace=False)].astype(np.float64) + 0.01
    auto = HipBackend(algorithm="auto").load(X[:5000])
    exact = HipBackend(algorithm="exact").load(X[:5000])
    assert auto.query_filter_applies(N, d, M, 1) and not exact.query_filter_applies(N, d, M, 1)
    da, ia = auto.bmu(W, 1, X)
    de, ie = exact.bmu(W, 1, X)
    assert np.array_equal(ia, ie) and np.array_equal(da, de)
    pick = np.random.default_rng(0).choice(N, 1000, replace=False)
    rd, ri = o.bmu_chain(X[pick], W, 1)
    assert np.array_equal(ia[pick], ri) and np.array_equal(da[pick], rd)
    # resident samples, digit planes cached by the query itself
    big = HipBackend(algorithm="auto").load(X)
    db, ib = big.bmu(W, 1)
    assert big.planes_cached and np.array_equal(ib, ie) and np.array_equal(db, de)
    d2, i2 = big.bmu(W, 2)          # k = 2 stays on the all-pairs kernel
    assert np.array_equal(i2[:, 0], ie)
    # the same queries with the search forced to its form without a sweep (what a context that has
    # settled on arm 0 during training runs for predict / the post-fit statistics)
    for be in (auto, big):
        be.sweep_planes = 4
    dp, ip = auto.bmu(W, 1, X)
    assert np.array_equal(ip, ie) and np.array_equal(dp, de)
    dp, ip = big.bmu(W, 1)
    assert np.array_equal(ip, ie) and np.array_equal(dp, de)
    np.testing.assert_allclose(big.quantization_error(W), de.mean(), rtol=1e-12)


def test_filtered_search_with_ties_and_bad_previous_winners(o):
    """Duplicated prototypes (exact ties -> lowest index) and deliberately wrong previous winners:
    the result must not depend on the quality of the hint."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(5)
    N, d, M = 6000, 32, 300
    X = rng.integers(-3, 4, size=(N, d)).astype(np.float32)
    base = rng.integers(-3, 4, size=(M // 3, d)).astype(np.float64)
    W = np.concatenate([base, base, base])  # every prototype appears 3 times
    be = HipBackend(algorithm="filtered_hint").load(X)
    hop = gi.lattice_hops(15, 20)
    r0 = be.epoch(W, hop, 2.0, 0.01, "compact", True)      # stateless two-pass, sets the hint
    r1 = be.epoch(W, hop, 2.0, 0.01, "compact", True)      # hinted, same W -> same answer
    assert np.array_equal(r0.winners, r1.winners) and np.array_equal(r0.distances, r1.distances)
    assert (r1.winners < M // 3).all()
    rd, ri = o.bmu_chain(X, W, 1)
    assert np.array_equal(r1.winners, ri) and np.array_equal(r1.distances, rd)
    # poison the hint: random previous winners (the bucket order must match them)
    be.set_hint(rng.integers(0, M, size=N), M)
    r2 = be.epoch(W, hop, 2.0, 0.01, "compact", True)
    assert np.array_equal(r2.winners, ri) and np.array_equal(r2.distances, rd)


def test_auto_policy_backs_off_on_near_duplicate_prototypes(o):
    """A map of near-duplicate prototypes makes every prototype a candidate: "auto" must notice
    and return to the exact kernel (results identical either way).  It first tries the arms that could still
    work (good seeds, finer sweeps: what unclustered data needs) -- each of them on trial: the call stops at its
    lists and the all-pairs kernel finds the winners, no exact stage ever runs over whole-map lists."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(8)
    N, d, M = 8000, 64, 400
    X = rng.normal(size=(N, d)).astype(np.float32)
    W = np.tile(rng.normal(size=(1, d)), (M, 1)) + 1e-9 * rng.normal(size=(M, d))
    hop = gi.lattice_hops(20, 20)
    be = HipBackend(algorithm="auto").load(X)
    ex = HipBackend(algorithm="exact").load(X)
    kinds = []
    for e in range(9):
        r = be.epoch(W, hop, 1.0, 1e-3, "aligned", True)
        q = ex.epoch(W, hop, 1.0, 1e-3, "aligned", True)
        assert np.array_equal(r.winners, q.winners) and np.array_equal(r.distances, q.distances)
        kinds.append(be.filter_log[-1][0])
    assert kinds[0] == "filtered"                            # a look ...
    n_tried = kinds.index("exact")
    assert 1 <= n_tried <= 5, kinds                          # ... at the few arms that could still work ...
    assert kinds[n_tried:] == ["exact"] * (9 - n_tried), kinds   # ... then back off
    assert all(entry[1] > HipBackend.FILTER_MAX_MEAN_CANDIDATES for entry in be.filter_log[:n_tried])
    assert be._get("guarded_calls") == n_tried               # every one of them stopped at its lists


def test_auto_policy_on_unclustered_data_and_across_a_growth_step(o):
    """Isotropic data in `auto` (the estimators' default): the first arm leaves the whole map a candidate --
    it is on trial, so the call stops at its lists and all pairs find the winners (no exact stage over the whole
    map); the next epochs go straight to the arms that can work (the winners as seeds, finer sweeps) instead of
    backing off to all pairs, and settle on short lists.  A growth step (one more lattice column) keeps what
    was learnt: no arm that failed is run again, nothing is on trial.  Results are exact throughout."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(21)
    N, d, rows, cols = 30_000, 256, 20, 20
    M = rows * cols
    X = rng.normal(size=(N, d)).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    be = HipBackend(algorithm="auto").load(X)
    ex = HipBackend(algorithm="exact").load(X)
    q = ex.epoch(W, hop, 3.0, 1e-3, "compact", True)
    log = []
    for e in range(8):
        r = be.epoch(W, hop, 3.0, 1e-3, "compact", True)
        assert np.array_equal(r.winners, q.winners) and np.array_equal(r.distances, q.distances)
        assert np.array_equal(r.new_weights, q.new_weights, equal_nan=True)
        log.append(be.filter_log[-1])
    kinds, means = [entry[0] for entry in log], [entry[1] for entry in log]
    assert kinds[0] == "filtered" and means[0] > 0.5 * M, log          # cheap seeds, coarse bound: (nearly) the whole map
    n_guarded = be._get("guarded_calls")
    assert n_guarded >= 1                                                # ... which stopped at its lists
    assert "exact" not in kinds[:4], kinds                               # no back-off in front of the arms that can work
    assert kinds[-1] == "filtered" and means[-1] < 0.25 * M, log         # settled on short lists
    # one more lattice column: the same context carries on
    cols2 = cols + 1
    M2 = rows * cols2
    W2 = np.concatenate([W, X[rng.choice(N, M2 - M, replace=False)].astype(np.float64)])
    hop2 = gi.lattice_hops(rows, cols2)
    q2 = ex.epoch(W2, hop2, 3.0, 1e-3, "compact", True)
    for e in range(4):
        r = be.epoch(W2, hop2, 3.0, 1e-3, "compact", True)
        assert np.array_equal(r.winners, q2.winners) and np.array_equal(r.distances, q2.distances)
        assert be.filter_log[-1][0] == "filtered" and be.filter_log[-1][1] < 0.25 * M2, be.filter_log[-4:]
    assert be._get("guarded_calls") == n_guarded                         # nothing went on trial again
    be.release()
    ex.release()


def test_seed_prepass_finds_the_informative_features(o):
    """Data whose information sits in two narrow feature ranges (everything else constant): the
    pre-pass samples the k-tiles in which the prototypes differ most, so the candidate lists stay
    short; results are exact either way."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(12)
    N, d, rows, cols = 30_000, 832, 16, 16
    M = rows * cols
    X = np.full((N, d), 0.5, dtype=np.float32)
    centres = rng.normal(size=(40, 128)).astype(np.float32) * 4
    lab = rng.integers(0, 40, size=N)
    informative = np.r_[70:134, 450:514]          # k-tiles 1-2 and 7-8 only
    X[:, informative] = centres[lab] + rng.normal(size=(N, 128)).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    fi = HipBackend(algorithm="filtered").load(X)
    fi.sweep_planes = 2     # (a constant 0.5 in 700 features: too coarse for the one-product sweep)
    ex = HipBackend(algorithm="exact").load(X)
    rf = fi.epoch(W, hop, 3.0, 1e-3, "compact", True)
    re_ = ex.epoch(W, hop, 3.0, 1e-3, "compact", True)
    assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
    counts = fi.filter_counts()
    assert counts.mean() < 0.5 * M, counts.mean()   # evenly spaced tiles {0, 4, 8} see almost nothing


@pytest.mark.parametrize("N,d,rows,cols", [(9000, 96, 14, 14), (5000, 200, 20, 21)])
def test_float64_samples_take_the_filtered_search(o, N, d, rows, cols):
    """float64 samples: digit planes from the float64 rows, exact re-evaluation on float64 tiles."""
    from dbgsom_amd.backend import HipBackend

    M = rows * cols
    X32, _ = gi.blobs_f32(N, d, 555)
    X = X32.astype(np.float64) + np.random.default_rng(6).normal(scale=1e-9, size=(N, d))  # not float32-representable
    W = X[np.random.default_rng(7).choice(N, M, replace=False)] + 1e-3
    hop = gi.lattice_hops(rows, cols)
    fi = HipBackend(algorithm="filtered").load(X)
    ex = HipBackend(algorithm="exact").load(X)
    for e in range(2):
        rf = fi.epoch(W, hop, 2.0, 1e-3, "compact", True)
        re_ = ex.epoch(W, hop, 2.0, 1e-3, "compact", True)
        assert fi.planes_cached
        assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
        assert np.array_equal(rf.new_weights, re_.new_weights)
        W = re_.new_weights
    pick = np.random.default_rng(0).choice(N, 1200, replace=False)
    rd, ri = o.bmu_chain(X[pick], W, 1)
    d1, i1 = fi.bmu(W, 1)
    assert np.array_equal(i1[pick], ri) and np.array_equal(d1[pick], rd)


def test_filtered_search_randomised_shapes_dtypes_and_options():
    """25 random configurations (float32 / float64 / bfloat16 samples, any feature count, 129-2600
    prototypes, clustered / uniform / integer / badly scaled data, seed strides, both sweep
    variants, stateless and hinted): winners, distances and new prototypes of the filtered search
    are those of the all-pairs kernel, bit for bit (`tools/stress_filtered.py` is the long form)."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(20260101)
    for case in range(25):
        N = int(rng.integers(130, 12000))
        d = int(rng.choice([16, 17, 31, 48, 64, 100, 129, 256, 500, 784]))
        M = int(rng.integers(129, 2600))
        kind = rng.choice(["blobs", "uniform", "dups", "scaled"])
        dt = rng.choice(["f32", "f64", "bf16"])
        if kind == "blobs":
            c = rng.normal(size=(int(rng.integers(2, 50)), d)) * 4
            X = c[rng.integers(0, len(c), size=N)] + rng.normal(size=(N, d))
        elif kind == "uniform":
            X = rng.uniform(-1, 1, size=(N, d))
        elif kind == "dups":
            X = rng.integers(-2, 3, size=(N, d)).astype(np.float64)
        else:
            X = rng.normal(size=(N, d)) * np.exp(rng.uniform(-6, 6, size=(1, d)))
        X = X.astype(np.float64 if dt == "f64" else np.float32)
        W = X[rng.choice(N, M, replace=M > N)].astype(np.float64)
        storage = "bf16" if dt == "bf16" else None
        ex = HipBackend(algorithm="exact").load(X, storage=storage)
        fi = HipBackend(algorithm="filtered_hint" if rng.random() < 0.5 else "filtered").load(X, storage=storage)
        fi.seed_stride = int(rng.choice([0, 1, 2, 8, 32]))
        fi.sweep_planes = int(rng.choice([0, 1, 2, 3, 4]))   # 0 = adaptive, 4 = no sweep (triangle pruning)
        hop = np.abs(np.subtract.outer(np.arange(M), np.arange(M))).astype(np.float64)
        for e in range(2):
            re_ = ex.epoch(W, hop, 1.5, 1e-3, "aligned", True)
            rf = fi.epoch(W, hop, 1.5, 1e-3, "aligned", True)
            tag = f"case {case} epoch {e}: N={N} d={d} M={M} {kind} {dt}"
            assert np.array_equal(re_.winners, rf.winners), tag
            assert np.array_equal(re_.distances, rf.distances), tag
            assert np.array_equal(re_.new_weights, rf.new_weights, equal_nan=True), tag
            W = np.nan_to_num(re_.new_weights)
        ex.release()
        fi.release()


@pytest.mark.parametrize("lo,hi", [(-12, 12), (-30, 5), (-5, 18)])
def test_filtered_search_with_extreme_row_scales(lo, hi):
    """Rows whose magnitudes span dozens of decades: the float32 chunk epilogue of the candidate
    sweep replaces operands that are not normal float32 numbers by "always marked" and otherwise
    keeps a rigorous slack; winners, distances and new prototypes stay those of the all-pairs
    kernel, bit for bit (zero rows and denormal-range rows included)."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(1000 + hi - lo)
    N, d, M = 6000, 96, 400
    X = (rng.normal(size=(N, d)) * 10.0 ** rng.uniform(lo, hi, size=(N, 1))).astype(np.float32)
    X[::97] = 0.0
    W = rng.normal(size=(M, d)) * 10.0 ** rng.uniform(lo, hi, size=(M, 1))
    W[5] = 0.0
    W[6] = X[3].astype(np.float64)
    hop = np.abs(np.subtract.outer(np.arange(M), np.arange(M))).astype(np.float64)
    ex = HipBackend(algorithm="exact").load(X)
    for planes in (1, 2, 0, 4):
        fi = HipBackend(algorithm="filtered").load(X)
        fi.sweep_planes = planes
        for e in range(2):
            re_ = ex.epoch(W, hop, 1.5, 1e-30, "aligned", True)
            rf = fi.epoch(W, hop, 1.5, 1e-30, "aligned", True)
            assert fi.filter_log[-1][0] == "filtered"
            assert np.array_equal(re_.winners, rf.winners), (planes, e)
            assert np.array_equal(re_.distances, rf.distances), (planes, e)
            assert np.array_equal(re_.new_weights, rf.new_weights, equal_nan=True), (planes, e)
            fi.algorithm = "filtered_hint"
        fi.release()
    ex.release()


def test_full_seed_prepass_on_weakly_clustered_data(o):
    """Isotropic data: the cheap stateless seeds (a subset of prototypes and features) leave nearly
    every prototype a candidate; the engine's seed policy moves to the full pre-pass (every
    prototype, every feature) and the lists collapse.  Results are those of the all-pairs kernel
    either way."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(21)
    N, d, rows, cols = 30_000, 256, 20, 20
    M = rows * cols
    X = rng.normal(size=(N, d)).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    fi = HipBackend(algorithm="filtered").load(X)
    ex = HipBackend(algorithm="exact").load(X)
    means, modes = [], []
    for e in range(6):
        rf = fi.epoch(W, hop, 3.0, 1e-3, "compact", True)
        re_ = ex.epoch(W, hop, 3.0, 1e-3, "compact", True)
        assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
        assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True)
        means.append(fi.filter_log[-1][1])
        modes.append(fi._get("seed_mode"))
    assert means[0] > 0.5 * M                       # cheap seeds: (nearly) everything is a candidate
    assert modes[-1] == 1 and means[-1] < 0.25 * M, (means, modes)
    pick = rng.choice(N, 1000, replace=False)
    rd, ri = o.bmu_chain(X[pick], W, 1)
    assert np.array_equal(rf.winners[pick], ri) and np.array_equal(rf.distances[pick], rd)


def test_adaptive_digit_planes_settle_on_the_cheaper_sweep(o):
    """sweep_planes = 0: the engine's arm policy (seeds x digit planes, cost model in engine.hip)
    starts with cheap seeds and the one-product sweep.  On clustered data a counting-only launch
    beside the first epoch's sweep shows that the triangle inequality leaves lists as short, and
    the policy drops the sweep (arm 0); where the coarse bounds mark the whole map (a tiny spread
    around a large mean: the same launch counts the whole map) it moves to a finer sweep.
    Results are exact under every arm it tries."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(3)
    N, d, rows, cols = 12_000, 256, 16, 20
    M = rows * cols
    hop = gi.lattice_hops(rows, cols)
    Xb, _ = gi.blobs_f32(N, d, 77)
    Xu = (rng.uniform(0.45, 0.55, size=(N, d))).astype(np.float32)   # tiny spread around a big mean
    for X, clustered in ((Xb, True), (Xu, False)):
        W = X[rng.choice(N, M, replace=False)].astype(np.float64)
        be = HipBackend(algorithm="filtered").load(X)
        ex = HipBackend(algorithm="exact").load(X)
        assert int(be.sweep_planes) == 0
        q = ex.epoch(W, hop, 2.0, 1e-3, "compact", True)
        probes = []
        for e in range(9):
            r = be.epoch(W, hop, 2.0, 1e-3, "compact", True)
            assert np.array_equal(r.winners, q.winners) and np.array_equal(r.distances, q.distances)
            probes.append(be.epoch_info()[6])
        used = [entry[2] for entry in be.filter_log]
        means = [entry[1] for entry in be.filter_log]
        assert used[0] == 1 and used[-1] == used[-2] and be._get("plane_hold") > 0, used   # settled
        # looked at arm 0 beside epoch 1 -- and once more beside epoch 2 when the first look met workgroups
        # with poor seeds (it is then repeated with the re-seeding passes on)
        seen = [v for v in probes[:2] if not np.isnan(v)]
        assert seen, probes
        if clustered:   # (on a set this small two more launches cost more than the sweep they replace:
            #              the one-product sweep stays; the full-size test sees arm 0 win)
            assert used[-1] in (0, 1) and means[-1] < 0.5 * M, (used, means, probes)
            assert seen[-1] < 0.5 * M, probes
        else:
            assert seen[-1] > 0.9 * M, probes                         # nothing to gain without a sweep
            assert 0 not in used, used                                # ... so it never ran
            assert used[-1] >= 2 and means[-1] < 0.5 * means[0], (used, means)
        # what it settled on is the cheapest arm it has seen (cost model of the digit planes alone)
        seen = {p: m for (_, m, p) in be.filter_log[-3:]}
        assert used[-1] == min(seen, key=lambda p: be.plane_cost(p, seen[p], M))


def test_search_arms_that_have_been_timed_are_compared_by_their_time():
    """The arm policy (engine.hip: adapt_arms) prices an arm it has never run by the cost model, but two
    arms that both ran clean -- nothing riding along, nothing copied to the host -- are compared by the
    measured wall clock of the epoch call (dbgsom_ctx_arm_ms).  Mid-clustered data (six clusters, lists of
    ~170 prototypes): the arm the policy settles on is, among the arms it tried, the fastest by the
    engine's own clock and within a tolerance the fastest by this test's (each arm forced on a fresh
    context).  Results are those of the all-pairs search whatever runs."""
    import time

    from dbgsom_amd.backend import RESIDENT, HipBackend

    rng = np.random.default_rng(21)
    N, d, rows, cols = 160_000, 256, 32, 32
    M = rows * cols
    c = rng.normal(size=(6, d)).astype(np.float32) * 4
    X = c[rng.integers(0, 6, N)] + rng.normal(size=(N, d)).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)

    arms = []

    def frozen(be, n):
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            be.epoch(RESIDENT, hop, 2.0, 1e-3, "compact", False, keep_on_device=True, frozen=True)
            out.append((time.perf_counter() - t0) * 1e3)
            info = be.epoch_info()
            arms.append((1 if info[7] else 0, int(info[2])))                     # (seeds, planes) of the epoch
        return out

    be = HipBackend(algorithm="filtered").load(X)
    be.set_weights(W)
    frozen(be, 24)
    tried, timed = list(arms), be.arm_ms()
    assert be._get("plane_hold") > 0 and tried[-1] == tried[-2] == tried[-3], (tried, timed)   # settled
    assert len(timed) >= 2, (timed, tried)                                       # a comparison by the clock took place
    settled = tried[-1]
    stateless = {a: t for a, t in timed.items() if a[0] != 2}
    assert settled in stateless and stateless[settled] <= 1.03 * min(stateless.values()), (settled, timed)
    lists = {e[2]: e[1] for e in be.filter_log if e[0] == "filtered"}
    assert 100 <= lists[0] <= 320, lists                                          # (the data the test is about)
    # the same arms, each forced, by this test's clock
    mine = {}
    for planes in sorted({a[1] for a in stateless}):
        fb = HipBackend(algorithm="filtered").load(X)
        fb.sweep_planes = planes if planes else 4
        fb.set_weights(W)
        mine[planes] = float(np.median(frozen(fb, 12)[3:]))
        fb.release()
    # (single-epoch wall clocks on a shared box: the engine timed its arms once each, this test nine times each;
    #  a quarter between the two clocks has been seen for the same arm)
    assert mine[settled[1]] <= 1.25 * min(mine.values()), (settled, mine, timed)
    # and the answers are the all-pairs kernel's
    r = be.epoch(RESIDENT, hop, 2.0, 1e-3, "compact", True, keep_on_device=True, frozen=True)
    ex = HipBackend(algorithm="exact").load(X)
    q = ex.epoch(W, hop, 2.0, 1e-3, "compact", True)
    assert np.array_equal(r.winners, q.winners) and np.array_equal(r.distances, q.distances)
    be.release(); ex.release()


def test_eight_wavefront_sweep_and_prepass_still_agree_with_the_all_pairs_kernel():
    """The 4-wavefront kernels are the default shape of the one-product sweep and of the seed
    pre-pass; the 8-wavefront ones (DBGSOM_SWEEP_SHAPE=8 / DBGSOM_PREPASS_SHAPE=8, read once per
    process) are checked in a child process by the randomised comparison against the all-pairs
    kernel."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DBGSOM_SWEEP_SHAPE="8", DBGSOM_PREPASS_SHAPE="8")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_filtered.py"), "7", "24"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "24 cases, 0 mismatches" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("N,d,M", [(3000, 32, 8192), (3000, 32, 9000), (100, 48, 200), (129, 16, 129),
                                   (5000, 4096, 300)])
def test_one_product_sweep_at_the_edges_of_its_shapes(N, d, M):
    """The two-workgroups-per-CU sweep takes maps of up to 8192 prototypes (its bitmask is 1 KB of
    LDS), larger ones fall back to the 8-wavefront kernel; fewer samples than one workgroup, a
    single prototype chunk and very long rows (64 k-tiles) go through it too.  Winners, distances
    and new prototypes are those of the all-pairs kernel, bit for bit."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(N + d + M)
    X = (rng.normal(size=(N, d)) + 2.0 * rng.integers(0, 5, size=(N, 1))).astype(np.float32)
    W = (X[rng.choice(N, M, replace=M > N)] + 1e-3 * rng.normal(size=(M, d))).astype(np.float64)
    hop = np.zeros((M, M))
    ex = HipBackend(algorithm="exact").load(X)
    fi = HipBackend(algorithm="filtered").load(X)
    fi.sweep_planes = 1
    for _ in range(2):  # stateless, then with the first epoch's winners as seeds
        re_ = ex.epoch(W, hop, 1.0, 1e-3, "compact", True)
        rf = fi.epoch(W, hop, 1.0, 1e-3, "compact", True)
        assert fi.filter_log[-1][0] == "filtered" and fi.filter_log[-1][2] == 1
        assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
        assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True)
        fi.algorithm = "filtered_hint"
    ex.release(); fi.release()


@pytest.mark.parametrize("dt", ["f32", "f64", "bf16"])
def test_candidates_from_the_triangle_inequality_alone(o, dt):
    """sweep_planes = 4: no candidate sweep.  The candidates of a sample are what the triangle
    inequality cannot rule out from a certified upper bound of its distance to the seed (one pass
    over the top digit plane of X) and certified lower bounds of the distances between prototypes
    (filter.hip section 2c).  Winners, distances and new prototypes are those of the all-pairs kernel and
    of the oracle, bit for bit -- stateless and seeded by the previous winners, on blobs (where the
    lists shrink to the sample's own cluster) and on data without clusters (where they cannot)."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(5)
    N, d, rows, cols = 20_000, 200, 18, 20
    M = rows * cols
    hop = gi.lattice_hops(rows, cols)
    Xb, _ = gi.blobs_f32(N, d, 31)
    Xi = rng.normal(size=(N, d)).astype(np.float32)
    for X, clustered in ((Xb, True), (Xi, False)):
        X = X.astype(np.float64) if dt == "f64" else X
        W = X[rng.choice(N, M, replace=False)].astype(np.float64)
        W[7] = W[3]                                   # duplicates: ties go to the lower index
        ex = HipBackend(algorithm="exact").load(X, storage="bf16" if dt == "bf16" else None)
        fi = HipBackend(algorithm="filtered").load(X, storage="bf16" if dt == "bf16" else None)
        fi.sweep_planes = 4
        for e in range(3):
            re_ = ex.epoch(W, hop, 2.0, 1e-3, "compact", True)
            rf = fi.epoch(W, hop, 2.0, 1e-3, "compact", True)
            assert fi.filter_log[-1][0] == "filtered" and fi.filter_log[-1][2] == 0
            assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
            assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True)
            if clustered and e == 0:   # (prototypes drawn from the samples: about M / 32 per blob)
                assert fi.filter_log[-1][1] < 0.35 * M, fi.filter_log[-1]
            if e == 0:
                Xr = X
                if dt == "bf16":
                    import torch
                    Xr = torch.from_numpy(X).to(torch.bfloat16).float().numpy()
                pick = rng.choice(N, 1500, replace=False)
                rd, ri = o.bmu_chain(Xr[pick], W, 1)
                assert np.array_equal(rf.winners[pick], ri) and np.array_equal(rf.distances[pick], rd)
            fi.algorithm = "filtered_hint"
            W = re_.new_weights
        ex.release(); fi.release()


def test_pruning_reseeds_workgroups_whose_cheap_seeds_missed_their_cluster(o):
    """The cheap stateless pre-pass looks at every 4th prototype.  A cluster none of whose prototypes
    has an index divisible by 4 sends its samples to seeds in OTHER clusters; twice that distance
    rules nothing out and their workgroups keep the whole map.  The engine notices (the kernel counts
    such workgroups), turns the re-seeding passes on (those workgroups alone are seeded again against
    every prototype and pruned again) and the lists shrink to the cluster; winners and distances are
    those of the all-pairs kernel before and after."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(17)
    N, d, M, k = 40_000, 64, 512, 8
    centres = rng.normal(size=(k, d)) * 6
    lab = rng.integers(0, k, N)
    X = (centres[lab] + rng.normal(size=(N, d))).astype(np.float32)
    # prototypes: cluster 0 only at indices that the stride-4 subset skips, everything else anywhere
    wl = np.empty(M, dtype=np.int64)
    skipped = [j for j in range(M) if j % 4 != 0][:24]
    wl[:] = rng.integers(1, k, M)
    wl[skipped] = 0
    wl[::4] = np.where(wl[::4] == 0, 1, wl[::4])
    W = centres[wl] + rng.normal(size=(M, d))
    hop = np.zeros((M, M))
    ex = HipBackend(algorithm="exact").load(X)
    fi = HipBackend(algorithm="filtered").load(X)
    fi.sweep_planes = 4
    q = ex.epoch(W, hop, 1.0, 1e-3, "compact", True)
    means, retry = [], []
    for e in range(3):
        r = fi.epoch(W, hop, 1.0, 1e-3, "compact", True)
        assert np.array_equal(r.winners, q.winners) and np.array_equal(r.distances, q.distances)
        assert np.array_equal(r.new_weights, q.new_weights, equal_nan=True)
        means.append(fi.filter_log[-1][1])
        retry.append(fi._get("prune_retry"))
    # (the third epoch may be the policy's look at the full pre-pass: only the first two are compared)
    assert retry[:2] == [1, 1] and means[1] < 0.75 * means[0], (means, retry)
    assert np.bincount(q.winners[lab == 0], minlength=M)[wl != 0].sum() == 0     # cluster 0 is won by its own
    ex.release(); fi.release()


@pytest.mark.parametrize("N,d,M", [(3000, 32, 8192), (3000, 32, 9000), (100, 48, 200), (129, 16, 129),
                                   (5000, 4096, 300), (20_000, 17, 1000)])
def test_pruning_at_the_edges_of_its_shapes(N, d, M):
    """The pruning form takes maps of up to 8192 prototypes (its gap matrix is 4 M^2 bytes); larger
    ones keep the one-product sweep.  Fewer samples than one workgroup, one prototype tile, very long
    rows, a feature count that is padded, a map that GROWS between two hinted epochs (the prototype
    shifts then cover the old rows only): winners, distances and new prototypes are those of the
    all-pairs kernel, bit for bit."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(N + d + M)
    X = (rng.normal(size=(N, d)) + 2.0 * rng.integers(0, 5, size=(N, 1))).astype(np.float32)
    W = (X[rng.choice(N, M, replace=M > N)] + 1e-3 * rng.normal(size=(M, d))).astype(np.float64)
    ex = HipBackend(algorithm="exact").load(X)
    fi = HipBackend(algorithm="filtered").load(X)
    fi.sweep_planes = 4
    Ms = [M, M, min(M + 7, 16000)] if M < 9000 else [M, M]
    for e, Me in enumerate(Ms):
        if Me > W.shape[0]:   # growth: seven more prototypes, the old ones nudged
            W = np.concatenate([W + 1e-4 * rng.normal(size=W.shape), X[rng.choice(N, Me - W.shape[0])].astype(np.float64)])
        hop = np.zeros((Me, Me))
        re_ = ex.epoch(W, hop, 1.0, 1e-3, "compact", True)
        rf = fi.epoch(W, hop, 1.0, 1e-3, "compact", True)
        assert fi.filter_log[-1][0] == "filtered" and fi.filter_log[-1][2] == (0 if Me <= 8192 else 1)
        assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
        assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True)
        fi.algorithm = "filtered_hint"   # from the second epoch on: seeds = the previous winners
    ex.release(); fi.release()


@pytest.mark.parametrize("planes", [1, 2, 3, 4])
def test_filtered_search_with_non_finite_prototype_rows(planes):
    """Dead neurons of the aligned centre layout are NaN rows of W (0/0 like the reference), and a
    diverged row may hold an infinity.  Such prototypes can never win; every form of the filtered
    search (sweeps with 1..3 digit planes, pruning without a sweep) must neither pick them nor let
    their meaningless digit planes rule out a real candidate -- also when a NaN row is the SEED."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(77 + planes)
    N, d, M = 9000, 80, 300
    X, _ = gi.blobs_f32(N, d, 9, n_centers=12)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    W[5] = np.nan
    W[17, 3] = np.nan
    W[40, 11] = np.inf
    W[41] = -np.inf
    hop = np.zeros((M, M))
    ex = HipBackend(algorithm="exact").load(X)
    fi = HipBackend(algorithm="filtered").load(X)
    fi.sweep_planes = planes
    q = ex.epoch(W, hop, 1.0, 1e-3, "aligned", True)
    assert not np.isin(q.winners, [5, 17, 40, 41]).any()
    for e in range(2):
        r = fi.epoch(W, hop, 1.0, 1e-3, "aligned", True)
        assert np.array_equal(r.winners, q.winners) and np.array_equal(r.distances, q.distances), (planes, e)
        fi.algorithm = "filtered_hint"
    # previous winners that have become NaN rows since: seeds without a distance
    bad = q.winners.copy()
    bad[::3] = 5
    bad[1::3] = 17
    fi.set_hint(bad, M)
    r = fi.epoch(W, hop, 1.0, 1e-3, "aligned", True)
    assert np.array_equal(r.winners, q.winners) and np.array_equal(r.distances, q.distances), planes
    ex.release(); fi.release()


# ---- per-sample refinement of the candidate lists (filter.hip section 2d, csrc/refine.h) ---------------
@pytest.mark.parametrize("dt", ["f32", "f64", "bf16"])
@pytest.mark.parametrize("planes", [0, 4])
def test_refined_search_is_identical_to_exact(o, dt, planes):
    """refine = 1: the four int8 digit products of the top two planes over each workgroup's list leave a
    sample its few possible winners, the samples are bucketed again by the likely one and the float64
    chain runs on the (sample, prototype) pairs alone.  Winners, distances and new prototypes are those
    of the all-pairs kernel and of the oracle bit for bit -- behind the sweep and behind the pruning
    form, stateless and hinted, for the three storage types (bfloat16 rows are streamed as stored),
    with duplicated prototypes (ties, and more than four inseparable candidates: the overflow path)."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(17)
    N, d, rows, cols = 30_000, 208, 20, 21           # d % 64 != 0: a partial last tile of the pair kernel
    M = rows * cols
    hop = gi.lattice_hops(rows, cols)
    X, _ = gi.blobs_f32(N, d, 77)
    X = X.astype(np.float64) if dt == "f64" else X
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    W[11] = W[5]                                      # a tie: the lower index wins
    W[40:47] = W[39]                                  # seven copies: more candidates than slots (the overflow
    W[100:140] = W[99]                                # kernel on their record); 41: more than a record (whole list)
    storage = "bf16" if dt == "bf16" else None
    ex = HipBackend(algorithm="exact").load(X, storage=storage)
    fi = HipBackend(algorithm="filtered").load(X, storage=storage)
    fi.refine = 1
    fi.sweep_planes = planes
    saw_overflow = False
    for e in range(4):
        re_ = ex.epoch(W, hop, 2.0, 1e-3, "compact", True)
        rf = fi.epoch(W, hop, 2.0, 1e-3, "compact", True)
        assert fi.filter_log[-1][0] == "filtered" and fi.refined
        assert np.array_equal(rf.winners, re_.winners), e
        assert np.array_equal(rf.distances, re_.distances), e
        assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True), e
        pairs, groups, overflow, _ = fi.refine_counts()
        assert groups == (N + 127) // 128 and N - overflow <= pairs <= 4 * N
        saw_overflow |= overflow > 0
        if e == 0:
            Xr = X
            if dt == "bf16":
                import torch
                Xr = torch.from_numpy(X).to(torch.bfloat16).float().numpy()
            pick = rng.choice(N, 1500, replace=False)
            rd, ri = o.bmu_chain(Xr[pick], W, 1)
            assert np.array_equal(rf.winners[pick], ri) and np.array_equal(rf.distances[pick], rd)
        fi.algorithm = "filtered_hint"
        W = np.nan_to_num(re_.new_weights)
        W[11] = W[5]
        W[40:47] = W[39]
        W[100:140] = W[99]
    assert saw_overflow, "the duplicated prototypes must exercise the overflow kernel"
    ex.release(); fi.release()


@pytest.mark.parametrize("N,d,M,kind", [(100, 48, 200, "blobs"), (129, 16, 129, "blobs"), (5000, 784, 700, "iso"),
                                          (4000, 64, 1500, "iso"), (6000, 2048, 300, "blobs"), (3000, 32, 4000, "iso")])
def test_refinement_at_the_edges_of_its_shapes(N, d, M, kind):
    """Fewer samples than a workgroup, lists in every tile class of the refinement, lists in two
    segments (isotropic data behind the pruning form: the whole map is a candidate -- up to 512 entries
    are refined in two passes), lists beyond that (left to the matrix-core stage), long rows."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(N + d)
    if kind == "blobs":
        X, _ = gi.blobs_f32(N, d, N + M)
    else:
        X = rng.normal(size=(N, d)).astype(np.float32)
    W = X[rng.choice(N, M, replace=M > N)].astype(np.float64) + (rng.normal(size=(M, d)) * 1e-3 if M > N else 0.0)
    hop = np.zeros((M, M))
    ex = HipBackend(algorithm="exact").load(X)
    for planes in (4, 2):
        fi = HipBackend(algorithm="filtered").load(X)
        fi.refine = 1
        fi.sweep_planes = planes
        # (experimental, off by default: the distance of a sample the refinement decided evaluated inside
        #  the sums kernel of the epoch -- segsum_dist_kernel; the same bits)
        fi.defer = planes == 2
        for e in range(2):
            re_ = ex.epoch(W, hop, 1.0, 1e-3, "aligned", True)
            rf = fi.epoch(W, hop, 1.0, 1e-3, "aligned", True)
            assert fi.filter_log[-1][0] == "filtered"
            assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
            assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True)
        fi.release()
    ex.release()


@pytest.mark.parametrize("dt,d,M", [("f32", 784, 300), ("f32", 256, 200), ("f32", 1040, 260), ("f32", 2048, 300),
                                    ("bf16", 2048, 400), ("bf16", 512, 300), ("bf16", 4096, 200), ("f64", 320, 300),
                                    ("f64", 784, 260), ("f32", 48, 200)])
def test_distances_of_decided_samples_inside_the_sums_kernel(o, dt, d, M):
    """`defer`: a sample the refinement narrows down to ONE candidate has its winner without its float row ever
    being read; its distance is the chain against that prototype, evaluated by the epoch's sums kernel
    (accumulate.hip: segsum_chain_kernel -- rows in registers, the chain on v_mfma_f64_4x4x4 through LDS ranges)
    on the one pass it makes over the rows anyway.  Winners, distances AND the new prototypes are the all-pairs
    search's bit for bit, for three storage types, rows of one and two column groups per thread, narrow rows with
    several row lanes (d = 256: the sums' order of additions must be segsum_kernel's), chunks with and without rows
    that need a distance, duplicated prototypes (undecided samples go through the pair kernel), ragged N."""
    import torch
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(d + M)
    N = 21_003
    X, _ = gi.blobs_f32(N, d, d, n_centers=12)
    if dt == "f64":
        X = X.astype(np.float64) * 1.0000001
    storage = "bf16" if dt == "bf16" else None
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    W[7] = W[3]
    W[50:54] = W[49]
    hop = np.abs(np.subtract.outer(np.arange(M), np.arange(M))).astype(np.float64)
    ex = HipBackend(algorithm="exact").load(X, storage=storage)
    fi = HipBackend(algorithm="filtered").load(X, storage=storage)
    fi.refine, fi.defer, fi.sweep_planes = 1, 1, 4
    for e in range(3):
        re_ = ex.epoch(W, hop, 1.5, 1e-3, "compact" if e else "aligned", True)
        rf = fi.epoch(W, hop, 1.5, 1e-3, "compact" if e else "aligned", True)
        assert fi.filter_log[-1][0] == "filtered" and fi.refined
        assert np.array_equal(rf.winners, re_.winners), e
        assert np.array_equal(rf.distances, re_.distances), e
        assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True), e
        assert np.array_equal(rf.errors, re_.errors)
        assert np.array_equal(rf.change_total, re_.change_total, equal_nan=True)   # (aligned layout: NaN rows of dead neurons)
        if e == 0:
            Xr = torch.from_numpy(X).to(torch.bfloat16).float().numpy() if dt == "bf16" else X
            pick = rng.choice(N, 1000, replace=False)
            rd, ri = o.bmu_chain(Xr[pick], W, 1)
            assert np.array_equal(rf.winners[pick], ri) and np.array_equal(rf.distances[pick], rd)
        W = np.nan_to_num(re_.new_weights)
        fi.algorithm = "filtered_hint"
    assert fi.defer_epochs == 3, fi.defer_epochs
    ex.release(); fi.release()


def test_refinement_with_non_finite_rows_and_extreme_scales():
    """A prototype row with a NaN or an infinity (dead neurons of the aligned layout) has no residual
    norm: the bound is void and every candidate is kept (overflow path); sample rows spanning dozens of
    decades keep the bound honest in float32."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(3)
    N, d, M = 12_000, 96, 260
    X, _ = gi.blobs_f32(N, d, 9)
    X *= np.exp(rng.uniform(-20, 12, size=(N, 1))).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = np.zeros((M, M))
    for bad in (None, np.nan, np.inf):
        Wb = W.copy()
        if bad is not None:
            Wb[17, 3] = bad
            Wb[200] = bad
        ex = HipBackend(algorithm="exact").load(X)
        fi = HipBackend(algorithm="filtered").load(X)
        fi.refine = 1
        for planes in (4, 1, 3):
            fi.sweep_planes = planes
            re_ = ex.epoch(Wb, hop, 1.0, 1e-3, "aligned", True)
            rf = fi.epoch(Wb, hop, 1.0, 1e-3, "aligned", True)
            assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances, equal_nan=True)
            assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True)
        ex.release(); fi.release()


def test_refinement_is_chosen_by_measurement():
    """refine = 2 (the default): the engine times the exact stage of the first training epochs of a map
    size with and without the refinement and keeps the faster form; results never depend on it."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(8)
    N, d, rows, cols = 200_000, 1024, 40, 40          # long rows, long lists: the refinement's ground
    M = rows * cols
    c = rng.normal(size=(12, d)).astype(np.float32) * 4
    X = c[rng.integers(0, 12, N)] + rng.normal(size=(N, d)).astype(np.float32)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    fi = HipBackend(algorithm="filtered").load(X)
    assert fi.refine == 2
    fi.set_weights(W)
    from dbgsom_amd.backend import RESIDENT
    seen = []
    ref = None
    for e in range(16):   # (the policy of the candidate lists -- seeds, digit planes -- settles first: every
        r = fi.epoch(RESIDENT, hop, 3.0, 1e-4, "compact", True, keep_on_device=True, frozen=True)  # change of
        seen.append(fi.refined)                                   # the lists by a quarter measures again)
        if ref is None:
            ref = r
        assert np.array_equal(r.winners, ref.winners) and np.array_equal(r.distances, ref.distances)
    assert seen[0] is False            # nothing is known about the lists yet
    assert True in seen[1:] and False in seen[1:], (seen, fi.filter_log)   # both forms were timed
    assert len(set(seen[-3:])) == 1, (seen, fi.filter_log)    # settled
    fi.release()


@pytest.mark.parametrize("dt", ["f32", "f64", "bf16"])
def test_two_nearest_prototypes_through_the_pruning_form(o, dt):
    """k = 2 (topographic error, BaseSom.py:924-953; Delaunay edges, :987) through the filtered search:
    with p2 the seed's nearest other prototype, two prototypes are within |x - w_p| + |w_p - w_p2| of the
    sample, so the pruning bound of k = 1 plus an upper bound of |w_p - w_p2| keeps everything that can be
    among the two nearest; the exact stage keeps (best, second).  Taken when the training epochs ran the
    pruning form; results are the all-pairs kernel's and the oracle's bit for bit -- ties, duplicated
    prototypes and a NaN row included -- and so is the topographic count."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(23)
    N, d, rows, cols = 30_000, 200, 18, 20
    M = rows * cols
    hop = gi.lattice_hops(rows, cols)
    X, _ = gi.blobs_f32(N, d, 55)
    X = X.astype(np.float64) if dt == "f64" else X
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    W[7] = W[3]                                   # duplicates: (best, second) = (3, 7) for their samples
    W[100:104] = W[99]
    storage = "bf16" if dt == "bf16" else None
    ex = HipBackend(algorithm="exact").load(X, storage=storage)
    fi = HipBackend(algorithm="filtered").load(X, storage=storage)
    fi.sweep_planes = 4
    coords = np.stack(np.divmod(np.arange(M), cols), axis=1)
    Xr = X
    if dt == "bf16":
        import torch
        Xr = torch.from_numpy(X).to(torch.bfloat16).float().numpy()
    for e in range(3):
        fi.epoch(W, hop, 0.8, 1e-3, "aligned", True)           # (the pruning form ran: k = 2 may follow it)
        Wn = W.copy()
        if e == 2:
            Wn[150] = np.nan                                    # a dead neuron of the aligned layout
        de, ie = ex.bmu(Wn, 2)
        df, i_f = fi.bmu(Wn, 2)
        assert fi._get("k2_filtered") == 1, (e, fi.epoch_info(), fi._get("planes_used"))
        assert np.array_equal(i_f, ie) and np.array_equal(df, de)
        pick = rng.choice(N, 1000, replace=False)
        rd, ri = o.bmu_chain(Xr[pick], Wn, 2)
        assert np.array_equal(i_f[pick], ri) and np.array_equal(df[pick], rd)
        assert fi.topographic_error_count(Wn, coords) == ex.topographic_error_count(Wn, coords)
        fi.algorithm = "filtered_hint"                          # seeds = the epoch's winners from now on
        W = W + rng.normal(size=W.shape) * 0.05                 # the map moves a little between epochs
        W[7] = W[3]
        W[100:104] = W[99]
    # data without clusters: the policy keeps a sweep, k = 2 stays with the all-pairs kernel
    Xi = rng.normal(size=(8000, 64)).astype(np.float32)
    Wi = Xi[rng.choice(8000, 300, replace=False)].astype(np.float64)
    fi2 = HipBackend(algorithm="filtered").load(Xi)
    ex2 = HipBackend(algorithm="exact").load(Xi)
    fi2.epoch(Wi, np.zeros((300, 300)), 1.0, 1e-2, "aligned", True)
    d2, i2 = fi2.bmu(Wi, 2)
    de2, ie2 = ex2.bmu(Wi, 2)
    assert np.array_equal(i2, ie2) and np.array_equal(d2, de2)
    for b in (ex, fi, fi2, ex2):
        b.release()


def test_smoothing_gemm_on_the_dma_ring_is_the_register_staged_one_bit_for_bit(o):
    """dbgsom_smooth: the LDS-DMA-ring GEMM against the register-staged kernel it replaced (same j-ascending
    accumulation chains, same split-K pieces): identical W' and change_total, bit for bit, for map sizes
    that are no multiples of the tiles, with and without split-K, and against the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, ROOT)
from dbgsom_amd.backend import HipBackend
rng = np.random.default_rng(4)
out = {}
for (N, d, M) in [(3000, 48, 130), (3000, 784, 1024), (2000, 100, 77), (2500, 2048, 4096), (1500, 64, 2025), (900, 16, 17)]:
    X = (rng.normal(size=(N, d)) + 2 * rng.integers(0, 6, size=(N, 1))).astype(np.float32)
    W = X[rng.choice(N, M, replace=M > N)].astype(np.float64) + rng.normal(size=(M, d)) * 1e-2
    r, c = np.divmod(np.arange(M), int(np.ceil(np.sqrt(M))))
    hop = (np.abs(r[:, None] - r[None]) + np.abs(c[:, None] - c[None])).astype(np.float64)
    be = HipBackend(algorithm="exact").load(X)
    res = be.epoch(W, hop, 1.7, 1e-3, "compact", True)
    out[(N, d, M)] = (res.new_weights, res.change_total)
    be.release()
np.savez(sys.argv[1], **{f"{k[0]}_{k[1]}_{k[2]}_W": v[0] for k, v in out.items()},
         **{f"{k[0]}_{k[1]}_{k[2]}_c": v[1] for k, v in out.items()})
'''.replace("ROOT", repr(root))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        files = {}
        for tag, env_val in (("dma", "0"), ("generic", "1")):
            files[tag] = os.path.join(td, tag + ".npz")
            env = dict(os.environ, DBGSOM_SMOOTH_GENERIC=env_val)
            out = subprocess.run([sys.executable, "-c", code, files[tag]], capture_output=True, text=True,
                                 timeout=600, env=env)
            assert out.returncode == 0, out.stderr[-3000:]
        a, b = np.load(files["dma"]), np.load(files["generic"])
        assert sorted(a.files) == sorted(b.files) and len(a.files) == 12
        for k in a.files:
            assert np.array_equal(a[k], b[k], equal_nan=True), k


def _step_shapes(counts):
    """(class, whole tiles, groups of four) of every step the exact stage takes for lists of these lengths
    (filter.hip, subset_exact_kernel: 16 / 32 / 48-entry steps; a last tile with up to 12 entries as groups)."""
    out = set()
    for c in np.unique(counts):
        c = int(c)
        cls = 1 if c <= 16 else (2 if c <= 32 else 3)
        sj, left = 16 * cls, c
        while left > 0:
            e = min(sj, left)
            tiles = (e + 15) // 16
            rem = e - 16 * (tiles - 1)
            out.add((cls, tiles - 1, (rem + 3) // 4) if rem <= 12 else (cls, tiles, 0))
            left -= e
    return out


@pytest.mark.parametrize("dtype,repeat,algo_opts", [(np.float32, 1, {}), (np.float32, 10, {}), (np.float64, 1, {}),
                                                    (np.float64, 10, {}), (np.float32, 1, {"refine": 1})])
def test_exact_stage_every_count_of_tiles_and_groups(o, dtype, repeat, algo_opts):
    """Lists of every length from 1 to 110: cluster c of the map is c + 1 nearly coincident prototypes, its samples
    (whole 128-sample buckets, seeded by their winners) can rule none of them out.  The exact stage then takes steps
    of every shape -- whole 16-prototype tiles x groups of four in each list-length class, the 64-sample workgroups
    of small sample sets (repeat = 1) and the 128-sample ones (repeat = 10: more than 1024 buckets) -- and winners
    and distances must be those of the all-pairs kernel bit for bit, and of the oracle's chain on a sample."""
    from dbgsom_amd.backend import HipBackend

    rng = np.random.default_rng(77)
    d, C = 32, 110
    centres = rng.normal(size=(C, d)) * 40.0
    mult = np.arange(1, C + 1)
    W = np.concatenate([centres[c] + rng.normal(size=(mult[c], d)) * 1e-6 for c in range(C)]).astype(np.float64)
    W[5] = W[4]                                        # (an exact duplicate inside a group of four: the lower index wins)
    M = W.shape[0]
    X = np.concatenate([centres[c] + rng.normal(size=(128 * repeat, d)) * 0.5 for c in range(C)]).astype(dtype)
    N = X.shape[0]
    ex = HipBackend(algorithm="exact").load(X)
    de, ie = ex.bmu(W, 1)
    fi = HipBackend(algorithm="filtered_hint").load(X)
    fi.sweep_planes = 4                                # candidates from the triangle inequality
    for k_, v_ in algo_opts.items():
        setattr(fi, k_, v_)
    fi.set_hint(ie, M)
    df, jf = fi.bmu(W, 1)
    assert np.array_equal(jf, ie) and np.array_equal(df, de)
    counts = fi.filter_counts()
    shapes = _step_shapes(counts)
    if not algo_opts:
        want = {(1, 0, 1), (1, 0, 2), (1, 0, 3), (1, 1, 0)} | {(2, 1, q) for q in (1, 2, 3)} | {(2, 2, 0)} | \
               {(3, f, q) for f in range(3) for q in range(4) if f + q} | {(3, 3, 0)}
        assert want <= shapes, sorted(want - shapes)
    pick = rng.choice(N, 1500, replace=False)
    rd, ri = o.bmu_chain(X[pick].astype(np.float32 if dtype == np.float32 else np.float64), W, 1)
    assert np.array_equal(jf[pick], ri) and np.array_equal(df[pick], rd)
    # the two nearest (k = 2 walks whole tiles) on the same lists
    d2, j2 = fi.bmu(W, 2)
    e2, i2 = ex.bmu(W, 2)
    assert np.array_equal(j2, i2) and np.array_equal(d2, e2)
    ex.release(); fi.release()
