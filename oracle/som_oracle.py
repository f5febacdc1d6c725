"""ORACLE -- TEST INFRASTRUCTURE ONLY (not the product, never imported by ``dbgsom_amd``).

CPU restatement (NumPy, plus ``bmu_chain.c`` for the order-pinned BMU) of the per-epoch hot path of
the reference ``dbgsom/BaseSom.py``.  Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and
``__graft_entry__.smoke()`` may import this module -- as the checker, never as the thing measured
or shipped.

Parity is PINNED: ``tests/test_oracle_golden.py`` checks every function here against the golden
vectors in ``tests/golden/`` that ``tools/make_golden.py`` captured by running the reference
itself (numpy 2.2.6, scikit-learn 1.7.2, networkx 3.4.2, scipy 1.15.3).

Map of functions to the reference (file:line under ``/root/reference``):

    bmu_blas / bmu_sklearn / bmu_chain   BaseSom.py:446-464   _get_winning_neurons
    exp_similarity                       BaseSom.py:533-538   _calculate_exp_similarity
    accumulate                           BaseSom.py:488-503, 1028-1055, 1058-1073
    voronoi_centers                      BaseSom.py:1044-1053 (+ compaction quirk Q1)
    gaussian_neighborhood                BaseSom.py:525-531
    smooth_matmul / smooth_broadcast     BaseSom.py:509-515
    change_total                         BaseSom.py:519-520
    epoch                                BaseSom.py:403-407   one pass of _grow_som's body
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force: bool = False) -> str:
    """Compile ``bmu_chain.c`` -> ``oracle/liboracle.so`` (gcc).  Building the checker is not
    using it; ``__graft_entry__.build()`` calls this too."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "bmu_chain.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build_c()
        lib = ctypes.CDLL(so)
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        lib.oracle_row_sqnorms.argtypes = [vp, ci, i64, i64, i64, vp]
        lib.oracle_row_sqnorms.restype = None
        lib.oracle_bmu_chain.argtypes = [vp, ci, i64, i64, i64, vp, i64, ci, vp, vp]
        lib.oracle_bmu_chain.restype = ci
        lib.oracle_accumulate.argtypes = [vp, ci, i64, i64, i64, vp, vp, vp, i64, vp, vp, vp, vp]
        lib.oracle_accumulate.restype = ci
        _LIB = lib
    return _LIB


def _as_x(X):
    X = np.asarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    X = np.ascontiguousarray(X)
    return X, (0 if X.dtype == np.float32 else 1)


# --------------------------------------------------------------------------------------
# a1  BMU search
# --------------------------------------------------------------------------------------
def _is_f32_pair(X, W) -> bool:
    """float32 samples AND float32 prototypes (epoch 0 of a float32 fit: the initial prototypes
    are rows of X, BaseSom.py:423-429): sklearn's ArgKmin32 still evaluates in float64 but
    hands the distances back rounded to float32.  Every other dtype mix returns full float64."""
    return np.asarray(X).dtype == np.float32 and np.asarray(W).dtype == np.float32


def bmu_blas(X, W, k: int = 1, chunk: int = 4096):
    """Expanded-L2 arg-k-min exactly as sklearn's brute engines state it
    (``r = (|x|^2 + (-2 x.w)) + |w|^2`` in float64, clamp 0, ties -> lowest j, sqrt), with the
    middle term from BLAS dgemm.  Reference: BaseSom.py:455-457 -> sklearn
    ``_argkmin.pyx.tp:471-510`` / ``pairwise.py:424-431``."""
    X = np.asarray(X)
    f32_pair = _is_f32_pair(X, W)
    W = np.asarray(W, dtype=np.float64)
    N, M = X.shape[0], W.shape[0]
    yy = np.einsum("ij,ij->i", W, W)
    idx = np.empty((N, k), dtype=np.int64)
    dist = np.empty((N, k), dtype=np.float64)
    for s in range(0, N, chunk):
        Xc = X[s:s + chunk].astype(np.float64, copy=False)
        xx = np.einsum("ij,ij->i", Xc, Xc)
        r = Xc @ W.T
        r *= -2.0
        r += xx[:, None]
        r += yy[None, :]
        np.maximum(r, 0.0, out=r)
        if k == 1:
            j = np.argmin(r, axis=1)  # first occurrence == lowest index
            idx[s:s + chunk, 0] = j
            dist[s:s + chunk, 0] = r[np.arange(r.shape[0]), j]
        else:
            order = np.argsort(r, axis=1, kind="stable")[:, :k]  # stable: (r, j) lexicographic
            idx[s:s + chunk] = order
            dist[s:s + chunk] = np.take_along_axis(r, order, axis=1)
    np.sqrt(dist, out=dist)
    if f32_pair:
        dist = dist.astype(np.float32).astype(np.float64)
    if k == 1:
        return dist.reshape(-1), idx.reshape(-1)
    return dist, idx


def bmu_sklearn(X, W, k: int = 1):
    """The reference's literal call (BaseSom.py:455-457).  Third-party engine; used to pin
    ``bmu_blas`` / ``bmu_chain`` and, when importable, as the CPU baseline's BMU step."""
    from sklearn.neighbors import NearestNeighbors

    nn = NearestNeighbors(n_neighbors=k)
    nn.fit(W)
    dist, idx = nn.kneighbors(X)
    if k == 1:
        return dist.reshape(-1), idx.reshape(-1)
    return dist, idx


def row_sqnorms_chain(A):
    A, dt = _as_x(A)
    out = np.empty(A.shape[0], dtype=np.float64)
    _lib().oracle_row_sqnorms(A.ctypes.data, dt, A.shape[0], A.shape[1], A.shape[1],
                              out.ctypes.data)
    return out


def bmu_chain(X, W, k: int = 1):
    """Same arithmetic with the dot-product order pinned to a sequential fma chain
    (``bmu_chain.c``) -- the bit-exact comparator for the HIP kernel."""
    f32_pair = _is_f32_pair(X, W)
    X, dt = _as_x(X)
    W = np.ascontiguousarray(W, dtype=np.float64)
    N, d = X.shape
    M = W.shape[0]
    if W.shape[1] != d:
        raise ValueError("feature mismatch")
    idx = np.empty((N, k), dtype=np.int64)
    dist = np.empty((N, k), dtype=np.float64)
    rc = _lib().oracle_bmu_chain(X.ctypes.data, dt, N, d, d, W.ctypes.data, M, k,
                                 idx.ctypes.data, dist.ctypes.data)
    if rc != 0:
        raise ValueError("oracle_bmu_chain: bad arguments")
    if f32_pair:
        dist = dist.astype(np.float32).astype(np.float64)
    if k == 1:
        return dist.reshape(-1), idx.reshape(-1)
    return dist, idx


# --------------------------------------------------------------------------------------
# a2  sample kernel
# --------------------------------------------------------------------------------------
def exp_similarity(distances, total_variance):
    """``k_i = 1 - sqrt(1 - exp(-gamma d_i^2))``, ``gamma = 1/total_variance``
    (BaseSom.py:533-538; numpy evaluates ``** 0.5`` as sqrt and ``** 2`` as square)."""
    # `total_variance` keeps the dtype np.var(X, axis=0).sum() gave it (BaseSom.py:363): for
    # float32 data the reciprocal is rounded to float32 before it meets the float64 distances.
    gamma = float(total_variance ** -1)
    d = np.asarray(distances, dtype=np.float64)
    return 1 - np.sqrt(1 - np.exp(-gamma * np.square(d)))


# --------------------------------------------------------------------------------------
# a3 / a4 / a7  per-neuron sums
# --------------------------------------------------------------------------------------
def accumulate(X, winners, kw, dist, M):
    """Id-indexed per-neuron sums of one epoch: ``S = sum kw_i x_i`` (M,d), ``K = sum kw_i``,
    ``a`` = hit counts (BaseSom.py:500-503), ``E = sum dist_i`` (BaseSom.py:1068-1073, serial
    semantics -- SURVEY.md Q2).  Serial C loop in sample order."""
    X, dt = _as_x(X)
    N, d = X.shape
    winners = np.ascontiguousarray(winners, dtype=np.int64)
    kw = np.ascontiguousarray(kw, dtype=np.float64)
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    S = np.empty((M, d), dtype=np.float64)
    K = np.empty(M, dtype=np.float64)
    a = np.empty(M, dtype=np.float64)
    E = np.empty(M, dtype=np.float64)
    rc = _lib().oracle_accumulate(X.ctypes.data, dt, N, d, d, winners.ctypes.data,
                                  kw.ctypes.data, dist.ctypes.data, M, S.ctypes.data,
                                  K.ctypes.data, a.ctypes.data, E.ctypes.data)
    if rc != 0:
        raise ValueError("winner index out of range")
    return S, K, a, E


def accumulate_numpy(X, winners, kw, dist, M):
    """Same sums with NumPy/SciPy only (CSR-matmul form); the CPU-baseline leg uses this one
    (all host cores through BLAS-free but vectorised sparse matmul)."""
    import scipy.sparse as sp

    X = np.asarray(X)
    N = X.shape[0]
    kw = np.asarray(kw, dtype=np.float64)
    A = sp.csr_matrix((kw, (np.asarray(winners), np.arange(N))), shape=(M, N))
    S = np.asarray(A @ X.astype(np.float64, copy=False))
    K = np.bincount(winners, weights=kw, minlength=M)
    a = np.bincount(winners, minlength=M).astype(np.float64)
    E = np.bincount(winners, weights=dist, minlength=M)
    return S, K, a, E


def voronoi_centers(S, K, a, layout: str = "compact"):
    """Weighted Voronoi centres ``c_j = S_j / K_j`` for non-empty neurons.

    ``layout="compact"`` reproduces the reference (quirk Q1, BaseSom.py:1045,1053): the centre of
    the r-th NON-EMPTY neuron is written to row r, trailing rows stay zero.  ``"aligned"`` keeps
    row = neuron id (the mathematically intended form)."""
    M, d = S.shape
    C = np.zeros((M, d), dtype=np.float64)
    live = np.flatnonzero(a > 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        c_live = S[live] / K[live, None]
    if layout == "compact":
        C[: live.size] = c_live
    elif layout == "aligned":
        C[live] = c_live
    else:
        raise ValueError(layout)
    return C


# --------------------------------------------------------------------------------------
# a5 / a6  neighbourhood, smoothing, convergence
# --------------------------------------------------------------------------------------
def gaussian_neighborhood(hop, sigma):
    """``h = exp(-(D^2 / (2 sigma^2)))`` (BaseSom.py:529); ``inf`` hop -> 0."""
    hop = np.asarray(hop, dtype=np.float64)
    return np.exp(-(hop ** 2 / (2 * sigma ** 2)))


def smooth_matmul(h, a, C):
    """``W' = ((h * a^T) @ C) / ((h * a^T) @ 1)`` -- BaseSom.py:509-515 as two matmuls."""
    g = h * a[None, :]
    num = g @ C
    den = g.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        return num / den[:, None]


def smooth_broadcast(h, a, C):
    """Literal form of BaseSom.py:509-515 with the (M, M, d) temporary (small M only)."""
    inter = h[:, :, np.newaxis] * a[:, np.newaxis]
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.sum(C * inter, axis=1) / np.sum(inter, axis=1)


def smooth_broadcast_rows(h, a, C, i0, i1):
    """Output rows [i0, i1) of the literal form (BaseSom.py:509-515): the (rows, M, d) slice of
    its (M, M, d) temporary -- lets a caller time / check the reference-faithful smoothing at
    sizes where the whole temporary does not fit (bench.py's CPU leg)."""
    inter = h[i0:i1, :, np.newaxis] * a[:, np.newaxis]
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.sum(C * inter, axis=1) / np.sum(inter, axis=1)


def change_total(W_old, W_new):
    """``sum_j |W_j - W'_j|_2`` (BaseSom.py:519-520)."""
    return float(np.sum(np.linalg.norm(W_old - W_new, axis=1)))


# --------------------------------------------------------------------------------------
# one epoch + a backend object for host-logic tests
# --------------------------------------------------------------------------------------
@dataclass
class EpochOut:
    new_weights: np.ndarray
    change_total: float
    errors: np.ndarray
    activations: np.ndarray
    winners: np.ndarray
    distances: np.ndarray
    sample_weights: np.ndarray
    centers: np.ndarray


def epoch(X, W, hop, sigma, total_variance, layout="compact", bmu="chain", smooth="matmul"):
    """a1..a7 in the order `_grow_som` runs them (BaseSom.py:403-407)."""
    M = W.shape[0]
    W64 = np.asarray(W, dtype=np.float64)
    fn = {"chain": bmu_chain, "blas": bmu_blas, "sklearn": bmu_sklearn}[bmu]
    dist, win = fn(X, W, 1)
    kw = exp_similarity(dist, total_variance)
    S, K, a, E = accumulate(X, win, kw, dist, M)
    C = voronoi_centers(S, K, a, layout)
    h = gaussian_neighborhood(hop, sigma)
    Wn = (smooth_matmul if smooth == "matmul" else smooth_broadcast)(h, a, C)
    return EpochOut(Wn, change_total(W64, Wn), E, a, win, dist, kw, C)


def exp_similarity_gamma(distances, gamma):
    """`exp_similarity` with gamma already formed (the backends receive gamma, not the variance)."""
    d = np.asarray(distances, dtype=np.float64)
    return 1 - np.sqrt(1 - np.exp(-float(gamma) * np.square(d)))


def _backend_base():
    from dbgsom_amd.backend import HotPathBackend

    return HotPathBackend


class OracleBackend(_backend_base()):
    """CPU stand-in with the same interface as ``dbgsom_amd.backend.HipBackend`` so that the
    host logic (growth, sigma schedule, estimator plumbing, the sharded all-reduce) can be tested
    without a GPU.  TESTS ONLY: the product's default backend is the HIP one and raises when it
    cannot load; nothing under ``dbgsom_amd/`` imports this class."""

    name = "oracle"

    def __init__(self, bmu: str = "chain"):
        self._fn = {"chain": bmu_chain, "blas": bmu_blas, "sklearn": bmu_sklearn}[bmu]
        self._init_args = (bmu,)
        self._X = None

    def load(self, X):
        self._X = np.ascontiguousarray(X)
        return self

    @property
    def n_samples(self):
        return self._X.shape[0]

    def bmu(self, W, k=1, X=None):
        Xq = self._X if X is None else np.ascontiguousarray(X)
        return self._fn(Xq, np.asarray(W), k)

    def exp_similarity(self, distances, gamma):
        return exp_similarity_gamma(distances, gamma)

    def _pack(self, S, K, a, E):
        import torch

        return torch.from_numpy(np.concatenate([S.reshape(-1), K, a, E]))

    def _local_sums(self, W, gamma, want_assignments):
        dist, win = self._fn(self._X, np.asarray(W), 1)
        kw = exp_similarity_gamma(dist, gamma)
        sums = self._pack(*accumulate(self._X, win, kw, dist, np.asarray(W).shape[0]))
        return sums, (win if want_assignments else None), (dist if want_assignments else None)

    def _sums_from(self, W, sample_weights, winners, distances):
        return self._pack(*accumulate(self._X, winners, sample_weights, distances,
                                      np.asarray(W).shape[0]))

    def _smooth(self, sums, W, hop, sigma, layout):
        W64 = np.asarray(W, dtype=np.float64)
        M, d = W64.shape
        v = sums.numpy()
        S, K, a, E = v[:M * d].reshape(M, d), v[M * d:M * d + M], v[M * d + M:M * d + 2 * M], \
            v[M * d + 2 * M:]
        C = voronoi_centers(S, K, a, layout)
        Wn = smooth_matmul(gaussian_neighborhood(hop, sigma), a, C)
        return Wn, change_total(W64, Wn), E.copy(), a.copy()

    def release(self):
        self._X = None
