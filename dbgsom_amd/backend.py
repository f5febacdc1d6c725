"""Hot-path backends: the object the estimator calls once per epoch.

``HipBackend`` drives the hand-written gfx950 kernels through the C ABI
(``include/dbgsom_hip.h``) with PyTorch used only as plumbing: device memory, the current HIP
stream, and ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI) for the one all-reduce of
the per-prototype sums per epoch.  There is no CPU fallback: constructing it without the built
library or without a GPU raises.

The four operations mirror the private methods of the reference's ``BaseSom``
(``dbgsom/BaseSom.py``):

    bmu(W, k)                    _get_winning_neurons(data, n_bmu)        :446-464
    exp_similarity(dist, gamma)  _calculate_exp_similarity(distances)     :533-538
    update(...)                  _update_weights(sample_weights, winners, data) :470-523
                                 + _write_accumulative_error              :541-561
    epoch(...)                   the fused body of _grow_som              :403-407
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _native


@dataclass
class EpochResult:
    new_weights: np.ndarray      # (M, d) float64
    change_total: float          # sum_j |W_j - W'_j|_2
    errors: np.ndarray           # (M,) per-neuron sum of BMU distances
    activations: np.ndarray      # (M,) hit counts
    winners: Optional[np.ndarray] = None    # (N_local,) int64
    distances: Optional[np.ndarray] = None  # (N_local,) float64
    new_weights_dev: object = None          # device-resident copy (HipBackend, keep_on_device)
    class_hist: Optional[np.ndarray] = None  # (M, n_classes) int64 when labels were attached


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous row shard [lo, hi) of rank `rank` (row-shard X once)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def dist_info():
    """(rank, world) of the default process group, (0, 1) when torch.distributed is not up."""
    try:
        import torch.distributed as td

        if td.is_available() and td.is_initialized():
            return td.get_rank(), td.get_world_size()
    except ImportError:  # pragma: no cover
        pass
    return 0, 1


def _group_is_up() -> bool:
    try:
        import torch.distributed as td

        return td.is_available() and td.is_initialized()
    except ImportError:  # pragma: no cover
        return False


class HotPathBackend:
    """Epoch template shared by the HIP backend and the test-only oracle backend:
    local per-prototype sums -> (all-reduce across sample shards) -> smoothing."""

    name = "abstract"

    # -- to implement -------------------------------------------------------------------------
    def load(self, X):  # upload-once residency
        raise NotImplementedError

    def bmu(self, W, k=1, X=None):
        raise NotImplementedError

    def exp_similarity(self, distances, gamma):
        raise NotImplementedError

    def _local_sums(self, W, gamma, want_assignments):
        """-> (sums tensor [M*(d+3)] float64 = [S | K | a | E], winners, distances)"""
        raise NotImplementedError

    def _sums_from(self, W, sample_weights, winners, distances):
        raise NotImplementedError

    def _smooth(self, sums, W, hop, sigma, layout):
        """-> (new_weights ndarray, change_total float, errors ndarray, activations ndarray)"""
        raise NotImplementedError

    # -- shared -------------------------------------------------------------------------------
    def _all_reduce(self, sums):
        rank, world = dist_info()
        # DBGSOM_FORCE_COLLECTIVE=1: issue the collective even in a 1-rank group (rehearsal of
        # the RCCL path on a single-GPU box)
        force = os.environ.get("DBGSOM_FORCE_COLLECTIVE") == "1"
        if world > 1 or (force and _group_is_up()):
            import torch.distributed as td

            td.all_reduce(sums, op=td.ReduceOp.SUM)  # one collective per epoch
        return sums

    def epoch(self, W, hop, sigma, gamma, layout="compact", want_assignments=False,
              n_classes=0):
        sums, win, dist = self._local_sums(W, gamma, want_assignments or n_classes > 0)
        sums = self._all_reduce(sums)
        Wn, chg, E, a = self._smooth(sums, W, hop, sigma, layout)
        res = EpochResult(Wn, chg, E, a, win if want_assignments else None,
                          dist if want_assignments else None)
        if n_classes > 0:
            res.class_hist = self.class_histogram(win, n_classes, np.asarray(W).shape[0])
        return res

    def update(self, W, hop, sigma, sample_weights, winners, distances, layout="compact"):
        sums = self._sums_from(W, sample_weights, winners, distances)
        sums = self._all_reduce(sums)
        return self._smooth(sums, W, hop, sigma, layout)

    # -- post-fit consumers of the BMU step; host defaults ----------
    def _reduce_host(self, arr):
        """Sum a small host array over the ranks (identity for one process)."""
        rank, world = dist_info()
        if world == 1:
            return arr
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        return self._all_reduce(t).numpy()

    def set_labels(self, y):
        """Attach integer class labels of the resident rows (entropy criterion)."""
        self._y = None if y is None else np.ascontiguousarray(y, dtype=np.int32)

    def quantization_error(self, W) -> float:
        dist, _ = self.bmu(W, 1)
        s = self._reduce_host(np.array([dist.sum(), dist.size], dtype=np.float64))
        return float(s[0] / s[1])

    def topographic_error_count(self, W, coords) -> int:
        _, idx = self.bmu(W, 2)
        pos = np.asarray(coords, dtype=np.float64)
        apart = np.linalg.norm(pos[idx[:, 0]] - pos[idx[:, 1]], axis=1) > 1.5
        return int(round(self._reduce_host(np.array([np.count_nonzero(apart)], np.float64))[0]))

    def node_statistics(self, W, sigma):
        """-> (hit_counts (M,), density_sums (M,)) of BaseSom._calculate_node_statistics."""
        dist, win = self.bmu(W, 1)
        m = np.asarray(W).shape[0]
        terms = np.exp(-(dist ** 2) / (2 * sigma ** 2)) / (sigma * np.sqrt(2 * np.pi))
        both = np.concatenate([np.bincount(win, minlength=m).astype(np.float64),
                               np.bincount(win, weights=terms, minlength=m)])
        both = self._reduce_host(both)
        return both[:m], both[m:]

    def class_histogram(self, winners, n_classes, M):
        h = np.zeros((M, n_classes), dtype=np.float64)
        np.add.at(h, (winners, self._y), 1.0)
        return self._reduce_host(h.reshape(-1)).reshape(M, n_classes).astype(np.int64)

    def release(self):
        pass

    def __deepcopy__(self, memo):
        # sklearn.clone deep-copies constructor parameters; device handles are not copyable,
        # a fresh backend with the same configuration is what a cloned estimator needs
        return self.__class__(*getattr(self, "_init_args", ()))


def _x_dtype_code(dt) -> int:
    if dt == "bf16":
        return _native.BF16
    if dt == np.float32:
        return _native.F32
    if dt == np.float64:
        return _native.F64
    raise ValueError(f"samples must be float32 or float64, got {dt}")


class HipBackend(HotPathBackend):
    """MI355X backend.  One instance per process / per GPU."""

    name = "hip"

    # the filtered search pays off once the all-pairs float64 work is large
    FILTER_MIN_PROTOTYPES = 129   # at or below 128 one chunk of the all-pairs kernel is cheaper (measured)
    # "auto": when the candidate lists of a filtered epoch average more than this many
    # prototypes per 128-sample workgroup (near-duplicate prototypes, e.g. a collapsed map) the
    # exact all-pairs kernel is cheaper: use it for the next FILTER_BACKOFF epochs, then re-probe
    FILTER_MAX_MEAN_CANDIDATES = 320
    FILTER_BACKOFF = 8
    # the stateless seed pre-pass looks at every seed_stride-th prototype (results do not depend on it)
    seed_stride = 0   # 0 = the library's choice (one 256-prototype chunk, at least every 4th)
    # digit planes of the candidate sweep: 1 = one int8 digit product (coarsest bound, half the
    # sweep time of 2), 2 = three, 3 = six (tightest, twice the time of 2); results do not depend on it
    sweep_planes = int(os.environ.get("DBGSOM_SWEEP_PLANES", "0"))   # 0 = adaptive (see _adapt_planes)

    def __init__(self, device: Optional[int] = None, algorithm: str = "auto"):
        """algorithm (all give IDENTICAL results):
          "exact"          all-pairs float64 MFMA search;
          "filtered"       stateless: coarse int8-MFMA pre-pass -> int8 candidate sweep -> exact
                           float64 search on the candidates; nothing from earlier epochs is used;
          "filtered_hint"  the same, but the previous epoch's winners replace the pre-pass when
                           they are available (training: they almost always still win);
          "auto"           "filtered_hint" with a back-off to "exact" while the candidate lists are
                           long (maps of near-duplicate prototypes).
        The filtered forms apply to float32 samples with d % 16 == 0 and 256 <= M <= 16000;
        otherwise the exact kernel runs."""
        self._lib = _native.load()  # raises when the extension is not built
        if algorithm not in ("auto", "exact", "filtered", "filtered_hint"):
            raise ValueError("algorithm must be 'auto', 'exact', 'filtered' or 'filtered_hint'")
        self.algorithm = algorithm
        self._init_args = (device, algorithm)
        import torch

        if not torch.cuda.is_available() or _native.device_count() < 1:
            raise RuntimeError(
                "dbgsom_amd.HipBackend needs a visible AMD GPU (MI355X / gfx950); found "
                f"torch.cuda.is_available()={torch.cuda.is_available()}, "
                f"hipGetDeviceCount()={_native.device_count()}. "
                "There is no CPU fallback in the product path.")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        self._torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._X = None
        self._xx = None
        self._x_np_dtype = None
        self._hop_key = None
        self._hop_dev = None
        self._ws = {}
        self._planes = None      # digit planes of the resident samples (filtered search)
        self._prev_idx = None    # winners of the previous epoch (device)
        self._order = None       # sample ids bucketed by those winners (device, int32)
        self._filter_backoff = 0
        self._filtered_this_epoch = False
        self._counts_host = None
        self.filter_log = []     # (epoch kind, mean candidates, digit planes) of the last epochs
        self._plane_state = {}
        self._planes_next = 1
        self._planes_used = 1
        # bench hook: a list here collects (name, start, end) HIP events recorded on the stream
        # the kernels are launched on
        self.kernel_events = None

    # -- helpers --------------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    def _timed_call(self, key, fn_name, *args):
        if self.kernel_events is None:
            _native.call(fn_name, *args)
            return
        ev = self._torch.cuda.Event
        a, b = ev(enable_timing=True), ev(enable_timing=True)
        stream = self._torch.cuda.current_stream(self.device)
        a.record(stream)
        _native.call(fn_name, *args)
        b.record(stream)
        self.kernel_events.append((key, a, b))

    def _buf(self, key, nbytes):
        """Reusable byte workspace (torch caching allocator blocks are >= 512-B aligned)."""
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = self._torch.empty(max(int(nbytes), 256), dtype=self._torch.uint8,
                                  device=self.device)
            self._ws[key] = t
        return t

    def _dev_f64(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return self._torch.from_numpy(a).to(self.device)

    # Feature padding.  The LDS-DMA kernels and the filtered search want rows of a multiple of 16
    # features; samples and prototypes are zero-padded to that on their way to the device.  Zeros
    # change nothing, bit for bit: every dot product and norm is a sequential fma chain, and
    # fma(0, 0, acc) == acc; sums, centres and new prototypes of the extra columns are zeros.
    @staticmethod
    def _padded(d):
        return (int(d) + 15) // 16 * 16

    def _pad_cols(self, t, dp):
        d = t.shape[1]
        if d == dp:
            return t.contiguous()
        out = self._torch.zeros((t.shape[0], dp), dtype=t.dtype, device=t.device)
        out[:, :d] = t
        return out

    def _dev_weights(self, W):
        """NumPy (M, d) prototypes -> float64 device tensor (M, padded d)."""
        Wd = self._dev_f64(W)
        if Wd.dim() != 2:
            raise ValueError("prototypes must be a 2-D array")
        return self._pad_cols(Wd, self._padded(Wd.shape[1]))

    def _norms(self, t, dtype_code, rows, d):
        out = self._torch.empty(rows, dtype=self._torch.float64, device=self.device)
        _native.call("dbgsom_row_sqnorms", self._p(t), dtype_code, rows, d, d, self._p(out),
                     self._stream())
        return out

    def _round_f32(self, W, xdtype):
        # float32 samples AND float32 prototypes: the reference's engine returns float32-rounded
        # distances (epoch 0 of a float32 fit).  Every other mix is full float64.
        return int((not isinstance(xdtype, str)) and xdtype == np.float32
                   and np.asarray(W).dtype == np.float32)

    # -- a8: residency --------------------------------------------------------------------------
    def load(self, X, storage=None):
        """Upload the samples once.  `storage="bf16"` keeps them in HBM as bfloat16 (rounded to
        nearest even on the device; all arithmetic stays float64 on the exactly widened values --
        an extension, the reference has no bf16)."""
        X = np.ascontiguousarray(X)
        code = _x_dtype_code(X.dtype)
        if X.ndim != 2 or X.shape[0] < 1:
            raise ValueError("X must be a non-empty 2-D array")
        Xd = self._torch.from_numpy(X).to(self.device)
        if storage == "bf16":
            return self.load_device(Xd.to(self._torch.bfloat16))
        if storage not in (None, "native"):
            raise ValueError("storage must be None or 'bf16'")
        return self.load_device(Xd)

    def load_device(self, X_dev):
        """Adopt samples that already live in HBM (bench: generated on the device)."""
        torch = self._torch
        if X_dev.dtype not in (torch.float32, torch.float64, torch.bfloat16) or X_dev.dim() != 2:
            raise ValueError("X_dev must be a 2-D float32/float64/bfloat16 tensor")
        self._d = int(X_dev.shape[1])
        X_dev = self._pad_cols(X_dev, self._padded(self._d))  # (N, padded d), zeros behind column d
        self._x_np_dtype = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64),
                            torch.bfloat16: "bf16"}[X_dev.dtype]
        self._X = X_dev
        self._xx = self._norms(X_dev, _x_dtype_code(self._x_np_dtype), X_dev.shape[0],
                               X_dev.shape[1])
        self._reset_filter_state()
        return self

    def _reset_filter_state(self):
        self._planes = self._prev_idx = self._order = None
        self._X32 = None

    def _bmu_samples(self):
        """(samples, dtype) the BMU kernels read.  bfloat16-resident samples get a float32 copy
        (exact widening, made once): the LDS-DMA kernels and the filtered search are written for
        float32 rows, and at 288 GB of HBM the copy (4 bytes per element next to the 2 of the
        resident array) is cheaper than a third tile format; the accumulate step, which is bound
        by the bytes of X it reads, keeps using the bfloat16 array."""
        if isinstance(self._x_np_dtype, str):
            if self._X32 is None:
                self._X32 = self._X.float()
            return self._X32, np.dtype(np.float32)
        return self._X, self._x_np_dtype

    def _filter_applies(self, M):
        if self.algorithm == "auto" and self._filter_backoff > 0:
            return False
        return (self.algorithm != "exact"
                and self.FILTER_MIN_PROTOTYPES <= M <= _native.MAX_PROTOTYPES)

    def _hint(self):
        """(previous winners, their bucket order) when the algorithm may use them."""
        if (self.algorithm in ("auto", "filtered_hint") and self._prev_idx is not None
                and self._order is not None and self._prev_idx.numel() == self._X.shape[0]):
            return self._p(self._prev_idx), self._p(self._order)
        return None, None

    def _make_planes(self, X32):
        N, d = X32.shape
        nbytes = self._lib.dbgsom_filter_planes_bytes(N, d)
        planes = self._torch.empty(nbytes, dtype=self._torch.uint8, device=self.device)
        _native.call("dbgsom_filter_prepare", self._p(X32), self._tcode(X32), N, d, d, self._p(planes),
                     nbytes, self._stream())
        return planes

    def _tcode(self, t):
        return _native.F32 if t.dtype == self._torch.float32 else _native.F64

    def _bmu_filtered_on(self, X32, xxd, planes, Wd, wwd, round_f32, prev_p, order_p, ws_key):
        # X32: float32 or float64 samples (bfloat16-resident ones come as their float32 copy)
        torch = self._torch
        N, d = X32.shape
        M = Wd.shape[0]
        idx = torch.empty((N, 1), dtype=torch.int64, device=self.device)
        dist = torch.empty((N, 1), dtype=torch.float64, device=self.device)
        need = self._lib.dbgsom_bmu_filtered_workspace_bytes(N, d, M)
        ws = self._buf(ws_key, need)
        self._planes_used = self._planes_for_call()
        self._timed_call("bmu", "dbgsom_bmu_filtered", self._p(X32), self._tcode(X32), N, d, d,
                         self._p(xxd), self._p(planes), self._p(Wd), M, self._p(wwd),
                         prev_p, order_p, int(self.seed_stride), int(self._planes_used), round_f32,
                         self._p(idx),
                         self._p(dist), self._p(ws), ws.numel(), self._stream())
        return dist, idx

    def _bmu_filtered_dev(self, Wd, wwd, round_f32):
        X32, _ = self._bmu_samples()
        if self._planes is None:  # digit planes of X: once per resident sample set
            self._planes = self._make_planes(X32)
        prev_p, order_p = self._hint()
        self._last_filter_M = Wd.shape[0]
        return self._bmu_filtered_on(X32, self._xx, self._planes, Wd, wwd, round_f32, prev_p,
                                     order_p, "filter")

    # queries (predict, post-fit statistics) below this many rows go to the all-pairs kernel: the
    # digit planes of a one-off X cost a pass over it
    FILTER_MIN_QUERY_ROWS = 32768

    def _query_filter_applies(self, N, d, M, xdtype, k):
        return (k == 1 and self.algorithm != "exact" and not isinstance(xdtype, str)
                and xdtype in (np.float32, np.float64) and d % 16 == 0 and N >= self.FILTER_MIN_QUERY_ROWS
                and self.FILTER_MIN_PROTOTYPES <= M <= _native.MAX_PROTOTYPES)

    def filter_counts(self):
        """Candidate-list length per 128-sample workgroup of the last filtered search."""
        N, d = self._X.shape
        nb = (N + 127) // 128
        out = np.empty(nb, dtype=np.uint32)
        _native.call("dbgsom_bmu_filtered_counts", self._p(self._ws["filter"]), N, d,
                     self._last_filter_M, out.ctypes.data, nb, self._stream())
        return out

    @property
    def n_samples(self):
        return int(self._X.shape[0])

    def _require_loaded(self):
        if self._X is None:
            raise RuntimeError("HipBackend: call load(X) first")

    # -- a1 -------------------------------------------------------------------------------------
    def _bmu_dev(self, Xd, xxd, xdtype, Wd, wwd, k, round_f32):
        torch = self._torch
        N, d = Xd.shape
        M = Wd.shape[0]
        idx = torch.empty((N, k), dtype=torch.int64, device=self.device)
        dist = torch.empty((N, k), dtype=torch.float64, device=self.device)
        self._timed_call("bmu", "dbgsom_bmu", self._p(Xd), _x_dtype_code(xdtype), N, d, d, self._p(xxd),
                     self._p(Wd), M, self._p(wwd), k, round_f32, self._p(idx), self._p(dist),
                     self._stream())
        return dist, idx

    def bmu(self, W, k=1, X=None):
        """-> (distances, winners) like BaseSom._get_winning_neurons: shape (N,) for k=1,
        (N, k) otherwise."""
        W = np.asarray(W)
        if X is None:
            self._require_loaded()
            (Xd, xdtype), xxd = self._bmu_samples(), self._xx
            src_dtype = self._x_np_dtype  # bf16-resident samples: no float32 rounding of distances
        else:
            X = np.ascontiguousarray(X)
            xdtype = src_dtype = X.dtype
            if X.ndim != 2 or W.ndim != 2 or W.shape[1] != X.shape[1]:
                raise ValueError("prototype / sample feature mismatch")
            Xd = self._torch.from_numpy(X).to(self.device)
            Xd = self._pad_cols(Xd, self._padded(X.shape[1]))
            xxd = self._norms(Xd, _x_dtype_code(xdtype), Xd.shape[0], Xd.shape[1])
        if W.ndim != 2 or self._padded(W.shape[1]) != Xd.shape[1] or (X is None and W.shape[1] != self._d):
            raise ValueError("prototype / sample feature mismatch")
        Wd = self._dev_weights(W)
        wwd = self._norms(Wd, _native.F64, Wd.shape[0], Wd.shape[1])
        rf = self._round_f32(W, src_dtype)
        if X is None and k == 1 and self._filter_applies(W.shape[0]):
            dist, idx = self._bmu_filtered_dev(Wd, wwd, rf)   # resident samples: planes are cached
        elif X is not None and self._query_filter_applies(Xd.shape[0], Xd.shape[1], W.shape[0], xdtype, k):
            dist, idx = self._bmu_filtered_on(Xd, xxd, self._make_planes(Xd), Wd, wwd, rf, None, None,
                                              "filter_query")
        else:
            dist, idx = self._bmu_dev(Xd, xxd, xdtype, Wd, wwd, k, rf)
        dist, idx = dist.cpu().numpy(), idx.cpu().numpy()
        if k == 1:
            return dist.reshape(-1), idx.reshape(-1)
        return dist, idx

    # -- a2 -------------------------------------------------------------------------------------
    def _exp_similarity_dev(self, dist_dev, gamma):
        kw = self._torch.empty_like(dist_dev)
        _native.call("dbgsom_exp_similarity", self._p(dist_dev), dist_dev.numel(), float(gamma),
                     self._p(kw), self._stream())
        return kw

    def exp_similarity(self, distances, gamma):
        dd = self._dev_f64(np.asarray(distances).reshape(-1))
        return self._exp_similarity_dev(dd, gamma).cpu().numpy()

    # -- a3 / a4 / a7 ---------------------------------------------------------------------------
    def _accumulate_dev(self, idx_dev, kw_dev, dist_dev, M):
        torch = self._torch
        N, d = self._X.shape
        if M > _native.MAX_PROTOTYPES:
            raise ValueError(f"M={M} exceeds DBGSOM_MAX_PROTOTYPES={_native.MAX_PROTOTYPES}")
        sums = torch.empty(M * (d + 3), dtype=torch.float64, device=self.device)
        # the workspace's bucket order is about to be rewritten: the (winners, order) pair the
        # filtered search relies on is only re-established by _local_sums
        self._prev_idx = self._order = None
        need = self._lib.dbgsom_accumulate_workspace_bytes(N, d, M)
        ws = self._buf("acc", need)
        status = self._buf("status", 256)
        self._timed_call("accumulate", "dbgsom_accumulate", self._p(self._X), _x_dtype_code(self._x_np_dtype), N, d,
                     d, self._p(idx_dev), self._p(kw_dev), self._p(dist_dev), M, self._p(sums),
                     self._p(status), self._p(ws), ws.numel(), self._stream())
        return sums

    def _as_dev_weights(self, W):
        """(device float64 tensor, round_f32 flag) from a NumPy array or a device tensor that a
        previous epoch left in HBM."""
        dp = self._X.shape[1]
        if self._torch.is_tensor(W):
            if W.dtype != self._torch.float64 or W.device != self.device or W.dim() != 2:
                raise ValueError("device weights must be a float64 matrix on the backend's GPU")
            if W.shape[1] == dp and W.is_contiguous():
                return W, 0
            if W.shape[1] == self._d and W.stride(1) == 1 and W.stride(0) == dp:
                # the (M, d) view of a padded (M, dp) result that a previous epoch handed out
                return W.as_strided((W.shape[0], dp), (dp, 1), W.storage_offset()), 0
            if W.shape[1] == self._d:
                return self._pad_cols(W, dp), 0
            raise ValueError("device weights do not match the resident samples' feature count")
        W = np.asarray(W)
        if W.ndim != 2 or W.shape[1] != self._d:
            raise ValueError("prototype / sample feature mismatch")
        return self._dev_weights(W), self._round_f32(W, self._x_np_dtype)

    def _local_sums(self, W, gamma, want_assignments):
        self._require_loaded()
        Wd, rf = self._as_dev_weights(W)
        self._W_dev = Wd
        wwd = self._norms(Wd, _native.F64, Wd.shape[0], Wd.shape[1])
        if self._filter_applies(Wd.shape[0]):
            self._last_filter_M = Wd.shape[0]
            self._filtered_this_epoch = True
            dist, idx = self._bmu_filtered_dev(Wd, wwd, rf)
        else:
            self._filtered_this_epoch = False
            if self._filter_backoff > 0:
                self._filter_backoff -= 1
            Xb, xb_dtype = self._bmu_samples()
            dist, idx = self._bmu_dev(Xb, self._xx, xb_dtype, Wd, wwd, 1, rf)
        dist, idx = dist.view(-1), idx.view(-1)
        kw = self._exp_similarity_dev(dist, gamma)
        self._last_idx = idx
        sums = self._accumulate_dev(idx, kw, dist, Wd.shape[0])
        # the next epoch's filter visits the samples bucketed by this epoch's winners: the
        # stable counting sort the accumulate step just did (first N int32 of its workspace)
        self._prev_idx = idx
        self._order = self._ws["acc"][: 4 * idx.numel()].view(self._torch.int32)
        if want_assignments:
            return sums, idx.cpu().numpy(), dist.cpu().numpy()
        return sums, None, None

    def _sums_from(self, W, sample_weights, winners, distances):
        self._require_loaded()
        torch = self._torch
        self._W_dev = self._dev_weights(W)
        idx = torch.from_numpy(np.ascontiguousarray(winners, dtype=np.int64)).to(self.device)
        kw = self._dev_f64(sample_weights)
        dist = self._dev_f64(distances)
        return self._accumulate_dev(idx, kw, dist, np.asarray(W).shape[0])

    # -- a5 / a6 --------------------------------------------------------------------------------
    def _hop(self, hop):
        # the estimator hands over the SAME array object until the lattice changes
        if self._hop_dev is None or hop is not self._hop_key:
            self._hop_dev = self._torch.from_numpy(
                np.ascontiguousarray(hop, dtype=np.float32)).to(self.device)
            self._hop_key = hop
        return self._hop_dev

    def epoch(self, W, hop, sigma, gamma, layout="compact", want_assignments=False,
              keep_on_device=False, n_classes=0):
        """One hot-path epoch.  `W` may be a NumPy array or the `new_weights_dev` tensor of the
        previous epoch; with `keep_on_device` the new prototypes stay in HBM (no PCIe round trip
        between epochs of a phase without growth) and only the O(M) statistics come back."""
        sums, win, dist = self._local_sums(W, gamma, want_assignments)
        sums = self._all_reduce(sums)
        Wn, chg, E, a = self._smooth(sums, W, hop, sigma, layout, keep_on_device)
        res = (EpochResult(None, chg, E, a, win, dist, Wn) if keep_on_device
               else EpochResult(Wn, chg, E, a, win, dist))
        if n_classes > 0:
            res.class_hist = self._class_hist_dev(self._last_idx, n_classes, Wn.shape[0])
        self._last_idx = None
        self._update_filter_policy()
        return res

    def _update_filter_policy(self):
        """After the epoch has completed (the stream is already synchronised): look at how long
        the candidate lists were and decide what the next epochs use."""
        if not self._filtered_this_epoch:
            if self.algorithm != "exact":
                self.filter_log.append(("exact", None))
            return
        counts, self._counts_host = self._counts_host, None
        mean = float((counts if counts is not None else self.filter_counts()).mean())
        self.filter_log.append(("filtered", mean, self._planes_used))
        del self.filter_log[:-64]
        if int(self.sweep_planes) == 0:
            self._adapt_planes(mean, self._last_filter_M)
        if self.algorithm == "auto":
            if mean > self.FILTER_MAX_MEAN_CANDIDATES:  # exponential back-off, capped
                self._filter_fail = min(getattr(self, "_filter_fail", 0) + 1, 6)
                self._filter_backoff = self.FILTER_BACKOFF << (self._filter_fail - 1)
            else:
                self._filter_fail = 0

    # Digit planes of the candidate sweep when sweep_planes == 0: one product costs about half of
    # three, six about twice as much; a list entry of the exact stage costs about as much as
    # sweeping 16 prototypes with three products (both scale with N and d).  Start with one
    # product; look at a finer sweep when even emptying the lists would pay for it, keep whichever
    # is cheaper in this model, look again every PLANES_REPROBE epochs.  Results never depend on it.
    SWEEP_COST = {1: 0.54, 2: 1.0, 3: 1.96}
    LIST_COST = 16.4
    PLANES_REPROBE = 64

    def _plane_cost(self, p, mean, M):
        return self.SWEEP_COST[p] * M + self.LIST_COST * mean

    def _adapt_planes(self, mean, M):
        st = self._plane_state
        p = self._planes_used
        if st.get("M") != M:            # another map size: what was learnt no longer applies
            st.clear()
            st.update(M=M, known={}, hold=0)
        st["known"][p] = mean
        if st["hold"] > 0:
            st["hold"] -= 1
            if st["hold"] == 0:
                st["known"] = {p: mean}  # forget the alternatives, they get another look
            return
        known = st["known"]
        best = min(known, key=lambda q: self._plane_cost(q, known[q], M))
        # a finer sweep can at best empty the lists; a coarser one at worst ... is simply tried
        finer, coarser = best + 1, best - 1
        if finer <= 3 and finer not in known and \
                self.LIST_COST * known[best] > (self.SWEEP_COST[finer] - self.SWEEP_COST[best]) * M:
            self._planes_next = finer
        elif coarser >= 1 and coarser not in known:
            self._planes_next = coarser
        else:
            self._planes_next = best
            st["hold"] = self.PLANES_REPROBE

    def _planes_for_call(self):
        fixed = int(self.sweep_planes)
        return fixed if fixed else self._planes_next

    # -- f-2 / f-3: reductions that keep the N-sized arrays in HBM -----------------------------
    def _bmu_resident_dev(self, W, k):
        self._require_loaded()
        Wd, rf = self._as_dev_weights(W)
        wwd = self._norms(Wd, _native.F64, Wd.shape[0], Wd.shape[1])
        if k == 1 and self._filter_applies(Wd.shape[0]):
            return self._bmu_filtered_dev(Wd, wwd, rf)
        Xb, xb_dtype = self._bmu_samples()
        return self._bmu_dev(Xb, self._xx, xb_dtype, Wd, wwd, k, rf)

    def column_moments(self):
        """(sum_i x_ij, sum_i (x_ij - mean_j)^2, N) over the resident samples in NumPy's axis-0
        arithmetic (sequential per column, X's dtype): np.var(X, 0) = s2 / N and
        np.std(X, 0, ddof=1) = sqrt(s2 / (N - 1)) bit for bit, without a host pass over X.
        None when the resident dtype has no NumPy counterpart (bfloat16)."""
        self._require_loaded()
        if isinstance(self._x_np_dtype, str):
            return None
        torch = self._torch
        N, d = self._X.shape
        code = _x_dtype_code(self._x_np_dtype)
        s1 = torch.empty(d, dtype=self._X.dtype, device=self.device)
        _native.call("dbgsom_column_sums", self._p(self._X), code, N, d, d, None, self._p(s1),
                     self._stream())
        mean = torch.from_numpy(np.true_divide(s1.cpu().numpy(), N)).to(self.device)
        s2 = torch.empty_like(s1)
        _native.call("dbgsom_column_sums", self._p(self._X), code, N, d, d, self._p(mean),
                     self._p(s2), self._stream())
        return s1.cpu().numpy()[:self._d], s2.cpu().numpy()[:self._d], N

    def _sum_dev(self, v):
        torch = self._torch
        out = torch.empty(1, dtype=torch.float64, device=self.device)
        ws = self._buf("sum", self._lib.dbgsom_sum_workspace_bytes())
        _native.call("dbgsom_sum_f64", self._p(v), v.numel(), self._p(out), self._p(ws),
                     ws.numel(), self._stream())
        return out

    def quantization_error(self, W) -> float:
        dist, _ = self._bmu_resident_dev(W, 1)
        t = self._torch.cat([self._sum_dev(dist.view(-1)),
                             self._torch.tensor([float(dist.numel())], dtype=self._torch.float64,
                                                device=self.device)])
        t = self._all_reduce(t).cpu().numpy()
        return float(t[0] / t[1])

    def topographic_error_count(self, W, coords) -> int:
        torch = self._torch
        _, idx = self._bmu_resident_dev(W, 2)
        xy = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.int32)).to(self.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        _native.call("dbgsom_topographic_count", self._p(idx), idx.shape[0], self._p(xy),
                     xy.shape[0], self._p(cnt), self._stream())
        return int(self._all_reduce(cnt.double()).item())

    def node_statistics(self, W, sigma):
        torch = self._torch
        dist, idx = self._bmu_resident_dev(W, 1)
        dist, idx = dist.view(-1), idx.view(-1)
        terms = torch.empty_like(dist)
        _native.call("dbgsom_density_terms", self._p(dist), dist.numel(), float(sigma),
                     self._p(terms), self._stream())
        M = np.asarray(W).shape[0] if not torch.is_tensor(W) else W.shape[0]
        d = self._X.shape[1]
        sums = self._accumulate_dev(idx, terms, dist, M)   # K = density sums, a = hit counts
        tail = self._all_reduce(sums[M * d:M * d + 2 * M].clone()).cpu().numpy()
        return tail[M:2 * M].copy(), tail[:M].copy()

    def set_labels(self, y):
        super().set_labels(y)
        self._y_dev = None if y is None else self._torch.from_numpy(self._y).to(self.device)

    def _class_hist_dev(self, idx_dev, n_classes, M):
        torch = self._torch
        if getattr(self, "_y_dev", None) is None:
            raise RuntimeError("class histogram requested but no labels attached (set_labels)")
        hist = torch.empty((M, n_classes), dtype=torch.int64, device=self.device)
        _native.call("dbgsom_class_histogram", self._p(idx_dev), self._p(self._y_dev),
                     idx_dev.numel(), M, n_classes, self._p(hist), self._stream())
        return self._all_reduce(hist.double()).cpu().numpy().astype(np.int64)

    def class_histogram(self, winners, n_classes, M):
        idx = self._torch.from_numpy(np.ascontiguousarray(winners, dtype=np.int64)).to(self.device)
        return self._class_hist_dev(idx, n_classes, M)

    def _smooth(self, sums, W, hop, sigma, layout, keep_on_device=False):
        torch = self._torch
        M = W.shape[0] if torch.is_tensor(W) else np.asarray(W).shape[0]
        d = self._X.shape[1]  # padded feature count: what every device array carries
        Wd = getattr(self, "_W_dev", None)
        if Wd is None or tuple(Wd.shape) != (M, d):
            Wd, _ = self._as_dev_weights(W)
        hop_d = self._hop(hop)
        if tuple(hop_d.shape) != (M, M):
            raise ValueError("hop matrix must be (M, M)")
        Wn = torch.empty((M, d), dtype=torch.float64, device=self.device)
        chg = torch.empty(1, dtype=torch.float64, device=self.device)
        need = self._lib.dbgsom_smooth_workspace_bytes(M, d)
        ws = self._buf("smooth", need)
        self._timed_call("smooth", "dbgsom_smooth", self._p(sums), M, d, self._p(hop_d), float(sigma),
                     _native.LAYOUTS[layout], self._p(Wd), self._p(Wn), self._p(chg), self._p(ws),
                     ws.numel(), self._stream())
        # the epoch's small results come back in ONE round trip: three queued copies into pinned
        # host memory, one stream synchronisation (three blocking .cpu() / .item() calls left the
        # GPU idle for ~0.15 ms per epoch between them)
        host = self._pinned(3 * M + 2)
        host[:3 * M].copy_(sums[M * d:], non_blocking=True)
        host[3 * M:3 * M + 1].copy_(chg, non_blocking=True)
        has_status = "status" in self._ws
        if has_status:
            host[3 * M + 1:].view(torch.int32)[:1].copy_(self._ws["status"][:4].view(torch.int32),
                                                         non_blocking=True)
        counts = (self._queue_filter_counts()
                  if self._filtered_this_epoch and "filter" in self._ws else None)
        Wv = Wn if d == self._d else Wn[:, :self._d]  # (M, d) view: the padded columns are zeros
        Wout = Wv if keep_on_device else Wv.to("cpu", non_blocking=False).numpy()
        torch.cuda.current_stream(self.device).synchronize()
        self._counts_host = None if counts is None else counts.numpy().view(np.uint32)
        tail = host.numpy()
        if has_status and int(host[3 * M + 1:].view(torch.int32)[0]):
            raise _native.DbgsomNativeError("dbgsom_accumulate", -5, "winner index out of range")
        self._W_dev = None
        return Wout, float(tail[3 * M]), tail[2 * M:3 * M].copy(), tail[M:2 * M].copy()

    def _pinned(self, n, dtype=None):
        """Cached pinned host buffer of n elements (float64 unless given; D2H staging)."""
        dtype = dtype or self._torch.float64
        buf = self._ws_host.get((n, dtype)) if hasattr(self, "_ws_host") else None
        if buf is None:
            if not hasattr(self, "_ws_host"):
                self._ws_host = {}
            buf = self._torch.empty(n, dtype=dtype).pin_memory()
            self._ws_host[(n, dtype)] = buf
        return buf

    def _queue_filter_counts(self):
        """Queue the D2H copy of the last filtered search's candidate-list lengths (what the
        filter policy looks at) on the stream: it comes back with the epoch's other small results
        instead of in a round trip of its own (~0.1 ms of idle GPU per epoch)."""
        N, d = self._X.shape
        nb = (N + 127) // 128
        buf = self._pinned(nb, self._torch.int32)
        _native.call("dbgsom_bmu_filtered_counts_async", self._p(self._ws["filter"]), N, d,
                     self._last_filter_M, buf.data_ptr(), nb, self._stream())
        return buf

    def release(self):
        self._X = self._xx = self._hop_dev = None
        self._reset_filter_state()
        self._ws.clear()
        if hasattr(self, "_ws_host"):
            self._ws_host.clear()
