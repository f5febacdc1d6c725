"""Hot-path backends: the object the estimator calls once per epoch.

``HipBackend`` is a thin caller of the context level of the C ABI (``include/dbgsom_hip.h``,
``dbgsom_ctx_*``): NumPy arrays in, NumPy arrays out.  Everything else -- device memory, feature
padding, bfloat16 storage, digit planes, the choice of BMU search, previous winners as seeds,
device-resident prototypes -- lives behind that boundary in ``csrc/engine.hip``.  PyTorch appears
in exactly one place: ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI) supplies the one
all-reduce of the per-prototype sums per epoch, plugged into the context as a callback.  There is
no CPU fallback: constructing a ``HipBackend`` without the built library or without a GPU raises.

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
    if isinstance(dt, str) and dt == "bf16":
        return _native.BF16
    if dt == np.float32:
        return _native.F32
    if dt == np.float64:
        return _native.F64
    raise ValueError(f"samples must be float32 or float64, got {dt}")


class _Resident:
    """Stands for "the prototypes resident in HBM" wherever a weight matrix is expected."""

    def __repr__(self):
        return "RESIDENT"


RESIDENT = _Resident()


class _DeviceArray:
    """A float64 vector in HBM as an object ``torch.as_tensor`` can alias (no copy)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 3, "strides": None}


_RCCL_COMMS = {}   # (device index, world, rank) -> ncclComm_t made by dbgsom_rccl_comm_init (process lifetime;
                    # a process group rebuilt with another size or rank gets a communicator of its own)


class HipBackend(HotPathBackend):
    """MI355X backend: one ``dbgsom_ctx`` (one GPU) per instance / process."""

    name = "hip"

    # the filtered search pays off once the all-pairs float64 work is large (mirrors engine.hip)
    FILTER_MIN_PROTOTYPES = 129
    FILTER_MAX_MEAN_CANDIDATES = 320
    FILTER_MIN_QUERY_ROWS = 32768
    # the PRIOR of the engine's search policy (engine.hip: adapt_arms): what an arm that has never been timed is
    # priced at; arms that have run clean are compared by the engine's clock (arm_ms()).  Mirrored here only for
    # the test of the prior on small inputs, where every epoch copies results to the host and nothing is timed.
    SWEEP_COST = {1: 0.35, 2: 1.0, 3: 1.96}
    LIST_COST = 12.5
    PRUNE_PASS_COST = 60.0

    def __init__(self, device: Optional[int] = None, algorithm: str = "auto", _ctx=None):
        """algorithm (all give IDENTICAL results):
          "exact"          all-pairs float64 MFMA search;
          "filtered"       stateless: coarse int8-MFMA pre-pass -> int8 candidate sweep -> exact
                           float64 search on the candidates; nothing from earlier epochs is used;
          "filtered_hint"  the same, but the previous epoch's winners replace the pre-pass when
                           they are available (training: they almost always still win);
          "auto"           "filtered_hint" with a back-off to "exact" while the candidate lists are
                           long (maps of near-duplicate prototypes).
        The filtered forms apply to 129 <= M <= 16000 prototypes and rows of up to 43690 features
        (float32, float64 or bfloat16-resident samples); otherwise the exact kernel runs."""
        self._lib = _native.load()  # raises when the extension is not built
        if algorithm not in _native.ALGORITHMS:
            raise ValueError("algorithm must be 'auto', 'exact', 'filtered' or 'filtered_hint'")
        self._init_args = (device, algorithm)
        n_dev = _native.device_count()
        if n_dev < 1:
            raise RuntimeError(
                "dbgsom_amd.HipBackend needs a visible AMD GPU (MI355X / gfx950); found "
                f"hipGetDeviceCount()={n_dev}. There is no CPU fallback in the product path.")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % n_dev
        self.device_index = int(device)
        self._ctx = ctypes.c_void_p()
        if _ctx is not None:
            self._ctx = _ctx            # adopted (a Voronoi subset made on the device)
        else:
            _native.call("dbgsom_ctx_create", self.device_index, ctypes.byref(self._ctx))
        self._set("algorithm", _native.ALGORITHMS[algorithm])
        self._loaded = _ctx is not None
        self._borrowed = None       # keeps an adopted device array alive
        self._hop_key = None
        self._cb = None
        self._cb_error = None
        self._last_M = 0
        self.filter_log = []        # (epoch kind, mean candidates, digit planes) of the last epochs
        self.phase_log = None       # bench hook: a list collects the per-epoch phase times (ms)
        if _ctx is not None:
            self._N, self._d = self._get("n_samples"), self._get("features")
            self._x_np_dtype = {_native.F32: np.dtype(np.float32), _native.F64: np.dtype(np.float64),
                                _native.BF16: "bf16"}[self._get("storage")]
        self._install_collective()

    # -- plumbing -------------------------------------------------------------------------------
    def _set(self, name, value):
        _native.call("dbgsom_ctx_set_option", self._ctx, name.encode(), int(value))

    def _get(self, name) -> int:
        v = ctypes.c_int64(0)
        _native.call("dbgsom_ctx_get_option", self._ctx, name.encode(), ctypes.byref(v))
        return int(v.value)

    def _call(self, fn, *args):
        self._cb_error = None
        try:
            _native.call(fn, *args)
        except _native.DbgsomNativeError:
            if self._cb_error is not None:
                raise self._cb_error
            raise

    algorithm = property(lambda self: {v: k for k, v in _native.ALGORITHMS.items()}[self._get("algorithm")],
                         lambda self, a: self._set("algorithm", _native.ALGORITHMS[a]))
    # digit planes of the candidate sweep: 0 = adaptive, 1 = one int8 digit product (coarsest
    # bound), 2 = three, 3 = six (tightest); results do not depend on it
    sweep_planes = property(lambda self: self._get("sweep_planes"),
                            lambda self, v: self._set("sweep_planes", v))
    # the stateless seed pre-pass looks at every seed_stride-th prototype (0 = library default)
    seed_stride = property(lambda self: self._get("seed_stride"),
                           lambda self, v: self._set("seed_stride", v))
    # per-sample refinement in front of the exact stage of the filtered search (filter.hip 2d)
    # 0 / False = off, 1 / True = on, 2 = by measurement (the default): the engine times the exact stage of
    # the first epochs of a map size with and without it and keeps the faster form
    refine = property(lambda self: self._get("refine"), lambda self, v: self._set("refine", int(v)))
    refined = property(lambda self: bool(self._get("refined")))   # what the last filtered search ran
    # with the refinement: the distance of a sample it decided is evaluated inside the sums kernel of the
    # epoch (one pass over the float rows for distance and sums) -- 1 / 0 (the engine's default: experimental, off)
    defer = property(lambda self: bool(self._get("defer")), lambda self, v: self._set("defer", int(bool(v))))
    # neighbourhood smoothing sharded over the ranks (reduce-scatter of column blocks of the sums, all-gather of
    # the new prototypes): 0 never, 1 whenever the collective can, 2 (default) on large maps
    shard_smooth = property(lambda self: self._get("shard_smooth"), lambda self, v: self._set("shard_smooth", int(v)))
    shard_epochs = property(lambda self: self._get("shard_epochs"))
    defer_epochs = property(lambda self: self._get("defer_epochs"))
    planes_cached = property(lambda self: bool(self._get("planes_cached")))
    padded_features = property(lambda self: self._get("padded_features"))

    @property
    def n_samples(self):
        return self._get("n_samples")

    def _require_loaded(self):
        if not self._loaded:
            raise RuntimeError("HipBackend: call load(X) first")

    # -- the one collective ---------------------------------------------------------------------
    def _install_collective(self):
        """Plug torch.distributed's all-reduce into the context when a process group is up (or
        DBGSOM_FORCE_COLLECTIVE=1 asks for the rehearsal of that path with one rank)."""
        rank, world = dist_info()
        force = os.environ.get("DBGSOM_FORCE_COLLECTIVE") == "1" and _group_is_up()
        if world == 1 and not force:
            _native.call("dbgsom_ctx_set_collectives", self._ctx, None, None, 0, 1)
            self._cb = None
            return
        import torch
        import torch.distributed as td

        dev = torch.device("cuda", self.device_index)
        on_device = td.get_backend() == "nccl"
        if on_device and os.environ.get("DBGSOM_COLLECTIVE", "rccl") != "callback":
            # RCCL driven by the library itself (dbgsom_ctx_set_rccl): no callback, no interpreter in the
            # epoch.  One communicator per process and device, made once (torch.distributed only carries
            # rank 0's 128-byte id to the other ranks) and shared by every context of this process.
            key = (self.device_index, world, rank)
            comm = _RCCL_COMMS.get(key, "untried")
            if comm == "untried":
                # (every rank goes through every step and the ranks then agree on the outcome: a communicator
                #  that came up on some ranks only must not be used by any)
                uid, err = ctypes.create_string_buffer(128), None
                if rank == 0:
                    try:
                        _native.call("dbgsom_rccl_unique_id", uid)
                    except Exception as e:  # noqa: BLE001
                        err = e
                box = [uid.raw if err is None else None]
                td.broadcast_object_list(box, src=0)
                comm = ctypes.c_void_p()
                if box[0] is not None:
                    try:
                        uid = ctypes.create_string_buffer(box[0], 128)
                        self._get("n_samples")   # (a context call: this thread is on the context's device)
                        _native.call("dbgsom_rccl_comm_init", uid, world, rank, ctypes.byref(comm))
                    except Exception as e:  # noqa: BLE001
                        err, comm = e, ctypes.c_void_p()
                else:
                    err = err or RuntimeError("rank 0 could not make an RCCL id")
                ok = torch.tensor([1 if (err is None and comm.value) else 0], device=dev, dtype=torch.int32)
                td.all_reduce(ok, op=td.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    _RCCL_COMMS[key] = comm
                else:
                    # the library could not drive RCCL itself in this process (its librccl is not the one of the
                    # HIP runtime in use, say): the same collectives through torch.distributed's communicator
                    import warnings
                    warnings.warn(f"dbgsom_amd: RCCL inside the library is unavailable ({err}); "
                                  "collectives go through torch.distributed")
                    _RCCL_COMMS[key] = comm = None
            if comm is not None:
                _native.call("dbgsom_ctx_set_rccl", self._ctx, comm)
                self._cb = None
                return
        cache = {}   # the context reuses its stream and (until the map grows) its buffers

        def collective(_user, op, ptr, count, stream):
            """dbgsom_collective_fn: all-reduce of `count` values, or -- for the smoothing sharded over the
            ranks -- reduce-scatter / all-gather in place over `world` blocks of `count` values."""
            try:
                ext = cache.get(("stream", stream))
                if ext is None:
                    ext = cache[("stream", stream)] = torch.cuda.ExternalStream(stream, device=dev)
                total = count if op == _native.COLL_ALLREDUCE else count * world
                with torch.cuda.stream(ext):
                    t = cache.get((ptr, total))
                    if t is None:
                        if len(cache) > 16:
                            cache.clear()
                            cache[("stream", stream)] = ext
                        t = cache[(ptr, total)] = torch.as_tensor(_DeviceArray(ptr, total), device=dev)
                    mine = t[rank * count:(rank + 1) * count] if op != _native.COLL_ALLREDUCE else None
                    if on_device:   # RCCL through torch.distributed, ordered on the context's stream
                        if op == _native.COLL_ALLREDUCE:
                            td.all_reduce(t, op=td.ReduceOp.SUM)
                        elif op == _native.COLL_REDUCE_SCATTER:
                            td.reduce_scatter_tensor(mine, t, op=td.ReduceOp.SUM)
                        else:
                            td.all_gather_into_tensor(t, mine.clone())
                    else:  # gloo and friends: through the host (tests: several ranks on one GPU)
                        h = t.cpu()
                        if op == _native.COLL_ALLREDUCE:
                            td.all_reduce(h, op=td.ReduceOp.SUM)
                            t.copy_(h)
                        elif op == _native.COLL_REDUCE_SCATTER:
                            # (gloo has no reduce-scatter: a reduce per block to its owner -- every block is
                            #  summed in one order, whoever owns it)
                            for r in range(world):
                                blk = h[r * count:(r + 1) * count]
                                td.reduce(blk, dst=r, op=td.ReduceOp.SUM)
                            mine.copy_(h[rank * count:(rank + 1) * count])
                        else:
                            parts = [torch.empty(count, dtype=h.dtype) for _ in range(world)]
                            td.all_gather(parts, h[rank * count:(rank + 1) * count].contiguous())
                            t.copy_(torch.cat(parts))
                        ext.synchronize()
                return 0
            except BaseException as e:  # nothing may propagate through the C frames
                self._cb_error = e
                return 1

        self._cb = _native.COLLECTIVE_FN(collective)
        _native.call("dbgsom_ctx_set_collectives", self._ctx, self._cb, None, rank, world)

    # -- a8: residency --------------------------------------------------------------------------
    def load(self, X, storage=None):
        """Upload the samples once.  `storage="bf16"` keeps them in HBM as bfloat16 (rounded to
        nearest even on the device; all arithmetic stays float64 on the exactly widened values --
        an extension, the reference has no bf16)."""
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):  # integer / half input: as check_array would
            X = X.astype(np.float64)
        code = _x_dtype_code(X.dtype)
        if X.ndim != 2 or X.shape[0] < 1:
            raise ValueError("X must be a non-empty 2-D array")
        if storage == "bf16":
            if X.dtype != np.float32:
                X = X.astype(np.float32)
                code = _native.F32
            st = _native.BF16
        elif storage in (None, "native"):
            st = code
        else:
            raise ValueError("storage must be None or 'bf16'")
        self._call("dbgsom_ctx_load", self._ctx, X.ctypes.data, code, X.shape[0], X.shape[1], st)
        self._after_load(X.shape, "bf16" if st == _native.BF16 else X.dtype)
        self._borrowed = None
        return self

    def load_device(self, X_dev):
        """Adopt samples that already live in HBM: anything with `data_ptr()`, `shape`, `stride()`
        and a float32 / float64 / bfloat16 dtype (a torch tensor; bench: generated on the device).
        The caller's writes must have completed; the array is borrowed when its rows are a
        multiple of 16 features, copied (padded) otherwise."""
        name = str(X_dev.dtype).split(".")[-1]
        code = {"float32": _native.F32, "float64": _native.F64, "bfloat16": _native.BF16}.get(name)
        if code is None or len(X_dev.shape) != 2 or X_dev.stride(1) != 1:
            raise ValueError("X_dev must be a 2-D float32/float64/bfloat16 array with unit column stride")
        N, d = int(X_dev.shape[0]), int(X_dev.shape[1])
        if type(X_dev).__module__.split(".")[0] == "torch":   # its producer may still be running
            import torch

            torch.cuda.current_stream(X_dev.device).synchronize()
        self._call("dbgsom_ctx_load_device", self._ctx, ctypes.c_void_p(X_dev.data_ptr()), code, N, d,
                   int(X_dev.stride(0)))
        self._after_load((N, d), {_native.F32: np.dtype(np.float32), _native.F64: np.dtype(np.float64),
                                  _native.BF16: "bf16"}[code])
        self._borrowed = X_dev
        return self

    def _after_load(self, shape, np_dtype):
        self._N, self._d = int(shape[0]), int(shape[1])
        self._x_np_dtype = np_dtype
        self._loaded = True
        self._hop_key = None
        self._last_M = 0
        self._y = None

    def read_samples(self, rows):
        """Rows of the resident samples as float64 (exactly widened)."""
        self._require_loaded()
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((rows.size, self._d))
        self._call("dbgsom_ctx_read_samples", self._ctx, rows.ctypes.data, rows.size, out.ctypes.data)
        return out

    # -- prototypes -----------------------------------------------------------------------------
    def _round_f32(self, W):
        # float32 samples AND float32 prototypes: the reference's engine returns float32-rounded
        # distances (epoch 0 of a float32 fit).  Every other mix is full float64.
        return int(W is not RESIDENT and not isinstance(self._x_np_dtype, str)
                   and self._x_np_dtype == np.float32 and np.asarray(W).dtype == np.float32)

    def _w_arg(self, W, d=None):
        """-> (keep-alive array, pointer or None, M, round_f32) for a weight argument."""
        if W is RESIDENT:
            M = self._get("prototypes")
            if M < 1:
                raise RuntimeError("no prototypes resident in HBM yet")
            return None, None, M, 0
        rf = self._round_f32(W) if d is None else 0
        W64 = np.ascontiguousarray(W, dtype=np.float64)
        if W64.ndim != 2 or W64.shape[1] != (self._d if d is None else d):
            raise ValueError("prototype / sample feature mismatch")
        return W64, W64.ctypes.data, W64.shape[0], rf

    def set_weights(self, W):
        self._require_loaded()
        keep, p, M, _ = self._w_arg(W)
        self._call("dbgsom_ctx_set_weights", self._ctx, p, M)
        return RESIDENT

    def get_weights(self, which=0):
        """which=0: the resident prototypes; 1: the other buffer (after an epoch: its input)."""
        M = self._get("prototypes") if which == 0 else self._last_M
        out = np.empty((M, self._d))
        self._call("dbgsom_ctx_get_weights", self._ctx, int(which), out.ctypes.data, M)
        return out

    def read_weight_rows(self, rows, which=0):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((rows.size, self._d))
        self._call("dbgsom_ctx_read_weight_rows", self._ctx, int(which), rows.ctypes.data, rows.size,
                   out.ctypes.data)
        return out

    def write_weight_rows(self, row0, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, self._d)
        self._call("dbgsom_ctx_write_weight_rows", self._ctx, int(row0), rows.shape[0], rows.ctypes.data)

    # -- a1 -------------------------------------------------------------------------------------
    def bmu(self, W, k=1, X=None):
        """-> (distances, winners) like BaseSom._get_winning_neurons: shape (N,) for k=1,
        (N, k) otherwise."""
        if X is None:
            self._require_loaded()
            keep, p, M, rf = self._w_arg(W)
            N = self._N
            idx = np.empty((N, k), dtype=np.int64)
            dist = np.empty((N, k), dtype=np.float64)
            self._call("dbgsom_ctx_bmu", self._ctx, p, M, int(k), rf, idx.ctypes.data, dist.ctypes.data)
        else:
            X = np.ascontiguousarray(X)
            if X.dtype not in (np.float32, np.float64):
                X = X.astype(np.float64)
            W = np.asarray(W)
            if X.ndim != 2 or W.ndim != 2 or W.shape[1] != X.shape[1]:
                raise ValueError("prototype / sample feature mismatch")
            rf = int(X.dtype == np.float32 and W.dtype == np.float32)
            W64 = np.ascontiguousarray(W, dtype=np.float64)
            N = X.shape[0]
            idx = np.empty((N, k), dtype=np.int64)
            dist = np.empty((N, k), dtype=np.float64)
            self._call("dbgsom_ctx_bmu_query", self._ctx, X.ctypes.data, _x_dtype_code(X.dtype), N,
                       X.shape[1], W64.ctypes.data, W64.shape[0], int(k), rf, idx.ctypes.data,
                       dist.ctypes.data)
        if k == 1:
            return dist.reshape(-1), idx.reshape(-1)
        return dist, idx

    def query_filter_applies(self, N, d, M, k=1):
        """Whether a k-BMU query on N other samples would go through the filtered search."""
        return (k == 1 and self.algorithm != "exact" and N >= self._get("filter_min_query_rows")
                and self.FILTER_MIN_PROTOTYPES <= M <= _native.MAX_PROTOTYPES and d <= 43690)

    # -- a2 -------------------------------------------------------------------------------------
    def exp_similarity(self, distances, gamma):
        dd = np.ascontiguousarray(np.asarray(distances).reshape(-1), dtype=np.float64)
        out = np.empty_like(dd)
        self._call("dbgsom_ctx_exp_similarity", self._ctx, dd.ctypes.data, dd.size, float(gamma),
                   out.ctypes.data)
        return out

    # -- a5 / a6 --------------------------------------------------------------------------------
    def _topology(self, hop, M):
        # the estimator hands over the SAME array object until the lattice changes
        if hop is not self._hop_key:
            h = np.ascontiguousarray(hop, dtype=np.float64)
            if h.shape != (M, M):
                raise ValueError("hop matrix must be (M, M)")
            self._call("dbgsom_ctx_set_topology", self._ctx, h.ctypes.data, M)
            self._hop_key = hop

    # -- the epoch ------------------------------------------------------------------------------
    def epoch(self, W, hop, sigma, gamma, layout="compact", want_assignments=False,
              keep_on_device=False, n_classes=0, frozen=False):
        """One hot-path epoch (the body of BaseSom._grow_som, BaseSom.py:403-407) as ONE call of
        the C ABI.  `W` is a NumPy array or RESIDENT (the prototypes the previous epoch left in
        HBM); with `keep_on_device` the new prototypes stay in HBM (no PCIe round trip between
        epochs of a phase without growth) and only the O(M) statistics come back; `frozen`
        leaves the resident prototypes untouched (bench: the same map every step)."""
        self._require_loaded()
        keep, p, M, rf = self._w_arg(W)
        self._topology(hop, M)
        Wn = None if keep_on_device else np.empty((M, self._d))
        chg = np.empty(1)
        E = np.empty(M)
        a = np.empty(M)
        win = np.empty(self._N, dtype=np.int64) if want_assignments else None
        dist = np.empty(self._N) if want_assignments else None
        self._call("dbgsom_ctx_epoch", self._ctx, p, M, rf, float(gamma), float(sigma),
                   _native.LAYOUTS[layout], _native.EPOCH_FROZEN if frozen else 0,
                   None if Wn is None else Wn.ctypes.data, chg.ctypes.data, E.ctypes.data, a.ctypes.data,
                   None if win is None else win.ctypes.data, None if dist is None else dist.ctypes.data)
        self._last_M = M
        res = EpochResult(Wn, float(chg[0]), E, a, win, dist, RESIDENT if keep_on_device else None)
        if n_classes > 0:
            res.class_hist = self.class_histogram(None, n_classes, M)
        self._log_epoch()
        return res

    def epoch_info(self):
        """dbgsom_ctx_epoch_info of the last epoch: [filtered (0/1), mean candidate-list length,
        digit planes (0 = no sweep), hinted (0/1), back-off epochs left, policy hold, mean list length
        of a counting-only pruning launch (NaN: none), full seed pre-pass (0/1)]."""
        info = (ctypes.c_double * 8)()
        _native.call("dbgsom_ctx_epoch_info", self._ctx, info)
        return [float(v) for v in info]

    def arm_ms(self):
        """dbgsom_ctx_arm_ms: {(seeds, planes): ms} of the arms of the search policy that have been timed."""
        ms = (ctypes.c_double * 12)()
        _native.call("dbgsom_ctx_arm_ms", self._ctx, ms)
        return {(i // 4, i % 4): float(v) for i, v in enumerate(ms) if v == v}

    def _log_epoch(self):
        info = self.epoch_info()
        if info[0]:
            self.filter_log.append(("filtered", float(info[1]), int(info[2])))
        elif self.algorithm != "exact":
            self.filter_log.append(("exact", None))
        del self.filter_log[:-64]
        if self.phase_log is not None:   # (needs the context option "timing")
            ms = (ctypes.c_double * 8)()
            _native.call("dbgsom_ctx_phase_ms", self._ctx, ms)
            self.phase_log.append([float(v) for v in ms])

    def update(self, W, hop, sigma, sample_weights, winners, distances, layout="compact"):
        """_update_weights + _write_accumulative_error with the caller's winners / sample weights
        (BaseSom.py:470-523, 541-561) -> (new_weights, change_total, errors, activations)."""
        self._require_loaded()
        keep, p, M, _ = self._w_arg(W)
        self._topology(hop, M)
        idx = np.ascontiguousarray(winners, dtype=np.int64)
        kw = np.ascontiguousarray(sample_weights, dtype=np.float64)
        dist = np.ascontiguousarray(distances, dtype=np.float64)
        if not (idx.size == kw.size == dist.size == self._N):
            raise ValueError("one winner / weight / distance per resident sample")
        Wn, chg, E, a = np.empty((M, self._d)), np.empty(1), np.empty(M), np.empty(M)
        self._call("dbgsom_ctx_update", self._ctx, p, M, idx.ctypes.data, kw.ctypes.data,
                   dist.ctypes.data, float(sigma), _native.LAYOUTS[layout], Wn.ctypes.data,
                   chg.ctypes.data, E.ctypes.data, a.ctypes.data)
        self._last_M = M
        return Wn, float(chg[0]), E, a

    def set_hint(self, winners, M):
        """Seeds of the next filtered search (any indices < M keep the result exact)."""
        idx = np.ascontiguousarray(winners, dtype=np.int64)
        if idx.size != self._N:
            raise ValueError("one seed per resident sample")
        self._call("dbgsom_ctx_set_hint", self._ctx, idx.ctypes.data, int(M))

    def read_sums(self, M):
        """The last epoch's reduced [S (M x d) | K | a | E] buffer."""
        out = np.empty(M * (self._d + 3))
        self._call("dbgsom_ctx_read_sums", self._ctx, out.ctypes.data, int(M))
        return out

    def filter_counts(self):
        """Candidate-list length per 128-sample workgroup of the last filtered search."""
        nb = (self._N + 127) // 128
        out = np.empty(nb, dtype=np.uint32)
        self._call("dbgsom_ctx_filter_counts", self._ctx, out.ctypes.data, nb)
        return out

    def refine_counts(self):
        """[(sample, prototype) pairs evaluated exactly, 128-sample workgroups refined, samples whose
        candidates overflowed the four slots] of the last filtered search."""
        out = np.zeros(4, dtype=np.uint64)
        self._call("dbgsom_ctx_refine_counts", self._ctx, out.ctypes.data)
        return [int(v) for v in out]

    def traffic(self):
        """PCIe traffic of the prototypes since the context was created / last released."""
        keys = ("w_upload_calls", "w_upload_bytes", "w_download_calls", "w_download_bytes",
                "w_row_writes", "w_row_reads")
        return {k: self._get(k) for k in keys}

    def plane_cost(self, p, mean, M):
        """Cost model of the engine's policy (engine.hip) for `p` digit planes of the candidate
        sweep and candidate lists of `mean` entries; p = 0: no sweep, candidates from the triangle
        inequality (one pass over the top digit plane of X and an M x M matrix of prototype gaps)."""
        if p == 0:
            launches = 25.0 / (2.8 * max(self._N, 1) * self.padded_features / (1.0e6 * 784.0))
            return (self.PRUNE_PASS_COST + self.SWEEP_COST[1] * M * 9.0 * M / max(self._N, 1) + launches
                    + self.LIST_COST * mean)
        return self.SWEEP_COST[p] * M + self.LIST_COST * mean

    # -- f-1 .. f-3: reductions that keep the N-sized arrays in HBM -----------------------------
    def column_moments(self):
        """(sum_i x_ij, sum_i (x_ij - mean_j)^2, N) over the resident samples in NumPy's axis-0
        arithmetic (sequential per column, X's dtype): np.var(X, 0) = s2 / N and
        np.std(X, 0, ddof=1) = sqrt(s2 / (N - 1)) bit for bit, without a host pass over X.
        None when the resident dtype has no NumPy counterpart (bfloat16)."""
        self._require_loaded()
        if isinstance(self._x_np_dtype, str):
            return None
        s1 = np.empty(self._d, dtype=self._x_np_dtype)
        self._call("dbgsom_ctx_column_sums", self._ctx, None, s1.ctypes.data)
        mean = np.ascontiguousarray(np.true_divide(s1, self._N))
        s2 = np.empty_like(s1)
        self._call("dbgsom_ctx_column_sums", self._ctx, mean.ctypes.data, s2.ctypes.data)
        return s1, s2, self._N

    def quantization_error(self, W) -> float:
        self._require_loaded()
        keep, p, M, rf = self._w_arg(W)
        out = np.empty(2)
        self._call("dbgsom_ctx_quantization_error", self._ctx, p, M, rf, out.ctypes.data)
        return float(out[0] / out[1])

    def topographic_error_count(self, W, coords) -> int:
        self._require_loaded()
        keep, p, M, rf = self._w_arg(W)
        xy = np.ascontiguousarray(coords, dtype=np.int32)
        if xy.shape != (M, 2):
            raise ValueError("coords must be (M, 2)")
        out = np.empty(1)
        self._call("dbgsom_ctx_topographic_count", self._ctx, p, M, rf, xy.ctypes.data, out.ctypes.data)
        return int(round(out[0]))

    def node_statistics(self, W, sigma):
        """-> (hit_counts (M,), density_sums (M,)) of BaseSom._calculate_node_statistics."""
        self._require_loaded()
        keep, p, M, rf = self._w_arg(W)
        hits, dens = np.empty(M), np.empty(M)
        self._call("dbgsom_ctx_node_statistics", self._ctx, p, M, rf, float(sigma), hits.ctypes.data,
                   dens.ctypes.data)
        return hits, dens

    def set_labels(self, y):
        super().set_labels(y)
        if y is None:
            self._call("dbgsom_ctx_set_labels", self._ctx, None, 0)
        else:
            self._call("dbgsom_ctx_set_labels", self._ctx, self._y.ctypes.data, self._y.size)

    def class_histogram(self, winners, n_classes, M):
        """(M, n_classes) int64 over all ranks; winners=None: the last epoch's (still in HBM)."""
        hist = np.empty((M, n_classes), dtype=np.int64)
        idx = None if winners is None else np.ascontiguousarray(winners, dtype=np.int64)
        self._call("dbgsom_ctx_class_histogram", self._ctx, None if idx is None else idx.ctypes.data,
                   int(n_classes), int(M), hist.ctypes.data)
        return hist

    # -- vertical growth on Voronoi subsets (f-4) -------------------------------------------------
    def partition(self, W, want_winners=False):
        """BMU of every resident sample + bucket order -> (samples per neuron, winners | None)."""
        self._require_loaded()
        keep, p, M, rf = self._w_arg(W)
        counts = np.empty(M, dtype=np.int64)
        win = np.empty(self._N, dtype=np.int64) if want_winners else None
        self._call("dbgsom_ctx_partition", self._ctx, p, M, rf, counts.ctypes.data,
                   None if win is None else win.ctypes.data)
        return counts, win

    def subset(self, neuron):
        """A backend whose resident samples are the Voronoi set of `neuron` (gathered in HBM)."""
        child = ctypes.c_void_p()
        self._call("dbgsom_ctx_subset_create", self._ctx, int(neuron), ctypes.byref(child))
        return HipBackend(self.device_index, self.algorithm, _ctx=child)

    _SETTABLE = ("algorithm", "sweep_planes", "seed_stride", "timing", "graph", "refine", "defer",
                 "filter_min_query_rows", "max_mean_candidates", "shard_smooth")

    def release(self):
        """Give the device memory back (the backend can be loaded again afterwards; options stay)."""
        if self._ctx:
            opts = {k: self._get(k) for k in self._SETTABLE}
            _native.call("dbgsom_ctx_destroy", self._ctx)
            self._ctx = ctypes.c_void_p()
            _native.call("dbgsom_ctx_create", self.device_index, ctypes.byref(self._ctx))
            for k, v in opts.items():
                self._set(k, v)
            self._install_collective()
        self._loaded = False
        self._borrowed = None
        self._hop_key = None

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.dbgsom_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:  # interpreter shutdown
            pass
