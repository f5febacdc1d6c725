"""ctypes binding of ``libdbgsom_hip.so`` (C ABI: ``include/dbgsom_hip.h``).

The product path has NO CPU fallback: :func:`load` raises when the HIP library is missing, and
``dbgsom_amd.backend.HipBackend`` raises when no MI355X is visible.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# DBGSOM_LIB: load another build of the same ABI (kernel experiments); default = the in-tree build
LIB_PATH = os.environ.get("DBGSOM_LIB") or os.path.join(CSRC, "libdbgsom_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "dbgsom_hip.h")

F32, F64, BF16 = 0, 1, 2
ALG_AUTO, ALG_EXACT, ALG_FILTERED, ALG_FILTERED_HINT = 0, 1, 2, 3
ALGORITHMS = {"auto": ALG_AUTO, "exact": ALG_EXACT, "filtered": ALG_FILTERED,
              "filtered_hint": ALG_FILTERED_HINT}
EPOCH_FROZEN = 1
ABI_VERSION = 4
# int (*)(void *user, double *buf_dev, int64_t count, void *stream)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                ctypes.c_void_p)
# dbgsom_collective_fn(user, op, buf_dev, count, stream): op 0 all-reduce, 1 reduce-scatter, 2 all-gather
COLLECTIVE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                 ctypes.c_void_p)
COLL_ALLREDUCE, COLL_REDUCE_SCATTER, COLL_ALLGATHER = 0, 1, 2
CENTRES_COMPACT, CENTRES_ALIGNED = 0, 1
LAYOUTS = {"compact": CENTRES_COMPACT, "aligned": CENTRES_ALIGNED}
MAX_PROTOTYPES = 16000

_i64, _vp, _ci, _dbl, _sz = (ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                             ctypes.c_size_t)

# name -> (restype, argtypes); every symbol declared in include/dbgsom_hip.h
SIGNATURES = {
    "dbgsom_abi_version": (_ci, []),
    "dbgsom_last_error": (ctypes.c_char_p, []),
    "dbgsom_device_count": (_ci, [ctypes.POINTER(_ci)]),
    "dbgsom_row_sqnorms": (_ci, [_vp, _ci, _i64, _i64, _i64, _vp, _vp]),
    "dbgsom_bmu": (_ci, [_vp, _ci, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _ci, _ci, _vp, _vp, _vp]),
    "dbgsom_exp_similarity": (_ci, [_vp, _i64, _dbl, _vp, _vp]),
    "dbgsom_accumulate_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dbgsom_accumulate": (_ci, [_vp, _ci, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                _sz, _vp]),
    "dbgsom_smooth_workspace_bytes": (_sz, [_i64, _i64]),
    "dbgsom_smooth": (_ci, [_vp, _i64, _i64, _vp, _dbl, _ci, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dbgsom_filter_planes_bytes": (_sz, [_i64, _i64]),
    "dbgsom_filter_prepare": (_ci, [_vp, _ci, _i64, _i64, _i64, _vp, _sz, _vp]),
    "dbgsom_bmu_filtered_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dbgsom_bmu_filtered": (_ci, [_vp, _ci, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                  _ci, _ci, _ci, _vp, _vp, _vp, _sz, _vp]),
    "dbgsom_bmu_filtered_counts": (_ci, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dbgsom_bmu_filtered_counts_async": (_ci, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dbgsom_bmu_filtered_refine_counts": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "dbgsom_sweep_shape": (_ci, [_i64, _i64]),
    "dbgsom_filter_timing": (_ci, [_ci]),
    "dbgsom_bmu_filtered_stage_ms": (_ci, [_vp]),
    "dbgsom_sum_workspace_bytes": (_sz, []),
    "dbgsom_sum_f64": (_ci, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "dbgsom_column_sums": (_ci, [_vp, _ci, _i64, _i64, _i64, _vp, _vp, _vp]),
    "dbgsom_topographic_count": (_ci, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "dbgsom_density_terms": (_ci, [_vp, _i64, _dbl, _vp, _vp]),
    "dbgsom_class_histogram": (_ci, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dbgsom_ctx_create": (_ci, [_ci, ctypes.POINTER(_vp)]),
    "dbgsom_ctx_destroy": (_ci, [_vp]),
    "dbgsom_ctx_set_option": (_ci, [_vp, ctypes.c_char_p, _i64]),
    "dbgsom_ctx_get_option": (_ci, [_vp, ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "dbgsom_ctx_stream": (_ci, [_vp, ctypes.POINTER(_vp)]),
    "dbgsom_ctx_load": (_ci, [_vp, _vp, _ci, _i64, _i64, _ci]),
    "dbgsom_ctx_load_device": (_ci, [_vp, _vp, _ci, _i64, _i64, _i64]),
    "dbgsom_ctx_read_samples": (_ci, [_vp, _vp, _i64, _vp]),
    "dbgsom_ctx_set_labels": (_ci, [_vp, _vp, _i64]),
    "dbgsom_ctx_set_topology": (_ci, [_vp, _vp, _i64]),
    "dbgsom_ctx_set_allreduce": (_ci, [_vp, _vp, _vp]),
    "dbgsom_ctx_set_collectives": (_ci, [_vp, _vp, _vp, _ci, _ci]),
    "dbgsom_rccl_unique_id": (_ci, [_vp]),
    "dbgsom_rccl_comm_init": (_ci, [_vp, _ci, _ci, ctypes.POINTER(_vp)]),
    "dbgsom_rccl_comm_destroy": (_ci, [_vp]),
    "dbgsom_ctx_set_rccl": (_ci, [_vp, _vp]),
    "dbgsom_ctx_allreduce_host": (_ci, [_vp, _vp, _i64]),
    "dbgsom_ctx_set_weights": (_ci, [_vp, _vp, _i64]),
    "dbgsom_ctx_get_weights": (_ci, [_vp, _ci, _vp, _i64]),
    "dbgsom_ctx_read_weight_rows": (_ci, [_vp, _ci, _vp, _i64, _vp]),
    "dbgsom_ctx_write_weight_rows": (_ci, [_vp, _i64, _i64, _vp]),
    "dbgsom_ctx_bmu": (_ci, [_vp, _vp, _i64, _ci, _ci, _vp, _vp]),
    "dbgsom_ctx_bmu_query": (_ci, [_vp, _vp, _ci, _i64, _i64, _vp, _i64, _ci, _ci, _vp, _vp]),
    "dbgsom_ctx_exp_similarity": (_ci, [_vp, _vp, _i64, _dbl, _vp]),
    "dbgsom_ctx_epoch": (_ci, [_vp, _vp, _i64, _ci, _dbl, _dbl, _ci, _ci, _vp, _vp, _vp, _vp, _vp,
                               _vp]),
    "dbgsom_ctx_update": (_ci, [_vp, _vp, _i64, _vp, _vp, _vp, _dbl, _ci, _vp, _vp, _vp, _vp]),
    "dbgsom_ctx_set_hint": (_ci, [_vp, _vp, _i64]),
    "dbgsom_ctx_read_sums": (_ci, [_vp, _vp, _i64]),
    "dbgsom_ctx_column_sums": (_ci, [_vp, _vp, _vp]),
    "dbgsom_ctx_quantization_error": (_ci, [_vp, _vp, _i64, _ci, _vp]),
    "dbgsom_ctx_topographic_count": (_ci, [_vp, _vp, _i64, _ci, _vp, _vp]),
    "dbgsom_ctx_node_statistics": (_ci, [_vp, _vp, _i64, _ci, _dbl, _vp, _vp]),
    "dbgsom_ctx_class_histogram": (_ci, [_vp, _vp, _i64, _i64, _vp]),
    "dbgsom_ctx_partition": (_ci, [_vp, _vp, _i64, _ci, _vp, _vp]),
    "dbgsom_ctx_subset_create": (_ci, [_vp, _i64, ctypes.POINTER(_vp)]),
    "dbgsom_ctx_epoch_info": (_ci, [_vp, _vp]),
    "dbgsom_ctx_arm_ms": (_ci, [_vp, _vp]),
    "dbgsom_ctx_filter_counts": (_ci, [_vp, _vp, _i64]),
    "dbgsom_ctx_refine_counts": (_ci, [_vp, _vp]),
    "dbgsom_ctx_phase_ms": (_ci, [_vp, _vp]),
}

_lib = None
_hip_rt = None


class DbgsomNativeError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, fn, code, msg):
        super().__init__(f"{fn} failed ({code}): {msg}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``dbgsom_amd/csrc/libdbgsom_hip.so`` (hipcc
    cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4", "libdbgsom_hip.so"], stdout=out)
    return LIB_PATH


def _load_hip_runtime():
    """Make ONE HIP runtime visible process-wide before the library (linked with -no-hip-rt)
    is loaded: the copy PyTorch bundles when torch is installed (so torch's streams and
    allocations are valid in our launches), else the system ROCm one."""
    candidates = []
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            candidates.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    except (ImportError, ValueError):  # pragma: no cover
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    errors = []
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            rt = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:  # try the next candidate
            errors.append(f"{path}: {e}")
            continue
        # the RCCL that belongs to THIS runtime (dbgsom_ctx_set_rccl resolves librccl at run time): a
        # librccl built against another libamdhip64 would bring a second HIP runtime into the process
        rccl = os.path.join(os.path.dirname(path), "librccl.so") if os.path.isabs(path) else ""
        if rccl and os.path.exists(rccl):
            os.environ.setdefault("DBGSOM_RCCL_LIB", rccl)
        return rt
    raise RuntimeError("no HIP runtime (libamdhip64.so) could be loaded: " + "; ".join(errors))


def load():
    """Load the HIP library; raises (never falls back) when it is not built."""
    global _lib, _hip_rt
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C dbgsom_amd/csrc`). "
            "dbgsom_amd has no CPU fallback.")
    _hip_rt = _load_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.dbgsom_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI version mismatch: library {lib.dbgsom_abi_version()}, "
                           f"binding {ABI_VERSION} (rebuild: make -C dbgsom_amd/csrc)")
    _lib = lib
    return lib


def check(fn_name: str, rc: int) -> None:
    if rc != 0:
        msg = load().dbgsom_last_error()
        err = DbgsomNativeError(fn_name, rc, msg.decode(errors="replace") if msg else "")
        if rc == -1:
            raise ValueError(str(err))
        raise err


def call(fn_name: str, *args):
    check(fn_name, getattr(load(), fn_name)(*args))


def device_count() -> int:
    n = _ci(0)
    call("dbgsom_device_count", ctypes.byref(n))
    return n.value
