"""How many (sample, prototype) pairs pass the sweep's marking test, against the size of the
per-workgroup union lists the exact stage evaluates (experiment build exp/libdbgsom_cnt.so,
SWEEP_EXPERIMENT=512; run with DBGSOM_LIB pointing at it)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd import _native  # noqa: E402
from dbgsom_amd.backend import HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
n, d, rows, cols, seed, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, seed, dev)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().contiguous()
lib = _native.load()
lib.dbgsom_debug_ulist_offset.restype = ctypes.c_size_t
lib.dbgsom_debug_ulist_offset.argtypes = [ctypes.c_int64] * 3
for planes in (1, 2, 3):
    hip = HipBackend(0, algorithm="filtered")
    hip.sweep_planes = planes
    hip.load_device(X)
    ww = hip._norms(W, _native.F64, M, d)
    dist, idx = hip._bmu_filtered_dev(W, ww, 0)
    torch.cuda.synchronize()
    off = lib.dbgsom_debug_ulist_offset(n, d, M)
    nb = (n + 127) // 128
    Mpad = (M + 511) // 512 * 512
    ws = hip._ws["filter"]
    rows_u16 = ws[off: off + nb * Mpad * 2].cpu().numpy().view(np.uint16).reshape(nb, Mpad)
    pairs = np.ascontiguousarray(rows_u16[:, 512:514]).view(np.uint32).reshape(-1).astype(np.int64)
    counts = hip.filter_counts().astype(np.int64)
    print(f"{name} planes {planes}: union list mean {counts.mean():.1f}; passing pairs per sample mean "
          f"{pairs.sum() / n:.2f} (per workgroup {pairs.mean():.0f} of {128 * counts.mean():.0f} evaluated)",
          flush=True)
    hip.release()
