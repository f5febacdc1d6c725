"""Timeline of the 4-wavefront sweep from in-kernel s_memtime stamps (experiment build
exp/libdbgsom_s4.so, SWEEP_EXPERIMENT=1024; run with DBGSOM_LIB pointing at it): cycles per tile in
[first half: products + reads + DMA issue | own-DMA wait | barrier | second half] per wavefront."""
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
hip = HipBackend(0, algorithm="filtered")
hip.sweep_planes = 1
X = bench.make_shard(torch, n, d, seed, dev)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().contiguous()
ww = hip._norms(W, _native.F64, M, d)
for _ in range(2):
    dist, idx = hip._bmu_filtered_dev(W, ww, 0)
torch.cuda.synchronize()
lib = _native.load()
lib.dbgsom_debug_ulist_offset.restype = ctypes.c_size_t
lib.dbgsom_debug_ulist_offset.argtypes = [ctypes.c_int64] * 3
off = lib.dbgsom_debug_ulist_offset(n, d, M)
nb = (n + 127) // 128
Mpad = (M + 511) // 512 * 512
ws = hip._ws["filter"]
rows_u16 = ws[off: off + nb * Mpad * 2].cpu().numpy().view(np.uint16).reshape(nb, Mpad)
dbg = np.ascontiguousarray(rows_u16[:, 512: 512 + 480]).view(np.uint32).view(np.int32).astype(np.int64)
for w in range(4):
    st = dbg[:, w * 60: w * 60 + 60].reshape(nb, 10, 6)[:, :, :5]
    dt = np.diff(st, axis=2)
    per_tile = st[:, 1:, 0] - st[:, :-1, 0]
    loop = st[:, 1:, 0] - st[:, :-1, 4]
    ok = (per_tile > 0).all(axis=1) & (per_tile < 100000).all(axis=1)
    print(f"{name} wave {w}: cycles per tile median {np.median(per_tile[ok]):.0f} (p10 {np.percentile(per_tile[ok], 10):.0f}, "
          f"p90 {np.percentile(per_tile[ok], 90):.0f}); median [first half {np.median(dt[ok][:, :, 0]):.0f} | dma wait "
          f"{np.median(dt[ok][:, :, 1]):.0f} | barrier {np.median(dt[ok][:, :, 2]):.0f} | second half {np.median(dt[ok][:, :, 3]):.0f}"
          f" | epilogue/loop {np.median(loop[ok]):.0f} (mean {loop[ok].mean():.0f})]; means [{dt[ok][:, :, 0].mean():.0f} | "
          f"{dt[ok][:, :, 1].mean():.0f} | {dt[ok][:, :, 2].mean():.0f} | {dt[ok][:, :, 3].mean():.0f}]")
