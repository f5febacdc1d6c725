"""Is the candidate sweep bound by the HBM gather of the X plane?  Times the filtered search's
stages with the real bucket order and with an order that only ever visits the first R samples
(their planes stay in L2 / MALL): same work for the matrix pipes, no HBM traffic for X.
(The results of the second run are meaningless -- timing only.)"""
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
hip = HipBackend(0, algorithm="filtered_hint")
hip.sweep_planes = 1
X = bench.make_shard(torch, n, d, seed, dev)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().contiguous()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
for _ in range(2):  # second epoch: hinted (previous winners + their bucket order)
    hip.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True)
ww = hip._norms(W, _native.F64, M, d)
_native.call("dbgsom_filter_timing", 1)


def stages(prev, order, label):
    X32, _ = hip._bmu_samples()
    for _ in range(3):
        hip._bmu_filtered_on(X32, hip._xx, hip._planes, W, ww, 0, hip._p(prev), hip._p(order), "filter")
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 5)()
    _native.call("dbgsom_bmu_filtered_stage_ms", ms)
    print(f"{name} {label}: sweep {ms[3]:.3f} ms, exact on candidates {ms[4]:.3f} ms", flush=True)


prev, order = hip._prev_idx, hip._order
stages(prev, order, "real bucket order          ")
for R in (8192, 65536):
    o2 = (order[:R].repeat((n + R - 1) // R))[:n].contiguous()
    stages(prev, o2, f"order visits {R:6d} samples")
