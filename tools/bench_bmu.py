"""A/B timing of the BMU kernel alone at a bench workload (C4 by default), one process per
variant because the knob is read once at first launch:  DBGSOM_BMU_PRIO=<mode> python tools/bench_bmu.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, d, rows, cols, seed, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0)
X = bench.make_shard(torch, n, d, seed, dev)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().contiguous()
ww = hip._norms(W, 1, M, d)
ts = []
for r in range(reps + 2):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    dist, idx = hip._bmu_dev(hip._X, hip._xx, hip._x_np_dtype, W, ww, 1, 0)
    b.record()
    torch.cuda.synchronize()
    if r >= 2:
        ts.append(a.elapsed_time(b))
ts = np.array(ts)
fl = 2.0 * n * M * d
print(f"PRIO={os.environ.get('DBGSOM_BMU_PRIO', 'default')} {name}: median {np.median(ts):.3f} ms  min {ts.min():.3f}  "
      f"-> {fl / np.median(ts) / 1e9:.1f} TFLOP/s  checksum {int(idx.sum().item())}")
