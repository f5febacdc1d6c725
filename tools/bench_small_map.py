"""Exact BMU kernel time on small maps (the regime a growing fit lives in): N x d samples, M prototypes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dbgsom_amd.backend import HipBackend
n, d = 1_000_000, 784
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, 1004, dev)
hip = HipBackend(0, algorithm="exact").load_device(X)
hip._set("timing", 1)
for M in (4, 16, 32, 33, 51, 64, 65, 100, 128, 129, 200):
    W = X[torch.randperm(n, device=dev)[:M]].double().cpu().numpy()
    hop = np.abs(np.subtract.outer(np.arange(M), np.arange(M))).astype(np.float64)
    ts = []
    for e in range(5):
        hip.phase_log = []
        hip.epoch(W, hop, 2.0, 1e-3, "compact", False, keep_on_device=True)
        ts.append(tuple(hip.phase_log[-1][:3]))
    b, a, s = np.median(np.array(ts), axis=0)
    print(f"M={M:4d}: bmu {b:6.3f} ms ({2.0 * n * M * d / b / 1e9:7.1f} TFLOP/s useful)  accumulate {a:.3f}  smooth {s:.3f}", flush=True)
