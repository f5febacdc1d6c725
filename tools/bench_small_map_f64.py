"""float64-resident samples: exact BMU kernel time by map size (DBGSOM_BMU_PATH=generic forces the
register-staged kernel for comparison)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dbgsom_amd.backend import HipBackend
n, d = 500_000, 784
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, 1004, dev).double()
hip = HipBackend(0, algorithm="exact").load_device(X)
for M in (4, 32, 51, 64, 100, 128, 200, 1024):
    W = X[torch.randperm(n, device=dev)[:M]].contiguous()
    hop = np.abs(np.subtract.outer(np.arange(M), np.arange(M))).astype(np.float64)
    ts = []
    for e in range(4):
        hip.kernel_events = []
        r = hip.epoch(W, hop, 2.0, 1e-3, "compact", False, keep_on_device=True)
        torch.cuda.synchronize()
        ev = {k: a.elapsed_time(b) for (k, a, b) in hip.kernel_events}
        ts.append((ev["bmu"], ev["accumulate"]))
    b, a = np.median(np.array(ts), axis=0)
    print(f"path={os.environ.get('DBGSOM_BMU_PATH','auto')} M={M:4d}: bmu {b:7.3f} ms ({2.0 * n * M * d / b / 1e9:6.1f} TFLOP/s useful) accumulate {a:.3f} wsum {float(r.new_weights_dev.sum()):.6f}", flush=True)
