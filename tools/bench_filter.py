"""Timing of the filtered search with a FIXED map (hint = the same winners): floor of sweep+subset."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dbgsom_amd.backend import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
n, d, rows, cols, seed, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0, algorithm=os.environ.get("ALGO", "filtered"))
X = bench.make_shard(torch, n, d, seed, dev)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().contiguous()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
ts = []
for e in range(7):
    hip.kernel_events = []
    hip.epoch(W, hop, 6.4, gamma, "compact", False, keep_on_device=True)
    torch.cuda.synchronize()
    ts.append({k: a.elapsed_time(b) for (k, a, b) in hip.kernel_events}["bmu"])
c = hip.filter_counts()
print(f"DBG={os.environ.get('DBGSOM_SWEEP_DBG','0')} bmu ms per epoch: {[round(t,2) for t in ts]} cand mean {c.mean():.1f} max {c.max()}")
