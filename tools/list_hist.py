"""Histogram of the candidate-list lengths of a bench workload (stateless filtered epoch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dbgsom_amd.backend import RESIDENT, HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
if name in bench.BF16_WORKLOADS:
    X = X.to(torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip = HipBackend(0, algorithm="filtered")
hip.load_device(X)
hip.set_weights(W)
for _ in range(3):
    hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
c = hip.filter_counts()
print(name, "groups", c.size, "mean", c.mean(), "max", c.max())
edges = [0, 16, 32, 48, 64, 96, 128, 160, 192, 224, 253, 288, 320, 384, 448, 512, 1024, 100000]
h, _ = np.histogram(c, bins=edges)
for a, b, v in zip(edges[:-1], edges[1:], h):
    print(f"  ({a:5d}, {b:5d}]  {v:6d}  {100.0 * v / c.size:5.1f} %")
