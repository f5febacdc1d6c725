"""Host side of HipBackend.epoch: cProfile over frozen-map epochs (the GPU runs ~4 ms per epoch,
so everything except the final stream synchronisation should stay far below that)."""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n, d, rows, cols, seed, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0, algorithm="filtered")
X = bench.make_shard(torch, n, d, seed, dev)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().contiguous()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
sig = 0.2 * np.sqrt(M)
for _ in range(3):
    hip.epoch(W, hop, sig, gamma, "compact", False, keep_on_device=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    hip.epoch(W, hop, sig, gamma, "compact", False, keep_on_device=True)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
