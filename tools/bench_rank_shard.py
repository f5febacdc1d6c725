"""What rank r > 0 of a multi-GPU bench run sees, rehearsed on one GPU: prototypes drawn from rank
0's rows, samples = rank r's shard of the same data set.  Prints the epoch time per rank.
    python tools/bench_rank_shard.py c4 0 1 7"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
ranks = [int(r) for r in sys.argv[2:]] or [0, 1]
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
X0 = bench.make_shard(torch, n, d, seed, dev)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X0[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X0.double().var(dim=0, unbiased=False).sum().item())
del X0
hop = bench.lattice_hops(rows, cols)
for r in ranks:
    X = bench.make_shard(torch, n, d, seed, dev, rank=r)
    if name in bench.BF16_WORKLOADS:
        X = X.to(torch.bfloat16)
    hip = HipBackend(0, algorithm="filtered")
    hip.load_device(X)
    for _ in range(3):
        hip.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        hip.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    c = hip.filter_counts()
    print(f"{name} rank {r}: {ms:.3f} ms/epoch, candidates per workgroup mean {c.mean():.1f} max {c.max()}",
          flush=True)
    hip.release()
    del X
