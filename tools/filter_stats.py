"""Candidate-list statistics of the filtered BMU search along a sigma-decay run.
usage: filter_stats.py <workload> <epochs> <layout> [decay]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
layout = sys.argv[3] if len(sys.argv) > 3 else "compact"
decay = float(sys.argv[4]) if len(sys.argv) > 4 else 0.35
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
sig0, sig1 = 0.2 * np.sqrt(M), max(0.7, 0.05 * np.sqrt(M))
for e in range(steps):
    sigma = sig1 + (sig0 - sig1) * np.exp(-decay * e)
    hip.kernel_events = []
    res = hip.epoch(W, hop, sigma, gamma, layout, False, keep_on_device=True)
    torch.cuda.synchronize()
    ms = {k: a.elapsed_time(b) for (k, a, b) in hip.kernel_events}
    W = res.new_weights_dev
    line = (f"epoch {e}: sigma {sigma:.2f} bmu {ms['bmu']:.2f} ms chg {res.change_total:.3e} "
            f"dead {(res.activations == 0).sum()}")
    if e >= 1:
        c = hip.filter_counts()
        line += f" | cand/workgroup mean {c.mean():.1f} median {np.median(c):.0f} p90 {np.percentile(c, 90):.0f} max {c.max()}"
    print(line, flush=True)
