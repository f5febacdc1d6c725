"""Candidate-list statistics of the filtered BMU search along a bench-like run."""
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
sig_scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
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
sigma = 0.2 * np.sqrt(M) * sig_scale
for e in range(steps):
    hip.kernel_events = []
    res = hip.epoch(W, hop, sigma, gamma, "compact", False, keep_on_device=True)
    torch.cuda.synchronize()
    ms = {k: a.elapsed_time(b) for (k, a, b) in hip.kernel_events}
    W = res.new_weights_dev
    line = f"epoch {e}: bmu {ms['bmu']:.2f} ms acc {ms['accumulate']:.2f} chg {res.change_total:.3e} dead {(res.activations == 0).sum()}"
    if e >= 1:
        c = hip.filter_counts()
        line += f" | cand/workgroup mean {c.mean():.1f} median {np.median(c):.0f} p90 {np.percentile(c, 90):.0f} max {c.max()}"
    print(line, flush=True)
