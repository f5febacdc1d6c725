"""Profiling driver: N epochs of the hot path at a bench workload, nothing else (no CPU baseline,
no JSON).  Used under rocprofv3 --pmc / --kernel-trace:

    rocprofv3 --pmc SQ_WAVE_CYCLES ... -d out --output-format csv -- python3 tools/run_epochs.py c4 2
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import RESIDENT, HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
if len(sys.argv) > 3:
    n = int(sys.argv[3])
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0)
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
if name in bench.BF16_WORKLOADS:
    X = X.to(torch.bfloat16)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
sel = torch.randperm(n, device=dev, generator=g)[:M]
W = X[sel].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip.set_weights(W)
for _ in range(steps):
    hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True)
torch.cuda.synchronize()
print("done", name, steps)
