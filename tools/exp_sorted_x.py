"""Experiment (no product change): how much of an epoch is the GATHER?  The same frozen epochs on the
workload's samples as they come and on the same samples stored in the order of their winners (every bucket of
the search and every neuron's chunk of the sums is then a contiguous run of rows).
    python tools/exp_sorted_x.py c4"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import RESIDENT, HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)


def run(Xd, label):
    if name in bench.BF16_WORKLOADS:
        Xd = Xd.to(torch.bfloat16)
    hip = HipBackend(0, algorithm="filtered")
    hip.load_device(Xd)
    hip.set_weights(W)
    hip._set("timing", 1)
    r = hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", True, keep_on_device=True, frozen=True)
    for _ in range(30):
        hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    ph = hip.phase_ms() if hasattr(hip, "phase_ms") else None
    print(f"{name} {label}: median {np.median(ts):.3f} ms/epoch (min {min(ts):.3f})", ph, flush=True)
    hip.release()
    return r.winners


win = run(X, "as generated")
order = torch.from_numpy(np.argsort(win, kind="stable")).to(dev)
run(X[order].contiguous(), "rows in the order of their winners")
