"""What the engine's search policy does epoch by epoch on a bench workload (frozen map):
arm (seeds, digit planes), mean candidate-list length, the counting-only pruning probe, re-seeding.
    python tools/policy_trace.py c4 [rows] [epochs]"""
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
if len(sys.argv) > 2:
    n = int(sys.argv[2])
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0, algorithm=os.environ.get("ALGO", "filtered"))
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip.set_weights(W)
for e in range(epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
    ms = (time.perf_counter() - t0) * 1e3
    info = hip.epoch_info()
    print(f"epoch {e:2d}: {ms:7.3f} ms  planes {int(info[2])}  full seeds {int(info[7])}  lists {info[1]:8.2f}  "
          f"probe {info[6]:8.2f}  next: planes {hip._get('planes_next')} seed_mode {hip._get('seed_mode')} "
          f"retry {hip._get('prune_retry')} hold {hip._get('plane_hold')}", flush=True)
if os.environ.get("GROW"):   # a growth step: one more lattice column of prototypes, the context carries on
    cols2 = cols + 1
    M2 = rows * cols2
    W2 = np.concatenate([hip.get_weights(0), X[torch.randperm(n, device=dev, generator=g)[:M2 - M]].double().cpu().numpy()])
    hop2 = bench.lattice_hops(rows, cols2)
    hip.set_weights(W2)
    for e in range(int(os.environ["GROW"])):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.epoch(RESIDENT, hop2, 0.2 * np.sqrt(M2), gamma, "compact", False, keep_on_device=True, frozen=True)
        ms = (time.perf_counter() - t0) * 1e3
        info = hip.epoch_info()
        print(f"grown {M2} epoch {e:2d}: {ms:7.3f} ms  planes {int(info[2])}  full seeds {int(info[7])}  lists {info[1]:8.2f}  "
              f"probe {info[6]:8.2f}  next: planes {hip._get('planes_next')} seed_mode {hip._get('seed_mode')}", flush=True)
