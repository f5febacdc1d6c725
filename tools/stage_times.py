"""Per-stage HIP-event times of frozen-map epochs (bench workload), median over the steps; one JSON line.
    [DBGSOM_LIB=exp_build/libdbgsom_x.so] python tools/stage_times.py c4 [steps=12] [algo=filtered] [rows=N] [map=RxC] [opt=value ...]"""
import json
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
opts = dict(a.split("=") for a in sys.argv[2:])
steps = int(opts.pop("steps", 12))
algo = opts.pop("algo", "filtered")
take = int(opts.pop("rows", 0))
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
if "map" in opts:   # another lattice on the workload's samples: map=8x8
    rows, cols = (int(v) for v in opts.pop("map").split("x"))
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0, algorithm=algo)
for k, v in opts.items():
    hip._set(k, int(v))
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
if take:
    X = X[:take].contiguous()
    n = take
if name in bench.BF16_WORKLOADS:
    X = X.to(torch.bfloat16)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip.set_weights(W)


def step():
    hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)


for _ in range(6):
    step()
hip._set("timing", 0)
wall = []
for _ in range(steps):
    t = time.perf_counter()
    step()
    wall.append(time.perf_counter() - t)
hip._set("timing", 1)
hip.phase_log = []
for _ in range(steps):
    step()
ph = dict(zip(bench.PHASES, np.median(np.array(hip.phase_log), axis=0).round(4).tolist()))
out = {"workload": name, "lib": os.environ.get("DBGSOM_LIB", "in-tree"), "epoch_ms": round(float(np.median(wall)) * 1e3, 4),
       "phases_ms": ph, "last": hip.filter_log[-1] if hip.filter_log else None, "refined": bool(hip.refined)}
if hip.filter_log and hip.filter_log[-1][0] == "filtered":
    c = hip.filter_counts()
    out["lists_mean"] = round(float(c.mean()), 2)
print(json.dumps(out))
