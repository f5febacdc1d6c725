"""Profiling driver: N frozen-map epochs (bench workload) with a chosen BMU algorithm, nothing else.
    rocprofv3 --pmc FETCH_SIZE -d out --output-format csv -- python3 tools/run_frozen.py c4 3 filtered"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import RESIDENT, HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
algo = sys.argv[3] if len(sys.argv) > 3 else "filtered"
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0, algorithm=algo)
take = 0
for opt in sys.argv[4:]:   # e.g. refine=1 defer=0 sweep_planes=4; rows=125000: a rank's share of the workload
    k, v = opt.split("=")
    if k == "rows":
        take = int(v)
    else:
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
for _ in range(steps):
    hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
torch.cuda.synchronize()
print("done", name, steps, algo)
