"""Per-phase cycles of segsum_chain_kernel's workgroups from an experiment build with -DCHAIN_STAMPS=1
    bash tools/build_variant.sh cs accumulate -DCHAIN_STAMPS=1
    DBGSOM_LIB=exp_build/libdbgsom_cs.so python tools/chain_stamps.py c4"""
import os
import sys

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
hip = HipBackend(0, algorithm="filtered")
hip.refine, hip.defer = 1, 1
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
if name in bench.BF16_WORKLOADS:
    X = X.to(torch.bfloat16)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip.set_weights(W)
for _ in range(4):
    res = hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", True, keep_on_device=True, frozen=True)
v = res.distances[res.distances < -1e11]
k = np.floor((-v - 1e12) / 1e10).astype(int)
cyc = -v - 1e12 - 1e10 * k
names = ["set-up", "loads issued", "loads landed", "range 0 written", "chain", "distances", "sums", "tail",
         "(chain wave 0: loop)", "(chain wave 0: barrier)", "rows"]
tot = sum(np.mean(cyc[k == i]) for i in range(8))
for i in range(11):
    c = cyc[k == i]
    if c.size:
        print(f"{names[i]:16s} n={c.size:6d} mean {c.mean():10.0f}  p10 {np.percentile(c, 10):9.0f}  p90 {np.percentile(c, 90):9.0f}" +
              (f"  {100 * c.mean() / tot:5.1f} %" if i < 8 else ""))
print(f"sum of the phases per workgroup: {tot:.0f} cycles (s_memtime ticks)")
