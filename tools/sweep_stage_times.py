"""Stage times of the filtered BMU search on the bench workload (frozen map), for kernel experiments:
    DBGSOM_LIB=exp/libdbgsom_exp4.so python tools/sweep_stage_times.py c4
(the experiment builds compute garbage: only `prepass` / `sweep` times mean anything there)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd import _native  # noqa: E402
from dbgsom_amd.backend import HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
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
_native.call("dbgsom_filter_timing", 1)
for _ in range(4):
    try:
        hip.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True)
    except _native.DbgsomNativeError as e:  # experiment builds produce invalid winners
        err = str(e)
torch.cuda.synchronize()
ms = (ctypes.c_double * 5)()
_native.call("dbgsom_bmu_filtered_stage_ms", ms)
c = hip.filter_counts()
err = globals().get("err", "")
c = np.asarray(c)
pad = np.where(c <= 16, 16, np.where(c <= 32, 32, (c + 47) // 48 * 48))
print("classes: <=16 %d  17..32 %d  33..48 %d  >48 %d (max %d);  padded candidates/WG %.1f -> exact-stage MFMA floor %.3f ms at 78.6 TF"
      % ((c <= 16).sum(), ((c > 16) & (c <= 32)).sum(), ((c > 32) & (c <= 48)).sum(), (c > 48).sum(), c.max(),
         pad.mean(), pad.sum() * 128.0 * d * 2 / 78.6e12 * 1e3))
print(os.environ.get("DBGSOM_LIB", "default"), name,
      dict(zip(("slice_w", "prepass", "sort", "sweep", "exact"), [round(float(v), 3) for v in ms])),
      "cand mean %.1f" % c.mean(), err)
