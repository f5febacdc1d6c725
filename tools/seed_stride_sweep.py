import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
from dbgsom_amd import _native
from dbgsom_amd.backend import HipBackend
name = sys.argv[1]
n, d, rows, cols, seed, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, seed, dev)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().contiguous()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
for stride in (2, 4, 6, 8, 12, 16):
    hip = HipBackend(0, algorithm="filtered"); hip.seed_stride = stride
    hip.load_device(X)
    _native.call("dbgsom_filter_timing", 1)
    for _ in range(4):
        hip.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 5)(); _native.call("dbgsom_bmu_filtered_stage_ms", ms)
    c = hip.filter_counts()
    print(name, "stride", stride, [round(float(v), 3) for v in ms], "bmu total %.3f" % sum(ms), "cand mean %.1f" % c.mean(), flush=True)
    hip.release()
