import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
from dbgsom_amd.backend import HipBackend
for name in ("c4", "c3"):
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
    hip.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True)
    c = hip.filter_counts().astype(np.int64)
    print(name, "n", len(c), "<=16", (c <= 16).sum(), "17-32", ((c > 16) & (c <= 32)).sum(), "33-48", ((c > 32) & (c <= 48)).sum(),
          "49-96", ((c > 48) & (c <= 96)).sum(), ">96", (c > 96).sum(), "max", c.max())
    # position of long lists in block order
    long_ = np.flatnonzero(c > 48)
    print("  long-list workgroups at (fraction of grid):", np.round(long_[-10:] / len(c), 3))
    hip.release(); del X
