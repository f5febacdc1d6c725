"""Soak run: 3000 chained epochs on resident prototypes, then repeated fits (leaks, drift, NaNs)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
from dbgsom_amd.backend import RESIDENT, HipBackend
from dbgsom_amd import SomVQ
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS["c2"]
M = rows * cols
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, seed, dev)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip = HipBackend(0, algorithm="auto"); hip.load_device(X)
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
t0 = time.time(); hip.set_weights(W); b0 = hip._get("device_bytes")
for e in range(3000):
    hip.epoch(RESIDENT, hop, max(0.7, 4.0 * 0.999 ** e), gamma, "aligned", False, keep_on_device=True)
Wd = hip.get_weights(0)
print("3000 chained epochs", round(time.time() - t0, 2), "s; context device MB before/after", b0 >> 20, hip._get("device_bytes") >> 20, "finite", bool(np.isfinite(Wd).all()))
hip.release()
Xh = X[:20000].cpu().numpy()
t0 = time.time()
for r in range(5):
    est = SomVQ(random_state=r, max_neurons=300, n_iter=60).fit(Xh)
print("5 fits", round(time.time() - t0, 2), "s", len(est.neurons_), "neurons, QE", round(est.quantization_error_, 4), "allocated MB", torch.cuda.memory_allocated() >> 20)
