"""Soak run: 3000 chained epochs on resident prototypes, then repeated fits (leaks, drift, NaNs).
    python tools/soak.py [workload = c2] [epochs = 3000] [verify]
verify: the same chain once more with the all-pairs search; the final prototypes must be identical
bit for bit (every arm of the policy, hinted seeds, probes and re-seeding on an evolving map)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
from dbgsom_amd.backend import RESIDENT, HipBackend
from dbgsom_amd import SomVQ
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n_epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
verify = len(sys.argv) > 3 and sys.argv[3] == "verify"
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip = HipBackend(0, algorithm="auto"); hip.load_device(X)
if "refine" in sys.argv[4:]:   # the per-sample refinement in every filtered search (default: by measurement)
    hip.refine = 1
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
t0 = time.time(); hip.set_weights(W); b0 = hip._get("device_bytes")
arms = {}
for e in range(n_epochs):
    hip.epoch(RESIDENT, hop, max(0.7, 4.0 * 0.999 ** e), gamma, "aligned", False, keep_on_device=True)
    key = hip.filter_log[-1][0] if hip.filter_log[-1][0] != "filtered" else f"filtered/{hip.filter_log[-1][2]}" + ("+refined" if hip.refined else "")
    arms[key] = arms.get(key, 0) + 1
Wd = hip.get_weights(0)
print("searches run:", arms)
if verify:
    ex = HipBackend(0, algorithm="exact"); ex.load_device(X); ex.set_weights(W)
    for e in range(n_epochs):
        ex.epoch(RESIDENT, hop, max(0.7, 4.0 * 0.999 ** e), gamma, "aligned", False, keep_on_device=True)
    We = ex.get_weights(0)
    print("identical to the all-pairs chain:", bool(np.array_equal(Wd, We, equal_nan=True)))
    ex.release()
    assert np.array_equal(Wd, We, equal_nan=True)
print(n_epochs, "chained epochs", round(time.time() - t0, 2), "s; context device MB before/after", b0 >> 20, hip._get("device_bytes") >> 20, "finite", bool(np.isfinite(Wd).all()))
hip.release()
Xh = X[:20000].cpu().numpy()
t0 = time.time()
for r in range(5):
    est = SomVQ(random_state=r, max_neurons=300, n_iter=60).fit(Xh)
print("5 fits", round(time.time() - t0, 2), "s", len(est.neurons_), "neurons, QE", round(est.quantization_error_, 4), "allocated MB", torch.cuda.memory_allocated() >> 20)
