"""Randomised parity stress of the all-pairs BMU kernels (LDS-DMA widths, register-staged kernel,
float32 / float64 / bfloat16 samples, k = 1, 2) against the oracle's chain form."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dbgsom_amd.backend import HipBackend  # noqa: E402
from oracle import som_oracle as o  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(n_cases):
    N = int(rng.integers(1, 4000))
    d = int(rng.choice([1, 3, 16, 17, 32, 48, 64, 100, 128, 200, 784]))
    M = int(rng.choice([1, 2, 4, 5, 16, 31, 32, 33, 63, 64, 65, 96, 100, 128, 129, 200, 300]))
    dt = rng.choice(["f32", "f64", "bf16"])
    k = int(rng.choice([1, 2])) if M >= 2 else 1
    X = (rng.normal(size=(N, d)) * 3 if rng.random() < 0.7 else rng.integers(-2, 3, size=(N, d))).astype(
        np.float64 if dt == "f64" else np.float32)
    W = (rng.normal(size=(M, d)) * 3 if rng.random() < 0.7 else rng.integers(-2, 3, size=(M, d))).astype(np.float64)
    Xr = X
    if dt == "bf16":
        Xr = torch.from_numpy(X).to(torch.bfloat16).float().numpy()
    be = HipBackend(algorithm="exact").load(X, storage="bf16" if dt == "bf16" else None)
    dist, idx = be.bmu(W, k)
    rd, ri = o.bmu_chain(Xr, W, k)
    ok = np.array_equal(idx, ri) and np.array_equal(dist, rd)
    # the same through a query on non-resident samples
    d2, i2 = be.bmu(W, k, Xr[: max(1, N // 2)])
    ok = ok and np.array_equal(i2, ri[: max(1, N // 2)]) and np.array_equal(d2, rd[: max(1, N // 2)])
    print(f"case {case:3d} N={N:5d} d={d:4d} M={M:4d} k={k} {dt:4s} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
    be.release()
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
