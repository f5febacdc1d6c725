"""CPU emulation: how long would the candidate lists of the exact stage be if they were kept per
128 / 64 / 32 / 16 samples (bucket order by seed)?  One-product sweep bound, stateless seeds (every
4th prototype) and perfect seeds (true winners: the fine-phase regime).
usage: python tools/list_granularity.py [c4|c3|c2|c5] [n_samples]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests.test_filter_bound import filter_eps, slice_rows  # noqa: E402

F = 127.0 * 65536.0
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
_, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
X = bench.make_shard_numpy(n, d, seed, kind)
if name in bench.BF16_WORKLOADS:
    import torch
    X = torch.from_numpy(X).to(torch.bfloat16).float().numpy()
W = X[np.random.default_rng(seed + 7).choice(n, M, replace=False)].astype(np.float64)
(w0, _, _), tw, l1w = slice_rows(W)
yy = (W ** 2).sum(1)
ctab = 2.0 * tw * 65536.0 / (F * F)
rt = np.empty((n, M))
eps = np.empty(n)
w0f = w0.astype(np.float64)
for s in range(0, n, 20000):
    Xb = X[s:s + 20000].astype(np.float64)
    (x0, _, _), sx, l1x = slice_rows(Xb)
    xx = (Xb ** 2).sum(1)
    T = (x0.astype(np.float64) @ w0f.T) * 65536.0
    rt[s:s + 20000] = (xx[:, None] + yy[None]) - sx[:, None] * (ctab[None] * T)
    eps[s:s + 20000] = filter_eps(sx, l1x, xx, l1w.max(), tw.max(), yy.max(), d, 1)
true_win = rt.argmin(1)
for label, seeds in (("stateless (every 4th prototype)", rt[:, ::4].argmin(1) * 4), ("previous winners", true_win)):
    order = np.argsort(seeds, kind="stable")
    thr = rt[np.arange(n), seeds] + 2 * eps
    cand = rt[order] <= thr[order][:, None]
    per_sample = cand.sum(1)
    print(f"{name} {label}: per-sample candidates mean {per_sample.mean():.1f} p90 {np.percentile(per_sample, 90):.0f}")
    for gsz in (128, 64, 32, 16):
        ng = n // gsz
        u = cand[: ng * gsz].reshape(ng, gsz, M).any(1).sum(1)
        pad16 = np.ceil(u / 16) * 16
        print(f"   groups of {gsz:3d}: list mean {u.mean():6.1f} p90 {np.percentile(u, 90):5.0f} max {u.max():4d}; "
              f"padded-to-16 mean {pad16.mean():6.1f}; f64 work vs all-pairs {pad16.mean() / M:.4f}")
