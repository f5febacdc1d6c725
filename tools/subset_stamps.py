"""Timeline of the exact-on-candidates kernel from in-kernel s_memtime stamps (experiment build
exp/libdbgsom_stamp.so, SUBSET_EXPERIMENT=128): per tile, cycles spent in [own-DMA wait | barrier |
DMA issue | reads + matrix products]."""
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
ww = hip._norms(W, _native.F64, M, d)
for _ in range(2):
    dist, idx = hip._bmu_filtered_dev(W, ww, 0)
torch.cuda.synchronize()
nb = n // 128
st = dist.cpu().numpy().reshape(-1)[: nb * 128].reshape(nb, 128)
cnt = st[:, 65]
for lo, hi in ((17, 32), (33, 48)):
    sel = st[(cnt >= lo) & (cnt <= hi)][:, :65].reshape(-1, 13, 5)
    dt = np.diff(sel, axis=2)                       # wait, barrier, issue, compute
    nxt = sel[:, 1:, 0] - sel[:, :-1, 4]            # end of tile -> top of the next
    per_tile = sel[:, 1:, 0] - sel[:, :-1, 0]
    print(f"lists {lo}..{hi}: {len(sel)} blocks; cycles per tile median {np.median(per_tile):.0f} "
          f"(p10 {np.percentile(per_tile, 10):.0f}, p90 {np.percentile(per_tile, 90):.0f}); "
          f"median [dma wait {np.median(dt[:, :, 0]):.0f} | barrier {np.median(dt[:, :, 1]):.0f} | "
          f"issue {np.median(dt[:, :, 2]):.0f} | reads+products {np.median(dt[:, :, 3]):.0f} | loop {np.median(nxt):.0f}]"
          f" mean [{dt[:, :, 0].mean():.0f} | {dt[:, :, 1].mean():.0f} | {dt[:, :, 2].mean():.0f} | {dt[:, :, 3].mean():.0f}]")

t0, t1 = st[:, 66], st[:, 67]
print('ticks of s_memtime per 10 ns of s_memrealtime: %.2f' % np.median(st[:, 69] / np.maximum(t1 - t0, 1)))
print('block phases in s_memtime ticks: prologue median %.0f, tile loop %.0f, tail %.0f' % (np.median(st[:, 70]), np.median(st[:, 71] - st[:, 70]), np.median(st[:, 69] - st[:, 71])))
base = t0.min()
for lo, hi in ((17, 32), (33, 1000)):
    m = (cnt >= lo) & (cnt <= hi)
    a, b = t0[m] - base, t1[m] - base
    k0, k1 = a.min(), b.max()
    dur = b - a
    # blocks resident over time
    ev = np.concatenate([np.stack([a, np.ones_like(a)], 1), np.stack([b, -np.ones_like(b)], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    lvl = np.cumsum(ev[:, 1])
    tt = ev[:, 0]
    avg = (lvl[:-1] * np.diff(tt)).sum() / (k1 - k0)
    print(f"lists {lo}..{hi}: kernel span {k1 - k0:.0f} ticks, block duration median {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}; "
          f"mean resident blocks {avg:.0f} (of {256 * (4 if hi == 32 else 3)} slots); first start {k0:.0f}; last start {a.max():.0f}")
    q = np.linspace(k0, k1, 11)
    prof = [int(lvl[np.searchsorted(tt, x, side='right') - 1]) for x in q[:-1] + np.diff(q) / 2]
    print("   resident blocks over the span (10 bins):", prof)
