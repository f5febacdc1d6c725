"""Where the exact stage's time goes, workgroup by workgroup: an experiment build with -DEXACT_TIMELINE=1 leaves every
workgroup's start / first tile / last push / end (s_memrealtime, 100 MHz), the shader clock over its life
(s_memtime) and the CU it ran on in the distances.
    bash tools/build_variant.sh xt filter -DEXACT_TIMELINE=1
    DBGSOM_LIB=exp_build/libdbgsom_xt.so python tools/exact_timeline.py c4
(EXACT_TIMELINE=2 | 3 in the environment: the per-entry sums of the walk of tools/exact_stage_walk.patch, whose
experiment builds carry those stamp modes.)"""
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
hip.refine = 0
X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
hip.load_device(X)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
hop = bench.lattice_hops(rows, cols)
hip.set_weights(W)
for _ in range(4):
    res = hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", True, keep_on_device=True, frozen=True)
v = res.distances[res.distances < -0.5]
code = (-v - 1.0).astype(np.uint64)
field = (code & np.uint64(7)).astype(int)
val = ((code >> np.uint64(3)) & np.uint64(0x3fffffff)).astype(np.int64)
key = (code >> np.uint64(33)).astype(np.int64)
wgs = {}
for k_, f_, v_ in zip(key, field, val):
    wgs.setdefault(int(k_), {})[int(f_)] = int(v_)
full = [w for w in wgs.values() if len(w) == 8]
print(f"{name}: {len(wgs)} workgroups stamped, {len(full)} complete")
start = np.array([w[0] for w in full]); end = np.array([w[1] for w in full])
cnt = np.array([w[2] >> 8 for w in full]); cls = np.array([(w[2] >> 4) & 15 for w in full])
cu = np.array([w[3] for w in full])   # XCC << 16 | HW_ID[15:0] (wave, simd, pipe, cu, sh, se)
first = np.array([w[4] for w in full]); pushed = np.array([w[5] for w in full])
cyc = (np.array([w[7] for w in full]) - np.array([w[6] for w in full])) % (1 << 30)   # s_memtime ticks of the life
t0 = start.min()
W30 = 1 << 30
rel = lambda a: ((a - t0) % W30) / 100.0   # us
s_us, e_us, f_us, p_us = rel(start), rel(end), rel(first), rel(pushed)
span = e_us.max()
print(f"span of the stage {span:.1f} us; s_memtime ticks per us of a workgroup's life: {np.median(cyc / np.maximum(e_us - s_us, 1e-9)):.1f} (the shader clock while the stage runs, MHz)")
for c in (3, 2, 1):
    m = cls == c
    if not m.any():
        continue
    dur = e_us[m] - s_us[m]
    steps = np.ceil(cnt[m] / (16 * c))
    tiles = steps * (d // 16)
    print(f"class {c}: {m.sum():5d} workgroups, lists {cnt[m].mean():6.1f}, life {dur.mean():7.1f} us (p10 {np.percentile(dur, 10):.1f}, p90 {np.percentile(dur, 90):.1f}), "
          f"until the first tile {np.mean(f_us[m] - s_us[m]):5.1f}, behind the last push {np.mean(e_us[m] - p_us[m]):5.1f}, "
          f"per k-tile {np.mean((p_us[m] - f_us[m]) / tiles):.3f} us; first start {s_us[m].min():.1f}, last start {s_us[m].max():.1f}, last end {e_us[m].max():.1f}")
if os.environ.get("EXACT_TIMELINE") in ("2", "3"):   # the walk: sums over a workgroup's entries
    ne = np.array([w[7] for w in full], dtype=float)
    names_ = (("arm -> first tile landed", 4), ("first tile -> last push", 5), ("last push -> next arm", 6)) if os.environ.get("EXACT_TIMELINE") == "2" else \
        (("arm begins -> arm ends", 4), ("-> own DMAs of tile 0 landed", 5), ("-> barrier passed", 6))
    for nm, f_ in names_:
        v_ = np.array([w[f_] for w in full]) / 100.0
        print(f"  {nm:26s}: {np.sum(v_) / np.sum(ne):7.2f} us per entry ({np.mean(v_):7.1f} us per workgroup)")
    print(f"  entries per workgroup {ne.mean():.1f} (min {ne.min():.0f}, max {ne.max():.0f})")
# per CU: busy union, last end
cuid = (cu >> 16) * 4096 + ((cu >> 8) & 0xf) + 16 * ((cu >> 13) & 7) + 128 * ((cu >> 12) & 1)   # xcc, cu_id, se_id, sh_id
ids = np.unique(cuid)
last = np.array([e_us[cuid == i].max() for i in ids]); firsts = np.array([s_us[cuid == i].min() for i in ids])
print(f"{ids.size} CUs seen; a CU's last workgroup ends {np.mean(span - last):.1f} us before the stage does on average (p50 {np.percentile(span - last, 50):.1f}, p90 {np.percentile(span - last, 90):.1f}); "
      f"its first starts at {firsts.mean():.1f} us")
# resident workgroups / wavefronts over time
bins = np.linspace(0, span, 21)
for a, b in zip(bins[:-1], bins[1:]):
    mid = 0.5 * (a + b)
    act = (s_us <= mid) & (e_us > mid)
    waves = sum(((cls == c) & act).sum() * (8 if c == 3 else 4) for c in (1, 2, 3))
    print(f"  t = {mid:7.1f} us: resident workgroups {act.sum():5d} (class 3/2/1: {((cls == 3) & act).sum()}/{((cls == 2) & act).sum()}/{((cls == 1) & act).sum()}), wavefronts per CU {waves / ids.size:.1f}")
# per XCD: the hardware deals workgroup k to XCD k % 8 whatever their pace
xcc_ = (cu >> 16)
for x in np.unique(xcc_):
    m = xcc_ == x
    print(f"  XCD {x}: {m.sum():5d} workgroups, life {np.mean(e_us[m] - s_us[m]):7.2f} us, last start {s_us[m].max():7.1f}, last end {e_us[m].max():7.1f}")
