"""Phases of prune_mark_kernel's workgroups (experiment build with -DPRUNE_STAMPS=1: thread 0's s_memrealtime at the
phase boundaries, read back through dbgsom_experiment_prune_stamps):
    bash tools/build_variant.sh pms filter -DPRUNE_STAMPS=1
    DBGSOM_LIB=exp_build/libdbgsom_pms.so python tools/prune_timeline.py c3"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import RESIDENT, HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
hip = HipBackend(0, algorithm="filtered")
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
    hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", True, keep_on_device=True, frozen=True)
lib = ctypes.CDLL(os.environ["DBGSOM_LIB"])
nb = min((n + 127) // 128, 16384)
buf = np.zeros(8 * nb, dtype=np.uint64)
rc = lib.dbgsom_experiment_prune_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(8 * nb))
assert rc == 0, rc
raw = buf.reshape(nb, 8)
ok_ = (raw[:, :6] > 0).all(axis=1)
st = raw[ok_, :6].astype(np.int64)
xcc = (raw[ok_, 6] >> np.uint64(16)).astype(int)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
print(f"{name}: {len(us)} workgroups; span {us[:, 5].max():.1f} us; life {np.mean(us[:, 5] - us[:, 0]):.2f} us "
      f"(p10 {np.percentile(us[:, 5] - us[:, 0], 10):.2f}, p90 {np.percentile(us[:, 5] - us[:, 0], 90):.2f})")
for k, nm in enumerate(("ids and seeds", "rows read", "bounds", "marks (gap matrix)", "list written")):
    dt = us[:, k + 1] - us[:, k]
    print(f"  {nm:20s} {dt.mean():7.2f} us (p10 {np.percentile(dt, 10):.2f}, p90 {np.percentile(dt, 90):.2f})")
bins = np.linspace(0, us[:, 5].max(), 11)
for a, b in zip(bins[:-1], bins[1:]):
    mid = 0.5 * (a + b)
    print(f"  t = {mid:7.1f} us: resident workgroups {int(((us[:, 0] <= mid) & (us[:, 5] > mid)).sum())}")
# lives by start time (is the ragged end made of longer-lived workgroups?)
order_ = np.argsort(us[:, 0])
for dec in range(10):
    sel = order_[dec * len(us) // 10:(dec + 1) * len(us) // 10]
    lf = us[sel, 5] - us[sel, 0]
    ph = [np.mean(us[sel, k + 1] - us[sel, k]) for k in range(5)]
    print(f"  started {us[sel, 0].min():6.1f} .. {us[sel, 0].max():6.1f} us: life {lf.mean():6.2f} (max {lf.max():6.2f}); phases " + " ".join(f"{v:5.2f}" for v in ph))
# per XCD: workgroups, mean life, when its last workgroup ends
for x in (np.unique(xcc) if raw[ok_, 6].any() else []):
    m = xcc == x
    print(f"  XCD {x}: {m.sum():5d} workgroups, life {np.mean(us[m, 5] - us[m, 0]):6.2f} us, last start {us[m, 0].max():6.1f}, last end {us[m, 5].max():6.1f}")
