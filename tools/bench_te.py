"""Topographic-error count (k = 2 BMU + device reduction) on a bench workload: through the pruning form
of the filtered search (after training epochs that ran it) against the all-pairs kernel.
    python tools/bench_te.py c4"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dbgsom_amd.backend import RESIDENT, HipBackend
for name in sys.argv[1:] or ["c4"]:
    n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
    M = rows * cols
    dev = torch.device("cuda", 0)
    X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
    if name in bench.BF16_WORKLOADS:
        X = X.to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(seed + 7)
    W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
    gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
    hop = bench.lattice_hops(rows, cols)
    coords = np.stack(np.divmod(np.arange(M), cols), axis=1)
    out = {}
    for algo in ("filtered", "exact"):
        hip = HipBackend(0, algorithm=algo)
        hip.load_device(X)
        hip.set_weights(W)
        for _ in range(4):
            hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
        te = hip.topographic_error_count(RESIDENT, coords)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            te = hip.topographic_error_count(RESIDENT, coords)
        ms = (time.perf_counter() - t0) * 200
        out[algo] = te
        print(f"{name} {algo:9s} topographic count {te} in {ms:.3f} ms, k2 filtered {hip._get('k2_filtered')}", flush=True)
        hip.release()
    assert out["filtered"] == out["exact"]
