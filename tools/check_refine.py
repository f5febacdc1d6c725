"""Refinement on / off on the bench workloads: epoch time, per-stage times, refinement counters, and
bit-identity of winners / distances / new prototypes between the two (and the all-pairs search).
    python tools/check_refine.py c4 c3 [--exact] [--algo filtered|filtered_hint]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd.backend import RESIDENT, HipBackend  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
with_exact = "--exact" in sys.argv
algo = "filtered"
for a in sys.argv[1:]:
    if a.startswith("--algo="):
        algo = a.split("=", 1)[1]
dev = torch.device("cuda", 0)
bad = 0
for name in args or ["c2"]:
    n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
    M = rows * cols
    X = bench.make_shard(torch, n, d, seed, dev, 0, kind)
    if name in bench.BF16_WORKLOADS:
        X = X.to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(seed + 7)
    W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
    gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
    hop = bench.lattice_hops(rows, cols)
    sigma = 0.2 * np.sqrt(M)
    results = {}
    variants = [("refine=0", algo, 0), ("refine=1", algo, 1), ("refine=1 d1", algo, 1), ("refine=2", algo, 2)]
    if with_exact:
        variants.append(("exact", "exact", False))
    for label, alg, refine in variants:
        hip = HipBackend(0, algorithm=alg)
        hip.refine = refine
        hip.defer = label.endswith("d1")
        hip.load_device(X)
        hip.set_weights(W)
        for _ in range(4):
            hip.epoch(RESIDENT, hop, sigma, gamma, "compact", False, keep_on_device=True, frozen=True)
        torch.cuda.synchronize()
        steps = 3 if alg == "exact" else 20
        t0 = time.perf_counter()
        for _ in range(steps):
            hip.epoch(RESIDENT, hop, sigma, gamma, "compact", False, keep_on_device=True, frozen=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        hip._set("timing", 1)
        hip.phase_log = []
        for _ in range(3):
            res = hip.epoch(RESIDENT, hop, sigma, gamma, "compact", True, keep_on_device=False, frozen=True)
        ph = np.array(hip.phase_log).min(axis=0)
        info = hip.epoch_info()
        extra = ""
        if alg != "exact":
            c = hip.filter_counts()
            extra = f" lists mean {c.mean():.1f} max {c.max()} planes {int(info[2])}"
            extra += f" refined {int(hip.refined)}"
            if hip.refined:
                pairs, ok, ovf, nun = hip.refine_counts()
                extra += (f" | pairs/sample {pairs / n:.3f} refined {ok} of {(n + 127) // 128} overflow samples {ovf} "
                          f"distinct candidates per 64 samples {nun / ((n + 63) // 64):.1f}")
        print(f"{name} {label:11s} {ms:8.3f} ms/epoch  phases " + " ".join(f"{v:.3f}" for v in ph) + extra, flush=True)
        results[label] = (res.winners.copy(), res.distances.copy(), res.new_weights.copy())
        hip.release()
    ref = results["refine=0"]
    for label in results:
        if label == "refine=0":
            continue
        same = [np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, results[label])]
        print(f"{name} refine=0 vs {label}: winners {same[0]} distances {same[1]} new prototypes {same[2]}", flush=True)
        bad += not all(same)
    del X
sys.exit(1 if bad else 0)
