#!/usr/bin/env python
"""Benchmark of the batch-SOM hot path on MI355X: samples/sec/epoch (BMU + update).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c3|c2]

One "step" = one epoch of the hot path (reference dbgsom/BaseSom.py:403-407: BMU search, sample
kernel, per-neuron sums, [all-reduce], neighbourhood smoothing, convergence norm, per-neuron
error) on a frozen rectangular map, the samples already resident in HBM.  Each step feeds the
previous step's new prototypes back in, as training does.

For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL);
samples are sharded by rows (each rank generates its own shard: weak scaling, per-GPU work
fixed) and the only collective is the all-reduce of the [S|K|a|E] sums, once per epoch.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (samples per GPU, features, lattice rows, cols, seed, BASELINE.json config)
    "c4": (1_000_000, 784, 32, 32, 1004, "Synthetic N=1e6 d=784 fp32, M=1024 (32x32)"),
    "c3": (1_000_000, 128, 45, 45, 1003, "Synthetic Gaussian blobs N=1e6 d=128 fp32, M=2025 (45x45)"),
    "c2": (60_000, 784, 22, 23, 1002, "Fashion-MNIST stand-in 60k x 784 fp32, M=506 (22x23)"),
}
F64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 2.4 GHz x 2048 flop / 64 cycles (v_mfma_f64_16x16x4_f64)


def lattice_hops(rows, cols):
    ii, jj = np.divmod(np.arange(rows * cols), cols)
    return (np.abs(ii[:, None] - ii[None]) + np.abs(jj[:, None] - jj[None])).astype(np.float64)


def make_shard(torch, n, d, seed, device):
    """Gaussian blobs: 32 centres ~ N(0, 16 I), unit noise; generated in HBM."""
    gen = torch.Generator(device=device).manual_seed(seed)
    centers = torch.randn(32, d, device=device, generator=gen) * 4.0
    X = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 100_000
    for s in range(0, n, step):  # chunked: no N x d float64 temporaries
        m = min(step, n - s)
        lab = torch.randint(0, 32, (m,), device=device, generator=gen)
        X[s:s + m] = centers[lab] + torch.randn(m, d, device=device, generator=gen)
    return X


def cpu_baseline(workload, Xs, W, hop, sigma, gamma, n_full):
    """The reference CPU path (oracle port: sklearn NearestNeighbors + NumPy) on a bounded row
    sample, extrapolated to the full N: t = (t_bmu + t_acc) * N / Ns + t_smooth."""
    from oracle import som_oracle as o

    try:
        import sklearn  # noqa: F401

        bmu, engine = o.bmu_sklearn, "sklearn NearestNeighbors.kneighbors"
    except ImportError:
        bmu, engine = o.bmu_blas, "NumPy dgemm expanded-L2"
    M = W.shape[0]
    # size the sample for ~15 s of CPU work: probe the BMU rate on 10k rows first
    tp = time.perf_counter()
    bmu(Xs[:10_000], W, 1)
    rate = 10_000 / (time.perf_counter() - tp)
    ns = int(min(Xs.shape[0], max(20_000, rate * 15.0)))
    Xs = Xs[:ns]
    t0 = time.perf_counter()
    dist, win = bmu(Xs, W, 1)
    t1 = time.perf_counter()
    kw = o.exp_similarity_gamma(dist, gamma)
    S, K, a, E = o.accumulate_numpy(Xs, win, kw, dist, M)
    t2 = time.perf_counter()
    C = o.voronoi_centers(S, K, a, "compact")
    Wn = o.smooth_matmul(o.gaussian_neighborhood(hop, sigma), a, C)
    o.change_total(W, Wn)
    t3 = time.perf_counter()
    t_epoch = (t2 - t0) * (n_full / ns) + (t3 - t2)
    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = None
    return {
        "value": n_full / t_epoch,
        "unit": "samples/s/epoch",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": (f"{ns} of {n_full} rows ({engine}, f32 X / f64 W as the reference runs it; "
                   f"CSR-matmul sums; matmul smoothing), bmu {t1 - t0:.2f}s acc {t2 - t1:.2f}s "
                   f"smooth {t3 - t2:.2f}s, BLAS threads {blas_threads}; "
                   "extrapolated t=(bmu+acc)*N/Ns+smooth"),
    }


def fine_phase_regime(torch, td, args, X, ctl, M, d, hop, gamma, local, world, sync):
    """Secondary measurement: epochs of the FINE training phase (BaseSom.py:395-396, 899-900:
    constant sigma_end = max(0.7, 0.05 sqrt(M)), no growth) on a map that a decaying-sigma
    warm-up has organised, prototypes evolving from step to step as in training.  Timed for the
    exact search and for the filtered search (int8-MFMA candidate filter seeded with the previous
    epoch's winners + exact float64 on the candidates; identical results).  Uses
    centres_layout="aligned": with the reference's compacted centre rows (quirk Q1) a 32x32 map
    with dead neurons scrambles itself into near-duplicate prototypes (the headline regime above
    does exactly that), which is not what a trained map looks like."""
    from dbgsom_amd.backend import HipBackend

    sig0, sig1 = 0.2 * np.sqrt(M), max(0.7, 0.05 * np.sqrt(M))
    schedule = [sig1 + (sig0 - sig1) * np.exp(-0.35 * e) for e in range(14)]
    out = {"sigma": sig1, "warmup_epochs": len(schedule), "centres_layout": "aligned"}
    for algo in ("exact", "filtered"):
        be = HipBackend(local, algorithm=algo)
        be.load_device(X)
        W = ctl[:M * d].reshape(M, d).clone()
        for s_ in schedule:  # untimed: organise the map
            W = be.epoch(W, hop, s_, gamma, "aligned", False, keep_on_device=True).new_weights_dev
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = be.epoch(W, hop, sig1, gamma, "aligned", False, keep_on_device=True)
            W = res.new_weights_dev
        sync()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=X.device)
        if td.is_initialized():
            td.all_reduce(el, op=td.ReduceOp.MAX)
        el = float(el.item())
        out[algo] = {"ms_per_step": el / args.steps * 1e3,
                     "value": X.shape[0] * world * args.steps / el,
                     "dead_neurons": int((res.activations == 0).sum())}
        if algo == "filtered":
            c = be.filter_counts()
            out[algo]["candidates_per_workgroup"] = {"mean": float(c.mean()),
                                                     "p90": float(np.percentile(c, 90)),
                                                     "max": int(c.max())}
            out[algo]["checksum_equal_to_exact"] = bool(
                abs(float(W.sum().item()) - out["_wsum"]) == 0.0)
        else:
            out["_wsum"] = float(W.sum().item())
        be.release()
    out.pop("_wsum", None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--samples-per-gpu", type=int, default=None, help="override N per GPU")
    ap.add_argument("--fine-phase", type=int, default=1,
                    help="also time a trained map in the fine phase (sigma_end, evolving W) with "
                         "the exact and the filtered search (0 disables)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000,
                    help="upper bound of rows timed by the CPU baseline (0 disables it); the "
                         "actual sample is sized for ~15 s of CPU work")
    args = ap.parse_args()

    import torch
    import torch.distributed as td

    from dbgsom_amd.backend import HipBackend

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                  f"`python -m torch.distributed.run --nproc-per-node {args.gpus} ...`",
                  file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # under torch.distributed.run the group is always created (also for one rank: that run is
    # the single-GPU rehearsal of the RCCL path)
    grouped = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or "GROUP_RANK" in os.environ
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    n_gpu, d, rows, cols, seed, cfg_name = WORKLOADS[args.workload]
    if args.samples_per_gpu:
        n_gpu = args.samples_per_gpu
    M = rows * cols
    # headline: the exact all-pairs float64 search -- its time does not depend on the data or on
    # anything a previous step left behind
    hip = HipBackend(local, algorithm="exact")
    X = make_shard(torch, n_gpu, d, seed + rank, device)
    hip.load_device(X)

    # frozen map: M rows of rank 0's shard, Manhattan hop distances, epoch-0 sigma, gamma = 1/var
    ctl = torch.zeros(M * d + 1, dtype=torch.float64, device=device)
    if rank == 0:
        g = torch.Generator(device=device).manual_seed(seed + 7)
        sel = torch.randperm(n_gpu, device=device, generator=g)[:M]
        ctl[:M * d] = X[sel].double().reshape(-1)
        ctl[M * d] = 1.0 / X.double().var(dim=0, unbiased=False).sum()
    if grouped:
        td.broadcast(ctl, 0)
    W = ctl[:M * d].reshape(M, d).cpu().numpy()
    gamma = float(ctl[M * d].item())
    hop = lattice_hops(rows, cols)
    sigma = 0.2 * np.sqrt(M)  # BaseSom.py:876 at epoch 0

    def sync():
        if grouped:
            td.barrier()
        torch.cuda.synchronize()

    hip.kernel_events = None
    # prototypes stay in HBM between steps (a training phase without growth); the per-neuron
    # errors, hit counts and the convergence norm come back to the host every step
    for _ in range(args.warmup):
        W = hip.epoch(W, hop, sigma, gamma, "compact", False, keep_on_device=True).new_weights_dev
    hip.kernel_events = []  # HIP events around the BMU and accumulate launches (their stream)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        W = hip.epoch(W, hop, sigma, gamma, "compact", False, keep_on_device=True).new_weights_dev
    sync()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if grouped:
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
    elapsed = float(tmax.item())

    ev = hip.kernel_events
    fine = None
    if args.fine_phase:
        fine = fine_phase_regime(torch, td, args, X, ctl, M, d, hop, gamma, local, world, sync)
    bmu_ms = float(np.mean([a.elapsed_time(b) for (k, a, b) in ev if k == "bmu"]))
    acc_ms = float(np.mean([a.elapsed_time(b) for (k, a, b) in ev if k == "accumulate"]))
    smooth_ms = float(np.mean([a.elapsed_time(b) for (k, a, b) in ev if k == "smooth"]))
    hip.kernel_events = None

    if rank == 0:
        total = n_gpu * world
        flops = 2.0 * n_gpu * M * d  # algorithmic flops of one BMU launch
        achieved = flops / (bmu_ms * 1e-3) / 1e12
        out = {
            "metric": "samples/sec/epoch (BMU+update)",
            "value": total * args.steps / elapsed,
            "unit": "samples/s/epoch",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": cfg_name, "samples_per_gpu": n_gpu, "features": d,
                       "prototypes": M, "x_storage": "f32", "sharding": f"rows/{world}"},
            "roofline": {"bound": "mfma", "kernel": "bmu_dma_kernel<float,1>", "achieved": achieved,
                         "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "kernel_ms": bmu_ms},
            "phases_ms": {"bmu": bmu_ms, "accumulate": acc_ms, "smooth": smooth_ms,
                          "accumulate_GBps": n_gpu * d * 4 / (acc_ms * 1e-3) / 1e9},
        }
        if args.fine_phase:
            out["fine_phase"] = fine
        if args.cpu_sample > 0:
            ns = min(args.cpu_sample, n_gpu)
            Xs = X[:ns].cpu().numpy()
            W0 = ctl[:M * d].reshape(M, d).cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(args.workload, Xs, W0, hop, sigma, gamma, n_gpu)
            out["gpu_vs_cpu"] = (n_gpu * args.steps / elapsed) / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if grouped:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
