#!/usr/bin/env python
"""Benchmark of the batch-SOM hot path on MI355X: samples/sec/epoch (BMU + update).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c3|c2|c5]

One "step" = one epoch of the hot path (reference dbgsom/BaseSom.py:403-407: BMU search, sample
kernel, per-neuron sums, [all-reduce], neighbourhood smoothing, convergence norm, per-neuron
error) at the FROZEN map of SURVEY.md 8(d): full rows x cols lattice, prototypes = M rows of the
samples, sigma = 0.2 sqrt(M), gamma = 1 / sum of variances, samples resident in HBM.  Every step
starts from the same prototypes and computes everything again.

The headline uses the stateless filtered search (`algorithm="filtered"`: coarse int8-MFMA pre-pass
-> int8 candidate sweep with a rigorous error bound -> exact float64 search on the candidates;
results bit-identical to the all-pairs float64 search, nothing carried over between steps).  The
same JSON line also carries the all-pairs exact search (`exact`) and a training-like secondary
regime (`fine_phase`).

For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL);
samples are sharded by rows (each rank generates its own shard: weak scaling, per-GPU work
fixed) and the only collective is the all-reduce of the [S|K|a|E] sums, once per epoch.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import ctypes
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
    # one GPU's shard of BASELINE config 5 (N=4e6 over 8 GPUs), samples resident as bfloat16
    "c5": (500_000, 2048, 64, 64, 1005, "Synthetic N=4e6/8 d=2048 bf16, M=4096 (64x64)"),
}
BF16_WORKLOADS = ("c5",)
F64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 2.4 GHz x 2048 flop / 64 cycles (v_mfma_f64_16x16x4_f64)
I8_MFMA_PEAK_TOPS = 5033.0   # 256 x 4 x 2.4 GHz x 65536 op / 32 cycles (v_mfma_i32_32x32x32_i8, dense)
# HBM-side bytes of one full-sweep launch at C4 from the PMC passes committed under profiles/
# (FETCH_SIZE x 2 per the gfx950 correction); not measurable inside this process.  Keyed by the
# number of digit planes the sweep reads (1 -> sweep_i8_kernel<0,1,4>, 2 -> <0,2,2>, 3 -> <0,3,1>)
SWEEP_TRAFFIC_C4_BYTES = {1: 1.99e9, 2: 7.90e9, 3: 2.42e10}
SWEEP4_TRAFFIC_C4_BYTES = 4.14e9  # sweep4_i8_kernel (one product, 4-wavefront workgroups): plane 0 of X once per 256 prototypes
SWEEP_KERNEL = {1: "sweep_i8_kernel<0,1,4>", 2: "sweep_i8_kernel<0,2,2>", 3: "sweep_i8_kernel<0,3,1>"}
SWEEP_PRODUCTS = {1: 1, 2: 3, 3: 6}   # int8 digit products kept per (sample, prototype, k)


def lattice_hops(rows, cols):
    ii, jj = np.divmod(np.arange(rows * cols), cols)
    return (np.abs(ii[:, None] - ii[None]) + np.abs(jj[:, None] - jj[None])).astype(np.float64)


def make_shard(torch, n, d, seed, device, rank=0):
    """This rank's rows of the Gaussian-blob data set of SURVEY.md 8(d): 32 centres ~ N(0, 16 I)
    (ONE set for the whole data set, from `seed`), unit noise; the rows come from the stream
    `seed + rank`.  Generated in HBM."""
    gen = torch.Generator(device=device).manual_seed(seed)
    centers = torch.randn(32, d, device=device, generator=gen) * 4.0
    if rank:  # rank 0 goes on with the stream that drew the centres
        gen = torch.Generator(device=device).manual_seed(seed + rank)
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


class Harness:
    def __init__(self, torch, td, args, local, world, grouped):
        self.torch, self.td, self.args = torch, td, args
        self.local, self.world, self.grouped = local, world, grouped

    def sync(self):
        if self.grouped:
            self.td.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        t = self.torch.tensor([seconds], dtype=self.torch.float64,
                              device=self.torch.device("cuda", self.local))
        if self.grouped:
            self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t.item())

    def timed_epochs(self, be, step_fn, warmup, steps):
        """W untimed steps, then exactly `steps` timed ones bracketed by barrier + synchronize;
        returns (max-over-ranks seconds, per-phase mean ms from HIP events on the launch stream)."""
        be.kernel_events = None
        for _ in range(warmup):
            step_fn()
        be.kernel_events = []
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        self.sync()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        ev, be.kernel_events = be.kernel_events, None
        phases = {k: float(np.mean([a.elapsed_time(b) for (kk, a, b) in ev if kk == k]))
                  for k in ("bmu", "accumulate", "smooth")}
        return elapsed, phases


def frozen_map_regime(h, algorithm, X, W0, hop, sigma, gamma):
    """every step = one full epoch from the SAME frozen prototypes."""
    from dbgsom_amd import _native
    from dbgsom_amd.backend import HipBackend

    be = HipBackend(h.local, algorithm=algorithm)
    be.load_device(X)
    stage = None
    if algorithm != "exact":
        _native.call("dbgsom_filter_timing", 1)
    last = {}

    def step():
        last["res"] = be.epoch(W0, hop, sigma, gamma, "compact", False, keep_on_device=True)

    elapsed, phases = h.timed_epochs(be, step, h.args.warmup, h.args.steps)
    if algorithm != "exact":
        ms = (ctypes.c_double * 5)()
        _native.call("dbgsom_bmu_filtered_stage_ms", ms)
        stage = dict(zip(("slice_w", "prepass", "bucket_sort", "sweep", "exact_on_candidates"),
                         [float(v) for v in ms]))
        _native.call("dbgsom_filter_timing", 0)
        stage["sweep_planes"] = int(be._planes_used)   # what the adaptive policy settled on
        counts = be.filter_counts()
        stage["candidates_per_workgroup"] = {"mean": float(counts.mean()),
                                             "p90": float(np.percentile(counts, 90)),
                                             "max": int(counts.max())}
    wsum = float(last["res"].new_weights_dev.sum().item())
    be.release()
    return elapsed, phases, stage, wsum


def fine_phase_regime(h, X, W0, M, d, hop, gamma):
    """Secondary measurement: epochs of the FINE training phase (BaseSom.py:395-396, 899-900:
    constant sigma_end = max(0.7, 0.05 sqrt(M)), no growth) on a map that a decaying-sigma
    warm-up has organised, prototypes evolving from step to step as in training.  Timed for the
    exact search and for "auto" (previous winners as the filter's starting point).  Uses
    centres_layout="aligned": with the reference's compacted centre rows (quirk Q1) a 32x32 map
    with dead neurons scrambles itself into near-duplicate prototypes, which is not what a
    trained map looks like."""
    from dbgsom_amd.backend import HipBackend

    sig0, sig1 = 0.2 * np.sqrt(M), max(0.7, 0.05 * np.sqrt(M))
    schedule = [sig1 + (sig0 - sig1) * np.exp(-0.35 * e) for e in range(14)]
    out = {"sigma": sig1, "warmup_epochs": len(schedule), "centres_layout": "aligned"}
    wsum = {}
    for algo in ("exact", "auto"):
        be = HipBackend(h.local, algorithm=algo)
        be.load_device(X)
        state = {"W": W0.clone()}
        for s_ in schedule:  # untimed: organise the map
            state["W"] = be.epoch(state["W"], hop, s_, gamma, "aligned", False,
                                  keep_on_device=True).new_weights_dev

        def step():
            state["res"] = be.epoch(state["W"], hop, sig1, gamma, "aligned", False,
                                    keep_on_device=True)
            state["W"] = state["res"].new_weights_dev

        elapsed, _ = h.timed_epochs(be, step, 0, h.args.steps)
        out[algo] = {"ms_per_step": elapsed / h.args.steps * 1e3,
                     "value": X.shape[0] * h.world * h.args.steps / elapsed,
                     "dead_neurons": int((state["res"].activations == 0).sum())}
        if algo != "exact":
            c = be.filter_counts()
            out[algo]["candidates_per_workgroup"] = {"mean": float(c.mean()),
                                                     "p90": float(np.percentile(c, 90)),
                                                     "max": int(c.max())}
        wsum[algo] = float(state["W"].sum().item())
        be.release()
    out["prototypes_identical"] = wsum["exact"] == wsum["auto"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--samples-per-gpu", type=int, default=None, help="override N per GPU")
    ap.add_argument("--algorithm", default="filtered", choices=["filtered", "exact"],
                    help="BMU search of the headline value (both give identical results)")
    ap.add_argument("--fine-phase", type=int, default=1,
                    help="also time a trained map in the fine phase (sigma_end, evolving W) with "
                         "the exact and the hinted filtered search (0 disables)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000,
                    help="upper bound of rows timed by the CPU baseline (0 disables it); the "
                         "actual sample is sized for ~15 s of CPU work")
    args = ap.parse_args()

    import torch
    import torch.distributed as td

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
        # RCCL prints a version banner on stdout when the first communicator comes up: create it
        # here with stdout pointed at stderr, so that the ONE line on stdout is the JSON result
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            td.all_reduce(torch.zeros(1, device=device))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    h = Harness(torch, td, args, local, world, grouped)

    n_gpu, d, rows, cols, seed, cfg_name = WORKLOADS[args.workload]
    if args.samples_per_gpu:
        n_gpu = args.samples_per_gpu
    M = rows * cols
    X = make_shard(torch, n_gpu, d, seed, device, rank=rank)
    if args.workload in BF16_WORKLOADS:
        X = X.to(torch.bfloat16)  # storage dtype of the workload; HipBackend.load_device keeps it

    # frozen map: M rows of rank 0's shard, Manhattan hop distances, epoch-0 sigma, gamma = 1/var
    ctl = torch.zeros(M * d + 1, dtype=torch.float64, device=device)
    if rank == 0:
        g = torch.Generator(device=device).manual_seed(seed + 7)
        sel = torch.randperm(n_gpu, device=device, generator=g)[:M]
        ctl[:M * d] = X[sel].double().reshape(-1)
        ctl[M * d] = 1.0 / X.double().var(dim=0, unbiased=False).sum()
    if grouped:
        td.broadcast(ctl, 0)
    W0 = ctl[:M * d].reshape(M, d).contiguous()
    gamma = float(ctl[M * d].item())
    hop = lattice_hops(rows, cols)
    sigma = 0.2 * np.sqrt(M)  # BaseSom.py:876 at epoch 0

    results = {}
    for algo in dict.fromkeys([args.algorithm, "exact"]):  # headline first, exact always reported
        results[algo] = frozen_map_regime(h, algo, X, W0, hop, sigma, gamma)
    fine = fine_phase_regime(h, X, W0, M, d, hop, gamma) if args.fine_phase else None

    if rank == 0:
        total = n_gpu * world
        elapsed, phases, stage, wsum = results[args.algorithm]
        e_elapsed, e_phases, _, e_wsum = results["exact"]
        flops = 2.0 * n_gpu * M * d  # algorithmic flops of one BMU search
        e_ach = flops / (e_phases["bmu"] * 1e-3) / 1e12
        exact_roof = {"bound": "mfma", "kernel": "bmu_dma_kernel<float,1>", "dtype": "f64",
                      "achieved": e_ach, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": e_ach / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                      "kernel_ms": e_phases["bmu"]}
        if args.algorithm == "exact":
            roof = exact_roof
        else:
            dpad = (d + 63) // 64 * 64
            planes = int(stage.get("sweep_planes", 2))
            from dbgsom_amd import _native

            small = planes == 1 and _native.load().dbgsom_sweep_shape(M, d) == 4
            kernel = "sweep4_i8_kernel" if small else SWEEP_KERNEL[planes]
            traffic = SWEEP4_TRAFFIC_C4_BYTES if small else SWEEP_TRAFFIC_C4_BYTES[planes]
            ops = 2.0 * n_gpu * M * dpad * SWEEP_PRODUCTS[planes]  # int8 ops the sweep executes
            ach = ops / (stage["sweep"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": kernel, "dtype": "i8",
                    "achieved": ach, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                    "frac": ach / I8_MFMA_PEAK_TOPS,
                    "traffic": traffic if args.workload == "c4" and
                    n_gpu == WORKLOADS["c4"][0] else None,
                    "kernel_ms": stage["sweep"], "digit_products": SWEEP_PRODUCTS[planes],
                    "algorithmic_equiv_TFLOPs": flops / (stage["sweep"] * 1e-3) / 1e12}
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
                       "prototypes": M, "x_storage": "bf16" if args.workload in BF16_WORKLOADS else "f32",
                       "sharding": f"rows/{world}",
                       "map": "frozen (same prototypes every step)",
                       "bmu_algorithm": args.algorithm},
            "roofline": roof,
            "phases_ms": dict(phases, accumulate_GBps=n_gpu * d * 4 / (phases["accumulate"] * 1e-3) / 1e9),
        }
        if stage:
            out["filter_stages_ms"] = stage
        out["exact"] = {"value": total * args.steps / e_elapsed,
                        "ms_per_step": e_elapsed / args.steps * 1e3, "phases_ms": e_phases,
                        "roofline": exact_roof,
                        "prototypes_identical_to_headline": e_wsum == wsum}
        if fine:
            out["fine_phase"] = fine
        if args.cpu_sample > 0 and world == 1:  # the CPU baseline is reported at N = 1 only
            ns = min(args.cpu_sample, n_gpu)
            out["cpu_baseline"] = cpu_baseline(args.workload, X[:ns].float().cpu().numpy(),
                                               W0.cpu().numpy(), hop, sigma, gamma, n_gpu)
            out["gpu_vs_cpu"] = (n_gpu * args.steps / elapsed) / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if grouped:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
