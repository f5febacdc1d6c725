#!/usr/bin/env python
"""Benchmark of the batch-SOM hot path on MI355X: samples/sec/epoch (BMU + update).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c3|c2|c5|c4iso|c2nn]
                    [--scaling strong|weak] [--via backend|ctx]

One "step" = one epoch of the hot path (reference dbgsom/BaseSom.py:403-407: BMU search, sample
kernel, per-neuron sums, [all-reduce], neighbourhood smoothing, convergence norm, per-neuron
error) at the FROZEN map of SURVEY.md 8(d): full rows x cols lattice, prototypes = M rows of the
samples, sigma = 0.2 sqrt(M), gamma = 1 / sum of variances, samples resident in HBM.  Every step
starts from the same prototypes and computes everything again -- ONE call of the C ABI
(`dbgsom_ctx_epoch`) per step.

The headline uses the stateless filtered search (`algorithm="filtered"`: coarse int8-MFMA pre-pass
-> int8 candidate sweep with a rigorous error bound -> exact float64 search on the candidates;
results bit-identical to the all-pairs float64 search, nothing carried over between steps).  The
same JSON line also carries the all-pairs exact search (`exact`), a training-like secondary
regime (`fine_phase`) and, at N = 1, two weakly clustered data sets (`other_data`) that show how
much of the headline is a property of the data.

`--via ctx` drives the same epoch with NumPy + ctypes only (no torch is imported): the seam a
maintainer of the reference would bind.

For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL).
BASELINE's metric is "N = 1e6 ... sample-sharded 1/2/4/8": the default `--scaling strong` keeps the
total N of the workload and gives every rank N / G rows; `--scaling weak` gives every rank the
whole N.  The only collective is the all-reduce of the [S|K|a|E] sums, once per epoch.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (samples, features, lattice rows, cols, seed, data kind, BASELINE.json config)
    "c4": (1_000_000, 784, 32, 32, 1004, "blobs", "Synthetic N=1e6 d=784 fp32, M=1024 (32x32)"),
    "c3": (1_000_000, 128, 45, 45, 1003, "blobs", "Synthetic Gaussian blobs N=1e6 d=128 fp32, M=2025 (45x45)"),
    "c2": (60_000, 784, 22, 23, 1002, "blobs", "Fashion-MNIST stand-in 60k x 784 fp32, M=506 (22x23)"),
    # one GPU's shard of BASELINE config 5 (N=4e6 over 8 GPUs), samples resident as bfloat16
    "c5": (500_000, 2048, 64, 64, 1005, "blobs", "Synthetic N=4e6/8 d=2048 bf16, M=4096 (64x64)"),
    # weakly clustered data: where the filter has nothing to hold on to
    "c4iso": (1_000_000, 784, 32, 32, 1014, "iso", "isotropic Gaussian N=1e6 d=784 fp32, M=1024 (32x32)"),
    "c2nn": (60_000, 784, 22, 23, 1012, "nonneg",
             "Fashion-MNIST stand-in, non-negative variant of SURVEY 8(d): clip(blobs, 0) scaled to "
             "[0, 255] then standardised, 60k x 784 fp32, M=506"),
}
BF16_WORKLOADS = ("c5",)
F64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 2.4 GHz x 2048 flop / 64 cycles (v_mfma_f64_16x16x4_f64)
I8_MFMA_PEAK_TOPS = 5033.0   # 256 x 4 x 2.4 GHz x 65536 op / 32 cycles (v_mfma_i32_32x32x32_i8, dense)
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured achievable)
SWEEP_PRODUCTS = {1: 1, 2: 3, 3: 6}   # int8 digit products kept per (sample, prototype, k)
SWEEP_KERNEL = {1: "sweep4_i8_kernel<0,8>", 2: "sweep_i8_kernel<0,2,2>", 3: "sweep_i8_kernel<0,3,1>"}
PHASES = ("bmu", "accumulate", "smooth", "slice_w", "prepass", "bucket_sort", "sweep",
          "exact_on_candidates")


def lattice_hops(rows, cols):
    ii, jj = np.divmod(np.arange(rows * cols), cols)
    return (np.abs(ii[:, None] - ii[None]) + np.abs(jj[:, None] - jj[None])).astype(np.float64)


def source_hash():
    """sha256 over the kernel sources: the key under which profiles/pmc_traffic.json holds the
    HBM bytes per launch that a rocprofv3 --pmc pass of THIS build measured."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "dbgsom_amd", "csrc")
    for name in sorted(os.listdir(src)):
        # (experiments.h is compiled by tools/build_variant.sh only, never into the library)
        if name.endswith((".hip", ".h")) and name != "experiments.h":
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass of this very build, or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    entry = table.get(source_hash(), {}).get(workload, {})
    for key, val in entry.items():
        if key != "source" and kernel.startswith(key):
            return float(val), entry.get("source")
    return None, None


# ---------------------------------------------------------------------------------------------
# synthetic data
# ---------------------------------------------------------------------------------------------
def make_shard(torch, n, d, seed, device, rank=0, kind="blobs"):
    """This rank's rows, generated in HBM.  blobs: 32 centres ~ N(0, 16 I) (ONE set for the whole
    data set, from `seed`), unit noise; the rows come from the stream `seed + rank`.
    iso: one isotropic Gaussian.  nonneg: clip(blobs, 0), scaled to [0, 255], standardised."""
    gen = torch.Generator(device=device).manual_seed(seed)
    centers = torch.randn(32, d, device=device, generator=gen) * 4.0
    if rank:  # rank 0 goes on with the stream that drew the centres
        gen = torch.Generator(device=device).manual_seed(seed + rank)
    X = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 100_000
    for s in range(0, n, step):  # chunked: no N x d float64 temporaries
        m = min(step, n - s)
        if kind == "iso":
            X[s:s + m] = torch.randn(m, d, device=device, generator=gen)
        else:
            lab = torch.randint(0, 32, (m,), device=device, generator=gen)
            X[s:s + m] = centers[lab] + torch.randn(m, d, device=device, generator=gen)
    if kind == "nonneg":
        X.clamp_(min=0)
        X.mul_(255.0 / float(X.max()))
        mean, std = X.mean(dim=0), X.std(dim=0)
        X.sub_(mean).div_(std.clamp(min=1e-6))
    return X


def iter_shard_numpy(n, d, seed, kind="blobs", rank=0, step=100_000):
    """The rows of make_shard_numpy in chunks of `step` (the stream is consumed chunk by chunk, so the chunks ARE
    the rows of the whole array): (first row, float32 chunk).  Not for kind="nonneg" (needs global moments)."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((32, d)).astype(np.float32) * 4.0
    if rank:
        rng = np.random.default_rng(seed + rank)
    for s in range(0, n, step):
        m = min(step, n - s)
        noise = rng.standard_normal((m, d), dtype=np.float32)
        yield s, (noise if kind == "iso" else centers[rng.integers(0, 32, m)] + noise)


def make_shard_numpy(n, d, seed, kind="blobs", rank=0):
    """`numpy.random.default_rng(seed)` on the host, reproducible on any box.  blobs: 32
    centres ~ N(0, 16 I) from the stream `seed` (ONE set for the whole data set), unit noise, labels
    uniform; the rows of rank r come from the stream `seed + r` (rank 0 goes on with the stream that
    drew the centres).  iso: one isotropic Gaussian.  nonneg: clip(blobs, 0), scaled to [0, 255],
    standardised."""
    X = np.empty((n, d), dtype=np.float32)
    for s, chunk in iter_shard_numpy(n, d, seed, kind, rank):
        X[s:s + chunk.shape[0]] = chunk
    if kind == "nonneg":
        np.clip(X, 0, None, out=X)
        X *= 255.0 / float(X.max())
        mean, std = X.mean(axis=0), X.std(axis=0)
        X -= mean
        X /= np.maximum(std, 1e-6)
    return X


def make_shard_device(torch, n, d, seed, device, rank=0, kind="blobs"):
    """make_shard_numpy, uploaded (the rows are generated on the host so that any box can regenerate the
    very inputs of a bench line; DBGSOM_BENCH_DATA=torch: the device generator of rounds 1-2, faster)."""
    if os.environ.get("DBGSOM_BENCH_DATA", "numpy") == "torch":
        return make_shard(torch, n, d, seed, device, rank, kind)
    X = torch.empty((n, d), dtype=torch.float32, device=device)
    if kind == "nonneg":
        Xh = make_shard_numpy(n, d, seed, kind, rank)
        for s0 in range(0, n, 250_000):
            X[s0:s0 + 250_000] = torch.from_numpy(Xh[s0:s0 + 250_000]).to(device)
        return X
    for s0, chunk in iter_shard_numpy(n, d, seed, kind, rank):   # (chunk by chunk: a C5 shard is 4 GB of float32)
        X[s0:s0 + chunk.shape[0]] = torch.from_numpy(chunk).to(device)
    return X


# ---------------------------------------------------------------------------------------------
# CPU baseline (the oracle port of the reference path), rank 0 at N = 1 only
# ---------------------------------------------------------------------------------------------
def cpu_baseline(Xs, W, hop, sigma, gamma, n_full, budget_s=15.0):
    """The reference CPU path (oracle port: sklearn NearestNeighbors + NumPy) on a bounded row
    sample, extrapolated to the full N: t = (t_bmu + t_acc) * N / Ns + t_smooth, with the smoothing
    both ways SURVEY.md 8(d) asks for: (ii) matmul form -- the fair baseline behind `value` -- and
    (i) the reference's own (M, M, d) broadcast (BaseSom.py:509-515), timed on a block of output
    rows and scaled to M (its temporary is M x M x d float64: 6.6 GB at C4, 550 GB at C5)."""
    from oracle import som_oracle as o

    try:
        import sklearn  # noqa: F401

        bmu, engine = o.bmu_sklearn, "sklearn NearestNeighbors.kneighbors"
    except ImportError:
        bmu, engine = o.bmu_blas, "NumPy dgemm expanded-L2"
    M, d = W.shape
    # size the sample for ~budget_s of CPU work: probe the BMU rate on 10k rows first
    tp = time.perf_counter()
    bmu(Xs[:10_000], W, 1)
    rate = 10_000 / (time.perf_counter() - tp)
    ns = int(min(Xs.shape[0], max(20_000, rate * budget_s)))
    Xs = Xs[:ns]
    t0 = time.perf_counter()
    dist, win = bmu(Xs, W, 1)
    t1 = time.perf_counter()
    kw = o.exp_similarity_gamma(dist, gamma)
    S, K, a, E = o.accumulate_numpy(Xs, win, kw, dist, M)
    t2 = time.perf_counter()
    C = o.voronoi_centers(S, K, a, "compact")
    h = o.gaussian_neighborhood(hop, sigma)
    Wn = o.smooth_matmul(h, a, C)
    o.change_total(W, Wn)
    t3 = time.perf_counter()
    # (i) reference-faithful broadcast on a row block sized for <= 1 GiB of temporary
    rows_b = int(max(1, min(M, (1 << 30) // (M * d * 8))))
    tb0 = time.perf_counter()
    Wb = o.smooth_broadcast_rows(h, a, C, 0, rows_b)
    t_block = time.perf_counter() - tb0
    t_broadcast = t_block * (M / rows_b)
    same = bool(np.allclose(Wb, Wn[:rows_b], rtol=1e-9, atol=1e-12, equal_nan=True))
    t_epoch = (t2 - t0) * (n_full / ns) + (t3 - t2)
    t_epoch_ref = (t2 - t0) * (n_full / ns) + t_broadcast
    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = None
    cpu_baseline.sample = (ns, win, dist)   # for the parity gate of the same line
    return {
        "value": n_full / t_epoch,
        "unit": "samples/s/epoch",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": (f"{ns} of {n_full} rows ({engine}, f32 X / f64 W as the reference runs it; "
                   f"CSR-matmul sums; matmul smoothing), bmu {t1 - t0:.2f}s acc {t2 - t1:.2f}s "
                   f"smooth {t3 - t2:.2f}s, BLAS threads {blas_threads}; "
                   "extrapolated t=(bmu+acc)*N/Ns+smooth"),
        "value_reference_faithful_smoothing": n_full / t_epoch_ref,
        "smoothing_s": {"matmul": t3 - t2, "broadcast_MMd": t_broadcast,
                        "broadcast_sample": f"{rows_b} of {M} output rows ({t_block:.2f}s, "
                                            f"{rows_b * M * d * 8 / 2**30:.2f} GiB temporary), x M/rows; "
                                            f"equal to the matmul form: {same}"},
    }


def parity_gate(X, W0, hop, sigma, gamma, info, Wn_hip, xbytes):
    """BASELINE.md 3 / SURVEY.md 8(d) "parity gates reported with every number", in the bench line itself:
    (1) the HIP winners of the rows the CPU leg searched (>= 5e5 at C4 on a bench host) against the
    reference path's own engine (sklearn NearestNeighbors on f32 X / f64 W; the oracle's chain form where
    sklearn is missing) -- indices identical, squared distances to 1e-9 of their scale; (2) the new prototypes of the WHOLE epoch
    against the oracle's update (per-neuron sums, Voronoi centres, neighbourhood smoothing in NumPy float64)
    fed the HIP winners and distances of all N rows -- north_star bound 1e-5, float64 builds reach 1e-10."""
    from oracle import som_oracle as o

    ns, win_cpu, dist_cpu = cpu_baseline.sample
    win, dist = info["winners"], info["distances"]
    differ = np.flatnonzero(win[:ns] != win_cpu)
    mism, ties = int(differ.size), 0
    if differ.size:
        # the reference's engine leaves the summation order of x.w to BLAS: two prototypes within its
        # rounding of each other may swap.  Such a row is no mismatch when the HIP winner is the order-pinned
        # oracle's (bit for bit) and the engine's own winner is as near to 1e-9.
        Xd = (X[differ].float() if xbytes == 2 else X[differ]).cpu().numpy()
        rd, ri = o.bmu_chain(Xd, W0, 1)
        tie = (win[differ] == ri) & (np.abs(dist_cpu[differ] - rd) <= 1e-9 * np.maximum(rd, 1e-300))
        ties = int(tie.sum())
        mism -= ties
    # squared distances, against the scale of the terms they are the difference of: the engine forms
    # |x|^2 - 2 x.w + |w|^2 in BLAS order, so a sample that IS a prototype comes out at ~1e-6 instead of 0
    scale = float(np.median(dist_cpu ** 2)) if ns else 1.0
    d_err = float(np.max(np.abs(dist[:ns] ** 2 - dist_cpu ** 2) / (dist_cpu ** 2 + scale))) if ns else 0.0
    M = W0.shape[0]
    kw = o.exp_similarity_gamma(dist, gamma)
    S = K = a = E = None
    for s0 in range(0, win.size, 250_000):   # (row chunks: a C5 shard is 4 GB of float32 on the host)
        s1 = min(win.size, s0 + 250_000)
        Xh = X[s0:s1].float().cpu().numpy() if xbytes == 2 else X[s0:s1].cpu().numpy()
        part = o.accumulate_numpy(Xh, win[s0:s1], kw[s0:s1], dist[s0:s1], M)
        S, K, a, E = part if S is None else (S + part[0], K + part[1], a + part[2], E + part[3])
    Wn = o.smooth_matmul(o.gaussian_neighborhood(hop, sigma), a, o.voronoi_centers(S, K, a, "compact"))
    scale = np.maximum(np.abs(Wn), 1e-12 * np.abs(Wn).max())
    w_err = float(np.nanmax(np.abs(Wn_hip - Wn) / scale))
    nan_same = bool(np.array_equal(np.isnan(Wn_hip), np.isnan(Wn)))
    ok = mism == 0 and d_err <= 1e-9 and w_err <= 1e-5 and nan_same
    return {"rows_checked": int(ns), "bmu_mismatches": mism, "bmu_ties_within_blas_rounding": ties,
            "dist2_rel_err_max": d_err,
            "w_rows_checked": int(win.size), "w_rel_err_max": w_err, "w_bound": 1e-5, "ok": bool(ok),
            "reference_engine": "sklearn NearestNeighbors (the reference's call) + oracle update in NumPy float64"}


# ---------------------------------------------------------------------------------------------
# roofline entries
# ---------------------------------------------------------------------------------------------
def list_flops(counts, n, d):
    """(useful, executed) float64 flops of the exact-on-candidates stage from the candidate-list
    lengths: a 128-sample workgroup evaluates its list in steps of 16 / 32 / 48 prototypes
    (classes <= 16, 17..32, > 32), every row of the workgroup against every listed prototype.
    Executed: per step whole 16-prototype tiles (v_mfma_f64_16x16x4), a last tile with up to 12 entries as
    groups of four (v_mfma_f64_4x4x4: a quarter of a tile's matrix time per group) -- filter.hip,
    subset_exact_kernel."""
    counts = counts.astype(np.float64)
    rows = np.full(counts.shape, 128.0)
    if n % 128:
        rows[-1] = n % 128
    step = np.where(counts <= 16, 16.0, np.where(counts <= 32, 32.0, 48.0))
    whole = np.floor(counts / step)                      # full steps
    last = counts - whole * step                         # entries of the last, partial step (0: none)
    tiles = np.ceil(last / 16.0)
    rem = last - 16.0 * np.maximum(tiles - 1.0, 0.0)
    last_exec = np.where(last > 0, np.where(rem <= 12, 16.0 * (tiles - 1.0) + 4.0 * np.ceil(rem / 4.0), 16.0 * tiles), 0.0)
    padded = whole * step + last_exec
    return float((2.0 * rows * counts * d).sum()), float((2.0 * 128.0 * padded * d).sum())


def rooflines(workload, n, d, M, xbytes, ph, planes, counts, refined=False):
    """One entry per dominant stage of the filtered epoch, each reproducible from profiles/:
    achieved = ALGORITHMIC work of one launch / its HIP-event duration in this run."""
    flops = 2.0 * n * M * d
    out = []
    if ph["sweep"] > 0 and planes == 0:
        # no sweep: candidates from the triangle inequality -- one pass over the top digit plane of X
        # (the seed distances) and an M x M matrix of prototype gaps; HBM-bound
        gbps = 1.0 * n * d / (ph["sweep"] * 1e-3) / 1e9
        traffic, src = measured_traffic(workload, "prune")
        out.append({"stage": "candidates by triangle inequality (no sweep)",
                    "kernel": "proto_gap_kernel + prune_mark_kernel", "bound": "hbm", "dtype": "i8",
                    "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                    "kernel_ms": ph["sweep"], "traffic": traffic, "traffic_source": src,
                    "algorithmic_bytes": 1.0 * n * d})
    elif ph["sweep"] > 0:
        ops = flops * SWEEP_PRODUCTS[planes]     # int8 multiply-adds x 2 on the d real features
        ach = ops / (ph["sweep"] * 1e-3) / 1e12
        kern = SWEEP_KERNEL[planes]
        traffic, src = measured_traffic(workload, "sweep")
        out.append({"stage": "candidate sweep", "kernel": kern, "bound": "mfma", "dtype": "i8",
                    "achieved": ach, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                    "frac": ach / I8_MFMA_PEAK_TOPS, "kernel_ms": ph["sweep"],
                    "digit_products": SWEEP_PRODUCTS[planes], "traffic": traffic,
                    "traffic_source": src, "algorithmic_bytes": float(n) * d})
    if ph["exact_on_candidates"] > 0 and refined:
        # refinement (the top two int8 digit planes of the gathered rows, once) + float64 chain on the
        # (sample, candidate) pairs (the stored rows, once): both stream X, HBM-bound
        nbytes = float(n) * d * (2 + xbytes)
        gbps = nbytes / (ph["exact_on_candidates"] * 1e-3) / 1e9
        out.append({"stage": "exact search on candidates: per-sample refinement + chain on pairs",
                    "kernel": "refine_i8_kernel + bucket sort + pair_exact_kernel", "bound": "hbm",
                    "dtype": "i8 digit products, f64 chain", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBPS, "kernel_ms": ph["exact_on_candidates"], "traffic": None,
                    "algorithmic_bytes": nbytes})
    elif ph["exact_on_candidates"] > 0 and counts is not None:
        useful, padded = list_flops(counts, n, d)
        t = ph["exact_on_candidates"] * 1e-3
        traffic, src = measured_traffic(workload, "subset_exact")
        out.append({"stage": "exact search on candidates", "kernel": "subset_exact_all_kernel / subset_exact_split_kernel (3 list-length classes, one launch)",
                    "bound": "mfma", "dtype": "f64", "achieved": useful / t / 1e12,
                    "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": useful / t / 1e12 / F64_MFMA_PEAK_TFLOPS,
                    "executed_TFLOPs": padded / t / 1e12,
                    "executed_frac": padded / t / 1e12 / F64_MFMA_PEAK_TFLOPS,
                    "kernel_ms": ph["exact_on_candidates"], "traffic": traffic, "traffic_source": src,
                    "algorithmic_bytes": float(n) * d * xbytes})
    if ph["bmu"] > 0:
        out.append({"stage": "whole BMU search", "kernel": "all stages of dbgsom_bmu_filtered", "bound": "mfma",
                    "dtype": "f64-equivalent", "achieved": flops / (ph["bmu"] * 1e-3) / 1e12,
                    "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (2 N M d of the search it replaces)",
                    "frac": flops / (ph["bmu"] * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                    "kernel_ms": ph["bmu"], "traffic": None})
    if ph["accumulate"] > 0:
        gbps = n * d * xbytes / (ph["accumulate"] * 1e-3) / 1e9
        traffic, src = measured_traffic(workload, "segsum")
        out.append({"stage": "accumulate (sort + segmented sums)", "kernel": "segsum_kernel + sort / finalize",
                    "bound": "hbm", "dtype": "f64 sums of " + ("bf16" if xbytes == 2 else "f32") + " rows",
                    "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                    "kernel_ms": ph["accumulate"], "traffic": traffic, "traffic_source": src,
                    "algorithmic_bytes": float(n) * d * xbytes})
    return out


def exact_roofline(n, d, M, bmu_ms):
    ach = 2.0 * n * M * d / (bmu_ms * 1e-3) / 1e12
    return {"stage": "all-pairs BMU search", "bound": "mfma", "kernel": "bmu_dma_kernel<float,1,4>",
            "dtype": "f64", "achieved": ach, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / F64_MFMA_PEAK_TFLOPS, "traffic": None, "kernel_ms": bmu_ms}


def count_stats(c):
    return {"mean": float(c.mean()), "p90": float(np.percentile(c, 90)), "max": int(c.max())}


# ---------------------------------------------------------------------------------------------
# --via ctx: NumPy + ctypes only
# ---------------------------------------------------------------------------------------------
def run_via_ctx(args):
    """The headline step through the raw context-level ABI: no torch, no HipBackend.  With --gpus N (one
    process per GPU, launched by torch.distributed.run or anything else that sets RANK / WORLD_SIZE /
    LOCAL_RANK): the per-epoch collective is RCCL driven by the library itself (dbgsom_ctx_set_rccl);
    rank 0's ncclUniqueId travels through a file named after MASTER_PORT, written before any GPU call of
    the other ranks; moments, start prototypes, barriers and timings go through dbgsom_ctx_allreduce_host."""
    import tempfile

    from dbgsom_amd import _native as nat

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_work, d, rows, cols, seed, kind, cfg_name = WORKLOADS[args.workload]
    if args.samples_per_gpu:
        n, n_total = args.samples_per_gpu, args.samples_per_gpu * world
    elif args.scaling == "strong":
        n, n_total = (n_work * (rank + 1)) // world - (n_work * rank) // world, n_work
    else:
        n, n_total = n_work, n_work * world
    M = rows * cols
    t_gen = time.perf_counter()
    X = make_shard_numpy(n, d, seed, kind, rank)
    t_gen = time.perf_counter() - t_gen
    hop = lattice_hops(rows, cols)
    sigma = 0.2 * np.sqrt(M)
    ctx = ctypes.c_void_p()
    nat.call("dbgsom_ctx_create", local % max(1, nat.device_count()), ctypes.byref(ctx))
    comm = ctypes.c_void_p()
    out = {}

    def allsum(v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        nat.call("dbgsom_ctx_allreduce_host", ctx, v.ctypes.data, v.size)
        return v

    def max_over_ranks(x):
        v = np.zeros(world)
        v[rank] = x
        return float(allsum(v).max())

    try:
        if world > 1 or os.environ.get("DBGSOM_FORCE_COLLECTIVE") == "1":
            # the id travels through a file named after THIS launch: the ranks share their launcher (torchrun's agent,
            # or whoever started them), so its pid and start time are a nonce every rank can compute and no earlier
            # run can have used; rank 0 also removes whatever may lie there before it writes
            ppid = os.getppid()
            try:
                with open("/proc/%d/stat" % ppid) as fh:
                    born = fh.read().rsplit(")", 1)[1].split()[19]
            except OSError:
                born = "0"
            path = os.path.join(tempfile.gettempdir(), "dbgsom_rccl_%s_%s_%d_%s.id" % (
                os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run"), ppid, born))
            uid = ctypes.create_string_buffer(128)
            if rank == 0:
                for stale in (path, path + ".tmp"):
                    if os.path.exists(stale):
                        os.unlink(stale)
                nat.call("dbgsom_rccl_unique_id", uid)
                with open(path + ".tmp", "wb") as fh:
                    fh.write(uid.raw)
                os.replace(path + ".tmp", path)
            else:
                t_wait = time.time()
                while not os.path.exists(path):
                    if time.time() - t_wait > 300:
                        sys.exit("bench.py: rank 0's ncclUniqueId never arrived at " + path)
                    time.sleep(0.05)
                uid = ctypes.create_string_buffer(open(path, "rb").read(), 128)
            nat.call("dbgsom_rccl_comm_init", uid, world, rank, ctypes.byref(comm))
            nat.call("dbgsom_ctx_set_rccl", ctx, comm)
            allsum(np.zeros(1))   # every rank has read the id
            if rank == 0:
                os.unlink(path)
        t_up = time.perf_counter()
        st = nat.BF16 if args.workload in BF16_WORKLOADS else nat.F32
        nat.call("dbgsom_ctx_load", ctx, X.ctypes.data, nat.F32, n, d, st)
        t_up = time.perf_counter() - t_up
        # gamma = 1 / sum of the variances of the WHOLE data set; W = M rows of rank 0's shard
        Xd = X[:200_000].astype(np.float64)
        mom = allsum(np.concatenate([Xd.sum(axis=0), (Xd * Xd).sum(axis=0), [Xd.shape[0]]]))
        del Xd
        nn = mom[2 * d]
        gamma = float(1.0 / (mom[d:2 * d] / nn - (mom[:d] / nn) ** 2).sum())
        W0 = np.zeros((M, d))
        if rank == 0:
            W0[:] = X[np.random.default_rng(seed + 7).choice(n, M, replace=False)]
        W0 = allsum(W0.reshape(-1)).reshape(M, d)
        nat.call("dbgsom_ctx_set_topology", ctx, hop.ctypes.data, M)
        chg, E, a = np.empty(1), np.empty(M), np.empty(M)
        Wn = {}
        for algo in dict.fromkeys([args.algorithm, "exact"]):
            nat.call("dbgsom_ctx_set_option", ctx, b"algorithm", nat.ALGORITHMS[algo])
            nat.call("dbgsom_ctx_set_option", ctx, b"timing", 0)   # the timed steps run un-instrumented
            nat.call("dbgsom_ctx_set_weights", ctx, W0.ctypes.data, M)

            def step():
                nat.call("dbgsom_ctx_epoch", ctx, None, M, 0, gamma, sigma, nat.CENTRES_COMPACT,
                         nat.EPOCH_FROZEN, None, chg.ctypes.data, E.ctypes.data, a.ctypes.data, None, None)

            for _ in range(args.warmup):
                step()
            ms = (ctypes.c_double * 8)()
            acc = np.zeros(8)
            allsum(np.zeros(1))   # barrier (every call of the ABI is blocking: the GPU is idle behind it)
            t0 = time.perf_counter()
            marks = []
            for _ in range(args.steps):
                step()
                marks.append(time.perf_counter())
            allsum(np.zeros(1))
            elapsed = max_over_ranks(time.perf_counter() - t0)
            per = np.zeros((world, args.steps))
            per[rank] = np.diff(np.array([t0] + marks))
            per = allsum(per.reshape(-1)).reshape(world, args.steps).max(axis=0)
            nat.call("dbgsom_ctx_set_option", ctx, b"timing", 1)
            for _ in range(3):   # phase times from three more (untimed) steps with HIP events
                step()
                nat.call("dbgsom_ctx_phase_ms", ctx, ms)
                acc += np.array(list(ms)) / 3
            Wn[algo] = np.empty((M, d))
            nat.call("dbgsom_ctx_get_weights", ctx, 1, Wn[algo].ctypes.data, M)
            out[algo] = (elapsed, dict(zip(PHASES, acc.tolist())), float(np.median(per)))
    finally:
        nat.call("dbgsom_ctx_destroy", ctx)
        if comm:
            nat.call("dbgsom_rccl_comm_destroy", comm)
    if rank != 0:
        return
    elapsed, ph, med = out[args.algorithm]
    e_elapsed, e_ph, e_med = out["exact"]
    line = {
        "metric": "samples/sec/epoch (BMU+update)", "value": n_total / med,
        "unit": "samples/s/epoch", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": med * 1e3, "ms_per_step_mean": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if (args.scaling == "weak" or args.samples_per_gpu) else "strong",
        "vs_baseline": None, "dtype": "f64",
        "data": "synthetic: numpy.random.default_rng(%d [+ rank]) on the host, SURVEY.md 8(d)" % seed,
        "config": {"workload": cfg_name, "samples_total": n_total, "samples_per_gpu": n, "features": d,
                   "prototypes": M, "x_storage": "bf16" if args.workload in BF16_WORKLOADS else "f32",
                   "sharding": f"rows/{world}", "map": "frozen (same prototypes every step)",
                   "bmu_algorithm": args.algorithm,
                   "via": "ctx: dbgsom_ctx_* through ctypes, NumPy host arrays only"
                          + ("; collective: RCCL inside the library (dbgsom_ctx_set_rccl)" if comm else "")},
        "roofline": exact_roofline(n, d, M, e_ph["bmu"]) if args.algorithm == "exact" else
        {"stage": "whole BMU search", "bound": "mfma", "dtype": "f64-equivalent",
         "achieved": 2.0 * n * M * d / (ph["bmu"] * 1e-3) / 1e12, "peak": F64_MFMA_PEAK_TFLOPS,
         "unit": "TFLOP/s (2 N M d of the search it replaces)",
         "frac": 2.0 * n * M * d / (ph["bmu"] * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, "traffic": None,
         "kernel_ms": ph["bmu"]},
        "phases_ms": ph,
        "exact": {"value": n_total / e_med, "ms_per_step": e_med * 1e3,
                  "prototypes_identical_to_headline": bool(np.array_equal(Wn["exact"], Wn[args.algorithm],
                                                                           equal_nan=True))},
        "host_s": {"generate": t_gen, "dbgsom_ctx_load (PCIe upload + norms)": t_up},
        "torch_imported": "torch" in sys.modules,
    }
    print(json.dumps(line))


# ---------------------------------------------------------------------------------------------
# default path: HipBackend (thin over the same ABI) + torch.distributed for N > 1
# ---------------------------------------------------------------------------------------------
class Harness:
    def __init__(self, torch, td, args, local, world, grouped):
        self.torch, self.td, self.args = torch, td, args
        self.local, self.world, self.grouped = local, world, grouped

    def sync(self):
        if self.grouped:
            self.td.barrier()
        self.torch.cuda.synchronize()

    def coll_device(self):
        """Where the harness's own small collectives live: HBM under RCCL, the host otherwise (the
        2-rank rehearsal on one GPU runs over gloo: DBGSOM_BENCH_BACKEND=gloo)."""
        on_gpu = not self.grouped or self.td.get_backend() == "nccl"
        return self.torch.device("cuda", self.local) if on_gpu else self.torch.device("cpu")

    def max_over_ranks(self, seconds):
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.coll_device())
        if self.grouped:
            self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t.item())

    def timed_epochs(self, be, step_fn, warmup, steps):
        """W untimed steps, then exactly `steps` timed ones bracketed by barrier + synchronize, with
        the context's event instrumentation OFF (ten event records per epoch are worth 2-5 % at small
        shapes); then min(steps, 5) more steps, untimed, with it ON for the per-phase HIP-event
        times (events on the context's stream).  Returns (max-over-ranks seconds, median ms per phase)."""
        be.phase_log = None
        be._set("timing", 0)
        for _ in range(warmup):
            step_fn()
        self.sync()
        t0 = time.perf_counter()
        marks = []
        for _ in range(steps):   # (a step = one BLOCKING call of the C ABI: its end is a synchronisation)
            step_fn()
            marks.append(time.perf_counter())
        self.sync()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        per = np.diff(np.array([t0] + marks))
        t = self.torch.tensor(per, dtype=self.torch.float64, device=self.coll_device())
        if self.grouped:
            self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        self.step_seconds = t.cpu().numpy()   # of the last timed regime, max over ranks per step
        be._set("timing", 1)
        be.phase_log = []
        for _ in range(min(steps, 5)):
            step_fn()
        log, be.phase_log = np.array(be.phase_log), None
        be._set("timing", 0)
        # (the median of the instrumented steps: one of them may be a trial of the policy -- the other form of the
        #  exact stage on a C5 shard takes three times the usual -- and a mean carries it into every roofline)
        return elapsed, dict(zip(PHASES, np.median(log, axis=0).tolist()))


def frozen_map_regime(h, algorithm, X, W0, hop, sigma, gamma, steps=None, warmup=None, assignments=False):
    """every step = one full epoch from the SAME frozen prototypes (resident in HBM,
    DBGSOM_EPOCH_FROZEN), new prototypes stay in HBM."""
    from dbgsom_amd.backend import RESIDENT, HipBackend

    be = HipBackend(h.local, algorithm=algorithm)
    be.load_device(X)
    be.set_weights(W0)
    M = W0.shape[0]

    def step():
        be.epoch(RESIDENT, hop, sigma, gamma, "compact", False, keep_on_device=True, frozen=True)

    elapsed, phases = h.timed_epochs(be, step, h.args.warmup if warmup is None else warmup,
                                     h.args.steps if steps is None else steps)
    info = {"filter_log": be.filter_log[-1] if be.filter_log else None, "step_seconds": h.step_seconds.copy()}
    counts = None
    if be.filter_log and be.filter_log[-1][0] == "filtered":
        counts = be.filter_counts()
        info["sweep_planes"] = int(be.filter_log[-1][2])
        info["candidates_per_workgroup"] = count_stats(counts)
        info["refined"] = bool(be.refined)
        if be.refined:
            pairs, groups, overflow, distinct = be.refine_counts()
            info["refinement"] = {"pairs_per_sample": pairs / be.n_samples, "workgroups_refined": groups,
                                  "overflow_samples": overflow}
    if assignments:   # one more (untimed) epoch of the same regime that also brings back winners and distances
        res = be.epoch(RESIDENT, hop, sigma, gamma, "compact", True, keep_on_device=True, frozen=True)
        info["winners"], info["distances"] = res.winners, res.distances
    Wn = be.get_weights(1)   # the output of the last (frozen) epoch
    be.release()
    del M
    return elapsed, phases, info, counts, Wn


def fine_phase_regime(h, X, W0, M, hop, gamma, n_total):
    """Secondary measurement: epochs of the FINE training phase (BaseSom.py:395-396, 899-900:
    constant sigma_end = max(0.7, 0.05 sqrt(M)), no growth) on a map that a decaying-sigma
    warm-up has organised, prototypes evolving from step to step as in training and never leaving
    HBM.  Timed for the exact search and for "auto" (previous winners as the filter's seeds).
    Uses centres_layout="aligned": with the reference's compacted centre rows (quirk Q1) a 32x32
    map with dead neurons scrambles itself into near-duplicate prototypes, which is not what a
    trained map looks like."""
    from dbgsom_amd.backend import RESIDENT, HipBackend

    sig0, sig1 = 0.2 * np.sqrt(M), max(0.7, 0.05 * np.sqrt(M))
    schedule = [sig1 + (sig0 - sig1) * np.exp(-0.35 * e) for e in range(14)]
    out = {"sigma": sig1, "warmup_epochs": len(schedule) + 24, "centres_layout": "aligned"}
    final = {}
    for algo in ("exact", "auto"):
        be = HipBackend(h.local, algorithm=algo)
        be.load_device(X)
        be.set_weights(W0)
        for s_ in schedule + [sig1] * 24:  # untimed: organise the map, then let it (and the search
            be.epoch(RESIDENT, hop, s_, gamma, "aligned", False, keep_on_device=True)  # policy) settle
        state = {}

        def step():
            state["res"] = be.epoch(RESIDENT, hop, sig1, gamma, "aligned", False, keep_on_device=True)

        elapsed, _ = h.timed_epochs(be, step, 0, h.args.steps)
        out[algo] = {"ms_per_step": elapsed / h.args.steps * 1e3,
                     "value": n_total * h.args.steps / elapsed,
                     "dead_neurons": int((state["res"].activations == 0).sum())}
        if algo != "exact" and be.filter_log and be.filter_log[-1][0] == "filtered":
            out[algo]["candidates_per_workgroup"] = count_stats(be.filter_counts())
        final[algo] = be.get_weights(0)
        be.release()
    out["prototypes_identical"] = bool(np.array_equal(final["exact"], final["auto"], equal_nan=True))
    return out


def other_data_regime(h, torch, device, name):
    """The frozen-map step on a weakly clustered data set, `auto` (what a fit would run) beside
    `exact`: which search `auto` settles on, how long the candidate lists are, and the rate."""
    n, d, rows, cols, seed, kind, cfg_name = WORKLOADS[name]
    M = rows * cols
    X = make_shard_device(torch, n, d, seed, device, 0, kind)
    sel = torch.from_numpy(np.random.default_rng(seed + 7).choice(n, M, replace=False)).to(device)
    W0 = X[sel].double().cpu().numpy()
    gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
    hop, sigma = lattice_hops(rows, cols), 0.2 * np.sqrt(M)
    out = {"workload": cfg_name}
    W = {}
    for algo in ("auto", "filtered", "exact"):
        steps = max(h.args.steps, 12) if algo == "auto" else h.args.steps   # auto: time to back off
        # (ten untimed epochs for the filtered searches: the engine's policy tries its arms first)
        el, ph, info, counts, W[algo] = frozen_map_regime(h, algo, X, W0, hop, sigma, gamma, steps=steps,
                                                          warmup=10 if algo != "exact" else h.args.warmup)
        out[algo] = {"value": n * steps / el, "ms_per_step": el / steps * 1e3, "bmu_ms": ph["bmu"],
                     "last_epoch": info.get("filter_log")}
        if counts is not None:
            out[algo]["candidates_per_workgroup"] = count_stats(counts)
            out[algo]["sweep_planes"] = info.get("sweep_planes")
    out["prototypes_identical"] = bool(np.array_equal(W["exact"], W["auto"], equal_nan=True) and
                                       np.array_equal(W["exact"], W["filtered"], equal_nan=True))
    del X
    torch.cuda.empty_cache()
    return out


def workload_entry(h, torch, device, name, cpu_budget_s):
    """One more BASELINE configuration in the same line (N = 1): the frozen-map step of workload `name` with the
    stateless filtered search and with the all-pairs search, its rooflines, its CPU leg on a bounded row sample and
    ITS OWN parity gate (winners of the sampled rows against the reference's engine, the new prototypes of the whole
    epoch against the oracle's update on all rows)."""
    n, d, rows, cols, seed, kind, cfg_name = WORKLOADS[name]
    M = rows * cols
    X = make_shard_device(torch, n, d, seed, device, 0, kind)
    xbytes = 4
    if name in BF16_WORKLOADS:
        X = X.to(torch.bfloat16)
        xbytes = 2
    mom = torch.zeros(2 * d, dtype=torch.float64, device=device)
    for s0 in range(0, n, 100_000):
        blk = X[s0:s0 + 100_000].double()
        mom[:d] += blk.sum(dim=0)
        mom[d:] += (blk * blk).sum(dim=0)
    gamma = float(1.0 / (mom[d:] / n - (mom[:d] / n) ** 2).sum().item())
    sel = torch.from_numpy(np.random.default_rng(seed + 7).choice(n, M, replace=False)).to(device)
    W0 = X[sel].double().cpu().numpy()
    hop, sigma = lattice_hops(rows, cols), 0.2 * np.sqrt(M)
    el, ph, info, counts, Wn = frozen_map_regime(h, "filtered", X, W0, hop, sigma, gamma, assignments=cpu_budget_s > 0)
    e_el, e_ph, e_info, _, e_Wn = frozen_map_regime(h, "exact", X, W0, hop, sigma, gamma,
                                                     steps=min(h.args.steps, 10), warmup=min(h.args.warmup, 2))
    med, e_med = float(np.median(info["step_seconds"])), float(np.median(e_info["step_seconds"]))
    roofs = rooflines(name, n, d, M, xbytes, ph, int(info.get("sweep_planes", 1)), counts, bool(info.get("refined")))
    out = {"workload": cfg_name, "value": n / med, "unit": "samples/s/epoch", "ms_per_step": med * 1e3,
           "floor": n / e_med, "floor_ms_per_step": e_med * 1e3,
           "floor_roofline": exact_roofline(n, d, M, e_ph["bmu"]),
           "roofline": max(roofs[:2], key=lambda r: r["kernel_ms"]) if len(roofs) >= 2 else roofs[0],
           "rooflines": roofs, "phases_ms": ph,
           "filter": {"sweep_planes": info.get("sweep_planes"), "refined": info.get("refined", False),
                      "candidates_per_workgroup": info.get("candidates_per_workgroup")},
           "prototypes_identical_to_all_pairs": bool(np.array_equal(e_Wn, Wn, equal_nan=True))}
    if cpu_budget_s > 0:
        Xs = X[:min(n, 200_000)].float().cpu().numpy()
        out["cpu_baseline"] = cpu_baseline(Xs, W0, hop, sigma, gamma, n, budget_s=cpu_budget_s)
        out["gpu_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        out["parity"] = parity_gate(X, W0, hop, sigma, gamma, info, Wn, xbytes)
    del X
    torch.cuda.empty_cache()
    return out


def fit_regime(X_host, max_neurons):
    """A whole `SomVQ.fit` (the reference's entry point, BaseSom.py:88-131) on the workload's samples handed over as
    a host array: wall clock, epochs, the share spent outside the C ABI, bytes over PCIe (tools/bench_fit.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_fit import fit_profile

    kw = dict(random_state=0, max_neurons=max_neurons, n_iter=120, spreading_factor=0.9, convergence_iter=1,
              coarse_training_frac=0.7)
    fit_profile(X_host[:4000], **dict(kw, n_iter=8))   # (library and allocator warm)
    return fit_profile(X_host, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the workload's N in total, N / G rows per rank (BASELINE's "
                         "metric: N=1e6 sample-sharded 1/2/4/8); weak = the workload's N per rank")
    ap.add_argument("--via", default="backend", choices=["backend", "ctx"],
                    help="ctx: drive the context-level C ABI with NumPy + ctypes only (no torch)")
    ap.add_argument("--samples-per-gpu", type=int, default=None, help="override N per GPU")
    ap.add_argument("--algorithm", default="filtered", choices=["filtered", "exact"],
                    help="BMU search of the headline value (both give identical results)")
    ap.add_argument("--fine-phase", type=int, default=1,
                    help="also time a trained map in the fine phase (sigma_end, evolving W) with "
                         "the exact and the hinted filtered search (0 disables)")
    ap.add_argument("--other-data", type=int, default=1,
                    help="N = 1, workload c4: also run the weakly clustered data sets c4iso and c2nn")
    ap.add_argument("--other-workloads", type=int, default=1,
                    help="N = 1, workload c4: also run BASELINE's c3, c2 and the c5 shard, each with its own "
                         "rooflines, CPU leg on a bounded row sample and parity gate (`other_workloads`)")
    ap.add_argument("--fit", type=int, default=1,
                    help="N = 1, workload c4: also time a whole SomVQ.fit on the workload's samples (`fit`)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000,
                    help="upper bound of rows timed by the CPU baseline (0 disables it); the "
                         "actual sample is sized for ~15 s of CPU work")
    args = ap.parse_args()
    if args.via == "ctx":
        return run_via_ctx(args)

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
    local = local % max(1, torch.cuda.device_count())   # (rehearsal: several ranks on one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = os.environ.get("DBGSOM_BENCH_BACKEND", "nccl")
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
            if backend == "nccl":
                td.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
                td.all_reduce(torch.zeros(1, device=device))
            else:
                td.init_process_group(backend, rank=rank, world_size=world)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    h = Harness(torch, td, args, local, world, grouped)

    n_work, d, rows, cols, seed, kind, cfg_name = WORKLOADS[args.workload]
    from dbgsom_amd.backend import shard_bounds

    if args.samples_per_gpu:
        n_gpu, n_total = args.samples_per_gpu, args.samples_per_gpu * world
    elif args.scaling == "strong":
        lo, hi = shard_bounds(n_work, rank, world)
        n_gpu, n_total = hi - lo, n_work
    else:
        n_gpu, n_total = n_work, n_work * world
    M = rows * cols
    X = make_shard_device(torch, n_gpu, d, seed, device, rank=rank, kind=kind)
    xbytes = 4
    if args.workload in BF16_WORKLOADS:
        X = X.to(torch.bfloat16)  # storage dtype of the workload; the context keeps it
        xbytes = 2

    # frozen map: M rows of rank 0's shard, Manhattan hop distances, epoch-0 sigma, gamma = 1/var
    # of the WHOLE data set (moments all-reduced)
    mom = torch.zeros(2 * d + 1, dtype=torch.float64, device=device)
    Xd = X.double() if n_gpu * d <= 4e8 else None
    if Xd is not None:
        mom[:d], mom[d:2 * d] = Xd.sum(dim=0), (Xd * Xd).sum(dim=0)
    else:
        for s in range(0, n_gpu, 100_000):
            blk = X[s:s + 100_000].double()
            mom[:d] += blk.sum(dim=0)
            mom[d:2 * d] += (blk * blk).sum(dim=0)
    del Xd
    mom[2 * d] = n_gpu
    if grouped:
        mom = mom.to(h.coll_device())
        td.all_reduce(mom)
    nn = float(mom[2 * d].item())
    var = mom[d:2 * d] / nn - (mom[:d] / nn) ** 2
    gamma = float(1.0 / var.sum().item())
    ctl = torch.zeros(M * d, dtype=torch.float64, device=device)
    if rank == 0:   # W = M rows of X chosen by default_rng(seed + 7).choice(N, M, replace=False)
        sel = torch.from_numpy(np.random.default_rng(seed + 7).choice(n_gpu, M, replace=False)).to(device)
        ctl[:] = X[sel].double().reshape(-1)
    if grouped:
        ctl = ctl.to(h.coll_device())
        td.broadcast(ctl, 0)
    W0 = ctl.reshape(M, d).cpu().numpy()
    hop = lattice_hops(rows, cols)
    sigma = 0.2 * np.sqrt(M)  # BaseSom.py:876 at epoch 0

    results = {}
    gate = world == 1 and args.cpu_sample > 0   # the parity gate needs the CPU leg's winners
    for algo in dict.fromkeys([args.algorithm, "exact"]):  # headline first, exact always reported
        results[algo] = frozen_map_regime(h, algo, X, W0, hop, sigma, gamma,
                                          assignments=gate and algo == args.algorithm)
    fine = fine_phase_regime(h, X, W0, M, hop, gamma, n_total) if args.fine_phase else None
    weak = None
    if world > 1 and args.scaling == "strong" and not args.samples_per_gpu:
        # beside the metric's strong-scaling line: every rank with the workload's full N
        Xw = make_shard_device(torch, n_work, d, seed, device, rank=rank, kind=kind)
        if args.workload in BF16_WORKLOADS:
            Xw = Xw.to(torch.bfloat16)
        el, _, _, _, _ = frozen_map_regime(h, args.algorithm, Xw, W0, hop, sigma, gamma)
        weak = {"samples_per_gpu": n_work, "value": n_work * world * args.steps / el,
                "ms_per_step": el / args.steps * 1e3}
        del Xw

    out = None
    if rank == 0:
        elapsed, phases, info, counts, Wn = results[args.algorithm]
        e_elapsed, e_phases, _, _, e_Wn = results["exact"]
        exact_roof = exact_roofline(n_gpu, d, M, e_phases["bmu"])
        if args.algorithm == "exact":
            roofs, roof = [exact_roof], exact_roof
        else:
            roofs = rooflines(args.workload if n_gpu == n_work else None, n_gpu, d, M, xbytes, phases,
                              int(info.get("sweep_planes", 1)), counts, bool(info.get("refined")))
            # the dominant kernel of the step = the longest stage that is one kernel family
            roof = max(roofs[:2], key=lambda r: r["kernel_ms"]) if len(roofs) >= 2 else roofs[0]
        med = float(np.median(info["step_seconds"]))    # the median epoch (max over ranks per step)
        e_med = float(np.median(results["exact"][2]["step_seconds"]))
        out = {
            "metric": "samples/sec/epoch (BMU+update)",
            "value": n_total / med,
            "unit": "samples/s/epoch",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": med * 1e3,
            "ms_per_step_mean": elapsed / args.steps * 1e3,   # the bracketed total / steps
            "value_definition": "N / median step time of the K timed steps (max over ranks per step)",
            "higher_is_better": True,
            "scaling": "weak" if (args.scaling == "weak" or args.samples_per_gpu) else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": ("synthetic: numpy.random.default_rng(%d [+ rank]) Gaussian blobs, SURVEY.md 8(d)" % seed
                     if os.environ.get("DBGSOM_BENCH_DATA", "numpy") != "torch" else "synthetic (torch device generator)"),
            # the data-independent floor (all-pairs search, the same inputs) and both against F_A = 2 N M d
            "floor": n_total / e_med,
            "frac_vs_2NMd": 2.0 * n_total * M * d / med / 1e12 / (F64_MFMA_PEAK_TFLOPS * world),
            "floor_frac_vs_2NMd": 2.0 * n_total * M * d / e_med / 1e12 / (F64_MFMA_PEAK_TFLOPS * world),
            "config": {"workload": cfg_name, "samples_total": n_total, "samples_per_gpu": n_gpu,
                       "features": d, "prototypes": M,
                       "x_storage": "bf16" if args.workload in BF16_WORKLOADS else "f32",
                       "sharding": f"rows/{world}",
                       "map": "frozen (same prototypes every step)",
                       "bmu_algorithm": args.algorithm,
                       "via": "HipBackend -> dbgsom_ctx_epoch (one C-ABI call per step)"},
            "roofline": roof,
            "rooflines": roofs,
            "build": source_hash(),
            "phases_ms": phases,
        }
        if info.get("candidates_per_workgroup"):
            out["filter"] = {"sweep_planes": info.get("sweep_planes"),
                             "candidates_per_workgroup": info["candidates_per_workgroup"],
                             "refined": info.get("refined", False), "refinement": info.get("refinement")}
        out["exact"] = {"value": n_total / e_med,
                        "ms_per_step": e_med * 1e3, "phases_ms": e_phases,
                        "roofline": exact_roof,
                        "prototypes_identical_to_headline": bool(np.array_equal(e_Wn, Wn, equal_nan=True))}
        if fine:
            out["fine_phase"] = fine
        if weak:
            out["weak_scaling_same_run"] = weak
    if world == 1 and args.other_data and args.workload == "c4" and not args.samples_per_gpu:
        out["other_data"] = {name: other_data_regime(h, torch, device, name) for name in ("c4iso", "c2nn")}
    if rank == 0:
        if args.cpu_sample > 0 and world == 1:  # the CPU baseline is reported at N = 1 only
            ns = min(args.cpu_sample, X.shape[0])
            Xs = X[:ns].float().cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(Xs, W0, hop, sigma, gamma, n_gpu)
            out["gpu_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["parity"] = parity_gate(X, W0, hop, sigma, gamma, results[args.algorithm][2],
                                        results[args.algorithm][4], xbytes)
        if world == 1 and args.fit and args.workload == "c4" and not args.samples_per_gpu:
            out["fit"] = fit_regime(X.float().cpu().numpy(), 1024)
        if world == 1 and args.other_workloads and args.workload == "c4" and not args.samples_per_gpu:
            del X
            torch.cuda.empty_cache()
            out["other_workloads"] = {name: workload_entry(h, torch, device, name, 4.0 if args.cpu_sample > 0 else 0.0)
                                      for name in ("c3", "c2", "c5")}
            bad = [k for k, v in out["other_workloads"].items() if v.get("parity") and not v["parity"]["ok"]]
            if bad and out.get("parity"):
                out["parity"]["other_workloads_failed"] = bad
                out["parity"]["ok"] = False
        if out.get("other_data"):   # the same step on data without clusters: what the headline owes its data set
            iso = out["other_data"]["c4iso"]["auto"]
            out["value_isotropic"] = iso["value"]
            out["ms_per_step_isotropic"] = iso["ms_per_step"]
        print(json.dumps(out))
        if out.get("parity") and not out["parity"]["ok"]:
            sys.stderr.write("bench.py: PARITY GATE FAILED: %r\n" % (out["parity"],))
            sys.stdout.flush()
            os._exit(3)
    if grouped:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
