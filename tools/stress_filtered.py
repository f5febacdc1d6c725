"""Randomised parity stress: filtered search vs all-pairs kernel over random shapes, dtypes, seed
strides and sweep variants (`run` is also a batch of tests/test_gpu_soak.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dbgsom_amd.backend import HipBackend  # noqa: E402



def run(seed=0, n_cases=60, refine=False, max_rows=30000, verbose=True):
    """n_cases random cases from the stream `seed`; returns the number of cases with any mismatch."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    for case in range(n_cases):
        N = int(rng.integers(130, max_rows))
        d = int(rng.choice([16, 17, 31, 48, 64, 100, 128, 129, 200, 256, 500, 784, 1000, 1234]))
        M = int(rng.integers(129, 2600))
        kind = rng.choice(["blobs", "uniform", "dups", "scaled"])
        dt = rng.choice(["f32", "f64", "bf16"])
        if kind == "blobs":
            c = rng.normal(size=(int(rng.integers(2, 50)), d)) * 4
            X = c[rng.integers(0, len(c), size=N)] + rng.normal(size=(N, d))
        elif kind == "uniform":
            X = rng.uniform(-1, 1, size=(N, d))
        elif kind == "dups":
            X = rng.integers(-2, 3, size=(N, d)).astype(np.float64)
        else:
            X = rng.normal(size=(N, d)) * np.exp(rng.uniform(-6, 6, size=(1, d)))
        X = X.astype(np.float64 if dt == "f64" else np.float32)
        W = X[rng.choice(N, M, replace=M > N)].astype(np.float64)
        if rng.random() < 0.3:
            W = W + rng.normal(size=W.shape) * 1e-3
        if rng.random() < 0.4:   # runs of identical prototypes: ties, more inseparable candidates than the refinement's slots
            for _ in range(int(rng.integers(1, 6))):
                a, k = int(rng.integers(0, M - 1)), int(rng.integers(2, 48))
                W[a:a + k] = W[int(rng.integers(0, M))]
        hop = np.zeros((M, M))
        storage = "bf16" if dt == "bf16" else None
        ex = HipBackend(algorithm="exact").load(X, storage=storage)
        fi = HipBackend(algorithm="filtered_hint" if rng.random() < 0.5 else "filtered").load(X, storage=storage)
        fi.seed_stride = int(rng.choice([0, 1, 2, 8, 32]))
        fi.sweep_planes = int(rng.choice([0, 1, 2, 3, 4, 4]))
        if refine:   # the per-sample refinement in half of the cases
            fi.refine = int(rng.random() < 0.5)
        if verbose:
            print(f"case {case:3d} N={N} d={d} M={M} {kind} {dt} ...", flush=True)
        ok = True
        for e in range(2):
            hop_e = np.abs(np.subtract.outer(np.arange(M), np.arange(M))).astype(np.float64)
            W_in = W
            re_ = ex.epoch(W, hop_e, 1.5, 1e-3, "aligned", True)
            rf = fi.epoch(W, hop_e, 1.5, 1e-3, "aligned", True)
            if not (np.array_equal(re_.winners, rf.winners) and np.array_equal(re_.distances, rf.distances)
                    and np.array_equal(re_.new_weights, rf.new_weights, equal_nan=True)):
                ok = False
                dw = np.flatnonzero(re_.winners != rf.winners)
                dd = np.flatnonzero(re_.distances != rf.distances)
                nanw = int(np.isnan(re_.new_weights).sum()), int(np.isnan(rf.new_weights).sum())
                print(f"   epoch {e}: winners differ {dw.size}, distances differ {dd.size}, new_weights equal "
                      f"{np.array_equal(re_.new_weights, rf.new_weights, equal_nan=True)} nan {nanw} "
                      f"max|dW| {np.nanmax(np.abs(re_.new_weights - rf.new_weights)):.3e}", flush=True)
                if dw.size:
                    from oracle import som_oracle as o
                    Xr = X.astype(np.float32) if dt != "f64" else X
                    if dt == "bf16":
                        Xr = torch.from_numpy(Xr).to(torch.bfloat16).float().numpy()
                    i = dw[:5]
                    od, oi = o.bmu_chain(Xr[i], W_in, 1)
                    print("   samples", i, "exact", re_.winners[i], "filtered", rf.winners[i], "oracle", oi,
                          "d exact", re_.distances[i], "d filt", rf.distances[i], "d oracle", od, flush=True)
            W = np.nan_to_num(re_.new_weights)   # dead neurons of the aligned layout are NaN rows
        c = fi.filter_counts()
        rc = fi.refine_counts() if fi.refined else None
        print(f"case {case:3d} N={N:6d} d={d:5d} M={M:5d} {kind:8s} {dt:4s} {fi.algorithm:13s} stride={fi.seed_stride:2d} "
              f"planes={fi.sweep_planes} lists mean {c.mean():7.1f} refined {rc} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        ex.release(); fi.release()
    print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60,
                      len(sys.argv) > 3 and sys.argv[3] == "refine") else 0)
