"""Whole `SomVQ.fit` on bench-shaped data: wall clock, epochs, growth steps, the share of the wall clock spent
outside the C ABI (host growth / lattice logic, input validation), bytes over PCIe.  SURVEY 8 f-1 / f-4.

    python tools/bench_fit.py [workload=c4] [max_neurons=1024] [n_iter=120] [spreading_factor=0.9]
                              [convergence_iter=1] [coarse_training_frac=0.7] [rows=N]

`fit_profile()` is what bench.py's `fit` sub-line calls."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def fit_profile(X, backend_opts=None, **kw):
    """Fit SomVQ(**kw) on the host array X through the default backend (`backend_opts`: options of an explicit
    HipBackend -- algorithm, refine, sweep_planes ... -- for experiments); time every call of the C ABI."""
    from dbgsom_amd import SomVQ, _native

    calls = {}
    orig = _native.call

    def timed(name, *a):
        t = time.perf_counter()
        try:
            return orig(name, *a)
        finally:
            c = calls.setdefault(name, [0, 0.0])
            c[0] += 1
            c[1] += time.perf_counter() - t

    _native.call = timed
    sizes = []
    try:
        if backend_opts:
            from dbgsom_amd.backend import HipBackend

            hb = HipBackend(algorithm=backend_opts.pop("algorithm", "auto"))
            for k, v in backend_opts.items():
                hb._set(k, int(v))
            kw = dict(kw, backend=hb)
        est = SomVQ(**kw)
        be = est._engine()
        log = be._log_epoch

        t_last = [time.perf_counter()]

        def spy():
            log()
            now = time.perf_counter()
            last = be.filter_log[-1] if be.filter_log else ("exact", None)
            sizes.append((be._last_M, last[0], round((now - t_last[0]) * 1e3, 3), last[1] if len(last) > 1 else None,
                          last[2] if len(last) > 2 else None))
            t_last[0] = now

        be._log_epoch = spy
        t0 = time.perf_counter()
        est.fit(X)
        wall = time.perf_counter() - t0
    finally:
        _native.call = orig
    in_abi = sum(c[1] for c in calls.values())
    epoch_s = calls.get("dbgsom_ctx_epoch", [0, 0.0])[1]
    tr = est._training_traffic
    top = sorted(calls.items(), key=lambda kv: -kv[1][1])[:6]
    return {
        "wall_s": wall, "epochs": int(est.n_iter_) + 1, "neurons_final": len(est.neurons_),
        "neurons_max": max(e[0] for e in sizes), "growth_steps": len(est._growth_epochs),
        "epochs_filtered": sum(1 for e in sizes if e[1] == "filtered"),
        "per_epoch": sizes if os.environ.get("DBGSOM_FIT_TRACE") else None,
        "in_abi_s": in_abi, "epoch_calls_s": epoch_s, "host_s": wall - in_abi, "host_share": (wall - in_abi) / wall,
        "pcie_bytes": {"samples_up": int(X.nbytes), "prototypes_up": int(tr["w_upload_bytes"]),
                       "prototypes_down": int(tr["w_download_bytes"]), "rows_written": int(tr["w_row_writes"])},
        "abi_top": {k: {"calls": v[0], "s": round(v[1], 4)} for k, v in top},
        "quantization_error": float(est.quantization_error_), "topographic_error": float(est.topographic_error_),
        "params": {k: (v if isinstance(v, (int, float, str, bool)) or v is None else str(v)) for k, v in kw.items()
                   if k != "backend"},
    }


if __name__ == "__main__":
    import json

    import bench

    opts = dict(a.split("=") for a in sys.argv[1:])
    n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[opts.get("workload", "c4")]
    n = int(opts.get("rows", n))
    X = bench.make_shard_numpy(n, d, seed, kind)
    kw = dict(random_state=0, max_neurons=int(opts.get("max_neurons", 1024)), n_iter=int(opts.get("n_iter", 120)),
              spreading_factor=float(opts.get("spreading_factor", 0.9)),
              convergence_iter=int(opts.get("convergence_iter", 1)),
              coarse_training_frac=float(opts.get("coarse_training_frac", 0.7)))
    bopts = {k: opts[k] for k in ("algorithm", "refine", "sweep_planes", "defer") if k in opts}
    fit_profile(X[:4000], **dict(kw, n_iter=10))   # warm up the library
    if os.environ.get("DBGSOM_FIT_CPROFILE"):
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        res = fit_profile(X, dict(bopts) or None, **kw)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(28)
        print(json.dumps(res))
    else:
        print(json.dumps(fit_profile(X, dict(bopts) or None, **kw)))
