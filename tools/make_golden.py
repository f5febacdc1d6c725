"""Generate the golden fixtures under ``tests/golden/`` by RUNNING THE REFERENCE.

TEST TOOLING.  Runs only in the build container (needs ``/root/reference``); the fixtures it
writes are plain data (inputs regenerated from seeds + the reference's outputs) and are what
travels to the GPU box.  No reference source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

What is recorded: for selected epochs of real ``fit`` runs, the inputs and
outputs of the four hot-path methods of ``dbgsom/BaseSom.py``

    _get_winning_neurons   :446-464   -> winners, distances
    _calculate_exp_similarity :533-538 -> sample_weights
    _update_weights        :470-523   -> voronoi_set_centers (compact rows, Q1), activations,
                                          new weights (read back from the graph), converged_
    _write_accumulative_error :541-561 -> per-neuron errors

plus a per-epoch trace of the host logic around them (neurons_, sigma, sum of weights) and the
final fitted attributes.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import networkx as nx  # noqa: E402
from sklearn.datasets import load_digits, make_blobs  # noqa: E402

import dbgsom.BaseSom as ref_base  # noqa: E402
from dbgsom.SomClassifier import SomClassifier  # noqa: E402
from dbgsom.SomVQ import SomVQ  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Recorder:
    """Wraps the hot-path entry points of one estimator instance and records chosen epochs."""

    def __init__(self, est, epochs_full):
        self.est = est
        self.epochs_full = set(epochs_full)
        self.full = {}  # epoch -> dict of arrays
        self.trace = {
            "n_neurons": [],
            "sigma": [],
            "weights_sum": [],
            "change_total": [],
            "n_dead": [],
            "phase_fine": [],
            "neurons": [],
        }
        self._cur = None
        self._centers = None
        self._errors = None
        # module-level numba functions (identity-decorated by the shim)
        self._orig_centers = ref_base.numba_voronoi_set_centers
        self._orig_qe = ref_base.numba_quantization_error

    def __enter__(self):
        est = self.est
        rec = self

        def centers(*a, **k):
            out = rec._orig_centers(*a, **k)
            rec._centers = out.copy()
            return out

        def qe(*a, **k):
            out = rec._orig_qe(*a, **k)
            rec._errors = out.copy()
            return out

        ref_base.numba_voronoi_set_centers = centers
        ref_base.numba_quantization_error = qe

        orig_bmu = est._get_winning_neurons
        orig_sim = est._calculate_exp_similarity
        orig_upd = est._update_weights
        orig_err = est._write_accumulative_error
        self._in_loop = False

        def upd(sample_weights, winners, data):
            e = est._current_epoch
            w_before = est.weights_.copy()
            dm = np.array(est._distance_matrix, dtype=np.float64)
            sigma = est._calculate_current_sigma()
            orig_upd(sample_weights, winners, data)
            w_after = np.array([d["weight"] for _, d in est.som_.nodes.data()])
            act = np.bincount(winners, minlength=w_before.shape[0]).astype(np.float64)
            rec.trace["n_neurons"].append(int(w_before.shape[0]))
            rec.trace["sigma"].append(float(sigma))
            rec.trace["weights_sum"].append(float(w_after.sum()))
            rec.trace["change_total"].append(
                float(np.sum(np.linalg.norm(w_before - w_after, axis=1)))
            )
            rec.trace["n_dead"].append(int((act == 0).sum()))
            rec.trace["phase_fine"].append(est._training_phase == "fine")
            rec.trace["neurons"].append([list(map(int, n)) for n in est.neurons_])
            if e in rec.epochs_full:
                rec._cur = {
                    "weights_in": w_before,
                    "hop_distance": dm,
                    "sigma": np.float64(sigma),
                    "total_variance": np.float64(est._total_variance),
                    "winners": np.asarray(winners).copy(),
                    "sample_weights": np.asarray(sample_weights).copy(),
                    "centers_compact": rec._centers,
                    "activations": act,
                    "weights_out": w_after,
                    "converged": np.bool_(est.converged_),
                }

        def err(winners, y, distances):
            orig_err(winners, y, distances)
            e = est._current_epoch
            if e in rec.epochs_full and rec._cur is not None:
                rec._cur["distances"] = np.asarray(distances).copy()
                rec._cur["errors"] = rec._errors
                rec.full[e] = rec._cur
                rec._cur = None

        est._update_weights = upd
        est._write_accumulative_error = err
        self._restore = (orig_bmu, orig_sim)
        return self

    def __exit__(self, *exc):
        ref_base.numba_voronoi_set_centers = self._orig_centers
        ref_base.numba_quantization_error = self._orig_qe
        del self.est._update_weights
        del self.est._write_accumulative_error

    def flat(self):
        out = {}
        for e, d in self.full.items():
            for k, v in d.items():
                if v is not None:  # (no numba error sum under the entropy criterion)
                    out[f"e{e}_{k}"] = v
        out["epochs_full"] = np.array(sorted(self.full), dtype=np.int64)
        for k in ("n_neurons", "sigma", "weights_sum", "change_total", "n_dead", "phase_fine"):
            out[f"trace_{k}"] = np.array(self.trace[k])
        # ragged neuron lists -> flat + offsets
        flat = [p for ep in self.trace["neurons"] for p in ep]
        out["trace_neurons_flat"] = np.array(flat, dtype=np.int64).reshape(-1, 2)
        out["trace_neurons_off"] = np.cumsum([0] + [len(ep) for ep in self.trace["neurons"]])
        return out


def final_attrs(est, X):
    d = {
        "final_weights": est.weights_,
        "final_neurons": np.array(est.neurons_, dtype=np.int64),
        "final_labels": np.asarray(est.labels_) if hasattr(est, "labels_") else np.zeros(0),
        "final_qe": np.float64(est.quantization_error_),
        "final_te": np.float64(est.topographic_error_),
        "final_n_iter": np.int64(est.n_iter_),
        "final_converged": np.bool_(est.converged_),
        "final_growing_threshold": np.float64(est.growing_threshold_),
        "final_hit_count": np.array([d["hit_count"] for _, d in est.som_.nodes.data()]),
        "final_density": np.array([d["density"] for _, d in est.som_.nodes.data()]),
        "final_average_distance": np.array(
            [d["average_distance"] for _, d in est.som_.nodes.data()]
        ),
        "final_error": np.array([d["error"] for _, d in est.som_.nodes.data()], dtype=np.float64),
        "final_epoch_created": np.array(
            [d["epoch_created"] for _, d in est.som_.nodes.data()], dtype=np.int64
        ),
    }
    # k=2 BMU on the fitted map (used by topographic error, BaseSom.py:945)
    dist2, idx2 = est._get_winning_neurons(X, n_bmu=2)
    d["final_bmu2_idx"] = idx2
    d["final_bmu2_dist"] = dist2
    return d


def run_fit_case(name, make_est, X, y, epochs_full, extra_meta):
    est = make_est()
    t0 = time.time()
    with Recorder(est, epochs_full) as rec:
        if y is None:
            est.fit(X)
        else:
            est.fit(X, y)
    out = rec.flat()
    out.update(final_attrs(est, X))
    if y is not None:
        out["final_classes"] = est.classes_
        out["final_node_label"] = np.array(
            [d["label"] for _, d in est.som_.nodes.data()], dtype=np.int64
        )
        out["final_node_probabilities"] = np.array(
            [d["probabilities"] for _, d in est.som_.nodes.data()]
        )
        out["final_predict"] = est.predict(X)
        out["final_score"] = np.float64(est.score(X, y))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    meta = {
        "wall_s": round(time.time() - t0, 2),
        "n_iter_": int(est.n_iter_),
        "n_neurons": len(est.neurons_),
        "epochs_full": sorted(int(e) for e in rec.full),
        "max_dead": int(max(rec.trace["n_dead"])),
        "epochs_with_dead": int(sum(1 for n in rec.trace["n_dead"] if n > 0)),
        "quantization_error_": float(est.quantization_error_),
        "topographic_error_": float(est.topographic_error_),
        "weights_sum": float(est.weights_.sum()),
    }
    meta.update(extra_meta)
    print(name, meta, flush=True)
    return meta


def _intended_grow_vertical(self, X, y=None):
    """BaseSom._grow_vertical (BaseSom.py:157-179) with its two slips corrected -- OUR restatement
    of the evident intent, not reference text: the comprehension unpacks (node, error) instead of
    comparing the tuple with a float, and the child is stored under the node's key instead of its
    position.  Everything a child does (clone, fit on X[winners == i]) is the reference's own."""
    from sklearn.base import clone

    self.vertical_growing_threshold_ = 1.5 * self.growing_threshold_
    _, winners = self._get_winning_neurons(X, n_bmu=1)
    relevant = [i for i, (_node, error) in enumerate(self.som_.nodes(data="error"))
                if error > self.vertical_growing_threshold_]
    for i in relevant:
        new_som = clone(self)
        X_f = X[winners == i]
        y_f = None if y is None else y[winners == i]
        if X_f.shape[0] > self.min_samples_vertical_growth:
            new_som.fit(X_f, y_f)
            self.som_.nodes[self.neurons_[i]]["som"] = new_som


def _walk_maps(est, path, out):
    k = len(out["paths"])
    out["paths"].append(list(path))
    out[f"map{k}_weights"] = est.weights_
    out[f"map{k}_neurons"] = np.array(est.neurons_, dtype=np.int64)
    out[f"map{k}_n_iter"] = np.int64(est.n_iter_)
    out[f"map{k}_qe"] = np.float64(est.quantization_error_)
    out[f"map{k}_te"] = np.float64(est.topographic_error_)
    out[f"map{k}_labels_sha"] = np.array(sha(np.asarray(est.labels_, dtype=np.int64)))
    out[f"map{k}_n_samples"] = np.int64(len(est.labels_))
    out[f"map{k}_threshold"] = np.float64(est.growing_threshold_)
    for i, node in enumerate(est.neurons_):
        child = est.som_.nodes[node].get("som")
        if child is not None:
            _walk_maps(child, path + [i], out)


def run_vertical_case(name, make_est, X, extra_meta):
    """vertical_growth=True: (1) what the reference does as it stands, (2) the tree of maps it
    evidently means to build (the two slips of _grow_vertical corrected, see above)."""
    t0 = time.time()
    raised = ""
    try:
        make_est().fit(X)
    except Exception as e:  # noqa: BLE001 -- recorded, that IS the reference's behaviour
        raised = f"{type(e).__name__}: {e}"
    orig = ref_base.BaseSom._grow_vertical
    ref_base.BaseSom._grow_vertical = _intended_grow_vertical
    try:
        est = make_est().fit(X)
    finally:
        ref_base.BaseSom._grow_vertical = orig
    out = {"paths": []}
    _walk_maps(est, [], out)
    paths = out.pop("paths")
    out["n_maps"] = np.int64(len(paths))
    out["paths_flat"] = np.array([p for path in paths for p in path], dtype=np.int64)
    out["paths_off"] = np.cumsum([0] + [len(p) for p in paths])
    out["reference_raises"] = np.array(raised)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    meta = {"wall_s": round(time.time() - t0, 2), "n_maps": len(paths), "paths": paths,
            "reference_as_is": raised or "no exception"}
    meta.update(extra_meta)
    print(name, meta, flush=True)
    return meta


def blobs_f32(n, d, seed, n_centers=32, scale=4.0):
    """Synthetic generator shared with bench.py / tests."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(0.0, scale, size=(n_centers, d))
    lab = rng.integers(0, n_centers, size=n)
    X = centers[lab] + rng.normal(0.0, 1.0, size=(n, d))
    return X.astype(np.float32), lab


def frozen_state(est, X, W, rows, cols, epoch=0, n_iter=200):
    """Put a reference estimator into the state `_grow_som` would hand to the hot path:
    a full rows x cols lattice with given weights (frozen map)."""
    g = nx.Graph()
    nodes = [(i, j) for i in range(rows) for j in range(cols)]
    for n, w in zip(nodes, W):
        g.add_node(n, weight=w, epoch_created=0, error=0)
    for (i, j) in nodes:
        for nb in ((i + 1, j), (i, j + 1)):
            if nb in g.nodes:
                g.add_edge((i, j), nb)
    est.som_ = g
    est.neurons_ = list(g.nodes)
    est.weights_ = np.array([d["weight"] for _, d in g.nodes.data()])
    est._distance_matrix = nx.floyd_warshall_numpy(g)
    est._total_variance = np.var(X, axis=0).sum()
    est._current_epoch = epoch
    est._training_phase = "coarse"
    est.converged_ = False
    return est


def run_frozen_case(name, X, rows, cols, seed, wdtype):
    """One hot-path epoch on a frozen rectangular map, through the reference's own methods."""
    M = rows * cols
    sel = np.random.default_rng(seed + 7).choice(X.shape[0], M, replace=False)
    W = X[sel].astype(wdtype)
    est = frozen_state(SomVQ(n_iter=200), X, W, rows, cols)
    t0 = time.time()
    centers_box = {}
    orig_centers = ref_base.numba_voronoi_set_centers

    def centers(*a, **k):
        out = orig_centers(*a, **k)
        centers_box["c"] = out.copy()
        return out

    ref_base.numba_voronoi_set_centers = centers
    try:
        distances, winners = est._get_winning_neurons(X, n_bmu=1)
        sw = est._calculate_exp_similarity(distances)
        sigma = est._calculate_current_sigma()
        est._update_weights(sw, winners, X)
        est._write_accumulative_error(winners, None, distances)
    finally:
        ref_base.numba_voronoi_set_centers = orig_centers
    w_after = np.array([d["weight"] for _, d in est.som_.nodes.data()])
    errors = np.array([d["error"] for _, d in est.som_.nodes.data()], dtype=np.float64)
    dist2, idx2 = est._get_winning_neurons(X, n_bmu=2)
    out = {
        "rows": np.int64(rows),
        "cols": np.int64(cols),
        "seed": np.int64(seed),
        "sel": sel.astype(np.int64),
        "sigma": np.float64(sigma),
        "total_variance": np.float64(est._total_variance),
        "winners": winners.astype(np.int32),
        "distances": distances,
        "sample_weights": sw,
        "activations": np.bincount(winners, minlength=M).astype(np.float64),
        "errors": errors,
        "weights_out_sum_rows": w_after.sum(axis=1),
        "weights_out_head": w_after[:8].copy(),
        "weights_out_sha": np.array(sha(w_after)),
        "centers_compact_sum_rows": centers_box["c"].sum(axis=1),
        "bmu2_idx": idx2.astype(np.int32),
        "bmu2_dist": dist2,
        "converged": np.bool_(est.converged_),
    }
    if w_after.nbytes <= 4 << 20:
        out["weights_out"] = w_after
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    meta = {
        "wall_s": round(time.time() - t0, 2),
        "N": int(X.shape[0]),
        "d": int(X.shape[1]),
        "M": M,
        "x_dtype": str(X.dtype),
        "w_dtype": str(np.dtype(wdtype)),
        "n_dead": int((out["activations"] == 0).sum()),
        "winners_sha": sha(winners.astype(np.int64)),
    }
    print(name, meta, flush=True)
    return meta


GROW_CASES = {
    # fits that grow well past the 128 prototypes below which the build's default search is the all-pairs kernel
    # (dbgsom_amd/backend.py FILTER_MIN_PROTOTYPES): the reference's own loop, BaseSom.py:387-417, growth steps
    # :411-417 / :588-614, with a third of the final lattice dead at the end (deleted, :119)
    "grow_blobs_f32": dict(
        X="make_blobs(n_samples=20000, n_features=32, centers=200, random_state=3)[0].astype(float32)",
        data_seed=3, dtype=np.float32,
        kw=dict(random_state=0, max_neurons=300, spreading_factor=0.9, n_iter=120, convergence_iter=2),
        epochs=[0, 27, 40, 58, 90, 118]),
    "grow_blobs_f64": dict(
        X="make_blobs(n_samples=20000, n_features=32, centers=200, random_state=4)[0] (float64)",
        data_seed=4, dtype=np.float64,
        kw=dict(random_state=0, max_neurons=300, spreading_factor=0.9, n_iter=100, convergence_iter=2),
        epochs=[0, 30, 52, 98]),
    # the same kind of fit on other blobs: at epoch 43 two dead neurons of the lattice (the same hop distance to every
    # LIVE neuron, own hit count 0) leave the smoothing as bit-identical rows (175 and 194),
    # and at epoch 44 the 389 samples nearest to them are EXACT ties.  The reference's winner among the two is
    # whatever its BLAS's summation order gives each column of the GEMM (here: 271 samples to 194, 118 to 175) -- not
    # a property of the algorithm.  Recorded to pin exactly that: everything up to the tie is reproduced, and at the
    # tie the reference's winners differ from lowest-index-wins only between bit-identical prototypes.
    "grow_dup_f64": dict(
        X="make_blobs(n_samples=20000, n_features=32, centers=200, random_state=3)[0] (float64)",
        data_seed=3, dtype=np.float64,
        kw=dict(random_state=0, max_neurons=260, spreading_factor=0.95, n_iter=100, convergence_iter=2),
        epochs=[0, 30, 43, 44]),
}


def run_grow_case(name):
    c = GROW_CASES[name]
    X = make_blobs(n_samples=20000, n_features=32, centers=200, random_state=c["data_seed"])[0].astype(c["dtype"])
    kw = c["kw"]
    return run_fit_case(name, lambda: SomVQ(**kw), X, None, c["epochs"],
                        {"X": c["X"], "est": "SomVQ(%s)" % ", ".join(f"{k}={v!r}" for k, v in kw.items())})


def main():
    only = set(sys.argv[1:])
    mpath = os.path.join(OUT, "manifest.json")
    manifest = {"versions": ref_shim.versions(), "cases": {}}
    if only and os.path.exists(mpath):
        manifest = json.load(open(mpath))
    digits = load_digits()
    Xd = digits.data  # float64, integer valued 0..16 (exact-tie prone, SURVEY hard part 1)

    if not only or "digits_entropy" in only:
        # supervised growth criterion: per-neuron label entropy (BaseSom.py:547-551)
        manifest["cases"]["digits_entropy"] = run_fit_case(
            "digits_entropy",
            lambda: SomClassifier(random_state=0, n_iter=30, growth_criterion="entropy",
                                  spreading_factor=0.4, max_neurons=40),
            Xd[:900], digits.target[:900], [0, 5, 29],
            {"X": "load_digits() data/target [:900]",
             "est": "SomClassifier(random_state=0, n_iter=30, growth_criterion='entropy', "
                    "spreading_factor=0.4, max_neurons=40)"},
        )
    if not only or "vertical_blobs" in only:
        Xv = make_blobs(n_samples=4000, n_features=10, centers=7, cluster_std=2.0, random_state=4)[0]
        manifest["cases"]["vertical_blobs"] = run_vertical_case(
            "vertical_blobs",
            lambda: SomVQ(random_state=2, vertical_growth=True, n_iter=24, max_neurons=9,
                          min_samples_vertical_growth=150, spreading_factor=0.6),
            Xv,
            {"X": "make_blobs(n_samples=4000, n_features=10, centers=7, cluster_std=2.0, random_state=4)[0]",
             "est": "SomVQ(random_state=2, vertical_growth=True, n_iter=24, max_neurons=9, "
                    "min_samples_vertical_growth=150, spreading_factor=0.6)"},
        )
    for name in GROW_CASES:
        if not only or name in only:
            manifest["cases"][name] = run_grow_case(name)
    if only:
        with open(mpath, "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        print("done (subset)")
        return

    manifest["cases"]["digits_f64"] = run_fit_case(
        "digits_f64", lambda: SomVQ(random_state=0), Xd, None, [0, 1, 7, 50, 111],
        {"X": "sklearn.datasets.load_digits().data (float64)", "est": "SomVQ(random_state=0)"},
    )
    manifest["cases"]["digits_f32"] = run_fit_case(
        "digits_f32", lambda: SomVQ(random_state=0), Xd.astype(np.float32), None, [0, 1, 7, 50],
        {"X": "load_digits().data.astype(float32)", "est": "SomVQ(random_state=0)"},
    )
    Xb, _ = make_blobs(n_samples=3000, n_features=16, centers=8, random_state=1)
    manifest["cases"]["blobs_dead"] = run_fit_case(
        "blobs_dead",
        lambda: SomVQ(random_state=0, spreading_factor=0.99, max_neurons=400, n_iter=60,
                      convergence_iter=2),
        Xb, None, [0, 3, 10, 30, 59],
        {"X": "make_blobs(n_samples=3000, n_features=16, centers=8, random_state=1)[0]",
         "est": "SomVQ(random_state=0, spreading_factor=0.99, max_neurons=400, n_iter=60, "
                "convergence_iter=2)"},
    )
    # low-d (kd-tree engine in sklearn: d <= 15), linear decay, explicit sigmas
    X2 = np.random.default_rng(5).normal(size=(2000, 3)) * np.array([3.0, 1.0, 0.3])
    manifest["cases"]["lowd_linear"] = run_fit_case(
        "lowd_linear",
        lambda: SomVQ(random_state=3, n_iter=40, decay_function="linear", max_neurons=60,
                      spreading_factor=0.3, sigma_start=2.0, sigma_end=0.5,
                      coarse_training_frac=0.6, convergence_iter=3),
        X2, None, [0, 5, 20, 39],
        {"X": "default_rng(5).normal(size=(2000,3)) * [3,1,.3]",
         "est": "SomVQ(random_state=3, n_iter=40, decay_function='linear', max_neurons=60, "
                "spreading_factor=0.3, sigma_start=2.0, sigma_end=0.5, "
                "coarse_training_frac=0.6, convergence_iter=3)"},
    )
    manifest["cases"]["digits_clf"] = run_fit_case(
        "digits_clf", lambda: SomClassifier(random_state=0), Xd, digits.target, [0, 20],
        {"X": "load_digits() data/target", "est": "SomClassifier(random_state=0)"},
    )
    # exact ties: duplicated prototypes + integer data (every fp64 op exact)
    Xt = np.random.default_rng(11).integers(0, 4, size=(600, 20)).astype(np.float64)
    Xt[300:] = Xt[:300]  # every sample appears twice
    manifest["cases"]["ties_int"] = run_fit_case(
        "ties_int", lambda: SomVQ(random_state=1, n_iter=12, max_neurons=30), Xt, None,
        [0, 1, 5, 11],
        {"X": "default_rng(11).integers(0,4,(600,20)).astype(f64); X[300:]=X[:300]",
         "est": "SomVQ(random_state=1, n_iter=12, max_neurons=30)"},
    )

    # frozen-map single epochs (bench-shaped, scaled down)
    Xf, _ = blobs_f32(20000, 784, 1002)
    manifest["cases"]["frozen_c2_f32"] = run_frozen_case(
        "frozen_c2_f32", Xf, 12, 13, 1002, np.float64)
    Xg, _ = blobs_f32(30000, 128, 1003)
    manifest["cases"]["frozen_c3_f32"] = run_frozen_case(
        "frozen_c3_f32", Xg, 15, 15, 1003, np.float64)
    Xh, _ = blobs_f32(8000, 100, 77)
    manifest["cases"]["frozen_f64"] = run_frozen_case(
        "frozen_f64", Xh.astype(np.float64) * 1.0000001, 9, 11, 77, np.float64)

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
