"""Regenerates the INPUTS of the golden cases from seeds (the fixtures hold only the reference's
outputs + small arrays).  Mirrors tools/make_golden.py, which produced tests/golden/*.npz by
running the reference."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def blobs_f32(n, d, seed, n_centers=32, scale=4.0):
    rng = np.random.default_rng(seed)
    centers = rng.normal(0.0, scale, size=(n_centers, d))
    lab = rng.integers(0, n_centers, size=n)
    X = centers[lab] + rng.normal(0.0, 1.0, size=(n, d))
    return X.astype(np.float32), lab


def load(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


def case_X(name):
    """-> (X, y or None)"""
    if name in ("digits_f64", "digits_clf"):
        from sklearn.datasets import load_digits

        dg = load_digits()
        return dg.data, (dg.target if name == "digits_clf" else None)
    if name == "digits_entropy":
        from sklearn.datasets import load_digits

        dg = load_digits()
        return dg.data[:900], dg.target[:900]
    if name == "digits_f32":
        from sklearn.datasets import load_digits

        return load_digits().data.astype(np.float32), None
    if name == "blobs_dead":
        from sklearn.datasets import make_blobs

        return make_blobs(n_samples=3000, n_features=16, centers=8, random_state=1)[0], None
    if name in ("grow_blobs_f32", "grow_blobs_f64", "grow_dup_f64"):
        from sklearn.datasets import make_blobs

        X = make_blobs(n_samples=20000, n_features=32, centers=200, random_state=4 if name == "grow_blobs_f64" else 3)[0]
        return X.astype(np.float32 if name.endswith("f32") else np.float64), None
    if name == "lowd_linear":
        return np.random.default_rng(5).normal(size=(2000, 3)) * np.array([3.0, 1.0, 0.3]), None
    if name == "ties_int":
        Xt = np.random.default_rng(11).integers(0, 4, size=(600, 20)).astype(np.float64)
        Xt[300:] = Xt[:300]
        return Xt, None
    if name == "vertical_blobs":
        from sklearn.datasets import make_blobs

        return make_blobs(n_samples=4000, n_features=10, centers=7, cluster_std=2.0, random_state=4)[0], None
    if name == "frozen_c2_f32":
        return blobs_f32(20000, 784, 1002)[0], None
    if name == "frozen_c3_f32":
        return blobs_f32(30000, 128, 1003)[0], None
    if name == "frozen_f64":
        return blobs_f32(8000, 100, 77)[0].astype(np.float64) * 1.0000001, None
    raise KeyError(name)


FIT_CASES = ["digits_f64", "digits_f32", "blobs_dead", "lowd_linear", "ties_int", "digits_clf",
             "digits_entropy", "grow_blobs_f32", "grow_blobs_f64"]
# fits of the reference that grow past the 128 prototypes from which the build's default search is the filtered one
# (up to 247 / 221 neurons, a third of them dead at the end): the estimator's default path inside a growing fit
GROW_CASES = ["grow_blobs_f32", "grow_blobs_f64"]
# a fit of the same kind in which two dead neurons at the same hop distances from every live one become bit-identical prototypes (epoch 43) and the reference's
# BLAS splits the exact ties of epoch 44 between them: reproduced up to the tie, see tools/make_golden.py GROW_CASES
DUP_CASE, DUP_EPOCH, DUP_ROWS = "grow_dup_f64", 44, (175, 194)
CLF_CASES = ("digits_clf", "digits_entropy")
FROZEN_CASES = ["frozen_c2_f32", "frozen_c3_f32", "frozen_f64"]

EST_KWARGS = {
    "digits_f64": dict(random_state=0),
    "digits_f32": dict(random_state=0),
    "digits_clf": dict(random_state=0),
    "blobs_dead": dict(random_state=0, spreading_factor=0.99, max_neurons=400, n_iter=60,
                       convergence_iter=2),
    "lowd_linear": dict(random_state=3, n_iter=40, decay_function="linear", max_neurons=60,
                        spreading_factor=0.3, sigma_start=2.0, sigma_end=0.5,
                        coarse_training_frac=0.6, convergence_iter=3),
    "ties_int": dict(random_state=1, n_iter=12, max_neurons=30),
    "grow_blobs_f32": dict(random_state=0, max_neurons=300, spreading_factor=0.9, n_iter=120, convergence_iter=2),
    "grow_blobs_f64": dict(random_state=0, max_neurons=300, spreading_factor=0.9, n_iter=100, convergence_iter=2),
    "grow_dup_f64": dict(random_state=0, max_neurons=260, spreading_factor=0.95, n_iter=100, convergence_iter=2),
    "digits_entropy": dict(random_state=0, n_iter=30, growth_criterion="entropy",
                           spreading_factor=0.4, max_neurons=40),
    "vertical_blobs": dict(random_state=2, vertical_growth=True, n_iter=24, max_neurons=9,
                           min_samples_vertical_growth=150, spreading_factor=0.6),
}


def check_vertical_tree(est, g, rtol=1e-8):
    """Compare a fitted estimator's tree of maps (vertical growth, BaseSom.py:157-179) with the
    recorded one: same nodes carry children, every map has the recorded neurons / epochs /
    prototypes / QE / TE / labels."""
    import hashlib

    paths = [list(g["paths_flat"][g["paths_off"][k]:g["paths_off"][k + 1]]) for k in range(int(g["n_maps"]))]
    seen = []

    def walk(e, path):
        k = len(seen)
        seen.append(path)
        assert path == [int(v) for v in paths[k]], (path, paths[k])
        assert [tuple(n) for n in g[f"map{k}_neurons"]] == e.neurons_, path
        assert e.n_iter_ == int(g[f"map{k}_n_iter"]), path
        np.testing.assert_allclose(e.weights_, g[f"map{k}_weights"], rtol=rtol, atol=1e-10)
        np.testing.assert_allclose(e.quantization_error_, float(g[f"map{k}_qe"]), rtol=1e-9)
        assert e.topographic_error_ == float(g[f"map{k}_te"]), path
        assert len(e.labels_) == int(g[f"map{k}_n_samples"])
        sha = hashlib.sha256(np.ascontiguousarray(e.labels_, dtype=np.int64).tobytes()).hexdigest()
        assert sha == str(g[f"map{k}_labels_sha"]), path
        np.testing.assert_allclose(e.growing_threshold_, float(g[f"map{k}_threshold"]), rtol=1e-12)
        for i, node in enumerate(e.neurons_):
            child = e.som_.nodes[node].get("som")
            if child is not None:
                walk(child, path + [i])

    walk(est, [])
    assert len(seen) == len(paths)


def frozen_W(name, X):
    g = load(name)
    return X[g["sel"]].astype(np.float64), int(g["rows"]), int(g["cols"])


def lattice_hops(rows, cols):
    """Hop (Manhattan) distances of a full rows x cols lattice, node order (i, j) row-major."""
    ii, jj = np.divmod(np.arange(rows * cols), cols)
    return (np.abs(ii[:, None] - ii[None, :]) + np.abs(jj[:, None] - jj[None, :])).astype(
        np.float64)
