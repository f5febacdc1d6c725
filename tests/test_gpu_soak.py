"""Long walks of the engine's policy state machine inside the driver-run suite (`-m gpu`): what
tools/soak.py and tools/stress_filtered.py do by hand, at sizes that fit a minute.

The filtered search chooses its form per epoch (arms = seeds x digit planes or pruning, holds, trial arms,
back-off to all pairs, re-probes as the map moves, refinement by measurement, growth steps on the resident
prototypes); none of that may change a result.  The check is the strongest one available at these sizes:
the same chain of epochs once more with the all-pairs search must end in bit-identical prototypes, and a
sample of the final winners must equal the oracle's chain (the reference's arithmetic, BaseSom.py:446-464)."""
import importlib.util
import os

import numpy as np
import pytest

from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(f"tools_{name}", os.path.join(ROOT, "tools", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _chain(algo, X, W0, rows, cols, epochs, grow_at, gamma, refine=None):
    """`epochs` chained epochs on resident, evolving prototypes (sigma decaying to the fine phase's), the lattice
    growing by one row of extrapolated prototypes at the epochs `grow_at` (BaseSom.py:616-646: 2 w - w_neighbour)."""
    from dbgsom_amd.backend import RESIDENT, HipBackend

    be = HipBackend(0, algorithm=algo).load(X)
    if refine is not None:
        be.refine = refine
    be.set_weights(W0)
    hop = gi.lattice_hops(rows, cols)
    arms = {}
    for e in range(epochs):
        if e in grow_at:
            W = be.get_weights(0)
            W = np.vstack([W, 2.0 * W[-cols:] - W[-2 * cols:-cols]])
            rows += 1
            hop = gi.lattice_hops(rows, cols)
            be.set_weights(W)
        be.epoch(RESIDENT, hop, max(0.8, 3.5 * 0.97 ** e), gamma, "aligned", False, keep_on_device=True)
        if be.filter_log:
            last = be.filter_log[-1]
            key = last[0] if last[0] != "filtered" else f"filtered/{last[2]}" + ("+refined" if be.refined else "")
            arms[key] = arms.get(key, 0) + 1
    W = be.get_weights(0)
    dist, win = be.bmu(W, 1)
    be.release()
    return W, win, dist, arms, rows


@pytest.mark.parametrize("case", ["blobs_auto", "blobs_refine", "iso_auto"])
def test_soak_evolving_map_with_growth_equals_the_all_pairs_chain(case):
    from oracle import som_oracle as o

    n, d, rows, cols, epochs = 200_000, 128, 18, 20, 150
    if case == "iso_auto":   # data without clusters: sweeps with several digit planes, trial arms, back-off
        n, epochs = 120_000, 60
        X = np.random.default_rng(21).standard_normal((n, d)).astype(np.float32)
    else:
        X, _ = gi.blobs_f32(n, d, 20, n_centers=40)
    W0 = X[np.random.default_rng(27).choice(n, rows * cols, replace=False)].astype(np.float64)
    gamma = float(1.0 / np.var(X, axis=0, dtype=np.float64).sum())
    grow_at = {40, 80} if case != "iso_auto" else {25}
    refine = 1 if case == "blobs_refine" else None
    Wf, win, dist, arms, rows_f = _chain("auto", X, W0, rows, cols, epochs, grow_at, gamma, refine)
    We, win_e, dist_e, _, _ = _chain("exact", X, W0, rows, cols, epochs, grow_at, gamma)
    assert rows_f == rows + len(grow_at) and Wf.shape[0] == rows_f * cols
    assert np.array_equal(Wf, We, equal_nan=True), arms
    assert np.array_equal(win, win_e) and np.array_equal(dist, dist_e)
    filtered = sum(v for k, v in arms.items() if k.startswith("filtered"))
    assert filtered >= epochs // 2, arms                       # the policy did run the filtered search
    if case == "blobs_refine":
        assert any("refined" in k for k in arms), arms
    if not np.isnan(Wf).any():   # (dead neurons of the aligned layout are NaN rows: the chain above covers those)
        pick = np.random.default_rng(3).choice(n, 4000, replace=False)
        rd, ri = o.bmu_chain(X[pick], Wf, 1)
        assert np.array_equal(win[pick], ri) and np.array_equal(dist[pick], rd)


@pytest.mark.parametrize("seed,refine", [(101, False), (102, True)])
def test_stress_batch_of_random_shapes(seed, refine):
    """tools/stress_filtered.py: 20 random cases per batch (shapes, storage types, seed strides, sweep variants,
    runs of duplicated prototypes), filtered search against the all-pairs kernel, two epochs each."""
    stress = _tool("stress_filtered")
    assert stress.run(seed, 20, refine=refine, max_rows=20000, verbose=False) == 0
