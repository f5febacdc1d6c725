"""Whole fits on the MI355X through the default (HIP) backend against the reference's recorded
results -- the drop-in check for SomVQ / SomClassifier fit / predict / fit_predict."""
import numpy as np
import pytest

from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gi.FIT_CASES)
def test_fit_on_gpu_matches_reference(name):
    from dbgsom_amd import SomClassifier, SomVQ
    from dbgsom_amd.backend import HipBackend

    g = gi.load(name)
    X, y = gi.case_X(name)
    cls = SomClassifier if name in gi.CLF_CASES else SomVQ
    est = cls(**gi.EST_KWARGS[name])  # backend=None -> HipBackend
    est.fit(X, y) if y is not None else est.fit(X)
    assert isinstance(est._engine(), HipBackend)
    assert est.n_iter_ == int(g["final_n_iter"])
    assert [tuple(n) for n in g["final_neurons"]] == est.neurons_
    np.testing.assert_allclose(est.weights_, g["final_weights"], rtol=1e-5)  # north_star bound
    np.testing.assert_allclose(est.weights_, g["final_weights"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(est.quantization_error_, float(g["final_qe"]), rtol=1e-10)
    assert est.topographic_error_ == float(g["final_te"])
    if name not in gi.CLF_CASES:
        assert np.array_equal(est.labels_, g["final_labels"])   # BMU indices bit-exact
        assert np.array_equal(est.predict(X), g["final_labels"])
        assert np.array_equal(cls(**gi.EST_KWARGS[name]).fit_predict(X), g["final_labels"])
    else:
        assert np.array_equal(est.predict(X), g["final_predict"])
        assert est.score(X, y) == float(g["final_score"])


def test_known_answers_digits_gpu():
    from dbgsom_amd import SomVQ

    X, _ = gi.case_X("digits_f64")
    est = SomVQ(random_state=0).fit(X)
    assert est.n_iter_ == 112 and len(est.neurons_) == 25
    np.testing.assert_allclose(est.quantization_error_, 23.80011502226172, rtol=1e-12)
    assert est.topographic_error_ == 0.05008347245409015
    np.testing.assert_allclose(est.weights_.sum(), 7793.246057345110, rtol=1e-11)
    assert est.labels_[:10].tolist() == [23, 16, 6, 18, 19, 14, 17, 2, 21, 5]


def test_bench_under_torchrun_single_rank_rccl():
    """Rehearsal of the multi-GPU launch on a one-GPU box: torch.distributed.run with one rank,
    the RCCL group is created and the per-epoch all-reduce is actually issued
    (DBGSOM_FORCE_COLLECTIVE=1)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DBGSOM_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "c2", "--cpu-sample", "0",
           "--fine-phase", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    js = json.loads(line)
    assert js["n_gpus"] == 1 and js["value"] > 0 and js["roofline"]["frac"] > 0


def test_full_size_c3_and_c2_properties():
    """BASELINE configs C3 (N=1e6, d=128, M=2025) and C2 (60k x 784, M=506) at full size:
    oracle spot check + conservation."""
    import torch

    from dbgsom_amd.backend import HipBackend
    from oracle import som_oracle as o

    import bench

    for name in ("c3", "c2"):
        n, d, rows, cols, seed, _, _ = bench.WORKLOADS[name]
        M = rows * cols
        dev = torch.device("cuda", 0)
        hip = HipBackend(0, algorithm="exact")
        X = bench.make_shard(torch, n, d, seed, dev)
        hip.load_device(X)
        g = torch.Generator(device=dev).manual_seed(seed + 7)
        W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
        gamma = float(1.0 / X.double().var(dim=0, unbiased=False).sum().item())
        res = hip.epoch(W, bench.lattice_hops(rows, cols), 0.2 * np.sqrt(M), gamma, "compact", True)
        pick = np.random.default_rng(1).choice(n, 2000, replace=False)
        Xs = X[torch.from_numpy(pick).to(dev)].cpu().numpy()
        rd, ri = o.bmu_chain(Xs, W, 1)
        assert np.array_equal(res.winners[pick], ri) and np.array_equal(res.distances[pick], rd)
        assert res.activations.sum() == n
        np.testing.assert_allclose(res.errors.sum(), res.distances.sum(), rtol=1e-10)
        hip.release()
