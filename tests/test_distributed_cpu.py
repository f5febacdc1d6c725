"""Sample-sharded path on CPU: world_size-2 (and 3) gloo groups must reproduce the single-process
result -- identical BMUs, sums equal up to float64 reassociation."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from dbgsom_amd import SomVQ
from dbgsom_amd.backend import shard_bounds
from oracle.som_oracle import OracleBackend
from tests import golden_inputs as gi

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(world, tmp_path):
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r),
                               str(world), port, outs[r]], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [np.load(o) for o in outs]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_epoch_and_fit_match_single_process(world, tmp_path):
    res = _run(world, tmp_path)
    X, _ = gi.blobs_f32(6001, 40, 21)
    rows, cols = 5, 6
    M = rows * cols
    W = X[np.random.default_rng(3).choice(len(X), M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    one = OracleBackend().load(X).epoch(W, hop, 1.1, 0.002, "compact", True)
    winners = np.concatenate([r["winners"] for r in res])
    assert np.array_equal(winners, one.winners)                 # BMUs do not depend on G
    for r in res:
        assert np.array_equal(r["activations"], one.activations)
        np.testing.assert_allclose(r["new_weights"], one.new_weights, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(r["errors"], one.errors, rtol=1e-12)
        np.testing.assert_allclose(r["change_total"], one.change_total, rtol=1e-10)
        assert np.array_equal(r["new_weights"], res[0]["new_weights"])  # ranks agree bitwise
    # bounds tile [0, N)
    assert [(int(r["lo"]), int(r["hi"])) for r in res] == \
        [shard_bounds(len(X), k, world) for k in range(world)]
    Xf, _ = gi.case_X("lowd_linear")
    ref = SomVQ(backend=OracleBackend(), **gi.EST_KWARGS["lowd_linear"]).fit(Xf)
    for r in res:
        assert np.array_equal(r["fit_labels"], ref.labels_)
        np.testing.assert_allclose(r["fit_weights"], ref.weights_, rtol=1e-9, atol=1e-11)
        assert int(r["fit_n_iter"]) == ref.n_iter_
        assert float(r["fit_te"]) == ref.topographic_error_
