"""The sample-sharded PRODUCT path (HipBackend, world > 1) on the one GPU of a test box: two (and
three) fresh processes, a gloo group, every rank its own dbgsom_ctx on GPU 0 with its row shard;
the all-reduce of the [S|K|a|E|status] buffer goes through the context's callback seam.  Compared
with the single-process HipBackend, the oracle and the reference's golden fits."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(world, tmp_path):
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker_hip.py"), str(r),
                               str(world), port, outs[r]], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [np.load(o) for o in outs]


@pytest.mark.parametrize("world", [2, 3])
def test_hip_backend_two_ranks_one_gpu(world, tmp_path):
    from dbgsom_amd import SomClassifier, SomVQ
    from dbgsom_amd.backend import HipBackend, shard_bounds
    from oracle import som_oracle as o

    res = _run(world, tmp_path)
    for tag, (N, d, rows, cols) in {"small": (6001, 40, 5, 6), "filt": (9000, 72, 13, 14),
                                    "prune": (9000, 72, 13, 14)}.items():
        X, _ = gi.blobs_f32(N, d, 21)
        M = rows * cols
        W = X[np.random.default_rng(3).choice(N, M, replace=False)].astype(np.float64)
        hop = gi.lattice_hops(rows, cols)
        y = (np.arange(N) % 4).astype(np.int32)
        one = HipBackend(0).load(X)
        one.set_labels(y)
        r1 = one.epoch(W, hop, 1.1, 0.002, "compact", True, n_classes=4)
        oo = o.epoch(X, W, hop, 1.1, np.float64(500.0), "compact", "chain")
        winners = np.concatenate([r[f"{tag}_winners"] for r in res])
        dists = np.concatenate([r[f"{tag}_distances"] for r in res])
        assert np.array_equal(winners, r1.winners) and np.array_equal(winners, oo.winners)  # BMUs do not depend on G
        assert np.array_equal(dists, r1.distances) and np.array_equal(dists, oo.distances)
        coords = [(i, j) for i in range(rows) for j in range(cols)]
        hits1, dens1 = one.node_statistics(W, 1.3)
        for r in res:
            assert bool(r[f"{tag}_filtered"]) == (tag != "small")
            assert tag != "prune" or int(r["prune_planes"]) == 0     # no sweep: candidates by pruning
            assert np.array_equal(r[f"{tag}_activations"], r1.activations)
            assert np.array_equal(r[f"{tag}_class_hist"], r1.class_hist)
            np.testing.assert_allclose(r[f"{tag}_new_weights"], r1.new_weights, rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(r[f"{tag}_new_weights"], oo.new_weights, rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(r[f"{tag}_errors"], r1.errors, rtol=1e-12)
            np.testing.assert_allclose(r[f"{tag}_change_total"], r1.change_total, rtol=1e-10)
            assert np.array_equal(r[f"{tag}_new_weights"], res[0][f"{tag}_new_weights"])  # ranks agree bitwise
            np.testing.assert_allclose(float(r[f"{tag}_qe"]), one.quantization_error(W), rtol=1e-13)
            assert int(r[f"{tag}_te"]) == one.topographic_error_count(W, coords)
            assert np.array_equal(r[f"{tag}_hits"], hits1)
            np.testing.assert_allclose(r[f"{tag}_dens"], dens1, rtol=1e-12, atol=1e-300)
            assert bool(r[f"{tag}_range_error"])        # every rank raised, none hung
            # smoothing sharded over the ranks: the same prototypes on every rank bit for bit; against the
            # replicated form bit for bit when the reduced sums are (two ranks: a + b in either collective)
            assert int(r[f"{tag}_shard_ran"]) == 1
            assert np.array_equal(r[f"{tag}_shard_new_weights"], res[0][f"{tag}_shard_new_weights"])
            assert np.array_equal(r[f"{tag}_shard_winners"], r[f"{tag}_winners"])
            assert np.array_equal(r[f"{tag}_shard_activations"], r1.activations)
            # what growth and convergence are decided from is the same on every rank, bit for bit, in both forms
            # and for any number of ranks (the small vectors are all-reduced, never reduce-scattered)
            for key in ("errors", "activations", "change_total", "shard_errors", "shard_activations", "shard_change_total"):
                assert np.array_equal(r[f"{tag}_{key}"], res[0][f"{tag}_{key}"]), (tag, key)
            if world == 2:
                assert np.array_equal(r[f"{tag}_shard_new_weights"], r[f"{tag}_new_weights"])
                assert float(r[f"{tag}_shard_change_total"]) == float(r[f"{tag}_change_total"])
                assert np.array_equal(r[f"{tag}_shard_errors"], r[f"{tag}_errors"])
            np.testing.assert_allclose(r[f"{tag}_shard_new_weights"], r1.new_weights, rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(r[f"{tag}_shard_errors"], r1.errors, rtol=1e-12)
            np.testing.assert_allclose(r[f"{tag}_shard_change_total"], r1.change_total, rtol=1e-10)
        one.release()
    for r in res:   # dbgsom_ctx_allreduce_host through the callback seam
        assert np.array_equal(r["host_sum"], [world * (world + 1) / 2, 10.0 * world, -0.5 * world * (world - 1) / 2])
    # whole fits: replicated X == the reference's golden fit; per-rank shards == the same map
    name = "lowd_linear"
    g = gi.load(name)
    Xf, _ = gi.case_X(name)
    ref = SomVQ(**gi.EST_KWARGS[name]).fit(Xf)
    for k, r in enumerate(res):
        assert np.array_equal(r["fit_labels"], g["final_labels"])
        np.testing.assert_allclose(r["fit_weights"], g["final_weights"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r["fit_weights"], ref.weights_, rtol=1e-9, atol=1e-11)
        assert int(r["fit_n_iter"]) == ref.n_iter_ == int(g["final_n_iter"])
        assert float(r["fit_te"]) == ref.topographic_error_
        np.testing.assert_allclose(float(r["fit_qe"]), ref.quantization_error_, rtol=1e-12)
        # sharded input: same map (moments reduced in float64 instead of NumPy's order), local labels
        lo, hi = shard_bounds(len(Xf), k, world)
        assert [tuple(n) for n in r["loc_neurons"]] == ref.neurons_
        assert int(r["loc_n_iter"]) == ref.n_iter_
        np.testing.assert_allclose(r["loc_weights"], ref.weights_, rtol=1e-6, atol=1e-8)
        assert np.array_equal(r["loc_labels"], ref.labels_[lo:hi])
        np.testing.assert_allclose(float(r["loc_qe"]), ref.quantization_error_, rtol=1e-6)
        assert float(r["loc_te"]) == ref.topographic_error_
        assert np.array_equal(r["rnd_weights"], res[0]["rnd_weights"])   # random_state=None: one seed
    Xc, yc = gi.case_X("digits_entropy")
    clf = SomClassifier(**gi.EST_KWARGS["digits_entropy"]).fit(Xc, yc)
    for r in res:
        assert int(r["clf_n_iter"]) == clf.n_iter_
        assert [tuple(n) for n in r["clf_neurons"]] == clf.neurons_
        np.testing.assert_allclose(r["clf_weights"], clf.weights_, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("G,d,rows,cols", [(2, 72, 13, 14), (3, 72, 13, 14), (8, 72, 13, 14), (8, 40, 5, 6), (12, 16, 5, 6),
                                           (8, 130, 20, 21)])
@pytest.mark.parametrize("layout", ["compact", "aligned"])
def test_sharded_smoothing_blocks_equal_the_replicated_form(G, d, rows, cols, layout):
    """Every rank's part of the sharded smoothing on ONE GPU: G virtual ranks, one after the other, through
    the callback seam (dbgsom_ctx_set_collectives).  With one real rank the reduce-scatter is the identity, so
    each virtual rank's block of W' comes from the full sums; a first pass collects the blocks, a second one
    hands every rank all of them at its all-gather.  The epoch's results must then equal the replicated
    smoothing's bit for bit -- for block widths that do not divide d, blocks beyond the last column, both
    layouts of the centres with dead neurons, maps with and without a split k range."""
    import torch

    from dbgsom_amd import _native
    from dbgsom_amd.backend import HipBackend, _DeviceArray

    N = 5000
    M = rows * cols
    X, _ = gi.blobs_f32(N, d, 33)
    rng = np.random.default_rng(5)
    W = X[rng.choice(N, M, replace=False)].astype(np.float64)
    W[::7] += 1e3          # dead neurons: rows nobody wins (quirk Q1 moves the centres in the compact layout)
    hop = gi.lattice_hops(rows, cols)
    one = HipBackend(0).load(X)
    ref = one.epoch(W, hop, 1.3, 0.004, layout, True)
    assert (ref.activations == 0).sum() >= M // 8
    one.release()
    dev = torch.device("cuda", 0)
    saved, ops = {}, []
    for pass_ in (0, 1):
        for r in range(G):
            be = HipBackend(0).load(X)

            def coll(_user, op, ptr, count, stream, r=r, pass_=pass_):
                try:
                    ops.append(op)
                    if op != _native.COLL_ALLGATHER:
                        return 0           # one real rank: sums and reduce-scattered blocks are already the totals
                    ext = torch.cuda.ExternalStream(stream, device=dev)
                    with torch.cuda.stream(ext):
                        t = torch.as_tensor(_DeviceArray(ptr, count * G), device=dev)
                        if pass_ == 0:
                            saved[r] = t[r * count:(r + 1) * count].clone()
                            t[:r * count].fill_(float("nan"))
                            t[(r + 1) * count:].fill_(float("nan"))
                        else:
                            for q in range(G):
                                if q != r:
                                    t[q * count:(q + 1) * count].copy_(saved[q])
                        ext.synchronize()
                    return 0
                except BaseException:  # noqa: BLE001
                    return 1

            cb = _native.COLLECTIVE_FN(coll)
            _native.call("dbgsom_ctx_set_collectives", be._ctx, cb, None, r, G)
            be.shard_smooth = 1
            res = be.epoch(W, hop, 1.3, 0.004, layout, True)
            assert be.shard_epochs == 1
            assert np.array_equal(res.winners, ref.winners) and np.array_equal(res.activations, ref.activations)
            assert np.array_equal(res.errors, ref.errors)
            if pass_ == 1:
                assert np.array_equal(res.new_weights, ref.new_weights, equal_nan=True)
                assert res.change_total == ref.change_total
            be.release()
    assert ops.count(_native.COLL_REDUCE_SCATTER) == 2 * G and ops.count(_native.COLL_ALLGATHER) == 2 * G
    assert ops.count(_native.COLL_ALLREDUCE) == 2 * G      # the small vectors [K | a | E | status], beside the blocks
