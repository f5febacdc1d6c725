"""Worker of tests/test_gpu_distributed.py: one rank of a world_size-N gloo group, every rank a
HipBackend (its own dbgsom_ctx) on GPU 0 -- the sharded product path on the one GPU a test box has."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch.distributed as td

    td.init_process_group("gloo", rank=rank, world_size=world)
    from dbgsom_amd import SomClassifier, SomVQ
    from dbgsom_amd.backend import HipBackend, shard_bounds
    from tests import golden_inputs as gi

    res = {}
    # (1) one epoch of the hot path on a row shard + all-reduce of the [S|K|a|E] sums, for a map
    # small enough for the all-pairs kernel and one that takes the filtered search
    # ("prune": the same with the filtered search forced to its form without a sweep)
    for tag, (N, d, rows, cols) in {"small": (6001, 40, 5, 6), "filt": (9000, 72, 13, 14),
                                    "prune": (9000, 72, 13, 14)}.items():
        X, _ = gi.blobs_f32(N, d, 21)
        M = rows * cols
        W = X[np.random.default_rng(3).choice(N, M, replace=False)].astype(np.float64)
        hop = gi.lattice_hops(rows, cols)
        lo, hi = shard_bounds(N, rank, world)
        be = HipBackend(0).load(X[lo:hi])
        if tag == "prune":
            be.sweep_planes = 4
        y = (np.arange(N) % 4).astype(np.int32)
        be.set_labels(y[lo:hi])
        r = be.epoch(W, hop, 1.1, 0.002, "compact", True, n_classes=4)
        res[f"{tag}_new_weights"], res[f"{tag}_change_total"] = r.new_weights, r.change_total
        res[f"{tag}_errors"], res[f"{tag}_activations"] = r.errors, r.activations
        res[f"{tag}_winners"], res[f"{tag}_distances"], res[f"{tag}_class_hist"] = r.winners, r.distances, r.class_hist
        res[f"{tag}_qe"] = be.quantization_error(W)
        coords = [(i, j) for i in range(rows) for j in range(cols)]
        res[f"{tag}_te"] = be.topographic_error_count(W, coords)
        hits, dens = be.node_statistics(W, 1.3)
        res[f"{tag}_hits"], res[f"{tag}_dens"] = hits, dens
        res[f"{tag}_filtered"] = bool(be.filter_log and be.filter_log[-1][0] == "filtered")
        res[f"{tag}_planes"] = int(be.filter_log[-1][2]) if res[f"{tag}_filtered"] else -1
        # the same epoch with the smoothing sharded over the ranks (reduce-scatter of column blocks of the sums,
        # every rank smooths d / world columns, all-gather of the new prototypes)
        be.shard_smooth = 1
        n0 = be.shard_epochs
        rs = be.epoch(W, hop, 1.1, 0.002, "compact", True, n_classes=4)
        res[f"{tag}_shard_ran"] = be.shard_epochs - n0
        res[f"{tag}_shard_new_weights"], res[f"{tag}_shard_change_total"] = rs.new_weights, rs.change_total
        res[f"{tag}_shard_errors"], res[f"{tag}_shard_activations"] = rs.errors, rs.activations
        res[f"{tag}_shard_winners"] = rs.winners
        if tag == "filt":   # (two legs in the default form, one with the sharded smoothing still on)
            be.shard_smooth = 2
            # small host vectors over the context's collective (what a caller without a communication
            # library of its own uses around the epochs)
            from dbgsom_amd import _native
            v = np.array([rank + 1.0, 10.0, -0.5 * rank])
            _native.call("dbgsom_ctx_allreduce_host", be._ctx, v.ctypes.data, v.size)
            res["host_sum"] = v
        # a winner out of range on ONE rank must fail on EVERY rank (status rides in the reduced buffer)
        bad = r.winners.copy()
        if rank == 0:
            bad[0] = M + 5
        try:
            be.update(W, hop, 1.1, np.ones(hi - lo), bad, r.distances)
            res[f"{tag}_range_error"] = False
        except Exception as e:  # noqa: BLE001
            res[f"{tag}_range_error"] = "out of range" in str(e)
        be.release()
    # (2) whole fits with the default backend: every rank holds X / every rank holds its rows only
    Xf, _ = gi.case_X("lowd_linear")
    est = SomVQ(**gi.EST_KWARGS["lowd_linear"]).fit(Xf)
    res.update(fit_weights=est.weights_, fit_labels=est.labels_, fit_qe=est.quantization_error_,
               fit_te=est.topographic_error_, fit_n_iter=est.n_iter_)
    lo, hi = shard_bounds(len(Xf), rank, world)
    loc = SomVQ(sharded_input=True, **gi.EST_KWARGS["lowd_linear"]).fit(Xf[lo:hi])
    res.update(loc_weights=loc.weights_, loc_labels=loc.labels_, loc_qe=loc.quantization_error_,
               loc_te=loc.topographic_error_, loc_n_iter=loc.n_iter_, loc_neurons=np.array(loc.neurons_))
    # (3) the entropy criterion (class histograms all-reduced every epoch)
    Xc, yc = gi.case_X("digits_entropy")
    clf = SomClassifier(**gi.EST_KWARGS["digits_entropy"]).fit(Xc, yc)
    res.update(clf_weights=clf.weights_, clf_n_iter=clf.n_iter_, clf_neurons=np.array(clf.neurons_))
    # (4) random_state=None: the ranks must still agree (rank 0 draws the seed)
    kw = dict(gi.EST_KWARGS["lowd_linear"], random_state=None, n_iter=6)
    rnd = SomVQ(**kw).fit(Xf)
    res.update(rnd_weights=rnd.weights_)
    np.savez(out, **res)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
