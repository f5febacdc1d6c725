"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-N gloo group (CPU)."""
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
    from dbgsom_amd import SomVQ
    from dbgsom_amd.backend import shard_bounds
    from oracle.som_oracle import OracleBackend
    from tests import golden_inputs as gi

    # (1) one epoch of the hot path on a row shard + all-reduce of the [S|K|a|E] sums
    X, _ = gi.blobs_f32(6001, 40, 21)
    rows, cols = 5, 6
    M = rows * cols
    W = X[np.random.default_rng(3).choice(len(X), M, replace=False)].astype(np.float64)
    hop = gi.lattice_hops(rows, cols)
    lo, hi = shard_bounds(len(X), rank, world)
    be = OracleBackend().load(X[lo:hi])
    res = be.epoch(W, hop, 1.1, 0.002, "compact", True)
    # (2) a whole fit: every rank holds X, the backend keeps only its shard resident
    Xf, _ = gi.case_X("lowd_linear")
    est = SomVQ(backend=OracleBackend(), **gi.EST_KWARGS["lowd_linear"]).fit(Xf)
    np.savez(out, new_weights=res.new_weights, change_total=res.change_total, errors=res.errors,
             activations=res.activations, winners=res.winners, lo=lo, hi=hi,
             fit_weights=est.weights_, fit_labels=est.labels_, fit_qe=est.quantization_error_,
             fit_te=est.topographic_error_, fit_n_iter=est.n_iter_)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
