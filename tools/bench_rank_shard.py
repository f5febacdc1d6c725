"""What ONE rank of a multi-GPU bench run does per epoch, rehearsed on one GPU: prototypes drawn from rank
0's rows, samples = rank r's shard of the data set (rows=N: the first N rows of it -- a rank's share under
strong scaling).  With ranks=G the smoothing runs in its sharded form as rank r of G would run it (column
blocks of the sums, 1 / G of the M x M x d product, W' put together from G blocks); the two collectives are
stand-ins that return at once, so the figure is the rank's COMPUTE per epoch, without the wire.
    python tools/bench_rank_shard.py c4 rows=125000 ranks=8 0 1
    python tools/bench_rank_shard.py c5 ranks=8 0"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dbgsom_amd import _native  # noqa: E402
from dbgsom_amd.backend import RESIDENT, HipBackend  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
opts = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
ranks = [int(a) for a in sys.argv[2:] if "=" not in a] or [0, 1]
rows_take, G = int(opts.get("rows", 0)), int(opts.get("ranks", 1))
algo, steps = opts.get("algo", "filtered"), int(opts.get("steps", 20))
n, d, rows, cols, seed, kind, _ = bench.WORKLOADS[name]
M = rows * cols
dev = torch.device("cuda", 0)
X0 = bench.make_shard(torch, n, d, seed, dev)
g = torch.Generator(device=dev).manual_seed(seed + 7)
W = X0[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
gamma = float(1.0 / X0.double().var(dim=0, unbiased=False).sum().item())
del X0
hop = bench.lattice_hops(rows, cols)
noop = _native.COLLECTIVE_FN(lambda user, op, ptr, count, stream: 0)
for r in ranks:
    X = bench.make_shard(torch, n, d, seed, dev, rank=r)
    if rows_take:
        X = X[:rows_take].contiguous()
    if name in bench.BF16_WORKLOADS:
        X = X.to(torch.bfloat16)
    for sharded in ([False, True] if G > 1 else [False]):
        hip = HipBackend(0, algorithm=algo)
        hip.load_device(X)
        if sharded:
            _native.call("dbgsom_ctx_set_collectives", hip._ctx, noop, None, r % G, G)
            hip.shard_smooth = 1
        hip.set_weights(W)    # (resident prototypes: no upload inside the timed epochs)
        for _ in range(30):   # (the search policy settles: arms and the refinement are timed in the first epochs)
            hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            hip.epoch(RESIDENT, hop, 0.2 * np.sqrt(M), gamma, "compact", False, keep_on_device=True, frozen=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        c = hip.filter_counts()
        form = f"smoothing sharded as rank {r % G} of {G}" if sharded else "smoothing replicated"
        print(f"{name} rank {r} rows {X.shape[0]}: median {np.median(ts):.3f} ms/epoch (min {min(ts):.3f}), {form}, "
              f"candidates per workgroup mean {c.mean():.1f} max {c.max()}", flush=True)
        hip.release()
    del X
