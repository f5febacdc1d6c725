"""Diagnostic: the arms the search policy runs on mid-clustered data, epoch by epoch, with the engine's clock.
    python tools/arms_trace.py [centres = 6] [epochs = 30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dbgsom_amd.backend import RESIDENT, HipBackend
from tests import golden_inputs as gi

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(21)
N, d, rows, cols = 160_000, 256, 32, 32
M = rows * cols
c = rng.normal(size=(nc, d)).astype(np.float32) * 4
X = c[rng.integers(0, nc, N)] + rng.normal(size=(N, d)).astype(np.float32)
W = X[rng.choice(N, M, replace=False)].astype(np.float64)
hop = gi.lattice_hops(rows, cols)
be = HipBackend(algorithm=sys.argv[3] if len(sys.argv) > 3 else "filtered").load(X)
be.set_weights(W)
for e in range(ne):
    t0 = time.perf_counter()
    be.epoch(RESIDENT, hop, 2.0, 1e-3, "compact", False, keep_on_device=True, frozen=True)
    ms = (time.perf_counter() - t0) * 1e3
    info = be.epoch_info()
    print(e, "arm", (2 if info[3] else (1 if info[7] else 0), int(info[2])), "lists %.1f" % info[1], "probe", info[6], "hold", int(info[5]),
          "%.3f ms" % ms, {k: round(v, 3) for k, v in be.arm_ms().items()})
