"""Where does a whole `fit` spend its time?  (host growth logic vs device epochs)"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbgsom_amd import SomVQ  # noqa: E402
from tests.golden_inputs import blobs_f32  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
max_neurons = int(sys.argv[3]) if len(sys.argv) > 3 else 600
n_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 80
X, _ = blobs_f32(n, d, 3)
est = SomVQ(random_state=0, max_neurons=max_neurons, n_iter=n_iter, spreading_factor=float(sys.argv[5]) if len(sys.argv) > 5 else 0.9,
            convergence_iter=2)
est.fit(X[:2000])  # warm up library / allocator
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
est.fit(X)
pr.disable()
t = time.perf_counter() - t0
print(f"fit: {t:.2f} s, epochs {est.n_iter_ + 1}, neurons {len(est.neurons_)}, "
      f"QE {est.quantization_error_:.4f} TE {est.topographic_error_:.4f}")
tr = est._training_traffic
M = len(est.neurons_)
print(f"prototype traffic over PCIe during the epoch loop ({est.n_iter_ + 1} epochs, {len(est._growth_epochs)} growth "
      f"steps, final M = {M}, M x d x 8 = {M * d * 8 / 1e6:.2f} MB): whole-matrix uploads {tr['w_upload_calls']} "
      f"({tr['w_upload_bytes'] / 1e6:.2f} MB), whole-matrix downloads {tr['w_download_calls']} "
      f"({tr['w_download_bytes'] / 1e6:.2f} MB), single rows written {tr['w_row_writes']}, read {tr['w_row_reads']}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
