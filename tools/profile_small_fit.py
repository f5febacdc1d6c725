"""Host-side profile of a small fit (sklearn digits, 1797 x 64, ~25 neurons): at this size an epoch is
launch latency and Python, not kernels."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sklearn.datasets import load_digits  # noqa: E402

from dbgsom_amd import SomVQ  # noqa: E402

X = load_digits().data
SomVQ(random_state=0, n_iter=5).fit(X)
t0 = time.perf_counter()
est = SomVQ(random_state=0).fit(X)
t = time.perf_counter() - t0
print(f"digits fit: {t * 1e3:.1f} ms, {est.n_iter_ + 1} epochs ({t / (est.n_iter_ + 1) * 1e3:.3f} ms per epoch), "
      f"{len(est.neurons_)} neurons")
pr = cProfile.Profile()
pr.enable()
SomVQ(random_state=0).fit(X)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
