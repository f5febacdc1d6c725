"""Host-side estimator core: the scikit-learn surface of the reference's ``BaseSom`` with the
per-epoch hot path delegated to a backend (MI355X by default, no CPU fallback).

Mirrors ``dbgsom/BaseSom.py`` of SandroMartens/DBGSOM: same constructor parameters (spelling
``convergence_treshold`` included, :42-80), same fitted attributes, same private hot-path methods
(``_get_winning_neurons``, ``_calculate_exp_similarity``, ``_update_weights``,
``_write_accumulative_error``) so call sites and tests read like the reference's.  Topology and
growth stay on the host (``lattice.GrowingLattice`` on NetworkX); only dense distance / update
arrays go to the device.

Extra, build-only parameters (defaults keep reference behaviour):
    backend         None -> ``HipBackend()``; or a ``HotPathBackend`` instance (tests inject the
                    oracle's CPU stand-in)
    centres_layout  "compact" reproduces the reference's compacted Voronoi-centre rows (quirk
                    Q1, BaseSom.py:1045,1053); "aligned" is the mathematically intended form
    device          HIP device ordinal for the default backend
    sharded_input   False: every rank of a torch.distributed group passes the SAME full X and keeps
                    its row shard resident; True: every rank passes only ITS rows (nothing of size
                    N is ever gathered; ``labels_`` then describes the local rows)
"""
from __future__ import annotations

import copy
from math import log, pi, sqrt

import networkx as nx
import numpy as np
import scipy.spatial.distance
import scipy.stats
from sklearn.base import BaseEstimator, clone
from sklearn.utils import check_array, check_random_state
from sklearn.utils.validation import check_is_fitted

from . import schedule
from .backend import RESIDENT, HotPathBackend, dist_info, shard_bounds
from .lattice import GrowingLattice


class DeviceSamples:
    """Samples that live in HBM only (a Voronoi subset gathered on the device): what ``fit`` and
    ``predict`` see in place of a NumPy array.  Carries shape and dtype; rows are fetched on
    demand."""

    def __init__(self, backend):
        self.backend = backend
        self.shape = (backend.n_samples, backend._d)
        self.dtype = backend._x_np_dtype if not isinstance(backend._x_np_dtype, str) else np.dtype(np.float32)
        self.ndim = 2

    def __len__(self):
        return self.shape[0]


class BaseSom(BaseEstimator):
    def __init__(
        self,
        n_iter: int = 200,
        convergence_iter: int = 1,
        spreading_factor: float = 0.5,
        sigma_start=None,
        sigma_end=None,
        vertical_growth: bool = False,
        decay_function: str = "exponential",
        learning_rate: float = 0.02,
        verbose: bool = False,
        coarse_training_frac: float = 0.5,
        random_state=None,
        convergence_treshold: float = 10 ** -5,
        max_neurons: int = 100,
        metric: str = "euclidean",
        threshold_method: str = "se",
        growth_criterion: str = "quantization_error",
        min_samples_vertical_growth: int = 100,
        n_jobs: int = 1,
        backend=None,
        centres_layout: str = "compact",
        device=None,
        sharded_input: bool = False,
    ) -> None:
        self.n_iter = n_iter
        self.convergence_iter = convergence_iter
        self.spreading_factor = spreading_factor
        self.sigma_start = sigma_start
        self.sigma_end = sigma_end
        self.vertical_growth = vertical_growth
        self.decay_function = decay_function
        self.learning_rate = learning_rate
        self.verbose = verbose
        self.coarse_training_frac = coarse_training_frac
        self.random_state = random_state
        self.convergence_treshold = convergence_treshold
        self.max_neurons = max_neurons
        self.metric = metric  # stored, never read: distances are always Euclidean (as in the reference)
        self.threshold_method = threshold_method
        self.growth_criterion = growth_criterion
        self.min_samples_vertical_growth = min_samples_vertical_growth
        self.n_jobs = n_jobs
        self.backend = backend
        self.centres_layout = centres_layout
        self.device = device
        self.sharded_input = sharded_input

    # ------------------------------------------------------------------------------------------
    # backend plumbing
    # ------------------------------------------------------------------------------------------
    def _make_backend(self) -> HotPathBackend:
        if self.backend is None:
            from .backend import HipBackend

            return HipBackend(self.device)  # raises without the built extension / a GPU
        if isinstance(self.backend, HotPathBackend):
            return self.backend
        raise TypeError("backend must be None or a HotPathBackend instance")

    def _engine(self) -> HotPathBackend:
        be = getattr(self, "_backend_obj", None)
        if be is None:
            be = self._backend_obj = self._make_backend()
        return be

    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        state = dict(state)
        state.pop("_backend_obj", None)  # device handles are not picklable
        state.pop("_resident", None)
        return state

    # ------------------------------------------------------------------------------------------
    # fit
    # ------------------------------------------------------------------------------------------
    def fit(self, X, y=None):
        """Train the map on X (BaseSom.fit, BaseSom.py:88-131)."""
        if isinstance(X, DeviceSamples):   # a Voronoi subset that already lives in HBM (f-4)
            if y is not None:
                y = np.asarray(y)
        else:
            X, y = self._check_input_data(X, y)
        if y is not None:
            classes, y = np.unique(y, return_inverse=True)
            self.classes_ = np.array(classes)
        self.random_state_ = check_random_state(self.random_state)
        engine = self._engine()
        self._load_resident(X)  # samples go to HBM once and stay there for the whole fit
        try:
            self._initialize_som(X)
            self._grow_som(X, y)
            self.topographic_error_ = self._calculate_topographic_error(X)
            self.quantization_error_ = self.calculate_quantization_error(X)
            self.n_features_in_ = X.shape[1]
            self._write_node_statistics(X)
            self._delete_dead_neurons_from_graph(X)
            self._label_prototypes(X, y)
            if self.vertical_growth:
                self._grow_vertical(X, y)
            self._fit(X)
            self.n_iter_ = self._current_epoch
        finally:
            self._resident = None
            engine.release()
        return self

    def _load_resident(self, X) -> None:
        """Make this rank's rows resident in HBM.  Default: every rank holds the same X and
        uploads its contiguous row shard (one process per GPU); ``sharded_input``: X already is
        this rank's shard; a ``DeviceSamples`` is resident as it is."""
        rank, world = dist_info()
        self._n_total = int(X.shape[0])
        if isinstance(X, DeviceSamples):
            self._shard = (0, X.shape[0])
        elif self.sharded_input and world > 1:
            sizes = self._all_gather_ints(X.shape[0])
            lo = int(sum(sizes[:rank]))
            self._shard = (lo, lo + X.shape[0])     # position of the local rows in the global order
            self._n_total = int(sum(sizes))
            self._engine().load(X)
        else:
            self._shard = shard_bounds(X.shape[0], rank, world)
            self._engine().load(X[self._shard[0]:self._shard[1]])
        self._resident = X

    def _local_input(self) -> bool:
        """True when the X handed to fit holds only this rank's rows."""
        return bool(self.sharded_input) and dist_info()[1] > 1

    @staticmethod
    def _all_gather_ints(value):
        import torch
        import torch.distributed as td

        world = td.get_world_size()
        dev = "cuda" if td.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(parts, mine)
        return [int(p.item()) for p in parts]

    @staticmethod
    def _all_reduce_f64(arr):
        """Element-wise sum of a small host array over the ranks."""
        import torch
        import torch.distributed as td

        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        if td.get_backend() == "nccl":
            t = t.cuda()
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t.cpu().numpy()

    def _check_input_data(self, X, y):
        raise NotImplementedError

    # -- input validation (SomVQ.py:122 / SomClassifier.py: check_array / check_X_y) -------------
    def _finite_check_on_device(self) -> bool:
        """Whether the "no NaN, no infinity" part of sklearn's input validation can ride on the column sums
        the initialisation takes from the resident samples anyway (one host pass over X saved: 0.15 s of a
        1.3 s fit at 1e6 x 784): the default backend, one process, the whole X resident."""
        from .backend import HipBackend

        return (self.backend is None or isinstance(self.backend, HipBackend)) and dist_info()[1] == 1 \
            and not self.sharded_input

    @staticmethod
    def _finite_kw(check: bool) -> dict:
        """`ensure_all_finite` (scikit-learn >= 1.6) / `force_all_finite` (before) of check_array."""
        import inspect

        from sklearn.utils import check_array

        name = "ensure_all_finite" if "ensure_all_finite" in inspect.signature(check_array).parameters \
            else "force_all_finite"
        return {name: check}

    def _assert_finite_from_moments(self, X, mom) -> None:
        """The deferred half of the validation: a NaN or an infinity anywhere in a column shows in the column's
        sum (NaN / +-inf / NaN for +inf and -inf together).  Only when a sum is not finite is sklearn's own
        check run -- it raises its usual ValueError, or passes (finite values whose sum overflowed)."""
        if not getattr(self, "_finite_deferred", False):
            return
        self._finite_deferred = False
        if mom is None or not (np.isfinite(mom[0]).all() and np.isfinite(mom[1]).all()):
            from sklearn.utils import assert_all_finite

            assert_all_finite(X)

    def _label_prototypes(self, X, y) -> None:
        raise NotImplementedError

    def _fit(self, X):
        pass

    def predict(self, X):
        raise NotImplementedError

    # -- initialisation (BaseSom.py:352-385, 419-444) -------------------------------------------
    def _initialize_som(self, data) -> None:
        self._current_epoch = 0
        self.converged_ = False
        self._training_phase = "coarse"
        engine = self._engine()
        rank, world = dist_info()
        n_total = self._n_total
        # np.var / np.std over the samples (two host passes over X, 1.4 s at 1e6 x 784) from the
        # resident copy when it is the whole data set: same values bit for bit (f-1)
        self._col_s2 = None
        on_device = isinstance(data, DeviceSamples)
        if self._local_input():
            # every rank holds its own rows: moments from two all-reduced passes in float64 (the
            # single-process values up to float64 reassociation -- not NumPy's sequential order)
            loc = np.asarray(data, dtype=np.float64)
            mean = self._all_reduce_f64(loc.sum(axis=0)) / n_total
            self._col_s2 = self._all_reduce_f64(((loc - mean) ** 2).sum(axis=0)).astype(data.dtype)
        elif self._shard == (0, data.shape[0]) and hasattr(engine, "column_moments"):
            mom = engine.column_moments()
            if not on_device:
                self._assert_finite_from_moments(data, mom)
            if mom is not None:
                self._col_s2 = mom[1]
        if getattr(self, "_finite_deferred", False):   # (no moments from the device after all: the host check now)
            self._assert_finite_from_moments(data, None)
        if self._col_s2 is None and on_device:
            raise ValueError("a DeviceSamples fit needs float32 / float64 resident samples")
        self.growing_threshold_ = self._calculate_growing_threshold(data)
        # keeps the dtype NumPy gives it: float32 data -> float32 variance -> float32 reciprocal
        if self._col_s2 is not None:
            self._total_variance = np.true_divide(self._col_s2, n_total).sum()
        else:
            self._total_variance = np.var(data, axis=0).sum()
        self._col_s2 = None
        seed = self.random_state
        if world > 1 and seed is None:
            # every rank must start from the same four prototypes: rank 0 draws the seed
            drawn = np.random.SeedSequence().entropy % (2 ** 62) if rank == 0 else 0
            seed = int(self._all_reduce_f64(np.array([float(drawn >> 31), float(drawn & (2 ** 31 - 1))]))
                       @ np.array([2.0 ** 31, 1.0]))
        rng = np.random.default_rng(seed=seed)
        if on_device or self._local_input():
            # rng.choice(a=data, size=4, replace=False) picks rows rng.choice(n, 4, replace=False)
            rows = rng.choice(n_total, size=4, replace=False)
            if on_device:
                start = engine.read_samples(rows).astype(data.dtype)
            else:
                lo, hi = self._shard
                start = np.zeros((4, data.shape[1]))
                for k, r in enumerate(rows):
                    if lo <= r < hi:
                        start[k] = data[r - lo]
                start = self._all_reduce_f64(start).astype(data.dtype)   # one owner per row: exact
        else:
            start = rng.choice(a=data, size=4, replace=False)
        self._lattice = GrowingLattice(start)
        self._sync_views(refresh_weights=True)

    def _calculate_growing_threshold(self, data: np.ndarray) -> float:
        if self.growth_criterion == "entropy":
            return self.spreading_factor
        if self.threshold_method == "classical":
            return -data.shape[1] * log(self.spreading_factor)
        if self.threshold_method == "se":
            if getattr(self, "_col_s2", None) is not None:
                spread = np.sqrt(np.true_divide(self._col_s2, max(self._n_total - 1, 0)))
            else:
                spread = np.std(data, axis=0, ddof=1)
            return float(150 * -log(self.spreading_factor) * np.linalg.norm(spread))
        raise ValueError("threshold_method not supported. Must be 'se' or 'classical'.")

    def _sync_views(self, refresh_weights: bool) -> None:
        lat = self._lattice
        self.som_ = lat.graph
        self.neurons_ = lat.nodes
        self._distance_matrix = lat.hop_distances()
        if refresh_weights:
            self.weights_ = np.array(lat.W)

    # -- epoch loop (BaseSom.py:387-417) --------------------------------------------------------
    def _grow_som(self, data, y) -> None:
        engine = self._engine()
        lat = self._lattice
        epochs = range(self.n_iter)
        if self.verbose:
            from tqdm import tqdm

            epochs = tqdm(iterable=epochs, unit=" epochs")
        need_assign = self.growth_criterion == "entropy"
        n_classes = 0
        if need_assign:
            if isinstance(data, DeviceSamples):
                # A device subset brought its PARENT's label codes along; this fit has re-coded y
                # (np.unique in fit), and the subset's rows are in the stable bucket order the
                # caller cut y_sub in (y[winners == j]): attach the child's own codes.
                engine.set_labels(y)
            else:
                lo, hi = (0, data.shape[0]) if self._local_input() else self._shard
                engine.set_labels(y[lo:hi])
            n_classes = int(self.classes_.shape[0])
        # The prototypes live in HBM for the whole fit: the first epoch uploads
        # the four start vectors, every later one consumes what the previous one left there; a
        # growth step writes only the inserted rows (and the new hop matrix); the host copy is
        # refreshed at growth steps and at the end.  Backends without resident prototypes (the
        # oracle's CPU stand-in in the tests) get the matrix handed over every epoch.
        resident = hasattr(engine, "write_weight_rows")
        on_device = False     # the current prototypes are in HBM, lat.W is stale
        ran = False
        self._growth_epochs = []
        for epoch in epochs:
            self._current_epoch = epoch
            if epoch > self.coarse_training_frac * self.n_iter:
                self._training_phase = "fine"
            self._sync_views(refresh_weights=not on_device)  # hop matrix recomputed only after growth
            w_in = RESIDENT if on_device else self.weights_

            res = engine.epoch(w_in, self._distance_matrix, self._calculate_current_sigma(),
                               self._gamma(), self.centres_layout,
                               n_classes=n_classes if need_assign else 0,
                               **({"keep_on_device": True} if resident else {}))
            ran = True
            if resident:
                on_device = True
            else:
                lat.set_weights(res.new_weights)  # like the reference: the graph moves on, the
            if res.change_total < self.convergence_treshold:  # weights_ snapshot stays (Q3)
                self.converged_ = True
            if need_assign:  # label entropy per neuron from the (M, n_classes) histogram
                lat.set_errors(np.array([scipy.stats.entropy(row[: np.flatnonzero(row).max() + 1]
                                                             if row.any() else row[:0], base=2)
                                         for row in res.class_hist]))
            else:
                lat.set_errors(res.errors)

            if self.converged_ and self._training_phase == "fine":
                break
            if (self._training_phase == "coarse" and len(self.neurons_) < self.max_neurons
                    and epoch % self.convergence_iter == self.convergence_iter - 1):
                lat.distribute_errors(self.growing_threshold_)
                if on_device:
                    if not lat.will_grow(self.growing_threshold_):
                        continue
                    lat.set_weights(engine.get_weights(0))   # growth extrapolates from W'
                    m_before = len(lat)
                    lat.grow(self.growing_threshold_, epoch)
                    self._growth_epochs.append(epoch)
                    for i in lat.pop_overwritten():           # occupied positions (rare)
                        engine.write_weight_rows(i, lat.W[i])
                    if len(lat) > m_before:                   # the inserted rows only
                        engine.write_weight_rows(m_before, lat.W[m_before:])
                else:
                    lat.grow(self.growing_threshold_, epoch)
        if on_device and ran:
            self.weights_ = engine.get_weights(1)   # the snapshot the last epoch consumed (Q3)
            lat.set_weights(engine.get_weights(0))
        if hasattr(engine, "traffic"):
            self._training_traffic = engine.traffic()   # what crossed PCIe during the epoch loop
        lat.write_attributes()

    def _gamma(self) -> float:
        return float(self._total_variance ** -1)

    def _calculate_current_sigma(self) -> float:
        return schedule.current_sigma(
            epoch=self._current_epoch, n_neurons=len(self._lattice), n_iter=self.n_iter,
            phase=self._training_phase, decay_function=self.decay_function,
            learning_rate=self.learning_rate, coarse_training_frac=self.coarse_training_frac,
            sigma_start=self.sigma_start, sigma_end=self.sigma_end)

    def _entropy_errors(self, winners, y):
        out = np.zeros(len(self.neurons_))
        for j in range(len(self.neurons_)):
            out[j] = scipy.stats.entropy(np.bincount(y[winners == j]), base=2)
        return out

    # ------------------------------------------------------------------------------------------
    # the reference's four hot-path methods, same names and argument meaning
    # ------------------------------------------------------------------------------------------
    def _is_resident(self, data) -> bool:
        return getattr(self, "_resident", None) is data

    def _gather_rows(self, local):
        """Concatenate per-rank row shards (identity for one process, and when every rank was
        given only its own rows): ONE padded tensor all_gather, no pickling."""
        rank, world = dist_info()
        if world == 1 or self._local_input():
            return local
        import torch
        import torch.distributed as td

        n = self._n_total
        bounds = [shard_bounds(n, r, world) for r in range(world)]
        longest = max(hi - lo for lo, hi in bounds)
        local = np.ascontiguousarray(local)
        pad = np.zeros((longest,) + local.shape[1:], dtype=local.dtype)
        pad[: local.shape[0]] = local
        mine = torch.from_numpy(pad)
        if td.get_backend() == "nccl":
            mine = mine.cuda()
        parts = [torch.empty_like(mine) for _ in range(world)]
        td.all_gather(parts, mine)
        return np.concatenate([p.cpu().numpy()[: hi - lo] for p, (lo, hi) in zip(parts, bounds)],
                              axis=0)

    def _get_winning_neurons(self, data, n_bmu: int):
        """Distances and indices of the n_bmu best matching units (BaseSom.py:446-464)."""
        engine = self._engine()
        if self._is_resident(data):
            dist, idx = engine.bmu(self.weights_, n_bmu)
            return self._gather_rows(dist), self._gather_rows(idx)
        return engine.bmu(self.weights_, n_bmu, X=data)

    def _calculate_exp_similarity(self, distances):
        """Per-sample weight 1 - sqrt(1 - exp(-gamma d^2)) (BaseSom.py:533-538)."""
        return self._engine().exp_similarity(distances, self._gamma())

    def _update_weights(self, sample_weights, winners, data) -> None:
        """Batch update of all prototypes (BaseSom.py:470-523) on the resident samples."""
        if not self._is_resident(data):
            self._load_resident(data)
        lo, hi = (0, len(winners)) if self._local_input() else self._shard
        winners = np.asarray(winners)[lo:hi]
        Wn, chg, _, _ = self._engine().update(
            self.weights_, self._distance_matrix, self._calculate_current_sigma(),
            np.asarray(sample_weights)[lo:hi], winners, np.zeros(len(winners)),
            self.centres_layout)
        self._lattice.set_weights(Wn)
        if chg < self.convergence_treshold:
            self.converged_ = True
        self._lattice.write_attributes()

    def _write_accumulative_error(self, winners, y, distances) -> None:
        """Per-neuron error = sum of BMU distances, or label entropy (BaseSom.py:541-561)."""
        if self.growth_criterion == "entropy":
            errors = self._entropy_errors(np.asarray(winners), y)
        else:
            errors = np.bincount(winners, weights=distances, minlength=len(self.neurons_))
        self._lattice.set_errors(errors)
        self._lattice.write_attributes()

    # ------------------------------------------------------------------------------------------
    # post-fit statistics (BaseSom.py:181-235, 904-953)
    # ------------------------------------------------------------------------------------------
    def calculate_quantization_error(self, X) -> float:
        """Average distance from each sample to its nearest prototype."""
        check_is_fitted(self)
        if self._is_resident(X):  # during fit: a device reduction, distances never leave HBM
            return self._engine().quantization_error(self.weights_)
        X = check_array(X, dtype=[np.float64, np.float32])
        distances, _ = self._get_winning_neurons(X, n_bmu=1)
        return float(np.mean(distances))

    def _calculate_topographic_error(self, X) -> float:
        """Fraction of samples whose two best matching units are not lattice neighbours."""
        if self._is_resident(X):
            return self._engine().topographic_error_count(self.weights_, self.neurons_) / self._n_total
        _, bmu = self._get_winning_neurons(X, n_bmu=2)
        pos = np.asarray(self.neurons_, dtype=np.float64)
        apart = np.linalg.norm(pos[bmu[:, 0]] - pos[bmu[:, 1]], axis=1) > 1.5
        return int(np.count_nonzero(apart)) / X.shape[0]

    def _get_u_matrix(self) -> np.ndarray:
        """Mean input-space distance from every prototype to the list of all neighbour
        prototypes (the reference averages over the WHOLE concatenated list, :320-337)."""
        lat = self._lattice
        nbr_rows = [lat.index_of(nb) for nbrs in lat.graph.adj.values() for nb in nbrs]
        return scipy.spatial.distance.cdist(lat.W, lat.W[nbr_rows]).mean(axis=1)

    def _calculate_node_statistics(self, X):
        average_distances = self._get_u_matrix()
        sigma = average_distances.mean()
        m = len(self._lattice)  # neurons inserted in the very last epoch count as dead
        if self._is_resident(X):
            hits, sums = self._engine().node_statistics(self.weights_, sigma)
        else:
            distances, winners = self._get_winning_neurons(X, n_bmu=1)
            hits = np.bincount(winners, minlength=len(self.weights_)).astype(np.float64)
            dens = np.exp(-(distances ** 2) / (2 * sigma ** 2)) / (sigma * sqrt(2 * pi))
            sums = np.bincount(winners, weights=dens, minlength=len(self.weights_))
        hit_counts, dens_sums = np.zeros(m), np.zeros(m)
        hit_counts[: len(hits)], dens_sums[: len(sums)] = hits, sums
        densities = np.divide(dens_sums, hit_counts, out=np.zeros(m), where=hit_counts > 0)
        return average_distances, densities, hit_counts

    def _write_node_statistics(self, X) -> None:
        avg, dens, hits = self._calculate_node_statistics(X)
        self._node_stats = {"density": dens, "hit_count": hits, "average_distance": avg}
        self._lattice.write_attributes(self._node_stats)

    def _delete_dead_neurons_from_graph(self, X) -> None:
        lat = self._lattice
        hits = self._node_stats["hit_count"]
        keep = hits != 0
        dead = [n for n, h in zip(lat.nodes, hits) if h == 0]
        lat.remove(dead)
        self._node_stats = {k: v[keep] for k, v in self._node_stats.items()}
        lat.write_attributes(self._node_stats)
        self._sync_views(refresh_weights=True)  # weights_ now holds the last update

    def _extract_values_from_graph(self, attribute: str) -> np.ndarray:
        return np.array([data[attribute] for _, data in self.som_.nodes.data()])

    # ------------------------------------------------------------------------------------------
    # parts of the surface outside the accelerated path (kept for drop-in completeness)
    # ------------------------------------------------------------------------------------------
    def transform(self, X, y=None) -> np.ndarray:
        """Non-negative LARS-lasso code of X over the prototypes (BaseSom.py:241-268;
        scikit-learn's SparseCoder on the host -- a different algorithm, not accelerated)."""
        from sklearn.decomposition import SparseCoder
        from sklearn.preprocessing import normalize

        check_is_fitted(self)
        X = check_array(X, dtype=[np.float64, np.float32])
        coder = SparseCoder(dictionary=normalize(self.weights_), n_jobs=self.n_jobs,
                            positive_code=True, transform_alpha=0,
                            transform_algorithm="lasso_lars")
        return coder.transform(normalize(X))

    def _grow_vertical(self, X, y=None) -> None:
        """Fit a child map on the Voronoi set of every neuron whose error exceeds 1.5x the
        growing threshold (BaseSom.py:157-179).  The reference's own loop cannot run: it compares
        the (node, error) tuple with a float (TypeError) and would index the graph by position;
        this is the evident intent, pinned by tests/golden/vertical_*.npz (made with exactly
        those two slips corrected, tools/make_golden.py).  On the MI355X the Voronoi sets are
        gathered in HBM (dbgsom_ctx_partition / dbgsom_ctx_subset_create): no X[mask] on the
        host, no second upload."""
        self.vertical_growing_threshold_ = 1.5 * self.growing_threshold_
        engine = self._engine()
        errors = self._lattice.error
        on_device = self._is_resident(X) and hasattr(engine, "subset") and dist_info()[1] == 1
        if on_device:
            counts, winners = engine.partition(self.weights_, want_winners=y is not None)
        else:
            _, winners = self._get_winning_neurons(X, n_bmu=1)
            counts = np.bincount(winners, minlength=len(self.neurons_))
        for j, node in enumerate(self.neurons_):
            if not errors[j] > self.vertical_growing_threshold_:
                continue
            if counts[j] > self.min_samples_vertical_growth:
                child = clone(self)
                y_sub = None if y is None else y[winners == j]
                if on_device:
                    sub = engine.subset(j)
                    child._backend_obj = sub
                    child.fit(DeviceSamples(sub), y_sub)
                else:
                    child.fit(X[winners == j], y_sub)
                self.som_.nodes[node]["som"] = child

    def plot(self, color=None, palette="magma_r", pointsize=None) -> None:
        """Scatter of the lattice coloured by a node attribute (needs seaborn >= 0.12)."""
        import pandas as pd
        import seaborn.objects as so

        data = pd.DataFrame(dict(self.som_.nodes)).T.reset_index(drop=True)
        for col in ("epoch_created", "error", "density", "hit_count", "average_distance"):
            if col in data:
                data[col] = pd.to_numeric(data[col])
        xy = pd.DataFrame(np.array(self.neurons_), columns=["x", "y"])
        so.Plot(pd.concat([xy, data], axis=1), x="x", y="y", color=color,
                pointsize=pointsize).add(so.Dot()).scale(color=palette).label(x="", y="").show()
