"""Whole fits on the MI355X through the default (HIP) backend against the reference's recorded
results -- the drop-in check for SomVQ / SomClassifier fit / predict / fit_predict."""
import numpy as np
import pytest

from tests import golden_inputs as gi


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

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


@pytest.mark.parametrize("mode", ["default", "filtered", "refine", "prune"])
@pytest.mark.parametrize("name", gi.GROW_CASES)
def test_growing_fit_through_the_filtered_search_matches_the_reference(name, mode):
    """Fits the REFERENCE ran (tools/make_golden.py GROW_CASES; its loop BaseSom.py:387-417, growth :411-417,
    :588-614) that grow to 247 / 224 neurons with up to a third of them dead: from 129 prototypes on the estimator's
    default search goes through the engine's policy (previous winners as seeds, the arms of the filtered search, the
    back-off to all pairs while the lists are most of the map, growth steps on the resident prototypes).  The whole
    fit must end in the recorded map -- default backend; the stateless filtered form; the hinted form with the
    per-sample refinement forced on; the pruning form forced -- and the log must show that the filtered search ran."""
    from dbgsom_amd import SomVQ
    from dbgsom_amd.backend import HipBackend

    g = gi.load(name)
    X, _ = gi.case_X(name)
    if mode == "default":
        est = SomVQ(**gi.EST_KWARGS[name])
    else:
        be = HipBackend(algorithm={"filtered": "filtered", "refine": "filtered_hint", "prune": "filtered_hint"}[mode])
        if mode == "refine":
            be.refine = 1
        if mode == "prune":
            be.sweep_planes = 4
        est = SomVQ(backend=be, **gi.EST_KWARGS[name])
    log = []
    be = est._engine()
    assert isinstance(be, HipBackend) and be.algorithm == ("auto" if mode == "default" else be.algorithm)
    orig = be._log_epoch

    def spy():
        orig()
        log.append((be._last_M,) + tuple(be.filter_log[-1]) + (be.refined,))

    be._log_epoch = spy
    est.fit(X)
    assert est.n_iter_ == int(g["final_n_iter"])
    assert [tuple(n) for n in g["final_neurons"]] == est.neurons_
    np.testing.assert_allclose(est.weights_, g["final_weights"], rtol=1e-8, atol=1e-10)
    assert np.array_equal(est.labels_, g["final_labels"])             # BMU indices bit-exact
    np.testing.assert_allclose(est.quantization_error_, float(g["final_qe"]), rtol=1e-10)
    assert est.topographic_error_ == float(g["final_te"])
    sizes = [e[0] for e in log]
    assert sizes == [int(m) for m in g["trace_n_neurons"]]             # the recorded map size, epoch by epoch
    big = [e for e in log if e[0] >= HipBackend.FILTER_MIN_PROTOTYPES]
    filt = [e for e in big if e[1] == "filtered"]
    # (`auto`, the default: these young maps are still collapsed -- the candidate lists are 0.65 .. 0.9 of the map --
    #  so the policy looks at the filtered search every so often and otherwise runs all pairs: engine.hip
    #  bearable_mean; the forced modes run the filtered search in every epoch from 129 prototypes on)
    assert len(big) >= 60 and len(filt) >= (len(big) if mode != "default" else 5), (len(big), len(filt))
    if mode == "refine":
        assert any(e[-1] for e in filt), "the refinement never ran"
    if mode == "prune":
        assert all(e[3] == 0 for e in filt)                            # no sweep: candidates by the triangle inequality


def test_non_finite_input_is_refused_as_by_the_reference():
    """SomVQ.py:122 / SomClassifier: check_array / check_X_y refuse NaN and infinities with a ValueError.  With the
    default backend the pass over X that finds them is the device's (the column sums of the initialisation); the
    error is still sklearn's own, raised before any epoch runs -- also for +inf and -inf in one column (their sum is
    NaN) and for a float32 column whose finite values overflow their float32 sum (no error: the host check decides)."""
    from dbgsom_amd import SomClassifier, SomVQ

    X, _ = gi.blobs_f32(5000, 24, 2)
    y = (X[:, 0] > 0).astype(int)
    for bad in (np.nan, np.inf, -np.inf):
        Xb = X.copy()
        Xb[4321, 7] = bad
        with pytest.raises(ValueError, match="NaN|infinity|inf"):
            SomVQ(random_state=0, n_iter=5).fit(Xb)
        with pytest.raises(ValueError, match="NaN|infinity|inf"):
            SomClassifier(random_state=0, n_iter=5).fit(Xb.astype(np.float64), y)
    Xb = X.copy()
    Xb[10, 3], Xb[11, 3] = np.inf, -np.inf
    with pytest.raises(ValueError, match="NaN|infinity|inf"):
        SomVQ(random_state=0, n_iter=5).fit(Xb)
    Xh = X.copy()
    Xh[:, 5] = 3.0e38                       # finite, but the column's float32 sum is not
    est = SomVQ(random_state=0, n_iter=3).fit(Xh[:, :5])     # (the huge column itself would swamp every distance)
    assert est.n_iter_ == 2
    ok = SomVQ(random_state=0, n_iter=3).fit(X)
    assert np.isfinite(ok.weights_).all()


def test_known_answers_digits_gpu():
    from dbgsom_amd import SomVQ

    X, _ = gi.case_X("digits_f64")
    est = SomVQ(random_state=0).fit(X)
    assert est.n_iter_ == 112 and len(est.neurons_) == 25
    np.testing.assert_allclose(est.quantization_error_, 23.80011502226172, rtol=1e-12)
    assert est.topographic_error_ == 0.05008347245409015
    np.testing.assert_allclose(est.weights_.sum(), 7793.246057345110, rtol=1e-11)
    assert est.labels_[:10].tolist() == [23, 16, 6, 18, 19, 14, 17, 2, 21, 5]


def test_predict_accepts_integer_and_half_input_like_the_reference():
    """fit(X_int) converts to float64; predict / calculate_quantization_error on the same integer
    (or float16) array must too -- the reference's engine (sklearn NearestNeighbors) does."""
    from dbgsom_amd import SomClassifier, SomVQ
    from dbgsom_amd.backend import HipBackend

    Xi, y = gi.case_X("ties_int")
    Xi = np.rint(Xi).astype(np.int64)
    est = SomVQ(random_state=1, n_iter=12).fit(Xi)
    lab = est.predict(Xi)
    assert np.array_equal(lab, est.labels_) and np.array_equal(lab, est.predict(Xi.astype(np.float64)))
    assert np.array_equal(est.predict(Xi.astype(np.uint8)), lab)
    assert np.array_equal(est.predict(Xi.astype(np.float16)), lab)
    qe = est.calculate_quantization_error(Xi)
    assert qe == est.calculate_quantization_error(Xi.astype(np.float64))
    d, i = HipBackend().bmu(est.weights_, 1, X=Xi)       # the backend converts too
    assert np.array_equal(i, lab)
    Xd, yd = gi.case_X("digits_clf")
    clf = SomClassifier(random_state=0, n_iter=10).fit(Xd, yd)
    assert np.array_equal(clf.predict(Xd.astype(np.int32)), clf.predict(Xd))


def test_prototypes_stay_in_hbm_between_epochs():
    """during the epoch loop the prototype matrix crosses PCIe once on the way in
    (the four start vectors), comes back only at growth steps (the host extrapolates the inserted
    rows from it) and at the end; a growth step uploads the inserted rows only."""
    from dbgsom_amd import SomVQ

    X, _ = gi.case_X("blobs_dead")
    est = SomVQ(**gi.EST_KWARGS["blobs_dead"]).fit(X)
    g = gi.load("blobs_dead")
    np.testing.assert_allclose(est.weights_, g["final_weights"], rtol=1e-8, atol=1e-10)
    tr, growth = est._training_traffic, est._growth_epochs
    assert est.n_iter_ + 1 >= 30 and len(growth) >= 3
    assert tr["w_upload_calls"] == 1 and tr["w_upload_bytes"] == 4 * X.shape[1] * 8   # epoch 0: the 2 x 2 start map
    assert tr["w_download_calls"] == len(growth) + 2                  # growth steps + the two final snapshots
    assert tr["w_row_writes"] >= len(growth)                          # inserted (and overwritten) rows only
    assert tr["w_row_writes"] < 4 * len(est.neurons_)


def test_full_size_c5_shard_properties():
    """One GPU's shard of BASELINE config C5 at its real shape (N = 5e5, d = 2048, M = 4096,
    bfloat16-resident samples): the filtered search equals the all-pairs search on the full
    arrays, an oracle spot check on 1000 rows of the rounded samples, conservation sums."""
    import torch

    import bench
    from dbgsom_amd.backend import HipBackend
    from oracle import som_oracle as o

    n, d, rows, cols, seed, kind, _ = bench.WORKLOADS["c5"]
    M = rows * cols
    dev = torch.device("cuda", 0)
    X = bench.make_shard(torch, n, d, seed, dev, 0, kind).to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(seed + 7)
    W = X[torch.randperm(n, device=dev, generator=g)[:M]].double().cpu().numpy()
    gamma = float(1.0 / X.float().var(dim=0, unbiased=False).double().sum().item())
    hop, sigma = bench.lattice_hops(rows, cols), 0.2 * np.sqrt(M)
    out = {}
    for algo in ("filtered", "exact"):
        be = HipBackend(0, algorithm=algo).load_device(X)
        assert be._x_np_dtype == "bf16"
        out[algo] = be.epoch(W, hop, sigma, gamma, "compact", True)
        if algo == "filtered":
            assert be.filter_log[-1][0] == "filtered" and be.planes_cached
            sums = be.read_sums(M)
        be.release()
    rf, re_ = out["filtered"], out["exact"]
    assert np.array_equal(rf.winners, re_.winners) and np.array_equal(rf.distances, re_.distances)
    assert np.array_equal(rf.new_weights, re_.new_weights, equal_nan=True)
    assert np.array_equal(rf.errors, re_.errors) and rf.change_total == re_.change_total
    pick = np.random.default_rng(1).choice(n, 1000, replace=False)
    Xs = X[torch.from_numpy(pick).to(dev)].float().cpu().numpy()     # the exactly widened rows
    rd, ri = o.bmu_chain(Xs, W, 1)
    assert np.array_equal(rf.winners[pick], ri) and np.array_equal(rf.distances[pick], rd)
    # conservation: hits sum to N, E to sum(dist), K to sum(kw), S to kw^T X
    assert rf.activations.sum() == n
    np.testing.assert_allclose(rf.errors.sum(), rf.distances.sum(), rtol=1e-10)
    kw = 1 - np.sqrt(1 - np.exp(-gamma * rf.distances ** 2))
    S = sums[: M * d].reshape(M, d)
    np.testing.assert_allclose(sums[M * d: M * d + M].sum(), kw.sum(), rtol=1e-10)
    col = torch.zeros(d, dtype=torch.float64, device=dev)
    kwd = torch.from_numpy(kw).to(dev)
    for s in range(0, n, 50_000):
        col += kwd[s:s + 50_000] @ X[s:s + 50_000].double()
    np.testing.assert_allclose(S.sum(axis=0), col.cpu().numpy(), rtol=1e-9, atol=1e-9)


def test_vertical_growth_on_device_subsets_matches_the_recorded_tree():
    """vertical_growth=True on the MI355X: the Voronoi sets are gathered in HBM
    (dbgsom_ctx_partition / dbgsom_ctx_subset_create), the children are fitted on them without a
    host copy or a second upload; the tree equals the one recorded from the reference (its two
    slips in _grow_vertical corrected; as it stands it raises -- also recorded)."""
    from dbgsom_amd import SomVQ
    from dbgsom_amd import base as base_mod

    g = gi.load("vertical_blobs")
    assert "TypeError" in str(g["reference_raises"])
    X, _ = gi.case_X("vertical_blobs")
    seen = []
    orig = base_mod.BaseSom._load_resident

    def spy(self, data):
        seen.append(type(data).__name__)
        return orig(self, data)

    base_mod.BaseSom._load_resident = spy
    try:
        est = SomVQ(**gi.EST_KWARGS["vertical_blobs"]).fit(X)
    finally:
        base_mod.BaseSom._load_resident = orig
    gi.check_vertical_tree(est, g)
    assert seen[0] == "ndarray" and set(seen[1:]) == {"DeviceSamples"} and len(seen) == int(g["n_maps"])
    # float32 samples take the same route
    est32 = SomVQ(**gi.EST_KWARGS["vertical_blobs"]).fit(X.astype(np.float32))
    assert any("som" in est32.som_.nodes[n] for n in est32.neurons_)


def test_entropy_vertical_growth_on_device_subsets_matches_the_host_path():
    """SomClassifier(vertical_growth=True, growth_criterion="entropy"): a child map fitted on a device
    subset re-codes its labels (np.unique of the subset's classes) while the subset context inherited
    the PARENT's codes -- the child's own codes must be attached, or the class histograms drop every
    sample whose parent code is >= the child's class count (round-2 advisor finding).  The tree must
    equal the one the host path (X[mask], labels re-coded and uploaded) builds."""
    from dbgsom_amd import SomClassifier
    from oracle.som_oracle import OracleBackend

    rng = np.random.default_rng(11)
    centres = rng.normal(size=(6, 12)) * 1.5
    lab = rng.integers(0, 6, size=2400)
    X = centres[lab] + rng.normal(size=(2400, 12)) * 1.6   # overlapping classes: mixed neurons
    y = np.array(["a", "b", "c", "d", "e", "f"])[lab]
    # (on the host path this grows 7 child maps, five of them on class sets that skip a parent code)
    kw = dict(random_state=1, n_iter=16, max_neurons=6, vertical_growth=True, growth_criterion="entropy",
              spreading_factor=0.2, min_samples_vertical_growth=150)
    dev = SomClassifier(**kw).fit(X, y)
    host = SomClassifier(backend=OracleBackend(), **kw).fit(X, y)
    n_children = [0]

    def walk(a, b, path):
        assert a.neurons_ == b.neurons_, path
        assert list(a.classes_) == list(b.classes_), path
        np.testing.assert_allclose(a.weights_, b.weights_, rtol=1e-8, atol=1e-10)
        for i, node in enumerate(a.neurons_):
            ca, cb = a.som_.nodes[node].get("som"), b.som_.nodes[node].get("som")
            assert (ca is None) == (cb is None), path + [i]
            if ca is not None:
                n_children[0] += 1
                walk(ca, cb, path + [i])

    walk(dev, host, [])
    assert n_children[0] >= 1, "the case must grow vertically"
    # a child whose classes are a proper subset with a non-prefix code set is what used to break
    assert np.array_equal(dev.predict(X[:200]), host.predict(X[:200]))


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
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "c2", "--cpu-sample", "0",
           "--fine-phase", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    js = json.loads(line)
    assert js["n_gpus"] == 1 and js["value"] > 0 and js["roofline"]["frac"] > 0


def test_bench_via_ctx_under_torchrun_without_torch_in_the_workers():
    """`bench.py --via ctx --gpus 1` launched the way the driver launches the multi-GPU bench: the worker
    never imports torch, the per-epoch collective is RCCL issued by the library (dbgsom_ctx_set_rccl), the
    ncclUniqueId travels through a file (DBGSOM_FORCE_COLLECTIVE=1: also for one rank)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DBGSOM_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "2", "--workload", "c2", "--via", "ctx"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    js = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert js["n_gpus"] == 1 and js["value"] > 0 and js["torch_imported"] is False
    assert "RCCL inside the library" in js["config"]["via"]
    assert js["exact"]["prototypes_identical_to_headline"] is True


def test_bench_two_ranks_strong_scaling_rehearsal_on_one_gpu():
    """The N > 1 path of bench.py (strong scaling: the workload's rows split over the ranks, the
    per-epoch all-reduce, the weak-scaling rate of the same run, one JSON line from rank 0) with two
    ranks on the one GPU of a test box: gloo instead of RCCL (DBGSOM_BENCH_BACKEND), everything else
    as the driver launches it."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DBGSOM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "2", "--workload", "c2", "--fine-phase", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["scaling"] == "strong" and js["value"] > 0
    assert js["config"]["samples_total"] == 60_000 and js["config"]["samples_per_gpu"] == 30_000
    assert js["weak_scaling_same_run"]["samples_per_gpu"] == 60_000
    assert js["exact"]["prototypes_identical_to_headline"] is True
    assert "cpu_baseline" not in js and js["roofline"]["frac"] > 0


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
        if name == "c3":
            # the filtered search with the engine's policy on its own: a counting-only launch beside the
            # first epoch's sweep finds that the triangle inequality alone leaves the sample's cluster,
            # and the sweep is dropped (arm 0) -- and nothing changes in the results
            fi = HipBackend(0, algorithm="filtered")
            fi.load_device(X)
            hop = bench.lattice_hops(rows, cols)
            used, probes = [], []
            for e in range(4):
                rf = fi.epoch(W, hop, 0.2 * np.sqrt(M), gamma, "compact", True)
                assert np.array_equal(rf.winners, res.winners) and np.array_equal(rf.distances, res.distances)
                assert np.array_equal(rf.new_weights, res.new_weights)
                info = fi.epoch_info()
                used.append(int(info[2])); probes.append(info[6])
            # (an epoch with the full seed pre-pass may come in between: a look at arm 0 from better seeds)
            assert used[0] == 1 and used[-1] == 0 and used.count(0) >= 2, (used, probes)
            assert probes[0] < 0.1 * M and np.isnan(probes[-1]), probes
            fi.release()
        hip.release()


def test_whole_fit_with_the_pruning_search_equals_the_fit_with_the_all_pairs_search():
    """A growing map trained through `SomVQ.fit` with the search forced to the pruning form (no
    sweep; seeds = the previous epoch's winners, seed distances from that epoch's exact distances
    plus the prototypes' shift, growth steps in between) ends in the map the all-pairs search gives:
    same neurons, bit-identical prototypes and labels."""
    from dbgsom_amd import SomVQ
    from dbgsom_amd.backend import HipBackend

    X, _ = gi.blobs_f32(60_000, 64, 5, n_centers=300)
    kw = dict(random_state=0, max_neurons=400, n_iter=120, spreading_factor=0.9, coarse_training_frac=0.9,
              convergence_iter=2)
    ref = SomVQ(backend=HipBackend(algorithm="exact"), **kw).fit(X)
    be = HipBackend(algorithm="filtered_hint")   # (`auto` would back off to the all-pairs kernel on the young map)
    be.sweep_planes = 4
    est = SomVQ(backend=be, **kw).fit(X)
    assert len(ref.neurons_) > 300, len(ref.neurons_)          # the filter takes over above 128 prototypes
    assert est.neurons_ == ref.neurons_ and est.n_iter_ == ref.n_iter_
    assert np.array_equal(est.weights_, ref.weights_)
    assert np.array_equal(est.labels_, ref.labels_)
    assert est.quantization_error_ == ref.quantization_error_ and est.topographic_error_ == ref.topographic_error_
    pruned = [e for e in be.filter_log if e[0] == "filtered" and e[2] == 0]
    assert len(pruned) >= 40, be.filter_log[-10:]
