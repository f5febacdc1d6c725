"""CPU check of the mathematics behind the filtered BMU search (dbgsom_amd/csrc/filter.hip):
a NumPy emulation of the digit planes, the six int8 digit products and r~, against the exact
distances, on random and adversarial rows -- the error must stay inside the bound the kernel uses
(`filter_eps`), and the candidate rule must keep every prototype that can win or tie."""
import numpy as np
import pytest

F = 127.0 * 65536.0


def slice_rows(A):
    A = np.asarray(A, dtype=np.float64)
    s = np.abs(A).max(axis=1)
    s[s == 0] = 1.0
    Q = np.rint(A / s[:, None] * F).astype(np.int64)
    d2 = ((Q + 128) & 255) - 128
    q1 = (Q - d2) >> 8
    d1 = ((q1 + 128) & 255) - 128
    d0 = (q1 - d1) >> 8
    assert np.abs(d0).max() <= 127 and np.abs(d1).max() <= 128 and np.abs(d2).max() <= 128
    assert np.array_equal(d0 * 65536 + d1 * 256 + d2, Q)
    return (d0, d1, d2), s, np.abs(A).sum(axis=1)


def r_tilde(X, W, levels=3):
    (x0, x1, x2), sx, l1x = slice_rows(X)
    (w0, w1, w2), tw, l1w = slice_rows(W)
    P0 = x0 @ w0.T
    P1 = x0 @ w1.T + x1 @ w0.T
    P2 = x0 @ w2.T + x1 @ w1.T + x2 @ w0.T
    # kept digit products: levels == 3 all with a + b <= 2 (six), 2: a + b <= 1 (three), 1: (0, 0) only
    T = (P0 * 256 + (P1 if levels >= 2 else 0)) * 256 + (P2 if levels == 3 else 0)
    xx = (X.astype(np.float64) ** 2).sum(axis=1)
    yy = (W.astype(np.float64) ** 2).sum(axis=1)
    ctab = 2.0 * tw * 65536.0 / (F * F)
    return (xx[:, None] + yy[None, :]) - sx[:, None] * (ctab[None, :] * T.astype(np.float64)), \
        (sx, l1x, xx, tw, l1w, yy)


def filter_eps(s, l1x, xx, l1w_max, t_max, yy_max, d, planes=3):
    quant = (s * l1w_max + t_max * l1x) / (2.0 * F) + d * s * t_max / (4.0 * F * F)
    # dropped digit products per feature in units of 128 * 128 (filter.hip `filter_eps`): levels
    # 3 and 4 always; level 2 (three products) below three planes; level 1 (two) with one plane
    per_k = 16384.0 * (513.0 if planes >= 3 else 3.0 * 65536.0 + 513.0 if planes == 2
                       else 2.0 * 16777216.0 + 3.0 * 65536.0 + 513.0)
    dropped = d * per_k * s * t_max / (F * F)
    rounding = 4.0 * (d + 16) * 1.1102230246251565e-16 * (xx + yy_max)
    return 2.0 * (quant + dropped) * (1.0 + 1e-7) + rounding


def exact_r(X, W):
    Xl, Wl = X.astype(np.longdouble), W.astype(np.longdouble)
    return ((Xl ** 2).sum(1)[:, None] - 2 * (Xl @ Wl.T) + (Wl ** 2).sum(1)[None, :]).astype(np.float64)


CASES = [
    ("gauss", lambda rng: (rng.normal(size=(300, 96)).astype(np.float32), rng.normal(size=(80, 96)))),
    ("blobs", lambda rng: (((rng.normal(size=(4, 64)) * 4)[rng.integers(0, 4, 400)]
                            + rng.normal(size=(400, 64))).astype(np.float32),
                           rng.normal(size=(120, 64)) * 4)),
    ("heavy_tail", lambda rng: (rng.standard_cauchy(size=(200, 48)).astype(np.float32),
                                rng.standard_cauchy(size=(60, 48)))),
    ("tiny_and_huge_rows", lambda rng: (np.concatenate([rng.normal(size=(50, 32)) * 1e-6,
                                                         rng.normal(size=(50, 32)) * 1e6]).astype(np.float32),
                                        np.concatenate([rng.normal(size=(20, 32)) * 1e-6,
                                                        rng.normal(size=(20, 32)) * 1e6]))),
    # adversarial for the coarse sweeps: a tiny spread around a large mean (every top digit equal)
    ("tiny_spread_large_mean", lambda rng: (rng.uniform(0.45, 0.55, size=(300, 128)).astype(np.float32),
                                            rng.uniform(0.45, 0.55, size=(90, 128)))),
    ("offset_blobs", lambda rng: ((100.0 + (rng.normal(size=(5, 80)))[rng.integers(0, 5, 250)]
                                   + 0.01 * rng.normal(size=(250, 80))).astype(np.float32),
                                  100.0 + rng.normal(size=(70, 80)))),
]


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_r_tilde_stays_inside_the_bound(name):
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 32)
    X, W = dict((c[0], c[1]) for c in CASES)[name](rng)
    X = np.asarray(X, dtype=np.float32).astype(np.float64)
    W = np.asarray(W, dtype=np.float64)
    r = exact_r(X, W)
    for planes in (3, 2, 1):     # 1 = the one-product sweep the shipped default starts with
        rt, (sx, l1x, xx, tw, l1w, yy) = r_tilde(X, W, levels=planes)
        eps = filter_eps(sx, l1x, xx, l1w.max(), tw.max(), yy.max(), X.shape[1], planes)
        err = np.abs(rt - r)
        assert (err <= eps[:, None]).all(), (planes, float((err / eps[:, None]).max()))
        # the bound is not absurdly loose either (within ~3-4 orders of magnitude of the worst error)
        assert (err / eps[:, None]).max() > 1e-5
        # and the candidate rule built on it keeps every prototype that can win or tie, any seed
        n = X.shape[0]
        for seeds in (r.argmin(1), rng.integers(0, W.shape[0], n)):
            thr = rt[np.arange(n), seeds] + 2 * eps
            must = r <= r[np.arange(n), seeds][:, None]
            assert ((rt <= thr[:, None]) | ~must).all(), planes


def test_spiky_rows_and_zero_rows():
    rng = np.random.default_rng(3)
    X = ((rng.random((150, 80)) < 0.05) * rng.normal(size=(150, 80)) * 100).astype(np.float32).astype(np.float64)
    X[7] = 0.0
    W = (rng.random((40, 80)) < 0.1) * rng.normal(size=(40, 80)) * 50
    W[3] = 0.0
    rt, (sx, l1x, xx, tw, l1w, yy) = r_tilde(X, W)
    eps = filter_eps(sx, l1x, xx, l1w.max(), tw.max(), yy.max(), 80)
    assert (np.abs(rt - exact_r(X, W)) <= eps[:, None]).all()


def test_candidate_rule_keeps_every_possible_winner():
    """For ANY seed: {j : r~_ij <= r~_i,seed + 2 eps_i} contains every j with r_ij <= r_i,seed,
    hence the exact winner and all its ties."""
    rng = np.random.default_rng(11)
    X = rng.normal(size=(500, 40)).astype(np.float32).astype(np.float64)
    W = rng.normal(size=(200, 40))
    W[50:60] = W[40:50]                      # exact duplicates
    W[60:70] = W[40:50] * (1 + 1e-12)        # near duplicates
    rt, (sx, l1x, xx, tw, l1w, yy) = r_tilde(X, W)
    r = exact_r(X, W)
    eps = filter_eps(sx, l1x, xx, l1w.max(), tw.max(), yy.max(), 40)
    for seeds in (r.argmin(1), rng.integers(0, 200, 500), np.zeros(500, int)):
        thr = rt[np.arange(500), seeds] + 2 * eps
        cand = rt <= thr[:, None]
        must = r <= r[np.arange(500), seeds][:, None]
        assert (cand | ~must).all()
        assert cand[np.arange(500), r.argmin(1)].all()


def test_coarse_prepass_is_a_valid_seed_source():
    rng = np.random.default_rng(5)
    X = rng.normal(size=(200, 64)).astype(np.float32).astype(np.float64)
    W = rng.normal(size=(300, 64))
    rt2, _ = r_tilde(X, W, levels=2)
    seeds = rt2[:, ::4].argmin(1) * 4       # every 4th prototype, two digit levels
    assert seeds.max() < 300
    r = exact_r(X, W)
    # coarse seeds are near-minimal: within a few coarse error units of the true minimum
    gap = r[np.arange(200), seeds] - r.min(1)
    assert np.median(gap) < np.median(np.sort(r, 1)[:, 8] - r.min(1))


# ---- candidates without a sweep: the triangle inequality (filter.hip section 2c) -------------------
PLANE0_ERR = 32897.0 / (127.0 * 65536.0) * (1.0 + 1e-6)


def prune_survivors(X, W, seeds):
    """NumPy statement of proto_gap_kernel + prune_mark_kernel (top digit plane only): survivors[i, j]
    is False only where the rule PROVES r(i, j) > r(i, seed_i)."""
    d = X.shape[1]
    (x0, _, _), sx, _ = slice_rows(X)
    (w0, _, _), tw, _ = slice_rows(W)
    root_d = np.sqrt(float(d)) * (1 + 1e-12)

    def hat_dist2(a0, sa, b0, sb, sign):
        """|a^ - b^|^2 of rows a^ = sa a0 / 127, nudged down (sign -1) or up (+1) by the rounding margin"""
        A = (a0 * a0).sum(1).astype(np.float64)
        B = (b0 * b0).sum(1).astype(np.float64)
        P = (a0 @ b0.T).astype(np.float64)
        sq = (sa * sa * A)[:, None] + (sb * sb * B)[None, :]
        cr = 2.0 * sa[:, None] * sb[None, :] * P
        return ((sq - cr) + sign * 1e-12 * (sq + np.abs(cr))) / 16129.0

    dh2 = hat_dist2(w0, tw, w0, tw, -1)
    lo = np.where(dh2 > 0, np.sqrt(np.maximum(dh2, 0)) * (1 - 1e-12), 0.0) \
        - root_d * (tw[:, None] + tw[None, :]) * PLANE0_ERR
    gap = np.where(lo > 0, lo * lo * (1 - 1e-6), 0.0)
    g32 = gap.astype(np.float32)
    g32 = np.where(g32.astype(np.float64) > gap, np.nextafter(g32, np.float32(0)), g32)   # rounded towards zero
    gap = g32.astype(np.float64)
    n = X.shape[0]
    xx = (X ** 2).sum(1)
    yy = (W ** 2).sum(1)
    rho = 4.0 * (d + 16) * 1.1102230246251565e-16 * (xx + yy.max())
    dxw = hat_dist2(x0, sx, w0, tw, +1)[np.arange(n), seeds]
    up = np.where(dxw > 0, np.sqrt(np.maximum(dxw, 0)) * (1 + 1e-12), 0.0) + root_d * (sx + tw[seeds]) * PLANE0_ERR
    b = 2.0 * up * (1 + 1e-12) + (np.sqrt(2.0 * rho) * 1.0001 + 1e-300)
    bound = b * b * (1 + 1e-12)
    keep = ~(gap[seeds] >= bound[:, None])
    keep[np.arange(n), seeds] = True
    return keep


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_triangle_pruning_never_removes_a_possible_winner(name):
    rng = np.random.default_rng(abs(hash("prune" + name)) % 2 ** 32)
    X, W = dict((c[0], c[1]) for c in CASES)[name](rng)
    X = np.asarray(X, dtype=np.float32).astype(np.float64)
    W = np.asarray(W, dtype=np.float64)
    W[1] = W[0]                                   # a duplicate and a near duplicate among the prototypes
    W[2] = W[0] * (1 + 1e-13)
    r = exact_r(X, W)
    n = X.shape[0]
    for seeds in (r.argmin(1), rng.integers(0, W.shape[0], n), np.zeros(n, int)):
        keep = prune_survivors(X, W, seeds)
        removed_can_win = ~keep & (r <= r[np.arange(n), seeds][:, None])
        assert not removed_can_win.any(), name
        assert keep[np.arange(n), r.argmin(1)].all()


def test_triangle_pruning_leaves_the_own_cluster_on_blobs():
    rng = np.random.default_rng(21)
    d, k = 96, 12
    centres = rng.normal(size=(k, d)) * 4
    lab = rng.integers(0, k, 600)
    X = (centres[lab] + rng.normal(size=(600, d))).astype(np.float32).astype(np.float64)
    wl = rng.integers(0, k, 240)
    W = centres[wl] + rng.normal(size=(240, d))
    r = exact_r(X, W)
    keep = prune_survivors(X, W, r.argmin(1))
    same = lab[:, None] == wl[None, :]
    assert not (keep & ~same).any()               # everything outside the sample's blob is ruled out
    assert keep.sum(1).mean() < 1.5 * same.sum(1).mean()


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_top_plane_stands_for_the_row_within_its_bound(name):
    """|a_k - s D0_k / 127| <= s PLANE0_ERR: what the Euclidean error e_a = sqrt(d) s PLANE0_ERR rests on."""
    rng = np.random.default_rng(abs(hash("res" + name)) % 2 ** 32)
    X, W = dict((c[0], c[1]) for c in CASES)[name](rng)
    for A in (np.asarray(X, dtype=np.float64), np.asarray(W, dtype=np.float64)):
        (d0, _, _), s, _ = slice_rows(A)
        res = np.abs(A - s[:, None] * d0 / 127.0)
        assert (res <= s[:, None] * PLANE0_ERR).all()
        assert np.sqrt((res ** 2).sum(1)).max() <= (np.sqrt(A.shape[1]) * s * PLANE0_ERR).max()


# ---- the refinement's bound (csrc/refine.h): top two digit planes = a 16-bit rounding of the row ------
F16 = 127.0 * 256.0


def plane16(A):
    (d0, d1, _), s, _ = slice_rows(A)
    q16 = d0 * 256 + d1
    A16 = s[:, None] * q16 / F16
    res = np.sqrt(((np.asarray(A, dtype=np.float64) - A16) ** 2).sum(axis=1))
    return q16, s, res


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_refinement_bound_holds_and_keeps_every_possible_winner(name):
    """v_ij = |w_j|^2 - 2 s_i t_j T / F16^2 with T = sum Q16x Q16w (exact integers);
    |v_ij + |x_i|^2 - r_ij| <= eps_i = 2 [rx_i (max|w| + max rw) + |x_i| max rw] + rounding, and the rule
    v_ij <= min_j v_ij + 2 eps_i keeps the arg-min and everything tied with it."""
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 32 + 1)
    X, W = dict((c[0], c[1]) for c in CASES)[name](rng)
    X = np.asarray(X, dtype=np.float32).astype(np.float64)
    W = np.asarray(W, dtype=np.float64)
    W[1] = W[0]                                     # an exact tie
    qx, sx, rx = plane16(X)
    qw, tw, rw = plane16(W)
    T = (qx.astype(object) @ qw.T.astype(object)).astype(np.float64)   # exact (python integers)
    xx, yy = (X ** 2).sum(1), (W ** 2).sum(1)
    v = yy[None, :] - 2.0 * sx[:, None] * tw[None, :] * T / (F16 * F16)
    r = exact_r(X, W)
    d = X.shape[1]
    wn, rwm = np.sqrt(yy.max()), rw.max()
    eps = 2.0 * (rx * (wn + rwm) + np.sqrt(xx) * rwm) * (1 + 1e-9) + 4.0 * (d + 16) * 1.1102230246251565e-16 * (xx + yy.max())
    err = np.abs(v + xx[:, None] - r)
    assert (err <= eps[:, None]).all(), float((err / eps[:, None]).max())
    keep = v <= v.min(axis=1)[:, None] + 2 * eps[:, None]
    best = r.min(axis=1)
    assert (keep | (r > best[:, None])).all()       # every arg-min (ties included) is kept
    # the filter is sharp where prototypes are apart: mean candidates per sample stays small on blobs
    if name == "blobs":
        assert keep.sum(axis=1).mean() < 3.0
