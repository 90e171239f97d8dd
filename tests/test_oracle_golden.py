"""Pins the CPU oracle against the reference's own golden vectors and property tests.

Sources (paths relative to /root/reference, Bijectors.jl v0.16.2; table in SURVEY.md §8c).
The reference is pure Julia and cannot run in this image; these are the values its docstrings
and tests hold for the hot path, plus its generic `test_bijector` properties
(test/bijectors/utils.jl:7-91) re-expressed with finite-difference Jacobians.
"""
import itertools
import math

import numpy as np
import pytest

E = math.e


def _jac_logabsdet(f, x, h=1e-6):
    """log|det J| of f at x by central differences (square Jacobian), float64."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    J = np.zeros((f(x).size, n))
    for i in range(n):
        d = np.zeros(n)
        d[i] = h
        J[:, i] = (f(x + d) - f(x - d)) / (2 * h)
    return np.linalg.slogdet(J)[1]


# ------------------------------------------------------------------ golden values
def test_elementwise_exp_docstring(orc):
    # src/interface.jl:29-30
    y, l = orc.chain([(orc.OP_EXP, None, None)], np.array([1.0, 2.0, 3.0]))
    assert y.tolist() == [2.718281828459045, 7.38905609893065, 20.085536923187668]
    assert l == 6.0


def test_elementwise_log_lognormal(orc):
    # README.md:17-28 ; src/Bijectors.jl:244-246
    y, l = orc.chain([(orc.OP_LOG, None, None)], np.array([0.6471106974390148]))
    assert y[0] == pytest.approx(-0.43523790570180304, abs=1e-15)
    assert l == pytest.approx(0.43523790570180304, abs=1e-15)
    _, l = orc.chain([(orc.OP_LOG, None, None)], np.array([E]))
    assert l == pytest.approx(-1.0, abs=1e-15)


def test_named_stacked_log_values(orc):
    # src/bijectors/named_stacked.jl:33-36 (log on a=1.0, b=2.0)
    y, l = orc.chain([(orc.OP_LOG, None, None)], np.array([1.0, 2.0]))
    assert y.tolist() == [0.0, 0.6931471805599453]
    assert l == -0.6931471805599453


def test_truncate_untruncate_docstrings(orc):
    # src/vector/interface.jl:98-101: Truncate(0,1) (inverse link) at 1.0
    x, l = orc.chain([(orc.OP_TRUNCATED_INV, 0.0, 1.0)], np.array([1.0]))
    assert x[0] == pytest.approx(0.7310585786300049, abs=1e-15)
    assert l == pytest.approx(-1.6265233750364456, abs=1e-15)
    # src/vector/interface.jl:129: Untruncate(0,1) at 0.5
    y, l = orc.chain([(orc.OP_TRUNCATED, 0.0, 1.0)], np.array([0.5]))
    assert y[0] == 0.0
    assert l == pytest.approx(1.3862943611198906, abs=1e-15)


def test_simplex_invlink_saturation(orc):
    # test/legacy_interface.jl:285 (needs the guarded logistic)
    x, _ = orc.simplex(np.array([-1000.0, -1000.0]), inverse=True)
    np.testing.assert_allclose(x, [0.0, 0.0, 1.0], atol=1e-9)


def test_simplex_logpdf_with_trans_value(orc):
    # test/legacy_interface.jl:289: logpdf_with_trans(Dirichlet(1,1,1), invlink(d,[-1,-2]), true)
    x, l = orc.simplex(np.array([-1.0, -2.0]), inverse=True)
    np.testing.assert_allclose(x, [0.15536240349696342, 0.1006832695529001, 0.7439543269501365], atol=1e-15)
    _, lf = orc.simplex(x)
    assert lf[0] == pytest.approx(4.45354607314081, abs=1e-12)
    assert l[0] == pytest.approx(-4.45354607314081, abs=1e-12)
    # Dirichlet(1,1,1) logpdf = log Γ(3) = log 2 ; with_trans subtracts the forward ladj
    assert math.log(2.0) - lf[0] == pytest.approx(-3.760398892580863, abs=1e-9)


def test_simplex_matrix_columns_sum_to_one(orc):
    # test/legacy_interface.jl:275-279
    x = np.array([[-2.72689, -2.92751, 1.63114, -1.62054, 0.0], [-1.24249, 2.58902, -3.73043, -3.53685, 0.0]]).T
    X, _ = orc.simplex(np.asfortranarray(x), inverse=True)
    assert X.shape == (6, 2)
    assert np.all(X.sum(axis=0) == 1.0)


def test_veccorr_docstring_pins_link_chol(orc):
    # src/bijectors/corr.jl:113-122 (input printed with 6 digits -> 6e-7)
    X = np.array([[1.0, -0.705273, -0.348638], [-0.705273, 1.0, 0.0534538], [-0.348638, 0.0534538, 1.0]])
    U = np.linalg.cholesky(X).T
    y, _ = orc.vec_cholesky(U, inverse=False, uplo="U")
    np.testing.assert_allclose(y, [-0.8777149781928181, -0.3638927608636788, -0.29813769428942216], atol=2e-6)
    # 'L' mode goes through transpose_eager (corr.jl:337)
    yl, _ = orc.vec_cholesky(np.ascontiguousarray(U.T), inverse=False, uplo="L")
    np.testing.assert_array_equal(y, yl)


def test_vec_cholesky_consistency_and_roundtrip(orc):
    # corr.jl:370-399 vs :485-501 ; roundtrip test/bijectors/corr.jl:46-64
    rng = np.random.default_rng(3)
    for K in (2, 3, 5, 8):
        y = rng.normal(size=K * (K - 1) // 2) * 0.7
        W, lj = orc.vec_cholesky(y, inverse=True)
        _, lj2 = orc.vec_cholesky(y, inverse=True, transform=False)
        assert lj[0] == pytest.approx(lj2[0], rel=1e-13)
        np.testing.assert_allclose(np.linalg.norm(W, axis=0), 1.0, atol=1e-14)
        assert np.all(np.tril(W, -1) == 0)
        y2, lf = orc.vec_cholesky(W, inverse=False)
        np.testing.assert_allclose(y2, y, atol=1e-12)
        assert lf[0] == pytest.approx(-lj[0], rel=1e-10)
        WL, _ = orc.vec_cholesky(y, inverse=True, uplo="L")
        np.testing.assert_array_equal(WL, W.T)


def test_vec_cholesky_logjac_vs_numeric_jacobian(orc):
    # free parameters of a K=4 factor: strict upper triangle of W (unit-norm columns)
    K = 4
    rng = np.random.default_rng(5)
    y = rng.normal(size=6) * 0.5
    iu = np.triu_indices(K, 1)

    def f(v):
        W, _ = orc.vec_cholesky(v, inverse=True)
        return W[iu]

    _, lj = orc.vec_cholesky(y, inverse=True)
    assert _jac_logabsdet(f, y) == pytest.approx(lj[0], abs=1e-7)


def test_find_alpha_sweep_and_value(orc):
    # test/normalising_flows.jl:47-70
    for wt_y, wu, b in itertools.product(
        (-20.3, -3, -1.5, 0.0, 5, 7.25, 12.3), (-1, -0.5, -1e-20, 0, 1e-20, 3, 11 / 3, 17.2), (-19.3, -8 / 3, -1, 0.0, 0.5, 3, 4.3)
    ):
        a = orc.find_alpha(wt_y, wu, b)
        res = a + wu * math.tanh(a + b)
        if wt_y == 0.0:
            assert abs(res - wt_y) <= 1e-14
        else:
            assert res == pytest.approx(wt_y, rel=1.5e-8)
    a = orc.find_alpha(0.8845640339582252, 0.8296950433716855, -1e8)
    assert a == pytest.approx(0.8845640339582252 + 0.8296950433716855, rel=1.5e-8)


def test_coupling_golden(orc):
    # src/bijectors/coupling.jl:147-172 ; test/bijectors/coupling.jl:18-56  (mask(3,[1],[2]))
    x = np.array([1.0, 2.0, 3.0])
    y, l = orc.coupling_affine([0], None, np.array([[x[1]]]), x)  # θ = x2 -> Shift(x2[1])
    assert y.tolist() == [3.0, 2.0, 3.0] and l[0] == 0.0
    xb, lb = orc.coupling_affine([0], None, np.array([[x[1]]]), y, inverse=True)
    assert xb.tolist() == x.tolist() and lb[0] == 0.0
    for x, yy in (([-1.0, -2.0, -3.0], [2.0, -2.0, -3.0]), ([1.0, 2.0, 3.0], [2.0, 2.0, 3.0])):
        x = np.array(x)
        y, l = orc.coupling_affine([0], np.array([[x[1]]]), None, x)  # θ -> Scale(x2[1])
        assert y.tolist() == yy
        assert l[0] == pytest.approx(math.log(2.0), abs=1e-15)


def test_stacked_values_via_chains(orc):
    # test/bijectors/stacked.jl:100-108: Stacked(exp, log, Shift(5)) on ones(3) -> [e, 0, 6]
    assert orc.chain([(orc.OP_EXP, None, None)], np.ones(1))[0][0] == E
    assert orc.chain([(orc.OP_LOG, None, None)], np.ones(1))[0][0] == 0.0
    assert orc.chain([(orc.OP_SHIFT, 5.0, None)], np.ones(1))[0][0] == 6.0


def test_permute_exact(orc):
    # test/bijectors/permute.jl:23-64: Permute([2,1,3]) maps [1,2,3] -> [2,1,3]; exact round trip
    src = [1, 0, 2]
    x = np.array([1.0, 2.0, 3.0])
    y = orc.permute(src, x)
    assert y.tolist() == [2.0, 1.0, 3.0]
    assert orc.permute(np.argsort(src), y).tolist() == x.tolist()


def test_rqs_identity_outside_and_knot_ends(orc):
    # test/bijectors/rational_quadratic_spline.jl:17-35,47-61
    rng = np.random.default_rng(0)
    d, K, B = 2, 3, 2.0
    w, h, dv = orc.rqs_params(rng.normal(size=(d, K)), rng.normal(size=(d, K)), rng.normal(size=(d, K - 1)), B)
    np.testing.assert_allclose(w[:, 0], -B)
    np.testing.assert_allclose(w[:, -1], B, atol=1e-15)
    assert np.all(dv[:, 0] == 1.0) and np.all(dv[:, -1] == 1.0)
    for xv in (5.0, -5.0):
        x = np.full(d, xv)
        y, l = orc.rqs(w, h, dv, x)
        assert y.tolist() == x.tolist() and l[0] == 0.0
        xi, li = orc.rqs(w, h, dv, x, inverse=True)
        assert xi.tolist() == x.tolist() and li[0] == 0.0


def test_logexpfunctions_guards(orc):
    assert orc.logistic(-1000.0) == 0.0 and orc.logistic(40.0) == 1.0
    assert orc.logistic(-110.0, np.float32) == 0.0 and orc.logistic(17.0, np.float32) == 1.0
    for x in (-50.0, -5.0, 0.3, 10.0, 20.0, 40.0):
        assert orc.log1pexp(x) == pytest.approx(np.logaddexp(0.0, x), rel=1e-15)
        assert orc.logcosh(x) == pytest.approx(abs(x) + math.log1p(math.exp(-2 * abs(x))) - math.log(2), rel=1e-14, abs=1e-16)


# ------------------------------------------------------------------ test_bijector properties
def _check_props(fwd, inv, x, dense_jac=True, tol=1e-7):
    """test/bijectors/utils.jl:43-62: roundtrip, ladj vs Jacobian, ladj(inv) == -ladj(fwd)."""
    y, l = fwd(x)
    xb, lb = inv(y)
    np.testing.assert_allclose(xb, x, rtol=1e-9, atol=1e-10)
    assert np.sum(lb) == pytest.approx(-np.sum(l), rel=1e-9, abs=1e-10)
    if dense_jac:
        assert _jac_logabsdet(lambda v: fwd(v)[0], x) == pytest.approx(float(np.sum(l)), abs=tol)


@pytest.mark.parametrize(
    "ops_f,ops_i,gen",
    [
        ("exp", "log", lambda r: r.normal(size=4)),
        ("logit", "logit_inv", lambda r: r.uniform(-0.9, 1.9, size=4)),
        ("leaky", "leaky_inv", lambda r: r.normal(size=5)),
        ("trunc_both", "trunc_both_inv", lambda r: r.uniform(0.1, 1.9, size=4)),
        ("trunc_lo", "trunc_lo_inv", lambda r: r.uniform(0.1, 4.0, size=4)),
        ("trunc_up", "trunc_up_inv", lambda r: r.uniform(-4, 1.9, size=4)),
        ("affexp", "affexp_inv", lambda r: r.normal(size=4)),
    ],
)
def test_elementwise_properties(orc, ops_f, ops_i, gen):
    o = orc
    a_vec = np.array([0.5, 1.5, -2.0, 0.7])
    table = {
        "exp": [(o.OP_EXP, None, None)], "log": [(o.OP_LOG, None, None)],
        "logit": [(o.OP_LOGIT, -1.0, 2.0)], "logit_inv": [(o.OP_LOGIT_INV, -1.0, 2.0)],
        "leaky": [(o.OP_LEAKY_RELU, 0.1, None)], "leaky_inv": [(o.OP_LEAKY_RELU, 1 / 0.1, None)],
        "trunc_both": [(o.OP_TRUNCATED, 0.0, 2.0)], "trunc_both_inv": [(o.OP_TRUNCATED_INV, 0.0, 2.0)],
        "trunc_lo": [(o.OP_TRUNCATED, 0.0, np.inf)], "trunc_lo_inv": [(o.OP_TRUNCATED_INV, 0.0, np.inf)],
        "trunc_up": [(o.OP_TRUNCATED, -np.inf, 2.0)], "trunc_up_inv": [(o.OP_TRUNCATED_INV, -np.inf, 2.0)],
        # exp ∘ Shift(b) ∘ Scale(a): application order Scale, Shift, Exp
        "affexp": [(o.OP_SCALE, a_vec, None), (o.OP_SHIFT, 0.1, None), (o.OP_EXP, None, None)],
        "affexp_inv": [(o.OP_LOG, None, None), (o.OP_SHIFT, -0.1, None), (o.OP_SCALE_INV, a_vec, None)],
    }
    x = gen(np.random.default_rng(1))
    _check_props(lambda v: o.chain(table[ops_f], v), lambda v: o.chain(table[ops_i], v), x)


def test_scale_ladj_quirk(orc):
    # scale.jl:26-32: scalar a -> log|a| * length(x); vector a -> sum(log|a_i|) even for a matrix
    x = np.asfortranarray(np.random.default_rng(0).normal(size=(3, 5)))
    _, l = orc.chain([(orc.OP_SCALE, -2.0, None)], x)
    assert l == pytest.approx(math.log(2.0) * 15)
    a = np.array([0.5, -3.0, 2.0])
    _, l = orc.chain([(orc.OP_SCALE, a, None)], x)
    assert l == pytest.approx(np.sum(np.log(np.abs(a))))
    _, l = orc.chain([(orc.OP_SHIFT, a, None)], x)
    assert l == 0.0


def test_leaky_relu_values(orc):
    # test/bijectors/leaky_relu.jl:6-50
    for dt in (np.float32, np.float64):
        y, l = orc.chain([(orc.OP_LEAKY_RELU, 0.1, None)], np.array([-1.0, 1.0], dtype=dt))
        assert y.dtype == dt
        np.testing.assert_allclose(y, [-0.1, 1.0], rtol=1e-7)
        assert l == pytest.approx(math.log(0.1), rel=1e-6)


def test_ordered_properties(orc):
    # test/bijectors/ordered.jl:25-38
    rng = np.random.default_rng(2)
    y = np.asfortranarray(rng.normal(size=(5, 4)))
    x, l = orc.ordered(y)
    assert np.all(np.diff(x, axis=0) > 0)
    np.testing.assert_allclose(l, y[1:].sum(axis=0), rtol=1e-14)
    yb, lb = orc.ordered(x, inverse=True)
    np.testing.assert_allclose(yb, y, atol=1e-12)
    np.testing.assert_allclose(lb, -l, atol=1e-12)
    v = y[:, 0].copy()
    assert _jac_logabsdet(lambda t: orc.ordered(t)[0], v) == pytest.approx(float(orc.ordered(v)[1]), abs=1e-7)
    one = np.array([0.3])
    assert orc.ordered(one)[0].tolist() == [0.3]


def test_simplex_properties(orc):
    # test/bijectors/simplex.jl:1-11, legacy_interface.jl:300-319 (Jacobian consistency)
    rng = np.random.default_rng(4)
    K = 5
    x = rng.dirichlet(np.ones(K))
    y, l = orc.simplex(x)
    xb, lb = orc.simplex(y, inverse=True)
    np.testing.assert_allclose(xb, x, atol=1e-12)
    assert lb[0] == pytest.approx(-l[0], rel=1e-10)
    # square Jacobian on the first K-1 coordinates (x_K = 1 - sum)
    def f(v):
        full = np.append(v, 1.0 - v.sum())
        return orc.simplex(full)[0]
    assert _jac_logabsdet(f, x[:-1], h=1e-7) == pytest.approx(l[0], abs=1e-5)


def test_flow_layers_logdet_vs_jacobian(orc):
    # test/normalising_flows.jl:7-35,74-84 (2 x 20 batches, per-column Jacobians)
    rng = np.random.default_rng(6)
    d = 2
    Z = np.asfortranarray(rng.normal(size=(d, 20)))
    w, u, b = rng.normal(size=d), rng.normal(size=d), rng.normal()
    out, l = orc.planar(w, u, [b], Z)
    for n in range(20):
        z = Z[:, n].copy()
        assert _jac_logabsdet(lambda t: orc.planar(w, u, [b], t)[0], z) == pytest.approx(l[n], abs=1e-7)
    a_, be, z0 = rng.normal(), rng.normal(), rng.normal(size=d)
    out, l = orc.radial(a_, be, z0, Z)
    for n in range(20):
        z = Z[:, n].copy()
        assert _jac_logabsdet(lambda t: orc.radial(a_, be, z0, t)[0], z) == pytest.approx(l[n], abs=1e-7)
    bb, logs, m, v = rng.normal(size=d), rng.normal(size=d) * 0.3, rng.normal(size=d), rng.uniform(0.5, 2, size=d)
    out, l = orc.batchnorm(bb, logs, m, v, 1e-5, Z)
    for n in range(3):
        z = Z[:, n].copy()
        assert _jac_logabsdet(lambda t: orc.batchnorm(bb, logs, m, v, 1e-5, t)[0], z) == pytest.approx(l[n], abs=1e-7)
    xb, lb = orc.batchnorm(bb, logs, m, v, 1e-5, out, inverse=True)
    np.testing.assert_allclose(xb, Z, atol=1e-12)
    np.testing.assert_allclose(lb, -l, atol=1e-13)


def test_flow_layers_inverse_roundtrip(orc):
    # test/normalising_flows.jl:37-42,86-91 (10 x 100)
    rng = np.random.default_rng(7)
    d = 10
    Z = np.asfortranarray(np.ones((d, 100)) + 0.1 * rng.normal(size=(d, 100)))
    w, u, b = rng.normal(size=(d, 3)), rng.normal(size=(d, 3)), rng.normal(size=3)
    Y, l = orc.planar(w, u, b, Z)
    Zb, lb = orc.planar(w, u, b, Y, inverse=True)
    np.testing.assert_allclose(Zb, Z, atol=1e-9)
    np.testing.assert_allclose(lb, -l, atol=1e-9)
    a_, be, z0 = rng.normal(), rng.normal(), rng.normal(size=d)
    Y, l = orc.radial(a_, be, z0, Z)
    Zb, lb = orc.radial(a_, be, z0, Y, inverse=True)
    np.testing.assert_allclose(Zb, Z, atol=1e-9)
    np.testing.assert_allclose(lb, -l, atol=1e-9)


def test_rqs_properties(orc):
    # test/bijectors/rational_quadratic_spline.jl:17-104
    rng = np.random.default_rng(8)
    d, K, B = 3, 8, 3.0
    w, h, dv = orc.rqs_params(rng.normal(size=(d, K)), rng.normal(size=(d, K)), rng.normal(size=(d, K - 1)), B)
    assert np.all(np.diff(w, axis=1) > 0) and np.all(np.diff(h, axis=1) > 0)
    X = np.asfortranarray(rng.uniform(-2.9, 2.9, size=(d, 50)))
    Y, l = orc.rqs(w, h, dv, X)
    Xb, lb = orc.rqs(w, h, dv, Y, inverse=True)
    np.testing.assert_allclose(Xb, X, atol=1e-10)
    np.testing.assert_allclose(lb, -l, atol=1e-9)
    assert np.all(np.abs(Y) < B)
    x = X[:, 0].copy()
    assert _jac_logabsdet(lambda t: orc.rqs(w, h, dv, t)[0], x, h=1e-7) == pytest.approx(l[0], abs=1e-5)
    # monotone
    xs = np.linspace(-3.5, 3.5, 200)
    ys = np.array([orc.rqs(w, h, dv, np.full(d, t))[0] for t in xs])
    assert np.all(np.diff(ys, axis=0) > 0)


def test_float32_type_preservation(orc):
    # test/bijectors/utils.jl:85-90
    x = np.random.default_rng(9).normal(size=(4, 3)).astype(np.float32, order="F")
    for ops in ([(orc.OP_EXP, None, None)], [(orc.OP_SCALE, 0.5, None), (orc.OP_SHIFT, 0.1, None), (orc.OP_EXP, None, None)]):
        y, l = orc.chain(ops, x)
        assert y.dtype == np.float32 and isinstance(l, np.float32)


# ------------------------------------------------------------------ pullbacks
def _fd_vjp(f, x, out_bar, ladj_bar, h=1e-6):
    """central-difference J^T out_bar + ladj_bar * grad(ladj) of f: x -> (y, ladj_per_column), column by column"""
    x = np.array(x, dtype=np.float64)
    g = np.zeros_like(x)
    for i in range(x.shape[0]):
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        yp, lp = f(xp)
        ym, lm = f(xm)
        g[i] = ((yp - ym) * out_bar).sum(axis=0) / (2 * h) + ladj_bar * (lp - lm) / (2 * h)
    return g


def test_ordered_pullbacks_match_finite_differences(orc):
    """The reference ships these rrules (ext/BijectorsChainRulesCoreExt.jl:65-197) and tests them against
    finite differences (test/ad/chainrules.jl); the restatement is pinned the same way."""
    r = np.random.default_rng(5)
    n, N = 6, 4
    y = np.asfortranarray(r.normal(size=(n, N)))
    gbar = r.normal(size=(n, N))
    lbar = r.normal(size=N)
    fwd = lambda v: orc.ordered(np.asfortranarray(v))
    got = orc.ordered_vjp(y, gbar, lbar)
    np.testing.assert_allclose(got, _fd_vjp(fwd, y, gbar, lbar), rtol=1e-6, atol=1e-7)
    x, _ = orc.ordered(y)
    inv = lambda v: orc.ordered(np.asfortranarray(v), inverse=True)
    got_inv = orc.ordered_vjp(x, gbar, lbar, inverse=True)
    np.testing.assert_allclose(got_inv, _fd_vjp(inv, x, gbar, lbar), rtol=1e-6, atol=1e-7)
    # pullback(inverse) at b(y) is the inverse-transpose of pullback(forward) at y (ladj terms off)
    back = orc.ordered_vjp(x, orc.ordered_vjp(y, gbar), inverse=True)
    np.testing.assert_allclose(back, gbar, rtol=1e-10, atol=1e-12)


def test_vec_cholesky_inverse_pullback_matches_finite_differences(orc):
    """corr.jl:402-451 restated; pinned like the reference pins it (finite differences, test/ad/chainrules.jl)."""
    r = np.random.default_rng(6)
    K, N = 5, 3
    n = K * (K - 1) // 2
    y = np.asfortranarray(0.6 * r.normal(size=(n, N)))
    Wbar = r.normal(size=(K, K, N))
    lbar = r.normal(size=N)
    for uplo in ("U", "L"):
        got = orc.vec_cholesky_inv_vjp(y, Wbar, lbar, uplo=uplo)
        fd = np.zeros_like(y)
        h = 1e-6
        for i in range(n):
            yp, ym = y.copy(), y.copy()
            yp[i] += h
            ym[i] -= h
            Wp, lp = orc.vec_cholesky(np.asfortranarray(yp), inverse=True, uplo=uplo)
            Wm, lm = orc.vec_cholesky(np.asfortranarray(ym), inverse=True, uplo=uplo)
            fd[i] = ((Wp - Wm) * Wbar).sum(axis=(0, 1)) / (2 * h) + lbar * (lp - lm) / (2 * h)
        np.testing.assert_allclose(got, fd, rtol=1e-6, atol=1e-7, err_msg=uplo)


def test_elementwise_chain_pullback_matches_finite_differences(orc):
    r = np.random.default_rng(7)
    dim, N = 5, 4
    a_vec = np.linspace(0.5, 1.5, dim)
    cases = [
        ([(orc.OP_SCALE, a_vec, None), (orc.OP_SHIFT, 0.1, None), (orc.OP_EXP, None, None)], lambda: r.normal(size=(dim, N))),
        ([(orc.OP_LOGIT, -1.0, 2.0)], lambda: r.uniform(-0.8, 1.8, size=(dim, N))),
        ([(orc.OP_LOGIT_INV, -1.0, 2.0)], lambda: r.normal(size=(dim, N))),
        ([(orc.OP_LOG, None, None), (orc.OP_SIGNFLIP, None, None)], lambda: r.uniform(0.2, 3.0, size=(dim, N))),
        ([(orc.OP_LEAKY_RELU, 0.1, None)], lambda: r.normal(size=(dim, N))),
        ([(orc.OP_TRUNCATED, 0.0, 3.0)], lambda: r.uniform(0.2, 2.8, size=(dim, N))),
        ([(orc.OP_TRUNCATED_INV, 0.0, np.inf), (orc.OP_SCALE_INV, 2.0, None)], lambda: r.normal(size=(dim, N))),
    ]
    h = 1e-6
    for ops, gen in cases:
        x = np.asfortranarray(gen())
        ybar, lbar = r.normal(size=(dim, N)), r.normal(size=N)
        got = orc.chain_vjp(ops, x, ybar, lbar)
        fd = np.zeros_like(x)
        for i in range(dim):                      # the Jacobian is diagonal: perturb one row at a time
            xp, xm = x.copy(), x.copy()
            xp[i] += h
            xm[i] -= h
            lp = np.array([float(orc.chain(ops, np.asfortranarray(xp[:, [c]]))[1]) for c in range(N)])
            lm = np.array([float(orc.chain(ops, np.asfortranarray(xm[:, [c]]))[1]) for c in range(N)])
            yp, ym = orc.chain(ops, np.asfortranarray(xp))[0], orc.chain(ops, np.asfortranarray(xm))[0]
            fd[i] = (yp[i] - ym[i]) / (2 * h) * ybar[i] + lbar * (lp - lm) / (2 * h)
        np.testing.assert_allclose(got, fd, rtol=2e-6, atol=2e-6, err_msg=str(ops))


def test_simplex_pullbacks_match_finite_differences(orc):
    """O(K) reverse sweeps of simplex.jl:47-64 / :102-120 / :122-138 (the reference's adjoints, simplex.jl:145-470,
    are O(K²) loops over the same terms), pinned by central differences of the golden-pinned forward oracle."""
    r = np.random.default_rng(3)
    K, N, h = 6, 4, 1e-6
    lb = r.normal(size=N)
    y = np.asfortranarray(r.normal(size=(K - 1, N)))
    gb = r.normal(size=(K, N))
    got = orc.simplex_vjp(y, gb, lb, inverse=True)
    fd = np.zeros_like(y)
    for i in range(K - 1):
        yp, ym = y.copy(), y.copy()
        yp[i] += h
        ym[i] -= h
        xp, lp = orc.simplex(np.asfortranarray(yp), inverse=True)
        xm, lm = orc.simplex(np.asfortranarray(ym), inverse=True)
        fd[i] = ((xp - xm) * gb).sum(0) / (2 * h) + lb * (lp - lm) / (2 * h)
    np.testing.assert_allclose(got, fd, rtol=1e-6, atol=1e-7)
    x = np.asfortranarray(r.dirichlet(np.ones(K), size=N).T)
    gb2 = r.normal(size=(K - 1, N))
    got = orc.simplex_vjp(x, gb2, lb)
    fd = np.zeros_like(x)
    for i in range(K):
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        yp, lp = orc.simplex(np.asfortranarray(xp))
        ym, lm = orc.simplex(np.asfortranarray(xm))
        fd[i] = ((yp - ym) * gb2).sum(0) / (2 * h) + lb * (lp - lm) / (2 * h)
    np.testing.assert_allclose(got, fd, rtol=1e-6, atol=1e-6)


def test_planar_pullback_matches_finite_differences(orc):
    """Input pullback of a PlanarLayer stack (planar_layer.jl:65-110): closed-form derivatives pinned against central
    differences of the golden-pinned forward oracle, one layer and a fused stack, with and without ℓ̄."""
    r = np.random.default_rng(9)
    for dim, nl, N in ((5, 1, 3), (6, 3, 4), (12, 8, 2)):
        w = r.normal(size=(dim, nl)) / np.sqrt(dim)
        u = r.normal(size=(dim, nl)) / np.sqrt(dim)
        b = r.normal(size=nl)
        z = np.asfortranarray(r.normal(size=(dim, N)))
        gbar, lbar = r.normal(size=(dim, N)), r.normal(size=N)
        fwd = lambda v: orc.planar(w, u, b, np.asfortranarray(v))
        np.testing.assert_allclose(orc.planar_vjp(w, u, b, z, gbar, lbar), _fd_vjp(fwd, z, gbar, lbar), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(orc.planar_vjp(w, u, b, z, gbar), _fd_vjp(fwd, z, gbar, np.zeros(N)), rtol=1e-6, atol=1e-7)
        inv = lambda v: orc.planar(w, u, b, np.asfortranarray(v), inverse=True)
        np.testing.assert_allclose(orc.planar_inv_vjp(w, u, b, z, gbar, lbar), _fd_vjp(inv, z, gbar, lbar), rtol=1e-6, atol=1e-7)


def test_forward_lkj_link_pullback_on_the_constraint_manifold(orc):
    """test/bijectors/chainrules.jl:57-150: the rule of `_link_chol_lkj_from_upper/lower` is tested by the reference
    against finite differences taken THROUGH the free parameters (strict triangle; the diagonal follows from the
    unit-norm constraint).  Same check for the restatement, K = 3 and 5, both triangles."""
    r = np.random.default_rng(11)
    for K in (3, 5):
        n = K * (K - 1) // 2
        y0 = 0.6 * r.normal(size=n)
        W, _ = orc.vec_cholesky(y0, inverse=True, uplo="U")                # a valid upper factor
        ybar = r.normal(size=n) * 3
        iu = [(i, j) for j in range(1, K) for i in range(j)]               # free parameters in packed order

        def from_free(v):
            A = np.zeros((K, K))
            A[0, 0] = 1.0
            for (i, j), val in zip(iu, v):
                A[i, j] = val
            for j in range(1, K):
                A[j, j] = np.sqrt(1.0 - np.sum(A[:j, j] ** 2))
            return A

        for uplo in ("U", "L"):
            def f(v):
                A = from_free(v)
                y, _ = orc.vec_cholesky(np.asfortranarray(A if uplo == "U" else A.T), uplo=uplo)
                return float(np.dot(y, ybar))
            v0 = np.array([W[i, j] for (i, j) in iu])
            np.testing.assert_allclose(from_free(v0), W, atol=1e-12)
            fd = np.zeros(n)
            h = 1e-6
            for k in range(n):
                vp, vm = v0.copy(), v0.copy()
                vp[k] += h
                vm[k] -= h
                fd[k] = (f(vp) - f(vm)) / (2 * h)
            got = orc.vec_cholesky_fwd_vjp(W if uplo == "U" else W.T, ybar, uplo=uplo)
            gotU = got if uplo == "U" else got.T
            np.testing.assert_allclose(np.array([gotU[i, j] for (i, j) in iu]), fd, rtol=1e-6, atol=1e-7)
            assert np.all(np.diag(got) == 0) and np.all((np.tril(gotU, -1)) == 0)


def test_radial_pullback_matches_finite_differences(orc):
    """RadialLayer and its inverse (radial_layer.jl:43-129): closed-form pullbacks (J = a I + c δδᵀ, Sherman–Morrison for
    the inverse) against central differences of the golden-pinned oracle."""
    r = np.random.default_rng(13)
    for dim, N in ((5, 3), (12, 4)):
        al, be, z0 = np.array([0.3]), np.array([0.7]), r.normal(size=dim)
        x = np.asfortranarray(r.normal(size=(dim, N)))
        gbar, lbar = r.normal(size=(dim, N)), r.normal(size=N)
        for inv in (False, True):
            f = lambda v: orc.radial(al, be, z0, np.asfortranarray(v), inverse=inv)
            np.testing.assert_allclose(orc.radial_vjp(al, be, z0, x, gbar, lbar, inverse=inv), _fd_vjp(f, x, gbar, lbar), rtol=1e-6, atol=1e-7)


def test_planar_parameter_pullback_matches_finite_differences(orc):
    """(w̄, ū, b̄) of a PlanarLayer stack incl. the chain rule through get_u_hat (planar_layer.jl:65-70), against central
    differences of the golden-pinned forward oracle with respect to the parameters."""
    r = np.random.default_rng(15)
    for dim, nl, N in ((5, 1, 3), (6, 3, 4)):
        w = r.normal(size=(dim, nl)) / np.sqrt(dim)
        u = r.normal(size=(dim, nl)) / np.sqrt(dim)
        b = r.normal(size=nl)
        z = np.asfortranarray(r.normal(size=(dim, N)))
        yb, lb = r.normal(size=(dim, N)), r.normal(size=N)
        wb, ub, bb = orc.planar_param_vjp(w, u, b, z, yb, lb)

        def F(w_, u_, b_):
            y, l = orc.planar(w_, u_, b_, z)
            return float((y * yb).sum() + (l * lb).sum())
        h = 1e-6
        for arr, grad, which in ((w, wb, 0), (u, ub, 1)):
            for i in range(dim):
                for k in range(nl):
                    ap, am = arr.copy(), arr.copy()
                    ap[i, k] += h
                    am[i, k] -= h
                    fd = (F(ap, u, b) - F(am, u, b)) / (2 * h) if which == 0 else (F(w, ap, b) - F(w, am, b)) / (2 * h)
                    assert abs(fd - grad[i, k]) < 1e-6
        for k in range(nl):
            bp, bm = b.copy(), b.copy()
            bp[k] += h
            bm[k] -= h
            assert abs((F(w, u, bp) - F(w, u, bm)) / (2 * h) - bb[k]) < 1e-6


def test_radial_parameter_pullback_matches_finite_differences(orc):
    """(ᾱ_, β̄, z̄₀) of a RadialLayer for the raw parameters behind softplus (radial_layer.jl:43-60), against central
    differences of the golden-pinned forward oracle with respect to the parameters."""
    r = np.random.default_rng(16)
    for dim, N, a_raw, b_raw in ((5, 4, 0.3, -0.2), (3, 6, -1.0, 1.5), (8, 2, 2.0, 0.1)):
        z0 = r.normal(size=dim)
        z = np.asfortranarray(r.normal(size=(dim, N)))
        yb, lb = r.normal(size=(dim, N)), r.normal(size=N)
        ab, bb, z0b = orc.radial_param_vjp(np.array([a_raw]), np.array([b_raw]), z0, z, yb, lb)

        def F(a_, b_, z0_):
            y, l = orc.radial(np.array([a_]), np.array([b_]), z0_, z)
            return float((y * yb).sum() + (l * lb).sum())
        h = 1e-6
        assert abs((F(a_raw + h, b_raw, z0) - F(a_raw - h, b_raw, z0)) / (2 * h) - ab) < 1e-6
        assert abs((F(a_raw, b_raw + h, z0) - F(a_raw, b_raw - h, z0)) / (2 * h) - bb) < 1e-6
        for i in range(dim):
            zp, zm = z0.copy(), z0.copy()
            zp[i] += h
            zm[i] -= h
            assert abs((F(a_raw, b_raw, zp) - F(a_raw, b_raw, zm)) / (2 * h) - z0b[i]) < 1e-6


def test_rqs_pullback_matches_finite_differences(orc):
    """Elementwise spline and its inverse: closed-form f' and (log f')' against central differences of the golden-pinned
    oracle, inside and outside [-B, B]."""
    r = np.random.default_rng(17)
    dim, K, N = 3, 6, 5
    w, h, d = orc.rqs_params(r.normal(size=(dim, K)), r.normal(size=(dim, K)), r.normal(size=(dim, K - 1)), 2.5)
    x = np.asfortranarray(r.normal(size=(dim, N)) * 1.2)
    x[0, 0] = 4.0
    gbar, lbar = r.normal(size=(dim, N)), r.normal(size=N)
    for inv in (False, True):
        f = lambda v: orc.rqs(w, h, d, np.asfortranarray(v), inverse=inv)
        np.testing.assert_allclose(orc.rqs_vjp(w, h, d, x, gbar, lbar, inverse=inv), _fd_vjp(f, x, gbar, lbar), rtol=1e-6, atol=1e-7)


def test_rqs_knot_pullback_matches_finite_differences(orc):
    """Cotangents of the knot arrays (orc.rqs_vjp_knots) and of the B-constructor's raw parameters (orc.rqs_params_vjp)
    against central differences of the golden-pinned oracle, forward and inverse, knot by knot."""
    r = np.random.default_rng(23)
    dim, K, N = 3, 6, 9
    raw = [r.normal(size=(dim, K)), r.normal(size=(dim, K)), r.normal(size=(dim, K - 1))]
    B = 2.5
    w, h, d = orc.rqs_params(*raw, B)
    x = np.asfortranarray(r.normal(size=(dim, N)) * 1.2)
    x[0, 0] = 4.0                                   # outside: contributes nothing
    gbar, lbar = r.normal(size=(dim, N)), r.normal(size=N)

    def loss(W, H, D, inv):
        y, l = orc.rqs(np.asfortranarray(W), np.asfortranarray(H), np.asfortranarray(D), x, inverse=inv)
        return float((np.asarray(y) * gbar).sum() + (np.asarray(l) * lbar).sum())

    eps = 1e-6
    for inv in (False, True):
        got = orc.rqs_vjp_knots(w, h, d, x, gbar, lbar, inverse=inv)
        for which in range(3):
            fd = np.zeros((dim, K + 1))
            for i in range(dim):
                for j in range(K + 1):
                    P = [np.array(w), np.array(h), np.array(d)]
                    M = [np.array(w), np.array(h), np.array(d)]
                    P[which][i, j] += eps
                    M[which][i, j] -= eps
                    fd[i, j] = (loss(*P, inv) - loss(*M, inv)) / (2 * eps)
            if which == 2:
                assert np.all(fd[:, -1] == 0) and np.all(got[2][:, -1] == 0)   # the last derivative is not read
            np.testing.assert_allclose(got[which], fd, rtol=2e-5, atol=2e-6)
    # constructor pullback
    wb, hb, db = r.normal(size=(dim, K + 1)), r.normal(size=(dim, K + 1)), r.normal(size=(dim, K + 1))

    def closs(rw, rh, rd):
        W, H, D = orc.rqs_params(rw, rh, rd, B)
        return float((W * wb).sum() + (H * hb).sum() + (D * db).sum())

    got = orc.rqs_params_vjp(*raw, B, wb, hb, db)
    for which in range(3):
        fd = np.zeros_like(raw[which])
        for i in range(dim):
            for j in range(raw[which].shape[1]):
                P, M = [a.copy() for a in raw], [a.copy() for a in raw]
                P[which][i, j] += eps
                M[which][i, j] -= eps
                fd[i, j] = (closs(*P) - closs(*M)) / (2 * eps)
        np.testing.assert_allclose(got[which], fd, rtol=1e-6, atol=1e-8)


# ------------------------------------------------------------------ Corr / VecCorr / PD / PDVec
def _free_to_input(kind, v, K):
    if kind == "corr":
        Y = np.zeros((K, K)); Y[np.triu_indices(K, 1)] = v; return Y
    if kind == "pd":
        Y = np.zeros((K, K)); Y[np.tril_indices(K)] = v; return Y
    return v


def _free_of_matrix(kind, X, K):
    return X[np.triu_indices(K, 1)] if kind in ("vec_corr", "corr") else X[np.tril_indices(K)]


def test_vec_corr_docstring_value(orc):
    """src/bijectors/corr.jl:113-122: the printed 3x3 correlation matrix (6 digits) -> y; round trip (:124-125)."""
    X = np.array([[1.0, -0.705273, -0.348638], [-0.705273, 1.0, 0.0534538], [-0.348638, 0.0534538, 1.0]])
    y, l = orc.vec_corr(X)
    np.testing.assert_allclose(y, [-0.8777149781928181, -0.3638927608636788, -0.29813769428942216], atol=2e-6)
    Xb, lb = orc.vec_corr(y, inverse=True)
    np.testing.assert_allclose(Xb, X, atol=1e-14)
    assert abs(lb[0] + l[0]) < 1e-13
    # CorrBijector on the same matrix holds the same values in its strict upper triangle (column-major order)
    Y, l2 = orc.corr(X)
    np.testing.assert_allclose(Y.T[np.tril_indices(3, -1)], y, atol=1e-12)   # (1,2), (1,3), (2,3)
    assert abs(l2[0] - l[0]) < 1e-12 and np.all(np.tril(Y) == 0)


@pytest.mark.parametrize("kind", ["vec_corr", "corr", "pd", "pd_vec"])
@pytest.mark.parametrize("K", [2, 3, 5, 9])
def test_matrix_bijector_properties(orc, kind, K):
    """test/bijectors/corr.jl:9-40, test/bijectors/pd.jl: round trip, ladj(inverse) = -ladj(forward), and the log-det
    against log|det| of a central-difference Jacobian of free parameters -> free entries of X (what test_bijector does
    with ForwardDiff); logabsdetjac(inverse(b), y) (_logabsdetjac_inv_corr, corr.jl:453-472) equals the fused value."""
    r = np.random.default_rng(K * 17 + len(kind))
    n_free = {"vec_corr": K * (K - 1) // 2, "corr": K * (K - 1) // 2, "pd": K * (K + 1) // 2, "pd_vec": K * (K + 1) // 2}[kind]
    v0 = 0.7 * r.normal(size=n_free)
    y0 = _free_to_input(kind, v0, K)
    X, lj = orc.matrix_bijector(kind, y0, inverse=True)
    np.testing.assert_allclose(X, X.T, atol=1e-14)
    assert np.all(np.linalg.eigvalsh(X) > 0)
    if kind in ("vec_corr", "corr"):
        np.testing.assert_allclose(np.diag(X), 1.0, atol=1e-13)
        assert abs(orc.logabsdetjac_inv_corr(y0) - lj[0]) < 1e-11 * max(1.0, abs(lj[0]))
    y1, lf = orc.matrix_bijector(kind, X)
    np.testing.assert_allclose(y1, y0, atol=1e-11)
    assert abs(lf[0] + lj[0]) < 1e-10 * max(1.0, abs(lj[0]))
    J = np.zeros((n_free, n_free))
    h = 1e-6
    for c in range(n_free):
        vp, vm = v0.copy(), v0.copy()
        vp[c] += h
        vm[c] -= h
        J[:, c] = (_free_of_matrix(kind, orc.matrix_bijector(kind, _free_to_input(kind, vp, K), inverse=True)[0], K)
                   - _free_of_matrix(kind, orc.matrix_bijector(kind, _free_to_input(kind, vm, K), inverse=True)[0], K)) / (2 * h)
    assert abs(np.linalg.slogdet(J)[1] - lj[0]) < 1e-6 * max(1.0, abs(lj[0]))


def test_pd_bijector_against_numpy_cholesky(orc):
    """pd.jl:10-31: Y = replace_diag(log, cholesky_lower(X)); logabsdetjac = -(sum((d+1):-1:2 .* log.(diag(L))) + d log 2)."""
    r = np.random.default_rng(5)
    A = r.normal(size=(6, 6))
    X = A @ A.T + 6 * np.eye(6)
    Lc = np.linalg.cholesky(X)
    Y, l = orc.pd(X)
    ref = Lc.copy()
    ref[np.diag_indices(6)] = np.log(np.diag(Lc))
    np.testing.assert_allclose(Y, ref, atol=1e-13)
    want = -(np.sum(np.arange(7, 1, -1) * np.log(np.diag(Lc))) + 6 * np.log(2.0))
    assert abs(l[0] - want) < 1e-12
    yv, l2 = orc.pd_vec(X)
    np.testing.assert_allclose(yv, np.concatenate([ref.T[:j + 1, j] for j in range(6)]), atol=1e-13)      # triu_to_vec(Y'), column-major
    assert abs(l2[0] - want) < 1e-12


def test_batchnorm_training_pullback_is_the_gradient_of_the_training_forward(orc):
    """oracle.batchnorm_train_vjp (closed form of the adjoint of normalise.jl:51-60, batch statistics differentiated) against
    central differences of oracle.batchnorm_train: L = Σ ȳ·y + Σ ℓ̄·logabsdetjac, perturbing x, b and logs."""
    r = np.random.default_rng(17)
    d, n, eps = 5, 37, 1e-5
    x = r.normal(size=(d, n)) * 1.7 + 0.4
    b, logs = r.normal(size=d), 0.3 * r.normal(size=d)
    g, lb = r.normal(size=(d, n)), r.normal(size=n)

    def loss(x_, b_, logs_):
        y, l, _, _ = orc.batchnorm_train(b_, logs_, np.zeros(d), np.ones(d), eps, 0.1, x_)
        return float((g * y).sum() + (lb * l).sum())

    xb, bb, lgb = orc.batchnorm_train_vjp(logs, eps, x, g, lb)
    h = 1e-6
    for (i, k) in [(0, 0), (2, 11), (4, 36), (1, 5)]:
        xp, xm = x.copy(), x.copy()
        xp[i, k] += h
        xm[i, k] -= h
        fd = (loss(xp, b, logs) - loss(xm, b, logs)) / (2 * h)
        assert abs(fd - xb[i, k]) <= 2e-6 * max(1.0, abs(fd)), (i, k, fd, xb[i, k])
    for c in range(d):
        e = np.zeros(d)
        e[c] = h
        assert abs((loss(x, b + e, logs) - loss(x, b - e, logs)) / (2 * h) - bb[c]) <= 2e-6 * max(1.0, abs(bb[c]))
        assert abs((loss(x, b, logs + e) - loss(x, b, logs - e)) / (2 * h) - lgb[c]) <= 2e-6 * max(1.0, abs(lgb[c]))


def test_full_covariance_normal_density_against_scipy(orc):
    from scipy.stats import multivariate_normal

    r = np.random.default_rng(3)
    d = 6
    A = r.normal(size=(d, d))
    cov, mu = A @ A.T + 0.5 * np.eye(d), r.normal(size=d)
    x = r.normal(size=(d, 40)) * 2.0
    np.testing.assert_allclose(orc.mvnormal_full_logpdf(x, mu, cov), multivariate_normal(mean=mu, cov=cov).logpdf(x.T), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("inverse", [True, False])
@pytest.mark.parametrize("K", [1, 2, 3, 5, 7])
@pytest.mark.parametrize("kind", ["vec_corr", "corr", "pd", "pd_vec"])
def test_matrix_bijector_pullbacks_against_central_differences(orc, kind, K, inverse):
    """oracle.matrix_bijector_vjp (the reference's rules chained: pd_from_upper ext/BijectorsChainRulesCoreExt.jl:324-331,
    pd_from_lower / replace_diag ext/BijectorsReverseDiffExt.jl:143-168, _inv_link_chol_lkj corr.jl:402-451; the forward links and
    the Cholesky reverse derived) is the gradient of Σ out_bar·out + Σ ladj_bar·logabsdetjac of oracle.matrix_bijector.  The
    forward direction is differentiated entry by entry of X: only the triangle the reference reads carries a gradient."""
    r = np.random.default_rng(1000 * K + 10 * len(kind) + int(inverse))
    N = 3
    n_free = K * (K - 1) // 2 if kind in ("vec_corr", "corr") else K * (K + 1) // 2
    v0 = 0.6 * r.normal(size=(n_free, N))
    y0 = np.asfortranarray(np.stack([_free_to_input(kind, v0[:, n], K) for n in range(N)], axis=-1))
    X0, _ = orc.matrix_bijector(kind, y0, inverse=True)
    inp = y0 if inverse else np.asfortranarray(X0)
    out0, l0 = orc.matrix_bijector(kind, inp, inverse=inverse)
    gbar, lbar = r.normal(size=out0.shape), r.normal(size=N)

    def loss(a):
        o, l = orc.matrix_bijector(kind, np.asfortranarray(a), inverse=inverse)
        return (gbar * o).sum(axis=tuple(range(o.ndim - 1))) + lbar * l

    got = orc.matrix_bijector_vjp(kind, inp, gbar, lbar, inverse=inverse)
    assert got.shape == inp.shape
    want = np.zeros_like(inp)
    h = 1e-6
    for idx in np.ndindex(*inp.shape[:-1]):
        ap, am = inp.copy(), inp.copy()
        ap[idx] += h
        am[idx] -= h
        want[idx] = (loss(ap) - loss(am)) / (2 * h)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7 * max(1.0, float(np.abs(want).max()) if want.size else 1.0))
    if not inverse and K > 1:       # the other triangle is never read (cholesky(Hermitian(X)) / Hermitian(X, :L), src/utils.jl:37,50)
        other = np.tril_indices(K, -1) if kind in ("vec_corr", "corr") else np.triu_indices(K, 1)
        assert np.all(got[other] == 0) and np.all(np.abs(want[other]) < 1e-9)
