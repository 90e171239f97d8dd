"""Regression tests for the advisor's round-4 findings, on the GPU through the library."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from test_gpu_parity import bj, close, dev, host, rng  # noqa: E402,F401


def test_batchnorm_last_stage_of_a_composition_training_pullback(bj):
    """`vjp_params` of a composition whose LAST stage is a training-mode InvertibleBatchNorm: the stage inputs are recomputed
    by the pullback and the last stage's forward is never re-run, so in round 4 its saved statistics carried the tag of another
    tensor and the call raised.  Checked against the chain rule applied stage by stage; the recomputation must not move the
    moving statistics a second time."""
    r = rng(21)
    d, N = 6, 257
    xh = np.asfortranarray(r.normal(size=(d, N)).astype(np.float64))
    gh = np.asfortranarray(r.normal(size=(d, N)).astype(np.float64))
    lbh = r.normal(size=N)
    w = torch.tensor(r.normal(size=d) / np.sqrt(d), device="cuda")
    u = torch.tensor(r.normal(size=d) / np.sqrt(d), device="cuda")
    b0 = torch.tensor(r.normal(size=1), device="cuda")
    layer = bj.PlanarLayer(w, u, b0)
    bn = bj.InvertibleBatchNorm(torch.tensor(r.normal(size=d)), torch.tensor(0.2 * r.normal(size=d)), torch.zeros(d, dtype=torch.float64),
                                torch.ones(d, dtype=torch.float64))
    comp = bn @ layer
    x, g, lb = dev(xh), dev(gh), dev(lbh)
    with bj.training():
        y, l = bj.with_logabsdet_jacobian(comp, x, per_sample=True)
        m_after, v_after = bn.m.clone(), bn.v.clone()
        xb, grads = bj.vjp_params(comp, x, g, lb)
        assert torch.equal(bn.m, m_after) and torch.equal(bn.v, v_after), "the pullback's recomputation moved the moving statistics"
        # stage by stage: pull (g, lb) back through the BatchNorm at its input z, then through the layer at x
        z = bj.transform(layer, x)
        zb, gr_bn = bj.vjp_params(bn, z, g, lb)
        xb_ref, gr_layer = bj.vjp_params(layer, x, zb, lb)
    close(host(xb), host(xb_ref).astype(np.float64), np.float64, scale=10, what="x_bar through BatchNorm ∘ Planar")
    st = grads["stages"]
    close(host(st[1]["logs"]), host(gr_bn["logs"]).astype(np.float64), np.float64, scale=100, what="logs_bar")
    close(host(st[0]["w"]).reshape(-1), host(gr_layer["w"]).astype(np.float64).reshape(-1), np.float64, scale=100, what="w_bar")


def test_parameter_writes_through_data_are_seen_by_default(bj):
    """`p.data.add_()` does not move `p._version`: with the parameter-table reuse OFF (the default since round 5) the next call
    still sees the new values — PlanarLayer runs (layer-major tables) and the spline (its LDS table)."""
    r = rng(22)
    d, N = 32, 1000
    x = dev(np.asfortranarray(r.normal(size=(d, N)).astype(np.float32)))
    layers = [bj.PlanarLayer(torch.tensor(r.normal(size=d).astype(np.float32) / 6, device="cuda"), torch.tensor(r.normal(size=d).astype(np.float32) / 6, device="cuda"),
                             torch.tensor(r.normal(size=1).astype(np.float32), device="cuda")) for _ in range(3)]
    flow = layers[2] @ layers[1] @ layers[0]
    y0 = bj.transform(flow, x).clone()
    layers[1].w.data.add_(0.25)
    y1 = bj.transform(flow, x).clone()
    ref = bj.transform(layers[2], bj.transform(layers[1], bj.transform(layers[0], x)))
    assert not torch.allclose(y0, y1)
    np.testing.assert_allclose(host(y1), host(ref), rtol=1e-5, atol=1e-5)
    K = 8
    raw = [torch.tensor(r.normal(size=(d, k)).astype(np.float32), device="cuda") for k in (K, K, K - 1)]
    sp = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)
    s0 = bj.transform(sp, x).clone()
    sp.heights.data.mul_(0.5)
    sp.heights.data[:, -1] = 3.0
    sp.heights.data[:, 0] = -3.0
    s1 = bj.transform(sp, x).clone()
    assert not torch.allclose(s0, s1), "the spline kept a table built from the old knots"


def test_cache_params_opt_in_and_invalidate(bj):
    """With `cache_params(True)` the tables are reused while `_version` stands still — a `.data` write is then NOT seen until
    `invalidate_params()`, which is what the switch documents; ordinary in-place operations are always seen."""
    r = rng(23)
    d, N = 16, 300
    x = dev(np.asfortranarray(r.normal(size=(d, N)).astype(np.float64)))
    layers = [bj.PlanarLayer(torch.tensor(r.normal(size=d) / 4, device="cuda"), torch.tensor(r.normal(size=d) / 4, device="cuda"), torch.tensor(r.normal(size=1), device="cuda"))
              for _ in range(2)]
    flow = layers[1] @ layers[0]

    def manual():
        return bj.transform(layers[1], bj.transform(layers[0], x))

    with bj.cache_params():
        bj.transform(flow, x)
        layers[0].u.add_(0.125)                                        # in place on the tensor itself: `_version` moves
        np.testing.assert_allclose(host(bj.transform(flow, x)), host(manual()), rtol=1e-12, atol=1e-12)
        layers[0].u.data.add_(0.125)                                   # invisible to the version counter
        bj.invalidate_params()
        np.testing.assert_allclose(host(bj.transform(flow, x)), host(manual()), rtol=1e-12, atol=1e-12)
    # outside the region the reuse is off again
    layers[1].w.data.add_(0.0625)
    np.testing.assert_allclose(host(bj.transform(flow, x)), host(manual()), rtol=1e-12, atol=1e-12)


def test_in_place_entries_move_the_version_counter(bj):
    x = dev(np.asfortranarray(rng(24).normal(size=(4, 50))))
    y = torch.empty_like(x)
    v0 = y._version
    bj.transform_(bj.elementwise(bj.exp), x, y)
    assert y._version > v0
    v1 = x._version
    bj.with_logabsdet_jacobian_(bj.elementwise(bj.exp) @ bj.Scale(0.5), x)      # in place on x
    assert x._version > v1


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("dim", [5, 64, 100, 128])
def test_scale_matrix_factorisation_in_lds_and_kept_per_epoch(bj, dt, dim):
    """Round 5: the [A⁻¹ | logabsdet] factorisation behind a matrix `Scale` runs in LDS (Float64 at 128 rows does not fit and keeps the
    global-memory sweep) and is kept per parameter epoch under `cache_params`; an in-place update of the matrix rebuilds it."""
    r = rng(300 + dim)
    A = (r.normal(size=(dim, dim)) / np.sqrt(dim) + 1.5 * np.eye(dim)).astype(dt)
    x = np.asfortranarray(r.normal(size=(dim, 257)).astype(dt))
    a = torch.tensor(np.asfortranarray(A).T.copy(), device="cuda").T          # column-major device matrix
    b = bj.Scale(a)
    A64, x64 = A.astype(np.float64), x.astype(np.float64)
    lad = np.linalg.slogdet(A64)[1]
    sc = 50.0 if dt == np.float32 else 1e3

    def check(a64):
        y, l = bj.with_logabsdet_jacobian(b, dev(x), per_sample=True)
        close(host(y), a64 @ x64, dt, scale=sc, what="a * x")
        close(host(l), np.full(257, np.linalg.slogdet(a64)[1]), dt, scale=sc, what="logabsdet")
        xb, li = bj.with_logabsdet_jacobian(bj.inverse(b), y, per_sample=True)
        close(host(xb), x64, dt, scale=sc * 4, what="a \\ y")
        close(host(li), np.full(257, -np.linalg.slogdet(a64)[1]), dt, scale=sc, what="-logabsdet")

    check(A64)
    with bj.cache_params():
        for _ in range(3):
            check(A64)                       # second and third round: the kept factorisation (forward calls reuse the inverse's slot)
        with torch.no_grad():
            a.mul_(1.25)                     # in place: `_version` moves, the epoch moves, the factorisation is rebuilt
        check(1.25 * A64)
    assert abs(lad) < 1e3


def test_spline_never_uses_a_stale_table_by_default(bj):
    """Default (no `cache_params`): the spline's table is rebuilt from the knots on every call, so a write the host cannot see
    (`.data`, another framework, a raw pointer) is seen by the next call — forward, inverse and the input pullback, over 300 calls
    with in-place updates in between, against a fresh spline object with cloned knots."""
    r = rng(31)
    d, K, N = 32, 16, 2049
    x = dev(np.asfortranarray(r.normal(size=(d, N)).astype(np.float32)))
    raw = [torch.tensor(r.normal(size=(d, k)).astype(np.float32), device="cuda") for k in (K, K, K - 1)]
    sp = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)
    g, lb = torch.ones_like(x), torch.zeros(N, device="cuda")

    def fresh():
        f = bj.RationalQuadraticSpline(sp.widths.clone(), sp.heights.clone(), sp.derivatives.clone())
        y, l = bj.with_logabsdet_jacobian(f, x)
        xb, li = bj.with_logabsdet_jacobian(bj.inverse(f), y)
        return y, l, xb, li, bj.vjp(f, x, g, lb)

    for it in range(300):
        if it in (3, 4, 150, 257, 258):
            sp.derivatives.data[:, 1:-1].mul_(1.0 + 0.01 * (it % 7 + 1))       # invisible to `_version`; same pointers
            sp.heights.data[:, 5].add_(1e-3)
        y, l = bj.with_logabsdet_jacobian(sp, x)
        if it in (0, 2, 3, 4, 5, 150, 151, 256, 257, 258, 259, 299):
            xb, li = bj.with_logabsdet_jacobian(bj.inverse(sp), y)
            gx = bj.vjp(sp, x, g, lb)
            ref = fresh()
            assert torch.equal(y, ref[0]) and torch.equal(l, ref[1]), f"call {it}: forward used a stale table"
            assert torch.equal(xb, ref[2]) and torch.equal(li, ref[3]), f"call {it}: inverse used a stale table"
            assert torch.equal(gx, ref[4]), f"call {it}: pullback used a stale table"
