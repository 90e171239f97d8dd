"""The composition planner: a flow written the reference's way — `l8 ∘ … ∘ l1`,
/root/reference docs/src/flows.md:115, src/bijectors/composed.jl:4-25 — must reach the FUSED kernels: one bjx_planar launch for a
run of PlanarLayers (1 028 instead of 8 224 B/sample at dim 128), in `with_logabsdet_jacobian`, `transform`, `inverse`, `vjp`,
`vjp_params` and `logpdf`; bit-equal with the stacked constructor, and equal to the layer-by-layer oracle."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from test_gpu_parity import bj, close, dev, host, rng  # noqa: E402,F401  (fixtures / helpers)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    return oracle


def _layers(bj, r, dim, nl, dt, on_device=True):
    w = (r.normal(size=(dim, nl)) / math.sqrt(dim)).astype(dt)
    u = (r.normal(size=(dim, nl)) / math.sqrt(dim)).astype(dt)
    b = r.normal(size=nl).astype(dt)
    mk = (lambda a: torch.tensor(a).cuda()) if on_device else torch.tensor
    ls = [bj.PlanarLayer(mk(np.ascontiguousarray(w[:, k])), mk(np.ascontiguousarray(u[:, k])), mk(b[k:k + 1])) for k in range(nl)]
    return ls, w, u, b


def _compose(ls):
    f = ls[0]
    for l in ls[1:]:
        f = l @ f          # l applied after f: `l ∘ f`
    return f


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("dim,nl,N", [(128, 8, 4096), (10, 3, 100), (64, 2, 513), (131, 5, 77)])
def test_composed_planar_flow_is_one_fused_launch(bj, orc, dim, nl, N, dt):
    r = rng(101)
    ls, w, u, b = _layers(bj, r, dim, nl, dt)
    flow = _compose(ls)
    Z = np.asfortranarray(r.normal(size=(dim, N)).astype(dt))
    Zd = dev(Z)
    bj.with_logabsdet_jacobian(flow, Zd)                                        # builds the plan and the parameter tables
    (y, l), _, launches = bj.kernel_timed(lambda: bj.with_logabsdet_jacobian(flow, Zd))
    assert launches == 1, f"{nl} composed PlanarLayers took {launches} hot launches"
    stacked = bj.PlanarLayer(torch.tensor(w).cuda(), torch.tensor(u).cuda(), torch.tensor(b).cuda())
    res = bj.with_logabsdet_jacobian(stacked, Zd)
    assert torch.equal(y, res.result) and torch.equal(l, res.logabsdetjac)      # same kernel, same tables: same bits
    Y_ref, l_ref = orc.planar(w, u, b, Z)
    close(host(y), Y_ref, dt, what="composed planar fwd")
    close(host(l), l_ref, dt, scale=nl, what="composed planar ladj")
    # transform alone, and the inverse flow: inverse(l_nl ∘ … ∘ l_1) = inverse(l_1) ∘ … ∘ inverse(l_nl)
    y2, _, k2 = bj.kernel_timed(lambda: bj.transform(flow, Zd))
    assert k2 == 1 and torch.equal(y2, y)
    inv = bj.inverse(flow)
    assert isinstance(inv, bj.ComposedFunction)
    bj.with_logabsdet_jacobian(inv, y)
    (zb, lb), _, k3 = bj.kernel_timed(lambda: bj.with_logabsdet_jacobian(inv, y))
    assert k3 == 1
    zs, ls_ = bj.with_logabsdet_jacobian(bj.inverse(stacked), y)
    assert torch.equal(zb, zs) and torch.equal(lb, ls_)
    np.testing.assert_allclose(host(zb), Z, rtol=1e-3 if dt == np.float32 else 1e-6, atol=(2e-3 if dt == np.float32 else 2e-8))


def test_flows_md_composition_runs_two_launches(bj, orc):
    """`PlanarLayer(10) ∘ PlanarLayer(10) ∘ RadialLayer(10)` (docs/src/flows.md:115): the radial layer, then ONE launch for the two
    planar layers; values against the oracle stage by stage."""
    r = rng(102)
    dim, N, dt = 10, 333, np.float64
    ls, w, u, b = _layers(bj, r, dim, 2, dt)
    a_, be, z0 = float(r.normal()), float(r.normal()), r.normal(size=dim)
    rad = bj.RadialLayer(torch.tensor([a_]), torch.tensor([be]), torch.tensor(z0))
    flow = ls[1] @ ls[0] @ rad
    Z = np.asfortranarray(r.normal(size=(dim, N)))
    bj.with_logabsdet_jacobian(flow, dev(Z))
    (y, l), _, launches = bj.kernel_timed(lambda: bj.with_logabsdet_jacobian(flow, dev(Z)))
    assert launches == 2
    Y1, l1 = orc.radial(a_, be, z0, Z)
    Y2, l2 = orc.planar(w, u, b, np.asfortranarray(Y1))
    close(host(y), Y2, dt, what="radial then 2 planar")
    close(host(l), l1 + l2, dt, scale=3, what="ladj")
    # an elementwise stage between planar runs splits them; the chain still evaluates
    flow2 = ls[1] @ bj.Shift(0.25) @ ls[0]
    y3, l3 = bj.with_logabsdet_jacobian(flow2, dev(Z), per_sample=True)
    Ya, la = orc.planar(w[:, :1], u[:, :1], b[:1], Z)
    Yb, lb_ = orc.planar(w[:, 1:], u[:, 1:], b[1:], np.asfortranarray(Ya + 0.25))
    close(host(y3), Yb, dt, what="planar shift planar")
    close(host(l3), la + lb_, dt, scale=2, what="ladj")


def test_plan_follows_parameter_updates(bj):
    """The gathered tables are rebuilt when a layer's parameter tensor changes in place (what an optimiser step does) and when a
    host-resident parameter is re-uploaded."""
    r = rng(103)
    dim, nl, N = 32, 4, 257
    for on_device in (True, False):
        ls, w, u, b = _layers(bj, r, dim, nl, np.float32, on_device=on_device)
        flow = _compose(ls)
        Zd = dev(np.asfortranarray(r.normal(size=(dim, N)).astype(np.float32)))
        y0, l0 = bj.with_logabsdet_jacobian(flow, Zd)
        with torch.no_grad():
            ls[2].w.mul_(0.5)
            ls[0].b.add_(0.125)
        y1, l1 = bj.with_logabsdet_jacobian(flow, Zd)
        assert not torch.equal(y0, y1)
        fresh = _compose([bj.PlanarLayer(l.w.clone(), l.u.clone(), l.b.clone()) for l in ls])
        y2, l2 = bj.with_logabsdet_jacobian(fresh, Zd)
        assert torch.equal(y1, y2) and torch.equal(l1, l2)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_pullbacks_of_a_composed_flow_use_the_fused_kernels(bj, dt):
    """vjp / vjp_params of `l4 ∘ … ∘ l1` and of its inverse: one hot launch each, the same numbers as the stacked constructor, and
    the parameter cotangents handed back stage by stage."""
    r = rng(104)
    dim, nl, N = 48, 4, 300
    ls, w, u, b = _layers(bj, r, dim, nl, dt)
    flow = _compose(ls)
    stacked = bj.PlanarLayer(torch.tensor(w).cuda(), torch.tensor(u).cuda(), torch.tensor(b).cuda())
    Zd = dev(np.asfortranarray(r.normal(size=(dim, N)).astype(dt)))
    G = dev(np.asfortranarray(r.normal(size=(dim, N)).astype(dt)))
    lbar = torch.tensor(r.normal(size=N).astype(dt)).cuda()
    bj.vjp(flow, Zd, G, lbar)
    xb, _, k = bj.kernel_timed(lambda: bj.vjp(flow, Zd, G, lbar))
    assert k == 1
    assert torch.equal(xb, bj.vjp(stacked, Zd, G, lbar))
    xb2, gr = bj.vjp_params(flow, Zd, G, lbar)
    xb3, gs = bj.vjp_params(stacked, Zd, G, lbar)
    assert torch.equal(xb2, xb3) and len(gr["stages"]) == nl
    for k_, st in enumerate(gr["stages"]):
        assert torch.equal(st["w"], gs["w"][:, k_]) and torch.equal(st["u"], gs["u"][:, k_]) and torch.equal(st["b"], gs["b"][k_:k_ + 1])
    # the inverse flow: stages are inverse(l4), …, inverse(l1) in application order
    inv = bj.inverse(flow)
    Y = bj.transform(flow, Zd)
    yb, gi = bj.vjp_params(inv, Y, G, lbar)
    yb_s, gis = bj.vjp_params(bj.inverse(stacked), Y, G, lbar)
    assert torch.equal(yb, yb_s)
    st_inv = inv._stages()
    assert [s.orig for s in st_inv] == list(reversed(ls))
    for j, st in enumerate(gi["stages"]):
        k_ = nl - 1 - j                       # stage j inverts layer nl-1-j
        np.testing.assert_array_equal(host(st["w"]), host(gis["w"][:, k_]))
        np.testing.assert_array_equal(host(st["b"]), host(gis["b"][k_:k_ + 1]))


def test_logpdf_of_a_composed_flow_is_the_fused_inverse(bj):
    r = rng(105)
    dim, nl, N = 16, 3, 500
    ls, w, u, b = _layers(bj, r, dim, nl, np.float64)
    td = bj.transformed(bj.MvNormal(dim), _compose(ls))
    td_s = bj.transformed(bj.MvNormal(dim), bj.PlanarLayer(torch.tensor(w).cuda(), torch.tensor(u).cuda(), torch.tensor(b).cuda()))
    Y = dev(np.asfortranarray(r.normal(size=(dim, N))))
    bj.logpdf(td, Y)
    lp, _, k = bj.kernel_timed(lambda: bj.logpdf(td, Y))
    assert k == 1
    assert torch.equal(lp, bj.logpdf(td_s, Y))
