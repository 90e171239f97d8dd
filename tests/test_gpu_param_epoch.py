"""BJX_OPT_PARAM_EPOCH (include/bjx.h; VERDICT r03 weak #3: the spline's knot blob was rebuilt on every call although the parameters
had not changed): the library keeps the LDS blob of an unchanged spline; the Python host moves the epoch exactly when a parameter
tensor was written or replaced.  Same bits with and without reuse; an in-place update, a new tensor at a recycled address and a
stream change all rebuild."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from test_gpu_parity import bj, dev, host, rng  # noqa: E402,F401


@pytest.fixture(autouse=True)
def _reuse_on(bj):
    """The reuse of parameter tables is opt-in since round 5: these tests are about the opted-in behaviour."""
    with bj.cache_params():
        yield


def _spline(bj, r, dim=32, K=16):
    raw = [torch.tensor(r.normal(size=(dim, k)).astype(np.float32)).cuda() for k in (K, K, K - 1)]
    return bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)


def test_unchanged_spline_reuses_its_blob_and_updates_rebuild_it(bj):
    r = rng(201)
    sp = _spline(bj, r)
    x = dev(np.asfortranarray(r.normal(size=(32, 4099)).astype(np.float32)))
    ctx = bj.context(x.device)
    st = bj.interface._PARAM_WATCH.setdefault(id(ctx), [1, 0, {}])
    # first calls: the blobs are built (forward and inverse tables are two slots of the cache)
    y0, l0 = bj.with_logabsdet_jacobian(sp, x)
    xb0, lb0 = bj.with_logabsdet_jacobian(bj.inverse(sp), y0)
    for _ in range(3):                                       # forward and inverse alternate: two slots of the cache
        y1, l1 = bj.with_logabsdet_jacobian(sp, x)
        xb1, lb1 = bj.with_logabsdet_jacobian(bj.inverse(sp), y1)
        assert torch.equal(y0, y1) and torch.equal(l0, l1) and torch.equal(xb0, xb1) and torch.equal(lb0, lb1)
    epoch = st[0]
    # an optimiser step: the knots change IN PLACE (same addresses) -> the epoch moves, the blob is rebuilt
    with torch.no_grad():
        sp.widths.mul_(0.75)
    y2, l2 = bj.with_logabsdet_jacobian(sp, x)
    assert st[0] != epoch and not torch.equal(y2, y0)
    fresh = bj.RationalQuadraticSpline(sp.widths.clone(), sp.heights.clone(), sp.derivatives.clone())
    y3, l3 = bj.with_logabsdet_jacobian(fresh, x)
    assert torch.equal(y2, y3) and torch.equal(l2, l3)
    # two different splines interleaved keep their own tables
    other = _spline(bj, r)
    ya, la = bj.with_logabsdet_jacobian(other, x)
    for _ in range(2):
        assert torch.equal(bj.with_logabsdet_jacobian(sp, x)[0], y2)
        assert torch.equal(bj.with_logabsdet_jacobian(other, x)[0], ya)


def test_a_new_tensor_at_a_recycled_address_is_not_mistaken_for_the_old_one(bj):
    r = rng(202)
    x = dev(np.asfortranarray(r.normal(size=(8, 513)).astype(np.float32)))
    outs = []
    for i in range(6):
        w = torch.tensor(np.cumsum(np.abs(r.normal(size=(8, 9))) + 0.1, axis=1).astype(np.float32)).cuda()
        w = (w - w[:, -1:] / 2)                                   # symmetric-ish knots around 0
        w = w.T.contiguous().T
        h = w.clone().T.contiguous().T
        d = torch.ones(8, 9, device="cuda").T.contiguous().T
        sp = bj.RationalQuadraticSpline(w, h, d)                  # identity-like spline with DIFFERENT knots every round
        y, l = bj.with_logabsdet_jacobian(sp, x)
        ref = bj.with_logabsdet_jacobian(bj.RationalQuadraticSpline(w.clone(), h.clone(), d.clone()), x)
        assert torch.equal(y, ref[0]) and torch.equal(l, ref[1])
        outs.append(w.data_ptr())
        del sp, w, h, d                                           # the caching allocator hands the same blocks to the next round
    assert len(set(outs)) < len(outs), "the allocator did not recycle an address: the test did not exercise what it is for"


def test_merged_affine_tail_of_a_density_chain_follows_in_place_updates(bj):
    """logpdf(transformed(MvNormal(μ, σ), exp ∘ Shift ∘ Scale)) collapses the affine stages at the end of its chain (tail of the inverse
    transform + the whitening) into one Scale and one Shift with per-row vectors kept on the distribution (`_merge_affine_tail`,
    src/transformed_distribution.jl:164-169): the kept vectors must follow an IN-PLACE update of μ or σ, and the value must be the density."""
    import numpy as np
    torch.manual_seed(3)
    dim, N = 64, 1000
    mu = torch.randn(dim, device="cuda")
    sigma = torch.rand(dim, device="cuda") + 0.5
    b = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    td = bj.transformed(bj.MvNormal(mu, sigma), b)
    y = torch.exp(0.3 * torch.randn(N, dim, device="cuda")).T

    def ref():
        x = (torch.log(y.double()) - 0.1) / 0.5
        z = (x - mu.double()[:, None]) / sigma.double()[:, None]
        return (-0.5 * z * z - 0.5 * np.log(2 * np.pi) - torch.log(sigma.double())[:, None]).sum(0) - (torch.log(y.double()).sum(0) + dim * np.log(0.5))

    lp0 = bj.logpdf(td, y)
    assert "_affine_runs" in td.dist.__dict__, "the density chain did not take the merged form"
    assert torch.allclose(lp0.double(), ref(), rtol=1e-4, atol=1e-3 * dim)
    mu.add_(0.7)                                            # in place: same tensor object, new version
    sigma.mul_(1.3)
    lp1 = bj.logpdf(td, y)
    assert torch.allclose(lp1.double(), ref(), rtol=1e-4, atol=1e-3 * dim)
    assert not torch.allclose(lp1, lp0)
