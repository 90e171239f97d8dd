"""BASELINE.json's FULL-SIZE configurations on the GPU, checked through size-independent properties
(round trips, log-det antisymmetry, sum-of-parts, shard invariance, structural invariants) plus an
oracle spot check on randomly sampled columns of the full-size result.  The oracle cannot run the full
sizes in seconds; `tests/test_gpu_parity.py` covers oracle parity at small sizes."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bj():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import bijectors_amd

    bijectors_amd._lib.load()
    return bijectors_amd


def cm(rows, batch, dtype=torch.float32):
    return torch.empty((batch, rows), dtype=dtype, device="cuda").T


def fill(bj, t, seed, std=1.0, mean=0.0, col0=0):
    L, ctx = bj._lib, bj.context(t.device)
    rows, batch = t.shape
    L.check(ctx.h, L.load().bjx_fill_normal(ctx.h, L.BJX_F32, t.data_ptr(), rows, batch, col0, seed, mean, std), "fill")
    return t


def sample_cols(t, idx):
    return np.asfortranarray(t[:, idx].cpu().numpy().astype(np.float64))


def test_c2_fused_chain_full_size(bj, orc):
    """configs[1]: exp ∘ Shift ∘ Scale, Float32, dim = 64, batch = 2^24."""
    d, N = 64, 1 << 24
    x = fill(bj, cm(d, N), 0)
    b = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    y, lps, lsum = bj.shard.with_logabsdet_jacobian_sharded(b, x)
    # closed form of the log-det: Σ (0.5 x + 0.1) + d·log 0.5 per column
    ref_ps = (0.5 * x.double().sum(dim=0) + d * 0.1 + d * math.log(0.5))
    assert torch.allclose(lps.double(), ref_ps, rtol=1e-4, atol=1e-3)
    assert abs(float(lsum) - float(ref_ps.sum())) <= 1e-6 * abs(float(ref_ps.sum()))
    # the float64 global sum is the sum of the per-sample values (different summation orders)
    assert abs(float(lsum) - float(lps.double().sum())) <= 1e-6 * abs(float(lsum))
    # inverse(b)(b(x)) == x and the log-dets cancel
    xb, lps_inv, _ = bj.shard.with_logabsdet_jacobian_sharded(bj.inverse(b), y)
    assert torch.allclose(xb, x, rtol=1e-4, atol=1e-4)
    assert torch.allclose(lps_inv, -lps, rtol=1e-4, atol=1e-2)
    # shard invariance: 8 contiguous column blocks reduce to the same float64 sum
    parts = sum(float(bj.shard.with_logabsdet_jacobian_sharded(b, x[:, lo:hi], per_sample=False)[2])
                for lo, hi in (bj.shard.shard_columns(N, 8, r) for r in range(8)))
    assert abs(parts - float(lsum)) <= 1e-9 * abs(float(lsum))
    # oracle spot check on sampled columns of the full-size output
    idx = torch.randint(0, N, (512,), generator=torch.Generator().manual_seed(0)).cuda()
    ys, _ = orc.chain([(orc.OP_SCALE, 0.5, None), (orc.OP_SHIFT, 0.1, None), (orc.OP_EXP, None, None)], sample_cols(x, idx).astype(np.float32))
    np.testing.assert_allclose(y[:, idx].cpu().numpy(), ys, rtol=1e-3)


def test_c3_rqs_full_size(bj, orc):
    """configs[2]: RationalQuadraticSpline K = 16, dim = 32, batch = 2^22, forward + inverse."""
    d, K, N = 32, 16, 1 << 22
    x = fill(bj, cm(d, N), 0)
    raw = [fill(bj, cm(d, k), 100 + i) for i, k in enumerate((K, K, K - 1))]
    b = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)
    y, lps, lsum = bj.shard.with_logabsdet_jacobian_sharded(b, x)
    xb, lps_inv, lsum_inv = bj.shard.with_logabsdet_jacobian_sharded(bj.inverse(b), y)
    assert torch.allclose(xb, x, rtol=1e-3, atol=2e-4)
    assert torch.allclose(lps_inv, -lps, rtol=1e-3, atol=2e-3)
    assert abs(float(lsum) + float(lsum_inv)) <= 1e-5 * abs(float(lsum)) + 1.0
    outside = x.abs() >= 3.0
    assert bool((y[outside] == x[outside]).all())                 # identity outside [-B, B]
    inside = ~outside
    assert bool((y[inside].abs() <= 3.0 + 1e-4).all())            # the spline maps [-B, B] onto itself
    # monotone in every row: sorting a column block by x sorts y (checked on sampled pairs)
    i1 = torch.randint(0, N, (4096,), generator=torch.Generator().manual_seed(1)).cuda()
    i2 = torch.randint(0, N, (4096,), generator=torch.Generator().manual_seed(2)).cuda()
    dx, dy = x[:, i1] - x[:, i2], y[:, i1] - y[:, i2]
    assert bool((dx * dy >= -1e-6).all())
    idx = torch.randint(0, N, (256,), generator=torch.Generator().manual_seed(3)).cuda()
    w, h, dd = (t.cpu().numpy().astype(np.float64) for t in (b.widths, b.heights, b.derivatives))
    ys, ls = orc.rqs(w, h, dd, sample_cols(x, idx))
    np.testing.assert_allclose(y[:, idx].cpu().numpy(), ys, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(lps[idx].cpu().numpy(), ls, rtol=1e-3, atol=2e-3)


def test_c4_planar_flow_full_size(bj, orc):
    """configs[3]: 8-layer PlanarLayer flow, dim = 128, batch = 2^22."""
    d, nl, N = 128, 8, 1 << 22
    z = fill(bj, cm(d, N), 0)
    w = fill(bj, cm(d, nl), 200, std=1 / math.sqrt(d))
    u = fill(bj, cm(d, nl), 201, std=1 / math.sqrt(d))
    bb = fill(bj, cm(nl, 1), 202).reshape(-1).contiguous()
    # the flow the way the reference writes it (docs/src/flows.md:115): l8 ∘ … ∘ l1, one PlanarLayer object per layer; the
    # composition planner must hand the run to ONE fused launch (1 028 B/sample, not 8 224)
    layers = [bj.PlanarLayer(w[:, l].contiguous(), u[:, l].contiguous(), bb[l:l + 1].contiguous()) for l in range(nl)]
    flow = layers[0]
    for l in layers[1:]:
        flow = l @ flow
    bj.transform(flow, z[:, :64])
    (zf, lps, lsum), _, launches = bj.kernel_timed(lambda: bj.shard.with_logabsdet_jacobian_sharded(flow, z))
    assert launches == 1
    stacked = bj.with_logabsdet_jacobian(bj.PlanarLayer(w, u, bb), z)
    assert torch.equal(zf, stacked.result) and torch.equal(lps, stacked.logabsdetjac)      # bit-equal with the stacked constructor
    del stacked
    (zb, lps_inv, _), _, launches = bj.kernel_timed(lambda: bj.shard.with_logabsdet_jacobian_sharded(bj.inverse(flow), zf))
    assert launches == 1
    assert torch.allclose(zb, z, rtol=1e-3, atol=2e-3)            # test/normalising_flows.jl:37-42
    assert torch.allclose(lps_inv, -lps, rtol=1e-3, atol=2e-3)
    assert abs(float(lsum) - float(lps.double().sum())) <= 1e-6 * abs(float(lsum)) + 1e-3
    # the fused 8-layer launch equals 8 single-layer launches
    cur, lacc = z, torch.zeros(N, device="cuda")
    for l in range(nl):
        one = bj.PlanarLayer(w[:, l].contiguous(), u[:, l].contiguous(), bb[l:l + 1].contiguous())
        cur, lp, _ = bj.shard.with_logabsdet_jacobian_sharded(one, cur)
        lacc = lacc + lp
    assert torch.allclose(cur, zf, rtol=1e-3, atol=1e-3)
    assert torch.allclose(lacc, lps, rtol=1e-3, atol=1e-3)
    idx = torch.randint(0, N, (128,), generator=torch.Generator().manual_seed(4)).cuda()
    zs, ls = orc.planar(w.cpu().numpy().astype(np.float64), u.cpu().numpy().astype(np.float64), bb.cpu().numpy().astype(np.float64), sample_cols(z, idx))
    np.testing.assert_allclose(zf[:, idx].cpu().numpy(), zs, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(lps[idx].cpu().numpy(), ls, rtol=1e-3, atol=1e-3)


def test_c5_simplex_and_cholesky_full_size(bj, orc):
    """configs[4]: SimplexBijector K = 64, batch = 2^20 and VecCholeskyBijector K = 64."""
    K, N = 64, 1 << 20
    x = torch.softmax(fill(bj, cm(K, N), 0).T, dim=1).T
    b = bj.SimplexBijector()
    y, lps, lsum = bj.shard.with_logabsdet_jacobian_sharded(b, x)
    assert tuple(y.shape) == (K - 1, N)
    xb, lps_inv, _ = bj.shard.with_logabsdet_jacobian_sharded(bj.inverse(b), y)
    assert torch.allclose(xb, x, rtol=2e-3, atol=2e-6)
    assert torch.allclose(xb.sum(dim=0), torch.ones(N, device="cuda"), atol=1e-5)     # test/legacy_interface.jl:275-279
    assert torch.allclose(lps_inv, -lps, rtol=1e-3, atol=5e-2)
    idx = torch.randint(0, N, (256,), generator=torch.Generator().manual_seed(5)).cuda()
    ys, ls = orc.simplex(sample_cols(x, idx))
    np.testing.assert_allclose(y[:, idx].cpu().numpy(), ys, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(lps[idx].cpu().numpy(), ls, rtol=1e-3, atol=1e-2)
    del y, xb, lps, lps_inv, x
    torch.cuda.empty_cache()
    # Cholesky-correlation factor: K = 64 at the FULL batch of BASELINE configs[4], 2^20 samples (8.5 GB of y, 17 GB of dense W —
    # what bench.py's c5b row times); invariants over every sample, the oracle on 16 sampled columns
    Nc = 1 << 20
    n = K * (K - 1) // 2
    yv = fill(bj, cm(n, Nc), 1, std=0.5)
    ib = bj.inverse(bj.VecCholeskyBijector("U"))
    W, lj, ljsum = bj.shard.with_logabsdet_jacobian_sharded(ib, yv)
    Wm = W.permute(2, 1, 0)                                        # (Nc, col, row): contiguous samples
    for lo in range(0, Nc, 1 << 17):                               # in slabs: the temporaries stay at 2 GB
        sl = Wm[lo:lo + (1 << 17)]
        assert torch.allclose((sl * sl).sum(dim=2), torch.ones(sl.shape[0], K, device="cuda"), atol=2e-5)   # unit columns of a correlation factor
        assert bool((torch.triu(sl, diagonal=1) == 0).all())       # strictly lower part of every W[:, :, n] is zero (corr.jl:391-395); sl[n] = W[:, :, n]'
        del sl
    # logabsdetjac(inverse(b), y) alone (corr.jl:252-254, no W written) equals the fused value
    assert abs(float(bj.logabsdetjac(ib, yv)) - float(ljsum)) <= 1e-4 * abs(float(ljsum))
    yb, lf, _ = bj.shard.with_logabsdet_jacobian_sharded(bj.VecCholeskyBijector("U"), W)
    for lo in range(0, Nc, 1 << 18):
        assert torch.allclose(yb[:, lo:lo + (1 << 18)], yv[:, lo:lo + (1 << 18)], rtol=2e-3, atol=2e-4)      # test/bijectors/corr.jl:46-64 roundtrip
    assert torch.allclose(lf, -lj, rtol=1e-3, atol=5e-2)
    idx = torch.randint(0, Nc, (16,), generator=torch.Generator().manual_seed(6)).cuda()
    Ws, ljs = orc.vec_cholesky(sample_cols(yv, idx), inverse=True, uplo="U")
    np.testing.assert_allclose(W[:, :, idx].cpu().numpy(), Ws, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(lj[idx].cpu().numpy(), ljs, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("mode", ["sum", "both"])
def test_shard_invariance_with_a_vector_scale(bj, mode):
    """A per-row vector Scale in a sharded chain (bench.py's c2v): the all-reduced Σ logabsdetjac must not depend on
    the shard count and must equal sum(ladj_ps).  The reference's un-multiplied Σ log|a_i| (scale.jl:31-32) is only the
    value of the UNSHARDED scalar return (`with_logabsdet_jacobian`), never of the partial sums that get all-reduced."""
    d, N = 64, (1 << 18) + 37
    x = fill(bj, cm(d, N), 3)
    a = torch.linspace(0.5, 1.5, d, device="cuda")
    b = bj.elementwise(bj.exp) @ bj.Shift(torch.full((d,), 0.1, device="cuda")) @ bj.Scale(a)
    per_sample = mode == "both"
    _, lps, whole = bj.shard.with_logabsdet_jacobian_sharded(b, x, per_sample=per_sample)
    ref = (x.double() * a.double()[:, None] + 0.1).sum() + N * a.double().log().sum()       # closed form, Float64
    assert abs(float(whole) - float(ref)) <= 1e-6 * abs(float(ref))
    if per_sample:
        assert abs(float(whole) - float(lps.double().sum())) <= 1e-6 * abs(float(whole))
    for G in (2, 8):
        parts = sum(float(bj.shard.with_logabsdet_jacobian_sharded(b, x[:, lo:hi], per_sample=per_sample)[2])
                    for lo, hi in (bj.shard.shard_columns(N, G, r) for r in range(G)))
        assert abs(parts - float(whole)) <= 1e-9 * abs(float(whole)), (G, parts, float(whole))
    # the unsharded reference-shaped scalar keeps the reference's value: data terms + Σ log|a_i| ONCE
    _, l_ref_shape = bj.with_logabsdet_jacobian(b, x)
    want = (x.double() * a.double()[:, None] + 0.1).sum() + a.double().log().sum()
    assert abs(float(l_ref_shape) - float(want)) <= 2e-6 * abs(float(want)) + 1e-2
