"""Regression tests for the advisor's round-3 findings, on the GPU through the library."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from test_gpu_parity import bj, close, dev, host, rng  # noqa: E402,F401


@pytest.mark.parametrize("dt,dim", [(np.float32, 261), (np.float32, 262), (np.float32, 263), (np.float32, 260), (np.float32, 300),
                                    (np.float64, 131), (np.float64, 130), (np.float64, 150)])
def test_row_permuting_stacked_beyond_one_slab(bj, dt, dim):
    """A `Stacked` whose segments MOVE rows (ranges_in not in output order: in_lo != out_lo) on columns a little taller than one
    row slab of 64 packs: the last slab is shorter than two packs at Float32 261-263 / Float64 131 and used to resolve to scalar
    packs while the functor's table stayed permuted for 16-byte packs (bjx_stream.h launch_colgroup) — rows 256.. came out wrong."""
    r = rng(7)
    N = 97
    x = np.asfortranarray(np.abs(r.normal(size=(dim, N))).astype(dt) + dt(0.1))
    cut = 2
    b = bj.Stacked([bj.elementwise(bj.exp) @ bj.Scale(0.5), bj.elementwise(bj.log)], [(cut + 1, dim), (1, cut)])
    y, l = bj.with_logabsdet_jacobian(b, dev(x), per_sample=True)
    x64 = x.astype(np.float64)
    y_ref = np.concatenate([np.exp(0.5 * x64[cut:]), np.log(x64[:cut])], axis=0)
    l_ref = (0.5 * x64[cut:] + np.log(0.5)).sum(axis=0) - np.log(x64[:cut]).sum(axis=0)
    close(host(y), y_ref, dt, what=f"row-permuting Stacked dim={dim}")
    close(host(l), l_ref, dt, scale=dim, what="ladj")
    # and the inverse moves them back
    xb = bj.transform(bj.inverse(b), y)
    close(host(xb), x64, dt, scale=10, what="inverse")


def test_batchnorm_training_pullback_never_uses_another_batchs_statistics(bj):
    """Two training-mode forward calls before one backward: the saved statistics belong to the second batch.  Round 3 refused the
    pullback of the first; since round 5 (storage identity is no batch identity) the pullback recomputes the batch
    statistics from the x it is given whenever that x is not the very tensor, unwritten, of the last forward call — so it is
    always the pullback of the training-mode map AT x."""
    r = rng(8)
    d, N = 5, 64
    x1h, x2h = (np.asfortranarray(r.normal(size=(d, N)).astype(np.float32)) for _ in range(2))
    gh = np.asfortranarray(r.normal(size=(d, N)).astype(np.float32))

    def fresh(xh):
        """pullback right after the forward call on the same tensor object (the saved-statistics path)"""
        bn_ = bj.InvertibleBatchNorm(d)
        x_ = dev(xh)
        with bj.training():
            bj.with_logabsdet_jacobian(bn_, x_)
            xb_, gr_ = bj.vjp_params(bn_, x_, dev(gh))
        return host(xb_), host(gr_["b"]), host(gr_["logs"])

    bn = bj.InvertibleBatchNorm(d)
    x1, x2, g = dev(x1h), dev(x2h), dev(gh)
    with bj.training():
        bj.with_logabsdet_jacobian(bn, x1)
        bj.with_logabsdet_jacobian(bn, x2)                  # another batch in between
        xb, gr = bj.vjp_params(bn, x1, g)                   # must be the pullback at x1, not x1 with x2's statistics
        ref = fresh(x1h)
        np.testing.assert_allclose(host(xb), ref[0], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(host(gr["logs"]), ref[2], rtol=2e-4, atol=2e-4)
        x2.mul_(2.0)                                         # written in place after its forward call
        xb2, gr2 = bj.vjp_params(bn, x2, g)
        ref2 = fresh(2.0 * x2h)
        np.testing.assert_allclose(host(xb2), ref2[0], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(host(gr2["logs"]), ref2[2], rtol=2e-4, atol=2e-4)
        # a different tensor OBJECT with the same values (what `dev(x)` twice gives; the allocator may or may not recycle the address)
        bj.with_logabsdet_jacobian(bn, dev(x1h))
        xb3, _ = bj.vjp_params(bn, dev(x1h), g)
        np.testing.assert_allclose(host(xb3), ref[0], rtol=2e-4, atol=2e-5)


def test_general_size_matrix_paths_refuse_sizes_that_would_run_for_minutes(bj):
    a = torch.eye(1025, dtype=torch.float32, device="cuda")
    x = torch.ones(1025, 2, dtype=torch.float32, device="cuda").T.contiguous().T
    with pytest.raises(NotImplementedError, match="1024"):
        bj.with_logabsdet_jacobian(bj.Scale(a), bj.colmajor(x))
