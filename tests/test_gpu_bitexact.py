"""Bit-level checks of the "ordering" bijectors (north_star: bit-exact for permutation/ordering bijectors).

OrderedBijector (ordered.jl:36-49) accumulates x_i = x_{i-1} + exp(y_i) in ascending i inside every column; the
Simplex inverse (simplex.jl:102-120) accumulates the running sum of its CLAMPED outputs in ascending k and ends with
x_K = clamp(1 - s, 0, 1).  Bit-level results of exp itself are parity-unpinned (the reference compares with `≈`), so
the checks take the DEVICE's own exp (the chain kernel evaluates the same function) and redo only the summation on
the host, in the reference's order and in the data type of the call, and require bit equality.  K covers every
kernel family: the register-streaming kernels (K = 8, 64), the one-lane-per-column LDS kernel (K = 100, 7) and the
chunked kernel for long columns.
"""
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a.T)).cuda().T


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(8, 4099), (64, 1000), (100, 257), (7, 333), (16, 64), (32, 4096), (1500, 9)])
def test_ordered_sum_order_is_bit_exact(bj, shape, dt):
    K, N = shape
    y = np.random.default_rng(K * 7 + N).normal(size=(K, N)).astype(dt)
    yd = dev(y)
    x = host(bj.transform(bj.OrderedBijector(), yd))
    e = host(bj.transform(bj.elementwise(bj.exp), yd))          # the device's exp, rounded to dt
    ref = np.empty_like(x)
    ref[0] = y[0]
    for i in range(1, K):                                       # ordered.jl:42-45, ascending i, arithmetic in dt
        ref[i] = (ref[i - 1] + e[i]).astype(dt)
    assert x.dtype == dt
    assert np.array_equal(x, ref), f"{np.count_nonzero(x != ref)} of {x.size} elements differ from the ascending-order sum"
    assert np.all(np.diff(x, axis=0) >= 0)                      # test/bijectors/ordered.jl:25-38: sort(b(x)) == b(x)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(8, 4099), (64, 1000), (100, 257), (7, 333), (32, 4096), (1000, 11)])
def test_simplex_running_sum_is_bit_exact(bj, shape, dt):
    K, N = shape
    y = np.random.default_rng(K * 3 + N).normal(size=(K - 1, N)).astype(dt)
    x = host(bj.transform(bj.inverse(bj.SimplexBijector()), dev(y)))
    assert x.dtype == dt and x.shape == (K, N)
    s = x[0].copy()                                             # simplex.jl:110 sum_tmp = x[1]
    for k in range(1, K - 1):                                   # :111-116, ascending k, arithmetic in dt
        s = (s + x[k]).astype(dt)
    last = np.clip((dt(1) - s).astype(dt), dt(0), dt(1))        # :118
    assert np.array_equal(x[K - 1], last), f"{np.count_nonzero(x[K - 1] != last)} of {N} last rows differ"
