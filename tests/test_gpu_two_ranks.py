"""The library with MORE THAN ONE RANK: two processes, one per shard of the batch, both on cuda:0 (the gpurun
boxes have one GPU), rendezvous over gloo at 127.0.0.1.  Every rank runs its column block through libbjx_hip.so — the product
path, not the oracle — and the test checks, against the SAME library run on the whole batch by one rank:

  * outputs of a shard are bit-identical to the same columns of the whole-batch call (the kernels are column-local);
  * the all-reduced Float64 Σ logabsdetjac equals the single-rank sum (to Float64 rounding: the partials are added in a
    different order) and is the same number on both ranks;
  * training-mode InvertibleBatchNorm (bjx_batchnorm_stats -> all-reduce of 2·dim+1 doubles -> bjx_batchnorm_train_apply, opted
    into with sync=True) normalises with the GLOBAL batch statistics and leaves every rank with the same moving statistics;
  * vjp_params_sharded returns the whole-batch parameter cotangents on every rank.

(The library's own RCCL communicator, bjx_comm_init, needs one GPU per rank: RCCL refuses two ranks on one device.  It is
covered with one rank in test_gpu_parity.py::test_rccl_communicator_single_rank and by the driver's multi-GPU bench.)
"""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return torch.from_numpy(np.ascontiguousarray(a.T)).cuda().T


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist

        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import bijectors_amd as bj

        bj._lib.load()
        sh = bj.shard
        r = np.random.default_rng(11)
        rep = {}

        def shard_of(a, N):
            lo, hi = sh.shard_columns(N, world, rank)
            return np.asfortranarray(a[:, lo:hi]), lo, hi

        def check(tag, b, x, f64_rtol=1e-11, dt=np.float32):
            """b on this rank's columns + the collective, against b on the whole batch (same library, this process)."""
            N = x.shape[1]
            xs, lo, hi = shard_of(x, N)
            y_s, lps_s, lsum = sh.with_logabsdet_jacobian_sharded(b, _dev(xs))
            res = bj.with_logabsdet_jacobian(b, _dev(x))
            y_full, l_full = (res.result, res.logabsdetjac) if hasattr(res, "result") else (res[0], res[1])
            ok_y = np.array_equal(y_s.cpu().numpy(), y_full.cpu().numpy()[:, lo:hi])
            per_col = isinstance(l_full, torch.Tensor) and l_full.numel() == N
            total = float(l_full.double().sum()) if isinstance(l_full, torch.Tensor) else float(l_full)
            tol = (2e-7 if dt == np.float32 else f64_rtol) * (abs(total) + np.sqrt(N))     # the reference scalar itself is Float32-rounded for f32
            ok_l = abs(float(lsum[0]) - total) <= tol
            ok_ps = (not per_col) or np.array_equal(lps_s.cpu().numpy(), l_full.cpu().numpy()[lo:hi])
            rep[tag] = (bool(ok_y), bool(ok_l), bool(ok_ps), float(lsum[0]))

        # (1) the headline chain on a C2-shaped batch that does not split evenly
        N = (1 << 16) + 37
        x = np.asfortranarray(r.normal(size=(64, N)).astype(np.float32))
        check("chain", bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5), x)
        # (2) Simplex (scalar log-det summed over the columns)
        xs = np.asfortranarray(r.dirichlet(np.ones(16), size=4099).T)
        check("simplex", bj.SimplexBijector(), xs, dt=np.float64)
        # (3) a 3-layer PlanarLayer stack (per-column log-dets)
        dim, nl = 32, 3
        w, u, b = r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=nl)
        flow = bj.PlanarLayer(torch.tensor(w, dtype=torch.float32), torch.tensor(u, dtype=torch.float32), torch.tensor(b, dtype=torch.float32))
        z = np.asfortranarray(r.normal(size=(dim, 5001)).astype(np.float32))
        check("planar", flow, z)

        # (4) training-mode InvertibleBatchNorm over the sharded batch vs one rank holding the whole batch
        dimb, Nb = 24, 3001
        xb = np.asfortranarray((r.normal(size=(dimb, Nb)) * 2.0 + 3.0))
        pb, plogs = r.normal(size=dimb), r.normal(size=dimb) * 0.1
        mk = lambda sync: bj.InvertibleBatchNorm(torch.tensor(pb), torch.tensor(plogs), torch.zeros(dimb, dtype=torch.float64),
                                                 torch.ones(dimb, dtype=torch.float64), eps=1e-5, mtm=0.1, sync=sync)
        bn_s, bn_f = mk(True), mk(None)
        xs_, lo, hi = shard_of(xb, Nb)
        with bj.training():
            ys, ls = bj.with_logabsdet_jacobian(bn_s, _dev(xs_))          # both ranks in lockstep: one all-reduce inside
            yf, lf = bj.with_logabsdet_jacobian(bn_f, _dev(xb))           # no collective: this process's whole batch
        ok = np.allclose(ys.cpu().numpy(), yf.cpu().numpy()[:, lo:hi], rtol=1e-12, atol=1e-12)
        ok = ok and np.allclose(ls.cpu().numpy(), lf.cpu().numpy()[lo:hi], rtol=1e-12, atol=1e-12)
        ok = ok and np.allclose(bn_s.m.cpu().numpy(), bn_f.m.cpu().numpy(), rtol=1e-12, atol=1e-13)
        ok = ok and np.allclose(bn_s.v.cpu().numpy(), bn_f.v.cpu().numpy(), rtol=1e-12, atol=1e-13)
        rep["batchnorm_train"] = (bool(ok), True, True, float(bn_s.m.double().sum()))

        # (5) parameter cotangents of the PlanarLayer stack: sharded + all-reduced vs whole batch
        g = np.asfortranarray(r.normal(size=z.shape).astype(np.float32))
        lb = r.normal(size=z.shape[1]).astype(np.float32)
        zs, lo, hi = shard_of(z, z.shape[1])
        gs = np.asfortranarray(g[:, lo:hi])
        xbar_s, grads_s = sh.vjp_params_sharded(flow, _dev(zs), _dev(gs), _dev(lb[lo:hi]))
        xbar_f, grads_f = bj.vjp_params(flow, _dev(z), _dev(g), _dev(lb))
        ok = np.array_equal(xbar_s.cpu().numpy(), xbar_f.cpu().numpy()[:, lo:hi])
        for k in ("w", "u", "b"):
            a_, b_ = grads_s[k].double().cpu().numpy(), grads_f[k].double().cpu().numpy()
            ok = ok and np.allclose(a_, b_, rtol=2e-4, atol=2e-4 * np.abs(b_).max())
        rep["vjp_params"] = (bool(ok), True, True, float(grads_s["w"].double().sum()))

        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, rep, None))
    except Exception as e:  # pragma: no cover - reported to the parent
        import traceback

        q.put((rank, {}, traceback.format_exc() + repr(e)))


def test_two_ranks_one_gpu_through_the_library():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for rank, rep, err in res:
        assert err is None, f"rank {rank}: {err}"
        assert set(rep) == {"chain", "simplex", "planar", "batchnorm_train", "vjp_params"}
        for tag, (ok_y, ok_l, ok_ps, _) in rep.items():
            assert ok_y and ok_l and ok_ps, f"rank {rank}: {tag}: outputs {ok_y}, Σlogabsdetjac {ok_l}, per-column {ok_ps}"
    # the global scalars are the same number on both ranks
    for tag in res[0][1]:
        assert res[0][1][tag][3] == res[1][1][tag][3], tag


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (VERDICT r03 missing #4): bench.py starts the two ranks itself (here both on the
    one GPU of the box, over gloo) and prints a 2-rank line — never a silent 1-GPU measurement."""
    import json
    import subprocess

    env = dict(os.environ, BJX_BENCH_BACKEND="gloo", BJX_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-rows", "--log2-batch", "18"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["batch_per_gpu"] == 1 << 18
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None          # the CPU leg runs on rank 0 at N = 1 only
