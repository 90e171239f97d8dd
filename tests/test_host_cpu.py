"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/bjx.h
declares, the host mirror's structural logic (chain walking, inverse mapping, output sizes,
constructors, error behaviour), and the N > 1 sharding path over `gloo` with world_size 2."""
import math
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bj():
    import bijectors_amd

    return bijectors_amd


def test_library_exports_every_declared_symbol(bj):
    hdr = open(os.path.join(ROOT, "include", "bjx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bjx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = bj._lib.load()      # raises if libbjx_hip.so is missing or incomplete (no fallback)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libbjx_hip.so does not export {name}"
    assert declared == set(bj._lib.SIGNATURES), declared ^ set(bj._lib.SIGNATURES)
    assert lib.bjx_version() == 100


def test_op_struct_layout_matches_header(bj):
    import ctypes as C

    assert C.sizeof(bj._lib.BjxOp) == 40        # int32,int32,double,double,ptr,ptr
    assert bj._lib.BjxOp.p0.offset == 8 and bj._lib.BjxOp.v0.offset == 24


def test_segment_struct_layout_and_stacked_ranges(bj):
    """bjx_segment (include/bjx.h) = 3 x int64 + 2 x int32 + 4 bjx_op; Stacked's output ranges are the
    cumulative ones of stacked.jl:50-57 and inverse swaps them (:113-118) — host logic, no GPU."""
    import ctypes as C

    assert C.sizeof(bj._lib.BjxSegment) == 24 + 8 + 4 * 40
    assert bj._lib.BjxSegment.ops.offset == 32
    b = bj.Stacked([bj.SimplexBijector(), bj.elementwise(bj.exp), bj.identity], [(1, 5), (6, 7), (8, 8)])
    assert b.ranges_out == [(1, 4), (5, 6), (7, 7)] and (b.length_in, b.length_out) == (8, 7)
    assert bj.output_size(b, (8,)) == (7,) and bj.output_size(b, (8, 3)) == (7, 3)
    ib = bj.inverse(b)
    assert ib.ranges_in == b.ranges_out and ib.ranges_out == b.ranges_in and (ib.length_in, ib.length_out) == (7, 8)
    with pytest.raises(ValueError):
        bj.Stacked([bj.identity, bj.identity], [(1, 2)])
    cw = bj.columnwise(bj.OrderedBijector())
    assert isinstance(bj.inverse(cw), bj.Columnwise) and isinstance(bj.inverse(cw).x, bj.Inverse)


def test_no_cpu_fallback(bj):
    import torch

    with pytest.raises(RuntimeError):
        bj.with_logabsdet_jacobian(bj.elementwise(bj.exp), torch.ones(3))
    with pytest.raises(RuntimeError):
        bj.transform(bj.SimplexBijector(), torch.ones(3, 2))


def test_chain_walk_and_inverse(bj):
    L = bj._lib
    b = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)      # exp ∘ Shift ∘ Scale (left-assoc)
    assert [type(s).__name__ for s in b._stages()] == ["Scale", "Shift", "Elementwise"]
    from bijectors_amd import interface as I

    assert [o[0] for o in I._fused_ops(b)] == [L.OP_SCALE, L.OP_SHIFT, L.OP_EXP]
    ib = bj.inverse(b)                                                # inverse(f∘g) = inverse(g) ∘ inverse(f)
    ops = I._fused_ops(ib)
    assert [o[0] for o in ops] == [L.OP_LOG, L.OP_SHIFT, L.OP_SCALE_INV]
    assert ops[1][1] == -0.1                                          # shift.jl:12
    assert bj.inverse(ib)._stages()[0] == bj.Scale(0.5)
    assert bj.inverse(bj.inverse(bj.Logit(0.0, 1.0))) == bj.Logit(0.0, 1.0)      # interface.jl:266
    assert bj.inverse(bj.LeakyReLU(0.25)) == bj.LeakyReLU(4.0)                   # leaky_relu.jl:16
    assert I._fused_ops(bj.inverse(bj.TruncatedBijector(0.0, 1.0)))[0][0] == L.OP_TRUNCATED_INV
    # a structured bijector breaks the fusable run
    mixed = bj.elementwise(bj.exp) @ bj.OrderedBijector() @ bj.Shift(1.0)
    assert I._fused_ops(mixed) is None
    assert bj.isinvertible(mixed) and bj.isclosedform(mixed)
    assert not bj.isclosedform(bj.inverse(bj.PlanarLayer([1.0, 2.0], [0.5, 0.1], [0.0])))   # planar_layer.jl:188
    # elementwise of a composition distributes (interface.jl:37-39)
    assert bj.elementwise(bj.identity) is bj.identity
    # `identity` is its own bijector: the default of transformed(d) (transformed_distribution.jl:20-28), its own inverse, one no-op stage
    assert bj.inverse(bj.identity) is bj.identity
    assert [o[0] for o in I._fused_ops(bj.identity)] == [L.OP_IDENTITY]
    td = bj.transformed(bj.MvNormal(3))
    assert td.transform is bj.identity
    # the inverse of a flow layer keeps the layer: its parameter pullback is the forward one at the pre-image (implicit function theorem)
    pl = bj.PlanarLayer([1.0, 2.0], [0.5, 0.1], [0.0])
    assert bj.inverse(pl).orig is pl and bj.inverse(bj.inverse(pl)) is pl


def test_composition_is_cut_into_pieces_with_a_device_pullback(bj):
    """interface._pieces: runs of elementwise stages of at most BJX_MAX_SEG_OPS ops become one fused piece each, every other stage
    stands alone, application order is kept (the chain rule of _vjp_composed walks these pieces backwards)."""
    from bijectors_amd import interface as I
    L = bj._lib
    chain = bj.elementwise(bj.exp) @ bj.Shift(1.0) @ bj.Scale(2.0) @ bj.LeakyReLU(0.3) @ bj.Shift(0.2) @ bj.Scale(1.7)
    pcs = I._pieces(chain)
    assert [len(I._fused_ops(p_)) for p_ in pcs] == [L.BJX_MAX_SEG_OPS, 2]
    assert [o[0] for p_ in pcs for o in I._fused_ops(p_)] == [L.OP_SCALE, L.OP_SHIFT, L.OP_LEAKY_RELU, L.OP_SCALE, L.OP_SHIFT, L.OP_EXP]
    flow = bj.elementwise(bj.exp) @ bj.OrderedBijector() @ bj.Shift(0.5) @ bj.Scale(2.0) @ bj.PlanarLayer([1.0, 2.0], [0.5, 0.1], [0.0])
    names = [type(p_).__name__ for p_ in I._pieces(flow)]
    assert names == ["PlanarLayer", "ComposedFunction", "OrderedBijector", "Elementwise"]
    assert I._has_own_params(flow._stages()[0]) and I._has_own_params(bj.inverse(flow._stages()[0])) and not I._has_own_params(bj.OrderedBijector())


def test_output_size(bj):
    assert bj.output_size(bj.SimplexBijector(), (5,)) == (4,)                       # simplex.jl:6-12
    assert bj.output_size(bj.inverse(bj.SimplexBijector()), (4, 7)) == (5, 7)
    assert bj.output_size(bj.VecCholeskyBijector("U"), (4, 4)) == (6,)              # corr.jl:256-259
    assert bj.output_size(bj.inverse(bj.VecCholeskyBijector("L")), (6,)) == (4, 4)
    assert bj.output_size(bj.elementwise(bj.exp) @ bj.SimplexBijector(), (5, 2)) == (4, 2)
    assert bj.output_size(bj.Shift(1.0), (3, 9)) == (3, 9)
    from bijectors_amd import interface as I

    for n in range(0, 200):
        K = I._triu1_dim_from_length(n * (n - 1) // 2) if n >= 2 else 1
        assert n < 2 or K == n                                                      # src/utils.jl:99


def test_constructors_and_errors(bj):
    with pytest.raises(ValueError):
        bj.VecCholeskyBijector("X")                                                 # corr.jl:215-219
    assert bj.VecCholeskyBijector(":L").mode == "L"
    # Permute constructors agree (test/bijectors/permute.jl:13-36)
    b1 = bj.Permute([[0, 1, 0], [1, 0, 0], [0, 0, 1]])
    assert b1 == bj.Permute([2, 1, 3]) == bj.Permute(3, (2, 1), (1, 2)) == bj.Permute(3, ([1, 2], [2, 1]))
    assert bj.inverse(bj.Permute([2, 3, 1])).src == bj.Permute([3, 1, 2]).src
    with pytest.raises(ValueError):
        bj.Permute(2, (2, 1))
    # PartitionMask (coupling.jl:83-113)
    m = bj.PartitionMask(3, [1], [2])
    assert (m.indices_1, m.indices_2, m.indices_3) == ([1], [2], [3])
    m = bj.PartitionMask(5, [2, 4])
    assert (m.indices_2, m.indices_3) == ([1, 3, 5], [])
    m = bj.PartitionMask(4, [1], None, [4])
    assert m.indices_2 == [2, 3]
    import torch

    with pytest.raises(AssertionError):
        bj.RationalQuadraticSpline(torch.zeros(2, 3), torch.zeros(2, 3), -torch.ones(2, 3))   # derivatives > 0 (rqs.jl:94)
    assert bj.Scale(torch.eye(3)).matrix and not bj.Scale(torch.ones(3)).matrix       # scale.jl:14: a matrix `a` means a * x
    with pytest.raises(ValueError):
        bj.Scale(torch.zeros(3, 4))                                                      # logabsdet needs a square matrix
    with pytest.raises(ValueError):
        bj.Inverse(object())


def test_shard_columns(bj):
    for N in (0, 1, 7, 1 << 20):
        for G in (1, 2, 3, 8):
            cover = [bj.shard.shard_columns(N, G, r) for r in range(G)]
            assert cover[0][0] == 0 and cover[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    with pytest.raises(ValueError):
        bj.shard.shard_columns(10, 2, 2)


# ------------------------------------------------------------------ world_size 2 over gloo
def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import bijectors_amd as bj
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, dim = 1000, 16
    x = np.asfortranarray(np.random.default_rng(0).normal(size=(dim, N)))
    ops = [(orc.OP_SCALE, 0.5, None), (orc.OP_SHIFT, 0.1, None), (orc.OP_EXP, None, None)]
    lo, hi = bj.shard.shard_columns(N, world, rank)
    # the CPU oracle stands in for this rank's local hot-path call (there is no GPU here); the
    # code under test is the shard bookkeeping + the single collective
    _, part = orc.chain(ops, np.asfortranarray(x[:, lo:hi]))
    t = torch.tensor([float(part)], dtype=torch.float64)
    bj.shard.allreduce_logabsdetjac(t)
    _, full = orc.chain(ops, x)
    ok = abs(float(t[0]) - float(full)) <= 1e-9 * abs(float(full))
    try:
        bj.shard.allreduce_logabsdetjac(torch.zeros(1, dtype=torch.float32))
        ok = False
    except TypeError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, float(t[0])))


def test_allreduce_logabsdetjac_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]      # every rank holds the same global scalar


def _worker_params(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import bijectors_amd as bj
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the CPU oracle stands in for this rank's vjp_params call: cotangents of a 2-layer PlanarLayer stack on the rank's columns
    r = np.random.default_rng(5)
    dim, nl, N = 6, 2, 101
    w, u, b = r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=nl)
    Z, G, lb = r.normal(size=(dim, N)), r.normal(size=(dim, N)), r.normal(size=N)
    lo, hi = bj.shard.shard_columns(N, world, rank)
    wb, ub, bb = orc.planar_param_vjp(w, u, b, Z[:, lo:hi], G[:, lo:hi], lb[lo:hi])
    grads = {"stages": [None, {"w": torch.tensor(wb, dtype=torch.float32), "u": torch.tensor(ub), "b": torch.tensor(bb)}, {"shift": torch.tensor(float(rank + 1))}]}
    out = bj.shard.allreduce_param_cotangents(grads)
    wf, uf, bf = orc.planar_param_vjp(w, u, b, Z, G, lb)
    st = out["stages"][1]
    ok = out is grads and st["w"].dtype == torch.float32 and st["u"].dtype == torch.float64
    ok = ok and np.allclose(st["w"].numpy(), wf, rtol=1e-5, atol=1e-6) and np.allclose(st["u"].numpy(), uf, rtol=1e-12, atol=1e-12)
    ok = ok and np.allclose(st["b"].numpy(), bf, rtol=1e-12, atol=1e-12) and float(out["stages"][2]["shift"]) == sum(range(1, world + 1))
    try:
        bj.shard.allreduce_param_cotangents({"a": 3.0})
        ok = False
    except TypeError:
        pass
    # InvertibleBatchNorm in a multi-process job: sync unspecified = per-rank statistics WITH a warning (once);
    # sync=False the same on purpose, silently; sync=True the default group; a sharded training step refuses sync=None
    import warnings

    bj.InvertibleBatchNorm._warned_unsynced[0] = False
    with warnings.catch_warnings(record=True) as wrec:
        warnings.simplefilter("always")
        g0 = bj.InvertibleBatchNorm(3)._sync_group()
        g0b = bj.InvertibleBatchNorm(3)._sync_group()
    ok = ok and g0 is False and g0b is False and len([w_ for w_ in wrec if issubclass(w_.category, RuntimeWarning)]) == 1
    with warnings.catch_warnings(record=True) as wrec:
        warnings.simplefilter("always")
        ok = ok and bj.InvertibleBatchNorm(3, sync=False)._sync_group() is False and bj.InvertibleBatchNorm(3, sync=True)._sync_group() is None
    ok = ok and not wrec
    with bj.training():
        try:
            bj.shard.vjp_params_sharded(bj.InvertibleBatchNorm(3) @ bj.Shift(1.0), None, None)
            ok = False
        except ValueError as e:
            ok = ok and "sync" in str(e)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok), st["u"].numpy().tobytes()))


def test_allreduce_param_cotangents_gloo_world2():
    """Data-parallel training step on two ranks: per-rank parameter cotangents (nested dictionaries, mixed dtypes, None holes) are
    summed in ONE float64 bucket and equal the single-process cotangents of the whole batch."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_params, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), [(r_, ok) for r_, ok, _ in res]
    assert res[0][2] == res[1][2]      # bit-identical on every rank


def test_vector_links_and_density_op_lists(bj):
    """Host logic of the §8(f) wrappers, no GPU: the VectorBijectors scalar links are op lists of the chain kernel
    (src/vector/univariate/positive.jl:11-50, truncated.jl:17-103), heterogeneous products are Stacked ranges, and a
    diagonal-normal base whitens with SHIFT(-μ), SCALE_INV(σ) in front of the density op."""
    import torch

    V, L = bj.vector, bj._lib
    assert V.Log(0.0, 1)._ops() == [(L.OP_LOG, None, None)]
    assert V.Log(-1.5, -1)._ops() == [(L.OP_SHIFT, 1.5, None), (L.OP_SIGNFLIP, None, None), (L.OP_LOG, None, None)]
    assert V.Exp(2.0, -1)._ops() == [(L.OP_EXP, None, None), (L.OP_SIGNFLIP, None, None), (L.OP_SHIFT, 2.0, None)]
    assert V.Exp(2.0, -1)._ops(True) == V.Log(2.0, -1)._ops()
    assert V.Untruncate(0.0, 1.0)._ops() == [(L.OP_TRUNCATED, 0.0, 1.0)] and V.Truncate(0.0, 1.0)._ops() == [(L.OP_TRUNCATED_INV, 0.0, 1.0)]
    assert isinstance(V.scalar_to_scalar_bijector(-math.inf, math.inf), V.TypedIdentity)
    assert isinstance(V.scalar_to_scalar_bijector(0.0, math.inf, positive_family=True), V.Log)
    assert isinstance(V.scalar_to_scalar_bijector(0.0, 1.0), V.Untruncate)
    st = V.to_linked_vec_product([(V.TypedIdentity(), 1), (V.Log(0.0, 1), 3), (V.Untruncate(0.0, 1.0), 4)])
    assert isinstance(st, bj.Stacked) and st.ranges_in == [(1, 1), (2, 4), (5, 8)]
    inv = V.from_linked_vec_product([(V.Log(0.0, 1), 2)])
    assert isinstance(inv.bs[0], V.Exp)
    t = V.to_linked_vec(bj.SimplexBijector(), (7,), base_size=(5,))
    assert t._n_components() == 7 and t.base_size == (5,)
    d = bj.MvNormal(torch.tensor([1.0, 2.0]), torch.tensor([0.5, 2.0]))
    w = d._whiten_ops()
    assert [k for k, _, _ in w] == [L.OP_SHIFT, L.OP_SCALE_INV] and torch.equal(w[0][1], torch.tensor([-1.0, -2.0]))
    assert [k for k, _, _ in d._color_ops()] == [L.OP_SCALE, L.OP_SHIFT]
    assert bj.MvNormal(3)._whiten_ops() == [] and bj.MvNormal(3).dim == 3
    assert L.OP_STDNORMAL_LOGPDF == 13 and L.BJX_BASE_STDNORMAL == 4 and L.BJX_INPUT_STDNORMAL == 8


def test_julia_binding_argument_counts_match_the_header():
    """julia/BijectorsBJX.jl (the binding INTEGRATION.md shows) must call every entry point with the number of
    arguments include/bjx.h declares."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = open(os.path.join(root, "include", "bjx.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"//.*", "", h)
    protos = {}
    for m in re.finditer(r"\b(?:int|void|const char\s*\*|size_t|uint64_t)\s+(bjx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        protos[m.group(1)] = len(args)
    assert len(protos) >= 43
    j = open(os.path.join(root, "julia", "BijectorsBJX.jl")).read()
    calls = list(re.finditer(r"ccall\(\(:(bjx_[a-z0-9_]+),\s*\w+\),\s*\w+,\s*\(([^)]*)\)", j))
    assert len(calls) >= 10
    for m in calls:
        name, types = m.group(1), [t.strip() for t in m.group(2).split(",") if t.strip()]
        assert name in protos, f"{name} is not declared in include/bjx.h"
        assert protos[name] == len(types), f"{name}: header has {protos[name]} arguments, the Julia ccall passes {len(types)}"


def test_composition_planner_merges_planar_runs(bj):
    """Host logic of the planner (no GPU): maximal runs of PlanarLayer stages become one `_PlanarRun`, inverse runs the inverse of
    the REVERSED run; other stages are untouched and the spans map the planned stages back onto `_stages()`."""
    import torch

    I = bj.interface
    ls = [bj.PlanarLayer(torch.full((3,), float(k)), torch.ones(3), torch.zeros(1)) for k in range(5)]
    rad = bj.RadialLayer(torch.zeros(1), torch.zeros(1), torch.zeros(3))
    flow = ls[4] @ ls[3] @ bj.Shift(1.0) @ ls[2] @ ls[1] @ ls[0] @ rad                  # application order: rad, l0, l1, l2, Shift, l3, l4
    st, spans = flow._plan()
    assert [type(x).__name__ for x in st] == ["RadialLayer", "_PlanarRun", "Shift", "_PlanarRun"] and spans == [(0, 1), (1, 4), (4, 5), (5, 7)]
    assert st[1].layers == ls[0:3] and st[3].layers == ls[3:5] and st[1].n_layers == 3
    assert flow._plan() is flow._plan()                                                   # cached while the stage objects are the same
    inv = bj.inverse(flow)                                                                # inv(l4), inv(l3), Shift(-1), inv(l2), inv(l1), inv(l0), inv(rad)
    sti, spi = inv._plan()
    assert [type(x).__name__ for x in sti] == ["Inverse", "Shift", "Inverse", "Inverse"] and spi == [(0, 2), (2, 3), (3, 6), (6, 7)]
    assert isinstance(sti[0].orig, I._PlanarRun) and sti[0].orig.layers == [ls[3], ls[4]]  # the forward flow of the run applies l3 then l4
    assert sti[2].orig.layers == ls[0:3] and isinstance(sti[3].orig, bj.RadialLayer)
    single = ls[1] @ bj.Shift(0.5)
    assert single._plan()[0][1] is ls[1]                                                  # a lone layer stays itself
    run = bj.PlanarLayer.stack(ls[:2])
    assert isinstance(run, I._PlanarRun) and tuple(run.w.shape) == (3, 2) and torch.equal(run.w[:, 1], ls[1].w)
    stacked2d = bj.PlanarLayer(torch.zeros(3, 2), torch.zeros(3, 2), torch.zeros(2))
    assert (ls[0] @ stacked2d)._plan()[0][0].n_layers == 3                                # a (dim, n_layers) layer joins a run column by column
