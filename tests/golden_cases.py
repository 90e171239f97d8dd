"""Registry of the golden-vector cases (one per SURVEY.md §8(a) row + the §8(f) f-4 matrix bijectors).

Each case = a Julia expression that builds the bijector from the dict `p` (evaluated by scripts/gen_golden.jl inside
`using Bijectors`), the parameters `p`, a list of single-sample inputs (the reference's own call shape) and the oracle
call that must reproduce `with_logabsdet_jacobian(b, x)`.  Inputs are generated here with a fixed seed and STORED in
tests/golden/<name>.json, so the Julia script evaluates exactly the same numbers.

    python scripts/gen_golden.py     # writes inputs + ORACLE outputs ("source": "oracle": format check, self-consistency)
    julia  scripts/gen_golden.jl     # overwrites the outputs with Bijectors.jl's ("source": "Bijectors.jl <version>": the pin)
"""
import numpy as np


def _cases():
    from oracle import oracle as o

    r = np.random.default_rng(20260926)
    d = 5
    C = {}

    def add(name, julia, p, xs, fn):
        C[name] = dict(julia=julia, params=p, xs=xs, fn=fn)

    vec = lambda n=d, s=1.0: [list(s * r.normal(size=n)) for _ in range(3)]
    pos = lambda n=d: [list(r.uniform(0.1, 4.0, size=n)) for _ in range(3)]
    unit = lambda lo, hi, n=d: [list(r.uniform(lo + 0.05, hi - 0.05, size=n)) for _ in range(3)]
    ch = lambda ops: (lambda p, x: o.chain(ops(p), np.asarray(x)))

    add("exp", "elementwise(exp)", {}, vec(), ch(lambda p: [(o.OP_EXP, None, None)]))
    add("log", "elementwise(log)", {}, pos(), ch(lambda p: [(o.OP_LOG, None, None)]))
    add("shift_scale_exp", 'elementwise(exp) ∘ Bijectors.Shift(p["b"]) ∘ Bijectors.Scale(p["a"])', {"a": 0.5, "b": 0.1}, vec(),
        ch(lambda p: [(o.OP_SCALE, p["a"], None), (o.OP_SHIFT, p["b"], None), (o.OP_EXP, None, None)]))
    av = list(np.linspace(0.5, 1.5, d) * np.array([1, -1, 1, 1, -1]))
    add("scale_vector", 'Bijectors.Scale(Float64.(p["a"]))', {"a": av}, vec(), ch(lambda p: [(o.OP_SCALE, np.asarray(p["a"]), None)]))
    add("logit", 'Bijectors.Logit(p["a"], p["b"])', {"a": -1.0, "b": 2.0}, unit(-1.0, 2.0), ch(lambda p: [(o.OP_LOGIT, p["a"], p["b"])]))
    add("inv_logit", 'inverse(Bijectors.Logit(p["a"], p["b"]))', {"a": -1.0, "b": 2.0}, vec(), ch(lambda p: [(o.OP_LOGIT_INV, p["a"], p["b"])]))
    add("leaky_relu", 'Bijectors.LeakyReLU(p["alpha"])', {"alpha": 0.1}, vec(), ch(lambda p: [(o.OP_LEAKY_RELU, p["alpha"], None)]))
    add("truncated", 'Bijectors.TruncatedBijector(p["lb"], p["ub"])', {"lb": 0.0, "ub": 2.0}, unit(0.0, 2.0), ch(lambda p: [(o.OP_TRUNCATED, p["lb"], p["ub"])]))
    add("inv_truncated", 'inverse(Bijectors.TruncatedBijector(p["lb"], p["ub"]))', {"lb": 0.0, "ub": 2.0}, vec(), ch(lambda p: [(o.OP_TRUNCATED_INV, p["lb"], p["ub"])]))
    add("truncated_lower", 'Bijectors.TruncatedBijector(p["lb"], Inf)', {"lb": 0.5}, unit(0.5, 5.0), ch(lambda p: [(o.OP_TRUNCATED, p["lb"], np.inf)]))
    one = lambda f: (lambda p, x: tuple(v if np.ndim(v) == 0 or k == 0 else v[0] for k, v in enumerate(f(p, np.asarray(x)))))
    add("ordered", "Bijectors.OrderedBijector()", {}, vec(), one(lambda p, x: o.ordered(x)))
    add("inv_ordered", "inverse(Bijectors.OrderedBijector())", {}, [list(np.cumsum(r.uniform(0.1, 1, size=d))) for _ in range(3)], one(lambda p, x: o.ordered(x, inverse=True)))
    add("simplex", "Bijectors.SimplexBijector()", {}, [list(r.dirichlet(np.ones(d))) for _ in range(3)], one(lambda p, x: o.simplex(x)))
    add("inv_simplex", "inverse(Bijectors.SimplexBijector())", {}, vec(d - 1), one(lambda p, x: o.simplex(x, inverse=True)))
    K = 4
    add("inv_vec_cholesky_U", "inverse(Bijectors.VecCholeskyBijector(:U))", {}, vec(K * (K - 1) // 2, 0.7), one(lambda p, x: o.vec_cholesky(x, inverse=True, uplo="U")))
    add("inv_vec_cholesky_L", "inverse(Bijectors.VecCholeskyBijector(:L))", {}, vec(K * (K - 1) // 2, 0.7), one(lambda p, x: o.vec_cholesky(x, inverse=True, uplo="L")))
    w, u = list(r.normal(size=d)), list(r.normal(size=d))
    add("planar", 'Bijectors.PlanarLayer(Float64.(p["w"]), Float64.(p["u"]), [p["b"]])', {"w": w, "u": u, "b": 0.3}, vec(),
        one(lambda p, x: o.planar(np.asarray(p["w"]), np.asarray(p["u"]), np.asarray([p["b"]]), x)))
    add("inv_planar", 'inverse(Bijectors.PlanarLayer(Float64.(p["w"]), Float64.(p["u"]), [p["b"]]))', {"w": w, "u": u, "b": 0.3}, vec(),
        one(lambda p, x: o.planar(np.asarray(p["w"]), np.asarray(p["u"]), np.asarray([p["b"]]), x, inverse=True)))
    z0 = list(r.normal(size=d))
    add("radial", 'Bijectors.RadialLayer([p["alpha_"]], [p["beta"]], Float64.(p["z0"]))', {"alpha_": 0.4, "beta": -0.7, "z0": z0}, vec(),
        one(lambda p, x: o.radial(p["alpha_"], p["beta"], np.asarray(p["z0"]), x)))
    add("inv_radial", 'inverse(Bijectors.RadialLayer([p["alpha_"]], [p["beta"]], Float64.(p["z0"])))', {"alpha_": 0.4, "beta": -0.7, "z0": z0}, vec(),
        one(lambda p, x: o.radial(p["alpha_"], p["beta"], np.asarray(p["z0"]), x, inverse=True)))
    Kk = 6
    rw, rh, rd = r.normal(size=(d, Kk)), r.normal(size=(d, Kk)), r.normal(size=(d, Kk - 1))
    add("rqs", 'Bijectors.RationalQuadraticSpline(reduce(hcat, p["rw"]), reduce(hcat, p["rh"]), reduce(hcat, p["rd"]), p["B"])',
        {"rw": [list(c) for c in rw.T], "rh": [list(c) for c in rh.T], "rd": [list(c) for c in rd.T], "B": 2.0}, vec(d, 1.5),
        one(lambda p, x: o.rqs(*o.rqs_params(np.array(p["rw"]).T, np.array(p["rh"]).T, np.array(p["rd"]).T, p["B"]), x)))
    add("permute", 'Bijectors.Permute(Int.(p["idx"]))', {"idx": [3, 1, 5, 2, 4]}, vec(),
        lambda p, x: (o.permute(_src(p["idx"]), np.asarray(x)), 0.0))
    mat = lambda kind, n: [[list(c) for c in o.matrix_bijector(kind, 0.6 * r.normal(size=n), inverse=True)[0].T] for _ in range(3)]
    mfn = lambda kind, inv=False: (lambda p, x: tuple(v if k == 0 else v[0] for k, v in enumerate(o.matrix_bijector(kind, np.array(x).T if np.ndim(x) == 2 else np.asarray(x), inverse=inv))))
    add("vec_corr", "Bijectors.VecCorrBijector()", {}, mat("vec_corr", 6), mfn("vec_corr"))
    add("inv_vec_corr", "inverse(Bijectors.VecCorrBijector())", {}, vec(6, 0.6), mfn("vec_corr", True))
    add("corr", "Bijectors.CorrBijector()", {}, mat("vec_corr", 6), mfn("corr"))
    add("pd", "Bijectors.PDBijector()", {}, mat("pd_vec", 10), mfn("pd"))
    add("pd_vec", "Bijectors.PDVecBijector()", {}, mat("pd_vec", 10), mfn("pd_vec"))
    add("inv_pd_vec", "inverse(Bijectors.PDVecBijector())", {}, vec(10, 0.6), mfn("pd_vec", True))
    return C


def _src(idx):
    """Permute(indices): y[idx_i] = x[i] (permute.jl:90-100) -> 0-based source of every output row."""
    src = [0] * len(idx)
    for i, dst in enumerate(idx):
        src[int(dst) - 1] = i
    return np.asarray(src, dtype=np.int32)


_CACHE = None


def cases():
    global _CACHE
    if _CACHE is None:
        _CACHE = _cases()
    return _CACHE
