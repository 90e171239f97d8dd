# NOT RUN IN THIS ENVIRONMENT (no Julia binary, no depot, no network) — shipped so that anyone with
# Julia can time the reference itself on the shapes bench.py uses (SURVEY.md §8(d), "CPU baseline
# beside it").  Bijectors 0.16.x, BenchmarkTools; shapes and seeds as in bench.py / BASELINE.json.
#
#   julia --project -e 'using Pkg; Pkg.add(["Bijectors", "BenchmarkTools", "Distributions"])'
#   julia --project -t 1 bench/reference_cpu.jl            # the reference is single-threaded
#
# Prints one line per config: median / minimum time and M samples/s (sample = one column).
using Bijectors, BenchmarkTools, LinearAlgebra, Random, Printf
using Bijectors: with_logabsdet_jacobian, elementwise, Shift, Scale, PlanarLayer, RationalQuadraticSpline,
                 SimplexBijector, VecCholeskyBijector, inverse

BLAS.set_num_threads(1)
Random.seed!(0)

function report(name, n, b)
    t = median(b).time * 1e-9
    tmin = minimum(b).time * 1e-9
    @printf("%-58s median %10.4f ms  min %10.4f ms  %10.3f Msamples/s\n", name, 1e3t, 1e3tmin, n / t / 1e6)
end

log2n(default) = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : default

# C1: Exp on a Float64 vector of 2^20
let x = randn(2^20)
    b = elementwise(exp)
    report("C1 elementwise(exp) Float64[2^20]", length(x), @benchmark with_logabsdet_jacobian($b, $x))
end

# C2: exp ∘ Shift ∘ Scale, Float32 d = 64 (reference: three allocating passes)
let N = 2^log2n(20), x = randn(Float32, 64, N)
    b = elementwise(exp) ∘ Shift(0.1f0) ∘ Scale(0.5f0)
    report("C2 exp∘Shift∘Scale Float32[64, 2^$(log2n(20))]", N, @benchmark with_logabsdet_jacobian($b, $x))
end

# C3: RationalQuadraticSpline K = 16, d = 32 — the reference has no matrix method: column by column
let N = 2^log2n(14), d = 32, K = 16
    b = RationalQuadraticSpline(randn(Float32, d, K), randn(Float32, d, K), randn(Float32, d, K - 1), 3.0f0)
    x = randn(Float32, d, N)
    f(b, x) = sum(c -> with_logabsdet_jacobian(b, c)[2], eachcol(x))
    report("C3 RQS K=16 forward + logabsdetjac Float32[32, 2^$(log2n(14))]", N, @benchmark $f($b, $x))
end

# C4: 8 PlanarLayers, d = 128
let N = 2^log2n(16), d = 128
    layers = [PlanarLayer(randn(Float32, d) ./ sqrt(Float32(d)), randn(Float32, d) ./ sqrt(Float32(d)), randn(Float32, 1)) for _ in 1:8]
    flow = foldl(∘, layers)
    z = randn(Float32, d, N)
    report("C4 8×PlanarLayer Float32[128, 2^$(log2n(16))]", N, @benchmark with_logabsdet_jacobian($flow, $z))
end

# C5a: SimplexBijector K = 64
let N = 2^log2n(16), K = 64
    x = exp.(randn(Float32, K, N)); x ./= sum(x; dims=1)
    b = SimplexBijector()
    report("C5a SimplexBijector Float32[64, 2^$(log2n(16))]", N, @benchmark with_logabsdet_jacobian($b, $x))
end

# C5b: inverse VecCholeskyBijector K = 64 — single-sample method only: column by column
let N = 2^log2n(10), K = 64, n = K * (K - 1) ÷ 2
    ib = inverse(VecCholeskyBijector(:U))
    y = 0.5f0 .* randn(Float32, n, N)
    f(ib, y) = sum(c -> with_logabsdet_jacobian(ib, c)[2], eachcol(y))
    report("C5b inverse(VecCholeskyBijector) K=64 Float32[2016, 2^$(log2n(10))]", N, @benchmark $f($ib, $y))
end
