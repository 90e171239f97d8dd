# gen_golden.jl — NOT RUN IN THIS ENVIRONMENT (no Julia in the image).  For anyone with Julia:
#
#     julia --project=/path/to/Bijectors.jl scripts/gen_golden.jl        # needs Bijectors (v0.16.x) and JSON
#
# For every tests/golden/<name>.json it evaluates `with_logabsdet_jacobian(b, x)` of the REAL package on the inputs
# stored in the file (b = the file's "julia" expression with its "params" bound to `p`) and rewrites "y",
# "logabsdetjac" and "source".  tests/test_golden_files.py then compares the CPU oracle with these numbers
# (rtol 1e-9, the reference's own link/invlink tolerance) instead of with its own output.
using Bijectors, JSON, LinearAlgebra
import Pkg

tomat(v) = (v isa AbstractVector && !isempty(v) && v[1] isa AbstractVector) ? reduce(hcat, [Float64.(c) for c in v]) : Float64.(v)
cols(y::AbstractMatrix) = [collect(Float64, c) for c in eachcol(y)]
cols(y::AbstractVector) = collect(Float64, y)
cols(y::Real) = Float64(y)

function main()
    dir = joinpath(@__DIR__, "..", "tests", "golden")
    ver = string(Pkg.dependencies()[Base.PkgId(Bijectors).uuid].version)
    for f in sort(filter(endswith(".json"), readdir(dir)))
        doc = JSON.parsefile(joinpath(dir, f))
        global p = doc["params"]
        b = eval(Meta.parse(doc["julia"]))
        ys, ls = Any[], Float64[]
        for x in doc["x"]
            y, l = with_logabsdet_jacobian(b, tomat(x))
            y isa NamedTuple && (l = y.logabsdetjac; y = y.result)       # PlanarLayer returns (result, logabsdetjac)
            push!(ys, cols(y))
            push!(ls, Float64(sum(l)))
        end
        doc["y"], doc["logabsdetjac"], doc["source"] = ys, ls, "Bijectors.jl " * ver
        open(joinpath(dir, f), "w") do io
            JSON.print(io, doc, 1)
        end
        println("ok  ", f)
    end
end
main()
