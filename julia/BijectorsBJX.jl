# BijectorsBJX.jl — the Julia side of the drop-in boundary (NOT EXECUTED IN THIS ENVIRONMENT:
# the build image has no `julia` binary; this file is written against include/bjx.h and is the
# binding a Bijectors.jl maintainer would add as a package extension, in the same way the AD
# extensions attach more specific methods (ext/BijectorsReverseDiffExt.jl:63-65,
# ext/BijectorsForwardDiffExt.jl:11-15; weak-dep wiring Project.toml:26-42)).
#
# AMDGPU.jl is used ONLY for the device pointer, the device id and the hipStream_t; no
# KernelAbstractions, no CUDA.jl compat layer.  Every method below dispatches on `ROCArray`
# inputs and falls through to the reference's generic CPU methods for anything else.
module BijectorsBJX

using AMDGPU: AMDGPU, ROCArray, ROCVector, ROCMatrix
using Bijectors
using Bijectors: Elementwise, Inverse, Shift, Scale, Logit, LeakyReLU, TruncatedBijector, OrderedBijector,
    SimplexBijector, VecCholeskyBijector, Permute, PlanarLayer, RadialLayer, InvertibleBatchNorm,
    RationalQuadraticSpline, Stacked, VecCorrBijector, CorrBijector, PDBijector, PDVecBijector, NamedStacked
using ChainRulesCore: ChainRulesCore
using Distributions: Distributions
const ROCVecOrMat{T} = Union{ROCVector{T},ROCMatrix{T}}
import Bijectors: transform, logabsdetjac, with_logabsdet_jacobian, with_logabsdet_jacobian!

const libbjx = get(ENV, "BJX_LIBRARY", "libbjx_hip.so")

# ---------------------------------------------------------------- include/bjx.h mirror
const BJX_F32, BJX_F64 = Cint(0), Cint(1)
const BJX_ACCUMULATE, BJX_REF_VECTOR_SCALE_LADJ = UInt32(1), UInt32(2)
const BJX_ERR_UNSUPPORTED = Cint(-3)
@enum OpKind::Int32 OP_EXP = 1 OP_LOG OP_SHIFT OP_SCALE OP_SCALE_INV OP_LOGIT OP_LOGIT_INV OP_LEAKY_RELU OP_TRUNCATED OP_TRUNCATED_INV OP_SIGNFLIP OP_IDENTITY

struct BjxOp            # layout of `bjx_op` (40 bytes)
    kind::Int32
    param_len::Int32
    p0::Float64
    p1::Float64
    v0::Ptr{Cvoid}
    v1::Ptr{Cvoid}
end

dtype(::Type{Float32}) = BJX_F32
dtype(::Type{Float64}) = BJX_F64

mutable struct Context
    h::Ptr{Cvoid}
end
function Context(dev::Integer=AMDGPU.device_id(AMDGPU.device()) - 1, stream=AMDGPU.stream())
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:bjx_create, libbjx), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), dev, stream.stream, h)
    rc == 0 || error("bjx_create failed with status $rc")
    ctx = Context(h[])
    finalizer(c -> ccall((:bjx_destroy, libbjx), Cint, (Ptr{Cvoid},), c.h), ctx)
    return ctx
end
const CTX = Ref{Union{Nothing,Context}}(nothing)
ctx() = something(CTX[], (CTX[] = Context()))

function check(rc::Cint, what)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:bjx_last_error, libbjx), Cstring, (Ptr{Cvoid},), ctx().h))
    rc == -1 && throw(ArgumentError("$what: $msg"))         # BJX_ERR_ARG
    rc == -2 && throw(DimensionMismatch("$what: $msg"))     # BJX_ERR_SHAPE
    error("$what: status $rc: $msg")                        # hipError_t / ncclResult_t
end

dims(x::ROCVector) = (length(x), 1)
dims(x::ROCMatrix) = size(x)
devptr(x::ROCArray) = Ptr{Cvoid}(pointer(x))

# ---------------------------------------------------------------- F1: fused elementwise chains
# Walk `outer ∘ inner` into application order; nothing => not fusable, use the generic method.
ops(b::Elementwise{typeof(exp)}, T, keep) = [BjxOp(Int32(OP_EXP), 0, 0, 0, C_NULL, C_NULL)]
ops(b::Elementwise{typeof(log)}, T, keep) = [BjxOp(Int32(OP_LOG), 0, 0, 0, C_NULL, C_NULL)]
function param_op(kind, a, T, keep, b=nothing)
    if a isa Real
        return BjxOp(Int32(kind), 1, Float64(a), b === nothing ? 0.0 : Float64(b), C_NULL, C_NULL)
    end
    va = ROCArray{T}(a); push!(keep, va)            # parameters may already live on the device
    vb = b === nothing ? nothing : ROCArray{T}(b isa Real ? fill(T(b), length(va)) : b)
    vb === nothing || push!(keep, vb)
    return BjxOp(Int32(kind), length(va), 0, 0, devptr(va), vb === nothing ? C_NULL : devptr(vb))
end
ops(b::Shift, T, keep) = [param_op(OP_SHIFT, b.a, T, keep)]
ops(b::Scale{<:Union{Real,AbstractVector}}, T, keep) = [param_op(OP_SCALE, b.a, T, keep)]
ops(b::Inverse{<:Scale{<:Union{Real,AbstractVector}}}, T, keep) = [param_op(OP_SCALE_INV, b.orig.a, T, keep)]
ops(b::Logit, T, keep) = [param_op(OP_LOGIT, b.a, T, keep, b.b)]
ops(b::Inverse{<:Logit}, T, keep) = [param_op(OP_LOGIT_INV, b.orig.a, T, keep, b.orig.b)]
ops(b::LeakyReLU, T, keep) = [param_op(OP_LEAKY_RELU, b.α, T, keep)]
ops(b::TruncatedBijector, T, keep) = [param_op(OP_TRUNCATED, b.lb, T, keep, b.ub)]
ops(b::Inverse{<:TruncatedBijector}, T, keep) = [param_op(OP_TRUNCATED_INV, b.orig.lb, T, keep, b.orig.ub)]
ops(b::Bijectors.SignFlip, T, keep) = [BjxOp(Int32(OP_SIGNFLIP), 0, 0, 0, C_NULL, C_NULL)]
function ops(b::ComposedFunction, T, keep)            # inner first (composed.jl:4)
    i, o = ops(b.inner, T, keep), ops(b.outer, T, keep)
    (i === nothing || o === nothing) && return nothing
    return vcat(i, o)
end
ops(b, T, keep) = nothing

const Fusable = Union{Elementwise{typeof(exp)},Elementwise{typeof(log)},Shift,Scale,Logit,LeakyReLU,
    TruncatedBijector,Inverse{<:Scale},Inverse{<:Logit},Inverse{<:TruncatedBijector},ComposedFunction}

function chain!(y::ROCArray{T}, b, x::ROCArray{T}; per_sample=nothing) where {T<:Union{Float32,Float64}}
    keep = Any[]
    o = ops(b, T, keep)
    o === nothing && return nothing
    length(o) <= 8 || return nothing
    d, n = dims(x)
    lsum = AMDGPU.zeros(Float64, 1)
    GC.@preserve keep x y lsum per_sample begin
        rc = ccall((:bjx_chain, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), o, length(o), devptr(x), devptr(y),
            per_sample === nothing ? C_NULL : devptr(per_sample), devptr(lsum), d, n, BJX_REF_VECTOR_SCALE_LADJ)
        check(rc, "bjx_chain")
    end
    return T(Array(lsum)[1])          # the reference returns one scalar for elementwise bijectors (§8a')
end

function with_logabsdet_jacobian(b::Fusable, x::ROCArray{T}) where {T<:Union{Float32,Float64}}
    y = similar(x)
    l = chain!(y, b, x)
    l === nothing && return invoke(with_logabsdet_jacobian, Tuple{typeof(b),AbstractArray}, b, x)
    return y, l
end
transform(b::Fusable, x::ROCArray{<:Union{Float32,Float64}}) = first(with_logabsdet_jacobian(b, x))
logabsdetjac(b::Fusable, x::ROCArray{<:Union{Float32,Float64}}) = last(with_logabsdet_jacobian(b, x))
function with_logabsdet_jacobian!(b::Fusable, x::ROCArray{T}, y::ROCArray{T}, logjac) where {T}  # interface.jl:212-218
    l = chain!(y, b, x)
    return y, logjac + l
end

# ---------------------------------------------------------------- structured bijectors
# One helper per ABI entry; `ladj_ps` is the per-column vector the reference returns for
# Ordered / Planar / Radial / BatchNorm, `lsum` the scalar it returns for Simplex (§8a').
function call_struct(sym, T, x, out, pre::Tuple, pretypes::Tuple, rows; per_column::Bool)
    _, n = dims(x)
    lps = per_column ? similar(x, T, n) : nothing
    lsum = per_column ? nothing : AMDGPU.zeros(Float64, 1)
    GC.@preserve x out lps lsum begin
        rc = ccall((sym, libbjx), Cint,
            (Ptr{Cvoid}, Cint, pretypes..., Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), pre..., devptr(x), devptr(out),
            lps === nothing ? C_NULL : devptr(lps), lsum === nothing ? C_NULL : devptr(lsum), rows, n, UInt32(0))
        check(rc, String(sym))
    end
    return per_column ? lps : T(Array(lsum)[1])
end

function with_logabsdet_jacobian(b::OrderedBijector, y::ROCMatrix{T}) where {T}          # ordered.jl:22,80
    x = similar(y)
    return x, call_struct(:bjx_ordered, T, y, x, (Cint(0),), (Cint,), size(y, 1); per_column=true)
end
function with_logabsdet_jacobian(ib::Inverse{OrderedBijector}, x::ROCMatrix{T}) where {T}
    y = similar(x)
    return y, call_struct(:bjx_ordered, T, x, y, (Cint(1),), (Cint,), size(x, 1); per_column=true)
end
function with_logabsdet_jacobian(b::SimplexBijector, x::ROCMatrix{T}) where {T}           # simplex.jl:14,141-143
    K = size(x, 1)
    y = similar(x, K - 1, size(x, 2))
    return y, call_struct(:bjx_simplex, T, x, y, (Cint(0),), (Cint,), K; per_column=false)
end
function with_logabsdet_jacobian(ib::Inverse{SimplexBijector}, y::ROCMatrix{T}) where {T}
    K = size(y, 1) + 1
    x = similar(y, K, size(y, 2))
    return x, call_struct(:bjx_simplex, T, y, x, (Cint(1),), (Cint,), K; per_column=false)
end
function with_logabsdet_jacobian(flow::PlanarLayer{<:ROCVector{T}}, z::ROCMatrix{T}) where {T}   # planar_layer.jl:102-110
    out = similar(z)
    l = call_struct(:bjx_planar, T, z, out,
        (Cint(0), devptr(flow.w), devptr(flow.u), devptr(flow.b), Cint(1)), (Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
        size(z, 1); per_column=true)
    return (result=out, logabsdetjac=l)
end
function with_logabsdet_jacobian(flow::RadialLayer, z::ROCMatrix{T}) where {T}                   # radial_layer.jl:58-72
    out = similar(z)
    l = call_struct(:bjx_radial, T, z, out, (Cint(0), devptr(flow.α_), devptr(flow.β), devptr(flow.z_0)),
        (Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), size(z, 1); per_column=true)
    return (result=out, logabsdetjac=l)
end
function with_logabsdet_jacobian(bn::InvertibleBatchNorm, x::ROCMatrix{T}) where {T}             # normalise.jl:41-68 (eval)
    Bijectors.istraining() && return invoke(with_logabsdet_jacobian, Tuple{InvertibleBatchNorm,Any}, bn, x)
    size(x, 1) == length(bn.b) || error("InvertibleBatchNorm expected $(length(bn.b)) channels, got $(size(x, 1))")
    out = similar(x)
    l = call_struct(:bjx_batchnorm, T, x, out,
        (Cint(0), devptr(bn.b), devptr(bn.logs), devptr(bn.m), devptr(bn.v), Float64(bn.eps)),
        (Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble), size(x, 1); per_column=true)
    return out, l
end
# RationalQuadraticSpline{<:ROCMatrix} on a batch (the reference has only the single-column method,
# rational_quadratic_spline.jl:173-178,304-309,363-367): returns the per-column log-det vector.
function with_logabsdet_jacobian(b::RationalQuadraticSpline{<:ROCMatrix{T}}, x::ROCMatrix{T}) where {T}
    y = similar(x)
    l = call_struct(:bjx_rqs, T, x, y,
        (Cint(0), devptr(b.widths), devptr(b.heights), devptr(b.derivatives), Cint(size(b.widths, 2))),
        (Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), size(x, 1); per_column=true)
    return y, l
end
# VecCholeskyBijector, Permute, Coupling and the Inverse{…} flow methods follow the same pattern
# (bjx_vec_cholesky / bjx_permute / bjx_coupling_* / inverse = Cint(1)); see INTEGRATION.md.

# ---------------------------------------------------------------- matrix-variate constraint bijectors (SURVEY.md §8f f-4)
# corr.jl:64-162, pd.jl:1-60.  The reference defines them for ONE matrix; a K x K x N ROCArray is a batch of N samples
# and returns the per-sample log-det vector.  (A single ROCMatrix is the N = 1 case and returns the scalar.)
const MatrixKinds = Union{VecCorrBijector,CorrBijector,PDBijector,PDVecBijector}
bjx_symbol(::VecCorrBijector) = :bjx_vec_corr
bjx_symbol(::CorrBijector) = :bjx_corr
bjx_symbol(::PDBijector) = :bjx_pd
bjx_symbol(::PDVecBijector) = :bjx_pd_vec
packed_length(::VecCorrBijector, K) = (K * (K - 1)) ÷ 2                      # corr.jl:150-154
packed_length(::PDVecBijector, K) = (K * (K + 1)) ÷ 2                        # pd.jl:50-54
function matrix_call(b::MatrixKinds, inv::Bool, inp::ROCArray{T}, out::ROCArray{T}, K, n) where {T}
    lps = similar(inp, T, n)
    GC.@preserve inp out lps begin
        rc = ccall((bjx_symbol(b), libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), Cint(inv), devptr(inp), devptr(out), devptr(lps), C_NULL, K, n, UInt32(0))
        check(rc, String(bjx_symbol(b)))
    end
    return lps
end
function with_logabsdet_jacobian(b::Union{VecCorrBijector,PDVecBijector}, X::ROCArray{T,3}) where {T}
    K, n = size(X, 1), size(X, 3)
    size(X, 2) == K || throw(DimensionMismatch("sizes should be equal; received $(size(X)[1:2])"))
    y = similar(X, packed_length(b, K), n)
    return y, matrix_call(b, false, X, y, K, n)
end
function with_logabsdet_jacobian(ib::Inverse{<:Union{VecCorrBijector,PDVecBijector}}, y::ROCMatrix{T}) where {T}
    b = ib.orig
    K = b isa VecCorrBijector ? Bijectors._triu1_dim_from_length(size(y, 1)) : Bijectors._triu_dim_from_length(size(y, 1))
    X = similar(y, K, K, size(y, 2))
    return X, matrix_call(b, true, y, X, K, size(y, 2))
end
function with_logabsdet_jacobian(b::Union{CorrBijector,PDBijector}, X::ROCArray{T,3}) where {T}
    Y = similar(X)
    return Y, matrix_call(b, false, X, Y, size(X, 1), size(X, 3))
end
function with_logabsdet_jacobian(ib::Inverse{<:Union{CorrBijector,PDBijector}}, Y::ROCArray{T,3}) where {T}
    X = similar(Y)
    return X, matrix_call(ib.orig, true, Y, X, size(Y, 1), size(Y, 3))
end
# one matrix (the reference's call shape): the N = 1 batch, scalar log-det
function with_logabsdet_jacobian(b::Union{MatrixKinds,Inverse{<:MatrixKinds}}, x::ROCVecOrMat{T}) where {T}
    out, l = with_logabsdet_jacobian(b, reshape(x, size(x)..., 1))
    return dropdims(out; dims=ndims(out)), Array(l)[1]
end

# Scale with a matrix parameter (scale.jl:14,17,35-36): a * x, a \ y, logabsdet(a) once
function scale_matrix(a::ROCMatrix{T}, x::ROCVecOrMat{T}, inv::Bool) where {T}
    d, n = dims(x)
    size(a) == (d, d) || throw(DimensionMismatch("Scale with a $(size(a)) matrix applied to $d rows"))
    y = similar(x)
    lsum = AMDGPU.zeros(Float64, 1)
    GC.@preserve a x y lsum begin
        rc = ccall((:bjx_scale_matrix, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), Cint(inv), devptr(a), devptr(x), devptr(y), C_NULL, devptr(lsum), d, n, BJX_REF_VECTOR_SCALE_LADJ)
        check(rc, "bjx_scale_matrix")
    end
    return y, T(Array(lsum)[1])
end
with_logabsdet_jacobian(b::Scale{<:ROCMatrix{T}}, x::ROCVecOrMat{T}) where {T} = scale_matrix(b.a, x, false)
with_logabsdet_jacobian(ib::Inverse{<:Scale{<:ROCMatrix{T}}}, y::ROCVecOrMat{T}) where {T} = scale_matrix(ib.orig.a, y, true)

# InvertibleBatchNorm in training mode on a batch sharded over ranks (normalise.jl:51-60; SURVEY.md §8e "Exception"):
# statistics of this rank's columns -> the host's collective (MPI.Allreduce!, or bjx_allreduce_sum_f64 after bjx_comm_init)
# -> update of the moving statistics and transform with the GLOBAL statistics.
function batchnorm_train!(bn::InvertibleBatchNorm, x::ROCMatrix{T}; allreduce! = identity) where {T}
    d, n = size(x)
    stats = AMDGPU.zeros(Float64, 2d + 1)
    GC.@preserve bn x stats check(ccall((:bjx_batchnorm_stats, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), devptr(bn.m), devptr(x), devptr(stats), d, n), "bjx_batchnorm_stats")
    allreduce!(stats)                                      # sum of the 2d+1 Float64 values over the ranks
    y = similar(x); lps = similar(x, T, n)
    GC.@preserve bn x y lps stats check(ccall((:bjx_batchnorm_train_apply, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), devptr(bn.b), devptr(bn.logs), devptr(bn.m), devptr(bn.v), Float64(bn.eps), Float64(bn.mtm),
        devptr(stats), devptr(x), devptr(y), devptr(lps), C_NULL, d, n, UInt32(0)), "bjx_batchnorm_train_apply")
    return y, lps
end

# ---------------------------------------------------------------- Stacked (SURVEY.md §8f f-4)
# stacked.jl:27-252: every segment whose bijector is a fusable elementwise chain goes into ONE launch.
struct BjxSegment
    in_lo::Int64; out_lo::Int64; len::Int64; n_ops::Int32; reserved::Int32
    ops::NTuple{4,BjxOp}
end
const NOOP = BjxOp(Int32(OP_IDENTITY), 0, 0, 0, C_NULL, C_NULL)
function with_logabsdet_jacobian(sb::Stacked, x::ROCVecOrMat{T}) where {T<:Union{Float32,Float64}}
    d, n = dims(x)
    sb.length_in == d || error("input length mismatch ($(sb.length_in) != $d)")          # stacked.jl:157
    keep = Any[]
    segs = BjxSegment[]
    for (b, rin, rout) in zip(sb.bs, sb.ranges_in, sb.ranges_out)
        o = b === identity ? BjxOp[] : ops(b, T, keep)
        (o === nothing || length(o) > 4 || length(rin) != length(rout)) &&
            return stacked_structured(sb, x)                                               # Simplex / Ordered blocks: in place, below
        push!(segs, BjxSegment(first(rin) - 1, first(rout) - 1, length(rin), length(o), 0,
                               ntuple(k -> k <= length(o) ? o[k] : NOOP, 4)))
    end
    y = similar(x)
    lsum = AMDGPU.zeros(Float64, 1)
    GC.@preserve keep x y lsum begin
        rc = ccall((:bjx_stacked, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), segs, length(segs), devptr(x), devptr(y), C_NULL, devptr(lsum), d, n, 0)
        check(rc, "bjx_stacked")
    end
    return y, T(Array(lsum)[1])
end

# Stacked with Simplex / Ordered segments (stacked.jl:142-166) without slicing copies: the elementwise segments in one
# bjx_stacked_ld launch between matrices of different heights (identity placeholders on the structured rows), then
# bjx_simplex_ld / bjx_ordered_ld on row windows of the same matrices, accumulating their log-dets.
struct BjxBlock            # include/bjx.h: bjx_block
    kind::Cint; reserved::Cint; in_lo::Int64; out_lo::Int64; len_in::Int64; len_out::Int64
end
structured_entry(::SimplexBijector) = (:bjx_simplex_ld, false)
structured_entry(::Inverse{SimplexBijector}) = (:bjx_simplex_ld, true)
structured_entry(::OrderedBijector) = (:bjx_ordered_ld, false)
structured_entry(::Inverse{OrderedBijector}) = (:bjx_ordered_ld, true)
structured_entry(b) = nothing
function stacked_structured(sb::Stacked, x::ROCMatrix{T}) where {T<:Union{Float32,Float64}}
    d, n = size(x)
    dout = last(last(sb.ranges_out))
    keep = Any[]; segs = BjxSegment[]; later = Any[]
    for (b, rin, rout) in zip(sb.bs, sb.ranges_in, sb.ranges_out)
        o = b === identity ? BjxOp[] : ops(b, T, keep)
        if o !== nothing && length(o) <= 4 && length(rin) == length(rout)
            push!(segs, BjxSegment(first(rin) - 1, first(rout) - 1, length(rin), length(o), 0, ntuple(k -> k <= length(o) ? o[k] : NOOP, 4)))
        else
            e = structured_entry(b)
            e === nothing && return invoke(with_logabsdet_jacobian, Tuple{Stacked,AbstractMatrix}, sb, x)
            push!(later, (e, rin, rout))
            push!(segs, BjxSegment(min(first(rin) - 1, d - length(rout)), first(rout) - 1, length(rout), 0, 0, ntuple(_ -> NOOP, 4)))
        end
    end
    y = similar(x, dout, n); lps = AMDGPU.zeros(T, n)
    # ONE launch when a column fits the LDS tile (bjx_stacked_mixed: a lane walks its column through every segment)
    blocks = [BjxBlock(Cint(sym === :bjx_simplex_ld ? (inv ? 2 : 1) : (inv ? 4 : 3)), Cint(0), first(rin) - 1, first(rout) - 1, length(rin), length(rout))
              for ((sym, inv), rin, rout) in later]
    elem = [sg for sg in segs if !any(b -> b.out_lo == sg.out_lo, blocks)]
    rc = GC.@preserve keep x y lps ccall((:bjx_stacked_mixed, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{BjxBlock}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt32),
        ctx().h, dtype(T), elem, length(elem), blocks, length(blocks), devptr(x), d, devptr(y), dout, devptr(lps), C_NULL, n, UInt32(0))
    rc == 0 && return y, lps
    rc == BJX_ERR_UNSUPPORTED || check(rc, "bjx_stacked_mixed")       # taller columns: the window launches below
    GC.@preserve keep x y lps begin
        check(ccall((:bjx_stacked_ld, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), segs, length(segs), devptr(x), d, devptr(y), dout, devptr(lps), C_NULL, dout, n, UInt32(0)), "bjx_stacked_ld")
        for ((sym, inv), rin, rout) in later
            K = sym === :bjx_simplex_ld ? (inv ? length(rout) : length(rin)) : length(rin)
            check(ccall((sym, libbjx), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
                ctx().h, dtype(T), Cint(inv), devptr(x) + (first(rin) - 1) * sizeof(T), d, devptr(y) + (first(rout) - 1) * sizeof(T), dout,
                devptr(lps), C_NULL, K, n, BJX_ACCUMULATE), String(sym))
        end
    end
    return y, lps
end

# Mean-field family y = tail(μ .+ σ .* z) (ADVI): input pullback and the (μ̄, σ̄) reductions in ONE pass over z and ȳ.
# moments[1:d] = Σ_n z̄, moments[d+1:2d] = Σ_n z̄ .* z  =>  μ̄ = moments[1:d] ./ σ,  σ̄ = (moments[d+1:2d] .+ sum(ℓ̄)) ./ σ
function meanfield_pullback(chain, z::ROCMatrix{T}, ȳ::ROCMatrix{T}, ℓ̄::ROCVector{T}) where {T<:Union{Float32,Float64}}
    d, n = dims(z)
    keep = Any[]
    o = ops(chain, T, keep)                                  # tail ∘ Shift(μ) ∘ Scale(σ) as <= 4 elementwise ops
    seg = [BjxSegment(0, 0, d, length(o), 0, ntuple(k -> k <= length(o) ? o[k] : NOOP, 4))]
    z̄ = similar(z)
    moments = AMDGPU.zeros(Float64, 2d + 1)
    GC.@preserve keep z ȳ ℓ̄ z̄ moments check(ccall((:bjx_stacked_vjp_moments, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), seg, 1, devptr(z), devptr(ȳ), devptr(ℓ̄), devptr(z̄), devptr(moments), d, n), "bjx_stacked_vjp_moments")
    return z̄, moments
end

# ---------------------------------------------------------------- reverse-mode pullbacks (SURVEY.md §8f f-1)
# The reference's own rrules (ext/BijectorsChainRulesCoreExt.jl:65-197, :311-320) for ROCArray primals:
# the pullback closure calls the `_vjp` entry with the saved primal input.
function ChainRulesCore.rrule(::typeof(Bijectors._transform_ordered), y::ROCMatrix{T}) where {T}
    x = first(with_logabsdet_jacobian(OrderedBijector(), y))
    function _transform_ordered_adjoint(Δ)
        ȳ = similar(y)
        Δc = ROCArray{T}(ChainRulesCore.unthunk(Δ))
        GC.@preserve y Δc ȳ check(ccall((:bjx_ordered_vjp, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
            ctx().h, dtype(T), 0, devptr(y), devptr(Δc), C_NULL, devptr(ȳ), size(y, 1), size(y, 2)), "bjx_ordered_vjp")
        return ChainRulesCore.NoTangent(), ȳ
    end
    return x, _transform_ordered_adjoint
end
function ChainRulesCore.rrule(::typeof(Bijectors._inv_link_chol_lkj), y::ROCMatrix{T}) where {T}   # columns = samples
    K = Bijectors._triu1_dim_from_length(size(y, 1)); n = size(y, 2)
    W = similar(y, K, K, n); logJ = similar(y, n)
    # primal: bjx_vec_cholesky(inverse = 1, uplo = 'U'); pullback:
    function pullback_inv_link_chol_lkj((ΔW, ΔlogJ))
        Δy = similar(y)
        GC.@preserve y ΔW ΔlogJ Δy check(ccall((:bjx_vec_cholesky_inv_vjp, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
            ctx().h, dtype(T), Cint('U'), devptr(y), devptr(ΔW), devptr(ΔlogJ), devptr(Δy), K, n), "bjx_vec_cholesky_inv_vjp")
        return ChainRulesCore.NoTangent(), Δy
    end
    return (W, logJ), pullback_inv_link_chol_lkj
end

# forward LKJ link on a batch of factors W[K, K, n] (ext/BijectorsChainRulesCoreExt.jl:199-311)
for (f, uplo) in ((:_link_chol_lkj_from_upper, 'U'), (:_link_chol_lkj_from_lower, 'L'))
    @eval function ChainRulesCore.rrule(::typeof(Bijectors.$f), W::ROCArray{T,3}) where {T}
        K, n = size(W, 1), size(W, 3)
        y = first(with_logabsdet_jacobian(VecCholeskyBijector(Symbol($uplo)), W))
        function pullback_link_chol_lkj(Δz)
            ΔW = similar(W); Δc = ROCArray{T}(ChainRulesCore.unthunk(Δz))
            GC.@preserve W Δc ΔW check(ccall((:bjx_vec_cholesky_fwd_vjp, libbjx), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
                ctx().h, dtype(T), Cint($uplo), devptr(W), devptr(Δc), devptr(ΔW), K, n), "bjx_vec_cholesky_fwd_vjp")
            return ChainRulesCore.NoTangent(), ΔW
        end
        return y, pullback_link_chol_lkj
    end
end
# PlanarLayer: input pullback + parameter cotangents (w̄, ū, b̄) through get_u_hat (bjx_planar_vjp_params)
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), flow::PlanarLayer, z::ROCMatrix{T}) where {T}
    out = with_logabsdet_jacobian(flow, z)
    function pullback_planar_params((Δy, Δl))
        z̄ = similar(z); Δyc = ROCArray{T}(ChainRulesCore.unthunk(Δy)); Δlc = ROCArray{T}(ChainRulesCore.unthunk(Δl))
        w, u, b = ROCArray{T}(flow.w), ROCArray{T}(flow.u), ROCArray{T}(flow.b)
        w̄, ū, b̄ = similar(w), similar(u), similar(b)
        work = similar(z, 2 * size(z, 2))                      # 2 * n_layers * batch, one layer
        GC.@preserve z Δyc Δlc z̄ w u b w̄ ū b̄ work check(ccall((:bjx_planar_vjp_params, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
            ctx().h, dtype(T), devptr(w), devptr(u), devptr(b), 1, devptr(z), devptr(Δyc), devptr(Δlc), devptr(z̄),
            devptr(w̄), devptr(ū), devptr(b̄), devptr(work), size(z, 1), size(z, 2)), "bjx_planar_vjp_params")
        return ChainRulesCore.NoTangent(), ChainRulesCore.Tangent{typeof(flow)}(w = w̄, u = ū, b = b̄), z̄
    end
    return out, pullback_planar_params
end
# inverse(PlanarLayer): input pullback AND parameter cotangents.  Implicit function theorem (the reference differentiates the
# Newton root through its find_alpha rule, ext/BijectorsChainRulesCoreExt.jl:42-46): with x = f⁻¹(y) and the inverse's log-det
# -ℓ(x),  ȳ = J⁻ᵀ(x̄ - ℓ̄ ∇ₓℓ)  (bjx_planar_vjp, inverse = 1)  and  θ̄ = the FORWARD parameter pullback at x with cotangents
# (-ȳ, -ℓ̄)  (bjx_planar_vjp_params) — no new kernel, the root is not differentiated through.
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), flow::Inverse{<:PlanarLayer}, z::ROCMatrix{T}) where {T}
    inv = true; pl = flow.orig
    out = with_logabsdet_jacobian(flow, z)
    x = out[1]                                                  # the pre-image: the point where the forward rule is evaluated
    function pullback_planar((Δy, Δl))
        z̄ = similar(z); Δyc = ROCArray{T}(ChainRulesCore.unthunk(Δy)); Δlc = ROCArray{T}(ChainRulesCore.unthunk(Δl))
        w, u, b = ROCArray{T}(pl.w), ROCArray{T}(pl.u), ROCArray{T}(pl.b)
        GC.@preserve z Δyc Δlc z̄ w u b check(ccall((:bjx_planar_vjp, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
            ctx().h, dtype(T), Cint(inv), devptr(w), devptr(u), devptr(b), 1, devptr(z), devptr(Δyc), devptr(Δlc), devptr(z̄),
            size(z, 1), size(z, 2)), "bjx_planar_vjp")
        g, gl = -z̄, -Δlc
        scratch = similar(z)                                    # the forward rule's input cotangent (= -x̄): not used
        w̄, ū, b̄ = similar(w), similar(u), similar(b)
        work = similar(z, 2 * size(z, 2))
        GC.@preserve x g gl scratch w u b w̄ ū b̄ work check(ccall((:bjx_planar_vjp_params, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
            ctx().h, dtype(T), devptr(w), devptr(u), devptr(b), 1, devptr(x), devptr(g), devptr(gl), devptr(scratch),
            devptr(w̄), devptr(ū), devptr(b̄), devptr(work), size(z, 1), size(z, 2)), "bjx_planar_vjp_params")
        return ChainRulesCore.NoTangent(), ChainRulesCore.Tangent{typeof(flow)}(orig = ChainRulesCore.Tangent{typeof(pl)}(w = w̄, u = ū, b = b̄)), z̄
    end
    return out, pullback_planar
end

# RadialLayer: input AND parameter cotangents (bjx_radial_vjp_params; raw α_, β behind softplus, radial_layer.jl:43-60)
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), flow::RadialLayer, z::ROCMatrix{T}) where {T}
    out = with_logabsdet_jacobian(flow, z)
    function pullback_radial((Δy, Δl))
        Δyc, Δlc = ROCArray{T}(ChainRulesCore.unthunk(Δy)), ROCArray{T}(ChainRulesCore.unthunk(Δl))
        z̄, ᾱ, β̄, z̄0 = similar(z), similar(flow.α_), similar(flow.β), similar(flow.z_0)
        work = similar(z, 2 * size(z, 2))
        GC.@preserve z Δyc Δlc z̄ ᾱ β̄ z̄0 work check(ccall((:bjx_radial_vjp_params, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
            ctx().h, dtype(T), devptr(flow.α_), devptr(flow.β), devptr(flow.z_0), devptr(z), devptr(Δyc), devptr(Δlc), devptr(z̄),
            devptr(ᾱ), devptr(β̄), devptr(z̄0), devptr(work), size(z, 1), size(z, 2)), "bjx_radial_vjp_params")
        return ChainRulesCore.NoTangent(), ChainRulesCore.Tangent{typeof(flow)}(α_ = ᾱ, β = β̄, z_0 = z̄0), z̄
    end
    return out, pullback_radial
end

# RationalQuadraticSpline with matrix parameters: input pullback AND the cotangents of the knot arrays summed over
# the batch in one pass (bjx_rqs_vjp_knots with in_bar; rational_quadratic_spline.jl:128-357 has no hand-written rule).  For a spline made by the `B`
# constructor (:109-123) the wrapper that owns the raw parameters chains on with `rqs_params_pullback` (bjx_rqs_params_vjp).
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), b::RationalQuadraticSpline{<:ROCMatrix{T}}, x::ROCMatrix{T}) where {T}
    out = with_logabsdet_jacobian(b, x)
    K1 = size(b.widths, 2)
    function pullback_rqs((Δy, Δl))
        Δyc, Δlc = ROCArray{T}(ChainRulesCore.unthunk(Δy)), ROCArray{T}(ChainRulesCore.unthunk(Δl))
        x̄, w̄, h̄, d̄ = similar(x), similar(b.widths), similar(b.heights), similar(b.derivatives)
        GC.@preserve x Δyc Δlc x̄ w̄ h̄ d̄ begin
            check(ccall((:bjx_rqs_vjp_knots, libbjx), Cint,    # x̄ and the knot cotangents in one pass over x, Δy, Δl
                (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
                ctx().h, dtype(T), 0, devptr(b.widths), devptr(b.heights), devptr(b.derivatives), K1, devptr(x), devptr(Δyc), devptr(Δlc), devptr(x̄),
                devptr(w̄), devptr(h̄), devptr(d̄), size(x, 1), size(x, 2)), "bjx_rqs_vjp_knots")
        end
        return ChainRulesCore.NoTangent(), ChainRulesCore.Tangent{typeof(b)}(widths = w̄, heights = h̄, derivatives = d̄), x̄
    end
    return out, pullback_rqs
end

# pullback of the `B` constructor: knot cotangents (dim, K+1) -> cotangents of the unconstrained (dim, K), (dim, K), (dim, K-1)
function rqs_params_pullback(raw_w::ROCMatrix{T}, raw_h::ROCMatrix{T}, raw_d::ROCMatrix{T}, B::Real, w̄, h̄, d̄) where {T}
    r̄w, r̄h, r̄d = similar(raw_w), similar(raw_h), similar(raw_d)
    GC.@preserve raw_w raw_h raw_d w̄ h̄ d̄ r̄w r̄h r̄d check(ccall((:bjx_rqs_params_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ctx().h, dtype(T), devptr(raw_w), devptr(raw_h), devptr(raw_d), size(raw_w, 2), size(raw_w, 1), Float64(B),
        devptr(w̄), devptr(h̄), devptr(d̄), devptr(r̄w), devptr(r̄h), devptr(r̄d)), "bjx_rqs_params_vjp")
    return r̄w, r̄h, r̄d
end

# captured steps (hipGraph): record the calls of `f()` once, replay them with one launch (include/bjx.h, bjx_graph_*)
function capture(f)
    check(ccall((:bjx_graph_begin, libbjx), Cint, (Ptr{Cvoid},), ctx().h), "bjx_graph_begin")
    g = Ref{Ptr{Cvoid}}(C_NULL)
    try
        f()
    finally
        check(ccall((:bjx_graph_end, libbjx), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), ctx().h, g), "bjx_graph_end")
    end
    return g[]
end
replay(g::Ptr{Cvoid}) = check(ccall((:bjx_graph_launch, libbjx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx().h, g), "bjx_graph_launch")

# ---------------------------------------------------------------- logpdf of a TransformedDistribution (SURVEY.md §8f f-3)
# src/transformed_distribution.jl:164-169 in ONE pass over y: the inverse chain, the whitening of the diagonal-normal
# base and the standard-normal density are ops of the same launch; the pre-image is not stored (y pointer = C_NULL).
const OP_STDNORMAL_LOGPDF = 13
const BJX_BASE_STDNORMAL = UInt32(1) << 2
function Distributions.logpdf(td::Bijectors.MvTransformed{<:Distributions.MvNormal}, y::ROCMatrix{T}) where {T<:Union{Float32,Float64}}
    Σ = td.dist.Σ
    Σ isa Union{Distributions.PDMats.PDiagMat,Distributions.PDMats.ScalMat} || return invoke(Distributions.logpdf, Tuple{Bijectors.MvTransformed,AbstractMatrix}, td, y)
    keep = Any[]
    o = ops(inverse(td.transform), T, keep)
    d, n = dims(y)
    lp = similar(y, n)
    μ, σ = td.dist.μ, sqrt.(Array(Distributions.PDMats.diag(Σ)))
    if o !== nothing && length(o) + 3 <= 8
        o = vcat(o, param_op(OP_SHIFT, -μ, T, keep), param_op(OP_SCALE_INV, σ, T, keep),
                 BjxOp(Int32(OP_STDNORMAL_LOGPDF), 0, 0, 0, C_NULL, C_NULL))
        GC.@preserve keep y lp check(ccall((:bjx_chain, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), o, length(o), devptr(y), C_NULL, devptr(lp), C_NULL, d, n, UInt32(0)), "bjx_chain")
        return lp
    end
    # PlanarLayer stacks with a standard-normal base: bjx_planar(inverse = 1, out = C_NULL, flags = BJX_BASE_STDNORMAL);
    # anything else: x, logjac = with_logabsdet_jacobian(inverse(td.transform), y), then the 3-op density chain on x.
    x, logjac = with_logabsdet_jacobian(inverse(td.transform), y)
    return Distributions.logpdf(Bijectors.transformed(td.dist), x) .+ logjac
end

# ---------------------------------------------------------------- multi-GPU (one process per GPU)
comm_unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:bjx_comm_unique_id, libbjx), Cint, (Ptr{UInt8},), id), "bjx_comm_unique_id"); id)
comm_init(nranks, rank, id::Vector{UInt8}) = check(ccall((:bjx_comm_init, libbjx), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx().h, nranks, rank, id), "bjx_comm_init")
allreduce_logabsdetjac!(partial::ROCVector{Float64}) = (check(ccall((:bjx_allreduce_sum_f64, libbjx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), ctx().h, devptr(partial), length(partial)), "bjx_allreduce_sum_f64"); partial)

end # module
