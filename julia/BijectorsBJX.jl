# BijectorsBJX.jl — the Julia side of the drop-in boundary (NOT EXECUTED IN THIS ENVIRONMENT:
# the build image has no `julia` binary; this file is written against include/bjx.h and is the
# binding a Bijectors.jl maintainer would add as a package extension, in the same way the AD
# extensions attach more specific methods (ext/BijectorsReverseDiffExt.jl:63-65,
# ext/BijectorsForwardDiffExt.jl:11-15; weak-dep wiring Project.toml:26-42)).
# tests/test_julia_binding.py checks it STATICALLY against include/bjx.h: every entry point of the
# header is `ccall`ed here with the header's arity and C types, and every bijector of SURVEY.md
# §8(b) x {plain, Inverse} has a launch plan and therefore the six interface methods below.
#
# AMDGPU.jl is used ONLY for the device pointer, the device id and the hipStream_t; no
# KernelAbstractions, no CUDA.jl compat layer.  Every method below dispatches on `ROCArray`
# inputs and falls through to the reference's generic CPU methods for anything else.
#
# Structure: `plan(b, x)` turns (bijector, input) into a `Plan` — output shape, the reference's
# log-det return shape (SURVEY.md §8a'), and a closure holding the one `ccall`.  The six entry
# points of src/interface.jl:156-218 are then written ONCE, for every planned bijector:
#   with_logabsdet_jacobian, transform, logabsdetjac, transform!, logabsdetjac!, with_logabsdet_jacobian!
module BijectorsBJX

using AMDGPU: AMDGPU, ROCArray, ROCVector, ROCMatrix
using Bijectors
using Bijectors: Elementwise, Inverse, Shift, Scale, Logit, LeakyReLU, TruncatedBijector, OrderedBijector,
    SimplexBijector, VecCholeskyBijector, Permute, PlanarLayer, RadialLayer, InvertibleBatchNorm,
    RationalQuadraticSpline, Stacked, VecCorrBijector, CorrBijector, PDBijector, PDVecBijector, NamedStacked,
    Coupling, PartitionMask, Columnwise
const VB = Bijectors.VectorBijectors
using ChainRulesCore: ChainRulesCore, NoTangent, Tangent, unthunk
using Distributions: Distributions
using SparseArrays: SparseArrays
using Random: Random
const ROCVecOrMat{T} = Union{ROCVector{T},ROCMatrix{T}}
const BjxFloat = Union{Float32,Float64}
import Bijectors: transform, transform!, logabsdetjac, logabsdetjac!, with_logabsdet_jacobian, with_logabsdet_jacobian!

const libbjx = get(ENV, "BJX_LIBRARY", "libbjx_hip.so")

# ---------------------------------------------------------------- include/bjx.h mirror
const BJX_F32, BJX_F64 = Cint(0), Cint(1)
const BJX_ACCUMULATE, BJX_REF_VECTOR_SCALE_LADJ = UInt32(1), UInt32(2)
const BJX_BASE_STDNORMAL, BJX_INPUT_STDNORMAL = UInt32(1) << 2, UInt32(1) << 3
const BJX_COUPLING_SCALE_BCAST, BJX_COUPLING_SHIFT_BCAST = UInt32(1) << 4, UInt32(1) << 5
const BJX_ERR_UNSUPPORTED = Cint(-3)
const BJX_OPT_INKERNEL_FINALIZE = Cint(1)
@enum OpKind::Int32 OP_EXP = 1 OP_LOG OP_SHIFT OP_SCALE OP_SCALE_INV OP_LOGIT OP_LOGIT_INV OP_LEAKY_RELU OP_TRUNCATED OP_TRUNCATED_INV OP_SIGNFLIP OP_IDENTITY OP_STDNORMAL_LOGPDF

struct BjxOp            # layout of `bjx_op` (40 bytes)
    kind::Int32
    param_len::Int32
    p0::Float64
    p1::Float64
    v0::Ptr{Cvoid}
    v1::Ptr{Cvoid}
end

struct BjxSegment       # layout of `bjx_segment` (BJX_MAX_SEG_OPS = 4)
    in_lo::Int64; out_lo::Int64; len::Int64; n_ops::Int32; reserved::Int32
    ops::NTuple{4,BjxOp}
end
const NOOP = BjxOp(Int32(OP_IDENTITY), 0, 0, 0, C_NULL, C_NULL)
struct BjxBlock         # layout of `bjx_block`
    kind::Cint; reserved::Cint; in_lo::Int64; out_lo::Int64; len_in::Int64; len_out::Int64
end

dtype(::Type{Float32}) = BJX_F32
dtype(::Type{Float64}) = BJX_F64

# One context per (device, hipStream_t) — the boundary is RE-ENTRANT (VERDICT r05 "do this" #3; SURVEY.md §8b "Threading": a context
# is not thread-safe, distinct contexts are).  The reference's entry points are pure functions callable from any task
# (src/interface.jl:156-218) and Turing's default multi-chain mode is MCMCThreads; AMDGPU.jl gives every Task its own HIPStream,
# so every task gets its own context: its own stream, its own scratch (reduction partials, hand-off slots, parameter tables), its
# own 8-byte result slot.  The library's launches of a task are therefore ordered against that task's own ROCArray work, and two
# tasks never share scratch.  Same scheme as the Python mirror (bijectors.jl_amd/interface.py `_ctx_for`: contexts keyed by
# (device, stream)).  Nothing at module level holds "the" context: the registry below is only ever touched under its lock, and the
# per-task cache lives in task_local_storage.
mutable struct Context
    h::Ptr{Cvoid}
    device::Int
    stream::Ptr{Cvoid}
    lsum::ROCVector{Float64}      # the Σ logabsdetjac of a scalar-log-det call lands here (one slot, reused: run! reads it back before it returns)
    hsum::Vector{Float64}         # its host copy (one copyto! per scalar result, no allocation per call)
end
function Context(dev::Integer, stream)
    ccall((:bjx_version, libbjx), Cint, ()) == 100 || error("libbjx_hip.so: unexpected ABI version")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    sp = Ptr{Cvoid}(stream.stream)
    rc = ccall((:bjx_create, libbjx), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), dev, sp, h)
    rc == 0 || error("bjx_create failed with status $rc")
    c = Context(h[], Int(dev), sp, AMDGPU.zeros(Float64, 1), zeros(Float64, 1))
    finalizer(x -> ccall((:bjx_destroy, libbjx), Cint, (Ptr{Cvoid},), x.h), c)
    return c
end
const CONTEXTS_LOCK = ReentrantLock()
const CONTEXTS = Dict{Tuple{Int,Ptr{Cvoid}},Context}()      # guarded by CONTEXTS_LOCK; values are never shared between streams
function ctx()
    dev = AMDGPU.device_id(AMDGPU.device()) - 1
    st = AMDGPU.stream()                                    # task-local in AMDGPU.jl
    key = (Int(dev), Ptr{Cvoid}(st.stream))
    tls = task_local_storage()
    c = get(tls, :bjx_context, nothing)
    (c isa Context && c.device == key[1] && c.stream == key[2]) && return c      # fast path: no lock, no lookup
    c = lock(CONTEXTS_LOCK) do
        get!(() -> Context(dev, st), CONTEXTS, key)
    end
    tls[:bjx_context] = c
    return c
end
# a task that is done with the GPU hands its context back (the registry otherwise keeps one per stream ever seen)
function release_context!()
    c = get(task_local_storage(), :bjx_context, nothing)
    c isa Context || return nothing
    lock(CONTEXTS_LOCK) do
        delete!(CONTEXTS, (c.device, c.stream))
    end
    delete!(task_local_storage(), :bjx_context)
    return nothing
end

function check(rc::Cint, what)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:bjx_last_error, libbjx), Cstring, (Ptr{Cvoid},), ctx().h))
    rc == -1 && throw(ArgumentError("$what: $msg"))         # BJX_ERR_ARG
    rc == -2 && throw(DimensionMismatch("$what: $msg"))     # BJX_ERR_SHAPE
    error("$what: status $rc: $msg")                        # hipError_t / ncclResult_t / BJX_ERR_FINALIZE (asynchronous: repeat the call)
end

# Move THIS task's context onto another stream (rarely needed: `ctx()` already follows `AMDGPU.stream()`); the library orders the new
# stream after the work in flight on the old one (bjx_set_stream records an event).  Wait for the library's stream; verify the
# hand-off state (diagnostics); scratch held by the context; tuning switches.
function set_stream!(stream=AMDGPU.stream())
    c = ctx()
    sp = Ptr{Cvoid}(stream.stream)
    sp == c.stream && return nothing
    lock(CONTEXTS_LOCK) do
        haskey(CONTEXTS, (c.device, sp)) && error("set_stream!: another context already serves that stream")
        check(ccall((:bjx_set_stream, libbjx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, sp), "bjx_set_stream")
        delete!(CONTEXTS, (c.device, c.stream)); c.stream = sp; CONTEXTS[(c.device, sp)] = c
    end
    return nothing
end
synchronize() = check(ccall((:bjx_synchronize, libbjx), Cint, (Ptr{Cvoid},), ctx().h), "bjx_synchronize")
check_state() = check(ccall((:bjx_check_state, libbjx), Cint, (Ptr{Cvoid},), ctx().h), "bjx_check_state")
launch_count() = Int(ccall((:bjx_launch_count, libbjx), UInt64, ()))      # kernel launches of the library so far (helpers included): launches of a region = a difference
workspace_bytes() = Int(ccall((:bjx_workspace_bytes, libbjx), Csize_t, (Ptr{Cvoid},), ctx().h))
const BJX_OPT_COLLECTIVE_TIMEOUT_MS = Cint(2)
collective_timeout!(ms::Integer) = check(ccall((:bjx_set_option, libbjx), Cint, (Ptr{Cvoid}, Cint, Cint), ctx().h, BJX_OPT_COLLECTIVE_TIMEOUT_MS, Cint(ms)), "bjx_set_option")   # watchdog of synchronize()
# BJX_OPT_PARAM_EPOCH (include/bjx.h): a non-zero epoch lets the library keep what it derives from parameter arrays with a helper
# launch (the spline's LDS blob, the [A⁻¹ | logabsdet] factorisation behind a matrix `Scale`) while it is unchanged.  ROCArrays carry
# no write counter, so the default stays 0 (rebuild on every call: always correct — the same default as the Python mirror since
# round 5, so the default numbers of bench.py are what a Julia caller gets); code that knows when it updates its parameters calls
# `param_epoch!(step)` after each update (any different non-zero value) and keeps the tables in between — what bench.py's
# `cache_params` rows measure.
const BJX_OPT_PARAM_EPOCH = Cint(3)
param_epoch!(n::Integer) = check(ccall((:bjx_set_option, libbjx), Cint, (Ptr{Cvoid}, Cint, Cint), ctx().h, BJX_OPT_PARAM_EPOCH, Cint(n)), "bjx_set_option")
# who finishes Σ logabsdetjac: 2 (default) the sentinel hand-off inside the hot kernel, 1 the arrival ticket, 0 two follow-up launches
inkernel_finalize!(mode::Integer) = check(ccall((:bjx_set_option, libbjx), Cint, (Ptr{Cvoid}, Cint, Cint), ctx().h, BJX_OPT_INKERNEL_FINALIZE, Cint(mode)), "bjx_set_option")

dims(x::ROCVector) = (length(x), 1)
dims(x::ROCMatrix) = size(x)
devptr(x::ROCArray) = Ptr{Cvoid}(pointer(x))
devptr(::Nothing) = C_NULL
ondevice(::Type{T}, a::ROCArray{T}) where {T} = a
ondevice(::Type{T}, a) where {T} = ROCArray{T}(a)           # parameters may already live on the device

# ---------------------------------------------------------------- launch plans
struct Plan
    name::Symbol            # the C entry (error messages)
    outsize::Dims           # size of the transformed array
    ladj::Symbol            # the reference's log-det return shape: :column (T[batch]), :scalar (one number), :zero (Permute)
    batch::Int
    null_out::Bool          # the entry accepts out = C_NULL (log-det only, no store traffic)
    alias_ok::Bool          # out may alias the input (transform!(b, x) == transform!(b, x, x), interface.jl:175)
    flags::UInt32
    keep::Vector{Any}       # device arrays the launch reads: alive until the call has been enqueued
    launch::Function        # (out::Ptr{Cvoid}, lps::Ptr{Cvoid}, lsum::Ptr{Cvoid}, flags::UInt32) -> Cint
end

# run a plan.  out === nothing -> no transformed values wanted (C_NULL where the entry allows it, a scratch array otherwise);
# lps_into -> a ROCVector the per-column log-dets are ADDED to (BJX_ACCUMULATE: the `!` forms of interface.jl:199-218).
function run!(p::Plan, ::Type{T}, x, out; want_ladj::Bool=true, lps_into=nothing) where {T}
    o = out === nothing ? (p.null_out ? nothing : similar(x, T, p.outsize)) : out
    lps = !want_ladj || p.ladj !== :column ? nothing : (lps_into === nothing ? similar(x, T, p.batch) : lps_into)
    c = ctx()
    lsum = want_ladj && p.ladj === :scalar ? c.lsum : nothing        # the context's reused 8-byte slot: nothing allocated per call
    fl = p.flags | (lps_into === nothing ? UInt32(0) : BJX_ACCUMULATE)
    keep = p.keep
    GC.@preserve keep x o lps lsum check(p.launch(devptr(o), devptr(lps), devptr(lsum), fl), String(p.name))
    want_ladj || return nothing
    p.ladj === :column && return lps
    if p.ladj === :scalar                                  # the reference returns a host scalar here (§8a'): ONE copy, stream-ordered
        copyto!(c.hsum, c.lsum)
        return T(c.hsum[1])
    end
    return zero(T)
end

# the same launch with the per-column log-det vector instead of the reference's scalar (what a composition with flow layers adds up)
column_plan(p::Plan) = p.ladj === :scalar ? Plan(p.name, p.outsize, :column, p.batch, p.null_out, p.alias_ok, p.flags & ~BJX_REF_VECTOR_SCALE_LADJ, p.keep, p.launch) : p

# ---------------------------------------------------------------- F1: fused elementwise chains
# Walk `outer ∘ inner` into application order; nothing => not fusable, use the generic method.
ops(b::Elementwise{typeof(exp)}, T, keep) = [BjxOp(Int32(OP_EXP), 0, 0, 0, C_NULL, C_NULL)]
ops(b::Elementwise{typeof(log)}, T, keep) = [BjxOp(Int32(OP_LOG), 0, 0, 0, C_NULL, C_NULL)]
# (inverse(elementwise(exp)) IS elementwise(log), inverse(Shift(a)) IS Shift(-a), inverse(LeakyReLU(α)) IS LeakyReLU(1/α),
#  inverse(SignFlip()) IS SignFlip() and inverse(f ∘ g) IS inverse(g) ∘ inverse(f) — shift.jl:12, leaky_relu.jl:16, ordered.jl:4,
#  InverseFunctions — so `Inverse{…}` wrappers exist only for Scale, Logit and TruncatedBijector.)
function param_op(kind, a, T, keep, b=nothing)
    if a isa Real && (b === nothing || b isa Real)
        return BjxOp(Int32(kind), 1, Float64(a), b === nothing ? 0.0 : Float64(b), C_NULL, C_NULL)
    end
    n = a isa Real ? length(b) : length(a)
    va = ondevice(T, a isa Real ? fill(T(a), n) : a); push!(keep, va)
    vb = b === nothing ? nothing : ondevice(T, b isa Real ? fill(T(b), n) : b)
    vb === nothing || push!(keep, vb)
    return BjxOp(Int32(kind), n, 0, 0, devptr(va), devptr(vb))
end
ops(b::Shift{<:Union{Real,AbstractVector}}, T, keep) = [param_op(OP_SHIFT, b.a, T, keep)]
ops(b::Scale{<:Union{Real,AbstractVector}}, T, keep) = [param_op(OP_SCALE, b.a, T, keep)]
ops(b::Inverse{<:Scale{<:Union{Real,AbstractVector}}}, T, keep) = [param_op(OP_SCALE_INV, b.orig.a, T, keep)]
ops(b::Logit, T, keep) = [param_op(OP_LOGIT, b.a, T, keep, b.b)]
ops(b::Inverse{<:Logit}, T, keep) = [param_op(OP_LOGIT_INV, b.orig.a, T, keep, b.orig.b)]
ops(b::LeakyReLU, T, keep) = [param_op(OP_LEAKY_RELU, b.α, T, keep)]
ops(b::TruncatedBijector, T, keep) = [param_op(OP_TRUNCATED, b.lb, T, keep, b.ub)]
ops(b::Inverse{<:TruncatedBijector}, T, keep) = [param_op(OP_TRUNCATED_INV, b.orig.lb, T, keep, b.orig.ub)]
ops(b::Bijectors.SignFlip, T, keep) = [BjxOp(Int32(OP_SIGNFLIP), 0, 0, 0, C_NULL, C_NULL)]
ops(::typeof(identity), T, keep) = BjxOp[]
function ops(b::ComposedFunction, T, keep)            # inner first (composed.jl:4)
    i, o = ops(b.inner, T, keep), ops(b.outer, T, keep)
    (i === nothing || o === nothing) && return nothing
    return vcat(i, o)
end
ops(b, T, keep) = nothing

const ElementwiseLeaf = Union{Elementwise{typeof(exp)},Elementwise{typeof(log)},Shift{<:Union{Real,AbstractVector}},
    Scale{<:Union{Real,AbstractVector}},Logit,LeakyReLU,TruncatedBijector,Bijectors.SignFlip}
const FusableLeaf = Union{ElementwiseLeaf,Inverse{<:Scale{<:Union{Real,AbstractVector}}},Inverse{<:Logit},Inverse{<:TruncatedBijector}}   # one op each
const Fusable = Union{FusableLeaf,ComposedFunction}

# C-level launch plans (include/bjx.h "plans"; VERDICT r05 "do this" #6): what a sampler calls on every log-density evaluation is the
# SAME chain on a small (param_dim x n_chains) array (src/vector/product/fill.jl:146-165, 192-213) — walking `outer ∘ inner`, uploading
# nothing but still allocating the op vector and re-validating it costs more than the 5-15 us kernel.  The op list of a (bijector,
# element type, height, flags) is marshalled ONCE into a `bjx_plan` kept by the CONTEXT of the calling task (a plan belongs to its
# context); a call is then bjx_plan_run(plan, x, y, lps, lsum, C_NULL, batch).  The plan holds parameter POINTERS (`keep` keeps the
# arrays alive), never values: parameters rewritten in place are seen by the next run.
mutable struct CPlan
    h::Ptr{Cvoid}
    keep::Vector{Any}
end
const CPLANS_LOCK = ReentrantLock()
const CPLANS = Dict{Tuple{Ptr{Cvoid},UInt,DataType,Int,UInt32},CPlan}()         # (context, objectid(bijector), T, rows, flags); guarded by CPLANS_LOCK
function chain_cplan(b, ::Type{T}, o::Vector{BjxOp}, keep, d::Int, flags::UInt32) where {T}
    c = ctx()
    key = (c.h, objectid(b), T, d, flags)
    lock(CPLANS_LOCK) do
        get!(CPLANS, key) do
            hp = Ref{Ptr{Cvoid}}(C_NULL)
            GC.@preserve o keep check(ccall((:bjx_plan_chain, libbjx), Cint, (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Int64, UInt32, Ptr{Ptr{Cvoid}}),
                                            c.h, dtype(T), o, length(o), d, flags, hp), "bjx_plan_chain")
            cp = CPlan(hp[], copy(keep))
            finalizer(x -> ccall((:bjx_plan_destroy, libbjx), Cint, (Ptr{Cvoid},), x.h), cp)
            cp
        end
    end
end
# the two structured bijectors a Dirichlet / ordered model variable uses on every evaluation
function structured_cplan(kind::Cint, inv::Bool, ::Type{T}, d::Int, flags::UInt32) where {T}
    c = ctx()
    key = (c.h, UInt(kind) * UInt(2) + UInt(inv), T, d, flags)
    lock(CPLANS_LOCK) do
        get!(CPLANS, key) do
            hp = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:bjx_plan_structured, libbjx), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Int64, UInt32, Ptr{Ptr{Cvoid}}),
                        c.h, dtype(T), kind, Cint(inv), d, flags, hp), "bjx_plan_structured")
            cp = CPlan(hp[], Any[])
            finalizer(x -> ccall((:bjx_plan_destroy, libbjx), Cint, (Ptr{Cvoid},), x.h), cp)
            cp
        end
    end
end
const BJX_PLAN_SIMPLEX, BJX_PLAN_ORDERED = Cint(2), Cint(3)
plan_run(cp::CPlan, px, out, lps, lsum, n) = ccall((:bjx_plan_run, libbjx), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64), cp.h, px, out, lps, lsum, C_NULL, n)

function plan(b::Fusable, x::ROCArray{T}) where {T<:BjxFloat}
    keep = Any[]
    o = ops(b, T, keep)
    (o === nothing || length(o) > 8) && return nothing          # BJX_MAX_OPS; arbitrary Transforms inside ∘: generic path
    d, n = x isa ROCVecOrMat ? dims(x) : (length(x), 1)          # higher-rank arrays: elementwise over everything
    h = ctx().h; px = devptr(x); push!(keep, o)
    base = BJX_REF_VECTOR_SCALE_LADJ
    cp = chain_cplan(b, T, o, keep, Int(d), base); push!(keep, cp)
    launch = (out, lps, lsum, fl) -> fl == base ? plan_run(cp, px, out, lps, lsum, n) :          # the plan's flags; anything else (BJX_ACCUMULATE, per-column shape) marshals per call
        ccall((:bjx_chain, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), o, length(o), px, out, lps, lsum, d, n, fl)
    # the reference returns ONE scalar for elementwise bijectors (§8a'), with scale.jl:31-32's no-xN quirk
    return Plan(:bjx_chain, size(x), :scalar, n, true, true, base, keep, launch)
end

# ---------------------------------------------------------------- structured bijectors
# OrderedBijector, ordered.jl:22-80 (per-column log-det vector)
function plan_ordered(inv::Bool, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    d, n = dims(x); h = ctx().h; px = devptr(x)
    cp = structured_cplan(BJX_PLAN_ORDERED, inv, T, Int(d), UInt32(0))
    launch = (out, lps, lsum, fl) -> fl == UInt32(0) ? plan_run(cp, px, out, lps, lsum, n) : ccall((:bjx_ordered, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), px, out, lps, lsum, d, n, fl)
    return Plan(:bjx_ordered, size(x), x isa ROCVector ? :scalar : :column, n, false, false, UInt32(0), Any[], launch)
end
plan(::OrderedBijector, y::ROCVecOrMat{<:BjxFloat}) = plan_ordered(false, y)
plan(::Inverse{OrderedBijector}, x::ROCVecOrMat{<:BjxFloat}) = plan_ordered(true, x)

# SimplexBijector, simplex.jl:14,28-143 (scalar log-det summed over the columns, :141-143)
function plan_simplex(inv::Bool, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    r, n = dims(x); K = inv ? r + 1 : r
    h = ctx().h; px = devptr(x)
    osz = x isa ROCVector ? (inv ? (K,) : (K - 1,)) : (inv ? (K, n) : (K - 1, n))
    cp = structured_cplan(BJX_PLAN_SIMPLEX, inv, T, Int(r), UInt32(0))           # `dim` of a structured plan = rows of the INPUT
    launch = (out, lps, lsum, fl) -> fl == UInt32(0) ? plan_run(cp, px, out, lps, lsum, n) : ccall((:bjx_simplex, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), px, out, lps, lsum, K, n, fl)
    return Plan(:bjx_simplex, osz, :scalar, n, !inv, false, UInt32(0), Any[], launch)
end
plan(::SimplexBijector, x::ROCVecOrMat{<:BjxFloat}) = plan_simplex(false, x)
plan(::Inverse{<:SimplexBijector}, y::ROCVecOrMat{<:BjxFloat}) = plan_simplex(true, y)

# VecCholeskyBijector, corr.jl:227-254: the reference handles ONE Cholesky factor; here a K x K x N ROCArray is a batch of N
# dense factors (upper filled / lower zero for :U, transposed for :L) and the log-det is the per-sample vector; one
# ROCMatrix (K x K) is the N = 1 case with a scalar log-det, like the reference.
uplo(b::VecCholeskyBijector) = Cint(b.mode === :U ? 'U' : 'L')
function plan(b::VecCholeskyBijector, W::ROCArray{T}) where {T<:BjxFloat}
    ndims(W) in (2, 3) || return nothing
    K, n = size(W, 1), size(W, 3)
    size(W, 2) == K || throw(DimensionMismatch("VecCholeskyBijector: expected square factors, got $(size(W)[1:2])"))
    h = ctx().h; pw = devptr(W); ul = uplo(b); m = (K * (K - 1)) ÷ 2
    launch = (out, lps, lsum, fl) -> ccall((:bjx_vec_cholesky, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(0), ul, pw, out, lps, lsum, K, n, fl)
    return Plan(:bjx_vec_cholesky, ndims(W) == 2 ? (m,) : (m, n), ndims(W) == 2 ? :scalar : :column, n, false, false, UInt32(0), Any[], launch)
end
function plan(ib::Inverse{VecCholeskyBijector}, y::ROCVecOrMat{T}) where {T<:BjxFloat}
    m, n = dims(y); K = Bijectors._triu1_dim_from_length(m)
    h = ctx().h; py = devptr(y); ul = uplo(ib.orig)
    launch = (out, lps, lsum, fl) -> ccall((:bjx_vec_cholesky, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(1), ul, py, out, lps, lsum, K, n, fl)
    return Plan(:bjx_vec_cholesky, y isa ROCVector ? (K, K) : (K, K, n), y isa ROCVector ? :scalar : :column, n, true, false, UInt32(0), Any[], launch)
end

# PlanarLayer, planar_layer.jl:65-127,160-185.  One Julia object per layer; a RUN of layers of a composition (`l8 ∘ … ∘ l1`,
# docs/src/flows.md:115) is the planner's `PlanarRun` below: ONE bjx_planar launch with n_layers = the run.
# n device vectors of length `len` -> the layer-major table the entry takes, in one launch (bjx_pack_vectors)
function pack(::Type{T}, vs::Vector, len::Integer) where {T}
    n = length(vs)
    dst = ROCArray{T}(undef, n * len)
    ptrs = Ptr{Cvoid}[devptr(v) for v in vs]
    GC.@preserve vs dst ptrs check(ccall((:bjx_pack_vectors, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Int64, Ptr{Cvoid}),
        ctx().h, dtype(T), Cint(n), ptrs, Int64(len), devptr(dst)), "bjx_pack_vectors")
    return dst
end
planar_b(::Type{T}, l::PlanarLayer) where {T} = ondevice(T, l.b isa Real ? [l.b] : l.b)
function planar_tables(::Type{T}, layers, d::Integer) where {T}       # layers in application order -> (w, u, b) tables
    for l in layers
        length(l.w) == d || throw(DimensionMismatch("PlanarLayer of dimension $(length(l.w)) applied to $d rows"))
    end
    if length(layers) == 1
        l = layers[1]
        return ondevice(T, l.w), ondevice(T, l.u), planar_b(T, l)
    end
    return pack(T, [ondevice(T, l.w) for l in layers], d), pack(T, [ondevice(T, l.u) for l in layers], d),
           pack(T, [planar_b(T, l) for l in layers], 1)
end
function plan_planar(layers, inv::Bool, z::ROCVecOrMat{T}, flags::UInt32=UInt32(0)) where {T<:BjxFloat}
    d, n = dims(z)
    w, u, b = planar_tables(T, layers, d)
    h = ctx().h; pz = devptr(z); pw, pu, pb = devptr(w), devptr(u), devptr(b); nl = Cint(length(layers))
    launch = (out, lps, lsum, fl) -> ccall((:bjx_planar, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), pw, pu, pb, nl, pz, out, lps, lsum, d, n, fl)
    return Plan(:bjx_planar, size(z), :column, n, true, false, flags, Any[w, u, b], launch)
end
plan(flow::PlanarLayer, z::ROCVecOrMat{<:BjxFloat}) = plan_planar(PlanarLayer[flow], false, z)
plan(ib::Inverse{<:PlanarLayer}, y::ROCVecOrMat{<:BjxFloat}) = plan_planar(PlanarLayer[ib.orig], true, y)

# A maximal run of PlanarLayer stages of a composition.  `layers` is the FORWARD application order of the underlying flow;
# inv = true: the run is inverse(l_a), inverse(l_b), … applied in that order = inverse(l_a ∘ l_b ∘ …), whose forward flow applies
# … , l_b, l_a — so `layers` holds the stages' layers REVERSED and bjx_planar (inverse = 1) undoes them last to first.
struct PlanarRun
    layers::Vector{PlanarLayer}
    inv::Bool
end
plan(r::PlanarRun, z::ROCVecOrMat{<:BjxFloat}) = plan_planar(r.layers, r.inv, z)

# RadialLayer, radial_layer.jl:43-129
function plan_radial(flow::RadialLayer, inv::Bool, z::ROCVecOrMat{T}) where {T<:BjxFloat}
    d, n = dims(z); length(flow.z_0) == d || throw(DimensionMismatch("RadialLayer of dimension $(length(flow.z_0)) applied to $d rows"))
    a, be, z0 = ondevice(T, flow.α_ isa Real ? [flow.α_] : flow.α_), ondevice(T, flow.β isa Real ? [flow.β] : flow.β), ondevice(T, flow.z_0)
    h = ctx().h; pz = devptr(z); pa, pbe, pz0 = devptr(a), devptr(be), devptr(z0)
    launch = (out, lps, lsum, fl) -> ccall((:bjx_radial, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), pa, pbe, pz0, pz, out, lps, lsum, d, n, fl)
    return Plan(:bjx_radial, size(z), :column, n, false, false, UInt32(0), Any[a, be, z0], launch)
end
plan(flow::RadialLayer, z::ROCVecOrMat{<:BjxFloat}) = plan_radial(flow, false, z)
plan(ib::Inverse{<:RadialLayer}, y::ROCVecOrMat{<:BjxFloat}) = plan_radial(ib.orig, true, y)

# InvertibleBatchNorm, normalise.jl:41-88.  Eval mode: bjx_batchnorm (both directions).  Training mode (istraining(), forward
# only — the reference asserts the same, :75): bjx_batchnorm_train; bn.m / bn.v must be ROCVectors (updated on the device,
# and when the context has a communicator the statistics are all-reduced over the ranks, SURVEY.md §8e).
function plan_batchnorm(bn::InvertibleBatchNorm, inv::Bool, x::ROCMatrix{T}) where {T<:BjxFloat}
    d, n = size(x)
    d == length(bn.b) || error("InvertibleBatchNorm expected $(length(bn.b)) channels, got $d")          # normalise.jl:43-45
    b, logs = ondevice(T, bn.b), ondevice(T, bn.logs)
    h = ctx().h; px = devptr(x); pb, pl = devptr(b), devptr(logs); eps = Float64(bn.eps)
    if Bijectors.istraining()
        inv && error("`with_logabsdet_jacobian(::Inverse{InvertibleBatchNorm})` is only available in test mode.")
        (bn.m isa ROCVector{T} && bn.v isa ROCVector{T}) || return nothing       # host statistics: the generic method updates them
        pm, pv = devptr(bn.m), devptr(bn.v); mtm = Float64(bn.mtm)
        launch = (out, lps, lsum, fl) -> ccall((:bjx_batchnorm_train, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            h, dtype(T), pb, pl, pm, pv, eps, mtm, px, out, lps, lsum, d, n, fl)
        return Plan(:bjx_batchnorm_train, size(x), :column, n, false, false, UInt32(0), Any[b, logs, bn.m, bn.v], launch)
    end
    m, v = ondevice(T, bn.m), ondevice(T, bn.v); pm, pv = devptr(m), devptr(v)
    launch = (out, lps, lsum, fl) -> ccall((:bjx_batchnorm, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), pb, pl, pm, pv, eps, px, out, lps, lsum, d, n, fl)
    return Plan(:bjx_batchnorm, size(x), :column, n, false, false, UInt32(0), Any[b, logs, m, v], launch)
end
plan(bn::InvertibleBatchNorm, x::ROCMatrix{<:BjxFloat}) = plan_batchnorm(bn, false, x)
plan(ib::Inverse{<:InvertibleBatchNorm}, y::ROCMatrix{<:BjxFloat}) = plan_batchnorm(ib.orig, true, y)

# RationalQuadraticSpline{<:AbstractMatrix} (rational_quadratic_spline.jl:173-178,227-233,304-309,363-367): the reference has the
# single-column method (scalar log-det); a ROCMatrix of columns returns the per-column vector.
function plan_rqs(b::RationalQuadraticSpline{<:AbstractMatrix}, inv::Bool, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    d, n = dims(x)
    size(b.widths, 1) == d || throw(DimensionMismatch("RationalQuadraticSpline with $(size(b.widths, 1)) rows applied to $d rows"))
    w, hh, dd = ondevice(T, b.widths), ondevice(T, b.heights), ondevice(T, b.derivatives)
    h = ctx().h; px = devptr(x); pw, ph, pd = devptr(w), devptr(hh), devptr(dd); K1 = Cint(size(b.widths, 2))
    launch = (out, lps, lsum, fl) -> ccall((:bjx_rqs, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), pw, ph, pd, K1, px, out, lps, lsum, d, n, fl)
    return Plan(:bjx_rqs, size(x), x isa ROCVector ? :scalar : :column, n, false, false, UInt32(0), Any[w, hh, dd], launch)
end
plan(b::RationalQuadraticSpline{<:AbstractMatrix}, x::ROCVecOrMat{<:BjxFloat}) = plan_rqs(b, false, x)
plan(ib::Inverse{<:RationalQuadraticSpline{<:AbstractMatrix}}, y::ROCVecOrMat{<:BjxFloat}) = plan_rqs(ib.orig, true, y)

# The `B` constructor on the device (rational_quadratic_spline.jl:109-123): raw (dim, K), (dim, K), (dim, K-1) -> knots (dim, K+1)
function rqs_from_raw(raw_w::ROCMatrix{T}, raw_h::ROCMatrix{T}, raw_d::ROCMatrix{T}, B::Real) where {T<:BjxFloat}
    d, K = size(raw_w)
    w, hh, dd = similar(raw_w, d, K + 1), similar(raw_w, d, K + 1), similar(raw_w, d, K + 1)
    GC.@preserve raw_w raw_h raw_d w hh dd check(ccall((:bjx_rqs_params, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ctx().h, dtype(T), devptr(raw_w), devptr(raw_h), devptr(raw_d), Cint(K), d, Float64(B), devptr(w), devptr(hh), devptr(dd)), "bjx_rqs_params")
    return RationalQuadraticSpline(w, hh, dd)
end

# Permute, permute.jl:152-157: `A * x` for a permutation matrix = a row gather; src[i] = column of the 1 in row i (0-based).
# inverse(::Permute) is a Permute again (:153), so one plan covers both directions.  Bit-exact; log-det zero.
gather_list(A) = Int32[findfirst(!iszero, view(A, i, :)) - 1 for i in 1:size(A, 1)]
function plan(b::Permute, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    d, n = dims(x); size(b.A) == (d, d) || throw(DimensionMismatch("Permute of size $(size(b.A)) applied to $d rows"))
    src = ROCArray{Int32}(gather_list(b.A))
    h = ctx().h; px = devptr(x); ps = Ptr{Int32}(pointer(src))
    launch = (out, lps, lsum, fl) -> ccall((:bjx_permute, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        h, dtype(T), ps, px, out, d, n)
    return Plan(:bjx_permute, size(x), :zero, n, false, false, UInt32(0), Any[src], launch)
end

# Coupling with a PartitionMask, coupling.jl:125-134,206-259.  θ is an arbitrary Julia closure: it runs HERE, on the x₂ rows
# (gathered on the device), and must return one of the laws the ABI carries — Shift, Scale, Shift ∘ Scale (parameters Real,
# length-n₁ vectors shared by every column, or n₁ x N matrices) or a RationalQuadraticSpline with n₁-row knot matrices.
# Anything else falls back to the reference's generic method.
mask_rows(A) = Int32.(SparseArrays.rowvals(SparseArrays.sparse(A)) .- 1)       # column j of A_k has its 1 in row idx[j]
function gather_rows(x::ROCVecOrMat{T}, idx0::Vector{Int32}) where {T}      # x[idx .+ 1, :] through bjx_stacked_ld (identity segments)
    d, n = dims(x); m = length(idx0)
    out = x isa ROCVector ? similar(x, m) : similar(x, m, n)
    m == 0 && return out
    segs = BjxSegment[]; lo = 1
    for j in 2:(m + 1)                                                         # maximal runs of consecutive source rows
        if j > m || idx0[j] != idx0[j - 1] + 1
            push!(segs, BjxSegment(idx0[lo], lo - 1, j - lo, 0, 0, ntuple(_ -> NOOP, 4))); lo = j
        end
    end
    GC.@preserve x out check(ccall((:bjx_stacked_ld, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), segs, length(segs), devptr(x), d, devptr(out), m, C_NULL, C_NULL, m, n, UInt32(0)), "bjx_stacked_ld")
    return out
end
# law -> (kind, scale, shift | knots) ; nothing = not carried by the ABI
coupling_law(l::Shift) = (:affine, nothing, l.a)
coupling_law(l::Scale) = (:affine, l.a, nothing)
coupling_law(l::ComposedFunction{<:Shift,<:Scale}) = (:affine, l.inner.a, l.outer.a)
coupling_law(l::RationalQuadraticSpline{<:AbstractMatrix}) = (:rqs, l, nothing)
coupling_law(l) = nothing
function affine_arg(::Type{T}, a, n1, n, keep) where {T}        # -> (device pointer, one value per row shared by every column?)
    a === nothing && return C_NULL, false
    v = ondevice(T, a isa Real ? fill(T(a), n1) : a); push!(keep, v)
    length(v) == n1 * n && return devptr(v), false           # T[n1, batch]
    length(v) == n1 || throw(DimensionMismatch("coupling law parameter of length $(length(v)) for $n1 transformed rows"))
    return devptr(v), true                                   # T[n1]: BJX_COUPLING_*_BCAST, nothing is expanded
end
function plan_coupling(cl::Coupling, inv::Bool, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    d, n = dims(x)
    idx1, idx2 = mask_rows(cl.mask.A_1), mask_rows(cl.mask.A_2)
    law = coupling_law(cl.θ(gather_rows(x, idx2)))                               # θ(x₂): host closure, device arrays
    law === nothing && return nothing
    n1 = length(idx1); keep = Any[]
    di = ROCArray{Int32}(idx1); push!(keep, di)
    h = ctx().h; px = devptr(x); pi1 = Ptr{Int32}(pointer(di))
    if law[1] === :affine
        ps, sb = affine_arg(T, law[2], n1, n, keep)
        pt, tb = affine_arg(T, law[3], n1, n, keep)
        cf = (sb ? BJX_COUPLING_SCALE_BCAST : UInt32(0)) | (tb ? BJX_COUPLING_SHIFT_BCAST : UInt32(0))
        launch = (out, lps, lsum, fl) -> ccall((:bjx_coupling_affine, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Int32}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            h, dtype(T), Cint(inv), pi1, n1, ps, pt, px, out, lps, lsum, d, n, fl)
        return Plan(:bjx_coupling_affine, size(x), x isa ROCVector ? :scalar : :column, n, false, false, cf, keep, launch)
    end
    sp = law[2]
    size(sp.widths, 1) == n1 || throw(DimensionMismatch("spline coupling law with $(size(sp.widths, 1)) rows for $n1 transformed rows"))
    w, hh, dd = ondevice(T, sp.widths), ondevice(T, sp.heights), ondevice(T, sp.derivatives); push!(keep, w, hh, dd)
    pw, ph, pd = devptr(w), devptr(hh), devptr(dd); K1 = Cint(size(sp.widths, 2))
    launch = (out, lps, lsum, fl) -> ccall((:bjx_coupling_rqs, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Int32}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), pi1, n1, pw, ph, pd, K1, px, out, lps, lsum, d, n, fl)
    return Plan(:bjx_coupling_rqs, size(x), x isa ROCVector ? :scalar : :column, n, false, false, UInt32(0), keep, launch)
end
plan(cl::Coupling, x::ROCVecOrMat{<:BjxFloat}) = plan_coupling(cl, false, x)
plan(icl::Inverse{<:Coupling}, y::ROCVecOrMat{<:BjxFloat}) = plan_coupling(icl.orig, true, y)     # θ(y₂) = θ(x₂): the rows pass through

# ---------------------------------------------------------------- matrix-variate constraint bijectors (SURVEY.md §8f f-4)
# corr.jl:64-162, pd.jl:1-60.  The reference defines them for ONE matrix; a K x K x N ROCArray is a batch of N samples
# and returns the per-sample log-det vector.  (A single ROCMatrix is the N = 1 case and returns the scalar.)
const MatrixKinds = Union{VecCorrBijector,CorrBijector,PDBijector,PDVecBijector}
packed(b) = b isa Union{VecCorrBijector,PDVecBijector}
packed_length(::VecCorrBijector, K) = (K * (K - 1)) ÷ 2                      # corr.jl:150-154
packed_length(::PDVecBijector, K) = (K * (K + 1)) ÷ 2                        # pd.jl:50-54
unpacked_dim(::VecCorrBijector, m) = Bijectors._triu1_dim_from_length(m)
unpacked_dim(::PDVecBijector, m) = Bijectors._triu_dim_from_length(m)
# one literal ccall per entry (the static ABI check reads them)
matrix_launch(::VecCorrBijector, h, dt, inv, pin, K, n) = (out, lps, lsum, fl) -> ccall((:bjx_vec_corr, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32), h, dt, inv, pin, out, lps, lsum, K, n, fl)
matrix_launch(::CorrBijector, h, dt, inv, pin, K, n) = (out, lps, lsum, fl) -> ccall((:bjx_corr, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32), h, dt, inv, pin, out, lps, lsum, K, n, fl)
matrix_launch(::PDBijector, h, dt, inv, pin, K, n) = (out, lps, lsum, fl) -> ccall((:bjx_pd, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32), h, dt, inv, pin, out, lps, lsum, K, n, fl)
matrix_launch(::PDVecBijector, h, dt, inv, pin, K, n) = (out, lps, lsum, fl) -> ccall((:bjx_pd_vec, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32), h, dt, inv, pin, out, lps, lsum, K, n, fl)
entry_name(::VecCorrBijector) = :bjx_vec_corr
entry_name(::CorrBijector) = :bjx_corr
entry_name(::PDBijector) = :bjx_pd
entry_name(::PDVecBijector) = :bjx_pd_vec
function plan(b::MatrixKinds, X::ROCArray{T}) where {T<:BjxFloat}
    ndims(X) in (2, 3) || return nothing
    K, n = size(X, 1), size(X, 3); single = ndims(X) == 2
    size(X, 2) == K || throw(DimensionMismatch("sizes should be equal; received $(size(X)[1:2])"))
    osz = packed(b) ? (single ? (packed_length(b, K),) : (packed_length(b, K), n)) : size(X)
    return Plan(entry_name(b), osz, single ? :scalar : :column, n, true, false, UInt32(0), Any[],
                matrix_launch(b, ctx().h, dtype(T), Cint(0), devptr(X), K, n))
end
function plan(ib::Inverse{<:MatrixKinds}, Y::ROCArray{T}) where {T<:BjxFloat}
    b = ib.orig
    if packed(b)
        ndims(Y) in (1, 2) || return nothing
        m, n = dims(Y); K = unpacked_dim(b, m); single = Y isa ROCVector
        osz = single ? (K, K) : (K, K, n)
    else
        ndims(Y) in (2, 3) || return nothing
        K, n = size(Y, 1), size(Y, 3); single = ndims(Y) == 2; osz = size(Y)
    end
    return Plan(entry_name(b), osz, single ? :scalar : :column, n, true, false, UInt32(0), Any[],
                matrix_launch(b, ctx().h, dtype(T), Cint(1), devptr(Y), K, n))
end

# Scale with a matrix parameter (scale.jl:14,17,35-36): a * x, a \ y, logabsdet(a) ONCE whatever the number of columns
function plan_scale_matrix(a::ROCMatrix{T}, inv::Bool, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    d, n = dims(x)
    size(a) == (d, d) || throw(DimensionMismatch("Scale with a $(size(a)) matrix applied to $d rows"))
    h = ctx().h; pa = devptr(a); px = devptr(x)
    launch = (out, lps, lsum, fl) -> ccall((:bjx_scale_matrix, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        h, dtype(T), Cint(inv), pa, px, out, lps, lsum, d, n, fl)
    return Plan(:bjx_scale_matrix, size(x), :scalar, n, true, false, BJX_REF_VECTOR_SCALE_LADJ, Any[a], launch)
end
plan(b::Scale{<:ROCMatrix{T}}, x::ROCVecOrMat{T}) where {T<:BjxFloat} = plan_scale_matrix(b.a, false, x)
# parameter pullback of the matrix Scale (ext/BijectorsReverseDiffExt.jl:72-115): ā = sign·(g xᵀ + Σℓ̄ · a⁻ᵀ) in one library entry — the batch-summed outer
# product on the matrix cores, a⁻ᵀ from the library's own factorisation.  forward: g = ȳ, x = the input, sign = +1; inverse (x = a \ y): g = the input
# cotangent a⁻ᵀx̄, x = a \ y, sign = −1.  ℓ̄ = nothing: no log-det cotangent.
function scale_matrix_vjp_params(a::ROCMatrix{T}, g::ROCMatrix{T}, x::ROCMatrix{T}, ℓ̄::Union{Nothing,ROCVector{T}}, sign::Real) where {T<:BjxFloat}
    d, n = size(x)
    (size(a) == (d, d) && size(g) == (d, n)) || throw(DimensionMismatch("scale_matrix_vjp_params: a $(size(a)), g $(size(g)), x $(size(x))"))
    ā = similar(a)
    GC.@preserve a g x ℓ̄ ā check(ccall((:bjx_scale_matrix_vjp_params, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), devptr(a), devptr(g), devptr(x), devptr(ℓ̄), Cdouble(sign), devptr(ā), d, n), "bjx_scale_matrix_vjp_params")
    return ā
end
plan(ib::Inverse{<:Scale{<:ROCMatrix{T}}}, y::ROCVecOrMat{T}) where {T<:BjxFloat} = plan_scale_matrix(ib.orig.a, true, y)

# ---------------------------------------------------------------- the six interface methods, once (src/interface.jl:156-218)
const Structured = Union{OrderedBijector,Inverse{OrderedBijector},SimplexBijector,Inverse{<:SimplexBijector},
    VecCholeskyBijector,Inverse{VecCholeskyBijector},PlanarLayer,Inverse{<:PlanarLayer},RadialLayer,Inverse{<:RadialLayer},
    InvertibleBatchNorm,Inverse{<:InvertibleBatchNorm},RationalQuadraticSpline{<:AbstractMatrix},
    Inverse{<:RationalQuadraticSpline{<:AbstractMatrix}},Permute,Coupling,Inverse{<:Coupling},
    MatrixKinds,Inverse{<:MatrixKinds},Scale{<:ROCMatrix},Inverse{<:Scale{<:ROCMatrix}},PlanarRun}
const Planned = Union{Fusable,Structured}

# the reference's return containers: NamedTuple for the flow layers (planar_layer.jl:102-110, radial_layer.jl:58-72), tuple otherwise
wrap(::Union{PlanarLayer,RadialLayer}, y, l) = (result=y, logabsdetjac=l)
wrap(b, y, l) = (y, l)

function with_logabsdet_jacobian(b::Planned, x::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    p === nothing && return invoke(with_logabsdet_jacobian, Tuple{typeof(b),Any}, b, x)      # not expressible across the ABI
    y = similar(x, T, p.outsize)
    return wrap(b, y, run!(p, T, x, y))
end
function transform(b::Planned, x::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    p === nothing && return invoke(transform, Tuple{typeof(b),Any}, b, x)
    y = similar(x, T, p.outsize)
    run!(p, T, x, y; want_ladj=false)                          # ladj_ps = ladj_sum = C_NULL: no reduction, no extra launch
    return y
end
function logabsdetjac(b::Planned, x::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    p === nothing && return invoke(logabsdetjac, Tuple{typeof(b),Any}, b, x)
    return run!(p, T, x, nothing)                              # out = C_NULL where the entry allows it: half the traffic
end
# transform!(b, x, y) (interface.jl:175-176); transform!(b, x) = transform!(b, x, x) only where the kernel may run in place
function transform!(b::Planned, x::ROCArray{T}, y::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    p === nothing && return invoke(transform!, Tuple{Any,Any,Any}, b, x, y)
    size(y) == p.outsize || throw(DimensionMismatch("transform!: output of size $(size(y)), expected $(p.outsize)"))
    if pointer(y) == pointer(x) && !p.alias_ok
        copyto!(y, transform(b, x))
    else
        run!(p, T, x, y; want_ladj=false)
    end
    return y
end
transform!(b::Planned, x::ROCArray{<:BjxFloat}) = transform!(b, x, x)
# logabsdetjac!(b, x, logjac) (interface.jl:199-200): returns logjac + logabsdetjac(b, x); a ROCVector accumulator of per-column
# log-dets is updated IN PLACE by the kernel (BJX_ACCUMULATE)
function logabsdetjac!(b::Planned, x::ROCArray{T}, logjac) where {T<:BjxFloat}
    p = plan(b, x)
    p === nothing && return invoke(logabsdetjac!, Tuple{Any,Any,Any}, b, x, logjac)
    if logjac isa ROCVector{T} && p.ladj === :column
        return run!(p, T, x, nothing; lps_into=logjac)
    end
    return logjac .+ run!(p, T, x, nothing)
end
logabsdetjac!(b::Planned, x::ROCArray{T}) where {T<:BjxFloat} = logabsdetjac!(b, x, zero(T))
# with_logabsdet_jacobian!(b, x, y, logjac) -> (y, logjac + new) (interface.jl:212-218)
function with_logabsdet_jacobian!(b::Planned, x::ROCArray{T}, y::ROCArray{T}, logjac) where {T<:BjxFloat}
    p = plan(b, x)
    p === nothing && return invoke(with_logabsdet_jacobian!, Tuple{Any,Any,Any,Any}, b, x, y, logjac)
    size(y) == p.outsize || throw(DimensionMismatch("with_logabsdet_jacobian!: output of size $(size(y)), expected $(p.outsize)"))
    if pointer(y) == pointer(x) && !p.alias_ok
        y_, l = with_logabsdet_jacobian(b, x)                  # (a NamedTuple destructures into its two values)
        return copyto!(y, y_), logjac .+ l
    end
    if logjac isa ROCVector{T} && p.ladj === :column
        return y, run!(p, T, x, y; lps_into=logjac)
    end
    return y, logjac .+ run!(p, T, x, y)
end
with_logabsdet_jacobian!(b::Planned, x::ROCArray{T}, y::ROCArray{T}) where {T<:BjxFloat} = with_logabsdet_jacobian!(b, x, y, zero(T))
with_logabsdet_jacobian!(b::Planned, x::ROCArray{<:BjxFloat}) = with_logabsdet_jacobian!(b, x, x)

# ---------------------------------------------------------------- composition planner (composed.jl:4-25; docs/src/flows.md:115)
# `l8 ∘ … ∘ l1` is a nest of ComposedFunctions; the reference applies it stage by stage: one pass over the batch per layer
# (8 224 B/sample for eight layers at dim 128).  Here the chain is flattened into application order and cut into PIECES, one launch each:
#   * a maximal run of fusable elementwise stages (<= 8 ops)                   -> bjx_chain       (a ComposedFunction of the run)
#   * a maximal run of PlanarLayers / of Inverse{<:PlanarLayer}s               -> bjx_planar, n_layers = the run  (PlanarRun; 1 028 B/sample)
#   * any other planned bijector                                               -> its own entry
#   * anything else (arbitrary Transforms)                                     -> the reference's generic method for that stage
stages(b::ComposedFunction) = vcat(stages(b.inner), stages(b.outer))          # inner first (composed.jl:4)
stages(::typeof(identity)) = Any[]
stages(b) = Any[b]
chain_of(run) = foldl((acc, st) -> st ∘ acc, run)                              # application order -> `last ∘ … ∘ first`
planar_kind(st) = st isa PlanarLayer ? 1 : (st isa Inverse{<:PlanarLayer} ? -1 : 0)
function pieces(b::ComposedFunction)
    st = stages(b); out = Any[]; run = Any[]; i = 1
    while i <= length(st)
        cur = st[i]
        if cur isa FusableLeaf                                                 # one op each: a run of at most 8 (BJX_MAX_OPS)
            length(run) == 8 && (push!(out, chain_of(run)); run = Any[])
            push!(run, cur); i += 1
            continue
        end
        isempty(run) || (push!(out, chain_of(run)); run = Any[])
        k = planar_kind(cur); j = i
        while k != 0 && j < length(st) && planar_kind(st[j + 1]) == k
            j += 1
        end
        if j > i
            push!(out, k == 1 ? PlanarRun(PlanarLayer[st[m] for m in i:j], false) : PlanarRun(PlanarLayer[st[m].orig for m in j:-1:i], true))
        else
            push!(out, cur)
        end
        i = j + 1
    end
    isempty(run) || push!(out, chain_of(run))
    return out
end
piece_wlj(pc, x) = (r = with_logabsdet_jacobian(pc, x); (r[1], r[2]))          # the flow layers return NamedTuples: by position
# Mixed shapes (ADVICE r04): a fused elementwise run returns ONE scalar for a matrix (the sum over all columns, §8a'), the flow
# layers / Ordered / BatchNorm a per-column vector — `scalar + vector` is a MethodError (in the reference too: composed.jl:12-15
# adds whatever the stages return), and broadcasting it would add the TOTAL to every column.  When any piece of a composition on a
# matrix is of the per-column kind, every planned piece is run with its per-column plan (`column_plan`), so the sum is the
# per-column log-det of the whole composition.
const PerColumnKind = Union{PlanarRun,PlanarLayer,Inverse{<:PlanarLayer},RadialLayer,Inverse{<:RadialLayer},InvertibleBatchNorm,Inverse{<:InvertibleBatchNorm},
    OrderedBijector,Inverse{OrderedBijector}}
wants_columns(pcs, x) = x isa ROCMatrix && any(pc -> pc isa PerColumnKind, pcs)
function piece_wlj_columns(pc, x::ROCMatrix{T}) where {T}
    p = pc isa Planned ? plan(pc, x) : nothing
    (p === nothing || p.ladj !== :scalar) && return piece_wlj(pc, x)
    y = similar(x, T, p.outsize)
    return y, run!(column_plan(p), T, x, y)
end
add_ladj(::Nothing, l) = l
add_ladj(a::Number, b::Number) = a + b
add_ladj(a::AbstractVector, b::AbstractVector) = a .+ b
add_ladj(a::Number, b::AbstractVector) = iszero(a) ? b : throw(ArgumentError("composition on a matrix: a stage returned the scalar log-det $(a) next to per-column log-dets; plan that stage per column"))
add_ladj(a::AbstractVector, b::Number) = add_ladj(b, a)

# the six interface methods for a composition that is NOT one fused elementwise chain: piece by piece
function with_logabsdet_jacobian(b::ComposedFunction, x::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    if p !== nothing                                                           # ONE fused elementwise chain
        y = similar(x, T, p.outsize)
        return y, run!(p, T, x, y)
    end
    pcs = pieces(b); cur = x; total = nothing
    col = wants_columns(pcs, x)
    for pc in pcs
        cur, l = col ? piece_wlj_columns(pc, cur) : piece_wlj(pc, cur)
        total = add_ladj(total, l)                                             # ChangesOfVariables' rule for ∘: ladj_inner + ladj_outer
    end
    return cur, total
end
function transform(b::ComposedFunction, x::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    if p !== nothing
        y = similar(x, T, p.outsize)
        run!(p, T, x, y; want_ladj=false)
        return y
    end
    return foldl((cur, pc) -> transform(pc, cur), pieces(b); init=x)           # composed.jl:4
end
function logabsdetjac(b::ComposedFunction, x::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    p === nothing || return run!(p, T, x, nothing)
    pcs = pieces(b); cur = x; total = nothing
    col = wants_columns(pcs, x)
    for pc in pcs[1:(end - 1)]                                                 # composed.jl:12-15: values of every piece but the last
        cur, l = col ? piece_wlj_columns(pc, cur) : piece_wlj(pc, cur)
        total = add_ladj(total, l)
    end
    l = col ? last(piece_wlj_columns(pcs[end], cur)) : logabsdetjac(pcs[end], cur)
    return add_ladj(total, l)
end
function transform!(b::ComposedFunction, x::ROCArray{T}, y::ROCArray{T}) where {T<:BjxFloat}
    p = plan(b, x)
    if p !== nothing
        size(y) == p.outsize || throw(DimensionMismatch("transform!: output of size $(size(y)), expected $(p.outsize)"))
        run!(p, T, x, y; want_ladj=false)
        return y
    end
    pcs = pieces(b)
    transform!(pcs[1], x, y)                                                   # composed.jl:7-10
    for pc in pcs[2:end]
        transform!(pc, y, y)
    end
    return y
end
transform!(b::ComposedFunction, x::ROCArray{<:BjxFloat}) = transform!(b, x, x)
function with_logabsdet_jacobian!(b::ComposedFunction, x::ROCArray{T}, y::ROCArray{T}, logjac) where {T<:BjxFloat}
    p = plan(b, x)
    if p !== nothing
        size(y) == p.outsize || throw(DimensionMismatch("with_logabsdet_jacobian!: output of size $(size(y)), expected $(p.outsize)"))
        return y, logjac .+ run!(p, T, x, y)
    end
    pcs = pieces(b)
    _, logjac = with_logabsdet_jacobian!(pcs[1], x, y, logjac)                 # composed.jl:22-25
    for pc in pcs[2:end]
        _, logjac = with_logabsdet_jacobian!(pc, y, y, logjac)
    end
    return y, logjac
end
with_logabsdet_jacobian!(b::ComposedFunction, x::ROCArray{T}, y::ROCArray{T}) where {T<:BjxFloat} = with_logabsdet_jacobian!(b, x, y, zero(T))
with_logabsdet_jacobian!(b::ComposedFunction, x::ROCArray{<:BjxFloat}) = with_logabsdet_jacobian!(b, x, x)
logabsdetjac!(b::ComposedFunction, x::ROCArray{T}, logjac) where {T<:BjxFloat} = logjac .+ logabsdetjac(b, x)
logabsdetjac!(b::ComposedFunction, x::ROCArray{T}) where {T<:BjxFloat} = logabsdetjac(b, x)

# `planar_stack(layers, z)`: the same single launch for code that already holds the layers as a vector (layer 1 applied first)
function planar_stack(layers::Vector{<:PlanarLayer}, z::ROCMatrix{T}; inverse::Bool=false, base_stdnormal::Bool=false, out=similar(z)) where {T<:BjxFloat}
    p = plan_planar(PlanarLayer[l for l in layers], inverse, z, base_stdnormal ? BJX_BASE_STDNORMAL : UInt32(0))
    return out, run!(p, T, z, out)
end

# Columnwise (interface.jl:41-78): `columnwise(f)` maps f over the columns and SUMS the log-dets — on a ROCMatrix of columns that
# is the batched kernel of f itself.  `f(x)` goes through eachcolmaphcat (Bijectors.jl:118): route that to the kernel too.
Bijectors.eachcolmaphcat(f::Planned, x::ROCMatrix{<:BjxFloat}) = transform(f, x)
transform(f::Columnwise{<:Planned}, x::ROCMatrix{<:BjxFloat}) = transform(f.x, x)
column_sum(l) = l isa ROCVector ? sum(l) : l                                   # per-column vector -> the scalar of interface.jl:75-77
logabsdetjac(f::Columnwise{<:Planned}, x::ROCMatrix{<:BjxFloat}) = column_sum(logabsdetjac(f.x, x))
function with_logabsdet_jacobian(f::Columnwise{<:Planned}, x::ROCMatrix{<:BjxFloat})
    y, l = piece_wlj(f.x, x)
    return y, column_sum(l)
end

# InvertibleBatchNorm in training mode on a batch sharded over ranks (normalise.jl:51-60; SURVEY.md §8e "Exception"), for
# hosts that own the collective: statistics of this rank's columns -> allreduce! (MPI.Allreduce!, or
# allreduce_logabsdetjac! after comm_init) -> update of the moving statistics and transform with the GLOBAL statistics.
function batchnorm_train!(bn::InvertibleBatchNorm, x::ROCMatrix{T}; allreduce! = identity) where {T}
    d, n = size(x)
    stats = AMDGPU.zeros(Float64, 2d + 1)
    GC.@preserve bn x stats check(ccall((:bjx_batchnorm_stats, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64),
        ctx().h, dtype(T), devptr(bn.m), devptr(x), Ptr{Cdouble}(pointer(stats)), d, n), "bjx_batchnorm_stats")
    allreduce!(stats)                                      # sum of the 2d+1 Float64 values over the ranks
    y = similar(x); lps = similar(x, T, n)
    GC.@preserve bn x y lps stats check(ccall((:bjx_batchnorm_train_apply, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), devptr(bn.b), devptr(bn.logs), devptr(bn.m), devptr(bn.v), Float64(bn.eps), Float64(bn.mtm),
        Ptr{Cdouble}(pointer(stats)), devptr(x), devptr(y), devptr(lps), C_NULL, d, n, UInt32(0)), "bjx_batchnorm_train_apply")
    return y, lps
end

# Pullback of the training-mode InvertibleBatchNorm (normalise.jl:51-60: batch mean and biased batch variance are functions of
# x).  (mean, var) = the batch statistics of the forward pass; moments = Σ_n ȳ, Σ_n ȳ·x, N of the WHOLE batch (row_moments on this
# rank's columns, then the host's all-reduce); ℓ̄sum = Σ_n ℓ̄.  -> (x̄, b̄, l̄ogs)
function batchnorm_train_pullback(bn::InvertibleBatchNorm, mean::ROCVector{T}, var::ROCVector{T}, x::ROCMatrix{T}, ȳ::ROCMatrix{T},
                                  ℓ̄sum::Real; allreduce! = identity) where {T<:BjxFloat}
    d, n = size(x)
    moments = row_moments(ȳ, x)
    allreduce!(moments)
    ls = ROCArray{Float64}([Float64(ℓ̄sum)])
    x̄ = similar(x); logs = ondevice(T, bn.logs); b̄ = similar(logs); l̄ogs = similar(logs)
    GC.@preserve logs mean var moments ls x ȳ x̄ b̄ l̄ogs check(ccall((:bjx_batchnorm_train_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), devptr(logs), devptr(mean), devptr(var), Float64(bn.eps), Ptr{Cdouble}(pointer(moments)), Ptr{Cdouble}(pointer(ls)),
        devptr(x), devptr(ȳ), devptr(x̄), devptr(b̄), devptr(l̄ogs), d, n), "bjx_batchnorm_train_vjp")
    return x̄, b̄, l̄ogs
end

# ---------------------------------------------------------------- Stacked (SURVEY.md §8f f-4)
# stacked.jl:27-252: every segment whose bijector is a fusable elementwise chain goes into ONE launch.
segment(rin, rout, o) = BjxSegment(first(rin) - 1, first(rout) - 1, length(rin), length(o), 0, ntuple(k -> k <= length(o) ? o[k] : NOOP, 4))
function elementwise_segments(sb::Stacked, T, keep)           # nothing when a segment is not an elementwise chain of <= 4 ops
    segs = BjxSegment[]
    for (b, rin, rout) in zip(sb.bs, sb.ranges_in, sb.ranges_out)
        o = ops(b, T, keep)
        (o === nothing || length(o) > 4 || length(rin) != length(rout)) && return nothing
        push!(segs, segment(rin, rout, o))
    end
    return segs
end
function with_logabsdet_jacobian(sb::Stacked, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    d, n = dims(x)
    sb.length_in == d || error("input length mismatch ($(sb.length_in) != $d)")          # stacked.jl:157
    keep = Any[]
    segs = elementwise_segments(sb, T, keep)
    segs === nothing && return stacked_structured(sb, x)                                   # Simplex / Ordered blocks: below
    y = similar(x)
    c = ctx()
    lsum = c.lsum                                              # the context's reused result slot (no device allocation per call)
    GC.@preserve keep x y lsum begin
        rc = ccall((:bjx_stacked, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, UInt32),
            c.h, dtype(T), segs, length(segs), devptr(x), devptr(y), C_NULL, Ptr{Cdouble}(pointer(lsum)), d, n, UInt32(0))
        check(rc, "bjx_stacked")
    end
    copyto!(c.hsum, c.lsum)
    return y, T(c.hsum[1])
end
transform(sb::Stacked, x::ROCVecOrMat{<:BjxFloat}) = first(with_logabsdet_jacobian(sb, x))
logabsdetjac(sb::Stacked, x::ROCVecOrMat{<:BjxFloat}) = last(with_logabsdet_jacobian(sb, x))
transform!(sb::Stacked, x::ROCVecOrMat{T}, y::ROCVecOrMat{T}) where {T<:BjxFloat} = copyto!(y, transform(sb, x))
logabsdetjac!(sb::Stacked, x::ROCVecOrMat{<:BjxFloat}, logjac) = logjac .+ logabsdetjac(sb, x)
function with_logabsdet_jacobian!(sb::Stacked, x::ROCVecOrMat{T}, y::ROCVecOrMat{T}, logjac) where {T<:BjxFloat}
    y_, l = with_logabsdet_jacobian(sb, x)
    return copyto!(y, y_), logjac .+ l
end
# NamedStacked (named_stacked.jl:42-199): a NamedTuple of bijectors for ProductNamedTupleDistribution samples.  Forward takes a
# NamedTuple of fields and returns ONE stacked array (:118-145): the fields are concatenated and sent through the equivalent
# `Stacked`, so elementwise fields cost one bjx_stacked launch whatever their number; the inverse takes the stacked array and
# returns the NamedTuple (:147-199).  Device batches: a field is a ROCMatrix (rows x batch), or a ROCVector holding one value per
# column for a scalar field (an Int range) / one unbatched vector otherwise; at least one field must be a ROCArray, else the
# reference's method runs.
field_rows(r::Integer) = 1
field_rows(r::AbstractUnitRange) = length(r)
as_range(r::Integer) = r:r
as_range(r::AbstractUnitRange) = r
function named_as_stacked(ns::NamedStacked{names}) where {names}               # input ranges = cumulative input lengths
    bs = Any[]; rin = UnitRange{Int}[]; off = 0
    for nm in names
        b = getfield(ns.transforms, nm); n_out = field_rows(getfield(ns.ranges, nm))
        n_in = b === identity ? n_out : first(Bijectors.output_size(inverse(b), (n_out,)))
        push!(bs, b); push!(rin, (off + 1):(off + n_in)); off += n_in
    end
    st = Stacked(Tuple(bs), rin)
    collect(st.ranges_out) == [as_range(getfield(ns.ranges, nm)) for nm in names] ||
        throw(ArgumentError("NamedStacked: ranges $(ns.ranges) do not match the output sizes of the transforms"))
    return st
end
function named_cat(ns::NamedStacked{names}, x::NamedTuple{names}) where {names}
    like = first(v for v in values(x) if v isa ROCArray); T = eltype(like)
    nb = maximum(v isa ROCMatrix ? size(v, 2) : 0 for v in values(x)); batched = nb > 0
    rows = Any[]
    for nm in names
        v = getfield(x, nm); scalar_field = getfield(ns.ranges, nm) isa Integer
        v = v isa Real ? ROCArray{T}(fill(T(v), 1)) : ondevice(T, v)
        if batched && v isa ROCVector
            v = scalar_field && length(v) != 1 ? reshape(v, 1, :) : repeat(reshape(v, :, 1), 1, nb)
        end
        batched && size(v, 2) != nb && throw(DimensionMismatch("NamedStacked: fields with different batch sizes"))
        push!(rows, v)
    end
    return reduce(vcat, rows)
end
const DeviceFields{names} = NamedTuple{names,<:Tuple{Vararg{Union{Real,AbstractVector,ROCArray}}}}
function with_logabsdet_jacobian(ns::NamedStacked{names}, x::DeviceFields{names}) where {names}
    any(v -> v isa ROCArray, values(x)) || return invoke(with_logabsdet_jacobian, Tuple{NamedStacked{names},NamedTuple{names}}, ns, x)
    return with_logabsdet_jacobian(named_as_stacked(ns), named_cat(ns, x))
end
function transform(ns::NamedStacked{names}, x::DeviceFields{names}) where {names}
    any(v -> v isa ROCArray, values(x)) || return invoke(transform, Tuple{NamedStacked{names},NamedTuple{names}}, ns, x)
    return transform(named_as_stacked(ns), named_cat(ns, x))
end
function named_split(ns::NamedStacked{names}, st::Stacked, xs::ROCVecOrMat) where {names}
    vals = map(names, Tuple(st.ranges_in)) do nm, r
        piece = xs isa ROCVector ? xs[r] : xs[r, :]
        getfield(ns.ranges, nm) isa Integer ? (xs isa ROCVector ? Array(piece)[1] : vec(piece)) : piece     # an Int range is a scalar field (:19-21)
    end
    return NamedTuple{names}(vals)
end
function with_logabsdet_jacobian(nsi::Inverse{<:NamedStacked{names}}, y::ROCVecOrMat{<:BjxFloat}) where {names}
    st = named_as_stacked(nsi.orig)
    xs, l = with_logabsdet_jacobian(inverse(st), y)
    return named_split(nsi.orig, st, xs), l
end
transform(nsi::Inverse{<:NamedStacked}, y::ROCVecOrMat{<:BjxFloat}) = first(with_logabsdet_jacobian(nsi, y))
logabsdetjac(nsi::Inverse{<:NamedStacked}, y::ROCVecOrMat{<:BjxFloat}) = last(with_logabsdet_jacobian(nsi, y))

# ---------------------------------------------------------------- VectorBijectors products batched over chains (SURVEY.md §8f f-2)
# src/vector/product/fill.jl:111-219: `product_distribution(fill(d, size...))` links every component with the SAME scalar map
# (positive.jl:11-50 Exp / Log, truncated.jl:17-103 Truncate / Untruncate, common.jl:27 TypedIdentity); DynamicPPL calls it on every
# log-density evaluation.  With one COLUMN per chain a (prod(size), n_chains) ROCMatrix is one bjx_chain launch; the log-det of a
# chain is its column's entry of the returned ROCVector (the reference's scalar, per chain).
const ScalarLink = Union{VB.Exp,VB.Log,VB.Truncate,VB.Untruncate,VB.TypedIdentity}
op0(kind) = BjxOp(Int32(kind), 0, 0, 0, C_NULL, C_NULL)
op2(kind, a, b=0.0) = BjxOp(Int32(kind), 1, Float64(a), Float64(b), C_NULL, C_NULL)
scalar_ops(e::VB.Exp) = vcat(op0(OP_EXP), e.sign < 0 ? [op0(OP_SIGNFLIP)] : BjxOp[], iszero(e.bound) ? BjxOp[] : [op2(OP_SHIFT, e.bound)])        # sign * exp(y) + bound
scalar_ops(l::VB.Log) = vcat(iszero(l.bound) ? BjxOp[] : [op2(OP_SHIFT, -l.bound)], l.sign < 0 ? [op0(OP_SIGNFLIP)] : BjxOp[], op0(OP_LOG))       # log(sign * (x - bound))
scalar_ops(t::VB.Truncate) = [op2(OP_TRUNCATED_INV, t.lower, t.upper)]         # (-Inf, Inf) -> (a, b): the four branches of truncated.jl:28-48
scalar_ops(u::VB.Untruncate) = [op2(OP_TRUNCATED, u.lower, u.upper)]           # (a, b) -> (-Inf, Inf), :79-99
scalar_ops(::VB.TypedIdentity) = [op0(OP_IDENTITY)]
function chains_launch(o::Vector{BjxOp}, x::ROCMatrix{T}, osz::Dims) where {T<:BjxFloat}
    d, n = size(x); y = similar(x, T, osz); lps = similar(x, T, n)
    GC.@preserve o x y lps check(ccall((:bjx_chain, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), o, length(o), devptr(x), devptr(y), devptr(lps), C_NULL, d, n, UInt32(0)), "bjx_chain")
    return y, lps
end
# to_linked_vec / to_vec of the product: columns = chains, rows = vec of the components (fill.jl:111-159)
function with_logabsdet_jacobian(t::VB.ProductVecTransform{<:VB.Elementwise{<:ScalarLink,Dims{M}},Nothing,Dims{0}}, x::ROCMatrix{T}) where {M,T<:BjxFloat}
    size(x, 1) == prod(t.transforms.size) || throw(DimensionMismatch("expected $(prod(t.transforms.size)) rows (one column per chain), got $(size(x, 1))"))
    return chains_launch(scalar_ops(t.transforms.value), x, size(x))
end
(t::VB.ProductVecTransform{<:VB.Elementwise{<:ScalarLink,Dims{M}},Nothing,Dims{0}})(x::ROCMatrix{<:BjxFloat}) where {M} = first(with_logabsdet_jacobian(t, x))
# from_linked_vec / from_vec (fill.jl:161-219): a (prod(size), n_chains) matrix of linked vectors -> size... x n_chains components
function with_logabsdet_jacobian(t::VB.ProductVecInvTransform{<:VB.Elementwise{<:ScalarLink,Dims{M}},Nothing,Dims{0}}, y::ROCMatrix{T}) where {M,T<:BjxFloat}
    size(y, 1) == prod(t.transforms.size) || throw(DimensionMismatch("expected $(prod(t.transforms.size)) rows (one column per chain), got $(size(y, 1))"))
    x, lps = chains_launch(scalar_ops(t.transforms.value), y, size(y))
    return reshape(x, t.transforms.size..., size(y, 2)), lps
end
(t::VB.ProductVecInvTransform{<:VB.Elementwise{<:ScalarLink,Dims{M}},Nothing,Dims{0}})(y::ROCMatrix{<:BjxFloat}) where {M} = first(with_logabsdet_jacobian(t, y))

# HETEROGENEOUS products of univariate components (src/vector/product/product.jl:95-131, 381-395: `product_distribution((d1, d2, …))`
# / a NamedTuple of univariate distributions whose links differ): component i owns row i of the (P, n_chains) matrix and its own
# scalar link (wrapped in VectWrap on the way to the linked vector, OnlyWrap on the way back, univariate.jl:7-30).  One bjx_stacked
# launch over all chains: one segment per run of consecutive components with the same link (<= 4 ops each); the log-det of a
# chain is its column's entry (the reference's scalar, per chain).
const WrappedLink = Union{VB.VectWrap{<:ScalarLink},VB.OnlyWrap{<:ScalarLink}}
function product_segments(links)
    segs = BjxSegment[]; lo = 0
    while lo < length(links)
        hi = lo + 1
        while hi < length(links) && links[hi + 1] == links[lo + 1]
            hi += 1
        end
        o = scalar_ops(links[lo + 1].bijector)
        length(o) <= 4 || return nothing
        push!(segs, BjxSegment(lo, lo, hi - lo, Int32(length(o)), 0, ntuple(k -> k <= length(o) ? o[k] : NOOP, 4)))
        lo = hi
    end
    return segs
end
function product_launch(links, x::ROCMatrix{T}) where {T<:BjxFloat}
    P, n = size(x)
    P == length(links) || throw(DimensionMismatch("expected $(length(links)) rows (one per component, one column per chain), got $P"))
    segs = product_segments(links)
    segs === nothing && throw(ArgumentError("a component link needs more than 4 fused ops"))
    y = similar(x); lps = similar(x, T, n)
    # what DynamicPPL repeats on every log-density evaluation: the segment list goes through a launch plan (validated once per context)
    cp = stacked_cplan(segs, Any[], T, Int(P), UInt32(0))
    GC.@preserve segs cp x y lps check(plan_run(cp, devptr(x), devptr(y), devptr(lps), C_NULL, n), "bjx_plan_run")
    return y, lps
end
const LinkTuple{P} = Union{NTuple{P,WrappedLink},NamedTuple{<:Any,<:NTuple{P,WrappedLink}}}
with_logabsdet_jacobian(t::VB.ProductVecTransform{<:LinkTuple{P},<:Any,Tuple{}}, x::ROCMatrix{<:BjxFloat}) where {P} = product_launch(collect(values(t.transforms)), x)
with_logabsdet_jacobian(t::VB.ProductVecInvTransform{<:LinkTuple{P},<:Any,Tuple{}}, y::ROCMatrix{<:BjxFloat}) where {P} = product_launch(collect(values(t.transforms)), y)
(t::VB.ProductVecTransform{<:LinkTuple{P},<:Any,Tuple{}})(x::ROCMatrix{<:BjxFloat}) where {P} = first(with_logabsdet_jacobian(t, x))
(t::VB.ProductVecInvTransform{<:LinkTuple{P},<:Any,Tuple{}})(y::ROCMatrix{<:BjxFloat}) where {P} = first(with_logabsdet_jacobian(t, y))
# arrays of univariate components with differing links (`product_distribution([d1, d2, …])` → AbstractArray of wrapped links)
with_logabsdet_jacobian(t::VB.ProductVecTransform{<:AbstractArray{<:WrappedLink},<:Any,Tuple{}}, x::ROCMatrix{<:BjxFloat}) = product_launch(vec(collect(t.transforms)), x)
with_logabsdet_jacobian(t::VB.ProductVecInvTransform{<:AbstractArray{<:WrappedLink},<:Any,Tuple{}}, y::ROCMatrix{<:BjxFloat}) = product_launch(vec(collect(t.transforms)), y)

# MvLogNormal links (src/vector/multivariate/mvlognormal.jl:1-15) over a matrix of chains: one bjx_chain launch, per-chain log-det
with_logabsdet_jacobian(::VB.MapLog, x::ROCMatrix{<:BjxFloat}) = chains_launch([op0(OP_LOG)], x, size(x))
with_logabsdet_jacobian(::VB.MapExp, x::ROCMatrix{<:BjxFloat}) = chains_launch([op0(OP_EXP)], x, size(x))
(t::VB.MapLog)(x::ROCMatrix{<:BjxFloat}) = first(with_logabsdet_jacobian(t, x))
(t::VB.MapExp)(x::ROCMatrix{<:BjxFloat}) = first(with_logabsdet_jacobian(t, x))
# JointOrderStatistics (src/vector/order/order.jl:14-76): the scalar link of the parent distribution over every element (sign flipped
# back when the link is decreasing), then the ordered vector to an unordered one — which IS inverse(OrderedBijector) (y₁, log(yᵢ − yᵢ₋₁)
# with log-det −Σ log(yᵢ − yᵢ₋₁), ordered.jl:50-80): one bjx_chain launch + one bjx_ordered launch over all chains, per-chain log-dets
function with_logabsdet_jacobian(m::VB.JointOrderWrap{<:ScalarLink}, x::ROCMatrix{T}) where {T<:BjxFloat}
    o = copy(scalar_ops(m.bijector))
    Bijectors.is_monotonically_decreasing(m.bijector) && push!(o, op0(OP_SIGNFLIP))
    y, l1 = chains_launch(o, x, size(x))
    z = similar(y)
    l2 = run!(plan_ordered(true, y), T, y, z)
    return z, l1 .+ l2
end
function with_logabsdet_jacobian(m::VB.InverseJointOrderWrap{<:ScalarLink}, y::ROCMatrix{T}) where {T<:BjxFloat}
    x = similar(y)
    l1 = run!(plan_ordered(false, y), T, y, x)                                  # xᵢ = exp(yᵢ) + xᵢ₋₁
    o = vcat(Bijectors.is_monotonically_decreasing(m.bijector) ? [op0(OP_SIGNFLIP)] : BjxOp[], scalar_ops(m.bijector))
    z, l2 = chains_launch(o, x, size(x))
    return z, l1 .+ l2
end
(m::VB.JointOrderWrap{<:ScalarLink})(x::ROCMatrix{<:BjxFloat}) = first(with_logabsdet_jacobian(m, x))
(m::VB.InverseJointOrderWrap{<:ScalarLink})(y::ROCMatrix{<:BjxFloat}) = first(with_logabsdet_jacobian(m, y))

# Stacked with Simplex / Ordered segments (stacked.jl:142-166) without slicing copies: the elementwise segments in one
# bjx_stacked_ld launch between matrices of different heights (identity placeholders on the structured rows), then
# bjx_simplex_ld / bjx_ordered_ld on row windows of the same matrices, accumulating their log-dets.
structured_entry(::SimplexBijector) = (:simplex, false)
structured_entry(::Inverse{<:SimplexBijector}) = (:simplex, true)
structured_entry(::OrderedBijector) = (:ordered, false)
structured_entry(::Inverse{OrderedBijector}) = (:ordered, true)
structured_entry(b) = nothing
function window_call(kind::Symbol, ::Type{T}, inv::Bool, pin, ld_in, pout, ld_out, lps, K, n) where {T}
    if kind === :simplex
        return ccall((:bjx_simplex_ld, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, UInt32),
            ctx().h, dtype(T), Cint(inv), pin, ld_in, pout, ld_out, lps, C_NULL, K, n, BJX_ACCUMULATE)
    end
    return ccall((:bjx_ordered_ld, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, UInt32),
        ctx().h, dtype(T), Cint(inv), pin, ld_in, pout, ld_out, lps, C_NULL, K, n, BJX_ACCUMULATE)
end
function stacked_structured(sb::Stacked, x::ROCMatrix{T}) where {T<:BjxFloat}
    d, n = size(x)
    dout = last(last(sb.ranges_out))
    keep = Any[]; segs = BjxSegment[]; later = Any[]
    for (b, rin, rout) in zip(sb.bs, sb.ranges_in, sb.ranges_out)
        o = ops(b, T, keep)
        if o !== nothing && length(o) <= 4 && length(rin) == length(rout)
            push!(segs, segment(rin, rout, o))
        else
            e = structured_entry(b)
            e === nothing && return invoke(with_logabsdet_jacobian, Tuple{Stacked,AbstractMatrix}, sb, x)
            push!(later, (e, rin, rout))
            push!(segs, BjxSegment(min(first(rin) - 1, d - length(rout)), first(rout) - 1, length(rout), 0, 0, ntuple(_ -> NOOP, 4)))
        end
    end
    y = similar(x, dout, n); lps = AMDGPU.zeros(T, n)
    # ONE launch when a column fits the LDS tile (bjx_stacked_mixed: a lane walks its column through every segment)
    blocks = [BjxBlock(Cint(kind === :simplex ? (inv ? 2 : 1) : (inv ? 4 : 3)), Cint(0), first(rin) - 1, first(rout) - 1, length(rin), length(rout))
              for ((kind, inv), rin, rout) in later]
    elem = [sg for sg in segs if !any(b -> b.out_lo == sg.out_lo, blocks)]
    rc = GC.@preserve keep x y lps ccall((:bjx_stacked_mixed, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{BjxBlock}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cdouble}, Int64, UInt32),
        ctx().h, dtype(T), elem, length(elem), blocks, length(blocks), devptr(x), d, devptr(y), dout, devptr(lps), C_NULL, n, UInt32(0))
    rc == 0 && return y, lps
    rc == BJX_ERR_UNSUPPORTED || check(rc, "bjx_stacked_mixed")       # taller columns: the window launches below
    GC.@preserve keep x y lps begin
        check(ccall((:bjx_stacked_ld, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
            ctx().h, dtype(T), segs, length(segs), devptr(x), d, devptr(y), dout, devptr(lps), C_NULL, dout, n, UInt32(0)), "bjx_stacked_ld")
        for ((kind, inv), rin, rout) in later
            K = kind === :simplex ? (inv ? length(rout) : length(rin)) : length(rin)
            check(window_call(kind, T, inv, devptr(x) + (first(rin) - 1) * sizeof(T), d, devptr(y) + (first(rout) - 1) * sizeof(T), dout,
                              devptr(lps), K, n), "bjx_$(kind)_ld")
        end
    end
    return y, lps
end

# ---------------------------------------------------------------- reverse-mode pullbacks (SURVEY.md §8f f-1)
# The reference's own rrules (ext/BijectorsChainRulesCoreExt.jl:65-197, :199-320) for ROCArray primals, and rrules of
# with_logabsdet_jacobian for the bijectors whose pullback the reference leaves to the AD package: the pullback closure
# calls the `_vjp` entry with the saved primal input.  Cotangent convention of every `_vjp` entry:
#   in_bar = J(in)ᵀ out_bar + ladj_bar ∇ logabsdetjac(in).
cotangent(::Type{T}, Δ, like) where {T} = (d = unthunk(Δ); d isa ChainRulesCore.AbstractZero ? fill!(similar(like), zero(T)) : ondevice(T, d))
ladj_cotangent(::Type{T}, Δ, n) where {T} = (d = unthunk(Δ); d isa ChainRulesCore.AbstractZero ? nothing : (d isa Real ? ROCArray{T}(fill(T(d), n)) : ondevice(T, d)))

function ordered_vjp(inv::Bool, x::ROCMatrix{T}, Δy, Δl=nothing) where {T}
    x̄ = similar(x)
    GC.@preserve x Δy Δl x̄ check(ccall((:bjx_ordered_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), Cint(inv), devptr(x), devptr(Δy), devptr(Δl), devptr(x̄), size(x, 1), size(x, 2)), "bjx_ordered_vjp")
    return x̄
end
function ChainRulesCore.rrule(::typeof(Bijectors._transform_ordered), y::ROCMatrix{T}) where {T<:BjxFloat}       # ext/…CoreExt.jl:92-117
    x = transform(OrderedBijector(), y)
    _transform_ordered_adjoint(Δ) = (NoTangent(), ordered_vjp(false, y, cotangent(T, Δ, y)))
    return x, _transform_ordered_adjoint
end
function ChainRulesCore.rrule(::typeof(Bijectors._transform_inverse_ordered), x::ROCMatrix{T}) where {T<:BjxFloat}  # :155-197
    y = transform(inverse(OrderedBijector()), x)
    _transform_inverse_ordered_adjoint(Δ) = (NoTangent(), ordered_vjp(true, x, cotangent(T, Δ, x)))
    return y, _transform_inverse_ordered_adjoint
end

# SimplexBijector: simplex.jl:145-215 (simplex_logabsdetjac_gradient), :248-308 (link adjoint), :358-470 (invlink adjoint) as one
# O(K) pullback of with_logabsdet_jacobian per direction.  The log-det is a scalar (summed over columns): its cotangent is a Real.
function simplex_vjp(inv::Bool, x::ROCMatrix{T}, Δy, Δl) where {T}
    K = inv ? size(x, 1) + 1 : size(x, 1); n = size(x, 2)
    x̄ = similar(x)
    GC.@preserve x Δy Δl x̄ check(ccall((:bjx_simplex_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), Cint(inv), devptr(x), devptr(Δy), devptr(Δl), devptr(x̄), K, n), "bjx_simplex_vjp")
    return x̄
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), b::Union{SimplexBijector,Inverse{<:SimplexBijector}}, x::ROCMatrix{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(b, x)
    inv = b isa Inverse
    function pullback_simplex((Δy, Δl))
        x̄ = simplex_vjp(inv, x, cotangent(T, Δy, out[1]), ladj_cotangent(T, Δl, size(x, 2)))
        return NoTangent(), NoTangent(), x̄
    end
    return out, pullback_simplex
end

# Elementwise chains and Stacked of elementwise chains: x̄ = (dy/dx) ȳ + ℓ̄ d logabsdetjac/dx, element by element
# (bjx_stacked_vjp; same segment list as bjx_stacked).  The log-det is a scalar: ℓ̄ is a Real, broadcast to the columns.
# The pullback a gradient-based sampler repeats on every leapfrog step goes through a pullback PLAN (bjx_plan_stacked_vjp: the segment list
# validated once, kept by the calling task's context; CPLANS is keyed by the segments' CONTENT — they hold pointers and scalars only).
const VJP_PLANS = Dict{Tuple{Ptr{Cvoid},UInt,DataType,Int},CPlan}()          # guarded by CPLANS_LOCK
function stacked_vjp_cplan(segs::Vector{BjxSegment}, keep, ::Type{T}, d::Int) where {T}
    c = ctx()
    key = (c.h, hash(segs), T, d)
    lock(CPLANS_LOCK) do
        get!(VJP_PLANS, key) do
            hp = Ref{Ptr{Cvoid}}(C_NULL)
            GC.@preserve segs keep check(ccall((:bjx_plan_stacked_vjp, libbjx), Cint, (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Int64, Ptr{Ptr{Cvoid}}),
                                               c.h, dtype(T), segs, length(segs), d, hp), "bjx_plan_stacked_vjp")
            cp = CPlan(hp[], copy(keep))
            finalizer(x -> ccall((:bjx_plan_destroy, libbjx), Cint, (Ptr{Cvoid},), x.h), cp)
            cp
        end
    end
end
# the forward `Stacked` of elementwise segments through a plan (bjx_plan_stacked, run by plan_run): the linked vector of a heterogeneous product
function stacked_cplan(segs::Vector{BjxSegment}, keep, ::Type{T}, d::Int, flags::UInt32) where {T}
    c = ctx()
    key = (c.h, hash(segs) ⊻ UInt(flags) ⊻ UInt(0x5), T, d)
    lock(CPLANS_LOCK) do
        get!(VJP_PLANS, key) do
            hp = Ref{Ptr{Cvoid}}(C_NULL)
            GC.@preserve segs keep check(ccall((:bjx_plan_stacked, libbjx), Cint, (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Int64, UInt32, Ptr{Ptr{Cvoid}}),
                                               c.h, dtype(T), segs, length(segs), d, flags, hp), "bjx_plan_stacked")
            cp = CPlan(hp[], copy(keep))
            finalizer(x -> ccall((:bjx_plan_destroy, libbjx), Cint, (Ptr{Cvoid},), x.h), cp)
            cp
        end
    end
end
function stacked_vjp(segs::Vector{BjxSegment}, keep, x::ROCVecOrMat{T}, Δy, Δl) where {T}
    d, n = dims(x)
    x̄ = similar(x)
    cp = stacked_vjp_cplan(segs, keep, T, Int(d))
    GC.@preserve keep cp x Δy Δl x̄ check(ccall((:bjx_plan_run_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
        cp.h, devptr(x), devptr(Δy), devptr(Δl), devptr(x̄), n), "bjx_plan_run_vjp")
    return x̄
end
# (the unplanned entry, for segment lists that are built once and thrown away)
function stacked_vjp_once(segs::Vector{BjxSegment}, keep, x::ROCVecOrMat{T}, Δy, Δl) where {T}
    d, n = dims(x)
    x̄ = similar(x)
    GC.@preserve keep x Δy Δl x̄ check(ccall((:bjx_stacked_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), segs, length(segs), devptr(x), devptr(Δy), devptr(Δl), devptr(x̄), d, n), "bjx_stacked_vjp")
    return x̄
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), b::Fusable, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    keep = Any[]
    o = ops(b, T, keep)
    (o === nothing || length(o) > 4) && return nothing          # `nothing` = no rule: the AD package differentiates the stages
    out = with_logabsdet_jacobian(b, x)
    segs = [segment(1:size(x, 1), 1:size(x, 1), o)]
    function pullback_chain((Δy, Δl))
        x̄ = stacked_vjp(segs, keep, x, cotangent(T, Δy, out[1]), ladj_cotangent(T, Δl, dims(x)[2]))
        return NoTangent(), NoTangent(), x̄              # parameter cotangents: meanfield_pullback / the AD package
    end
    return out, pullback_chain
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), sb::Stacked, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    keep = Any[]
    segs = elementwise_segments(sb, T, keep)
    segs === nothing && return nothing
    out = with_logabsdet_jacobian(sb, x)
    function pullback_stacked((Δy, Δl))
        x̄ = stacked_vjp(segs, keep, x, cotangent(T, Δy, out[1]), ladj_cotangent(T, Δl, dims(x)[2]))
        return NoTangent(), NoTangent(), x̄
    end
    return out, pullback_stacked
end

# Mean-field family y = tail(μ .+ σ .* z) (ADVI): input pullback and the (μ̄, σ̄) reductions in ONE pass over z and ȳ.
# moments[1:d] = Σ_n z̄, moments[d+1:2d] = Σ_n z̄ .* z  =>  μ̄ = moments[1:d] ./ σ,  σ̄ = (moments[d+1:2d] .+ sum(ℓ̄)) ./ σ
function meanfield_pullback(chain, z::ROCMatrix{T}, ȳ::ROCMatrix{T}, ℓ̄::ROCVector{T}) where {T<:BjxFloat}
    d, n = dims(z)
    keep = Any[]
    o = ops(chain, T, keep)                                  # tail ∘ Shift(μ) ∘ Scale(σ) as <= 4 elementwise ops
    seg = [segment(1:d, 1:d, o)]
    z̄ = similar(z)
    moments = AMDGPU.zeros(Float64, 2d + 1)
    GC.@preserve keep z ȳ ℓ̄ z̄ moments check(ccall((:bjx_stacked_vjp_moments, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxSegment}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64),
        ctx().h, dtype(T), seg, 1, devptr(z), devptr(ȳ), devptr(ℓ̄), devptr(z̄), Ptr{Cdouble}(pointer(moments)), d, n), "bjx_stacked_vjp_moments")
    return z̄, moments
end
# the same sums for cotangents that already exist (shapes the fused kernel does not take): Σ_n a, Σ_n a .* b per row
function row_moments(a::ROCMatrix{T}, b::Union{Nothing,ROCMatrix{T}}=nothing) where {T<:BjxFloat}
    d, n = size(a)
    out = AMDGPU.zeros(Float64, 2d + 1)
    GC.@preserve a b out check(ccall((:bjx_row_moments, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64),
        ctx().h, dtype(T), devptr(a), devptr(b), Ptr{Cdouble}(pointer(out)), d, n), "bjx_row_moments")
    return out
end

# VecCholeskyBijector: the rule the reference ships for ONE packed vector (`_inv_link_chol_lkj(y::AbstractVector)`, corr.jl:370-451;
# ext/BijectorsChainRulesCoreExt.jl:311-320) on a ROCVector, and the batched form — columns = samples — on the PUBLIC call
# with_logabsdet_jacobian(inverse(VecCholeskyBijector), y::ROCMatrix).  (Not on `_inv_link_chol_lkj(::ROCMatrix)`: in the
# reference that method takes ONE K x K matrix of free parameters, corr.jl:344-368, a different meaning of the argument.)
function vec_cholesky_inv_pullback(ul::Cint, y::ROCVecOrMat{T}, ΔW, Δl) where {T}
    m, n = dims(y); K = Bijectors._triu1_dim_from_length(m)
    Δy = similar(y)
    GC.@preserve y ΔW Δl Δy check(ccall((:bjx_vec_cholesky_inv_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), ul, devptr(y), devptr(ΔW), devptr(Δl), devptr(Δy), K, n), "bjx_vec_cholesky_inv_vjp")
    return Δy
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), ib::Inverse{VecCholeskyBijector}, y::ROCVecOrMat{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(ib, y)
    function pullback_inverse_vec_cholesky((ΔW, ΔlogJ))
        Δy = vec_cholesky_inv_pullback(uplo(ib.orig), y, cotangent(T, ΔW, out[1]), ladj_cotangent(T, ΔlogJ, dims(y)[2]))
        return NoTangent(), NoTangent(), Δy
    end
    return out, pullback_inverse_vec_cholesky
end
function ChainRulesCore.rrule(::typeof(Bijectors._inv_link_chol_lkj), y::ROCVector{T}) where {T<:BjxFloat}       # one sample, the reference's meaning
    W, logJ = with_logabsdet_jacobian(inverse(VecCholeskyBijector(:U)), y)
    function pullback_inv_link_chol_lkj((ΔW, ΔlogJ))
        return NoTangent(), vec_cholesky_inv_pullback(Cint('U'), y, cotangent(T, ΔW, W), ladj_cotangent(T, ΔlogJ, 1))
    end
    return (W, logJ), pullback_inv_link_chol_lkj
end
# forward LKJ link on a batch of factors W[K, K, n] (ext/BijectorsChainRulesCoreExt.jl:199-311)
for (f, ul) in ((:_link_chol_lkj_from_upper, 'U'), (:_link_chol_lkj_from_lower, 'L'))
    @eval function ChainRulesCore.rrule(::typeof(Bijectors.$f), W::ROCArray{T,3}) where {T<:BjxFloat}
        K, n = size(W, 1), size(W, 3)
        y = transform(VecCholeskyBijector(Symbol($ul)), W)
        function pullback_link_chol_lkj(Δz)
            ΔW = similar(W); Δc = cotangent(T, Δz, y)
            GC.@preserve W Δc ΔW check(ccall((:bjx_vec_cholesky_fwd_vjp, libbjx), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
                ctx().h, dtype(T), Cint($ul), devptr(W), devptr(Δc), devptr(ΔW), K, n), "bjx_vec_cholesky_fwd_vjp")
            return NoTangent(), ΔW
        end
        return y, pullback_link_chol_lkj
    end
end

# Matrix-variate constraint bijectors (SURVEY.md §8f f-1 x f-4): the rules the reference ships piecewise — pd_from_upper
# (ext/BijectorsChainRulesCoreExt.jl:324-331), replace_diag / pd_from_lower / lower_ / upper_triangular (ext/BijectorsReverseDiffExt.jl:
# 143-193), cholesky_lower / _upper (ext/BijectorsReverseDiffChainRulesExt.jl:11-40), _inv_link_chol_lkj (corr.jl:402-461) — chained
# per sample in ONE launch, attached to the PUBLIC call on device arrays.  Forward direction: the cotangent of X lands on the
# triangle the reference reads (upper for the correlation bijectors, lower for PD).  One literal ccall per entry.
matrix_vjp_launch(::VecCorrBijector, h, dt, inv, pin, pg, pl, pout, K, n) = ccall((:bjx_vec_corr_vjp, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), h, dt, inv, pin, pg, pl, pout, K, n)
matrix_vjp_launch(::CorrBijector, h, dt, inv, pin, pg, pl, pout, K, n) = ccall((:bjx_corr_vjp, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), h, dt, inv, pin, pg, pl, pout, K, n)
matrix_vjp_launch(::PDBijector, h, dt, inv, pin, pg, pl, pout, K, n) = ccall((:bjx_pd_vjp, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), h, dt, inv, pin, pg, pl, pout, K, n)
matrix_vjp_launch(::PDVecBijector, h, dt, inv, pin, pg, pl, pout, K, n) = ccall((:bjx_pd_vec_vjp, libbjx), Cint,
    (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), h, dt, inv, pin, pg, pl, pout, K, n)
function matrix_pullback(b::MatrixKinds, inv::Bool, x::ROCArray{T}, Δout, Δl, K::Integer, n::Integer) where {T}
    x̄ = similar(x)
    GC.@preserve x Δout Δl x̄ check(matrix_vjp_launch(b, ctx().h, dtype(T), Cint(inv), devptr(x), devptr(Δout), devptr(Δl), devptr(x̄), Int64(K), Int64(n)),
                                  String(entry_name(b)) * "_vjp")
    return x̄
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), b::MatrixKinds, X::ROCArray{T}) where {T<:BjxFloat}
    ndims(X) in (2, 3) || return nothing
    out = with_logabsdet_jacobian(b, X)
    K, n = size(X, 1), size(X, 3)
    function pullback_matrix_link((Δy, Δl))
        return NoTangent(), NoTangent(), matrix_pullback(b, false, X, cotangent(T, Δy, out[1]), ladj_cotangent(T, Δl, n), K, n)
    end
    return out, pullback_matrix_link
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), ib::Inverse{<:MatrixKinds}, Y::ROCArray{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(ib, Y)                       # X (K, K[, n]) and the log-det
    K, n = size(out[1], 1), size(out[1], 3)
    function pullback_matrix_invlink((ΔX, Δl))
        return NoTangent(), NoTangent(), matrix_pullback(ib.orig, true, Y, cotangent(T, ΔX, out[1]), ladj_cotangent(T, Δl, n), K, n)
    end
    return out, pullback_matrix_invlink
end
# Scale with a matrix (scale.jl:14,17,35-36; ext/BijectorsReverseDiffExt.jl:72-115): x̄ = a'ȳ through the same entry with the transposed
# matrix; ā = ȳ x' + (Σ ℓ̄) a⁻ᵀ is a dense GEMM over the batch and a solve — the host's library calls (rocBLAS / rocSOLVER via AMDGPU.jl).
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), b::Scale{<:ROCMatrix{T}}, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(b, x)
    function pullback_scale_matrix((Δy, Δl))
        ȳ = cotangent(T, Δy, x)
        x̄ = transform(Scale(ROCArray(permutedims(b.a))), ȳ)
        dl = unthunk(Δl)
        ā = reshape(ȳ, size(b.a, 1), :) * reshape(x, size(b.a, 1), :)'
        dl isa ChainRulesCore.AbstractZero || (ā = ā .+ T(dl) .* permutedims(inv(b.a)))      # the reference's scalar log-det: logabsdet(a) ONCE (:36)
        return NoTangent(), Tangent{typeof(b)}(a = ā), x̄
    end
    return out, pullback_scale_matrix
end

# PlanarLayer (one layer or a planner run): input pullback (bjx_planar_vjp) and, for the forward flow, the parameter cotangents
# (w̄, ū, b̄) through get_u_hat in the same pass (bjx_planar_vjp_params), on the layer-major tables of `planar_tables`.
function planar_vjp(w, u, b, nl::Integer, inv::Bool, z::ROCMatrix{T}, Δy, Δl) where {T}
    z̄ = similar(z)
    GC.@preserve z Δy Δl z̄ w u b check(ccall((:bjx_planar_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), Cint(inv), devptr(w), devptr(u), devptr(b), Cint(nl), devptr(z), devptr(Δy), devptr(Δl), devptr(z̄),
        size(z, 1), size(z, 2)), "bjx_planar_vjp")
    return z̄
end
function planar_vjp_params(w, u, b, nl::Integer, z::ROCMatrix{T}, Δy, Δl) where {T}
    z̄ = similar(z)
    w̄, ū, b̄ = similar(w), similar(u), similar(b)
    work = similar(z, 2 * nl * size(z, 2))                 # 2 * n_layers * batch
    GC.@preserve z Δy Δl z̄ w u b w̄ ū b̄ work check(ccall((:bjx_planar_vjp_params, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
         Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), devptr(w), devptr(u), devptr(b), Cint(nl), devptr(z), devptr(Δy), devptr(Δl), devptr(z̄),
        devptr(w̄), devptr(ū), devptr(b̄), devptr(work), size(z, 1), size(z, 2)), "bjx_planar_vjp_params")
    return z̄, w̄, ū, b̄
end
# (z̄, per-layer tangents in the order of `layers`) of a run; inv: implicit function theorem (the reference differentiates the
# Newton root through its find_alpha rule, ext/BijectorsChainRulesCoreExt.jl:42-46): with x = f⁻¹(y) and the inverse's log-det
# -ℓ(x),  ȳ = J⁻ᵀ(x̄ - ℓ̄ ∇ₓℓ)  (bjx_planar_vjp, inverse = 1)  and  θ̄ = the FORWARD parameter pullback at x with cotangents
# (-ȳ, -ℓ̄)  (bjx_planar_vjp_params) — no new kernel, the root is not differentiated through.
function planar_run_pullback(layers, inv::Bool, z::ROCMatrix{T}, x, Δy, Δl) where {T}
    d = size(z, 1); nl = length(layers)
    w, u, b = planar_tables(T, layers, d)
    if inv
        z̄ = planar_vjp(w, u, b, nl, true, z, Δy, Δl)
        _, w̄, ū, b̄ = planar_vjp_params(w, u, b, nl, x, -z̄, Δl === nothing ? nothing : -Δl)
    else
        z̄, w̄, ū, b̄ = planar_vjp_params(w, u, b, nl, z, Δy, Δl)
    end
    ts = [Tangent{typeof(layers[k])}(w = w̄[((k - 1) * d + 1):(k * d)], u = ū[((k - 1) * d + 1):(k * d)], b = b̄[k:k]) for k in 1:nl]
    return z̄, ts
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), flow::PlanarLayer, z::ROCMatrix{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(flow, z)
    function pullback_planar_params((Δy, Δl))
        z̄, ts = planar_run_pullback(PlanarLayer[flow], false, z, nothing, cotangent(T, Δy, z), ladj_cotangent(T, Δl, size(z, 2)))
        return NoTangent(), ts[1], z̄
    end
    return out, pullback_planar_params
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), flow::Inverse{<:PlanarLayer}, y::ROCMatrix{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(flow, y)
    function pullback_planar((Δx, Δl))
        ȳ, ts = planar_run_pullback(PlanarLayer[flow.orig], true, y, out[1], cotangent(T, Δx, y), ladj_cotangent(T, Δl, size(y, 2)))
        return NoTangent(), Tangent{typeof(flow)}(orig = ts[1]), ȳ
    end
    return out, pullback_planar
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), r::PlanarRun, z::ROCMatrix{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(r, z)
    function pullback_planar_run((Δy, Δl))
        z̄, ts = planar_run_pullback(r.layers, r.inv, z, out[1], cotangent(T, Δy, z), ladj_cotangent(T, Δl, size(z, 2)))
        return NoTangent(), Tangent{PlanarRun}(layers = ts), z̄
    end
    return out, pullback_planar_run
end

# A composition on a ROCArray: the chain rule over the planner's pieces, every piece through its own rule (PlanarRun, RadialLayer,
# splines, elementwise chains …) or, for stages the ABI does not carry, the AD package.  The log-det is the SUM of the pieces'
# log-dets, so every piece receives the same ℓ̄.  The piece tangents are folded back onto the nest of ComposedFunctions.
stage_tangents(pc::PlanarRun, t) = (ts = unthunk(t).layers; pc.inv ? Any[Tangent{Inverse{typeof(l)}}(orig = tl) for (l, tl) in zip(reverse(pc.layers), reverse(ts))] : Any[ts...])
stage_tangents(pc::ComposedFunction, t) = Any[NoTangent() for _ in stages(pc)]      # an elementwise run: parameters via meanfield_pullback
stage_tangents(pc, t) = Any[t]
fold_tangent(::typeof(identity), ts, i) = (NoTangent(), i)
fold_tangent(b, ts, i) = (ts[i], i + 1)
function fold_tangent(b::ComposedFunction, ts, i)                                   # stages are numbered inner first
    ti, i = fold_tangent(b.inner, ts, i)
    to, i = fold_tangent(b.outer, ts, i)
    return Tangent{typeof(b)}(outer = to, inner = ti), i
end
function ChainRulesCore.rrule(cfg::ChainRulesCore.RuleConfig{>:ChainRulesCore.HasReverseMode}, ::typeof(with_logabsdet_jacobian),
                              b::ComposedFunction, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    fused = ChainRulesCore.rrule(with_logabsdet_jacobian, b, x)                     # ONE elementwise segment (<= 4 ops): its own rule
    fused === nothing || return fused
    pcs = pieces(b); cur = x; total = nothing; backs = Any[]
    for pc in pcs
        out, back = ChainRulesCore.rrule_via_ad(cfg, with_logabsdet_jacobian, pc, cur)
        push!(backs, back)
        cur = out[1]; total = total === nothing ? out[2] : total + out[2]
    end
    function pullback_composed((Δy, Δl))
        g = Δy; ts = Any[]
        for k in length(pcs):-1:1
            _, t, g = backs[k]((g, Δl))
            prepend!(ts, stage_tangents(pcs[k], t))
        end
        return NoTangent(), first(fold_tangent(b, ts, 1)), g
    end
    return (cur, total), pullback_composed
end

# RadialLayer: input pullback of both directions (bjx_radial_vjp) and the parameter cotangents (ᾱ_, β̄, z̄₀) of the forward
# flow (bjx_radial_vjp_params; raw α_, β behind softplus, radial_layer.jl:43-60); inverse parameters by the same IFT as Planar.
function radial_vjp(fl::RadialLayer, inv::Bool, z::ROCMatrix{T}, Δy, Δl) where {T}
    z̄ = similar(z)
    a, be, z0 = ondevice(T, fl.α_ isa Real ? [fl.α_] : fl.α_), ondevice(T, fl.β isa Real ? [fl.β] : fl.β), ondevice(T, fl.z_0)
    GC.@preserve z Δy Δl z̄ a be z0 check(ccall((:bjx_radial_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), Cint(inv), devptr(a), devptr(be), devptr(z0), devptr(z), devptr(Δy), devptr(Δl), devptr(z̄),
        size(z, 1), size(z, 2)), "bjx_radial_vjp")
    return z̄
end
function radial_vjp_params(fl::RadialLayer, z::ROCMatrix{T}, Δy, Δl) where {T}
    a, be, z0 = ondevice(T, fl.α_ isa Real ? [fl.α_] : fl.α_), ondevice(T, fl.β isa Real ? [fl.β] : fl.β), ondevice(T, fl.z_0)
    z̄, ᾱ, β̄, z̄0 = similar(z), similar(a), similar(be), similar(z0)
    work = similar(z, 2 * size(z, 2))
    GC.@preserve z Δy Δl z̄ a be z0 ᾱ β̄ z̄0 work check(ccall((:bjx_radial_vjp_params, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), devptr(a), devptr(be), devptr(z0), devptr(z), devptr(Δy), devptr(Δl), devptr(z̄),
        devptr(ᾱ), devptr(β̄), devptr(z̄0), devptr(work), size(z, 1), size(z, 2)), "bjx_radial_vjp_params")
    return z̄, ᾱ, β̄, z̄0
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), flow::RadialLayer, z::ROCMatrix{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(flow, z)
    function pullback_radial((Δy, Δl))
        z̄, ᾱ, β̄, z̄0 = radial_vjp_params(flow, z, cotangent(T, Δy, z), ladj_cotangent(T, Δl, size(z, 2)))
        return NoTangent(), Tangent{typeof(flow)}(α_ = ᾱ, β = β̄, z_0 = z̄0), z̄
    end
    return out, pullback_radial
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), flow::Inverse{<:RadialLayer}, y::ROCMatrix{T}) where {T<:BjxFloat}
    rl = flow.orig
    out = with_logabsdet_jacobian(flow, y)
    x = out[1]
    function pullback_radial_inverse((Δx, Δl))
        Δlc = ladj_cotangent(T, Δl, size(y, 2))
        ȳ = radial_vjp(rl, true, y, cotangent(T, Δx, x), Δlc)
        _, ᾱ, β̄, z̄0 = radial_vjp_params(rl, x, -ȳ, Δlc === nothing ? nothing : -Δlc)
        return NoTangent(), Tangent{typeof(flow)}(orig = Tangent{typeof(rl)}(α_ = ᾱ, β = β̄, z_0 = z̄0)), ȳ
    end
    return out, pullback_radial_inverse
end

# RationalQuadraticSpline with matrix parameters: input pullback AND the cotangents of the knot arrays summed over the batch in
# one pass (bjx_rqs_vjp_knots with in_bar), both directions (inverse: implicit function theorem at x = f⁻¹(y));
# bjx_rqs_vjp alone when only the input cotangent is wanted (rqs_input_pullback: HMC over x with fixed knots).
# rational_quadratic_spline.jl:128-357 has no hand-written rule.  For a spline made by the `B` constructor (:109-123) the
# wrapper that owns the raw parameters chains on with `rqs_params_pullback` (bjx_rqs_params_vjp).
function rqs_input_pullback(b::RationalQuadraticSpline{<:ROCMatrix{T}}, inv::Bool, x::ROCMatrix{T}, Δy, Δl) where {T<:BjxFloat}
    x̄ = similar(x)
    GC.@preserve b x Δy Δl x̄ check(ccall((:bjx_rqs_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), Cint(inv), devptr(b.widths), devptr(b.heights), devptr(b.derivatives), Cint(size(b.widths, 2)),
        devptr(x), devptr(Δy), devptr(Δl), devptr(x̄), size(x, 1), size(x, 2)), "bjx_rqs_vjp")
    return x̄
end
function rqs_knot_pullback(b::RationalQuadraticSpline{<:ROCMatrix{T}}, inv::Bool, x::ROCMatrix{T}, Δy, Δl) where {T<:BjxFloat}
    x̄, w̄, h̄, d̄ = similar(x), similar(b.widths), similar(b.heights), similar(b.derivatives)
    GC.@preserve b x Δy Δl x̄ w̄ h̄ d̄ check(ccall((:bjx_rqs_vjp_knots, libbjx), Cint,    # x̄ and the knot cotangents in one pass over x, Δy, Δl
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
        ctx().h, dtype(T), Cint(inv), devptr(b.widths), devptr(b.heights), devptr(b.derivatives), Cint(size(b.widths, 2)), devptr(x), devptr(Δy), devptr(Δl), devptr(x̄),
        devptr(w̄), devptr(h̄), devptr(d̄), size(x, 1), size(x, 2)), "bjx_rqs_vjp_knots")
    return x̄, w̄, h̄, d̄
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), b::RationalQuadraticSpline{<:ROCMatrix{T}}, x::ROCMatrix{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(b, x)
    function pullback_rqs((Δy, Δl))
        x̄, w̄, h̄, d̄ = rqs_knot_pullback(b, false, x, cotangent(T, Δy, x), ladj_cotangent(T, Δl, size(x, 2)))
        return NoTangent(), Tangent{typeof(b)}(widths = w̄, heights = h̄, derivatives = d̄), x̄
    end
    return out, pullback_rqs
end
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), ib::Inverse{<:RationalQuadraticSpline{<:ROCMatrix{T}}}, y::ROCMatrix{T}) where {T<:BjxFloat}
    b = ib.orig
    out = with_logabsdet_jacobian(ib, y)
    function pullback_rqs_inverse((Δx, Δl))
        ȳ, w̄, h̄, d̄ = rqs_knot_pullback(b, true, y, cotangent(T, Δx, y), ladj_cotangent(T, Δl, size(y, 2)))
        return NoTangent(), Tangent{typeof(ib)}(orig = Tangent{typeof(b)}(widths = w̄, heights = h̄, derivatives = d̄)), ȳ
    end
    return out, pullback_rqs_inverse
end
# pullback of the `B` constructor: knot cotangents (dim, K+1) -> cotangents of the unconstrained (dim, K), (dim, K), (dim, K-1)
function rqs_params_pullback(raw_w::ROCMatrix{T}, raw_h::ROCMatrix{T}, raw_d::ROCMatrix{T}, B::Real, w̄, h̄, d̄) where {T}
    r̄w, r̄h, r̄d = similar(raw_w), similar(raw_h), similar(raw_d)
    GC.@preserve raw_w raw_h raw_d w̄ h̄ d̄ r̄w r̄h r̄d check(ccall((:bjx_rqs_params_vjp, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ctx().h, dtype(T), devptr(raw_w), devptr(raw_h), devptr(raw_d), Cint(size(raw_w, 2)), size(raw_w, 1), Float64(B),
        devptr(w̄), devptr(h̄), devptr(d̄), devptr(r̄w), devptr(r̄h), devptr(r̄d)), "bjx_rqs_params_vjp")
    return r̄w, r̄h, r̄d
end

# Coupling with the affine law (coupling.jl:206-259; together with the reference's rrule(::typeof(combine), …),
# ext/BijectorsChainRulesCoreExt.jl:48-63): the kernel returns x̄ (x₁ rows and pass-through rows) and the cotangents (s̄, t̄) of
# θ's outputs; θ's own pullback (an arbitrary closure: the AD package) turns them into the x₂ contribution, added on the x₂ rows.
function ChainRulesCore.rrule(cfg::ChainRulesCore.RuleConfig{>:ChainRulesCore.HasReverseMode}, ::typeof(with_logabsdet_jacobian),
                              cl::Union{Coupling,Inverse{<:Coupling}}, x::ROCMatrix{T}) where {T<:BjxFloat}
    inv = cl isa Inverse; c = inv ? cl.orig : cl
    d, n = size(x)
    idx1, idx2 = mask_rows(c.mask.A_1), mask_rows(c.mask.A_2); n1 = length(idx1)
    x2 = gather_rows(x, idx2)
    law_obj, θ_back = ChainRulesCore.rrule_via_ad(cfg, c.θ, x2)
    law = coupling_law(law_obj)
    (law === nothing || law[1] !== :affine) && return nothing     # spline laws: bjx_rqs_vjp_knots on the x₁ rows via the stage rules
    out = with_logabsdet_jacobian(cl, x)
    function pullback_coupling((Δy, Δl))
        keep = Any[]; di = ROCArray{Int32}(idx1)
        # the pullback entry takes T[n1, batch] parameters: a per-row law is expanded here and its cotangent summed over the columns
        full(a) = a === nothing ? nothing : (v = ondevice(T, a isa Real ? fill(T(a), n1) : a); length(v) == n1 * n ? v : repeat(reshape(v, n1, 1), 1, n))
        reduce_like(ā, a) = (ā === nothing || a === nothing || length(a) == n1 * n) ? ā : ROCArray{T}(Array(row_moments(ā))[1:n1])
        sv, tv = full(law[2]), full(law[3]); push!(keep, sv, tv)
        ps, pt = devptr(sv), devptr(tv)
        x̄ = similar(x); s̄ = sv === nothing ? nothing : similar(x, n1, n); t̄ = tv === nothing ? nothing : similar(x, n1, n)
        Δyc = cotangent(T, Δy, x); Δlc = ladj_cotangent(T, Δl, n)
        GC.@preserve keep di x Δyc Δlc x̄ s̄ t̄ check(ccall((:bjx_coupling_affine_vjp, libbjx), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Int32}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
            ctx().h, dtype(T), Cint(inv), Ptr{Int32}(pointer(di)), n1, ps, pt, devptr(x), devptr(Δyc), devptr(Δlc), devptr(x̄), devptr(s̄), devptr(t̄), d, n),
            "bjx_coupling_affine_vjp")
        s̄, t̄ = reduce_like(s̄, law[2]), reduce_like(t̄, law[3])
        law_bar = law_obj isa Shift ? Tangent{typeof(law_obj)}(a = t̄) :
                  law_obj isa Scale ? Tangent{typeof(law_obj)}(a = s̄) :
                  Tangent{typeof(law_obj)}(outer = Tangent{typeof(law_obj.outer)}(a = t̄), inner = Tangent{typeof(law_obj.inner)}(a = s̄))
        _, x̄2 = θ_back(law_bar)
        x̄[Int.(idx2) .+ 1, :] .+= unthunk(x̄2)                    # the x₂ rows also pass ȳ through (already in x̄)
        return NoTangent(), NoTangent(), NoTangent(), x̄
    end
    return out, pullback_coupling
end

# Permute: the pullback of a gather is the gather with the inverse permutation (bit-exact data movement)
function ChainRulesCore.rrule(::typeof(with_logabsdet_jacobian), b::Permute, x::ROCVecOrMat{T}) where {T<:BjxFloat}
    out = with_logabsdet_jacobian(b, x)
    pullback_permute((Δy, Δl)) = (NoTangent(), NoTangent(), transform(inverse(b), cotangent(T, Δy, x)))
    return out, pullback_permute
end

# captured steps (hipGraph): record the calls of `f()` once, replay them with one launch (include/bjx.h, bjx_graph_*)
mutable struct Graph
    h::Ptr{Cvoid}
end
function capture(f)
    check(ccall((:bjx_graph_begin, libbjx), Cint, (Ptr{Cvoid},), ctx().h), "bjx_graph_begin")
    g = Ref{Ptr{Cvoid}}(C_NULL)
    try
        f()
    finally
        check(ccall((:bjx_graph_end, libbjx), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), ctx().h, g), "bjx_graph_end")
    end
    gr = Graph(g[])
    finalizer(x -> ccall((:bjx_graph_destroy, libbjx), Cint, (Ptr{Cvoid},), x.h), gr)
    return gr
end
replay(g::Graph) = check(ccall((:bjx_graph_launch, libbjx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx().h, g.h), "bjx_graph_launch")

# ---------------------------------------------------------------- logpdf / rand of a TransformedDistribution (SURVEY.md §8f f-3)
# src/transformed_distribution.jl:164-169 in ONE pass over y: the inverse chain, the whitening of the diagonal-normal
# base and the standard-normal density are ops of the same launch; the pre-image is not stored (y pointer = C_NULL).
diag_normal(d::Distributions.MvNormal) = d.Σ isa Union{Distributions.PDMats.PDiagMat,Distributions.PDMats.ScalMat}

# ---- the base density on device columns: the extension point for ANY base (the Python mirror's `TorchBase`).
# `base_logpdf(d, x::ROCMatrix) -> ROCVector` (one value per column).  Methods here: a diagonal MvNormal (one fused launch: whitening
# + standard-normal density as ops of bjx_chain), a full-covariance MvNormal (whitening = the matrix Scale of scale.jl:14-36 with
# L⁻¹, then the same density op), and the fall-back: Distributions' own `logpdf(d, x)` on the device array (generic array code of
# the base — correct for every distribution whose logpdf is written with array operations; define a method for yours otherwise).
function stdnormal_chain(x::ROCMatrix{T}, pre::Vector{BjxOp}, keep) where {T<:BjxFloat}
    d, n = size(x); lp = similar(x, T, n)
    o = vcat(pre, BjxOp(Int32(OP_STDNORMAL_LOGPDF), 0, 0, 0, C_NULL, C_NULL))
    GC.@preserve keep x lp o check(ccall((:bjx_chain, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), o, length(o), devptr(x), C_NULL, devptr(lp), C_NULL, d, n, UInt32(0)), "bjx_chain")
    return lp
end
# bjx_scale_matrix_chain (include/bjx.h): `pre` (at most four stages of exp / log / Shift / Scale — the inverse of the transform and the shift
# by the mean) applied to each tile of `y` as the matrix-core kernel loads it, whitening by L⁻¹ and log N(z; 0, I) − logabsdet L per column
# while the tile is in LDS: src/transformed_distribution.jl:164-169 in one pass over y.  `nothing` when the library answers
# BJX_ERR_UNSUPPORTED (other stages, dim > 128 or not whole 16-byte packs, BJX_SCALE_MFMA=0): the caller runs bjx_chain + bjx_scale_matrix.
function full_cov_logpdf_fused(Lc::ROCMatrix{T}, pre::Vector{BjxOp}, y::ROCMatrix{T}, keep) where {T<:BjxFloat}
    (length(pre) <= 4 && all(o -> o.kind in (Int32(OP_EXP), Int32(OP_LOG), Int32(OP_SHIFT), Int32(OP_SCALE), Int32(OP_SCALE_INV)), pre)) || return nothing
    d, n = size(y); lp = similar(y, T, n)
    rc = GC.@preserve keep Lc y lp pre ccall((:bjx_scale_matrix_chain, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), Cint(1), devptr(Lc), pre, Cint(length(pre)), devptr(y), C_NULL, devptr(lp), d, n, BJX_BASE_STDNORMAL)
    rc == 0 && return lp
    rc == BJX_ERR_UNSUPPORTED || check(rc, "bjx_scale_matrix_chain")
    return nothing
end
function base_logpdf(d::Distributions.MvNormal, x::ROCMatrix{T}) where {T<:BjxFloat}
    keep = Any[]
    if diag_normal(d)
        σ = sqrt.(Array(Distributions.PDMats.diag(d.Σ)))
        return stdnormal_chain(x, [param_op(OP_SHIFT, -d.μ, T, keep), param_op(OP_SCALE_INV, σ, T, keep)], keep)
    end
    # Σ = L Lᵀ: z = L⁻¹ (x − μ), log N(x; μ, Σ) = log N(z; 0, I) − logabsdet L
    Lc = ROCArray{T}(Matrix(Distributions.PDMats.cholesky(d.Σ).L)); push!(keep, Lc)
    o = [param_op(OP_SHIFT, -d.μ, T, keep)]
    lp1 = full_cov_logpdf_fused(Lc, o, x, keep)                  # ONE launch when the matrix-core kernel serves the shape
    lp1 === nothing || return lp1
    xc = similar(x)
    GC.@preserve keep x xc o check(ccall((:bjx_chain, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), o, length(o), devptr(x), devptr(xc), C_NULL, C_NULL, size(x, 1), size(x, 2), UInt32(0)), "bjx_chain")
    p = plan_scale_matrix(Lc, true, xc)
    if size(x, 1) <= 128
        # ONE launch: whitening on the matrix cores, log N(z; 0, I) − logabsdet L accumulated per column while the tile is in LDS,
        # nothing stored (BJX_BASE_STDNORMAL on bjx_scale_matrix, include/bjx.h).  The flag is served by the matrix-core kernel only:
        # Float64 beyond 112 rows (its tile would need 194 KiB of LDS) and any run with BJX_SCALE_MFMA=0 answer BJX_ERR_UNSUPPORTED —
        # then fall through to the two-launch path below, as the Python mirror does (`_logpdf_full_cov_fused`; ADVICE r05).
        lp = similar(x, T, size(x, 2))
        rc = GC.@preserve keep xc lp p.launch(C_NULL, devptr(lp), C_NULL, BJX_BASE_STDNORMAL)
        rc == 0 && return lp
        rc == BJX_ERR_UNSUPPORTED || check(rc, "bjx_scale_matrix")
    end
    lj = run!(column_plan(p), T, xc, xc)                         # z in place; per-column −logabsdet L
    return stdnormal_chain(xc, BjxOp[], keep) .+ lj
end
base_logpdf(d::Distributions.Distribution, x::ROCMatrix) = Distributions.logpdf(d, x)

# The affine stages at the END of a density chain — the tail of the inverse transform (Shift / Scale / Scale⁻¹ with host scalars) followed by the
# whitening (x − μ)/σ — as ONE Scale and ONE Shift with per-row vectors: ((x + s)·c − μ)/σ = x·(c/σ) + (s·c − μ)/σ.  The chain kernel is
# issue-bound on read-only passes (six stages 50 % of the HBM peak, four 64 %: profiles/r06_rows.md; `_merge_affine_tail` of the Python mirror);
# the log-det is unchanged (Σ log|c/σ| is what the Scale stages add up to).
function merge_affine_tail(o::Vector{BjxOp}, μ::AbstractVector, σ::AbstractVector, ::Type{T}, keep) where {T}
    affine = (Int32(OP_SHIFT), Int32(OP_SCALE), Int32(OP_SCALE_INV))
    k = length(o)
    while k > 0 && o[k].kind in affine && o[k].param_len == 1 && o[k].v0 == C_NULL
        k -= 1
    end
    A = ones(Float64, length(μ)); B = zeros(Float64, length(μ))
    for j in (k + 1):length(o)
        p = o[j].p0
        if o[j].kind == Int32(OP_SHIFT)
            B .+= p
        elseif o[j].kind == Int32(OP_SCALE)
            A .*= p; B .*= p
        else
            A ./= p; B ./= p
        end
    end
    B .-= μ; A ./= σ; B ./= σ
    return vcat(o[1:k], param_op(OP_SCALE, A, T, keep), param_op(OP_SHIFT, B, T, keep))
end

function Distributions.logpdf(td::Bijectors.MvTransformed, y::ROCMatrix{T}) where {T<:BjxFloat}
    if td.dist isa Distributions.MvNormal && diag_normal(td.dist)
        keep = Any[]
        o = ops(inverse(td.transform), T, keep)
        d, n = dims(y)
        μ, σ = td.dist.μ, sqrt.(Array(Distributions.PDMats.diag(td.dist.Σ)))
        if o !== nothing && length(o) + 3 <= 8
            # ONE pass over y: the inverse chain, the whitening and the standard-normal density are ops of the same launch;
            # the pre-image is not stored (src/transformed_distribution.jl:164-169 without its two intermediate arrays)
            return stdnormal_chain(y, merge_affine_tail(o, μ, σ, T, keep), keep)
        end
        # a PlanarLayer flow with a standard-normal base: the inverse flow with BJX_BASE_STDNORMAL, pre-image not stored
        if all(iszero, μ) && all(isone, σ)
            tr = td.transform
            pcs = tr isa ComposedFunction ? pieces(tr) : Any[tr]                         # l8 ∘ … ∘ l1: the planner's single forward run
            layers = length(pcs) != 1 ? nothing : pcs[1] isa PlanarLayer ? PlanarLayer[pcs[1]] : (pcs[1] isa PlanarRun && !pcs[1].inv ? pcs[1].layers : nothing)
            if layers !== nothing
                p = plan_planar(layers, true, y, BJX_BASE_STDNORMAL)
                return run!(p, T, y, nothing)
            end
        end
    end
    if td.dist isa Distributions.MvNormal && !diag_normal(td.dist)
        # full covariance and an elementwise inverse of at most three stages: inverse chain, shift by the mean, whitening and density in ONE
        # launch over y (bjx_scale_matrix_chain); anything the library does not serve falls through to the general path below
        keep = Any[]
        o = ops(inverse(td.transform), T, keep)
        if o !== nothing && length(o) + 1 <= 4
            Lc = ROCArray{T}(Matrix(Distributions.PDMats.cholesky(td.dist.Σ).L)); push!(keep, Lc)
            lp1 = full_cov_logpdf_fused(Lc, vcat(o, param_op(OP_SHIFT, -td.dist.μ, T, keep)), y, keep)
            lp1 === nothing || return lp1
        end
    end
    # any base, any planned transform: x, logjac = with_logabsdet_jacobian(inverse(td.transform), y), per COLUMN (a scalar log-det
    # — the reference's return shape for elementwise transforms, :165-169 and its TODO — would be added to every column), then the
    # base density on the device columns
    itr = inverse(td.transform)
    pcs = itr isa ComposedFunction ? pieces(itr) : Any[itr]
    cur = y; total = nothing
    for pc in pcs
        cur, l = piece_wlj_columns(pc, cur)
        total = add_ladj(total, l)
    end
    total isa Number && !iszero(total) && throw(ArgumentError("logpdf(td, y::ROCMatrix): the transform $(typeof(td.transform)) has no per-column log-det on the device"))
    lp = base_logpdf(td.dist, cur)
    return total isa Number ? lp : lp .+ total
end
# the reference's Dirichlet method (:171-177: the base density at x + ε) would be ambiguous with the method above on a ROCMatrix
function Distributions.logpdf(td::Bijectors.MvTransformed{<:Distributions.Dirichlet}, y::ROCMatrix{T}) where {T<:BjxFloat}
    itr = inverse(td.transform)                                                 # per-column log-dets of the planned pieces
    pcs = itr isa ComposedFunction ? pieces(itr) : Any[itr]
    cur = y; total = nothing
    for pc in pcs
        cur, l = piece_wlj_columns(pc, cur)
        total = add_ladj(total, l)
    end
    lp = base_logpdf(td.dist, cur .+ Bijectors._eps(T))
    return total isa Number ? lp : lp .+ total
end

# rand(rng, td, n) on the device (src/transformed_distribution.jl:214-224; the reference's own signature).  `BjxRNG` is the device
# generator: the counter-based Philox stream of the library keyed by (seed, GLOBAL column, row) — identical for any shard count;
# `col0` is this rank's first global column.  A fusable transform on a diagonal-normal base draws its samples INSIDE the
# transforming launch (never written); every other planned transform (flows, splines, Stacked, …) and a full covariance:
# bjx_fill_normal, colouring, then the transform over all columns (what the Python mirror's `rand` does).
struct BjxRNG <: Random.AbstractRNG
    seed::UInt64
    col0::Int64
end
BjxRNG(seed::Integer=0; col0::Integer=0) = BjxRNG(UInt64(seed), Int64(col0))
function base_rand(rng::BjxRNG, d::Distributions.MvNormal, ::Type{T}, n::Integer) where {T<:BjxFloat}
    z = fill_normal!(ROCArray{T}(undef, length(d), n); col0=rng.col0, seed=rng.seed)
    keep = Any[]
    if diag_normal(d)
        σ = sqrt.(Array(Distributions.PDMats.diag(d.Σ)))
        o = [param_op(OP_SCALE, σ, T, keep), param_op(OP_SHIFT, d.μ, T, keep)]
    else
        Lc = ROCArray{T}(Matrix(Distributions.PDMats.cholesky(d.Σ).L)); push!(keep, Lc)
        run!(plan_scale_matrix(Lc, false, z), T, z, z; want_ladj=false)          # x = L z in place
        o = [param_op(OP_SHIFT, d.μ, T, keep)]
    end
    GC.@preserve keep z o check(ccall((:bjx_chain, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
        ctx().h, dtype(T), o, length(o), devptr(z), devptr(z), C_NULL, C_NULL, size(z, 1), n, UInt32(0)), "bjx_chain")
    return z
end
# any other base: its own sampler on the host generator seeded from the stream's seed, uploaded (define `base_rand` for yours to stay on the device)
base_rand(rng::BjxRNG, d::Distributions.Distribution, ::Type{T}, n::Integer) where {T<:BjxFloat} =
    ROCArray{T}(rand(Random.Xoshiro(rng.seed + UInt64(rng.col0)), d, n))
Base.rand(rng::BjxRNG, td::Bijectors.MvTransformed, n::Int) = rand(rng, td, n, Float32)       # the reference's signature (:215); Float32 columns
function Base.rand(rng::BjxRNG, td::Bijectors.MvTransformed, n::Int, ::Type{T}) where {T<:BjxFloat}
    d = td.dist
    if d isa Distributions.MvNormal && diag_normal(d)
        keep = Any[]
        o = ops(td.transform, T, keep)
        if o !== nothing && length(o) + 2 <= 8
            μ, σ = d.μ, sqrt.(Array(Distributions.PDMats.diag(d.Σ)))
            o = vcat(param_op(OP_SCALE, σ, T, keep), param_op(OP_SHIFT, μ, T, keep), o)
            y = ROCArray{T}(undef, length(μ), n)
            check(ccall((:bjx_set_rng, libbjx), Cint, (Ptr{Cvoid}, UInt64, Int64), ctx().h, rng.seed, rng.col0), "bjx_set_rng")
            GC.@preserve keep y o check(ccall((:bjx_chain, libbjx), Cint,
                (Ptr{Cvoid}, Cint, Ptr{BjxOp}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt32),
                ctx().h, dtype(T), o, length(o), C_NULL, devptr(y), C_NULL, C_NULL, length(μ), n, BJX_INPUT_STDNORMAL), "bjx_chain")
            return y
        end
    end
    x = base_rand(rng, d, T, n)
    return td.transform === identity ? x : transform(td.transform, x)               # a planned flow: ONE launch over all columns (:215-224 loops over columns)
end
# round-4 spelling, kept for callers of that round
rand_transformed(td::Bijectors.MvTransformed, ::Type{T}, n::Integer; seed::Integer=0, col0::Integer=0) where {T<:BjxFloat} = rand(BjxRNG(seed; col0=col0), td, Int(n), T)

# ---------------------------------------------------------------- multi-GPU (one process per GPU)
comm_unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:bjx_comm_unique_id, libbjx), Cint, (Ptr{Cvoid},), id), "bjx_comm_unique_id"); id)
comm_init(nranks, rank, id::Vector{UInt8}) = check(ccall((:bjx_comm_init, libbjx), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), ctx().h, nranks, rank, id), "bjx_comm_init")
comm_destroy() = check(ccall((:bjx_comm_destroy, libbjx), Cint, (Ptr{Cvoid},), ctx().h), "bjx_comm_destroy")
allreduce_logabsdetjac!(partial::ROCVector{Float64}) = (check(ccall((:bjx_allreduce_sum_f64, libbjx), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64), ctx().h, Ptr{Cdouble}(pointer(partial)), length(partial)), "bjx_allreduce_sum_f64"); partial)

# ---------------------------------------------------------------- measurement helpers (bench/reference_cpu.jl's GPU leg)
function fill_normal!(out::ROCVecOrMat{T}; col0::Integer=0, seed::Integer=0, mean::Real=0, std::Real=1) where {T<:BjxFloat}
    d, n = dims(out)
    GC.@preserve out check(ccall((:bjx_fill_normal, libbjx), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Int64, Int64, UInt64, Cdouble, Cdouble),
        ctx().h, dtype(T), devptr(out), d, n, Int64(col0), UInt64(seed), Float64(mean), Float64(std)), "bjx_fill_normal")
    return out
end
# milliseconds of the stream region `f()`; and the summed DOMINANT-kernel milliseconds + launch count of the same region
function timed(f)
    check(ccall((:bjx_time_begin, libbjx), Cint, (Ptr{Cvoid},), ctx().h), "bjx_time_begin")
    f()
    ms = Ref{Cfloat}(0)
    check(ccall((:bjx_time_end, libbjx), Cint, (Ptr{Cvoid}, Ptr{Cfloat}), ctx().h, ms), "bjx_time_end")
    return ms[]
end
function kernel_timed(f)
    check(ccall((:bjx_kernel_time_begin, libbjx), Cint, (Ptr{Cvoid},), ctx().h), "bjx_kernel_time_begin")
    f()
    ms = Ref{Cfloat}(0); k = Ref{Cint}(0)
    check(ccall((:bjx_kernel_time_end, libbjx), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cint}), ctx().h, ms, k), "bjx_kernel_time_end")
    return ms[], k[]
end

end # module
