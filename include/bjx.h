/*
 * bjx.h — C ABI of libbjx_hip.so: the MI355X (gfx950) implementation of Bijectors.jl's
 * batched transform + log-abs-det-Jacobian hot path.
 *
 * The reference (Bijectors.jl v0.16.2) has no FFI: its extension point is Julia multiple
 * dispatch on `transform / logabsdetjac / with_logabsdet_jacobian / inverse`
 * (src/interface.jl:156-218, :265-281).  Every entry below is what a Julia method
 *     with_logabsdet_jacobian(b::<Bijector>, x::ROCArray)
 * would `ccall`; the reference function each entry replaces is cited on the entry.
 * The Julia-side binding is shown in INTEGRATION.md and julia/BijectorsBJX.jl.
 *
 * Conventions (all entries)
 *  - Arrays are Julia column-major X[dim, batch]: `dim` is the contiguous axis, one sample
 *    (column) = `dim` consecutive elements, leading dimension == dim.  batch may be 1 (a
 *    Julia Vector).  All array pointers are DEVICE pointers owned by the caller; the
 *    library never allocates or frees caller memory and allocates nothing per call (a small
 *    scratch for reduction partials lives inside the context).
 *  - `dt` selects Float32 / Float64 for every `void*` data/parameter pointer of the call.
 *  - `ladj_ps`  : optional T[batch] per-sample (per-column) log|det J|   (may be NULL)
 *    `ladj_sum` : optional double[1] sum over the batch of the above      (may be NULL)
 *    With BJX_ACCUMULATE both are added to instead of overwritten — this is the
 *    `with_logabsdet_jacobian!(b, x, y, logjac)` contract (src/interface.jl:212-218).
 *    The reference's per-bijector return shape (scalar vs per-column vector,
 *    SURVEY.md §8a') is rebuilt by the host wrapper from these two outputs.
 *  - Calls are asynchronous on the context's HIP stream.  A context is not thread-safe;
 *    distinct contexts are independent.  No exception crosses the ABI.
 *  - Return value: 0 ok; <0 argument/shape error (mirrors the reference's precondition
 *    checks: simplex.jl:30, normalise.jl:43-45, corr.jl:215-219, rational_quadratic_spline.jl:84-85);
 *    >0 a hipError_t (or 1000+ncclResult_t).  bjx_last_error(ctx) gives the message.
 */
#ifndef BJX_H
#define BJX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BJX_VERSION 100

typedef struct bjx_ctx bjx_ctx;

typedef enum { BJX_F32 = 0, BJX_F64 = 1 } bjx_dtype;

/* error codes (<0) */
enum {
  BJX_OK = 0,
  BJX_ERR_ARG = -1,       /* NULL / negative size / bad enum  (Julia: ArgumentError)      */
  BJX_ERR_SHAPE = -2,     /* size mismatch                    (Julia: DimensionMismatch)  */
  BJX_ERR_UNSUPPORTED = -3,
  BJX_ERR_NOCOMM = -4,    /* collective requested without bjx_comm_init                  */
  /* ASYNCHRONOUS, reported like a hipError_t by the first entry point (or bjx_synchronize / bjx_check_state) that runs after the
   * fact: the in-kernel finalize of an EARLIER launch on this context gave up waiting (see BJX_OPT_INKERNEL_FINALIZE), or a hand-off
   * slot was found dirty.  The `ladj_sum` of that launch, and of the launches enqueued after it on the context until the report, is
   * NaN — never a plausible number.  The reporting call launches nothing, re-arms the slots in stream order and switches the
   * context to the two-pass finalize for good; the caller repeats its call.  (Julia: ErrorException) */
  BJX_ERR_FINALIZE = -5
};

/* flags */
enum {
  BJX_ACCUMULATE = 1u << 0,          /* add into ladj_ps / ladj_sum instead of overwriting   */
  /* scale.jl:31-32: for a VECTOR-parameter Scale applied to a d x N matrix the reference
   * returns sum_i log|a_i| (NOT multiplied by N).  With this flag the chain's `ladj_sum`
   * reproduces that value; without it the mathematically consistent N * sum_i log|a_i| is
   * returned.  `ladj_ps` is always the true per-sample value. */
  BJX_REF_VECTOR_SCALE_LADJ = 1u << 1,
  /* SURVEY.md §8(f) f-3 — logpdf(td::MvTransformed, Y) in ONE pass (src/transformed_distribution.jl:164-169,
   * "TODO: implement more efficiently for flows"): add the standard-normal log-density of every OUTPUT column,
   * -1/2 |out|^2 - dim/2 log(2 pi), to ladj_ps / ladj_sum.  With the inverse flow as the bijector the call
   * returns logpdf(td, y) per column; `out` may then be NULL (the pre-image is not stored: half the traffic).
   * Honoured by bjx_chain (same as appending BJX_OP_STDNORMAL_LOGPDF) and bjx_planar. */
  BJX_BASE_STDNORMAL = 1u << 2,
  /* On-device sampling (rand(td, n), src/transformed_distribution.jl:214-224): the INPUT of bjx_chain is not read but
   * drawn — standard normals from the counter-based generator of bjx_fill_normal, stream (seed, first global column)
   * set with bjx_set_rng; `x` may be NULL.  The values are bit-identical to bjx_fill_normal followed by the chain,
   * for any shard count. */
  BJX_INPUT_STDNORMAL = 1u << 3,
  /* bjx_coupling_affine: the `scale` / `shift` array is T[n1] shared by every column instead of T[n1, batch] (a law whose
   * parameter does not depend on x_2, e.g. Shift(0.25) or Scale(vector): nothing is expanded on the host). */
  BJX_COUPLING_SCALE_BCAST = 1u << 4,
  BJX_COUPLING_SHIFT_BCAST = 1u << 5
};

/* ---------------------------------------------------------------- context */
/* `hip_stream` is a hipStream_t (NULL = default stream).  AMDGPU.jl: AMDGPU.stream().stream  */
int bjx_create(int device, void* hip_stream, bjx_ctx** out);
int bjx_destroy(bjx_ctx* ctx);
int bjx_set_stream(bjx_ctx* ctx, void* hip_stream);
const char* bjx_last_error(bjx_ctx* ctx);
int bjx_version(void);
/* bytes of scratch a context holds (informational; SURVEY.md §8b "bjx_workspace_bytes") */
size_t bjx_workspace_bytes(bjx_ctx* ctx);
int bjx_synchronize(bjx_ctx* ctx);
/* Tuning switches (per context).  BJX_OPT_INKERNEL_FINALIZE: who finishes the deterministic sum `ladj_sum` of a call.
 *   2 (default since round 5) = sentinel hand-off inside the hot kernel: ONE launch per call (grids of <= 65 536 blocks; larger
 *       grids and the kernels that do not carry the epilogue fall back to 0).  Blocks publish their partial with one atomic
 *       8-byte store and never wait; group-closing blocks and the last block poll slots that hold a sentinel between launches
 *       (csrc/bjx_internal.h, block_publish_sentinel).  Fixed order: run-to-run identical bits.
 *   1 = arrival ticket (round 3): the last block to ARRIVE sums (<= 4 096 blocks); every block waits for its own output stores
 *       before it draws the ticket, which costs short kernels more than the launches it saves (profiles/r03_finalize_ab.txt).
 *   0 = two small follow-up launches (the partials reduced by bjx_finalize*_kernel).
 * 0 and 1 give the same bits; 2 sums in a different (fixed) order: equal to <= 1e-15 relative.
 * ASSUMPTION of mode 2, stated here because nothing in HIP guarantees it: within one grid the dispatcher starts blocks of lower index
 * no later than blocks of higher index, so a block that waits for lower-indexed blocks never holds a slot they need.  True on every
 * queue configuration of this pool; CU masking, priority pre-emption or a debugger could break it.  If it breaks the wait times out
 * (~1 s), that launch's sum is NaN and the context reports BJX_ERR_FINALIZE at the next call, re-arms and stays on mode 0.
 * A context must not have launches in flight on two streams at once (its scratch is shared): bjx_set_stream orders the new stream
 * after the work already enqueued on the old one. */
enum {
  BJX_OPT_INKERNEL_FINALIZE = 1,
  /* Watchdog of the library's own collective (bjx_comm_init + bjx_allreduce_sum_f64): with a communicator attached,
   * bjx_synchronize polls the stream for at most `value` milliseconds; on time-out it aborts the communicator (ncclCommAbort) and
   * returns 1000 + ncclRemoteError instead of hanging on a rank that never arrived.  0 (default) = wait for ever. */
  BJX_OPT_COLLECTIVE_TIMEOUT_MS = 2,
  /* Parameter epoch.  Some entries derive a table from their parameter arrays with a helper launch before the hot kernel (bjx_rqs:
   * the spline's 17 - 64 KiB LDS blob of search keys and per-bin records; bjx_scale_matrix: [A^-1 | logabsdet A]).  With value != 0
   * the library keeps such tables, keyed by the parameter POINTERS and shapes, and reuses them while the epoch is unchanged: the
   * host promises that the memory behind a pointer it passes again has not been written since the epoch was set, and sets a
   * different non-zero epoch (or 0) whenever it may have been (an optimiser step, a new array at a recycled address).  0 (default) =
   * rebuild on every call — the safe setting for hosts that cannot track writes (arrays mutated in place without a version counter). */
  BJX_OPT_PARAM_EPOCH = 3,
  /* Fault injection for the tests of the failure channel (tests/test_gpu_fin_faults.py), never set by a host binding:
   * DROP_BLOCK: the block with this index does not publish its partial in sentinel-mode launches (value < 0: off) — the closing block
   * of its group times out; POISON_SLOT: writes 1.0 into hand-off slot `value` now (what an aborted launch would leave behind). */
  BJX_OPT_DEBUG_FIN_DROP_BLOCK = 4,
  BJX_OPT_DEBUG_FIN_POISON_SLOT = 5
};
int bjx_set_option(bjx_ctx* ctx, int option, int value);
/* Diagnostics: synchronises the stream and verifies that the context's hand-off state is clean (every slot holds its sentinel, the
 * arrival counter is zero, no time-out is pending).  BJX_OK, or BJX_ERR_FINALIZE after repairing the state (see the error code).  For
 * hosts that caught a hipError_t of their own, and for tests; bjx_synchronize reports time-outs without the extra launch. */
int bjx_check_state(bjx_ctx* ctx);
/* Stream of BJX_INPUT_STDNORMAL: element (row, col) of a call draws value number (col0 + col) * dim + row of `seed`. */
int bjx_set_rng(bjx_ctx* ctx, uint64_t seed, int64_t col0);

/* ------------------------------------------- F1: fused elementwise chains */
/* One entry of a `ComposedFunction` chain (src/bijectors/composed.jl:4-25) after the host has
 * walked `outer ∘ inner` into application order (ops[0] is applied first).
 * Inverse bijectors are their own op kinds (the host maps inverse(b) -> kind). */
typedef enum {
  BJX_OP_EXP = 1,            /* elementwise(exp)        interface.jl:6,33; exp_log.jl:5-6     */
  BJX_OP_LOG = 2,            /* elementwise(log)        exp_log.jl:8-9                        */
  BJX_OP_SHIFT = 3,          /* Shift(a): a .+ x        shift.jl:14,21   (inverse: a := -a)   */
  BJX_OP_SCALE = 4,          /* Scale(a): a .* x        scale.jl:13,26-32                     */
  BJX_OP_SCALE_INV = 5,      /* Inverse{Scale}: inv(a) .* y          scale.jl:15-16           */
  BJX_OP_LOGIT = 6,          /* Logit(a,b)              logit.jl:15,24-30                     */
  BJX_OP_LOGIT_INV = 7,      /* Inverse{Logit}          logit.jl:19-21 + interface.jl:276-281 */
  BJX_OP_LEAKY_RELU = 8,     /* LeakyReLU(alpha)        leaky_relu.jl:18-29 (inverse: 1/alpha)*/
  BJX_OP_TRUNCATED = 9,      /* TruncatedBijector(lb,ub)          truncated.jl:15-31,51-67    */
  BJX_OP_TRUNCATED_INV = 10, /* Inverse{TruncatedBijector}        truncated.jl:33-49,71-91    */
  BJX_OP_SIGNFLIP = 11,      /* SignFlip                ordered.jl:3                          */
  BJX_OP_IDENTITY = 12,
  /* not a bijector: leaves the value unchanged and adds log N(v; 0, 1) = -v^2/2 - log(2 pi)/2 to the log-det
   * accumulator — the base density of a TransformedDistribution evaluated inside the same pass
   * (src/transformed_distribution.jl:165-169).  A diagonal MvNormal(mu, sigma) base is
   * SHIFT(-mu), SCALE_INV(sigma), STDNORMAL_LOGPDF (the -sum log sigma comes from SCALE_INV's log-det). */
  BJX_OP_STDNORMAL_LOGPDF = 13
} bjx_op_kind;

typedef struct {
  int32_t kind;      /* bjx_op_kind                                                          */
  int32_t param_len; /* 0: no parameter; 1: scalar; ==dim: one value per row                 */
  double p0, p1;     /* host scalars used when v0/v1 == NULL (a | a,b | alpha | lb,ub)       */
  const void* v0;    /* device T[param_len] (overrides p0) or NULL                           */
  const void* v1;    /* device T[param_len] (overrides p1) or NULL                           */
} bjx_op;

#define BJX_MAX_OPS 8

/* y = (ops[n-1] ∘ ... ∘ ops[0])(x) and the summed log|det J| in ONE pass over x.
 * Replaces the 3+ allocating passes of SURVEY.md §3.1.  y may alias x (transform!). */
int bjx_chain(bjx_ctx* ctx, bjx_dtype dt, const bjx_op* ops, int n_ops,
              const void* x, void* y, void* ladj_ps, double* ladj_sum,
              int64_t dim, int64_t batch, uint32_t flags);

/* --------------------------------- F3: sequential / prefix along `dim`    */
/* OrderedBijector, ordered.jl:24-80.  inverse=0: x1=y1, xi=x(i-1)+exp(yi); ladj[n]=sum_{i>=2} y[i,n].
 * inverse=1: y1=x1, yi=log(xi-x(i-1)); ladj = -sum_{i>=2} y_out[i,n] (interface.jl:276-281). */
int bjx_ordered(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out,
                void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* SimplexBijector, simplex.jl:28-143.  K = rows of the simplex side.
 * inverse=0: in  X[K,N]   -> out Y[K-1,N], ladj = logabsdetjac(b, X)      (:122-143)
 * inverse=1: in  Y[K-1,N] -> out X[K,N],   ladj = -logabsdetjac(b, X_out) (interface.jl:276-281)
 * `out` may be NULL to compute only the log-det (logabsdetjac(b, x)); K < 2 -> BJX_ERR_SHAPE. */
int bjx_simplex(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out,
                void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags);

/* VecCholeskyBijector, corr.jl:227-254 (+ :314-337, :370-399, :485-501), batched over samples.
 * K x K Cholesky factor of a correlation matrix <-> vector of n = K(K-1)/2 unconstrained reals.
 * uplo = 'U' or 'L' (corr.jl:215-219, anything else -> BJX_ERR_ARG).
 * inverse=1: in y[n,N] -> out W[K,K,N] dense column-major factor (upper filled, lower zero for
 *            'U'; transposed for 'L'), ladj = logJ of _inv_link_chol_lkj.
 * inverse=0: in W[K,K,N] -> out y[n,N], ladj = -_logabsdetjac_inv_chol(y) (corr.jl:235-237).
 * `out` may be NULL with inverse=1 to compute only logabsdetjac(inverse(b), y) (:252-254). */
int bjx_vec_cholesky(bjx_ctx* ctx, bjx_dtype dt, int inverse, int uplo, const void* in, void* out,
                     void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags);

/* ------------------------------- matrix-variate constraint bijectors (per-sample Cholesky) */
/* What bijector(::LKJ) / bijector(::Wishart-family) return.  X is a dense K x K column-major matrix per sample
 * (X[K,K,batch]); K <= 64 (one wave per sample, the factor's rows in registers; larger K -> BJX_ERR_UNSUPPORTED).
 * The triangle of X that is READ is the one the reference's Cholesky wrapper reads (src/utils.jl:37,50):
 * the UPPER one for the correlation bijectors (cholesky(Hermitian(X)).U), the LOWER one for the PD bijectors
 * (cholesky(Hermitian(X, :L)).L).  The inverse writes the full symmetric matrix (pd_from_upper / pd_from_lower,
 * src/utils.jl:17-24).  `out` may be NULL to compute only the log-det.
 *
 * bjx_vec_corr — VecCorrBijector, corr.jl:128-162.
 *   inverse=0: X[K,K,N] -> y[K(K-1)/2, N] = _link_chol_lkj_from_upper(cholesky_upper(X)) (:133);
 *              ladj = -_logabsdetjac_inv_corr(y) (:135-137, :463-472)
 *   inverse=1: y -> X = U'U with U, logJ = _inv_link_chol_lkj(y); ladj = logJ + sum_{j=2}^{K-1} (K-j) log U[j,j] (:139-148) */
int bjx_vec_corr(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out,
                 void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags);
/* bjx_corr — CorrBijector, corr.jl:64-92: the same maps with the unconstrained side a K x K matrix Y whose strict upper
 * triangle holds the free values (zeros on and below the diagonal, :292-294); inverse reads only that triangle (:345-368). */
int bjx_corr(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out,
             void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags);
/* bjx_pd — PDBijector, pd.jl:1-36.
 *   inverse=0: X -> Y = replace_diag(log, cholesky_lower(X)) (lower triangular, zeros above);
 *              ladj = -(sum_i (K+2-i) log L_ii + K log 2) (:27-31)
 *   inverse=1: Y -> X = L L', L = lower_triangular(replace_diag(exp, Y)) (:13-16); ladj = -(forward ladj at X) (interface.jl:278-281) */
int bjx_pd(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out,
           void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags);
/* bjx_pd_vec — PDVecBijector, pd.jl:38-60: the same with the unconstrained side packed,
 * y[K(K+1)/2, N] = triu_to_vec(transpose(pd_link(X))) (:41; column-major upper triangle with the diagonal). */
int bjx_pd_vec(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out,
               void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags);

/* SURVEY.md §8(f) f-1 x f-4: pullbacks of the four matrix-variate bijectors above, either direction:
 *   in_bar = J(in)' out_bar + ladj_bar[n] * grad logabsdetjac(in)       (ladj_bar: T[batch] or NULL = 0)
 * inverse=1 (unconstrained -> matrix, what HMC on an LKJ / Wishart / covariance prior differentiates every leapfrog step):
 *   in = y (packed T[n, batch] or dense T[K, K, batch], the layouts of the forward entries), out_bar = X̄ dense T[K, K, batch]
 *   (any matrix, not assumed symmetric), in_bar = ȳ in the layout of y.  The rules the reference ships, chained per sample:
 *   pd_from_upper / pd_from_lower (ext/BijectorsChainRulesCoreExt.jl:324-331, ext/BijectorsReverseDiffExt.jl:160-168), then
 *   _inv_link_chol_lkj's reverse sweep (corr.jl:402-451) + the (K-j) log U[j,j] terms of corr.jl:77-79, or replace_diag(exp)
 *   (ext/BijectorsReverseDiffExt.jl:153-158) + the weights of pd.jl:27-31.
 * inverse=0 (matrix -> unconstrained): in = X dense, out_bar = ȳ in the layout of the forward output, in_bar = X̄ dense; the
 *   cotangent of the link (corr.jl:299-335, pd.jl:11) goes through the reverse of cholesky(Hermitian(X)) and lands on the
 *   triangle the reference READS (upper for the correlation bijectors, src/utils.jl:50; lower for PD, :37) — the other
 *   triangle of X̄ is zero.  in_bar may alias in.  K <= 12: one lane per sample, factor and cotangent in registers; larger K:
 *   the same code on a global workspace (correct, not fast). */
int bjx_vec_corr_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar,
                     void* in_bar, int64_t K, int64_t batch);
int bjx_corr_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar,
                 void* in_bar, int64_t K, int64_t batch);
int bjx_pd_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar,
               void* in_bar, int64_t K, int64_t batch);
int bjx_pd_vec_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar,
                   void* in_bar, int64_t K, int64_t batch);

/* Scale with a MATRIX parameter, scale.jl:14,17,35-36: out = a * in (inverse=1: out = a \ in), a: device T[dim, dim]
 * column-major, dim <= 128.  logabsdetjac = logabsdet(a) (negated for the inverse): ladj_ps[n] holds it for every column;
 * ladj_sum = batch * logabsdet(a), or logabsdet(a) ONCE with BJX_REF_VECTOR_SCALE_LADJ — the value the reference returns
 * for a matrix of columns (:36).  The LU (partial pivoting) behind logabsdet / the inverse runs on the device per call (kept per
 * BJX_OPT_PARAM_EPOCH).  BJX_BASE_STDNORMAL (dim <= 128, ladj_ps only, out may be NULL): ladj_ps[n] additionally receives
 * log N(out[:, n]; 0, I) — with inverse = 1 and a = the Cholesky factor L of a covariance, one launch turns x - mu into the
 * full-covariance normal log-density without storing the whitened values (src/transformed_distribution.jl:164-169). */
int bjx_scale_matrix(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* a, const void* in, void* out,
                     void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);
/* bjx_scale_matrix with an elementwise chain in front, in ONE pass over `in` (src/transformed_distribution.jl:164-169 with a
 * full-covariance base: the inverse of the transform, the shift by the mean, the whitening a \ (.) and — with BJX_BASE_STDNORMAL — the
 * standard-normal density, nothing stored in between):  out = a * c(in)  (inverse = 1: a \ c(in)),  c = ops[n_ops-1] ∘ … ∘ ops[0]
 * applied to every element as its tile is loaded.  Served stages: BJX_OP_EXP, BJX_OP_LOG, BJX_OP_SHIFT, BJX_OP_SCALE, BJX_OP_SCALE_INV with a host scalar
 * (param_len 1, p0) or one value per row (param_len == dim, v0); n_ops <= 4.  ladj_ps[n] = logabsdetjac of c at column n + logabsdet(a)
 * (negated for the inverse) (+ log N(out[:, n]; 0, I) with BJX_BASE_STDNORMAL; added to its content with BJX_ACCUMULATE); out may be
 * NULL when ladj_ps is wanted alone.  BJX_ERR_UNSUPPORTED (nothing launched) for any other stage, for dim > 128 or not a whole number
 * of 16-byte packs, for arrays off 16-byte boundaries and with BJX_SCALE_MFMA=0: the caller then runs bjx_chain and bjx_scale_matrix. */
int bjx_scale_matrix_chain(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* a, const bjx_op* ops, int n_ops, const void* in,
                           void* out, void* ladj_ps, int64_t dim, int64_t batch, uint32_t flags);
/* Parameter pullback of the matrix Scale (ext/BijectorsReverseDiffExt.jl:72-115; scale.jl:14,17,35-36):
 *     a_bar = sign * ( g x^T + (sum_n ladj_bar[n]) a^-T )        a_bar, a: T[dim, dim] column-major; g, x: T[dim, batch]
 * forward  y = a x     : g = y_bar, x = the input,            sign = +1   (the input cotangent is bjx_scale_matrix with a^T)
 * inverse  x = a \ y   : g = the INPUT cotangent a^-T x_bar,  x = a \ y, sign = -1
 * g x^T — a sum of `batch` outer products — runs on the matrix cores (v_mfma_{f32,f64}_16x16x4, 64 x 64 tiles, operands read straight from the
 * column-major arrays), the partial tiles are folded in a fixed order (deterministic); a^-T comes from the factorisation of bjx_scale_matrix.
 * ladj_bar may be NULL (no log-det cotangent: no factorisation).  dim <= 1024. */
int bjx_scale_matrix_vjp_params(bjx_ctx* ctx, bjx_dtype dt, const void* a, const void* g, const void* x, const void* ladj_bar, double sign, void* a_bar,
                                int64_t dim, int64_t batch);

/* ------------------------------- F2: per-sample reduce + broadcast        */
/* PlanarLayer, planar_layer.jl:65-127,160-185; `n_layers` stacked layers (composition
 * layer[n_layers-1] ∘ ... ∘ layer[0]) are fused into one pass over Z.
 * w,u: device T[dim*n_layers] (layer-major), b: device T[n_layers].
 * inverse=0: z' = z + û tanh(wᵀz+b), ladj[n] = Σ_layers log1p(wᵀû sech²(wᵀz+b)).
 * inverse=1: layers are undone last-to-first with find_alpha; ladj = -(forward ladj at the result).
 * Any column height and any alignment (like the reference, which has no limit): since round 5 this also holds for
 * bjx_planar_vjp, bjx_planar_vjp_params, bjx_radial_vjp and bjx_radial_vjp_params (which used to return BJX_ERR_UNSUPPORTED
 * beyond 8 192 / 4 096 rows, the parameter pullback beyond 1 024 / 512).  `out`, `in_bar` may alias `in` / `out_bar`. */
int bjx_planar(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* w, const void* u,
               const void* b, int n_layers, const void* in, void* out,
               void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* A flow written the reference's way — `l8 ∘ … ∘ l1` (docs/src/flows.md:115, src/bijectors/composed.jl:4-25) — is one
 * PlanarLayer object per layer, each with its own w, u, b device vectors.  The host's composition planner hands a maximal RUN
 * of PlanarLayer stages to ONE bjx_planar call; this entry builds the layer-major tables that call takes:
 *   dst[i*len + k] = src[i][k],  i < n, k < len       (src: HOST array of n device pointers to T[len]; dst: device T[n*len])
 * in one launch per 64 vectors (the pointers travel in the kernel arguments).  Used for w and u (len = dim) and b (len = 1);
 * the cotangent tables of bjx_planar_vjp_params are layer-major too, so layer i's w̄ is the slice [i*dim, (i+1)*dim). */
int bjx_pack_vectors(bjx_ctx* ctx, bjx_dtype dt, int n, const void* const* src, int64_t len, void* dst);

/* input pullback of with_logabsdet_jacobian for the fused PlanarLayer stack (the reference
 * leaves it to the AD package; closed-form derivatives of planar_layer.jl:65-127):
 *   in_bar = (d out/d in)^T out_bar + ladj_bar[n] * d logabsdetjac[n] / d in.
 * inverse=0: the flow itself (in = z).  inverse=1: inverse(flow) (in = y; the primal is re-solved with find_alpha and
 * differentiated with its implicit-function rule, ext/BijectorsChainRulesCoreExt.jl:42-46).
 * in: primal input [dim, batch]; out_bar: cotangent of the output [dim, batch]; ladj_bar: cotangent of the per-column
 * log-det, T[batch] or NULL.  Parameter gradients are not produced. */
int bjx_planar_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* w, const void* u, const void* b,
                   int n_layers, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar,
                   int64_t dim, int64_t batch);

/* The same pullback plus the PARAMETER cotangents of the stack, summed over the batch: w_bar, u_bar: T[dim*n_layers]
 * (layer-major like w, u), b_bar: T[n_layers] — including the chain rule through get_u_hat (planar_layer.jl:65-70).
 * `work`: caller-owned device scratch of 2*n_layers*batch elements of T (tanh and the cotangent of w^T z + b of every
 * layer and column).  A batch sharded over GPUs all-reduces the three outputs. */
int bjx_planar_vjp_params(bjx_ctx* ctx, bjx_dtype dt, const void* w, const void* u, const void* b, int n_layers,
                          const void* in, const void* out_bar, const void* ladj_bar, void* in_bar,
                          void* w_bar, void* u_bar, void* b_bar, void* work, int64_t dim, int64_t batch);

/* RadialLayer, radial_layer.jl:43-129.  alpha_, beta: device T[1]; z0: device T[dim]. */
int bjx_radial(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* alpha_, const void* beta,
               const void* z0, const void* in, void* out,
               void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* input pullback of with_logabsdet_jacobian for a RadialLayer (inverse=0) and its inverse
 * (inverse=1): closed-form derivatives of radial_layer.jl:43-129 (J = a I + c dd^T, inverted with Sherman-Morrison).
 * in: primal input [dim, batch]; out_bar [dim, batch]; ladj_bar T[batch] or NULL. */
int bjx_radial_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* alpha_, const void* beta, const void* z0,
                   const void* in, const void* out_bar, const void* ladj_bar, void* in_bar,
                   int64_t dim, int64_t batch);

/* The forward pullback plus the PARAMETER cotangents of the RadialLayer (radial_layer.jl:43-60; alpha_, beta are the
 * raw parameters behind softplus), summed over the batch: alpha_bar, beta_bar: T[1], z0_bar: T[dim].
 * `work`: caller-owned device scratch of 2*batch elements of T (r and (z - z0)^T out_bar of every column).
 * A batch sharded over GPUs all-reduces the three outputs.  dim <= 1024 (Float32) / 512 (Float64), the limit of
 * bjx_row_moments, which produces z0_bar. */
int bjx_radial_vjp_params(bjx_ctx* ctx, bjx_dtype dt, const void* alpha_, const void* beta, const void* z0,
                          const void* in, const void* out_bar, const void* ladj_bar, void* in_bar,
                          void* alpha_bar, void* beta_bar, void* z0_bar, void* work, int64_t dim, int64_t batch);

/* InvertibleBatchNorm in eval mode (istraining() == false), normalise.jl:41-88.
 * b, logs, m, v: device T[dim] (channels = dim for 2-D input, :43-47). */
int bjx_batchnorm(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* b, const void* logs,
                  const void* m, const void* v, double eps, const void* in, void* out,
                  void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* Row moments over the batch: out[i] = sum_n a[i,n], out[dim+i] = sum_n a[i,n]*b[i,n] (b NULL: a^2), out[2 dim] = batch;
 * out: device double[2*dim+1], Float64 accumulation in a fixed order.  With a = the input cotangent of a chain
 * `tail o Shift(mu) o Scale(sigma)` (bjx_stacked_vjp) and b = its input these are the parameter cotangents of the
 * leading per-row affine stage (mean-field families): mu_bar = out[i]/sigma, sigma_bar = (out[dim+i] + sum ladj_bar)/sigma.
 * A batch sharded over GPUs all-reduces `out` (bjx_allreduce_sum_f64). */
int bjx_row_moments(bjx_ctx* ctx, bjx_dtype dt, const void* a, const void* b, double* out, int64_t dim, int64_t batch);

/* InvertibleBatchNorm in TRAINING mode (istraining() == true), normalise.jl:51-60: the batch mean and the
 * biased batch variance of every channel replace m / v in the transform and the log-det, and the moving
 * statistics `m`, `v` (device T[dim], read AND written) are updated with momentum `mtm`
 * (bn.v with the n/(n-1) correction, :59).  When the context has a communicator (bjx_comm_init) the batch is
 * taken to be sharded over the ranks and the 2·dim+1 Float64 sums (Σx, Σx², n) are all-reduced once
 * (SURVEY.md §8e): every rank then normalises with the GLOBAL statistics.  The reference has no inverse in
 * training mode (:71). */
int bjx_batchnorm_train(bjx_ctx* ctx, bjx_dtype dt, const void* b, const void* logs, void* m, void* v,
                        double eps, double mtm, const void* in, void* out, void* ladj_ps,
                        double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* The two halves of bjx_batchnorm_train, for hosts that own the collective (and for shard emulation on one GPU):
 *   bjx_batchnorm_stats        stats[0..dim) = sum_n (x[i,n] - shift[i]), stats[dim..2 dim) = sum_n (x[i,n] - shift[i])^2,
 *                              stats[2 dim] = batch  — device double[2 dim + 1], Float64, fixed summation order.
 *                              `shift` (device T[dim] or NULL = 0) must be identical on every rank: pass the moving
 *                              mean `m`, which makes the one-pass variance as well conditioned as normalise.jl:54's two-pass form.
 *   (host)                     sum `stats` over the ranks (bjx_allreduce_sum_f64 / MPI / torch.distributed)
 *   bjx_batchnorm_train_apply  statistics from the GLOBAL sums (shift = the `m` passed in, read before it is updated),
 *                              moving-statistics update (:58-59), transform and log-det of THIS rank's columns.
 * bjx_batchnorm_train(x) == stats(m, x) -> all-reduce when the context has a communicator -> train_apply. */
int bjx_batchnorm_stats(bjx_ctx* ctx, bjx_dtype dt, const void* shift, const void* in, double* stats, int64_t dim, int64_t batch);
int bjx_batchnorm_train_apply(bjx_ctx* ctx, bjx_dtype dt, const void* b, const void* logs, void* m, void* v,
                              double eps, double mtm, const double* stats, const void* in, void* out, void* ladj_ps,
                              double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* pullback of InvertibleBatchNorm in TRAINING mode (normalise.jl:51-60: the batch mean and the biased batch
 * variance are functions of x; the reference leaves the adjoint to the AD package).  With sigma = sqrt(var + eps),
 * xh = (in - mean)/sigma, gamma = exp(logs):
 *   in_bar   = gamma/sigma [out_bar - mean_n(out_bar) - xh mean_n(out_bar xh)] - (sum_n ladj_bar / N) xh / sigma
 *   b_bar    = sum_n out_bar,      logs_bar = gamma sum_n out_bar xh + sum_n ladj_bar
 * mean, var: device T[dim], the batch statistics the forward pass normalised with (from the sums of bjx_batchnorm_stats).
 * moments: device double[2 dim + 1] = (sum_n out_bar, sum_n out_bar*in, N) of the WHOLE batch: bjx_row_moments(out_bar, in) on
 * this rank's columns, summed over the ranks by the host — the one extra all-reduce of a sharded training step.
 * ladj_bar_sum: device double[1] = sum_n ladj_bar over the whole batch, or NULL (= 0).  in_bar (may not alias in), b_bar,
 * logs_bar (device T[dim]) may each be NULL. */
int bjx_batchnorm_train_vjp(bjx_ctx* ctx, bjx_dtype dt, const void* logs, const void* mean, const void* var, double eps,
                            const double* moments, const double* ladj_bar_sum, const void* in, const void* out_bar,
                            void* in_bar, void* b_bar, void* logs_bar, int64_t dim, int64_t batch);

/* ------------------------------- F4: table lookup                         */
/* RationalQuadraticSpline with matrix parameters, rational_quadratic_spline.jl:128-367,
 * applied to every column of X[dim,batch].  widths/heights/derivs: device T[dim, n_knots]
 * column-major (row i of the Julia matrices = knots of dimension i).
 * inverse=0: y = rqs_univariate, ladj[n] = Σ_i rqs_logabsdetjac (fused like rqs_forward :317-357)
 * inverse=1: x = rqs_univariate_inverse, ladj = -(forward ladj at x). */
int bjx_rqs(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* widths, const void* heights,
            const void* derivs, int n_knots, const void* in, void* out,
            void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* input pullback of with_logabsdet_jacobian for the elementwise spline (inverse=0) and its
 * inverse (inverse=1): in_bar = out_bar * f'(x) + ladj_bar[n] * (log f')'(x) (closed-form derivatives of
 * rational_quadratic_spline.jl:128-357); knot parameters as for bjx_rqs.  Knot gradients are not produced. */
int bjx_rqs_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* widths, const void* heights,
                const void* derivs, int n_knots, const void* in, const void* out_bar, const void* ladj_bar,
                void* in_bar, int64_t dim, int64_t batch);
/* the PARAMETER side of the same pullback.  widths_bar/heights_bar/derivs_bar: device T[dim, n_knots]
 * (overwritten) = cotangents of the knot arrays summed over the batch, for the forward map (inverse=0: `in` = x,
 * `out_bar` = ȳ) or the inverse map (inverse=1: `in` = y, `out_bar` = x̄; implicit function theorem at x = f⁻¹(y)).
 * The derivative at the last knot is not read by the spline (constant 1, as in bjx_rqs): its cotangent is 0.
 * in_bar: NULL, or device T[dim, batch] that receives the input cotangent of bjx_rqs_vjp in the same pass (training wants
 * both: one read of in / out_bar / ladj_bar instead of two).
 * dim <= 256; Float64 accumulation across blocks.  Reference counterpart: the AD package's pullback of
 * rqs_forward / rqs_univariate_inverse (the reference has no hand-written rule). */
int bjx_rqs_vjp_knots(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* widths, const void* heights,
                      const void* derivs, int n_knots, const void* in, const void* out_bar, const void* ladj_bar,
                      void* in_bar, void* widths_bar, void* heights_bar, void* derivs_bar, int64_t dim, int64_t batch);

/* The `B` constructor, rational_quadratic_spline.jl:109-123: raw_w, raw_h: T[dim,K];
 * raw_d: T[dim,K-1]  ->  widths, heights, derivs: T[dim,K+1]. */
int bjx_rqs_params(bjx_ctx* ctx, bjx_dtype dt, const void* raw_w, const void* raw_h,
                   const void* raw_d, int K, int64_t dim, double B,
                   void* widths, void* heights, void* derivs);
/* Pullback of bjx_rqs_params (the `B` constructor, rational_quadratic_spline.jl:109-123): knot cotangents
 * T[dim, K+1] -> raw_w_bar, raw_h_bar T[dim, K], raw_d_bar T[dim, K-1] (softmax/cumsum and log1pexp backwards). */
int bjx_rqs_params_vjp(bjx_ctx* ctx, bjx_dtype dt, const void* raw_w, const void* raw_h, const void* raw_d, int K,
                       int64_t dim, double B, const void* widths_bar, const void* heights_bar, const void* derivs_bar,
                       void* raw_w_bar, void* raw_h_bar, void* raw_d_bar);

/* ------------------------------- F5: gather / scatter wrappers            */
/* Permute, permute.jl:152-157: out[i,n] = in[src[i],n]  (src = column index of the nonzero in
 * row i of the permutation matrix A; 0-based).  Bit-exact data movement; log-det is 0. */
int bjx_permute(bjx_ctx* ctx, bjx_dtype dt, const int32_t* src, const void* in, void* out,
                int64_t dim, int64_t batch);

/* Coupling with a PartitionMask, coupling.jl:125-134,206-259, for coupling laws whose
 * parameters the Julia side has already evaluated (θ(x₂) is an arbitrary closure and cannot
 * cross a C ABI).  idx1: int32[n1] rows that are transformed (0-based); all other rows copy
 * through.  Affine law  y₁ = shift + scale·x₁  (θ ↦ Shift ∘ Scale):
 * scale, shift: device T[n1, batch] (NULL = 1 / 0).  inverse=1 undoes it. */
int bjx_coupling_affine(bjx_ctx* ctx, bjx_dtype dt, int inverse, const int32_t* idx1, int64_t n1,
                        const void* scale, const void* shift, const void* in, void* out,
                        void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch,
                        uint32_t flags);
/* pullback of the affine Coupling (coupling.jl:206-259, b = Shift(t) o Scale(s)):
 * in_bar [dim, batch] (rows outside x_1 pass out_bar through) and the cotangents of theta's outputs, scale_bar and
 * shift_bar [n1, batch] (either may be NULL) — the host continues through the closure theta and adds its x_2
 * cotangent.  inverse=1: in = y, the pre-image x_1 = (y_1 - t)/s is recomputed. */
int bjx_coupling_affine_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const int32_t* idx1, int64_t n1,
                            const void* scale, const void* shift, const void* in, const void* out_bar,
                            const void* ladj_bar, void* in_bar, void* scale_bar, void* shift_bar,
                            int64_t dim, int64_t batch);

/* Spline law (θ ↦ RationalQuadraticSpline(w,h,d)): knots T[n1, n_knots] shared over the batch. */
int bjx_coupling_rqs(bjx_ctx* ctx, bjx_dtype dt, int inverse, const int32_t* idx1, int64_t n1,
                     const void* widths, const void* heights, const void* derivs, int n_knots,
                     const void* in, void* out, void* ladj_ps, double* ladj_sum,
                     int64_t dim, int64_t batch, uint32_t flags);

/* ------------------------------- Stacked with elementwise segments   */
/* Stacked(bs, ranges), src/bijectors/stacked.jl:27-252: y = vcat(bs[i](x[ranges_in[i]])...) and
 * logabsdetjac = Σ_i sum(logabsdetjac(bs[i], x[ranges_in[i]])) (:172-196), applied to every column
 * of X[dim, batch] in ONE launch.  Segment i reads rows [in_lo, in_lo+len) and writes rows
 * [out_lo, out_lo+len) (0-based; `ranges_out` are the cumulative output ranges of :50-57); its
 * bijector is a chain of <= BJX_MAX_SEG_OPS elementwise ops (bjx_op, parameters scalar or one
 * device value per row of the segment).  The segments must use every row exactly once
 * ("input length mismatch", :157).  `segs` is HOST memory; the call copies it.
 * Segments whose bijector is not elementwise (Simplex, Ordered, ...) are the host wrapper's job
 * (slice -> structured entry point -> write back). */
#define BJX_MAX_SEG_OPS 4
typedef struct {
  int64_t in_lo, out_lo, len;
  int32_t n_ops;      /* 0 = identity */
  int32_t reserved;
  bjx_op ops[BJX_MAX_SEG_OPS];
} bjx_segment;
int bjx_stacked(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x, void* y,
                void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);

/* Stacked with STRUCTURED segments (Simplex / Ordered blocks of a Turing model) without slicing copies, stacked.jl:142-166:
 *   1. bjx_stacked_ld — the elementwise segments in ONE launch between matrices of different heights: x is [ldx, batch],
 *      y is [ldy, batch], the segments (row ranges of x -> row ranges of y, every one of the first `dim` rows of y exactly
 *      once) are gathered row by row; rows that belong to a structured segment are covered by identity placeholders.
 *   2. bjx_simplex_ld / bjx_ordered_ld — the structured bijector on a row window of the same matrices: `in` / `out` point at
 *      the first row of the window in column 0, ld_in / ld_out are the heights of the matrices; with BJX_ACCUMULATE the
 *      log-dets add to the ones of step 1.  (VecCholesky / Corr / PD have a matrix on one side and cannot be Stacked segments.) */
int bjx_stacked_ld(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x, int64_t ldx, void* y, int64_t ldy,
                   void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);
int bjx_ordered_ld(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, int64_t ld_in, void* out, int64_t ld_out,
                   void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);
int bjx_simplex_ld(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, int64_t ld_in, void* out, int64_t ld_out,
                   void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags);

/*   3. bjx_stacked_mixed — steps 1 and 2 as ONE launch when a column fits an LDS tile (64 columns x rows x sizeof(T) <= 64 KiB):
 *      a lane owns a column and walks it top to bottom, elementwise rows through the same slots as bjx_stacked, every
 *      structured block with the walker of bjx_simplex / bjx_ordered.  `segs`: the elementwise segments only, in ascending
 *      row order; `blocks`: the structured ones, ascending and disjoint; together they cover every row of y once.  x is
 *      [rows_in, batch], y is [rows_out, batch], both dense.  BJX_ERR_UNSUPPORTED when the column is too tall: fall back to steps 1 + 2. */
typedef enum { BJX_BLOCK_SIMPLEX = 1, BJX_BLOCK_SIMPLEX_INV = 2, BJX_BLOCK_ORDERED = 3, BJX_BLOCK_ORDERED_INV = 4 } bjx_block_kind;
typedef struct {
  int32_t kind;      /* bjx_block_kind */
  int32_t reserved;
  int64_t in_lo, out_lo, len_in, len_out;   /* Simplex: len_out = len_in - 1; its inverse: len_in + 1; Ordered: equal */
} bjx_block;
int bjx_stacked_mixed(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const bjx_block* blocks, int n_blocks,
                      const void* x, int64_t rows_in, void* y, int64_t rows_out, void* ladj_ps, double* ladj_sum,
                      int64_t batch, uint32_t flags);

/* ------------------------------- reverse-mode pullbacks (first rows)  */
/* Pullback of with_logabsdet_jacobian for one launch over the batch:
 *     in_bar = J(in)^T * out_bar + ladj_bar * grad_in logabsdetjac
 * `in` is the PRIMAL INPUT of the direction being differentiated, `out_bar` the cotangent of its
 * output (same shape), `ladj_bar` the cotangent of the per-sample log-det (T[batch], may be NULL = 0),
 * `in_bar` the result (may not alias `in`; may alias `out_bar`).
 * OrderedBijector: ext/BijectorsChainRulesCoreExt.jl:65-197 (rrules of _transform_ordered and
 * _transform_inverse_ordered, matrix methods). */
/* Stacked / any chain of elementwise bijectors (same segment list as bjx_stacked): x_bar[src row] =
 * (dy/dx) y_bar[out row] + ladj_bar (d logabsdetjac / dx), element by element (the Jacobian is diagonal up
 * to the row mapping).  Gradients with respect to the bijectors' PARAMETERS are not produced. */
int bjx_stacked_vjp(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x,
                    const void* y_bar, const void* ladj_bar, void* x_bar, int64_t dim, int64_t batch);

/* bjx_stacked_vjp plus the row moments of its result in the same pass: moments[i] = sum_n x_bar[i,n],
 * moments[dim+i] = sum_n x_bar[i,n]*x[i,n], moments[2 dim] = batch (the layout of bjx_row_moments, device double[2*dim+1]):
 * the parameter cotangents of a leading per-row Shift(mu) / Scale(sigma) stage (mean-field ADVI) without reading x_bar and
 * x a second time.  Shapes the fused kernel does not take (gathered rows, heights that are not whole aligned packs, > 64 packs per
 * column) run the plain pullback (row slabs on tall columns) and then bjx_row_moments, which takes any number of rows. */
int bjx_stacked_vjp_moments(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x,
                            const void* y_bar, const void* ladj_bar, void* x_bar, double* moments,
                            int64_t dim, int64_t batch);
int bjx_ordered_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar,
                    const void* ladj_bar, void* in_bar, int64_t dim, int64_t batch);
/* SimplexBijector (K = rows of the simplex side): reverse sweeps of simplex.jl:47-64 / :102-120 and of the
 * log-det terms :122-138 (the reference's adjoints: simplex.jl:145-215, :248-308, :358-470).
 * inverse=1: in = y[K-1,batch], out_bar = x_bar[K,batch], in_bar = y_bar[K-1,batch];
 * inverse=0: in = x[K,batch],   out_bar = y_bar[K-1,batch], in_bar = x_bar[K,batch]. */
int bjx_simplex_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar,
                    const void* ladj_bar, void* in_bar, int64_t K, int64_t batch);
/* VecCholeskyBijector forward link W[K,K,batch] -> y[n,batch]: the rule the reference ships for
 * _link_chol_lkj_from_upper / _from_lower (ext/BijectorsChainRulesCoreExt.jl:199-311).  It lives on the constraint
 * manifold of Cholesky factors of correlation matrices (unit-norm columns): W_bar[j,j] = 0 and the entries outside
 * the strict triangle — undefined in the reference — are written as zeros.  No log-det cotangent: the reference's
 * rule covers the link only. */
int bjx_vec_cholesky_fwd_vjp(bjx_ctx* ctx, bjx_dtype dt, int uplo, const void* W, const void* y_bar,
                             void* W_bar, int64_t K, int64_t batch);
/* inverse(VecCholeskyBijector): y[n,batch] -> (W[K,K,batch], logJ[batch]); pullback
 * src/bijectors/corr.jl:402-451 (_inv_link_chol_lkj_rrule; ext/BijectorsChainRulesCoreExt.jl:311-320).
 * W_bar: dense K x K per sample (entries outside the stored triangle are ignored), logJ_bar: T[batch]
 * or NULL, y_bar: T[n,batch]. */
int bjx_vec_cholesky_inv_vjp(bjx_ctx* ctx, bjx_dtype dt, int uplo, const void* y, const void* W_bar,
                             const void* logJ_bar, void* y_bar, int64_t K, int64_t batch);

/* ------------------------------- multi-GPU                */
/* RCCL communicator owned by the context (one process per GPU).  `unique_id` is the 128-byte
 * ncclUniqueId made by bjx_comm_unique_id on rank 0 and broadcast by the host (Julia: MPI.jl
 * or a file; Python: the torch.distributed store). */
int bjx_comm_unique_id(void* out128);
int bjx_comm_init(bjx_ctx* ctx, int nranks, int rank, const void* unique_id128);
int bjx_comm_destroy(bjx_ctx* ctx);
/* The single collective of the path: in-place sum all-reduce of the partial Σ logabsdetjac. */
int bjx_allreduce_sum_f64(bjx_ctx* ctx, double* ptr, int64_t n);

/* ------------------------------- measurement helpers                      */
/* Counter-based (Philox4x32-10) standard-normal fill, keyed by (seed, global element index)
 * so synthetic batches are identical for any shard count.
 * `col0` is the global index of this shard's first column. */
int bjx_fill_normal(bjx_ctx* ctx, bjx_dtype dt, void* out, int64_t dim, int64_t batch,
                    int64_t col0, uint64_t seed, double mean, double std);
/* Launch `bjx_chain` `iters` times bracketed by hipEvents on the context stream and return
 * the average milliseconds per launch (used by bench.py for roofline.achieved). */
int bjx_time_begin(bjx_ctx* ctx);
int bjx_time_end(bjx_ctx* ctx, float* ms_out);
/* Between _begin and _end every entry point brackets its DOMINANT kernel launch (not the
 * parameter-prep / finalize helpers) with its own hipEvent pair on the context stream; _end
 * synchronises and returns the summed kernel milliseconds and the number of launches recorded
 * (at most 1024 per region).  bench.py's roofline.achieved is computed from this. */
int bjx_kernel_time_begin(bjx_ctx* ctx);
int bjx_kernel_time_end(bjx_ctx* ctx, float* total_ms, int* launches);
/* Kernel launches issued by the library in this process so far, all contexts, helpers included (table builders, finalize passes,
 * packers; memsets and copies are not kernels).  A difference of two readings around a region = the launches of that region: what
 * bench.py reports as `launches_per_step_all` next to the dominant-kernel count of bjx_kernel_time_end. */
uint64_t bjx_launch_count(void);

/* ---- captured steps (hipGraph) -------------------------------------------------------------------------------------
 * For small shards a step (kernel + finalize + the 8-byte all-reduce) is launch-bound: capture it once, replay it
 * with one hipGraphLaunch.  Between _begin and _end every call on this context is RECORDED on the context stream
 * (hipStreamBeginCapture, relaxed mode) instead of executed; pointers and sizes are baked into the graph, so a replay
 * reads and writes the same buffers.  Entry points that must synchronise with the host (bjx_synchronize,
 * bjx_time_end, bjx_kernel_time_end, bjx_stacked with more segments than the staging buffer was last used for)
 * return BJX_ERR_UNSUPPORTED while a capture is open; per-kernel timing is off inside a capture.
 * bjx_allreduce_sum_f64 is capturable (RCCL records into the graph).  Reference counterpart: none — the reference
 * is a CPU library; this replaces the per-call dispatch a Julia host pays per step. */
typedef struct bjx_graph bjx_graph;
int bjx_graph_begin(bjx_ctx* ctx);
int bjx_graph_end(bjx_ctx* ctx, bjx_graph** out);
int bjx_graph_launch(bjx_ctx* ctx, bjx_graph* graph);   /* asynchronous on the context stream */
int bjx_graph_destroy(bjx_graph* graph);

/* ---------------------------------------------------------------- plans
 * What a sampler calls thousands of times per second is the SAME bijector on a small (param_dim x n_chains) array
 * (src/vector/product/fill.jl:146-165, 192-213: one `from_linked_vec` / `to_linked_vec` per log-density evaluation): the kernel takes
 * 5-15 us, and a host that walks the bijector, marshals the op list and checks the shapes on every call spends 3-8x that
 * (profiles/r05_host_overhead.txt).  A plan holds everything of the call that does not change — validated once — so that a call is
 * the data pointers and the batch.  A plan belongs to the context it was made for (same stream, same thread rules) and holds parameter
 * POINTERS, never values: the caller keeps the parameter arrays alive and may rewrite them in place between runs.
 *   bjx_plan_chain        an elementwise chain (bjx_chain's ops / dim / flags)
 *   bjx_plan_structured   BJX_PLAN_SIMPLEX / BJX_PLAN_ORDERED (bjx_simplex / bjx_ordered; `dim` = rows of the INPUT)
 *   bjx_plan_stacked_vjp  the input pullback of a chain / Stacked of elementwise bijectors (bjx_stacked_vjp), run by bjx_plan_run_vjp
 *   bjx_plan_run          in / out / ladj_ps / ladj_sum as in the planned entry.  `ladj_sum_t` (Float32 plans only, may be NULL): the
 *                         sum once more as T[1] = Float32, written by the same epilogue that finishes ladj_sum — the reference returns
 *                         the scalar in the element type, and a host-side conversion is another launch.  With ladj_sum_t given,
 *                         ladj_sum may be NULL (the Float64 accumulator then lives in the context). */
typedef struct bjx_plan bjx_plan;
enum { BJX_PLAN_CHAIN = 1, BJX_PLAN_SIMPLEX = 2, BJX_PLAN_ORDERED = 3, BJX_PLAN_STACKED_VJP = 4, BJX_PLAN_STACKED = 5 };
int bjx_plan_chain(bjx_ctx* ctx, bjx_dtype dt, const bjx_op* ops, int n_ops, int64_t dim, uint32_t flags, bjx_plan** out);
int bjx_plan_structured(bjx_ctx* ctx, bjx_dtype dt, int kind, int inverse, int64_t dim, uint32_t flags, bjx_plan** out);
int bjx_plan_run(bjx_plan* plan, const void* in, void* out, void* ladj_ps, double* ladj_sum, void* ladj_sum_t, int64_t batch);
/* The pullback a gradient-based sampler repeats on every leapfrog step: bjx_stacked_vjp's segment list (an elementwise chain = one segment over all
 * rows) validated once; bjx_plan_run_vjp(plan, x, y_bar, ladj_bar, x_bar, batch) has bjx_stacked_vjp's meaning. */
/* A `Stacked` of elementwise chains (bjx_stacked: the linked vector of a heterogeneous product distribution, src/vector/interface.jl:86-129), run by bjx_plan_run. */
int bjx_plan_stacked(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, int64_t dim, uint32_t flags, bjx_plan** out);
int bjx_plan_stacked_vjp(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, int64_t dim, bjx_plan** out);
int bjx_plan_run_vjp(bjx_plan* plan, const void* x, const void* y_bar, const void* ladj_bar, void* x_bar, int64_t batch);
int bjx_plan_destroy(bjx_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* BJX_H */
