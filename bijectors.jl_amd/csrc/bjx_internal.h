// bjx_internal.h — shared host/device helpers of libbjx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>

#include "../../include/bjx.h"

// Every kernel launch of the library is counted (one relaxed atomic add on the host: bjx_launch_count, include/bjx.h) so that a
// measurement can say how many launches a step REALLY is — the hot kernel, its helpers (table builders, finalize passes, packers)
// and all (VERDICT r05 weak #7d: `kernel_launches_per_step` counted the dominant kernel only).  Memsets / copies are not launches.
#include <atomic>
extern std::atomic<unsigned long long> bjx_g_launches;      // bjx_ctx.hip
#ifdef hipLaunchKernelGGL
#undef hipLaunchKernelGGL
#endif
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                         \
  do {                                                                                                            \
    bjx_g_launches.fetch_add(1ull, std::memory_order_relaxed);                                                    \
    kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                            \
  } while (0)

#define BJX_API extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ context
constexpr int BJX_INKERNEL_FIN_MAX = 4096;    // partials one block reduces: in-kernel (BJX_OPT_INKERNEL_FINALIZE = 1) and in bjx_finalize_kernel
constexpr int BJX_MAX_BLOCKS = 4096;        // persistent-grid cap AND size of the 2nd-stage partial buffer
constexpr int BJX_FIN_SENT_GROUPS = 1040;   // sentinel hand-off: 8 residue classes x ceil(8192 / 64) groups of 64 blocks (+ slack)
constexpr int BJX_FIN_WIDE_MAX = 65536;     // partials ONE 1024-thread block still sums in a single launch (bjx_finalize_wide_kernel)
constexpr int BJX_CONSTS = 8;               // device doubles for parameter-only log-det terms
constexpr size_t BJX_SCRATCH_BYTES = 1 << 20;  // û tables, small parameter staging
constexpr size_t BJX_HOST_STAGE_BYTES = 256 << 10;  // pinned host staging for descriptor lists (bjx_stacked)

struct bjx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  double* partials = nullptr;   // [partials_cap] one f64 per publishing block (grown on demand, cached)
  size_t partials_cap = 0;
  double* partials2 = nullptr;  // [BJX_MAX_BLOCKS] second reduction stage
  void* host_stage = nullptr;      // pinned, BJX_HOST_STAGE_BYTES, created on first use
  hipEvent_t stage_ev = nullptr;   // recorded after the last copy out of host_stage
  unsigned* fin_counter = nullptr;  // arrival counter of the in-kernel finalize (zero between launches)
  // Failure channel of the sentinel hand-off (round 6): word [4] of the fin_counter allocation is a sticky DEVICE error word (a poll
  // that timed out sets it; the last block of every later launch reads it and writes NaN instead of a sum built on a slot the
  // faulted launch may have left dirty); `fin_err_host` is the same fact in pinned, device-mapped host memory, which the host reads
  // for free at the next entry and in bjx_synchronize -> BJX_ERR_FINALIZE, re-arm, two-pass from then on.
  unsigned* fin_err = nullptr;            // = fin_counter + 4
  volatile unsigned* fin_err_host = nullptr;   // hipHostMalloc(mapped), 64 bytes
  unsigned* fin_err_host_dev = nullptr;   // the device's address of the same word
  int fin_faults = 0;                     // hand-off faults seen by this context (after the first the context stays on the two-pass finalize)
  int dbg_fin_drop = -1;                  // BJX_OPT_DEBUG_FIN_DROP_BLOCK: fault injection — this block index does not publish (tests)
  hipEvent_t stream_ev = nullptr;         // bjx_set_stream: the new stream waits for the work in flight on the old one
  // bjx_plan_run(..., ladj_sum_t): where the NEXT sum-producing launch of this context also writes its sum as Float32 (taken by
  // bjx_make_fin / bjx_launch_finalize; whatever path did not take it is served by a one-thread cast launch in bjx_plan_run)
  float* fin_out32 = nullptr;
  int fin_out32_taken = 0;
  double* sent_l1 = nullptr;    // [BJX_FIN_SENT_GROUPS * 64] block partials of the sentinel hand-off (BJX_FIN_SENT between launches)
  double* sent_l2 = nullptr;    // [BJX_FIN_SENT_GROUPS] group sums of the sentinel hand-off (BJX_FIN_SENT between launches)
  double* consts = nullptr;     // [BJX_CONSTS]
  void* scratch = nullptr;      // [BJX_SCRATCH_BYTES]
  void* big_ws = nullptr;       // workspace of the general-size matrix kernels (K > 64, matrix Scale beyond 128 rows): grown on demand, cached
  size_t big_ws_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cu = 256;
  // BJX_OPT_PARAM_EPOCH (0 = off): while the host keeps the epoch unchanged, tables DERIVED from parameter arrays (the spline's LDS blob,
  // the factorisation behind a matrix `Scale`) are reused when the same device pointers come back, instead of being rebuilt by a helper
  // launch on every call.
  int param_epoch = 0;
  // BJX_OPT_PARAM_EPOCH: the spline's LDS blob (built by a one-block helper launch) is kept per slot while the epoch stands
  struct RqsBlobSlot {
    const void *w = nullptr, *h = nullptr, *d = nullptr;
    int K1 = 0, V = 0, nstep_hi = 0, dual = 0, G = 0, inverse = 0, dt = 0, epoch = 0;
    int64_t rows = 0, trows = 0;
    void* buf = nullptr;          // [64 bytes flag][blob], kRqsBlobMax + 64, allocated on first use
  } rqs_slots[4];
  int rqs_next = 0;
  // the factorisation behind a matrix `Scale` ([A^-1 | logabsdet], bjx_scale_matrix) under the epoch contract
  struct ScaleSlot {
    const void* a = nullptr;
    int64_t dim = 0;
    int dt = 0, has_inverse = 0, epoch = 0;
    void* buf = nullptr;          // [dim][2 dim] of T, then one double (logabsdet); allocated on first use
    size_t cap = 0;
  } scale_slot;
  int opt_inkernel_fin = 2;     // BJX_OPT_INKERNEL_FINALIZE: 0 two follow-up launches, 1 arrival ticket (<= 4096 blocks), 2 (default) sentinel hand-off (<= 65536 blocks)
  uint64_t rng_seed = 0;        // bjx_set_rng: stream of the fused sampling path (BJX_INPUT_STDNORMAL)
  int64_t rng_col0 = 0;
  // per-launch timing of the DOMINANT kernel of each call (bjx_kernel_time_begin/_end): event pairs
  // recorded right around the hot kernel, so helper launches and host gaps are excluded
  static constexpr int PROF_MAX = 1024;
  hipEvent_t* prof_ev = nullptr;   // [2 * PROF_MAX], created lazily
  int prof_on = 0, prof_n = 0, prof_dropped = 0;
  int capturing = 0;            // between bjx_graph_begin and bjx_graph_end
  // RCCL (lazily dlopen'ed)
  void* rccl_handle = nullptr;
  void* comm = nullptr;
  int nranks = 1, rank = 0;
  int collective_timeout_ms = 0;   // BJX_OPT_COLLECTIVE_TIMEOUT_MS: watchdog of bjx_synchronize while a communicator is attached (0 = none)
  char err[512] = {0};
};

// A launch plan (include/bjx.h "plans"): everything of a call that does not change from call to call, validated once.
struct bjx_plan {
  bjx_ctx* ctx = nullptr;
  int kind = 0;                 // BJX_PLAN_*
  bjx_dtype dt = BJX_F32;
  int n_ops = 0;
  bjx_op ops[BJX_MAX_OPS];
  int inverse = 0;
  int64_t dim = 0;              // rows of the INPUT
  uint32_t flags = 0;
  int n_segs = 0;               // BJX_PLAN_STACKED_VJP: the segment list of bjx_stacked_vjp
  bjx_segment* segs = nullptr;
  ~bjx_plan() { delete[] segs; }
};

inline int bjx_fail(bjx_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define BJX_HIP(ctx, expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return bjx_fail((ctx), (int)e_, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

#define BJX_CHECK_LAUNCH(ctx) BJX_HIP(ctx, hipGetLastError())

// RAII event pair around the hot kernel launch of an entry point (no-op unless profiling is on)
struct BjxProf {
  bjx_ctx* c;
  int slot;
  explicit BjxProf(bjx_ctx* ctx) : c(ctx), slot(-1) {
    if (c->prof_on && !c->capturing) {
      if (c->prof_n < bjx_ctx::PROF_MAX) { slot = c->prof_n++; (void)hipEventRecord(c->prof_ev[2 * slot], c->stream); }
      else ++c->prof_dropped;
    }
  }
  ~BjxProf() { if (slot >= 0) (void)hipEventRecord(c->prof_ev[2 * slot + 1], c->stream); }
  BjxProf(const BjxProf&) = delete;
  BjxProf& operator=(const BjxProf&) = delete;
};

#define BJX_REQUIRE(ctx, cond, code, ...)                    \
  do {                                                       \
    if (!(cond)) return bjx_fail((ctx), (code), __VA_ARGS__); \
  } while (0)

// Σ log|det J| epilogue descriptor, passed BY VALUE to the hot kernels.  With `counter` set the
// last block to arrive reduces all partials in the fixed order of bjx_finalize_kernel inside the same
// launch (no extra dispatch, ~5 us each on MI355X); otherwise the host launches the two-pass finalize.
struct BjxFin {
  double* partials = nullptr;       // null: no sum requested
  unsigned* counter = nullptr;      // null: host-launched finalize (or the sentinel hand-off, `l2` set)
  double* l2 = nullptr;             // sentinel hand-off: the group sums; `partials` then points at the sentinel-initialised block slots
  double* out = nullptr;
  double host_const = 0.0;
  const double* dev_const = nullptr;
  int accumulate = 0;
  int drop_block = -1;              // fault injection (BJX_OPT_DEBUG_FIN_DROP_BLOCK): the block that does not publish
  unsigned* err = nullptr;          // sentinel hand-off: sticky device error word (ctx->fin_err)
  unsigned* err_host = nullptr;     // ... and its host-visible twin (ctx->fin_err_host_dev)
  float* out32 = nullptr;           // bjx_plan_run: the finished sum once more, as Float32 (the element type of the call), for hosts that return a T scalar
};
// host side: builds the descriptor for a launch of `grid` blocks; *second_pass = launch bjx_launch_finalize afterwards
int bjx_make_fin(bjx_ctx* ctx, int64_t grid, double* ladj_sum, double host_const, int use_dev_const, uint32_t flags,
                 BjxFin* fin, bool* second_pass);
// host side: BJX_ERR_FINALIZE (after re-arming the slots and switching the context to the two-pass finalize) when a hand-off of an earlier launch timed out
int bjx_fin_fault_check(bjx_ctx* ctx);

// host side: turn a descriptor built by bjx_make_fin back into the two-pass form (per-block partials + bjx_launch_finalize) — for kernels
// whose long-lived blocks at low occupancy lose more to a closing block's wait than the follow-up launch costs
int bjx_fin_two_pass(bjx_ctx* ctx, int64_t grid, BjxFin* fin, bool* second_pass);

// host side: make sure ctx->big_ws holds `bytes` (cached; growing it synchronises the device and cannot be captured)
int bjx_ensure_big_ws(bjx_ctx* ctx, size_t bytes);
// host side: make sure ctx->partials can hold n doubles (cached; reallocation synchronises the device)
int bjx_ensure_partials(bjx_ctx* ctx, size_t n);
// host side: launch the fixed-order reduction of per-block partials (+ constant term)
int bjx_launch_finalize(bjx_ctx* ctx, int n_partials, double* ladj_sum, double host_const,
                        int use_dev_const, double dev_const_mult, uint32_t flags);

// bjx_tall.hip: Ordered / Simplex on columns taller than the quad frames of bjx_seq.hip (G lanes per column, any height up to
// 64 lanes x 128 bytes); *taken = false and nothing is launched when the shape is not for that kernel
enum { BJX_TALL_ORDERED_FWD = 0, BJX_TALL_ORDERED_INV = 1, BJX_TALL_SIMPLEX_FWD = 2, BJX_TALL_SIMPLEX_INV = 3 };
int bjx_tall_stream(bjx_ctx* ctx, bjx_dtype dt, int which, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t rows_in,
                    int64_t rows_out, int64_t batch, uint32_t flags, bool* taken);
// bjx_tiny.hip: the same maps on columns of 1 ... 7 rows (lane = column in registers); same contract
int bjx_seq_tiny(bjx_ctx* ctx, bjx_dtype dt, int which, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags,
                 bool* taken);
int bjx_seq_tiny_vjp(bjx_ctx* ctx, bjx_dtype dt, int simplex, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K,
                     int64_t batch, bool* taken);
int bjx_tall_simplex_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K,
                         int64_t batch, bool* taken);

// Planar on the column-tile mapping (bjx_flow_cols.hip), called from planar_impl / planar_vjp_impl (bjx_flow.hip) with the prepared
// û / wᵀû tables; 1 = the shape is not theirs.
namespace bjx {
template <class T>
int planar_cols_launch(bjx_ctx* ctx, int inverse, const T* w, const T* u_hat, const T* wtu, const T* b, int nl, const T* in, T* out, T* ladj_ps,
                       double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags);
template <class T>
int planar_vjp_cols_launch(bjx_ctx* ctx, int inverse, const T* w, const T* u_hat, const T* wtu, const T* b, int nl, const T* in, const T* out_bar,
                           const T* ladj_bar, T* in_bar, int64_t dim, int64_t batch, T* t_out, T* s_out);
// the Float32 register-tile input pullback (bjx_flow_vjp_reg.hip); 1 = shape not served
int planar_vjp_reg_launch(bjx_ctx* ctx, int inverse, const float* w, const float* u_hat, const float* wtu, const float* b, int nl, const float* in,
                          const float* out_bar, const float* ladj_bar, float* in_bar, int64_t dim, int64_t batch, float* t_out, float* s_out);
}  // namespace bjx

// ------------------------------------------------------------------ device math
// Same definitions as the reference's third-party scalar functions (LogExpFunctions), see
// oracle/bjx_oracle.cpp for the citations; thresholds are identical to the CPU restatement.
namespace bjx {

template <class T> struct Num;
template <> struct Num<float> {
  static constexpr float eps = 1.1920928955078125e-07f;
  static constexpr float logistic_lo = -103.27893f, logistic_hi = 16.635532f;
  static constexpr float l1pe0 = -16.635532f, l1pe1 = 7.9711924f, l1pe2 = 13.993f;
  static constexpr float log2 = 0.69314718055994530942f;
  static constexpr float inf = __builtin_huge_valf();
};
template <> struct Num<double> {
  static constexpr double eps = 2.220446049250313e-16;
  static constexpr double logistic_lo = -744.4400719213812, logistic_hi = 36.7368005696771;
  static constexpr double l1pe0 = -36.7368005696771, l1pe1 = 18.021826694558577, l1pe2 = 33.23111882352963;
  static constexpr double log2 = 0.69314718055994530942;
  static constexpr double inf = __builtin_huge_val();
};

__device__ __forceinline__ float d_exp(float x) { return expf(x); }
__device__ __forceinline__ double d_exp(double x) { return exp(x); }
__device__ __forceinline__ float d_log(float x) { return logf(x); }
__device__ __forceinline__ double d_log(double x) { return log(x); }
__device__ __forceinline__ float d_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ double d_log1p(double x) { return log1p(x); }
__device__ __forceinline__ float d_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double d_tanh(double x) { return tanh(x); }
__device__ __forceinline__ float d_cosh(float x) { return coshf(x); }
__device__ __forceinline__ double d_cosh(double x) { return cosh(x); }
__device__ __forceinline__ float d_atanh(float x) { return atanhf(x); }
__device__ __forceinline__ double d_atanh(double x) { return atanh(x); }
__device__ __forceinline__ float d_asinh(float x) { return asinhf(x); }
__device__ __forceinline__ double d_asinh(double x) { return asinh(x); }
__device__ __forceinline__ float d_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double d_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float d_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double d_abs(double x) { return fabs(x); }
__device__ __forceinline__ float d_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double d_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float d_copysign(float m, float s) { return __builtin_copysignf(m, s); }
__device__ __forceinline__ double d_copysign(double m, double s) { return __builtin_copysign(m, s); }
__device__ __forceinline__ bool d_isfinite(float x) { return fabsf(x) < Num<float>::inf; }
__device__ __forceinline__ bool d_isfinite(double x) { return fabs(x) < Num<double>::inf; }

// src/Bijectors.jl:95-100
template <class T> __device__ __forceinline__ T d_clamp(T x, T a, T b) { return x < a ? a : (x > b ? b : x); }
// clamp in one instruction (v_med3_f32); bounds may be +-inf
__device__ __forceinline__ float d_med3(float x, float a, float b) { return __builtin_amdgcn_fmed3f(x, a, b); }
__device__ __forceinline__ double d_med3(double x, double a, double b) { return fmin(fmax(x, a), b); }
template <class T> __device__ __forceinline__ T d_logit(T x) { return d_log(x / (T(1) - x)); }
template <class T> __device__ __forceinline__ T d_logistic(T x) {
  T e = d_exp(x);
  return x < Num<T>::logistic_lo ? T(0) : (x > Num<T>::logistic_hi ? T(1) : e / (T(1) + e));
}
template <class T> __device__ __forceinline__ T d_log1pexp(T x) {
  if (x < Num<T>::l1pe0) return d_exp(x);
  if (x < Num<T>::l1pe1) return d_log1p(d_exp(x));
  if (x < Num<T>::l1pe2) return x + d_exp(-x);
  return x;
}
template <class T> __device__ __forceinline__ T d_logcosh(T x) {
  T ax = d_abs(x);
  return ax + d_log1pexp(T(-2) * ax) - Num<T>::log2;
}

// ------------------------------------------------------------------ fast math for ALU-bound kernels
// Float32: hardware transcendental units (v_log_f32 / v_exp_f32 / v_rcp_f32, ~1 ulp each, relative
// error of the composite <= ~1e-6) — used only in kernels that are VALU-bound with the OCML
// routines (Simplex, RQS, VecCholesky, Planar recurrence; PMC evidence in profiles/).  The Float32
// parity bar is 1e-3.  Float64 always uses the exact OCML functions (parity bar 1e-6).
template <class T> struct Fast;
template <> struct Fast<float> {
  static __device__ __forceinline__ float log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
  static __device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }
  static __device__ __forceinline__ float exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float log1p(float x) {
    // log1p via the compensated log(1+x) * x / ((1+x) - 1) form (exact when 1+x rounds to 1)
    const float u = 1.0f + x;
    const float d = u - 1.0f;
    return d == 0.0f ? x : log(u) * (x * __builtin_amdgcn_rcpf(d));
  }
};
// Float64: lean versions of the elementary functions (BJX_F64_LEAN, default on).  The OCML routines are correctly
// rounded-ish with full denormal / flag care and cost 40-100 VALU each, which makes every Float64 kernel VALU-bound at
// 15-58 % of the HBM roofline; the parity bar is 1e-6 relative.  These keep ~1e-15 relative accuracy (scripts/f64math_bench.hip
// prints the measured maximum error against long double libm and the throughput next to OCML) and the IEEE special values
// the reference's code paths rely on (log(0) = -Inf, log(<0) = NaN, exp(-Inf) = 0, 1/0 = Inf, NaN in -> NaN out); they do
// not care about gradual underflow of results or exception flags.
#ifndef BJX_F64_LEAN
#define BJX_F64_LEAN 1
#endif
namespace f64lean {
__device__ __forceinline__ double rcp(double x) {
  const double r0 = __builtin_amdgcn_rcp(x);                 // v_rcp_f64: ~2^-26 relative
  double e = __builtin_fma(-x, r0, 1.0);
  double r = __builtin_fma(e, r0, r0);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(e, r, r);
  return r == r ? r : r0;                                    // x = 0, +-Inf, NaN: the Newton step is NaN, the hardware value is right
}
__device__ __forceinline__ void sqrt_rsqrt(double x, double& sq, double& rs) {
  const double y = __builtin_amdgcn_rsq(x);                  // v_rsq_f64
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  const bool special = !(x > 0.0) || x == Num<double>::inf;  // 0, negative, NaN, +Inf: g is NaN there
  sq = special ? (x == 0.0 || x == Num<double>::inf ? x : y * 0.0) : g;   // sqrt(-x) = NaN, sqrt(0) = 0, sqrt(Inf) = Inf
  rs = special ? y : h + h;
}
__device__ __forceinline__ double exp(double x) {
  const double xc = __builtin_fmin(__builtin_fmax(x, -1100.0), 1100.0);
  const double k = __builtin_rint(xc * 1.4426950408889634074);
  double r = __builtin_fma(k, -6.93147180369123816490e-01, xc);      // ln2 split: hi has 32 significant bits
  r = __builtin_fma(k, -1.90821492927058770002e-10, r);
  // exp(r), |r| <= ln2/2: Taylor to r^13 (truncation 2e-18 relative)
  double p = 1.6059043836821614599e-10;                     // 1/13!
  p = __builtin_fma(p, r, 2.0876756987868098979e-09);       // 1/12!
  p = __builtin_fma(p, r, 2.5052108385441718775e-08);
  p = __builtin_fma(p, r, 2.7557319223985890653e-07);
  p = __builtin_fma(p, r, 2.7557319223985890653e-06);
  p = __builtin_fma(p, r, 2.4801587301587301587e-05);
  p = __builtin_fma(p, r, 1.9841269841269841270e-04);
  p = __builtin_fma(p, r, 1.3888888888888888889e-03);
  p = __builtin_fma(p, r, 8.3333333333333333333e-03);
  p = __builtin_fma(p, r, 4.1666666666666666667e-02);
  p = __builtin_fma(p, r, 1.6666666666666666667e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const double y = __builtin_ldexp(p, (int)k);
  return x == x ? y : x;                                     // fmin/fmax drop a NaN
}
// expm1 for a <= 0 (the tanh / logistic tails): exact slope at 0, no cancellation
__device__ __forceinline__ double expm1_neg(double a) {
  const double ac = __builtin_fmax(a, -1100.0);
  const double k = __builtin_rint(ac * 1.4426950408889634074);
  double r = __builtin_fma(k, -6.93147180369123816490e-01, ac);
  r = __builtin_fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821614599e-10;
  p = __builtin_fma(p, r, 2.0876756987868098979e-09);
  p = __builtin_fma(p, r, 2.5052108385441718775e-08);
  p = __builtin_fma(p, r, 2.7557319223985890653e-07);
  p = __builtin_fma(p, r, 2.7557319223985890653e-06);
  p = __builtin_fma(p, r, 2.4801587301587301587e-05);
  p = __builtin_fma(p, r, 1.9841269841269841270e-04);
  p = __builtin_fma(p, r, 1.3888888888888888889e-03);
  p = __builtin_fma(p, r, 8.3333333333333333333e-03);
  p = __builtin_fma(p, r, 4.1666666666666666667e-02);
  p = __builtin_fma(p, r, 1.6666666666666666667e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p *= r;                                                    // expm1(r) = r (1 + r/2 + ...)
  const double s = __builtin_ldexp(1.0, (int)k);             // 2^k, k <= 0
  const double y = __builtin_fma(s, p, s - 1.0);             // 2^k expm1(r) + (2^k - 1)
  return a == a ? y : a;
}
__device__ __forceinline__ double log(double x) {
  // x = m 2^e, m in [sqrt(1/2), sqrt(2)); log(m) = 2 atanh(s), s = f/(2+f), f = m - 1 (fdlibm's scheme, Remez coefficients)
  double m = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1), denormals handled by the instruction
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f * rcp(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  double y = __builtin_fma(dk, 6.93147180369123816490e-01, f - (hfsq - __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10)));
  // specials: log(0) = -Inf, log(<0) = NaN, log(+Inf) = +Inf, NaN -> NaN
  y = x == Num<double>::inf ? x : y;
  y = x == 0.0 ? -Num<double>::inf : y;
  y = x < 0.0 ? __builtin_nan("") : y;
  return x == x ? y : x;
}
__device__ __forceinline__ double log1p(double x) {
  const double u = 1.0 + x;
  const double c = x - (u - 1.0);                            // what the addition lost
  const double l = log(u);
  return __builtin_fabs(c) > 0.0 && u > 0.0 && u < Num<double>::inf ? __builtin_fma(c, rcp(u), l) : l;
}
}  // namespace f64lean

template <> struct Fast<double> {
#if BJX_F64_LEAN
  static __device__ __forceinline__ double log(double x) { return f64lean::log(x); }
  static __device__ __forceinline__ double log2(double x) { return f64lean::log(x) * 1.4426950408889634074; }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }      // OCML's exp is already lean (measured: 1353 vs 1206 G/s)
  static __device__ __forceinline__ double rcp(double x) { return f64lean::rcp(x); }
  static __device__ __forceinline__ double div(double a, double b) { return a * f64lean::rcp(b); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }    // same rate as the lean form
  static __device__ __forceinline__ double rsqrt(double x) { double s, r; f64lean::sqrt_rsqrt(x, s, r); return r; }
  static __device__ __forceinline__ double log1p(double x) { return f64lean::log1p(x); }
#else
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ double div(double a, double b) { return a / b; }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
  static __device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
#endif
};
// tanh / asinh / atanh / logcosh in Float64 from the lean pieces (Float32 callers have their own hardware-unit forms)
__device__ __forceinline__ double fast_tanh64(double x) {
  const double em = f64lean::expm1_neg(-2.0 * __builtin_fabs(x));          // e^{-2|x|} - 1 in [-1, 0]
  const double t = -em * f64lean::rcp(2.0 + em);
  return __builtin_copysign(t, x);
}
__device__ __forceinline__ double fast_asinh64(double x) {
  const double ax = __builtin_fabs(x);
  double sq, rs;
  f64lean::sqrt_rsqrt(__builtin_fma(ax, ax, 1.0), sq, rs);
  const double small = f64lean::log1p(ax + ax * ax * f64lean::rcp(1.0 + sq));   // log(|x| + sqrt(x^2+1)) without cancellation
  const double big = f64lean::log(ax) + 0.69314718055994530942;              // |x| > 1e150: x^2 overflows
  return __builtin_copysign(ax > 1e150 ? big : small, x);
}

// tanh / (tanh, sech^2) / asinh / atanh: Float32 -> OCML (the Float32 hot kernels have their own hardware-unit forms),
// Float64 -> the lean pieces (3.7x OCML's tanh, scripts/f64math_bench.hip)
__device__ __forceinline__ float x_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double x_tanh(double x) {
#if BJX_F64_LEAN
  return fast_tanh64(x);
#else
  return ::tanh(x);
#endif
}
__device__ __forceinline__ void x_tanh_sech2(float arg, float& th, float& s2) { th = tanhf(arg); const float sc = 1.0f / coshf(arg); s2 = sc * sc; }
__device__ __forceinline__ void x_tanh_sech2(double arg, double& th, double& s2) {
#if BJX_F64_LEAN
  const double em = f64lean::expm1_neg(-2.0 * __builtin_fabs(arg));     // e - 1, e = exp(-2|arg|)
  const double r = f64lean::rcp(2.0 + em);                               // 1 / (1 + e)
  th = __builtin_copysign(-em * r, arg);
  s2 = 4.0 * (1.0 + em) * r * r;                                        // sech^2 = 4e / (1+e)^2
#else
  th = ::tanh(arg); const double sc = 1.0 / ::cosh(arg); s2 = sc * sc;
#endif
}
__device__ __forceinline__ float x_asinh(float x) { return asinhf(x); }
__device__ __forceinline__ float x_atanh(float x) { return atanhf(x); }
#if BJX_F64_LEAN
__device__ __forceinline__ double x_asinh(double x) { return fast_asinh64(x); }
__device__ __forceinline__ double x_atanh(double x) { return 0.5 * f64lean::log1p((x + x) * f64lean::rcp(1.0 - x)); }
#else
__device__ __forceinline__ double x_asinh(double x) { return ::asinh(x); }
__device__ __forceinline__ double x_atanh(double x) { return ::atanh(x); }
#endif

// LogExpFunctions.logistic / log1pexp on the fast units (same saturation / branch thresholds as above)
template <class T> __device__ __forceinline__ T f_logistic(T x) {
  // the quotient is computed before the selects: a nested ?: with the rcp in its last arm compiles to two
  // branches per element (vjp(inverse(Simplex)): 1.53 -> 1.31 ms without them)
  // The lower saturation needs no select: below logistic_lo exp(x) has underflowed to 0 (or to a denormal, which
  // gives a result below 1.5e-38 / 5e-324 instead of the reference's exact 0), and 0·rcp(1) = 0.
  const T e = Fast<T>::exp(x);
  const T q = e * Fast<T>::rcp(T(1) + e);
  return x > Num<T>::logistic_hi ? T(1) : q;
}
template <class T> __device__ __forceinline__ T f_log1pexp(T x) {
  const T e = Fast<T>::exp(x < Num<T>::l1pe1 ? x : -x);
  if (x < Num<T>::l1pe0) return e;
  if (x < Num<T>::l1pe1) return Fast<T>::log1p(e);
  if (x < Num<T>::l1pe2) return x + e;
  return x;
}

// LogExpFunctions.logcosh on the fast units: |x| + log1pexp(-2|x|) - log 2
template <class T> __device__ __forceinline__ T f_logcosh(T x) {
  const T ax = d_abs(x);
  return ax + f_log1pexp(T(-2) * ax) - Num<T>::log2;
}

// ------------------------------------------------------------------ Philox4x32-10 standard normals
// Counter = GLOBAL element index / 4, key = seed; 4 x u32 -> 2 Box-Muller pairs -> 4 normals (Float64 math, the
// result is rounded to T once).  Element e of the global array (col0*dim + local index) always gets the same
// value, so a batch is identical for any shard count; bjx_fill_normal and the fused sampling path
// of bjx_chain (BJX_INPUT_STDNORMAL) draw from the same stream.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair; same bits.
    // (Half the multiply instructions but only +1-2 % on the fused sampler: the 64-bit mad issues like the pair.)
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void philox_normal4(uint64_t seed, int64_t counter, double (&z)[4]) {
  uint32_t r[4];
  philox4x32_10((uint32_t)counter, (uint32_t)((uint64_t)counter >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const double u1 = ((double)r[2 * p] + 1.0) * (1.0 / 4294967296.0);    // (0, 1]
    const double u2 = (double)r[2 * p + 1] * (1.0 / 4294967296.0);        // [0, 1)
    const double rad = sqrt(-2.0 * log(u1));
    double sn, co;
    sincospi(2.0 * u2, &sn, &co);
    z[2 * p] = rad * co;
    z[2 * p + 1] = rad * sn;
  }
}
// Float32 outputs: the Box-Muller step on the hardware units (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 take the
// angle in revolutions) — the Float64 version above is ~150 VALU per normal and made both the fill and the fused
// sampling kernel VALU-bound at 13 % of the HBM write rate.  Same counters, same uniforms; the stream of a dtype is
// defined by its own routine, identical in bjx_fill_normal and in the fused path.
__device__ __forceinline__ void philox_normal4(uint64_t seed, int64_t counter, float (&z)[4]) {
  uint32_t r[4];
  philox4x32_10((uint32_t)counter, (uint32_t)((uint64_t)counter >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = ((float)r[2 * p] + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)r[2 * p + 1] * 2.3283064365386963e-10f;       // [0, 1]
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    z[2 * p] = rad * __builtin_amdgcn_cosf(u2);
    z[2 * p + 1] = rad * __builtin_amdgcn_sinf(u2);
  }
}

// ------------------------------------------------------------------ reductions (wave = 64)
template <class T> __device__ __forceinline__ T shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }

// sum over aligned groups of G consecutive lanes (G power of two <= 64); every lane gets the sum
template <int G, class T> __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int m = G >> 1; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}
template <class T> __device__ __forceinline__ T group_sum_rt(T v, int G) {
  for (int m = G >> 1; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}
// Float32: the four in-row stages as DPP butterflies (one v_add_f32 with a DPP operand each, no LDS
// crossbar traffic, no per-stage address arithmetic: a ds_bpermute stage costs ~6 VALU + 1 LDS op);
// the two cross-row stages stay shuffles.  G is wave-uniform, so the branches are scalar.
template <int CTRL> __device__ __forceinline__ float dpp_xadd(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int G> __device__ __forceinline__ float group_sum_f32_dpp(float v) {          // compile-time G: the same DPP stages
  if constexpr (G >= 2) v = dpp_xadd<0xB1>(v);
  if constexpr (G >= 4) v = dpp_xadd<0x4E>(v);
  if constexpr (G >= 8) v = dpp_xadd<0x141>(v);
  if constexpr (G >= 16) v = dpp_xadd<0x140>(v);
  if constexpr (G >= 32) v += shfl_xor(v, 16);
  if constexpr (G >= 64) v += shfl_xor(v, 32);
  return v;
}
template <> __device__ __forceinline__ float group_sum_rt<float>(float v, int G) {
  if (G >= 2) v = dpp_xadd<0xB1>(v);     // quad_perm [1,0,3,2]
  if (G >= 4) v = dpp_xadd<0x4E>(v);     // quad_perm [2,3,0,1]
  if (G >= 8) v = dpp_xadd<0x141>(v);    // row_half_mirror (quads already uniform)
  if (G >= 16) v = dpp_xadd<0x140>(v);   // row_mirror (halves already uniform)
  if (G >= 32) v += shfl_xor(v, 16);
  if (G >= 64) v += shfl_xor(v, 32);
  return v;
}

// Block-wide sum of one double per thread -> partials[blockIdx.x] (fixed order, deterministic).
// `red` is an LDS array of >= blockDim.x/64 doubles.
__device__ __forceinline__ void block_publish_partial(double acc, double* red, double* partials) {
  acc = group_sum<64>(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[w];
    partials[blockIdx.x] = s;
  }
}

// Same block sum; when f.counter is set, the LAST block to arrive finishes the global reduction
// (thread t sums partials t, t+256, ... then the fixed tree of bjx_finalize_kernel -> the result is
// independent of the arrival order) and the call is ONE launch.  Hand-off: the "drained sc1 payload + flag" form of
// MI355X_MICROARCH.md ("Workgroup dispatch ... inter-workgroup visibility", price list row handoff-flag): the partial is an
// 8-byte agent-scope (sc1, write-through) store, `s_waitcnt vmcnt(0)` (inline assembly: the compiler may drop a fence's wait
// when it believes the scoreboard empty) acknowledges it, then the relaxed agent-scope arrival RMW; the block that draws the
// last ticket reads the partials with sc1 loads (served at the device-coherent level, never from its L1).
// OFF by default, measured (profiles/r03_finalize_ab.txt, same box, 200 steps): the wait makes every block's publishing wave sit
// through the acknowledgement of its own output stores, and the arrival + last-block reduction are a serial tail behind the
// slowest block — C1 (1024 blocks) 8.1 -> 20.6 us of kernel, 16.6 -> 26.1 us per call; C2 at 2^16 columns 20.5 -> 26.7 us;
// with agent-scope release / acquire FENCES around the hand-off (buffer_wbl2 / buffer_inv: tried in round 3, removed)
// 31.3 and 33.5 us.  Two small follow-up launches cost less than that tail on this chip, for every grid size tried.
// tests/test_gpu_parity.py::test_inkernel_finalize_is_bit_identical_to_two_pass stresses the hand-off under load.
// `flag_lds`: one LDS int for the "I am the last block" broadcast.  Kernels that budget their LDS to the byte (rqs_lds_kernel: 5 blocks
// of 32 KiB per CU) pass a word of their own dynamic allocation; the wrapper below keeps a static one.
// ---- sentinel hand-off (BJX_OPT_INKERNEL_FINALIZE = 2, the default since round 5): ONE launch per call without any wait in the
// publishing blocks.  Every slot of `f.partials` / `f.l2` holds BJX_FIN_SENT (a NaN pattern no sum can produce: sums of NaNs are
// canonicalised before they are published) between launches.  A block publishes its partial with ONE relaxed agent-scope 8-byte
// store — atomic, so no flag and no ordering against its other stores is needed, the wave retires with its output stores still in
// flight.  The block that closes a group of 64 consecutive blocks (index ≡ 63 mod 64, or the last block) polls the 64 slots of
// its group with its first wave (one slot per lane), puts the sentinel back, and publishes the group sum the same way; the LAST
// block polls the group sums (<= 1024: four per thread) and writes the result.  A block only ever waits for blocks with LOWER
// indices, which the dispatcher started before it — they never need the slot the waiting wave occupies, so the wait cannot
// deadlock — an ASSUMPTION about the dispatcher (lower block indices of a grid start first), true on today's queues and stated in
// include/bjx.h, not a guarantee (CU masking, priority pre-emption).  If it breaks, a poll gives up after BJX_FIN_SPIN_MAX rounds: the sum of that
// launch is NaN AND the failure channel is raised (fin_raise): the next entry / bjx_synchronize return BJX_ERR_FINALIZE, the slots are
// re-armed in stream order and the context finishes its sums with the two-pass finalize from then on — never a hung GPU, never a NaN with BJX_OK.  Fixed order (lanes of a group by butterfly, groups t, t+256, ... per thread, wave trees,
// (r0+r1)+(r2+r3)): run-to-run identical, within 1e-15 relative of the two-pass order.
constexpr unsigned long long BJX_FIN_SENT = 0xFFFFDEADFFFFDEADull;
constexpr int BJX_FIN_SPIN_MAX = 1 << 21;
__device__ __forceinline__ void fin_publish(double* slot, double v) {
  unsigned long long b = (v != v) ? 0x7FF8000000000000ull : (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A poll that gives up (BJX_FIN_SPIN_MAX rounds: ~0.3 s) raises the failure channel: the sticky device word `f.err` (every later
// launch on this context then writes NaN, never a sum that may contain what the late block leaves in the slot) and its twin in
// host memory, which the next entry point and bjx_synchronize turn into BJX_ERR_FINALIZE (bjx_ctx.hip: bjx_fin_fault_check).
__device__ __forceinline__ void fin_raise(const BjxFin& f) {      // (inlined: a real call in the epilogue would put every kernel on the callable-function ABI)
  if (f.err) __hip_atomic_store(f.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (f.err_host) __hip_atomic_store(f.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double fin_poll(double* slot, const BjxFin& f) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(slot);
  unsigned long long v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (v == BJX_FIN_SENT && ++spins < BJX_FIN_SPIN_MAX) {
    __builtin_amdgcn_s_sleep(2);
    v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (v == BJX_FIN_SENT) fin_raise(f);
  __hip_atomic_store(q, BJX_FIN_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the slot is ready for the next launch
  return __longlong_as_double((long long)v);
}
// Groups are XCD-LOCAL: block b runs on XCD b % 8 (observed placement, used for speed only — the protocol is correct under any
// placement), and the XCDs work through their shares of the grid at their own pace.  A group of 64 CONSECUTIVE blocks spans all
// eight XCDs, and its closing block then waits for the slowest of them while it holds a CU slot of a faster one: measured
// (round 5, same-process A/B) +5 % on C4, +1 % on C5a instead of the gain.  So a group is 64 blocks of ONE residue class:
// x = b % 8, j = b / 8, group (x, q = j / 64), member l = j % 64; its closing block is the member with l = 63 or the last block
// of the residue class.  Slot of block b: (q * 8 + x) * 64 + l (contiguous per group: one 512-byte run per poll).
__device__ __forceinline__ void block_publish_sentinel(double acc, double* red, const BjxFin& f) {
  acc = group_sum<64>(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (nw > 1) {
    if (lane == 0) red[wave] = acc;
    __syncthreads();
  }
  if (wave != 0) return;
  const unsigned n = gridDim.x, b = blockIdx.x;
  const unsigned x = b & 7u, j = b >> 3;
  const unsigned cnt_x = (n - x + 7u) >> 3;              // blocks of this residue class (>= 1: b is one of them)
  const unsigned q = j >> 6, l = j & 63u, g = q * 8u + x;
  if (lane == 0) {
    double s = acc;
    if (nw > 1) { s = 0.0; for (int w = 0; w < nw; ++w) s += red[w]; }
    if ((int)b != f.drop_block) fin_publish(&f.partials[g * 64u + l], s);
  }
  if (!(l == 63u || j == cnt_x - 1)) return;
  const unsigned members = cnt_x - q * 64u < 64u ? cnt_x - q * 64u : 64u;
  double v = (unsigned)lane < members ? fin_poll(&f.partials[g * 64u + lane], f) : 0.0;
  v = group_sum<64>(v);
  if (lane == 0) fin_publish(&f.l2[g], v);
  if (b != n - 1) return;
  // the last block: every (x, q) that has members, in the fixed order k = q * 8 + x
  const unsigned cnt_0 = (n + 7u) >> 3;                   // the largest residue class
  const unsigned nk = ((cnt_0 + 63u) >> 6) * 8u;          // <= 1032 for 65 536 blocks
  double t = 0.0;
  for (unsigned k = lane; k < nk; k += 64) {
    const unsigned xk = k & 7u, qk = k >> 3;
    const unsigned ck = n > xk ? (n - xk + 7u) >> 3 : 0u;
    if (qk * 64u < ck) t += fin_poll(&f.l2[k], f);
  }
  t = group_sum<64>(t);
  if (lane == 0) {
    t += f.host_const;
    if (f.dev_const) t += *f.dev_const;
    // sticky: a hand-off of THIS or of an EARLIER launch on the context timed out -> no number (the host sees BJX_ERR_FINALIZE)
    if (f.err && __hip_atomic_load(f.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) t = __longlong_as_double(0x7FF8000000000000ll);
    { const double r_ = f.accumulate ? (*f.out + t) : t; *f.out = r_; if (f.out32) *f.out32 = (float)r_; }
  }
}

__device__ __forceinline__ void block_publish_partial_at(double acc, double* red, int* flag_lds, const BjxFin& f) {
  if (!f.partials) return;
  if (f.l2) { block_publish_sentinel(acc, red, f); return; }
  if (!f.counter) { block_publish_partial(acc, red, f.partials); return; }
  int& is_last = *flag_lds;
  acc = group_sum<64>(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[w];
    __hip_atomic_store(&f.partials[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1: write-through to the device-coherent level
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               // the payload store is acknowledged before the flag goes out
    const unsigned prev = __hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev == gridDim.x - 1) ? 1 : 0;
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  if (blockDim.x == 64) {
    // one-wave blocks (seq_wave / colwalk / stacked_mixed kernels): the lane plays lane `l` of each of the FOUR waves of
    // bjx_finalize_kernel in turn, so the order of the additions — and the bits — are those of the two-pass finalize
    double a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = 0.0;
      for (unsigned i = k * 64 + lane; i < gridDim.x; i += 256)
        a[k] += __hip_atomic_load(&f.partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a[k] = group_sum<64>(a[k]);
    }
    if (threadIdx.x == 0) {
      double t = ((a[0] + a[1]) + (a[2] + a[3])) + f.host_const;
      if (f.dev_const) t += *f.dev_const;
      { const double r_ = f.accumulate ? (*f.out + t) : t; *f.out = r_; if (f.out32) *f.out32 = (float)r_; }
      __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  double s = 0.0;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x)
    s += __hip_atomic_load(&f.partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // blockDim.x == 256 for every other kernel that uses the in-kernel finalize (fixed tree of 4 waves)
  s = group_sum<64>(s);
  __syncthreads();
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    if (nw == 4) t = (red[0] + red[1]) + (red[2] + red[3]);
    else for (int w = 0; w < nw; ++w) t += red[w];
    t += f.host_const;
    if (f.dev_const) t += *f.dev_const;
    { const double r_ = f.accumulate ? (*f.out + t) : t; *f.out = r_; if (f.out32) *f.out32 = (float)r_; }
    __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void block_publish_partial(double acc, double* red, const BjxFin& f) {
  __shared__ int is_last_static;
  block_publish_partial_at(acc, red, &is_last_static, f);
}

// two values through one butterfly (shares the wave-uniform branches)
template <class T> __device__ __forceinline__ void group_sum2_rt(T& a, T& b, int G) { a = group_sum_rt(a, G); b = group_sum_rt(b, G); }
template <> __device__ __forceinline__ void group_sum2_rt<float>(float& a, float& b, int G) {
  if (G >= 2) { a = dpp_xadd<0xB1>(a); b = dpp_xadd<0xB1>(b); }
  if (G >= 4) { a = dpp_xadd<0x4E>(a); b = dpp_xadd<0x4E>(b); }
  if (G >= 8) { a = dpp_xadd<0x141>(a); b = dpp_xadd<0x141>(b); }
  if (G >= 16) { a = dpp_xadd<0x140>(a); b = dpp_xadd<0x140>(b); }
  if (G >= 32) { a += shfl_xor(a, 16); b += shfl_xor(b, 16); }
  if (G >= 64) { a += shfl_xor(a, 32); b += shfl_xor(b, 32); }
}

// The same sums WITHOUT control flow: each butterfly stage runs under an all-or-nothing EXEC mask (all ones when the
// group is at least that wide, zero otherwise) instead of behind a wave-uniform branch.  A column loop whose body is
// one basic block lets the compiler count the memory operations in flight across the back edge (s_waitcnt vmcnt(k));
// with branches in the body it drains them all at the loop head.  Every lane of the wave must be active at the call.
struct GroupMasks { int m2, m4, m8, m16, m32, m64; int i16, i32; };   // masks: -1 / 0, the same word for both EXEC halves
__device__ __forceinline__ GroupMasks make_group_masks(int G) {
  GroupMasks g;
  // (built by scalar instructions in inline assembly: a C++ select of a uniform condition may be lowered to a
  // v_cndmask, and the "s" operands of group_sum2_flat would then be handed a VGPR)
  const int Gu = __builtin_amdgcn_readfirstlane(G);
  auto mask = [&](int k) -> int {
    int m;
    asm volatile("s_cmp_ge_i32 %1, %2\n\ts_cselect_b32 %0, -1, 0" : "=s"(m) : "s"(Gu), "s"(k) : "scc");
    return m;
  };
  g.m2 = mask(2); g.m4 = mask(4); g.m8 = mask(8); g.m16 = mask(16); g.m32 = mask(32); g.m64 = mask(64);
  const int lane = (int)(threadIdx.x & 63);
  g.i16 = (lane ^ 16) << 2; g.i32 = (lane ^ 32) << 2;
  return g;
}
#define BJX_DPP2_(ctrl_) \
  "v_add_f32_dpp %[a], %[a], %[a] " ctrl_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %[b], %[b], %[b] " ctrl_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void group_sum2_flat(float& a, float& b, const GroupMasks& g) {
  uint64_t sv;
  float t, u;
  // (a DPP operand needs two wait states after the VALU write of its register: the other value's add and the s_mov
  // sit between two stages of one value; the s_nop covers the first stage)
  asm volatile("s_mov_b64 %[sv], exec\n\t"
               "s_mov_b32 exec_lo, %[m2]\n\ts_mov_b32 exec_hi, %[m2]\n\t"
               "s_nop 1\n\t" BJX_DPP2_("quad_perm:[1,0,3,2]")
               "s_mov_b32 exec_lo, %[m4]\n\ts_mov_b32 exec_hi, %[m4]\n\t" BJX_DPP2_("quad_perm:[2,3,0,1]")
               "s_mov_b32 exec_lo, %[m8]\n\ts_mov_b32 exec_hi, %[m8]\n\t" BJX_DPP2_("row_half_mirror")
               "s_mov_b32 exec_lo, %[m16]\n\ts_mov_b32 exec_hi, %[m16]\n\t" BJX_DPP2_("row_mirror")
               "s_mov_b32 exec_lo, %[m32]\n\ts_mov_b32 exec_hi, %[m32]\n\t"
               "ds_bpermute_b32 %[t], %[i16], %[a]\n\t"
               "ds_bpermute_b32 %[u], %[i16], %[b]\n\t"
               "s_waitcnt lgkmcnt(0)\n\t"
               "v_add_f32 %[a], %[a], %[t]\n\t"
               "v_add_f32 %[b], %[b], %[u]\n\t"
               "s_mov_b32 exec_lo, %[m64]\n\ts_mov_b32 exec_hi, %[m64]\n\t"
               "ds_bpermute_b32 %[t], %[i32], %[a]\n\t"
               "ds_bpermute_b32 %[u], %[i32], %[b]\n\t"
               "s_waitcnt lgkmcnt(0)\n\t"
               "v_add_f32 %[a], %[a], %[t]\n\t"
               "v_add_f32 %[b], %[b], %[u]\n\t"
               "s_mov_b64 exec, %[sv]"
               : [a] "+v"(a), [b] "+v"(b), [t] "=&v"(t), [u] "=&v"(u), [sv] "=&s"(sv)
               : [m2] "s"(g.m2), [m4] "s"(g.m4), [m8] "s"(g.m8), [m16] "s"(g.m16), [m32] "s"(g.m32), [m64] "s"(g.m64), [i16] "v"(g.i16), [i32] "v"(g.i32)
               : "memory");
}
#undef BJX_DPP2_
// Float64: the partner's halves are fetched under the mask into zeroed registers and added outside (x + 0.0 = x).
#define BJX_DPPMOV4_(ctrl_) \
  "v_mov_b32_dpp %[t0], %[a0] " ctrl_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_mov_b32_dpp %[t1], %[a1] " ctrl_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_mov_b32_dpp %[u0], %[b0] " ctrl_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_mov_b32_dpp %[u1], %[b1] " ctrl_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
template <int STAGE> __device__ __forceinline__ void group_stage2_flat(double& a, double& b, int m, int idx) {
  int a0 = __double2loint(a), a1 = __double2hiint(a), b0 = __double2loint(b), b1 = __double2hiint(b);
  int t0 = 0, t1 = 0, u0 = 0, u1 = 0;
  uint64_t sv;
#define BJX_ST_(body_) \
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[m]\n\ts_mov_b32 exec_hi, %[m]\n\ts_nop 1\n\t" body_ "s_mov_b64 exec, %[sv]" \
               : [t0] "+v"(t0), [t1] "+v"(t1), [u0] "+v"(u0), [u1] "+v"(u1), [sv] "=&s"(sv) \
               : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [m] "s"(m), [ix] "v"(idx) : "memory")
  if constexpr (STAGE == 0) BJX_ST_(BJX_DPPMOV4_("quad_perm:[1,0,3,2]"));
  else if constexpr (STAGE == 1) BJX_ST_(BJX_DPPMOV4_("quad_perm:[2,3,0,1]"));
  else if constexpr (STAGE == 2) BJX_ST_(BJX_DPPMOV4_("row_half_mirror"));
  else if constexpr (STAGE == 3) BJX_ST_(BJX_DPPMOV4_("row_mirror"));
  else BJX_ST_("ds_bpermute_b32 %[t0], %[ix], %[a0]\n\tds_bpermute_b32 %[t1], %[ix], %[a1]\n\t"
               "ds_bpermute_b32 %[u0], %[ix], %[b0]\n\tds_bpermute_b32 %[u1], %[ix], %[b1]\n\ts_waitcnt lgkmcnt(0)\n\t");
#undef BJX_ST_
  a += __hiloint2double(t1, t0);
  b += __hiloint2double(u1, u0);
}
#undef BJX_DPPMOV4_
__device__ __forceinline__ void group_sum2_flat(double& a, double& b, const GroupMasks& g) {
  group_stage2_flat<0>(a, b, g.m2, 0);
  group_stage2_flat<1>(a, b, g.m4, 0);
  group_stage2_flat<2>(a, b, g.m8, 0);
  group_stage2_flat<3>(a, b, g.m16, 0);
  group_stage2_flat<4>(a, b, g.m32, g.i16);
  group_stage2_flat<4>(a, b, g.m64, g.i32);
}

// 16-byte vector types per element type
template <class T> struct Vec16;
typedef float bjx_f32x4 __attribute__((ext_vector_type(4)));
typedef double bjx_f64x2 __attribute__((ext_vector_type(2)));
template <> struct Vec16<float> { using type = bjx_f32x4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = bjx_f64x2; static constexpr int N = 2; };

template <class T, int V> struct Pack { T v[V]; };

template <class T, int V, bool NT> __device__ __forceinline__ Pack<T, V> load_pack(const T* p) {
  Pack<T, V> r;
  if constexpr (V == 1) {
    r.v[0] = NT ? __builtin_nontemporal_load(p) : *p;
  } else {
    using VT = typename Vec16<T>::type;
    static_assert(V == Vec16<T>::N, "pack width");
    VT t = NT ? __builtin_nontemporal_load(reinterpret_cast<const VT*>(p)) : *reinterpret_cast<const VT*>(p);
    __builtin_memcpy(&r, &t, sizeof(t));
  }
  return r;
}
template <class T, int V, bool NT> __device__ __forceinline__ void store_pack(T* p, const Pack<T, V>& r) {
  if constexpr (V == 1) {
    if (NT) __builtin_nontemporal_store(r.v[0], p); else *p = r.v[0];
  } else {
    using VT = typename Vec16<T>::type;
    VT t;
    __builtin_memcpy(&t, &r, sizeof(t));
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<VT*>(p)); else *reinterpret_cast<VT*>(p) = t;
  }
}

// A whole short column as one object, element-aligned: the compiler moves it as dwordx2 / x3 / x4 pieces (global accesses need
// only element alignment), so a lane reads its column of 2 ... 13 rows in one to four instructions
template <class T, int DIM> struct __attribute__((aligned(sizeof(T)))) TinyCol { T v[DIM]; };

// elements [j0, j0 + n) of a pack to p[j0 ...] as ONE multi-dword store (n = 1 ... V-1; the tail rows of an odd column: n is the
// same for every column, so each case is one instruction for the wave instead of n one-dword stores)
template <class T, int V> __device__ __forceinline__ void store_pack_run(T* p, const Pack<T, V>& r, int j0, int n) {
  if constexpr (V == 4) {
    if (n == 3) { TinyCol<T, 3> t; if (j0 == 0) { t.v[0] = r.v[0]; t.v[1] = r.v[1]; t.v[2] = r.v[2]; } else { t.v[0] = r.v[1]; t.v[1] = r.v[2]; t.v[2] = r.v[3]; } *reinterpret_cast<TinyCol<T, 3>*>(p + j0) = t; }
    else if (n == 2) { TinyCol<T, 2> t; t.v[0] = j0 == 0 ? r.v[0] : r.v[2]; t.v[1] = j0 == 0 ? r.v[1] : r.v[3]; *reinterpret_cast<TinyCol<T, 2>*>(p + j0) = t; }
    else if (n == 1) p[j0] = j0 == 0 ? r.v[0] : r.v[3];
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) if (j >= j0 && j < j0 + n) p[j] = r.v[j];
  }
}

// Buffer-addressed packs (raw buffer, stride 0): address = base + voffset, and an access whose voffset + size exceeds
// the descriptor's extent reads zeros / is dropped.  Streaming (nt) like the NT packs above.
typedef unsigned int bjx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bjx_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bjx_make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <class T, int V> __device__ __forceinline__ Pack<T, V> buf_load_pack(__amdgpu_buffer_rsrc_t r, int voffset) {
  Pack<T, V> p;
  constexpr int B = V * (int)sizeof(T);
  static_assert(B == 4 || B == 8 || B == 16, "pack bytes");
  if constexpr (B == 16) { const bjx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, 0, 2); __builtin_memcpy(&p, &t, 16); }
  else if constexpr (B == 8) { const bjx_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voffset, 0, 2); __builtin_memcpy(&p, &t, 8); }
  else { const unsigned int t = __builtin_amdgcn_raw_buffer_load_b32(r, voffset, 0, 2); __builtin_memcpy(&p, &t, 4); }
  return p;
}
template <class T, int V> __device__ __forceinline__ void buf_store_pack(__amdgpu_buffer_rsrc_t r, int voffset, const Pack<T, V>& p) {
  constexpr int B = V * (int)sizeof(T);
  static_assert(B == 4 || B == 8 || B == 16, "pack bytes");
  if constexpr (B == 16) { bjx_u32x4 t; __builtin_memcpy(&t, &p, 16); __builtin_amdgcn_raw_buffer_store_b128(t, r, voffset, 0, 2); }
  else if constexpr (B == 8) { bjx_u32x2 t; __builtin_memcpy(&t, &p, 8); __builtin_amdgcn_raw_buffer_store_b64(t, r, voffset, 0, 2); }
  else { unsigned int t; __builtin_memcpy(&t, &p, 4); __builtin_amdgcn_raw_buffer_store_b32(t, r, voffset, 0, 2); }
}

// the same with a wave-uniform byte offset in an SGPR (soffset): a run of loads that differ only by a uniform stride shares ONE
// per-lane offset register instead of one 64-bit address pair per load
template <class T, int V> __device__ __forceinline__ Pack<T, V> buf_load_pack_s(__amdgpu_buffer_rsrc_t r, int voffset, int soffset) {
  Pack<T, V> p;
  constexpr int B = V * (int)sizeof(T);
  static_assert(B == 4 || B == 8 || B == 16, "pack bytes");
  if constexpr (B == 16) { const bjx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 2); __builtin_memcpy(&p, &t, 16); }
  else if constexpr (B == 8) { const bjx_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voffset, soffset, 2); __builtin_memcpy(&p, &t, 8); }
  else { const unsigned int t = __builtin_amdgcn_raw_buffer_load_b32(r, voffset, soffset, 2); __builtin_memcpy(&p, &t, 4); }
  return p;
}
template <class T, int V> __device__ __forceinline__ void buf_store_pack_s(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, const Pack<T, V>& p) {
  constexpr int B = V * (int)sizeof(T);
  static_assert(B == 4 || B == 8 || B == 16, "pack bytes");
  if constexpr (B == 16) { bjx_u32x4 t; __builtin_memcpy(&t, &p, 16); __builtin_amdgcn_raw_buffer_store_b128(t, r, voffset, soffset, 2); }
  else if constexpr (B == 8) { bjx_u32x2 t; __builtin_memcpy(&t, &p, 8); __builtin_amdgcn_raw_buffer_store_b64(t, r, voffset, soffset, 2); }
  else { unsigned int t; __builtin_memcpy(&t, &p, 4); __builtin_amdgcn_raw_buffer_store_b32(t, r, voffset, soffset, 2); }
}

}  // namespace bjx

// grid sizing for streaming kernels: enough blocks to fill 256 CUs x 8 resident blocks
inline int bjx_stream_grid(const bjx_ctx* ctx, int64_t work_items, int per_block) {
  int64_t need = (work_items + per_block - 1) / per_block;
  int64_t cap = (int64_t)ctx->num_cu * 8;
  if (cap > BJX_MAX_BLOCKS) cap = BJX_MAX_BLOCKS;
  if (need < 1) need = 1;
  return (int)(need < cap ? need : cap);
}

// MI355X: a workgroup may use all 160 KiB of a CU's LDS; dynamic allocations above 64 KiB are opted into per kernel
constexpr size_t BJX_LDS_MAX = 160 * 1024;
template <class K> inline void bjx_allow_big_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline bool bjx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ bool bjx_aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
