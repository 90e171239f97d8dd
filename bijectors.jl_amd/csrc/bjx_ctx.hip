// bjx_ctx.hip — context, deterministic log-det reduction, timing and synthetic-data helpers.
#include <dlfcn.h>

#include <cstdlib>

#include <chrono>
#include <new>
#include <thread>

#include "bjx_internal.h"

// ------------------------------------------------------------------ finalize
// One 256-thread block sums the per-block partials in a fixed order (thread t takes
// t, t+256, ...; then a fixed tree) so the result does not depend on dispatch order.
__global__ __launch_bounds__(256) void bjx_finalize_kernel(const double* __restrict__ partials, int n,
                                                            double* __restrict__ out, double host_const,
                                                            const double* __restrict__ dev_const, int accumulate, float* __restrict__ out32) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  s = bjx::group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = ((red[0] + red[1]) + (red[2] + red[3])) + host_const;
    if (dev_const) t += *dev_const;
    const double r_ = accumulate ? (*out + t) : t;
    *out = r_;
    if (out32) *out32 = (float)r_;
  }
}

// 4 097 ... 65 536 partials (C3: 5 462 blocks; C5a): ONE launch of one 1024-thread block instead of the reduce-slices + finalize
// pair — a step then carries one tail launch, not two (4.8 + 4.7 µs of helper kernels per call next
// to a 100 - 220 µs hot kernel).  Fixed order: thread t takes t, t+1024, ...; wave trees; the 16 wave sums in index order.
__global__ __launch_bounds__(1024) void bjx_finalize_wide_kernel(const double* __restrict__ partials, int n, double* __restrict__ out, double host_const,
                                                                  const double* __restrict__ dev_const, int accumulate, float* __restrict__ out32) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += partials[i];
  s = bjx::group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    t += host_const;
    if (dev_const) t += *dev_const;
    const double r_ = accumulate ? (*out + t) : t;
    *out = r_;
    if (out32) *out32 = (float)r_;
  }
}

// Stage 1 for large grids (one-pack-per-thread kernels publish up to millions of partials): block b
// sums the contiguous slice [b*per, (b+1)*per) in a fixed order -> out[b].
__global__ __launch_bounds__(256) void bjx_reduce_slices_kernel(const double* __restrict__ partials, long n, long per,
                                                                 double* __restrict__ out) {
  __shared__ double red[4];
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double s = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) s += partials[i];
  s = bjx::group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

int bjx_ensure_partials(bjx_ctx* ctx, size_t n) {
  if (n <= ctx->partials_cap) return BJX_OK;
  // growing frees and allocates (both synchronise): not something a stream capture can record — run the step once
  // outside the capture first (CapturedStep does), the buffer then has its size
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "the per-block partials buffer must grow (%zu blocks) inside a graph capture: run the step once before capturing it", n);
  size_t cap = ctx->partials_cap ? ctx->partials_cap : (size_t)BJX_MAX_BLOCKS;
  while (cap < n) cap *= 2;
  if (ctx->partials) BJX_HIP(ctx, hipFree(ctx->partials));   // synchronises: earlier launches are done with it
  ctx->partials = nullptr;
  ctx->partials_cap = 0;
  BJX_HIP(ctx, hipMalloc(&ctx->partials, cap * sizeof(double)));
  ctx->partials_cap = cap;
  return BJX_OK;
}

int bjx_ensure_big_ws(bjx_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->big_ws_bytes) return BJX_OK;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "the matrix workspace must grow (%zu bytes) inside a graph capture: run the step once before capturing it", bytes);
  if (ctx->big_ws) BJX_HIP(ctx, hipFree(ctx->big_ws));     // synchronises: earlier launches are done with it
  ctx->big_ws = nullptr;
  ctx->big_ws_bytes = 0;
  BJX_HIP(ctx, hipMalloc(&ctx->big_ws, bytes));
  ctx->big_ws_bytes = bytes;
  return BJX_OK;
}

int bjx_launch_finalize(bjx_ctx* ctx, int n_partials, double* ladj_sum, double host_const,
                        int use_dev_const, double /*unused*/, uint32_t flags) {
  const double* src = ctx->partials;
  int n = n_partials;
  if (n > BJX_MAX_BLOCKS && n <= BJX_FIN_WIDE_MAX) {
    hipLaunchKernelGGL(bjx_finalize_wide_kernel, dim3(1), dim3(1024), 0, ctx->stream, src, n, ladj_sum, host_const,
                       use_dev_const ? ctx->consts + 1 : nullptr, (flags & BJX_ACCUMULATE) ? 1 : 0, ctx->fin_out32);
    ctx->fin_out32_taken = ctx->fin_out32 ? 1 : 0;
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  }
  if (n > BJX_MAX_BLOCKS) {
    const long per = ((long)n + BJX_MAX_BLOCKS - 1) / BJX_MAX_BLOCKS;
    const int nb = (int)(((long)n + per - 1) / per);
    hipLaunchKernelGGL(bjx_reduce_slices_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->partials, (long)n, per, ctx->partials2);
    BJX_CHECK_LAUNCH(ctx);
    src = ctx->partials2;
    n = nb;
  }
  hipLaunchKernelGGL(bjx_finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, src, n,
                     ladj_sum, host_const, use_dev_const ? ctx->consts + 1 : nullptr,
                     (flags & BJX_ACCUMULATE) ? 1 : 0, ctx->fin_out32);
  ctx->fin_out32_taken = ctx->fin_out32 ? 1 : 0;
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

// ------------------------------------------------------------------ failure channel of the sentinel hand-off
// Every slot back to the sentinel and the device error word to zero, IN STREAM ORDER (after whatever launch faulted, before the
// next one); the host word is cleared here and now — it is only ever set by a kernel that precedes these memsets on the stream.
static hipError_t bjx_fin_rearm(bjx_ctx* ctx) {
  hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->sent_l1), (int)0xFFFFDEAD, (size_t)BJX_FIN_SENT_GROUPS * 64 * 2, ctx->stream);
  if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->sent_l2), (int)0xFFFFDEAD, (size_t)BJX_FIN_SENT_GROUPS * 2, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->fin_counter, 0, 64, ctx->stream);
  if (ctx->fin_err_host) *ctx->fin_err_host = 0u;
  return e;
}

// Called by every entry that asks for a sum (bjx_make_fin) and by bjx_synchronize / bjx_check_state: has a hand-off of an EARLIER
// launch on this context timed out?  Reading pinned host memory costs nothing; the report is asynchronous like a hipError_t
// (the faulted launch itself returned BJX_OK and wrote NaN; this call reports it, repairs the context and launches nothing).
int bjx_fin_fault_check(bjx_ctx* ctx) {
  if (!ctx->fin_err_host || !*ctx->fin_err_host) return BJX_OK;
  ctx->fin_faults++;
  ctx->opt_inkernel_fin = 0;            // after one fault the context finishes its sums with the two follow-up launches
  if (ctx->capturing) return bjx_fail(ctx, BJX_ERR_FINALIZE, "the sentinel hand-off of an earlier launch timed out while a graph capture is open: end the capture, the context re-arms at the next call");
  const hipError_t e = bjx_fin_rearm(ctx);
  if (e != hipSuccess) return bjx_fail(ctx, (int)e, "re-arming the hand-off slots failed: %s", hipGetErrorString(e));
  return bjx_fail(ctx, BJX_ERR_FINALIZE,
                  "the in-kernel finalize (sentinel hand-off) of an earlier launch on this context timed out: a block waited %d poll rounds for a lower-indexed "
                  "block of its own grid (dispatch-order assumption, include/bjx.h). The Σ logabsdetjac of that launch — and of every launch enqueued "
                  "after it until now — is NaN. The slots were re-armed in stream order and the context now uses the two-pass finalize (fault %d).",
                  bjx::BJX_FIN_SPIN_MAX, ctx->fin_faults);
}

// bjx_check_state: every slot must hold the sentinel between launches.  One small launch that reads them all (diagnostics / tests /
// after a hipError_t of the caller's own: not on the hot path).
__global__ __launch_bounds__(256) void bjx_fin_verify_kernel(const unsigned long long* __restrict__ l1, long n1, const unsigned long long* __restrict__ l2, long n2,
                                                              const unsigned* __restrict__ counter, unsigned* err, unsigned* err_host) {
  bool bad = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n1 + n2; i += (long)gridDim.x * 256) {
    const unsigned long long v = i < n1 ? l1[i] : l2[i - n1];
    bad |= (v != bjx::BJX_FIN_SENT);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) bad |= (*counter != 0u);
  if (bad) {
    __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(err_host, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

BJX_API int bjx_check_state(bjx_ctx* ctx) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "bjx_check_state: a graph capture is open on this context");
  hipLaunchKernelGGL(bjx_fin_verify_kernel, dim3(64), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned long long*>(ctx->sent_l1), (long)BJX_FIN_SENT_GROUPS * 64,
                     reinterpret_cast<const unsigned long long*>(ctx->sent_l2), (long)BJX_FIN_SENT_GROUPS, ctx->fin_counter, ctx->fin_err, ctx->fin_err_host_dev);
  BJX_CHECK_LAUNCH(ctx);
  BJX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const unsigned code = *ctx->fin_err_host;
  const int rc = bjx_fin_fault_check(ctx);
  if (rc == BJX_ERR_FINALIZE && code == 2u)
    return bjx_fail(ctx, BJX_ERR_FINALIZE, "bjx_check_state: a hand-off slot did not hold the sentinel between launches (a launch was aborted mid-flight, or the memory was written by "
                    "someone else). The slots were re-armed in stream order and the context now uses the two-pass finalize (fault %d).", ctx->fin_faults);
  return rc;
}

int bjx_make_fin(bjx_ctx* ctx, int64_t grid, double* ladj_sum, double host_const, int use_dev_const, uint32_t flags,
                 BjxFin* fin, bool* second_pass) {
  *fin = BjxFin{};
  *second_pass = false;
  if (!ladj_sum) return BJX_OK;
  { const int rc_f = bjx_fin_fault_check(ctx); if (rc_f) return rc_f; }
  int rc = bjx_ensure_partials(ctx, (size_t)grid);
  if (rc) return rc;
  fin->partials = ctx->partials;
  if (ctx->opt_inkernel_fin == 2 && grid <= BJX_FIN_WIDE_MAX && ctx->sent_l1) {
    // sentinel hand-off: the slots hold BJX_FIN_SENT between launches (set at bjx_create, put back by the polling blocks)
    fin->partials = ctx->sent_l1;
    fin->l2 = ctx->sent_l2;
    fin->out = ladj_sum;
    fin->host_const = host_const;
    fin->dev_const = use_dev_const ? ctx->consts + 1 : nullptr;
    fin->accumulate = (flags & BJX_ACCUMULATE) ? 1 : 0;
    fin->err = ctx->fin_err;
    fin->err_host = ctx->fin_err_host_dev;
    fin->drop_block = ctx->dbg_fin_drop;
    fin->out32 = ctx->fin_out32;
    ctx->fin_out32_taken = ctx->fin_out32 ? 1 : 0;
  } else if (ctx->opt_inkernel_fin == 1 && grid <= BJX_INKERNEL_FIN_MAX) {
    // The arrival counter is zero between launches: the block that draws the last ticket resets it (a launch that faults
    // leaves the HIP context in a sticky error state, so no later launch can see a stale count); nothing is enqueued here.
    fin->counter = ctx->fin_counter;
    fin->out = ladj_sum;
    fin->host_const = host_const;
    fin->dev_const = use_dev_const ? ctx->consts + 1 : nullptr;
    fin->accumulate = (flags & BJX_ACCUMULATE) ? 1 : 0;
    fin->out32 = ctx->fin_out32;
    ctx->fin_out32_taken = ctx->fin_out32 ? 1 : 0;
  } else {
    *second_pass = true;
  }
  return BJX_OK;
}

int bjx_fin_two_pass(bjx_ctx* ctx, int64_t grid, BjxFin* fin, bool* second_pass) {
  if (!fin->partials) return BJX_OK;                     // no sum requested
  if (!fin->l2 && !fin->counter) return BJX_OK;          // already two-pass
  int rc = bjx_ensure_partials(ctx, (size_t)grid);
  if (rc) return rc;
  *fin = BjxFin{};
  fin->partials = ctx->partials;
  *second_pass = true;
  ctx->fin_out32_taken = 0;          // bjx_launch_finalize will take it
  return BJX_OK;
}

// ------------------------------------------------------------------ context
std::atomic<unsigned long long> bjx_g_launches{0};
BJX_API int bjx_version(void) { return BJX_VERSION; }
BJX_API uint64_t bjx_launch_count(void) { return (uint64_t)bjx_g_launches.load(std::memory_order_relaxed); }

BJX_API int bjx_create(int device, void* hip_stream, bjx_ctx** out) {
  if (!out) return BJX_ERR_ARG;
  *out = nullptr;
  bjx_ctx* ctx = new (std::nothrow) bjx_ctx();
  if (!ctx) return BJX_ERR_ARG;
  ctx->device = device;
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  // the caller's current device is put back before returning: the context allocates on ITS device, it does not move the
  // process (a host runtime such as torch keeps its own notion of the current device)
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc(&ctx->partials, sizeof(double) * BJX_MAX_BLOCKS);
  if (e == hipSuccess) ctx->partials_cap = BJX_MAX_BLOCKS;
  if (e == hipSuccess) e = hipMalloc(&ctx->partials2, sizeof(double) * BJX_MAX_BLOCKS);
  if (e == hipSuccess) e = hipMalloc(&ctx->consts, sizeof(double) * BJX_CONSTS);
  if (e == hipSuccess) e = hipMalloc(&ctx->fin_counter, 64);
  if (e == hipSuccess) ctx->fin_err = ctx->fin_counter + 4;
  if (e == hipSuccess) e = hipMalloc(&ctx->sent_l1, sizeof(double) * BJX_FIN_SENT_GROUPS * 64);
  if (e == hipSuccess) e = hipMalloc(&ctx->sent_l2, sizeof(double) * BJX_FIN_SENT_GROUPS);
  if (e == hipSuccess) {
    void* hp = nullptr;
    e = hipHostMalloc(&hp, 64, hipHostMallocMapped);
    if (e == hipSuccess) {
      ctx->fin_err_host = static_cast<volatile unsigned*>(hp);
      *ctx->fin_err_host = 0u;
      void* dp = nullptr;
      e = hipHostGetDevicePointer(&dp, hp, 0);
      ctx->fin_err_host_dev = static_cast<unsigned*>(dp);
    }
  }
  // the slots get their sentinel ON THE CONTEXT'S STREAM (a null-stream memset is not ordered before the first launch on a
  // non-blocking stream); bjx_set_stream keeps the order when the stream changes
  if (e == hipSuccess) e = bjx_fin_rearm(ctx);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->stream_ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(&ctx->scratch, BJX_SCRATCH_BYTES);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
  if (e == hipSuccess) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device);
  if (e != hipSuccess) {
    int code = (int)e;
    bjx_destroy(ctx);
    return code;
  }
  *out = ctx;
  return BJX_OK;
}

BJX_API int bjx_destroy(bjx_ctx* ctx) {
  if (!ctx) return BJX_OK;
  bjx_comm_destroy(ctx);
  if (ctx->partials) (void)hipFree(ctx->partials);
  if (ctx->partials2) (void)hipFree(ctx->partials2);
  if (ctx->consts) (void)hipFree(ctx->consts);
  if (ctx->fin_counter) (void)hipFree(ctx->fin_counter);
  if (ctx->fin_err_host) (void)hipHostFree(const_cast<unsigned*>(ctx->fin_err_host));
  if (ctx->stream_ev) (void)hipEventDestroy(ctx->stream_ev);
  if (ctx->sent_l1) (void)hipFree(ctx->sent_l1);
  if (ctx->sent_l2) (void)hipFree(ctx->sent_l2);
  if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
  if (ctx->stage_ev) (void)hipEventDestroy(ctx->stage_ev);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->big_ws) (void)hipFree(ctx->big_ws);
  if (ctx->scale_slot.buf) (void)hipFree(ctx->scale_slot.buf);
  for (auto& sl : ctx->rqs_slots) if (sl.buf) (void)hipFree(sl.buf);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->prof_ev) {
    for (int i = 0; i < 2 * bjx_ctx::PROF_MAX; ++i) if (ctx->prof_ev[i]) (void)hipEventDestroy(ctx->prof_ev[i]);
    delete[] ctx->prof_ev;
  }
  delete ctx;
  return BJX_OK;
}

// ------------------------------------------------------------------ launch plans (include/bjx.h "plans")
__global__ void bjx_cast_sum_kernel(const double* __restrict__ src, float* __restrict__ dst) { *dst = (float)*src; }

static int bjx_plan_new(bjx_ctx* ctx, bjx_plan** out, bjx_plan** p) {
  if (!ctx || !out) return BJX_ERR_ARG;
  *out = nullptr;
  *p = new (std::nothrow) bjx_plan();
  if (!*p) return bjx_fail(ctx, BJX_ERR_ARG, "out of host memory");
  (*p)->ctx = ctx;
  return BJX_OK;
}

BJX_API int bjx_plan_chain(bjx_ctx* ctx, bjx_dtype dt, const bjx_op* ops, int n_ops, int64_t dim, uint32_t flags, bjx_plan** out) {
  if (!ctx || !out) return BJX_ERR_ARG;
  *out = nullptr;
  BJX_REQUIRE(ctx, n_ops >= 0 && n_ops <= BJX_MAX_OPS && (ops || n_ops == 0), BJX_ERR_ARG, "bjx_plan_chain: n_ops must be in [0, %d]", BJX_MAX_OPS);
  BJX_REQUIRE(ctx, dt == BJX_F32 || dt == BJX_F64, BJX_ERR_ARG, "bjx_plan_chain: bad dtype %d", (int)dt);
  BJX_REQUIRE(ctx, dim >= 0, BJX_ERR_SHAPE, "bjx_plan_chain: negative size");
  for (int k = 0; k < n_ops; ++k) {
    BJX_REQUIRE(ctx, ops[k].kind >= BJX_OP_EXP && ops[k].kind <= BJX_OP_STDNORMAL_LOGPDF, BJX_ERR_ARG, "bjx_plan_chain: op %d has unknown kind %d", k, (int)ops[k].kind);
    BJX_REQUIRE(ctx, ops[k].param_len == 0 || ops[k].param_len == 1 || ops[k].param_len == dim, BJX_ERR_SHAPE,
                "bjx_plan_chain: op %d has a parameter of length %d for %lld rows", k, (int)ops[k].param_len, (long long)dim);
  }
  bjx_plan* p;
  { const int rc = bjx_plan_new(ctx, out, &p); if (rc) return rc; }
  p->kind = BJX_PLAN_CHAIN; p->dt = dt; p->n_ops = n_ops; p->dim = dim; p->flags = flags;
  for (int k = 0; k < n_ops; ++k) p->ops[k] = ops[k];
  *out = p;
  return BJX_OK;
}

BJX_API int bjx_plan_structured(bjx_ctx* ctx, bjx_dtype dt, int kind, int inverse, int64_t dim, uint32_t flags, bjx_plan** out) {
  if (!ctx || !out) return BJX_ERR_ARG;
  *out = nullptr;
  BJX_REQUIRE(ctx, kind == BJX_PLAN_SIMPLEX || kind == BJX_PLAN_ORDERED, BJX_ERR_ARG, "bjx_plan_structured: kind must be BJX_PLAN_SIMPLEX or BJX_PLAN_ORDERED, got %d", kind);
  BJX_REQUIRE(ctx, dt == BJX_F32 || dt == BJX_F64, BJX_ERR_ARG, "bjx_plan_structured: bad dtype %d", (int)dt);
  BJX_REQUIRE(ctx, dim >= 0, BJX_ERR_SHAPE, "bjx_plan_structured: negative size");
  bjx_plan* p;
  { const int rc = bjx_plan_new(ctx, out, &p); if (rc) return rc; }
  p->kind = kind; p->dt = dt; p->inverse = inverse ? 1 : 0; p->dim = dim; p->flags = flags;
  *out = p;
  return BJX_OK;
}

BJX_API int bjx_plan_stacked_vjp(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, int64_t dim, bjx_plan** out) {
  if (!ctx || !out) return BJX_ERR_ARG;
  *out = nullptr;
  BJX_REQUIRE(ctx, dt == BJX_F32 || dt == BJX_F64, BJX_ERR_ARG, "bjx_plan_stacked_vjp: bad dtype %d", (int)dt);
  BJX_REQUIRE(ctx, segs && n_segs >= 1 && n_segs <= 4096 && dim >= 0, BJX_ERR_ARG, "bjx_plan_stacked_vjp: 1 ... 4096 segments");
  for (int k = 0; k < n_segs; ++k)
    BJX_REQUIRE(ctx, segs[k].n_ops >= 0 && segs[k].n_ops <= BJX_MAX_SEG_OPS && segs[k].len >= 0, BJX_ERR_ARG, "bjx_plan_stacked_vjp: segment %d is malformed", k);
  bjx_plan* p;
  { const int rc = bjx_plan_new(ctx, out, &p); if (rc) return rc; }
  p->segs = new (std::nothrow) bjx_segment[n_segs];
  if (!p->segs) { delete p; return bjx_fail(ctx, BJX_ERR_ARG, "out of host memory"); }
  for (int k = 0; k < n_segs; ++k) p->segs[k] = segs[k];
  p->kind = BJX_PLAN_STACKED_VJP; p->dt = dt; p->n_segs = n_segs; p->dim = dim;
  *out = p;
  return BJX_OK;
}

BJX_API int bjx_plan_stacked(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, int64_t dim, uint32_t flags, bjx_plan** out) {
  // same record as the pullback plan, run by bjx_plan_run through bjx_stacked
  const int rc = bjx_plan_stacked_vjp(ctx, dt, segs, n_segs, dim, out);
  if (rc) return rc;
  (*out)->kind = BJX_PLAN_STACKED;
  (*out)->flags = flags;
  return BJX_OK;
}

BJX_API int bjx_plan_run_vjp(bjx_plan* plan, const void* x, const void* y_bar, const void* ladj_bar, void* x_bar, int64_t batch) {
  if (!plan || !plan->ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(plan->ctx, plan->kind == BJX_PLAN_STACKED_VJP, BJX_ERR_ARG, "bjx_plan_run_vjp: not a pullback plan (kind %d)", plan->kind);
  return bjx_stacked_vjp(plan->ctx, plan->dt, plan->segs, plan->n_segs, x, y_bar, ladj_bar, x_bar, plan->dim, batch);
}

BJX_API int bjx_plan_destroy(bjx_plan* plan) {
  delete plan;
  return BJX_OK;
}

BJX_API int bjx_plan_run(bjx_plan* plan, const void* in, void* out, void* ladj_ps, double* ladj_sum, void* ladj_sum_t, int64_t batch) {
  if (!plan || !plan->ctx) return BJX_ERR_ARG;
  bjx_ctx* ctx = plan->ctx;
  BJX_REQUIRE(ctx, plan->kind != BJX_PLAN_STACKED_VJP, BJX_ERR_ARG, "bjx_plan_run: a pullback plan runs through bjx_plan_run_vjp");
  double* sum = ladj_sum;
  if (ladj_sum_t) {
    BJX_REQUIRE(ctx, plan->dt == BJX_F32, BJX_ERR_ARG, "bjx_plan_run: ladj_sum_t is the Float32 copy of the sum; a Float64 plan returns it in ladj_sum");
    if (!sum) {
      BJX_REQUIRE(ctx, !(plan->flags & BJX_ACCUMULATE), BJX_ERR_ARG, "bjx_plan_run: BJX_ACCUMULATE needs the Float64 accumulator ladj_sum");
      sum = reinterpret_cast<double*>(ctx->fin_counter + 8);          // the context's own 8-byte slot (stream-ordered reuse)
    }
    ctx->fin_out32 = static_cast<float*>(ladj_sum_t);
    ctx->fin_out32_taken = 0;
  }
  int rc;
  if (plan->kind == BJX_PLAN_CHAIN) rc = bjx_chain(ctx, plan->dt, plan->ops, plan->n_ops, in, out, ladj_ps, sum, plan->dim, batch, plan->flags);
  else if (plan->kind == BJX_PLAN_STACKED) rc = bjx_stacked(ctx, plan->dt, plan->segs, plan->n_segs, in, out, ladj_ps, sum, plan->dim, batch, plan->flags);
  else if (plan->kind == BJX_PLAN_SIMPLEX) rc = bjx_simplex(ctx, plan->dt, plan->inverse, in, out, ladj_ps, sum, plan->inverse ? plan->dim + 1 : plan->dim, batch, plan->flags);
  else rc = bjx_ordered(ctx, plan->dt, plan->inverse, in, out, ladj_ps, sum, plan->dim, batch, plan->flags);
  if (ladj_sum_t) {
    float* dst = ctx->fin_out32;
    const int taken = ctx->fin_out32_taken;
    ctx->fin_out32 = nullptr;
    ctx->fin_out32_taken = 0;
    if (rc == BJX_OK && !taken) {          // a path that finished its sum without the shared epilogue (constant log-dets, empty inputs)
      hipLaunchKernelGGL(bjx_cast_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, sum, dst);
      BJX_CHECK_LAUNCH(ctx);
    }
  }
  return rc;
}

BJX_API int bjx_set_stream(bjx_ctx* ctx, void* hip_stream) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "bjx_set_stream: a graph capture is open on this context");
  hipStream_t next = static_cast<hipStream_t>(hip_stream);
  if (next != ctx->stream) {
    // The context's scratch (partials, hand-off slots, parameter tables) is shared by everything it launches: work already enqueued
    // on the old stream must be finished before a launch on the new one touches it.  An event, not a host wait.
    BJX_HIP(ctx, hipEventRecord(ctx->stream_ev, ctx->stream));
    BJX_HIP(ctx, hipStreamWaitEvent(next, ctx->stream_ev, 0));
  }
  ctx->stream = next;
  ctx->scale_slot.epoch = 0;          // cached parameter tables were built on the old stream: rebuild on first use
  for (auto& sl : ctx->rqs_slots) sl.epoch = 0;
  return BJX_OK;
}

BJX_API const char* bjx_last_error(bjx_ctx* ctx) { return ctx ? ctx->err : "null context"; }

BJX_API size_t bjx_workspace_bytes(bjx_ctx* ctx) {
  (void)ctx;
  return sizeof(double) * ((ctx ? ctx->partials_cap : (size_t)BJX_MAX_BLOCKS) + BJX_MAX_BLOCKS + BJX_CONSTS) + BJX_SCRATCH_BYTES + (ctx ? ctx->big_ws_bytes : 0);
}

BJX_API int bjx_set_option(bjx_ctx* ctx, int option, int value) {
  if (!ctx) return BJX_ERR_ARG;
  if (option == BJX_OPT_INKERNEL_FINALIZE) {
    BJX_REQUIRE(ctx, value >= 0 && value <= 2, BJX_ERR_ARG, "BJX_OPT_INKERNEL_FINALIZE: 0 (two-pass), 1 (arrival ticket) or 2 (sentinel hand-off), got %d", value);
    ctx->opt_inkernel_fin = value;
    return BJX_OK;
  }
  if (option == BJX_OPT_PARAM_EPOCH) {
    BJX_REQUIRE(ctx, value >= 0, BJX_ERR_ARG, "BJX_OPT_PARAM_EPOCH: an epoch >= 0 (0 = no reuse of parameter-derived tables)");
    ctx->param_epoch = value;
    return BJX_OK;
  }
  if (option == BJX_OPT_DEBUG_FIN_DROP_BLOCK) {
    ctx->dbg_fin_drop = value;          // < 0: off
    return BJX_OK;
  }
  if (option == BJX_OPT_DEBUG_FIN_POISON_SLOT) {
    BJX_REQUIRE(ctx, value >= 0 && value < BJX_FIN_SENT_GROUPS * 64, BJX_ERR_ARG, "BJX_OPT_DEBUG_FIN_POISON_SLOT: slot index out of range");
    const double one = 1.0;
    BJX_HIP(ctx, hipMemcpyAsync(ctx->sent_l1 + value, &one, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    BJX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BJX_OK;
  }
  if (option == BJX_OPT_COLLECTIVE_TIMEOUT_MS) {
    BJX_REQUIRE(ctx, value >= 0, BJX_ERR_ARG, "BJX_OPT_COLLECTIVE_TIMEOUT_MS: milliseconds >= 0 (0 = wait for ever)");
    ctx->collective_timeout_ms = value;
    return BJX_OK;
  }
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_set_option: unknown option %d", option);
}

BJX_API int bjx_synchronize(bjx_ctx* ctx) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "bjx_synchronize: a graph capture is open on this context");
  if (ctx->comm && ctx->collective_timeout_ms > 0) {
    // Watchdog (BJX_OPT_COLLECTIVE_TIMEOUT_MS): a rank that never arrives leaves ncclAllReduce spinning on the stream for ever.
    // Poll instead of blocking; on time-out abort the communicator (ncclCommAbort releases the kernel) and report an error —
    // the caller gets a status, not a hang.  Polling costs nothing on the launch path: only this explicit wait uses it.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipSuccess) return bjx_fin_fault_check(ctx);
      if (q != hipErrorNotReady) return bjx_fail(ctx, (int)q, "bjx_synchronize: %s", hipGetErrorString(q));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms > (double)ctx->collective_timeout_ms) {
        typedef int (*fn_abort)(void*);
        fn_abort ab = (fn_abort)dlsym(ctx->rccl_handle, "ncclCommAbort");
        if (ab) ab(ctx->comm);
        ctx->comm = nullptr;
        ctx->nranks = 1;
        return bjx_fail(ctx, 1000 + 6 /* ncclRemoteError */, "bjx_synchronize: the stream did not drain within %d ms with a communicator of %s ranks attached: a collective is stuck (a rank did not arrive); the communicator was aborted",
                        ctx->collective_timeout_ms, "several");
      }
      std::this_thread::sleep_for(std::chrono::microseconds(ms < 5.0 ? 20 : 500));
    }
  }
  BJX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return bjx_fin_fault_check(ctx);
}

BJX_API int bjx_time_begin(bjx_ctx* ctx) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return BJX_OK;
}

BJX_API int bjx_time_end(bjx_ctx* ctx, float* ms_out) {
  if (!ctx || !ms_out) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "bjx_time_end: a graph capture is open on this context");
  BJX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  BJX_HIP(ctx, hipEventSynchronize(ctx->ev1));
  BJX_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
  return BJX_OK;
}

BJX_API int bjx_kernel_time_begin(bjx_ctx* ctx) {
  if (!ctx) return BJX_ERR_ARG;
  if (!ctx->prof_ev) {
    ctx->prof_ev = new (std::nothrow) hipEvent_t[2 * bjx_ctx::PROF_MAX]();
    if (!ctx->prof_ev) return bjx_fail(ctx, BJX_ERR_ARG, "out of host memory");
    for (int i = 0; i < 2 * bjx_ctx::PROF_MAX; ++i) BJX_HIP(ctx, hipEventCreate(&ctx->prof_ev[i]));
  }
  ctx->prof_n = 0;
  ctx->prof_dropped = 0;
  ctx->prof_on = 1;
  return BJX_OK;
}

BJX_API int bjx_kernel_time_end(bjx_ctx* ctx, float* total_ms, int* launches) {
  if (!ctx || !total_ms || !launches) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "bjx_kernel_time_end: a graph capture is open on this context");
  ctx->prof_on = 0;
  BJX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  for (int i = 0; i < ctx->prof_n; ++i) {
    float ms = 0.f;
    BJX_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
    tot += ms;
  }
  *total_ms = (float)tot;
  *launches = ctx->prof_n;
  if (ctx->prof_dropped) return bjx_fail(ctx, BJX_ERR_UNSUPPORTED, "bjx_kernel_time_end: %d launches not recorded (more than %d in the region)", ctx->prof_dropped, bjx_ctx::PROF_MAX);
  return BJX_OK;
}

// ------------------------------------------------------------------ captured steps (hipGraph)
struct bjx_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

BJX_API int bjx_graph_begin(bjx_ctx* ctx) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_ARG, "bjx_graph_begin: a capture is already open on this context");
  BJX_REQUIRE(ctx, ctx->stream != nullptr, BJX_ERR_UNSUPPORTED, "bjx_graph_begin: the NULL stream cannot be captured; give the context a stream");
  BJX_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  ctx->capturing = 1;
  return BJX_OK;
}

BJX_API int bjx_graph_end(bjx_ctx* ctx, bjx_graph** out) {
  if (!ctx || !out) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, ctx->capturing, BJX_ERR_ARG, "bjx_graph_end: no capture is open on this context");
  ctx->capturing = 0;
  hipGraph_t g = nullptr;
  BJX_HIP(ctx, hipStreamEndCapture(ctx->stream, &g));
  BJX_REQUIRE(ctx, g != nullptr, BJX_ERR_UNSUPPORTED, "bjx_graph_end: the capture was invalidated (a call inside it synchronised or used another stream)");
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    return bjx_fail(ctx, BJX_ERR_UNSUPPORTED, "hipGraphInstantiate: %s", hipGetErrorString(err));
  }
  bjx_graph* r = new (std::nothrow) bjx_graph();
  if (!r) { (void)hipGraphExecDestroy(e); (void)hipGraphDestroy(g); return bjx_fail(ctx, BJX_ERR_ARG, "out of host memory"); }
  r->graph = g;
  r->exec = e;
  *out = r;
  return BJX_OK;
}

BJX_API int bjx_graph_launch(bjx_ctx* ctx, bjx_graph* graph) {
  if (!ctx || !graph || !graph->exec) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_ARG, "bjx_graph_launch: a capture is open on this context");
  BJX_HIP(ctx, hipGraphLaunch(graph->exec, ctx->stream));
  return BJX_OK;
}

BJX_API int bjx_graph_destroy(bjx_graph* graph) {
  if (!graph) return BJX_OK;
  if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
  if (graph->graph) (void)hipGraphDestroy(graph->graph);
  delete graph;
  return BJX_OK;
}

// ------------------------------------------------------------------ Philox4x32-10 normal fill
// Counter = global element index / 4, key = seed; 4 x u32 -> 2 Box-Muller pairs -> 4 normals.
// Element e of the GLOBAL array (col0*dim + local index) always gets the same value, so a batch
// is identical for any shard count.
namespace {
template <class T>
__global__ __launch_bounds__(256) void fill_normal_kernel(T* __restrict__ out, int64_t n_local, int64_t e0,
                                                           uint64_t seed, double mean, double std) {
  // each thread produces the 4 normals of one Philox counter; counters are aligned to GLOBAL index/4
  const int64_t c_first = e0 >> 2, c_last = (e0 + n_local - 1) >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = c_first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= c_last; c += stride) {
    T z[4];
    bjx::philox_normal4(seed, c, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t e = (c << 2) + j - e0;
      if (e >= 0 && e < n_local) out[e] = (T)(mean + std * z[j]);
    }
  }
}
}  // namespace

BJX_API int bjx_set_rng(bjx_ctx* ctx, uint64_t seed, int64_t col0) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, col0 >= 0, BJX_ERR_ARG, "bjx_set_rng: negative column offset");
  ctx->rng_seed = seed;
  ctx->rng_col0 = col0;
  return BJX_OK;
}

BJX_API int bjx_fill_normal(bjx_ctx* ctx, bjx_dtype dt, void* out, int64_t dim, int64_t batch, int64_t col0,
                            uint64_t seed, double mean, double std) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, out && dim >= 0 && batch >= 0 && col0 >= 0, BJX_ERR_ARG, "bjx_fill_normal: bad argument");
  int64_t n = dim * batch;
  if (n == 0) return BJX_OK;
  int grid = bjx_stream_grid(ctx, (n + 3) / 4, 256);
  if (dt == BJX_F32)
    hipLaunchKernelGGL(fill_normal_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (float*)out, n, col0 * dim, seed, mean, std);
  else if (dt == BJX_F64)
    hipLaunchKernelGGL(fill_normal_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (double*)out, n, col0 * dim, seed, mean, std);
  else
    return bjx_fail(ctx, BJX_ERR_ARG, "bjx_fill_normal: bad dtype %d", (int)dt);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

// ------------------------------------------------------------------ parameter gather for composed flows
// The reference writes a flow as `l8 ∘ … ∘ l1` (docs/src/flows.md:115, composed.jl:4): one PlanarLayer object, i.e. one (w, u, b)
// triple of separate device vectors, per layer.  The host walks the composition and hands the RUN of layers to the fused kernel;
// this entry copies the n vectors into the layer-major table bjx_planar takes, in one launch (pointer table in the kernel
// arguments, 64 vectors per launch).
namespace {
struct BjxPtrTable { const void* p[64]; };
template <class T>
__global__ __launch_bounds__(256) void bjx_pack_vectors_kernel(BjxPtrTable tab, int64_t len, T* __restrict__ dst) {
  const T* __restrict__ src = static_cast<const T*>(tab.p[blockIdx.y]);
  T* __restrict__ out = dst + (int64_t)blockIdx.y * len;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) out[i] = src[i];
}
}  // namespace

BJX_API int bjx_pack_vectors(bjx_ctx* ctx, bjx_dtype dt, int n, const void* const* src, int64_t len, void* dst) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, n >= 0 && len >= 0, BJX_ERR_SHAPE, "bjx_pack_vectors: bad size (n=%d, len=%lld)", n, (long long)len);
  BJX_REQUIRE(ctx, dt == BJX_F32 || dt == BJX_F64, BJX_ERR_ARG, "bjx_pack_vectors: bad dtype %d", (int)dt);
  if (n == 0 || len == 0) return BJX_OK;
  BJX_REQUIRE(ctx, src && dst, BJX_ERR_ARG, "bjx_pack_vectors: null pointer");
  for (int i = 0; i < n; ++i) BJX_REQUIRE(ctx, src[i], BJX_ERR_ARG, "bjx_pack_vectors: vector %d is a null pointer", i);
  const size_t esz = dt == BJX_F32 ? 4 : 8;
  int64_t gx = (len + 255) / 256;
  if (gx > 1024) gx = 1024;
  for (int lo = 0; lo < n; lo += 64) {
    const int cnt = n - lo < 64 ? n - lo : 64;
    BjxPtrTable tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < cnt ? src[lo + i] : nullptr;
    char* d = static_cast<char*>(dst) + (size_t)lo * (size_t)len * esz;
    if (dt == BJX_F32)
      hipLaunchKernelGGL(bjx_pack_vectors_kernel<float>, dim3((unsigned)gx, (unsigned)cnt), dim3(256), 0, ctx->stream, tab, len, reinterpret_cast<float*>(d));
    else
      hipLaunchKernelGGL(bjx_pack_vectors_kernel<double>, dim3((unsigned)gx, (unsigned)cnt), dim3(256), 0, ctx->stream, tab, len, reinterpret_cast<double*>(d));
    BJX_CHECK_LAUNCH(ctx);
  }
  return BJX_OK;
}

// ------------------------------------------------------------------ RCCL (lazy)
// RCCL is dlopen'ed on first use so the library loads (and single-GPU use works) in processes
// that never touch a communicator, and so it shares the RCCL already mapped by the host
// runtime if there is one.
namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_uid)(nccl_uid*);
typedef int (*fn_init)(void**, int, nccl_uid, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;

void* rccl_open() {
  static void* h = nullptr;
  if (h) return h;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) return h;
  }
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) return h;
  }
  return nullptr;
}
}  // namespace

BJX_API int bjx_comm_unique_id(void* out128) {
  if (!out128) return BJX_ERR_ARG;
  void* h = rccl_open();
  if (!h) return BJX_ERR_NOCOMM;
  fn_uid f = (fn_uid)dlsym(h, "ncclGetUniqueId");
  if (!f) return BJX_ERR_NOCOMM;
  nccl_uid id;
  int r = f(&id);
  if (r != 0) return 1000 + r;
  memcpy(out128, &id, sizeof(id));
  return BJX_OK;
}

BJX_API int bjx_comm_init(bjx_ctx* ctx, int nranks, int rank, const void* unique_id128) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, nranks >= 1 && rank >= 0 && rank < nranks && unique_id128, BJX_ERR_ARG, "bjx_comm_init: bad argument");
  void* h = rccl_open();
  BJX_REQUIRE(ctx, h, BJX_ERR_NOCOMM, "bjx_comm_init: cannot dlopen librccl.so: %s", dlerror());
  fn_init f = (fn_init)dlsym(h, "ncclCommInitRank");
  BJX_REQUIRE(ctx, f, BJX_ERR_NOCOMM, "bjx_comm_init: ncclCommInitRank not found");
  BJX_HIP(ctx, hipSetDevice(ctx->device));
  nccl_uid id;
  memcpy(&id, unique_id128, sizeof(id));
  void* comm = nullptr;
  int r = f(&comm, nranks, id, rank);
  if (r != 0) return bjx_fail(ctx, 1000 + r, "ncclCommInitRank failed: %d", r);
  ctx->rccl_handle = h;
  ctx->comm = comm;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return BJX_OK;
}

BJX_API int bjx_comm_destroy(bjx_ctx* ctx) {
  if (!ctx || !ctx->comm) return BJX_OK;
  fn_destroy f = (fn_destroy)dlsym(ctx->rccl_handle, "ncclCommDestroy");
  if (f) f(ctx->comm);
  ctx->comm = nullptr;
  return BJX_OK;
}

BJX_API int bjx_allreduce_sum_f64(bjx_ctx* ctx, double* ptr, int64_t n) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, ptr && n >= 0, BJX_ERR_ARG, "bjx_allreduce_sum_f64: bad argument");
  if (ctx->nranks == 1 && !ctx->comm) return BJX_OK;  // single shard: the sum is already global
  BJX_REQUIRE(ctx, ctx->comm, BJX_ERR_NOCOMM, "bjx_allreduce_sum_f64: call bjx_comm_init first");
  fn_allreduce f = (fn_allreduce)dlsym(ctx->rccl_handle, "ncclAllReduce");
  BJX_REQUIRE(ctx, f, BJX_ERR_NOCOMM, "ncclAllReduce not found");
  int r = f(ptr, ptr, (size_t)n, NCCL_FLOAT64, NCCL_SUM, ctx->comm, ctx->stream);
  if (r != 0) return bjx_fail(ctx, 1000 + r, "ncclAllReduce failed: %d", r);
  return BJX_OK;
}
