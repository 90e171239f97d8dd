// bjx_stacked.hip — SURVEY.md §8(f) f-4 (first part): `Stacked` with elementwise segments in ONE launch.
//   Stacked(bs, ranges)   src/bijectors/stacked.jl:27-252:  y = vcat(bs[i](x[ranges_in[i]])...),
//                         logabsdetjac = Σ_i sum(logabsdetjac(bs[i], x[ranges_in[i]]))      (:172-196, :236-244)
// This is what `bijector(d)` of a mixed-constraint model hands to link/invlink: every row range of the
// parameter vector has its own (chain of) elementwise bijector(s).  The reference slices, transforms
// and `vcat`s per segment (one allocation per segment per call); launching the chain kernel once per
// segment would touch a few rows of every column per launch (uncoalesced).  Here a 1-block prep
// launch expands the segment list into a per-OUTPUT-row table {source row, <= 4 op kinds, parameters}
// and one streaming pass on the column-group skeleton applies each row's ops (per-lane `switch`; lanes
// of a wave that hold different kinds serialise only over the kinds present).
#include <cstdlib>

#include "bjx_stream.h"

namespace {
using namespace bjx;

template <class T> struct StackedRow {
  int32_t src;                       // input row feeding this output row
  uint32_t kinds;                    // op kinds, 8 bits each, applied from the low byte up (0 = end)
  T p0[BJX_MAX_SEG_OPS];
  T p1[BJX_MAX_SEG_OPS];             // SCALE / SCALE_INV: ± log|a| (the parameter-only log-det term)
};

struct SegDev {                      // device copy of one bjx_segment (pointers are device pointers)
  int64_t in_lo, out_lo, len;
  int32_t n_ops, pad;
  int32_t kind[BJX_MAX_SEG_OPS], plen[BJX_MAX_SEG_OPS];
  double s0[BJX_MAX_SEG_OPS], s1[BJX_MAX_SEG_OPS];
  const void* v0[BJX_MAX_SEG_OPS];
  const void* v1[BJX_MAX_SEG_OPS];
};

// flag[0] |= 1 if some output row is not the identity source (then the main kernel gathers)
template <class T>
__global__ __launch_bounds__(256) void stacked_table_kernel(const SegDev* segs, int n_segs, int64_t dim, StackedRow<T>* tab, int* flag) {
  for (int64_t r = threadIdx.x; r < dim; r += blockDim.x) { tab[r].src = -1; tab[r].kinds = 0; }
  __syncthreads();
  int gather = 0;
  for (int s = 0; s < n_segs; ++s) {
    const SegDev& g = segs[s];
    for (int64_t i = threadIdx.x; i < g.len; i += blockDim.x) {
      StackedRow<T> row;
      row.src = (int32_t)(g.in_lo + i);
      row.kinds = 0;
      if (g.in_lo != g.out_lo) gather = 1;
      for (int k = 0; k < BJX_MAX_SEG_OPS; ++k) {
        row.p0[k] = T(0); row.p1[k] = T(0);
        if (k < g.n_ops) {
          const int kind = g.kind[k];
          row.kinds |= (uint32_t)kind << (8 * k);
          T a = g.plen[k] > 1 ? reinterpret_cast<const T*>(g.v0[k])[i] : (g.plen[k] == 1 && g.v0[k] ? reinterpret_cast<const T*>(g.v0[k])[0] : (T)g.s0[k]);
          T b = (g.plen[k] > 1 && g.v1[k]) ? reinterpret_cast<const T*>(g.v1[k])[i] : (g.plen[k] == 1 && g.v1[k] ? reinterpret_cast<const T*>(g.v1[k])[0] : (T)g.s1[k]);
          if (kind == BJX_OP_SCALE) b = d_log(d_abs(a));                          // scale.jl:26-32
          if (kind == BJX_OP_SCALE_INV) { b = -d_log(d_abs(a)); a = T(1) / a; }     // Scale(inv(a)), scale.jl:15-16
          row.p0[k] = a; row.p1[k] = b;
        }
      }
      tab[g.out_lo + i] = row;
    }
  }
  if (gather) atomicOr(flag, 1);
}

// one op on one element (the scalar form of apply_op in bjx_chain.hip; same reference lines)
template <class T> __device__ __forceinline__ void stacked_op(int kind, T a, T b, T& x, T& l) {
  using F = Fast<T>;
  switch (kind) {
    case BJX_OP_EXP: l += x; x = d_exp(x); break;                                   // exp_log.jl:5-6
    case BJX_OP_LOG: { const T t = d_log(x); l -= t; x = t; } break;                // exp_log.jl:8-9
    case BJX_OP_SHIFT: x = a + x; break;                                            // shift.jl:14
    case BJX_OP_SCALE:
    case BJX_OP_SCALE_INV: x = a * x; l += b; break;                                // scale.jl:13,26-32
    case BJX_OP_LOGIT: {                                                            // logit.jl:15,24
      const T inv = F::rcp(b - a), xa = x - a;
      l -= F::log(xa * (b - x) * inv);
      const T z = xa * inv;
      x = F::log(z * F::rcp(T(1) - z));
    } break;
    case BJX_OP_LOGIT_INV: {                                                        // logit.jl:19
      const T w = b - a, xx = w * f_logistic(x) + a;
      l += F::log((xx - a) * (b - xx) * F::rcp(w));
      x = xx;
    } break;
    case BJX_OP_LEAKY_RELU: { const T J = x < T(0) ? a : T(1); l += d_log(d_abs(J)); x = J * x; } break;   // leaky_relu.jl:25-29
    case BJX_OP_TRUNCATED: {                                                        // truncated.jl:15-31,51-67
      const T xc = d_clamp(x, a, b);
      const bool lb = d_isfinite(a), ub = d_isfinite(b);
      if (lb && ub) { const T inv = F::rcp(b - a), xa = xc - a; l -= F::log(xa * (b - xc) * inv); const T z = xa * inv; x = F::log(z * F::rcp(T(1) - z)); }
      else if (lb) { const T t = F::log(xc - a); l -= t; x = t; }
      else if (ub) { const T t = F::log(b - xc); l -= t; x = t; }
      else x = xc;
    } break;
    case BJX_OP_TRUNCATED_INV: {                                                    // truncated.jl:33-49,71-91
      const bool lb = d_isfinite(a), ub = d_isfinite(b);
      T xx;
      if (lb && ub) { const T ay = d_abs(x); l += F::log(b - a) - ay - T(2) * f_log1pexp(-ay); xx = (b - a) * f_logistic(x) + a; }
      else if (lb) { l += x; xx = F::exp(x) + a; }
      else if (ub) { l += x; xx = b - F::exp(x); }
      else xx = x;
      x = d_clamp(xx, a, b);
    } break;
    case BJX_OP_SIGNFLIP: x = -x; break;                                            // ordered.jl:3
    default: break;                                                                 // identity
  }
}

template <class T, bool GATHER> struct StackedF {
  static constexpr bool kLoadInput = !GATHER;
  const StackedRow<T>* tab;
  int64_t dim;
  int in_lds, max_ops;
  double per_sample_const;
  const double* per_sample_dev;
  __device__ void stage(char* smem) const {
    if (in_lds) {
      const int n16 = (int)(dim * sizeof(StackedRow<T>) / 16);
      const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(tab);
      bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(smem);
      for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
      __syncthreads();
    }
  }
  template <int V> __device__ T apply(const char* smem, Pack<T, V>& p, const T* xcol, int64_t row, int64_t) const {
    const StackedRow<T>* t = in_lds ? reinterpret_cast<const StackedRow<T>*>(smem) : tab;
    T l = T(0);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const StackedRow<T>& e = t[row + j];
      T x = GATHER ? xcol[e.src] : p.v[j];
      uint32_t kinds = e.kinds;
      for (int k = 0; k < max_ops; ++k) {
        stacked_op<T>((int)(kinds & 0xFFu), e.p0[k], e.p1[k], x, l);
        kinds >>= 8;
      }
      p.v[j] = x;
    }
    return l;
  }
};

template <class T>
int stacked_impl(bjx_ctx* ctx, const bjx_segment* segs, int n_segs, const T* x, T* y, T* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch,
                 uint32_t flags) {
  if (dim * batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  // validate on the host: every output row and every input row exactly once (stacked.jl:156-165 checks the lengths)
  int64_t total = 0;
  int max_ops = 0;
  bool gather = false;
  for (int s = 0; s < n_segs; ++s) {
    const bjx_segment& g = segs[s];
    BJX_REQUIRE(ctx, g.len >= 0 && g.in_lo >= 0 && g.out_lo >= 0 && g.in_lo + g.len <= dim && g.out_lo + g.len <= dim, BJX_ERR_SHAPE,
                "bjx_stacked: segment %d [%lld, +%lld) is outside the %lld rows", s, (long long)g.in_lo, (long long)g.len, (long long)dim);
    BJX_REQUIRE(ctx, g.n_ops >= 0 && g.n_ops <= BJX_MAX_SEG_OPS, BJX_ERR_ARG, "bjx_stacked: segment %d has %d ops (max %d)", s, g.n_ops, BJX_MAX_SEG_OPS);
    for (int k = 0; k < g.n_ops; ++k)
      BJX_REQUIRE(ctx, g.ops[k].param_len == 0 || g.ops[k].param_len == 1 || g.ops[k].param_len == g.len, BJX_ERR_SHAPE,
                  "bjx_stacked: segment %d op %d: parameter of length %d for %lld rows", s, k, g.ops[k].param_len, (long long)g.len);
    total += g.len;
    if (g.n_ops > max_ops) max_ops = g.n_ops;
    if (g.in_lo != g.out_lo) gather = true;
  }
  BJX_REQUIRE(ctx, total == dim, BJX_ERR_SHAPE, "input length mismatch (%lld != %lld)", (long long)total, (long long)dim);   // stacked.jl:157
  const size_t seg_bytes = ((size_t)n_segs * sizeof(SegDev) + 63) / 64 * 64;
  const size_t tab_bytes = (size_t)dim * sizeof(StackedRow<T>);
  BJX_REQUIRE(ctx, 64 + seg_bytes + tab_bytes <= BJX_SCRATCH_BYTES, BJX_ERR_UNSUPPORTED, "bjx_stacked: %d segments / %lld rows exceed the context scratch", n_segs, (long long)dim);
  BJX_REQUIRE(ctx, !gather || x != y, BJX_ERR_ARG, "bjx_stacked: in-place is only supported when every segment keeps its rows (ranges_in == ranges_out)");
  // segment list -> device (host staging copy: the caller's array may be reused right after the call)
  SegDev* hseg = static_cast<SegDev*>(malloc(seg_bytes ? seg_bytes : 64));
  BJX_REQUIRE(ctx, hseg, BJX_ERR_ARG, "out of host memory");
  for (int s = 0; s < n_segs; ++s) {
    SegDev d{};
    d.in_lo = segs[s].in_lo; d.out_lo = segs[s].out_lo; d.len = segs[s].len; d.n_ops = segs[s].n_ops;
    for (int k = 0; k < segs[s].n_ops; ++k) {
      d.kind[k] = segs[s].ops[k].kind; d.plen[k] = segs[s].ops[k].param_len;
      d.s0[k] = segs[s].ops[k].p0; d.s1[k] = segs[s].ops[k].p1; d.v0[k] = segs[s].ops[k].v0; d.v1[k] = segs[s].ops[k].v1;
    }
    hseg[s] = d;
  }
  char* sc = static_cast<char*>(ctx->scratch);
  int* flag = reinterpret_cast<int*>(sc);
  SegDev* dseg = reinterpret_cast<SegDev*>(sc + 64);
  StackedRow<T>* tab = reinterpret_cast<StackedRow<T>*>(sc + 64 + seg_bytes);
  hipError_t e = hipMemsetAsync(flag, 0, 64, ctx->stream);
  if (e == hipSuccess && n_segs) e = hipMemcpyAsync(dseg, hseg, (size_t)n_segs * sizeof(SegDev), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // pageable staging buffer: must outlive the copy
  free(hseg);
  BJX_HIP(ctx, e);
  hipLaunchKernelGGL(stacked_table_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, dseg, n_segs, dim, tab, flag);
  BJX_CHECK_LAUNCH(ctx);
  const int lds = tab_bytes <= 48 * 1024 && tab_bytes % 16 == 0 ? 1 : 0;
  const size_t fsm = lds ? tab_bytes : 0;
  if (gather) { StackedF<T, true> f{tab, dim, lds, max_ops, 0.0, nullptr}; return launch_colgroup<T>(ctx, f, fsm, x, y, ladj_ps, ladj_sum, dim, batch, flags, 0.0); }
  StackedF<T, false> f{tab, dim, lds, max_ops, 0.0, nullptr};
  return launch_colgroup<T>(ctx, f, fsm, x, y, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
}
}  // namespace

BJX_API int bjx_stacked(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x, void* y, void* ladj_ps,
                        double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0 && n_segs >= 0, BJX_ERR_SHAPE, "bjx_stacked: negative size");
  BJX_REQUIRE(ctx, (segs || n_segs == 0) && ((x && y) || dim * batch == 0), BJX_ERR_ARG, "bjx_stacked: null pointer");
  BJX_REQUIRE(ctx, dim < ((int64_t)1 << 31), BJX_ERR_UNSUPPORTED, "bjx_stacked: too many rows");
  if (dt == BJX_F32) return stacked_impl<float>(ctx, segs, n_segs, (const float*)x, (float*)y, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return stacked_impl<double>(ctx, segs, n_segs, (const double*)x, (double*)y, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_stacked: bad dtype %d", (int)dt);
}
