// bjx_tiny.hip — OrderedBijector / SimplexBijector on SHORT columns (1 ... 8 rows; ordered.jl:24-80, simplex.jl:28-143)
//
// A 3-class simplex or a pair of ordered cut points is 12 bytes per sample.  The single-wave tile walkers of bjx_seq.hip stage 64 such
// columns — a few hundred bytes — through LDS per block, one memory round trip per array: 32-43 % of the HBM peak at K = 3 ... 4
// (profiles/r03_small_sizes.md).  Here lane = column: the column is one TinyCol object (element-aligned multi-dword accesses, a wave
// instruction still covers one contiguous run), the walk of bjx_seqops.h runs on registers in the reference's order with the row
// count a template parameter (log(K-1-i) are compile-time constants), and the log-det is the lane's own value.
#include <cstdlib>

#include "bjx_internal.h"

namespace {
using namespace bjx;

#include "bjx_seqops.h"

__device__ constexpr double kLogN[9] = {0.0, 0.0, 0.69314718055994530942, 1.09861228866810969140, 1.38629436111989061883, 1.60943791243410037460,
                                        1.79175946922805500081, 1.94591014905531330511, 2.07944154167983592825};

template <class T, class Op, int KI, int KO>
__global__ __launch_bounds__(256) void seq_tiny_kernel(const Op op0, const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps, int64_t batch,
                                                       int accumulate, const BjxFin fin) {
  __shared__ double red[4];
  constexpr int R = KI > KO ? KI : KO;                                // rows of the walk
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  T l = T(0);
  if (col < batch) {
    const TinyCol<T, KI> t = *reinterpret_cast<const TinyCol<T, KI>*>(in + col * KI);
    T v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = i < KI ? t.v[i < KI ? i : 0] : T(0);
    Op op = op0;
    op.init();
    const T lk0 = (T)kLogN[R - 1 > 0 ? R - 1 : 0];                    // log(K-1-0), simplex.jl:35,41
    v[0] = op.first(v[0], &lk0);
    constexpr int mid_end = (Op::HAS_LAST && R > 1) ? R - 1 : R;
#pragma unroll
    for (int i = 1; i < mid_end; ++i) v[i] = op.mid(i, v[i], Op::USES_LOGK ? (T)kLogN[R - 1 - i] : T(0));
    if (Op::HAS_LAST && R > 1) v[R - 1] = op.last(v[R - 1]);
    l = op.result();
    if (out) {
      TinyCol<T, KO> o;
#pragma unroll
      for (int i = 0; i < KO; ++i) o.v[i] = v[i];
      *reinterpret_cast<TinyCol<T, KO>*>(out + col * KO) = o;
    }
    if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + l : l;
  }
  block_publish_partial(col < batch ? (double)l : 0.0, red, fin);
}

template <class T, class Op, int KI, int KO>
int launch_tiny(bjx_ctx* ctx, const Op& op, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t batch, uint32_t flags) {
  const int64_t grid = (batch + 255) / 256;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  BjxFin fin;
  bool second = false;
  { int rc = bjx_make_fin(ctx, grid, ladj_sum, 0.0, 0, flags, &fin, &second); if (rc) return rc; }
  {
    BjxProf prof_(ctx);
    hipLaunchKernelGGL((seq_tiny_kernel<T, Op, KI, KO>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, op, in, out, ladj_ps, batch, (flags & BJX_ACCUMULATE) ? 1 : 0, fin);
  }
  BJX_CHECK_LAUNCH(ctx);
  if (second) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

template <class T, int K>
int tiny_k(bjx_ctx* ctx, int which, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t batch, uint32_t flags) {
  const bool want = ladj_ps || ladj_sum;
  switch (which) {
    case BJX_TALL_ORDERED_FWD: return launch_tiny<T, OrderedFwd<T>, K, K>(ctx, OrderedFwd<T>{}, in, out, ladj_ps, ladj_sum, batch, flags);
    case BJX_TALL_ORDERED_INV: return launch_tiny<T, OrderedInv<T>, K, K>(ctx, OrderedInv<T>{}, in, out, ladj_ps, ladj_sum, batch, flags);
    case BJX_TALL_SIMPLEX_FWD:
      if constexpr (K >= 2) {
        if (want) { SimplexFwd<T, true> op; op.K = K; return launch_tiny<T, SimplexFwd<T, true>, K, K - 1>(ctx, op, in, out, ladj_ps, ladj_sum, batch, flags); }
        SimplexFwd<T, false> op; op.K = K;
        return launch_tiny<T, SimplexFwd<T, false>, K, K - 1>(ctx, op, in, out, ladj_ps, ladj_sum, batch, flags);
      }
      break;
    case BJX_TALL_SIMPLEX_INV:
      if constexpr (K >= 2) {
        if (want) { SimplexInv<T, true> op; op.K = K; return launch_tiny<T, SimplexInv<T, true>, K - 1, K>(ctx, op, in, out, ladj_ps, ladj_sum, batch, flags); }
        SimplexInv<T, false> op; op.K = K;
        return launch_tiny<T, SimplexInv<T, false>, K - 1, K>(ctx, op, in, out, ladj_ps, ladj_sum, batch, flags);
      }
      break;
  }
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_seq_tiny: bad map %d for K = %d", which, K);
}

template <class T>
int tiny_dispatch(bjx_ctx* ctx, int which, int K, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t batch, uint32_t flags) {
  switch (K) {
    case 1: return tiny_k<T, 1>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
    case 2: return tiny_k<T, 2>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
    case 3: return tiny_k<T, 3>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
    case 4: return tiny_k<T, 4>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
    case 5: return tiny_k<T, 5>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
    case 6: return tiny_k<T, 6>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
    case 7: return tiny_k<T, 7>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
    default: return tiny_k<T, 8>(ctx, which, in, out, ladj_ps, ladj_sum, batch, flags);
  }
}
// ---------------------------------------------------------------- pullbacks (the per-row sweeps of simplex_vjp_kernel / ordered_vjp_kernel
// in bjx_seq.hip, on registers with the row count a template parameter)
template <class T, bool INV, int K>
__global__ __launch_bounds__(256) void simplex_vjp_tiny_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                               T* __restrict__ in_bar, int64_t batch) {
  using F = Fast<T>;
  constexpr int KA = INV ? K - 1 : K, KG = INV ? K : K - 1;          // rows of in / in_bar ; of out_bar
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= batch) return;
  const TinyCol<T, KA> ta = *reinterpret_cast<const TinyCol<T, KA>*>(in + col * KA);
  const TinyCol<T, KG> tg = *reinterpret_cast<const TinyCol<T, KG>*>(out_bar + col * KG);
  T a[K], g[K];
#pragma unroll
  for (int i = 0; i < K; ++i) { a[i] = i < KA ? ta.v[i < KA ? i : 0] : T(0); g[i] = i < KG ? tg.v[i < KG ? i : 0] : T(0); }
  const T e = Num<T>::eps;
  const T c = T(1) / (T(1) - 2 * e), E = T(1) + e, c2 = T(1) - 2 * e;
  const T lb = ladj_bar ? ladj_bar[col] : T(0);
  if (INV) {
    // forward recurrence: a[k] <- x_k, s = Σ x
    T s = T(0);
    { const T xk = d_clamp((f_logistic(a[0] - (T)kLogN[K - 1]) - e) * c, T(0), T(1)); a[0] = xk; s = xk; }
#pragma unroll
    for (int k = 1; k < K - 1; ++k) { const T xk = d_clamp((E - s) * c * f_logistic(a[k] - (T)kLogN[K - 1 - k]) - e, T(0), T(1)); a[k] = xk; s += xk; }
    const T last = T(1) - s;
    T sb = (last > T(0) && last < T(1)) ? -g[K - 1] : T(0);
#pragma unroll
    for (int k = K - 2; k >= 0; --k) {
      const T xk = a[k];
      const T sk = s - xk;
      T dtdx, dtds;
      simplex_t_partials<T>(xk, sk, k == 0, dtdx, dtds);
      const T xb = g[k] + sb + lb * dtdx;
      sb += lb * dtds;
      const T ub = (xk > T(0) && xk < T(1)) ? xb : T(0);
      T zb, z;
      if (k == 0) { zb = ub * c; z = xk * c2 + e; }
      else { const T rc = (E - sk) * c; zb = ub * rc; z = (xk + e) * F::rcp(rc); sb -= ub * c * z; }
      g[k] = zb * z * (T(1) - z);
      s = sk;
    }
  } else {
    T s = T(0);
#pragma unroll
    for (int k = 0; k < K - 1; ++k) s += a[k];                         // s_{K-1} = Σ_{j<K-1} x_j
    T sbn = T(0);
    g[K - 1] = T(0);                                                   // row K enters neither y nor the log-det
#pragma unroll
    for (int k = K - 2; k >= 0; --k) {
      const T xk = a[k];
      const T sk = s - xk;
      const T gy = g[k];
      T xb = sbn, sb = sbn;
      if (k == 0) {
        const T zf = xk * c2 + e;
        const T zfb = gy * F::rcp(zf * (T(1) - zf));
        xb += zfb * c2;
      } else {
        const T rd = F::rcp(E - sk);
        const T an = (xk + e) * c2;
        const T zf = an * rd;
        const T zfb = gy * F::rcp(zf * (T(1) - zf));
        xb += zfb * c2 * rd;
        sb += zfb * an * rd * rd;
      }
      T dtdx, dtds;
      simplex_t_partials<T>(xk, sk, k == 0, dtdx, dtds);
      xb -= lb * dtdx;
      sb -= lb * dtds;
      g[k] = xb;
      sbn = sb;
      s = sk;
    }
  }
  TinyCol<T, KA> o;
#pragma unroll
  for (int i = 0; i < KA; ++i) o.v[i] = g[i];
  *reinterpret_cast<TinyCol<T, KA>*>(in_bar + col * KA) = o;
}

template <class T, bool INV, int K>
__global__ __launch_bounds__(256) void ordered_vjp_tiny_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                               T* __restrict__ in_bar, int64_t batch) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= batch) return;
  const TinyCol<T, K> a = *reinterpret_cast<const TinyCol<T, K>*>(in + col * K);
  TinyCol<T, K> g = *reinterpret_cast<const TinyCol<T, K>*>(out_bar + col * K);
  const T lb = ladj_bar ? ladj_bar[col] : T(0);
  if (!INV) {
    T sfx = T(0);
#pragma unroll
    for (int i = K - 1; i >= 1; --i) { sfx += g.v[i]; g.v[i] = sfx * d_exp(a.v[i]) + lb; }
    g.v[0] = sfx + g.v[0];
  } else {
    T nxt = T(0);                                                      // Δ_{i+1}/r_{i+1} of the row above
#pragma unroll
    for (int i = K - 1; i >= 1; --i) {
      const T q = (g.v[i] - lb) / (a.v[i] - a.v[i - 1]);
      g.v[i] = q - nxt;
      nxt = q;
    }
    g.v[0] = g.v[0] - nxt;
  }
  *reinterpret_cast<TinyCol<T, K>*>(in_bar + col * K) = g;
}

template <class T, int K>
int tiny_vjp_k(bjx_ctx* ctx, int simplex, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t batch) {
  const int64_t grid = (batch + 255) / 256;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  {
    BjxProf prof_(ctx);
    if (simplex) {
      if constexpr (K >= 2) {
        if (inverse) hipLaunchKernelGGL((simplex_vjp_tiny_kernel<T, true, K>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, batch);
        else hipLaunchKernelGGL((simplex_vjp_tiny_kernel<T, false, K>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, batch);
      }
    } else {
      if (inverse) hipLaunchKernelGGL((ordered_vjp_tiny_kernel<T, true, K>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, batch);
      else hipLaunchKernelGGL((ordered_vjp_tiny_kernel<T, false, K>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, batch);
    }
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
template <class T>
int tiny_vjp_dispatch(bjx_ctx* ctx, int simplex, int inverse, int K, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t batch) {
  switch (K) {
    case 1: return tiny_vjp_k<T, 1>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
    case 2: return tiny_vjp_k<T, 2>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
    case 3: return tiny_vjp_k<T, 3>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
    case 4: return tiny_vjp_k<T, 4>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
    case 5: return tiny_vjp_k<T, 5>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
    case 6: return tiny_vjp_k<T, 6>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
    case 7: return tiny_vjp_k<T, 7>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
    default: return tiny_vjp_k<T, 8>(ctx, simplex, inverse, in, out_bar, ladj_bar, in_bar, batch);
  }
}
}  // namespace

// Pullbacks of the Ordered (simplex = 0) / Simplex (simplex = 1) maps on columns of K <= 8 rows; *taken as above
int bjx_seq_tiny_vjp(bjx_ctx* ctx, bjx_dtype dt, int simplex, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K,
                     int64_t batch, bool* taken) {
  *taken = false;
  static const int use_tiny = getenv("BJX_SEQ_TINY") ? atoi(getenv("BJX_SEQ_TINY")) : 1;
  static const int kmax = 8;
  if (!use_tiny || batch <= 0 || K < (simplex ? 2 : 1) || K > kmax || K > 8 || (const void*)in == (const void*)in_bar) return BJX_OK;
  *taken = true;
  if (dt == BJX_F32) return tiny_vjp_dispatch<float>(ctx, simplex, inverse, (int)K, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, batch);
  return tiny_vjp_dispatch<double>(ctx, simplex, inverse, (int)K, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, batch);
}

// Columns of K rows (K = the larger of the input and output heights), contiguous.  Same contract as bjx_tall_stream.
int bjx_seq_tiny(bjx_ctx* ctx, bjx_dtype dt, int which, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags,
                 bool* taken) {
  *taken = false;
  static const int use_tiny = getenv("BJX_SEQ_TINY") ? atoi(getenv("BJX_SEQ_TINY")) : 1;
  static const int kmax = 8;
  const bool simplex = which == BJX_TALL_SIMPLEX_FWD || which == BJX_TALL_SIMPLEX_INV;
  if (!use_tiny || batch <= 0 || K < (simplex ? 2 : 1) || K > kmax || K > 8 || !in) return BJX_OK;
  if (which == BJX_TALL_SIMPLEX_INV && !out) return BJX_OK;
  *taken = true;
  if (dt == BJX_F32) return tiny_dispatch<float>(ctx, which, (int)K, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, batch, flags);
  return tiny_dispatch<double>(ctx, which, (int)K, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, batch, flags);
}
