// bjx_chain.hip — F1: fused elementwise ComposedFunction chains.
//
// Replaces the reference's one-allocating-pass-per-stage evaluation of
//   with_logabsdet_jacobian(elementwise(exp) ∘ Shift(b) ∘ Scale(a), X)      (SURVEY.md §3.1;
//   src/bijectors/composed.jl:4-25, scale.jl:11-32, shift.jl:14-24, exp_log.jl:5-9,
//   logit.jl:15-30, leaky_relu.jl:25-29, truncated.jl:15-91)
// by ONE streaming pass: 16-byte loads, all stages applied in registers, log-det contributions
// reduced per block (f64: wave shuffle butterfly + LDS), one partial per block, fixed-order final
// reduce.  Algorithmic HBM traffic: read x once + write y once (8 B/elt f32).
//
// Launch geometry (measured, scripts/membench.hip, profiles/r01_membench.txt): on MI355X a
// NON-persistent launch — ONE 16-byte pack per thread, grid = packs/256, non-temporal loads and
// stores — streams 6.5 TB/s, while grid-stride persistent loops reach 4.5-5.2 TB/s: with ~1M short
// blocks dispatched in order, all resident waves sit in one moving ~5 MB window of x and y.
//
// Two kernels:
//   chain_flat_kernel     : flat index over dim*batch, only the global Σ log-det (what the
//                           reference returns for elementwise bijectors, SURVEY.md §8a').
//   chain_colgroup_kernel : G lanes per column, additionally writes the per-sample log-det vector.
#include "bjx_internal.h"

namespace {
using namespace bjx;

template <class T> struct DevOp {
  int kind, plen;      // plen: 0 none, 1 scalar, >1 per-row
  T s0, s1;            // host scalars
  const T* v0;         // device params (scalar if plen==1) or null
  const T* v1;
};
template <class T> struct ChainArgs {
  DevOp<T> ops[BJX_MAX_OPS];
  int n_ops;
};

// ROWMODE: 0 no per-row parameters; 1 the V rows of a pack are contiguous and 16-byte aligned
//          (dim % V == 0): one vector load per parameter (L1/L2-resident table, <= a few KiB);
//          2 rows wrap inside a pack: scalar loads.
template <class T, int V, int ROWMODE>
__device__ __forceinline__ void load_params(const DevOp<T>& op, int64_t r, int64_t dim, T* a, T* b) {
  if (ROWMODE == 0 || op.plen <= 1) {
    T s0 = (op.plen == 1 && op.v0) ? op.v0[0] : op.s0;
    T s1 = (op.plen == 1 && op.v1) ? op.v1[0] : op.s1;
#pragma unroll
    for (int j = 0; j < V; ++j) { a[j] = s0; b[j] = s1; }
    return;
  }
  if constexpr (ROWMODE == 1) {
    Pack<T, V> pa = load_pack<T, V, false>(op.v0 + r);
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = pa.v[j];
    if (op.v1) {
      Pack<T, V> pb = load_pack<T, V, false>(op.v1 + r);
#pragma unroll
      for (int j = 0; j < V; ++j) b[j] = pb.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) b[j] = op.s1;
    }
  } else {
    if (V > 1 && dim >= V) {
      // Rows wrap inside a pack (a flat pack of a column height that is not a multiple of V straddles two columns): TWO element-aligned
      // 16-byte loads per parameter — the rows from r on (clamped to the table's last V) and the table's first V, where a wrapped pack
      // continues — and three selects per element.  (V scalar gathers behind `while (rj >= dim)` ran the mean-field chain — per-row mu
      // and sigma, summed log-det — at 38 % of the HBM peak at 101 / 333 / 1 001 rows against 73 % at 64 / 100 rows.)
      const int64_t rc = r + V <= dim ? r : dim - V;
      const int sh = (int)(r - rc);                                  // element j of the pack is entry sh + j of (tail pack | head pack)
      T ca[2 * V], cb[2 * V];
      {
        const Pack<T, V> t0 = load_pack<T, V, false>(op.v0 + rc), h0 = load_pack<T, V, false>(op.v0);
#pragma unroll
        for (int j = 0; j < V; ++j) { ca[j] = t0.v[j]; ca[V + j] = h0.v[j]; }
      }
      if (op.v1) {
        const Pack<T, V> t1 = load_pack<T, V, false>(op.v1 + rc), h1 = load_pack<T, V, false>(op.v1);
#pragma unroll
        for (int j = 0; j < V; ++j) { cb[j] = t1.v[j]; cb[V + j] = h1.v[j]; }
      }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        T av = ca[j], bv = op.v1 ? cb[j] : op.s1;
#pragma unroll
        for (int q = 1; q < V; ++q) { av = sh == q ? ca[j + q] : av; if (op.v1) bv = sh == q ? cb[j + q] : bv; }
        a[j] = av;
        b[j] = bv;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int64_t rj = r + j;
      while (rj >= dim) rj -= dim;
      a[j] = op.v0[rj];
      b[j] = op.v1 ? op.v1[rj] : op.s1;
    }
  }
}

// log(a / b) for a, b >= 0 at the price of ONE lean logarithm: with a = ma 2^ea, b = mb 2^eb the atanh argument of fdlibm's scheme is
// s = (m - 1) / (m + 1) = (ma - mb) / (ma + mb) for m = ma / mb — no quotient is ever formed; log(m) = 2 atanh(s) = 2 s + s R(s²) with the
// coefficients of f64lean::log (the rounding of s costs 2e-16 |s| absolutely: two orders inside the 1e-6 bar).
__device__ __forceinline__ double lean_log_ratio(double a, double b) {
  double ma = __builtin_amdgcn_frexp_mant(a), mb = __builtin_amdgcn_frexp_mant(b);      // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(a) - __builtin_amdgcn_frexp_exp(b);
  const bool lo = ma < 0.70710678118654752440 * mb;                                       // m in (0.5, 2) -> [sqrt(1/2), sqrt(2))
  ma = lo ? ma + ma : ma;
  e = lo ? e - 1 : e;
  const bool hi = ma > 1.41421356237309504880 * mb;
  ma = hi ? 0.5 * ma : ma;
  e = hi ? e + 1 : e;
  const double s = (ma - mb) * f64lean::rcp(ma + mb);
  const double z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double dk = (double)e;
  double y = __builtin_fma(dk, 6.93147180369123816490e-01, (s + s) + __builtin_fma(s, t2 + t1, dk * 1.90821492927058770002e-10));
  // a = 0: -Inf, b = 0: +Inf (the bounds of the support); anything negative, infinite or NaN: NaN (log of a negative number)
  y = a == 0.0 ? -Num<double>::inf : y;
  y = b == 0.0 ? Num<double>::inf : y;
  const bool bad = !(a >= 0.0) || !(b >= 0.0) || a == Num<double>::inf || b == Num<double>::inf || (a == 0.0 && b == 0.0);
  return bad ? __builtin_nan("") : y;
}

// One stage applied to U packs of V consecutive elements (wave-uniform `switch`); returns the
// data-dependent log-det contribution (Scale's parameter-only term is added by finalize).
#define BJX_FOR_UJ _Pragma("unroll") for (int u = 0; u < U; ++u) _Pragma("unroll") for (int j = 0; j < V; ++j)
// SAMEROW: the U packs of a lane sit at the same rows (the pack stride 256·V is a multiple of dim, or the
// per-sample geometry): the per-row parameters are loaded ONCE instead of U times — each such load is as wide
// as the data pack itself, and two vector-parameter stages tripled the load traffic of a density chain.
// The arithmetic of one stage on U packs; UA = 1: one parameter pack serves all U packs (same rows), UA = U: one per pack.
// LSUM: the caller adds the U log-det terms up (the summed log-det of a flat pass) — a stage may then return their sum in l[0] alone.
template <class T, int V, int U, int UA, bool LSUM = false>
__device__ __forceinline__ void apply_kind(const int kind, Pack<T, V> (&p)[U], const T (&a)[UA][V], const T (&b)[UA][V], T (&l)[U]) {
  using F = Fast<T>;
  switch (kind) {
    case BJX_OP_EXP:  // exp_log.jl:5-6: ladj = sum(x)
      BJX_FOR_UJ { l[u] += p[u].v[j]; p[u].v[j] = F::exp(p[u].v[j]); }       // Float32: v_exp_f32 (the OCML expf is ~15 VALU: a read-only density chain is VALU-bound with it)
      break;
    case BJX_OP_LOG:  // exp_log.jl:8-9: ladj = -sum(log, x)
      BJX_FOR_UJ { T t = F::log(p[u].v[j]); l[u] -= t; p[u].v[j] = t; }
      break;
    case BJX_OP_SHIFT:  // shift.jl:14
      BJX_FOR_UJ p[u].v[j] = a[UA == 1 ? 0 : u][j] + p[u].v[j];
      break;
    case BJX_OP_SCALE:  // scale.jl:13
      BJX_FOR_UJ p[u].v[j] = a[UA == 1 ? 0 : u][j] * p[u].v[j];
      break;
    case BJX_OP_SCALE_INV:  // scale.jl:15-16: Scale(inv(a))
      BJX_FOR_UJ p[u].v[j] = F::rcp(a[UA == 1 ? 0 : u][j]) * p[u].v[j];
      break;
    case BJX_OP_LOGIT:  // logit.jl:15,24.  Float32: hardware log/rcp (the OCML versions make this op VALU-bound at 50 % of the roofline)
      if constexpr (sizeof(T) == 8 && UA == 1) {
        // Float64 with one parameter pack for the lane's U packs (scalars, or the same rows).  A lean logarithm is ~40 Float64 operations
        // (~3.8 reciprocals, scripts/f64math_bench.hip) and this stage was two of them per element (44 % of the roofline, VALU-bound):
        //   value    log((x-a)/(b-x)) as ONE logarithm of the ratio (lean_log_ratio: no quotient formed);
        //   log-det  -log g, g = (x-a)(b-x)/(b-a): the log-det of a column is a SUM over its rows, so the g of the V rows of a pack are
        //            multiplied and ONE logarithm is taken per pack — per U packs when the caller only wants the sum (g <= (b-a)/4 and one of its two factors is >= (b-a)/2: the product
        //            of a pack cannot vanish before a factor does, except past 1e-300 — then the logarithms are taken one by one).
        // Same limits at the bounds: value ±Inf, log-det +Inf; outside the support NaN (a negative g must not cancel another one).
        T iw[V];
#pragma unroll
        for (int j = 0; j < V; ++j) iw[j] = F::rcp(b[0][j] - a[0][j]);
        constexpr int UG = LSUM ? U : 1;                           // packs whose g share one logarithm (all of the lane's when only the sum is wanted)
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += UG) {
          T prod = T(1), gmin = Num<T>::inf;
          T g[UG][V];
#pragma unroll
          for (int uu = 0; uu < UG; ++uu) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
              const T x = p[u0 + uu].v[j];
              const T xa = x - a[0][j], xb = b[0][j] - x;
              g[uu][j] = (xa < xb ? xa : xb) * ((xa < xb ? xb : xa) * iw[j]);       // (the larger factor over b - a is in [1/2, 1]: g vanishes only with x - a or b - x)
              prod *= g[uu][j];
              gmin = g[uu][j] < gmin ? g[uu][j] : gmin;
              p[u0 + uu].v[j] = (T)lean_log_ratio((double)xa, (double)xb);
            }
          }
          T lp = F::log(prod);
          if (prod == T(0) && gmin > T(0)) {                     // underflow of the product, not a zero factor
            lp = T(0);
#pragma unroll
            for (int uu = 0; uu < UG; ++uu) {
#pragma unroll
              for (int j = 0; j < V; ++j) lp += F::log(g[uu][j]);
            }
          }
          lp = gmin < T(0) ? (T)__builtin_nan("") : lp;
          lp = prod != prod ? prod : lp;
          l[u0] -= lp;
        }
        break;
      }
      BJX_FOR_UJ {
        const T x = p[u].v[j];
        const T inv = F::rcp(b[UA == 1 ? 0 : u][j] - a[UA == 1 ? 0 : u][j]);
        const T xa = x - a[UA == 1 ? 0 : u][j], xb = b[UA == 1 ? 0 : u][j] - x;
        l[u] -= F::log(xa * xb * inv);
        p[u].v[j] = F::log(xa * F::rcp(xb));                    // logit((x-a)/(b-a)) = log((x-a)/(b-x)), exact at the bounds (±Inf)
      }
      break;
    case BJX_OP_LOGIT_INV:  // logit.jl:19 ; interface.jl:276-281
      BJX_FOR_UJ {
        const T w = b[UA == 1 ? 0 : u][j] - a[UA == 1 ? 0 : u][j];
        const T x = w * f_logistic(p[u].v[j]) + a[UA == 1 ? 0 : u][j];
        l[u] += F::log((x - a[UA == 1 ? 0 : u][j]) * (b[UA == 1 ? 0 : u][j] - x) * F::rcp(w));
        p[u].v[j] = x;
      }
      break;
    case BJX_OP_LEAKY_RELU:  // leaky_relu.jl:25-29
      BJX_FOR_UJ {
        T J = p[u].v[j] < T(0) ? a[UA == 1 ? 0 : u][j] : T(1);
        l[u] += F::log(d_abs(J));
        p[u].v[j] = J * p[u].v[j];
      }
      break;
    case BJX_OP_TRUNCATED:  // truncated.jl:15-31,51-67
      BJX_FOR_UJ {
        T lo = a[UA == 1 ? 0 : u][j], up = b[UA == 1 ? 0 : u][j];
        T x = d_clamp(p[u].v[j], lo, up);
        bool lb = d_isfinite(lo), ub = d_isfinite(up);
        if (lb && ub) {
          // logit((x-a)/(b-a)) = log((x-a)/(b-x)): exact at the bounds (x = b gives +Inf and x = a gives -Inf like
          // the reference's logit(1) / logit(0); z = (x-a)·rcp(b-a) can round to 1 ± ulp there)
          const T inv = F::rcp(up - lo), xa = x - lo, xb = up - x;
          l[u] -= F::log(xa * xb * inv);
          p[u].v[j] = F::log(xa * F::rcp(xb));
        }
        else if (lb) { T t = F::log(x - lo); l[u] -= t; p[u].v[j] = t; }
        else if (ub) { T t = F::log(up - x); l[u] -= t; p[u].v[j] = t; }
        else p[u].v[j] = x;
      }
      break;
    case BJX_OP_TRUNCATED_INV:  // truncated.jl:33-49,71-91
      BJX_FOR_UJ {
        T lo = a[UA == 1 ? 0 : u][j], up = b[UA == 1 ? 0 : u][j], yv = p[u].v[j];
        bool lb = d_isfinite(lo), ub = d_isfinite(up);
        T x;
        if (lb && ub) {
          // one exp serves both: t = exp(-|y|) ∈ (0, 1];  logistic(y) = 1/(1+t) or t/(1+t);  log1pexp(-|y|) = log1p(t)
          // (LogExpFunctions' saturation thresholds are reproduced by t underflowing to 0 and by 1 + t rounding to 1)
          const T ay = d_abs(yv);
          const T t = F::exp(-ay);
          const T r = F::rcp(T(1) + t);
          l[u] += F::log(up - lo) - ay - T(2) * F::log1p(t);
          x = (up - lo) * (yv < T(0) ? t * r : r) + lo;
        }
        else if (lb) { l[u] += yv; x = F::exp(yv) + lo; }
        else if (ub) { l[u] += yv; x = up - F::exp(yv); }
        else x = yv;
        p[u].v[j] = d_clamp(x, lo, up);
      }
      break;
    case BJX_OP_SIGNFLIP:  // ordered.jl:3
      BJX_FOR_UJ p[u].v[j] = -p[u].v[j];
      break;
    case BJX_OP_STDNORMAL_LOGPDF:  // base density of a TransformedDistribution (transformed_distribution.jl:165-169)
      BJX_FOR_UJ l[u] += (T(-0.5) * p[u].v[j]) * p[u].v[j] - T(0.91893853320467274178);
      break;
    default: break;
  }
}
__device__ __forceinline__ bool kind_has_params(int kind) {
  return kind != BJX_OP_EXP && kind != BJX_OP_LOG && kind != BJX_OP_SIGNFLIP && kind != BJX_OP_STDNORMAL_LOGPDF;
}
// SAMEROW: the U packs of a lane sit at the same rows (the pack stride 256·V is a multiple of dim, or the
// per-sample geometry): the per-row parameters are loaded ONCE instead of U times — each such load is as wide
// as the data pack itself, and two vector-parameter stages tripled the load traffic of a density chain.
template <class T, int V, int U, int ROWMODE, bool SAMEROW = false, bool LSUM = false>
__device__ __forceinline__ void apply_op(const DevOp<T>& op, Pack<T, V> (&p)[U], const int64_t (&r)[U], int64_t dim, T (&l)[U]) {
  const int kind = op.kind;
  if constexpr (ROWMODE == 0 && sizeof(T) == 8) {
    // scalar parameters, Float64: ONE parameter pack for the U packs (UA = 1) — lets the Logit stage take its two-log form
    // (the Float32 instantiations keep the round-2 code path: the C2 headline kernel is not touched)
    T a1[1][V], b1[1][V];
#pragma unroll
    for (int j = 0; j < V; ++j) { a1[0][j] = T(0); b1[0][j] = T(0); }
    if (kind_has_params(kind)) load_params<T, V, ROWMODE>(op, r[0], dim, a1[0], b1[0]);
    apply_kind<T, V, U, 1, LSUM>(kind, p, a1, b1, l);
    return;
  }
  T a[U][V], b[U][V];
  if (kind_has_params(kind)) {
    if constexpr (SAMEROW && U > 1) {
      load_params<T, V, ROWMODE>(op, r[0], dim, a[0], b[0]);
#pragma unroll
      for (int u = 1; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < V; ++j) { a[u][j] = a[0][j]; b[u][j] = b[0][j]; }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) load_params<T, V, ROWMODE>(op, r[u], dim, a[u], b[u]);
    }
  }
  apply_kind<T, V, U, U, LSUM>(kind, p, a, b, l);
}

// per-pack log-det contributions lu[u] (the per-sample kernel reduces each pack's column separately)
// One parameter pack per stage (SAMEROW, or no per-row parameters at all): the descriptor and the parameters of stage
// k + 1 are fetched BEFORE the arithmetic of stage k.  A stage is one dependent chain kernarg s_load -> branch ->
// pointer s_load -> table load -> wait, ~0.2 us that nothing else of the wave covers; on a read-only density chain
// of six stages those chains were a third of the wave's life (PMC: VALU 45 % busy, HBM 37 %, 4 waves/SIMD resident).
template <class T, int V, int U, int ROWMODE, bool LSUM = false>
__device__ __forceinline__ void apply_chain_u_prefetch(const ChainArgs<T>& A, Pack<T, V> (&p)[U], int64_t r0, int64_t dim, T (&lu)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) lu[u] = T(0);
  const int n_ops = A.n_ops;
  if (n_ops <= 0) return;
  int kind = A.ops[0].kind;
  T a[1][V], b[1][V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a[0][j] = T(0); b[0][j] = T(0); }
  if (kind_has_params(kind)) load_params<T, V, ROWMODE>(A.ops[0], r0, dim, a[0], b[0]);
  for (int k = 0; k < n_ops; ++k) {
    int kn = BJX_OP_IDENTITY;
    T an[1][V], bn[1][V];
#pragma unroll
    for (int j = 0; j < V; ++j) { an[0][j] = T(0); bn[0][j] = T(0); }
    if (k + 1 < n_ops) {
      kn = A.ops[k + 1].kind;
      if (kind_has_params(kn)) load_params<T, V, ROWMODE>(A.ops[k + 1], r0, dim, an[0], bn[0]);
    }
    apply_kind<T, V, U, 1, LSUM>(kind, p, a, b, lu);
    kind = kn;
#pragma unroll
    for (int j = 0; j < V; ++j) { a[0][j] = an[0][j]; b[0][j] = bn[0][j]; }
  }
}
template <class T, int V, int U, int ROWMODE, bool SAMEROW = false, bool LSUM = false>
__device__ __forceinline__ void apply_chain_u(const ChainArgs<T>& A, Pack<T, V> (&p)[U], const int64_t (&r)[U], int64_t dim, T (&lu)[U]) {
  // scalar-only chains keep the plain loop: nothing but one s_load to cover, and the look-ahead's copies cost 2-6 % (same-call A/B)
  if constexpr (ROWMODE != 0 && (SAMEROW || U == 1)) {
    apply_chain_u_prefetch<T, V, U, ROWMODE, LSUM>(A, p, r[0], dim, lu);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) lu[u] = T(0);
    for (int k = 0; k < A.n_ops; ++k) apply_op<T, V, U, ROWMODE, SAMEROW, LSUM>(A.ops[k], p, r, dim, lu);
  }
}
template <class T, int V, int U, int ROWMODE, bool SAMEROW = false>
__device__ __forceinline__ T apply_chain(const ChainArgs<T>& A, Pack<T, V> (&p)[U], const int64_t (&r)[U], int64_t dim) {
  T lu[U];
  apply_chain_u<T, V, U, ROWMODE, SAMEROW, true>(A, p, r, dim, lu);
  T l = lu[0];
#pragma unroll
  for (int u = 1; u < U; ++u) l += lu[u];
  return l;
}

// U 16-byte packs per thread (the block owns U consecutive 4 KiB rows), non-persistent grid (see the
// header comment).
// GEN: the input packs are drawn instead of loaded (BJX_INPUT_STDNORMAL): pack i = elements 4i..4i+3 of the global
// stream = one Philox counter (Float32) or half of one (Float64).
template <class T, int V> __device__ __forceinline__ Pack<T, V> gen_pack(uint64_t seed, int64_t e_global) {
  Pack<T, V> p;
  T z[4];
  philox_normal4(seed, e_global >> 2, z);
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int k = (int)((e_global + j) & 3);
    if (j > 0 && k == 0) philox_normal4(seed, (e_global + j) >> 2, z);   // the pack straddles two counters (col0·dim not a multiple of 4)
    p.v[j] = (T)(k == 0 ? z[0] : (k == 1 ? z[1] : (k == 2 ? z[2] : z[3])));
  }
  return p;
}
template <class T, int V, int ROWMODE, bool NT, int U, bool GEN = false>
__global__ __launch_bounds__(256) void chain_flat_kernel(const ChainArgs<T> A, const T* x, T* y, int64_t n,
                                                         int64_t dim, int dim_pow2, const BjxFin fin, uint64_t seed = 0, int64_t e0 = 0) {
  __shared__ double red[4];
  const int64_t nv = n / V;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
  double acc = 0.0;
  if (i0 + (U - 1) * 256 < nv) {
    Pack<T, V> p[U];
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = GEN ? gen_pack<T, V>(seed, e0 + (i0 + u * 256) * V) : load_pack<T, V, NT>(x + (i0 + u * 256) * V);
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = 0;
    if constexpr (ROWMODE != 0) {
      if (dim_pow2) {
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = ((i0 + u * 256) * V) & (dim - 1);
      } else {
        // one 64-bit remainder per lane, the other packs by their fixed distance (a 64-bit `%` is ~100 instructions, U of them per trip)
        r[0] = (i0 * V) % dim;
        const int64_t step = (int64_t)(256 * V) % dim;
#pragma unroll
        for (int u = 1; u < U; ++u) { r[u] = r[u - 1] + step; r[u] = r[u] >= dim ? r[u] - dim : r[u]; }
      }
    }
    T l;
    if (ROWMODE == 1 && U > 1 && (256 * V) % dim == 0) l = apply_chain<T, V, U, ROWMODE, true>(A, p, r, dim);   // same rows in every pack
    else l = apply_chain<T, V, U, ROWMODE>(A, p, r, dim);
    if (y) {
#pragma unroll
      for (int u = 0; u < U; ++u) store_pack<T, V, NT>(y + (i0 + u * 256) * V, p[u]);
    }
    acc = (double)l;
  } else {
    // last block: packs one at a time, then the n % V tail elements by one lane
#pragma unroll 1
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 256;
      if (i < nv) {
        Pack<T, V> p[1];
        p[0] = GEN ? gen_pack<T, V>(seed, e0 + i * V) : load_pack<T, V, NT>(x + i * V);
        int64_t r[1] = {0};
        if constexpr (ROWMODE != 0) r[0] = (i * V) % dim;
        T l = apply_chain<T, V, 1, ROWMODE>(A, p, r, dim);
        if (y) store_pack<T, V, NT>(y + i * V, p[0]);
        acc += (double)l;
      } else if (V > 1 && i == nv) {
        for (int64_t e = nv * V; e < n; ++e) {
          Pack<T, 1> q[1];
          q[0].v[0] = GEN ? gen_pack<T, 1>(seed, e0 + e).v[0] : x[e];
          int64_t r[1] = {ROWMODE == 0 ? 0 : e % dim};
          T l = apply_chain<T, 1, 1, (ROWMODE == 0 ? 0 : 2)>(A, q, r, dim);
          if (y) y[e] = q[0].v[0];
          acc += (double)l;
        }
      }
    }
  }
  block_publish_partial(acc, red, fin);      // small grids: the last block to arrive finishes the sum (one launch per call)
}

// Per-sample log-det with the FLAT geometry (U packs in flight per lane, 16 KiB per block) when a
// column is G = dim/V packs with G a power of two <= 64: the G lanes of an aligned group hold one
// column of each of their U packs, so U butterflies over G lanes give the U per-column sums.
// (chain_colgroup_kernel below keeps one pack per lane in flight for such shapes: 36 % of the HBM
// roofline at dim = 64 against 79 % for the flat kernel, profiles/r01b_rows.md.)
template <class T, int V, int ROWMODE, bool NT, int U, int G>
__global__ __launch_bounds__(256) void chain_flatcol_kernel(const ChainArgs<T> A, const T* x, T* y, T* ladj_ps, int64_t n, int64_t dim,
                                                            double c_ps_host, const double* c_ps_dev, int accumulate, double* partials) {
  __shared__ double red[4];
  const int64_t nv = n / V;                         // dim % V == 0 here: no element tail
  const int64_t i0 = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
  const int gl = threadIdx.x & (G - 1);
  double acc = 0.0;
  Pack<T, V> p[U];
  int64_t r[U];
  const bool full = i0 + (U - 1) * 256 < nv;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = i0 + u * 256;
    if (full || i < nv) p[u] = load_pack<T, V, NT>(x + i * V);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) p[u].v[j] = T(1);   // harmless input for every op; results discarded
    }
    r[u] = (int64_t)gl * V;                         // row of the pack inside its column
  }
  T lu[U];
  apply_chain_u<T, V, U, ROWMODE, true>(A, p, r, dim, lu);      // r[u] = gl·V for every pack
  const double c_ps = c_ps_host + (c_ps_dev ? *c_ps_dev : 0.0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = i0 + u * 256;
    const bool ok = full || i < nv;
    if (ok && y) store_pack<T, V, NT>(y + i * V, p[u]);
    T l;
    if constexpr (sizeof(T) == 4) l = group_sum_f32_dpp<G>(ok ? lu[u] : T(0));     // in-row stages as DPP adds (a ds_bpermute stage is ~6 VALU + an LDS round trip)
    else l = group_sum<G>(ok ? lu[u] : T(0));
    if (ok && gl == 0) {
      const int64_t col = i / G;
      T out = l + (T)c_ps;
      if (accumulate) out += ladj_ps[col];
      ladj_ps[col] = out;
      acc += (double)l;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// Per-sample variant: G consecutive lanes own one column (coalesced because a column is
// contiguous), 256/G columns per block, non-persistent grid.  Writes ladj_ps[col] (+ per-sample
// constant) and the block partial of the sum.
constexpr int CHAIN_U = 4;  // packs in flight per lane when a column needs more than G packs
template <class T, int V, int ROWMODE, bool NT>
__global__ __launch_bounds__(256) void chain_colgroup_kernel(const ChainArgs<T> A, const T* x, T* y, T* ladj_ps,
                                                             int64_t dim, int64_t batch, int G, double c_ps_host,
                                                             const double* c_ps_dev, int accumulate,
                                                             double* partials) {
  __shared__ double red[4];
  const int gl = threadIdx.x & (G - 1);            // lane within the column group
  const int cols_per_block = 256 / G;
  const int64_t col = (int64_t)blockIdx.x * cols_per_block + threadIdx.x / G;
  const int64_t nvc = dim / V;                     // packs per column (dim % V == 0 when V > 1)
  double acc = 0.0;
  T l = T(0);
  if (col < batch) {
    const T* xc = x + col * dim;
    T* yc = y + col * dim;
    int64_t v = gl;
    if (v + (int64_t)(CHAIN_U - 1) * G < nvc) {
      for (; v + (int64_t)(CHAIN_U - 1) * G < nvc; v += (int64_t)G * CHAIN_U) {
        Pack<T, V> p[CHAIN_U];
        int64_t r[CHAIN_U];
#pragma unroll
        for (int u = 0; u < CHAIN_U; ++u) { p[u] = load_pack<T, V, NT>(xc + (v + (int64_t)u * G) * V); r[u] = (v + (int64_t)u * G) * V; }
        l += apply_chain<T, V, CHAIN_U, ROWMODE>(A, p, r, dim);
#pragma unroll
        for (int u = 0; u < CHAIN_U; ++u) if (y) store_pack<T, V, NT>(yc + (v + (int64_t)u * G) * V, p[u]);
      }
    }
    if (v < nvc) {
      // the last (up to CHAIN_U - 1) packs of every lane: requested together, not one round trip each (64 < packs < 4·G per column:
      // dim = 333 ran at 25 % with one pack per lane in flight)
      constexpr int UR = CHAIN_U - 1;
      Pack<T, V> p[UR];
      int64_t r[UR];
      bool ok[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int64_t vu = v + (int64_t)u * G;
        ok[u] = vu < nvc;
        r[u] = ok[u] ? vu * V : 0;
        if (ok[u]) p[u] = load_pack<T, V, NT>(xc + vu * V);
        else {
#pragma unroll
          for (int j = 0; j < V; ++j) p[u].v[j] = T(1);
        }
      }
      T lu[UR];
      apply_chain_u<T, V, UR, ROWMODE, false>(A, p, r, dim, lu);
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        if (ok[u]) {
          l += lu[u];
          if (y) store_pack<T, V, NT>(yc + (v + (int64_t)u * G) * V, p[u]);
        }
      }
    }
    // column heights that are not whole packs (dim = 127, 333, 1001 ...): the packs above are then only ELEMENT-aligned — global
    // accesses take that — and the last dim % V rows go one by one on the lane after the last pack's
    if (V > 1) {
      const int tail = (int)(dim - nvc * V);
      if (tail && gl == (int)(nvc % G)) {
        for (int t = 0; t < tail; ++t) {
          Pack<T, 1> p1[1];
          int64_t r1[1] = {nvc * V + t};
          p1[0].v[0] = xc[r1[0]];
          l += apply_chain<T, 1, 1, (ROWMODE == 0 ? 0 : 2)>(A, p1, r1, dim);
          if (y) yc[r1[0]] = p1[0].v[0];
        }
      }
    }
  }
  l = group_sum_rt(l, G);
  if (col < batch && gl == 0) {
    const double c_ps = c_ps_host + (c_ps_dev ? *c_ps_dev : 0.0);
    T out = l + (T)c_ps;
    if (accumulate) out += ladj_ps[col];
    ladj_ps[col] = out;
    acc = (double)l;
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// Per-sample variant for SHORT columns that are not whole 16-byte packs (dim = 1 ... 13 except 4, 8, 12; same-box A/B: 43-67 % against 20-53 % for the walker, level at 14, behind at 15): lane = column, the column in
// registers, no LDS and no cross-lane sum.  A wave instruction reads 64 consecutive columns = one contiguous run (dim·256 bytes), row
// as one or two multi-dword accesses per lane (element-aligned: dwordx2 / x3 / x4 pieces).  UC columns per thread in flight.  (The mixed walker these
// shapes took before spends ~150 VALU per row on slot decoding and tile addressing: 18-35 % of the HBM peak at dim = 2 ... 5, where
// the whole-pack dim = 4 runs at 63-68 % — profiles/r03_small_sizes.md.  With one 4-byte load per row and lane instead of the
// multi-dword pieces the same kernel ran at 37-43 % for dim <= 3 and BEHIND the walker from dim = 6.)
template <class T, int DIM, int ROWMODE, bool NT, int UC>
__global__ __launch_bounds__(256) void chain_tiny_kernel(const ChainArgs<T> A, const T* __restrict__ x, T* __restrict__ y, T* __restrict__ ladj_ps, int64_t batch,
                                                         double c_ps_host, const double* c_ps_dev, int accumulate, double* partials) {
  __shared__ double red[4];
  const int64_t col0 = (int64_t)blockIdx.x * (256 * UC) + threadIdx.x;
  Pack<T, 1> p[UC][DIM];
#pragma unroll
  for (int k = 0; k < UC; ++k) {
    const int64_t col = col0 + (int64_t)k * 256;
    TinyCol<T, DIM> t{};
    if (col < batch) t = *reinterpret_cast<const TinyCol<T, DIM>*>(x + col * DIM);
#pragma unroll
    for (int u = 0; u < DIM; ++u) p[k][u].v[0] = t.v[u];
  }
  const double c_ps = c_ps_host + (c_ps_dev ? *c_ps_dev : 0.0);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < UC; ++k) {
    const int64_t col = col0 + (int64_t)k * 256;
    int64_t r[DIM];
#pragma unroll
    for (int u = 0; u < DIM; ++u) r[u] = u;
    const T l = apply_chain<T, 1, DIM, ROWMODE>(A, p[k], r, DIM);
    if (col < batch) {
      if (y) {
        TinyCol<T, DIM> t;
#pragma unroll
        for (int u = 0; u < DIM; ++u) t.v[u] = p[k][u].v[0];
        *reinterpret_cast<TinyCol<T, DIM>*>(y + col * DIM) = t;
      }
      T out = l + (T)c_ps;
      if (accumulate) out += ladj_ps[col];
      ladj_ps[col] = out;
      acc += (double)l;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// Per-sample variant for columns of at most G packs (one pack per lane) that are NOT a power of two of them (dim = 100, 200,
// 252 ...): the group kernel above keeps ONE pack per lane in flight for such shapes (33 % of the roofline).  Here a block
// takes U times as many columns and every lane holds the packs of U different columns — the same rows, so the per-row
// parameters are fetched once — before the chain runs: U loads in flight per lane like the flat kernels.
template <class T, int V, int ROWMODE, bool NT, int U>
__global__ __launch_bounds__(256) void chain_colbatch_kernel(const ChainArgs<T> A, const T* x, T* y, T* ladj_ps, int64_t dim, int64_t batch, int G,
                                                             double c_ps_host, const double* c_ps_dev, int accumulate, double* partials) {
  __shared__ double red[4];
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = 256 / G;
  const int64_t col0 = (int64_t)blockIdx.x * cols_per_block * U + threadIdx.x / G;
  const int64_t nvc = dim / V;
  const bool lane_ok = gl < nvc;
  Pack<T, V> p[U];
  int64_t r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t col = col0 + (int64_t)u * cols_per_block;
    r[u] = (int64_t)gl * V;
    if (lane_ok && col < batch) p[u] = load_pack<T, V, NT>(x + col * dim + (int64_t)gl * V);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) p[u].v[j] = T(1);      // harmless input for every op; results discarded
    }
  }
  T lu[U];
  apply_chain_u<T, V, U, ROWMODE, true>(A, p, r, dim, lu);
  // column heights that are not whole packs: the packs above are then only element-aligned, and the last dim % V rows go to the
  // lanes after the last pack's, ONE row each (all U columns at once: the same loads-in-flight as the packs; one lane walking the
  // three tail rows of four columns one after the other held dim = 63 / 127 / 255 at 26 %)
  T ltail[U];
#pragma unroll
  for (int u = 0; u < U; ++u) ltail[u] = T(0);
  if (V > 1) {
    const int tail = (int)(dim - nvc * V);
    if (tail) {                                                        // wave-uniform
      const int tt = (gl - (int)(nvc & (G - 1))) & (G - 1);
      const bool has_tail = tt < tail;
      Pack<T, 1> pt[U];
      int64_t rt[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t col = col0 + (int64_t)u * cols_per_block;
        rt[u] = nvc * V + (has_tail ? tt : 0);
        pt[u].v[0] = (has_tail && col < batch) ? x[col * dim + rt[u]] : T(1);
      }
      T lt[U];
      apply_chain_u<T, 1, U, (ROWMODE == 0 ? 0 : 2), true>(A, pt, rt, dim, lt);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t col = col0 + (int64_t)u * cols_per_block;
        if (has_tail && col < batch) {
          if (y) y[col * dim + rt[u]] = pt[u].v[0];
          ltail[u] = lt[u];
        }
      }
    }
  }
  const double c_ps = c_ps_host + (c_ps_dev ? *c_ps_dev : 0.0);
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t col = col0 + (int64_t)u * cols_per_block;
    const bool ok = lane_ok && col < batch;
    if (ok && y) store_pack<T, V, NT>(y + col * dim + (int64_t)gl * V, p[u]);
    const T l = group_sum_rt((ok ? lu[u] : T(0)) + ltail[u], G);
    if (col < batch && gl == 0) {
      T out = l + (T)c_ps;
      if (accumulate) out += ladj_ps[col];
      ladj_ps[col] = out;
      acc += (double)l;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// Parameter-only log-det terms of Scale ops whose `a` lives on the device (scale.jl:26-32):
//   consts[0] = per-sample constant  Σ_ops ± Σ_i log|a_i|           (scalar a: dim * log|a|)
//   consts[1] = total added to ladj_sum: batch * consts[0], except that with
//               BJX_REF_VECTOR_SCALE_LADJ a vector-a op contributes Σ_i log|a_i| only once.
template <class T>
__global__ __launch_bounds__(256) void chain_consts_kernel(const ChainArgs<T> A, int64_t dim, int64_t batch, int ref_quirk,
                                                           double* consts) {
  __shared__ double red[4];
  double c_ps = 0.0, c_sum = 0.0;
  for (int k = 0; k < A.n_ops; ++k) {
    const DevOp<T>& op = A.ops[k];
    if ((op.kind != BJX_OP_SCALE && op.kind != BJX_OP_SCALE_INV) || !op.v0) continue;
    double s = 0.0;
    for (int i = threadIdx.x; i < op.plen; i += blockDim.x) s += (double)d_log(d_abs(op.v0[i]));
    s = group_sum<64>(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    double sign = op.kind == BJX_OP_SCALE ? 1.0 : -1.0;
    if (op.plen == 1) { c_ps += sign * s * (double)dim; c_sum += sign * s * (double)dim * (double)batch; }
    else { c_ps += sign * s; c_sum += sign * s * (ref_quirk ? 1.0 : (double)batch); }
  }
  if (threadIdx.x == 0) { consts[0] = c_ps; consts[1] = c_sum; }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
bool env_nt() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("BJX_NT"); v = e ? atoi(e) : 1; }   // non-temporal by default (+4 %, membench)
  return v != 0;
}

template <class T>
int chain_impl(bjx_ctx* ctx, const bjx_op* ops, int n_ops, const T* x, T* y, T* ladj_ps, double* ladj_sum,
               int64_t dim, int64_t batch, uint32_t flags) {
  ChainArgs<T> A;
  memset(&A, 0, sizeof(A));
  A.n_ops = n_ops;
  bool any_row = false, any_dev_scale = false;
  double c_ps_host = 0.0, c_sum_host = 0.0;
  const bool quirk = (flags & BJX_REF_VECTOR_SCALE_LADJ) != 0;
  for (int k = 0; k < n_ops; ++k) {
    const bjx_op& o = ops[k];
    DevOp<T>& d = A.ops[k];
    BJX_REQUIRE(ctx, o.kind >= BJX_OP_EXP && o.kind <= BJX_OP_STDNORMAL_LOGPDF, BJX_ERR_ARG, "bjx_chain: op %d has unknown kind %d", k, o.kind);
    BJX_REQUIRE(ctx, o.param_len == 0 || o.param_len == 1 || o.param_len == dim, BJX_ERR_SHAPE,
                "bjx_chain: op %d parameter length %d does not match dim %lld", k, o.param_len, (long long)dim);
    d.kind = o.kind;
    d.plen = o.param_len;
    d.s0 = (T)o.p0;
    d.s1 = (T)o.p1;
    d.v0 = static_cast<const T*>(o.v0);
    d.v1 = static_cast<const T*>(o.v1);
    if (d.plen > 1) {
      BJX_REQUIRE(ctx, d.v0, BJX_ERR_ARG, "bjx_chain: op %d has per-row parameters but v0 == NULL", k);
      any_row = true;
    }
    if (o.kind == BJX_OP_SCALE || o.kind == BJX_OP_SCALE_INV) {
      if (d.v0) any_dev_scale = true;
      else {
        double s = std::log(std::fabs((double)d.s0)) * (o.kind == BJX_OP_SCALE ? 1.0 : -1.0);
        c_ps_host += s * (double)dim;
        c_sum_host += s * (double)dim * (double)batch;
      }
    }
  }
  const int64_t n = dim * batch;
  if (n == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  if (any_dev_scale) {
    hipLaunchKernelGGL(chain_consts_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, A, dim, batch, quirk ? 1 : 0, ctx->consts);
    BJX_CHECK_LAUNCH(ctx);
  }
  constexpr int VW = Vec16<T>::N;
  const bool vec_ok = bjx_aligned16(x) && bjx_aligned16(y);
  // per-row parameter tables are read as 16-byte packs when the rows of a pack are contiguous
  bool rows_vec = vec_ok && dim % VW == 0;
  for (int k = 0; k < n_ops && rows_vec; ++k)
    if (A.ops[k].plen > 1 && (!bjx_aligned16(A.ops[k].v0) || (A.ops[k].v1 && !bjx_aligned16(A.ops[k].v1)))) rows_vec = false;
  const bool nt = env_nt();
  const int dim_pow2 = (dim & (dim - 1)) == 0 ? 1 : 0;
  int64_t grid = 1;
  BjxFin fin;                 // flat kernels: Σ log|det J| epilogue (in-kernel for small grids, bjx_make_fin)
  bool second = false;

  if (flags & BJX_INPUT_STDNORMAL) {
    // on-device sampling: the input packs are drawn inside the kernel (flat geometry; sum-only log-det)
    BJX_REQUIRE(ctx, !ladj_ps, BJX_ERR_UNSUPPORTED, "bjx_chain: BJX_INPUT_STDNORMAL returns the values and, optionally, the summed log-det only");
    BJX_REQUIRE(ctx, y, BJX_ERR_ARG, "bjx_chain: BJX_INPUT_STDNORMAL needs an output buffer");
    constexpr int UG = 2;
    const int64_t e0 = ctx->rng_col0 * dim;
    const bool yv = bjx_aligned16(y);
#define LAUNCH_GEN(V_, RM_)                                                                                          \
  do {                                                                                                               \
    grid = (n / V_ + 1 + 256 * UG - 1) / (256 * UG);                                                                 \
    BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_chain: input too large for one launch");     \
    { int rc_ = bjx_make_fin(ctx, grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, flags, &fin, &second); if (rc_) return rc_; } \
    BjxProf prof_(ctx);                                                                                              \
    hipLaunchKernelGGL((chain_flat_kernel<T, V_, RM_, true, UG, true>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, (const T*)nullptr, y, n, dim, dim_pow2, fin, ctx->rng_seed, e0); \
  } while (0)
    if (!any_row) { if (yv) LAUNCH_GEN(VW, 0); else LAUNCH_GEN(1, 0); }
    else if (yv && rows_vec) LAUNCH_GEN(VW, 1);
    else if (yv) LAUNCH_GEN(VW, 2);
    else LAUNCH_GEN(1, 2);
#undef LAUNCH_GEN
    BJX_CHECK_LAUNCH(ctx);
    if (second) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
    return BJX_OK;
  }

  // Packs per thread (measured on MI355X, profiles/r01_chain_tuning.txt): a wave lives for one memory
  // round trip, so its compute latency must be amortised over enough bytes in flight.  1 light
  // stage: 2 packs (6.3 TB/s vs 5.9 with 1 / 5.7 with 4); anything heavier: 4 packs
  // (C2 6.1 TB/s, C2 with per-row vectors 5.8 vs 5.0 with 2).
  static const int tune_u = 0;
  const int upt = tune_u ? tune_u : ((n_ops <= 1 && !any_row) ? 2 : 4);
#define LAUNCH_FLAT_UV(V_, RM_, U_)                                                                           \
  do {                                                                                                        \
    grid = (n / V_ + 1 + 256 * U_ - 1) / (256 * U_);   /* +1: the lane that owns the n % V tail */            \
    BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_chain: input too large for one launch"); \
    { int rc_ = bjx_make_fin(ctx, grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, flags, &fin, &second); if (rc_) return rc_; } \
    BjxProf prof_(ctx);                                                                                       \
    if (nt) hipLaunchKernelGGL((chain_flat_kernel<T, V_, RM_, true, U_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, n, dim, dim_pow2, fin); \
    else hipLaunchKernelGGL((chain_flat_kernel<T, V_, RM_, false, U_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, n, dim, dim_pow2, fin);  \
  } while (0)
#define LAUNCH_FLAT(V_, RM_) LAUNCH_FLAT_UV(V_, RM_, 1)
#define LAUNCH_FLAT_TUNED(V_, RM_)                                  \
  do {                                                              \
    if (upt == 1) LAUNCH_FLAT_UV(V_, RM_, 1);                       \
    else if (upt == 2) LAUNCH_FLAT_UV(V_, RM_, 2);                  \
    else LAUNCH_FLAT_UV(V_, RM_, 4);                                \
  } while (0)

  if (!ladj_ps) {
    if (!any_row) { if (vec_ok) LAUNCH_FLAT_TUNED(VW, 0); else LAUNCH_FLAT(1, 0); }
    else if (rows_vec) LAUNCH_FLAT_TUNED(VW, 1);
    else if (vec_ok) LAUNCH_FLAT_TUNED(VW, 2);      // (one pack per thread until round 4: 38 % of the HBM peak at odd heights, with load_params' wrapped form and four packs: see DESIGN)
    else LAUNCH_FLAT(1, 2);
  } else {
    // per-sample: G lanes per column
    const bool v_ok = vec_ok && dim % VW == 0;
    // Columns that are not a whole number of 16-byte packs (dim = 3, 10, 13, ...: most real parameter vectors) would fall to
    // 4-byte accesses here (dim = 10: 12 % of the roofline).  The column walker of bjx_stacked_mixed moves 64 consecutive
    // columns as ONE contiguous run of 16-byte packs whatever the column height and gives every column to a lane, so the
    // per-sample log-det needs no cross-lane sum either: the chain goes there as a single elementwise segment.
    // The same holds for columns that ARE whole packs but not a power-of-two number of them (dim = 24, 48, 100, 200 ...): the
    // group kernel below keeps one pack per lane in flight there (33 % of the roofline; the walker: 63-67 %).
    static const int use_tiny = env_int("BJX_CHAIN_TINY", 1);
    // dim <= 7: the one-segment Stacked route below reaches stacked_tiny_kernel, which is ahead there (68-73 % against 48-67 %); this
    // kernel then serves what that route does not take (in-place calls, chains with more than two nonlinear stages) and dim = 9 ... 13
    bool via_stacked = false;
    if (dim <= 7 && (!v_ok || dim * sizeof(T) <= 16) && y && (const void*)x != (const void*)y && n_ops <= BJX_MAX_SEG_OPS && (flags & ~(uint32_t)BJX_ACCUMULATE) == 0 && env_int("BJX_CHAIN_WALKER", 1)) {
      via_stacked = true;
      for (int k = 0; k < n_ops; ++k) via_stacked = via_stacked && ops[k].kind >= BJX_OP_EXP && ops[k].kind <= BJX_OP_IDENTITY;
    }
    if (via_stacked) {
      bjx_segment sg;
      memset(&sg, 0, sizeof(sg));
      sg.in_lo = 0; sg.out_lo = 0; sg.len = dim; sg.n_ops = n_ops;
      for (int k = 0; k < n_ops; ++k) sg.ops[k] = ops[k];
      const int rc_w = bjx_stacked_mixed(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, &sg, 1, nullptr, 0, x, dim, y, dim, ladj_ps, ladj_sum, batch, flags);
      if (rc_w != BJX_ERR_UNSUPPORTED) return rc_w;
    }
    if (use_tiny && !v_ok && dim >= 1 && dim <= 13 && x && (flags & ~(uint32_t)BJX_ACCUMULATE) == 0) {
      // short columns that are not whole packs: lane = column, the column in registers (chain_tiny_kernel)
      const int UCt = dim <= 7 ? 4 : 2;
      const double* cdev_t = any_dev_scale ? ctx->consts : nullptr;
      const int accum_t = (flags & BJX_ACCUMULATE) ? 1 : 0;
      grid = (batch + 256 * UCt - 1) / (256 * UCt);
      BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_chain: input too large for one launch");
      if (ladj_sum) { int rc_ = bjx_ensure_partials(ctx, (size_t)grid); if (rc_) return rc_; }
      double* partials_t = ladj_sum ? ctx->partials : nullptr;
#define LAUNCH_TINY2(D_, RM_, UC_) hipLaunchKernelGGL((chain_tiny_kernel<T, D_, RM_, false, UC_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, ladj_ps, batch, c_ps_host, cdev_t, accum_t, partials_t)
#define LAUNCH_TINY(D_, UC_) do { if (any_row) LAUNCH_TINY2(D_, 2, UC_); else LAUNCH_TINY2(D_, 0, UC_); } while (0)
      bool launched = true;
      {
        BjxProf prof_(ctx);
        switch ((int)dim) {
          case 1: LAUNCH_TINY(1, 4); break;
          case 2: LAUNCH_TINY(2, 4); break;
          case 3: LAUNCH_TINY(3, 4); break;
          case 5: LAUNCH_TINY(5, 4); break;
          case 6: LAUNCH_TINY(6, 4); break;
          case 7: LAUNCH_TINY(7, 4); break;
          case 9: LAUNCH_TINY(9, 2); break;
          case 10: LAUNCH_TINY(10, 2); break;
          case 11: LAUNCH_TINY(11, 2); break;
          case 13: LAUNCH_TINY(13, 2); break;
          default: launched = false; break;      // Float64 even heights are whole packs (v_ok) and never get here; kept for safety
        }
      }
#undef LAUNCH_TINY
#undef LAUNCH_TINY2
      if (launched) {
        BJX_CHECK_LAUNCH(ctx);
        if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
        return BJX_OK;
      }
    }
    // Per-row parameters (Shift(mu), Scale(sigma) with vectors: the mean-field family) with a per-sample log-det on columns that
    // are not whole packs, or taller than 64 packs: as ONE segment of `Stacked`, whose kernels keep a row's parameters in an LDS table
    // shared by the columns in flight (row slabs on tall columns).  Same call, 256 MiB of input, this route against the group kernels
    // below (which fetch mu and sigma again for every pack): 101 rows 56 / 50 %, 333 rows 49 / 28 %, 1 001 rows 42 / 29 %, 1 000 rows
    // 44 / 41 %; whole-pack heights up to 64 packs stay here (64 / 100 rows: 67 / 75 %).
    if (any_row && y && n_ops <= BJX_MAX_SEG_OPS && (flags & ~(uint32_t)BJX_ACCUMULATE) == 0 && ((!v_ok && dim >= 96) || dim / VW > 64)) {
      // (a row of the Stacked table holds two nonlinear stages: chains with at most two non-affine stages and no Truncated stage,
      //  whose slot rules are its own — everything else stays on the kernels below)
      bool plain = true;
      int nonlin = 0;
      for (int k = 0; k < n_ops; ++k) {
        const int kd = ops[k].kind;
        plain = plain && kd >= BJX_OP_EXP && kd <= BJX_OP_IDENTITY && kd != BJX_OP_TRUNCATED && kd != BJX_OP_TRUNCATED_INV;
        if (!(kd == BJX_OP_SHIFT || kd == BJX_OP_SCALE || kd == BJX_OP_SCALE_INV || kd == BJX_OP_SIGNFLIP || kd == BJX_OP_IDENTITY)) ++nonlin;
      }
      if (plain && nonlin <= 2) {
        bjx_segment sg;
        memset(&sg, 0, sizeof(sg));
        sg.in_lo = 0; sg.out_lo = 0; sg.len = dim; sg.n_ops = n_ops;
        for (int k = 0; k < n_ops; ++k) sg.ops[k] = ops[k];
        return bjx_stacked(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, &sg, 1, x, y, ladj_ps, ladj_sum, dim, batch, flags);
      }
    }
    static const int use_unal = env_int("BJX_CHAIN_UNALIGNED", 1);
    static const int unal_min = 48;
    // (same-box A/B, 2^22 columns: 63 ... 257 rows 56-67 % against 12-46 %; 1001 / 2049 rows 56 / 64 % against 41 / 54 %; between 65
    //  and ~250 packs per column the lanes of a 64-lane group hold one to three packs each and the group kernel runs its three-pack
    //  remainder for all of them: 26-47 % against 37-52 % for the 4-byte path, which keeps those heights)
    if (use_unal && !v_ok && dim >= unal_min && (dim / VW <= 64 || dim / VW >= 250) && x && (flags & ~(uint32_t)BJX_ACCUMULATE) == 0) {
      // taller columns that are not whole packs: G lanes per column with ELEMENT-aligned 16-byte packs + a tail (chain_colgroup_kernel).
      // The tile walker below loses its occupancy with the height (dim = 127: 24 %, 255: 12 %, 1001 on 4-byte accesses: 39 %).
      const int64_t packs_u = dim / VW;
      int Gu = 1;
      while (Gu < 64 && Gu < packs_u) Gu <<= 1;
      const double* cdev_u = any_dev_scale ? ctx->consts : nullptr;
      const int accum_u = (flags & BJX_ACCUMULATE) ? 1 : 0;
      const int cpb_u = 256 / Gu;
      grid = (batch + cpb_u - 1) / cpb_u;
      BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_chain: input too large for one launch");
      if (ladj_sum) { int rc_ = bjx_ensure_partials(ctx, (size_t)grid); if (rc_) return rc_; }
      double* partials_u = ladj_sum ? ctx->partials : nullptr;
      if (packs_u <= 64) {                                             // one pack per lane: four columns in flight per lane (chain_colbatch_kernel)
        constexpr int UBu = 4;
        grid = (batch + (int64_t)cpb_u * UBu - 1) / ((int64_t)cpb_u * UBu);
        if (ladj_sum) { int rc_ = bjx_ensure_partials(ctx, (size_t)grid); if (rc_) return rc_; }
        partials_u = ladj_sum ? ctx->partials : nullptr;
        BjxProf prof_(ctx);
#define LAUNCH_CBU(RM_) hipLaunchKernelGGL((chain_colbatch_kernel<T, VW, RM_, false, UBu>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, ladj_ps, dim, batch, Gu, c_ps_host, cdev_u, accum_u, partials_u)
        if (!any_row) LAUNCH_CBU(0);
        else LAUNCH_CBU(2);
#undef LAUNCH_CBU
      } else {
        BjxProf prof_(ctx);
#define LAUNCH_CGU(RM_) hipLaunchKernelGGL((chain_colgroup_kernel<T, VW, RM_, false>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, ladj_ps, dim, batch, Gu, c_ps_host, cdev_u, accum_u, partials_u)
        if (!any_row) LAUNCH_CGU(0);
        else LAUNCH_CGU(2);
#undef LAUNCH_CGU
      }
      BJX_CHECK_LAUNCH(ctx);
      if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
      return BJX_OK;
    }
    static const int use_walker = env_int("BJX_CHAIN_WALKER", 1);
    bool pow2_packs = false;
    if (v_ok) { const int64_t pk = dim / VW; pow2_packs = pk <= 64 && (pk & (pk - 1)) == 0; }
    static const int walker_max = 32;      // whole-pack columns taller than this: chain_colbatch_kernel (the walker's tile costs occupancy)
    if (use_walker && !pow2_packs && (!v_ok || dim <= walker_max) && y && (const void*)x != (const void*)y && (flags & ~(uint32_t)BJX_ACCUMULATE) == 0 && n_ops <= BJX_MAX_SEG_OPS && dim >= 1) {
      bool plain = true;
      for (int k = 0; k < n_ops; ++k) plain = plain && ops[k].kind >= BJX_OP_EXP && ops[k].kind <= BJX_OP_IDENTITY;
      if (plain) {
        bjx_segment sg;
        memset(&sg, 0, sizeof(sg));
        sg.in_lo = 0; sg.out_lo = 0; sg.len = dim; sg.n_ops = n_ops;
        for (int k = 0; k < n_ops; ++k) sg.ops[k] = ops[k];
        const int rc_w = bjx_stacked_mixed(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, &sg, 1, nullptr, 0, x, dim, y, dim, ladj_ps, ladj_sum, batch, flags);
        if (rc_w != BJX_ERR_UNSUPPORTED) return rc_w;              // too tall for the tile / more than two nonlinear stages: the group kernels below
      }
    }
    const int64_t packs = v_ok ? dim / VW : dim;
    int G = 1;
    while (G < 64 && G < packs) G <<= 1;
    const double* cdev = any_dev_scale ? ctx->consts : nullptr;
    const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
    // flat geometry with per-column butterflies: packs per column a power of two <= 64, parameters per row
    // either absent or readable as aligned packs
    static const int use_flatcol = env_int("BJX_CHAIN_FLATCOL", 1);
    if (use_flatcol && v_ok && packs == G && (!any_row || rows_vec)) {
      constexpr int UF = 4;
      grid = (n / VW + 256 * UF - 1) / (256 * UF);
      BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_chain: input too large for one launch");
      if (ladj_sum) { int rc_ = bjx_ensure_partials(ctx, (size_t)grid); if (rc_) return rc_; }
      double* partials = ladj_sum ? ctx->partials : nullptr;
#define LAUNCH_FC2(RM_, NT_, G_) hipLaunchKernelGGL((chain_flatcol_kernel<T, VW, RM_, NT_, UF, G_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, ladj_ps, n, dim, c_ps_host, cdev, accum, partials)
#define LAUNCH_FC1(RM_, NT_) switch (G) { case 1: LAUNCH_FC2(RM_, NT_, 1); break; case 2: LAUNCH_FC2(RM_, NT_, 2); break; case 4: LAUNCH_FC2(RM_, NT_, 4); break; case 8: LAUNCH_FC2(RM_, NT_, 8); break; \
                                           case 16: LAUNCH_FC2(RM_, NT_, 16); break; case 32: LAUNCH_FC2(RM_, NT_, 32); break; default: LAUNCH_FC2(RM_, NT_, 64); break; }
      {
        BjxProf prof_(ctx);
        if (!any_row) { if (nt) { LAUNCH_FC1(0, true) } else { LAUNCH_FC1(0, false) } }
        else { if (nt) { LAUNCH_FC1(1, true) } else { LAUNCH_FC1(1, false) } }
      }
#undef LAUNCH_FC1
#undef LAUNCH_FC2
      BJX_CHECK_LAUNCH(ctx);
      if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
      return BJX_OK;
    }
    const int cols_per_block = 256 / G;
    static const int use_colbatch = env_int("BJX_CHAIN_COLBATCH", 1);
    if (use_colbatch && v_ok && packs <= G && (!any_row || rows_vec)) {
      // one pack per lane and column, not a power of two of them: four columns in flight per lane (chain_colbatch_kernel)
      constexpr int UB = 4;
      grid = (batch + (int64_t)cols_per_block * UB - 1) / ((int64_t)cols_per_block * UB);
      BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_chain: input too large for one launch");
      if (ladj_sum) { int rc_ = bjx_ensure_partials(ctx, (size_t)grid); if (rc_) return rc_; }
      double* partials = ladj_sum ? ctx->partials : nullptr;
#define LAUNCH_CB(RM_, NT_) hipLaunchKernelGGL((chain_colbatch_kernel<T, VW, RM_, NT_, UB>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, ladj_ps, dim, batch, G, c_ps_host, cdev, accum, partials)
      {
        BjxProf prof_(ctx);
        if (!any_row) { if (nt) LAUNCH_CB(0, true); else LAUNCH_CB(0, false); }
        else { if (nt) LAUNCH_CB(1, true); else LAUNCH_CB(1, false); }
      }
#undef LAUNCH_CB
      BJX_CHECK_LAUNCH(ctx);
      if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
      return BJX_OK;
    }
    grid = (batch + cols_per_block - 1) / cols_per_block;
    BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_chain: input too large for one launch");
    if (ladj_sum) { int rc_ = bjx_ensure_partials(ctx, (size_t)grid); if (rc_) return rc_; }
    double* partials = ladj_sum ? ctx->partials : nullptr;
#define LAUNCH_COL(V_, RM_)                                                                                                 \
  do {                                                                                                                      \
    BjxProf prof_(ctx);                                                                                                     \
    if (nt) hipLaunchKernelGGL((chain_colgroup_kernel<T, V_, RM_, true>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, ladj_ps, dim, batch, G, c_ps_host, cdev, accum, partials);  \
    else hipLaunchKernelGGL((chain_colgroup_kernel<T, V_, RM_, false>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, x, y, ladj_ps, dim, batch, G, c_ps_host, cdev, accum, partials);   \
  } while (0)
    if (!any_row) { if (v_ok) LAUNCH_COL(VW, 0); else LAUNCH_COL(1, 0); }
    else if (v_ok && rows_vec) LAUNCH_COL(VW, 1);
    else if (v_ok) LAUNCH_COL(VW, 2);
    else LAUNCH_COL(1, 2);
#undef LAUNCH_COL
  }
#undef LAUNCH_FLAT
#undef LAUNCH_FLAT_UV
#undef LAUNCH_FLAT_TUNED
  BJX_CHECK_LAUNCH(ctx);
  if (!ladj_ps) {             // flat kernels: bjx_make_fin decided who finishes the sum
    if (second) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
    return BJX_OK;
  }
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_chain(bjx_ctx* ctx, bjx_dtype dt, const bjx_op* ops, int n_ops, const void* x, void* y,
                      void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, n_ops >= 0 && n_ops <= BJX_MAX_OPS && (ops || n_ops == 0), BJX_ERR_ARG, "bjx_chain: n_ops must be in [0, %d]", BJX_MAX_OPS);
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0, BJX_ERR_SHAPE, "bjx_chain: negative size");
  // y == NULL: only the log-det (and, with the density op, logpdf) is wanted — the values are not stored
  BJX_REQUIRE(ctx, ((x || (flags & BJX_INPUT_STDNORMAL)) && (y || ladj_ps || ladj_sum)) || dim * batch == 0, BJX_ERR_ARG, "bjx_chain: null data pointer");
  bjx_op ext[BJX_MAX_OPS];
  if (flags & BJX_BASE_STDNORMAL) {
    BJX_REQUIRE(ctx, n_ops < BJX_MAX_OPS, BJX_ERR_UNSUPPORTED, "bjx_chain: BJX_BASE_STDNORMAL needs a free op slot (n_ops < %d)", BJX_MAX_OPS);
    for (int k = 0; k < n_ops; ++k) ext[k] = ops[k];
    memset(&ext[n_ops], 0, sizeof(bjx_op));
    ext[n_ops].kind = BJX_OP_STDNORMAL_LOGPDF;
    ops = ext;
    ++n_ops;
  }
  if (dt == BJX_F32) return chain_impl<float>(ctx, ops, n_ops, (const float*)x, (float*)y, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return chain_impl<double>(ctx, ops, n_ops, (const double*)x, (double*)y, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_chain: bad dtype %d", (int)dt);
}
