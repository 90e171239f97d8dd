// bjx_matrix_vjp.hip — SURVEY.md §8(f) f-1 x f-4: reverse-mode pullbacks of the matrix-variate constraint bijectors
//   VecCorrBijector / CorrBijector (corr.jl:64-162), PDBijector / PDVecBijector (pd.jl:1-60) and their inverses,
// batched over samples:  in_bar = J(in)' out_bar + ladj_bar[n] ∇ logabsdetjac(in).
//
// inverse (unconstrained -> matrix; what every leapfrog step of HMC on an LKJ / Wishart / covariance prior differentiates): the
// rules the reference ships, chained in one pass over the sample —
//   X = L L' (pd_from_upper / pd_from_lower, ext/BijectorsChainRulesCoreExt.jl:324-331, ext/BijectorsReverseDiffExt.jl:160-168):
//       L̄ = tril((X̄ + X̄') L)
//   corr kinds: the reverse sweep of _inv_link_chol_lkj (corr.jl:402-451), with (1 - z²)·exp(log_remainder) in place of its
//       (inv(z) - z)·W (the same number, finite at z = 0), exp(log_remainder) recovered as √(Σ_{m>=i} L[c][m]²), plus the
//       (K - j) log U[j,j] terms of corr.jl:77-79, :144-146;
//   PD kinds: replace_diag(exp) (ext/BijectorsReverseDiffExt.jl:153-158) and the weights of pd.jl:27-31.
// forward (matrix -> unconstrained): the factor's cotangent from the link — y = asinh(w/√R), logcosh(y) = ½ log(1 + w²/R) with R the
// running remainder of corr.jl:299-305, atanh on the first row of the vector form (:322); replace_diag(log) and the weights of
// pd.jl:27-31 — then the reverse of cholesky(Hermitian(X)), unblocked, column by column from the last.  The cotangent lands on
// the triangle the reference READS (upper for the correlation bijectors, src/utils.jl:50; lower for PD, :37); the other is zero.
//
// Mapping on gfx950, three ranges of K:
//   K <= 8   ONE LANE per sample, like matrix_lane_kernel: the factor and its cotangent live in the lane's registers (fully
//            unrolled, wave-uniform guards), the primal input and the output cotangent of 64 consecutive samples travel through
//            two [64][odd pitch] LDS tiles with 16-byte global accesses, the input cotangent leaves through the first tile.
//            (The 12-row instantiation stays as the A/B of the next range: BJX_MATRIX_VJP_GRP = 0.)
//   9 ... 64 one GROUP of 16 / 32 / 64 lanes per sample with the factor in LDS: bjx_matrix_vjp_grp.hip (Float64: to 32).
//   beyond   the one-lane code on a lane-strided global workspace (entry e of lane l at ws[e * lanes + l]: coalesced), plain
//            per-lane global accesses for the arrays — correct, not fast (0.4 % of the HBM peak at K = 32 before the group kernel;
//            the LKJ / Wishart blocks of real models are 2x2 ... 8x8).
// Algorithmic bytes per sample: 2 x (unconstrained side) + K² (matrix side: one of in / out_bar is the matrix) + K² when the
// matrix is the input (its cotangent is written) — e.g. inverse(VecCorr): (K(K-1) + K²)·sizeof(T) + ladj_bar.
#include <cstdlib>

#include "bjx_internal.h"
#include "bjx_tile.h"
#include "bjx_matrix_vjp.h"

using namespace bjx;

namespace {

// storage of a K x K triangle per lane: registers (KMAX > 0, every index a compile-time constant after unrolling) or a
// lane-strided slice of a global workspace (KMAX = 0)
template <class T, int KMAX> struct TriStore {
  T a[KMAX * KMAX];
  __device__ __forceinline__ TriStore(T*, int, int64_t) {}
  __device__ __forceinline__ T& at(int i, int j) { return a[i * KMAX + j]; }
};
template <class T> struct TriStore<T, 0> {
  T* p; int K; int64_t st;
  __device__ __forceinline__ TriStore(T* p_, int K_, int64_t st_) : p(p_), K(K_), st(st_) {}
  __device__ __forceinline__ T& at(int i, int j) { return p[(int64_t)(i * K + j) * st]; }
};
// element e of this lane's sample: a row of the LDS tile (KMAX > 0) or the sample's run in global memory (KMAX = 0)
template <class T> struct Elems {
  T* p;
  __device__ __forceinline__ T get(int e) const { return p[e]; }
  __device__ __forceinline__ void put(int e, T v) const { p[e] = v; }
};

#define BJX_KLOOP(i_, lo_) _Pragma("unroll") for (int i_ = (lo_); i_ < (KMAX > 0 ? KMAX : K); ++i_)

// The whole pullback of ONE sample.  `xin`: the primal input (overwritten with the input cotangent, same layout), `g`: the
// output cotangent, `dl`: the log-det cotangent.  L / G: the lower factor and its cotangent (row-major, j <= i used).
template <class T, int KMAX, int KIND, bool INV>
__device__ __forceinline__ void matrix_vjp_sample(const Elems<T> xin, const Elems<T> g, const T dl, const int K, TriStore<T, KMAX>& L, TriStore<T, KMAX>& G) {
  using M = VjpMath<T>;
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  constexpr int KTOP = KMAX > 0 ? KMAX - 1 : 0;             // (KMAX = 0: the descending loops start at K - 1)
  auto free_idx = [&](int c, int i) -> int {                 // position of the free parameter of (factor row c, column i) on the unconstrained side
    if (KIND == MK_VEC_CORR) return c * (c - 1) / 2 + i;     // triu1_to_vec: column c of U, row i < c
    if (KIND == MK_CORR) return c * K + i;                   // Y[i, c]
    if (KIND == MK_PD) return i * K + c;                     // Y[c, i], i <= c
    return c * (c + 1) / 2 + i;                              // triu_to_vec(Y'): column c, row i <= c
  };
  if constexpr (INV) {
    // ---- primal: L from the unconstrained input (corr.jl:345-399; pd.jl:13-16)
    BJX_KLOOP(c, 0) {
      if (c < K) {
        if constexpr (CORR) {
          T E = T(1);                                        // exp(log_remainder): a product of sech(y)
          BJX_KLOOP(i, 0) {
            if (i < c) {
              T z, s2;
              M::tanh_sech2(xin.get(free_idx(c, i)), z, s2);
              L.at(c, i) = z * E;
              E *= M::sqrt(s2);
            }
          }
          L.at(c, c) = E;
        } else {
          BJX_KLOOP(i, 0) {
            if (i <= c) { const T t = xin.get(free_idx(c, i)); L.at(c, i) = i == c ? M::exp(t) : t; }
          }
        }
      }
    }
    // ---- L̄ = tril((X̄ + X̄') L): G[i][j] = Σ_{m >= j} (X̄[i][m] + X̄[m][i]) L[m][j]
    BJX_KLOOP(i, 0) {
      if (i < K) {
        BJX_KLOOP(j, 0) { if (j <= i) G.at(i, j) = T(0); }
        BJX_KLOOP(m, 0) {
          if (m < K) {
            const T s = g.get(m * K + i) + g.get(i * K + m);
            BJX_KLOOP(j, 0) { if (j <= i && j <= m) G.at(i, j) += s * L.at(m, j); }
          }
        }
      }
    }
    // ---- back through the link, written over the input
    BJX_KLOOP(c, 0) {
      if (c < K) {
        if constexpr (CORR) {
          T dlr = L.at(c, c) * G.at(c, c) + (dl + dl) + ((c >= 1 && c <= K - 2) ? T(K - 1 - c) * dl : T(0));
          T rem = L.at(c, c) * L.at(c, c);
          _Pragma("unroll") for (int i = (KMAX > 0 ? KTOP : K - 1); i >= 0; --i) {
            if (i < c) {
              const int e = free_idx(c, i);
              T z, s2;
              M::tanh_sech2(xin.get(e), z, s2);
              const T w = L.at(c, i), gw = G.at(c, i);
              rem += w * w;                                   // exp(log_remainder before entry i)² = Σ_{m >= i} L[c][m]²
              xin.put(e, s2 * M::sqrt(rem) * gw - z * dlr);
              dlr += dl + w * gw;
            }
          }
          if (KIND == MK_CORR) {                              // on and below the diagonal the K x K input is not read: zero cotangent
            BJX_KLOOP(i, 0) { if (i >= c && i < K) xin.put(c * K + i, T(0)); }
          }
        } else {
          BJX_KLOOP(i, 0) {
            if (i < K) {
              if (i < c) xin.put(free_idx(c, i), G.at(c, i));
              else if (i == c) xin.put(free_idx(c, c), G.at(c, c) * L.at(c, c) + dl * T(K + 1 - c));
              else if (KIND == MK_PD) xin.put(i * K + c, T(0));          // Y[c, i], i > c: not read
            }
          }
        }
      }
    }
  } else {
    // ---- primal: A[i][j], j <= i, from the triangle the reference reads, then the right-looking Cholesky of matrix_lane_kernel
    BJX_KLOOP(i, 0) {
      if (i < K) {
        BJX_KLOOP(j, 0) { if (j <= i) L.at(i, j) = CORR ? xin.get(i * K + j) : xin.get(j * K + i); }
      }
    }
    BJX_KLOOP(k, 0) {
      if (k < K) {
        T rs, sq;
        M::pivot(L.at(k, k), rs, sq);
        L.at(k, k) = sq;
        BJX_KLOOP(i, 0) { if (i > k && i < K) L.at(i, k) *= rs; }
        BJX_KLOOP(i, 0) {
          if (i > k && i < K) {
            BJX_KLOOP(j, 0) { if (j > k && j <= i) L.at(i, j) -= L.at(i, k) * L.at(j, k); }
          }
        }
      }
    }
    // ---- cotangent of the factor from the link
    BJX_KLOOP(c, 0) {
      if (c < K) {
        if constexpr (CORR) {
          // row c of L = column c of U: w_i = L[c][i], d = L[c][c]; R_i = d² + Σ_{m > i} w_m² (the remainder BEFORE entry i, corr.jl:300-304).
          // Pass 1, bottom-up like the reference: R_i parked in G[c][i] (sums of squares only: no cancellation for small pivots);
          // pass 2, top-down: the cotangent of w_m needs Σ_{i < m} ∂F/∂R_i.
          {
            T rem = L.at(c, c) * L.at(c, c);
            _Pragma("unroll") for (int i = (KMAX > 0 ? KTOP : K - 1); i >= 0; --i) {
              if (i < c) { G.at(c, i) = rem; rem += L.at(c, i) * L.at(c, i); }
            }
          }
          T gsum = T(0);
          BJX_KLOOP(m, 0) {
            if (m < c) {
              const T w = L.at(c, m), yb = g.get(free_idx(c, m)), wt = T(K - m) * dl;
              if (KIND == MK_VEC_CORR && m == 0) {            // y = atanh(w), logcosh = -½ log(1 - w²): no remainder involved
                G.at(c, m) = (yb + wt * w) * M::rcp(T(1) - w * w);
              } else {
                const T R = G.at(c, m);
                const T S2 = R + w * w;
                const T rS = M::rcp(M::sqrt(S2)), rS2 = M::rcp(S2), rR = M::rcp(R);
                const T t = yb * rS + wt * w * rS2;           // ∂F/∂w_m at fixed remainders
                G.at(c, m) = t + (w + w) * gsum;
                gsum -= T(0.5) * rR * w * t;                  // ∂F/∂R_m
              }
            }
          }
          G.at(c, c) = (L.at(c, c) + L.at(c, c)) * gsum;
        } else {
          const T rd = M::rcp(L.at(c, c));
          BJX_KLOOP(i, 0) {
            if (i < c) G.at(c, i) = g.get(free_idx(c, i));
            else if (i == c) G.at(c, c) = (g.get(free_idx(c, c)) - dl * T(K + 1 - c)) * rd;
          }
        }
      }
    }
    // ---- reverse of the factorisation (A read from one triangle); Ā[i][j] replaces the input entry it was read from
    _Pragma("unroll") for (int j = (KMAX > 0 ? KTOP : K - 1); j >= 0; --j) {
      if (j < K) {
        const T rd = M::rcp(L.at(j, j));
        BJX_KLOOP(i, 0) { if (i > j && i < K) G.at(j, j) -= G.at(i, j) * L.at(i, j) * rd; }
        const T sjj = T(0.5) * G.at(j, j) * rd;
        xin.put(j * K + j, sjj);
        BJX_KLOOP(m, 0) { if (m < j) G.at(j, m) -= (sjj + sjj) * L.at(j, m); }
        BJX_KLOOP(i, 0) {
          if (i > j && i < K) {
            const T sij = G.at(i, j) * rd;
            if (CORR) { xin.put(i * K + j, sij); xin.put(j * K + i, T(0)); }       // A[i][j] was X[j, i]: element (row j, column i)
            else { xin.put(j * K + i, sij); xin.put(i * K + j, T(0)); }            // A[i][j] was X[i, j]
            BJX_KLOOP(m, 0) {
              if (m < j) { G.at(i, m) -= sij * L.at(j, m); G.at(j, m) -= sij * L.at(i, m); }
            }
          }
        }
      }
    }
  }
}

// K <= 12: registers + two wave-private LDS tiles
template <class T, int KMAX, int KIND, bool INV, int V>
__global__ __launch_bounds__(64) void matrix_lane_vjp_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                             T* __restrict__ in_bar, int K, int P_in, int P_out, int64_t batch) {
  extern __shared__ __align__(16) unsigned char smem_[];
  T* tile_in = reinterpret_cast<T*>(smem_);
  T* tile_g = tile_in + (((size_t)64 * P_in + 3) / 4) * 4;
  const int lane = threadIdx.x;
  const int n_in = (int)(INV ? free_len<KIND>(K) : (int64_t)K * K), n_out = (int)(INV ? (int64_t)K * K : free_len<KIND>(K));
  for (int64_t s0 = (int64_t)blockIdx.x * 64; s0 < batch; s0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - s0) < 64 ? (batch - s0) : 64);
    if (n_in > 0) tile_stage_in<T, V>(tile_in, in + s0 * n_in, n_in, P_in, ncols, lane);
    if (n_out > 0) tile_stage_in<T, V>(tile_g, out_bar + s0 * n_out, n_out, P_out, ncols, lane);
    tile_sync();
    if (lane < ncols) {
      TriStore<T, KMAX> L(nullptr, K, 0), G(nullptr, K, 0);
      const T dl = ladj_bar ? ladj_bar[s0 + lane] : T(0);
      matrix_vjp_sample<T, KMAX, KIND, INV>(Elems<T>{tile_in + lane * P_in}, Elems<T>{tile_g + lane * P_out}, dl, K, L, G);
    }
    tile_sync();
    if (n_in > 0) tile_stage_out<T, V>(tile_in, in_bar + s0 * n_in, n_in, P_in, ncols, lane);
    tile_sync();
  }
}

// any K: the same code on a lane-strided global workspace; `in` is copied to `in_bar` first (the sample code works in place)
template <class T, int KIND, bool INV>
__global__ __launch_bounds__(64) void matrix_mem_vjp_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                            T* __restrict__ in_bar, T* __restrict__ ws, int K, int64_t batch) {
  const int64_t lanes = (int64_t)gridDim.x * 64;
  const int64_t me = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t n_in = INV ? free_len<KIND>(K) : (int64_t)K * K, n_out = INV ? (int64_t)K * K : free_len<KIND>(K);
  TriStore<T, 0> L(ws + me, K, lanes), G(ws + (int64_t)K * K * lanes + me, K, lanes);
  for (int64_t s = me; s < batch; s += lanes) {
    T* mine = in_bar + s * n_in;
    if (mine != in + s * n_in) for (int64_t e = 0; e < n_in; ++e) mine[e] = in[s * n_in + e];
    const T dl = ladj_bar ? ladj_bar[s] : T(0);
    matrix_vjp_sample<T, 0, KIND, INV>(Elems<T>{mine}, Elems<T>{const_cast<T*>(out_bar + s * n_out)}, dl, K, L, G);
  }
}

template <class T, int KIND>
int matrix_vjp_impl(bjx_ctx* ctx, const char* who, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  if (batch == 0) return BJX_OK;
  const int64_t KK = K * K, nv = free_len<KIND>(K);
  const int64_t n_in = inverse ? nv : KK, n_out = inverse ? KK : nv;
  if (n_in == 0) return BJX_OK;                                   // VecCorr with K = 1: nothing to differentiate
  if (K > 8) {
    const int rc = bjx_matrix_vjp_grp(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, KIND, inverse, in, out_bar, ladj_bar, in_bar, K, batch);
    if (rc != 1) return rc;                                       // 1 = shape not served by the group kernel
  }
  if (K <= 12) {
    const int P_in = (int)(n_in | 1), P_out = (int)((n_out > 0 ? n_out : 1) | 1);
    const size_t smem = ((((size_t)64 * P_in + 3) / 4) * 4 + (size_t)64 * P_out) * sizeof(T);
    const int64_t tiles = (batch + 63) / 64;
    const int64_t cap = (int64_t)ctx->num_cu * 32;
    const int grid = (int)(tiles < cap ? tiles : cap);
    const bool vec = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
    constexpr int VW = Vec16<T>::N;
    {
      BjxProf prof_(ctx);
#define BJX_MV(KM_, INV_, V_) do { bjx_allow_big_lds(matrix_lane_vjp_kernel<T, KM_, KIND, INV_, V_>, smem); \
  hipLaunchKernelGGL((matrix_lane_vjp_kernel<T, KM_, KIND, INV_, V_>), dim3(grid), dim3(64), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, P_in, P_out, batch); } while (0)
#define BJX_MV_V(KM_, INV_) do { if (vec) BJX_MV(KM_, INV_, VW); else BJX_MV(KM_, INV_, 1); } while (0)
#define BJX_MV_K(INV_) do { if (K <= 4) BJX_MV_V(4, INV_); else if (K <= 8) BJX_MV_V(8, INV_); else BJX_MV_V(12, INV_); } while (0)
      if (inverse) BJX_MV_K(true); else BJX_MV_K(false);
#undef BJX_MV_K
#undef BJX_MV_V
#undef BJX_MV
    }
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  }
  BJX_REQUIRE(ctx, K <= 1024, BJX_ERR_UNSUPPORTED, "%s: K = %lld: the general-size pullback stops at 1024", who, (long long)K);
  // lanes in flight: as many as a 512 MiB workspace holds (two K x K triangles per lane), at most the batch
  const size_t per_lane = (size_t)2 * KK * sizeof(T);
  int64_t blocks = (int64_t)(((size_t)512 << 20) / (per_lane * 64));
  if (blocks < 1) blocks = 1;
  const int64_t need = (batch + 63) / 64;
  if (blocks > need) blocks = need;
  if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
  { int rc = bjx_ensure_big_ws(ctx, (size_t)blocks * 64 * per_lane); if (rc) return rc; }
  {
    BjxProf prof_(ctx);
    if (inverse) hipLaunchKernelGGL((matrix_mem_vjp_kernel<T, KIND, true>), dim3((unsigned)blocks), dim3(64), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, (T*)ctx->big_ws, (int)K, batch);
    else hipLaunchKernelGGL((matrix_mem_vjp_kernel<T, KIND, false>), dim3((unsigned)blocks), dim3(64), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, (T*)ctx->big_ws, (int)K, batch);
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <int KIND>
int matrix_vjp_entry(bjx_ctx* ctx, const char* who, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar,
                     int64_t K, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, K >= 1 && batch >= 0, BJX_ERR_SHAPE, "%s: bad size", who);
  const bool empty = batch == 0 || (KIND == MK_VEC_CORR && K == 1 && inverse);
  BJX_REQUIRE(ctx, empty || (in && out_bar && in_bar) || (KIND == MK_VEC_CORR && K == 1 && in && in_bar), BJX_ERR_ARG, "%s: null pointer", who);
  if (dt == BJX_F32) return matrix_vjp_impl<float, KIND>(ctx, who, inverse, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
  if (dt == BJX_F64) return matrix_vjp_impl<double, KIND>(ctx, who, inverse, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "%s: bad dtype %d", who, (int)dt);
}

}  // namespace

BJX_API int bjx_vec_corr_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  return matrix_vjp_entry<MK_VEC_CORR>(ctx, "bjx_vec_corr_vjp", dt, inverse, in, out_bar, ladj_bar, in_bar, K, batch);
}
BJX_API int bjx_corr_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  return matrix_vjp_entry<MK_CORR>(ctx, "bjx_corr_vjp", dt, inverse, in, out_bar, ladj_bar, in_bar, K, batch);
}
BJX_API int bjx_pd_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  return matrix_vjp_entry<MK_PD>(ctx, "bjx_pd_vjp", dt, inverse, in, out_bar, ladj_bar, in_bar, K, batch);
}
BJX_API int bjx_pd_vec_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  return matrix_vjp_entry<MK_PD_VEC>(ctx, "bjx_pd_vec_vjp", dt, inverse, in, out_bar, ladj_bar, in_bar, K, batch);
}
