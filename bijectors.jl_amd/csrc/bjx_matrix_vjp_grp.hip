// bjx_matrix_vjp_grp.hip — pullbacks of VecCorrBijector / CorrBijector / PDBijector / PDVecBijector and their inverses
// (SURVEY.md §8(f) f-1 x f-4; the rules and their reference lines: bjx_matrix_vjp.hip) for 8 < K <= 64 (Float64: <= 32).
//
// bjx_matrix_vjp.hip gives a sample to ONE lane; past 12 rows its triangles no longer fit the lane's registers and the same code
// ran on a lane-strided global workspace: 0.4 % of the HBM peak at K = 32 (and 19-27 % at K = 12, on 150 registers of triangle).  Here a GROUP of GS = 16 / 32 / 64 lanes owns a sample
// (4 / 2 / 1 samples per wave, the group never leaves its wave: LDS traffic is ordered by the wave's queue, no block barrier), the
// factor L and a second K x K buffer B live in LDS (rows on 16-byte boundaries: see GrpLds), and every phase is either "lane = row,
// sequential along the row", "lane = column with 16-byte broadcast reads of a row of L", or elementwise over the triangle:
//
// inverse (unconstrained y -> X = L L'; what a leapfrog step differentiates):
//   I1  y -> B (coalesced)                                I2  tanh / sech of every entry (rows p and K - p together fill the group),
//       then lane c builds row c of L from them (the LKJ sweep) / replace_diag(exp); z = tanh(y) stays parked in the dead upper
//       triangle of L for the way back
//   I3  X̄ -> B                                            I4  lane i: row i of L̄ = tril((X̄ + X̄') L) in registers
//   I5  lane c: reverse sweep of its row, with sech² · E recovered as Σ_{m>i} L[c][m]² / √(Σ_{m>=i} L[c][m]²) (sums of
//       squares only — the same number as (1 - z²) exp(log_remainder), no second tanh) -> B      I6  B -> in_bar (coalesced)
// forward (X -> y):
//   F1  X -> B     F2  right-looking Cholesky, lane i keeps row i in registers, column k published for the broadcast reads
//   F3  ȳ -> B     F4  lane c: cotangent of row c of the factor from the link (two sweeps along the row) -> B as L̄
//   F5-F8  the reverse of the factorisation in its level-3 form:  S = L⁻ᵀ Φ(LᵀL̄) L⁻¹ (Φ: lower triangle, half the diagonal),
//       Ā[i][j] = S[i][j] + S[j][i] on the triangle the reference reads, S[j][j] on the diagonal — lane = column of LᵀL̄ and of
//       the first triangular solve, a transpose through B, lane = row of the second solve.  (Checked against the unblocked
//       column-by-column reverse of oracle.py::_chol_reverse: 2e-16.)
//   F9  Ā -> B in the input's layout -> in_bar (coalesced)
// Algorithmic bytes per sample as in bjx_matrix_vjp.hip.
#include <cstdlib>

#include "bjx_internal.h"
#include "bjx_tile.h"
#include "bjx_matrix_vjp.h"

using namespace bjx;

namespace {

#define GRP_UNROLL _Pragma("unroll")
// Left alone, the compiler SINKS the whole FMA chain of a fully unrolled phase below all of its LDS reads (nothing uses the
// accumulators until the next phase) and hoists the reads as far as the register file lets it: 512 registers and scratch.  A fence
// every GRP_FENCE_EVERY rows of L pins the accumulators (an empty asm that "modifies" each of them: the FMAs before it must be
// done) and stops the scheduler there; a few rows stay in flight.
#ifndef GRP_FENCE_EVERY
#define GRP_FENCE_EVERY 2
#endif
template <class T, int KMAX> __device__ __forceinline__ void grp_pin(T (&v)[KMAX]) {
  GRP_UNROLL for (int j = 0; j < KMAX; ++j) asm volatile("" : "+v"(v[j]));
  __builtin_amdgcn_sched_barrier(0);
}
// phase boundary: the wave's LDS queue is in order (the group never leaves its wave); pin the compiler and the scheduler
__device__ __forceinline__ void grp_sync() { tile_sync(); __builtin_amdgcn_sched_barrier(0); }
#define GRP_FENCE4(i_) do { if (((i_) % 4) == 3) __builtin_amdgcn_sched_barrier(0); } while (0)
#define GRP_FENCE(i_, arr_) do { if (((i_) % GRP_FENCE_EVERY) == GRP_FENCE_EVERY - 1) grp_pin(arr_); } while (0)

// 16-byte broadcast reads of a row of L: every lane of the group reads the same address
template <class T> struct Row16;
template <> struct Row16<float> { static constexpr int N = 4; typedef float V __attribute__((ext_vector_type(4))); };
template <> struct Row16<double> { static constexpr int N = 2; typedef double V __attribute__((ext_vector_type(2))); };

// LDS of one sample, in elements: L [KMAX][P] | B [KMAX][P] | column vector [KMAX] | bank padding.  P = KMAX + one 16-byte pack:
// rows start on 16-byte boundaries (broadcast reads of row segments as one ds_read_b128), lane = row accesses are conflict-free at
// KMAX = 16 and two-way at 32.  The padding puts the groups of a wave (Float32) / of a half-wave (Float64) GS banks apart.
template <class T, int GS, int KMAX> struct GrpLds {
  static constexpr int N = Row16<T>::N;
  static constexpr int P = KMAX + N;
  static constexpr int BASE = 2 * KMAX * P + KMAX;
  static constexpr int W = sizeof(T) / 4;
  static constexpr int TARGET = sizeof(T) == 4 ? GS % 64 : (GS >= 32 ? 0 : 32);
  static constexpr int pad() { int q = 0; while (((BASE + q) * W) % 64 != TARGET) q += N; return q; }
  static constexpr int SS = BASE + pad();
};

template <class T, int GS, int KMAX, int KIND, bool INV>
__global__ __launch_bounds__(256) void matrix_grp_vjp_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                            T* __restrict__ in_bar, int K, int64_t batch) {
  using M = VjpMath<T>;
  using RV = typename Row16<T>::V;
  constexpr int N = Row16<T>::N, P = GrpLds<T, GS, KMAX>::P, SPB = 256 / GS, SS = GrpLds<T, GS, KMAX>::SS;
  constexpr int NIT = (KMAX * KMAX + GS - 1) / GS;           // staging rounds of the group over a K x K array
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  constexpr bool VECK = KIND == MK_VEC_CORR || KIND == MK_PD_VEC;
  extern __shared__ __align__(16) unsigned char smem_[];
  // tl: my lane in the group (staging: every lane moves its own elements).  t: the row / column I compute — lanes past KMAX (a
  // 24-row problem on a 32-lane group) repeat the last one: same values to the same addresses.
  const int tl = threadIdx.x & (GS - 1), sl = threadIdx.x / GS;
  const int t = tl < KMAX ? tl : KMAX - 1;
  T* Lb = reinterpret_cast<T*>(smem_) + (size_t)sl * SS;
  T* B = Lb + KMAX * P;
  T* colv = B + KMAX * P;
  const int64_t s_raw = (int64_t)blockIdx.x * SPB + sl;
  const bool live = s_raw < batch;                       // uniform over the group; a dead group computes on the last sample and stores nothing
  const int64_t s = live ? s_raw : batch - 1;
  const int64_t KK = (int64_t)K * K, nfree = free_len<KIND>(K);
  const T dl = ladj_bar ? ladj_bar[s] : T(0);
  const bool act = t < K;
  const int gbase = (threadIdx.x & 63) & ~(GS - 1);      // first lane of my group inside the wave

  // global <-> B.  Every load of a sample is issued before the first LDS store (at most KMAX per lane: a rolled loop waits for each
  // load in turn — 32 round trips to HBM per phase; dead slots read element 0); consecutive lanes on consecutive addresses.
  auto stage_lin = [&](const T* src, int64_t n) {
    T v[NIT];
    GRP_UNROLL for (int it = 0; it < NIT; ++it) { const int e = tl + it * GS; v[it] = src[e < n ? e : 0]; }
    GRP_UNROLL for (int it = 0; it < NIT; ++it) { const int e = tl + it * GS; if (e < n) B[e] = v[it]; }
  };
  auto unstage_lin = [&](T* dst, int64_t n) {
    GRP_UNROLL for (int it = 0; it < NIT; ++it) { const int e = tl + it * GS; if (e < n) dst[e] = B[e]; }
  };
  auto stage_mat = [&](const T* src) {                   // K x K row-major -> pitch P
    T v[NIT];
    GRP_UNROLL for (int it = 0; it < NIT; ++it) { const int e = tl + it * GS; v[it] = src[e < KK ? e : 0]; }
    int r = 0, c = tl;
    GRP_UNROLL for (int it = 0; it < NIT; ++it) {
      while (c >= K) { c -= K; ++r; }
      if (r < K) B[r * P + c] = v[it];
      c += GS;
    }
  };
  auto unstage_mat = [&](T* dst) {
    int r = 0, c = tl;
    GRP_UNROLL for (int it = 0; it < NIT; ++it) {
      while (c >= K) { c -= K; ++r; }
      if (r < K) dst[r * K + c] = B[r * P + c];
      c += GS;
    }
  };
  // where the free parameter of (factor row c, column i) sits in B (the unconstrained side staged as above); in bounds for any c, i < KMAX
  auto pos = [&](int c, int i) -> int {
    if (KIND == MK_VEC_CORR) return c * (c - 1) / 2 + i;
    if (KIND == MK_PD_VEC) return c * (c + 1) / 2 + i;
    if (KIND == MK_CORR) return c * P + i;               // memory index c K + i
    return i * P + c;                                    // MK_PD: memory index i K + c
  };
  // Style of everything below: NO branch around an update of a register array or of a loop-carried scalar — selects only, branches
  // around plain stores.  (With `if (k < K) { ... a[j] -= ... }` the compiler merged the arrays at every join: 3 700 v_mov_b64,
  // 1 200 AGPR moves and 1 100 lane spills in the 16 600 instructions of the forward kernel.)  The problem is padded to KMAX rows
  // instead: the matrix with an identity block, the cotangents with zeros — every loop runs its full compile-time length.

  if constexpr (INV) {
    // rows K .. KMAX-1 of L read as zero
    for (int e = K * P + tl; e < KMAX * P; e += GS) Lb[e] = T(0);
    // ---- I1 / I2
    if (VECK) stage_lin(in + s * nfree, nfree); else stage_mat(in + s * KK);
    grp_sync();
    T dcc = T(0);
    if constexpr (CORR) {
      // tanh / sech of every free parameter, all lanes busy: rows p and K - p of the strict lower triangle hold K entries together.
      // z -> upper triangle (dead storage of L, read back in I5), sech -> lower
      for (int p = 1; 2 * p <= K; ++p) {
        int c, i;
        if (t < p) { c = p; i = t; } else { c = K - p; i = t - p; }
        if (act && i < c && (2 * p < K || t < p)) {
          T z, s2;
          M::tanh_sech2(B[pos(c, i)], z, s2);
          Lb[i * P + c] = z;
          Lb[c * P + i] = M::sqrt(s2);
        }
      }
      grp_sync();
      T E = T(1);
      GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
        const bool on = act && i < t;
        const T z = Lb[i * P + t], sech = Lb[t * P + i];
        if (on) Lb[t * P + i] = z * E;
        E = on ? E * sech : E;
        GRP_FENCE4(i);
      }
      if (act) Lb[t * P + t] = E;
      dcc = E;
    } else {
      GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
        const T raw = B[pos(t, i)];
        const T ex = M::exp(raw);
        const T v = i == t ? ex : raw;
        dcc = i == t ? ex : dcc;
        if (act && i <= t) Lb[t * P + i] = v;
      }
    }
    grp_sync();
    // ---- I3 / I4
    stage_mat(out_bar + s * KK);
    grp_sync();
    T acc[KMAX];
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) acc[j] = T(0);
    GRP_UNROLL for (int m = 0; m < KMAX; ++m) {
      const T raw = B[m * P + t] + B[t * P + m];
      const T sv = (act && m < K) ? raw : T(0);
      GRP_UNROLL for (int j0 = 0; j0 <= m; j0 += N) {                   // row m of L: 16-byte broadcast reads
        const RV x = *reinterpret_cast<const RV*>(Lb + m * P + j0);
        GRP_UNROLL for (int u = 0; u < N; ++u) if (j0 + u <= m) acc[j0 + u] += sv * x[u];
      }
      GRP_FENCE(m, acc);
    }
    grp_sync();
    // ---- I5
    T gcc = T(0);
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) gcc = j == t ? acc[j] : gcc;
    if constexpr (CORR) {
      T dlr = dcc * gcc + (dl + dl) + ((t >= 1 && t <= K - 2) ? T(K - 1 - t) * dl : T(0));
      T rem = dcc * dcc;
      GRP_UNROLL for (int i = KMAX - 1; i >= 0; --i) {
        const bool on = act && i < t;
        const T z = Lb[i * P + t], w = Lb[t * P + i], gw = acc[i];
        const T rem2 = rem + w * w;
        T rs, sq;
        M::pivot(rem2, rs, sq);
        const T f = rem2 > T(0) ? rem * rs : T(0);             // sech²(y) exp(log_remainder before entry i)
        if (on) B[pos(t, i)] = f * gw - z * dlr;
        rem = on ? rem2 : rem;
        dlr = on ? dlr + dl + w * gw : dlr;
        GRP_FENCE4(i);
      }
      if (KIND == MK_CORR) {
        GRP_UNROLL for (int i = 0; i < KMAX; ++i) if (act && i >= t && i < K) B[t * P + i] = T(0);
      }
    } else {
      GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
        const T v = i < t ? acc[i] : (i == t ? gcc * dcc + dl * T(K + 1 - t) : T(0));
        if (act && i < K && (i <= t || KIND == MK_PD)) B[pos(t, i)] = v;
      }
    }
    grp_sync();
    if (live) { if (VECK) unstage_lin(in_bar + s * nfree, nfree); else unstage_mat(in_bar + s * KK); }
  } else {
    // ---- F1 / F2: rows and columns K .. KMAX-1 are an identity block
    stage_mat(in + s * KK);
    grp_sync();
    T a[KMAX];
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) {
      const T raw = CORR ? B[t * P + j] : B[j * P + t];
      a[j] = (act && j <= t) ? raw : ((!act && j == t) ? T(1) : T(0));
    }
    grp_sync();
    GRP_UNROLL for (int k = 0; k < KMAX; ++k) {
      const T d = __shfl(a[k], gbase + k, 64);
      T rs, sq;
      M::pivot(d, rs, sq);
      a[k] = t == k ? sq : (t > k ? a[k] * rs : a[k]);
      Lb[t * P + k] = a[k];                                   // (lanes t < k write dead storage above the diagonal)
      colv[t] = t > k ? a[k] : T(0);                          // column k of L below the pivot, for the broadcast reads
      grp_sync();
      // a[j] -= L[t][k] L[j][k], j > k (entries j > t of a are never used: no mask)
      GRP_UNROLL for (int j0 = ((k + 1) / N) * N; j0 < KMAX; j0 += N) {
        const RV x = *reinterpret_cast<const RV*>(colv + j0);
        GRP_UNROLL for (int u = 0; u < N; ++u) if (j0 + u > k) a[j0 + u] -= a[k] * x[u];
      }
      GRP_FENCE(k, a);
    }
    // ---- F3 / F4
    if (nfree > 0) { if (VECK) stage_lin(out_bar + s * nfree, nfree); else stage_mat(out_bar + s * KK); }
    grp_sync();
    T g[KMAX];
    {
      T dcc = T(1);
      GRP_UNROLL for (int j = 0; j < KMAX; ++j) dcc = j == t ? a[j] : dcc;
      if constexpr (CORR) {
        T rem = dcc * dcc;
        GRP_UNROLL for (int i = KMAX - 1; i >= 0; --i) {
          g[i] = rem;
          rem = i < t ? rem + a[i] * a[i] : rem;
        }
        T gsum = T(0);
        GRP_UNROLL for (int m = 0; m < KMAX; ++m) {
          const bool on = m < t;
          const T w = a[m], yb = B[pos(t, m)], wt = T(K - m) * dl;
          T gm, dsum;
          if (KIND == MK_VEC_CORR && m == 0) {
            gm = (yb + wt * w) * M::rcp(T(1) - w * w);
            dsum = T(0);
          } else {
            const T R = g[m];
            const T S2 = R + w * w;
            const T rS = M::rcp(M::sqrt(S2)), rS2 = M::rcp(S2), rR = M::rcp(R);
            const T tt = yb * rS + wt * w * rS2;
            gm = tt + (w + w) * gsum;
            dsum = T(0.5) * rR * w * tt;
          }
          g[m] = on ? gm : T(0);
          gsum = on ? gsum - dsum : gsum;
          GRP_FENCE4(m);
        }
        const T gd = (dcc + dcc) * gsum;
        GRP_UNROLL for (int j = 0; j < KMAX; ++j) g[j] = j == t ? gd : g[j];
      } else {
        const T rd = M::rcp(dcc);
        GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
          const T raw = B[pos(t, i)];
          g[i] = i < t ? raw : (i == t ? (raw - dl * T(K + 1 - t)) * rd : T(0));
        }
      }
    }
    grp_sync();
    GRP_UNROLL for (int i = 0; i < KMAX; ++i) B[t * P + i] = (act && i <= t) ? g[i] : T(0);   // L̄, zero outside the K x K triangle
    grp_sync();
    // ---- F5: column t of Φ(L' L̄)
    T z[KMAX];
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) z[j] = T(0);
    GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
      const T gv = B[i * P + t];
      GRP_UNROLL for (int j0 = 0; j0 <= i; j0 += N) {
        const RV x = *reinterpret_cast<const RV*>(Lb + i * P + j0);
        GRP_UNROLL for (int u = 0; u < N; ++u) if (j0 + u <= i) z[j0 + u] += gv * x[u];
      }
      GRP_FENCE(i, z);
    }
    GRP_UNROLL for (int r = 0; r < KMAX; ++r) z[r] = r > t ? z[r] : (r == t ? T(0.5) * z[r] : T(0));
    // ---- F6: L' Z = Φ, column t — from the last row up, each solved entry eliminated with ONE row of L
    auto solve_lt = [&](T (&v)[KMAX]) __attribute__((always_inline)) {     // (left to the inliner, the 64-row instantiation calls it: the arrays go to scratch, 0.5 % of the HBM peak)
      GRP_UNROLL for (int r = KMAX - 1; r >= 0; --r) {
        v[r] *= M::rcp(Lb[r * P + r]);
        const T mv = -v[r];
        GRP_UNROLL for (int j0 = 0; j0 < r; j0 += N) {
          const RV x = *reinterpret_cast<const RV*>(Lb + r * P + j0);
          GRP_UNROLL for (int u = 0; u < N; ++u) if (j0 + u < r) v[j0 + u] += mv * x[u];
        }
        GRP_FENCE(r, v);
      }
    };
    solve_lt(z);
    // ---- F7: transpose
    grp_sync();
    GRP_UNROLL for (int r = 0; r < KMAX; ++r) B[r * P + t] = z[r];
    grp_sync();
    GRP_UNROLL for (int b = 0; b < KMAX; ++b) z[b] = B[t * P + b];
    // ---- F8: S L = Z, row t: the same elimination
    solve_lt(z);
    // ---- F9: Ā on the triangle the reference reads
    grp_sync();
    GRP_UNROLL for (int b = 0; b < KMAX; ++b) B[t * P + b] = z[b];
    grp_sync();
    T diag = T(0);
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) {
      diag = j == t ? z[j] : diag;
      const T other = B[j * P + t];
      z[j] = j < t ? z[j] + other : z[j];
    }
    grp_sync();
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) {
      const T v = j < t ? z[j] : (j == t ? diag : T(0));
      if (CORR) B[t * P + j] = v; else B[j * P + t] = v;
    }
    grp_sync();
    if (live) unstage_mat(in_bar + s * KK);
  }
}

template <class T, int GS, int KMAX, int KIND>
int grp_launch(bjx_ctx* ctx, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  constexpr int SPB = 256 / GS;
  const size_t smem = (size_t)SPB * GrpLds<T, GS, KMAX>::SS * sizeof(T);
  const int64_t grid = (batch + SPB - 1) / SPB;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "matrix pullback: batch too large for one launch");
  {
    BjxProf prof_(ctx);
    if (inverse) {
      bjx_allow_big_lds(matrix_grp_vjp_kernel<T, GS, KMAX, KIND, true>, smem);
      hipLaunchKernelGGL((matrix_grp_vjp_kernel<T, GS, KMAX, KIND, true>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, batch);
    } else {
      bjx_allow_big_lds(matrix_grp_vjp_kernel<T, GS, KMAX, KIND, false>, smem);
      hipLaunchKernelGGL((matrix_grp_vjp_kernel<T, GS, KMAX, KIND, false>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, batch);
    }
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T>
int grp_kind(bjx_ctx* ctx, int kind, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  // the unrolled phases cost ~KMAX² instructions whatever K is: 16, 24 and 32 rows
  if constexpr (sizeof(T) == 4) {
    // 32 < K <= 64 (Float32): the whole wave on one sample (the Float64 arrays would not fit the register file: those sizes stay on
    // the one-lane workspace kernel)
    if (K > 32) {
#define GRP_W(KIND_) (K <= 48 ? grp_launch<T, 64, 48, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch) \
                              : grp_launch<T, 64, 64, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch))
      switch (kind) {
        case MK_VEC_CORR: return GRP_W(MK_VEC_CORR);
        case MK_CORR: return GRP_W(MK_CORR);
        case MK_PD: return GRP_W(MK_PD);
        default: return GRP_W(MK_PD_VEC);
      }
#undef GRP_W
    }
  }
#define GRP_K(KIND_) (K <= 12 ? grp_launch<T, 16, 12, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch) \
                    : K <= 16 ? grp_launch<T, 16, 16, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch) \
                    : K <= 24 ? grp_launch<T, 32, 24, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch) \
                              : grp_launch<T, 32, 32, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch))
  switch (kind) {
    case MK_VEC_CORR: return GRP_K(MK_VEC_CORR);
    case MK_CORR: return GRP_K(MK_CORR);
    case MK_PD: return GRP_K(MK_PD);
    default: return GRP_K(MK_PD_VEC);
  }
#undef GRP_K
}

}  // namespace

namespace bjx {

int bjx_matrix_vjp_grp(bjx_ctx* ctx, bjx_dtype dt, int kind, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  static const int use_grp = getenv("BJX_MATRIX_VJP_GRP") ? atoi(getenv("BJX_MATRIX_VJP_GRP")) : 1;      // 0: the one-lane-per-sample workspace kernel (its A/B)
  // K = 9 ... 12 too: same call, 2^19 samples, K = 12: 24-48 % of the HBM peak here against 19-27 % for the twelve-row register kernel
  if (inverse && use_grp) {                               // the product on the matrix cores, odd pitch, in-place reverse sweep (bjx_matrix_vjp_mfma.hip)
    const int rc = bjx_matrix_inv_vjp_mfma(ctx, dt, kind, in, out_bar, ladj_bar, in_bar, K, batch);
    if (rc != 1) return rc;
  }
  if (!inverse && use_grp) {                              // W = L⁻¹ by blocks, the reverse of the factorisation as MFMA products (bjx_matrix_vjp_mfma_fwd.hip)
    const int rc = bjx_matrix_fwd_vjp_mfma(ctx, dt, kind, in, out_bar, ladj_bar, in_bar, K, batch);
    if (rc != 1) return rc;
  }
  if (!use_grp || K < 9 || K > 64 || (K > 32 && dt != BJX_F32)) return 1;
  if (dt == BJX_F32) return grp_kind<float>(ctx, kind, inverse, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
  return grp_kind<double>(ctx, kind, inverse, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
}

}  // namespace bjx
