// bjx_matrix_vjp_grp.hip — pullbacks of VecCorrBijector / CorrBijector / PDBijector / PDVecBijector and their inverses
// (SURVEY.md §8(f) f-1 x f-4; the rules and their reference lines: bjx_matrix_vjp.hip) for 12 < K <= 32.
//
// bjx_matrix_vjp.hip gives a sample to ONE lane; past 12 rows its triangles no longer fit the lane's registers and the same code
// ran on a lane-strided global workspace: 0.4 % of the HBM peak at K = 32.  Here a GROUP of GS = 16 / 32 lanes owns a sample
// (4 / 2 samples per wave, the group never leaves its wave: LDS traffic is ordered by the wave's queue, no block barrier), the
// factor L and a second K x K buffer B live in LDS at an odd pitch (lane = row and lane = column accesses are both conflict-free),
// and every phase is either "lane = row, sequential along the row" or "lane = column with wave-wide broadcast reads of L":
//
// inverse (unconstrained y -> X = L L'; what a leapfrog step differentiates):
//   I1  y -> B (coalesced)                                I2  lane c builds row c of L (the LKJ sweep / replace_diag(exp));
//       the correlation kinds park z = tanh(y) in the dead upper triangle of L for the way back
//   I3  X̄ -> B                                            I4  lane i: row i of L̄ = tril((X̄ + X̄') L) in registers
//   I5  lane c: reverse sweep of its row, with sech² · E recovered as Σ_{m>i} L[c][m]² / √(Σ_{m>=i} L[c][m]²) (sums of
//       squares only — the same number as (1 - z²) exp(log_remainder), no second tanh) -> B      I6  B -> in_bar (coalesced)
// forward (X -> y):
//   F1  X -> B     F2  right-looking Cholesky, lane i keeps row i in registers, column k published to L for the broadcast reads
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

// bank padding between the samples of a wave (in elements): neighbouring groups start GS banks apart (Float64: half-waves are
// served separately, only GS = 16 needs it)
template <class T, int GS> struct GrpPad { static constexpr int value = sizeof(T) == 4 ? (GS == 32 ? 32 : 48) : (GS == 32 ? 0 : 16); };

template <class T, int KMAX, int KIND, bool INV>
__global__ __launch_bounds__(256) void matrix_grp_vjp_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                            T* __restrict__ in_bar, int K, int64_t batch) {
  using M = VjpMath<T>;
  constexpr int GS = KMAX, P = KMAX + 1, SPB = 256 / GS;
  constexpr int SS = 2 * KMAX * P + GrpPad<T, GS>::value;
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  constexpr bool VECK = KIND == MK_VEC_CORR || KIND == MK_PD_VEC;
  extern __shared__ __align__(16) unsigned char smem_[];
  const int t = threadIdx.x & (GS - 1), sl = threadIdx.x / GS;
  T* Lb = reinterpret_cast<T*>(smem_) + (size_t)sl * SS;
  T* B = Lb + KMAX * P;
  const int64_t s_raw = (int64_t)blockIdx.x * SPB + sl;
  const bool live = s_raw < batch;                       // uniform over the group; a dead group computes on the last sample and stores nothing
  const int64_t s = live ? s_raw : batch - 1;
  const int64_t KK = (int64_t)K * K, nfree = free_len<KIND>(K);
  const T dl = ladj_bar ? ladj_bar[s] : T(0);
  const bool act = t < K;
  const int gbase = (threadIdx.x & 63) & ~(GS - 1);      // first lane of my group inside the wave

  auto stage_lin = [&](const T* src, int64_t n) { for (int64_t e = t; e < n; e += GS) B[e] = src[e]; };
  auto unstage_lin = [&](T* dst, int64_t n) { for (int64_t e = t; e < n; e += GS) dst[e] = B[e]; };
  auto stage_mat = [&](const T* src) {                   // K x K row-major -> pitch P, consecutive lanes on consecutive addresses
    int r = 0, c = t;
    while (c >= K) { c -= K; ++r; }
    while (r < K) {
      B[r * P + c] = src[r * K + c];
      c += GS;
      while (c >= K) { c -= K; ++r; }
    }
  };
  auto unstage_mat = [&](T* dst) {
    int r = 0, c = t;
    while (c >= K) { c -= K; ++r; }
    while (r < K) {
      dst[r * K + c] = B[r * P + c];
      c += GS;
      while (c >= K) { c -= K; ++r; }
    }
  };
  // where the free parameter of (factor row c, column i) sits in B (the unconstrained side staged as above)
  auto pos = [&](int c, int i) -> int {
    if (KIND == MK_VEC_CORR) return c * (c - 1) / 2 + i;
    if (KIND == MK_PD_VEC) return c * (c + 1) / 2 + i;
    if (KIND == MK_CORR) return c * P + i;               // memory index c K + i
    return i * P + c;                                    // MK_PD: memory index i K + c
  };
  // rows K .. KMAX-1 of L read as zero: the unrolled inner loops need no guards
  for (int e = K * P + t; e < KMAX * P; e += GS) Lb[e] = T(0);

  if constexpr (INV) {
    // ---- I1 / I2
    if (VECK) stage_lin(in + s * nfree, nfree); else stage_mat(in + s * KK);
    tile_sync();
    T dcc = T(0);
    if (act) {
      if constexpr (CORR) {
        T E = T(1);
        GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
          if (i < t) {
            T z, s2;
            M::tanh_sech2(B[pos(t, i)], z, s2);
            Lb[t * P + i] = z * E;
            Lb[i * P + t] = z;                            // upper triangle: dead storage, read back in I5
            E *= M::sqrt(s2);
          }
        }
        Lb[t * P + t] = E;
        dcc = E;
      } else {
        GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
          if (i <= t) {
            T v = B[pos(t, i)];
            if (i == t) { v = M::exp(v); dcc = v; }
            Lb[t * P + i] = v;
          }
        }
      }
    }
    tile_sync();
    // ---- I3 / I4
    stage_mat(out_bar + s * KK);
    tile_sync();
    T acc[KMAX];
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) acc[j] = T(0);
    GRP_UNROLL for (int m = 0; m < KMAX; ++m) {
      if (m < K) {
        const T sv = act ? B[m * P + t] + B[t * P + m] : T(0);
        GRP_UNROLL for (int j = 0; j <= m; ++j) acc[j] += sv * Lb[m * P + j];
      }
    }
    tile_sync();
    // ---- I5
    if (act) {
      T gcc = T(0);
      GRP_UNROLL for (int j = 0; j < KMAX; ++j) if (j == t) gcc = acc[j];
      if constexpr (CORR) {
        T dlr = dcc * gcc + (dl + dl) + ((t >= 1 && t <= K - 2) ? T(K - 1 - t) * dl : T(0));
        T rem = dcc * dcc;
        GRP_UNROLL for (int i = KMAX - 1; i >= 0; --i) {
          if (i < t) {
            const T z = Lb[i * P + t], w = Lb[t * P + i], gw = acc[i];
            const T prev = rem;
            rem += w * w;
            T rs, sq;
            M::pivot(rem, rs, sq);
            const T f = rem > T(0) ? prev * rs : T(0);     // sech²(y) exp(log_remainder before entry i)
            B[pos(t, i)] = f * gw - z * dlr;
            dlr += dl + w * gw;
          }
        }
        if (KIND == MK_CORR) {
          GRP_UNROLL for (int i = 0; i < KMAX; ++i) if (i >= t && i < K) B[t * P + i] = T(0);
        }
      } else {
        GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
          if (i < K) {
            if (i < t) B[pos(t, i)] = acc[i];
            else if (i == t) B[pos(t, t)] = gcc * dcc + dl * T(K + 1 - t);
            else if (KIND == MK_PD) B[i * P + t] = T(0);
          }
        }
      }
    }
    tile_sync();
    if (live) { if (VECK) unstage_lin(in_bar + s * nfree, nfree); else unstage_mat(in_bar + s * KK); }
  } else {
    // ---- F1 / F2
    stage_mat(in + s * KK);
    tile_sync();
    T a[KMAX];
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) a[j] = (act && j <= t) ? (CORR ? B[t * P + j] : B[j * P + t]) : T(0);
    tile_sync();
    GRP_UNROLL for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const T d = __shfl(a[k], gbase + k, 64);
        T rs, sq;
        M::pivot(d, rs, sq);
        if (t == k) a[k] = sq; else if (t > k) a[k] *= rs;
        if (act && t >= k) Lb[t * P + k] = a[k];
        tile_sync();
        GRP_UNROLL for (int j = k + 1; j < KMAX; ++j) {
          const T ljk = Lb[j * P + k];
          if (j <= t) a[j] -= a[k] * ljk;
        }
      }
    }
    // ---- F3 / F4
    if (nfree > 0) { if (VECK) stage_lin(out_bar + s * nfree, nfree); else stage_mat(out_bar + s * KK); }
    tile_sync();
    T g[KMAX];
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) g[j] = T(0);
    if (act) {
      T dcc = T(0);
      GRP_UNROLL for (int j = 0; j < KMAX; ++j) if (j == t) dcc = a[j];
      if constexpr (CORR) {
        T rem = dcc * dcc;
        GRP_UNROLL for (int i = KMAX - 1; i >= 0; --i) {
          if (i < t) { g[i] = rem; rem += a[i] * a[i]; }
        }
        T gsum = T(0);
        GRP_UNROLL for (int m = 0; m < KMAX; ++m) {
          if (m < t) {
            const T w = a[m], yb = B[pos(t, m)], wt = T(K - m) * dl;
            if (KIND == MK_VEC_CORR && m == 0) {
              g[m] = (yb + wt * w) * M::rcp(T(1) - w * w);
            } else {
              const T R = g[m];
              const T S2 = R + w * w;
              const T rS = M::rcp(M::sqrt(S2)), rS2 = M::rcp(S2), rR = M::rcp(R);
              const T tt = yb * rS + wt * w * rS2;
              g[m] = tt + (w + w) * gsum;
              gsum -= T(0.5) * rR * w * tt;
            }
          }
        }
        const T gd = (dcc + dcc) * gsum;
        GRP_UNROLL for (int j = 0; j < KMAX; ++j) if (j == t) g[j] = gd;
      } else {
        const T rd = M::rcp(dcc);
        GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
          if (i < t) g[i] = B[pos(t, i)];
          else if (i == t) g[i] = (B[pos(t, t)] - dl * T(K + 1 - t)) * rd;
        }
      }
    }
    tile_sync();
    if (act) {
      GRP_UNROLL for (int i = 0; i < KMAX; ++i) if (i < K) B[t * P + i] = i <= t ? g[i] : T(0);
    }
    tile_sync();
    // ---- F5: column t of Φ(L' L̄)
    T z[KMAX];
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) z[j] = T(0);
    GRP_UNROLL for (int i = 0; i < KMAX; ++i) {
      if (i < K) {
        const T gv = act ? B[i * P + t] : T(0);
        GRP_UNROLL for (int r = 0; r <= i; ++r) z[r] += Lb[i * P + r] * gv;
      }
    }
    GRP_UNROLL for (int r = 0; r < KMAX; ++r) z[r] = r > t ? z[r] : (r == t ? T(0.5) * z[r] : T(0));
    // ---- F6: L' Z = Φ, column t
    GRP_UNROLL for (int r = KMAX - 1; r >= 0; --r) {
      if (r < K) {
        T sum = z[r];
        GRP_UNROLL for (int q = r + 1; q < KMAX; ++q) sum -= Lb[q * P + r] * z[q];
        z[r] = sum * M::rcp(Lb[r * P + r]);
      } else z[r] = T(0);
    }
    // ---- F7: transpose
    tile_sync();
    if (act) { GRP_UNROLL for (int r = 0; r < KMAX; ++r) if (r < K) B[r * P + t] = z[r]; }
    tile_sync();
    GRP_UNROLL for (int b = 0; b < KMAX; ++b) z[b] = (act && b < K) ? B[t * P + b] : T(0);
    // ---- F8: S L = Z, row t
    GRP_UNROLL for (int b = KMAX - 1; b >= 0; --b) {
      if (b < K) {
        T sum = z[b];
        GRP_UNROLL for (int c = b + 1; c < KMAX; ++c) sum -= z[c] * Lb[c * P + b];
        z[b] = sum * M::rcp(Lb[b * P + b]);
      } else z[b] = T(0);
    }
    // ---- F9: Ā on the triangle the reference reads
    tile_sync();
    if (act) { GRP_UNROLL for (int b = 0; b < KMAX; ++b) if (b < K) B[t * P + b] = z[b]; }
    tile_sync();
    T diag = T(0);
    GRP_UNROLL for (int j = 0; j < KMAX; ++j) {
      if (j == t) diag = z[j];
      if (j < t) z[j] += B[j * P + t];
    }
    tile_sync();
    if (act) {
      GRP_UNROLL for (int j = 0; j < KMAX; ++j) {
        if (j < K) {
          const T v = j < t ? z[j] : (j == t ? diag : T(0));
          if (CORR) B[t * P + j] = v; else B[j * P + t] = v;
        }
      }
    }
    tile_sync();
    if (live) unstage_mat(in_bar + s * KK);
  }
}

template <class T, int KMAX, int KIND>
int grp_launch(bjx_ctx* ctx, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  constexpr int GS = KMAX, P = KMAX + 1, SPB = 256 / GS;
  const size_t smem = (size_t)SPB * (2 * KMAX * P + GrpPad<T, GS>::value) * sizeof(T);
  const int64_t grid = (batch + SPB - 1) / SPB;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "matrix pullback: batch too large for one launch");
  {
    BjxProf prof_(ctx);
    if (inverse) {
      bjx_allow_big_lds(matrix_grp_vjp_kernel<T, KMAX, KIND, true>, smem);
      hipLaunchKernelGGL((matrix_grp_vjp_kernel<T, KMAX, KIND, true>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, batch);
    } else {
      bjx_allow_big_lds(matrix_grp_vjp_kernel<T, KMAX, KIND, false>, smem);
      hipLaunchKernelGGL((matrix_grp_vjp_kernel<T, KMAX, KIND, false>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, batch);
    }
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T>
int grp_kind(bjx_ctx* ctx, int kind, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
#define GRP_K(KIND_) (K <= 16 ? grp_launch<T, 16, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch) : grp_launch<T, 32, KIND_>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch))
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
  if (!use_grp || K <= 12 || K > 32) return 1;
  if (dt == BJX_F32) return grp_kind<float>(ctx, kind, inverse, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
  return grp_kind<double>(ctx, kind, inverse, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
}

}  // namespace bjx
