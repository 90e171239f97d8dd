// bjx_matrix_vjp_mfma.hip — pullback of the INVERSE VecCorrBijector / CorrBijector / PDBijector / PDVecBijector (unconstrained y -> X = L L',
// what a leapfrog step differentiates; SURVEY.md §8(f) f-1 x f-4; the rules and their reference lines: bjx_matrix_vjp.hip) for 8 < K <= 64
// (both element types; Float64 at 49 .. 64 rows in blocks of two waves: 66 KiB of LDS a sample), with the one cubic step on the matrix cores.
//
// bjx_matrix_vjp_grp.hip does L̄ = tril((X̄ + X̄') L) with "lane = row": K²/2 FMAs per lane fed by 16-byte broadcast reads of L out of LDS,
// a K-register accumulator row per lane, and a row pitch that has to stay a multiple of 16 bytes for those reads — which puts every
// "lane = row" access of the two sweeps on 16 of the 64 banks.  Measured with phases removed (K = 32, 2^16 samples, 0.292 ms in all):
// the product 0.061 ms, the tanh / LKJ sweep 0.049 ms, the reverse sweep 0.029 ms, and 0.155 ms of staging and waiting that no
// arithmetic explains (two request phases per sample, two waves per SIMD).  Here
//   * the product runs as 16 x 16 x 4 MFMA blocks, a whole wave on one sample at a time (only the blocks on and below the diagonal, only
//     the k-blocks where L is not zero: 20 instructions at K = 32, 120 at K = 64); the operands come straight from the LDS copies of
//     X̄ and L (X̄ + X̄' is formed in the load), the result goes back into the buffer that held X̄;
//   * the row pitch is ODD: every "lane = row" and every "lane = column" access of the sweeps is conflict-free;
//   * the reverse sweep works IN PLACE on the product (lane t reads L̄[t][i] and writes ȳ's entry to the same word), so no lane
//     holds a row in registers; the packed (vec) layouts leave through a 16-bit offset table built once per block;
//   * blocks are persistent and the two arrays of a group's NEXT sample travel in registers while it computes.
// Phases (per sample, a group of GS = 16 / 32 / 64 lanes; the group never leaves its wave — LDS traffic is ordered by the wave's queue):
//   I1  y -> B                      I2  tanh / sech of every entry, then lane c builds row c of L (LKJ sweep) / replace_diag(exp);
//       z = tanh(y) parked in the dead upper triangle of L          I3  X̄ -> B
//   I4  B <- tril((X̄ + X̄') L) (MFMA; stored transposed for the PD kinds, whose free parameter (c, i) lives at [i][c])
//   I5  lane c: reverse sweep of its row, in place                   I6  B -> in_bar
// Algorithmic bytes per sample as in bjx_matrix_vjp.hip.
#include <cstdlib>
#include <type_traits>

#include "bjx_internal.h"
#include "bjx_tile.h"
#include "bjx_matrix_vjp.h"

using namespace bjx;

namespace {

#define MF_UNROLL _Pragma("unroll")
__device__ __forceinline__ void mf_sync() { tile_sync(); __builtin_amdgcn_sched_barrier(0); }
#define MF_FENCE4(i_) do { if (((i_) % 4) == 3) __builtin_amdgcn_sched_barrier(0); } while (0)

// LDS of one sample, in elements: L [KMAX][P] | B [KMAX][P], P odd.  The groups of a wave sit GS banks apart (Float32; Float64 moves
// half a wave per pass), so the "lane = row" accesses of a whole wave cover the 64 banks once.
template <class T, int GS, int KMAX> struct MfLds {
  static constexpr int P = KMAX + 1;
  static constexpr int BASE = 2 * KMAX * P;
  static constexpr int W = sizeof(T) / 4;
  static constexpr int TARGET = sizeof(T) == 4 ? GS % 64 : (GS >= 32 ? 0 : 32);
  static constexpr int pad() { int q = 0; while (((BASE + q) * W) % 64 != TARGET) ++q; return q; }
  static constexpr int SS = BASE + pad();
};

template <class T, int GS, int KMAX, int KIND, int VWT, int NT>
__global__ __launch_bounds__(NT) void matrix_inv_vjp_mfma_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                                 T* __restrict__ in_bar, int K, int64_t batch) {
  using M = VjpMath<T>;
  using O = VjpMfma<T>;
  // VWT: elements per global access — 1, a 16-byte pack, or (Float32) an 8-byte pair for the sizes whose rows and free lengths are even
  // but not whole packs (K = 12, 20, 28, ...)
  typedef T RV __attribute__((ext_vector_type(VWT == 1 ? 2 : VWT)));
  using ACC = typename O::acc_t;
  constexpr bool VEC = VWT > 1;
  constexpr int N = VWT, P = MfLds<T, GS, KMAX>::P, SPB = NT / GS, SPW = 64 / GS, SS = MfLds<T, GS, KMAX>::SS;
  constexpr int NIT = (KMAX * KMAX + GS - 1) / GS;           // staging rounds of the group over a K x K array, one element per lane
  constexpr int NITV = (NIT + N - 1) / N;                    // ... one 16-byte pack per lane (VEC: K and the free length whole packs, arrays on 16-byte boundaries)
  constexpr int NREG = VEC ? NITV * N : NIT;
  constexpr int NB = (KMAX + 15) / 16;                       // 16-blocks per side
  constexpr int SC = 4;                                      // columns per chunk of the two sweeps (KMAX is a multiple)
  static_assert(KMAX % SC == 0, "sweep chunks");
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  constexpr bool VECK = KIND == MK_VEC_CORR || KIND == MK_PD_VEC;
  extern __shared__ __align__(16) unsigned char smem_[];
  // (not const: the persistent loop below launders the lane-derived values once per sample — see there)
  int tl = threadIdx.x & (GS - 1), sl = threadIdx.x / GS;
  int t = tl < KMAX ? tl : KMAX - 1;                         // lanes past KMAX repeat the last row: same values to the same addresses
  int loff = sl * SS;
  T* Lb = reinterpret_cast<T*>(smem_) + loff;
  T* B = Lb + KMAX * P;
  unsigned short* tab = reinterpret_cast<unsigned short*>(reinterpret_cast<T*>(smem_) + (size_t)SPB * SS);
  const int64_t KK = (int64_t)K * K, nfree = free_len<KIND>(K);
  bool act = t < K;

  // where the free parameter of (factor row c, column i) sits in B after I1 (the unconstrained side as staged) ...
  auto pos = [&](int c, int i) -> int {
    if (KIND == MK_VEC_CORR) return c * (c - 1) / 2 + i;
    if (KIND == MK_PD_VEC) return c * (c + 1) / 2 + i;
    if (KIND == MK_CORR) return c * P + i;               // memory index c K + i
    return i * P + c;                                    // MK_PD: memory index i K + c
  };
  // ... and where its cotangent is produced (I4 / I5, in place): L̄[c][i] at [c][i] for the Corr kinds, transposed for the PD kinds
  auto slot = [&](int c, int i) -> int { return CORR ? c * P + i : i * P + c; };

  constexpr int VW = VEC ? N : 1, NV = VEC ? NITV : NIT;  // elements per staging access, accesses per K x K array
  constexpr int NFMAX = KIND == MK_VEC_CORR ? KMAX * (KMAX - 1) / 2 : (KIND == MK_PD_VEC ? KMAX * (KMAX + 1) / 2 : KMAX * KMAX);
  constexpr int NVA = (NFMAX + GS * VW - 1) / (GS * VW);   // ... per array of free parameters (half as many for the packed layouts)
  if constexpr (VECK) {
    // packed index e -> (c << 8) | i, once per block (the block is persistent)
    for (int e = threadIdx.x; e < (int)nfree; e += NT) {
      int c;
      if (KIND == MK_VEC_CORR) {                          // e = c (c - 1) / 2 + i, i < c
        c = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
        while (c * (c - 1) / 2 > e) --c;
        while ((c + 1) * c / 2 <= e) ++c;
        tab[e] = (unsigned short)((c << 8) | (e - c * (c - 1) / 2));
      } else {                                            // e = c (c + 1) / 2 + i, i <= c
        c = (int)((__builtin_sqrtf(1.0f + 8.0f * (float)e) - 1.0f) * 0.5f);
        while (c * (c + 1) / 2 > e) --c;
        while ((c + 1) * (c + 2) / 2 <= e) ++c;
        tab[e] = (unsigned short)((c << 8) | (e - c * (c + 1) / 2));
      }
    }
  }
  // Everything outside the K x K corner of both buffers is zero and STAYS zero (staging writes the corner only, the product of zero
  // rows / columns is zero, the sweeps write back what they read): no masks on K in the loops below.
  for (int e = tl; e < 2 * KMAX * P; e += GS) Lb[e] = T(0);
  __syncthreads();

  // Staging never branches: a slot past the end of the array re-reads element 0 and re-writes it — the same value to the same word.
  // (r, c) of my first element in a K x K array, and the step of one round of the group
  const int e0 = tl * VW;
  int r0 = e0 / K, c0 = e0 - r0 * K;
  const int dr = (GS * VW) / K, dc = (GS * VW) - dr * K;
  auto issue = [&](const T* src, int64_t n, T (&v)[NREG], auto nv_) {
    constexpr int NV_ = decltype(nv_)::value;
    MF_UNROLL for (int it = 0; it < NV_; ++it) {
      const int e = (tl + it * GS) * VW;
      if constexpr (VEC) {
        const RV x = *reinterpret_cast<const RV*>(src + (e < n ? e : 0));
        MF_UNROLL for (int u = 0; u < N; ++u) v[it * N + u] = x[u];
      } else {
        v[it] = src[e < n ? e : 0];
      }
    }
  };
  // I1 + the elementwise half of I2 for the packed layouts, straight from the registers: every lane busy, no pass over B
  auto scatter_packed = [&](const T (&v)[NREG]) {
    MF_UNROLL for (int it = 0; it < NVA; ++it) {
      const int e = (tl + it * GS) * VW;
      const int ee = e < (int)nfree ? e : 0;
      MF_UNROLL for (int u = 0; u < VW; ++u) {
        const int ci = tab[ee + u], c = ci >> 8, i = ci & 255;
        const T raw = v[it * VW + u];
        if constexpr (CORR) {
          T z, sech;
          M::tanh_sech(raw, z, sech);
          Lb[i * P + c] = z;                                 // z -> upper triangle (dead storage of L, read back in I5), sech -> lower
          Lb[c * P + i] = sech;
        } else {
          Lb[c * P + i] = raw;
        }
      }
    }
  };
  auto commit_mat = [&](const T (&v)[NREG]) {            // K x K row-major -> pitch P
    int r = r0, c = c0;
    MF_UNROLL for (int it = 0; it < NV; ++it) {
      const int a = r < K ? r * P + c : 0;
      MF_UNROLL for (int u = 0; u < VW; ++u) B[a + u] = v[it * VW + u];
      c += dc; r += dr;
      const bool wrap = c >= K;
      c = wrap ? c - K : c;
      r = wrap ? r + 1 : r;
    }
  };
  auto unstage_packed = [&](T* dst) {                     // in_bar[e] = the slot of (c, i)
    MF_UNROLL for (int it = 0; it < NVA; ++it) {
      const int e = (tl + it * GS) * VW;
      const int ee = e < (int)nfree ? e : 0;
      T x[VW];
      MF_UNROLL for (int u = 0; u < VW; ++u) {
        const int ci = tab[ee + u], c = ci >> 8, i = ci & 255;
        x[u] = B[CORR ? c * P + i : i * P + c];
      }
      if constexpr (VEC) {
        RV xv;
        MF_UNROLL for (int u = 0; u < N; ++u) xv[u] = x[u];
        *reinterpret_cast<RV*>(dst + ee) = xv;
      } else {
        dst[ee] = x[0];
      }
    }
  };
  auto unstage_mat = [&](T* dst) {
    int r = r0, c = c0;
    MF_UNROLL for (int it = 0; it < NV; ++it) {
      const bool ok = r < K;
      const int a = ok ? r * P + c : 0;
      const int g = ok ? r * K + c : 0;
      if constexpr (VEC) {
        RV xv;
        MF_UNROLL for (int u = 0; u < N; ++u) xv[u] = B[a + u];
        *reinterpret_cast<RV*>(dst + g) = xv;
      } else {
        dst[g] = B[a];
      }
      c += dc; r += dr;
      const bool wrap = c >= K;
      c = wrap ? c - K : c;
      r = wrap ? r + 1 : r;
    }
  };

  const int64_t stride = (int64_t)gridDim.x * SPB;
  const int64_t n_a = VECK ? nfree : KK;
  int64_t s_raw = (int64_t)blockIdx.x * SPB + sl;
  int64_t w_raw = (int64_t)blockIdx.x * SPB + (threadIdx.x / 64) * SPW;                          // first group of my wave: the trip count is the wave's
  T pa[NREG], pb[NREG], dl_next;
  {
    const int64_t s0 = s_raw < batch ? s_raw : batch - 1;
    issue(in + s0 * n_a, n_a, pa, std::integral_constant<int, NVA>{});
    issue(out_bar + s0 * KK, KK, pb, std::integral_constant<int, NV>{});
    dl_next = ladj_bar ? ladj_bar[s0] : T(0);
  }
  // MFMA lane coordinates and the LDS of the wave's first group
  int lane = threadIdx.x & 63, mn = lane & 15, mq = lane >> 4;
  int woff = (threadIdx.x / 64) * SPW * SS;
  T* Lw = reinterpret_cast<T*>(smem_) + woff;
  unsigned te = act ? (unsigned)t : 0u;                    // "column i is left of my diagonal": i < te (never, for a lane without a row)

  for (; w_raw < batch; w_raw += stride, s_raw += stride) {
    // Everything below that depends only on the lane — LDS addresses, masks, the (row, column) pairs of the staging rounds — is loop-
    // invariant: the compiler hoists it out of the persistent loop and keeps it in registers for the whole kernel.  Laundering the
    // seeds once per sample makes it recompute them where they are used.
    // (From 24 rows on: K = 48 17 % faster, K = 12 18 % slower — four small samples a wave, the recomputation is what it then measures.)
    if constexpr (KMAX >= 24) asm volatile("" : "+v"(tl), "+v"(t), "+v"(loff), "+v"(woff), "+v"(r0), "+v"(c0), "+v"(mn), "+v"(mq));
    Lb = reinterpret_cast<T*>(smem_) + loff;
    B = Lb + KMAX * P;
    Lw = reinterpret_cast<T*>(smem_) + woff;
    act = t < K;
    te = act ? (unsigned)t : 0u;
    const bool live = s_raw < batch;                     // uniform over the group; a dead group computes on the last sample and stores nothing
    const int64_t s = live ? s_raw : batch - 1;
    const int64_t sn = s_raw + stride < batch ? s_raw + stride : batch - 1;
    const T dl = dl_next;
    dl_next = ladj_bar ? ladj_bar[sn] : T(0);
    // ---- I1 / I2
    if constexpr (VECK) {
      scatter_packed(pa);
    } else {
      commit_mat(pa);
      mf_sync();
      if constexpr (CORR) {
        // tanh / sech of every free parameter, all lanes busy: rows p and K - p of the strict lower triangle hold K entries together
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
      } else {
        MF_UNROLL for (int i = 0; i < KMAX; ++i) {
          const T raw = B[pos(t, i)], old = Lb[t * P + i];
          Lb[t * P + i] = (unsigned)i <= te && act ? raw : old;
        }
      }
    }
    mf_sync();
    issue(in + sn * n_a, n_a, pa, std::integral_constant<int, NVA>{});
    T dcc;
    if constexpr (CORR) {
      // the LKJ sweep: lane t owns row t; a column at or right of the diagonal gets its old word back
      // (chunks of SC columns, the reads of the next chunk in flight while this one is computed: at one wave per SIMD nothing else
      // covers the LDS round trip)
      T E = T(1);
      T zb[2][SC], sb[2][SC];
      MF_UNROLL for (int u = 0; u < SC; ++u) { zb[0][u] = Lb[u * P + t]; sb[0][u] = Lb[t * P + u]; }
      MF_UNROLL for (int ch = 0; ch < KMAX / SC; ++ch) {
        if (ch + 1 < KMAX / SC) {
          MF_UNROLL for (int u = 0; u < SC; ++u) { const int i = (ch + 1) * SC + u; zb[(ch + 1) & 1][u] = Lb[i * P + t]; sb[(ch + 1) & 1][u] = Lb[t * P + i]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        MF_UNROLL for (int u = 0; u < SC; ++u) {
          const int i = ch * SC + u;
          const bool on = (unsigned)i < te;
          const T z = zb[ch & 1][u], sech = sb[ch & 1][u];
          Lb[t * P + i] = on ? z * E : sech;
          E = on ? E * sech : E;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (act) Lb[t * P + t] = E;
      dcc = E;
    } else {
      dcc = M::exp(Lb[t * P + t]);
      if (act) Lb[t * P + t] = dcc;
    }
    mf_sync();
    // ---- I3
    commit_mat(pb);
    mf_sync();
    issue(out_bar + sn * KK, KK, pb, std::integral_constant<int, NV>{});
    // ---- I4: the whole wave on one sample at a time.  A operand: (X̄ + X̄')[16 bi + n][k]; B operand: L[k][16 bj + n] on and below the
    // diagonal (the upper triangle of the buffer holds z: masked in the diagonal blocks); k = 16 kb + 4 ks + q.  Sizes that are not whole
    // 16-blocks (12, 24, 48 rows) clamp the address and read zero past KMAX.
    constexpr bool RAGGED = KMAX % 16 != 0;
    MF_UNROLL for (int j = 0; j < SPW; ++j) {
      const T* Lj = Lw + (size_t)j * SS;
      T* Bj = const_cast<T*>(Lj) + KMAX * P;
      ACC d[NB][NB];
      MF_UNROLL for (int bi = 0; bi < NB; ++bi) MF_UNROLL for (int bj = 0; bj <= bi; ++bj) d[bi][bj] = ACC{T(0), T(0), T(0), T(0)};
      MF_UNROLL for (int kb = 0; kb < NB; ++kb) {
        MF_UNROLL for (int ks = 0; ks < 4; ++ks) {
          const int k = 16 * kb + 4 * ks + mq;
          const bool kin = !RAGGED || 16 * kb + 4 * ks + 3 < KMAX || k < KMAX;
          const int kc = kin ? k : 0;
          T a[NB], b[NB];
          MF_UNROLL for (int bi = 0; bi < NB; ++bi) {
            const int r = 16 * bi + mn;
            const bool rin = !RAGGED || 16 * bi + 15 < KMAX || r < KMAX;
            const int rc = rin ? r : 0;
            const T x = Bj[rc * P + kc] + Bj[kc * P + rc];
            a[bi] = (rin && kin) ? x : T(0);
          }
          MF_UNROLL for (int bj = 0; bj <= kb; ++bj) {
            const int c = 16 * bj + mn;
            const bool cin = !RAGGED || 16 * bj + 15 < KMAX || c < KMAX;
            const T x = Lj[kc * P + (cin ? c : 0)];
            b[bj] = (kin && cin && (bj < kb || mn <= 4 * ks + mq)) ? x : T(0);
          }
          MF_UNROLL for (int bj = 0; bj <= kb; ++bj) MF_UNROLL for (int bi = bj; bi < NB; ++bi) d[bi][bj] = O::mfma(a[bi], b[bj], d[bi][bj]);
        }
      }
      mf_sync();                                           // every read of X̄ is done: the product takes its place
      MF_UNROLL for (int bi = 0; bi < NB; ++bi) MF_UNROLL for (int bj = 0; bj <= bi; ++bj) MF_UNROLL for (int r = 0; r < 4; ++r) {
        const int row = 16 * bi + O::row(mq, r), col = 16 * bj + mn;
        if (!RAGGED || (row < KMAX && col < KMAX)) Bj[CORR ? row * P + col : col * P + row] = d[bi][bj][r];
      }
    }
    mf_sync();
    // ---- I5, in place
    const T gcc = B[slot(t, t)];
    if constexpr (CORR) {
      T dlr = dcc * gcc + (dl + dl) + ((t >= 1 && t <= K - 2) ? T(K - 1 - t) * dl : T(0));
      T rem = dcc * dcc;
      T zr[2][SC], wr[2][SC], gr[2][SC];
      MF_UNROLL for (int u = 0; u < SC; ++u) { const int i = KMAX - 1 - u; zr[0][u] = Lb[i * P + t]; wr[0][u] = Lb[t * P + i]; gr[0][u] = B[t * P + i]; }
      MF_UNROLL for (int ch = 0; ch < KMAX / SC; ++ch) {
        if (ch + 1 < KMAX / SC) {
          MF_UNROLL for (int u = 0; u < SC; ++u) {
            const int i = KMAX - 1 - ((ch + 1) * SC + u);
            zr[(ch + 1) & 1][u] = Lb[i * P + t]; wr[(ch + 1) & 1][u] = Lb[t * P + i]; gr[(ch + 1) & 1][u] = B[t * P + i];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        MF_UNROLL for (int u = 0; u < SC; ++u) {
          const int i = KMAX - 1 - (ch * SC + u);
          const bool on = (unsigned)i < te;
          const T z = zr[ch & 1][u], w = wr[ch & 1][u], gw = gr[ch & 1][u];
          const T rem2 = rem + w * w;
          T rs, sq;
          M::pivot(rem2 > T(0) ? rem2 : T(1), rs, sq);
          const T f = rem * rs;                                  // sech²(y) exp(log_remainder before entry i)
          B[t * P + i] = on ? f * gw - z * dlr : (KIND == MK_CORR ? T(0) : gw);
          rem = on ? rem2 : rem;
          dlr = on ? dlr + dl + w * gw : dlr;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (act) B[slot(t, t)] = gcc * dcc + dl * T(K + 1 - t);
      if (KIND == MK_PD) {
        MF_UNROLL for (int i = 0; i < KMAX; ++i) if (act && i > t && i < K) B[slot(t, i)] = T(0);
      }
    }
    mf_sync();
    if (live) { if constexpr (VECK) unstage_packed(in_bar + s * nfree); else unstage_mat(in_bar + s * KK); }
    mf_sync();
  }
}

template <class T, int GS, int KMAX, int KIND, int VWT, int NT>
void mf_launch_one(bjx_ctx* ctx, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  constexpr int SPB = NT / GS;
  const size_t smem = (size_t)SPB * MfLds<T, GS, KMAX>::SS * sizeof(T) + ((size_t)KMAX * (KMAX + 1) / 2) * sizeof(unsigned short) + 16;
  auto kern = matrix_inv_vjp_mfma_kernel<T, GS, KMAX, KIND, VWT, NT>;
  // persistent blocks: as many as are resident at once (LDS decides), each walks its share of the samples
  static int per_cu = 0;                                  // (one value per instantiation; the same on every device of a node)
  if (per_cu == 0) {
    bjx_allow_big_lds(kern, smem);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), NT, smem) != hipSuccess || nb < 1) nb = 1;
    per_cu = nb;
  }
  const int64_t need = (batch + SPB - 1) / SPB, cap = (int64_t)ctx->num_cu * per_cu;
  const int64_t grid = need < cap ? need : cap;
  BjxProf prof_(ctx);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, batch);
}

template <class T, int GS, int KMAX, int KIND, int NT = 256>
int mf_launch(bjx_ctx* ctx, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  constexpr int N = VjpMfma<T>::N;
  // 16-byte staging: rows and free lengths in whole packs, the arrays on 16-byte boundaries (every sample then starts on one); Float32
  // with even rows and free lengths: 8-byte pairs
  const int64_t nf = free_len<KIND>(K);
  const auto al = [&](unsigned m) { return ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out_bar) | reinterpret_cast<uintptr_t>(in_bar)) & m) == 0; };
  if (K % N == 0 && nf % N == 0 && al(15)) mf_launch_one<T, GS, KMAX, KIND, N, NT>(ctx, in, out_bar, ladj_bar, in_bar, K, batch);
  else if (sizeof(T) == 4 && K % 2 == 0 && nf % 2 == 0 && al(7)) mf_launch_one<T, GS, KMAX, KIND, 2, NT>(ctx, in, out_bar, ladj_bar, in_bar, K, batch);
  else mf_launch_one<T, GS, KMAX, KIND, 1, NT>(ctx, in, out_bar, ladj_bar, in_bar, K, batch);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T>
int mf_kind(bjx_ctx* ctx, int kind, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  if (K > 32) {
    // 33 .. 64 rows: the whole wave on one sample.  Float64 at 49 .. 64 rows: 66 KiB of LDS a sample — blocks of two waves (two samples a CU)
    constexpr int NT64 = sizeof(T) == 4 ? 256 : 128;
#define MF_W(KIND_) (K <= 48 ? mf_launch<T, 64, 48, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                             : mf_launch<T, 64, 64, KIND_, NT64>(ctx, in, out_bar, ladj_bar, in_bar, K, batch))
    switch (kind) {
      case MK_VEC_CORR: return MF_W(MK_VEC_CORR);
      case MK_CORR: return MF_W(MK_CORR);
      case MK_PD: return MF_W(MK_PD);
      default: return MF_W(MK_PD_VEC);
    }
#undef MF_W
  }
#define MF_K(KIND_) (K <= 12 ? mf_launch<T, 16, 12, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                   : K <= 16 ? mf_launch<T, 16, 16, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                   : K <= 24 ? mf_launch<T, 32, 24, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                             : mf_launch<T, 32, 32, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch))
  switch (kind) {
    case MK_VEC_CORR: return MF_K(MK_VEC_CORR);
    case MK_CORR: return MF_K(MK_CORR);
    case MK_PD: return MF_K(MK_PD);
    default: return MF_K(MK_PD_VEC);
  }
#undef MF_K
}

}  // namespace

namespace bjx {

// 1: not served (the caller goes on to the lane = row group kernel)
int bjx_matrix_inv_vjp_mfma(bjx_ctx* ctx, bjx_dtype dt, int kind, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  static const int use = getenv("BJX_MATRIX_VJP_MFMA") ? atoi(getenv("BJX_MATRIX_VJP_MFMA")) : 1;      // 0: the lane = row group kernel (its A/B)
  if (!use || K < 9 || K > 64) return 1;
  if (dt == BJX_F32) return mf_kind<float>(ctx, kind, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
  return mf_kind<double>(ctx, kind, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
}

}  // namespace bjx
