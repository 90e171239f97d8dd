// bjx_matrix_vjp_mfma_fwd.hip — pullback of the FORWARD VecCorrBijector / CorrBijector / PDBijector / PDVecBijector (X -> unconstrained y;
// SURVEY.md §8(f) f-1 x f-4; the rules and their reference lines: bjx_matrix_vjp.hip) for 8 < K <= 64 (both element types), with every cubic
// step after the factorisation on the matrix cores.
//
// bjx_matrix_vjp_grp.hip reverses the Cholesky factorisation as S = L⁻ᵀ Φ(LᵀL̄) L⁻¹ with a triangular product and two triangular SOLVES,
// each "lane = row / column, K²/2 FMAs per lane out of 16-byte broadcast reads" with a K-register row per lane: 19 % of the HBM peak at
// K = 32, 7 % at K = 64.  A solve does not map on MFMA, a product does: here W = L⁻¹ is formed explicitly, BY BLOCKS of 16 —
//   the diagonal blocks by forward substitution (lane = (block, column), 16 steps, all blocks of a sample at once),
//   the blocks below them as W(i,j) = −W(i,i) Σ_{j<=m<i} L(i,m) W(m,j), top row first —
// and S = Wᵀ Φ(LᵀL̄) W is four block products in a row: P = LᵀL̄ (lower blocks), T = Φ(P) W (lower), S = Wᵀ T (all blocks).  86 block
// products of 4 MFMA instructions at K = 64, 15 at K = 32, against ~3 K² FMAs per lane before.  (W is as accurate as L's condition
// number allows; the factors of the suites' matrices keep the pullback inside the flat 1e-3 / 1e-6 bars, measured in
// profiles/r06_vjp_errors.md.)
// Two K x K buffers per sample, odd pitch (every lane = row and lane = column access conflict-free):
//   U: X, then L (the factorisation in place, as before: right-looking, lane = row, the row in registers), then W below the diagonal
//      blocks; the strict upper triangle of a diagonal block holds that block of W transposed, its diagonal a small vector;
//   V: ȳ scattered to (row, column), then L̄ from the link (two sweeps along the row, in place; the running remainders parked in the
//      column above the diagonal), then Φ(P), T, S in turn — each product is collected in registers and stored when its reads are done.
// The cotangent leaves as Ā[i][j] = S[i][j] + S[j][i] on the triangle the reference reads, formed in the store.
// Blocks are persistent; both arrays of a group's next sample are requested as soon as the current ones are in LDS.
// Algorithmic bytes per sample as in bjx_matrix_vjp.hip.
#include <cstdlib>
#include <type_traits>

#include "bjx_internal.h"
#include "bjx_tile.h"
#include "bjx_matrix_vjp.h"

using namespace bjx;

namespace {

#define FW_UNROLL _Pragma("unroll")
__device__ __forceinline__ void fw_sync() { tile_sync(); __builtin_amdgcn_sched_barrier(0); }
// inside a phase: LDS reads after this point are not issued before it (left alone, every read of an unrolled phase is hoisted to its
// top — nothing between them writes — and the kernel lives on 500 registers and scratch)
__device__ __forceinline__ void fw_fence(int& seed) { asm volatile("" : "+v"(seed) : : "memory"); __builtin_amdgcn_sched_barrier(0); }
template <class T, int KMAX> __device__ __forceinline__ void fw_pin(T (&v)[KMAX]) {
  FW_UNROLL for (int j = 0; j < KMAX; ++j) asm volatile("" : "+v"(v[j]));
  __builtin_amdgcn_sched_barrier(0);
}

// LDS of one sample, in elements: U [KMAX][P] | V [KMAX][P] | column vector [KMAX] (16-byte aligned: broadcast reads of the
// factorisation) | diagonal of W [KMAX], P odd.  The groups of a wave sit GS banks apart.
template <class T, int GS, int KMAX> struct FwLds {
  static constexpr int P = KMAX + 1;
  static constexpr int BASE = 2 * KMAX * P + 2 * KMAX;
  static constexpr int W = sizeof(T) / 4;
  static constexpr int TARGET = sizeof(T) == 4 ? GS % 64 : (GS >= 32 ? 0 : 32);
  static constexpr int pad() { int q = 0; while (((BASE + q) * W) % 64 != TARGET || (BASE + q) % 4 != 0) ++q; return q; }
  static constexpr int SS = BASE + pad();
};

template <class T, int GS, int KMAX, int KIND, bool VEC, int NT>
__global__ __launch_bounds__(NT) void matrix_fwd_vjp_mfma_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                                 T* __restrict__ in_bar, int K, int64_t batch) {
  using M = VjpMath<T>;
  using O = VjpMfma<T>;
  using RV = typename O::V;
  using ACC = typename O::acc_t;
  constexpr int N = O::N, P = FwLds<T, GS, KMAX>::P, SPB = NT / GS, SPW = 64 / GS, SS = FwLds<T, GS, KMAX>::SS;
  constexpr int NIT = (KMAX * KMAX + GS - 1) / GS;
  constexpr int NITV = (NIT + N - 1) / N;
  constexpr int NREG = VEC ? NITV * N : NIT;
  constexpr int NB = (KMAX + 15) / 16;
  constexpr bool RAGGED = KMAX % 16 != 0;
  constexpr int LA = NB <= 2 ? 4 : 1, RB = LA + 1;           // operand sets of a block product in flight ahead of the MFMAs (one K-block of the contraction at up to 32 rows)
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  constexpr bool VECK = KIND == MK_VEC_CORR || KIND == MK_PD_VEC;
  constexpr int VW = VEC ? N : 1, NV = VEC ? NITV : NIT;
  constexpr int NFMAX = KIND == MK_VEC_CORR ? KMAX * (KMAX - 1) / 2 : (KIND == MK_PD_VEC ? KMAX * (KMAX + 1) / 2 : KMAX * KMAX);
  constexpr int NVB = (NFMAX + GS * VW - 1) / (GS * VW);
  static_assert(GS >= 16 * NB, "one lane per column of every diagonal block");
  extern __shared__ __align__(16) unsigned char smem_[];
  // (not const: the persistent loop below launders the lane-derived values once per sample — see there)
  int tl = threadIdx.x & (GS - 1), sl = threadIdx.x / GS;
  int t = tl < KMAX ? tl : KMAX - 1;
  int uoff = sl * SS;
  T* U = reinterpret_cast<T*>(smem_) + uoff;
  T* V = U + KMAX * P;
  T* colv = V + KMAX * P;
  unsigned short* tab = reinterpret_cast<unsigned short*>(reinterpret_cast<T*>(smem_) + (size_t)SPB * SS);
  const int64_t KK = (int64_t)K * K, nfree = free_len<KIND>(K);
  bool act = t < K;

  if constexpr (VECK) {
    for (int e = threadIdx.x; e < (int)nfree; e += NT) {
      int c;
      if (KIND == MK_VEC_CORR) {
        c = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
        while (c * (c - 1) / 2 > e) --c;
        while ((c + 1) * c / 2 <= e) ++c;
        tab[e] = (unsigned short)((c << 8) | (e - c * (c - 1) / 2));
      } else {
        c = (int)((__builtin_sqrtf(1.0f + 8.0f * (float)e) - 1.0f) * 0.5f);
        while (c * (c + 1) / 2 > e) --c;
        while ((c + 1) * (c + 2) / 2 <= e) ++c;
        tab[e] = (unsigned short)((c << 8) | (e - c * (c + 1) / 2));
      }
    }
  }
  // Outside the K x K corner V is zero and stays zero (staging writes the corner only; products of zero rows are zero); U is rewritten
  // whole by the factorisation (an identity block past K)
  for (int e = tl; e < SS; e += GS) U[e] = T(0);
  __syncthreads();

  const int e0 = tl * VW;
  int r0 = e0 / K, c0 = e0 - r0 * K;
  const int dr = (GS * VW) / K, dc = (GS * VW) - dr * K;
  auto issue = [&](const T* src, int64_t n, T (&v)[NREG], auto nv_) {
    constexpr int NV_ = decltype(nv_)::value;
    FW_UNROLL for (int it = 0; it < NV_; ++it) {
      const int e = (tl + it * GS) * VW;
      if constexpr (VEC) {
        const RV x = *reinterpret_cast<const RV*>(src + (e < n ? e : 0));
        FW_UNROLL for (int u = 0; u < N; ++u) v[it * N + u] = x[u];
      } else {
        v[it] = src[e < n ? e : 0];
      }
    }
  };
  // K x K row-major -> pitch P (TR: transposed).  A slot past the end re-writes element 0.
  auto commit_mat = [&](T* dstm, const T (&v)[NREG], auto tr_) {
    constexpr bool TR = decltype(tr_)::value;
    int r = r0, c = c0;
    FW_UNROLL for (int it = 0; it < NV; ++it) {
      const bool ok = r < K;
      const int rr = ok ? r : 0, cc = ok ? c : 0;
      FW_UNROLL for (int u = 0; u < VW; ++u) dstm[TR ? (cc + u) * P + rr : rr * P + cc + u] = v[it * VW + u];
      c += dc; r += dr;
      const bool wrap = c >= K;
      c = wrap ? c - K : c;
      r = wrap ? r + 1 : r;
    }
  };
  auto scatter_packed = [&](const T (&v)[NREG]) {          // ȳ(c, i) -> V[c][i]
    FW_UNROLL for (int it = 0; it < NVB; ++it) {
      const int e = (tl + it * GS) * VW;
      const int ee = e < (int)nfree ? e : 0;
      FW_UNROLL for (int u = 0; u < VW; ++u) {
        const int ci = tab[ee + u], c = ci >> 8, i = ci & 255;
        V[c * P + i] = v[it * VW + u];
      }
    }
  };
  // in_bar (r, c) from S in V: the sum of the two triangles on the one the reference reads, S[r][r] on the diagonal, zero on the other
  auto unstage_sym = [&](T* dst) {
    int r = r0, c = c0;
    FW_UNROLL for (int it = 0; it < NV; ++it) {
      const bool ok = r < K;
      const int rr = ok ? r : 0, cc = ok ? c : 0;
      T x[VW];
      FW_UNROLL for (int u = 0; u < VW; ++u) {
        const int c1 = cc + u;
        const T s1 = V[rr * P + c1], s2 = V[c1 * P + rr];
        const bool keep = CORR ? c1 < rr : c1 > rr;
        x[u] = keep ? s1 + s2 : (c1 == rr ? s1 : T(0));
      }
      if constexpr (VEC) {
        RV xv;
        FW_UNROLL for (int u = 0; u < N; ++u) xv[u] = x[u];
        *reinterpret_cast<RV*>(dst + rr * K + cc) = xv;
      } else {
        dst[rr * K + cc] = x[0];
      }
      c += dc; r += dr;
      const bool wrap = c >= K;
      c = wrap ? c - K : c;
      r = wrap ? r + 1 : r;
    }
  };

  const int64_t stride = (int64_t)gridDim.x * SPB;
  const int64_t n_b = VECK ? nfree : KK;
  int64_t s_raw = (int64_t)blockIdx.x * SPB + sl;
  int64_t w_raw = (int64_t)blockIdx.x * SPB + (threadIdx.x / 64) * SPW;
  T pa[NREG], pb[NREG], dl_next;
  {
    const int64_t s0 = s_raw < batch ? s_raw : batch - 1;
    issue(in + s0 * KK, KK, pa, std::integral_constant<int, NV>{});
    issue(out_bar + s0 * n_b, n_b, pb, std::integral_constant<int, NVB>{});
    dl_next = ladj_bar ? ladj_bar[s0] : T(0);
  }
  int lane = threadIdx.x & 63, mn = lane & 15, mq = lane >> 4;
  int woff = (threadIdx.x / 64) * SPW * SS;
  T* Uw = reinterpret_cast<T*>(smem_) + woff;
  unsigned te = act ? (unsigned)t : 0u;

  for (; w_raw < batch; w_raw += stride, s_raw += stride) {
    // Everything below that depends only on the lane — a hundred LDS addresses, masks, (row, column) pairs of the staging rounds — is
    // loop-invariant, and the compiler hoists ALL of it out of the persistent loop and keeps it in registers for the whole kernel
    // (256 VGPRs + 250 AGPRs + scratch).  Laundering the seeds once per sample makes it recompute them where they are used.
    asm volatile("" : "+v"(tl), "+v"(t), "+v"(uoff), "+v"(woff), "+v"(r0), "+v"(c0), "+v"(mn), "+v"(mq), "+v"(lane));
    U = reinterpret_cast<T*>(smem_) + uoff;
    V = U + KMAX * P;
    colv = V + KMAX * P;
    Uw = reinterpret_cast<T*>(smem_) + woff;
    act = t < K;
    te = act ? (unsigned)t : 0u;
    const bool live = s_raw < batch;
    const int64_t s = live ? s_raw : batch - 1;
    const int64_t sn = s_raw + stride < batch ? s_raw + stride : batch - 1;
    const T dl = dl_next;
    dl_next = ladj_bar ? ladj_bar[sn] : T(0);
    // ---- F1 / F3: X -> U, ȳ -> V; the next sample's arrays leave at once
    commit_mat(U, pa, std::false_type{});
    if constexpr (VECK) scatter_packed(pb);
    else if constexpr (KIND == MK_PD) commit_mat(V, pb, std::true_type{});     // memory index i K + c holds ȳ(c, i)
    else commit_mat(V, pb, std::false_type{});
    fw_sync();
    // ---- F2: right-looking Cholesky, lane t keeps row t in registers; rows and columns K .. KMAX-1 are an identity block.
    // One LDS round trip per pivot: every lane publishes its UNSCALED entry of column k, reads the pivot d = A[k][k] and the column back,
    // and updates with a[j] -= (a[k] / d) colv[j] (the scaled column is L[j][k] = colv[j] / √d: never needed in this step).
    T a[KMAX];
    T dcc;
    {
      FW_UNROLL for (int j = 0; j < KMAX; ++j) {
        const T raw = CORR ? U[t * P + j] : U[j * P + t];
        a[j] = (act && j <= t) ? raw : ((!act && j == t) ? T(1) : T(0));
      }
      fw_sync();
      FW_UNROLL for (int k = 0; k < KMAX; ++k) {
        colv[t] = a[k];                                          // (lanes t < k publish dead values: masked below)
        fw_sync();
        const T d = colv[k];
        T rs, sq;
        M::pivot(d, rs, sq);
        const T akr = t > k ? -(a[k] * rs) * rs : T(0);          // rows at or above the pivot do not change
        FW_UNROLL for (int j0 = ((k + 1) / N) * N; j0 < KMAX; j0 += N) {
          const RV x = *reinterpret_cast<const RV*>(colv + j0);
          FW_UNROLL for (int u = 0; u < N; ++u) if (j0 + u > k) a[j0 + u] += akr * x[u];
        }
        a[k] = t == k ? sq : (t > k ? a[k] * rs : a[k]);
        U[t * P + k] = a[k];                                    // (lanes t < k write dead storage above the diagonal)
        fw_sync();
        if (k % 2 == 1) fw_pin(a);
      }
      dcc = T(1);
      FW_UNROLL for (int j = 0; j < KMAX; ++j) dcc = j == t ? a[j] : dcc;
    }
    fw_sync();
    // ---- F4: cotangent of row t of the factor from the link, in place in V[t][.] (row t of L still in registers)
    if constexpr (CORR) {
      T g[KMAX];                                               // remainder before each entry, from the right
      T rem = dcc * dcc;
      FW_UNROLL for (int i = KMAX - 1; i >= 0; --i) {
        g[i] = rem;
        rem = (unsigned)i < te ? rem + a[i] * a[i] : rem;
      }
      T yv[KMAX];
      FW_UNROLL for (int m = 0; m < KMAX; ++m) yv[m] = V[t * P + m];
      T gsum = T(0);
      FW_UNROLL for (int m = 0; m < KMAX; ++m) {
        const bool on = (unsigned)m < te;
        const T w = a[m], yb = yv[m], R = g[m], wt = T(K - m) * dl;
        T gm, dsum;
        if (KIND == MK_VEC_CORR && m == 0) {
          gm = (yb + wt * w) * M::rcp(T(1) - w * w);
          dsum = T(0);
        } else {
          const T S2 = R + w * w;
          T rS, sqS;
          M::pivot(S2, rS, sqS);
          const T rS2 = rS * rS, rR = M::rcp(R);
          const T tt = yb * rS + wt * w * rS2;
          gm = tt + (w + w) * gsum;
          dsum = T(0.5) * rR * w * tt;
        }
        yv[m] = on ? gm : T(0);
        gsum = on ? gsum - dsum : gsum;
      }
      FW_UNROLL for (int m = 0; m < KMAX; ++m) yv[m] = m == t ? (dcc + dcc) * gsum : yv[m];
      FW_UNROLL for (int m = 0; m < KMAX; ++m) V[t * P + m] = yv[m];       // L̄ row t, zero right of the diagonal
    } else {
      const T raw = V[t * P + t];
      V[t * P + t] = act ? (raw - dl * T(K + 1 - t)) * M::rcp(dcc) : T(0);
    }
    fw_sync();
    // the next sample's arrays leave now: they travel during the block algebra, whose accumulators need few registers (requested before
    // the factorisation they would sit on 48 - 96 registers through the two phases that keep whole rows in registers)
    issue(in + sn * KK, KK, pa, std::integral_constant<int, NV>{});
    issue(out_bar + sn * n_b, n_b, pb, std::integral_constant<int, NVB>{});
    // ---- the block algebra: the whole wave on one sample at a time (rolled loops over the wave's samples).  Operand element of a
    // logical matrix at (16 br + i, 16 bc + c), masks only in the diagonal blocks; inside a product k = 4 ks + mq is the index of the
    // contraction in its 16-block and mn the free index of the operand.  The operands of step n + 1 are read before the MFMAs of
    // step n are issued (two register sets): at one or two waves per SIMD nothing else covers the LDS round trip.
    auto inb = [&](int base, int x) -> bool { return !RAGGED || base + 15 < KMAX || base + x < KMAX; };
    // L (lower, from U)
    auto Lel = [&](const T* Uj, int br, int bc, int i, int c) -> T {
      const bool ok = inb(16 * br, i) && inb(16 * bc, c);
      const T x = Uj[(ok ? 16 * br + i : 0) * P + (ok ? 16 * bc + c : 0)];
      return (ok && (br > bc || i >= c)) ? x : T(0);
    };
    // W = L⁻¹ (lower): below the diagonal blocks in U; a diagonal block transposed in the strict upper triangle of U's, its diagonal in wj
    auto Wel = [&](const T* Uj, int br, int bc, int i, int c) -> T {
      const bool ok = inb(16 * br, i) && inb(16 * bc, c);
      if (br > bc) {
        const T x = Uj[(ok ? 16 * br + i : 0) * P + (ok ? 16 * bc + c : 0)];
        return ok ? x : T(0);
      }
      const int rr = ok ? 16 * br + i : 0, cc = ok ? 16 * bc + c : 0;
      const T lo = Uj[cc * P + rr], dg = Uj[2 * KMAX * P + KMAX + rr];
      return (ok && i > c) ? lo : ((ok && i == c) ? dg : T(0));
    };
    // plain element of a buffer (Q, Φ, T: zeros where they are zero); tri: L̄ — the triangle above the diagonal holds other things
    auto Pel = [&](const T* Xj, int br, int bc, int i, int c, bool tri) -> T {
      const bool ok = inb(16 * br, i) && inb(16 * bc, c);
      const T x = Xj[(ok ? 16 * br + i : 0) * P + (ok ? 16 * bc + c : 0)];
      return (ok && (!tri || br > bc || i >= c)) ? x : T(0);
    };
    auto store_blk = [&](T* dstm, int br, int bc, const ACC& d, T scale) {
      FW_UNROLL for (int r = 0; r < 4; ++r) {
        const int i = O::row(mq, r);
        if (inb(16 * br, i) && inb(16 * bc, mn)) dstm[(16 * br + i) * P + 16 * bc + mn] = scale * d[r];
      }
    };
    const ACC zero = ACC{T(0), T(0), T(0), T(0)};
    // ---- M1: P = L' L̄, blocks on and below the diagonal; Φ: lower triangle, half the diagonal -> V
    _Pragma("unroll 1") for (int j = 0; j < SPW; ++j) {
      const T* Uj = Uw + (size_t)j * SS;
      T* Vj = const_cast<T*>(Uj) + KMAX * P;
      ACC d[NB][NB];
      FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj <= bi; ++bj) d[bi][bj] = zero;
      T a[RB][NB], b[RB][NB];
      auto ld = [&](int st, T (&aa)[NB], T (&bb)[NB]) {
        const int m = st / 4, ks = st % 4;
        FW_UNROLL for (int bi = 0; bi < NB; ++bi) if (bi <= m) aa[bi] = Lel(Uj, m, bi, 4 * ks + mq, mn);              // A[i][k] = L[16 m + k][16 bi + i]
        FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj <= m) bb[bj] = Pel(Vj, m, bj, 4 * ks + mq, mn, true);        // B[k][j] = L̄[16 m + k][16 bj + j]
      };
      FW_UNROLL for (int p0 = 0; p0 < LA; ++p0) if (p0 < 4 * NB) ld(p0, a[p0 % RB], b[p0 % RB]);
      FW_UNROLL for (int st = 0; st < 4 * NB; ++st) {
        if (st + LA < 4 * NB) ld(st + LA, a[(st + LA) % RB], b[(st + LA) % RB]);
        fw_fence(mn);
        const int m = st / 4;
        FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj < NB; ++bj)
          if (bi <= m && bj <= bi) d[bi][bj] = O::mfma(a[st % RB][bi], b[st % RB][bj], d[bi][bj]);
        fw_fence(mn);
      }
      fw_sync();
      FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj <= bi; ++bj) {
        if (bi == bj) {
          FW_UNROLL for (int r = 0; r < 4; ++r) {
            const int i = O::row(mq, r);
            d[bi][bj][r] = i > mn ? d[bi][bj][r] : (i == mn ? T(0.5) * d[bi][bj][r] : T(0));
          }
        }
        store_blk(Vj, bi, bj, d[bi][bj], T(1));
      }
    }
    fw_sync();
    // ---- M2: the diagonal blocks of W by forward substitution, every sample of the wave at once: 16 lanes = (sample, block), lane =
    // column c of the block: w[r] = (δ_rc − Σ_{m<r} L[r][m] w[m]) / L[r][r]
    {
      constexpr int NBP = GS / 16;                           // 16-lane slots of a sample
      const int lb = lane >> 4, js = lb / NBP, bb = lb % NBP;
      const bool mine = bb < NB;
      int b16 = mine ? 16 * bb : 0;
      T* Uj = Uw + (size_t)js * SS;
      T w[16];
      FW_UNROLL for (int r = 0; r < 16; ++r) {
        // (row r is read when w[r-4] exists: without a tie all 120 reads of the block are issued up front, on 120 registers; tied to
        // w[r-1] every row waits out a whole LDS round trip)
        if (r >= 4) asm volatile("" : "+v"(b16) : "v"(w[r - 4]));   // (four rows of look-ahead)
        const bool rin = !RAGGED || b16 + r < KMAX;
        const int rr = rin ? b16 + r : 0;
        T acc = r == mn ? T(1) : T(0);
        FW_UNROLL for (int m = 0; m < r; ++m) acc -= Uj[rr * P + b16 + m] * w[m];
        const T rd = M::rcp(Uj[rr * P + rr]);
        w[r] = rin ? acc * rd : (r == mn ? T(1) : T(0));
      }
      fw_sync();
      FW_UNROLL for (int r = 0; r < 16; ++r) {
        const bool rin = mine && (!RAGGED || (b16 + r < KMAX && b16 + mn < KMAX));
        if (rin && r > mn) Uj[(b16 + mn) * P + b16 + r] = w[r];
        if (rin && r == mn) Uj[2 * KMAX * P + KMAX + b16 + mn] = w[r];
      }
    }
    fw_sync();
    _Pragma("unroll 1") for (int j = 0; j < SPW; ++j) {
      T* Uj = Uw + (size_t)j * SS;
      T* Vj = Uj + KMAX * P;
      // ---- M3: W below the diagonal blocks, top block row first: Q = Σ_{bj<=m<bi} L(bi,m) W(m,bj) over L(bi,bj), then −W(bi,bi) Q over Q
      FW_UNROLL for (int bi = 1; bi < NB; ++bi) {
        ACC q[NB];
        FW_UNROLL for (int bj = 0; bj < NB; ++bj) q[bj] = zero;
        {
          T a[RB], b[RB][NB];
          auto ld = [&](int st, T& aa, T (&bb)[NB]) {
            const int m = st / 4, ks = st % 4;
            aa = Lel(Uj, bi, m, mn, 4 * ks + mq);                                                                    // A[i][k] = L[16 bi + i][16 m + k]
            FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj <= m) bb[bj] = Wel(Uj, m, bj, 4 * ks + mq, mn);           // B[k][j] = W[16 m + k][16 bj + j]
          };
          FW_UNROLL for (int p0 = 0; p0 < LA; ++p0) if (p0 < 4 * bi) ld(p0, a[p0 % RB], b[p0 % RB]);
          FW_UNROLL for (int st = 0; st < 4 * bi; ++st) {
            if (st + LA < 4 * bi) ld(st + LA, a[(st + LA) % RB], b[(st + LA) % RB]);
            fw_fence(mn);
            const int m = st / 4;
            FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj <= m) q[bj] = O::mfma(a[st % RB], b[st % RB][bj], q[bj]);
            fw_fence(mn);
          }
        }
        fw_sync();
        FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj < bi) store_blk(Uj, bi, bj, q[bj], T(1));
        fw_sync();
        FW_UNROLL for (int bj = 0; bj < NB; ++bj) q[bj] = zero;
        FW_UNROLL for (int ks = 0; ks < 4; ++ks) {
          const T a = Wel(Uj, bi, bi, mn, 4 * ks + mq);                                                               // A[i][k] = W[16 bi + i][16 bi + k]
          FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj < bi) q[bj] = O::mfma(a, Pel(Uj, bi, bj, 4 * ks + mq, mn, false), q[bj]);   // B[k][j] = Q[16 bi + k][16 bj + j]
        }
        fw_sync();
        FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj < bi) store_blk(Uj, bi, bj, q[bj], T(-1));
        fw_sync();
      }
      // ---- M4: T = Φ W, blocks on and below the diagonal -> V
      {
        ACC d[NB][NB];
        FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj <= bi; ++bj) d[bi][bj] = zero;
        T a[RB][NB], b[RB][NB];
        auto ld = [&](int st, T (&aa)[NB], T (&bb)[NB]) {
          const int m = st / 4, ks = st % 4;
          FW_UNROLL for (int bi = 0; bi < NB; ++bi) if (bi >= m) aa[bi] = Pel(Vj, bi, m, mn, 4 * ks + mq, false);      // A[i][k] = Φ[16 bi + i][16 m + k]
          FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj <= m) bb[bj] = Wel(Uj, m, bj, 4 * ks + mq, mn);             // B[k][j] = W[16 m + k][16 bj + j]
        };
        FW_UNROLL for (int p0 = 0; p0 < LA; ++p0) if (p0 < 4 * NB) ld(p0, a[p0 % RB], b[p0 % RB]);
        FW_UNROLL for (int st = 0; st < 4 * NB; ++st) {
          if (st + LA < 4 * NB) ld(st + LA, a[(st + LA) % RB], b[(st + LA) % RB]);
          fw_fence(mn);
          const int m = st / 4;
          FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj < NB; ++bj)
            if (bi >= m && bj <= m) d[bi][bj] = O::mfma(a[st % RB][bi], b[st % RB][bj], d[bi][bj]);
          fw_fence(mn);
        }
        fw_sync();
        FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj <= bi; ++bj) store_blk(Vj, bi, bj, d[bi][bj], T(1));
      }
      fw_sync();
      // ---- M5: S = W' T, every block -> V
      {
        ACC d[NB][NB];
        FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj < NB; ++bj) d[bi][bj] = zero;
        T a[RB][NB], b[RB][NB];
        auto ld = [&](int st, T (&aa)[NB], T (&bb)[NB]) {
          const int m = st / 4, ks = st % 4;
          FW_UNROLL for (int bi = 0; bi < NB; ++bi) if (bi <= m) aa[bi] = Wel(Uj, m, bi, 4 * ks + mq, mn);             // A[i][k] = W[16 m + k][16 bi + i]
          FW_UNROLL for (int bj = 0; bj < NB; ++bj) if (bj <= m) bb[bj] = Pel(Vj, m, bj, 4 * ks + mq, mn, false);      // B[k][j] = T[16 m + k][16 bj + j]
        };
        FW_UNROLL for (int p0 = 0; p0 < LA; ++p0) if (p0 < 4 * NB) ld(p0, a[p0 % RB], b[p0 % RB]);
        FW_UNROLL for (int st = 0; st < 4 * NB; ++st) {
          if (st + LA < 4 * NB) ld(st + LA, a[(st + LA) % RB], b[(st + LA) % RB]);
          fw_fence(mn);
          const int m = st / 4;
          FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj < NB; ++bj)
            if (bi <= m && bj <= m) d[bi][bj] = O::mfma(a[st % RB][bi], b[st % RB][bj], d[bi][bj]);
          fw_fence(mn);
        }
        fw_sync();
        FW_UNROLL for (int bi = 0; bi < NB; ++bi) FW_UNROLL for (int bj = 0; bj < NB; ++bj) store_blk(Vj, bi, bj, d[bi][bj], T(1));
      }
      fw_sync();
    }
    fw_sync();
    // ---- F9
    if (live) unstage_sym(in_bar + s * KK);
    fw_sync();
  }
}

template <class T, int GS, int KMAX, int KIND, bool VEC, int NT>
void fw_launch_one(bjx_ctx* ctx, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  constexpr int SPB = NT / GS;
  const size_t smem = (size_t)SPB * FwLds<T, GS, KMAX>::SS * sizeof(T) + ((size_t)KMAX * (KMAX + 1) / 2) * sizeof(unsigned short) + 16;
  auto kern = matrix_fwd_vjp_mfma_kernel<T, GS, KMAX, KIND, VEC, NT>;
  static int per_cu = 0;                                  // persistent blocks: as many as are resident at once
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
int fw_launch(bjx_ctx* ctx, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  constexpr int N = VjpMfma<T>::N;
  const bool vec = K % N == 0 && free_len<KIND>(K) % N == 0 && bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
  if (vec) fw_launch_one<T, GS, KMAX, KIND, true, NT>(ctx, in, out_bar, ladj_bar, in_bar, K, batch);
  else fw_launch_one<T, GS, KMAX, KIND, false, NT>(ctx, in, out_bar, ladj_bar, in_bar, K, batch);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T>
int fw_kind(bjx_ctx* ctx, int kind, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  if (K > 32) {
    // 33 .. 64 rows: the whole wave on one sample.  Float64 at 49 .. 64 rows: 68 KiB of LDS a sample — blocks of two waves (two samples a CU)
    constexpr int NT64 = sizeof(T) == 4 ? 256 : 128;
#define FW_W(KIND_) (K <= 48 ? fw_launch<T, 64, 48, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                             : fw_launch<T, 64, 64, KIND_, NT64>(ctx, in, out_bar, ladj_bar, in_bar, K, batch))
    switch (kind) {
      case MK_VEC_CORR: return FW_W(MK_VEC_CORR);
#ifndef FW_DEV
      case MK_CORR: return FW_W(MK_CORR);
      case MK_PD: return FW_W(MK_PD);
#endif
      default: return FW_W(MK_PD_VEC);
    }
#undef FW_W
  }
#define FW_K(KIND_) (K <= 12 ? fw_launch<T, 16, 12, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                   : K <= 16 ? fw_launch<T, 16, 16, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                   : K <= 24 ? fw_launch<T, 32, 24, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch) \
                             : fw_launch<T, 32, 32, KIND_>(ctx, in, out_bar, ladj_bar, in_bar, K, batch))
  switch (kind) {
    case MK_VEC_CORR: return FW_K(MK_VEC_CORR);
#ifndef FW_DEV
    case MK_CORR: return FW_K(MK_CORR);
    case MK_PD: return FW_K(MK_PD);
#endif
    default: return FW_K(MK_PD_VEC);
  }
#undef FW_K
}

}  // namespace

namespace bjx {

// 1: not served (the caller goes on to the lane = row group kernel).  Served where it is the faster one on the same box (2^14 .. 2^19
// samples, Float32, percent of the HBM peak, group kernel -> this one, VecCorr / PDVec): K = 16: 31 / 33 -> 32 / 36, 32: 19 / 20 -> 25 / 28,
// 48: 12 / 13 -> 13 / 23, 64: 7 / 7 -> 14 / 16; NOT at K <= 12 (30 / 33 -> 16 / 18: a 16 x 16 block algebra on a 12 x 12 problem, four
// samples a wave one after the other) nor at 17 .. 24 (22 / 24 -> 15 / 21).  Float64 (call times of scripts/probe_matrix_vjp.py, group -> this): K = 32 0.71 / 0.61 -> 0.50 / 0.42 ms, K = 16 the same, 12 and 24 slower: from 25 rows.  Float64 at 33 .. 64 rows had only the one-lane workspace kernel: 18.6 / 47.7 ms -> 0.43 / 2.0 ms at K = 48 / 64, 2^13 samples.
int bjx_matrix_fwd_vjp_mfma(bjx_ctx* ctx, bjx_dtype dt, int kind, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  static const int use = getenv("BJX_MATRIX_VJP_MFMA") ? atoi(getenv("BJX_MATRIX_VJP_MFMA")) : 1;      // 0: the lane = row group kernel (its A/B); 2: every shape this kernel can do
#ifdef FW_DEV
  if (kind == MK_CORR || kind == MK_PD) return 1;
#endif
  if (!use || K < 9 || K > 64) return 1;
  if (use != 2 && (K < 13 || (K > 16 && K < 25) || (dt != BJX_F32 && K < 25))) return 1;      // (K > 32 in Float64: nothing else but the workspace kernel)
  if (dt == BJX_F32) return fw_kind<float>(ctx, kind, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
#ifdef FW_DEV
  return 1;
#else
  return fw_kind<double>(ctx, kind, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
#endif
}

}  // namespace bjx
