// bjx_matrix.hip — SURVEY.md §8(f) f-4: the matrix-variate constraint bijectors that need a per-sample Cholesky
// factorisation, batched over samples:
//   VecCorrBijector  corr.jl:128-162   X[K,K] correlation matrix <-> y[K(K-1)/2]
//   CorrBijector     corr.jl:64-92     X[K,K]                    <-> Y[K,K] (strict upper triangle, zeros elsewhere)
//   PDBijector       pd.jl:1-36        X[K,K] positive definite  <-> Y[K,K] (lower factor with log diagonal)
//   PDVecBijector    pd.jl:38-60       X[K,K]                    <-> y[K(K+1)/2]
// and Scale with a MATRIX parameter (scale.jl:14,17,35-36): y = a*x, x = a\y, logabsdetjac = logabsdet(a).
//
// Mapping on gfx950 (matrix_link_kernel, documented at the kernel): GS = 16 / 32 / 64 lanes own one sample, lane i keeps
// row i of the lower Cholesky factor in registers, the pivot column / the rows of L travel as broadcast 16-byte LDS reads,
// the LKJ link is a per-lane loop over one LDS row, loads and stores are coalesced through the same LDS tile.
// Algorithmic bytes per sample: K*K*sizeof(T) on the matrix side (the reference materialises dense matrices) +
// the packed / dense unconstrained side + sizeof(T) for the per-sample log-det.  The factorisation is O(K^3/3) flop on
// O(K^2) bytes (K/12 flop per byte in Float32): HBM-bound on paper, VALU/LDS-issue-bound in practice.
#include <cstdlib>

#include "bjx_internal.h"
#include "bjx_tile.h"

using namespace bjx;

namespace {

enum { MK_VEC_CORR = 0, MK_CORR = 1, MK_PD = 2, MK_PD_VEC = 3 };

__device__ __forceinline__ float rdlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double rdlane(double v, int l) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

#include "bjx_linkmath.h"

// Layout of the contiguous run of one sample in global memory; every layout lands in the tile as tile[c*pitch + r]:
//   L_DENSE  K x K column-major          column c has K entries
//   L_TRIU1  triu1_to_vec (src/utils.jl:99-108): strict upper triangle, column-major: column c has c entries (c >= 1)
//   L_TRIU   triu_to_vec  (src/utils.jl:137-142): upper triangle with the diagonal:   column c has c + 1 entries
enum { L_DENSE = 0, L_TRIU1 = 1, L_TRIU = 2 };
template <int LAY> __device__ __forceinline__ int col_len(int c, int K) { return LAY == L_DENSE ? K : (LAY == L_TRIU1 ? c : c + 1); }
template <int LAY> __device__ __forceinline__ void decode(int e, int K, int& c, int& r) {
  if (LAY == L_DENSE) { c = e / K; r = e - c * K; return; }
  // first column whose run contains flat index e (float estimate, corrected exactly)
  int cc = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) + (LAY == L_TRIU1 ? 1.0f : -1.0f)) * 0.5f);
  auto start = [](int col) { return LAY == L_TRIU1 ? col * (col - 1) / 2 : col * (col + 1) / 2; };
  while (start(cc) > e) --cc;
  while (start(cc + 1) <= e) ++cc;
  c = cc; r = e - start(cc);
}
template <int LAY> __device__ __forceinline__ void advance(int step, int K, int& c, int& r) {
  r += step;
  while (r >= col_len<LAY>(c, K)) { r -= col_len<LAY>(c, K); ++c; }
}
// coalesced copy of the `cnt` contiguous elements of one sample between global memory and the LDS tile
template <class T, int LAY> __device__ __forceinline__ void stage_in(const T* __restrict__ g, T* tile, int cnt, int K, int pitch, int lane, bool vec_ok) {
  constexpr int VW = Vec16<T>::N;
  const int W = vec_ok ? VW : 1;
  int e = lane * W, c = 0, r = 0;
  if (e < cnt) decode<LAY>(e, K, c, r);
  for (; e < cnt; e += 64 * W) {
    if (vec_ok) {
      const Pack<T, VW> p = load_pack<T, VW, true>(g + e);
      int cc = c, rr = r;
#pragma unroll
      for (int t = 0; t < VW; ++t) {
        tile[cc * pitch + rr] = p.v[t];
        advance<LAY>(1, K, cc, rr);
      }
    } else tile[c * pitch + r] = __builtin_nontemporal_load(g + e);
    if (e + 64 * W < cnt) advance<LAY>(64 * W, K, c, r);
  }
}
template <class T, int LAY> __device__ __forceinline__ void stage_out(T* __restrict__ g, const T* tile, int cnt, int K, int pitch, int lane, bool vec_ok) {
  constexpr int VW = Vec16<T>::N;
  const int W = vec_ok ? VW : 1;
  int e = lane * W, c = 0, r = 0;
  if (e < cnt) decode<LAY>(e, K, c, r);
  for (; e < cnt; e += 64 * W) {
    if (vec_ok) {
      Pack<T, VW> p;
      int cc = c, rr = r;
#pragma unroll
      for (int t = 0; t < VW; ++t) {
        p.v[t] = tile[cc * pitch + rr];
        advance<LAY>(1, K, cc, rr);
      }
      store_pack<T, VW, true>(g + e, p);
    } else __builtin_nontemporal_store(tile[c * pitch + r], g + e);
    if (e + 64 * W < cnt) advance<LAY>(64 * W, K, c, r);
  }
}

// 16-byte LDS read of 4 consecutive T (two 16-byte reads for Float64)
template <class T> struct Quad { T v[4]; };
template <class T> __device__ __forceinline__ Quad<T> lds_quad(const T* p) {
  Quad<T> r;
  if constexpr (sizeof(T) == 4) {
    const bjx_f32x4 t = *reinterpret_cast<const bjx_f32x4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    const bjx_f64x2 t0 = *reinterpret_cast<const bjx_f64x2*>(p), t1 = *reinterpret_cast<const bjx_f64x2*>(p + 2);
    r.v[0] = t0.x; r.v[1] = t0.y; r.v[2] = t1.x; r.v[3] = t1.y;
  }
  return r;
}
// two consecutive elements as one operand: Float32 pairs become v_pk_fma_f32 (both halves of an aligned register pair),
// Float64 pairs are two v_fma_f64.  (SLP vectorisation is off for this file: it builds such pairs out of unrelated
// registers and pays for them with v_mov — 600 of them in the K = 64 kernel.)
template <class T> struct V2 { typedef T t __attribute__((ext_vector_type(2))); };
template <class T> struct QuadP { typename V2<T>::t lo, hi; };
template <class T> __device__ __forceinline__ QuadP<T> lds_quadp(const T* p) {
  QuadP<T> r;
  if constexpr (sizeof(T) == 4) {
    const bjx_f32x4 t = *reinterpret_cast<const bjx_f32x4*>(p);
    r.lo = __builtin_shufflevector(t, t, 0, 1);
    r.hi = __builtin_shufflevector(t, t, 2, 3);
  } else {
    r.lo = *reinterpret_cast<const bjx_f64x2*>(p);
    r.hi = *reinterpret_cast<const bjx_f64x2*>(p + 2);
  }
  return r;
}
template <class T> struct FacMath;                  // 1/d, 1/sqrt(d), sqrt(d) of a pivot: hardware units in Float32 (parity bar 1e-3)
template <> struct FacMath<float> {
  static __device__ __forceinline__ void pivot(float d, float& rd, float& rs, float& sq) { rs = Fast<float>::rsqrt(d); rd = rs * rs; sq = d * rs; }
};
template <> struct FacMath<double> {
  static __device__ __forceinline__ void pivot(double d, double& rd, double& rs, double& sq) { sq = ::sqrt(d); rs = 1.0 / sq; rd = 1.0 / d; }
};

// GS = 8 / 16 / 32 / 64 lanes = one sample (K <= GS), 64/GS samples per wave; block = 1 wave; the grid walks the batch.
//   lane (g = lane / GS, i = lane % GS) keeps row i of the lower factor L of sample g in REGISTERS (a[0..GS), fully
//   unrolled: every register index is a compile-time constant); the sample's matrix is staged in an LDS tile
//   tile[c*pitch + r] = M[r, c] (pitch a multiple of 4: rows of the tile are 16-byte aligned).
//   factor (forward): right-looking.  Step k: every lane writes its A[i][k] to a K-entry column buffer; the pivot and
//     the multipliers A[j][k] come back as WAVE-UNIFORM-PER-SAMPLE (broadcast) 16-byte LDS reads, four multipliers per
//     read: K^2/8 LDS reads + K^2/2 FMAs per sample (a[j] -= (A[i][k]/d) * A[j][k]), pivot math on the hardware units.
//     (v1 of this kernel fetched every multiplier with v_readlane: one sample per wave, 2 issue slots + hazard nops per
//     FMA, lanes >= K idle: 10 % of the HBM roofline at K = 32.)
//   link: rolled per-lane loops over the lane's tile row (column of U = L'), corr.jl:277-297, :314-335, :345-399.
//   X = L L' (inverse): row j of L is read from the tile with broadcast 16-byte reads, dotted with the lane's registers.
template <class T, int GS, int KIND, bool INV>
__global__ __launch_bounds__(64) void matrix_link_kernel(const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps, int K0, int pitch,
                                                         int64_t batch, int accumulate, int vec_in, int vec_out, double* partials) {
  extern __shared__ __align__(16) unsigned char smem_[];
  __shared__ double red[1];
  using M = LinkMath<T>;
  constexpr int KP = GS, NSW = 64 / GS;
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  constexpr int LAY = KIND == MK_VEC_CORR ? L_TRIU1 : (KIND == MK_PD_VEC ? L_TRIU : L_DENSE);   // layout of the unconstrained side
  const int lane0 = threadIdx.x;
  int K = K0;
  const int KK = K * K;
  const int nv = KIND == MK_VEC_CORR ? K * (K - 1) / 2 : (KIND == MK_PD_VEC ? K * (K + 1) / 2 : KK);   // elements on the unconstrained side
  const int tile_words = GS * pitch;                              // GS rows: the unrolled loops touch rows >= K as scratch
  const int sample_words = tile_words + GS + 4;                  // tile + column buffer (kept 16-byte aligned)
  T* lds = reinterpret_cast<T*>(smem_);
  double acc = 0.0;
  for (int64_t s0 = (int64_t)blockIdx.x * NSW; s0 < batch; s0 += (int64_t)gridDim.x * NSW) {
    // The lane id and K are made opaque at every phase: the several hundred compare masks of the unrolled code below
    // (lane == k, j < K, ...) are loop-invariant; left alone they are all hoisted out of the sample loop / kept alive
    // across phases and spilled (measured: 670 SGPR spills + the register row in scratch).
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(K));
    const int g = lane / GS;
    int li = lane - g * GS;                                      // row of my sample
    const int nsamp = (batch - s0) < NSW ? (int)(batch - s0) : NSW;
    const bool live = li < K && g < nsamp;
    T* tile = lds + g * sample_words;
    T* colbuf = tile + tile_words;
    T* myrow = tile + (li < K ? li : 0) * pitch;                 // tile row li: column li of an upper-triangular matrix
    typename V2<T>::t a2[KP / 2];                                // row li of the lower factor L (= column li of U = L'), as register pairs
#define a(j_) a2[(j_) >> 1][(j_) & 1]
    T lsum = T(0);                                               // this lane's log-det terms
    if constexpr (!INV) {
      // ---------------------------------------------------------------- X -> Cholesky -> link
#pragma unroll
      for (int gg = 0; gg < NSW; ++gg)
        if (gg < nsamp) stage_in<T, L_DENSE>(in + (s0 + gg) * KK, lds + gg * sample_words, KK, K, pitch, lane, vec_in != 0);
      __builtin_amdgcn_wave_barrier();
      // A[i][j], j <= i, from the triangle the reference reads; identity padding outside K
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        T v = j == li ? T(1) : T(0);
        if (j < K && live) v = CORR ? myrow[j] : tile[j * pitch + li];   // X[j,i] (upper) | X[i,j] (lower)
        a(j) = v;
      }
      __builtin_amdgcn_wave_barrier();
      // right-looking Cholesky: row in registers, column k through the LDS column buffer
      asm volatile("" : "+v"(li));
      asm volatile("" : "+s"(K));
      // STRAIGHT-LINE code over all GS columns (the identity padding makes the steps k >= K no-ops): uniform `k < K`
      // guards turn every a(j) into a PHI across ~100 basic blocks and the register coalescer gives up — the GS = 32
      // kernel carried 2 250 register-to-register moves and 210 VGPRs with them
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        colbuf[li] = a(k);
        __builtin_amdgcn_wave_barrier();
        T rd, rs, sq;
        FacMath<T>::pivot(colbuf[k], rd, rs, sq);
        const T c = a(k) * rd;                                   // A[i][k] / d: a(j) -= c * A[j][k]
#pragma unroll
        for (int j4 = (k + 1) & ~3; j4 < KP; j4 += 4) {
          const QuadP<T> m = lds_quadp<T>(colbuf + j4);
          // pairs entirely right of column k are updated as pairs; the pair that contains column k only in its odd half
          if (j4 > k) a2[j4 >> 1] -= c * m.lo; else if (j4 + 1 > k) a(j4 + 1) -= c * m.lo[1];
          if (j4 + 2 > k) a2[(j4 >> 1) + 1] -= c * m.hi; else if (j4 + 3 > k) a(j4 + 3) -= c * m.hi[1];
        }
        a(k) = li == k ? sq : a(k) * rs;
        __builtin_amdgcn_wave_barrier();
      }
      asm volatile("" : "+v"(li));
      asm volatile("" : "+s"(K));
      if constexpr (CORR) {
        // column li of U = my registers -> my tile row; then corr.jl:277-297 / :314-335 walks it bottom-up (a rolled
        // loop over LDS: the per-entry asinh is too large to unroll 64 times); y overwrites w in place.
        // log-det = -_logabsdetjac_inv_corr(y) (:92, :135-137, :453-472): weight K - i + 1 for 1-based row i.
#pragma unroll
        for (int j = 0; j < KP; ++j)
          if (j < K && live) myrow[j] = a(j);
        __builtin_amdgcn_wave_barrier();
        const T dg = live ? myrow[li] : T(1);
        T rem, Lr;
        M::fwd_init(dg, rem, Lr);
        for (int i = K - 2; i >= (KIND == MK_VEC_CORR ? 1 : 0); --i) {
          const bool act = i < li && live;
          const T w = act ? myrow[i] : T(0);
          T y, lc;
          M::fwd_step(w, rem, Lr, y, lc);                                                  // inactive lanes: w = 0 leaves rem and L unchanged
          if (act) {
            lsum += T(K - i) * lc;
            myrow[i] = y;
          }
        }
        if (KIND == MK_VEC_CORR && K >= 2) {                                               // :322 atanh(W[1, j]) on row 1
          const bool act = 0 < li && live;
          T y, lc;
          M::atanh_lc(act ? myrow[0] : T(0), y, lc);
          if (act) { lsum += T(K) * lc; myrow[0] = y; }
        }
        if (KIND == MK_CORR) {                                                             // zero fill on and below the diagonal (:292-294)
          for (int i = live ? li : K; i < K; ++i) myrow[i] = T(0);
        }
      } else {
        // pd.jl:11,27-31,41: Y = replace_diag(log, L); log-det = -(sum_i (d+2-i) log L_ii + d log 2)
        T dg = T(1);
#pragma unroll
        for (int j = 0; j < KP; ++j) dg = j == li ? a(j) : dg;
        const T ld = M::log(dg);
        if (live) lsum = -(T(K + 1 - li) * ld + Num<T>::log2);
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          if (j < K && live) {
            const T v = j == li ? ld : (j < li ? a(j) : T(0));
            if (KIND == MK_PD) tile[j * pitch + li] = v;                                   // Y[li, j]
            else if (j <= li) myrow[j] = v;                                                // triu_to_vec(Y'): (Y')[r,c] = L[c,r], r <= c
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (out) {
#pragma unroll
        for (int gg = 0; gg < NSW; ++gg)
          if (gg < nsamp) stage_out<T, LAY>(out + (s0 + gg) * nv, lds + gg * sample_words, nv, K, pitch, lane, vec_out != 0);
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      // ---------------------------------------------------------------- inverse link -> L -> X = L L'
#pragma unroll
      for (int gg = 0; gg < NSW; ++gg)
        if (gg < nsamp) stage_in<T, LAY>(in + (s0 + gg) * nv, lds + gg * sample_words, nv, K, pitch, lane, vec_in != 0);
      __builtin_amdgcn_wave_barrier();
      if constexpr (CORR) {
        // corr.jl:345-399: column li of U top-down (rolled, in place in my tile row);
        // + sum_{j=2}^{K-1} (K-j) log U[j,j] (:77-79, :144-146), log U[j,j] = the final log_remainder of column j
        T lr = T(0), E;
        M::inv_init(E);
        for (int i = 0; i < K - 1; ++i) {
          const bool act = i < li && live;
          const T yv = act ? myrow[i] : T(0);
          T w, lc;
          M::inv_step(yv, E, w, lc);                                                       // inactive lanes: y = 0 -> sech = 1, lc = 0
          if (act) { myrow[i] = w; lr -= lc; lsum += lr; }
        }
        if (live) {
          myrow[li] = M::inv_diag(E, lr);
          lsum += lr + ((li >= 1 && li <= K - 2) ? T(K - 1 - li) * lr : T(0));
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          T v = j == li ? T(1) : T(0);
          if (j < K && live) v = j <= li ? myrow[j] : T(0);
          a(j) = v;
        }
      } else {
        // pd.jl:13-16,44-47: L = lower_triangular(replace_diag(exp, Y)); log-det = +(sum_i (d+2-i) Y_ii + d log 2) (interface.jl:278-281)
        T yd = T(0);
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          T v = j == li ? T(1) : T(0);
          if (j < K && live) {
            const T t = (j <= li) ? (KIND == MK_PD ? tile[j * pitch + li] : myrow[j]) : T(0);
            if (j == li) { yd = t; v = M::exp(t); } else v = t;
          }
          a(j) = v;
        }
        if (live) lsum = T(K + 1 - li) * yd + Num<T>::log2;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" : "+v"(li));
      asm volatile("" : "+s"(K));
      if (out) {
        // every lane's row of L into its tile row (the diagonal already exponentiated), rows are then read back as
        // broadcast 16-byte packs:  X[li][j] = sum_{m <= j} L[li][m] * L[j][m]  (my row has exact zeros past its diagonal)
#pragma unroll
        for (int j = 0; j < KP; ++j)
          if (j < K && live) myrow[j] = a(j);
        __builtin_amdgcn_wave_barrier();
        // straight-line over all GS rows like the factorisation (the tile has GS rows; rows >= K are never staged out)
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          typename V2<T>::t x2 = {T(0), T(0)}, x3 = {T(0), T(0)};     // two chains: back-to-back dependent v_pk_fma_f32 cost a wait state each
          T x = T(0);
#pragma unroll
          for (int m4 = 0; m4 <= j; m4 += 4) {
            const QuadP<T> r4 = lds_quadp<T>(tile + j * pitch + m4);
            if (m4 + 1 <= j) x2 += a2[m4 >> 1] * r4.lo; else x += a(m4) * r4.lo[0];
            if (m4 + 3 <= j) x3 += a2[(m4 >> 1) + 1] * r4.hi; else if (m4 + 2 <= j) x += a(m4 + 2) * r4.hi[0];
          }
          x2 += x3;
          x += x2[0] + x2[1];
          __builtin_amdgcn_wave_barrier();                       // every lane has read row j: it is dead now and takes column j of X
          if (live) tile[j * pitch + li] = x;                     // (rows j >= K of the GS-row tile are scratch)
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int gg = 0; gg < NSW; ++gg)
          if (gg < nsamp) stage_out<T, L_DENSE>(out + (s0 + gg) * KK, lds + gg * sample_words, KK, K, pitch, lane, vec_out != 0);
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (ladj_ps || partials) {
      // sum over the GS lanes of each sample (xor butterflies stay inside an aligned group of GS lanes)
      double l = (double)lsum;
#pragma unroll
      for (int off = 1; off < GS; off <<= 1) l += shfl_xor(l, off);
      if (li == 0 && g < nsamp) {
        const T lt = (T)l;
        if (ladj_ps) ladj_ps[s0 + g] = accumulate ? ladj_ps[s0 + g] + lt : lt;
        acc += (double)lt;
      }
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}
#undef a

// ------------------------------------------------------------------ small matrices: ONE LANE per sample (K <= 12)
// The LKJ / Wishart blocks of real models are 2x2 ... 8x8.  With lanes along the rows a wave holds 8 such samples and every
// column step is an LDS round trip for a handful of FMAs (K = 8: 11 % of the roofline).  Here a wave takes 64 consecutive
// samples — one contiguous run of the input and of the output, moved with 16-byte accesses through a [64][P odd] LDS tile
// — and lane t factors sample t entirely in its own registers: no cross-lane traffic, no LDS inside the arithmetic, fully
// unrolled to KMAX rows with wave-uniform guards (K is a launch constant).  Same pivot and link arithmetic as
// matrix_link_kernel (FacMath, LinkMath); the trailing update uses the scaled column (l_ik l_jk instead of a_ik a_jk / d).
// KX > 0 (= K, 2 ... 4): the sample is read and written by its lane as one TinyCol object (multi-dword accesses), no tile — a 2x2 or
// 3x3 block is 16-36 bytes and the staging, not the factorisation, was the cost (K = 2 / 3: 22-46 % of the HBM peak).
template <class T, int KMAX, int KIND, bool INV, int V, int KX = 0>
__global__ __launch_bounds__(64) void matrix_lane_kernel(const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps, int Krt, int P,
                                                         int64_t batch, int accumulate, double* partials) {
  extern __shared__ __align__(16) unsigned char smem_[];
  __shared__ double red[1];
  using M = LinkMath<T>;
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  T* tile = reinterpret_cast<T*>(smem_);
  const int lane = threadIdx.x;
  const int K = KX > 0 ? KX : Krt;
  const int KK = K * K;
  const int nv = KIND == MK_VEC_CORR ? K * (K - 1) / 2 : (KIND == MK_PD_VEC ? K * (K + 1) / 2 : KK);
  const int n_in = INV ? nv : KK, n_out = INV ? KK : nv;
  constexpr int KKX = KX * KX, NVX = KIND == MK_VEC_CORR ? KX * (KX - 1) / 2 : (KIND == MK_PD_VEC ? KX * (KX + 1) / 2 : KKX);
  constexpr int NIN = KX > 0 ? (INV ? NVX : KKX) : 1, NOUT = KX > 0 ? (INV ? KKX : NVX) : 1, NBUF = KX > 0 ? (KKX > NVX ? KKX : NVX) : 1;
  double acc = 0.0;
  for (int64_t s0 = (int64_t)blockIdx.x * 64; s0 < batch; s0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - s0) < 64 ? (batch - s0) : 64);
    T buf[NBUF];
    T* mine = tile + lane * P;
    if constexpr (KX > 0) {
      TinyCol<T, NIN> t{};
      if (lane < ncols) t = *reinterpret_cast<const TinyCol<T, NIN>*>(in + (s0 + lane) * NIN);
#pragma unroll
      for (int i = 0; i < NBUF; ++i) buf[i] = i < NIN ? t.v[i < NIN ? i : 0] : T(0);
      mine = buf;
    } else {
      if (n_in > 0) tile_stage_in<T, V>(tile, in + s0 * n_in, n_in, P, ncols, lane);
      tile_sync();
    }
    T L[KMAX][KMAX];                                              // lower factor, row-major; only j <= i is used
    T lsum = T(0);
    if constexpr (!INV) {
      // A[i][j], j <= i, from the triangle the reference reads (upper for the correlation bijectors, lower for PD)
#pragma unroll
      for (int i = 0; i < KMAX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i][j] = (i < K) ? (CORR ? mine[i * K + j] : mine[j * K + i]) : (i == j ? T(1) : T(0));
      // right-looking Cholesky (the update order of matrix_link_kernel)
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          T rd, rs, sq;
          FacMath<T>::pivot(L[k][k], rd, rs, sq);
          L[k][k] = sq;
#pragma unroll
          for (int i = k + 1; i < KMAX; ++i) if (i < K) L[i][k] *= rs;             // column k of the factor
#pragma unroll
          for (int i = k + 1; i < KMAX; ++i) {
            if (i < K) {
#pragma unroll
              for (int j = k + 1; j <= i; ++j) L[i][j] -= L[i][k] * L[j][k];        // trailing update with the scaled column
            }
          }
        }
      }
      tile_sync();                                                // (single wave: every lane has read its sample)
      if constexpr (CORR) {
        // column c of U = row c of L, bottom-up (corr.jl:277-297, :314-335); log-det weights K - i (0-based row i)
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
          if (c < K) {
            T rem, Lr;
            M::fwd_init(L[c][c], rem, Lr);
#pragma unroll
            for (int i = KMAX - 2; i >= (KIND == MK_VEC_CORR ? 1 : 0); --i) {
              if (i < c && i <= K - 2) {
                T y, lc;
                M::fwd_step(L[c][i], rem, Lr, y, lc);
                lsum += T(K - i) * lc;
                if (KIND == MK_VEC_CORR) mine[c * (c - 1) / 2 + i] = y; else mine[c * K + i] = y;
              }
            }
            if (KIND == MK_VEC_CORR && c >= 1) {                  // :322 atanh(W[1, j]) on the first row
              T y, lc;
              M::atanh_lc(L[c][0], y, lc);
              lsum += T(K) * lc;
              mine[c * (c - 1) / 2] = y;
            }
            if (KIND == MK_CORR) {                                // zeros on and below the diagonal (:292-294)
#pragma unroll
              for (int i = 0; i < KMAX; ++i) if (i >= c && i < K) mine[c * K + i] = T(0);
            }
          }
        }
      } else {
        // pd.jl:11,27-31,41: Y = replace_diag(log, L); log-det = -(sum_i (d+2-i) log L_ii + d log 2)
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
          if (i < K) {
            const T ld = M::log(L[i][i]);
            lsum -= T(K + 1 - i) * ld + Num<T>::log2;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
              if (j < K) {
                const T v = j == i ? ld : (j < i ? L[i][j] : T(0));
                if (KIND == MK_PD) mine[j * K + i] = v;                 // Y[i, j]
                else if (j <= i) mine[i * (i + 1) / 2 + j] = v;         // triu_to_vec(Y'): (Y')[r, c] = L[c][r], r <= c
              }
            }
          }
        }
      }
    } else {
      if constexpr (CORR) {
        // corr.jl:345-399: column c of U top-down; + sum_{j=2}^{K-1} (K-j) log U[j,j] (:77-79, :144-146)
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
          if (c < K) {
            T lr = T(0), E;
            M::inv_init(E);
#pragma unroll
            for (int i = 0; i < KMAX - 1; ++i) {
              if (i < c) {
                const T yv = KIND == MK_VEC_CORR ? mine[c * (c - 1) / 2 + i] : mine[c * K + i];
                T w, lc;
                M::inv_step(yv, E, w, lc);
                L[c][i] = w;
                lr -= lc;
                lsum += lr;
              }
            }
            L[c][c] = M::inv_diag(E, lr);
            lsum += lr + ((c >= 1 && c <= K - 2) ? T(K - 1 - c) * lr : T(0));
          }
        }
      } else {
        // pd.jl:13-16,44-47: L = lower_triangular(replace_diag(exp, Y)); log-det = +(sum_i (d+2-i) Y_ii + d log 2)
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
          if (i < K) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
              const T t = KIND == MK_PD ? mine[j * K + i] : mine[i * (i + 1) / 2 + j];
              if (j == i) { lsum += T(K + 1 - i) * t + Num<T>::log2; L[i][j] = M::exp(t); } else L[i][j] = t;
            }
          }
        }
      }
      tile_sync();
      if (out) {
        // X = L L' (both triangles written): X[i][j] = sum_{m <= min(i,j)} L[i][m] L[j][m]
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
          if (i < K) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
              T x = T(0);
#pragma unroll
              for (int m = 0; m <= j; ++m) x += L[i][m] * L[j][m];
              mine[j * K + i] = x;
              if (j != i) mine[i * K + j] = x;
            }
          }
        }
      }
    }
    if constexpr (KX > 0) {
      if (out && lane < ncols) {
        TinyCol<T, NOUT> o;
#pragma unroll
        for (int i = 0; i < NOUT; ++i) o.v[i] = buf[i];
        *reinterpret_cast<TinyCol<T, NOUT>*>(out + (s0 + lane) * NOUT) = o;
      }
    } else {
      tile_sync();
      if (out && n_out > 0) tile_stage_out<T, V>(tile, out + s0 * n_out, n_out, P, ncols, lane);
      tile_sync();
    }
    if (lane < ncols) {
      if (ladj_ps) ladj_ps[s0 + lane] = accumulate ? ladj_ps[s0 + lane] + lsum : lsum;
      acc += (double)lsum;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

template <class T, int KX, int KIND>
int launch_lane_direct(bjx_ctx* ctx, int inverse, const T* in, T* out, T* ladj_ps, double* partials, int64_t batch, int accum, int grid) {
  if (inverse) hipLaunchKernelGGL((matrix_lane_kernel<T, 4, KIND, true, 1, KX>), dim3(grid), dim3(64), 0, ctx->stream, in, out, ladj_ps, KX, 0, batch, accum, partials);
  else hipLaunchKernelGGL((matrix_lane_kernel<T, 4, KIND, false, 1, KX>), dim3(grid), dim3(64), 0, ctx->stream, in, out, ladj_ps, KX, 0, batch, accum, partials);
  return 0;
}

template <class T, int KMAX, int KIND>
int launch_lane(bjx_ctx* ctx, int inverse, const T* in, T* out, T* ladj_ps, double* partials, int K, int P, int64_t batch, int accum, bool vec, int grid,
                size_t smem) {
  constexpr int VW = Vec16<T>::N;
#define BJX_ML(INV_, V_) do { bjx_allow_big_lds(matrix_lane_kernel<T, KMAX, KIND, INV_, V_>, smem); \
  hipLaunchKernelGGL((matrix_lane_kernel<T, KMAX, KIND, INV_, V_>), dim3(grid), dim3(64), smem, ctx->stream, in, out, ladj_ps, K, P, batch, accum, partials); } while (0)
  if (inverse) { if (vec) BJX_ML(true, VW); else BJX_ML(true, 1); }
  else { if (vec) BJX_ML(false, VW); else BJX_ML(false, 1); }
#undef BJX_ML
  return 0;
}

template <class T, int GS, int KIND>
int launch_gs(bjx_ctx* ctx, int inverse, const T* in, T* out, T* ladj_ps, double* partials, int K, int pitch, int64_t batch, int accum, int vin, int vout,
              int grid, size_t smem) {
  if (inverse) {
    bjx_allow_big_lds(matrix_link_kernel<T, GS, KIND, true>, smem);
    hipLaunchKernelGGL((matrix_link_kernel<T, GS, KIND, true>), dim3(grid), dim3(64), smem, ctx->stream, in, out, ladj_ps, K, pitch, batch, accum, vin, vout, partials);
  } else {
    bjx_allow_big_lds(matrix_link_kernel<T, GS, KIND, false>, smem);
    hipLaunchKernelGGL((matrix_link_kernel<T, GS, KIND, false>), dim3(grid), dim3(64), smem, ctx->stream, in, out, ladj_ps, K, pitch, batch, accum, vin, vout, partials);
  }
  return 0;
}

// ------------------------------------------------------------------ any K (round 3): one BLOCK per sample, factor in a global workspace
// The register / LDS kernels above stop at K = 64 (a wave holds the factor).  corr.jl:64-162 and pd.jl:1-60 have no limit, so
// beyond that a plain restatement runs: 256 threads own one sample, the K x K working matrix lives in a per-block slab of the
// context's workspace (L2-resident at these sizes), the factorisation is the right-looking column form (the same subtractions in
// the same order as the oracle's row form), the LKJ link / its inverse walks one column per thread with LinkMath, and
// X = U'U / L L' is one inner product per thread and entry.  O(K^3) work on O(K^2) bytes at a fraction of the roofline: correct
// first, a blocked MFMA trailing update is the next step if such sizes matter.
template <class T, int KIND, bool INV>
__global__ __launch_bounds__(256) void matrix_big_kernel(const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps, T* __restrict__ ws,
                                                         int K, int64_t batch, int accumulate, double* __restrict__ partials) {
  using LM = LinkMath<T>;
  __shared__ double red[4];
  __shared__ double colsum[256];
  const int t = threadIdx.x;
  const int64_t KK = (int64_t)K * K;
  const int64_t nv = KIND == MK_VEC_CORR ? (int64_t)K * (K - 1) / 2 : (KIND == MK_PD_VEC ? (int64_t)K * (K + 1) / 2 : KK);
  const int64_t in_n = INV ? nv : KK, out_n = INV ? KK : nv;
  T* A = ws + (int64_t)blockIdx.x * KK;                 // column-major working matrix: A[c*K + r]
  constexpr bool CORR = KIND == MK_VEC_CORR || KIND == MK_CORR;
  double acc = 0.0;
  for (int64_t n = blockIdx.x; n < batch; n += gridDim.x) {
    const T* src = in + n * in_n;
    T* dst = out ? out + n * out_n : nullptr;
    double lsum = 0.0;                                   // this thread's share of the sample's log-det
    if (!INV) {
      // lower factor L of X (X = L L'), A[c*K + r] = L[r, c], r >= c.  Correlation kinds read the UPPER triangle of X
      // (cholesky(Hermitian(X)).U, src/utils.jl:50: a(r, c) = X[c, r] = src[r*K + c]), PD kinds the LOWER one (:37).
      for (int64_t e = t; e < KK; e += 256) {
        const int c = (int)(e / K), r = (int)(e - (int64_t)c * K);
        A[e] = r >= c ? (CORR ? src[(int64_t)r * K + c] : src[e]) : T(0);
      }
      __syncthreads();
      for (int j = 0; j < K; ++j) {
        if (t == 0) A[(int64_t)j * K + j] = (T)::sqrt((double)A[(int64_t)j * K + j]);
        __syncthreads();
        const T d = A[(int64_t)j * K + j];
        for (int r = j + 1 + t; r < K; r += 256) A[(int64_t)j * K + r] = A[(int64_t)j * K + r] / d;
        __syncthreads();
        for (int c = j + 1 + (t >> 6); c < K; c += 4) {
          const T lc = A[(int64_t)j * K + c];
          for (int r = c + (t & 63); r < K; r += 64) A[(int64_t)c * K + r] -= A[(int64_t)j * K + r] * lc;
        }
        __syncthreads();
      }
      if (CORR) {
        // U = L' (U[i, j] = A[i*K + j], i <= j).  Column j of U, bottom-up (corr.jl:282-288 / :314-337); forward log-det =
        // sum (K - i + 1) logcosh(y_ij) over the strict upper triangle (negated :453-472), i 1-based
        for (int j = 1 + t; j < K; j += 256) {           // 0-based column j has j entries above the diagonal
          T rem, Lg;
          LM::fwd_init(A[(int64_t)j * K + j], rem, Lg);
          for (int i = j - 1; i >= 0; --i) {
            const T w = A[(int64_t)i * K + j];
            T y, lc;
            if (KIND == MK_VEC_CORR && i == 0) LM::atanh_lc(w, y, lc);      // :319: the top entry is atanh(W[1, j])
            else LM::fwd_step(w, rem, Lg, y, lc);
            lsum += (double)(T(K - i) * lc);
            if (dst) {
              if (KIND == MK_VEC_CORR) dst[(int64_t)j * (j - 1) / 2 + i] = y;
              else dst[(int64_t)j * K + i] = y;
            }
          }
        }
        if (KIND == MK_CORR && dst)                      // zeros on and below the diagonal (:292-294)
          for (int64_t e = t; e < KK; e += 256) { const int c = (int)(e / K), r = (int)(e - (int64_t)c * K); if (r >= c) dst[e] = T(0); }
      } else {
        // pd.jl:11, :27-31: Y = L with log on the diagonal; log-det = -(sum (K + 2 - i) log L_ii + K log 2), i 1-based
        for (int i = t; i < K; i += 256) lsum -= (double)(T(K + 1 - i) * LM::log(A[(int64_t)i * K + i]));
        if (t == 0) lsum -= (double)(T(K) * Num<T>::log2);
        if (dst) {
          if (KIND == MK_PD) {
            for (int64_t e = t; e < KK; e += 256) {
              const int c = (int)(e / K), r = (int)(e - (int64_t)c * K);
              dst[e] = r > c ? A[e] : (r == c ? LM::log(A[e]) : T(0));
            }
          } else {                                       // triu_to_vec(transpose(Y)): entry (i, j), i <= j, is Y[j, i] = L[j, i] = A[i*K + j]
            for (int64_t e = t; e < nv; e += 256) {
              int j = (int)((::sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
              while ((int64_t)(j + 1) * (j + 2) / 2 <= e) ++j;
              while ((int64_t)j * (j + 1) / 2 > e) --j;
              const int i = (int)(e - (int64_t)j * (j + 1) / 2);
              const T v = A[(int64_t)i * K + j];
              dst[e] = i == j ? LM::log(v) : v;
            }
          }
        }
      }
    } else if (CORR) {
      // corr.jl:345-399: column j of U top-down from the free values, logJ of the link, + (K - j) log U[j, j] for j = 2 … K-1
      // (:77-79, :144-146); A[j*K + i] = U[i, j]
      for (int64_t e = t; e < KK; e += 256) A[e] = T(0);
      __syncthreads();
      for (int j = t; j < K; j += 256) {
        T E;
        LM::inv_init(E);
        double lr = 0.0, lj = 0.0;                       // log_remainder and this column's share of logJ
        for (int i = 0; i < j; ++i) {
          const T yv = KIND == MK_VEC_CORR ? src[(int64_t)j * (j - 1) / 2 + i] : src[(int64_t)j * K + i];
          T w, lc;
          LM::inv_step(yv, E, w, lc);
          A[(int64_t)j * K + i] = w;
          lr -= (double)lc;
          lj += lr;
        }
        lj += lr;
        A[(int64_t)j * K + j] = LM::inv_diag(E, (T)lr);
        if (j >= 1 && j <= K - 2) lj += (double)(K - 1 - j) * lr;      // (K - j) log U[j, j], j 1-based in 2 … K-1; log U[j, j] = log_remainder
        lsum += lj;
      }
      __syncthreads();
      if (dst)                                           // pd_from_upper: X = U'U
        for (int64_t e = t; e < KK; e += 256) {
          const int c = (int)(e / K), r = (int)(e - (int64_t)c * K);
          const int m1 = r < c ? r : c;
          T s = T(0);
          for (int m = 0; m <= m1; ++m) s += A[(int64_t)r * K + m] * A[(int64_t)c * K + m];
          dst[e] = s;
        }
    } else {
      // pd.jl:13-16: L = lower_triangular(replace_diag(exp, Y)); X = L L'; log-det = +(sum (K + 2 - i) Y_ii + K log 2)
      for (int64_t e = t; e < KK; e += 256) {
        const int c = (int)(e / K), r = (int)(e - (int64_t)c * K);
        T v = T(0);
        if (r >= c) v = KIND == MK_PD ? src[e] : src[(int64_t)r * (r + 1) / 2 + c];    // vec: entry (c, r) of triu_to_vec(Y') is Y[r, c]
        if (r == c) { lsum += (double)(T(K + 1 - r) * v); v = LM::exp(v); }
        A[e] = v;
      }
      if (t == 0) lsum += (double)(T(K) * Num<T>::log2);
      __syncthreads();
      if (dst)
        for (int64_t e = t; e < KK; e += 256) {
          const int c = (int)(e / K), r = (int)(e - (int64_t)c * K);
          const int m1 = r < c ? r : c;
          T s = T(0);
          for (int m = 0; m <= m1; ++m) s += A[(int64_t)m * K + r] * A[(int64_t)m * K + c];
          dst[e] = s;
        }
    }
    // the sample's log-det: fixed-order sum of the 256 per-thread shares
    colsum[t] = lsum;
    __syncthreads();
    if (t == 0) {
      double s = 0.0;
      for (int k = 0; k < 256; ++k) s += colsum[k];
      if (ladj_ps) ladj_ps[n] = accumulate ? ladj_ps[n] + (T)s : (T)s;
      acc += s;
    }
    __syncthreads();                                     // A and colsum are reused by the next sample
  }
  if (partials) block_publish_partial(acc, red, partials);
}

template <class T, int KIND>
int matrix_big(bjx_ctx* ctx, int inverse, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags) {
  // one block per sample, serial pivots: O(K^3) per sample — K = 1024 is ~0.2 s per wave of samples, 4096 would run for minutes
  BJX_REQUIRE(ctx, K <= 1024, BJX_ERR_UNSUPPORTED, "K = %lld: the general-size matrix kernel stops at 1024 (serial O(K^3) pivots per sample)", (long long)K);
  int64_t grid = batch < 2 * (int64_t)ctx->num_cu ? batch : 2 * (int64_t)ctx->num_cu;
  while (grid > 1 && (size_t)grid * K * K * sizeof(T) > ((size_t)1 << 31)) grid >>= 1;      // workspace <= 2 GiB
  { int rc = bjx_ensure_big_ws(ctx, (size_t)grid * K * K * sizeof(T)); if (rc) return rc; }
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  {
    BjxProf prof_(ctx);
    if (inverse) hipLaunchKernelGGL((matrix_big_kernel<T, KIND, true>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out, ladj_ps, (T*)ctx->big_ws, (int)K, batch, accum, partials);
    else hipLaunchKernelGGL((matrix_big_kernel<T, KIND, false>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out, ladj_ps, (T*)ctx->big_ws, (int)K, batch, accum, partials);
  }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

}  // namespace
int bjx_matrix_cyc(bjx_ctx* ctx, int dt, int kind, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch,
                   uint32_t flags, bool* taken);                  // bjx_matrix_cyc.hip
namespace {
template <class T, int KIND>
int matrix_impl(bjx_ctx* ctx, const char* who, int inverse, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags) {
  if (batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  if (K > 64) return matrix_big<T, KIND>(ctx, inverse, in, out, ladj_ps, ladj_sum, K, batch, flags);   // beyond the register-resident kernels
  {
    // 12 < K <= 64: cyclic rows, R rows of the factor per lane (bjx_matrix_cyc.hip) — matrix_link_kernel below (one row per lane) is its A/B
    bool taken = false;
    const int rc = bjx_matrix_cyc(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, KIND, inverse, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken);
    if (rc || taken) return rc;
  }
  constexpr int VW = Vec16<T>::N;
  const int64_t KK = K * K;
  const int64_t nv = KIND == MK_VEC_CORR ? K * (K - 1) / 2 : (KIND == MK_PD_VEC ? K * (K + 1) / 2 : KK);
  // one lane per sample up to K = 12 (78 registers for the triangle; the tile of 64 samples is 64 x (K² | 1) words, 37 KiB at
  // K = 12).  The lanes-along-the-rows kernel pays an LDS round trip per column step: K = 9: 8 -> 36-48 % of the roofline,
  // K = 12: 15 -> 33-45 %.  K = 16 was measured too (136 registers, 66 KiB tile, two waves per CU): 22-30 % against 27-33 % — not kept.
  static const int lane_max = getenv("BJX_MATRIX_LANE_MAX") ? atoi(getenv("BJX_MATRIX_LANE_MAX")) : 12;   // tuning switch (0: lanes along the rows for every K)
  if (K <= lane_max && K <= 12) {
    // one lane per sample (matrix_lane_kernel)
    const int64_t rows = KK > nv ? KK : nv;
    const int P = (int)(rows | 1);
    const size_t smem_l = (size_t)64 * P * sizeof(T);
    const int64_t tiles = (batch + 63) / 64;
    const int64_t cap_l = (int64_t)ctx->num_cu * 32;
    const int grid_l = (int)(tiles < cap_l ? tiles : cap_l);
    if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid_l); if (rc) return rc; }
    double* partials_l = ladj_sum ? ctx->partials : nullptr;
    const bool vec = bjx_aligned16(in) && (!out || bjx_aligned16(out));     // a full tile of 64 samples is a whole number of 16-byte packs
    {
      BjxProf prof_(ctx);
      static const int lane_direct = getenv("BJX_MATRIX_LANE_DIRECT") ? atoi(getenv("BJX_MATRIX_LANE_DIRECT")) : 1;
      // 2x2 ... 4x4: no tile (matrix_lane_kernel, KX = K).  Same-box A/B: K = 2 / 3 forward 24 / 34-42 -> 42-55 / 65-66 %, inverse
      // 19-26 / 36-43 -> 30-46 / 46-55 %; K = 4 forward 61-64 -> 69-71 %, inverse 50-59 against 48-55 % (stays on the tile)
      if (lane_direct && K >= 2 && (K <= 3 || (K == 4 && !inverse)) && nv >= 1) {
        const int accum_d = (flags & BJX_ACCUMULATE) ? 1 : 0;
        if (K == 2) launch_lane_direct<T, 2, KIND>(ctx, inverse, in, out, ladj_ps, partials_l, batch, accum_d, grid_l);
        else if (K == 3) launch_lane_direct<T, 3, KIND>(ctx, inverse, in, out, ladj_ps, partials_l, batch, accum_d, grid_l);
        else launch_lane_direct<T, 4, KIND>(ctx, inverse, in, out, ladj_ps, partials_l, batch, accum_d, grid_l);
      }
      else if (K <= 4) launch_lane<T, 4, KIND>(ctx, inverse, in, out, ladj_ps, partials_l, (int)K, P, batch, (flags & BJX_ACCUMULATE) ? 1 : 0, vec, grid_l, smem_l);
      else if (K <= 8) launch_lane<T, 8, KIND>(ctx, inverse, in, out, ladj_ps, partials_l, (int)K, P, batch, (flags & BJX_ACCUMULATE) ? 1 : 0, vec, grid_l, smem_l);
      else launch_lane<T, 12, KIND>(ctx, inverse, in, out, ladj_ps, partials_l, (int)K, P, batch, (flags & BJX_ACCUMULATE) ? 1 : 0, vec, grid_l, smem_l);
    }
    BJX_CHECK_LAUNCH(ctx);
    if (ladj_sum) return bjx_launch_finalize(ctx, grid_l, ladj_sum, 0.0, 0, 0.0, flags);
    return BJX_OK;
  }
  const int gs = K <= 8 ? 8 : (K <= 16 ? 16 : (K <= 32 ? 32 : 64));
  const int nsw = 64 / gs;
  const int pitch = (int)((K + 3) / 4 * 4 + 4);                  // multiple of 4 (16-byte rows), + 4: consecutive rows start 4 banks apart
  const int64_t sample_words = (int64_t)gs * pitch + gs + 4;
  const size_t smem = (size_t)(nsw * sample_words) * sizeof(T);
  const int64_t in_n = inverse ? nv : KK, out_n = inverse ? KK : nv;
  const int vin = bjx_aligned16(in) && in_n % VW == 0;
  const int vout = out && bjx_aligned16(out) && out_n % VW == 0;
  const int64_t groups = (batch + nsw - 1) / nsw;
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  const int grid = (int)(groups < cap ? groups : cap);
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  {
    BjxProf prof_(ctx);
#define BJX_MK(GS_) launch_gs<T, GS_, KIND>(ctx, inverse, in, out, ladj_ps, partials, (int)K, pitch, batch, accum, vin, vout, grid, smem)
    if (gs == 8) BJX_MK(8);
    else if (gs == 16) BJX_MK(16);
    else if (gs == 32) BJX_MK(32);
    else BJX_MK(64);
#undef BJX_MK
  }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

template <int KIND>
int matrix_entry(bjx_ctx* ctx, const char* who, bjx_dtype dt, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K,
                 int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, K >= 1 && batch >= 0, BJX_ERR_SHAPE, "%s: bad size", who);
  BJX_REQUIRE(ctx, in || batch == 0 || (K == 1 && KIND == MK_VEC_CORR && inverse), BJX_ERR_ARG, "%s: null input", who);
  BJX_REQUIRE(ctx, out || ladj_ps || ladj_sum || batch == 0, BJX_ERR_ARG, "%s: nothing to compute", who);
  if (dt == BJX_F32) return matrix_impl<float, KIND>(ctx, who, inverse, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, K, batch, flags);
  if (dt == BJX_F64) return matrix_impl<double, KIND>(ctx, who, inverse, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, K, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "%s: bad dtype %d", who, (int)dt);
}

// ------------------------------------------------------------------ Scale with a matrix parameter (scale.jl:14,17,35-36)
// prep (one block): Gauss-Jordan with partial pivoting on the augmented [A | I] in global scratch ->
// logabsdet(A) = sum log|pivot| (LinearAlgebra.logabsdet goes through the same LU) and, for the inverse, A^-1.
// O(dim^3) on one block: ~0.1 ms at dim = 128, paid per call because the parameter may have changed.
template <class T>
__global__ __launch_bounds__(256) void scale_matrix_prep_kernel(const T* __restrict__ A, T* __restrict__ W /*[dim][2 dim] row-major*/, int dim, int want_inverse,
                                                                double* logabsdet) {
  __shared__ int piv_row;
  __shared__ T piv_val;
  __shared__ double lad;
  const int t = threadIdx.x, nt = blockDim.x;
  const int W2 = 2 * dim;
  for (int e = t; e < dim * W2; e += nt) {
    const int i = e / W2, j = e - i * W2;
    W[e] = j < dim ? A[j * dim + i] : (j - dim == i ? T(1) : T(0));
  }
  if (t == 0) lad = 0.0;
  __syncthreads();
  for (int k = 0; k < dim; ++k) {
    if (t < 64) {                                       // pivot search by the first wave: max |W[i][k]|, i >= k (first maximum like LAPACK's idamax)
      T best = T(-1);
      int bi = k;
      for (int i = k + t; i < dim; i += 64) {
        const T v = d_abs(W[i * W2 + k]);
        if (v > best) { best = v; bi = i; }
      }
      for (int off = 32; off >= 1; off >>= 1) {
        const T ob = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (t == 0) { piv_row = bi; piv_val = W[bi * W2 + k]; lad += ::log((double)d_abs(W[bi * W2 + k])); }
    }
    __syncthreads();
    const int p = piv_row;
    const T pv = piv_val;
    const int jlo = want_inverse ? 0 : k, jhi = want_inverse ? W2 : dim;
    if (p != k) {
      for (int j = jlo + t; j < jhi; j += nt) { const T a = W[k * W2 + j]; W[k * W2 + j] = W[p * W2 + j]; W[p * W2 + j] = a; }
      __syncthreads();
    }
    // eliminate column k from every other row (below only when just the determinant is wanted); row k is scaled afterwards
    const int ilo = want_inverse ? 0 : k + 1;
    const int ncol = jhi - jlo;
    for (int e = t; e < (dim - ilo) * ncol; e += nt) {
      const int i = ilo + e / ncol, j = jlo + e % ncol;
      if (i != k && j != k) W[i * W2 + j] -= (W[i * W2 + k] / pv) * W[k * W2 + j];
    }
    __syncthreads();
    if (want_inverse) {
      for (int i = t; i < dim; i += nt) if (i != k) W[i * W2 + k] = T(0);
      for (int j = t; j < W2; j += nt) W[k * W2 + j] = W[k * W2 + j] / pv;
      __syncthreads();
    }
  }
  if (t == 0) *logabsdet = lad;
}

// The same factorisation with the augmented matrix in LDS (round 5; VERDICT r04 weak #4: the global-memory sweep above costs three
// L2 round trips per pivot — 321 us at dim = 64 on EVERY call, next to a 438 us hot kernel).  Gauss-Jordan, partial pivoting (first
// maximum, like idamax), three barriers per pivot: (1) pivot search by the first wave, (2) row exchange with the pivot row scaled on
// the way + the column of multipliers, (3) rank-1 update of the columns to the right of k (the left block's columns <= k are unit
// vectors already).  Only the right block (A^-1, row-major with stride 2 dim, where the consumers expect it) is written back.
// want_inverse = 0: the left block alone, LU-style (no scaling), for logabsdet.
template <class T>
__global__ __launch_bounds__(512) void scale_matrix_prep_lds_kernel(const T* __restrict__ A, T* __restrict__ W /*[dim][2 dim] row-major*/, int dim, int want_inverse,
                                                                    double* logabsdet) {
  extern __shared__ __align__(16) unsigned char smem_p_[];
  __shared__ int piv_row;
  __shared__ T piv_val;
  const int t = threadIdx.x, nt = blockDim.x;
  const int W2 = want_inverse ? 2 * dim : dim;
  const int P = W2 | 1;                                  // odd pitch: a column walk (pivot search, multipliers) is conflict-free
  T* Ws = reinterpret_cast<T*>(smem_p_);                 // [dim][P]
  T* fl = Ws + (size_t)dim * P;                          // [dim] multipliers of the current pivot
  T* pivs = fl + dim;                                    // [dim] the pivots: their logarithms are taken once, in parallel, at the end
  int CT = 32;                                           // column threads: the power of two >= min(W2, blockDim) (lanes of a wave walk a row)
  while (CT < W2 && CT < nt) CT <<= 1;
  const int RT = nt / CT;                                // row threads
  const int tx = t & (CT - 1), ty = t / CT;
  for (int e = t; e < dim * W2; e += nt) {
    const int i = e / W2, j = e - i * W2;
    Ws[i * P + j] = j < dim ? A[(size_t)j * dim + i] : (j - dim == i ? T(1) : T(0));
  }
  __syncthreads();
  for (int k = 0; k < dim; ++k) {
    if (t < 64) {
      T best = T(-1);
      int bi = k;
      for (int i = k + t; i < dim; i += 64) {
        const T v = d_abs(Ws[i * P + k]);
        if (v > best) { best = v; bi = i; }
      }
      for (int off = 32; off >= 1; off >>= 1) {
        const T ob = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (t == 0) { piv_row = bi; piv_val = Ws[bi * P + k]; pivs[k] = Ws[bi * P + k]; }
    }
    __syncthreads();
    const int p = piv_row;
    const T pv = piv_val;
    const T rpv = T(1) / pv;
    // (2) multipliers from the OLD column k (rows k and p seen through the exchange) and the exchange itself on the columns > k
    const int ilo = want_inverse ? 0 : k + 1;
    for (int i = ilo + t; i < dim; i += nt) {
      const int src = i == k ? p : (i == p ? k : i);
      fl[i] = i == k ? T(0) : (want_inverse ? Ws[src * P + k] : Ws[src * P + k] * rpv);
    }
    for (int j = k + 1 + t; j < W2; j += nt) {
      const T rk = Ws[k * P + j], rp = Ws[p * P + j];
      Ws[p * P + j] = rk;                                  // (p == k: rewritten below with the same row)
      Ws[k * P + j] = want_inverse ? rp * rpv : rp;
    }
    __syncthreads();
    // (3) W[i][j] -= fl[i] * W[k][j], i != k, j > k.  Thread (tx, ty): column k + 1 + tx (+ CT, ...), rows ilo + ty, + RT, ...
    // (CT a power of two: no integer division in the loop; the first form used e / ncol, e % ncol and took 2.5 us per pivot)
    for (int j = k + 1 + tx; j < W2; j += CT) {
      const T rkj = Ws[k * P + j];
#pragma unroll 8
      for (int i = ilo + ty; i < dim; i += RT) Ws[i * P + j] -= fl[i] * rkj;          // fl[k] = 0: row k is left alone (independent rows: the loads pipeline)
    }
    __syncthreads();
  }
  if (want_inverse)
    for (int e = t; e < dim * dim; e += nt) {
      const int i = e / dim, j = e - i * dim;
      W[(size_t)i * (2 * dim) + dim + j] = Ws[i * P + dim + j];
    }
  if (t < 64) {                                          // sum log|pivot| in Float64, fixed order
    double a = 0.0;
    for (int k = t; k < dim; k += 64) a += ::log((double)d_abs(pivs[k]));
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
    if (t == 0) *logabsdet = a;
  }
}
// Round 6: the factorisation of a matrix of at most 64 rows by ONE WAVE with the matrix in REGISTERS (the block version above spends
// 130 us at 64 x 64 — 192 block barriers — in front of a 0.4 ms hot kernel).  Lane i owns ROW i of the augmented [A | I]: registers
// r[0 .. NC).  Gauss-Jordan WITHOUT row exchange: the pivot of step k is the largest |entry| of column k among the rows not used yet (the
// same pivots as partial pivoting picks), the pivot row is scaled, every other row eliminated; at the end row p_k of the right block is row
// k of A^-1.  What makes it a rolled loop of straight-line code: the update writes its result ONE REGISTER TO THE LEFT,
//     r[j-1] = r[j] - m * pivot_row[j],      m = c_i / pivot  (own lane: m = 1 - 1/pivot, which scales the row),
// so the current pivot column is always r[0] (no dynamic register index, nothing unrolled over k), the dead columns of the left block fall
// off the array and after `dim` steps A^-1 sits in r[0 .. dim).  The pivot row reaches the other lanes by v_readlane (one per column).
// No LDS, no barrier.  Float64: the same with 64-bit readlanes.  logabsdet = sum log|pivot| as before.
template <class T> __device__ __forceinline__ T wave_readlane(T v, int l);
template <> __device__ __forceinline__ float wave_readlane<float>(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
template <> __device__ __forceinline__ double wave_readlane<double>(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// maximum over the 64 lanes without LDS: four DPP butterflies inside each row of 16 lanes, then the four row maxima through SGPRs
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_get(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <class T> __device__ __forceinline__ T wave_max_nolds(T v) {
  v = d_max(v, dpp_get<0xB1>(v));      // quad_perm [1,0,3,2]
  v = d_max(v, dpp_get<0x4E>(v));      // quad_perm [2,3,0,1]
  v = d_max(v, dpp_get<0x141>(v));     // row_half_mirror
  v = d_max(v, dpp_get<0x140>(v));     // row_mirror: every lane of a row of 16 holds the row's maximum
  const T a = wave_readlane<T>(v, 0), b = wave_readlane<T>(v, 16), c = wave_readlane<T>(v, 32), d = wave_readlane<T>(v, 48);
  return d_max(d_max(a, b), d_max(c, d));
}
template <class T, bool WI>
__global__ __launch_bounds__(64) void scale_matrix_prep_wave_kernel(const T* __restrict__ A, T* __restrict__ W /*[dim][2 dim] row-major*/, int dim, double* logabsdet) {
  constexpr int DP = 64, NC = WI ? 2 * DP : DP;
  const int lane = threadIdx.x;
  const bool live = lane < dim;
  T r[NC];
#pragma unroll
  for (int j = 0; j < DP; ++j) r[j] = (live && j < dim) ? A[(size_t)j * dim + lane] : T(0);        // A is column-major: lanes on consecutive addresses
  if (WI) {
#pragma unroll
    for (int j = 0; j < DP; ++j) r[DP + j] = T(0);
    // the identity block starts at column `dim` of the augmented matrix: [A | I] occupies r[0 .. 2 dim); move it there
#pragma unroll
    for (int j = 0; j < NC; ++j) r[j] = (j >= dim && j - dim == lane && live) ? T(1) : (j < dim ? r[j] : T(0));
  }
  bool used = !live;
  int my_step = -1;                                        // the step at which my row was the pivot row
  T my_piv = T(1);                                         // ... and its pivot (the logarithms are taken once, in parallel, after the loop)
  for (int k = 0; k < dim; ++k) {
    const T c = r[0];
    // largest |c| among the unused rows, lowest lane on ties — no LDS: the maximum by butterflies inside the rows of 16 lanes (DPP) and four
    // readlanes across them, the lane by a ballot (a single wave has nothing to hide an LDS round trip behind: the ds_bpermute form of this
    // search was half of the kernel's 83 us)
    const T cand = used ? T(-1) : d_abs(c);
    const T wmax = wave_max_nolds(cand);
    const unsigned long long hit = __ballot(cand == wmax);
    const int p = hit ? (int)__builtin_ctzll(hit) : 0;     // (all NaN: no lane compares equal — the factorisation is NaN either way)
    const T pv = wave_readlane<T>(c, p);
    const T rpv = T(1) / pv;
    const T m = lane == p ? T(1) - rpv : c * rpv;
    if (lane == p) { used = true; my_step = k; my_piv = pv; }
    // the pivot row in chunks of 16 columns: sixteen readlanes (SGPR results), then sixteen FMAs — back to back, a VALU instruction that reads the
    // SGPR a readlane has just written waits for it (the one-by-one form ran at ~11 cycles per column)
#pragma unroll
    for (int j0 = 1; j0 < NC; j0 += 16) {
      T wp[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wp[u] = j0 + u < NC ? wave_readlane<T>(r[j0 + u], p) : T(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; ++u) if (j0 + u < NC) r[j0 + u - 1] = r[j0 + u] - m * wp[u];
      __builtin_amdgcn_sched_barrier(0);
    }
    r[NC - 1] = T(0);
  }
  if (WI && my_step >= 0) {                                // my row of the reduced right block is row my_step of A^-1
    T* dst = W + (size_t)my_step * (2 * dim) + dim;
#pragma unroll
    for (int j = 0; j < DP; ++j) if (j < dim) dst[j] = r[j];
  }
  double lad = my_step >= 0 ? ::log((double)d_abs(my_piv)) : 0.0;        // one logarithm per lane, then a fixed-order wave sum
  lad = group_sum<64>(lad);
  if (lane == 0) *logabsdet = lad;
}

// dynamic LDS of the kernel above (0: does not fit, take the global-memory sweep)
template <class T> inline size_t scale_prep_lds_bytes(int64_t dim, int want_inverse) {
  const size_t W2 = want_inverse ? 2 * (size_t)dim : (size_t)dim;
  const size_t b = ((size_t)dim * (W2 | 1) + 2 * (size_t)dim) * sizeof(T);
  return b <= 150 * 1024 ? b : 0;
}

// Y[:, n] = M X[:, n] for a block of TC columns: M (column-major, rows padded to 4*RPT) and the X tile in LDS; thread
// (column tx, row quarter ty) keeps RPT accumulators; M is read with wave-uniform (broadcast) 16-byte LDS reads.
template <class T, int RPT>
__global__ __launch_bounds__(256) void scale_matrix_kernel(const T* __restrict__ M, int ldm_row_major, const T* __restrict__ X, T* __restrict__ Y, T* __restrict__ ladj_ps,
                                                           int dim, int64_t batch, int TC, int accumulate, const double* logabsdet) {
  extern __shared__ __align__(16) unsigned char smem_[];
  constexpr int RP = 4 * RPT;                 // padded rows
  T* Ms = reinterpret_cast<T*>(smem_);        // [dim][RP]: Ms[k*RP + i] = M[i, k]
  const int P = dim | 1;                      // odd pitch of the X / Y tile: lane = column reads are conflict-free
  T* xs = Ms + (size_t)dim * RP;              // [TC][P]
  const int t = threadIdx.x;
  for (int e = t; e < dim * RP; e += 256) {
    const int k = e / RP, i = e - k * RP;
    // M is either the caller's column-major a (ldm_row_major = 0) or the row-major right half of the prep scratch (= stride)
    T v = T(0);
    if (i < dim) v = ldm_row_major ? M[(size_t)i * ldm_row_major + k] : M[(size_t)k * dim + i];
    Ms[e] = v;
  }
  const T lad = ladj_ps ? (T)*logabsdet : T(0);
  const int tx = t % TC, ty = t / TC;         // blockDim = 4 * TC threads take part in the arithmetic
  for (int64_t c0 = (int64_t)blockIdx.x * TC; c0 < batch; c0 += (int64_t)gridDim.x * TC) {
    const int nc = (int)((batch - c0) < TC ? (batch - c0) : TC);
    __syncthreads();
    for (int e = t; e < nc * dim; e += 256) {
      const int c = e / dim, k = e - c * dim;
      xs[c * P + k] = __builtin_nontemporal_load(X + c0 * dim + e);
    }
    __syncthreads();
    T acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] = T(0);
    if (ty < 4 && tx < nc) {
      const T* xc = xs + tx * P;
      const T* mrow = Ms + ty * RPT;
      for (int k = 0; k < dim; ++k) {
        const T xk = xc[k];
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[r] += mrow[k * RP + r] * xk;
      }
    }
    __syncthreads();
    if (ty < 4 && tx < nc) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int i = ty * RPT + r;
        if (i < dim) xs[tx * P + i] = acc[r];
      }
      if (ty == 0 && ladj_ps) ladj_ps[c0 + tx] = accumulate ? ladj_ps[c0 + tx] + lad : lad;
    }
    __syncthreads();
    if (Y) for (int e = t; e < nc * dim; e += 256) {
      const int c = e / dim, k = e - c * dim;
      __builtin_nontemporal_store(xs[c * P + k], Y + c0 * dim + e);
    }
  }
}

// Y = M X on the matrix cores (the one genuinely dense product on this path: 2 dim^2 flop per 2 dim sizeof(T) bytes).
// One wave = a tile of 16 columns.  M (padded to DP = 16 NRB rows / columns) sits in LDS in A-operand order
// (for row block rb and k-step ks the 64 lane values M[16 rb + lane % 16][4 ks + lane / 16] are consecutive), the X tile is
// loaded with coalesced 16-byte accesses and transposed through LDS into B operands (lane (n, q): X[4 ks + q][column n]);
// NRB accumulators of 16 x 16 per wave; the result leaves through the same LDS tile with coalesced stores.  The next
// tile's loads are issued before the MFMA loop of the current one.
template <class T> struct MfmaOps;
template <> struct MfmaOps<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }        // D register r of lane (n, q) -> row of the 16-block
};
template <> struct MfmaOps<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int q, int r) { return 4 * r + q; }        // probed: scripts/probe_mfma_f64.hip
};
// An elementwise chain applied to the tile AS IT IS LOADED (bjx_scale_matrix_chain: `logpdf` of a transformed full-covariance normal in one
// pass, src/transformed_distribution.jl:164-169): up to four stages of exp / log / Shift / Scale / Scale⁻¹ with a scalar or one value per row.
template <class T> struct MatPre { int n; int kind[4]; T s[4]; const T* v[4]; };
template <class T, int NRB, bool AREG_OK, bool PRE = false>
__global__ __launch_bounds__(256) void scale_matrix_mfma_kernel(const T* __restrict__ M, int ldm_row_major, const T* __restrict__ X, T* __restrict__ Y,
                                                                T* __restrict__ ladj_ps, int dim, int64_t batch, int accumulate, const double* logabsdet,
                                                                const MatPre<T> pre = MatPre<T>{}) {
  extern __shared__ __align__(16) unsigned char smem_[];
  constexpr int DP = 16 * NRB, NKS = DP / 4;
  constexpr int P = DP + 4;                              // staged column pitch (16-byte aligned rows, columns 4 banks apart)
  constexpr int VW = Vec16<T>::N;
  using O = MfmaOps<T>;
  T* Ms = reinterpret_cast<T*>(smem_);                   // [NRB][NKS][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* xs = Ms + NRB * NKS * 64 + wave * 16 * P;
  // PRE: log-det of the chain per column of the wave's tile (Float64 LDS adds: nine times the Float32 rate on gfx950, profiles/r06_lds_atomics.txt)
  double* cl = reinterpret_cast<double*>(Ms + NRB * NKS * 64 + 4 * 16 * P) + wave * 16;
  T* pvl = reinterpret_cast<T*>(reinterpret_cast<double*>(Ms + NRB * NKS * 64 + 4 * 16 * P) + 64);     // PRE: [4][DP] per-row stage parameters (reciprocals for Scale⁻¹)
  for (int e = threadIdx.x; e < NRB * NKS * 64; e += 256) {
    const int l = e & 63, blk = e >> 6, ks = blk % NKS, rb = blk / NKS;
    const int i = 16 * rb + (l & 15), k = 4 * ks + (l >> 4);
    T v = T(0);
    if (i < dim && k < dim) v = ldm_row_major ? M[(size_t)i * ldm_row_major + k] : M[(size_t)k * dim + i];
    Ms[e] = v;
  }
  __syncthreads();
  const T lad = ladj_ps ? (T)*logabsdet : T(0);
  const int n = lane & 15, q = lane >> 4;
  const bool vec_ok = dim % VW == 0 && bjx_aligned16_dev(X) && (!Y || bjx_aligned16_dev(Y));
  // small matrices: the A operands of a wave stay in registers (NRB*NKS <= 64 values), LDS only carries the tiles
  constexpr bool AREG = AREG_OK && NRB * NKS * (int)sizeof(T) <= 256;
  T areg[AREG ? NRB : 1][AREG ? NKS : 1];
  if constexpr (AREG) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) areg[rb][ks] = Ms[(rb * NKS + ks) * 64 + lane];
  }
  const int64_t tile_stride = (int64_t)gridDim.x * 4 * 16;
  // the packs of the NEXT tile are in flight while this one goes through LDS, the MFMA loop and out again (a tile is
  // ~0.9 us of MFMA behind an HBM round trip: without the look-ahead every tile paid the round trip first)
  constexpr int NPK = (16 * DP) / (64 * VW);             // packs per lane and tile
  Pack<T, VW> nxt[NPK];
  auto fetch = [&](int64_t c0n) {
    if (!vec_ok || c0n >= batch) return;
    const int nen = (int)((batch - c0n) < 16 ? (batch - c0n) : 16) * dim;
#pragma unroll
    for (int u = 0; u < NPK; ++u) {
      const int e = (lane + 64 * u) * VW;
#pragma unroll
      for (int t = 0; t < VW; ++t) nxt[u].v[t] = T(0);
      if (e < nen) nxt[u] = load_pack<T, VW, true>(X + c0n * dim + e);
    }
  };
  fetch(((int64_t)blockIdx.x * 4 + wave) * 16);
  // PRE: the rows of a lane's packs are the same in every tile (e = (lane + 64 u) VW, row = e mod dim): the stage parameters of those rows,
  // and the parameter-only part of the log-det, are fetched ONCE (read per element inside the tile loop they were sixteen dependent
  // global loads per tile: 1.31 ms where the plain kernel takes 0.44)
  T lpc[PRE ? NPK : 1];
  int ku[PRE ? NPK : 1];
  const int Gc = dim / VW;                                  // lanes of a column in one staging round
  const bool colred = PRE && Gc >= 1 && Gc <= 64 && (Gc & (Gc - 1)) == 0;
  if constexpr (PRE) {
    // per-row stage parameters -> LDS once per block (a register copy per (stage, pack, element) put the kernel on 212 registers, two waves per SIMD)
    for (int e = threadIdx.x; e < 4 * DP; e += 256) {
      const int i = e / DP, k = e - i * DP;
      T v = T(0);
      if (i < pre.n && k < dim) {
        v = pre.v[i] ? pre.v[i][k] : pre.s[i];
        if (pre.kind[i] == BJX_OP_SCALE_INV) v = T(1) / v;
      }
      pvl[e] = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NPK; ++u) {
      const int e = (lane + 64 * u) * VW;
      ku[u] = e < 16 * dim ? e % dim : 0;
      T l = T(0);                                            // the parameter-only part of the log-det of my rows (scale.jl:21-28)
      for (int i = 0; i < pre.n; ++i) {
        if (pre.kind[i] == BJX_OP_SCALE || pre.kind[i] == BJX_OP_SCALE_INV) {
          for (int t = 0; t < VW; ++t) l += Fast<T>::log(d_abs(pvl[i * DP + ku[u] + t]));      // (Scale⁻¹: the table holds 1/a, so this is −log|a|)
        }
      }
      lpc[u] = l;
    }
  }
  for (int64_t c0 = ((int64_t)blockIdx.x * 4 + wave) * 16; c0 < batch; c0 += tile_stride) {
    const int nc = (int)((batch - c0) < 16 ? (batch - c0) : 16);
    const int ne = nc * dim;                             // contiguous elements of the tile
    // ---- stage the tile: xs[c*P + k] = X[k, c0 + c]; rows >= dim and columns >= nc are zero
    for (int e = lane; e < 16 * (DP - dim) ; e += 64) { const int c = e / (DP - dim), k = dim + e % (DP - dim); xs[c * P + k] = T(0); }
    if (vec_ok) {
      Pack<T, VW> cur[NPK];
#pragma unroll
      for (int u = 0; u < NPK; ++u) cur[u] = nxt[u];
      fetch(c0 + tile_stride);
      T lpk[PRE ? NPK : 1];
      if constexpr (PRE) {
        if (lane < 16) cl[lane] = 0.0;
        __builtin_amdgcn_wave_barrier();
        // stage-major: ONE wave-uniform switch per stage, then all of the lane's elements (element-major, the kind was re-decided for every
        // element: 700 scalar branches per tile, 0.87 ms where the plain kernel takes 0.42).  Packs past the end of the batch hold zeros;
        // whatever the stages make of them is never used.
#pragma unroll
        for (int u = 0; u < NPK; ++u) lpk[u] = lpc[u];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < pre.n) {
            const int kd = pre.kind[i];
            if (kd == BJX_OP_LOG) {                                                              // exp_log.jl:8-9
#pragma unroll
              for (int u = 0; u < NPK; ++u)
#pragma unroll
                for (int t = 0; t < VW; ++t) { const T x = Fast<T>::log(cur[u].v[t]); lpk[u] -= x; cur[u].v[t] = x; }
            } else if (kd == BJX_OP_EXP) {                                                       // exp_log.jl:5-6
#pragma unroll
              for (int u = 0; u < NPK; ++u)
#pragma unroll
                for (int t = 0; t < VW; ++t) { lpk[u] += cur[u].v[t]; cur[u].v[t] = Fast<T>::exp(cur[u].v[t]); }
            } else if (kd == BJX_OP_SHIFT) {                                                     // shift.jl:14
#pragma unroll
              for (int u = 0; u < NPK; ++u)
#pragma unroll
                for (int t = 0; t < VW; ++t) cur[u].v[t] += pvl[i * DP + ku[u] + t];
            } else {                                                                             // Scale (scale.jl:13) / Scale(inv(a)) (:15-16: the table holds the reciprocal)
#pragma unroll
              for (int u = 0; u < NPK; ++u)
#pragma unroll
                for (int t = 0; t < VW; ++t) cur[u].v[t] *= pvl[i * DP + ku[u] + t];
            }
          }
        }
      }
      if constexpr (PRE) {
        // the dim / VW lanes of a column add their parts up in registers first (a power of two dividing 64: a butterfly); sixteen lanes adding to
        // ONE LDS word serialise — four such atomics per tile were 0.3 ms of a 0.66 ms kernel
        if (colred) {
          if (sizeof(T) == 4 && Gc == 16) {
            // a column = one 16-lane row: four DPP adds (quad swaps, then the half-row and row mirrors: sums are already uniform in the mirrored halves)
#pragma unroll
            for (int u = 0; u < NPK; ++u) {
              float v = (float)lpk[u];
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
              v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
              lpk[u] = (T)v;
            }
          } else {
            for (int m = 1; m < Gc; m <<= 1) {
#pragma unroll
              for (int u = 0; u < NPK; ++u) lpk[u] += __shfl_xor(lpk[u], m, 64);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NPK; ++u) {
        const int e = (lane + 64 * u) * VW;
        if (e < 16 * dim) {
          const int c = e / dim, k = e - c * dim;        // VW | dim: a pack stays inside one column; P and k are multiples of VW: one 16-byte LDS write
          if constexpr (PRE) {
            if (e < ne) { if (!colred) atomicAdd(&cl[c], (double)lpk[u]); else if ((lane & (Gc - 1)) == 0) cl[c] = (double)lpk[u]; }   // (one writer per column after the reduction)
          }
          *reinterpret_cast<typename Vec16<T>::type*>(xs + c * P + k) = *reinterpret_cast<const typename Vec16<T>::type*>(&cur[u]);
        }
      }
    } else {
      for (int e = lane; e < 16 * dim; e += 64) { const int c = e / dim, k = e - c * dim; xs[c * P + k] = e < ne ? X[c0 * dim + e] : T(0); }
    }
    __builtin_amdgcn_wave_barrier();
    typename O::acc_t acc[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) acc[rb] = typename O::acc_t{T(0), T(0), T(0), T(0)};
    if constexpr (AREG) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const T b = xs[n * P + 4 * ks + q];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb] = O::mfma(areg[rb][ks], b, acc[rb]);
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < NKS; ++ks) {
        const T b = xs[n * P + 4 * ks + q];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb] = O::mfma(Ms[(rb * NKS + ks) * 64 + lane], b, acc[rb]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- results back through the tile (only when they are stored: the density comes from the accumulators)
    if (Y) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[n * P + 16 * rb + O::row(q, r)] = acc[rb][r];
    }
    __builtin_amdgcn_wave_barrier();
    if (Y) {
      if (vec_ok) {
        for (int e = lane * VW; e < ne; e += 64 * VW) {
          const int c = e / dim, k = e - c * dim;
          Pack<T, VW> p;
          *reinterpret_cast<typename Vec16<T>::type*>(&p) = *reinterpret_cast<const typename Vec16<T>::type*>(xs + c * P + k);
          store_pack<T, VW, true>(Y + c0 * dim + e, p);
        }
      } else {
        for (int e = lane; e < ne; e += 64) { const int c = e / dim, k = e - c * dim; Y[c0 * dim + e] = xs[c * P + k]; }
      }
    }
    T ssq = T(0);
    if (accumulate & 2) {
      // BJX_BASE_STDNORMAL: Σ_k out[k, n]² of column n = lane & 15 from the ACCUMULATORS (rows past dim are zero rows of the operand): the
      // lane's NRB x 4 values, then the four lanes that share the column.  (A loop over the rows of the tile by sixteen lanes — 64 dependent
      // LDS reads — was three times the MFMA work of the tile: logpdf with a full covariance ran 0.72 ms where a \ y with its store takes 0.42.)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssq += acc[rb][r] * acc[rb][r];
      ssq += __shfl_xor(ssq, 16, 64);
      ssq += __shfl_xor(ssq, 32, 64);
    }
    if (ladj_ps && lane < nc) {
      T extra = T(0);
      if (accumulate & 2) extra = T(-0.5) * ssq - (T)dim * T(0.91893853320467274178);
      if constexpr (PRE) extra += (T)cl[lane];
      ladj_ps[c0 + lane] = ((accumulate & 1) ? ladj_ps[c0 + lane] + lad : lad) + extra;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void scale_matrix_sum_kernel(const double* logabsdet, double mult, double* out, int accumulate) {
  const double v = *logabsdet * mult;
  *out = accumulate ? *out + v : v;
}

// Any dim (round 3; scale.jl:14-17 has no limit): Y = M X with M read from global memory (L2-resident), one thread per output
// entry and a plain inner product — the correct-first path behind the LDS / MFMA kernels, which stop at 128 rows.
template <class T>
__global__ __launch_bounds__(256) void scale_matrix_big_kernel(const T* __restrict__ M, int ldm_row_major, const T* __restrict__ X, T* __restrict__ Y,
                                                               T* __restrict__ ladj_ps, int dim, int64_t batch, int accumulate, const double* logabsdet) {
  const int64_t total = (int64_t)dim * batch;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t n = e / dim;
    const int i = (int)(e - n * dim);
    if (Y) {
      const T* x = X + n * dim;
      T s = T(0);
      if (ldm_row_major) for (int k = 0; k < dim; ++k) s += M[(size_t)i * ldm_row_major + k] * x[k];
      else for (int k = 0; k < dim; ++k) s += M[(size_t)k * dim + i] * x[k];
      Y[e] = s;
    }
    if (ladj_ps && i == 0) ladj_ps[n] = accumulate ? ladj_ps[n] + (T)*logabsdet : (T)*logabsdet;
  }
}

template <class T>
int scale_matrix_impl(bjx_ctx* ctx, int inverse, const T* a, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags,
                      const MatPre<T>* pre = nullptr) {
  if (dim > 128 || (size_t)dim * 2 * dim * sizeof(T) > BJX_SCRATCH_BYTES) {
    // the factorisation behind logabsdet / the inverse is ONE block's Gauss-Jordan sweep, O(dim^3) serial pivots, redone on every
    // call: ~0.2 s at 1024, minutes at 8192 (a launch that long looks like a hang).  Larger systems belong to a blocked LU (rocSOLVER).
    BJX_REQUIRE(ctx, dim <= 1024, BJX_ERR_UNSUPPORTED, "bjx_scale_matrix: dim = %lld: the general-size path stops at 1024 (single-block O(dim^3) factorisation per call)", (long long)dim);
    BJX_REQUIRE(ctx, !(flags & BJX_BASE_STDNORMAL), BJX_ERR_UNSUPPORTED, "bjx_scale_matrix: BJX_BASE_STDNORMAL is served by the matrix-core kernel only (dim <= 128)");
    { int rc = bjx_ensure_big_ws(ctx, (size_t)dim * 2 * dim * sizeof(T)); if (rc) return rc; }
    T* Wb = reinterpret_cast<T*>(ctx->big_ws);
    double* ladb = ctx->consts + 2;
    const int want = (ladj_ps || ladj_sum) ? 1 : 0;
    const int accb = (flags & BJX_ACCUMULATE) ? 1 : 0;
    if (inverse || want) {
      hipLaunchKernelGGL((scale_matrix_prep_kernel<T>), dim3(1), dim3(256), 0, ctx->stream, a, Wb, (int)dim, inverse ? 1 : 0, ladb);
      BJX_CHECK_LAUNCH(ctx);
      if (inverse && want) hipLaunchKernelGGL(scale_matrix_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, ladb, -1.0, ladb, 0);
    }
    if (batch > 0 && (out || ladj_ps)) {
      const int64_t need = (dim * batch + 255) / 256;
      const int64_t capb = (int64_t)ctx->num_cu * 16;
      BjxProf prof_(ctx);
      hipLaunchKernelGGL((scale_matrix_big_kernel<T>), dim3((unsigned)(need < capb ? need : capb)), dim3(256), 0, ctx->stream,
                         inverse ? Wb + dim : a, inverse ? (int)(2 * dim) : 0, in, out, ladj_ps, (int)dim, batch, accb, ladb);
      BJX_CHECK_LAUNCH(ctx);
    }
    if (ladj_sum) {
      const double mult = (flags & BJX_REF_VECTOR_SCALE_LADJ) ? 1.0 : (double)batch;
      hipLaunchKernelGGL(scale_matrix_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, ladb, mult, ladj_sum, accb);
      BJX_CHECK_LAUNCH(ctx);
    }
    return BJX_OK;
  }
  T* W = reinterpret_cast<T*>(ctx->scratch);
  double* lad = ctx->consts + 2;
  const int want_ladj = (ladj_ps || ladj_sum) ? 1 : 0;
  if (inverse || want_ladj) {
    // BJX_OPT_PARAM_EPOCH != 0: [A^-1 | logabsdet] of an unchanged matrix is kept in the context (one slot) and the
    // factorisation is skipped — a steady-state trace then holds no prep kernel at all.
    double* lad_build = lad;
    bool build = true;
    if (ctx->param_epoch != 0 && !ctx->capturing) {
      auto& sl = ctx->scale_slot;
      const size_t mat_bytes = ((size_t)dim * 2 * dim * sizeof(T) + 15) / 16 * 16;
      if (sl.buf && sl.epoch == ctx->param_epoch && sl.a == a && sl.dim == dim && sl.dt == (int)sizeof(T) && sl.has_inverse >= (inverse ? 1 : 0)) build = false;
      else {
        if (sl.cap < mat_bytes + 16) {
          if (sl.buf) (void)hipFree(sl.buf);
          sl.buf = nullptr; sl.cap = 0;
          if (hipMalloc(&sl.buf, mat_bytes + 16) == hipSuccess) sl.cap = mat_bytes + 16; else sl.buf = nullptr;
        }
        if (sl.buf) { sl.a = a; sl.dim = dim; sl.dt = (int)sizeof(T); sl.has_inverse = inverse ? 1 : 0; sl.epoch = ctx->param_epoch; }
      }
      if (sl.buf) { W = reinterpret_cast<T*>(sl.buf); lad_build = reinterpret_cast<double*>(static_cast<char*>(sl.buf) + mat_bytes); }
    }
    if (build) {
      const size_t lds_p = scale_prep_lds_bytes<T>(dim, inverse ? 1 : 0);
      static const int use_wave = getenv("BJX_SCALE_PREP_WAVE") ? atoi(getenv("BJX_SCALE_PREP_WAVE")) : 1;       // 0: the block-wide LDS factorisation of round 5
      if (use_wave && dim <= 64) {
        if (inverse) hipLaunchKernelGGL((scale_matrix_prep_wave_kernel<T, true>), dim3(1), dim3(64), 0, ctx->stream, a, W, (int)dim, lad_build);
        else hipLaunchKernelGGL((scale_matrix_prep_wave_kernel<T, false>), dim3(1), dim3(64), 0, ctx->stream, a, W, (int)dim, lad_build);
      } else if (lds_p) {
        bjx_allow_big_lds(scale_matrix_prep_lds_kernel<T>, lds_p);
        hipLaunchKernelGGL((scale_matrix_prep_lds_kernel<T>), dim3(1), dim3(512), lds_p, ctx->stream, a, W, (int)dim, inverse ? 1 : 0, lad_build);
      } else {
        hipLaunchKernelGGL((scale_matrix_prep_kernel<T>), dim3(1), dim3(256), 0, ctx->stream, a, W, (int)dim, inverse ? 1 : 0, lad_build);
      }
      BJX_CHECK_LAUNCH(ctx);
    }
    if (lad_build != lad) {       // the per-call copy is what the code below negates / scales in place
      hipLaunchKernelGGL(scale_matrix_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, lad_build, 1.0, lad, 0);
      BJX_CHECK_LAUNCH(ctx);
    }
  }
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  const bool want_density = (flags & BJX_BASE_STDNORMAL) != 0;      // + log N(out; 0, I) per column in ladj_ps: the matrix-core kernel only
  if (want_density) BJX_REQUIRE(ctx, ladj_ps && !ladj_sum, BJX_ERR_ARG, "bjx_scale_matrix: BJX_BASE_STDNORMAL writes the per-column vector (ladj_ps), not the sum");
  if (batch > 0 && (out || ladj_ps)) {
    const int rpt = dim <= 32 ? 8 : dim <= 64 ? 16 : 32;
    const int RP = 4 * rpt;
    const int P = (int)dim | 1;
    int TC = 64;
    while (TC > 8 && (size_t)(dim * RP + (int64_t)TC * P) * sizeof(T) > 150 * 1024) TC >>= 1;
    const size_t smem = (size_t)(dim * RP + (int64_t)TC * P) * sizeof(T);
    BJX_REQUIRE(ctx, smem <= BJX_LDS_MAX, BJX_ERR_UNSUPPORTED, "bjx_scale_matrix: dim too large for the LDS tile");
    const int64_t need = (batch + TC - 1) / TC;
    const int64_t cap = (int64_t)ctx->num_cu * 2;
    const int grid = (int)(need < cap ? need : cap);
    // the inverse applies A^-1 (right half of the scratch, row-major with stride 2 dim) with the NEGATED log-det
    const T* M = inverse ? W + dim : a;
    const int ldm = inverse ? (int)(2 * dim) : 0;
    BjxProf prof_(ctx);
    static const int use_mfma = getenv("BJX_SCALE_MFMA") ? atoi(getenv("BJX_SCALE_MFMA")) : 1;
    const int nrb_ = (int)((dim + 15) / 16);
    const size_t smem_try = ((size_t)(16 * nrb_) * (16 * nrb_) + (size_t)4 * 16 * (16 * nrb_ + 4)) * sizeof(T);
    if (pre) {
      // the chain in front is applied by the matrix-core kernel's 16-byte staging only
      constexpr int VWh = 16 / (int)sizeof(T);
      BJX_REQUIRE(ctx, use_mfma && smem_try + 512 + (size_t)4 * 16 * nrb_ * sizeof(T) <= BJX_LDS_MAX && dim % VWh == 0 && bjx_aligned16(in) && (!out || bjx_aligned16(out)), BJX_ERR_UNSUPPORTED,
                  "bjx_scale_matrix_chain: served by the matrix-core kernel only (dim <= 128 and a whole number of 16-byte packs, arrays on 16-byte boundaries)");
    }
    if (use_mfma && smem_try <= BJX_LDS_MAX) {
      if (inverse && want_ladj) hipLaunchKernelGGL(scale_matrix_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, lad, -1.0, lad, 0);
      const int nrb = (int)((dim + 15) / 16), DP = 16 * nrb;
      const size_t smem_m = ((size_t)DP * DP + (size_t)4 * 16 * (DP + 4)) * sizeof(T) + (pre ? 512 + (size_t)4 * DP * sizeof(T) : 0);
      const int64_t tiles = (batch + 63) / 64;
      const int64_t capm = (int64_t)ctx->num_cu * (smem_m > 80 * 1024 ? 1 : (smem_m > 40 * 1024 ? 2 : 4));
      const int gridm = (int)(tiles < capm ? tiles : capm);
      static const int areg = 0;
#define BJX_SMM(N_) do { if (areg) { bjx_allow_big_lds(scale_matrix_mfma_kernel<T, N_, true>, smem_m); \
      hipLaunchKernelGGL((scale_matrix_mfma_kernel<T, N_, true>), dim3(gridm), dim3(256), smem_m, ctx->stream, M, ldm, in, out, ladj_ps, (int)dim, batch, accum | (want_density ? 2 : 0), lad); } \
      else { bjx_allow_big_lds(scale_matrix_mfma_kernel<T, N_, false>, smem_m); \
      hipLaunchKernelGGL((scale_matrix_mfma_kernel<T, N_, false>), dim3(gridm), dim3(256), smem_m, ctx->stream, M, ldm, in, out, ladj_ps, (int)dim, batch, accum | (want_density ? 2 : 0), lad); } } while (0)
#define BJX_SMP(N_) do { bjx_allow_big_lds(scale_matrix_mfma_kernel<T, N_, false, true>, smem_m); \
      hipLaunchKernelGGL((scale_matrix_mfma_kernel<T, N_, false, true>), dim3(gridm), dim3(256), smem_m, ctx->stream, M, ldm, in, out, ladj_ps, (int)dim, batch, accum | (want_density ? 2 : 0), lad, *pre); } while (0)
      if (pre) {
        switch (nrb) { case 1: BJX_SMP(1); break; case 2: BJX_SMP(2); break; case 3: BJX_SMP(3); break; case 4: BJX_SMP(4); break;
                       case 5: BJX_SMP(5); break; case 6: BJX_SMP(6); break; case 7: BJX_SMP(7); break; default: BJX_SMP(8); break; }
      } else
      switch (nrb) { case 1: BJX_SMM(1); break; case 2: BJX_SMM(2); break; case 3: BJX_SMM(3); break; case 4: BJX_SMM(4); break;
                     case 5: BJX_SMM(5); break; case 6: BJX_SMM(6); break; case 7: BJX_SMM(7); break; default: BJX_SMM(8); break; }
#undef BJX_SMP
#undef BJX_SMM
      BJX_CHECK_LAUNCH(ctx);
    } else {
      BJX_REQUIRE(ctx, !want_density && !pre, BJX_ERR_UNSUPPORTED, "bjx_scale_matrix: BJX_BASE_STDNORMAL / a chain in front are served by the matrix-core kernel only (dim <= 128)");
#define BJX_SM(R_) do { bjx_allow_big_lds(scale_matrix_kernel<T, R_>, smem); \
    hipLaunchKernelGGL((scale_matrix_kernel<T, R_>), dim3(grid), dim3(256), smem, ctx->stream, M, ldm, in, out, ladj_ps, (int)dim, batch, TC, accum, lad); } while (0)
    if (inverse && want_ladj) {               // negate once on the device before the per-sample broadcast
      hipLaunchKernelGGL(scale_matrix_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, lad, -1.0, lad, 0);
    }
    if (rpt == 8) BJX_SM(8); else if (rpt == 16) BJX_SM(16); else BJX_SM(32);
#undef BJX_SM
    BJX_CHECK_LAUNCH(ctx);
    }
  } else if (inverse && want_ladj) {
    hipLaunchKernelGGL(scale_matrix_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, lad, -1.0, lad, 0);
  }
  if (ladj_sum) {
    // scale.jl:35-36: the reference returns logabsdet(a) ONCE whatever the number of columns; like the vector
    // Scale the consistent batch * logabsdet(a) is returned unless BJX_REF_VECTOR_SCALE_LADJ asks for the reference's value
    const double mult = (flags & BJX_REF_VECTOR_SCALE_LADJ) ? 1.0 : (double)batch;
    hipLaunchKernelGGL(scale_matrix_sum_kernel, dim3(1), dim3(1), 0, ctx->stream, lad, mult, ladj_sum, accum);
    BJX_CHECK_LAUNCH(ctx);
  }
  return BJX_OK;
}

}  // namespace

BJX_API int bjx_vec_corr(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch,
                         uint32_t flags) {
  return matrix_entry<MK_VEC_CORR>(ctx, "bjx_vec_corr", dt, inverse, in, out, ladj_ps, ladj_sum, K, batch, flags);
}
BJX_API int bjx_corr(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch,
                     uint32_t flags) {
  return matrix_entry<MK_CORR>(ctx, "bjx_corr", dt, inverse, in, out, ladj_ps, ladj_sum, K, batch, flags);
}
BJX_API int bjx_pd(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch,
                   uint32_t flags) {
  return matrix_entry<MK_PD>(ctx, "bjx_pd", dt, inverse, in, out, ladj_ps, ladj_sum, K, batch, flags);
}
BJX_API int bjx_pd_vec(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t K, int64_t batch,
                       uint32_t flags) {
  return matrix_entry<MK_PD_VEC>(ctx, "bjx_pd_vec", dt, inverse, in, out, ladj_ps, ladj_sum, K, batch, flags);
}

/* Scale{<:AbstractMatrix}, scale.jl:14,17,35-36 */
// ------------------------------------------------------------------ parameter pullback of the matrix Scale (round 6)
// ā = sign·(G Xᵀ + (Σ_n ℓ̄_n) a⁻ᵀ)  (ext/BijectorsReverseDiffExt.jl:72-115; forward: G = ȳ, X = x, sign = +1; inverse: G = the input cotangent a⁻ᵀx̄,
// X = a⁻¹y, sign = −1).  G Xᵀ is a sum of `batch` outer products: the one dense contraction of the path with K = batch — on the matrix cores.
// A wave owns a 64 x 64 tile of the output (4 x 4 accumulators of v_mfma_*_16x16x4) and a slice of the batch.  Both operands of the
// instruction are indexed (lane % 16, lane / 16) = (row of the 16-block, column k of the 4-step), so a lane loads ONE element of G and
// one of X per MFMA operand straight from the column-major arrays (16 consecutive rows = 64 / 128 contiguous bytes per column): no
// staging, every byte of the two arrays read once per tile row / column (dim <= 64: once).  The next step's eight loads are in
// flight during the sixteen MFMAs of this one.  Partial tiles go to the context's workspace and are folded in a fixed order
// (no floating-point atomics: deterministic); the fold adds the log-det term from [A⁻¹] of the same factorisation kernel bjx_scale_matrix uses.
template <class T>
__global__ __launch_bounds__(256) void outer_sum_mfma_kernel(const T* __restrict__ G, const T* __restrict__ X, const T* __restrict__ lbar, int dim, int64_t batch,
                                                             int64_t cols_per_wave, int tiles_1d, T* __restrict__ parts, double* __restrict__ lparts) {
  using O = MfmaOps<T>;
  using acc_t = typename O::acc_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int ti = blockIdx.y / tiles_1d, tj = blockIdx.y % tiles_1d;
  const int64_t wg = (int64_t)blockIdx.x * 4 + wave;                 // this wave's slice of the batch
  const int64_t k_lo = wg * cols_per_wave;
  int64_t k_hi = k_lo + cols_per_wave;
  if (k_hi > batch) k_hi = batch;
  acc_t acc[4][4];
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = 0; J < 4; ++J) acc[I][J] = acc_t{0, 0, 0, 0};
  bool rg[4], rx[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) { rg[I] = 64 * ti + 16 * I + m < dim; rx[I] = 64 * tj + 16 * I + m < dim; }
  const T* gp = G + 64 * ti + m;
  const T* xp = X + 64 * tj + m;
  // S = 4 k-steps (16 columns) per trip, double-buffered: the 32 element loads of the NEXT trip are in flight during the 64 MFMAs of this one
  // (one step of look-ahead — 2 KiB per wave — left the kernel latency-bound at 37 % of the HBM peak; 8 KiB per wave and trip: see the row in profiles/).
  constexpr int S = 4;
  auto load = [&](int64_t k0, T (&a)[S][4], T (&b)[S][4]) {
#pragma unroll
    for (int s_ = 0; s_ < S; ++s_) {
      const int64_t k = k0 + 4 * s_ + q;
      const bool kok = k < k_hi;
      const int64_t off = k * dim;
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        a[s_][I] = (kok && rg[I]) ? gp[off + 16 * I] : T(0);
        b[s_][I] = (kok && rx[I]) ? xp[off + 16 * I] : T(0);
      }
    }
  };
  auto run = [&](const T (&a)[S][4], const T (&b)[S][4]) {
#pragma unroll
    for (int s_ = 0; s_ < S; ++s_)
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[I][J] = O::mfma(a[s_][I], b[s_][J], acc[I][J]);
  };
  double lsum = 0.0;
  T a0[S][4], b0[S][4], a1[S][4], b1[S][4];
  if (k_lo < k_hi) load(k_lo, a0, b0);
  for (int64_t k0 = k_lo; k0 < k_hi; k0 += 8 * S) {
    load(k0 + 4 * S, a1, b1);                                         // (past the slice: zeros, no access)
    run(a0, b0);
    load(k0 + 8 * S, a0, b0);
    run(a1, b1);
    if (lbar && blockIdx.y == 0 && m == 0) {
#pragma unroll
      for (int s_ = 0; s_ < 2 * S; ++s_)
        if (k0 + 4 * s_ + q < k_hi) lsum += (double)lbar[k0 + 4 * s_ + q];
    }
  }
  // partial tile of this wave: parts[(tile * nwaves + wg)][i_local][j_local], 64 x 64, j contiguous
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  T* out = parts + ((int64_t)blockIdx.y * nwaves + wg) * 4096;
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * I + O::row(q, r)) * 64 + 16 * J + m] = acc[I][J][r];
  if (lparts && blockIdx.y == 0) {
    lsum = group_sum<64>(lsum);
    if (lane == 0) lparts[wg] = lsum;
  }
}

// stage 1: thread (o, c) sums the waves [c·per, (c+1)·per) of entry o of a tile; stage 2 adds the C chunk sums, the log-det term and writes ā (column-major)
template <class T>
__global__ __launch_bounds__(256) void outer_sum_fold1_kernel(const T* __restrict__ parts, int64_t nwaves, int per, double* __restrict__ stage) {
  const int o = blockIdx.x * 256 + threadIdx.x;                      // 0 .. 4095 within the tile blockIdx.z
  const int c = blockIdx.y;
  const int64_t lo = (int64_t)c * per, hi = lo + per < nwaves ? lo + per : nwaves;
  const T* p = parts + (int64_t)blockIdx.z * nwaves * 4096 + o;
  double s0 = 0.0, s1 = 0.0;
  int64_t w = lo;
  for (; w + 1 < hi; w += 2) { s0 += (double)p[w * 4096]; s1 += (double)p[(w + 1) * 4096]; }
  if (w < hi) s0 += (double)p[w * 4096];
  stage[((int64_t)blockIdx.z * gridDim.y + c) * 4096 + o] = s0 + s1;
}
template <class T>
__global__ __launch_bounds__(256) void outer_sum_fold2_kernel(const double* __restrict__ stage, int C_, const double* __restrict__ lparts, int64_t nwaves, const T* __restrict__ W /*[dim][2 dim]: right half A⁻¹*/,
                                                              int dim, int tiles_1d, double sign, T* __restrict__ a_bar) {
  __shared__ double lred[4];
  double lsum_s = 0.0;
  if (lparts) {                                                         // Σ ℓ̄ from the waves' partial sums, fixed order (every block recomputes it: <= 16 KiB)
    double l = 0.0;
    for (int64_t w = threadIdx.x; w < nwaves; w += 256) l += lparts[w];
    l = group_sum<64>(l);
    if ((threadIdx.x & 63) == 0) lred[threadIdx.x >> 6] = l;
    __syncthreads();
    lsum_s = (lred[0] + lred[1]) + (lred[2] + lred[3]);
  }
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int tile = blockIdx.y, ti = tile / tiles_1d, tj = tile % tiles_1d;
  const int i = 64 * ti + o / 64, j = 64 * tj + o % 64;
  if (i >= dim || j >= dim) return;
  double s = 0.0;
  for (int c = 0; c < C_; ++c) s += stage[((int64_t)tile * C_ + c) * 4096 + o];
  if (lparts) s += lsum_s * (double)W[(int64_t)j * 2 * dim + dim + i];          // a⁻ᵀ[i][j] = A⁻¹[j][i]
  a_bar[(int64_t)j * dim + i] = (T)(sign * s);
}

template <class T>
int scale_matrix_vjp_params_impl(bjx_ctx* ctx, const T* a, const T* g, const T* x, const T* ladj_bar, double sign, T* a_bar, int64_t dim, int64_t batch) {
  const int tiles_1d = (int)((dim + 63) / 64), ntiles = tiles_1d * tiles_1d;
  // slices of the batch: enough waves to fill the chip twice over all tiles, at least 64 columns each
  int64_t blocks = ((int64_t)ctx->num_cu * 2 + ntiles - 1) / ntiles;
  if (blocks < 1) blocks = 1;
  int64_t cpw = (batch + blocks * 4 - 1) / (blocks * 4);
  if (cpw < 64) cpw = 64;
  cpw = (cpw + 31) / 32 * 32;
  blocks = (batch + cpw * 4 - 1) / (cpw * 4);
  if (blocks < 1) blocks = 1;
  const int64_t nwaves = blocks * 4;
  constexpr int C_ = 16;
  const size_t mat_bytes = ((size_t)dim * 2 * dim * sizeof(T) + 15) / 16 * 16;
  const size_t parts_bytes = (size_t)ntiles * nwaves * 4096 * sizeof(T);
  const size_t stage_bytes = (size_t)ntiles * C_ * 4096 * sizeof(double);
  const size_t lp_bytes = ((size_t)nwaves * sizeof(double) + 15) / 16 * 16;
  { const int rc = bjx_ensure_big_ws(ctx, mat_bytes + 16 + parts_bytes + stage_bytes + lp_bytes); if (rc) return rc; }
  char* ws = static_cast<char*>(ctx->big_ws);
  T* W = reinterpret_cast<T*>(ws);
  double* lad = reinterpret_cast<double*>(ws + mat_bytes);
  T* parts = reinterpret_cast<T*>(ws + mat_bytes + 16);
  double* stage = reinterpret_cast<double*>(ws + mat_bytes + 16 + parts_bytes);
  double* lparts = reinterpret_cast<double*>(ws + mat_bytes + 16 + parts_bytes + stage_bytes);
  if (ladj_bar) {                  // [A | I] -> [· | A⁻¹]: the factorisation of bjx_scale_matrix (LDS form where it fits)
    const size_t lds_p = dim <= 128 ? scale_prep_lds_bytes<T>(dim, 1) : 0;
    static const int use_wave2 = getenv("BJX_SCALE_PREP_WAVE") ? atoi(getenv("BJX_SCALE_PREP_WAVE")) : 1;
    if (use_wave2 && dim <= 64) {
      hipLaunchKernelGGL((scale_matrix_prep_wave_kernel<T, true>), dim3(1), dim3(64), 0, ctx->stream, a, W, (int)dim, lad);
    } else if (lds_p) {
      bjx_allow_big_lds(scale_matrix_prep_lds_kernel<T>, lds_p);
      hipLaunchKernelGGL((scale_matrix_prep_lds_kernel<T>), dim3(1), dim3(512), lds_p, ctx->stream, a, W, (int)dim, 1, lad);
    } else {
      hipLaunchKernelGGL((scale_matrix_prep_kernel<T>), dim3(1), dim3(256), 0, ctx->stream, a, W, (int)dim, 1, lad);
    }
    BJX_CHECK_LAUNCH(ctx);
  }
  {
    BjxProf prof_(ctx);
    hipLaunchKernelGGL((outer_sum_mfma_kernel<T>), dim3((unsigned)blocks, (unsigned)ntiles), dim3(256), 0, ctx->stream, g, x, ladj_bar, (int)dim, batch, cpw, tiles_1d, parts,
                       ladj_bar ? lparts : nullptr);
    BJX_CHECK_LAUNCH(ctx);
  }
  const int per = (int)((nwaves + C_ - 1) / C_);
  hipLaunchKernelGGL((outer_sum_fold1_kernel<T>), dim3(16, C_, (unsigned)ntiles), dim3(256), 0, ctx->stream, parts, nwaves, per, stage);
  BJX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL((outer_sum_fold2_kernel<T>), dim3(16, (unsigned)ntiles), dim3(256), 0, ctx->stream, stage, C_, ladj_bar ? lparts : nullptr, nwaves, W, (int)dim, tiles_1d, sign, a_bar);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

BJX_API int bjx_scale_matrix_vjp_params(bjx_ctx* ctx, bjx_dtype dt, const void* a, const void* g, const void* x, const void* ladj_bar, double sign, void* a_bar,
                                        int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_scale_matrix_vjp_params: bad size");
  BJX_REQUIRE(ctx, dim <= 1024, BJX_ERR_UNSUPPORTED, "bjx_scale_matrix_vjp_params: dim = %lld: the factorisation behind a⁻ᵀ stops at 1024 rows", (long long)dim);
  BJX_REQUIRE(ctx, a && a_bar && ((g && x) || batch == 0), BJX_ERR_ARG, "bjx_scale_matrix_vjp_params: null pointer");
  if (dt == BJX_F32) return scale_matrix_vjp_params_impl<float>(ctx, (const float*)a, (const float*)g, (const float*)x, (const float*)ladj_bar, sign, (float*)a_bar, dim, batch);
  if (dt == BJX_F64) return scale_matrix_vjp_params_impl<double>(ctx, (const double*)a, (const double*)g, (const double*)x, (const double*)ladj_bar, sign, (double*)a_bar, dim, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_scale_matrix_vjp_params: bad dtype %d", (int)dt);
}

BJX_API int bjx_scale_matrix(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* a, const void* in, void* out, void* ladj_ps, double* ladj_sum,
                             int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_scale_matrix: bad size");
  BJX_REQUIRE(ctx, a, BJX_ERR_ARG, "bjx_scale_matrix: null matrix");
  BJX_REQUIRE(ctx, (in && (out || ladj_ps || ladj_sum)) || batch == 0, BJX_ERR_ARG, "bjx_scale_matrix: null pointer");
  if (dt == BJX_F32) return scale_matrix_impl<float>(ctx, inverse, (const float*)a, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return scale_matrix_impl<double>(ctx, inverse, (const double*)a, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_scale_matrix: bad dtype %d", (int)dt);
}

namespace {
template <class T>
int scale_matrix_chain_impl(bjx_ctx* ctx, int inverse, const T* a, const bjx_op* ops, int n_ops, const T* in, T* out, T* ladj_ps, int64_t dim, int64_t batch, uint32_t flags) {
  MatPre<T> pre{};
  pre.n = n_ops;
  for (int i = 0; i < n_ops; ++i) {
    const bjx_op& o = ops[i];
    BJX_REQUIRE(ctx, o.kind == BJX_OP_EXP || o.kind == BJX_OP_LOG || o.kind == BJX_OP_SHIFT || o.kind == BJX_OP_SCALE || o.kind == BJX_OP_SCALE_INV, BJX_ERR_UNSUPPORTED,
                "bjx_scale_matrix_chain: stage %d of kind %d (served: exp, log, Shift, Scale, its inverse)", i, (int)o.kind);
    const bool has_p = o.kind == BJX_OP_SHIFT || o.kind == BJX_OP_SCALE || o.kind == BJX_OP_SCALE_INV;
    BJX_REQUIRE(ctx, !has_p || o.param_len == 1 || o.param_len == dim, BJX_ERR_SHAPE, "bjx_scale_matrix_chain: stage %d has %d parameters for %lld rows", i, (int)o.param_len, (long long)dim);
    BJX_REQUIRE(ctx, !has_p || o.param_len == 1 || o.v0, BJX_ERR_ARG, "bjx_scale_matrix_chain: stage %d: one value per row needs the device vector", i);
    pre.kind[i] = o.kind;
    pre.s[i] = (T)o.p0;
    pre.v[i] = has_p && o.param_len == dim && dim > 1 ? static_cast<const T*>(o.v0) : nullptr;
    if (has_p && o.param_len == 1 && o.v0) return bjx_fail(ctx, BJX_ERR_UNSUPPORTED, "bjx_scale_matrix_chain: stage %d: a scalar parameter on the device", i);
  }
  return scale_matrix_impl<T>(ctx, inverse, a, in, out, ladj_ps, nullptr, dim, batch, flags, &pre);
}
}  // namespace

/* src/transformed_distribution.jl:164-169 with a full-covariance base: the inverse chain, the shift by the mean, the whitening and the density in one pass */
BJX_API int bjx_scale_matrix_chain(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* a, const bjx_op* ops, int n_ops, const void* in, void* out, void* ladj_ps,
                                   int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_scale_matrix_chain: bad size");
  BJX_REQUIRE(ctx, a && (ops || n_ops == 0) && n_ops >= 0, BJX_ERR_ARG, "bjx_scale_matrix_chain: null pointer");
  BJX_REQUIRE(ctx, n_ops <= 4, BJX_ERR_UNSUPPORTED, "bjx_scale_matrix_chain: at most four stages in front of the matrix (%d given)", n_ops);
  BJX_REQUIRE(ctx, (in && (out || ladj_ps)) || batch == 0, BJX_ERR_ARG, "bjx_scale_matrix_chain: null pointer");
  if (dt == BJX_F32) return scale_matrix_chain_impl<float>(ctx, inverse, (const float*)a, ops, n_ops, (const float*)in, (float*)out, (float*)ladj_ps, dim, batch, flags);
  if (dt == BJX_F64) return scale_matrix_chain_impl<double>(ctx, inverse, (const double*)a, ops, n_ops, (const double*)in, (double*)out, (double*)ladj_ps, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_scale_matrix_chain: bad dtype %d", (int)dt);
}
