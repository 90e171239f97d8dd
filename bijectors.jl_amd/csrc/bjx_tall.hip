// bjx_tall.hip — OrderedBijector / SimplexBijector on TALL columns (ordered.jl:24-80, simplex.jl:28-143)
//
// The quad frames of bjx_seq.hip (quad_stream_kernel) hold a column in the registers of 4 lanes and stop at 128 rows (Float32);
// beyond, round 2-3 used walkers with ONE LANE PER COLUMN over an LDS tile (seq_wave_kernel, seq_chunk_kernel).  Their memory
// side is 64 runs of 256 bytes per wave instruction group, one run per column, starting wherever the column starts: counters
// (profiles/r03_tall_counters.md) show 1.34-1.42 x the algorithmic bytes fetched when a column is not a whole number of cache
// lines (K = 200, 1000) and 57-66 % of the wave cycles spent in issue stalls, 30-43 % of the HBM peak.
//
// Here a column is held by G = ceil(rows / RPL) lanes, RPL = 32 (Float32) / 16 (Float64) rows = 128 bytes each, ANY G up to 64:
//   * a wave instruction group owns CPS = 64/G whole columns = ONE contiguous run of input and ONE of output (8 KiB): 16-byte
//     coalesced accesses, only the two ends of a run can share a cache line with a neighbour;
//   * the run is re-dealt through an LDS strip of 64 slots (one per lane, 144-byte pitch: the 16-byte slot reads are bank-conflict
//     free), each lane keeps its 128 bytes of the column in registers;
//   * the running value of the column (Σ_{j<k} x_j, Σ exp x_j, the stick-breaking remainder) is accumulated IN THE REFERENCE'S
//     ORDER: the G lanes take turns — G-1 rounds in which every lane re-runs its chain from the carry it holds and then takes
//     its left neighbour's end value; lane t's carry is final after round t-1.  (A parallel scan changes the association of the
//     Float32 sum; where 1 - Σ is small — the tail of every tall simplex — that moves z_k = x_k / (1 - Σ) by more than the parity
//     bar, see the note in bjx_seq.hip.)  The rounds cost (G-1)·RPL chain operations per lane: one add per row for Ordered and the
//     Simplex transform, four for the Simplex inverse;
//   * log(K-1-r) comes from an LDS table in slot layout, built once per block; a wave walks `nsteps` column sets.
//
// Test infrastructure note: parity against oracle/ is in tests/test_gpu_parity.py (tall columns) and tests/test_gpu_small_shapes.py.
#include <cstdlib>

#include "bjx_internal.h"

namespace {
using namespace bjx;

#include "bjx_seqops.h"   // simplex_t_partials

struct TallGeom {
  int gl;    // lane inside its column group
  int G;     // lanes per column
  int nin;   // rows of this lane that exist in the input (0..RPL)
  int nout;  // rows of this lane that exist in the output
  int iK;    // index, inside this lane, of the column's final row (rows-1); >= RPL on the lanes to its left, < 0 to its right
  int scan;  // Simplex inverse: carry by an affine scan (1) or by take-turns rounds in the reference's order (0)
};

// s -> A s + B maps composed over the G lanes of a column group: on return (A, B) of lane gl is the composition of the maps of
// lanes 0 .. gl (lane 0 applied first).  Guarded shuffles: G is any value.
template <class T> __device__ __forceinline__ void tall_affine_scan_up(T& A, T& B, int gl, int G) {
  for (int d = 1; d < G; d <<= 1) {
    const T Al = __shfl_up(A, d, 64), Bl = __shfl_up(B, d, 64);
    if (gl >= d) { B = A * Bl + B; A = A * Al; }
  }
}
// the same from the last lane down: (A, B) of lane gl = composition of the maps of lanes G-1 .. gl (lane G-1 applied first)
template <class T> __device__ __forceinline__ void tall_affine_scan_down(T& A, T& B, int gl, int G) {
  for (int d = 1; d < G; d <<= 1) {
    const T Ar = __shfl_down(A, d, 64), Br = __shfl_down(B, d, 64);
    if (gl + d < G) { B = A * Br + B; A = A * Ar; }
  }
}

template <class T> __device__ __forceinline__ T lane_left(T v) { return __shfl_up(v, 1, 64); }

// ---------------------------------------------------------------- the maps, RPL rows per lane in registers
template <class T> struct TOrderedFwd {                  // ordered.jl:36-49, :80
  static constexpr bool USES_LOGK = false;
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], const TallGeom& g, const T*) const {
    using F = Fast<T>;
    T l = T(0);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool live = i < g.nin, first = i == 0 && g.gl == 0;
      const T v = x[i];
      l += (live && !first) ? v : T(0);                                // logabsdetjac = Σ_{k>=2} x_k
      const T e = first ? v : F::exp(v);                               // y_1 = x_1 ; y_k = y_{k-1} + exp(x_k)
      x[i] = live ? e : T(0);
    }
    T carry = T(0);                                                    // 0 + x_1 is exact: the chain starts like y_1 = x_1
    for (int t = 1; t < g.G; ++t) {
      T run = carry;
#pragma unroll
      for (int i = 0; i < RPL; ++i) run += x[i];
      const T bc = lane_left(run);
      carry = g.gl == 0 ? T(0) : bc;
    }
    T run = carry;
#pragma unroll
    for (int i = 0; i < RPL; ++i) { run += x[i]; x[i] = run; }         // rows beyond the column: not stored, and masked when the strip is read again
    return l;
  }
};

template <class T> struct TOrderedInv {                  // ordered.jl:63-77 ; interface.jl:276-281
  static constexpr bool USES_LOGK = false;
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], const TallGeom& g, const T*) const {
    using F = Fast<T>;
    const T left = lane_left(x[RPL - 1]);                              // y of the row before my first one (the lanes to the left are full)
    T l = T(0), prev = left;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool live = i < g.nin, first = i == 0 && g.gl == 0;
      const T y = x[i];
      const T o = first ? y : F::log(y - prev);                        // x_1 = y_1 ; x_k = log(y_k - y_{k-1})
      l -= (live && !first) ? o : T(0);
      prev = y;
      x[i] = o;
    }
    return l;
  }
};

template <class T, bool LADJ> struct TSimplexFwd {       // simplex.jl:47-64 + :122-138 (the arithmetic of QSimplexFwd in bjx_seq.hip)
  static constexpr bool USES_LOGK = true;
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], const TallGeom& g, const T* lk) const {
    using F = Fast<T>;
    constexpr int V = Vec16<T>::N;
    const T e = Num<T>::eps, c2 = T(1) - 2 * e, E = T(1) + e;
    // x_K (row K-1) and the padding are not inputs of the map (:47-64 read x_1..x_{K-1})
#pragma unroll
    for (int i = 0; i < RPL; ++i) x[i] = i < g.iK ? x[i] : T(0);
    T carry = T(0);
    for (int t = 1; t < g.G; ++t) {                                    // rounds: only the running sum
      T run = carry;
#pragma unroll
      for (int i = 0; i < RPL; ++i) run += x[i];
      const T bc = lane_left(run);
      carry = g.gl == 0 ? T(0) : bc;
    }
    T lp = T(0), s = carry;                                            // s = Σ_{j<k} x_j in the reference's order
    T Pp = T(1), mp = T(1);
#pragma unroll
    for (int q = 0; q < RPL / V; ++q) {
      const Pack<T, V> lkq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const typename Vec16<T>::type*>(lk + q * V));
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int i = q * V + j;
        const T xk = x[i];
        const bool first = i == 0 && g.gl == 0;
        const T a = first ? xk * c2 + e : (xk + e) * c2;               // :53 / :58
        const T dn = first ? T(1) : E - s;
        const T o = F::log2(a * F::rcp(dn - a)) * Num<T>::log2 + lkq.v[j];   // logit(z) + log(K-k)
        if (LADJ) {
          // term_k = max(z,ε)·max(1-z,ε)·m, z = x_k/m, m = max(1-Σ,ε) (:130-135) = max(x_k, εm)·max(m - x_k, εm)/m ;
          // two rows share one reciprocal and one logarithm (each term >= ε²)
          const T m = d_max(T(1) - s, e), em = e * m;
          const T P = d_max(xk, em) * d_max(m - xk, em);
          if (i & 1) lp += F::log2(Pp * F::rcp(mp * m) * P);
          else { Pp = P; mp = m; }
        }
        x[i] = o;
        s += xk;
      }
      __builtin_amdgcn_sched_barrier(0);                               // V rows in flight, not 32: the scheduler otherwise spills under the 128-VGPR target
    }
    if (LADJ) {
      // The rows without a term (x_K and the padding; x = 0 there, Σ no longer moves) each put the factor ε·m_end into the products
      // above, m_end = max(1 - Σ_end, ε): taken out here in one step instead of three selects per row
      const int nd = g.iK >= RPL ? 0 : (g.iK < 0 ? RPL : RPL - g.iK);
      lp -= T(nd) * F::log2(e * d_max(T(1) - s, e));
    }
    // Julia's max(NaN, ε) is NaN (v_max drops it): a NaN among x_1..x_{K-1} makes the reference's log-det NaN
    return s != s ? s : -lp * Num<T>::log2;
  }
};

template <class T, bool LADJ> struct TSimplexInv {       // simplex.jl:102-120 ; log-det = -logabsdetjac(b, x_out)  (QSimplexInv's arithmetic)
  static constexpr bool USES_LOGK = true;
  // FAST: clamp(v, 0, 1) as one v_med3_f32; med3 does not keep a NaN (the reference's clamp does), so only for waves without NaN
  template <bool FAST> static __device__ __forceinline__ T cl01(T v) {
    if constexpr (FAST) return d_med3(v, T(0), T(1));
    else return d_clamp(v, T(0), T(1));
  }
  template <bool FAST, int RPL> static __device__ __forceinline__ T rounds(T (&x)[RPL], const TallGeom& g) {
    using F = Fast<T>;
    const T e = Num<T>::eps, E = T(1) + e;
    const T e0 = e * (T(1) / (T(1) - 2 * e));
    T carry = T(0);
    if (g.scan) {
      // Σ entering this lane by an AFFINE SCAN: without its clamp a row is Σ' = (1 - a_k) Σ + ((1+ε) a_k - ε), a_k = z_k/(1-2ε) — one
      // composition per lane (4 operations per row, the price of ONE round) and log2 G shuffle steps instead of G-1 rounds of the
      // four-operation recurrence.  Not the reference's association, and a clamped row moves the carry by <= ε; both are harmless
      // HERE (unlike in the transform): the recurrence contracts — an error δ in Σ becomes (1 - a_k) δ after the row — the outputs
      // carry it as a_k δ.  The log-det terms see it relative to 1 - Σ, and a clamped row breaks the affine model: not the default
      // (see bjx_tall_stream below).
      T A = T(1), B = T(0);
#pragma unroll
      for (int i = 0; i < RPL; ++i) {
        const bool first = i == 0 && g.gl == 0;
        const T Ar = first ? T(1) : T(1) - x[i];
        const T Br = first ? x[i] - e0 : E * x[i] - e;
        B = Ar * B + Br;
        A = Ar * A;
      }
      tall_affine_scan_up(A, B, g.gl, g.G);
      const T bc = lane_left(B);                                       // the composition of the lanes to my left, applied to Σ = 0
      carry = g.gl == 0 ? T(0) : bc;
    } else {
      for (int t = 1; t < g.G; ++t) {                                  // rounds: only the recurrence Σ -> x_k -> Σ
        T s = carry;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
          const bool first = i == 0 && g.gl == 0;
          const T xi = first ? cl01<FAST>(x[i] - e0) : cl01<FAST>((E - s) * x[i] - e);
          s += xi;
        }
        const T bc = lane_left(s);
        carry = g.gl == 0 ? T(0) : bc;
      }
    }
    T lp = T(0), s = carry;
    T Pp = T(1), mp = T(1);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool first = i == 0 && g.gl == 0;
      const bool rowK = i == g.iK;
      const T xi = first ? cl01<FAST>(x[i] - e0)                       // :109
                         : cl01<FAST>((E - s) * x[i] - e);             // :113   (x[i] = 0 on the rows without input: xi = 0)
      if (LADJ) {
        const T m = d_max(T(1) - s, e), em = e * m;
        const T P = d_max(xi, em) * d_max(m - xi, em);
        if (i & 1) lp += F::log2(Pp * F::rcp(mp * m) * P);
        else { Pp = P; mp = m; }
      }
      x[i] = rowK ? cl01<FAST>(T(1) - s) : xi;                         // :116
      s += xi;
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (LADJ) {                                                        // rows without a term (x_i = 0, Σ fixed): the factor ε·m_end each, taken out here (see TSimplexFwd)
      const int nd = g.iK >= RPL ? 0 : (g.iK < 0 ? RPL : RPL - g.iK);
      lp -= T(nd) * F::log2(e * d_max(T(1) - s, e));
    }
    if (!FAST && s != s) return s;                                     // Julia's max(NaN, ε) is NaN: the log-det of a poisoned column is NaN
    return lp * Num<T>::log2;
  }
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], const TallGeom& g, const T* lk) const {
    constexpr int V = Vec16<T>::N;
    const T e = Num<T>::eps;
    const T inv12e = T(1) / (T(1) - 2 * e);
    // z_k = logistic(y_k - log(K-k)) with LogExpFunctions' exact 0/1 saturation; rows without an input (row K-1, padding) -> 0
    T poison = T(0);
#pragma unroll
    for (int q = 0; q < RPL / V; ++q) {
      const Pack<T, V> lkq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const typename Vec16<T>::type*>(lk + q * V));
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int i = q * V + j;
        const T z = f_logistic(x[i] - lkq.v[j]);
        const T zz = i < g.iK ? z : T(0);
        poison = zz * T(0) + poison;                                   // NaN iff some y_k is NaN (z is in [0, 1] otherwise)
        x[i] = zz * inv12e;                                            // z_k / (1 - 2ε): the factor both :109 and :113 apply
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (__builtin_amdgcn_ballot_w64(poison != poison) == 0) return rounds<true, RPL>(x, g);
    return rounds<false, RPL>(x, g);
  }
};

// ---------------------------------------------------------------- the kernel
template <class T> struct TallCfg {
  static constexpr int V = Vec16<T>::N, RPL = 128 / (int)sizeof(T), SLOT = RPL + V, NQ = RPL / V, WPB = 4;
};

// (column, row) of a run element -> element index in the strip; RPL is a power of two
template <class T> __device__ __forceinline__ int tall_slot(int dc, int dr, int G) {
  constexpr int RPL = TallCfg<T>::RPL, SLOT = TallCfg<T>::SLOT;
  return (dc * G + dr / RPL) * SLOT + (dr & (RPL - 1));
}
// advance (dc, dr) by `step` elements of a run of columns with `rows` rows (rows > step / 4)
__device__ __forceinline__ void tall_advance(int& dc, int& dr, int step, int rows) {
  dr += step;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (dr >= rows) { dr -= rows; ++dc; } }
}

// the same for a step of at most `rows` elements (one wrap)
__device__ __forceinline__ void tall_advance1(int& dc, int& dr, int step, int rows) {
  dr += step;
  if (dr >= rows) { dr -= rows; ++dc; }
}

template <class T, class Op, bool VIN, bool VOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void tall_stream_kernel(const Op op, const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps,
                                                          int rows_in, int rows_out, int64_t batch, int G, int nsteps,
                                                          int accumulate, double* __restrict__ partials, int inv_scan) {
  using C = TallCfg<T>;
  constexpr int V = C::V, RPL = C::RPL, SLOT = C::SLOT, NQ = C::NQ, WPB = C::WPB;
  using VT = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[WPB];
  T* strips = reinterpret_cast<T*>(smem);                              // [WPB][64 slots]
  T* lktab = strips + WPB * 64 * SLOT;                                 // [G slots] when the map uses log(K-1-r)
  // strip position of every element of a run, for the sides that move 4-byte elements (a column is not whole 16-byte packs):
  // built once per block — computing (column, row) -> slot per element and step cost ~10 VALU per element (+25 % per step)
  unsigned short* tab_in = reinterpret_cast<unsigned short*>(lktab + (Op::USES_LOGK ? G * SLOT : 0));
  unsigned short* tab_out = tab_in + (VIN ? 0 : ((64 / G) * rows_in + 1) / 2 * 2);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the buffer descriptors below live in SGPRs
  const int rows = rows_in > rows_out ? rows_in : rows_out;
  const int CPS = 64 / G;                                              // columns of a wave instruction group
  const int cg = lane / G, gl = lane - cg * G;
  const bool idle = cg >= CPS;                                         // 64 - CPS·G lanes have no column
  T* st = strips + wave * 64 * SLOT;
  // The strip is zeroed once so that the first step reads no uninitialised LDS in the positions no run element maps to (rows beyond
  // the column, idle lanes); later steps find the previous step's values there, and every map masks the rows that do not exist.
  for (int i = lane; i < 64 * SLOT; i += 64) st[i] = T(0);
  if (Op::USES_LOGK) {
    for (int r = threadIdx.x; r < G * RPL; r += 64 * WPB)              // log(K-1-r), simplex.jl:35,41 (precise logs, once per block)
      lktab[(r / RPL) * SLOT + (r & (RPL - 1))] = r < rows - 1 ? d_log(T(rows - 1 - r)) : T(0);
  }
  if (!VIN) {
    int dc = 0, dr = 0;
    tall_advance(dc, dr, (int)threadIdx.x, rows_in);
    for (int e = threadIdx.x; e < CPS * rows_in; e += 64 * WPB) { tab_in[e] = (unsigned short)tall_slot<T>(dc, dr, G); tall_advance(dc, dr, 64 * WPB, rows_in); }
  }
  if (!VOUT) {
    int dc = 0, dr = 0;
    tall_advance(dc, dr, (int)threadIdx.x, rows_out);
    for (int e = threadIdx.x; e < CPS * rows_out; e += 64 * WPB) { tab_out[e] = (unsigned short)tall_slot<T>(dc, dr, G); tall_advance(dc, dr, 64 * WPB, rows_out); }
  }
  __syncthreads();
  TallGeom g;
  g.gl = gl; g.G = G; g.scan = inv_scan;
  { const int a = rows_in - gl * RPL; g.nin = idle ? 0 : (a < 0 ? 0 : (a > RPL ? RPL : a)); }
  { const int a = rows_out - gl * RPL; g.nout = idle ? 0 : (a < 0 ? 0 : (a > RPL ? RPL : a)); }
  g.iK = idle ? -1 : rows - 1 - gl * RPL;
  const T* lk = lktab + gl * SLOT;
  // VIN / VOUT: every run starts on a 16-byte boundary and a column is whole packs — a pack of the run is a pack of one lane slot.
  // Strip positions of this lane's packs (the same in every step: a run starts at a column boundary); -1 = beyond the run.
  int wa[VIN ? NQ : 1], ra[VOUT ? NQ : 1];
  if (VIN) {
    int dc = 0, dr = 0;
    tall_advance(dc, dr, lane * V, rows_in);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { wa[q] = dc < CPS ? tall_slot<T>(dc, dr, G) : -1; tall_advance(dc, dr, 64 * V, rows_in); }
  }
  if (VOUT) {
    int dc = 0, dr = 0;
    tall_advance(dc, dr, lane * V, rows_out);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { ra[q] = dc < CPS ? tall_slot<T>(dc, dr, G) : -1; tall_advance(dc, dr, 64 * V, rows_out); }
  }
  double acc = 0.0;
  const int64_t set0 = ((int64_t)blockIdx.x * WPB + wave) * nsteps;
  for (int sidx = 0; sidx < nsteps; ++sidx) {
    const int64_t colw = (set0 + sidx) * CPS;
    if (colw >= batch) break;                                          // wave-uniform
    const int ncol = (int)((batch - colw) < CPS ? (batch - colw) : CPS);
    const int nel_in = ncol * rows_in, nel_out = ncol * rows_out;
    const auto r_in = bjx_make_rsrc(in + colw * rows_in, (uint32_t)((size_t)nel_in * sizeof(T)));
    __builtin_amdgcn_wave_barrier();                                   // the previous step's stores have read the strip
    if (VIN && ncol == CPS) {
      Pack<T, V> raw[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) raw[q] = buf_load_pack<T, V>(r_in, wa[q] >= 0 ? (lane + 64 * q) * V * (int)sizeof(T) : 0x7fffff00);
#pragma unroll
      for (int q = 0; q < NQ; ++q) { if (wa[q] >= 0) *reinterpret_cast<VT*>(st + wa[q]) = __builtin_bit_cast(VT, raw[q]); }
    } else if (!VIN) {
      // a column is not whole 16-byte packs: element accesses, 8 in flight, strip positions from the table
      for (int e0 = 0; e0 < nel_in; e0 += 64 * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = buf_load_pack<T, 1>(r_in, (e0 + 64 * u + lane) * (int)sizeof(T)).v[0];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + 64 * u + lane;
          if (e < nel_in) st[tab_in[e]] = v[u];
        }
      }
    } else {
      // the ragged last set of a launch with whole-pack columns
      int dc = 0, dr = 0;
      tall_advance(dc, dr, lane, rows_in);
      for (int e0 = 0; e0 < nel_in; e0 += 64 * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = buf_load_pack<T, 1>(r_in, (e0 + 64 * u + lane) * (int)sizeof(T)).v[0];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (e0 + 64 * u + lane < nel_in) st[tall_slot<T>(dc, dr, G)] = v[u];
          tall_advance1(dc, dr, 64, rows_in);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    T xv[RPL];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const VT*>(st + lane * SLOT + q * V));
#pragma unroll
      for (int j = 0; j < V; ++j) xv[q * V + j] = pq.v[j];
    }
    T l = op.template run<RPL>(xv, g, lk);
    // Σ over the G lanes of the column, fixed order (a tree towards the group's first lane)
    for (int d = 1; d < G; d <<= 1) {
      const T o = __shfl_down(l, d, 64);
      if (gl + d < G) l += o;
    }
    if (!idle && gl == 0 && cg < ncol) {
      const int64_t col = colw + cg;
      if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + l : l;
      acc += (double)l;
    }
    if (out) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        Pack<T, V> pq;
#pragma unroll
        for (int j = 0; j < V; ++j) pq.v[j] = xv[q * V + j];
        *reinterpret_cast<VT*>(st + lane * SLOT + q * V) = __builtin_bit_cast(VT, pq);
      }
      __builtin_amdgcn_wave_barrier();
      const auto r_out = bjx_make_rsrc(out + colw * rows_out, (uint32_t)((size_t)nel_out * sizeof(T)));
      if (VOUT && ncol == CPS) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (ra[q] >= 0) {
            const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const VT*>(st + ra[q]));
            buf_store_pack<T, V>(r_out, (lane + 64 * q) * V * (int)sizeof(T), pq);
          }
        }
      } else if (!VOUT) {
        for (int e0 = 0; e0 < nel_out; e0 += 64 * 4) {
          T v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int e = e0 + 64 * u + lane; v[u] = e < nel_out ? st[tab_out[e]] : T(0); }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            Pack<T, 1> p1;
            p1.v[0] = v[u];
            buf_store_pack<T, 1>(r_out, (e0 + 64 * u + lane) * (int)sizeof(T), p1);       // beyond the run: outside the descriptor, dropped
          }
        }
      } else {
        int dc = 0, dr = 0;
        tall_advance(dc, dr, lane, rows_out);
        for (int e0 = 0; e0 < nel_out; e0 += 64) {
          if (e0 + lane < nel_out) {
            Pack<T, 1> p1;
            p1.v[0] = st[tall_slot<T>(dc, dr, G)];
            buf_store_pack<T, 1>(r_out, (e0 + lane) * (int)sizeof(T), p1);
          }
          tall_advance1(dc, dr, 64, rows_out);
        }
      }
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// ---------------------------------------------------------------- the run <-> strip moves as pieces (kernels with several runs)
// strip position of every element of a run of CPS columns of `rows` rows; all threads of the block
template <class T> __device__ __forceinline__ void tall_build_table(unsigned short* tab, int rows, int G, int CPS) {
  int dc = 0, dr = 0;
  tall_advance(dc, dr, (int)threadIdx.x, rows);
  for (int e = threadIdx.x; e < CPS * rows; e += 256) { tab[e] = (unsigned short)tall_slot<T>(dc, dr, G); tall_advance(dc, dr, 256, rows); }
}
// strip positions of this lane's 16-byte packs of a run whose columns are whole packs; -1 = beyond the run
template <class T> __device__ __forceinline__ void tall_pack_positions(int (&pos)[TallCfg<T>::NQ], int lane, int rows, int G, int CPS) {
  constexpr int V = TallCfg<T>::V, NQ = TallCfg<T>::NQ;
  int dc = 0, dr = 0;
  tall_advance(dc, dr, lane * V, rows);
#pragma unroll
  for (int q = 0; q < NQ; ++q) { pos[q] = dc < CPS ? tall_slot<T>(dc, dr, G) : -1; tall_advance(dc, dr, 64 * V, rows); }
}
// global run of ncol columns -> strip.  VEC: whole-pack columns (pos from tall_pack_positions); else element accesses through `tab`
template <class T, bool VEC>
__device__ __forceinline__ void tall_run_load(T* st, const T* __restrict__ run, int ncol, int rows, int G, int CPS, int lane,
                                              const int (&pos)[TallCfg<T>::NQ], const unsigned short* tab) {
  constexpr int V = TallCfg<T>::V, NQ = TallCfg<T>::NQ;
  using VT = typename Vec16<T>::type;
  const int nel = ncol * rows;
  const auto rs = bjx_make_rsrc(run, (uint32_t)((size_t)nel * sizeof(T)));
  if (VEC && ncol == CPS) {
    Pack<T, V> raw[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) raw[q] = buf_load_pack<T, V>(rs, pos[q] >= 0 ? (lane + 64 * q) * V * (int)sizeof(T) : 0x7fffff00);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { if (pos[q] >= 0) *reinterpret_cast<VT*>(st + pos[q]) = __builtin_bit_cast(VT, raw[q]); }
  } else if (!VEC) {
    for (int e0 = 0; e0 < nel; e0 += 64 * 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = buf_load_pack<T, 1>(rs, (e0 + 64 * u + lane) * (int)sizeof(T)).v[0];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 64 * u + lane;
        if (e < nel) st[tab[e]] = v[u];
      }
    }
  } else {                                                             // the ragged last set of a launch with whole-pack columns
    int dc = 0, dr = 0;
    tall_advance(dc, dr, lane, rows);
    for (int e0 = 0; e0 < nel; e0 += 64) {
      const T v = buf_load_pack<T, 1>(rs, (e0 + lane) * (int)sizeof(T)).v[0];
      if (e0 + lane < nel) st[tall_slot<T>(dc, dr, G)] = v;
      tall_advance1(dc, dr, 64, rows);
    }
  }
}
template <class T, bool VEC>
__device__ __forceinline__ void tall_run_store(const T* st, T* __restrict__ run, int ncol, int rows, int G, int CPS, int lane,
                                               const int (&pos)[TallCfg<T>::NQ], const unsigned short* tab) {
  constexpr int V = TallCfg<T>::V, NQ = TallCfg<T>::NQ;
  using VT = typename Vec16<T>::type;
  const int nel = ncol * rows;
  const auto rs = bjx_make_rsrc(run, (uint32_t)((size_t)nel * sizeof(T)));
  if (VEC && ncol == CPS) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (pos[q] >= 0) {
        const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const VT*>(st + pos[q]));
        buf_store_pack<T, V>(rs, (lane + 64 * q) * V * (int)sizeof(T), pq);
      }
    }
  } else if (!VEC) {
    for (int e0 = 0; e0 < nel; e0 += 64 * 4) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = e0 + 64 * u + lane; v[u] = e < nel ? st[tab[e]] : T(0); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        Pack<T, 1> p1;
        p1.v[0] = v[u];
        buf_store_pack<T, 1>(rs, (e0 + 64 * u + lane) * (int)sizeof(T), p1);            // beyond the run: outside the descriptor, dropped
      }
    }
  } else {
    int dc = 0, dr = 0;
    tall_advance(dc, dr, lane, rows);
    for (int e0 = 0; e0 < nel; e0 += 64) {
      if (e0 + lane < nel) {
        Pack<T, 1> p1;
        p1.v[0] = st[tall_slot<T>(dc, dr, G)];
        buf_store_pack<T, 1>(rs, (e0 + lane) * (int)sizeof(T), p1);
      }
      tall_advance1(dc, dr, 64, rows);
    }
  }
}
// this lane's slot -> registers; rows >= n read as zero (the strip is shared by runs of different heights)
template <class T> __device__ __forceinline__ void tall_slot_read(const T* st, int lane, T (&xv)[TallCfg<T>::RPL], int n) {
  constexpr int V = TallCfg<T>::V, NQ = TallCfg<T>::NQ, SLOT = TallCfg<T>::SLOT;
  using VT = typename Vec16<T>::type;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const VT*>(st + lane * SLOT + q * V));
#pragma unroll
    for (int j = 0; j < V; ++j) xv[q * V + j] = q * V + j < n ? pq.v[j] : T(0);
  }
}
template <class T> __device__ __forceinline__ void tall_slot_write(T* st, int lane, const T (&xv)[TallCfg<T>::RPL]) {
  constexpr int V = TallCfg<T>::V, NQ = TallCfg<T>::NQ, SLOT = TallCfg<T>::SLOT;
  using VT = typename Vec16<T>::type;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    Pack<T, V> pq;
#pragma unroll
    for (int j = 0; j < V; ++j) pq.v[j] = xv[q * V + j];
    *reinterpret_cast<VT*>(st + lane * SLOT + q * V) = __builtin_bit_cast(VT, pq);
  }
}
// exclusive sums over the G lanes of a column group, ascending / descending lane order (G any value: guarded shuffles)
template <class T> __device__ __forceinline__ T tall_excl_up(T v, int gl, int G) {
  T inc = v;
  for (int d = 1; d < G; d <<= 1) { const T t = __shfl_up(inc, d, 64); if (gl >= d) inc += t; }
  return inc - v;
}
template <class T> __device__ __forceinline__ T tall_excl_down(T v, int gl, int G) {
  T inc = v;
  for (int d = 1; d < G; d <<= 1) { const T t = __shfl_down(inc, d, 64); if (gl + d < G) inc += t; }
  return inc - v;
}

// ---------------------------------------------------------------- SimplexBijector pullbacks on tall columns
// The math of simplex_vjp_stream_kernel (bjx_seq.hip: O(K) reverse sweeps of simplex.jl:47-64, :102-120, :122-138) in the G-lane
// layout above.  in: K rows (forward map) / K-1 rows (inverse), out_bar: K-1 / K rows, in_bar like in.  Forward map: Σ_{j<k} x_j and
// the suffix sums of the adjoint are plain scans (a pullback has no reference summation order to keep).  Inverse map: the clamped
// recurrence is re-run in the reference's order by take-turns rounds and the adjoint of Σ, sb <- B_k sb + A_k, the same way from
// the last lane down.
template <class T, bool INV, bool VK, bool VK1>     // VK: the K-row runs are whole 16-byte packs; VK1: the (K-1)-row runs
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void tall_simplex_vjp_kernel(const T* __restrict__ in, const T* __restrict__ out_bar,
                                                                                                           const T* __restrict__ ladj_bar, T* __restrict__ in_bar,
                                                                                                           int K, int64_t batch, int G, int nsteps, int scan) {
  using F = Fast<T>;
  using C = TallCfg<T>;
  constexpr int V = C::V, RPL = C::RPL, SLOT = C::SLOT, NQ = C::NQ, WPB = C::WPB;
  constexpr bool VA = INV ? VK1 : VK, VG = INV ? VK : VK1;            // in / in_bar ; out_bar
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* strips = reinterpret_cast<T*>(smem);
  T* lktab = strips + WPB * 64 * SLOT;                                 // [G slots], inverse only
  const int CPS = 64 / G;
  const int rows_a = INV ? K - 1 : K, rows_g = INV ? K : K - 1;
  unsigned short* tab_a = reinterpret_cast<unsigned short*>(lktab + (INV ? G * SLOT : 0));
  unsigned short* tab_g = tab_a + (VA ? 0 : (CPS * rows_a + 1) / 2 * 2);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = lane / G, gl = lane - cg * G;
  const bool idle = cg >= CPS;
  T* st = strips + wave * 64 * SLOT;
  if (INV) {
    for (int r = threadIdx.x; r < G * RPL; r += 64 * WPB) lktab[(r / RPL) * SLOT + (r & (RPL - 1))] = r < K - 1 ? d_log(T(K - 1 - r)) : T(0);
  }
  if (!VA) tall_build_table<T>(tab_a, rows_a, G, CPS);
  if (!VG) tall_build_table<T>(tab_g, rows_g, G, CPS);
  __syncthreads();
  int pos_a[NQ], pos_g[NQ];
  if (VA) tall_pack_positions<T>(pos_a, lane, rows_a, G, CPS);
  if (VG) tall_pack_positions<T>(pos_g, lane, rows_g, G, CPS);
  const int iK = idle ? -1 : K - 1 - gl * RPL;                         // index of row K-1 inside this lane
  const int na = idle ? 0 : (rows_a - gl * RPL < 0 ? 0 : (rows_a - gl * RPL > RPL ? RPL : rows_a - gl * RPL));
  const int ng = idle ? 0 : (rows_g - gl * RPL < 0 ? 0 : (rows_g - gl * RPL > RPL ? RPL : rows_g - gl * RPL));
  const T e = Num<T>::eps;
  const T c = T(1) / (T(1) - 2 * e), E = T(1) + e, c2 = T(1) - 2 * e;
  const T* lk = lktab + gl * SLOT;
  const int64_t set0 = ((int64_t)blockIdx.x * WPB + wave) * nsteps;
  for (int sidx = 0; sidx < nsteps; ++sidx) {
    const int64_t colw = (set0 + sidx) * CPS;
    if (colw >= batch) break;
    const int ncol = (int)((batch - colw) < CPS ? (batch - colw) : CPS);
    const int64_t col = colw + cg;
    T a[RPL], g[RPL];
    __builtin_amdgcn_wave_barrier();
    tall_run_load<T, VA>(st, in + colw * rows_a, ncol, rows_a, G, CPS, lane, pos_a, tab_a);
    __builtin_amdgcn_wave_barrier();
    tall_slot_read<T>(st, lane, a, na);
    __builtin_amdgcn_wave_barrier();
    tall_run_load<T, VG>(st, out_bar + colw * rows_g, ncol, rows_g, G, CPS, lane, pos_g, tab_g);
    __builtin_amdgcn_wave_barrier();
    tall_slot_read<T>(st, lane, g, ng);
    const T lb = (ladj_bar && !idle && cg < ncol) ? ladj_bar[col] : T(0);
    if constexpr (!INV) {
      // ---- pullback of x -> (y, logabsdetjac): a = x (K rows), g = ȳ (K-1 rows)
      T tot = T(0);
#pragma unroll
      for (int i = 0; i < RPL; ++i) tot += a[i];
      T s = tall_excl_up(tot, gl, G);                                  // Σ of the rows before my first one
      T as_[RPL];
      T asum = T(0);
#pragma unroll
      for (int i = 0; i < RPL; ++i) {
        const bool row0 = i == 0 && gl == 0;
        const bool dead = i >= iK;                                     // x_K and the padding: no gradient
        const T xk = a[i];
        T dtdx, dtds;
        simplex_t_partials<T>(xk, s, false, dtdx, dtds);
        if (i == 0) { T d0, d1; simplex_t_partials<T>(xk, s, true, d0, d1); dtdx = row0 ? d0 : dtdx; dtds = row0 ? T(0) : dtds; }
        const T rd = row0 ? T(1) : F::rcp(E - s);
        const T an = row0 ? xk * c2 + e : (xk + e) * c2;               // zf = an·rd
        const T zf = an * rd;
        const T zfb = g[i] * F::rcp(zf * (T(1) - zf));
        const T ax = zfb * c2 * rd - lb * dtdx;
        const T asv = row0 ? T(0) : zfb * an * rd * rd - lb * dtds;
        a[i] = dead ? T(0) : ax;
        as_[i] = dead ? T(0) : asv;
        asum += as_[i];
        s += xk;
      }
      T sfx = tall_excl_down(asum, gl, G);                             // Σ as over the rows after my last one
#pragma unroll
      for (int i = RPL - 1; i >= 0; --i) { a[i] = i < na ? a[i] + sfx : T(0); sfx += as_[i]; }
    } else {
      // ---- pullback of y -> (x, logabsdetjac): a = y (K-1 rows), g = x̄ (K rows)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const Pack<T, V> lkq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const typename Vec16<T>::type*>(lk + q * V));
#pragma unroll
        for (int j = 0; j < V; ++j) { const int i = q * V + j; a[i] = i < iK ? f_logistic(a[i] - lkq.v[j]) * c : T(0); }   // c·z_k
      }
      T carry = T(0);
      bool use_scan = scan != 0;
      if (use_scan) {                                                  // Σ entering this lane by the affine scan of TSimplexInv (see there)
        T As = T(1), Bs = T(0);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
          const bool row0 = i == 0 && gl == 0;
          const T Ar = row0 ? T(1) : T(1) - a[i];
          const T Br = row0 ? a[i] - e * c : E * a[i] - e;
          Bs = Ar * Bs + Br;
          As = Ar * As;
        }
        tall_affine_scan_up(As, Bs, gl, G);
        const T bc = __shfl_up(Bs, 1, 64);
        carry = gl == 0 ? T(0) : bc;
        // The scan models the recurrence WITHOUT its clamp: exact (up to association) as long as no clamp changes a value.  Where one
        // does — a saturated row hands over the rest of the stick and every later row is pinned at 0 with a closed gate — the
        // unclamped Σ drifts and would open those gates again.  One clamped trip from the scan's carry finds out (the first lane a
        // clamp binds in starts from an uncontaminated carry); if any row of the wave was clamped the rounds below take over.
        bool viol = false;
        T sc = carry;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
          const bool row0 = i == 0 && gl == 0;
          const T raw = row0 ? a[i] - e * c : (E - sc) * a[i] - e;
          const T xi = d_clamp(raw, T(0), T(1));
          viol = viol || (raw != xi && i < iK);                        // (a NaN counts)
          sc += xi;
        }
        if (__builtin_amdgcn_ballot_w64(viol && !idle && cg < ncol) != 0) use_scan = false;   // wave-uniform
      }
      if (!use_scan) {
        carry = T(0);
        for (int t = 1; t < G; ++t) {
          T s = carry;
#pragma unroll
          for (int i = 0; i < RPL; ++i) {
            const bool row0 = i == 0 && gl == 0;
            const T xi = row0 ? d_clamp(a[i] - e * c, T(0), T(1)) : d_clamp((E - s) * a[i] - e, T(0), T(1));
            s += xi;
          }
          const T bc = __shfl_up(s, 1, 64);
          carry = gl == 0 ? T(0) : bc;
        }
      }
      T xk[RPL], A[RPL], B[RPL];
      T s = carry;
      T sb0 = T(0);
#pragma unroll
      for (int i = 0; i < RPL; ++i) {
        const bool row0 = i == 0 && gl == 0;
        const bool rowK = i == iK, dead = i >= iK;
        const T xi = row0 ? d_clamp(a[i] - e * c, T(0), T(1)) : d_clamp((E - s) * a[i] - e, T(0), T(1));
        T dtdx, dtds;
        simplex_t_partials<T>(xi, s, false, dtdx, dtds);
        if (i == 0) { T d0, d1; simplex_t_partials<T>(xi, s, true, d0, d1); dtdx = row0 ? d0 : dtdx; dtds = row0 ? T(0) : dtds; }
        const bool gate = xi > T(0) && xi < T(1);
        const T rc = (E - s) * c;
        const T z = row0 ? xi * c2 + e : (xi + e) * F::rcp(rc);
        const T cz = (gate && !row0) ? c * z : T(0);
        // sb_next = B sb + A ;  ȳ = gate (g + sb_in + lb dtdx) w,  w = rc z (1-z)  (row 0: c z (1-z))
        B[i] = dead ? T(1) : T(1) - cz;
        A[i] = dead ? T(0) : lb * dtds - cz * (g[i] + lb * dtdx);
        xk[i] = (gate && !dead) ? (row0 ? c : rc) * z * (T(1) - z) : T(0);   // w_k (0 where the clamp is active)
        if (rowK) { const T last = T(1) - s; sb0 = (last > T(0) && last < T(1)) ? -g[i] : T(0); }
        g[i] = dead ? T(0) : g[i] + lb * dtdx;                         // g + lb dtdx
        s += xi;
      }
      // adjoint of Σ, from the last row down: the lanes take turns from the right
      const bool lastlane = iK >= 0 && iK < RPL;
      T cin = lastlane ? sb0 : T(0);
      if (use_scan) {
        // sb <- B_k sb + A_k is affine in sb: compose this lane's rows (last row first), scan the compositions from the last lane down,
        // apply the composition of the lanes to my right to the value the last row starts from
        T P = T(1), Q = T(0);
#pragma unroll
        for (int i = RPL - 1; i >= 0; --i) { Q = B[i] * Q + A[i]; P = B[i] * P; }
        tall_affine_scan_down(P, Q, gl, G);
        const T sbK = __shfl(sb0, lane - gl + (G - 1), 64);            // the column's last lane holds row K-1
        const T Pr = __shfl_down(P, 1, 64), Qr = __shfl_down(Q, 1, 64);
        cin = lastlane ? sb0 : Pr * sbK + Qr;
      } else {
        for (int t = 1; t < G; ++t) {
          T sb = cin;
#pragma unroll
          for (int i = RPL - 1; i >= 0; --i) sb = B[i] * sb + A[i];
          const T bc = __shfl_down(sb, 1, 64);
          cin = lastlane ? sb0 : bc;
        }
      }
      T sb = cin;
#pragma unroll
      for (int i = RPL - 1; i >= 0; --i) {
        a[i] = (g[i] + sb) * xk[i];
        sb = B[i] * sb + A[i];
      }
    }
    __builtin_amdgcn_wave_barrier();
    tall_slot_write<T>(st, lane, a);
    __builtin_amdgcn_wave_barrier();
    tall_run_store<T, VA>(st, in_bar + colw * rows_a, ncol, rows_a, G, CPS, lane, pos_a, tab_a);
  }
}

template <class T>
int launch_tall_simplex_vjp(bjx_ctx* ctx, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  using C = TallCfg<T>;
  const int G = (int)((K + C::RPL - 1) / C::RPL), CPS = 64 / G;
  const int64_t sets = (batch + CPS - 1) / CPS;
  int nsteps = (int)(sets / 4096);
  nsteps = nsteps < 1 ? 1 : (nsteps > 8 ? 8 : nsteps);
  const int64_t waves = (sets + nsteps - 1) / nsteps;
  const int64_t grid = (waves + C::WPB - 1) / C::WPB;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  const int64_t rows_a = inverse ? K - 1 : K, rows_g = inverse ? K : K - 1;
  static const int vjp_scan = 1;   // 0: take-turns rounds in the pullback of the inverse
  const bool al = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
  const bool vk = al && K % C::V == 0, vk1 = al && (K - 1) % C::V == 0;
  const bool va = inverse ? vk1 : vk, vg = inverse ? vk : vk1;
  const size_t smem = ((size_t)C::WPB * 64 * C::SLOT + (inverse ? (size_t)G * C::SLOT : 0)) * sizeof(T) +
                      (va ? 0 : ((size_t)CPS * rows_a + 1) / 2 * 2 * sizeof(unsigned short)) + (vg ? 0 : ((size_t)CPS * rows_g + 1) / 2 * 2 * sizeof(unsigned short));
  {
    BjxProf prof_(ctx);
#define TSV(I_, A_, B_) hipLaunchKernelGGL((tall_simplex_vjp_kernel<T, I_, A_, B_>), dim3((unsigned)grid), dim3(64 * C::WPB), smem, ctx->stream, in, out_bar, ladj_bar, \
                                           in_bar, (int)K, batch, G, nsteps, vjp_scan)
#define TSV2(I_) do { if (vk) { if (vk1) TSV(I_, true, true); else TSV(I_, true, false); } else { if (vk1) TSV(I_, false, true); else TSV(I_, false, false); } } while (0)
    if (inverse) TSV2(true); else TSV2(false);
#undef TSV2
#undef TSV
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T, class Op>
int launch_tall(bjx_ctx* ctx, const Op& op, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t rows_in, int64_t rows_out,
                int64_t batch, uint32_t flags, int inv_scan) {
  using C = TallCfg<T>;
  const int64_t rows = rows_in > rows_out ? rows_in : rows_out;
  const int G = (int)((rows + C::RPL - 1) / C::RPL), CPS = 64 / G;
  const int64_t sets = (batch + CPS - 1) / CPS;
  // steps per wave: the log table and the strip set-up are per block; keep >= ~4096 waves in the grid when the batch allows
  int nsteps = (int)(sets / 4096);
  nsteps = nsteps < 1 ? 1 : (nsteps > 8 ? 8 : nsteps);
  const int64_t waves = (sets + nsteps - 1) / nsteps;
  const int64_t grid = (waves + C::WPB - 1) / C::WPB;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  // 16-byte accesses: a column is whole packs and the buffer starts on a 16-byte boundary
  const bool vin = bjx_aligned16(in) && rows_in % C::V == 0;
  const bool vout = out && bjx_aligned16(out) && rows_out % C::V == 0;
  const size_t smem = ((size_t)C::WPB * 64 * C::SLOT + (Op::USES_LOGK ? (size_t)G * C::SLOT : 0)) * sizeof(T) +
                     (vin ? 0 : ((size_t)CPS * rows_in + 1) / 2 * 2 * sizeof(unsigned short)) + (vout ? 0 : ((size_t)CPS * rows_out + 1) / 2 * 2 * sizeof(unsigned short));
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  {
    BjxProf prof_(ctx);
#define TSK(VI_, VO_) hipLaunchKernelGGL((tall_stream_kernel<T, Op, VI_, VO_>), dim3((unsigned)grid), dim3(64 * C::WPB), smem, ctx->stream, op, in, out, \
                                         ladj_ps, (int)rows_in, (int)rows_out, batch, G, nsteps, (flags & BJX_ACCUMULATE) ? 1 : 0, partials, inv_scan)
    if (vin) { if (vout) TSK(true, true); else TSK(true, false); }
    else { if (vout) TSK(false, true); else TSK(false, false); }
#undef TSK
  }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

template <class T>
int tall_dispatch(bjx_ctx* ctx, int which, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t rows_in, int64_t rows_out, int64_t batch,
                  uint32_t flags, int inv_scan) {
  const bool want = ladj_ps || ladj_sum;
  switch (which) {
    case BJX_TALL_ORDERED_FWD: return launch_tall<T>(ctx, TOrderedFwd<T>{}, in, out, ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan);
    case BJX_TALL_ORDERED_INV: return launch_tall<T>(ctx, TOrderedInv<T>{}, in, out, ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan);
    case BJX_TALL_SIMPLEX_FWD:
      return want ? launch_tall<T>(ctx, TSimplexFwd<T, true>{}, in, out, ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan)
                  : launch_tall<T>(ctx, TSimplexFwd<T, false>{}, in, out, ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan);
    case BJX_TALL_SIMPLEX_INV:
      return want ? launch_tall<T>(ctx, TSimplexInv<T, true>{}, in, out, ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan)
                  : launch_tall<T>(ctx, TSimplexInv<T, false>{}, in, out, ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan);
  }
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_tall_stream: bad map %d", which);
}
}  // namespace

// Columns of rows_in -> rows_out rows, contiguous.  *taken = false (and nothing launched) when the shape is not for this kernel:
// the callers in bjx_seq.hip then fall through to the walkers.
int bjx_tall_stream(bjx_ctx* ctx, bjx_dtype dt, int which, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t rows_in,
                    int64_t rows_out, int64_t batch, uint32_t flags, bool* taken) {
  *taken = false;
  static const int use_tall = getenv("BJX_SEQ_TALL") ? atoi(getenv("BJX_SEQ_TALL")) : 1;
  // the Simplex inverse pays four chain operations per row and round: beyond `inv_max` rows the chunked walker is ahead (same-box A/B)
  static const long inv_max = 512;
  static const long min_rows = 65;
  const int rpl = dt == BJX_F32 ? 32 : 16;
  const int64_t rows = rows_in > rows_out ? rows_in : rows_out;
  if (!use_tall || batch <= 0 || rows < min_rows || rows > 64 * rpl) return BJX_OK;
  // Carry of the Simplex inverse: take-turns rounds in the reference's order (default) or the affine scan (BJX_SEQ_TALL_INV_SCAN=1, an
  // experiment).  The scan is NOT parity-safe for the transform: on a long simplex the stick is used up before the last rows (1 - Σ
  // falls to ε), the log-det terms divide by what is left, and a re-associated Σ moves the per-column log-det by up to 0.7 % there
  // (Float32, K = 1000, y ~ N(0, 1.5²): 7 % of the columns beyond the 1e-3 bar); and it ignores the clamp — after a saturated row
  // (z_k = 1: the rest of the stick at once) the reference pins Σ at 1 and every later term at log ε, the unclamped Σ drifts back by ε
  // per row and the terms with it (tests/test_gpu_parity.py::test_simplex_inverse_tall_columns_with_clamped_rows, either dtype).
  static const int scan_env = -1;
  const int inv_scan = scan_env > 0 ? 1 : 0;
  // with rounds the chunked walker is ahead beyond 512 rows (four chain operations per row and round)
  if (which == BJX_TALL_SIMPLEX_INV && (rows > (inv_scan ? 2048 : inv_max) || rows < 129)) return BJX_OK;   // 65-128 rows: the whole-column tile is ahead (45 against 43 % at K = 100)
  // Float64 (same-box A/B, profiles/r03_tall_columns.md): the rounds of the Simplex inverse are four Float64 operations per row — the
  // walkers stay ahead at every height (39 / 35 % against 31 / 27 % at K = 200 / 500), so the Float64 inverse is not taken here; beyond
  // 32 lanes per column (512 rows) the walkers are level or ahead for the other maps too (Ordered at K = 1000: 54 against 45 %)
  if (dt == BJX_F64 && (rows > 512 || (which == BJX_TALL_SIMPLEX_INV && !inv_scan))) return BJX_OK;
  const int G = (int)((rows + rpl - 1) / rpl), CPS = 64 / G;
  // lanes that hold rows of a column / lanes of the wave: K just above a multiple of RPL wastes most of the last lane
  const double eff = (double)rows * CPS / (64.0 * rpl);
  static const double min_eff = 0.6;
  if (eff < min_eff) return BJX_OK;
  *taken = true;
  if (dt == BJX_F32) return tall_dispatch<float>(ctx, which, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan);
  return tall_dispatch<double>(ctx, which, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, rows_in, rows_out, batch, flags, inv_scan);
}

// Pullback of the Simplex maps on tall columns; same contract as bjx_tall_stream
int bjx_tall_simplex_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K,
                         int64_t batch, bool* taken) {
  *taken = false;
  static const int use_tall = getenv("BJX_SIMPLEX_VJP_TALL") ? atoi(getenv("BJX_SIMPLEX_VJP_TALL")) : 1;
  static const long min_rows = 65;
  static const long inv_max = 2048;
  const int rpl = dt == BJX_F32 ? 32 : 16;
  if (!use_tall || batch <= 0 || K < min_rows || K > 64 * rpl) return BJX_OK;
  if (inverse && K > inv_max) return BJX_OK;
  const int G = (int)((K + rpl - 1) / rpl), CPS = 64 / G;
  static const double min_eff = 0.6;
  if ((double)K * CPS / (64.0 * rpl) < min_eff) return BJX_OK;
  *taken = true;
  if (dt == BJX_F32) return launch_tall_simplex_vjp<float>(ctx, inverse, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
  return launch_tall_simplex_vjp<double>(ctx, inverse, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
}
