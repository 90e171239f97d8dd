// bjx_elem.hip — per-element bijectors with a per-sample log-det on the column-group skeleton:
//   RationalQuadraticSpline (F4), InvertibleBatchNorm eval (F2 row a18), Coupling (F5), Permute (F5).
#include "bjx_stream.h"

namespace {
using namespace bjx;

// ------------------------------------------------------------------ RQS scalar maps
// Knot tables are [rows, K1] column-major: knot k (1-based) of row r is tab[(k-1)*rows + r].
// Base.searchsortedfirst(v, x): first 1-based index with v[i] >= x, else len+1.
template <class T> __device__ __forceinline__ int ssf(const T* v, int64_t st, int len, T x) {
  int lo = 0, hi = len + 1;
  while (lo < hi - 1) {
    int m = lo + ((hi - lo) >> 1);
    if (v[(int64_t)(m - 1) * st] < x) lo = m; else hi = m;
  }
  return hi;
}

// rational_quadratic_spline.jl:317-357 (rqs_forward: value + logjac share s, xi, denominator)
template <class T>
__device__ __forceinline__ void rqs_forward_dev(const T* w_, const T* h_, const T* d_, int64_t st, int K, T x, T& y, T& lj) {
  const T wK = w_[(int64_t)(K - 1) * st];
  if ((x <= -wK) || (x >= wK)) { y = x; lj = T(0) * x; return; }   // :322-324 (`zero(T) * x`: NaN for ±Inf)
  int k = ssf<T>(w_, st, K, x) - 1;
  T w_k = (k == 0) ? -wK : w_[(int64_t)(k - 1) * st];
  T w = w_[(int64_t)k * st] - w_k;
  T h_k = (k == 0) ? -h_[(int64_t)(K - 1) * st] : h_[(int64_t)(k - 1) * st];
  T dy = h_[(int64_t)k * st] - h_k;
  T s = dy / w;
  T xi = (x - w_k) / w;
  T d_k = (k == 0) ? T(1) : d_[(int64_t)(k - 1) * st];
  T d_k1 = (k == K - 1) ? T(1) : d_[(int64_t)k * st];
  T om = T(1) - xi;
  T den = s + (d_k1 + d_k - 2 * s) * xi * om;
  T num_jl = s * s * (d_k1 * (xi * xi) + 2 * s * xi * om + d_k * (om * om));
  lj = d_log(num_jl) - 2 * d_log(den);
  T num_y = dy * (s * (xi * xi) + d_k * xi * om);
  y = h_k + num_y / den;
}
// rational_quadratic_spline.jl:183-220
template <class T>
__device__ __forceinline__ T rqs_inverse_dev(const T* w_, const T* h_, const T* d_, int64_t st, int K, T y) {
  const T hK = h_[(int64_t)(K - 1) * st];
  if ((y <= -hK) || (y >= hK)) return y;
  int k = ssf<T>(h_, st, K, y) - 1;
  T w_k = (k == 0) ? -w_[(int64_t)(K - 1) * st] : w_[(int64_t)(k - 1) * st];
  T w = w_[(int64_t)k * st] - w_k;
  T h_k = (k == 0) ? -hK : h_[(int64_t)(k - 1) * st];
  T dy = h_[(int64_t)k * st] - h_k;
  T s = dy / w;
  T d_k = (k == 0) ? T(1) : d_[(int64_t)(k - 1) * st];
  T d_k1 = (k == K - 1) ? T(1) : d_[(int64_t)k * st];
  T ds = d_k1 + d_k - 2 * s;
  T a1 = dy * (s - d_k) + (y - h_k) * ds;
  T a2 = dy * d_k - (y - h_k) * ds;
  T a3 = -s * (y - h_k);
  T num = -2 * a3;
  T den = a2 + d_sqrt(a2 * a2 - 4 * a1 * a3);
  T xi = num / den;
  return xi * w + w_k;
}
template <class T, bool INV>
__device__ __forceinline__ T rqs_elem(const T* w, const T* h, const T* d, int64_t st, int K, T& v) {
  T y, lj;
  if (!INV) { rqs_forward_dev<T>(w, h, d, st, K, v, y, lj); v = y; return lj; }
  T x = rqs_inverse_dev<T>(w, h, d, st, K, v);
  rqs_forward_dev<T>(w, h, d, st, K, x, y, lj);   // interface.jl:276-281: -logabsdetjac(orig, x)
  v = x;
  return -lj;
}

// stage three [rows,K1] tables into LDS
template <class T> __device__ __forceinline__ void stage3(T* dst, const T* w, const T* h, const T* d, int64_t n) {
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { dst[i] = w[i]; dst[n + i] = h[i]; dst[2 * n + i] = d[i]; }
}

template <class T, bool INV> struct RqsF {
  static constexpr bool kLoadInput = true;
  const T *w, *h, *d;
  int K1;
  int64_t rows;
  int in_lds;
  double per_sample_const;
  const double* per_sample_dev;
  int walk_smem_offset = 0;   // colwalk_kernel: where the column tile starts behind the functor's LDS tables (set by launch_colgroup)
  __device__ void stage(char* smem) const {
    if (in_lds) { stage3<T>(reinterpret_cast<T*>(smem), w, h, d, rows * K1); __syncthreads(); }
  }
  template <int V> __device__ T apply(const char* smem, Pack<T, V>& p, const T*, int64_t row, int64_t) const {
    T l = T(0);
    if (in_lds) {
      const T* W = reinterpret_cast<const T*>(smem);
      const T* H = W + rows * K1;
      const T* D = H + rows * K1;
#pragma unroll
      for (int j = 0; j < V; ++j) l += rqs_elem<T, INV>(W + row + j, H + row + j, D + row + j, rows, K1, p.v[j]);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) l += rqs_elem<T, INV>(w + row + j, h + row + j, d + row + j, rows, K1, p.v[j]);
    }
    return l;
  }
};

// rational_quadratic_spline.jl:109-123: B-constructor (softmax -> cumsum -> affine; log1pexp derivatives)
template <class T>
__global__ __launch_bounds__(256) void rqs_params_kernel(const T* rw, const T* rh, const T* rd, int K, int64_t dim, T B,
                                                         T* w, T* h, T* d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (int64_t)gridDim.x * blockDim.x) {
    for (int pass = 0; pass < 2; ++pass) {
      const T* r = pass == 0 ? rw : rh;
      T* o = pass == 0 ? w : h;
      T mx = r[i];
      for (int k = 1; k < K; ++k) mx = d_max(mx, r[(int64_t)k * dim + i]);
      T s = T(0);
      for (int k = 0; k < K; ++k) s += d_exp(r[(int64_t)k * dim + i] - mx);
      T c = T(0);
      o[i] = (2 * B) * c - B;
      for (int k = 0; k < K; ++k) { c += d_exp(r[(int64_t)k * dim + i] - mx) / s; o[(int64_t)(k + 1) * dim + i] = (2 * B) * c - B; }
    }
    d[i] = T(1);
    for (int k = 0; k < K - 1; ++k) d[(int64_t)(k + 1) * dim + i] = d_log1pexp(rd[(int64_t)k * dim + i]);
    d[(int64_t)K * dim + i] = T(1);
  }
}

// ------------------------------------------------------------------ RQS, LDS table kernel
// The functor path above costs 178 (fwd) / 295 (inv) VALU instructions per element and suffers
// 8-way LDS bank conflicts on the [rows, K] column-major knot tables (PMC: SQ_LDS_BANK_CONFLICT =
// 63 % of LDS cycles), i.e. it is VALU/LDS-bound at 13 % of the HBM roofline.  This kernel keeps the
// reference's arithmetic (rational_quadratic_spline.jl:128-164,183-220,266-357) but
//  * precomputes, once per call, a per-(row, bin) record {w_k, 1/w, h_k, Δy | s, d_k, d_k+1, Σd-2s}
//    (two 16-byte LDS reads replace six 4-byte reads and two divisions per element);
//  * stores the searched knots as an implicit binary tree, LEVEL BY LEVEL, with the rows permuted so
//    that the rows touched by one wave instruction are adjacent: level l of the 8 rows a wave
//    instruction touches at dim = 32 is 8·2^(l-1) consecutive words -> levels 1-3 are bank-conflict
//    free, level 4 at most 2-way (the old [row][bin] layout put every row on the same 16 banks);
//  * keeps each lane's level-1/2 keys and range limit in registers (a lane owns the same rows of
//    every column), so only levels >= 3 are read from LDS per element;
//  * is branch-free per element (outside [-B, B] is a select), searches the 4 elements of a pack
//    in lock step, and uses one hardware log + one reciprocal per element in Float32;
//  * reuses the inverse's ξ for its log-det instead of a second search + forward evaluation;
//  * drops the empty bin 0 (knot 1 = -knot K for the `B` constructor, :109-123) when every row has
//    it, which saves one search level for K+1 = 2^m + 1 knots (flag computed on the device);
//  * lets one block walk ITER column groups so the table staging is amortised.
// Bin selection is exact (same knot values and comparisons as the oracle).
//
// LDS blob (built by rqs_blob_kernel in the context scratch, copied verbatim), key part in units of T:
//   [0, dimp)                         lim[rp]          = knot K (range limit) of permuted row rp
//   [dimp·2^(l-1), dimp·2^l)          level l keys     [rp][2^(l-1)],  l = 1..NSTEP
// then the bin records in 256-byte LDS rows of sixteen 16-byte slots.  A lane always reads slot (lane & 15): the
// ds_read_b128 lane groups hold one lane of every residue mod 16, so sixteen different records — whatever bins the
// sixteen lanes landed in — come from sixteen different slots of the bank row, without conflicts.  (With the records
// of one table row contiguous the sixteen random bins of a lane group collided ~3-way: SQ_LDS_BANK_CONFLICT was
// 62 % of the LDS cycles and the LDS pipe as busy as the VALU.)  Row index:
//   ((j·GH + hi)·NS + pos)·RQ + q     j = element of the pack, hi = lane-in-group / 16, pos = bin slot,
//                                     q = quad of the record: Float32 A, B; Float64 A.lo, A.hi, B.lo, B.hi with
//                                     A = {-w_k/w, 1/w, h_k, Δy}  (inverse: {h_k, Δy, w_k, w}),
//                                     B = {s, d_k, d_k+1 + d_k - 2s, d_k+1 - d_k}
// and slot s of a row holds the record of the group lane gl = hi·16 + s (groups of 16+ lanes) or gl = s mod G
// (narrower groups: 16/G copies).  rp = j·nvc + gl for row gl·V + j (V = pack width, nvc = packs per column).
struct RqsGeom {
  int K1;       // knots per row
  int nvc;      // packs per column (dim / V)
  int V;
  int dimp;     // rows, padded to a multiple of 4
  int nstep;    // search levels
  int kbase;    // 1: bin 0 dropped
  int nslots;   // bins kept = K1 - kbase
  int G;        // lanes per column group (power of two >= nvc)
  int GH;       // 16-lane slices of a group: max(1, G / 16)
};
__host__ __device__ inline RqsGeom rqs_geom(int K1, int64_t dim, int V, int skip0, int nstep_hi, int G) {
  RqsGeom g;
  g.K1 = K1; g.V = V; g.nvc = (int)(dim / V); g.dimp = (int)((dim + 3) / 4 * 4);
  g.kbase = skip0 ? 1 : 0;
  g.nstep = nstep_hi - g.kbase;
  g.nslots = K1 - g.kbase;
  g.G = G; g.GH = G > 16 ? G / 16 : 1;
  return g;
}
template <class T> struct RqsRec { static constexpr int RQ = sizeof(T) == 4 ? 2 : 4; static constexpr int PS = sizeof(T) == 4 ? 9 : 10; };   // quads per record, log2(bytes per bin)
template <class T> __host__ __device__ inline size_t rqs_key_bytes(const RqsGeom& g) { return (size_t)g.dimp * (1u << g.nstep) * sizeof(T); }
template <class T> __host__ __device__ inline size_t rqs_blob_bytes(const RqsGeom& g) {
  return rqs_key_bytes<T>(g) + (size_t)g.V * g.GH * g.nslots * RqsRec<T>::RQ * 256;
}
// byte offset (from the blob start) of the record of (pack element j, lane) at bin 0; bin pos adds pos << PS, quad q adds 256 q
template <class T> __device__ __forceinline__ int rqs_rec_base(const RqsGeom& g, int j, int lane, int gl) {
  const int hi = g.G > 16 ? gl >> 4 : 0;
  return (int)rqs_key_bytes<T>(g) + (((j * g.GH + hi) * g.nslots) << RqsRec<T>::PS) + (lane & 15) * 16;
}

// One block: (i) flag[0] = 1 iff knot 1 <= -knot K for widths and heights of every row (bin 0
// unreachable, only evaluated when `dual`), (ii) the LDS blob in the layout above.
template <class T, bool INV>
__global__ __launch_bounds__(256) void rqs_blob_kernel(const T* w, const T* h, const T* d, int K1, int64_t rows, int V, int nstep_hi,
                                                       int dual, int G, int* flag, T* blob, int64_t tstride = 0) {
  // `rows` rows of the knot tables starting at w / h / d; `tstride` = rows of the WHOLE tables when this is a row slab of a
  // taller spline (knot j of row r is w[(j-1)*tstride + r]); 0 = the tables have exactly `rows` rows
  const int64_t ts = tstride ? tstride : rows;
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  if (dual) {
    for (int64_t r = threadIdx.x; r < rows; r += blockDim.x)
      if (!(w[r] <= -w[(int64_t)(K1 - 1) * ts + r]) || !(h[r] <= -h[(int64_t)(K1 - 1) * ts + r])) bad = 1;
  }
  __syncthreads();
  const int skip0 = (dual && !bad) ? 1 : 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) flag[0] = skip0;
  const RqsGeom g = rqs_geom(K1, rows, V, skip0, nstep_hi, G);
  const int nkeys = (1 << g.nstep) - 1;
  const int per_row = nkeys > 1 ? nkeys : 1;
  const int64_t total = (int64_t)g.dimp * per_row;
  // (round 5: the items are dealt over the blocks of the launch — one block took 7.4 us, four dependent global reads per thread)
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, tstep = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = t0; i < total; i += tstep) {
    const int rp = (int)(i / per_row), s = (int)(i % per_row);
    const int64_t r = (int64_t)(rp % g.nvc) * g.V + rp / g.nvc;   // actual row
    const bool live = rp < g.V * g.nvc && r < rows;
    auto W = [&](int j) { return w[(int64_t)(j - 1) * ts + r]; };   // 1-based knot j of row r
    auto H = [&](int j) { return h[(int64_t)(j - 1) * ts + r]; };
    if (s == 0) blob[rp] = live ? (INV ? H(K1) : W(K1)) : T(0);
    if (s < nkeys) {
      // sorted searched key s (0-based) = knot kbase + s + 1, padded with +inf; tree position:
      const int t = __builtin_ctz(s + 1);
      const int lvl = g.nstep - t, path = (s + 1) >> (t + 1);
      T kv = Num<T>::inf;
      if (live && s < g.nslots - 1) kv = INV ? H(g.kbase + s + 1) : W(g.kbase + s + 1);
      blob[(size_t)g.dimp * (1u << (lvl - 1)) + (size_t)rp * (1u << (lvl - 1)) + path] = kv;
    }
  }
  // records: one per (pack element j, 16-lane slice hi, bin slot, slot of the LDS row)
  char* rec = reinterpret_cast<char*>(blob) + rqs_key_bytes<T>(g);
  constexpr int RQ = RqsRec<T>::RQ;
  const int64_t nrec = (int64_t)g.V * g.GH * g.nslots * 16;
  for (int64_t i = t0; i < nrec; i += tstep) {
    const int slot = (int)(i & 15);
    const int s = (int)((i >> 4) % g.nslots);
    const int jh = (int)((i >> 4) / g.nslots);
    const int j = jh / g.GH, hi = jh % g.GH;
    const int gl = g.G > 16 ? hi * 16 + slot : (slot & (g.G - 1));
    const int64_t r = (int64_t)gl * g.V + j;                          // actual row
    const bool live = gl < g.nvc && r < rows;
    auto W = [&](int q) { return w[(int64_t)(q - 1) * ts + r]; };   // 1-based knot q of row r
    auto H = [&](int q) { return h[(int64_t)(q - 1) * ts + r]; };
    auto D = [&](int q) { return d[(int64_t)(q - 1) * ts + r]; };
    T a[4] = {T(0), T(1), T(0), T(0)}, b[4] = {T(1), T(1), T(0), T(0)};
    if (live) {
      const int k = s + g.kbase;                                   // bin k spans knots k..k+1 (knot 0 = -knot K)
      const T w_k = (k == 0) ? -W(K1) : W(k);                      // :140,:192
      const T wd = W(k + 1) - w_k;
      const T h_k = (k == 0) ? -H(K1) : H(k);
      const T dy = H(k + 1) - h_k;
      const T sl = dy / wd;                                        // s = Δy/w
      const T d_k = (k == 0) ? T(1) : D(k);
      const T d_k1 = (k == K1 - 1) ? T(1) : D(k + 1);
      if (!INV) { a[1] = T(1) / wd; a[0] = -w_k * a[1]; a[2] = h_k; a[3] = dy; }   // ξ = x·(1/w) − w_k/w: one FMA per element
      else { a[0] = h_k; a[1] = dy; a[2] = w_k; a[3] = wd; }
      b[0] = sl; b[1] = d_k; b[2] = d_k1 + d_k - 2 * sl; b[3] = d_k1 - d_k;
    }
    char* row0 = rec + ((((size_t)(j * g.GH + hi) * g.nslots + s) * RQ) << 8) + slot * 16;
    constexpr int PER = 16 / (int)sizeof(T);                           // values per 16-byte quad
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      reinterpret_cast<T*>(row0 + ((q / PER) << 8))[q % PER] = a[q];
      reinterpret_cast<T*>(row0 + ((RQ / 2 + q / PER) << 8))[q % PER] = b[q];
    }
  }
}

template <class T> struct Rec4 { T v[4]; };
// record half (A: q0 = 0, B: q0 = RQ/2) from the slotted rows: quads 256 bytes apart
template <class T> __device__ __forceinline__ Rec4<T> lds_rec(const char* p, int q0) {
  Rec4<T> r;
  if constexpr (sizeof(T) == 4) {
    bjx_f32x4 t = *reinterpret_cast<const bjx_f32x4*>(p + (q0 << 8));
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    bjx_f64x2 t0 = *reinterpret_cast<const bjx_f64x2*>(p + (q0 << 8)), t1 = *reinterpret_cast<const bjx_f64x2*>(p + ((q0 + 1) << 8));
    r.v[0] = t0.x; r.v[1] = t0.y; r.v[2] = t1.x; r.v[3] = t1.y;
  }
  return r;
}

// value + log2-det of one element given its bin records; x is replaced by the result.
// Forward: rational_quadratic_spline.jl:317-357.  Inverse: :183-220 + the forward log-det at the
// recovered ξ, negated (interface.jl:276-281).  With p = ξ(1-ξ) (one fma: ξ - ξ²), ξ² = ξ - p and
// (1-ξ)² = (1-ξ) - p, the reference's three quadratics are LINEAR in (ξ, p):
//   denominator      s + (d_{k+1} + d_k - 2s) ξ(1-ξ)                       = s + ds·p
//   numerator of y   s ξ² + d_k ξ(1-ξ)                                     = ξ (d_k + (s - d_k) ξ)
//   numerator of J   d_{k+1} ξ² + 2s ξ(1-ξ) + d_k (1-ξ)²                   = d_k + (d_{k+1} - d_k) ξ - ds·p
// (the last is a convex interpolation minus a term of at most the same size: cancellation <= ~2x).
// Returns log2|J| (the caller multiplies the per-column sum by ln 2 once).
template <class T, bool INV>
__device__ __forceinline__ T rqs_eval(const Rec4<T>& A, const Rec4<T>& B, T lim, T& x) {
  using F = Fast<T>;
  const T s = B.v[0], d_k = B.v[1], ds = B.v[2], dd = B.v[3];
  const T xin = x;
  T xi, res;
  if (!INV) {
    xi = xin * A.v[1] + A.v[0];                                             // ξ = (x - w_k)/w as one FMA (A.v[0] = -w_k/w)
  } else {
    const T yh = xin - A.v[0];
    const T t = yh * ds;
    const T a1 = A.v[1] * (s - d_k) + t;                                    // Eq. (25)
    const T a2 = A.v[1] * d_k - t;                                          // Eq. (26)
    const T q = s * yh;                                                     // -a3, Eq. (27)
    xi = F::div(q + q, a2 + F::sqrt(a2 * a2 + 4 * (a1 * q)));               // Eq. (24)
  }
  const T p = xi - xi * xi;                                                 // contracts to fma(-ξ, ξ, ξ)
  T den, tq;
  if constexpr (sizeof(T) == 4) {
    // (den, t) = (s, d_k) + (ds, dd)·(p, ξ): both operand pairs are adjacent halves of the 16-byte record read -> one v_pk_fma_f32
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 r2 = __builtin_elementwise_fma(f2{ds, dd}, f2{p, xi}, f2{s, d_k});
    den = r2.x; tq = r2.y;
  } else { den = s + ds * p; tq = d_k + dd * xi; }
  const T rden = F::rcp(den);
  const T nj = tq - ds * p;
  const T sr = s * rden;
  T lj = F::log2(nj * (sr * sr));                                           // log(s²·nj) - 2 log(den)
  if (!INV) res = A.v[2] + (A.v[3] * (xi * (d_k + (s - d_k) * xi))) * rden;
  else { res = xi * A.v[3] + A.v[2]; lj = -lj; }
  // identity outside [-B, B] (:132, :186): (x <= -lim || x >= lim) == (|x| >= lim); a NaN is NOT outside (both
  // comparisons are false in the reference too) and leaves through the arithmetic as NaN value and NaN log-det
  const bool outside = d_abs(xin) >= lim;
  x = outside ? xin : res;
  return outside ? T(0) * xin : lj;                                          // `zero(T) * x` (:272, :323): NaN for x = ±Inf, like the reference
}

// Float64 (round 6): the FACTORS of |J| instead of its logarithm.  The log-det of a column is a sum over its rows, so the caller
// multiplies the factors of the V elements of a pack and takes ONE lean log (≈ 40 Float64 VALU operations, scripts/f64math_bench.hip)
// per pack instead of one per element; the inverse does not need 1/den per element either (one reciprocal of the product of the
// dens per pack).  In Float32 the same change was measured and reverted (profiles/r06_c3_experiments.md: a hardware log is 4 issue
// slots, the range guard of the product costs what it saves).
//   forward:  jn = nj·(s/den)²;   inverse:  jn = nj·s², jd = den  ->  log2|J⁻¹| of the pack = −log2(Π jn · (1/Π jd)²)
// Outside [-B, B]: jn = 1 + 0·x (1, or NaN for x = ±Inf like the reference's `zero(T) * x`, :272, :323), jd = 1.
template <class T, bool INV>
__device__ __forceinline__ void rqs_eval_factors(const Rec4<T>& A, const Rec4<T>& B, T lim, T& x, T& jn, T& jd) {
  using F = Fast<T>;
  const T s = B.v[0], d_k = B.v[1], ds = B.v[2], dd = B.v[3];
  const T xin = x;
  T xi, res;
  if (!INV) {
    xi = xin * A.v[1] + A.v[0];
  } else {
    const T yh = xin - A.v[0];
    const T t = yh * ds;
    const T a1 = A.v[1] * (s - d_k) + t;                                    // Eq. (25)
    const T a2 = A.v[1] * d_k - t;                                          // Eq. (26)
    const T q = s * yh;                                                     // -a3, Eq. (27)
    xi = F::div(q + q, a2 + F::sqrt(a2 * a2 + 4 * (a1 * q)));               // Eq. (24)
  }
  const T p = xi - xi * xi;
  const T den = s + ds * p, tq = d_k + dd * xi;
  const T nj = tq - ds * p;
  T jn_in;
  if (!INV) {
    const T rden = F::rcp(den);
    const T sr = s * rden;
    jn_in = nj * (sr * sr);
    res = A.v[2] + (A.v[3] * (xi * (d_k + (s - d_k) * xi))) * rden;
  } else {
    jn_in = nj * (s * s);
    res = xi * A.v[3] + A.v[2];
  }
  const bool outside = d_abs(xin) >= lim;
  x = outside ? xin : res;
  jn = outside ? __builtin_fma(T(0), xin, T(1)) : jn_in;
  jd = (INV && !outside) ? den : T(1);
}
// log2 of the pack's Jacobian from the factors.  The product of V factors can leave the range where no single factor does (or reach 0 /
// a negative value / NaN where one factor would decide): then the logs are taken one by one — the rare branch.
template <class T, int V, bool INV> __device__ __forceinline__ T rqs_log2_of_factors(const T (&jn)[V], const T (&jd)[V]) {
  using F = Fast<T>;
  constexpr T lo = T(1e-140), hi = T(1e140);
  T pn = jn[0], pd = jd[0];
#pragma unroll
  for (int j = 1; j < V; ++j) { pn *= jn[j]; pd *= jd[j]; }
  const bool ok = pn > lo && pn < hi && (!INV || (pd > lo && pd < hi));
  if (__builtin_expect(ok, 1)) {
    if (!INV) return F::log2(pn);
    const T r = F::rcp(pd);
    return -F::log2(pn * (r * r));
  }
  T l = T(0);
#pragma unroll
  for (int j = 0; j < V; ++j) l += INV ? T(2) * F::log2(jd[j]) - F::log2(jn[j]) : F::log2(jn[j]);
  return l;
}

// pos = 2*pos + (key < x): one compare + one add-with-carry (the compiler's own lowering of this
// line is compare + cndmask + shift-or).
__device__ __forceinline__ void search_step(int& pos, float key, float x) {
  asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(pos) : "v"(key), "v"(x) : "vcc");
}
__device__ __forceinline__ void search_step(int& pos, double key, double x) {
  asm("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(pos) : "v"(key), "v"(x) : "vcc");
}

template <class T, int V, int NSTEP, bool INV>
__device__ __forceinline__ void rqs_body(const T* __restrict__ blob_l, const RqsGeom g, const T* __restrict__ x, T* __restrict__ y,
                                         T* __restrict__ ladj_ps, int64_t dim, int64_t batch, int G, int iters, int accumulate,
                                         double& acc, int64_t ld) {
  // `ld`: elements between the starts of consecutive columns (== dim for dense arrays; the full column height when x / y point
  // at a row slab of taller columns)
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = 256 / G;
  const bool lane_ok = gl < g.nvc;
  const int glc = lane_ok ? gl : 0;
  const char* base = reinterpret_cast<const char*>(blob_l);
  // per-lane constants: a lane owns rows glc*V + j of every column.  lb[l][j] / ra / rb are LDS BYTE
  // offsets so the per-element address is one v_lshl_add_u32 of the search position.
  T lim[V], k1[V], k2a[V], k2b[V];
  int lb[NSTEP > 2 ? NSTEP - 2 : 1][V], ra[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int rp = j * g.nvc + glc;
    lim[j] = lane_ok ? blob_l[rp] : T(0);          // an idle lane of the group (zeros from its out-of-range loads) is "outside": log-det 0
    k1[j] = blob_l[g.dimp + rp];
    if (NSTEP >= 2) { k2a[j] = blob_l[2 * g.dimp + 2 * rp]; k2b[j] = blob_l[2 * g.dimp + 2 * rp + 1]; }
    else { k2a[j] = k2b[j] = T(0); }
#pragma unroll
    for (int lvl = 3; lvl <= NSTEP; ++lvl) lb[lvl - 3][j] = (int)sizeof(T) * ((g.dimp + rp) << (lvl - 1));
    ra[j] = rqs_rec_base<T>(g, j, (int)threadIdx.x, glc);
    // opaque to the optimiser: otherwise the constants are re-read from LDS inside the column loop
    asm volatile("" : "+v"(lim[j]), "+v"(k1[j]), "+v"(k2a[j]), "+v"(k2b[j]));
  }
  constexpr int SH = sizeof(T) == 4 ? 2 : 3;
  // The column loop is STRAIGHT-LINE code: every global access goes through a buffer descriptor whose extent is the
  // block's remaining columns, so a lane past the end (or an idle lane of the group) reads zeros and its stores are
  // dropped by the range check instead of sitting behind an exec branch.  With branches around the loads and stores
  // the compiler could not count the memory operations in flight and drained ALL of them (s_waitcnt vmcnt(0), twice
  // per trip) — every wave waited for the acknowledgement of the stores it had just issued before it could touch the
  // columns that had been prefetched a trip earlier (PMC: 54 % of the wave cycles in s_waitcnt).  Now the wait is
  // "all but the newest k operations".
  const int64_t bcol0 = (int64_t)blockIdx.x * iters * cols_per_block;
  const int cg = threadIdx.x / G;
  const int64_t left_blk = batch - bcol0;
  const int ncols_blk = left_blk > (int64_t)iters * cols_per_block ? iters * cols_per_block : (left_blk > 0 ? (int)left_blk : 0);
  const int col_bytes = (int)ld * (int)sizeof(T);                      // a block spans < 2^31 bytes (<= 64 trips of <= 256 columns)
  const int trip_cols = cols_per_block;
  constexpr int kOob = 0x7fffff00;                                     // beyond any extent: loads give 0, stores are dropped
  const int vo0 = lane_ok ? (cg * (int)ld + gl * V) * (int)sizeof(T) : kOob;
  const int vo1 = lane_ok ? vo0 + trip_cols * col_bytes : kOob;
  const int lo0 = gl == 0 ? cg * (int)sizeof(T) : kOob;                // the group's first lane owns the column's log-det
  const int lo1 = gl == 0 ? lo0 + trip_cols * (int)sizeof(T) : kOob;
  const int my_cols = gl == 0 ? ncols_blk - cg : 0;                    // columns whose log-det this lane adds to the block partial
  const char* xb = reinterpret_cast<const char*>(x + bcol0 * ld);
  char* yb = reinterpret_cast<char*>(y + bcol0 * ld);
  char* lpb = reinterpret_cast<char*>(ladj_ps ? ladj_ps + bcol0 : nullptr);
  // descriptors of trip `it` (two column groups from block column it*trip_cols on)
  auto extent = [&](int it, int unit) -> uint32_t { const int r = ncols_blk - it * trip_cols; return r > 0 ? (uint32_t)r * (uint32_t)unit : 0u; };
  auto rsrc_x = [&](int it) { return bjx_make_rsrc(xb + (int64_t)it * trip_cols * col_bytes, extent(it, col_bytes)); };
  auto rsrc_y = [&](int it) { return bjx_make_rsrc(yb + (int64_t)it * trip_cols * col_bytes, extent(it, col_bytes)); };
  auto rsrc_l = [&](int it) { return bjx_make_rsrc(lpb + (int64_t)it * trip_cols * (int)sizeof(T), lpb ? extent(it, (int)sizeof(T)) : 0u); };
  auto rsrc_l_in = [&](int it) { return bjx_make_rsrc(lpb + (int64_t)it * trip_cols * (int)sizeof(T), (lpb && accumulate) ? extent(it, (int)sizeof(T)) : 0u); };
  // TWO columns per trip (the per-column epilogue — G-lane butterfly behind wave-uniform branches, log-det
  // store, loop control — costs ~40 VALU, a quarter of a 4-element pack's evaluation) and one trip of
  // look-ahead: the next two loads are in flight while these are evaluated.
  auto eval_pack = [&](Pack<T, V>& p) -> T {
    int pos[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      pos[j] = (k1[j] < p.v[j]) ? 1 : 0;
      if (NSTEP >= 2) { const T kk = pos[j] ? k2b[j] : k2a[j]; search_step(pos[j], kk, p.v[j]); }
    }
#pragma unroll
    for (int lvl = 3; lvl <= NSTEP; ++lvl) {
      T kv[V];
#pragma unroll
      for (int j = 0; j < V; ++j) kv[j] = *reinterpret_cast<const T*>(base + ((pos[j] << SH) + lb[lvl - 3][j]));
#pragma unroll
      for (int j = 0; j < V; ++j) search_step(pos[j], kv[j], p.v[j]);
    }
    Rec4<T> A[V], B[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const char* rec = base + ((pos[j] << RqsRec<T>::PS) + ra[j]);
      A[j] = lds_rec<T>(rec, 0);
      B[j] = lds_rec<T>(rec, RqsRec<T>::RQ / 2);
    }
    if constexpr (sizeof(T) == 8 && V > 1) {
      T jn[V], jd[V];
#pragma unroll
      for (int j = 0; j < V; ++j) rqs_eval_factors<T, INV>(A[j], B[j], lim[j], p.v[j], jn[j], jd[j]);
      return rqs_log2_of_factors<T, V, INV>(jn, jd);
    } else {
      T l = T(0);
#pragma unroll
      for (int j = 0; j < V; ++j) l += rqs_eval<T, INV>(A[j], B[j], lim[j], p.v[j]);
      return l;
    }
  };
  const GroupMasks gm = make_group_masks(G);
  Pack<T, V> pn0, pn1;
  {
    const auto r0 = rsrc_x(0);
    pn0 = buf_load_pack<T, V>(r0, vo0);
    pn1 = buf_load_pack<T, V>(r0, vo1);
    // the first columns land before the loop: otherwise the loop head inherits "wait until at most 4 operations are
    // outstanding" from this entry edge, which on every later trip means waiting for the previous trip's stores
#pragma unroll
    for (int j = 0; j < V; ++j) asm volatile("" : "+v"(pn0.v[j]), "+v"(pn1.v[j]));   // a use: the wait sits here, outside the loop
  }
  for (int it = 0; it < iters; it += 2) {
    Pack<T, V> p0 = pn0, p1 = pn1;
    // BJX_ACCUMULATE: read-modify-write of the log-det.  The reads are issued FIRST in the trip (the counter of
    // outstanding memory operations is in-order: waiting for the newest operation waits for everything before it) and,
    // without the flag, go through an EMPTY descriptor (zeros, no memory access) rather than around a branch — a
    // conditional load inside the loop makes the compiler drain every outstanding operation at the loop head.
    const auto rl_in = rsrc_l_in(it);
    const T lin0 = buf_load_pack<T, 1>(rl_in, lo0).v[0], lin1 = buf_load_pack<T, 1>(rl_in, lo1).v[0];
    {
      const auto rn = rsrc_x(it + 2);
      pn0 = buf_load_pack<T, V>(rn, vo0);
      pn1 = buf_load_pack<T, V>(rn, vo1);
    }
    const auto ry = rsrc_y(it);
    // the fences keep the loads at the head of the trip (a full trip of look-ahead) and the two evaluations apart
    // (interleaved by the scheduler they need 130 VGPRs instead of ~100)
    __builtin_amdgcn_sched_barrier(0);
    T l0 = eval_pack(p0);
    buf_store_pack<T, V>(ry, vo0, p0);
    __builtin_amdgcn_sched_barrier(0);
    T l1 = eval_pack(p1);
    buf_store_pack<T, V>(ry, vo1, p1);
    __builtin_amdgcn_sched_barrier(0);
    group_sum2_flat(l0, l1, gm);
    l0 *= Num<T>::log2;                               // log2 -> natural log, once per column
    l1 *= Num<T>::log2;
    const auto rl = rsrc_l(it);
    const T s0 = l0 + lin0, s1 = l1 + lin1;
    buf_store_pack<T, 1>(rl, lo0, Pack<T, 1>{{s0}});
    buf_store_pack<T, 1>(rl, lo1, Pack<T, 1>{{s1}});
    acc += (double)(it * trip_cols < my_cols ? l0 : T(0));
    acc += (double)((it + 1) * trip_cols < my_cols ? l1 : T(0));
  }
}

template <class T, int V, int NSTEP_HI, bool DUAL, bool INV>
__global__ __launch_bounds__(256) void rqs_lds_kernel(const T* __restrict__ blob, const int* __restrict__ flag, int K1,
                                                      const T* __restrict__ x, T* __restrict__ y, T* __restrict__ ladj_ps, int64_t dim,
                                                      int64_t batch, int G, int iters, int accumulate, const BjxFin fin, int64_t ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // no static LDS (the reduction scratch aliases the table once every wave is done with it).  The launch asks for the NO-SKIP layout
  // (C3: 38 912 bytes — whether bin 0 can be dropped is decided on the device, after the launch was sized): FOUR blocks per CU.  Five
  // (97 -> 96 VGPRs and a 32 KiB request) were built and measured in round 6: no gain (profiles/r06_c3_experiments.md).
  T* blob_l = reinterpret_cast<T*>(smem);
  const int skip0 = DUAL ? flag[0] : 0;
  const RqsGeom g = rqs_geom(K1, dim, V, skip0, NSTEP_HI, G);
  {
    const int n16 = (int)(rqs_blob_bytes<T>(g) / 16);
    const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(blob);
    bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(blob_l);
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    __syncthreads();
  }
  double acc = 0.0;
  if constexpr (DUAL) {
    if (skip0) rqs_body<T, V, NSTEP_HI - 1, INV>(blob_l, g, x, y, ladj_ps, dim, batch, G, iters, accumulate, acc, ld);
    else rqs_body<T, V, NSTEP_HI, INV>(blob_l, g, x, y, ladj_ps, dim, batch, G, iters, accumulate, acc, ld);
  } else {
    rqs_body<T, V, NSTEP_HI, INV>(blob_l, g, x, y, ladj_ps, dim, batch, G, iters, accumulate, acc, ld);
  }
  __syncthreads();                                                  // the table is dead: its first bytes become the scratch
  block_publish_partial_at(acc, reinterpret_cast<double*>(smem), reinterpret_cast<int*>(smem + 64), fin);
}

// ------------------------------------------------------------------ RQS input pullback
// x̄ = ȳ·f'(x) + ℓ̄·(log f')'(x) for the elementwise spline (rational_quadratic_spline.jl:128-357; closed-form
// derivatives).  With the quantities rqs_eval already has (ξ, p = ξ(1-ξ), den = s + ds·p, nj = d_k + dd·ξ - ds·p):
//   f' = s²·nj/den²,   d log f'/dx = [ (dd - ds(1-2ξ))/nj - 2 ds(1-2ξ)/den ] / w
// Inverse map (x = f⁻¹(y), log-det -log f'(x)):  ȳ = (x̄ - ℓ̄·d log f'/dx)/f'.  Outside [-B, B]: identity.
// Same LDS blob, search and records as rqs_lds_kernel; the search depth is a runtime loop here (one instantiation per
// direction), one column per trip.
template <class T, bool INV>
__device__ __forceinline__ T rqs_eval_vjp(const Rec4<T>& A, const Rec4<T>& B, T lim, T xin, T g, T lb) {
  using F = Fast<T>;
  const T s = B.v[0], d_k = B.v[1], ds = B.v[2], dd = B.v[3];
  T xi, iw;
  if (!INV) { xi = xin * A.v[1] + A.v[0]; iw = A.v[1]; }
  else {
    const T yh = xin - A.v[0];
    const T t = yh * ds;
    const T a1 = A.v[1] * (s - d_k) + t;
    const T a2 = A.v[1] * d_k - t;
    const T q = s * yh;
    xi = F::div(q + q, a2 + F::sqrt(a2 * a2 + 4 * (a1 * q)));
    iw = F::rcp(A.v[3]);
  }
  const T p = xi - xi * xi;
  const T den = s + ds * p;
  const T rden = F::rcp(den);
  const T nj = (d_k + dd * xi) - ds * p;
  const T sr = s * rden;
  const T J = nj * (sr * sr);                                               // f'
  const T om = T(1) - (xi + xi);                                            // dp/dξ
  const T dl = ((dd - ds * om) * F::rcp(nj) - T(2) * ds * om * rden) * iw;  // d log f'/dx
  const T out = !INV ? g * J + lb * dl : (g - lb * dl) * F::rcp(J);
  return d_abs(xin) >= lim ? g : out;                                        // NaN: not outside, NaN cotangent
}

template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void rqs_vjp_kernel(const T* __restrict__ blob, const int* __restrict__ flag, int K1, int nstep_hi, int dual,
                                                      const T* __restrict__ x, const T* __restrict__ gbar, const T* __restrict__ lbar, T* __restrict__ xbar,
                                                      int64_t dim, int64_t batch, int G, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* blob_l = reinterpret_cast<T*>(smem);
  const int skip0 = dual ? flag[0] : 0;
  const RqsGeom g = rqs_geom(K1, dim, V, skip0, nstep_hi, G);
  {
    const int n16 = (int)(rqs_blob_bytes<T>(g) / 16);
    const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(blob);
    bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(blob_l);
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    __syncthreads();
  }
  const int NS = g.nstep;
  const int gl = threadIdx.x & (G - 1), cg = threadIdx.x / G;
  const int cols_per_block = 256 / G;
  const bool lane_ok = gl < g.nvc;
  const int glc = lane_ok ? gl : 0;
  const char* base = reinterpret_cast<const char*>(blob_l);
  T lim[V];
  int rp[V], ra[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    rp[j] = j * g.nvc + glc;
    lim[j] = blob_l[rp[j]];
    ra[j] = rqs_rec_base<T>(g, j, (int)threadIdx.x, glc);
  }
  // Straight-line column loop on buffer descriptors with one trip of look-ahead (cf. rqs_body): the next column's x, ȳ and ℓ̄ are in
  // flight while this one is searched and evaluated; lanes past the end read zeros and drop their stores.
  const int64_t bcol0 = (int64_t)blockIdx.x * iters * cols_per_block;
  const int64_t left_blk = batch - bcol0;
  const int ncols_blk = left_blk > (int64_t)iters * cols_per_block ? iters * cols_per_block : (left_blk > 0 ? (int)left_blk : 0);
  const int col_bytes = (int)dim * (int)sizeof(T);
  constexpr int kOob = 0x7fffff00;
  const int vo = lane_ok ? (cg * (int)dim + gl * V) * (int)sizeof(T) : kOob;
  const int lo = cg * (int)sizeof(T);
  const char* xb = reinterpret_cast<const char*>(x + bcol0 * dim);
  const char* gb = reinterpret_cast<const char*>(gbar + bcol0 * dim);
  char* ob = reinterpret_cast<char*>(xbar + bcol0 * dim);
  const char* lbp = reinterpret_cast<const char*>(lbar ? lbar + bcol0 : nullptr);
  auto extent = [&](int it, int unit) -> uint32_t { const int r = ncols_blk - it * cols_per_block; return r > 0 ? (uint32_t)r * (uint32_t)unit : 0u; };
  auto rs = [&](const char* b, int it, int unit, bool on) { return bjx_make_rsrc(b + (int64_t)it * cols_per_block * unit, on ? extent(it, unit) : 0u); };
  Pack<T, V> pn = buf_load_pack<T, V>(rs(xb, 0, col_bytes, true), vo);
  Pack<T, V> gn = buf_load_pack<T, V>(rs(gb, 0, col_bytes, true), vo);
  T ln = buf_load_pack<T, 1>(rs(lbp, 0, (int)sizeof(T), lbp != nullptr), lo).v[0];
  for (int it = 0; it < iters; ++it) {
    Pack<T, V> p = pn;
    const Pack<T, V> gp = gn;
    const T lb = ln;
    pn = buf_load_pack<T, V>(rs(xb, it + 1, col_bytes, true), vo);
    gn = buf_load_pack<T, V>(rs(gb, it + 1, col_bytes, true), vo);
    ln = buf_load_pack<T, 1>(rs(lbp, it + 1, (int)sizeof(T), lbp != nullptr), lo).v[0];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int pos = 0;
      for (int lvl = 1; lvl <= NS; ++lvl) {                                  // level l keys: [dimp·2^(l-1) + rp·2^(l-1) + pos]
        const T key = blob_l[((size_t)(g.dimp + rp[j]) << (lvl - 1)) + pos];
        pos = 2 * pos + (key < p.v[j] ? 1 : 0);
      }
      const char* rec = base + ((pos << RqsRec<T>::PS) + ra[j]);
      const Rec4<T> A = lds_rec<T>(rec, 0);
      const Rec4<T> B = lds_rec<T>(rec, RqsRec<T>::RQ / 2);
      p.v[j] = rqs_eval_vjp<T, INV>(A, B, lim[j], p.v[j], gp.v[j], lb);
    }
    buf_store_pack<T, V>(rs(ob, it, col_bytes, true), vo, p);
  }
}

// ------------------------------------------------------------------ BatchNorm (eval)
// normalise.jl:41-88.  LDS rows: s = exp(logs), m, q = sqrt(v + eps), b.
template <class T, bool INV> struct BnF {
  static constexpr bool kLoadInput = true;
  static constexpr bool kMulti = true;            // the row parameters are fetched once for the columns a lane has in flight
  static constexpr bool kMasked = true;           // any first row; the per-element log-det is a per-column constant (no mask needed)
  template <int V, int U> __device__ void apply_multi_masked(const char* smem, Pack<T, V> (&p)[U], int64_t row, T (&l)[U], uint32_t) const { apply_multi<V, U>(smem, p, row, l); }
  template <int V> __device__ T apply_masked(const char* smem, Pack<T, V>& p, const T* xc, int64_t row, int64_t col, uint32_t) const { return apply<V>(smem, p, xc, row, col); }
  template <int V, int U> __device__ void apply_multi(const char* smem, Pack<T, V> (&p)[U], int64_t row, T (&l)[U]) const {
    const T* t = reinterpret_cast<const T*>(smem);
#pragma unroll
    for (int i = 0; i < U; ++i) l[i] = T(0);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int64_t r = row + j;
      T s, mm, q, bb;
      if (in_lds) { s = t[r]; mm = t[dim + r]; q = t[2 * dim + r]; bb = t[3 * dim + r]; }
      else { s = d_exp(logs[r]); mm = m[r]; q = d_sqrt(v[r] + eps); bb = b[r]; }
#pragma unroll
      for (int i = 0; i < U; ++i) {
        if (!INV) p[i].v[j] = s * (p[i].v[j] - mm) / q + bb;        // :62
        else p[i].v[j] = (p[i].v[j] - bb) / s * q + mm;              // :83
      }
    }
  }
  const T *b, *logs, *m, *v;
  T eps;
  int64_t dim;
  int in_lds;
  double per_sample_const;
  const double* per_sample_dev;
  int walk_smem_offset = 0;   // colwalk_kernel: where the column tile starts behind the functor's LDS tables (set by launch_colgroup)
  __device__ void stage(char* smem) const {
    if (in_lds) {
      T* t = reinterpret_cast<T*>(smem);
      for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) {
        t[i] = d_exp(logs[i]); t[dim + i] = m[i]; t[2 * dim + i] = d_sqrt(v[i] + eps); t[3 * dim + i] = b[i];
      }
      __syncthreads();
    }
  }
  template <int V> __device__ T apply(const char* smem, Pack<T, V>& p, const T*, int64_t row, int64_t) const {
    const T* t = reinterpret_cast<const T*>(smem);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int64_t r = row + j;
      T s, mm, q, bb;
      if (in_lds) { s = t[r]; mm = t[dim + r]; q = t[2 * dim + r]; bb = t[3 * dim + r]; }
      else { s = d_exp(logs[r]); mm = m[r]; q = d_sqrt(v[r] + eps); bb = b[r]; }
      if (!INV) p.v[j] = s * (p.v[j] - mm) / q + bb;        // :62
      else p.v[j] = (p.v[j] - bb) / s * q + mm;              // :83
    }
    return T(0);
  }
};
// consts[0] = ± Σ_c (logs_c - log(v_c+eps)/2) (:63); consts[1] = batch * consts[0]
template <class T>
__global__ __launch_bounds__(256) void bn_const_kernel(const T* logs, const T* v, T eps, int64_t dim, int64_t batch, int inv, double* consts) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) s += (double)(logs[i] - d_log(v[i] + eps) / T(2));
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double c = (red[0] + red[1]) + (red[2] + red[3]);
    if (inv) c = -c;
    consts[0] = c;
    consts[1] = c * (double)batch;
  }
}

// ------------------------------------------------------------------ Coupling
// coupling.jl:125-134,206-259.  rowmap[r] = position of row r in idx1 (the transformed partition
// x_1) or -1 (rows of x_2 / x_3 copy through: combine() adds A_2 x_2 + A_3 x_3 unchanged).
__global__ void rowmap_kernel(const int32_t* idx1, int64_t n1, int64_t dim, int32_t* map, int* bad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) {
    int32_t r = idx1[i];
    if (r < 0 || r >= dim) *bad = 1; else map[r] = (int32_t)i;
  }
}

template <class T, bool INV> struct CouplingAffineF {
  static constexpr bool kLoadInput = true;
  static constexpr bool kMasked = true;           // fetch() takes any first row; apply_masked drops the log-det terms of rows outside the mask
  const int32_t* map;
  const T *scale, *shift;   // [n1, batch] or null
  int64_t n1, dim;
  int64_t ss, st;           // column strides of scale / shift: n1, or 0 when the array is T[n1] shared by every column
  int map_in_lds;           // the row map is staged in LDS (dim <= 12 Ki rows), else read from the context scratch
  double per_sample_const;
  const double* per_sample_dev;
  int walk_smem_offset = 0;   // colwalk_kernel: where the column tile starts behind the functor's LDS tables (set by launch_colgroup)
  // θ of one pack: fetched with the input packs (all of a lane's loads in flight together).  When the pack's
  // rows sit at consecutive, 16-byte aligned positions of x_1 (PartitionMask over a row range — the usual
  // mask) scale and shift are one 16-byte load each; otherwise V scalar gathers.
  template <int V> struct AuxV { Pack<T, V> s, t; int32_t m0; uint32_t on; };
  using Aux = AuxV<Vec16<T>::N>;
  __device__ void stage(char* smem) const {
    if (!map_in_lds) return;
    int32_t* m = reinterpret_cast<int32_t*>(smem);
    for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) m[i] = map[i];
    __syncthreads();
  }
  template <int V> __device__ Aux fetch(const char* smem, int64_t row, int64_t col) const {
    if (map_in_lds) return fetch_from<V>(reinterpret_cast<const int32_t*>(smem), row, col);   // two calls: each keeps its address space
    return fetch_from<V>(map, row, col);
  }
  template <int V> __device__ __forceinline__ Aux fetch_from(const int32_t* m, int64_t row, int64_t col) const {
    static_assert(V <= Vec16<T>::N, "pack wider than Aux");
    Aux a;
    a.on = 0;
    a.m0 = m[row];
    bool run = a.m0 >= 0 && V > 1 && (a.m0 % V) == 0 && (n1 % V) == 0 && bjx_aligned16_dev(scale) && bjx_aligned16_dev(shift);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int32_t mi = m[row + j];
      if (mi >= 0) a.on |= 1u << j;
      run = run && mi == a.m0 + j;
      a.s.v[j] = T(1);
      a.t.v[j] = T(0);
    }
    if (run) {
      Pack<T, V> s1, t1;
      if (scale) s1 = load_pack<T, V, true>(scale + col * ss + a.m0);
      if (shift) t1 = load_pack<T, V, true>(shift + col * st + a.m0);
#pragma unroll
      for (int j = 0; j < V; ++j) { if (scale) a.s.v[j] = s1.v[j]; if (shift) a.t.v[j] = t1.v[j]; }
    } else if (a.on) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int32_t mi = m[row + j];
        if (mi >= 0) {
          if (scale) a.s.v[j] = scale[col * ss + mi];
          if (shift) a.t.v[j] = shift[col * st + mi];
        }
      }
    }
    return a;
  }
  template <int V> __device__ T apply(const char* sm, Pack<T, V>& p, const Aux& a, const T* xc, int64_t row, int64_t col) const {
    return apply_masked<V>(sm, p, a, xc, row, col, ~0u);
  }
  template <int V> __device__ T apply_masked(const char*, Pack<T, V>& p, const Aux& a, const T*, int64_t, int64_t, uint32_t mask) const {
    using F = Fast<T>;
    T l = T(0);
    if (a.on) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const bool on = (a.on >> j) & (mask >> j) & 1u;
        const T s = a.s.v[j], t = a.t.v[j];
        // Shift(t) ∘ Scale(s) (coupling.jl:206-219); inverse(Scale) ∘ inverse(Shift) (:236-250)
        const T v = !INV ? t + s * p.v[j] : F::rcp(s) * (-t + p.v[j]);
        if (on) { p.v[j] = v; l += F::log2(d_abs(s)); }
      }
      l *= INV ? -Num<T>::log2 : Num<T>::log2;
    }
    return l;
  }
};

template <class T, bool INV> struct CouplingRqsF {
  static constexpr bool kLoadInput = true;
  const int32_t* map;
  const T *w, *h, *d;   // [n1, K1]
  int K1;
  int64_t n1;
  int in_lds;
  double per_sample_const;
  const double* per_sample_dev;
  int walk_smem_offset = 0;   // colwalk_kernel: where the column tile starts behind the functor's LDS tables (set by launch_colgroup)
  __device__ void stage(char* smem) const {
    if (in_lds) { stage3<T>(reinterpret_cast<T*>(smem), w, h, d, n1 * K1); __syncthreads(); }
  }
  template <int V> __device__ T apply(const char* smem, Pack<T, V>& p, const T*, int64_t row, int64_t) const {
    T l = T(0);
    const T* W = in_lds ? reinterpret_cast<const T*>(smem) : w;
    const T* H = in_lds ? W + n1 * K1 : h;
    const T* D = in_lds ? H + n1 * K1 : d;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int32_t mi = map[row + j];
      if (mi >= 0) l += rqs_elem<T, INV>(W + mi, H + mi, D + mi, n1, K1, p.v[j]);
    }
    return l;
  }
};

// ------------------------------------------------------------------ Permute
// permute.jl:152: out = A * in for a permutation matrix A; src[i] = column of the 1 in row i.
template <class T> struct PermuteF {
  static constexpr bool kLoadInput = false;
  const int32_t* src;
  double per_sample_const;
  const double* per_sample_dev;
  int walk_smem_offset = 0;   // colwalk_kernel: where the column tile starts behind the functor's LDS tables (set by launch_colgroup)
  __device__ void stage(char*) const {}
  template <int V> __device__ T apply(const char*, Pack<T, V>& p, const T* xcol, int64_t row, int64_t) const {
#pragma unroll
    for (int j = 0; j < V; ++j) p.v[j] = xcol[src[row + j]];
    return T(0);
  }
};

template <class T> bool knots_fit_lds(int64_t rows, int K1) { return (size_t)rows * K1 * 3 * sizeof(T) <= 60 * 1024; }

// largest knot blob the LDS kernels take (the default dynamic-LDS limit of a launch); beyond it the generic functor path runs
constexpr size_t kRqsBlobMax = 64 * 1024;
constexpr int kRqsBlobGrid = 8;           // blocks of the table-building helper launch (every block recomputes the skip flag for itself)
// Column groups per block: enough to amortise the table staging (>= ~3x the table bytes of data; same-call sweeps at 32 x 2^22,
// forward + inverse: 8 groups 0.53 ms, 12: 0.50, 17: 0.49, 24: 0.48, 34: 0.49, 64: 0.48, 128: 0.50).
// Round 3, tried and dropped (profiles/r03_c3_experiments.md, same-box A/Bs): SHORT-LIVED blocks — NP = 2 / 4 column groups per block,
// all loads up front, table staged per block — 0.39 / 0.48 of the HBM peak against 0.56 for this looping form (the 34 KiB table per
// 8-16 KiB of data is not free); and FIVE blocks per CU instead of four (no static LDS, the reduction scratch aliases the dead
// table: 5 x 32 KiB = 160 KiB) — neutral, kept because it costs nothing.
inline int rqs_iters(const bjx_ctx* ctx, size_t blob_bytes, int64_t bytes_per_group, int64_t groups) {
  static const int forced = 0;   // (round 6 swept 16 ... 64 trips per block through a temporary switch: all within the run-to-run spread, profiles/r06_c3_experiments.md)
  if (forced > 0) return forced;
  int64_t amort = (3 * (int64_t)blob_bytes + bytes_per_group - 1) / bytes_per_group;
  if (amort < 1) amort = 1;
  if (amort > 64) amort = 64;
  return (int)amort;
}
inline int ceil_log2(int n) { int s = 0; while ((1 << s) < n) ++s; return s; }

template <class T, int V, bool INV>
int rqs_launch_lds(bjx_ctx* ctx, int nstep_hi, int dual, const T* blob, const int* flag, int K1, size_t smem, int64_t grid, const T* in,
                   T* out, T* ladj_ps, int64_t dim, int64_t batch, int G, int iters, int accum, const BjxFin& fin, int64_t ld) {
  BjxProf prof_(ctx);
#define RQS_L(NS_, DUAL_) hipLaunchKernelGGL((rqs_lds_kernel<T, V, NS_, DUAL_, INV>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, blob, flag, K1, in, out, ladj_ps, dim, batch, G, iters, accum, fin, ld)
  switch (nstep_hi * 2 + (dual ? 1 : 0)) {
    case 2: RQS_L(1, false); break;
    case 4: RQS_L(2, false); break;  case 5: RQS_L(2, true); break;
    case 6: RQS_L(3, false); break;  case 7: RQS_L(3, true); break;
    case 8: RQS_L(4, false); break;  case 9: RQS_L(4, true); break;
    case 10: RQS_L(5, false); break;
    case 11: RQS_L(5, true); break;
    case 12: RQS_L(6, false); break; case 13: RQS_L(6, true); break;
    default: return bjx_fail(ctx, BJX_ERR_UNSUPPORTED, "bjx_rqs: unsupported search depth %d", nstep_hi);
  }
#undef RQS_L
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

// One launch pair (blob + rqs_lds_kernel) for `rows` rows of columns that are `ld` elements apart; the knot tables have `trows`
// rows in all (w / h / d point at the slab's first row).  -> BJX_ERR_UNSUPPORTED-free: returns 1 when the shape does not fit
// the LDS kernel (the caller then takes another path).
template <class T>
int rqs_lds_slab(bjx_ctx* ctx, int inverse, const T* w, const T* h, const T* d, int K1, int64_t trows, const T* in, T* out, T* ladj_ps,
                 double* ladj_sum, int64_t rows, int64_t ld, int64_t batch, uint32_t flags, bool* taken) {
  *taken = false;
  constexpr int VW = Vec16<T>::N;
  ColLaunch c = col_launch_cfg<T>(ctx, in, out, rows, batch, ld, ld);
  const int nstep_hi = ceil_log2(K1 < 2 ? 2 : K1);
  const int dual = (K1 >= 3 && ceil_log2(K1 - 1) < nstep_hi) ? 1 : 0;
  const RqsGeom g_hi = rqs_geom(K1, rows, c.V, 0, nstep_hi, c.G);
  const size_t blob_bytes = rqs_blob_bytes<T>(g_hi);      // the no-skip layout is the larger one
  const bool lds_path = nstep_hi <= 6 && rows / c.V <= 64 && ld < (1 << 20) && blob_bytes <= kRqsBlobMax && blob_bytes + 64 <= BJX_SCRATCH_BYTES;
  if (!lds_path) return BJX_OK;
  *taken = true;
  int* flag = reinterpret_cast<int*>(ctx->scratch);
  T* blob = reinterpret_cast<T*>(static_cast<char*>(ctx->scratch) + 64);
  // BJX_OPT_PARAM_EPOCH != 0: the blob of an unchanged spline is kept (4 slots: forward and inverse tables of two splines) and the
  // helper launch is skipped — per C3 step two launches of 7.4 us next to two hot kernels of ~220 us.
  bool build = true;
  if (ctx->param_epoch != 0 && !ctx->capturing) {          // (a captured step bakes pointers into the graph: it keeps the per-call build in the scratch)
    bjx_ctx::RqsBlobSlot* hit = nullptr;
    for (auto& sl : ctx->rqs_slots)
      if (sl.buf && sl.epoch == ctx->param_epoch && sl.w == w && sl.h == h && sl.d == d && sl.K1 == K1 && sl.rows == rows && sl.trows == trows && sl.V == c.V &&
          sl.nstep_hi == nstep_hi && sl.dual == dual && sl.G == c.G && sl.inverse == inverse && sl.dt == (int)sizeof(T)) { hit = &sl; break; }
    if (hit) build = false;
    else {
      hit = &ctx->rqs_slots[ctx->rqs_next];
      if (!hit->buf) { if (hipMalloc(&hit->buf, kRqsBlobMax + 64) != hipSuccess) hit->buf = nullptr; }
      if (hit->buf) {
        ctx->rqs_next = (ctx->rqs_next + 1) % 4;
        *hit = bjx_ctx::RqsBlobSlot{w, h, d, K1, c.V, nstep_hi, dual, c.G, inverse, (int)sizeof(T), ctx->param_epoch, rows, trows, hit->buf};
      } else hit = nullptr;                                   // (no buffer: the shared scratch, rebuilt every call)
    }
    if (hit) { flag = reinterpret_cast<int*>(hit->buf); blob = reinterpret_cast<T*>(static_cast<char*>(hit->buf) + 64); }
  }
  if (build) {
    if (inverse) hipLaunchKernelGGL((rqs_blob_kernel<T, true>), dim3(kRqsBlobGrid), dim3(256), 0, ctx->stream, w, h, d, K1, rows, c.V, nstep_hi, dual, c.G, flag, blob, trows);
    else hipLaunchKernelGGL((rqs_blob_kernel<T, false>), dim3(kRqsBlobGrid), dim3(256), 0, ctx->stream, w, h, d, K1, rows, c.V, nstep_hi, dual, c.G, flag, blob, trows);
    BJX_CHECK_LAUNCH(ctx);
  }
  const int cols_per_block = 256 / c.G;
  const int64_t groups = (batch + cols_per_block - 1) / cols_per_block;
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  const int iters = rqs_iters(ctx, blob_bytes, (int64_t)cols_per_block * rows * sizeof(T), groups);
  const int64_t grid = (groups + iters - 1) / iters;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_rqs: batch too large for one launch");
  BjxFin fin;
  bool second = false;
  { int rc = bjx_make_fin(ctx, grid, ladj_sum, 0.0, 0, flags, &fin, &second); if (rc) return rc; }
  int rc;
  if (c.V == VW) rc = inverse ? rqs_launch_lds<T, VW, true>(ctx, nstep_hi, dual, blob, flag, K1, blob_bytes, grid, in, out, ladj_ps, rows, batch, c.G, iters, accum, fin, ld)
                              : rqs_launch_lds<T, VW, false>(ctx, nstep_hi, dual, blob, flag, K1, blob_bytes, grid, in, out, ladj_ps, rows, batch, c.G, iters, accum, fin, ld);
  else rc = inverse ? rqs_launch_lds<T, 1, true>(ctx, nstep_hi, dual, blob, flag, K1, blob_bytes, grid, in, out, ladj_ps, rows, batch, c.G, iters, accum, fin, ld)
                    : rqs_launch_lds<T, 1, false>(ctx, nstep_hi, dual, blob, flag, K1, blob_bytes, grid, in, out, ladj_ps, rows, batch, c.G, iters, accum, fin, ld);
  if (rc) return rc;
  if (second) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

template <class T>
int rqs_impl(bjx_ctx* ctx, int inverse, const T* w, const T* h, const T* d, int K1, const T* in, T* out, T* ladj_ps,
             double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (dim * batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  // (Short odd columns, dim <= 7: the generic functor on the lane-per-column kernel of bjx_stream.h was tried in round 3 — 29-34 %
  //  against 29-37 % for the LDS-blob kernel with one-element packs: the functor's search, not the access width, is the cost there.)
  {
    bool taken = false;
    const int rc = rqs_lds_slab<T>(ctx, inverse, w, h, d, K1, dim, in, out, ladj_ps, ladj_sum, dim, dim, batch, flags, &taken);
    if (rc || taken) return rc;
  }
  // Taller columns than one table takes (the record part of the blob grows with the lanes per column: 65 KiB at 200 rows x 9
  // knots): ROW SLABS of 128 rows (32 lanes per column; same-box A/B at 200 / 500 / 1000 rows, K = 8, % of the HBM peak: slabs of 64
  // rows 34 / 34 / 40, 96 rows 39 / 39 / 45, 128 rows 45 / 44 / 49 — fewer launches, 512-byte runs), one launch pair per slab on a row window of the same
  // arrays (column stride = dim), the log-dets of the slabs accumulated in launch order (BJX_ACCUMULATE from the second slab on:
  // deterministic).  Round 2 sent these shapes to the generic functor kernel: 23 % of the HBM peak at dim = 200, 8 % at 1000.
  constexpr int VWs = Vec16<T>::N;
  static const int slab_rows = getenv("BJX_RQS_SLAB") ? atoi(getenv("BJX_RQS_SLAB")) : 128;     // tuning switch: 0 = the generic path as in round 2
  // (heights that are not whole aligned packs take the same slabs on 4-byte accesses, 64 lanes per column: the generic functor path
  //  ran them at 8-18 % of the HBM peak, its knots read from L2)
  const bool whole_packs = dim % VWs == 0 && bjx_aligned16(in) && bjx_aligned16(out);
  const int64_t slab = (whole_packs || slab_rows < 64) ? slab_rows : 64;       // one-element packs: 64 lanes per column take 64 rows
  if (slab > 0 && out) {
    bool ok = true;
    for (int64_t r0 = 0; r0 < dim && ok; r0 += slab) {
      const int64_t rs = dim - r0 < slab ? dim - r0 : slab;
      bool taken = false;
      const uint32_t fl = r0 == 0 ? flags : (flags | BJX_ACCUMULATE);
      const int rc = rqs_lds_slab<T>(ctx, inverse, w + r0, h + r0, d + r0, K1, dim, in + r0, out + r0, ladj_ps, ladj_sum, rs, dim, batch, fl, &taken);
      if (rc) return rc;
      if (!taken) { ok = false; BJX_REQUIRE(ctx, r0 == 0, BJX_ERR_UNSUPPORTED, "bjx_rqs: row slab %lld does not fit the LDS kernel", (long long)r0); }
    }
    if (ok) return BJX_OK;
  }
  // huge knot tables / odd very wide columns: generic functor path
  const bool lds = knots_fit_lds<T>(dim, K1);
  const size_t fsm = lds ? (size_t)dim * K1 * 3 * sizeof(T) : 0;
  if (!inverse) { RqsF<T, false> f{w, h, d, K1, dim, lds ? 1 : 0, 0.0, nullptr}; return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0); }
  RqsF<T, true> f{w, h, d, K1, dim, lds ? 1 : 0, 0.0, nullptr};
  return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
}

// Fallback of the spline pullback for tables the LDS kernel does not take (very wide columns, > 64 knots): one
// element per thread, knots read from global memory with the reference's search (searchsortedfirst, rqs.jl:137) and
// the same closed-form derivatives as rqs_eval_vjp, in plain divisions.
template <class T, bool INV>
__global__ __launch_bounds__(256) void rqs_vjp_generic_kernel(const T* __restrict__ w, const T* __restrict__ h, const T* __restrict__ d, int K,
                                                              const T* __restrict__ x, const T* __restrict__ gbar, const T* __restrict__ lbar,
                                                              T* __restrict__ xbar, int64_t dim, int64_t batch) {
  const int64_t total = dim * batch;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx % dim, col = idx / dim, st = dim;
    const T *w_ = w + row, *h_ = h + row, *d_ = d + row;
    const T xin = x[idx], g = gbar[idx], lb = lbar ? lbar[col] : T(0);
    const T wK = w_[(int64_t)(K - 1) * st], hK = h_[(int64_t)(K - 1) * st];
    const T lim = INV ? hK : wK;
    if (d_abs(xin) >= lim) { xbar[idx] = g; continue; }
    const int k = ssf<T>(INV ? h_ : w_, st, K, xin) - 1;
    const T w_k = (k == 0) ? -wK : w_[(int64_t)(k - 1) * st];
    const T wd = w_[(int64_t)k * st] - w_k;
    const T h_k = (k == 0) ? -hK : h_[(int64_t)(k - 1) * st];
    const T dy = h_[(int64_t)k * st] - h_k;
    const T s = dy / wd;
    const T d_k = (k == 0) ? T(1) : d_[(int64_t)(k - 1) * st];
    const T d_k1 = (k == K - 1) ? T(1) : d_[(int64_t)k * st];
    const T ds = d_k1 + d_k - 2 * s, dd = d_k1 - d_k;
    T xi;
    if (!INV) xi = (xin - w_k) / wd;
    else {
      const T yh = xin - h_k;
      const T a1 = dy * (s - d_k) + yh * ds;
      const T a2 = dy * d_k - yh * ds;
      const T q = s * yh;
      xi = (q + q) / (a2 + d_sqrt(a2 * a2 + 4 * (a1 * q)));
    }
    const T p = xi - xi * xi;
    const T den = s + ds * p;
    const T nj = (d_k + dd * xi) - ds * p;
    const T sr = s / den;
    const T J = nj * (sr * sr);
    const T om = T(1) - (xi + xi);
    const T dl = ((dd - ds * om) / nj - T(2) * ds * om / den) / wd;
    xbar[idx] = !INV ? g * J + lb * dl : (g - lb * dl) / J;
  }
}

template <class T>
int rqs_vjp_impl(bjx_ctx* ctx, int inverse, const T* w, const T* h, const T* d, int K1, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar,
                 int64_t dim, int64_t batch) {
  if (dim * batch == 0) return BJX_OK;
  ColLaunch c = col_launch_cfg<T>(ctx, in, in_bar, dim, batch);
  if (c.V > 1 && !bjx_aligned16(out_bar)) { c.V = 1; int G = 1; while (G < 64 && G < dim) G <<= 1; c.G = G; }
  const int nstep_hi = ceil_log2(K1 < 2 ? 2 : K1);
  const int dual = (K1 >= 3 && ceil_log2(K1 - 1) < nstep_hi) ? 1 : 0;
  const RqsGeom g_hi = rqs_geom(K1, dim, c.V, 0, nstep_hi, c.G);
  const size_t blob_bytes = rqs_blob_bytes<T>(g_hi);
  if (!(nstep_hi <= 6 && dim / c.V <= 64 && dim < (1 << 20) && blob_bytes <= kRqsBlobMax && blob_bytes + 64 <= BJX_SCRATCH_BYTES)) {
    int64_t nb = (dim * batch + 255) / 256;
    if (nb > 256 * 64) nb = 256 * 64;
    BjxProf prof_(ctx);
    if (inverse) hipLaunchKernelGGL((rqs_vjp_generic_kernel<T, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, w, h, d, K1, in, out_bar, ladj_bar, in_bar, dim, batch);
    else hipLaunchKernelGGL((rqs_vjp_generic_kernel<T, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, w, h, d, K1, in, out_bar, ladj_bar, in_bar, dim, batch);
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  }
  int* flag = reinterpret_cast<int*>(ctx->scratch);
  T* blob = reinterpret_cast<T*>(static_cast<char*>(ctx->scratch) + 64);
  if (inverse) hipLaunchKernelGGL((rqs_blob_kernel<T, true>), dim3(kRqsBlobGrid), dim3(256), 0, ctx->stream, w, h, d, K1, dim, c.V, nstep_hi, dual, c.G, flag, blob);
  else hipLaunchKernelGGL((rqs_blob_kernel<T, false>), dim3(kRqsBlobGrid), dim3(256), 0, ctx->stream, w, h, d, K1, dim, c.V, nstep_hi, dual, c.G, flag, blob);
  BJX_CHECK_LAUNCH(ctx);
  const int cols_per_block = 256 / c.G;
  const int64_t groups = (batch + cols_per_block - 1) / cols_per_block;
  const int iters = rqs_iters(ctx, blob_bytes, (int64_t)cols_per_block * dim * sizeof(T), groups);
  const int64_t grid = (groups + iters - 1) / iters;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_rqs_vjp: batch too large for one launch");
  constexpr int VW = Vec16<T>::N;
  {
    BjxProf prof_(ctx);
#define RV(V_, I_) hipLaunchKernelGGL((rqs_vjp_kernel<T, V_, I_>), dim3((unsigned)grid), dim3(256), blob_bytes, ctx->stream, blob, flag, K1, nstep_hi, dual, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, iters)
    if (c.V == VW) { if (inverse) RV(VW, true); else RV(VW, false); }
    else { if (inverse) RV(1, true); else RV(1, false); }
#undef RV
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

// ------------------------------------------------------------------ RQS knot pullback
// Cotangents of the knot arrays (widths, heights, derivatives: T[dim, K]) of with_logabsdet_jacobian for the elementwise
// spline and its inverse, summed over the batch.  With the element in bin k (knots k, k+1; knot 0 = -knot K) and
//   Δw, Δh, s = Δh/Δw, ξ = (x - w_k)/Δw, p = ξ(1-ξ), N = sξ² + d_k p, Dn = s + (d_k+1 + d_k - 2s) p, M = d_k+1 ξ² + 2sp + d_k(1-ξ)²
//   y = h_k + Δh N/Dn,   ℓ = log(s² M / Dn²)                                  (rational_quadratic_spline.jl:128-164, 266-297)
// the partials in (ξ, s, Δh, d_k, d_k+1) are
//   y_ξ = Δh (N_ξ Dn - N Dn_ξ)/Dn²      N_ξ = 2sξ + d_k(1-2ξ), Dn_ξ = (d_k+1 + d_k - 2s)(1-2ξ)
//   y_s = Δh (ξ² Dn - N (1-2p))/Dn²     y_Δh = N/Dn     y_dk = Δh p (Dn - N)/Dn²     y_dk+1 = -Δh N p/Dn²
//   ℓ_ξ = M_ξ/M - 2 Dn_ξ/Dn             ℓ_s = 2/s + 2p/M - 2(1-2p)/Dn     ℓ_dk = (1-ξ)²/M - 2p/Dn     ℓ_dk+1 = ξ²/M - 2p/Dn
// and ξ, s, Δh reach the knots through ∂ξ/∂w_k = (ξ-1)/Δw, ∂ξ/∂w_k+1 = -ξ/Δw, ∂s/∂w_k = s/Δw = -∂s/∂w_k+1,
// ∂s/∂h_k+1 = 1/Δw = -∂s/∂h_k, ∂Δh/∂h_k+1 = 1 = -∂Δh/∂h_k, ∂y/∂h_k = 1.  The inverse map x = f⁻¹(y) with log-det -ℓ(x)
// is the same accumulation at the point x with ȳ := -(x̄ - ℓ̄ ℓ_x)/f'(x), ℓ̄ := -ℓ̄ (implicit function theorem).
// Elements outside [-B, B] are the identity: no contribution.  d_1 and d_K are the constant 1 of the reference (:146-147).
//
// Every thread owns ONE row (and one of `cpp` column lanes) and a private slice of LDS accumulators [3K][threads]
// (bank = thread: conflict-free, no atomics, a fixed order of additions); the block sums its lanes in a fixed order and
// adds its table to the Float64 table of the call (one atomic add per entry and block).
// SHARED (round 6): ONE Float64 table [3][K][dim] per block, filled with `ds_add_f64` — the Float64 LDS atomic runs at 3.0 lane-adds
// per clock per CU on gfx950, nine times the Float32 one (0.33; scripts/probe_lds_atomics.hip, profiles/r06_lds_atomics.txt) — instead
// of 3·K private Float32 slots per thread (49 KiB per block, 3 blocks per CU, read-add-write chains of LDS latency and a 12 288-read
// fold at the end): 18 KiB per block, 8 blocks per CU, fire-and-forget adds, sums exact to Float64.
template <class T, class A, bool INV, bool SHARED>
__global__ void rqs_knot_vjp_kernel(const T* __restrict__ w, const T* __restrict__ h, const T* __restrict__ d, int K,
                                    const T* __restrict__ x, const T* __restrict__ gbar, const T* __restrict__ lbar,
                                    T* __restrict__ xbar, double* __restrict__ acc, int64_t dim, int64_t batch, int cpp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nthr = (int)blockDim.x, t = (int)threadIdx.x;
  const int64_t nk = (int64_t)K * dim;
  T* kw = reinterpret_cast<T*>(smem);
  T* kh = kw + nk;
  T* kd = kh + nk;
  A* priv = reinterpret_cast<A*>(smem + ((3 * nk * sizeof(T) + 15) / 16) * 16);       // [3K][nthr]
  double* tab = reinterpret_cast<double*>(smem + ((3 * nk * sizeof(T) + 15) / 16) * 16);   // SHARED: [3K][dim]
  for (int64_t i = t; i < nk; i += nthr) { kw[i] = w[i]; kh[i] = h[i]; kd[i] = d[i]; }
  if constexpr (SHARED) { for (int64_t i = t; i < 3 * nk; i += nthr) tab[i] = 0.0; }
  else { for (int e = 0; e < 3 * K; ++e) priv[(size_t)e * nthr + t] = A(0); }
  __syncthreads();
  const bool active = t < cpp * (int)dim;
  const int row = t % (int)dim, cl = t / (int)dim;
  using F = Fast<T>;
  const T *w_ = kw + row, *h_ = kh + row, *d_ = kd + row;
  const int st = (int)dim;                                          // 32-bit LDS indices: 3·K·dim words fit the tile
  const int64_t passes = (batch + cpp - 1) / cpp;
  // (plain read-add-write of the thread's own slots below; `ds_add_f32` on the same addresses was measured 2.7x SLOWER: 4.8 vs 1.75 ms)
  // one pass of look-ahead: the next pass's x, ȳ, ℓ̄ are in flight while this one is evaluated (a pass is a few hundred dependent
  // instructions behind three loads; without the look-ahead every pass paid the full memory latency first)
  // A ring of LA prefetched (x, ȳ, ℓ̄) triples in registers, slot u refilled for pass ps + LA·grid as soon as it has been consumed.
  // LA = 1 (round 3 tried 4: 1.10 -> 1.54 ms at K = 16, dim = 32, 2^22 columns — the four unrolled bodies, ~300 instructions each with
  // divergent early exits, cost more than the extra loads in flight bring: the kernel is bound by its LDS read-modify-write chains).
  constexpr int LA = 1;                    // (round 6, SHARED: two columns in flight per thread measured too — 0.682 vs 0.689 ms, nothing)
  T qx[LA], qg[LA], ql[LA];
  bool qok[LA];
  auto fetch = [&](int64_t ps, T& fx, T& fg, T& fl, bool& fok) {
    const int64_t col = ps * cpp + cl;
    fok = active && ps < passes && col < batch;
    fx = T(0); fg = T(0); fl = T(0);
    if (fok) { const int64_t idx = col * dim + row; fx = x[idx]; fg = gbar[idx]; fl = lbar ? lbar[col] : T(0); }
  };
#pragma unroll
  for (int u = 0; u < LA; ++u) fetch((int64_t)blockIdx.x + (int64_t)u * gridDim.x, qx[u], qg[u], ql[u], qok[u]);
  for (int64_t ps0 = blockIdx.x; ps0 < passes; ps0 += (int64_t)LA * gridDim.x) {
#pragma unroll
  for (int u = 0; u < LA; ++u) {
    const int64_t ps = ps0 + (int64_t)u * gridDim.x;
    const bool ok = qok[u];
    const T xin = qx[u];
    T g = qg[u], lb = ql[u];
    fetch(ps + (int64_t)LA * gridDim.x, qx[u], qg[u], ql[u], qok[u]);
    if (!ok) continue;
    const int64_t oidx = (ps * cpp + cl) * dim + row;
    const T wK = w_[(K - 1) * st], hK = h_[(K - 1) * st];
    if (!(d_abs(xin) < (INV ? hK : wK))) {                            // identity outside (-B, B): x̄ = ȳ, no knot contribution (NaN: nothing)
      if (xbar) xbar[oidx] = g;
      continue;
    }
    int k;                                                            // bin k: knots k, k+1 (1-based), knot 0 = -knot K
    {
      const T* sv = INV ? h_ : w_;                                    // Base.searchsortedfirst (ssf above) with 32-bit indices
      int lo = 0, hi = K + 1;
      while (lo < hi - 1) { const int m = lo + ((hi - lo) >> 1); if (sv[(m - 1) * st] < xin) lo = m; else hi = m; }
      k = hi - 1;
    }
    const T w_k = (k == 0) ? -wK : w_[(k - 1) * st];
    const T wd = w_[k * st] - w_k;
    const T h_k = (k == 0) ? -hK : h_[(k - 1) * st];
    const T dy = h_[k * st] - h_k;
    const T iw = F::rcp(wd);
    const T s = dy * iw;
    const T d_k = (k == 0) ? T(1) : d_[(k - 1) * st];
    const T d_k1 = (k == K - 1) ? T(1) : d_[k * st];
    const T ds = d_k1 + d_k - 2 * s, dd = d_k1 - d_k;
    T xi;
    if (!INV) xi = (xin - w_k) * iw;
    else {
      const T yh = xin - h_k;
      const T a1 = dy * (s - d_k) + yh * ds;
      const T a2 = dy * d_k - yh * ds;
      const T q = s * yh;
      xi = (q + q) * F::rcp(a2 + F::sqrt(a2 * a2 + 4 * (a1 * q)));
    }
    const T p = xi - xi * xi, om = T(1) - (xi + xi);
    const T den = s + ds * p, rden = F::rcp(den);
    const T M = (d_k + dd * xi) - ds * p, rM = F::rcp(M);
    const T N = xi * (d_k + (s - d_k) * xi);
    const T l_xi = (dd - ds * om) * rM - T(2) * ds * om * rden;
    if (INV) {                                                        // implicit function theorem at x = f⁻¹(y)
      const T sr = s * rden;
      const T J = M * (sr * sr);
      g = -(g - lb * (l_xi * iw)) * F::rcp(J);
      lb = -lb;
    }
    const T r2 = rden * rden;
    const T y_xi = dy * ((T(2) * s * xi + d_k * om) * den - N * (ds * om)) * r2;
    const T y_s = dy * (xi * xi * den - N * (T(1) - 2 * p)) * r2;
    const T y_dh = N * rden;
    const T y_dk = dy * p * (den - N) * r2;
    const T y_dk1 = -dy * N * p * r2;
    const T l_s = T(2) * F::rcp(s) + T(2) * p * rM - T(2) * (T(1) - 2 * p) * rden;
    const T omx = T(1) - xi;
    const T l_dk = omx * omx * rM - T(2) * p * rden;
    const T l_dk1 = xi * xi * rM - T(2) * p * rden;
    const T Gxi = g * y_xi + lb * l_xi, Gs = g * y_s + lb * l_s, Gdh = g * y_dh;
    // the input cotangent of the same element (what bjx_rqs_vjp returns): forward ȳ f' + ℓ̄ ℓ_x = Gξ/Δw; inverse (x̄ - ℓ̄ ℓ_x)/f' = -g here
    if (xbar) xbar[oidx] = INV ? -g : Gxi * iw;
    const T gw_k = (Gxi * (xi - T(1)) + Gs * s) * iw, gw_k1 = -(Gxi * xi + Gs * s) * iw;
    const T gh_k = g - Gdh - Gs * iw, gh_k1 = Gdh + Gs * iw;
    // knot k (1-based) lives at index k-1; knot 0 is -knot K.  Six slots, always distinct (k = 0: the unread derivative of the last
    // knot takes the +0): all six reads, then the adds, then the writes — one LDS round trip instead of six dependent ones.
    const int lo_i = k == 0 ? K - 1 : k - 1;
    const T sgn = k == 0 ? T(-1) : T(1);
    if constexpr (SHARED) {
      double* const c0 = tab + lo_i * st + row;                        // knot k (index lo_i) and knot k+1 (index k) of this row, three tables nk apart
      double* const c1 = tab + k * st + row;
      atomicAdd(c0, (double)(sgn * gw_k));
      atomicAdd(c1, (double)gw_k1);
      atomicAdd(c0 + nk, (double)(sgn * gh_k));
      atomicAdd(c1 + nk, (double)gh_k1);
      if (k != 0) atomicAdd(c0 + 2 * nk, (double)(g * y_dk + lb * l_dk));
      if (k != K - 1) atomicAdd(c1 + 2 * nk, (double)(g * y_dk1 + lb * l_dk1));
      continue;
    }
    A* const pw0 = priv + (0 * K + lo_i) * nthr + t;
    A* const pw1 = priv + (0 * K + k) * nthr + t;
    A* const ph0 = priv + (1 * K + lo_i) * nthr + t;
    A* const ph1 = priv + (1 * K + k) * nthr + t;
    A* const pd0 = priv + (2 * K + lo_i) * nthr + t;
    A* const pd1 = priv + (2 * K + k) * nthr + t;
    const A q0 = *pw0, q1 = *pw1, q2 = *ph0, q3 = *ph1, q4 = *pd0, q5 = *pd1;
    *pw0 = q0 + (A)(sgn * gw_k);
    *pw1 = q1 + (A)gw_k1;
    *ph0 = q2 + (A)(sgn * gh_k);
    *ph1 = q3 + (A)gh_k1;
    *pd0 = q4 + (A)(k == 0 ? T(0) : g * y_dk + lb * l_dk);
    *pd1 = q5 + (A)(k == K - 1 ? T(0) : g * y_dk1 + lb * l_dk1);
  }
  }
  __syncthreads();
  if constexpr (SHARED) {
    // the block's table, as it is, to ITS slice of the partials array (plain coalesced stores: 2 048 blocks x 1 536 Float64 global
    // atomics on the same 96 cache lines were a serial tail of ~0.1 ms); rqs_knot_fold_kernel sums the slices in block order
    double* mine = acc + (int64_t)blockIdx.x * 3 * nk;
    for (int64_t o = t; o < 3 * nk; o += nthr) mine[o] = tab[o];
    return;
  }
  for (int64_t o = t; o < 3 * nk; o += nthr) {                         // o = (table*K + knot)*dim + row
    const int r = (int)(o % dim);
    const int64_t e = o / dim;
    double sum = 0.0;
    for (int c = 0; c < cpp; ++c) sum += (double)priv[(size_t)e * nthr + c * (int)dim + r];
    if (sum != 0.0) atomicAdd(acc + o, sum);
  }
}
// (Round 4 tried the pack / LDS-blob skeleton of rqs_vjp_kernel with ONE table per block: profiles/r04_rqs_knots.md.  The front end
//  alone runs at 0.54 ms against this kernel's 1.12 ms, but the accumulation has no cheap race-free form there: LDS Float32 atomics run
//  at 0.33 lane-adds per clock per CU on gfx950 — 28x slower than ds_add_u32 (scripts/probe_lds_atomics.hip) — which made that kernel
//  4.2 ms; fixed-point integer atomics cost 6-10 VALU per contribution in a VALU-bound kernel; private accumulators are what this
//  kernel already has.)
// Σ over the blocks' tables, fixed order: thread (o, c) sums the slices [c·per, (c+1)·per) of entry o -> stage[c][o]; the out kernel
// adds the C stage rows in order.  Deterministic (no floating-point atomics anywhere on this path).
__global__ __launch_bounds__(256) void rqs_knot_fold_kernel(const double* __restrict__ parts, int nb, int64_t n3, int per, double* __restrict__ stage) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n3) return;
  const int c = blockIdx.y;
  const int lo = c * per, hi = lo + per < nb ? lo + per : nb;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = lo;
  for (; b + 3 < hi; b += 4) {
    s0 += parts[(int64_t)b * n3 + o]; s1 += parts[(int64_t)(b + 1) * n3 + o];
    s2 += parts[(int64_t)(b + 2) * n3 + o]; s3 += parts[(int64_t)(b + 3) * n3 + o];
  }
  for (; b < hi; ++b) s0 += parts[(int64_t)b * n3 + o];
  stage[(int64_t)c * n3 + o] = (s0 + s1) + (s2 + s3);
}
template <class T>
__global__ __launch_bounds__(256) void rqs_knot_vjp_out_stage_kernel(const double* __restrict__ stage, int C_, int K, int64_t dim, T* wb, T* hb, T* db) {
  const int64_t nk = (int64_t)K * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int q = 0; q < C_; ++q) { const double* st = stage + (int64_t)q * 3 * nk; a += st[i]; b += st[nk + i]; c += st[2 * nk + i]; }
    wb[i] = (T)a; hb[i] = (T)b; db[i] = (T)c;
  }
}

template <class T>
__global__ __launch_bounds__(256) void rqs_knot_vjp_out_kernel(const double* __restrict__ acc, int K, int64_t dim, T* wb, T* hb, T* db) {
  const int64_t nk = (int64_t)K * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    wb[i] = (T)acc[i]; hb[i] = (T)acc[nk + i]; db[i] = (T)acc[2 * nk + i];
  }
}

template <class T>
int rqs_knot_vjp_impl(bjx_ctx* ctx, int inverse, const T* w, const T* h, const T* d, int K, const T* in, const T* out_bar, const T* ladj_bar,
                      T* in_bar, T* wb, T* hb, T* db, int64_t dim, int64_t batch) {
  const int64_t nk = (int64_t)K * dim;
  BJX_REQUIRE(ctx, dim <= 256, BJX_ERR_UNSUPPORTED, "bjx_rqs_vjp_knots: at most 256 rows (got %lld)", (long long)dim);
  BJX_REQUIRE(ctx, (size_t)(3 * nk) * sizeof(double) <= BJX_SCRATCH_BYTES, BJX_ERR_UNSUPPORTED, "bjx_rqs_vjp_knots: knot table too large");
  double* acc = reinterpret_cast<double*>(ctx->scratch);
  // tuning switch: 0 = the private Float32 slices of rounds 2-5; n > 0 = the shared Float64 table with at most n blocks per CU
  static const int shared_tab = getenv("BJX_RQS_KNOTS_SHARED") ? atoi(getenv("BJX_RQS_KNOTS_SHARED")) : 8;
  const size_t shared_bytes = ((size_t)(3 * nk) * sizeof(T) + 15) / 16 * 16 + (size_t)(3 * nk) * sizeof(double);
  if (batch > 0 && shared_tab && shared_bytes <= 60 * 1024) {
    using A = T;
    const int nthr = 256;
    const int cpp = nthr / (int)dim;                                   // whole columns per pass (threads past cpp·dim idle)
    const int64_t passes = (batch + cpp - 1) / cpp;
    int per_cu = (int)((BJX_LDS_MAX - 2048) / shared_bytes);
    if (per_cu < 1) per_cu = 1;
    if (per_cu * nthr > 2048) per_cu = 2048 / nthr;
    if (per_cu > shared_tab) per_cu = shared_tab;        // (every block ends with one Float64 global atomic per table entry)
    int64_t grid = (int64_t)ctx->num_cu * per_cu;
    if (grid > passes) grid = passes;
    // partials: one [3][K][dim] Float64 table per block, then C stage rows (workspace of the context, grown on demand)
    constexpr int C_ = 32;
    const int64_t n3 = 3 * nk;
    { const int rc = bjx_ensure_big_ws(ctx, (size_t)(grid + C_) * n3 * sizeof(double)); if (rc) return rc; }
    double* parts = reinterpret_cast<double*>(ctx->big_ws);
    double* stage = parts + grid * n3;
    {
      BjxProf prof_(ctx);
      if (inverse) hipLaunchKernelGGL((rqs_knot_vjp_kernel<T, A, true, true>), dim3((unsigned)grid), dim3(nthr), shared_bytes, ctx->stream, w, h, d, K, in, out_bar, ladj_bar, in_bar, parts, dim, batch, cpp);
      else hipLaunchKernelGGL((rqs_knot_vjp_kernel<T, A, false, true>), dim3((unsigned)grid), dim3(nthr), shared_bytes, ctx->stream, w, h, d, K, in, out_bar, ladj_bar, in_bar, parts, dim, batch, cpp);
      BJX_CHECK_LAUNCH(ctx);
    }
    const int per = (int)((grid + C_ - 1) / C_);
    hipLaunchKernelGGL(rqs_knot_fold_kernel, dim3((unsigned)((n3 + 255) / 256), C_), dim3(256), 0, ctx->stream, parts, (int)grid, n3, per, stage);
    BJX_CHECK_LAUNCH(ctx);
    int g3 = (int)((nk + 255) / 256);
    if (g3 > 1024) g3 = 1024;
    hipLaunchKernelGGL(rqs_knot_vjp_out_stage_kernel<T>, dim3(g3), dim3(256), 0, ctx->stream, stage, C_, K, dim, wb, hb, db);
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  }
  BJX_HIP(ctx, hipMemsetAsync(acc, 0, (size_t)(3 * nk) * sizeof(double), ctx->stream));
  if (batch > 0) {
    using A = T;                                                     // accumulator type of the private slices
    int nthr = 256;
    auto bytes = [&](int n) { return ((size_t)(3 * nk) * sizeof(T) + 15) / 16 * 16 + (size_t)3 * K * n * sizeof(A); };
    while (nthr > 16 && nthr / 2 >= dim && bytes(nthr) > 60 * 1024) nthr /= 2;      // a thread per row at least
    BJX_REQUIRE(ctx, bytes(nthr) <= 64 * 1024, BJX_ERR_UNSUPPORTED, "bjx_rqs_vjp_knots: %d knots x %lld rows do not fit the LDS accumulators", K, (long long)dim);
    const int cpp = nthr / (int)dim;
    const int64_t passes = (batch + cpp - 1) / cpp;
    int per_cu = (int)((BJX_LDS_MAX - 2048) / bytes(nthr));            // as many blocks per CU as their LDS slices allow (the walk is latency-bound)
    if (per_cu < 1) per_cu = 1;
    if (per_cu * nthr > 2048) per_cu = 2048 / nthr;
    int64_t grid = (int64_t)ctx->num_cu * per_cu;
    if (grid > passes) grid = passes;
    BjxProf prof_(ctx);
    if (inverse) hipLaunchKernelGGL((rqs_knot_vjp_kernel<T, A, true, false>), dim3((unsigned)grid), dim3(nthr), bytes(nthr), ctx->stream, w, h, d, K, in, out_bar, ladj_bar, in_bar, acc, dim, batch, cpp);
    else hipLaunchKernelGGL((rqs_knot_vjp_kernel<T, A, false, false>), dim3((unsigned)grid), dim3(nthr), bytes(nthr), ctx->stream, w, h, d, K, in, out_bar, ladj_bar, in_bar, acc, dim, batch, cpp);
    BJX_CHECK_LAUNCH(ctx);
  }
  int g2 = (int)((nk + 255) / 256);
  if (g2 > 1024) g2 = 1024;
  if (g2 < 1) g2 = 1;
  hipLaunchKernelGGL(rqs_knot_vjp_out_kernel<T>, dim3(g2), dim3(256), 0, ctx->stream, acc, K, dim, wb, hb, db);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

// Pullback of the B-constructor (rational_quadratic_spline.jl:109-123; rqs_params_kernel above): knots
// c_j = 2B Σ_{i<=j} softmax(raw)_i - B (c_0 = -B constant) and d_j = log1pexp(raw_d_j) for the interior knots.
//   p̄_i = 2B Σ_{j>=i} c̄_j,   raw̄_i = p_i (p̄_i - Σ_m p_m p̄_m),   raw̄_d_j = d̄_j sigmoid(raw_d_j).
template <class T>
__global__ __launch_bounds__(256) void rqs_params_vjp_kernel(const T* rw, const T* rh, const T* rd, int K, int64_t dim, T B,
                                                             const T* wb, const T* hb, const T* db, T* rwb, T* rhb, T* rdb) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (int64_t)gridDim.x * blockDim.x) {
    for (int pass = 0; pass < 2; ++pass) {
      const T* r = pass == 0 ? rw : rh;
      const T* cb = pass == 0 ? wb : hb;                 // cotangents of the K+1 knots [dim, K+1]
      T* o = pass == 0 ? rwb : rhb;
      T mx = r[i];
      for (int k = 1; k < K; ++k) mx = d_max(mx, r[(int64_t)k * dim + i]);
      T s = T(0);
      for (int k = 0; k < K; ++k) s += d_exp(r[(int64_t)k * dim + i] - mx);
      T tail = T(0), dot = T(0);
      for (int k = K - 1; k >= 0; --k) {                 // p̄_k = 2B Σ_{j>=k} c̄_{j+1}
        tail += cb[(int64_t)(k + 1) * dim + i];
        dot += (d_exp(r[(int64_t)k * dim + i] - mx) / s) * ((2 * B) * tail);
      }
      tail = T(0);
      for (int k = K - 1; k >= 0; --k) {
        tail += cb[(int64_t)(k + 1) * dim + i];
        o[(int64_t)k * dim + i] = (d_exp(r[(int64_t)k * dim + i] - mx) / s) * ((2 * B) * tail - dot);
      }
    }
    for (int k = 0; k < K - 1; ++k) {
      const T v = rd[(int64_t)k * dim + i];
      rdb[(int64_t)k * dim + i] = db[(int64_t)(k + 1) * dim + i] / (T(1) + d_exp(-v));
    }
  }
}

template <class T>
int bn_impl(bjx_ctx* ctx, int inverse, const T* b, const T* logs, const T* m, const T* v, T eps, const T* in, T* out,
            T* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  hipLaunchKernelGGL(bn_const_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, logs, v, eps, dim, batch, inverse, ctx->consts);
  BJX_CHECK_LAUNCH(ctx);
  const bool lds = (size_t)dim * 4 * sizeof(T) <= 60 * 1024;
  const size_t fsm = lds ? (size_t)dim * 4 * sizeof(T) : 0;
  if (!inverse) { BnF<T, false> f{b, logs, m, v, eps, dim, lds ? 1 : 0, 0.0, ctx->consts}; return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0); }
  BnF<T, true> f{b, logs, m, v, eps, dim, lds ? 1 : 0, 0.0, ctx->consts};
  return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
}

// ------------------------------------------------------------------ InvertibleBatchNorm, training mode
// normalise.jl:51-60 (`istraining() == true`): the batch statistics of every channel (= row for a 2-D
// input, :43-47) replace the moving ones and the moving ones are updated:
//   m = mean(x; dims=batch);  v = sum((x .- m).^2; dims=batch) ./ n
//   bn.m = (1-mtm) bn.m + mtm m;   bn.v = (1-mtm) bn.v + (mtm n/(n-1)) v
// This is the one place on the hot path with a cross-batch reduction (SURVEY.md §8e "Exception"):
//   1. bn_stats_kernel   — every block streams a slab of columns, lanes along the rows, Float64 Σx and Σx² per
//                          row in registers, column groups of the block combined through LDS -> partial[block][row][2]
//   2. bn_stats_reduce   — fixed-order sum over the blocks -> stats[0..dim) = Σx, [dim..2dim) = Σx², [2dim] = n
//   3. (sharded batch)   — ONE RCCL all-reduce of those 2·dim+1 doubles (bjx_comm_init'ed context)
//   4. bn_train_finalize — mean, biased variance (Σx²/n - mean², Float64), moving-statistics update, batch
//                          statistics for the apply pass, Σ_c (logs_c - log(v_c+eps)/2)
//   5. the eval-mode apply kernel with the batch statistics.
// x is read twice (statistics, apply): 3·dim·sizeof(T) + sizeof(T) bytes per sample is the algorithmic traffic.
template <class T, int V, int R>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, const T* __restrict__ shift, int64_t dim, int64_t batch, int G, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);            // [cols_per_block][dim][2]
  const int gl = threadIdx.x & (G - 1), cg = threadIdx.x / G;
  const int cols_per_block = 256 / G;
  const int64_t nvc = dim / V;                               // packs per column; lane gl owns packs gl, gl+G, ... (R of them)
  double sx[R][V], sxx[R][V];
  // Sums of x - c and (x - c)^2 with a per-row shift c that every rank holds identically (the moving mean): the
  // variance S2/n - (S1/n)^2 then cancels on the scale of the batch spread, not of |mean| (the reference's two-pass
  // sum((x - m)^2)/n, normalise.jl:54, has no cancellation; raw one-pass sums lose it for Float64 data with |mean| >> std)
  T sh[R][V];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      sx[k][j] = 0.0; sxx[k][j] = 0.0;
      sh[k][j] = (shift && gl + k * G < nvc) ? shift[(int64_t)(gl + k * G) * V + j] : T(0);
    }
  constexpr int U = R == 1 ? 8 : (R == 2 ? 4 : 2);          // columns in flight per lane group
  const int64_t stride = (int64_t)gridDim.x * cols_per_block;
  int64_t col = (int64_t)blockIdx.x * cols_per_block + cg;
  for (; col + (U - 1) * stride < batch; col += U * stride) {
    Pack<T, V> p[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < R; ++k)
        if (gl + k * G < nvc) p[u][k] = load_pack<T, V, false>(x + (col + u * stride) * dim + (int64_t)(gl + k * G) * V);
    // the U columns of a trip are summed in T first (U <= 8 terms), then added to the Float64 accumulators: one
    // conversion + two Float64 adds per row and trip instead of per element
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (gl + k * G < nvc) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          T s1 = T(0), s2 = T(0);
#pragma unroll
          for (int u = 0; u < U; ++u) { const T v = p[u][k].v[j] - sh[k][j]; s1 += v; s2 += v * v; }
          sx[k][j] += (double)s1; sxx[k][j] += (double)s2;
        }
      }
  }
  for (; col < batch; col += stride) {
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (gl + k * G < nvc) {
        Pack<T, V> p = load_pack<T, V, false>(x + col * dim + (int64_t)(gl + k * G) * V);
#pragma unroll
        for (int j = 0; j < V; ++j) { const double v = (double)(p.v[j] - sh[k][j]); sx[k][j] += v; sxx[k][j] += v * v; }
      }
  }
  // combine the column groups of the block in a fixed order
#pragma unroll
  for (int k = 0; k < R; ++k)
    if (gl + k * G < nvc) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const size_t row = (size_t)(gl + k * G) * V + j;
        red[((size_t)cg * dim + row) * 2] = sx[k][j];
        red[((size_t)cg * dim + row) * 2 + 1] = sxx[k][j];
      }
    }
  __syncthreads();
  for (int64_t r = threadIdx.x; r < dim; r += 256) {
    double a = 0.0, b = 0.0;
    for (int c = 0; c < cols_per_block; ++c) { a += red[((size_t)c * dim + r) * 2]; b += red[((size_t)c * dim + r) * 2 + 1]; }
    partial[((size_t)blockIdx.x * dim + r) * 2] = a;
    partial[((size_t)blockIdx.x * dim + r) * 2 + 1] = b;
  }
}

// Fixed-order sum of the per-block partials.  A block owns 64 rows; its four waves take a quarter of the partial
// sets each, every thread with four independent accumulator pairs (8 loads in flight), combined through LDS in a
// fixed order.  (One thread per row walking all sets — 2048 dependent-latency loads — took 250 µs next to a 365 µs
// streaming pass; launch with grid = ceil(dim / 64), 256 threads.)
__global__ __launch_bounds__(256) void bn_stats_reduce_kernel(const double* __restrict__ partial, int nblocks, int64_t dim, int64_t batch, double* __restrict__ stats,
                                                              double* __restrict__ stats2 = nullptr, double* __restrict__ statn = nullptr) {
  // stats2 / statn: where the second sums and the count go when `dim` rows are a WINDOW of a taller problem (default: stats + dim, stats + 2 dim)
  if (!stats2) stats2 = stats + dim;
  if (!statn) statn = stats + 2 * dim;
  __shared__ double red[2][4][64];
  const int rl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 64 + rl;
  const int k0 = (int)((int64_t)nblocks * q / 4), k1 = (int)((int64_t)nblocks * (q + 1) / 4);
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  if (r < dim) {
    const double2* p = reinterpret_cast<const double2*>(partial) + r;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const double2 v = p[(size_t)(k + j) * dim]; a[j] += v.x; b[j] += v.y; }
    }
    for (; k < k1; ++k) { const double2 v = p[(size_t)k * dim]; a[0] += v.x; b[0] += v.y; }
  }
  red[0][q][rl] = (a[0] + a[1]) + (a[2] + a[3]);
  red[1][q][rl] = (b[0] + b[1]) + (b[2] + b[3]);
  __syncthreads();
  if (q == 0 && r < dim) {
    stats[r] = (red[0][0][rl] + red[0][1][rl]) + (red[0][2][rl] + red[0][3][rl]);
    stats2[r] = (red[1][0][rl] + red[1][1][rl]) + (red[1][2][rl] + red[1][3][rl]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) statn[0] = (double)batch;
}

// ------------------------------------------------------------------ row moments over the batch (parameter pullbacks of per-row affine stages)
// out[i] = Σ_n a[i,n],  out[dim + i] = Σ_n a[i,n]·b[i,n]  (b = NULL: a²), Float64, fixed order.  With a = the input
// cotangent z̄ of a chain `tail ∘ Shift(μ) ∘ Scale(σ)` and b = its input z these are the parameter cotangents of the
// leading per-row affine stage — the mean-field family of ADVI:  μ̄ = Σ z̄/σ,  σ̄ = (Σ z̄ z + Σ ℓ̄)/σ.
// Same streaming shape as bn_stats_kernel (lanes along the rows, Float64 accumulators, LDS combine, one partial per block).
template <class T, int V, int R>
__global__ __launch_bounds__(256) void row_moments_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t dim, int64_t batch, int G,
                                                          double* __restrict__ partial, int64_t ld, int64_t prow) {
  // dim rows per WINDOW of columns that are ld elements apart; blockIdx.y picks the window (rows blockIdx.y * dim ... of the `prow`
  // rows this launch covers: the last window may be shorter); the partials form one [block][prow][2] array
  const int64_t w0 = (int64_t)blockIdx.y * dim;
  a += w0; if (b) b += w0;
  if (w0 + dim > prow) dim = prow - w0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  const int gl = threadIdx.x & (G - 1), cg = threadIdx.x / G;
  const int cols_per_block = 256 / G;
  const int64_t nvc = dim / V;
  double s1[R][V], s2[R][V];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[k][j] = 0.0; s2[k][j] = 0.0; }
  constexpr int U = R == 1 ? 4 : (R == 2 ? 2 : 1);
  const int64_t stride = (int64_t)gridDim.x * cols_per_block;
  for (int64_t col = (int64_t)blockIdx.x * cols_per_block + cg; col < batch; col += U * stride) {
    Pack<T, V> pa[U][R], pb[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int64_t c = col + u * stride;
        const bool ok = c < batch && gl + k * G < nvc;
        if (ok) { pa[u][k] = load_pack<T, V, false>(a + c * ld + (int64_t)(gl + k * G) * V); pb[u][k] = b ? load_pack<T, V, false>(b + c * ld + (int64_t)(gl + k * G) * V) : pa[u][k]; }
        else {
#pragma unroll
          for (int j = 0; j < V; ++j) { pa[u][k].v[j] = T(0); pb[u][k].v[j] = T(0); }
        }
      }
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        T t1 = T(0), t2 = T(0);
#pragma unroll
        for (int u = 0; u < U; ++u) { t1 += pa[u][k].v[j]; t2 += pa[u][k].v[j] * pb[u][k].v[j]; }
        s1[k][j] += (double)t1; s2[k][j] += (double)t2;
      }
  }
#pragma unroll
  for (int k = 0; k < R; ++k)
    if (gl + k * G < nvc) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const size_t row = (size_t)(gl + k * G) * V + j;
        red[((size_t)cg * dim + row) * 2] = s1[k][j];
        red[((size_t)cg * dim + row) * 2 + 1] = s2[k][j];
      }
    }
  __syncthreads();
  for (int64_t r = threadIdx.x; r < dim; r += 256) {
    double x = 0.0, y = 0.0;
    for (int c = 0; c < cols_per_block; ++c) { x += red[((size_t)c * dim + r) * 2]; y += red[((size_t)c * dim + r) * 2 + 1]; }
    partial[((size_t)blockIdx.x * prow + w0 + r) * 2] = x;
    partial[((size_t)blockIdx.x * prow + w0 + r) * 2 + 1] = y;
  }
}

// Rows [r0, r0 + rows) in windows of `win` rows (a multiple of V; one pack per lane: four columns in flight), the windows as
// blockIdx.y of ONE launch, then one fixed-order reduction of the block partials over all of them.
template <class T, int V>
int row_moments_windows(bjx_ctx* ctx, const T* a, const T* b, double* out, int64_t dim, int64_t batch, int64_t r0, int64_t rows, int64_t win) {
  if (win > rows) win = rows;
  const int64_t nwin = (rows + win - 1) / win;
  BJX_REQUIRE(ctx, nwin < 65536, BJX_ERR_UNSUPPORTED, "bjx_row_moments: too many rows");
  const int64_t nvc = win / V;
  int G = 1;
  while (G < 64 && G < nvc) G <<= 1;
  const int cols_per_block = 256 / G;
  int nblocks = (int)((batch + (int64_t)cols_per_block * 16 - 1) / ((int64_t)cols_per_block * 16));
  const int cap = nwin > 1 ? (int)(1024 / (nwin < 8 ? nwin : 8) < 64 ? 64 : 1024 / (nwin < 8 ? nwin : 8)) : 1024;   // ~1 024 blocks in all
  if (nblocks > cap) nblocks = cap;
  if (nblocks < 1) nblocks = 1;
  { int rc = bjx_ensure_partials(ctx, (size_t)nblocks * rows * 2); if (rc) return rc; }
  const size_t smem = (size_t)cols_per_block * win * 2 * sizeof(double);
  BJX_REQUIRE(ctx, smem <= BJX_LDS_MAX, BJX_ERR_UNSUPPORTED, "bjx_row_moments: LDS");
  {
    BjxProf prof_(ctx);
    bjx_allow_big_lds(row_moments_kernel<T, V, 1>, smem);
    hipLaunchKernelGGL((row_moments_kernel<T, V, 1>), dim3(nblocks, (unsigned)nwin), dim3(256), smem, ctx->stream, a + r0, b ? b + r0 : nullptr, win, batch, G, ctx->partials, dim, rows);
  }
  BJX_CHECK_LAUNCH(ctx);
  { BjxProf prof_(ctx);
  hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, ctx->stream, ctx->partials, nblocks, rows, batch, out + r0, out + dim + r0, out + 2 * dim); }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T>
int row_moments_impl(bjx_ctx* ctx, const T* a, const T* b, double* out, int64_t dim, int64_t batch) {
  if (dim == 0) return BJX_OK;
  if (batch == 0) { BJX_HIP(ctx, hipMemsetAsync(out, 0, (size_t)(2 * dim + 1) * sizeof(double), ctx->stream)); return BJX_OK; }
  // Any number of rows (round 4; the register accumulators hold 256 packs per column, and columns that were not whole aligned packs
  // fell to one-element packs: 256 rows — a mean-field family of 333 parameters was refused): windows of 64 packs of 16 bytes on
  // element-aligned addresses — the blocks of one grid —, the dim mod V rows that are left as a last launch of one-element packs.
  constexpr int VW = Vec16<T>::N;
  const int64_t whole = dim / VW * VW;
  if (whole > 0) { const int rc = row_moments_windows<T, VW>(ctx, a, b, out, dim, batch, 0, whole, (int64_t)64 * VW); if (rc) return rc; }
  if (whole < dim) { const int rc = row_moments_windows<T, 1>(ctx, a, b, out, dim, batch, whole, dim - whole, 64); if (rc) return rc; }
  return BJX_OK;
}

// batch statistics, moving-statistics update (normalise.jl:56-60) and the per-sample log-det constant (:63)
template <class T>
__global__ __launch_bounds__(256) void bn_train_finalize_kernel(const double* __restrict__ stats, int64_t dim, const T* __restrict__ logs, T eps, T mtm,
                                                                T* __restrict__ m_mov, T* __restrict__ v_mov, T* __restrict__ m_batch,
                                                                T* __restrict__ v_batch, int64_t local_batch, double* __restrict__ consts) {
  __shared__ double red[4];
  const double n = stats[2 * dim];
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < dim; r += blockDim.x) {
    // stats hold the sums of x - c, c = the moving mean BEFORE this update (the shift of bn_stats_kernel)
    const double dm = stats[r] / n;
    const double mean = (double)m_mov[r] + dm;
    double var = stats[dim + r] / n - dm * dm;               // biased (÷ n), :54
    if (var < 0.0) var = 0.0;
    const T mT = (T)mean, vT = (T)var;
    m_batch[r] = mT;
    v_batch[r] = vT;
    m_mov[r] = (T(1) - mtm) * m_mov[r] + mtm * mT;                               // :58
    v_mov[r] = (T(1) - mtm) * v_mov[r] + (T)((double)mtm * n / (n - 1.0)) * vT;  // :59
    s += (double)(logs[r] - d_log(vT + eps) / T(2));
  }
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double c = (red[0] + red[1]) + (red[2] + red[3]);
    consts[0] = c;
    consts[1] = c * (double)local_batch;
  }
}

int build_rowmap(bjx_ctx* ctx, const int32_t* idx1, int64_t n1, int64_t dim, int32_t** map_out) {
  BJX_REQUIRE(ctx, (size_t)dim * sizeof(int32_t) + 16 <= BJX_SCRATCH_BYTES, BJX_ERR_UNSUPPORTED, "coupling: dim %lld too large for the context scratch", (long long)dim);
  int32_t* map = static_cast<int32_t*>(ctx->scratch);
  BJX_HIP(ctx, hipMemsetAsync(map, 0xFF, (size_t)dim * sizeof(int32_t), ctx->stream));
  if (n1 > 0) {
    int* bad = reinterpret_cast<int*>(ctx->consts + 4);
    hipLaunchKernelGGL(rowmap_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream, idx1, n1, dim, map, bad);
    BJX_CHECK_LAUNCH(ctx);
  }
  *map_out = map;
  return BJX_OK;
}
}  // namespace

#define DISPATCH_DT(ctx, dt, CALL32, CALL64, NAME)                          \
  do {                                                                      \
    if ((dt) == BJX_F32) return CALL32;                                     \
    if ((dt) == BJX_F64) return CALL64;                                     \
    return bjx_fail((ctx), BJX_ERR_ARG, NAME ": bad dtype %d", (int)(dt));  \
  } while (0)

BJX_API int bjx_rqs(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* widths, const void* heights, const void* derivs,
                    int n_knots, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch,
                    uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0, BJX_ERR_SHAPE, "bjx_rqs: negative size");
  BJX_REQUIRE(ctx, n_knots >= 2, BJX_ERR_SHAPE, "bjx_rqs: need at least 2 knots, got %d", n_knots);
  BJX_REQUIRE(ctx, widths && heights && derivs && ((in && out) || dim * batch == 0), BJX_ERR_ARG, "bjx_rqs: null pointer");
  DISPATCH_DT(ctx, dt,
              rqs_impl<float>(ctx, inverse, (const float*)widths, (const float*)heights, (const float*)derivs, n_knots, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags),
              rqs_impl<double>(ctx, inverse, (const double*)widths, (const double*)heights, (const double*)derivs, n_knots, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags),
              "bjx_rqs");
}

BJX_API int bjx_row_moments(bjx_ctx* ctx, bjx_dtype dt, const void* a, const void* b, double* out, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0, BJX_ERR_SHAPE, "bjx_row_moments: negative size");
  BJX_REQUIRE(ctx, out && (a || dim * batch == 0), BJX_ERR_ARG, "bjx_row_moments: null pointer");
  DISPATCH_DT(ctx, dt, row_moments_impl<float>(ctx, (const float*)a, (const float*)b, out, dim, batch),
              row_moments_impl<double>(ctx, (const double*)a, (const double*)b, out, dim, batch), "bjx_row_moments");
}

BJX_API int bjx_rqs_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* widths, const void* heights, const void* derivs, int n_knots,
                        const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0, BJX_ERR_SHAPE, "bjx_rqs_vjp: negative size");
  BJX_REQUIRE(ctx, n_knots >= 2, BJX_ERR_SHAPE, "bjx_rqs_vjp: need at least 2 knots");
  BJX_REQUIRE(ctx, widths && heights && derivs && ((in && out_bar && in_bar) || dim * batch == 0), BJX_ERR_ARG, "bjx_rqs_vjp: null pointer");
  DISPATCH_DT(ctx, dt,
              rqs_vjp_impl<float>(ctx, inverse, (const float*)widths, (const float*)heights, (const float*)derivs, n_knots, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, dim, batch),
              rqs_vjp_impl<double>(ctx, inverse, (const double*)widths, (const double*)heights, (const double*)derivs, n_knots, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, dim, batch),
              "bjx_rqs_vjp");
}

BJX_API int bjx_rqs_params(bjx_ctx* ctx, bjx_dtype dt, const void* raw_w, const void* raw_h, const void* raw_d, int K,
                           int64_t dim, double B, void* widths, void* heights, void* derivs) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, K >= 1 && dim >= 0, BJX_ERR_SHAPE, "bjx_rqs_params: bad size");
  BJX_REQUIRE(ctx, raw_w && raw_h && (raw_d || K == 1) && widths && heights && derivs, BJX_ERR_ARG, "bjx_rqs_params: null pointer");
  if (dim == 0) return BJX_OK;
  int grid = (int)((dim + 255) / 256);
  if (grid > 1024) grid = 1024;
  if (dt == BJX_F32)
    hipLaunchKernelGGL(rqs_params_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)raw_w, (const float*)raw_h, (const float*)raw_d, K, dim, (float)B, (float*)widths, (float*)heights, (float*)derivs);
  else if (dt == BJX_F64)
    hipLaunchKernelGGL(rqs_params_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)raw_w, (const double*)raw_h, (const double*)raw_d, K, dim, B, (double*)widths, (double*)heights, (double*)derivs);
  else
    return bjx_fail(ctx, BJX_ERR_ARG, "bjx_rqs_params: bad dtype %d", (int)dt);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

BJX_API int bjx_rqs_vjp_knots(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* widths, const void* heights, const void* derivs, int n_knots,
                              const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, void* widths_bar, void* heights_bar, void* derivs_bar,
                              int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_rqs_vjp_knots: bad size");
  BJX_REQUIRE(ctx, n_knots >= 2, BJX_ERR_SHAPE, "bjx_rqs_vjp_knots: need at least 2 knots");
  BJX_REQUIRE(ctx, widths && heights && derivs && widths_bar && heights_bar && derivs_bar && ((in && out_bar) || batch == 0), BJX_ERR_ARG, "bjx_rqs_vjp_knots: null pointer");
  DISPATCH_DT(ctx, dt,
               rqs_knot_vjp_impl<float>(ctx, inverse, (const float*)widths, (const float*)heights, (const float*)derivs, n_knots, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, (float*)widths_bar, (float*)heights_bar, (float*)derivs_bar, dim, batch),
               rqs_knot_vjp_impl<double>(ctx, inverse, (const double*)widths, (const double*)heights, (const double*)derivs, n_knots, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, (double*)widths_bar, (double*)heights_bar, (double*)derivs_bar, dim, batch),
               "bjx_rqs_vjp_knots");
}

BJX_API int bjx_rqs_params_vjp(bjx_ctx* ctx, bjx_dtype dt, const void* raw_w, const void* raw_h, const void* raw_d, int K, int64_t dim, double B,
                               const void* widths_bar, const void* heights_bar, const void* derivs_bar, void* raw_w_bar, void* raw_h_bar, void* raw_d_bar) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, K >= 1 && dim >= 0, BJX_ERR_SHAPE, "bjx_rqs_params_vjp: bad size");
  BJX_REQUIRE(ctx, raw_w && raw_h && widths_bar && heights_bar && raw_w_bar && raw_h_bar && ((raw_d && derivs_bar && raw_d_bar) || K == 1), BJX_ERR_ARG, "bjx_rqs_params_vjp: null pointer");
  if (dim == 0) return BJX_OK;
  int grid = (int)((dim + 255) / 256);
  if (grid > 1024) grid = 1024;
  if (dt == BJX_F32)
    hipLaunchKernelGGL(rqs_params_vjp_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)raw_w, (const float*)raw_h, (const float*)raw_d, K, dim, (float)B, (const float*)widths_bar, (const float*)heights_bar, (const float*)derivs_bar, (float*)raw_w_bar, (float*)raw_h_bar, (float*)raw_d_bar);
  else if (dt == BJX_F64)
    hipLaunchKernelGGL(rqs_params_vjp_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)raw_w, (const double*)raw_h, (const double*)raw_d, K, dim, B, (const double*)widths_bar, (const double*)heights_bar, (const double*)derivs_bar, (double*)raw_w_bar, (double*)raw_h_bar, (double*)raw_d_bar);
  else
    return bjx_fail(ctx, BJX_ERR_ARG, "bjx_rqs_params_vjp: bad dtype %d", (int)dt);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

BJX_API int bjx_batchnorm(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* b, const void* logs, const void* m,
                          const void* v, double eps, const void* in, void* out, void* ladj_ps, double* ladj_sum,
                          int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_batchnorm: bad size");
  BJX_REQUIRE(ctx, b && logs && m && v && ((in && out) || batch == 0), BJX_ERR_ARG, "bjx_batchnorm: null pointer");
  DISPATCH_DT(ctx, dt,
              bn_impl<float>(ctx, inverse, (const float*)b, (const float*)logs, (const float*)m, (const float*)v, (float)eps, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags),
              bn_impl<double>(ctx, inverse, (const double*)b, (const double*)logs, (const double*)m, (const double*)v, eps, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags),
              "bjx_batchnorm");
}

namespace {
// statistics of this column block: stats[0..dim) = Σ (x - shift), [dim..2 dim) = Σ (x - shift)², [2 dim] = columns
template <class T>
int bn_stats_impl(bjx_ctx* ctx, const T* shift, const T* in, double* stats, double* partial, int max_blocks, int64_t dim, int64_t batch) {
  ColLaunch c = col_launch_cfg<T>(ctx, in, in, dim, batch);
  const int64_t nvc = dim / c.V;
  BJX_REQUIRE(ctx, nvc <= 256, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_train: %lld channels exceed the register-accumulator kernel (max %d)", (long long)dim, 256 * c.V);
  const int R = nvc <= c.G ? 1 : (nvc <= 2 * c.G ? 2 : 4);
  const int cols_per_block = 256 / c.G;
  int nblocks = (int)((batch + cols_per_block * 16 - 1) / (cols_per_block * 16));     // >= 16 columns per lane group
  if (nblocks > 1024) nblocks = 1024;
  if (nblocks > max_blocks) nblocks = max_blocks;       // fewer, longer blocks: the slabs of partial sums live in the context scratch
  if (nblocks < 1) nblocks = 1;
  const size_t smem = (size_t)cols_per_block * dim * 2 * sizeof(double);
  BJX_REQUIRE(ctx, smem <= 64 * 1024, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_train: LDS");
  constexpr int VW = Vec16<T>::N;
  {
    BjxProf prof_(ctx);
#define BN_ST(V_, R_) hipLaunchKernelGGL((bn_stats_kernel<T, V_, R_>), dim3(nblocks), dim3(256), smem, ctx->stream, in, shift, dim, batch, c.G, partial)
#define BN_STV(V_) do { if (R == 1) BN_ST(V_, 1); else if (R == 2) BN_ST(V_, 2); else BN_ST(V_, 4); } while (0)
    if (c.V == VW) BN_STV(VW); else BN_STV(1);
#undef BN_STV
#undef BN_ST
  }
  BJX_CHECK_LAUNCH(ctx);
  { BjxProf prof_(ctx);
  hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3((unsigned)((dim + 63) / 64)), dim3(256), 0, ctx->stream, partial, nblocks, dim, batch, stats, (double*)nullptr, (double*)nullptr); }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

// scratch layout of the training path: [stats 2 dim + 1 (+1 pad)][partials max_blocks*dim*2] doubles, then batch m / v (T)
template <class T> struct BnScratch {
  double* stats; double* partial; T* m_batch; T* v_batch; int max_blocks; bool ok;
  BnScratch(bjx_ctx* ctx, int64_t dim) {
    const size_t stats_n = 2 * (size_t)dim + 1;
    const size_t fixed = (stats_n + 1) * sizeof(double) + 2 * (size_t)dim * sizeof(T);
    const size_t fit = fixed < BJX_SCRATCH_BYTES ? (BJX_SCRATCH_BYTES - fixed) / ((size_t)dim * 2 * sizeof(double)) : 0;
    max_blocks = (int)(fit < 1024 ? fit : 1024);
    ok = max_blocks >= 1;
    stats = static_cast<double*>(ctx->scratch);
    partial = stats + stats_n + 1;
    m_batch = reinterpret_cast<T*>(partial + (size_t)(ok ? max_blocks : 0) * dim * 2);
    v_batch = m_batch + dim;
  }
};

// moving-statistics update + the eval-mode apply kernel with the batch statistics behind `stats` (GLOBAL sums)
template <class T>
int bn_apply_stats_impl(bjx_ctx* ctx, const T* b, const T* logs, T* m, T* v, T eps, T mtm, const double* stats, const T* in, T* out, T* ladj_ps,
                        double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  BnScratch<T> sc(ctx, dim);
  BJX_REQUIRE(ctx, sc.ok, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_train: scratch too small for %lld channels", (long long)dim);
  hipLaunchKernelGGL(bn_train_finalize_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, stats, dim, logs, eps, mtm, m, v, sc.m_batch, sc.v_batch, batch, ctx->consts);
  BJX_CHECK_LAUNCH(ctx);
  const bool lds = (size_t)dim * 4 * sizeof(T) <= 60 * 1024;
  const size_t fsm = lds ? (size_t)dim * 4 * sizeof(T) : 0;
  BnF<T, false> f{b, logs, sc.m_batch, sc.v_batch, eps, dim, lds ? 1 : 0, 0.0, ctx->consts};
  return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
}

template <class T>
int bn_train_impl(bjx_ctx* ctx, const T* b, const T* logs, T* m, T* v, T eps, T mtm, const T* in, T* out, T* ladj_ps, double* ladj_sum,
                  int64_t dim, int64_t batch, uint32_t flags) {
  BJX_REQUIRE(ctx, batch >= 1, BJX_ERR_SHAPE, "bjx_batchnorm_train: empty batch");
  BnScratch<T> sc(ctx, dim);
  BJX_REQUIRE(ctx, sc.ok, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_train: scratch too small for %lld channels", (long long)dim);
  int rc = bn_stats_impl<T>(ctx, m, in, sc.stats, sc.partial, sc.max_blocks, dim, batch);
  if (rc) return rc;
  if (ctx->comm && ctx->nranks > 1) {      // batch sharded over GPUs: the second collective of SURVEY.md §8(e)
    rc = bjx_allreduce_sum_f64(ctx, sc.stats, (int64_t)(2 * dim + 1));
    if (rc) return rc;
  }
  return bn_apply_stats_impl<T>(ctx, b, logs, m, v, eps, mtm, sc.stats, in, out, ladj_ps, ladj_sum, dim, batch, flags);
}
}  // namespace

BJX_API int bjx_batchnorm_stats(bjx_ctx* ctx, bjx_dtype dt, const void* shift, const void* in, double* stats, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_batchnorm_stats: bad size");
  BJX_REQUIRE(ctx, stats && (in || batch == 0), BJX_ERR_ARG, "bjx_batchnorm_stats: null pointer");
  if (batch == 0) { BJX_HIP(ctx, hipMemsetAsync(stats, 0, (size_t)(2 * dim + 1) * sizeof(double), ctx->stream)); return BJX_OK; }
  if (dt == BJX_F32) { BnScratch<float> sc(ctx, dim); BJX_REQUIRE(ctx, sc.ok, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_stats: too many channels");
    return bn_stats_impl<float>(ctx, (const float*)shift, (const float*)in, stats, sc.partial, sc.max_blocks, dim, batch); }
  if (dt == BJX_F64) { BnScratch<double> sc(ctx, dim); BJX_REQUIRE(ctx, sc.ok, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_stats: too many channels");
    return bn_stats_impl<double>(ctx, (const double*)shift, (const double*)in, stats, sc.partial, sc.max_blocks, dim, batch); }
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_batchnorm_stats: bad dtype %d", (int)dt);
}

BJX_API int bjx_batchnorm_train_apply(bjx_ctx* ctx, bjx_dtype dt, const void* b, const void* logs, void* m, void* v, double eps, double mtm,
                                      const double* stats, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch,
                                      uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_batchnorm_train_apply: bad size");
  BJX_REQUIRE(ctx, b && logs && m && v && stats && ((in && out) || batch == 0), BJX_ERR_ARG, "bjx_batchnorm_train_apply: null pointer");
  DISPATCH_DT(ctx, dt,
              bn_apply_stats_impl<float>(ctx, (const float*)b, (const float*)logs, (float*)m, (float*)v, (float)eps, (float)mtm, stats, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags),
              bn_apply_stats_impl<double>(ctx, (const double*)b, (const double*)logs, (double*)m, (double*)v, eps, mtm, stats, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags),
              "bjx_batchnorm_train_apply");
}

BJX_API int bjx_batchnorm_train(bjx_ctx* ctx, bjx_dtype dt, const void* b, const void* logs, void* m, void* v, double eps, double mtm,
                                const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_batchnorm_train: bad size");
  BJX_REQUIRE(ctx, b && logs && m && v && in && out, BJX_ERR_ARG, "bjx_batchnorm_train: null pointer");
  DISPATCH_DT(ctx, dt,
              bn_train_impl<float>(ctx, (const float*)b, (const float*)logs, (float*)m, (float*)v, (float)eps, (float)mtm, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags),
              bn_train_impl<double>(ctx, (const double*)b, (const double*)logs, (double*)m, (double*)v, eps, mtm, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags),
              "bjx_batchnorm_train");
}

namespace {
// ------------------------------------------------------------------ InvertibleBatchNorm, TRAINING-mode pullback
// normalise.jl:51-60: m = mean(x), v = Σ(x - m)²/N are functions of the batch; the reference leaves the adjoint to the AD
// package.  With σ = sqrt(v + ε), x̂ = (x - m)/σ, γ = exp(logs), y = γ x̂ + b, logabsdetjac[n] = Σ_c (logs_c - ½ log(v_c + ε)):
//   x̄ = (γ/σ) [ȳ - mean_n ȳ - x̂ mean_n(ȳ x̂)] - (Σ_n ℓ̄ / N) x̂ / σ        (the last term: ∂/∂x of -½ Σℓ̄ log(v + ε))
//      = p ȳ + q x + r   per channel,  p = γ/σ,  q = -(p mean(ȳ x̂) + Σℓ̄/(N σ))/σ,  r = -p mean ȳ - q m
//   b̄ = Σ_n ȳ,   l̄ogs = γ Σ_n ȳ x̂ + Σ_n ℓ̄,   Σ_n ȳ x̂ = (Σ ȳ x - m Σ ȳ)/σ
// The batch sums (Σȳ, Σȳx, N) come in as `moments` (bjx_row_moments(ȳ, x), all-reduced by the host when the batch is sharded).
template <class T>
__global__ __launch_bounds__(256) void bn_train_vjp_coef_kernel(const T* __restrict__ logs, const T* __restrict__ mean, const T* __restrict__ var, double eps,
                                                                const double* __restrict__ mom, const double* __restrict__ lsum, int dim,
                                                                T* __restrict__ coef /*[3 dim]: p | q | r*/, T* __restrict__ b_bar, T* __restrict__ logs_bar) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= dim) return;
  const double n = mom[2 * dim], Sy = mom[c], Syx = mom[dim + c], L = lsum ? *lsum : 0.0;
  const double m = (double)mean[c], sig = ::sqrt((double)var[c] + eps), gam = ::exp((double)logs[c]);
  const double Syxh = (Syx - m * Sy) / sig;
  const double pp = gam / sig;
  const double qq = -(pp * (Syxh / n) + L / (n * sig)) / sig;
  coef[c] = (T)pp;
  coef[dim + c] = (T)qq;
  coef[2 * dim + c] = (T)(-pp * (Sy / n) - qq * m);
  if (b_bar) b_bar[c] = (T)Sy;
  if (logs_bar) logs_bar[c] = (T)(gam * Syxh + L);
}
template <class T, int V>
__global__ __launch_bounds__(256) void bn_train_vjp_apply_kernel(const T* __restrict__ coef, const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ xb,
                                                                 int64_t dim, int64_t total) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= total) return;
  const int64_t row = i % dim;                           // V > 1 only when dim % V == 0: a pack stays inside one column
  const Pack<T, V> xv = load_pack<T, V, true>(x + i), gv = load_pack<T, V, true>(g + i);
  Pack<T, V> o;
#pragma unroll
  for (int j = 0; j < V; ++j) o.v[j] = coef[row + j] * gv.v[j] + (coef[dim + row + j] * xv.v[j] + coef[2 * dim + row + j]);
  store_pack<T, V, true>(xb + i, o);
}
template <class T>
int bn_train_vjp_impl(bjx_ctx* ctx, const T* logs, const T* mean, const T* var, double eps, const double* mom, const double* lsum, const T* in,
                      const T* out_bar, T* in_bar, T* b_bar, T* logs_bar, int64_t dim, int64_t batch) {
  BJX_REQUIRE(ctx, (size_t)3 * dim * sizeof(T) <= BJX_SCRATCH_BYTES, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_train_vjp: too many channels");
  T* coef = reinterpret_cast<T*>(ctx->scratch);
  hipLaunchKernelGGL((bn_train_vjp_coef_kernel<T>), dim3((unsigned)((dim + 255) / 256)), dim3(256), 0, ctx->stream, logs, mean, var, eps, mom, lsum, (int)dim, coef, b_bar, logs_bar);
  BJX_CHECK_LAUNCH(ctx);
  if (batch == 0 || !in_bar) return BJX_OK;
  constexpr int VW = Vec16<T>::N;
  const int64_t total = dim * batch;
  const bool vec = dim % VW == 0 && bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
  BjxProf prof_(ctx);
  if (vec) {
    const int64_t grid = (total / VW + 255) / 256;
    BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_train_vjp: input too large for one launch");
    hipLaunchKernelGGL((bn_train_vjp_apply_kernel<T, VW>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, coef, in, out_bar, in_bar, dim, total);
  } else {
    const int64_t grid = (total + 255) / 256;
    BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_batchnorm_train_vjp: input too large for one launch");
    hipLaunchKernelGGL((bn_train_vjp_apply_kernel<T, 1>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, coef, in, out_bar, in_bar, dim, total);
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_batchnorm_train_vjp(bjx_ctx* ctx, bjx_dtype dt, const void* logs, const void* mean, const void* var, double eps, const double* moments,
                                    const double* ladj_bar_sum, const void* in, const void* out_bar, void* in_bar, void* b_bar, void* logs_bar,
                                    int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_batchnorm_train_vjp: bad size");
  BJX_REQUIRE(ctx, logs && mean && var && moments && ((in && out_bar) || batch == 0 || !in_bar), BJX_ERR_ARG, "bjx_batchnorm_train_vjp: null pointer");
  BJX_REQUIRE(ctx, !in_bar || ((const void*)in_bar != in), BJX_ERR_ARG, "bjx_batchnorm_train_vjp: in_bar may not alias the primal input");
  DISPATCH_DT(ctx, dt,
              bn_train_vjp_impl<float>(ctx, (const float*)logs, (const float*)mean, (const float*)var, eps, moments, ladj_bar_sum, (const float*)in, (const float*)out_bar, (float*)in_bar, (float*)b_bar, (float*)logs_bar, dim, batch),
              bn_train_vjp_impl<double>(ctx, (const double*)logs, (const double*)mean, (const double*)var, eps, moments, ladj_bar_sum, (const double*)in, (const double*)out_bar, (double*)in_bar, (double*)b_bar, (double*)logs_bar, dim, batch),
              "bjx_batchnorm_train_vjp");
}

namespace {
// Permute through an LDS tile (permute.jl: y = x[src] per column): a block takes a contiguous run of columns with
// coalesced 16-byte loads, gathers the rows from LDS and leaves with coalesced 16-byte stores.  (The functor path
// gathers from global memory: four 4-byte loads + four index loads per pack, 68 % of the HBM roofline for a copy.)
template <class T, int V>
__global__ __launch_bounds__(256) void permute_lds_kernel(const int32_t* __restrict__ src, const T* __restrict__ x, T* __restrict__ y, int dim,
                                                          int64_t batch, int cols_per_block, int tile_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* ssrc = reinterpret_cast<int32_t*>(smem);
  T* tile = reinterpret_cast<T*>(smem + tile_off);
  for (int i = threadIdx.x; i < dim; i += blockDim.x) ssrc[i] = src[i];
  const int64_t col0 = (int64_t)blockIdx.x * cols_per_block;
  const int64_t left = batch - col0;
  const int total = (int)(left < cols_per_block ? left : cols_per_block) * dim;      // elements of this block's run
  const T* xb = x + col0 * dim;
  T* yb = y + col0 * dim;
  for (int e = threadIdx.x * V; e < total; e += 256 * V) {
    const Pack<T, V> p = load_pack<T, V, true>(xb + e);
    if constexpr (V > 1) *reinterpret_cast<typename Vec16<T>::type*>(tile + e) = __builtin_bit_cast(typename Vec16<T>::type, p);
    else tile[e] = p.v[0];
  }
  __syncthreads();
  int c = (threadIdx.x * V) / dim, r = (threadIdx.x * V) % dim;                     // V | dim: a pack stays inside one column
  const int dc = (256 * V) / dim, dr = (256 * V) % dim;
  for (int e = threadIdx.x * V; e < total; e += 256 * V) {
    Pack<T, V> o;
    const T* tc = tile + c * dim;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = tc[ssrc[r + j]];
    store_pack<T, V, true>(yb + e, o);
    c += dc; r += dr;
    if (r >= dim) { r -= dim; ++c; }
  }
}

inline int bjx_check_launch(bjx_ctx* ctx) { BJX_CHECK_LAUNCH(ctx); return BJX_OK; }
template <class T>
bool permute_lds(bjx_ctx* ctx, const int32_t* src, const T* in, T* out, int64_t dim, int64_t batch, int* rc) {
  constexpr int VW = Vec16<T>::N;
  *rc = BJX_OK;
  if (dim < 1 || dim > 4096 || batch < 1) return false;
  const int V = (dim % VW == 0 && bjx_aligned16(in) && bjx_aligned16(out)) ? VW : 1;
  int cpb = (int)(16384 / (dim * sizeof(T)));
  if (cpb < 1) cpb = 1;
  if (V > 1) { while (cpb > 1 && ((int64_t)cpb * dim * sizeof(T)) % 16 != 0) --cpb; }   // every block's run starts 16-byte aligned
  const int tile_off = (int)(((size_t)dim * sizeof(int32_t) + 15) / 16 * 16);
  const size_t smem = tile_off + (size_t)cpb * dim * sizeof(T);
  const int64_t grid = (batch + cpb - 1) / cpb;
  if (smem > 48 * 1024 || grid >= ((int64_t)1 << 31)) return false;
  BjxProf prof_(ctx);
  if (V == VW) hipLaunchKernelGGL((permute_lds_kernel<T, VW>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, src, in, out, (int)dim, batch, cpb, tile_off);
  else hipLaunchKernelGGL((permute_lds_kernel<T, 1>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, src, in, out, (int)dim, batch, cpb, tile_off);
  *rc = bjx_check_launch(ctx);
  return true;
}
}  // namespace

BJX_API int bjx_permute(bjx_ctx* ctx, bjx_dtype dt, const int32_t* src, const void* in, void* out, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0, BJX_ERR_SHAPE, "bjx_permute: negative size");
  BJX_REQUIRE(ctx, (src && in && out) || dim * batch == 0, BJX_ERR_ARG, "bjx_permute: null pointer");
  BJX_REQUIRE(ctx, in != out || dim * batch == 0, BJX_ERR_ARG, "bjx_permute: in-place permutation is not supported");
  {
    int rc = BJX_OK;
    if (dt == BJX_F32 && permute_lds<float>(ctx, src, (const float*)in, (float*)out, dim, batch, &rc)) return rc;
    if (dt == BJX_F64 && permute_lds<double>(ctx, src, (const double*)in, (double*)out, dim, batch, &rc)) return rc;
  }
  if (dt == BJX_F32) { PermuteF<float> f{src, 0.0, nullptr}; return launch_colgroup<float>(ctx, f, 0, (const float*)in, (float*)out, nullptr, nullptr, dim, batch, 0, 0.0); }
  if (dt == BJX_F64) { PermuteF<double> f{src, 0.0, nullptr}; return launch_colgroup<double>(ctx, f, 0, (const double*)in, (double*)out, nullptr, nullptr, dim, batch, 0, 0.0); }
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_permute: bad dtype %d", (int)dt);
}

namespace {
template <class T>
int coupling_affine_impl(bjx_ctx* ctx, int inverse, const int32_t* idx1, int64_t n1, const T* scale, const T* shift, const T* in,
                         T* out, T* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  int32_t* map = nullptr;
  int rc = build_rowmap(ctx, idx1, n1, dim, &map);
  if (rc) return rc;
  const int in_lds = dim <= 12 * 1024 ? 1 : 0;
  const size_t fsm = in_lds ? (size_t)dim * sizeof(int32_t) + 16 : 0;      // row map in LDS
  const int64_t ss = (flags & BJX_COUPLING_SCALE_BCAST) ? 0 : n1, st = (flags & BJX_COUPLING_SHIFT_BCAST) ? 0 : n1;
  if (!inverse) { CouplingAffineF<T, false> f{map, scale, shift, n1, dim, ss, st, in_lds, 0.0, nullptr}; return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0); }
  CouplingAffineF<T, true> f{map, scale, shift, n1, dim, ss, st, in_lds, 0.0, nullptr};
  return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
}
template <class T>
int coupling_rqs_impl(bjx_ctx* ctx, int inverse, const int32_t* idx1, int64_t n1, const T* w, const T* h, const T* d, int K1,
                      const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  int32_t* map = nullptr;
  int rc = build_rowmap(ctx, idx1, n1, dim, &map);
  if (rc) return rc;
  const bool lds = knots_fit_lds<T>(n1, K1);
  const size_t fsm = lds ? (size_t)n1 * K1 * 3 * sizeof(T) : 0;
  if (!inverse) { CouplingRqsF<T, false> f{map, w, h, d, K1, n1, lds ? 1 : 0, 0.0, nullptr}; return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0); }
  CouplingRqsF<T, true> f{map, w, h, d, K1, n1, lds ? 1 : 0, 0.0, nullptr};
  return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
}
}  // namespace

namespace {
// ------------------------------------------------------------------ Coupling (affine law) pullback
// coupling.jl:206-259 with b = Shift(t) ∘ Scale(s), s, t = θ(x₂) per column:  y₁ = t + s x₁, logabsdetjac = Σ log|s|.
//   forward:  x̄₁ = s ȳ₁,        s̄ = ȳ₁ x₁ + ℓ̄/s,            t̄ = ȳ₁
//   inverse:  ȳ₁ = x̄₁/s,        s̄ = -(x̄₁/s) x₁ - ℓ̄/s,       t̄ = -x̄₁/s       (x₁ = (y₁ - t)/s)
// rows outside x₁ pass their cotangent through.  s̄, t̄ ([n1, batch]) are what the host needs to continue through θ
// (an arbitrary closure: its pullback stays with the AD package); the cotangent θ adds to x₂ is added there too.
// G lanes per column, 4 columns in flight, the row map in LDS.
template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void coupling_affine_vjp_kernel(const int32_t* __restrict__ map, const T* __restrict__ scale, const T* __restrict__ shift,
                                                                  const T* __restrict__ x, const T* __restrict__ gbar, const T* __restrict__ lbar,
                                                                  T* __restrict__ xbar, T* __restrict__ sbar, T* __restrict__ tbar, int64_t n1, int64_t dim,
                                                                  int64_t batch, int G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* m = reinterpret_cast<int32_t*>(smem);
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) m[i] = map[i];
  __syncthreads();
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = 256 / G;
  const int64_t nvc = dim / V;
  constexpr int UC = 4;
  const int64_t col0 = (int64_t)blockIdx.x * cols_per_block * UC + threadIdx.x / G;
  for (int64_t v = gl; v < nvc; v += G) {
    const int64_t row = v * V;
    Pack<T, V> px[UC], pg[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int64_t col = col0 + (int64_t)u * cols_per_block;
      if (col < batch) { px[u] = load_pack<T, V, true>(x + col * dim + row); pg[u] = load_pack<T, V, true>(gbar + col * dim + row); }
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int64_t col = col0 + (int64_t)u * cols_per_block;
      if (col >= batch) continue;
      const T lb = lbar ? lbar[col] : T(0);
      Pack<T, V> o;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int32_t mi = m[row + j];
        T out = pg[u].v[j];
        if (mi >= 0) {
          const T sc = scale ? scale[col * n1 + mi] : T(1);
          const T tv = shift ? shift[col * n1 + mi] : T(0);
          const T rs = T(1) / sc;
          if (!INV) {
            out = sc * pg[u].v[j];
            if (sbar) sbar[col * n1 + mi] = pg[u].v[j] * px[u].v[j] + lb * rs;
            if (tbar) tbar[col * n1 + mi] = pg[u].v[j];
          } else {
            const T x1 = (px[u].v[j] - tv) * rs;
            out = pg[u].v[j] * rs;
            if (sbar) sbar[col * n1 + mi] = -out * x1 - lb * rs;
            if (tbar) tbar[col * n1 + mi] = -out;
          }
        }
        o.v[j] = out;
      }
      store_pack<T, V, true>(xbar + col * dim + row, o);
    }
  }
}

template <class T>
int coupling_affine_vjp_impl(bjx_ctx* ctx, int inverse, const int32_t* idx1, int64_t n1, const T* scale, const T* shift, const T* in, const T* out_bar,
                             const T* ladj_bar, T* in_bar, T* scale_bar, T* shift_bar, int64_t dim, int64_t batch) {
  if (dim * batch == 0) return BJX_OK;
  int32_t* map = nullptr;
  int rc = build_rowmap(ctx, idx1, n1, dim, &map);
  if (rc) return rc;
  const size_t smem = (size_t)dim * sizeof(int32_t);
  BJX_REQUIRE(ctx, smem <= 60 * 1024, BJX_ERR_UNSUPPORTED, "bjx_coupling_affine_vjp: dim %lld too large for the LDS row map", (long long)dim);
  constexpr int VW = Vec16<T>::N;
  const bool v_ok = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar) && dim % VW == 0;
  const int V = v_ok ? VW : 1;
  const int64_t packs = dim / V;
  int G = 1;
  while (G < 64 && G < packs) G <<= 1;
  const int64_t cpb = (int64_t)(256 / G) * 4;
  const int64_t grid = (batch + cpb - 1) / cpb;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_coupling_affine_vjp: batch too large for one launch");
  {
    BjxProf prof_(ctx);
#define CAV(V_, I_) hipLaunchKernelGGL((coupling_affine_vjp_kernel<T, V_, I_>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, map, scale, shift, in, out_bar, ladj_bar, in_bar, scale_bar, shift_bar, n1, dim, batch, G)
    if (v_ok) { if (inverse) CAV(VW, true); else CAV(VW, false); }
    else { if (inverse) CAV(1, true); else CAV(1, false); }
#undef CAV
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_coupling_affine_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const int32_t* idx1, int64_t n1, const void* scale, const void* shift,
                                    const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, void* scale_bar, void* shift_bar,
                                    int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0 && n1 >= 0 && n1 <= dim, BJX_ERR_SHAPE, "bjx_coupling_affine_vjp: bad size (n1=%lld, dim=%lld)", (long long)n1, (long long)dim);
  BJX_REQUIRE(ctx, (idx1 || n1 == 0) && ((in && out_bar && in_bar) || dim * batch == 0), BJX_ERR_ARG, "bjx_coupling_affine_vjp: null pointer");
  DISPATCH_DT(ctx, dt,
              coupling_affine_vjp_impl<float>(ctx, inverse, idx1, n1, (const float*)scale, (const float*)shift, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, (float*)scale_bar, (float*)shift_bar, dim, batch),
              coupling_affine_vjp_impl<double>(ctx, inverse, idx1, n1, (const double*)scale, (const double*)shift, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, (double*)scale_bar, (double*)shift_bar, dim, batch),
              "bjx_coupling_affine_vjp");
}

BJX_API int bjx_coupling_affine(bjx_ctx* ctx, bjx_dtype dt, int inverse, const int32_t* idx1, int64_t n1, const void* scale,
                                const void* shift, const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t dim,
                                int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0 && n1 >= 0 && n1 <= dim, BJX_ERR_SHAPE, "bjx_coupling_affine: bad size (n1=%lld, dim=%lld)", (long long)n1, (long long)dim);
  BJX_REQUIRE(ctx, (idx1 || n1 == 0) && ((in && out) || dim * batch == 0), BJX_ERR_ARG, "bjx_coupling_affine: null pointer");
  DISPATCH_DT(ctx, dt,
              coupling_affine_impl<float>(ctx, inverse, idx1, n1, (const float*)scale, (const float*)shift, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags),
              coupling_affine_impl<double>(ctx, inverse, idx1, n1, (const double*)scale, (const double*)shift, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags),
              "bjx_coupling_affine");
}

BJX_API int bjx_coupling_rqs(bjx_ctx* ctx, bjx_dtype dt, int inverse, const int32_t* idx1, int64_t n1, const void* widths,
                             const void* heights, const void* derivs, int n_knots, const void* in, void* out, void* ladj_ps,
                             double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0 && n1 >= 0 && n1 <= dim, BJX_ERR_SHAPE, "bjx_coupling_rqs: bad size");
  BJX_REQUIRE(ctx, n_knots >= 2, BJX_ERR_SHAPE, "bjx_coupling_rqs: need at least 2 knots");
  BJX_REQUIRE(ctx, (idx1 || n1 == 0) && widths && heights && derivs && ((in && out) || dim * batch == 0), BJX_ERR_ARG, "bjx_coupling_rqs: null pointer");
  DISPATCH_DT(ctx, dt,
              coupling_rqs_impl<float>(ctx, inverse, idx1, n1, (const float*)widths, (const float*)heights, (const float*)derivs, n_knots, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags),
              coupling_rqs_impl<double>(ctx, inverse, idx1, n1, (const double*)widths, (const double*)heights, (const double*)derivs, n_knots, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags),
              "bjx_coupling_rqs");
}
