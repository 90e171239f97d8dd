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
  if ((x <= -wK) || (x >= wK)) { y = x; lj = T(0); return; }   // :324-326
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

// ------------------------------------------------------------------ RQS, table kernel
// The functor path above costs 178 (fwd) / 295 (inv) VALU instructions per element and suffers
// 8-way LDS bank conflicts on the [rows, K] column-major knot tables (PMC: SQ_LDS_BANK_CONFLICT =
// 63 % of LDS cycles), i.e. it is VALU/LDS-bound at 13 % of the HBM roofline.  This kernel
//  * precomputes, once per call, a per-(row, bin) record {w_k, 1/w, h_k, Δy | s, d_k, d_k+1, w}
//    (two 16-byte LDS reads replace six 4-byte reads and two divisions per element),
//  * stores search keys and records row-major ([row][bin]) so that lanes that share a row but fall
//    into different bins hit different banks,
//  * uses one hardware log + one reciprocal per element in Float32 (log(num/den²)),
//  * reuses the inverse's ξ for its log-det instead of a second search + forward evaluation, and
//  * lets one block walk ITER column groups so the 20 KiB table staging is amortised.
// Bin selection is exact (same Float32/Float64 knot values and comparisons as the oracle).
template <class T> struct RqsRec { T a[4]; T b[4]; };   // a = {w_k, 1/w, h_k, dy}, b = {s, d_k, d_k1, w}

// rational_quadratic_spline.jl:139-156: bin k in 0..K-1 spans knots k..k+1 (knot 0 = -knot K mirrored)
template <class T>
__global__ __launch_bounds__(256) void rqs_prep_kernel(const T* w, const T* h, const T* d, int K, int64_t rows,
                                                       T* keyW, T* keyH, RqsRec<T>* rec) {
  const int64_t n = rows * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / K;
    const int k = (int)(i % K);
    auto W = [&](int j) { return w[(int64_t)(j - 1) * rows + r]; };   // 1-based knot j of row r
    auto H = [&](int j) { return h[(int64_t)(j - 1) * rows + r]; };
    auto D = [&](int j) { return d[(int64_t)(j - 1) * rows + r]; };
    keyW[i] = W(k + 1);
    keyH[i] = H(k + 1);
    const T w_k = (k == 0) ? -W(K) : W(k);
    const T wd = W(k + 1) - w_k;
    const T h_k = (k == 0) ? -H(K) : H(k);
    const T dy = H(k + 1) - h_k;
    RqsRec<T> q;
    q.a[0] = w_k; q.a[1] = T(1) / wd; q.a[2] = h_k; q.a[3] = dy;
    q.b[0] = dy / wd;                                  // s
    q.b[1] = (k == 0) ? T(1) : D(k);                   // d_k
    q.b[2] = (k == K - 1) ? T(1) : D(k + 1);           // d_{k+1}
    q.b[3] = wd;
    rec[i] = q;
  }
}

// number of keys (ascending, length K) strictly below x == searchsortedfirst(keys, x) - 1
template <class T> __device__ __forceinline__ int count_below(const T* keys, int K, int top, T x) {
  int pos = 0;
  for (int step = top; step >= 1; step >>= 1) {
    const int nx = pos + step;
    if (nx <= K && keys[nx - 1] < x) pos = nx;
  }
  return pos;
}

template <class T, bool INV>
__device__ __forceinline__ T rqs_table_elem(const T* keys, const RqsRec<T>* rec, int K, int top, T& v) {
  using F = Fast<T>;
  const T lim = keys[K - 1];
  const T x = v;
  if ((x <= -lim) || (x >= lim)) return T(0);        // identity outside [-B, B], log-det 0 (:132, :186)
  const int k = count_below<T>(keys, K, top, x);
  const RqsRec<T> q = rec[k];
  const T s = q.b[0], d_k = q.b[1], d_k1 = q.b[2];
  T xi;
  if (!INV) {
    xi = (x - q.a[0]) * q.a[1];                                             // ξ
  } else {
    const T yh = x - q.a[2];
    const T ds = d_k1 + d_k - 2 * s;
    const T a1 = q.a[3] * (s - d_k) + yh * ds;                              // Eq. (25)
    const T a2 = q.a[3] * d_k - yh * ds;                                    // Eq. (26)
    const T a3 = -s * yh;                                                   // Eq. (27)
    xi = F::div(-2 * a3, a2 + F::sqrt(a2 * a2 - 4 * a1 * a3));              // Eq. (24)
  }
  const T om = T(1) - xi;
  const T xo = xi * om;
  const T den = s + (d_k1 + d_k - 2 * s) * xo;
  const T rden = F::rcp(den);
  const T num_jl = s * s * (d_k1 * (xi * xi) + 2 * s * xo + d_k * (om * om));
  const T lj = F::log(num_jl * rden * rden);                                // log(num) - 2 log(den)
  if (!INV) { v = q.a[2] + q.a[3] * (s * (xi * xi) + d_k * xo) * rden; return lj; }
  v = xi * q.b[3] + q.a[0];
  return -lj;                                                               // interface.jl:276-281
}

template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void rqs_table_kernel(const T* keys_g, const RqsRec<T>* rec_g, int K, int top, int in_lds,
                                                        const T* x, T* y, T* ladj_ps, int64_t dim, int64_t batch, int G,
                                                        int iters, int accumulate, double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[4];
  const int64_t nrec = dim * K;
  RqsRec<T>* rec_l = reinterpret_cast<RqsRec<T>*>(smem);
  T* keys_l = reinterpret_cast<T*>(smem + (size_t)nrec * sizeof(RqsRec<T>));
  if (in_lds) {
    const int npk = (int)(nrec * (sizeof(RqsRec<T>) / 16));
    const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(rec_g);
    bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(rec_l);
    for (int i = threadIdx.x; i < npk; i += 256) dst[i] = src[i];
    for (int i = threadIdx.x; i < nrec; i += 256) keys_l[i] = keys_g[i];
    __syncthreads();
  }
  const RqsRec<T>* rec = in_lds ? rec_l : rec_g;
  const T* keys = in_lds ? keys_l : keys_g;
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = 256 / G;
  const int64_t nvc = dim / V;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    const int64_t col = ((int64_t)blockIdx.x * iters + it) * cols_per_block + threadIdx.x / G;
    T l = T(0);
    if (col < batch) {
      const T* xc = x + col * dim;
      T* yc = y + col * dim;
      for (int64_t v = gl; v < nvc; v += G) {
        Pack<T, V> p = load_pack<T, V, true>(xc + v * V);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const int64_t r = v * V + j;
          l += rqs_table_elem<T, INV>(keys + r * K, rec + r * K, K, top, p.v[j]);
        }
        store_pack<T, V, true>(yc + v * V, p);
      }
    }
    l = group_sum_rt(l, G);
    if (col < batch && gl == 0) {
      if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + l : l;
      acc += (double)l;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// ------------------------------------------------------------------ BatchNorm (eval)
// normalise.jl:41-88.  LDS rows: s = exp(logs), m, q = sqrt(v + eps), b.
template <class T, bool INV> struct BnF {
  static constexpr bool kLoadInput = true;
  const T *b, *logs, *m, *v;
  T eps;
  int64_t dim;
  int in_lds;
  double per_sample_const;
  const double* per_sample_dev;
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
  const int32_t* map;
  const T *scale, *shift;   // [n1, batch] or null
  int64_t n1;
  double per_sample_const;
  const double* per_sample_dev;
  __device__ void stage(char*) const {}
  template <int V> __device__ T apply(const char*, Pack<T, V>& p, const T*, int64_t row, int64_t col) const {
    T l = T(0);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int32_t mi = map[row + j];
      if (mi >= 0) {
        T s = scale ? scale[col * n1 + mi] : T(1);
        T t = shift ? shift[col * n1 + mi] : T(0);
        if (!INV) { p.v[j] = t + s * p.v[j]; l += d_log(d_abs(s)); }            // Shift(t) ∘ Scale(s)
        else { p.v[j] = (T(1) / s) * (-t + p.v[j]); l -= d_log(d_abs(s)); }     // inverse(Scale) ∘ inverse(Shift)
      }
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
  __device__ void stage(char*) const {}
  template <int V> __device__ T apply(const char*, Pack<T, V>& p, const T* xcol, int64_t row, int64_t) const {
#pragma unroll
    for (int j = 0; j < V; ++j) p.v[j] = xcol[src[row + j]];
    return T(0);
  }
};

template <class T> bool knots_fit_lds(int64_t rows, int K1) { return (size_t)rows * K1 * 3 * sizeof(T) <= 60 * 1024; }

template <class T>
int rqs_impl(bjx_ctx* ctx, int inverse, const T* w, const T* h, const T* d, int K1, const T* in, T* out, T* ladj_ps,
             double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  const size_t nrec = (size_t)dim * K1;
  const size_t tab_bytes = nrec * (sizeof(RqsRec<T>) + 2 * sizeof(T));
  if (tab_bytes + 64 > BJX_SCRATCH_BYTES) {   // huge knot tables: generic functor path, tables from global memory
    const bool lds = knots_fit_lds<T>(dim, K1);
    const size_t fsm = lds ? (size_t)dim * K1 * 3 * sizeof(T) : 0;
    if (!inverse) { RqsF<T, false> f{w, h, d, K1, dim, lds ? 1 : 0, 0.0, nullptr}; return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0); }
    RqsF<T, true> f{w, h, d, K1, dim, lds ? 1 : 0, 0.0, nullptr};
    return launch_colgroup<T>(ctx, f, fsm, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
  }
  if (dim * batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  RqsRec<T>* rec = static_cast<RqsRec<T>*>(ctx->scratch);
  T* keyW = reinterpret_cast<T*>(rec + nrec);
  T* keyH = keyW + nrec;
  int pg = (int)((nrec + 255) / 256);
  if (pg > 256) pg = 256;
  hipLaunchKernelGGL(rqs_prep_kernel<T>, dim3(pg), dim3(256), 0, ctx->stream, w, h, d, K1, dim, keyW, keyH, rec);
  BJX_CHECK_LAUNCH(ctx);
  ColLaunch c = col_launch_cfg<T>(ctx, in, out, dim, batch);
  const size_t lds_bytes = nrec * (sizeof(RqsRec<T>) + sizeof(T));
  const int in_lds = lds_bytes <= 60 * 1024 ? 1 : 0;
  const int cols_per_block = 256 / c.G;
  // amortise the table staging: each block walks `iters` column groups (>= ~64 KiB of data)
  int iters = 1;
  if (in_lds) {
    const int64_t bytes_per_group = (int64_t)cols_per_block * dim * sizeof(T);
    iters = (int)((4 * (int64_t)lds_bytes + bytes_per_group - 1) / bytes_per_group);
    if (iters < 1) iters = 1;
    if (iters > 64) iters = 64;
  }
  const int64_t groups = (batch + cols_per_block - 1) / cols_per_block;
  const int64_t grid = (groups + iters - 1) / iters;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_rqs: batch too large for one launch");
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  int top = 1;
  while (top * 2 <= K1) top *= 2;
  const T* keys = inverse ? keyH : keyW;
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  const size_t smem = in_lds ? lds_bytes : 0;
  constexpr int VW = Vec16<T>::N;
#define LAUNCH_RQS(V_, INV_) hipLaunchKernelGGL((rqs_table_kernel<T, V_, INV_>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, keys, rec, K1, top, in_lds, in, out, ladj_ps, dim, batch, c.G, iters, accum, partials)
  if (c.V == VW) { if (inverse) LAUNCH_RQS(VW, true); else LAUNCH_RQS(VW, false); }
  else { if (inverse) LAUNCH_RQS(1, true); else LAUNCH_RQS(1, false); }
#undef LAUNCH_RQS
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
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

BJX_API int bjx_permute(bjx_ctx* ctx, bjx_dtype dt, const int32_t* src, const void* in, void* out, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0, BJX_ERR_SHAPE, "bjx_permute: negative size");
  BJX_REQUIRE(ctx, (src && in && out) || dim * batch == 0, BJX_ERR_ARG, "bjx_permute: null pointer");
  BJX_REQUIRE(ctx, in != out || dim * batch == 0, BJX_ERR_ARG, "bjx_permute: in-place permutation is not supported");
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
  if (!inverse) { CouplingAffineF<T, false> f{map, scale, shift, n1, 0.0, nullptr}; return launch_colgroup<T>(ctx, f, 0, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0); }
  CouplingAffineF<T, true> f{map, scale, shift, n1, 0.0, nullptr};
  return launch_colgroup<T>(ctx, f, 0, in, out, ladj_ps, ladj_sum, dim, batch, flags, 0.0);
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
