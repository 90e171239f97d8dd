// bjx_chain.hip — F1: fused elementwise ComposedFunction chains.
//
// Replaces the reference's one-allocating-pass-per-stage evaluation of
//   with_logabsdet_jacobian(elementwise(exp) ∘ Shift(b) ∘ Scale(a), X)      (SURVEY.md §3.1;
//   src/bijectors/composed.jl:4-25, scale.jl:11-32, shift.jl:14-24, exp_log.jl:5-9,
//   logit.jl:15-30, leaky_relu.jl:25-29, truncated.jl:15-91)
// by ONE streaming pass: 16-byte loads, all stages applied in registers, log-det contributions
// accumulated per lane (f64), wave shuffle + LDS block reduce, one partial per block, fixed-order
// final reduce.  Algorithmic HBM traffic: read x once + write y once (8 B/elt f32).
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
  int off0, off1;      // element offsets into the LDS table (per-row params)
};
constexpr int CHAIN_U = 4;  // independent 16-B loads in flight per lane

template <class T> struct ChainArgs {
  DevOp<T> ops[BJX_MAX_OPS];
  int n_ops;
  int tab_elems;       // LDS table size in elements (multiple of 4)
};

// ROWMODE: 0 no per-row parameters; 1 rows of a pack are contiguous (dim % V == 0), table in LDS;
//          2 rows wrap inside a pack, table in LDS; 3 rows wrap, parameters read from global.
template <class T, int V, int ROWMODE>
__device__ __forceinline__ void load_params(const DevOp<T>& op, const T* tab, int64_t r, int64_t dim, T* a, T* b) {
  if (ROWMODE == 0 || op.plen <= 1) {
    T s0 = (op.plen == 1 && op.v0) ? op.v0[0] : op.s0;
    T s1 = (op.plen == 1 && op.v1) ? op.v1[0] : op.s1;
#pragma unroll
    for (int j = 0; j < V; ++j) { a[j] = s0; b[j] = s1; }
    return;
  }
  if constexpr (ROWMODE == 1) {
    if constexpr (V > 1) {
      Pack<T, V> pa = load_pack<T, V, false>(tab + op.off0 + r);
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] = pa.v[j];
      if (op.v1) {
        Pack<T, V> pb = load_pack<T, V, false>(tab + op.off1 + r);
#pragma unroll
        for (int j = 0; j < V; ++j) b[j] = pb.v[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) b[j] = op.s1;
      }
    } else {
      a[0] = tab[op.off0 + r];
      b[0] = op.v1 ? tab[op.off1 + r] : op.s1;
    }
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int64_t rj = r + j;
      while (rj >= dim) rj -= dim;
      if constexpr (ROWMODE == 2) {
        a[j] = tab[op.off0 + rj];
        b[j] = op.v1 ? tab[op.off1 + rj] : op.s1;
      } else {
        a[j] = op.v0[rj];
        b[j] = op.v1 ? op.v1[rj] : op.s1;
      }
    }
  }
}

// Applies every stage of the chain to U packs of V consecutive elements each.  The op loop is the
// OUTER loop: one wave-uniform `switch` per stage handles all U*V elements of the lane, so the
// interpreter's scalar overhead is amortised over 16-32 elements and every case is a straight run
// of independent VALU work.  Returns Σ of the data-dependent log-det contributions (Scale's
// parameter-only term is added by the caller / finalize).
#define BJX_FOR_UJ _Pragma("unroll") for (int u = 0; u < U; ++u) _Pragma("unroll") for (int j = 0; j < V; ++j)
template <class T, int V, int U, int ROWMODE>
__device__ __forceinline__ T apply_chain(const ChainArgs<T>& A, const T* tab, Pack<T, V> (&p)[U], const int64_t (&r)[U], int64_t dim) {
  T l = T(0);
  for (int k = 0; k < A.n_ops; ++k) {
    const DevOp<T>& op = A.ops[k];
    const int kind = op.kind;
    T a[U][V], b[U][V];
    if (kind != BJX_OP_EXP && kind != BJX_OP_LOG && kind != BJX_OP_SIGNFLIP) {
#pragma unroll
      for (int u = 0; u < U; ++u) load_params<T, V, ROWMODE>(op, tab, r[u], dim, a[u], b[u]);
    }
    switch (kind) {
      case BJX_OP_EXP:  // exp_log.jl:5-6: ladj = sum(x)
        BJX_FOR_UJ { l += p[u].v[j]; p[u].v[j] = d_exp(p[u].v[j]); }
        break;
      case BJX_OP_LOG:  // exp_log.jl:8-9: ladj = -sum(log, x)
        BJX_FOR_UJ { T t = d_log(p[u].v[j]); l -= t; p[u].v[j] = t; }
        break;
      case BJX_OP_SHIFT:  // shift.jl:14
        BJX_FOR_UJ p[u].v[j] = a[u][j] + p[u].v[j];
        break;
      case BJX_OP_SCALE:  // scale.jl:13
        BJX_FOR_UJ p[u].v[j] = a[u][j] * p[u].v[j];
        break;
      case BJX_OP_SCALE_INV:  // scale.jl:15-16: Scale(inv(a))
        BJX_FOR_UJ p[u].v[j] = (T(1) / a[u][j]) * p[u].v[j];
        break;
      case BJX_OP_LOGIT:  // logit.jl:15,24
        BJX_FOR_UJ {
          T x = p[u].v[j];
          l += -d_log((x - a[u][j]) * (b[u][j] - x) / (b[u][j] - a[u][j]));
          p[u].v[j] = d_logit((x - a[u][j]) / (b[u][j] - a[u][j]));
        }
        break;
      case BJX_OP_LOGIT_INV:  // logit.jl:19 ; interface.jl:276-281
        BJX_FOR_UJ {
          T x = (b[u][j] - a[u][j]) * d_logistic(p[u].v[j]) + a[u][j];
          l += d_log((x - a[u][j]) * (b[u][j] - x) / (b[u][j] - a[u][j]));
          p[u].v[j] = x;
        }
        break;
      case BJX_OP_LEAKY_RELU:  // leaky_relu.jl:25-29
        BJX_FOR_UJ {
          T J = p[u].v[j] < T(0) ? a[u][j] : T(1);
          l += d_log(d_abs(J));
          p[u].v[j] = J * p[u].v[j];
        }
        break;
      case BJX_OP_TRUNCATED:  // truncated.jl:15-31,51-67
        BJX_FOR_UJ {
          T lo = a[u][j], up = b[u][j];
          T x = d_clamp(p[u].v[j], lo, up);
          bool lb = d_isfinite(lo), ub = d_isfinite(up);
          if (lb && ub) { l += -d_log((x - lo) * (up - x) / (up - lo)); p[u].v[j] = d_logit((x - lo) / (up - lo)); }
          else if (lb) { T t = d_log(x - lo); l -= t; p[u].v[j] = t; }
          else if (ub) { T t = d_log(up - x); l -= t; p[u].v[j] = t; }
          else p[u].v[j] = x;
        }
        break;
      case BJX_OP_TRUNCATED_INV:  // truncated.jl:33-49,71-91
        BJX_FOR_UJ {
          T lo = a[u][j], up = b[u][j], yv = p[u].v[j];
          bool lb = d_isfinite(lo), ub = d_isfinite(up);
          T x;
          if (lb && ub) { T ay = d_abs(yv); l += d_log(up - lo) - ay - T(2) * d_log1pexp(-ay); x = (up - lo) * d_logistic(yv) + lo; }
          else if (lb) { l += yv; x = d_exp(yv) + lo; }
          else if (ub) { l += yv; x = up - d_exp(yv); }
          else x = yv;
          p[u].v[j] = d_clamp(x, lo, up);
        }
        break;
      case BJX_OP_SIGNFLIP:  // ordered.jl:3
        BJX_FOR_UJ p[u].v[j] = -p[u].v[j];
        break;
      default: break;
    }
  }
  return l;
}

template <class T, int ROWMODE>
__device__ __forceinline__ void stage_table(const ChainArgs<T>& A, T* tab) {
  if constexpr (ROWMODE == 1 || ROWMODE == 2) {
    for (int k = 0; k < A.n_ops; ++k) {
      const DevOp<T>& op = A.ops[k];
      if (op.plen > 1) {
        for (int i = threadIdx.x; i < op.plen; i += blockDim.x) {
          tab[op.off0 + i] = op.v0[i];
          if (op.v1) tab[op.off1 + i] = op.v1[i];
        }
      }
    }
    __syncthreads();
  }
}

template <class T, int V, int U, int ROWMODE, bool NT>
__global__ __launch_bounds__(256) void chain_flat_kernel(const ChainArgs<T> A, const T* x, T* y, int64_t n,
                                                         int64_t dim, int64_t row_step, double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tab = reinterpret_cast<T*>(smem);
  double* red = reinterpret_cast<double*>(smem + (size_t)A.tab_elems * sizeof(T));
  stage_table<T, ROWMODE>(A, tab);

  const int64_t nv = n / V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t r0 = 0;
  if constexpr (ROWMODE != 0) r0 = (i * V) % dim;
  double acc = 0.0;
  // full iterations: U independent 16-byte loads in flight per lane
  for (; i + (U - 1) * stride < nv; i += stride * U) {
    Pack<T, V> p[U];
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = load_pack<T, V, NT>(x + (i + u * stride) * V);
      r[u] = r0;
      if constexpr (ROWMODE != 0) { r0 += row_step; if (r0 >= dim) r0 -= dim; }
    }
    T l = apply_chain<T, V, U, ROWMODE>(A, tab, p, r, dim);
#pragma unroll
    for (int u = 0; u < U; ++u) store_pack<T, V, NT>(y + (i + u * stride) * V, p[u]);
    acc += (double)l;
  }
  // remainder packs, one at a time
  for (; i < nv; i += stride) {
    Pack<T, V> p[1];
    int64_t r[1] = {r0};
    p[0] = load_pack<T, V, NT>(x + i * V);
    T l = apply_chain<T, V, 1, ROWMODE>(A, tab, p, r, dim);
    store_pack<T, V, NT>(y + i * V, p[0]);
    acc += (double)l;
    if constexpr (ROWMODE != 0) { r0 += row_step; if (r0 >= dim) r0 -= dim; }
  }
  // tail elements (n % V) by one lane
  if (V > 1 && blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t e = nv * V; e < n; ++e) {
      Pack<T, 1> q[1];
      q[0].v[0] = x[e];
      int64_t r[1] = {ROWMODE == 0 ? 0 : e % dim};
      T l;
      if constexpr (ROWMODE == 0) l = apply_chain<T, 1, 1, 0>(A, tab, q, r, dim);
      else if constexpr (ROWMODE == 3) l = apply_chain<T, 1, 1, 3>(A, tab, q, r, dim);
      else l = apply_chain<T, 1, 1, 2>(A, tab, q, r, dim);
      y[e] = q[0].v[0];
      acc += (double)l;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// Per-sample variant: G consecutive lanes own one column (G*V elements per step, coalesced because
// a column is contiguous); 256/G columns per block step.  Writes ladj_ps[col] (+ per-sample
// constant) and the block partial of the sum.
template <class T, int V, int U, int ROWMODE, bool NT>
__global__ __launch_bounds__(256) void chain_colgroup_kernel(const ChainArgs<T> A, const T* x, T* y, T* ladj_ps,
                                                             int64_t dim, int64_t batch, int G, double c_ps_host,
                                                             const double* c_ps_dev, int accumulate,
                                                             double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tab = reinterpret_cast<T*>(smem);
  double* red = reinterpret_cast<double*>(smem + (size_t)A.tab_elems * sizeof(T));
  stage_table<T, ROWMODE>(A, tab);

  const int gl = threadIdx.x & (G - 1);            // lane within the column group
  const int cols_per_block = blockDim.x / G;
  const int64_t col_stride = (int64_t)gridDim.x * cols_per_block;
  const double c_ps = c_ps_host + (c_ps_dev ? *c_ps_dev : 0.0);
  const int64_t nvc = dim / V;                     // full packs per column (dim % V == 0 when V > 1)
  double acc = 0.0;
  for (int64_t col = (int64_t)blockIdx.x * cols_per_block + threadIdx.x / G; col < batch; col += col_stride) {
    const T* xc = x + col * dim;
    T* yc = y + col * dim;
    T l = T(0);
    int64_t v = gl;
    for (; v + (int64_t)(U - 1) * G < nvc; v += (int64_t)G * U) {
      Pack<T, V> p[U];
      int64_t r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { p[u] = load_pack<T, V, NT>(xc + (v + (int64_t)u * G) * V); r[u] = (v + (int64_t)u * G) * V; }
      l += apply_chain<T, V, U, ROWMODE>(A, tab, p, r, dim);
#pragma unroll
      for (int u = 0; u < U; ++u) store_pack<T, V, NT>(yc + (v + (int64_t)u * G) * V, p[u]);
    }
    for (; v < nvc; v += G) {
      Pack<T, V> p[1];
      int64_t r[1] = {v * V};
      p[0] = load_pack<T, V, NT>(xc + v * V);
      l += apply_chain<T, V, 1, ROWMODE>(A, tab, p, r, dim);
      store_pack<T, V, NT>(yc + v * V, p[0]);
    }
    l = group_sum_rt(l, G);
    if (gl == 0) {
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
  if (v < 0) { const char* e = getenv("BJX_NT"); v = e ? atoi(e) : 0; }
  return v != 0;
}

template <class T>
int chain_impl(bjx_ctx* ctx, const bjx_op* ops, int n_ops, const T* x, T* y, T* ladj_ps, double* ladj_sum,
               int64_t dim, int64_t batch, uint32_t flags) {
  ChainArgs<T> A;
  memset(&A, 0, sizeof(A));
  A.n_ops = n_ops;
  int tab = 0;
  bool any_row = false, any_dev_scale = false;
  double c_ps_host = 0.0, c_sum_host = 0.0;
  const bool quirk = (flags & BJX_REF_VECTOR_SCALE_LADJ) != 0;
  for (int k = 0; k < n_ops; ++k) {
    const bjx_op& o = ops[k];
    DevOp<T>& d = A.ops[k];
    BJX_REQUIRE(ctx, o.kind >= BJX_OP_EXP && o.kind <= BJX_OP_IDENTITY, BJX_ERR_ARG, "bjx_chain: op %d has unknown kind %d", k, o.kind);
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
      d.off0 = tab; tab += (d.plen + 3) & ~3;
      if (d.v1) { d.off1 = tab; tab += (d.plen + 3) & ~3; }
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
  const bool nt = env_nt();
  size_t tab_bytes = (size_t)tab * sizeof(T);
  const bool tab_lds = tab_bytes <= 60 * 1024;
  A.tab_elems = tab_lds ? tab : 0;
  size_t smem = (tab_lds ? tab_bytes : 0) + 4 * sizeof(double);
  double* partials = ladj_sum ? ctx->partials : nullptr;
  int grid = 1;

  static const int bpc = env_int("BJX_BPC", 8);   // resident 256-thread blocks per CU the grid is sized for
  static const int uu = env_int("BJX_U", CHAIN_U); // packs in flight per lane (tuning knob: 4 or 8)
#define LAUNCH_FLAT_U(V_, RM_, U_)                                                                            \
  do {                                                                                                        \
    int64_t need_ = (n / V_ + U_ * 256 - 1) / (U_ * 256);                                                     \
    int64_t cap_ = (int64_t)ctx->num_cu * bpc;                                                                \
    if (cap_ > BJX_MAX_BLOCKS) cap_ = BJX_MAX_BLOCKS;                                                         \
    grid = (int)(need_ < 1 ? 1 : (need_ < cap_ ? need_ : cap_));                                              \
    int64_t row_step = any_row ? (((int64_t)grid * 256 * V_) % dim) : 0;                                      \
    if (nt) hipLaunchKernelGGL((chain_flat_kernel<T, V_, U_, RM_, true>), dim3(grid), dim3(256), smem, ctx->stream, A, x, y, n, dim, row_step, partials); \
    else hipLaunchKernelGGL((chain_flat_kernel<T, V_, U_, RM_, false>), dim3(grid), dim3(256), smem, ctx->stream, A, x, y, n, dim, row_step, partials);  \
  } while (0)
#define LAUNCH_FLAT(V_, RM_) LAUNCH_FLAT_U(V_, RM_, CHAIN_U)
#define LAUNCH_FLAT_TUNED(V_, RM_) do { if (uu == 8) LAUNCH_FLAT_U(V_, RM_, 8); else if (uu == 2) LAUNCH_FLAT_U(V_, RM_, 2); else LAUNCH_FLAT_U(V_, RM_, CHAIN_U); } while (0)

  if (!ladj_ps) {
    if (!any_row) { if (vec_ok) LAUNCH_FLAT_TUNED(VW, 0); else LAUNCH_FLAT(1, 0); }
    else if (!tab_lds) { if (vec_ok) LAUNCH_FLAT(VW, 3); else LAUNCH_FLAT(1, 3); }
    else if (vec_ok && dim % VW == 0) LAUNCH_FLAT_TUNED(VW, 1);
    else if (vec_ok) LAUNCH_FLAT(VW, 2);
    else LAUNCH_FLAT(1, 2);
  } else {
    // per-sample: G lanes per column
    const bool v_ok = vec_ok && dim % VW == 0;
    const int64_t packs = v_ok ? dim / VW : dim;
    int G = 1;
    while (G < 64 && G < packs) G <<= 1;
    const int cols_per_block = 256 / G;
    grid = bjx_stream_grid(ctx, batch, cols_per_block);
    const double* cdev = any_dev_scale ? ctx->consts : nullptr;
    const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
#define LAUNCH_COL(V_, RM_)                                                                                                 \
  do {                                                                                                                      \
    if (nt) hipLaunchKernelGGL((chain_colgroup_kernel<T, V_, CHAIN_U, RM_, true>), dim3(grid), dim3(256), smem, ctx->stream, A, x, y, ladj_ps, dim, batch, G, c_ps_host, cdev, accum, partials);  \
    else hipLaunchKernelGGL((chain_colgroup_kernel<T, V_, CHAIN_U, RM_, false>), dim3(grid), dim3(256), smem, ctx->stream, A, x, y, ladj_ps, dim, batch, G, c_ps_host, cdev, accum, partials);   \
  } while (0)
    if (!any_row) { if (v_ok) LAUNCH_COL(VW, 0); else LAUNCH_COL(1, 0); }
    else if (!tab_lds) { if (v_ok) LAUNCH_COL(VW, 3); else LAUNCH_COL(1, 3); }
    else if (v_ok) LAUNCH_COL(VW, 1);
    else LAUNCH_COL(1, 2);
#undef LAUNCH_COL
  }
#undef LAUNCH_FLAT
#undef LAUNCH_FLAT_U
#undef LAUNCH_FLAT_TUNED
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, grid, ladj_sum, c_sum_host, any_dev_scale ? 1 : 0, 0.0, flags);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_chain(bjx_ctx* ctx, bjx_dtype dt, const bjx_op* ops, int n_ops, const void* x, void* y,
                      void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, n_ops >= 0 && n_ops <= BJX_MAX_OPS && (ops || n_ops == 0), BJX_ERR_ARG, "bjx_chain: n_ops must be in [0, %d]", BJX_MAX_OPS);
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0, BJX_ERR_SHAPE, "bjx_chain: negative size");
  BJX_REQUIRE(ctx, (x && y) || dim * batch == 0, BJX_ERR_ARG, "bjx_chain: null data pointer");
  if (dt == BJX_F32) return chain_impl<float>(ctx, ops, n_ops, (const float*)x, (float*)y, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return chain_impl<double>(ctx, ops, n_ops, (const double*)x, (double*)y, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_chain: bad dtype %d", (int)dt);
}
