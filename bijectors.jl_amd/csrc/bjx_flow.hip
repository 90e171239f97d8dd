// bjx_flow.hip — F2: per-sample reduce + broadcast flow layers.
//   PlanarLayer  planar_layer.jl:65-127,160-185   (n_layers fused in one pass)
//   RadialLayer  radial_layer.jl:43-129
//
// Mapping: G consecutive lanes own one column and keep it in registers (R 16-byte packs per lane),
// so every element is read from HBM once and written once even for an 8-layer stack: the
// reference's GEMV pass + rank-1 update pass + sech/log1p pass per layer become
// in-register dot products reduced with wave shuffles inside the G-lane group.
#include "bjx_internal.h"
#include "bjx_tile.h"

namespace {
using namespace bjx;

// planar_layer.jl:65-70: û = u + ((log1pexp(-wᵀu) - 1)/‖w‖²) w ;  wᵀû = log1pexp(wᵀu) - 1.
// One block per layer; writes û[l,:] and wᵀû[l] to scratch.
template <class T>
__global__ __launch_bounds__(256) void planar_prep_kernel(const T* w, const T* u, int64_t dim, T* u_hat, T* wtu_hat) {
  __shared__ double red[8];
  const int l = blockIdx.x;
  const T* wl = w + (int64_t)l * dim;
  const T* ul = u + (int64_t)l * dim;
  double dot = 0.0, w2 = 0.0;   // accumulated in f64, rounded to T once (reference: T dot / sum)
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) { dot += (double)wl[i] * (double)ul[i]; w2 += (double)wl[i] * (double)wl[i]; }
  dot = group_sum<64>(dot);
  w2 = group_sum<64>(w2);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = dot; red[4 + (threadIdx.x >> 6)] = w2; }
  __syncthreads();
  T wT_u = (T)((red[0] + red[1]) + (red[2] + red[3]));
  T ww = (T)((red[4] + red[5]) + (red[6] + red[7]));
  T c = (d_log1pexp(-wT_u) - T(1)) / ww;
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) u_hat[(int64_t)l * dim + i] = ul[i] + c * wl[i];
  if (threadIdx.x == 0) wtu_hat[l] = d_log1pexp(wT_u) - T(1);
}

#include "bjx_flow_common.inc"

template <class T, int V, int R, bool INV>
__global__ __launch_bounds__(256) void planar_kernel(const PlanarArgs<T> A, const T* x, T* y, T* ladj_ps, int64_t dim,
                                                     int64_t batch, int G, int accumulate, double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  T* tab = reinterpret_cast<T*>(smem + 32);
  const int64_t nld = (int64_t)A.n_layers * dim;
  if (A.in_lds) {
    for (int64_t i = threadIdx.x; i < nld; i += blockDim.x) { tab[i] = A.w[i]; tab[nld + i] = A.u_hat[i]; }
    __syncthreads();
  }
  const T* W = A.in_lds ? tab : A.w;
  const T* UH = A.in_lds ? tab + nld : A.u_hat;

  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = blockDim.x / G;
  const int64_t nvc = (dim + V - 1) / V;         // the last pack may be partial (odd heights: element-aligned packs, load_pack_part)
  double acc = 0.0;
  // non-persistent grid: one column per G-lane group.  Lanes of a group past the batch keep
  // running (on column batch-1, results discarded) so the group shuffles stay convergent.
  const int64_t col_raw = (int64_t)blockIdx.x * cols_per_block + threadIdx.x / G;
  const bool col_ok = col_raw < batch;
  {
    const int64_t col = col_ok ? col_raw : batch - 1;
    const T* xc = x + col * dim;
    T* yc = y + col * dim;
    Pack<T, V> z[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t v = gl + (int64_t)r * G;
      if (v < nvc) z[r] = load_pack_part<T, V>(xc + v * V, (int)(dim - v * V < V ? dim - v * V : V));
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) z[r].v[j] = T(0);
      }
    }
    T ladj = T(0);
    for (int li = 0; li < A.n_layers; ++li) {
      const int l = INV ? A.n_layers - 1 - li : li;
      const T* wl = W + (int64_t)l * dim;
      const T* ul = UH + (int64_t)l * dim;
      T s = T(0);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t v = gl + (int64_t)r * G;
        if (v < nvc) {
#pragma unroll
          for (int j = 0; j < V; ++j) s += (v * V + j < dim ? wl[v * V + j] : T(0)) * z[r].v[j];
        }
      }
      s = group_sum_rt(s, G);                 // wᵀz   (src/utils.jl:2)
      const T bl = A.b[l], c = A.wtu_hat[l];
      T t, s2;
      if (!INV) flow_tanh_sech2(s + bl, t, s2);
      else planar_inv_act<T>(s, c, bl, t, s2);
      const T ld = Fast<T>::log1p(c * s2);      // planar_layer.jl:107
      ladj += INV ? -ld : ld;
      const T tt = INV ? -t : t;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t v = gl + (int64_t)r * G;
        if (v < nvc) {
#pragma unroll
          for (int j = 0; j < V; ++j) z[r].v[j] += (v * V + j < dim ? ul[v * V + j] : T(0)) * tt;   // z ± û tanh(·)
        }
      }
    }
    if (accumulate & 2) {                       // BJX_BASE_STDNORMAL: + log N(out; 0, I)
      T q = T(0);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t v = gl + (int64_t)r * G;
        if (v < nvc) {
#pragma unroll
          for (int j = 0; j < V; ++j) q += z[r].v[j] * z[r].v[j];
        }
      }
      q = group_sum_rt(q, G);
      ladj += T(-0.5) * q - (T)dim * T(0.91893853320467274178);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t v = gl + (int64_t)r * G;
      if (col_ok && y && v < nvc) store_pack_part<T, V>(yc + v * V, z[r], (int)(dim - v * V < V ? dim - v * V : V));
    }
    if (col_ok && gl == 0) {
      if (ladj_ps) ladj_ps[col] = (accumulate & 1) ? ladj_ps[col] + ladj : ladj;
      acc += (double)ladj;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// ------------------------------------------------------------------ Planar input pullback, lanes kernel (any dtype / shape)
// SURVEY.md §8(f) f-1.  z̄ = (∂y/∂z)ᵀ ȳ + ℓ̄ ∂logabsdetjac/∂z for the fused stack of planar_layer.jl:65-110:
//   forward sweep:  s_k = w_kᵀz_{k-1} + b_k, t_k = tanh s_k (kept in LDS, [column][layer]), z_k = z_{k-1} + û_k t_k
//   reverse sweep:  s̄_k = (û_kᵀz̄_k)(1 - t_k²) + ℓ̄ c_k(-2 t_k)(1 - t_k²)/(1 + c_k(1 - t_k²)),  z̄_{k-1} = z̄_k + w_k s̄_k
// Same mapping as planar_kernel: G lanes own a column in registers; the primal column is dead once the t_k are
// known, so ȳ is loaded into the same registers.
template <class T, int V, int R, bool INV>
__global__ __launch_bounds__(256) void planar_vjp_kernel(const PlanarArgs<T> A, const T* __restrict__ x, const T* __restrict__ ybar,
                                                         const T* __restrict__ lbar, T* __restrict__ xbar, int64_t dim, int64_t batch, int G,
                                                         T* __restrict__ t_out, T* __restrict__ s_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cols_per_block = blockDim.x / G;
  T* tsave = reinterpret_cast<T*>(smem);                                   // [cols_per_block][n_layers]
  T* tab = tsave + (size_t)cols_per_block * A.n_layers;
  const int64_t nld = (int64_t)A.n_layers * dim;
  if (A.in_lds) {
    for (int64_t i = threadIdx.x; i < nld; i += blockDim.x) { tab[i] = A.w[i]; tab[nld + i] = A.u_hat[i]; }
    __syncthreads();
  }
  const T* W = A.in_lds ? tab : A.w;
  const T* UH = A.in_lds ? tab + nld : A.u_hat;
  const int gl = threadIdx.x & (G - 1), cl = threadIdx.x / G;
  const int64_t nvc = (dim + V - 1) / V;                 // the last pack may be partial (odd heights: element-aligned packs, load_pack_part)
  const int64_t col_raw = (int64_t)blockIdx.x * cols_per_block + cl;
  const bool col_ok = col_raw < batch;
  const int64_t col = col_ok ? col_raw : batch - 1;
  Pack<T, V> z[R];
  auto load_col = [&](const T* base) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) z[r] = load_pack_part<T, V>(base + col * dim + v * V, (int)(dim - v * V < V ? dim - v * V : V));
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) z[r].v[j] = T(0);
      }
    }
  };
  auto dot = [&](const T* row) -> T {
    T s = T(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
#pragma unroll
        for (int j = 0; j < V; ++j) s += (v * V + j < dim ? row[v * V + j] : T(0)) * z[r].v[j];
      }
    }
    return group_sum_rt(s, G);
  };
  auto axpy = [&](const T* row, T a) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
#pragma unroll
        for (int j = 0; j < V; ++j) z[r].v[j] += (v * V + j < dim ? row[v * V + j] : T(0)) * a;
      }
    }
  };
  load_col(x);
  T* tmine = tsave + (size_t)cl * A.n_layers;
  if (!INV) {
    for (int l = 0; l < A.n_layers; ++l) {
      const T t = flow_tanh(dot(W + (int64_t)l * dim) + A.b[l]);
      if (gl == 0) tmine[l] = t;
      axpy(UH + (int64_t)l * dim, t);
    }
  } else {
    // the inverse primal (planar_layer.jl:112-127): last layer first, t_l = tanh(α_l + b_l)
    for (int l = A.n_layers - 1; l >= 0; --l) {
      T t, s2u;
      planar_inv_act<T>(dot(W + (int64_t)l * dim), A.wtu_hat[l], A.b[l], t, s2u);
      if (gl == 0) tmine[l] = t;
      axpy(UH + (int64_t)l * dim, -t);
    }
  }
  __syncthreads();
  load_col(ybar);
  const T lb = lbar ? lbar[col] : T(0);
  if (!INV) {
    for (int l = A.n_layers - 1; l >= 0; --l) {
      const T t = tmine[l], c = A.wtu_hat[l];
      const T q = T(1) - t * t;
      const T sb = dot(UH + (int64_t)l * dim) * q + lb * c * (T(-2) * t) * q / (T(1) + c * q);
      if (s_out && col_ok && gl == 0) { s_out[col * A.n_layers + l] = sb; t_out[col * A.n_layers + l] = t; }   // for the parameter pullback
      axpy(W + (int64_t)l * dim, sb);
    }
  } else {
    // find_alpha's implicit-function rule (ext/BijectorsChainRulesCoreExt.jl:42-46): dα/d(wᵀy) = 1/(1 + c q)
    for (int l = 0; l < A.n_layers; ++l) {
      const T t = tmine[l], c = A.wtu_hat[l];
      const T q = T(1) - t * t;
      const T den = T(1) + c * q;
      const T sb = q / den * (-dot(UH + (int64_t)l * dim) + lb * T(2) * c * t / den);
      axpy(W + (int64_t)l * dim, sb);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t v = gl + (int64_t)r * G;
    if (col_ok && v < nvc) store_pack_part<T, V>(xbar + col * dim + v * V, z[r], (int)(dim - v * V < V ? dim - v * V : V));
  }
}

// ------------------------------------------------------------------ Planar, tile kernel
// One WAVE owns 64 columns.  The lanes-along-dim kernel above evaluates tanh/cosh/log1p for only
// 2-4 distinct samples per wave instruction, which makes an 8-layer stack ALU-bound at 8 % of the
// HBM roofline (profiles/r01_bench_lines_first_pass.jsonl).  Here the 64 x dim tile is staged
// through LDS (coalesced 16-byte global accesses; XOR-swizzled so both the pack-wise stores and the
// lane-per-column reads are bank-conflict free) and then LANE = COLUMN: all 64 lanes run the
// per-sample scalar recurrence on different samples.
//
// Algebra: with t_k = tanh(a_k + b_k), a_k = w_kᵀ z_{k-1} and z_k = z_{k-1} + û_k t_k,
//   a_k = w_kᵀ z_0 + Σ_{j<k} (w_kᵀ û_j) t_j ,   z_K = z_0 + Σ_k û_k t_k
// so the K dot products against z_0 are independent (one pass over the tile), the layer-to-layer
// dependency is an K-step scalar recurrence with the K x K table G[k][j] = w_kᵀ û_j, and the update
// is one more pass.  This re-associates the reference's sums (planar_layer.jl:73-80); differences
// are O(eps·‖w‖‖z‖), inside the 1e-3 / 1e-6 parity bars (tests/test_gpu_parity.py::test_planar).
// Layers are processed in groups of NLMAX; the tile in LDS is updated between groups.
constexpr int PLANAR_NLMAX = 8;
constexpr int PLANAR_REG_DEFAULT_COLS = 64;
constexpr int PLANAR_MFMA_DEFAULT = 0;       // BJX_PLANAR_MFMA: see planar_mfma_kernel (A/B numbers in DESIGN.md)

template <class T>
__global__ __launch_bounds__(256) void planar_prep2_kernel(const T* w, const T* u_hat, int64_t dim, int nl, T* G, T* wT, T* uT) {
  // block (k, j): G[k*nl + j] = w_kᵀ û_j ; block row 0 also writes the [row][layer] transposes
  __shared__ double red[4];
  const int k = blockIdx.x / nl, j = blockIdx.x % nl;
  double dot = 0.0;
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) dot += (double)w[(int64_t)k * dim + i] * (double)u_hat[(int64_t)j * dim + i];
  dot = group_sum<64>(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) G[k * nl + j] = (T)((red[0] + red[1]) + (red[2] + red[3]));
  if (j == 0) {
    for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) { wT[i * nl + k] = w[(int64_t)k * dim + i]; uT[i * nl + k] = u_hat[(int64_t)k * dim + i]; }
  }
}

template <class T> struct PlanarTileArgs {
  const T *wT, *uT;      // [dim][nl]
  const T *G;            // [nl][nl]
  const T *wtu_hat, *b;  // [nl]
  int nl;
};

__device__ __forceinline__ int tile_addr(int row, int c) { return row * 64 + (c ^ ((row >> 2) & 31)); }

// One group of ng <= NLMAX layers applied to this lane's column of the LDS tile; returns the
// group's log-det contribution.  FULL = (ng == NLMAX): no per-layer guards in the hot loops.
template <class T, bool INV, bool FULL>
__device__ __forceinline__ T planar_tile_group(const PlanarTileArgs<T>& A, T* tile, int dim, int lane, int l0, int ng) {
  constexpr int NL = PLANAR_NLMAX;
  // ---- dot products of the column against the group's w rows (wave-uniform -> scalar loads)
  T s[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) s[l] = T(0);
#pragma unroll 4
  for (int r = 0; r < dim; ++r) {
    const T z = tile[tile_addr(r, lane)];
    const T* wr = A.wT + (int64_t)r * A.nl + l0;
#pragma unroll
    for (int l = 0; l < NL; ++l) if (FULL || l < ng) s[l] += wr[l] * z;
  }
  // ---- scalar recurrence, one sample per lane
  T t[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) t[l] = T(0);
  T ladj = T(0);
#pragma unroll
  for (int kk = 0; kk < NL; ++kk) {
    const int k = INV ? (NL - 1 - kk) : kk;     // the inverse undoes the group's layers last-to-first
    if (FULL || k < ng) {
      const T* Gk = A.G + (int64_t)(l0 + k) * A.nl + l0;
      T a = s[k];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        if (!INV) { if (j < k) a += Gk[j] * t[j]; }
        else { if (j > k && (FULL || j < ng)) a -= Gk[j] * t[j]; }
      }
      const T bl = A.b[l0 + k], c = A.wtu_hat[l0 + k];
      T th, s2;
      if (INV) planar_inv_act<T>(a, c, bl, th, s2);
      else flow_tanh_sech2(a + bl, th, s2);
      const T ld = Fast<T>::log1p(c * s2);                // planar_layer.jl:107
      ladj += INV ? -ld : ld;
      t[k] = th;
    }
  }
  // ---- rank-ng update of the lane's column in LDS
#pragma unroll 4
  for (int r = 0; r < dim; ++r) {
    const T* ur = A.uT + (int64_t)r * A.nl + l0;
    T d = T(0);
#pragma unroll
    for (int l = 0; l < NL; ++l) if (FULL || l < ng) d += ur[l] * t[l];
    const int ad = tile_addr(r, lane);
    tile[ad] = INV ? tile[ad] - d : tile[ad] + d;
  }
  return ladj;
}

template <class T, int V, bool INV>
__global__ __launch_bounds__(64) void planar_tile_kernel(const PlanarTileArgs<T> A, const T* x, T* y, T* ladj_ps, int dim,
                                                        int64_t batch, int accumulate, double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);
  const int lane = threadIdx.x;
  const int64_t col0 = (int64_t)blockIdx.x * 64;
  const int ncols = (int)((batch - col0) < 64 ? (batch - col0) : 64);
  const int nelem = ncols * dim;
  const T* xt = x + col0 * dim;
  T* yt = y + col0 * dim;
  const int npk = (64 * dim) / V;
  // (column, row) of this lane's first pack; every step advances by 64 packs = 64*V elements
  const int c_first = (lane * V) / dim, r_first = (lane * V) % dim;
  // ---- stage in: coalesced packs -> swizzled [row][col] tile
  {
    int c = c_first, r = r_first;
#pragma unroll 8
    for (int q = lane; q < npk; q += 64) {
      const int e = q * V;
      Pack<T, V> p;
      if (e < nelem) p = load_pack<T, V, true>(xt + e);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) p.v[j] = T(0);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) tile[tile_addr(r + j, c)] = p.v[j];
      r += 64 * V;
      while (r >= dim) { r -= dim; ++c; }
    }
  }
  __builtin_amdgcn_wave_barrier();   // single-wave block: LDS is in order per wave, only stop compiler motion

  T ladj = T(0);
  const int ngroups = (A.nl + PLANAR_NLMAX - 1) / PLANAR_NLMAX;
  for (int gi = 0; gi < ngroups; ++gi) {
    const int g = INV ? ngroups - 1 - gi : gi;      // the inverse undoes the LAST group first
    const int l0 = g * PLANAR_NLMAX;
    const int ng = (A.nl - l0) < PLANAR_NLMAX ? (A.nl - l0) : PLANAR_NLMAX;
    if (ng == PLANAR_NLMAX) ladj += planar_tile_group<T, INV, true>(A, tile, dim, lane, l0, ng);
    else ladj += planar_tile_group<T, INV, false>(A, tile, dim, lane, l0, ng);
  }
  __builtin_amdgcn_wave_barrier();
  if (accumulate & 2) {                         // BJX_BASE_STDNORMAL: + log N(out; 0, I), lane = column
    T q0 = T(0), q1 = T(0);
    int r = 0;
    for (; r + 2 <= dim; r += 2) { const T a = tile[tile_addr(r, lane)], b = tile[tile_addr(r + 1, lane)]; q0 += a * a; q1 += b * b; }
    if (r < dim) { const T a = tile[tile_addr(r, lane)]; q0 += a * a; }
    ladj += T(-0.5) * (q0 + q1) - (T)dim * T(0.91893853320467274178);
  }
  // ---- stage out: swizzled tile -> coalesced packs
  if (y) {
    int c = c_first, r = r_first;
#pragma unroll 8
    for (int q = lane; q < npk; q += 64) {
      const int e = q * V;
      if (e < nelem) {
        Pack<T, V> p;
#pragma unroll
        for (int j = 0; j < V; ++j) p.v[j] = tile[tile_addr(r + j, c)];
        store_pack<T, V, true>(yt + e, p);
      }
      r += 64 * V;
      while (r >= dim) { r -= dim; ++c; }
    }
  }
  const bool ok = lane < ncols;
  if (ok && ladj_ps) ladj_ps[col0 + lane] = (accumulate & 1) ? ladj_ps[col0 + lane] + ladj : ladj;
  if (partials) {
    double acc = group_sum<64>(ok ? (double)ladj : 0.0);
    if (lane == 0) partials[blockIdx.x] = acc;
  }
}

// ------------------------------------------------------------------ Planar, register kernel (Float32)
// The LDS tile kernel above needs 32 KiB of LDS per wave at dim = 128, i.e. 5 waves per CU with no
// overlap of loads and arithmetic (15 % of the HBM roofline, profiles/r01_*).  Here the 64 x dim
// tile of a wave stays in REGISTERS in the coalesced layout the loads deliver (G = dim/4 lanes own
// one column as 16-byte packs, 64/G columns per wave instruction, G instructions in flight), and
// only the tiny [64 columns x NL layers] matrices of dot products / tanh values cross lanes:
//   1. p[k] = w_k[4gl..4gl+3] · z-pack                    (4 FMA per layer, lanes-along-dim)
//   2. transposed reduction over the G lanes of a column: v_permlane16_swap halves the number of
//      live values while it folds lane i+16 onto lane i, then DPP butterflies (quad_perm,
//      row_half_mirror, row_mirror) finish inside the 16-lane row; one lane per row stores the sums
//      to S[column][layer] in LDS (2 KiB per wave)
//   3. LANE = COLUMN: all 64 lanes run the NL-step scalar recurrence (same algebra as the tile
//      kernel) with hardware exp/log/rcp, and write t[column][layer] back to the same 2 KiB
//   4. z-pack += Σ_k û_k[4gl..4gl+3] t[column][k]          (4 FMA per layer), stored straight from
//      registers.
// 128 + 32 VGPRs at dim = 128 -> 2 waves per SIMD, every load of a wave in flight at once.

#include "bjx_flow_reg.inc"

template <int G, int NL, bool INV, int COLS, bool UNAL = false>
__global__ __launch_bounds__(256) void planar_reg_kernel(const PlanarRegArgs A, const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ ladj_ps, int dim, int64_t batch, int accumulate,
                                                         const BjxFin fin) {
  constexpr int CPS = 64 / G;                       // columns per wave instruction
  constexpr int NS = (COLS * G) / 64;               // pack steps for COLS columns
  static_assert(NS >= 1, "COLS too small for this G");
  constexpr bool SWAP = (G == 32) && (NL >= 2);     // fold lane i+16 onto i with a transposed halving
  constexpr int NV = SWAP ? NL / 2 : NL;            // live values per lane after the fold
  constexpr int RW = G < 16 ? G : 16;               // lanes of a row that still have to be summed
  constexpr int NP = NL >= 2 ? NL / 2 : 1;          // layer pairs (packed-FP32 math)
  __shared__ __attribute__((aligned(16))) float st_all[4][COLS * NL];
  __shared__ double red[4];
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* st = st_all[wave];
  const int gl = lane & (G - 1);
  const int cg = lane / G;                          // column inside a wave instruction
  const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * COLS;
  const RegGrid gr = reg_grid<UNAL>(col0 + cg, dim, 4 * gl);      // UNAL: see reg_load_pack
  const bool row_ok = gr.ok;
  const float* Aw = A.w + (UNAL ? A.lead - gr.phi : 0);
  const float* Au = A.u_hat + (UNAL ? A.lead - gr.phi : 0);
  const int64_t left = batch - col0;
  const int nvalid = left >= COLS ? COLS : (left > 0 ? (int)left : 0);
  const int64_t step_elems = (int64_t)CPS * dim;

  f4 z[NS];
  {
    const float* px = x + (col0 + cg) * dim + gr.rel;
    if constexpr (UNAL) {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) z[r] = reg_load_pack(px);
        else z[r] = f4{0.f, 0.f, 0.f, 0.f};
        px += step_elems;
      }
      reg_mask_tile(z, gr.lo, gr.hi);
    } else {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) z[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(px));
        else z[r] = f4{0.f, 0.f, 0.f, 0.f};
        px += step_elems;
      }
    }
  }
  float ladj = 0.f;
  const int ngroups = A.nl_pad / NL;
  for (int gi = 0; gi < ngroups; ++gi) {
    const int l0 = (INV ? ngroups - 1 - gi : gi) * NL;   // the inverse undoes the LAST group first
    // ---- 1+2: dot products against the group's w rows, reduced over the G lanes of each column
    {
      // wq[kp][j] = (w_{2kp}[4gl+j], w_{2kp+1}[4gl+j]): two layers per packed-FP32 instruction
      f2 wq[NP][4];
#pragma unroll
      for (int kp = 0; kp < NP; ++kp) {
        f4 a = f4{0.f, 0.f, 0.f, 0.f}, b = a;
        if (row_ok) {
          a = *reinterpret_cast<const bjx_pk4u*>(Aw + (int64_t)(l0 + 2 * kp) * (UNAL ? A.ldw : dim) + 4 * gl);
          if (NL >= 2) b = *reinterpret_cast<const bjx_pk4u*>(Aw + (int64_t)(l0 + 2 * kp + 1) * (UNAL ? A.ldw : dim) + 4 * gl);
        }
        wq[kp][0] = f2{a.x, b.x}; wq[kp][1] = f2{a.y, b.y}; wq[kp][2] = f2{a.z, b.z}; wq[kp][3] = f2{a.w, b.w};
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        float p[NL >= 2 ? NL : 2];
#pragma unroll
        for (int kp = 0; kp < NP; ++kp) {
          f2 acc = wq[kp][0] * z[r].x;
          acc += wq[kp][1] * z[r].y;
          acc += wq[kp][2] * z[r].z;
          acc += wq[kp][3] * z[r].w;
          p[2 * kp] = acc.x; p[2 * kp + 1] = acc.y;
        }
        float q[NV];
        if constexpr (SWAP) {
          swap_fold<NV>(p, q);
        } else if constexpr (G == 32) {   // NL == 1: plain fold of lane i+16 onto lane i
          p[1] = p[0];
          swap_fold<1>(p, q);
        } else {
#pragma unroll
          for (int k = 0; k < NV; ++k) q[k] = p[k];
        }
        row_allsum<RW, NV>(q);
        // rows (16 lanes): G=32 -> row 0/1 = layers lo/hi of column A, row 2/3 = of column B
        if ((lane & (RW - 1)) == 0) {
          int cl, lo;
          if (G == 32) { cl = r * CPS + (lane >> 5); lo = SWAP ? ((lane >> 4) & 1) * NV : 0; }
          else { cl = r * CPS + cg; lo = 0; }
          if (!(G == 32 && !SWAP && ((lane >> 4) & 1))) {
#pragma unroll
            for (int k = 0; k < NV; ++k) st[cl * NL + lo + k] = q[k];
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 3: scalar recurrence, one sample per lane
    if (COLS == 64 || lane < COLS) {
      float s[NL], t[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) { s[k] = st[lane * NL + k]; t[k] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = INV ? NL - 1 - kk : kk;
        const float* Gk = A.G + (int64_t)(l0 + k) * A.nl_pad + l0;
        float a = s[k];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if (!INV) { if (j < k) a += Gk[j] * t[j]; }
          else { if (j > k) a += Gk[j] * t[j]; }        // t holds -tanh for the inverse
        }
        const float bl = A.b[l0 + k], c = A.wtu_hat[l0 + k];
        float th, ld;
        if (INV) find_alpha_act(a, c, bl, th, ld);
        else planar_act(a + bl, c, th, ld);
        if (l0 + k >= A.n_layers) { th = 0.f; ld = 0.f; }    // padding layer (wave-uniform)
        ladj += INV ? -ld : ld;
        t[k] = INV ? -th : th;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NL; ++k) st[lane * NL + k] = t[k];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 4: rank-NL update of the register tile
    {
      f4 uv[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k)
        uv[k] = row_ok ? (f4)*reinterpret_cast<const bjx_pk4u*>(Au + (int64_t)(l0 + k) * (UNAL ? A.ldw : dim) + 4 * gl) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const float* tc = st + (r * CPS + cg) * NL;
#pragma unroll
        for (int k = 0; k < NL; ++k) { const float tk = tc[k]; z[r] += uv[k] * tk; }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (accumulate & 2) {
    // BJX_BASE_STDNORMAL: + log N(out; 0, I).  |out|² of a column is reduced over its G lanes exactly like a
    // one-layer dot product (fold + DPP butterflies), lands in st[column] and is picked up by lane = column.
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      float p[2];
      p[0] = z[r].x * z[r].x + z[r].y * z[r].y + z[r].z * z[r].z + z[r].w * z[r].w;
      float q[1];
      if constexpr (G == 32) { p[1] = p[0]; swap_fold<1>(p, q); }                  // lane i + lane i+16
      else q[0] = p[0];
      row_allsum<RW, 1>(q);
      if ((lane & (RW - 1)) == 0) {
        const int cl = G == 32 ? r * CPS + (lane >> 5) : r * CPS + cg;
        if (!(G == 32 && ((lane >> 4) & 1))) st[cl] = q[0];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (COLS == 64 || lane < COLS) ladj += -0.5f * st[lane] - (float)dim * 0.91893853320467274178f;
    __builtin_amdgcn_wave_barrier();
  }
  if (y) {
    float* py = y + (col0 + cg) * dim + gr.rel;
    if constexpr (UNAL) {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) reg_store_pack(py, z[r], gr.lo, gr.hi, A.unal == 2);
        py += step_elems;
      }
    } else {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) __builtin_nontemporal_store(z[r], reinterpret_cast<f4*>(py));
        py += step_elems;
      }
    }
  }
  const bool ok = lane < nvalid;   // nvalid <= COLS
  if (ok && ladj_ps) ladj_ps[col0 + lane] = (accumulate & 1) ? ladj_ps[col0 + lane] + ladj : ladj;
  block_publish_partial(ok ? (double)ladj : 0.0, red, fin);
}

// ------------------------------------------------------------------ Planar, MFMA kernel (Float32, forward, dim % 16 == 0)
// north_star: "MFMA used only for PlanarLayer's u·Wᵀ contraction".  Both dense steps of a layer group go to the matrix
// cores with the tile kept in the C/D layout of v_mfma_f32_16x16x4_f32 for its whole life:
//   lane (n = lane % 16, q = lane / 16) holds, for row block b, the 16-byte pack rows 16b + 4q .. +3 of column n
//   (16 columns per tile, 4 lanes per column, dim/16 packs per lane).
//   contraction  S[layer][col] = Σ_rows W[layer][row] Z[row][col]: register r of pack b IS the B operand
//                B[k = q][n] of the MFMA that contracts the rows {16b + 4q + r} (the order inside a dot product is
//                free), A = W[layer = lane % 16][16b + 4(lane/16) + r] (8 layers padded to 16): dim/4 MFMAs per tile.
//   update       Z[16b + m][col] += Σ_k Û[k][16b + m] t[k][col]: the pack is the C/D operand as it stands,
//                A = Û[4g + lane/16][16b + lane % 16], B = t[col = lane % 16][4g + lane/16]: 2 MFMAs per pack.
// The per-layer 32-lane butterflies (permlane swap + 4 DPP stages per layer pair) and the 4·NL FMAs per pack of
// planar_reg_kernel disappear; the NL-step scalar recurrence still runs lane = column on S through 2 KiB of LDS.
// LOADS: with STAGE = 0 a lane loads its packs directly (64 contiguous bytes per column and instruction, 16 columns
// per instruction); with STAGE = 1 a tile is loaded with fully coalesced 16-byte accesses and transposed through LDS.
typedef float mf4 __attribute__((ext_vector_type(4)));
template <int NB, int TILES, int STAGE>
__global__ __launch_bounds__(256) void planar_mfma_kernel(const PlanarRegArgs A, const float* __restrict__ x, float* __restrict__ y,
                                                          float* __restrict__ ladj_ps, int dim, int64_t batch, int accumulate,
                                                          const BjxFin fin) {
  constexpr int NL = 8, COLS = 16 * TILES;
  constexpr int PK = 4 * NB;                        // packs per column
  constexpr int PITCH = 4 * PK + 4;                 // floats per staged column (+1 pack: conflict-free column walk)
  __shared__ __attribute__((aligned(16))) float st_all[4][COLS * NL];
  __shared__ __attribute__((aligned(16))) float stage_all[STAGE ? 4 : 1][STAGE ? 16 * PITCH : 4];
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* st = st_all[wave];
  float* sg = stage_all[STAGE ? wave : 0];
  const int n = lane & 15, q = lane >> 4;
  const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * COLS;
  const int64_t left = batch - col0;
  const int nvalid = left >= COLS ? COLS : (left > 0 ? (int)left : 0);

  mf4 z[TILES][NB];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    if constexpr (STAGE == 0) {
      const bool ok = t * 16 + n < nvalid;
      const float* px = x + (col0 + t * 16 + n) * dim + 4 * q;
#pragma unroll
      for (int b = 0; b < NB; ++b)
        z[t][b] = ok ? __builtin_nontemporal_load(reinterpret_cast<const mf4*>(px + 16 * b)) : mf4{0.f, 0.f, 0.f, 0.f};
    } else {
      // coalesced: the tile's 16 columns are one contiguous run of 16*PK packs
      const float* px = x + (col0 + t * 16) * dim;
      mf4 tmp[PK / 4];
#pragma unroll
      for (int it = 0; it < PK / 4; ++it) {
        const int p = it * 64 + lane, c = p / PK;
        tmp[it] = (t * 16 + c < nvalid) ? __builtin_nontemporal_load(reinterpret_cast<const mf4*>(px) + p) : mf4{0.f, 0.f, 0.f, 0.f};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < PK / 4; ++it) {
        const int p = it * 64 + lane, c = p / PK, k = p - c * PK;
        *reinterpret_cast<mf4*>(sg + c * PITCH + 4 * k) = tmp[it];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int b = 0; b < NB; ++b) z[t][b] = *reinterpret_cast<const mf4*>(sg + n * PITCH + 4 * (4 * b + q));
      __builtin_amdgcn_wave_barrier();
    }
  }
  float ladj = 0.f;
  const int ngroups = A.nl_pad / NL;
  for (int gi = 0; gi < ngroups; ++gi) {
    const int l0 = gi * NL;
    // A operands of the group: W rows (layer = n < 8, else 0) and the Û rows of the update
    float wa[NB][4], ua[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      mf4 wv = mf4{0.f, 0.f, 0.f, 0.f};
      if (n < NL) wv = *reinterpret_cast<const mf4*>(A.w + (int64_t)(l0 + n) * dim + 16 * b + 4 * q);
      wa[b][0] = wv.x; wa[b][1] = wv.y; wa[b][2] = wv.z; wa[b][3] = wv.w;
      ua[b][0] = A.u_hat[(int64_t)(l0 + q) * dim + 16 * b + n];
      ua[b][1] = A.u_hat[(int64_t)(l0 + 4 + q) * dim + 16 * b + n];
    }
    // ---- contraction: S[4q + r][column n] in acc[r]; layers 0..7 live in q = 0, 1
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      mf4 acc = mf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][0], z[t][b].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][1], z[t][b].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][2], z[t][b].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][3], z[t][b].w, acc, 0, 0, 0);
      }
      if (q < 2) *reinterpret_cast<mf4*>(st + (t * 16 + n) * NL + 4 * q) = acc;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- scalar recurrence, one sample per lane (same algebra as planar_reg_kernel)
    if (COLS == 64 || lane < COLS) {
      float s[NL], tt[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) { s[k] = st[lane * NL + k]; tt[k] = 0.f; }
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const float* Gk = A.G + (int64_t)(l0 + k) * A.nl_pad + l0;
        float a = s[k];
#pragma unroll
        for (int j = 0; j < NL; ++j)
          if (j < k) a += Gk[j] * tt[j];
        float th, ld;
        planar_act(a + A.b[l0 + k], A.wtu_hat[l0 + k], th, ld);
        if (l0 + k >= A.n_layers) { th = 0.f; ld = 0.f; }    // padding layer (wave-uniform)
        ladj += ld;
        tt[k] = th;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NL; ++k) st[lane * NL + k] = tt[k];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- rank-8 update on the matrix cores: the pack is the accumulator
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const float t0 = st[(t * 16 + n) * NL + q], t1 = st[(t * 16 + n) * NL + 4 + q];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        z[t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b][0], t0, z[t][b], 0, 0, 0);
        z[t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b][1], t1, z[t][b], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (y) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      if constexpr (STAGE == 0) {
        const bool ok = t * 16 + n < nvalid;
        float* py = y + (col0 + t * 16 + n) * dim + 4 * q;
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (ok) __builtin_nontemporal_store(z[t][b], reinterpret_cast<mf4*>(py + 16 * b));
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) *reinterpret_cast<mf4*>(sg + n * PITCH + 4 * (4 * b + q)) = z[t][b];
        __builtin_amdgcn_wave_barrier();
        float* py = y + (col0 + t * 16) * dim;
#pragma unroll
        for (int it = 0; it < PK / 4; ++it) {
          const int p = it * 64 + lane, c = p / PK, k = p - c * PK;
          if (t * 16 + c < nvalid) __builtin_nontemporal_store(*reinterpret_cast<const mf4*>(sg + c * PITCH + 4 * k), reinterpret_cast<mf4*>(py) + p);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  const bool ok = lane < nvalid;
  if (ok && ladj_ps) ladj_ps[col0 + lane] = (accumulate & 1) ? ladj_ps[col0 + lane] + ladj : ladj;
  block_publish_partial(ok ? (double)ladj : 0.0, red, fin);
}

// ------------------------------------------------------------------ Planar, MFMA kernel (Float64, forward and inverse)
// Float64 is the data type of the reference's tests and Turing's default.  The register kernel above is Float32-only and the
// LDS-tile fallback ran 8 layers at d = 128 at 15 % of the HBM roofline.  In Float64 the matrix cores are the natural home of
// both dense steps: v_mfma_f64_16x16x4_f64 runs at twice the vector FP64 rate.  Same scheme as planar_mfma_kernel with the
// C/D layout of the Float64 instruction (probed: scripts/probe_mfma_f64.hip): lane (n = lane % 16, q = lane / 16), register r
// holds row 4r + q of the 16-row block, i.e. z[b][r] = Z[16b + 4r + q][column n].
//   contraction  MFMA (b, r) contracts the four consecutive rows 16b + 4r .. +3: B = z[b][r], A = W[layer = lane % 16][16b + 4r + q]
//                -> acc[rr] = S[layer 4rr + q][column n] (8 layers: rr = 0, 1 on all four q)
//   update       z[b] += MFMA(A = Û[4g + q][16b + n], B = t[column n][4g + q]), g = 0, 1
// The tile is loaded / stored with coalesced 16-byte accesses and transposed through LDS (the rows of a lane are 4 apart);
// W and Û of a group come from L1/L2 per use (8 KiB per group, shared by every wave).  INV: groups and layers last to
// first, find_alpha_dev per layer (planar_layer.jl:112-127,160-185), update with -tanh.
typedef double md4 __attribute__((ext_vector_type(4)));
// SPLIT: the tile goes through the LDS transpose in SPLIT row chunks (the staging tile of a wave is 16 columns x ROWS / SPLIT rows).
// Round 5 (profiles/r05_c4f64_pmc.md): with TILES = 2, SPLIT = 1 a wave holds 128 VGPRs of tile and 18.9 KiB of LDS — two waves
// per SIMD, 48 % of the wave cycles parked in s_waitcnt and nothing to switch to.  TILES = 1, SPLIT = 2 halves both: four waves
// per SIMD on the same bytes in flight per CU.
template <int NB, int TILES, bool INV, int SPLIT = 1>
__global__ __launch_bounds__(256) void planar_mfma64_kernel(const double* __restrict__ Wp, const double* __restrict__ Up, const double* __restrict__ Gp,
                                                            const double* __restrict__ cp, const double* __restrict__ bp, int nl_pad, int n_layers,
                                                            const double* __restrict__ x, double* __restrict__ y, double* __restrict__ ladj_ps, int dim,
                                                            int64_t batch, int accumulate, const BjxFin fin) {
  constexpr int NL = 8, COLS = 16 * TILES;
  constexpr int ROWS = 16 * NB;
  static_assert(NB % SPLIT == 0, "a staging chunk is whole 16-row blocks");
  constexpr int NBC = NB / SPLIT, RC = 16 * NBC;   // 16-row blocks / rows per staging chunk
  // doubles per staged column.  Round 6: + 2, not + 4.  The tile leaves the staging area with ds_read_b64 (lane groups {0-31}, {32-63}, 64 banks
  // of 4 bytes: MI355X_MICROARCH.md, LDS): lane (n, q) reads dword 2·(n·PITCH + 16b + 4r + q), so columns n and n + 8 met on the same banks
  // with PITCH = RC + 4 (264 n mod 64 = 8 n: 2-way on every read; the transposing ds_write_b64 of the way out 4-way, its groups are 16 lanes on 32
  // banks) — profiles/r05_pmc_notes.md: LDS 32 % busy, 64 % of it conflicts.  With RC + 2 the 32 lanes of a read group cover the 64 banks once
  // (260 n mod 64 = 4 n, + 2q + {0, 1}); the write is 2-way.  Columns stay 16-byte aligned for the d2 accesses of the coalesced side.
  constexpr int PITCH = RC + 2;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sg = reinterpret_cast<double*>(smem_) + (size_t)wave * (16 * PITCH + COLS * NL);
  double* st = sg + 16 * PITCH;
  const int n = lane & 15, q = lane >> 4;
  const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * COLS;
  const int64_t left = batch - col0;
  const int nvalid = left >= COLS ? COLS : (left > 0 ? (int)left : 0);
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int PK = RC / 2;                        // 16-byte packs per column and chunk
  constexpr int NIT = (16 * PK + 63) / 64;          // pack loads per lane, tile and chunk

  double z[TILES][NB][4];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
#pragma unroll
    for (int h = 0; h < SPLIT; ++h) {
      const double* px = x + (col0 + t * 16) * dim + h * RC;
      d2 tmp[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int p = it * 64 + lane, c = p / PK, k = p - c * PK;
        tmp[it] = (p < 16 * PK && t * 16 + c < nvalid) ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(px + (int64_t)c * dim) + k) : d2{0., 0.};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int p = it * 64 + lane, c = p / PK, k = p - c * PK;
        if (p < 16 * PK) *reinterpret_cast<d2*>(sg + c * PITCH + 2 * k) = tmp[it];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int b = 0; b < NBC; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][h * NBC + b][r] = sg[n * PITCH + 16 * b + 4 * r + q];
      __builtin_amdgcn_wave_barrier();
    }
  }
  double ladj = 0.0;
  const int ngroups = nl_pad / NL;
  for (int gi = 0; gi < ngroups; ++gi) {
    const int l0 = (INV ? ngroups - 1 - gi : gi) * NL;
    // ---- contraction
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      md4 acc = md4{0., 0., 0., 0.}, acc2 = md4{0., 0., 0., 0.};      // two chains: a dependent MFMA waits for the whole pass of its predecessor
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double wa = n < NL ? Wp[(int64_t)(l0 + n) * dim + 16 * b + 4 * r + q] : 0.0;
          if (r & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, z[t][b][r], acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, z[t][b][r], acc, 0, 0, 0);
        }
      st[(t * 16 + n) * NL + q] = acc[0] + acc2[0];
      st[(t * 16 + n) * NL + 4 + q] = acc[1] + acc2[1];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- scalar recurrence, one sample per lane.  Only tanh feeds the next layer; the log-det term log1p(wᵀû·sech²) of a layer (a lean
    // Float64 log1p: ~50 of the ~120 dependent operations per layer) does not, so it leaves the serial chain (round 6): the lanes
    // that run the recurrence park the eight arguments c·sech² in the idle staging area, and ALL 64 lanes take the logs afterwards —
    // NL·COLS / 64 each (four at 32 columns per wave, where half the lanes used to idle through the whole recurrence).
    constexpr int LG = 64 / COLS;                             // lane groups per column set
    static_assert(NL % LG == 0, "the layers of a group split evenly over the lane groups");
    if (COLS == 64 || lane < COLS) {
      double s[NL], tt[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) { s[k] = st[lane * NL + k]; tt[k] = 0.0; }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = INV ? NL - 1 - kk : kk;
        const double* Gk = Gp + (int64_t)(l0 + k) * nl_pad + l0;
        double a = s[k];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if (!INV) { if (j < k) a += Gk[j] * tt[j]; }
          else { if (j > k) a += Gk[j] * tt[j]; }         // tt holds -tanh for the inverse
        }
        const double bl = bp[l0 + k], c = cp[l0 + k];
        double th, s2;
        if (INV) find_alpha_act64(a, c, bl, th, s2);
        else flow_tanh_sech2(a + bl, th, s2);
        double arg = c * s2;                                // planar_layer.jl:107: logdet term = log1p(wᵀû·sech²)
        if (l0 + k >= n_layers) { th = 0.0; arg = 0.0; }    // padding layer (wave-uniform): log1p(0) = 0
        sg[lane * NL + k] = arg;
        tt[k] = INV ? -th : th;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NL; ++k) st[lane * NL + k] = tt[k];
    }
    __builtin_amdgcn_wave_barrier();
    {
      const int colL = lane & (COLS - 1), grp = lane / COLS;
      double ld = 0.0;
#pragma unroll
      for (int k = 0; k < NL / LG; ++k) ld += Fast<double>::log1p(sg[colL * NL + grp * (NL / LG) + k]);
#pragma unroll
      for (int m = COLS; m < 64; m <<= 1) ld += shfl_xor(ld, m);          // the lane groups of a column (every lane of it gets the sum)
      ladj += INV ? -ld : ld;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- rank-8 update: the tile registers are the accumulator
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const double t0 = st[(t * 16 + n) * NL + q], t1 = st[(t * 16 + n) * NL + 4 + q];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        md4 zz = md4{z[t][b][0], z[t][b][1], z[t][b][2], z[t][b][3]};
        zz = __builtin_amdgcn_mfma_f64_16x16x4f64(Up[(int64_t)(l0 + q) * dim + 16 * b + n], t0, zz, 0, 0, 0);
        zz = __builtin_amdgcn_mfma_f64_16x16x4f64(Up[(int64_t)(l0 + 4 + q) * dim + 16 * b + n], t1, zz, 0, 0, 0);
        z[t][b][0] = zz[0]; z[t][b][1] = zz[1]; z[t][b][2] = zz[2]; z[t][b][3] = zz[3];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (accumulate & 2) {
    // BJX_BASE_STDNORMAL: + log N(out; 0, I): |out|^2 of a column = my 4*NB entries, summed over the four q lanes
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      double p = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) p += z[t][b][r] * z[t][b][r];
      p += shfl_xor(p, 16);
      p += shfl_xor(p, 32);
      if (q == 0) st[t * 16 + n] = p;
    }
    __builtin_amdgcn_wave_barrier();
    if (COLS == 64 || lane < COLS) ladj += -0.5 * st[lane] - (double)dim * 0.91893853320467274178;
    __builtin_amdgcn_wave_barrier();
  }
  if (y) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
      for (int h = 0; h < SPLIT; ++h) {
#pragma unroll
        for (int b = 0; b < NBC; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) sg[n * PITCH + 16 * b + 4 * r + q] = z[t][h * NBC + b][r];
        __builtin_amdgcn_wave_barrier();
        double* py = y + (col0 + t * 16) * dim + h * RC;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int p = it * 64 + lane, c = p / PK, k = p - c * PK;
          if (p < 16 * PK && t * 16 + c < nvalid) __builtin_nontemporal_store(*reinterpret_cast<const d2*>(sg + c * PITCH + 2 * k), reinterpret_cast<d2*>(py + (int64_t)c * dim) + k);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  const bool ok = lane < nvalid;
  if (ok && ladj_ps) ladj_ps[col0 + lane] = (accumulate & 1) ? ladj_ps[col0 + lane] + ladj : ladj;
  block_publish_partial(ok ? ladj : 0.0, red, fin);
}

// ------------------------------------------------------------------ Planar input pullback, register kernel (Float32)
// bjx_planar_vjp for 16 < dim <= 128: the two sweeps of planar_vjp_kernel on the register tile of planar_reg_kernel.
// The reverse sweep IS the forward structure with the roles of w and û exchanged: with s̄_k the cotangent of s_k,
//   û_kᵀz̄_k = û_kᵀȳ + Σ_{j>k} (û_kᵀw_j) s̄_j        (the same G table, read transposed),   z̄_0 = ȳ + Σ_k w_k s̄_k
// so it is NL dot products against û per pack, the transposed reduction, a lane = column scalar recurrence from the
// last layer down, and a rank-NL update with w.  The primal tile is dead once tanh(s_k) of every layer sits in LDS
// ([column][layer], 4·nl_pad bytes per column), so ȳ is loaded into the same registers: read z, read ȳ, write z̄.
template <int NL, bool INV, int NW = 2, bool UNAL = false>
__global__ __launch_bounds__(NW <= 4 ? 256 : NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 1, 8))) void planar_reg2_kernel(const PlanarRegArgs A, const float* __restrict__ x, float* __restrict__ y,
                                                          float* __restrict__ ladj_ps, int dim, int64_t batch, int accumulate, const BjxFin fin) {
  constexpr int NWB = NW <= 4 ? 4 : NW;                      // waves per block
  // (a tile of 32 columns — NS = 8 packs, 32 floats — is merged by the compiler into ONE 32-register value that it copies whole at
  //  every conditional load: 211+ VGPRs, thousands of spills under the 128-register budget of a 1024-thread block.  Keep NS = 16.)
  constexpr int G = 16, COLS = 64, CPS = 4, NS = COLS / CPS;
  __shared__ __attribute__((aligned(16))) float sS[2][NWB][COLS * NL];   // partial dot products [parity][wave]
  __shared__ __attribute__((aligned(16))) float sT[NWB][COLS * NL];      // tanh values of my tile (each wave its own copy)
  __shared__ double red[NWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = wave / NW, half = wave % NW;              // half = which 64-row slice of the tile
  const int gl = lane & (G - 1), cg = lane / G;
  const int lc = lane & (COLS - 1);                          // my column in the lane = column steps
  const int row0 = half * 64;
  const int64_t col0 = ((int64_t)blockIdx.x * (NWB / NW) + tile) * COLS;
  const RegGrid gr = reg_grid<UNAL>(col0 + cg, dim, row0 + 4 * gl);      // UNAL: see reg_load_pack
  const bool row_ok = gr.ok;
  const float* Aw = A.w + (UNAL ? A.lead - gr.phi : 0);
  const float* Au = A.u_hat + (UNAL ? A.lead - gr.phi : 0);
  const int64_t left = batch - col0;
  const int nvalid = left >= COLS ? COLS : (left > 0 ? (int)left : 0);
  const int64_t step_elems = (int64_t)CPS * dim;
  bjx_f4 z[NS];
  {
    const float* px = x + (col0 + cg) * dim + gr.rel;
    if constexpr (UNAL) {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) z[r] = reg_load_pack(px);
        else z[r] = bjx_f4{0.f, 0.f, 0.f, 0.f};
        px += step_elems;
      }
      reg_mask_tile(z, gr.lo, gr.hi);
    } else {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) z[r] = __builtin_nontemporal_load(reinterpret_cast<const bjx_f4*>(px));
        else z[r] = bjx_f4{0.f, 0.f, 0.f, 0.f};
        px += step_elems;
      }
    }
  }
  float ladj = 0.f;
  const int ngroups = A.nl_pad / NL;
  float* stT = sT[wave];
  for (int gi = 0; gi < ngroups; ++gi) {
    const int l0 = (INV ? ngroups - 1 - gi : gi) * NL;
    float* mineS = sS[gi & 1][wave];
    reg_dots<G, NL, NS>(Aw, l0, (UNAL ? A.ldw : dim), z, mineS, lane, gl, cg, row_ok, row0);
    __syncthreads();                                         // every slice of every tile has published its partial sums
    {
      // (COLS = 32: lanes 32..63 run the recurrence of column lane - 32 along — same values, same stores; a divergent region
      //  around it made the compiler copy the register tile at every merge)
      float s[NL], t[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) { s[k] = sS[gi & 1][tile * NW][lc * NL + k]; t[k] = 0.f; }
      // fixed order over the slices: every wave of the tile gets the same bits (four slices at a time: all sixteen in flight
      // cost the 1024-thread variant its 128-register budget)
#pragma unroll 4
      for (int pp = 1; pp < NW; ++pp) {
#pragma unroll
        for (int k = 0; k < NL; ++k) s[k] += sS[gi & 1][tile * NW + pp][lc * NL + k];
      }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = INV ? NL - 1 - kk : kk;
        const float* Gk = A.G + (int64_t)(l0 + k) * A.nl_pad + l0;
        float a = s[k];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if (!INV) { if (j < k) a += Gk[j] * t[j]; }
          else { if (j > k) a += Gk[j] * t[j]; }
        }
        const float bl = A.b[l0 + k], c = A.wtu_hat[l0 + k];
        float th, ld;
        if (INV) find_alpha_act(a, c, bl, th, ld);
        else planar_act(a + bl, c, th, ld);
        if (l0 + k >= A.n_layers) { th = 0.f; ld = 0.f; }    // padding layer (wave-uniform)
        ladj += INV ? -ld : ld;
        t[k] = INV ? -th : th;
      }
      __builtin_amdgcn_wave_barrier();                       // my previous group's reads of stT are done (same wave)
#pragma unroll
      for (int k = 0; k < NL; ++k) stT[lc * NL + k] = t[k];
    }
    __builtin_amdgcn_wave_barrier();
    reg_update<G, NL, NS>(Au, l0, (UNAL ? A.ldw : dim), z, stT, gl, cg, row_ok, row0);
    __builtin_amdgcn_wave_barrier();
  }
  if (accumulate & 2) {
    // BJX_BASE_STDNORMAL: |out|² of a column = the 64-row slices of all the tile's waves
    float* mineS = sS[ngroups & 1][wave];
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      float q[1];
      q[0] = z[r].x * z[r].x + z[r].y * z[r].y + z[r].z * z[r].z + z[r].w * z[r].w;
      row_allsum<16, 1>(q);
      if ((lane & 15) == 0) mineS[r * CPS + cg] = q[0];
    }
    __syncthreads();
    float q2 = sS[ngroups & 1][tile * NW][lane & (COLS - 1)];
#pragma unroll
    for (int pp = 1; pp < NW; ++pp) q2 += sS[ngroups & 1][tile * NW + pp][lane & (COLS - 1)];
    ladj += -0.5f * q2 - (float)dim * 0.91893853320467274178f;
  }
  if (y) {
    float* py = y + (col0 + cg) * dim + gr.rel;
    if constexpr (UNAL) {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) reg_store_pack(py, z[r], gr.lo, gr.hi, A.unal == 2);
        py += step_elems;
      }
    } else {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        if (row_ok && r * CPS + cg < nvalid) __builtin_nontemporal_store(z[r], reinterpret_cast<bjx_f4*>(py));
        py += step_elems;
      }
    }
  }
  const bool ok = lane < nvalid && half == 0;                // the two halves hold the same log-det: one of them reports it
  if (ok && ladj_ps) ladj_ps[col0 + lane] = (accumulate & 1) ? ladj_ps[col0 + lane] + ladj : ladj;
  block_publish_partial(ok ? (double)ladj : 0.0, red, fin);
}

template <class T> struct RadialArgs {
  const T *alpha_, *beta, *z0;
  int in_lds;
};

// radial_layer.jl:43-72 (forward) and :88-129 (inverse).  UC = 4/R columns per lane group are in
// flight at once (one 16-byte pack per lane and column is latency-bound: 39 % of the HBM roofline).
template <int R> struct RadialUC { static constexpr int value = R == 1 ? 4 : (R == 2 ? 2 : 1); };
template <class T, int V, int R, bool INV>
__global__ __launch_bounds__(256) void radial_kernel(const RadialArgs<T> A, const T* x, T* y, T* ladj_ps, int64_t dim,
                                                     int64_t batch, int G, int accumulate, double* partials) {
  constexpr int UC = RadialUC<R>::value;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  T* tab = reinterpret_cast<T*>(smem + 32);
  if (A.in_lds) {
    for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) tab[i] = A.z0[i];
    __syncthreads();
  }
  const T* Z0 = A.in_lds ? tab : A.z0;
  const T alpha = d_log1pexp(A.alpha_[0]);          // :44
  const T apb = d_log1pexp(A.beta[0]);              // α + β̂
  const T beta_hat = -alpha + apb;                  // :45

  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = blockDim.x / G;
  const int64_t nvc = (dim + V - 1) / V;         // the last pack may be partial (odd heights: element-aligned packs, load_pack_part)
  double acc = 0.0;
  // non-persistent grid: UC columns per G-lane group.  Lanes of a group past the batch keep
  // running (on column batch-1, results discarded) so the group shuffles stay convergent.
  const int64_t col_first = (int64_t)blockIdx.x * cols_per_block * UC + threadIdx.x / G;
  Pack<T, V> zz[UC][R];
  T z0r[R][V];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t v = gl + (int64_t)r * G;
#pragma unroll
    for (int j = 0; j < V; ++j) z0r[r][j] = v * V + j < dim ? Z0[v * V + j] : T(0);
  }
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t col_raw = col_first + (int64_t)u * cols_per_block;
    const int64_t col = col_raw < batch ? col_raw : batch - 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) zz[u][r] = load_pack_part<T, V>(x + col * dim + v * V, (int)(dim - v * V < V ? dim - v * V : V));
    }
  }
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t col_raw = col_first + (int64_t)u * cols_per_block;
    const bool col_ok = col_raw < batch;
    const int64_t col = col_ok ? col_raw : batch - 1;
    T* yc = y + col * dim;
    T ss = T(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
#pragma unroll
        for (int j = 0; j < V; ++j) { const T dlt = zz[u][r].v[j] - z0r[r][j]; ss += dlt * dlt; }
      }
    }
    ss = group_sum_rt(ss, G);
    T r_fwd, gain;   // out = z0 + gain * dz ; r_fwd = ‖z_fwd_input − z0‖ used by the log-det
    if (!INV) {
      r_fwd = d_sqrt(ss);
      gain = T(1) + beta_hat / (alpha + r_fwd);     // z + β̂/(α+r)(z−z0) = z0 + (1+β̂h)(z−z0)
    } else {
      const T gam = d_sqrt(ss);                     // compute_r :124-129
      const T a = apb - gam;
      const T rr = (d_sqrt(a * a + 4 * alpha * gam) - a) / 2;
      gain = (alpha + rr) / (apb + rr);             // γ :96-101
      r_fwd = gain * gam;                           // ‖z − z0‖ of the result
    }
    const T h_ = T(1) / (alpha + r_fwd);
    T ld = T(dim - 1) * d_log(T(1) + beta_hat * h_) + d_log(T(1) + beta_hat * h_ + beta_hat * (-(h_ * h_)) * r_fwd);   // :68-70
    if (INV) ld = -ld;
    const T fwd_gain = beta_hat / (alpha + r_fwd);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
        Pack<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const T dlt = zz[u][r].v[j] - z0r[r][j];
          if (!INV) o.v[j] = zz[u][r].v[j] + fwd_gain * dlt;                  // :52
          else o.v[j] = z0r[r][j] + gain * dlt;                              // :101
        }
        if (col_ok) store_pack_part<T, V>(yc + v * V, o, (int)(dim - v * V < V ? dim - v * V : V));
      }
    }
    if (col_ok && gl == 0) {
      if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + ld : ld;
      acc += (double)ld;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// ------------------------------------------------------------------ Radial input pullback
// Closed-form derivatives of radial_layer.jl:43-129.  With δ = z - z₀, r = ‖δ‖, h = 1/(α + r), a = 1 + β̂h, c = -β̂h²/r:
//   J = a I + c δδᵀ (symmetric),  ℓ'(r) = (d-1)(-β̂h²)/a + (-2β̂h² + 2β̂h³r)/(1 + β̂h - β̂h²r)
//   forward:  z̄ = a ȳ + c (δᵀȳ) δ + ℓ̄ ℓ'(r) δ/r
//   inverse:  v = z̄ - ℓ̄ ℓ'(r) δ/r,  ȳ = (v - c (δᵀv) δ/(a + c r²))/a      (Sherman–Morrison; δ, r at the pre-image,
//             which the closed-form inverse gives as δ = γ (y - z₀))
// Same mapping as radial_kernel: G lanes own a column in registers; two group reductions (‖δ‖², δᵀḡ) per column.
template <class T, int V, int R, bool INV>
__global__ __launch_bounds__(256) void radial_vjp_kernel(const RadialArgs<T> A, const T* __restrict__ x, const T* __restrict__ gbar,
                                                         const T* __restrict__ lbar, T* __restrict__ xbar, int64_t dim, int64_t batch, int G,
                                                         T* __restrict__ work = nullptr, double* __restrict__ zpart = nullptr, int zoff = 0) {
  // work (forward map only, may be null): r and δᵀȳ of every column, [2, batch] — the inputs of the parameter pullback
  // zpart (may be null): Σ over the block's columns of ȳ - z̄ per row -> zpart[blockIdx][dim] (z̄₀ without a second pass)
  constexpr int UC = R == 1 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tab = reinterpret_cast<T*>(smem);
  if (A.in_lds) {
    for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) tab[i] = A.z0[i];
    __syncthreads();
  }
  const T* Z0 = A.in_lds ? tab : A.z0;
  const T alpha = d_log1pexp(A.alpha_[0]);
  const T apb = d_log1pexp(A.beta[0]);
  const T bh = -alpha + apb;
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = blockDim.x / G;
  const int64_t nvc = (dim + V - 1) / V;                 // the last pack may be partial (odd heights: element-aligned packs)
  const int64_t col_first = (int64_t)blockIdx.x * cols_per_block * UC + threadIdx.x / G;
  Pack<T, V> zz[UC][R], gg[UC][R];
  T z0r[R][V];
  T zsum[R][V];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t v = gl + (int64_t)r * G;
#pragma unroll
    for (int j = 0; j < V; ++j) { z0r[r][j] = v * V + j < dim ? Z0[v * V + j] : T(0); zsum[r][j] = T(0); }
  }
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t col_raw = col_first + (int64_t)u * cols_per_block;
    const int64_t col = col_raw < batch ? col_raw : batch - 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
        const int nrow = (int)(dim - v * V < V ? dim - v * V : V);
        zz[u][r] = load_pack_part<T, V>(x + col * dim + v * V, nrow); gg[u][r] = load_pack_part<T, V>(gbar + col * dim + v * V, nrow);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t col_raw = col_first + (int64_t)u * cols_per_block;
    const bool col_ok = col_raw < batch;
    const int64_t col = col_ok ? col_raw : batch - 1;
    T ss = T(0), dg = T(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
#pragma unroll
        for (int j = 0; j < V; ++j) { const T dlt = zz[u][r].v[j] - z0r[r][j]; ss += dlt * dlt; dg += dlt * gg[u][r].v[j]; }
      }
    }
    ss = group_sum_rt(ss, G);
    dg = group_sum_rt(dg, G);
    T rr, gain = T(1);                       // δ at the point where J is evaluated = gain · (input - z₀)
    if (!INV) rr = d_sqrt(ss);
    else {
      const T gam = d_sqrt(ss);              // compute_r, radial_layer.jl:124-129
      const T aa = apb - gam;
      const T r0 = (d_sqrt(aa * aa + 4 * alpha * gam) - aa) / 2;
      gain = (alpha + r0) / (apb + r0);
      rr = gain * gam;
    }
    const T h = T(1) / (alpha + rr);
    const T a = T(1) + bh * h;
    const T rinv = rr > T(0) ? T(1) / rr : T(0);
    const T c = -bh * h * h * rinv;
    const T lr = T(dim - 1) * (-bh * h * h) / a + (T(-2) * bh * h * h + T(2) * bh * h * h * h * rr) / (T(1) + bh * h - bh * h * h * rr);
    const T lb = lbar ? lbar[col] : T(0);
    const T kl = lb * lr * rinv;             // coefficient of δ from the log-det term
    if (!INV && work && col_ok && gl == 0) { work[col] = rr; work[batch + col] = dg; }
    T ca, cd;                                // out = ca · ḡ + cd · δ_in   (δ_in = input - z₀; δ = gain · δ_in)
    if (!INV) { ca = a; cd = c * dg + kl; }
    else {
      // v = ḡ - kl δ;  δᵀv = gain·dg - kl r²;  out = v/a - c (δᵀv) δ / (a (a + c r²))
      const T dv = gain * dg - kl * rr * rr;
      ca = T(1) / a;
      cd = gain * (-kl / a - c * dv / (a * (a + c * rr * rr)));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
        Pack<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = ca * gg[u][r].v[j] + cd * (zz[u][r].v[j] - z0r[r][j]);
        if (col_ok) store_pack_part<T, V>(xbar + col * dim + v * V, o, (int)(dim - v * V < V ? dim - v * V : V));
        if (zpart && col_ok) {
#pragma unroll
          for (int j = 0; j < V; ++j) zsum[r][j] += gg[u][r].v[j] - o.v[j];
        }
      }
    }
  }
  if (zpart) {
    // the column groups of a wave (fixed butterfly over lanes gl, gl + G, ...), then the four waves through LDS
    double* zp = reinterpret_cast<double*>(smem + zoff);
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        T a = zsum[r][j];
        for (int m = G; m < 64; m <<= 1) a += shfl_xor(a, m);
        const int64_t v = gl + (int64_t)r * G;
        if ((threadIdx.x & 63) < G && v * V + j < dim) zp[(size_t)wv * dim + v * V + j] = (double)a;
      }
    }
    __syncthreads();
    for (int64_t e = threadIdx.x; e < dim; e += blockDim.x)
      zpart[(size_t)blockIdx.x * dim + e] = (zp[e] + zp[dim + e]) + (zp[2 * dim + e] + zp[3 * dim + e]);
  }
}

// out[b][e] = Σ_{k in chunk b} in[k][e]: fixed order, coalesced, four accumulators (the same reduction as in bjx_stacked.hip)
__global__ __launch_bounds__(256) void flow_sets_reduce_kernel(const double* __restrict__ in, int64_t nsets, int per, int chunk, double* __restrict__ out) {
  const int64_t k0 = (int64_t)blockIdx.x * chunk;
  const int64_t k1 = k0 + chunk < nsets ? k0 + chunk : nsets;
  for (int e = threadIdx.x; e < per; e += blockDim.x) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t k = k0;
    for (; k + 4 <= k1; k += 4) { a0 += in[k * per + e]; a1 += in[(k + 1) * per + e]; a2 += in[(k + 2) * per + e]; a3 += in[(k + 3) * per + e]; }
    for (; k < k1; ++k) a0 += in[k * per + e];
    out[(size_t)blockIdx.x * per + e] = (a0 + a1) + (a2 + a3);
  }
}

// ------------------------------------------------------------------ low-dimensional flows: ONE LANE per column (dim <= 16)
// Planar and radial flows are mostly used on 2 ... 10 dimensional densities.  With G lanes along a column such columns leave
// most of a wave idle and every layer pays a cross-lane reduction for a dot product of a few terms (dim = 2, 8 layers: 6 % of the
// roofline).  Here a wave takes 64 consecutive columns — one contiguous run, 16-byte packs through a [64][P odd] LDS tile — lane t
// keeps column t in registers and runs the whole stack on it: dot products and rank-1 updates are plain FMAs of the lane, the
// layer parameters are wave-uniform (scalar loads), nothing crosses lanes.
// DX > 0: columns of exactly DX <= 8 rows are read and written by their lane directly (TinyCol: one or two
// multi-dword accesses), no tile: at dim = 2 ... 7 the staging, not the stack, was the cost (1 layer, dim = 2 / 3 / 5: 24 / 34 / 43 %).
template <class T, int DMAX, bool INV, int V, int DX = 0>
__global__ __launch_bounds__(64) void planar_walk_kernel(const T* __restrict__ Aw, const T* __restrict__ Auh, const T* __restrict__ Ac, const T* __restrict__ Ab, int n_layers,
                                                         const T* __restrict__ x, T* __restrict__ y, T* __restrict__ ladj_ps, int dim, int P,
                                                         int64_t batch, int accumulate, double* partials) {
  // The layer tables go to LDS once per block, zero-padded to DMAX rows: [w | û | b, wᵀû] per layer, read back as wave-uniform
  // (broadcast) 16-byte LDS reads.  (Read in place they were 2·dim + 2 dependent loads of one address per layer — per-lane global
  // loads through the argument struct, one-dword scalar loads as `const __restrict__` arguments — and the walk was bound by them.)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[1];
  T* tile = reinterpret_cast<T*>(smem);
  constexpr int LW = 2 * DMAX + 4;
  T* tab = tile + (((size_t)64 * P + 3) / 4) * 4;
  const int lane = threadIdx.x;
  for (int i = lane; i < n_layers * LW; i += 64) {
    const int l = i / LW, q = i - l * LW;
    T v = T(0);
    if (q < DMAX) { if (q < dim) v = Aw[l * dim + q]; }
    else if (q < 2 * DMAX) { if (q - DMAX < dim) v = Auh[l * dim + q - DMAX]; }
    else if (q == 2 * DMAX) v = Ab[l];
    else if (q == 2 * DMAX + 1) v = Ac[l];
    tab[i] = v;
  }
  tile_sync();
  double acc = 0.0;
  for (int64_t c0 = (int64_t)blockIdx.x * 64; c0 < batch; c0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
    T* mine = tile + lane * P;
    T z[DMAX];
    if constexpr (DX > 0) {
      TinyCol<T, DX> t{};
      if (lane < ncols) t = *reinterpret_cast<const TinyCol<T, DX>*>(x + (c0 + lane) * DX);
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] = r < DX ? t.v[r < DX ? r : 0] : T(0);
    } else {
      tile_stage_in<T, V>(tile, x + c0 * dim, dim, P, ncols, lane);
      tile_sync();
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] = r < dim ? mine[r] : T(0);
    }
    T ladj = T(0);
    for (int li = 0; li < n_layers; ++li) {
      const int l = INV ? n_layers - 1 - li : li;
      const T* tl = tab + l * LW;
      T wv[DMAX], uv[DMAX];
#pragma unroll
      for (int r = 0; r < DMAX; ++r) { wv[r] = tl[r]; uv[r] = tl[DMAX + r]; }
      const T bl = tl[2 * DMAX], c = tl[2 * DMAX + 1];
      T s0 = T(0), s1 = T(0);
#pragma unroll
      for (int r = 0; r < DMAX; r += 2) { s0 += wv[r] * z[r]; s1 += wv[r + 1] * z[r + 1]; }   // padded rows: 0 · 0
      const T s = s0 + s1;                                           // wᵀz (src/utils.jl:2)
      T t, ld;
      if constexpr (sizeof(T) == 4) {                                // the activation of the register kernels: tanh, sech², log1p from one exp
        if (!INV) planar_act(s + bl, c, t, ld); else find_alpha_act(s, c, bl, t, ld);
      } else {
        T s2;
        if (INV) planar_inv_act<T>(s, c, bl, t, s2);
        else flow_tanh_sech2(s + bl, t, s2);
        ld = Fast<T>::log1p(c * s2);                                 // planar_layer.jl:107
      }
      ladj += INV ? -ld : ld;
      const T tt = INV ? -t : t;
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] += uv[r] * tt;             // z ± û tanh(·)
    }
    if (accumulate & 2) {                                            // BJX_BASE_STDNORMAL: + log N(out; 0, I)
      T q = T(0);
#pragma unroll
      for (int r = 0; r < DMAX; ++r) q += z[r] * z[r];
      ladj += T(-0.5) * q - (T)dim * T(0.91893853320467274178);
    }
    if constexpr (DX > 0) {
      if (y && lane < ncols) {
        TinyCol<T, DX> t;
#pragma unroll
        for (int r = 0; r < DX; ++r) t.v[r] = z[r];
        *reinterpret_cast<TinyCol<T, DX>*>(y + (c0 + lane) * DX) = t;
      }
    } else {
      if (y) {
#pragma unroll
        for (int r = 0; r < DMAX; ++r) if (r < dim) mine[r] = z[r];
      }
      tile_sync();
      if (y) tile_stage_out<T, V>(tile, y + c0 * dim, dim, P, ncols, lane);
      tile_sync();
    }
    if (lane < ncols) {
      if (ladj_ps) ladj_ps[c0 + lane] = (accumulate & 1) ? ladj_ps[c0 + lane] + ladj : ladj;
      acc += (double)ladj;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

__device__ __forceinline__ float walk_tanh(float v) { return fast_tanh(v); }      // the register kernels' one-exp tanh (parity bar 1e-3)
__device__ __forceinline__ double walk_tanh(double v) { return flow_tanh(v); }
// Pullback of the Planar stack on low-dimensional columns, one lane per column (the arithmetic of planar_vjp_kernel): x and ȳ
// through two odd-pitch tiles, the primal sweep leaves t_k = tanh(·) of every layer in the lane's strip of LDS scratch, the reverse
// sweep runs on the cotangent in registers.  t_out / s_out (the per-column, per-layer values the parameter pullback reads) as in
// planar_vjp_kernel.
template <class T, int DMAX, bool INV, int V, int DX = 0>     // DX > 0: x, ȳ and x̄ move per lane as whole columns (see planar_walk_kernel)
__global__ __launch_bounds__(64) void planar_vjp_walk_kernel(const T* __restrict__ Aw, const T* __restrict__ Auh, const T* __restrict__ Ac, const T* __restrict__ Ab, int n_layers,
                                                             const T* __restrict__ x, const T* __restrict__ ybar, const T* __restrict__ lbar, T* __restrict__ xbar, int dim, int P, int NLP,
                                                             int64_t batch, T* __restrict__ t_out, T* __restrict__ s_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tx = reinterpret_cast<T*>(smem);
  T* tg = tx + (size_t)64 * P;
  T* tsave = tg + (size_t)64 * P;                                   // [64][NLP]: tanh of every layer
  T* ssave = tsave + (size_t)64 * NLP;                              // [64][NLP]: s̄ of every layer (for t_out / s_out: stored as contiguous runs)
  constexpr int LW = 2 * DMAX + 4;
  T* tab = ssave + (((size_t)64 * NLP + 3) / 4) * 4;
  const int lane = threadIdx.x;
  for (int i = lane; i < n_layers * LW; i += 64) {
    const int l = i / LW, q = i - l * LW;
    T v = T(0);
    if (q < DMAX) { if (q < dim) v = Aw[l * dim + q]; }
    else if (q < 2 * DMAX) { if (q - DMAX < dim) v = Auh[l * dim + q - DMAX]; }
    else if (q == 2 * DMAX) v = Ab[l];
    else if (q == 2 * DMAX + 1) v = Ac[l];
    tab[i] = v;
  }
  tile_sync();
  for (int64_t c0 = (int64_t)blockIdx.x * 64; c0 < batch; c0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
    T* mx = tx + lane * P;
    const T* mg = tg + lane * P;
    T* tm = tsave + lane * NLP;
    T z[DMAX];
    T gin[DX > 0 ? DX : 1];
    if constexpr (DX > 0) {
      TinyCol<T, DX> t{}, g{};
      if (lane < ncols) { t = *reinterpret_cast<const TinyCol<T, DX>*>(x + (c0 + lane) * DX); g = *reinterpret_cast<const TinyCol<T, DX>*>(ybar + (c0 + lane) * DX); }
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] = r < DX ? t.v[r < DX ? r : 0] : T(0);
#pragma unroll
      for (int r = 0; r < DX; ++r) gin[r] = g.v[r];
    } else {
      tile_stage_in<T, V>(tx, x + c0 * dim, dim, P, ncols, lane);
      tile_stage_in<T, V>(tg, ybar + c0 * dim, dim, P, ncols, lane);
      tile_sync();
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] = r < dim ? mx[r] : T(0);
    }
    auto dot = [&](const T* row) -> T {
      T s0 = T(0), s1 = T(0);
#pragma unroll
      for (int r = 0; r < DMAX; r += 2) { s0 += row[r] * z[r]; s1 += row[r + 1] * z[r + 1]; }
      return s0 + s1;
    };
    auto axpy = [&](const T* row, T a) {
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] += row[r] * a;
    };
    if (!INV) {
      for (int l = 0; l < n_layers; ++l) {
        const T* tl = tab + l * LW;
        const T t = walk_tanh(dot(tl) + tl[2 * DMAX]);
        tm[l] = t;
        axpy(tl + DMAX, t);
      }
    } else {
      for (int l = n_layers - 1; l >= 0; --l) {
        const T* tl = tab + l * LW;
        T t;
        if constexpr (sizeof(T) == 8) { T s2u; planar_inv_act<T>(dot(tl), tl[2 * DMAX + 1], tl[2 * DMAX], t, s2u); }      // find_alpha_act64, not the safeguarded Float64 loop
        else { T ldu; find_alpha_act(dot(tl), tl[2 * DMAX + 1], tl[2 * DMAX], t, ldu); }
        tm[l] = t;
        axpy(tl + DMAX, -t);
      }
    }
    if constexpr (DX > 0) {
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] = r < DX ? gin[r < DX ? r : 0] : T(0);
    } else {
#pragma unroll
      for (int r = 0; r < DMAX; ++r) z[r] = r < dim ? mg[r] : T(0);
    }
    const int64_t col = c0 + lane;
    const T lb = (lbar && lane < ncols) ? lbar[col] : T(0);
    if (!INV) {
      for (int l = n_layers - 1; l >= 0; --l) {
        const T* tl = tab + l * LW;
        const T t = tm[l], c = tl[2 * DMAX + 1];
        const T q = T(1) - t * t;
        const T sb = dot(tl + DMAX) * q + lb * c * (T(-2) * t) * q / (T(1) + c * q);
        if (s_out) ssave[lane * NLP + l] = sb;
        axpy(tl, sb);
      }
    } else {
      for (int l = 0; l < n_layers; ++l) {
        const T* tl = tab + l * LW;
        const T t = tm[l], c = tl[2 * DMAX + 1];
        const T q = T(1) - t * t;
        const T den = T(1) + c * q;
        const T sb = q / den * (-dot(tl + DMAX) + lb * T(2) * c * t / den);
        axpy(tl, sb);
      }
    }
    if constexpr (DX > 0) {
      if (lane < ncols) {
        TinyCol<T, DX> t;
#pragma unroll
        for (int r = 0; r < DX; ++r) t.v[r] = z[r];
        *reinterpret_cast<TinyCol<T, DX>*>(xbar + (c0 + lane) * DX) = t;
      }
      tile_sync();
    } else {
#pragma unroll
      for (int r = 0; r < DMAX; ++r) if (r < dim) mx[r] = z[r];
      tile_sync();
      tile_stage_out<T, V>(tx, xbar + c0 * dim, dim, P, ncols, lane);
    }
    if (s_out) {
      tile_stage_out<T, V>(ssave, s_out + c0 * n_layers, n_layers, NLP, ncols, lane);
      tile_stage_out<T, V>(tsave, t_out + c0 * n_layers, n_layers, NLP, ncols, lane);
    }
    tile_sync();
  }
}

// radial_layer.jl:43-72 (forward) and :88-129 (inverse), same arithmetic as radial_kernel
template <class T, int DMAX, bool INV, int V, int DX = 0>     // DX > 0: see planar_walk_kernel
__global__ __launch_bounds__(64) void radial_walk_kernel(const T* __restrict__ Aalpha, const T* __restrict__ Abeta, const T* __restrict__ Az0, const T* __restrict__ x,
                                                         T* __restrict__ y, T* __restrict__ ladj_ps, int dim, int P, int64_t batch, int accumulate, double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[1];
  T* tile = reinterpret_cast<T*>(smem);
  const int lane = threadIdx.x;
  const T alpha = d_log1pexp(Aalpha[0]);          // :44
  const T apb = d_log1pexp(Abeta[0]);              // α + β̂
  const T beta_hat = -alpha + apb;                  // :45
  double acc = 0.0;
  for (int64_t c0 = (int64_t)blockIdx.x * 64; c0 < batch; c0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
    T* mine = tile + lane * P;
    T xin[DX > 0 ? DX : 1];
    T dz[DMAX];
    T ss = T(0);
    if constexpr (DX > 0) {
      TinyCol<T, DX> t{};
      if (lane < ncols) t = *reinterpret_cast<const TinyCol<T, DX>*>(x + (c0 + lane) * DX);
#pragma unroll
      for (int r = 0; r < DX; ++r) xin[r] = t.v[r];
#pragma unroll
      for (int r = 0; r < DMAX; ++r) { dz[r] = r < DX ? xin[r < DX ? r : 0] - Az0[r < DX ? r : 0] : T(0); ss += dz[r] * dz[r]; }
    } else {
      tile_stage_in<T, V>(tile, x + c0 * dim, dim, P, ncols, lane);
      tile_sync();
#pragma unroll
      for (int r = 0; r < DMAX; ++r) { dz[r] = r < dim ? mine[r] - Az0[r] : T(0); ss += dz[r] * dz[r]; }
    }
    T r_fwd, gain;
    if (!INV) {
      r_fwd = d_sqrt(ss);
      gain = T(1) + beta_hat / (alpha + r_fwd);
    } else {
      const T gam = d_sqrt(ss);                     // compute_r :124-129
      const T a = apb - gam;
      const T rr = (d_sqrt(a * a + 4 * alpha * gam) - a) / 2;
      gain = (alpha + rr) / (apb + rr);             // γ :96-101
      r_fwd = gain * gam;
    }
    const T h_ = T(1) / (alpha + r_fwd);
    T ld = T(dim - 1) * d_log(T(1) + beta_hat * h_) + d_log(T(1) + beta_hat * h_ + beta_hat * (-(h_ * h_)) * r_fwd);   // :68-70
    if (INV) ld = -ld;
    const T fwd_gain = beta_hat / (alpha + r_fwd);
    if constexpr (DX > 0) {
      if (lane < ncols) {
        TinyCol<T, DX> t;
#pragma unroll
        for (int r = 0; r < DX; ++r) t.v[r] = !INV ? xin[r] + fwd_gain * dz[r] : Az0[r] + gain * dz[r];   // :52 / :101
        *reinterpret_cast<TinyCol<T, DX>*>(y + (c0 + lane) * DX) = t;
      }
    } else {
#pragma unroll
      for (int r = 0; r < DMAX; ++r) {
        if (r < dim) {
          if (!INV) mine[r] = mine[r] + fwd_gain * dz[r];             // :52
          else mine[r] = Az0[r] + gain * dz[r];                      // :101
        }
      }
      tile_sync();
      tile_stage_out<T, V>(tile, y + c0 * dim, dim, P, ncols, lane);
      tile_sync();
    }
    if (lane < ncols) {
      if (ladj_ps) ladj_ps[c0 + lane] = accumulate ? ladj_ps[c0 + lane] + ld : ld;
      acc += (double)ld;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

// Pullback of the RadialLayer on odd column heights, one lane per column (the arithmetic of radial_vjp_kernel)
template <class T, int DMAX, bool INV, int V>
__global__ __launch_bounds__(64) void radial_vjp_walk_kernel(const T* __restrict__ Aalpha, const T* __restrict__ Abeta, const T* __restrict__ Az0, const T* __restrict__ x,
                                                             const T* __restrict__ gbar, const T* __restrict__ lbar, T* __restrict__ xbar, int dim, int P, int64_t batch,
                                                             T* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tx = reinterpret_cast<T*>(smem);
  T* tg = tx + (size_t)64 * P;
  const int lane = threadIdx.x;
  const T alpha = d_log1pexp(Aalpha[0]);
  const T apb = d_log1pexp(Abeta[0]);
  const T bh = -alpha + apb;
  for (int64_t c0 = (int64_t)blockIdx.x * 64; c0 < batch; c0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
    tile_stage_in<T, V>(tx, x + c0 * dim, dim, P, ncols, lane);
    tile_stage_in<T, V>(tg, gbar + c0 * dim, dim, P, ncols, lane);
    tile_sync();
    T* mx = tx + lane * P;
    const T* mg = tg + lane * P;
    T dz[DMAX], gv[DMAX];
    T ss = T(0), dg = T(0);
#pragma unroll
    for (int r = 0; r < DMAX; ++r) {
      dz[r] = r < dim ? mx[r] - Az0[r] : T(0);
      gv[r] = r < dim ? mg[r] : T(0);
      ss += dz[r] * dz[r];
      dg += dz[r] * gv[r];
    }
    T rr, gain = T(1);
    if (!INV) rr = d_sqrt(ss);
    else {
      const T gam = d_sqrt(ss);              // compute_r, radial_layer.jl:124-129
      const T aa = apb - gam;
      const T r0 = (d_sqrt(aa * aa + 4 * alpha * gam) - aa) / 2;
      gain = (alpha + r0) / (apb + r0);
      rr = gain * gam;
    }
    const T h = T(1) / (alpha + rr);
    const T a = T(1) + bh * h;
    const T rinv = rr > T(0) ? T(1) / rr : T(0);
    const T c = -bh * h * h * rinv;
    const T lr = T(dim - 1) * (-bh * h * h) / a + (T(-2) * bh * h * h + T(2) * bh * h * h * h * rr) / (T(1) + bh * h - bh * h * h * rr);
    const int64_t col = c0 + lane;
    const T lb = (lbar && lane < ncols) ? lbar[col] : T(0);
    const T kl = lb * lr * rinv;
    if (!INV && work && lane < ncols) { work[col] = rr; work[batch + col] = dg; }
    T ca, cd;
    if (!INV) { ca = a; cd = c * dg + kl; }
    else {
      const T dv = gain * dg - kl * rr * rr;
      ca = T(1) / a;
      cd = gain * (-kl / a - c * dv / (a * (a + c * rr * rr)));
    }
#pragma unroll
    for (int r = 0; r < DMAX; ++r) if (r < dim) mx[r] = ca * gv[r] + cd * dz[r];
    tile_sync();
    tile_stage_out<T, V>(tx, xbar + c0 * dim, dim, P, ncols, lane);
    tile_sync();
  }
}

struct FlowCfg { int V, G, R; int64_t grid; };
// packs per lane the lanes-per-column kernels are instantiated for (64 lanes x 8 packs: 2 048 rows Float32, 1 024 Float64).  Until
// round 5 they went to 32 packs — 48 of the 63 unrolled packs of every kernel family, most of this file's compile time — for heights
// the column-tile and block-per-column kernels now serve better.
constexpr int FLOW_R_MAX = 8;

// ------------------------------------------------------------------ Planar / Radial on columns of ANY height (round 5)
// The register kernels stop at 64 lanes x 32 packs (8 192 rows Float32, 4 096 Float64) and bjx_planar / bjx_radial refused taller
// columns (the reference has no limit: planar_layer.jl:73-80, radial_layer.jl:43-53).  Here ONE BLOCK owns a column and walks it
// in passes of 16-byte packs (element accesses when the height or a base is not pack-aligned); between passes the column lives in
// the OUTPUT array (a workspace when only the log-det is asked for) — a column of this height is a few hundred KiB, it stays in the
// XCD's L2 between the passes, so the HBM traffic is still one read and one write.
//   Planar: n_layers + 1 passes — pass l applies layer l-1's update (z += û tanh(·)) and accumulates w_lᵀz for layer l in the same walk.
//   Radial: 2 passes — ‖z - z₀‖², then the update.
// Blocks take columns blockIdx.x, + gridDim.x, ...; the block sums are reduced through LDS (two barriers per pass).
template <class T> __device__ __forceinline__ T block_sum_256(T v, T* red /*[5]*/) {
  v = group_sum<64>(v);
  __syncthreads();                                     // the previous use of red[] is over
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void planar_tall_kernel(const PlanarArgs<T> A, const T* __restrict__ x, T* __restrict__ y, T* __restrict__ ws, T* __restrict__ ladj_ps,
                                                          int64_t dim, int64_t batch, int accumulate, double* partials) {
  __shared__ T red[5];
  __shared__ double redd[4];
  const int64_t nv = dim / V;                          // V = 1 when the height or a base is not pack-aligned
  double acc = 0.0;
  for (int64_t col = blockIdx.x; col < batch; col += gridDim.x) {
    const T* xc = x + col * dim;
    T* zc = y ? y + col * dim : ws + (int64_t)blockIdx.x * dim;
    T ladj = T(0);
    T tt = T(0);                                       // ± tanh of the layer applied in this pass
    for (int li = 0; li <= A.n_layers; ++li) {
      const int l = INV ? A.n_layers - 1 - li : li;          // the layer whose wᵀz this pass accumulates
      const int lp = INV ? l + 1 : l - 1;                     // the layer whose update this pass applies
      const T* wl = li < A.n_layers ? A.w + (int64_t)l * dim : nullptr;
      const T* ul = li > 0 ? A.u_hat + (int64_t)lp * dim : nullptr;
      const T* src = li == 0 ? xc : zc;
      T s = T(0), q = T(0);
      for (int64_t v = threadIdx.x; v < nv; v += 256) {
        Pack<T, V> z = load_pack<T, V, false>(src + v * V);
        if (ul) {
          const Pack<T, V> u = load_pack<T, V, false>(ul + v * V);
#pragma unroll
          for (int j = 0; j < V; ++j) z.v[j] += u.v[j] * tt;
        }
        if (wl) {
          const Pack<T, V> w = load_pack<T, V, false>(wl + v * V);
#pragma unroll
          for (int j = 0; j < V; ++j) s += w.v[j] * z.v[j];
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) q += z.v[j] * z.v[j];
        }
        if (ul || li == 0) store_pack<T, V, false>(zc + v * V, z);
      }
      if (li < A.n_layers) {
        s = block_sum_256(s, red);                     // wᵀz (src/utils.jl:2); the barriers also order this pass's stores before the next pass's loads
        const T bl = A.b[l], c = A.wtu_hat[l];
        T t, s2;
        if (!INV) flow_tanh_sech2(s + bl, t, s2);
        else planar_inv_act<T>(s, c, bl, t, s2);
        const T ld = Fast<T>::log1p(c * s2);           // planar_layer.jl:107
        ladj += INV ? -ld : ld;
        tt = INV ? -t : t;
      } else if (accumulate & 2) {                     // BJX_BASE_STDNORMAL: + log N(out; 0, I)
        q = block_sum_256(q, red);
        ladj += T(-0.5) * q - (T)dim * T(0.91893853320467274178);
      }
    }
    if (threadIdx.x == 0) {
      if (ladj_ps) ladj_ps[col] = (accumulate & 1) ? ladj_ps[col] + ladj : ladj;
      acc += (double)ladj;
    }
    __syncthreads();                                   // the workspace column is reused by the block's next column
  }
  if (partials) block_publish_partial(acc, redd, partials);
}

template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void radial_tall_kernel(const RadialArgs<T> A, const T* __restrict__ x, T* __restrict__ y, T* __restrict__ ladj_ps, int64_t dim,
                                                          int64_t batch, int accumulate, double* partials) {
  __shared__ T red[5];
  __shared__ double redd[4];
  const T alpha = d_log1pexp(A.alpha_[0]);          // radial_layer.jl:44
  const T apb = d_log1pexp(A.beta[0]);              // α + β̂
  const T beta_hat = -alpha + apb;                  // :45
  const int64_t nv = dim / V;
  double acc = 0.0;
  for (int64_t col = blockIdx.x; col < batch; col += gridDim.x) {
    const T* xc = x + col * dim;
    T ss = T(0);
    for (int64_t v = threadIdx.x; v < nv; v += 256) {
      const Pack<T, V> z = load_pack<T, V, false>(xc + v * V), z0 = load_pack<T, V, false>(A.z0 + v * V);
#pragma unroll
      for (int j = 0; j < V; ++j) { const T dlt = z.v[j] - z0.v[j]; ss += dlt * dlt; }
    }
    ss = block_sum_256(ss, red);
    T r_fwd, gain;
    if (!INV) {
      r_fwd = d_sqrt(ss);
      gain = T(1) + beta_hat / (alpha + r_fwd);
    } else {
      const T gam = d_sqrt(ss);                     // compute_r :124-129
      const T a = apb - gam;
      const T rr = (d_sqrt(a * a + 4 * alpha * gam) - a) / 2;
      gain = (alpha + rr) / (apb + rr);             // γ :96-101
      r_fwd = gain * gam;
    }
    const T h_ = T(1) / (alpha + r_fwd);
    T ld = T(dim - 1) * d_log(T(1) + beta_hat * h_) + d_log(T(1) + beta_hat * h_ + beta_hat * (-(h_ * h_)) * r_fwd);   // :68-70
    if (INV) ld = -ld;
    const T fwd_gain = beta_hat / (alpha + r_fwd);
    if (y) {
      T* yc = y + col * dim;
      for (int64_t v = threadIdx.x; v < nv; v += 256) {
        const Pack<T, V> z = load_pack<T, V, false>(xc + v * V), z0 = load_pack<T, V, false>(A.z0 + v * V);
        Pack<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const T dlt = z.v[j] - z0.v[j];
          if (!INV) o.v[j] = z.v[j] + fwd_gain * dlt;                        // :52
          else o.v[j] = z0.v[j] + gain * dlt;                                // :101
        }
        store_pack<T, V, false>(yc + v * V, o);
      }
    }
    if (threadIdx.x == 0) {
      if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + ld : ld;
      acc += (double)ld;
    }
  }
  if (partials) block_publish_partial(acc, redd, partials);
}

// The input pullbacks on columns of any height (same mapping: one block per column, passes of 16-byte packs).
//   Planar: the primal sweep walks the column n_layers times (pass l applies layer l-1's update and accumulates w_lᵀz; the column
//   between passes lives in the block's workspace column, t_l in LDS), the reverse sweep n_layers + 1 times (pass i adds
//   w s̄ of the layer before and accumulates ûᵀz̄ of the next; between passes z̄ lives in the OUTPUT column).  Formulas as in
//   planar_vjp_kernel above.
template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void planar_vjp_tall_kernel(const PlanarArgs<T> A, const T* x, const T* ybar, const T* lbar, T* xbar, T* ws,
                                                              int64_t dim, int64_t batch, T* t_out, T* s_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tsave = reinterpret_cast<T*>(smem);               // [n_layers]
  __shared__ T red[5];
  const int64_t nv = dim / V;
  const int nl = A.n_layers;
  for (int64_t col = blockIdx.x; col < batch; col += gridDim.x) {
    const T* xc = x + col * dim;
    T* zc = ws + (int64_t)blockIdx.x * dim;
    T tt = T(0);
    for (int li = 0; li < nl; ++li) {
      const int l = INV ? nl - 1 - li : li;
      const int lp = INV ? l + 1 : l - 1;
      const T* wl = A.w + (int64_t)l * dim;
      const T* ul = li > 0 ? A.u_hat + (int64_t)lp * dim : nullptr;
      const T* src = li <= 1 ? xc : zc;                // pass 0 only reads; pass 1 reads x again and writes the workspace
      const bool keep = li + 1 < nl;                   // the column after the last layer is not needed: only the t_l are
      T s = T(0);
      for (int64_t v = threadIdx.x; v < nv; v += 256) {
        Pack<T, V> z = load_pack<T, V, false>(src + v * V);
        if (ul) {
          const Pack<T, V> u = load_pack<T, V, false>(ul + v * V);
#pragma unroll
          for (int j = 0; j < V; ++j) z.v[j] += u.v[j] * tt;
        }
        const Pack<T, V> w = load_pack<T, V, false>(wl + v * V);
#pragma unroll
        for (int j = 0; j < V; ++j) s += w.v[j] * z.v[j];
        if (ul && keep) store_pack<T, V, false>(zc + v * V, z);
      }
      s = block_sum_256(s, red);
      T t;
      if (!INV) { t = flow_tanh(s + A.b[l]); tt = t; }
      else { T s2u; planar_inv_act<T>(s, A.wtu_hat[l], A.b[l], t, s2u); tt = -t; }
      if (threadIdx.x == 0) tsave[l] = t;
    }
    __syncthreads();
    const T lb = lbar ? lbar[col] : T(0);
    const T* gc = ybar + col * dim;
    T* oc = xbar + col * dim;
    T sbp = T(0);
    int lprev = 0;
    for (int li = 0; li <= nl; ++li) {
      const int l = INV ? li : nl - 1 - li;
      const T* ul = li < nl ? A.u_hat + (int64_t)l * dim : nullptr;
      const T* wp = li > 0 ? A.w + (int64_t)lprev * dim : nullptr;
      const T* src = li <= 1 ? gc : oc;
      T d = T(0);
      for (int64_t v = threadIdx.x; v < nv; v += 256) {
        Pack<T, V> z = load_pack<T, V, false>(src + v * V);
        if (wp) {
          const Pack<T, V> w = load_pack<T, V, false>(wp + v * V);
#pragma unroll
          for (int j = 0; j < V; ++j) z.v[j] += w.v[j] * sbp;
        }
        if (ul) {
          const Pack<T, V> u = load_pack<T, V, false>(ul + v * V);
#pragma unroll
          for (int j = 0; j < V; ++j) d += u.v[j] * z.v[j];
        }
        if (wp) store_pack<T, V, false>(oc + v * V, z);
      }
      if (li < nl) {
        d = block_sum_256(d, red);
        const T t = tsave[l], c = A.wtu_hat[l];
        const T q = T(1) - t * t;
        if (!INV) {
          sbp = d * q + lb * c * (T(-2) * t) * q / (T(1) + c * q);
          if (s_out && threadIdx.x == 0) { s_out[col * nl + l] = sbp; t_out[col * nl + l] = t; }
        } else {
          const T den = T(1) + c * q;
          sbp = q / den * (-d + lb * T(2) * c * t / den);
        }
        lprev = l;
      }
    }
    __syncthreads();                                   // tsave and the workspace column are reused by the block's next column
  }
}

//   Radial: two passes — (‖δ‖², δᵀȳ), then z̄ = ca ȳ + cd δ (coefficients as in radial_vjp_kernel above).
template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void radial_vjp_tall_kernel(const RadialArgs<T> A, const T* x, const T* gbar, const T* lbar, T* xbar, int64_t dim,
                                                              int64_t batch, T* work) {
  __shared__ T red[5];
  const T alpha = d_log1pexp(A.alpha_[0]);
  const T apb = d_log1pexp(A.beta[0]);
  const T bh = -alpha + apb;
  const int64_t nv = dim / V;
  for (int64_t col = blockIdx.x; col < batch; col += gridDim.x) {
    const T* xc = x + col * dim;
    const T* gc = gbar + col * dim;
    T ss = T(0), dg = T(0);
    for (int64_t v = threadIdx.x; v < nv; v += 256) {
      const Pack<T, V> z = load_pack<T, V, false>(xc + v * V), z0 = load_pack<T, V, false>(A.z0 + v * V), g = load_pack<T, V, false>(gc + v * V);
#pragma unroll
      for (int j = 0; j < V; ++j) { const T dlt = z.v[j] - z0.v[j]; ss += dlt * dlt; dg += dlt * g.v[j]; }
    }
    ss = block_sum_256(ss, red);
    dg = block_sum_256(dg, red);
    T rr, gain = T(1);
    if (!INV) rr = d_sqrt(ss);
    else {
      const T gam = d_sqrt(ss);
      const T aa = apb - gam;
      const T r0 = (d_sqrt(aa * aa + 4 * alpha * gam) - aa) / 2;
      gain = (alpha + r0) / (apb + r0);
      rr = gain * gam;
    }
    const T h = T(1) / (alpha + rr);
    const T a = T(1) + bh * h;
    const T rinv = rr > T(0) ? T(1) / rr : T(0);
    const T c = -bh * h * h * rinv;
    const T lr = T(dim - 1) * (-bh * h * h) / a + (T(-2) * bh * h * h + T(2) * bh * h * h * h * rr) / (T(1) + bh * h - bh * h * h * rr);
    const T lb = lbar ? lbar[col] : T(0);
    const T kl = lb * lr * rinv;
    if (!INV && work && threadIdx.x == 0) { work[col] = rr; work[batch + col] = dg; }
    T ca, cd;
    if (!INV) { ca = a; cd = c * dg + kl; }
    else {
      const T dv = gain * dg - kl * rr * rr;
      ca = T(1) / a;
      cd = gain * (-kl / a - c * dv / (a * (a + c * rr * rr)));
    }
    T* oc = xbar + col * dim;
    for (int64_t v = threadIdx.x; v < nv; v += 256) {
      const Pack<T, V> z = load_pack<T, V, false>(xc + v * V), z0 = load_pack<T, V, false>(A.z0 + v * V), g = load_pack<T, V, false>(gc + v * V);
      Pack<T, V> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.v[j] = ca * g.v[j] + cd * (z.v[j] - z0.v[j]);
      store_pack<T, V, false>(oc + v * V, o);
    }
  }
}

template <class T> bool flow_cfg(const bjx_ctx* ctx, const void* x, const void* y, int64_t dim, int64_t batch, FlowCfg* c, bool allow_unal = false) {
  constexpr int VW = Vec16<T>::N;
  const bool v_ok = bjx_aligned16(x) && bjx_aligned16(y) && dim % VW == 0;
  c->V = v_ok ? VW : 1;
  int64_t packs = dim / c->V;
  // odd heights / element-aligned bases (kernels that take a partial last pack): 16-byte packs all the same
  static const int use_unal = getenv("BJX_FLOW_UNALIGNED") ? atoi(getenv("BJX_FLOW_UNALIGNED")) : 1;
  if (allow_unal && use_unal && !v_ok && dim >= 32) { c->V = VW; packs = (dim + VW - 1) / VW; }
  int G = 1;
  while (G < 64 && G < packs) G <<= 1;
  // prefer fewer lanes per column with 2 packs each when that keeps >= 16-lane groups (more ILP)
  int64_t need = (packs + G - 1) / G;
  int R = 1;
  while (R < need) R <<= 1;
  if (R > FLOW_R_MAX) return false;             // beyond: the column-tile kernels (bjx_flow_cols.hip) and the block-per-column ones
  c->G = G;
  c->R = R;
  (void)ctx;
  c->grid = (batch + (256 / G) - 1) / (256 / G);
  return c->grid < ((int64_t)1 << 31);
}

#define FLOW_SWITCH_R(KERNEL, TT, VV, INVV, ...)                                                                         \
  switch (c.R) {                                                                                                        \
    case 1: hipLaunchKernelGGL((KERNEL<TT, VV, 1, INVV>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, __VA_ARGS__); break;  \
    case 2: hipLaunchKernelGGL((KERNEL<TT, VV, 2, INVV>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, __VA_ARGS__); break;  \
    case 4: hipLaunchKernelGGL((KERNEL<TT, VV, 4, INVV>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, __VA_ARGS__); break;  \
    default: hipLaunchKernelGGL((KERNEL<TT, VV, 8, INVV>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, __VA_ARGS__); break;  \
  }

template <class T>
int planar_impl(bjx_ctx* ctx, int inverse, const T* w, const T* u, const T* b, int nl, const T* in, T* out, T* ladj_ps,
                double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  const size_t need = (((size_t)nl * dim + nl) * sizeof(T) + 255) / 256 * 256;
  T* u_hat = static_cast<T*>(ctx->scratch);
  size_t ws_off = 0;                                          // (tall columns, log-det only) where the column workspace starts in big_ws
  if (need > BJX_SCRATCH_BYTES) {
    // û of a stack this large (only the tall-column kernel gets here: 8 layers x 16 384 Float64 rows) lives in the grown workspace,
    // in front of the column workspace of a log-det-only call
    const size_t ws_cols = out ? 0 : (size_t)ctx->num_cu * 8 * dim * sizeof(T);
    { int rc = bjx_ensure_big_ws(ctx, need + ws_cols); if (rc) return rc; }
    u_hat = static_cast<T*>(ctx->big_ws);
    ws_off = need;
  }
  T* wtu = u_hat + (size_t)nl * dim;
  hipLaunchKernelGGL(planar_prep_kernel<T>, dim3(nl), dim3(256), 0, ctx->stream, w, u, dim, u_hat, wtu);
  BJX_CHECK_LAUNCH(ctx);
  if (batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  // low-dimensional stacks: one lane per column (planar_walk_kernel)
  static const int walk_max = getenv("BJX_FLOW_WALK_MAX") ? atoi(getenv("BJX_FLOW_WALK_MAX")) : 32;       // tuning switch (0: off)
  // (same-call A/B, 2^22 columns, 1 / 8 layers: dim 12-32 walker 62-69 % / 43-57 % vs 29-44 % / 23-34 % on the group kernels;
  //  dim 40-64 walker 56-60 % / 27-31 % vs 72-74 % / 55-71 %: the register tile wins once a column fills 10+ lanes)
  if (dim <= walk_max && dim <= 32 && (size_t)nl * 68 * sizeof(T) <= 32 * 1024) {
    constexpr int VW = Vec16<T>::N;
    static const int use_direct = getenv("BJX_PLANAR_WALK_DIRECT") ? atoi(getenv("BJX_PLANAR_WALK_DIRECT")) : 1;
    const bool direct = use_direct && dim <= 8;                      // short columns: no tile (DX = dim)
    const int P = direct ? 0 : (int)(dim | 1);
    const int dmax = dim <= 4 ? 4 : (dim <= 8 ? 8 : (dim <= 16 ? 16 : 32));
    const size_t smem_w = ((((size_t)64 * P + 3) / 4) * 4 + (size_t)nl * (2 * dmax + 4)) * sizeof(T);
    const int64_t tiles = (batch + 63) / 64;
    const int64_t cap = (int64_t)ctx->num_cu * 32;
    const int grid_w = (int)(tiles < cap ? tiles : cap);
    if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid_w); if (rc) return rc; }
    double* partials_w = ladj_sum ? ctx->partials : nullptr;
    const bool vec = bjx_aligned16(in) && (!out || bjx_aligned16(out));
    const int accum = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
    {
      BjxProf prof_(ctx);
#define PW(D_, I_, V_) hipLaunchKernelGGL((planar_walk_kernel<T, D_, I_, V_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, (const T*)w, (const T*)u_hat, (const T*)wtu, (const T*)b, nl, in, out, ladj_ps, (int)dim, P, batch, accum, partials_w)
#define PW_V(D_, I_) do { if (vec) PW(D_, I_, VW); else PW(D_, I_, 1); } while (0)
#define PW_D(I_) do { if (dim <= 4) PW_V(4, I_); else if (dim <= 8) PW_V(8, I_); else if (dim <= 16) PW_V(16, I_); else PW_V(32, I_); } while (0)
#define PWX(D_, X_) do { if (inverse) hipLaunchKernelGGL((planar_walk_kernel<T, D_, true, 1, X_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, (const T*)w, (const T*)u_hat, (const T*)wtu, (const T*)b, nl, in, out, ladj_ps, (int)dim, P, batch, accum, partials_w); \
                          else hipLaunchKernelGGL((planar_walk_kernel<T, D_, false, 1, X_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, (const T*)w, (const T*)u_hat, (const T*)wtu, (const T*)b, nl, in, out, ladj_ps, (int)dim, P, batch, accum, partials_w); } while (0)
      if (direct) {
        switch ((int)dim) {
          case 1: PWX(4, 1); break;
          case 2: PWX(4, 2); break;
          case 3: PWX(4, 3); break;
          case 4: PWX(4, 4); break;
          case 5: PWX(8, 5); break;
          case 6: PWX(8, 6); break;
          case 7: PWX(8, 7); break;
          default: PWX(8, 8); break;
        }
      }
      else if (inverse) PW_D(true); else PW_D(false);
#undef PWX
#undef PW_D
#undef PW_V
#undef PW
    }
    BJX_CHECK_LAUNCH(ctx);
    if (ladj_sum) return bjx_launch_finalize(ctx, grid_w, ladj_sum, 0.0, 0, 0.0, flags);
    return BJX_OK;
  }
  // register kernel (Float32, 16-byte packs, 20 <= dim <= 128): see planar_reg_kernel
  static const int use_reg = getenv("BJX_PLANAR_REG") ? atoi(getenv("BJX_PLANAR_REG")) : 1;
  if constexpr (sizeof(T) == 4) {
    // columns that are not whole aligned packs (odd heights, or a base that is only 4-byte aligned) run the same kernels on
    // element-aligned packs (reg_load_pack).  They used to fall to the LDS-tile kernel: 4-28 % of the HBM peak at 33-255 rows.
    static const int use_unal = getenv("BJX_PLANAR_REG_UNALIGNED") ? atoi(getenv("BJX_PLANAR_REG_UNALIGNED")) : 1;
    static const int unal_nt = 0;
    const bool packs_ok = dim % 4 == 0 && bjx_aligned16(in) && bjx_aligned16(out);
    // heights that are not a multiple of four, on the 16-byte grid of memory (reg_load_pack): the lanes cover up to dim + 3 rows
    const bool grid_ok = use_unal && dim > 32 && dim % 4 != 0 && bjx_aligned16(in) && bjx_aligned16(out);
    const int64_t de = packs_ok ? dim : dim + 3;
    // 256 < dim <= 1024, two layers or more: the tile split over 8 / 16 waves of one block (planar_reg2_kernel, NW = 8 / 16)
    static const int use_big = getenv("BJX_PLANAR_REG_BIG") ? atoi(getenv("BJX_PLANAR_REG_BIG")) : 1;
    const bool big = use_big && de > 256 && de <= 1024 && nl >= 2;
    if (use_reg && (packs_ok || grid_ok) && dim > 16 && (de <= 256 || big)) {
      const int NL = (nl >= 8 && !big) ? 8 : (nl > 2 ? 4 : nl);           // 8 / 16 waves a block: groups of four layers (118 VGPRs: two 512-thread blocks a CU)
      const int nl_pad = (nl + NL - 1) / NL * NL;
      const int lead = packs_ok ? 0 : 4;
      const int64_t ldw = (dim + 3) / 4 * 4 + 2 * lead;
      const size_t off0 = ((size_t)nl * dim + nl + 3) / 4 * 4;   // floats, keeps the padded tables 16-byte aligned
      const size_t need_reg = (off0 + (size_t)2 * nl_pad * ldw + (size_t)nl_pad * nl_pad + 2 * (size_t)nl_pad) * sizeof(float);
      if (need_reg <= BJX_SCRATCH_BYTES) {
        float* base = reinterpret_cast<float*>(ctx->scratch);
        float* wp = base + off0;
        float* up = wp + (size_t)nl_pad * ldw;
        float* Gp = up + (size_t)nl_pad * ldw;
        float* cp = Gp + (size_t)nl_pad * nl_pad;
        float* bp = cp + nl_pad;
        hipLaunchKernelGGL(planar_prep_reg_kernel<float>, dim3(nl_pad * nl_pad), dim3(256), 0, ctx->stream, (const float*)w, (const float*)u_hat,
                           (const float*)wtu, (const float*)b, dim, nl, nl_pad, wp, up, Gp, cp, bp, ldw, lead);
        BJX_CHECK_LAUNCH(ctx);
        static const int cols_env = 0;
        const int G = de > 64 ? 32 : (de > 32 ? 16 : 8);
        const int cols = (G == 32 && (cols_env ? cols_env == 32 : PLANAR_REG_DEFAULT_COLS == 32)) ? 32 : 64;
        // two waves per tile for 64 < dim <= 128.  Measured (A/B in one run, 2^22 columns, d = 128): 8 layers forward
        // 0.759 vs 0.783 ms; 1 layer 0.74 vs 0.71 ms and the inverse 0.66 vs 0.62 ms are SLOWER (the barrier and the
        // redundant recurrence cost more than the third wave per SIMD buys: 147 VGPRs, and forcing 128 spills) —
        // so only deep forward stacks take it.  BJX_PLANAR_SPLIT = 0 / 1 forces it off / on.
        static const int split_env = getenv("BJX_PLANAR_SPLIT") ? atoi(getenv("BJX_PLANAR_SPLIT")) : -1;
        const bool quad = de > 128;                                  // four waves per tile (128 < dim <= 256)
        if (big) {
          const int64_t gridb = (batch + 63) / 64;
          BJX_REQUIRE(ctx, gridb < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar: batch too large for one launch");
          BjxFin finb;
          bool secondb = false;
          { int rc = bjx_make_fin(ctx, gridb, ladj_sum, 0.0, 0, flags, &finb, &secondb); if (rc) return rc; }
          // blocks of 8 / 16 waves: the ARRIVAL-TICKET epilogue (mode 1) is written for 64- and 256-thread blocks -> two-pass there.  Under the default
          // (mode 2) these 512 / 1024-thread blocks DO take the sentinel hand-off: block_publish_sentinel folds any number of waves through red[NWB]
          // (covered by test_finalize_modes_give_the_same_bits[planar_big_blocks])
          if (finb.counter) { finb.counter = nullptr; secondb = true; }
          PlanarRegArgs RB{wp, up, Gp, cp, bp, nl_pad, nl, (int)ldw, packs_ok ? 0 : (unal_nt ? 2 : 1), lead};
          const int accumb = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
#define LAUNCH_BIG_U(NL_, INV_, NW_, U_) hipLaunchKernelGGL((planar_reg2_kernel<NL_, INV_, NW_, U_>), dim3((unsigned)gridb), dim3(NW_ * 64), 0, ctx->stream, RB, (const float*)in, (float*)out, (float*)ladj_ps, (int)dim, batch, accumb, finb)
#define LAUNCH_BIG(NL_, INV_, NW_) do { if (packs_ok) LAUNCH_BIG_U(NL_, INV_, NW_, false); else LAUNCH_BIG_U(NL_, INV_, NW_, true); } while (0)
#define LAUNCH_BIG_I(NL_, NW_) do { if (inverse) LAUNCH_BIG(NL_, true, NW_); else LAUNCH_BIG(NL_, false, NW_); } while (0)
          { BjxProf prof_(ctx);
            if (de <= 512) { if (NL == 4) LAUNCH_BIG_I(4, 8); else LAUNCH_BIG_I(2, 8); }
            else { if (NL == 4) LAUNCH_BIG_I(4, 16); else LAUNCH_BIG_I(2, 16); } }
#undef LAUNCH_BIG_I
#undef LAUNCH_BIG
#undef LAUNCH_BIG_U
          BJX_CHECK_LAUNCH(ctx);
          if (secondb) return bjx_launch_finalize(ctx, (int)gridb, ladj_sum, 0.0, 0, 0.0, flags);
          return BJX_OK;
        }
        // (the aligned grid needs 16-lane groups — four columns per wave instruction: past 64 rows always the split tile)
        const bool split = quad || (G == 32 && (!packs_ok || (split_env >= 0 ? split_env != 0 : (!inverse && nl >= 8))));
        const int64_t grid = quad ? (batch + 63) / 64 : (split ? (batch + 2 * 64 - 1) / (2 * 64) : (batch + 4 * cols - 1) / (4 * cols));
        BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar: batch too large for one launch");
        BjxFin fin;
        bool second = false;
        { int rc = bjx_make_fin(ctx, grid, ladj_sum, 0.0, 0, flags, &fin, &second); if (rc) return rc; }
        PlanarRegArgs RA{wp, up, Gp, cp, bp, nl_pad, nl, (int)ldw, packs_ok ? 0 : (unal_nt ? 2 : 1), lead};
        const int accum = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
#define LAUNCH_REG_U(G_, NL_, INV_, U_) if (G_ == 32 && cols == 32) hipLaunchKernelGGL((planar_reg_kernel<G_, NL_, INV_, (G_ == 32 ? 32 : 64), (G_ == 16) && U_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, RA, (const float*)in, (float*)out, (float*)ladj_ps, (int)dim, batch, accum, fin); else hipLaunchKernelGGL((planar_reg_kernel<G_, NL_, INV_, 64, (G_ == 16) && U_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, RA, (const float*)in, (float*)out, (float*)ladj_ps, (int)dim, batch, accum, fin)
#define LAUNCH_REG(G_, NL_, INV_) do { if (packs_ok) { LAUNCH_REG_U(G_, NL_, INV_, false); } else { LAUNCH_REG_U(G_, NL_, INV_, true); } } while (0)
#define LAUNCH_REG_NL(G_, INV_) switch (NL) { case 1: LAUNCH_REG(G_, 1, INV_); break; case 2: LAUNCH_REG(G_, 2, INV_); break; case 4: LAUNCH_REG(G_, 4, INV_); break; default: LAUNCH_REG(G_, 8, INV_); break; }
#define LAUNCH_REG_G(INV_) switch (G) { case 8: LAUNCH_REG_NL(8, INV_) break; case 16: LAUNCH_REG_NL(16, INV_) break; default: LAUNCH_REG_NL(32, INV_) break; }
#define LAUNCH_REG2_U(NL_, INV_, U_) do { if (quad) hipLaunchKernelGGL((planar_reg2_kernel<NL_, INV_, 4, U_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, RA, (const float*)in, (float*)out, (float*)ladj_ps, (int)dim, batch, accum, fin); \
        else hipLaunchKernelGGL((planar_reg2_kernel<NL_, INV_, 2, U_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, RA, (const float*)in, (float*)out, (float*)ladj_ps, (int)dim, batch, accum, fin); } while (0)
#define LAUNCH_REG2(NL_, INV_) do { if (packs_ok) LAUNCH_REG2_U(NL_, INV_, false); else LAUNCH_REG2_U(NL_, INV_, true); } while (0)
#define LAUNCH_REG2_NL(INV_) switch (NL) { case 1: LAUNCH_REG2(1, INV_); break; case 2: LAUNCH_REG2(2, INV_); break; case 4: LAUNCH_REG2(4, INV_); break; default: LAUNCH_REG2(8, INV_); break; }
        // matrix-core kernel (forward, groups of 8 layers, dim a multiple of 16, no fused base density):
        // BJX_PLANAR_MFMA = 0 off | 1 direct loads, 64 columns per wave | 2 LDS-staged, 64 | 3 direct, 32 | 4 staged, 32 | 5 direct, 16 | 6 staged, 16
        static const int mfma_env = getenv("BJX_PLANAR_MFMA") ? atoi(getenv("BJX_PLANAR_MFMA")) : PLANAR_MFMA_DEFAULT;
        if (mfma_env && !inverse && NL == 8 && dim % 16 == 0 && dim >= 32 && dim <= 128 && !(flags & BJX_BASE_STDNORMAL)) {
          const int tiles = mfma_env >= 5 ? 1 : (mfma_env >= 3 ? 2 : 4), stage = (mfma_env % 2 == 0) ? 1 : 0;
          const int64_t gridm = (batch + 4 * 16 * tiles - 1) / (4 * 16 * tiles);
          BJX_REQUIRE(ctx, gridm < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar: batch too large for one launch");
          BjxFin finm;
          bool secondm = false;
          { int rc = bjx_make_fin(ctx, gridm, ladj_sum, 0.0, 0, flags, &finm, &secondm); if (rc) return rc; }
#define LAUNCH_MF(NB_, T_, S_) hipLaunchKernelGGL((planar_mfma_kernel<NB_, T_, S_>), dim3((unsigned)gridm), dim3(256), 0, ctx->stream, RA, (const float*)in, (float*)out, (float*)ladj_ps, (int)dim, batch, accum, finm)
#define LAUNCH_MF_TS(NB_) do { if (tiles == 4) { if (stage) LAUNCH_MF(NB_, 4, 1); else LAUNCH_MF(NB_, 4, 0); } else if (tiles == 2) { if (stage) LAUNCH_MF(NB_, 2, 1); else LAUNCH_MF(NB_, 2, 0); } \
                                else { if (stage) LAUNCH_MF(NB_, 1, 1); else LAUNCH_MF(NB_, 1, 0); } } while (0)
          { BjxProf prof_(ctx);
            switch (dim / 16) { case 2: LAUNCH_MF_TS(2); break; case 3: LAUNCH_MF_TS(3); break; case 4: LAUNCH_MF_TS(4); break; case 5: LAUNCH_MF_TS(5); break;
                                case 6: LAUNCH_MF_TS(6); break; case 7: LAUNCH_MF_TS(7); break; default: LAUNCH_MF_TS(8); break; } }
#undef LAUNCH_MF_TS
#undef LAUNCH_MF
          BJX_CHECK_LAUNCH(ctx);
          if (secondm) return bjx_launch_finalize(ctx, (int)gridm, ladj_sum, 0.0, 0, 0.0, flags);
          return BJX_OK;
        }
        { BjxProf prof_(ctx);
          if (split) { if (inverse) { LAUNCH_REG2_NL(true) } else { LAUNCH_REG2_NL(false) } }
          else if (inverse) { LAUNCH_REG_G(true) } else { LAUNCH_REG_G(false) } }
#undef LAUNCH_REG2_NL
#undef LAUNCH_REG2
#undef LAUNCH_REG2_U
#undef LAUNCH_REG_G
#undef LAUNCH_REG_NL
#undef LAUNCH_REG
#undef LAUNCH_REG_U
        BJX_CHECK_LAUNCH(ctx);
        if (second) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
        return BJX_OK;
      }
    }
  }
  // Float64: matrix-core kernel (groups of 8 layers, dim a multiple of 16 up to 128)
  if constexpr (sizeof(T) == 8) {
    static const int use_mf64 = getenv("BJX_PLANAR_MFMA64") ? atoi(getenv("BJX_PLANAR_MFMA64")) : 2;    // 0 off | 1: 16 columns per wave | 2: 32
    if (use_mf64 && dim % 16 == 0 && dim >= 16 && dim <= 128 && bjx_aligned16(in) && (!out || bjx_aligned16(out))) {
      const int nl_pad = (nl + 7) / 8 * 8;
      const size_t off0 = ((size_t)nl * dim + nl + 1) / 2 * 2;
      const size_t need_reg = (off0 + (size_t)2 * nl_pad * dim + (size_t)nl_pad * nl_pad + 2 * (size_t)nl_pad) * sizeof(double);
      if (need_reg <= BJX_SCRATCH_BYTES) {
        double* base = reinterpret_cast<double*>(ctx->scratch);
        double* wp = base + off0;
        double* up = wp + (size_t)nl_pad * dim;
        double* Gp = up + (size_t)nl_pad * dim;
        double* cp = Gp + (size_t)nl_pad * nl_pad;
        double* bp = cp + nl_pad;
        hipLaunchKernelGGL(planar_prep_reg_kernel<double>, dim3(nl_pad * nl_pad), dim3(256), 0, ctx->stream, (const double*)w, (const double*)u_hat,
                           (const double*)wtu, (const double*)b, dim, nl, nl_pad, wp, up, Gp, cp, bp);
        BJX_CHECK_LAUNCH(ctx);
        // (round 5: 16 columns per wave staged in two 64-row chunks — SPLIT = 2, four waves per SIMD — measured 0.478 of the HBM peak
        //  against 0.542 for this form and 0.498 for TILES = 1 unsplit, profiles/r05_c4f64_modes.md: not dispatched)
        constexpr bool split2 = false;
        const int tiles = use_mf64 == 1 ? 1 : 2;
        const int cols = 16 * tiles;
        const int64_t gridm = (batch + 4 * cols - 1) / (4 * cols);
        BJX_REQUIRE(ctx, gridm < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar: batch too large for one launch");
        BjxFin finm;
        bool secondm = false;
        { int rc = bjx_make_fin(ctx, gridm, ladj_sum, 0.0, 0, flags, &finm, &secondm); if (rc) return rc; }
        // 28 us blocks at two per CU: a group-closing block that waits holds half a CU — the sentinel hand-off measured +0.9 % here
        // (LAB_NOTEBOOK.md "Round 5"): the follow-up launch stays
        { int rc = bjx_fin_two_pass(ctx, gridm, &finm, &secondm); if (rc) return rc; }
        const int accum = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
        const size_t smem = (size_t)4 * (16 * (dim / (split2 ? 2 : 1) + 4) + cols * 8) * sizeof(double);
#define LAUNCH_MF64_S(NB_, T_, I_, S_) do { bjx_allow_big_lds(planar_mfma64_kernel<NB_, T_, I_, S_>, smem); \
          hipLaunchKernelGGL((planar_mfma64_kernel<NB_, T_, I_, S_>), dim3((unsigned)gridm), dim3(256), smem, ctx->stream, wp, up, Gp, cp, bp, nl_pad, nl, \
                             (const double*)in, (double*)out, (double*)ladj_ps, (int)dim, batch, accum, finm); } while (0)
#define LAUNCH_MF64(NB_, T_, I_) LAUNCH_MF64_S(NB_, T_, I_, 1)
#define LAUNCH_MF64_TI(NB_) do { if (tiles == 1) { if (inverse) LAUNCH_MF64(NB_, 1, true); else LAUNCH_MF64(NB_, 1, false); } \
                                 else { if (inverse) LAUNCH_MF64(NB_, 2, true); else LAUNCH_MF64(NB_, 2, false); } } while (0)
        { BjxProf prof_(ctx);
          switch (dim / 16) { case 1: LAUNCH_MF64_TI(1); break; case 2: LAUNCH_MF64_TI(2); break; case 3: LAUNCH_MF64_TI(3); break; case 4: LAUNCH_MF64_TI(4); break;
                              case 5: LAUNCH_MF64_TI(5); break; case 6: LAUNCH_MF64_TI(6); break; case 7: LAUNCH_MF64_TI(7); break; default: LAUNCH_MF64_TI(8); break; } }
#undef LAUNCH_MF64_TI
#undef LAUNCH_MF64
#undef LAUNCH_MF64_S
        BJX_CHECK_LAUNCH(ctx);
        if (secondm) return bjx_launch_finalize(ctx, (int)gridm, ladj_sum, 0.0, 0, 0.0, flags);
        return BJX_OK;
      }
    }
  }
  auto try_cols = [&]() -> int { return bjx::planar_cols_launch<T>(ctx, inverse, w, u_hat, wtu, b, nl, in, out, ladj_ps, ladj_sum, dim, batch, flags); };   // 1 = not served (bjx_flow_cols.hip)
  // Float64 beyond 64 rows: the column-tile kernel before the LDS tile kernel (100 rows: 20 % on the tile kernel)
  static const int tile_max_f64 = getenv("BJX_PLANAR_TILE_MAX_F64") ? atoi(getenv("BJX_PLANAR_TILE_MAX_F64")) : 64;
  if (sizeof(T) == 8 && dim > tile_max_f64) { const int rc_cols = try_cols(); if (rc_cols != 1) return rc_cols; }
  // tile kernel: lane = column (full-lane scalar recurrence); needs the 64 x dim tile in LDS
  static const int use_tile = getenv("BJX_PLANAR_TILE") ? atoi(getenv("BJX_PLANAR_TILE")) : 1;
  const size_t tile_bytes = (size_t)64 * dim * sizeof(T);
  const size_t need2 = need + ((size_t)nl * nl + 2 * (size_t)nl * dim) * sizeof(T);
  if (use_tile && tile_bytes <= 64 * 1024 && need2 <= BJX_SCRATCH_BYTES && dim < (1 << 20)) {
    T* G = wtu + nl;
    T* wT = G + (size_t)nl * nl;
    T* uT = wT + (size_t)nl * dim;
    hipLaunchKernelGGL(planar_prep2_kernel<T>, dim3(nl * nl), dim3(256), 0, ctx->stream, w, u_hat, dim, nl, G, wT, uT);
    BJX_CHECK_LAUNCH(ctx);
    const int64_t grid = (batch + 63) / 64;
    BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar: batch too large for one launch");
    if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
    double* partials = ladj_sum ? ctx->partials : nullptr;
    PlanarTileArgs<T> TA{wT, uT, G, wtu, b, nl};
    const int accum = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
    constexpr int VW = Vec16<T>::N;
    const bool v_ok = bjx_aligned16(in) && bjx_aligned16(out) && dim % VW == 0;
#define LAUNCH_TILE(V_, INV_) hipLaunchKernelGGL((planar_tile_kernel<T, V_, INV_>), dim3((unsigned)grid), dim3(64), tile_bytes, ctx->stream, TA, in, out, ladj_ps, (int)dim, batch, accum, partials)
    { BjxProf prof_(ctx);
    if (v_ok) { if (inverse) LAUNCH_TILE(VW, true); else LAUNCH_TILE(VW, false); }
    else { if (inverse) LAUNCH_TILE(1, true); else LAUNCH_TILE(1, false); } }
#undef LAUNCH_TILE
    BJX_CHECK_LAUNCH(ctx);
    if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
    return BJX_OK;
  }
  { const int rc_cols = try_cols(); if (rc_cols != 1) return rc_cols; }
  FlowCfg c;
  if (!flow_cfg<T>(ctx, in, out, dim, batch, &c, true)) {
    // columns taller than the register kernels hold: one block per column, n_layers + 1 passes (planar_tall_kernel)
    BJX_REQUIRE(ctx, batch < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar: batch too large for one launch");
    const int64_t capt = (int64_t)ctx->num_cu * 8;
    const int gridt = (int)(batch < capt ? batch : capt);
    T* ws = nullptr;
    if (!out) {
      if (ws_off == 0) { int rc = bjx_ensure_big_ws(ctx, (size_t)gridt * dim * sizeof(T)); if (rc) return rc; }
      ws = reinterpret_cast<T*>(static_cast<char*>(ctx->big_ws) + ws_off);
    }
    if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)gridt); if (rc) return rc; }
    double* partials_t = ladj_sum ? ctx->partials : nullptr;
    PlanarArgs<T> At{w, u_hat, wtu, b, nl, 0};
    const int accum_t = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
    constexpr int VWt = Vec16<T>::N;
    const bool v_ok = dim % VWt == 0 && bjx_aligned16(in) && (!out || bjx_aligned16(out)) && bjx_aligned16(u_hat) && bjx_aligned16(w) && ((size_t)dim * sizeof(T)) % 16 == 0;
    {
      BjxProf prof_(ctx);
#define LAUNCH_PT(V_, INV_) hipLaunchKernelGGL((planar_tall_kernel<T, V_, INV_>), dim3(gridt), dim3(256), 0, ctx->stream, At, in, out, ws, ladj_ps, dim, batch, accum_t, partials_t)
      if (v_ok) { if (inverse) LAUNCH_PT(VWt, true); else LAUNCH_PT(VWt, false); }
      else { if (inverse) LAUNCH_PT(1, true); else LAUNCH_PT(1, false); }
#undef LAUNCH_PT
    }
    BJX_CHECK_LAUNCH(ctx);
    if (ladj_sum) return bjx_launch_finalize(ctx, gridt, ladj_sum, 0.0, 0, 0.0, flags);
    return BJX_OK;
  }
  const size_t tab_bytes = (size_t)2 * nl * dim * sizeof(T);
  const bool lds = tab_bytes <= 60 * 1024;
  PlanarArgs<T> A{w, u_hat, wtu, b, nl, lds ? 1 : 0};
  const size_t smem = 32 + (lds ? tab_bytes : 0);
  const int accum = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)c.grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  constexpr int VW = Vec16<T>::N;
  BjxProf prof_(ctx);
  if (c.V == VW) {
    if (!inverse) { FLOW_SWITCH_R(planar_kernel, T, VW, false, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
    else { FLOW_SWITCH_R(planar_kernel, T, VW, true, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
  } else {
    if (!inverse) { FLOW_SWITCH_R(planar_kernel, T, 1, false, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
    else { FLOW_SWITCH_R(planar_kernel, T, 1, true, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
  }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)c.grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

// register-kernel fast path of the pullback (Float32, 16 < dim <= 128); returns 1 when the shape is not served
template <class T>
int planar_vjp_reg(bjx_ctx*, int, const T*, const T*, const T*, const T*, int, const T*, const T*, const T*, T*, int64_t, int64_t, T*, T*) { return 1; }
inline int planar_vjp_reg(bjx_ctx* ctx, int inverse, const float* w, const float* u_hat, const float* wtu, const float* b, int nl, const float* in,
                          const float* out_bar, const float* ladj_bar, float* in_bar, int64_t dim, int64_t batch, float* t_out, float* s_out) {
  return bjx::planar_vjp_reg_launch(ctx, inverse, w, u_hat, wtu, b, nl, in, out_bar, ladj_bar, in_bar, dim, batch, t_out, s_out);   // bjx_flow_vjp_reg.hip
}

template <class T>
int planar_vjp_impl(bjx_ctx* ctx, int inverse, const T* w, const T* u, const T* b, int nl, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar,
                    int64_t dim, int64_t batch, T* t_out = nullptr, T* s_out = nullptr) {
  const size_t need = (((size_t)nl * dim + nl) * sizeof(T) + 255) / 256 * 256;
  T* u_hat = static_cast<T*>(ctx->scratch);
  const int64_t capt = (int64_t)ctx->num_cu * 8;                       // blocks (= workspace columns) of the tall-column kernel
  const int gridt = (int)(batch < capt ? batch : capt);
  size_t ws_off = 0;
  if (need > BJX_SCRATCH_BYTES) {
    // û of a stack this large (only the tall-column kernel gets here) lives in the grown workspace, in front of the column workspace
    { int rc = bjx_ensure_big_ws(ctx, need + (size_t)gridt * dim * sizeof(T)); if (rc) return rc; }
    u_hat = static_cast<T*>(ctx->big_ws);
    ws_off = need;
  }
  T* wtu = u_hat + (size_t)nl * dim;
  hipLaunchKernelGGL(planar_prep_kernel<T>, dim3(nl), dim3(256), 0, ctx->stream, w, u, dim, u_hat, wtu);
  BJX_CHECK_LAUNCH(ctx);
  if (batch == 0) return BJX_OK;
  auto launch_tall = [&]() -> int {
    // columns taller than the register kernels hold: one block per column (planar_vjp_tall_kernel)
    BJX_REQUIRE(ctx, batch < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar_vjp: batch too large for one launch");
    BJX_REQUIRE(ctx, (size_t)nl * sizeof(T) <= 32 * 1024, BJX_ERR_UNSUPPORTED, "bjx_planar_vjp: too many layers (%d)", nl);
    if (ws_off == 0) { int rc = bjx_ensure_big_ws(ctx, (size_t)gridt * dim * sizeof(T)); if (rc) return rc; }
    T* ws = reinterpret_cast<T*>(static_cast<char*>(ctx->big_ws) + ws_off);
    PlanarArgs<T> At{w, u_hat, wtu, b, nl, 0};
    constexpr int VWt = Vec16<T>::N;
    const bool v_ok = dim % VWt == 0 && bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar) && bjx_aligned16(u_hat) && bjx_aligned16(w);
    const size_t smem_t = (size_t)nl * sizeof(T);
    {
      BjxProf prof_(ctx);
#define LAUNCH_PVT(V_, INV_) hipLaunchKernelGGL((planar_vjp_tall_kernel<T, V_, INV_>), dim3(gridt), dim3(256), smem_t, ctx->stream, At, in, out_bar, ladj_bar, in_bar, ws, dim, batch, t_out, s_out)
      if (v_ok) { if (inverse) LAUNCH_PVT(VWt, true); else LAUNCH_PVT(VWt, false); }
      else { if (inverse) LAUNCH_PVT(1, true); else LAUNCH_PVT(1, false); }
#undef LAUNCH_PVT
    }
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  };
  {
    // low-dimensional columns: one lane per column (planar_vjp_walk_kernel)
    static const int walk_max = getenv("BJX_FLOW_WALK_MAX") ? atoi(getenv("BJX_FLOW_WALK_MAX")) : 32;
    const int dmax = dim <= 4 ? 4 : (dim <= 8 ? 8 : (dim <= 16 ? 16 : 32));
    static const int use_direct = getenv("BJX_PLANAR_WALK_DIRECT") ? atoi(getenv("BJX_PLANAR_WALK_DIRECT")) : 1;
    const bool direct = use_direct && dim <= 8;                      // short columns: no data tiles (DX = dim)
    const int64_t P = direct ? 0 : (dim | 1), NLP = nl | 1;
    const size_t smem_w = ((size_t)2 * 64 * P + (size_t)64 * NLP + (((size_t)64 * NLP + 3) / 4) * 4 + (size_t)nl * (2 * dmax + 4)) * sizeof(T);
    if (dim <= walk_max && dim <= 32 && smem_w <= 60 * 1024 && (const void*)in != (const void*)in_bar) {
      constexpr int VW = Vec16<T>::N;
      const int64_t tiles = (batch + 63) / 64;
      const int64_t cap = (int64_t)ctx->num_cu * 32;
      const int grid_w = (int)(tiles < cap ? tiles : cap);
      const bool vec = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
      {
        BjxProf prof_(ctx);
#define PVW(D_, I_, V_) hipLaunchKernelGGL((planar_vjp_walk_kernel<T, D_, I_, V_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, (const T*)w, (const T*)u_hat, (const T*)wtu, (const T*)b, nl, \
                                           in, out_bar, ladj_bar, in_bar, (int)dim, (int)P, (int)NLP, batch, t_out, s_out)
#define PVW_V(D_, I_) do { if (vec) PVW(D_, I_, VW); else PVW(D_, I_, 1); } while (0)
#define PVW_D(I_) do { if (dim <= 4) PVW_V(4, I_); else if (dim <= 8) PVW_V(8, I_); else if (dim <= 16) PVW_V(16, I_); else PVW_V(32, I_); } while (0)
#define PVWX2(D_, X_, I_, V_) hipLaunchKernelGGL((planar_vjp_walk_kernel<T, D_, I_, V_, X_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, (const T*)w, (const T*)u_hat, (const T*)wtu, (const T*)b, nl, \
                                                 in, out_bar, ladj_bar, in_bar, (int)dim, (int)P, (int)NLP, batch, t_out, s_out)
        const bool vec_ts = !t_out || (bjx_aligned16(t_out) && bjx_aligned16(s_out));   // V only moves the per-layer t / s̄ tiles in this mode
#define PVWX(D_, X_) do { if (inverse) { if (vec_ts) PVWX2(D_, X_, true, VW); else PVWX2(D_, X_, true, 1); } else { if (vec_ts) PVWX2(D_, X_, false, VW); else PVWX2(D_, X_, false, 1); } } while (0)
        if (direct) {
          switch ((int)dim) {
            case 1: PVWX(4, 1); break;
            case 2: PVWX(4, 2); break;
            case 3: PVWX(4, 3); break;
            case 4: PVWX(4, 4); break;
            case 5: PVWX(8, 5); break;
            case 6: PVWX(8, 6); break;
            case 7: PVWX(8, 7); break;
            default: PVWX(8, 8); break;
          }
        }
        else if (inverse) PVW_D(true); else PVW_D(false);
#undef PVWX
#undef PVWX2
#undef PVW_D
#undef PVW_V
#undef PVW
      }
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
  }
  {
    int rc = planar_vjp_reg(ctx, inverse, w, u_hat, wtu, b, nl, in, out_bar, ladj_bar, in_bar, dim, batch, t_out, s_out);
    if (rc != 1) return rc;                               // 1 = shape not served by the register kernel
  }
  {
    // a block per C columns, the columns in registers (planar_vjp_cols_kernel, bjx_flow_cols.hip): beyond the Float32 register tiles, and Float64
    const int rc_cols = bjx::planar_vjp_cols_launch<T>(ctx, inverse, w, u_hat, wtu, b, nl, in, out_bar, ladj_bar, in_bar, dim, batch, t_out, s_out);
    if (rc_cols != 1) return rc_cols;
  }
  FlowCfg c;
  // (odd heights / element-aligned bases take 16-byte packs with a partial last pack, like the forward group kernel: the 4-byte
  //  form ran the pullback of eight layers at 201 rows at 16 % of the HBM peak)
  if (!flow_cfg<T>(ctx, in, in_bar, dim, batch, &c, true)) return launch_tall();
  const int cols_per_block = 256 / c.G;
  const size_t tsave_bytes = ((size_t)cols_per_block * nl * sizeof(T) + 15) / 16 * 16;
  const size_t tab_bytes = (size_t)2 * nl * dim * sizeof(T);
  const bool lds = tsave_bytes + tab_bytes <= 60 * 1024;
  BJX_REQUIRE(ctx, tsave_bytes <= 60 * 1024, BJX_ERR_UNSUPPORTED, "bjx_planar_vjp: too many layers (%d)", nl);
  PlanarArgs<T> A{w, u_hat, wtu, b, nl, lds ? 1 : 0};
  const size_t smem = (size_t)cols_per_block * nl * sizeof(T) + (lds ? tab_bytes : 0);
  constexpr int VW = Vec16<T>::N;
  const bool whole = dim % VW == 0 && bjx_aligned16(in) && bjx_aligned16(in_bar);
  const bool v_ok = c.V == VW && (!whole || bjx_aligned16(out_bar));      // element-aligned packs do not need an aligned cotangent either
  BjxProf prof_(ctx);
#define PVJ(V_, R_) do { if (inverse) hipLaunchKernelGGL((planar_vjp_kernel<T, V_, R_, true>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, t_out, s_out); \
                         else hipLaunchKernelGGL((planar_vjp_kernel<T, V_, R_, false>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, t_out, s_out); } while (0)
#define PVJ_R(V_) switch (c.R) { case 1: PVJ(V_, 1); break; case 2: PVJ(V_, 2); break; case 4: PVJ(V_, 4); break; default: PVJ(V_, 8); break; }
  if (v_ok) { PVJ_R(VW) } else {
    // scalar packs: recompute the geometry for V = 1
    int G = 1;
    while (G < 64 && G < dim) G <<= 1;
    int64_t need_r = (dim + G - 1) / G;
    int R = 1;
    while (R < need_r) R <<= 1;
    if (R > FLOW_R_MAX) return launch_tall();
    c.G = G; c.R = R; c.grid = (batch + (256 / G) - 1) / (256 / G);
    const int cpb = 256 / G;
    const size_t smem1 = (size_t)cpb * nl * sizeof(T) + (lds ? tab_bytes : 0);
#define PVJ1(R_) do { if (inverse) hipLaunchKernelGGL((planar_vjp_kernel<T, 1, R_, true>), dim3((unsigned)c.grid), dim3(256), smem1, ctx->stream, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, t_out, s_out); \
                      else hipLaunchKernelGGL((planar_vjp_kernel<T, 1, R_, false>), dim3((unsigned)c.grid), dim3(256), smem1, ctx->stream, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, t_out, s_out); } while (0)
    switch (c.R) { case 1: PVJ1(1); break; case 2: PVJ1(2); break; case 4: PVJ1(4); break; default: PVJ1(8); break; }
#undef PVJ1
  }
#undef PVJ_R
#undef PVJ
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T>
int radial_impl(bjx_ctx* ctx, int inverse, const T* alpha_, const T* beta, const T* z0, const T* in, T* out, T* ladj_ps,
                double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  static const int walk_max = getenv("BJX_FLOW_WALK_MAX") ? atoi(getenv("BJX_FLOW_WALK_MAX")) : 32;
  static const int walk_all = getenv("BJX_RADIAL_WALK_ALL") ? atoi(getenv("BJX_RADIAL_WALK_ALL")) : 0;
  static const int use_direct = getenv("BJX_PLANAR_WALK_DIRECT") ? atoi(getenv("BJX_PLANAR_WALK_DIRECT")) : 1;
  // (Float32 whole-pack columns stream at 56–73 % on the group kernel already; Float64 ones did not: 4 / 10 / 20 / 32 rows 38 / 27 / 26 / 41 %
  //  against 54 / 66 / 64 / 68 % one lane per column — round 5)
  if (dim <= walk_max && dim <= 32 && (dim % Vec16<T>::N != 0 || walk_all || sizeof(T) == 8)) {
    constexpr int VW = Vec16<T>::N;
    const bool direct = use_direct && dim <= 7;                      // short columns: no tile (DX = dim)
    const int P = direct ? 0 : (int)(dim | 1);
    const size_t smem_w = (size_t)64 * P * sizeof(T);
    const int64_t tiles = (batch + 63) / 64;
    const int64_t cap = (int64_t)ctx->num_cu * 32;
    const int grid_w = (int)(tiles < cap ? tiles : cap);
    if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid_w); if (rc) return rc; }
    double* partials_w = ladj_sum ? ctx->partials : nullptr;
    const bool vec = bjx_aligned16(in) && bjx_aligned16(out);
    const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
    {
      BjxProf prof_(ctx);
#define RW(D_, I_, V_) hipLaunchKernelGGL((radial_walk_kernel<T, D_, I_, V_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, alpha_, beta, z0, in, out, ladj_ps, (int)dim, P, batch, accum, partials_w)
#define RW_V(D_, I_) do { if (vec) RW(D_, I_, VW); else RW(D_, I_, 1); } while (0)
#define RW_D(I_) do { if (dim <= 4) RW_V(4, I_); else if (dim <= 8) RW_V(8, I_); else if (dim <= 16) RW_V(16, I_); else RW_V(32, I_); } while (0)
#define RWX(D_, X_) do { if (inverse) hipLaunchKernelGGL((radial_walk_kernel<T, D_, true, 1, X_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, alpha_, beta, z0, in, out, ladj_ps, (int)dim, P, batch, accum, partials_w); \
                          else hipLaunchKernelGGL((radial_walk_kernel<T, D_, false, 1, X_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, alpha_, beta, z0, in, out, ladj_ps, (int)dim, P, batch, accum, partials_w); } while (0)
      if (direct) {
        switch ((int)dim) {
          case 1: RWX(4, 1); break;
          case 2: RWX(4, 2); break;
          case 3: RWX(4, 3); break;
          case 4: RWX(4, 4); break;
          case 5: RWX(8, 5); break;
          case 6: RWX(8, 6); break;
          default: RWX(8, 7); break;
        }
      }
      else if (inverse) RW_D(true); else RW_D(false);
#undef RWX
#undef RW_D
#undef RW_V
#undef RW
    }
    BJX_CHECK_LAUNCH(ctx);
    if (ladj_sum) return bjx_launch_finalize(ctx, grid_w, ladj_sum, 0.0, 0, 0.0, flags);
    return BJX_OK;
  }
  FlowCfg c;
  if (!flow_cfg<T>(ctx, in, out, dim, batch, &c, true)) {
    // columns taller than the register kernels hold: one block per column, two passes (radial_tall_kernel)
    BJX_REQUIRE(ctx, batch < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_radial: batch too large for one launch");
    const int64_t capt = (int64_t)ctx->num_cu * 8;
    const int gridt = (int)(batch < capt ? batch : capt);
    if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)gridt); if (rc) return rc; }
    double* partials_t = ladj_sum ? ctx->partials : nullptr;
    RadialArgs<T> At{alpha_, beta, z0, 0};
    constexpr int VWt = Vec16<T>::N;
    const bool v_ok = dim % VWt == 0 && bjx_aligned16(in) && (!out || bjx_aligned16(out)) && bjx_aligned16(z0);
    {
      BjxProf prof_(ctx);
#define LAUNCH_RT(V_, INV_) hipLaunchKernelGGL((radial_tall_kernel<T, V_, INV_>), dim3(gridt), dim3(256), 0, ctx->stream, At, in, out, ladj_ps, dim, batch, (flags & BJX_ACCUMULATE) ? 1 : 0, partials_t)
      if (v_ok) { if (inverse) LAUNCH_RT(VWt, true); else LAUNCH_RT(VWt, false); }
      else { if (inverse) LAUNCH_RT(1, true); else LAUNCH_RT(1, false); }
#undef LAUNCH_RT
    }
    BJX_CHECK_LAUNCH(ctx);
    if (ladj_sum) return bjx_launch_finalize(ctx, gridt, ladj_sum, 0.0, 0, 0.0, flags);
    return BJX_OK;
  }
  {
    const int uc = c.R == 1 ? 4 : (c.R == 2 ? 2 : 1);          // RadialUC<R>
    const int64_t cpb = (int64_t)(256 / c.G) * uc;
    c.grid = (batch + cpb - 1) / cpb;
  }
  const size_t tab_bytes = (size_t)dim * sizeof(T);
  const bool lds = tab_bytes <= 60 * 1024;
  RadialArgs<T> A{alpha_, beta, z0, lds ? 1 : 0};
  const size_t smem = 32 + (lds ? tab_bytes : 0);
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)c.grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  constexpr int VW = Vec16<T>::N;
  BjxProf prof_(ctx);
  if (c.V == VW) {
    if (!inverse) { FLOW_SWITCH_R(radial_kernel, T, VW, false, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
    else { FLOW_SWITCH_R(radial_kernel, T, VW, true, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
  } else {
    if (!inverse) { FLOW_SWITCH_R(radial_kernel, T, 1, false, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
    else { FLOW_SWITCH_R(radial_kernel, T, 1, true, A, in, out, ladj_ps, dim, batch, c.G, accum, partials) }
  }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)c.grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_planar(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* w, const void* u, const void* b, int n_layers,
                       const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0 && n_layers >= 1, BJX_ERR_SHAPE, "bjx_planar: bad size (dim=%lld, n_layers=%d)", (long long)dim, n_layers);
  // out == NULL: only the log-det / log-density is wanted (BJX_BASE_STDNORMAL: logpdf(td, y) without storing the pre-image)
  BJX_REQUIRE(ctx, w && u && b && ((in && (out || ladj_ps || ladj_sum)) || batch == 0), BJX_ERR_ARG, "bjx_planar: null pointer");
  if (dt == BJX_F32) return planar_impl<float>(ctx, inverse, (const float*)w, (const float*)u, (const float*)b, n_layers, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return planar_impl<double>(ctx, inverse, (const double*)w, (const double*)u, (const double*)b, n_layers, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_planar: bad dtype %d", (int)dt);
}

namespace {
// ------------------------------------------------------------------ Planar PARAMETER pullback
// (w̄, ū, b̄) of the fused stack, summed over the batch.  With s̄_k (cotangent of s_k) and t_k = tanh s_k of every layer
// and column — emitted by the input-pullback kernels as [n_layers, batch] arrays — everything that couples the batch is
//   M1 = Z₀·S̄ᵀ  [dim, nl],   M2 = Ȳ·Tᵀ  [dim, nl],   ST[j][k] = Σ_n s̄_jn t_kn,   b̄_k = Σ_n s̄_kn,
//   c̄_k = Σ_n ℓ̄_n q_kn/(1 + c_k q_kn)
// because z_{k-1} = z₀ + Σ_{j<k} û_j t_j and z̄_k = ȳ + Σ_{j>k} w_j s̄_j:
//   w̄_k(direct) = M1[:,k] + Σ_{j<k} û_j ST[k][j],      û̄_k = M2[:,k] + Σ_{j>k} w_j ST[j][k].
// planar_param_reduce_kernel streams Z₀ and Ȳ once more (G lanes per column, NLG <= 8 layers per launch, the
// [rows of my pack] x [layers] accumulators in registers), combines the column groups of a block through LDS and
// writes one Float64 partial per block; planar_param_finalize_kernel sums the partials in a fixed order and applies
// the chain rule through get_u_hat (planar_layer.jl:65-70).
constexpr int PP_NLG = 8;
template <class T, int V, int R>
__global__ __launch_bounds__(256) void planar_param_reduce_kernel(const T* __restrict__ z0, const T* __restrict__ ybar, const T* __restrict__ sbar,
                                                                  const T* __restrict__ tt, const T* __restrict__ lbar, const T* __restrict__ wtu_hat,
                                                                  int64_t dim, int64_t batch, int G, int nl, int l0, int nlg, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  const int gl = threadIdx.x & (G - 1), cgp = threadIdx.x / G;
  const int cols_per_block = 256 / G;
  const int64_t nvc = (dim + V - 1) / V;                 // the last pack may be partial (odd heights: element-aligned packs; its dead rows read as zero)
  T m1[R][V][PP_NLG], m2[R][V][PP_NLG];
  T stg[PP_NLG];                 // lane gl < nlg: ST[l0+gl][l0 .. l0+nlg) restricted to this layer group's columns... (full row below)
  T bsum = T(0), csum = T(0);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
      for (int k = 0; k < PP_NLG; ++k) { m1[r][j][k] = T(0); m2[r][j][k] = T(0); }
#pragma unroll
  for (int k = 0; k < PP_NLG; ++k) stg[k] = T(0);
  const T cmine = (gl < nlg) ? wtu_hat[l0 + gl] : T(0);
  const int64_t stride = (int64_t)gridDim.x * cols_per_block;
  for (int64_t col = (int64_t)blockIdx.x * cols_per_block + cgp; col < batch; col += stride) {
    Pack<T, V> pz[R], pg[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
        const int nrow = (int)(dim - v * V < V ? dim - v * V : V);
        pz[r] = load_pack_part<T, V>(z0 + col * dim + v * V, nrow); pg[r] = load_pack_part<T, V>(ybar + col * dim + v * V, nrow);
      }
    }
    T sk[PP_NLG], tk[PP_NLG];
#pragma unroll
    for (int k = 0; k < PP_NLG; ++k) { sk[k] = k < nlg ? sbar[col * nl + l0 + k] : T(0); tk[k] = k < nlg ? tt[col * nl + l0 + k] : T(0); }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t v = gl + (int64_t)r * G;
      if (v < nvc) {
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
          for (int k = 0; k < PP_NLG; ++k) { m1[r][j][k] += pz[r].v[j] * sk[k]; m2[r][j][k] += pg[r].v[j] * tk[k]; }
      }
    }
    // Gram row / b̄ / c̄ of layer l0 + gl (lanes gl < nlg); the row of ST spans ALL layers
    if (gl < nlg) {
      const T sme = sbar[col * nl + l0 + gl], tme = tt[col * nl + l0 + gl];
      bsum += sme;
      const T q = T(1) - tme * tme;
      csum += (lbar ? lbar[col] : T(0)) * q / (T(1) + cmine * q);
#pragma unroll
      for (int k = 0; k < PP_NLG; ++k) stg[k] += sme * tk[k];
    }
  }
  // ---- combine the column groups of the block in a fixed order; layout of a block partial:
  //      [M1 dim*nlg][M2 dim*nlg][ST nlg*nlg (this group's diagonal block)][b nlg][c nlg]
  const size_t n_m = (size_t)dim * nlg;
  const size_t per = 2 * n_m + (size_t)nlg * nlg + 2 * (size_t)nlg;
  for (int pass = 0; pass < cols_per_block; ++pass) {
    if (cgp == pass) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t v = gl + (int64_t)r * G;
        if (v < nvc) {
#pragma unroll
          for (int j = 0; j < V; ++j)
            for (int k = 0; k < nlg && v * V + j < dim; ++k) {
              const size_t i1 = (size_t)(v * V + j) * nlg + k;
              if (pass == 0) { red[i1] = (double)m1[r][j][k]; red[n_m + i1] = (double)m2[r][j][k]; }
              else { red[i1] += (double)m1[r][j][k]; red[n_m + i1] += (double)m2[r][j][k]; }
            }
        }
      }
      if (gl < nlg) {
        for (int k = 0; k < nlg; ++k) {
          const size_t i2 = 2 * n_m + (size_t)gl * nlg + k;
          if (pass == 0) red[i2] = (double)stg[k]; else red[i2] += (double)stg[k];
        }
        const size_t ib = 2 * n_m + (size_t)nlg * nlg + gl;
        if (pass == 0) { red[ib] = (double)bsum; red[ib + nlg] = (double)csum; } else { red[ib] += (double)bsum; red[ib + nlg] += (double)csum; }
      }
    }
    __syncthreads();
  }
  for (size_t i = threadIdx.x; i < per; i += blockDim.x) partial[(size_t)blockIdx.x * per + i] = red[i];
}

// ---- the same reduction for LOW-DIMENSIONAL columns (dim <= 12), one lane per column: Z₀ and Ȳ of 64 consecutive columns through two
// odd-pitch LDS tiles, the lane's s̄ / tanh of the layer group from the work arrays, M1 / M2 accumulated in the lane's own strip of
// LDS ([2·dim·nlg][64]: bank = lane), the Gram block / b̄ / c̄ sums in registers; at the end the 64 strips are summed in a fixed
// order into ONE Float64 partial per block with the layout of planar_param_reduce_kernel.  (With G lanes per column a 2 … 10 row
// column keeps 1 - 3 lanes of every 8 busy: 2.5 - 12 % of the roofline.)
template <class T, int V>
__global__ __launch_bounds__(64) void planar_param_walk_kernel(const T* __restrict__ z0, const T* __restrict__ ybar, const T* __restrict__ sbar, const T* __restrict__ tt,
                                                               const T* __restrict__ lbar, const T* __restrict__ wtu_hat, int dim, int P, int64_t batch, int nl, int l0, int nlg,
                                                               double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tz = reinterpret_cast<T*>(smem);
  T* tg = tz + (size_t)64 * P;
  T* acc = tg + (((size_t)64 * P + 3) / 4) * 4;                   // [2·dim·nlg][64]
  const int lane = threadIdx.x;
  const int n_m = dim * nlg;
  for (int i = 0; i < 2 * n_m; ++i) acc[i * 64 + lane] = T(0);
  T stg[PP_NLG][PP_NLG], bs[PP_NLG], cs[PP_NLG], ck[PP_NLG];
#pragma unroll
  for (int j = 0; j < PP_NLG; ++j) {
    bs[j] = T(0); cs[j] = T(0); ck[j] = j < nlg ? wtu_hat[l0 + j] : T(0);
#pragma unroll
    for (int k = 0; k < PP_NLG; ++k) stg[j][k] = T(0);
  }
  for (int64_t c0 = (int64_t)blockIdx.x * 64; c0 < batch; c0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
    tile_stage_in<T, V>(tz, z0 + c0 * dim, dim, P, ncols, lane);
    tile_stage_in<T, V>(tg, ybar + c0 * dim, dim, P, ncols, lane);
    tile_sync();
    if (lane < ncols) {
      const int64_t col = c0 + lane;
      T sk[PP_NLG], tk[PP_NLG];
#pragma unroll
      for (int k = 0; k < PP_NLG; ++k) { sk[k] = k < nlg ? sbar[col * nl + l0 + k] : T(0); tk[k] = k < nlg ? tt[col * nl + l0 + k] : T(0); }
      const T lb = lbar ? lbar[col] : T(0);
      const T* mz = tz + lane * P;
      const T* mg = tg + lane * P;
      for (int r = 0; r < dim; ++r) {
        const T zr = mz[r], gr = mg[r];
        T* a1 = acc + (size_t)(r * nlg) * 64 + lane;
        T* a2 = acc + (size_t)(n_m + r * nlg) * 64 + lane;
#pragma unroll
        for (int k = 0; k < PP_NLG; ++k) {
          if (k < nlg) { a1[k * 64] += zr * sk[k]; a2[k * 64] += gr * tk[k]; }
        }
      }
#pragma unroll
      for (int j = 0; j < PP_NLG; ++j) {
        bs[j] += sk[j];
        const T q = T(1) - tk[j] * tk[j];
        cs[j] += lb * q / (T(1) + ck[j] * q);
#pragma unroll
        for (int k = 0; k < PP_NLG; ++k) stg[j][k] += sk[j] * tk[k];
      }
    }
    tile_sync();
  }
  // lanes -> one partial set: M1 / M2 from the strips, the register sums through a fixed xor butterfly
  const size_t per = 2 * (size_t)n_m + (size_t)nlg * nlg + 2 * (size_t)nlg;
  double* out = partial + (size_t)blockIdx.x * per;
  for (int i = lane; i < 2 * n_m; i += 64) {
    double sum = 0.0;
    for (int l = 0; l < 64; ++l) sum += (double)acc[i * 64 + l];
    out[i] = sum;
  }
#pragma unroll
  for (int j = 0; j < PP_NLG; ++j) {
    double b_ = (double)bs[j], c_ = (double)cs[j];
    for (int m = 1; m < 64; m <<= 1) { b_ += shfl_xor(b_, m); c_ += shfl_xor(c_, m); }
    if (lane == 0 && j < nlg) { out[2 * n_m + (size_t)nlg * nlg + j] = b_; out[2 * n_m + (size_t)nlg * nlg + nlg + j] = c_; }
#pragma unroll
    for (int k = 0; k < PP_NLG; ++k) {
      double s_ = (double)stg[j][k];
      for (int m = 1; m < 64; m <<= 1) s_ += shfl_xor(s_, m);
      if (lane == 0 && j < nlg && k < nlg) out[2 * n_m + (size_t)j * nlg + k] = s_;
    }
  }
}

// ---- the same reduction on the matrix cores (Float32, dim = 64·NH <= 256, 16-byte aligned inputs).
// M1 = Z₀·S̄ᵀ and M2 = Ȳ·Tᵀ are [dim x columns]·[columns x layers] products with the COLUMN index as k.
// v_mfma_f32_16x16x4_f32 (layout probed in scripts/probe_mfma.hip): A[i][k] on lane (i = l%16, k = l/16), B[k][n] on
// lane (n = l%16, k = l/16), D[4(l/16)+v][l%16] in VGPR v.  One wave instruction covers 4 columns (k = l/16): lane
// (i, k) loads the 16-byte pack rows 64h+4i..+3 of column k (16 lanes = 256 contiguous bytes per column, the natural
// coalesced layout) — four consecutive ROWS, not one A entry; but the MFMA that takes register r of every lane computes
// the products of the rows {64h + 4i + r}, a fixed permutation of the output rows, so 4·NH MFMAs per operand pair use
// the packs as they are.  B = s̄ / tanh of (layer n, column k) straight from the [n_layers, batch] work arrays (lanes
// n < nlg; zero above).  The Gram block ST = S̄·Tᵀ is one more MFMA with A = the s̄ register and B = the tanh register.
// Accumulators: 8·NH + 1 quads of VGPRs per wave (68 at dim = 128) instead of 128 scalar accumulators per lane, and
// 4 columns per trip with 4·NH pack loads in flight.  Block combine and partial layout as planar_param_reduce_kernel.
typedef float bjx_mf4 __attribute__((ext_vector_type(4)));
// HS = row slices per column: wave w owns the 64·NHW rows of slice w % HS (dim = 64·NHW·HS) of the columns of its
// group w / HS, so a block covers 16/HS columns per trip.  One slice per wave (NHW = 1) keeps the accumulators at
// 8 + 1 quads: ~80 VGPRs, 6 waves per SIMD instead of 2 with the whole column in one wave — the loads in flight
// per CU, not the matrix pipe (≈ 20 % busy), are what this reduction runs on.  Only slice 0 accumulates ST / b̄ / c̄.
template <int NHW, int HS>
__global__ __launch_bounds__(256) void planar_param_mfma_kernel(const float* __restrict__ z0, const float* __restrict__ ybar, const float* __restrict__ sbar,
                                                                const float* __restrict__ tt, const float* __restrict__ lbar, const float* __restrict__ wtu_hat,
                                                                int64_t batch, int nl, int l0, int nlg, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  constexpr int dim = 64 * NHW * HS;
  constexpr int WG = 4 / HS;                        // column groups (of 4 columns) per block and trip
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hs = wave % HS, wg = wave / HS;
  const int i = lane & 15, kq = lane >> 4;
  bjx_mf4 m1[NHW][4], m2[NHW][4], st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < NHW; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) { m1[h][r] = bjx_mf4{0.f, 0.f, 0.f, 0.f}; m2[h][r] = bjx_mf4{0.f, 0.f, 0.f, 0.f}; }
  float bsum = 0.f, csum = 0.f;
  const bool lay = i < nlg;
  const float cmine = lay ? wtu_hat[l0 + i] : 0.f;
  const int row0 = 64 * NHW * hs + 4 * i;            // first row of my first pack
  // software pipeline: the packs and scalars of the NEXT trip are in flight while this trip's MFMAs run
  const int64_t cstep = (int64_t)gridDim.x * (4 * WG);
  int64_t col = ((int64_t)blockIdx.x * WG + wg) * 4 + kq;
  bjx_f4 nz[NHW], ng[NHW];
  float nsv = 0.f, ntv = 0.f, nlb = 0.f;
  auto fetch = [&](int64_t c) {
    const bool okc = c < batch;
#pragma unroll
    for (int h = 0; h < NHW; ++h) {
      nz[h] = bjx_f4{0.f, 0.f, 0.f, 0.f}; ng[h] = nz[h];
      if (okc) {
        nz[h] = __builtin_nontemporal_load(reinterpret_cast<const bjx_f4*>(z0 + c * dim + row0 + 64 * h));
        ng[h] = __builtin_nontemporal_load(reinterpret_cast<const bjx_f4*>(ybar + c * dim + row0 + 64 * h));
      }
    }
    nsv = (okc && lay) ? sbar[c * nl + l0 + i] : 0.f;
    ntv = (okc && lay) ? tt[c * nl + l0 + i] : 0.f;
    nlb = (okc && lay && lbar && hs == 0) ? lbar[c] : 0.f;
  };
  fetch(col);
  for (; col - kq < batch; col += cstep) {
    bjx_f4 pz[NHW], pg[NHW];
#pragma unroll
    for (int h = 0; h < NHW; ++h) { pz[h] = nz[h]; pg[h] = ng[h]; }
    const float sv = nsv, tv = ntv, lb = nlb;
    fetch(col + cstep);
#pragma unroll
    for (int h = 0; h < NHW; ++h) {
      m1[h][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pz[h].x, sv, m1[h][0], 0, 0, 0);
      m1[h][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pz[h].y, sv, m1[h][1], 0, 0, 0);
      m1[h][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pz[h].z, sv, m1[h][2], 0, 0, 0);
      m1[h][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pz[h].w, sv, m1[h][3], 0, 0, 0);
      m2[h][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pg[h].x, tv, m2[h][0], 0, 0, 0);
      m2[h][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pg[h].y, tv, m2[h][1], 0, 0, 0);
      m2[h][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pg[h].z, tv, m2[h][2], 0, 0, 0);
      m2[h][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pg[h].w, tv, m2[h][3], 0, 0, 0);
    }
    if (hs == 0) {
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(sv, tv, st, 0, 0, 0);          // ST[j][k] += s̄_j t_k
      bsum += sv;
      const float q = 1.f - tv * tv;
      csum += lb * q / (1.f + cmine * q);
    }
  }
  // ---- block combine in a fixed order (waves one after another); D element (vgpr v) of lane (n = i, mq = kq):
  //      A-row m = 4 mq + v  ->  row 64(NHW hs + h) + 4m + r of M1 / M2 (MFMA r), layer n;  ST[j = m][k = n]
  const size_t n_m = (size_t)dim * nlg;
  const size_t per = 2 * n_m + (size_t)nlg * nlg + 2 * (size_t)nlg;
  for (int pass = 0; pass < 4; ++pass) {
    if (wave == pass) {
      const bool first = pass < HS;                  // the first wave of every row slice initialises its rows
      if (lay) {
#pragma unroll
        for (int h = 0; h < NHW; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const size_t i1 = (size_t)(64 * (NHW * hs + h) + 4 * (4 * kq + v) + r) * nlg + i;
              if (first) { red[i1] = (double)m1[h][r][v]; red[n_m + i1] = (double)m2[h][r][v]; }
              else { red[i1] += (double)m1[h][r][v]; red[n_m + i1] += (double)m2[h][r][v]; }
            }
        if (hs == 0) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int j = 4 * kq + v;
            if (j < nlg) {
              const size_t i2 = 2 * n_m + (size_t)j * nlg + i;
              if (pass == 0) red[i2] = (double)st[v]; else red[i2] += (double)st[v];
            }
          }
        }
      }
      // b̄ / c̄ of layer i: the four column quads of the wave one after another
      if (hs == 0) {
        const size_t ib = 2 * n_m + (size_t)nlg * nlg + i;
        for (int kp = 0; kp < 4; ++kp) {
          if (lay && kq == kp) {
            if (pass == 0 && kp == 0) { red[ib] = (double)bsum; red[ib + nlg] = (double)csum; }
            else { red[ib] += (double)bsum; red[ib + nlg] += (double)csum; }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    __syncthreads();
  }
  for (size_t e = threadIdx.x; e < per; e += blockDim.x) partial[(size_t)blockIdx.x * per + e] = red[e];
}

// cross-group Gram entries ST[j][k] with j, k in DIFFERENT layer groups are produced by a small launch of this kernel
// ---- low-dimensional columns (dim <= 32, Float32) on the matrix cores: M1 = Z₀·S̄ᵀ and M2 = Ȳ·Tᵀ are [rows x columns]·[columns x layers]
// products with at most 16 (or 2 x 16) rows — ONE 16x16 accumulator each, the column index as k.  A single-wave block stages 64 consecutive
// columns of Z₀, Ȳ (odd-pitch tiles) and of the [n_layers, batch] work arrays s̄, tanh (contiguous too) through LDS; step m feeds the four
// columns 4m … 4m+3: A = Z[row l%16][column 4m + l/16] (a conflict-free 4-byte LDS read), B = s̄[column 4m + l/16][layer l%16], and the Gram
// block ST = S̄·Tᵀ is one more MFMA on the s̄ and tanh registers.  No per-lane accumulators, no cross-lane reduction for the matrices: the
// accumulators ARE the block's partial (same layout as planar_param_reduce_kernel); b̄ and c̄ are two xor-butterflies at the end.
template <int NRB, int V>
__global__ __launch_bounds__(64) void planar_param_mfma_small_kernel(const float* __restrict__ z0, const float* __restrict__ ybar, const float* __restrict__ sbar,
                                                                     const float* __restrict__ tt, const float* __restrict__ lbar, const float* __restrict__ wtu_hat,
                                                                     int dim, int P, int PN, int64_t batch, int nl, int l0, int nlg, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tz = reinterpret_cast<float*>(smem);
  float* tg = tz + (size_t)64 * P;
  float* ts = tg + (size_t)64 * P;                                  // [64][PN]: s̄ of the tile's columns, all layers
  float* tth = ts + (size_t)64 * PN;
  float* tl = tth + (size_t)64 * PN;                                // ℓ̄ of the tile's columns (one coalesced load per tile, not one per step)
  const int lane = threadIdx.x;
  const int li = lane & 15, lk = lane >> 4;
  bjx_mf4 m1[NRB], m2[NRB], st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < NRB; ++h) { m1[h] = bjx_mf4{0.f, 0.f, 0.f, 0.f}; m2[h] = bjx_mf4{0.f, 0.f, 0.f, 0.f}; }
  float bsum = 0.f, csum = 0.f;
  const bool lay_ok = li < nlg;
  const float cme = lay_ok ? wtu_hat[l0 + li] : 0.f;
  for (int64_t c0 = (int64_t)blockIdx.x * 64; c0 < batch; c0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
    tile_stage_in<float, V>(tz, z0 + c0 * dim, dim, P, ncols, lane);
    tile_stage_in<float, V>(tg, ybar + c0 * dim, dim, P, ncols, lane);
    tile_stage_in<float, V>(ts, sbar + c0 * nl, nl, PN, ncols, lane);
    tile_stage_in<float, V>(tth, tt + c0 * nl, nl, PN, ncols, lane);
    tl[lane] = (lbar && lane < ncols) ? lbar[c0 + lane] : 0.f;
    tile_sync();
    for (int m = 0; m < 16; ++m) {
      const int c = 4 * m + lk;                                      // my column of this step
      const bool col_ok = c < ncols;
      const float sv = (col_ok && lay_ok) ? ts[c * PN + l0 + li] : 0.f;
      const float tv = (col_ok && lay_ok) ? tth[c * PN + l0 + li] : 0.f;
#pragma unroll
      for (int h = 0; h < NRB; ++h) {
        const int row = 16 * h + li;
        const bool ok = col_ok && row < dim;
        const float za = ok ? tz[c * P + row] : 0.f;
        const float ga = ok ? tg[c * P + row] : 0.f;
        m1[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(za, sv, m1[h], 0, 0, 0);
        m2[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, tv, m2[h], 0, 0, 0);
      }
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(sv, tv, st, 0, 0, 0);           // ST[j][k] += s̄_j t_k
      if (col_ok && lay_ok) {
        bsum += sv;
        const float q = 1.f - tv * tv;
        csum += tl[c] * q / (1.f + cme * q);
      }
    }
    tile_sync();
  }
  const size_t n_m = (size_t)dim * nlg;
  const size_t per = 2 * n_m + (size_t)nlg * nlg + 2 * (size_t)nlg;
  double* out = partial + (size_t)blockIdx.x * per;
  // D[4 (l/16) + r][l%16] in register r: rows 4 lk + r of the block, layer li
#pragma unroll
  for (int h = 0; h < NRB; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * h + 4 * lk + r;
      if (row < dim && lay_ok) { out[(size_t)row * nlg + li] = (double)m1[h][r]; out[n_m + (size_t)row * nlg + li] = (double)m2[h][r]; }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 4 * lk + r;
    if (j < nlg && lay_ok) out[2 * n_m + (size_t)j * nlg + li] = (double)st[r];
  }
  double b_ = (double)bsum, c_ = (double)csum;
  b_ += shfl_xor(b_, 16); c_ += shfl_xor(c_, 16);
  b_ += shfl_xor(b_, 32); c_ += shfl_xor(c_, 32);
  if (lk == 0 && lay_ok) { out[2 * n_m + (size_t)nlg * nlg + li] = b_; out[2 * n_m + (size_t)nlg * nlg + nlg + li] = c_; }
}

// ---- the same sums with ROWS owned by threads (round 5): columns beyond one pack per lane of a 64-lane group.
// planar_param_reduce_kernel keeps [R packs] x [V rows] x [8 layers] x 2 accumulators per lane — 128 registers at R = 2, 256 at R = 4
// (10–14 % of the roofline at 333 / 509 rows) and stops at R = 4 (1 024 rows Float32, 512 Float64: a loud error until round 5).  Here a
// thread owns ONE pack of rows (2·V·8 accumulators) and walks columns: TP = 64 / 128 / 256 threads span a chunk of TP packs
// (blockIdx.x), the CG = 256 / TP thread groups of a block and the blockIdx.y take different columns, and every (blockIdx.y, group)
// writes its own Float64 set — the existing column sum adds the sets in a fixed order.  Packs on element-aligned addresses, the last
// one partial; the s̄ / tanh rows of a column are wave-uniform (scalar loads).  The Gram block, b̄ and c̄ do not depend on the rows:
// planar_param_sums_kernel writes them into the summed set.
// (the column walk: FULL = the layer group has all PP_NLG layers — the s̄ / tanh rows are then read without per-layer guards and the
//  scalar loads merge; two columns per trip in Float32, one in Float64 (the scalars of two columns did not fit the SGPR file: 117
//  v_readlane / 99 v_writelane per trip); whole packs take ONE branch around all their loads, not one per load)
template <class T, int V, bool FULL>
__device__ __forceinline__ void planar_param_rows_walk(const T* __restrict__ z0, const T* __restrict__ ybar, const T* __restrict__ sbar, const T* __restrict__ tt,
                                                       int64_t dim, int64_t batch, int nl, int l0, int nlg, int64_t set, int64_t nsets, int64_t row0, int nrow,
                                                       T (&m1)[V][PP_NLG], T (&m2)[V][PP_NLG]) {
  constexpr int U = sizeof(T) == 4 ? 2 : 1;
  Pack<T, V> pz[U], pg[U], nz[U], ng[U];
  auto fetch = [&](int64_t col, Pack<T, V> (&az)[U], Pack<T, V> (&ag)[U]) {
    if (nrow == V) {
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int64_t cq = col + q * nsets;
        if (cq < batch) { az[q] = load_pack<T, V, true>(z0 + cq * dim + row0); ag[q] = load_pack<T, V, true>(ybar + cq * dim + row0); }
      }
    } else if (nrow > 0) {
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int64_t cq = col + q * nsets;
        if (cq < batch) { az[q] = load_pack_part<T, V>(z0 + cq * dim + row0, nrow); ag[q] = load_pack_part<T, V>(ybar + cq * dim + row0, nrow); }
      }
    }
  };
  if (set < batch) fetch(set, pz, pg);
  for (int64_t col = set; col < batch; col += U * nsets) {
    if (col + U * nsets < batch) fetch(col + U * nsets, nz, ng);          // the next trip's packs are in flight during this trip's products
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t cq = col + q * nsets;
      if (cq < batch) {
        T sk[PP_NLG], tk[PP_NLG];
        const T* sp = sbar + cq * nl + l0;
        const T* tp = tt + cq * nl + l0;
#pragma unroll
        for (int k = 0; k < PP_NLG; ++k) {
          if constexpr (FULL) { sk[k] = sp[k]; tk[k] = tp[k]; }
          else { sk[k] = k < nlg ? sp[k] : T(0); tk[k] = k < nlg ? tp[k] : T(0); }
        }
        if (nrow > 0) {
#pragma unroll
          for (int j = 0; j < V; ++j)
#pragma unroll
            for (int k = 0; k < PP_NLG; ++k) { m1[j][k] += pz[q].v[j] * sk[k]; m2[j][k] += pg[q].v[j] * tk[k]; }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < U; ++q) { pz[q] = nz[q]; pg[q] = ng[q]; }
  }
}

template <class T, int V>
__global__ __launch_bounds__(256) void planar_param_rows_kernel(const T* __restrict__ z0, const T* __restrict__ ybar, const T* __restrict__ sbar,
                                                                const T* __restrict__ tt, int64_t dim, int64_t batch, int nl, int l0, int nlg, int TP,
                                                                double* __restrict__ partial) {
  const int tp = threadIdx.x & (TP - 1);
  const int cg = __builtin_amdgcn_readfirstlane((int)threadIdx.x / TP);       // TP >= 64: a wave lies in one group
  const int CG = 256 / TP;
  const int64_t row0 = ((int64_t)blockIdx.x * TP + tp) * V;
  const bool ok = row0 < dim;
  const int nrow = ok ? (int)(dim - row0 < V ? dim - row0 : V) : 0;
  T m1[V][PP_NLG], m2[V][PP_NLG];
#pragma unroll
  for (int j = 0; j < V; ++j)
#pragma unroll
    for (int k = 0; k < PP_NLG; ++k) { m1[j][k] = T(0); m2[j][k] = T(0); }
  const int64_t set = (int64_t)blockIdx.y * CG + cg, nsets = (int64_t)gridDim.y * CG;
  if (nlg == PP_NLG) planar_param_rows_walk<T, V, true>(z0, ybar, sbar, tt, dim, batch, nl, l0, nlg, set, nsets, row0, nrow, m1, m2);
  else planar_param_rows_walk<T, V, false>(z0, ybar, sbar, tt, dim, batch, nl, l0, nlg, set, nsets, row0, nrow, m1, m2);
  const size_t n_m = (size_t)dim * nlg;
  const size_t per = 2 * n_m + (size_t)nlg * nlg + 2 * (size_t)nlg;
  double* out = partial + (size_t)set * per;
  if (ok) {
#pragma unroll
    for (int j = 0; j < V; ++j)
      if (j < nrow)
        for (int k = 0; k < nlg; ++k) {
          out[(size_t)(row0 + j) * nlg + k] = (double)m1[j][k];
          out[n_m + (size_t)(row0 + j) * nlg + k] = (double)m2[j][k];
        }
  }
  if (blockIdx.x == 0)
    for (int i = tp; i < nlg * nlg + 2 * nlg; i += TP) out[2 * n_m + i] = 0.0;
}

// [ST nlg*nlg][b̄ nlg][c̄ nlg] of a layer group over the whole batch: a thread owns columns (the [batch][n_layers] rows of s̄ and tanh
// are contiguous: coalesced), keeps the nlg² + 2·nlg sums in registers, the block combines them in Float64 and writes one set; the
// column sum adds the sets.  (First form: one block per ENTRY walking the whole batch with a stride of n_layers elements — 80 blocks,
// 0.5 ms at 2¹⁹ columns, as long as the row reduction it completes.)
template <class T>
__global__ __launch_bounds__(256) void planar_param_sums_kernel(const T* __restrict__ sbar, const T* __restrict__ tt, const T* __restrict__ lbar,
                                                                const T* __restrict__ wtu_hat, int64_t batch, int nl, int l0, int nlg, double* __restrict__ part) {
  __shared__ double red[4][PP_NLG * PP_NLG + 2 * PP_NLG];
  T g[PP_NLG][PP_NLG], bs[PP_NLG], cs[PP_NLG], cw[PP_NLG];
#pragma unroll
  for (int j = 0; j < PP_NLG; ++j) {
    bs[j] = T(0); cs[j] = T(0); cw[j] = j < nlg ? wtu_hat[l0 + j] : T(0);
#pragma unroll
    for (int k = 0; k < PP_NLG; ++k) g[j][k] = T(0);
  }
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < batch; n += (int64_t)gridDim.x * 256) {
    T sk[PP_NLG], tk[PP_NLG];
#pragma unroll
    for (int k = 0; k < PP_NLG; ++k) { sk[k] = k < nlg ? sbar[n * nl + l0 + k] : T(0); tk[k] = k < nlg ? tt[n * nl + l0 + k] : T(0); }
    const T lb = lbar ? lbar[n] : T(0);
#pragma unroll
    for (int j = 0; j < PP_NLG; ++j) {
      bs[j] += sk[j];
      const T q = T(1) - tk[j] * tk[j];
      cs[j] += lb * q / (T(1) + cw[j] * q);
#pragma unroll
      for (int k = 0; k < PP_NLG; ++k) g[j][k] += sk[j] * tk[k];
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int E = nlg * nlg + 2 * nlg;
#pragma unroll
  for (int j = 0; j < PP_NLG; ++j) {
#pragma unroll
    for (int k = 0; k < PP_NLG; ++k) {
      const double v = group_sum<64>((double)g[j][k]);
      if (lane == 0 && j < nlg && k < nlg) red[wv][j * nlg + k] = v;
    }
    const double vb = group_sum<64>((double)bs[j]), vc = group_sum<64>((double)cs[j]);
    if (lane == 0 && j < nlg) { red[wv][nlg * nlg + j] = vb; red[wv][nlg * nlg + nlg + j] = vc; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) part[(size_t)blockIdx.x * E + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// The cross-group Gram matrix ST[j][k] = Σ_n s̄_jn t_kn of a stack of more than PP_NLG layers, in 8 x 8 tiles: blockIdx.y = tile,
// blockIdx.x = slice of the batch, a thread owns columns (coalesced rows of the [batch][n_layers] arrays) and the tile's 64 sums;
// planar_gram_final_kernel adds the slices.  (First form: one block per ENTRY walking the whole batch with a stride of n_layers
// elements: 8 ms at 2²² columns and 12 layers — vjp_params of 12 / 16 layers ran at 11 / 10 % where 8 layers reach 61 %.)
template <class T>
__global__ __launch_bounds__(256) void planar_gram_tiles_kernel(const T* __restrict__ sbar, const T* __restrict__ tt, int64_t batch, int nl, int nt,
                                                                double* __restrict__ part) {
  __shared__ double red[4][64];
  const int j0 = ((int)blockIdx.y / nt) * 8, k0 = ((int)blockIdx.y % nt) * 8;
  T g[8][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) g[j][k] = T(0);
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < batch; n += (int64_t)gridDim.x * 256) {
    T sk[8], tk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sk[k] = j0 + k < nl ? sbar[n * nl + j0 + k] : T(0); tk[k] = k0 + k < nl ? tt[n * nl + k0 + k] : T(0); }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) g[j][k] += sk[j] * tk[k];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double v = group_sum<64>((double)g[j][k]);
      if (lane == 0) red[wv][j * 8 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < 64) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void planar_gram_final_kernel(const double* __restrict__ part, int slices, int nl, int nt, double* __restrict__ st) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nl * nl) return;
  const int j = e / nl, k = e % nl;
  const size_t tile = (size_t)(j / 8) * nt + k / 8;
  const int in = (j % 8) * 8 + k % 8;
  double a = 0.0;
  for (int b = 0; b < slices; ++b) a += part[(tile * slices + b) * 64 + in];
  st[e] = a;
}
constexpr int PP_GRAM_SLICES = 256;
template <class T>
int planar_gram_launch(bjx_ctx* ctx, const T* s_out, const T* t_out, int64_t batch, int nl, double* work, double* st) {
  const int nt = (nl + 7) / 8;
  int slices = (int)((batch + 1023) / 1024);
  if (slices > PP_GRAM_SLICES) slices = PP_GRAM_SLICES;
  if (slices < 1) slices = 1;
  BjxProf prof_(ctx);
  hipLaunchKernelGGL(planar_gram_tiles_kernel<T>, dim3(slices, nt * nt), dim3(256), 0, ctx->stream, s_out, t_out, batch, nl, nt, work);
  hipLaunchKernelGGL(planar_gram_final_kernel, dim3((nl * nl + 255) / 256), dim3(256), 0, ctx->stream, work, slices, nl, nt, st);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

// Fixed-order sum of the per-block partial sets into one set: thread e owns element e of the set, adjacent threads
// read adjacent doubles of every block's set (coalesced), four independent accumulators keep loads in flight.
// (planar_param_finalize_kernel used to walk the 1024 sets itself, 8 blocks of strided dependent-latency loads:
// 1.1 ms at 8 layers x 128 rows — as long as the streaming reduction it finishes.)
// gridDim.y > 1: slice y sums the sets [y·chunk, (y+1)·chunk) into out[y·per + e] (a second launch with gridDim.y = 1 sums the slices):
// a small set (a few hundred entries, low-dimensional flows) would otherwise be summed over thousands of blocks by ONE thread block.
__global__ __launch_bounds__(256) void planar_param_colsum_kernel(const double* __restrict__ partial, int nblocks, size_t per, double* __restrict__ out, int chunk = 0) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per) return;
  if (chunk > 0) {
    const int b0 = (int)blockIdx.y * chunk;
    partial += (size_t)b0 * per;
    out += (size_t)blockIdx.y * per;
    nblocks = nblocks - b0 < chunk ? nblocks - b0 : chunk;
    if (nblocks < 0) nblocks = 0;
  }
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = 0;
  for (; b + 4 <= nblocks; b += 4) {
    a0 += partial[(size_t)b * per + e];
    a1 += partial[(size_t)(b + 1) * per + e];
    a2 += partial[(size_t)(b + 2) * per + e];
    a3 += partial[(size_t)(b + 3) * per + e];
  }
  for (; b < nblocks; ++b) a0 += partial[(size_t)b * per + e];
  out[e] = (a0 + a1) + (a2 + a3);
}

// one block per layer: the chain rule through get_u_hat on the summed partial set (nblocks = 1 after the column sum)
template <class T>
__global__ __launch_bounds__(256) void planar_param_finalize_kernel(const double* __restrict__ partial, int nblocks, int64_t dim, int nl, int l0, int nlg,
                                                                   const double* st, const T* __restrict__ w, const T* __restrict__ u,
                                                                   const T* __restrict__ u_hat, T* __restrict__ w_bar, T* __restrict__ u_bar, T* __restrict__ b_bar) {
  __shared__ double red[12];
  __shared__ double st_l[PP_NLG * PP_NLG];
  const int kk = blockIdx.x;                      // layer inside the group
  const int k = l0 + kk;
  const size_t n_m = (size_t)dim * nlg;
  const size_t per = 2 * n_m + (size_t)nlg * nlg + 2 * (size_t)nlg;
  if (!st) {                                      // a single layer group: the Gram matrix is the block the reduce kernel produced
    for (int e = threadIdx.x; e < nlg * nlg; e += blockDim.x) {
      double acc = 0.0;
      for (int bidx = 0; bidx < nblocks; ++bidx) acc += partial[(size_t)bidx * per + 2 * n_m + e];
      st_l[e] = acc;
    }
    __syncthreads();
    st = st_l;                                    // nl == nlg here: same [j * nl + k] indexing
  }
  const T* wk = w + (int64_t)k * dim;
  const T* uk = u + (int64_t)k * dim;
  // a = wᵀu, ‖w‖²
  double a = 0.0, n2 = 0.0;
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) { a += (double)wk[i] * (double)uk[i]; n2 += (double)wk[i] * (double)wk[i]; }
  a = group_sum<64>(a); n2 = group_sum<64>(n2);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = n2; }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  n2 = (red[4] + red[5]) + (red[6] + red[7]);
  __syncthreads();
  double bsum = 0.0, csum = 0.0;
  for (int bidx = 0; bidx < nblocks; ++bidx) {
    const double* pb = partial + (size_t)bidx * per + 2 * n_m + (size_t)nlg * nlg;
    bsum += pb[kk]; csum += pb[nlg + kk];
  }
  const double sa = 1.0 / (1.0 + exp(-a));                                   // logistic(a) = d(log1pexp a)/da
  const double kap = ((double)d_log1pexp((T)(-a)) - 1.0) / n2;
  const double ka = -(1.0 - sa) / n2;                                        // dκ/da = -logistic(-a)/‖w‖²
  // û̄ and w̄(direct) per row; wᵀû̄ needs a block reduction, so two sweeps over the rows
  double wtuhb = 0.0;
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) {
    double m2 = 0.0;
    for (int bidx = 0; bidx < nblocks; ++bidx) m2 += partial[(size_t)bidx * per + n_m + (size_t)i * nlg + kk];
    for (int j = k + 1; j < nl; ++j) m2 += (double)w[(int64_t)j * dim + i] * st[j * nl + k];
    u_bar[(int64_t)k * dim + i] = (T)m2;                                     // û̄ for now
    wtuhb += (double)wk[i] * m2;
  }
  wtuhb = group_sum<64>(wtuhb);
  if ((threadIdx.x & 63) == 0) red[8 + (threadIdx.x >> 6)] = wtuhb;
  __syncthreads();
  wtuhb = (red[8] + red[9]) + (red[10] + red[11]);
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) {
    double m1 = 0.0;
    for (int bidx = 0; bidx < nblocks; ++bidx) m1 += partial[(size_t)bidx * per + (size_t)i * nlg + kk];
    for (int j = 0; j < k; ++j) m1 += (double)u_hat[(int64_t)j * dim + i] * st[k * nl + j];
    const double uhb = (double)u_bar[(int64_t)k * dim + i];
    const double wi = (double)wk[i], ui = (double)uk[i];
    u_bar[(int64_t)k * dim + i] = (T)(uhb + wi * (ka * wtuhb + csum * sa));
    w_bar[(int64_t)k * dim + i] = (T)(m1 + kap * uhb + wtuhb * (ka * ui - 2.0 * kap / n2 * wi) + csum * sa * ui);
  }
  if (threadIdx.x == 0) b_bar[k] = (T)bsum;
}

template <class T>
int planar_vjp_params_impl(bjx_ctx* ctx, const T* w, const T* u, const T* b, int nl, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar,
                           T* w_bar, T* u_bar, T* b_bar, T* work, int64_t dim, int64_t batch) {
  BJX_REQUIRE(ctx, batch >= 1, BJX_ERR_SHAPE, "bjx_planar_vjp_params: empty batch");
  T* s_out = work;                                   // [nl, batch]
  T* t_out = work + (size_t)nl * batch;
  int rc = planar_vjp_impl<T>(ctx, 0, w, u, b, nl, in, out_bar, ladj_bar, in_bar, dim, batch, t_out, s_out);
  if (rc) return rc;
  // tables left by planar_vjp_impl: û [nl][dim], wᵀû [nl] — in the scratch, or (a stack too large for it) at the head of the grown workspace
  const T* u_hat = (((size_t)nl * dim + nl) * sizeof(T) + 255) / 256 * 256 > BJX_SCRATCH_BYTES ? static_cast<const T*>(ctx->big_ws) : static_cast<const T*>(ctx->scratch);
  const T* wtu = u_hat + (size_t)nl * dim;
  FlowCfg c;
  const size_t nlg_max = nl < PP_NLG ? nl : PP_NLG;
  constexpr int VWr = Vec16<T>::N;
  const bool cfg_ok = flow_cfg<T>(ctx, in, out_bar, dim, batch, &c, true);
  // more than one pack per lane of a 64-lane group (256 rows Float32, 128 Float64): rows owned by threads (planar_param_rows_kernel);
  // the register accumulators stop at four packs, the block combine at the LDS
  static const int use_rows = getenv("BJX_PLANAR_PARAM_ROWS") ? atoi(getenv("BJX_PLANAR_PARAM_ROWS")) : 1;
  const bool must_rows = !(cfg_ok && c.R <= 4) || (2 * (size_t)dim * nlg_max + nlg_max * nlg_max + 2 * nlg_max) * sizeof(double) > BJX_LDS_MAX;
  // (a value above 1 = the first height that takes the rows path; Float64 from 33 rows: 40 / 72 / 100 rows 14.8 / 19.2 / 25.7 → 16.9 / 28.8 / 38.1 %
  //  of the whole call; Float32 below 257 rows: 72 / 100 / 250 rows lose, 192 / 200 win — the register accumulators stay)
  const int64_t rows_min = use_rows > 1 ? use_rows : (std::is_same<T, double>::value ? 33 : 64 * VWr + 1);
  const bool mfma_shape = std::is_same<T, float>::value && (dim == 64 || dim == 128 || dim == 256) && c.V == VWr && bjx_aligned16(out_bar);
  const bool rows_path = must_rows || (use_rows && dim >= rows_min && dim > 32 && !mfma_shape);
  if (rows_path) {
    const int64_t packs = (dim + VWr - 1) / VWr;
    const int TP = packs > 128 ? 256 : (packs > 64 ? 128 : 64);
    const int CG = 256 / TP;
    const int64_t chunks = (packs + TP - 1) / TP;
    BJX_REQUIRE(ctx, chunks < 65536, BJX_ERR_UNSUPPORTED, "bjx_planar_vjp_params: dim %lld too large", (long long)dim);
    const size_t per_max_r = 2 * (size_t)dim * nlg_max + nlg_max * nlg_max + 2 * nlg_max;
    // sets = (blockIdx.y, thread group): four resident blocks per CU (two — 512 blocks — left the loads of two waves per SIMD in
    // flight: 20 % of the roofline), at least 256 columns per set (a set is written and read once in Float64: 128·dim bytes against
    // 2 048·dim bytes of input), at most 192 MiB of sets
    int64_t S = ((int64_t)ctx->num_cu * 4 + chunks - 1) / chunks;
    const int64_t s_mem = (int64_t)(((size_t)192 << 20) / (per_max_r * sizeof(double))) / CG;
    const int64_t s_cols = batch / (256 * (int64_t)CG);
    if (S > s_mem) S = s_mem;
    if (S > s_cols) S = s_cols;
    if (S > 65535) S = 65535;
    if (S < 1) S = 1;
    const int nsets = (int)(S * CG);
    const size_t st_n = (size_t)nl * nl;
    constexpr int SL = 32;                             // slices of the first column-sum stage
    const bool one_group = nl <= PP_NLG;
    const size_t gram_n = one_group ? 0 : (size_t)((nl + 7) / 8) * ((nl + 7) / 8) * 64 * PP_GRAM_SLICES;
    { int rc2 = bjx_ensure_partials(ctx, (size_t)nsets * per_max_r + st_n + per_max_r + (size_t)SL * per_max_r + gram_n); if (rc2) return rc2; }
    double* partial = ctx->partials;
    double* st = partial + (size_t)nsets * per_max_r;
    double* psum = st + st_n;
    double* slices = psum + per_max_r;
    if (!one_group) { int rcg = planar_gram_launch<T>(ctx, s_out, t_out, batch, nl, slices + (size_t)SL * per_max_r, st); if (rcg) return rcg; }
    for (int l0 = 0; l0 < nl; l0 += PP_NLG) {
      const int nlg = nl - l0 < PP_NLG ? nl - l0 : PP_NLG;
      const size_t per = 2 * (size_t)dim * nlg + (size_t)nlg * nlg + 2 * (size_t)nlg;
      const unsigned gx = (unsigned)((per + 255) / 256);
      BjxProf prof_(ctx);
      hipLaunchKernelGGL((planar_param_rows_kernel<T, VWr>), dim3((unsigned)chunks, (unsigned)S), dim3(256), 0, ctx->stream, in, out_bar, s_out, t_out, dim, batch, nl, l0, nlg, TP, partial);
      if (nsets > 64) {
        const int chunk = (nsets + SL - 1) / SL;
        hipLaunchKernelGGL(planar_param_colsum_kernel, dim3(gx, SL), dim3(256), 0, ctx->stream, partial, nsets, per, slices, chunk);
        hipLaunchKernelGGL(planar_param_colsum_kernel, dim3(gx), dim3(256), 0, ctx->stream, slices, SL, per, psum, 0);
      } else {
        hipLaunchKernelGGL(planar_param_colsum_kernel, dim3(gx), dim3(256), 0, ctx->stream, partial, nsets, per, psum, 0);
      }
      {
        // Gram block / b̄ / c̄: one set per block into the (now free) head of the row sets, summed into the tail of the summed set
        const int E = nlg * nlg + 2 * nlg;
        int sb = (int)((batch + 511) / 512);
        if (sb > 1024) sb = 1024;
        if ((size_t)sb * E > (size_t)nsets * per) sb = (int)((size_t)nsets * per / E);
        if (sb < 1) sb = 1;
        hipLaunchKernelGGL(planar_param_sums_kernel<T>, dim3(sb), dim3(256), 0, ctx->stream, s_out, t_out, ladj_bar, wtu, batch, nl, l0, nlg, partial);
        hipLaunchKernelGGL(planar_param_colsum_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, sb, (size_t)E, psum + 2 * (size_t)dim * nlg, 0);
      }
      hipLaunchKernelGGL(planar_param_finalize_kernel<T>, dim3(nlg), dim3(256), 0, ctx->stream, psum, 1, dim, nl, l0, nlg, one_group ? (const double*)nullptr : (const double*)st, w, u, u_hat, w_bar, u_bar, b_bar);
      BJX_CHECK_LAUNCH(ctx);
    }
    return BJX_OK;
  }
  // planar_param_reduce_kernel gives the Gram row / b̄ / c̄ of layer l0 + gl to lane gl of a column's group: the group must have at
  // least PP_NLG lanes even when the column is only one or two packs (dim <= 4·PP_NLG: lanes without a pack only do that part).
  // (Found with dim = 2, 4, 8: with G < 8 the rows of the upper layers were never accumulated — wrong w̄, ū, b̄.)
  if (c.G < PP_NLG) c.G = PP_NLG;
  const int R = c.R;
  const int cols_per_block = 256 / c.G;
  constexpr int VW = Vec16<T>::N;
  static const int use_mfma = getenv("BJX_PLANAR_PARAM_MFMA") ? atoi(getenv("BJX_PLANAR_PARAM_MFMA")) : 1;
  static const int mfma_blocks = 1024;
  const bool mfma = use_mfma && std::is_same<T, float>::value && c.V == VW && (dim == 64 || dim == 128 || dim == 256) && bjx_aligned16(out_bar);   // 192 rows: three slices do not divide the four waves, one wave per column needs 357 registers
  // persistent grids with equal grid-stride shares: every block must be resident (a second round doubles the time)
  const bool small_walk = (std::is_same<T, float>::value && dim <= 32) || dim <= 12;   // single-wave blocks of 64 columns (planar_param_mfma_small_kernel / planar_param_walk_kernel)
  const int block_cap = mfma ? mfma_blocks : (small_walk ? 4096 : 1024);
  int nblocks = (mfma || small_walk) ? (int)((batch + 63) / 64) : (int)((batch + (int64_t)cols_per_block * 16 - 1) / ((int64_t)cols_per_block * 16));
  if (nblocks > block_cap) nblocks = block_cap;
  if (nblocks < 1) nblocks = 1;
  const size_t per_max = 2 * (size_t)dim * PP_NLG + PP_NLG * PP_NLG + 2 * PP_NLG;
  const size_t st_n = (size_t)nl * nl;
  const bool one_group = nl <= PP_NLG;
  const size_t gram_n = one_group ? 0 : (size_t)((nl + 7) / 8) * ((nl + 7) / 8) * 64 * PP_GRAM_SLICES;
  { int rc2 = bjx_ensure_partials(ctx, (size_t)nblocks * per_max + st_n + per_max + 32 * per_max + gram_n); if (rc2) return rc2; }
  double* partial = ctx->partials;
  double* st = partial + (size_t)nblocks * per_max;
  double* psum = st + st_n;                          // the block partials summed into one set
  double* slices = psum + per_max;                   // [32][per]: first stage of the column sum when there are many sets of few entries
  if (!one_group) { int rcg = planar_gram_launch<T>(ctx, s_out, t_out, batch, nl, slices + 32 * per_max, st); if (rcg) return rcg; }   // cross-group Gram entries
  for (int l0 = 0; l0 < nl; l0 += PP_NLG) {
    const int nlg = nl - l0 < PP_NLG ? nl - l0 : PP_NLG;
    const size_t per = 2 * (size_t)dim * nlg + (size_t)nlg * nlg + 2 * (size_t)nlg;
    const size_t smem = per * sizeof(double);
    BJX_REQUIRE(ctx, smem <= BJX_LDS_MAX, BJX_ERR_UNSUPPORTED, "bjx_planar_vjp_params: dim %lld too large for the block combine", (long long)dim);
    static const int walk_max = getenv("BJX_FLOW_WALK_MAX") ? atoi(getenv("BJX_FLOW_WALK_MAX")) : 32;
    const int64_t Pw = dim | 1;
    const size_t smem_walk = ((size_t)64 * Pw + (((size_t)64 * Pw + 3) / 4) * 4 + (size_t)2 * dim * nlg * 64) * sizeof(T);
    const int64_t PNw = nl | 1;
    const size_t smem_ms = ((size_t)2 * 64 * Pw + (size_t)2 * 64 * PNw + 64) * sizeof(float);
    if (std::is_same<T, float>::value && dim <= 32 && dim <= walk_max && smem_ms <= 48 * 1024) {
      BjxProf prof_(ctx);
      const bool vec = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(s_out) && bjx_aligned16(t_out);
      const float* zf = reinterpret_cast<const float*>(in); const float* gf = reinterpret_cast<const float*>(out_bar);
      const float* sf = reinterpret_cast<const float*>(s_out); const float* tf = reinterpret_cast<const float*>(t_out);
      const float* lf = reinterpret_cast<const float*>(ladj_bar); const float* cf = reinterpret_cast<const float*>(wtu);
#define PPS(NRB_, V_) hipLaunchKernelGGL((planar_param_mfma_small_kernel<NRB_, V_>), dim3(nblocks), dim3(64), smem_ms, ctx->stream, zf, gf, sf, tf, lf, cf, (int)dim, (int)Pw, (int)PNw, batch, nl, l0, nlg, partial)
      if (dim <= 16) { if (vec) PPS(1, 4); else PPS(1, 1); } else { if (vec) PPS(2, 4); else PPS(2, 1); }
#undef PPS
    } else if (dim <= 12 && dim <= walk_max && smem_walk <= 64 * 1024) {
      BjxProf prof_(ctx);
      const bool vec = bjx_aligned16(in) && bjx_aligned16(out_bar);
      if (vec) hipLaunchKernelGGL((planar_param_walk_kernel<T, VW>), dim3(nblocks), dim3(64), smem_walk, ctx->stream, in, out_bar, s_out, t_out, ladj_bar, wtu, (int)dim, (int)Pw, batch, nl, l0, nlg, partial);
      else hipLaunchKernelGGL((planar_param_walk_kernel<T, 1>), dim3(nblocks), dim3(64), smem_walk, ctx->stream, in, out_bar, s_out, t_out, ladj_bar, wtu, (int)dim, (int)Pw, batch, nl, l0, nlg, partial);
    } else if (mfma) {
      BjxProf prof_(ctx);
      const float* zf = reinterpret_cast<const float*>(in); const float* gf = reinterpret_cast<const float*>(out_bar);
      const float* sf = reinterpret_cast<const float*>(s_out); const float* tf = reinterpret_cast<const float*>(t_out);
      const float* lf = reinterpret_cast<const float*>(ladj_bar); const float* cf = reinterpret_cast<const float*>(wtu);
#define PPM(NHW_, HS_) do { bjx_allow_big_lds(planar_param_mfma_kernel<NHW_, HS_>, smem); hipLaunchKernelGGL((planar_param_mfma_kernel<NHW_, HS_>), dim3(nblocks), dim3(256), smem, ctx->stream, zf, gf, sf, tf, lf, cf, batch, nl, l0, nlg, partial); } while (0)
      switch ((int)(dim / 64)) { case 1: PPM(1, 1); break; case 2: PPM(1, 2); break; default: PPM(1, 4); break; }
#undef PPM
    } else {
      BjxProf prof_(ctx);
#define PPR(V_, R_) do { bjx_allow_big_lds(planar_param_reduce_kernel<T, V_, R_>, smem); hipLaunchKernelGGL((planar_param_reduce_kernel<T, V_, R_>), dim3(nblocks), dim3(256), smem, ctx->stream, in, out_bar, s_out, t_out, ladj_bar, wtu, dim, batch, c.G, nl, l0, nlg, partial); } while (0)
#define PPR_V(V_) do { if (R == 1) PPR(V_, 1); else if (R == 2) PPR(V_, 2); else PPR(V_, 4); } while (0)
      if (c.V == VW) PPR_V(VW); else PPR_V(1);
#undef PPR_V
#undef PPR
    }
    BJX_CHECK_LAUNCH(ctx);
    {
      BjxProf prof_(ctx);
      const unsigned gx = (unsigned)((per + 255) / 256);
      if (nblocks > 256 && gx <= 8) {                 // few entries, many sets: 32 slices, then the slices
        constexpr int S = 32;
        const int chunk = (nblocks + S - 1) / S;
        hipLaunchKernelGGL(planar_param_colsum_kernel, dim3(gx, S), dim3(256), 0, ctx->stream, partial, nblocks, per, slices, chunk);
        hipLaunchKernelGGL(planar_param_colsum_kernel, dim3(gx), dim3(256), 0, ctx->stream, slices, S, per, psum, 0);
      } else {
        hipLaunchKernelGGL(planar_param_colsum_kernel, dim3(gx), dim3(256), 0, ctx->stream, partial, nblocks, per, psum, 0);
      }
    }
    BJX_CHECK_LAUNCH(ctx);
    {
      BjxProf prof_(ctx);
      hipLaunchKernelGGL(planar_param_finalize_kernel<T>, dim3(nlg), dim3(256), 0, ctx->stream, psum, 1, dim, nl, l0, nlg, one_group ? (const double*)nullptr : (const double*)st, w, u, u_hat, w_bar, u_bar, b_bar);
    }
    BJX_CHECK_LAUNCH(ctx);
  }
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_planar_vjp_params(bjx_ctx* ctx, bjx_dtype dt, const void* w, const void* u, const void* b, int n_layers, const void* in,
                                  const void* out_bar, const void* ladj_bar, void* in_bar, void* w_bar, void* u_bar, void* b_bar, void* work,
                                  int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0 && n_layers >= 1, BJX_ERR_SHAPE, "bjx_planar_vjp_params: bad size (dim=%lld, n_layers=%d)", (long long)dim, n_layers);
  BJX_REQUIRE(ctx, w && u && b && in && out_bar && in_bar && w_bar && u_bar && b_bar && work, BJX_ERR_ARG, "bjx_planar_vjp_params: null pointer");
  if (dt == BJX_F32) return planar_vjp_params_impl<float>(ctx, (const float*)w, (const float*)u, (const float*)b, n_layers, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, (float*)w_bar, (float*)u_bar, (float*)b_bar, (float*)work, dim, batch);
  if (dt == BJX_F64) return planar_vjp_params_impl<double>(ctx, (const double*)w, (const double*)u, (const double*)b, n_layers, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, (double*)w_bar, (double*)u_bar, (double*)b_bar, (double*)work, dim, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_planar_vjp_params: bad dtype %d", (int)dt);
}

BJX_API int bjx_planar_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* w, const void* u, const void* b, int n_layers, const void* in,
                           const void* out_bar, const void* ladj_bar, void* in_bar, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0 && n_layers >= 1, BJX_ERR_SHAPE, "bjx_planar_vjp: bad size (dim=%lld, n_layers=%d)", (long long)dim, n_layers);
  BJX_REQUIRE(ctx, w && u && b && ((in && out_bar && in_bar) || batch == 0), BJX_ERR_ARG, "bjx_planar_vjp: null pointer");
  if (dt == BJX_F32) return planar_vjp_impl<float>(ctx, inverse, (const float*)w, (const float*)u, (const float*)b, n_layers, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, dim, batch);
  if (dt == BJX_F64) return planar_vjp_impl<double>(ctx, inverse, (const double*)w, (const double*)u, (const double*)b, n_layers, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, dim, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_planar_vjp: bad dtype %d", (int)dt);
}

namespace {
template <class T>
int radial_vjp_impl(bjx_ctx* ctx, int inverse, const T* alpha_, const T* beta, const T* z0, const T* in, const T* out_bar, const T* ladj_bar,
                    T* in_bar, int64_t dim, int64_t batch, T* work = nullptr, double* zsum_out = nullptr, bool* zsum_done = nullptr) {
  if (zsum_done) *zsum_done = false;
  if (batch == 0) return BJX_OK;
  {
    static const int walk_max = getenv("BJX_FLOW_WALK_MAX") ? atoi(getenv("BJX_FLOW_WALK_MAX")) : 32;
    if (dim <= walk_max && dim <= 32 && (dim % Vec16<T>::N != 0 || (sizeof(T) == 8 && dim > 2)) && (const void*)in != (const void*)in_bar) {   // (zsum_out stays unset: the caller reduces ȳ - z̄ itself)
      constexpr int VWW = Vec16<T>::N;
      const int P = (int)(dim | 1);
      const size_t smem_w = (size_t)2 * 64 * P * sizeof(T);
      const int64_t tiles = (batch + 63) / 64;
      const int64_t cap = (int64_t)ctx->num_cu * 32;
      const int grid_w = (int)(tiles < cap ? tiles : cap);
      const bool vec = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
      {
        BjxProf prof_(ctx);
#define RVW(D_, I_, V_) hipLaunchKernelGGL((radial_vjp_walk_kernel<T, D_, I_, V_>), dim3(grid_w), dim3(64), smem_w, ctx->stream, alpha_, beta, z0, in, out_bar, ladj_bar, in_bar, (int)dim, P, batch, work)
#define RVW_V(D_, I_) do { if (vec) RVW(D_, I_, VWW); else RVW(D_, I_, 1); } while (0)
#define RVW_D(I_) do { if (dim <= 4) RVW_V(4, I_); else if (dim <= 8) RVW_V(8, I_); else if (dim <= 16) RVW_V(16, I_); else RVW_V(32, I_); } while (0)
        if (inverse) RVW_D(true); else RVW_D(false);
#undef RVW_D
#undef RVW_V
#undef RVW
      }
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
  }
  FlowCfg c;
  constexpr int VW = Vec16<T>::N;
  auto launch_tall = [&]() -> int {
    // columns taller than the register kernels hold: one block per column, two passes (radial_vjp_tall_kernel)
    BJX_REQUIRE(ctx, batch < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_radial_vjp: batch too large for one launch");
    const int64_t capt = (int64_t)ctx->num_cu * 8;
    const int gridt = (int)(batch < capt ? batch : capt);
    RadialArgs<T> At{alpha_, beta, z0, 0};
    const bool v_ok = dim % VW == 0 && bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar) && bjx_aligned16(z0);
    {
      BjxProf prof_(ctx);
#define LAUNCH_RVT(V_, INV_) hipLaunchKernelGGL((radial_vjp_tall_kernel<T, V_, INV_>), dim3(gridt), dim3(256), 0, ctx->stream, At, in, out_bar, ladj_bar, in_bar, dim, batch, work)
      if (v_ok) { if (inverse) LAUNCH_RVT(VW, true); else LAUNCH_RVT(VW, false); }
      else { if (inverse) LAUNCH_RVT(1, true); else LAUNCH_RVT(1, false); }
#undef LAUNCH_RVT
    }
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  };
  if (!flow_cfg<T>(ctx, in, in_bar, dim, batch, &c, true)) return launch_tall();
  const bool whole = dim % VW == 0 && bjx_aligned16(in) && bjx_aligned16(in_bar);   // otherwise: element-aligned packs, partial last pack (any alignment)
  if (c.V == VW && whole && !bjx_aligned16(out_bar)) {           // scalar packs
    int G = 1;
    while (G < 64 && G < dim) G <<= 1;
    int64_t need_r = (dim + G - 1) / G;
    int R = 1;
    while (R < need_r) R <<= 1;
    if (R > FLOW_R_MAX) return launch_tall();
    c.V = 1; c.G = G; c.R = R;
  }
  {
    const int uc = c.R == 1 ? 2 : 1;
    const int64_t cpb = (int64_t)(256 / c.G) * uc;
    c.grid = (batch + cpb - 1) / cpb;
  }
  const size_t tab_bytes = (size_t)dim * sizeof(T);
  const bool lds = tab_bytes <= 60 * 1024;
  RadialArgs<T> A{alpha_, beta, z0, lds ? 1 : 0};
  size_t smem = lds ? tab_bytes : 0;
  if (zsum_out && !inverse && c.V == VW && c.R <= 4) {
    // row sums of ȳ - z̄ from the same pass: one Float64 partial set per block, two reduction stages
    const size_t zoff = (smem + 15) / 16 * 16;
    const size_t sets = (size_t)c.grid, stage1 = (sets + 255) / 256;
    if (zoff + 4 * (size_t)dim * sizeof(double) <= 64 * 1024 && stage1 <= 4096) {
      { int rc = bjx_ensure_partials(ctx, (sets + stage1) * (size_t)dim); if (rc) return rc; }
      double* zpart = ctx->partials;
      double* st1 = zpart + sets * (size_t)dim;
      smem = zoff + 4 * (size_t)dim * sizeof(double);
      {
        BjxProf prof_(ctx);
        FLOW_SWITCH_R(radial_vjp_kernel, T, VW, false, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, work, zpart, (int)zoff)
      }
      BJX_CHECK_LAUNCH(ctx);
      {
        BjxProf prof_(ctx);
        hipLaunchKernelGGL(flow_sets_reduce_kernel, dim3((unsigned)stage1), dim3(256), 0, ctx->stream, zpart, (int64_t)sets, (int)dim, 256, st1);
        hipLaunchKernelGGL(flow_sets_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st1, (int64_t)stage1, (int)dim, (int)stage1, zsum_out);
      }
      BJX_CHECK_LAUNCH(ctx);
      *zsum_done = true;
      return BJX_OK;
    }
  }
  BjxProf prof_(ctx);
  if (c.V == VW) {
    if (!inverse) { FLOW_SWITCH_R(radial_vjp_kernel, T, VW, false, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, work) }
    else { FLOW_SWITCH_R(radial_vjp_kernel, T, VW, true, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, work) }
  } else {
    if (!inverse) { FLOW_SWITCH_R(radial_vjp_kernel, T, 1, false, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, work) }
    else { FLOW_SWITCH_R(radial_vjp_kernel, T, 1, true, A, in, out_bar, ladj_bar, in_bar, dim, batch, c.G, work) }
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

namespace {
// ------------------------------------------------------------------ Radial parameter pullback
// radial_layer.jl:43-60 with α̂ = softplus(α_), β̂ = -α̂ + softplus(β), h = 1/(α̂ + r), a = 1 + β̂h, D = 1 + β̂h - β̂h²r:
//   y = z + β̂hδ,  ℓ = (d-1) log a + log D.     Per column, from r and δᵀȳ (written by radial_vjp_kernel):
//   g_β̂ = h δᵀȳ + ℓ̄ [(d-1)h/a + (h - h²r)/D],   g_α̂ = -h² [β̂ δᵀȳ + ℓ̄ ((d-1)β̂/a + (β̂ - 2β̂hr)/D)]   (∂h/∂α̂ = -h²)
//   ᾱ_ = σ(α_)(Σ g_α̂ - Σ g_β̂),  β̄ = σ(β) Σ g_β̂ ;   z̄₀ = Σ_n (ȳ_n - z̄_n)  (y and ℓ depend on z, z₀ through δ = z - z₀ only).
template <class T>
__global__ __launch_bounds__(256) void radial_param_partial_kernel(const T* __restrict__ alpha_, const T* __restrict__ beta, const T* __restrict__ work,
                                                                   const T* __restrict__ lbar, int64_t dim, int64_t batch, double* __restrict__ partials) {
  __shared__ double red[8];
  const double al = (double)d_log1pexp(alpha_[0]);
  const double bh = -al + (double)d_log1pexp(beta[0]);
  double ga = 0.0, gb = 0.0;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < batch; n += (int64_t)gridDim.x * blockDim.x) {
    const double r = (double)work[n], dg = (double)work[batch + n], lb = lbar ? (double)lbar[n] : 0.0;
    const double h = 1.0 / (al + r), a = 1.0 + bh * h, D = 1.0 + bh * h - bh * h * h * r;
    gb += h * dg + lb * ((double)(dim - 1) * h / a + (h - h * h * r) / D);
    ga += -h * h * (bh * dg + lb * ((double)(dim - 1) * bh / a + (bh - 2.0 * bh * h * r) / D));
  }
  ga = group_sum<64>(ga); gb = group_sum<64>(gb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave] = ga; red[4 + wave] = gb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    partials[2 * blockIdx.x + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}
template <class T>
__global__ __launch_bounds__(256) void radial_param_finalize_kernel(const T* __restrict__ alpha_, const T* __restrict__ beta, const double* __restrict__ partials,
                                                                    int nblocks, const double* __restrict__ sy, const double* __restrict__ sz, int64_t dim,
                                                                    T* __restrict__ alpha_bar, T* __restrict__ beta_bar, T* __restrict__ z0_bar) {
  __shared__ double red[8];
  for (int64_t i = threadIdx.x; i < dim; i += blockDim.x) z0_bar[i] = (T)(sy[i] - sz[i]);
  double ga = 0.0, gb = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) { ga += partials[2 * b]; gb += partials[2 * b + 1]; }
  ga = group_sum<64>(ga); gb = group_sum<64>(gb);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = ga; red[4 + (threadIdx.x >> 6)] = gb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ga = (red[0] + red[1]) + (red[2] + red[3]);
    gb = (red[4] + red[5]) + (red[6] + red[7]);
    const double sa = 1.0 / (1.0 + exp(-(double)alpha_[0])), sb = 1.0 / (1.0 + exp(-(double)beta[0]));
    alpha_bar[0] = (T)(sa * (ga - gb));
    beta_bar[0] = (T)(sb * gb);
  }
}

template <class T>
int radial_vjp_params_impl(bjx_ctx* ctx, bjx_dtype dt, const T* alpha_, const T* beta, const T* z0, const T* in, const T* out_bar, const T* ladj_bar,
                           T* in_bar, T* alpha_bar, T* beta_bar, T* z0_bar, T* work, int64_t dim, int64_t batch) {
  BJX_REQUIRE(ctx, (size_t)(2 * (2 * dim + 1)) * sizeof(double) <= BJX_SCRATCH_BYTES, BJX_ERR_UNSUPPORTED, "bjx_radial_vjp_params: dim %lld too large", (long long)dim);
  double* sy = reinterpret_cast<double*>(ctx->scratch);
  double* sz = sy + (2 * dim + 1);
  if (batch == 0) {
    BJX_HIP(ctx, hipMemsetAsync(z0_bar, 0, (size_t)dim * sizeof(T), ctx->stream));
    BJX_HIP(ctx, hipMemsetAsync(alpha_bar, 0, sizeof(T), ctx->stream));
    BJX_HIP(ctx, hipMemsetAsync(beta_bar, 0, sizeof(T), ctx->stream));
    return BJX_OK;
  }
  bool fused = false;
  { int rc = radial_vjp_impl<T>(ctx, 0, alpha_, beta, z0, in, out_bar, ladj_bar, in_bar, dim, batch, work, sy, &fused); if (rc) return rc; }
  if (fused) {                                       // sy = Σ_n (ȳ - z̄) from the pullback kernel itself
    BJX_HIP(ctx, hipMemsetAsync(sz, 0, (size_t)dim * sizeof(double), ctx->stream));
  } else {
    { int rc = bjx_row_moments(ctx, dt, out_bar, nullptr, sy, dim, batch); if (rc) return rc; }    // Σ_n ȳ  (rows)
    { int rc = bjx_row_moments(ctx, dt, in_bar, nullptr, sz, dim, batch); if (rc) return rc; }     // Σ_n z̄
  }
  int nblocks = (int)((batch + 255) / 256);
  if (nblocks > 512) nblocks = 512;
  { int rc = bjx_ensure_partials(ctx, (size_t)2 * nblocks); if (rc) return rc; }
  hipLaunchKernelGGL(radial_param_partial_kernel<T>, dim3(nblocks), dim3(256), 0, ctx->stream, alpha_, beta, work, ladj_bar, dim, batch, ctx->partials);
  BJX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(radial_param_finalize_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, alpha_, beta, ctx->partials, nblocks, sy, sz, dim, alpha_bar, beta_bar, z0_bar);
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_radial_vjp_params(bjx_ctx* ctx, bjx_dtype dt, const void* alpha_, const void* beta, const void* z0, const void* in,
                                  const void* out_bar, const void* ladj_bar, void* in_bar, void* alpha_bar, void* beta_bar, void* z0_bar,
                                  void* work, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_radial_vjp_params: bad size");
  BJX_REQUIRE(ctx, alpha_ && beta && z0 && alpha_bar && beta_bar && z0_bar && ((in && out_bar && in_bar && work) || batch == 0), BJX_ERR_ARG,
              "bjx_radial_vjp_params: null pointer");
  if (dt == BJX_F32) return radial_vjp_params_impl<float>(ctx, dt, (const float*)alpha_, (const float*)beta, (const float*)z0, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, (float*)alpha_bar, (float*)beta_bar, (float*)z0_bar, (float*)work, dim, batch);
  if (dt == BJX_F64) return radial_vjp_params_impl<double>(ctx, dt, (const double*)alpha_, (const double*)beta, (const double*)z0, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, (double*)alpha_bar, (double*)beta_bar, (double*)z0_bar, (double*)work, dim, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_radial_vjp_params: bad dtype %d", (int)dt);
}

BJX_API int bjx_radial_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* alpha_, const void* beta, const void* z0, const void* in,
                           const void* out_bar, const void* ladj_bar, void* in_bar, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_radial_vjp: bad size");
  BJX_REQUIRE(ctx, alpha_ && beta && z0 && ((in && out_bar && in_bar) || batch == 0), BJX_ERR_ARG, "bjx_radial_vjp: null pointer");
  if (dt == BJX_F32) return radial_vjp_impl<float>(ctx, inverse, (const float*)alpha_, (const float*)beta, (const float*)z0, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, dim, batch);
  if (dt == BJX_F64) return radial_vjp_impl<double>(ctx, inverse, (const double*)alpha_, (const double*)beta, (const double*)z0, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, dim, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_radial_vjp: bad dtype %d", (int)dt);
}

BJX_API int bjx_radial(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* alpha_, const void* beta, const void* z0,
                       const void* in, void* out, void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_radial: bad size");
  BJX_REQUIRE(ctx, alpha_ && beta && z0 && ((in && out) || batch == 0), BJX_ERR_ARG, "bjx_radial: null pointer");
  if (dt == BJX_F32) return radial_impl<float>(ctx, inverse, (const float*)alpha_, (const float*)beta, (const float*)z0, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return radial_impl<double>(ctx, inverse, (const double*)alpha_, (const double*)beta, (const double*)z0, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_radial: bad dtype %d", (int)dt);
}
