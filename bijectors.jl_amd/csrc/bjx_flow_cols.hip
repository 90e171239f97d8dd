// bjx_flow_cols.hip — Planar on the column-tile mapping (round 5): a block owns C columns at a time with the columns in registers.
// Split from bjx_flow.hip (whose compile time the ~100 instantiations here had doubled); the launchers are called from planar_impl /
// planar_vjp_impl there and return 1 when a shape is not theirs.
#include "bjx_internal.h"
#include <type_traits>

namespace {
using namespace bjx;

#include "bjx_flow_common.inc"

// Between the register tiles (Float32 to 1 024 rows) and planar_vjp_tall_kernel (bjx_flow.hip): a BLOCK owns C columns at a time, thread t holds the packs
// t, t + 256, ... (R of them) of each of the C columns in registers.  planar_vjp_kernel gives a column to 64 lanes: every wave loads the
// w / û rows of every layer for ONE column (2·n_layers·dim per column through the L2: 4–5 TB/s of parameter traffic, 5 % of the HBM
// roofline at 1 500 … 8 192 rows) and evaluates ONE tanh on 64 lanes; here a parameter pack is loaded once for C columns and the
// C dot products of a layer are reduced together (one butterfly each, one barrier per layer: the LDS slots alternate by layer parity).
// Packs on element-aligned addresses, the last one partial.  C·R = 16 (8 at R = 1): 64 data registers.  (Blocks of 512 / 1 024 threads that keep
// R small and C large beyond 4 096 rows: no faster at 512 — 153–179 registers — and spilled at 1 024.)
__device__ __forceinline__ float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double lane_bcast(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int G> __device__ __forceinline__ float group_sum_fast(float v) { return group_sum_f32_dpp<G>(v); }
template <int G> __device__ __forceinline__ double group_sum_fast(double v) { return group_sum<G>(v); }
// C values per lane -> lane L holds the wave sum of value L / (64 / C): log2(C) halving exchanges (C/2 + C/4 + ... shuffles in all)
// and one butterfly over the 64 / C lanes that are left, instead of C full butterflies (6·C shuffles).
template <class T, int C, int G> __device__ __forceinline__ T wave_sum_scatter_rec(const T (&s)[C], int lane) {
  if constexpr (C == 1) return group_sum_fast<G>(s[0]);
  else {
    constexpr int H = G / 2;
    const bool hi = lane & H;
    T a[C / 2];
#pragma unroll
    for (int i = 0; i < C / 2; ++i) a[i] = (hi ? s[C / 2 + i] : s[i]) + shfl_xor(hi ? s[i] : s[C / 2 + i], H);
    return wave_sum_scatter_rec<T, C / 2, H>(a, lane);
  }
}
template <class T, int C> __device__ __forceinline__ T wave_sum_scatter(const T (&s)[C], int lane) {
  static_assert(C == 1 || C == 2 || C == 4 || C == 8 || C == 16, "C");
  return wave_sum_scatter_rec<T, C, 64>(s, lane);
}

template <class T, int V, int R, int C, bool INV, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu((R <= 4 && sizeof(T) == 4) ? 4 : 1, 8))) void planar_vjp_cols_kernel(const PlanarArgs<T> A, const T* __restrict__ x, const T* __restrict__ ybar, const T* __restrict__ lbar,
                                                             T* __restrict__ xbar, int64_t dim, int64_t batch, T* __restrict__ t_out, T* __restrict__ s_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NWV = NT / 64;
  constexpr bool PF = R <= 4;                          // request parameter rows ahead of the barrier (three row buffers: not at 8+ packs per thread)
  T* red = reinterpret_cast<T*>(smem);                 // [2][NWV][C]
  T* tsave = red + 2 * NWV * C;                        // [C][n_layers]
  const int nl = A.n_layers;
  const int64_t nvc = (dim + V - 1) / V;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cme = lane & (C - 1);                      // the column whose scalar recurrence this lane evaluates (every wave redundantly, once)
  int nrow[R];
  int64_t off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t v = threadIdx.x + (int64_t)r * NT;
    off[r] = v * V;
    nrow[r] = v < nvc ? (int)(dim - v * V < V ? dim - v * V : V) : 0;
  }
  int par = 0;
  // block sums of C values: lane L of every wave gets the sum of value L & (C - 1)
  auto reduce = [&](const T (&s)[C]) -> T {
    const T v = wave_sum_scatter<T, C>(s, lane);
    T* rp = red + par * NWV * C;
    if ((lane & (64 / C - 1)) == 0) rp[wv * C + lane / (64 / C)] = v;
    __syncthreads();
    T a = T(0);
#pragma unroll
    for (int q = 0; q < NWV; ++q) a += rp[q * C + cme];
    par ^= 1;
    return a;
  };
  auto load_row = [&](const T* row, Pack<T, V> (&p)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (nrow[r] > 0) p[r] = load_pack_part<T, V>(row + off[r], nrow[r]);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) p[r].v[j] = T(0);
      }
    }
  };
  auto load_par = [&](const T* row, Pack<T, V> (&p)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (nrow[r] > 0) p[r] = load_pack_part_cached<T, V>(row + off[r], nrow[r]);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) p[r].v[j] = T(0);
      }
    }
  };
  const int64_t tiles = (batch + C - 1) / C;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t col0 = tile * C;
    Pack<T, V> z[C][R];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t col = col0 + c < batch ? col0 + c : batch - 1;
      load_row(x + col * dim, z[c]);
    }
    const bool me_ok = col0 + cme < batch;
    const T lbme = (lbar && me_ok) ? lbar[col0 + cme] : T(0);
    // ---- primal sweep: t_l of every layer and column.  The parameter rows do not depend on the data: with PF the row a step needs
    //      AFTER its reduction (û_l) and the next step's w are requested before the barrier
    Pack<T, V> pw[R], pu[R], pnx[R];
    if (PF) load_par(A.w + (int64_t)(INV ? nl - 1 : 0) * dim, pw);
    for (int li = 0; li < nl; ++li) {
      const int l = INV ? nl - 1 - li : li;
      const bool more = li + 1 < nl;
      if constexpr (PF) { if (more) load_par(A.u_hat + (int64_t)l * dim, pu); }
      else load_par(A.w + (int64_t)l * dim, pw);
      T s[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        s[c] = T(0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < V; ++j) s[c] += pw[r].v[j] * z[c][r].v[j];
      }
      if constexpr (PF) { if (more) load_par(A.w + (int64_t)(INV ? l - 1 : l + 1) * dim, pnx); }
      const T sme = reduce(s);
      T tme;
      if (!INV) tme = flow_tanh(sme + A.b[l]);
      else { T s2u; planar_inv_act<T>(sme, A.wtu_hat[l], A.b[l], tme, s2u); }      // Float64: Float32 pre-solve + two Newton steps (find_alpha_act64), not the safeguarded loop
      if (threadIdx.x < C) tsave[threadIdx.x * nl + l] = tme;
      if (more) {
        if constexpr (!PF) load_par(A.u_hat + (int64_t)l * dim, pu);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const T tc = lane_bcast(tme, c);
          const T a = INV ? -tc : tc;
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < V; ++j) z[c][r].v[j] += pu[r].v[j] * a;
        }
        if constexpr (PF) {
#pragma unroll
          for (int r = 0; r < R; ++r) pw[r] = pnx[r];
        }
      }
    }
    __syncthreads();                                   // tsave complete
    // ---- reverse sweep on the cotangent
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t col = col0 + c < batch ? col0 + c : batch - 1;
      load_row(ybar + col * dim, z[c]);
    }
    if (PF) load_par(A.u_hat + (int64_t)(INV ? 0 : nl - 1) * dim, pu);
    for (int li = 0; li < nl; ++li) {
      const int l = INV ? li : nl - 1 - li;
      const bool more = li + 1 < nl;
      if constexpr (PF) load_par(A.w + (int64_t)l * dim, pw);            // for the update after the reduction
      else load_par(A.u_hat + (int64_t)l * dim, pu);
      T d[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        d[c] = T(0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < V; ++j) d[c] += pu[r].v[j] * z[c][r].v[j];
      }
      if constexpr (PF) { if (more) load_par(A.u_hat + (int64_t)(INV ? l + 1 : l - 1) * dim, pnx); }
      const T dme = reduce(d);
      const T cw = A.wtu_hat[l];
      const T t = tsave[cme * nl + l];
      const T q = T(1) - t * t;
      T sbme;
      if (!INV) {
        sbme = dme * q + lbme * cw * (T(-2) * t) * q / (T(1) + cw * q);
        if (s_out && threadIdx.x < C && me_ok) { s_out[(col0 + cme) * nl + l] = sbme; t_out[(col0 + cme) * nl + l] = t; }
      } else {
        const T den = T(1) + cw * q;
        sbme = q / den * (-dme + lbme * T(2) * cw * t / den);
      }
      if constexpr (!PF) load_par(A.w + (int64_t)l * dim, pw);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const T sb = lane_bcast(sbme, c);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < V; ++j) z[c][r].v[j] += pw[r].v[j] * sb;
      }
      if constexpr (PF) {
        if (more) {
#pragma unroll
          for (int r = 0; r < R; ++r) pu[r] = pnx[r];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (col0 + c < batch) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (nrow[r] > 0) store_pack_part<T, V>(xbar + (col0 + c) * dim + off[r], z[c][r], nrow[r]);
      }
    }
    __syncthreads();                                   // tsave is rewritten by the next tile
  }
}

// The forward / inverse map on the same mapping (round 5): planar_kernel gives a column to 64 lanes — one tanh / log1p per wave and
// layer, the w / û rows of every layer loaded per column: 7 % of the roofline beyond 1 024 rows Float32, 8–19 % in Float64 beyond 128.
template <class T, int V, int R, int C, bool INV, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu((R <= 4 && sizeof(T) == 4) ? 4 : 1, 8))) void planar_cols_kernel(const PlanarArgs<T> A, const T* __restrict__ x, T* __restrict__ y, T* __restrict__ ladj_ps, int64_t dim,
                                                         int64_t batch, int accumulate, double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NWV = NT / 64;
  constexpr bool PF = R <= 4;
  T* red = reinterpret_cast<T*>(smem);                 // [2][NWV][C]
  double* redd = reinterpret_cast<double*>(red + 2 * NWV * C + (2 * NWV * C) % 2);
  const int nl = A.n_layers;
  const int64_t nvc = (dim + V - 1) / V;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cme = lane & (C - 1);
  int nrow[R];
  int64_t off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t v = threadIdx.x + (int64_t)r * NT;
    off[r] = v * V;
    nrow[r] = v < nvc ? (int)(dim - v * V < V ? dim - v * V : V) : 0;
  }
  int par = 0;
  auto reduce = [&](const T (&s)[C]) -> T {
    const T v = wave_sum_scatter<T, C>(s, lane);
    T* rp = red + par * NWV * C;
    if ((lane & (64 / C - 1)) == 0) rp[wv * C + lane / (64 / C)] = v;
    __syncthreads();
    T a = T(0);
#pragma unroll
    for (int q = 0; q < NWV; ++q) a += rp[q * C + cme];
    par ^= 1;
    return a;
  };
  auto load_row = [&](const T* row, Pack<T, V> (&p)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (nrow[r] > 0) p[r] = load_pack_part<T, V>(row + off[r], nrow[r]);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) p[r].v[j] = T(0);
      }
    }
  };
  auto load_par = [&](const T* row, Pack<T, V> (&p)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (nrow[r] > 0) p[r] = load_pack_part_cached<T, V>(row + off[r], nrow[r]);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) p[r].v[j] = T(0);
      }
    }
  };
  double acc = 0.0;
  const int64_t tiles = (batch + C - 1) / C;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t col0 = tile * C;
    Pack<T, V> z[C][R];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t col = col0 + c < batch ? col0 + c : batch - 1;
      load_row(x + col * dim, z[c]);
    }
    const bool me_ok = col0 + cme < batch;
    T ladj = T(0);
    Pack<T, V> pw[R], pu[R], pnx[R];
    if (PF) load_par(A.w + (int64_t)(INV ? nl - 1 : 0) * dim, pw);
    for (int li = 0; li < nl; ++li) {
      const int l = INV ? nl - 1 - li : li;
      const bool more = li + 1 < nl;
      if constexpr (PF) load_par(A.u_hat + (int64_t)l * dim, pu);
      else load_par(A.w + (int64_t)l * dim, pw);
      T s[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        s[c] = T(0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < V; ++j) s[c] += pw[r].v[j] * z[c][r].v[j];
      }
      if constexpr (PF) { if (more) load_par(A.w + (int64_t)(INV ? l - 1 : l + 1) * dim, pnx); }
      const T sme = reduce(s);
      const T bl = A.b[l], cw = A.wtu_hat[l];
      T t, s2;
      if (!INV) flow_tanh_sech2(sme + bl, t, s2);
      else planar_inv_act<T>(sme, cw, bl, t, s2);
      const T ld = Fast<T>::log1p(cw * s2);            // planar_layer.jl:107
      ladj += INV ? -ld : ld;
      const T tme = INV ? -t : t;
      if constexpr (!PF) load_par(A.u_hat + (int64_t)l * dim, pu);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const T a = lane_bcast(tme, c);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < V; ++j) z[c][r].v[j] += pu[r].v[j] * a;
      }
      if constexpr (PF) {
        if (more) {
#pragma unroll
          for (int r = 0; r < R; ++r) pw[r] = pnx[r];
        }
      }
    }
    if (accumulate & 2) {                              // BJX_BASE_STDNORMAL: + log N(out; 0, I)
      T q[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        q[c] = T(0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < V; ++j) q[c] += z[c][r].v[j] * z[c][r].v[j];     // rows that do not exist hold zeros
      }
      ladj += T(-0.5) * reduce(q) - (T)dim * T(0.91893853320467274178);
    }
    if (y) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (col0 + c < batch) {
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (nrow[r] > 0) store_pack_part<T, V>(y + (col0 + c) * dim + off[r], z[c][r], nrow[r]);
        }
      }
    }
    if (threadIdx.x < C && me_ok) {
      if (ladj_ps) ladj_ps[col0 + cme] = (accumulate & 1) ? ladj_ps[col0 + cme] + ladj : ladj;
      acc += (double)ladj;
    }
  }
  if (partials) block_publish_partial(acc, redd, partials);
}

}  // namespace

namespace bjx {
template <class T>
int planar_cols_launch(bjx_ctx* ctx, int inverse, const T* w, const T* u_hat, const T* wtu, const T* b, int nl, const T* in, T* out, T* ladj_ps,
                       double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
    // a block per C columns, the columns in registers (planar_cols_kernel): beyond the Float32 register tiles, and Float64
    constexpr int VWc = Vec16<T>::N;
    static const int cols_min_f32 = getenv("BJX_PLANAR_COLS_MIN_F32") ? atoi(getenv("BJX_PLANAR_COLS_MIN_F32")) : 1025;
    static const int cols_min_f64 = getenv("BJX_PLANAR_COLS_MIN_F64") ? atoi(getenv("BJX_PLANAR_COLS_MIN_F64")) : 33;
    const int64_t cols_min = std::is_same<T, float>::value ? cols_min_f32 : cols_min_f64;
    const int64_t packs_c = (dim + VWc - 1) / VWc;
    if (cols_min > 0 && dim >= cols_min && dim >= 2 * VWc && packs_c <= 256 * 32 && batch < ((int64_t)1 << 40)) {
      int NTc = 256, Rc = 32;
      for (int r = 32; r >= 1; r >>= 1)
        for (int nt = 256; nt >= (r == 1 ? 64 : (r <= 4 ? 192 : 256)); nt -= 64)
          if ((int64_t)nt * r >= packs_c && nt * r <= NTc * Rc) { NTc = nt; Rc = r; }
      constexpr bool is_f64 = std::is_same<T, double>::value;
      const int Cc = Rc == 1 ? (is_f64 ? 16 : 8) : (Rc >= 16 ? 1 : 16 / Rc);        // (Float64: the scalar recurrence of a layer, evaluated once per wave, costs as much as the products of 8 columns)
      const int64_t tiles = (batch + Cc - 1) / Cc;
      const int64_t capc = (int64_t)ctx->num_cu * (2048 / NTc);
      const int gridc = (int)(tiles < capc ? tiles : capc);
      if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)gridc); if (rc) return rc; }
      double* partials_c = ladj_sum ? ctx->partials : nullptr;
      PlanarArgs<T> Ac{w, u_hat, wtu, b, nl, 0};
      const int accum_c = ((flags & BJX_ACCUMULATE) ? 1 : 0) | ((flags & BJX_BASE_STDNORMAL) ? 2 : 0);
      const size_t smem_c = ((size_t)2 * (NTc / 64) * Cc + 2) * sizeof(T) + 8 * sizeof(double);
      {
        BjxProf prof_(ctx);
#define PFC(R_, C_, NT_) do { if (inverse) hipLaunchKernelGGL((planar_cols_kernel<T, VWc, R_, C_, true, NT_>), dim3(gridc), dim3(NT_), smem_c, ctx->stream, Ac, in, out, ladj_ps, dim, batch, accum_c, partials_c); \
                              else hipLaunchKernelGGL((planar_cols_kernel<T, VWc, R_, C_, false, NT_>), dim3(gridc), dim3(NT_), smem_c, ctx->stream, Ac, in, out, ladj_ps, dim, batch, accum_c, partials_c); } while (0)
      if constexpr (is_f64) {
        if (Rc == 1) { switch (NTc) { case 64: PFC(1, 16, 64); break; case 128: PFC(1, 16, 128); break; case 192: PFC(1, 16, 192); break; default: PFC(1, 16, 256); break; } }
      }
      if (is_f64 && Rc == 1) {}
      else if (NTc == 64) PFC(1, 8, 64);
      else if (NTc == 128) PFC(1, 8, 128);
      else if (NTc == 192) { switch (Rc) { case 1: PFC(1, 8, 192); break; case 2: PFC(2, 8, 192); break; default: PFC(4, 4, 192); break; } }
        else switch (Rc) { case 1: PFC(1, 8, 256); break; case 2: PFC(2, 8, 256); break; case 4: PFC(4, 4, 256); break; case 8: PFC(8, 2, 256); break; case 16: PFC(16, 1, 256); break; default: PFC(32, 1, 256); break; }
#undef PFC
      }
      BJX_CHECK_LAUNCH(ctx);
      if (ladj_sum) return bjx_launch_finalize(ctx, gridc, ladj_sum, 0.0, 0, 0.0, flags);
      return BJX_OK;
    }
    return 1;                                          // 1 = not served
}

template <class T>
int planar_vjp_cols_launch(bjx_ctx* ctx, int inverse, const T* w, const T* u_hat, const T* wtu, const T* b, int nl, const T* in, const T* out_bar,
                           const T* ladj_bar, T* in_bar, int64_t dim, int64_t batch, T* t_out, T* s_out) {
    // a block per C columns, the columns in registers (planar_vjp_cols_kernel): beyond the Float32 register tiles, and Float64
    constexpr int VWc = Vec16<T>::N;
    static const int cols_min_f32 = getenv("BJX_PLANAR_VJP_COLS_MIN_F32") ? atoi(getenv("BJX_PLANAR_VJP_COLS_MIN_F32")) : 1025;
    static const int cols_min_f64 = getenv("BJX_PLANAR_VJP_COLS_MIN_F64") ? atoi(getenv("BJX_PLANAR_VJP_COLS_MIN_F64")) : 33;
    const int64_t cols_min = std::is_same<T, float>::value ? cols_min_f32 : cols_min_f64;
    const int64_t packs_c = (dim + VWc - 1) / VWc;
    if (cols_min > 0 && dim >= cols_min && dim >= 2 * VWc && packs_c <= 256 * 32 && (size_t)nl * 16 * sizeof(T) <= 32 * 1024 && batch < ((int64_t)1 << 40)) {
      // threads per block x packs per thread: the smallest NT·R that covers the column (NT = 64 … 256 in waves, R a power of two)
      int NTc = 256, Rc = 32;
      for (int r = 32; r >= 1; r >>= 1)
        for (int nt = 256; nt >= (r == 1 ? 64 : (r <= 4 ? 192 : 256)); nt -= 64)
          if ((int64_t)nt * r >= packs_c && nt * r <= NTc * Rc) { NTc = nt; Rc = r; }
      constexpr bool is_f64 = std::is_same<T, double>::value;
      const int Cc = Rc == 1 ? (is_f64 ? 16 : 8) : (Rc >= 16 ? 1 : 16 / Rc);        // (Float64: the scalar recurrence of a layer, evaluated once per wave, costs as much as the products of 8 columns)
      const int64_t tiles = (batch + Cc - 1) / Cc;
      const int64_t capc = (int64_t)ctx->num_cu * (2048 / NTc);
      const int gridc = (int)(tiles < capc ? tiles : capc);
      PlanarArgs<T> Ac{w, u_hat, wtu, b, nl, 0};
      const size_t smem_c = ((size_t)2 * (NTc / 64) * Cc + (size_t)Cc * nl) * sizeof(T);
      BjxProf prof_(ctx);
#define PVC(R_, C_, NT_) do { if (inverse) hipLaunchKernelGGL((planar_vjp_cols_kernel<T, VWc, R_, C_, true, NT_>), dim3(gridc), dim3(NT_), smem_c, ctx->stream, Ac, in, out_bar, ladj_bar, in_bar, dim, batch, t_out, s_out); \
                              else hipLaunchKernelGGL((planar_vjp_cols_kernel<T, VWc, R_, C_, false, NT_>), dim3(gridc), dim3(NT_), smem_c, ctx->stream, Ac, in, out_bar, ladj_bar, in_bar, dim, batch, t_out, s_out); } while (0)
      if constexpr (is_f64) {
        if (Rc == 1) { switch (NTc) { case 64: PVC(1, 16, 64); break; case 128: PVC(1, 16, 128); break; case 192: PVC(1, 16, 192); break; default: PVC(1, 16, 256); break; } }
      }
      if (is_f64 && Rc == 1) {}
      else if (NTc == 64) PVC(1, 8, 64);
      else if (NTc == 128) PVC(1, 8, 128);
      else if (NTc == 192) { switch (Rc) { case 1: PVC(1, 8, 192); break; case 2: PVC(2, 8, 192); break; default: PVC(4, 4, 192); break; } }
      else switch (Rc) { case 1: PVC(1, 8, 256); break; case 2: PVC(2, 8, 256); break; case 4: PVC(4, 4, 256); break; case 8: PVC(8, 2, 256); break; case 16: PVC(16, 1, 256); break; default: PVC(32, 1, 256); break; }
#undef PVC
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
    return 1;
}

template int planar_cols_launch<float>(bjx_ctx*, int, const float*, const float*, const float*, const float*, int, const float*, float*, float*, double*, int64_t, int64_t, uint32_t);
template int planar_cols_launch<double>(bjx_ctx*, int, const double*, const double*, const double*, const double*, int, const double*, double*, double*, double*, int64_t, int64_t, uint32_t);
template int planar_vjp_cols_launch<float>(bjx_ctx*, int, const float*, const float*, const float*, const float*, int, const float*, const float*, const float*, float*, int64_t, int64_t, float*, float*);
template int planar_vjp_cols_launch<double>(bjx_ctx*, int, const double*, const double*, const double*, const double*, int, const double*, const double*, const double*, double*, int64_t, int64_t, double*, double*);
}  // namespace bjx
