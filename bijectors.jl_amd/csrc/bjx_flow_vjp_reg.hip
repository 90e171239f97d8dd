// bjx_flow_vjp_reg.hip — the Float32 register-tile input pullback of the PlanarLayer stack (planar_vjp_reg_kernel,
// planar_vjp_reg2_kernel and their launcher), split from bjx_flow.hip for compile time.
#include "bjx_internal.h"
#include "bjx_tile.h"

namespace {
using namespace bjx;

#include "bjx_flow_common.inc"
#include "bjx_flow_reg.inc"

#ifndef BJX_VJP_REG_WAVES
#define BJX_VJP_REG_WAVES 2
#endif
template <int G, int NL, bool INV, bool UNAL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BJX_VJP_REG_WAVES, 8))) void planar_vjp_reg_kernel(const PlanarRegArgs A, const float* __restrict__ x, const float* __restrict__ ybar,
                                                             const float* __restrict__ lbar, float* __restrict__ xbar, int dim, int64_t batch,
                                                             float* __restrict__ t_out, float* __restrict__ s_out, int nl) {
  constexpr int COLS = 64;
  constexpr int CPS = 64 / G;
  constexpr int NS = (COLS * G) / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* st = reinterpret_cast<float*>(smem) + (size_t)wave * COLS * NL;
  float* tsave = reinterpret_cast<float*>(smem) + (size_t)4 * COLS * NL + (size_t)wave * COLS * A.nl_pad;   // [column][layer]
  const int gl = lane & (G - 1);
  const int cg = lane / G;
  const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * COLS;
  const RegGrid gr = reg_grid<UNAL>(col0 + cg, dim, 4 * gl);      // UNAL: see reg_load_pack
  const bool row_ok = gr.ok;
  const float* Aw = A.w + (UNAL ? A.lead - gr.phi : 0);
  const float* Au = A.u_hat + (UNAL ? A.lead - gr.phi : 0);
  const int64_t left = batch - col0;
  const int nvalid = left >= COLS ? COLS : (left > 0 ? (int)left : 0);
  const int64_t step_elems = (int64_t)CPS * dim;
  bjx_f4 z[NS];
  auto load_tile = [&](const float* base) {
    const float* px = base + (col0 + cg) * dim + gr.rel;
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
  };
  const int ngroups = A.nl_pad / NL;
  // ---- primal sweep: tanh(s_k) (forward map) / tanh(α_k + b_k) (inverse map) of every layer -> tsave
  load_tile(x);
  for (int gi = 0; gi < ngroups; ++gi) {
    const int l0 = (INV ? ngroups - 1 - gi : gi) * NL;         // the inverse undoes the LAST group first
    reg_dots<G, NL, NS>(Aw, l0, (UNAL ? A.ldw : dim), z, st, lane, gl, cg, row_ok);
    __builtin_amdgcn_wave_barrier();
    {
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
          else { if (j > k) a += Gk[j] * t[j]; }                 // t holds -tanh for the inverse
        }
        if (!INV) t[k] = fast_tanh(a + A.b[l0 + k]);
        else { float th, ld; find_alpha_act(a, A.wtu_hat[l0 + k], A.b[l0 + k], th, ld); t[k] = -th; }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NL; ++k) { st[lane * NL + k] = t[k]; tsave[lane * A.nl_pad + l0 + k] = INV ? -t[k] : t[k]; }
    }
    __builtin_amdgcn_wave_barrier();
    if (gi + 1 < ngroups) reg_update<G, NL, NS>(Au, l0, (UNAL ? A.ldw : dim), z, st, gl, cg, row_ok);
    __builtin_amdgcn_wave_barrier();
  }
  // ---- cotangent sweep, in the opposite order of the primal
  load_tile(ybar);
  const float lb = (lbar && lane < nvalid) ? lbar[col0 + lane] : 0.f;
  for (int gi = 0; gi < ngroups; ++gi) {
    const int l0 = (INV ? gi : ngroups - 1 - gi) * NL;
    reg_dots<G, NL, NS>(Au, l0, (UNAL ? A.ldw : dim), z, st, lane, gl, cg, row_ok);
    __builtin_amdgcn_wave_barrier();
    {
      float g[NL], sb[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) { g[k] = st[lane * NL + k]; sb[k] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = INV ? kk : NL - 1 - kk;
        float tb = g[k];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if (INV ? (j < k) : (j > k)) tb += A.G[(int64_t)(l0 + j) * A.nl_pad + l0 + k] * sb[j];   // û_kᵀ w_j
        }
        const float t = tsave[lane * A.nl_pad + l0 + k], c = A.wtu_hat[l0 + k];
        const float q = 1.0f - t * t;
        const float rden = Fast<float>::rcp(1.0f + c * q);
        if (!INV) sb[k] = tb * q + lb * c * (-2.0f * t) * q * rden;
        else sb[k] = q * rden * (-tb + lb * 2.0f * c * t * rden);        // find_alpha rule: dα/d(wᵀy) = 1/(1 + c q)
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NL; ++k) st[lane * NL + k] = sb[k];
      if (s_out && lane < nvalid) {                        // s̄ and tanh of every layer, [n_layers, batch]: input of the parameter pullback
#pragma unroll
        for (int k = 0; k < NL; ++k)
          if (l0 + k < nl) { s_out[(col0 + lane) * nl + l0 + k] = sb[k]; t_out[(col0 + lane) * nl + l0 + k] = tsave[lane * A.nl_pad + l0 + k]; }
      }
    }
    __builtin_amdgcn_wave_barrier();
    reg_update<G, NL, NS>(Aw, l0, (UNAL ? A.ldw : dim), z, st, gl, cg, row_ok);
    __builtin_amdgcn_wave_barrier();
  }
  {
    float* py = xbar + (col0 + cg) * dim + gr.rel;
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
}

// ------------------------------------------------------------------ Planar input pullback, NW waves per tile (64 < dim <= 1024, Float32)
// planar_vjp_reg_kernel on the tile split of planar_reg2_kernel: the rows of a 64-column tile over NW waves (64 rows each, 16 lanes per
// column, 64 VGPRs of tile), partial dot products exchanged through LDS with one block barrier per layer group, the lane = column
// recurrence run redundantly by every wave of the tile on the summed values.  Until round 4 the pullback had the one-wave tile only
// (dim <= 128: 128 VGPRs of tile, 31 % of the HBM peak at 101 rows) and the group kernel beyond (a 64-lane reduction and a tanh per
// layer and column: 15 % at 201 rows).  The partial-sum buffers alternate by a group counter that runs through BOTH sweeps.
// Dynamic LDS: sS [2][NWB][64 NL] | sT [NWB][64 NL] | tsave [tiles][64 nl_pad] (written by the first slice of a tile).
template <int NL, bool INV, int NW, bool UNAL>
__global__ __launch_bounds__(NW <= 4 ? 256 : NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 1, 8))) void planar_vjp_reg2_kernel(const PlanarRegArgs A, const float* __restrict__ x, const float* __restrict__ ybar,
                                                             const float* __restrict__ lbar, float* __restrict__ xbar, int dim, int64_t batch,
                                                             float* __restrict__ t_out, float* __restrict__ s_out, int nl) {
  constexpr int NWB = NW <= 4 ? 4 : NW;
  constexpr int G = 16, COLS = 64, CPS = 4, NS = COLS / CPS, TILES = NWB / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sS = reinterpret_cast<float*>(smem);                          // [2][NWB][COLS * NL]
  float* sT = sS + (size_t)2 * NWB * COLS * NL;                        // [NWB][COLS * NL]
  float* sV = sT + (size_t)NWB * COLS * NL;                            // [TILES][COLS * nl_pad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = wave / NW, half = wave % NW;
  const int gl = lane & (G - 1), cg = lane / G;
  const int row0 = half * 64;
  const int64_t col0 = ((int64_t)blockIdx.x * TILES + tile) * COLS;
  const RegGrid gr = reg_grid<UNAL>(col0 + cg, dim, row0 + 4 * gl);      // UNAL: see reg_load_pack
  const bool row_ok = gr.ok;
  const float* Aw = A.w + (UNAL ? A.lead - gr.phi : 0);
  const float* Au = A.u_hat + (UNAL ? A.lead - gr.phi : 0);
  const int64_t left = batch - col0;
  const int nvalid = left >= COLS ? COLS : (left > 0 ? (int)left : 0);
  const int64_t step_elems = (int64_t)CPS * dim;
  const int ldt = UNAL ? A.ldw : dim;
  float* stT = sT + (size_t)wave * COLS * NL;
  float* tsave = sV + (size_t)tile * COLS * A.nl_pad;
  bjx_f4 z[NS];
  auto load_tile = [&](const float* base) {
    const float* px = base + (col0 + cg) * dim + gr.rel;
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
  };
  // Σ over the tile's slices, in a fixed order: every wave of the tile gets the same bits
  auto gather = [&](int par, float (&s)[NL]) {
    const float* p0 = sS + ((size_t)par * NWB + tile * NW) * COLS * NL + lane * NL;
#pragma unroll
    for (int k = 0; k < NL; ++k) s[k] = p0[k];
#pragma unroll 4
    for (int pp = 1; pp < NW; ++pp) {
#pragma unroll
      for (int k = 0; k < NL; ++k) s[k] += p0[(size_t)pp * COLS * NL + k];
    }
  };
  const int ngroups = A.nl_pad / NL;
  int gc = 0;                                                          // group counter through both sweeps: parity of the partial-sum buffer
  // ---- primal sweep: tanh(s_k) (forward map) / tanh(α_k + b_k) (inverse map) of every layer -> tsave
  load_tile(x);
  for (int gi = 0; gi < ngroups; ++gi, ++gc) {
    const int l0 = (INV ? ngroups - 1 - gi : gi) * NL;
    reg_dots<G, NL, NS>(Aw, l0, ldt, z, sS + ((size_t)(gc & 1) * NWB + wave) * COLS * NL, lane, gl, cg, row_ok, row0);
    __syncthreads();
    {
      float s[NL], t[NL];
      gather(gc & 1, s);
#pragma unroll
      for (int k = 0; k < NL; ++k) t[k] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = INV ? NL - 1 - kk : kk;
        const float* Gk = A.G + (int64_t)(l0 + k) * A.nl_pad + l0;
        float a = s[k];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if (!INV) { if (j < k) a += Gk[j] * t[j]; }
          else { if (j > k) a += Gk[j] * t[j]; }                 // t holds -tanh for the inverse
        }
        if (!INV) t[k] = fast_tanh(a + A.b[l0 + k]);
        else { float th, ld; find_alpha_act(a, A.wtu_hat[l0 + k], A.b[l0 + k], th, ld); t[k] = -th; }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NL; ++k) stT[lane * NL + k] = t[k];
      if (half == 0) {
#pragma unroll
        for (int k = 0; k < NL; ++k) tsave[lane * A.nl_pad + l0 + k] = INV ? -t[k] : t[k];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (gi + 1 < ngroups) reg_update<G, NL, NS>(Au, l0, ldt, z, stT, gl, cg, row_ok, row0);
    __builtin_amdgcn_wave_barrier();
  }
  // ---- cotangent sweep, in the opposite order of the primal (tsave of the first slice is visible after the first barrier below)
  load_tile(ybar);
  const float lb = (lbar && lane < nvalid) ? lbar[col0 + lane] : 0.f;
  for (int gi = 0; gi < ngroups; ++gi, ++gc) {
    const int l0 = (INV ? gi : ngroups - 1 - gi) * NL;
    reg_dots<G, NL, NS>(Au, l0, ldt, z, sS + ((size_t)(gc & 1) * NWB + wave) * COLS * NL, lane, gl, cg, row_ok, row0);
    __syncthreads();
    {
      float g[NL], sb[NL], tk[NL];
      gather(gc & 1, g);
#pragma unroll
      for (int k = 0; k < NL; ++k) { sb[k] = 0.f; tk[k] = tsave[lane * A.nl_pad + l0 + k]; }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = INV ? kk : NL - 1 - kk;
        float tb = g[k];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if (INV ? (j < k) : (j > k)) tb += A.G[(int64_t)(l0 + j) * A.nl_pad + l0 + k] * sb[j];   // û_kᵀ w_j
        }
        const float t = tk[k], c = A.wtu_hat[l0 + k];
        const float q = 1.0f - t * t;
        const float rden = Fast<float>::rcp(1.0f + c * q);
        if (!INV) sb[k] = tb * q + lb * c * (-2.0f * t) * q * rden;
        else sb[k] = q * rden * (-tb + lb * 2.0f * c * t * rden);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NL; ++k) stT[lane * NL + k] = sb[k];
      if (s_out && half == 0 && lane < nvalid) {
#pragma unroll
        for (int k = 0; k < NL; ++k)
          if (l0 + k < nl) { s_out[(col0 + lane) * nl + l0 + k] = sb[k]; t_out[(col0 + lane) * nl + l0 + k] = tk[k]; }
      }
    }
    __builtin_amdgcn_wave_barrier();
    reg_update<G, NL, NS>(Aw, l0, ldt, z, stT, gl, cg, row_ok, row0);
    __builtin_amdgcn_wave_barrier();
  }
  {
    float* py = xbar + (col0 + cg) * dim + gr.rel;
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
}

}  // namespace

namespace bjx {
// returns 1 when the shape is not served
int planar_vjp_reg_launch(bjx_ctx* ctx, int inverse, const float* w, const float* u_hat, const float* wtu, const float* b, int nl, const float* in,
                          const float* out_bar, const float* ladj_bar, float* in_bar, int64_t dim, int64_t batch, float* t_out, float* s_out) {
  static const int use_reg = getenv("BJX_PLANAR_REG") ? atoi(getenv("BJX_PLANAR_REG")) : 1;
  static const int use_unal = getenv("BJX_PLANAR_REG_UNALIGNED") ? atoi(getenv("BJX_PLANAR_REG_UNALIGNED")) : 1;
  static const int unal_nt = 0;
  const bool packs_ok = dim % 4 == 0 && bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
  const bool grid_ok = use_unal && dim > 32 && dim % 4 != 0 && bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);   // reg_load_pack
  const int64_t de = packs_ok ? dim : dim + 3;
  // the tile split over NW waves (planar_vjp_reg2_kernel): 2 for 64 < dim <= 128, 4 to 256, 8 to 512, 16 to 1024.  (Same call, 2^22
  // columns, 8 layers, two waves against the one-wave tile: 72 rows 45.5 / 43.4 %, 101 rows 37.9 / 31.5 %, 128 rows 73.0 / 67.7 %.)
  constexpr int split_env = 1;
  const bool big = de > 256 && de <= 1024 && nl >= 2;
  if (!(use_reg && (packs_ok || grid_ok) && dim > 16 && (de <= 256 || big))) return 1;
  const int NW = de > 512 ? 16 : (de > 256 ? 8 : (de > 128 ? 4 : ((de > 64 && split_env) ? 2 : 1)));
  const int NL = (nl >= 8 && !big) ? 8 : (nl > 2 ? 4 : nl);
  const int nl_pad = (nl + NL - 1) / NL * NL;
  const int lead = packs_ok ? 0 : 4;
  const int64_t ldw = (dim + 3) / 4 * 4 + 2 * lead;
  const size_t off0 = ((size_t)nl * dim + nl + 3) / 4 * 4;
  const size_t need_reg = (off0 + (size_t)2 * nl_pad * ldw + (size_t)nl_pad * nl_pad + 2 * (size_t)nl_pad) * sizeof(float);
  const int NWB = NW <= 4 ? 4 : NW;
  const size_t smem = NW == 1 ? (size_t)4 * 64 * (NL + nl_pad) * sizeof(float)
                              : ((size_t)3 * NWB * 64 * NL + (size_t)(NWB / NW) * 64 * nl_pad) * sizeof(float);
  if (need_reg > BJX_SCRATCH_BYTES || smem > 64 * 1024) return 1;
  float* base = reinterpret_cast<float*>(ctx->scratch);
  float* wp = base + off0;
  float* up = wp + (size_t)nl_pad * ldw;
  float* Gp = up + (size_t)nl_pad * ldw;
  float* cp = Gp + (size_t)nl_pad * nl_pad;
  float* bp = cp + nl_pad;
  hipLaunchKernelGGL(planar_prep_reg_kernel<float>, dim3(nl_pad * nl_pad), dim3(256), 0, ctx->stream, w, u_hat, wtu, b, dim, nl, nl_pad, wp, up, Gp, cp, bp, ldw, lead);
  BJX_CHECK_LAUNCH(ctx);
  const int G = de > 64 ? 32 : (de > 32 ? 16 : 8);
  const int64_t grid = (batch + 4 * 64 - 1) / (4 * 64);
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar_vjp: batch too large for one launch");
  PlanarRegArgs RA{wp, up, Gp, cp, bp, nl_pad, nl, (int)ldw, packs_ok ? 0 : (unal_nt ? 2 : 1), lead};
  if (NW > 1) {
    const int64_t grid2 = (batch + (int64_t)(NWB / NW) * 64 - 1) / ((int64_t)(NWB / NW) * 64);
    BJX_REQUIRE(ctx, grid2 < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_planar_vjp: batch too large for one launch");
#define LV2(NL_, I_, NW_, U_) hipLaunchKernelGGL((planar_vjp_reg2_kernel<NL_, I_, NW_, U_>), dim3((unsigned)grid2), dim3(NW_ <= 4 ? 256 : NW_ * 64), smem, ctx->stream, RA, in, out_bar, ladj_bar, in_bar, (int)dim, batch, t_out, s_out, nl)
#define LV2_U(NL_, I_, NW_) do { if (packs_ok) LV2(NL_, I_, NW_, false); else LV2(NL_, I_, NW_, true); } while (0)
#define LV2_I(NL_, NW_) do { if (inverse) LV2_U(NL_, true, NW_); else LV2_U(NL_, false, NW_); } while (0)
#define LV2_SMALL(NW_) switch (NL) { case 1: LV2_I(1, NW_); break; case 2: LV2_I(2, NW_); break; case 4: LV2_I(4, NW_); break; default: LV2_I(8, NW_); break; }
#define LV2_BIG(NW_) do { if (NL == 4) LV2_I(4, NW_); else LV2_I(2, NW_); } while (0)
    {
      BjxProf prof_(ctx);
      if (NW == 2) { LV2_SMALL(2) } else if (NW == 4) { LV2_SMALL(4) } else if (NW == 8) LV2_BIG(8); else LV2_BIG(16);
    }
#undef LV2_BIG
#undef LV2_SMALL
#undef LV2_I
#undef LV2_U
#undef LV2
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  }
#define LVU(G_, NL_, U_) do { if (inverse) hipLaunchKernelGGL((planar_vjp_reg_kernel<G_, NL_, true, (G_ == 16) && U_>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, RA, in, out_bar, ladj_bar, in_bar, (int)dim, batch, t_out, s_out, nl); \
                          else hipLaunchKernelGGL((planar_vjp_reg_kernel<G_, NL_, false, (G_ == 16) && U_>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, RA, in, out_bar, ladj_bar, in_bar, (int)dim, batch, t_out, s_out, nl); } while (0)
#define LV(G_, NL_) do { if (packs_ok) LVU(G_, NL_, false); else LVU(G_, NL_, true); } while (0)
#define LV_NL(G_) switch (NL) { case 1: LV(G_, 1); break; case 2: LV(G_, 2); break; case 4: LV(G_, 4); break; default: LV(G_, 8); break; }
  {
    BjxProf prof_(ctx);
    switch (G) { case 8: LV_NL(8) break; case 16: LV_NL(16) break; default: LV_NL(32) break; }
  }
#undef LV_NL
#undef LV
#undef LVU
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

}  // namespace bjx
