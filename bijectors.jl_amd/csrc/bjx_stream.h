// bjx_stream.h — column-group streaming skeleton shared by the per-element bijectors that need
// a per-sample log-det (RQS, BatchNorm, Coupling) and by Permute.
//
// Mapping (SURVEY.md §7 family F2/F4): a column (sample) is `dim` contiguous elements, so G
// consecutive lanes own one column and read it as 16-byte packs -> every wave instruction covers a
// contiguous 1 KiB run of HBM.  The functor F transforms a pack in registers and returns the pack's
// log-det contribution; the G lanes are summed with a butterfly of wave shuffles, lane 0 of the
// group writes ladj_ps[col], and the block publishes one f64 partial for the global sum.
#pragma once
#include <cstdlib>

#include "bjx_internal.h"
#include "bjx_tile.h"

namespace bjx {

constexpr int STREAM_U = 4;

// F interface:
//   __device__ void stage(char* smem) const;                       // cooperative LDS staging (+sync)
//   template <int V> __device__ T apply(const char* smem, Pack<T,V>& p, const T* xcol, int64_t row, int64_t col) const;
//       row = index of p.v[0] inside the column; xcol = start of the input column (for gathers)
//   static constexpr bool kLoadInput   (false: apply() gathers from xcol itself, e.g. Permute)
//   double per_sample_const            (host-known constant added to every ladj_ps entry)
//   const double* per_sample_dev       (device constant added likewise, or null)
//   optional: `using Aux = ...; template <int V> Aux fetch(const char* smem, int64_t row, int64_t col) const;`
//       per-pack operands the functor reads from HBM (Coupling's θ arrays): fetched together with the
//       input packs so all loads of a lane are in flight at once; apply() then takes `const Aux&` after `p`
template <class F, class = void> struct col_has_aux { static constexpr bool value = false; };
template <class F> struct col_has_aux<F, decltype((void)sizeof(typename F::Aux))> { static constexpr bool value = true; };
template <class F, class = void> struct col_has_multi { static constexpr bool value = false; };
template <class F> struct col_has_multi<F, decltype((void)F::kMulti)> { static constexpr bool value = F::kMulti; };
template <class F, class = void> struct col_has_masked { static constexpr bool value = false; };
template <class F> struct col_has_masked<F, decltype((void)F::kMasked)> { static constexpr bool value = F::kMasked; };
struct ColNoAux {};
template <class F, bool H = col_has_aux<F>::value> struct col_aux_of { using type = ColNoAux; };
template <class F> struct col_aux_of<F, true> { using type = typename F::Aux; };

constexpr int COL_UC = 4;   // columns in flight per lane group (one 16-byte pack each) when a column fits in G packs

template <class T, int V, bool NT, class F, bool TAIL = false>
__global__ __launch_bounds__(256) void colgroup_kernel(const F f, const T* x, T* y, T* ladj_ps, int64_t dim,
                                                       int64_t batch, int G, int accumulate, const BjxFin fin, int64_t ldx, int64_t ldy, int64_t row0) {
  // row0: first row of a row window (x and y already point at it; the functor and its gathers see the rows of the whole column)
  // ldx / ldy: leading dimensions of x / y (== dim for a dense [dim, batch] array; larger when the `dim` rows are a
  // window of a taller matrix — Stacked segments, stacked.jl:142-166)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);   // first 32 bytes
  char* fsm = smem + 32;
  f.stage(fsm);
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = blockDim.x / G;
  const int64_t nvc = dim / V;
  double acc = 0.0;
  // non-persistent: COL_UC columns per G-lane group (scripts/membench.hip: in-order short blocks stream
  // ~30 % faster than grid-stride loops on MI355X; one pack per lane in flight is latency-bound)
  const int64_t col0 = (int64_t)blockIdx.x * cols_per_block * COL_UC + threadIdx.x / G;
  const double psc = f.per_sample_const + (f.per_sample_dev ? *f.per_sample_dev : 0.0);
  // Column heights that are not whole packs (dim = 101, 201, 1001 ...; V > 1 only): the packs are then ELEMENT-aligned — global
  // accesses take that — and the last tail = dim % V rows go one row per lane on the lanes after the last pack's (the launcher
  // sizes G for nvc + tail lanes when the column fits one pack per lane).
  // TAIL is a template flag: with the tail code compiled into the whole-pack instantiation, Coupling and Stacked at 252 rows lost a
  // fifth of their rate (56 -> 46 %, 63 -> 51 %) to the second functor call's registers.
  const int tail = (V > 1 && TAIL) ? (int)(dim - nvc * V) : 0;
  if (nvc + tail <= G) {
    Pack<T, V> p[COL_UC];
    typename col_aux_of<F>::type aux[COL_UC];
    const bool lane_ok = gl < nvc;
    const bool tail_ok = V > 1 && TAIL && gl >= nvc && gl < nvc + tail;
    const int64_t trow = nvc * V + (gl - nvc);                   // the tail lane's row
#pragma unroll
    for (int u = 0; u < COL_UC; ++u) {
      const int64_t col = col0 + (int64_t)u * cols_per_block;
      if (F::kLoadInput && lane_ok && col < batch) p[u] = load_pack<T, V, NT>(x + col * ldx + (int64_t)gl * V);
      if constexpr (col_has_aux<F>::value) { if (lane_ok && col < batch) aux[u] = f.template fetch<V>(fsm, row0 + (int64_t)gl * V, col); }
      if constexpr (V > 1 && TAIL) {
        if (tail_ok && col < batch) {
          if (F::kLoadInput) p[u].v[0] = x[col * ldx + trow];
          if constexpr (col_has_aux<F>::value) aux[u] = f.template fetch<1>(fsm, row0 + trow, col);
        }
      }
    }
    T lm[COL_UC];
    if constexpr (V > 1 && TAIL) {
      if (tail_ok) {
        Pack<T, 1> q[COL_UC];
#pragma unroll
        for (int u = 0; u < COL_UC; ++u) q[u].v[0] = (col0 + (int64_t)u * cols_per_block < batch) ? p[u].v[0] : p[0].v[0];
        if constexpr (col_has_multi<F>::value) {
          if (col0 < batch) f.template apply_multi<1, COL_UC>(fsm, q, row0 + trow, lm);
        } else {
#pragma unroll
          for (int u = 0; u < COL_UC; ++u) {
            const int64_t col = col0 + (int64_t)u * cols_per_block;
            lm[u] = T(0);
            if (col < batch) {
              if constexpr (col_has_aux<F>::value) lm[u] = f.template apply<1>(fsm, q[u], aux[u], x + col * ldx - row0, row0 + trow, col);
              else lm[u] = f.template apply<1>(fsm, q[u], x + col * ldx - row0, row0 + trow, col);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < COL_UC; ++u) p[u].v[0] = q[u].v[0];
      }
    }
    if constexpr (col_has_multi<F>::value) {
      // optional `apply_multi<V,U>(smem, p[U], row, l[U])`: the COL_UC packs of a lane sit at the same rows
      if (lane_ok) {
        if (col0 + (int64_t)(COL_UC - 1) * cols_per_block >= batch) {      // ragged end: harmless inputs for the missing columns
#pragma unroll
          for (int u = 0; u < COL_UC; ++u)
            if (col0 + (int64_t)u * cols_per_block >= batch) { p[u] = p[0]; }
        }
        if (col0 < batch) f.template apply_multi<V, COL_UC>(fsm, p, row0 + (int64_t)gl * V, lm);
      }
    }
#pragma unroll
    for (int u = 0; u < COL_UC; ++u) {
      const int64_t col = col0 + (int64_t)u * cols_per_block;
      T l = T(0);
      if (lane_ok && col < batch) {
        if constexpr (col_has_multi<F>::value) l = lm[u];
        else if constexpr (col_has_aux<F>::value) l = f.template apply<V>(fsm, p[u], aux[u], x + col * ldx - row0, row0 + (int64_t)gl * V, col);
        else l = f.template apply<V>(fsm, p[u], x + col * ldx - row0, row0 + (int64_t)gl * V, col);
        store_pack<T, V, NT>(y + col * ldy + (int64_t)gl * V, p[u]);
      }
      if constexpr (V > 1 && TAIL) {
        if (tail_ok && col < batch) { l = lm[u]; y[col * ldy + trow] = p[u].v[0]; }
      }
      l = group_sum_rt(l, G);
      if (col < batch && gl == 0) {
        if (ladj_ps) {
          T out = l + (T)psc;
          if (accumulate) out += ladj_ps[col];
          ladj_ps[col] = out;
        }
        acc += (double)l;
      }
    }
  } else if (!TAIL && nvc <= 2 * (int64_t)G) {
    // up to two packs per lane: TWO columns in flight (see colgroup_tail_kernel)
    for (int uc = 0; uc < COL_UC; uc += 2) {
      Pack<T, V> p[2][2];
      typename col_aux_of<F>::type aux[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int64_t col = col0 + (int64_t)(uc + c) * cols_per_block;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int64_t v = (int64_t)r * G + gl;
          if (F::kLoadInput && col < batch && v < nvc) p[c][r] = load_pack<T, V, NT>(x + col * ldx + v * V);
          if constexpr (col_has_aux<F>::value) { if (col < batch && v < nvc) aux[c][r] = f.template fetch<V>(fsm, row0 + v * V, col); }
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int64_t col = col0 + (int64_t)(uc + c) * cols_per_block;
        T l = T(0);
        if (col < batch) {
          const T* xc = x + col * ldx;
          T* yc = y + col * ldy;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int64_t v = (int64_t)r * G + gl;
            if (v < nvc) {
              if constexpr (col_has_aux<F>::value) l += f.template apply<V>(fsm, p[c][r], aux[c][r], xc - row0, row0 + v * V, col);
              else l += f.template apply<V>(fsm, p[c][r], xc - row0, row0 + v * V, col);
              store_pack<T, V, NT>(yc + v * V, p[c][r]);
            }
          }
        }
        l = group_sum_rt(l, G);
        if (col < batch && gl == 0) {
          if (ladj_ps) {
            T out = l + (T)psc;
            if (accumulate) out += ladj_ps[col];
            ladj_ps[col] = out;
          }
          acc += (double)l;
        }
      }
    }
  } else {
    for (int uc = 0; uc < COL_UC; ++uc) {
      const int64_t col = col0 + (int64_t)uc * cols_per_block;
      T l = T(0);
      if (col < batch) {
        const T* xc = x + col * ldx;
        T* yc = y + col * ldy;
        for (int64_t v0 = 0; v0 < nvc; v0 += (int64_t)G * STREAM_U) {
          Pack<T, V> p[STREAM_U];
          typename col_aux_of<F>::type aux[STREAM_U];
#pragma unroll
          for (int u = 0; u < STREAM_U; ++u) {
            int64_t v = v0 + (int64_t)u * G + gl;
            if (F::kLoadInput && v < nvc) p[u] = load_pack<T, V, NT>(xc + v * V);
            if constexpr (col_has_aux<F>::value) { if (v < nvc) aux[u] = f.template fetch<V>(fsm, row0 + v * V, col); }
          }
#pragma unroll
          for (int u = 0; u < STREAM_U; ++u) {
            int64_t v = v0 + (int64_t)u * G + gl;
            if (v < nvc) {
              if constexpr (col_has_aux<F>::value) l += f.template apply<V>(fsm, p[u], aux[u], xc - row0, row0 + v * V, col);
              else l += f.template apply<V>(fsm, p[u], xc - row0, row0 + v * V, col);
              store_pack<T, V, NT>(yc + v * V, p[u]);
            }
          }
        }
        if constexpr (V > 1 && TAIL) {
          const int tl = (int)((gl - nvc) & (G - 1));              // tail row t goes to lane (nvc + t) % G
          if (tl < tail) {
            const int64_t trow = nvc * V + tl;
            Pack<T, 1> q;
            if (F::kLoadInput) q.v[0] = xc[trow];
            if constexpr (col_has_aux<F>::value) { const typename col_aux_of<F>::type a1 = f.template fetch<1>(fsm, row0 + trow, col); l += f.template apply<1>(fsm, q, a1, xc - row0, row0 + trow, col); }
            else l += f.template apply<1>(fsm, q, xc - row0, row0 + trow, col);
            yc[trow] = q.v[0];
          }
        }
      }
      l = group_sum_rt(l, G);
      if (col < batch && gl == 0) {
        if (ladj_ps) {
          T out = l + (T)psc;
          if (accumulate) out += ladj_ps[col];
          ladj_ps[col] = out;
        }
        acc += (double)l;
      }
    }
  }
  block_publish_partial(acc, red, fin);
}


// Odd column heights for functors that take ANY first row and a row mask (`kMasked`: apply_masked / apply_multi_masked): the dim % V
// tail rows are one more unit of the SAME code path — the lane after the last whole pack reads the LAST V rows of the column
// (element-aligned, overlapping the pack before it), evaluates all V of them, and keeps only its own: the mask drops the overlap's
// log-det terms and the store writes the tail rows alone.  No second functor call (colgroup_kernel<..., TAIL = true> runs the tail
// through a one-row call behind a divergent branch: BatchNorm / Stacked 43-46 % of the HBM peak at 101 / 201 rows against 62-67 %
// at 252).  In place: the overlap may already hold outputs when the tail unit runs in a later trip; those values are never used.
template <class T, int V, class F>
__global__ __launch_bounds__(256) void colgroup_tail_kernel(const F f, const T* x, T* y, T* ladj_ps, int64_t dim,
                                                            int64_t batch, int G, int accumulate, const BjxFin fin, int64_t ldx, int64_t ldy, int64_t row0, int nt) {
  // row0: first row of a row window (x and y already point at it; the functor and its gathers see the rows of the whole column)
  // nt: streaming (nontemporal) stores.  OFF by default: columns of an odd height end and begin inside a 64-byte sector, and the two
  // halves are written by different instructions at different times — with streaming stores each half goes to HBM as a partial write
  // (WRITE_SIZE 1.15 x the output at 101 rows, `profiles/r03_odd_counters.md`); ordinary stores let the L2 merge them.
  static_assert(V > 1, "whole packs only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  char* fsm = smem + 32;
  f.stage(fsm);
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = blockDim.x / G;
  const int64_t nvc = dim / V;
  const int tail = (int)(dim - nvc * V);                          // 1 .. V-1
  const int64_t nun = nvc + 1;                                    // units of a column: the whole packs and the tail
  constexpr uint32_t full = (1u << V) - 1u;
  const uint32_t tmask = (full << (V - tail)) & full;             // the tail unit's own rows are the LAST `tail` of its pack
  double acc = 0.0;
  const int64_t col0 = (int64_t)blockIdx.x * cols_per_block * COL_UC + threadIdx.x / G;
  const double psc = f.per_sample_const + (f.per_sample_dev ? *f.per_sample_dev : 0.0);
  if (nun <= G) {
    Pack<T, V> p[COL_UC];
    typename col_aux_of<F>::type aux[COL_UC];
    const bool lane_ok = gl < nun;
    const bool is_tail = gl == nvc;
    const int64_t prow = is_tail ? dim - V : (int64_t)gl * V;
    const uint32_t mask = is_tail ? tmask : full;
#pragma unroll
    for (int u = 0; u < COL_UC; ++u) {
      const int64_t col = col0 + (int64_t)u * cols_per_block;
      if (F::kLoadInput && lane_ok && col < batch) p[u] = load_pack<T, V, true>(x + col * ldx + prow);
      if constexpr (col_has_aux<F>::value) { if (lane_ok && col < batch) aux[u] = f.template fetch<V>(fsm, row0 + prow, col); }
    }
    T lm[COL_UC];
    if constexpr (col_has_multi<F>::value) {
      if (lane_ok) {
        if (col0 + (int64_t)(COL_UC - 1) * cols_per_block >= batch) {
#pragma unroll
          for (int u = 0; u < COL_UC; ++u)
            if (col0 + (int64_t)u * cols_per_block >= batch) { p[u] = p[0]; }
        }
        if (col0 < batch) f.template apply_multi_masked<V, COL_UC>(fsm, p, row0 + prow, lm, mask);
      }
    }
#pragma unroll
    for (int u = 0; u < COL_UC; ++u) {
      const int64_t col = col0 + (int64_t)u * cols_per_block;
      T l = T(0);
      if (lane_ok && col < batch) {
        if constexpr (col_has_multi<F>::value) l = lm[u];
        else if constexpr (col_has_aux<F>::value) l = f.template apply_masked<V>(fsm, p[u], aux[u], x + col * ldx - row0, row0 + prow, col, mask);
        else l = f.template apply_masked<V>(fsm, p[u], x + col * ldx - row0, row0 + prow, col, mask);
        T* yp = y + col * ldy + prow;
        if (!is_tail) { if (nt) store_pack<T, V, true>(yp, p[u]); else store_pack<T, V, false>(yp, p[u]); }
        else store_pack_run<T, V>(yp, p[u], V - tail, tail);
      }
      l = group_sum_rt(l, G);
      if (col < batch && gl == 0) {
        if (ladj_ps) {
          T out = l + (T)psc;
          if (accumulate) out += ladj_ps[col];
          ladj_ps[col] = out;
        }
        acc += (double)l;
      }
    }
  } else if (nun <= 2 * (int64_t)G) {
    // up to two units per lane: TWO columns in flight (one column alone is a single round trip of 1-2 KiB per wave: 333 rows ran at
    // 24-38 % of the HBM peak, two thirds of what 500 rows get from the loop below)
    for (int uc = 0; uc < COL_UC; uc += 2) {
      Pack<T, V> p[2][2];
      typename col_aux_of<F>::type aux[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int64_t col = col0 + (int64_t)(uc + c) * cols_per_block;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int64_t v = (int64_t)r * G + gl;
          const int64_t prow = v == nvc ? dim - V : v * V;
          if (F::kLoadInput && col < batch && v < nun) p[c][r] = load_pack<T, V, true>(x + col * ldx + prow);
          if constexpr (col_has_aux<F>::value) { if (col < batch && v < nun) aux[c][r] = f.template fetch<V>(fsm, row0 + prow, col); }
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int64_t col = col0 + (int64_t)(uc + c) * cols_per_block;
        T l = T(0);
        if (col < batch) {
          const T* xc = x + col * ldx;
          T* yc = y + col * ldy;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int64_t v = (int64_t)r * G + gl;
            if (v < nun) {
              const bool is_tail = v == nvc;
              const int64_t prow = is_tail ? dim - V : v * V;
              const uint32_t mask = is_tail ? tmask : full;
              if constexpr (col_has_aux<F>::value) l += f.template apply_masked<V>(fsm, p[c][r], aux[c][r], xc - row0, row0 + prow, col, mask);
              else l += f.template apply_masked<V>(fsm, p[c][r], xc - row0, row0 + prow, col, mask);
              if (!is_tail) { if (nt) store_pack<T, V, true>(yc + prow, p[c][r]); else store_pack<T, V, false>(yc + prow, p[c][r]); }
              else store_pack_run<T, V>(yc + prow, p[c][r], V - tail, tail);
            }
          }
        }
        l = group_sum_rt(l, G);
        if (col < batch && gl == 0) {
          if (ladj_ps) {
            T out = l + (T)psc;
            if (accumulate) out += ladj_ps[col];
            ladj_ps[col] = out;
          }
          acc += (double)l;
        }
      }
    }
  } else {
    for (int uc = 0; uc < COL_UC; ++uc) {
      const int64_t col = col0 + (int64_t)uc * cols_per_block;
      T l = T(0);
      if (col < batch) {
        const T* xc = x + col * ldx;
        T* yc = y + col * ldy;
        for (int64_t v0 = 0; v0 < nun; v0 += (int64_t)G * STREAM_U) {
          Pack<T, V> p[STREAM_U];
          typename col_aux_of<F>::type aux[STREAM_U];
#pragma unroll
          for (int u = 0; u < STREAM_U; ++u) {
            const int64_t v = v0 + (int64_t)u * G + gl;
            const int64_t prow = v == nvc ? dim - V : v * V;
            if (F::kLoadInput && v < nun) p[u] = load_pack<T, V, true>(xc + prow);
            if constexpr (col_has_aux<F>::value) { if (v < nun) aux[u] = f.template fetch<V>(fsm, row0 + prow, col); }
          }
#pragma unroll
          for (int u = 0; u < STREAM_U; ++u) {
            const int64_t v = v0 + (int64_t)u * G + gl;
            if (v < nun) {
              const bool is_tail = v == nvc;
              const int64_t prow = is_tail ? dim - V : v * V;
              const uint32_t mask = is_tail ? tmask : full;
              if constexpr (col_has_aux<F>::value) l += f.template apply_masked<V>(fsm, p[u], aux[u], xc - row0, row0 + prow, col, mask);
              else l += f.template apply_masked<V>(fsm, p[u], xc - row0, row0 + prow, col, mask);
              if (!is_tail) { if (nt) store_pack<T, V, true>(yc + prow, p[u]); else store_pack<T, V, false>(yc + prow, p[u]); }
              else store_pack_run<T, V>(yc + prow, p[u], V - tail, tail);
            }
          }
        }
      }
      l = group_sum_rt(l, G);
      if (col < batch && gl == 0) {
        if (ladj_ps) {
          T out = l + (T)psc;
          if (accumulate) out += ladj_ps[col];
          ladj_ps[col] = out;
        }
        acc += (double)l;
      }
    }
  }
  block_publish_partial(acc, red, fin);
}

// Column-walker form of the same skeleton, for columns that are NOT a whole number of 16-byte packs (dim = 3, 10, 13 ...): the group
// kernel reads such columns 4 bytes at a time.  One single-wave block moves 64 consecutive columns — one contiguous run, 16-byte
// packs — through a [64][P odd] LDS tile and lane t walks column t with the functor's one-element `apply`; the row is the same
// in every lane, so the functor's per-row parameters are wave-uniform reads, and the column's log-det is the lane's own sum.
template <class T, int V, class F>
__global__ __launch_bounds__(64) void colwalk_kernel(const F f, const T* x, T* y, T* ladj_ps, int dim, int P, int64_t batch, int accumulate, const BjxFin fin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);   // first 32 bytes
  char* fsm = smem + 32;
  f.stage(fsm);
  T* tile = reinterpret_cast<T*>(smem + 32 + f.walk_smem_offset);
  const int lane = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
  const double psc = f.per_sample_const + (f.per_sample_dev ? *f.per_sample_dev : 0.0);
  if (F::kLoadInput) tile_stage_in<T, V>(tile, x + c0 * dim, dim, P, ncols, lane);
  tile_sync();
  T l = T(0);
  if (lane < ncols) {
    T* mine = tile + lane * P;
    const int64_t col = c0 + lane;
    const T* xcol = x + col * dim;
    for (int r = 0; r < dim; ++r) {
      Pack<T, 1> p;
      p.v[0] = F::kLoadInput ? mine[r] : T(0);
      if constexpr (col_has_aux<F>::value) { const auto aux = f.template fetch<1>(fsm, (int64_t)r, col); l += f.template apply<1>(fsm, p, aux, xcol, (int64_t)r, col); }
      else l += f.template apply<1>(fsm, p, xcol, (int64_t)r, col);
      mine[r] = p.v[0];
    }
    if (ladj_ps) {
      T out = l + (T)psc;
      if (accumulate) out += ladj_ps[col];
      ladj_ps[col] = out;
    }
  }
  tile_sync();
  tile_stage_out<T, V>(tile, y + c0 * dim, dim, P, ncols, lane);
  block_publish_partial(lane < ncols ? (double)l : 0.0, red, fin);
}

// The same for SHORT columns (DX = dim <= 7 rows, not whole packs): lane = column, the column read and written by its lane as one
// or two multi-dword accesses (TinyCol) — no tile, no staging; 256-thread blocks, the functor's tables staged once per block.  At
// dim = 2 ... 5 the walker above spends its time on the tile, not on the map (profiles/r03_small_sizes.md: 22-42 % whatever the functor).
template <class T, int DX, class F>
__global__ __launch_bounds__(256) void coldirect_kernel(const F f, const T* x, T* y, T* ladj_ps, int64_t batch, int accumulate, const BjxFin fin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);   // first 32 bytes
  char* fsm = smem + 32;
  f.stage(fsm);
  __syncthreads();
  const double psc = f.per_sample_const + (f.per_sample_dev ? *f.per_sample_dev : 0.0);
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  T l = T(0);
  if (col < batch) {
    const T* xcol = x + col * DX;
    TinyCol<T, DX> t{};
    if (F::kLoadInput) t = *reinterpret_cast<const TinyCol<T, DX>*>(xcol);
#pragma unroll
    for (int r = 0; r < DX; ++r) {
      Pack<T, 1> p;
      p.v[0] = t.v[r];
      if constexpr (col_has_aux<F>::value) { const auto aux = f.template fetch<1>(fsm, (int64_t)r, col); l += f.template apply<1>(fsm, p, aux, xcol, (int64_t)r, col); }
      else l += f.template apply<1>(fsm, p, xcol, (int64_t)r, col);
      t.v[r] = p.v[0];
    }
    *reinterpret_cast<TinyCol<T, DX>*>(y + col * DX) = t;
    if (ladj_ps) {
      T out = l + (T)psc;
      if (accumulate) out += ladj_ps[col];
      ladj_ps[col] = out;
    }
  }
  block_publish_partial(col < batch ? (double)l : 0.0, red, fin);
}

struct ColLaunch {
  int V, G;
  int64_t grid;
  int unal;       // V-wide packs on columns that are only ELEMENT-aligned (odd heights): tail rows one per lane, see colgroup_kernel
};

// choose pack width / lanes per column / grid for a [dim, batch] problem
template <class T> inline ColLaunch col_launch_cfg(const bjx_ctx* ctx, const void* x, const void* y, int64_t dim, int64_t batch, int64_t ldx = 0,
                                                   int64_t ldy = 0, bool allow_unal = false, int unal_from = 0) {
  ColLaunch c;
  constexpr int VW = Vec16<T>::N;
  const bool v_ok = bjx_aligned16(x) && bjx_aligned16(y) && dim % VW == 0 && ldx % VW == 0 && ldy % VW == 0;
  c.V = v_ok ? VW : 1;
  c.unal = 0;
  int64_t packs = dim / c.V;
  // Columns that are not whole aligned packs and too tall for the tile walker to keep its occupancy (same-box A/B at 101 / 201
  // rows, % of the HBM peak: BatchNorm 32 / 15 -> 44 / 45, Stacked 36 / 20 -> 43 / 41, Coupling 36 / 21 -> 29 / 31; at 49 / 63 rows the
  // walker still wins, 58 / 54 and 54 / 49 against 56 / 46 and 40 / 42; at 77 rows it is 43 / 46 against 50 / 50): 16-byte packs on element-aligned addresses, the dim % V tail rows on one lane each.
  // Callers that build V-permuted tables must ask with the same flag.
  static const int use_unal = getenv("BJX_COL_UNALIGNED") ? atoi(getenv("BJX_COL_UNALIGNED")) : 1;
  const int unal_min = unal_from > 0 ? unal_from : 80;      // (the pullback of chains asks from fewer rows: see stacked_vjp_impl)
  const bool window = ldx > dim && ldy > dim;             // a row window of taller arrays (slabs): no tile walker to fall back on
  if (allow_unal && use_unal && !v_ok && dim >= (window ? 2 * VW : unal_min) && dim >= VW) {
    c.V = VW;
    c.unal = 1;
    packs = dim / VW + dim % VW;                       // lanes a column needs at one pack (or one tail row) per lane
  }
  int G = 1;
  while (G < 64 && G < packs) G <<= 1;
  c.G = G;
  (void)ctx;
  const int64_t cpb = (int64_t)(256 / G) * COL_UC;   // columns per block
  c.grid = (batch + cpb - 1) / cpb;
  return c;
}

template <class T, class F>
inline int launch_colgroup(bjx_ctx* ctx, const F& f, size_t f_smem, const T* x, T* y, T* ladj_ps, double* ladj_sum,
                           int64_t dim, int64_t batch, uint32_t flags, double sum_const, int64_t ldx = 0, int64_t ldy = 0, int force_v1 = 0,
                           int64_t row0 = 0) {
  // row0 > 0: `dim` rows starting at row row0 of columns that are ldx / ldy apart (x and y point at row 0; see the slabs below)
  const bool dense = ldx == 0 && ldy == 0;
  if (ldx == 0) ldx = dim;
  if (ldy == 0) ldy = dim;
  if (dim * batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  {
    // Columns of more than 64 packs: ROW SLABS of 64 packs.  One pack per lane is the form that keeps four columns in flight per lane
    // and shares a row's parameters between them (apply_multi); beyond it a lane walks its column's packs one latency round trip at
    // a time (BatchNorm 39 % of the HBM peak at 257 rows, 46 % at 333; `Stacked`, which slices its table as well, went 24 -> 51 % at
    // 300 rows with the same step).  A slab is a launch on a row window of the same arrays; the functor keeps seeing the rows of the
    // whole column (row0), the log-dets of the slabs accumulate in launch order (deterministic).
    static const int use_slab = getenv("BJX_COL_SLAB") ? atoi(getenv("BJX_COL_SLAB")) : 1;
    constexpr int VWs = Vec16<T>::N;
    const int64_t slab = (int64_t)64 * VWs;
    // (up to 96 packs only — same-box A/B, % of the HBM peak, slabs / column loop: BatchNorm 257 rows 53 / 39, 300 rows 53 / 52,
    //  500 rows 57 / 61; Coupling 300 rows 47 / 35, 500 rows 49 / 50: every block stages the functor's whole table, and from two
    //  packs per lane on the column loop has two columns in flight anyway)
    // Every slab window must resolve to the pack width of the whole-column call: functors with a V-permuted table (StackedF) were
    // built for THAT width, and a short last slab (fewer than two packs of rows: Float32 dim 261-263, Float64 dim 131) would fall
    // to V = 1 and read the table at the wrong slots.  Such heights take the single launch.
    auto slabs_keep_v = [&]() {
      for (int64_t r0 = 0; r0 < dim; r0 += slab) {
        const int64_t rs = dim - r0 < slab ? dim - r0 : slab;
        if (col_launch_cfg<T>(ctx, x + r0, y + r0, rs, batch, dim, dim, true).V != VWs) return false;
      }
      return true;
    };
    if (use_slab && dense && row0 == 0 && !force_v1 && dim > slab && dim <= slab + slab / 2 && col_launch_cfg<T>(ctx, x, y, dim, batch, ldx, ldy, true).V == VWs && slabs_keep_v()) {
      for (int64_t r0 = 0; r0 < dim; r0 += slab) {
        const int64_t rs = dim - r0 < slab ? dim - r0 : slab;
        F fw = f;
        if (r0 > 0) { fw.per_sample_const = 0.0; fw.per_sample_dev = nullptr; }
        const uint32_t fl = r0 == 0 ? flags : (flags | BJX_ACCUMULATE);
        const int rc = launch_colgroup<T, F>(ctx, fw, f_smem, x, y, ladj_ps, ladj_sum, rs, batch, fl, r0 == 0 ? sum_const : 0.0, dim, dim, 0, r0 == 0 ? -1 : r0);
        if (rc) return rc;
      }
      return BJX_OK;
    }
    if (row0 < 0) row0 = 0;                              // (-1: the first slab — row 0, but a window all the same)
  }
  x += row0;
  y += row0;
  ColLaunch c = col_launch_cfg<T>(ctx, x, y, dim, batch, ldx, ldy, !force_v1);
  {
    // odd column heights of dense arrays: the column-walker form (colwalk_kernel)
    static const int use_walk = getenv("BJX_COLWALK") ? atoi(getenv("BJX_COLWALK")) : 1;
    constexpr int VWW = Vec16<T>::N;
    const int64_t P = dim | 1;
    const size_t f_pad = (f_smem + 15) / 16 * 16;
    const size_t smem_w = 32 + f_pad + (size_t)64 * P * sizeof(T);
    static const int use_direct = getenv("BJX_COLDIRECT") ? atoi(getenv("BJX_COLDIRECT")) : 1;
    if (use_direct && !force_v1 && dim % VWW != 0 && dim <= 7 && ldx == dim && ldy == dim && 32 + f_smem <= 64 * 1024) {
      const int64_t grid_d = (batch + 255) / 256;
      BJX_REQUIRE(ctx, grid_d < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
      BjxFin fin;
      bool second = false;
      { int rc = bjx_make_fin(ctx, grid_d, ladj_sum, sum_const, f.per_sample_dev ? 1 : 0, flags, &fin, &second); if (rc) return rc; }
      const int accum_d = (flags & BJX_ACCUMULATE) ? 1 : 0;
#define BJX_CD(X_) hipLaunchKernelGGL((coldirect_kernel<T, X_, F>), dim3((unsigned)grid_d), dim3(256), 32 + f_smem, ctx->stream, f, x, y, ladj_ps, batch, accum_d, fin)
      {
        BjxProf prof_(ctx);
        switch ((int)dim) {
          case 1: BJX_CD(1); break;
          case 2: BJX_CD(2); break;
          case 3: BJX_CD(3); break;
          case 5: BJX_CD(5); break;
          case 6: BJX_CD(6); break;
          default: BJX_CD(7); break;
        }
      }
#undef BJX_CD
      BJX_CHECK_LAUNCH(ctx);
      if (second) return bjx_launch_finalize(ctx, (int)grid_d, ladj_sum, sum_const, f.per_sample_dev ? 1 : 0, 0.0, flags);
      return BJX_OK;
    }
    if (use_walk && !force_v1 && !c.unal && dim % VWW != 0 && ldx == dim && ldy == dim && (const void*)x != (const void*)y && smem_w <= 64 * 1024) {
      const int64_t grid_w = (batch + 63) / 64;
      BJX_REQUIRE(ctx, grid_w < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
      BjxFin fin;
      bool second = false;
      { int rc = bjx_make_fin(ctx, grid_w, ladj_sum, sum_const, f.per_sample_dev ? 1 : 0, flags, &fin, &second); if (rc) return rc; }
      if (fin.counter) { fin.counter = nullptr; second = true; }        // single-wave blocks: two-pass finalize
      F fw = f;
      fw.walk_smem_offset = (int)f_pad;
      const bool vec = bjx_aligned16(x) && bjx_aligned16(y);
      {
        BjxProf prof_(ctx);
        if (vec) hipLaunchKernelGGL((colwalk_kernel<T, VWW, F>), dim3((unsigned)grid_w), dim3(64), smem_w, ctx->stream, fw, x, y, ladj_ps, (int)dim, (int)P, batch, (flags & BJX_ACCUMULATE) ? 1 : 0, fin);
        else hipLaunchKernelGGL((colwalk_kernel<T, 1, F>), dim3((unsigned)grid_w), dim3(64), smem_w, ctx->stream, fw, x, y, ladj_ps, (int)dim, (int)P, batch, (flags & BJX_ACCUMULATE) ? 1 : 0, fin);
      }
      BJX_CHECK_LAUNCH(ctx);
      if (second) return bjx_launch_finalize(ctx, (int)grid_w, ladj_sum, sum_const, f.per_sample_dev ? 1 : 0, 0.0, flags);
      return BJX_OK;
    }
  }
  if (force_v1 && c.V != 1) { c.V = 1; c.unal = 0; int G = 1; while (G < 64 && G < dim) G <<= 1; c.G = G; const int64_t cpb = (int64_t)(256 / G) * COL_UC; c.grid = (batch + cpb - 1) / cpb; }
  constexpr int VW = Vec16<T>::N;
  const size_t smem = 32 + f_smem;
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  BJX_REQUIRE(ctx, c.grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  BjxFin fin;
  bool second = false;
  { int rc = bjx_make_fin(ctx, c.grid, ladj_sum, sum_const, f.per_sample_dev ? 1 : 0, flags, &fin, &second); if (rc) return rc; }
  {
  BjxProf prof_(ctx);
  if (c.V == VW && c.unal && dim % VW != 0) {
    static const int unal_nt = 0;
    if constexpr (col_has_masked<F>::value && Vec16<T>::N > 1)
      hipLaunchKernelGGL((colgroup_tail_kernel<T, VW, F>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, f, x, y, ladj_ps, dim, batch, c.G, accum, fin, ldx, ldy, row0, unal_nt);
    else
      hipLaunchKernelGGL((colgroup_kernel<T, VW, true, F, true>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, f, x, y, ladj_ps, dim, batch, c.G, accum, fin, ldx, ldy, row0);
  } else if (c.V == VW)
    hipLaunchKernelGGL((colgroup_kernel<T, VW, true, F>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, f, x, y, ladj_ps, dim, batch, c.G, accum, fin, ldx, ldy, row0);
  else
    hipLaunchKernelGGL((colgroup_kernel<T, 1, true, F>), dim3((unsigned)c.grid), dim3(256), smem, ctx->stream, f, x, y, ladj_ps, dim, batch, c.G, accum, fin, ldx, ldy, row0);
  }
  BJX_CHECK_LAUNCH(ctx);
  if (second) return bjx_launch_finalize(ctx, (int)c.grid, ladj_sum, sum_const, f.per_sample_dev ? 1 : 0, 0.0, flags);
  return BJX_OK;
}

}  // namespace bjx
