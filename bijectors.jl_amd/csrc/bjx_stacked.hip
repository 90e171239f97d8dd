// bjx_stacked.hip — SURVEY.md §8(f) f-4 (first part): `Stacked` with elementwise segments in ONE launch.
//   Stacked(bs, ranges)   src/bijectors/stacked.jl:27-252:  y = vcat(bs[i](x[ranges_in[i]])...),
//                         logabsdetjac = Σ_i sum(logabsdetjac(bs[i], x[ranges_in[i]]))      (:172-196, :236-244)
// This is what `bijector(d)` of a mixed-constraint model hands to link/invlink: every row range of the
// parameter vector has its own (chain of) elementwise bijector(s).  The reference slices, transforms
// and `vcat`s per segment (one allocation per segment per call); launching the chain kernel once per
// segment would touch a few rows of every column per launch (uncoalesced).  Here a 1-block prep
// launch expands the segment list into a per-OUTPUT-row table {source row, <= 4 op kinds, parameters}
// and one streaming pass on the column-group skeleton applies each row's ops (per-lane `switch`; lanes
// of a wave that hold different kinds serialise only over the kinds present).
#include <cstdlib>
#include <vector>

#include "bjx_stream.h"

namespace {
using namespace bjx;

// ---- canonical slot: every chain of elementwise bijectors is (a composition of at most two of)
//        y = clamp_post( a2 · N( a1 · clamp_pre(x) + b1 ) + b2 ),   N ∈ {id, exp, log, logit, logistic, leaky}
//      log|J| = ladj_N(a1·x+b1) + log|a1| + log|a2|
//   exp ∘ Shift(b) ∘ Scale(a)      -> N = exp, a1 = a, b1 = b                        (one slot instead of three ops)
//   Logit(lo, hi)                  -> N = logit, a1 = 1/(hi-lo), b1 = -lo/(hi-lo)    (logit.jl:15,24)
//   inverse(Logit(lo, hi))         -> N = logistic, a2 = hi-lo, b2 = lo              (logit.jl:19)
//   TruncatedBijector(lo, hi)      -> clamp_pre + logit / log(x-lo) / log(hi-x) / id (truncated.jl:15-31,51-67)
//   inverse(TruncatedBijector)     -> logistic / exp with a2, b2 and clamp_post      (truncated.jl:33-49,71-91)
// The per-row compile runs on the device (one thread per output row walks its op list), so per-row
// vector parameters fold exactly like scalars.  The streaming kernel then has a 6-way switch per slot
// instead of the 12-way op switch per op (which, unrolled over 16 elements x 3 ops, produced a 120 KB
// kernel that thrashed the instruction cache: 19 % of the HBM roofline).
enum { SK_END = 0, SK_ID = 1, SK_EXP = 2, SK_LOG = 3, SK_LOGIT = 4, SK_LOGISTIC = 5, SK_LEAKY = 6 };
constexpr int STACKED_SLOTS = 2;

template <class T> struct alignas(16) Slot {      // 12 words (Float32), 16-byte aligned so a slot is read with vector loads
  T clo, chi;      // pre-clamp  (-inf, +inf when absent)
  T a1, b1;
  T a2, b2;
  T plo, phi;      // post-clamp
  T alpha;         // LeakyReLU slope
  T c;             // log|a1| + log|a2| (+ log|hi-lo| terms)
  int32_t kind;
  int32_t src;     // slot 0 only: input row feeding this output row
};

struct SegDev {                      // device copy of one bjx_segment (pointers are device pointers)
  int64_t in_lo, out_lo, len;
  int32_t n_ops, pad;
  int32_t kind[BJX_MAX_SEG_OPS], plen[BJX_MAX_SEG_OPS];
  double s0[BJX_MAX_SEG_OPS], s1[BJX_MAX_SEG_OPS];
  const void* v0[BJX_MAX_SEG_OPS];
  const void* v1[BJX_MAX_SEG_OPS];
};

// A few segment descriptors as one kernel argument (<= 4 KiB), stored to the device list by a 1-wave launch.
struct SegPack {
  static constexpr int N = (3584 / (int)sizeof(SegDev)) < 1 ? 1 : (3584 / (int)sizeof(SegDev));
  SegDev s[N];
};
__global__ void stacked_seg_store_kernel(const SegPack pk, int cnt, SegDev* __restrict__ dst) {
  const int words = cnt * (int)(sizeof(SegDev) / 8);
  const int64_t* src = reinterpret_cast<const int64_t*>(pk.s);
  int64_t* d = reinterpret_cast<int64_t*>(dst);
  for (int i = threadIdx.x; i < words; i += blockDim.x) d[i] = src[i];
}

// Table layout: one entry of STACKED_SLOTS slots per OUTPUT row, padded to an ODD number of 16-byte units
// (7 for Float32, 11 for Float64) and indexed by the PERMUTED row rp = (row % V)·nvc + row / V, so that
// the rows one wave instruction touches (same element of consecutive packs) are adjacent: their 16-byte
// LDS reads then fall into different bank groups.  (Unpermuted, a 96-byte entry put 16 lanes on 2 bank
// groups: SQ_LDS_BANK_CONFLICT = 89 % of the LDS cycles and 23 % of the HBM roofline.)
template <class T> __host__ __device__ constexpr size_t stacked_row_bytes() {
  return ((STACKED_SLOTS * sizeof(Slot<T>) + 15) / 16) % 2 == 1 ? (STACKED_SLOTS * sizeof(Slot<T>) + 15) / 16 * 16
                                                                : (STACKED_SLOTS * sizeof(Slot<T>) + 15) / 16 * 16 + 16;
}
// (rows past the last whole pack — odd heights on element-aligned packs — keep their natural place behind the permuted block)
__host__ __device__ inline int64_t stacked_row_index(int64_t row, int V, int64_t nvc) { return (V > 1 && row < nvc * V) ? (row % V) * nvc + row / V : row; }

template <class T> __device__ __forceinline__ void slot_reset(Slot<T>& s) {
  s.clo = -Num<T>::inf; s.chi = Num<T>::inf; s.a1 = T(1); s.b1 = T(0); s.a2 = T(1); s.b2 = T(0);
  s.plo = -Num<T>::inf; s.phi = Num<T>::inf; s.alpha = T(1); s.c = T(0); s.kind = SK_ID; s.src = 0;
}

// flag[0] |= 1: some row is not its own source (gather); flag[0] |= 2: a chain needs more than two slots
template <class T>
__global__ __launch_bounds__(256) void stacked_table_kernel(const SegDev* segs, int n_segs, int64_t dim, int V, char* tab, int* flag) {
  const int64_t nvc = dim / V;
  int bits = 0;
  for (int sidx = 0; sidx < n_segs; ++sidx) {
    const SegDev& g = segs[sidx];
    if (g.in_lo != g.out_lo) bits |= 1;
    for (int64_t i = threadIdx.x; i < g.len; i += blockDim.x) {
      Slot<T> sl[STACKED_SLOTS];
      int cur = 0;
      bool has_n = false;        // the current slot already has its nonlinearity
      bool seen_affine = false;  // ... or a folded affine op (value-independent: the host counts slots the same way)
      bool post_clamped = false;
      bool overflow = false;
      slot_reset(sl[0]);
      for (int k = 0; k < g.n_ops; ++k) {
        const int kind = g.kind[k];
        const T a = g.plen[k] > 1 ? reinterpret_cast<const T*>(g.v0[k])[i] : (g.plen[k] == 1 && g.v0[k] ? reinterpret_cast<const T*>(g.v0[k])[0] : (T)g.s0[k]);
        const T b = (g.plen[k] > 1 && g.v1[k]) ? reinterpret_cast<const T*>(g.v1[k])[i] : (g.plen[k] == 1 && g.v1[k] ? reinterpret_cast<const T*>(g.v1[k])[0] : (T)g.s1[k]);
        const bool affine = kind == BJX_OP_SHIFT || kind == BJX_OP_SCALE || kind == BJX_OP_SCALE_INV || kind == BJX_OP_SIGNFLIP || kind == BJX_OP_IDENTITY;
        Slot<T>* s = &sl[cur];
        if (affine) {
          T m = T(1), t = T(0);
          if (kind == BJX_OP_SHIFT) t = a;                        // shift.jl:14
          else if (kind == BJX_OP_SCALE) m = a;                   // scale.jl:13
          else if (kind == BJX_OP_SCALE_INV) m = T(1) / a;        // scale.jl:15-16
          else if (kind == BJX_OP_SIGNFLIP) m = T(-1);            // ordered.jl:3
          if (has_n && post_clamped) {
            if (cur + 1 < STACKED_SLOTS) { ++cur; s = &sl[cur]; slot_reset(*s); has_n = false; post_clamped = false; seen_affine = false; }
            else { overflow = true; continue; }
          }
          seen_affine = true;
          if (!has_n) { s->a1 *= m; s->b1 = s->b1 * m + t; } else { s->a2 *= m; s->b2 = s->b2 * m + t; }
          s->c += d_log(d_abs(m));                                // scale.jl:26-32 (0 for Shift)
          continue;
        }
        // a nonlinearity: needs a slot without one (and, for the clamping ones, a fresh slot)
        const bool clamps = kind == BJX_OP_TRUNCATED;
        if (has_n || (clamps && seen_affine)) {
          if (cur + 1 < STACKED_SLOTS) { ++cur; s = &sl[cur]; slot_reset(*s); post_clamped = false; seen_affine = false; } else { overflow = true; continue; }
        }
        has_n = true;
        if (kind == BJX_OP_TRUNCATED_INV) post_clamped = true;
        const T lo = a, hi = b;
        const bool lb = d_isfinite(lo), ub = d_isfinite(hi);
        switch (kind) {
          case BJX_OP_EXP: s->kind = SK_EXP; break;
          case BJX_OP_LOG: s->kind = SK_LOG; break;
          // SK_LOGIT works on u = x' - lo in [0, w] (w kept in `alpha`): N(u) = log(u/(w-u)) is exact at both bounds
          // (u = 0 and w - u = 0 there; (x'-lo)/w can round to 1 ± ulp), log|N'| = log w - log(u(w-u))
          case BJX_OP_LOGIT: { const T w = hi - lo; s->b1 = s->b1 - lo; s->alpha = w; s->c += d_log(w); s->kind = SK_LOGIT; } break;
          case BJX_OP_LOGIT_INV: { const T w = hi - lo; s->a2 = w; s->b2 = lo; s->c += d_log(w); s->kind = SK_LOGISTIC; } break;
          case BJX_OP_LEAKY_RELU: s->kind = SK_LEAKY; s->alpha = a; break;
          case BJX_OP_TRUNCATED:
            s->clo = lo; s->chi = hi;
            if (lb && ub) { const T w = hi - lo; s->a1 = T(1); s->b1 = -lo; s->alpha = w; s->c += d_log(w); s->kind = SK_LOGIT; }
            else if (lb) { s->b1 = -lo; s->kind = SK_LOG; }
            else if (ub) { s->a1 = T(-1); s->b1 = hi; s->kind = SK_LOG; }
            else s->kind = SK_ID;
            break;
          case BJX_OP_TRUNCATED_INV:
            s->plo = lo; s->phi = hi;
            if (lb && ub) { const T w = hi - lo; s->a2 = w; s->b2 = lo; s->c += d_log(w); s->kind = SK_LOGISTIC; }
            else if (lb) { s->b2 = lo; s->kind = SK_EXP; }
            else if (ub) { s->a2 = T(-1); s->b2 = hi; s->kind = SK_EXP; }
            else s->kind = SK_ID;
            break;
          default: break;
        }
      }
      if (overflow) bits |= 2;
      sl[0].src = (int32_t)(g.in_lo + i);
      for (int q = cur + 1; q < STACKED_SLOTS; ++q) { slot_reset(sl[q]); sl[q].kind = SK_END; }
      {
        const int64_t r = g.out_lo + i;
        Slot<T>* dst = reinterpret_cast<Slot<T>*>(tab + stacked_row_index(r, V, nvc) * stacked_row_bytes<T>());
        for (int q = 0; q < STACKED_SLOTS; ++q) dst[q] = sl[q];
      }
    }
  }
  if (bits) atomicOr(flag, bits);
}

// one slot on one element; returns the log-det contribution
template <class T> __device__ __forceinline__ T slot_eval(const Slot<T>& s, T& x) {
  using F = Fast<T>;
  const T xc = d_med3(x, s.clo, s.chi);
  const T u = s.a1 * xc + s.b1;
  T v = u, l = s.c;
  switch (s.kind) {
    case SK_EXP: v = F::exp(u); l += u; break;                                                        // exp_log.jl:5-6
    case SK_LOG: v = F::log(u); l -= v; break;                                                        // exp_log.jl:8-9
    case SK_LOGIT: { const T q = s.alpha - u; l -= F::log(u * q); v = F::log(u * F::rcp(q)); } break;   // logit.jl:15,24
    case SK_LOGISTIC: { const T au = d_abs(u); v = f_logistic(u); l += -au - T(2) * f_log1pexp(-au); } break;   // logit.jl:19, truncated.jl:71-82
    case SK_LEAKY: { const T J = u < T(0) ? s.alpha : T(1); v = J * u; l += Fast<T>::log(d_abs(J)); } break; // leaky_relu.jl:25-29
    default: break;
  }
  x = d_med3(s.a2 * v + s.b2, s.plo, s.phi);
  return l;
}

// the same slot on the U elements a lane holds at one row (U columns in flight): one fetch of the slot and ONE
// pass through the per-lane switch for U elements — lanes of a wave hold different kinds, so every kind present in
// the wave is executed serially; amortising that over the columns in flight is what this buys.
template <class T, int U> __device__ __forceinline__ void slot_eval_multi(const Slot<T>& s, T (&x)[U], T (&l)[U]) {
  using F = Fast<T>;
  T u[U], v[U];
#pragma unroll
  for (int i = 0; i < U; ++i) { u[i] = s.a1 * d_med3(x[i], s.clo, s.chi) + s.b1; v[i] = u[i]; l[i] += s.c; }
  switch (s.kind) {
    case SK_EXP:
#pragma unroll
      for (int i = 0; i < U; ++i) { v[i] = F::exp(u[i]); l[i] += u[i]; }
      break;
    case SK_LOG:
#pragma unroll
      for (int i = 0; i < U; ++i) { v[i] = F::log(u[i]); l[i] -= v[i]; }
      break;
    case SK_LOGIT:
#pragma unroll
      for (int i = 0; i < U; ++i) { const T q = s.alpha - u[i]; l[i] -= F::log(u[i] * q); v[i] = F::log(u[i] * F::rcp(q)); }
      break;
    case SK_LOGISTIC:
#pragma unroll
      for (int i = 0; i < U; ++i) { const T au = d_abs(u[i]); v[i] = f_logistic(u[i]); l[i] += -au - T(2) * f_log1pexp(-au); }
      break;
    case SK_LEAKY:
#pragma unroll
      for (int i = 0; i < U; ++i) { const T J = u[i] < T(0) ? s.alpha : T(1); v[i] = J * u[i]; l[i] += Fast<T>::log(d_abs(J)); }
      break;
    default: break;
  }
#pragma unroll
  for (int i = 0; i < U; ++i) x[i] = d_med3(s.a2 * v[i] + s.b2, s.plo, s.phi);
}

template <class T, bool GATHER, bool IN_LDS> struct StackedF {
  static constexpr bool kLoadInput = !GATHER;
  static constexpr bool kMulti = !GATHER;          // apply_multi: all columns in flight of a lane group at once
  static constexpr bool kMasked = !GATHER;         // the masked forms take any first row (every element looks up its own table place)
  template <int V, int U> __device__ void apply_multi_masked(const char* smem, Pack<T, V> (&p)[U], int64_t row, T (&l)[U], uint32_t mask) const {
    const char* t = IN_LDS ? smem : tab;
    const int64_t nvc = dim / V;
#pragma unroll
    for (int i = 0; i < U; ++i) l[i] = T(0);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const Slot<T>* e = reinterpret_cast<const Slot<T>*>(t + stacked_row_index(row + j, V, nvc) * stacked_row_bytes<T>());
      T x[U], lj[U];
#pragma unroll
      for (int i = 0; i < U; ++i) { x[i] = p[i].v[j]; lj[i] = T(0); }
#pragma unroll 1
      for (int q = 0; q < (two_slots ? STACKED_SLOTS : 1); ++q) {
        const Slot<T> sq = e[q];
        if (sq.kind == SK_END) break;
        slot_eval_multi<T, U>(sq, x, lj);
      }
      const bool on = (mask >> j) & 1u;
#pragma unroll
      for (int i = 0; i < U; ++i) { p[i].v[j] = x[i]; l[i] += on ? lj[i] : T(0); }
    }
  }
  template <int V> __device__ T apply_masked(const char* smem, Pack<T, V>& p, const T*, int64_t row, int64_t, uint32_t mask) const {
    Pack<T, V> pp[1] = {p};
    T l1[1];
    apply_multi_masked<V, 1>(smem, pp, row, l1, mask);
    p = pp[0];
    return l1[0];
  }
  template <int V, int U> __device__ void apply_multi(const char* smem, Pack<T, V> (&p)[U], int64_t row, T (&l)[U]) const {
    const char* t = IN_LDS ? smem : tab;
    const int64_t nvc = dim / V;
#pragma unroll
    for (int i = 0; i < U; ++i) l[i] = T(0);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const Slot<T>* e = reinterpret_cast<const Slot<T>*>(t + (V > 1 ? j * nvc + row / V : row) * stacked_row_bytes<T>());
      T x[U];
#pragma unroll
      for (int i = 0; i < U; ++i) x[i] = p[i].v[j];
#pragma unroll 1
      for (int q = 0; q < (two_slots ? STACKED_SLOTS : 1); ++q) {
        const Slot<T> sq = e[q];
        if (sq.kind == SK_END) break;
        slot_eval_multi<T, U>(sq, x, l);
      }
#pragma unroll
      for (int i = 0; i < U; ++i) p[i].v[j] = x[i];
    }
  }
  const char* tab;
  int64_t dim;
  int two_slots;
  double per_sample_const;
  const double* per_sample_dev;
  int walk_smem_offset = 0;   // colwalk_kernel: where the column tile starts behind the functor's LDS tables (set by launch_colgroup)
  __device__ void stage(char* smem) const {
    if (IN_LDS) {
      const int n16 = (int)(dim * stacked_row_bytes<T>() / 16);
      const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(tab);
      bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(smem);
      for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
      __syncthreads();
    }
  }
  template <int V> __device__ T apply(const char* smem, Pack<T, V>& p, const T* xcol, int64_t row, int64_t) const {
    const char* t = IN_LDS ? smem : tab;          // two instantiations: never a generic (flat) pointer
    const int64_t nvc = dim / V;
    T l = T(0);
    // NOT unrolled: slot_eval is ~200 instructions; 4 elements x 2 slots x the skeleton's own unrolling made
    // a 15k-instruction kernel
#pragma unroll 1
    for (int j = 0; j < V; ++j) {
      const Slot<T>* e = reinterpret_cast<const Slot<T>*>(t + (V > 1 ? j * nvc + row / V : row) * stacked_row_bytes<T>());
      T x = p.v[0];
      if (V > 1) x = j == 1 ? p.v[1 % V] : (j == 2 ? p.v[2 % V] : (j == 3 ? p.v[3 % V] : x));
      if (GATHER) x = xcol[e[0].src];
#pragma unroll 1
      for (int q = 0; q < (two_slots ? STACKED_SLOTS : 1); ++q) {
        const Slot<T> sq = e[q];                  // 16-byte aligned: whole-slot vector reads
        if (sq.kind == SK_END) break;
        l += slot_eval<T>(sq, x);
      }
      if (V == 1) p.v[0] = x;
      else {
        p.v[0] = j == 0 ? x : p.v[0];
        p.v[1 % V] = j == 1 ? x : p.v[1 % V];
        p.v[2 % V] = j == 2 ? x : p.v[2 % V];
        p.v[3 % V] = j == 3 ? x : p.v[3 % V];
      }
    }
    return l;
  }
};

struct StackedPlan { char* tab; size_t tab_bytes; bool gather; int two; int V; };

// lanes per column of the slab kernels: 64 or 32, whichever leaves fewer idle lanes in the last slab (84 units — 333 rows — are
// 64 + 20 on 64-lane slabs, a third of the lanes idle, and 32 + 32 + 20 on 32-lane slabs)
inline int stacked_slab_lanes(int64_t units) {
  const int64_t w64 = (units + 63) / 64 * 64 - units, w32 = (units + 31) / 32 * 32 - units;
  return w32 < w64 ? 32 : 64;
}

// validates the segment list, uploads it and launches the table build; `x`/`y` only decide the pack width
// `ldx` = rows of the input matrix (== dim unless the call is bjx_stacked_ld: the segments then read rows of a taller /
// shorter matrix and every row is gathered), `ldy` = rows of the output matrix (>= dim)
template <class T>
int stacked_prepare(bjx_ctx* ctx, const bjx_segment* segs, int n_segs, const void* x, const void* y, int64_t dim, int64_t batch, bool inplace_check,
                    bool packs_ok, StackedPlan* plan, int64_t ldx = 0, int64_t ldy = 0, bool allow_unal = false, int unal_from = 0) {
  if (ldx == 0) ldx = dim;
  if (ldy == 0) ldy = dim;
  // validate on the host: every output row and every input row exactly once (stacked.jl:156-165 checks the lengths)
  int64_t total = 0;
  int max_ops = 0;
  bool gather = false;
  for (int s = 0; s < n_segs; ++s) {
    const bjx_segment& g = segs[s];
    BJX_REQUIRE(ctx, g.len >= 0 && g.in_lo >= 0 && g.out_lo >= 0 && g.in_lo + g.len <= ldx && g.out_lo + g.len <= dim, BJX_ERR_SHAPE,
                "bjx_stacked: segment %d [%lld, +%lld) is outside the %lld rows", s, (long long)g.in_lo, (long long)g.len, (long long)dim);
    BJX_REQUIRE(ctx, g.n_ops >= 0 && g.n_ops <= BJX_MAX_SEG_OPS, BJX_ERR_ARG, "bjx_stacked: segment %d has %d ops (max %d)", s, g.n_ops, BJX_MAX_SEG_OPS);
    for (int k = 0; k < g.n_ops; ++k)
      BJX_REQUIRE(ctx, g.ops[k].param_len == 0 || g.ops[k].param_len == 1 || g.ops[k].param_len == g.len, BJX_ERR_SHAPE,
                  "bjx_stacked: segment %d op %d: parameter of length %d for %lld rows", s, k, g.ops[k].param_len, (long long)g.len);
    total += g.len;
    if (g.n_ops > max_ops) max_ops = g.n_ops;
    if (g.in_lo != g.out_lo) gather = true;
  }
  // (a window of taller matrices is NOT a gather: rows keep their place inside the window, only the column stride differs)
  BJX_REQUIRE(ctx, total == dim, BJX_ERR_SHAPE, "input length mismatch (%lld != %lld)", (long long)total, (long long)dim);   // stacked.jl:157
  const size_t seg_bytes = ((size_t)n_segs * sizeof(SegDev) + 63) / 64 * 64;
  const size_t tab_bytes = (size_t)dim * stacked_row_bytes<T>();
  BJX_REQUIRE(ctx, 64 + seg_bytes + tab_bytes <= BJX_SCRATCH_BYTES, BJX_ERR_UNSUPPORTED, "bjx_stacked: %d segments / %lld rows exceed the context scratch", n_segs, (long long)dim);
  BJX_REQUIRE(ctx, !inplace_check || !gather || x != y, BJX_ERR_ARG, "bjx_stacked: in-place is only supported when every segment keeps its rows (ranges_in == ranges_out)");
  // value-independent slot count per segment (same rules as stacked_table_kernel)
  for (int sg = 0; sg < n_segs; ++sg) {
    int cur = 0;
    bool has_n = false, seen_affine = false, post_clamped = false;
    for (int k = 0; k < segs[sg].n_ops; ++k) {
      const int kind = segs[sg].ops[k].kind;
      const bool affine = kind == BJX_OP_SHIFT || kind == BJX_OP_SCALE || kind == BJX_OP_SCALE_INV || kind == BJX_OP_SIGNFLIP || kind == BJX_OP_IDENTITY;
      if (affine) {
        if (has_n && post_clamped) { ++cur; has_n = false; post_clamped = false; seen_affine = false; }
        seen_affine = true;
      } else {
        BJX_REQUIRE(ctx, kind >= BJX_OP_EXP && kind <= BJX_OP_IDENTITY, BJX_ERR_ARG, "bjx_stacked: segment %d op %d: bad kind %d", sg, k, kind);
        if (has_n || (kind == BJX_OP_TRUNCATED && seen_affine)) { ++cur; post_clamped = false; seen_affine = false; }
        has_n = true;
        if (kind == BJX_OP_TRUNCATED_INV) post_clamped = true;
      }
    }
    BJX_REQUIRE(ctx, cur < STACKED_SLOTS, BJX_ERR_UNSUPPORTED, "bjx_stacked: the chain of segment %d needs more than %d nonlinear stages", sg, STACKED_SLOTS);
  }
  auto fill = [&](int sg) {
    SegDev d{};
    d.in_lo = segs[sg].in_lo; d.out_lo = segs[sg].out_lo; d.len = segs[sg].len; d.n_ops = segs[sg].n_ops;
    for (int k = 0; k < segs[sg].n_ops; ++k) {
      d.kind[k] = segs[sg].ops[k].kind; d.plen[k] = segs[sg].ops[k].param_len;
      d.s0[k] = segs[sg].ops[k].p0; d.s1[k] = segs[sg].ops[k].p1; d.v0[k] = segs[sg].ops[k].v0; d.v1[k] = segs[sg].ops[k].v1;
    }
    return d;
  };
  char* sc = static_cast<char*>(ctx->scratch);
  int* flag = reinterpret_cast<int*>(sc);
  SegDev* dseg = reinterpret_cast<SegDev*>(sc + 64);
  char* tab = sc + 64 + seg_bytes;
  BJX_HIP(ctx, hipMemsetAsync(flag, 0, 64, ctx->stream));
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(ctx->stream, &cap);
  if (n_segs <= 8 * SegPack::N || cap != hipStreamCaptureStatusNone) {
    // segment list -> device as KERNEL ARGUMENTS, SegPack::N segments per tiny launch: no pinned staging buffer, no
    // event wait between calls, and the call can be captured into a hipGraph (the descriptors live in the graph node)
    for (int s0 = 0; s0 < n_segs; s0 += SegPack::N) {
      SegPack pk;
      const int cnt = n_segs - s0 < SegPack::N ? n_segs - s0 : SegPack::N;
      for (int i = 0; i < cnt; ++i) pk.s[i] = fill(s0 + i);
      hipLaunchKernelGGL(stacked_seg_store_kernel, dim3(1), dim3(64), 0, ctx->stream, pk, cnt, dseg + s0);
      BJX_CHECK_LAUNCH(ctx);
    }
  } else {
    // long segment lists: through the context's pinned staging buffer (asynchronous; the event guards its reuse)
    BJX_REQUIRE(ctx, seg_bytes <= BJX_HOST_STAGE_BYTES, BJX_ERR_UNSUPPORTED, "bjx_stacked: too many segments (%d)", n_segs);
    BJX_REQUIRE(ctx, !ctx->capturing, BJX_ERR_UNSUPPORTED, "bjx_stacked: %d segments go through the pinned staging buffer, which cannot be recorded into a graph", n_segs);
    if (!ctx->host_stage) {
      BJX_HIP(ctx, hipHostMalloc(&ctx->host_stage, BJX_HOST_STAGE_BYTES, hipHostMallocDefault));
      BJX_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev, hipEventDisableTiming));
    } else {
      BJX_HIP(ctx, hipEventSynchronize(ctx->stage_ev));      // the previous call's copy has left the buffer
    }
    SegDev* hseg = static_cast<SegDev*>(ctx->host_stage);
    for (int sg = 0; sg < n_segs; ++sg) hseg[sg] = fill(sg);
    BJX_HIP(ctx, hipMemcpyAsync(dseg, hseg, (size_t)n_segs * sizeof(SegDev), hipMemcpyHostToDevice, ctx->stream));
    BJX_HIP(ctx, hipEventRecord(ctx->stage_ev, ctx->stream));
  }
  // the main kernel's pack width decides the row permutation of the table (same rule as col_launch_cfg)
  ColLaunch cl = col_launch_cfg<T>(ctx, x, y, dim, batch, ldx, ldy, allow_unal, unal_from);
  if (!packs_ok) cl.V = 1;                               // a third buffer of the caller is not 16-byte aligned
  hipLaunchKernelGGL(stacked_table_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, dseg, n_segs, dim, cl.V, tab, flag);
  BJX_CHECK_LAUNCH(ctx);
  plan->tab = tab; plan->tab_bytes = tab_bytes; plan->gather = gather; plan->two = max_ops > 1 ? 1 : 0; plan->V = cl.V;
  return BJX_OK;
}

template <class T>
int stacked_mixed_impl(bjx_ctx* ctx, const bjx_segment* segs, int n_segs, const bjx_block* blocks, int n_blocks, const T* x, int64_t rows_in, T* y,
                       int64_t rows_out, T* ladj_ps, double* ladj_sum, int64_t batch, uint32_t flags);

// ------------------------------------------------------------------ tall columns in ONE launch (round 4)
// Columns of more than 64 packs with every segment on its own rows: the blocks of one grid are (block of columns) x (ROW SLAB of
// G units) — the slab index runs fastest — each staging its slab's part of the row-permuted table in LDS ([V][G] entries + the
// tail unit's rows), four columns in flight per lane sharing a row's slots (slot_eval_multi).  A block leaves one partial log-det
// per column and slab in a [slab][column] scratch; stacked_slab_combine_kernel adds the slabs of a column in slab order (fixed:
// deterministic), writes the per-sample log-det and the block partials of the sum.  The host loop this replaces launched a table
// build and a kernel per slab and accumulated in launch order: 16 columns x 5 000 rows took 358 us a call.
template <class T, int V, bool UNAL>
__global__ __launch_bounds__(256) void stacked_fwd_slab_kernel(const char* __restrict__ tab_g, int two_slots, const T* __restrict__ x, T* __restrict__ y,
                                                               T* __restrict__ lpart, int64_t dim, int64_t batch, int G, int nslab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int U = COL_UC;
  constexpr int RB16 = (int)(stacked_row_bytes<T>() / 16);
  const int64_t nvc = dim / V;
  const int tail = UNAL ? (int)(dim - nvc * V) : 0;
  const int64_t nun = nvc + (tail ? 1 : 0);
  const int slab = (int)(blockIdx.x % (unsigned)nslab);
  const int64_t cblk = blockIdx.x / (unsigned)nslab;
  const int64_t v0 = (int64_t)slab * G;
  {
    const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(tab_g);
    bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(smem);
    for (int i = threadIdx.x; i < V * G * RB16; i += blockDim.x) {
      const int q = i % RB16, en = i / RB16, vv = en & (G - 1), j = en / G;
      if (v0 + vv < nvc) dst[i] = src[((int64_t)j * nvc + v0 + vv) * RB16 + q];
    }
    if (UNAL && v0 <= nvc && nvc < v0 + G) {
      for (int i = threadIdx.x; i < V * RB16; i += blockDim.x) {
        const int q = i % RB16, j = i / RB16;
        dst[V * G * RB16 + i] = src[stacked_row_index(dim - V + j, V, nvc) * RB16 + q];
      }
    }
    __syncthreads();
  }
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = blockDim.x / G;
  const int64_t col0 = cblk * cols_per_block * U + threadIdx.x / G;
  const int64_t v = v0 + gl;
  const bool lane_ok = v < nun;
  const bool is_tail = UNAL && v == nvc;
  const int64_t prow = is_tail ? dim - V : v * V;
  constexpr uint32_t full = (1u << V) - 1u;
  const uint32_t mask = is_tail ? ((full << (V - tail)) & full) : full;     // the tail unit owns the LAST `tail` rows of its pack
  Pack<T, V> p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t col = col0 + (int64_t)u * cols_per_block;
    if (lane_ok && col < batch) p[u] = load_pack<T, V, true>(x + col * dim + prow);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) p[u].v[j] = T(1);          // harmless input for every slot kind; nothing of it is kept
    }
  }
  T l[U];
#pragma unroll
  for (int u = 0; u < U; ++u) l[u] = T(0);
  if (lane_ok) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const Slot<T>* e = reinterpret_cast<const Slot<T>*>(smem + (size_t)(is_tail ? V * G + j : j * G + gl) * stacked_row_bytes<T>());
      T xv[U], lj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { xv[u] = p[u].v[j]; lj[u] = T(0); }
#pragma unroll 1
      for (int q = 0; q < (two_slots ? STACKED_SLOTS : 1); ++q) {
        const Slot<T> sq = e[q];
        if (sq.kind == SK_END) break;
        slot_eval_multi<T, U>(sq, xv, lj);
      }
      const bool on = (mask >> j) & 1u;
#pragma unroll
      for (int u = 0; u < U; ++u) { p[u].v[j] = xv[u]; l[u] += on ? lj[u] : T(0); }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t col = col0 + (int64_t)u * cols_per_block;
    if (y && lane_ok && col < batch) {
      T* yp = y + col * dim + prow;
      if (!is_tail) store_pack<T, V, false>(yp, p[u]);
      else store_pack_run<T, V>(yp, p[u], V - tail, tail);
    }
    const T ls = group_sum_rt(lane_ok ? l[u] : T(0), G);
    if (gl == 0 && col < batch) lpart[(int64_t)slab * batch + col] = ls;
  }
}

template <class T>
__global__ __launch_bounds__(256) void stacked_slab_combine_kernel(const T* __restrict__ lpart, int nslab, int64_t batch, T* __restrict__ ladj_ps, int accumulate,
                                                                   double* __restrict__ partials) {
  __shared__ double red[4];
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  if (col < batch) {
    T l = T(0);
    for (int sidx = 0; sidx < nslab; ++sidx) l += lpart[(int64_t)sidx * batch + col];      // slab order: fixed
    if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + l : l;
    acc = (double)l;
  }
  if (partials) block_publish_partial(acc, red, partials);
}

template <class T>
int stacked_impl(bjx_ctx* ctx, const bjx_segment* segs, int n_segs, const T* x, T* y, T* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch,
                 uint32_t flags, int64_t ldx = 0, int64_t ldy = 0) {
  if (dim * batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  // columns that are not whole 16-byte packs: the row-owner kernel below would read them 4 bytes at a time; the column walker moves
  // 64 columns as one contiguous run of packs whatever their height (cf. bjx_chain)
  static const int use_walker = getenv("BJX_STACKED_WALKER") ? atoi(getenv("BJX_STACKED_WALKER")) : 1;
  const bool unal = col_launch_cfg<T>(ctx, x, y, dim, batch, ldx, ldy, true).unal != 0;     // odd heights from 48 rows: the group kernel on element-aligned packs
  if (use_walker && !unal && ldx == 0 && ldy == 0 && (const void*)x != (const void*)y && (dim % Vec16<T>::N != 0 || dim * sizeof(T) <= 16) && (flags & ~(uint32_t)BJX_ACCUMULATE) == 0) {   // one pack per column: stacked_tiny_kernel (69-71 % against 63-68 %; two packs: 65 against 72-74 %)
    const int rc_w = stacked_mixed_impl<T>(ctx, segs, n_segs, nullptr, 0, x, dim, y, dim, ladj_ps, ladj_sum, batch, flags);
    if (rc_w != BJX_ERR_UNSUPPORTED) return rc_w;                  // permuted ranges / taller than the tile: below
  }
  // Tall columns: ROW SLABS of 64 packs (256 rows Float32, 128 Float64).  Up to 64 packs a column is one pack per lane, and the
  // group kernel then keeps four columns in flight per lane and evaluates a row's slots once for the four (apply_multi); past that it
  // walks a column pack by pack through the one-element form (300 rows: 24 % of the HBM peak against 62 % at 252), its blocks stage
  // more table than they move data (2 x 48 bytes a row) and past 512 rows the table no longer fits the LDS at all (1000 rows: every
  // element reads its slots from L2, 22 %).  A slab is a window of rows of the same arrays (column pitch = dim) with the segments —
  // and their per-row parameters — clipped to it: one launch pair per slab, the log-dets of the slabs accumulated in launch order
  // (BJX_ACCUMULATE from the second slab on: deterministic), as for the spline tables.  BJX_STACKED_SLAB = rows per slab (0: off).
  static const int slab_env = getenv("BJX_STACKED_SLAB") ? atoi(getenv("BJX_STACKED_SLAB")) : -1;
  const int slab = slab_env >= 0 ? slab_env : 64 * Vec16<T>::N;
  if (slab >= 16 && ldx == 0 && ldy == 0 && dim > slab && n_segs > 0) {
    bool keep = true;
    int64_t total = 0;
    for (int s = 0; s < n_segs && keep; ++s) {
      const bjx_segment& g = segs[s];
      keep = g.in_lo == g.out_lo && g.len >= 0 && g.in_lo >= 0 && g.in_lo + g.len <= dim && g.n_ops >= 0 && g.n_ops <= BJX_MAX_SEG_OPS;
      for (int k = 0; keep && k < g.n_ops; ++k) keep = g.ops[k].param_len == 0 || g.ops[k].param_len == 1 || g.ops[k].param_len == g.len;
      total += g.len;
    }
    if (keep && total == dim) {                        // (anything else: the one-launch path below reports it)
      {
        // one launch for all slabs (stacked_fwd_slab_kernel) when the column is made of 16-byte packs, aligned or element-aligned
        constexpr int VWo = Vec16<T>::N;
        StackedPlan po;
        po.V = 0; po.gather = true;
        if ((double)batch * (double)dim * sizeof(T) <= 256.0 * 1024 * 1024) {
          int rc = stacked_prepare<T>(ctx, segs, n_segs, x, y ? (const void*)y : (const void*)x, dim, batch, true, true, &po, 0, 0, true);
          if (rc) return rc;
        }
        // ... for inputs up to 256 MiB, where the launches of the loop below are what a call costs (16 columns x 5 000 rows: 358 -> 81 us,
        // x 1 001 rows: 88 -> 45 us); on the large batches of the throughput tables the loop's whole-table-in-LDS kernels are 5-15 %
        // ahead (2^22 columns x 333 / 1 001 rows: 47 / 49 % of the HBM peak against 40 / 45 %) and keep the job
        const bool small_job = (double)batch * (double)dim * sizeof(T) <= 256.0 * 1024 * 1024;
        if (small_job && po.V == VWo && !po.gather && VWo > 1) {
          const bool whole = dim % VWo == 0;
          const int64_t units = dim / VWo + (whole ? 0 : 1);
          const int Go = stacked_slab_lanes(units);
          const int64_t nslab = (units + Go - 1) / Go;
          const int64_t cpb = (int64_t)(256 / Go) * COL_UC;
          const int64_t gridc = (batch + cpb - 1) / cpb;
          const int64_t gridk = (batch + 255) / 256;
          if (gridc * nslab < (int64_t)1 << 31) {
            { int rc = bjx_ensure_big_ws(ctx, (size_t)nslab * batch * sizeof(T)); if (rc) return rc; }
            if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)gridk); if (rc) return rc; }
            T* lpart = static_cast<T*>(ctx->big_ws);
            const size_t smem_o = (size_t)VWo * (Go + 1) * stacked_row_bytes<T>();
            {
              BjxProf prof_(ctx);
              if (whole) hipLaunchKernelGGL((stacked_fwd_slab_kernel<T, VWo, false>), dim3((unsigned)(gridc * nslab)), dim3(256), smem_o, ctx->stream, po.tab, po.two, x, y, lpart, dim, batch, Go, (int)nslab);
              else hipLaunchKernelGGL((stacked_fwd_slab_kernel<T, VWo, true>), dim3((unsigned)(gridc * nslab)), dim3(256), smem_o, ctx->stream, po.tab, po.two, x, y, lpart, dim, batch, Go, (int)nslab);
            }
            BJX_CHECK_LAUNCH(ctx);
            {
              BjxProf prof_(ctx);
              hipLaunchKernelGGL(stacked_slab_combine_kernel<T>, dim3((unsigned)gridk), dim3(256), 0, ctx->stream, (const T*)lpart, (int)nslab, batch, ladj_ps, (flags & BJX_ACCUMULATE) ? 1 : 0,
                                 ladj_sum ? ctx->partials : (double*)nullptr);
            }
            BJX_CHECK_LAUNCH(ctx);
            if (ladj_sum) return bjx_launch_finalize(ctx, (int)gridk, ladj_sum, 0.0, 0, 0.0, flags);
            return BJX_OK;
          }
        }
      }
      std::vector<bjx_segment> clip;
      for (int64_t r0 = 0, rs = 0; r0 < dim; r0 += rs) {
        rs = dim - r0 < slab ? dim - r0 : slab;
        clip.clear();
        for (int s = 0; s < n_segs; ++s) {
          const bjx_segment& g = segs[s];
          const int64_t lo = g.in_lo > r0 ? g.in_lo : r0, hi = g.in_lo + g.len < r0 + rs ? g.in_lo + g.len : r0 + rs;
          if (hi <= lo) continue;
          bjx_segment c = g;
          c.in_lo = c.out_lo = lo - r0;
          c.len = hi - lo;
          for (int k = 0; k < g.n_ops; ++k) {
            if (g.ops[k].param_len > 1) {              // one value per row of the segment: the window's part of it
              const size_t off = (size_t)(lo - g.in_lo) * sizeof(T);
              if (c.ops[k].v0) c.ops[k].v0 = static_cast<const char*>(g.ops[k].v0) + off;
              if (c.ops[k].v1) c.ops[k].v1 = static_cast<const char*>(g.ops[k].v1) + off;
              c.ops[k].param_len = (int32_t)(hi - lo);
            }
          }
          clip.push_back(c);
        }
        const uint32_t fl = r0 == 0 ? flags : (flags | BJX_ACCUMULATE);
        const int rc = stacked_impl<T>(ctx, clip.data(), (int)clip.size(), x + r0, y + r0, ladj_ps, ladj_sum, rs, batch, fl, dim, dim);
        if (rc) return rc;
      }
      return BJX_OK;
    }
  }
  StackedPlan pl;
  { int rc = stacked_prepare<T>(ctx, segs, n_segs, x, y, dim, batch, true, true, &pl, ldx, ldy, true); if (rc) return rc; }
  char* tab = pl.tab;
  const int two = pl.two;
  const bool lds = pl.tab_bytes <= 48 * 1024;
  const size_t fsm = lds ? pl.tab_bytes : 0;
#define STK_LAUNCH(G_, L_) do { StackedF<T, G_, L_> f{tab, dim, two, 0.0, nullptr}; return launch_colgroup<T>(ctx, f, fsm, x, y, ladj_ps, ladj_sum, dim, batch, flags, 0.0, ldx, ldy); } while (0)
  if (pl.gather) { if (lds) STK_LAUNCH(true, true); else STK_LAUNCH(true, false); }
  if (lds) STK_LAUNCH(false, true);
  STK_LAUNCH(false, false);
#undef STK_LAUNCH
}

// ------------------------------------------------------------------ Stacked with structured blocks, ONE launch
// stacked.jl:142-166 for a mixed-constraint model (what `bijector(d)` builds): elementwise chains on some row ranges, a
// Simplex or Ordered bijector on others.  Elementwise rows and structured blocks have nothing in common as a per-lane
// program when a lane owns a ROW (the streaming skeleton above), so here a lane owns a COLUMN: one wave stages 64
// consecutive columns (one contiguous run of the input) through a [64][P] LDS tile with an odd pitch and every lane walks
// its column top to bottom.  All lanes are at the same row at the same time, so the row's slots are wave-uniform (scalar
// loads, scalar branch on the kind — none of the per-lane kind divergence of the row-owner kernel), a structured block is
// the walker of bjx_seq.hip on a row window of the column, and the log-det of a column is one lane's running sum (no
// cross-lane reduction).  The walk is IN PLACE: the input sits `shift` rows down the column, shift = the largest amount by
// which an output range starts (or the output ends) above its input range, so a write never overtakes an unread input
// (inverse Simplex blocks lengthen the column, forward ones shorten it).
#include "bjx_seqops.h"

enum { MB_SIMPLEX = 1, MB_SIMPLEX_INV = 2, MB_ORDERED = 3, MB_ORDERED_INV = 4 };
struct MixBlock { int kind, in_lo, out_lo, len_in, len_out, pad; };
struct MixBlocks { static constexpr int N = 64; MixBlock b[N]; };

template <class T, class Op>
__device__ __forceinline__ T mixed_run_block(Op op, const T* pin, T* pout, int len_in, int len_out, const T* logn) {
  // logn[m] = log(m); the walkers want log(K-1-i) (simplex.jl:35,41): logn[K-1-i]
  op.init();
  const int rows = len_in > len_out ? len_in : len_out;
  const int K1 = (int)(Op::USES_LOGK ? (len_in > len_out ? len_in : len_out) - 1 : 0);     // K - 1
  const T lk0 = Op::USES_LOGK ? logn[K1] : T(0);
  {
    const T v = pin[0];                                   // (len_in >= 1 always)
    const T o = op.first(v, &lk0);
    if (len_out >= 1) pout[0] = o;
  }
  const int mid_end = (Op::HAS_LAST && rows > 1) ? rows - 1 : rows;
  for (int i = 1; i < mid_end; ++i) pout[i] = op.mid(i, pin[i], Op::USES_LOGK ? logn[K1 - i] : T(0));
  if (Op::HAS_LAST && rows > 1) {
    const T v = (rows - 1 < len_in) ? pin[rows - 1] : T(0);
    const T o = op.last(v);
    if (rows - 1 < len_out) pout[rows - 1] = o;
  }
  return op.result();
}

template <class T, int V>
__global__ __launch_bounds__(64) void stacked_mixed_kernel(const char* __restrict__ tab, int two_slots, const MixBlocks blocks, int n_blocks,
                                                         const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps, int rows_in, int rows_out,
                                                         int shift, int P, int n_logn, int64_t batch, int accumulate, const BjxFin fin, int gpb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[1];
  T* tile = reinterpret_cast<T*>(smem);
  T* logn = tile + (size_t)64 * P;
  const int lane = threadIdx.x;
  for (int i = lane; i < n_logn; i += 64) logn[i] = d_log(T(i));
  double acc = 0.0;
  // gpb groups of 64 columns per block (short columns: a group is a few hundred bytes)
  for (int gi = 0; gi < gpb; ++gi) {
  const int64_t col0 = ((int64_t)blockIdx.x * gpb + gi) * 64;
  if (col0 >= batch) break;
  const int ncols = (int)((batch - col0) < 64 ? (batch - col0) : 64);
  if (gi) tile_sync();                                                 // the previous group's stores have read the tile
  tile_stage_in<T, V>(tile + shift, in + col0 * rows_in, rows_in, P, ncols, lane);
  tile_sync();
  T lres = T(0);
  if (lane < ncols) {
    T* mine = tile + lane * P;
    const T* src = mine + shift;
    constexpr size_t RB = stacked_row_bytes<T>();
    int r = 0, bi = 0;
    while (r < rows_out) {
      const int stop = bi < n_blocks ? blocks.b[bi].out_lo : rows_out;
      // elementwise rows: the row's slots are the same for every lane (scalar loads).  Four rows at a time — the slot
      // fetches, then the LDS reads they address, then the evaluations: row by row the walk is one serial chain of
      // scalar-load -> LDS-read -> transcendental latencies per row (42 % of the roofline on an all-elementwise stack).
      for (; r + 4 <= stop; r += 4) {
        T v[4];
        {
          Slot<T> s0[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) s0[u] = reinterpret_cast<const Slot<T>*>(tab + (size_t)(r + u) * RB)[0];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = src[s0[u].src];
#pragma unroll
          for (int u = 0; u < 4; ++u) lres += slot_eval(s0[u], v[u]);
        }
        if (two_slots) {
          Slot<T> s1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) s1[u] = reinterpret_cast<const Slot<T>*>(tab + (size_t)(r + u) * RB)[1];
#pragma unroll
          for (int u = 0; u < 4; ++u) if (s1[u].kind != SK_END) lres += slot_eval(s1[u], v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mine[r + u] = v[u];
      }
      for (; r < stop; ++r) {
        const Slot<T>* sl = reinterpret_cast<const Slot<T>*>(tab + (size_t)r * RB);
        const Slot<T> s0 = sl[0];
        T v = src[s0.src];
        lres += slot_eval(s0, v);
        if (two_slots) { const Slot<T> s1 = sl[1]; if (s1.kind != SK_END) lres += slot_eval(s1, v); }
        mine[r] = v;
      }
      if (bi < n_blocks) {
        const MixBlock b = blocks.b[bi];
        const T* pin = src + b.in_lo;
        T* pout = mine + b.out_lo;
        switch (b.kind) {
          case MB_SIMPLEX: { SimplexFwd<T, true> op; op.K = b.len_in; lres += mixed_run_block<T>(op, pin, pout, b.len_in, b.len_out, logn); } break;
          case MB_SIMPLEX_INV: { SimplexInv<T, true> op; op.K = b.len_out; lres += mixed_run_block<T>(op, pin, pout, b.len_in, b.len_out, logn); } break;
          case MB_ORDERED: { OrderedFwd<T> op; lres += mixed_run_block<T>(op, pin, pout, b.len_in, b.len_out, logn); } break;
          default: { OrderedInv<T> op; lres += mixed_run_block<T>(op, pin, pout, b.len_in, b.len_out, logn); } break;
        }
        r += b.len_out;
        ++bi;
      }
    }
    if (ladj_ps) ladj_ps[col0 + lane] = accumulate ? ladj_ps[col0 + lane] + lres : lres;
  }
  tile_sync();
  tile_stage_out<T, V>(tile, out + col0 * rows_out, rows_out, P, ncols, lane);
  acc += lane < ncols ? (double)lres : 0.0;
  }
  block_publish_partial(acc, red, fin);
}

// SHORT columns without structured blocks (a model's joint link: Stacked(exp | Logit | identity ...) on 2 ... 10 parameters; same-box A/B: 68-73 % at 3 ... 7 rows, 56 % at 9 ... 10, level at 11, behind at 13): lane =
// column, the column read and written by its lane as multi-dword accesses (TinyCol), the slots of a row wave-uniform scalar loads, the
// gather `src` a select chain over the lane's DX registers.  No tile: at dim = 3 ... 7 the walker above spends ~150 VALU per row on
// staging and slot fetches (27-45 % of the HBM peak, profiles/r03_small_sizes.md).
template <class T, int DX>
__global__ __launch_bounds__(256) void stacked_tiny_kernel(const char* __restrict__ tab, int two_slots, const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps,
                                                           int64_t batch, int accumulate, const BjxFin fin) {
  __shared__ double red[4];
  constexpr size_t RB = stacked_row_bytes<T>();
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  T lres = T(0);
  if (col < batch) {
    const TinyCol<T, DX> t = *reinterpret_cast<const TinyCol<T, DX>*>(in + col * DX);
    TinyCol<T, DX> o;
#pragma unroll
    for (int r = 0; r < DX; ++r) {
      const Slot<T>* sl = reinterpret_cast<const Slot<T>*>(tab + (size_t)r * RB);
      const Slot<T> s0 = sl[0];
      T v = t.v[0];
#pragma unroll
      for (int k = 1; k < DX; ++k) v = s0.src == k ? t.v[k] : v;       // the source row is the same in every lane
      lres += slot_eval(s0, v);
      if (two_slots) { const Slot<T> s1 = sl[1]; if (s1.kind != SK_END) lres += slot_eval(s1, v); }
      o.v[r] = v;
    }
    *reinterpret_cast<TinyCol<T, DX>*>(out + col * DX) = o;
    if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + lres : lres;
  }
  block_publish_partial(col < batch ? (double)lres : 0.0, red, fin);
}

template <class T>
int stacked_mixed_impl(bjx_ctx* ctx, const bjx_segment* segs, int n_segs, const bjx_block* blocks, int n_blocks, const T* x, int64_t rows_in, T* y,
                       int64_t rows_out, T* ladj_ps, double* ladj_sum, int64_t batch, uint32_t flags) {
  if (batch == 0 || rows_out == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  BJX_REQUIRE(ctx, n_blocks <= MixBlocks::N, BJX_ERR_UNSUPPORTED, "bjx_stacked_mixed: %d structured blocks (max %d)", n_blocks, MixBlocks::N);
  // blocks: ascending and disjoint in the output AND the input; the elementwise segments cover the remaining output rows
  // (the structured rows carry identity placeholders, as for bjx_stacked_ld), with non-decreasing input ranges
  MixBlocks mb{};
  int64_t shift = rows_out > rows_in ? rows_out - rows_in : 0, max_len = 2, prev_out = 0, prev_in = 0;
  for (int i = 0; i < n_blocks; ++i) {
    const bjx_block& b = blocks[i];
    BJX_REQUIRE(ctx, b.kind >= MB_SIMPLEX && b.kind <= MB_ORDERED_INV, BJX_ERR_ARG, "bjx_stacked_mixed: block %d: bad kind %d", i, b.kind);
    const int64_t want_out = b.kind == MB_SIMPLEX ? b.len_in - 1 : (b.kind == MB_SIMPLEX_INV ? b.len_in + 1 : b.len_in);
    BJX_REQUIRE(ctx, b.len_in >= 1 && b.len_out == want_out && b.len_out >= 1, BJX_ERR_SHAPE, "bjx_stacked_mixed: block %d: %lld rows in, %lld rows out", i, (long long)b.len_in, (long long)b.len_out);
    BJX_REQUIRE(ctx, b.in_lo >= prev_in && b.out_lo >= prev_out && b.in_lo + b.len_in <= rows_in && b.out_lo + b.len_out <= rows_out, BJX_ERR_SHAPE,
                "bjx_stacked_mixed: block %d is out of order or outside the column", i);
    prev_in = b.in_lo + b.len_in; prev_out = b.out_lo + b.len_out;
    if (b.out_lo - b.in_lo > shift) shift = b.out_lo - b.in_lo;
    if (b.len_in + 1 > max_len) max_len = b.len_in + 1;
    if (b.len_out + 1 > max_len) max_len = b.len_out + 1;
    mb.b[i] = MixBlock{b.kind, (int)b.in_lo, (int)b.out_lo, (int)b.len_in, (int)b.len_out, 0};
  }
  {   // elementwise segments: ascending in the output with non-decreasing input rows (the walk is in place)
    int64_t po = -1, pi = -1;
    for (int s = 0; s < n_segs; ++s) {
      if (segs[s].len == 0) continue;
      BJX_REQUIRE(ctx, segs[s].out_lo > po && segs[s].in_lo >= pi, BJX_ERR_UNSUPPORTED, "bjx_stacked_mixed: segments must be listed in ascending row order");
      po = segs[s].out_lo; pi = segs[s].in_lo;
      if (segs[s].out_lo - segs[s].in_lo > shift) shift = segs[s].out_lo - segs[s].in_lo;
    }
  }
  // the slot table wants every output row once: identity placeholders on the rows of the structured blocks (never evaluated)
  std::vector<bjx_segment> full(segs, segs + n_segs);
  for (int i = 0; i < n_blocks; ++i) {
    bjx_segment ph{};
    ph.len = blocks[i].len_out;
    BJX_REQUIRE(ctx, ph.len <= rows_in, BJX_ERR_UNSUPPORTED, "bjx_stacked_mixed: block %d is longer than the input column", i);
    ph.in_lo = blocks[i].in_lo + ph.len <= rows_in ? blocks[i].in_lo : rows_in - ph.len;
    ph.out_lo = blocks[i].out_lo;
    ph.n_ops = 0;
    full.push_back(ph);
  }
  const int64_t rows_tile = rows_out > shift + rows_in ? rows_out : shift + rows_in;
  const int64_t P = rows_tile | 1;
  const size_t smem = ((size_t)64 * P + (size_t)max_len) * sizeof(T);
  BJX_REQUIRE(ctx, smem <= 64 * 1024, BJX_ERR_UNSUPPORTED, "bjx_stacked_mixed: columns of %lld rows exceed the LDS tile", (long long)rows_tile);
  StackedPlan pl;
  { int rc = stacked_prepare<T>(ctx, full.data(), (int)full.size(), x, y, rows_out, batch, false, false, &pl, rows_in, rows_out); if (rc) return rc; }
  static const int use_tiny = getenv("BJX_STACKED_TINY") ? atoi(getenv("BJX_STACKED_TINY")) : 1;
  if (use_tiny && n_blocks == 0 && rows_in == rows_out && shift == 0 && rows_out <= 10 && (rows_out % Vec16<T>::N != 0 || rows_out * sizeof(T) <= 16)) {
    const int64_t grid_t = (batch + 255) / 256;
    BJX_REQUIRE(ctx, grid_t < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
    BjxFin fin_t;
    bool second_t = false;
    { int rc = bjx_make_fin(ctx, grid_t, ladj_sum, 0.0, 0, flags, &fin_t, &second_t); if (rc) return rc; }
    const int accum_t = (flags & BJX_ACCUMULATE) ? 1 : 0;
#define BJX_ST(X_) hipLaunchKernelGGL((stacked_tiny_kernel<T, X_>), dim3((unsigned)grid_t), dim3(256), 0, ctx->stream, pl.tab, pl.two, x, y, ladj_ps, batch, accum_t, fin_t)
    bool launched = true;
    {
      BjxProf prof_(ctx);
      switch ((int)rows_out) {
        case 1: BJX_ST(1); break;
        case 2: BJX_ST(2); break;
        case 3: BJX_ST(3); break;
        case 4: BJX_ST(4); break;
        case 5: BJX_ST(5); break;
        case 6: BJX_ST(6); break;
        case 7: BJX_ST(7); break;
        case 9: BJX_ST(9); break;
        case 10: BJX_ST(10); break;
        default: launched = false; break;
      }
    }
#undef BJX_ST
    if (launched) {
      BJX_CHECK_LAUNCH(ctx);
      if (second_t) return bjx_launch_finalize(ctx, (int)grid_t, ladj_sum, 0.0, 0, 0.0, flags);
      return BJX_OK;
    }
  }
  // groups of 64 columns per block: up to three when a group is under ~4 KiB, while the grid keeps >= 16 384 blocks (same-box sweep
  // of BJX_MIXED_GPB = 1 / 2 / 3 / 4 / 6 / 8 at dim = 2, 3, 5, 10: +5-14 % at 2-3, nothing beyond — a wave that lives for more groups
  // leaves too few waves per CU)
  const int64_t groups = (batch + 63) / 64;
  static const int gpb_env = 0;
  int64_t gpb = gpb_env > 0 ? gpb_env : 8192 / (64 * (rows_in > rows_out ? rows_in : rows_out) * (int64_t)sizeof(T));
  if (gpb_env <= 0) { if (gpb > 3) gpb = 3; if (gpb > groups / 16384) gpb = groups / 16384; }
  gpb = gpb < 1 ? 1 : (gpb > 16 ? 16 : gpb);
  const int64_t grid = (groups + gpb - 1) / gpb;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  BjxFin fin;
  bool second = false;
  { int rc = bjx_make_fin(ctx, grid, ladj_sum, 0.0, 0, flags, &fin, &second); if (rc) return rc; }
  if (fin.counter) { fin.counter = nullptr; second = true; }           // single-wave blocks: two-pass finalize (see launch_seq)
  constexpr int VW = Vec16<T>::N;
  const bool v_ok = bjx_aligned16(x) && bjx_aligned16(y);
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  {
    BjxProf prof_(ctx);
    if (v_ok) hipLaunchKernelGGL((stacked_mixed_kernel<T, VW>), dim3((unsigned)grid), dim3(64), smem, ctx->stream, pl.tab, pl.two, mb, n_blocks, x, y, ladj_ps,
                                 (int)rows_in, (int)rows_out, (int)shift, (int)P, (int)max_len, batch, accum, fin, (int)gpb);
    else hipLaunchKernelGGL((stacked_mixed_kernel<T, 1>), dim3((unsigned)grid), dim3(64), smem, ctx->stream, pl.tab, pl.two, mb, n_blocks, x, y, ladj_ps,
                            (int)rows_in, (int)rows_out, (int)shift, (int)P, (int)max_len, batch, accum, fin, (int)gpb);
  }
  BJX_CHECK_LAUNCH(ctx);
  if (second) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

// ------------------------------------------------------------------ pullback (elementwise part)
// x_bar = (dy/dx) y_bar + ladj_bar (d ladj / dx), element by element through the same canonical slots:
//   u = a1 clamp(x) + b1,  v = N(u),  y = clamp(a2 v + b2):   dy/dx = a2 N'(u) a1 (0 where a clamp is active),
//   d ladj/dx = l_N'(u) a1.   N: exp (v, 1) | log (1/u, -1/u) | logit on [0, w] (w/(u(w-u)), -(w-2u)/(u(w-u))) |
//   logistic (s(1-s), 1-2s) | leaky (J, 0) | id (1, 0).   Two slots chain: x -> x1 -> y.
template <class T> __device__ __forceinline__ void slot_grad(const Slot<T>& s, T x, T& y, T& dy, T& dl) {
  using F = Fast<T>;
  const bool pre_active = x < s.clo || x > s.chi;
  const T xc = d_med3(x, s.clo, s.chi);
  const T u = s.a1 * xc + s.b1;
  T v = u, np = T(1), lp = T(0);
  switch (s.kind) {
    case SK_EXP: v = F::exp(u); np = v; lp = T(1); break;
    case SK_LOG: { const T r = F::rcp(u); v = F::log(u); np = r; lp = -r; } break;
    case SK_LOGIT: { const T q = s.alpha - u; const T r = F::rcp(u * q); v = F::log(u * F::rcp(q)); np = s.alpha * r; lp = -(q - u) * r; } break;
    case SK_LOGISTIC: { v = f_logistic(u); np = v * (T(1) - v); lp = T(1) - 2 * v; } break;
    case SK_LEAKY: { const T J = u < T(0) ? s.alpha : T(1); v = J * u; np = J; } break;
    default: break;
  }
  const T yr = s.a2 * v + s.b2;
  const bool post_active = yr < s.plo || yr > s.phi;
  y = d_med3(yr, s.plo, s.phi);
  const T m = pre_active ? T(0) : s.a1;
  dy = post_active ? T(0) : s.a2 * np * m;
  dl = lp * m;
}

// MOM: additionally the row moments of the result over the block's columns, Σ x̄ and Σ x̄·x per row (the parameter
// cotangents of a leading per-row affine stage, bjx_row_moments) -> mpart[blockIdx][2 dim] in Float64, so that the
// mean-field pullback does not read x̄ and x a second time (1 284 -> 772 + ~2 % B/sample).  One pack per lane
// (dim <= G·V), partial sums over the COL_UC columns of a lane, a fixed shuffle butterfly over the column groups of a wave, the four waves through LDS.
// UNAL (round 3, odd column heights from 80 rows; whole-segment rows in place, no fused moments): 16-byte packs on element-aligned
// addresses, and the dim mod V tail rows as one more unit of the same loop — the LAST V rows of the column (overlapping the pack
// before them), every element looking up its own place in the row-permuted table; the pullback is elementwise, so the overlap is
// simply recomputed and only the tail rows are stored.  The tile walker these heights used ran the chain pullback at 18 / 32 % of the
// HBM peak at 101 / 201 rows (70 % at 100 rows on this kernel).
// SLAB (round 4): columns of more than G units in ONE launch — blockIdx.y picks a slab of G units, the block stages only that slab's
// part of the row-permuted table (V runs of G entries -> [V][G] in LDS; the whole table of a tall column does not fit) and every
// lane owns one unit.  The pullback is elementwise: slabs need nothing from each other, so they are blocks of one grid instead of the
// launches of a host loop (which is what the forward maps use, because their log-dets accumulate in launch order) — at 16 columns of
// 5 000 rows the host loop was 20 launch / table-build pairs, 375 µs a call.
template <class T, int V, bool GATHER, bool IN_LDS, bool MOM = false, bool UNAL = false, bool SLAB = false>
__global__ __launch_bounds__(256) void stacked_vjp_kernel(const char* __restrict__ tab_g, int two_slots, const T* __restrict__ x, const T* __restrict__ ybar,
                                                          const T* __restrict__ lbar, T* __restrict__ xbar, int64_t dim, int64_t batch, int G,
                                                          double* __restrict__ mpart = nullptr, int mom_off = 0, int64_t ld = 0) {
  // ld: elements between the starts of consecutive columns when x / ȳ / x̄ point at a ROW WINDOW of taller columns (0: dense, = dim)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t cs = ld ? ld : dim;
  T ms1[V], ms2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { ms1[j] = T(0); ms2[j] = T(0); }
  const int64_t nvc = dim / V;
  // SLAB: `mom_off` carries the number of slabs; the slab index runs FASTEST over the grid (neighbouring blocks work on the same columns)
  const int64_t v0 = SLAB ? (int64_t)(blockIdx.x % (unsigned)mom_off) * G : 0;          // first unit of my slab
  const int64_t cblk = SLAB ? blockIdx.x / (unsigned)mom_off : blockIdx.x;               // my block of columns
  if (IN_LDS && !SLAB) {
    const int n16 = (int)(dim * stacked_row_bytes<T>() / 16);
    const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(tab_g);
    bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(smem);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  if (IN_LDS && SLAB) {
    constexpr int RB16 = (int)(stacked_row_bytes<T>() / 16);
    const bjx_f32x4* src = reinterpret_cast<const bjx_f32x4*>(tab_g);
    bjx_f32x4* dst = reinterpret_cast<bjx_f32x4*>(smem);
    for (int i = threadIdx.x; i < V * G * RB16; i += blockDim.x) {
      const int q = i % RB16, en = i / RB16, vv = en & (G - 1), j = en / G;
      if (v0 + vv < nvc) dst[i] = src[((int64_t)j * nvc + v0 + vv) * RB16 + q];
    }
    if (UNAL && v0 <= nvc && nvc < v0 + G) {
      // the tail unit lives in this slab: its V rows (the last V of the column; they overlap the pack before it, which may belong to
      // the previous slab) behind the slab's entries — every slot read of this kernel stays an LDS read (a pointer chosen between
      // the global table and LDS compiles to flat loads: 56 -> 40 % of the HBM peak at 1 001 rows)
      for (int i = threadIdx.x; i < V * RB16; i += blockDim.x) {
        const int q = i % RB16, j = i / RB16;
        dst[V * G * RB16 + i] = src[stacked_row_index(dim - V + j, V, nvc) * RB16 + q];
      }
    }
    __syncthreads();
  }
  const char* t = IN_LDS ? smem : tab_g;
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = blockDim.x / G;
  const int tail = UNAL ? (int)(dim - nvc * V) : 0;
  const int64_t nun = nvc + (tail ? 1 : 0);           // units of a column: the whole packs and (UNAL) the tail
  for (int uc = 0; uc < COL_UC; ++uc) {
    const int64_t col = (cblk * COL_UC + uc) * cols_per_block + threadIdx.x / G;
    if (col >= batch) continue;
    const T lb = lbar ? lbar[col] : T(0);
    const T* xc = x + col * cs;
    const T* gc = ybar + col * cs;
    T* oc = xbar + col * cs;
    for (int64_t v = v0 + gl; v < (SLAB && v0 + G < nun ? v0 + G : nun); v += G) {
      const bool is_tail = UNAL && v == nvc;
      const int64_t prow = is_tail ? dim - V : v * V;
      Pack<T, V> px, pg;
      if (!GATHER) px = load_pack<T, V, true>(xc + prow);
      pg = load_pack<T, V, true>(gc + prow);
      const Pack<T, V> pin = px;                      // the inputs (px is overwritten by the results row by row)
#pragma unroll 1
      for (int j = 0; j < V; ++j) {
                const Slot<T>* e = (SLAB && IN_LDS)
            ? reinterpret_cast<const Slot<T>*>(t + (size_t)(is_tail ? V * G + j : j * G + gl) * stacked_row_bytes<T>())
            : reinterpret_cast<const Slot<T>*>(t + (is_tail ? stacked_row_index(prow + j, V, nvc) : (V > 1 ? j * nvc + v : v)) * stacked_row_bytes<T>());
        const Slot<T> s0 = e[0];
        T xv = px.v[0], gv = pg.v[0];
        if (V > 1) {
          xv = j == 1 ? px.v[1 % V] : (j == 2 ? px.v[2 % V] : (j == 3 ? px.v[3 % V] : xv));
          gv = j == 1 ? pg.v[1 % V] : (j == 2 ? pg.v[2 % V] : (j == 3 ? pg.v[3 % V] : gv));
        }
        if (GATHER) xv = xc[s0.src];
        T x1, dy0, dl0;
        slot_grad<T>(s0, xv, x1, dy0, dl0);
        T g = gv;                                     // cotangent of the value entering the remaining slots
        if (two_slots) {
          const Slot<T> s1 = e[1];
          if (s1.kind != SK_END) {
            T y2, dy1, dl1;
            slot_grad<T>(s1, x1, y2, dy1, dl1);
            g = gv * dy1 + lb * dl1;
          }
        }
        const T res = g * dy0 + lb * dl0;
        if (GATHER) oc[s0.src] = res;
        else if (V == 1) px.v[0] = res;
        else {
          px.v[0] = j == 0 ? res : px.v[0];
          px.v[1 % V] = j == 1 ? res : px.v[1 % V];
          px.v[2 % V] = j == 2 ? res : px.v[2 % V];
          px.v[3 % V] = j == 3 ? res : px.v[3 % V];
        }
      }
      if (MOM && !GATHER) {
#pragma unroll
        for (int j = 0; j < V; ++j) { ms1[j] += px.v[j]; ms2[j] += px.v[j] * pin.v[j]; }
      }
      if (!GATHER) {
        if (!is_tail) store_pack<T, V, true>(oc + prow, px);
        else store_pack_run<T, V>(oc + prow, px, V - tail, tail);
      }
    }
  }
  if (MOM) {
    // the column groups of a wave first (lanes gl, gl + G, ...: a fixed butterfly), then the four waves through
    // [wave][row][2] doubles behind the slot table; lanes past the last pack hold zeros
#pragma unroll
    for (int j = 0; j < V; ++j) {
      for (int m = G; m < 64; m <<= 1) { ms1[j] += shfl_xor(ms1[j], m); ms2[j] += shfl_xor(ms2[j], m); }
    }
    double* mp = reinterpret_cast<double*>(smem + mom_off);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < G && gl < nvc) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        mp[((size_t)wv * dim + gl * V + j) * 2] = (double)ms1[j];
        mp[((size_t)wv * dim + gl * V + j) * 2 + 1] = (double)ms2[j];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * dim; e += blockDim.x) {
      const int row = e % (int)dim, which = e / (int)dim;
      double acc = 0.0;
      for (int c = 0; c < 4; ++c) acc += mp[((size_t)c * dim + row) * 2 + which];
      mpart[(size_t)blockIdx.x * 2 * dim + e] = acc;
    }
  }
}

// sum `chunk` consecutive partial sets of `per` doubles each: out[b][e] = Σ_{k in chunk b} in[k][e] (fixed order, coalesced)
__global__ __launch_bounds__(256) void sets_reduce_kernel(const double* __restrict__ in, int64_t nsets, int per, int chunk, double* __restrict__ out) {
  const int64_t k0 = (int64_t)blockIdx.x * chunk;
  const int64_t k1 = k0 + chunk < nsets ? k0 + chunk : nsets;
  for (int e = threadIdx.x; e < per; e += blockDim.x) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t k = k0;
    for (; k + 4 <= k1; k += 4) {
      a0 += in[k * per + e]; a1 += in[(k + 1) * per + e]; a2 += in[(k + 2) * per + e]; a3 += in[(k + 3) * per + e];
    }
    for (; k < k1; ++k) a0 += in[k * per + e];
    out[(size_t)blockIdx.x * per + e] = (a0 + a1) + (a2 + a3);
  }
}
__global__ void moments_count_kernel(double* out, int64_t dim, int64_t batch) { if (threadIdx.x == 0 && blockIdx.x == 0) out[2 * dim] = (double)batch; }

// Column-walker form of the pullback for columns that are not whole 16-byte packs (cf. stacked_mixed_kernel): x and ȳ of 64
// consecutive columns through two odd-pitch LDS tiles, lane = column, the row's slots as wave-uniform scalar loads, x̄ written
// over x in the tile (every input row is the source of exactly one output row) and stored as one contiguous run.
template <class T, int V>
__global__ __launch_bounds__(64) void stacked_vjp_walk_kernel(const char* __restrict__ tab, int two_slots, const T* __restrict__ x, const T* __restrict__ ybar,
                                                            const T* __restrict__ lbar, T* __restrict__ xbar, int dim, int P, int64_t batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tx = reinterpret_cast<T*>(smem);
  T* tg = tx + (size_t)64 * P;
  const int lane = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int ncols = (int)((batch - c0) < 64 ? (batch - c0) : 64);
  tile_stage_in<T, V>(tx, x + c0 * dim, dim, P, ncols, lane);
  tile_stage_in<T, V>(tg, ybar + c0 * dim, dim, P, ncols, lane);
  tile_sync();
  if (lane < ncols) {
    T* mx = tx + lane * P;
    const T* mg = tg + lane * P;
    const T lb = lbar ? lbar[c0 + lane] : T(0);
    constexpr size_t RB = stacked_row_bytes<T>();
    for (int r = 0; r < dim; ++r) {
      const Slot<T>* e = reinterpret_cast<const Slot<T>*>(tab + (size_t)r * RB);
      const Slot<T> s0 = e[0];
      T x1, dy0, dl0;
      slot_grad<T>(s0, mx[s0.src], x1, dy0, dl0);
      T g = mg[r];
      if (two_slots) {
        const Slot<T> s1 = e[1];
        if (s1.kind != SK_END) {
          T y2, dy1, dl1;
          slot_grad<T>(s1, x1, y2, dy1, dl1);
          g = g * dy1 + lb * dl1;
        }
      }
      mx[s0.src] = g * dy0 + lb * dl0;
    }
  }
  tile_sync();
  tile_stage_out<T, V>(tx, xbar + c0 * dim, dim, P, ncols, lane);
}

// The same on SHORT columns (cf. stacked_tiny_kernel): lane = column, x and ȳ read and x̄ written as whole columns, the scatter to
// the source row a select chain over the lane's registers.
template <class T, int DX>
__global__ __launch_bounds__(256) void stacked_vjp_tiny_kernel(const char* __restrict__ tab, int two_slots, const T* __restrict__ x, const T* __restrict__ ybar,
                                                               const T* __restrict__ lbar, T* __restrict__ xbar, int64_t batch) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= batch) return;
  const TinyCol<T, DX> tx = *reinterpret_cast<const TinyCol<T, DX>*>(x + col * DX);
  const TinyCol<T, DX> tg = *reinterpret_cast<const TinyCol<T, DX>*>(ybar + col * DX);
  TinyCol<T, DX> o = tx;
  const T lb = lbar ? lbar[col] : T(0);
  constexpr size_t RB = stacked_row_bytes<T>();
#pragma unroll
  for (int r = 0; r < DX; ++r) {
    const Slot<T>* e = reinterpret_cast<const Slot<T>*>(tab + (size_t)r * RB);
    const Slot<T> s0 = e[0];
    T xin = tx.v[0];
#pragma unroll
    for (int k = 1; k < DX; ++k) xin = s0.src == k ? tx.v[k] : xin;    // the source row is the same in every lane
    T x1, dy0, dl0;
    slot_grad<T>(s0, xin, x1, dy0, dl0);
    T g = tg.v[r];
    if (two_slots) {
      const Slot<T> s1 = e[1];
      if (s1.kind != SK_END) {
        T y2, dy1, dl1;
        slot_grad<T>(s1, x1, y2, dy1, dl1);
        g = g * dy1 + lb * dl1;
      }
    }
    const T res = g * dy0 + lb * dl0;
#pragma unroll
    for (int k = 0; k < DX; ++k) o.v[k] = s0.src == k ? res : o.v[k];
  }
  *reinterpret_cast<TinyCol<T, DX>*>(xbar + col * DX) = o;
}

template <class T>
int stacked_vjp_impl(bjx_ctx* ctx, const bjx_segment* segs, int n_segs, const T* x, const T* ybar, const T* lbar, T* xbar, int64_t dim, int64_t batch,
                     double* moments = nullptr, bool* moments_done = nullptr, int64_t ld = 0) {
  // ld != 0: `dim` rows of columns that are ld apart (a row slab of the loop below)
  if (moments_done) *moments_done = false;
  if (dim * batch == 0) return BJX_OK;
  if (moments) {
    // The row moments ride along only in the one-pack-per-lane form on whole aligned packs (below).  Every other shape — odd heights,
    // more than 64 packs per column — takes the best plain pullback (element-aligned packs, row slabs) and leaves the moments to the
    // caller's second pass (bjx_row_moments): asked WITH moments those shapes used to fall to the column loop (the mean-field
    // parameter pullback at 1 001 rows: 15.6 ms, of which 2.9 for the pullback proper and 1.7 for the moments).
    constexpr int VWm = Vec16<T>::N;
    const bool fusable = dim % VWm == 0 && dim / VWm <= 64 && bjx_aligned16(x) && bjx_aligned16(ybar) && bjx_aligned16(xbar);
    if (!fusable) return stacked_vjp_impl<T>(ctx, segs, n_segs, x, ybar, lbar, xbar, dim, batch, nullptr, nullptr, ld);
  }
  {
    static const int use_tiny = getenv("BJX_STACKED_TINY") ? atoi(getenv("BJX_STACKED_TINY")) : 1;
    if (use_tiny && !moments && ld == 0 && dim <= 7 && dim % Vec16<T>::N != 0) {     // same-box A/B: 61-78 % against 28-51 % at 2-5 rows, level from 7
      StackedPlan plt;
      { int rc = stacked_prepare<T>(ctx, segs, n_segs, x, xbar, dim, batch, false, false, &plt); if (rc) return rc; }   // V = 1: unpermuted table
      const int64_t grid_t = (batch + 255) / 256;
      BJX_REQUIRE(ctx, grid_t < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_stacked_vjp: batch too large for one launch");
#define BJX_SVT(X_) hipLaunchKernelGGL((stacked_vjp_tiny_kernel<T, X_>), dim3((unsigned)grid_t), dim3(256), 0, ctx->stream, plt.tab, plt.two, x, ybar, lbar, xbar, batch)
      bool launched = true;
      {
        BjxProf prof_(ctx);
        switch ((int)dim) {
          case 1: BJX_SVT(1); break;
          case 2: BJX_SVT(2); break;
          case 3: BJX_SVT(3); break;
          case 5: BJX_SVT(5); break;
          case 6: BJX_SVT(6); break;
          case 7: BJX_SVT(7); break;
          default: launched = false; break;
        }
      }
#undef BJX_SVT
      if (launched) { BJX_CHECK_LAUNCH(ctx); return BJX_OK; }
    }
    // odd heights, rows in place: the group kernel on element-aligned packs (UNAL)
    static const int use_unal_vjp = getenv("BJX_STACKED_VJP_UNALIGNED") ? atoi(getenv("BJX_STACKED_VJP_UNALIGNED")) : 1;
    // from 24 rows (the forward maps switch at 80): same call, 2^22 columns, walker / group kernel — 17 rows 60 / 48 %, 21 rows 56 / 56,
    // 25 rows 53 / 62, 29 rows 49 / 64, 45 rows 37 / 59, 61 rows 29 / 65, 77 rows 24 / 52 % of the HBM peak
    constexpr int vjp_unal_from = 24;
    if constexpr (Vec16<T>::N > 1) {
      bool in_place_rows = true;
      for (int sgi = 0; sgi < n_segs; ++sgi) in_place_rows = in_place_rows && segs[sgi].in_lo == segs[sgi].out_lo;
      // (a slab of columns whose pitch is not a whole number of packs is element-aligned whatever its own height is)
      if (use_unal_vjp && !moments && in_place_rows && (dim % Vec16<T>::N != 0 || ld % Vec16<T>::N != 0) && col_launch_cfg<T>(ctx, x, xbar, dim, batch, ld, ld, true, vjp_unal_from).unal) {
        constexpr int VWu = Vec16<T>::N;
        StackedPlan plu;
        { int rc = stacked_prepare<T>(ctx, segs, n_segs, x, xbar, dim, batch, false, true, &plu, ld, ld, true, vjp_unal_from); if (rc) return rc; }
        if (plu.V == VWu && !plu.gather) {
          const bool ldsu = plu.tab_bytes <= 48 * 1024;
          const size_t smemu = ldsu ? plu.tab_bytes : 0;
          const int64_t units = dim / VWu + 1;
          int Gu = 1;
          while (Gu < 64 && Gu < units) Gu <<= 1;
          if (units > 64) Gu = stacked_slab_lanes(units);
          const int64_t cpbu = (int64_t)(256 / Gu) * COL_UC;
          const int64_t gridu = (batch + cpbu - 1) / cpbu;
          BJX_REQUIRE(ctx, gridu < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_stacked_vjp: batch too large for one launch");
          const int64_t nslab = (units + Gu - 1) / Gu;               // > 1: more than 64 units per column — slabs of one grid (SLAB)
          BJX_REQUIRE(ctx, gridu * nslab < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_stacked_vjp: too many rows for one launch");
          {
            BjxProf prof_(ctx);
            if (nslab > 1) hipLaunchKernelGGL((stacked_vjp_kernel<T, VWu, false, true, false, true, true>), dim3((unsigned)(gridu * nslab)), dim3(256), (size_t)VWu * (Gu + 1) * stacked_row_bytes<T>(), ctx->stream, plu.tab, plu.two, x, ybar, lbar, xbar, dim, batch, Gu, (double*)nullptr, (int)nslab, ld);
            else if (ldsu) hipLaunchKernelGGL((stacked_vjp_kernel<T, VWu, false, true, false, true>), dim3((unsigned)gridu), dim3(256), smemu, ctx->stream, plu.tab, plu.two, x, ybar, lbar, xbar, dim, batch, Gu, (double*)nullptr, 0, ld);
            else hipLaunchKernelGGL((stacked_vjp_kernel<T, VWu, false, false, false, true>), dim3((unsigned)gridu), dim3(256), smemu, ctx->stream, plu.tab, plu.two, x, ybar, lbar, xbar, dim, batch, Gu, (double*)nullptr, 0, ld);
          }
          BJX_CHECK_LAUNCH(ctx);
          return BJX_OK;
        }
      }
    }
    static const int use_walker = getenv("BJX_STACKED_WALKER") ? atoi(getenv("BJX_STACKED_WALKER")) : 1;
    const int64_t P = dim | 1;
    const size_t smem_w = (size_t)2 * 64 * P * sizeof(T);
    if (use_walker && !moments && ld == 0 && dim % Vec16<T>::N != 0 && smem_w <= 64 * 1024 && (const void*)x != (const void*)xbar) {
      StackedPlan plw;
      { int rc = stacked_prepare<T>(ctx, segs, n_segs, x, xbar, dim, batch, false, false, &plw); if (rc) return rc; }   // V = 1: unpermuted table
      const int64_t grid_w = (batch + 63) / 64;
      BJX_REQUIRE(ctx, grid_w < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_stacked_vjp: batch too large for one launch");
      const bool vec = bjx_aligned16(x) && bjx_aligned16(ybar) && bjx_aligned16(xbar);
      {
        BjxProf prof_(ctx);
        if (vec) hipLaunchKernelGGL((stacked_vjp_walk_kernel<T, Vec16<T>::N>), dim3((unsigned)grid_w), dim3(64), smem_w, ctx->stream, plw.tab, plw.two, x, ybar, lbar, xbar, (int)dim, (int)P, batch);
        else hipLaunchKernelGGL((stacked_vjp_walk_kernel<T, 1>), dim3((unsigned)grid_w), dim3(64), smem_w, ctx->stream, plw.tab, plw.two, x, ybar, lbar, xbar, (int)dim, (int)P, batch);
      }
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
  }
  StackedPlan pl;
  { int rc = stacked_prepare<T>(ctx, segs, n_segs, x, xbar, dim, batch, false, bjx_aligned16(ybar), &pl, ld, ld); if (rc) return rc; }
  const bool lds = pl.tab_bytes <= 48 * 1024;
  const size_t smem = lds ? pl.tab_bytes : 0;
  const int64_t packs = dim / pl.V;
  int G = 1;
  while (G < 64 && G < packs) G <<= 1;
  if (packs > 64 && !pl.gather && !moments) G = stacked_slab_lanes(packs);
  const int64_t cpb = (int64_t)(256 / G) * COL_UC;
  const int64_t grid = (batch + cpb - 1) / cpb;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "bjx_stacked_vjp: batch too large for one launch");
  constexpr int VW = Vec16<T>::N;
  // fused row moments: one pack per lane, rows in place (no gather), LDS for the [column group][row][2] combine
  {
    const size_t mom_off = (smem + 15) / 16 * 16;
    const size_t mom_bytes = (size_t)4 * dim * 2 * sizeof(double);
    const size_t sets = (size_t)grid, per = (size_t)2 * dim;
    const size_t stage1 = (sets + 255) / 256;
    if (moments && !pl.gather && packs <= G && mom_off + mom_bytes <= 96 * 1024 && stage1 <= 4096) {
      { int rc = bjx_ensure_partials(ctx, (sets + stage1) * per); if (rc) return rc; }
      double* mpart = ctx->partials;
      double* st1 = mpart + sets * per;
      const size_t smem_m = mom_off + mom_bytes;
#define SVJM(V_, L_) do { bjx_allow_big_lds(stacked_vjp_kernel<T, V_, false, L_, true>, smem_m); hipLaunchKernelGGL((stacked_vjp_kernel<T, V_, false, L_, true>), dim3((unsigned)grid), dim3(256), smem_m, ctx->stream, pl.tab, pl.two, x, ybar, lbar, xbar, dim, batch, G, mpart, (int)mom_off); } while (0)
      {
        BjxProf prof_(ctx);
        if (pl.V == VW) { if (lds) SVJM(VW, true); else SVJM(VW, false); } else { if (lds) SVJM(1, true); else SVJM(1, false); }
      }
#undef SVJM
      BJX_CHECK_LAUNCH(ctx);
      {
        BjxProf prof_(ctx);
        hipLaunchKernelGGL(sets_reduce_kernel, dim3((unsigned)stage1), dim3(256), 0, ctx->stream, mpart, (int64_t)sets, (int)per, 256, st1);
        hipLaunchKernelGGL(sets_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st1, (int64_t)stage1, (int)per, (int)stage1, moments);
        hipLaunchKernelGGL(moments_count_kernel, dim3(1), dim3(64), 0, ctx->stream, moments, dim, batch);
      }
      BJX_CHECK_LAUNCH(ctx);
      *moments_done = true;
      return BJX_OK;
    }
  }
#define SVJP(V_, G_, L_) hipLaunchKernelGGL((stacked_vjp_kernel<T, V_, G_, L_>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, pl.tab, pl.two, x, ybar, lbar, xbar, dim, batch, G, (double*)nullptr, 0, ld)
#define SVJP_V(V_) do { if (pl.gather) { if (lds) SVJP(V_, true, true); else SVJP(V_, true, false); } else { if (lds) SVJP(V_, false, true); else SVJP(V_, false, false); } } while (0)
  {
    BjxProf prof_(ctx);
    const int64_t nslab = (packs + G - 1) / G;
    if (nslab > 1 && !pl.gather && grid * nslab < (int64_t)1 << 31) {              // rows in place, more than 64 packs per column: slabs of one grid (SLAB)
      const size_t smem_s = (size_t)pl.V * G * stacked_row_bytes<T>();
      if (pl.V == VW) hipLaunchKernelGGL((stacked_vjp_kernel<T, VW, false, true, false, false, true>), dim3((unsigned)(grid * nslab)), dim3(256), smem_s, ctx->stream, pl.tab, pl.two, x, ybar, lbar, xbar, dim, batch, G, (double*)nullptr, (int)nslab, ld);
      else hipLaunchKernelGGL((stacked_vjp_kernel<T, 1, false, true, false, false, true>), dim3((unsigned)(grid * nslab)), dim3(256), smem_s, ctx->stream, pl.tab, pl.two, x, ybar, lbar, xbar, dim, batch, G, (double*)nullptr, (int)nslab, ld);
    }
    else if (pl.V == VW) SVJP_V(VW); else SVJP_V(1);
  }
#undef SVJP_V
#undef SVJP
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_stacked(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x, void* y, void* ladj_ps,
                        double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0 && n_segs >= 0, BJX_ERR_SHAPE, "bjx_stacked: negative size");
  BJX_REQUIRE(ctx, (segs || n_segs == 0) && ((x && y) || dim * batch == 0), BJX_ERR_ARG, "bjx_stacked: null pointer");
  BJX_REQUIRE(ctx, dim < ((int64_t)1 << 31), BJX_ERR_UNSUPPORTED, "bjx_stacked: too many rows");
  if (dt == BJX_F32) return stacked_impl<float>(ctx, segs, n_segs, (const float*)x, (float*)y, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return stacked_impl<double>(ctx, segs, n_segs, (const double*)x, (double*)y, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_stacked: bad dtype %d", (int)dt);
}

/* bjx_stacked with separate leading dimensions: x is [ldx, batch], y is [ldy, batch], the segments produce the first `dim`
 * rows of y (from `y` on) out of rows of x (from `x` on); rows that keep their offset (in_lo == out_lo for every segment: a
 * WINDOW of the two matrices) stream as packs, otherwise every row is gathered.  What a Stacked with structured segments needs (stacked.jl:142-166):
 * the elementwise segments in ONE launch here (the rows of the structured segments are covered by identity placeholders),
 * then the structured entry points with a leading dimension (bjx_simplex_ld, bjx_ordered_ld) overwrite their rows in place. */
BJX_API int bjx_stacked_ld(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x, int64_t ldx, void* y, int64_t ldy,
                           void* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0 && n_segs >= 0 && ldx >= 0 && ldy >= dim, BJX_ERR_SHAPE, "bjx_stacked_ld: bad size");
  BJX_REQUIRE(ctx, (segs || n_segs == 0) && ((x && y) || dim * batch == 0), BJX_ERR_ARG, "bjx_stacked_ld: null pointer");
  BJX_REQUIRE(ctx, x != y, BJX_ERR_ARG, "bjx_stacked_ld: in-place is not supported");
  BJX_REQUIRE(ctx, dim < ((int64_t)1 << 31) && ldx < ((int64_t)1 << 31), BJX_ERR_UNSUPPORTED, "bjx_stacked_ld: too many rows");
  if (dt == BJX_F32) return stacked_impl<float>(ctx, segs, n_segs, (const float*)x, (float*)y, (float*)ladj_ps, ladj_sum, dim, batch, flags, ldx, ldy);
  if (dt == BJX_F64) return stacked_impl<double>(ctx, segs, n_segs, (const double*)x, (double*)y, (double*)ladj_ps, ladj_sum, dim, batch, flags, ldx, ldy);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_stacked_ld: bad dtype %d", (int)dt);
}

BJX_API int bjx_stacked_mixed(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const bjx_block* blocks, int n_blocks, const void* x,
                              int64_t rows_in, void* y, int64_t rows_out, void* ladj_ps, double* ladj_sum, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, rows_in >= 0 && rows_out >= 0 && batch >= 0 && n_segs >= 0 && n_blocks >= 0, BJX_ERR_SHAPE, "bjx_stacked_mixed: bad size");
  BJX_REQUIRE(ctx, (segs || n_segs == 0) && (blocks || n_blocks == 0) && ((x && y) || rows_out * batch == 0), BJX_ERR_ARG, "bjx_stacked_mixed: null pointer");
  BJX_REQUIRE(ctx, x != y, BJX_ERR_ARG, "bjx_stacked_mixed: in-place is not supported");
  BJX_REQUIRE(ctx, rows_in < (1 << 20) && rows_out < (1 << 20), BJX_ERR_UNSUPPORTED, "bjx_stacked_mixed: too many rows");
  if (dt == BJX_F32) return stacked_mixed_impl<float>(ctx, segs, n_segs, blocks, n_blocks, (const float*)x, rows_in, (float*)y, rows_out, (float*)ladj_ps, ladj_sum, batch, flags);
  if (dt == BJX_F64) return stacked_mixed_impl<double>(ctx, segs, n_segs, blocks, n_blocks, (const double*)x, rows_in, (double*)y, rows_out, (double*)ladj_ps, ladj_sum, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_stacked_mixed: bad dtype %d", (int)dt);
}

BJX_API int bjx_stacked_vjp_moments(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x, const void* y_bar,
                                    const void* ladj_bar, void* x_bar, double* moments, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0 && n_segs >= 0, BJX_ERR_SHAPE, "bjx_stacked_vjp_moments: bad size");
  BJX_REQUIRE(ctx, moments && (segs || n_segs == 0) && ((x && y_bar && x_bar) || batch == 0), BJX_ERR_ARG, "bjx_stacked_vjp_moments: null pointer");
  BJX_REQUIRE(ctx, x_bar != x || batch == 0, BJX_ERR_ARG, "bjx_stacked_vjp_moments: x_bar may not alias x");
  BJX_REQUIRE(ctx, dim < ((int64_t)1 << 31), BJX_ERR_UNSUPPORTED, "bjx_stacked_vjp_moments: too many rows");
  bool done = false;
  int rc;
  if (dt == BJX_F32) rc = stacked_vjp_impl<float>(ctx, segs, n_segs, (const float*)x, (const float*)y_bar, (const float*)ladj_bar, (float*)x_bar, dim, batch, moments, &done);
  else if (dt == BJX_F64) rc = stacked_vjp_impl<double>(ctx, segs, n_segs, (const double*)x, (const double*)y_bar, (const double*)ladj_bar, (double*)x_bar, dim, batch, moments, &done);
  else return bjx_fail(ctx, BJX_ERR_ARG, "bjx_stacked_vjp_moments: bad dtype %d", (int)dt);
  if (rc || done) return rc;
  return bjx_row_moments(ctx, dt, x_bar, x, moments, dim, batch);     // shapes outside the fused kernel: a second pass
}

BJX_API int bjx_stacked_vjp(bjx_ctx* ctx, bjx_dtype dt, const bjx_segment* segs, int n_segs, const void* x, const void* y_bar,
                            const void* ladj_bar, void* x_bar, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 0 && batch >= 0 && n_segs >= 0, BJX_ERR_SHAPE, "bjx_stacked_vjp: negative size");
  BJX_REQUIRE(ctx, (segs || n_segs == 0) && ((x && y_bar && x_bar) || dim * batch == 0), BJX_ERR_ARG, "bjx_stacked_vjp: null pointer");
  BJX_REQUIRE(ctx, x_bar != x || dim * batch == 0, BJX_ERR_ARG, "bjx_stacked_vjp: x_bar may not alias x");
  BJX_REQUIRE(ctx, dim < ((int64_t)1 << 31), BJX_ERR_UNSUPPORTED, "bjx_stacked_vjp: too many rows");
  if (dt == BJX_F32) return stacked_vjp_impl<float>(ctx, segs, n_segs, (const float*)x, (const float*)y_bar, (const float*)ladj_bar, (float*)x_bar, dim, batch);
  if (dt == BJX_F64) return stacked_vjp_impl<double>(ctx, segs, n_segs, (const double*)x, (const double*)y_bar, (const double*)ladj_bar, (double*)x_bar, dim, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_stacked_vjp: bad dtype %d", (int)dt);
}
