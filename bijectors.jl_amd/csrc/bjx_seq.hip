// bjx_seq.hip — F3: bijectors with a sequential / prefix dependency along `dim`
// (SURVEY.md §8a rows a9-a14).
//   OrderedBijector      ordered.jl:24-80
//   SimplexBijector      simplex.jl:28-143
//   VecCholeskyBijector  corr.jl:227-254, 314-337, 370-399, 485-501
//
// Ordered / Simplex: the block stages a [256 columns x C rows] tile through padded LDS with
// coalesced global accesses (a lane pair of 128 B per column chunk), then ONE LANE PER SAMPLE walks
// its column in ascending row order — exactly the reference's summation order — writing results
// back into the same tile, which is then stored coalesced.  LDS pitch C+1 makes the column walk
// bank-conflict free (lane t reads tile[t*(C+1)+i]).
//
// VecCholesky: one WAVE per sample walks the packed strict-upper vector 64 entries at a time; the
// per-column running sums (Σ logcosh for the inverse, Σ w² for the forward link) are segmented
// wave scans built from one inclusive shuffle scan + one bpermute gather.
#include <cstdlib>

#include "bjx_internal.h"
#include "bjx_tile.h"

namespace {
using namespace bjx;

template <class T> struct SeqCfg { static constexpr int C = 128 / sizeof(T); static constexpr int NT = 256; };

#include "bjx_seqops.h"

template <class T, class Op>
__global__ __launch_bounds__(256) void seq_kernel(Op op0, const T* in, T* out, T* ladj_ps, int64_t rows_in, int64_t rows_out,
                                                  int64_t batch, int n_logk, int accumulate, double* partials) {
  constexpr int C = SeqCfg<T>::C, NT = SeqCfg<T>::NT, P = C + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  T* tile = reinterpret_cast<T*>(smem + 32);
  T* logk = tile + (size_t)NT * P;
  // log(T(K-1-i)) table (simplex.jl:35,41); broadcast LDS reads in the walk
  for (int i = threadIdx.x; i < n_logk; i += NT) logk[i] = d_log(T(n_logk - i));
  const int64_t rows = rows_in > rows_out ? rows_in : rows_out;
  const int li = threadIdx.x & (C - 1);       // row inside a chunk handled by this lane in load/store
  const int lc0 = threadIdx.x / C;            // first column handled by this lane in load/store
  double acc = 0.0;
  {  // non-persistent: block b owns columns [b*NT, (b+1)*NT)
    const int64_t col0 = (int64_t)blockIdx.x * NT;
    const int ncols = (int)((batch - col0) < NT ? (batch - col0) : NT);
    Op op = op0;
    op.init();
    for (int64_t c0 = 0; c0 < rows; c0 += C) {
      __syncthreads();   // previous chunk's stores / logk staging are done
      if (c0 + li < rows_in) {
        for (int c = lc0; c < ncols; c += NT / C) tile[c * P + li] = in[(col0 + c) * rows_in + c0 + li];
      }
      __syncthreads();
      if ((int)threadIdx.x < ncols) {
        T* mine = tile + threadIdx.x * P;
        const int nr = (int)((rows - c0) < C ? (rows - c0) : C);
#pragma unroll 4
        for (int i = 0; i < nr; ++i) mine[i] = seq_step<T, Op>(op, c0 + i, rows, mine[i], logk);
      }
      __syncthreads();
      if (out && c0 + li < rows_out) {
        for (int c = lc0; c < ncols; c += NT / C) out[(col0 + c) * rows_out + c0 + li] = tile[c * P + li];
      }
    }
    if ((int)threadIdx.x < ncols) {
      T l = op.result();
      if (ladj_ps) ladj_ps[col0 + threadIdx.x] = accumulate ? ladj_ps[col0 + threadIdx.x] + l : l;
      acc += (double)l;
    }
  }
  __syncthreads();
  if (partials) block_publish_partial(acc, red, partials);
}

// Wave-private variant (whole columns in LDS): ONE WAVE owns 64 consecutive columns, i.e. one
// CONTIGUOUS run of 64*rows elements of the input and of the output.  The run is moved with
// 16-byte flat accesses (no per-column alignment requirement: K-1 = 63 rows are fine), transposed
// through a [64][P] LDS tile with odd pitch P (the column walk of lane t reads tile[t*P+i]: 32
// distinct banks per 32-lane group), walked by one lane per column in ascending row order (the
// reference's summation order) and written back the same way.  No block barrier anywhere; blocks
// are single waves, so a CU holds ~9 independent tiles at K = 64 and loads, transcendental math and
// stores of different waves overlap.  (The chunked block kernel above did 3 barriers per 32 rows
// with 4-byte global accesses: 32 % of the HBM roofline at C5a.)
template <class T, class Op, int V>
__global__ __launch_bounds__(64) void seq_wave_kernel(Op op0, const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps,
                                                     int rows_in, int rows_out, int P, int64_t batch, int n_logk, int accumulate,
                                                     const BjxFin fin, int64_t ld_in, int64_t ld_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[1];
  T* tile = reinterpret_cast<T*>(smem);
  T* logk = tile + (size_t)64 * P;
  const int lane = threadIdx.x;
  for (int i = lane; i < n_logk; i += 64) logk[i] = d_log(T(n_logk - i));      // log(K-1-i), simplex.jl:35,41
  const int64_t col0 = (int64_t)blockIdx.x * 64;
  const int ncols = (int)((batch - col0) < 64 ? (batch - col0) : 64);
  if (ld_in == rows_in) tile_stage_in<T, V>(tile, in + col0 * rows_in, rows_in, P, ncols, lane);
  else tile_stage_in_ld<T>(tile, in + col0 * ld_in, rows_in, ld_in, P, ncols, lane);
  tile_sync();
  // ---- walk: lane = column
  T lres = T(0);
  if (lane < ncols) {
    Op op = op0;
    op.init();
    T* mine = tile + lane * P;
    const int rows = rows_in > rows_out ? rows_in : rows_out;
    const int mid_end = (Op::HAS_LAST && rows > 1) ? rows - 1 : rows;       // interior rows are [1, mid_end)
    mine[0] = op.first(mine[0], logk);
    int i = 1;
    for (; i + 4 <= mid_end; i += 4) {
      T v[4], lk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = mine[i + j]; lk[j] = Op::USES_LOGK ? logk[i + j] : T(0); }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = op.mid(i + j, v[j], lk[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[i + j] = v[j];
    }
    for (; i < mid_end; ++i) mine[i] = op.mid(i, mine[i], Op::USES_LOGK ? logk[i] : T(0));
    if (Op::HAS_LAST && rows > 1) mine[rows - 1] = op.last(mine[rows - 1]);
    lres = op.result();
    if (ladj_ps) ladj_ps[col0 + lane] = accumulate ? ladj_ps[col0 + lane] + lres : lres;
  }
  tile_sync();
  if (out) {
    if (ld_out == rows_out) tile_stage_out<T, V>(tile, out + col0 * rows_out, rows_out, P, ncols, lane);
    else tile_stage_out_ld<T>(tile, out + col0 * ld_out, rows_out, ld_out, P, ncols, lane);
  }
  block_publish_partial(lane < ncols ? (double)lres : 0.0, red, fin);
}

// Chunked wave walker (round 3): ANY column height at a constant 16.6 KiB of LDS per wave.  seq_wave_kernel keeps whole
// columns in its tile — 64 x (rows | 1) words: 25 KiB at 100 rows (6 waves per CU, 49 % of the HBM peak), 51 KiB at 200 (3 waves,
// 31 %), nothing beyond 255 rows (seq_kernel: 256-column blocks, three barriers per 32 rows, 33 %).  Here one wave still owns 64
// columns and a lane still walks ONE column top to bottom in the reference's order, but the tile holds a CHUNK of C = 64 (Float32)
// / 32 (Float64) rows: load chunk -> walk (the op's running state stays in the lane's registers from chunk to chunk) -> store
// chunk.  Chunk loads are 64 runs of C rows, one per column: with whole-pack column strides 16 lanes move one run as 16-byte
// packs (4 columns per instruction), otherwise the 64 lanes of an instruction move one run of 64 consecutive rows with 4-byte
// accesses — 256 contiguous bytes either way, the full TA rate.  Nine waves per CU, no block barrier, any leading dimension.
template <class T, class Op, int V, bool TABLE>
__global__ __launch_bounds__(64) void seq_chunk_kernel(Op op0, const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps,
                                                       int64_t rows_in, int64_t rows_out, int64_t batch, int n_logk, int accumulate,
                                                       double* __restrict__ partials, int64_t ld_in, int64_t ld_out) {
  constexpr int C = 256 / (int)sizeof(T), P = C + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[1];
  T* tile = reinterpret_cast<T*>(smem);
  T* logk = tile + (size_t)64 * P;
  const int lane = threadIdx.x;
  if (TABLE) for (int i = lane; i < n_logk; i += 64) logk[i] = d_log(T(n_logk - i));      // log(K-1-i), simplex.jl:35,41
  else if (lane == 0) logk[0] = d_log(T(n_logk > 0 ? n_logk : 1));                        // op.first reads entry 0
  const int64_t col0 = (int64_t)blockIdx.x * 64;
  const int ncols = (int)((batch - col0) < 64 ? (batch - col0) : 64);
  const int64_t rows = rows_in > rows_out ? rows_in : rows_out;
  const T* src = in + col0 * ld_in;
  T* dst = out ? out + col0 * ld_out : nullptr;
  Op op = op0;
  op.init();
  // Chunk I/O: an instruction moves CPI runs of C rows (one run per column) as V-element packs, LPR lanes per run; ALL NLD
  // instructions of a chunk are issued before the first result is used (64 VGPRs of data in flight per lane) — issued one by
  // one behind their own waits the chunk loads cost NLD memory round trips and the kernel ran at 15 % of the HBM peak.
  // Raw buffer accesses: the descriptor spans this wave's ncols columns (columns beyond read zeros / drop their stores), the
  // per-lane byte offset is the same register for every instruction of a chunk and the column step is a wave-uniform SGPR
  // offset; rows that do not exist get the out-of-range offset.
  constexpr int LPR = C / V, CPI = 64 / LPR, NLD = 64 / CPI;
  constexpr int kOob = 0x7fffff00;
  const int lr = (lane % LPR) * V, lc = lane / LPR;
  const auto r_in = bjx_make_rsrc(src, (uint32_t)((int64_t)ncols * ld_in * (int64_t)sizeof(T)));
  const auto r_out = bjx_make_rsrc(dst, dst ? (uint32_t)((int64_t)ncols * ld_out * (int64_t)sizeof(T)) : 0u);
  const int step_in = (int)(CPI * ld_in * (int64_t)sizeof(T)), step_out = (int)(CPI * ld_out * (int64_t)sizeof(T));
  Pack<T, V> regs[NLD];
  for (int64_t c0 = 0; c0 < rows; c0 += C) {
    const int nr = (int)((rows - c0) < C ? (rows - c0) : C);
    // ---- load rows [c0, c0 + nr) of the 64 columns (rows >= rows_in do not exist: zero)
    const bool full_in = c0 + lr + V <= rows_in;
    const int vo_in = full_in ? (int)(((int64_t)lc * ld_in + c0 + lr) * (int64_t)sizeof(T)) : kOob;
#pragma unroll
    for (int u = 0; u < NLD; ++u) regs[u] = buf_load_pack_s<T, V>(r_in, vo_in, u * step_in);
    if (V > 1 && !full_in && c0 + lr < rows_in) {          // the ragged last pack of a column (rows_in % V != 0): element by element
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int c = u * CPI + lc;
#pragma unroll
        for (int j = 0; j < V; ++j) regs[u].v[j] = (c < ncols && c0 + lr + j < rows_in) ? src[(int64_t)c * ld_in + c0 + lr + j] : T(0);
      }
    }
    tile_sync();                                          // the previous chunk's stores have read the tile
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int c = u * CPI + lc;
#pragma unroll
      for (int j = 0; j < V; ++j) tile[c * P + lr + j] = regs[u].v[j];
    }
    tile_sync();
    // ---- walk: lane = column, ascending rows (the reference's order)
    if (lane < ncols) {
      T* mine = tile + lane * P;
      int i = 0;
      if (c0 == 0) { mine[0] = op.first(mine[0], logk); i = 1; }
      const int64_t mid_end64 = (Op::HAS_LAST && rows > 1) ? rows - 1 : rows;            // interior rows are [1, mid_end)
      const int mid_end = (int)((mid_end64 - c0) < nr ? (mid_end64 - c0) : nr);
      for (; i + 4 <= mid_end; i += 4) {
        T v[4], lk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = mine[i + j];
          lk[j] = !Op::USES_LOGK ? T(0) : (TABLE ? logk[c0 + i + j] : d_log(T(n_logk - (c0 + i + j))));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = op.mid((int)(c0 + i + j), v[j], lk[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[i + j] = v[j];
      }
      for (; i < mid_end; ++i) mine[i] = op.mid((int)(c0 + i), mine[i], !Op::USES_LOGK ? T(0) : (TABLE ? logk[c0 + i] : d_log(T(n_logk - (c0 + i)))));
      if (Op::HAS_LAST && rows > 1 && c0 + nr == rows) mine[nr - 1] = op.last(mine[nr - 1]);
    }
    tile_sync();
    // ---- store rows [c0, c0 + nr) that exist in the output
    if (dst) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int c = u * CPI + lc;
#pragma unroll
        for (int j = 0; j < V; ++j) regs[u].v[j] = tile[c * P + lr + j];
      }
      const bool full_out = c0 + lr + V <= rows_out;
      const int vo_out = full_out ? (int)(((int64_t)lc * ld_out + c0 + lr) * (int64_t)sizeof(T)) : kOob;
#pragma unroll
      for (int u = 0; u < NLD; ++u) buf_store_pack_s<T, V>(r_out, vo_out, u * step_out, regs[u]);
      if (V > 1 && !full_out && c0 + lr < rows_out) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
          const int c = u * CPI + lc;
#pragma unroll
          for (int j = 0; j < V; ++j) if (c < ncols && c0 + lr + j < rows_out) dst[(int64_t)c * ld_out + c0 + lr + j] = regs[u].v[j];
        }
      }
    }
  }
  T lres = T(0);
  if (lane < ncols) {
    lres = op.result();
    if (ladj_ps) ladj_ps[col0 + lane] = accumulate ? ladj_ps[col0 + lane] + lres : lres;
  }
  if (partials) block_publish_partial(lane < ncols ? (double)lres : 0.0, red, partials);
}

template <class T, class Op>
int launch_seq_chunk(bjx_ctx* ctx, const Op& op, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t rows_in, int64_t rows_out,
                     int64_t batch, int n_logk, uint32_t flags, int64_t ld_in, int64_t ld_out) {
  constexpr int C = 256 / (int)sizeof(T), VW = Vec16<T>::N;
  const bool table = (size_t)n_logk * sizeof(T) <= 40 * 1024;                  // taller: log(K-1-i) on the fly
  const size_t smem = ((size_t)64 * (C + 1) + (table ? (size_t)n_logk : 1)) * sizeof(T);
  BJX_REQUIRE(ctx, (int64_t)64 * (ld_in > ld_out ? ld_in : ld_out) * (int64_t)sizeof(T) < ((int64_t)1 << 31), BJX_ERR_UNSUPPORTED,
              "columns of %lld elements: a wave's 64 columns exceed the 2 GiB range of one buffer descriptor", (long long)(ld_in > ld_out ? ld_in : ld_out));
  const int64_t grid = (batch + 63) / 64;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  // 16-byte packs need every run to start on a 16-byte boundary: whole-pack column strides and aligned bases
  const bool v_ok = bjx_aligned16(in) && (!out || bjx_aligned16(out)) && ld_in % VW == 0 && ld_out % VW == 0;
  {
    BjxProf prof_(ctx);
#define SCK(V_, TB_) hipLaunchKernelGGL((seq_chunk_kernel<T, Op, V_, TB_>), dim3((unsigned)grid), dim3(64), smem, ctx->stream, op, in, out, ladj_ps, rows_in, rows_out, batch, n_logk, accum, partials, ld_in, ld_out)
    if (v_ok) { if (table) SCK(VW, true); else SCK(VW, false); }
    else { if (table) SCK(1, true); else SCK(1, false); }
#undef SCK
  }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

template <class T, class Op>
int launch_seq(bjx_ctx* ctx, const Op& op, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t rows_in, int64_t rows_out,
               int64_t batch, int n_logk, uint32_t flags, int64_t ld_in = 0, int64_t ld_out = 0) {
  if (ld_in == 0) ld_in = rows_in;
  if (ld_out == 0) ld_out = rows_out;
  const bool strided = ld_in != rows_in || ld_out != rows_out;
  if (batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  // wave-private tiles when a [64][rows] tile (+ log table) fits in 64 KiB of LDS
  {
    const int64_t rows = rows_in > rows_out ? rows_in : rows_out;
    const int64_t P = rows | 1;
    const size_t smem_w = ((size_t)64 * P + (size_t)n_logk) * sizeof(T);
    static const int use_wave = getenv("BJX_SEQ_WAVE") ? atoi(getenv("BJX_SEQ_WAVE")) : 1;
    // whole-column tiles up to `chunk_min` bytes (tuning switch for the same-box A/B): beyond, the chunked walker keeps 9 waves per CU
    // (same-box A/B, profiles/r03_tall_columns.md: the chunk loads are runs of 256 bytes that straddle cache lines, so part of
    //  every line is fetched twice — the chunked walker wins only where the whole-column tile leaves < 3 waves per CU:
    //  K = 100: 30 % against 49 %, 200: 29-34 % against 27-31 %, 256: 39-68 % against 35-61 %, 1000: 34-41 % (no tile at all))
    static const long chunk_min = 50 * 1024;
    if (smem_w > (size_t)chunk_min && rows_in >= 1 && rows_out >= 1)
      return launch_seq_chunk<T, Op>(ctx, op, in, out, ladj_ps, ladj_sum, rows_in, rows_out, batch, n_logk, flags, ld_in, ld_out);
    if (use_wave && smem_w <= 64 * 1024 && rows_in >= 1 && rows_out >= 1) {   // larger columns: the chunked block kernel below
      const int64_t grid = (batch + 63) / 64;
      BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
      BjxFin fin;
      bool second = false;
      { int rc = bjx_make_fin(ctx, grid, ladj_sum, 0.0, 0, flags, &fin, &second); if (rc) return rc; }
      // single-wave blocks: the in-kernel finalize would make EVERY wave wait for its own tile stores
      // (vmcnt(0) before the arrival atomic) -> 2x slower (measured 0.27 vs 0.13 ms at C5a); use the
      // two-pass finalize here.
      if (fin.counter) { fin.counter = nullptr; second = true; }
      constexpr int VW = Vec16<T>::N;
      const bool v_ok = bjx_aligned16(in) && (!out || bjx_aligned16(out)) && !(ld_in != rows_in && ld_out != rows_out);
      const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
      { BjxProf prof_(ctx);
      if (v_ok) hipLaunchKernelGGL((seq_wave_kernel<T, Op, VW>), dim3((unsigned)grid), dim3(64), smem_w, ctx->stream, op, in, out, ladj_ps, (int)rows_in, (int)rows_out, (int)P, batch, n_logk, accum, fin, ld_in, ld_out);
      else hipLaunchKernelGGL((seq_wave_kernel<T, Op, 1>), dim3((unsigned)grid), dim3(64), smem_w, ctx->stream, op, in, out, ladj_ps, (int)rows_in, (int)rows_out, (int)P, batch, n_logk, accum, fin, ld_in, ld_out); }
      BJX_CHECK_LAUNCH(ctx);
      if (second) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
      return BJX_OK;
    }
  }
  BJX_REQUIRE(ctx, !strided, BJX_ERR_UNSUPPORTED, "columns of %lld rows with a leading dimension exceed the LDS tile kernel", (long long)(rows_in > rows_out ? rows_in : rows_out));
  constexpr int C = SeqCfg<T>::C, NT = SeqCfg<T>::NT;
  const size_t smem = 32 + ((size_t)NT * (C + 1) + (size_t)n_logk) * sizeof(T);
  BJX_REQUIRE(ctx, smem <= 64 * 1024, BJX_ERR_UNSUPPORTED, "simplex: K = %d too large for the LDS log-table", n_logk + 1);
  const int64_t grid = (batch + NT - 1) / NT;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  { BjxProf prof_(ctx);
  hipLaunchKernelGGL((seq_kernel<T, Op>), dim3((unsigned)grid), dim3(NT), smem, ctx->stream, op, in, out, ladj_ps, rows_in, rows_out, batch, n_logk,
                     (flags & BJX_ACCUMULATE) ? 1 : 0, partials); }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}

// ------------------------------------------------------------------ VecCholesky (wave per sample)
template <class T> __device__ __forceinline__ T wave_incl_scan(T v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
// packed strict-upper index e (column-major, 0-based) -> column c (1..K-1) and row i0 (0..c-1)
__device__ __forceinline__ void triu1_decode(int64_t e, int& c, int& i0) {
  int cc = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
  while ((int64_t)cc * (cc - 1) / 2 > e) --cc;
  while ((int64_t)(cc + 1) * cc / 2 <= e) ++cc;
  c = cc;
  i0 = (int)(e - (int64_t)cc * (cc - 1) / 2);
}

// Inclusive segmented PREFIX sum inside this lane's column (ascending order), including the part
// carried over from earlier 64-entry steps.  `head` marks the first entry of a column.
template <class T> __device__ __forceinline__ T seg_prefix_incl(T v, bool head, T carry) {
  const int lane = threadIdx.x & 63;
  int f = head ? 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T vo = __shfl_up(v, d, 64);
    const int fo = __shfl_up(f, d, 64);
    if (lane >= d && !f) { v += vo; f |= fo; }
  }
  return v + (f ? T(0) : carry);
}

// Wave-wide inclusive prefix sum with DPP (row_shr 1,2,4,8 inside the 16-lane rows, then the GFX9
// row_bcast:15 / row_bcast:31 steps): 6 fused v_add_dpp instead of 6 x (ds_bpermute + select + add).
__device__ __forceinline__ float wave_incl_scan_dpp(float v) {
#define BJX_DPP_ADD(ctrl, rmask)                                                                         \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, true))
  BJX_DPP_ADD(0x111, 0xF);   // row_shr:1
  BJX_DPP_ADD(0x112, 0xF);   // row_shr:2
  BJX_DPP_ADD(0x114, 0xF);   // row_shr:4
  BJX_DPP_ADD(0x118, 0xF);   // row_shr:8
  BJX_DPP_ADD(0x142, 0xA);   // row_bcast:15 -> rows 1 and 3
  BJX_DPP_ADD(0x143, 0xC);   // row_bcast:31 -> rows 2 and 3
#undef BJX_DPP_ADD
  return v;
}
__device__ __forceinline__ double wave_incl_scan_dpp(double v) { return wave_incl_scan(v); }

// corr.jl:370-399 (_inv_link_chol_lkj, vector form) and :485-501 (_logabsdetjac_inv_chol).
// VALU-bound with the first version (340 VALU / entry, PMC in profiles/r01_pmc_notes.md); now:
//  * (column, row) of a lane's entry is advanced incrementally (+64 entries per step) instead of an
//    isqrt-style decode per entry;
//  * tanh and logcosh share one exp(-2|y|) (tanh = (1-t)/(1+t), logcosh = |y| + log1p(t) - log 2);
//    Float32 uses the hardware exp/log/rcp units;
//  * the per-column running Σ logcosh is a DPP wave scan minus the scan value just before the
//    column head (all terms are >= 0 and O(0.1), so the difference loses < 1e-6 absolute).
template <class T, bool WRITE_W>
__global__ __launch_bounds__(256) void chol_inv_kernel(const T* y, T* W, T* ladj_ps, int64_t K, int64_t batch, int lower,
                                                       int accumulate, double* partials) {
  using F = Fast<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* diag = reinterpret_cast<T*>(smem + 32) + (size_t)wave * K;
  const int64_t nv = K * (K - 1) / 2;
  double acc = 0.0;
  int c_lane0, i0_lane0;
  triu1_decode(lane, c_lane0, i0_lane0);
  const int64_t s = (int64_t)blockIdx.x * 4 + wave;
  if (s < batch) {
    const T* ys = y + s * nv;
    T* Ws = W + s * K * K;
    T carry = T(0), lj = T(0);
    int c = c_lane0, i0 = i0_lane0;      // entry e = e0 + lane sits in column c (1..K-1), row i0 (0..c-1)
    for (int64_t e0 = 0; e0 < nv; e0 += 64) {
      const int64_t e = e0 + lane;
      const bool valid = e < nv;
      const T yv = valid ? ys[e] : T(0);
      const T ay = d_abs(yv);
      const T t = F::exp(T(-2) * ay);
      const T lc = valid ? ay + F::log1p(t) - Num<T>::log2 : T(0);          // LogExpFunctions.logcosh
      // inclusive Σ logcosh over my column: wave prefix minus the prefix just before the column head
      const T S = wave_incl_scan_dpp(lc);
      const int head = lane - i0;                                             // lane of the column's first entry (< 0: earlier step)
      const T base = __shfl(S, head > 0 ? head - 1 : 0, 64);
      const T incl = S - (head > 0 ? base : T(0)) + (head < 0 ? carry : T(0));
      const bool last = valid && (i0 == c - 1);
      if (valid) {
        lj += last ? T(-2) * incl : -incl;        // logJ += log_remainder (each entry) + once more per column (:385-389)
        if (WRITE_W) {
          const T th = (T(1) - t) * F::rcp(T(1) + t);                         // tanh|y|
          const T wv = (yv < T(0) ? -th : th) * F::exp(-(incl - lc));         // z * exp(log_remainder_before) (:383)
          if (!lower) Ws[(int64_t)c * K + i0] = wv; else Ws[(int64_t)i0 * K + c] = wv;
          if (last) diag[c] = F::exp(-incl);       // W[j,j] = exp(log_remainder) (:390)
        }
      }
      const T incl63 = __shfl(incl, 63, 64);
      const int last63 = __shfl((int)last, 63, 64);
      carry = last63 ? T(0) : incl63;
      i0 += 64;
      while (i0 >= c) { i0 -= c; ++c; }
    }
    if (WRITE_W) {
      if (lane == 0) diag[0] = T(1);
      // diagonal + zero fill of the other triangle (:391-395); column-major, rows c..K-1 of column c
      for (int cc = 0; cc < K; ++cc) {
        for (int r = cc + lane; r < K; r += 64) {
          const T v = (r == cc) ? diag[cc] : T(0);
          if (!lower) Ws[(int64_t)cc * K + r] = v; else Ws[(int64_t)r * K + cc] = v;
        }
      }
    }
    lj = group_sum<64>(lj);
    if (lane == 0) {
      if (ladj_ps) ladj_ps[s] = accumulate ? ladj_ps[s] + lj : lj;
      acc += (double)lj;
    }
  }
  __syncthreads();
  if (partials) block_publish_partial(acc, red, partials);
}

// ------------------------------------------------------------------ VecCholesky inverse, chunk kernel
// The kernel above handles one packed entry per lane per step: 4-byte loads, a 4-byte scattered
// global store per entry plus K partial-row stores for the diagonal and the zero triangle (96 store
// instructions for a 16 KiB sample at K = 64) — 33 % of the HBM roofline, bound by the store path.
// Here ONE WAVE still owns one sample, but
//  * LANE L OWNS THE CONTIGUOUS CHUNK [L*CH, (L+1)*CH) of the packed vector (CH = 32 entries = one
//    128-byte line at K = 64, read with 16-byte loads) and walks it serially with 100 % lane
//    utilisation; the only cross-lane step is ONE segmented scan per sample that hands each lane the
//    Σ logcosh of the part of its first column that lives in earlier lanes (a column spans <= 3 lanes);
//  * the K x K factor is assembled in a zero-initialised LDS tile (scattered 4-byte LDS writes) and
//    leaves as 16 fully coalesced 16-byte stores per lane (the dense column-major W of corr.jl:391-395
//    IS the tile);
//  * pass 1 evaluates t = exp(-2|y|) and logcosh = |y| + log(1+t) - log 2 once per entry, pass 2
//    needs one more exp per entry: w = tanh(y)·exp(-Σ_before logcosh) (:383), tanh = ±(1-t)/(1+t).
// corr.jl:370-399 (_inv_link_chol_lkj) and :485-501 (_logabsdetjac_inv_chol when W is not wanted).
constexpr int CHOL_WPB = 2;   // waves (= samples) per block: 2 x ~17 KiB tiles at K = 64, 4 blocks per CU
// Column bookkeeping without divisions or per-entry branches: `d` = entries left in the current
// column (counts down), `len` = its length (= column index c), wrap = "this entry is the last of its
// column" = "the next entry is a column head".  Pass 1 shifts the wrap bits into a per-lane mask so
// pass 2 only tests a bit.  Entries past the end of the vector (only in the last lanes; loaded as 0)
// form phantom columns c >= K whose logcosh is EXACTLY 0 (t = 1, log2(2) = 1), so they add nothing to
// the log-det and need no predicate; their tile writes land in the padding columns behind the K x K
// factor ('U') or are redirected to a dummy word ('L').
template <class T, int V, int CHV, bool WRITE_W, bool LOWER>
__global__ __launch_bounds__(64 * CHOL_WPB) void chol_inv_chunk_kernel(const T* __restrict__ y, T* __restrict__ W, T* __restrict__ ladj_ps, int K,
                                                             int tile_words, int64_t batch, int accumulate, double* partials) {
  using F = Fast<T>;
  constexpr int CH = CHV * V;                      // entries per lane
  static_assert(CH <= 32, "wrap mask is 32 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[CHOL_WPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* tile = reinterpret_cast<T*>(smem) + (size_t)wave * tile_words;   // [K + pad columns][K] + 1 dummy word
  const int dummy = tile_words - 1;
  const int nv = K * (K - 1) / 2;
  const int64_t s = (int64_t)blockIdx.x * CHOL_WPB + wave;
  double acc = 0.0;
  if (s < batch) {
    const T* ys = y + s * nv;
    const int e0 = lane * CH;
    // ---- the packed vector arrives with COALESCED flat 16-byte loads (pack p = lane + 64 i) and is
    //      re-dealt through LDS so that lane L holds the contiguous chunk [L*CH, (L+1)*CH).  (Loading the
    //      chunks directly makes every wave instruction touch 64 different 128-byte lines: the kernel then
    //      runs at the same 0.29 ms whether or not W is written.)  Chunk pitch CH + V words: the 16-byte
    //      LDS accesses of both phases are bank-conflict free.  Without WRITE_W there is no LDS tile and the
    //      chunk is loaded directly.
    Pack<T, V> yp[CHV];
    if (WRITE_W) {
      constexpr int S = CH + V;
#pragma unroll
      for (int i = 0; i < CHV; ++i) {
        const int e = (lane + 64 * i) * V;                              // flat element index of this pack
        Pack<T, V> t;
        if (V > 1) {
          if (e + V <= nv) t = load_pack<T, V, true>(ys + e);           // nv % V == 0 on this path: whole packs only
          else {
#pragma unroll
            for (int j = 0; j < V; ++j) t.v[j] = T(0);
          }
        } else {
          t.v[0] = (e < nv) ? ys[e] : T(0);
        }
        yp[i] = t;
      }
#pragma unroll
      for (int i = 0; i < CHV; ++i) {
        const int e = (lane + 64 * i) * V;
        T* dst = tile + (e / CH) * S + e % CH;
        if (V > 1) *reinterpret_cast<typename Vec16<T>::type*>(dst) = *reinterpret_cast<const typename Vec16<T>::type*>(&yp[i]);
        else dst[0] = yp[i].v[0];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < CHV; ++q) {
        const T* src = tile + lane * S + q * V;
        if (V > 1) *reinterpret_cast<typename Vec16<T>::type*>(&yp[q]) = *reinterpret_cast<const typename Vec16<T>::type*>(src);
        else yp[q].v[0] = src[0];
      }
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int q = 0; q < CHV; ++q) {
        const int e = e0 + q * V;
        if (V > 1) {
          if (e + V <= nv) yp[q] = load_pack<T, V, false>(ys + e);
          else {
#pragma unroll
            for (int j = 0; j < V; ++j) yp[q].v[j] = T(0);
          }
        } else {
          yp[q].v[0] = (e < nv) ? ys[e] : T(0);
        }
      }
    }
    if (WRITE_W) {
      constexpr int ZV = 16 / sizeof(T);
      const int nz = K * K / ZV;
      typename Vec16<T>::type zero = {};
      for (int i = lane; i < nz; i += 64) reinterpret_cast<typename Vec16<T>::type*>(tile)[i] = zero;
      for (int i = nz * ZV + lane; i < K * K; i += 64) tile[i] = T(0);
    }
    int c0, i00;
    triu1_decode(e0, c0, i00);                     // column (1-based count = its length) and row of my first entry
    // ---- pass 1: t, logcosh per entry; wrap mask; Σ logcosh of my last (open) column segment
    T lc[CH], tt[CH];
    unsigned wmask = 0;
    T tail = T(0);
    bool has_head = (i00 == 0);
    {
      int d = c0 - i00, len = c0;
      bool prev_wrap = false;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const T ay = d_abs(yp[k / V].v[k % V]);
        const T t = F::exp(T(-2) * ay);
        lc[k] = F::log2(T(1) + t) * Num<T>::log2 + (ay - Num<T>::log2);    // LogExpFunctions.logcosh (exactly 0 at y = 0)
        tt[k] = t;
        tail = (prev_wrap ? T(0) : tail) + lc[k];
        has_head |= prev_wrap;
        d -= 1;
        const bool wrap = (d == 0);
        len += wrap ? 1 : 0;
        d = wrap ? len : d;
        wmask = (wmask << 1) | (wrap ? 1u : 0u);
        prev_wrap = wrap;
      }
    }
    // ---- carry-in: Σ logcosh of my first column's entries held by earlier lanes (one segmented scan per sample)
    T carry;
    {
      const T incl = seg_prefix_incl<T>(tail, has_head, T(0));
      const T up = __shfl_up(incl, 1, 64);
      carry = lane == 0 ? T(0) : up;
    }
    __builtin_amdgcn_wave_barrier();   // tile zeroing (same wave, in-order LDS queue) precedes the scatter
    // ---- pass 2
    T lj = T(0);
    {
      T run = (i00 == 0) ? T(0) : carry;
      int len = c0;
      int addr = LOWER ? i00 * K + c0 : c0 * K + i00;       // word index of my first entry in the tile
      bool prev_wrap = false;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const bool wrap = (wmask >> (CH - 1 - k)) & 1u;      // last entry of its column
        run = prev_wrap ? T(0) : run;
        const T excl = run;                                  // log_remainder before this entry = -excl
        run += lc[k];
        lj -= run;                                           // logJ += log_remainder after the entry (:385-387)
        lj -= wrap ? run : T(0);                             // ... and once more at the end of a column (:389)
        if (WRITE_W) {
          const T yv = yp[k / V].v[k % V];
          const T t = tt[k];
          const T th = (T(1) - t) * F::rcp(T(1) + t);        // tanh|y|
          const T wv = d_copysign(th, yv) * F::exp(-excl);   // z * exp(log_remainder) (:383)
          int wa = addr, da;
          if (LOWER) {
            const bool real = len < K;
            wa = real ? addr : dummy;
            da = (wrap && real) ? addr + K : dummy;          // (c-1)*K + c  ->  c*K + c
          } else {
            da = wrap ? addr + 1 : dummy;                    // c*K + c-1    ->  c*K + c
          }
          tile[wa] = wv;
          tile[da] = run;                                    // Σ logcosh of the whole column, exponentiated below
          if (LOWER) addr = wrap ? len + 1 : addr + K;
          else addr += wrap ? K + 1 - len : 1;
          len += wrap ? 1 : 0;
        }
        prev_wrap = wrap;
      }
    }
    lj = group_sum<64>(lj);
    if (lane == 0) {
      if (ladj_ps) ladj_ps[s] = accumulate ? ladj_ps[s] + lj : lj;
      acc = (double)lj;
    }
    if (WRITE_W) {
      __builtin_amdgcn_wave_barrier();
      // diagonal: W[j,j] = exp(log_remainder at the end of column j) (:390); W[1,1] = 1 (:376)
      for (int c = lane; c < K; c += 64) {
        const T r = tile[c * K + c];
        tile[c * K + c] = c == 0 ? T(1) : F::exp(-r);
      }
      __builtin_amdgcn_wave_barrier();
      T* Ws = W + s * (int64_t)K * K;
      constexpr int ZV = 16 / sizeof(T);
      const int nz = K * K / ZV;
      if (bjx_aligned16_dev(Ws)) {
        for (int i = lane; i < nz; i += 64)
          __builtin_nontemporal_store(reinterpret_cast<const typename Vec16<T>::type*>(tile)[i], reinterpret_cast<typename Vec16<T>::type*>(Ws) + i);
        for (int i = nz * ZV + lane; i < K * K; i += 64) Ws[i] = tile[i];
      } else {
        for (int i = lane; i < K * K; i += 64) Ws[i] = tile[i];
      }
    }
  }
  __syncthreads();
  if (partials) block_publish_partial(acc, red, partials);
}

// ------------------------------------------------------------------ VecCholesky forward link, chunk kernel
// corr.jl:314-337 (_link_chol_lkj_from_upper / _from_lower) with the same decomposition as the
// inverse chunk kernel: the K x K factor is staged into LDS with coalesced 16-byte loads (the tile IS
// the input layout), lane L owns packed entries [L*CH, (L+1)*CH) of the OUTPUT vector and gathers
// its W[i,j] from the tile.  remainder_sq = W[j,j]² + Σ_{k>i} W[k,j]² is a true SUFFIX sum inside a
// column (a prefix difference would cancel catastrophically: deep entries are ~1e-7 next to O(1)
// neighbours): each lane walks its chunk in DESCENDING order, and one reversed segmented scan per
// sample hands it the part of its last column that lives in later lanes.  asinh(w/√rem) =
// log((|w| + √(w²+rem))/√rem); the first row's atanh(w) (:322) is the same expression with
// rem := 1 - w² (atanh w = asinh(w/√(1-w²))), so every entry costs one rsq, one sqrt and one log.
// The log-det -_logabsdetjac_inv_chol(y) (:235-237) reuses the inverse kernel's ascending machinery.
template <class T> __device__ __forceinline__ T seg_suffix_excl(T v, bool stop, T /*unused*/) {
  // R(L) = v(L) + (stop(L) ? 0 : R(L+1)); returns R(L+1) (0 for lane 63)
  const int lane = threadIdx.x & 63;
  int f = stop ? 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T vo = __shfl_down(v, d, 64);
    const int fo = __shfl_down(f, d, 64);
    if (lane + d < 64 && !f) { v += vo; f |= fo; }
  }
  const T nxt = __shfl_down(v, 1, 64);
  return lane == 63 ? T(0) : nxt;
}

template <class T, int V, int CHV, bool LOWER, bool LADJ>
__global__ __launch_bounds__(64 * CHOL_WPB) void chol_fwd_chunk_kernel(const T* __restrict__ W, T* __restrict__ y, T* __restrict__ ladj_ps, int K,
                                                             int tile_words, int64_t batch, int accumulate, double* partials) {
  using F = Fast<T>;
  constexpr int CH = CHV * V;
  static_assert(CH <= 32, "wrap mask is 32 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[CHOL_WPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* tile = reinterpret_cast<T*>(smem) + (size_t)wave * tile_words;
  const int nv = K * (K - 1) / 2;
  const int64_t s = (int64_t)blockIdx.x * CHOL_WPB + wave;
  double acc = 0.0;
  if (s < batch) {
    // ---- stage W into the tile (flat copy)
    {
      const T* Ws = W + s * (int64_t)K * K;
      constexpr int ZV = 16 / sizeof(T);
      const int nz = K * K / ZV;
      if (bjx_aligned16_dev(Ws)) {
        constexpr int SU = 16;
        for (int i0 = lane; i0 < nz; i0 += 64 * SU) {
          typename Vec16<T>::type t[SU];
#pragma unroll
          for (int u = 0; u < SU; ++u) if (i0 + u * 64 < nz) t[u] = __builtin_nontemporal_load(reinterpret_cast<const typename Vec16<T>::type*>(Ws) + i0 + u * 64);
#pragma unroll
          for (int u = 0; u < SU; ++u) if (i0 + u * 64 < nz) reinterpret_cast<typename Vec16<T>::type*>(tile)[i0 + u * 64] = t[u];
        }
        for (int i = nz * ZV + lane; i < K * K; i += 64) tile[i] = Ws[i];
      } else {
        for (int i = lane; i < K * K; i += 64) tile[i] = Ws[i];
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int e0 = lane * CH;
    int c0, i00;
    triu1_decode(e0, c0, i00);
    // ---- ascending pass: gather w (and the diagonal at column ends), wrap mask, my contribution to
    //      earlier lanes' suffixes: pre = Σ w² up to and including my first column end (+ its diagonal²)
    T wv[CH], aux[CH];     // aux: diagonal (pass 1/2) then logcosh (pass 3/4)
    unsigned wmask = 0;
    T pre = T(0);
    bool seen_wrap = false;
    int d = c0 - i00;                                                  // entries left in my current column, this one included
    {
      int len = c0;
      int addr = LOWER ? i00 * K + c0 : c0 * K + i00;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int ra = addr < tile_words ? addr : 0;                   // phantom entries: any in-bounds word
        const int da = LOWER ? addr + K : addr + 1;
        const T w = tile[ra];
        const T dg = tile[da < tile_words ? da : 0];
        wv[k] = w; aux[k] = dg;
        d -= 1;
        const bool wrap = (d == 0);
        pre += seen_wrap ? T(0) : w * w + (wrap ? dg * dg : T(0));
        seen_wrap |= wrap;
        if (LOWER) addr = wrap ? len + 1 : addr + K;
        else addr += wrap ? K + 1 - len : 1;
        len += wrap ? 1 : 0;
        d = wrap ? len : d;
        wmask = (wmask << 1) | (wrap ? 1u : 0u);
      }
    }
    const T carry_sfx = seg_suffix_excl<T>(pre, seen_wrap, T(0));
    // ---- descending pass: suffix of w² inside the column, y, and the log-det.
    //      -_logabsdetjac_inv_chol(y) (corr.jl:239-250) adds, per column, the running sums of logcosh(y) once per
    //      entry and once more at the column end: entry k of a column enters (entries from k to the column end) + 1
    //      times.  With z = w / sqrt(rem):  logcosh(asinh z) = log sqrt(1 + z²) = log(sqrt(w² + rem) · rsqrt(rem)),
    //      both factors already at hand — one v_log per entry instead of exp + log, and no scan.
    T lsum = T(0);
    {
      int dd = d;
      T sfx = carry_sfx;
#pragma unroll
      for (int k = CH - 1; k >= 0; --k) {
        const bool wrap = (wmask >> (CH - 1 - k)) & 1u;
        const bool head = k == 0 ? (i00 == 0) : ((wmask >> (CH - k)) & 1u);
        const T w = wv[k];
        const T w2 = w * w;
        sfx = wrap ? aux[k] * aux[k] : sfx;                            // remainder_sq starts at W[j,j]² (:318)
        const T rem = head ? T(1) - w2 : sfx;                          // first row: atanh(w) (:322)
        const T rs = F::rsqrt(rem);
        const T sq = F::sqrt(w2 + rem);
        const T q = (d_abs(w) + sq) * rs;
        wv[k] = d_copysign(F::log(q), w);                      // asinh(w / sqrt(remainder_sq)) (:327-329)
        sfx += w2;
        if (LADJ) {
          dd = wrap ? 1 : dd + 1;
          const T lc = F::log2(sq * rs) * T(dd + 1);
          lsum += (e0 + k < nv) ? lc : T(0);
        }
      }
    }
    // ---- store y: my chunk goes to LDS (pitch CH + V: conflict-free 16-byte accesses; the W tile is dead
    //      by now) and leaves as COALESCED flat 16-byte stores (pack p = lane + 64 i).  Storing the chunks
    //      directly makes every wave instruction write 64 partial lines: 14 % of the HBM roofline.
    {
      constexpr int S = CH + V;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < CHV; ++q) {
        T* dst = tile + lane * S + q * V;
        if (V > 1) {
          Pack<T, V> p;
#pragma unroll
          for (int j = 0; j < V; ++j) p.v[j] = wv[q * V + j];
          *reinterpret_cast<typename Vec16<T>::type*>(dst) = *reinterpret_cast<const typename Vec16<T>::type*>(&p);
        } else dst[0] = wv[q];
      }
      __builtin_amdgcn_wave_barrier();
      T* ys = y + s * nv;
#pragma unroll
      for (int i = 0; i < CHV; ++i) {
        const int e = (lane + 64 * i) * V;
        const T* src = tile + (e / CH) * S + e % CH;
        if (V > 1) {
          if (e + V <= nv) {
            Pack<T, V> p;
            *reinterpret_cast<typename Vec16<T>::type*>(&p) = *reinterpret_cast<const typename Vec16<T>::type*>(src);
            store_pack<T, V, true>(ys + e, p);
          }
        } else if (e < nv) ys[e] = src[0];
      }
    }
    if (LADJ) {
      T lj = lsum * Num<T>::log2;
      lj = group_sum<64>(lj);
      if (lane == 0) {
        if (ladj_ps) ladj_ps[s] = accumulate ? ladj_ps[s] + lj : lj;
        acc = (double)lj;
      }
    }
  }
  __syncthreads();
  if (partials) block_publish_partial(acc, red, partials);
}

// ------------------------------------------------------------------ pullback of _inv_link_chol_lkj
// corr.jl:402-451 (_inv_link_chol_lkj_rrule, the rule behind ext/BijectorsChainRulesCoreExt.jl:311-320):
// given y, ΔW (K x K, dense) and ΔlogJ,
//   for j = K..2:  Δlr = W[j,j] ΔW[j,j] + 2 ΔlogJ
//     for i = j-1..1:  Δy[idx] = (1/z - z) W[i,j] ΔW[i,j] - z Δlr ;  Δlr += ΔlogJ + W[i,j] ΔW[i,j]
// with z = tanh y, W[i,j] = z exp(lr_before), W[j,j] = exp(lr_end).  Same chunk decomposition as the
// forward kernels: pass 1 (ascending) rebuilds lr_before from Σ logcosh (one forward segmented scan),
// e = exp(lr_before) ΔW[i,j] is gathered from the staged ΔW tile, and Δlr — a SUFFIX sum inside the
// column seeded by the diagonal term — comes from a descending walk plus one reversed segmented scan.
// (1/z - z) W[i,j] is evaluated as (1 - z²) exp(lr_before): no 0·inf at y = 0.
template <class T, int V, int CHV, bool LOWER>
__global__ __launch_bounds__(64 * CHOL_WPB) void chol_inv_vjp_kernel(const T* __restrict__ y, const T* __restrict__ Wbar, const T* __restrict__ lbar,
                                                           T* __restrict__ ybar, int K, int tile_words, int64_t batch) {
  using F = Fast<T>;
  constexpr int CH = CHV * V;
  static_assert(CH <= 32, "wrap mask is 32 bits");
  constexpr int S = CH + V;                          // chunk pitch of the packed-vector staging
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* tile = reinterpret_cast<T*>(smem) + (size_t)wave * tile_words;
  const int nv = K * (K - 1) / 2;
  const int64_t s = (int64_t)blockIdx.x * CHOL_WPB + wave;
  if (s >= batch) return;
  const T dl = lbar ? lbar[s] : T(0);
  const int e0 = lane * CH;
  // ---- y: coalesced flat loads -> LDS -> my contiguous chunk
  Pack<T, V> yp[CHV];
  {
    const T* ys = y + s * nv;
#pragma unroll
    for (int i = 0; i < CHV; ++i) {
      const int e = (lane + 64 * i) * V;
      Pack<T, V> t;
      if (V > 1) {
        if (e + V <= nv) t = load_pack<T, V, true>(ys + e);
        else {
#pragma unroll
          for (int j = 0; j < V; ++j) t.v[j] = T(0);
        }
      } else t.v[0] = (e < nv) ? ys[e] : T(0);
      T* dst = tile + (e / CH) * S + e % CH;
      if (V > 1) *reinterpret_cast<typename Vec16<T>::type*>(dst) = *reinterpret_cast<const typename Vec16<T>::type*>(&t);
      else dst[0] = t.v[0];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < CHV; ++q) {
      const T* src = tile + lane * S + q * V;
      if (V > 1) *reinterpret_cast<typename Vec16<T>::type*>(&yp[q]) = *reinterpret_cast<const typename Vec16<T>::type*>(src);
      else yp[q].v[0] = src[0];
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- ΔW -> tile (flat copy: the tile is ΔW's own layout)
  {
    const T* Ws = Wbar + s * (int64_t)K * K;
    constexpr int ZV = 16 / sizeof(T);
    const int nz = K * K / ZV;
    if (bjx_aligned16_dev(Ws)) {
      constexpr int SU = 16;
      for (int i0 = lane; i0 < nz; i0 += 64 * SU) {
        typename Vec16<T>::type t[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) if (i0 + u * 64 < nz) t[u] = __builtin_nontemporal_load(reinterpret_cast<const typename Vec16<T>::type*>(Ws) + i0 + u * 64);
#pragma unroll
        for (int u = 0; u < SU; ++u) if (i0 + u * 64 < nz) reinterpret_cast<typename Vec16<T>::type*>(tile)[i0 + u * 64] = t[u];
      }
      for (int i = nz * ZV + lane; i < K * K; i += 64) tile[i] = Ws[i];
    } else {
      for (int i = lane; i < K * K; i += 64) tile[i] = Ws[i];
    }
  }
  __builtin_amdgcn_wave_barrier();
  int c0, i00;
  triu1_decode(e0, c0, i00);
  // ---- pass 1 (ascending): signed t, logcosh, wrap mask, tail of my open column; gather ΔW[i,j] and ΔW[j,j]
  T ts[CH], ed[CH], aux[CH];      // ts = copysign(exp(-2|y|), y); ed = ΔW[i,j] (then exp(lr_before) ΔW[i,j]); aux = logcosh, then D at column ends
  unsigned wmask = 0;
  T tail = T(0);
  bool has_head = (i00 == 0);
  {
    int d = c0 - i00, len = c0;
    int addr = LOWER ? i00 * K + c0 : c0 * K + i00;
    bool prev_wrap = false;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const T yv = yp[k / V].v[k % V];
      const T ay = d_abs(yv);
      const T t = F::exp(T(-2) * ay);
      aux[k] = F::log2(T(1) + t) * Num<T>::log2 + (ay - Num<T>::log2);      // logcosh (exactly 0 for the zero padding)
      ts[k] = d_copysign(t, yv);
      tail = (prev_wrap ? T(0) : tail) + aux[k];
      has_head |= prev_wrap;
      const bool real = len < K;                                             // phantom columns read word 0
      ed[k] = tile[real ? addr : 0];
      d -= 1;
      const bool wrap = (d == 0);
      if (LOWER) addr = wrap ? len + 1 : addr + K;
      else addr += wrap ? K + 1 - len : 1;
      len += wrap ? 1 : 0;
      d = wrap ? len : d;
      wmask = (wmask << 1) | (wrap ? 1u : 0u);
      prev_wrap = wrap;
    }
  }
  T carry;
  {
    const T incl = seg_prefix_incl<T>(tail, has_head, T(0));
    const T up = __shfl_up(incl, 1, 64);
    carry = lane == 0 ? T(0) : up;
  }
  // ---- pass 2 (ascending): e = exp(lr_before) ΔW[i,j]; s_k = ΔlogJ + z e; D = exp(lr_end) ΔW[j,j] + 2 ΔlogJ at column ends;
  //      my contribution to earlier lanes' suffixes
  T pre = T(0);
  bool seen_wrap = false;
  {
    T run = (i00 == 0) ? T(0) : carry;
    T Eb = F::exp(-run);                               // exp(lr_before) of the current entry; one exp per entry below
    int len = c0;
    bool prev_wrap = false;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const bool wrap = (wmask >> (CH - 1 - k)) & 1u;
      run = prev_wrap ? T(0) : run;
      Eb = prev_wrap ? T(1) : Eb;
      const T e = Eb * ed[k];
      run += aux[k];
      const T Ea = F::exp(-run);                       // exp(lr_after): W[j,j] at a column end, exp(lr_before) of the next entry
      const T t = d_abs(ts[k]);
      const T z = d_copysign((T(1) - t) * F::rcp(T(1) + t), ts[k]);
      const T sk = dl + z * e;
      const T dgbar = tile[len < K ? len * K + len : 0];
      const T D = wrap ? Ea * dgbar + 2 * dl : T(0);   // W[j,j] ΔW[j,j] + 2 ΔlogJ
      ed[k] = e;
      aux[k] = D;
      pre += seen_wrap ? T(0) : sk + D;
      seen_wrap |= wrap;
      len += wrap ? 1 : 0;
      Eb = Ea;
      prev_wrap = wrap;
    }
  }
  const T carry_sfx = seg_suffix_excl<T>(pre, seen_wrap, T(0));
  // ---- pass 3 (descending): Δlr and Δy
  {
    T sfx = carry_sfx;
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
      const bool wrap = (wmask >> (CH - 1 - k)) & 1u;
      sfx = wrap ? aux[k] : sfx;
      const T t = d_abs(ts[k]);
      const T z = d_copysign((T(1) - t) * F::rcp(T(1) + t), ts[k]);
      const T e = ed[k];
      ed[k] = (T(1) - z * z) * e - z * sfx;                                  // Δy
      sfx += dl + z * e;
    }
  }
  // ---- Δy: chunk -> LDS -> coalesced flat stores
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < CHV; ++q) {
    T* dst = tile + lane * S + q * V;
    if (V > 1) {
      Pack<T, V> p;
#pragma unroll
      for (int j = 0; j < V; ++j) p.v[j] = ed[q * V + j];
      *reinterpret_cast<typename Vec16<T>::type*>(dst) = *reinterpret_cast<const typename Vec16<T>::type*>(&p);
    } else dst[0] = ed[q];
  }
  __builtin_amdgcn_wave_barrier();
  T* ybs = ybar + s * nv;
#pragma unroll
  for (int i = 0; i < CHV; ++i) {
    const int e = (lane + 64 * i) * V;
    const T* src = tile + (e / CH) * S + e % CH;
    if (V > 1) {
      if (e + V <= nv) {
        Pack<T, V> p;
        *reinterpret_cast<typename Vec16<T>::type*>(&p) = *reinterpret_cast<const typename Vec16<T>::type*>(src);
        store_pack<T, V, true>(ybs + e, p);
      }
    } else if (e < nv) ybs[e] = src[0];
  }
}

// corr.jl:314-337 (_link_chol_lkj_from_upper / _from_lower) ; log-det = -_logabsdetjac_inv_chol(y) (:235-237)
template <class T>
__global__ __launch_bounds__(256) void chol_fwd_kernel(const T* W, T* y, T* ladj_ps, int64_t K, int64_t batch, int lower,
                                                       int accumulate, int want_ladj, double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nv = K * (K - 1) / 2;
  double acc = 0.0;
  for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < batch; s += (int64_t)gridDim.x * 4) {   // grid = batch/4: one trip
    const T* Ws = W + s * K * K;
    T* ys = y + s * nv;
    // pass A, descending: remainder_sq = W[j,j]^2 + Σ_{k>i} W[k,j]^2 (suffix sum inside the column)
    T carry = T(0);   // Σ w² of the straddling column's entries that live in higher steps
    const int64_t steps = (nv + 63) / 64;
    for (int64_t st = steps - 1; st >= 0; --st) {
      const int64_t e = st * 64 + lane;
      const bool valid = e < nv;
      int c = 1, i0 = 0;
      if (valid) triu1_decode(e, c, i0);
      const T w = valid ? (!lower ? Ws[(int64_t)c * K + i0] : Ws[(int64_t)i0 * K + c]) : T(0);
      const T dg = valid ? Ws[(int64_t)c * K + c] : T(1);
      // inclusive segmented SUFFIX scan of w² (a prefix-difference would cancel catastrophically:
      // deep entries are ~1e-7 next to O(1) neighbours).  Flags mark column tails.
      const bool is_tail = valid && (i0 == c - 1);
      T v = w * w;
      int f = (is_tail || !valid) ? 1 : 0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const T vo = __shfl_down(v, d, 64);
        const int fo = __shfl_down(f, d, 64);
        if (lane + d < 64 && !f) { v += vo; f |= fo; }
      }
      const T incl = v + (f ? T(0) : carry);          // Σ_{k>=i0} w_k² over my column
      const T nxt = __shfl_down(incl, 1, 64);
      const T suffix = is_tail ? T(0) : (lane == 63 ? carry : nxt);   // Σ_{k>i0} w_k²
      if (valid) {
        T yv;
        if (i0 == 0) yv = x_atanh(w);                            // :322
        else yv = x_asinh(w * Fast<T>::rsqrt(dg * dg + suffix)); // :327-329
        ys[e] = yv;
      }
      // carry for the next (lower) step: the part of lane 0's column that lives in this and higher steps
      int c0, i00;
      triu1_decode(st * 64, c0, i00);
      const T incl0 = __shfl(incl, 0, 64);
      carry = (i00 == 0) ? T(0) : incl0;
    }
    if (want_ladj) {
      // pass B, ascending over the y just written (same wave, same lanes -> program order)
      T cr = T(0), lj = T(0);
      for (int64_t e0 = 0; e0 < nv; e0 += 64) {
        const int64_t e = e0 + lane;
        const bool valid = e < nv;
        int c = 1, i0 = 0;
        if (valid) triu1_decode(e, c, i0);
        const T lc = valid ? f_logcosh(ys[e]) : T(0);
        const T incl = seg_prefix_incl<T>(lc, !valid || i0 == 0, cr);
        const bool last = valid && (i0 == c - 1);
        if (valid) lj += last ? T(-2) * incl : -incl;
        const T incl63 = __shfl(incl, 63, 64);
        const int last63 = __shfl((int)last, 63, 64);
        cr = last63 ? T(0) : incl63;
      }
      lj = -group_sum<64>(lj);
      if (lane == 0) {
        if (ladj_ps) ladj_ps[s] = accumulate ? ladj_ps[s] + lj : lj;
        acc += (double)lj;
      }
    }
  }
  __syncthreads();
  if (partials) block_publish_partial(acc, red, partials);
}

// ------------------------------------------------------------------ VecCholesky, small K: ONE LANE per sample
// LKJCholesky blocks in models are 2x2 ... 10x10; the wave-per-sample kernels above spend 64 lanes on a handful of entries
// (K = 4: 3 % of the roofline).  A wave takes 64 consecutive samples — one contiguous run of the input and of the output,
// 16-byte accesses through a [64][P odd] LDS tile — and lane t walks the columns of sample t in place: forward
// (corr.jl:314-335) columns ascending (y lands below the column it came from), inverse (corr.jl:370-399) columns DESCENDING
// (W lands above the y it came from); :L factors are transposed inside the lane's tile row before / after the walk.
#include "bjx_linkmath.h"

template <class T, bool INV, int V>
__global__ __launch_bounds__(64) void chol_lane_kernel(const T* __restrict__ in, T* __restrict__ out, T* __restrict__ ladj_ps, int K, int P, int lower,
                                                       int64_t batch, int accumulate, double* partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[1];
  using M = LinkMath<T>;
  T* tile = reinterpret_cast<T*>(smem);
  const int lane = threadIdx.x;
  const int KK = K * K, nv = K * (K - 1) / 2;
  const int n_in = INV ? nv : KK, n_out = INV ? KK : nv;
  double acc = 0.0;
  for (int64_t s0 = (int64_t)blockIdx.x * 64; s0 < batch; s0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - s0) < 64 ? (batch - s0) : 64);
    if (n_in > 0) tile_stage_in<T, V>(tile, in + s0 * n_in, n_in, P, ncols, lane);
    tile_sync();
    T* mine = tile + lane * P;
    T lsum = T(0);
    if constexpr (!INV) {
      if (lower)                                                   // W' is the upper factor
        for (int j = 1; j < K; ++j)
          for (int i = 0; i < j; ++i) mine[j * K + i] = mine[i * K + j];
      for (int c = 1; c < K; ++c) {
        const T* col = mine + c * K;
        T* yo = mine + c * (c - 1) / 2;
        T rem, Lr;
        M::fwd_init(col[c], rem, Lr);
        for (int i = c - 1; i >= 1; --i) {
          T y, lc;
          M::fwd_step(col[i], rem, Lr, y, lc);
          lsum += T(c - i + 1) * lc;                               // -logabsdetjac_inv_chol (corr.jl:235-237, :485-501): entry i of column c counts c - i + 1 times
          yo[i] = y;
        }
        T y0, lc0;
        M::atanh_lc(col[0], y0, lc0);                              // :322
        lsum += T(c + 1) * lc0;
        yo[0] = y0;
      }
    } else {
      for (int c = K - 1; c >= 0; --c) {
        const T* yi = mine + c * (c - 1) / 2;
        T* col = mine + c * K;
        T lr = T(0), E;
        M::inv_init(E);
        for (int i = 0; i < c; ++i) {
          T w, lc;
          M::inv_step(yi[i], E, w, lc);
          lr -= lc;
          lsum += lr;
          if (out) col[i] = w;
        }
        lsum += lr;
        if (out) {
          col[c] = M::inv_diag(E, lr);
          for (int i = c + 1; i < K; ++i) col[i] = T(0);
        }
      }
      if (out && lower)
        for (int j = 1; j < K; ++j)
          for (int i = 0; i < j; ++i) { mine[i * K + j] = mine[j * K + i]; mine[j * K + i] = T(0); }
    }
    tile_sync();
    if (out && n_out > 0) tile_stage_out<T, V>(tile, out + s0 * n_out, n_out, P, ncols, lane);
    tile_sync();
    if (lane < ncols) {
      if (ladj_ps) ladj_ps[s0 + lane] = accumulate ? ladj_ps[s0 + lane] + lsum : lsum;
      acc += (double)lsum;
    }
  }
  if (partials) block_publish_partial(acc, red, partials);
}

template <class T>
int chol_impl(bjx_ctx* ctx, int inverse, int uplo, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t K, int64_t batch,
              uint32_t flags) {
  if (batch == 0) {
    if (ladj_sum && !(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
    return BJX_OK;
  }
  const int lower = (uplo == 'L') ? 1 : 0;
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  {
    // small factors: one lane per sample (chol_lane_kernel) while 64 samples fit an LDS tile with room for 4+ waves per CU
    static const int lane_max = getenv("BJX_CHOL_LANE_MAX") ? atoi(getenv("BJX_CHOL_LANE_MAX")) : 11;     // tuning switch (0: off)
    const int64_t P = (K * K) | 1;
    const size_t smem_l = (size_t)64 * P * sizeof(T);
    if (K >= 2 && K <= lane_max && smem_l <= 36 * 1024) {
      const int64_t tiles = (batch + 63) / 64;
      const int64_t cap = (int64_t)ctx->num_cu * 32;
      const int grid_l = (int)(tiles < cap ? tiles : cap);
      if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid_l); if (rc) return rc; }
      double* partials_l = ladj_sum ? ctx->partials : nullptr;
      constexpr int VW = Vec16<T>::N;
      const bool vec = bjx_aligned16(in) && (!out || bjx_aligned16(out));
      {
        BjxProf prof_(ctx);
#define CHOL_LN(INV_, V_) hipLaunchKernelGGL((chol_lane_kernel<T, INV_, V_>), dim3(grid_l), dim3(64), smem_l, ctx->stream, in, out, ladj_ps, (int)K, (int)P, lower, batch, accum, partials_l)
        if (inverse) { if (vec) CHOL_LN(true, VW); else CHOL_LN(true, 1); }
        else { if (vec) CHOL_LN(false, VW); else CHOL_LN(false, 1); }
#undef CHOL_LN
      }
      BJX_CHECK_LAUNCH(ctx);
      if (ladj_sum) return bjx_launch_finalize(ctx, grid_l, ladj_sum, 0.0, 0, 0.0, flags);
      return BJX_OK;
    }
  }
  const int64_t grid = (batch + 3) / 4;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
  double* partials = ladj_sum ? ctx->partials : nullptr;
  if (inverse) {
    // chunk kernel: lane = contiguous chunk of <= 32 packed entries, W tile in LDS (K <= 64 for Float32)
    static const int use_chunk = getenv("BJX_CHOL_CHUNK") ? atoi(getenv("BJX_CHOL_CHUNK")) : 1;
    const int64_t nv = K * (K - 1) / 2;
    constexpr int VW = Vec16<T>::N;
    const bool v_ok = bjx_aligned16(in) && nv % VW == 0;
    const int ch = (int)((nv + 63) / 64);                         // entries per lane
    int chv, vv;                                                  // template CHV, V
    if (v_ok) { vv = VW; const int need = (ch + VW - 1) / VW; chv = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16; }
    else { vv = 1; chv = ch <= 2 ? 2 : ch <= 8 ? 8 : ch <= 16 ? 16 : 32; }
    const int CHn = chv * vv;
    // phantom entries (< 64*CH) reach column cmax: pad the tile so their 'U' writes stay inside it
    int64_t cmax = K;
    while (cmax * (cmax + 1) / 2 <= (int64_t)64 * CHn - 1) ++cmax;
    int64_t tile_words = cmax * (K + 1) + 1;                      // highest phantom write: [cmax][cmax-1] and its diagonal slot
    if (tile_words < K * K) tile_words = K * K;
    if (tile_words < (int64_t)64 * (CHn + vv)) tile_words = (int64_t)64 * (CHn + vv);   // staging of the packed vector
    tile_words = (tile_words + 1 + 3) / 4 * 4;                    // + dummy word, 16-byte multiple
    const size_t tile_bytes = out ? (size_t)CHOL_WPB * tile_words * sizeof(T) : 0;
    // (Float64 at K = 64: two 32.8 KiB tiles per block = 65.6 KiB — above the default dynamic-LDS limit, opted into per kernel;
    //  round 2 sent that shape to the generic block kernel: 34 % of the HBM peak, the forward link 17 %)
    if (use_chunk && K >= 2 && CHn <= 32 && nv <= 64 * 32 && tile_bytes <= 80 * 1024) {
      const int64_t grid = (batch + CHOL_WPB - 1) / CHOL_WPB;
      BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
      if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
      double* partials = ladj_sum ? ctx->partials : nullptr;
#define CHOL_K(V_, CHV_, W_, L_) bjx_allow_big_lds(chol_inv_chunk_kernel<T, V_, CHV_, W_, L_>, tile_bytes); hipLaunchKernelGGL((chol_inv_chunk_kernel<T, V_, CHV_, W_, L_>), dim3((unsigned)grid), dim3(64 * CHOL_WPB), tile_bytes, ctx->stream, in, out, ladj_ps, (int)K, (int)tile_words, batch, accum, partials)
#define CHOL_L(V_, CHV_) do { if (!out) { CHOL_K(V_, CHV_, false, false); } else if (lower) { CHOL_K(V_, CHV_, true, true); } else { CHOL_K(V_, CHV_, true, false); } } while (0)
      { BjxProf prof_(ctx);
      if (vv == VW) {
        if (chv == 1) CHOL_L(VW, 1); else if (chv == 2) CHOL_L(VW, 2); else if (chv == 4) CHOL_L(VW, 4);
        else if (chv == 8 && VW * 8 <= 32) CHOL_L(VW, (VW * 8 <= 32 ? 8 : 1));
        else if (VW == 2 && chv == 8) CHOL_L(VW, 8); else CHOL_L(VW, (VW == 2 ? 16 : 1));
      } else {
        if (chv == 2) CHOL_L(1, 2); else if (chv == 8) CHOL_L(1, 8); else if (chv == 16) CHOL_L(1, 16); else CHOL_L(1, 32);
      } }
#undef CHOL_L
#undef CHOL_K
      BJX_CHECK_LAUNCH(ctx);
      if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
      return BJX_OK;
    }
    const size_t smem = 32 + (size_t)4 * K * sizeof(T);
    BJX_REQUIRE(ctx, smem <= 64 * 1024, BJX_ERR_UNSUPPORTED, "bjx_vec_cholesky: K = %lld too large", (long long)K);
    BjxProf prof_(ctx);
    if (out) hipLaunchKernelGGL((chol_inv_kernel<T, true>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, in, out, ladj_ps, K, batch, lower, accum, partials);
    else hipLaunchKernelGGL((chol_inv_kernel<T, false>), dim3((unsigned)grid), dim3(256), smem, ctx->stream, in, out, ladj_ps, K, batch, lower, accum, partials);
  } else {
    const int want = (ladj_ps || ladj_sum) ? 1 : 0;
    {
      static const int use_chunk = getenv("BJX_CHOL_CHUNK") ? atoi(getenv("BJX_CHOL_CHUNK")) : 1;
      const int64_t nv = K * (K - 1) / 2;
      constexpr int VW = Vec16<T>::N;
      const bool v_ok = bjx_aligned16(out) && nv % VW == 0;
      const int ch = (int)((nv + 63) / 64);
      int chv, vv;
      if (v_ok) { vv = VW; const int need = (ch + VW - 1) / VW; chv = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16; }
      else { vv = 1; chv = ch <= 2 ? 2 : ch <= 8 ? 8 : ch <= 16 ? 16 : 32; }
      const int CHn = chv * vv;
      int64_t tile_words = (K * K + 3) / 4 * 4;
      if (tile_words < (int64_t)64 * (CHn + vv)) tile_words = (int64_t)64 * (CHn + vv);   // staging of the packed vector
      const size_t tile_bytes = (size_t)CHOL_WPB * tile_words * sizeof(T);
      if (use_chunk && K >= 2 && CHn <= 32 && nv <= 64 * 32 && tile_bytes <= 80 * 1024) {
        const int64_t grid = (batch + CHOL_WPB - 1) / CHOL_WPB;
        BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
        if (ladj_sum) { int rc = bjx_ensure_partials(ctx, (size_t)grid); if (rc) return rc; }
        double* partials = ladj_sum ? ctx->partials : nullptr;
#define CHOLF_K(V_, CHV_, L_, J_) bjx_allow_big_lds(chol_fwd_chunk_kernel<T, V_, CHV_, L_, J_>, tile_bytes); hipLaunchKernelGGL((chol_fwd_chunk_kernel<T, V_, CHV_, L_, J_>), dim3((unsigned)grid), dim3(64 * CHOL_WPB), tile_bytes, ctx->stream, in, out, ladj_ps, (int)K, (int)tile_words, batch, accum, partials)
#define CHOLF_L(V_, CHV_) do { if (lower) { if (want) { CHOLF_K(V_, CHV_, true, true); } else { CHOLF_K(V_, CHV_, true, false); } } else { if (want) { CHOLF_K(V_, CHV_, false, true); } else { CHOLF_K(V_, CHV_, false, false); } } } while (0)
        { BjxProf prof_(ctx);
        if (vv == VW) {
          if (chv == 1) CHOLF_L(VW, 1); else if (chv == 2) CHOLF_L(VW, 2); else if (chv == 4) CHOLF_L(VW, 4);
          else if (chv == 8 && VW * 8 <= 32) CHOLF_L(VW, (VW * 8 <= 32 ? 8 : 1));
          else if (VW == 2 && chv == 8) CHOLF_L(VW, 8); else CHOLF_L(VW, (VW == 2 ? 16 : 1));
        } else {
          if (chv == 2) CHOLF_L(1, 2); else if (chv == 8) CHOLF_L(1, 8); else if (chv == 16) CHOLF_L(1, 16); else CHOLF_L(1, 32);
        } }
#undef CHOLF_L
#undef CHOLF_K
        BJX_CHECK_LAUNCH(ctx);
        if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
        return BJX_OK;
      }
    }
    BjxProf prof_(ctx);
    hipLaunchKernelGGL((chol_fwd_kernel<T>), dim3((unsigned)grid), dim3(256), 32, ctx->stream, in, out, ladj_ps, K, batch, lower, accum, want, partials);
  }
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) return bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags);
  return BJX_OK;
}
}  // namespace

namespace {
// ------------------------------------------------------------------ OrderedBijector pullbacks
// ext/BijectorsChainRulesCoreExt.jl:65-197 (rrules of _transform_ordered / _transform_inverse_ordered,
// matrix methods), extended by the log-det cotangent so that ONE call is the pullback of
// with_logabsdet_jacobian:   in_bar = J(in)ᵀ · out_bar + ladj_bar · ∇_in logabsdetjac.
//   forward  x = b(y): x_1 = y_1, x_i = x_{i-1} + exp(y_i), ladj = Σ_{i>=2} y_i
//            y_bar[1] = Σ_k x_bar[k];  y_bar[i] = (Σ_{k>=i} x_bar[k]) · exp(y_i) + ladj_bar       (:74-88)
//   inverse  y = b⁻¹(x): r_1 = 1, r_i = x_i - x_{i-1}, y_i = log r_i, ladj = -Σ_{i>=2} y_i
//            with Δ_1 = y_bar[1], Δ_i = y_bar[i] - ladj_bar:  x_bar[j] = Δ_j / r_j - Δ_{j+1} / r_{j+1}  (:136-147)
// Same wave-private tiles as the forward kernels: two [64][P] tiles (primal input, output cotangent),
// one lane per column; the suffix sum runs from the last row up.
template <class T, int V, bool INV>
__global__ __launch_bounds__(64) void ordered_vjp_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                        T* __restrict__ in_bar, int rows, int P, int64_t batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tin = reinterpret_cast<T*>(smem);
  T* tg = tin + (size_t)64 * P;
  const int lane = threadIdx.x;
  const int64_t col0 = (int64_t)blockIdx.x * 64;
  const int ncols = (int)((batch - col0) < 64 ? (batch - col0) : 64);
  tile_stage_in<T, V>(tin, in + col0 * rows, rows, P, ncols, lane);
  tile_stage_in<T, V>(tg, out_bar + col0 * rows, rows, P, ncols, lane);
  tile_sync();
  if (lane < ncols) {
    const T lb = ladj_bar ? ladj_bar[col0 + lane] : T(0);
    const T* a = tin + lane * P;
    T* g = tg + lane * P;
    if (!INV) {
      T sfx = T(0);
      for (int i = rows - 1; i >= 1; --i) { sfx += g[i]; g[i] = sfx * d_exp(a[i]) + lb; }
      g[0] = sfx + g[0];
    } else {
      // walk up keeping Δ_{i+1}/r_{i+1}
      T nxt = T(0);
      for (int i = rows - 1; i >= 1; --i) {
        const T q = (g[i] - lb) / (a[i] - a[i - 1]);
        g[i] = q - nxt;
        nxt = q;
      }
      g[0] = g[0] - nxt;
    }
  }
  tile_sync();
  tile_stage_out<T, V>(tg, in_bar + col0 * rows, rows, P, ncols, lane);
}

// last pack of a column whose height is not a whole number of packs: nrow < V live rows one by one, dead rows read as zero
template <class T, int V> __device__ __forceinline__ Pack<T, V> seq_load_pack_part(const T* p, int nrow) {
  if (nrow >= V) return load_pack<T, V, true>(p);
  Pack<T, V> r;
#pragma unroll
  for (int j = 0; j < V; ++j) r.v[j] = j < nrow ? p[j] : T(0);
  return r;
}
template <class T, int V> __device__ __forceinline__ void seq_store_pack_part(T* p, const Pack<T, V>& r, int nrow) {
  if (nrow >= V) { store_pack<T, V, true>(p, r); return; }
#pragma unroll
  for (int j = 0; j < V; ++j) if (j < nrow) p[j] = r.v[j];
}
// Streaming variant (no LDS tiles) when a column is at most 64 packs: G lanes own one column as 16-byte
// packs (coalesced), 4 columns in flight per lane group.  The inverse's pullback is local (each entry needs its
// two neighbours: one lane shuffle each way); the forward's needs the suffix sum of the output cotangent
// along the column: a 4-element suffix per lane + a reversed inclusive scan over the G lanes.
// (The two-tile kernel below holds 33 KiB of LDS per wave at dim = 64: 4 waves per CU, 42 % of the HBM roofline.)
// Round 4: R packs per lane (pack v = r G + gl: every load / store instruction of a column group is one contiguous run) and a PARTIAL
// last pack on element-aligned addresses — any height up to 64 R packs (2 048 rows in Float32).  Odd heights and columns taller than
// 64 packs used to fall to the two-tile walker (103 KiB of LDS per wave at 201 rows: 12 % of the HBM peak) or, past the LDS, to one
// thread per column (333 rows: 3.5 %).
template <class T, int V, int R, bool INV>
__global__ __launch_bounds__(256) void ordered_vjp_stream_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                                T* __restrict__ in_bar, int64_t dim, int64_t batch, int G) {
  constexpr int UC = R == 1 ? 4 : (R == 2 ? 2 : 1);
  const int gl = threadIdx.x & (G - 1);
  const int cols_per_block = 256 / G;
  const int nvc = (int)((dim + V - 1) / V);
  const int64_t col0 = (int64_t)blockIdx.x * cols_per_block * UC + threadIdx.x / G;
  Pack<T, V> a[UC][R], g[UC][R];
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t col = col0 + (int64_t)u * cols_per_block;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int v = r * G + gl;
      if (v < nvc && col < batch) {
        const int nrow = (int)(dim - (int64_t)v * V < V ? dim - (int64_t)v * V : V);
        a[u][r] = seq_load_pack_part<T, V>(in + col * dim + (int64_t)v * V, nrow);
        g[u][r] = seq_load_pack_part<T, V>(out_bar + col * dim + (int64_t)v * V, nrow);
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) { a[u][r].v[j] = T(0); g[u][r].v[j] = T(0); }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t col = col0 + (int64_t)u * cols_per_block;
    const T lb = (ladj_bar && col < batch) ? ladj_bar[col] : T(0);
    Pack<T, V> o[R];
    if (!INV) {
      // suffix sums of the output cotangent along the column (dead rows of the last pack and dead packs hold zeros): inside a
      // pack, then over the G lanes of pack round r (reversed inclusive scan), then the totals of the rounds above
      T above = T(0);
#pragma unroll
      for (int r = R - 1; r >= 0; --r) {
        T sfx[V];
        T run = T(0);
#pragma unroll
        for (int j = V - 1; j >= 0; --j) { run += g[u][r].v[j]; sfx[j] = run; }
        T tot = run;
        for (int d = 1; d < G; d <<= 1) {
          const T o2 = __shfl_down(tot, d, 64);
          if (gl + d < G) tot += o2;
        }
        const T right = tot - run + above;                      // Σ of everything below my pack in the column
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const T sv = sfx[j] + right;
          o[r].v[j] = (r == 0 && gl == 0 && j == 0) ? sv : sv * d_exp(a[u][r].v[j]) + lb;
        }
        above += __shfl(tot, (threadIdx.x & 63) & ~(G - 1), 64);  // the round's total sits in the group's first lane
      }
    } else {
      // q_i = (ȳ_i - ℓ̄)/(x_i - x_{i-1}) (q_0 = ȳ_0), x̄_i = q_i - q_{i+1}; rows past the column's end have q = 0
      T q[R][V + 1];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T xprev = __shfl_up(a[u][r].v[V - 1], 1, 64);             // x_{i-1} of my first element: the lane to my left in this round ...
        if (r > 0) { const T wrap = __shfl(a[u][r - 1].v[V - 1], ((threadIdx.x & 63) & ~(G - 1)) + G - 1, 64); if (gl == 0) xprev = wrap; }   // ... or the last lane of the round before
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const int64_t row = ((int64_t)r * G + gl) * V + j;
          const T xm1 = j == 0 ? xprev : a[u][r].v[j - 1];
          q[r][j] = row == 0 ? g[u][r].v[0] : (row < dim ? (g[u][r].v[j] - lb) / (a[u][r].v[j] - xm1) : T(0));
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T qn = __shfl_down(q[r][0], 1, 64);                      // q of the next pack: my right neighbour, or the first lane of the next round
        if (gl == G - 1) qn = T(0);
        if (r + 1 < R) { const T wrap = __shfl(q[r + 1][0], (threadIdx.x & 63) & ~(G - 1), 64); if (gl == G - 1) qn = wrap; }
        q[r][V] = qn;
#pragma unroll
        for (int j = 0; j < V; ++j) o[r].v[j] = q[r][j] - q[r][j + 1];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int v = r * G + gl;
      if (v < nvc && col < batch) {
        const int nrow = (int)(dim - (int64_t)v * V < V ? dim - (int64_t)v * V : V);
        seq_store_pack_part<T, V>(in_bar + col * dim + (int64_t)v * V, o[r], nrow);
      }
    }
  }
}

// fallback for columns too long for two LDS tiles: one thread per column, straight from global memory
template <class T, bool INV>
__global__ __launch_bounds__(256) void ordered_vjp_column_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                                 T* __restrict__ in_bar, int64_t rows, int64_t batch) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= batch) return;
  const T lb = ladj_bar ? ladj_bar[col] : T(0);
  const T* a = in + col * rows;
  const T* g = out_bar + col * rows;
  T* o = in_bar + col * rows;
  if (!INV) {
    T sfx = T(0);
    for (int64_t i = rows - 1; i >= 1; --i) { sfx += g[i]; o[i] = sfx * d_exp(a[i]) + lb; }
    o[0] = sfx + g[0];
  } else {
    T nxt = T(0);
    for (int64_t i = rows - 1; i >= 1; --i) {
      const T q = (g[i] - lb) / (a[i] - a[i - 1]);
      o[i] = q - nxt;
      nxt = q;
    }
    o[0] = g[0] - nxt;
  }
}

// Columns beyond the stream kernel's 64 lanes x 8 packs (2 048 rows Float32): ONE BLOCK per column (round 5; the one-thread-per-column
// kernel above reads a column with one lane: 4-8 % of the HBM peak).  The column is walked in tiles of 256 packs FROM ITS END, every
// access a coalesced 16-byte pack (element accesses when the height or a base is not pack-aligned):
//   forward  ȳ_i = (Σ_{k>=i} x̄_k) exp(y_i) + ℓ̄  (i >= 2), ȳ_1 = Σ_k x̄_k: a suffix sum — in-pack suffix, wave scan (shuffles), the
//            four wave totals through LDS, plus the carry of the tiles already done;
//   inverse  x̄_j = q_j - q_{j+1}, q_j = (ȳ_j - ℓ̄)/(x_j - x_{j-1}) (q_1 := ȳ_1 ... see ordered_vjp_column_kernel): neighbours only —
//            the pack's own rows plus one row on either side.
template <class T, int V, bool INV>
__global__ __launch_bounds__(256) void ordered_vjp_tall_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                               T* __restrict__ in_bar, int64_t rows, int64_t batch) {
  __shared__ T wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t npk = (rows + V - 1) / V;                       // the last pack may be partial (V = 1 when nothing is pack-aligned)
  const int64_t ntile = (npk + 255) / 256;
  for (int64_t col = blockIdx.x; col < batch; col += gridDim.x) {
    const T lb = ladj_bar ? ladj_bar[col] : T(0);
    const T* a = in + col * rows;
    const T* g = out_bar + col * rows;
    T* o = in_bar + col * rows;
    T carry = T(0);                                              // forward: Σ x̄ of the tiles behind this one
    for (int64_t tl = ntile - 1; tl >= 0; --tl) {
      const int64_t pk = tl * 256 + threadIdx.x;
      const int64_t r0 = pk * V;
      const bool live = pk < npk;
      T gv[V], av[V];
#pragma unroll
      for (int j = 0; j < V; ++j) { const bool ok = live && r0 + j < rows; gv[j] = ok ? g[r0 + j] : T(0); av[j] = ok ? a[r0 + j] : T(0); }
      if (!INV) {
        // suffix sums inside the pack, then over the lanes behind this one, the waves behind this one, the tiles behind this one
        T sfx[V];
        T run = T(0);
#pragma unroll
        for (int j = V - 1; j >= 0; --j) { run += gv[j]; sfx[j] = run; }
        T inc = run;                                             // inclusive suffix scan of the pack totals across the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const T v = __shfl_down(inc, off, 64); if (lane + off < 64) inc += v; }
        const T wave_total = __shfl(inc, 0, 64);
        __syncthreads();                                         // wsum of the previous tile has been read
        if (lane == 0) wsum[wave] = wave_total;
        __syncthreads();
        T behind = carry;
#pragma unroll
        for (int w = 3; w >= 0; --w) if (w > wave) behind += wsum[w];
        const T after = behind + (inc - run);                    // Σ x̄ of every row behind this pack
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const int64_t r = r0 + j;
          if (live && r < rows) o[r] = r == 0 ? (sfx[j] + after) : (sfx[j] + after) * d_exp(av[j]) + lb;
        }
        carry += (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
      } else {
        // q of the pack's rows and of the row behind it; row r needs x_{r-1}
        const T a_prev = (live && r0 >= 1 && r0 < rows) ? a[r0 - 1] : T(0);
        const int64_t rn = r0 + V;                               // the row behind the pack
        const T q_next = (live && rn < rows) ? (g[rn] - lb) / (a[rn] - a[rn - 1]) : T(0);
        T q[V + 1];
        q[V] = q_next;
#pragma unroll
        for (int j = V - 1; j >= 0; --j) {
          const int64_t r = r0 + j;
          const T below = j == 0 ? a_prev : av[j - 1];
          q[j] = (live && r >= 1 && r < rows) ? (gv[j] - lb) / (av[j] - below) : T(0);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const int64_t r = r0 + j;
          if (live && r < rows) o[r] = r == 0 ? gv[j] - q[j + 1] : q[j] - q[j + 1];
        }
      }
    }
    if (!INV) __syncthreads();                                   // the next column's first tile rewrites wsum
  }
}

template <class T>
int ordered_vjp_impl(bjx_ctx* ctx, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t dim, int64_t batch) {
  if (batch == 0) return BJX_OK;
  {
    bool taken = false;                                                // 1 ... 8 rows: lane = column in registers (bjx_tiny.hip)
    const int rc = bjx_seq_tiny_vjp(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, 0, inverse, in, out_bar, ladj_bar, in_bar, dim, batch, &taken);
    if (rc || taken) return rc;
  }
  {
    constexpr int VW = Vec16<T>::N;
    static const int use_stream = getenv("BJX_ORDERED_VJP_STREAM") ? atoi(getenv("BJX_ORDERED_VJP_STREAM")) : 1;
    const int64_t packs = (dim + VW - 1) / VW;
    if (use_stream && dim >= 2 * VW && packs <= 64 * 8) {
      // whole aligned packs or not: 16-byte packs on element-aligned addresses, a partial last pack, R packs per lane beyond 64
      int G = 1;
      while (G < 64 && G < packs) G <<= 1;
      int R = 1;
      while ((int64_t)R * G < packs) R <<= 1;
      const int uc = R == 1 ? 4 : (R == 2 ? 2 : 1);
      const int64_t cpb = (int64_t)(256 / G) * uc;
      const int64_t grid = (batch + cpb - 1) / cpb;
      BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
      {
        BjxProf prof_(ctx);
#define OVS(R_) do { if (inverse) hipLaunchKernelGGL((ordered_vjp_stream_kernel<T, VW, R_, true>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, dim, batch, G); \
                     else hipLaunchKernelGGL((ordered_vjp_stream_kernel<T, VW, R_, false>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, dim, batch, G); } while (0)
        switch (R) { case 1: OVS(1); break; case 2: OVS(2); break; case 4: OVS(4); break; default: OVS(8); break; }
#undef OVS
      }
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
  }
  const int64_t P = dim | 1;
  const size_t smem = (size_t)2 * 64 * P * sizeof(T);
  if (smem > BJX_LDS_MAX) {
    BJX_REQUIRE(ctx, in_bar != out_bar, BJX_ERR_ARG, "bjx_ordered_vjp: in_bar may not alias out_bar for dim = %lld", (long long)dim);
    static const int use_tall = getenv("BJX_ORDERED_VJP_TALL") ? atoi(getenv("BJX_ORDERED_VJP_TALL")) : 1;
    if (use_tall && dim >= 1024 && in_bar != in) {
      const int64_t capt = (int64_t)ctx->num_cu * 8;
      const int gridt = (int)(batch < capt ? batch : capt);
      constexpr int VWt = Vec16<T>::N;
      const bool v_ok = dim % VWt == 0 && bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
      BjxProf prof_(ctx);
#define OVT(V_, I_) hipLaunchKernelGGL((ordered_vjp_tall_kernel<T, V_, I_>), dim3(gridt), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, dim, batch)
      if (v_ok) { if (inverse) OVT(VWt, true); else OVT(VWt, false); }
      else { if (inverse) OVT(1, true); else OVT(1, false); }
#undef OVT
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
    const int64_t g2 = (batch + 255) / 256;
    BJX_REQUIRE(ctx, g2 < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
    BjxProf prof_(ctx);
    if (inverse) hipLaunchKernelGGL((ordered_vjp_column_kernel<T, true>), dim3((unsigned)g2), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, dim, batch);
    else hipLaunchKernelGGL((ordered_vjp_column_kernel<T, false>), dim3((unsigned)g2), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, dim, batch);
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  }
  const int64_t grid = (batch + 63) / 64;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  constexpr int VW = Vec16<T>::N;
  const bool v_ok = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
  {
    BjxProf prof_(ctx);
#define OVJP(V_, I_) bjx_allow_big_lds(ordered_vjp_kernel<T, V_, I_>, smem); hipLaunchKernelGGL((ordered_vjp_kernel<T, V_, I_>), dim3((unsigned)grid), dim3(64), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)dim, (int)P, batch)
    if (v_ok) { if (inverse) { OVJP(VW, true); } else { OVJP(VW, false); } }
    else { if (inverse) { OVJP(1, true); } else { OVJP(1, false); } }
#undef OVJP
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
// ------------------------------------------------------------------ SimplexBijector pullbacks
// Reverse sweeps of the stick-breaking recurrences (simplex.jl:47-64 forward, :102-120 inverse) and of the
// log-det terms (:122-138); the reference's own adjoints are simplex.jl:145-215 (log-det gradient, O(K²)),
// :248-308 (link) and :358-470 (invlink).  Conventions as there: a clamped value has zero derivative and
// max(v, ε) has derivative 1 only where v > ε.  One call = pullback of with_logabsdet_jacobian:
//   inverse (y[K-1] -> x[K], ladj = +Σ t_k):  in = y, out_bar = x_bar[K], in_bar = y_bar[K-1]
//   forward (x[K] -> y[K-1], ladj = -Σ t_k):  in = x, out_bar = y_bar[K-1], in_bar = x_bar[K]
// Two wave-private [64][P] tiles (primal, cotangent), lane = column.  The inverse first re-runs the forward
// recurrence (ascending) leaving x_k in the primal tile; the backward sweep recovers s_k = s_{k+1} - x_k and,
// where x_k is not clamped, z_k = (x_k + ε)/((1+ε-s_k)/(1-2ε)) — a clamped x_k has zero derivative anyway.
template <class T, int V, bool INV>
__global__ __launch_bounds__(64) void simplex_vjp_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                        T* __restrict__ in_bar, int K, int P, int64_t batch, int C) {
  // C = columns per block (64 when two 64-column tiles fit the LDS; fewer lanes work on longer columns otherwise)
  using F = Fast<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* ta = reinterpret_cast<T*>(smem);
  T* tb = ta + (size_t)C * P;
  T* logk = tb + (size_t)C * P;
  const int lane = threadIdx.x;
  for (int i = lane; i < K - 1; i += 64) logk[i] = d_log(T(K - 1 - i));
  const int rows_in = INV ? K - 1 : K, rows_g = INV ? K : K - 1;
  const int64_t col0 = (int64_t)blockIdx.x * C;
  const int ncols = (int)((batch - col0) < C ? (batch - col0) : C);
  tile_stage_in<T, V>(ta, in + col0 * rows_in, rows_in, P, ncols, lane);
  tile_stage_in<T, V>(tb, out_bar + col0 * rows_g, rows_g, P, ncols, lane);
  tile_sync();
  if (lane < ncols) {
    const T e = Num<T>::eps;
    const T c = T(1) / (T(1) - 2 * e), E = T(1) + e, c2 = T(1) - 2 * e;
    const T lb = ladj_bar ? ladj_bar[col0 + lane] : T(0);
    T* a = ta + lane * P;
    T* g = tb + lane * P;
    if (INV) {
      // forward recurrence: a[k] <- x_k, s = Σ x
      T s = T(0);
      auto logistic_at = [&](int k) -> T {
        const T yk = a[k] - logk[k];
        return f_logistic(yk);
      };
      { const T xk = d_clamp((logistic_at(0) - e) * c, T(0), T(1)); a[0] = xk; s = xk; }
      int kf = 1;
      for (; kf + 4 <= K - 1; kf += 4) {               // the four logistics are independent of the running sum
        T z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = logistic_at(kf + j) * c;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const T xk = d_clamp((E - s) * z[j] - e, T(0), T(1)); a[kf + j] = xk; s += xk; }
      }
      for (; kf < K - 1; ++kf) { const T xk = d_clamp((E - s) * c * logistic_at(kf) - e, T(0), T(1)); a[kf] = xk; s += xk; }
      const T last = T(1) - s;
      T sb = (last > T(0) && last < T(1)) ? -g[K - 1] : T(0);
      // rows K-2 .. 1 in groups of 4: everything that depends only on (x_k, s_k) — four reciprocals per row —
      // is evaluated for the whole group first; only the short adjoint chain (sb) is sequential
      int k = K - 2;
      for (; k >= 4; k -= 4) {
        T xk[4], gk[4], dtdx[4], dtds[4], rc[4], z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xk[j] = a[k - j]; gk[j] = g[k - j]; }
        T sk[4];
        sk[0] = s - xk[0]; sk[1] = sk[0] - xk[1]; sk[2] = sk[1] - xk[2]; sk[3] = sk[2] - xk[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          simplex_t_partials<T>(xk[j], sk[j], false, dtdx[j], dtds[j]);
          rc[j] = (E - sk[j]) * c;
          z[j] = (xk[j] + e) * F::rcp(rc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const T xb = gk[j] + sb + lb * dtdx[j];
          sb += lb * dtds[j];
          const T ub = (xk[j] > T(0) && xk[j] < T(1)) ? xb : T(0);
          sb -= ub * c * z[j];
          g[k - j] = ub * rc[j] * z[j] * (T(1) - z[j]);
        }
        s = sk[3];
      }
      for (; k >= 0; --k) {
        const T xk = a[k];
        const T sk = s - xk;
        T dtdx, dtds;
        simplex_t_partials<T>(xk, sk, k == 0, dtdx, dtds);
        const T xb = g[k] + sb + lb * dtdx;
        sb += lb * dtds;
        const T ub = (xk > T(0) && xk < T(1)) ? xb : T(0);
        T zb, z;
        if (k == 0) { zb = ub * c; z = xk * c2 + e; }
        else { const T rc = (E - sk) * c; zb = ub * rc; z = (xk + e) * F::rcp(rc); sb -= ub * c * z; }
        g[k] = zb * z * (T(1) - z);
        s = sk;
      }
    } else {
      T s = T(0);
      {
        T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
        int k = 0;
        for (; k + 4 <= K - 1; k += 4) { s0 += a[k]; s1 += a[k + 1]; s2 += a[k + 2]; s3 += a[k + 3]; }
        for (; k < K - 1; ++k) s0 += a[k];
        s = (s0 + s1) + (s2 + s3);                                    // s_{K-1} = Σ_{j<K-1} x_j
      }
      T sbn = T(0);
      g[K - 1] = T(0);                                                  // row K enters neither y nor the log-det
      int k = K - 2;
      for (; k >= 4; k -= 4) {
        T xk[4], gy[4], dtdx[4], dtds[4], ax[4], as_[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xk[j] = a[k - j]; gy[j] = g[k - j]; }
        T sk[4];
        sk[0] = s - xk[0]; sk[1] = sk[0] - xk[1]; sk[2] = sk[1] - xk[2]; sk[3] = sk[2] - xk[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          simplex_t_partials<T>(xk[j], sk[j], false, dtdx[j], dtds[j]);
          const T rd = F::rcp(E - sk[j]);
          const T an = (xk[j] + e) * c2;
          const T zf = an * rd;
          const T zfb = gy[j] * F::rcp(zf * (T(1) - zf));
          ax[j] = zfb * c2 * rd - lb * dtdx[j];                       // what row k adds to its own cotangent
          as_[j] = zfb * an * rd * rd - lb * dtds[j];                 // ... and to the adjoint of s_k
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[k - j] = sbn + ax[j]; sbn += as_[j]; }
        s = sk[3];
      }
      for (; k >= 0; --k) {
        const T xk = a[k];
        const T sk = s - xk;
        const T gy = g[k];
        T xb = sbn, sb = sbn;
        if (k == 0) {
          const T zf = xk * c2 + e;
          const T zfb = gy * F::rcp(zf * (T(1) - zf));
          xb += zfb * c2;
        } else {
          const T rd = F::rcp(E - sk);
          const T an = (xk + e) * c2;
          const T zf = an * rd;
          const T zfb = gy * F::rcp(zf * (T(1) - zf));
          xb += zfb * c2 * rd;
          sb += zfb * an * rd * rd;
        }
        T dtdx, dtds;
        simplex_t_partials<T>(xk, sk, k == 0, dtdx, dtds);
        xb -= lb * dtdx;
        sb -= lb * dtds;
        g[k] = xb;
        sbn = sb;
        s = sk;
      }
    }
  }
  tile_sync();
  tile_stage_out<T, V>(tb, in_bar + col0 * rows_in, rows_in, P, ncols, lane);
}

// Chunked form of simplex_vjp_kernel for TALL columns (round 3).  The whole-column kernel above needs two [C][K|1] tiles per
// wave and halves its working lanes until they fit: 28 % of the HBM peak at K = 100, 11 % at 200, 6 % at 500.  Here a wave keeps
// its 64 columns and two CHUNK tiles of CH rows (17 KiB, nine waves per CU) and makes two passes over the column:
//   forward map  (in = x):  pass 1 ascending: s = Σ_{j<K-1} x_j;   pass 2 descending: the reverse sweep, chunk by chunk;
//   inverse map  (in = y):  pass 1 ascending: the recurrence x_k = clamp(…) — x_k parked in in_bar (same shape as y) — and s;
//                           pass 2 descending: reads x_k back, sweeps, overwrites in_bar with ȳ.
// The running values (s, the adjoint of s) stay in the lane's registers across chunks.  One extra read of the primal
// (forward) / one extra write + read of x (inverse) against the single-pass kernel: 4/3 and 5/3 of the algorithmic bytes.
template <class T, int CH>
__device__ __forceinline__ void vchunk_load(T* tile, const T* __restrict__ src, int64_t ld, int64_t c0, int64_t rows_lim, int ncols, int lane) {
  // every load of the chunk is issued before the first LDS write; raw buffer loads: one per-lane offset register for the whole
  // chunk, a wave-uniform SGPR step from column pair to column pair, zeros beyond the wave's columns and beyond the rows that exist
  constexpr int P = CH + 1, CPI = 64 / CH, NLD = 64 / CPI;
  const int r = lane % CH, cc = lane / CH;
  const auto rs = bjx_make_rsrc(src, (uint32_t)((int64_t)ncols * ld * (int64_t)sizeof(T)));
  const int vo = c0 + r < rows_lim ? (int)(((int64_t)cc * ld + c0 + r) * (int64_t)sizeof(T)) : 0x7fffff00;
  const int step = (int)(CPI * ld * (int64_t)sizeof(T));
  Pack<T, 1> v[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) v[u] = buf_load_pack_s<T, 1>(rs, vo, u * step);
#pragma unroll
  for (int u = 0; u < NLD; ++u) tile[(u * CPI + cc) * P + r] = v[u].v[0];
}
template <class T, int CH>
__device__ __forceinline__ void vchunk_store(const T* tile, T* __restrict__ dst, int64_t ld, int64_t c0, int64_t rows_lim, int ncols, int lane) {
  constexpr int P = CH + 1, CPI = 64 / CH, NLD = 64 / CPI;
  const int r = lane % CH, cc = lane / CH;
  const auto rs = bjx_make_rsrc(dst, (uint32_t)((int64_t)ncols * ld * (int64_t)sizeof(T)));
  const int vo = c0 + r < rows_lim ? (int)(((int64_t)cc * ld + c0 + r) * (int64_t)sizeof(T)) : 0x7fffff00;
  const int step = (int)(CPI * ld * (int64_t)sizeof(T));
  Pack<T, 1> v[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) v[u].v[0] = tile[(u * CPI + cc) * P + r];
#pragma unroll
  for (int u = 0; u < NLD; ++u) buf_store_pack_s<T, 1>(rs, vo, u * step, v[u]);
}
template <class T, bool INV, bool TABLE>
__global__ __launch_bounds__(64) void simplex_vjp_chunk_kernel(const T* __restrict__ in, const T* __restrict__ out_bar, const T* __restrict__ ladj_bar,
                                                               T* __restrict__ in_bar, int K, int64_t batch) {
  using F = Fast<T>;
  constexpr int CH = 128 / (int)sizeof(T), P = CH + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* ta = reinterpret_cast<T*>(smem);
  T* tb = ta + (size_t)64 * P;
  T* logk = tb + (size_t)64 * P;
  const int lane = threadIdx.x;
  if (TABLE) for (int i = lane; i < K - 1; i += 64) logk[i] = d_log(T(K - 1 - i));
  auto lk = [&](int k) -> T { return TABLE ? logk[k] : d_log(T(K - 1 - k)); };
  const int rows_in = INV ? K - 1 : K, rows_g = INV ? K : K - 1;
  const int64_t col0 = (int64_t)blockIdx.x * 64;
  const int ncols = (int)((batch - col0) < 64 ? (batch - col0) : 64);
  const T* src = in + col0 * rows_in;
  const T* gsrc = out_bar + col0 * rows_g;
  T* dst = in_bar + col0 * rows_in;
  const T e = Num<T>::eps;
  const T c = T(1) / (T(1) - 2 * e), E = T(1) + e, c2 = T(1) - 2 * e;
  const T lb = (ladj_bar && lane < ncols) ? ladj_bar[col0 + lane] : T(0);
  T* a = ta + lane * P;
  T* g = tb + lane * P;
  // ---- pass 1 (ascending)
  T s = T(0);
  T s4[4] = {T(0), T(0), T(0), T(0)};
  for (int c0 = 0; c0 < K - 1; c0 += CH) {
    const int nr = (K - 1 - c0) < CH ? (K - 1 - c0) : CH;
    tile_sync();
    vchunk_load<T, CH>(ta, src, rows_in, c0, K - 1, ncols, lane);
    tile_sync();
    if (lane < ncols) {
      if (!INV) {
        for (int i = 0; i < nr; ++i) s4[(c0 + i) & 3] += a[i];        // the whole-column kernel's four interleaved sums
      } else {
        for (int i = 0; i < nr; ++i) {
          const int k = c0 + i;
          const T z = f_logistic(a[i] - lk(k));
          const T xk = k == 0 ? d_clamp((z - e) * c, T(0), T(1)) : d_clamp((E - s) * c * z - e, T(0), T(1));
          a[i] = xk;
          s += xk;
        }
      }
    }
    if (INV) { tile_sync(); vchunk_store<T, CH>(ta, dst, rows_in, c0, K - 1, ncols, lane); }    // x_k parked in in_bar
  }
  if (!INV) s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  if (INV) __threadfence_block();                                        // the parked x_k are read back below
  // ---- pass 2 (descending): rows K-1 … 0 of the cotangent, K-2 … 0 of the primal
  T sb = T(0);
  bool top = true;
  const int nchunks = (K + CH - 1) / CH;                                  // chunks over K rows (the longer of the two sides)
  for (int ci = nchunks - 1; ci >= 0; --ci) {
    const int c0 = ci * CH;
    tile_sync();
    vchunk_load<T, CH>(ta, INV ? (const T*)dst : src, rows_in, c0, K - 1, ncols, lane);     // x_k, k <= K-2
    vchunk_load<T, CH>(tb, gsrc, rows_g, c0, rows_g, ncols, lane);
    tile_sync();
    if (lane < ncols) {
      int khi = (c0 + CH < K ? c0 + CH : K) - 1;                          // highest row of this chunk (<= K-1)
      if (top) {
        if (INV) { const T last = T(1) - s; sb = (last > T(0) && last < T(1)) ? -g[K - 1 - c0] : T(0); }
        else if (K - 1 - c0 < CH) g[K - 1 - c0] = T(0);                   // row K enters neither y nor the log-det (forward map: rows_in = K)
        top = false;
      }
      if (khi > K - 2) khi = K - 2;
#pragma unroll 2
      for (int k = khi; k >= c0; --k) {
        const int i = k - c0;
        const T xk = a[i];
        const T sk = s - xk;
        T dtdx, dtds;
        simplex_t_partials<T>(xk, sk, k == 0, dtdx, dtds);
        if (INV) {
          const T xb = g[i] + sb + lb * dtdx;
          sb += lb * dtds;
          const T ub = (xk > T(0) && xk < T(1)) ? xb : T(0);
          T zb, z;
          if (k == 0) { zb = ub * c; z = xk * c2 + e; }
          else { const T rc = (E - sk) * c; zb = ub * rc; z = (xk + e) * F::rcp(rc); sb -= ub * c * z; }
          g[i] = zb * z * (T(1) - z);
        } else {
          const T gy = g[i];
          T xb = sb, sn = sb;
          if (k == 0) {
            const T zf = xk * c2 + e;
            xb += gy * F::rcp(zf * (T(1) - zf)) * c2;
          } else {
            const T rd = F::rcp(E - sk);
            const T an = (xk + e) * c2;
            const T zf = an * rd;
            const T zfb = gy * F::rcp(zf * (T(1) - zf));
            xb += zfb * c2 * rd;
            sn += zfb * an * rd * rd;
          }
          xb -= lb * dtdx;
          sn -= lb * dtds;
          g[i] = xb;
          sb = sn;
        }
        s = sk;
      }
    }
    tile_sync();
    vchunk_store<T, CH>(tb, dst, rows_in, c0, rows_in, ncols, lane);
  }
}

template <class T, int G>
int launch_simplex_vjp_stream(bjx_ctx* ctx, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch, bool* taken);

template <class T>
int simplex_vjp_impl(bjx_ctx* ctx, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch) {
  if (batch == 0) return BJX_OK;
  {
    bool taken = false;                                                // 2 ... 8 rows: lane = column in registers (bjx_tiny.hip)
    const int rc = bjx_seq_tiny_vjp(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, 1, inverse, in, out_bar, ladj_bar, in_bar, K, batch, &taken);
    if (rc || taken) return rc;
  }
  {
    bool taken = false;
    static const int g_inv = 2;
    int rc;
    if (!inverse || g_inv == 4) rc = launch_simplex_vjp_stream<T, 4>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch, &taken);
    else rc = launch_simplex_vjp_stream<T, 2>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch, &taken);
    if (rc || taken) return rc;
    if (inverse && g_inv != 4) {         // K not a multiple of 2·V·{1,2,4,8}: try four lanes per column
      rc = launch_simplex_vjp_stream<T, 4>(ctx, inverse, in, out_bar, ladj_bar, in_bar, K, batch, &taken);
      if (rc || taken) return rc;
    }
    rc = bjx_tall_simplex_vjp(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, inverse, in, out_bar, ladj_bar, in_bar, K, batch, &taken);   // taller: G lanes per column (bjx_tall.hip)
    if (rc || taken) return rc;
  }
  // tall columns: the chunked two-pass kernel (two whole-column tiles of 64 columns cost 2·64·K words: beyond `chunk_min` bytes
  // the whole-column kernel runs with too few waves, then with too few lanes)
  // (K = 100: 22 / 13 % against 29 / 23 % for the whole-column kernel; K = 200: 24 / 13 % against 11 / 9 %; K = 500: 24 / 14 % against 6 / 5 %)
  static const long vjp_chunk_min = 80 * 1024;
  if (((size_t)2 * 64 * (K | 1) + (size_t)K) * sizeof(T) > (size_t)vjp_chunk_min) {
    constexpr int CH = 128 / (int)sizeof(T);
    const bool table = (size_t)K * sizeof(T) <= 40 * 1024;
    const size_t smem_c = ((size_t)2 * 64 * (CH + 1) + (table ? (size_t)K : 1)) * sizeof(T);
    BJX_REQUIRE(ctx, (int64_t)64 * K * (int64_t)sizeof(T) < ((int64_t)1 << 31), BJX_ERR_UNSUPPORTED, "bjx_simplex_vjp: K = %lld too large for one buffer descriptor per wave", (long long)K);
    const int64_t grid_c = (batch + 63) / 64;
    BJX_REQUIRE(ctx, grid_c < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
    {
      BjxProf prof_(ctx);
#define SVC(I_, TB_) hipLaunchKernelGGL((simplex_vjp_chunk_kernel<T, I_, TB_>), dim3((unsigned)grid_c), dim3(64), smem_c, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, batch)
      if (inverse) { if (table) SVC(true, true); else SVC(true, false); }
      else { if (table) SVC(false, true); else SVC(false, false); }
#undef SVC
    }
    BJX_CHECK_LAUNCH(ctx);
    return BJX_OK;
  }
  const int64_t P = K | 1;
  int C = 64;
  while (C > 1 && ((size_t)2 * C * P + (size_t)K) * sizeof(T) > BJX_LDS_MAX) C >>= 1;
  const size_t smem = ((size_t)2 * C * P + (size_t)K) * sizeof(T);
  BJX_REQUIRE(ctx, smem <= BJX_LDS_MAX, BJX_ERR_UNSUPPORTED, "bjx_simplex_vjp: K = %lld too large for the LDS tiles", (long long)K);
  const int64_t grid = (batch + C - 1) / C;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  constexpr int VW = Vec16<T>::N;
  const bool v_ok = bjx_aligned16(in) && bjx_aligned16(out_bar) && bjx_aligned16(in_bar);
  {
    BjxProf prof_(ctx);
#define SVJPK(V_, I_) bjx_allow_big_lds(simplex_vjp_kernel<T, V_, I_>, smem); hipLaunchKernelGGL((simplex_vjp_kernel<T, V_, I_>), dim3((unsigned)grid), dim3(64), smem, ctx->stream, in, out_bar, ladj_bar, in_bar, (int)K, (int)P, batch, C)
    if (v_ok) { if (inverse) { SVJPK(VW, true); } else { SVJPK(VW, false); } }
    else { if (inverse) { SVJPK(1, true); } else { SVJPK(1, false); } }
#undef SVJPK
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_simplex_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar,
                            void* in_bar, int64_t K, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, K > 1, BJX_ERR_SHAPE, "bjx_simplex_vjp: x needs to be of length greater than 1 (simplex.jl:30), got K=%lld", (long long)K);
  BJX_REQUIRE(ctx, batch >= 0, BJX_ERR_SHAPE, "bjx_simplex_vjp: negative batch");
  BJX_REQUIRE(ctx, (in && out_bar && in_bar) || batch == 0, BJX_ERR_ARG, "bjx_simplex_vjp: null pointer");
  if (dt == BJX_F32) return simplex_vjp_impl<float>(ctx, inverse, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
  if (dt == BJX_F64) return simplex_vjp_impl<double>(ctx, inverse, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_simplex_vjp: bad dtype %d", (int)dt);
}

BJX_API int bjx_ordered_vjp(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, const void* out_bar, const void* ladj_bar,
                            void* in_bar, int64_t dim, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_ordered_vjp: bad size");
  BJX_REQUIRE(ctx, (in && out_bar && in_bar) || batch == 0, BJX_ERR_ARG, "bjx_ordered_vjp: null pointer");
  if (dt == BJX_F32) return ordered_vjp_impl<float>(ctx, inverse, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, dim, batch);
  if (dt == BJX_F64) return ordered_vjp_impl<double>(ctx, inverse, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, dim, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_ordered_vjp: bad dtype %d", (int)dt);
}

namespace {
// ------------------------------------------------------------------ Ordered / Simplex, streaming kernels (no whole-column tile)
// ordered.jl:36-80, simplex.jl:47-138 for columns of R = G·NP·V rows (R <= 64 Float32 / 32 Float64).  These maps
// couple the rows of a column only through ONE running value (Σ_{j<k} x_j, the previous y), so a column does not
// need a walker lane with the whole column in LDS (seq_wave_kernel: 16.6 KiB per wave, ~2 waves per SIMD, and
// ~37 VALU per element of staging/addressing):
//   * a wave instruction owns 64/G columns = ONE contiguous run of input and of output: coalesced 16-byte
//     loads -> a padded LDS strip -> G lanes per column, each a contiguous 1/G of it in registers (loading the
//     lane pieces directly as scattered 16-byte accesses ran at 50 %: TCP pending-stall 80 %);
//   * the running value is still accumulated IN THE REFERENCE'S ORDER, one row after the other: the G lanes of
//     a column take turns — G rounds, every lane re-runs its chain from the carry it holds and then takes its
//     left neighbour's final value (quad_perm DPP); lane t's carry is final after round t-1, so no selects.
//     (A parallel prefix scan moved 1 Simplex element in 63 000 by 2 % where 1 - Σ is small: rejected.)
//   * outputs are re-dealt through the same strip into whole coalesced 16-byte packs (rows_out = R-1 columns
//     of a wave instruction still form one aligned contiguous run).
// G = 4 for maps whose chain is one add per row (4 redundant adds per element), G = 2 for the Simplex inverse
// whose chain is the 5-op recurrence x_k = clamp(((1+ε) - Σ)·z_k/(1-2ε) - ε), Σ += x_k.
template <int G> __device__ __forceinline__ float quad_from_left(float v) {
  // value of the lane to the left inside the G-lane group (undefined for the group's first lane)
  if constexpr (G == 4) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x90, 0xF, 0xF, true));   // quad_perm [0,0,1,2]
  else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xA0, 0xF, 0xF, true));                   // quad_perm [0,0,2,2]
}
template <int G> __device__ __forceinline__ double quad_from_left(double v) {
  const int lane = threadIdx.x & 63;
  return __shfl(v, lane > 0 ? lane - 1 : 0, 64);
}

template <class T, bool LADJ> struct QSimplexFwd {      // simplex.jl:47-64 + :122-138
  static constexpr int G = 4, IN_LESS = 0, OUT_LESS = 1;
  static constexpr bool USES_LOGK = true, HAS_LADJ = LADJ;
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], int gl, const T* lk) const {
    using F = Fast<T>;
    const T e = Num<T>::eps, c2 = T(1) - 2 * e, E = T(1) + e;
    T carry = T(0);
#pragma unroll
    for (int t = 0; t < G - 1; ++t) {                                  // rounds 0..G-2: only the running sum
      T run = carry;
#pragma unroll
      for (int i = 0; i < RPL; ++i) run += x[i];
      const T bc = quad_from_left<G>(run);
      carry = gl == 0 ? T(0) : bc;
    }
    T lp = T(0), s = carry;                                            // final round: s = Σ_{j<k} x_j in the reference's order
    T Pp = T(1), mp = T(1), chk = T(0);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const T xk = x[i];
      const bool row0 = i == 0 && gl == 0;
      const T a = row0 ? xk * c2 + e : (xk + e) * c2;                 // :53 / :58
      const T dn = row0 ? T(1) : E - s;
      x[i] = F::log2(a * F::rcp(dn - a)) * Num<T>::log2 + lk[gl * RPL + i];   // logit(z) + log(K-k)
      if (LADJ) {
        // term_k = max(z,ε)·max(1-z,ε)·m, z = x_k/m, m = max(1-Σ,ε) (:130-135; m = 1 on row 1 since Σ = 0)
        //        = max(x_k, εm)·max(m - x_k, εm)/m ; two rows share one reciprocal and one logarithm (each term >= ε²)
        const bool rowK = i == RPL - 1 && gl == G - 1;                 // row K has no term
        const T m = d_max(T(1) - s, e), em = e * m;                    // :133
        const T P = d_max(xk, em) * d_max(m - xk, em);
        if constexpr (sizeof(T) == 8) {
          // Float64: a lean logarithm is ~40 operations — FOUR rows share one reciprocal and one logarithm (each term >= ε³ = 1e-47)
          Pp *= rowK ? T(1) : P;
          mp *= rowK ? T(1) : m;
          if ((i & 3) == 3 || i == RPL - 1) { lp += F::log2(Pp * F::rcp(mp)); Pp = T(1); mp = T(1); }
        } else {
          if (i & 1) lp += F::log2(Pp * F::rcp(mp * (rowK ? T(1) : m)) * (rowK ? T(1) : P));
          else { Pp = P; mp = m; }
        }
        if (i == RPL - 1) chk = rowK ? s : s + xk;                     // Σ over this lane's rows < K and everything above
      }
      s += xk;
    }
    // Julia's max(NaN, ε) is NaN (v_max drops it): a NaN among x_1..x_{K-1} makes the reference's log-det NaN
    return chk != chk ? chk : -lp * Num<T>::log2;
  }
};

template <class T, bool LADJ, int G_> struct QSimplexInv {      // simplex.jl:102-120 ; log-det = -logabsdetjac(b, x_out)
  static constexpr int G = G_, IN_LESS = 1, OUT_LESS = 0;
  static constexpr bool USES_LOGK = true, HAS_LADJ = LADJ;
  // FAST: clamp(v, 0, 1) as one v_med3_f32.  med3 does not keep a NaN (the reference's clamp does, and the NaN then
  // poisons Σ and every later row), so run() takes this path only for waves whose inputs are all finite.
  template <bool FAST> static __device__ __forceinline__ T cl01(T v) {
    if constexpr (FAST) return d_med3(v, T(0), T(1));
    else return d_clamp(v, T(0), T(1));
  }
  template <bool FAST, int RPL> static __device__ __forceinline__ T rounds(T (&x)[RPL], int gl) {
    using F = Fast<T>;
    const T e = Num<T>::eps, E = T(1) + e;
    const T e0 = e * (T(1) / (T(1) - 2 * e));
    T carry = T(0);
#pragma unroll
    for (int t = 0; t < G - 1; ++t) {                                  // rounds 0..G-2: only the recurrence Σ -> x_k -> Σ
      T s = carry;
#pragma unroll
      for (int i = 0; i < RPL; ++i) {
        const bool row0 = i == 0 && gl == 0;
        const T xi = row0 ? cl01<FAST>(x[i] - e0) : cl01<FAST>((E - s) * x[i] - e);
        s += xi;
      }
      const T bc = quad_from_left<G>(s);
      carry = gl == 0 ? T(0) : bc;
    }
    T lp = T(0), s = carry;
    T Pp = T(1), mp = T(1);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool row0 = i == 0 && gl == 0;
      const bool rowK = i == RPL - 1 && gl == G - 1;
      const T xi = row0 ? cl01<FAST>(x[i] - e0)                                   // :109
                        : cl01<FAST>((E - s) * x[i] - e);                        // :113
      if (LADJ) {
        // term_k = max(z,ε)·max(1-z,ε)·m, z = x_k/m, m = max(1-Σ,ε) (:130-135; m = 1 on row 1 since Σ = 0)
        //        = max(x_k, εm)·max(m - x_k, εm)/m ; two rows share one reciprocal and one logarithm (each term >= ε²)
        const T m = d_max(T(1) - s, e), em = e * m;
        const T P = d_max(xi, em) * d_max(m - xi, em);
        if constexpr (sizeof(T) == 8) {                                  // Float64: four rows per reciprocal and logarithm, as in QSimplexFwd
          Pp *= rowK ? T(1) : P;
          mp *= rowK ? T(1) : m;
          if ((i & 3) == 3 || i == RPL - 1) { lp += F::log2(Pp * F::rcp(mp)); Pp = T(1); mp = T(1); }
        } else {
          if (i & 1) lp += F::log2(Pp * F::rcp(mp * (rowK ? T(1) : m)) * (rowK ? T(1) : P));
          else { Pp = P; mp = m; }
        }
      }
      x[i] = rowK ? cl01<FAST>(T(1) - s) : xi;                                   // :116
      s += xi;
    }
    if (!FAST && s != s) return s;                                     // Julia's max(NaN, ε) is NaN: the log-det of a poisoned column is NaN
    return lp * Num<T>::log2;
  }
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], int gl, const T* lk) const {
    const T e = Num<T>::eps;
    const T inv12e = T(1) / (T(1) - 2 * e);
    // z_k = logistic(y_k - log(K-k)) with LogExpFunctions' exact 0/1 saturation
    T poison = T(0);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const T v = x[i] - lk[gl * RPL + i];
      const T z = f_logistic(v);
      poison = z * T(0) + poison;                                      // NaN iff some y_k is NaN (z is in [0, 1] otherwise)
      x[i] = z * inv12e;                                               // z_k / (1 - 2ε): the factor both :109 and :113 apply
    }
    if (__builtin_amdgcn_ballot_w64(poison != poison) == 0) return rounds<true, RPL>(x, gl);
    return rounds<false, RPL>(x, gl);
  }
};

template <class T> struct QOrderedFwd {                  // ordered.jl:36-49, :80
  static constexpr int G = 4, IN_LESS = 0, OUT_LESS = 0;
  static constexpr bool USES_LOGK = false, HAS_LADJ = true;
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], int gl, const T*) const {
    using F = Fast<T>;
    T l = T(0);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool row0 = i == 0 && gl == 0;
      l += row0 ? T(0) : x[i];                                         // logabsdetjac = Σ_{k>=2} x_k
      x[i] = row0 ? x[i] : F::exp(x[i]);                               // y_1 = x_1 ; y_k = y_{k-1} + exp(x_k)
    }
    T carry = T(0);                                                    // 0 + x_1 is exact: the chain starts like y_1 = x_1
#pragma unroll
    for (int r = 0; r < G - 1; ++r) {
      T run = carry;
#pragma unroll
      for (int i = 0; i < RPL; ++i) run += x[i];
      const T bc = quad_from_left<G>(run);
      carry = gl == 0 ? T(0) : bc;
    }
    T run = carry;
#pragma unroll
    for (int i = 0; i < RPL; ++i) { run += x[i]; x[i] = run; }
    return l;
  }
};

template <class T> struct QOrderedInv {                  // ordered.jl:63-77 ; interface.jl:276-281
  static constexpr int G = 4, IN_LESS = 0, OUT_LESS = 0;
  static constexpr bool USES_LOGK = false, HAS_LADJ = true;
  template <int RPL> __device__ __forceinline__ T run(T (&x)[RPL], int gl, const T*) const {
    using F = Fast<T>;
    const T left = quad_from_left<G>(x[RPL - 1]);                      // y of the row before my first one
    T l = T(0), prev = left;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool row0 = i == 0 && gl == 0;
      const T y = x[i];
      const T o = row0 ? y : F::log(y - prev);                         // x_1 = y_1 ; x_k = log(y_k - y_{k-1})
      l -= row0 ? T(0) : o;
      prev = y;
      x[i] = o;
    }
    return l;
  }
};

template <class T, int V, int NP, class Op, int UC, int WPB = 4, int H = 1>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 8))) void quad_stream_kernel(const Op op, const T* __restrict__ x, T* __restrict__ y,
                                                                                                           T* __restrict__ ladj_ps, int64_t batch,
                                                                                                           int accumulate, const BjxFin fin) {
  constexpr int G = Op::G;
  constexpr int RPL = NP * V;                        // rows per lane
  constexpr int R = G * RPL;                         // rows of the column frame
  constexpr int RI = R - Op::IN_LESS, RO = R - Op::OUT_LESS;
  constexpr int CPS = 64 / G;                        // columns per wave instruction
  constexpr int PITCH = R + V;                       // LDS pitch of a frame column: one pack of padding keeps the 16-byte accesses conflict-free
  constexpr int PPC = G * NP;                        // packs per frame column
  static_assert((CPS * RI) % V == 0 && (CPS * RO) % V == 0, "a wave instruction's run is whole packs");
  constexpr int NPI = CPS * RI / V, NPO = CPS * RO / V;          // packs of the input / output run
  constexpr int NLI = (NPI + 63) / 64, NLO = (NPO + 63) / 64;    // pack loads / stores per lane
  // The re-deal goes through the strip in H column slices (H = 1: the whole run at once).  A slice is CS columns:
  // its packs of the run are written, its lanes read their rows; 1/H of the LDS per wave, so long columns
  // (one lane per column, 64 rows: 17 KiB per wave unsliced) keep 4 waves per SIMD.
  constexpr int CS = CPS / H;
  static_assert(CPS % H == 0 && (CS * RI) % V == 0 && (CS * RO) % V == 0, "a slice is whole columns and whole packs");
  constexpr int NSI = CS * RI / V, NSO = CS * RO / V;            // packs per slice
  __shared__ __attribute__((aligned(16))) T strip[WPB][CS * PITCH];
  __shared__ __attribute__((aligned(16))) T lktab[Op::USES_LOGK ? R : V];
  __shared__ double red[WPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1), cg = lane / G;
  const int cs = cg % CS, hs = cg / CS;              // column inside its slice, slice of this lane
  const int64_t colb = ((int64_t)blockIdx.x * WPB + wave) * (CPS * UC);
  // coalesced loads: the columns of a wave instruction are one contiguous run
  Pack<T, V> raw[UC][NLI];
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t colw = colb + u * CPS;
    const int64_t left = batch - colw;
    const int64_t ne = left >= CPS ? (int64_t)CPS * RI : (left > 0 ? left * RI : 0);   // elements of this run inside the batch
#pragma unroll
    for (int q = 0; q < NLI; ++q) {
      const int pk = lane + 64 * q;
      const T* src = x + colw * RI + (int64_t)pk * V;
      if ((int64_t)(pk + 1) * V <= ne) raw[u][q] = load_pack<T, V, true>(src);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) raw[u][q].v[j] = (int64_t)pk * V + j < ne ? src[j] : T(0);
      }
    }
  }
  if (Op::USES_LOGK) {
    // log(K-k) per row (simplex.jl:35,41): precise logs, once per block
    for (int i = threadIdx.x; i < R; i += 64 * WPB) lktab[i] = i < R - 1 ? d_log(T(R - 1 - i)) : T(0);
    __syncthreads();
  }
  double acc = 0.0;
  T* st = strip[wave];
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    const int64_t colw = colb + u * CPS;
    const int64_t col = colw + cg;
    // ---- re-deal the run: lane (cg, gl) takes rows gl·RPL .. gl·RPL+RPL-1 of column cg
    T xv[RPL];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      __builtin_amdgcn_wave_barrier();                                 // the previous reads of the strip are done
      int dc = (lane * V) / RI, dr = (lane * V) % RI;                  // (column, row) of this lane's first element; advanced per load, no per-element division
#pragma unroll
      for (int q = 0; q < NLI; ++q) {
        const int pk = lane + 64 * q;
        if (64 * q + 63 >= h * NSI && 64 * q < (h + 1) * NSI) {        // compile time: this load holds packs of slice h
          if ((H == 1 || (pk >= h * NSI && pk < (h + 1) * NSI)) && (NPI % 64 == 0 || pk < NPI)) {
            if constexpr (Op::IN_LESS == 0) {
              *reinterpret_cast<typename Vec16<T>::type*>(st + (pk / PPC - h * CS) * PITCH + (pk % PPC) * V) = __builtin_bit_cast(typename Vec16<T>::type, raw[u][q]);
            } else {
              T* dst = st + (dc - h * CS) * PITCH + dr;
#pragma unroll
              for (int j = 0; j < V; ++j) dst[j + (dr + j >= RI ? PITCH - RI : 0)] = raw[u][q].v[j];
            }
          }
        }
        if constexpr (Op::IN_LESS != 0) {
          dc += (64 * V) / RI; dr += (64 * V) % RI;
          if (dr >= RI) { dr -= RI; ++dc; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (H == 1 || hs == h) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const typename Vec16<T>::type*>(st + cs * PITCH + (gl * NP + q) * V));
#pragma unroll
          for (int j = 0; j < V; ++j) xv[q * V + j] = pq.v[j];
        }
      }
    }
    if (Op::IN_LESS) { if (gl == G - 1) xv[RPL - 1] = T(0); }          // the frame's last row has no input
    const T l = op.template run<RPL>(xv, gl, lktab);
    if (Op::HAS_LADJ) {
      const T ls = group_sum_rt(l, G);
      if (col < batch && gl == 0) {
        if (ladj_ps) ladj_ps[col] = accumulate ? ladj_ps[col] + ls : ls;
        acc += (double)ls;
      }
    }
    if (y) {
      if (colw + CPS <= batch) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          __builtin_amdgcn_wave_barrier();                             // everybody has read its inputs / the previous slice has left
          if (H == 1 || hs == h) {
            if constexpr (Op::OUT_LESS == 0) {
#pragma unroll
              for (int q = 0; q < NP; ++q) {
                Pack<T, V> pq;
#pragma unroll
                for (int j = 0; j < V; ++j) pq.v[j] = xv[q * V + j];
                *reinterpret_cast<typename Vec16<T>::type*>(st + cs * PITCH + (gl * NP + q) * V) = __builtin_bit_cast(typename Vec16<T>::type, pq);
              }
            } else {
#pragma unroll
              for (int i = 0; i < RPL; ++i) { const int r = gl * RPL + i; if (r < RO) st[cs * RO + r] = xv[i]; }
            }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < NLO; ++q) {
            const int pk = lane + 64 * q;
            if (64 * q + 63 >= h * NSO && 64 * q < (h + 1) * NSO) {
              if ((H == 1 || (pk >= h * NSO && pk < (h + 1) * NSO)) && (NPO % 64 == 0 || pk < NPO)) {
                const T* src = Op::OUT_LESS == 0 ? st + (pk / PPC - h * CS) * PITCH + (pk % PPC) * V : st + (pk - h * NSO) * V;
                const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const typename Vec16<T>::type*>(src));
                store_pack<T, V, true>(y + colw * RO + (int64_t)pk * V, pq);
              }
            }
          }
        }
      } else if (col < batch) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) { const int r = gl * RPL + i; if (r < RO) y[col * RO + r] = xv[i]; }
      }
    }
  }
  block_publish_partial(acc, red, fin);      // the sentinel hand-off (one launch per call) or the two-pass partials
}

// launch for R = G·NP·V rows; returns BJX_ERR_UNSUPPORTED-free "not taken" (1) when the shape does not fit
template <class T, class Op>
int launch_quad_stream(bjx_ctx* ctx, const Op& op, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t R, int64_t batch, uint32_t flags, bool* taken) {
  constexpr int VW = Vec16<T>::N;
  constexpr int G = Op::G;
  *taken = false;
  static const int use_stream = getenv("BJX_SEQ_STREAM") ? atoi(getenv("BJX_SEQ_STREAM")) : 1;
  const int64_t np = R / (G * VW);
  constexpr int NPMAX = G >= 4 ? 8 : 16 / G;                 // four lanes per column: R <= 128 (Float32) / 64 (Float64), 32 rows per lane; fewer lanes: R <= 64 / 32
  if (!use_stream || batch <= 0 || R % (G * VW) != 0 || np < 1 || np > NPMAX || !bjx_aligned16(in) || (out && !bjx_aligned16(out))) return BJX_OK;
  if (np > 8 && np != 16) return BJX_OK;                     // np = 1 … 8 (every multiple of 16 rows up to 128 in Float32), 16
  *taken = true;
  const int64_t cpb = 4 * (64 / G);                          // columns per block: 4 waves x 64/G columns
  const int64_t grid = (batch + cpb - 1) / cpb;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  BjxFin fin;
  bool second = false;
  if (Op::HAS_LADJ) { int rc = bjx_make_fin(ctx, grid, ladj_sum, 0.0, 0, flags, &fin, &second); if (rc) return rc; }
  if (fin.counter) { fin.counter = nullptr; second = true; }      // short-lived blocks: the arrival-ticket form makes every wave wait for its own stores
  const int accum = (flags & BJX_ACCUMULATE) ? 1 : 0;
  // column slices of the re-deal: keep the wave's strip at <= ~9 KiB (4 waves per SIMD by LDS)
#define QS(NP_) do { constexpr size_t SB_ = (size_t)(64 / G) * (size_t)(G * NP_ * VW + VW) * sizeof(T);                                                        \
    constexpr int H_ = (SB_ > 9 * 1024 && (64 / G) % 2 == 0 && ((64 / G) / 2 * (G * NP_ * VW - Op::IN_LESS)) % VW == 0 && ((64 / G) / 2 * (G * NP_ * VW - Op::OUT_LESS)) % VW == 0) ? 2 : 1; \
    hipLaunchKernelGGL((quad_stream_kernel<T, VW, NP_, Op, 1, 4, H_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, op, in, out, ladj_ps, batch, accum, fin); } while (0)
  {
    BjxProf prof_(ctx);
    switch ((int)np) {
      case 1: QS(1); break;
      case 2: QS(2); break;
      case 3: if constexpr (NPMAX >= 3) QS(3); break;
      case 4: if constexpr (NPMAX >= 4) QS(4); break;
      case 5: if constexpr (NPMAX >= 8) QS(5); break;
      case 6: if constexpr (NPMAX >= 8) QS(6); break;
      case 7: if constexpr (NPMAX >= 8) QS(7); break;
      case 8: if constexpr (NPMAX >= 8) QS(8); break;
      case 16: if constexpr (NPMAX >= 16) QS(16); break;
    }
  }
#undef QS
  BJX_CHECK_LAUNCH(ctx);
  if (ladj_sum) {
    if (Op::HAS_LADJ) return second ? bjx_launch_finalize(ctx, (int)grid, ladj_sum, 0.0, 0, 0.0, flags) : BJX_OK;
    if (!(flags & BJX_ACCUMULATE)) BJX_HIP(ctx, hipMemsetAsync(ladj_sum, 0, sizeof(double), ctx->stream));
  }
  return BJX_OK;
}

// ---- the re-deal of quad_stream_kernel as reusable pieces (kernels with more than one input run)
template <class T, int V, int NP, int G> struct QuadStrip {
  static constexpr int RPL = NP * V, R = G * RPL, CPS = 64 / G, PITCH = R + V, PPC = G * NP;
  static constexpr int npk(int less) { return CPS * (R - less) / V; }
  static constexpr int nl(int less) { return (npk(less) + 63) / 64; }
  static constexpr int NLMAX = (CPS * R / V + 63) / 64;
  // coalesced loads of one wave instruction's run of (R - LESS)-row columns starting at column colw
  template <int LESS> static __device__ __forceinline__ void load(Pack<T, V> (&raw)[NLMAX], const T* __restrict__ x, int64_t colw, int64_t batch, int lane) {
    constexpr int RI = R - LESS;
    const int64_t left = batch - colw;
    const int64_t ne = left >= CPS ? (int64_t)CPS * RI : (left > 0 ? left * RI : 0);
#pragma unroll
    for (int q = 0; q < nl(LESS); ++q) {
      const int pk = lane + 64 * q;
      const T* src = x + colw * RI + (int64_t)pk * V;
      if ((int64_t)(pk + 1) * V <= ne) raw[q] = load_pack<T, V, true>(src);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) raw[q].v[j] = (int64_t)pk * V + j < ne ? src[j] : T(0);
      }
    }
  }
  // run -> rows gl·RPL .. gl·RPL+RPL-1 of column cg (the frame's missing last row reads as 0)
  template <int LESS> static __device__ __forceinline__ void deal_in(T* st, const Pack<T, V> (&raw)[NLMAX], T (&xv)[RPL], int lane) {
    constexpr int RI = R - LESS;
    const int gl = lane & (G - 1), cg = lane / G;
    __builtin_amdgcn_wave_barrier();
    int dc = (lane * V) / RI, dr = (lane * V) % RI;                      // advanced per load: no per-element division
#pragma unroll
    for (int q = 0; q < nl(LESS); ++q) {
      const int pk = lane + 64 * q;
      if (npk(LESS) % 64 == 0 || pk < npk(LESS)) {
        if constexpr (LESS == 0) {
          *reinterpret_cast<typename Vec16<T>::type*>(st + (pk / PPC) * PITCH + (pk % PPC) * V) = __builtin_bit_cast(typename Vec16<T>::type, raw[q]);
        } else {
          T* dst = st + dc * PITCH + dr;
#pragma unroll
          for (int j = 0; j < V; ++j) dst[j + (dr + j >= RI ? PITCH - RI : 0)] = raw[q].v[j];
        }
      }
      if constexpr (LESS != 0) {
        dc += (64 * V) / RI; dr += (64 * V) % RI;
        if (dr >= RI) { dr -= RI; ++dc; }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const typename Vec16<T>::type*>(st + cg * PITCH + (gl * NP + q) * V));
#pragma unroll
      for (int j = 0; j < V; ++j) xv[q * V + j] = pq.v[j];
    }
    if (LESS) { if (gl == G - 1) xv[RPL - 1] = T(0); }
  }
  template <int LESS> static __device__ __forceinline__ void deal_out(T* st, const T (&xv)[RPL], T* __restrict__ y, int64_t colw, int64_t batch, int lane) {
    constexpr int RO = R - LESS;
    const int gl = lane & (G - 1), cg = lane / G;
    if (colw + CPS <= batch) {
      __builtin_amdgcn_wave_barrier();
      if constexpr (LESS == 0) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          Pack<T, V> pq;
#pragma unroll
          for (int j = 0; j < V; ++j) pq.v[j] = xv[q * V + j];
          *reinterpret_cast<typename Vec16<T>::type*>(st + cg * PITCH + (gl * NP + q) * V) = __builtin_bit_cast(typename Vec16<T>::type, pq);
        }
      } else {
#pragma unroll
        for (int i = 0; i < RPL; ++i) { const int r = gl * RPL + i; if (r < RO) st[cg * RO + r] = xv[i]; }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < nl(LESS); ++q) {
        const int pk = lane + 64 * q;
        if (npk(LESS) % 64 == 0 || pk < npk(LESS)) {
          const T* src = LESS == 0 ? st + (pk / PPC) * PITCH + (pk % PPC) * V : st + pk * V;
          const Pack<T, V> pq = __builtin_bit_cast(Pack<T, V>, *reinterpret_cast<const typename Vec16<T>::type*>(src));
          store_pack<T, V, true>(y + colw * RO + (int64_t)pk * V, pq);
        }
      }
    } else if (colw + cg < batch) {
#pragma unroll
      for (int i = 0; i < RPL; ++i) { const int r = gl * RPL + i; if (r < RO) y[(colw + cg) * RO + r] = xv[i]; }
    }
  }
};
template <int G> __device__ __forceinline__ float quad_from_right(float v) {
  if constexpr (G == 4) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF9, 0xF, 0xF, true));   // quad_perm [1,2,3,3]
  else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF5, 0xF, 0xF, true));                   // quad_perm [1,1,3,3]
}
template <int G> __device__ __forceinline__ double quad_from_right(double v) {
  const int lane = threadIdx.x & 63;
  return __shfl(v, lane < 63 ? lane + 1 : 63, 64);
}
// exclusive scans over the G lanes of a column group (ascending / descending lane order)
template <int G, class T> __device__ __forceinline__ T group_excl_up(T v, int gl) {
  T inc = v;
#pragma unroll
  for (int d = 1; d < G; d <<= 1) { const T t = __shfl_up(inc, d, G); if (gl >= d) inc += t; }
  return inc - v;
}
template <int G, class T> __device__ __forceinline__ T group_excl_down(T v, int gl) {
  T inc = v;
#pragma unroll
  for (int d = 1; d < G; d <<= 1) { const T t = __shfl_down(inc, d, G); if (gl + d < G) inc += t; }
  return inc - v;
}

// ------------------------------------------------------------------ SimplexBijector pullbacks, streaming kernel
// Same math as simplex_vjp_kernel below/above (O(K) reverse sweeps of simplex.jl:47-64, :102-120, :122-138) in the
// G-lanes-per-column register layout: the two whole-column LDS tiles of that kernel allow ONE wave per SIMD (30-36 %
// of the HBM roofline).  Forward map: s_k and the suffix sums of the adjoint are plain scans (local + group scan —
// a pullback has no reference summation order to keep).  Inverse map: the clamped recurrence is re-run in the
// reference's order by the take-turns rounds of quad_stream_kernel, and the adjoint of Σ is the affine recurrence
// sb <- B_k sb + A_k run the same way from the last lane down.
template <class T, int V, int NP, int G, bool INV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void simplex_vjp_stream_kernel(const T* __restrict__ in, const T* __restrict__ out_bar,
                                                                                                              const T* __restrict__ ladj_bar, T* __restrict__ in_bar,
                                                                                                              int64_t batch) {
  using F = Fast<T>;
  using Q = QuadStrip<T, V, NP, G>;
  constexpr int RPL = Q::RPL, R = Q::R, CPS = Q::CPS;
  __shared__ __attribute__((aligned(16))) T strip[4][CPS * Q::PITCH];
  __shared__ __attribute__((aligned(16))) T lktab[R];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1), cg = lane / G;
  const int64_t colw = ((int64_t)blockIdx.x * 4 + wave) * CPS;
  const int64_t col = colw + cg;
  Pack<T, V> raw_a[Q::NLMAX], raw_g[Q::NLMAX];
  Q::template load<INV ? 1 : 0>(raw_a, in, colw, batch, lane);
  Q::template load<INV ? 0 : 1>(raw_g, out_bar, colw, batch, lane);
  const T lb = (ladj_bar && col < batch) ? ladj_bar[col] : T(0);
  if (INV) {
    if ((int)threadIdx.x < R) lktab[threadIdx.x] = (int)threadIdx.x < R - 1 ? d_log(T(R - 1 - (int)threadIdx.x)) : T(0);
    __syncthreads();
  }
  T* st = strip[wave];
  T a[RPL], g[RPL];
  Q::template deal_in<INV ? 1 : 0>(st, raw_a, a, lane);
  Q::template deal_in<INV ? 0 : 1>(st, raw_g, g, lane);
  const T e = Num<T>::eps;
  const T c = T(1) / (T(1) - 2 * e), E = T(1) + e, c2 = T(1) - 2 * e;
  if constexpr (!INV) {
    // ---- pullback of x -> (y, logabsdetjac): a = x (K rows), g = ȳ (K-1 rows, frame row K-1 reads 0)
    T tot = T(0);
#pragma unroll
    for (int i = 0; i < RPL; ++i) tot += a[i];
    T s = group_excl_up<G>(tot, gl);                                   // Σ of the rows before my first one
    T as_[RPL];
    T asum = T(0);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool row0 = i == 0 && gl == 0;
      const bool rowK = i == RPL - 1 && gl == G - 1;
      const T xk = a[i];
      T dtdx, dtds;
      simplex_t_partials<T>(xk, s, false, dtdx, dtds);
      T dtdx0, dtds0;
      if (i == 0) { simplex_t_partials<T>(xk, s, true, dtdx0, dtds0); dtdx = row0 ? dtdx0 : dtdx; dtds = row0 ? T(0) : dtds; }
      const T rd = row0 ? T(1) : F::rcp(E - s);
      const T an = row0 ? xk * c2 + e : (xk + e) * c2;               // zf = an·rd
      const T zf = an * rd;
      const T zfb = g[i] * F::rcp(zf * (T(1) - zf));
      const T ax = zfb * c2 * rd - lb * dtdx;
      const T asv = row0 ? T(0) : zfb * an * rd * rd - lb * dtds;
      a[i] = rowK ? T(0) : ax;
      as_[i] = rowK ? T(0) : asv;
      asum += as_[i];
      s += xk;
    }
    T sfx = group_excl_down<G>(asum, gl);                              // Σ as over the rows after my last one
#pragma unroll
    for (int i = RPL - 1; i >= 0; --i) { a[i] += sfx; sfx += as_[i]; }
    Q::template deal_out<0>(st, a, in_bar, colw, batch, lane);
  } else {
    // ---- pullback of y -> (x, logabsdetjac): a = y (K-1 rows), g = x̄ (K rows)
#pragma unroll
    for (int i = 0; i < RPL; ++i) a[i] = f_logistic(a[i] - lktab[gl * RPL + i]) * c;   // c·z_k
    T carry = T(0);
#pragma unroll
    for (int t = 0; t < G - 1; ++t) {
      T s = carry;
#pragma unroll
      for (int i = 0; i < RPL; ++i) {
        const bool row0 = i == 0 && gl == 0;
        const T xi = row0 ? d_clamp(a[i] - e * c, T(0), T(1)) : d_clamp((E - s) * a[i] - e, T(0), T(1));
        s += xi;
      }
      const T bc = quad_from_left<G>(s);
      carry = gl == 0 ? T(0) : bc;
    }
    T xk[RPL], A[RPL], B[RPL];
    T s = carry;
    T sb0 = T(0);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const bool row0 = i == 0 && gl == 0;
      const bool rowK = i == RPL - 1 && gl == G - 1;
      const T xi = row0 ? d_clamp(a[i] - e * c, T(0), T(1)) : d_clamp((E - s) * a[i] - e, T(0), T(1));
      T dtdx, dtds;
      simplex_t_partials<T>(xi, s, false, dtdx, dtds);
      if (i == 0) { T d0, d1; simplex_t_partials<T>(xi, s, true, d0, d1); dtdx = row0 ? d0 : dtdx; dtds = row0 ? T(0) : dtds; }
      const bool gate = xi > T(0) && xi < T(1);
      const T rc = (E - s) * c;
      const T z = row0 ? xi * c2 + e : (xi + e) * F::rcp(rc);
      const T cz = (gate && !row0) ? c * z : T(0);
      // sb_next = B sb + A ;  ȳ = gate (g + sb_in + lb dtdx) w,  w = rc z (1-z)  (row 0: c z (1-z))
      B[i] = rowK ? T(1) : T(1) - cz;
      A[i] = rowK ? T(0) : lb * dtds - cz * (g[i] + lb * dtdx);
      xk[i] = gate ? (row0 ? c : rc) * z * (T(1) - z) : T(0);          // w_k (0 where the clamp is active)
      if (rowK) { const T last = T(1) - s; sb0 = (last > T(0) && last < T(1)) ? -g[i] : T(0); }
      g[i] = g[i] + lb * dtdx;                                         // g + lb dtdx
      s += xi;
    }
    // adjoint of Σ, from the last row down: the lanes take turns from the right
    T cin = gl == G - 1 ? sb0 : T(0);
#pragma unroll
    for (int t = 0; t < G - 1; ++t) {
      T sb = cin;
#pragma unroll
      for (int i = RPL - 1; i >= 0; --i) sb = B[i] * sb + A[i];
      const T bc = quad_from_right<G>(sb);
      cin = gl == G - 1 ? sb0 : bc;
    }
    T sb = cin;
#pragma unroll
    for (int i = RPL - 1; i >= 0; --i) {
      a[i] = (g[i] + sb) * xk[i];
      sb = B[i] * sb + A[i];
    }
    Q::template deal_out<1>(st, a, in_bar, colw, batch, lane);
  }
}

template <class T, int G>
int launch_simplex_vjp_stream(bjx_ctx* ctx, int inverse, const T* in, const T* out_bar, const T* ladj_bar, T* in_bar, int64_t K, int64_t batch, bool* taken) {
  constexpr int VW = Vec16<T>::N;
  *taken = false;
  static const int use_stream = getenv("BJX_SIMPLEX_VJP_STREAM") ? atoi(getenv("BJX_SIMPLEX_VJP_STREAM")) : 1;
  const int64_t np = K / (G * VW);
  constexpr int NPMAX = G >= 4 ? 8 : 16 / G;                  // four lanes per column: K <= 128 (Float32) / 64 (Float64)
  if (!use_stream || batch <= 0 || K % (G * VW) != 0 || np < 1 || np > NPMAX || !bjx_aligned16(in) || !bjx_aligned16(out_bar) || !bjx_aligned16(in_bar)) return BJX_OK;
  *taken = true;
  const int64_t cpb = 4 * (64 / G);
  const int64_t grid = (batch + cpb - 1) / cpb;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
#define SVS(NP_, I_) hipLaunchKernelGGL((simplex_vjp_stream_kernel<T, VW, NP_, G, I_>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, in, out_bar, ladj_bar, in_bar, batch)
#define SVS_I(NP_) do { if (inverse) SVS(NP_, true); else SVS(NP_, false); } while (0)
  {
    BjxProf prof_(ctx);
    switch ((int)np) {
      case 1: SVS_I(1); break;
      case 2: SVS_I(2); break;
      case 3: if constexpr (NPMAX >= 3) SVS_I(3); break;
      case 4: if constexpr (NPMAX >= 4) SVS_I(4); break;
      case 5: if constexpr (NPMAX >= 5) SVS_I(5); break;
      case 6: if constexpr (NPMAX >= 6) SVS_I(6); break;
      case 7: if constexpr (NPMAX >= 7) SVS_I(7); break;
      case 8: if constexpr (NPMAX >= 8) SVS_I(8); break;
    }
  }
#undef SVS_I
#undef SVS
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}

template <class T>
int ordered_impl(bjx_ctx* ctx, int inverse, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags,
                 int64_t ld_in = 0, int64_t ld_out = 0) {
  const bool strided = (ld_in && ld_in != dim) || (ld_out && ld_out != dim);
  if (strided) {
    if (!inverse) return launch_seq<T>(ctx, OrderedFwd<T>{}, in, out, ladj_ps, ladj_sum, dim, dim, batch, 0, flags, ld_in, ld_out);
    return launch_seq<T>(ctx, OrderedInv<T>{}, in, out, ladj_ps, ladj_sum, dim, dim, batch, 0, flags, ld_in, ld_out);
  }
  {
    bool taken = false;                                                // 1 ... 7 rows: lane = column in registers (bjx_tiny.hip)
    const int rc = bjx_seq_tiny(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, inverse ? BJX_TALL_ORDERED_INV : BJX_TALL_ORDERED_FWD, in, out, ladj_ps, ladj_sum, dim,
                                batch, flags, &taken);
    if (rc || taken) return rc;
  }
  {
    bool taken = false;
    const int rc = !inverse ? launch_quad_stream<T>(ctx, QOrderedFwd<T>{}, in, out, ladj_ps, ladj_sum, dim, batch, flags, &taken)
                            : launch_quad_stream<T>(ctx, QOrderedInv<T>{}, in, out, ladj_ps, ladj_sum, dim, batch, flags, &taken);
    if (rc || taken) return rc;
  }
  {
    bool taken = false;                                                // taller than the quad frames: G lanes per column (bjx_tall.hip)
    const int rc = bjx_tall_stream(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, inverse ? BJX_TALL_ORDERED_INV : BJX_TALL_ORDERED_FWD, in, out, ladj_ps,
                                   ladj_sum, dim, dim, batch, flags, &taken);
    if (rc || taken) return rc;
  }
  if (!inverse) return launch_seq<T>(ctx, OrderedFwd<T>{}, in, out, ladj_ps, ladj_sum, dim, dim, batch, 0, flags);
  return launch_seq<T>(ctx, OrderedInv<T>{}, in, out, ladj_ps, ladj_sum, dim, dim, batch, 0, flags);
}

template <class T>
int simplex_impl(bjx_ctx* ctx, int inverse, const T* in, T* out, T* ladj_ps, double* ladj_sum, int64_t K, int64_t batch, uint32_t flags,
                 int64_t ld_in = 0, int64_t ld_out = 0) {
  const bool want = ladj_ps || ladj_sum;
  const int nlk = (int)(K - 1);
  const int64_t ri = inverse ? K - 1 : K, ro = inverse ? K : K - 1;
  const bool strided = (ld_in && ld_in != ri) || (ld_out && ld_out != ro);
  if (!strided) {
    bool taken = false;
    int rc = bjx_seq_tiny(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, inverse ? BJX_TALL_SIMPLEX_INV : BJX_TALL_SIMPLEX_FWD, in, out, ladj_ps, ladj_sum, K, batch,
                          flags, &taken);                              // 2 ... 7 rows: lane = column in registers (bjx_tiny.hip)
    if (rc || taken) return rc;
    if (!inverse) rc = want ? launch_quad_stream<T>(ctx, QSimplexFwd<T, true>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken)
                            : launch_quad_stream<T>(ctx, QSimplexFwd<T, false>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken);
    else {
      static const int g_env = 2;
      const int g4 = g_env == 4;
      constexpr int VWq = Vec16<T>::N;
      // BJX_SIMPLEX_INV_G=1: one lane per column — the clamped recurrence runs once per element (19 % fewer VALU
      // instructions than two lanes per column), but 64 rows per lane spill next to the two clamp paths: measured
      // equal (single path) to 25 % slower, so two lanes per column stay the default
      if (g_env == 1)
        rc = want ? launch_quad_stream<T>(ctx, QSimplexInv<T, true, 1>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken)
                  : launch_quad_stream<T>(ctx, QSimplexInv<T, false, 1>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken);
      if (rc || taken) return rc;
      if (g4 || K % (2 * VWq) != 0 || K / (2 * VWq) > 8 || (K / (2 * VWq) > 4 && K / (2 * VWq) != 8))
        rc = want ? launch_quad_stream<T>(ctx, QSimplexInv<T, true, 4>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken)
                  : launch_quad_stream<T>(ctx, QSimplexInv<T, false, 4>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken);
      if (!rc && !taken)
        rc = want ? launch_quad_stream<T>(ctx, QSimplexInv<T, true, 2>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken)
                  : launch_quad_stream<T>(ctx, QSimplexInv<T, false, 2>{}, in, out, ladj_ps, ladj_sum, K, batch, flags, &taken);
    }
    if (rc || taken) return rc;
    rc = bjx_tall_stream(ctx, sizeof(T) == 4 ? BJX_F32 : BJX_F64, inverse ? BJX_TALL_SIMPLEX_INV : BJX_TALL_SIMPLEX_FWD, in, out, ladj_ps, ladj_sum,
                         ri, ro, batch, flags, &taken);              // taller than the quad frames: G lanes per column (bjx_tall.hip)
    if (rc || taken) return rc;
  }
  if (!inverse) {
    if (want) { SimplexFwd<T, true> op; op.K = K; return launch_seq<T>(ctx, op, in, out, ladj_ps, ladj_sum, K, K - 1, batch, nlk, flags, ld_in, ld_out); }
    SimplexFwd<T, false> op; op.K = K;
    return launch_seq<T>(ctx, op, in, out, ladj_ps, ladj_sum, K, K - 1, batch, nlk, flags, ld_in, ld_out);
  }
  if (want) { SimplexInv<T, true> op; op.K = K; return launch_seq<T>(ctx, op, in, out, ladj_ps, ladj_sum, K - 1, K, batch, nlk, flags, ld_in, ld_out); }
  SimplexInv<T, false> op; op.K = K;
  return launch_seq<T>(ctx, op, in, out, ladj_ps, ladj_sum, K - 1, K, batch, nlk, flags, ld_in, ld_out);
}
}  // namespace

BJX_API int bjx_ordered(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum,
                        int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_ordered: input must not be empty (ordered.jl:26)");
  BJX_REQUIRE(ctx, (in && out) || batch == 0, BJX_ERR_ARG, "bjx_ordered: null pointer");
  if (dt == BJX_F32) return ordered_impl<float>(ctx, inverse, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags);
  if (dt == BJX_F64) return ordered_impl<double>(ctx, inverse, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_ordered: bad dtype %d", (int)dt);
}

BJX_API int bjx_simplex(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, void* out, void* ladj_ps, double* ladj_sum,
                        int64_t K, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, K > 1, BJX_ERR_SHAPE, "bjx_simplex: x needs to be of length greater than 1 (simplex.jl:30), got K=%lld", (long long)K);
  BJX_REQUIRE(ctx, batch >= 0, BJX_ERR_SHAPE, "bjx_simplex: negative batch");
  BJX_REQUIRE(ctx, in || batch == 0, BJX_ERR_ARG, "bjx_simplex: null input");
  BJX_REQUIRE(ctx, out || !inverse || batch == 0, BJX_ERR_ARG, "bjx_simplex: the inverse needs an output buffer");
  if (dt == BJX_F32) return simplex_impl<float>(ctx, inverse, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, K, batch, flags);
  if (dt == BJX_F64) return simplex_impl<double>(ctx, inverse, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, K, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_simplex: bad dtype %d", (int)dt);
}

namespace {
// Small K: one lane per sample (cf. chol_lane_kernel).  Two odd-pitch tiles: y (+ K words of scratch) and ΔW.  Column by
// column: forward sweep z = tanh y (kept in y's place), logcosh (scratch), log_remainder; reverse sweep (corr.jl:437-447)
// with (1/z − z)·W[i,j] written as (1 − z²)·exp(log_remainder) — the same value, and finite at z = 0.
template <class T, int V>
__global__ __launch_bounds__(64) void chol_inv_vjp_lane_kernel(const T* __restrict__ y, const T* __restrict__ Wbar, const T* __restrict__ lbar,
                                                               T* __restrict__ ybar, int K, int Py, int Pw, int lower, int64_t batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using M = LinkMath<T>;
  T* ty = reinterpret_cast<T*>(smem);
  T* tw = ty + (size_t)64 * Py;
  const int lane = threadIdx.x;
  const int KK = K * K, nv = K * (K - 1) / 2;
  for (int64_t s0 = (int64_t)blockIdx.x * 64; s0 < batch; s0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - s0) < 64 ? (batch - s0) : 64);
    tile_stage_in<T, V>(ty, y + s0 * nv, nv, Py, ncols, lane);
    tile_stage_in<T, V>(tw, Wbar + s0 * KK, KK, Pw, ncols, lane);
    tile_sync();
    if (lane < ncols) {
      T* my = ty + lane * Py;
      T* sc = my + nv;
      const T* dw = tw + lane * Pw;
      const T lb = lbar ? lbar[s0 + lane] : T(0);
      for (int j = 1; j < K; ++j) {
        T* yj = my + j * (j - 1) / 2;
        T lr = T(0);
        for (int i = 0; i < j; ++i) {
          T z, lc;
          M::tanh_lc(yj[i], z, lc);
          yj[i] = z;
          sc[i] = lc;
          lr -= lc;
        }
        T dlr = M::exp(lr) * dw[j * K + j] + T(2) * lb;                    // :438
        for (int i = j - 1; i >= 0; --i) {
          lr += sc[i];                                                     // log_remainder BEFORE entry i
          const T E = M::exp(lr);
          const T z = yj[i];
          const T d = lower ? dw[i * K + j] : dw[j * K + i];
          const T EdW = E * d;
          yj[i] = (T(1) - z * z) * EdW - z * dlr;                           // :441-443
          dlr += lb + z * EdW;                                             // :444
        }
      }
    }
    tile_sync();
    tile_stage_out<T, V>(ty, ybar + s0 * nv, nv, Py, ncols, lane);
    tile_sync();
  }
}
}  // namespace

namespace {
template <class T>
int chol_inv_vjp_impl(bjx_ctx* ctx, int uplo, const T* y, const T* Wbar, const T* lbar, T* ybar, int64_t K, int64_t batch) {
  if (batch == 0 || K < 2) return BJX_OK;
  const int lower = (uplo == 'L') ? 1 : 0;
  const int64_t nv = K * (K - 1) / 2;
  constexpr int VW = Vec16<T>::N;
  {
    static const int lane_max = getenv("BJX_CHOL_LANE_MAX") ? atoi(getenv("BJX_CHOL_LANE_MAX")) : 11;
    const int64_t Py = (nv + K) | 1, Pw = (K * K) | 1;
    const size_t smem_l = (size_t)64 * (Py + Pw) * sizeof(T);
    if (K <= lane_max && smem_l <= 56 * 1024) {
      const int64_t tiles = (batch + 63) / 64;
      const int64_t cap = (int64_t)ctx->num_cu * 32;
      const int grid_l = (int)(tiles < cap ? tiles : cap);
      const bool vec = bjx_aligned16(y) && bjx_aligned16(ybar) && bjx_aligned16(Wbar);
      BjxProf prof_(ctx);
      if (vec) hipLaunchKernelGGL((chol_inv_vjp_lane_kernel<T, VW>), dim3(grid_l), dim3(64), smem_l, ctx->stream, y, Wbar, lbar, ybar, (int)K, (int)Py, (int)Pw, lower, batch);
      else hipLaunchKernelGGL((chol_inv_vjp_lane_kernel<T, 1>), dim3(grid_l), dim3(64), smem_l, ctx->stream, y, Wbar, lbar, ybar, (int)K, (int)Py, (int)Pw, lower, batch);
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
  }
  const bool v_ok = bjx_aligned16(y) && bjx_aligned16(ybar) && nv % VW == 0;
  const int ch = (int)((nv + 63) / 64);
  int chv, vv;
  if (v_ok) { vv = VW; const int need = (ch + VW - 1) / VW; chv = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16; }
  else { vv = 1; chv = ch <= 2 ? 2 : ch <= 8 ? 8 : ch <= 16 ? 16 : 32; }
  const int CHn = chv * vv;
  int64_t tile_words = (K * K + 3) / 4 * 4;
  if (tile_words < (int64_t)64 * (CHn + vv)) tile_words = (int64_t)64 * (CHn + vv);
  const size_t tile_bytes = (size_t)CHOL_WPB * tile_words * sizeof(T);
  BJX_REQUIRE(ctx, CHn <= 32 && nv <= 64 * 32 && tile_bytes <= BJX_LDS_MAX, BJX_ERR_UNSUPPORTED,
              "bjx_vec_cholesky_inv_vjp: K = %lld is too large for the LDS tile kernel", (long long)K);
  const int64_t grid = (batch + CHOL_WPB - 1) / CHOL_WPB;
  BJX_REQUIRE(ctx, grid < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
#define CVJP_K(V_, CHV_, L_) bjx_allow_big_lds(chol_inv_vjp_kernel<T, V_, CHV_, L_>, tile_bytes); hipLaunchKernelGGL((chol_inv_vjp_kernel<T, V_, CHV_, L_>), dim3((unsigned)grid), dim3(64 * CHOL_WPB), tile_bytes, ctx->stream, y, Wbar, lbar, ybar, (int)K, (int)tile_words, batch)
#define CVJP_L(V_, CHV_) do { if (lower) { CVJP_K(V_, CHV_, true); } else { CVJP_K(V_, CHV_, false); } } while (0)
  {
    BjxProf prof_(ctx);
    if (vv == VW) {
      if (chv == 1) CVJP_L(VW, 1); else if (chv == 2) CVJP_L(VW, 2); else if (chv == 4) CVJP_L(VW, 4);
      else if (chv == 8 && VW * 8 <= 32) CVJP_L(VW, (VW * 8 <= 32 ? 8 : 1));
      else if (VW == 2 && chv == 8) CVJP_L(VW, 8); else CVJP_L(VW, (VW == 2 ? 16 : 1));
    } else {
      if (chv == 2) CVJP_L(1, 2); else if (chv == 8) CVJP_L(1, 8); else if (chv == 16) CVJP_L(1, 16); else CVJP_L(1, 32);
    }
  }
#undef CVJP_L
#undef CVJP_K
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

namespace {
// ------------------------------------------------------------------ pullback of _link_chol_lkj_from_upper / _from_lower
// ext/BijectorsChainRulesCoreExt.jl:199-254 / :256-311.  The rule lives on the constraint manifold (unit-norm
// columns: the strict triangle is free, the diagonal follows), ΔW[j,j] = 0.  Per column j, descending i = j-1..2:
//   tmp = sqrt(W[j,j]² + Σ_{i'>=i} W[i',j]²),  p = W[i,j]/tmp,  Δp = Δz/(1-p²) - Δtmp·tmp·p/sqrt(1-p²),
//   ΔW[i,j] = Δp/tmp,  Δtmp <- -Δp·W[i,j]/tmp² + Δtmp·sqrt(1-p²);   ΔW[1,j] = Δz/(1-W[1,j]²) - Δtmp·W[1,j]/sqrt(1-W[1,j]²)
// ONE WAVE per sample, LANE = COLUMN: W is staged into a [K][K+1] LDS tile (coalesced 16-byte loads; the odd pitch
// makes both the :U column walk and the :L row walk conflict-free), Δz into a flat strip; every lane walks its
// column from the diagonal up, writes ΔW in place, zeroes the diagonal and the other triangle (the reference leaves
// them undefined), and the tile leaves as coalesced 16-byte stores.  Lanes of short columns idle (triangular work:
// ~50 % lane utilisation).  Measured alternative (round 2, removed again): TWO samples per wave, lane l of a half-wave
// walking the column pair (K-1-l, l) — K-1 rows for every lane — on a triangle-only tile (16 KiB per sample): 0.90 ms
// against 0.66 ms for this kernel at K = 64 x 2^16.  PMC of this kernel: 3 130 VALU instructions per sample of which only
// ~1 500 are the walk; the rest is the element-wise LDS staging of W (odd pitch: four ds_write_b32 + index arithmetic
// per 16-byte pack) and the way out.  Balancing the walk therefore removes at most a quarter of the instructions and
// pays for it with occupancy (4 waves of two samples per CU instead of 6 of one).  What would help is a staging that
// keeps 16-byte LDS accesses (a swizzled pitch-64 tile), not a different walk.
// SWZ (K a power of two, 16-byte packs): the tile keeps W's own pitch K and column c is XOR-swizzled by c,
//   W[r, c] at c·K + (r ^ c):  a 16-byte pack of W stays one aligned 16-byte LDS access (its V words permuted by c mod V),
// and both walks stay conflict-free (:U lanes = columns at a fixed row: banks (r ^ c) are distinct; :L lanes along a row of
// the tile).  With the odd pitch every pack was V scalar LDS writes with their index arithmetic on the way in and V
// scalar reads on the way out — PMC: half of the kernel's 3 130 VALU instructions per sample — plus a zero-fill pass
// over the other triangle, which is now a mask on the way out.
template <class T, int V> __device__ __forceinline__ Pack<T, V> swz_permute(Pack<T, V> a, int lo) {       // b[m] = a[m ^ lo], lo < V
  if constexpr (V >= 2) {
    const bool s1 = lo & 1;
#pragma unroll
    for (int m = 0; m < V; m += 2) { const T x = a.v[m], y = a.v[m + 1]; a.v[m] = s1 ? y : x; a.v[m + 1] = s1 ? x : y; }
  }
  if constexpr (V >= 4) {
    const bool s2 = lo & 2;
#pragma unroll
    for (int m = 0; m < 2; ++m) { const T x = a.v[m], y = a.v[m + 2]; a.v[m] = s2 ? y : x; a.v[m + 2] = s2 ? x : y; }
  }
  return a;
}
template <class T, int V, bool LOWER, bool SWZ>
__global__ __launch_bounds__(64) void chol_fwd_vjp_kernel(const T* __restrict__ W, const T* __restrict__ ybar, T* __restrict__ Wbar, int K, int64_t batch) {
  using F = Fast<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = SWZ ? K : K + 1;
  T* tile = reinterpret_cast<T*>(smem);
  const int nv = K * (K - 1) / 2;
  // (Δz in the spare triangle of the swizzled tile — 16 KiB per sample instead of 24 — was measured: 0.60 vs 0.59 ms, the
  // scatter costs what the occupancy gives)
  T* dz = tile + (((size_t)K * P + 3) / 4) * 4;
  const int lane = threadIdx.x;
  const int64_t s = blockIdx.x;
  const T* Ws = W + s * (int64_t)K * K;
  const T* zs = ybar + s * (int64_t)nv;
  const int ne = K * K;
  {
    int e = lane * V, c = e / K, r = e % K;
    const int dc = (64 * V) / K, dr = (64 * V) % K;
    // every load of the sample in flight before the first LDS write (K = 64: 16 + 8 packs per lane): staged in
    // rounds of 4 the load phase was four serial HBM round trips per wave, and with ~1.5 waves per SIMD (24 KiB of LDS
    // per sample) that latency was the whole kernel
    constexpr int SU = 16, SZ = 8;
    Pack<T, V> pz[SZ];
#pragma unroll
    for (int u = 0; u < SZ; ++u) { const int i = (lane + 64 * u) * V; if (i < nv) pz[u] = load_pack<T, V, true>(zs + i); }
    for (; e < ne; e += SU * 64 * V) {
      Pack<T, V> p[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) if (e + u * 64 * V < ne) p[u] = load_pack<T, V, true>(Ws + e + u * 64 * V);
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (e + u * 64 * V < ne) {
          if constexpr (SWZ) {
            const Pack<T, V> q = swz_permute<T, V>(p[u], c & (V - 1));
            *reinterpret_cast<typename Vec16<T>::type*>(tile + c * K + (r ^ (c & ~(V - 1)))) = *reinterpret_cast<const typename Vec16<T>::type*>(&q);
          } else {
            int cc = c, rr = r;
#pragma unroll
            for (int j = 0; j < V; ++j) { tile[cc * P + rr] = p[u].v[j]; if (++rr == K) { rr = 0; ++cc; } }
          }
        }
        c += dc; r += dr;
        if (r >= K) { r -= K; ++c; }
      }
    }
#pragma unroll
    for (int u = 0; u < SZ; ++u) {
      const int i = (lane + 64 * u) * V;
      if (i < nv) {
        if constexpr (SWZ) *reinterpret_cast<typename Vec16<T>::type*>(dz + i) = *reinterpret_cast<const typename Vec16<T>::type*>(&pz[u]);
        else {
#pragma unroll
          for (int j = 0; j < V; ++j) dz[i + j] = pz[u].v[j];
        }
      }
    }
    for (int i = (lane + 64 * SZ) * V; i < nv; i += 64 * V) {
      const Pack<T, V> q = load_pack<T, V, true>(zs + i);
#pragma unroll
      for (int j = 0; j < V; ++j) dz[i + j] = q.v[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  // memory index of A[i][j] (A = the upper factor; :L stores its transpose): column-major W -> tile[col*P + row]
  auto at = [&](int i, int j) -> int { return SWZ ? (LOWER ? i * K + (j ^ i) : j * K + (i ^ j)) : (LOWER ? i * P + j : j * P + i); };
  for (int jb = 0; jb < K; jb += 64) {
    const int j = jb + lane;
    const bool live = j >= 1 && j < K;
    const int jj = live ? j : 1;
    const int base = jj * (jj - 1) / 2;
    T rs = tile[at(jj, jj)];
    rs *= rs;
    T dtmp = T(0);
    const int imax = (jb + 63 < K - 1 ? jb + 63 : K - 1) - 1;            // longest column of this strip
    // Rows in groups of 4: the LDS reads of a group are issued together, everything that does not depend on the
    // adjoint chain (tmp, p, the coefficients of Δtmp <- B·Δtmp + A) is evaluated for the four rows side by side,
    // and only the one-FMA chain and the in-place stores are serial.  (Row by row the loop was a chain of LDS read
    // -> two dependent transcendentals -> LDS write that the compiler may not reorder across the stores.)
    for (int i = imax; i >= 1; i -= 4) {
      bool on[4];
      int a[4];
      T w[4], dzi[4], rs2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ii = i - u;
        on[u] = live && ii >= 1 && ii < j;
        a[u] = at(on[u] ? ii : 0, jj);
        w[u] = on[u] ? tile[a[u]] : T(0);
        dzi[u] = dz[base + (on[u] ? ii : 0)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { rs2[u] = rs + w[u] * w[u]; rs = rs2[u]; }      // off rows add 0
      T rt[4], X[4], A[4], B[4], D[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        rt[u] = F::rsqrt(rs2[u]);                                        // 1/tmp
        const T p = w[u] * rt[u];
        const T q = T(1) - p * p;
        const T rf = F::rsqrt(q);                                        // 1/ftmp
        X[u] = (rs2[u] * rt[u]) * (p * rf);                              // tmp · p/ftmp
        D[u] = dzi[u] * F::rcp(q);                                       // Δz/(1-p²)
        const T g = w[u] * rt[u] * rt[u];                                // W/tmp²
        B[u] = q * rf + X[u] * g;                                        // Δtmp' = B Δtmp + A
        A[u] = -D[u] * g;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T dp = D[u] - dtmp * X[u];
        if (on[u]) { tile[a[u]] = dp * rt[u]; dtmp = B[u] * dtmp + A[u]; }
      }
    }
    if (live) {
      const int a0 = at(0, j);
      const T w0 = tile[a0];
      const T q0 = T(1) - w0 * w0;
      tile[a0] = dz[base] * F::rcp(q0) - dtmp * F::rsqrt(q0) * w0;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // diagonal, the other triangle and column/row 0 of it: zeros (SWZ: a mask on the way out)
  if constexpr (!SWZ) {
    for (int j = lane; j < K; j += 64)
      for (int i = j; i < K; ++i) tile[at(i, j)] = T(0);
    __builtin_amdgcn_wave_barrier();
  }
  {
    T* Os = Wbar + s * (int64_t)K * K;
    int e = lane * V, c = e / K, r = e % K;
    const int dc = (64 * V) / K, dr = (64 * V) % K;
    for (; e < ne; e += 64 * V) {
      Pack<T, V> p;
      if constexpr (SWZ) {
        Pack<T, V> q;
        *reinterpret_cast<typename Vec16<T>::type*>(&q) = *reinterpret_cast<const typename Vec16<T>::type*>(tile + c * K + (r ^ (c & ~(V - 1))));
        p = swz_permute<T, V>(q, c & (V - 1));
#pragma unroll
        for (int j = 0; j < V; ++j) p.v[j] = (LOWER ? r + j > c : r + j < c) ? p.v[j] : T(0);
      } else {
        int cc = c, rr = r;
#pragma unroll
        for (int j = 0; j < V; ++j) { p.v[j] = tile[cc * P + rr]; if (++rr == K) { rr = 0; ++cc; } }
      }
      store_pack<T, V, true>(Os + e, p);
      c += dc; r += dr;
      if (r >= K) { r -= K; ++c; }
    }
  }
}

// Small K: one lane per sample (cf. chol_lane_kernel): W in an odd-pitch tile, Δz in a second one; every lane walks the columns
// of its factor from the diagonal up with the recurrences of chol_fwd_vjp_kernel and writes ΔW over W in place.
template <class T, int V>
__global__ __launch_bounds__(64) void chol_fwd_vjp_lane_kernel(const T* __restrict__ W, const T* __restrict__ ybar, T* __restrict__ Wbar, int K, int Pw, int Py,
                                                               int lower, int64_t batch) {
  using F = Fast<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tw = reinterpret_cast<T*>(smem);
  T* tz = tw + (size_t)64 * Pw;
  const int lane = threadIdx.x;
  const int KK = K * K, nv = K * (K - 1) / 2;
  for (int64_t s0 = (int64_t)blockIdx.x * 64; s0 < batch; s0 += (int64_t)gridDim.x * 64) {
    const int ncols = (int)((batch - s0) < 64 ? (batch - s0) : 64);
    tile_stage_in<T, V>(tw, W + s0 * KK, KK, Pw, ncols, lane);
    tile_stage_in<T, V>(tz, ybar + s0 * nv, nv, Py, ncols, lane);
    tile_sync();
    if (lane < ncols) {
      T* w = tw + lane * Pw;
      const T* dz = tz + lane * Py;
      auto at = [&](int i, int j) -> int { return lower ? i * K + j : j * K + i; };     // A[i][j] of the upper factor
      for (int j = 1; j < K; ++j) {
        const int base = j * (j - 1) / 2;
        T rs = w[at(j, j)];
        rs *= rs;
        T dtmp = T(0);
        for (int i = j - 1; i >= 1; --i) {
          const int a = at(i, j);
          const T wi = w[a];
          rs += wi * wi;
          const T rt = F::rsqrt(rs);                                       // 1/tmp
          const T p = wi * rt;
          const T q = T(1) - p * p;
          const T rf = F::rsqrt(q);                                        // 1/ftmp
          const T X = (rs * rt) * (p * rf);                                // tmp · p/ftmp
          const T D = dz[base + i] * F::rcp(q);                            // Δz/(1-p²)
          const T g = wi * rt * rt;                                        // W/tmp²
          const T dp = D - dtmp * X;
          w[a] = dp * rt;
          dtmp = (q * rf + X * g) * dtmp - D * g;
        }
        const int a0 = at(0, j);
        const T w0 = w[a0];
        const T q0 = T(1) - w0 * w0;
        w[a0] = dz[base] * F::rcp(q0) - dtmp * F::rsqrt(q0) * w0;
      }
      for (int j = 0; j < K; ++j)                                          // diagonal and the other triangle: zeros
        for (int i = j; i < K; ++i) w[at(i, j)] = T(0);
    }
    tile_sync();
    tile_stage_out<T, V>(tw, Wbar + s0 * KK, KK, Pw, ncols, lane);
    tile_sync();
  }
}

template <class T>
int chol_fwd_vjp_impl(bjx_ctx* ctx, int uplo, const T* W, const T* y_bar, T* W_bar, int64_t K, int64_t batch) {
  if (batch == 0 || K < 1) return BJX_OK;
  if (K == 1) { BJX_HIP(ctx, hipMemsetAsync(W_bar, 0, (size_t)batch * sizeof(T), ctx->stream)); return BJX_OK; }   // no free parameter
  const int64_t nv = K * (K - 1) / 2;
  {
    static const int lane_max = getenv("BJX_CHOL_LANE_MAX") ? atoi(getenv("BJX_CHOL_LANE_MAX")) : 11;
    const int64_t Pw = (K * K) | 1, Py = nv | 1;
    const size_t smem_l = (size_t)64 * (Pw + Py) * sizeof(T);
    if (K <= lane_max && smem_l <= 56 * 1024) {
      constexpr int VW = Vec16<T>::N;
      const int64_t tiles = (batch + 63) / 64;
      const int64_t cap = (int64_t)ctx->num_cu * 32;
      const int grid_l = (int)(tiles < cap ? tiles : cap);
      const bool vec = bjx_aligned16(W) && bjx_aligned16(y_bar) && bjx_aligned16(W_bar);
      BjxProf prof_(ctx);
      if (vec) hipLaunchKernelGGL((chol_fwd_vjp_lane_kernel<T, VW>), dim3(grid_l), dim3(64), smem_l, ctx->stream, W, y_bar, W_bar, (int)K, (int)Pw, (int)Py, uplo == 'L' ? 1 : 0, batch);
      else hipLaunchKernelGGL((chol_fwd_vjp_lane_kernel<T, 1>), dim3(grid_l), dim3(64), smem_l, ctx->stream, W, y_bar, W_bar, (int)K, (int)Pw, (int)Py, uplo == 'L' ? 1 : 0, batch);
      BJX_CHECK_LAUNCH(ctx);
      return BJX_OK;
    }
  }
  constexpr int VW = Vec16<T>::N;
  const bool v_ok = bjx_aligned16(W) && bjx_aligned16(y_bar) && bjx_aligned16(W_bar) && (K * K) % VW == 0 && nv % VW == 0;
  static const int use_swz = getenv("BJX_CHOL_FWD_VJP_SWZ") ? atoi(getenv("BJX_CHOL_FWD_VJP_SWZ")) : 1;
  const bool swz = use_swz && v_ok && (K & (K - 1)) == 0 && K >= VW;       // pitch-K tile, XOR-swizzled columns, 16-byte LDS accesses
  const size_t smem = ((((size_t)K * (K + 1) + 3) / 4) * 4 + (size_t)nv + 4) * sizeof(T);
  BJX_REQUIRE(ctx, smem <= BJX_LDS_MAX, BJX_ERR_UNSUPPORTED, "bjx_vec_cholesky_fwd_vjp: K = %lld too large for the LDS tile", (long long)K);
  BJX_REQUIRE(ctx, batch < (int64_t)1 << 31, BJX_ERR_UNSUPPORTED, "batch too large for one launch");
  const bool lower = uplo == 'L';
  {
    BjxProf prof_(ctx);
#define CFV(V_, L_, S_) do { bjx_allow_big_lds(chol_fwd_vjp_kernel<T, V_, L_, S_>, smem); hipLaunchKernelGGL((chol_fwd_vjp_kernel<T, V_, L_, S_>), dim3((unsigned)batch), dim3(64), smem, ctx->stream, W, y_bar, W_bar, (int)K, batch); } while (0)
    if (swz) { if (lower) CFV(VW, true, true); else CFV(VW, false, true); }
    else if (v_ok) { if (lower) CFV(VW, true, false); else CFV(VW, false, false); }
    else { if (lower) CFV(1, true, false); else CFV(1, false, false); }
#undef CFV
  }
  BJX_CHECK_LAUNCH(ctx);
  return BJX_OK;
}
}  // namespace

BJX_API int bjx_vec_cholesky_fwd_vjp(bjx_ctx* ctx, bjx_dtype dt, int uplo, const void* W, const void* y_bar, void* W_bar, int64_t K, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, uplo == 'U' || uplo == 'L', BJX_ERR_ARG, "mode must be either :U (upper triangular) or :L (lower triangular)");
  BJX_REQUIRE(ctx, K >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_vec_cholesky_fwd_vjp: bad size");
  BJX_REQUIRE(ctx, (W && W_bar && (y_bar || K == 1)) || batch == 0, BJX_ERR_ARG, "bjx_vec_cholesky_fwd_vjp: null pointer");
  if (dt == BJX_F32) return chol_fwd_vjp_impl<float>(ctx, uplo, (const float*)W, (const float*)y_bar, (float*)W_bar, K, batch);
  if (dt == BJX_F64) return chol_fwd_vjp_impl<double>(ctx, uplo, (const double*)W, (const double*)y_bar, (double*)W_bar, K, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_vec_cholesky_fwd_vjp: bad dtype %d", (int)dt);
}

BJX_API int bjx_vec_cholesky_inv_vjp(bjx_ctx* ctx, bjx_dtype dt, int uplo, const void* y, const void* W_bar, const void* logJ_bar,
                                     void* y_bar, int64_t K, int64_t batch) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, uplo == 'U' || uplo == 'L', BJX_ERR_ARG, "mode must be either :U (upper triangular) or :L (lower triangular)");
  BJX_REQUIRE(ctx, K >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_vec_cholesky_inv_vjp: bad size");
  BJX_REQUIRE(ctx, (y && W_bar && y_bar) || batch == 0 || K == 1, BJX_ERR_ARG, "bjx_vec_cholesky_inv_vjp: null pointer");
  if (dt == BJX_F32) return chol_inv_vjp_impl<float>(ctx, uplo, (const float*)y, (const float*)W_bar, (const float*)logJ_bar, (float*)y_bar, K, batch);
  if (dt == BJX_F64) return chol_inv_vjp_impl<double>(ctx, uplo, (const double*)y, (const double*)W_bar, (const double*)logJ_bar, (double*)y_bar, K, batch);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_vec_cholesky_inv_vjp: bad dtype %d", (int)dt);
}

BJX_API int bjx_vec_cholesky(bjx_ctx* ctx, bjx_dtype dt, int inverse, int uplo, const void* in, void* out, void* ladj_ps,
                             double* ladj_sum, int64_t K, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, uplo == 'U' || uplo == 'L', BJX_ERR_ARG, "mode must be either :U (upper triangular) or :L (lower triangular)");  // corr.jl:215-219
  BJX_REQUIRE(ctx, K >= 1 && batch >= 0, BJX_ERR_SHAPE, "bjx_vec_cholesky: bad size");
  BJX_REQUIRE(ctx, in || batch == 0 || K == 1, BJX_ERR_ARG, "bjx_vec_cholesky: null input");
  BJX_REQUIRE(ctx, out || inverse || batch == 0, BJX_ERR_ARG, "bjx_vec_cholesky: the forward link needs an output buffer");
  if (dt == BJX_F32) return chol_impl<float>(ctx, inverse, uplo, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, K, batch, flags);
  if (dt == BJX_F64) return chol_impl<double>(ctx, inverse, uplo, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, K, batch, flags);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_vec_cholesky: bad dtype %d", (int)dt);
}

/* Leading-dimension variants: the columns are windows [row0, row0 + rows) of a taller matrix (segments of a Stacked,
 * stacked.jl:142-166); in / out point at the first row of the window in column 0.  One-lane-per-column LDS tile kernel. */
BJX_API int bjx_ordered_ld(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, int64_t ld_in, void* out, int64_t ld_out, void* ladj_ps,
                           double* ladj_sum, int64_t dim, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, dim >= 1 && batch >= 0 && ld_in >= dim && ld_out >= dim, BJX_ERR_SHAPE, "bjx_ordered_ld: bad size");
  BJX_REQUIRE(ctx, (in && out) || batch == 0, BJX_ERR_ARG, "bjx_ordered_ld: null pointer");
  if (dt == BJX_F32) return ordered_impl<float>(ctx, inverse, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, dim, batch, flags, ld_in, ld_out);
  if (dt == BJX_F64) return ordered_impl<double>(ctx, inverse, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, dim, batch, flags, ld_in, ld_out);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_ordered_ld: bad dtype %d", (int)dt);
}
BJX_API int bjx_simplex_ld(bjx_ctx* ctx, bjx_dtype dt, int inverse, const void* in, int64_t ld_in, void* out, int64_t ld_out, void* ladj_ps,
                           double* ladj_sum, int64_t K, int64_t batch, uint32_t flags) {
  if (!ctx) return BJX_ERR_ARG;
  BJX_REQUIRE(ctx, K > 1, BJX_ERR_SHAPE, "bjx_simplex_ld: x needs to be of length greater than 1 (simplex.jl:30), got K=%lld", (long long)K);
  BJX_REQUIRE(ctx, batch >= 0 && ld_in >= (inverse ? K - 1 : K) && ld_out >= (inverse ? K : K - 1), BJX_ERR_SHAPE, "bjx_simplex_ld: bad size");
  BJX_REQUIRE(ctx, (in && out) || batch == 0, BJX_ERR_ARG, "bjx_simplex_ld: null pointer");
  if (dt == BJX_F32) return simplex_impl<float>(ctx, inverse, (const float*)in, (float*)out, (float*)ladj_ps, ladj_sum, K, batch, flags, ld_in, ld_out);
  if (dt == BJX_F64) return simplex_impl<double>(ctx, inverse, (const double*)in, (double*)out, (double*)ladj_ps, ladj_sum, K, batch, flags, ld_in, ld_out);
  return bjx_fail(ctx, BJX_ERR_ARG, "bjx_simplex_ld: bad dtype %d", (int)dt);
}
