// Wave-private [64][P] LDS tile staging: one wave moves 64 consecutive columns (one contiguous run of global memory) with 16-byte
// accesses and hands every column to a lane.  Shared by the column walkers of bjx_seq.hip, bjx_stacked.hip, bjx_matrix.hip,
// bjx_flow.hip and the column-walker form of the group skeleton (bjx_stream.h).
#pragma once
#include "bjx_internal.h"

namespace bjx {
// ---- wave-private [64][P] tile staging (single-wave blocks; shared by seq_wave_kernel and the VJP kernels)
// A full tile (64 columns) is a whole number of 16-byte packs (64*rows*sizeof(T) % 16 == 0); the ragged
// last wave of the batch takes the element-wise path.  SU independent 16-byte loads are in flight per lane.
template <class T, int V>
__device__ __forceinline__ void tile_stage_in(T* tile, const T* __restrict__ src, int rows, int P, int ncols, int lane) {
  if (ncols == 64) {
    constexpr int SU = 8;
    const int ne = 64 * rows;
    const int dc = (64 * V) / rows, dr = (64 * V) % rows;
    int e = lane * V, c = e / rows, r = e % rows;
    for (; e < ne; e += SU * 64 * V) {
      Pack<T, V> p[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (e + u * 64 * V < ne) p[u] = load_pack<T, V, true>(src + e + u * 64 * V);
      }
      if (V > 1 && rows % V == 0) {
        // column heights that are whole packs: a pack never straddles two columns — one address, V stores at constant offsets
        // (the generic path below pays a compare, two selects and an add per ELEMENT: a third of a walker's instructions)
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          if (e + u * 64 * V < ne) {
            T* d = tile + c * P + r;
#pragma unroll
            for (int j = 0; j < V; ++j) d[j] = p[u].v[j];
          }
          c += dc; r += dr;
          if (r >= rows) { r -= rows; ++c; }
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (e + u * 64 * V < ne) {
          int cc = c, rr = r;
#pragma unroll
          for (int j = 0; j < V; ++j) {
            tile[cc * P + rr] = p[u].v[j];
            if (++rr == rows) { rr = 0; ++cc; }
          }
        }
        c += dc; r += dr;
        if (r >= rows) { r -= rows; ++c; }
      }
    }
  } else {
    const int ne = ncols * rows;
    for (int e = lane; e < ne; e += 64) tile[(e / rows) * P + e % rows] = src[e];
  }
}
template <class T, int V>
__device__ __forceinline__ void tile_stage_out(const T* tile, T* __restrict__ dst, int rows, int P, int ncols, int lane) {
  if (ncols == 64) {
    constexpr int SU = 4;
    const int ne = 64 * rows;
    const int dc = (64 * V) / rows, dr = (64 * V) % rows;
    int e = lane * V, c = e / rows, r = e % rows;
    for (; e < ne; e += SU * 64 * V) {
      Pack<T, V> p[SU];
      if (V > 1 && rows % V == 0) {
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          if (e + u * 64 * V < ne) {
            const T* d = tile + c * P + r;
#pragma unroll
            for (int j = 0; j < V; ++j) p[u].v[j] = d[j];
          }
          c += dc; r += dr;
          if (r >= rows) { r -= rows; ++c; }
        }
      } else {
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        int cc = c, rr = r;
        if (e + u * 64 * V < ne) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            p[u].v[j] = tile[cc * P + rr];
            if (++rr == rows) { rr = 0; ++cc; }
          }
        }
        c += dc; r += dr;
        if (r >= rows) { r -= rows; ++c; }
      }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (e + u * 64 * V < ne) store_pack<T, V, true>(dst + e + u * 64 * V, p[u]);
      }
    }
  } else {
    const int ne = ncols * rows;
    for (int e = lane; e < ne; e += 64) dst[e] = tile[(e / rows) * P + e % rows];
  }
}
// strided variants (leading dimension ld != rows: the columns are windows of a taller matrix — Stacked segments):
// consecutive lanes walk the rows of a column, 4- / 8-byte accesses, runs of `rows` contiguous elements
template <class T>
__device__ __forceinline__ void tile_stage_in_ld(T* tile, const T* __restrict__ src, int rows, int64_t ld, int P, int ncols, int lane) {
  const int ne = ncols * rows;
  const int dc = 64 / rows, dr = 64 % rows;
  int c = lane / rows, r = lane % rows;
  for (int e = lane; e < ne; e += 64) {
    tile[c * P + r] = src[(int64_t)c * ld + r];
    c += dc; r += dr;
    if (r >= rows) { r -= rows; ++c; }
  }
}
template <class T>
__device__ __forceinline__ void tile_stage_out_ld(const T* tile, T* __restrict__ dst, int rows, int64_t ld, int P, int ncols, int lane) {
  const int ne = ncols * rows;
  const int dc = 64 / rows, dr = 64 % rows;
  int c = lane / rows, r = lane % rows;
  for (int e = lane; e < ne; e += 64) {
    dst[(int64_t)c * ld + r] = tile[c * P + r];
    c += dc; r += dr;
    if (r >= rows) { r -= rows; ++c; }
  }
}
// single-wave block: the LDS queue is in order, only pin the compiler
__device__ __forceinline__ void tile_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }


}  // namespace bjx
