// membench.hip — streaming-pattern microbenchmark (not part of the product): which launch geometry
// reaches the HBM copy ceiling on this MI355X for a read-4GiB / write-4GiB float4 stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ f4 op(f4 v) { f4 r; for (int j = 0; j < 4; ++j) r[j] = __expf(v[j] * 0.5f + 0.1f); return r; }

// A: one pack per thread
template <bool NT> __global__ __launch_bounds__(256) void k_one(const f4* x, f4* y, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) st<NT>(y + i, op(ld<NT>(x + i)));
}
// B: grid-stride, U in flight
template <int U, bool NT> __global__ __launch_bounds__(256) void k_gs(const f4* x, f4* y, long n) {
  long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += stride * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(x + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(y + i + u * stride, op(v[u]));
  }
  for (; i < n; i += stride) st<NT>(y + i, op(ld<NT>(x + i)));
}
// C: block-contiguous tiles: block b owns packs [b*chunk, (b+1)*chunk); U consecutive 4 KiB rows in flight
template <int U, bool NT> __global__ __launch_bounds__(256) void k_tile(const f4* x, f4* y, long n, long chunk) {
  long lo = (long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  long i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += 256 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(x + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(y + i + u * 256, op(v[u]));
  }
  for (; i < hi; i += 256) st<NT>(y + i, op(ld<NT>(x + i)));
}
// D: U consecutive packs per thread, huge grid, no loop
template <int U, bool NT> __global__ __launch_bounds__(256) void k_multi(const f4* x, f4* y, long n) {
  long base = (long)blockIdx.x * 256 * U + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = ld<NT>(x + base + u * 256);
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) st<NT>(y + base + u * 256, op(v[u]));
}
// E: one pack per thread + per-block f64 partial (the log-det epilogue), block size BS
template <int BS, bool NT> __global__ __launch_bounds__(BS) void k_one_red(const f4* x, f4* y, long n, double* partials) {
  __shared__ double red[16];
  long i = (long)blockIdx.x * BS + threadIdx.x;
  double acc = 0.0;
  if (i < n) { f4 v = ld<NT>(x + i); acc = (double)(v[0] + v[1] + v[2] + v[3]); st<NT>(y + i, op(v)); }
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0; for (int w = 0; w < BS / 64; ++w) s += red[w]; partials[blockIdx.x] = s; }
}
// F: S packs per thread processed one after the other (row s of the block's S consecutive 4 KiB rows)
template <int S, bool NT> __global__ __launch_bounds__(256) void k_seq(const f4* x, f4* y, long n) {
  long base = (long)blockIdx.x * 256 * S + threadIdx.x;
#pragma unroll 1
  for (int s = 0; s < S; ++s) { long i = base + s * 256; if (i < n) st<NT>(y + i, op(ld<NT>(x + i))); }
}
template <int BS, bool NT> __global__ __launch_bounds__(BS) void k_one_bs(const f4* x, f4* y, long n) {
  long i = (long)blockIdx.x * BS + threadIdx.x;
  if (i < n) st<NT>(y + i, op(ld<NT>(x + i)));
}
// read-only / write-only
__global__ __launch_bounds__(256) void k_read(const f4* x, float* out, long n) {
  long stride = (long)gridDim.x * 256; float s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { f4 v = x[i]; s += v[0] + v[1] + v[2] + v[3]; }
  if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(f4* y, long n) {
  long stride = (long)gridDim.x * 256; f4 v = {1, 2, 3, 4};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = v;
}

template <class F> double timeit(F f, int reps = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main() {
  const long n = 1L << 28;  // packs: 4 GiB in, 4 GiB out
  f4 *x, *y; CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16));
  CK(hipMemset(x, 0x3c, n * 16)); CK(hipMemset(y, 0, n * 16));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); int cu = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.name, cu);
  auto rep = [&](const char* name, double ms, double bytes) { printf("%-44s %8.4f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0); fflush(stdout); };
  const double B = 2.0 * n * 16;
  rep("memcpyD2D", timeit([&] { CK(hipMemcpyAsync(y, x, n * 16, hipMemcpyDeviceToDevice, 0)); }), B);
  rep("A one-pack/thread", timeit([&] { k_one<false><<<(unsigned)(n / 256), 256>>>(x, y, n); }), B);
  rep("A one-pack/thread NT", timeit([&] { k_one<true><<<(unsigned)(n / 256), 256>>>(x, y, n); }), B);
  rep("D 2 packs/thread no loop", timeit([&] { k_multi<2, false><<<(unsigned)(n / 512), 256>>>(x, y, n); }), B);
  rep("D 4 packs/thread no loop", timeit([&] { k_multi<4, false><<<(unsigned)(n / 1024), 256>>>(x, y, n); }), B);
  rep("D 4 packs/thread no loop NT", timeit([&] { k_multi<4, true><<<(unsigned)(n / 1024), 256>>>(x, y, n); }), B);
  rep("D 8 packs/thread no loop", timeit([&] { k_multi<8, false><<<(unsigned)(n / 2048), 256>>>(x, y, n); }), B);
  rep("D 8 packs/thread no loop NT", timeit([&] { k_multi<8, true><<<(unsigned)(n / 2048), 256>>>(x, y, n); }), B);
  double* parts; CK(hipMalloc(&parts, (n / 64) * 8));
  rep("E one-pack + block partial BS=256", timeit([&] { k_one_red<256, false><<<(unsigned)(n / 256), 256>>>(x, y, n, parts); }), B);
  rep("E one-pack + block partial BS=256 NT", timeit([&] { k_one_red<256, true><<<(unsigned)(n / 256), 256>>>(x, y, n, parts); }), B);
  rep("E one-pack + block partial BS=512 NT", timeit([&] { k_one_red<512, true><<<(unsigned)(n / 512), 512>>>(x, y, n, parts); }), B);
  rep("E one-pack + block partial BS=1024 NT", timeit([&] { k_one_red<1024, true><<<(unsigned)(n / 1024), 1024>>>(x, y, n, parts); }), B);
  rep("A one-pack BS=64 NT", timeit([&] { k_one_bs<64, true><<<(unsigned)(n / 64), 64>>>(x, y, n); }), B);
  rep("A one-pack BS=128 NT", timeit([&] { k_one_bs<128, true><<<(unsigned)(n / 128), 128>>>(x, y, n); }), B);
  rep("A one-pack BS=512 NT", timeit([&] { k_one_bs<512, true><<<(unsigned)(n / 512), 512>>>(x, y, n); }), B);
  rep("A one-pack BS=1024 NT", timeit([&] { k_one_bs<1024, true><<<(unsigned)(n / 1024), 1024>>>(x, y, n); }), B);
  rep("F 2 rows sequential NT", timeit([&] { k_seq<2, true><<<(unsigned)(n / 512), 256>>>(x, y, n); }), B);
  rep("F 4 rows sequential NT", timeit([&] { k_seq<4, true><<<(unsigned)(n / 1024), 256>>>(x, y, n); }), B);
  rep("F 16 rows sequential NT", timeit([&] { k_seq<16, true><<<(unsigned)(n / 4096), 256>>>(x, y, n); }), B);
  for (int k : {6}) {
    char nm[64];
    snprintf(nm, 64, "B grid-stride U=1 blocks/CU=%d", k); rep(nm, timeit([&] { k_gs<1, false><<<cu * k, 256>>>(x, y, n); }), B);
    snprintf(nm, 64, "B grid-stride U=2 blocks/CU=%d", k); rep(nm, timeit([&] { k_gs<2, false><<<cu * k, 256>>>(x, y, n); }), B);
    snprintf(nm, 64, "B grid-stride U=4 blocks/CU=%d", k); rep(nm, timeit([&] { k_gs<4, false><<<cu * k, 256>>>(x, y, n); }), B);
    snprintf(nm, 64, "B grid-stride U=4 NT blocks/CU=%d", k); rep(nm, timeit([&] { k_gs<4, true><<<cu * k, 256>>>(x, y, n); }), B);
    snprintf(nm, 64, "B grid-stride U=8 blocks/CU=%d", k); rep(nm, timeit([&] { k_gs<8, false><<<cu * k, 256>>>(x, y, n); }), B);
  }
  for (int k : {256}) {
    char nm[64]; long grid = (long)cu * k; long chunk = ((n + grid - 1) / grid + 255) / 256 * 256;
    snprintf(nm, 64, "C tiles U=4 blocks/CU=%d", k); rep(nm, timeit([&] { k_tile<4, false><<<(unsigned)grid, 256>>>(x, y, n, chunk); }), B);
    snprintf(nm, 64, "C tiles U=4 NT blocks/CU=%d", k); rep(nm, timeit([&] { k_tile<4, true><<<(unsigned)grid, 256>>>(x, y, n, chunk); }), B);
    snprintf(nm, 64, "C tiles U=8 blocks/CU=%d", k); rep(nm, timeit([&] { k_tile<8, false><<<(unsigned)grid, 256>>>(x, y, n, chunk); }), B);
  }
  float* o; CK(hipMalloc(&o, 4));
  rep("read-only grid-stride 8/CU", timeit([&] { k_read<<<cu * 8, 256>>>(x, o, n); }), n * 16.0);
  rep("write-only grid-stride 8/CU", timeit([&] { k_write<<<cu * 8, 256>>>(y, n); }), n * 16.0);
  return 0;
}
