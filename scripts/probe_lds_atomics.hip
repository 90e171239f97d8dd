// LDS atomic throughput on gfx950 (round 4: `rqs_knots_lds_kernel` ran 8x slower with ds_add_f32 than with racy plain read-modify-write).
// Each lane adds into a 3 x 17 x 33-word table at a pseudo-random (bin, row) address, 6 adds per "element", like the spline's knot
// cotangents: row = 4 * (lane & 7) + j, bin random.   hipcc --offload-arch=gfx950 -O3 -o probe_lds_atomics.co probe_lds_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned long long tab64[3 * 17 * 33];
  float* tf = reinterpret_cast<float*>(tab64);
  unsigned* tu = reinterpret_cast<unsigned*>(tab64);
  for (int i = threadIdx.x; i < 3 * 17 * 33; i += 256) tab64[i] = 0;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const int gl = threadIdx.x & 7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s = s * 1664525u + 1013904223u;
      const int kb = (s >> 20) & 15, row = gl * 4 + j;
      const float v = (float)(s & 1023) * 1e-3f;
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int a = ((t >> 1) * 17 + kb + (t & 1)) * 33 + row;
        if (MODE == 0) atomicAdd(tf + a, v);                                  // ds_add_f32
        else if (MODE == 1) atomicAdd(tu + a, (unsigned)(v * 1024.f));        // ds_add_u32
        else if (MODE == 2) atomicAdd(tab64 + a, (unsigned long long)(long long)(v * 1048576.f));   // ds_add_u64
        else if (MODE == 3) { const float o = tf[a]; tf[a] = o + v; }         // racy plain RMW
        else if (MODE == 4) __hip_atomic_fetch_add(tf + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 5) (void)__builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(tf + a), v, 0, 0, false);
        else if (MODE == 6) atomicAdd(reinterpret_cast<double*>(tab64) + a, (double)v);                       // ds_add_f64 (round 6: is the Float64 form slow too?)
        else if (MODE == 7) __hip_atomic_fetch_add(reinterpret_cast<double*>(tab64) + a, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = tf[threadIdx.x] + (float)tab64[threadIdx.x + 64];
}
template <int MODE> void run(const char* name, float* out) {
  const int grid = 1024, iters = 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 16);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double adds = (double)grid * 256 * iters * 24;
  printf("%-28s %8.3f ms  %7.1f G lane-adds/s  (%.2f per clock per CU at 2.4 GHz, 256 CUs)\n", name, ms, adds / ms * 1e-6, adds / (ms * 1e-3) / 2.4e9 / 256);
}
int main() {
  float* out; hipMalloc(&out, 1024 * 64 * 4);
  run<0>("atomicAdd float (ds_add_f32)", out); run<4>("hip_atomic_fetch_add wg f32", out); run<5>("ds_faddf builtin", out);
  run<6>("atomicAdd double (ds_add_f64)", out); run<7>("hip_atomic_fetch_add wg f64", out);
  run<1>("atomicAdd u32 (ds_add_u32)", out); run<2>("atomicAdd u64 (ds_add_u64)", out); run<3>("plain read-add-write (racy)", out);
  return 0;
}
