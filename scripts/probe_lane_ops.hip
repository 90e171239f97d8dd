// probe_lane_ops.hip — prints what the gfx950 cross-lane primitives used by the planar register
// kernel do to lane ids (run on the GPU box; documentation of semantics, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  int a = lane, b = 100 + lane;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  out[lane] = a; out[64 + lane] = b;
  int c = lane, d = 100 + lane;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
  out[128 + lane] = c; out[192 + lane] = d;
  out[256 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0xB1, 0xF, 0xF, true);
  out[320 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x4E, 0xF, 0xF, true);
  out[384 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x141, 0xF, 0xF, true);
  out[448 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x140, 0xF, 0xF, true);
}
int main() {
  int* d; hipMalloc(&d, 512 * 4);
  k<<<1, 64>>>(d);
  int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"p16swap.a", "p16swap.b", "p32swap.a", "p32swap.b", "quad[1,0,3,2]", "quad[2,3,0,1]", "row_half_mirror", "row_mirror"};
  for (int r = 0; r < 8; ++r) { printf("%-16s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %d", h[r * 64 + i]); printf("\n"); }
  return 0;
}
