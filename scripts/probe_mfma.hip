// Layout probe for v_mfma_f32_16x16x4_f32 on gfx950: prints which (m, n) of D = A·B every (lane, vgpr) holds.
// hipcc --offload-arch=gfx950 -O2 scripts/probe_mfma.hip -o /tmp/probe_mfma && /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  // assume A[i][k] on lane (i = l % 16, k = l / 16), B[k][n] on lane (n = l % 16, k = l / 16)
  const float a = (float)((l % 16) * 100 + (l / 16) * 1000);       // A[i][k] = 100 i + 1000 k
  const float b = (l / 16 == 2) ? (float)(l % 16 + 1) : 0.0f;       // B[k][n] = (k == 2) (n + 1)
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);      // D[m][n] = A[m][2] (n + 1) = (100 m + 2000)(n + 1)
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
  float* d; hipMalloc(&d, 256 * sizeof(float));
  probe<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const int m = 4 * (l / 16) + v, n = l % 16;
    const float want = (100.f * m + 2000.f) * (n + 1);
    if (h[l * 4 + v] != want) { ok = 0; if (l < 20) printf("lane %d vgpr %d: got %g want %g\n", l, v, h[l * 4 + v], want); }
  }
  printf("layout D[4*(lane/16)+vgpr][lane%%16] with A[lane%%16][lane/16], B[lane/16][lane%%16]: %s\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}
