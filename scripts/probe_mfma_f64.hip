// Layout probe for v_mfma_f64_16x16x4_f64 on gfx950: which (m, n) of D = A·B every (lane, register) holds.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe_mfma_f64.hip -o scripts/probe_mfma_f64.co && scripts/probe_mfma_f64.co
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  // assume A[i][k] on lane (i = l % 16, k = l / 16), B[k][n] on lane (n = l % 16, k = l / 16)
  const double a = (double)((l % 16) * 100 + (l / 16) * 1000);      // A[i][k] = 100 i + 1000 k
  const double b = (l / 16 == 2) ? (double)(l % 16 + 1) : 0.0;       // B[k][n] = (k == 2) (n + 1)
  d4 c = {0., 0., 0., 0.};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);        // D[m][n] = A[m][2] (n + 1) = (100 m + 2000)(n + 1)
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
  double* d; hipMalloc(&d, 256 * sizeof(double));
  probe<<<1, 64>>>(d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // decode: D value = (100 m + 2000)(n + 1) -> find (m, n) for every (lane, reg)
  int okA = 1, okB = 1;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const double got = h[l * 4 + v];
    const int mA = 4 * (l / 16) + v, mB = 4 * v + l / 16, n = l % 16;
    if (got != (100.0 * mA + 2000.0) * (n + 1)) okA = 0;
    if (got != (100.0 * mB + 2000.0) * (n + 1)) okB = 0;
  }
  printf("D[4*(lane/16)+reg][lane%%16]: %s   D[4*reg+lane/16][lane%%16]: %s\n", okA ? "CONFIRMED" : "no", okB ? "CONFIRMED" : "no");
  for (int l = 0; l < 64; l += 16) printf("lane %2d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
