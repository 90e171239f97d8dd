/* abi_smoke.c — a plain C99 consumer of include/bjx.h: what a Julia `ccall` relies on (C linkage, plain pointers and
 * sizes, no C++ types).  Built and run by tests/test_abi_c.py:
 *     gcc -std=c99 -Wall -Werror -pedantic -I include tests/abi_smoke.c -o abi_smoke \
 *         bijectors.jl_amd/libbjx_hip.so -L/opt/rocm/lib -lamdhip64 -lm
 * It creates a context, runs exp ∘ Shift(0.1) ∘ Scale(0.5) (BASELINE configs[1]) on a 64 x 1024 Float32 batch through
 * bjx_chain, reads the result back and checks every value and the summed log-det against libm.  The HIP runtime is
 * used for device memory only and is declared by hand so that this file needs no C++-flavoured header. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bjx.h"

extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind); /* 1 = host to device, 2 = device to host */
extern int hipSetDevice(int device);

#define CHECK(expr)                                                          \
  do {                                                                       \
    int rc_ = (expr);                                                        \
    if (rc_ != 0) {                                                          \
      fprintf(stderr, "%s -> %d (%s)\n", #expr, rc_, ctx ? bjx_last_error(ctx) : ""); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

int main(void) {
  const int64_t dim = 64, batch = 1024;
  const size_t n = (size_t)(dim * batch);
  bjx_ctx* ctx = NULL;
  float *hx, *hy, *dx = NULL, *dy = NULL, *dl = NULL;
  double *dsum = NULL, hsum = 0.0, ref = 0.0;
  bjx_op ops[3];
  size_t i;
  int bad = 0;

  if (bjx_version() != BJX_VERSION) { fprintf(stderr, "header/library version mismatch\n"); return 1; }
  CHECK(hipSetDevice(0));
  CHECK(bjx_create(0, NULL, &ctx));
  hx = (float*)malloc(n * sizeof(float));
  hy = (float*)malloc(n * sizeof(float));
  for (i = 0; i < n; ++i) hx[i] = (float)((double)((i * 2654435761u) % 20001u) / 10000.0 - 1.0); /* [-1, 1] */
  CHECK(hipMalloc((void**)&dx, n * sizeof(float)));
  CHECK(hipMalloc((void**)&dy, n * sizeof(float)));
  CHECK(hipMalloc((void**)&dl, (size_t)batch * sizeof(float)));
  CHECK(hipMalloc((void**)&dsum, sizeof(double)));
  CHECK(hipMemcpy(dx, hx, n * sizeof(float), 1));

  memset(ops, 0, sizeof(ops));
  ops[0].kind = BJX_OP_SCALE; ops[0].param_len = 1; ops[0].p0 = 0.5;
  ops[1].kind = BJX_OP_SHIFT; ops[1].param_len = 1; ops[1].p0 = 0.1;
  ops[2].kind = BJX_OP_EXP;
  CHECK(bjx_chain(ctx, BJX_F32, ops, 3, dx, dy, dl, dsum, dim, batch, 0u));
  CHECK(bjx_synchronize(ctx));
  CHECK(hipMemcpy(hy, dy, n * sizeof(float), 2));
  CHECK(hipMemcpy(&hsum, dsum, sizeof(double), 2));

  for (i = 0; i < n; ++i) {
    const double u = 0.5 * (double)hx[i] + 0.1;
    const double want = exp(u);
    ref += u + log(0.5);
    if (fabs((double)hy[i] - want) > 1e-3 * want) ++bad;
  }
  if (bad) { fprintf(stderr, "%d values differ from libm by more than 1e-3 relative\n", bad); return 1; }
  if (fabs(hsum - ref) > 1e-5 * fabs(ref)) { fprintf(stderr, "sum log-det %.9g vs %.9g\n", hsum, ref); return 1; }
  /* an argument error comes back as a status code, never as a C++ exception */
  if (bjx_chain(ctx, BJX_F32, ops, 3, NULL, dy, NULL, NULL, dim, batch, 0u) >= 0) { fprintf(stderr, "NULL input accepted\n"); return 1; }
  if (strlen(bjx_last_error(ctx)) == 0) { fprintf(stderr, "no error message\n"); return 1; }
  hipFree(dx); hipFree(dy); hipFree(dl); hipFree(dsum);
  CHECK(bjx_destroy(ctx));
  free(hx); free(hy);
  printf("abi_smoke ok: sum logabsdetjac = %.6f (libm %.6f)\n", hsum, ref);
  return 0;
}
