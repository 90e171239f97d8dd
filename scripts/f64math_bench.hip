// Microbenchmark + accuracy check of Float64 elementary functions on gfx950: OCML (::exp, ::log, ...) against the lean
// versions of bjx_internal.h (Fast<double>): throughput in G evaluations/s (one wave-wide loop, 8 independent chains)
// and the maximum relative error against the host's long double libm on 2^20 points.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bijectors.jl_amd/csrc scripts/f64math_bench.hip -o /tmp/f64bench && /tmp/f64bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "bjx_internal.h"
using namespace bjx;

enum { F_EXP_OCML, F_EXP_FAST, F_LOG_OCML, F_LOG_FAST, F_DIV_OCML, F_DIV_FAST, F_TANH_OCML, F_TANH_FAST, F_LOG1P_OCML, F_LOG1P_FAST, F_SQRT_OCML, F_SQRT_FAST,
       F_RSQRT_OCML, F_RSQRT_FAST, F_ASINH_OCML, F_ASINH_FAST, F_N };
const char* NAMES[F_N] = {"exp ocml", "exp fast", "log ocml", "log fast", "1/x ocml", "1/x fast", "tanh ocml", "tanh fast", "log1p ocml", "log1p fast",
                          "sqrt ocml", "sqrt fast", "rsqrt ocml", "rsqrt fast", "asinh ocml", "asinh fast"};

template <int F> __device__ __forceinline__ double ev(double x) {
  if (F == F_EXP_OCML) return ::exp(x);
  if (F == F_EXP_FAST) return Fast<double>::exp(x);
  if (F == F_LOG_OCML) return ::log(x);
  if (F == F_LOG_FAST) return Fast<double>::log(x);
  if (F == F_DIV_OCML) return 1.0 / x;
  if (F == F_DIV_FAST) return Fast<double>::rcp(x);
  if (F == F_TANH_OCML) return ::tanh(x);
  if (F == F_TANH_FAST) return fast_tanh64(x);
  if (F == F_LOG1P_OCML) return ::log1p(x);
  if (F == F_LOG1P_FAST) return Fast<double>::log1p(x);
  if (F == F_SQRT_OCML) return ::sqrt(x);
  if (F == F_SQRT_FAST) return Fast<double>::sqrt(x);
  if (F == F_RSQRT_OCML) return 1.0 / ::sqrt(x);
  if (F == F_RSQRT_FAST) return Fast<double>::rsqrt(x);
  if (F == F_ASINH_OCML) return ::asinh(x);
  return fast_asinh64(x);
}
template <int F> __global__ void acc_kernel(const double* x, double* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = ev<F>(x[i]);
}
template <int F> __global__ void speed_kernel(double* out, double a, double b, int iters) {
  // 8 independent chains per lane; x stays inside the function's domain: x <- a + b * frac-like map of the result
  double v[8];
  for (int j = 0; j < 8; ++j) v[j] = a + b * (double)((threadIdx.x * 8 + j) % 97) / 97.0;
  for (int it = 0; it < iters; ++it)
    for (int j = 0; j < 8; ++j) { const double r = ev<F>(v[j]); v[j] = a + b * (r - floor(r)); }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int F> void run(const std::vector<double>& hx, double a, double b, long double (*ref)(long double)) {
  const int n = (int)hx.size();
  double *dx, *dy;
  hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8);
  hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
  acc_kernel<F><<<(n + 255) / 256, 256>>>(dx, dy, n);
  std::vector<double> hy(n);
  hipMemcpy(hy.data(), dy, n * 8, hipMemcpyDeviceToHost);
  long double worst = 0;
  for (int i = 0; i < n; ++i) {
    const long double r = ref((long double)hx[i]);
    const long double e = r == 0 ? fabsl((long double)hy[i]) : fabsl(((long double)hy[i] - r) / r);
    if (e > worst) worst = e;
  }
  const int blocks = 256 * 8, iters = 2000;
  double* dout; hipMalloc(&dout, blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  speed_kernel<F><<<blocks, 256>>>(dout, a, b, 10);
  hipEventRecord(e0);
  speed_kernel<F><<<blocks, 256>>>(dout, a, b, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-12s max rel err %.2Le   %8.1f G eval/s  (incl. ~6 Float64 ops of loop overhead per eval)\n", NAMES[F], worst, (double)blocks * 256 * 8 * iters / (ms * 1e6));
  hipFree(dx); hipFree(dy); hipFree(dout);
}
static std::vector<double> grid(double lo, double hi, int n, bool logspace) {
  std::vector<double> v(n);
  for (int i = 0; i < n; ++i) { const double t = (i + 0.5) / n; v[i] = logspace ? std::exp(std::log(lo) + t * (std::log(hi) - std::log(lo))) : lo + (hi - lo) * t; }
  return v;
}
static long double r_rcp(long double x) { return 1 / x; }
static long double r_rsqrt(long double x) { return 1 / sqrtl(x); }
int main() {
  const int n = 1 << 20;
  auto ex = grid(-700, 700, n, false), lg = grid(1e-300, 1e300, n, true), th = grid(-20, 20, n, false), l1 = grid(-0.999, 1e6, n, false);
  auto l1s = grid(1e-18, 1.0, n, true);
  run<F_EXP_OCML>(ex, -3, 6, expl); run<F_EXP_FAST>(ex, -3, 6, expl);
  run<F_LOG_OCML>(lg, 0.1, 50, logl); run<F_LOG_FAST>(lg, 0.1, 50, logl);
  run<F_DIV_OCML>(lg, 0.1, 50, r_rcp); run<F_DIV_FAST>(lg, 0.1, 50, r_rcp);
  run<F_TANH_OCML>(th, -3, 6, tanhl); run<F_TANH_FAST>(th, -3, 6, tanhl);
  run<F_LOG1P_OCML>(l1, -0.5, 5, log1pl); run<F_LOG1P_FAST>(l1, -0.5, 5, log1pl);
  printf("  (small arguments) "); run<F_LOG1P_FAST>(l1s, -0.5, 5, log1pl);
  run<F_SQRT_OCML>(lg, 0.1, 50, sqrtl); run<F_SQRT_FAST>(lg, 0.1, 50, sqrtl);
  run<F_RSQRT_OCML>(lg, 0.1, 50, r_rsqrt); run<F_RSQRT_FAST>(lg, 0.1, 50, r_rsqrt);
  run<F_ASINH_OCML>(th, -3, 6, asinhl); run<F_ASINH_FAST>(th, -3, 6, asinhl);
  return 0;
}
