// bjx_matrix_vjp.h — pieces shared by the pullbacks of the matrix-variate bijectors (bjx_matrix_vjp.hip: one lane per sample;
// bjx_matrix_vjp_grp.hip: one group of lanes per sample).
#pragma once
#include "bjx_internal.h"

namespace bjx {

enum { MK_VEC_CORR = 0, MK_CORR = 1, MK_PD = 2, MK_PD_VEC = 3 };

// tanh(y), sech²(y) and the pivot / remainder math: Float32 on the hardware units (parity bar 1e-3), Float64 on the lean pieces
template <class T> struct VjpMath;
template <> struct VjpMath<float> {
  using F = Fast<float>;
  static __device__ __forceinline__ void tanh_sech2(float y, float& z, float& s2) {
    const float u = F::exp(-fabsf(y));
    const float t = u * u;
    const float r = F::rcp(1.0f + t);
    z = __builtin_copysignf((1.0f - t) * r, y);
    const float sech = (u + u) * r;
    s2 = sech * sech;
  }
  // tanh and sech themselves (the LKJ sweep multiplies by sech: no square, no root)
  static __device__ __forceinline__ void tanh_sech(float y, float& z, float& sech) {
    const float u = F::exp(-fabsf(y));
    const float t = u * u;
    const float r = F::rcp(1.0f + t);
    z = __builtin_copysignf((1.0f - t) * r, y);
    sech = (u + u) * r;
  }
  static __device__ __forceinline__ float exp(float x) { return F::exp(x); }
  static __device__ __forceinline__ float sqrt(float x) { return F::sqrt(x); }
  static __device__ __forceinline__ float rcp(float x) { return F::rcp(x); }
  static __device__ __forceinline__ void pivot(float d, float& rs, float& sq) { rs = F::rsqrt(d); sq = d * rs; }
};
template <> struct VjpMath<double> {
  using F = Fast<double>;
  static __device__ __forceinline__ void tanh_sech2(double y, double& z, double& s2) { x_tanh_sech2(y, z, s2); }
  static __device__ __forceinline__ void tanh_sech(double y, double& z, double& sech) { double s2; x_tanh_sech2(y, z, s2); sech = ::sqrt(s2); }
  static __device__ __forceinline__ double exp(double x) { return F::exp(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ void pivot(double d, double& rs, double& sq) { sq = ::sqrt(d); rs = 1.0 / sq; }
};

template <int KIND> __host__ __device__ inline int64_t free_len(int64_t K) {
  return KIND == MK_VEC_CORR ? K * (K - 1) / 2 : (KIND == MK_PD_VEC ? K * (K + 1) / 2 : K * K);
}


// 16 x 16 x 4 MFMA of the element type, and where register r of lane (n, q) of its result sits (bjx_matrix_vjp_mfma*.hip)
template <class T> struct VjpMfma;
template <> struct VjpMfma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static constexpr int N = 4;
  typedef float V __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }        // D register r of lane (n, q) -> row of the 16-block
};
template <> struct VjpMfma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static constexpr int N = 2;
  typedef double V __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int q, int r) { return 4 * r + q; }        // probed: scripts/probe_mfma_f64.hip
};

// 8 < K <= 64 (Float64: <= 32): one group of 16 / 32 / 64 lanes per sample, the factor and its cotangent in LDS (bjx_matrix_vjp_grp.hip).
// Returns 1 when the shape is not served.
int bjx_matrix_vjp_grp(bjx_ctx* ctx, bjx_dtype dt, int kind, int inverse, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch);

// The forward direction (X -> y): Cholesky reverse as S = W' Φ(L' L̄) W with W = L⁻¹ by blocks, every product on MFMA (bjx_matrix_vjp_mfma_fwd.hip).  Returns 1 when not served.
int bjx_matrix_fwd_vjp_mfma(bjx_ctx* ctx, bjx_dtype dt, int kind, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch);

// The inverse direction (y -> X) of the same shapes with the cubic step as MFMA blocks (bjx_matrix_vjp_mfma.hip).  Returns 1 when not served.
int bjx_matrix_inv_vjp_mfma(bjx_ctx* ctx, bjx_dtype dt, int kind, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch);

}  // namespace bjx
