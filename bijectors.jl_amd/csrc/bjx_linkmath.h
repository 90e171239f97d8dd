// LKJ link arithmetic (corr.jl:277-399) shared by bjx_matrix.hip (Corr / VecCorr) and bjx_seq.hip (VecCholesky, one lane per
// sample).  Included INSIDE the anonymous namespace of each translation unit (after `using namespace bjx;`).
// asinh / atanh / tanh / logcosh of the link: Float32 on the hardware log/exp/rcp units (parity bar 1e-3), Float64 exact OCML
template <class T> struct LinkMath;
template <> struct LinkMath<float> {
  using F = Fast<float>;
  // y = asinh(w / sqrt(rem)), lc = logcosh(y) = log(sqrt(1 + z^2))
  static __device__ __forceinline__ void asinh_lc(float w, float rem, float& y, float& lc) {
    const float r = F::rsqrt(rem);
    const float z = w * r;
    const float t = F::sqrt(z * z + 1.0f);
    const float az = fabsf(z);
    y = __builtin_copysignf(F::log1p(az + z * z * F::rcp(1.0f + t)), z);     // log(|z| + sqrt(z^2+1)) without cancellation near 0
    lc = F::log(t);
  }
  static __device__ __forceinline__ void atanh_lc(float w, float& y, float& lc) {
    y = 0.5f * F::log((1.0f + w) * F::rcp(1.0f - w));
    lc = -0.5f * F::log((1.0f - w) * (1.0f + w));                            // logcosh(atanh w) = -log(1 - w^2)/2
  }
  // z = tanh(y), lc = logcosh(y) from one exp (LogExpFunctions.logcosh: |y| + log1pexp(-2|y|) - log 2)
  static __device__ __forceinline__ void tanh_lc(float y, float& z, float& lc) {
    const float ay = fabsf(y);
    const float t = F::exp(-2.0f * ay);
    z = __builtin_copysignf((1.0f - t) * F::rcp(1.0f + t), y);
    lc = ay + F::log1p(t) - Num<float>::log2;
  }
  static __device__ __forceinline__ float exp(float x) { return F::exp(x); }
  static __device__ __forceinline__ float log(float x) { return F::log(x); }
  // One row of the bottom-up column walk (corr.jl:282-288) with the logarithm of the running remainder carried along:
  //   asinh(w/√R) = log(|w| + √(R + w²)) − ½ log R,   logcosh(asinh(w/√R)) = ½ (log(R + w²) − log R)
  // -> sqrt + 2 log per entry instead of rsqrt, sqrt, rcp, log1p, log.  L = log2 of the remainder.
  static __device__ __forceinline__ void fwd_init(float dg, float& rem, float& L) { rem = dg * dg; L = F::log2(rem); }
  static __device__ __forceinline__ void fwd_step(float w, float& rem, float& L, float& y, float& lc) {
    const float rn = rem + w * w;
    const float Ln = F::log2(rn);
    const float a = F::log2(fabsf(w) + F::sqrt(rn)) - 0.5f * L;
    y = __builtin_copysignf(a * Num<float>::log2, w);
    lc = (0.5f * Num<float>::log2) * (Ln - L);
    rem = rn; L = Ln;
  }
  // One row of the top-down walk (corr.jl:352-357): E = exp(log_remainder) is carried as a product of sech(y),
  // tanh and sech come from one exp(-|y|) and one rcp, logcosh = -log(sech): exp + rcp + log per entry.
  static __device__ __forceinline__ void inv_init(float& E) { E = 1.0f; }
  static __device__ __forceinline__ void inv_step(float yv, float& E, float& w, float& lc) {
    const float u = F::exp(-fabsf(yv));
    const float t = u * u;
    const float r = F::rcp(1.0f + t);
    w = __builtin_copysignf((1.0f - t) * r, yv) * E;
    const float sech = (u + u) * r;
    lc = -F::log(sech);
    E *= sech;
  }
  static __device__ __forceinline__ float inv_diag(float E, float) { return E; }
};
template <> struct LinkMath<double> {
  using F = Fast<double>;
  static __device__ __forceinline__ void asinh_lc(double w, double rem, double& y, double& lc) {
    const double z = w * F::rsqrt(rem);
    y = x_asinh(z);
    lc = 0.5 * F::log1p(z * z);
  }
  static __device__ __forceinline__ void atanh_lc(double w, double& y, double& lc) {
    y = x_atanh(w);
    lc = -0.5 * F::log1p(-w * w);
  }
  static __device__ __forceinline__ void tanh_lc(double y, double& z, double& lc) {
    z = x_tanh(y);
    lc = f_logcosh(y);
  }
  static __device__ __forceinline__ double exp(double x) { return F::exp(x); }
  static __device__ __forceinline__ double log(double x) { return F::log(x); }
  static __device__ __forceinline__ void fwd_init(double dg, double& rem, double& L) { rem = dg * dg; L = 0.0; }
  static __device__ __forceinline__ void fwd_step(double w, double& rem, double& L, double& y, double& lc) {
    asinh_lc(w, rem, y, lc);
    rem += w * w;
  }
  static __device__ __forceinline__ void inv_init(double& E) { E = 0.0; }           // E holds log_remainder in Float64 (no product: no underflow question)
  static __device__ __forceinline__ void inv_step(double yv, double& E, double& w, double& lc) {
    double z;
    tanh_lc(yv, z, lc);
    w = z * F::exp(E);
    E -= lc;
  }
  static __device__ __forceinline__ double inv_diag(double E, double) { return F::exp(E); }
};

