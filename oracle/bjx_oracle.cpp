// bjx_oracle.cpp — CPU restatement of Bijectors.jl's batched transform + logabsdetjac path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load this library, and only as the checker / the timed CPU baseline.  The product
// (libbjx_hip.so and the bijectors.jl_amd host package) never links, imports or calls it.
//
// Every function restates the cited reference lines (paths relative to /root/reference,
// Bijectors.jl v0.16.2) with the SAME loop order, epsilon placement, clamp points and branch
// conditions, templated on float/double exactly like the Julia code is generic in T.
//
// Parity status: the reference is pure Julia and cannot run in this image (no julia binary),
// so the oracle is pinned against the golden vectors / known-answer tests the reference's
// docstrings and tests contain (tests/test_oracle_golden.py) and against the
// reference's own property tests re-expressed in pytest (Jacobian log-det, round trips,
// ladj(inverse) == -ladj(forward)).  Bit-level results of exp/log/tanh are "parity unpinned":
// the reference itself only ever compares with isapprox.
//
// Third-party scalar math that is NOT under /root/reference (compat ranges only, Project.toml:44-68)
// is restated from the published definitions:
//   LogExpFunctions (0.3.3 / 1.0): logit, logistic, log1pexp, logcosh, softmax
//   ChangesOfVariables 0.1: scalar exp/log rules, broadcast rule (sum of ladj), ComposedFunction rule
//   Roots (1.3.15/2/3) A42: bracketing to adjacent floats (restated as safeguarded bisection that
//       terminates on adjacent floats; pinned by residual only, test/normalising_flows.jl:47-70)
//   Base.sum: pairwise summation with 1024-element leaves (base/reduce.jl mapreduce_impl)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/bjx.h"

namespace {

// ---------------------------------------------------------------- third-party scalar math
template <class T> inline T bj_eps() { return std::numeric_limits<T>::epsilon(); }  // src/Bijectors.jl:91

// src/Bijectors.jl:95-100
template <class T> inline T bj_clamp(T x, T a, T b) { return x < a ? a : (x > b ? b : x); }

// LogExpFunctions.logit: log(x / (1 - x))
template <class T> inline T logit_(T x) { return std::log(x / (T(1) - x)); }

// LogExpFunctions.logistic: e/(1+e) with exact 0 / 1 saturation outside the representable range
template <class T> struct LogisticBounds;
template <> struct LogisticBounds<float> { static constexpr float lo = -103.27893f, hi = 16.635532f; };
template <> struct LogisticBounds<double> { static constexpr double lo = -744.4400719213812, hi = 36.7368005696771; };
template <class T> inline T logistic_(T x) {
  T e = std::exp(x);
  return x < LogisticBounds<T>::lo ? T(0) : (x > LogisticBounds<T>::hi ? T(1) : e / (T(1) + e));
}

// LogExpFunctions.log1pexp: 4 branches
template <class T> struct L1peThr;
template <> struct L1peThr<float> { static constexpr float x0 = -16.635532f, x1 = 7.9711924f, x2 = 13.993f; };
template <> struct L1peThr<double> { static constexpr double x0 = -36.7368005696771, x1 = 18.021826694558577, x2 = 33.23111882352963; };
template <class T> inline T log1pexp_(T x) {
  if (x < L1peThr<T>::x0) return std::exp(x);
  if (x < L1peThr<T>::x1) return std::log1p(std::exp(x));
  if (x < L1peThr<T>::x2) return x + std::exp(-x);
  return x;
}

// LogExpFunctions.logcosh: abs(x) + log1pexp(-2abs(x)) - log(2)
template <class T> inline T logcosh_(T x) {
  T ax = std::fabs(x);
  return ax + log1pexp_<T>(T(-2) * ax) - T(0.6931471805599453094172321214581766L);
}

// Base.sum over an array: pairwise with 1024-element leaves (base/reduce.jl mapreduce_impl)
template <class T, class F> T pairwise_sum(F f, int64_t first, int64_t last /*inclusive*/) {
  if (first == last) return f(first);
  if (last - first < 1024) {
    T v = f(first) + f(first + 1);
    for (int64_t i = first + 2; i <= last; ++i) v = v + f(i);
    return v;
  }
  int64_t mid = first + ((last - first) >> 1);
  T v1 = pairwise_sum<T>(f, first, mid);
  T v2 = pairwise_sum<T>(f, mid + 1, last);
  return v1 + v2;
}
template <class T, class F> T jl_sum(F f, int64_t n) { return n <= 0 ? T(0) : pairwise_sum<T>(f, 0, n - 1); }

// ---------------------------------------------------------------- F1 elementwise
// exp_log.jl:5-6 + ChangesOfVariables broadcast rule: y = exp.(x), ladj = sum(x)
template <class T> T exp_fwd(const T* x, T* y, int64_t n) {
  T l = jl_sum<T>([&](int64_t i) { return x[i]; }, n);
  for (int64_t i = 0; i < n; ++i) y[i] = std::exp(x[i]);
  return l;
}
// exp_log.jl:8-9: y = log.(x), ladj = -sum(log, x)
template <class T> T log_fwd(const T* x, T* y, int64_t n) {
  T l = -jl_sum<T>([&](int64_t i) { return std::log(x[i]); }, n);
  for (int64_t i = 0; i < n; ++i) y[i] = std::log(x[i]);
  return l;
}
// shift.jl:14,21
template <class T> T shift_fwd(const T* x, T* y, int64_t dim, int64_t batch, const T* a, int64_t alen) {
  for (int64_t n = 0; n < batch; ++n)
    for (int64_t i = 0; i < dim; ++i) y[n * dim + i] = (alen == 1 ? a[0] : a[i]) + x[n * dim + i];
  return T(0);
}
// scale.jl:13,26-32.  scalar a: log|a| * length(x); vector a: sum(log∘abs, a) (NOT x N, :31-32)
template <class T> T scale_fwd(const T* x, T* y, int64_t dim, int64_t batch, const T* a, int64_t alen) {
  for (int64_t n = 0; n < batch; ++n)
    for (int64_t i = 0; i < dim; ++i) y[n * dim + i] = (alen == 1 ? a[0] : a[i]) * x[n * dim + i];
  if (alen == 1) return std::log(std::fabs(a[0])) * T(dim * batch);
  return jl_sum<T>([&](int64_t i) { return std::log(std::fabs(a[i])); }, alen);
}
// scale.jl:15-16: transform(Scale(inv(a)), y); ladj via interface.jl:276-281 = -logabsdetjac(Scale(a), x)
template <class T> T scale_inv(const T* y, T* x, int64_t dim, int64_t batch, const T* a, int64_t alen) {
  for (int64_t n = 0; n < batch; ++n)
    for (int64_t i = 0; i < dim; ++i) x[n * dim + i] = (T(1) / (alen == 1 ? a[0] : a[i])) * y[n * dim + i];
  if (alen == 1) return -(std::log(std::fabs(a[0])) * T(dim * batch));
  return -jl_sum<T>([&](int64_t i) { return std::log(std::fabs(a[i])); }, alen);
}
// logit.jl:15,24-30
template <class T> T logit_fwd(const T* x, T* y, int64_t n, T a, T b) {
  T l = jl_sum<T>([&](int64_t i) { return -std::log((x[i] - a) * (b - x[i]) / (b - a)); }, n);
  for (int64_t i = 0; i < n; ++i) y[i] = logit_<T>((x[i] - a) / (b - a));
  return l;
}
// logit.jl:19-21; ladj = -logabsdetjac(Logit, x) by interface.jl:276-281
template <class T> T logit_inv(const T* y, T* x, int64_t n, T a, T b) {
  for (int64_t i = 0; i < n; ++i) x[i] = (b - a) * logistic_<T>(y[i]) + a;
  return -jl_sum<T>([&](int64_t i) { return -std::log((x[i] - a) * (b - x[i]) / (b - a)); }, n);
}
// leaky_relu.jl:25-29: mask = x < 0; J = mask*α + !mask; (J.*x, sum(log.(abs.(J))))
template <class T> T leaky_fwd(const T* x, T* y, int64_t n, T alpha) {
  T l = jl_sum<T>([&](int64_t i) { T J = (x[i] < T(0)) ? alpha : T(1); return std::log(std::fabs(J)); }, n);
  for (int64_t i = 0; i < n; ++i) { T J = (x[i] < T(0)) ? alpha : T(1); y[i] = J * x[i]; }
  return l;
}
// truncated.jl:20-31
template <class T> inline T truncated_link(T x, T a, T b) {
  bool lo = std::isfinite(a), up = std::isfinite(b);
  if (lo && up) return logit_<T>((x - a) / (b - a));
  if (lo) return std::log(x - a);
  if (up) return std::log(b - x);
  return x;
}
// truncated.jl:38-49
template <class T> inline T truncated_invlink(T y, T a, T b) {
  bool lo = std::isfinite(a), up = std::isfinite(b);
  if (lo && up) return (b - a) * logistic_<T>(y) + a;
  if (lo) return std::exp(y) + a;
  if (up) return b - std::exp(y);
  return y;
}
// truncated.jl:56-67
template <class T> inline T truncated_ladj(T x, T a, T b) {
  bool lo = std::isfinite(a), up = std::isfinite(b);
  if (lo && up) return -std::log((x - a) * (b - x) / (b - a));
  if (lo) return -std::log(x - a);
  if (up) return -std::log(b - x);
  return T(0);
}
// truncated.jl:71-82
template <class T> inline T truncated_inv_ladj(T y, T a, T b) {
  bool lo = std::isfinite(a), up = std::isfinite(b);
  if (lo && up) { T ay = std::fabs(y); return std::log(b - a) - ay - T(2) * log1pexp_<T>(-ay); }
  if (lo || up) return y;
  return T(0);
}
// truncated.jl:15-18,51-54,69
template <class T> T truncated_fwd(const T* x, T* y, int64_t dim, int64_t batch, const T* lb, const T* ub, int64_t blen) {
  auto A = [&](int64_t i) { return blen == 1 ? lb[0] : lb[i % dim]; };
  auto B = [&](int64_t i) { return blen == 1 ? ub[0] : ub[i % dim]; };
  int64_t n = dim * batch;
  T l = jl_sum<T>([&](int64_t i) { return truncated_ladj<T>(bj_clamp<T>(x[i], A(i), B(i)), A(i), B(i)); }, n);
  for (int64_t i = 0; i < n; ++i) y[i] = truncated_link<T>(bj_clamp<T>(x[i], A(i), B(i)), A(i), B(i));
  return l;
}
// truncated.jl:33-36,84-91
template <class T> T truncated_inv(const T* y, T* x, int64_t dim, int64_t batch, const T* lb, const T* ub, int64_t blen) {
  auto A = [&](int64_t i) { return blen == 1 ? lb[0] : lb[i % dim]; };
  auto B = [&](int64_t i) { return blen == 1 ? ub[0] : ub[i % dim]; };
  int64_t n = dim * batch;
  T l = jl_sum<T>([&](int64_t i) { return truncated_inv_ladj<T>(y[i], A(i), B(i)); }, n);
  for (int64_t i = 0; i < n; ++i) x[i] = bj_clamp<T>(truncated_invlink<T>(y[i], A(i), B(i)), A(i), B(i));
  return l;
}

// composed.jl:4-14 + ChangesOfVariables ComposedFunction rule: inner first, ladj terms added.
// Reference-structured: one allocating full pass per stage.
// Host pointers in bjx_op::v0/v1 (same struct as the device ABI, host memory here).
template <class T> T chain_fwd(const bjx_op* ops, int n_ops, const T* x, T* y, int64_t dim, int64_t batch) {
  int64_t n = dim * batch;
  std::vector<T> cur(x, x + n), nxt(n);
  T total = T(0);
  for (int k = 0; k < n_ops; ++k) {
    const bjx_op& op = ops[k];
    T s0 = T(op.p0), s1 = T(op.p1);
    const T* v0 = op.v0 ? static_cast<const T*>(op.v0) : &s0;
    const T* v1 = op.v1 ? static_cast<const T*>(op.v1) : &s1;
    int64_t plen = op.param_len <= 1 ? 1 : op.param_len;
    T l = T(0);
    switch (op.kind) {
      case BJX_OP_EXP: l = exp_fwd<T>(cur.data(), nxt.data(), n); break;
      case BJX_OP_LOG: l = log_fwd<T>(cur.data(), nxt.data(), n); break;
      case BJX_OP_SHIFT: l = shift_fwd<T>(cur.data(), nxt.data(), dim, batch, v0, plen); break;
      case BJX_OP_SCALE: l = scale_fwd<T>(cur.data(), nxt.data(), dim, batch, v0, plen); break;
      case BJX_OP_SCALE_INV: l = scale_inv<T>(cur.data(), nxt.data(), dim, batch, v0, plen); break;
      case BJX_OP_LOGIT: l = logit_fwd<T>(cur.data(), nxt.data(), n, v0[0], v1[0]); break;
      case BJX_OP_LOGIT_INV: l = logit_inv<T>(cur.data(), nxt.data(), n, v0[0], v1[0]); break;
      case BJX_OP_LEAKY_RELU: l = leaky_fwd<T>(cur.data(), nxt.data(), n, v0[0]); break;
      case BJX_OP_TRUNCATED: l = truncated_fwd<T>(cur.data(), nxt.data(), dim, batch, v0, v1, plen); break;
      case BJX_OP_TRUNCATED_INV: l = truncated_inv<T>(cur.data(), nxt.data(), dim, batch, v0, v1, plen); break;
      case BJX_OP_SIGNFLIP: for (int64_t i = 0; i < n; ++i) nxt[i] = -cur[i]; break;  // ordered.jl:3
      default: nxt = cur; break;
    }
    total = total + l;
    cur.swap(nxt);
  }
  std::memcpy(y, cur.data(), sizeof(T) * n);
  return total;
}

// The same chain fused into ONE pass (the best a CPU can do with the same math; BASELINE.md §2
// "fused" variant).  Sums per-element ladj in double.  Used only as a CPU baseline leg.
template <class T> double chain_fused(const bjx_op* ops, int n_ops, const T* x, T* y, int64_t dim, int64_t batch) {
  double total = 0.0;
  for (int64_t n = 0; n < batch; ++n) {
    T part = T(0);
    for (int64_t i = 0; i < dim; ++i) {
      T v = x[n * dim + i];
      for (int k = 0; k < n_ops; ++k) {
        const bjx_op& op = ops[k];
        T a = op.v0 ? static_cast<const T*>(op.v0)[op.param_len > 1 ? i : 0] : T(op.p0);
        T b = op.v1 ? static_cast<const T*>(op.v1)[op.param_len > 1 ? i : 0] : T(op.p1);
        switch (op.kind) {
          case BJX_OP_EXP: part += v; v = std::exp(v); break;
          case BJX_OP_LOG: { T l = std::log(v); part -= l; v = l; } break;
          case BJX_OP_SHIFT: v = a + v; break;
          case BJX_OP_SCALE: part += std::log(std::fabs(a)); v = a * v; break;
          case BJX_OP_SCALE_INV: part -= std::log(std::fabs(a)); v = (T(1) / a) * v; break;
          case BJX_OP_LOGIT: part += -std::log((v - a) * (b - v) / (b - a)); v = logit_<T>((v - a) / (b - a)); break;
          case BJX_OP_LOGIT_INV: v = (b - a) * logistic_<T>(v) + a; part += std::log((v - a) * (b - v) / (b - a)); break;
          case BJX_OP_LEAKY_RELU: { T J = v < T(0) ? a : T(1); part += std::log(std::fabs(J)); v = J * v; } break;
          case BJX_OP_TRUNCATED: { T c = bj_clamp<T>(v, a, b); part += truncated_ladj<T>(c, a, b); v = truncated_link<T>(c, a, b); } break;
          case BJX_OP_TRUNCATED_INV: part += truncated_inv_ladj<T>(v, a, b); v = bj_clamp<T>(truncated_invlink<T>(v, a, b), a, b); break;
          case BJX_OP_SIGNFLIP: v = -v; break;
          default: break;
        }
      }
      y[n * dim + i] = v;
    }
    total += double(part);
  }
  return total;
}

// ---------------------------------------------------------------- F3 sequential
// ordered.jl:36-49 (matrix), :79-80
template <class T> void ordered_fwd(const T* y, T* x, int64_t dim, int64_t batch, T* ladj) {
  for (int64_t j = 0; j < batch; ++j) {
    for (int64_t i = 0; i < dim; ++i) {
      if (i == 0) x[j * dim] = y[j * dim];
      else x[j * dim + i] = x[j * dim + i - 1] + std::exp(y[j * dim + i]);
    }
    if (ladj) ladj[j] = jl_sum<T>([&](int64_t i) { return y[j * dim + 1 + i]; }, dim - 1);
  }
}
// ordered.jl:63-77; ladj via interface.jl:276-281: -(logabsdetjac(OrderedBijector, y_out))
template <class T> void ordered_inv(const T* x, T* y, int64_t dim, int64_t batch, T* ladj) {
  for (int64_t j = 0; j < batch; ++j) {
    for (int64_t i = 0; i < dim; ++i) {
      if (i == 0) y[j * dim] = x[j * dim];
      else y[j * dim + i] = std::log(x[j * dim + i] - x[j * dim + i - 1]);
    }
    if (ladj) ladj[j] = -jl_sum<T>([&](int64_t i) { return y[j * dim + 1 + i]; }, dim - 1);
  }
}

// simplex.jl:47-64 (matrix form)
template <class T> void simplex_fwd(const T* X, T* Y, int64_t K, int64_t N) {
  const T e = bj_eps<T>();
  for (int64_t n = 0; n < N; ++n) {
    const T* x = X + n * K; T* y = Y + n * (K - 1);
    T sum_tmp = T(0);
    T z = x[0] * (T(1) - 2 * e) + e;
    y[0] = logit_<T>(z) + std::log(T(K - 1));
    for (int64_t k = 2; k <= K - 1; ++k) {
      sum_tmp += x[k - 2];
      z = (x[k - 1] + e) * (T(1) - 2 * e) / ((T(1) + e) - sum_tmp);
      y[k - 1] = logit_<T>(z) + std::log(T(K - k));
    }
  }
}
// simplex.jl:102-120
template <class T> void simplex_inv(const T* Y, T* X, int64_t K, int64_t N) {
  const T e = bj_eps<T>();
  for (int64_t n = 0; n < N; ++n) {
    const T* y = Y + n * (K - 1); T* x = X + n * K;
    T sum_tmp = T(0);
    T z = logistic_<T>(y[0] - std::log(T(K - 1)));
    x[0] = bj_clamp<T>((z - e) / (T(1) - 2 * e), T(0), T(1));
    for (int64_t k = 2; k <= K - 1; ++k) {
      z = logistic_<T>(y[k - 1] - std::log(T(K - k)));
      sum_tmp += x[k - 2];
      x[k - 1] = bj_clamp<T>(((T(1) + e) - sum_tmp) / (T(1) - 2 * e) * z - e, T(0), T(1));
    }
    sum_tmp += x[K - 2];
    x[K - 1] = bj_clamp<T>(T(1) - sum_tmp, T(0), T(1));
  }
}
// simplex.jl:122-138 (one column)
template <class T> T simplex_ladj_col(const T* x, int64_t K) {
  const T e = bj_eps<T>();
  T lp = T(0), sum_tmp = T(0);
  T z = x[0];
  lp += std::log(std::max(z, e)) + std::log(std::max(T(1) - z, e));
  for (int64_t k = 2; k <= K - 1; ++k) {
    sum_tmp += x[k - 2];
    z = x[k - 1] / std::max(T(1) - sum_tmp, e);
    lp += std::log(std::max(z, e)) + std::log(std::max(T(1) - z, e)) + std::log(std::max(T(1) - sum_tmp, e));
  }
  return -lp;
}

// src/utils.jl:99
inline int64_t triu1_dim_from_length(int64_t d) {
  int64_t s = (int64_t)std::floor(std::sqrt((double)(1 + 8 * d)));
  while (s * s > 1 + 8 * d) --s;
  while ((s + 1) * (s + 1) <= 1 + 8 * d) ++s;
  return (1 + s) / 2;
}
// corr.jl:314-335 (W upper triangular K x K column-major, one sample) -> y[K(K-1)/2]
template <class T> void link_chol_lkj_from_upper(const T* W, T* y, int64_t K) {
  int64_t starting_idx = 0;  // 0-based
  for (int64_t j = 2; j <= K; ++j) {
    y[starting_idx] = std::atanh(W[(j - 1) * K + 0]);
    starting_idx += 1;
    T remainder_sq = W[(j - 1) * K + (j - 1)] * W[(j - 1) * K + (j - 1)];
    for (int64_t i = j - 1; i >= 2; --i) {
      int64_t idx = starting_idx + i - 2;
      T w = W[(j - 1) * K + (i - 1)];
      T z = w / std::sqrt(remainder_sq);
      y[idx] = std::asinh(z);
      remainder_sq += w * w;
    }
    starting_idx += (j - 2 > 0 ? j - 2 : 0);
  }
}
// corr.jl:370-399 (vector form) -> W[K,K] upper (lower zero-filled), returns logJ
template <class T> T inv_link_chol_lkj(const T* y, T* W, int64_t K) {
  T logJ = T(0);
  int64_t idx = 0;
  for (int64_t j = 1; j <= K; ++j) {
    T log_remainder = T(0);
    for (int64_t i = 1; i <= j - 1; ++i) {
      T z = std::tanh(y[idx]);
      if (W) W[(j - 1) * K + (i - 1)] = z * std::exp(log_remainder);
      log_remainder -= logcosh_<T>(y[idx]);
      logJ += log_remainder;
      idx += 1;
    }
    logJ += log_remainder;
    if (W) {
      W[(j - 1) * K + (j - 1)] = std::exp(log_remainder);
      for (int64_t i = j + 1; i <= K; ++i) W[(j - 1) * K + (i - 1)] = T(0);
    }
  }
  return logJ;
}
// corr.jl:485-501
template <class T> T logabsdetjac_inv_chol(const T* y, int64_t K) {
  T result = T(0);
  int64_t idx = 0;
  for (int64_t j = 2; j <= K; ++j) {
    T tmp = T(0);
    for (int64_t c = 1; c <= j - 1; ++c) {
      T lc = logcosh_<T>(y[idx]);
      tmp -= lc;
      result += tmp - lc;
      idx += 1;
    }
  }
  return result;
}

// ---------------------------------------------------------------- matrix-variate constraint bijectors
// src/utils.jl:37,50 — cholesky(Hermitian(X)).U reads the UPPER triangle of X, cholesky(Hermitian(X, :L)).L the LOWER one
// (LinearAlgebra -> LAPACK potrf, not under /root/reference).  The factor of a positive definite matrix is unique, so
// the restatement is the textbook unblocked factorisation (row by row, inner products in ascending order); it differs
// from LAPACK's blocked order at rounding level only.  L is returned DENSE column-major with the other triangle zeroed
// (lower_triangular / upper_triangular, src/utils.jl:14-15).
template <class T> void cholesky_lower_from(const T* X, T* L, int64_t K, bool upper_storage) {
  for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i < K; ++i) L[j * K + i] = T(0);
  for (int64_t i = 0; i < K; ++i) {
    for (int64_t j = 0; j <= i; ++j) {
      T a = upper_storage ? X[i * K + j] : X[j * K + i];      // A[i,j], i >= j
      T s = a;
      for (int64_t m = 0; m < j; ++m) s -= L[m * K + i] * L[m * K + j];
      if (i == j) L[j * K + i] = std::sqrt(s);
      else L[j * K + i] = s / L[j * K + j];
    }
  }
}
template <class T> void cholesky_upper(const T* X, T* U, int64_t K) {   // src/utils.jl:50
  std::vector<T> L(K * K);
  cholesky_lower_from<T>(X, L.data(), K, true);
  for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i < K; ++i) U[j * K + i] = L[i * K + j];
}
template <class T> void cholesky_lower(const T* X, T* L, int64_t K) { cholesky_lower_from<T>(X, L, K, false); }   // src/utils.jl:37
// corr.jl:277-297 (matrix form: asinh on every row, zero fill on and below the diagonal)
template <class T> void link_chol_lkj_mat(const T* W, T* Y, int64_t K) {
  for (int64_t j = 1; j <= K; ++j) {
    T remainder_sq = W[(j - 1) * K + (j - 1)] * W[(j - 1) * K + (j - 1)];
    for (int64_t i = j - 1; i >= 1; --i) {
      T w = W[(j - 1) * K + (i - 1)];
      T z = w / std::sqrt(remainder_sq);
      Y[(j - 1) * K + (i - 1)] = std::asinh(z);
      remainder_sq += w * w;
    }
    for (int64_t i = j; i <= K; ++i) Y[(j - 1) * K + (i - 1)] = T(0);
  }
}
// corr.jl:345-368 (matrix form of the inverse link; only the strict upper triangle of Y is read)
template <class T> T inv_link_chol_lkj_mat(const T* Y, T* W, int64_t K) {
  T logJ = T(0);
  for (int64_t j = 1; j <= K; ++j) {
    T log_remainder = T(0);
    for (int64_t i = 1; i <= j - 1; ++i) {
      T yv = Y[(j - 1) * K + (i - 1)];
      T z = std::tanh(yv);
      W[(j - 1) * K + (i - 1)] = z * std::exp(log_remainder);
      log_remainder -= logcosh_<T>(yv);
      logJ += log_remainder;
    }
    logJ += log_remainder;
    W[(j - 1) * K + (j - 1)] = std::exp(log_remainder);
    for (int64_t i = j + 1; i <= K; ++i) W[(j - 1) * K + (i - 1)] = T(0);
  }
  return logJ;
}
// corr.jl:453-461 / :463-472 (vec_to_triu1_row_index: src/utils.jl:123-128)
template <class T> T logabsdetjac_inv_corr_mat(const T* Y, int64_t K) {
  T result = T(0);
  for (int64_t j = 2; j <= K; ++j) for (int64_t i = 1; i <= j - 1; ++i) result -= T(K - i + 1) * logcosh_<T>(Y[(j - 1) * K + (i - 1)]);
  return result;
}
template <class T> T logabsdetjac_inv_corr_vec(const T* y, int64_t K) {
  T result = T(0);
  int64_t n = K * (K - 1) / 2;
  for (int64_t idx = 1; idx <= n; ++idx) {
    int64_t M = triu1_dim_from_length(idx - 1);
    int64_t row_idx = idx - (M * (M - 1) / 2);
    result -= T(K - row_idx + 1) * logcosh_<T>(y[idx - 1]);
  }
  return result;
}
// pd_from_upper / pd_from_lower (src/utils.jl:17-24): U'U, L L'
template <class T> void pd_from_upper(const T* U, T* X, int64_t K) {
  for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i < K; ++i) {
    T s = T(0);
    for (int64_t m = 0; m <= (i < j ? i : j); ++m) s += U[i * K + m] * U[j * K + m];
    X[j * K + i] = s;
  }
}
template <class T> void pd_from_lower(const T* L, T* X, int64_t K) {
  for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i < K; ++i) {
    T s = T(0);
    for (int64_t m = 0; m <= (i < j ? i : j); ++m) s += L[m * K + i] * L[m * K + j];
    X[j * K + i] = s;
  }
}
// pd.jl:27-31
template <class T> T logabsdetjac_pdbijector_chol(const T* L, int64_t d) {
  T z = T(0);
  for (int64_t i = 1; i <= d; ++i) z += T(d + 2 - i) * std::log(L[(i - 1) * d + (i - 1)]);
  return -(z + T(d) * T(0.6931471805599453094172321214581766));
}
// kind 0: VecCorrBijector (corr.jl:128-162)   X[K,K] <-> y[K(K-1)/2]
// kind 1: CorrBijector    (corr.jl:64-92)     X[K,K] <-> Y[K,K] (strict upper triangle, zeros elsewhere)
// kind 2: PDBijector      (pd.jl:1-36)        X[K,K] <-> Y[K,K] (lower factor with log diagonal)
// kind 3: PDVecBijector   (pd.jl:38-60)       X[K,K] <-> y[K(K+1)/2] (triu_to_vec of the transposed PD link)
template <class T> T matrix_bijector(int kind, int inv, const T* in, T* out, int64_t K) {
  std::vector<T> A(K * K), B(K * K);
  if (kind == 0 || kind == 1) {
    if (!inv) {
      cholesky_upper<T>(in, A.data(), K);                                   // corr.jl:69, :133
      if (kind == 0) { link_chol_lkj_from_upper<T>(A.data(), out, K); return -logabsdetjac_inv_corr_vec<T>(out, K); }   // :135-137
      link_chol_lkj_mat<T>(A.data(), out, K);                               // :70
      return -logabsdetjac_inv_corr_mat<T>(out, K);                         // :92
    }
    T logJ = kind == 0 ? inv_link_chol_lkj<T>(in, A.data(), K) : inv_link_chol_lkj_mat<T>(in, A.data(), K);   // :140, :75
    for (int64_t j = 2; j <= K - 1; ++j) logJ += T(K - j) * std::log(A[(j - 1) * K + (j - 1)]);               // :144-146, :77-79
    pd_from_upper<T>(A.data(), out, K);
    return logJ;
  }
  if (!inv) {
    cholesky_lower<T>(in, A.data(), K);                                     // pd.jl:34
    T l = logabsdetjac_pdbijector_chol<T>(A.data(), K);
    for (int64_t i = 0; i < K; ++i) A[i * K + i] = std::log(A[i * K + i]);  // replace_diag(log, L), :11
    if (kind == 2) { std::memcpy(out, A.data(), sizeof(T) * K * K); return l; }
    int64_t idx = 0;                                                        // triu_to_vec(transpose(Y)), :41 ; src/utils.jl:68
    for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i <= j; ++i) out[idx++] = A[i * K + j];   // (Y')[i,j] = Y[j,i]
    return l;
  }
  if (kind == 2) { for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i < K; ++i) A[j * K + i] = i >= j ? in[j * K + i] : T(0); }
  else {                                                                    // transpose_eager(vec_to_triu(y)), pd.jl:44
    std::fill(A.begin(), A.end(), T(0));
    int64_t idx = 0;
    for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i <= j; ++i) A[i * K + j] = in[idx++];
  }
  for (int64_t i = 0; i < K; ++i) A[i * K + i] = std::exp(A[i * K + i]);    // replace_diag(exp, Y), :14 ; pd_from_lower drops the upper part
  pd_from_lower<T>(A.data(), out, K);
  cholesky_lower<T>(out, B.data(), K);                                      // interface.jl:278-281: -logabsdetjac(PDBijector(), x)
  return -logabsdetjac_pdbijector_chol<T>(B.data(), K);
}

// ---------------------------------------------------------------- F2 flows
// planar_layer.jl:65-70
template <class T> T get_u_hat(const T* u, const T* w, int64_t d, T* u_hat) {
  T wT_u = T(0);
  for (int64_t i = 0; i < d; ++i) wT_u += w[i] * u[i];  // dot(w, u)
  T w2 = jl_sum<T>([&](int64_t i) { return w[i] * w[i]; }, d);  // sum(abs2, w)
  T c = (log1pexp_<T>(-wT_u) - T(1)) / w2;
  for (int64_t i = 0; i < d; ++i) u_hat[i] = u[i] + c * w[i];
  return log1pexp_<T>(wT_u) - T(1);
}
// planar_layer.jl:73-80,102-110 : one layer, batch of columns
template <class T> void planar_fwd(const T* w, const T* u, T b, const T* Z, T* out, int64_t d, int64_t N, T* ladj) {
  std::vector<T> uh(d);
  T wT_uh = get_u_hat<T>(u, w, d, uh.data());
  for (int64_t n = 0; n < N; ++n) {
    const T* z = Z + n * d;
    T wT_z = T(0);
    for (int64_t i = 0; i < d; ++i) wT_z += w[i] * z[i];  // permutedims(w) * Z  (src/utils.jl:2)
    T t = std::tanh(wT_z + b);
    if (ladj) { T sech = T(1) / std::cosh(wT_z + b); ladj[n] = std::log1p(wT_uh * (sech * sech)); }
    for (int64_t i = 0; i < d; ++i) out[n * d + i] = z[i] + uh[i] * t;
  }
}
// planar_layer.jl:160-185.  Roots.A42 restated as bisection to adjacent floats (parity: residual only).
template <class T> T find_alpha(T wt_y, T wt_u_hat, T b) {
  T delta = T(2) * std::fabs(wt_u_hat);
  T lower = wt_y - delta, upper = wt_y + delta;
  if (lower == upper) return lower;  // :171-173
  auto f = [&](T a) { return a + wt_u_hat * std::tanh(a + b) - wt_y; };
  T flo = f(lower), fhi = f(upper);
  if (flo == T(0)) return lower;
  if (fhi == T(0)) return upper;
  for (int it = 0; it < 4096; ++it) {
    T mid = lower + (upper - lower) / T(2);
    if (!(mid > lower && mid < upper)) break;  // adjacent floats
    T fm = f(mid);
    if (fm == T(0)) return mid;
    if ((fm < T(0)) == (flo < T(0))) { lower = mid; flo = fm; } else { upper = mid; fhi = fm; }
  }
  return std::fabs(flo) <= std::fabs(fhi) ? lower : upper;
}
// planar_layer.jl:112-127; ladj via interface.jl:276-281
template <class T> void planar_inv(const T* w, const T* u, T b, const T* Y, T* out, int64_t d, int64_t N, T* ladj) {
  std::vector<T> uh(d);
  T wT_uh = get_u_hat<T>(u, w, d, uh.data());
  for (int64_t n = 0; n < N; ++n) {
    const T* y = Y + n * d;
    T wT_y = T(0);
    for (int64_t i = 0; i < d; ++i) wT_y += w[i] * y[i];
    T alpha = find_alpha<T>(wT_y, wT_uh, b);
    T t = std::tanh(alpha + b);
    for (int64_t i = 0; i < d; ++i) out[n * d + i] = y[i] - uh[i] * t;
  }
  if (ladj) {
    std::vector<T> tmp(d * N);
    planar_fwd<T>(w, u, b, out, tmp.data(), d, N, ladj);
    for (int64_t n = 0; n < N; ++n) ladj[n] = -ladj[n];
  }
}

// radial_layer.jl:43-72
template <class T> void radial_fwd(T alpha_, T beta, const T* z0, const T* Z, T* out, int64_t d, int64_t N, T* ladj) {
  T alpha = log1pexp_<T>(alpha_);
  T beta_hat = -alpha + log1pexp_<T>(beta);
  for (int64_t n = 0; n < N; ++n) {
    const T* z = Z + n * d;
    T ss = jl_sum<T>([&](int64_t i) { T dlt = z[i] - z0[i]; return dlt * dlt; }, d);
    T r = std::sqrt(ss);
    T h_ = T(1) / (alpha + r);
    for (int64_t i = 0; i < d; ++i) out[n * d + i] = z[i] + beta_hat / (alpha + r) * (z[i] - z0[i]);
    if (ladj) ladj[n] = T(d - 1) * std::log(T(1) + beta_hat * h_) + std::log(T(1) + beta_hat * h_ + beta_hat * (-(h_ * h_)) * r);
  }
}
// radial_layer.jl:88-129; ladj via interface.jl:276-281
template <class T> void radial_inv(T alpha_, T beta, const T* z0, const T* Y, T* out, int64_t d, int64_t N, T* ladj) {
  T alpha = log1pexp_<T>(alpha_);
  T apb = log1pexp_<T>(beta);
  for (int64_t n = 0; n < N; ++n) {
    const T* y = Y + n * d;
    T ss = jl_sum<T>([&](int64_t i) { T dlt = y[i] - z0[i]; return dlt * dlt; }, d);
    T gamma_n = std::sqrt(ss);               // norm(y_minus_z0)
    T a = apb - gamma_n;
    T r = (std::sqrt(a * a + 4 * alpha * gamma_n) - a) / 2;   // compute_r :124-129
    T g = (alpha + r) / (apb + r);
    for (int64_t i = 0; i < d; ++i) out[n * d + i] = z0[i] + g * (y[i] - z0[i]);
  }
  if (ladj) {
    std::vector<T> tmp(d * N);
    radial_fwd<T>(alpha_, beta, z0, out, tmp.data(), d, N, ladj);
    for (int64_t n = 0; n < N; ++n) ladj[n] = -ladj[n];
  }
}

// normalise.jl:41-68 (istraining() == false)
template <class T> void batchnorm_fwd(const T* b, const T* logs, const T* m, const T* v, T eps, const T* X, T* out, int64_t d, int64_t N, T* ladj) {
  for (int64_t n = 0; n < N; ++n)
    for (int64_t i = 0; i < d; ++i)
      out[n * d + i] = std::exp(logs[i]) * (X[n * d + i] - m[i]) / std::sqrt(v[i] + eps) + b[i];
  if (ladj) {
    T s = jl_sum<T>([&](int64_t i) { return logs[i] - std::log(v[i] + eps) / T(2); }, d);
    for (int64_t n = 0; n < N; ++n) ladj[n] = s;
  }
}
// normalise.jl:74-86
template <class T> void batchnorm_inv(const T* b, const T* logs, const T* m, const T* v, T eps, const T* Y, T* out, int64_t d, int64_t N, T* ladj) {
  for (int64_t n = 0; n < N; ++n)
    for (int64_t i = 0; i < d; ++i)
      out[n * d + i] = (Y[n * d + i] - b[i]) / std::exp(logs[i]) * std::sqrt(v[i] + eps) + m[i];
  if (ladj) {
    T s = jl_sum<T>([&](int64_t i) { return logs[i] - std::log(v[i] + eps) / T(2); }, d);
    for (int64_t n = 0; n < N; ++n) ladj[n] = -s;
  }
}

// ---------------------------------------------------------------- F4 RQS
// Base.searchsortedfirst(v, x): first 1-based index i with v[i] >= x, else length+1.
// v is row i of a [dim, K1] column-major matrix: v[k] = base[k*stride].
template <class T> inline int64_t searchsortedfirst_(const T* base, int64_t stride, int64_t len, T x) {
  int64_t lo = 0, hi = len + 1;  // Julia: lo = firstindex-1, hi = lastindex+1
  while (lo < hi - 1) {
    int64_t m = lo + ((hi - lo) >> 1);
    if (base[(m - 1) * stride] < x) lo = m; else hi = m;
  }
  return hi;
}
// rational_quadratic_spline.jl:317-357 (rqs_forward) == :128-164 (value) + :266-297 (logjac)
template <class T> inline void rqs_forward_(const T* w_, const T* h_, const T* d_, int64_t st, int64_t K, T x, T* y, T* lj) {
  auto W = [&](int64_t k) { return w_[(k - 1) * st]; };
  auto H = [&](int64_t k) { return h_[(k - 1) * st]; };
  auto D = [&](int64_t k) { return d_[(k - 1) * st]; };
  if ((x <= -W(K)) || (x >= W(K))) { *y = T(1) * x; *lj = T(0) * x; return; }
  int64_t k = searchsortedfirst_<T>(w_, st, K, x) - 1;
  T w_k = (k == 0) ? -W(K) : W(k);
  T w = W(k + 1) - w_k;
  T h_k = (k == 0) ? -H(K) : H(k);
  T dy = H(k + 1) - h_k;
  T s = dy / w;
  T xi = (x - w_k) / w;
  T d_k = (k == 0) ? T(1) : D(k);
  T d_k1 = (k == K - 1) ? T(1) : D(k + 1);
  T den = s + (d_k1 + d_k - 2 * s) * xi * (T(1) - xi);
  T num_jl = s * s * (d_k1 * (xi * xi) + 2 * s * xi * (T(1) - xi) + d_k * ((T(1) - xi) * (T(1) - xi)));
  *lj = std::log(num_jl) - 2 * std::log(den);
  T num_y = dy * (s * (xi * xi) + d_k * xi * (T(1) - xi));
  *y = h_k + num_y / den;
}
// rational_quadratic_spline.jl:183-220
template <class T> inline T rqs_inverse_(const T* w_, const T* h_, const T* d_, int64_t st, int64_t K, T y) {
  auto W = [&](int64_t k) { return w_[(k - 1) * st]; };
  auto H = [&](int64_t k) { return h_[(k - 1) * st]; };
  auto D = [&](int64_t k) { return d_[(k - 1) * st]; };
  if ((y <= -H(K)) || (y >= H(K))) return T(1) * y;
  int64_t k = searchsortedfirst_<T>(h_, st, K, y) - 1;
  T w_k = (k == 0) ? -W(K) : W(k);
  T w = W(k + 1) - w_k;
  T h_k = (k == 0) ? -H(K) : H(k);
  T dy = H(k + 1) - h_k;
  T s = dy / w;
  T d_k = (k == 0) ? T(1) : D(k);
  T d_k1 = (k == K - 1) ? T(1) : D(k + 1);
  T ds = d_k1 + d_k - 2 * s;
  T a1 = dy * (s - d_k) + (y - h_k) * ds;
  T a2 = dy * d_k - (y - h_k) * ds;
  T a3 = -s * (y - h_k);
  T num = -2 * a3;
  T den = (a2 + std::sqrt(a2 * a2 - 4 * a1 * a3));
  T xi = num / den;
  return xi * w + w_k;
}
// multivariate wrappers :173-178, :304-309, :363-367 applied per column; ladj = sum over rows
template <class T> void rqs_fwd(const T* w, const T* h, const T* d, int64_t K, const T* X, T* Y, int64_t dim, int64_t N, T* ladj) {
  std::vector<T> lj(dim);
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t i = 0; i < dim; ++i) rqs_forward_<T>(w + i, h + i, d + i, dim, K, X[n * dim + i], &Y[n * dim + i], &lj[i]);
    if (ladj) ladj[n] = jl_sum<T>([&](int64_t i) { return lj[i]; }, dim);
  }
}
template <class T> void rqs_inv(const T* w, const T* h, const T* d, int64_t K, const T* Y, T* X, int64_t dim, int64_t N, T* ladj) {
  std::vector<T> lj(dim);
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t i = 0; i < dim; ++i) {
      X[n * dim + i] = rqs_inverse_<T>(w + i, h + i, d + i, dim, K, Y[n * dim + i]);
      T yy; rqs_forward_<T>(w + i, h + i, d + i, dim, K, X[n * dim + i], &yy, &lj[i]);
    }
    if (ladj) ladj[n] = -jl_sum<T>([&](int64_t i) { return lj[i]; }, dim);   // interface.jl:276-281
  }
}
// rational_quadratic_spline.jl:109-123 (matrix B-constructor); raw_w, raw_h [dim,K], raw_d [dim,K-1]
template <class T> void rqs_params(const T* rw, const T* rh, const T* rd, int64_t K, int64_t dim, T B, T* w, T* h, T* d) {
  for (int64_t i = 0; i < dim; ++i) {
    for (int pass = 0; pass < 2; ++pass) {
      const T* r = pass == 0 ? rw : rh; T* o = pass == 0 ? w : h;
      T mx = r[i];
      for (int64_t k = 1; k < K; ++k) mx = std::max(mx, r[k * dim + i]);
      std::vector<T> e(K);
      T s = T(0);
      for (int64_t k = 0; k < K; ++k) { e[k] = std::exp(r[k * dim + i] - mx); s += e[k]; }
      T c = T(0);  // cumsum(hcat(0, softmax)); then (2B) .* c .- B
      o[i] = (2 * B) * c - B;
      for (int64_t k = 0; k < K; ++k) { c += e[k] / s; o[(k + 1) * dim + i] = (2 * B) * c - B; }
    }
    d[i] = T(1);
    for (int64_t k = 0; k < K - 1; ++k) d[(k + 1) * dim + i] = log1pexp_<T>(rd[k * dim + i]);
    d[K * dim + i] = T(1);
  }
}

// ---------------------------------------------------------------- F5
// permute.jl:152: y = A * x with A a permutation matrix; src[i] = column of the 1 in row i.
template <class T> void permute_fwd(const int32_t* src, const T* X, T* Y, int64_t dim, int64_t N) {
  for (int64_t n = 0; n < N; ++n)
    for (int64_t i = 0; i < dim; ++i) Y[n * dim + i] = X[n * dim + src[i]];
}
// coupling.jl:125-134,206-232 with the affine law b = Shift(t) ∘ Scale(s) applied to x_1.
template <class T> void coupling_affine(int inverse, const int32_t* idx1, int64_t n1, const T* scale, const T* shift,
                                       const T* X, T* Y, int64_t dim, int64_t N, T* ladj) {
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t i = 0; i < dim; ++i) Y[n * dim + i] = X[n * dim + i];   // A_2 x_2 + A_3 x_3
    T l = T(0);
    for (int64_t j = 0; j < n1; ++j) {
      T s = scale ? scale[n * n1 + j] : T(1), t = shift ? shift[n * n1 + j] : T(0);
      T x1 = X[n * dim + idx1[j]];
      if (!inverse) { Y[n * dim + idx1[j]] = t + s * x1; l += std::log(std::fabs(s)); }
      else { Y[n * dim + idx1[j]] = (T(1) / s) * (-t + x1); l -= std::log(std::fabs(s)); }
    }
    if (ladj) ladj[n] = l;
  }
}
template <class T> void coupling_rqs(int inverse, const int32_t* idx1, int64_t n1, const T* w, const T* h, const T* d, int64_t K,
                                    const T* X, T* Y, int64_t dim, int64_t N, T* ladj) {
  std::vector<T> lj(n1);
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t i = 0; i < dim; ++i) Y[n * dim + i] = X[n * dim + i];
    for (int64_t j = 0; j < n1; ++j) {
      T x1 = X[n * dim + idx1[j]];
      if (!inverse) rqs_forward_<T>(w + j, h + j, d + j, n1, K, x1, &Y[n * dim + idx1[j]], &lj[j]);
      else {
        T xo = rqs_inverse_<T>(w + j, h + j, d + j, n1, K, x1);
        T yy; rqs_forward_<T>(w + j, h + j, d + j, n1, K, xo, &yy, &lj[j]);
        lj[j] = -lj[j];
        Y[n * dim + idx1[j]] = xo;
      }
    }
    if (ladj) ladj[n] = jl_sum<T>([&](int64_t j) { return lj[j]; }, n1);
  }
}

}  // namespace

// ================================================================= C exports (host pointers)
#define BJO_EXPORT extern "C" __attribute__((visibility("default")))

#define BJO_INST(SUF, T)                                                                                        \
  BJO_EXPORT double bjo_chain_##SUF(const bjx_op* ops, int n, const T* x, T* y, int64_t dim, int64_t batch) {    \
    return (double)chain_fwd<T>(ops, n, x, y, dim, batch); }                                                     \
  BJO_EXPORT double bjo_chain_fused_##SUF(const bjx_op* ops, int n, const T* x, T* y, int64_t dim, int64_t b) {  \
    return chain_fused<T>(ops, n, x, y, dim, b); }                                                               \
  BJO_EXPORT void bjo_ordered_##SUF(int inv, const T* in, T* out, int64_t dim, int64_t batch, T* ladj) {         \
    if (inv) ordered_inv<T>(in, out, dim, batch, ladj); else ordered_fwd<T>(in, out, dim, batch, ladj); }        \
  BJO_EXPORT void bjo_simplex_##SUF(int inv, const T* in, T* out, int64_t K, int64_t N, T* ladj) {               \
    if (!inv) { if (out) simplex_fwd<T>(in, out, K, N);                                                          \
                if (ladj) for (int64_t n = 0; n < N; ++n) ladj[n] = simplex_ladj_col<T>(in + n * K, K); }        \
    else { simplex_inv<T>(in, out, K, N);                                                                        \
           if (ladj) for (int64_t n = 0; n < N; ++n) ladj[n] = -simplex_ladj_col<T>(out + n * K, K); } }         \
  BJO_EXPORT void bjo_vec_cholesky_##SUF(int inv, int uplo, const T* in, T* out, int64_t K, int64_t N, T* ladj) {\
    int64_t nv = K * (K - 1) / 2;                                                                                \
    std::vector<T> tmp(K * K);                                                                                   \
    for (int64_t n = 0; n < N; ++n) {                                                                            \
      if (inv) {                                                                                                 \
        T lj;                                                                                                    \
        if (!out) lj = logabsdetjac_inv_chol<T>(in + n * nv, K);                                                 \
        else if (uplo == 'U') lj = inv_link_chol_lkj<T>(in + n * nv, out + n * K * K, K);                        \
        else { lj = inv_link_chol_lkj<T>(in + n * nv, tmp.data(), K);   /* corr.jl:248 transpose_eager */        \
               for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i < K; ++i) out[n*K*K + j*K + i] = tmp[i*K + j]; } \
        if (ladj) ladj[n] = lj;                                                                                  \
      } else {                                                                                                   \
        const T* W = in + n * K * K;                                                                             \
        if (uplo != 'U') { for (int64_t j = 0; j < K; ++j) for (int64_t i = 0; i < K; ++i) tmp[j*K + i] = W[i*K + j]; W = tmp.data(); } \
        link_chol_lkj_from_upper<T>(W, out + n * nv, K);                                                         \
        if (ladj) ladj[n] = -logabsdetjac_inv_chol<T>(out + n * nv, K);   /* corr.jl:235-237 */                  \
      }                                                                                                          \
    } }                                                                                                          \
  BJO_EXPORT void bjo_matrix_bijector_##SUF(int kind, int inv, const T* in, T* out, int64_t K, int64_t N, T* ladj) { \
    int64_t nv = kind == 0 ? K * (K - 1) / 2 : (kind == 3 ? K * (K + 1) / 2 : K * K);                            \
    for (int64_t n = 0; n < N; ++n) {                                                                            \
      T l = matrix_bijector<T>(kind, inv, in + n * (inv ? nv : K * K), out + n * (inv ? K * K : nv), K);         \
      if (ladj) ladj[n] = l; } }                                                                                 \
  BJO_EXPORT double bjo_logabsdetjac_inv_corr_##SUF(int vec, const T* y, int64_t K) {                            \
    return (double)(vec ? logabsdetjac_inv_corr_vec<T>(y, K) : logabsdetjac_inv_corr_mat<T>(y, K)); }            \
  BJO_EXPORT void bjo_planar_##SUF(int inv, const T* w, const T* u, const T* b, int nl, const T* in, T* out,     \
                                   int64_t d, int64_t N, T* ladj) {                                              \
    std::vector<T> cur(in, in + d * N), nxt(d * N), l(N), acc(N, T(0));                                          \
    for (int k = 0; k < nl; ++k) {                                                                               \
      int L = inv ? nl - 1 - k : k;                                                                              \
      if (inv) planar_inv<T>(w + L * d, u + L * d, b[L], cur.data(), nxt.data(), d, N, l.data());                \
      else planar_fwd<T>(w + L * d, u + L * d, b[L], cur.data(), nxt.data(), d, N, l.data());                    \
      for (int64_t n = 0; n < N; ++n) acc[n] += l[n];                                                            \
      cur.swap(nxt);                                                                                             \
    }                                                                                                            \
    std::memcpy(out, cur.data(), sizeof(T) * d * N);                                                             \
    if (ladj) std::memcpy(ladj, acc.data(), sizeof(T) * N); }                                                    \
  BJO_EXPORT double bjo_find_alpha_##SUF(T wt_y, T wt_u_hat, T b) { return (double)find_alpha<T>(wt_y, wt_u_hat, b); } \
  BJO_EXPORT void bjo_radial_##SUF(int inv, T alpha_, T beta, const T* z0, const T* in, T* out, int64_t d, int64_t N, T* ladj) { \
    if (inv) radial_inv<T>(alpha_, beta, z0, in, out, d, N, ladj); else radial_fwd<T>(alpha_, beta, z0, in, out, d, N, ladj); } \
  BJO_EXPORT void bjo_batchnorm_##SUF(int inv, const T* b, const T* logs, const T* m, const T* v, T eps,         \
                                      const T* in, T* out, int64_t d, int64_t N, T* ladj) {                      \
    if (inv) batchnorm_inv<T>(b, logs, m, v, eps, in, out, d, N, ladj);                                          \
    else batchnorm_fwd<T>(b, logs, m, v, eps, in, out, d, N, ladj); }                                            \
  BJO_EXPORT void bjo_rqs_##SUF(int inv, const T* w, const T* h, const T* d, int64_t K1, const T* in, T* out,    \
                                int64_t dim, int64_t N, T* ladj) {                                               \
    if (inv) rqs_inv<T>(w, h, d, K1, in, out, dim, N, ladj); else rqs_fwd<T>(w, h, d, K1, in, out, dim, N, ladj); } \
  BJO_EXPORT void bjo_rqs_params_##SUF(const T* rw, const T* rh, const T* rd, int64_t K, int64_t dim, T B,       \
                                       T* w, T* h, T* d) { rqs_params<T>(rw, rh, rd, K, dim, B, w, h, d); }      \
  BJO_EXPORT void bjo_permute_##SUF(const int32_t* src, const T* in, T* out, int64_t dim, int64_t N) {           \
    permute_fwd<T>(src, in, out, dim, N); }                                                                      \
  BJO_EXPORT void bjo_coupling_affine_##SUF(int inv, const int32_t* idx1, int64_t n1, const T* s, const T* t,    \
                                            const T* in, T* out, int64_t dim, int64_t N, T* ladj) {              \
    coupling_affine<T>(inv, idx1, n1, s, t, in, out, dim, N, ladj); }                                            \
  BJO_EXPORT void bjo_coupling_rqs_##SUF(int inv, const int32_t* idx1, int64_t n1, const T* w, const T* h,       \
                                         const T* d, int64_t K1, const T* in, T* out, int64_t dim, int64_t N, T* ladj) { \
    coupling_rqs<T>(inv, idx1, n1, w, h, d, K1, in, out, dim, N, ladj); }                                        \
  BJO_EXPORT double bjo_logistic_##SUF(T x) { return (double)logistic_<T>(x); }                                  \
  BJO_EXPORT double bjo_log1pexp_##SUF(T x) { return (double)log1pexp_<T>(x); }                                  \
  BJO_EXPORT double bjo_logcosh_##SUF(T x) { return (double)logcosh_<T>(x); }

BJO_INST(f32, float)
BJO_INST(f64, double)

BJO_EXPORT int64_t bjo_triu1_dim_from_length(int64_t d) { return triu1_dim_from_length(d); }
BJO_EXPORT int bjo_version(void) { return BJX_VERSION; }
