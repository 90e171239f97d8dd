// Per-column walkers (Ordered / Simplex): shared by bjx_seq.hip (one structured bijector per launch) and bjx_stacked.hip (Stacked
// with structured blocks in one launch).  Included INSIDE the anonymous namespace of each translation unit (after
// `using namespace bjx;`).  The [64][P] tile staging they run on is in bjx_tile.h.
// ---- per-column walkers.  A column is walked in ascending row order: first(v) for row 0,
// mid(i, v, log(K-1-i)) for the interior rows, last(v) for the final row when the op has a special
// one (HAS_LAST).  The split keeps the interior loop free of row-index branches, so four rows are
// in flight per lane (the only carried dependency is the running sum).  step() is the generic form
// used by the chunked kernel.
template <class T, class Op> __device__ __forceinline__ T seq_step(Op& op, int64_t i, int64_t rows, T v, const T* logk) {
  if (i == 0) return op.first(v, logk);
  if (Op::HAS_LAST && i == rows - 1) return op.last(v);
  return op.mid((int)i, v, Op::USES_LOGK ? logk[i] : T(0));
}
template <class T> struct OrderedFwd {   // ordered.jl:36-49, :80
  static constexpr bool HAS_LAST = false, USES_LOGK = false;
  T prev, ladj;
  __device__ void init() { prev = T(0); ladj = T(0); }
  __device__ T first(T v, const T*) { prev = v; return v; }
  // Fast<T>::exp = the exp of every other kernel (chain, quad_stream): one input gives the same bits whatever kernel a shape selects
  __device__ T mid(int, T v, T) { const T o = prev + Fast<T>::exp(v); ladj += v; prev = o; return o; }
  __device__ T last(T v) { return v; }
  __device__ T result() const { return ladj; }
};
template <class T> struct OrderedInv {   // ordered.jl:63-77 ; interface.jl:276-281
  static constexpr bool HAS_LAST = false, USES_LOGK = false;
  T prev, ladj;
  __device__ void init() { prev = T(0); ladj = T(0); }
  __device__ T first(T v, const T*) { prev = v; return v; }
  __device__ T mid(int, T v, T) { const T o = Fast<T>::log(v - prev); ladj -= o; prev = v; return o; }
  __device__ T last(T v) { return v; }
  __device__ T result() const { return ladj; }
};
// simplex.jl:47-64 (transform) fused with :122-138 (logabsdetjac); lk = log(T(K-1-i)).
// The kernel is VALU-bound with exact OCML logs/divisions (118 VALU per element, PMC in
// profiles/), so Float32 uses the hardware log/rcp units (Fast<T>), logit(z) = log(a/(d-a)) for
// z = a/d needs one reciprocal instead of two, the three logs of one log-det term are merged into
// the log of their product (>= eps^3, no underflow) and log2 values are summed (x ln 2 once).
template <class T, bool LADJ> struct SimplexFwd {
  static constexpr bool HAS_LAST = true, USES_LOGK = true;
  int64_t K;
  T sum_tmp, lp;   // lp accumulates log2 terms
  // Float64: a lean logarithm is ~40 operations and the log-det term of a row was one of the two per element.  The terms of a column are
  // SUMMED, so four of them (each >= eps³ = 1e-47: their product cannot underflow) are multiplied before ONE logarithm is taken.
  static constexpr bool PROD = sizeof(T) == 8;
  T pr;
  int np;
  __device__ void term(T v) {
    if constexpr (PROD) {
      pr *= v;
      if (++np == 4) { lp += Fast<T>::log2(pr); pr = T(1); np = 0; }
    } else {
      lp += Fast<T>::log2(v);
    }
  }
  __device__ void init() { sum_tmp = T(0); lp = T(0); pr = T(1); np = 0; }
  __device__ T first(T x, const T* logk) {
    using F = Fast<T>;
    const T e = Num<T>::eps;
    sum_tmp = x;                                                            // Σ_{j<1} x_j for the next row
    if (K < 2) return T(0);
    const T z = x * (T(1) - 2 * e) + e;                                     // :53
    if (LADJ) term(d_max(x, e) * d_max(T(1) - x, e));                       // :130-131
    return F::log2(z * F::rcp(T(1) - z)) * Num<T>::log2 + logk[0];          // logit(z) + log(K-1)
  }
  __device__ T mid(int, T x, T lk) {
    using F = Fast<T>;
    const T e = Num<T>::eps;
    const T s = sum_tmp;
    sum_tmp = s + x;
    const T a = (x + e) * (T(1) - 2 * e);                                   // z = a / ((1+ε) - Σ)   (:58)
    const T dn = (T(1) + e) - s;
    const T o = F::log2(a * F::rcp(dn - a)) * Num<T>::log2 + lk;            // logit(z) + log(K-1-i)
    if (LADJ) {
      const T m = d_max(T(1) - s, e);                                       // :133
      const T zl = x * F::rcp(m);                                           // :134
      term(d_max(zl, e) * d_max(T(1) - zl, e) * m);                         // :135
    }
    return o;
  }
  __device__ T last(T) { return T(0); }                                     // row K has no output
  // Julia's max(NaN, ε) is NaN (v_max drops it): a NaN among x_1..x_{K-1} makes the reference's log-det NaN
  __device__ T result() const {
    T l = lp;
    if constexpr (PROD) l += Fast<T>::log2(pr);                             // (log2(1) = 0 when nothing is pending)
    return sum_tmp != sum_tmp ? sum_tmp : -l * Num<T>::log2;
  }
};
// simplex.jl:102-120 ; log-det = -logabsdetjac(b, x_out)
template <class T, bool LADJ> struct SimplexInv {
  static constexpr bool HAS_LAST = true, USES_LOGK = true;
  int64_t K;
  T sum_tmp, lp;
  static constexpr bool PROD = sizeof(T) == 8;          // as in SimplexFwd: four log-det terms per logarithm in Float64
  T pr;
  int np;
  __device__ void term(T v) {
    if constexpr (PROD) {
      pr *= v;
      if (++np == 4) { lp += Fast<T>::log2(pr); pr = T(1); np = 0; }
    } else {
      lp += Fast<T>::log2(v);
    }
  }
  __device__ void init() { sum_tmp = T(0); lp = T(0); pr = T(1); np = 0; }
  static __device__ __forceinline__ T logistic(T v) {   // LogExpFunctions.logistic with its exact 0/1 saturation
    return f_logistic(v);
  }
  __device__ T first(T y, const T* logk) {
    using F = Fast<T>;
    const T e = Num<T>::eps;
    if (K < 2) return T(1);
    const T inv12e = T(1) / (T(1) - 2 * e);
    const T z = logistic(y - logk[0]);
    const T x = d_clamp((z - e) * inv12e, T(0), T(1));                      // :109
    if (LADJ) term(d_max(x, e) * d_max(T(1) - x, e));
    sum_tmp = x;
    return x;
  }
  __device__ T mid(int, T y, T lk) {
    using F = Fast<T>;
    const T e = Num<T>::eps;
    const T inv12e = T(1) / (T(1) - 2 * e);
    const T z = logistic(y - lk);
    const T s = sum_tmp;
    const T x = d_clamp(((T(1) + e) - s) * inv12e * z - e, T(0), T(1));     // :113
    sum_tmp = s + x;
    if (LADJ) {
      const T m = d_max(T(1) - s, e);
      const T zl = x * F::rcp(m);
      term(d_max(zl, e) * d_max(T(1) - zl, e) * m);
    }
    return x;
  }
  __device__ T last(T) { return d_clamp(T(1) - sum_tmp, T(0), T(1)); }      // :116
  __device__ T result() const {                                             // NaN rows: as above
    T l = lp;
    if constexpr (PROD) l += Fast<T>::log2(pr);
    return sum_tmp != sum_tmp ? sum_tmp : l * Num<T>::log2;
  }
};

// partials of one log-det term t_k(x_k, s_k) of simplex.jl:122-138 (s_k = Σ_{j<k} x_j): used by every Simplex pullback kernel
template <class T> __device__ __forceinline__ void simplex_t_partials(T xk, T sk, bool first, T& dtdx, T& dtds) {
  using F = Fast<T>;
  const T e = Num<T>::eps;
  if (first) {
    dtdx = (xk > e ? F::rcp(xk) : T(0)) - (T(1) - xk > e ? F::rcp(T(1) - xk) : T(0));
    dtds = T(0);
    return;
  }
  const T M = d_max(T(1) - sk, e);
  const T rM = F::rcp(M);
  const T zl = xk * rM;
  const T dtdzl = (zl > e ? F::rcp(zl) : T(0)) - (T(1) - zl > e ? F::rcp(T(1) - zl) : T(0));
  dtdx = dtdzl * rM;
  const T dtdM = rM - dtdzl * zl * rM;                 // d/dM [log max(zl) + log max(1-zl) + log M], zl = x/M
  dtds = (T(1) - sk > e) ? -dtdM : T(0);
}

