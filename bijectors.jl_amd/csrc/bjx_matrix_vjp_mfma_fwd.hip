// bjx_matrix_vjp_mfma_fwd.hip — the Float32 instantiations of matrix_fwd_vjp_mfma_kernel (bjx_matrix_vjp_mfma_fwd.inc) and the dispatcher.
#include "bjx_matrix_vjp_mfma_fwd.inc"

namespace bjx {
int bjx_matrix_fwd_vjp_mfma_f64(bjx_ctx* ctx, int kind, const double* in, const double* out_bar, const double* ladj_bar, double* in_bar, int64_t K, int64_t batch);   // bjx_matrix_vjp_mfma_fwd_f64.hip
}

namespace bjx {

// 1: not served (the caller goes on to the lane = row group kernel).  Served where it is the faster one on the same box (2^14 .. 2^19
// samples, Float32, percent of the HBM peak, group kernel -> this one, VecCorr / PDVec): K = 16: 31 / 33 -> 32 / 36, 32: 19 / 20 -> 25 / 28,
// 48: 12 / 13 -> 13 / 23, 64: 7 / 7 -> 14 / 16; NOT at K <= 12 (30 / 33 -> 16 / 18: a 16 x 16 block algebra on a 12 x 12 problem, four
// samples a wave one after the other) nor at 17 .. 24 (22 / 24 -> 15 / 21).  Float64 (call times of scripts/probe_matrix_vjp.py, group -> this): K = 32 0.71 / 0.61 -> 0.50 / 0.42 ms, K = 16 the same, 12 and 24 slower: from 25 rows.  Float64 at 33 .. 64 rows had only the one-lane workspace kernel: 18.6 / 47.7 ms -> 0.43 / 2.0 ms at K = 48 / 64, 2^13 samples.
int bjx_matrix_fwd_vjp_mfma(bjx_ctx* ctx, bjx_dtype dt, int kind, const void* in, const void* out_bar, const void* ladj_bar, void* in_bar, int64_t K, int64_t batch) {
  static const int use = getenv("BJX_MATRIX_VJP_MFMA") ? atoi(getenv("BJX_MATRIX_VJP_MFMA")) : 1;      // 0: the lane = row group kernel (its A/B); 2: every shape this kernel can do
  if (!use || K < 9 || K > 64) return 1;
  if (use != 2 && (K < 13 || (K > 16 && K < 25) || (dt != BJX_F32 && K < 25))) return 1;      // (K > 32 in Float64: nothing else but the workspace kernel)
  if (dt == BJX_F32) return fw_kind<float>(ctx, kind, (const float*)in, (const float*)out_bar, (const float*)ladj_bar, (float*)in_bar, K, batch);
  return bjx_matrix_fwd_vjp_mfma_f64(ctx, kind, (const double*)in, (const double*)out_bar, (const double*)ladj_bar, (double*)in_bar, K, batch);
}

}  // namespace bjx
