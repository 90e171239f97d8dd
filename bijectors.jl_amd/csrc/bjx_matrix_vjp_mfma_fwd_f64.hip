// bjx_matrix_vjp_mfma_fwd_f64.hip — the Float64 instantiations of matrix_fwd_vjp_mfma_kernel (bjx_matrix_vjp_mfma_fwd.inc).
#include "bjx_matrix_vjp_mfma_fwd.inc"

namespace bjx {
int bjx_matrix_fwd_vjp_mfma_f64(bjx_ctx* ctx, int kind, const double* in, const double* out_bar, const double* ladj_bar, double* in_bar, int64_t K, int64_t batch) {
  return fw_kind<double>(ctx, kind, in, out_bar, ladj_bar, in_bar, K, batch);
}
}  // namespace bjx
