// G2 instantiation of the MSM pipeline (no call site in keaki; requested by the north star).
#define KEAKI_FQ2_OUTLINE 1
#include "msm_host.cuh"
namespace keaki_internal {
keaki_status msm_g2_run(keaki_hip_ctx* ctx, const void* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac) {
  return msm_dev<Fq2>(ctx, (const G2Aff*)d_points, srs_len, d_scalars, n, d_out_jac);
}
}  // namespace keaki_internal
