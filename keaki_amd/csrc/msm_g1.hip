// G1 instantiation of the MSM pipeline (reference src/kzg.rs:98) + partial-sum combine.
#include "msm_host.cuh"
namespace keaki_internal {
keaki_status msm_g1_run(keaki_hip_ctx* ctx, const void* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac) {
  return msm_dev<Fq>(ctx, (const G1Aff*)d_points, srs_len, d_scalars, n, d_out_jac);
}
keaki_status g1_sum_run(keaki_hip_ctx* ctx, const void* d_points_jac, size_t k, void* d_out_jac) {
  hipLaunchKernelGGL((k_sum_jac<Fq>), dim3(1), dim3(64), 0, ctx->stream, (const Fq*)d_points_jac, (u32)k, (Fq*)d_out_jac);
  return launch_check(ctx, "g1_sum");
}
}  // namespace keaki_internal
