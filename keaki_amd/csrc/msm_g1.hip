// G1 instantiation of the MSM pipeline (reference src/kzg.rs:98) + partial-sum combine.
#include "msm_host.hip.h"
namespace keaki_internal {
keaki_status msm_g1_run(keaki_hip_ctx* ctx, const void* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac,
                        const void* d_table, int c_table, const MsmPipe* pipe) {
  return msm_dev<Fq>(ctx, (const G1Aff*)d_points, srs_len, d_scalars, n, d_out_jac, (const G1Aff*)d_table, c_table, pipe);
}
keaki_status msm_g1_precompute_run(keaki_hip_ctx* ctx, const void* d_points, size_t N, int* c_table_out, size_t* table_bytes_out, void** d_table_out) {
  const int c = choose_window_shared(N, ctx->tune.msm_c_shared);
  const size_t bytes = (size_t)msm_plan_windows(N, c) * N * sizeof(G1Aff);
  void* t = nullptr;
  ST_TRY(dev_alloc(ctx, &t, bytes ? bytes : 64));
  keaki_status st = msm_build_tables<Fq>(ctx, (const G1Aff*)d_points, N, c, (G1Aff*)t);
  if (st != KEAKI_OK) { (void)hipFree(t); return st; }
  *c_table_out = c; *table_bytes_out = bytes; *d_table_out = t;
  return KEAKI_OK;
}
keaki_status g1_sum_run(keaki_hip_ctx* ctx, const void* d_points_jac, size_t k, void* d_out_jac) {
  hipLaunchKernelGGL((k_sum_jac<Fq>), dim3(1), dim3(64), 0, ctx->stream, (const Fq*)d_points_jac, (u32)k, (Fq*)d_out_jac);
  return launch_check(ctx, "g1_sum");
}
}  // namespace keaki_internal
