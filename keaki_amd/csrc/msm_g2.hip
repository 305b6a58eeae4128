// G2 instantiation of the MSM pipeline (no call site in keaki; requested by the north star).
#define KEAKI_FQ2_OUTLINE 1
#include "msm_host.hip.h"
namespace keaki_internal {
keaki_status msm_g2_run(keaki_hip_ctx* ctx, const void* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac, const void* d_table,
                        int c_table, const MsmPipe* pipe) {
  return msm_dev<Fq2>(ctx, (const G2Aff*)d_points, srs_len, d_scalars, n, d_out_jac, (const G2Aff*)d_table, c_table, pipe);
}
// window tables of a fixed G2 basis: all windows share one bucket set, and the per-window Horner doublings -- a serial chain of ~240
// Fq2 doublings on one lane, 4.4 ms -- disappear (profiles/r02_msm_g2_kernel_stats.csv)
keaki_status msm_g2_precompute_run(keaki_hip_ctx* ctx, const void* d_points, size_t N, int* c_table_out, size_t* table_bytes_out, void** d_table_out) {
  const int c = choose_window_shared(N, ctx->tune.msm_c_shared);
  const size_t bytes = (size_t)msm_plan_windows(N, c) * N * sizeof(G2Aff);
  void* t = nullptr;
  ST_TRY(dev_alloc(ctx, &t, bytes ? bytes : 64));
  keaki_status st = msm_build_tables<Fq2>(ctx, (const G2Aff*)d_points, N, c, (G2Aff*)t);
  if (st != KEAKI_OK) { (void)hipFree(t); return st; }
  *c_table_out = c; *table_bytes_out = bytes; *d_table_out = t;
  return KEAKI_OK;
}
}  // namespace keaki_internal
