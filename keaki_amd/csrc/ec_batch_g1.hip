// G1 batched scalar multiplication and the G1 half of encapsulate (reference src/kem.rs:22,30).
#include "ec_batch.hip.h"
#include "internal.h"
namespace keaki_internal {
using namespace bn254;
keaki_status g1_mul_batch_run(keaki_hip_ctx* ctx, const void* d_pts, int stride, const void* d_scalars, size_t n, void* d_out) {
  hipLaunchKernelGGL((k_mul_batch<Fq>), dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)d_pts, stride, (const Fr*)d_scalars, (u32)n,
                     (G1Aff*)d_out);
  return launch_check(ctx, "g1_mul_batch");
}
keaki_status encap_g1_run(keaki_hip_ctx* ctx, const void* d_com, const void* d_values, const void* d_r, size_t n, void* d_out) {
  hipLaunchKernelGGL(k_encap_g1, dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)d_com, (const Fr*)d_values, (const Fr*)d_r, (u32)n,
                     (G1Aff*)d_out);
  return launch_check(ctx, "encap_g1");
}
// (a kernel, not hipMemcpyFromSymbol: two device variables named from the host came out of hipcc in either order -- see ec_batch_g2.hip)
static __global__ void k_g1_generator_to(Fq* __restrict__ dst) {
  if (threadIdx.x == 0) { dst[0] = G1_GEN_X; dst[1] = G1_GEN_Y; }
}
keaki_status g1_generator_to(keaki_hip_ctx* ctx, void* d_dst) {
  hipLaunchKernelGGL(k_g1_generator_to, dim3(1), dim3(64), 0, ctx->stream, (Fq*)d_dst);
  return launch_check(ctx, "g1_generator_to");
}
// table[j * entries + d] = d 2^(wb j) * base
keaki_status g1_fb_table_run(keaki_hip_ctx* ctx, const void* d_base, void* d_table, uint32_t wb) {
  const FbShape g = fb_shape(wb);
  ST_TRY(reserve(ctx, ctx->fb_bases, 64 * sizeof(G2Aff)));
  hipLaunchKernelGGL((k_fb_window_bases<Fq>), dim3(1), dim3(64), 0, ctx->stream, (const G1Aff*)d_base, g, (G1Aff*)ctx->fb_bases.p);
  hipLaunchKernelGGL((k_fb_table_entries<Fq>), dim3(cdiv((size_t)g.windows * g.entries, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)ctx->fb_bases.p, g,
                     (G1Aff*)d_table);
  return launch_check(ctx, "g1_fb_table");
}
keaki_status encap_g1_fixed_run(keaki_hip_ctx* ctx, const void* d_tab_a, uint32_t wb_a, const void* d_tab_b, uint32_t wb_b, const void* d_xs,
                                const void* d_rs, size_t n, void* d_out) {
  hipLaunchKernelGGL((k_encap_fixed<Fq, 1>), dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)d_tab_a, fb_shape(wb_a), (const G1Aff*)d_tab_b,
                     fb_shape(wb_b), (const Fr*)d_xs,
                     (const Fr*)d_rs, (u32)n, (G1Aff*)d_out);
  return launch_check(ctx, "encap_g1_fixed");
}
keaki_status g1_curve_check_run(keaki_hip_ctx* ctx, const void* d_pts, size_t n, void* d_bad2) {
  hipLaunchKernelGGL((k_curve_check<Fq>), dim3(cdiv(n ? n : 1, 256)), dim3(256), 0, ctx->stream, (const G1Aff*)d_pts, (u32)n,
                     (unsigned long long*)d_bad2, (unsigned long long*)d_bad2 + 1);
  return launch_check(ctx, "g1_curve_check");
}
size_t fb_table_entries(uint32_t wb) { FbShape g = fb_shape(wb); return (size_t)g.windows * g.entries; }
}  // namespace keaki_internal
