#include <cstdlib>
// G2 batched scalar multiplication and the G2 half of encapsulate (reference src/kem.rs:36-37).
#define KEAKI_FQ2_OUTLINE 1
#include "ec_batch.hip.h"
#include "internal.h"
namespace keaki_internal {
using namespace bn254;
keaki_status g2_mul_batch_run(keaki_hip_ctx* ctx, const void* d_pts, int stride, const void* d_scalars, size_t n, void* d_out) {
  hipLaunchKernelGGL((k_mul_batch<Fq2>), dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, (const G2Aff*)d_pts, stride, (const Fr*)d_scalars, (u32)n,
                     (G2Aff*)d_out);
  return launch_check(ctx, "g2_mul_batch");
}
// (a kernel, not hipMemcpyFromSymbol: a device variable named from the host is externalised under a per-compilation `__hip_cuid_` symbol whose place
// in the symbol table changed from build to build -- this object was the one file that kept a clean build from being byte-identical to the last)
static __global__ void k_g2_generator_to(Fq2* __restrict__ dst) {
  if (threadIdx.x < 2) dst[threadIdx.x] = G2_GEN_XY[threadIdx.x];
}
keaki_status g2_generator_to(keaki_hip_ctx* ctx, void* d_dst) {
  hipLaunchKernelGGL(k_g2_generator_to, dim3(1), dim3(64), 0, ctx->stream, (Fq2*)d_dst);
  return launch_check(ctx, "g2_generator_to");
}
// table[j*256+d] = d 2^(8j) * base   (8192 affine entries)
keaki_status g2_fb_table_run(keaki_hip_ctx* ctx, const void* d_base, void* d_table, uint32_t wb) {
  const FbShape g = fb_shape(wb);
  ST_TRY(reserve(ctx, ctx->fb_bases, 64 * sizeof(G2Aff)));
  hipLaunchKernelGGL((k_fb_window_bases<Fq2>), dim3(1), dim3(64), 0, ctx->stream, (const G2Aff*)d_base, g, (G2Aff*)ctx->fb_bases.p);
  hipLaunchKernelGGL((k_fb_table_entries<Fq2>), dim3(cdiv((size_t)g.windows * g.entries, 64)), dim3(64), 0, ctx->stream, (const G2Aff*)ctx->fb_bases.p, g,
                     (G2Aff*)d_table);
  return launch_check(ctx, "g2_fb_table");
}
keaki_status verify_points_run(keaki_hip_ctx* ctx, const void* d_tab_g1, const void* d_tab_g2, uint32_t wb, const void* d_com, const void* d_tau_g2,
                               const void* d_value, const void* d_point, void* d_out_a, void* d_out_q) {
  hipLaunchKernelGGL(k_verify_points, dim3(1), dim3(128), 0, ctx->stream, (const G1Aff*)d_tab_g1, (const G2Aff*)d_tab_g2, fb_shape(wb), (const G1Aff*)d_com,
                     (const G2Aff*)d_tau_g2, (const Fr*)d_value, (const Fr*)d_point, (G1Aff*)d_out_a, (G2Aff*)d_out_q);
  return launch_check(ctx, "verify_points");
}
keaki_status g2_pow2_multiples_run(keaki_hip_ctx* ctx, const void* d_base, uint32_t count, void* d_out) {
  const FbShape g = {1u, count, 0u};                 // "windows" of one bit: base_s = 2^s base
  hipLaunchKernelGGL((k_fb_window_bases<Fq2>), dim3(cdiv(count, 64)), dim3(64), 0, ctx->stream, (const G2Aff*)d_base, g, (G2Aff*)d_out);
  return launch_check(ctx, "g2_pow2_multiples");
}
// share_simds: a latency-bound job of this context runs beside this kernel (the table of a new commitment): the 256-register form, whose waves
// can sit on a SIMD next to that job's, also for batches that would otherwise take the 285-register one
keaki_status encap_g2_fixed_run(keaki_hip_ctx* ctx, const void* d_tab_a, uint32_t wb_a, const void* d_tab_b, uint32_t wb_b, const void* d_xs,
                                const void* d_rs, size_t n, void* d_out, bool share_simds) {
  // 285 registers per lane: one wave per SIMD. Batches that fill every SIMD more than once do better with two waves and 29 spilled
  // registers (2^20 items: 32.2 -> 29.1 ms per encap batch); up to one wave per SIMD (2^16 items) the unspilled kernel wins by 3 %.
  const size_t wide_max = ctx->tune.pair_wide_max < 0 ? (size_t)2048 : (size_t)ctx->tune.pair_wide_max;
  if (n <= wide_max && fb_shape(wb_a).windows + fb_shape(wb_b).windows <= FBW_SLOTS * 16u) {
    // few items: sixteen lanes per item (the additions of an item side by side, then a tree): a single encapsulate 0.96 -> 0.3 ms
    hipLaunchKernelGGL(k_encap_fixed_g2_wide, dim3(cdiv(n, 4)), dim3(64), 0, ctx->stream, (const G2Aff*)d_tab_a, fb_shape(wb_a), (const G2Aff*)d_tab_b,
                       fb_shape(wb_b), (const Fr*)d_xs, (const Fr*)d_rs, (u32)n, (G2Aff*)d_out);
    return launch_check(ctx, "encap_g2_fixed");
  }
  if ((n <= 65536 && !share_simds) || ctx->tune.fb_occ1) {
    hipLaunchKernelGGL((k_encap_fixed<Fq2, 1>), dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, (const G2Aff*)d_tab_a, fb_shape(wb_a), (const G2Aff*)d_tab_b,
                     fb_shape(wb_b), (const Fr*)d_xs, (const Fr*)d_rs, (u32)n, (G2Aff*)d_out);
  } else {
    hipLaunchKernelGGL((k_encap_fixed<Fq2, 2>), dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, (const G2Aff*)d_tab_a, fb_shape(wb_a), (const G2Aff*)d_tab_b,
                     fb_shape(wb_b), (const Fr*)d_xs, (const Fr*)d_rs, (u32)n, (G2Aff*)d_out);
  }
  return launch_check(ctx, "encap_g2_fixed");
}
keaki_status g2_curve_check_run(keaki_hip_ctx* ctx, const void* d_pts, size_t n, void* d_bad2) {
  hipLaunchKernelGGL((k_curve_check<Fq2>), dim3(cdiv(n ? n : 1, 256)), dim3(256), 0, ctx->stream, (const G2Aff*)d_pts, (u32)n,
                     (unsigned long long*)d_bad2, (unsigned long long*)d_bad2 + 1);
  return launch_check(ctx, "g2_curve_check");
}
}  // namespace keaki_internal
