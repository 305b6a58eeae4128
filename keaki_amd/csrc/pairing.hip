// Batched pairing + GT serialisation + BLAKE3 KDF kernels (reference src/kem.rs:30-46,58-69).
#define KEAKI_FQ2_OUTLINE 1
#include "internal.h"
#include "pairing.cuh"
namespace keaki_internal {
using namespace bn254;
keaki_status pairing_run(keaki_hip_ctx* ctx, const void* d_g1, const void* d_g2, int g2_stride, size_t n, void* d_gt, const void* d_fixed_lines,
                         uint32_t lines_stride) {
  // two lanes per pairing
  hipLaunchKernelGGL(k_pairing_batch, dim3(cdiv(2 * n, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)d_g1, (const G2Aff*)d_g2, g2_stride, (u32)n,
                     (const Line*)d_fixed_lines, lines_stride, (u32*)d_gt);
  return launch_check(ctx, "pairing_batch");
}
uint32_t g2_prepared_lines() { return (uint32_t)MILLER_MAX_LINES * 2; }
size_t g2_prepared_bytes() { return (size_t)MILLER_MAX_LINES * 2 * sizeof(Line); }
keaki_status g2_prepare_run(keaki_hip_ctx* ctx, const void* d_q, void* d_lines) {
  hipLaunchKernelGGL(k_g2_prepare, dim3(1), dim3(64), 0, ctx->stream, (const G2Aff*)d_q, (Line*)d_lines);
  return launch_check(ctx, "g2_prepare");
}
keaki_status pairing_raw_fixed_run(keaki_hip_ctx* ctx, const void* d_g1, size_t n, const void* d_lines, void* d_out) {
  hipLaunchKernelGGL(k_pairing_raw_fixed, dim3(cdiv(2 * n, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)d_g1, (u32)n, (const Line*)d_lines, (Fq*)d_out);
  return launch_check(ctx, "pairing_raw_fixed");
}
size_t gt_table_bytes(uint32_t wb) { GtShape g = gt_shape(wb); return (size_t)g.windows * g.entries * 12 * sizeof(Fq); }
uint32_t gt_table_powers(uint32_t wb) { GtShape g = gt_shape(wb); return g.wb * g.windows; }
// d_table[j][d] = base^(d 2^(wb j)), d = 1 .. 2^(wb-1), base = e(P, Q): d_pows = e(2^s P, Q), s < gt_table_powers(wb), 12 Fq each
keaki_status gt_table_run(keaki_hip_ctx* ctx, const void* d_pows, void* d_table, uint32_t wb) {
  const GtShape g = gt_shape(wb);
  hipLaunchKernelGGL(k_gt_table_scatter, dim3(cdiv(g.wb * g.windows * 12, 256)), dim3(256), 0, ctx->stream, (const Fq*)d_pows, (Fq*)d_table, g);
  for (u32 L = 1; L + 2 <= g.wb; L++)
    hipLaunchKernelGGL(k_gt_table_fill, dim3(cdiv(2 * g.windows * ((1u << L) - 1u), 64)), dim3(64), 0, ctx->stream, (Fq*)d_table, L, g);
  return launch_check(ctx, "gt_table");
}
keaki_status gt_encap_exp_run(keaki_hip_ctx* ctx, const void* d_tab_a, uint32_t wb_a, const void* d_tab_b, uint32_t wb_b, const void* d_betas,
                              const void* d_rs, size_t n, void* d_gt) {
  hipLaunchKernelGGL(k_gt_encap_exp, dim3(cdiv(2 * n, 64)), dim3(64), 0, ctx->stream, (const Fq*)d_tab_a, gt_shape(wb_a), (const Fq*)d_tab_b, gt_shape(wb_b),
                     (const Fr*)d_betas, (const Fr*)d_rs, (u32)n, (u32*)d_gt);
  return launch_check(ctx, "gt_encap_exp");
}
keaki_status miller_only_run(keaki_hip_ctx* ctx, const void* d_g1, const void* d_g2, size_t n, void* d_out) {
  hipLaunchKernelGGL(k_miller_only, dim3(cdiv(2 * n, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)d_g1, (const G2Aff*)d_g2, (u32)n, (Fq*)d_out);
  return launch_check(ctx, "miller_only");
}
keaki_status final_exp_only_run(keaki_hip_ctx* ctx, const void* d_in, size_t n, void* d_gt) {
  hipLaunchKernelGGL(k_final_exp_only, dim3(cdiv(2 * n, 64)), dim3(64), 0, ctx->stream, (const Fq*)d_in, (u32)n, (u32*)d_gt);
  return launch_check(ctx, "final_exp_only");
}
keaki_status blake3_gt_run(keaki_hip_ctx* ctx, const void* d_gt, size_t n, void* d_key, size_t msg_len) {
  hipLaunchKernelGGL(k_blake3_gt_xof, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const u32*)d_gt, (u32)n, (unsigned char*)d_key, (u32)msg_len);
  return launch_check(ctx, "blake3_gt_xof");
}
}  // namespace keaki_internal
