// Batched pairing + GT serialisation + BLAKE3 KDF kernels (reference src/kem.rs:30-46,58-69).
#define KEAKI_FQ2_OUTLINE 1
#include "internal.h"
#include "pairing.hip.h"
#include "pairing_wide.hip.h"
namespace keaki_internal {
using namespace bn254;
// One launch of k_pairing per chunk of at most PAIR_CHUNK items: the final exponentiation keeps FE_NSLOTS Fq12 values per item in HBM
// (3.75 KB per item; 2^17 items = 480 MB, grow-only in the context).
constexpr size_t PAIR_CHUNK = (size_t)1 << 17;
constexpr size_t PAIR_WIDE_AUTO = 4096;
constexpr size_t PAIR_WIDE2_AUTO = 1024;       // 72 KB of LDS per two-wave workgroup: two per CU
constexpr size_t GT_EXP_WIDE_AUTO = 2048;      // items of an encapsulation batch whose GT exponentiations run twelve lanes per item
static keaki_status pairing_launch(keaki_hip_ctx* ctx, u32 mode, const void* d_g1, const void* d_g2, int g2_stride, const void* d_f_in, size_t n,
                                   void* d_out, size_t out_item_bytes, const void* d_fixed_lines, uint32_t lines_stride, const char* what,
                                   uint32_t p_stride = 1) {
  if (n == 0) return KEAKI_OK;
  // Few pairings: the twelve-lanes-per-pairing kernel (pairing_wide.hip.h) -- a third of the latency for 2.5 x the wave-instructions, so only
  // while its waves (four pairings each) find the device not full: automatic = up to PAIR_WIDE_AUTO items.
  const size_t wide_max = ctx->tune.pair_wide_max < 0 ? PAIR_WIDE_AUTO : (size_t)ctx->tune.pair_wide_max;
  if (n <= wide_max) {
    PairArgs a;
    a.ps = (const G1Aff*)d_g1; a.p_stride = p_stride; a.qs = (const G2Aff*)d_g2; a.q_stride = g2_stride; a.n = (u32)n;
    a.fixed_lines = (const Line*)d_fixed_lines; a.lines_stride = lines_stride; a.f_in = (const Fq*)d_f_in;
    a.ws = nullptr; a.ws_n = 0; a.out = d_out; a.mode = mode;
    // few enough pairings that a second wave per workgroup finds room: the line functions (or, with tabulated lines, the line's product forms)
    // on a wave of their own
    if ((mode & PAIR_MILLER) && !d_fixed_lines && n <= PAIR_WIDE2_AUTO && ctx->tune.pair_two_waves)
      hipLaunchKernelGGL(pw::k_pairing_wide2<false>, dim3(cdiv(n, 4)), dim3(128), 0, ctx->stream, a);
    else if ((mode & PAIR_MILLER) && d_fixed_lines && n <= PAIR_WIDE2_AUTO && ctx->tune.pair_two_waves)
      hipLaunchKernelGGL(pw::k_pairing_wide2<true>, dim3(cdiv(n, 4)), dim3(128), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL(pw::k_pairing_wide, dim3(cdiv(n, 4)), dim3(64), 0, ctx->stream, a);
    return launch_check(ctx, what);
  }
  const size_t ch = n < PAIR_CHUNK ? n : PAIR_CHUNK;
  if (mode & PAIR_FINAL_EXP) ST_TRY(reserve(ctx, ctx->pair_ws, (size_t)FE_NSLOTS * 12 * sizeof(Fq) * ch));
  for (size_t lo = 0; lo < n; lo += ch) {
    const size_t m = n - lo < ch ? n - lo : ch;
    PairArgs a;
    a.ps = d_g1 ? (const G1Aff*)d_g1 + lo * (size_t)p_stride : nullptr;
    a.p_stride = p_stride;
    a.qs = d_g2 ? (const G2Aff*)d_g2 + lo * (size_t)g2_stride : nullptr;
    a.q_stride = g2_stride;
    a.n = (u32)m;
    a.fixed_lines = d_fixed_lines ? (const Line*)d_fixed_lines + lo * (size_t)lines_stride : nullptr;
    a.lines_stride = lines_stride;
    a.f_in = d_f_in ? (const Fq*)d_f_in + 12 * lo : nullptr;
    a.ws = (Fq*)ctx->pair_ws.p;
    a.ws_n = ch;
    a.out = (char*)d_out + lo * out_item_bytes;
    a.mode = mode;
    hipLaunchKernelGGL(k_pairing, dim3(cdiv(2 * m, 64)), dim3(64), 0, ctx->stream, a);
  }
  return launch_check(ctx, what);
}
keaki_status pairing_run(keaki_hip_ctx* ctx, const void* d_g1, const void* d_g2, int g2_stride, size_t n, void* d_gt, const void* d_fixed_lines,
                         uint32_t lines_stride) {
  return pairing_launch(ctx, PAIR_MILLER | PAIR_FINAL_EXP | PAIR_OUT_BYTES, d_g1, d_g2, g2_stride, nullptr, n, d_gt, 384, d_fixed_lines, lines_stride,
                        "pairing_batch");
}
size_t pairing_launch_items() { return PAIR_CHUNK; }
uint32_t g2_prepared_lines() { return (uint32_t)MILLER_MAX_LINES * 2; }
size_t g2_prepared_bytes() { return (size_t)MILLER_MAX_LINES * 2 * sizeof(Line); }
keaki_status g2_prepare_run(keaki_hip_ctx* ctx, const void* d_q, void* d_lines, uint32_t n_points) {
  hipLaunchKernelGGL(k_g2_prepare, dim3(n_points), dim3(64), 0, ctx->stream, (const G2Aff*)d_q, (Line*)d_lines, g2_prepared_lines());
  return launch_check(ctx, "g2_prepare");
}
// test hook: the table in the 2^256 Montgomery form of the ABI (the kernels keep it in the 2^261 form)
keaki_status lines_to256_run(keaki_hip_ctx* ctx, const void* d_lines261, void* d_lines256) {
  const u32 count = (u32)(g2_prepared_bytes() / sizeof(Fq));
  hipLaunchKernelGGL(k_lines_to256, dim3(cdiv(count, 64)), dim3(64), 0, ctx->stream, (const Fq*)d_lines261, (Fq*)d_lines256, count);
  return launch_check(ctx, "lines_to256");
}
keaki_status pairing_raw_fixed_run(keaki_hip_ctx* ctx, const void* d_g1, uint32_t p_stride, size_t n, const void* d_lines, uint32_t lines_stride, void* d_out) {
  return pairing_launch(ctx, PAIR_MILLER | PAIR_FINAL_EXP | PAIR_OUT_RAW261, d_g1, nullptr, 0, nullptr, n, d_out, 12 * sizeof(Fq), d_lines, lines_stride,
                        "pairing_raw_fixed", p_stride);
}
size_t gt_table_bytes(uint32_t wb) { GtShape g = gt_shape(wb); return (size_t)g.windows * g.entries * 12 * sizeof(Fq); }
uint32_t gt_table_powers(uint32_t wb) { GtShape g = gt_shape(wb); return g.wb * g.windows; }
// d_table[j][d] = base^(d 2^(wb j)), d = 1 .. 2^(wb-1), base = e(P, Q): d_pows = e(2^s P, Q), s < gt_table_powers(wb), 12 Fq each
keaki_status gt_table_run(keaki_hip_ctx* ctx, const void* d_pows, void* d_table, uint32_t wb) {
  const GtShape g = gt_shape(wb);
  hipLaunchKernelGGL(k_gt_table_scatter, dim3(cdiv(g.wb * g.windows * 12, 256)), dim3(256), 0, ctx->stream, (const Fq*)d_pows, (Fq*)d_table, g);
  // the first levels hold a few entries each and wait for the latency of ONE product: twelve lanes per product there, a lane pair per product
  // once a level fills the device
  for (u32 L = 1; L + 2 <= g.wb; L++) {
    const u32 entries = g.windows * ((1u << L) - 1u);
    if (entries <= PAIR_WIDE_AUTO && ctx->tune.pair_wide_max != 0)
      hipLaunchKernelGGL(pw::k_gt_table_fill_wide, dim3(cdiv(entries, 4)), dim3(64), 0, ctx->stream, (Fq*)d_table, L, g);
    else
      hipLaunchKernelGGL(k_gt_table_fill, dim3(cdiv(2 * entries, 64)), dim3(64), 0, ctx->stream, (Fq*)d_table, L, g);
  }
  return launch_check(ctx, "gt_table");
}
keaki_status gt_encap_exp_run(keaki_hip_ctx* ctx, const void* d_tab_a, uint32_t wb_a, const void* d_tab_b, uint32_t wb_b, const void* d_betas,
                              const void* d_rs, size_t n, void* d_gt, const void* d_acc_in, void* d_acc_out) {
  const size_t wide_max = ctx->tune.pair_wide_max < 0 ? GT_EXP_WIDE_AUTO : (size_t)ctx->tune.pair_wide_max;
  const GtShape ga = d_tab_a ? gt_shape(wb_a) : GtShape{0, 0, 0}, gb = d_tab_b ? gt_shape(wb_b) : GtShape{0, 0, 0};
  if (n <= wide_max && d_tab_a && d_tab_b && !d_acc_in && !d_acc_out)
    hipLaunchKernelGGL(pw::k_gt_encap_exp_wide, dim3(cdiv(n, 4)), dim3(64), 0, ctx->stream, (const Fq*)d_tab_a, ga, (const Fq*)d_tab_b, gb,
                       (const Fr*)d_betas, (const Fr*)d_rs, (u32)n, (u32*)d_gt);
  else
    hipLaunchKernelGGL(k_gt_encap_exp, dim3(cdiv(2 * n, 64)), dim3(64), 0, ctx->stream, (const Fq*)d_tab_a, ga, (const Fq*)d_tab_b, gb,
                       (const Fr*)d_betas, (const Fr*)d_rs, (u32)n, (u32*)d_gt, (const Fq*)d_acc_in, (Fq*)d_acc_out);
  return launch_check(ctx, "gt_encap_exp");
}
keaki_status miller_only_run(keaki_hip_ctx* ctx, const void* d_g1, const void* d_g2, size_t n, void* d_out) {
  return pairing_launch(ctx, PAIR_MILLER | PAIR_OUT_RAW256, d_g1, d_g2, 1, nullptr, n, d_out, 12 * sizeof(Fq), nullptr, 0, "miller_only");
}
keaki_status final_exp_only_run(keaki_hip_ctx* ctx, const void* d_in, size_t n, void* d_gt) {
  return pairing_launch(ctx, PAIR_FINAL_EXP | PAIR_OUT_BYTES, nullptr, nullptr, 0, d_in, n, d_gt, 384, nullptr, 0, "final_exp_only");
}
keaki_status blake3_gt_run(keaki_hip_ctx* ctx, const void* d_gt, size_t n, void* d_key, size_t msg_len, bool xor_into) {
  hipLaunchKernelGGL(k_blake3_gt_xof, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const u32*)d_gt, (u32)n, (unsigned char*)d_key, (u32)msg_len,
                     xor_into ? 1u : 0u);
  return launch_check(ctx, "blake3_gt_xof");
}
}  // namespace keaki_internal
