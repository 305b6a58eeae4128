// Batched pairing + GT serialisation + BLAKE3 KDF kernels (reference src/kem.rs:30-46,58-69).
#define KEAKI_FQ2_OUTLINE 1
#define KEAKI_PAIRING_INLINE_TOWER 1
#include "internal.h"
#include "pairing.cuh"
namespace keaki_internal {
using namespace bn254;
keaki_status pairing_run(keaki_hip_ctx* ctx, const void* d_g1, const void* d_g2, int g2_stride, size_t n, void* d_gt) {
  hipLaunchKernelGGL(k_pairing_batch, dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, (const G1Aff*)d_g1, (const G2Aff*)d_g2, g2_stride, (u32)n,
                     (u32*)d_gt);
  return launch_check(ctx, "pairing_batch");
}
keaki_status blake3_gt_run(keaki_hip_ctx* ctx, const void* d_gt, size_t n, void* d_key, size_t msg_len) {
  hipLaunchKernelGGL(k_blake3_gt_xof, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const u32*)d_gt, (u32)n, (unsigned char*)d_key, (u32)msg_len);
  return launch_check(ctx, "blake3_gt_xof");
}
}  // namespace keaki_internal
