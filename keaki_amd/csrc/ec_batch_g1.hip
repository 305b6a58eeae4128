// G1 batched scalar multiplication and the G1 half of encapsulate (reference src/kem.rs:22,30).
#include "ec_batch.cuh"
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
}  // namespace keaki_internal
