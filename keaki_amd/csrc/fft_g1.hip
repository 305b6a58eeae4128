// FK23 batch openings on the GPU: all d KZG opening proofs at the d-th roots of unity in O(d log d) group operations.
//
// Replaces kzg::open_fk (reference src/kzg.rs:157-203; caller src/vec.rs:40), whose work is three ark-poly group FFTs
// (`domain_2d.fft(&s)`, `domain_2d.ifft(&hat_h)`, `domain_d.fft(&h)`) and 2d scalar multiplications (`hat_s[i].mul(hat_a[i])`,
// :189-191). Here:
//   S[i] = [tau^(d-1-i)]_1 (i < d), identity (i >= d)            k_fk_load          (SRS reversed, src/kzg.rs:166-174)
//   hat_s <- DFT_2d(S)                                           k_bitrev + k_g1_fft_stage x log2(2d); cached per (SRS, d)
//   S[i] <- hat_a[i] * hat_s[i]                                  k_g1_mul_jac_oop   (hat_a = DFT_2d(0..0, p) / 2d, from the host)
//   S <- DFT_2d^-1(S) (scaling folded into hat_a); h = S[0..d]   same kernels with inverse twiddles
//   proofs <- DFT_d(h), normalised to affine                     k_g1_fft_stage x log2(d), k_g1_jac_to_aff
// A butterfly is one scalar multiplication of a Jacobian point by a twiddle factor plus an add and a subtract; one lane
// per butterfly, N/2 lanes per stage. Twiddle tables (omega^k, k < N/2) come from the host (scalar-field work stays there).
#include "ec_batch.cuh"
#include "internal.h"

namespace bn254 {

// k * P for a Jacobian P (full Jacobian additions), MSB first. k: Montgomery Fr.
KDEV G1Jac jac_scalar_mul(const G1Jac& p, const Fr& k_mont) {
  u32 v[8];
  fp_from_mont<FrParams>(v, k_mont);
  G1Jac acc = jac_inf<Fq>();
  if (jac_is_inf(p)) return acc;
#pragma unroll
  for (int s = 0; s < 2; s++) {   // the two top bits of a 254-bit scalar are zero
#pragma unroll
    for (int j = 7; j > 0; j--) v[j] = (v[j] << 1) | (v[j - 1] >> 31);
    v[0] <<= 1;
  }
#pragma unroll 1
  for (int i = 0; i < 254; i++) {
    acc = jac_dbl(acc);
    if (v[7] >> 31) acc = jac_add(acc, p);
#pragma unroll
    for (int j = 7; j > 0; j--) v[j] = (v[j] << 1) | (v[j - 1] >> 31);
    v[0] <<= 1;
  }
  return acc;
}
KDEV bool fr_is_one(const Fr& a) {
  u32 o = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) o |= a.l[j] ^ FrParams::ONE[j];
  return o == 0;
}

static __global__ void __launch_bounds__(256) k_fk_load(const G1Aff* __restrict__ srs, u32 d, G1Jac* __restrict__ s) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * d) return;
  s[i] = i < d ? jac_from_aff(srs[d - 1 - i]) : jac_inf<Fq>();
}
static __global__ void __launch_bounds__(256) k_bitrev_jac(G1Jac* __restrict__ a, u32 log2n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log2n)) return;
  u32 r = __brev(i) >> (32 - log2n);
  if (i < r) { G1Jac t = a[i]; a[i] = a[r]; a[r] = t; }
}
// stage with butterfly span `len`: for block b and j < len/2: (u, v) = (a[i], w^j a[i + len/2]), w = omega^(n/len)
static __global__ void __launch_bounds__(64) k_g1_fft_stage(G1Jac* __restrict__ a, const Fr* __restrict__ tw, u32 n, u32 len) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n / 2) return;
  u32 half = len >> 1;
  u32 j = b % half, i0 = (b / half) * len + j, i1 = i0 + half;
  Fr w = tw[(size_t)j * (n / len)];
  G1Jac u = a[i0], v = a[i1];
  if (!fr_is_one(w)) v = jac_scalar_mul(v, w);
  a[i0] = jac_add(u, v);
  v.y = -v.y;
  a[i1] = jac_add(u, v);
}
static __global__ void __launch_bounds__(64) k_g1_mul_jac(G1Jac* __restrict__ a, const Fr* __restrict__ s, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  a[i] = jac_scalar_mul(a[i], s[i]);
}
static __global__ void __launch_bounds__(64) k_g1_jac_to_aff(const G1Jac* __restrict__ a, u32 n, G1Aff* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = jac_to_aff(a[i]);
}

}  // namespace bn254

namespace keaki_internal {
using namespace bn254;

static keaki_status g1_fft(keaki_hip_ctx* ctx, G1Jac* a, u32 log2n, const Fr* tw) {
  const u32 n = 1u << log2n;
  hipLaunchKernelGGL(k_bitrev_jac, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, a, log2n);
  for (u32 len = 2; len <= n; len <<= 1)
    hipLaunchKernelGGL(k_g1_fft_stage, dim3(cdiv(n / 2, 64)), dim3(64), 0, ctx->stream, a, tw, n, len);
  return launch_check(ctx, "g1_fft");
}

// out[i] = s[i] * in[i] (out-of-place pointwise product; in = cached hat_s)
static __global__ void __launch_bounds__(64) k_g1_mul_jac_oop(const G1Jac* __restrict__ in, const Fr* __restrict__ s, u32 n, G1Jac* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = jac_scalar_mul(in[i], s[i]);
}

// hat_s = DFT_2d(reversed SRS padded with identities): depends on the SRS only, so it is computed once per (SRS, d) and cached.
keaki_status fk_hat_s_run(keaki_hip_ctx* ctx, const void* d_srs, u32 log2d, const void* d_tw2d, void* d_hat_s) {
  const u32 d = 1u << log2d;
  G1Jac* s = (G1Jac*)d_hat_s;
  hipLaunchKernelGGL(k_fk_load, dim3(cdiv(2 * d, 256)), dim3(256), 0, ctx->stream, (const G1Aff*)d_srs, d, s);
  return g1_fft(ctx, s, log2d + 1, (const Fr*)d_tw2d);
}
// d = 2^log2d openings from the cached hat_s. d_work: 2d Jacobian points. d_hat_a: 2d Fr (already divided by 2d).
// d_tw2d_inv: d Fr (omega_2d^-k); d_twd: d/2 Fr (omega_d^k). Output: d affine proofs.
keaki_status open_fk_run(keaki_hip_ctx* ctx, const void* d_hat_s, u32 log2d, const void* d_hat_a, const void* d_tw2d_inv, const void* d_twd,
                         void* d_work, void* d_proofs_aff) {
  const u32 d = 1u << log2d;
  G1Jac* s = (G1Jac*)d_work;
  hipLaunchKernelGGL(k_g1_mul_jac_oop, dim3(cdiv(2 * d, 64)), dim3(64), 0, ctx->stream, (const G1Jac*)d_hat_s, (const Fr*)d_hat_a, 2 * d, s);
  ST_TRY(g1_fft(ctx, s, log2d + 1, (const Fr*)d_tw2d_inv));
  if (log2d > 0) ST_TRY(g1_fft(ctx, s, log2d, (const Fr*)d_twd));
  hipLaunchKernelGGL(k_g1_jac_to_aff, dim3(cdiv(d, 64)), dim3(64), 0, ctx->stream, (const G1Jac*)s, d, (G1Aff*)d_proofs_aff);
  return launch_check(ctx, "open_fk");
}

}  // namespace keaki_internal
