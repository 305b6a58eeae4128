// On-device self-test of the hand-scheduled Fq streams against the portable template code.
#include "pairing.cuh"
#include "fq29.cuh"
#include "internal.h"
namespace bn254 {
using FqRef = Fp<FqParamsRef>;
KDEV u32 mix(u32& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
// a value < p with a choice of corner shapes
KDEV Fq pick(u32& s, u32 shape) {
  Fq x;
#pragma unroll
  for (int j = 0; j < 8; j++) x.l[j] = mix(s);
  if (shape == 1) { for (int j = 0; j < 8; j++) x.l[j] = 0; }
  else if (shape == 2) { for (int j = 0; j < 8; j++) x.l[j] = FqParams::MOD[j]; x.l[0] -= 1; }          // p - 1
  else if (shape == 3) { for (int j = 0; j < 7; j++) x.l[j] = 0xFFFFFFFFu; x.l[7] = 0x2FFFFFFFu; }       // dense ones below p
  else if (shape == 4) { for (int j = 1; j < 8; j++) x.l[j] = 0; x.l[0] = mix(s) & 3; }                 // tiny
  else if (shape == 5) { for (int j = 0; j < 8; j++) x.l[j] = (j & 1) ? 0xFFFFFFFFu : 0u; x.l[7] = 0x1FFFFFFFu; }
  x.l[7] &= 0x3FFFFFFFu;  // < 2^254
  u32 t[8];
#pragma unroll
  for (int j = 0; j < 8; j++) t[j] = x.l[j];
  fp_reduce_once<FqParamsRef>(t);   // < 2^254 < 2p -> < p
#pragma unroll
  for (int j = 0; j < 8; j++) x.l[j] = t[j];
  return x;
}
KDEV FqRef as_ref(const Fq& a) { FqRef r; for (int j = 0; j < 8; j++) r.l[j] = a.l[j]; return r; }
KDEV bool same(const Fq& a, const FqRef& b) { u32 o = 0; for (int j = 0; j < 8; j++) o |= a.l[j] ^ b.l[j]; return o == 0; }

__global__ void __launch_bounds__(256) k_selftest_field(u32 seed, u32 iters, unsigned long long* mismatches) {
  u32 s = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (threadIdx.x * 0x85EBCA6Bu);
  s |= 1;
  unsigned long long bad = 0;
  Fq chain = pick(s, 0);
  FqRef chain_ref = as_ref(chain);
  for (u32 it = 0; it < iters; it++) {
    u32 sh = mix(s);
    Fq a = pick(s, (sh & 15) < 6 ? (sh & 15) : 0), b = pick(s, ((sh >> 4) & 15) < 6 ? ((sh >> 4) & 15) : 0);
    FqRef ar = as_ref(a), br = as_ref(b);
    bad += !same(a * b, fp_mul<FqParamsRef>(ar, br));
    bad += !same(a + b, fp_add<FqParamsRef>(ar, br));
    bad += !same(a - b, fp_sub<FqParamsRef>(ar, br));
    bad += !same(-a, fp_neg<FqParamsRef>(ar));
    bad += !same(fq_sqr(a), fp_mul<FqParamsRef>(ar, ar));
    if ((it & 15) == 0) bad += !fq_eq(fq_inv_xgcd(a), fq_inv(a));     // binary-GCD inverse == Fermat ladder (incl. 0, p-1, tiny values)
    {  // Fq2 product (one dual 29-bit product per component) against the Karatsuba form on the saturated streams
      Fq2 x = {a, b}, y = {chain, pick(s, (sh >> 8) & 7)};
      Fq t0 = x.c0 * y.c0, t1 = x.c1 * y.c1, t2 = (x.c0 + x.c1) * (y.c0 + y.c1);
      Fq2 z = fq2_mul_inl(x, y);
      bad += !fq_eq(z.c0, t0 - t1) + !fq_eq(z.c1, t2 - t0 - t1);
    }
    {  // Montgomery -> canonical of a scalar (fr_from_mont_asm) against the portable loop
      Fr x; Fp<FrParamsRef> xr;
      for (int j = 0; j < 8; j++) x.l[j] = mix(s);
      x.l[7] &= 0x1FFFFFFFu;                                       // < 2^253 < r
      if ((sh & 0x300) == 0x100) { for (int j = 0; j < 8; j++) x.l[j] = FrParams::MOD[j]; x.l[0] -= 1; }
      if ((sh & 0x300) == 0x200) { for (int j = 0; j < 8; j++) x.l[j] = (j == 0); }
      for (int j = 0; j < 8; j++) xr.l[j] = x.l[j];
      u32 c1[8], c2[8];
      fp_from_mont<FrParams>(c1, x);
      fp_from_mont<FrParamsRef>(c2, xr);
      for (int j = 0; j < 8; j++) bad += (c1[j] != c2[j]);
    }
    // dependent chain (exercises back-to-back streams)
    chain = chain * a + b - chain * chain;
    chain_ref = fp_sub<FqParamsRef>(fp_add<FqParamsRef>(fp_mul<FqParamsRef>(chain_ref, ar), br), fp_mul<FqParamsRef>(chain_ref, chain_ref));
    bad += !same(chain, chain_ref);
  }
  if (bad) atomicAdd(mismatches, bad);
}

// lane-pair Fq2 primitives (pairing.cuh) against the single-lane Fq2 code
__global__ void __launch_bounds__(64) k_selftest_fq2d(u32 seed, u32 iters, unsigned long long* mismatches) {
  u32 pair = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  u32 s = (seed ^ (pair * 0x9E3779B9u)) | 1u;
  const u32 par = lane_odd();
  unsigned long long bad = 0;
  for (u32 it = 0; it < iters; it++) {
    Fq2 a = {pick(s, 0), pick(s, 0)}, b = {pick(s, 0), pick(s, 0)};   // identical in both lanes of the pair
    if ((it & 7) == 3) a.c1 = fq_zero();
    if ((it & 7) == 5) b.c0 = fq_zero();
    Fq k = pick(s, 0);
    Fq2d ad = fq2d_load(&a), bd = fq2d_load(&b);
    auto comp = [&](const Fq2& x) { return par ? x.c1 : x.c0; };
    bad += !fq_eq((ad * bd).v, comp(a * b));
    bad += !fq_eq(fq2_sqr(ad).v, comp(fq2_sqr(a)));
    bad += !fq_eq(fq2_mul_xi(ad).v, comp(fq2_mul_xi(a)));
    bad += !fq_eq(fq2_conj(ad).v, comp(fq2_conj(a)));
    bad += !fq_eq(fq2_inv(ad).v, comp(fq2_inv(a)));
    bad += !fq_eq(fq2_mul_fq(ad, k).v, comp(fq2_mul_fq(a, k)));
    bad += !fq_eq((ad + bd).v, comp(a + b));
    bad += !fq_eq((ad - bd).v, comp(a - b));
    bad += !fq_eq(fq2_dbl(ad).v, comp(fq2_dbl(a)));
    bad += !fq_eq((-ad).v, comp(-a));
    bad += !fq_eq(fq2d_one().v, comp(fq2_one()));
    {  // unary minus right after a subtraction of a product (the line_add shape)
      Fq2d x = ad - bd * ad;
      Fq2d n1 = -x, n2 = fq2d_zero() - x;
      bad += !fq_eq(n1.v, n2.v);
      Fq2 xs = a - b * a;
      bad += !fq_eq(n1.v, comp(-xs));
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace bn254
namespace keaki_internal {
keaki_status selftest_field_run(keaki_hip_ctx* ctx, uint32_t blocks, uint32_t iters, uint32_t seed, void* d_mismatches) {
  hipLaunchKernelGGL(bn254::k_selftest_field, dim3(blocks), dim3(256), 0, ctx->stream, seed, iters, (unsigned long long*)d_mismatches);
  hipLaunchKernelGGL(bn254::k_selftest_fq2d, dim3(blocks), dim3(64), 0, ctx->stream, seed, iters > 8 ? 8u : iters, (unsigned long long*)d_mismatches);
  keaki_status st = selftest_u29_run(ctx, blocks, iters, seed, d_mismatches);
  if (st != KEAKI_OK) return st;
  return launch_check(ctx, "selftest_field");
}
}  // namespace keaki_internal
