// On-device self-test of the hand-scheduled Fq streams against the portable template code.
#include "pairing.hip.h"
#include "fq29.hip.h"
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

// lane-pair primitives of the pairing tower (pair261.hip.h: 2^261 form, multi-product streams) against the single-lane Fq2 code of
// bn254_field.hip.h (2^256 form): operands are converted on the way in (to261), results on the way out (to256)
KDEV Fq2 ref_add3(const Fq2& a, const Fq2& b, const Fq2& c) { return a + b + c; }
struct RefFq6 { Fq2 c0, c1, c2; };
KDEV RefFq6 ref_fq6_mul(const RefFq6& a, const RefFq6& b) {
  return {ref_add3(a.c0 * b.c0, fq2_mul_xi(a.c1 * b.c2), fq2_mul_xi(a.c2 * b.c1)), ref_add3(a.c0 * b.c1, a.c1 * b.c0, fq2_mul_xi(a.c2 * b.c2)),
          ref_add3(a.c0 * b.c2, a.c1 * b.c1, a.c2 * b.c0)};
}
KDEV RefFq6 ref_fq6_add(const RefFq6& a, const RefFq6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
KDEV RefFq6 ref_fq6_mul_v(const RefFq6& a) { return {fq2_mul_xi(a.c2), a.c0, a.c1}; }
__global__ void __launch_bounds__(64) k_selftest_fq2d(u32 seed, u32 iters, unsigned long long* mismatches) {
  using namespace p261;
  u32 pair = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  u32 s = (seed ^ (pair * 0x9E3779B9u)) | 1u;
  const u32 par = lane_odd();
  unsigned long long bad = 0;
  auto comp = [&](const Fq2& x) { return par ? x.c1 : x.c0; };
  auto ld = [&](const Fq2& x) { return Fq2d{to261(comp(x))}; };                       // this lane's component, 2^261 form
  auto eq = [&](const Fq2d& got, const Fq2& exp) { return fq_eq(to256(got.v), comp(exp)); };
  for (u32 it = 0; it < iters; it++) {
    const u32 sh = mix(s);
    Fq2 a = {pick(s, (sh & 7) < 6 ? (sh & 7) : 0), pick(s, ((sh >> 3) & 7) < 6 ? ((sh >> 3) & 7) : 0)}, b = {pick(s, 0), pick(s, ((sh >> 6) & 7) < 6 ? ((sh >> 6) & 7) : 0)};
    if ((it & 7) == 3) a.c1 = fq_zero();                                               // identical in both lanes of the pair
    if ((it & 7) == 5) b.c0 = fq_zero();
    Fq k = pick(s, 0);
    Fq2d ad = ld(a), bd = ld(b);
    bad += !fq_eq(to256(to261(k)), k);
    bad += !eq(ad * bd, a * b);
    bad += !eq(fq2_sqr(ad), bn254::fq2_sqr(a));
    bad += !eq(fq2_mul_xi(ad), bn254::fq2_mul_xi(a));
    {  // (9 + u) a and a / 2 on WORD values at the ends of the range (0, 1, p - 1, p - 2): the single-reduction forms against plain arithmetic
      const u32 c = (sh >> 9) & 15u;
      Fq w0 = fq_zero(), w1 = fq_zero();
      if (c & 1u) w0.l[0] = 1u;
      if (c & 2u) { for (int j = 0; j < 8; j++) w0.l[j] = FqParams::MOD[j]; w0.l[0] -= (c & 1u) ? 2u : 1u; }
      if (c & 4u) w1.l[0] = 1u;
      if (c & 8u) { for (int j = 0; j < 8; j++) w1.l[j] = FqParams::MOD[j]; w1.l[0] -= (c & 4u) ? 2u : 1u; }
      const Fq2d e = {par ? w1 : w0};
      const Fq2d got = fq2_mul_xi(e);
      const Fq2d other = {fq_partner(e.v)};
      const Fq nine = fq_dbl(fq_dbl(fq_dbl(e.v))) + e.v;
      const Fq want = par ? nine + other.v : nine - other.v;
      bad += !fq_eq(got.v, want);
      bad += !fq_eq(fq_dbl(fq_half(e.v)), e.v);
      bad += !fq_eq(fq_inv_safegcd(e.v), fq_inv_fermat(e.v));                            // division steps against the Fermat ladder: ends of the range ...
      bad += !fq_eq(fq_inv_safegcd(k), fq_inv_fermat(k));
      bad += !fq_eq(fq_inv261(to261(k)), fq_inv261_fermat(to261(k)));              // the 2^261-form wrapper of the pairing tower                                // ... and a random residue (0 -> 0 in both)
      bad += !fq_eq(fq_half(ad.v) + fq_half(ad.v), ad.v);
    }
    bad += !eq(fq2_conj(ad), bn254::fq2_conj(a));
    bad += !eq(fq2_inv(ad), bn254::fq2_inv(a));
    bad += !eq(fq2_mul_fq(ad, to261(k)), bn254::fq2_mul_fq(a, k));
    bad += !eq(ad + bd, a + b);
    bad += !eq(ad - bd, a - b);
    bad += !eq(fq2_dbl(ad), bn254::fq2_dbl(a));
    bad += !eq(fq2_neg(ad), -a);
    bad += !eq(fq2d_one(), fq2_one());
    {  // canonical bytes of a 2^261 residue == from-Montgomery of the 2^256 residue
      u32 w1[8], w2[8];
      canon_words(w1, ad.v);
      fp_from_mont<FqParams>(w2, comp(a));
      for (int j = 0; j < 8; j++) bad += (w1[j] != w2[j]);
    }
    // the multi-product tower operations
    Fq2 c = {pick(s, 0), pick(s, 0)}, d = {pick(s, 0), pick(s, 0)}, e = {pick(s, 0), pick(s, 0)}, f = {pick(s, 0), pick(s, 0)};
    if ((it & 7) == 6) { c = a; d = a; }
    const RefFq6 ra = {a, c, d}, rb = {b, e, f};
    {
      Fq6 xa = {ld(a), ld(c), ld(d)}, xb = {ld(b), ld(e), ld(f)}, xr;
      fq6_mul(&xr, &xa, &xb);
      const RefFq6 rr = ref_fq6_mul(ra, rb);
      bad += !eq(xr.c0, rr.c0) + !eq(xr.c1, rr.c1) + !eq(xr.c2, rr.c2);
      fq6_mul(&xa, &xa, &xa);                                                          // aliased: squares in place
      const RefFq6 rs = ref_fq6_mul(ra, ra);
      bad += !eq(xa.c0, rs.c0) + !eq(xa.c1, rs.c1) + !eq(xa.c2, rs.c2);
    }
    {  // Fq4 squaring (x + y s)^2, s^2 = xi
      Fq2d t0, t1;
      fq4_sqr(&t0, &t1, ad, bd);
      bad += !eq(t0, bn254::fq2_sqr(a) + bn254::fq2_mul_xi(bn254::fq2_sqr(b))) + !eq(t1, bn254::fq2_dbl(a * b));
    }
    {  // sparse line product f * (l0 + (l1 + l2 v) w) with the line's coefficients as limbs (l0, l1 outputs of products by an Fq: < 2p)
      Fq12 xf = {{ld(a), ld(c), ld(d)}, {ld(b), ld(e), ld(f)}};
      Fq2 l0 = {pick(s, 0), pick(s, 0)}, l1 = {pick(s, 0), pick(s, 0)}, l2 = {pick(s, 0), pick(s, 2)};
      const Fq py = pick(s, 0), px = pick(s, 2);
      const U29 c0 = u29_mul(cut(ld(l0).v), cut(to261(py))), d0 = u29_mul(cut(ld(l1).v), cut(to261(px)));
      fq12_mul_by_034_limbs(&xf, c0, d0, cut(ld(l2).v), true);
      const Fq2 C0 = bn254::fq2_mul_fq(l0, py), D0 = bn254::fq2_mul_fq(l1, px);
      const RefFq6 f0 = ra, f1 = rb, D = {D0, l2, fq2_zero()}, Cc = {C0, fq2_zero(), fq2_zero()};
      const RefFq6 r0 = ref_fq6_add(ref_fq6_mul(f0, Cc), ref_fq6_mul_v(ref_fq6_mul(f1, D))), r1 = ref_fq6_add(ref_fq6_mul(f1, Cc), ref_fq6_mul(f0, D));
      bad += !eq(xf.c0.c0, r0.c0) + !eq(xf.c0.c1, r0.c1) + !eq(xf.c0.c2, r0.c2) + !eq(xf.c1.c0, r1.c0) + !eq(xf.c1.c1, r1.c1) + !eq(xf.c1.c2, r1.c2);
    }
    {  // Fq12 product, squaring and inverse through the one-instance loops: (x y) == schoolbook over the reference Fq6; x * x^-1 == 1
      Fq12 x = {{ld(a), ld(c), ld(d)}, {ld(b), ld(e), ld(f)}}, y = {{ld(e), ld(a), ld(f)}, {ld(d), ld(b), ld(c)}}, z, q;
      const RefFq6 x0 = ra, x1 = rb, y0 = {e, a, f}, y1 = {d, b, c};
      fq12_mul(&z, &x, &y);
      const RefFq6 z0 = ref_fq6_add(ref_fq6_mul(x0, y0), ref_fq6_mul_v(ref_fq6_mul(x1, y1))), z1 = ref_fq6_add(ref_fq6_mul(x0, y1), ref_fq6_mul(x1, y0));
      bad += !eq(z.c0.c0, z0.c0) + !eq(z.c0.c1, z0.c1) + !eq(z.c0.c2, z0.c2) + !eq(z.c1.c0, z1.c0) + !eq(z.c1.c1, z1.c1) + !eq(z.c1.c2, z1.c2);
      fq12_sqr(&q, &x);
      fq12_mul(&z, &x, &x);
      const Fq2d* qa = reinterpret_cast<const Fq2d*>(&q);
      const Fq2d* za = reinterpret_cast<const Fq2d*>(&z);
      for (int j = 0; j < 6; j++) bad += !fq_eq(qa[j].v, za[j].v);
      if ((it & 3) == 0) {
        fq12_inv(&q, &x);
        fq12_mul(&z, &q, &x);
        Fq12 one;
        fq12_set_one(&one);
        const Fq2d* oa = reinterpret_cast<const Fq2d*>(&one);
        for (int j = 0; j < 6; j++) bad += !fq_eq(za[j].v, oa[j].v);
      }
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
