// (core part: no dependency on the Fq type, so that bn254_field.hip.h itself can use it for the Fq2 product)
// BN254 base field in 9 x 29-bit limbs, Montgomery radix 2^261, lazily reduced -- the arithmetic of the bucket-accumulation
// kernel (k_msm_accumulate_g1_u29 in msm.hip.h; replaces the inner loop of ark-ec's VariableBaseMSM::msm_bigint_wnaf bucket
// phase behind src/kzg.rs:98).
//
// Why a second representation: on gfx950 v_mad_u64_u32 issues at the rate of a 32-bit add (profiles/r01_ubench_int_gfx950.txt),
// so a modular product costs its instruction count. With saturated 8 x 32-bit limbs every 32x32 product needs a carry
// instruction behind it (2 x 128 issues + slides + final subtraction ~ 300). With 29-bit limbs a column of 9 + 9 products of
// < 2^59 fits one 64-bit accumulator with no carry at all: 81 + 81 v_mad_u64_u32 + 9 (m_k) + ~50 slides/masks ~ 225 issues,
// and a squaring needs 45 + 81. Sums and differences are limb-wise 32-bit adds with one parallel carry pass.
//
// Lazy values: 2^261 / p = 169, so an element is any 9-limb integer x == value (mod p) with x < ~20p and limbs < 2^29 + 8
// (limb 8 holds the excess). A Montgomery product of a < alpha p and b < beta p is < (1 + alpha beta / 169) p, so products pull
// everything back below 2p and no conditional subtraction is ever needed; a difference adds a multiple of p whose limbs are
// biased by 2^30 (or 2^31) so that no limb goes negative.
//
// Bridge to the saturated R = 2^256 form used everywhere else (tables, buckets): x R 2^5 == x 2^261, so the 2^261-form of a
// saturated residue is the same integer shifted left by 5 bits -- taken for free while the limbs are cut (u29_from_sat_shift5,
// value < 32p); back: one product by the plain constant 2^256 mod p, one conditional subtraction, repack (u29_to_sat).
// tests: keaki_hip_selftest_field runs k_selftest_u29 (products, squares, differences, zero filter, round trips against the
// saturated asm field) and the MSM parity tests cover the kernel.
#pragma once
// included from bn254_field.hip.h after the basic types and bn254_constants.hip.h

namespace bn254 {

struct U29 {
  u32 l[9];
};
typedef Fq29Params Q29;

KDEV U29 u29_const(const u32 (&c)[9]) {
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c[i];
  return r;
}
KDEV U29 u29_one() { return u29_const(Q29::ONE); }

// limbs of (x << 5): the 2^261-Montgomery form (value < 32 p) of a saturated 2^256-Montgomery residue
KDEV U29 u29_from_sat_shift5(const u32* x) {
  U29 r;
  r.l[0] = (x[0] << 5) & Q29::MASK;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    const int s = 29 * i - 5, w = s >> 5, o = s & 31;
    u32 lo = x[w], hi = (w + 1 < 8) ? x[w + 1] : 0u;
    u32 v = o ? __builtin_amdgcn_alignbit(hi, lo, o) : lo;
    r.l[i] = (i < 8) ? (v & Q29::MASK) : v;
  }
  return r;
}

// limbs of x itself (no shift): the integer of a saturated residue cut into 29-bit limbs (limb 8 = bits 232..255)
KDEV U29 u29_from_sat_plain(const u32* x) {
  U29 r;
  r.l[0] = x[0] & Q29::MASK;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    const int s = 29 * i, w = s >> 5, o = s & 31;
    u32 lo = x[w], hi = (w + 1 < 8) ? x[w + 1] : 0u;
    u32 v = o ? __builtin_amdgcn_alignbit(hi, lo, o) : lo;
    r.l[i] = (i < 8) ? (v & Q29::MASK) : v;
  }
  return r;
}

// slide the 64-bit column accumulator down one limb
KDEV void u29_slide(u32& lo, u32& hi) {
  lo = __builtin_amdgcn_alignbit(hi, lo, 29);
  hi >>= 29;
}

// Portable statements of the product and the square (the shipped ones are the asm streams of fq29_asm.hip.h; the self-test
// compares the two). Montgomery product, radix 2^261. Limbs of a, b at most 2^30 + 16 on one side and 2^29 + 8 on the other (columns stay < 2^64).
KDEV U29 u29_mul_ref(const U29& a, const U29& b) {
  u64 acc = 0;
  u32 m[9];
  U29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * Q29::MOD[k - i];
    m[k] = ((u32)acc * Q29::INV) & Q29::MASK;
    acc += (u64)m[k] * Q29::MOD[0];
    u32 lo = (u32)acc, hi = (u32)(acc >> 32);
    u29_slide(lo, hi);
    acc = ((u64)hi << 32) | lo;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * Q29::MOD[k - i];
    u32 lo = (u32)acc, hi = (u32)(acc >> 32);
    r.l[k - 9] = lo & Q29::MASK;
    u29_slide(lo, hi);
    acc = ((u64)hi << 32) | lo;
  }
  r.l[8] = (u32)acc;
  return r;
}
// Montgomery square: cross products once, against the doubled operand (limbs of a at most 2^29 + 8)
KDEV U29 u29_sqr_ref(const U29& a) {
  u64 acc = 0;
  u32 m[9], d[9];
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (u64)d[i] * a.l[k - i];
    if ((k & 1) == 0) acc += (u64)a.l[k >> 1] * a.l[k >> 1];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * Q29::MOD[k - i];
    m[k] = ((u32)acc * Q29::INV) & Q29::MASK;
    acc += (u64)m[k] * Q29::MOD[0];
    u32 lo = (u32)acc, hi = (u32)(acc >> 32);
    u29_slide(lo, hi);
    acc = ((u64)hi << 32) | lo;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; 2 * i < k; i++) acc += (u64)d[i] * a.l[k - i];
    if ((k & 1) == 0) acc += (u64)a.l[k >> 1] * a.l[k >> 1];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * Q29::MOD[k - i];
    u32 lo = (u32)acc, hi = (u32)(acc >> 32);
    r.l[k - 9] = lo & Q29::MASK;
    u29_slide(lo, hi);
    acc = ((u64)hi << 32) | lo;
  }
  r.l[8] = (u32)acc;
  return r;
}

}  // namespace bn254
#include "fq29_asm.hip.h"
namespace bn254 {
KDEV U29 u29_mul(const U29& a, const U29& b) {
  U29 r;
  u29_mul_asm(r.l, a.l, b.l);
  return r;
}
KDEV U29 u29_sqr(const U29& a) {
  U29 r;
  u29_sqr_asm(r.l, a.l);
  return r;
}
// (a b + c d) / 2^261 with one reduction (limb bounds: see fq29_asm.hip.h)
KDEV U29 u29_mul2(const U29& a, const U29& b, const U29& c, const U29& d) {
  U29 r;
  u29_mul2_asm(r.l, a.l, b.l, c.l, d.l);
  return r;
}

// one parallel carry pass: limbs 0..7 back below 2^29 + 8 (inputs: any u32 limbs), value unchanged
KDEV U29 u29_carry(const U29& x) {
  U29 r;
  r.l[0] = x.l[0] & Q29::MASK;
#pragma unroll
  for (int i = 1; i < 8; i++) r.l[i] = (x.l[i] & Q29::MASK) + (x.l[i - 1] >> 29);
  r.l[8] = x.l[8] + (x.l[7] >> 29);
  return r;
}
// a - b + K, K a biased multiple of p that is at least the value bound of b
KDEV U29 u29_sub(const U29& a, const U29& b, const u32 (&K)[9]) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = a.l[i] - b.l[i] + K[i];
  return u29_carry(t);
}
// the same without the carry pass: limbs below 2^31. Allowed as ONE operand of a product whose other operand is carried, as
// the subtrahend of a difference biased by 2^31, and into u29_to_sat; never squared.
KDEV U29 u29_sub_raw(const U29& a, const U29& b, const u32 (&K)[9]) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = a.l[i] - b.l[i] + K[i];
  return t;
}
// a - b - 2c + 8p  (bias 2^31)
KDEV U29 u29_sub3(const U29& a, const U29& b, const U29& c) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = a.l[i] - b.l[i] - 2u * c.l[i] + Q29::K8W[i];
  return u29_carry(t);
}
// cheap necessary condition for x == 0 (mod p) when x < 18p and limb 0 is exact (after u29_carry): x = k p => l[0] * p^-1 = k
KDEV bool u29_maybe_zero(const U29& x) { return ((x.l[0] * Q29::PINV) & Q29::MASK) <= 17u; }
// the same for x < 35p (the H of a mixed addition: U2 - X1 + 32p with X1 < 19p)
KDEV bool u29_maybe_zero34(const U29& x) { return ((x.l[0] * Q29::PINV) & Q29::MASK) <= 34u; }

// t < 2p with exact limbs (a product's output) -> the canonical integer below p, packed into 8 x 32 bits
KDEV void fq_cond_sub_p_asm(u32* __restrict__ r, const u32* __restrict__ t);   // bn254_field_asm.hip.h (included after this file)
KDEV void u29_pack_canonical(u32* out, const U29& t) {
  // pack the (exact) limbs into 8 x 32 bits first -- 2p < 2^255 fits -- then ONE conditional subtraction on the hardware borrow chain
  // (16 instructions); a signed ripple over nine 29-bit limbs costs 45
  u32 v[10], w[8];
#pragma unroll
  for (int i = 0; i < 9; i++) v[i] = t.l[i];
  v[9] = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int s = 32 * j, q = s / 29, o = s % 29;
    u64 x = ((u64)v[q + 1] << 29) | v[q];
    if (q + 2 < 10 && 58 - o < 32) x |= (u64)v[q + 2] << 58;
    w[j] = (u32)(x >> o);
  }
  fq_cond_sub_p_asm(out, w);
}
// 2^261-form lazy value -> canonical saturated 2^256-form residue (8 x 32)
KDEV void u29_to_sat(u32* out, const U29& a) {
  u29_pack_canonical(out, u29_mul(a, u29_const(Q29::R256)));   // == value * 2^256 (mod p), < 2p, limbs exact
}
}  // namespace bn254
