// BN254 prime-field arithmetic for gfx950 (CDNA4), Montgomery form, 8 x 32-bit limbs.
//
// Replaces ark-ff 0.4.2 `Fp<MontBackend<_, 4>>` (4 x u64 Montgomery, R = 2^256) on keaki's hot
// path (reference call sites: src/kzg.rs:98, src/kem.rs:22-37,58). The residues are bit-identical
// to ark-ff's: a u64[4] limb array from the host is read here as u32[8] little-endian, so the FFI
// copies `Fp.0.0` without conversion.
//
// Integer-only (no MFMA: 256-bit modular products are not a dense contraction). The multiplier is
// v_mad_u64_u32-based; everything is fully unrolled so limbs live in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn254 {

typedef uint32_t u32;
typedef uint64_t u64;

#define KDEV __device__ __forceinline__

struct FqParams;
struct FrParams;

template <class P>
struct alignas(16) Fp {
  u32 l[8];
};
using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;
struct alignas(16) Fq2 {
  Fq c0, c1;
};

}  // namespace bn254

#include "bn254_constants.hip.h"
#include "fq29_core.hip.h"

namespace bn254 {

// ---------------------------------------------------------------------------------------------
// raw 256-bit helpers
// ---------------------------------------------------------------------------------------------
template <class P>
KDEV bool fp_is_zero(const Fp<P>& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.l[i];
  return o == 0;
}
template <class P>
KDEV bool fp_eq(const Fp<P>& a, const Fp<P>& b) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}
template <class P>
KDEV Fp<P> fp_zero() {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = 0;
  return r;
}
template <class P>
KDEV Fp<P> fp_one() {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = P::ONE[i];
  return r;
}

// r = a - MOD if a >= MOD else a   (a < 2*MOD)
template <class P>
KDEV void fp_reduce_once(u32* t) {
  u32 s[8];
  u32 borrow = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 d = (u64)t[j] - P::MOD[j] - borrow;
    s[j] = (u32)d;
    borrow = (u32)(d >> 63);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) t[j] = borrow ? t[j] : s[j];
}

template <class P>
KDEV Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
  u32 carry = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 s = (u64)a.l[j] + b.l[j] + carry;
    r.l[j] = (u32)s;
    carry = (u32)(s >> 32);
  }
  // p < 2^254 so a+b < 2^255: no carry out of 256 bits
  fp_reduce_once<P>(r.l);
  return r;
}
template <class P>
KDEV Fp<P> fp_dbl(const Fp<P>& a) {
  return fp_add<P>(a, a);
}
template <class P>
KDEV Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
  u32 borrow = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 d = (u64)a.l[j] - b.l[j] - borrow;
    r.l[j] = (u32)d;
    borrow = (u32)(d >> 63);
  }
  u32 mask = 0u - borrow;
  u32 carry = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 s = (u64)r.l[j] + (P::MOD[j] & mask) + carry;
    r.l[j] = (u32)s;
    carry = (u32)(s >> 32);
  }
  return r;
}
template <class P>
KDEV Fp<P> fp_neg(const Fp<P>& a) {
  Fp<P> r;
  u32 nz = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) nz |= a.l[j];
  u32 borrow = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 d = (u64)P::MOD[j] - a.l[j] - borrow;
    r.l[j] = nz ? (u32)d : 0u;
    borrow = (u32)(d >> 63);
  }
  return r;
}
// conditional negate: r = neg ? -a : a
template <class P>
KDEV Fp<P> fp_cneg(const Fp<P>& a, bool neg) {
  Fp<P> n = fp_neg<P>(a);
  Fp<P> r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = neg ? n.l[j] : a.l[j];
  return r;
}

// ---------------------------------------------------------------------------------------------
// Montgomery multiplication: CIOS over 32-bit limbs. Because p < 2^254 (two spare bits) the
// running value stays < 2p < 2^255 after every outer iteration, so no ninth limb is carried.
// ---------------------------------------------------------------------------------------------
template <class P>
KDEV Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
  u32 t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      u64 x = (u64)a.l[j] * b.l[i] + (i ? t[j] : 0u) + c;
      t[j] = (u32)x;
      c = x >> 32;
    }
    u32 t8 = (u32)c;
    u32 m = t[0] * P::INV;
    u64 x = (u64)m * P::MOD[0] + t[0];
    c = x >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      x = (u64)m * P::MOD[j] + t[j] + c;
      t[j - 1] = (u32)x;
      c = x >> 32;
    }
    t[7] = t8 + (u32)c;
  }
  fp_reduce_once<P>(t);
  Fp<P> r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = t[j];
  return r;
}
template <class P>
KDEV Fp<P> fp_sqr(const Fp<P>& a) {
  return fp_mul<P>(a, a);
}
// Montgomery -> canonical integer (ark-ff `into_bigint`): one Montgomery reduction of (a, 0)
template <class P>
KDEV void fp_from_mont(u32* out, const Fp<P>& a) {
  u32 t[8];
#pragma unroll
  for (int j = 0; j < 8; j++) t[j] = a.l[j];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u32 m = t[0] * P::INV;
    u64 x = (u64)m * P::MOD[0] + t[0];
    u64 c = x >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      x = (u64)m * P::MOD[j] + t[j] + c;
      t[j - 1] = (u32)x;
      c = x >> 32;
    }
    t[7] = (u32)c;
  }
  fp_reduce_once<P>(t);
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = t[j];
}
template <class P>
KDEV Fp<P> fp_to_mont(const u32* canon) {
  Fp<P> a, r2;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    a.l[j] = canon[j];
    r2.l[j] = P::R2[j];
  }
  return fp_mul<P>(a, r2);
}

// ---------------------------------------------------------------------------------------------
// Fq: hand-scheduled gfx950 streams (bn254_field_asm.hip.h) replace the portable templates above.
// The portable code stays the implementation for Fr and the reference the on-device self-test
// (k_selftest_field) compares the streams with, through FqParamsRef.
// ---------------------------------------------------------------------------------------------
struct FqParamsRef : FqParams {};   // same constants, but takes the portable template path
struct FrParamsRef : FrParams {};
}  // namespace bn254
#ifndef KEAKI_PORTABLE_FIELD
#include "bn254_field_asm.hip.h"
namespace bn254 {
template <> KDEV Fq fp_mul<FqParams>(const Fq& a, const Fq& b) { Fq r; fq_mul_asm(r.l, a.l, b.l); return r; }
template <> KDEV Fq fp_add<FqParams>(const Fq& a, const Fq& b) { Fq r; fq_add_asm(r.l, a.l, b.l); return r; }
template <> KDEV Fq fp_sub<FqParams>(const Fq& a, const Fq& b) { Fq r; fq_sub_asm(r.l, a.l, b.l); return r; }
template <> KDEV Fq fp_neg<FqParams>(const Fq& a) { Fq r; fq_neg_asm(r.l, a.l); return r; }
template <> KDEV void fp_from_mont<FrParams>(u32* out, const Fr& a) { fr_from_mont_asm(out, a.l); }   // the MSM's digit extraction
}  // namespace bn254
#endif
namespace bn254 {

// a^(p-2). Not inlined: 254 squarings + 110 products, in the 29-bit lazy limbs of fq29_core.hip.h (every intermediate stays below 2p). The
// reference the division-step inverse below is tested against (the shipped fq_inv since round 3).
static __device__ __noinline__ Fq fq_inv_fermat(const Fq& a) {
  const U29 one = u29_const(Fq29Params::ONE);
  const U29 base = u29_mul(u29_from_sat_shift5(a.l), one);        // 2^261-form, below 2p
  U29 acc = one;
  for (int i = 253; i >= 0; i--) {
    acc = u29_sqr(acc);
    if ((FQ_PM2[i >> 5] >> (i & 31)) & 1) acc = u29_mul(acc, base);
  }
  Fq r;
  u29_pack_canonical(r.l, u29_mul(acc, u29_const(Fq29Params::R256)));   // back to the 2^256 form, canonical
  return r;
}

// Inverse by the binary extended GCD (HAC 14.61 shape): ~1.4 x 254 rounds of 256-bit shifts and subtractions (~30 K instructions)
// instead of the 380 products (~115 K) of the Fermat ladder above. Same value (the inverse is unique). Works on the integer aR, so
// the plain inverse a^-1 R^-1 is brought back to Montgomery form by one product with R^3. Trip counts are data dependent: meant for
// single-lane tails (k_msm_final); 0 -> 0 like the ladder.
static __device__ __noinline__ Fq fq_inv_xgcd(const Fq& a) {
  u32 u[8], v[8];
  Fq x1 = fp_zero<FqParams>(), x2 = fp_zero<FqParams>();
  u32 nz = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { u[i] = a.l[i]; v[i] = FqParams::MOD[i]; nz |= a.l[i]; }
  if (!nz) return x1;
  x1.l[0] = 1;
  auto is_one = [](const u32* t) { u32 o = t[0] ^ 1u; for (int i = 1; i < 8; i++) o |= t[i]; return o == 0; };
  auto shr1 = [](u32* t) {
#pragma unroll
    for (int i = 0; i < 7; i++) t[i] = __builtin_amdgcn_alignbit(t[i + 1], t[i], 1);
    t[7] >>= 1;
  };
  auto half_mod = [&](Fq& x) {          // x / 2 mod p for x < p
    u32 carry = 0;
    if (x.l[0] & 1u) {
      u64 c = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) { c += (u64)x.l[i] + FqParams::MOD[i]; x.l[i] = (u32)c; c >>= 32; }
      carry = (u32)c;                    // x + p < 2^255: always 0, kept for clarity
    }
    shr1(x.l);
    x.l[7] |= carry << 31;
  };
  auto geq = [](const u32* s, const u32* t) {
#pragma unroll
    for (int i = 7; i >= 0; i--) if (s[i] != t[i]) return s[i] > t[i];
    return true;
  };
  auto sub = [](u32* s, const u32* t) {  // s -= t, s >= t
    u64 b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 d = (u64)s[i] - t[i] - b; s[i] = (u32)d; b = (d >> 32) & 1u; }
  };
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1u)) { shr1(u); half_mod(x1); }
    while (!(v[0] & 1u)) { shr1(v); half_mod(x2); }
    if (geq(u, v)) { sub(u, v); x1 = fp_sub<FqParams>(x1, x2); }
    else { sub(v, u); x2 = fp_sub<FqParams>(x2, x1); }
  }
  Fq y = is_one(u) ? x1 : x2, r3;
#pragma unroll
  for (int i = 0; i < 8; i++) r3.l[i] = FqParams::R3[i];
  return fp_mul<FqParams>(y, r3);
}

// Inverse by Bernstein-Yang division steps ("safegcd": the constant-time form libsecp256k1 uses, here for BN254's p): 20 rounds of 30 divsteps
// on the low 30 bits of (f, g) -- each round yields a 2 x 2 transition matrix that is then applied to the full (f, g) and, modulo p, to (d, e)
// -- bring g to 0 and f to +-1 whatever the input is (590 steps suffice for 256-bit inputs), and d to +-1/x. No branch depends on the data:
// every lane of a wave does the same ~16 K plain 32-bit instructions, where the Fermat ladder spends 254 squarings + 60 products (~57 K, most
// of them multiply-adds). Values are nine signed 30-bit limbs. `a`: a canonical word value x < p (whatever form the caller reads into it); returns x^-1 mod p as canonical words (0 for 0).
// (csrc/models: the same loop in Python against pow(x, -1, p); the device self-test compares it with the ladder.)
static __device__ __noinline__ Fq fq_inv_safegcd_words(const Fq a) {      // by value: pointers to the caller's registers would send them to scratch memory
  constexpr int32_t M30 = (int32_t)(0xFFFFFFFFu >> 2);
  constexpr int32_t P30[9] = {0x187cfd47, 0x3082305b, 0x071ca8d3, 0x205aa45a, 0x01585d97, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x00003064};
  constexpr u32 PINV30 = 0x1b799c77u;                 // p^-1 mod 2^30
  int32_t d[9], e[9], f[9], g[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int s = 30 * i, w = s >> 5, o = s & 31;
    const u32 lo = a.l[w], hi = (w + 1 < 8) ? a.l[w + 1] : 0u;
    const u32 v = o ? __builtin_amdgcn_alignbit(hi, lo, o) : lo;
    g[i] = (int32_t)(i < 8 ? (v & (u32)M30) : v);
    f[i] = P30[i]; d[i] = 0; e[i] = i == 0 ? 1 : 0;
  }
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; it++) {
    // 30 division steps on the low bits: the transition matrix (u v; q r), scaled by 2^30
    u32 u = 1, v = 0, q = 0, r = 1, ff = (u32)f[0], gg = (u32)g[0];
#pragma unroll 1
    for (int i = 0; i < 30; i++) {
      u32 c1 = (u32)(zeta >> 31);
      const u32 c2 = 0u - (gg & 1u);
      const u32 x = (ff ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
      gg += x & c2; q += y & c2; r += z & c2;
      c1 &= c2;
      zeta = (int32_t)((u32)zeta ^ c1) - 1;
      ff += gg & c1; u += q & c1; v += r & c1;
      gg >>= 1; u <<= 1; v <<= 1;
    }
    const int32_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
    {  // (d, e) <- (u d + v e, q d + r e) / 2^30 mod p: a multiple of p makes the low 30 bits vanish
      const int32_t sd = d[8] >> 31, se = e[8] >> 31;
      int32_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
      long long cd = (long long)tu * d[0] + (long long)tv * e[0], ce = (long long)tq * d[0] + (long long)tr * e[0];
      md -= (int32_t)((PINV30 * (u32)cd + (u32)md) & (u32)M30);
      me -= (int32_t)((PINV30 * (u32)ce + (u32)me) & (u32)M30);
      cd += (long long)P30[0] * md; ce += (long long)P30[0] * me;
      cd >>= 30; ce >>= 30;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        cd += (long long)tu * d[i] + (long long)tv * e[i] + (long long)P30[i] * md;
        ce += (long long)tq * d[i] + (long long)tr * e[i] + (long long)P30[i] * me;
        d[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e[i - 1] = (int32_t)ce & M30; ce >>= 30;
      }
      d[8] = (int32_t)cd; e[8] = (int32_t)ce;
    }
    {  // (f, g) <- (u f + v g, q f + r g) / 2^30 (exact)
      long long cf = (long long)tu * f[0] + (long long)tv * g[0], cg = (long long)tq * f[0] + (long long)tr * g[0];
      cf >>= 30; cg >>= 30;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        cf += (long long)tu * f[i] + (long long)tv * g[i];
        cg += (long long)tq * f[i] + (long long)tr * g[i];
        f[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g[i - 1] = (int32_t)cg & M30; cg >>= 30;
      }
      f[8] = (int32_t)cf; g[8] = (int32_t)cg;
    }
  }
  // d = +-1/x with the sign of f: add p if negative, negate if f < 0, carry, add p again if still negative -> [0, p)
  {
    const int32_t neg = f[8] >> 31;
    int32_t add = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = ((d[i] + (P30[i] & add)) ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    add = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] += P30[i] & add;
#pragma unroll
    for (int i = 0; i < 8; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
  }
  // nine 30-bit limbs -> eight words
  Fq out;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int s = 32 * j, k = s / 30, o = s % 30;
    unsigned long long x = ((unsigned long long)(u32)d[k + 1] << 30) | (u32)d[k];
    if (k + 2 < 9) x |= (unsigned long long)(u32)d[k + 2] << 60;
    out.l[j] = (u32)(x >> o);
  }
  return out;
}
// a^-1 in the 2^256 Montgomery form through the division steps: the integer a R has the inverse a^-1 R^-1; one product by R^3 brings the form back
static __device__ __noinline__ Fq fq_inv_safegcd(const Fq& a) {
  Fq r3;
  const Fq y = fq_inv_safegcd_words(a);
#pragma unroll
  for (int i = 0; i < 8; i++) r3.l[i] = FqParams::R3[i];
  return fp_mul<FqParams>(y, r3);
}
// the inverse every affine conversion uses (one per item in the batched scalar-mult and encapsulation kernels)
KDEV Fq fq_inv(const Fq& a) { return fq_inv_safegcd(a); }

// shorthand for Fq
KDEV Fq operator+(const Fq& a, const Fq& b) { return fp_add<FqParams>(a, b); }
KDEV Fq operator-(const Fq& a, const Fq& b) { return fp_sub<FqParams>(a, b); }
KDEV Fq operator*(const Fq& a, const Fq& b) { return fp_mul<FqParams>(a, b); }
KDEV Fq operator-(const Fq& a) { return fp_neg<FqParams>(a); }
KDEV Fq fq_sqr(const Fq& a) { return fp_sqr<FqParams>(a); }
KDEV Fq fq_dbl(const Fq& a) { return fp_dbl<FqParams>(a); }
KDEV Fq fq_zero() { return fp_zero<FqParams>(); }
KDEV Fq fq_one() { return fp_one<FqParams>(); }
KDEV bool fq_is_zero(const Fq& a) { return fp_is_zero<FqParams>(a); }
KDEV bool fq_eq(const Fq& a, const Fq& b) { return fp_eq<FqParams>(a, b); }

// ---------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2+1)
// ---------------------------------------------------------------------------------------------
KDEV Fq2 operator+(const Fq2& a, const Fq2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
KDEV Fq2 operator-(const Fq2& a, const Fq2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
KDEV Fq2 operator-(const Fq2& a) { return {-a.c0, -a.c1}; }
KDEV Fq2 fq2_dbl(const Fq2& a) { return {fq_dbl(a.c0), fq_dbl(a.c1)}; }
KDEV Fq2 fq2_conj(const Fq2& a) { return {a.c0, -a.c1}; }
KDEV Fq2 fq2_zero() { return {fq_zero(), fq_zero()}; }
KDEV Fq2 fq2_one() { return {fq_one(), fq_zero()}; }
KDEV bool fq2_is_zero(const Fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
KDEV bool fq2_eq(const Fq2& a, const Fq2& b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
// Each component is ONE dual product in the 9 x 29-bit limbs of fq29_core.hip.h -- two products sharing a single Montgomery
// reduction, no carry instructions: c0 = a0 b0 + (64p - 32 a1) b1, c1 = a0 b1 + a1 b0. The left factors enter shifted by 5 bits
// (2^256 -> 2^261 Montgomery form), results come back as canonical saturated residues. ~770 instructions against ~1020 for the
// Karatsuba form on the saturated streams (3 products + 5 modular additions).
KDEV Fq2 fq2_mul_inl(const Fq2& a, const Fq2& b) {
  const U29 A0 = u29_from_sat_shift5(a.c0.l), A1 = u29_from_sat_shift5(a.c1.l);
  const U29 B0 = u29_from_sat_plain(b.c0.l), B1 = u29_from_sat_plain(b.c1.l);
  U29 NA1;
#pragma unroll
  for (int i = 0; i < 9; i++) NA1.l[i] = Fq29Params::K64[i] - A1.l[i];
  Fq2 r;
  u29_pack_canonical(r.c0.l, u29_mul2(A0, B0, NA1, B1));
  u29_pack_canonical(r.c1.l, u29_mul2(A0, B1, A1, B0));
  return r;
}
KDEV Fq2 fq2_sqr_inl(const Fq2& a) {  // (a0+a1)(a0-a1) + 2 a0 a1 u
  Fq m = a.c0 * a.c1;
  return {(a.c0 + a.c1) * (a.c0 - a.c1), fq_dbl(m)};
}
#ifdef KEAKI_FQ2_OUTLINE
// Translation units whose kernels are Fq2-heavy (G2 MSM, G2 ladders, the pairing tower) call the
// Fq2 product as a real function: ~25x less code, minutes less compile time, and the hot loops stay
// inside the instruction cache.
static __device__ __noinline__ Fq2 fq2_mul_ol(const Fq2 a, const Fq2 b) { return fq2_mul_inl(a, b); }
static __device__ __noinline__ Fq2 fq2_sqr_ol(const Fq2 a) { return fq2_sqr_inl(a); }
KDEV Fq2 operator*(const Fq2& a, const Fq2& b) { return fq2_mul_ol(a, b); }
KDEV Fq2 fq2_sqr(const Fq2& a) { return fq2_sqr_ol(a); }
#else
KDEV Fq2 operator*(const Fq2& a, const Fq2& b) { return fq2_mul_inl(a, b); }
KDEV Fq2 fq2_sqr(const Fq2& a) { return fq2_sqr_inl(a); }
#endif
KDEV Fq2 fq2_mul_fq(const Fq2& a, const Fq& k) { return {a.c0 * k, a.c1 * k}; }
KDEV Fq2 fq2_mul_xi(const Fq2& a) {  // (9+u) a
  Fq t0 = fq_dbl(fq_dbl(fq_dbl(a.c0))) + a.c0;
  Fq t1 = fq_dbl(fq_dbl(fq_dbl(a.c1))) + a.c1;
  return {t0 - a.c1, t1 + a.c0};
}
KDEV Fq2 fq2_inv(const Fq2& a) {
  Fq n = fq_sqr(a.c0) + fq_sqr(a.c1);
  Fq ni = fq_inv(n);
  return {a.c0 * ni, -(a.c1 * ni)};
}
KDEV Fq2 fq2_cneg(const Fq2& a, bool neg) { return {fp_cneg<FqParams>(a.c0, neg), fp_cneg<FqParams>(a.c1, neg)}; }

// field-generic front (lets the curve code be written once for Fq and Fq2)
KDEV Fq f_sqr(const Fq& a) { return fq_sqr(a); }
KDEV Fq2 f_sqr(const Fq2& a) { return fq2_sqr(a); }
KDEV Fq f_dbl(const Fq& a) { return fq_dbl(a); }
KDEV Fq2 f_dbl(const Fq2& a) { return fq2_dbl(a); }
KDEV bool f_is_zero(const Fq& a) { return fq_is_zero(a); }
KDEV bool f_is_zero(const Fq2& a) { return fq2_is_zero(a); }
KDEV bool f_eq(const Fq& a, const Fq& b) { return fq_eq(a, b); }
KDEV bool f_eq(const Fq2& a, const Fq2& b) { return fq2_eq(a, b); }
KDEV Fq f_inv(const Fq& a) { return fq_inv(a); }
KDEV Fq2 f_inv(const Fq2& a) { return fq2_inv(a); }
KDEV Fq f_cneg(const Fq& a, bool n) { return fp_cneg<FqParams>(a, n); }
KDEV Fq2 f_cneg(const Fq2& a, bool n) { return fq2_cneg(a, n); }
template <class F> KDEV F f_zero();
template <> KDEV Fq f_zero<Fq>() { return fq_zero(); }
template <> KDEV Fq2 f_zero<Fq2>() { return fq2_zero(); }
template <class F> KDEV F f_one();
template <> KDEV Fq f_one<Fq>() { return fq_one(); }
template <> KDEV Fq2 f_one<Fq2>() { return fq2_one(); }

}  // namespace bn254
