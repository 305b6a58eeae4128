// Jacobian scalar multiplication of BN254 G1 points in the 9 x 29-bit lazy representation (fq29.cuh): the inner loop of the FK23
// butterflies (fft_g1.hip; reference src/kzg.rs:182-200 = ark-poly group FFTs, every butterfly one `Group * ScalarField`).
// Left-to-right NAF (digit_i = bit_(i+1)(3k) - bit_(i+1)(k): ~85 additions for 254 doublings, no table), doubling dbl-2009-l with
// D = 4 X Y^2 taken as a product (keeps every value small), addition add-2007-bl with the addend's Z^2, Z^3 precomputed.
// Value bounds (multiples of p) on the running point: X < 17.6, Y < 19.3, Z < 3.4; the addend is below 1.2 (u29_from_fq). A limb-exact
// model with 64-bit overflow assertions ran full 254-bit multiplications before this was written; test: k_selftest_j29 and the
// FK23 parity tests.
#pragma once
#include "bn254_curve.cuh"
#include "fq29.cuh"

namespace bn254 {

struct J29 {
  U29 x, y, z;
};
// carry(k * a), k <= 4, a carried
KDEV U29 u29_scale(const U29& a, u32 k) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = k * a.l[i];
  return u29_carry(t);
}
// carry(a - 2 b + K), K biased by 2^31
KDEV U29 u29_sub2x(const U29& a, const U29& b, const u32 (&K)[9]) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = a.l[i] - 2u * b.l[i] + K[i];
  return u29_carry(t);
}
KDEV J29 j29_dbl(const J29& p) {
  const U29 A = u29_sqr(p.x), B = u29_sqr(p.y), C = u29_sqr(B), S = u29_mul(p.x, B);
  const U29 D = u29_scale(S, 4), E = u29_scale(A, 3);
  J29 r;
  r.x = u29_sub2x(u29_sqr(E), D, Q29::K16W);
  const U29 T = u29_sub_raw(D, r.x, Q29::K32);
  r.y = u29_sub2x(u29_mul(E, T), u29_scale(C, 4), Q29::K16W);
  r.z = u29_scale(u29_mul(p.y, p.z), 2);
  return r;
}
// a + (X2, Y2, Z2) with Z2Z2 = Z2^2, Z2cu = Z2^3. special: 0 = ordinary sum (returned), 1 = the two points are equal (the caller doubles),
// 2 = they are opposite (the sum is the identity). In a ladder k P with k < r that only happens for k = r - 2 (running multiple -P,
// last digit -1), but it costs three instructions to notice (zero filter of fq29_core.cuh).
KDEV J29 j29_add(const J29& a, const U29& X2, const U29& Y2, const U29& Z2, const U29& Z2Z2, const U29& Z2cu, int& special) {
  const U29 Z1Z1 = u29_sqr(a.z), U1 = u29_mul(a.x, Z2Z2), U2 = u29_mul(X2, Z1Z1), S1 = u29_mul(a.y, Z2cu);
  const U29 S2 = u29_mul(Y2, u29_mul(a.z, Z1Z1));
  const U29 H = u29_sub(U2, U1, Q29::K2);
  special = 0;
  if (u29_maybe_zero(H)) {
    if (u29_is_zero(H)) {
      special = u29_is_zero(u29_sub(S2, S1, Q29::K2)) ? 1 : 2;
      return a;
    }
  }
  const U29 I = u29_scale(u29_sqr(H), 4), J = u29_mul(H, I);
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = 2u * (S2.l[i] - S1.l[i] + Q29::K2[i]);
  const U29 rr = u29_carry(t);
  const U29 V = u29_mul(U1, I);
  J29 r;
  r.x = u29_sub3(u29_sqr(rr), J, V);
  const U29 T = u29_sub_raw(V, r.x, Q29::K16);
  r.y = u29_sub2x(u29_mul(rr, T), u29_mul(S1, J), Q29::K4W);
  r.z = u29_scale(u29_mul(u29_mul(a.z, Z2), H), 2);
  return r;
}

// k * P, P Jacobian (saturated, any Z), k a Montgomery Fr below r. P of prime order or infinity.
KDEV Jac<Fq> jac_scalar_mul_u29(const Jac<Fq>& p, const Fr& k_mont) {
  if (jac_is_inf(p)) return jac_inf<Fq>();
  u32 k[8], h[8];
  fp_from_mont<FrParams>(k, k_mont);
  {  // h = 3 k (< 2^256)
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += 3ull * k[j]; h[j] = (u32)c; c >>= 32; }
  }
  const U29 X2 = u29_from_fq(p.x), Y2 = u29_from_fq(p.y), Z2 = u29_from_fq(p.z);
  const U29 Z2Z2 = u29_sqr(Z2), Z2cu = u29_mul(Z2, Z2Z2);
  U29 zero;
#pragma unroll
  for (int i = 0; i < 9; i++) zero.l[i] = 0;
  const U29 Y2n = u29_sub(zero, Y2, Q29::K2);
  J29 acc;
  acc.x = X2; acc.y = Y2; acc.z = Z2;
  bool empty = true;
#pragma unroll 1
  for (int it = 0; it < 255; it++) {
    const u32 hb = h[7] >> 31, kb = k[7] >> 31;
#pragma unroll
    for (int j = 7; j > 0; j--) { h[j] = (h[j] << 1) | (h[j - 1] >> 31); k[j] = (k[j] << 1) | (k[j - 1] >> 31); }
    h[0] <<= 1; k[0] <<= 1;
    if (!empty) acc = j29_dbl(acc);
    if (hb != kb) {
      const bool neg = kb != 0;                         // digit = hb - kb
      if (empty) {
        acc.x = X2; acc.y = neg ? Y2n : Y2; acc.z = Z2;
        empty = false;
      } else {
        int special;
        acc = j29_add(acc, X2, neg ? Y2n : Y2, Z2, Z2Z2, Z2cu, special);
        if (special == 1) acc = j29_dbl(acc);
        if (special == 2) empty = true;
      }
    }
  }
  if (empty) return jac_inf<Fq>();
  return {u29_to_fq(acc.x), u29_to_fq(acc.y), u29_to_fq(acc.z)};
}

}  // namespace bn254
