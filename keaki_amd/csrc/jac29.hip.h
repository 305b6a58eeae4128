// Jacobian scalar multiplication of BN254 G1 points in the 9 x 29-bit lazy representation (fq29.hip.h): the inner loop of the FK23
// butterflies (fft_g1.hip; reference src/kzg.rs:182-200 = ark-poly group FFTs, every butterfly one `Group * ScalarField`).
// GLV split k = k1 + k2 lambda (128-bit halves, phi(x, y) = (beta x, y)), windows over ONE effective-affine table per point (every entry on one
// isomorphic curve: mixed additions without an inversion, see below), doubling dbl-2009-l with D = 4 X Y^2 taken as a product (keeps every value
// small) and no scaling passes, mixed addition madd without factors of two; the Y coordinate of both formulas is one dual stream (two products,
// one reduction). The butterflies' add / subtract pair shares its products (j29_addsub); two-term sums share their doubling chain (j29_mul2_uniform).
// Value bounds (multiples of p) on the running point: X < 17.6, Y < 3.8 (a dual stream's output), Z < 2.1; the addend is below 1.2
// (u29_from_fq). A limb-exact model with 64-bit overflow assertions (models/model_jac29.py, run by the CPU suite) executes whole ladders
// with these formulas; tests: k_selftest_j29 and the FK23 parity tests.
#pragma once
#include "bn254_curve.hip.h"
#include "fq29.hip.h"

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
KDEV J29 j29_dbl(const J29& p) {
  // dbl-2009-l with D = 4 X Y^2 as a product. No scaling passes: the factors of two ride on raw doubled limbs (a stream's wide operand).
  const U29 A = u29_sqr(p.x), B = u29_sqr(p.y);
  U29 b2, y2, n4, t;
#pragma unroll
  for (int i = 0; i < 9; i++) { b2.l[i] = 2u * B.l[i]; y2.l[i] = 2u * p.y.l[i]; n4.l[i] = Q29::K16W[i] - 4u * B.l[i]; }
  const U29 S2 = u29_mul(b2, p.x);                       // 2 X B = D / 2 (p.x carried)
  const U29 E = u29_scale(A, 3);
  const U29 EE = u29_sqr(E);
  J29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = EE.l[i] - 4u * S2.l[i] + Q29::K16W[i];                 // E^2 - 2D + 16p
  r.x = u29_carry(t);
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = 2u * S2.l[i] - r.x.l[i] + Q29::K32[i];                 // D - X3 + 32p
  const U29 T = u29_carry(t);
  // Y3 = E T - 8 B^2 as ONE dual stream (two products, one reduction): E T + (2B)(16p - 4B). 2B stays raw (limbs < 2^30: the wide operand),
  // 16p - 4B is carried. The square C = B^2 and the product E T are no separate streams any more (models/model_jac29.py).
  r.y = u29_mul2(E, T, b2, u29_carry(n4));
  r.z = u29_mul(y2, p.z);                                // (2Y) Z
  return r;
}
KDEV U29 ld9(const u32* __restrict__ p) {
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = p[i];
  return r;
}
// (u + v, u - v) of an FK23 butterfly in the lazy limbs. u, v: Jacobian, every coordinate below 32 p with carried limbs (a saturated residue
// through u29_from_sat_shift5, or a ladder's running point), neither the identity. The products up to H = U2 - U1 and Z3 = Z1 Z2 H are shared
// by the two results; r^2 and the dual stream of Y are per result (add-1998-cmo-2, r' = -S2 - S1 for the difference). Returns false when
// u = +-v (H = 0): the caller takes the generic path. Bounds: models/model_jac29.py (j29_addsub). 15 streams + 2 x 2 instead of two saturated
// additions of 16 products each and three conversions of the ladder's result.
KDEV bool j29_addsub(const J29& u, const J29& v, J29& sum, J29& diff) {
  const U29 Z1Z1 = u29_sqr(u.z), Z2Z2 = u29_sqr(v.z);
  const U29 U1 = u29_mul(u.x, Z2Z2), U2 = u29_mul(v.x, Z1Z1);
  const U29 S1 = u29_mul(u.y, u29_mul(v.z, Z2Z2)), S2 = u29_mul(v.y, u29_mul(u.z, Z1Z1));
  const U29 H = u29_sub(U2, U1, Q29::K4);
  if (u29_maybe_zero(H)) {
    if (u29_is_zero(H)) return false;
  }
  sum.z = u29_mul(u29_mul(u.z, v.z), H);
  diff.z = sum.z;
  const U29 HH = u29_sqr(H), HHH = u29_mul(H, HH), V = u29_mul(U1, HH);
  U29 ns, t;
#pragma unroll
  for (int i = 0; i < 9; i++) { ns.l[i] = Q29::K2[i] - S1.l[i]; t.l[i] = Q29::K4[i] - S2.l[i] - S1.l[i]; }
  {
    const U29 rr = u29_sub(S2, S1, Q29::K2);
    sum.x = u29_sub3(u29_sqr(rr), HHH, V);
    sum.y = u29_mul2(rr, u29_sub(V, sum.x, Q29::K16), ns, HHH);
  }
  {
    const U29 rr = u29_carry(t);
    diff.x = u29_sub3(u29_sqr(rr), HHH, V);
    diff.y = u29_mul2(rr, u29_sub(V, diff.x, Q29::K16), ns, HHH);
  }
  return true;
}
KDEV J29 j29_from_sat(const Jac<Fq>& p) { return {u29_from_sat_shift5(p.x.l), u29_from_sat_shift5(p.y.l), u29_from_sat_shift5(p.z.l)}; }
KDEV Jac<Fq> j29_to_sat(const J29& a) { return {u29_to_fq(a.x), u29_to_fq(a.y), u29_to_fq(a.z)}; }

// ---- GLV: k = k1 + k2 lambda (mod r), |k1|, |k2| < 2^127, phi(x, y) = (beta x, y) = lambda (x, y) ------------------------------
// Babai rounding on the short basis (a1, b1), (a2, b2): c_i = (k g_i) >> 256, k1 = k - c1 a1 - c2 a2, k2 = -c1 b1 - c2 b2, all computed
// modulo 2^160 (the results fit 128 signed bits). out: magnitudes (5 words, top word 0) and signs.
KDEV void glv_mul_acc(u64* col, const u32* x, int xn, const u32* y, int yn) {   // col[i + j] += x[i] y[j], 32-bit columns with 64-bit sums (lazy carries)
#pragma unroll
  for (int i = 0; i < xn; i++)
#pragma unroll
    for (int j = 0; j < yn; j++) {
      const u64 p = (u64)x[i] * y[j];
      col[i + j] += (u32)p;
      col[i + j + 1] += p >> 32;
    }
}
KDEV void glv_decompose(const u32* k, u32* k1, bool& neg1, u32* k2, bool& neg2) {
  u32 c1[3], c2[5];
  {  // c1 = (k g1) >> 256  (< 2^65), c2 = (k g2) >> 256  (< 2^129)
    u64 col[14];
#pragma unroll
    for (int i = 0; i < 14; i++) col[i] = 0;
    glv_mul_acc(col, k, 8, GlvParams::G1, 3);
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) { c += col[i]; if (i >= 8) c1[i - 8] = (u32)c; c >>= 32; }
#pragma unroll
    for (int i = 0; i < 14; i++) col[i] = 0;
    glv_mul_acc(col, k, 8, GlvParams::G2, 5);
    c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) { c += col[i]; if (i >= 8) c2[i - 8] = (u32)c; c >>= 32; }
  }
  // t = c1 a1 + c2 a2, s = c1 (-b1), u = c2 b2 (b2 = a1), each modulo 2^160
  u32 t[5], s[5], uu[5];
  {
    u64 col[10];
#pragma unroll
    for (int i = 0; i < 10; i++) col[i] = 0;
    glv_mul_acc(col, c1, 3, GlvParams::A1, 2);
    glv_mul_acc(col, c2, 5, GlvParams::A2, 4);
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { c += col[i]; t[i] = (u32)c; c >>= 32; }
#pragma unroll
    for (int i = 0; i < 10; i++) col[i] = 0;
    glv_mul_acc(col, c1, 3, GlvParams::NB1, 4);
    c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { c += col[i]; s[i] = (u32)c; c >>= 32; }
#pragma unroll
    for (int i = 0; i < 10; i++) col[i] = 0;
    glv_mul_acc(col, c2, 5, GlvParams::A1, 2);
    c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { c += col[i]; uu[i] = (u32)c; c >>= 32; }
  }
  // k1 = k - t, k2 = s - u  (mod 2^160, two's complement), then sign / magnitude
  long long b = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) { long long d = (long long)k[i] - t[i] + b; k1[i] = (u32)d; b = d >> 32; }
  b = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) { long long d = (long long)s[i] - uu[i] + b; k2[i] = (u32)d; b = d >> 32; }
  neg1 = (k1[4] >> 31) != 0;
  neg2 = (k2[4] >> 31) != 0;
  {
    u64 c = neg1 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { c += neg1 ? (u32)~k1[i] : k1[i]; k1[i] = (u32)c; c >>= 32; }
    c = neg2 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { c += neg2 ? (u32)~k2[i] : k2[i]; k2[i] = (u32)c; c >>= 32; }
  }
}

// ---- effective-affine window tables (round 3) ------------------------------------------------------------------------------------
// The ladders add table entries to a running point ~43 .. 66 times. With Jacobian entries that is a general addition (12M + 4S); with AFFINE
// entries a mixed one (8M + 3S). An inversion per lane to normalise the entries costs more than that saves -- but no inversion is needed:
// a Jacobian point (X, Y, Z) of E: y^2 = x^3 + b IS the affine point (X, Y) of the isomorphic curve E_Z: y^2 = x^3 + b Z^6, and neither the
// doubling (a = 0) nor the mixed addition reads b. So: build the table with every entry brought to ONE common Z_g (the build records the
// ratio Z_(i+1) / Z_i of every step -- the H of the mixed addition -- and rescales entry i by the product of the later ratios: 4M + 1S per
// entry), run the whole ladder on E_(Z_g) with the entries as affine points, and multiply the result's Z by Z_g at the end (the trick of
// libsecp256k1's ecmult, here per lane). phi((x, y)) = (beta x, y) holds on every E_Z. An entry is (x, beta x, y): 27 words instead of 54.
// models/model_jac29.py runs these builds and ladders limb by limb.
struct J29A {            // table entry: affine on the working curve, with beta x for phi
  U29 x, xb, y;
};
// a (Jacobian on the working curve: X < 19 p, Y <= 4 p, limbs carried) + (x2, y2) affine. 8M + 3S without factors of two, Y3 as one dual
// stream. special: 0 = ordinary sum (returned), 1 = the two points are equal (the caller doubles), 2 = they are opposite (the sum is the identity);
// in a ladder k P with k < r that only happens for k = r - 2, but it costs three instructions to notice. H (= Z3 / Z1) goes to *ratio when the
// build asks for it.
KDEV J29 j29_madd(const J29& a, const U29& x2, const U29& y2, int& special, U29* ratio = nullptr) {
  const U29 Z1Z1 = u29_sqr(a.z);
  const U29 U2 = u29_mul(x2, Z1Z1);
  const U29 S2 = u29_mul(y2, u29_mul(a.z, Z1Z1));
  const U29 H = u29_sub(U2, a.x, Q29::K32);
  special = 0;
  if (u29_maybe_zero34(H)) {
    if (u29_is_zero(H)) {
      special = u29_is_zero(u29_sub(S2, a.y, Q29::K4)) ? 1 : 2;
      return a;
    }
  }
  if (ratio) *ratio = H;
  const U29 HH = u29_sqr(H), HHH = u29_mul(H, HH), V = u29_mul(a.x, HH);
  const U29 rr = u29_sub(S2, a.y, Q29::K4);
  J29 r;
  r.x = u29_sub3(u29_sqr(rr), HHH, V);
  const U29 T = u29_sub(V, r.x, Q29::K16);
  U29 ns;                                   // raw 8p - Y1: Y1 <= 4p with carried limbs, so every limb stays positive and below 1.5 * 2^30
#pragma unroll
  for (int i = 0; i < 9; i++) ns.l[i] = Q29::K8[i] - a.y.l[i];
  r.y = u29_mul2(rr, T, ns, HHH);
  r.z = u29_mul(a.z, H);
  return r;
}
// Table builds. `St` is where the entries wait: raw(i, x, y, ratio) keeps entry i as the chain produced it together with Z_i / Z_(i-1);
// get(i, x, y, ratio) reads it back; fin(i, x, y) stores the final entry (the store adds beta x). Both chains are LINEAR (entry i + 1 =
// entry i + one fixed affine point), so only the last entry is alive while the chain runs. Returns zfix: Z on E = Z on the working curve * zfix.
//   multiples: entries m P, m = 1..8, on the curve where P = (X, Y) is affine (isomorphic by Z_P): 2P = dbl(P), then + P six times
//   odd:       entries (2i + 1) P, i = 0..7, on the curve where D = 2P is affine (isomorphic by Z_D): P -> (X Z_D^2, Y Z_D^3, Z_P), then + D
// m1: the point in Jacobian limbs on the CURRENT working curve (E itself, or any curve of the family: the pair tables below build the second
// table on the curve of the first); the returned zfix is relative to that curve.
template <bool ODD, class St>
KDEV U29 j29_build_table_j(const J29& m1, St& st) {
  U29 ax, ay, zbase;                        // the affine point the chain adds, and the Z of the isomorphism
  J29 run;
  int special;
  if constexpr (ODD) {
    const J29 d = j29_dbl(m1);
    const U29 zd2 = u29_sqr(d.z), zd3 = u29_mul(d.z, zd2);
    ax = d.x; ay = d.y; zbase = d.z;
    run.x = u29_mul(m1.x, zd2); run.y = u29_mul(m1.y, zd3); run.z = m1.z;
    st.raw(0, run.x, run.y, run.z);
#pragma unroll 1
    for (int i = 1; i < 8; i++) {
      U29 h;
      run = j29_madd(run, ax, ay, special, &h);
      st.raw(i, run.x, run.y, h);
    }
  } else {
    ax = m1.x; ay = m1.y; zbase = m1.z;
    run.x = m1.x; run.y = m1.y; run.z = u29_one();
    st.raw(0, run.x, run.y, run.z);
    run = j29_dbl(run);
    st.raw(1, run.x, run.y, run.z);         // Z_2 / Z_1 = Z_2 (Z_1 = 1)
#pragma unroll 1
    for (int i = 2; i < 8; i++) {
      U29 h;
      run = j29_madd(run, ax, ay, special, &h);
      st.raw(i, run.x, run.y, h);
    }
  }
  // every entry to the last one's Z: zs = Z_7 / Z_i, x_i zs^2, y_i zs^3
  const U29 zfix = u29_mul(run.z, zbase);
  U29 zs, nextratio;
  {
    U29 x, y;
    st.get(7, x, y, nextratio);             // before fin(7): the ratio waits in the slot of beta x
  }
  st.fin(7, run.x, run.y);
#pragma unroll 1
  for (int i = 6; i >= 0; i--) {
    zs = (i == 6) ? nextratio : u29_mul(zs, nextratio);
    U29 x, y, ratio;
    st.get(i, x, y, ratio);
    const U29 zs2 = u29_sqr(zs);
    st.fin(i, u29_mul(x, zs2), u29_mul(y, u29_mul(zs, zs2)));
    nextratio = ratio;
  }
  return zfix;
}

template <bool ODD, class St>
KDEV U29 j29_build_table(const Jac<Fq>& p, St& st) {
  J29 m1;
  m1.x = u29_from_fq(p.x); m1.y = u29_from_fq(p.y); m1.z = u29_from_fq(p.z);
  return j29_build_table_j<ODD>(m1, st);
}

// signed 4-bit digits of a 127-bit magnitude: d_j in [-8, 8], j = 0..32 (digit 32 is the carry); magnitudes packed 8 per word, signs apart
KDEV void glv_fixed_digits(const u32* k, u32* dig, u32* sg) {
#pragma unroll
  for (int w = 0; w < 5; w++) dig[w] = 0;
  sg[0] = sg[1] = 0;
  u32 car = 0;
#pragma unroll 1
  for (int j = 0; j < 33; j++) {
    u32 d = ((j < 32) ? ((k[j >> 3] >> ((j & 7) * 4)) & 15u) : 0u) + car;
    car = d > 8u ? 1u : 0u;
    const u32 mag = car ? 16u - d : d;
    dig[j >> 3] |= mag << ((j & 7) * 4);
    sg[j >> 5] |= car << (j & 31);
  }
}

// k * P, P Jacobian (saturated, any Z), k a Montgomery Fr below r. P of prime order or infinity.
// GLV split k = k1 + k2 lambda (128-bit halves), then fixed signed 4-bit windows on both halves over ONE table {1..8} P (phi of an entry
// is the same entry with beta x): 33 windows x (4 doublings + 2 mixed additions) = 129 doublings + 66 additions + the table build. Every
// lane of a wave does the same work whatever its scalar -- a NAF ladder with per-lane scalars executes its addition in nearly every
// iteration (some lane always has a non-zero digit): 258 additions. The table lives in private memory (per-lane index).
// The *_j forms return the running point in the lazy limbs (false: the product is the identity): the FK23 butterflies add and subtract it
// without a detour through the saturated words.
struct J29ArrTable {
  J29A* T;
  U29 beta;
  KDEV void raw(int i, const U29& x, const U29& y, const U29& ratio) { T[i].x = x; T[i].y = y; T[i].xb = ratio; }
  KDEV void get(int i, U29& x, U29& y, U29& ratio) const { x = T[i].x; y = T[i].y; ratio = T[i].xb; }
  KDEV void fin(int i, const U29& x, const U29& y) { T[i].x = x; T[i].y = y; T[i].xb = u29_mul(x, beta); }
};
struct J29PrivTable {
  J29A T[8];
  U29 beta;
  KDEV void raw(int i, const U29& x, const U29& y, const U29& ratio) { T[i].x = x; T[i].y = y; T[i].xb = ratio; }
  KDEV void get(int i, U29& x, U29& y, U29& ratio) const { x = T[i].x; y = T[i].y; ratio = T[i].xb; }
  KDEV void fin(int i, const U29& x, const U29& y) { T[i].x = x; T[i].y = y; T[i].xb = u29_mul(x, beta); }
};
KDEV bool jac_scalar_mul_u29_j(const Jac<Fq>& p, const Fr& k_mont, J29& out) {
  if (jac_is_inf(p)) return false;
  u32 k[8], k1[5], k2[5];
  bool neg1, neg2;
  fp_from_mont<FrParams>(k, k_mont);
  glv_decompose(k, k1, neg1, k2, neg2);
  J29PrivTable tb;
  tb.beta = u29_const(GlvParams::BETA29);
  const U29 zfix = j29_build_table<false>(p, tb);
  U29 zero;
#pragma unroll
  for (int i = 0; i < 9; i++) zero.l[i] = 0;
  u32 dig1[5], dig2[5], sg1[2], sg2[2];
  glv_fixed_digits(k1, dig1, sg1);
  glv_fixed_digits(k2, dig2, sg2);
  J29 acc;
  acc.x = zero; acc.y = zero; acc.z = zero;
  bool empty = true;
#pragma unroll 1
  for (int j = 32; j >= 0; j--) {
    if (!empty) { acc = j29_dbl(acc); acc = j29_dbl(acc); acc = j29_dbl(acc); acc = j29_dbl(acc); }
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
      const u32 mag = ((which ? dig2[j >> 3] : dig1[j >> 3]) >> ((j & 7) * 4)) & 15u;
      if (mag) {
        const bool neg = (((which ? sg2[j >> 5] : sg1[j >> 5]) >> (j & 31)) & 1u) != (which ? neg2 : neg1);
        const J29A& e = tb.T[mag - 1];
        const U29 ex = which ? e.xb : e.x;
        const U29 ey = neg ? u29_sub(zero, e.y, Q29::K4) : e.y;      // table y < 2 p (a product's output)
        if (empty) {
          acc.x = ex; acc.y = ey; acc.z = u29_one();
          empty = false;
        } else {
          int special;
          acc = j29_madd(acc, ex, ey, special);
          if (special == 1) acc = j29_dbl(acc);
          if (special == 2) empty = true;
        }
      }
    }
  }
  if (empty) return false;
  out.x = acc.x; out.y = acc.y; out.z = u29_mul(acc.z, zfix);
  return true;
}
KDEV Jac<Fq> jac_scalar_mul_u29(const Jac<Fq>& p, const Fr& k_mont) {
  J29 a;
  if (!jac_scalar_mul_u29_j(p, k_mont, a)) return jac_inf<Fq>();
  return {u29_to_fq(a.x), u29_to_fq(a.y), u29_to_fq(a.z)};
}


// ---- the same ladder with the window table in GLOBAL memory, one contiguous slot per lane (round 3) --------------------------------------
// With a scalar per lane every lane indexes its table differently. In private (scratch) memory the hardware interleaves the lanes dword by
// dword, so the 27 dwords of "entry e of lane l" lie in 27 different 256-byte rows, and a wave whose lanes ask for up to 8 different entries
// touches up to 8 x 27 rows per addition (with the 45-dword Jacobian entries of the first version: 48 - 64 GB of fabric traffic per butterfly
// stage of 2^20 lanes, profiles/r03_fk_pairing_hbm_traffic_pmc.json, and the ladder 19 % slower than with one scalar per wave). Here lane l
// owns 8 x 128 contiguous bytes of a workspace: an entry is ONE cache line whatever the other lanes read. Entry layout in words: [0..8] y,
// [9..17] x, [18..26] beta x (while the table is built: the ratio Z_i / Z_(i-1)), [27..31] unused. `tab`: this lane's 64 x 16 bytes.
constexpr u32 GTAB_UINT4_PER_LANE = 64;
KDEV void st9(u32* __restrict__ p, const U29& a) {
#pragma unroll
  for (int i = 0; i < 9; i++) p[i] = a.l[i];
}
struct J29GlobTable {
  u32* base;                 // this lane's 8 x 32 words
  U29 beta;
  // the fences keep the compiler from forwarding the stored values to the loads, i.e. from keeping the whole table alive in registers
  KDEV void raw(int i, const U29& x, const U29& y, const U29& ratio) { u32* e = base + 32 * i; st9(e + 9, x); st9(e, y); st9(e + 18, ratio); }
  KDEV void get(int i, U29& x, U29& y, U29& ratio) const {
    asm volatile("" ::: "memory");
    const u32* e = base + 32 * i;
    x = ld9(e + 9); y = ld9(e); ratio = ld9(e + 18);
  }
  KDEV void fin(int i, const U29& x, const U29& y) { u32* e = base + 32 * i; st9(e + 9, x); st9(e, y); st9(e + 18, u29_mul(x, beta)); }
};
KDEV bool jac_scalar_mul_gtab_u29_j(const Jac<Fq>& p, const Fr& k_mont, uint4* __restrict__ tab, J29& out) {
  if (jac_is_inf(p)) return false;
  u32 k[8], k1[5], k2[5];
  bool neg1, neg2;
  fp_from_mont<FrParams>(k, k_mont);
  glv_decompose(k, k1, neg1, k2, neg2);
  U29 zfix;
  {
    J29GlobTable tb;
    tb.base = reinterpret_cast<u32*>(tab);
    tb.beta = u29_const(GlvParams::BETA29);
    zfix = j29_build_table<false>(p, tb);
    asm volatile("" ::: "memory");
  }
  U29 zero;
#pragma unroll
  for (int i = 0; i < 9; i++) zero.l[i] = 0;
  u32 dig1[5], dig2[5], sg1[2], sg2[2];
  glv_fixed_digits(k1, dig1, sg1);
  glv_fixed_digits(k2, dig2, sg2);
  J29 acc;
  acc.x = zero; acc.y = zero; acc.z = zero;
  bool empty = true;
#pragma unroll 1
  for (int j = 32; j >= 0; j--) {
    if (!empty) { acc = j29_dbl(acc); acc = j29_dbl(acc); acc = j29_dbl(acc); acc = j29_dbl(acc); }
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
      const u32 mag = ((which ? dig2[j >> 3] : dig1[j >> 3]) >> ((j & 7) * 4)) & 15u;
      if (mag) {
        const bool neg = (((which ? sg2[j >> 5] : sg1[j >> 5]) >> (j & 31)) & 1u) != (which ? neg2 : neg1);
        const u32* e32 = reinterpret_cast<const u32*>(tab) + (mag - 1u) * 32u;
        const U29 ex = ld9(e32 + (which ? 18 : 9));
        const U29 y = ld9(e32);
        const U29 ey = neg ? u29_sub(zero, y, Q29::K4) : y;
        if (empty) {
          acc.x = ex; acc.y = ey; acc.z = u29_one();
          empty = false;
        } else {
          int special;
          acc = j29_madd(acc, ex, ey, special);
          if (special == 1) acc = j29_dbl(acc);
          if (special == 2) empty = true;
        }
      }
    }
  }
  if (empty) return false;
  out.x = acc.x; out.y = acc.y; out.z = u29_mul(acc.z, zfix);
  return true;
}
KDEV Jac<Fq> jac_scalar_mul_gtab_u29(const Jac<Fq>& p, const Fr& k_mont, uint4* __restrict__ tab) {
  J29 a;
  if (!jac_scalar_mul_gtab_u29_j(p, k_mont, tab, a)) return jac_inf<Fq>();
  return {u29_to_fq(a.x), u29_to_fq(a.y), u29_to_fq(a.z)};
}


// ---- the same product when EVERY lane of the wave multiplies ITS point by the SAME k (the FK23 stages with at least 64 blocks: one
// twiddle per wave). A uniform scalar costs no divergence whatever its digits are, so the windows may slide: width-5 NAF of both GLV
// halves -- odd digits in [-15, 15], on average one non-zero digit in six positions -- over ONE table of the odd multiples 1, 3 .. 15
// (phi of an entry is the entry with beta x): 129 doublings + ~43 mixed additions + the table build instead of 129 + 66. The digits are
// computed once per wave from the scalar's first-lane copy (scalar unit) and wait in LDS (`dig`: 2 x 132 bytes of this wave), one byte per
// position: (|d| + 1) / 2 in the low bits, the sign in bit 7. Bounds: j29_dbl and j29_madd map the running point's bound set into itself
// (header of this file), so the order of the operations does not matter.
constexpr int UNIFORM_DIG_STRIDE = 132;
// width-5 NAF digits of the two GLV halves of a wave-uniform scalar into dig[0 .. 2 * UNIFORM_DIG_STRIDE) (the first lane's copy decides)
KDEV void glv_uniform_digits(const Fr& k_mont, unsigned char* dig, bool& neg1, bool& neg2) {
  u32 k[8], k1[5], k2[5];
  fp_from_mont<FrParams>(k, k_mont);
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = (u32)__builtin_amdgcn_readfirstlane((int)k[i]);
  glv_decompose(k, k1, neg1, k2, neg2);
  // width-5 NAF, least significant digit first
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
    u32 a0 = h ? k2[0] : k1[0], a1 = h ? k2[1] : k1[1], a2 = h ? k2[2] : k1[2], a3 = h ? k2[3] : k1[3];      // < 2^127
#pragma unroll 1
    for (int pos = 0; pos < 129; pos++) {
      u32 byte = 0;
      if (a0 & 1u) {
        const u32 m = a0 & 31u;
        if (m < 16u) {                      // digit +m
          a0 -= m;
          byte = (m + 1u) >> 1;
        } else {                            // digit -(32 - m): the value goes UP to the next multiple of 32
          const u32 add = 32u - m;
          u64 c = (u64)a0 + add; a0 = (u32)c; c >>= 32;
          c += a1; a1 = (u32)c; c >>= 32;
          c += a2; a2 = (u32)c; c >>= 32;
          a3 += (u32)c;
          byte = ((add + 1u) >> 1) | 0x80u;
        }
      }
      dig[h * UNIFORM_DIG_STRIDE + pos] = (unsigned char)byte;
      a0 = (a0 >> 1) | (a1 << 31); a1 = (a1 >> 1) | (a2 << 31); a2 = (a2 >> 1) | (a3 << 31); a3 >>= 1;
    }
  }
}
// T: eight entries of table storage (the radix-4 kernel hands every ladder of a lane the same ones)
KDEV bool jac_scalar_mul_uniform_u29_t(const Jac<Fq>& p, const Fr& k_mont, unsigned char* dig, J29& out, J29A* T) {
  bool neg1, neg2;
  glv_uniform_digits(k_mont, dig, neg1, neg2);
  // table T[i] = (2 i + 1) P, affine on the working curve
  J29ArrTable tb = {T, u29_const(GlvParams::BETA29)};
  const U29 zfix = j29_build_table<true>(p, tb);
  U29 zero;
#pragma unroll
  for (int i = 0; i < 9; i++) zero.l[i] = 0;
  J29 acc;
  acc.x = zero; acc.y = zero; acc.z = zero;
  bool empty = true;
#pragma unroll 1
  for (int pos = 128; pos >= 0; pos--) {
    if (!empty) acc = j29_dbl(acc);
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
      const u32 byte = (u32)__builtin_amdgcn_readfirstlane((int)dig[which * UNIFORM_DIG_STRIDE + pos]);
      if (byte) {
        const bool neg = ((byte >> 7) != 0) != (which ? neg2 : neg1);
        const J29A& e = T[(byte & 0x7Fu) - 1u];
        const U29 ex = which ? e.xb : e.x;
        const U29 ey = neg ? u29_sub(zero, e.y, Q29::K4) : e.y;      // table y < 2 p; the negation may become the running point's y (<= 4 p)
        if (empty) {
          acc.x = ex; acc.y = ey; acc.z = u29_one();
          empty = false;
        } else {
          int special;
          acc = j29_madd(acc, ex, ey, special);
          if (special == 1) acc = j29_dbl(acc);
          if (special == 2) empty = true;
        }
      }
    }
  }
  if (empty || jac_is_inf(p)) return false;
  out.x = acc.x; out.y = acc.y; out.z = u29_mul(acc.z, zfix);
  return true;
}
KDEV bool jac_scalar_mul_uniform_u29_j(const Jac<Fq>& p, const Fr& k_mont, unsigned char* dig, J29& out) {
  J29A T[8];
  return jac_scalar_mul_uniform_u29_t(p, k_mont, dig, out, T);
}
KDEV Jac<Fq> jac_scalar_mul_uniform_u29(const Jac<Fq>& p, const Fr& k_mont, unsigned char* dig) {
  J29 a;
  if (!jac_scalar_mul_uniform_u29_j(p, k_mont, dig, a)) return jac_inf<Fq>();
  return {u29_to_fq(a.x), u29_to_fq(a.y), u29_to_fq(a.z)};
}

// ---- two-term products kA A + kB B with ONE doubling chain (the radix-4 butterflies of the FK23 stages with one twiddle per wave) ----------
// Straus: the odd-multiple tables of A and of B on ONE working curve, one running point, the width-5 NAF digits of the four GLV half-scalars
// in LDS (4 x UNIFORM_DIG_STRIDE bytes per product). A's table is built first (-> the curve isomorphic to E by zA), B is moved there as
// (X zA^2, Y zA^3, Z) and its table built on that curve (-> isomorphic by zA zB'), A's entries are rescaled by zB' (2M + the product by beta
// each). Two products over the same pair of points (the radix-4 butterfly needs w2 e2 + w2 w1 e3 and w2' e2 - w2' w1 e3) share the tables.
// models/model_jac29.py: tables_pair and the two-term ladders.
struct J29PairTables {
  J29A* TA;                 // 8 entries each (caller's storage)
  J29A* TB;
  U29 zfix;
};
// A, B: saturated Jacobian, neither the identity
KDEV void j29_build_pair(const Jac<Fq>& A, const Jac<Fq>& B, J29PairTables& t) {
  const U29 beta = u29_const(GlvParams::BETA29);
  J29ArrTable sa = {t.TA, beta}, sb = {t.TB, beta};
  const U29 zA = j29_build_table<true>(A, sa);
  const U29 zA2 = u29_sqr(zA), zA3 = u29_mul(zA, zA2);
  J29 bc;
  bc.x = u29_mul(u29_from_fq(B.x), zA2); bc.y = u29_mul(u29_from_fq(B.y), zA3); bc.z = u29_from_fq(B.z);
  const U29 zB = j29_build_table_j<true>(bc, sb);
  const U29 zB2 = u29_sqr(zB), zB3 = u29_mul(zB, zB2);
#pragma unroll 1
  for (int i = 0; i < 8; i++) {
    t.TA[i].x = u29_mul(t.TA[i].x, zB2);
    t.TA[i].xb = u29_mul(t.TA[i].xb, zB2);
    t.TA[i].y = u29_mul(t.TA[i].y, zB3);
  }
  t.zfix = u29_mul(zA, zB);
}
// kA A + kB B (both scalars wave-uniform; negB: the second term is subtracted). dig: 4 * UNIFORM_DIG_STRIDE bytes of this wave's LDS.
KDEV bool j29_mul2_uniform(const J29PairTables& t, const Fr& kA, const Fr& kB, bool negB, unsigned char* dig, J29& out) {
  bool neg[4];
  glv_uniform_digits(kA, dig, neg[0], neg[1]);
  glv_uniform_digits(kB, dig + 2 * UNIFORM_DIG_STRIDE, neg[2], neg[3]);
  if (negB) { neg[2] = !neg[2]; neg[3] = !neg[3]; }
  U29 zero;
#pragma unroll
  for (int i = 0; i < 9; i++) zero.l[i] = 0;
  J29 acc;
  acc.x = zero; acc.y = zero; acc.z = zero;
  bool empty = true;
#pragma unroll 1
  for (int pos = 128; pos >= 0; pos--) {
    if (!empty) acc = j29_dbl(acc);
#pragma unroll 1
    for (int which = 0; which < 4; which++) {
      const u32 byte = (u32)__builtin_amdgcn_readfirstlane((int)dig[which * UNIFORM_DIG_STRIDE + pos]);
      if (byte) {
        const bool ng = ((byte >> 7) != 0) != neg[which];
        const J29A& e = (which & 2) ? t.TB[(byte & 0x7Fu) - 1u] : t.TA[(byte & 0x7Fu) - 1u];
        const U29 ex = (which & 1) ? e.xb : e.x;
        const U29 ey = ng ? u29_sub(zero, e.y, Q29::K4) : e.y;
        if (empty) {
          acc.x = ex; acc.y = ey; acc.z = u29_one();
          empty = false;
        } else {
          int special;
          acc = j29_madd(acc, ex, ey, special);
          if (special == 1) acc = j29_dbl(acc);
          if (special == 2) empty = true;
        }
      }
    }
  }
  if (empty) return false;
  out.x = acc.x; out.y = acc.y; out.z = u29_mul(acc.z, t.zfix);
  return true;
}

}  // namespace bn254
