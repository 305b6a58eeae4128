// XYZZ accumulation on BN254's G2 (coordinates in Fq2) in the 9 x 29-bit lazy limbs of fq29.hip.h, one lane per point: the mixed addition
// of the G2 bucket kernel (k_msm_accumulate_g2_u29, msm.hip.h) and of the fixed-base sums of `encapsulate` (k_encap_fixed<Fq2>,
// ec_batch.hip.h: ct = r [tau]_2 - (r alpha) g2, reference src/kem.rs:36-37).
//
// An Fq2 value is two limb vectors (re, im), Montgomery radix 2^261, lazily reduced. A product component is ONE dual stream
//   re = a0 b0 + (K p - a1) b1        im = a0 b1 + a1 b0          (u29_mul2: one reduction for two products)
// a square is one product per component, (a0 + a1)(a0 - a1 + K p) and (2 a0) a1. Because every component of a product is a DUAL product
// the value bounds grow twice as fast as in the G1 kernel; the chain is kept stable by bringing X3 back below 2p with one product by
// `one` per component (Y3, ZZ3, ZZZ3 are product outputs). keaki_amd/csrc/models/model_g2_add29.py runs the same operations in the same
// order on Python integers with assertions on every limb and bound (400 chains of six additions, worst-case coordinates included)
// against plain Fq2 arithmetic; tests/test_pair261_model.py keeps it in the CPU suite. Bounds (multiples of p), accumulator < 4 (X < 2),
// table coordinates < 32 (entered by the free 5-bit shift): U2, S2 < 3.3; P < 5.3; R < 7.3; PP < 2; PPP < 1.2; Q < 1.1; RR < 2.4;
// X3 before the reduction < 6.4, after < 1.1; T < 3.1; Y3 < 1.5; ZZ3, ZZZ3 < 1.1.
// ~5,600 instructions per addition (4,700 v_mad_u64_u32) against ~9,000 for the generic saturated formulas over the Fq2 product.
#pragma once
#include "bn254_curve.hip.h"
#include "fq29.hip.h"
#include "fq29_dot_asm.hip.h"
#include "xyzz29.hip.h"

namespace bn254 {

struct L2 { U29 a, b; };          // re, im: limbs carried (<= 2^29 + 8)

KDEV L2 l2_mul(const L2& x, const L2& y, const u32 (&KX)[9]) {      // KX >= bound of x.b, bias 2^30 (the c operand of the stream stays uncarried)
  U29 nb;
#pragma unroll
  for (int i = 0; i < 9; i++) nb.l[i] = KX[i] - x.b.l[i];
  return {u29_mul2(x.a, y.a, nb, y.b), u29_mul2(x.a, y.b, x.b, y.a)};
}
KDEV L2 l2_sqr(const L2& x, const u32 (&KX)[9]) {
  U29 s, d, t;
#pragma unroll
  for (int i = 0; i < 9; i++) { s.l[i] = x.a.l[i] + x.b.l[i]; d.l[i] = x.a.l[i] - x.b.l[i] + KX[i]; t.l[i] = 2u * x.a.l[i]; }
  return {u29_mul(s, u29_carry(d)), u29_mul(t, x.b)};
}
KDEV L2 l2_sub(const L2& x, const L2& y, const u32 (&K)[9]) { return {u29_sub(x.a, y.a, K), u29_sub(x.b, y.b, K)}; }
KDEV U29 u29_neg_carried(const U29& b, const u32 (&K)[9]) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = K[i] - b.l[i];
  return u29_carry(t);
}
KDEV L2 l2_from_table(const Fq2& x) { return {u29_from_sat_shift5(x.c0.l), u29_from_sat_shift5(x.c1.l)}; }             // < 32p, exact limbs
KDEV L2 l2_from_fq2(const Fq2& x) { return {u29_from_fq(x.c0), u29_from_fq(x.c1)}; }                                   // reduced: < 2p
KDEV Fq2 l2_to_fq2(const L2& x) { return {u29_to_fq(x.a), u29_to_fq(x.b)}; }
KDEV L2 l2_one() { L2 r; r.a = u29_one(); r.b = u29_one(); for (int i = 0; i < 9; i++) r.b.l[i] = 0; return r; }

struct X29G2 {
  L2 x, y, zz, zzz;
  bool empty;
};
KDEV X29G2 x29g2_inf() { X29G2 r; r.x = r.y = r.zz = r.zzz = l2_one(); r.empty = true; return r; }
KDEV X29G2 x29g2_load(const Xyzz<Fq2>& p) {
  X29G2 r;
  r.empty = xyzz_is_inf(p);
  r.x = l2_from_fq2(p.x); r.y = l2_from_fq2(p.y); r.zz = l2_from_fq2(p.zz); r.zzz = l2_from_fq2(p.zzz);
  return r;
}
KDEV Xyzz<Fq2> x29g2_store(const X29G2& p) {
  if (p.empty) return xyzz_inf<Fq2>();
  return {l2_to_fq2(p.x), l2_to_fq2(p.y), l2_to_fq2(p.zz), l2_to_fq2(p.zzz)};
}
struct X29G2;
KDEV X29G2 x29g2_dbl_of_acc(const X29G2& acc);
// acc += q (affine, saturated 2^256 form as the tables hold it; q.y already negated by the caller for a negative digit). Identity in
// either slot, equal points (doubling) and opposite points are handled. The doubling (18 in 2^29 additions on unrelated points) doubles the
// ACCUMULATOR in the lazy limbs (round 6: until then the affine point went through the saturated formulas, whose out-of-line Fq2 products gave
// the bucket kernel a call frame -- the 80 B/lane of scratch -Rpass reported for it).
KDEV void x29g2_add_mixed(X29G2& acc, const Aff<Fq2>& q) {
  if (aff_is_inf(q)) return;
  const L2 X2 = l2_from_table(q.x), Y2 = l2_from_table(q.y);
  if (acc.empty) {
    const U29 one = u29_one();
    acc.x = {u29_mul(X2.a, one), u29_mul(X2.b, one)};
    acc.y = {u29_mul(Y2.a, one), u29_mul(Y2.b, one)};
    acc.zz = l2_one(); acc.zzz = l2_one();
    acc.empty = false;
    return;
  }
  const L2 U2 = l2_mul(X2, acc.zz, Q29::K64), S2 = l2_mul(Y2, acc.zzz, Q29::K64);
  const L2 P = l2_sub(U2, acc.x, Q29::K2), R = l2_sub(S2, acc.y, Q29::K4);
  if (u29_maybe_zero(P.a) && u29_maybe_zero(P.b)) {        // cheap filter first (limb 0 is exact after the carry pass); exact test only then
    if (u29_is_zero(P.a) && u29_is_zero(P.b)) {
      if (u29_is_zero(R.a) && u29_is_zero(R.b)) acc = x29g2_dbl_of_acc(acc);            // same point: acc == q as a point, so 2 q = 2 acc
      else acc.empty = true;                                                            // opposite points
      return;
    }
  }
  const L2 PP = l2_sqr(P, Q29::K8);
  const L2 PPP = l2_mul(P, PP, Q29::K8), Q = l2_mul(acc.x, PP, Q29::K2);
  const L2 RR = l2_sqr(R, Q29::K8);
  L2 X3;
  {
    U29 ta, tb;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      ta.l[i] = RR.a.l[i] - PPP.a.l[i] - 2u * Q.a.l[i] + Q29::K4W[i];
      tb.l[i] = RR.b.l[i] - PPP.b.l[i] - 2u * Q.b.l[i] + Q29::K4W[i];
    }
    const U29 one = u29_one();
    X3 = {u29_mul(u29_carry(ta), one), u29_mul(u29_carry(tb), one)};      // back below 2p: keeps the chain's bounds where the model put them
  }
  const L2 T = l2_sub(Q, X3, Q29::K2);
  const U29 nR1 = u29_neg_carried(R.b, Q29::K8), nY0 = u29_neg_carried(acc.y.a, Q29::K4), nY1 = u29_neg_carried(acc.y.b, Q29::K4);
  L2 Y3;
  u29_dot4_asm(Y3.a.l, R.a.l, T.a.l, nR1.l, T.b.l, nY0.l, PPP.a.l, acc.y.b.l, PPP.b.l);      // R0 T0 - R1 T1 - Y0 PPP0 + Y1 PPP1
  u29_dot4_asm(Y3.b.l, R.a.l, T.b.l, R.b.l, T.a.l, nY0.l, PPP.b.l, nY1.l, PPP.a.l);          // R0 T1 + R1 T0 - Y0 PPP1 - Y1 PPP0
  acc.zz = l2_mul(acc.zz, PP, Q29::K4);
  acc.zzz = l2_mul(acc.zzz, PPP, Q29::K4);
  acc.x = X3; acc.y = Y3;
}


// ---- the MSM tail on G2 (k_msm_reduce, k_msm_partial_groups, k_msm_window_finish): general addition and doubling of two lazy accumulators.
// Working form: every component a load of a reduced value (l2_from_fq2, < 1.2p) or an output of these functions (< 1.2p: X3 is brought back
// by the product by `one` as in the mixed addition). models/model_g2_add29.py (add_full, dbl_full, run_tail) runs the same operations in
// the same order with assertions on every limb and bound. Bounds: U, S < 1.1; P, R < 3.2; PP, RR < 1.3; T < 3.2; doubling: U = 2Y < 2.4,
// V < 1.2, M = 3 X^2 < 3.4.
KDEV L2 l2_renorm(const U29& ta, const U29& tb) {
  const U29 one = u29_one();
  return {u29_mul(u29_carry(ta), one), u29_mul(u29_carry(tb), one)};
}
KDEV X29G2 x29g2_dbl(const X29G2& a) {
  if (a.empty) return a;
  L2 U;
#pragma unroll
  for (int i = 0; i < 9; i++) { U.a.l[i] = 2u * a.y.a.l[i]; U.b.l[i] = 2u * a.y.b.l[i]; }
  U.a = u29_carry(U.a); U.b = u29_carry(U.b);
  const L2 V = l2_sqr(U, Q29::K8), W = l2_mul(U, V, Q29::K8), S = l2_mul(a.x, V, Q29::K2), XX = l2_sqr(a.x, Q29::K2);
  L2 M;
#pragma unroll
  for (int i = 0; i < 9; i++) { M.a.l[i] = 3u * XX.a.l[i]; M.b.l[i] = 3u * XX.b.l[i]; }
  M.a = u29_carry(M.a); M.b = u29_carry(M.b);
  const L2 MM = l2_sqr(M, Q29::K4);
  X29G2 r;
  {
    U29 ta, tb;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      ta.l[i] = MM.a.l[i] - 2u * S.a.l[i] + Q29::K4W[i];
      tb.l[i] = MM.b.l[i] - 2u * S.b.l[i] + Q29::K4W[i];
    }
    r.x = l2_renorm(ta, tb);
  }
  const L2 T = l2_sub(S, r.x, Q29::K2);
  const U29 nM1 = u29_neg_carried(M.b, Q29::K4), nW0 = u29_neg_carried(W.a, Q29::K2), nW1 = u29_neg_carried(W.b, Q29::K2);
  u29_dot4_asm(r.y.a.l, M.a.l, T.a.l, nM1.l, T.b.l, nW0.l, a.y.a.l, W.b.l, a.y.b.l);        // M0 T0 - M1 T1 - W0 Y0 + W1 Y1
  u29_dot4_asm(r.y.b.l, M.a.l, T.b.l, M.b.l, T.a.l, nW0.l, a.y.b.l, nW1.l, a.y.a.l);        // M0 T1 + M1 T0 - W0 Y1 - W1 Y0
  r.zz = l2_mul(V, a.zz, Q29::K4);
  r.zzz = l2_mul(W, a.zzz, Q29::K2);
  r.empty = false;
  return r;
}
// the mixed addition's accumulator doubled: first brought to the tail's working form (canonical value, reloaded: < 1.2p, the bounds x29g2_dbl is
// modelled for -- the mixed addition's own chain allows its accumulator more)
KDEV X29G2 x29g2_dbl_of_acc(const X29G2& acc) { return x29g2_dbl(x29g2_load(x29g2_store(acc))); }
KDEV X29G2 x29g2_add(const X29G2& a, const X29G2& b) {
  if (a.empty) return b;
  if (b.empty) return a;
  const L2 U1 = l2_mul(a.x, b.zz, Q29::K2), U2 = l2_mul(b.x, a.zz, Q29::K2);
  const L2 S1 = l2_mul(a.y, b.zzz, Q29::K4), S2 = l2_mul(b.y, a.zzz, Q29::K4);
  const L2 P = l2_sub(U2, U1, Q29::K2), R = l2_sub(S2, S1, Q29::K2);
  if (u29_maybe_zero(P.a) && u29_maybe_zero(P.b)) {
    if (u29_is_zero(P.a) && u29_is_zero(P.b)) {
      if (u29_is_zero(R.a) && u29_is_zero(R.b)) return x29g2_dbl(a);
      return x29g2_inf();
    }
  }
  const L2 PP = l2_sqr(P, Q29::K4);
  const L2 PPP = l2_mul(P, PP, Q29::K4), Q = l2_mul(U1, PP, Q29::K2);
  const L2 RR = l2_sqr(R, Q29::K4);
  X29G2 r;
  {
    U29 ta, tb;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      ta.l[i] = RR.a.l[i] - PPP.a.l[i] - 2u * Q.a.l[i] + Q29::K4W[i];
      tb.l[i] = RR.b.l[i] - PPP.b.l[i] - 2u * Q.b.l[i] + Q29::K4W[i];
    }
    r.x = l2_renorm(ta, tb);
  }
  const L2 T = l2_sub(Q, r.x, Q29::K2);
  const U29 nR1 = u29_neg_carried(R.b, Q29::K4), nS0 = u29_neg_carried(S1.a, Q29::K2), nS1 = u29_neg_carried(S1.b, Q29::K2);
  u29_dot4_asm(r.y.a.l, R.a.l, T.a.l, nR1.l, T.b.l, nS0.l, PPP.a.l, S1.b.l, PPP.b.l);       // R0 T0 - R1 T1 - S0 PPP0 + S1 PPP1
  u29_dot4_asm(r.y.b.l, R.a.l, T.b.l, R.b.l, T.a.l, nS0.l, PPP.b.l, nS1.l, PPP.a.l);       // R0 T1 + R1 T0 - S0 PPP1 - S1 PPP0
  r.zz = l2_mul(l2_mul(a.zz, b.zz, Q29::K4), PP, Q29::K2);
  r.zzz = l2_mul(l2_mul(a.zzz, b.zzz, Q29::K4), PPP, Q29::K2);
  r.empty = false;
  return r;
}

// the working point of the MSM tail for G2 (see TailOps in xyzz29.hip.h)
template <>
struct TailOps<Fq2> {
  typedef X29G2 P;
  static KDEV P inf() { return x29g2_inf(); }
  static KDEV P load(const Xyzz<Fq2>& p) { return x29g2_load(p); }
  static KDEV Xyzz<Fq2> store(const P& p) { return x29g2_store(p); }
  static KDEV P add(const P& a, const P& b) { return x29g2_add(a, b); }
  static KDEV P dbl(const P& a) { return x29g2_dbl(a); }
};

}  // namespace bn254
