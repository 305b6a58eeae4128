// XYZZ points of BN254 G1 in the 9 x 29-bit lazy representation of fq29.hip.h: the working form of the MSM tail (k_msm_reduce,
// k_msm_partial_groups, k_msm_window_finish) for G1. Every coordinate is carried (limbs 0..7 below 2^29 + 8) and below 32 p, so a
// saturated coordinate enters by u29_from_sat_shift5 alone; formulas are EFD add-2008-s and dbl-2008-s-1 (a = 0), the same ones
// bn254_curve.hip.h uses in the saturated arithmetic. Bounds (multiples of p), inputs < 32:
//   add: U1, U2, S1, S2 < 7.1; P, R = difference + 8p < 15.1; PP < 2.4; PPP, Q < 1.3; X3 < 10.4; T < 17.1 (uncarried, only multiplied by
//        the carried R); Y3 < 4.6; ZZ3, ZZZ3 < 1.1
//   dbl: U = 2Y < 64; V < 25.2; W < 10.5; S < 5.8; M = 3X^2 < 21.2; X3 < 19.7; T < 37.8 (uncarried); Y3 < 9.7; ZZ3 < 5.8; ZZZ3 < 3
// (a limb-exact model with 64-bit overflow assertions ran these chains before the kernels were written).
#pragma once
#include "bn254_curve.hip.h"
#include "fq29.hip.h"

namespace bn254 {

struct X29 {
  U29 x, y, zz, zzz;
  u32 inf;
};
KDEV X29 x29_inf() {
  X29 r;
  r.x = r.y = r.zz = r.zzz = u29_one();
  r.inf = 1;
  return r;
}
KDEV X29 x29_load(const Xyzz<Fq>& p) {
  X29 r;
  r.x = u29_from_sat_shift5(p.x.l); r.y = u29_from_sat_shift5(p.y.l);
  r.zz = u29_from_sat_shift5(p.zz.l); r.zzz = u29_from_sat_shift5(p.zzz.l);
  r.inf = fq_is_zero(p.zz) ? 1u : 0u;
  return r;
}
KDEV Xyzz<Fq> x29_store(const X29& p) {
  if (p.inf) return xyzz_inf<Fq>();
  return {u29_to_fq(p.x), u29_to_fq(p.y), u29_to_fq(p.zz), u29_to_fq(p.zzz)};
}
KDEV X29 x29_dbl(const X29& a) {
  if (a.inf) return a;
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = a.y.l[i] << 1;
  const U29 U = u29_carry(t);
  const U29 V = u29_sqr(U), W = u29_mul(U, V), S = u29_mul(a.x, V), XX = u29_sqr(a.x);
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = 3u * XX.l[i];
  const U29 M = u29_carry(t);
  const U29 M2 = u29_sqr(M);
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = M2.l[i] - 2u * S.l[i] + Q29::K16W[i];
  X29 r;
  r.x = u29_carry(t);
  const U29 T = u29_sub_raw(S, r.x, Q29::K32);
  r.y = u29_sub(u29_mul(M, T), u29_mul(W, a.y), Q29::K4);
  r.zz = u29_mul(V, a.zz);
  r.zzz = u29_mul(W, a.zzz);
  r.inf = 0;
  return r;
}
KDEV X29 x29_add(const X29& a, const X29& b) {
  if (a.inf) return b;
  if (b.inf) return a;
  const U29 U1 = u29_mul(a.x, b.zz), U2 = u29_mul(b.x, a.zz), S1 = u29_mul(a.y, b.zzz), S2 = u29_mul(b.y, a.zzz);
  const U29 P = u29_sub(U2, U1, Q29::K8), R = u29_sub(S2, S1, Q29::K8);
  if (u29_maybe_zero(P)) {
    if (u29_is_zero(P)) {
      if (u29_is_zero(R)) return x29_dbl(a);
      return x29_inf();
    }
  }
  const U29 PP = u29_sqr(P), PPP = u29_mul(P, PP), Q = u29_mul(U1, PP);
  X29 r;
  r.x = u29_sub3(u29_sqr(R), PPP, Q);
  const U29 T = u29_sub_raw(Q, r.x, Q29::K16);
  r.y = u29_sub(u29_mul(R, T), u29_mul(S1, PPP), Q29::K2);
  r.zz = u29_mul(u29_mul(a.zz, b.zz), PP);
  r.zzz = u29_mul(u29_mul(a.zzz, b.zzz), PPP);
  r.inf = 0;
  return r;
}

// the working point of the MSM tail: saturated XYZZ in general, the 29-bit form for G1
template <class F>
struct TailOps {
  typedef Xyzz<F> P;
  static KDEV P inf() { return xyzz_inf<F>(); }
  static KDEV P load(const Xyzz<F>& p) { return p; }
  static KDEV Xyzz<F> store(const P& p) { return p; }
  static KDEV P add(const P& a, const P& b) { return xyzz_add(a, b); }
  static KDEV P dbl(const P& a) { return xyzz_dbl(a); }
};
template <>
struct TailOps<Fq> {
  typedef X29 P;
  static KDEV P inf() { return x29_inf(); }
  static KDEV P load(const Xyzz<Fq>& p) { return x29_load(p); }
  static KDEV Xyzz<Fq> store(const P& p) { return x29_store(p); }
  static KDEV P add(const P& a, const P& b) { return x29_add(a, b); }
  static KDEV P dbl(const P& a) { return x29_dbl(a); }
};

}  // namespace bn254
