// Short-Weierstrass a = 0 group arithmetic for BN254 G1 (over Fq) and G2 (over Fq2), written once
// and instantiated for both fields.
//
// Replaces ark-ec 0.4.2 `short_weierstrass::{Affine, Projective}` arithmetic reached from keaki at
// src/kzg.rs:98 (MSM bucket adds) and src/kem.rs:22,30,36,37 (`.mul`). Coordinates systems are
// chosen for the GPU, not copied from arkworks: buckets use XYZZ (8M+2S mixed add, no inversion),
// scalar-mult ladders use Jacobian (2M+5S doubling). All results are compared at the affine level,
// where they are representation-independent.
#pragma once
#include "bn254_field.hip.h"

namespace bn254 {

// Affine point; the identity is encoded as (0,0) (not on either curve since b != 0).
template <class F>
struct Aff {
  F x, y;
};
template <class F>
struct Jac {  // (X/Z^2, Y/Z^3); identity: Z = 0
  F x, y, z;
};
template <class F>
struct Xyzz {  // (X/ZZ, Y/ZZZ) with ZZ^3 = ZZZ^2; identity: ZZ = 0
  F x, y, zz, zzz;
};
using G1Aff = Aff<Fq>;
using G2Aff = Aff<Fq2>;
using G1Jac = Jac<Fq>;
using G2Jac = Jac<Fq2>;
using G1Xyzz = Xyzz<Fq>;
using G2Xyzz = Xyzz<Fq2>;

template <class F> KDEV bool aff_is_inf(const Aff<F>& p) { return f_is_zero(p.x) && f_is_zero(p.y); }
template <class F> KDEV Aff<F> aff_inf() { return {f_zero<F>(), f_zero<F>()}; }
template <class F> KDEV Aff<F> aff_cneg(const Aff<F>& p, bool neg) { return {p.x, f_cneg(p.y, neg)}; }

// ------------------------------------------------------------------------------------ XYZZ
template <class F> KDEV Xyzz<F> xyzz_inf() { return {f_one<F>(), f_one<F>(), f_zero<F>(), f_zero<F>()}; }
template <class F> KDEV bool xyzz_is_inf(const Xyzz<F>& p) { return f_is_zero(p.zz); }
template <class F> KDEV Xyzz<F> xyzz_from_aff(const Aff<F>& p) {
  if (aff_is_inf(p)) return xyzz_inf<F>();
  return {p.x, p.y, f_one<F>(), f_one<F>()};
}
// 2 * (affine) -> XYZZ   (EFD mdbl-2008-s-1)
template <class F> KDEV Xyzz<F> xyzz_dbl_aff(const Aff<F>& p) {
  F u = f_dbl(p.y);
  F v = f_sqr(u);
  F w = u * v;
  F s = p.x * v;
  F xx = f_sqr(p.x);
  F m = f_dbl(xx) + xx;
  F x3 = f_sqr(m) - f_dbl(s);
  F y3 = m * (s - x3) - w * p.y;
  return {x3, y3, v, w};
}
// EFD dbl-2008-s-1
template <class F> KDEV Xyzz<F> xyzz_dbl(const Xyzz<F>& p) {
  if (xyzz_is_inf(p)) return p;
  F u = f_dbl(p.y);
  F v = f_sqr(u);
  F w = u * v;
  F s = p.x * v;
  F xx = f_sqr(p.x);
  F m = f_dbl(xx) + xx;
  F x3 = f_sqr(m) - f_dbl(s);
  F y3 = m * (s - x3) - w * p.y;
  return {x3, y3, v * p.zz, w * p.zzz};
}
// acc + affine  (EFD madd-2008-s), all special cases handled
template <class F> KDEV Xyzz<F> xyzz_add_mixed(const Xyzz<F>& a, const Aff<F>& q) {
  if (aff_is_inf(q)) return a;
  if (xyzz_is_inf(a)) return {q.x, q.y, f_one<F>(), f_one<F>()};
  F u2 = q.x * a.zz;
  F s2 = q.y * a.zzz;
  F p = u2 - a.x;
  F r = s2 - a.y;
  if (f_is_zero(p)) {
    if (f_is_zero(r)) return xyzz_dbl_aff(q);
    return xyzz_inf<F>();
  }
  F pp = f_sqr(p);
  F ppp = p * pp;
  F qq = a.x * pp;
  F x3 = f_sqr(r) - ppp - f_dbl(qq);
  F y3 = r * (qq - x3) - a.y * ppp;
  return {x3, y3, a.zz * pp, a.zzz * ppp};
}
// EFD add-2008-s
template <class F> KDEV Xyzz<F> xyzz_add(const Xyzz<F>& a, const Xyzz<F>& b) {
  if (xyzz_is_inf(a)) return b;
  if (xyzz_is_inf(b)) return a;
  F u1 = a.x * b.zz;
  F u2 = b.x * a.zz;
  F s1 = a.y * b.zzz;
  F s2 = b.y * a.zzz;
  F p = u2 - u1;
  F r = s2 - s1;
  if (f_is_zero(p)) {
    if (f_is_zero(r)) return xyzz_dbl(a);
    return xyzz_inf<F>();
  }
  F pp = f_sqr(p);
  F ppp = p * pp;
  F qq = u1 * pp;
  F x3 = f_sqr(r) - ppp - f_dbl(qq);
  F y3 = r * (qq - x3) - s1 * ppp;
  return {x3, y3, a.zz * b.zz * pp, a.zzz * b.zzz * ppp};
}
template <class F> KDEV Xyzz<F> xyzz_neg(const Xyzz<F>& a) { return {a.x, -a.y, a.zz, a.zzz}; }
// one inversion
template <class F> KDEV Aff<F> xyzz_to_aff(const Xyzz<F>& p) {
  if (xyzz_is_inf(p)) return aff_inf<F>();
  // 1/zzz, then 1/zz = zzz^-1 * (zzz/zz) ... use: zz^-1 = (zzz^-1)^2 * zz^2 (since zz^3 = zzz^2)
  F izzz = f_inv(p.zzz);
  F izz = f_sqr(izzz) * f_sqr(p.zz);
  return {p.x * izz, p.y * izzz};
}

// ------------------------------------------------------------------------------------ Jacobian
template <class F> KDEV Jac<F> jac_inf() { return {f_one<F>(), f_one<F>(), f_zero<F>()}; }
template <class F> KDEV bool jac_is_inf(const Jac<F>& p) { return f_is_zero(p.z); }
template <class F> KDEV Jac<F> jac_from_aff(const Aff<F>& p) {
  if (aff_is_inf(p)) return jac_inf<F>();
  return {p.x, p.y, f_one<F>()};
}
// EFD dbl-2009-l
template <class F> KDEV Jac<F> jac_dbl(const Jac<F>& p) {
  if (jac_is_inf(p)) return p;
  F a = f_sqr(p.x);
  F b = f_sqr(p.y);
  F c = f_sqr(b);
  F d = f_dbl(f_sqr(p.x + b) - a - c);
  F e = f_dbl(a) + a;
  F f = f_sqr(e);
  F x3 = f - f_dbl(d);
  F y3 = e * (d - x3) - f_dbl(f_dbl(f_dbl(c)));
  F z3 = f_dbl(p.y * p.z);
  return {x3, y3, z3};
}
// EFD madd-2007-bl
template <class F> KDEV Jac<F> jac_add_mixed(const Jac<F>& p, const Aff<F>& q) {
  if (aff_is_inf(q)) return p;
  if (jac_is_inf(p)) return {q.x, q.y, f_one<F>()};
  F z1z1 = f_sqr(p.z);
  F u2 = q.x * z1z1;
  F s2 = q.y * p.z * z1z1;
  F h = u2 - p.x;
  F rr = s2 - p.y;
  if (f_is_zero(h)) {
    if (f_is_zero(rr)) return jac_dbl<F>({q.x, q.y, f_one<F>()});
    return jac_inf<F>();
  }
  rr = f_dbl(rr);
  F hh = f_sqr(h);
  F i = f_dbl(f_dbl(hh));
  F j = h * i;
  F v = p.x * i;
  F x3 = f_sqr(rr) - j - f_dbl(v);
  F y3 = rr * (v - x3) - f_dbl(p.y * j);
  F z3 = f_sqr(p.z + h) - z1z1 - hh;
  return {x3, y3, z3};
}
// EFD add-2007-bl
template <class F> KDEV Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
  if (jac_is_inf(p)) return q;
  if (jac_is_inf(q)) return p;
  F z1z1 = f_sqr(p.z), z2z2 = f_sqr(q.z);
  F u1 = p.x * z2z2, u2 = q.x * z1z1;
  F s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
  F h = u2 - u1;
  F rr = s2 - s1;
  if (f_is_zero(h)) {
    if (f_is_zero(rr)) return jac_dbl(p);
    return jac_inf<F>();
  }
  rr = f_dbl(rr);
  F i = f_sqr(f_dbl(h));
  F j = h * i;
  F v = u1 * i;
  F x3 = f_sqr(rr) - j - f_dbl(v);
  F y3 = rr * (v - x3) - f_dbl(s1 * j);
  F z3 = (f_sqr(p.z + q.z) - z1z1 - z2z2) * h;
  return {x3, y3, z3};
}
template <class F> KDEV Aff<F> jac_to_aff(const Jac<F>& p) {
  if (jac_is_inf(p)) return aff_inf<F>();
  F zi = f_inv(p.z);
  F zi2 = f_sqr(zi);
  return {p.x * zi2, p.y * zi2 * zi};
}
template <class F> KDEV Jac<F> xyzz_to_jac(const Xyzz<F>& p) {  // (X*ZZ, Y*ZZZ, ZZ)
  if (xyzz_is_inf(p)) return jac_inf<F>();
  return {p.x * p.zz, p.y * p.zzz, p.zz};
}

}  // namespace bn254
