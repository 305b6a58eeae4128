// On-device self-tests of the 29-bit lazy arithmetic (fq29 / xyzz29 / jac29) against the saturated streams. Split from selftest.hip
// so that the two translation units compile in parallel.
#include "bn254_curve.hip.h"
#include "fq29.hip.h"
#include "xyzz29.hip.h"
#include "jac29.hip.h"
#include "ec_batch.hip.h"
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

// 9 x 29-bit lazy arithmetic (fq29.hip.h) against the saturated streams: every result is brought back with u29_to_fq
__global__ void __launch_bounds__(256) k_selftest_u29(u32 seed, u32 iters, unsigned long long* mismatches) {
  u32 s = (seed ^ (blockIdx.x * 0x9E3779B9u) ^ (threadIdx.x * 0xC2B2AE35u)) | 1u;
  unsigned long long bad = 0;
  for (u32 it = 0; it < iters; it++) {
    u32 sh = mix(s);
    Fq a = pick(s, (sh & 15) < 6 ? (sh & 15) : 0), b = pick(s, ((sh >> 4) & 15) < 6 ? ((sh >> 4) & 15) : 0), c = pick(s, 0), d = pick(s, 0);
    // entry by the free 5-bit shift (value < 32p) and by the reducing entry (< 2p)
    U29 al = u29_from_sat_shift5(a.l);
    U29 ar = u29_from_fq(a), br = u29_from_fq(b), cr = u29_from_fq(c), dr = u29_from_fq(d);
    bad += !fq_eq(u29_to_fq(ar), a);
    bad += !fq_eq(u29_to_fq(al), a);
    bad += !fq_eq(u29_to_fq(u29_mul(al, br)), a * b);           // table coordinate x accumulator coordinate
    bad += !fq_eq(u29_to_fq(u29_mul(ar, br)), a * b);
    bad += !fq_eq(u29_to_fq(u29_sqr(ar)), fq_sqr(a));
    {  // the asm streams against the portable statements of the same column algorithm, limb for limb
      U29 m1 = u29_mul(al, br), m2 = u29_mul_ref(al, br), s1 = u29_sqr(ar), s2 = u29_sqr_ref(ar);
      for (int j = 0; j < 9; j++) bad += (m1.l[j] != m2.l[j]) + (s1.l[j] != s2.l[j]);
    }
    // differences at every bias, then used as product operands (the shapes of the mixed addition)
    U29 p16 = u29_sub(ar, br, Q29::K16), p4 = u29_sub(ar, br, Q29::K4), p2 = u29_sub(ar, br, Q29::K2);
    bad += !fq_eq(u29_to_fq(p16), a - b);
    bad += !fq_eq(u29_to_fq(p4), a - b);
    bad += !fq_eq(u29_to_fq(p2), a - b);
    bad += !fq_eq(u29_to_fq(u29_sqr(p16)), fq_sqr(a - b));
    bad += !fq_eq(u29_to_fq(u29_mul(p16, u29_sqr(p16))), (a - b) * fq_sqr(a - b));
    U29 x3 = u29_sub3(ar, br, cr);
    bad += !fq_eq(u29_to_fq(x3), a - b - c - c);
    U29 t = u29_sub_raw(dr, x3, Q29::K16);
    bad += !fq_eq(u29_to_fq(u29_mul(p4, t)), (a - b) * (d - (a - b - c - c)));
    {  // uncarried difference as subtrahend (bias 2^31), as a factor, and on the way out
      U29 y = u29_sub_raw(u29_mul(ar, br), u29_mul(cr, dr), Q29::K2);
      bad += !fq_eq(u29_to_fq(y), a * b - c * d);
      bad += !fq_eq(u29_to_fq(u29_sub(ar, y, Q29::K4W)), a - (a * b - c * d));
      bad += !fq_eq(u29_to_fq(u29_mul(y, br)), (a * b - c * d) * b);
    }
    {  // the dual product of the mixed addition: (a b + (2p - c) d) / R with one reduction == a b - c d
      U29 nc;
      for (int j = 0; j < 9; j++) nc.l[j] = Q29::K2[j] - cr.l[j];
      bad += !fq_eq(u29_to_fq(u29_mul2(p4, u29_sub(dr, x3, Q29::K16), nc, br)), (a - b) * (d - (a - b - c - c)) - c * b);
    }
    // zero filter and exact zero test
    U29 z = u29_sub(ar, ar, Q29::K16);
    bad += !u29_maybe_zero(z);
    bad += !u29_is_zero(z);
    bad += u29_is_zero(p16) != fq_eq(a, b);
    bad += (fq_eq(a, b) && !u29_maybe_zero(p16));
    bad += u29_is_zero(ar) != fq_is_zero(a);
  }
  if (bad) atomicAdd(mismatches, bad);
}

// XYZZ addition / doubling in the 29-bit representation (xyzz29.hip.h) against the saturated formulas, projectively compared
KDEV bool xyzz_same(const Xyzz<Fq>& a, const Xyzz<Fq>& b) {
  if (xyzz_is_inf(a) || xyzz_is_inf(b)) return xyzz_is_inf(a) && xyzz_is_inf(b);
  return fq_eq(a.x * b.zz, b.x * a.zz) && fq_eq(a.y * b.zzz, b.y * a.zzz);
}
__global__ void __launch_bounds__(64) k_selftest_x29(u32 seed, u32 iters, unsigned long long* mismatches) {
  u32 s = (seed ^ ((blockIdx.x * 64 + threadIdx.x) * 0x9E3779B9u)) | 1u;
  unsigned long long bad = 0;
  const Aff<Fq> g = {G1_GEN_X, G1_GEN_Y};
  Xyzz<Fq> p = xyzz_from_aff(g), q = xyzz_dbl_aff(g);
  for (u32 it = 0; it < iters; it++) {
    u32 r = mix(s);
    // walk two pseudo-random multiples of the generator with the saturated arithmetic
    p = (r & 1) ? xyzz_add(xyzz_dbl(p), q) : xyzz_add_mixed(xyzz_dbl(p), g);
    q = (r & 2) ? xyzz_add(q, p) : xyzz_dbl(q);
    const X29 pl = x29_load(p), ql = x29_load(q);
    bad += !xyzz_same(x29_store(x29_add(pl, ql)), xyzz_add(p, q));
    bad += !xyzz_same(x29_store(x29_dbl(pl)), xyzz_dbl(p));
    bad += !xyzz_same(x29_store(x29_add(pl, pl)), xyzz_dbl(p));                            // equal inputs -> doubling branch
    bad += !xyzz_is_inf(x29_store(x29_add(pl, x29_load(xyzz_neg(p)))));                    // opposite inputs -> infinity
    bad += !xyzz_same(x29_store(x29_add(x29_inf(), ql)), q);
    bad += !xyzz_same(x29_store(x29_add(ql, x29_load(xyzz_inf<Fq>()))), q);
    // chains stay inside the working form: (p + q) + 2p + q
    bad += !xyzz_same(x29_store(x29_add(x29_add(x29_add(pl, ql), x29_dbl(pl)), ql)), xyzz_add(xyzz_add(xyzz_add(p, q), xyzz_dbl(p)), q));
    bad += !xyzz_same(x29_store(x29_dbl(x29_dbl(x29_dbl(ql)))), xyzz_dbl(xyzz_dbl(xyzz_dbl(q))));
    // stored coordinates are canonical
    Xyzz<Fq> st = x29_store(x29_add(pl, ql));
    u32 t[8];
    for (int j = 0; j < 8; j++) t[j] = st.x.l[j];
    fp_reduce_once<FqParamsRef>(t);
    for (int j = 0; j < 8; j++) bad += (t[j] != st.x.l[j]);
  }
  if (bad) atomicAdd(mismatches, bad);
}

// NAF ladder in the 29-bit Jacobian arithmetic (jac29.hip.h) against the saturated double-and-add of ec_batch.hip.h
__global__ void __launch_bounds__(64) k_selftest_j29(u32 seed, u32 iters, unsigned long long* mismatches) {
  u32 s = (seed ^ ((blockIdx.x * 64 + threadIdx.x) * 0x9E3779B9u)) | 1u;
  unsigned long long bad = 0;
  const Aff<Fq> g = {G1_GEN_X, G1_GEN_Y};
  Jac<Fq> p = jac_from_aff(g);
  for (u32 it = 0; it < iters; it++) {
    Fr k;
    for (int j = 0; j < 8; j++) k.l[j] = mix(s);
    k.l[7] &= 0x1FFFFFFFu;                              // < 2^253 < r: a valid Montgomery residue of some scalar
    u32 sh = mix(s) & 7;
    if (sh == 0) { for (int j = 0; j < 8; j++) k.l[j] = 0; }                       // 0
    if (sh == 1) { for (int j = 0; j < 8; j++) k.l[j] = FrParams::ONE[j]; }        // 1
    if (sh == 2) { Fr one = fp_one<FrParams>(); k = fp_neg<FrParams>(one); }                       // r - 1
    if (sh == 3) { Fr one = fp_one<FrParams>(); k = fp_neg<FrParams>(fp_add<FrParams>(one, one)); }  // r - 2: the ladder's last addition is a doubling
    Jac<Fq> a = jac_scalar_mul_u29(p, k);
    Aff<Fq> ref = jac_to_aff(scalar_mul_sat(jac_to_aff(p), k));
    Aff<Fq> got = jac_to_aff(a);
    bad += !(fq_eq(got.x, ref.x) && fq_eq(got.y, ref.y));
    bad += !jac_is_inf(jac_scalar_mul_u29(jac_inf<Fq>(), k));
    // the butterflies' shared add / subtract: the ladder's running point against the base, against the saturated additions
    J29 aj, sj, dj;
    if (jac_scalar_mul_u29_j(p, k, aj)) {
      const bool ok = j29_addsub(j29_from_sat(p), aj, sj, dj);
      Jac<Fq> na = a;
      na.y = -na.y;
      const Jac<Fq> es = jac_add(p, a), ed = jac_add(p, na);
      if (ok) {
        const Aff<Fq> gs = jac_to_aff(j29_to_sat(sj)), gd = jac_to_aff(j29_to_sat(dj)), rs = jac_to_aff(es), rd = jac_to_aff(ed);
        bad += !(fq_eq(gs.x, rs.x) && fq_eq(gs.y, rs.y) && fq_eq(gd.x, rd.x) && fq_eq(gd.y, rd.y));
      } else {
        bad += !(jac_is_inf(es) || jac_is_inf(ed));       // refused only for u = +-v (k = 1 and k = r - 1 above)
      }
      // the same point in two Jacobian representations (a + p and p + a: Z differs by the sign of H), and its negative: both refused
      if (!jac_is_inf(es)) {
        const Jac<Fq> q1 = es, q2 = jac_add(a, p);
        Jac<Fq> nq2 = q2;
        nq2.y = -nq2.y;
        bad += j29_addsub(j29_from_sat(q1), j29_from_sat(q2), sj, dj) ? 1u : 0u;
        bad += j29_addsub(j29_from_sat(q1), j29_from_sat(nq2), sj, dj) ? 1u : 0u;
      }
    }
    p = jac_is_inf(a) ? jac_dbl(p) : jac_add(a, p);      // next base: some other multiple, non-trivial Z
  }
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace bn254
namespace keaki_internal {
keaki_status selftest_u29_run(keaki_hip_ctx* ctx, uint32_t blocks, uint32_t iters, uint32_t seed, void* d_mismatches) {
  hipLaunchKernelGGL(bn254::k_selftest_j29, dim3(blocks > 16 ? 16 : blocks), dim3(64), 0, ctx->stream, seed, iters > 4 ? 4u : iters, (unsigned long long*)d_mismatches);
  hipLaunchKernelGGL(bn254::k_selftest_x29, dim3(blocks > 64 ? 64 : blocks), dim3(64), 0, ctx->stream, seed, iters > 16 ? 16u : iters, (unsigned long long*)d_mismatches);
  hipLaunchKernelGGL(bn254::k_selftest_u29, dim3(blocks), dim3(256), 0, ctx->stream, seed, iters, (unsigned long long*)d_mismatches);
  return launch_check(ctx, "selftest_u29");
}
}  // namespace keaki_internal
