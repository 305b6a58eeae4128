// Batched BN254 optimal-ate pairing on gfx950: one lane PAIR per pairing.
//
// Replaces `E::pairing(p, q)` (reference src/kem.rs:30,58; src/kzg.rs:148; ark-ec 0.4.2 models/bn:
// G2Prepared line coefficients + multi_miller_loop + final_exponentiation) and
// `serialize_uncompressed` of the GT element (src/kem.rs:32,61).
//
// Tower (ark-bn254): Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - (9+u)), Fq12 = Fq6[w]/(w^2 - v),
// D-type twist. The reduced pairing value is independent of the Miller-loop addition chain and of
// subfield scalings of the line functions, so this kernel is free to (a) compute the lines on the
// fly (or read a table when Q is constant), (b) use the proper NAF of 6z+2 (22 additions instead of
// arkworks' 26) and a NAF of z in the hard part. The final exponent is arkworks' exactly:
// (p^12-1)/r * 2z(6z^2+3z+1).
//
// Lane-pair layout: every Fq2 element a0 + a1 u is split over two adjacent lanes -- the even lane
// holds a0, the odd lane a1 (type Fq2d = "this lane's component"). Consequences:
//   * Fq2 add/sub/double are ONE Fq operation per lane; an Fq12 is 48 VGPRs per lane, not 96, so the
//     Miller loop / final exponentiation live in registers instead of a 10 KB/lane scratch stack;
//   * an Fq2 product is two Fq products per lane (a_self*b_x and a_other*b_y) instead of three
//     sequential Karatsuba products; a squaring is one;
//   * the only cross-lane traffic is the partner's component, fetched with a DPP quad-perm
//     (v_mov_b32_dpp quad_perm:[1,0,3,2]: register-to-register, no LDS);
//   * both lanes run the SAME instruction stream (component-dependent operands are chosen with
//     v_cndmask), so there is no divergence; 2n lanes give twice the waves to hide latency.
// The tower code above Fq2 is written once against the Fq2d primitives.
#pragma once
#include "bn254_curve.cuh"
#include "fq29.cuh"

namespace bn254 {

#define KNOINLINE __device__ __noinline__
#define KTOWER __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// Fq2d: one component of an Fq2 element per lane
// ---------------------------------------------------------------------------------------------
struct Fq2d { Fq v; };

KDEV u32 lane_odd() { return threadIdx.x & 1u; }
// partner lane's value (lane ^ 1). All 64 lanes of the wave must be active.
// gfx950 needs two wait states between a VALU write of a VGPR and a DPP read of it. hipcc pads that for instructions it
// scheduled itself, but the limbs usually come straight out of an inline-asm field stream, which the hazard recogniser does
// not look into -- so the value is first passed through an `s_nop 1` statement that owns all eight limbs.
KDEV Fq fq_partner(const Fq& a) {
  u32 x0 = a.l[0], x1 = a.l[1], x2 = a.l[2], x3 = a.l[3], x4 = a.l[4], x5 = a.l[5], x6 = a.l[6], x7 = a.l[7];
  asm volatile("s_nop 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
  Fq r;
  r.l[0] = (u32)__builtin_amdgcn_update_dpp(0, (int)x0, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
  r.l[1] = (u32)__builtin_amdgcn_update_dpp(0, (int)x1, 0xB1, 0xF, 0xF, true);
  r.l[2] = (u32)__builtin_amdgcn_update_dpp(0, (int)x2, 0xB1, 0xF, 0xF, true);
  r.l[3] = (u32)__builtin_amdgcn_update_dpp(0, (int)x3, 0xB1, 0xF, 0xF, true);
  r.l[4] = (u32)__builtin_amdgcn_update_dpp(0, (int)x4, 0xB1, 0xF, 0xF, true);
  r.l[5] = (u32)__builtin_amdgcn_update_dpp(0, (int)x5, 0xB1, 0xF, 0xF, true);
  r.l[6] = (u32)__builtin_amdgcn_update_dpp(0, (int)x6, 0xB1, 0xF, 0xF, true);
  r.l[7] = (u32)__builtin_amdgcn_update_dpp(0, (int)x7, 0xB1, 0xF, 0xF, true);
  return r;
}
KDEV Fq fq_select(bool c, const Fq& a, const Fq& b) {  // c ? a : b
  Fq r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = c ? a.l[j] : b.l[j];
  return r;
}

KDEV Fq2d operator+(const Fq2d& a, const Fq2d& b) { return {a.v + b.v}; }
KDEV Fq2d operator-(const Fq2d& a, const Fq2d& b) { return {a.v - b.v}; }
KDEV Fq2d operator-(const Fq2d& a) { return {-a.v}; }
KDEV Fq2d fq2_dbl(const Fq2d& a) { return {fq_dbl(a.v)}; }
KDEV Fq2d fq2d_zero() { return {fq_zero()}; }
KDEV Fq2d fq2d_one() { return {fq_select(lane_odd() != 0, fq_zero(), fq_one())}; }
KDEV Fq2d fq2_conj(const Fq2d& a) { return {fp_cneg<FqParams>(a.v, lane_odd() != 0)}; }
KDEV Fq2d fq2_mul_fq(const Fq2d& a, const Fq& k) { return {a.v * k}; }
// this lane's component of an Fq2 stored as (c0, c1)
KDEV Fq2d fq2d_load(const Fq2* a) { return {reinterpret_cast<const Fq*>(a)[lane_odd()]}; }

// (a0 + a1 u)(b0 + b1 u):  even lane: a0 b0 - a1 b1 ; odd lane: a1 b0 + a0 b1
// Both products of a lane go through ONE double-width column pass with a single Montgomery reduction, in the 9 x 29-bit limbs of
// fq29.cuh (no carry instructions): the saturated operands are cut into limbs on the way in -- the left factors shifted by 5 bits,
// which turns the 2^256 Montgomery form into the 2^261 one the stream reduces by -- and the result (< 1.6p) is brought back to
// the canonical saturated residue, so nothing above this function changes. The even lane's subtraction is (64p - 32 a1) b1.
// ~1,700 SIMD-cycles instead of ~2,550 for two saturated products and a modular addition.
// limbs of a value this kernel's own (compiler-scheduled) instructions produced, fetched across the lane pair: the hazard recogniser pads
// the DPP reads itself. CTRL = quad_perm: 0xB1 [1,0,3,2] the partner's value, 0xA0 [0,0,2,2] the even lane's, 0xF5 [1,1,3,3] the odd lane's.
template <int CTRL>
KDEV U29 u29_quad(const U29& a) {
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xF, 0xF, true);
  return r;
}
static KNOINLINE Fq2d fq2d_mul(const Fq2d a, const Fq2d b) {
  const bool odd = lane_odd() != 0;
  // every lane cuts its OWN two operands into limbs and fetches limbs over DPP (27 moves) instead of fetching the saturated words and cutting
  // four operands. even lane: a0 b0 + (64p - 32 a1) b1, odd lane: a1 b0 + a0 b1 -- the right factors are b0 and b1 in BOTH lanes
  const U29 A1 = u29_from_sat_shift5(a.v.l), Bs = u29_from_sat_plain(b.v.l);
  const U29 Ao = u29_quad<0xB1>(A1), B0 = u29_quad<0xA0>(Bs), B1 = u29_quad<0xF5>(Bs);
  U29 C;
#pragma unroll
  for (int i = 0; i < 9; i++) C.l[i] = odd ? Ao.l[i] : Q29::K64[i] - Ao.l[i];   // 64p - 32 a1 > 0 in every limb (32p would underflow the top limb for
                                                                                 // a1 near p); limbs below 1.5 * 2^30
  Fq2d r;
  u29_pack_canonical(r.v.l, u29_mul2(A1, B0, C, B1));
  return r;
}
// (a0 + a1 u)^2:  even lane: (a0 + a1)(a0 - a1) ; odd lane: 2 a0 a1  -- one product per lane, in the 29-bit limbs: the sum and the doubling are
// lazy limb operations (no reduction), only the difference a0 - a1 is a saturated modular subtraction. The factor 2^5 that turns the 2^256
// Montgomery form into the 2^261 one rides on the second factor. even: (a0 + a1) < 2p times 32 (a0 - a1 mod p) < 32p; odd: a0 < p times
// 64 a1 < 64p: products below 64 p^2, results below 1.4p; one factor has exact limbs, the other limbs below 2^30.
// 162 v_mad_u64_u32 in 383 instructions instead of 128 + 148 carry instructions in 444.
static KNOINLINE Fq2d fq2d_sqr(const Fq2d a) {
  const bool odd = lane_odd() != 0;
  const Fq ao = fq_partner(a.v);
  const Fq d = a.v - ao;                                    // used by the even lane: a0 - a1
  const U29 As = u29_from_sat_plain(a.v.l), Ao = u29_quad<0xB1>(As);
  const U29 A5 = u29_from_sat_shift5(a.v.l), D5 = u29_from_sat_shift5(d.l);
  U29 x, y;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x.l[i] = odd ? Ao.l[i] : As.l[i] + Ao.l[i];
    y.l[i] = odd ? 2u * A5.l[i] : D5.l[i];
  }
  Fq2d r;
  u29_pack_canonical(r.v.l, u29_mul(x, y));
  return r;
}
KDEV Fq2d operator*(const Fq2d& a, const Fq2d& b) { return fq2d_mul(a, b); }
KDEV Fq2d fq2_sqr(const Fq2d& a) { return fq2d_sqr(a); }
// (9 + u)(a0 + a1 u) = (9 a0 - a1) + (9 a1 + a0) u
KDEV Fq2d fq2_mul_xi(const Fq2d& a) {
  Fq t = fq_dbl(fq_dbl(fq_dbl(a.v))) + a.v;   // 9 * self
  Fq o = fq_partner(a.v);
  return {t + fp_cneg<FqParams>(o, lane_odd() == 0)};
}
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2); the norm inverse is computed redundantly in both lanes
KDEV Fq2d fq2_inv(const Fq2d& a) {
  Fq sq = fq_sqr(a.v);
  Fq n = sq + fq_partner(sq);
  Fq ni = fq_inv(n);
  return {fp_cneg<FqParams>(a.v * ni, lane_odd() != 0)};
}

struct Fq6 { Fq2d c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };

#define M2(a, b) ((a) * (b))
#define S2(a) fq2_sqr((a))

KDEV Fq6 operator+(const Fq6& a, const Fq6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
KDEV Fq6 operator-(const Fq6& a, const Fq6& b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
KDEV Fq6 operator-(const Fq6& a) { return {-a.c0, -a.c1, -a.c2}; }
KDEV Fq6 fq6_mul_v(const Fq6& a) { return {fq2_mul_xi(a.c2), a.c0, a.c1}; }
KDEV Fq6 fq6_zero() { return {fq2d_zero(), fq2d_zero(), fq2d_zero()}; }

static KTOWER void fq6_mul(Fq6* r, const Fq6* a, const Fq6* b) {
  Fq2d v0 = M2(a->c0, b->c0), v1 = M2(a->c1, b->c1), v2 = M2(a->c2, b->c2);
  Fq2d t0 = fq2_mul_xi(M2(a->c1 + a->c2, b->c1 + b->c2) - v1 - v2) + v0;
  Fq2d t1 = M2(a->c0 + a->c1, b->c0 + b->c1) - v0 - v1 + fq2_mul_xi(v2);
  Fq2d t2 = M2(a->c0 + a->c2, b->c0 + b->c2) - v0 - v2 + v1;
  r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
// a * (c0 + c1 v)
static KTOWER void fq6_mul_by_01(Fq6* r, const Fq6* a, const Fq2d* c0, const Fq2d* c1) {
  Fq2d aa = M2(a->c0, *c0), bb = M2(a->c1, *c1);
  Fq2d t1 = fq2_mul_xi(M2(*c1, a->c1 + a->c2) - bb) + aa;
  Fq2d t3 = M2(*c0, a->c0 + a->c2) - aa + bb;
  Fq2d t2 = M2(*c0 + *c1, a->c0 + a->c1) - aa - bb;
  r->c0 = t1; r->c1 = t2; r->c2 = t3;
}
static KTOWER void fq6_inv(Fq6* r, const Fq6* a) {
  Fq2d t0 = S2(a->c0) - fq2_mul_xi(M2(a->c1, a->c2));
  Fq2d t1 = fq2_mul_xi(S2(a->c2)) - M2(a->c0, a->c1);
  Fq2d t2 = S2(a->c1) - M2(a->c0, a->c2);
  Fq2d n = M2(a->c0, t0) + fq2_mul_xi(M2(a->c2, t1) + M2(a->c1, t2));
  Fq2d ni = fq2_inv(n);
  r->c0 = M2(t0, ni); r->c1 = M2(t1, ni); r->c2 = M2(t2, ni);
}

KDEV void fq12_set_one(Fq12* f) {
  f->c0 = fq6_zero(); f->c1 = fq6_zero();
  f->c0.c0 = fq2d_one();
}
static KTOWER void fq12_mul(Fq12* r, const Fq12* a, const Fq12* b) {
  Fq6 t0, t1, m, s0 = a->c0 + a->c1, s1 = b->c0 + b->c1;
  fq6_mul(&t0, &a->c0, &b->c0);
  fq6_mul(&t1, &a->c1, &b->c1);
  fq6_mul(&m, &s0, &s1);
  r->c1 = m - t0 - t1;
  r->c0 = t0 + fq6_mul_v(t1);
}
static KTOWER void fq12_sqr(Fq12* r, const Fq12* a) {  // complex squaring: 2 Fq6 products
  Fq6 ab, s0 = a->c0 + a->c1, s1 = a->c0 + fq6_mul_v(a->c1), t;
  fq6_mul(&ab, &a->c0, &a->c1);
  fq6_mul(&t, &s0, &s1);
  r->c0 = t - ab - fq6_mul_v(ab);
  r->c1 = ab + ab;
}
KDEV void fq12_conj(Fq12* r, const Fq12* a) { r->c0 = a->c0; r->c1 = -a->c1; }
static KTOWER void fq12_inv(Fq12* r, const Fq12* a) {
  Fq6 n, t, ni;
  fq6_mul(&n, &a->c0, &a->c0);
  fq6_mul(&t, &a->c1, &a->c1);
  n = n - fq6_mul_v(t);
  fq6_inv(&ni, &n);
  fq6_mul(&r->c0, &a->c0, &ni);
  fq6_mul(&t, &a->c1, &ni);
  r->c1 = -t;
}
// f *= c0 + (d0 + d1 v) w   (13 Fq2 products instead of 18)
static KTOWER void fq12_mul_by_034(Fq12* f, const Fq2d* c0, const Fq2d* d0, const Fq2d* d1) {
  Fq6 a = {M2(f->c0.c0, *c0), M2(f->c0.c1, *c0), M2(f->c0.c2, *c0)};
  Fq6 b, e, s = f->c0 + f->c1;
  fq6_mul_by_01(&b, &f->c1, d0, d1);
  Fq2d cs = *c0 + *d0;
  fq6_mul_by_01(&e, &s, &cs, d1);
  f->c1 = e - a - b;
  f->c0 = fq6_mul_v(b) + a;
}
// x -> x^(p^k), k = 1, 2, 3
static KTOWER void fq12_frob(Fq12* r, const Fq12* a, int k) {
  Fq2d c[6] = {a->c0.c0, a->c1.c0, a->c0.c1, a->c1.c1, a->c0.c2, a->c1.c2};
  Fq2d o[6];
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    Fq2d t = (k & 1) ? fq2_conj(c[i]) : c[i];
    o[i] = M2(t, fq2d_load(&FROB_W[k][i]));
  }
  r->c0.c0 = o[0]; r->c1.c0 = o[1]; r->c0.c1 = o[2]; r->c1.c1 = o[3]; r->c0.c2 = o[4]; r->c1.c2 = o[5];
}
// Granger-Scott squaring, valid on the cyclotomic subgroup (after the easy part)
static KTOWER void fq12_cyc_sqr(Fq12* r, const Fq12* a) {
  const Fq2d r0 = a->c0.c0, r4 = a->c0.c1, r3 = a->c0.c2, r2 = a->c1.c0, r1 = a->c1.c1, r5 = a->c1.c2;
  Fq2d tmp, t0, t1, t2, t3, t4, t5;
  tmp = M2(r0, r1); t0 = M2(r0 + r1, fq2_mul_xi(r1) + r0) - tmp - fq2_mul_xi(tmp); t1 = fq2_dbl(tmp);
  tmp = M2(r2, r3); t2 = M2(r2 + r3, fq2_mul_xi(r3) + r2) - tmp - fq2_mul_xi(tmp); t3 = fq2_dbl(tmp);
  tmp = M2(r4, r5); t4 = M2(r4 + r5, fq2_mul_xi(r5) + r4) - tmp - fq2_mul_xi(tmp); t5 = fq2_dbl(tmp);
  Fq2d x5 = fq2_mul_xi(t5);
  r->c0.c0 = fq2_dbl(t0 - r0) + t0;
  r->c1.c1 = fq2_dbl(t1 + r1) + t1;
  r->c1.c0 = fq2_dbl(x5 + r2) + x5;
  r->c0.c2 = fq2_dbl(t4 - r3) + t4;
  r->c0.c1 = fq2_dbl(t2 - r4) + t2;
  r->c1.c2 = fq2_dbl(t3 + r5) + t3;
}
// f^(-z): width-3 NAF square-and-multiply with cyclotomic squarings -- on the cyclotomic subgroup the inverse is the conjugate, so
// negative digits are free; digits +-1, +-3 need f and f^3 (one cyclotomic squaring + one product up front) and leave 17 products in
// the loop instead of the 23 of the plain NAF (z has 28 one bits).
static KTOWER void fq12_exp_by_neg_z(Fq12* r, const Fq12* f) {
  Fq12 f3, acc, t;
  fq12_cyc_sqr(&t, f);
  fq12_mul(&f3, &t, f);
  const int top = Z_WNAF3[Z_WNAF3_LEN - 1];      // +1 or +3
  acc = top == 3 ? f3 : *f;
#pragma unroll 1
  for (int i = Z_WNAF3_LEN - 2; i >= 0; i--) {
    fq12_cyc_sqr(&acc, &acc);
    const int d = Z_WNAF3[i];
    if (d != 0) {
      t = (d == 3 || d == -3) ? f3 : *f;
      if (d < 0) fq12_conj(&t, &t);
      fq12_mul(&acc, &acc, &t);
    }
  }
  fq12_conj(r, &acc);
}

// ---- line functions on the twist, homogeneous projective (same formulas as ark-ec bn/g2.rs) ----
struct G2Hom { Fq2d x, y, z; };
struct Line { Fq2d c0, c1, c2; };  // evaluated as c0 * P.y + (c1 * P.x + c2 v) w

static KTOWER void line_double(G2Hom* r, Line* l) {
  Fq2d a = fq2_mul_fq(M2(r->x, r->y), FQ_TWO_INV);
  Fq2d b = S2(r->y), c = S2(r->z);
  Fq2d e = M2(fq2d_load(&G2_B), fq2_dbl(c) + c);
  Fq2d f = fq2_dbl(e) + e;
  Fq2d g = fq2_mul_fq(b + f, FQ_TWO_INV);
  Fq2d h = S2(r->y + r->z) - (b + c);
  Fq2d i = e - b;
  Fq2d j = S2(r->x);
  Fq2d e2 = S2(e);
  r->x = M2(a, b - f);
  r->y = S2(g) - (fq2_dbl(e2) + e2);
  r->z = M2(b, h);
  l->c0 = -h; l->c1 = fq2_dbl(j) + j; l->c2 = i;
}
static KTOWER void line_add(G2Hom* r, const Fq2d* qx, const Fq2d* qy, Line* l) {
  Fq2d theta = r->y - M2(*qy, r->z);
  Fq2d lam = r->x - M2(*qx, r->z);
  Fq2d c = S2(theta), d = S2(lam);
  Fq2d e = M2(lam, d), f = M2(r->z, c), g = M2(r->x, d);
  Fq2d h = e + f - fq2_dbl(g);
  Fq2d ny = M2(theta, g - h) - M2(e, r->y);
  r->x = M2(lam, h);
  r->y = ny;
  r->z = M2(r->z, e);
  // NOTE: written as 0 - theta on purpose. In this (fully inlined, 512-register) context hipcc 7.2 produced a wrong
  // value for the unary form `-theta` here and only here, with the asm and the portable negation alike, while the
  // binary form is correct; the isolated pattern passes the on-device self-test. tests/test_gpu_parity.py pins the whole
  // line table against the big-int oracle (test_g2_line_table_vs_oracle) so any recurrence is caught.
  l->c0 = lam; l->c1 = fq2d_zero() - theta; l->c2 = M2(theta, *qx) - M2(lam, *qy);
}
KDEV void ell(Fq12* f, const Line* l, const G1Aff* p) {
  Fq2d c0 = fq2_mul_fq(l->c0, p->y), c1 = fq2_mul_fq(l->c1, p->x);
  fq12_mul_by_034(f, &c0, &c1, &l->c2);
}

// Line table of a fixed Q (ark-ec's G2Prepared), per lane parity: lines[li * 2 + parity]
constexpr int MILLER_MAX_LINES = 96;
// lines == nullptr: compute the lines on the fly from (qx, qy). lines_out != nullptr: only tabulate them.
static KTOWER void miller_loop(Fq12* f, const G1Aff* p, const Fq2d* qx, const Fq2d* qy, const Line* __restrict__ lines,
                                   Line* __restrict__ lines_out) {
  const u32 par = lane_odd();
  fq12_set_one(f);
  G2Hom r = {*qx, *qy, fq2d_one()};
  Fq2d nqy = -*qy;
  Line l;
  int li = 0;
#pragma unroll 1
  for (int i = ATE_LEN - 2; i >= 0; i--) {
    if (i != ATE_LEN - 2 && !lines_out) fq12_sqr(f, f);
    if (lines) l = lines[li * 2 + par]; else line_double(&r, &l);
    if (lines_out) lines_out[li * 2 + par] = l; else ell(f, &l, p);
    li++;
    int d = ATE_NAF[i];
    if (d != 0) {
      if (lines) l = lines[li * 2 + par]; else line_add(&r, qx, d > 0 ? qy : &nqy, &l);
      if (lines_out) lines_out[li * 2 + par] = l; else ell(f, &l, p);
      li++;
    }
  }
  // Q1 = pi(Q), Q2 = -pi^2(Q)
  if (lines) {
    l = lines[li * 2 + par]; li++; ell(f, &l, p);
    l = lines[li * 2 + par]; li++; ell(f, &l, p);
  } else {
    Fq2d q1x = M2(fq2_conj(*qx), fq2d_load(&TWIST_MUL_BY_Q_X)), q1y = M2(fq2_conj(*qy), fq2d_load(&TWIST_MUL_BY_Q_Y));
    Fq2d q2x = M2(fq2_conj(q1x), fq2d_load(&TWIST_MUL_BY_Q_X)), q2y = -M2(fq2_conj(q1y), fq2d_load(&TWIST_MUL_BY_Q_Y));
    line_add(&r, &q1x, &q1y, &l);
    if (lines_out) lines_out[li * 2 + par] = l; else ell(f, &l, p);
    li++;
    line_add(&r, &q2x, &q2y, &l);
    if (lines_out) lines_out[li * 2 + par] = l; else ell(f, &l, p);
    li++;
  }
}

// easy part (p^6-1)(p^2+1), hard part = arkworks' Fuentes-Castaneda chain (exponent 2z(6z^2+3z+1)(p^4-p^2+1)/r)
static KTOWER void final_exponentiation(Fq12* out, const Fq12* fin) {
  Fq12 f1, f2, r, y0, y1, y2, y3, y4, y5, y6, t;
  fq12_conj(&f1, fin);
  fq12_inv(&f2, fin);
  fq12_mul(&r, &f1, &f2);
  f2 = r;
  fq12_frob(&r, &r, 2);
  fq12_mul(&r, &r, &f2);
  fq12_exp_by_neg_z(&y0, &r);
  fq12_cyc_sqr(&y1, &y0);
  fq12_cyc_sqr(&y2, &y1);
  fq12_mul(&y3, &y2, &y1);
  fq12_exp_by_neg_z(&y4, &y3);
  fq12_cyc_sqr(&y5, &y4);
  fq12_exp_by_neg_z(&y6, &y5);
  fq12_conj(&y3, &y3);
  fq12_conj(&y6, &y6);
  Fq12 y7, y8, y9, y10, y11, y12, y13, y14, y15;
  fq12_mul(&y7, &y6, &y4);
  fq12_mul(&y8, &y7, &y3);
  fq12_mul(&y9, &y8, &y1);
  fq12_mul(&y10, &y8, &y4);
  fq12_mul(&y11, &y10, &r);
  fq12_frob(&y12, &y9, 1);
  fq12_mul(&y13, &y12, &y11);
  fq12_frob(&y8, &y8, 2);
  fq12_mul(&y14, &y8, &y13);
  fq12_conj(&r, &r);
  fq12_mul(&t, &r, &y9);
  fq12_frob(&y15, &t, 3);
  fq12_mul(out, &y15, &y14);
}

// GT -> 384 canonical little-endian bytes in ark-serialize order (c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1):
// Fq2 coefficient k of the element (k = 0..5 in memory order) fills the 32-byte slots 2k (even lane) and 2k+1 (odd lane).
KDEV void gt_serialize(u32* out96, const Fq12* f) {
  const Fq2d* c = reinterpret_cast<const Fq2d*>(f);
  const u32 par = lane_odd();
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    u32 w[8];
    fp_from_mont<FqParams>(w, c[i].v);
#pragma unroll
    for (int j = 0; j < 8; j++) out96[8 * (2 * i + par) + j] = w[j];
  }
}

// gt_out[i] = serialize(e(P_i, Q_{i*stride})); identity in either slot -> one. TWO lanes per item.
// fixed_lines != nullptr: the second slots are fixed points whose lines were tabulated by k_g2_prepare -- one table for every item
// (lines_stride == 0, q_stride == 0) or table i * lines_stride for item i (kzg verify: g2 and [tau]_2).
static __global__ void __launch_bounds__(64, 2) k_pairing_batch(const G1Aff* __restrict__ ps, const G2Aff* __restrict__ qs, int q_stride, u32 n,
                                                      const Line* __restrict__ fixed_lines, u32 lines_stride, u32* __restrict__ gt_out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 item = t >> 1;
  const bool live = item < n;
  const u32 i = live ? item : (n - 1);   // tail lanes redo the last item: all 64 lanes must stay active for the DPP exchanges
  G1Aff p = ps[i];
  const G2Aff* q = qs + (size_t)i * q_stride;
  Fq2d qx = fq2d_load(&q->x), qy = fq2d_load(&q->y);
  // identity in either slot (Q = all four components zero, or P = (0,0)): both lanes of the pair agree on `ident`
  u32 qz = (fq_is_zero(qx.v) && fq_is_zero(qy.v)) ? 1u : 0u;
  qz &= (u32)__builtin_amdgcn_update_dpp(0, (int)qz, 0xB1, 0xF, 0xF, true);
  const bool ident = aff_is_inf(p) || qz != 0;
  // wave-uniform control flow: identity items run the same arithmetic (total on zeros) and discard it
  Fq12 f, e;
  miller_loop(&f, &p, &qx, &qy, fixed_lines ? fixed_lines + (size_t)i * lines_stride : nullptr, nullptr);
  final_exponentiation(&e, &f);
  if (ident) fq12_set_one(&e);
  if (live) gt_serialize(gt_out + (size_t)96 * i, &e);
}
// the line sequence of a fixed Q; every lane pair of the single wave computes the same values, pair 0's layout is the table
static __global__ void __launch_bounds__(64) k_g2_prepare(const G2Aff* __restrict__ q, Line* __restrict__ lines_out) {
  Fq12 f;
  G1Aff dummy = {fq_zero(), fq_zero()};
  Fq2d qx = fq2d_load(&q->x), qy = fq2d_load(&q->y);
  miller_loop(&f, &dummy, &qx, &qy, nullptr, lines_out);
}

// ---------------------------------------------------------------------------------------------
// Encapsulation at scale: no pairing per item. In the loop of src/vec.rs:63-66 the commitment C is the same for every
// item, so by bilinearity
//     e(r (C - beta g1), g2) = A^r * B^(-beta r),      A = e(C, g2),  B = e(g1, g2)
// with A, B FIXED for the batch: two fixed-base exponentiations in GT with SIGNED 13-bit window tables T[j][d] = base^(d 2^(13j)),
// d = 1..4096 (20 x 4096 Fq12 entries = 31.5 MB per base). A and B are outputs of the final exponentiation, i.e. unitary, so
// base^(-d) is the conjugate of T[j][d]: at most 40 Fq12 products per item instead of a Miller loop + final exponentiation
// (~8x fewer Fq products; 8-bit unsigned windows needed 64). The value -- hence the serialised bytes and the key -- is identical.
// GT elements are stored in the lane-pair order: 12 Fq per element, slot 2k + parity = Fq2 coefficient k, component parity
// (which is also ark-serialize's coefficient order).
// ---------------------------------------------------------------------------------------------
// window width per table: 13 bits (20 windows x 4096 entries, 31.5 MB) for A = e(C, g2), rebuilt per commitment; 16 bits (16 x 32768,
// 201 MB) for the constant B = e(g1, g2), built once per context. entries = 2^(wb-1) + 1 slots per window, slot 0 unused.
struct GtShape { u32 wb, windows, entries; };
__host__ __device__ inline GtShape gt_shape(u32 wb) { return {wb, (254u + wb - 1u) / wb + ((254u % wb) == 0u ? 1u : 0u), (1u << (wb - 1)) + 1u}; }

KDEV void gt_load(Fq12* f, const Fq* __restrict__ src) {
  Fq2d* c = reinterpret_cast<Fq2d*>(f);
  const u32 par = lane_odd();
#pragma unroll
  for (int k = 0; k < 6; k++) c[k].v = src[2 * k + par];
}
KDEV void gt_store(Fq* __restrict__ dst, const Fq12* f) {
  const Fq2d* c = reinterpret_cast<const Fq2d*>(f);
  const u32 par = lane_odd();
#pragma unroll
  for (int k = 0; k < 6; k++) dst[2 * k + par] = c[k].v;
}
// out[i] = e(P_i, Q_fixed) as a raw GT element (12 Fq, Montgomery). Lines of the fixed Q are given. Two lanes per item.
static __global__ void __launch_bounds__(64, 2) k_pairing_raw_fixed(const G1Aff* __restrict__ ps, u32 n, const Line* __restrict__ fixed_lines,
                                                                   Fq* __restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 item = t >> 1;
  const bool live = item < n;
  const u32 i = live ? item : (n - 1);
  G1Aff p = ps[i];
  Fq2d dummy = fq2d_zero();
  Fq12 f, e;
  miller_loop(&f, &p, &dummy, &dummy, fixed_lines, nullptr);
  final_exponentiation(&e, &f);
  if (aff_is_inf(p)) fq12_set_one(&e);
  if (live) gt_store(out + (size_t)12 * i, &e);
}
// table[j * entries + d] = base^(d 2^(wb j)).  Step 1: the powers of two base^(2^s), s < wb * windows (pows[s]: 12 Fq each, from
// k_pairing_raw_fixed over the multiples 2^s P of k_g1_pow2_chain -- base = e(P, Q)) go to slot 2^(s mod wb) of window s / wb.
static __global__ void __launch_bounds__(256) k_gt_table_scatter(const Fq* __restrict__ pows, Fq* __restrict__ table, GtShape g) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g.wb * g.windows * 12) return;
  const u32 s = t / 12, c = t % 12;
  table[((size_t)(s / g.wb) * g.entries + (1u << (s % g.wb))) * 12 + c] = pows[(size_t)s * 12 + c];
}
// Step 2, level L = 1 .. wb - 2: table[j][2^L + x] = table[j][2^L] * table[j][x], 1 <= x < 2^L (all known from the levels below)
static __global__ void __launch_bounds__(64, 2) k_gt_table_fill(Fq* __restrict__ table, u32 L, GtShape g) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 pairi = t >> 1;
  const u32 per = (1u << L) - 1u;                    // entries of this level per window
  const bool live = pairi < g.windows * per;
  const u32 pi = live ? pairi : 0u;
  const u32 j = pi / per, x = 1u + pi % per;
  Fq12 a, b;
  gt_load(&a, table + ((size_t)j * g.entries + (1u << L)) * 12);
  gt_load(&b, table + ((size_t)j * g.entries + x) * 12);
  fq12_mul(&a, &a, &b);
  if (live) gt_store(table + ((size_t)j * g.entries + (1u << L) + x) * 12, &a);
}
// acc *= base^k from the signed-window table of `base` (k canonical, consumed): digits in (-2^(wb-1), 2^(wb-1)], a digit above the half
// becomes d - 2^wb with a carry (2^wb itself: digit 0, carry 1); a negative digit multiplies by the conjugate (unitary: inverse = conjugate)
static KTOWER void gt_table_exp(Fq12* acc, const Fq* __restrict__ tab, GtShape g, u32* k) {
  Fq12 e;
  u32 carry = 0;
  const u32 half = 1u << (g.wb - 1);
#pragma unroll 1
  for (u32 j = 0; j < g.windows; j++) {
    u32 d = (k[0] & (2u * half - 1u)) + carry;
#pragma unroll
    for (int w = 0; w < 7; w++) k[w] = (k[w] >> g.wb) | (k[w + 1] << (32u - g.wb));
    k[7] >>= g.wb;
    const bool neg = d > half;
    carry = neg ? 1u : 0u;
    if (neg) d = 2u * half - d;
    if (d) {
      gt_load(&e, tab + ((size_t)j * g.entries + d) * 12);
      if (neg) fq12_conj(&e, &e);
      fq12_mul(acc, acc, &e);
    }
  }
}
// gt_out[i] = serialize(A^(r_i) * B^(-(r_i * beta_i)))   (tables of A and B). Two lanes per item.
static __global__ void __launch_bounds__(64, 2) k_gt_encap_exp(const Fq* __restrict__ tab_a, GtShape ga, const Fq* __restrict__ tab_b, GtShape gb,
                                                              const Fr* __restrict__ betas, const Fr* __restrict__ rs, u32 n, u32* __restrict__ gt_out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 item = t >> 1;
  const bool live = item < n;
  const u32 i = live ? item : (n - 1);
  Fr r = rs[i];
  Fr m = fp_neg<FrParams>(fp_mul<FrParams>(r, betas[i]));
  u32 u[8], v[8];
  fp_from_mont<FrParams>(u, r);
  fp_from_mont<FrParams>(v, m);
  Fq12 acc;
  fq12_set_one(&acc);
  gt_table_exp(&acc, tab_a, ga, u);
  gt_table_exp(&acc, tab_b, gb, v);
  if (live) gt_serialize(gt_out + (size_t)96 * i, &acc);
}

// debug / test entry: Miller loop only. out: n x 12 Fq (Montgomery), single-element layout.
static __global__ void __launch_bounds__(64, 2) k_miller_only(const G1Aff* __restrict__ ps, const G2Aff* __restrict__ qs, u32 n, Fq* __restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 item = t >> 1;
  const bool live = item < n;
  const u32 i = live ? item : (n - 1);
  G1Aff p = ps[i];
  Fq2d qx = fq2d_load(&qs[i].x), qy = fq2d_load(&qs[i].y);
  Fq12 f;
  miller_loop(&f, &p, &qx, &qy, nullptr, nullptr);
  const Fq2d* c = reinterpret_cast<const Fq2d*>(&f);
  if (live) {
#pragma unroll 1
    for (int k = 0; k < 6; k++) out[(size_t)12 * i + 2 * k + lane_odd()] = c[k].v;
  }
}
// debug / test entry: final exponentiation only. in: n x Fq12 in Montgomery form, single-element layout
// (c0.c0.c0, c0.c0.c1, ... 12 x Fq); out: n x 384 GT bytes. Two lanes per item.
static __global__ void __launch_bounds__(64, 2) k_final_exp_only(const Fq* __restrict__ in, u32 n, u32* __restrict__ gt_out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 item = t >> 1;
  const bool live = item < n;
  const u32 i = live ? item : (n - 1);
  Fq12 f, e;
  Fq2d* c = reinterpret_cast<Fq2d*>(&f);
#pragma unroll 1
  for (int k = 0; k < 6; k++) c[k].v = in[(size_t)12 * i + 2 * k + lane_odd()];
  final_exponentiation(&e, &f);
  if (live) gt_serialize(gt_out + (size_t)96 * i, &e);
}

// ---- BLAKE3 XOF of a 384-byte GT encoding (single chunk, 6 blocks): replaces src/kem.rs:42-46,65-69 ----
__device__ __constant__ const u32 B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
KDEV u32 rotr32(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
KDEV void b3_g(u32* s, int a, int b, int c, int d, u32 mx, u32 my) {
  s[a] += s[b] + mx; s[d] = rotr32(s[d] ^ s[a], 16); s[c] += s[d]; s[b] = rotr32(s[b] ^ s[c], 12);
  s[a] += s[b] + my; s[d] = rotr32(s[d] ^ s[a], 8);  s[c] += s[d]; s[b] = rotr32(s[b] ^ s[c], 7);
}
KDEV void b3_compress(u32* out16, const u32* cv, const u32* blk, u64 counter, u32 blen, u32 flags) {
  u32 s[16], m[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { s[i] = cv[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) s[8 + i] = B3_IV[i];
  s[12] = (u32)counter; s[13] = (u32)(counter >> 32); s[14] = blen; s[15] = flags;
#pragma unroll
  for (int i = 0; i < 16; i++) m[i] = blk[i];
#pragma unroll
  for (int r = 0; r < 7; r++) {
    b3_g(s, 0, 4, 8, 12, m[0], m[1]); b3_g(s, 1, 5, 9, 13, m[2], m[3]);
    b3_g(s, 2, 6, 10, 14, m[4], m[5]); b3_g(s, 3, 7, 11, 15, m[6], m[7]);
    b3_g(s, 0, 5, 10, 15, m[8], m[9]); b3_g(s, 1, 6, 11, 12, m[10], m[11]);
    b3_g(s, 2, 7, 8, 13, m[12], m[13]); b3_g(s, 3, 4, 9, 14, m[14], m[15]);
    if (r != 6) {
      u32 t[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
#pragma unroll
      for (int i = 0; i < 16; i++) m[i] = t[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) { out16[i] = s[i] ^ s[i + 8]; out16[i + 8] = s[i + 8] ^ cv[i]; }
}
static __global__ void __launch_bounds__(256) k_blake3_gt_xof(const u32* __restrict__ gt, u32 n, unsigned char* __restrict__ key_out, u32 msg_len) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32* in = gt + (size_t)96 * i;
  u32 cv[8], blk[16], o[16];
#pragma unroll
  for (int k = 0; k < 8; k++) cv[k] = B3_IV[k];
#pragma unroll 1
  for (int b = 0; b < 5; b++) {
#pragma unroll
    for (int k = 0; k < 16; k++) blk[k] = in[16 * b + k];
    b3_compress(o, cv, blk, 0, 64, b == 0 ? 1u : 0u);
#pragma unroll
    for (int k = 0; k < 8; k++) cv[k] = o[k];
  }
#pragma unroll
  for (int k = 0; k < 16; k++) blk[k] = in[80 + k];
  unsigned char* dst = key_out + (size_t)i * msg_len;
  for (u32 t = 0; t * 64 < msg_len; t++) {
    b3_compress(o, cv, blk, t, 64, 2u | 8u);  // CHUNK_END | ROOT, output block counter t
    u32 nb = msg_len - t * 64; if (nb > 64) nb = 64;
    for (u32 k = 0; k < nb; k++) dst[t * 64 + k] = (unsigned char)(o[k >> 2] >> (8 * (k & 3)));
  }
}

}  // namespace bn254
