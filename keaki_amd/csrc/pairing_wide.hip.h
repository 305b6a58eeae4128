// The LOW-LATENCY form of the BN254 pairing for gfx950: ONE pairing on twelve lanes (six lane pairs), four pairings per wave.
//
// k_pairing (pairing.hip.h) gives a pairing to one lane PAIR: 2.47 M instructions one after the other -- 4.0-4.9 ms however few pairings a
// call holds, with the device idle: a single `verify` (two pairings, src/kzg.rs:127-151), `decapsulate` (src/kem.rs:55-72) or
// `encapsulate`, and the 260 pairings e(2^s C, g2) behind the GT table of a commitment seen for the first time (api.hip: 4.0 of the 7.3 ms
// of a fresh 2^16-item batch = `kem.encaps_per_s`). Here the six Fq2 coefficients of an Fq12 value live in six lane pairs of one 16-lane
// row (lanes 12..15 shadow pairs 0 and 1 and never deliver anything):
//   * flat basis: Fq12 = Fq2[w]/(w^6 - xi); pair k holds the coefficient of w^k (tower coefficient c_e.c_j sits at k = 2j + e). Every
//     Fq2-level operation is pair261.hip.h's, unchanged, running in six pairs at once.
//   * a general product is the schoolbook sum  c_k = sum_i a_i b'_(k-i),  b' = b or xi b when the index wraps: SIX Fq2 products per pair
//     = two of the six-product streams (u29_dot6_asm) -- 1,220 multiply-adds where the lane-pair form runs the Karatsuba tree of 27
//     products (5,100) one after the other. Operands travel as limbs through an exchange area in LDS (each lane publishes the limbs of
//     its a, the product forms of its b and of xi b; a consumer picks the form by ADDRESS).
//   * the sparse line product is ONE stream per pair, a cyclotomic squaring one Fq4 squaring per pair (pairs (a0,a3), (a1,a4), (a2,a5)
//     of the Granger-Scott form), a Frobenius map one Fq2 product per pair.
//   * the line functions keep the running point T replicated in every pair and spread their Fq2 products over the pairs in rounds (three
//     rounds for a doubling step, four for an addition step); the one inversion of the easy part runs replicated (every pair gathers the
//     whole element and runs the lane-pair fq12_inv).
// 0.78 M instructions per pairing on one wave (0.54 M on the f wave of the two-wave form) instead of 2.47 M; 2.5 x the wave-instructions per
// pairing of k_pairing (profiles/r04_pairing_kernels_pmc_sq_insts.txt), so the launcher uses it while the device is not full (pairing.hip: pairing_launch).
// Same final-exponent program (FE_PROG), same Miller step table, same constants, bit-identical outputs (tests/test_gpu_parity.py).
#pragma once
#include "pairing.hip.h"

namespace bn254 {
namespace pw {
using namespace p261;

// ---- geometry ---------------------------------------------------------------------------------------------------------------------
KDEV u32 wlane() { return threadIdx.x & 63u; }                      // lane inside the wave (k_pairing_wide2 runs two waves per workgroup)
KDEV u32 row_base() { return threadIdx.x & 48u; }
KDEV u32 my_pair() { const u32 p = (threadIdx.x & 15u) >> 1; return p >= 6u ? p - 6u : p; }       // lanes 12..15: shadows of pairs 0, 1
KDEV bool real_lane() { return (threadIdx.x & 15u) < 12u; }
KDEV u32 lane_of(u32 pair, u32 parity) { return row_base() + 2u * pair + parity; }
// The exchange area belongs to ONE wave: its LDS operations execute in program order, so all that is needed between a publish and the fetches
// of other lanes (and between those fetches and the next publish) is that the COMPILER keeps that order -- a wavefront-scope fence and a
// scheduling barrier, no s_barrier (k_pairing_wide2 has two waves per workgroup that run different code between their common barriers).
KDEV void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- exchange area: vector v (nine limbs in three 16-byte chunks) of lane l at ex[(v * 3 + c) * 64 + l] ---------------------------------
constexpr int EX_VECS = 5;                       // limbs of a | y0, y1 of b | y0, y1 of xi b
constexpr int EX_UINT4 = EX_VECS * 3 * 64;
KDEV void publish9(uint4* ex, int v, const U29& x) {
  ex[(v * 3 + 0) * 64 + wlane()] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
  ex[(v * 3 + 1) * 64 + wlane()] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
  ex[(v * 3 + 2) * 64 + wlane()] = make_uint4(x.l[8], 0u, 0u, 0u);
}
KDEV U29 fetch9(const uint4* ex, u32 v, u32 lane) {
  const uint4 a = ex[(v * 3u + 0u) * 64u + lane], b = ex[(v * 3u + 1u) * 64u + lane], c = ex[(v * 3u + 2u) * 64u + lane];
  U29 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
  return r;
}
// the same area seen as Fq words: value i of lane l in chunks 2i, 2i + 1
KDEV void publish_fq(uint4* ex, int i, const Fq& v) {
  const uint4* w = reinterpret_cast<const uint4*>(v.l);
  ex[(2 * i) * 64 + wlane()] = w[0];
  ex[(2 * i + 1) * 64 + wlane()] = w[1];
}
KDEV Fq fetch_fq(const uint4* ex, u32 i, u32 lane) {
  Fq r;
  uint4* w = reinterpret_cast<uint4*>(r.l);
  w[0] = ex[(2u * i) * 64u + lane];
  w[1] = ex[(2u * i + 1u) * 64u + lane];
  return r;
}
KDEV U29 sel9(bool c, const U29& a, const U29& b) {
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// ---- the general product: this lane's component of the coefficient of w^k of a b ---------------------------------------------------------
// Bounds: a canonical (1); y forms of b: 1 | 2 (K2), of xi b (< 11 p): 11 | 16 (K16): a plain term 3, a wrapped term 27; a stream holds at most
// three wrapped terms: 81 < 169, so each stream leaves exact limbs below 2p.
static KNOINLINE Fq w12_mul(const Fq a, const Fq b, uint4* ex) {
  const u32 q = lane_odd(), k = my_pair();
  const U29 A = cut(a), B = cut(b);
  const YF yb = y_of(B, Q29::K2);
  const YF yxb = y_of(xi_limbs(B, quad<0xB1>(B), Q29::K2), Q29::K16);
  wsync();                                            // the readers of the previous exchange are done
  publish9(ex, 0, A); publish9(ex, 1, yb.y0); publish9(ex, 2, yb.y1); publish9(ex, 3, yxb.y0); publish9(ex, 4, yxb.y1);
  wsync();
  Fq acc;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    XF x[3];
    YF y[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const u32 i = 3u * h + t;
      const bool wrap = k < i;
      const u32 j = wrap ? k + 6u - i : k - i;
      x[t].s = fetch9(ex, 0, lane_of(i, q));
      x[t].o = fetch9(ex, 0, lane_of(i, q ^ 1u));
      y[t].y0 = fetch9(ex, wrap ? 3u : 1u, lane_of(j, q));
      y[t].y1 = fetch9(ex, wrap ? 4u : 2u, lane_of(j, q));
    }
    const Fq part = pack(dot3(x[0], y[0], x[1], y[1], x[2], y[2]));
    acc = h ? acc + part : part;
  }
  return acc;
}

// ---- the sparse line product: f *= c0 + (d0 + d1 v) w = c0 + d0 w + d1 w^3 --------------------------------------------------------------
//   c_k = a_k c0 + a_(k-1) d0 + a_(k-3) d1, indices mod 6, xi on the line's side where they wrap (k = 0; k < 3)
// The line's coefficients are known to every pair as LIMBS of this lane's component: c0, d0 exact and < 2p, d1 exact and < p.
// The line's side of it, for THIS lane's pair: the product forms of c0, of d0 or xi d0 (k = 0 wraps), of d1 or xi d1 (k < 3 wraps). The choice is
// made on the limbs, before the forms: bounds c0 4 | d0 with the xi-sized bias 2 + 32 = 34, xi d0 54 | d1 1 + 16 = 17, xi d1 27: at most 85.
struct LineForms { YF c0, y1, y3; };
KDEV LineForms line_forms(U29 c0, U29 d0, const U29& d1) {
  fence9(c0); fence9(d0);
  const u32 k = my_pair();
  const U29 xd0 = xi_limbs(d0, quad<0xB1>(d0), Q29::K4), xd1 = xi_limbs(d1, quad<0xB1>(d1), Q29::K2);
  LineForms r;
  r.c0 = y_of(c0, Q29::K2);
  r.y1 = y_of(sel9(k >= 1u, d0, xd0), Q29::K32);
  r.y3 = y_of(sel9(k >= 3u, d1, xd1), Q29::K16);
  return r;
}
static KNOINLINE Fq w12_mul_forms(const Fq a, const LineForms lf, uint4* ex) {
  const u32 q = lane_odd(), k = my_pair();
  const U29 A = cut(a);
  wsync();
  publish9(ex, 0, A);
  wsync();
  const u32 i1 = k >= 1u ? k - 1u : 5u, i3 = k >= 3u ? k - 3u : k + 3u;
  const XF x0 = x_of(A);
  XF x1, x3;
  x1.s = fetch9(ex, 0, lane_of(i1, q)); x1.o = fetch9(ex, 0, lane_of(i1, q ^ 1u));
  x3.s = fetch9(ex, 0, lane_of(i3, q)); x3.o = fetch9(ex, 0, lane_of(i3, q ^ 1u));
  return pack(dot3(x0, lf.c0, x1, lf.y1, x3, lf.y3));
}
static KNOINLINE Fq w12_mul_034(const Fq a, U29 c0, U29 d0, const U29 d1, uint4* ex) {
  return w12_mul_forms(a, line_forms(c0, d0, d1), ex);
}

// ---- Granger-Scott squaring on the cyclotomic subgroup: one Fq4 squaring per pair ---------------------------------------------------------
// (x, y) = (a0, a3), (a1, a4), (a2, a5);  t_e = x^2 + xi y^2,  t_o = 2 x y  (fq4_sqr);  with the tower positions written in the flat basis:
//   out0 = 3 t_e(a0,a3) - 2 a0    out3 = 3 t_o(a0,a3) + 2 a3    out2 = 3 t_e(a1,a4) - 2 a2    out5 = 3 t_o(a1,a4) + 2 a5
//   out4 = 3 t_e(a2,a5) - 2 a4    out1 = 3 xi t_o(a2,a5) + 2 a1
// One stream per pair: the even pairs need t_e (three Fq products per lane: the square x^2 costs a lane one, (xi y) y two), the odd pairs t_o
// (two), so everybody runs the three-product stream and the odd pairs feed a zero into the first product. Bounds: 6 + 33 | 2 + 4 (pair 1: 22 + 44).
static KNOINLINE Fq w12_cyc_sqr(const Fq a, uint4* ex) {
  const u32 q = lane_odd(), k = my_pair();
  const bool odd = q != 0, even_k = (k & 1u) == 0u;
  wsync();
  publish9(ex, 0, cut(a));
  wsync();
  const u32 xs = k == 0u || k == 3u ? 0u : (k == 2u || k == 5u ? 1u : 2u);
  const U29 xsl = fetch9(ex, 0, lane_of(xs, q)), xol = fetch9(ex, 0, lane_of(xs, q ^ 1u));
  const U29 ysl = fetch9(ex, 0, lane_of(xs + 3u, q)), yol = fetch9(ex, 0, lane_of(xs + 3u, q ^ 1u));
  const YF yy = y_of(ysl, Q29::K2);
  const XF xxy = x_of(xi_limbs(ysl, yol, Q29::K2));
  // pair 1 needs xi t_o = 2 (xi x) y: xi goes onto x (limbs doubled: < 2^30, the wide side of a product; bound 22 + 44)
  const XF xix = x_of(xi_limbs(xsl, xol, Q29::K2));
  const bool k1 = k == 1u;
  U29 p0a, sy, p1a, p2a;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const u32 sx = odd ? xol.l[i] : xsl.l[i] + xol.l[i];                              // limbs < 2^30: one side of a product may be that wide
    sy.l[i] = odd ? 2u * xsl.l[i] : xsl.l[i] - xol.l[i] + Q29::K2[i];
    p0a.l[i] = even_k ? sx : 0u;
    p1a.l[i] = even_k ? xxy.s.l[i] : 2u * (k1 ? xix.s.l[i] : xsl.l[i]);
    p2a.l[i] = even_k ? xxy.o.l[i] : 2u * (k1 ? xix.o.l[i] : xol.l[i]);
  }
  sy = carry(sy);
  U29 r;
  u29_dot3_asm(r.l, p0a.l, sy.l, p1a.l, yy.y0.l, p2a.l, yy.y1.l);
  const Fq2d t = {pack(r)};
  // 3 t - 2 a (even pairs) | 3 t + 2 a (odd pairs)
  const Fq2d own = {fp_cneg<FqParams>(a, even_k)};
  return (fq2_dbl(t + own) + t).v;
}

// ---- Frobenius maps, conjugation ---------------------------------------------------------------------------------------------------
static KNOINLINE Fq w12_frob(const Fq a, int kk) {
  Fq2d t = {a};
  if (kk & 1) t = fq2_conj(t);
  u32 k = my_pair();
  asm volatile("" : "+v"(k));
  return M2(t, fq2d_load(&p261::FROB_W[kk][k])).v;
}
KDEV Fq w12_conj(const Fq& a) { return fq_select((my_pair() & 1u) != 0u, fq_zero() - a, a); }       // f^(p^6): the odd powers of w change sign
KDEV Fq w12_one() { return fq_select(my_pair() == 0u, fq2d_one().v, fq_zero()); }

// ---- gather into the lane-pair form (every pair holds the whole element) and back ---------------------------------------------------------
// tower memory order c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 = flat 0, 2, 4, 1, 3, 5
KDEV void w12_gather(Fq12* f, const Fq a, uint4* ex) {
  const u32 q = lane_odd();
  wsync();
  publish_fq(ex, 0, a);
  wsync();
  f->c0.c0.v = fetch_fq(ex, 0, lane_of(0, q)); f->c0.c1.v = fetch_fq(ex, 0, lane_of(2, q)); f->c0.c2.v = fetch_fq(ex, 0, lane_of(4, q));
  f->c1.c0.v = fetch_fq(ex, 0, lane_of(1, q)); f->c1.c1.v = fetch_fq(ex, 0, lane_of(3, q)); f->c1.c2.v = fetch_fq(ex, 0, lane_of(5, q));
}
KDEV Fq w12_own(const Fq12* f) {
  const u32 k = my_pair();
  Fq r = f->c0.c0.v;
  r = fq_select(k == 1u, f->c1.c0.v, r); r = fq_select(k == 2u, f->c0.c1.v, r); r = fq_select(k == 3u, f->c1.c1.v, r);
  r = fq_select(k == 4u, f->c0.c2.v, r); r = fq_select(k == 5u, f->c1.c2.v, r);
  return r;
}
static KNOINLINE Fq w12_inv(const Fq a, uint4* ex) {
  Fq12 f, r;
  w12_gather(&f, a, ex);
  fq12_inv<false>(&r, &f);
  return w12_own(&r);
}

// ---- line functions: the running point T = (X, Y, Z) is REPLICATED in every pair, the Fq2 products of a line function are not --------------
// A round: pair k multiplies the two operands it picked, everybody fetches the results it needs (Fq words through the exchange area). The
// formulas are line_double / line_add of pairing.hip.h (ark-ec bn/g2.rs) with the products grouped by dependency: three rounds for a doubling
// step (5 + 2 + 3 products) instead of ten products one after the other, four for an addition step (2 + 4 + 3 + 4 instead of thirteen).
KDEV Fq2d pick6(u32 k, const Fq2d& v0, const Fq2d& v1, const Fq2d& v2, const Fq2d& v3, const Fq2d& v4, const Fq2d& v5) {
  Fq r = v0.v;
  r = fq_select(k == 1u, v1.v, r); r = fq_select(k == 2u, v2.v, r); r = fq_select(k == 3u, v3.v, r);
  r = fq_select(k == 4u, v4.v, r); r = fq_select(k == 5u, v5.v, r);
  return {r};
}
template <int N>
KDEV void round_of_products(Fq2d (&out)[N], const Fq2d& u, const Fq2d& v, uint4* ex) {
  const Fq2d r = M2(u, v);
  const u32 q = lane_odd();
  wsync();
  publish_fq(ex, 0, r.v);
  wsync();
#pragma unroll
  for (int t = 0; t < N; t++) out[t].v = fetch_fq(ex, 0, lane_of((u32)t, q));
}
static KNOINLINE void w_line_double(G2Hom* r, Line* l, uint4* ex) {
  const u32 k = my_pair();
  const Fq2d X = r->x, Y = r->y, Z = r->z;
  const Fq2d yz = Y + Z;
  Fq2d p[5];                                                            // X Y | Y^2 | Z^2 | (Y + Z)^2 | X^2
  round_of_products(p, pick6(k, X, Y, Z, yz, X, X), pick6(k, Y, Y, Z, yz, X, Y), ex);
  const Fq2d a = fq2_half(p[0]), b = p[1], c = p[2], h = p[3] - (b + c), j = p[4];
  Fq2d s[2];                                                            // e = B 3c | Z' = b h
  round_of_products(s, pick6(k, fq2d_load(&p261::G2_B), b, b, b, b, b), pick6(k, fq2_dbl(c) + c, h, h, h, h, h), ex);
  const Fq2d e = s[0], f = fq2_dbl(e) + e, g = fq2_half(b + f), bf = b - f;
  Fq2d t[3];                                                            // e^2 | a (b - f) | g^2
  round_of_products(t, pick6(k, e, a, g, g, g, g), pick6(k, e, bf, g, g, g, g), ex);
  r->x = t[1];
  r->y = t[2] - (fq2_dbl(t[0]) + t[0]);
  r->z = s[1];
  l->c0 = fq2_neg(h); l->c1 = fq2_dbl(j) + j; l->c2 = e - b;
}
static KNOINLINE void w_line_add(G2Hom* r, const Fq2d qx, const Fq2d qy, Line* l, uint4* ex) {
  const u32 k = my_pair();
  const Fq2d X = r->x, Y = r->y, Z = r->z;
  Fq2d p[2];                                                            // qy Z | qx Z
  round_of_products(p, pick6(k, qy, qx, qx, qx, qx, qx), Z, ex);
  const Fq2d theta = Y - p[0], lam = X - p[1];
  Fq2d s[4];                                                            // theta^2 | lam^2 | theta qx | lam qy
  round_of_products(s, pick6(k, theta, lam, theta, lam, lam, lam), pick6(k, theta, lam, qx, qy, qy, qy), ex);
  const Fq2d c = s[0], d = s[1];
  Fq2d t[3];                                                            // e = lam d | f = Z c | g = X d
  round_of_products(t, pick6(k, lam, Z, X, X, X, X), pick6(k, d, c, d, d, d, d), ex);
  const Fq2d e = t[0], f = t[1], g = t[2], h = e + f - fq2_dbl(g), gh = g - h;
  Fq2d w[4];                                                            // lam h | theta (g - h) | e Y | Z e
  round_of_products(w, pick6(k, lam, theta, e, Z, Z, Z), pick6(k, h, gh, Y, e, e, e), ex);
  r->x = w[0];
  r->y = w[1] - w[2];
  r->z = w[3];
  l->c0 = lam; l->c1 = fq2_neg(theta); l->c2 = s[2] - s[3];
}

// ---- Miller loop ---------------------------------------------------------------------------------------------------------------------
static KTOWER Fq w_miller(const Fq& px, const Fq& py, const Fq* __restrict__ qw, const Line* __restrict__ lines, uint4* ex) {
  const u32 par = lane_odd();
  Fq f = w12_one();
  G2Hom r;
  if (!lines) { r.x.v = to261(qw[par]); r.y.v = to261(qw[2 + par]); r.z = fq2d_one(); }
  const U29 pxl = cut(px), pyl = cut(py);
#pragma unroll 1
  for (int li = 0; li < MILLER_NSTEPS; li++) {
    const int st = MILLER_STEPS[li];
    if (st == 1) f = w12_mul(f, f, ex);
    Line l;
    if (lines) {
      l = lines[li * 2 + par];
    } else if (st <= 1) {
      w_line_double(&r, &l, ex);
    } else {
      Fq2d ax = {to261(qw[par])}, ay = {to261(qw[2 + par])};                                    // Q, -Q, pi(Q), -pi^2(Q)
      if (st == 3) ay = fq2_neg(ay);
      if (st >= 4) {
        ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y));
        if (st == 5) { ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = fq2_neg(M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y))); }
      }
      w_line_add(&r, ax, ay, &l, ex);
    }
    const U29 c0 = u29_mul(cut(l.c0.v), pyl), d0 = u29_mul(cut(l.c1.v), pxl);
    f = w12_mul_034(f, c0, d0, cut(l.c2.v), ex);
  }
  return f;
}

// ---- the final exponentiation: FE_PROG on the accumulator, the twelve slots in LDS (slot s of lane l: chunks (2s, 2s + 1) x 64 + l) -----------
static KTOWER Fq w_final_exp(Fq acc, uint4* ex, uint4* slots) {
#pragma unroll 1
  for (int pc = 0; pc < FE_NOPS; pc++) {
    const u32 op = FE_PROG[pc], code = op & 15u, s = op >> 4;
    if (code == 0) {
      acc = fetch_fq(slots, s, wlane());
    } else if (code == 1) {
      publish_fq(slots, (int)s, acc);
    } else if (code == 2) {
      acc = w12_cyc_sqr(acc, ex);
    } else if (code == 3 || code == 4) {
      Fq b = fetch_fq(slots, s, wlane());
      if (code == 4) b = w12_conj(b);
      acc = w12_mul(acc, b, ex);
    } else if (code == 5) {
      acc = w12_conj(acc);
    } else if (code == 6) {
      acc = w12_frob(acc, (int)s);
    } else {
      acc = w12_inv(acc, ex);
    }
  }
  return acc;
}

// ---- lines on the fly, TWO waves per workgroup: wave 1 runs the line functions one step AHEAD, wave 0 the f-updates ---------------------------
// The line of step li + 1 (the rounds of w_line_double / w_line_add and the two products by P's coordinates) does not depend on f: with the running
// point T in a wave of its own the Miller loop's critical path is the f-chain alone -- 64 x (product + line product) + 24 line products instead of
// that PLUS 64 doubling and 24 addition steps of T. The line travels through a double-buffered mailbox in LDS (c0, d0, d1 as limbs of the lane's
// parity; lane l of wave 0 reads what lane l of wave 1 wrote); one workgroup barrier per step.
// With TABULATED lines (FIXED: the second slot is g2, [tau]_2 or a multiple 2^s g2) the second wave has no T to walk: it loads the line, multiplies
// it by P's coordinates and prepares its product forms for every lane's pair -- the mailbox then carries the six form vectors and the f wave's line
// product is publish, fetch, one stream.
constexpr int MB_VECS_LIMBS = 3, MB_VECS_FORMS = 6;
template <bool FIXED>
static __global__ void __launch_bounds__(128) k_pairing_wide2(PairArgs a) {
  constexpr int MBV = FIXED ? MB_VECS_FORMS : MB_VECS_LIMBS;
  __shared__ uint4 ex[2 * EX_UINT4];
  __shared__ uint4 slots[FE_NSLOTS * 2 * 64];
  __shared__ uint4 mb[2 * MBV * 3 * 64];
  const u32 wave = threadIdx.x >> 6;
  uint4* myex = ex + wave * EX_UINT4;
  const u32 item = blockIdx.x * 4u + (wlane() >> 4);
  const bool live = item < a.n;
  const u32 i = live ? item : (a.n - 1);
  const u32 par = lane_odd(), k = my_pair();
  const u32 m = (k & 1u) * 3u + (k >> 1);
  const G1Aff p = a.ps[(size_t)i * a.p_stride];
  const Fq* qw = nullptr;
  u32 qz = 0;
  if constexpr (!FIXED) {
    qw = reinterpret_cast<const Fq*>(a.qs + (size_t)i * a.q_stride);
    qz = (fq_is_zero(qw[par]) && fq_is_zero(qw[2 + par])) ? 1u : 0u;
    qz &= (u32)__builtin_amdgcn_update_dpp(0, (int)qz, 0xB1, 0xF, 0xF, true);
  }
  const bool ident = aff_is_inf(p) || qz != 0;
  Fq f = w12_one();
  if (wave == 1) {
    // the line wave
    const U29 pxl = cut(to261(p.x)), pyl = cut(to261(p.y));
    const Line* lines = FIXED ? a.fixed_lines + (size_t)i * a.lines_stride : nullptr;
    G2Hom r;
    if constexpr (!FIXED) { r.x.v = to261(qw[par]); r.y.v = to261(qw[2 + par]); r.z = fq2d_one(); }
#pragma unroll 1
    for (int li = 0; li < MILLER_NSTEPS; li++) {
      const int st = MILLER_STEPS[li];
      Line l;
      if constexpr (FIXED) {
        l = lines[li * 2 + par];
      } else if (st <= 1) {
        w_line_double(&r, &l, myex);
      } else {
        Fq2d ax = {to261(qw[par])}, ay = {to261(qw[2 + par])};
        if (st == 3) ay = fq2_neg(ay);
        if (st >= 4) {
          ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y));
          if (st == 5) { ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = fq2_neg(M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y))); }
        }
        w_line_add(&r, ax, ay, &l, myex);
      }
      const U29 c0 = u29_mul(cut(l.c0.v), pyl), d0 = u29_mul(cut(l.c1.v), pxl), d1 = cut(l.c2.v);
      uint4* box = mb + (li & 1) * (MBV * 3 * 64);
      if constexpr (FIXED) {
        const LineForms lf = line_forms(c0, d0, d1);
        publish9(box, 0, lf.c0.y0); publish9(box, 1, lf.c0.y1); publish9(box, 2, lf.y1.y0); publish9(box, 3, lf.y1.y1);
        publish9(box, 4, lf.y3.y0); publish9(box, 5, lf.y3.y1);
      } else {
        publish9(box, 0, c0); publish9(box, 1, d0); publish9(box, 2, d1);
      }
      __syncthreads();                                   // line li is there; wave 0 is done with the buffer of line li - 1
    }
    return;
  }
  // the f wave
#pragma unroll 1
  for (int li = 0; li < MILLER_NSTEPS; li++) {
    if (MILLER_STEPS[li] == 1) f = w12_mul(f, f, myex);    // overlaps the line wave's work on line li
    __syncthreads();
    const uint4* box = mb + (li & 1) * (MBV * 3 * 64);
    if constexpr (FIXED) {
      LineForms lf;
      lf.c0.y0 = fetch9(box, 0, wlane()); lf.c0.y1 = fetch9(box, 1, wlane()); lf.y1.y0 = fetch9(box, 2, wlane()); lf.y1.y1 = fetch9(box, 3, wlane());
      lf.y3.y0 = fetch9(box, 4, wlane()); lf.y3.y1 = fetch9(box, 5, wlane());
      f = w12_mul_forms(f, lf, myex);
    } else {
      const U29 c0 = fetch9(box, 0, wlane()), d0 = fetch9(box, 1, wlane()), d1 = fetch9(box, 2, wlane());
      f = w12_mul_034(f, c0, d0, d1, myex);
    }
  }
  if (a.mode & PAIR_FINAL_EXP) f = w_final_exp(f, myex, slots);
  if (ident) f = w12_one();
  if (!live || !real_lane()) return;
  if (a.mode & PAIR_OUT_BYTES) {
    u32 w[8];
    canon_words(w, f);
    u32* o = (u32*)a.out + (size_t)96 * item + 8 * (2 * m + par);
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = w[j];
  } else {
    Fq* o = (Fq*)a.out + (size_t)12 * item;
    o[2 * m + par] = (a.mode & PAIR_OUT_RAW256) ? to256(f) : f;
  }
}

// Same arguments and modes as k_pairing (PairArgs; `ws` is not used: the slots live in LDS). Sixteen lanes per item, four items per wave.
static __global__ void __launch_bounds__(64) k_pairing_wide(PairArgs a) {
  __shared__ uint4 ex[EX_UINT4];
  __shared__ uint4 slots[FE_NSLOTS * 2 * 64];
  const u32 item = blockIdx.x * 4u + (threadIdx.x >> 4);
  const bool live = item < a.n;
  const u32 i = live ? item : (a.n - 1);
  const u32 par = lane_odd(), k = my_pair();
  const u32 m = (k & 1u) * 3u + (k >> 1);               // position of w^k in the tower's memory order (and in ark-serialize's)
  Fq f;
  bool ident = false;
  if (a.mode & PAIR_MILLER) {
    const G1Aff p = a.ps[(size_t)i * a.p_stride];
    const Fq* qw = nullptr;
    u32 qz = 0;
    if (!a.fixed_lines) {
      qw = reinterpret_cast<const Fq*>(a.qs + (size_t)i * a.q_stride);
      qz = (fq_is_zero(qw[par]) && fq_is_zero(qw[2 + par])) ? 1u : 0u;
      qz &= (u32)__builtin_amdgcn_update_dpp(0, (int)qz, 0xB1, 0xF, 0xF, true);
    }
    ident = aff_is_inf(p) || qz != 0;
    f = w_miller(to261(p.x), to261(p.y), qw, a.fixed_lines ? a.fixed_lines + (size_t)i * a.lines_stride : nullptr, ex);
  } else {
    f = to261(a.f_in[(size_t)12 * i + 2 * m + par]);
  }
  if (a.mode & PAIR_FINAL_EXP) f = w_final_exp(f, ex, slots);
  if (ident) f = w12_one();
  if (!live || !real_lane()) return;
  if (a.mode & PAIR_OUT_BYTES) {
    u32 w[8];
    canon_words(w, f);
    u32* o = (u32*)a.out + (size_t)96 * item + 8 * (2 * m + par);
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = w[j];
  } else {
    Fq* o = (Fq*)a.out + (size_t)12 * item;
    o[2 * m + par] = (a.mode & PAIR_OUT_RAW256) ? to256(f) : f;
  }
}

}  // namespace pw
}  // namespace bn254

// ---- the GT side of a SMALL encapsulation batch and of a table's first levels in the twelve-lane form ------------------------------------
// (k_gt_encap_exp / k_gt_table_fill of pairing.hip.h, same tables -- 12 Fq per element, slot 2m + parity -- same digits, same products: an item's
// ~30 Fq12 products run 1,800 instructions each instead of ~8,000, which is what a single `encapsulate` call and the first, latency-bound fill
// levels of a new commitment's table wait for.)
namespace bn254 {
namespace pw {

static KTOWER Fq w_gt_table_exp(Fq acc, const Fq* __restrict__ tab, GtShape g, const u32 (&kx)[8], uint4* ex, u32* kw) {
  const u32 par = lane_odd(), k = my_pair(), m = (k & 1u) * 3u + (k >> 1);
#pragma unroll
  for (int w = 0; w < 8; w++) kw[w * 64 + wlane()] = kx[w];          // a window's bits are read from LDS: the word index is a run-time value
  auto word = [&](u32 w) -> u32 { return w < 8u ? kw[w * 64u + wlane()] : 0u; };
  u32 carry_d = 0;
  const u32 half = 1u << (g.wb - 1);
#pragma unroll 1
  for (u32 j = 0; j < g.windows; j++) {
    const u32 off = j * g.wb, w = off >> 5, sh = off & 31u;
    u32 d = (__builtin_amdgcn_alignbit(word(w + 1), word(w), sh) & (2u * half - 1u)) + carry_d;
    const bool neg = d > half;
    carry_d = neg ? 1u : 0u;
    if (neg) d = 2u * half - d;
    Fq b = w12_one();                                                     // a zero digit multiplies by one: the rows of a wave hold different digits
    if (d) b = tab[((size_t)j * g.entries + d) * 12 + 2 * m + par];
    if (neg) b = w12_conj(b);                                             // unitary: the inverse is the conjugate
    acc = w12_mul(acc, b, ex);
  }
  return acc;
}
static __global__ void __launch_bounds__(64) k_gt_encap_exp_wide(const Fq* __restrict__ tab_a, GtShape ga, const Fq* __restrict__ tab_b, GtShape gb,
                                                                const Fr* __restrict__ betas, const Fr* __restrict__ rs, u32 n, u32* __restrict__ gt_out) {
  __shared__ uint4 ex[EX_UINT4];
  __shared__ u32 kw[8 * 64];
  const u32 item = blockIdx.x * 4u + (threadIdx.x >> 4);
  const bool live = item < n;
  const u32 i = live ? item : (n - 1);
  Fq acc = w12_one();
  {
    u32 u[8];
    fp_from_mont<FrParams>(u, rs[i]);
    acc = w_gt_table_exp(acc, tab_a, ga, u, ex, kw);
  }
  {
    u32 v[8];
    fp_from_mont<FrParams>(v, fp_neg<FrParams>(fp_mul<FrParams>(rs[i], betas[i])));
    acc = w_gt_table_exp(acc, tab_b, gb, v, ex, kw);
  }
  if (!live || !real_lane()) return;
  const u32 k = my_pair(), m = (k & 1u) * 3u + (k >> 1);
  u32 w[8];
  canon_words(w, acc);
  u32* o = gt_out + (size_t)96 * item + 8 * (2 * m + lane_odd());
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = w[j];
}
// table[j][2^L + x] = table[j][2^L] * table[j][x], 1 <= x < 2^L: one row per entry
static __global__ void __launch_bounds__(64) k_gt_table_fill_wide(Fq* __restrict__ table, u32 L, GtShape g) {
  __shared__ uint4 ex[EX_UINT4];
  const u32 e = blockIdx.x * 4u + (threadIdx.x >> 4);
  const u32 per = (1u << L) - 1u;
  const bool live = e < g.windows * per;
  const u32 pi = live ? e : 0u;
  const u32 j = pi / per, x = 1u + pi % per;
  const u32 k = my_pair(), m = (k & 1u) * 3u + (k >> 1), c = 2 * m + lane_odd();
  const Fq a = table[((size_t)j * g.entries + (1u << L)) * 12 + c], b = table[((size_t)j * g.entries + x) * 12 + c];
  const Fq r = w12_mul(a, b, ex);
  if (live && real_lane()) table[((size_t)j * g.entries + (1u << L) + x) * 12 + c] = r;
}

}  // namespace pw
}  // namespace bn254
