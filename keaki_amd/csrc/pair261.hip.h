// Fq2 / Fq6 / Fq12 tower of the BN254 pairing for gfx950, one lane PAIR per element, residues in the 2^261 Montgomery form.
//
// Replaces what `E::pairing` runs inside ark-ec 0.4.2 (models/bn: Fp12 arithmetic under multi_miller_loop and
// final_exponentiation; reference call sites src/kem.rs:30,58, src/kzg.rs:148). Tower as ark-bn254: Fq2 = Fq[u]/(u^2+1),
// Fq6 = Fq2[v]/(v^3 - (9+u)), Fq12 = Fq6[w]/(w^2 - v).
//
// Lane-pair layout (as round 1): an Fq2 element a0 + a1 u lives in two adjacent lanes -- even lane a0, odd lane a1 (type Fq2d =
// "this lane's component", 8 x 32-bit canonical words). Additions are one Fq operation per lane; the partner's component arrives
// by DPP quad_perm (register to register); both lanes run the same instruction stream.
//
// What is new in round 2 -- where the instructions went, and how they were removed:
//   * Radix. Everything in this file is a residue x 2^261 mod p, the radix the 9 x 29-bit product streams of fq29_asm.hip.h reduce
//     by, so no operand needs the 5-bit shift that bridged from the 2^256 form (and multiplied its bound by 32). Inputs are
//     converted once at kernel entry (to261), GT leaves through the same from-Montgomery product it always needed.
//   * Lazy reduction ACROSS products (Aranha et al., "Faster explicit formulas for computing pairings over ordinary curves",
//     restated for the column streams): an output coefficient that is a sum of up to three Fq2 products -- every coefficient of an Fq6
//     product written out by the schoolbook rule, of the sparse line product, of an Fq4 squaring -- is ONE stream of up to six Fq
//     products with ONE Montgomery reduction (u29_dot6_asm: 567 v_mad_u64_u32 in 610 instructions; a column holds 63 terms below
//     2^58, the budget of a 64-bit accumulator). The Karatsuba forms of round 1 needed fewer multiply-adds (6 x 243 per Fq6 product
//     against 3 x 567) but ~27 saturated additions (25 instructions each), two multiplications by 9+u (143 each) and a limb cut +
//     pack round trip per product around them: 3,540 instructions per Fq6 product then, ~2,500 now.
//   * Operands are cut into limbs ONCE per tower operation (not once per product), the partner exchange is done on the limbs once,
//     the multiplication by 9+u is applied to limbs (xi_limbs: shifts and adds, one carry pass) and the sign of the even lane's
//     -a1 b1 term is carried by the y operand (2p - y1), prepared once per operand.
// Value bounds (multiples of p) and limb classes are stated at every helper; the streams' budget is: every limb <= 2^29 + 8.
#pragma once
#include "bn254_curve.hip.h"
#include "fq29.hip.h"
#include "fq29_dot_asm.hip.h"
#include "pair261_constants.hip.h"

namespace bn254 {
namespace p261 {

#define KNOINLINE __device__ __noinline__
#define KTOWER __device__ __forceinline__

struct Fq2d { Fq v; };   // this lane's component; canonical (< p), 2^261-form
struct Fq6 { Fq2d c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };

KDEV u32 lane_odd() { return threadIdx.x & 1u; }

// ---- lane-pair exchange ---------------------------------------------------------------------------------------------------------
// gfx950 needs two wait states between a VALU write of a VGPR and a DPP read of it. hipcc pads that for instructions it scheduled
// itself but does not look into inline-asm streams: a value that comes straight out of a stream passes through a fence first.
KDEV void fence8(u32 (&x)[8]) {
  asm volatile("s_nop 1" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
KDEV void fence9(U29& x) {
  asm volatile("s_nop 1" : "+v"(x.l[0]), "+v"(x.l[1]), "+v"(x.l[2]), "+v"(x.l[3]), "+v"(x.l[4]), "+v"(x.l[5]), "+v"(x.l[6]), "+v"(x.l[7]), "+v"(x.l[8]));
}
// CTRL = quad_perm: 0xB1 [1,0,3,2] the partner's value, 0xA0 [0,0,2,2] the even lane's, 0xF5 [1,1,3,3] the odd lane's. All 64 lanes active.
// The moves are kept as moves: hipcc's DPP combiner likes to fold a v_mov_b32_dpp into the VALU instruction that consumes it, and for
// `own - partner(own)` it produced  v_subrev_u32_dpp v0, v0, v0 quad_perm:[1,0,3,2]  -- DPP operand and plain operand in the SAME
// register -- which gives wrong values on gfx950 (found with bench_tools/dbg/dbg_pair.hip: every variant of fq2d_sqr with that
// instruction failed in the even lanes, the variant whose partner limbs pass through an asm statement is correct).
// The empty statement below owns the nine results, so nothing is folded; the streams need the limbs in registers of their own anyway.
template <int CTRL>
KDEV U29 quad(const U29& a) {
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xF, 0xF, true);
  asm volatile("" : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]), "+v"(r.l[8]));
  return r;
}
KDEV Fq fq_partner(const Fq& a) {        // safe for stream outputs (fenced)
  Fq t = a, r;
  fence8(t.l);
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)t.l[i], 0xB1, 0xF, 0xF, true);
  asm volatile("" : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]));   // no DPP folding (see quad)
  return r;
}
KDEV Fq fq_select(bool c, const Fq& a, const Fq& b) {  // c ? a : b
  Fq r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = c ? a.l[j] : b.l[j];
  return r;
}

// ---- LDS parking ----------------------------------------------------------------------------------------------------------------
// Values that are not needed while a big tower operation runs wait in LDS instead of in registers the operation wants (the compiler
// would spill them to scratch -- and saves every live caller-saved register around each call of an Fq2 product). One 16-byte chunk per
// lane and index: chunk c of lane l at park[c * 64 + l] (conflict-free). 18 chunks per lane: 18 KB per 64-lane workgroup, 8 per CU.
// Fq index i = chunks 2i, 2i + 1. Map: 0..5 = an Fq12 (or T of the Miller loop in 0..2 and temporaries in 3..5), 6..7 = Q of the Miller loop.
constexpr int PARK_CHUNKS = 18;
KDEV void park_fq(uint4* park, int idx, const Fq& v) {
  const uint4* w = reinterpret_cast<const uint4*>(v.l);
  park[(2 * idx) * 64 + threadIdx.x] = w[0];
  park[(2 * idx + 1) * 64 + threadIdx.x] = w[1];
}
KDEV Fq unpark_fq(const uint4* park, int idx) {
  Fq r;
  uint4* w = reinterpret_cast<uint4*>(r.l);
  w[0] = park[(2 * idx) * 64 + threadIdx.x];
  w[1] = park[(2 * idx + 1) * 64 + threadIdx.x];
  return r;
}

// ---- limbs <-> words -----------------------------------------------------------------------------------------------------------
KDEV U29 cut(const Fq& a) { return u29_from_sat_plain(a.l); }                       // exact limbs (< 2^29), value = the residue (< p)
KDEV Fq pack(const U29& t) { Fq r; u29_pack_canonical(r.l, t); return r; }          // t: exact limbs, value < 2p  ->  canonical words
KDEV U29 carry(const U29& x) { return u29_carry(x); }                               // any u32 limbs -> limbs <= 2^29 + 7, same value

// ---- Fq in the 2^261 form (one lane) --------------------------------------------------------------------------------------------
KDEV Fq fq_mul261(const Fq& a, const Fq& b) { return pack(u29_mul(cut(a), cut(b))); }
KDEV Fq fq_sqr261(const Fq& a) { return pack(u29_sqr(cut(a))); }
KDEV Fq to261(const Fq& a256) { return pack(u29_mul(cut(a256), u29_const(Conv::C266))); }     // x 2^256 -> x 2^261
KDEV Fq to256(const Fq& a261) { return pack(u29_mul(cut(a261), u29_const(Q29::R256))); }      // x 2^261 -> x 2^256
KDEV void canon_words(u32* out8, const Fq& a261) { u29_pack_canonical(out8, u29_mul(cut(a261), u29_const(Conv::PLAIN_ONE))); }   // -> x itself
// 1 / a in the 2^261 form: the division-step inverse of bn254_field.hip.h (constant time, ~22 K plain instructions) on the integer a 2^261, whose
// inverse is a^-1 2^-261; one product by 2^783 brings the form back. Once per pairing (easy part of the final exponentiation). The Fermat
// ladder it replaced (254 squarings + 110 products, then 60 with sliding windows: ~57 K instructions) is kept for the self-test.
static KNOINLINE Fq fq_inv261_fermat(const Fq a) {
  const U29 a1 = cut(a), a2 = u29_sqr(a1), a3 = u29_mul(a2, a1), a5 = u29_mul(a3, a2), a7 = u29_mul(a5, a2);
  auto bit = [](int i) { return (FQ_PM2[i >> 5] >> (i & 31)) & 1u; };
  U29 acc = u29_const(Q29::ONE);
  int i = 253;
#pragma unroll 1
  while (i >= 0) {
    if (!bit(i)) { acc = u29_sqr(acc); i--; continue; }
    int l = i >= 2 ? 3 : i + 1;
    while (!bit(i - l + 1)) l--;                      // the window ends in a set bit: its value is odd
    u32 v = 0;
    for (int k = 0; k < l; k++) { acc = u29_sqr(acc); v = (v << 1) | bit(i - k); }
    acc = u29_mul(acc, v == 1 ? a1 : v == 3 ? a3 : v == 5 ? a5 : a7);
    i -= l;
  }
  return pack(acc);
}
KDEV Fq fq_inv261(const Fq& a) {
  return pack(u29_mul(cut(fq_inv_safegcd_words(a)), u29_const(Conv::C783)));
}

// ---- Fq2d: saturated front (same interface as round 1) ---------------------------------------------------------------------------
KDEV Fq2d operator+(const Fq2d& a, const Fq2d& b) { return {a.v + b.v}; }
KDEV Fq2d operator-(const Fq2d& a, const Fq2d& b) { return {a.v - b.v}; }
// Negation. Round 1 saw a wrong value for the unary form (-theta, the fq_neg_asm stream) at ONE site of the fully inlined Miller loop and
// switched to 0 - theta. Round 2 re-ran the experiment on the restructured kernel (build with -DKEAKI_NEG_UNARY: every fq2_neg becomes
// the unary stream; bench_tools/dbg/neg_unary_experiment.sh): see DESIGN.md section 4.3 for the outcome. The binary form stays the default.
#ifdef KEAKI_NEG_UNARY
KDEV Fq2d fq2_neg(const Fq2d& a) { return {-a.v}; }
#else
KDEV Fq2d fq2_neg(const Fq2d& a) { return {fq_zero() - a.v}; }
#endif
KDEV Fq2d fq2_dbl(const Fq2d& a) { return {fq_dbl(a.v)}; }
KDEV Fq2d fq2d_zero() { return {fq_zero()}; }
KDEV Fq2d fq2d_one() { return {fq_select(lane_odd() != 0, fq_zero(), ONE)}; }
KDEV Fq2d fq2_conj(const Fq2d& a) { return {fp_cneg<FqParams>(a.v, lane_odd() != 0)}; }
// this lane's component of a constant (c0, c1). The lane's offset passes through an empty volatile statement so that the address is formed
// HERE: hipcc otherwise forms base + lane offset for every constant at kernel entry and keeps (spills) the 64-bit addresses until the use.
KDEV Fq2d fq2d_load(const Fq2* a) {
  u32 odd = lane_odd();
  asm volatile("" : "+v"(odd));
  return {reinterpret_cast<const Fq*>(a)[odd]};
}
KDEV Fq2d fq2_mul_fq(const Fq2d& a, const Fq& k) { return {fq_mul261(a.v, k)}; }
// a / 2 (any Montgomery radix: halving commutes with it): a canonical word value v < p is v / 2 when even, (v + p) / 2 when odd (v + p < 2^255).
// Branch-free: p masked by the low bit, one carry chain, one funnel shift per word -- ~30 instructions where the product by 1/2 took ~275.
KDEV Fq fq_half(const Fq& a) {
  const u32 m = 0u - (a.l[0] & 1u);
  u32 t[8];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (u64)a.l[i] + (FqParams::MOD[i] & m); t[i] = (u32)c; c >>= 32; }
  Fq r;
#pragma unroll
  for (int i = 0; i < 7; i++) r.l[i] = __builtin_amdgcn_alignbit(t[i + 1], t[i], 1);
  r.l[7] = t[7] >> 1;
  return r;
}
KDEV Fq2d fq2_half(const Fq2d& a) { return {fq_half(a.v)}; }

// ---- product operands in limb form ---------------------------------------------------------------------------------------------
// x side: this lane's component and the partner's (limbs <= 2^29 + 8)
struct XF { U29 s, o; };
// y side: the real component's limbs (in BOTH lanes) and the imaginary component's -- in even lanes NEGATED (K p - y1), so that the one
// stream  x.s y.y0 + x.o y.y1  is  a0 b0 - a1 b1  in the even lane and  a1 b0 + a0 b1  in the odd lane. Limbs <= 2^29 + 8.
struct YF { U29 y0, y1; };

KDEV XF x_of(const U29& own) { return {own, quad<0xB1>(own)}; }
// K: a multiple of p not below the bound of the imaginary component, limbs biased by 2^30 (Q29::K2 / K4 / K16 / K32)
KDEV YF y_of(const U29& own, const u32 (&K)[9]) {
  const bool odd = lane_odd() != 0;
  YF r;
  r.y0 = quad<0xA0>(own);
  const U29 y1 = quad<0xF5>(own);
  U29 n;
#pragma unroll
  for (int i = 0; i < 9; i++) n.l[i] = K[i] - y1.l[i];
  n = carry(n);
#pragma unroll
  for (int i = 0; i < 9; i++) r.y1.l[i] = odd ? y1.l[i] : n.l[i];
  return r;
}
// This lane's component of (9 + u)(a0 + a1 u) = (9 a0 - a1) + (9 a1 + a0) u from limbs: even 9 a + (K p - o), odd 9 a + o.
// a: EXACT limbs (< 2^29: a cut or a stream output), o: the partner's limbs, K >= bound of o. 9 a_i = ((a_i << 3) & MASK) + a_i with the
// three bits shifted out carried into the next limb, so no intermediate exceeds 32 bits. Result: limbs <= 2^29 + 8, value < (9 + K) p.
KDEV U29 xi_limbs(const U29& a, const U29& o, const u32 (&K)[9]) {
  const bool odd = lane_odd() != 0;
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const u32 add = odd ? o.l[i] : K[i] - o.l[i];
    const u32 lo8 = (i < 8) ? ((a.l[i] << 3) & Q29::MASK) : (a.l[i] << 3);
    t.l[i] = lo8 + a.l[i] + add + (i ? (a.l[i - 1] >> 26) : 0u);
  }
  return carry(t);
}

// one Fq2 product, two, three: (sum of the Fq2 products)'s component of this lane, exact limbs, value < 2p for the operand bounds used here
KDEV U29 dot1(const XF& x0, const YF& y0) { return u29_mul2(x0.s, y0.y0, x0.o, y0.y1); }
KDEV U29 dot2(const XF& x0, const YF& y0, const XF& x1, const YF& y1) {
  U29 r;
  u29_dot4_asm(r.l, x0.s.l, y0.y0.l, x0.o.l, y0.y1.l, x1.s.l, y1.y0.l, x1.o.l, y1.y1.l);
  return r;
}
KDEV U29 dot3(const XF& x0, const YF& y0, const XF& x1, const YF& y1, const XF& x2, const YF& y2) {
  U29 r;
  u29_dot6_asm(r.l, x0.s.l, y0.y0.l, x0.o.l, y0.y1.l, x1.s.l, y1.y0.l, x1.o.l, y1.y1.l, x2.s.l, y2.y0.l, x2.o.l, y2.y1.l);
  return r;
}

// ---- single Fq2 product / square (line functions, Frobenius, inversions) -------------------------------------------------------
// (a0 + a1 u)(b0 + b1 u): even lane a0 b0 + a1 (2p - b1), odd lane a1 b0 + a0 b1: one dual stream, one reduction. Bound (1 + 2)/169 + 1.
static KNOINLINE Fq2d fq2d_mul(const Fq2d a, const Fq2d b) {
  return {pack(dot1(x_of(cut(a.v)), y_of(cut(b.v), Q29::K2)))};
}
// (a0 + a1 u)^2: even lane (a0 + a1)(a0 - a1), odd lane 2 a0 a1 -- one product per lane. x = a0 + a1 (limbs <= 2^30) | a0;  y = a0 - a1 + 2p | 2 a1
static KNOINLINE Fq2d fq2d_sqr(const Fq2d a) {
  const bool odd = lane_odd() != 0;
  const U29 A = cut(a.v), O = quad<0xB1>(A);
  U29 x, y;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x.l[i] = odd ? O.l[i] : A.l[i] + O.l[i];
    y.l[i] = odd ? 2u * A.l[i] : A.l[i] - O.l[i] + Q29::K2[i];
  }
  return {pack(u29_mul(x, carry(y)))};       // u29_mul: one side up to 2^30 + 16, the other carried. Bound (2 * 3)/169 + 1.
}
KDEV Fq2d operator*(const Fq2d& a, const Fq2d& b) { return fq2d_mul(a, b); }
KDEV Fq2d fq2_sqr(const Fq2d& a) { return fq2d_sqr(a); }
// (9 + u) a in the saturated words (only where a single value is needed outside a product)
// This lane's component of (9 + u)(a0 + a1 u): even 9 a0 - a1, odd 9 a1 + a0, as ONE multi-word expression with ONE reduction:
// r = 9 own + s with s = p - partner (even) | partner (odd), r < 10 p < 2^258; quotient estimate q = floor(floor(r / 2^227) * 42 / 2^32)
// (42 < 2^32 / (p / 2^227 + 1): q <= floor(r / p), and >= floor(r / p) - 1 because 42 p / 2^259 = 0.9937 and r / p < 10), so r - q p lies
// in [0, 2p) and one conditional subtraction makes it canonical. ~90 instructions where three doublings, two additions and a conditional
// negation of the saturated arithmetic took ~170 (390 uses per pairing).
KDEV Fq2d fq2_mul_xi(const Fq2d& a) {
  const bool even = lane_odd() == 0;
  const Fq o = fq_partner(a.v);
  u32 s[8], t[9], r[9], cy;
  // s = p - partner (even) | partner (odd): one borrow chain, one select per word
  cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { const u32 d = __builtin_subc(FqParams::MOD[i], o.l[i], cy, &cy); s[i] = even ? d : o.l[i]; }
  // r = (own << 3) + own + s: funnel shifts and two carry chains over nine words
  t[0] = a.v.l[0] << 3;
#pragma unroll
  for (int i = 1; i < 8; i++) t[i] = __builtin_amdgcn_alignbit(a.v.l[i], a.v.l[i - 1], 29);
  t[8] = a.v.l[7] >> 29;
  cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = __builtin_addc(t[i], a.v.l[i], cy, &cy);
  r[8] = t[8] + cy;
  cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = __builtin_addc(r[i], s[i], cy, &cy);
  r[8] += cy;
  const u32 q = __umulhi(__builtin_amdgcn_alignbit(r[8], r[7], 3), 42u);
  // r - q p: the eight word products stand alone (no carry between the multiplications), one carry chain joins them, one borrow chain subtracts
  u32 lo[8], hi[8], w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { const u64 pr = (u64)FqParams::MOD[i] * q; lo[i] = (u32)pr; hi[i] = (u32)(pr >> 32); }
  cy = 0;
  w[0] = lo[0];
#pragma unroll
  for (int i = 1; i < 8; i++) w[i] = __builtin_addc(lo[i], hi[i - 1], cy, &cy);
  cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = __builtin_subc(r[i], w[i], cy, &cy);
  Fq2d out;
  fq_cond_sub_p_asm(out.v.l, t);
  return out;
}
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2); the norm's inverse is computed redundantly in both lanes
KDEV Fq2d fq2_inv(const Fq2d& a) {
  Fq sq = fq_sqr261(a.v);
  Fq n = sq + fq_partner(sq);
  Fq ni = fq_inv261(n);
  return {fp_cneg<FqParams>(fq_mul261(a.v, ni), lane_odd() != 0)};
}

#define M2(a, b) ((a) * (b))
#define S2(a) fq2_sqr((a))

// ---- Fq6 ------------------------------------------------------------------------------------------------------------------------
KDEV Fq6 operator+(const Fq6& a, const Fq6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
KDEV Fq6 operator-(const Fq6& a, const Fq6& b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
KDEV Fq6 fq6_neg(const Fq6& a) { return {fq2_neg(a.c0), fq2_neg(a.c1), fq2_neg(a.c2)}; }
KDEV Fq6 fq6_mul_v(const Fq6& a) { return {fq2_mul_xi(a.c2), a.c0, a.c1}; }
KDEV Fq6 fq6_zero() { return {fq2d_zero(), fq2d_zero(), fq2d_zero()}; }

// (a0 + a1 v + a2 v^2)(b0 + b1 v + b2 v^2), v^3 = xi, written out:
//   c0 = a0 b0 + (xi a1) b2 + (xi a2) b1      c1 = a0 b1 + a1 b0 + (xi a2) b2      c2 = a0 b2 + a1 b1 + a2 b0
// three streams of six Fq products. Bounds: plain term 1*1 + 1*2 = 3, xi term 11 + 22 = 33: c0 < (69/169 + 1) p.
static KTOWER void fq6_mul(Fq6* r, const Fq6* a, const Fq6* b) {
  const XF x0 = x_of(cut(a->c0.v)), x1 = x_of(cut(a->c1.v)), x2 = x_of(cut(a->c2.v));
  const XF xx1 = x_of(xi_limbs(x1.s, x1.o, Q29::K2)), xx2 = x_of(xi_limbs(x2.s, x2.o, Q29::K2));
  const YF y0 = y_of(cut(b->c0.v), Q29::K2), y1 = y_of(cut(b->c1.v), Q29::K2), y2 = y_of(cut(b->c2.v), Q29::K2);
  const Fq c0 = pack(dot3(x0, y0, xx1, y2, xx2, y1));
  const Fq c1 = pack(dot3(x0, y1, x1, y0, xx2, y2));
  const Fq c2 = pack(dot3(x0, y2, x1, y1, x2, y0));
  r->c0.v = c0; r->c1.v = c1; r->c2.v = c2;
}
static KTOWER void fq6_inv(Fq6* r, const Fq6* a) {
  Fq2d t0 = S2(a->c0) - fq2_mul_xi(M2(a->c1, a->c2));
  Fq2d t1 = fq2_mul_xi(S2(a->c2)) - M2(a->c0, a->c1);
  Fq2d t2 = S2(a->c1) - M2(a->c0, a->c2);
  Fq2d n = M2(a->c0, t0) + fq2_mul_xi(M2(a->c2, t1) + M2(a->c1, t2));
  Fq2d ni = fq2_inv(n);
  r->c0 = M2(t0, ni); r->c1 = M2(t1, ni); r->c2 = M2(t2, ni);
}

// ---- Fq12 -----------------------------------------------------------------------------------------------------------------------
KDEV void fq12_set_one(Fq12* f) {
  f->c0 = fq6_zero(); f->c1 = fq6_zero();
  f->c0.c0 = fq2d_one();
}
KDEV void fq12_conj(Fq12* r, const Fq12* a) { r->c0 = a->c0; r->c1 = fq6_neg(a->c1); }
// Karatsuba over Fq6: three Fq6 products. Straight-line on purpose: a loop over one instance of fq6_mul made hipcc pick the k-th operand
// through a scratch array (48 stores + 48 loads per iteration: 40 K scratch instructions per pairing, profiles/r02_pairing_pmc_v1.csv).
static KTOWER void fq12_mul(Fq12* r, const Fq12* a, const Fq12* b) {
  Fq6 t0, t1, m;
  {
    const Fq6 s0 = a->c0 + a->c1, s1 = b->c0 + b->c1;
    fq6_mul(&m, &s0, &s1);
  }
  fq6_mul(&t0, &a->c0, &b->c0);
  fq6_mul(&t1, &a->c1, &b->c1);
  r->c1 = m - t0 - t1;
  r->c0 = t0 + fq6_mul_v(t1);
}
// The same with the second operand in MEMORY: `ld(h)` returns half h (c0 / c1) of b, this lane's components. b is read three times (both
// halves for the Karatsuba sum, then one half per product) instead of being held in 48 registers next to the accumulator -- the difference
// between spilling and not spilling in the kernels that multiply an accumulator by a table entry or a slot.
KDEV void park_fq6(uint4* park, int idx, const Fq6& v) { park_fq(park, idx, v.c0.v); park_fq(park, idx + 1, v.c1.v); park_fq(park, idx + 2, v.c2.v); }
KDEV Fq6 unpark_fq6(const uint4* park, int idx) { return {{unpark_fq(park, idx)}, {unpark_fq(park, idx + 1)}, {unpark_fq(park, idx + 2)}}; }
// `park` (may be null: then everything stays in registers): six Fq of LDS per lane (indices 0..5) for the first two partial products
template <bool PARK = false, class LoadHalf>
static KTOWER void fq12_mul_ld(Fq12* r, const Fq12* a, LoadHalf ld, uint4* park = nullptr) {
  Fq6 t0, t1, m;
  {
    const Fq6 b0 = ld(0), b1 = ld(1);
    const Fq6 s0 = a->c0 + a->c1, s1 = b0 + b1;
    fq6_mul(&m, &s0, &s1);
  }
  if constexpr (PARK) park_fq6(park, 0, m);
  asm volatile("" ::: "memory");          // the halves are loaded again, not kept
  {
    const Fq6 b0 = ld(0);
    fq6_mul(&t0, &a->c0, &b0);
  }
  if constexpr (PARK) park_fq6(park, 3, t0);
  asm volatile("" ::: "memory");
  {
    const Fq6 b1 = ld(1);
    fq6_mul(&t1, &a->c1, &b1);
  }
  if constexpr (PARK) { m = unpark_fq6(park, 0); t0 = unpark_fq6(park, 3); }
  r->c1 = m - t0 - t1;
  r->c0 = t0 + fq6_mul_v(t1);
}
// complex squaring: a0 a1 and (a0 + a1)(a0 + v a1)
template <bool PARK = false>
static KTOWER void fq12_sqr(Fq12* r, const Fq12* a, uint4* park = nullptr, int pidx = 0) {
  Fq6 ab, t;
  {
    const Fq6 s0 = a->c0 + a->c1, s1 = a->c0 + fq6_mul_v(a->c1);
    fq6_mul(&t, &s0, &s1);
  }
  if constexpr (PARK) park_fq6(park, pidx, t);
  fq6_mul(&ab, &a->c0, &a->c1);
  if constexpr (PARK) t = unpark_fq6(park, pidx);
  r->c0 = t - ab - fq6_mul_v(ab);
  r->c1 = ab + ab;
}
template <bool PARK = false>
static KTOWER void fq12_inv(Fq12* r, const Fq12* a, uint4* park = nullptr) {
  if constexpr (!PARK) {
    Fq6 n0, n1, ni, r0, r1;
    fq6_mul(&n0, &a->c0, &a->c0);
    fq6_mul(&n1, &a->c1, &a->c1);
    Fq6 n = n0 - fq6_mul_v(n1);
    fq6_inv(&ni, &n);
    fq6_mul(&r0, &a->c0, &ni);
    fq6_mul(&r1, &a->c1, &ni);
    r->c0 = r0;
    r->c1 = fq6_neg(r1);
  } else {
  // with LDS: a waits in indices 0..5, its halves visit registers one at a time; n0 in 6..8
#define PARK_FENCE() asm volatile("" ::: "memory")      /* keeps the scheduler from hoisting an LDS read (and with it a live range) over a big block */
  park_fq6(park, 0, a->c0); park_fq6(park, 3, a->c1);
  Fq6 n;
  {
    Fq6 n0, n1;
    PARK_FENCE();
    { const Fq6 a0 = unpark_fq6(park, 0); fq6_mul(&n0, &a0, &a0); }
    park_fq6(park, 6, n0);
    PARK_FENCE();
    { const Fq6 a1 = unpark_fq6(park, 3); fq6_mul(&n1, &a1, &a1); }
    PARK_FENCE();
    n0 = unpark_fq6(park, 6);
    n = n0 - fq6_mul_v(n1);
  }
  Fq6 r0, r1;
  {
    Fq6 ni;
    fq6_inv(&ni, &n);                     // a chain of function calls
    park_fq6(park, 6, ni);                // read back per product: otherwise its operand forms (54 limbs + the xi forms) are shared by the
  }                                       // two products and stay alive across the first one
  PARK_FENCE();
  { const Fq6 a0 = unpark_fq6(park, 0), nb = unpark_fq6(park, 6); fq6_mul(&r0, &a0, &nb); }
  park_fq6(park, 0, r0);
  PARK_FENCE();
  { const Fq6 a1 = unpark_fq6(park, 3), nb = unpark_fq6(park, 6); fq6_mul(&r1, &a1, &nb); }
  PARK_FENCE();
  r->c0 = unpark_fq6(park, 0);
  r->c1 = fq6_neg(r1);
  }
}
// the same with the element parked in LDS (indices 0..5) while the six Fq2 products (function calls) run: nothing big is alive across a call
static KTOWER void fq12_frob_parked(Fq12* r, const Fq12* a, int k, uint4* park) {
  const Fq2d* c = reinterpret_cast<const Fq2d*>(a);          // memory order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2
#pragma unroll
  for (int i = 0; i < 6; i++) park_fq(park, i, c[i].v);
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    // memory slot i holds the coefficient of w^e with e = 2 * (i % 3) + i / 3
    const int e = 2 * (i % 3) + i / 3;
    Fq2d t = {unpark_fq(park, i)};
    if (k & 1) t = fq2_conj(t);
    park_fq(park, i, M2(t, fq2d_load(&FROB_W[k][e])).v);
  }
  Fq2d* o = reinterpret_cast<Fq2d*>(r);
#pragma unroll
  for (int i = 0; i < 6; i++) o[i].v = unpark_fq(park, i);
}
static KTOWER void fq12_frob(Fq12* r, const Fq12* a, int k) {
  Fq2d c[6] = {a->c0.c0, a->c1.c0, a->c0.c1, a->c1.c1, a->c0.c2, a->c1.c2};
  Fq2d o[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    Fq2d t = (k & 1) ? fq2_conj(c[i]) : c[i];
    o[i] = M2(t, fq2d_load(&FROB_W[k][i]));
  }
  r->c0.c0 = o[0]; r->c1.c0 = o[1]; r->c0.c1 = o[2]; r->c1.c1 = o[3]; r->c0.c2 = o[4]; r->c1.c2 = o[5];
}

// Granger-Scott squaring on the cyclotomic subgroup: three Fq4 squarings (x + y s)^2 = (x^2 + xi y^2) + 2 x y s, each as
//   t0 = x^2 + (xi y) y   one stream of THREE Fq products: x^2 costs a lane one product -- even lane (x0 + x1)(x0 - x1 + 2p), odd lane
//                         x0 (2 x1) -- where the general Fq2 product costs it two      (bound 6 + 33)
//   t1 = (2x) y           one dual stream                                              (bound 2 + 4)
// then the linear recombination in the saturated words.
KDEV void fq4_sqr(Fq2d* t0, Fq2d* t1, const Fq2d& x, const Fq2d& y) {
  const bool odd = lane_odd() != 0;
  const XF xx = x_of(cut(x.v)), xy = x_of(cut(y.v));
  const XF xxy = x_of(xi_limbs(xy.s, xy.o, Q29::K2));
  const YF yy = y_of(xy.s, Q29::K2);
  U29 sx, sy;
  XF x2;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    sx.l[i] = odd ? xx.o.l[i] : xx.s.l[i] + xx.o.l[i];                            // limbs < 2^30: one side of a product may be that wide
    sy.l[i] = odd ? 2u * xx.s.l[i] : xx.s.l[i] - xx.o.l[i] + Q29::K2[i];
    x2.s.l[i] = 2u * xx.s.l[i]; x2.o.l[i] = 2u * xx.o.l[i];                       // limbs < 2^30: within the dual stream's budget
  }
  sy = carry(sy);
  U29 r;
  u29_dot3_asm(r.l, sx.l, sy.l, xxy.s.l, yy.y0.l, xxy.o.l, yy.y1.l);
  t0->v = pack(r);
  t1->v = pack(dot1(x2, yy));
}
template <bool PARK = false>
static KTOWER void fq12_cyc_sqr(Fq12* r, const Fq12* a, uint4* park = nullptr) {
  Fq2d r0 = a->c0.c0, r4 = a->c0.c1, r3 = a->c0.c2, r2 = a->c1.c0, r1 = a->c1.c1, r5 = a->c1.c2;
  Fq2d t0, t1, t2, t3, t4, t5;
  if constexpr (PARK) { park_fq(park, 0, r0.v); park_fq(park, 1, r1.v); park_fq(park, 2, r2.v); park_fq(park, 3, r3.v); }
  fq4_sqr(&t4, &t5, r4, r5);
  if constexpr (PARK) { park_fq(park, 4, r4.v); park_fq(park, 5, r5.v); r2.v = unpark_fq(park, 2); r3.v = unpark_fq(park, 3); }
  fq4_sqr(&t2, &t3, r2, r3);
  if constexpr (PARK) { r0.v = unpark_fq(park, 0); r1.v = unpark_fq(park, 1); }
  fq4_sqr(&t0, &t1, r0, r1);
  if constexpr (PARK) { r2.v = unpark_fq(park, 2); r3.v = unpark_fq(park, 3); r4.v = unpark_fq(park, 4); r5.v = unpark_fq(park, 5); }
  Fq2d x5 = fq2_mul_xi(t5);
  r->c0.c0 = fq2_dbl(t0 - r0) + t0;
  r->c1.c1 = fq2_dbl(t1 + r1) + t1;
  r->c1.c0 = fq2_dbl(x5 + r2) + x5;
  r->c0.c2 = fq2_dbl(t4 - r3) + t4;
  r->c0.c1 = fq2_dbl(t2 - r4) + t2;
  r->c1.c2 = fq2_dbl(t3 + r5) + t3;
}

// ---- the sparse line product: f *= c0 + (d0 + d1 v) w ---------------------------------------------------------------------------
// f = (a0 + a1 v + a2 v^2) + (e0 + e1 v + e2 v^2) w. With w^2 = v, v^3 = xi:
//   r0.c0 = a0 c0 + e1 (xi d1) + e2 (xi d0)     r0.c1 = a1 c0 + e0 d0 + e2 (xi d1)     r0.c2 = a2 c0 + e0 d1 + e1 d0
//   r1.c0 = e0 c0 + a0 d0 + a2 (xi d1)          r1.c1 = e1 c0 + a0 d1 + a1 d0          r1.c2 = e2 c0 + a1 d1 + a2 d0
// six streams of six Fq products; the multiplications by xi sit on the line's side (two per line instead of one per product).
// The line's coefficients arrive as LIMBS of this lane's component: c0, d0 exact and < 2p (outputs of the products by P's coordinates),
// d1 exact and < p. Bounds: f's coefficients 1; terms c0: 2 + 2, d0: 2 + 2, d1: 1 + 2, xi d1 (< 11p, negated 16p): 27, xi d0 (< 22p, 32p): 54.
template <bool PARK = false>
static KTOWER void fq12_mul_by_034_limbs(Fq12* f, U29 c0, U29 d0, const U29& d1, bool fence_c0_d0, uint4* park = nullptr, int pidx = 0) {
  if (fence_c0_d0) { fence9(c0); fence9(d0); }
  const U29 c0o = quad<0xB1>(c0), d0o = quad<0xB1>(d0), d1o = quad<0xB1>(d1);
  (void)c0o;
  const YF yc0 = y_of(c0, Q29::K2), yd0 = y_of(d0, Q29::K2), yd1 = y_of(d1, Q29::K2);
  const YF yxd0 = y_of(xi_limbs(d0, d0o, Q29::K4), Q29::K32), yxd1 = y_of(xi_limbs(d1, d1o, Q29::K2), Q29::K16);
  const XF a0 = x_of(cut(f->c0.c0.v)), a1 = x_of(cut(f->c0.c1.v)), a2 = x_of(cut(f->c0.c2.v));
  const XF e0 = x_of(cut(f->c1.c0.v)), e1 = x_of(cut(f->c1.c1.v)), e2 = x_of(cut(f->c1.c2.v));
  const Fq r00 = pack(dot3(a0, yc0, e1, yxd1, e2, yxd0));
  const Fq r01 = pack(dot3(a1, yc0, e0, yd0, e2, yxd1));
  const Fq r02 = pack(dot3(a2, yc0, e0, yd1, e1, yd0));
  if constexpr (PARK) { park_fq(park, pidx, r00); park_fq(park, pidx + 1, r01); park_fq(park, pidx + 2, r02); }
  const Fq r10 = pack(dot3(e0, yc0, a0, yd0, a2, yxd1));
  const Fq r11 = pack(dot3(e1, yc0, a0, yd1, a1, yd0));
  const Fq r12 = pack(dot3(e2, yc0, a1, yd1, a2, yd0));
  if constexpr (PARK) { f->c0.c0.v = unpark_fq(park, pidx); f->c0.c1.v = unpark_fq(park, pidx + 1); f->c0.c2.v = unpark_fq(park, pidx + 2); }
  else { f->c0.c0.v = r00; f->c0.c1.v = r01; f->c0.c2.v = r02; }
  f->c1.c0.v = r10; f->c1.c1.v = r11; f->c1.c2.v = r12;
}

}  // namespace p261
}  // namespace bn254
