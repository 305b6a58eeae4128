// Batched BN254 optimal-ate pairing on gfx950: one lane PAIR per pairing, ONE kernel, no function call above an Fq2 product.
//
// Replaces `E::pairing(p, q)` (reference src/kem.rs:30,58; src/kzg.rs:148; ark-ec 0.4.2 models/bn: G2Prepared line coefficients +
// multi_miller_loop + final_exponentiation) and `serialize_uncompressed` of the GT element (src/kem.rs:32,61).
//
// The reduced pairing value is independent of the Miller-loop addition chain and of subfield scalings of the line functions, so the
// kernel is free to (a) compute the lines on the fly (or read a table when Q is constant), (b) use the proper NAF of 6z + 2 (22
// additions instead of arkworks' 26) and a width-4 NAF of z in the hard part. The final exponent is arkworks' exactly:
// (p^12 - 1)/r * 2z(6z^2 + 3z + 1). The tower arithmetic is pair261.hip.h (2^261 Montgomery form, lane pairs, multi-product streams).
//
// Structure (round 2): the round-1 kernel inlined ~45 Fq12 products and kept up to five Fq12 values alive in the hard part: 459 KB of
// code, 630 spilled VGPRs. Now
//   * the Miller loop is a flat loop over the 88 line steps of MILLER_STEPS with ONE instance of the Fq12 squaring and ONE of the
//     sparse line product;
//   * the final exponentiation is a PROGRAM (FE_PROG, 292 one-byte ops, generated and checked against the big-int oracle on the CPU:
//     tests/test_pair261_model.py) run by an accumulator machine: one Fq12 accumulator in registers, every other value in a per-item
//     slot in HBM (12 slots x 384 B, structure-of-arrays so that a wave's loads are contiguous), ONE instance each of the Fq12
//     product, the cyclotomic squaring, the Frobenius maps and the inversion. Control flow is wave-uniform (the program counter is scalar).
// Slot traffic: ~60 loads / stores of 384 B per pairing (23 KB) against ~2.6 M instructions: noise, and mostly served by the LLC.
#pragma once
#include "pair261.hip.h"

namespace bn254 {
using namespace p261;

// ---- line functions on the twist, homogeneous projective (same formulas as ark-ec bn/g2.rs) ----
struct G2Hom { Fq2d x, y, z; };
struct Line { Fq2d c0, c1, c2; };  // evaluated as c0 * P.y + (c1 * P.x + c2 v) w

static KTOWER void line_double(G2Hom* r, Line* l) {
  Fq2d a = fq2_half(M2(r->x, r->y));
  Fq2d b = S2(r->y), c = S2(r->z);
  Fq2d e = M2(fq2d_load(&p261::G2_B), fq2_dbl(c) + c);
  Fq2d f = fq2_dbl(e) + e;
  Fq2d g = fq2_half(b + f);
  Fq2d h = S2(r->y + r->z) - (b + c);
  Fq2d i = e - b;
  Fq2d j = S2(r->x);
  Fq2d e2 = S2(e);
  r->x = M2(a, b - f);
  r->y = S2(g) - (fq2_dbl(e2) + e2);
  r->z = M2(b, h);
  l->c0 = fq2_neg(h); l->c1 = fq2_dbl(j) + j; l->c2 = i;
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
  l->c0 = lam; l->c1 = fq2_neg(theta); l->c2 = M2(theta, *qx) - M2(lam, *qy);
}
// f *= l(P): the two products by P's coordinates leave the streams as limbs and go straight into the line product
KDEV void ell(Fq12* f, const Line* l, const U29& px, const U29& py, uint4* park, int pidx) {
  const U29 c0 = u29_mul(cut(l->c0.v), py), d0 = u29_mul(cut(l->c1.v), px);
  fq12_mul_by_034_limbs<true>(f, c0, d0, cut(l->c2.v), true, park, pidx);
}

// Line table of a fixed Q (ark-ec's G2Prepared), per lane parity: lines[li * 2 + parity], li = index into MILLER_STEPS. 2^261 form.
constexpr int MILLER_MAX_LINES = 96;
static_assert(MILLER_NSTEPS <= MILLER_MAX_LINES, "line table too small");

// lines == nullptr: compute the lines on the fly from Q (`qw`: its four coordinates in the 2^256 form, this lane's components at [par] and [2 + par]).
// P = (px, py) in the 2^261 form.
// Register choreography through LDS (`park`, Fq indices): while f is squared and multiplied by the line, the running point T waits in
// 0..2 and P in 6..7, and the two big operations use 3..5 for their own temporaries; while the line function runs (a chain of CALLS of the Fq2
// product, around each of which every live caller-saved register would be saved), f waits in 0..5 and T visits registers. Q is only
// needed by the 24 addition steps: it is read again from global memory there (two conversions, 0.5 % of the loop).
static KTOWER void miller_loop(Fq12* f, const Fq& px, const Fq& py, const Fq* __restrict__ qw, const Line* __restrict__ lines, uint4* park) {
  const u32 par = lane_odd();
  fq12_set_one(f);
  if (!lines) {
    park_fq(park, 0, to261(qw[par])); park_fq(park, 1, to261(qw[2 + par])); park_fq(park, 2, fq2d_one().v);     // T = (Q.x, Q.y, 1)
  }
  park_fq(park, 6, px); park_fq(park, 7, py);
#pragma unroll 1
  for (int li = 0; li < MILLER_NSTEPS; li++) {
    const int st = MILLER_STEPS[li];
    if (st == 1) fq12_sqr<true>(f, f, park, 3);
    Line l;
    if (lines) {
      l = lines[li * 2 + par];
    } else {
      G2Hom r = {{unpark_fq(park, 0)}, {unpark_fq(park, 1)}, {unpark_fq(park, 2)}};
      {
        const Fq2d* c = reinterpret_cast<const Fq2d*>(f);
#pragma unroll
        for (int k = 0; k < 6; k++) park_fq(park, k, c[k].v);                              // f out of the way of the calls
      }
      if (st <= 1) {
        line_double(&r, &l);
      } else {
        // Q, -Q, pi(Q), -pi^2(Q)
        Fq2d ax = {to261(qw[par])}, ay = {to261(qw[2 + par])};
        if (st == 3) ay = fq2_neg(ay);
        if (st >= 4) {
          ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y));
          if (st == 5) { ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = fq2_neg(M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y))); }
        }
        line_add(&r, &ax, &ay, &l);
      }
      {
        Fq2d* c = reinterpret_cast<Fq2d*>(f);
#pragma unroll
        for (int k = 0; k < 6; k++) c[k].v = unpark_fq(park, k);
      }
      park_fq(park, 0, r.x.v); park_fq(park, 1, r.y.v); park_fq(park, 2, r.z.v);
    }
    ell(f, &l, cut(unpark_fq(park, 6)), cut(unpark_fq(park, 7)), park, 3);
  }
}
// the line sequence alone (k_g2_prepare)
static KTOWER void miller_lines(const Fq2d* qx, const Fq2d* qy, Line* __restrict__ lines_out) {
  const u32 par = lane_odd();
  G2Hom r = {*qx, *qy, fq2d_one()};
#pragma unroll 1
  for (int li = 0; li < MILLER_NSTEPS; li++) {
    const int st = MILLER_STEPS[li];
    Line l;
    if (st <= 1) {
      line_double(&r, &l);
    } else {
      Fq2d ax = *qx, ay = *qy;
      if (st == 3) ay = fq2_neg(ay);
      if (st >= 4) {
        ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y));
        if (st == 5) { ax = M2(fq2_conj(ax), fq2d_load(&p261::TWIST_MUL_BY_Q_X)); ay = fq2_neg(M2(fq2_conj(ay), fq2d_load(&p261::TWIST_MUL_BY_Q_Y))); }
      }
      line_add(&r, &ax, &ay, &l);
    }
    lines_out[li * 2 + par] = l;
  }
}

// ---- per-item slots in HBM, structure of arrays: component c (= 2k + parity) of slot s of item i at ws[(s * 12 + c) * ws_n + i] ----
KDEV void slot_load(Fq12* f, const Fq* __restrict__ ws, size_t ws_n, u32 slot, u32 i) {
  Fq2d* c = reinterpret_cast<Fq2d*>(f);
  const u32 par = lane_odd();
#pragma unroll
  for (int k = 0; k < 6; k++) c[k].v = ws[(size_t)(slot * 12 + 2 * k + par) * ws_n + i];
}
KDEV void slot_store(Fq* __restrict__ ws, size_t ws_n, u32 slot, u32 i, const Fq12* f) {
  const Fq2d* c = reinterpret_cast<const Fq2d*>(f);
  const u32 par = lane_odd();
#pragma unroll
  for (int k = 0; k < 6; k++) ws[(size_t)(slot * 12 + 2 * k + par) * ws_n + i] = c[k].v;
}

// the final exponentiation: FE_PROG on the accumulator `acc` (= the Miller loop's output on entry, the GT element on exit)
// The lane's item index waits in LDS (chunk PARK_CHUNKS of `park`) and is read where it is needed: a register that lives from the kernel's
// entry to its output is the one value hipcc spills to scratch memory.
KDEV void item_park(uint4* park, u32 i) { reinterpret_cast<u32*>(park + PARK_CHUNKS * 64)[threadIdx.x] = i; }
KDEV u32 item_unpark(const uint4* park) {
  asm volatile("" ::: "memory");
  return reinterpret_cast<const u32*>(park + PARK_CHUNKS * 64)[threadIdx.x];
}
static KTOWER void fe_run(Fq12* acc, Fq* __restrict__ ws, size_t ws_n, uint4* park) {
#pragma unroll 1
  for (int pc = 0; pc < FE_NOPS; pc++) {
    const u32 op = FE_PROG[pc], code = op & 15u, s = op >> 4;
    const u32 item = item_unpark(park);
    if (code == 0) {
      slot_load(acc, ws, ws_n, s, item);
    } else if (code == 1) {
      slot_store(ws, ws_n, s, item, acc);
    } else if (code == 2) {
      fq12_cyc_sqr<true>(acc, acc, park);
    } else if (code == 3 || code == 4) {
      const u32 par = lane_odd();
      const bool cj = code == 4;                  // multiply by the conjugate: the second half negated
      fq12_mul_ld<true>(acc, acc, [&](int h) {
        Fq6 x;
        Fq2d* c = reinterpret_cast<Fq2d*>(&x);
#pragma unroll
        for (int k = 0; k < 3; k++) c[k].v = ws[(size_t)(s * 12 + 6 * h + 2 * k + par) * ws_n + item];
        if (h == 1 && cj) x = fq6_neg(x);
        return x;
      }, park);
    } else if (code == 5) {
      fq12_conj(acc, acc);
    } else if (code == 6) {
      fq12_frob_parked(acc, acc, (int)s, park);
    } else {
      fq12_inv<true>(acc, acc, park);
    }
  }
}

// GT -> 384 canonical little-endian bytes in ark-serialize order (c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1):
// Fq2 coefficient k of the element (k = 0..5 in memory order) fills the 32-byte slots 2k (even lane) and 2k+1 (odd lane).
// The loop runs over LDS (`park`, indices 0..5): a dynamic index into the REGISTER copy of f would force the whole element into scratch memory.
KDEV void gt_serialize(u32* out96, const Fq12* f, uint4* park) {
  const Fq2d* c = reinterpret_cast<const Fq2d*>(f);
  const u32 par = lane_odd();
#pragma unroll
  for (int i = 0; i < 6; i++) park_fq(park, i, c[i].v);
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    u32 w[8];
    canon_words(w, unpark_fq(park, i));
#pragma unroll
    for (int j = 0; j < 8; j++) out96[8 * (2 * i + par) + j] = w[j];
  }
}

// What one launch of k_pairing does (bit mask)
enum : u32 {
  PAIR_MILLER = 1,        // run the Miller loop on (P, Q) / (P, fixed lines); else the accumulator is loaded from f_in (2^256 form, 12 Fq per item)
  PAIR_FINAL_EXP = 2,     // run the final exponentiation
  PAIR_OUT_BYTES = 4,     // out = 384 serialised bytes per item (identity in either slot -> GT one)
  PAIR_OUT_RAW256 = 8,    // out = 12 Fq per item, 2^256 Montgomery form, slot 2k + parity (test hook: Miller loop alone)
  PAIR_OUT_RAW261 = 16,   // out = 12 Fq per item, 2^261 form (feeds the GT window tables)
};
struct PairArgs {
  const G1Aff* ps;          // P_(i * p_stride), affine, 2^256 form ((0, 0) = identity); p_stride 0: one point for every item
  u32 p_stride;
  const G2Aff* qs;          // Q_(i * q_stride); ignored when fixed_lines
  int q_stride;
  u32 n;
  const Line* fixed_lines;  // tabulated lines of the second slot: table (i * lines_stride) for item i
  u32 lines_stride;
  const Fq* f_in;
  Fq* ws;                   // FE_NSLOTS * 12 * ws_n Fq
  size_t ws_n;
  void* out;
  u32 mode;
};

// TWO lanes per item. Tail lanes redo the last item: all 64 lanes must stay active for the DPP exchanges.
static __global__ void __launch_bounds__(64, 2) k_pairing(PairArgs a) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 item = t >> 1;
  const bool live = item < a.n;
  const u32 i = live ? item : (a.n - 1);
  __shared__ uint4 park[PARK_CHUNKS * 64 + 16];     // + 64 words: the lanes' item indices
  item_park(park, i);
  Fq12 f;
  bool ident = false;
  if (a.mode & PAIR_MILLER) {
    const G1Aff p = a.ps[(size_t)i * a.p_stride];
    const Fq* qw = nullptr;
    u32 qz = 0;
    if (!a.fixed_lines) {
      qw = reinterpret_cast<const Fq*>(a.qs + (size_t)i * a.q_stride);          // (x.c0, x.c1, y.c0, y.c1)
      // identity in the second slot: all four components zero; both lanes of the pair must agree
      qz = (fq_is_zero(qw[lane_odd()]) && fq_is_zero(qw[2 + lane_odd()])) ? 1u : 0u;
      qz &= (u32)__builtin_amdgcn_update_dpp(0, (int)qz, 0xB1, 0xF, 0xF, true);
    }
    ident = aff_is_inf(p) || qz != 0;
    // wave-uniform control flow: identity items run the same arithmetic (total on zeros) and discard it
    miller_loop(&f, to261(p.x), to261(p.y), qw, a.fixed_lines ? a.fixed_lines + (size_t)i * a.lines_stride : nullptr, park);
  } else {
    Fq2d* c = reinterpret_cast<Fq2d*>(&f);
#pragma unroll 1
    for (int k = 0; k < 6; k++) park_fq(park, k, to261(a.f_in[(size_t)12 * item_unpark(park) + 2 * k + lane_odd()]));
#pragma unroll
    for (int k = 0; k < 6; k++) c[k].v = unpark_fq(park, k);
  }
  if (a.mode & PAIR_FINAL_EXP) fe_run(&f, a.ws, a.ws_n, park);
  if (ident) fq12_set_one(&f);
  // the item index again, from a thread index the compiler cannot connect to the one of the kernel's entry (it would keep -- spill -- that one)
  u32 tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const u32 item_out = (blockIdx.x * blockDim.x + tid) >> 1;
  if (item_out >= a.n) return;
  if (a.mode & PAIR_OUT_BYTES) {
    gt_serialize((u32*)a.out + (size_t)96 * item_out, &f, park);
  } else {
    const Fq2d* c = reinterpret_cast<const Fq2d*>(&f);
    Fq* o = (Fq*)a.out + (size_t)12 * item_out;
#pragma unroll
    for (int k = 0; k < 6; k++) park_fq(park, k, c[k].v);
#pragma unroll 1
    for (int k = 0; k < 6; k++) {
      const Fq v = unpark_fq(park, k);
      o[2 * k + lane_odd()] = (a.mode & PAIR_OUT_RAW256) ? to256(v) : v;
    }
  }
}
// the line sequence of a fixed Q (2^256 form in), one workgroup per point: table b of lines_per_table lines for the point q[b]; every lane
// pair of the wave computes the same values, pair 0's layout is the table
static __global__ void __launch_bounds__(64) k_g2_prepare(const G2Aff* __restrict__ q, Line* __restrict__ lines_out, u32 lines_per_table) {
  q += blockIdx.x;
  lines_out += (size_t)blockIdx.x * lines_per_table;
  const Fq* qxw = reinterpret_cast<const Fq*>(&q->x);
  const Fq* qyw = reinterpret_cast<const Fq*>(&q->y);
  Fq2d qx = {to261(qxw[lane_odd()])}, qy = {to261(qyw[lane_odd()])};
  miller_lines(&qx, &qy, lines_out);
}
// test hook: a line table in the 2^256 form (what the oracle tabulates)
static __global__ void __launch_bounds__(64) k_lines_to256(const Fq* __restrict__ in, Fq* __restrict__ out, u32 count) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) out[t] = to256(in[t]);
}

// ---------------------------------------------------------------------------------------------
// Encapsulation at scale: no pairing per item. In the loop of src/vec.rs:63-66 the commitment C is the same for every
// item, so by bilinearity
//     e(r (C - beta g1), g2) = A^r * B^(-beta r),      A = e(C, g2),  B = e(g1, g2)
// with A, B FIXED for the batch: two fixed-base exponentiations in GT with SIGNED window tables T[j][d] = base^(d 2^(wb j)),
// d = 1..2^(wb-1). A and B are outputs of the final exponentiation, i.e. unitary, so base^(-d) is the conjugate of T[j][d]: at most
// ~30 Fq12 products per item instead of a Miller loop + final exponentiation. The value -- hence the serialised bytes and the key --
// is identical. GT elements are stored in the lane-pair order: 12 Fq per element (2^261 form), slot 2k + parity = Fq2 coefficient k,
// component parity (which is also ark-serialize's coefficient order).
// ---------------------------------------------------------------------------------------------
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
// table[j * entries + d] = base^(d 2^(wb j)).  Step 1: the powers of two base^(2^s), s < wb * windows (pows[s]: 12 Fq each, from
// one pairing launch of P against the tabulated multiples 2^s g2: e(P, g2)^(2^s) = e(P, 2^s g2)) go to slot 2^(s mod wb) of window s / wb.
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
// becomes d - 2^wb with a carry (2^wb itself: digit 0, carry 1); a negative digit multiplies by the conjugate (unitary: inverse = conjugate).
// A zero digit multiplies by one: the product is never skipped per lane (the lane pairs of a wave hold different digits).
// The exponent waits in LDS (chunks 12, 13 of the parking area: the Fq12 product uses 0..11) and a window's bits are read from there:
// eight registers fewer across the product than a shift register of the exponent words (the last spills of k_gt_encap_exp).
static KTOWER void gt_table_exp(Fq12* acc, const Fq* __restrict__ tab, GtShape g, const u32 (&k)[8], uint4* park) {
  park[12 * 64 + threadIdx.x] = make_uint4(k[0], k[1], k[2], k[3]);
  park[13 * 64 + threadIdx.x] = make_uint4(k[4], k[5], k[6], k[7]);
  const u32* kw = reinterpret_cast<const u32*>(park);
  auto word = [&](u32 w) -> u32 { return w < 8u ? kw[((12u + (w >> 2)) * 64u + threadIdx.x) * 4u + (w & 3u)] : 0u; };
  u32 carry_d = 0;
  const u32 half = 1u << (g.wb - 1);
#pragma unroll 1
  for (u32 j = 0; j < g.windows; j++) {
    const u32 off = j * g.wb, w = off >> 5, sh = off & 31u;
    u32 d = (__builtin_amdgcn_alignbit(word(w + 1), word(w), sh) & (2u * half - 1u)) + carry_d;
    const bool neg = d > half;
    carry_d = neg ? 1u : 0u;
    if (neg) d = 2u * half - d;
    const Fq* src = tab + ((size_t)j * g.entries + d) * 12;
    const u32 par = lane_odd();
    fq12_mul_ld<true>(acc, acc, [&](int h) {
      Fq6 x;
      Fq2d* c = reinterpret_cast<Fq2d*>(&x);
      if (d) {
#pragma unroll
        for (int k = 0; k < 3; k++) c[k].v = src[6 * h + 2 * k + par];
        if (h == 1 && neg) x = fq6_neg(x);
      } else {
        x = fq6_zero();
        if (h == 0) x.c0 = fq2d_one();
      }
      return x;
    }, park);
  }
}
// gt_out[i] = serialize(A^(r_i) * B^(-(r_i * beta_i)))   (tables of A and B). Two lanes per item.
// The two factors are independent, and only A's table belongs to the commitment: a batch to a NEW commitment runs the B factor first (tab_a null,
// acc_out = 12 Fq per item in the lane-pair order) while A's table is still being built on the side stream, and the A factor behind it (tab_b null,
// acc_in = what the first launch left). Both tables given: one launch, nothing parked.
static __global__ void __launch_bounds__(64, 2) k_gt_encap_exp(const Fq* __restrict__ tab_a, GtShape ga, const Fq* __restrict__ tab_b, GtShape gb,
                                                              const Fr* __restrict__ betas, const Fr* __restrict__ rs, u32 n, u32* __restrict__ gt_out,
                                                              const Fq* __restrict__ acc_in, Fq* __restrict__ acc_out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 item = t >> 1;
  const bool live = item < n;
  const u32 i = live ? item : (n - 1);
  __shared__ uint4 park[PARK_CHUNKS * 64];
  Fq12 acc;
  if (acc_in) gt_load(&acc, acc_in + (size_t)12 * i);
  else fq12_set_one(&acc);
  if (tab_a) {
    u32 u[8];
    fp_from_mont<FrParams>(u, rs[i]);
    gt_table_exp(&acc, tab_a, ga, u, park);
  }
  asm volatile("" ::: "memory");          // the second exponent is formed here, not carried across the first exponentiation
  if (tab_b) {
    u32 v[8];
    fp_from_mont<FrParams>(v, fp_neg<FrParams>(fp_mul<FrParams>(rs[i], betas[i])));
    gt_table_exp(&acc, tab_b, gb, v, park);
  }
  if (!live) return;
  if (acc_out) gt_store(acc_out + (size_t)12 * i, &acc);
  else gt_serialize(gt_out + (size_t)96 * i, &acc, park);
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
// xor_into != 0: the DEM of src/enc.rs:32-36 / :48-52 behind the KDF -- key_out holds the messages (ciphertext bodies) on entry and the
// ciphertext bodies (messages) on exit; the key never leaves the device.
static __global__ void __launch_bounds__(256) k_blake3_gt_xof(const u32* __restrict__ gt, u32 n, unsigned char* __restrict__ key_out, u32 msg_len, u32 xor_into) {
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
    if (xor_into) { for (u32 k = 0; k < nb; k++) dst[t * 64 + k] ^= (unsigned char)(o[k >> 2] >> (8 * (k & 3))); }
    else { for (u32 k = 0; k < nb; k++) dst[t * 64 + k] = (unsigned char)(o[k >> 2] >> (8 * (k & 3))); }
  }
}

}  // namespace bn254
