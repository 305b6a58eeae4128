// Batched BN254 optimal-ate pairing on gfx950: one lane per pairing.
//
// Replaces `E::pairing(p, q)` (reference src/kem.rs:30,58; src/kzg.rs:148; ark-ec 0.4.2 models/bn:
// G2Prepared line coefficients + multi_miller_loop + final_exponentiation) and
// `serialize_uncompressed` of the GT element (src/kem.rs:32,61).
//
// Tower (ark-bn254): Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - (9+u)), Fq12 = Fq6[w]/(w^2 - v),
// D-type twist. The reduced pairing value is independent of the Miller-loop addition chain and of
// subfield scalings of the line functions, so this kernel is free to (a) compute the lines on the
// fly instead of materialising G2Prepared, (b) use the proper NAF of 6z+2 (22 additions instead of
// arkworks' 26). The final exponent is arkworks' exactly: (p^12-1)/r * 2z(6z^2+3z+1).
//
// Register pressure: an Fq12 is 96 dwords; the Fq2 product and the Fq12-level routines are real
// (non-inlined) functions so code size stays in the instruction cache.
#pragma once
#include "bn254_curve.cuh"

namespace bn254 {

#define KNOINLINE __device__ __noinline__
#ifdef KEAKI_PAIRING_INLINE_TOWER
#define KTOWER __device__ __forceinline__
#else
#define KTOWER __device__ __noinline__
#endif

struct Fq6 { Fq2 c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };

// Fq2 product / square are out-of-line functions in this translation unit (KEAKI_FQ2_OUTLINE): the
// unit of code reuse for the tower
#ifndef KEAKI_FQ2_OUTLINE
#error "pairing.cuh expects KEAKI_FQ2_OUTLINE (see pairing.hip)"
#endif
#define M2(a, b) ((a) * (b))
#define S2(a) fq2_sqr((a))

KDEV Fq6 operator+(const Fq6& a, const Fq6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
KDEV Fq6 operator-(const Fq6& a, const Fq6& b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
KDEV Fq6 operator-(const Fq6& a) { return {-a.c0, -a.c1, -a.c2}; }
KDEV Fq6 fq6_mul_v(const Fq6& a) { return {fq2_mul_xi(a.c2), a.c0, a.c1}; }
KDEV Fq6 fq6_zero() { return {fq2_zero(), fq2_zero(), fq2_zero()}; }

static KTOWER void fq6_mul(Fq6* r, const Fq6* a, const Fq6* b) {
  Fq2 v0 = M2(a->c0, b->c0), v1 = M2(a->c1, b->c1), v2 = M2(a->c2, b->c2);
  Fq2 t0 = fq2_mul_xi(M2(a->c1 + a->c2, b->c1 + b->c2) - v1 - v2) + v0;
  Fq2 t1 = M2(a->c0 + a->c1, b->c0 + b->c1) - v0 - v1 + fq2_mul_xi(v2);
  Fq2 t2 = M2(a->c0 + a->c2, b->c0 + b->c2) - v0 - v2 + v1;
  r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
// a * (c0 + c1 v)
static KTOWER void fq6_mul_by_01(Fq6* r, const Fq6* a, const Fq2* c0, const Fq2* c1) {
  Fq2 aa = M2(a->c0, *c0), bb = M2(a->c1, *c1);
  Fq2 t1 = fq2_mul_xi(M2(*c1, a->c1 + a->c2) - bb) + aa;
  Fq2 t3 = M2(*c0, a->c0 + a->c2) - aa + bb;
  Fq2 t2 = M2(*c0 + *c1, a->c0 + a->c1) - aa - bb;
  r->c0 = t1; r->c1 = t2; r->c2 = t3;
}
static KTOWER void fq6_inv(Fq6* r, const Fq6* a) {
  Fq2 t0 = S2(a->c0) - fq2_mul_xi(M2(a->c1, a->c2));
  Fq2 t1 = fq2_mul_xi(S2(a->c2)) - M2(a->c0, a->c1);
  Fq2 t2 = S2(a->c1) - M2(a->c0, a->c2);
  Fq2 n = M2(a->c0, t0) + fq2_mul_xi(M2(a->c2, t1) + M2(a->c1, t2));
  Fq2 ni = fq2_inv(n);
  r->c0 = M2(t0, ni); r->c1 = M2(t1, ni); r->c2 = M2(t2, ni);
}

KDEV void fq12_set_one(Fq12* f) {
  f->c0 = fq6_zero(); f->c1 = fq6_zero();
  f->c0.c0.c0 = fq_one();
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
static KTOWER void fq12_mul_by_034(Fq12* f, const Fq2* c0, const Fq2* d0, const Fq2* d1) {
  Fq6 a = {M2(f->c0.c0, *c0), M2(f->c0.c1, *c0), M2(f->c0.c2, *c0)};
  Fq6 b, e, s = f->c0 + f->c1;
  fq6_mul_by_01(&b, &f->c1, d0, d1);
  Fq2 cs = *c0 + *d0;
  fq6_mul_by_01(&e, &s, &cs, d1);
  f->c1 = e - a - b;
  f->c0 = fq6_mul_v(b) + a;
}
// x -> x^(p^k), k = 1, 2, 3
static KTOWER void fq12_frob(Fq12* r, const Fq12* a, int k) {
  Fq2 c[6] = {a->c0.c0, a->c1.c0, a->c0.c1, a->c1.c1, a->c0.c2, a->c1.c2};
  Fq2 o[6];
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    Fq2 t = (k & 1) ? fq2_conj(c[i]) : c[i];
    o[i] = M2(t, FROB_W[k][i]);
  }
  r->c0.c0 = o[0]; r->c1.c0 = o[1]; r->c0.c1 = o[2]; r->c1.c1 = o[3]; r->c0.c2 = o[4]; r->c1.c2 = o[5];
}
// Granger-Scott squaring, valid on the cyclotomic subgroup (after the easy part): 9 Fq2 products... as
// 3 x (1 product + 1 product) pairs
static KTOWER void fq12_cyc_sqr(Fq12* r, const Fq12* a) {
  const Fq2 r0 = a->c0.c0, r4 = a->c0.c1, r3 = a->c0.c2, r2 = a->c1.c0, r1 = a->c1.c1, r5 = a->c1.c2;
  Fq2 tmp, t0, t1, t2, t3, t4, t5;
  tmp = M2(r0, r1); t0 = M2(r0 + r1, fq2_mul_xi(r1) + r0) - tmp - fq2_mul_xi(tmp); t1 = fq2_dbl(tmp);
  tmp = M2(r2, r3); t2 = M2(r2 + r3, fq2_mul_xi(r3) + r2) - tmp - fq2_mul_xi(tmp); t3 = fq2_dbl(tmp);
  tmp = M2(r4, r5); t4 = M2(r4 + r5, fq2_mul_xi(r5) + r4) - tmp - fq2_mul_xi(tmp); t5 = fq2_dbl(tmp);
  Fq2 x5 = fq2_mul_xi(t5);
  r->c0.c0 = fq2_dbl(t0 - r0) + t0;
  r->c1.c1 = fq2_dbl(t1 + r1) + t1;
  r->c1.c0 = fq2_dbl(x5 + r2) + x5;
  r->c0.c2 = fq2_dbl(t4 - r3) + t4;
  r->c0.c1 = fq2_dbl(t2 - r4) + t2;
  r->c1.c2 = fq2_dbl(t3 + r5) + t3;
}
// f^(-z): square-and-multiply over the bits of z with cyclotomic squarings, then conjugate
static KTOWER void fq12_exp_by_neg_z(Fq12* r, const Fq12* f) {
  Fq12 acc = *f;
#pragma unroll 1
  for (int i = 61; i >= 0; i--) {  // z has 63 bits, top bit handled by acc = f
    fq12_cyc_sqr(&acc, &acc);
    if ((BN_Z >> i) & 1) fq12_mul(&acc, &acc, f);
  }
  fq12_conj(r, &acc);
}
static_assert((BN_Z >> 62) == 1, "z must be a 63-bit value");

// ---- line functions on the twist, homogeneous projective (same formulas as ark-ec bn/g2.rs) ----
struct G2Hom { Fq2 x, y, z; };
struct Line { Fq2 c0, c1, c2; };  // evaluated as c0 * P.y + (c1 * P.x + c2 v) w

static KTOWER void line_double(G2Hom* r, Line* l) {
  Fq2 a = fq2_mul_fq(M2(r->x, r->y), FQ_TWO_INV);
  Fq2 b = S2(r->y), c = S2(r->z);
  Fq2 e = M2(G2_B, fq2_dbl(c) + c);
  Fq2 f = fq2_dbl(e) + e;
  Fq2 g = fq2_mul_fq(b + f, FQ_TWO_INV);
  Fq2 h = S2(r->y + r->z) - (b + c);
  Fq2 i = e - b;
  Fq2 j = S2(r->x);
  Fq2 e2 = S2(e);
  r->x = M2(a, b - f);
  r->y = S2(g) - (fq2_dbl(e2) + e2);
  r->z = M2(b, h);
  l->c0 = -h; l->c1 = fq2_dbl(j) + j; l->c2 = i;
}
static KTOWER void line_add(G2Hom* r, const Fq2* qx, const Fq2* qy, Line* l) {
  Fq2 theta = r->y - M2(*qy, r->z);
  Fq2 lam = r->x - M2(*qx, r->z);
  Fq2 c = S2(theta), d = S2(lam);
  Fq2 e = M2(lam, d), f = M2(r->z, c), g = M2(r->x, d);
  Fq2 h = e + f - fq2_dbl(g);
  Fq2 ny = M2(theta, g - h) - M2(e, r->y);
  r->x = M2(lam, h);
  r->y = ny;
  r->z = M2(r->z, e);
  l->c0 = lam; l->c1 = -theta; l->c2 = M2(theta, *qx) - M2(lam, *qy);
}
KDEV void ell(Fq12* f, const Line* l, const G1Aff* p) {
  Fq2 c0 = fq2_mul_fq(l->c0, p->y), c1 = fq2_mul_fq(l->c1, p->x);
  fq12_mul_by_034(f, &c0, &c1, &l->c2);
}

static __device__ void miller_loop(Fq12* f, const G1Aff* p, const G2Aff* q) {
  fq12_set_one(f);
  G2Hom r = {q->x, q->y, fq2_one()};
  Fq2 nqy = -q->y;
  Line l;
#pragma unroll 1
  for (int i = ATE_LEN - 2; i >= 0; i--) {
    if (i != ATE_LEN - 2) fq12_sqr(f, f);
    line_double(&r, &l);
    ell(f, &l, p);
    int d = ATE_NAF[i];
    if (d != 0) {
      line_add(&r, &q->x, d > 0 ? &q->y : &nqy, &l);
      ell(f, &l, p);
    }
  }
  // Q1 = pi(Q), Q2 = -pi^2(Q)
  Fq2 q1x = M2(fq2_conj(q->x), TWIST_MUL_BY_Q_X), q1y = M2(fq2_conj(q->y), TWIST_MUL_BY_Q_Y);
  Fq2 q2x = M2(fq2_conj(q1x), TWIST_MUL_BY_Q_X), q2y = -M2(fq2_conj(q1y), TWIST_MUL_BY_Q_Y);
  line_add(&r, &q1x, &q1y, &l);
  ell(f, &l, p);
  line_add(&r, &q2x, &q2y, &l);
  ell(f, &l, p);
}

// easy part (p^6-1)(p^2+1), hard part = arkworks' Fuentes-Castaneda chain (exponent 2z(6z^2+3z+1)(p^4-p^2+1)/r)
static __device__ void final_exponentiation(Fq12* out, const Fq12* fin) {
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

// GT -> 384 canonical little-endian bytes in ark-serialize order (c0.c0.c0 ... c1.c2.c1)
KDEV void gt_serialize(u32* out96, const Fq12* f) {
  const Fq* c = reinterpret_cast<const Fq*>(f);
#pragma unroll 1
  for (int i = 0; i < 12; i++) {
    u32 w[8];
    fp_from_mont<FqParams>(w, c[i]);
#pragma unroll
    for (int j = 0; j < 8; j++) out96[8 * i + j] = w[j];
  }
}

// gt_out[i] = serialize(e(P_i, Q_{i*stride})); identity in either slot -> one
__global__ void __launch_bounds__(64) k_pairing_batch(const G1Aff* __restrict__ ps, const G2Aff* __restrict__ qs, int q_stride, u32 n,
                                                      u32* __restrict__ gt_out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Aff p = ps[i];
  G2Aff q = qs[(size_t)i * q_stride];
  Fq12 f, e;
  if (aff_is_inf(p) || aff_is_inf(q)) {
    fq12_set_one(&e);
  } else {
    miller_loop(&f, &p, &q);
    final_exponentiation(&e, &f);
  }
  gt_serialize(gt_out + (size_t)96 * i, &e);
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
__global__ void __launch_bounds__(256) k_blake3_gt_xof(const u32* __restrict__ gt, u32 n, unsigned char* __restrict__ key_out, u32 msg_len) {
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
