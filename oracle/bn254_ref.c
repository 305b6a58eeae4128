/* TEST INFRASTRUCTURE ONLY -- fast CPU restatement (plain C, 4x u64 Montgomery limbs, unsigned
 * __int128) of the arkworks 0.4 algorithms keaki's hot path executes on BN254. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (keaki_amd/) never links or calls it.
 *
 * Restates (reference call site -> function here); the arithmetic lives in un-vendored crates
 * (ark-ec 0.4.2, ark-ff 0.4.2, ark-bn254 0.4.0, ark-serialize 0.4.2, blake3 1.5.4 -- absent from
 * /root/reference, see oracle/bn254_py.py header), so this follows their published algorithms:
 *   src/kzg.rs:98        msm_unchecked (msm_bigint_wnaf + make_digits)   -> ref_msm_g1 / ref_msm_g2
 *   src/kem.rs:22,30,36  .mul(scalar) = MSB-first double-and-add         -> ref_g1_mul_batch / ref_g2_mul_batch
 *   src/kem.rs:30,58     E::pairing: G2Prepared + Miller loop + final exp -> ref_pairing_batch
 *   src/kem.rs:32,61     serialize_uncompressed(GT), 384 B                -> inside ref_pairing_batch
 *   src/kem.rs:42-46     BLAKE3 XOF                                       -> ref_blake3_xof
 *   src/kem.rs:13-50     encapsulate, looped as src/vec.rs:63-66          -> ref_encap_batch
 *   src/kem.rs:55-72     decapsulate, looped as src/vec.rs:75-78          -> ref_decap_batch
 *
 * PARITY: pinned against oracle/bn254_py.py (big-int restatement) by tests/test_oracle.py, which in
 * turn is pinned by public KATs + the reference's ptau fixture. Versus arkworks itself: "parity
 * unpinned" (no arkworks build possible here; the reference holds no absolute vectors).
 *
 * Layouts at this API = the C-ABI layouts of include/keaki_hip.h: Fr/Fq = u64[4] Montgomery limbs
 * exactly as ark-ff holds them; G1 affine u64[8] (x,y), identity = all-zero; G2 affine u64[16]
 * (x.c0,x.c1,y.c0,y.c1), identity = all-zero; GT = 384 canonical LE bytes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef struct { u64 l[4]; } fq;
typedef struct { fq c0, c1; } fq2;
typedef struct { fq2 c0, c1, c2; } fq6;
typedef struct { fq6 c0, c1; } fq12;

#include "bn254_constants.h"

/* ------------------------------------------------------------------ base field (generic modulus) */
static inline int geq(const u64 a[4], const u64 m[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > m[i]) return 1; if (a[i] < m[i]) return 0; }
    return 1;
}
static inline void sub_nb(u64 r[4], const u64 a[4], const u64 b[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - (u64)br; r[i] = (u64)t; br = (t >> 64) & 1; }
}
static inline void mod_add(u64 r[4], const u64 a[4], const u64 b[4], const u64 m[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; }
    if (c || geq(r, m)) sub_nb(r, r, m);
}
static inline void mod_sub(u64 r[4], const u64 a[4], const u64 b[4], const u64 m[4]) {
    u128 br = 0; u64 t[4];
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - (u64)br; t[i] = (u64)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + m[i]; t[i] = (u64)c; c >>= 64; } }
    memcpy(r, t, 32);
}
static inline void mont_mul(u64 r[4], const u64 a[4], const u64 b[4], const u64 m[4], u64 inv) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        u64 mm = t[0] * inv;
        c = (u128)mm * m[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)mm * m[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq(t, m)) sub_nb(t, t, m);
    memcpy(r, t, 32);
}

#define FQ_(n) fq_##n
static inline void fq_add(fq *r, const fq *a, const fq *b) { mod_add(r->l, a->l, b->l, FQ_MOD); }
static inline void fq_sub(fq *r, const fq *a, const fq *b) { mod_sub(r->l, a->l, b->l, FQ_MOD); }
static inline void fq_dbl(fq *r, const fq *a) { mod_add(r->l, a->l, a->l, FQ_MOD); }
static inline void fq_mul(fq *r, const fq *a, const fq *b) { mont_mul(r->l, a->l, b->l, FQ_MOD, FQ_INV); }
static inline void fq_sqr(fq *r, const fq *a) { mont_mul(r->l, a->l, a->l, FQ_MOD, FQ_INV); }
static inline void fq_zero(fq *r) { memset(r, 0, sizeof *r); }
static inline void fq_one(fq *r) { *r = FQ_ONE; }
static inline int fq_is_zero(const fq *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fq_eq(const fq *a, const fq *b) { return memcmp(a, b, 32) == 0; }
static inline void fq_neg(fq *r, const fq *a) { if (fq_is_zero(a)) *r = *a; else sub_nb(r->l, FQ_MOD, a->l); }
static void fq_pow(fq *r, const fq *a, const u64 e[4]) {
    fq acc = FQ_ONE;
    for (int i = 255; i >= 0; i--) { fq_sqr(&acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) fq_mul(&acc, &acc, a); }
    *r = acc;
}
static void fq_inv(fq *r, const fq *a) {  /* Fermat: a^(p-2) */
    u64 e[4]; u64 two[4] = {2, 0, 0, 0}; sub_nb(e, FQ_MOD, two); fq_pow(r, a, e);
}
static inline void fq_from_mont(u64 out[4], const fq *a) { u64 one[4] = {1, 0, 0, 0}; mont_mul(out, a->l, one, FQ_MOD, FQ_INV); }
static inline void fr_from_mont(u64 out[4], const u64 a[4]) { u64 one[4] = {1, 0, 0, 0}; mont_mul(out, a, one, FR_MOD, FR_INV); }
static inline void fr_to_mont(u64 out[4], const u64 a[4]) { mont_mul(out, a, FR_R2, FR_MOD, FR_INV); }
static inline void fr_mul(u64 r[4], const u64 a[4], const u64 b[4]) { mont_mul(r, a, b, FR_MOD, FR_INV); }

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1) */
static inline void fq2_add(fq2 *r, const fq2 *a, const fq2 *b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static inline void fq2_sub(fq2 *r, const fq2 *a, const fq2 *b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static inline void fq2_dbl(fq2 *r, const fq2 *a) { fq_dbl(&r->c0, &a->c0); fq_dbl(&r->c1, &a->c1); }
static inline void fq2_neg(fq2 *r, const fq2 *a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static inline void fq2_conj(fq2 *r, const fq2 *a) { r->c0 = a->c0; fq_neg(&r->c1, &a->c1); }
static inline void fq2_zero(fq2 *r) { memset(r, 0, sizeof *r); }
static inline void fq2_one(fq2 *r) { r->c0 = FQ_ONE; fq_zero(&r->c1); }
static inline int fq2_is_zero(const fq2 *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static inline int fq2_eq(const fq2 *a, const fq2 *b) { return memcmp(a, b, 64) == 0; }
static inline void fq2_mul(fq2 *r, const fq2 *a, const fq2 *b) {  /* Karatsuba, 3 mults */
    fq t0, t1, s0, s1, t2;
    fq_mul(&t0, &a->c0, &b->c0); fq_mul(&t1, &a->c1, &b->c1);
    fq_add(&s0, &a->c0, &a->c1); fq_add(&s1, &b->c0, &b->c1); fq_mul(&t2, &s0, &s1);
    fq_sub(&r->c0, &t0, &t1);
    fq_sub(&t2, &t2, &t0); fq_sub(&r->c1, &t2, &t1);
}
static inline void fq2_sqr(fq2 *r, const fq2 *a) {  /* (a0+a1)(a0-a1), 2 a0 a1 */
    fq s, d, m;
    fq_add(&s, &a->c0, &a->c1); fq_sub(&d, &a->c0, &a->c1); fq_mul(&m, &a->c0, &a->c1);
    fq_mul(&r->c0, &s, &d); fq_dbl(&r->c1, &m);
}
static inline void fq2_mul_fp(fq2 *r, const fq2 *a, const fq *k) { fq_mul(&r->c0, &a->c0, k); fq_mul(&r->c1, &a->c1, k); }
static inline void fq2_mul_xi(fq2 *r, const fq2 *a) {  /* (9+u) a */
    fq t0, t1, n0, n1;
    fq_dbl(&t0, &a->c0); fq_dbl(&t0, &t0); fq_dbl(&t0, &t0); fq_add(&t0, &t0, &a->c0);  /* 9 a0 */
    fq_dbl(&t1, &a->c1); fq_dbl(&t1, &t1); fq_dbl(&t1, &t1); fq_add(&t1, &t1, &a->c1);  /* 9 a1 */
    fq_sub(&n0, &t0, &a->c1); fq_add(&n1, &t1, &a->c0);
    r->c0 = n0; r->c1 = n1;
}
static void fq2_inv(fq2 *r, const fq2 *a) {
    fq n, t, ni;
    fq_sqr(&n, &a->c0); fq_sqr(&t, &a->c1); fq_add(&n, &n, &t); fq_inv(&ni, &n);
    fq_mul(&r->c0, &a->c0, &ni); fq_mul(&t, &a->c1, &ni); fq_neg(&r->c1, &t);
}
#define FQ2_(n) fq2_##n

/* ------------------------------------------------------------------ G1 / G2 (template stamped twice) */
#define FE fq
#define FE_(n) fq_##n
#define PT(n) g1_##n
#define AFF g1_aff
#define JAC g1_jac
#include "ec_tmpl.inc"
#undef FE
#undef FE_
#undef PT
#undef AFF
#undef JAC
#define FE fq2
#define FE_(n) fq2_##n
#define PT(n) g2_##n
#define AFF g2_aff
#define JAC g2_jac
#include "ec_tmpl.inc"
#undef FE
#undef FE_
#undef PT
#undef AFF
#undef JAC

/* ------------------------------------------------------------------ Fq6, Fq12 */
static inline void fq6_add(fq6 *r, const fq6 *a, const fq6 *b) { fq2_add(&r->c0, &a->c0, &b->c0); fq2_add(&r->c1, &a->c1, &b->c1); fq2_add(&r->c2, &a->c2, &b->c2); }
static inline void fq6_sub(fq6 *r, const fq6 *a, const fq6 *b) { fq2_sub(&r->c0, &a->c0, &b->c0); fq2_sub(&r->c1, &a->c1, &b->c1); fq2_sub(&r->c2, &a->c2, &b->c2); }
static inline void fq6_neg(fq6 *r, const fq6 *a) { fq2_neg(&r->c0, &a->c0); fq2_neg(&r->c1, &a->c1); fq2_neg(&r->c2, &a->c2); }
static inline void fq6_mul_v(fq6 *r, const fq6 *a) { fq2 t; fq2_mul_xi(&t, &a->c2); r->c2 = a->c1; r->c1 = a->c0; r->c0 = t; }
static void fq6_mul(fq6 *r, const fq6 *a, const fq6 *b) {  /* Karatsuba/Toom (6 Fq2 mults) */
    fq2 v0, v1, v2, t0, t1, t2, s0, s1;
    fq2_mul(&v0, &a->c0, &b->c0); fq2_mul(&v1, &a->c1, &b->c1); fq2_mul(&v2, &a->c2, &b->c2);
    fq2_add(&s0, &a->c1, &a->c2); fq2_add(&s1, &b->c1, &b->c2); fq2_mul(&t0, &s0, &s1);
    fq2_sub(&t0, &t0, &v1); fq2_sub(&t0, &t0, &v2); fq2_mul_xi(&t0, &t0); fq2_add(&t0, &t0, &v0);
    fq2_add(&s0, &a->c0, &a->c1); fq2_add(&s1, &b->c0, &b->c1); fq2_mul(&t1, &s0, &s1);
    fq2_sub(&t1, &t1, &v0); fq2_sub(&t1, &t1, &v1); fq2_mul_xi(&s0, &v2); fq2_add(&t1, &t1, &s0);
    fq2_add(&s0, &a->c0, &a->c2); fq2_add(&s1, &b->c0, &b->c2); fq2_mul(&t2, &s0, &s1);
    fq2_sub(&t2, &t2, &v0); fq2_sub(&t2, &t2, &v2); fq2_add(&t2, &t2, &v1);
    r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
/* ark-ff Fp6::mul_by_01: multiply by c0 + c1 v */
static void fq6_mul_by_01(fq6 *r, const fq6 *a, const fq2 *c0, const fq2 *c1) {
    fq2 aa, bb, t1, t2, t3, s;
    fq2_mul(&aa, &a->c0, c0); fq2_mul(&bb, &a->c1, c1);
    fq2_add(&s, &a->c1, &a->c2); fq2_mul(&t1, c1, &s); fq2_sub(&t1, &t1, &bb); fq2_mul_xi(&t1, &t1); fq2_add(&t1, &t1, &aa);
    fq2_add(&s, &a->c0, &a->c2); fq2_mul(&t3, c0, &s); fq2_sub(&t3, &t3, &aa); fq2_add(&t3, &t3, &bb);
    fq2 cs; fq2_add(&cs, c0, c1); fq2_add(&s, &a->c0, &a->c1); fq2_mul(&t2, &cs, &s); fq2_sub(&t2, &t2, &aa); fq2_sub(&t2, &t2, &bb);
    r->c0 = t1; r->c1 = t2; r->c2 = t3;
}
static void fq6_inv(fq6 *r, const fq6 *a) {
    fq2 t0, t1, t2, s, n, ni;
    fq2_sqr(&t0, &a->c0); fq2_mul(&s, &a->c1, &a->c2); fq2_mul_xi(&s, &s); fq2_sub(&t0, &t0, &s);
    fq2_sqr(&t1, &a->c2); fq2_mul_xi(&t1, &t1); fq2_mul(&s, &a->c0, &a->c1); fq2_sub(&t1, &t1, &s);
    fq2_sqr(&t2, &a->c1); fq2_mul(&s, &a->c0, &a->c2); fq2_sub(&t2, &t2, &s);
    fq2 u, v; fq2_mul(&u, &a->c2, &t1); fq2_mul(&v, &a->c1, &t2); fq2_add(&u, &u, &v); fq2_mul_xi(&u, &u);
    fq2_mul(&n, &a->c0, &t0); fq2_add(&n, &n, &u); fq2_inv(&ni, &n);
    fq2_mul(&r->c0, &t0, &ni); fq2_mul(&r->c1, &t1, &ni); fq2_mul(&r->c2, &t2, &ni);
}
static inline void fq12_one(fq12 *r) { memset(r, 0, sizeof *r); r->c0.c0.c0 = FQ_ONE; }
static void fq12_mul(fq12 *r, const fq12 *a, const fq12 *b) {
    fq6 t0, t1, s0, s1, m;
    fq6_mul(&t0, &a->c0, &b->c0); fq6_mul(&t1, &a->c1, &b->c1);
    fq6_add(&s0, &a->c0, &a->c1); fq6_add(&s1, &b->c0, &b->c1); fq6_mul(&m, &s0, &s1);
    fq6_sub(&m, &m, &t0); fq6_sub(&m, &m, &t1);
    fq6_mul_v(&s0, &t1); fq6_add(&r->c0, &t0, &s0);
    r->c1 = m;
}
static void fq12_sqr(fq12 *r, const fq12 *a) {  /* complex squaring */
    fq6 ab, s0, s1, t;
    fq6_mul(&ab, &a->c0, &a->c1);
    fq6_add(&s0, &a->c0, &a->c1);
    fq6_mul_v(&t, &a->c1); fq6_add(&s1, &a->c0, &t);
    fq6_mul(&s0, &s0, &s1);
    fq6_sub(&s0, &s0, &ab); fq6_mul_v(&t, &ab); fq6_sub(&r->c0, &s0, &t);
    fq6_add(&r->c1, &ab, &ab);
}
static inline void fq12_conj(fq12 *r, const fq12 *a) { r->c0 = a->c0; fq6_neg(&r->c1, &a->c1); }
static void fq12_inv(fq12 *r, const fq12 *a) {
    fq6 n, t, ni;
    fq6_mul(&n, &a->c0, &a->c0); fq6_mul(&t, &a->c1, &a->c1); fq6_mul_v(&t, &t); fq6_sub(&n, &n, &t);
    fq6_inv(&ni, &n);
    fq6_mul(&r->c0, &a->c0, &ni); fq6_mul(&t, &a->c1, &ni); fq6_neg(&r->c1, &t);
}
/* ark-ff Fp12::mul_by_034: f *= c0 + (d0 + d1 v) w */
static void fq12_mul_by_034(fq12 *f, const fq2 *c0, const fq2 *d0, const fq2 *d1) {
    fq6 a, b, e, s; fq2 cs;
    fq2_mul(&a.c0, &f->c0.c0, c0); fq2_mul(&a.c1, &f->c0.c1, c0); fq2_mul(&a.c2, &f->c0.c2, c0);
    fq6_mul_by_01(&b, &f->c1, d0, d1);
    fq2_add(&cs, c0, d0);
    fq6_add(&s, &f->c0, &f->c1); fq6_mul_by_01(&e, &s, &cs, d1);
    fq6_sub(&e, &e, &a); fq6_sub(&f->c1, &e, &b);
    fq6_mul_v(&s, &b); fq6_add(&f->c0, &s, &a);
}
/* Frobenius x -> x^(p^k), k = 1..3, in the w-basis (see bn254_py.f12_frob) */
static void fq12_frob(fq12 *r, const fq12 *a, int k) {
    const fq2 *c[6] = {&a->c0.c0, &a->c1.c0, &a->c0.c1, &a->c1.c1, &a->c0.c2, &a->c1.c2};
    fq2 o[6];
    for (int i = 0; i < 6; i++) {
        fq2 t = *c[i];
        if (k & 1) fq2_conj(&t, &t);
        fq2_mul(&o[i], &t, &FROB_W[k][i]);
    }
    r->c0.c0 = o[0]; r->c1.c0 = o[1]; r->c0.c1 = o[2]; r->c1.c1 = o[3]; r->c0.c2 = o[4]; r->c1.c2 = o[5];
}
/* Granger-Scott squaring in the cyclotomic subgroup (ark-ff Fp12::cyclotomic_square) */
static void fq12_cyc_sqr(fq12 *r, const fq12 *a) {
    const fq2 *r0 = &a->c0.c0, *r4 = &a->c0.c1, *r3 = &a->c0.c2, *r2 = &a->c1.c0, *r1 = &a->c1.c1, *r5 = &a->c1.c2;
    fq2 tmp, t0, t1, t2, t3, t4, t5, s0, s1, x;
#define GS_PAIR(ta, tb, ra, rb) \
    fq2_mul(&tmp, ra, rb); fq2_add(&s0, ra, rb); fq2_mul_xi(&s1, rb); fq2_add(&s1, &s1, ra); \
    fq2_mul(&ta, &s0, &s1); fq2_sub(&ta, &ta, &tmp); fq2_mul_xi(&x, &tmp); fq2_sub(&ta, &ta, &x); fq2_dbl(&tb, &tmp);
    GS_PAIR(t0, t1, r0, r1)
    GS_PAIR(t2, t3, r2, r3)
    GS_PAIR(t4, t5, r4, r5)
#undef GS_PAIR
    fq12 z;
    fq2_sub(&z.c0.c0, &t0, r0); fq2_dbl(&z.c0.c0, &z.c0.c0); fq2_add(&z.c0.c0, &z.c0.c0, &t0);
    fq2_add(&z.c1.c1, &t1, r1); fq2_dbl(&z.c1.c1, &z.c1.c1); fq2_add(&z.c1.c1, &z.c1.c1, &t1);
    fq2_mul_xi(&tmp, &t5);
    fq2_add(&z.c1.c0, &tmp, r2); fq2_dbl(&z.c1.c0, &z.c1.c0); fq2_add(&z.c1.c0, &z.c1.c0, &tmp);
    fq2_sub(&z.c0.c2, &t4, r3); fq2_dbl(&z.c0.c2, &z.c0.c2); fq2_add(&z.c0.c2, &z.c0.c2, &t4);
    fq2_sub(&z.c0.c1, &t2, r4); fq2_dbl(&z.c0.c1, &z.c0.c1); fq2_add(&z.c0.c1, &z.c0.c1, &t2);
    fq2_add(&z.c1.c2, &t3, r5); fq2_dbl(&z.c1.c2, &z.c1.c2); fq2_add(&z.c1.c2, &z.c1.c2, &t3);
    *r = z;
}
/* f^z by square-and-multiply with cyclotomic squarings, then conjugate = exp_by_neg_x (X positive) */
static void fq12_exp_by_neg_x(fq12 *r, const fq12 *f) {
    fq12 acc = *f;
    int top = 63; while (!((BN_Z >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) { fq12_cyc_sqr(&acc, &acc); if ((BN_Z >> i) & 1) fq12_mul(&acc, &acc, f); }
    fq12_conj(r, &acc);
}

/* ------------------------------------------------------------------ pairing (ark-ec models/bn) */
typedef struct { fq2 c0, c1, c2; } ell_coeff;
typedef struct { fq2 x, y, z; } g2_hom;

static void line_double(g2_hom *r, ell_coeff *l) {
    fq2 a, b, c, e, f, g, h, i, j, e2, t;
    fq2_mul(&a, &r->x, &r->y); fq2_mul_fp(&a, &a, &FQ_TWO_INV);
    fq2_sqr(&b, &r->y); fq2_sqr(&c, &r->z);
    fq2_dbl(&t, &c); fq2_add(&t, &t, &c); fq2_mul(&e, &FQ2_B2, &t);
    fq2_dbl(&f, &e); fq2_add(&f, &f, &e);
    fq2_add(&g, &b, &f); fq2_mul_fp(&g, &g, &FQ_TWO_INV);
    fq2_add(&h, &r->y, &r->z); fq2_sqr(&h, &h); fq2_add(&t, &b, &c); fq2_sub(&h, &h, &t);
    fq2_sub(&i, &e, &b);
    fq2_sqr(&j, &r->x);
    fq2_sqr(&e2, &e);
    fq2_sub(&t, &b, &f); fq2_mul(&r->x, &a, &t);
    fq2_sqr(&g, &g); fq2_dbl(&t, &e2); fq2_add(&t, &t, &e2); fq2_sub(&r->y, &g, &t);
    fq2_mul(&r->z, &b, &h);
    fq2_neg(&l->c0, &h); fq2_dbl(&t, &j); fq2_add(&l->c1, &t, &j); l->c2 = i;
}
static void line_add(g2_hom *r, const fq2 *qx, const fq2 *qy, ell_coeff *l) {
    fq2 theta, lam, c, d, e, f, g, h, t, j;
    fq2_mul(&t, qy, &r->z); fq2_sub(&theta, &r->y, &t);
    fq2_mul(&t, qx, &r->z); fq2_sub(&lam, &r->x, &t);
    fq2_sqr(&c, &theta); fq2_sqr(&d, &lam);
    fq2_mul(&e, &lam, &d); fq2_mul(&f, &r->z, &c); fq2_mul(&g, &r->x, &d);
    fq2_add(&h, &e, &f); fq2_dbl(&t, &g); fq2_sub(&h, &h, &t);
    fq2 ny; fq2_sub(&t, &g, &h); fq2_mul(&ny, &theta, &t); fq2_mul(&t, &e, &r->y); fq2_sub(&ny, &ny, &t);
    fq2_mul(&r->x, &lam, &h); r->y = ny; fq2_mul(&r->z, &r->z, &e);
    fq2_mul(&j, &theta, qx); fq2_mul(&t, &lam, qy); fq2_sub(&j, &j, &t);
    l->c0 = lam; fq2_neg(&l->c1, &theta); l->c2 = j;
}
static inline void ell(fq12 *f, const ell_coeff *l, const g1_aff *p) {
    fq2 c0, c1;
    fq2_mul_fp(&c0, &l->c0, &p->y); fq2_mul_fp(&c1, &l->c1, &p->x);
    fq12_mul_by_034(f, &c0, &c1, &l->c2);
}
static int g2_prepare(const g2_aff *q, ell_coeff *out) {
    int n = 0;
    g2_hom r; r.x = q->x; r.y = q->y; fq2_one(&r.z);
    fq2 nqy; fq2_neg(&nqy, &q->y);
    for (int i = 63; i >= 0; i--) {
        line_double(&r, &out[n++]);
        if (ATE_LOOP_COUNT[i] == 1) line_add(&r, &q->x, &q->y, &out[n++]);
        else if (ATE_LOOP_COUNT[i] == -1) line_add(&r, &q->x, &nqy, &out[n++]);
    }
    fq2 q1x, q1y, q2x, q2y;
    fq2_conj(&q1x, &q->x); fq2_mul(&q1x, &q1x, &TWIST_MUL_BY_Q_X);
    fq2_conj(&q1y, &q->y); fq2_mul(&q1y, &q1y, &TWIST_MUL_BY_Q_Y);
    fq2_conj(&q2x, &q1x); fq2_mul(&q2x, &q2x, &TWIST_MUL_BY_Q_X);
    fq2_conj(&q2y, &q1y); fq2_mul(&q2y, &q2y, &TWIST_MUL_BY_Q_Y); fq2_neg(&q2y, &q2y);
    line_add(&r, &q1x, &q1y, &out[n++]);
    line_add(&r, &q2x, &q2y, &out[n++]);
    return n;
}
static void miller_loop(fq12 *f, const g1_aff *p, const g2_aff *q) {
    fq12_one(f);
    if (p->inf || q->inf) return;
    ell_coeff co[128];
    g2_prepare(q, co);
    int k = 0;
    for (int i = 64; i >= 1; i--) {
        if (i != 64) fq12_sqr(f, f);
        ell(f, &co[k++], p);
        if (ATE_LOOP_COUNT[i - 1] != 0) ell(f, &co[k++], p);
    }
    ell(f, &co[k++], p);
    ell(f, &co[k++], p);
}
static void final_exp(fq12 *out, const fq12 *fin) {
    fq12 f1, f2, r, y0, y1, y2, y3, y4, y5, y6, y7, y8, y9, y10, y11, y12, y13, y14, y15;
    fq12_conj(&f1, fin); fq12_inv(&f2, fin);
    fq12_mul(&r, &f1, &f2); f2 = r;
    fq12_frob(&r, &r, 2); fq12_mul(&r, &r, &f2);
    fq12_exp_by_neg_x(&y0, &r);
    fq12_cyc_sqr(&y1, &y0); fq12_cyc_sqr(&y2, &y1);
    fq12_mul(&y3, &y2, &y1);
    fq12_exp_by_neg_x(&y4, &y3);
    fq12_cyc_sqr(&y5, &y4);
    fq12_exp_by_neg_x(&y6, &y5);
    fq12_conj(&y3, &y3); fq12_conj(&y6, &y6);
    fq12_mul(&y7, &y6, &y4); fq12_mul(&y8, &y7, &y3);
    fq12_mul(&y9, &y8, &y1); fq12_mul(&y10, &y8, &y4); fq12_mul(&y11, &y10, &r);
    fq12_frob(&y12, &y9, 1); fq12_mul(&y13, &y12, &y11);
    fq12_frob(&y8, &y8, 2); fq12_mul(&y14, &y8, &y13);
    fq12_conj(&r, &r); fq12_mul(&y15, &r, &y9); fq12_frob(&y15, &y15, 3);
    fq12_mul(out, &y15, &y14);
}
static void gt_serialize(uint8_t out[384], const fq12 *f) {
    const fq *c = (const fq *)f;  /* c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1 in memory order */
    for (int i = 0; i < 12; i++) { u64 t[4]; fq_from_mont(t, &c[i]); memcpy(out + 32 * i, t, 32); }
}

/* ------------------------------------------------------------------ BLAKE3 (hash mode, XOF) */
static const uint32_t B3_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };
static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define B3G(a, b, c, d, mx, my) \
    s[a] += s[b] + (mx); s[d] = rotr32(s[d] ^ s[a], 16); s[c] += s[d]; s[b] = rotr32(s[b] ^ s[c], 12); \
    s[a] += s[b] + (my); s[d] = rotr32(s[d] ^ s[a], 8);  s[c] += s[d]; s[b] = rotr32(s[b] ^ s[c], 7);
static void b3_compress(uint32_t out[16], const uint32_t cv[8], const uint32_t blk[16], u64 counter, uint32_t blen, uint32_t flags) {
    uint32_t s[16], m[16], t[16];
    memcpy(s, cv, 32); memcpy(s + 8, B3_IV, 16);
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = blen; s[15] = flags;
    memcpy(m, blk, 64);
    for (int r = 0; r < 7; r++) {
        B3G(0, 4, 8, 12, m[0], m[1]) B3G(1, 5, 9, 13, m[2], m[3]) B3G(2, 6, 10, 14, m[4], m[5]) B3G(3, 7, 11, 15, m[6], m[7])
        B3G(0, 5, 10, 15, m[8], m[9]) B3G(1, 6, 11, 12, m[10], m[11]) B3G(2, 7, 8, 13, m[12], m[13]) B3G(3, 4, 9, 14, m[14], m[15])
        if (r != 6) { for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]]; memcpy(m, t, 64); }
    }
    for (int i = 0; i < 8; i++) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
typedef struct { uint32_t cv[8]; uint32_t blk[16]; u64 counter; uint32_t blen, flags; } b3_output;
static void b3_chunk(b3_output *o, const uint8_t *data, size_t len, u64 counter) {
    memcpy(o->cv, B3_IV, 32);
    size_t nblk = len ? (len + 63) / 64 : 1;
    for (size_t i = 0; i < nblk; i++) {
        size_t bl = (i == nblk - 1) ? len - 64 * i : 64;
        uint8_t buf[64] = {0}; memcpy(buf, data + 64 * i, bl);
        uint32_t w[16]; memcpy(w, buf, 64);
        uint32_t fl = (i == 0 ? B3_CHUNK_START : 0);
        if (i == nblk - 1) { memcpy(o->blk, w, 64); o->counter = counter; o->blen = (uint32_t)bl; o->flags = fl | B3_CHUNK_END; return; }
        uint32_t out[16]; b3_compress(out, o->cv, w, counter, 64, fl); memcpy(o->cv, out, 32);
    }
}
static void b3_subtree(b3_output *o, const uint8_t *data, size_t len, u64 chunk0) {
    if (len <= 1024) { b3_chunk(o, data, len, chunk0); return; }
    size_t nch = (len + 1023) / 1024, left = 1;
    while (left * 2 < nch) left *= 2;
    b3_output l, r; uint32_t t[16];
    b3_subtree(&l, data, left * 1024, chunk0);
    b3_subtree(&r, data + left * 1024, len - left * 1024, chunk0 + left);
    b3_compress(t, l.cv, l.blk, l.counter, l.blen, l.flags); memcpy(o->blk, t, 32);
    b3_compress(t, r.cv, r.blk, r.counter, r.blen, r.flags); memcpy(o->blk + 8, t, 32);
    memcpy(o->cv, B3_IV, 32); o->counter = 0; o->blen = 64; o->flags = B3_PARENT;
}
void ref_blake3_xof(const uint8_t *data, size_t len, uint8_t *out, size_t out_len) {
    b3_output o; b3_subtree(&o, data, len, 0);
    for (u64 t = 0; t * 64 < out_len; t++) {
        uint32_t w[16]; b3_compress(w, o.cv, o.blk, t, o.blen, o.flags | B3_ROOT);
        size_t n = out_len - t * 64; if (n > 64) n = 64;
        memcpy(out + 64 * t, w, n);
    }
}

/* ------------------------------------------------------------------ tiny thread fan-out */
typedef struct { void (*fn)(void *, size_t); void *arg; size_t n; size_t next; pthread_mutex_t mu; } fan_t;
static void *fan_worker(void *p) {
    fan_t *f = (fan_t *)p;
    for (;;) {
        pthread_mutex_lock(&f->mu); size_t i = f->next++; pthread_mutex_unlock(&f->mu);
        if (i >= f->n) return NULL;
        f->fn(f->arg, i);
    }
}
static void fan_out(void (*fn)(void *, size_t), void *arg, size_t n, int threads) {
    if (threads <= 1 || n <= 1) { for (size_t i = 0; i < n; i++) fn(arg, i); return; }
    fan_t f = {fn, arg, n, 0, PTHREAD_MUTEX_INITIALIZER};
    pthread_t th[64]; if (threads > 64) threads = 64;
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, fan_worker, &f);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------ layout helpers */
static void load_g1(g1_aff *p, const u64 *w) {
    memcpy(&p->x, w, 32); memcpy(&p->y, w + 4, 32);
    p->inf = fq_is_zero(&p->x) && fq_is_zero(&p->y);
}
static void store_g1(u64 *w, const g1_aff *p) { if (p->inf) memset(w, 0, 64); else { memcpy(w, &p->x, 32); memcpy(w + 4, &p->y, 32); } }
static void load_g2(g2_aff *p, const u64 *w) {
    memcpy(&p->x, w, 64); memcpy(&p->y, w + 8, 64);
    p->inf = fq2_is_zero(&p->x) && fq2_is_zero(&p->y);
}
static void store_g2(u64 *w, const g2_aff *p) { if (p->inf) memset(w, 0, 128); else { memcpy(w, &p->x, 64); memcpy(w + 8, &p->y, 64); } }

/* ------------------------------------------------------------------ MSM */
static int ark_log2(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }  /* ceil log2 */
int ref_window_size(size_t n) { return n < 32 ? 3 : ark_log2(n) * 69 / 100 + 2; }
/* ark-ec make_digits on a canonical 4-limb scalar */
static void make_digits(int32_t *digits, const u64 s[4], int w, int nd) {
    u64 radix = (u64)1 << w, mask = radix - 1, carry = 0;
    for (int i = 0; i < nd; i++) {
        int bit_offset = i * w, u = bit_offset / 64, b = bit_offset % 64;
        u64 buf = (b < 64 - w || u == 3) ? (s[u] >> b) : ((s[u] >> b) | (s[u + 1] << (64 - b)));
        u64 coef = carry + (buf & mask);
        carry = (coef + radix / 2) >> w;
        digits[i] = (int32_t)((int64_t)coef - (int64_t)(carry << w));
    }
    digits[nd - 1] += (int32_t)(carry << w);
}
void ref_make_digits(const u64 *scalar_canonical, int w, int32_t *digits) { make_digits(digits, scalar_canonical, w, (254 + w - 1) / w); }

typedef struct { g1_win_job *jobs; } g1_msm_ctx;
static void g1_win_fn(void *a, size_t i) { g1_window_sum(&((g1_win_job *)a)[i]); }
typedef struct { g2_win_job *jobs; } g2_msm_ctx;
static void g2_win_fn(void *a, size_t i) { g2_window_sum(&((g2_win_job *)a)[i]); }

/* out: affine u64[8] (identity = zeros). scalars: Montgomery Fr limbs as ark-ff holds them. */
void ref_msm_g1(const u64 *points, const u64 *scalars, size_t n, u64 *out_aff, int threads) {
    g1_aff res; res.inf = 1;
    if (n == 0) { store_g1(out_aff, &res); return; }
    int c = ref_window_size(n), nd = (254 + c - 1) / c;
    g1_aff *bases = (g1_aff *)malloc(n * sizeof(g1_aff));
    int32_t *digits = (int32_t *)malloc(n * nd * sizeof(int32_t));
    for (size_t i = 0; i < n; i++) {
        load_g1(&bases[i], points + 8 * i);
        u64 k[4]; fr_from_mont(k, scalars + 4 * i);  /* into_bigint */
        make_digits(digits + i * nd, k, c, nd);
    }
    g1_win_job *jobs = (g1_win_job *)malloc(nd * sizeof(g1_win_job));
    for (int w = 0; w < nd; w++) { jobs[w].bases = bases; jobs[w].digits = digits; jobs[w].n = n; jobs[w].nd = nd; jobs[w].c = c; jobs[w].w = w; }
    fan_out(g1_win_fn, jobs, nd, threads);
    g1_jac total; g1_set_inf(&total);
    for (int w = nd - 1; w >= 1; w--) {
        g1_add(&total, &total, &jobs[w].out);
        for (int k = 0; k < c; k++) g1_dbl(&total, &total);
    }
    g1_add(&total, &total, &jobs[0].out);
    g1_to_aff(&res, &total);
    store_g1(out_aff, &res);
    free(jobs); free(digits); free(bases);
}
void ref_msm_g2(const u64 *points, const u64 *scalars, size_t n, u64 *out_aff, int threads) {
    g2_aff res; res.inf = 1;
    if (n == 0) { store_g2(out_aff, &res); return; }
    int c = ref_window_size(n), nd = (254 + c - 1) / c;
    g2_aff *bases = (g2_aff *)malloc(n * sizeof(g2_aff));
    int32_t *digits = (int32_t *)malloc(n * nd * sizeof(int32_t));
    for (size_t i = 0; i < n; i++) {
        load_g2(&bases[i], points + 16 * i);
        u64 k[4]; fr_from_mont(k, scalars + 4 * i);
        make_digits(digits + i * nd, k, c, nd);
    }
    g2_win_job *jobs = (g2_win_job *)malloc(nd * sizeof(g2_win_job));
    for (int w = 0; w < nd; w++) { jobs[w].bases = bases; jobs[w].digits = digits; jobs[w].n = n; jobs[w].nd = nd; jobs[w].c = c; jobs[w].w = w; }
    fan_out(g2_win_fn, jobs, nd, threads);
    g2_jac total; g2_set_inf(&total);
    for (int w = nd - 1; w >= 1; w--) {
        g2_add(&total, &total, &jobs[w].out);
        for (int k = 0; k < c; k++) g2_dbl(&total, &total);
    }
    g2_add(&total, &total, &jobs[0].out);
    g2_to_aff(&res, &total);
    store_g2(out_aff, &res);
    free(jobs); free(digits); free(bases);
}

/* ------------------------------------------------------------------ batched scalar mult (n independent outputs) */
typedef struct { const u64 *pts; const u64 *sc; u64 *out; int pt_stride; size_t chunk, n; } mulb_t;
static void g1_mulb_fn(void *a, size_t ci) {
    mulb_t *m = (mulb_t *)a;
    size_t lo = ci * m->chunk, hi = lo + m->chunk; if (hi > m->n) hi = m->n;
    for (size_t i = lo; i < hi; i++) {
        g1_aff p, r; g1_jac j, o; u64 k[4];
        load_g1(&p, m->pts + (size_t)m->pt_stride * i); g1_from_aff(&j, &p);
        fr_from_mont(k, m->sc + 4 * i);
        g1_mul(&o, &j, k); g1_to_aff(&r, &o); store_g1(m->out + 8 * i, &r);
    }
}
static void g2_mulb_fn(void *a, size_t ci) {
    mulb_t *m = (mulb_t *)a;
    size_t lo = ci * m->chunk, hi = lo + m->chunk; if (hi > m->n) hi = m->n;
    for (size_t i = lo; i < hi; i++) {
        g2_aff p, r; g2_jac j, o; u64 k[4];
        load_g2(&p, m->pts + (size_t)m->pt_stride * i); g2_from_aff(&j, &p);
        fr_from_mont(k, m->sc + 4 * i);
        g2_mul(&o, &j, k); g2_to_aff(&r, &o); store_g2(m->out + 16 * i, &r);
    }
}
/* pt_stride = 8 (16 for G2) for per-item points, 0 to broadcast points[0] (fixed base). */
void ref_g1_mul_batch(const u64 *points, int pt_stride, const u64 *scalars, size_t n, u64 *out, int threads) {
    mulb_t m = {points, scalars, out, pt_stride, 64, n};
    fan_out(g1_mulb_fn, &m, (n + 63) / 64, threads);
}
void ref_g2_mul_batch(const u64 *points, int pt_stride, const u64 *scalars, size_t n, u64 *out, int threads) {
    mulb_t m = {points, scalars, out, pt_stride, 64, n};
    fan_out(g2_mulb_fn, &m, (n + 63) / 64, threads);
}
/* sum of n affine points (used to combine per-GPU partials in tests) */
void ref_g1_sum(const u64 *points, size_t n, u64 *out) {
    g1_jac acc; g1_set_inf(&acc);
    for (size_t i = 0; i < n; i++) { g1_aff p; load_g1(&p, points + 8 * i); g1_add_mixed(&acc, &acc, &p); }
    g1_aff r; g1_to_aff(&r, &acc); store_g1(out, &r);
}
int ref_g1_on_curve(const u64 *pt) {
    g1_aff p; load_g1(&p, pt); if (p.inf) return 1;
    fq l, r; fq_sqr(&l, &p.y); fq_sqr(&r, &p.x); fq_mul(&r, &r, &p.x); fq_add(&r, &r, &FQ_B1);
    return fq_eq(&l, &r);
}
int ref_g2_on_curve(const u64 *pt) {
    g2_aff p; load_g2(&p, pt); if (p.inf) return 1;
    fq2 l, r; fq2_sqr(&l, &p.y); fq2_sqr(&r, &p.x); fq2_mul(&r, &r, &p.x); fq2_add(&r, &r, &FQ2_B2);
    return fq2_eq(&l, &r);
}
void ref_generators(u64 *g1, u64 *g2) {
    memcpy(g1, &G1_GEN_X, 32); memcpy(g1 + 4, &G1_GEN_Y, 32);
    memcpy(g2, &G2_GEN_X, 64); memcpy(g2 + 8, &G2_GEN_Y, 64);
}

/* ------------------------------------------------------------------ pairing batch */
typedef struct { const u64 *p; const u64 *q; int q_stride; uint8_t *out; size_t chunk, n; } pairb_t;
static void pair_fn(void *a, size_t ci) {
    pairb_t *m = (pairb_t *)a;
    size_t lo = ci * m->chunk, hi = lo + m->chunk; if (hi > m->n) hi = m->n;
    for (size_t i = lo; i < hi; i++) {
        g1_aff p; g2_aff q; fq12 f, e;
        load_g1(&p, m->p + 8 * i); load_g2(&q, m->q + (size_t)m->q_stride * i);
        miller_loop(&f, &p, &q); final_exp(&e, &f); gt_serialize(m->out + 384 * i, &e);
    }
}
/* q_stride = 16 for per-item Q, 0 to use q[0] for all (the encapsulate case; G2Prepared is still
 * rebuilt per item, as src/kem.rs:30 does). */
void ref_pairing_batch(const u64 *g1, const u64 *g2, int q_stride, size_t n, uint8_t *gt_out, int threads) {
    pairb_t m = {g1, g2, q_stride, gt_out, 8, n};
    fan_out(pair_fn, &m, (n + 7) / 8, threads);
}
/* raw Miller-loop output (Montgomery limbs, 48 u64) for debugging device kernels stage by stage */
void ref_miller_loop_raw(const u64 *g1, const u64 *g2, u64 *out48) {
    g1_aff p; g2_aff q; fq12 f; load_g1(&p, g1); load_g2(&q, g2); miller_loop(&f, &p, &q); memcpy(out48, &f, 384);
}
void ref_final_exp_raw(const u64 *in48, u64 *out48) { fq12 f, e; memcpy(&f, in48, 384); final_exp(&e, &f); memcpy(out48, &e, 384); }

/* ------------------------------------------------------------------ KEM batch (src/kem.rs looped as src/vec.rs) */
typedef struct {
    g1_aff com; g2_aff tau_g2; const u64 *points, *values, *rs; u64 *ct_out; uint8_t *gt_out, *key_out; size_t msg_len, chunk, n;
} encb_t;
static void encap_fn(void *a, size_t ci) {
    encb_t *m = (encb_t *)a;
    size_t lo = ci * m->chunk, hi = lo + m->chunk; if (hi > m->n) hi = m->n;
    g1_aff g1; g1.x = G1_GEN_X; g1.y = G1_GEN_Y; g1.inf = 0;
    g2_aff g2; g2.x = G2_GEN_X; g2.y = G2_GEN_Y; g2.inf = 0;
    g1_jac g1j; g1_from_aff(&g1j, &g1); g2_jac g2j; g2_from_aff(&g2j, &g2);
    for (size_t i = lo; i < hi; i++) {
        u64 kv[4], kp[4], kr[4];
        fr_from_mont(kv, m->values + 4 * i); fr_from_mont(kp, m->points + 4 * i); fr_from_mont(kr, m->rs + 4 * i);
        /* com_beta = com - g1*value  (src/kem.rs:22) */
        g1_jac t, cb; g1_mul(&t, &g1j, kv); g1_neg(&t, &t); g1_add_mixed(&cb, &t, &m->com);
        /* secret = e(com_beta * r, g2)  (src/kem.rs:30) */
        g1_jac cbr; g1_mul(&cbr, &cb, kr); g1_aff pa; g1_to_aff(&pa, &cbr);
        fq12 f, e; miller_loop(&f, &pa, &g2); final_exp(&e, &f);
        uint8_t *gt = m->gt_out + 384 * i; gt_serialize(gt, &e);
        /* ct = r * ([tau]_2 - g2*point)  (src/kem.rs:36-37) */
        g2_jac u, ta, ct; g2_mul(&u, &g2j, kp); g2_neg(&u, &u); g2_add_mixed(&ta, &u, &m->tau_g2);
        g2_mul(&ct, &ta, kr); g2_aff cta; g2_to_aff(&cta, &ct); store_g2(m->ct_out + 16 * i, &cta);
        if (m->key_out) ref_blake3_xof(gt, 384, m->key_out + m->msg_len * i, m->msg_len);
    }
}
void ref_encap_batch(const u64 *com_aff, const u64 *tau_g2_aff, const u64 *points, const u64 *values, const u64 *rs,
                     size_t n, u64 *ct_out, uint8_t *gt_out, uint8_t *key_out, size_t msg_len, int threads) {
    encb_t m; load_g1(&m.com, com_aff); load_g2(&m.tau_g2, tau_g2_aff);
    m.points = points; m.values = values; m.rs = rs; m.ct_out = ct_out; m.gt_out = gt_out; m.key_out = key_out;
    m.msg_len = msg_len; m.chunk = 4; m.n = n;
    fan_out(encap_fn, &m, (n + 3) / 4, threads);
}
void ref_decap_batch(const u64 *proofs, const u64 *cts, size_t n, uint8_t *gt_out, uint8_t *key_out, size_t msg_len, int threads) {
    ref_pairing_batch(proofs, cts, 16, n, gt_out, threads);
    if (key_out) for (size_t i = 0; i < n; i++) ref_blake3_xof(gt_out + 384 * i, 384, key_out + msg_len * i, msg_len);
}

/* ------------------------------------------------------------------ misc scalar helpers for tests */
void ref_fr_to_mont(const u64 *canon, u64 *mont, size_t n) { for (size_t i = 0; i < n; i++) fr_to_mont(mont + 4 * i, canon + 4 * i); }
void ref_fr_from_mont(const u64 *mont, u64 *canon, size_t n) { for (size_t i = 0; i < n; i++) fr_from_mont(canon + 4 * i, mont + 4 * i); }
void ref_fq_from_mont(const u64 *mont, u64 *canon, size_t n) { for (size_t i = 0; i < n; i++) fq_from_mont(canon + 4 * i, (const fq *)(mont + 4 * i)); }
void ref_fq_to_mont(const u64 *canon, u64 *mont, size_t n) { for (size_t i = 0; i < n; i++) mont_mul(mont + 4 * i, canon + 4 * i, FQ_R2, FQ_MOD, FQ_INV); }
/* sum_i a_i * b_i mod r on Montgomery inputs -> Montgomery output (expected-MSM-scalar helper) */
void ref_fr_dot(const u64 *a, const u64 *b, size_t n, u64 *out) {
    u64 acc[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) { u64 t[4]; fr_mul(t, a + 4 * i, b + 4 * i); mod_add(acc, acc, t, FR_MOD); }
    memcpy(out, acc, 32);
}

/* ------------------------------------------------------------------ scalar-field polynomial helpers (checkers of vec_commit / open / open_fk)
 * ark-poly 0.4.2 (Cargo.lock of the reference; not vendored) Radix2EvaluationDomain::fft / ifft as keaki calls them at src/vec.rs:36-37
 * and src/kzg.rs:182-200: out[j] = sum_i in[i] omega^(i j) over the size-n domain (n = 2^log2n, omega = its group_gen); ifft = the same
 * with omega^-1, then times n^-1. In place; data and omega are Montgomery limbs. Plain radix-2 decimation in time. */
void ref_fr_fft(u64 *data, int log2n, const u64 *omega) {
    size_t n = (size_t)1 << log2n;
    for (size_t i = 1, j = 0; i < n; i++) {                 /* bit reversal */
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { u64 t[4]; memcpy(t, data + 4 * i, 32); memcpy(data + 4 * i, data + 4 * j, 32); memcpy(data + 4 * j, t, 32); }
    }
    u64 one_m[4], one_c[4] = {1, 0, 0, 0};
    fr_to_mont(one_m, one_c);
    for (size_t len = 2; len <= n; len <<= 1) {
        u64 w[4]; memcpy(w, omega, 32);
        for (size_t k = len; k < n; k <<= 1) fr_mul(w, w, w);      /* omega^(n/len) */
        for (size_t i = 0; i < n; i += len) {
            u64 x[4]; memcpy(x, one_m, 32);
            for (size_t j = 0; j < len / 2; j++) {
                u64 *a = data + 4 * (i + j), *b = data + 4 * (i + j + len / 2), v[4], u[4];
                memcpy(u, a, 32);
                fr_mul(v, b, x);
                mod_add(a, u, v, FR_MOD);
                mod_sub(b, u, v, FR_MOD);
                fr_mul(x, x, w);
            }
        }
    }
}
void ref_fr_scale(u64 *data, size_t n, const u64 *k) { for (size_t i = 0; i < n; i++) fr_mul(data + 4 * i, data + 4 * i, k); }
/* `open` of the reference, src/kzg.rs:109-120: value = p(point), quotient = (p(x) - value) / (x - point) by synthetic division
 * (what DensePolynomial::div computes for a monic linear divisor). coeffs: n Fr low degree first; q_out: n - 1 Fr; all Montgomery. */
void ref_fr_quotient(const u64 *coeffs, size_t n, const u64 *point, u64 *q_out, u64 *value_out) {
    u64 carry[4] = {0, 0, 0, 0};
    for (size_t k = n; k-- > 0;) {
        u64 t[4];
        fr_mul(t, carry, point);
        mod_add(carry, t, coeffs + 4 * k, FR_MOD);
        if (k >= 1) memcpy(q_out + 4 * (k - 1), carry, 32);
    }
    if (value_out) memcpy(value_out, carry, 32);
}
