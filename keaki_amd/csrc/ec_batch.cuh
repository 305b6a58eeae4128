// Batched (n independent outputs) scalar multiplication in G1 / G2 and the group part of
// kem::encapsulate. One lane per item, Jacobian double-and-add from the top bit.
//
// Replaces `.mul(scalar)` at reference src/kem.rs:22,30,36,37 and src/kzg.rs:57,60,135,144 as they
// occur inside the loops of src/vec.rs:63-66.
#pragma once
#include "bn254_curve.cuh"

namespace bn254 {

// k * P, k given as Montgomery Fr. MSB-first; the 256-bit scalar is shifted left one bit per step so
// register indices stay static.
template <class F>
KDEV Jac<F> scalar_mul(const Aff<F>& p, const Fr& k_mont) {
  u32 v[8];
  fp_from_mont<FrParams>(v, k_mont);
  Jac<F> acc = jac_inf<F>();
  if (aff_is_inf(p)) return acc;
  // skip the two always-zero top bits (r < 2^254)
#pragma unroll
  for (int s = 0; s < 2; s++) {
#pragma unroll
    for (int j = 7; j > 0; j--) v[j] = (v[j] << 1) | (v[j - 1] >> 31);
    v[0] <<= 1;
  }
#pragma unroll 1
  for (int i = 0; i < 254; i++) {
    acc = jac_dbl(acc);
    if (v[7] >> 31) acc = jac_add_mixed(acc, p);
#pragma unroll
    for (int j = 7; j > 0; j--) v[j] = (v[j] << 1) | (v[j - 1] >> 31);
    v[0] <<= 1;
  }
  return acc;
}

template <class F>
__global__ void __launch_bounds__(64) k_mul_batch(const Aff<F>* __restrict__ pts, int stride, const Fr* __restrict__ scalars, u32 n,
                                                  Aff<F>* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p = pts[(size_t)i * stride];
  out[i] = jac_to_aff(scalar_mul(p, scalars[i]));
}

// encapsulate, G1 side (src/kem.rs:22,30): out[i] = r[i] * (com - values[i] * g1)   (affine)
static __global__ void __launch_bounds__(64) k_encap_g1(const G1Aff* __restrict__ com, const Fr* __restrict__ values, const Fr* __restrict__ rs,
                                                 u32 n, G1Aff* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Aff g = {G1_GEN_X, G1_GEN_Y};
  G1Jac t = scalar_mul(g, values[i]);
  t.y = -t.y;
  G1Aff cb = jac_to_aff(jac_add_mixed(t, *com));
  out[i] = jac_to_aff(scalar_mul(cb, rs[i]));
}
// encapsulate, G2 side (src/kem.rs:36-37): ct[i] = r[i] * (tau_g2 - points[i] * g2)   (affine)
static __global__ void __launch_bounds__(64) k_encap_g2(const G2Aff* __restrict__ tau_g2, const Fr* __restrict__ points, const Fr* __restrict__ rs,
                                                 u32 n, G2Aff* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G2Aff g = {G2_GEN_X, G2_GEN_Y};
  G2Jac t = scalar_mul(g, points[i]);
  t.y = -t.y;
  G2Aff ta = jac_to_aff(jac_add_mixed(t, *tau_g2));
  out[i] = jac_to_aff(scalar_mul(ta, rs[i]));
}


// ------------------------------------------------------------------------------------------------
// Fixed-base path for encapsulate. In the batch loop of src/vec.rs:63-66 every item uses the SAME
// bases: g1, g2 (generators), C (the commitment) and [tau]_2. So
//     r (C - beta g1)       = r C + (-(r beta)) g1          (src/kem.rs:22,30)
//     r ([tau]_2 - alpha g2) = r [tau]_2 + (-(r alpha)) g2   (src/kem.rs:36-37)
// are sums of two FIXED-base multiples: with 8-bit window tables T[j][d] = d 2^(8j) B (32 x 256
// affine entries per base, built once per batch by k_mul_batch) each costs 32 mixed additions and
// no doublings, instead of a 254-step double-and-add ladder per scalar-mult.
// ------------------------------------------------------------------------------------------------
constexpr u32 FB_WINDOWS = 32, FB_ENTRIES = 256;

// scalars[j * 256 + d] = Montgomery(d * 2^(8j)) (0 when the value is >= r: never indexed, scalars are < r)
static __global__ void __launch_bounds__(256) k_fb_table_scalars(Fr* __restrict__ out) {
  u32 j = blockIdx.x, d = threadIdx.x;
  u32 v[8];
#pragma unroll
  for (int t = 0; t < 8; t++) v[t] = 0;
  u32 bit = 8 * j;
#pragma unroll
  for (int t = 0; t < 8; t++) if ((bit >> 5) == (u32)t) v[t] = d << (bit & 31);   // 8-bit window never straddles a word
  // >= r ?
  bool ge = true;
#pragma unroll
  for (int t = 7; t >= 0; t--) {
    if (v[t] != FrParams::MOD[t]) { ge = v[t] > FrParams::MOD[t]; break; }
  }
  if (ge) {
#pragma unroll
    for (int t = 0; t < 8; t++) v[t] = 0;
  }
  out[j * FB_ENTRIES + d] = fp_to_mont<FrParams>(v);
}

// acc += T[j][byte_j(k)] for all windows; k canonical (consumed)
template <class F>
KDEV Xyzz<F> fb_accumulate(Xyzz<F> acc, const Aff<F>* __restrict__ table, u32* v) {
#pragma unroll 1
  for (u32 j = 0; j < FB_WINDOWS; j++) {
    u32 d = v[0] & 255u;
#pragma unroll
    for (int t = 0; t < 7; t++) v[t] = (v[t] >> 8) | (v[t + 1] << 24);
    v[7] >>= 8;
    if (d) acc = xyzz_add_mixed(acc, table[j * FB_ENTRIES + d]);
  }
  return acc;
}

// out[i] = r_i * BaseA + (-(r_i * x_i)) * BaseB   with tables for BaseA (C or [tau]_2) and BaseB (g1 or g2)
template <class F>
__global__ void __launch_bounds__(64) k_encap_fixed(const Aff<F>* __restrict__ tab_a, const Aff<F>* __restrict__ tab_b,
                                                    const Fr* __restrict__ xs, const Fr* __restrict__ rs, u32 n, Aff<F>* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr r = rs[i];
  Fr t = fp_neg<FrParams>(fp_mul<FrParams>(r, xs[i]));
  u32 u[8], v[8];
  fp_from_mont<FrParams>(u, r);
  fp_from_mont<FrParams>(v, t);
  Xyzz<F> acc = xyzz_inf<F>();
  acc = fb_accumulate(acc, tab_a, u);
  acc = fb_accumulate(acc, tab_b, v);
  out[i] = xyzz_to_aff(acc);
}


// ---- on-curve check of affine points (SRS ingest, reference src/kzg/ptau.rs:266,314 deserialises *unchecked*) ----------------
// counts points with y^2 != x^3 + b; (0, 0) is the identity and passes. b = 3 on G1, 3/(9+u) on the twist.
KDEV Fq curve_b(const Fq*) {
  Fq three = fq_one();
  three = three + three + three;
  return three;
}
KDEV Fq2 curve_b(const Fq2*) { return G2_B; }
template <class F>
__global__ void __launch_bounds__(256) k_curve_check(const Aff<F>* __restrict__ pts, u32 n, unsigned long long* __restrict__ bad,
                                                     unsigned long long* __restrict__ first_bad) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p = pts[i];
  if (aff_is_inf(p)) return;
  F lhs = f_sqr(p.y), rhs = f_sqr(p.x) * p.x + curve_b((const F*)nullptr);
  if (!f_eq(lhs, rhs)) {
    atomicAdd(bad, 1ull);
    atomicMin(first_bad, (unsigned long long)i);
  }
}
}  // namespace bn254
