// Batched (n independent outputs) scalar multiplication in G1 / G2 and the group part of
// kem::encapsulate. One lane per item, Jacobian double-and-add from the top bit.
//
// Replaces `.mul(scalar)` at reference src/kem.rs:22,30,36,37 and src/kzg.rs:57,60,135,144 as they
// occur inside the loops of src/vec.rs:63-66.
#pragma once
#include "bn254_curve.hip.h"
#include <type_traits>
#include "jac29.hip.h"
#include "xyzz29_g2.hip.h"

namespace bn254 {

// k * P, k given as Montgomery Fr. MSB-first; the 256-bit scalar is shifted left one bit per step so
// register indices stay static.
template <class F>
KDEV Jac<F> scalar_mul_sat(const Aff<F>& p, const Fr& k_mont) {
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

// G2 (and the reference ladder of the self-test): the saturated double-and-add above. G1: the NAF ladder in 29-bit limbs (jac29.hip.h).
template <class F> KDEV Jac<F> scalar_mul(const Aff<F>& p, const Fr& k_mont) { return scalar_mul_sat(p, k_mont); }
KDEV Jac<Fq> scalar_mul(const Aff<Fq>& p, const Fr& k_mont) { return jac_scalar_mul_u29(jac_from_aff(p), k_mont); }

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

// ------------------------------------------------------------------------------------------------
// Fixed-base path for encapsulate. In the batch loop of src/vec.rs:63-66 every item uses the SAME
// bases: g1, g2 (generators), C (the commitment) and [tau]_2. So
//     r (C - beta g1)       = r C + (-(r beta)) g1          (src/kem.rs:22,30)
//     r ([tau]_2 - alpha g2) = r [tau]_2 + (-(r alpha)) g2   (src/kem.rs:36-37)
// are sums of two FIXED-base multiples: with signed 13-bit window tables T[j][d] = d 2^(13j) B, d = 1..4096 (20 x 4097
// affine entries per base, built by k_fb_window_bases + k_fb_table_entries: a ladder over the bits of d; negative digits negate y) each costs at most 20 mixed additions and
// no doublings, instead of a 254-step double-and-add ladder per scalar-mult.
// ------------------------------------------------------------------------------------------------
// window width per table: 16 bits (16 windows x 32768 entries) for bases that outlive a batch (the generators: per context; [tau]_2: per
// setup), 13 bits (20 x 4096) for the commitment's table, rebuilt per batch. entries = 2^(wb-1) + 1 slots per window, slot 0 unused.
struct FbShape { u32 wb, windows, entries; };
__host__ __device__ inline FbShape fb_shape(u32 wb) { return {wb, (254u + wb - 1u) / wb + ((254u % wb) == 0u ? 1u : 0u), (1u << (wb - 1)) + 1u}; }

// Table build: T[j][d] = d * base_j with base_j = 2^(wb j) * base. Two launches: the window bases (lane j doubles wb * j times: the longest lane
// is the only chain there is), then one lane per entry with a ladder over the BITS OF d -- at most wb - 1 doublings and additions. (Until round 4
// an entry was a full 254-bit scalar multiplication by d 2^(wb j) mod r: 20 ms for a 16-bit G2 table, 2 x that in the first call of a context.)
template <class F>
__global__ void __launch_bounds__(64) k_fb_window_bases(const Aff<F>* __restrict__ base, FbShape g, Aff<F>* __restrict__ out) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= g.windows) return;
  Jac<F> a = jac_from_aff(*base);
  for (u32 t = 0; t < g.wb * j; t++) a = jac_dbl(a);
  out[j] = jac_to_aff(a);
}
template <class F>
__global__ void __launch_bounds__(64) k_fb_table_entries(const Aff<F>* __restrict__ bases, FbShape g, Aff<F>* __restrict__ table) {
  const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.windows * g.entries) return;
  const u32 j = idx / g.entries, d = idx % g.entries;
  const Aff<F> p = bases[j];
  Jac<F> acc = jac_inf<F>();
  if (d && !aff_is_inf(p)) {
#pragma unroll 1
    for (int b = 31 - __clz((int)d); b >= 0; b--) {
      acc = jac_dbl(acc);
      if ((d >> b) & 1u) acc = jac_add_mixed(acc, p);
    }
  }
  table[idx] = jac_to_aff(acc);                      // d = 0: the identity (the slot is never indexed)
}

// acc += sign_j * T[j][|digit_j(k)|] for all windows, signed wb-bit digits in (-2^(wb-1), 2^(wb-1)]; k canonical (consumed)
template <class F>
KDEV Xyzz<F> fb_accumulate(Xyzz<F> acc, const Aff<F>* __restrict__ table, FbShape g, u32* v) {
  u32 carry = 0;
  const u32 half = 1u << (g.wb - 1);
#pragma unroll 1
  for (u32 j = 0; j < g.windows; j++) {
    u32 d = (v[0] & (2u * half - 1u)) + carry;
#pragma unroll
    for (int t = 0; t < 7; t++) v[t] = (v[t] >> g.wb) | (v[t + 1] << (32u - g.wb));
    v[7] >>= g.wb;
    const bool neg = d > half;                       // d - 2^wb and a carry (2^wb itself: digit 0, carry 1)
    carry = neg ? 1u : 0u;
    if (neg) d = 2u * half - d;
    if (d) acc = xyzz_add_mixed(acc, aff_cneg(table[(size_t)j * g.entries + d], neg));
  }
  return acc;
}

// the same walk over the windows with the accumulator in the lazy limbs of xyzz29_g2.hip.h (G2: the ciphertext side of `encapsulate`)
KDEV void fb_accumulate_g2_u29(X29G2& acc, const Aff<Fq2>* __restrict__ table, FbShape g, u32* v) {
  u32 carry = 0;
  const u32 half = 1u << (g.wb - 1);
#pragma unroll 1
  for (u32 j = 0; j < g.windows; j++) {
    u32 d = (v[0] & (2u * half - 1u)) + carry;
#pragma unroll
    for (int t = 0; t < 7; t++) v[t] = (v[t] >> g.wb) | (v[t + 1] << (32u - g.wb));
    v[7] >>= g.wb;
    const bool neg = d > half;
    carry = neg ? 1u : 0u;
    if (neg) d = 2u * half - d;
    if (d) x29g2_add_mixed(acc, aff_cneg(table[(size_t)j * g.entries + d], neg));
  }
}

// out[i] = r_i * BaseA + (-(r_i * x_i)) * BaseB   with tables for BaseA (C or [tau]_2) and BaseB (g1 or g2)
template <class F, int OCC>
__global__ void __launch_bounds__(64, OCC) k_encap_fixed(const Aff<F>* __restrict__ tab_a, FbShape ga, const Aff<F>* __restrict__ tab_b, FbShape gb,
                                                    const Fr* __restrict__ xs, const Fr* __restrict__ rs, u32 n, Aff<F>* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr r = rs[i];
  Fr t = fp_neg<FrParams>(fp_mul<FrParams>(r, xs[i]));
  u32 u[8], v[8];
  fp_from_mont<FrParams>(u, r);
  fp_from_mont<FrParams>(v, t);
  if constexpr (std::is_same<F, Fq2>::value) {
    X29G2 acc = x29g2_inf();
    fb_accumulate_g2_u29(acc, tab_a, ga, u);
    fb_accumulate_g2_u29(acc, tab_b, gb, v);
    out[i] = xyzz_to_aff(x29g2_store(acc));
  } else {
    Xyzz<F> acc = xyzz_inf<F>();
    acc = fb_accumulate(acc, tab_a, ga, u);
    acc = fb_accumulate(acc, tab_b, gb, v);
    out[i] = xyzz_to_aff(acc);
  }
}

// ---- the same sum for a FEW items: sixteen lanes per item ------------------------------------------------------------------------------
// k_encap_fixed adds an item's 32-64 table entries one after the other in one lane: 0.96 ms for a single `encapsulate`, the device idle.
// Here lane l of a 16-lane row takes the windows l, l + 16, ... of the two tables' concatenated window list (the digits come out of a cheap
// serial walk every lane runs; the ADDITIONS of a step happen in all lanes at once), then the sixteen partial sums meet in a four-level tree of
// general additions through LDS (x29g2_add, chunk-major layout: conflict-free 16-byte accesses). Same table entries, same group element, same
// affine bytes; 2-4 mixed additions + 4 general ones + the conversion on the critical path instead of 32-64 + the conversion.
constexpr u32 FBW_SLOTS = 5;                      // ceil((33 + 33) / 16): 8-bit tables have 32-33 windows each
struct FbPick { u32 d[FBW_SLOTS]; u32 neg; };     // digit of slot t (0: nothing to add), bit t of neg: negative
KDEV void fb_pick_digits(FbPick& pk, FbShape g, u32* v, u32 first_window, u32 lane16) {
  u32 carry = 0;
  const u32 half = 1u << (g.wb - 1);
#pragma unroll 1
  for (u32 j = 0; j < g.windows; j++) {
    u32 d = (v[0] & (2u * half - 1u)) + carry;
#pragma unroll
    for (int t = 0; t < 7; t++) v[t] = (v[t] >> g.wb) | (v[t + 1] << (32u - g.wb));
    v[7] >>= g.wb;
    const bool neg = d > half;
    carry = neg ? 1u : 0u;
    if (neg) d = 2u * half - d;
    const u32 w = first_window + j;
    if ((w & 15u) == lane16) {
      const u32 slot = w >> 4;
#pragma unroll
      for (u32 t = 0; t < FBW_SLOTS; t++) if (t == slot) pk.d[t] = d;
      if (neg) pk.neg |= 1u << slot;
    }
  }
}
static __global__ void __launch_bounds__(64) k_encap_fixed_g2_wide(const Aff<Fq2>* __restrict__ tab_a, FbShape ga, const Aff<Fq2>* __restrict__ tab_b, FbShape gb,
                                                                   const Fr* __restrict__ xs, const Fr* __restrict__ rs, u32 n, Aff<Fq2>* __restrict__ out) {
  __shared__ uint4 sh[16 * 64];                   // one XYZZ point (4 Fq2 = 16 x 16 bytes) per lane, chunk c of lane l at sh[c * 64 + l]
  const u32 lane = threadIdx.x, l16 = lane & 15u;
  const u32 item = blockIdx.x * 4u + (lane >> 4);
  const bool live = item < n;
  const u32 i = live ? item : n - 1;
  const Fr r = rs[i];
  const Fr t = fp_neg<FrParams>(fp_mul<FrParams>(r, xs[i]));
  u32 u[8], v[8];
  fp_from_mont<FrParams>(u, r);
  fp_from_mont<FrParams>(v, t);
  FbPick pk;
#pragma unroll
  for (u32 s = 0; s < FBW_SLOTS; s++) pk.d[s] = 0;
  pk.neg = 0;
  fb_pick_digits(pk, ga, u, 0, l16);
  fb_pick_digits(pk, gb, v, ga.windows, l16);
  X29G2 acc = x29g2_inf();
#pragma unroll 1
  for (u32 s = 0; s < FBW_SLOTS; s++) {
    const u32 w = s * 16u + l16;                   // this lane's window of the step: below ga.windows table A, else table B
    u32 d = 0;
#pragma unroll
    for (u32 q = 0; q < FBW_SLOTS; q++) if (q == s) d = pk.d[q];
    if (w >= ga.windows + gb.windows) d = 0;
    if (d) {
      const bool in_a = w < ga.windows;
      const Aff<Fq2>* e = in_a ? tab_a + (size_t)w * ga.entries + d : tab_b + (size_t)(w - ga.windows) * gb.entries + d;
      x29g2_add_mixed(acc, aff_cneg(*e, ((pk.neg >> s) & 1u) != 0));
    }
  }
  // tree over the sixteen lanes of the row
#pragma unroll 1
  for (u32 step = 1; step < 16u; step <<= 1) {
    const Xyzz<Fq2> mine = x29g2_store(acc);
    const uint4* mw = reinterpret_cast<const uint4*>(&mine);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; c++) sh[c * 64 + lane] = mw[c];
    __syncthreads();
    Xyzz<Fq2> other;
    uint4* ow = reinterpret_cast<uint4*>(&other);
    const u32 src = lane ^ step;                   // inside the row: step < 16
#pragma unroll
    for (int c = 0; c < 16; c++) ow[c] = sh[c * 64 + src];
    acc = x29g2_add(acc, x29g2_load(other));       // every lane adds (lanes that are not tree nodes carry values nobody reads)
  }
  const Aff<Fq2> res = xyzz_to_aff(x29g2_store(acc));
  if (live && l16 == 0) out[item] = res;
}
// ---- kzg verify in the reference's own form (src/kzg.rs:135-143): e(com - value g1, g2) == e(proof, [tau]_2 - point g2) ----------------------
// Both inner points are a caller's point plus a FIXED-base multiple: with the 8-bit window tables of g1 and g2 and sixteen lanes per sum they cost
// 2-3 additions, a four-level tree and one conversion each (0.25 ms), where moving point * proof across the pairing (the form of rounds 2-4)
// needs a variable-base ladder of 129 doublings in one lane (0.99 ms). Wave 0: Q = [tau]_2 + (-point) g2. Wave 1: A = com + (-value) g1.
// Orders the LDS traffic of ONE wave: its DS operations execute in order, so a row written by all lanes is complete for every lane's read once
// the counter has drained; the clobber keeps the compiler from moving accesses across. The two waves of k_verify_points run DIFFERENT
// code on disjoint LDS (sh1 / sh2): a workgroup barrier inside their arms would only work while both arms happen to execute the same number of
// barriers -- none is needed.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
static __global__ void __launch_bounds__(128) k_verify_points(const Aff<Fq>* __restrict__ tab_g1, const Aff<Fq2>* __restrict__ tab_g2, FbShape g,
                                                              const G1Aff* __restrict__ com, const G2Aff* __restrict__ tau_g2, const Fr* __restrict__ value,
                                                              const Fr* __restrict__ point, G1Aff* __restrict__ out_a, G2Aff* __restrict__ out_q) {
  __shared__ uint4 sh2[16 * 64];
  __shared__ uint4 sh1[8 * 64];
  const u32 lane = threadIdx.x & 63u, l16 = lane & 15u;
  const bool g2side = threadIdx.x < 64u;
  u32 k[8];
  fp_from_mont<FrParams>(k, fp_neg<FrParams>(g2side ? *point : *value));
  FbPick pk;
#pragma unroll
  for (u32 s = 0; s < FBW_SLOTS; s++) pk.d[s] = 0;
  pk.neg = 0;
  fb_pick_digits(pk, g, k, 0, l16);
  if (g2side) {
    X29G2 acc = x29g2_inf();
#pragma unroll 1
    for (u32 s = 0; s * 16u < g.windows; s++) {
      const u32 w = s * 16u + l16;
      u32 d = 0;
#pragma unroll
      for (u32 q = 0; q < FBW_SLOTS; q++) if (q == s) d = pk.d[q];
      if (w >= g.windows) d = 0;
      if (d) x29g2_add_mixed(acc, aff_cneg(tab_g2[(size_t)w * g.entries + d], ((pk.neg >> s) & 1u) != 0));
    }
#pragma unroll 1
    for (u32 step = 1; step < 16u; step <<= 1) {
      const Xyzz<Fq2> mine = x29g2_store(acc);
      const uint4* mw = reinterpret_cast<const uint4*>(&mine);
      wave_lds_fence();
#pragma unroll
      for (int c = 0; c < 16; c++) sh2[c * 64 + lane] = mw[c];
      wave_lds_fence();
      Xyzz<Fq2> other;
      uint4* ow = reinterpret_cast<uint4*>(&other);
#pragma unroll
      for (int c = 0; c < 16; c++) ow[c] = sh2[c * 64 + (lane ^ step)];
      acc = x29g2_add(acc, x29g2_load(other));
    }
    x29g2_add_mixed(acc, *tau_g2);                 // (the identity in either slot is handled inside, as for `com` below)
    const Aff<Fq2> q = xyzz_to_aff(x29g2_store(acc));
    if (lane == 0) *out_q = q;
  } else {
    X29 acc = x29_inf();
#pragma unroll 1
    for (u32 s = 0; s * 16u < g.windows; s++) {
      const u32 w = s * 16u + l16;
      u32 d = 0;
#pragma unroll
      for (u32 q = 0; q < FBW_SLOTS; q++) if (q == s) d = pk.d[q];
      if (w >= g.windows) d = 0;
      if (d) {
        const Aff<Fq> e = aff_cneg(tab_g1[(size_t)w * g.entries + d], ((pk.neg >> s) & 1u) != 0);
        acc = x29_add(acc, x29_load(xyzz_from_aff(e)));
      }
    }
#pragma unroll 1
    for (u32 step = 1; step < 16u; step <<= 1) {
      const Xyzz<Fq> mine = x29_store(acc);
      const uint4* mw = reinterpret_cast<const uint4*>(&mine);
      wave_lds_fence();
#pragma unroll
      for (int c = 0; c < 8; c++) sh1[c * 64 + lane] = mw[c];
      wave_lds_fence();
      Xyzz<Fq> other;
      uint4* ow = reinterpret_cast<uint4*>(&other);
#pragma unroll
      for (int c = 0; c < 8; c++) ow[c] = sh1[c * 64 + (lane ^ step)];
      acc = x29_add(acc, x29_load(other));
    }
    if (!aff_is_inf(*com)) acc = x29_add(acc, x29_load(xyzz_from_aff(*com)));
    const Aff<Fq> a = xyzz_to_aff(x29_store(acc));
    if (lane == 0) *out_a = a;
  }
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
