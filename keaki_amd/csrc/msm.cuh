// Pippenger variable-base MSM for BN254 G1 / G2 on gfx950.
//
// Replaces `<E::G1 as VariableBaseMSM>::msm_unchecked(&setup.g1_aff, p)` (reference src/kzg.rs:98;
// ark-ec 0.4.2 msm_bigint_wnaf). The *result* is the same group element; the schedule is GPU-native:
//
//   K1 digits     one lane per scalar: Montgomery -> canonical, signed radix-2^c digits,
//                 per-(window,bucket) histogram                              [coalesced 32 B/lane]
//   K2 scan       exclusive prefix over the histogram (bucket start offsets)
//   K3 scatter    counting-sort of (point index | sign) into bucket order
//   K4 accumulate one lane per bucket: gather affine points (64 B rows), XYZZ mixed adds -- the
//                 dominant kernel: n * windows adds of 8M+2S
//   K5 reduce     per-window weighted bucket sum  sum_b (b+1) S_b  by chunked running sums
//   K6 finish     window sums -> Horner by 2^c -> one point, normalised
//
// Order of additions inside a bucket depends on atomics; EC addition is exact and commutative, so
// the affine result is bit-identical run to run.
#pragma once
#include "bn254_curve.cuh"

namespace bn254 {

constexpr u32 DIGIT_NONE = 0xFFFFFFFFu;

struct MsmShape {
  u32 n;        // number of (scalar, point) pairs
  u32 c;        // window bits
  u32 W;        // number of windows = ceil(254 / c) (top window never carries out, see digit rule)
  u32 B;        // buckets per window = 2^(c-1)
};

// signed-digit rule: coef in [0, 2^c]; if coef > 2^(c-1): digit = coef - 2^c, carry 1.
// => digits in [-(2^(c-1) - 1), 2^(c-1)]; bucket id = |digit| - 1 in [0, B).
// The top window holds < c - 1 significant bits whenever c does not divide 254 (guaranteed by the
// host-side choice of c), so it never produces a carry.
static __global__ void __launch_bounds__(256) k_msm_digits(const Fr* __restrict__ scalars, MsmShape s, u32* __restrict__ digits,
                                                    u32* __restrict__ hist) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n) return;
  Fr k = scalars[i];
  u32 v[8];
  fp_from_mont<FrParams>(v, k);
  u32 carry = 0;
  const u32 mask = (1u << s.c) - 1u, half = 1u << (s.c - 1);
  for (u32 w = 0; w < s.W; w++) {
    u32 coef = (v[0] & mask) + carry;
    // shift the 256-bit scalar right by c bits (static register indices; c <= 31)
#pragma unroll
    for (int j = 0; j < 7; j++) v[j] = (u32)((((u64)v[j + 1] << 32) | v[j]) >> s.c);
    v[7] >>= s.c;
    u32 out;
    if (coef > half) {
      out = ((1u << s.c) - coef - 1u) | 0x80000000u;  // negative digit, bucket = |d| - 1
      carry = 1;
    } else {
      carry = 0;
      out = coef ? coef - 1u : DIGIT_NONE;
    }
    digits[(size_t)w * s.n + i] = out;
    if (out != DIGIT_NONE) atomicAdd(&hist[w * s.B + (out & 0x7FFFFFFFu)], 1u);
  }
}

// ---- exclusive scan over `len` counters, restarted at every multiple of `seg` (one window) ----
// pass 1: per-block sums (block handles SCAN_ELEMS contiguous elements)
constexpr u32 SCAN_THREADS = 256, SCAN_PER_THREAD = 8, SCAN_ELEMS = SCAN_THREADS * SCAN_PER_THREAD;
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* total) {
  __shared__ u32 wsum[SCAN_THREADS / 64];
  u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    u32 y = __shfl_up(x, o, 64);
    if (lane >= (u32)o) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (u32 k = 0; k < SCAN_THREADS / 64; k++) {
    u32 s = wsum[k];
    if (k < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}
static __global__ void __launch_bounds__(SCAN_THREADS) k_scan_block_sums(const u32* __restrict__ in, u32 len, u32* __restrict__ block_sums) {
  u32 base = blockIdx.x * SCAN_ELEMS + threadIdx.x * SCAN_PER_THREAD;
  u32 s = 0;
#pragma unroll
  for (u32 k = 0; k < SCAN_PER_THREAD; k++) s += (base + k < len) ? in[base + k] : 0u;
  u32 tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// pass 2: one block scans the block sums; restart at segment boundaries (seg is a multiple of SCAN_ELEMS
// or the whole array is one block)
static __global__ void __launch_bounds__(SCAN_THREADS) k_scan_top(u32* __restrict__ block_sums, u32 nblocks, u32 blocks_per_seg) {
  // serial over chunks of SCAN_THREADS blocks; nblocks is small (<= len / 2048)
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (u32 start = 0; start < nblocks; start += SCAN_THREADS) {
    u32 idx = start + threadIdx.x;
    u32 v = idx < nblocks ? block_sums[idx] : 0u;
    u32 tot;
    u32 ex = block_exclusive_scan(v, &tot);
    u32 c0 = carry;
    if (idx < nblocks) block_sums[idx] = ex + c0;
    __syncthreads();
    if (threadIdx.x == 0) carry = c0 + tot;
    __syncthreads();
  }
  (void)blocks_per_seg;
}
// pass 3: final offsets. offsets are GLOBAL positions into the sorted array (no per-window restart:
// the sorted array is one dense stream, windows follow each other).
static __global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(const u32* __restrict__ in, u32 len, const u32* __restrict__ block_sums,
                                                             u32* __restrict__ out) {
  u32 base = blockIdx.x * SCAN_ELEMS + threadIdx.x * SCAN_PER_THREAD;
  u32 v[SCAN_PER_THREAD];
  u32 s = 0;
#pragma unroll
  for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
    v[k] = (base + k < len) ? in[base + k] : 0u;
    s += v[k];
  }
  u32 tot;
  u32 ex = block_exclusive_scan(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
    if (base + k < len) out[base + k] = ex;
    ex += v[k];
  }
}

// ---- K3: scatter point indices into bucket order ------------------------------------------------
static __global__ void __launch_bounds__(256) k_msm_scatter(const u32* __restrict__ digits, MsmShape s, const u32* __restrict__ offsets,
                                                     u32* __restrict__ cursor, u32* __restrict__ sorted) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 w = blockIdx.y;
  if (i >= s.n) return;
  u32 d = digits[(size_t)w * s.n + i];
  if (d == DIGIT_NONE) return;
  u32 key = w * s.B + (d & 0x7FFFFFFFu);
  u32 pos = offsets[key] + atomicAdd(&cursor[key], 1u);
  sorted[pos] = i | (d & 0x80000000u);
}

// ---- K4: bucket accumulation (dominant kernel) ----------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_msm_accumulate(const Aff<F>* __restrict__ points, const u32* __restrict__ sorted,
                                                        const u32* __restrict__ offsets, const u32* __restrict__ counts,
                                                        u32 nbuckets_total, Xyzz<F>* __restrict__ buckets) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbuckets_total) return;
  u32 start = offsets[t], cnt = counts[t];
  Xyzz<F> acc = xyzz_inf<F>();
  for (u32 k = 0; k < cnt; k++) {
    u32 e = sorted[start + k];
    Aff<F> p = points[e & 0x7FFFFFFFu];
    acc = xyzz_add_mixed(acc, aff_cneg(p, (e >> 31) != 0));
  }
  buckets[t] = acc;
}

// ---- K5: per-window weighted sum, chunked ----------------------------------------------------------
// thread (w, t) covers buckets [t*L, (t+1)*L) of window w: partial = sum (j+1) S_j over the chunk.
template <class F>
__global__ void __launch_bounds__(64) k_msm_reduce(const Xyzz<F>* __restrict__ buckets, MsmShape s, u32 L, u32 chunks_per_window,
                                                   Xyzz<F>* __restrict__ partials) {
  u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= s.W * chunks_per_window) return;
  u32 w = g / chunks_per_window, t = g % chunks_per_window;
  u32 lo = t * L;
  const Xyzz<F>* bk = buckets + (size_t)w * s.B;
  Xyzz<F> run = xyzz_inf<F>(), ws = xyzz_inf<F>();
  for (u32 j = L; j-- > 0;) {
    if (lo + j < s.B) {
      run = xyzz_add(run, bk[lo + j]);
      ws = xyzz_add(ws, run);
    }
  }
  // ws += lo * run   (lo < 2^(c-1)), MSB-first double-and-add
  if (lo != 0) {
    Xyzz<F> m = xyzz_inf<F>();
    for (int b = 31 - __clz(lo); b >= 0; b--) {
      m = xyzz_dbl(m);
      if ((lo >> b) & 1) m = xyzz_add(m, run);
    }
    ws = xyzz_add(ws, m);
  }
  partials[g] = ws;
}

// ---- K6a: sum the chunk partials of each window, then scale by 2^(w c) ---------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_msm_window_finish(const Xyzz<F>* __restrict__ partials, MsmShape s, u32 chunks_per_window,
                                                          Xyzz<F>* __restrict__ window_sums) {
  __shared__ Xyzz<F> sh[64];
  u32 w = blockIdx.x, l = threadIdx.x;
  Xyzz<F> acc = xyzz_inf<F>();
  for (u32 t = l; t < chunks_per_window; t += 64) acc = xyzz_add(acc, partials[(size_t)w * chunks_per_window + t]);
  sh[l] = acc;
  __syncthreads();
  for (u32 o = 32; o > 0; o >>= 1) {
    if (l < o) sh[l] = xyzz_add(sh[l], sh[l + o]);
    __syncthreads();
  }
  if (l == 0) {
    Xyzz<F> r = sh[0];
    for (u32 k = 0; k < w * s.c; k++) r = xyzz_dbl(r);
    window_sums[w] = r;
  }
}

// write a point as normalised Jacobian (x, y, 1) / (1, 1, 0)
template <class F>
KDEV void store_norm_jac(F* out, const Xyzz<F>& p) {
  Aff<F> a = xyzz_to_aff(p);
  if (xyzz_is_inf(p)) {
    out[0] = f_one<F>(); out[1] = f_one<F>(); out[2] = f_zero<F>();
  } else {
    out[0] = a.x; out[1] = a.y; out[2] = f_one<F>();
  }
}
// ---- K6b: add the window sums, normalise ------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_msm_final(const Xyzz<F>* __restrict__ window_sums, u32 W, F* __restrict__ out_jac) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Xyzz<F> acc = xyzz_inf<F>();
  for (u32 w = 0; w < W; w++) acc = xyzz_add(acc, window_sums[w]);
  store_norm_jac(out_jac, acc);
}

// ---- sum of k normalised-Jacobian points (multi-GPU partial combine) -------------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_sum_jac(const F* __restrict__ pts, u32 k, F* __restrict__ out_jac) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Xyzz<F> acc = xyzz_inf<F>();
  for (u32 i = 0; i < k; i++) {
    Jac<F> p = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    if (jac_is_inf(p)) continue;
    // inputs are normalised (z = 1); tolerate general z by converting through affine
    Aff<F> a = f_eq(p.z, f_one<F>()) ? Aff<F>{p.x, p.y} : jac_to_aff(p);
    acc = xyzz_add_mixed(acc, a);
  }
  store_norm_jac(out_jac, acc);
}

}  // namespace bn254
