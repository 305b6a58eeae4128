// Pippenger variable-base MSM for BN254 G1 / G2 on gfx950.
//
// Replaces `<E::G1 as VariableBaseMSM>::msm_unchecked(&setup.g1_aff, p)` (reference src/kzg.rs:98;
// ark-ec 0.4.2 msm_bigint_wnaf). The *result* is the same group element; the schedule is GPU-native:
//
//   K1 tile_sort / K2 cell_prefix, bin_scan / K3 chunk_sort
//                 one lane per scalar: Montgomery -> canonical, signed radix-2^c digits; the n*W
//                 (bucket, point) pairs are bucket-sorted by two LDS counting sorts whose outputs are
//                 contiguous images (no global atomics, no histogram pre-pass) [coalesced 32 B/lane]
//   K3c cnt_*    counting sort of the buckets by size (largest first) so waves have equal trip counts
//   K4 accumulate one lane per bucket: gather affine points (64 B rows), XYZZ mixed adds -- the
//                 dominant kernel: n * windows adds of 8M+2S
//   K5 reduce     per-window weighted bucket sum  sum_b (b+1) S_b  by chunked running sums
//   K6 finish     window sums -> Horner by 2^c -> one point, normalised
//
// Order of additions inside a bucket depends on LDS-atomic arrival order; EC addition is exact and
// commutative, so the affine result is bit-identical run to run.
#pragma once
#include <type_traits>
#include "bn254_curve.hip.h"
#include "fq29.hip.h"
#include "xyzz29.hip.h"
#include "jac29.hip.h"
#include "xyzz29_g2.hip.h"

namespace bn254 {

typedef uint16_t u16;
typedef u32 v4u_t __attribute__((ext_vector_type(4)));
constexpr u32 DIGIT_NONE = 0xFFFFFFFFu;
#ifndef STAMP            // bench_tools/dbg/sort_harness.hip defines it: in-kernel phase stamps of one workgroup
#define STAMP(cond, i) do { } while (0)
#endif

// Window plan. The 254 scalar bits are spread over W windows as evenly as possible: the first `k`
// windows are `c` bits wide, the remaining W - k are c - 1 bits wide (k >= 1, k*c + (W-k)*(c-1) = 254).
// Equal widths keep every bucket equally loaded; a plain radix-2^c split would leave a narrow top
// window whose few buckets receive up to 64x the points (one straggling workgroup / wave).
// Windows 0..W-2 use signed digits (2^(width-1) buckets); the top window keeps its digit unsigned
// (2^width buckets, ~76% populated since scalars are < r), so no carry ever leaves it.
struct MsmShape {
  u32 n;   // number of (scalar, point) pairs
  u32 c;   // width of the wide windows
  u32 W;   // number of windows
  u32 k;   // number of wide windows
  u32 stride;  // 0: every window has its own bucket range and reads point i.
               // N > 0 (precomputed SRS): all windows SHARE one bucket range and window w reads point
               // table[w * N + i] = 2^(bit offset of w) * P_i, so no per-window reduction / Horner step is left.
};
KDEV u32 msm_width(const MsmShape& s, u32 w) { return w < s.k ? s.c : s.c - 1; }
KDEV u32 msm_bit_offset(const MsmShape& s, u32 w) { return w < s.k ? w * s.c : s.k * s.c + (w - s.k) * (s.c - 1); }
KDEV u32 msm_nbuckets(const MsmShape& s, u32 w) { u32 wd = msm_width(s, w); return w == s.W - 1 ? (1u << wd) : (1u << (wd - 1)); }
KDEV u32 msm_bucket_base(const MsmShape& s, u32 w) {
  return w <= s.k ? w * (1u << (s.c - 1)) : s.k * (1u << (s.c - 1)) + (w - s.k) * (1u << (s.c - 2));
}
// host-side mirror of the same plan
struct MsmPlan {
  MsmShape s;
  size_t nb;       // total buckets
  u32 max_b;       // largest per-window bucket count
};
inline MsmPlan msm_make_plan(size_t n, int c_target) {
  MsmPlan p;
  u32 W = (254 + c_target - 1) / c_target;
  u32 base = 254 / W, rem = 254 % W;
  p.s.n = (u32)n;
  p.s.W = W;
  p.s.stride = 0;
  if (rem == 0) { p.s.c = base; p.s.k = W; } else { p.s.c = base + 1; p.s.k = rem; }
  size_t nb = 0; u32 mb = 0;
  for (u32 w = 0; w < W; w++) {
    u32 wd = w < p.s.k ? p.s.c : p.s.c - 1;
    u32 b = (w == W - 1) ? (1u << wd) : (1u << (wd - 1));
    nb += b; if (b > mb) mb = b;
  }
  p.nb = nb; p.max_b = mb;
  return p;
}

// Digits of one scalar, lowest window first. `emit(w, code)` is called for every non-zero digit with
// code = bucket (= |d| - 1) | sign << 31. Signed rule for windows below the top: coef in [0, 2^wd];
// coef > 2^(wd-1) -> digit coef - 2^wd with carry 1 (coef == 2^wd is digit 0 with carry 1).
template <class Emit>
KDEV void msm_for_each_digit_canon(const u32 (&canon)[8], const MsmShape& s, Emit emit) {
  u32 v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = canon[j];
  u32 carry = 0;
  for (u32 w = 0; w < s.W; w++) {
    const u32 wd = msm_width(s, w);
    const u32 full = 1u << wd, half = full >> 1;
    u32 coef = (v[0] & (full - 1u)) + carry;
    // shift the 256-bit scalar right by wd bits (static register indices; wd <= 31)
#pragma unroll
    for (int j = 0; j < 7; j++) v[j] = (u32)((((u64)v[j + 1] << 32) | v[j]) >> wd);
    v[7] >>= wd;
    if (w == s.W - 1) {
      if (coef) emit(w, coef - 1u);      // top window: unsigned, 2^wd buckets (coef <= 0.76 * 2^wd + 1)
    } else if (coef > half) {
      carry = 1;
      if (coef != full) emit(w, (full - coef - 1u) | 0x80000000u);
    } else {
      carry = 0;
      if (coef) emit(w, coef - 1u);
    }
  }
}

// The same digits for a plan that is known at compile time. msm_make_plan derives c and k from the number of windows alone, so W fixes
// every bit offset: a digit is one funnel shift of two NAMED limbs and a mask (the run-time walk above shifts the whole 256-bit value
// per window, or -- cut limb by limb -- spends ~45 scalar instructions per window on the bookkeeping; the CU's one scalar unit was the
// limit of the first tile sort). code[w] = bucket | sign << 31, DIGIT_NONE for a zero digit.
template <u32 W> struct StaticPlan {
  static constexpr u32 base = 254 / W, rem = 254 % W;
  static constexpr u32 c = rem ? base + 1 : base, k = rem ? rem : W;
  static constexpr u32 width(u32 w) { return w < k ? c : c - 1; }
  static constexpr u32 offset(u32 w) { return w < k ? w * c : k * c + (w - k) * (c - 1); }
};
template <u32 W>
KDEV void msm_digits_static(const u32 (&v)[8], u32 (&code)[W]) {
  typedef StaticPlan<W> P;
  u32 carry = 0;
#pragma unroll
  for (u32 w = 0; w < W; w++) {
    const u32 off = P::offset(w), wd = P::width(w), j = off >> 5, sh = off & 31;
    const u32 lo = v[j], hi = j + 1 < 8 ? v[j + 1] : 0u;
    const u32 d = (sh + wd <= 32 ? lo >> sh : __builtin_amdgcn_alignbit(hi, lo, sh)) & ((1u << wd) - 1u);
    const u32 full = 1u << wd, half = full >> 1;
    const u32 coef = d + carry;
    if (w == W - 1) {
      code[w] = coef ? coef - 1u : DIGIT_NONE;
    } else {
      const bool neg = coef > half;
      carry = neg ? 1u : 0u;
      code[w] = neg ? (coef != full ? ((full - coef - 1u) | 0x80000000u) : DIGIT_NONE) : (coef ? coef - 1u : DIGIT_NONE);
    }
  }
}

template <class Emit>
KDEV void msm_for_each_digit(const Fr& k, const MsmShape& s, Emit emit) {
  u32 v[8];
  fp_from_mont<FrParams>(v, k);
  msm_for_each_digit_canon(v, s, emit);
}

// ---- bucket sort of the (bucket, point) pairs: two LDS counting sorts, every global store part of a contiguous image -------------------
// Global bucket id g = w * B + bucket (shared-bucket mode: g = bucket). Its MIDDLE bits are the bin, the bits around them the fine index:
//     g = [ hb high bits | lb bin bits | shift low bits ],  fine = high << shift | low   (nf = 2^(shift + hb) buckets per bin)
// so every bin holds a slice of each 2^(shift + lb)-aligned stretch of the bucket range: with window tables the low half of the buckets
// takes the digits of all twelve windows and the high half those of two, and bins of consecutive buckets would differ elevenfold.
//   pass 1, k_tile_sort   one workgroup per TILE of <= 3072 scalars: Montgomery -> canonical once, the digits cut twice (count, place);
//                         the tile's <= 36 K pairs are counting-sorted by bin in 144 KB of LDS and leave as ONE contiguous image of 4-byte
//                         entries (fine | sign | window | index inside the tile) plus the bins' start positions. No global histogram,
//                         no scan, no second read of the scalars.
//   k_cell_prefix / k_bin_scan   per bin: running position of its cell in every tile; per-bin totals -> image base, chunk ids
//   pass 2, k_chunk_sort  one workgroup per BIN: the bin's cells (tile by tile) are gathered 32 K pairs at a time, counting-sorted by fine
//                         bucket in 128 KB of LDS and written as ONE contiguous image per chunk; the tile a cell came from completes the
//                         point index, which is why 4-byte entries suffice. Per bucket and chunk one word first | end << 16.
// A bucket's pairs therefore lie in one SEGMENT per chunk of its bin (three at 2^24 points); the bucket kernels walk them (SegWalker).
// No global atomics. Barriers inside the two passes wait for LDS only (lds_barrier): __syncthreads also waits for the workgroup's global
// stores, i.e. for the image that is still draining to HBM -- with one workgroup per CU nothing else hides that.
constexpr u32 T1_THREADS = 1024;
constexpr u32 T1_PER = 3;                 // scalars per lane of a pass-1 tile
constexpr u32 T1_CAP = 36864;             // pairs staged per pass-1 tile (4 B each = 144 KB of LDS)
constexpr u32 PART_MAX_BINS = 2048;       // coarse bins
constexpr u32 C2_THREADS = 1024;
constexpr u32 C2_CAP = 32768;             // pairs per pass-2 chunk (4 B each = 128 KB of LDS). Long chunks: the bucket kernel pays for every segment
                                          // (15 K chunks, 13 segments per bucket: +0.5 ms at 2^24 points)
constexpr u32 PART_MAX_FINE_SHIFT = 11;
constexpr u32 PART_MAX_FINE = 1u << PART_MAX_FINE_SHIFT;
constexpr u32 SEG_INLINE = 4;             // segment words kept per bucket in the bucket-major table (bins of more chunks: chunk-major rows behind them)
// pass-1 entry: index inside the tile (12 bits: tile <= 3072) | window (7 bits: W <= 85) | sign | fine bucket (<= 11 bits) | 1: a word that is
// zero -- what an out-of-range buffer load returns -- is no entry
constexpr u32 TE_WPOS = 12, TE_SIGN = 19, TE_FINE = 20, TE_VALID = 0x80000000u;

struct PartShape {
  u32 nbins, lb;   // coarse bins = 2^lb
  u32 shift, hb;   // low / high fine bits
  u32 nf;          // buckets per bin = 2^(shift + hb)
  u32 tile;        // scalars per pass-1 tile (<= T1_THREADS * T1_PER, tile * W <= T1_CAP)
  u32 ntiles;      // ceil(n / tile)
  u32 te;          // entry slots per tile image = tile * W
  u32 geom;        // pass-2 gather shape for the expected cell length te / nbins (k_chunk_sort)
};
KDEV u32 part_bin(const PartShape& ps, u32 g) { return (g >> ps.shift) & (ps.nbins - 1u); }
KDEV u32 part_fine(const PartShape& ps, u32 g) { return ((g >> (ps.shift + ps.lb)) << ps.shift) | (g & ((1u << ps.shift) - 1u)); }
KDEV u32 part_bucket(const PartShape& ps, u32 bin, u32 f) { return ((f >> ps.shift) << (ps.shift + ps.lb)) | (bin << ps.shift) | (f & ((1u << ps.shift) - 1u)); }
inline bool part_make_shape(size_t n, u32 W, size_t nb, PartShape* ps, int lb_override = -1, int hb_override = -1, int geom_override = -1) {
  u32 lg = 0;
  while (((size_t)1 << lg) < nb) lg++;
  const size_t pairs = n * (size_t)W;
  // about one chunk per bin where the input allows (one segment per bucket), at most 1024 bins below 2^28 pairs: measured at 2^24 points, 1024
  // bins (six segments per bucket, 36-entry cells) against 2048 (three, 18): pass 2 0.84 vs 0.95 ms, cell table 0.055 vs 0.10 ms, bucket
  // kernel +0.06 ms
  const u32 lb_max = pairs > ((size_t)C2_CAP << 13) ? 11u : 10u;
  u32 lb = 0;
  while (lb < lb_max && ((size_t)C2_CAP << lb) < pairs) lb++;
  if (lb_override >= 0 && lb_override <= 11) lb = (u32)lb_override;
  if (lb > lg) lb = lg;
  if (lg - lb > PART_MAX_FINE_SHIFT) lb = lg - PART_MAX_FINE_SHIFT;
  if (lb > 11) return false;
  const u32 fb = lg - lb;
  ps->lb = lb; ps->nbins = 1u << lb;
  ps->hb = fb < 3 ? fb : 3u;
  if (hb_override >= 0 && (u32)hb_override <= fb) ps->hb = (u32)hb_override;
  ps->shift = fb - ps->hb;
  ps->nf = 1u << fb;
  u32 tile = W ? T1_CAP / W : T1_THREADS * T1_PER;
  if (tile > T1_THREADS * T1_PER) tile = T1_THREADS * T1_PER;
  if (tile == 0 || W >= (1u << (TE_SIGN - TE_WPOS))) return false;
  ps->tile = tile;
  ps->ntiles = (u32)((n + tile - 1) / tile);
  ps->te = tile * W;
  const u32 mean = ps->te / ps->nbins;
  ps->geom = mean <= 24 ? 0u : mean <= 48 ? 1u : mean <= 100 ? 2u : 3u;
  if (geom_override >= 0) ps->geom = (u32)geom_override;
  return true;
}
// chunks the second pass can produce for `pairs` pairs in `nbins` bins (every bin's last chunk may be partial)
inline size_t part_max_chunks(size_t pairs, u32 nbins) { return pairs / C2_CAP + nbins + 1; }

// barrier that orders LDS traffic only
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive scan, in place, of a[0..len) in LDS, len <= 4 * blockDim.x (four consecutive elements per lane); returns the total.
// Whole workgroup calls it; LDS-only barriers.
__device__ __forceinline__ u32 lds_exclusive_scan4(u32* a, u32 len, u32* wsum) {
  const u32 t = threadIdx.x, lane = t & 63, wid = t >> 6, nw = blockDim.x >> 6;
  lds_barrier();
  u32 x[4], sum = 0;
#pragma unroll
  for (u32 i = 0; i < 4; i++) { x[i] = 4 * t + i < len ? a[4 * t + i] : 0u; sum += x[i]; }
  u32 v = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { u32 y = __shfl_up(v, o, 64); if (lane >= (u32)o) v += y; }
  if (lane == 63) wsum[wid] = v;
  lds_barrier();
  u32 base = 0, tot = 0;
  for (u32 k = 0; k < nw; k++) { u32 sv = wsum[k]; if (k < wid) base += sv; tot += sv; }
  u32 ex = base + v - sum;
#pragma unroll
  for (u32 i = 0; i < 4; i++) { if (4 * t + i < len) a[4 * t + i] = ex; ex += x[i]; }
  lds_barrier();
  return tot;
}

// pass 1: tiles[tile * te + q], q < m = the tile's pairs ordered by bin; tstart[tile * (nbins + 1) + b] = first position of bin b, [nbins] = m.
// A workgroup walks tiles blockIdx.x, + gridDim.x, ...: the stores of a finished image drain under the next tile's loads and arithmetic.
// WS > 0: the plan with WS windows, digits cut by msm_digits_static; WS = 0: any plan, the run-time digit walk. Either way the canonical
// scalars stay in registers and the digits are cut twice, for the counting and for the placing pass.
template <u32 WS>
static __global__ void __launch_bounds__(T1_THREADS) k_tile_sort(const Fr* __restrict__ scalars, MsmShape s, PartShape ps, u32* __restrict__ tiles,
                                                                 u16* __restrict__ tstart) {
  __shared__ u32 ent[T1_CAP + 64];         // [T1_CAP + lane]: where the (rare) zero digits of the branch-free loops go
  __shared__ u32 cur[PART_MAX_BINS + 64];  // [PART_MAX_BINS + lane]: their bin (one word per lane: same-address LDS atomics of a wave are serial)
  __shared__ u32 wsum[T1_THREADS / 64];
  constexpr u32 NC = WS ? WS : 1u;
  // the scalars of a workgroup's NEXT tile are requested while the current one is counted and placed (round 5: the kernel needs 83
  // registers since the second cut stopped living in scratch, the 24 of a tile in flight fit)
  Fr nxt[T1_PER];
  auto request = [&](u32 tile, u32 tl) {
#pragma unroll
    for (u32 p = 0; p < T1_PER; p++) {                     // branch-free: a lane without a scalar reads the last one and drops its digits
      const u32 li = p * T1_THREADS + tl;
      const size_t i = (size_t)tile * ps.tile + li;
      nxt[p] = scalars[li < ps.tile && i < s.n ? i : (size_t)s.n - 1];
    }
  };
  if (blockIdx.x < ps.ntiles) request(blockIdx.x, threadIdx.x);
  for (u32 tile = blockIdx.x; tile < ps.ntiles; tile += gridDim.x) {
    u32 t = threadIdx.x;
    asm volatile("" : "+v"(t));                              // per-lane invariants are re-formed per tile, not kept (and spilled) across it
    [[maybe_unused]] const u32 sk = (tile - blockIdx.x) / gridDim.x;     // (the STAMP harness reads it)
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 0);
    u32 canon[T1_PER][8];          // the canonical scalars; the digits are cut twice (count, place): 24 registers instead of 3 W
    bool act[T1_PER];
#pragma unroll
    for (u32 p = 0; p < T1_PER; p++) {
      const u32 li = p * T1_THREADS + t;
      act[p] = li < ps.tile && (size_t)tile * ps.tile + li < s.n;
      fp_from_mont<FrParams>(canon[p], nxt[p]);
    }
    if (tile + gridDim.x < ps.ntiles) request(tile + gridDim.x, t);
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 1);
    for (u32 b = t; b < ps.nbins; b += T1_THREADS) cur[b] = 0;
    if (t < 64) cur[PART_MAX_BINS + t] = 0;
    lds_barrier();
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 2);
#pragma unroll
    for (u32 p = 0; p < T1_PER; p++) {
      if constexpr (WS != 0) {
        u32 code[NC];
        msm_digits_static<NC>(canon[p], code);
#pragma unroll
        for (u32 w = 0; w < NC; w++) {            // no branch per digit: a zero digit (one in 2^c) counts into the dummy bin
          const u32 g = (s.stride ? 0u : msm_bucket_base(s, w)) + (code[w] & 0x7FFFFFFFu);
          atomicAdd(&cur[act[p] && code[w] != DIGIT_NONE ? part_bin(ps, g) : PART_MAX_BINS + (t & 63u)], 1u);
        }
      } else if (act[p]) {
        msm_for_each_digit_canon(canon[p], s, [&](u32 w, u32 cd) {
          const u32 g = (s.stride ? 0u : msm_bucket_base(s, w)) + (cd & 0x7FFFFFFFu);
          atomicAdd(&cur[part_bin(ps, g)], 1u);
        });
      }
    }
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 3);
    const u32 m = lds_exclusive_scan4(cur, ps.nbins, wsum);      // cur[b] = start of bin b inside the tile
    u16* ts = tstart + (size_t)tile * (ps.nbins + 1);
    for (u32 b = t; b < ps.nbins; b += T1_THREADS) ts[b] = (u16)cur[b];
    if (t == 0) ts[ps.nbins] = (u16)m;
    lds_barrier();
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 4);
    // "cut twice" has to be said to the compiler: it recognises the second cut as the first one's value and keeps all 3 W codes alive across
    // the scan -- in scratch (35 dwords per lane at W = 12). Behind this statement the canonical limbs are new values to it.
#pragma unroll
    for (u32 p = 0; p < T1_PER; p++)
#pragma unroll
      for (u32 j = 0; j < 8; j++) asm volatile("" : "+v"(canon[p][j]));
#pragma unroll
    for (u32 p = 0; p < T1_PER; p++) {
      const u32 li = p * T1_THREADS + t;
      if constexpr (WS != 0) {
        u32 code[NC];
        msm_digits_static<NC>(canon[p], code);
#pragma unroll
        for (u32 w = 0; w < NC; w++) {
          const u32 cd = code[w];
          const bool ok = act[p] && cd != DIGIT_NONE;
          const u32 g = (s.stride ? 0u : msm_bucket_base(s, w)) + (cd & 0x7FFFFFFFu);
          const u32 pos = atomicAdd(&cur[ok ? part_bin(ps, g) : PART_MAX_BINS + (t & 63u)], 1u);
          ent[ok ? pos : T1_CAP + (t & 63u)] = TE_VALID | (part_fine(ps, g) << TE_FINE) | ((cd >> 31) << TE_SIGN) | (w << TE_WPOS) | li;
        }
      } else if (act[p]) {
        msm_for_each_digit_canon(canon[p], s, [&](u32 w, u32 cd) {
          const u32 g = (s.stride ? 0u : msm_bucket_base(s, w)) + (cd & 0x7FFFFFFFu);
          const u32 pos = atomicAdd(&cur[part_bin(ps, g)], 1u);
          ent[pos] = TE_VALID | (part_fine(ps, g) << TE_FINE) | ((cd >> 31) << TE_SIGN) | (w << TE_WPOS) | li;
        });
      }
    }
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 5);
    lds_barrier();
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 6);
    u32* dst = tiles + (size_t)tile * ps.te;
#pragma unroll 4
    for (u32 q = t; q < m; q += T1_THREADS) dst[q] = ent[q];
    STAMP(blockIdx.x == 100 && t == 0 && sk < 4, 2048 + sk * 16 + 7);
    // no barrier here: the next tile touches `ent` again only behind two more barriers
  }
}

// per bin: where its cell of every tile begins in the bin's own order (tile after tile), and how many pairs the bin holds
static __global__ void __launch_bounds__(1024) k_cell_prefix(const u16* __restrict__ tstart, PartShape ps, uint2* __restrict__ cellmeta,
                                                             u32* __restrict__ bin_total) {
  __shared__ u32 a[4096];
  __shared__ u32 wsum[16];
  const u32 b = blockIdx.x, t = threadIdx.x;
  u32 carry = 0;
  for (u32 base = 0; base < ps.ntiles; base += 4096) {
    u32 st[4], ln[4];
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      const u32 tile = base + 4 * t + k;
      st[k] = 0; ln[k] = 0;
      if (tile < ps.ntiles) {
        const u16* row = tstart + (size_t)tile * (ps.nbins + 1) + b;
        st[k] = row[0];
        ln[k] = (u32)row[1] - st[k];
      }
      a[4 * t + k] = ln[k];
    }
    const u32 tot = lds_exclusive_scan4(a, 4096, wsum);
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      const u32 tile = base + 4 * t + k;
      if (tile < ps.ntiles) cellmeta[(size_t)b * ps.ntiles + tile] = make_uint2(carry + a[4 * t + k], st[k] | (ln[k] << 16));
    }
    carry += tot;
    lds_barrier();
  }
  if (t == 0) bin_total[b] = carry;
}
// where a bin's chunk images begin in `sorted`, and which chunk ids (rows of the chunk-major offset table) it owns
struct BinMeta {
  u32 img_base, chunk_first, nch, total;
};
static __global__ void __launch_bounds__(1024) k_bin_scan(const u32* __restrict__ bin_total, u32 nbins, BinMeta* __restrict__ bins) {
  __shared__ u32 a[2048];
  __shared__ u32 c[2048];
  __shared__ u32 wsum[16];
  const u32 t = threadIdx.x;
  u32 tt[2];
#pragma unroll
  for (u32 k = 0; k < 2; k++) {
    const u32 b = 2 * t + k;
    tt[k] = b < nbins ? bin_total[b] : 0u;
    a[b] = tt[k];
    c[b] = (tt[k] + C2_CAP - 1) / C2_CAP;
  }
  lds_exclusive_scan4(a, 2048, wsum);
  lds_exclusive_scan4(c, 2048, wsum);
#pragma unroll
  for (u32 k = 0; k < 2; k++) {
    const u32 b = 2 * t + k;
    if (b < nbins) bins[b] = {a[b], c[b], (tt[k] + C2_CAP - 1) / C2_CAP, tt[k]};
  }
}

// pass 2: one workgroup per bin. A lane's chain of dependent loads, not the bandwidth, set the time of the first version, and 4-byte gathers
// ran at ~40 cycles per wave instruction in the texture addresser (20 us per chunk and pass). So a chunk costs TWO dependent loads:
// the descriptors of its cells come in with coalesced loads -- requested while the previous chunk is still being scanned and placed -- and go
// through LDS to the groups of L lanes that own a cell each; then every lane requests R 16-byte pieces (4 entries each) of each of its
// group's Q cells at once, through a buffer descriptor (one offset register per load; a slot without entries asks for an offset out of
// range and gets zeros). The placing pass asks again (L2): kept in registers across the scan the entries were spilled.
// The gather shape <L, R, Q> follows the expected cell length te / nbins (PartShape::geom): a group's slots hold 4 L R entries of a cell,
// a chunk up to Q * 1024 / L cells. The tail of a longer cell is fetched by its own group in a (divergent, rare) loop; a chunk with more
// cells takes chunk_slow for both passes: the plain walk, cell after cell (skewed scalars only).
// Five LDS-only barriers per chunk; the empty slots of the branch-free loops go to one dummy word PER LANE (hist[PART_MAX_FINE + lane],
// pay[C2_CAP + lane]): LDS atomics of a wave to one address are executed one after the other -- with a single dummy counter the ~40 %
// empty slots of a wave instruction cost ~60 cycles instead of 7, and both passes ran at a tenth of their speed.
constexpr u32 C2_META = 2048;
KDEV u32 chunk_payload(u32 en, u32 ibase, u32 stride) {
  return (((en >> TE_SIGN) & 1u) << 31) | (ibase + (en & ((1u << TE_WPOS) - 1u)) + ((en >> TE_WPOS) & ((1u << (TE_SIGN - TE_WPOS)) - 1u)) * stride);
}
template <int PHASE>   // 0 counts, 1 places
KDEV void chunk_slow(const u32* __restrict__ tiles, const uint2* __restrict__ cm, const PartShape& ps, u32 stride, u32 c0, u32 lo, u32 hi, u32* hist,
                     u32* pay, u32* sh_next) {
  const u32 grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  for (u32 c = c0 + grp; c < ps.ntiles; c += C2_THREADS / 16) {
    const uint2 m = cm[c];
    const u32 len = m.y >> 16;
    if (PHASE == 0 && m.x + len > hi && l16 == 0) atomicMin(sh_next, c);     // the first cell that reaches past the chunk opens the next one
    if (m.x > hi || (m.x == hi && len)) break;                               // (an EMPTY cell that sits exactly at the chunk's end is stepped over)
    const u32 j0 = m.x < lo ? lo - m.x : 0u, j1 = hi - m.x < len ? hi - m.x : len;
    const u32* src = tiles + (size_t)c * ps.te + (m.y & 0xFFFFu);
    for (u32 j = j0 + l16; j < j1; j += 16) {
      const u32 en = src[j];
      const u32 f = (en >> TE_FINE) & (PART_MAX_FINE - 1u);
      if (PHASE == 0) {
        atomicAdd(&hist[f], 1u);
      } else {
        const u32 pos = atomicAdd(&hist[f], 1u);
        pay[pos] = chunk_payload(en, c * ps.tile, stride);
      }
    }
  }
}
// MASKED (round 5, the default): the ~45 % empty slots of the gather shape are skipped under the exec mask; before, every empty slot was an LDS
// atomic on a dummy word per lane (branch-free loops): the LDS pipeline, not the VALU, bounds this kernel, and it saw twice the operations it
// needed (0.84 -> 0.70 ms at 2^24 points, same box; option cs_masked = 0 for the A/B)
template <u32 L, u32 R, u32 Q, bool MASKED = true>
static __global__ void __launch_bounds__(C2_THREADS) k_chunk_sort(const u32* __restrict__ tiles, const uint2* __restrict__ cellmeta,
                                                                  const BinMeta* __restrict__ bins, MsmShape s, PartShape ps, u32 nbuckets_total,
                                                                  u32* __restrict__ sorted, v4u_t* __restrict__ segtab, u32* __restrict__ segoff,
                                                                  u32* __restrict__ bucket_counts) {
  constexpr u32 GROUPS = C2_THREADS / L, CELLS = GROUPS * Q, CAPC = 4 * L * R, NB = 4, QB = Q / NB;
  static_assert(CELLS <= C2_META && Q * R * 4 <= 64 && Q <= 32 && Q % NB == 0 && PART_MAX_FINE == 2 * C2_THREADS, "gather shape");
  __shared__ u32 pay[C2_CAP + 64];
  __shared__ uint2 meta[C2_META];
  __shared__ u32 hist[PART_MAX_FINE + 64];
  __shared__ u32 wsum[C2_THREADS / 64];
  __shared__ u32 sh_next, sh_odd;
  const u32 b = blockIdx.x, t = threadIdx.x, nf = ps.nf;
  const BinMeta bm = bins[b];
  const uint2* cm = cellmeta + (size_t)b * ps.ntiles;
  u32 tot[2] = {0, 0};                                      // the bin's per-bucket totals: buckets 2 t, 2 t + 1 of this lane
  u32 c0 = 0;                                               // first cell that reaches into the chunk
  // prologue: counters cleared, descriptors of the first chunk's cells staged
  for (u32 f = t; f < PART_MAX_FINE + 64; f += C2_THREADS) hist[f] = 0;
#pragma unroll
  for (u32 i = 0; i < C2_META / C2_THREADS; i++) {
    const u32 ci = t + i * C2_THREADS;
    if (ci < CELLS) meta[ci] = ci < ps.ntiles ? cm[ci] : make_uint2(0xFFFFFFFFu, 0u);
  }
  if (t == 0) { sh_next = ps.ntiles; sh_odd = 0; }
  lds_barrier();
  for (u32 k = 0; k < bm.nch; k++) {
    const u32 lo = k * C2_CAP, hi = bm.total - lo < C2_CAP ? bm.total : lo + C2_CAP;
    u32 tk = t;
    asm volatile("" : "+v"(tk));                            // per-cell invariants are re-formed per chunk, not kept (and spilled) across it
    const u32 grp = tk / L, ll = tk % L, dummy = PART_MAX_FINE + (tk & 63u);
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 0);
    const size_t span = (size_t)(ps.ntiles - c0) * ps.te * 4;
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc((void*)(tiles + (size_t)c0 * ps.te), 0,
                                                                       span < 0xFFFFFFF0ull ? (u32)span : 0xFFFFFFF0u, 0x00020000);
    // the entries of the group's cells; which cells are longer than the group's slots; which cell opens the next chunk; more cells than the shape holds?
    auto fetch = [&](u32 tt, u32 q0, u32 (&e)[QB][R][4], u32& longcells, u32& cand, bool& odd) {
      const u32 grp = tt / L, ll = tt % L;
#pragma unroll
      for (u32 qq = 0; qq < QB; qq++) {
        const u32 q = q0 + qq;
        const u32 ci = grp + q * GROUPS;
        const uint2 m = meta[ci];
        const u32 len = m.y >> 16;
        const bool beyond = m.x > hi || (m.x == hi && len);         // (an EMPTY cell that sits exactly at the chunk's end is stepped over)
        if (m.x != 0xFFFFFFFFu && m.x + len > hi && c0 + ci < cand) cand = c0 + ci;
        const u32 j0 = m.x < lo ? lo - m.x : 0u;
        const u32 j1 = beyond ? 0u : (hi - m.x < len ? hi - m.x : len);
        const u32 off = ci * ps.te + (m.y & 0xFFFFu);               // relative to the image of cell c0: < 2048 * 36864 entries
#pragma unroll
        for (u32 r = 0; r < R; r++) {                               // a cell's image is dword-aligned, which is all a buffer load asks for
          const u32 j = j0 + 4 * (ll + r * L);
          const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(img, j < j1 ? (off + j) * 4u : 0xFFFFFFFFu, 0, 0);
#pragma unroll
          for (u32 x = 0; x < 4; x++) e[qq][r][x] = j + x < j1 ? v[x] : 0u;     // the image goes on with other bins' entries
        }
        if (j1 > j0 + CAPC) longcells |= 1u << q;
        if (q == Q - 1) odd = !beyond && m.x != 0xFFFFFFFFu;
      }
    };
    // the tails of the group's long cells (rare; the group alone)
    auto tails = [&](u32 longcells, auto use) {
      while (longcells) {
        const u32 q = __builtin_ctz(longcells);
        longcells &= longcells - 1u;
        const u32 ci = grp + q * GROUPS;
        const uint2 m = meta[ci];
        const u32 len = m.y >> 16;
        const u32 j0 = m.x < lo ? lo - m.x : 0u, j1 = hi - m.x < len ? hi - m.x : len;
        const u32* src = tiles + (size_t)(c0 + ci) * ps.te + (m.y & 0xFFFFu);
        for (u32 j = j0 + CAPC + ll; j < j1; j += L) use(src[j], (c0 + ci) * ps.tile);
      }
    };
    u32 longcells = 0;
    {
      u32 cand = ps.ntiles;
      bool odd = false;
      // in batches of QB cells, the next batch's loads in flight while this one's are counted (64 entry registers at once were spilled)
      u32 ea[QB][R][4], eb[QB][R][4];
      u32 th = t;
      asm volatile("" : "+v"(th));
      fetch(th, 0, ea, longcells, cand, odd);
      STAMP(b == 100 && t == 0 && k < 8, k * 16 + 1);
#pragma unroll
      for (u32 h = 0; h < NB; h++) {
        u32 (&cur)[QB][R][4] = h % 2 ? eb : ea;
        if (h + 1 < NB) fetch(th, (h + 1) * QB, h % 2 ? ea : eb, longcells, cand, odd);
#pragma unroll
        for (u32 q = 0; q < QB; q++)
#pragma unroll
          for (u32 r = 0; r < R; r++)
#pragma unroll
            for (u32 x = 0; x < 4; x++) {
              const u32 en = cur[q][r][x];
              if constexpr (MASKED) { if (en) atomicAdd(&hist[(en >> TE_FINE) & (PART_MAX_FINE - 1u)], 1u); }
              else atomicAdd(&hist[en ? (en >> TE_FINE) & (PART_MAX_FINE - 1u) : dummy], 1u);    // an entry: its fine bucket; 0: the lane's dummy counter
            }
        __builtin_amdgcn_sched_barrier(0);
      }
      tails(longcells, [&](u32 en, u32) { atomicAdd(&hist[(en >> TE_FINE) & (PART_MAX_FINE - 1u)], 1u); });
      if (ll == 0 && cand < ps.ntiles) atomicMin(&sh_next, cand);
      if (odd) sh_odd = 1;
    }
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 2);
    lds_barrier();                                                                                  // ---- 1: counts complete
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 3);
    const bool slow = sh_odd != 0;
    if (slow) {                                             // count again, the plain way
      lds_barrier();
      u32 ts = threadIdx.x;
      asm volatile("" : "+v"(ts));
      for (u32 f = ts; f < nf; f += C2_THREADS) hist[f] = 0;
      lds_barrier();
      chunk_slow<0>(tiles, cm, ps, s.stride, c0, lo, hi, hist, pay, &sh_next);
      lds_barrier();
    }
    const u32 c_next = sh_next;
    u32 tm = threadIdx.x;
    asm volatile("" : "+v"(tm));                          // (addresses formed from the lane index are formed here, per chunk: hoisted out of the loop they were spilled)
    uint2 mnext[C2_META / C2_THREADS];
    if (k + 1 < bm.nch) {                                   // the next chunk's descriptors: in flight under the scan and the placing pass
#pragma unroll
      for (u32 i = 0; i < C2_META / C2_THREADS; i++) {
        const u32 ci = tm + i * C2_THREADS;
        mnext[i] = ci < CELLS && c_next + ci < ps.ntiles ? cm[c_next + ci] : make_uint2(0xFFFFFFFFu, 0u);
      }
    }
    // exclusive scan of the 2048 counters: two per lane, wave scan, wave totals through LDS
    const u32 x0 = hist[2 * tm], x1 = hist[2 * tm + 1], sum = x0 + x1;
    tot[0] += x0; tot[1] += x1;
    u32 v = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(v, o, 64); if ((tm & 63u) >= (u32)o) v += y; }
    if ((tm & 63u) == 63u) wsum[tm >> 6] = v;
    lds_barrier();                                                                                  // ---- 2: wave totals
    u32 base = 0;
#pragma unroll
    for (u32 w = 0; w < C2_THREADS / 64; w++) base += w < (tm >> 6) ? wsum[w] : 0u;
    const u32 ex = base + v - sum;
    hist[2 * tm] = ex; hist[2 * tm + 1] = ex + x0;            // first position of buckets 2 tm, 2 tm + 1 inside the chunk
    {
      const u32 w0 = ex | ((ex + x0) << 16), w1 = (ex + x0) | ((ex + sum) << 16);                   // first | end << 16, <= 32768 each
      // the first SEG_INLINE words of a bucket go straight to its row of the bucket-major table (round 5: kept in eight registers until the end of
      // the bin they were what the masked gather loops spilled), the later ones to the chunk-major rows
      if (k < SEG_INLINE) {
#pragma unroll
        for (u32 i = 0; i < 2; i++) {
          const u32 f = 2 * tm + i, g = part_bucket(ps, b, f);
          if (f < nf && g < nbuckets_total) reinterpret_cast<u32*>(segtab)[(size_t)g * SEG_INLINE + k] = i ? w1 : w0;
        }
      } else if (2 * tm < nf) { u32* so = segoff + (size_t)(bm.chunk_first + k) * nf + 2 * tm; so[0] = w0; so[1] = w1; }
    }
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 4);
    lds_barrier();                                                                                  // ---- 3: positions complete
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 5);
    if (!slow) {
      u32 lc2 = 0, cand = 0;                                // the entries are asked for again (L2): kept across the scan they were spilled
      bool odd = false;
      u32 ea[QB][R][4], eb[QB][R][4];
      u32 th = t;
      asm volatile("" : "+v"(th));                          // (or hipcc keeps the first pass's offsets in scratch for this one)
      fetch(th, 0, ea, lc2, cand, odd);
#pragma unroll
      for (u32 h = 0; h < NB; h++) {
        u32 (&cur)[QB][R][4] = h % 2 ? eb : ea;
        if (h + 1 < NB) fetch(th, (h + 1) * QB, h % 2 ? ea : eb, lc2, cand, odd);
#pragma unroll
        for (u32 q = 0; q < QB; q++) {
          const u32 ibase = (c0 + th / L + (h * QB + q) * GROUPS) * ps.tile;     // the tile of a cell completes the point index
#pragma unroll
          for (u32 r = 0; r < R; r++)
#pragma unroll
            for (u32 x = 0; x < 4; x++) {
              const u32 en = cur[q][r][x];
              if constexpr (MASKED) {
                if (en) { const u32 pos = atomicAdd(&hist[(en >> TE_FINE) & (PART_MAX_FINE - 1u)], 1u); pay[pos] = chunk_payload(en, ibase, s.stride); }
              } else {
                const u32 pos = atomicAdd(&hist[en ? (en >> TE_FINE) & (PART_MAX_FINE - 1u) : dummy], 1u);
                pay[en ? pos : C2_CAP + (t & 63u)] = chunk_payload(en, ibase, s.stride);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                  // a batch's returning atomics in flight, not all
      }
      tails(lc2, [&](u32 en, u32 ibase) {
        const u32 pos = atomicAdd(&hist[(en >> TE_FINE) & (PART_MAX_FINE - 1u)], 1u);
        pay[pos] = chunk_payload(en, ibase, s.stride);
      });
    } else {
      chunk_slow<1>(tiles, cm, ps, s.stride, c0, lo, hi, hist, pay, &sh_next);
    }
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 6);
    lds_barrier();                                                                                  // ---- 4: image complete
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 7);
    u32 tw = threadIdx.x;
    asm volatile("" : "+v"(tw));
    u32* dst = sorted + bm.img_base + lo;
#pragma unroll 4
    for (u32 q = tw; q < hi - lo; q += C2_THREADS) dst[q] = pay[q];
    // ready the next chunk: counters cleared, descriptors staged
    for (u32 f = tw; f < PART_MAX_FINE + 64; f += C2_THREADS) hist[f] = 0;
    if (k + 1 < bm.nch) {
#pragma unroll
      for (u32 i = 0; i < C2_META / C2_THREADS; i++) if (tw + i * C2_THREADS < CELLS) meta[tw + i * C2_THREADS] = mnext[i];
    }
    if (t == 0) { sh_next = ps.ntiles; sh_odd = 0; }
    c0 = c_next;
    STAMP(b == 100 && t == 0 && k < 8, k * 16 + 8);
    lds_barrier();                                                                                  // ---- 0 of the next chunk
  }
  u32 te = threadIdx.x;
  asm volatile("" : "+v"(te));                                // (2 t kept across the chunk loop lived in scratch)
#pragma unroll
  for (u32 i = 0; i < 2; i++) {
    const u32 f = 2 * te + i, g = part_bucket(ps, b, f);
    if (f < nf && g < nbuckets_total) {
      bucket_counts[g] = tot[i];
      for (u32 kk = bm.nch; kk < SEG_INLINE; kk++) reinterpret_cast<u32*>(segtab)[(size_t)g * SEG_INLINE + kk] = 0u;     // chunks the bin does not have
    }
  }
}

// ---- the bucket kernels' view of the sorted pairs ----------------------------------------------------------------------------------
// Bucket t of bin b, fine index f, owns in chunk c of its bin the entries [first, end) of the image at img_base + c * C2_CAP, where
// first | end << 16 = segtab[t][c] for c < SEG_INLINE and segoff[(chunk_first + c) * nf + f] beyond (bins of more than SEG_INLINE chunks:
// skewed scalars, or more than ~2^25 pairs per bin count). The walker holds the bucket's next four words in registers: a bucket's walk
// adds no memory access to the loop for uniform scalars.
struct SortView {
  const u32* sorted;
  const BinMeta* bins;
  const v4u_t* segtab;
  const u32* segoff;
  PartShape ps;
};
struct SegWalker {
  u32 pos, left;        // next entry of the current segment, entries left in it
  u32 w0, w1, w2, w3;   // the next segment words
  u32 c, nch;           // chunks consumed, chunks of the bin
  u32 img_base, o;      // image base of the bin in `sorted`; index of this bucket's word of chunk 0 in the chunk-major table
};
KDEV void seg_init(SegWalker& w, const SortView& v, u32 t) {
  const BinMeta bm = v.bins[part_bin(v.ps, t)];
  const v4u_t sw = v.segtab[t];
  w.w0 = sw[0]; w.w1 = sw[1]; w.w2 = sw[2]; w.w3 = sw[3];
  w.img_base = bm.img_base;
  w.o = bm.chunk_first * v.ps.nf + part_fine(v.ps, t);
  w.nch = bm.nch;
  w.c = 0; w.left = 0; w.pos = 0;
}
// next entry of the bucket; the caller asks for exactly counts[t] of them
KDEV u32 seg_next(SegWalker& w, const SortView& v) {
  if (w.left == 0) {
    u32 s, e, ci;
    do {
      s = w.w0 & 0xFFFFu; e = w.w0 >> 16; ci = w.c;
      w.w0 = w.w1; w.w1 = w.w2; w.w2 = w.w3;
      w.w3 = w.c + SEG_INLINE < w.nch ? v.segoff[w.o + (w.c + SEG_INLINE) * v.ps.nf] : 0u;
      w.c++;
    } while (e == s && w.c <= w.nch);
    w.pos = w.img_base + ci * C2_CAP + s;
    w.left = e - s;
  }
  w.left--;
  return v.sorted[w.pos++];
}
// ---- the index stream by QUADS (round 6; G1 bucket kernel) ---------------------------------------------------------------------------
// seg_next reads ONE 4-byte entry per call at a per-lane address. A lane's entries are consecutive words, but between two calls of one lane
// the wave gathers 64 table rows (and the CU's other eleven waves theirs): the 32-KB L1 has long dropped the line, so every entry is an
// L1 -> L2 request of its own (2.0 requests per table row: one row, one entry), and L2 -- 4 MiB per XCD against 3 MiB of index lines in
// use plus the stream of rows -- misses 42 % of them: each 128-byte index line came over the fabric 13.5 times, 10.9 GB of a launch's
// 37 GB (profiles/r06_bucket_clock_diagnosis.txt; the same counters with the rows confined to 2 MiB leave exactly this part). Here
// a lane fetches the ALIGNED 64-byte group (four quads) that holds its next entry into a lane-private LDS slot by LDS-DMA loads (no register
// for the data: the kernel sits at 168 of the 168 it may use) and takes its entries from there; the next group is requested when the last
// entry of the current one has been read, an iteration before its first entry is needed. ~8 entries per request on 16-entry segments.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;
// The quad walker keeps TWO words in registers (seg_next's SegWalker: ten). The bucket's next four segment words and its chunk counter are parked in
// LDS between segment changes (eight words per lane beside the index slot), the bin's image base, chunk count and row of the chunk-major table are
// read again from bins[] (16 KB, L2-resident) at every change -- once per ~16 entries. With all of it in registers the kernel's chunked modes
// spilled (12-16 B/lane) at the 168 registers three waves per SIMD allow.
struct SegWalkerQ {
  u32 pos, left;        // next entry of the current segment, entries left in it
};
constexpr u32 SEGQ_PARK = 8;        // parked words per lane: w0..w3 (the next segment words), c (chunks consumed), 3 unused
KDEV void segq_init(SegWalkerQ& w, const SortView& v, u32 t, u32* park) {
  const v4u_t sw = v.segtab[t];
  *reinterpret_cast<v4u_t*>(park) = sw;
  park[4] = 0;
  w.left = 0; w.pos = 0;
}
// to the bucket's next non-empty segment (call with left == 0); leaves left == 0 when there is none
KDEV void segq_advance(SegWalkerQ& w, const SortView& v, u32 t, u32* park) {
  const BinMeta bm = v.bins[part_bin(v.ps, t)];
  const u32 o = bm.chunk_first * v.ps.nf + part_fine(v.ps, t);
  const v4u_t sw = *reinterpret_cast<const v4u_t*>(park);
  u32 w0 = sw[0], w1 = sw[1], w2 = sw[2], w3 = sw[3], c = park[4];
  u32 s, e, ci;
  do {
    s = w0 & 0xFFFFu; e = w0 >> 16; ci = c;
    w0 = w1; w1 = w2; w2 = w3;
    w3 = c + SEG_INLINE < bm.nch ? v.segoff[o + (c + SEG_INLINE) * v.ps.nf] : 0u;
    c++;
  } while (e == s && c <= bm.nch);
  *reinterpret_cast<v4u_t*>(park) = v4u_t{w0, w1, w2, w3};
  park[4] = c;
  w.pos = bm.img_base + ci * C2_CAP + s;
  w.left = e - s;
}
// A lane's slot holds the aligned group of IDXQ_NQ quads (16 IDXQ_NQ bytes) around its next entry: quad j of every lane of the workgroup in plane j
// (256 x 16 bytes), because one LDS-DMA instruction writes lane i's 16 bytes at base + 16 i. The IDXQ_NQ requests of a group go out back to
// back and meet in the L1 like the four loads of a table row: one L1 -> L2 request per group.
constexpr u32 IDXQ_NQ = 4, IDXQ_PLANE = 256 * 4;      // quads per group; words per plane
// request the group of entry `pos` into the lane's slot: `wave_slots` is the wave's 1-KiB block of plane 0 (wave-uniform)
KDEV void segq_fetch(const SegWalkerQ& w, const SortView& v, u32* wave_slots) {
  const u32* src = v.sorted + (w.pos & ~(4u * IDXQ_NQ - 1u));
#pragma unroll
  for (u32 j = 0; j < IDXQ_NQ; j++)
    __builtin_amdgcn_global_load_lds((global_void_ptr)(src + 4 * j), (lds_void_ptr)(wave_slots + j * IDXQ_PLANE), 16, 0, 0);
}
// next entry of the bucket. The group that holds it must have LANDED: the caller has waited (s_waitcnt vmcnt(0)) since the fetch was issued.
KDEV u32 segq_next(SegWalkerQ& w, const SortView& v, u32 t, const u32* my_slot, u32* wave_slots, u32* park) {
  const u32 e = my_slot[((w.pos >> 2) & (IDXQ_NQ - 1u)) * IDXQ_PLANE + (w.pos & 3u)];
  w.pos++; w.left--;
  bool refill = (w.pos & (4u * IDXQ_NQ - 1u)) == 0;
  if (w.left == 0) { segq_advance(w, v, t, park); refill = true; }
  // `left` is re-read from its register behind the advance loop: asked as `w.left != 0` directly, hipcc (ROCm 7.2) takes the lane mask of the
  // loop's exit test e != s from the LAST trip of the loop only -- lanes that had left the loop a trip earlier (their neighbours stepped over an
  // empty segment: skewed scalars) came out with left = 0 and no request. Found by tests/test_gpu_msm_pipe.py::test_chunked_heavy_buckets.
  u32 left_now = w.left;
  asm volatile("" : "+v"(left_now));
  w.left = left_now;
  if (refill && left_now != 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the read above has left the slot before the DMA may write it
    segq_fetch(w, v, wave_slots);
  }
  return e;
}
// segment word of bucket t in chunk c (random access: the heavy-bucket kernel)
KDEV u32 seg_word(const SortView& v, u32 t, const BinMeta& bm, u32 c) {
  if (c < SEG_INLINE) return ((const u32*)v.segtab)[(size_t)t * SEG_INLINE + c];
  return v.segoff[(size_t)(bm.chunk_first + c) * v.ps.nf + part_fine(v.ps, t)];
}

// block-wide exclusive scan of one value per lane, 256 lanes
constexpr u32 SCAN_THREADS = 256;
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


// ---- bucket scheduling: process buckets in DESCENDING size order -----------------------------------------
// One lane accumulates one bucket, so a wave runs as long as its largest bucket. With Poisson bucket
// sizes (mean ~40) the largest of 64 is ~56: 30% of the lanes' issue slots idle. Ordering the buckets
// by size (counting sort on min(count, CNT_BINS-1), largest first) gives every wave equal trip counts
// and starts the heaviest buckets first. perm[t] = bucket handled by global lane t.
constexpr u32 CNT_BINS = 1024;
// Heavy buckets (see below) are found here, BY COUNT, while the counts are being read anyway: every bucket with >= HEAVY_MIN points
// takes a slot in hv->bucket[] and a contiguous range of ceil(count / HEAVY_SLICE) slice ids.
constexpr u32 HEAVY_MIN = 8192, HEAVY_SLICE = 16384;
struct HeavyList {
  u32 n;            // heavy buckets found
  u32 nslices;      // slice ids handed out
};
// capacities (host side, msm_host.hip.h): cap = pairs / HEAVY_MIN + 1 >= the number of buckets that can hold HEAVY_MIN pairs, and
// slice_cap = cap + pairs / HEAVY_SLICE + 1 >= sum of ceil(count / HEAVY_SLICE) over them: neither can overflow; the bounds checks
// in the kernels are belt and braces.
static __global__ void __launch_bounds__(256) k_cnt_hist(const u32* __restrict__ counts, u32 nb, u32* __restrict__ ghist, HeavyList* __restrict__ hv,
                                                         u32 hv_cap, u32 hv_slice_cap, u32* __restrict__ hv_bucket, u32* __restrict__ hv_first,
                                                         u32* __restrict__ hv_owner) {
  __shared__ u32 h[CNT_BINS];
  for (u32 b = threadIdx.x; b < CNT_BINS; b += 256) h[b] = 0;
  __syncthreads();
  const u32 base = blockIdx.x * 1024;
  for (u32 t = threadIdx.x; t < 1024; t += 256) {
    u32 g = base + t;
    if (g < nb) {
      u32 c = counts[g];
      atomicAdd(&h[c < CNT_BINS - 1 ? c : CNT_BINS - 1], 1u);
      if (c >= HEAVY_MIN) {
        const u32 nsl = (c + HEAVY_SLICE - 1) / HEAVY_SLICE;
        const u32 slot = atomicAdd(&hv->n, 1u);
        if (slot < hv_cap) {
          const u32 first = atomicAdd(&hv->nslices, nsl);
          hv_bucket[slot] = g;
          hv_first[slot] = first;
          for (u32 k = 0; k < nsl && first + k < hv_slice_cap; k++) hv_owner[first + k] = slot;
        }
      }
    }
  }
  __syncthreads();
  for (u32 b = threadIdx.x; b < CNT_BINS; b += 256) if (h[b]) atomicAdd(&ghist[b], h[b]);
}
// descending exclusive offsets: start[b] = number of buckets with a LARGER bin; one block of CNT_BINS threads
static __global__ void __launch_bounds__(CNT_BINS) k_cnt_offsets(const u32* __restrict__ ghist, u32* __restrict__ start) {
  __shared__ u32 sh[CNT_BINS];
  u32 b = threadIdx.x;
  sh[b] = ghist[CNT_BINS - 1 - b];   // reversed: index 0 = largest bin
  __syncthreads();
  for (u32 o = 1; o < CNT_BINS; o <<= 1) {
    u32 v = b >= o ? sh[b - o] : 0u;
    __syncthreads();
    sh[b] += v;
    __syncthreads();
  }
  start[CNT_BINS - 1 - b] = sh[b] - ghist[CNT_BINS - 1 - b];   // exclusive
}
static __global__ void __launch_bounds__(256) k_cnt_scatter(const u32* __restrict__ counts, u32 nb, u32* __restrict__ cursor /* = start, consumed */,
                                                            u32* __restrict__ perm) {
  __shared__ u32 h[CNT_BINS];      // per-block histogram, then per-block base position of each bin
  for (u32 b = threadIdx.x; b < CNT_BINS; b += 256) h[b] = 0;
  __syncthreads();
  const u32 base = blockIdx.x * 1024;
  u32 bin[4], rank[4];
#pragma unroll
  for (u32 k = 0; k < 4; k++) {
    u32 g = base + threadIdx.x + 256 * k;
    bin[k] = CNT_BINS;
    if (g < nb) { u32 c = counts[g]; bin[k] = c < CNT_BINS - 1 ? c : CNT_BINS - 1; rank[k] = atomicAdd(&h[bin[k]], 1u); }
  }
  __syncthreads();
  for (u32 b = threadIdx.x; b < CNT_BINS; b += 256) { u32 c = h[b]; if (c) h[b] = atomicAdd(&cursor[b], c); }   // reserve a range per bin
  __syncthreads();
#pragma unroll
  for (u32 k = 0; k < 4; k++) {
    u32 g = base + threadIdx.x + 256 * k;
    if (bin[k] != CNT_BINS) perm[h[bin[k]] + rank[k]] = g;
  }
}


// ---- heavy buckets ----------------------------------------------------------------------------------------------------
// Structured scalars (0/1 coefficients, repeated values) can put millions of points into ONE bucket; a single lane would
// need minutes for it. Every bucket with >= HEAVY_MIN points (found by k_cnt_hist, by count -- NOT by its place in the size order:
// the size sort clamps at CNT_BINS - 1, so thousands of mid-size buckets can stand in front of a huge one) is cut into slices of
// HEAVY_SLICE points; a fixed grid of workgroups walks the slice list, 256 lanes per slice + an LDS tree; k_msm_heavy_combine folds
// the slices of each bucket (one wave per bucket), and k_msm_accumulate* skip those buckets. Uniformly random scalars never trigger
// it (buckets hold tens to hundreds of points): the grids then exit after one load.
constexpr u32 HEAVY_GRID = 2048, HEAVY_COMBINE_GRID = 512;
// A slice is the range [lo, hi) of a bucket's pairs in the order of its segments. The workgroup finds the segments that meet the slice
// 256 chunks at a time (their lengths scanned across the lanes), then walks each of them with all lanes.
template <class F>
__global__ void __launch_bounds__(256) k_msm_heavy(const Aff<F>* __restrict__ points, SortView v, const u32* __restrict__ counts,
                                                   const HeavyList* __restrict__ hv, u32 hv_slice_cap, const u32* __restrict__ hv_bucket,
                                                   const u32* __restrict__ hv_first, const u32* __restrict__ hv_owner, Xyzz<F>* __restrict__ slices) {
  __shared__ Xyzz<F> sh[256];
  __shared__ u32 seg_pos[256], seg_n[256];
  __shared__ u32 nlist;
  const u32 total = min(hv->nslices, hv_slice_cap);
  for (u32 sid = blockIdx.x; sid < total; sid += gridDim.x) {        // block-uniform
    const u32 slot = hv_owner[sid];
    const u32 t = hv_bucket[slot];
    const u32 cnt = counts[t];
    const u32 lo = (sid - hv_first[slot]) * HEAVY_SLICE, hi = min(cnt, lo + HEAVY_SLICE);
    const BinMeta bm = v.bins[part_bin(v.ps, t)];
    Xyzz<F> acc = xyzz_inf<F>();
    u32 vbase = 0;                                                    // pairs of the bucket in the chunks before cb
    for (u32 cb = 0; cb < bm.nch && vbase < hi; cb += 256) {
      const u32 c = cb + threadIdx.x;
      u32 s = 0, e = 0;
      if (c < bm.nch) { const u32 se = seg_word(v, t, bm, c); s = se & 0xFFFFu; e = se >> 16; }
      const u32 len = e - s;
      u32 tot;
      const u32 v0 = vbase + block_exclusive_scan(len, &tot), v1 = v0 + len;
      if (threadIdx.x == 0) nlist = 0;
      __syncthreads();
      if (len && v1 > lo && v0 < hi) {
        const u32 a0 = v0 > lo ? v0 : lo, a1 = v1 < hi ? v1 : hi;
        const u32 k = atomicAdd(&nlist, 1u);
        seg_pos[k] = bm.img_base + c * C2_CAP + s + (a0 - v0);
        seg_n[k] = a1 - a0;
      }
      __syncthreads();
      const u32 nl = nlist;
      for (u32 j = 0; j < nl; j++) {
        const u32 p0 = seg_pos[j], nn = seg_n[j];
        for (u32 q = threadIdx.x; q < nn; q += 256) {
          u32 en = v.sorted[p0 + q];
          Aff<F> p = points[en & 0x7FFFFFFFu];
          acc = xyzz_add_mixed(acc, aff_cneg(p, (en >> 31) != 0));
        }
      }
      __syncthreads();
      vbase += tot;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 o2 = 128; o2 > 0; o2 >>= 1) {
      if (threadIdx.x < o2) sh[threadIdx.x] = xyzz_add(sh[threadIdx.x], sh[threadIdx.x + o2]);
      __syncthreads();
    }
    if (threadIdx.x == 0) slices[sid] = sh[0];
    __syncthreads();
  }
}
// Bucket state between the passes of a CHUNKED MSM (msm_host.hip.h: the host-pointer entries run the scalars in point-range chunks so
// that the upload of chunk j + 1 hides behind the kernels of chunk j). G2 and the saturated G1 kernel keep the canonical XYZZ value in
// `buckets`; the G1 kernel in the lazy limbs keeps its loop-carried registers AS THEY ARE (4 x 9 limbs, 144 B, no reduction, no
// conversion) in a side array and writes the canonical value only in the last pass. All limbs zero = empty bucket (ZZ of a point is
// never 0 mod p, and a lazy value with all limbs zero IS 0).
struct Acc29 {
  v4u_t q[9];
};
KDEV void acc29_store(Acc29* dst, const U29& X, const U29& Y, const U29& ZZ, const U29& ZZZ, bool empty) {
  u32 w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) { w[i] = empty ? 0u : X.l[i]; w[9 + i] = empty ? 0u : Y.l[i]; w[18 + i] = empty ? 0u : ZZ.l[i]; w[27 + i] = empty ? 0u : ZZZ.l[i]; }
#pragma unroll
  for (int i = 0; i < 9; i++) dst->q[i] = v4u_t{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
}
KDEV bool acc29_load(const Acc29* src, U29& X, U29& Y, U29& ZZ, U29& ZZZ) {      // returns `empty`
  u32 w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) { const v4u_t v = src->q[i]; w[4 * i] = v[0]; w[4 * i + 1] = v[1]; w[4 * i + 2] = v[2]; w[4 * i + 3] = v[3]; }
  u32 any = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { X.l[i] = w[i]; Y.l[i] = w[9 + i]; ZZ.l[i] = w[18 + i]; ZZZ.l[i] = w[27 + i]; any |= w[18 + i]; }
  return any == 0;
}
// mode: 0 buckets[t] = sum (one-pass MSM) | 1 buckets[t] += sum (canonical state) | 2 state29[t] = sum | 3 state29[t] += sum (G1 only)
enum { HV_SET = 0, HV_ADD = 1, HV_SET29 = 2, HV_ADD29 = 3 };
template <class F>
__global__ void __launch_bounds__(64) k_msm_heavy_combine(const u32* __restrict__ counts, const HeavyList* __restrict__ hv, u32 hv_cap, u32 hv_slice_cap,
                                                          const u32* __restrict__ hv_bucket, const u32* __restrict__ hv_first,
                                                          const Xyzz<F>* __restrict__ slices, Xyzz<F>* __restrict__ buckets, u32 mode,
                                                          Acc29* __restrict__ state29) {
  __shared__ Xyzz<F> sh[64];
  const u32 nh = min(hv->n, hv_cap);
  for (u32 slot = blockIdx.x; slot < nh; slot += gridDim.x) {
    const u32 t = hv_bucket[slot], first = hv_first[slot];
    const u32 nsl = (counts[t] + HEAVY_SLICE - 1) / HEAVY_SLICE;
    Xyzz<F> acc = xyzz_inf<F>();
    for (u32 k = threadIdx.x; k < nsl; k += 64)
      if (first + k < hv_slice_cap) acc = xyzz_add(acc, slices[first + k]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 o = 32; o > 0; o >>= 1) {
      if (threadIdx.x < o) sh[threadIdx.x] = xyzz_add(sh[threadIdx.x], sh[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      Xyzz<F> r = sh[0];
      if (mode == HV_ADD) r = xyzz_add(buckets[t], r);
      if constexpr (std::is_same<F, Fq>::value) {
        if (mode == HV_ADD29) {
          U29 X, Y, ZZ, ZZZ;
          if (!acc29_load(state29 + t, X, Y, ZZ, ZZZ)) r = xyzz_add(Xyzz<Fq>{u29_to_fq(X), u29_to_fq(Y), u29_to_fq(ZZ), u29_to_fq(ZZZ)}, r);
        }
        if (mode == HV_SET29 || mode == HV_ADD29)
          acc29_store(state29 + t, u29_from_fq(r.x), u29_from_fq(r.y), u29_from_fq(r.zz), u29_from_fq(r.zzz), xyzz_is_inf(r));
        else
          buckets[t] = r;
      } else {
        buckets[t] = r;
      }
    }
    __syncthreads();
  }
}
// the ordinary bucket kernels leave these buckets to the heavy path
KDEV bool msm_bucket_is_heavy(u32 cnt) { return cnt >= HEAVY_MIN; }

// ---- K4: bucket accumulation (dominant kernel) ----------------------------------------------------
// cont != 0 (a later pass of a chunked MSM): the bucket goes on from the value the earlier passes left in buckets[t]
template <class F>
__global__ void __launch_bounds__(256) k_msm_accumulate(const Aff<F>* __restrict__ points, SortView v, const u32* __restrict__ counts,
                                                        const u32* __restrict__ perm, u32 nbuckets_total, Xyzz<F>* __restrict__ buckets, u32 cont) {
  u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= nbuckets_total) return;
  u32 t = perm[lane];
  u32 cnt = counts[t];
  if (msm_bucket_is_heavy(cnt)) return;   // done by k_msm_heavy / k_msm_heavy_combine
  if (cont && cnt == 0) return;
  SegWalker sw;
  seg_init(sw, v, t);
  Xyzz<F> acc = xyzz_inf<F>();
  if (cont) acc = buckets[t];
  for (u32 k = 0; k < cnt; k++) {
    u32 e = seg_next(sw, v);
    Aff<F> p = points[e & 0x7FFFFFFFu];
    acc = xyzz_add_mixed(acc, aff_cneg(p, (e >> 31) != 0));
  }
  buckets[t] = acc;
}

// G1 bucket accumulation in the 9 x 29-bit lazy representation (fq29.hip.h): same schedule and memory traffic as the generic
// kernel above, ~2450 instead of ~3200 issues per mixed addition. Table / SRS rows stay in the saturated 2^256 form (the
// 2^261 form is the same integer shifted by 5 bits); buckets are written back saturated and canonical.
// Value bounds (multiples of p; measured maxima in brackets): X < 9.4 [8.7], Y < 1.6 [1.4], ZZ, ZZZ < 1.4; P = U2 - X + 16p < 17.4;
// R = S2 - Y + 4p < 5.3 [4.9]; every product of those stays < 3p.
// One 64-byte table row. A plain load makes the vector L1 fill the whole 128-byte line the row sits in: two 64-byte requests to L2 per
// row, the second one for a neighbour row nobody wants (profiles/r03_msm_2p24_l2_counters.json: 404.9 M L1->L2 read requests for 201.3 M
// rows, 1.48 fabric requests per row). NT = 1: non-temporal loads (the rows are used exactly once).
template <int NT>
KDEV Aff<Fq> msm_load_row(const Aff<Fq>* __restrict__ p) {
  if constexpr (NT == 0) {
    return *p;
  } else {
    const v4u_t* s = reinterpret_cast<const v4u_t*>(p);
    const v4u_t w0 = __builtin_nontemporal_load(s), w1 = __builtin_nontemporal_load(s + 1), w2 = __builtin_nontemporal_load(s + 2),
                w3 = __builtin_nontemporal_load(s + 3);
    Aff<Fq> q;
#pragma unroll
    for (int i = 0; i < 4; i++) { q.x.l[i] = w0[i]; q.x.l[4 + i] = w1[i]; q.y.l[i] = w2[i]; q.y.l[4 + i] = w3[i]; }
    return q;
  }
}
#ifdef KEAKI_DIAG
// diagnostic build only (Tuning::diag_row_mask): confines the gathers of the bucket kernel to the first mask + 1 table rows; sign bits stay
static __global__ void k_diag_mask_rows(u32* __restrict__ sorted, size_t words, u32 mask) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
    const u32 e = sorted[i];
    sorted[i] = (e & 0x80000000u) | (e & mask);
  }
}
#endif
// MODE (passes of a chunked MSM, see Acc29): ACC_WHOLE the one-pass MSM: starts empty, writes the canonical bucket | ACC_FIRST starts
// empty, leaves the registers in state29 | ACC_MIDDLE state29 -> state29 (buckets without pairs in this pass are not touched) |
// ACC_LAST state29 -> canonical bucket (every bucket, also the ones the heavy path owns in this pass: k_msm_heavy_combine adds to them)
enum { ACC_WHOLE = 0, ACC_FIRST = 1, ACC_MIDDLE = 2, ACC_LAST = 3 };
// PF = 0: the loop of rounds 1-4 (loads at the top of every iteration): A/B switch acc_prefetch. PF = 2 (round 6, the default): the pipeline of
// PF = 1 with the index stream by quads through LDS (segq_*): A/B switch acc_idxq.
template <int NT, int MODE, int PF = 2>
static __global__ void __launch_bounds__(256, 3) k_msm_accumulate_g1_u29(const Aff<Fq>* __restrict__ points, SortView v,
                                                                      const u32* __restrict__ counts, const u32* __restrict__ perm,
                                                                      u32 nbuckets_total, Xyzz<Fq>* __restrict__ buckets,
                                                                      Acc29* __restrict__ state29) {
  u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= nbuckets_total) return;
  u32 t = perm[lane];
  u32 cnt = counts[t];
  if constexpr (MODE == ACC_LAST) {
    if (msm_bucket_is_heavy(cnt)) cnt = 0;
  } else {
    if (msm_bucket_is_heavy(cnt)) return;   // done by k_msm_heavy / k_msm_heavy_combine
  }
  if constexpr (MODE == ACC_MIDDLE) {
    if (cnt == 0) return;
  }
  SegWalker sw;
  SegWalkerQ sq;
  if constexpr (PF != 2) seg_init(sw, v, t);
  U29 X1, Y1, ZZ, ZZZ;
  bool empty = true;
  if constexpr (MODE == ACC_MIDDLE || MODE == ACC_LAST) empty = acc29_load(state29 + t, X1, Y1, ZZ, ZZZ);
  // Software pipeline (round 5): the index of pair k + 2 and the table row of pair k + 1 are requested before pair k is added, so a wave
  // never waits for its gather -- until then every iteration began with two dependent loads (index, then the 64-byte row) that only the
  // other two waves of the SIMD could cover (alu.frac 0.94). 18 more registers: 167 of the 168 that three waves per SIMD allow.
  u32 e1 = 0, e2 = 0;
  Aff<Fq> q1;
  __shared__ __attribute__((aligned(16))) u32 idx_slots[PF == 2 ? IDXQ_NQ * IDXQ_PLANE : 4];       // PF == 2: IDXQ_NQ quads of the index stream per lane
  __shared__ __attribute__((aligned(16))) u32 idx_park[PF == 2 ? 256 * SEGQ_PARK : 4];             //          and the walker's parked words
  const u32* my_slot = idx_slots + threadIdx.x * 4;
  u32* wave_slots = idx_slots + (threadIdx.x & ~63u) * 4;
  u32* park = idx_park + threadIdx.x * SEGQ_PARK;
  if constexpr (PF == 2) {
    segq_init(sq, v, t, park);
    if (cnt > 0) {
      segq_advance(sq, v, t, park);
      segq_fetch(sq, v, wave_slots);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      e1 = segq_next(sq, v, t, my_slot, wave_slots, park);
      if (cnt > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the first call may have requested the next quad
        e2 = segq_next(sq, v, t, my_slot, wave_slots, park);
      }
    }
    q1 = msm_load_row<NT>(points + (e1 & 0x7FFFFFFFu));
  } else if constexpr (PF != 0) {
    e1 = cnt > 0 ? seg_next(sw, v) : 0u; e2 = cnt > 1 ? seg_next(sw, v) : 0u;
    q1 = msm_load_row<NT>(points + (e1 & 0x7FFFFFFFu));
  }
  for (u32 k = 0; k < cnt; k++) {
    u32 e;
    Aff<Fq> q;
    if constexpr (PF == 2) {
      e = e1;
      // row k (requested an iteration ago) is needed now, and so is the quad a call of the last iteration may have requested: one wait for both,
      // BEFORE row k + 1 is requested (a wait behind that request would expose its latency)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      q = q1;
      // the copy happens HERE: left to itself hipcc sinks these moves below the quad request, and -- a use of a load result while an LDS-DMA is in
      // flight -- puts an s_waitcnt vmcnt(0) in front of them that waits for the request just made
#pragma unroll
      for (int i = 0; i < 8; i++) { asm volatile("" : "+v"(q.x.l[i])); asm volatile("" : "+v"(q.y.l[i])); }
      u32 e3 = 0;
      if (k + 2 < cnt) e3 = segq_next(sq, v, t, my_slot, wave_slots, park);
      e1 = e2;
      if (k + 1 < cnt) q1 = msm_load_row<NT>(points + (e1 & 0x7FFFFFFFu));
      e2 = e3;
    } else if constexpr (PF != 0) {
      e = e1;
      q = q1;
      e1 = e2;
      if (k + 1 < cnt) q1 = msm_load_row<NT>(points + (e1 & 0x7FFFFFFFu));
      if (k + 2 < cnt) e2 = seg_next(sw, v);
    } else {
      e = seg_next(sw, v);
      q = msm_load_row<NT>(points + (e & 0x7FFFFFFFu));
    }
    if (aff_is_inf(q)) continue;
    q.y = f_cneg(q.y, (e >> 31) != 0);
    const U29 X2 = u29_from_sat_shift5(q.x.l), Y2 = u29_from_sat_shift5(q.y.l);
    if (empty) {
      X1 = u29_mul(X2, u29_one());
      Y1 = u29_mul(Y2, u29_one());
      ZZ = u29_one();
      ZZZ = u29_one();
      empty = false;
      continue;
    }
    const U29 U2 = u29_mul(X2, ZZ), S2 = u29_mul(Y2, ZZZ);
    const U29 P = u29_sub(U2, X1, Q29::K16), R = u29_sub(S2, Y1, Q29::K4);
    if (u29_maybe_zero(P)) {            // 18 in 2^29 for unrelated points; exact test only then
      if (u29_is_zero(P)) {
        if (u29_is_zero(R)) {           // same point: double it in the saturated arithmetic, re-enter
          Xyzz<Fq> d = xyzz_dbl_aff(q);
          X1 = u29_from_fq(d.x); Y1 = u29_from_fq(d.y); ZZ = u29_from_fq(d.zz); ZZZ = u29_from_fq(d.zzz);
        } else {
          empty = true;                 // opposite points
        }
        continue;
      }
    }
    const U29 PP = u29_sqr(P), PPP = u29_mul(P, PP), Q = u29_mul(X1, PP);
    const U29 X3 = u29_sub3(u29_sqr(R), PPP, Q);
    const U29 T = u29_sub(Q, X3, Q29::K16);
    // Y3 = R T - Y1 PPP as ONE double-width column pass with a single reduction: R T + (2p - Y1) PPP  (Y1 < 1.6p, limbs exact)
    U29 NY1;
#pragma unroll
    for (int i = 0; i < 9; i++) NY1.l[i] = Q29::K2[i] - Y1.l[i];
    Y1 = u29_mul2(R, T, NY1, PPP);
    X1 = X3;
    ZZ = u29_mul(ZZ, PP);
    ZZZ = u29_mul(ZZZ, PPP);
  }
  // the addresses of the final stores are formed HERE, from a copy of t the compiler cannot connect with the one the state was loaded through: kept
  // from the top of the kernel (the chunked modes load through state29 + t) the 64-bit address was spilled across the loop (12-16 B/lane of scratch)
  u32 ts = t;
  asm volatile("" : "+v"(ts));
  if constexpr (MODE == ACC_FIRST || MODE == ACC_MIDDLE) {
    acc29_store(state29 + ts, X1, Y1, ZZ, ZZZ, empty);
  } else {
    Xyzz<Fq> out = xyzz_inf<Fq>();
    if (!empty) {
      out.x = u29_to_fq(X1); out.y = u29_to_fq(Y1); out.zz = u29_to_fq(ZZ); out.zzz = u29_to_fq(ZZZ);
    }
    buckets[ts] = out;
  }
}

// G2 bucket accumulation in the lazy limbs (xyzz29_g2.hip.h): same schedule, ~5,600 instead of ~9,000 instructions per mixed addition.
// Buckets are written back saturated and canonical (the tail keeps the generic arithmetic).
template <int CONT>
static __global__ void __launch_bounds__(256, 2) k_msm_accumulate_g2_u29(const Aff<Fq2>* __restrict__ points, SortView v,
                                                                      const u32* __restrict__ counts, const u32* __restrict__ perm,
                                                                      u32 nbuckets_total, Xyzz<Fq2>* __restrict__ buckets) {
  u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= nbuckets_total) return;
  u32 t = perm[lane];
  u32 cnt = counts[t];
  if (msm_bucket_is_heavy(cnt)) return;   // done by k_msm_heavy / k_msm_heavy_combine
  if (CONT && cnt == 0) return;
  // the index stream as in the G1 kernel (round 6): aligned 64-byte groups through a lane-private LDS slot, the walker's segment words parked in LDS
  // (eight registers fewer across the addition: the one dword this kernel reloaded from scratch in every iteration is gone; what -Rpass still
  // reports as scratch is the call frame of the out-of-line saturated Fq2 products of the same-point path, 18 in 2^29 additions)
  __shared__ __attribute__((aligned(16))) u32 idx_slots[IDXQ_NQ * IDXQ_PLANE];
  __shared__ __attribute__((aligned(16))) u32 idx_park[256 * SEGQ_PARK];
  const u32* my_slot = idx_slots + threadIdx.x * 4;
  u32* wave_slots = idx_slots + (threadIdx.x & ~63u) * 4;
  u32* park = idx_park + threadIdx.x * SEGQ_PARK;
  SegWalkerQ sq;
  segq_init(sq, v, t, park);
  X29G2 acc = x29g2_inf();
  if constexpr (CONT != 0) acc = x29g2_load(buckets[t]);
  if (cnt > 0) {
    segq_advance(sq, v, t, park);
    segq_fetch(sq, v, wave_slots);
  }
  for (u32 k = 0; k < cnt; k++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the group requested by the call of the last iteration (or in front of the loop) has landed
    const u32 e = segq_next(sq, v, t, my_slot, wave_slots, park);
    Aff<Fq2> q = points[e & 0x7FFFFFFFu];
    x29g2_add_mixed(acc, aff_cneg(q, (e >> 31) != 0));
  }
  u32 ts = t;
  asm volatile("" : "+v"(ts));
  buckets[ts] = x29g2_store(acc);
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
  const u32 Bw = msm_nbuckets(s, w);
  const Xyzz<F>* bk = buckets + msm_bucket_base(s, w);
  typedef TailOps<F> O;
  typename O::P run = O::inf(), ws = O::inf();
  for (u32 j = L; j-- > 0;) {
    if (lo + j < Bw) {
      run = O::add(run, O::load(bk[lo + j]));
      ws = O::add(ws, run);
    }
  }
  // ws += lo * run   (lo < 2^c), MSB-first double-and-add
  if (lo != 0 && lo < Bw) {
    typename O::P m = O::inf();
    for (int b = 31 - __clz(lo); b >= 0; b--) {
      m = O::dbl(m);
      if ((lo >> b) & 1) m = O::add(m, run);
    }
    ws = O::add(ws, m);
  }
  partials[g] = O::store(ws);
}

// ---- K5b: sum groups of G consecutive chunk partials of each window (keeps K6a's serial part short) -------------
template <class F>
__global__ void __launch_bounds__(64) k_msm_partial_groups(const Xyzz<F>* __restrict__ in, u32 chunks_in, u32 G, u32 chunks_out,
                                                           Xyzz<F>* __restrict__ out) {
  typedef TailOps<F> O;
  __shared__ typename O::P sh[64];
  u32 w = blockIdx.y, x = blockIdx.x, l = threadIdx.x;
  typename O::P acc = O::inf();
  for (u32 t = x * G + l; t < (x + 1) * G && t < chunks_in; t += 64) acc = O::add(acc, O::load(in[(size_t)w * chunks_in + t]));
  sh[l] = acc;
  __syncthreads();
  for (u32 o = 32; o > 0; o >>= 1) {
    if (l < o) sh[l] = O::add(sh[l], sh[l + o]);
    __syncthreads();
  }
  if (l == 0) out[(size_t)w * chunks_out + x] = O::store(sh[0]);
}

// ---- K6a: sum the chunk partials of each window, then scale by 2^(w c) ---------------------------------
template <class F> KDEV void store_norm_jac(F* out, const Xyzz<F>& p);
// out_jac != nullptr (one window: the shared-bucket path): the window sum IS the result -- normalised and written here, no k_msm_final launch
template <class F>
__global__ void __launch_bounds__(64) k_msm_window_finish(const Xyzz<F>* __restrict__ partials, MsmShape s, u32 chunks_per_window,
                                                          Xyzz<F>* __restrict__ window_sums, F* __restrict__ out_jac) {
  typedef TailOps<F> O;
  __shared__ typename O::P sh[64];
  u32 w = blockIdx.x, l = threadIdx.x;
  typename O::P acc = O::inf();
  for (u32 t = l; t < chunks_per_window; t += 64) acc = O::add(acc, O::load(partials[(size_t)w * chunks_per_window + t]));
  sh[l] = acc;
  __syncthreads();
  for (u32 o = 32; o > 0; o >>= 1) {
    if (l < o) sh[l] = O::add(sh[l], sh[l + o]);
    __syncthreads();
  }
  if (l == 0) {
    typename O::P r = sh[0];
    for (u32 k = 0, nd = msm_bit_offset(s, w); k < nd; k++) r = O::dbl(r);
    if (out_jac) store_norm_jac(out_jac, O::store(r));
    else window_sums[w] = O::store(r);
  }
}

// write a point as normalised Jacobian (x, y, 1) / (1, 1, 0)
template <class F> KDEV Aff<F> xyzz_to_aff_tail(const Xyzz<F>& p) { return xyzz_to_aff(p); }
template <> KDEV Aff<Fq> xyzz_to_aff_tail<Fq>(const Xyzz<Fq>& p) {   // single lane: the binary-GCD inverse instead of the Fermat ladder
  if (xyzz_is_inf(p)) return aff_inf<Fq>();
  Fq izzz = fq_inv(p.zzz);            // division steps (safegcd): ~22 K plain instructions, no data-dependent trip count
  Fq izz = fq_sqr(izzz) * fq_sqr(p.zz);
  return {p.x * izz, p.y * izzz};
}
template <class F>
KDEV void store_norm_jac(F* out, const Xyzz<F>& p) {
  Aff<F> a = xyzz_to_aff_tail(p);
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


// ---- precomputed window tables (one-time, at SRS upload) ------------------------------------------------------
// table[w * N + i] = 2^(bit offset of window w) * P_i, affine. Lane i walks the windows, doubling
// width(w-1) times between them (Jacobian) and normalising each multiple (one inversion each).
template <class F>
__global__ void __launch_bounds__(64) k_msm_build_tables(const Aff<F>* __restrict__ points, u32 N, MsmShape s, Aff<F>* __restrict__ table) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  Aff<F> p = points[i];
  table[i] = p;
#pragma unroll 1
  for (u32 w = 1; w < s.W; w++) {
    Jac<F> j = jac_from_aff(p);
    for (u32 d = msm_width(s, w - 1); d > 0; d--) j = jac_dbl(j);
    p = jac_to_aff(j);
    table[(size_t)w * N + i] = p;
  }
}

// G1 version of the table build in the 29-bit lazy arithmetic with ONE inversion per point instead of one per window: the lane keeps
// the running Jacobian multiple (never renormalised), parks X and Y of every window in the table slot, keeps the W - 1 Z's in
// registers, inverts their product once and walks back (the partial products of the Z's are recomputed: ~W^2/2 products, nothing
// against the W - 1 Fermat ladders of ~380 products each that the generic kernel spends).
constexpr u32 TABLE_MAX_W = 16;
static __global__ void __launch_bounds__(64) k_msm_build_tables_g1(const Aff<Fq>* __restrict__ points, u32 N, MsmShape s, Aff<Fq>* __restrict__ table) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Aff<Fq> p = points[i];
  table[i] = p;
  if (aff_is_inf(p)) {                             // every multiple of the identity is the identity
    for (u32 w = 1; w < s.W; w++) table[(size_t)w * N + i] = p;
    return;
  }
  J29 j;
  j.x = u29_from_fq(p.x); j.y = u29_from_fq(p.y); j.z = u29_one();
  U29 zs[TABLE_MAX_W];
  U29 prod = u29_one();
#pragma unroll 1
  for (u32 w = 1; w < s.W; w++) {
    for (u32 d = msm_width(s, w - 1); d > 0; d--) j = j29_dbl(j);
    Aff<Fq> tmp = {u29_to_fq(j.x), u29_to_fq(j.y)};       // Jacobian X, Y parked in the slot
    table[(size_t)w * N + i] = tmp;
#pragma unroll
    for (u32 k = 1; k < TABLE_MAX_W; k++) if (k == w) zs[k] = j.z;
    prod = u29_mul(prod, j.z);
  }
  // 1 / (Z_1 ... Z_(W-1))
  U29 tinv = u29_from_fq(fq_inv(u29_to_fq(prod)));
#pragma unroll 1
  for (u32 w = s.W - 1; w >= 1; w--) {
    U29 pre = u29_one(), zw = u29_one();
#pragma unroll
    for (u32 k = 1; k < TABLE_MAX_W; k++) {
      if (k < w) pre = u29_mul(pre, zs[k]);
      if (k == w) zw = zs[k];
    }
    const U29 zinv = u29_mul(tinv, pre);            // 1 / Z_w
    tinv = u29_mul(tinv, zw);                        // 1 / (Z_1 ... Z_(w-1))
    const U29 zi2 = u29_sqr(zinv), zi3 = u29_mul(zi2, zinv);
    const Aff<Fq> jxy = table[(size_t)w * N + i];
    Aff<Fq> a = {u29_to_fq(u29_mul(u29_from_sat_shift5(jxy.x.l), zi2)), u29_to_fq(u29_mul(u29_from_sat_shift5(jxy.y.l), zi3))};
    table[(size_t)w * N + i] = a;
  }
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
