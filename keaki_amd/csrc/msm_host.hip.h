// Host-side driver of the MSM pipeline (workspace sizing, window choice, kernel sequence), written
// once and instantiated for G1 (msm_g1.hip) and G2 (msm_g2.hip).
#pragma once
#include <algorithm>
#include <functional>
#include <type_traits>
#include <vector>
#include "internal.h"
#include "msm.hip.h"

namespace keaki_internal {
using namespace bn254;

// GPU window choice: minimise (n * W mixed adds) + (2 * W * B full adds, ~1.4x a mixed add each)
// while keeping enough buckets (= lanes of the accumulate kernel) to fill 256 CUs.
inline int choose_window(size_t n, int forced = 0) {
  if (forced >= 3 && forced <= 24) return forced;         // Tuning::msm_c
  if (n < 32) return 3;
  double best = 1e300;
  int bc = 3;
  for (int c = 3; c <= 22; c++) {
    MsmPlan p = msm_make_plan(n, c);
    if (p.nb > ((size_t)PART_MAX_BINS << PART_MAX_FINE_SHIFT)) continue;      // the two-pass partition addresses 2048 x 2048 buckets
    double cost = (double)n * p.s.W + 2.8 * (double)p.nb;
    if ((double)p.nb < 131072.0) cost *= 131072.0 / (double)p.nb;   // too few buckets cannot fill 256 CUs
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

// d_table != nullptr: precomputed path (tables built by msm_build_tables with window target c_table for N = srs_len points)
template <class F>
keaki_status msm_dev(keaki_hip_ctx* ctx, const Aff<F>* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac,
                     const Aff<F>* d_table = nullptr, int c_table = 0, const MsmPipe* pipe = nullptr) {
  if (!d_out_jac || (n && (!d_points || !d_scalars))) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: null pointer");
  if (n > srs_len) return fail(ctx, KEAKI_ERR_TOO_LARGE, "msm: %zu scalars but the SRS holds %zu points", n, srs_len);
  if (n >= (1ull << 31)) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: n must be < 2^31 per device");
  // An SRS with window tables uses them for EVERY length. Until round 4 a polynomial shorter than half of the SRS took the generic path ("its own
  // window size is faster"): true for the additions, but the generic tail is a chain of ~250 doublings and W additions in one lane -- 1.8-2.3 ms
  // whatever n is, against 0.4-0.7 ms for the one shared window (bench_tools/ab_short_msm.py: SRS 2^10 ... 2^24, n = 1 ... srs/2: the tables win
  // every cell, 3-5 x at the lengths of BASELINE config 1). Option msm_short_tables = 0 brings the old rule back (A/B, tests).
  const bool shared = d_table != nullptr && (n * 2 > srs_len || ctx->tune.msm_short_tables != 0);
  const MsmPlan plan = shared ? msm_make_plan(n, c_table) : msm_make_plan(n, choose_window(n, ctx->tune.msm_c));
  MsmShape s = plan.s;
  // reduction shape: generic = the plan itself; shared = ONE window holding max_b = 2^cr buckets (top-window rule: 2^width buckets)
  MsmShape rs = s;
  size_t nb = plan.nb;
  if (shared) {
    s.stride = (u32)srs_len;
    u32 cr = 0;
    while ((1u << cr) < plan.max_b) cr++;
    rs.n = s.n; rs.c = cr; rs.W = 1; rs.k = 1; rs.stride = 0;
    nb = plan.max_b;
    d_points = d_table;
    if ((double)srs_len * s.W >= 2147483647.0) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: precomputed table index overflows 31 bits");
  }
  ctx->last_c = (int)s.c;
  if ((double)n * s.W >= 4294967295.0) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: n * windows overflows 32-bit positions");
  // reduce chunk length: the kernel is a serial chain of 2L additions (+ a scalar multiplication by the chunk index) per lane; measured
  // (bench_tools/sweep_reduce_l.py, round 3, whole MSM with tables): 2^21 buckets (2^23, 2^24 points): 32 -- 9.42 / 17.69 ms against 9.60 /
  // 17.88 with 16; 2^19 buckets (2^21, 2^22 points): 8 -- 2.92 / 5.31 ms against 2.99 / 5.41; 16 in between; 8 below
  // (G2, whose additions cost 2.3 x as much: 8 from 2^19 buckets on as well -- 5.10 vs 5.34 ms at 2^20 points)
  u32 L = sizeof(F) > sizeof(Fq) ? (plan.max_b >= 64 ? 8 : plan.max_b)
                                 : plan.max_b >= (1u << 21) ? 32 : plan.max_b >= (1u << 20) ? 16 : (plan.max_b >= 64 ? 8 : plan.max_b);
  if (ctx->tune.reduce_l >= 1 && ctx->tune.reduce_l <= 4096 && (u32)ctx->tune.reduce_l <= plan.max_b) L = (u32)ctx->tune.reduce_l;
  const u32 chunks = cdiv(plan.max_b, L);
  ST_TRY(reserve(ctx, ctx->wsums, (size_t)rs.W * sizeof(Xyzz<F>)));
  Xyzz<F>* wsums = (Xyzz<F>*)ctx->wsums.p;
  F* out = (F*)d_out_jac;
  hipStream_t st = ctx->stream;
  if (ctx->timing) (void)hipEventRecord(ctx->ev[0], st);
  if (n == 0) {
    hipLaunchKernelGGL((k_msm_final<F>), dim3(1), dim3(64), 0, st, (const Xyzz<F>*)wsums, 0u, out);
    return launch_check(ctx, "msm_final");
  }
  // ---- the passes: one for a resident scalar vector, one per chunk of a pipelined call ----------------------------------------------
  const size_t K = pipe ? pipe->ranges.size() : 1;
  if (pipe && K < 1) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: bad chunk bounds");
  struct Pass {
    size_t lo, m, pairs, max_chunks, cm_bytes, bt_bytes, bm_bytes, sg_bytes, ts_bytes;
    PartShape ps;
  };
  std::vector<Pass> passes(K);
  size_t w_digits = 0, w_sorted = 0, w_offsets = 0, w_cursor = 0, w_pairs = 0, w_total = 0;
  for (size_t j = 0; j < K; j++) {
    Pass& q = passes[j];
    q.lo = pipe ? pipe->ranges[j].first : 0;
    q.m = pipe ? pipe->ranges[j].second : n;
    if (q.m == 0 || q.lo > n || q.m > n - q.lo) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: bad chunk bounds");
    w_total += q.m;
    if (!part_make_shape(q.m, s.W, nb, &q.ps, ctx->tune.part_shift))
      return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: %zu buckets / %u windows exceed the bucket sort's LDS budget (window too large)", nb, s.W);
    q.pairs = q.m * (size_t)s.W;
    q.max_chunks = part_max_chunks(q.pairs, q.ps.nbins);
    // [cell table: (position in the bin, start | length in the tile) per bin and tile | bin totals | bin descriptors | bucket-major segment
    //  words | tile starts (u16)]
    q.cm_bytes = (size_t)q.ps.nbins * q.ps.ntiles * sizeof(uint2);
    q.bt_bytes = ((size_t)q.ps.nbins * 4 + 15) & ~(size_t)15;
    q.bm_bytes = (size_t)q.ps.nbins * sizeof(BinMeta);
    q.sg_bytes = nb * sizeof(v4u_t);
    q.ts_bytes = (size_t)q.ps.ntiles * (q.ps.nbins + 1) * 2;
    w_digits = std::max(w_digits, (size_t)q.ps.ntiles * q.ps.te * 4);
    w_sorted = std::max(w_sorted, q.pairs * 4 + 64);      // + 64: the bucket kernel reads the index stream in aligned 64-byte groups (segq_fetch)
    w_offsets = std::max(w_offsets, q.max_chunks * q.ps.nf * 4);
    w_cursor = std::max(w_cursor, q.cm_bytes + q.bt_bytes + q.bm_bytes + q.sg_bytes + q.ts_bytes);
    w_pairs = std::max(w_pairs, q.pairs);
  }
  if (w_total != n) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: the chunks cover %zu of %zu pairs", w_total, n);
  bool u29 = false;
  if constexpr (std::is_same<F, Fq>::value) u29 = ctx->tune.acc_u29;          // A/B switches for profiling
  if constexpr (std::is_same<F, Fq2>::value) u29 = ctx->tune.acc_u29_g2;
  bool lazy_state = K > 1 && u29 && std::is_same<F, Fq>::value;               // the G1 kernel's registers stay in Acc29 between the passes
  // every workspace is reserved BEFORE the first pass (sized for the largest one): a reserve that grows a buffer waits for the stream
  // pass-1 images | bucket-ordered index stream | per-bucket counts | chunk-major segment words of the chunks beyond SEG_INLINE
  ST_TRY(reserve(ctx, ctx->digits, w_digits));
  ST_TRY(reserve(ctx, ctx->sorted, w_sorted));
  ST_TRY(reserve(ctx, ctx->hist, nb * 4));
  ST_TRY(reserve(ctx, ctx->offsets, w_offsets));
  ST_TRY(reserve(ctx, ctx->cursor, w_cursor));
  ST_TRY(reserve(ctx, ctx->buckets, nb * sizeof(Xyzz<F>)));
  ST_TRY(reserve(ctx, ctx->partials, ((size_t)rs.W * chunks + (size_t)rs.W * 256) * sizeof(Xyzz<F>)));
  ST_TRY(reserve(ctx, ctx->perm, nb * 4 + 2 * CNT_BINS * 4 + sizeof(HeavyList)));
  // heavy-bucket list: [bucket[cap] | first[cap] | owner[slice_cap]] then the slice sums
  const u32 hv_cap = (u32)(w_pairs / HEAVY_MIN + 1), hv_slice_cap = (u32)(hv_cap + w_pairs / HEAVY_SLICE + 1);
  const size_t hv_hdr = ((2 * (size_t)hv_cap + hv_slice_cap) * 4 + 255) & ~(size_t)255;
  ST_TRY(reserve(ctx, ctx->heavy, hv_hdr + (size_t)hv_slice_cap * sizeof(Xyzz<F>)));
  if (lazy_state) {
    // 144 B per bucket on top of the canonical 128 (302 MB at 2^21 buckets): optional memory like the window tables -- when it does not fit, the
    // passes go on from the canonical bucket through the saturated kernel (slower, same result) instead of failing commit / open
    const keaki_status st29 = reserve(ctx, ctx->acc29, nb * sizeof(Acc29));
    if (st29 == KEAKI_ERR_OOM) { lazy_state = false; u29 = false; ctx->err.clear(); }
    else if (st29 != KEAKI_OK) return st29;
  }
  u32 *tiles = (u32*)ctx->digits.p, *sorted = (u32*)ctx->sorted.p, *hist = (u32*)ctx->hist.p;
  u32* segoff = (u32*)ctx->offsets.p;
  Xyzz<F>* buckets = (Xyzz<F>*)ctx->buckets.p;
  Xyzz<F>* partials = (Xyzz<F>*)ctx->partials.p;
  Acc29* state29 = lazy_state ? (Acc29*)ctx->acc29.p : nullptr;
  u32* perm = (u32*)ctx->perm.p;
  u32 *gstart = perm + nb, *ghist = gstart + CNT_BINS;
  HeavyList* hv = (HeavyList*)(ghist + CNT_BINS);                        // right behind the histogram: one memset clears both
  u32 *hv_bucket = (u32*)ctx->heavy.p, *hv_first = hv_bucket + hv_cap, *hv_owner = hv_first + hv_cap;
  Xyzz<F>* hv_slices = (Xyzz<F>*)((char*)ctx->heavy.p + hv_hdr);
  for (size_t j = 0; j < K; j++) {
    const Pass& q = passes[j];
    const PartShape& ps = q.ps;
    if (pipe && pipe->stage) ST_TRY(pipe->stage(j));
    MsmShape sj = s;
    sj.n = (u32)q.m;
    const Fr* scal = (const Fr*)d_scalars + q.lo;
    const Aff<F>* pts = d_points + q.lo;                  // table row w of point lo + i = (table + lo)[w * stride + i]
    const bool first = j == 0, last = j + 1 == K;
    uint2* cellmeta = (uint2*)ctx->cursor.p;
    u32* bin_total = (u32*)((char*)ctx->cursor.p + q.cm_bytes);
    BinMeta* bins = (BinMeta*)((char*)bin_total + q.bt_bytes);
    v4u_t* segtab = (v4u_t*)((char*)bins + q.bm_bytes);
    u16* tstart = (u16*)((char*)segtab + q.sg_bytes);
    {
      const dim3 g1(ps.ntiles < ctx->n_cu ? ps.ntiles : ctx->n_cu), b1(T1_THREADS);
#define KEAKI_TILE_SORT(WS) hipLaunchKernelGGL(k_tile_sort<WS>, g1, b1, 0, st, scal, sj, ps, tiles, tstart)
      switch (s.W) {                    // plans with 11..16 windows (what 2^16..2^26 points choose) have their digit cuts compiled in
        case 11: KEAKI_TILE_SORT(11); break;
        case 12: KEAKI_TILE_SORT(12); break;
        case 13: KEAKI_TILE_SORT(13); break;
        case 14: KEAKI_TILE_SORT(14); break;
        case 15: KEAKI_TILE_SORT(15); break;
        case 16: KEAKI_TILE_SORT(16); break;
        default: KEAKI_TILE_SORT(0); break;
      }
#undef KEAKI_TILE_SORT
    }
    ST_TRY(launch_check(ctx, "tile_sort"));
    hipLaunchKernelGGL(k_cell_prefix, dim3(ps.nbins), dim3(1024), 0, st, (const u16*)tstart, ps, cellmeta, bin_total);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, st, (const u32*)bin_total, ps.nbins, bins);
    {
#define KEAKI_CHUNK_SORT1(L, R, Q, M)                                                                                                                  \
  hipLaunchKernelGGL((k_chunk_sort<L, R, Q, M>), dim3(ps.nbins), dim3(C2_THREADS), 0, st, (const u32*)tiles, (const uint2*)cellmeta, (const BinMeta*)bins, sj, \
                     ps, (u32)nb, sorted, segtab, segoff, hist)
#define KEAKI_CHUNK_SORT(L, R, Q) do { if (ctx->tune.cs_masked) KEAKI_CHUNK_SORT1(L, R, Q, true); else KEAKI_CHUNK_SORT1(L, R, Q, false); } while (0)
      switch (ps.geom) {                // lanes per cell, 16-byte pieces per lane and cell, cells per group: for cells of about 18 / 36 / 72 / 144+ entries
        case 0: KEAKI_CHUNK_SORT(8, 1, 16); break;
        case 1: KEAKI_CHUNK_SORT(16, 1, 16); break;
        case 2: KEAKI_CHUNK_SORT(16, 2, 8); break;
        default: KEAKI_CHUNK_SORT(16, 4, 4); break;
      }
#undef KEAKI_CHUNK_SORT1
#undef KEAKI_CHUNK_SORT
    }
    ST_TRY(launch_check(ctx, "chunk_sort"));
#ifdef KEAKI_DIAG
    if (ctx->tune.diag_row_mask) hipLaunchKernelGGL(k_diag_mask_rows, dim3(4096), dim3(256), 0, st, sorted, (size_t)q.pairs, (u32)ctx->tune.diag_row_mask);
#endif
    const SortView view = {sorted, bins, segtab, segoff, ps};
    // bucket schedule: descending size
    HIP_TRY(ctx, hipMemsetAsync(ghist, 0, CNT_BINS * 4 + sizeof(HeavyList), st));
    hipLaunchKernelGGL(k_cnt_hist, dim3(cdiv(nb, 1024)), dim3(256), 0, st, (const u32*)hist, (u32)nb, ghist, hv, hv_cap, hv_slice_cap, hv_bucket, hv_first,
                       hv_owner);
    hipLaunchKernelGGL(k_cnt_offsets, dim3(1), dim3(CNT_BINS), 0, st, (const u32*)ghist, gstart);
    hipLaunchKernelGGL(k_cnt_scatter, dim3(cdiv(nb, 1024)), dim3(256), 0, st, (const u32*)hist, (u32)nb, gstart, perm);
    ST_TRY(launch_check(ctx, "cnt_sort"));
    if (ctx->timing && last) (void)hipEventRecord(ctx->ev[1], st);
    const dim3 ga(cdiv(nb, 256)), ba(256);
    bool done = false;
    if constexpr (std::is_same<F, Fq>::value) {
      if (u29) {
#define KEAKI_ACC(NT, MODE) hipLaunchKernelGGL((k_msm_accumulate_g1_u29<NT, MODE>), ga, ba, 0, st, pts, view, (const u32*)hist, (const u32*)perm, (u32)nb, buckets, state29)
        if (K == 1) {
          if (ctx->tune.acc_nt) KEAKI_ACC(1, ACC_WHOLE);
          else if (!ctx->tune.acc_prefetch)
            hipLaunchKernelGGL((k_msm_accumulate_g1_u29<0, ACC_WHOLE, 0>), ga, ba, 0, st, pts, view, (const u32*)hist, (const u32*)perm, (u32)nb, buckets, state29);
          else if (!ctx->tune.acc_idxq)
            hipLaunchKernelGGL((k_msm_accumulate_g1_u29<0, ACC_WHOLE, 1>), ga, ba, 0, st, pts, view, (const u32*)hist, (const u32*)perm, (u32)nb, buckets, state29);
          else KEAKI_ACC(0, ACC_WHOLE);
        }
        else if (first) KEAKI_ACC(0, ACC_FIRST);
        else if (!last) KEAKI_ACC(0, ACC_MIDDLE);
        else KEAKI_ACC(0, ACC_LAST);
#undef KEAKI_ACC
        done = true;
      }
    }
    if constexpr (std::is_same<F, Fq2>::value) {
      if (u29) {
        if (first) hipLaunchKernelGGL(k_msm_accumulate_g2_u29<0>, ga, ba, 0, st, pts, view, (const u32*)hist, (const u32*)perm, (u32)nb, buckets);
        else hipLaunchKernelGGL(k_msm_accumulate_g2_u29<1>, ga, ba, 0, st, pts, view, (const u32*)hist, (const u32*)perm, (u32)nb, buckets);
        done = true;
      }
    }
    if (!done)
      hipLaunchKernelGGL((k_msm_accumulate<F>), ga, ba, 0, st, pts, view, (const u32*)hist, (const u32*)perm, (u32)nb, buckets, first ? 0u : 1u);
    ST_TRY(launch_check(ctx, "msm_accumulate"));
    // heavy buckets (structured scalars only; the grids exit after one load otherwise)
    const u32 hv_mode = lazy_state && !last ? (first ? HV_SET29 : HV_ADD29) : (first ? HV_SET : HV_ADD);
    hipLaunchKernelGGL((k_msm_heavy<F>), dim3(HEAVY_GRID), dim3(256), 0, st, pts, view, (const u32*)hist,
                       (const HeavyList*)hv, hv_slice_cap, (const u32*)hv_bucket, (const u32*)hv_first, (const u32*)hv_owner, hv_slices);
    hipLaunchKernelGGL((k_msm_heavy_combine<F>), dim3(HEAVY_COMBINE_GRID), dim3(64), 0, st, (const u32*)hist, (const HeavyList*)hv, hv_cap, hv_slice_cap,
                       (const u32*)hv_bucket, (const u32*)hv_first, (const Xyzz<F>*)hv_slices, buckets, hv_mode, state29);
    ST_TRY(launch_check(ctx, "msm_heavy"));
  }
  if (ctx->timing) (void)hipEventRecord(ctx->ev[2], st);
  hipLaunchKernelGGL((k_msm_reduce<F>), dim3(cdiv((size_t)rs.W * chunks, 64)), dim3(64), 0, st, (const Xyzz<F>*)buckets, rs, L, chunks, partials);
  // chunk partials -> (at most 128 per window) -> window sums
  const Xyzz<F>* fin_in = partials;
  u32 fin_chunks = chunks;
  if (chunks > 256) {
    const u32 G = cdiv(chunks, 128);
    const u32 chunks2 = cdiv(chunks, G);
    Xyzz<F>* partials2 = partials + (size_t)rs.W * chunks;
    hipLaunchKernelGGL((k_msm_partial_groups<F>), dim3(chunks2, rs.W), dim3(64), 0, st, (const Xyzz<F>*)partials, chunks, G, chunks2, partials2);
    fin_in = partials2; fin_chunks = chunks2;
  }
  hipLaunchKernelGGL((k_msm_window_finish<F>), dim3(rs.W), dim3(64), 0, st, fin_in, rs, fin_chunks, wsums, rs.W == 1 ? out : (F*)nullptr);
  if (rs.W != 1) hipLaunchKernelGGL((k_msm_final<F>), dim3(1), dim3(64), 0, st, (const Xyzz<F>*)wsums, rs.W, out);
  ST_TRY(launch_check(ctx, "msm_reduce/final"));
  if (ctx->timing) {
    (void)hipEventRecord(ctx->ev[3], st);
    ctx->timing_pending = true;
  }
  return KEAKI_OK;
}

// one-time table build for the precomputed path
template <class F>
keaki_status msm_build_tables(keaki_hip_ctx* ctx, const Aff<F>* d_points, size_t N, int c_table, Aff<F>* d_table) {
  MsmPlan plan = msm_make_plan(N, c_table);
  bool done = false;
  if constexpr (std::is_same<F, Fq>::value) {
    if (plan.s.W <= TABLE_MAX_W) {
      hipLaunchKernelGGL(k_msm_build_tables_g1, dim3(cdiv(N, 64)), dim3(64), 0, ctx->stream, d_points, (u32)N, plan.s, d_table);
      done = true;
    }
  }
  if (!done)
  hipLaunchKernelGGL((k_msm_build_tables<F>), dim3(cdiv(N, 64)), dim3(64), 0, ctx->stream, d_points, (u32)N, plan.s, d_table);
  return launch_check(ctx, "msm_build_tables");
}
// window target for the shared-bucket (precomputed) path: adds = n * W(c); bucket reduction ~ 2.8 * max_b once
inline int choose_window_shared(size_t n, int forced = 0) {
  if (forced >= 3 && forced <= 24) return forced;         // Tuning::msm_c_shared
  double best = 1e300;
  int bc = 8;
  for (int c = 8; c <= 23; c++) {
    MsmPlan p = msm_make_plan(n, c);
    if (p.max_b > ((size_t)PART_MAX_BINS << PART_MAX_FINE_SHIFT)) continue;
    double cost = (double)n * p.s.W + 2.8 * (double)p.max_b;
    if ((double)p.max_b < 131072.0) cost *= 131072.0 / (double)p.max_b;
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}
inline u32 msm_plan_windows(size_t n, int c) { return msm_make_plan(n, c).s.W; }

}  // namespace keaki_internal
