// Host-side driver of the MSM pipeline (workspace sizing, window choice, kernel sequence), written
// once and instantiated for G1 (msm_g1.hip) and G2 (msm_g2.hip).
#pragma once
#include "internal.h"
#include "msm.cuh"

namespace keaki_internal {
using namespace bn254;

// GPU window choice: minimise (n * W mixed adds) + (2 * W * B full adds, ~1.4x a mixed add each)
// while keeping enough buckets (= lanes of the accumulate kernel) to fill 256 CUs.
inline int choose_window(size_t n) {
  if (const char* e = getenv("KEAKI_MSM_C")) {
    int c = atoi(e);
    if (c >= 3 && c <= 24 && 254 % c != 0) return c;
  }
  if (n < 32) return 3;
  double best = 1e300;
  int bc = 3;
  for (int c = 3; c <= 22; c++) {
    if (254 % c == 0) continue;
    double W = (254 + c - 1) / c, B = (double)(1u << (c - 1));
    double cost = (double)n * W + 2.8 * W * B;
    double lanes = W * B;
    if (lanes < 131072.0) cost *= 131072.0 / lanes;
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

// exclusive scan of `len` u32 counters in[] -> out[] (global positions)
inline keaki_status device_scan(keaki_hip_ctx* ctx, const u32* in, u32 len, u32* out) {
  u32 nblocks = cdiv(len, SCAN_ELEMS);
  ST_TRY(reserve(ctx, ctx->bsums, (size_t)nblocks * 4));
  u32* bs = (u32*)ctx->bsums.p;
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, in, len, bs);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, bs, nblocks, 0u);
  hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, in, len, (const u32*)bs, out);
  return launch_check(ctx, "scan");
}

template <class F>
keaki_status msm_dev(keaki_hip_ctx* ctx, const Aff<F>* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac) {
  if (!d_out_jac || (n && (!d_points || !d_scalars))) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: null pointer");
  if (n > srs_len) return fail(ctx, KEAKI_ERR_TOO_LARGE, "msm: %zu scalars but the SRS holds %zu points", n, srs_len);
  if (n >= (1ull << 31)) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: n must be < 2^31 per device");
  MsmShape s;
  s.n = (u32)n;
  s.c = (u32)choose_window(n);
  s.W = (254 + s.c - 1) / s.c;
  s.B = 1u << (s.c - 1);
  ctx->last_c = (int)s.c;
  const size_t nb = (size_t)s.W * s.B;
  if ((double)n * s.W >= 4294967295.0) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm: n * windows overflows 32-bit positions");
  const u32 L = s.B >= 4096 ? 64 : (s.B >= 64 ? 16 : s.B);  // reduce chunk length
  const u32 chunks = cdiv(s.B, L);
  ST_TRY(reserve(ctx, ctx->wsums, (size_t)s.W * sizeof(Xyzz<F>)));
  Xyzz<F>* wsums = (Xyzz<F>*)ctx->wsums.p;
  F* out = (F*)d_out_jac;
  hipStream_t st = ctx->stream;
  if (ctx->timing) (void)hipEventRecord(ctx->ev[0], st);
  if (n == 0) {
    hipLaunchKernelGGL((k_msm_final<F>), dim3(1), dim3(64), 0, st, (const Xyzz<F>*)wsums, 0u, out);
    return launch_check(ctx, "msm_final");
  }
  ST_TRY(reserve(ctx, ctx->digits, n * s.W * 4));
  ST_TRY(reserve(ctx, ctx->sorted, n * s.W * 4));
  ST_TRY(reserve(ctx, ctx->hist, nb * 4));
  ST_TRY(reserve(ctx, ctx->offsets, nb * 4));
  ST_TRY(reserve(ctx, ctx->cursor, nb * 4));
  ST_TRY(reserve(ctx, ctx->buckets, nb * sizeof(Xyzz<F>)));
  ST_TRY(reserve(ctx, ctx->partials, (size_t)s.W * chunks * sizeof(Xyzz<F>)));
  u32 *digits = (u32*)ctx->digits.p, *sorted = (u32*)ctx->sorted.p, *hist = (u32*)ctx->hist.p, *offsets = (u32*)ctx->offsets.p,
      *cursor = (u32*)ctx->cursor.p;
  Xyzz<F>* buckets = (Xyzz<F>*)ctx->buckets.p;
  Xyzz<F>* partials = (Xyzz<F>*)ctx->partials.p;
  HIP_TRY(ctx, hipMemsetAsync(hist, 0, nb * 4, st));
  HIP_TRY(ctx, hipMemsetAsync(cursor, 0, nb * 4, st));
  hipLaunchKernelGGL(k_msm_digits, dim3(cdiv(n, 256)), dim3(256), 0, st, (const Fr*)d_scalars, s, digits, hist);
  ST_TRY(launch_check(ctx, "msm_digits"));
  ST_TRY(device_scan(ctx, hist, (u32)nb, offsets));
  hipLaunchKernelGGL(k_msm_scatter, dim3(cdiv(n, 256), s.W), dim3(256), 0, st, (const u32*)digits, s, (const u32*)offsets, cursor, sorted);
  ST_TRY(launch_check(ctx, "msm_scatter"));
  if (ctx->timing) (void)hipEventRecord(ctx->ev[1], st);
  hipLaunchKernelGGL((k_msm_accumulate<F>), dim3(cdiv(nb, 256)), dim3(256), 0, st, d_points, (const u32*)sorted, (const u32*)offsets,
                     (const u32*)hist, (u32)nb, buckets);
  ST_TRY(launch_check(ctx, "msm_accumulate"));
  if (ctx->timing) (void)hipEventRecord(ctx->ev[2], st);
  hipLaunchKernelGGL((k_msm_reduce<F>), dim3(cdiv((size_t)s.W * chunks, 64)), dim3(64), 0, st, (const Xyzz<F>*)buckets, s, L, chunks, partials);
  hipLaunchKernelGGL((k_msm_window_finish<F>), dim3(s.W), dim3(64), 0, st, (const Xyzz<F>*)partials, s, chunks, wsums);
  hipLaunchKernelGGL((k_msm_final<F>), dim3(1), dim3(64), 0, st, (const Xyzz<F>*)wsums, s.W, out);
  ST_TRY(launch_check(ctx, "msm_reduce/final"));
  if (ctx->timing) {
    (void)hipEventRecord(ctx->ev[3], st);
    ctx->timing_pending = true;
  }
  return KEAKI_OK;
}

}  // namespace keaki_internal
