// Stand-alone timing harness of the MSM bucket sort (k_tile_sort / k_cell_prefix / k_bin_scan / k_chunk_sort) on synthetic scalars:
// per-kernel HIP-event times and, built with -DKEAKI_STAMP, in-kernel phase stamps of one workgroup of k_chunk_sort / k_tile_sort.
// Build (repo root): hipcc -O3 -std=c++17 --offload-arch=gfx950 -DKEAKI_STAMP -Ikeaki_amd/csrc -o bench_tools/dbg/sort_harness bench_tools/dbg/sort_harness.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifdef KEAKI_STAMP
__device__ unsigned long long g_stamp[4096];
#define STAMP(cond, i) do { if (cond) g_stamp[(i)] = __builtin_readcyclecounter(); } while (0)
#endif
#include "msm.hip.h"
using namespace bn254;
__global__ void k_fill(u32* p, size_t nwords) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  u32 v = (u32)z;
  if ((i & 7) == 7) v &= 0x0FFFFFFFu;      // < 2^252
  p[i] = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int log2n = argc > 1 ? atoi(argv[1]) : 24;
  const size_t n = (size_t)1 << log2n;
  MsmPlan plan = msm_make_plan(n, 22);
  MsmShape s = plan.s; s.stride = (u32)n;
  size_t nb = plan.max_b;
  PartShape ps;
  if (!part_make_shape(n, s.W, nb, &ps, argc > 2 ? atoi(argv[2]) : -1, argc > 3 ? atoi(argv[3]) : -1, argc > 4 ? atoi(argv[4]) : -1)) { printf("shape\n"); return 1; }
  printf("n=2^%d W=%u c=%u nbins=%u shift=%u hb=%u nf=%u tile=%u ntiles=%u te=%u geom=%u\n", log2n, s.W, s.c, ps.nbins, ps.shift, ps.hb, ps.nf, ps.tile, ps.ntiles, ps.te, ps.geom);
  const size_t pairs = n * s.W;
  u32 *scal, *tiles, *sorted, *counts, *segoff, *bin_total; u16* tstart; uint2* cellmeta; BinMeta* bins; v4u_t* segtab;
  CK(hipMalloc(&scal, n * 32)); CK(hipMalloc(&tiles, (size_t)ps.ntiles * ps.te * 4)); CK(hipMalloc(&sorted, pairs * 4)); CK(hipMalloc(&counts, nb * 4));
  CK(hipMalloc(&segoff, part_max_chunks(pairs, ps.nbins) * ps.nf * 4)); CK(hipMalloc(&segtab, nb * 16)); CK(hipMalloc(&bin_total, ps.nbins * 4));
  CK(hipMalloc(&tstart, (size_t)ps.ntiles * (ps.nbins + 1) * 2)); CK(hipMalloc(&cellmeta, (size_t)ps.nbins * ps.ntiles * 8)); CK(hipMalloc(&bins, ps.nbins * sizeof(BinMeta)));
  hipLaunchKernelGGL(k_fill, dim3((n * 8 + 255) / 256), dim3(256), 0, 0, scal, n * 8);
  CK(hipDeviceSynchronize());
  hipEvent_t ev[6]; for (auto& e : ev) CK(hipEventCreate(&e));
  int ncu = 256;
  for (int it = 0; it < 4; it++) {
    CK(hipEventRecord(ev[0]));
    hipLaunchKernelGGL(k_tile_sort<12>, dim3(ps.ntiles < (u32)ncu ? ps.ntiles : ncu), dim3(T1_THREADS), 0, 0, (const Fr*)scal, s, ps, tiles, tstart);
    CK(hipEventRecord(ev[1]));
    hipLaunchKernelGGL(k_cell_prefix, dim3(ps.nbins), dim3(1024), 0, 0, (const u16*)tstart, ps, cellmeta, bin_total);
    CK(hipEventRecord(ev[2]));
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, 0, (const u32*)bin_total, ps.nbins, bins);
    CK(hipEventRecord(ev[3]));
#define CS(L, R, Q) hipLaunchKernelGGL((k_chunk_sort<L, R, Q>), dim3(ps.nbins), dim3(C2_THREADS), 0, 0, (const u32*)tiles, (const uint2*)cellmeta, (const BinMeta*)bins, s, ps, (u32)nb, sorted, segtab, segoff, counts)
    switch (ps.geom) { case 0: CS(8, 1, 16); break; case 1: CS(16, 1, 16); break; case 2: CS(16, 2, 8); break; default: CS(16, 4, 4); break; }
    CK(hipEventRecord(ev[4]));
    CK(hipDeviceSynchronize());
    float a, b, c, d;
    CK(hipEventElapsedTime(&a, ev[0], ev[1])); CK(hipEventElapsedTime(&b, ev[1], ev[2])); CK(hipEventElapsedTime(&c, ev[2], ev[3])); CK(hipEventElapsedTime(&d, ev[3], ev[4]));
    printf("tile_sort %.3f  cell_prefix %.3f  bin_scan %.3f  chunk_sort %.3f ms\n", a, b, c, d);
  }
  // sanity: totals
  std::vector<u32> hc(nb); CK(hipMemcpy(hc.data(), counts, nb * 4, hipMemcpyDeviceToHost));
  unsigned long long tot = 0; for (u32 v : hc) tot += v;
  printf("sum of bucket counts %llu (pairs %zu minus zero digits)\n", tot, pairs);
#ifdef KEAKI_STAMP
  std::vector<unsigned long long> st(4096);
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), 4096 * 8));
  printf("chunk_sort stamps (one workgroup), cycles between marks per chunk:\n");
  for (int k = 0; k < 8; k++) {
    printf(" chunk %d:", k);
    for (int i = 1; i < 12; i++) { long long d = (long long)(st[k * 16 + i] - st[k * 16 + i - 1]); printf(" %lld", st[k * 16 + i] ? d : -1); }
    printf("  | slow=%llu odd lanes=%llu last odd grp/c0=%llx\n", st[k * 16 + 12], st[k * 16 + 13], st[k * 16 + 14]);
  }
  printf("tile_sort stamps (one workgroup, first tiles):\n");
  for (int k = 0; k < 4; k++) {
    printf(" tile %d:", k);
    for (int i = 1; i < 10; i++) { long long d = (long long)(st[2048 + k * 16 + i] - st[2048 + k * 16 + i - 1]); printf(" %lld", st[2048 + k * 16 + i] ? d : -1); }
    printf("\n");
  }
#endif
  return 0;
}
