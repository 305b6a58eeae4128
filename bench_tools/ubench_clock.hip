// The shader clock under the load of the field arithmetic: s_memtime (shader clock counter) against the wall time of the launch
// (HIP events). Every "SIMD-cycle" figure in profiles/ and DESIGN.md is ms x 2.4 GHz (the peak clock); this says what the clock really is
// while the product streams run on every SIMD, so that those figures can be read in real cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I keaki_amd/csrc -o bench_tools/ubench_clock bench_tools/ubench_clock.hip
#include "fq29.hip.h"
#include <stdio.h>
using namespace bn254;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned long long* clk, u32* out, int iters) {
  U29 a, b;
  for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i) & Q29::MASK; b.l[i] = (blockIdx.x * 40503u + 7 * i) & Q29::MASK; }
  a.l[8] &= 0xFFFFF; b.l[8] &= 0xFFFFF;
  u32 x = a.l[0];
  const unsigned long long t0 = __builtin_readcyclecounter();         // s_memtime
  for (int it = 0; it < iters; it++) {
    if (OP == 0) a = u29_mul(a, b);                                       // the product stream (162 multiply-adds of 205)
    else {
#pragma unroll
      for (int u = 0; u < 64; u++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(b.l[0]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
  u32 r = x;
  for (int i = 0; i < 9; i++) r ^= a.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
int run(const char* name, int waves, int iters) {
  const int blocks = 256 * waves;
  unsigned long long* clk; u32* out;
  CK(hipMalloc(&clk, blocks * 8)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, clk, out, 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, clk, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* h = new unsigned long long[blocks];
  CK(hipMemcpy(h, clk, blocks * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (int i = 0; i < blocks; i++) mean += (double)h[i];
  mean /= blocks;
  // a workgroup's span is (nearly) the whole launch when all of them are resident at once: blocks = 256 CUs x waves
  printf("%-44s waves/SIMD=%d  launch %8.3f ms  s_memtime ticks per workgroup %.4g  -> %.0f MHz if s_memtime counts shader cycles (100 MHz would be the constant clock)\n",
         name, waves, ms, mean, mean / (ms * 1e-3) / 1e6);
  delete[] h; CK(hipFree(clk)); CK(hipFree(out));
  return 0;
}
int main() {
  for (int w : {1, 2, 3, 4}) { run<0>("u29_mul chain on every SIMD", w, 40000); run<1>("dependent v_and_b32 chain", w, 40000); }
  return 0;
}
