// The shader clock under the load of the field arithmetic. Ground truth: a one-wave PROBE kernel on a second stream that counts s_memtime ticks
// (shader clock) over a window of s_memrealtime (constant 100 MHz) while the test kernel runs. (A first version divided the s_memtime span of a
// workgroup by the wall time of the launch, assuming every workgroup is resident for the whole launch; the dispatcher does not spread
// 256 x N workgroups N per CU, so that reading came out too low -- the column is kept for comparison.)
// Every "SIMD-cycle" figure in profiles/ and DESIGN.md is ms x 2.4 GHz (the peak clock); this says what the clock really is.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I keaki_amd/csrc -o bench_tools/ubench_clock bench_tools/ubench_clock.hip
#include "fq29.hip.h"
#include <stdio.h>
using namespace bn254;
#include <unistd.h>
__global__ void k_probe(unsigned long long* out, unsigned long long us) {
  if (threadIdx.x) return;
  __builtin_amdgcn_s_setprio(3);                      // the youngest wave of a saturated SIMD would otherwise wait between its two clock reads
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
  unsigned long long r1 = r0;
  while (r1 - r0 < us * 100) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
  out[0] = __builtin_readcyclecounter() - t0; out[1] = r1 - r0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned long long* clk, u32* out, int iters) {
  U29 a, b;
  for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i) & Q29::MASK; b.l[i] = (blockIdx.x * 40503u + 7 * i) & Q29::MASK; }
  a.l[8] &= 0xFFFFF; b.l[8] &= 0xFFFFF;
  u32 x = a.l[0];
  U29 Y1 = b, ZZ = u29_one(), ZZZ = u29_one(), Y2 = a;
  const unsigned long long t0 = __builtin_readcyclecounter();         // s_memtime
  for (int it = 0; it < iters; it++) {
    if (OP == 0) a = u29_mul(a, b);                                       // the product stream (162 multiply-adds of 205)
    else if (OP == 2) {                                                   // the bucket kernel's mixed addition (8M + 2S, one dual product), operands synthetic
      b.l[0] = (b.l[0] + 1u) & Q29::MASK;
      const U29 U2 = u29_mul(b, ZZ), S2 = u29_mul(Y2, ZZZ);
      const U29 P = u29_sub(U2, a, Q29::K16), R = u29_sub(S2, Y1, Q29::K4);
      const U29 PP = u29_sqr(P), PPP = u29_mul(P, PP), Q = u29_mul(a, PP);
      const U29 X3 = u29_sub3(u29_sqr(R), PPP, Q);
      const U29 T = u29_sub(Q, X3, Q29::K16);
      U29 NY1;
#pragma unroll
      for (int i = 0; i < 9; i++) NY1.l[i] = Q29::K2[i] - Y1.l[i];
      Y1 = u29_mul2(R, T, NY1, PPP);
      a = u29_carry(X3);
      ZZ = u29_mul(ZZ, PP);
      ZZZ = u29_mul(ZZZ, PPP);
    } else if (OP == 3) {                                                 // k_pairing's mix: 60 % v_mad_u64_u32 -- one product stream (162 of 214) + 56 plain
      a = u29_mul(a, b);                                                  // instructions on eight independent registers (162 / 270 = 0.60; counts from the emitted loop)
#pragma unroll
      for (int u = 0; u < 56; u++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(Y1.l[u & 7]) : "v"(b.l[1]));
    } else {
#pragma unroll
      for (int u = 0; u < 64; u++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(b.l[0]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
  u32 r = x;
  for (int i = 0; i < 9; i++) r ^= a.l[i] ^ Y1.l[i] ^ ZZ.l[i] ^ ZZZ.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static hipStream_t g_s2 = nullptr;
static unsigned long long* g_probe = nullptr;
template <int OP>
int run(const char* name, int waves, int iters) {
  // ONE workgroup of 256 x waves threads per CU: `waves` waves on every SIMD
  const int blocks = 256, threads = 256 * waves;
  unsigned long long* clk; u32* out;
  CK(hipMalloc(&clk, blocks * 8)); CK(hipMalloc(&out, (size_t)blocks * threads * 4));
  if (!g_s2) { CK(hipStreamCreateWithFlags(&g_s2, hipStreamNonBlocking)); CK(hipMalloc(&g_probe, 16)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, clk, out, 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, clk, out, iters);
  CK(hipEventRecord(e1));
  usleep(3000);                                                        // well inside the launch (every launch here runs for >= 15 ms)
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, g_s2, g_probe, 5000ull);
  CK(hipStreamSynchronize(g_s2));
  const bool still_running = hipEventQuery(e1) == hipErrorNotReady;
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long hp[2]; CK(hipMemcpy(hp, g_probe, 16, hipMemcpyDeviceToHost));
  unsigned long long* h = new unsigned long long[blocks];
  CK(hipMemcpy(h, clk, blocks * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (int i = 0; i < blocks; i++) mean += (double)h[i];
  mean /= blocks;
  const double units = (double)waves * iters;                          // wave-units per SIMD
  printf("%-42s waves/SIMD=%d  launch %8.3f ms  probe %5.0f MHz%s  (workgroup ticks / launch time: %4.0f MHz)  %7.1f ns per wave-unit and SIMD\n", name, waves, ms,
         (double)hp[0] / ((double)hp[1] / 100.0), still_running ? "" : " [probe finished after the launch: ignore]", mean / (ms * 1e-3) / 1e6, ms * 1e6 / units);
  (void)mean;
  delete[] h; CK(hipFree(clk)); CK(hipFree(out));
  return 0;
}
int main() {
  for (int w : {1, 2, 3, 4}) { run<0>("u29_mul chain (205 instr, 162 mads)", w, 60000); run<1>("dependent v_and_b32 chain (64 per unit)", w, 100000); }
  for (int w : {1, 2, 3}) run<2>("XYZZ mixed addition (2093 VALU, 1467 mads)", w, 8000);
  for (int w : {1, 2, 3}) run<3>("pairing mix (270 VALU, 162 mads = 60 %)", w, 45000);
  return 0;
}
