// Micro-benchmark of the fq29 streams on gfx950: cycles per Montgomery product / square at 1..4 waves per SIMD, against
// a dependent v_mad_u64_u32 chain, v_lshrrev_b64 and the saturated fq_mul_asm. Prices the "issue ceiling" of DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I keaki_amd/csrc -o bench_tools/ubench_u29 bench_tools/ubench_u29.hip
#include "fq29.hip.h"
#include <stdio.h>
using namespace bn254;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(u32* out, int iters) {
  U29 a, b, c, d;
  for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i) & Q29::MASK; b.l[i] = (blockIdx.x * 40503u + 7 * i) & Q29::MASK; }
  a.l[8] &= 0xFFFFF; b.l[8] &= 0xFFFFF;
  c = b; d = a;
  Fq fa, fb;
  for (int i = 0; i < 8; i++) { fa.l[i] = a.l[i]; fb.l[i] = b.l[i]; }
  u64 acc = a.l[0], acc2 = b.l[0];
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { a = u29_mul(a, b); }                                  // one dependent product chain
    else if (OP == 1) { a = u29_sqr(a); }
    else if (OP == 2) { a = u29_mul(a, b); c = u29_mul(c, d); }          // two independent chains (compiler may not interleave asm blocks)
    else if (OP == 3) { fa = fa * fb; }                                  // saturated 8 x 32 product
    else if (OP == 4) {                                                  // 64 dependent v_mad_u64_u32
#pragma unroll
      for (int u = 0; u < 64; u++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[0]), "v"(b.l[0]) : "vcc");
    } else if (OP == 5) {                                                // 2 x 32 interleaved independent chains
#pragma unroll
      for (int u = 0; u < 32; u++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(acc), "+v"(acc2) : "v"(a.l[0]), "v"(b.l[0]) : "vcc");
    } else if (OP == 6) {                                                // 64 dependent 64-bit shifts
#pragma unroll
      for (int u = 0; u < 64; u++) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc));
    } else if (OP == 7) {                                                // 64 dependent v_mul_lo_u32
#pragma unroll
      for (int u = 0; u < 64; u++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a.l[0]) : "v"(b.l[0]));
    } else if (OP == 8) {                                                // 64 dependent v_and
#pragma unroll
      for (int u = 0; u < 64; u++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a.l[0]) : "v"(b.l[0]));
    }
  }
  u32 r = (u32)acc ^ (u32)(acc >> 32) ^ (u32)acc2;
  for (int i = 0; i < 9; i++) r ^= a.l[i] ^ c.l[i];
  for (int i = 0; i < 8; i++) r ^= fa.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
int run(const char* name, int waves, double units_per_iter) {
  int blocks = 256 * waves;
  u32* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double cycles_per_unit_per_simd = ms * 1e-3 * 2.4e9 / ((double)waves * iters * units_per_iter);   // SIMD cycles per unit of one wave
  printf("%-34s waves/SIMD=%d  %8.3f ms  %9.1f SIMD-cycles per unit (per wave-unit, all waves interleaved)\n", name, waves, ms, cycles_per_unit_per_simd);
  CK(hipFree(out));
  return 0;
}
int main() {
  for (int w : {1, 2, 3, 4}) {
    run<0>("u29_mul (205 instr)", w, 1); run<1>("u29_sqr (177 instr)", w, 1); run<2>("2 independent u29_mul", w, 2);
    run<3>("fq_mul_asm saturated", w, 1);
    run<4>("v_mad_u64_u32 dependent", w, 64); run<5>("v_mad_u64_u32 2 chains", w, 64);
    run<6>("v_lshrrev_b64 dependent", w, 64); run<7>("v_mul_lo_u32 dependent", w, 64); run<8>("v_and_b32 dependent", w, 64);
  }
  return 0;
}
