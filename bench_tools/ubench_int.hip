// Integer-ALU micro-benchmark for gfx950: issue rate of the instructions a 256-bit Montgomery
// multiplier is made of. Prints wave-instructions per cycle per SIMD-equivalent figures so the
// "integer roofline" in DESIGN.md is measured, not guessed.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_int ubench_int.hip && ./ubench_int
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k_ops(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7;
  uint64_t acc0 = a, acc1 = b, acc2 = a ^ b, acc3 = a + b, acc4 = 1, acc5 = 2, acc6 = 3, acc7 = 4;
  uint32_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = 5, x5 = 6, x6 = 7, x7 = 8;
  double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (OP == 0) {  // v_mad_u64_u32, 8 independent chains
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7) : "v"(a), "v"(b) : "vcc");
      } else if (OP == 1) {  // v_add_co_u32 (full rate reference)
        asm volatile("v_add_co_u32 %0, vcc, %8, %0\n v_add_co_u32 %1, vcc, %8, %1\n v_add_co_u32 %2, vcc, %8, %2\n v_add_co_u32 %3, vcc, %8, %3\n"
                     "v_add_co_u32 %4, vcc, %8, %4\n v_add_co_u32 %5, vcc, %8, %5\n v_add_co_u32 %6, vcc, %8, %6\n v_add_co_u32 %7, vcc, %8, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
      } else if (OP == 2) {  // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n"
                     "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
      } else if (OP == 3) {  // v_mul_hi_u32
        asm volatile("v_mul_hi_u32 %0, %8, %0\n v_mul_hi_u32 %1, %8, %1\n v_mul_hi_u32 %2, %8, %2\n v_mul_hi_u32 %3, %8, %3\n"
                     "v_mul_hi_u32 %4, %8, %4\n v_mul_hi_u32 %5, %8, %5\n v_mul_hi_u32 %6, %8, %6\n v_mul_hi_u32 %7, %8, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
      } else if (OP == 4) {  // v_lshl_add_u64 (64-bit add)
        asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                     "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                     : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7) : "v"(acc0 | 1));
      } else if (OP == 5) {  // v_fma_f64
        asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                     "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1.0000001), "v"(0.5));
      } else if (OP == 6) {  // v_mad_u32_u24
        asm volatile("v_mad_u32_u24 %0, %8, %0, %0\n v_mad_u32_u24 %1, %8, %1, %1\n v_mad_u32_u24 %2, %8, %2, %2\n v_mad_u32_u24 %3, %8, %3, %3\n"
                     "v_mad_u32_u24 %4, %8, %4, %4\n v_mad_u32_u24 %5, %8, %5, %5\n v_mad_u32_u24 %6, %8, %6, %6\n v_mad_u32_u24 %7, %8, %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
      } else if (OP == 7) {  // v_addc_co_u32 chain (carry in + out)
        asm volatile("v_addc_co_u32 %0, vcc, %8, %0, vcc\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n"
                     "v_addc_co_u32 %4, vcc, %8, %4, vcc\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n v_addc_co_u32 %6, vcc, %8, %6, vcc\n v_addc_co_u32 %7, vcc, %8, %7, vcc\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
      } else if (OP == 8) {  // v_mov_b32
        asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 9) {  // v_mul_u32_u24 + v_mul_hi_u32_u24
        asm volatile("v_mul_u32_u24 %0, %8, %0\n v_mul_hi_u32_u24 %1, %8, %1\n v_mul_u32_u24 %2, %8, %2\n v_mul_hi_u32_u24 %3, %8, %3\n"
                     "v_mul_u32_u24 %4, %8, %4\n v_mul_hi_u32_u24 %5, %8, %5\n v_mul_u32_u24 %6, %8, %6\n v_mul_hi_u32_u24 %7, %8, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
      }
    }
  }
  uint64_t r = acc0 ^ acc1 ^ acc2 ^ acc3 ^ acc4 ^ acc5 ^ acc6 ^ acc7 ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  r ^= (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

template <int OP>
int run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_ops<OP>, dim3(blocks), dim3(256), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_ops<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double winstr = (double)blocks * 4 * iters * 16 * 8;  // wave-instructions
  double per_simd_per_s = winstr / 1024.0 / (ms * 1e-3);
  printf("%-22s waves/SIMD=%d  %8.3f ms  %7.2f G wave-instr/s/SIMD  => %.2f cycles/wave-instr @2.4GHz\n", name, waves_per_simd, ms,
         per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s);
  CK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_mad_u64_u32", w); run<2>("v_mul_lo_u32", w); run<3>("v_mul_hi_u32", w); run<1>("v_add_co_u32", w);
    run<7>("v_addc_co_u32", w); run<4>("v_lshl_add_u64", w); run<8>("v_mov_b32", w); run<5>("v_fma_f64", w);
    run<6>("v_mad_u32_u24", w); run<9>("v_mul(_hi)_u32_u24", w);
  }
  return 0;
}
