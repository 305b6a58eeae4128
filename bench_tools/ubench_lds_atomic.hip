// Micro-benchmark: cost of LDS atomics with random addresses (the bucket sort's counting passes), gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_lds_atomic ubench_lds_atomic.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
__device__ __forceinline__ u32 rnd(u32& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>   // 0: returning add, 1: add without return, 2: plain store, 3: plain load, 4: returning add on keys with runs of equal neighbours
__global__ void __launch_bounds__(1024) k(u32* out, u32 nbins, u32 iters) {
  __shared__ u32 h[8192];
  for (u32 i = threadIdx.x; i < 8192; i += 1024) h[i] = 0;
  __syncthreads();
  u32 s = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
  for (u32 it = 0; it < iters; it++) {
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
      u32 a = rnd(s) & (nbins - 1);
      if (MODE == 0) acc += atomicAdd(&h[a], 1u);
      else if (MODE == 1) atomicAdd(&h[a], 1u);
      else if (MODE == 2) h[a] = s;
      else if (MODE == 3) acc += h[a];
      else { a = (a & ~7u) | ((threadIdx.x >> 3) & 7u); acc += atomicAdd(&h[a], 1u); }
    }
  }
  __syncthreads();
  out[blockIdx.x * 1024 + threadIdx.x] = acc + h[threadIdx.x];
}
template <int MODE>
void run(const char* name, u32 nbins, u32 threads) {
  u32* d; (void)hipMalloc(&d, 256 * 1024 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const u32 iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, nbins, 10u);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, nbins, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double waveops = (double)(threads / 64) * iters * 8;     // per CU
  printf("%-28s bins=%5u waves/CU=%2u  %.3f ms  %.1f cycles(2.4GHz) per wave-op per CU\n", name, nbins, threads / 64, ms, ms * 1e-3 * 2.4e9 / waveops);
  (void)hipFree(d);
}
int main() {
  for (u32 nb : {64u, 1024u, 2048u, 8192u}) {
    run<0>("ds_add_rtn random", nb, 1024);
    run<1>("ds_add (no return) random", nb, 1024);
    run<2>("ds_write_b32 random", nb, 1024);
    run<3>("ds_read_b32 random", nb, 1024);
  }
  run<0>("ds_add_rtn random", 1024, 256);
  run<1>("ds_add (no return) random", 1024, 256);
  run<4>("ds_add_rtn 8 lanes/line", 1024, 1024);
  return 0;
}
