// A one-wave probe kernel that measures the shader clock WHILE other kernels run: it spins for `us` microseconds of the constant 100 MHz
// clock (s_memrealtime) and returns the s_memtime ticks that passed. Launched on a stream of its own next to the kernel under study.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o bench_tools/libclock_probe.so bench_tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_probe(uint64_t* out, uint64_t us) {
  if (threadIdx.x) return;
  __builtin_amdgcn_s_setprio(3);                      // the youngest wave of a saturated SIMD would otherwise wait between its two clock reads
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
  uint64_t r1 = r0;
  while (r1 - r0 < us * 100) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[0] = t1 - t0; out[1] = r1 - r0;
}
static hipStream_t g_stream = nullptr;
static uint64_t* g_out = nullptr;
extern "C" int probe_start(uint64_t us) {
  if (!g_stream) { if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1; if (hipMalloc(&g_out, 16) != hipSuccess) return -2; }
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, g_stream, g_out, us);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// shader MHz averaged over the probe's window
extern "C" double probe_wait() {
  uint64_t h[2];
  if (hipStreamSynchronize(g_stream) != hipSuccess) return -1;
  if (hipMemcpy(h, g_out, 16, hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return (double)h[0] / ((double)h[1] / 100.0);
}
