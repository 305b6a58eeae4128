// Calibration of rocprofv3's FETCH_SIZE on the access shapes of this tree (MI355X_MICROARCH.md: "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a
// wide coalesced streaming read ... Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel below reads a KNOWN number of bytes, each byte exactly once, from a table far larger than the Infinity Cache (4 GiB against 256 MiB);
// run it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (the program directly after `--`) and divide: bench_tools/r6_fetch_calibration.sh does both.
//   stream16      : the control -- 16 B per lane, coalesced (the guide's calibrated shape: expect 0.5)
//   stream4       : 4 B per lane, coalesced
//   row64         : THE BUCKET KERNEL'S SHAPE -- every lane reads one 64-byte row (four 16-byte loads of its own row), rows in a scattered order (a
//                   multiplicative permutation of all rows: neighbouring lanes are megabytes apart; the other half of a row's 128-byte line is read by
//                   another wave at another time, as in the bucket kernel, where every table row is read exactly once per MSM in bucket order)
//   row64_nt      : the same with non-temporal loads (option acc_nt)
//   row64_coop    : four adjacent lanes read one row, 16 bytes each (the cooperative load of VERDICT r05 next-2 iii)
//   row128        : every lane reads BOTH halves of a 128-byte line (eight 16-byte loads)
//   row64_pairs   : every lane reads one 64-byte row, lanes 2k and 2k+1 the two halves of one 128-byte line
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench_tools/ubench_fetch_calib bench_tools/ubench_fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static constexpr u32 MULT = 0x9E3779B1u;          // odd: i -> i * MULT mod 2^k is a permutation of the 2^k rows

__global__ void __launch_bounds__(256) k_calib_stream16(const v4u* __restrict__ t, size_t n16, u32* __restrict__ out) {
  u32 acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const v4u v = t[i]; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_calib_stream4(const u32* __restrict__ t, size_t n4, u32* __restrict__ out) {
  u32 acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += t[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// every lane: rows r = perm(i) for its i's; NT: non-temporal loads
template <int NT>
__global__ void __launch_bounds__(256) k_calib_row64(const v4u* __restrict__ t, u32 log2rows, u32* __restrict__ out) {
  const u32 rows = 1u << log2rows;
  u32 acc = 0;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const u32 r = (i * MULT) & (rows - 1);
    const v4u* p = t + (size_t)r * 4;
    v4u a, b, c, d;
    if (NT) { a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 1); c = __builtin_nontemporal_load(p + 2); d = __builtin_nontemporal_load(p + 3); }
    else { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
    acc += a[0] ^ b[1] ^ c[2] ^ d[3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// four adjacent lanes share a row: lane q of the quad reads bytes [16q, 16q + 16)
template <int NT>
__global__ void __launch_bounds__(256) k_calib_row64_coop(const v4u* __restrict__ t, u32 log2rows, u32* __restrict__ out) {
  const u32 rows = 1u << log2rows;
  u32 acc = 0;
  const u32 q = threadIdx.x & 3;
  for (u32 i = (blockIdx.x * 256 + threadIdx.x) >> 2; i < rows; i += (gridDim.x * 256) >> 2) {
    const u32 r = (i * MULT) & (rows - 1);
    const v4u* p = t + (size_t)r * 4 + q;
    const v4u a = NT ? __builtin_nontemporal_load(p) : *p;
    acc += a[0] ^ a[3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_calib_row128(const v4u* __restrict__ t, u32 log2lines, u32* __restrict__ out) {
  const u32 lines = 1u << log2lines;
  u32 acc = 0;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < lines; i += gridDim.x * 256) {
    const u32 r = (i * MULT) & (lines - 1);
    const v4u* p = t + (size_t)r * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) { const v4u a = p[k]; acc += a[k & 3]; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// lanes 2k, 2k + 1: the two 64-byte halves of one scattered 128-byte line
__global__ void __launch_bounds__(256) k_calib_row64_pairs(const v4u* __restrict__ t, u32 log2rows, u32* __restrict__ out) {
  const u32 rows = 1u << log2rows, lines = rows >> 1;
  u32 acc = 0;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const u32 r = ((((i >> 1) * MULT) & (lines - 1)) << 1) | (i & 1);
    const v4u* p = t + (size_t)r * 4;
    const v4u a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a[0] ^ b[1] ^ c[2] ^ d[3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// ONE workgroup (one XCD's L2): first every EVEN 64-byte row of a small table (scattered order), then every ODD one. If a miss on a 64-byte row
// brings the whole 128-byte line into L2, the second phase hits (fabric requests = lines); if L2 fills 64-byte sectors, it misses again (= 2 x lines).
__global__ void __launch_bounds__(256) k_calib_halves(const v4u* __restrict__ t, u32 log2lines, u32* __restrict__ out) {
  const u32 lines = 1u << log2lines;
  u32 acc = 0;
  for (u32 half = 0; half < 2; half++) {
    for (u32 i = threadIdx.x; i < lines; i += 256) {
      const u32 r = (((i * MULT) & (lines - 1)) << 1) | half;
      const v4u* p = t + (size_t)r * 4;
      const v4u a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    __syncthreads();
  }
  out[threadIdx.x] = acc;
}

template <class F>
static void timed(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-22s requested_bytes %.0f  %.3f ms  %.2f TB/s\n", name, bytes, ms, bytes / ms * 1e-9);
}
int main(int argc, char** argv) {
  const u32 log2rows = argc > 1 ? (u32)atoi(argv[1]) : 26;              // 2^26 rows x 64 B = 4 GiB
  const size_t bytes = (size_t)64 << log2rows;
  // argv[2]: how the table is allocated -- 0 hipMalloc (what the library does), 1 hipDeviceMallocUncached, 2 hipDeviceMallocFinegrained: would a
  // non-cached memory type make the fabric fetch 64 bytes for a 64-byte row? (answer in profiles/r06_fetch_size_calibration.txt)
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  v4u* t; u32* out;
  if (mode == 0) CK(hipMalloc(&t, bytes));
  else CK(hipExtMallocWithFlags((void**)&t, bytes, mode == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
  printf("# allocation mode %d\n", mode);
  CK(hipMalloc(&out, 4096 * 256 * 4));
  CK(hipMemset(t, 0x5A, bytes)); CK(hipDeviceSynchronize());
  const dim3 g(4096), b(256);
  printf("# table %zu MiB, every kernel reads every byte of it exactly once\n", bytes >> 20);
  timed("stream16", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_stream16, g, b, 0, 0, t, bytes / 16, out); });
  timed("stream4", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_stream4, g, b, 0, 0, (const u32*)t, bytes / 4, out); });
  timed("row64", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_row64<0>, g, b, 0, 0, t, log2rows, out); });
  timed("row64_nt", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_row64<1>, g, b, 0, 0, t, log2rows, out); });
  timed("row64_coop", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_row64_coop<0>, g, b, 0, 0, t, log2rows, out); });
  timed("row64_coop_nt", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_row64_coop<1>, g, b, 0, 0, t, log2rows, out); });
  timed("row128", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_row128, g, b, 0, 0, t, log2rows - 1, out); });
  timed("row64_pairs", (double)bytes, [&] { hipLaunchKernelGGL(k_calib_row64_pairs, g, b, 0, 0, t, log2rows, out); });
  // 2 MiB of the table (16,384 lines of 128 B), one workgroup: requested = 2 MiB
  timed("halves_2MiB_1wg", 2097152.0, [&] { hipLaunchKernelGGL(k_calib_halves, dim3(1), b, 0, 0, t + ((size_t)1 << 24), 14u, out); });
  CK(hipDeviceSynchronize());
  return 0;
}
