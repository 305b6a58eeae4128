// Micro-benchmark: rate of the bucket sort's second-pass gather -- workgroup b reads "cell" b of every tile image (cells of S bytes at a
// stride of one image), 16 or 8 lanes per cell, 4- or 16-byte loads, Q cells in flight per group -- against the cell size. gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
// W: bytes per load (4 / 16); L: lanes per cell; each lane issues R loads per cell; Q cells per group in flight
template <int W, int L, int R, int Q>
__global__ void __launch_bounds__(1024) k_gather(const u32* __restrict__ buf, u32 ntiles, u32 te /* words per image */, u32 cellw /* words per cell */, u32 misalign,
                                                 u32* __restrict__ out) {
  const u32 b = blockIdx.x, t = threadIdx.x, grp = t / L, ll = t % L, G = 1024 / L;
  u32 acc = 0;
  for (u32 c0 = 0; c0 < ntiles; c0 += G * Q) {
    u32 e[Q][R][W / 4];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const u32 c = c0 + grp + q * G;
      const size_t base = (size_t)(c < ntiles ? c : 0) * te + (size_t)b * cellw + misalign;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const u32 j = (ll + r * L) * (W / 4);
        if (W == 4) {
          e[q][r][0] = j < cellw ? buf[base + j] : 0u;
        } else {
          const v4u v = j < cellw ? *reinterpret_cast<const v4u*>(buf + base + j) : v4u{0, 0, 0, 0};
#pragma unroll
          for (int x = 0; x < W / 4; x++) e[q][r][x] = v[x];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++)
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int x = 0; x < W / 4; x++) acc += e[q][r][x];
  }
  out[b * 1024 + t] = acc;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int W, int L, int R, int Q>
void run(const char* name, const u32* buf, u32 ntiles, u32 te, u32 nbins, u32 misalign, u32* out) {
  const u32 cellw = te / nbins;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 3; it++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<W, L, R, Q>), dim3(nbins), dim3(1024), 0, 0, buf, ntiles, te, cellw, misalign, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double bytes = (double)ntiles * te * 4;
  printf("%-34s cell %5u B  misalign %u  %.3f ms  %.2f TB/s useful\n", name, cellw * 4, misalign * 4, best, bytes / best * 1e-9);
}
int main() {
  const u32 ntiles = 5462, te = 36864;
  u32 *buf, *out;
  CK(hipMalloc(&buf, (size_t)ntiles * te * 4 + 4096)); CK(hipMalloc(&out, 4096 * 1024 * 4));
  CK(hipMemset(buf, 1, (size_t)ntiles * te * 4 + 4096));
  for (u32 nbins : {2048u, 1024u, 512u, 256u, 128u}) {
    for (u32 mis : {0u, 1u}) {
      run<4, 16, 4, 16>("dword, 16 lanes x4, 16 cells", buf, ntiles, te, nbins, mis, out);   // up to 64 words per cell... larger cells truncated: rate is per useful subset
      run<16, 16, 1, 16>("dwordx4, 16 lanes x1, 16 cells", buf, ntiles, te, nbins, mis, out);
      run<16, 16, 2, 8>("dwordx4, 16 lanes x2, 8 cells", buf, ntiles, te, nbins, mis, out);
      run<16, 64, 1, 4>("dwordx4, 64 lanes x1, 4 cells", buf, ntiles, te, nbins, mis, out);
    }
  }
  return 0;
}
