// Pricing of BATCHED-AFFINE bucket accumulation against the shipped XYZZ mixed addition.
//
// Batched affine: a lane keeps B pending affine additions (x1, y1) + (x2, y2), shares ONE field inversion with the other 63 lanes of its wave
// (Montgomery's trick: per-lane prefix products, a product scan across the wave, one inverse, back down), then finishes every addition with
// lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1:
//     per addition  : 1 M (prefix) + 2 M (peel the inverse) + 2 M + 1 S (the addition itself) = 5 M + 1 S      -- against 8 M + 2 S of XYZZ
//     per wave step : 12 products of the two scans + 2 + ONE inversion, paid by every lane of the wave, amortised over B additions per lane
// This program measures each piece in the 29-bit arithmetic of fq29.hip.h on gfx950, in SIMD-cycles per wave-addition (the unit of
// profiles/r01_ubench_u29_gfx950.txt), at the occupancy the LDS footprint allows. The operands of the pending additions are SYNTHESISED in
// registers (no table gather, no second read of the points, no store of the partial sums): what comes out is a LOWER bound of a real kernel.
// Only the prefix products live in LDS (36 B per pending addition), which is what bounds B: 64 lanes x B x 36 B per wave of 160 KB per CU.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I keaki_amd/csrc -o bench_tools/ubench_batch_affine bench_tools/ubench_batch_affine.hip
#include "fq29.hip.h"
#include <stdio.h>
using namespace bn254;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ U29 seed(u32 a, u32 b) {
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (a * 2654435761u + b * 40503u + 977u * i) & Q29::MASK;
  r.l[8] &= 0xFFFFF;
  return r;
}
__device__ __forceinline__ u32 fold(const U29& a) {
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) r ^= a.l[i];
  return r;
}
__device__ __forceinline__ U29 shfl(const U29& a, int src_lane) {
  U29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (u32)__shfl((int)a.l[i], src_lane, 64);
  return r;
}

// ---- (1) the shipped XYZZ mixed addition (msm.hip.h, k_msm_accumulate_g1_u29, common path): 8 M + 2 S with the dual product ----
__global__ void __launch_bounds__(256) k_xyzz(u32* out, int iters) {
  U29 X1 = seed(threadIdx.x, 1), Y1 = seed(blockIdx.x, 2), ZZ = u29_one(), ZZZ = u29_one();
  U29 X2 = seed(threadIdx.x, 3), Y2 = seed(blockIdx.x, 4);
  for (int it = 0; it < iters; it++) {
    X2.l[0] = (X2.l[0] + 1u) & Q29::MASK;
    const U29 U2 = u29_mul(X2, ZZ), S2 = u29_mul(Y2, ZZZ);
    const U29 P = u29_sub(U2, X1, Q29::K16), R = u29_sub(S2, Y1, Q29::K4);
    const U29 PP = u29_sqr(P), PPP = u29_mul(P, PP), Q = u29_mul(X1, PP);
    const U29 X3 = u29_sub3(u29_sqr(R), PPP, Q);
    const U29 T = u29_sub(Q, X3, Q29::K16);
    U29 NY1;
#pragma unroll
    for (int i = 0; i < 9; i++) NY1.l[i] = Q29::K2[i] - Y1.l[i];
    Y1 = u29_mul2(R, T, NY1, PPP);
    X1 = u29_mul(X3, u29_one());
    ZZ = u29_mul(ZZ, PP);
    ZZZ = u29_mul(ZZZ, PPP);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(X1) ^ fold(Y1) ^ fold(ZZ) ^ fold(ZZZ);
}

// ---- (2) the per-lane part of a batched-affine step: B pending additions, prefix products through LDS; no inversion (inv = a constant) ----
template <int B>
__global__ void __launch_bounds__(64) k_lane_part(u32* out, int iters) {
  extern __shared__ u32 lds[];                       // prefix products: [b][limb][lane]
  const u32 l = threadIdx.x;
  U29 x1 = seed(l, 11), y1 = seed(blockIdx.x, 12), x2 = seed(l, 13), y2 = seed(blockIdx.x, 14);
  u32 sum = 0;
  for (int it = 0; it < iters; it++) {
    // pass 1: d_b = x2_b - x1_b, c_b = c_(b-1) d_b
    U29 c = u29_one();
#pragma unroll 1
    for (int b = 0; b < B; b++) {
      U29 a2 = x2; a2.l[0] = (a2.l[0] + (u32)b) & Q29::MASK;            // the b-th pending addition's operands (synthetic)
      const U29 d = u29_sub(a2, x1, Q29::K4);
      c = u29_mul(c, d);
#pragma unroll
      for (int i = 0; i < 9; i++) lds[(b * 9 + i) * 64 + l] = c.l[i];
    }
    U29 run = c;                                                         // stands for the inverse of the lane's total (the wave step provides it)
    // pass 2: peel the inverses, finish the additions
#pragma unroll 1
    for (int b = B - 1; b >= 0; b--) {
      U29 a2 = x2; a2.l[0] = (a2.l[0] + (u32)b) & Q29::MASK;
      const U29 d = u29_sub(a2, x1, Q29::K4);
      U29 cp = u29_one();
      if (b > 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) cp.l[i] = lds[((b - 1) * 9 + i) * 64 + l];
      }
      const U29 inv = u29_mul(run, cp);
      run = u29_mul(run, d);
      const U29 lam = u29_mul(u29_sub(y2, y1, Q29::K4), inv);
      U29 t;
#pragma unroll
      for (int i = 0; i < 9; i++) t.l[i] = x1.l[i] + a2.l[i];
      const U29 x3 = u29_sub(u29_sqr(lam), u29_carry(t), Q29::K8);
      const U29 y3 = u29_sub(u29_mul(lam, u29_sub(x1, x3, Q29::K16)), y1, Q29::K4);
      sum ^= fold(x3) ^ fold(y3);
    }
    x1.l[1] = (x1.l[1] + sum) & Q29::MASK;
  }
  out[blockIdx.x * blockDim.x + l] = sum;
}

// ---- (3) the wave step: inclusive product scans up and down the 64 lanes + each lane's share of the inverse (inverse itself: (4)) ----
__global__ void __launch_bounds__(64) k_wave_scan(u32* out, int iters) {
  const int l = threadIdx.x;
  U29 t = seed(l, 21);
  u32 sum = 0;
  for (int it = 0; it < iters; it++) {
    U29 P = t, S = t;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const U29 up = shfl(P, l - s), dn = shfl(S, l + s);
      const U29 pm = u29_mul(P, up), sm = u29_mul(S, dn);
#pragma unroll
      for (int i = 0; i < 9; i++) { P.l[i] = l >= s ? pm.l[i] : P.l[i]; S.l[i] = l + s < 64 ? sm.l[i] : S.l[i]; }
    }
    const U29 tinv = shfl(P, 63);                                        // stands for the inverse of the wave's total
    const U29 left = shfl(P, l - 1), right = shfl(S, l + 1);
    U29 mine = u29_mul(tinv, l > 0 ? left : u29_one());
    mine = u29_mul(mine, l < 63 ? right : u29_one());
    sum ^= fold(mine);
    t.l[0] = (t.l[0] + sum) & Q29::MASK;
  }
  out[blockIdx.x * blockDim.x + l] = sum;
}

// ---- (4) one inversion per wave: the Fermat ladder in the 29-bit arithmetic (254 S + ~127 M) and the binary extended GCD (saturated words) ----
__global__ void __launch_bounds__(64) k_inv_fermat(u32* out, int iters) {
  U29 a = seed(threadIdx.x, 31);
  u32 sum = 0;
  for (int it = 0; it < iters; it++) {
    U29 acc = u29_one();
#pragma unroll 1
    for (int i = 253; i >= 0; i--) {
      acc = u29_sqr(acc);
      if ((FQ_PM2[i >> 5] >> (i & 31)) & 1) acc = u29_mul(acc, a);
    }
    sum ^= fold(acc);
    a.l[0] = (a.l[0] + sum) & Q29::MASK;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
__global__ void __launch_bounds__(64) k_inv_xgcd(u32* out, int iters, int uniform) {
  // uniform = 1: every lane inverts the SAME value (what a wave-level inversion of one total is); 0: a value per lane (divergent rounds)
  Fq a;
  for (int i = 0; i < 8; i++) a.l[i] = (uniform ? 12345u : threadIdx.x * 2654435761u) + 977u * i + blockIdx.x * (uniform ? 0u : 7u);
  a.l[7] &= 0x0FFFFFFFu;
  u32 sum = 0;
  for (int it = 0; it < iters; it++) {
    const Fq r = fq_inv_xgcd(a);
    for (int i = 0; i < 8; i++) sum ^= r.l[i];
    a.l[0] += uniform ? 1u : sum;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

static double cycles(float ms, double waves_per_simd, double units) { return ms * 1e-3 * 2.4e9 / (waves_per_simd * units); }

int main() {
  u32* out; CK(hipMalloc(&out, (size_t)1024 * 16 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  printf("unit: SIMD-cycles per wave-unit at 2.4 GHz, 1024 SIMDs, all waves of a SIMD interleaved (as profiles/r01_ubench_u29_gfx950.txt)\n");
  double xyzz3 = 0;
  for (int w : {1, 2, 3}) {
    const int iters = 400;
    hipLaunchKernelGGL(k_xyzz, dim3(256 * w), dim3(256), 0, 0, out, 4); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_xyzz, dim3(256 * w), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double c = cycles(ms, w, iters);
    if (w == 3) xyzz3 = c;
    printf("XYZZ mixed addition (shipped, 8M+2S)         waves/SIMD=%d  %9.0f cycles per addition\n", w, c);
  }
  double scan1 = 0, fermat1 = 0, xgcd_u1 = 0, xgcd_d1 = 0;
  {
    const int iters = 50;
    for (int w : {1, 2}) {
      hipLaunchKernelGGL(k_wave_scan, dim3(1024 * w), dim3(64), 0, 0, out, 2); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_wave_scan, dim3(1024 * w), dim3(64), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double c = cycles(ms, w, iters);
      if (w == 1) scan1 = c;
      printf("wave step: 2 product scans over 64 lanes + 2 M waves/SIMD=%d  %9.0f cycles per wave step\n", w, c);
    }
    const int it2 = 4;
    hipLaunchKernelGGL(k_inv_fermat, dim3(1024), dim3(64), 0, 0, out, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_inv_fermat, dim3(1024), dim3(64), 0, 0, out, it2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); fermat1 = cycles(ms, 1, it2);
    printf("inversion, Fermat ladder (254 S + 127 M)        waves/SIMD=1  %9.0f cycles per inversion\n", fermat1);
    for (int uni : {1, 0}) {
      hipLaunchKernelGGL(k_inv_xgcd, dim3(1024), dim3(64), 0, 0, out, 1, uni); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_inv_xgcd, dim3(1024), dim3(64), 0, 0, out, it2, uni); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      (uni ? xgcd_u1 : xgcd_d1) = cycles(ms, 1, it2);
      printf("inversion, binary extended GCD, %-16s waves/SIMD=1  %9.0f cycles per inversion\n", uni ? "one value / wave" : "a value per lane", uni ? xgcd_u1 : xgcd_d1);
    }
  }
  auto lane_part = [&](auto kern, int B, int w, double* res) -> int {
    const int iters = 40;
    const size_t shm = (size_t)B * 9 * 64 * 4;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(kern, dim3(1024 * w), dim3(64), shm, 0, out, 2); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(1024 * w), dim3(64), shm, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    *res = cycles(ms, w, (double)iters * B);
    return 0;
  };
  printf("\nper-lane part (5 M + 1 S per addition, prefix products through LDS; operands synthetic = no gather, no re-read, no store):\n");
  struct Row { int B, w; double lane; } rows[6];
  int nr = 0;
  for (int w : {1, 2}) {
    double c;
    if (lane_part(k_lane_part<8>, 8, w, &c)) return 1;   rows[nr++] = {8, w, c};
    if (lane_part(k_lane_part<16>, 16, w, &c)) return 1; rows[nr++] = {16, w, c};
    if (w == 1 || 32 * 9 * 64 * 4 * 2 * 4 <= 160 * 1024) { if (lane_part(k_lane_part<32>, 32, w, &c)) return 1; rows[nr++] = {32, w, c}; }
  }
  for (int i = 0; i < nr; i++) {
    const int B = rows[i].B, w = rows[i].w;
    const double lds_kb = B * 9 * 64 * 4 / 1024.0;
    // the wave step and the inversion are per wave: divided by the B additions each lane finishes with them. Their cost is taken at the
    // occupancy of one wave per SIMD (a second wave can overlap them with its own lane part only as far as the issue port allows).
    const double wave_best = (scan1 + xgcd_u1) / B, wave_fermat = (scan1 + fermat1) / B;
    printf("B=%2d  waves/SIMD=%d  LDS %5.1f KB/wave (max %d waves per CU)  lane part %7.0f  + wave step/B: %6.0f (xgcd) | %6.0f (Fermat)  = %7.0f | %7.0f cycles per addition   [XYZZ at 3 waves: %.0f]\n",
           B, w, lds_kb, (int)(160.0 / lds_kb), rows[i].lane, wave_best, wave_fermat, rows[i].lane + wave_best, rows[i].lane + wave_fermat, xyzz3);
  }
  printf("\nnot included in the batched-affine figures: the gather of the table point (as XYZZ), a SECOND read of both operands in pass 2 (or 72 B of LDS per pending "
         "addition to keep them), the bucket's running sum living in LDS / memory instead of registers, the P = +-Q and identity cases per pending addition.\n");
  return 0;
}
