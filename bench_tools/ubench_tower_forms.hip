// Decides the "limb-form tower" item of the throughput pairing kernel with a measurement.
//
// The shipped tower (keaki_amd/csrc/pair261.hip.h) keeps every Fq2 component as 8 canonical 32-bit words between tower operations: an Fq6
// product CUTS its six operands into 29-bit limbs, runs three six-product streams and PACKS the three results (pack = repack + one
// conditional subtraction on the hardware carry chain: the cheap canonicaliser), and the additions / subtractions around the products are
// saturated modular operations (25 instructions each, result canonical). "Limb form end to end" would drop the cuts and packs.
//
// What this program measures, at the kernel's occupancy (two waves per SIMD) and on the kernel's own functions:
//   (1) fq6_mul as shipped                                  -- chain r = a b, a = r
//   (2) fq6_mul with NO cut and NO pack                     -- the operands stay limbs, the stream outputs (exact limbs, < 2p) feed the next product:
//                                                              the bounds close for a pure Fq6 chain (inputs < 2p -> outputs < (168/169 + 1) p), so this
//                                                              is a legitimate limb-form product AND the upper bound of what dropping cut / pack can buy
//   (3) fq12_mul as shipped (Karatsuba over Fq6)
//   (4) fq12_mul in limb form: sums, differences and the shift by v on limbs (limb-wise + one carry pass), products without cut / pack, and the
//       one thing the limb form cannot avoid: its recombined outputs are < 6p / < 22p (m - t0 - t1 + 4p, t0 + xi t1), the next product needs
//       < 2p for its bounds to close, and without the saturated carry chain the only reduction is a product by one (u29_mul(x, 1): 205
//       instructions) -- twelve per Fq12 product (six sums, six results)
//   (5) fq12_cyc_sqr as shipped, (6) in limb form (same rule: recombination on limbs, six reductions by one)
//   (7) the pieces alone: cut, pack, saturated add, limb add + carry, product by one
// (4) and (6) are checked against (3) and (5): the canonical value of every component after the chain must be equal.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I keaki_amd/csrc -o bench_tools/ubench_tower_forms bench_tools/ubench_tower_forms.hip
#include "pair261.hip.h"
#include <stdio.h>
#include <vector>
using namespace bn254;
using namespace bn254::p261;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---- limb form ---------------------------------------------------------------------------------------------------------------------------
struct L6 { U29 c0, c1, c2; };          // this lane's components: carried limbs (<= 2^29 + 8), value bound stated at each use
struct L12 { L6 c0, c1; };
KDEV U29 ladd(const U29& a, const U29& b) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = a.l[i] + b.l[i];
  return u29_carry(t);
}
// a - b - c + K (K >= bound of b + c, limbs biased by 2^31: K4W / K8W tables carry that bias)
KDEV U29 lsub2(const U29& a, const U29& b, const U29& c, const u32 (&KW)[9]) {
  U29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = a.l[i] - b.l[i] - c.l[i] + KW[i];
  return u29_carry(t);
}
KDEV U29 renorm(const U29& x) { return u29_mul(x, u29_one()); }      // any carried value < ~100p -> exact limbs, < 2p
// inputs < 2p (carried), outputs exact limbs < 2p: c0 < (2*(2+2) + 2*(9*2+2)*(2+2))/169 + 1 = 1.99 p
KDEV void fq6_mul_l(L6* r, const L6* a, const L6* b) {
  const XF x0 = x_of(a->c0), x1 = x_of(a->c1), x2 = x_of(a->c2);
  const XF xx1 = x_of(xi_limbs(x1.s, x1.o, Q29::K2)), xx2 = x_of(xi_limbs(x2.s, x2.o, Q29::K2));
  const YF y0 = y_of(b->c0, Q29::K2), y1 = y_of(b->c1, Q29::K2), y2 = y_of(b->c2, Q29::K2);
  const U29 c0 = dot3(x0, y0, xx1, y2, xx2, y1);
  const U29 c1 = dot3(x0, y1, x1, y0, xx2, y2);
  const U29 c2 = dot3(x0, y2, x1, y1, x2, y0);
  r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
// a, b: components < p (what a chain of these functions keeps: every output passes renorm -> < 2p ... and the sums of two such values, < 4p,
// do not close the Fq6 bound (f(4, K4) = 4.98): the inputs of the Karatsuba product are renormed sums -- counted below as the limb form needs them)
KDEV void fq12_mul_l(L12* r, const L12* a, const L12* b) {
  L6 t0, t1, m;
  {
    // s0, s1 < 4p: one side of a product may be < 4p when the other is < 2p? f(x < 4, y < 2, K2): (4*4 + 2*(40)*4)/169 + 1 = 2.99 p: no.
    // Both sums are brought back below 2p: six products by one.
    L6 s0 = {renorm(ladd(a->c0.c0, a->c1.c0)), renorm(ladd(a->c0.c1, a->c1.c1)), renorm(ladd(a->c0.c2, a->c1.c2))};
    L6 s1 = {renorm(ladd(b->c0.c0, b->c1.c0)), renorm(ladd(b->c0.c1, b->c1.c1)), renorm(ladd(b->c0.c2, b->c1.c2))};
    fq6_mul_l(&m, &s0, &s1);
  }
  fq6_mul_l(&t0, &a->c0, &b->c0);
  fq6_mul_l(&t1, &a->c1, &b->c1);
  // c1 = m - t0 - t1 (+ 4p) < 6p; c0 = t0 + v t1 = (t0.c0 + xi t1.c2, t0.c1 + t1.c0, t0.c2 + t1.c1) < 22p, 4p, 4p -> all six back below 2p
  r->c1.c0 = renorm(lsub2(m.c0, t0.c0, t1.c0, Q29::K4W));
  r->c1.c1 = renorm(lsub2(m.c1, t0.c1, t1.c1, Q29::K4W));
  r->c1.c2 = renorm(lsub2(m.c2, t0.c2, t1.c2, Q29::K4W));
  const U29 t1c2o = quad<0xB1>(t1.c2);
  r->c0.c0 = renorm(ladd(t0.c0, xi_limbs(t1.c2, t1c2o, Q29::K2)));
  r->c0.c1 = renorm(ladd(t0.c1, t1.c0));
  r->c0.c2 = renorm(ladd(t0.c2, t1.c1));
}
// Granger-Scott squaring, limb form: the three Fq4 squarings on limbs (x, y < 2p), recombination 3 t -+ 2 r on limbs, six reductions by one
KDEV void fq4_sqr_l(U29* t0, U29* t1, const U29& x, const U29& y) {
  const bool odd = lane_odd() != 0;
  const XF xx = x_of(x), xy = x_of(y);
  const XF xxy = x_of(xi_limbs(xy.s, xy.o, Q29::K2));
  const YF yy = y_of(xy.s, Q29::K2);
  U29 sx, sy;
  XF x2;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    sx.l[i] = odd ? xx.o.l[i] : xx.s.l[i] + xx.o.l[i];
    sy.l[i] = odd ? 2u * xx.s.l[i] : xx.s.l[i] - xx.o.l[i] + Q29::K2[i];
    x2.s.l[i] = 2u * xx.s.l[i]; x2.o.l[i] = 2u * xx.o.l[i];
  }
  sy = u29_carry(sy);
  U29 r;
  u29_dot3_asm(r.l, sx.l, sy.l, xxy.s.l, yy.y0.l, xxy.o.l, yy.y1.l);
  *t0 = r;
  *t1 = dot1(x2, yy);
}
KDEV U29 l3m2(const U29& t, const U29& r, bool plus) {      // 3 t + 2 r  |  3 t - 2 r + 4p
  U29 o;
#pragma unroll
  for (int i = 0; i < 9; i++) o.l[i] = plus ? 3u * t.l[i] + 2u * r.l[i] : 3u * t.l[i] - 2u * r.l[i] + Q29::K4W[i];
  return renorm(u29_carry(o));
}
KDEV void fq12_cyc_sqr_l(L12* r, const L12* a) {
  const U29 r0 = a->c0.c0, r4 = a->c0.c1, r3 = a->c0.c2, r2 = a->c1.c0, r1 = a->c1.c1, r5 = a->c1.c2;
  U29 t0, t1, t2, t3, t4, t5;
  fq4_sqr_l(&t4, &t5, r4, r5);
  fq4_sqr_l(&t2, &t3, r2, r3);
  fq4_sqr_l(&t0, &t1, r0, r1);
  const U29 x5 = xi_limbs(t5, quad<0xB1>(t5), Q29::K2);     // < 20p
  r->c0.c0 = l3m2(t0, r0, false);
  r->c1.c1 = l3m2(t1, r1, true);
  r->c1.c0 = l3m2(x5, r2, true);
  r->c0.c2 = l3m2(t4, r3, false);
  r->c0.c1 = l3m2(t2, r4, false);
  r->c1.c2 = l3m2(t3, r5, true);
}

// ---- kernels -----------------------------------------------------------------------------------------------------------------------------
KDEV Fq seed_fq(u32 a, u32 b) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = a * 2654435761u + b * 40503u + 977u * i;
  r.l[7] &= 0x0FFFFFFFu;                                   // < 2^252 < p
  return r;
}
KDEV Fq12 seed12(u32 s) {
  Fq12 f;
  Fq* c = reinterpret_cast<Fq*>(&f);
  for (int i = 0; i < 6; i++) c[i] = seed_fq(threadIdx.x + 64 * blockIdx.x + s, i + 1);
  return f;
}
KDEV L12 to_l(const Fq12& f) {
  return {{cut(f.c0.c0.v), cut(f.c0.c1.v), cut(f.c0.c2.v)}, {cut(f.c1.c0.v), cut(f.c1.c1.v), cut(f.c1.c2.v)}};
}
KDEV Fq12 from_l(const L12& f) {     // canonical words of every component (value mod p)
  Fq12 r;
  r.c0.c0.v = pack(renorm(f.c0.c0)); r.c0.c1.v = pack(renorm(f.c0.c1)); r.c0.c2.v = pack(renorm(f.c0.c2));
  r.c1.c0.v = pack(renorm(f.c1.c0)); r.c1.c1.v = pack(renorm(f.c1.c1)); r.c1.c2.v = pack(renorm(f.c1.c2));
  return r;
}
// renorm multiplies by one in the 2^261 form, i.e. leaves the residue unchanged: from_l(to_l(f)) == f for canonical f
template <int OP>
__global__ void __launch_bounds__(256) k(u32* out, int iters) {
  Fq12 f = seed12(1), g = seed12(77);
  L12 lf = to_l(f), lg = to_l(g);
  Fq pa = f.c0.c0.v, pb = g.c0.c0.v;
  U29 la = lf.c0.c0, lb = lg.c0.c0;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { Fq6 r; fq6_mul(&r, &f.c0, &g.c0); f.c0 = r; }
    else if (OP == 1) { L6 r; fq6_mul_l(&r, &lf.c0, &lg.c0); lf.c0 = r; }
    else if (OP == 2) { Fq12 r; fq12_mul(&r, &f, &g); f = r; }
    else if (OP == 3) { L12 r; fq12_mul_l(&r, &lf, &lg); lf = r; }
    else if (OP == 4) { Fq12 r; fq12_cyc_sqr(&r, &f); f = r; }
    else if (OP == 5) { L12 r; fq12_cyc_sqr_l(&r, &lf); lf = r; }
    else if (OP == 6) { for (int u = 0; u < 16; u++) { la = cut(pa); for (int i = 0; i < 8; i++) pa.l[i] ^= la.l[i] + la.l[8]; } }   // 16 cuts (+ 16 plain instructions that keep every limb alive)
    else if (OP == 7) { for (int u = 0; u < 16; u++) { pa = pack(la); la.l[0] = (la.l[0] ^ pa.l[1]) & Q29::MASK; } }   // 16 packs
    else if (OP == 8) { for (int u = 0; u < 16; u++) pa = pa + pb; }                                         // 16 saturated modular additions
    else if (OP == 9) { for (int u = 0; u < 16; u++) la = ladd(la, lb); }                                    // 16 limb additions + carry pass
    else if (OP == 10) { for (int u = 0; u < 16; u++) la = renorm(la); }                                     // 16 products by one
  }
  if (OP == 1 || OP == 3 || OP == 5) f = from_l(lf);
  u32 r = 0;
  const u32* w = reinterpret_cast<const u32*>(&f);
  for (int i = 0; i < 48; i++) r ^= w[i] * (2 * i + 1);
  for (int i = 0; i < 8; i++) r ^= pa.l[i];
  for (int i = 0; i < 9; i++) r ^= la.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// equality of the two forms after a short chain (every lane, every component)
template <int OPA, int OPB>
__global__ void __launch_bounds__(64) k_check(u32* bad, int iters) {
  Fq12 f = seed12(1), g = seed12(77);
  L12 lf = to_l(f), lg = to_l(g);
  for (int it = 0; it < iters; it++) {
    if (OPA == 2) { Fq12 r; fq12_mul(&r, &f, &g); f = r; L12 q; fq12_mul_l(&q, &lf, &lg); lf = q; }
    if (OPA == 4) { Fq12 r; fq12_cyc_sqr(&r, &f); f = r; L12 q; fq12_cyc_sqr_l(&q, &lf); lf = q; }
    if (OPA == 0) { Fq6 r; fq6_mul(&r, &f.c0, &g.c0); f.c0 = r; L6 q; fq6_mul_l(&q, &lf.c0, &lg.c0); lf.c0 = q; }
  }
  const Fq12 h = from_l(lf);
  const u32 *a = reinterpret_cast<const u32*>(&f), *b = reinterpret_cast<const u32*>(&h);
  u32 d = 0;
  for (int i = 0; i < 48; i++) d |= a[i] ^ b[i];
  if (d) atomicAdd(bad, 1u);
}

template <int OP>
int run(const char* name, int waves, double units_per_iter, int iters) {
  int blocks = 256 * waves;
  u32* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 4);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<OP>)));
  printf("%-62s waves/SIMD=%d  %8.3f ms  %10.1f SIMD-cycles per unit   [%d VGPRs, %zu B scratch]\n", name, waves, ms,
         ms * 1e-3 * 2.4e9 / ((double)waves * iters * units_per_iter), fa.numRegs, (size_t)fa.localSizeBytes);
  CK(hipFree(out));
  return 0;
}
int main() {
  u32* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL((k_check<0, 1>), dim3(64), dim3(64), 0, 0, bad, 5);
  hipLaunchKernelGGL((k_check<2, 3>), dim3(64), dim3(64), 0, 0, bad, 5);
  hipLaunchKernelGGL((k_check<4, 5>), dim3(64), dim3(64), 0, 0, bad, 5);
  u32 hb = 1; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  printf("limb forms == shipped forms after 5-step chains (fq6_mul, fq12_mul, fq12_cyc_sqr; 4096 lanes each): %s (%u mismatching lanes)\n", hb ? "NO" : "yes", hb);
  printf("unit: SIMD-cycles per wave-unit at 2.4 GHz, 1024 SIMDs, all waves of a SIMD interleaved (as profiles/r01_ubench_u29_gfx950.txt)\n");
  for (int w : {1, 2}) {
    run<0>("fq6_mul shipped (cut 6, 3 x dot3, pack 3)", w, 1, 600);
    run<1>("fq6_mul limbs in, limbs out (no cut, no pack)", w, 1, 600);
    run<2>("fq12_mul shipped (Karatsuba over Fq6, saturated recombination)", w, 1, 200);
    run<3>("fq12_mul limb form (recombination on limbs, 12 products by one)", w, 1, 200);
    run<4>("fq12_cyc_sqr shipped", w, 1, 300);
    run<5>("fq12_cyc_sqr limb form (6 products by one)", w, 1, 300);
    run<6>("cut (words -> 9 limbs)", w, 16, 2000);
    run<7>("pack (exact limbs < 2p -> canonical words)", w, 16, 2000);
    run<8>("saturated modular addition", w, 16, 2000);
    run<9>("limb addition + carry pass", w, 16, 2000);
    run<10>("product by one (the limb form's only reduction)", w, 16, 500);
  }
  return hb ? 1 : 0;
}
