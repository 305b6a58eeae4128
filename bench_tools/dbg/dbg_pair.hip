// Debug harness (not part of the library): per-check mismatch counters of the lane-pair tower primitives.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I keaki_amd/csrc bench_tools/dbg/dbg_pair.hip -o /tmp/dbg_pair && /tmp/dbg_pair
#include <cstdio>
#include "pairing.hip.h"
using namespace bn254;
using namespace bn254::p261;
KDEV u32 mix(u32& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
KDEV Fq pick(u32& s) {
  Fq x;
  for (int j = 0; j < 8; j++) x.l[j] = mix(s);
  x.l[7] &= 0x3FFFFFFFu;
  u32 t[8];
  for (int j = 0; j < 8; j++) t[j] = x.l[j];
  fp_reduce_once<FqParamsRef>(t);
  for (int j = 0; j < 8; j++) x.l[j] = t[j];
  return x;
}
static KNOINLINE Fq2d sqr_v2(const Fq2d a) {      // fenced limbs before the DPP reads
  const bool odd = lane_odd() != 0;
  U29 A = cut(a.v);
  fence9(A);
  const U29 O = quad<0xB1>(A);
  U29 x, y;
  for (int i = 0; i < 9; i++) { x.l[i] = odd ? O.l[i] : A.l[i] + O.l[i]; y.l[i] = odd ? 2u * A.l[i] : A.l[i] - O.l[i] + Q29::K2[i]; }
  return {pack(u29_mul(x, carry(y)))};
}
static KNOINLINE Fq2d sqr_v3(const Fq2d a) {      // fence AFTER the DPP reads (before the stream)
  const bool odd = lane_odd() != 0;
  const U29 A = cut(a.v), O = quad<0xB1>(A);
  U29 x, y;
  for (int i = 0; i < 9; i++) { x.l[i] = odd ? O.l[i] : A.l[i] + O.l[i]; y.l[i] = odd ? 2u * A.l[i] : A.l[i] - O.l[i] + Q29::K2[i]; }
  y = carry(y);
  fence9(x); fence9(y);
  return {pack(u29_mul(x, y))};
}
static KNOINLINE Fq2d sqr_v5(const Fq2d a) {      // the partner's limbs pass through an asm statement: the DPP mov cannot be folded into the subtraction
  const bool odd = lane_odd() != 0;
  const U29 A = cut(a.v);
  U29 O = quad<0xB1>(A);
  fence9(O);
  U29 x, y;
  for (int i = 0; i < 9; i++) { x.l[i] = odd ? O.l[i] : A.l[i] + O.l[i]; y.l[i] = odd ? 2u * A.l[i] : A.l[i] - O.l[i] + Q29::K2[i]; }
  return {pack(u29_mul(x, carry(y)))};
}
static KNOINLINE Fq2d sqr_v6(const Fq2d a) {      // subtraction written the other way round: K - O + A
  const bool odd = lane_odd() != 0;
  const U29 A = cut(a.v), O = quad<0xB1>(A);
  U29 x, y;
  for (int i = 0; i < 9; i++) { x.l[i] = odd ? O.l[i] : A.l[i] + O.l[i]; const u32 n = Q29::K2[i] - O.l[i]; y.l[i] = odd ? 2u * A.l[i] : n + A.l[i]; }
  return {pack(u29_mul(x, carry(y)))};
}
static KNOINLINE Fq2d sqr_v4(const Fq2d a) {      // x carried too
  const bool odd = lane_odd() != 0;
  const U29 A = cut(a.v), O = quad<0xB1>(A);
  U29 x, y;
  for (int i = 0; i < 9; i++) { x.l[i] = odd ? O.l[i] : A.l[i] + O.l[i]; y.l[i] = odd ? 2u * A.l[i] : A.l[i] - O.l[i] + Q29::K2[i]; }
  return {pack(u29_mul(carry(x), carry(y)))};
}
__global__ void __launch_bounds__(64) k(unsigned long long* cnt) {
  u32 pair = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  u32 s = (12345u ^ (pair * 0x9E3779B9u)) | 1u;
  const u32 par = lane_odd();
  auto comp = [&](const Fq2& x) { return par ? x.c1 : x.c0; };
  auto ld = [&](const Fq2& x) { return Fq2d{to261(comp(x))}; };
  auto eq = [&](const Fq2d& got, const Fq2& exp) { return fq_eq(to256(got.v), comp(exp)); };
  auto bump = [&](int idx, bool ok) { if (!ok) atomicAdd(&cnt[idx], 1ull); };
  for (int it = 0; it < 4; it++) {
    Fq2 a = {pick(s), pick(s)}, b = {pick(s), pick(s)};
    Fq k0 = pick(s);
    Fq2d ad = ld(a), bd = ld(b);
    bump(0, fq_eq(to256(to261(k0)), k0));
    bump(1, fq_eq(to256(fq_mul261(to261(k0), to261(a.c0))), k0 * a.c0));
    bump(2, eq(ad * bd, a * b));
    bump(3, eq(fq2_sqr(ad), bn254::fq2_sqr(a)));
    bump(4, eq(fq2_mul_xi(ad), bn254::fq2_mul_xi(a)));
    bump(25, eq(sqr_v2(ad), bn254::fq2_sqr(a)));
    bump(26, eq(sqr_v3(ad), bn254::fq2_sqr(a)));
    bump(27, eq(sqr_v4(ad), bn254::fq2_sqr(a)));
    bump(28, eq(sqr_v5(ad), bn254::fq2_sqr(a)));
    bump(29, eq(sqr_v6(ad), bn254::fq2_sqr(a)));
    if (blockIdx.x == 0 && threadIdx.x < 4 && it == 0) {
      Fq got = to256(fq2_sqr(ad).v), exp = comp(bn254::fq2_sqr(a));
      Fq df = got - exp;
      printf("sqr lane %u got %08x..%08x exp %08x..%08x diff %08x %08x %08x %08x %08x %08x %08x %08x\n", threadIdx.x, got.l[0], got.l[7], exp.l[0], exp.l[7], df.l[0], df.l[1], df.l[2], df.l[3], df.l[4], df.l[5], df.l[6], df.l[7]);
    }
    bump(5, eq(fq2_inv(ad), bn254::fq2_inv(a)));
    bump(6, eq(fq2_mul_fq(ad, to261(k0)), bn254::fq2_mul_fq(a, k0)));
    {  // pieces of fq2d_sqr
      const bool odd = par != 0;
      const U29 A = cut(ad.v), O = quad<0xB1>(A);
      U29 x, y;
      for (int i = 0; i < 9; i++) { x.l[i] = odd ? O.l[i] : A.l[i] + O.l[i]; y.l[i] = odd ? 2u * A.l[i] : A.l[i] - O.l[i] + Q29::K2[i]; }
      if (blockIdx.x == 0 && threadIdx.x < 2 && it == 0) {
        const U29 cy = carry(y);
        printf("lane %u A: %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", threadIdx.x, A.l[0], A.l[1], A.l[2], A.l[3], A.l[4], A.l[5], A.l[6], A.l[7], A.l[8]);
        printf("lane %u O: %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", threadIdx.x, O.l[0], O.l[1], O.l[2], O.l[3], O.l[4], O.l[5], O.l[6], O.l[7], O.l[8]);
        printf("lane %u y: %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", threadIdx.x, y.l[0], y.l[1], y.l[2], y.l[3], y.l[4], y.l[5], y.l[6], y.l[7], y.l[8]);
        printf("lane %u c: %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", threadIdx.x, cy.l[0], cy.l[1], cy.l[2], cy.l[3], cy.l[4], cy.l[5], cy.l[6], cy.l[7], cy.l[8]);
      }
      Fq xb = pack(u29_mul(carry(x), u29_const(Q29::ONE))), yb = pack(u29_mul(carry(y), u29_const(Q29::ONE)));
      bump(18, fq_eq(to256(xb), odd ? a.c0 : a.c0 + a.c1));
      bump(19, fq_eq(to256(yb), odd ? a.c1 + a.c1 : a.c0 - a.c1));
      bump(23 + par, fq_eq(to256(yb), odd ? a.c1 + a.c1 : a.c0 - a.c1));
      if (blockIdx.x == 0 && threadIdx.x < 2 && it == 0) {
        Fq got = to256(yb), exp = odd ? a.c1 + a.c1 : a.c0 - a.c1;
        printf("lane %u got %08x %08x .. %08x exp %08x %08x .. %08x | a.c0 %08x a.c1 %08x comp %08x\n", threadIdx.x, got.l[0], got.l[1], got.l[7], exp.l[0], exp.l[1], exp.l[7], a.c0.l[0], a.c1.l[0], comp(a).l[0]);
      }
      Fq pr = pack(u29_mul(x, carry(y)));
      bump(20, fq_eq(to256(pr), comp(bn254::fq2_sqr(a))));
      Fq pr2 = pack(u29_mul(carry(x), carry(y)));
      bump(21, fq_eq(to256(pr2), comp(bn254::fq2_sqr(a))));
      bump(22, fq_eq(comp(bn254::fq2_sqr(a)), odd ? fq_dbl(a.c0 * a.c1) : (a.c0 + a.c1) * (a.c0 - a.c1)));
    }
    // xi_limbs
    {
      const XF x = x_of(cut(ad.v));
      U29 xl = xi_limbs(x.s, x.o, Q29::K2);
      // bring back: value < 11p: multiply by one (2^261-form one as plain factor reduces): u29_mul(xl, ONE-limbs) keeps the residue x*2^261 -> *2^261/2^261
      Fq back = pack(u29_mul(xl, u29_const(Q29::ONE)));
      bump(7, fq_eq(to256(back), comp(bn254::fq2_mul_xi(a))));
    }
    // y_of: y0 is the real part in both lanes, y1 the imaginary part (negated in even lanes)
    {
      const YF y = y_of(cut(bd.v), Q29::K2);
      Fq y0 = pack(u29_mul(y.y0, u29_const(Q29::ONE))), y1 = pack(u29_mul(y.y1, u29_const(Q29::ONE)));
      bump(8, fq_eq(to256(y0), b.c0));
      bump(9, fq_eq(to256(y1), par ? b.c1 : fq_zero() - b.c1));
    }
    // dot1 / dot2 / dot3 with plain operands
    Fq2 c = {pick(s), pick(s)}, d = {pick(s), pick(s)}, e = {pick(s), pick(s)}, f = {pick(s), pick(s)};
    {
      const XF xa = x_of(cut(ad.v)), xc = x_of(cut(ld(c).v)), xe = x_of(cut(ld(e).v));
      const YF yb = y_of(cut(bd.v), Q29::K2), yd = y_of(cut(ld(d).v), Q29::K2), yf = y_of(cut(ld(f).v), Q29::K2);
      bump(10, eq(Fq2d{pack(dot1(xa, yb))}, a * b));
      bump(11, eq(Fq2d{pack(dot2(xa, yb, xc, yd))}, a * b + c * d));
      bump(12, eq(Fq2d{pack(dot3(xa, yb, xc, yd, xe, yf))}, a * b + c * d + e * f));
    }
    {
      Fq6 xa = {ld(a), ld(c), ld(d)}, xb = {ld(b), ld(e), ld(f)}, xr;
      fq6_mul(&xr, &xa, &xb);
      Fq2 r0 = a * b + bn254::fq2_mul_xi(c * f) + bn254::fq2_mul_xi(d * e), r1 = a * e + c * b + bn254::fq2_mul_xi(d * f), r2 = a * f + c * e + d * b;
      bump(13, eq(xr.c0, r0)); bump(14, eq(xr.c1, r1)); bump(15, eq(xr.c2, r2));
    }
    {
      Fq2d t0, t1;
      fq4_sqr(&t0, &t1, ad, bd);
      bump(16, eq(t0, bn254::fq2_sqr(a) + bn254::fq2_mul_xi(bn254::fq2_sqr(b))));
      bump(17, eq(t1, bn254::fq2_dbl(a * b)));
    }
  }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 32 * 8);
  hipMemset(d, 0, 32 * 8);
  hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
  unsigned long long h[32];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[] = {"to261/to256", "fq_mul261", "fq2d_mul", "fq2d_sqr", "fq2_mul_xi(sat)", "fq2_inv", "fq2_mul_fq", "xi_limbs", "y_of.y0", "y_of.y1",
                         "dot1", "dot2", "dot3", "fq6.c0", "fq6.c1", "fq6.c2", "fq4.t0", "fq4.t1", "sqr.x", "sqr.y", "sqr.prod", "sqr.prod_carried", "ref_sqr_consistent", "sqr.y even", "sqr.y odd", "sqr_v2 fence-before", "sqr_v3 fence-after", "sqr_v4 carry x", "sqr_v5 no-fold", "sqr_v6 K-O+A"};
  for (int i = 0; i < 30; i++) printf("%-16s %llu\n", names[i], h[i]);
  printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
