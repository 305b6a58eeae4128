// FK23 batch openings on the GPU: all d KZG opening proofs at the d-th roots of unity in O(d log d) group operations.
//
// Replaces kzg::open_fk (reference src/kzg.rs:157-203; caller src/vec.rs:40), whose work is three ark-poly group FFTs
// (`domain_2d.fft(&s)`, `domain_2d.ifft(&hat_h)`, `domain_d.fft(&h)`) and 2d scalar multiplications (`hat_s[i].mul(hat_a[i])`,
// :189-191). The same proofs with a third less work: with V = hat_a o hat_s (the 2d products) and w = omega_2d,
//   h_i = (1/2) (DFT_d^-1(V_even)_i + w^-i DFT_d^-1(V_odd)_i)  (i < d)   =>   proofs = DFT_d(h) = (1/2) V_even + (1/2) DFT_d(D o DFT_d^-1(V_odd))
// (D_i = w^-i): the even half of the products IS half of the answer, only the odd half goes through one inverse and one forward transform of
// size d with a twist between them -- d (log2 d + 3) scalar multiplications instead of the d (1.5 log2 d + 3) of the three transforms
// (24 d instead of 34.5 d at d = 2^21). Forward transforms run decimation-in-frequency (natural order in, bit-reversed positions out), the
// inverse one decimation-in-time (bit-reversed in, natural out), so the only permutation is the one on the d affine proofs at the end:
//   hat_s <- DIF_2d(S), S[i] = [tau^(d-1-i)]_1 (i < d), identity (i >= d)     k_fk_load + stages; cached per (SRS, d). Position q < d holds
//                                                                             hat_s[2 brev(q)], position d + q holds hat_s[2 brev(q) + 1]
//   E[q] = (d a[2 brev(q)]) hat_s[q],  O[q] = a[2 brev(q) + 1] hat_s[d + q]   k_fk_pointwise   (a = DFT_2d(0..0, p) / 2d, natural order)
//   O <- DIT_d (inverse twiddles), O[i] <- w^-i O[i], O <- DIF_d              k_g1_fft_stage_map, k_g1_mul_jac_strided
//   proofs[brev(q)] = affine(E[q] + O[q])                                     k_fk_finish
// A butterfly is one scalar multiplication of a Jacobian point by a twiddle factor plus an add and a subtract; one lane per butterfly.
// tests/fk_shard_model.py::open_fk_split is this pipeline over Z_q, index for index.
#include "ec_batch.hip.h"
#include "jac29.hip.h"
#include "internal.h"

namespace bn254 {

// k * P for a Jacobian P: NAF ladder in the 29-bit lazy arithmetic (jac29.hip.h). k: Montgomery Fr.
// tab_lane: this lane's slot of the window-table workspace (null: the table lives in private memory)
// GT: the kernel was launched with a table workspace (one instantiation per form: a kernel that carries both ladders needs 190 registers)
template <bool GT>
KDEV G1Jac jac_scalar_mul_t(const G1Jac& p, const Fr& k_mont, uint4* tab_lane) {
  if constexpr (GT) return jac_scalar_mul_gtab_u29(p, k_mont, tab_lane);
  else return jac_scalar_mul_u29(p, k_mont);
}
// A launch of a per-lane-scalar ladder kernel covers the lanes [first, first + gridDim.x * 64) of the whole job; with a table workspace the
// host launches at most FK_TAB_LANES lanes at a time and lane b uses slot b - first.
KDEV uint4* ladder_slot(uint4* tab, u32 b, u32 first) { return tab ? tab + (size_t)(b - first) * GTAB_UINT4_PER_LANE : nullptr; }
KDEV bool fr_is_one(const Fr& a) {
  u32 o = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) o |= a.l[j] ^ FrParams::ONE[j];
  return o == 0;
}

static __global__ void __launch_bounds__(256) k_fk_load(const G1Aff* __restrict__ srs, u32 d, G1Jac* __restrict__ s) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * d) return;
  s[i] = i < d ? jac_from_aff(srs[d - 1 - i]) : jac_inf<Fq>();
}
static __global__ void __launch_bounds__(64) k_g1_jac_to_aff(const G1Jac* __restrict__ a, u32 n, G1Aff* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = jac_to_aff(a[i]);
}

// ---- FK23 sharded over R = 2^rho ranks (keaki_hip_fk_shard_*; index maps modelled in tests/fk_shard_model.py) ---------------------
// A transform of d = R * M points never exists in one memory. Two layouts of the d positions over the ranks:
//   cyclic: rank r holds position i = k R + r at local index k     (the HIGH log2 M bits of a position are local)
//   block:  rank r holds position i = r M + k at local index k     (the LOW  log2 M bits are local)
// A radix-2 stage of span `len` pairs positions that differ in bit log2(len) - 1, so spans 2R..d run on the cyclic layout, spans 2..M
// on the block layout, and ONE all-to-all (the caller's: RCCL over xGMI) switches between them. Forward transforms run
// decimation-in-frequency (natural order in, bit-reversed positions out: cyclic -> block), the inverse one decimation-in-time
// (bit-reversed in, natural out: block -> cyclic), so no bit-reversal permutation -- which would be a second all-to-all -- is needed
// anywhere but on the d affine proofs at the very end.
// One stage over a LOCAL array of m points: local span 2 * half; the butterfly at local offset j takes the twiddle tw[(j A + B) stride]
// ((A, B) = (R, rank) on the cyclic layout, (1, 0) on the block layout). Lane order as k_g1_fft_stage: a wave shares one twiddle
// while the stage has at least 64 blocks.
// UNIFORM (chosen by the host: at least 64 blocks and at least one full wave): every wave has ONE twiddle and takes the sliding-window ladder.
// AS29: the butterfly's add and subtract in the lazy limbs with their shared products computed once (option fk_addsub29, default on)
template <bool DIT, bool UNIFORM, bool GT, bool AS29>
static __global__ void __launch_bounds__(64, 3) k_g1_fft_stage_map(G1Jac* __restrict__ a, const Fr* __restrict__ tw, u32 m, u32 half, u32 A, u32 B, u32 stride,
                                                                   uint4* __restrict__ tab, u32 first) {
  u32 b = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= m / 2) return;
  const u32 nblocks = m / (2 * half);
  u32 j, blk;
  if (nblocks >= 64) { blk = b % nblocks; j = b / nblocks; }
  else { j = b % half; blk = b / half; }
  const u32 i0 = blk * 2 * half + j, i1 = i0 + half;
  Fr w = tw[((size_t)j * A + B) * stride];
  __shared__ unsigned char dig[UNIFORM ? 2 * UNIFORM_DIG_STRIDE : 4];
  // w * pt into the lazy limbs; false: the product is the identity
  auto mul_j = [&](const G1Jac& pt, J29& out) -> bool {
    if constexpr (UNIFORM) return jac_scalar_mul_uniform_u29_j(pt, w, dig, out);
    else if constexpr (GT) return jac_scalar_mul_gtab_u29_j(pt, w, ladder_slot(tab, b, first), out);
    else return jac_scalar_mul_u29_j(pt, w, out);
  };
  // (u + v, u - v): in the lazy limbs with the shared products computed once (jac29.hip.h: j29_addsub); an identity on either side or
  // u = +-v takes the generic saturated additions
  auto addsub = [&](const G1Jac& u, bool v_inf, const J29& vj, G1Jac& sum, G1Jac& diff) {
    if constexpr (AS29) {
      J29 sj, dj;
      if (!v_inf && !jac_is_inf(u) && j29_addsub(j29_from_sat(u), vj, sj, dj)) {
        sum = j29_to_sat(sj);
        diff = j29_to_sat(dj);
        return;
      }
    }
    G1Jac v = v_inf ? jac_inf<Fq>() : j29_to_sat(vj);
    sum = jac_add(u, v);
    v.y = -v.y;
    diff = jac_add(u, v);
  };
  if (DIT) {
    const G1Jac v = a[i1];
    J29 vj;
    bool v_inf;
    if (fr_is_one(w)) { v_inf = jac_is_inf(v); vj = j29_from_sat(v); }
    else v_inf = !mul_j(v, vj);
    const G1Jac u = a[i0];                 // read after the ladder: 24 registers less across it
    G1Jac s_, d_;
    addsub(u, v_inf, vj, s_, d_);
    a[i0] = s_;
    a[i1] = d_;
  } else {
    const G1Jac u = a[i0], v = a[i1];
    G1Jac s_, t;
    addsub(u, jac_is_inf(v), j29_from_sat(v), s_, t);
    a[i0] = s_;
    if (!fr_is_one(w)) {
      J29 tj;
      t = mul_j(t, tj) ? j29_to_sat(tj) : jac_inf<Fq>();
    }
    a[i1] = t;
  }
}
// ---- TWO consecutive stages in one pass (radix 4) for stages with one twiddle per wave (option fk_radix4) -------------------------------------
// Two radix-2 stages over the four points p0, p0 + h, p0 + 2h, p0 + 3h cost four scalar-mults = four chains of 129 doublings. Written out,
//   DIT (spans h, then 2h):  y0, y2 = (e0 + w1 e1) +- L1,   y1, y3 = (e0 - w1 e1) +- L2,   L1 = w2 e2 + (w2 w1) e3,   L2 = w2' e2 - (w2' w1) e3
//   DIF (spans 2h, then h):  c0 = S02 + S13,  c1 = w2 (S02 - S13),  c2 = w1 D02 + w1' D13,  c3 = (w2 w1) D02 - (w2 w1') D13   (S, D = e0 +- e2, e1 +- e3)
// (w2' / w1' = the twiddle of offset j + h) the two-term sums run as ONE chain each over a PAIR of window tables on one working curve
// (jac29.hip.h: j29_build_pair, j29_mul2_uniform): three chains instead of four, the pair of tables built once for both sums, and three
// add / subtract pairs instead of four. Lanes with a trivial twiddle (offset 0) or an identity among their points take the radix-2 sequence.
template <bool DIT, bool AS29>
static __global__ void __launch_bounds__(64, 3) k_g1_fft_stage4(G1Jac* __restrict__ a, const Fr* __restrict__ tw, u32 m, u32 h, u32 A, u32 B, u32 stride_h,
                                                                u32 stride_2h) {
  const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= m / 4) return;
  const u32 nb4 = m / (4 * h);                   // >= 64 (host): the lanes of a wave share the offset j
  const u32 blk = g % nb4, j = g / nb4;
  const u32 p0 = blk * 4 * h + j, p1 = p0 + h, p2 = p0 + 2 * h, p3 = p0 + 3 * h;
  __shared__ unsigned char dig[4 * UNIFORM_DIG_STRIDE];
  J29A T[16];                                    // ONE table storage for every ladder of this lane (16 entries: the pair tables)
  auto ladder1 = [&](const G1Jac& pt, const Fr& w, J29& out) { return jac_scalar_mul_uniform_u29_t(pt, w, dig, out, T); };
  auto addsub = [&](const G1Jac& u, bool v_inf, const J29& vj, G1Jac& sum, G1Jac& diff) {
    if constexpr (AS29) {
      J29 sj, dj;
      if (!v_inf && !jac_is_inf(u) && j29_addsub(j29_from_sat(u), vj, sj, dj)) {
        sum = j29_to_sat(sj);
        diff = j29_to_sat(dj);
        return;
      }
    }
    G1Jac v = v_inf ? jac_inf<Fq>() : j29_to_sat(vj);
    sum = jac_add(u, v);
    v.y = -v.y;
    diff = jac_add(u, v);
  };
  // one radix-2 butterfly (the fallback sequence): DIT (u, v) -> (u + w v, u - w v), DIF (u, v) -> (u + v, w (u - v))
  auto butterfly2 = [&](u32 i0, u32 i1, const Fr& w) {
    if (DIT) {
      const G1Jac v = a[i1];
      J29 vj;
      bool v_inf;
      if (fr_is_one(w)) { v_inf = jac_is_inf(v); vj = j29_from_sat(v); }
      else v_inf = !ladder1(v, w, vj);
      const G1Jac u = a[i0];
      G1Jac s_, d_;
      addsub(u, v_inf, vj, s_, d_);
      a[i0] = s_; a[i1] = d_;
    } else {
      const G1Jac u = a[i0], v = a[i1];
      G1Jac s_, t;
      addsub(u, jac_is_inf(v), j29_from_sat(v), s_, t);
      a[i0] = s_;
      if (!fr_is_one(w)) {
        J29 tj;
        t = ladder1(t, w, tj) ? j29_to_sat(tj) : jac_inf<Fq>();
      }
      a[i1] = t;
    }
  };
  // the four radix-2 butterflies of the two stages, ONE instance of the butterfly (a loop): the fallback of lanes with a trivial twiddle or an identity
  auto plain_pass = [&](const Fr* t0, const Fr* t1, const Fr* t2, const Fr* t3) {
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
      u32 i0, i1;
      const Fr* w;
      if (DIT) { i0 = q == 0 ? p0 : q == 1 ? p2 : q == 2 ? p0 : p1; i1 = q == 0 ? p1 : q == 1 ? p3 : q == 2 ? p2 : p3; }
      else { i0 = q == 0 ? p0 : q == 1 ? p1 : q == 2 ? p0 : p2; i1 = q == 0 ? p2 : q == 1 ? p3 : q == 2 ? p1 : p3; }
      w = q == 0 ? t0 : q == 1 ? t1 : q == 2 ? t2 : t3;
      butterfly2(i0, i1, *w);
    }
  };
  // the three twiddles are read where they are used (24 registers each otherwise live across the ladders):
  //   ta: offset j of the span-h stage, tb: offset j of the span-2h stage, tc: offset j + h of the span-2h stage
  const Fr* ta = tw + ((size_t)j * A + B) * stride_h;
  const Fr* tb = tw + ((size_t)j * A + B) * stride_2h;
  const Fr* tc = tw + ((size_t)(j + h) * A + B) * stride_2h;
  auto z_is_zero = [&](u32 idx) { return fq_is_zero(a[idx].z); };
  bool plain = j == 0 || z_is_zero(p0) || z_is_zero(p1) || z_is_zero(p2) || z_is_zero(p3);
  if (DIT) {
    if (plain) {
      plain_pass(ta, ta, tb, tc);
      return;
    }
    {                                                            // x0, x1 = e0 +- w1 e1 -> a[p0], a[p1]
      J29 mj;
      const bool m_inf = !ladder1(a[p1], *ta, mj);
      G1Jac x0, x1;
      addsub(a[p0], m_inf, mj, x0, x1);
      a[p0] = x0; a[p1] = x1;
    }
    J29PairTables t = {T, T + 8, {}};
    j29_build_pair(a[p2], a[p3], t);
    // L1 = w2 e2 + w2 w1 e3 -> y0, y2 = x0 +- L1;   L2 = w2' e2 - w2' w1 e3 -> y1, y3 = x1 +- L2   (ONE instance of the two-term ladder: a loop)
#pragma unroll 1
    for (int q = 0; q < 2; q++) {
      const Fr w = q ? *tc : *tb;
      J29 lj;
      const bool l_inf = !j29_mul2_uniform(t, w, fp_mul<FrParams>(w, *ta), q != 0, dig, lj);
      const u32 iu = q ? p1 : p0, iv = q ? p3 : p2;
      G1Jac ys, yd;
      addsub(a[iu], l_inf, lj, ys, yd);
      a[iu] = ys; a[iv] = yd;
    }
  } else {
    // here stride_2h belongs to the FIRST stage (span 2h: twiddles tb, tc), stride_h to the second (span h: ta)
    if (plain) {
      plain_pass(tb, tc, ta, ta);
      return;
    }
#pragma unroll 1
    for (int q = 0; q < 2; q++) {                                 // S02, D02 -> a[p0], a[p2];  S13, D13 -> a[p1], a[p3]
      const u32 iu = q ? p1 : p0, iv = q ? p3 : p2;
      G1Jac s_, d_;
      addsub(a[iu], false, j29_from_sat(a[iv]), s_, d_);
      a[iu] = s_; a[iv] = d_;
      plain = plain || jac_is_inf(s_) || jac_is_inf(d_);
    }
    if (plain) {                                                  // e0 = +-e2 or e1 = +-e3: finish with radix-2 steps
#pragma unroll 1
      for (int q = 0; q < 2; q++) {
        const Fr w = q ? *tc : *tb;
        const u32 i = q ? p3 : p2;
        if (!fr_is_one(w)) { J29 tj; a[i] = ladder1(a[i], w, tj) ? j29_to_sat(tj) : jac_inf<Fq>(); }
      }
#pragma unroll 1
      for (int q = 0; q < 2; q++) butterfly2(q ? p2 : p0, q ? p3 : p1, *ta);
      return;
    }
    {
      G1Jac c0, tdiff;
      addsub(a[p0], false, j29_from_sat(a[p1]), c0, tdiff);
      a[p0] = c0;
      J29 tj;
      a[p1] = ladder1(tdiff, *ta, tj) ? j29_to_sat(tj) : jac_inf<Fq>();       // c1 = w2 (S02 - S13)
    }
    J29PairTables t = {T, T + 8, {}};
    j29_build_pair(a[p2], a[p3], t);
    J29 c2j, c3j;
    bool ok2 = false, ok3 = false;
    // c2 = w1 D02 + w1' D13,  c3 = w2 w1 D02 - w2 w1' D13  (ONE instance of the two-term ladder: a loop; both read the tables, so the results wait)
#pragma unroll 1
    for (int q = 0; q < 2; q++) {
      const Fr kA = q ? fp_mul<FrParams>(*ta, *tb) : *tb, kB = q ? fp_mul<FrParams>(*ta, *tc) : *tc;
      J29 cj;
      const bool ok = j29_mul2_uniform(t, kA, kB, q != 0, dig, cj);
      if (q) { c3j = cj; ok3 = ok; } else { c2j = cj; ok2 = ok; }
    }
    a[p2] = ok2 ? j29_to_sat(c2j) : jac_inf<Fq>();
    a[p3] = ok3 ? j29_to_sat(c3j) : jac_inf<Fq>();
  }
}
// reversed SRS padded with identities, the cyclic slice of rank r: out[k] = S[k R + r]
static __global__ void __launch_bounds__(256) k_fk_load_cyclic(const G1Aff* __restrict__ srs, u32 d, u32 R, u32 r, u32 m, G1Jac* __restrict__ out) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const u32 i = k * R + r;
  out[k] = i < d ? jac_from_aff(srs[d - 1 - i]) : jac_inf<Fq>();
}
// out[c * rows + r] = in[r * cols + c]: the local half of a layout switch (the other half is the all-to-all)
static __global__ void __launch_bounds__(256) k_jac_transpose(const G1Jac* __restrict__ in, u32 rows, u32 cols, G1Jac* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const u32 r = i / cols, c = i % cols;
  out[(size_t)c * rows + r] = in[i];
}
KDEV u32 brev_bits(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }
// s * 2^k in the scalar field (k doublings: the factor d of the even half)
KDEV Fr fr_shl(Fr s, u32 k) {
  for (u32 i = 0; i < k; i++) s = fp_add<FrParams>(s, s);
  return s;
}
// The 2m products of the positions [base, base + m) of the d: lane t < 2m, part = t / m, k = t % m, q = base + k:
//   part 0: out_e[k] = (d a[2 brev(q)]) hs_even[k]        part 1: out_o[k] = a[2 brev(q) + 1] hs_odd[k]
// (un-sharded: base = 0, m = d, hs_even = hat_s, hs_odd = hat_s + d, out_e = work, out_o = work + d)
template <bool GT>
static __global__ void __launch_bounds__(64) k_fk_pointwise(const G1Jac* __restrict__ hs_even, const G1Jac* __restrict__ hs_odd, const Fr* __restrict__ a,
                                                            u32 log2d, u32 base, u32 m, G1Jac* __restrict__ out_e, G1Jac* __restrict__ out_o,
                                                            uint4* __restrict__ tab, u32 first) {
  u32 t = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * m) return;
  const u32 part = t >= m ? 1u : 0u, k = part ? t - m : t;
  Fr sc = a[2 * (size_t)brev_bits(base + k, log2d) + part];
  if (!part) sc = fr_shl(sc, log2d);
  const G1Jac r = jac_scalar_mul_t<GT>(part ? hs_odd[k] : hs_even[k], sc, ladder_slot(tab, t, first));
  if (part) out_o[k] = r; else out_e[k] = r;
}
// a[k] <- s[k * stride + offset] * a[k]   (the twist by omega_2d^-i: un-sharded stride 1, offset 0; cyclic layout stride R, offset rank)
template <bool GT>
static __global__ void __launch_bounds__(64) k_g1_mul_jac_strided(G1Jac* __restrict__ a, const Fr* __restrict__ s, u32 stride, u32 offset, u32 m,
                                                                  uint4* __restrict__ tab, u32 first) {
  u32 k = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  a[k] = jac_scalar_mul_t<GT>(a[k], s[(size_t)k * stride + offset], ladder_slot(tab, k, first));
}
template <bool GT>
static __global__ void __launch_bounds__(64) k_g1_mul_jac_strided_oop(const G1Jac* __restrict__ in, const Fr* __restrict__ s, u32 stride, u32 offset, u32 m,
                                                                      G1Jac* __restrict__ out, uint4* __restrict__ tab, u32 first) {
  u32 k = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  out[k] = jac_scalar_mul_t<GT>(in[k], s[(size_t)k * stride + offset], ladder_slot(tab, k, first));
}
// out[perm(k)] = affine(e[k] + o[k]); natural == true: perm = bit reversal over log2d bits of (base + k) (un-sharded), else perm = k (this
// rank's d / R proofs in position order; the permutation happens after the all-gather)
static __global__ void __launch_bounds__(64) k_fk_finish(const G1Jac* __restrict__ e, const G1Jac* __restrict__ o, u32 m, u32 log2d, bool natural,
                                                         G1Aff* __restrict__ out) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  out[natural ? brev_bits(k, log2d) : k] = jac_to_aff(jac_add(e[k], o[k]));
}
// the layout switch of TWO arrays in one exchange: the chunk for / from rank q is [part 0 | part 1], c points each
//   pack:   send[(q * 2 + part) * c + t] = arr[part * m + q * c + t]          (arr = [part 0 (m) | part 1 (m)], m = R c)
//   unpack: arr[part * m + t * R + q] = recv[(q * 2 + part) * c + t]          (with the transpose to the block layout)
static __global__ void __launch_bounds__(256) k_fk_pack2(const G1Jac* __restrict__ arr, u32 R, u32 c, G1Jac* __restrict__ send) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * R * c) return;
  const u32 t = i % c, part = (i / c) & 1u, q = i / (2 * c);
  send[i] = arr[(size_t)part * R * c + q * c + t];
}
static __global__ void __launch_bounds__(256) k_fk_unpack2(const G1Jac* __restrict__ recv, u32 R, u32 c, G1Jac* __restrict__ arr) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * R * c) return;
  const u32 t = i % c, part = (i / c) & 1u, q = i / (2 * c);
  arr[(size_t)part * R * c + (size_t)t * R + q] = recv[i];
}
// proofs[bitrev(q)] = gathered[q]
static __global__ void __launch_bounds__(256) k_aff_unscramble(const G1Aff* __restrict__ in, u32 log2d, G1Aff* __restrict__ out) {
  u32 q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (1u << log2d)) return;
  out[log2d ? (__brev(q) >> (32 - log2d)) : 0u] = in[q];
}


// ---- scalar-field (Fr) FFT on the device: row f-4 (the host-side ark-poly work of src/kzg.rs:182-185 and src/vec.rs:36-37) --------
KDEV Fr fr_mul(const Fr& a, const Fr& b) { return fp_mul<FrParams>(a, b); }
// out[k] = omega^k, k < n   (square-and-multiply per lane; n <= 2^27)
static __global__ void __launch_bounds__(256) k_fr_powers(Fr omega, u32 n, Fr* __restrict__ out) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  Fr acc = fp_one<FrParams>(), b = omega;
  for (u32 e = k; e; e >>= 1) {
    if (e & 1u) acc = fr_mul(acc, b);
    b = fr_mul(b, b);
  }
  out[k] = acc;
}
static __global__ void __launch_bounds__(256) k_fr_bitrev(Fr* __restrict__ a, u32 log2n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log2n)) return;
  u32 r = log2n ? (__brev(i) >> (32 - log2n)) : 0u;
  if (i < r) { Fr t = a[i]; a[i] = a[r]; a[r] = t; }
}
// tw[k * tw_stride] = root^k for the transform's own root of order n
static __global__ void __launch_bounds__(256) k_fr_fft_stage(Fr* __restrict__ a, const Fr* __restrict__ tw, u32 tw_stride, u32 n, u32 len) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n / 2) return;
  u32 half = len >> 1;
  u32 j = b % half, i0 = (b / half) * len + j, i1 = i0 + half;
  Fr u = a[i0], v = fr_mul(a[i1], tw[(size_t)j * (n / len) * tw_stride]);
  a[i0] = fp_add<FrParams>(u, v);
  a[i1] = fp_sub<FrParams>(u, v);
}
static __global__ void __launch_bounds__(256) k_fr_scale(Fr* __restrict__ a, Fr s, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = fr_mul(a[i], s);
}
// a2[i] = 0 (i < d), p[i - d] (d <= i < 2d)
static __global__ void __launch_bounds__(256) k_fk_pad(const Fr* __restrict__ p, u32 d, Fr* __restrict__ a2) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * d) return;
  a2[i] = i < d ? fp_zero<FrParams>() : p[i - d];
}
static __global__ void __launch_bounds__(256) k_fr_stride2(const Fr* __restrict__ in, u32 n_out, Fr* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_out) out[i] = in[2 * (size_t)i];
}

// ---- synthetic division by (x - z) on the device: the quotient of `open` (src/kzg.rs:109-120), row f-4 ------------------------
// Q_i = sum_{j >= i} c_j z^(j-i)  (Q_0 = p(z), quotient coefficient q_i = Q_(i+1)) obeys Q_i = c_i + z Q_(i+1): a linear recurrence,
// evaluated blockwise. Level k works on a^k (a^0 = c, a^(k+1) = the block values of level k) with multiplier w_k = z^(L^k):
//   up:   H_b = sum_{t < L} a[bL + t] w^t                       (one lane per block, Horner from the top of the block)
//   top:  all suffix values of the last level by one lane
//   down: a lane re-runs its block from the carry Q_((b+1)L) supplied by the level above and writes every Q of the block
// Block length 32 (256 until round 5): a lane's chain is 2 x 32 steps per level instead of 2 x 256, so a CHUNK of a polynomial (`open` from host
// memory runs the quotient chunk by chunk from the top, api.hip) costs ~0.15 ms of dependent steps instead of ~0.4, and a whole 2^24-coefficient
// quotient keeps eight waves per SIMD busy instead of one.
constexpr u32 HORNER_L = 32;
static __global__ void k_fr_pow_chain(Fr z, u32 levels, Fr* __restrict__ w) {      // w[k] = z^(HORNER_L^k)
  if (threadIdx.x || blockIdx.x) return;
  Fr x = z;
  for (u32 k = 0; k < levels; k++) {
    w[k] = x;
    for (u32 s = 1; s < HORNER_L; s <<= 1) x = fr_mul(x, x);
  }
}
static __global__ void __launch_bounds__(64) k_fr_horner_up(const Fr* __restrict__ a, u32 m, const Fr* __restrict__ wp, Fr* __restrict__ H) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  u32 lo = b * HORNER_L;
  if (lo >= m) return;
  u32 hi = min(m, lo + HORNER_L);
  const Fr w = *wp;
  Fr h = a[hi - 1];
  for (u32 k = hi - 1; k-- > lo;) h = fp_add<FrParams>(fr_mul(h, w), a[k]);
  H[b] = h;
}
// q_up[b] = suffix value at the start of block b of this level (nullptr at the top level: nothing above). out[k - shift] = Q_k;
// with shift = 1 (level 0) Q_0 goes to *value_out instead.
// keep_top: the top `keep_top` outputs are not written (a chunk of a longer polynomial whose top "coefficient" is the carry from the chunk above:
// that slot of the quotient already holds the same value and another stream may be reading it)
static __global__ void __launch_bounds__(64) k_fr_horner_down(const Fr* __restrict__ a, u32 m, const Fr* __restrict__ wp, const Fr* __restrict__ q_up,
                                                              u32 nblocks, Fr* __restrict__ out, u32 shift, Fr* __restrict__ value_out, u32 keep_top) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  u32 lo = b * HORNER_L;
  if (lo >= m) return;
  u32 hi = min(m, lo + HORNER_L);
  const Fr w = *wp;
  Fr carry = (q_up && b + 1 < nblocks) ? q_up[b + 1] : fp_zero<FrParams>();
  for (u32 k = hi; k-- > lo;) {
    carry = fp_add<FrParams>(fr_mul(carry, w), a[k]);
    if (k >= shift) { if (k + keep_top < m) out[k - shift] = carry; }
    else if (value_out) *value_out = carry;
  }
}

}  // namespace bn254

namespace keaki_internal {
using namespace bn254;

// ---- launching the kernels whose lanes run the per-lane-scalar ladder --------------------------------------------------------------------
// Their window tables live in a workspace of the context, one 1 KB slot per lane of a launch (jac29.hip.h: jac_scalar_mul_gtab_u29: eight
// effective-affine entries of 128 bytes): at most FK_TAB_LANES lanes per launch, i.e. 2 GB. When the workspace cannot be had (or option fk_gtab = 0) the kernels keep the table in private
// memory, one launch for all lanes.
constexpr u32 FK_TAB_LANES = 1u << 21;
static uint4* fk_table(keaki_hip_ctx* ctx, u32 lanes) {
  if (!ctx->tune.fk_gtab || lanes == 0) return nullptr;
  const size_t want = (size_t)std::min(lanes, FK_TAB_LANES) * GTAB_UINT4_PER_LANE * sizeof(uint4);
  if (reserve(ctx, ctx->fk_tab, want) != KEAKI_OK) { (void)hipGetLastError(); return nullptr; }       // optional memory
  return (uint4*)ctx->fk_tab.p;
}
// launch(tab, first, lanes_in_this_launch) for every piece of `total` lanes
template <class L>
static void ladder_launches(keaki_hip_ctx* ctx, u32 total, L launch) {
  uint4* tab = fk_table(ctx, total);
  if (!tab) { launch((uint4*)nullptr, 0u, total); return; }
  for (u32 first = 0; first < total; first += FK_TAB_LANES) launch(tab, first, std::min(FK_TAB_LANES, total - first));
}
// the instantiation for (dit, uniform, table workspace, lazy butterflies); uniform never has a table workspace
template <bool DIT, bool UNI, bool GT>
static void launch_stage3(bool as29, dim3 grid, dim3 block, hipStream_t st, G1Jac* a, const Fr* tw, u32 m, u32 half, u32 A, u32 B, u32 stride, uint4* tab, u32 first) {
  if (as29) hipLaunchKernelGGL((k_g1_fft_stage_map<DIT, UNI, GT, true>), grid, block, 0, st, a, tw, m, half, A, B, stride, tab, first);
  else hipLaunchKernelGGL((k_g1_fft_stage_map<DIT, UNI, GT, false>), grid, block, 0, st, a, tw, m, half, A, B, stride, tab, first);
}
static void launch_stage(bool dit, bool uni, bool gt, bool as29, dim3 grid, dim3 block, hipStream_t st, G1Jac* a, const Fr* tw, u32 m, u32 half, u32 A, u32 B,
                         u32 stride, uint4* tab, u32 first) {
  if (uni) {
    if (dit) launch_stage3<true, true, false>(as29, grid, block, st, a, tw, m, half, A, B, stride, tab, first);
    else launch_stage3<false, true, false>(as29, grid, block, st, a, tw, m, half, A, B, stride, tab, first);
  } else if (gt) {
    if (dit) launch_stage3<true, false, true>(as29, grid, block, st, a, tw, m, half, A, B, stride, tab, first);
    else launch_stage3<false, false, true>(as29, grid, block, st, a, tw, m, half, A, B, stride, tab, first);
  } else {
    if (dit) launch_stage3<true, false, false>(as29, grid, block, st, a, tw, m, half, A, B, stride, tab, first);
    else launch_stage3<false, false, false>(as29, grid, block, st, a, tw, m, half, A, B, stride, tab, first);
  }
}
static void stage_map(keaki_hip_ctx* ctx, bool dit, G1Jac* a, const Fr* tw, u32 m, u32 half, u32 A, u32 B, u32 stride) {
  const bool as29 = ctx->tune.fk_addsub29;
  const bool uniform = ctx->tune.fk_uniform && m / (2 * half) >= 64;        // blocks: a power of two, so every 64-lane workgroup then shares one twiddle
  const dim3 block(64);
  if (uniform) {
    const dim3 grid(cdiv(m / 2, 64));
    launch_stage(dit, true, false, as29, grid, block, ctx->stream, a, tw, m, half, A, B, stride, (uint4*)nullptr, 0u);
    return;
  }
  ladder_launches(ctx, m / 2, [&](uint4* tab, u32 first, u32 cnt) {
    const dim3 grid(cdiv(cnt, 64));
    launch_stage(dit, false, tab != nullptr, as29, grid, block, ctx->stream, a, tw, m, half, A, B, stride, tab, first);
  });
}
// The stages of a transform over m local points with spans 2 * half, half = first .. last (doubling for the inverse DIT transform, halving for the
// forward DIF one); stride_of(half) is the twiddle stride of that stage. Two consecutive stages that both have one twiddle per wave (at least 64
// blocks of the LARGER span), none of them the all-trivial span-2 stage, run as one radix-4 pass (option fk_radix4).
template <class StrideOf>
static void run_stages(keaki_hip_ctx* ctx, bool dit, G1Jac* a, const Fr* tw, u32 m, u32 first, u32 last, u32 A, u32 B, StrideOf stride_of) {
  const bool r4 = ctx->tune.fk_radix4 && ctx->tune.fk_uniform;
  // h: the smaller half of the pair. h >= 8: the twiddles of the first stages are short scalars (their ladders skip most of the doublings:
  // 94 K / 198 K / 256 K instructions per wave-butterfly at spans 4 / 8 / 16 against 312 K), a radix-4 pass over them costs MORE than two
  // radix-2 stages (738 K against 584 K per group at h = 2); from h = 8 on it saves 2.4 % .. 7.7 % (bench_tools/pmc_fk_radix4.sh).
  auto pairable = [&](u32 h) { return r4 && h >= 8 && m / (4 * h) >= 64 && (m / 4) % 64 == 0; };
  if (dit) {
    for (u32 half = first; half <= last;) {
      if (2 * half <= last && pairable(half)) {
        if (ctx->tune.fk_addsub29) hipLaunchKernelGGL((k_g1_fft_stage4<true, true>), dim3(cdiv(m / 4, 64)), dim3(64), 0, ctx->stream, a, tw, m, half, A, B, stride_of(half), stride_of(2 * half));
        else hipLaunchKernelGGL((k_g1_fft_stage4<true, false>), dim3(cdiv(m / 4, 64)), dim3(64), 0, ctx->stream, a, tw, m, half, A, B, stride_of(half), stride_of(2 * half));
        half <<= 2;
      } else {
        stage_map(ctx, true, a, tw, m, half, A, B, stride_of(half));
        half <<= 1;
      }
    }
  } else {
    for (u32 half = first; half >= last && half >= 1;) {
      const u32 h = half / 2;                       // the pair: spans 2 * half (first) and 2 * h
      if (h >= last && h >= 1 && pairable(h)) {
        if (ctx->tune.fk_addsub29) hipLaunchKernelGGL((k_g1_fft_stage4<false, true>), dim3(cdiv(m / 4, 64)), dim3(64), 0, ctx->stream, a, tw, m, h, A, B, stride_of(h), stride_of(half));
        else hipLaunchKernelGGL((k_g1_fft_stage4<false, false>), dim3(cdiv(m / 4, 64)), dim3(64), 0, ctx->stream, a, tw, m, h, A, B, stride_of(h), stride_of(half));
        half >>= 2;
      } else {
        stage_map(ctx, false, a, tw, m, half, A, B, stride_of(half));
        half >>= 1;
      }
      if (half == 0) break;
    }
  }
}
static void launch_pointwise(keaki_hip_ctx* ctx, const G1Jac* hs_even, const G1Jac* hs_odd, const Fr* a, u32 log2d, u32 base, u32 m, G1Jac* out_e, G1Jac* out_o) {
  ladder_launches(ctx, 2 * m, [&](uint4* tab, u32 first, u32 cnt) {
    if (tab) hipLaunchKernelGGL(k_fk_pointwise<true>, dim3(cdiv(cnt, 64)), dim3(64), 0, ctx->stream, hs_even, hs_odd, a, log2d, base, m, out_e, out_o, tab, first);
    else hipLaunchKernelGGL(k_fk_pointwise<false>, dim3(cdiv(cnt, 64)), dim3(64), 0, ctx->stream, hs_even, hs_odd, a, log2d, base, m, out_e, out_o, tab, first);
  });
}
static void launch_mul_strided(keaki_hip_ctx* ctx, G1Jac* a, const Fr* s, u32 stride, u32 offset, u32 m) {
  ladder_launches(ctx, m, [&](uint4* tab, u32 first, u32 cnt) {
    if (tab) hipLaunchKernelGGL(k_g1_mul_jac_strided<true>, dim3(cdiv(cnt, 64)), dim3(64), 0, ctx->stream, a, s, stride, offset, m, tab, first);
    else hipLaunchKernelGGL(k_g1_mul_jac_strided<false>, dim3(cdiv(cnt, 64)), dim3(64), 0, ctx->stream, a, s, stride, offset, m, tab, first);
  });
}
static void launch_mul_strided_oop(keaki_hip_ctx* ctx, const G1Jac* in, const Fr* s, u32 stride, u32 offset, u32 m, G1Jac* out) {
  ladder_launches(ctx, m, [&](uint4* tab, u32 first, u32 cnt) {
    if (tab) hipLaunchKernelGGL(k_g1_mul_jac_strided_oop<true>, dim3(cdiv(cnt, 64)), dim3(64), 0, ctx->stream, in, s, stride, offset, m, out, tab, first);
    else hipLaunchKernelGGL(k_g1_mul_jac_strided_oop<false>, dim3(cdiv(cnt, 64)), dim3(64), 0, ctx->stream, in, s, stride, offset, m, out, tab, first);
  });
}

// hat_s = DIF_2d(reversed SRS padded with identities): depends on the SRS only, so it is computed once per (SRS, d) and cached.
// d_tw2d: omega_2d^k, k < d.
keaki_status fk_hat_s_run(keaki_hip_ctx* ctx, const void* d_srs, u32 log2d, const void* d_tw2d, void* d_hat_s) {
  const u32 d = 1u << log2d, N = 2 * d;
  G1Jac* s = (G1Jac*)d_hat_s;
  hipLaunchKernelGGL(k_fk_load, dim3(cdiv(N, 256)), dim3(256), 0, ctx->stream, (const G1Aff*)d_srs, d, s);
  run_stages(ctx, false, s, (const Fr*)d_tw2d, N, d, 1u, 1, 0, [&](u32 half) { return N / (2 * half); });
  return launch_check(ctx, "fk_hat_s");
}
// d = 2^log2d openings from the cached hat_s. d_work: 2d Jacobian points. d_hat_a: 2d Fr, natural order, already divided by 2d.
// d_tw2d: omega_2d^k, d_tw2d_inv: omega_2d^-k (k < d). Output: d affine proofs, natural order.
keaki_status open_fk_run(keaki_hip_ctx* ctx, const void* d_hat_s, u32 log2d, const void* d_hat_a, const void* d_tw2d, const void* d_tw2d_inv,
                         void* d_work, void* d_proofs_aff) {
  const u32 d = 1u << log2d;
  hipStream_t st = ctx->stream;
  const G1Jac* hs = (const G1Jac*)d_hat_s;
  G1Jac *e = (G1Jac*)d_work, *o = e + d;
  const Fr *tw = (const Fr*)d_tw2d, *twi = (const Fr*)d_tw2d_inv;
  const bool timed = ctx->timing && ctx->fk_ev[0];
  if (timed) (void)hipEventRecord(ctx->fk_ev[0], st);
  launch_pointwise(ctx, hs, hs + d, (const Fr*)d_hat_a, log2d, 0u, d, e, o);
  if (timed) (void)hipEventRecord(ctx->fk_ev[1], st);
  run_stages(ctx, true, o, twi, d, 1u, d / 2, 1, 0, [&](u32 half) { return 2 * (d / (2 * half)); });
  launch_mul_strided(ctx, o, twi, 1u, 0u, d);
  run_stages(ctx, false, o, tw, d, d / 2, 1u, 1, 0, [&](u32 half) { return 2 * (d / (2 * half)); });
  if (timed) (void)hipEventRecord(ctx->fk_ev[2], st);
  hipLaunchKernelGGL(k_fk_finish, dim3(cdiv(d, 64)), dim3(64), 0, st, (const G1Jac*)e, (const G1Jac*)o, d, log2d, true, (G1Aff*)d_proofs_aff);
  if (timed) { (void)hipEventRecord(ctx->fk_ev[3], st); ctx->fk_timing_pending = true; }
  return launch_check(ctx, "open_fk");
}


// in-place DFT of n = 2^log2n Fr elements with the given twiddle table (tw[k * tw_stride] = root^k, k < n/2)
static keaki_status fr_fft(keaki_hip_ctx* ctx, Fr* a, u32 log2n, const Fr* tw, u32 tw_stride) {
  const u32 n = 1u << log2n;
  hipLaunchKernelGGL(k_fr_bitrev, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, a, log2n);
  for (u32 len = 2; len <= n; len <<= 1)
    hipLaunchKernelGGL(k_fr_fft_stage, dim3(cdiv(n / 2 ? n / 2 : 1, 256)), dim3(256), 0, ctx->stream, a, tw, tw_stride, n, len);
  return launch_check(ctx, "fr_fft");
}
// generic scalar-field transform: d_data (n Fr) <- DFT with root `omega` (order n), then optionally * scale. d_tw: scratch for n/2 Fr.
keaki_status fr_fft_run(keaki_hip_ctx* ctx, void* d_data, u32 log2n, const uint64_t* omega, const uint64_t* scale_or_null, void* d_tw) {
  const u32 n = 1u << log2n;
  Fr w, s;
  memcpy(&w, omega, 32);
  if (n >= 2) {
    hipLaunchKernelGGL(k_fr_powers, dim3(cdiv(n / 2, 256)), dim3(256), 0, ctx->stream, w, n / 2, (Fr*)d_tw);
    ST_TRY(fr_fft(ctx, (Fr*)d_data, log2n, (const Fr*)d_tw, 1));
  }
  if (scale_or_null) {
    memcpy(&s, scale_or_null, 32);
    hipLaunchKernelGGL(k_fr_scale, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (Fr*)d_data, s, n);
  }
  return launch_check(ctx, "fr_fft_run");
}
// FK23 from the polynomial itself: everything (twiddles, hat_a) is derived on the device from three scalars.
// d_p: d Fr coefficients. d_fr_work: room for (2d + d + d + 1) Fr. d_g_work: 2d Jacobian points. Output: d affine proofs.
keaki_status open_fk_poly_run(keaki_hip_ctx* ctx, const void* d_srs, void** hat_s_cache, int* hat_s_log2d, u32 log2d, const void* d_p,
                              const uint64_t* omega_2d, const uint64_t* omega_2d_inv, const uint64_t* inv_2d, void* d_fr_work, void* d_g_work,
                              void* d_proofs_aff) {
  const u32 d = 1u << log2d;
  Fr w, wi, s;
  memcpy(&w, omega_2d, 32); memcpy(&wi, omega_2d_inv, 32); memcpy(&s, inv_2d, 32);
  Fr* hat_a = (Fr*)d_fr_work;
  Fr* tw = hat_a + 2 * (size_t)d;       // omega_2d^k,  k < d
  Fr* twi = tw + d;                      // omega_2d^-k, k < d
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_fr_powers, dim3(cdiv(d, 256)), dim3(256), 0, st, w, d, tw);
  hipLaunchKernelGGL(k_fr_powers, dim3(cdiv(d, 256)), dim3(256), 0, st, wi, d, twi);
  hipLaunchKernelGGL(k_fk_pad, dim3(cdiv(2 * d, 256)), dim3(256), 0, st, (const Fr*)d_p, d, hat_a);
  ST_TRY(fr_fft(ctx, hat_a, log2d + 1, tw, 1));
  hipLaunchKernelGGL(k_fr_scale, dim3(cdiv(2 * d, 256)), dim3(256), 0, st, hat_a, s, 2 * d);
  if (*hat_s_log2d != (int)log2d) {
    if (*hat_s_cache) { HIP_TRY(ctx, hipStreamSynchronize(st)); (void)hipFree(*hat_s_cache); *hat_s_cache = nullptr; *hat_s_log2d = -1; }
    HIP_TRY(ctx, hipMalloc(hat_s_cache, 2 * (size_t)d * sizeof(G1Jac)));
    ST_TRY(fk_hat_s_run(ctx, d_srs, log2d, tw, *hat_s_cache));
    *hat_s_log2d = (int)log2d;
  }
  return open_fk_run(ctx, *hat_s_cache, log2d, hat_a, tw, twi, d_g_work, d_proofs_aff);
}

// ---- FK23 sharded (keaki_hip_fk_shard_*): the steps between the caller's exchanges; tests/fk_shard_model.py::ShardModel step for step -----
static void fk_shard_tables(keaki_hip_ctx* ctx, FkShard& fk) {
  if (fk.tables_ready) return;
  const u32 d = 1u << fk.log2d;
  Fr w, wi;
  memcpy(&w, fk.omega, 32); memcpy(&wi, fk.omega_inv, 32);
  hipLaunchKernelGGL(k_fr_powers, dim3(cdiv(d, 256)), dim3(256), 0, ctx->stream, w, d, (Fr*)fk.tw);
  hipLaunchKernelGGL(k_fr_powers, dim3(cdiv(d, 256)), dim3(256), 0, ctx->stream, wi, d, (Fr*)fk.twi);
  fk.tables_ready = true;
}
// one size-d transform, forward (decimation in frequency), distributed: spans d..2R on the cyclic layout, spans R..2 on the block layout
static void dif_cyclic(keaki_hip_ctx* ctx, G1Jac* a, const Fr* tw, u32 d, u32 R, u32 r) {
  const u32 Md = d / R;
  run_stages(ctx, false, a, tw, Md, Md / 2, 1u, R, r, [&](u32 half) { return 2 * (d / (2 * half * R)); });
}
static void dif_block(keaki_hip_ctx* ctx, G1Jac* a, u32 m, const Fr* tw, u32 d, u32 R) {      // m: local points (a multiple of R)
  if (R >= 2) run_stages(ctx, false, a, tw, m, R / 2, 1u, 1, 0, [&](u32 half) { return 2 * (d / (2 * half)); });
}
// hat_s of this rank: even entries = DFT_d(S), odd entries = DFT_d(S_i omega_2d^i), both at bit-reversed positions, block layout.
// step 0: the cyclic slices, spans d..2R, packed [even | odd] per peer -> d_send; step 1: d_recv -> block layout, spans R..2 -> fk.hat_s.
keaki_status fk_shard_setup_run(keaki_hip_ctx* ctx, FkShard& fk, const void* d_srs, int step, void* d_send, void* d_recv) {
  const u32 d = 1u << fk.log2d, R = 1u << fk.rho, Md = d / R, r = fk.rank;
  hipStream_t st = ctx->stream;
  const Fr* tw = (const Fr*)fk.tw;
  if (step == 0) {
    fk_shard_tables(ctx, fk);
    G1Jac *ev = (G1Jac*)fk.work, *od = ev + Md;
    hipLaunchKernelGGL(k_fk_load_cyclic, dim3(cdiv(Md, 256)), dim3(256), 0, st, (const G1Aff*)d_srs, d, R, r, Md, ev);      // all of them below d
    launch_mul_strided_oop(ctx, (const G1Jac*)ev, tw, R, r, Md, od);
    dif_cyclic(ctx, ev, tw, d, R, r);
    dif_cyclic(ctx, od, tw, d, R, r);
    hipLaunchKernelGGL(k_fk_pack2, dim3(cdiv(2 * Md, 256)), dim3(256), 0, st, (const G1Jac*)ev, R, Md / R, (G1Jac*)d_send);
    return launch_check(ctx, "fk_shard_setup 0");
  }
  G1Jac* hs = (G1Jac*)fk.hat_s;                 // [even (Md) | odd (Md)]
  hipLaunchKernelGGL(k_fk_unpack2, dim3(cdiv(2 * Md, 256)), dim3(256), 0, st, (const G1Jac*)d_recv, R, Md / R, hs);
  dif_block(ctx, hs, 2 * Md, tw, d, R);
  fk.hat_s_ready = true;
  return launch_check(ctx, "fk_shard_setup 1");
}
// the openings. step 0: hat_a (replicated scalar-field work), the 2 d/R products of this rank's positions (E kept in fk.e), inverse
// transform spans 2..d/R (block) -> d_send packed for the switch to cyclic; step 1: d_recv = cyclic layout, spans 2d/R..d, the twist,
// forward transform spans d..2R -> d_send; step 2: d_recv -> block layout, spans R..2, + E, to affine -> d_send (d/R affine points:
// positions [rank d/R, ..) of the bit-reversed proof order); step 3: d_recv = all d affine points in position order -> d_out_aff natural.
keaki_status fk_shard_open_run(keaki_hip_ctx* ctx, FkShard& fk, int step, void* d_send, void* d_recv, void* d_out_aff) {
  const u32 d = 1u << fk.log2d, N = 2 * d, R = 1u << fk.rho, Md = d / R, r = fk.rank;
  hipStream_t st = ctx->stream;
  const Fr *tw = (const Fr*)fk.tw, *twi = (const Fr*)fk.twi;
  if (step == 0) {
    Fr s;
    memcpy(&s, fk.inv_2d, 32);
    Fr* hat_a = (Fr*)fk.hat_a;
    hipLaunchKernelGGL(k_fk_pad, dim3(cdiv(N, 256)), dim3(256), 0, st, (const Fr*)fk.coeffs, d, hat_a);
    ST_TRY(fr_fft(ctx, hat_a, fk.log2d + 1, tw, 1));
    hipLaunchKernelGGL(k_fr_scale, dim3(cdiv(N, 256)), dim3(256), 0, st, hat_a, s, N);
    const G1Jac* hs = (const G1Jac*)fk.hat_s;
    G1Jac* o = (G1Jac*)fk.work;
    launch_pointwise(ctx, hs, hs + Md, (const Fr*)hat_a, fk.log2d, r * Md, Md, (G1Jac*)fk.e, o);
    if (Md >= 2) run_stages(ctx, true, o, twi, Md, 1u, Md / 2, 1, 0, [&](u32 half) { return 2 * (d / (2 * half)); });
    hipLaunchKernelGGL(k_jac_transpose, dim3(cdiv(Md, 256)), dim3(256), 0, st, (const G1Jac*)o, Md / R, R, (G1Jac*)d_send);
    return launch_check(ctx, "fk_shard_open 0");
  }
  if (step == 1) {
    G1Jac* a = (G1Jac*)d_recv;
    if (2 * (Md / R) <= Md) run_stages(ctx, true, a, twi, Md, Md / R, Md / 2, R, r, [&](u32 half) { return 2 * (d / (2 * half * R)); });
    launch_mul_strided(ctx, a, twi, R, r, Md);
    dif_cyclic(ctx, a, tw, d, R, r);
    HIP_TRY(ctx, hipMemcpyAsync(d_send, a, (size_t)Md * sizeof(G1Jac), hipMemcpyDeviceToDevice, st));
    return launch_check(ctx, "fk_shard_open 1");
  }
  if (step == 2) {
    G1Jac* a = (G1Jac*)fk.work;
    hipLaunchKernelGGL(k_jac_transpose, dim3(cdiv(Md, 256)), dim3(256), 0, st, (const G1Jac*)d_recv, R, Md / R, a);
    dif_block(ctx, a, Md, tw, d, R);
    hipLaunchKernelGGL(k_fk_finish, dim3(cdiv(Md, 64)), dim3(64), 0, st, (const G1Jac*)fk.e, (const G1Jac*)a, Md, fk.log2d, false, (G1Aff*)d_send);
    return launch_check(ctx, "fk_shard_open 2");
  }
  hipLaunchKernelGGL(k_aff_unscramble, dim3(cdiv(d, 256)), dim3(256), 0, st, (const G1Aff*)d_recv, fk.log2d, (G1Aff*)d_out_aff);
  return launch_check(ctx, "fk_shard_open 3");
}

// hat_s = DFT_2d(reversed SRS) ahead of time (it depends on the SRS and d only): setup-time work like the MSM window tables
keaki_status fk_precompute_run(keaki_hip_ctx* ctx, const void* d_srs, void** hat_s_cache, int* hat_s_log2d, u32 log2d, const uint64_t* omega_2d,
                               void* d_tw_work) {
  const u32 d = 1u << log2d;
  if (*hat_s_log2d == (int)log2d) return KEAKI_OK;
  Fr w;
  memcpy(&w, omega_2d, 32);
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_fr_powers, dim3(cdiv(d, 256)), dim3(256), 0, st, w, d, (Fr*)d_tw_work);
  if (*hat_s_cache) { HIP_TRY(ctx, hipStreamSynchronize(st)); (void)hipFree(*hat_s_cache); *hat_s_cache = nullptr; *hat_s_log2d = -1; }
  HIP_TRY(ctx, hipMalloc(hat_s_cache, 2 * (size_t)d * sizeof(G1Jac)));
  ST_TRY(fk_hat_s_run(ctx, d_srs, log2d, d_tw_work, *hat_s_cache));
  *hat_s_log2d = (int)log2d;
  return KEAKI_OK;
}

// d_c: n coefficients (Fr). d_q: n - 1 quotient coefficients out (n >= 1; n == 1: nothing written). d_value: p(z) out (1 Fr).
// d_work: open_quotient_work_bytes(n).
size_t open_quotient_work_bytes(size_t n) { return (2 * (n / (HORNER_L - 1) + 16) + 8) * sizeof(Fr); }
keaki_status open_quotient_run(keaki_hip_ctx* ctx, const void* d_c, size_t n, const uint64_t* z, void* d_q, void* d_value, void* d_work, bool top_is_carry) {
  hipStream_t st = ctx->stream;
  Fr zz;
  memcpy(&zz, z, 32);
  // level sizes
  u32 m[8];
  u32 levels = 0;
  for (size_t cur = n;; cur = (cur + HORNER_L - 1) / HORNER_L) { m[levels++] = (u32)cur; if (cur <= HORNER_L || levels == 8) break; }
  Fr* w = (Fr*)d_work;                       // w[k], k < levels
  Fr* base = w + 8;
  // a[k]: level arrays (a[0] = coefficients), Q[k]: suffix values of level k >= 1
  const Fr* a[8]; Fr* Q[8];
  a[0] = (const Fr*)d_c; Q[0] = nullptr;
  Fr* p = base;
  for (u32 k = 1; k < levels; k++) { a[k] = p; p += m[k]; }
  for (u32 k = 1; k < levels; k++) { Q[k] = p; p += m[k]; }
  hipLaunchKernelGGL(k_fr_pow_chain, dim3(1), dim3(1), 0, st, zz, levels, w);
  for (u32 k = 0; k + 1 < levels; k++)
    hipLaunchKernelGGL(k_fr_horner_up, dim3(cdiv(m[k + 1], 64)), dim3(64), 0, st, a[k], m[k], (const Fr*)(w + k), (Fr*)a[k + 1]);
  for (u32 k = levels; k-- > 0;) {
    const Fr* q_up = (k + 1 < levels) ? Q[k + 1] : nullptr;
    const u32 nblocks = cdiv(m[k], HORNER_L);
    if (k == 0)
      hipLaunchKernelGGL(k_fr_horner_down, dim3(cdiv(nblocks, 64)), dim3(64), 0, st, a[0], m[0], (const Fr*)w, q_up, nblocks, (Fr*)d_q, 1u, (Fr*)d_value, top_is_carry ? 1u : 0u);
    else
      hipLaunchKernelGGL(k_fr_horner_down, dim3(cdiv(nblocks, 64)), dim3(64), 0, st, a[k], m[k], (const Fr*)(w + k), q_up, nblocks, Q[k], 0u, (Fr*)nullptr, 0u);
  }
  return launch_check(ctx, "open_quotient");
}

}  // namespace keaki_internal
