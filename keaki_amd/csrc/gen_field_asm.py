#!/usr/bin/env python3
"""Emits keaki_amd/csrc/bn254_field_asm.hip.h: hand-scheduled gfx950 instruction streams for the BN254
base-field (Fq) Montgomery product, addition, subtraction and negation on 8 x 32-bit limbs.

Why hand-written: on gfx950 v_mad_u64_u32 issues at the same rate as a 32-bit add (measured:
profiles/r01_ubench_int_gfx950.txt), so the cost of a modular product is its instruction count.
hipcc's schedule of the portable CIOS loop spends ~500 issues per product (it assembles every 64-bit
addend pair with v_mov and adds carries with v_lshl_add_u64) and ~55 per modular add (it does not
form a carry chain at all). The streams below need ~300 and 24.

Montgomery product = product scanning (Comba) with the reduction interleaved:
  column k accumulates  sum_{i+j=k} a_i b_j + sum_{i+j=k, i<k} m_i p_j  in one 64-bit VGPR pair with
  v_mad_u64_u32 (carry-out to VCC) and counts the pair's overflows with v_addc_co_u32 in a third
  word; m_k = lo * (-p^-1 mod 2^32) makes the low word vanish; then the accumulator slides down one
  limb. p < 2^254, so the result is < 2p and one conditional subtraction finishes.
  One asm statement per column (the halves of a 64-bit asm operand cannot be named inside the
  string, so m_k and the slide are plain C++ between statements: sub-register reads, no code).

Hazards (hipcc pads nothing inside an asm string): carries travel through VCC and are consumed only
by VOP2 (_e32) forms, whose VCC read is implicit (it still occupies the single constant-bus slot, so the
modulus limbs of the carry chains are VGPR operands); gfx940+ needs two wait states between a VALU SGPR
write and an EXPLICIT SGPR operand read, which these streams never do. VGPR RAW is interlocked.

    python keaki_amd/csrc/gen_field_asm.py > keaki_amd/csrc/bn254_field_asm.hip.h
"""

SEP = "\\n\\t"


def column_stmt(k):
    """asm statement adding all products of column k (except m_k p_0) into (acc, ovf)."""
    ab = [(i, k - i) for i in range(8) if 0 <= k - i <= 7]
    mp = [(i, k - i) for i in range(8) if 0 <= k - i <= 7 and i < min(k, 8) and k - i >= 1] if k < 8 else \
         [(i, k - i) for i in range(8) if 0 <= k - i <= 7]
    # operands: %0 acc (u64 +v), %1 ovf (+v), then inputs
    ins = []
    def ref(kind, idx):
        key = (kind, idx)
        if key not in [x[0] for x in ins]:
            if kind == "a": ins.append((key, '"v"(a[%d])' % idx))
            elif kind == "b": ins.append((key, '"v"(b[%d])' % idx))
            elif kind == "m": ins.append((key, '"v"(m[%d])' % idx))
            elif kind == "p": ins.append((key, '"s"(FqParams::MOD[%d])' % idx))
        return "%%%d" % (2 + [x[0] for x in ins].index(key))
    lines = []
    first = True
    for (i, j) in ab:
        ra, rb = ref("a", i), ref("b", j)
        if k == 0 and first:
            lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, 0" % (ra, rb))   # nothing to overflow yet
        else:
            lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, %%0" % (ra, rb))
            lines.append("v_addc_co_u32_e32 %1, vcc, 0, %1, vcc")
        first = False
    for (i, j) in mp:
        rm, rp = ref("m", i), ref("p", j)
        lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, %%0" % (rm, rp))
        lines.append("v_addc_co_u32_e32 %1, vcc, 0, %1, vcc")
    acc_c = '"=&v"(acc)' if k == 0 else '"+v"(acc)'
    body = SEP.join(lines)
    return '  asm("%s"\n      : %s, "+v"(ovf)\n      : %s\n      : "vcc");' % (body, acc_c, ", ".join(x[1] for x in ins))


def emit_mul():
    print("// r = a * b * R^-1 mod p   (a, b < p; r < p). r may alias a or b only through copies (arrays are by value).")
    print("KDEV void fq_mul_asm(u32* __restrict__ r, const u32* __restrict__ a, const u32* __restrict__ b) {")
    print("  u64 acc;\n  u32 ovf = 0;\n  u32 m[8];")
    for k in range(8):
        print("  // ---- column %d" % k)
        print(column_stmt(k))
        print("  m[%d] = (u32)acc * FqParams::INV;" % k)
        print('  asm("v_mad_u64_u32 %%0, vcc, %%2, %%3, %%0%sv_addc_co_u32_e32 %%1, vcc, 0, %%1, vcc" : "+v"(acc), "+v"(ovf) : "v"(m[%d]), "s"(FqParams::MOD[0]) : "vcc");' % (SEP, k))
        print("  acc = (acc >> 32) | ((u64)ovf << 32);\n  ovf = 0;")
    print("  u32 t[8];")
    for k in range(8, 15):
        print("  // ---- column %d" % k)
        print(column_stmt(k))
        print("  t[%d] = (u32)acc;" % (k - 8))
        print("  acc = (acc >> 32) | ((u64)ovf << 32);\n  ovf = 0;")
    print("  t[7] = (u32)acc;")
    print("  fq_cond_sub_p_asm(r, t);")
    print("}")


def emit_cond_sub():
    print("// r = t - p if t >= p else t   (t < 2p)")
    print("KDEV void fq_cond_sub_p_asm(u32* __restrict__ r, const u32* __restrict__ t) {")
    lines = ["v_subrev_co_u32_e32 %0, vcc, %16, %8"]
    for j in range(1, 8):
        lines.append("v_subbrev_co_u32_e32 %%%d, vcc, %%%d, %%%d, vcc" % (j, 16 + j, 8 + j))
    for j in range(8):
        lines.append("v_cndmask_b32_e32 %%%d, %%%d, %%%d, vcc" % (j, j, 8 + j))   # borrow -> keep t
    outs = ", ".join('"=&v"(r[%d])' % j for j in range(8))
    ins = ", ".join('"v"(t[%d])' % j for j in range(8)) + ", " + ", ".join('"v"(FqParams::MOD[%d])' % j for j in range(8))
    print('  asm("%s"\n      : %s\n      : %s\n      : "vcc");' % (SEP.join(lines), outs, ins))
    print("}")


def emit_add():
    print("// r = a + b mod p   (one statement: the sum, the trial subtraction and the selection; a + b < 2p < 2^255, no carry out of 256 bits)")
    print("KDEV void fq_add_asm(u32* __restrict__ r, const u32* __restrict__ a, const u32* __restrict__ b) {")
    print("  u32 t[8];")
    lines = ["v_add_co_u32_e32 %8, vcc, %16, %24"]
    for j in range(1, 8):
        lines.append("v_addc_co_u32_e32 %%%d, vcc, %%%d, %%%d, vcc" % (8 + j, 16 + j, 24 + j))
    lines.append("v_subrev_co_u32_e32 %0, vcc, %32, %8")
    for j in range(1, 8):
        lines.append("v_subbrev_co_u32_e32 %%%d, vcc, %%%d, %%%d, vcc" % (j, 32 + j, 8 + j))
    for j in range(8):
        lines.append("v_cndmask_b32_e32 %%%d, %%%d, %%%d, vcc" % (j, j, 8 + j))   # borrow -> keep the sum
    outs = ", ".join('"=&v"(r[%d])' % j for j in range(8)) + ", " + ", ".join('"=&v"(t[%d])' % j for j in range(8))
    ins = ", ".join('"v"(a[%d])' % j for j in range(8)) + ", " + ", ".join('"v"(b[%d])' % j for j in range(8)) + ", " + \
        ", ".join('"v"(FqParams::MOD[%d])' % j for j in range(8))
    print('  asm("%s"\n      : %s\n      : %s\n      : "vcc");' % (SEP.join(lines), outs, ins))
    print("}")


def emit_sub():
    print("// r = a - b mod p")
    print("KDEV void fq_sub_asm(u32* __restrict__ r, const u32* __restrict__ a, const u32* __restrict__ b) {")
    # d = a - b (borrow in vcc); save borrow in an SGPR pair via s_mov (SALU write: no VALU hazard);
    # e = d + p; r = borrow ? e : d
    lines = ["v_sub_co_u32_e32 %0, vcc, %17, %25"]
    for j in range(1, 8):
        lines.append("v_subb_co_u32_e32 %%%d, vcc, %%%d, %%%d, vcc" % (j, 17 + j, 25 + j))
    lines.append("s_mov_b64 %16, vcc")
    lines.append("v_add_co_u32_e32 %8, vcc, %33, %0")
    for j in range(1, 8):
        lines.append("v_addc_co_u32_e32 %%%d, vcc, %%%d, %%%d, vcc" % (8 + j, 33 + j, j))
    # SALU-written SGPR read by VALU as an explicit operand: no wait states required
    for j in range(8):
        lines.append("v_cndmask_b32_e64 %%%d, %%%d, %%%d, %%16" % (j, j, 8 + j))
    outs = ", ".join('"=&v"(r[%d])' % j for j in range(8)) + ", " + ", ".join('"=&v"(e[%d])' % j for j in range(8)) + ', "=&s"(bw)'
    ins = ", ".join('"v"(a[%d])' % j for j in range(8)) + ", " + ", ".join('"v"(b[%d])' % j for j in range(8)) + ", " + \
        ", ".join('"v"(FqParams::MOD[%d])' % j for j in range(8))
    print("  u32 e[8];\n  u64 bw;")
    print('  asm("%s"\n      : %s\n      : %s\n      : "vcc");' % (SEP.join(lines), outs, ins))
    print("}")


def emit_neg():
    print("// r = -a mod p  (0 -> 0)")
    print("KDEV void fq_neg_asm(u32* __restrict__ r, const u32* __restrict__ a) {")
    lines = ["v_sub_co_u32_e32 %0, vcc, %16, %8"]
    for j in range(1, 8):
        lines.append("v_subb_co_u32_e32 %%%d, vcc, %%%d, %%%d, vcc" % (j, 16 + j, 8 + j))
    outs = ", ".join('"=&v"(t[%d])' % j for j in range(8))
    ins = ", ".join('"v"(a[%d])' % j for j in range(8)) + ", " + ", ".join('"v"(FqParams::MOD[%d])' % j for j in range(8))
    print("  u32 t[8];")
    print('  asm("%s"\n      : %s\n      : %s\n      : "vcc");' % (SEP.join(lines), outs, ins))
    print("  u32 nz = a[0] | a[1] | a[2] | a[3] | a[4] | a[5] | a[6] | a[7];")
    print("#pragma unroll\n  for (int j = 0; j < 8; j++) r[j] = nz ? t[j] : 0u;")
    print("}")


def emit_redc(params, name):
    """x / 2^256 mod the modulus of `params` (Montgomery -> canonical integer, ark-ff into_bigint), ONE asm block: the MSM's digit
    extraction runs it once per scalar per pass and hipcc's schedule of the portable loop is ~450 instructions (250 of them v_mov).
    Accumulator v[52:53] + overflow counter v54: fixed registers of a caller-saved block, declared as clobbers."""
    ACC, LO, HI, OVF = "v[52:53]", "v52", "v53", "v54"
    ops = []
    idx = {}
    def op(key, cons, expr):
        if key not in idx:
            idx[key] = len(ops); ops.append((cons, expr))
        return "%%%d" % idx[key]
    T = [op(("t", i), '"=&v"', "t[%d]" % i) for i in range(8)]       # m_k lives in t[k] until result limb k is produced (column k + 8)
    nout = len(ops)
    X = [op(("x", i), '"v"', "x[%d]" % i) for i in range(8)]
    P = [op(("p", i), '"s"', "%s::MOD[%d]" % (params, i)) for i in range(8)]
    INV = op(("inv",), '"s"', "%s::INV" % params)
    L = ["v_mov_b32 %s, 0" % LO, "v_mov_b32 %s, 0" % HI, "v_mov_b32 %s, 0" % OVF]
    def mad(a, b):
        L.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, a, b, ACC))
        L.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (OVF, OVF))
    def slide():
        L.append("v_mov_b32 %s, %s" % (LO, HI)); L.append("v_mov_b32 %s, %s" % (HI, OVF)); L.append("v_mov_b32 %s, 0" % OVF)
    for k in range(8):
        L.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (LO, X[k], LO))          # acc < 2^34 after a slide: cannot leave 64 bits
        L.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (HI, HI))
        for i in range(k):
            mad(T[i], P[k - i])
        L.append("v_mul_lo_u32 %s, %s, %s" % (T[k], LO, INV))
        mad(T[k], P[0])
        slide()
    res = []
    for k in range(8, 15):
        for i in range(k - 7, 8):
            mad(T[i], P[k - i])
        # result limb k - 8 may overwrite m_(k-8): last read in column k - 1 ... but m_(k-8) p_(8)? no: p has 8 limbs, m_i is read up to column i + 7
        L.append("v_mov_b32 %s, %s" % (T[k - 8], LO))
        slide()
    L.append("v_mov_b32 %s, %s" % (T[7], LO))
    body = "\\n\\t".join(L)
    print("// r = x / 2^256 mod p(%s), canonical. %d instructions." % (params, len(L) + 16))
    print("KDEV void %s(u32* __restrict__ r, const u32* __restrict__ x) {" % name)
    print("  u32 t[8];")
    print('  asm("%s"\n      : %s\n      : %s\n      : "vcc", "%s", "%s", "%s");' % (body, ", ".join("%s(%s)" % o for o in ops[:nout]),
          ", ".join("%s(%s)" % o for o in ops[nout:]), LO, HI, OVF))
    print("  fp_reduce_once<%s>(t);" % params)
    print("#pragma unroll\n  for (int j = 0; j < 8; j++) r[j] = t[j];")
    print("}")


def main():
    print("// GENERATED by keaki_amd/csrc/gen_field_asm.py -- do not edit.")
    print("// Hand-scheduled gfx950 streams for Fq (BN254 base field): see the generator's header for the design.")
    print("#pragma once")
    print("namespace bn254 {")
    emit_cond_sub()
    emit_mul()
    emit_add()
    emit_sub()
    emit_neg()
    emit_redc("FrParams", "fr_from_mont_asm")
    print("}  // namespace bn254")


if __name__ == "__main__":
    main()
