#!/usr/bin/env python3
"""Emits keaki_amd/csrc/fq29_asm.hip.h: gfx950 instruction streams for the 9 x 29-bit lazy Montgomery product and square
(radix 2^261) of fq29.hip.h.

ONE asm statement per product: 162 (square: 126) v_mad_u64_u32 in column order into a single 64-bit accumulator, per column
one v_lshrrev_b64 (slide), for the low nine columns v_mul_lo_u32 + v_and (m_k), for the high eight one v_and (result limb):
205 (square: 177) instructions. Columns never overflow 64 bits (fq29.hip.h), so the carry-out that v_mad_u64_u32 must name
(VCC) is dead.

Why one statement: hipcc's own schedule of the portable loops splits every column into two chains joined by 64-bit adds (289
instructions); and with one statement per column it puts a wait state behind every statement (on gfx940+ it must assume that
an asm block ends in a partial-register write, the "dst_sel forwarding" hazard) -- 24 dead issue slots per product. The halves
of a 64-bit asm operand cannot be named inside the string, so the accumulator is a fixed VGPR pair declared as clobbered
(v[54:55]: inside the allocation of every kernel that uses these streams, so it costs no occupancy, and in a
caller-saved block of the AMDGPU calling convention, so a non-inlined function that contains a stream -- fq2d_mul in pairing.hip.h --
need not save and restore it; v[126:127] cost two AGPR spill slots there and with them the second wave per SIMD).
m_k lives in the register of result limb k (m_k is last read in column k + 8, limb k is produced in column k + 9).
Hazards: every VGPR dependency in the block is interlocked by the hardware; VCC is written and never read.

    python keaki_amd/csrc/gen_fq29_asm.py > keaki_amd/csrc/fq29_asm.hip.h
"""
ACC = "v[54:55]"
ACC_LO, ACC_HI = "v54", "v55"


def block(square, dual=False, nprod=None):
    """nprod (2..6): r = (a0 b0 + a1 b1 + ... ) / 2^261, operands a[k*9+i], b[k*9+i]; overrides `dual`"""
    ops = []          # (constraint, expr) in operand order
    index = {}

    def op(key, constraint, expr):
        if key not in index:
            index[key] = len(ops)
            ops.append((constraint, expr))
        return "%%%d" % index[key]

    # outputs first
    R = [op(("r", i), '"=&v"', "r[%d]" % i) for i in range(9)]
    D = [op(("d", i), '"=&v"', "d[%d]" % i) for i in range(8)] if square else None
    nout = len(ops)
    if nprod:
        AN = [[op(("a", k, i), '"v"', "a%d[%d]" % (k, i)) for i in range(9)] for k in range(nprod)]
        BN = [[op(("b", k, i), '"v"', "b%d[%d]" % (k, i)) for i in range(9)] for k in range(nprod)]
        A = B = Cc = Dd = None
    else:
        A = [op(("a", i), '"v"', "a[%d]" % i) for i in range(9)]
        B = A if square else [op(("b", i), '"v"', "b[%d]" % i) for i in range(9)]
        Cc = [op(("c", i), '"v"', "c[%d]" % i) for i in range(9)] if dual else None
        Dd = [op(("e", i), '"v"', "d[%d]" % i) for i in range(9)] if dual else None
    P = [op(("p", i), '"s"', "Q29::MOD[%d]" % i) for i in range(9)]
    INV = op(("inv",), '"s"', "Q29::INV")
    L = []
    if square:
        for i in range(8):
            L.append("v_lshlrev_b32 %s, 1, %s" % (D[i], A[i]))
    first = True
    for k in range(17):
        lo_i = max(0, k - 8)
        terms = []
        if square:
            for i in range(lo_i, 9):
                j = k - i
                if j > 8 or i >= j:
                    continue
                terms.append((D[i], A[j]))
            if k % 2 == 0:
                terms.append((A[k // 2], A[k // 2]))
        elif nprod:
            for q in range(nprod):
                for i in range(lo_i, min(k, 8) + 1):
                    terms.append((AN[q][i], BN[q][k - i]))
        else:
            for i in range(lo_i, min(k, 8) + 1):
                terms.append((A[i], B[k - i]))
            if dual:
                for i in range(lo_i, min(k, 8) + 1):
                    terms.append((Cc[i], Dd[k - i]))
        for i in range(lo_i, min(k, 9)):          # m_i p_{k-i}, i < k
            if k - i <= 8:
                terms.append((R[i], P[k - i]))
        for (x, y) in terms:
            L.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, x, y, "0" if first else ACC))
            first = False
        if k < 9:
            L.append("v_mul_lo_u32 %s, %s, %s" % (R[k], ACC_LO, INV))
            L.append("v_and_b32 %s, 0x1fffffff, %s" % (R[k], R[k]))
            L.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, R[k], P[0], ACC))
            L.append("v_lshrrev_b64 %s, 29, %s" % (ACC, ACC))
        elif k < 16:
            L.append("v_and_b32 %s, 0x1fffffff, %s" % (R[k - 9], ACC_LO))
            L.append("v_lshrrev_b64 %s, 29, %s" % (ACC, ACC))
        else:
            L.append("v_and_b32 %s, 0x1fffffff, %s" % (R[7], ACC_LO))
            L.append("v_alignbit_b32 %s, %s, %s, 29" % (R[8], ACC_HI, ACC_LO))
    outs = ", ".join("%s(%s)" % o for o in ops[:nout])
    ins = ", ".join("%s(%s)" % o for o in ops[nout:])
    body = "\\n\\t".join(L)
    nmad = sum(1 for x in L if x.startswith("v_mad"))
    return '  asm("%s"\n      : %s\n      : %s\n      : "vcc", "%s", "%s");' % (body, outs, ins, ACC_LO, ACC_HI), len(L), nmad


def emit(name, square, dual=False):
    stmt, n, nmad = block(square, dual)
    if dual:
        print("// r = (a b + c d) / 2^261 (lazy): two products, ONE reduction. Limbs: a, b, d at most 2^29 + 8, c at most 1.5 * 2^30 (columns stay"
              " below 45 * 2^58). %d instructions, %d v_mad_u64_u32." % (n, nmad))
        print("KDEV void %s(u32* __restrict__ r, const u32* __restrict__ a, const u32* __restrict__ b, const u32* __restrict__ c, const u32* __restrict__ d) {" % name)
    elif square:
        print("// r = a^2 / 2^261 (lazy). Limbs of a at most 2^29 + 8. %d instructions, %d v_mad_u64_u32." % (n, nmad))
        print("KDEV void %s(u32* __restrict__ r, const u32* __restrict__ a) {" % name)
        print("  u32 d[8];")
    else:
        print("// r = a b / 2^261 (lazy). Limbs at most 2^30 + 16 on one side, 2^29 + 8 on the other. %d instructions, %d v_mad_u64_u32." % (n, nmad))
        print("KDEV void %s(u32* __restrict__ r, const u32* __restrict__ a, const u32* __restrict__ b) {" % name)
    print(stmt)
    print("}")


def emit_dot(nprod):
    stmt, n, nmad = block(False, False, nprod)
    print("// r = (sum_k a_k b_k) / 2^261 (lazy), k < %d: %d products, ONE reduction. All limbs at most 2^29 + 8 (a column holds %d products of < 2^58 + ..."
          " and stays below 2^64). %d instructions, %d v_mad_u64_u32." % (nprod, nprod, 9 * nprod + 9, n, nmad))
    args = ", ".join("const u32* __restrict__ a%d, const u32* __restrict__ b%d" % (k, k) for k in range(nprod))
    print("KDEV void u29_dot%d_asm(u32* __restrict__ r, %s) {" % (nprod, args))
    print(stmt)
    print("}")


def main():
    import sys
    dots = len(sys.argv) > 1 and sys.argv[1] == "--dots"
    print("// GENERATED by keaki_amd/csrc/gen_fq29_asm.py%s -- do not edit." % (" --dots" if dots else ""))
    print("#pragma once")
    print("namespace bn254 {")
    if dots:
        # the multi-product streams of the pairing tower (pair261.hip.h): python gen_fq29_asm.py --dots > fq29_dot_asm.hip.h
        for nprod in (3, 4, 6):
            emit_dot(nprod)
    else:
        emit("u29_mul_asm", False)
        emit("u29_sqr_asm", True)
        emit("u29_mul2_asm", False, True)
    print("}  // namespace bn254")


if __name__ == "__main__":
    main()
