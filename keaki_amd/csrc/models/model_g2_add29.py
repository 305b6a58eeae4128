#!/usr/bin/env python3
"""Limb-exact model of the G2 mixed addition in the 9 x 29-bit lazy arithmetic (xyzz29_g2.hip.h): same operations in the same order on Python
integers, with assertions on every limb (no negative value, no 32-bit overflow, stream operands within their budgets) and on every value
bound; checked against plain Fq2 XYZZ arithmetic. Run before the kernel was written; tests/test_pair261_model.py runs it in the CPU suite.

Streams (fq29_asm.hip.h / fq29_dot_asm.hip.h): limbs exact on output, value < (sum of products)/2^261 + p.
  mul(a, b):       a <= 2^30 + 16, b <= 2^29 + 8
  mul2(a,b,c,d):   a, b, d <= 2^29 + 8, c <= 1.5 * 2^30
  dot4:            every limb <= 2^29 + 8
"""
import random

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MASK = (1 << 29) - 1
R261 = 1 << 261
RINV = pow(R261, -1, P)
C = (1 << 29) + 8          # carried
W = (1 << 30) + 16         # wide side of mul
W15 = 3 << 29              # 1.5 * 2^30


def limbs(v):
    return [(v >> (29 * i)) & MASK for i in range(8)] + [v >> 232]


def val(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def biased(c, bias_log=30):
    k = limbs(c * P)
    b, cy = 1 << bias_log, (1 << bias_log) >> 29
    out = [k[0] + b] + [k[i] + b - cy for i in range(1, 8)] + [k[8] - cy]
    assert val(out) == c * P and all(0 <= x < 1 << 32 for x in out)
    return out


K = {c: biased(c) for c in (2, 4, 8, 16, 32, 64, 128)}
K31 = {c: biased(c, 31) for c in (4, 8, 16, 32, 64)}
maxima = {}


def note(name, l, bound):
    v = val(l)
    assert v < bound * P, (name, v / P, bound)
    maxima[name] = max(maxima.get(name, 0), v / P)


def chk(l, lim, what):
    assert all(0 <= x <= lim for x in l[:8]) and 0 <= l[8] < 1 << 32, (what, [hex(x) for x in l])


def carry(x):
    assert all(0 <= v < 1 << 32 for v in x)
    r = [x[0] & MASK] + [(x[i] & MASK) + (x[i - 1] >> 29) for i in range(1, 8)] + [x[8] + (x[7] >> 29)]
    assert r[8] < 1 << 32 and val(r) == val(x)
    return r


def redc(t):
    """what a stream returns: some r == t / 2^261 (mod p) with r < t / 2^261 + p, limbs exact"""
    m = (t * ((-pow(P, -1, R261)) % R261)) % R261
    r = (t + m * P) >> 261
    assert (t + m * P) % R261 == 0
    return limbs(r)


def mul(a, b):
    chk(a, W, "mul.a"); chk(b, C, "mul.b")
    return redc(val(a) * val(b))


def mul2(a, b, c, d):
    chk(a, C, "mul2.a"); chk(b, C, "mul2.b"); chk(c, W15, "mul2.c"); chk(d, C, "mul2.d")
    return redc(val(a) * val(b) + val(c) * val(d))


def dot4(ops):
    t = 0
    for a, b in ops:
        chk(a, C, "dot4"); chk(b, C, "dot4")
        t += val(a) * val(b)
    return redc(t)


def sub_l(a, b, k):   # a - b + K, carried
    x = [a[i] - b[i] + k[i] for i in range(9)]
    assert all(0 <= v < 1 << 32 for v in x), "negative or overflowing limb in a difference"
    return carry(x)


def neg_raw(b, k):    # K - b, NOT carried (the c operand of mul2)
    x = [k[i] - b[i] for i in range(9)]
    assert all(0 <= v <= W15 for v in x[:8]) and x[8] >= 0
    return x


# ---- Fq2 on limbs: (a, b) = (re, im)
def l2_mul(x, y, kx):
    return (mul2(x[0], y[0], neg_raw(x[1], kx), y[1]), mul2(x[0], y[1], x[1], y[0]))


def l2_sqr(x, kx):
    s = [x[0][i] + x[1][i] for i in range(9)]
    d = sub_l(x[0], x[1], kx)
    t = [2 * x[0][i] for i in range(9)]
    return (mul(s, d), mul(t, x[1]))


def l2_sub(x, y, k):
    return (sub_l(x[0], y[0], k), sub_l(x[1], y[1], k))


ONE = None  # set below: limbs of 2^261 mod p (a plain factor that reduces a lazy value: x * ONE / 2^261 = x)


def add_mixed(acc, q):
    """acc = (X1, Y1, ZZ, ZZZ) lazy Fq2 limbs, every component below 4p (X1 below 2p), q = (X2, Y2) table coordinates entered by the 5-bit
    shift (< 32p, exact limbs). In Fq2 every component of a product is a DUAL product, so bounds grow twice as fast as in the G1 kernel: the
    chain is kept stable by bringing X3 back below 2p with one product by `one` per component (the other three coordinates are product outputs)."""
    X1, Y1, ZZ, ZZZ = acc
    X2, Y2 = q
    U2 = l2_mul(X2, ZZ, K[64]); S2 = l2_mul(Y2, ZZZ, K[64])         # (32 * 4 + 64 * 4) / 169 + 1
    for n, v in (("U2", U2), ("S2", S2)):
        note(n, v[0], 3.3); note(n, v[1], 3.3)
    Pd = l2_sub(U2, X1, K[2]); Rd = l2_sub(S2, Y1, K[4])
    for c in Pd: note("P", c, 5.3)
    for c in Rd: note("R", c, 7.3)
    PP = l2_sqr(Pd, K[8])
    for c in PP: note("PP", c, 2)
    PPP = l2_mul(Pd, PP, K[8]); Q = l2_mul(X1, PP, K[2])
    for c in PPP: note("PPP", c, 1.2)
    for c in Q: note("Q", c, 1.1)
    RR = l2_sqr(Rd, K[8])
    for c in RR: note("RR", c, 2.4)
    X3raw = tuple(carry([RR[j][i] - PPP[j][i] - 2 * Q[j][i] + K31[4][i] for i in range(9)]) for j in range(2))
    for c in X3raw: note("X3raw", c, 6.4)
    X3 = tuple(mul(c, ONE) for c in X3raw)                             # < 6.4 / 169 + 1
    for c in X3: note("X3", c, 1.1)
    T = l2_sub(Q, X3, K[2])
    for c in T: note("T", c, 3.1)
    nR1 = carry(neg_raw_any(Rd[1], K[8])); nY0 = carry(neg_raw_any(Y1[0], K[4])); nY1 = carry(neg_raw_any(Y1[1], K[4]))
    Y3 = (dot4([(Rd[0], T[0]), (nR1, T[1]), (nY0, PPP[0]), (Y1[1], PPP[1])]),
          dot4([(Rd[0], T[1]), (Rd[1], T[0]), (nY0, PPP[1]), (nY1, PPP[0])]))
    for c in Y3: note("Y3", c, 1.5)
    ZZ3 = l2_mul(ZZ, PP, K[4]); ZZZ3 = l2_mul(ZZZ, PPP, K[4])
    return (X3, Y3, ZZ3, ZZZ3)


def renorm(c):
    return mul(c, ONE)


def add_full(a, b):
    """general XYZZ addition of two lazy accumulators (the MSM tail: k_msm_reduce and friends): every component below 4p, X below 2p."""
    X1, Y1, ZZ1, ZZZ1 = a
    X2, Y2, ZZ2, ZZZ2 = b
    U1 = l2_mul(X1, ZZ2, K[2]); U2 = l2_mul(X2, ZZ1, K[2])
    S1 = l2_mul(Y1, ZZZ2, K[4]); S2 = l2_mul(Y2, ZZZ1, K[4])
    for n, v in (("fU", U1), ("fU", U2), ("fS", S1), ("fS", S2)):
        note(n, v[0], 1.2); note(n, v[1], 1.2)
    Pd = l2_sub(U2, U1, K[2]); Rd = l2_sub(S2, S1, K[2])
    for c in Pd + Rd: note("fPR", c, 3.2)
    PP = l2_sqr(Pd, K[4])
    for c in PP: note("fPP", c, 1.3)
    PPP = l2_mul(Pd, PP, K[4]); Q = l2_mul(U1, PP, K[2])
    for c in PPP + Q: note("fPPPQ", c, 1.15)
    RR = l2_sqr(Rd, K[4])
    for c in RR: note("fRR", c, 1.3)
    X3raw = tuple(carry([RR[j][i] - PPP[j][i] - 2 * Q[j][i] + K31[4][i] for i in range(9)]) for j in range(2))
    for c in X3raw: note("fX3raw", c, 5.3)
    X3 = tuple(renorm(c) for c in X3raw)
    for c in X3: note("fX3", c, 1.1)
    T = l2_sub(Q, X3, K[2])
    for c in T: note("fT", c, 3.2)
    nR1 = carry(neg_raw_any(Rd[1], K[4])); nS0 = carry(neg_raw_any(S1[0], K[2])); nS1 = carry(neg_raw_any(S1[1], K[2]))
    Y3 = (dot4([(Rd[0], T[0]), (nR1, T[1]), (nS0, PPP[0]), (S1[1], PPP[1])]),
          dot4([(Rd[0], T[1]), (Rd[1], T[0]), (nS0, PPP[1]), (nS1, PPP[0])]))
    for c in Y3: note("fY3", c, 1.3)
    ZZ3 = l2_mul(l2_mul(ZZ1, ZZ2, K[4]), PP, K[2]); ZZZ3 = l2_mul(l2_mul(ZZZ1, ZZZ2, K[4]), PPP, K[2])
    for c in ZZ3 + ZZZ3: note("fZZ", c, 1.1)
    return (X3, Y3, ZZ3, ZZZ3)


def dbl_full(a):
    """XYZZ doubling (a = 0) of a lazy accumulator: dbl-2008-s-1"""
    X1, Y1, ZZ1, ZZZ1 = a
    U = tuple(carry([2 * c[i] for i in range(9)]) for c in Y1)
    for c in U: note("dU", c, 8)
    V = l2_sqr(U, K[8])
    for c in V: note("dV", c, 2.6)
    Wd = l2_mul(U, V, K[8]); S = l2_mul(X1, V, K[2])
    for c in Wd: note("dW", c, 1.3)
    for c in S: note("dS", c, 1.1)
    XX = l2_sqr(X1, K[2])
    M = tuple(carry([3 * c[i] for i in range(9)]) for c in XX)
    for c in M: note("dM", c, 3.4)
    MM = l2_sqr(M, K[4])
    for c in MM: note("dMM", c, 1.35)
    X3raw = tuple(carry([MM[j][i] - 2 * S[j][i] + K31[4][i] for i in range(9)]) for j in range(2))
    X3 = tuple(renorm(c) for c in X3raw)
    for c in X3: note("dX3", c, 1.1)
    T = l2_sub(S, X3, K[2])
    nM1 = carry(neg_raw_any(M[1], K[4])); nW0 = carry(neg_raw_any(Wd[0], K[2])); nW1 = carry(neg_raw_any(Wd[1], K[2]))
    Y3 = (dot4([(M[0], T[0]), (nM1, T[1]), (nW0, Y1[0]), (Wd[1], Y1[1])]),
          dot4([(M[0], T[1]), (M[1], T[0]), (nW0, Y1[1]), (nW1, Y1[0])]))
    for c in Y3: note("dY3", c, 1.3)
    ZZ3 = l2_mul(V, ZZ1, K[4]); ZZZ3 = l2_mul(Wd, ZZZ1, K[2])
    for c in ZZ3 + ZZZ3: note("dZZ", c, 1.2)
    return (X3, Y3, ZZ3, ZZZ3)


def neg_raw_any(b, k):
    x = [k[i] - b[i] for i in range(9)]
    assert all(0 <= v < 1 << 32 for v in x)
    return x


# ---- reference: plain Fq2 XYZZ mixed addition on residues
def f2m(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2s(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def ref_add(acc, q):
    X1, Y1, ZZ, ZZZ = acc
    U2 = f2m(q[0], ZZ); S2 = f2m(q[1], ZZZ)
    Pd = f2s(U2, X1); Rd = f2s(S2, Y1)
    PP = f2m(Pd, Pd); PPP = f2m(Pd, PP); Q = f2m(X1, PP)
    X3 = f2s(f2s(f2m(Rd, Rd), PPP), ((2 * Q[0]) % P, (2 * Q[1]) % P))
    Y3 = f2s(f2m(Rd, f2s(Q, X3)), f2m(Y1, PPP))
    return (X3, Y3, f2m(ZZ, PP), f2m(ZZZ, PPP))


def ref_add_full(a, b):
    X1, Y1, ZZ1, ZZZ1 = a
    X2, Y2, ZZ2, ZZZ2 = b
    U1 = f2m(X1, ZZ2); U2 = f2m(X2, ZZ1); S1 = f2m(Y1, ZZZ2); S2 = f2m(Y2, ZZZ1)
    Pd = f2s(U2, U1); Rd = f2s(S2, S1)
    PP = f2m(Pd, Pd); PPP = f2m(Pd, PP); Q = f2m(U1, PP)
    X3 = f2s(f2s(f2m(Rd, Rd), PPP), ((2 * Q[0]) % P, (2 * Q[1]) % P))
    Y3 = f2s(f2m(Rd, f2s(Q, X3)), f2m(S1, PPP))
    return (X3, Y3, f2m(f2m(ZZ1, ZZ2), PP), f2m(f2m(ZZZ1, ZZZ2), PPP))


def ref_dbl(a):
    X1, Y1, ZZ1, ZZZ1 = a
    U = ((2 * Y1[0]) % P, (2 * Y1[1]) % P)
    V = f2m(U, U); Wd = f2m(U, V); S = f2m(X1, V)
    XX = f2m(X1, X1); M = ((3 * XX[0]) % P, (3 * XX[1]) % P)
    X3 = f2s(f2m(M, M), ((2 * S[0]) % P, (2 * S[1]) % P))
    Y3 = f2s(f2m(M, f2s(S, X3)), f2m(Wd, Y1))
    return (X3, Y3, f2m(V, ZZ1), f2m(Wd, ZZZ1))


def res(l):          # residue of a lazy 2^261-form value
    return val(l) * RINV % P


def run(trials=400, seed=1):
    global ONE
    ONE = limbs(R261 % P)
    rng = random.Random(seed)
    for t in range(trials):
        # a chain of additions: the accumulator's bounds are whatever the previous addition left
        first = tuple((limbs(rng.randrange(2 * P)), limbs(rng.randrange(2 * P))) for _ in range(2))      # X1, Y1 < 2p (entry product by one)
        acc = (first[0], first[1], (limbs(R261 % P), limbs(0)), (limbs(R261 % P), limbs(0)))
        racc = tuple((res(c[0]), res(c[1])) for c in acc)
        for step in range(6):
            worst = t < 20
            coord = lambda: (P - 1 if worst else rng.randrange(P))
            q_res = ((coord(), coord()), (coord(), coord()))                # residues x (what the table holds is x 2^256, cut shifted by 5: integer x 2^256 2^5 ... )
            # table word value = x * 2^256 mod p (< p); shift5 -> integer 32 * that (< 32p) == x 2^261 (mod p)
            q = tuple((limbs(32 * (c[0] * (1 << 256) % P)), limbs(32 * (c[1] * (1 << 256) % P))) for c in q_res)
            acc = add_mixed(acc, q)
            racc = ref_add(racc, q_res)
            got = tuple((res(c[0]), res(c[1])) for c in acc)
            assert got == racc, (t, step)
            for n, c, b in (("X", acc[0], 2), ("Y", acc[1], 4), ("ZZ", acc[2], 4), ("ZZZ", acc[3], 4)):
                note(n + "acc", c[0], b); note(n + "acc", c[1], b)
    run_tail(trials // 2, seed + 1)
    return maxima


def run_tail(trials, seed):
    """the tail's working form: accumulators that are loads of reduced values (< 2p) or outputs of add_full / dbl_full, in any mix"""
    rng = random.Random(seed)

    def fresh(worst):
        top = 12 * P // 10                                                  # l2_from_fq2 = (x << 5) * one / 2^261 < (32 / 169 + 1) p
        v = lambda: (top - 1 if worst else rng.randrange(top))
        return tuple((limbs(v()), limbs(v())) for _ in range(4))
    for t in range(trials):
        worst = t < 10
        acc = fresh(worst)
        racc = tuple((res(c[0]), res(c[1])) for c in acc)
        for step in range(8):
            kind = rng.randrange(3)
            if kind == 0:
                acc = dbl_full(acc); racc = ref_dbl(racc)
            else:
                other = fresh(worst) if kind == 1 else acc_prev if step else fresh(worst)
                rother = tuple((res(c[0]), res(c[1])) for c in other)
                acc_prev = acc
                acc = add_full(acc, other); racc = ref_add_full(racc, rother)
            if kind == 0:
                acc_prev = acc
            assert tuple((res(c[0]), res(c[1])) for c in acc) == racc, (t, step, kind)
            for n, c, b in (("tX", acc[0], 2), ("tY", acc[1], 4), ("tZZ", acc[2], 4), ("tZZZ", acc[3], 4)):
                note(n, c[0], b); note(n, c[1], b)


if __name__ == "__main__":
    m = run()
    for k in sorted(m):
        print("%-6s max %.2f p" % (k, m[k]))
