#!/usr/bin/env python3
"""Limb-exact model of the point arithmetic of jac29.hip.h (the ladders of the FK23 butterflies) in the 9 x 29-bit lazy arithmetic: the same
operations in the same order on Python integers, with assertions on every limb (no negative value, no 32-bit overflow, stream operands
within their budgets) and on every value bound, checked against plain affine arithmetic: the doubling (Y3 = E T - 8 B^2 as ONE dual stream,
no scaling passes), the mixed addition (Y3 = r T - Y1 H^3 as one dual stream), the shared add / subtract pair of the butterflies, the
effective-affine window tables (every entry on one isomorphic curve) in both shapes, whole ladders over them, and the pair of tables and
two-term ladders of the radix-4 passes. tests/test_pair261_model.py runs this in the CPU suite.

Streams (fq29_asm.hip.h): limbs exact on output, value < (sum of products) / 2^261 + p.
  mul(a, b):      a <= 2^30 + 16, b <= 2^29 + 8          sqr(a): a <= 2^29 + 8
  mul2(a,b,c,d):  a, b, d <= 2^29 + 8, c <= 1.5 * 2^30
"""
import random

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MASK = (1 << 29) - 1
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R261 = 1 << 261
C = (1 << 29) + 8
W = (1 << 30) + 16
W15 = 3 << 29


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


K2, K4, K8, K16, K32 = (biased(c) for c in (2, 4, 8, 16, 32))
K8W, K16W = biased(8, 31), biased(16, 31)
maxima = {}


def note(name, l, bound):
    v = val(l)
    assert v < bound * P, (name, v / P, bound)
    maxima[name] = max(maxima.get(name, 0.0), v / P)
    return l


def chk(l, lim, what):
    assert all(0 <= x <= lim for x in l[:8]) and 0 <= l[8] < 1 << 32, (what, [hex(x) for x in l])


def u32s(l, what):
    assert all(0 <= x < 1 << 32 for x in l), (what, l)
    return l


def carry(x):
    u32s(x, "carry.in")
    r = [x[0] & MASK] + [(x[i] & MASK) + (x[i - 1] >> 29) for i in range(1, 8)] + [x[8] + (x[7] >> 29)]
    assert r[8] < 1 << 32 and val(r) == val(x)
    return r


def redc(t):
    m = (t * ((-pow(P, -1, R261)) % R261)) % R261
    r = (t + m * P) >> 261
    assert r < t // R261 + P + 1
    return limbs(r)


def mul(a, b):
    chk(a, W, "mul.a"); chk(b, C, "mul.b")
    return redc(val(a) * val(b))


def sqr(a):
    chk(a, C, "sqr.a")
    return redc(val(a) ** 2)


def mul2(a, b, c, d):
    chk(a, C, "mul2.a"); chk(b, C, "mul2.b"); chk(c, W15, "mul2.c"); chk(d, C, "mul2.d")
    return redc(val(a) * val(b) + val(c) * val(d))


def scale(a, k):
    return carry(u32s([k * x for x in a], "scale"))


def sub(a, b, K):
    return carry(u32s([x - y + k for x, y, k in zip(a, b, K)], "sub"))


def sub2x(a, b, K):
    return carry(u32s([x - 2 * y + k for x, y, k in zip(a, b, K)], "sub2x"))


def sub3(a, b, c):
    return carry(u32s([x - y - 2 * z + k for x, y, z, k in zip(a, b, c, K8W)], "sub3"))


# ---- the two formulas, as jac29.hip.h writes them ------------------------------------------------------------------------------
def j29_dbl(p):
    """dbl-2009-l with D = 4 X Y^2 as a product: A = X^2, B = Y^2, S2 = X (2B) = D / 2, E = 3A, X3 = E^2 - 2D, Y3 = E (D - X3) - 8 B^2, Z3 = (2Y) Z"""
    x, y, z = p
    A, B = sqr(x), sqr(y)
    B2 = u32s([2 * b for b in B], "dbl.2B")                                  # raw: limbs < 2^30, a wide operand
    S2 = mul(B2, x)                                                          # 2 X B; x carried
    E = scale(A, 3)
    x3 = note("dbl.x", carry(u32s([e - 4 * s + k for e, s, k in zip(sqr(E), S2, K16W)], "dbl.x3")), 19)      # E^2 - 8 X B + 16p
    T = note("dbl.T", carry(u32s([2 * s - v + k for s, v, k in zip(S2, x3, K32)], "dbl.T")), 41)            # 4 X B - X3 + 32p
    N4 = note("dbl.N4", carry(u32s([k - 4 * b for k, b in zip(K16W, B)], "dbl.N4")), 16.01)                   # 16p - 4B == -4B
    y3 = note("dbl.y", mul2(E, T, B2, N4), 3.8)                              # E T - 8 B^2
    Y2 = u32s([2 * v for v in y], "dbl.2y")
    z3 = note("dbl.z", mul(Y2, z), 2.01)
    return (x3, y3, z3)


def j29_addsub(u, v):
    """(u + v, u - v) of the FK23 butterflies, both Jacobian with every coordinate below 32 p (a saturated residue shifted by 5 bits, or a ladder's
    running point): the products up to H and Z3 are shared, r^2 and the dual stream of Y are per output. None when u = +-v."""
    X1, Y1, Z1 = u
    X2, Y2, Z2 = v
    Z1Z1, Z2Z2 = sqr(Z1), sqr(Z2)
    U1, U2 = note("as.U1", mul(X1, Z2Z2), 2.35), note("as.U2", mul(X2, Z1Z1), 2.35)
    Z1cu, Z2cu = mul(Z1, Z1Z1), mul(Z2, Z2Z2)
    S1, S2 = note("as.S1", mul(Y1, Z2cu), 1.45), note("as.S2", mul(Y2, Z1cu), 1.45)
    ZZ = mul(Z1, Z2)
    H = note("as.H", sub(U2, U1, K4), 6.35)
    if val(H) % P == 0:
        return None
    z3 = note("as.z", mul(ZZ, H), 1.3)
    HH = sqr(H)
    HHH, V = mul(H, HH), mul(U1, HH)
    N = u32s([k - s1 for k, s1 in zip(K2, S1)], "as.N")
    out = []
    for minus in (False, True):
        if minus:
            rr = note("as.r-", carry(u32s([k - a - b for k, a, b in zip(K4, S2, S1)], "as.r-")), 4.01)
        else:
            rr = note("as.r+", sub(S2, S1, K2), 3.45)
        x3 = note("as.x", sub3(sqr(rr), HHH, V), 9.2)
        T = note("as.T", sub(V, x3, K16), 17.1)
        y3 = note("as.y", mul2(rr, T, N, HHH), 1.45)
        out.append((x3, y3, z3))
    return out


# ---- effective-affine window tables: every entry affine on ONE isomorphic curve, mixed additions in the ladder ---------------------
ONE = None  # set below (limbs of 2^261 mod p)


def j29_madd(a, x2, y2):
    """a (Jacobian, X < 19 p, Y <= 4 p, Z < 2.1 p) + (x2, y2) affine, both on the working curve: madd without factors of two (8M + 3S, Y3 one
    dual stream). Returns (point, H) -- H = Z3 / Z1 is the ratio the table build records -- or None when the points are equal or opposite."""
    x, y, z = a
    Z1Z1 = sqr(z)
    U2 = mul(x2, Z1Z1)
    S2 = mul(y2, mul(z, Z1Z1))
    H = note("madd.H", sub(U2, x, K32), 34.01)
    if val(H) % P == 0:
        return None
    HH = sqr(H)
    HHH, V = note("madd.HHH", mul(H, HH), 2.7), note("madd.V", mul(x, HH), 1.9)
    rr = note("madd.r", sub(S2, y, K4), 6.01)
    x3 = note("madd.x", sub3(sqr(rr), HHH, V), 9.3)
    T = note("madd.T", sub(V, x3, K16), 17.9)
    N = u32s([k - v for k, v in zip(K8, y)], "madd.N")                       # raw 8p - Y1 (Y1 <= 4p, limbs carried)
    chk(N, W15, "madd.N")
    y3 = note("madd.y", mul2(rr, T, N, HHH), 1.8)
    z3 = note("madd.z", mul(z, H), 1.5)
    return (x3, y3, z3), H


def table_multiples(Pj):
    """entries m P, m = 1..8, affine on the curve isomorphic to E by Z_g; returns (entries [(x, y)], Zfix) with Z_E = Z_acc * Zfix.
    Base curve: the one on which P itself is affine (X, Y) -- isomorphic by Z_P."""
    X, Y, ZP = Pj
    e = [(X, Y, ONE)]
    zr = [None]
    d = j29_dbl(e[0])                                                        # Z = 2 Y: the ratio to the previous Z = 1
    e.append(d); zr.append(d[2])
    for m in range(3, 9):
        r, H = j29_madd(e[-1], X, Y)
        e.append(r); zr.append(H)
    return globalz(e, zr, ZP)


def table_odd(Pj):
    """entries (2i + 1) P, i = 0..7, built on the curve on which 2P is affine"""
    X, Y, ZP = Pj
    d = j29_dbl(Pj)
    zd2 = sqr(d[2]); zd3 = mul(d[2], zd2)
    e = [(mul(X, zd2), mul(Y, zd3), ZP)]                                      # P on E_d: (X Zd^2, Y Zd^3, Z_P)
    zr = [None]
    for i in range(1, 8):
        r, H = j29_madd(e[-1], d[0], d[1])
        e.append(r); zr.append(H)
    return globalz(e, zr, d[2])


def globalz(e, zr, zbase):
    """bring every entry to the last one's Z: x_i *= zs^2, y_i *= zs^3 with zs = Z_last / Z_i (the product of the recorded ratios)"""
    out = [None] * len(e)
    out[-1] = (e[-1][0], e[-1][1])
    zs = None
    for i in range(len(e) - 2, -1, -1):
        zs = zr[i + 1] if zs is None else mul(zs, zr[i + 1])                 # zr: H (carried, < 34 p) or 2Y-type Z (exact)
        zs2 = sqr(zs) if max(zs[:8]) <= C else mul(zs, zs)
        out[i] = (mul(e[i][0], zs2), mul(e[i][1], mul(zs, zs2)))
        note("tab.x", out[i][0], 4.3); note("tab.y", out[i][1], 2.0)
    zfix = mul(e[-1][2], zbase)
    return out, zfix


def table_odd_on(m1):
    """table_odd for a point m1 given as Jacobian limbs ON THE CURRENT WORKING CURVE (any curve of the family): entries affine on the curve
    isomorphic to it by the returned zfix"""
    X, Y, ZP = m1
    d = j29_dbl(m1)
    zd2 = sqr(d[2]); zd3 = mul(d[2], zd2)
    e = [(mul(X, zd2), mul(Y, zd3), ZP)]
    zr = [None]
    for i in range(1, 8):
        r, H = j29_madd(e[-1], d[0], d[1])
        e.append(r); zr.append(H)
    return globalz(e, zr, d[2])


def tables_pair(Aj, Bj):
    """odd-multiple tables of TWO points on ONE working curve (the two-term ladders of the radix-4 butterflies): A's table on E_(zA); B moved
    to that curve as (X zA^2, Y zA^3, Z), its table built there (-> the curve isomorphic by zA zB'); A's entries rescaled by zB'.
    Returns (TA, TB, zfix)."""
    TA, zA = table_odd_on(Aj)
    zA2 = sqr(zA); zA3 = mul(zA, zA2)
    Bc = (mul(Bj[0], zA2), mul(Bj[1], zA3), Bj[2])
    TB, zB = table_odd_on(Bc)
    zB2 = sqr(zB); zB3 = mul(zB, zB2)
    TA2 = [(note("pair.x", mul(x, zB2), 2.0), note("pair.y", mul(y, zB3), 2.0)) for x, y in TA]
    return TA2, TB, mul(zA, zB)


# ---- plain arithmetic to compare with -------------------------------------------------------------------------------------------
def aff_add(p, q):
    if p is None: return q
    if q is None: return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if (y1 + y2) % P == 0: return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def aff_mul(k, p):
    r = None
    while k:
        if k & 1: r = aff_add(r, p)
        p = aff_add(p, p); k >>= 1
    return r


def to_aff(j):
    X, Y, Z = (val(c) * pow(R261, -1, P) % P for c in j)
    if Z == 0: return None
    zi = pow(Z, -1, P)
    return (X * zi * zi % P, Y * zi * zi * zi % P)


def mont(v):
    return limbs(v * R261 % P)


def run(seed=1, ladders=6, bits=127):
    rnd = random.Random(seed)
    # ---- ladders over effective-affine tables (mixed additions), both table shapes
    global ONE
    ONE = mont(1)
    for it in range(ladders):
        while True:
            x = rnd.randrange(P); y2 = (x * x * x + 3) % P
            y = pow(y2, (P + 1) // 4, P)
            if y * y % P == y2: break
        base = (x, y)
        zz = rnd.randrange(1, P)
        Pj = (mont(x * zz * zz % P), mont(y * zz * zz * zz % P), mont(zz))
        for shape in ("multiples", "odd"):
            tab, zfix = (table_multiples if shape == "multiples" else table_odd)(Pj)
            mult = (lambda i: i + 1) if shape == "multiples" else (lambda i: 2 * i + 1)
            zero = [0] * 9
            acc, accv = None, 0
            for step in range(40):
                if acc is not None:
                    for _ in range(rnd.randrange(1, 5)):
                        acc = j29_dbl(acc); accv *= 2
                i = rnd.randrange(8)
                neg = rnd.random() < 0.5
                ex, ey = tab[i]
                if neg: ey = sub(zero, ey, K4)
                sm = -mult(i) if neg else mult(i)
                if acc is None:
                    acc, accv = (ex, ey, ONE), sm
                else:
                    r = j29_madd(acc, ex, ey)
                    if r is None:
                        if (accv - sm) % R_ORDER == 0: acc = j29_dbl(acc); accv *= 2
                        else: acc, accv = None, 0
                    else:
                        acc, accv = r[0], accv + sm
            if acc is not None:
                final = (acc[0], acc[1], mul(acc[2], zfix))
                assert to_aff(final) == aff_mul(accv % R_ORDER, base), shape

    # ---- two-term ladders kA A + kB B over a pair of tables on one curve
    for it in range(max(2, ladders // 2)):
        pts = []
        for _ in range(2):
            while True:
                x = rnd.randrange(P); y2 = (x * x * x + 3) % P
                y = pow(y2, (P + 1) // 4, P)
                if y * y % P == y2: break
            zz = rnd.randrange(1, P)
            pts.append(((x, y), (mont(x * zz * zz % P), mont(y * zz * zz * zz % P), mont(zz))))
        (Aa, Aj), (Ba, Bj) = pts
        TA, TB, zfix = tables_pair(Aj, Bj)
        zero = [0] * 9
        acc, va, vb = None, 0, 0
        for step in range(60):
            if acc is not None:
                for _ in range(rnd.randrange(1, 4)):
                    acc = j29_dbl(acc); va *= 2; vb *= 2
            which = rnd.randrange(2)
            i = rnd.randrange(8)
            neg = rnd.random() < 0.5
            ex, ey = (TA if which == 0 else TB)[i]
            if neg: ey = sub(zero, ey, K4)
            sm = -(2 * i + 1) if neg else (2 * i + 1)
            if acc is None:
                acc = (ex, ey, ONE)
            else:
                r = j29_madd(acc, ex, ey)
                assert r is not None
                acc = r[0]
            if which == 0: va += sm
            else: vb += sm
        final = (acc[0], acc[1], mul(acc[2], zfix))
        assert to_aff(final) == aff_add(aff_mul(va % R_ORDER, Aa), aff_mul(vb % R_ORDER, Ba))

    # the butterflies' (u + v, u - v): inputs are canonical residues in the 2^256 form shifted by 5 bits (values up to 32 p), or a ladder's output
    for it in range(4 * ladders):
        pts = []
        for _ in range(2):
            while True:
                x = rnd.randrange(P); y2 = (x * x * x + 3) % P
                y = pow(y2, (P + 1) // 4, P)
                if y * y % P == y2: break
            zz = rnd.randrange(1, P)
            R256 = 1 << 256
            jac = (x * zz * zz % P, y * zz * zz * zz % P, zz)
            big = it % 2 == 0                              # worst case: residues close to p
            coords = tuple(limbs((((P - 1 - rnd.randrange(1 << 20)) if big and k != 2 else (c * R256 % P)) << 5)) for k, c in enumerate(jac))
            if big:
                # keep it a curve point: only the bounds matter in this branch, compare nothing
                pts.append((coords, None))
            else:
                pts.append((coords, (x, y)))
        (u, ua), (v, va) = pts
        res = j29_addsub(u, v)
        if res is not None and ua is not None and va is not None:
            assert to_aff(res[0]) == aff_add(ua, va) and to_aff(res[1]) == aff_add(ua, (va[0], (P - va[1]) % P))
    return dict(maxima)


if __name__ == "__main__":
    m = run(1, 12)
    for k in sorted(m):
        print("%-8s max %.3f p" % (k, m[k]))
