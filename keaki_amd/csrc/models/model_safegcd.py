#!/usr/bin/env python3
"""The division-step inverse of bn254_field.hip.h (fq_inv_safegcd_words) statement by statement on Python integers: nine signed 30-bit limbs,
20 rounds of 30 constant-time divsteps on the low bits, the 2 x 2 transition matrix applied to (f, g) exactly and to (d, e) modulo p (a multiple
of p clears the low 30 bits), the final sign / range normalisation. Asserts the exactness of every shift and compares with pow(x, -1, p).
tests/test_pair261_model.py runs it in the CPU suite."""
import random

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
M30 = (1 << 30) - 1
P30 = [(P >> (30 * i)) & M30 for i in range(9)]
PINV30 = pow(P, -1, 1 << 30)
M32 = 0xFFFFFFFF


def sx32(v):
    v &= M32
    return v - (1 << 32) if v >> 31 else v


def s30(v):
    out = []
    for _ in range(8):
        out.append(v & M30); v >>= 30
    return out + [v]


def val(l):
    return sum(c << (30 * i) for i, c in enumerate(l))


def modinv(x):
    d, e, f, g, zeta = [0] * 9, [1] + [0] * 8, s30(P), s30(x), -1
    for _ in range(20):
        u, v, q, r = 1, 0, 0, 1
        ff, gg = f[0] & M32, g[0] & M32
        for _ in range(30):
            c1 = M32 if zeta < 0 else 0
            c2 = (-(gg & 1)) & M32
            x_, y_, z_ = ((ff ^ c1) - c1) & M32, ((u ^ c1) - c1) & M32, ((v ^ c1) - c1) & M32
            gg, q, r = (gg + (x_ & c2)) & M32, (q + (y_ & c2)) & M32, (r + (z_ & c2)) & M32
            c1 &= c2
            zeta = sx32((zeta & M32) ^ c1) - 1
            ff, u, v = (ff + (gg & c1)) & M32, (u + (q & c1)) & M32, (v + (r & c1)) & M32
            gg, u, v = gg >> 1, (u << 1) & M32, (v << 1) & M32
        u, v, q, r = map(sx32, (u, v, q, r))
        assert all(-(1 << 30) <= t <= 1 << 30 for t in (u, v, q, r))
        sd, se = (-1 if d[8] < 0 else 0), (-1 if e[8] < 0 else 0)
        md, me = (u & sd) + (v & se), (q & sd) + (r & se)
        cd, ce = u * d[0] + v * e[0], q * d[0] + r * e[0]
        md -= (PINV30 * (cd & M32) + md) & M30
        me -= (PINV30 * (ce & M32) + me) & M30
        cd += P30[0] * md; ce += P30[0] * me
        assert cd & M30 == 0 and ce & M30 == 0
        cd >>= 30; ce >>= 30
        nd, ne = [0] * 9, [0] * 9
        for i in range(1, 9):
            cd += u * d[i] + v * e[i] + P30[i] * md
            ce += q * d[i] + r * e[i] + P30[i] * me
            assert abs(cd) < 1 << 63 and abs(ce) < 1 << 63
            nd[i - 1], ne[i - 1] = cd & M30, ce & M30
            cd >>= 30; ce >>= 30
        nd[8], ne[8] = cd, ce
        assert -(1 << 31) <= cd < 1 << 31 and -(1 << 31) <= ce < 1 << 31
        cf, cg = u * f[0] + v * g[0], q * f[0] + r * g[0]
        assert cf & M30 == 0 and cg & M30 == 0
        cf >>= 30; cg >>= 30
        nf, ng = [0] * 9, [0] * 9
        for i in range(1, 9):
            cf += u * f[i] + v * g[i]; cg += q * f[i] + r * g[i]
            nf[i - 1], ng[i - 1] = cf & M30, cg & M30
            cf >>= 30; cg >>= 30
        nf[8], ng[8] = cf, cg
        d, e, f, g = nd, ne, nf, ng
    assert val(g) == 0 and (x == 0 or val(f) in (1, -1))
    # normalisation as the kernel does it
    neg = -1 if f[8] < 0 else 0
    add = -1 if d[8] < 0 else 0
    d = [((c + (p & add)) ^ neg) - neg for c, p in zip(d, P30)]
    for i in range(8):
        d[i + 1] += d[i] >> 30; d[i] &= M30
    add = -1 if d[8] < 0 else 0
    d = [c + (p & add) for c, p in zip(d, P30)]
    for i in range(8):
        d[i + 1] += d[i] >> 30; d[i] &= M30
    r = val(d)
    assert 0 <= r < P
    return r


def run(n=200, seed=1):
    rnd = random.Random(seed)
    for x in [1, 2, 3, P - 1, P - 2, (P - 1) // 2, 1 << 253] + [rnd.randrange(P) for _ in range(n)]:
        assert modinv(x) * x % P == 1, x
    assert modinv(0) == 0
    return True


if __name__ == "__main__":
    print("division-step inverse, 20 x 30 steps:", run())
