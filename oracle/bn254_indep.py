"""TEST INFRASTRUCTURE ONLY -- an INDEPENDENT derivation of the BN254 optimal-ate pairing, used to pin oracle/bn254_py.py.

bn254_py.pairing() restates arkworks' schedule (G2Prepared line coefficients on the twist in homogeneous projective coordinates,
`mul_by_034` sparse products, the Fq2/Fq6/Fq12 tower, Frobenius coefficient tables, the Fuentes-Castaneda chain) -- and
oracle/bn254_ref.c restates the same formulas again. This file shares NONE of that:

  * Fq12 is ONE flat extension Fq[w]/(w^12 - 18 w^6 + 82)  (from w^6 = 9 + u, u^2 = -1), elements are lists of 12 integers;
    products are schoolbook polynomial products, inverses come from the extended Euclidean algorithm on polynomials;
  * Q in E'(Fq2) is UNTWISTED to the curve E: y^2 = x^3 + 3 over Fq12 as (x' w^2, y' w^3) -- and asserted to lie on E, so the twist type
    and the embedding are checked, not recalled;
  * the Miller loop is the textbook one in affine coordinates on E(Fq12): plain binary expansion of 6z + 2 (no NAF table), tangent and chord
    lines (y_P - y_T) - lambda (x_P - x_T) evaluated at P, dense products; the two closing lines use pi(Q) = (x^p, y^p) computed by a
    generic power (no Frobenius constants);
  * the final exponentiation is one plain square-and-multiply by (p^12 - 1)/r * m.

What stays RECALLED (and is named as such in DESIGN.md section 2): the multiple m = 2z(6z^2 + 3z + 1) that arkworks' hard part applies,
and the order in which serialize_uncompressed walks the tower (c0.c0.c0 ... c1.c2.c1, 32-byte little-endian canonical integers) --
the map flat_to_tower() below is that basis convention: coefficient of w^(2j + e) is tower component c_e.c_j.

Only tests/ may import this file (it follows no reference line: it is the mathematics of the pairing the reference calls at
src/kem.rs:30, :58 and src/kzg.rs:148).
"""
from __future__ import annotations

Z = 4965661367192848881
P = 36 * Z**4 + 36 * Z**3 + 24 * Z**2 + 6 * Z + 1
R = 36 * Z**4 + 36 * Z**3 + 18 * Z**2 + 6 * Z + 1
HARD_MULT = 2 * Z * (6 * Z * Z + 3 * Z + 1)          # ark-ec 0.4.2 models/bn final_exponentiation (hard part) = the published Fuentes-Castaneda decomposition (tests/test_oracle.py checks the identity)
N = 12
# w^12 = 18 w^6 - 82


def _reduce(c):
    """c: list of up to 23 coefficients -> 12, using w^12 = 18 w^6 - 82."""
    c = list(c) + [0] * (2 * N - 1 - len(c))
    for i in range(2 * N - 2, N - 1, -1):
        t = c[i]
        if t:
            c[i - 6] = (c[i - 6] + 18 * t) % P
            c[i - 12] = (c[i - 12] - 82 * t) % P
            c[i] = 0
    return [x % P for x in c[:N]]


def fmul(a, b):
    c = [0] * (2 * N - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                c[i + j] += x * y
    return _reduce(c)


def fadd(a, b): return [(x + y) % P for x, y in zip(a, b)]
def fsub(a, b): return [(x - y) % P for x, y in zip(a, b)]
def fconst(k): return [k % P] + [0] * (N - 1)
ONE = fconst(1)


def fpow(a, e):
    r = ONE
    while e:
        if e & 1:
            r = fmul(r, a)
        a = fmul(a, a)
        e >>= 1
    return r


# polynomial helpers over Fq for the inverse (coefficients low -> high)
def _trim(a):
    while a and a[-1] % P == 0:
        a = a[:-1]
    return a


def _divmod(a, b):
    a = [x % P for x in a]; b = _trim([x % P for x in b])
    q = [0] * max(1, len(a) - len(b) + 1)
    inv = pow(b[-1], -1, P)
    a = _trim(a)
    while len(a) >= len(b):
        d = len(a) - len(b)
        k = a[-1] * inv % P
        q[d] = k
        for i, y in enumerate(b):
            a[i + d] = (a[i + d] - k * y) % P
        a = _trim(a)
    return q, a


def _pmul(a, b):
    if not a or not b:
        return []
    c = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            c[i + j] = (c[i + j] + x * y) % P
    return c


def _psub(a, b):
    n = max(len(a), len(b))
    a = a + [0] * (n - len(a)); b = b + [0] * (n - len(b))
    return _trim([(x - y) % P for x, y in zip(a, b)])


MODULUS = [82, 0, 0, 0, 0, 0, (-18) % P, 0, 0, 0, 0, 0, 1]


def finv(a):
    """Inverse in Fq[w]/(modulus) by the extended Euclidean algorithm."""
    r0, r1 = MODULUS[:], _trim([x % P for x in a])
    assert r1, "inverse of zero"
    t0, t1 = [], [1]
    while len(r1) > 1:
        q, r = _divmod(r0, r1)
        r0, r1 = r1, r
        t0, t1 = t1, _psub(t0, _pmul(q, t1))
        if not r1:
            raise ZeroDivisionError("not invertible")
    k = pow(r1[0], -1, P)
    out = [x * k % P for x in t1]
    out = out + [0] * (N - len(out))
    assert fmul(out, [x % P for x in a] + [0] * (N - len(a))) == ONE
    return out[:N]


# --- the curve E: y^2 = x^3 + 3 over Fq12; points are (x, y) of flat elements, None = identity ---------------------------------------
def on_curve(pt):
    x, y = pt
    return fsub(fmul(y, y), fadd(fmul(fmul(x, x), x), fconst(3))) == [0] * N


def untwist(q2):
    """(x', y') on E': y^2 = x^3 + 3/(9 + u) over Fq2 (components (re, im))  ->  (x' w^2, y' w^3) on E over Fq12."""
    (xr, xi), (yr, yi) = q2

    def emb(re, im, shift):     # (re + im u) w^shift with u = w^6 - 9
        c = [0] * (2 * N - 1)
        c[shift] = (re - 9 * im) % P
        c[shift + 6] = im % P
        return _reduce(c)
    pt = (emb(xr, xi, 2), emb(yr, yi, 3))
    assert on_curve(pt), "the untwisted point is not on y^2 = x^3 + 3: wrong twist type / embedding"
    return pt


def _line_and_add(t, q, px, py):
    """l_{T,Q}(P) and T + Q on E(Fq12), affine; T, Q finite, T != -Q."""
    (x1, y1), (x2, y2) = t, q
    if x1 == x2 and y1 == y2:
        lam = fmul(fmul(fconst(3), fmul(x1, x1)), finv(fadd(y1, y1)))
    else:
        lam = fmul(fsub(y2, y1), finv(fsub(x2, x1)))
    x3 = fsub(fsub(fmul(lam, lam), x1), x2)
    y3 = fsub(fmul(lam, fsub(x1, x3)), y1)
    line = fsub(fsub(fconst(py), y1), fmul(lam, fsub(fconst(px), x1)))
    return line, (x3, y3)


def frob_point(q):
    return (fpow(q[0], P), fpow(q[1], P))


def pairing_flat(p1, q2):
    """Optimal ate pairing of P in E(Fq) (affine ints) and Q in E'(Fq2), raised to (p^12 - 1)/r * HARD_MULT; flat Fq12 element."""
    if p1 is None or q2 is None:
        return ONE
    px, py = p1
    assert (py * py - px * px * px - 3) % P == 0
    q = untwist(q2)
    s = 6 * Z + 2
    f = ONE
    t = q
    for bit in bin(s)[3:]:                      # MSB-first, below the leading one
        l, t = _line_and_add(t, t, px, py)
        f = fmul(fmul(f, f), l)
        if bit == "1":
            l, t = _line_and_add(t, q, px, py)
            f = fmul(f, l)
    q1 = frob_point(q)                          # pi(Q)
    q2p = frob_point(q1)                        # pi^2(Q)
    nq2 = (q2p[0], [(-c) % P for c in q2p[1]])
    l, t = _line_and_add(t, q1, px, py)
    f = fmul(f, l)
    l, t = _line_and_add(t, nq2, px, py)
    f = fmul(f, l)
    return fpow(f, (P**12 - 1) // R * HARD_MULT)


def flat_to_tower(c):
    """Flat coefficients of w^i -> ((c0.c0, c0.c1, c0.c2), (c1.c0, c1.c1, c1.c2)), each (re, im) over u: the tower
    Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - (9 + u)), Fq12 = Fq6[w]/(w^2 - v); the coefficient of w^(2j + e) is c_e.c_j."""
    def comp(i):
        im = c[i + 6] % P
        return ((c[i] + 9 * im) % P, im)
    return (tuple(comp(2 * j) for j in range(3)), tuple(comp(2 * j + 1) for j in range(3)))
