"""TEST INFRASTRUCTURE ONLY -- CPU oracle (pure-Python big integers) for keaki's BN254 hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (keaki_amd/) never does.

What it restates
----------------
keaki (reference, /root/reference) is generic glue over arkworks 0.4; all arithmetic on
the hot path lives in un-vendored crates pinned by Cargo.lock:
  ark-ec 0.4.2 (Cargo.lock:41-42), ark-ff 0.4.2 (:58-59), ark-bn254 0.4.0 (:30-31),
  ark-serialize 0.4.2 (:114-115), blake3 1.5.4 (:165-166).
Those sources are NOT in this container, so this file restates their *published*
algorithms from the call sites keaki makes:
  src/kzg.rs:98      VariableBaseMSM::msm_unchecked   -> msm_pippenger_arkworks()
  src/kzg.rs:89-124  commit / open                    -> kzg_commit() / kzg_open()
  src/kzg.rs:127-151 verify (2 pairings)              -> kzg_verify()
  src/kem.rs:13-50   encapsulate                      -> kem_encapsulate()
  src/kem.rs:55-72   decapsulate                      -> kem_decapsulate()
  src/kem.rs:30,58   E::pairing (BN optimal ate)      -> pairing() = final_exponentiation(miller_loop())
  src/kem.rs:32,61   serialize_uncompressed(GT)       -> gt_serialize()
  src/kem.rs:42-46   BLAKE3 XOF                       -> blake3_xof()
  src/kem.rs:26      Fr::rand                         -> fr_rand_from_u64s()

PARITY STATUS: the reference holds no hard-coded group element / GT value / key for this
path (every reference test is relational and runs on BLS12-381), and
arkworks cannot be built here, so ABSOLUTE values are "parity unpinned" against arkworks
itself. What pins this oracle instead:
  * public constants: BN254 p, r, z, generators (EIP-196/197), 2*G1 known coordinates;
  * the official BLAKE3 test vectors (empty input / 1 byte / multi-chunk) for the KDF;
  * real ceremony points from the reference's own fixture ptau/ppot_0080_01.ptau.test
    (on-curve after de-Montgomery; pairing relations e(tau G1, G2) == e(G1, tau G2) ...);
  * internal cross-checks: the arkworks final-exponentiation chain == naive pow by
    (p^12-1)/r * 2z(6z^2+3z+1); bilinearity; Pippenger == naive double-and-add.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------------------
# Constants (ark-bn254 0.4.0: fields/fq.rs, fields/fr.rs, curves/mod.rs, curves/g1.rs, g2.rs)
# ----------------------------------------------------------------------------------------
Z = 4965661367192848881  # BN parameter "X", X_IS_NEGATIVE = false
P = 36 * Z**4 + 36 * Z**3 + 24 * Z**2 + 6 * Z + 1
R = 36 * Z**4 + 36 * Z**3 + 18 * Z**2 + 6 * Z + 1
assert P == 21888242871839275222246405745257275088696311157297823662689037894645226208583
assert R == 21888242871839275222246405745257275088548364400416034343698204186575808495617

MONT_R = 1 << 256          # Montgomery radix for 4x64-bit limbs (ark-ff MontBackend<_, 4>)
FQ_R = MONT_R % P
FR_R = MONT_R % R
FQ_RINV = pow(MONT_R, -1, P)
FR_RINV = pow(MONT_R, -1, R)

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
B1 = 3

# signed NAF-like digits of 6z+2, least significant first (ark-bn254 ATE_LOOP_COUNT)
ATE_LOOP_COUNT = [
    0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0,
    0, 1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0,
    -1, 0, 0, 1, 0, 1, 1,
]
assert sum(d << i for i, d in enumerate(ATE_LOOP_COUNT)) == 6 * Z + 2
assert len(ATE_LOOP_COUNT) == 65


# ----------------------------------------------------------------------------------------
# Fq2 = Fq[u]/(u^2+1)   (ark-bn254 Fq2Config: NONRESIDUE = -1)
# ----------------------------------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_dbl(a): return ((2 * a[0]) % P, (2 * a[1]) % P)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a):
    return ((a[0] * a[0] - a[1] * a[1]) % P, (2 * a[0] * a[1]) % P)
def f2_mul_fp(a, k): return ((a[0] * k) % P, (a[1] * k) % P)
def f2_conj(a): return (a[0], (-a[1]) % P)
def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return ((a[0] * n) % P, (-a[1] * n) % P)
def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1: r = f2_mul(r, a)
        a = f2_sqr(a); e >>= 1
    return r

F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (9, 1)  # Fq6 non-residue 9+u (ark-bn254 Fq6Config::NONRESIDUE)
def f2_mul_xi(a):  # (a0 + a1 u)(9 + u)
    return ((9 * a[0] - a[1]) % P, (9 * a[1] + a[0]) % P)

B2 = f2_mul_fp(f2_inv(XI), 3)  # G2 COEFF_B = 3/(9+u), D-type twist


# ----------------------------------------------------------------------------------------
# Fq6 = Fq2[v]/(v^3 - XI),  Fq12 = Fq6[w]/(w^2 - v)
# ----------------------------------------------------------------------------------------
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)
def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)
def f6_mul(a, b):
    a0, a1, a2 = a; b0, b1, b2 = b
    c0 = f2_add(f2_mul(a0, b0), f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(f2_mul(a2, b2)))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (c0, c1, c2)
def f6_mul_v(a):  # multiply by v: (a0,a1,a2) -> (xi*a2, a0, a1)
    return (f2_mul_xi(a[2]), a[0], a[1])
def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    n = f2_add(f2_mul(a0, t0), f2_mul_xi(f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    ni = f2_inv(n)
    return (f2_mul(t0, ni), f2_mul(t1, ni), f2_mul(t2, ni))

F12_ONE = (F6_ONE, F6_ZERO)
def f12_mul(a, b):
    a0, a1 = a; b0, b1 = b
    t0 = f6_mul(a0, b0); t1 = f6_mul(a1, b1)
    c0 = f6_add(t0, f6_mul_v(t1))
    c1 = f6_add(f6_mul(a0, b1), f6_mul(a1, b0))
    return (c0, c1)
def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return (a[0], f6_neg(a[1]))  # = a^(p^6): "cyclotomic_inverse" on the cyclotomic subgroup
def f12_inv(a):
    a0, a1 = a
    n = f6_sub(f6_mul(a0, a0), f6_mul_v(f6_mul(a1, a1)))
    ni = f6_inv(n)
    return (f6_mul(a0, ni), f6_neg(f6_mul(a1, ni)))
def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1: r = f12_mul(r, a)
        a = f12_sqr(a); e >>= 1
    return r

# Frobenius: coefficients derived from XI (ark-bn254 Fq6Config/Fq12Config FROBENIUS_COEFF_*).
# For x = sum_{i<6} c_i w^i (c_i in Fq2, w^6 = XI):  x^(p^k) = sum conj^k(c_i) * XI^(i (p^k-1)/6) * w^i
_FROB_W = [[f2_pow(XI, i * (P**k - 1) // 6) for i in range(6)] for k in range(4)]

def _f12_to_w(a):  # -> [c_0..c_5] coefficients of w^i ; w^2 = v so v^j w^e -> w^(2j+e)
    (a00, a01, a02), (a10, a11, a12) = a
    return [a00, a10, a01, a11, a02, a12]
def _f12_from_w(c):
    return ((c[0], c[2], c[4]), (c[1], c[3], c[5]))
def f12_frob(a, k):
    c = _f12_to_w(a)
    out = []
    for i in range(6):
        ci = c[i] if k % 2 == 0 else f2_conj(c[i])
        out.append(f2_mul(ci, _FROB_W[k][i]))
    return _f12_from_w(out)

def f12_mul_by_034(f, c0, c3, c4):
    """ark-ff Fp12::mul_by_034: multiply by the sparse element c0 + (c3 + c4 v) w."""
    s = ((c0, F2_ZERO, F2_ZERO), (c3, c4, F2_ZERO))
    return f12_mul(f, s)


# ----------------------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fq ; G2: y^2 = x^3 + 3/XI over Fq2.  Affine, None = identity.
# ----------------------------------------------------------------------------------------
def g1_is_on_curve(pt):
    if pt is None: return True
    x, y = pt
    return (y * y - x * x * x - B1) % P == 0
def g1_neg(pt): return None if pt is None else (pt[0], (-pt[1]) % P)
def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    x1, y1 = a; x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0: return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)
def g1_mul(pt, k):
    k %= R
    acc = None
    while k:
        if k & 1: acc = g1_add(acc, pt)
        pt = g1_add(pt, pt); k >>= 1
    return acc

def g2_is_on_curve(pt):
    if pt is None: return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO
def g2_neg(pt): return None if pt is None else (pt[0], f2_neg(pt[1]))
def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    x1, y1 = a; x2, y2 = b
    if x1 == x2:
        if f2_add(y1, y2) == F2_ZERO: return None
        lam = f2_mul(f2_mul_fp(f2_sqr(x1), 3), f2_inv(f2_dbl(y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))
def g2_mul(pt, k):
    k %= R
    acc = None
    while k:
        if k & 1: acc = g2_add(acc, pt)
        pt = g2_add(pt, pt); k >>= 1
    return acc


# ----------------------------------------------------------------------------------------
# MSM  (ark-ec 0.4.2 scalar_mul/variable_base/mod.rs: msm_bigint_wnaf + make_digits)
# ----------------------------------------------------------------------------------------
def ark_ceil_log2(n):  # ark_std::log2
    if n <= 1: return 0
    return (n - 1).bit_length()
def ark_window_size(n):
    """c = 3 if n < 32 else ln_without_floats(n) + 2, ln_without_floats(a) = log2(a)*69/100."""
    return 3 if n < 32 else ark_ceil_log2(n) * 69 // 100 + 2
def ark_make_digits(k, w, num_bits=254):
    """Signed radix-2^w digits in [-2^(w-1), 2^(w-1)), carry folded into the last digit."""
    radix = 1 << w
    mask = radix - 1
    carry = 0
    ndig = (num_bits + w - 1) // w
    digits = []
    for i in range(ndig):
        coef = carry + ((k >> (i * w)) & mask)
        carry = (coef + radix // 2) >> w
        digits.append(coef - (carry << w))
    digits[-1] += carry << w
    return digits
def msm_pippenger_arkworks(bases, scalars, group_add=g1_add, group_neg=g1_neg):
    """bases: affine points; scalars: canonical ints (< r). zip-truncates like msm_unchecked."""
    n = min(len(bases), len(scalars))
    if n == 0: return None
    c = ark_window_size(n)
    digs = [ark_make_digits(s, c) for s in scalars[:n]]
    nd = len(digs[0])
    window_sums = []
    for i in range(nd):
        buckets = [None] * (1 << c)
        for d, base in zip(digs, bases):
            s = d[i]
            if s > 0: buckets[s - 1] = group_add(buckets[s - 1], base)
            elif s < 0: buckets[-s - 1] = group_add(buckets[-s - 1], group_neg(base))
        running = None; res = None
        for b in reversed(buckets):
            running = group_add(running, b)
            res = group_add(res, running)
        window_sums.append(res)
    total = None
    for s in reversed(window_sums[1:]):
        total = group_add(total, s)
        for _ in range(c):
            total = group_add(total, total)
    return group_add(window_sums[0], total)
def msm_naive(bases, scalars, group_add=g1_add, group_mul=g1_mul):
    acc = None
    for b, s in zip(bases, scalars):
        acc = group_add(acc, group_mul(b, s))
    return acc


# ----------------------------------------------------------------------------------------
# Pairing (ark-ec 0.4.2 models/bn/{mod.rs,g2.rs}): G2Prepared line coefficients in
# homogeneous projective coordinates, D-type twist, ell() = mul_by_034.
# ----------------------------------------------------------------------------------------
TWO_INV = pow(2, -1, P)
TWIST_MUL_BY_Q_X = f2_pow(XI, (P - 1) // 3)
TWIST_MUL_BY_Q_Y = f2_pow(XI, (P - 1) // 2)

def _mul_by_char(q):
    x, y = q
    return (f2_mul(f2_conj(x), TWIST_MUL_BY_Q_X), f2_mul(f2_conj(y), TWIST_MUL_BY_Q_Y))

def _line_double(r):
    x, y, z = r
    a = f2_mul_fp(f2_mul(x, y), TWO_INV)
    b = f2_sqr(y)
    c = f2_sqr(z)
    e = f2_mul(B2, f2_add(f2_dbl(c), c))
    f = f2_add(f2_dbl(e), e)
    g = f2_mul_fp(f2_add(b, f), TWO_INV)
    h = f2_sub(f2_sqr(f2_add(y, z)), f2_add(b, c))
    i = f2_sub(e, b)
    j = f2_sqr(x)
    e2 = f2_sqr(e)
    nx = f2_mul(a, f2_sub(b, f))
    ny = f2_sub(f2_sqr(g), f2_add(f2_dbl(e2), e2))
    nz = f2_mul(b, h)
    return (nx, ny, nz), (f2_neg(h), f2_add(f2_dbl(j), j), i)

def _line_add(r, q):
    x, y, z = r
    qx, qy = q
    theta = f2_sub(y, f2_mul(qy, z))
    lam = f2_sub(x, f2_mul(qx, z))
    c = f2_sqr(theta)
    d = f2_sqr(lam)
    e = f2_mul(lam, d)
    f = f2_mul(z, c)
    g = f2_mul(x, d)
    h = f2_sub(f2_add(e, f), f2_dbl(g))
    nx = f2_mul(lam, h)
    ny = f2_sub(f2_mul(theta, f2_sub(g, h)), f2_mul(e, y))
    nz = f2_mul(z, e)
    j = f2_sub(f2_mul(theta, qx), f2_mul(lam, qy))
    return (nx, ny, nz), (lam, f2_neg(theta), j)

def g2_prepare(q):
    """G2Prepared::from(G2Affine): list of (c0,c1,c2) line coefficients."""
    if q is None: return []
    coeffs = []
    r = (q[0], q[1], F2_ONE)
    negq = g2_neg(q)
    for bit in reversed(ATE_LOOP_COUNT[:-1]):
        r, l = _line_double(r); coeffs.append(l)
        if bit == 1:
            r, l = _line_add(r, q); coeffs.append(l)
        elif bit == -1:
            r, l = _line_add(r, negq); coeffs.append(l)
    q1 = _mul_by_char(q)
    q2 = _mul_by_char(q1)
    q2 = (q2[0], f2_neg(q2[1]))
    r, l = _line_add(r, q1); coeffs.append(l)
    r, l = _line_add(r, q2); coeffs.append(l)
    return coeffs

def _ell(f, coeffs, p):
    c0, c1, c2 = coeffs
    return f12_mul_by_034(f, f2_mul_fp(c0, p[1]), f2_mul_fp(c1, p[0]), c2)

def miller_loop(p, q):
    """Bn::multi_miller_loop for one pair; identity in either slot -> one."""
    if p is None or q is None: return F12_ONE
    coeffs = iter(g2_prepare(q))
    f = F12_ONE
    n = len(ATE_LOOP_COUNT)
    for i in range(n - 1, 0, -1):
        if i != n - 1: f = f12_sqr(f)
        f = _ell(f, next(coeffs), p)
        bit = ATE_LOOP_COUNT[i - 1]
        if bit in (1, -1):
            f = _ell(f, next(coeffs), p)
    f = _ell(f, next(coeffs), p)
    f = _ell(f, next(coeffs), p)
    return f

def _exp_by_neg_x(f):
    return f12_conj(f12_pow(f, Z))  # cyclotomic_exp(X) then cyclotomic inverse (X positive)

def final_exponentiation(f):
    """Bn::final_exponentiation: easy part (p^6-1)(p^2+1), hard part Fuentes-Castaneda chain."""
    f1 = f12_conj(f)
    f2 = f12_inv(f)
    r = f12_mul(f1, f2)
    f2 = r
    r = f12_frob(r, 2)
    r = f12_mul(r, f2)
    y0 = _exp_by_neg_x(r)
    y1 = f12_sqr(y0)
    y2 = f12_sqr(y1)
    y3 = f12_mul(y2, y1)
    y4 = _exp_by_neg_x(y3)
    y5 = f12_sqr(y4)
    y6 = _exp_by_neg_x(y5)
    y3 = f12_conj(y3)
    y6 = f12_conj(y6)
    y7 = f12_mul(y6, y4)
    y8 = f12_mul(y7, y3)
    y9 = f12_mul(y8, y1)
    y10 = f12_mul(y8, y4)
    y11 = f12_mul(y10, r)
    y12 = f12_frob(y9, 1)
    y13 = f12_mul(y12, y11)
    y8 = f12_frob(y8, 2)
    y14 = f12_mul(y8, y13)
    r = f12_conj(r)
    y15 = f12_frob(f12_mul(r, y9), 3)
    return f12_mul(y15, y14)

# the exponent the chain above realises: (p^12-1)/r * HARD_MULT
HARD_MULT = 2 * Z * (6 * Z * Z + 3 * Z + 1)
def final_exponentiation_naive(f):
    return f12_pow(f, (P**12 - 1) // R * HARD_MULT)

def pairing(p, q):
    return final_exponentiation(miller_loop(p, q))


# ----------------------------------------------------------------------------------------
# Wire formats
# ----------------------------------------------------------------------------------------
def gt_serialize(f):
    """ark-serialize serialize_uncompressed of Fq12: c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1,
    each a canonical (non-Montgomery) 32-byte little-endian integer. 384 bytes."""
    out = bytearray()
    for f6 in f:
        for f2 in f6:
            for c in f2:
                out += int(c).to_bytes(32, "little")
    return bytes(out)

def to_mont_limbs(x, mod):
    """canonical int -> 4 u64 limbs of x*2^256 mod `mod` (how ark-ff stores Fp)."""
    v = (x * MONT_R) % mod
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
def from_mont_limbs(limbs, mod):
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return (v * pow(MONT_R, -1, mod)) % mod


# ----------------------------------------------------------------------------------------
# BLAKE3 (public spec, blake3 1.5.4): hash mode, arbitrary input, XOF output
# ----------------------------------------------------------------------------------------
_B3_IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_B3_PERM = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
_CHUNK_START, _CHUNK_END, _PARENT, _ROOT = 1, 2, 4, 8
_M32 = 0xFFFFFFFF

def _rotr(x, n): return ((x >> n) | (x << (32 - n))) & _M32
def _g(s, a, b, c, d, mx, my):
    s[a] = (s[a] + s[b] + mx) & _M32; s[d] = _rotr(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & _M32;      s[b] = _rotr(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b] + my) & _M32; s[d] = _rotr(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & _M32;      s[b] = _rotr(s[b] ^ s[c], 7)
def _compress(cv, block_words, counter, block_len, flags):
    s = list(cv) + _B3_IV[:4] + [counter & _M32, (counter >> 32) & _M32, block_len, flags]
    m = list(block_words)
    for rnd in range(7):
        _g(s, 0, 4, 8, 12, m[0], m[1]); _g(s, 1, 5, 9, 13, m[2], m[3])
        _g(s, 2, 6, 10, 14, m[4], m[5]); _g(s, 3, 7, 11, 15, m[6], m[7])
        _g(s, 0, 5, 10, 15, m[8], m[9]); _g(s, 1, 6, 11, 12, m[10], m[11])
        _g(s, 2, 7, 8, 13, m[12], m[13]); _g(s, 3, 4, 9, 14, m[14], m[15])
        if rnd != 6: m = [m[i] for i in _B3_PERM]
    for i in range(8):
        s[i] ^= s[i + 8]
        s[i + 8] ^= cv[i]
    return s
def _words(block):
    block = block + b"\0" * (64 - len(block))
    return [int.from_bytes(block[4 * i:4 * i + 4], "little") for i in range(16)]

def _chunk_output(chunk, counter):
    """returns (cv_in, block_words, counter, block_len, flags) of the chunk's last block."""
    cv = list(_B3_IV)
    blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
    for i, blk in enumerate(blocks):
        flags = (_CHUNK_START if i == 0 else 0)
        if i == len(blocks) - 1:
            return (cv, _words(blk), counter, len(blk), flags | _CHUNK_END)
        cv = _compress(cv, _words(blk), counter, 64, flags)[:8]

def _parent_output(l_cv, r_cv):
    return (list(_B3_IV), list(l_cv) + list(r_cv), 0, 64, _PARENT)

def _cv_of(out):
    cv, w, c, bl, fl = out
    return _compress(cv, w, c, bl, fl)[:8]

def blake3_xof(data: bytes, out_len: int) -> bytes:
    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]
    def subtree(lo, hi):  # output node over chunks[lo:hi]
        if hi - lo == 1:
            return _chunk_output(chunks[lo], lo)
        n = hi - lo
        left = 1 << ((n - 1).bit_length() - 1)  # largest power of two < n
        return _parent_output(_cv_of(subtree(lo, lo + left)), _cv_of(subtree(lo + left, hi)))
    cv, w, c, bl, fl = subtree(0, len(chunks))
    out = bytearray()
    t = 0
    while len(out) < out_len:
        s = _compress(cv, w, t, bl, fl | _ROOT)
        for x in s:
            out += x.to_bytes(4, "little")
        t += 1
    return bytes(out[:out_len])


# ----------------------------------------------------------------------------------------
# Scheme layer restated (keaki src/kzg.rs, src/kem.rs, src/enc.rs)
# ----------------------------------------------------------------------------------------
def fr_rand_from_u64s(next_u64):
    """ark-ff 0.4.2 `impl Distribution<Fp> for Standard`: 4 x next_u64 (limb 0 first), clear the
    top 2 bits (256-254), reject if >= modulus. The accepted limbs ARE the Montgomery
    representation (src/kem.rs:26 draws r this way). Returns (mont_limbs, canonical_int)."""
    while True:
        limbs = [next_u64() for _ in range(4)]
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 2
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < R:
            return limbs, (v * FR_RINV) % R

def kzg_setup(secret, max_d):
    """src/kzg.rs:55-70: g1_pow[i] = g1 * secret^i, tau_g2 = g2 * secret."""
    g1 = [g1_mul(G1_GEN, pow(secret, i, R)) for i in range(max_d)]
    return g1, g2_mul(G2_GEN, secret)

def kzg_commit(g1_pow, coeffs):
    """src/kzg.rs:89-101. Raises ValueError((len, max)) like KZGError::PolynomialTooLarge."""
    if len(coeffs) > len(g1_pow):
        raise ValueError((len(coeffs), len(g1_pow)))
    return msm_pippenger_arkworks(g1_pow, coeffs)

def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs): acc = (acc * x + c) % R
    return acc

def poly_quotient(coeffs, z):
    """(p(x) - p(z)) / (x - z), exact; src/kzg.rs:109-120. Trailing zero coeffs trimmed like DensePolynomial."""
    coeffs = list(coeffs)
    while coeffs and coeffs[-1] % R == 0: coeffs.pop()
    if len(coeffs) <= 1: return []
    q = [0] * (len(coeffs) - 1)
    acc = 0
    for i in range(len(coeffs) - 1, 0, -1):
        acc = (acc * z + coeffs[i]) % R
        q[i - 1] = acc
    return q

def kzg_open(g1_pow, coeffs, z):
    return kzg_commit(g1_pow, poly_quotient(coeffs, z))

def fr_root_of_unity(n):
    """ark-poly Radix2EvaluationDomain generator of order n (n a power of two <= 2^28): 5^((r-1)/2^28) ^ (2^28 / n)"""
    w = pow(5, (R - 1) >> 28, R)
    return pow(w, (1 << 28) // n, R)

def kzg_open_fk(g1_pow, p):
    """src/kzg.rs:157-203 restated literally (naive O(d^2) DFTs): FK23 all-openings at the d-th roots of unity, d = len(p)."""
    d = len(p)
    n2 = 2 * d
    w2 = fr_root_of_unity(n2)
    s = [g1_pow[d - 1 - i] for i in range(d)] + [None] * d
    a = [0] * d + [c % R for c in p]
    def g_dft(v, w):
        out = []
        for j in range(len(v)):
            acc = None
            for i, pt in enumerate(v):
                acc = g1_add(acc, g1_mul(pt, pow(w, i * j, R)))
            out.append(acc)
        return out
    hat_s = g_dft(s, w2)
    hat_a = [sum(a[i] * pow(w2, i * j, R) for i in range(n2)) % R for j in range(n2)]
    hat_h = [g1_mul(hat_s[i], hat_a[i]) for i in range(n2)]
    inv = pow(n2, -1, R)
    h_prime = [g1_mul(x, inv) for x in g_dft(hat_h, pow(w2, -1, R))]
    h = h_prime[:d]
    return g_dft(h, fr_root_of_unity(d) if d > 1 else 1)

def kzg_verify(tau_g2, commitment, point, value, proof):
    """src/kzg.rs:127-151: e(C - [v]_1, g2) == e(proof, [tau]_2 - [point]_2)."""
    lhs = pairing(g1_add(commitment, g1_neg(g1_mul(G1_GEN, value))), G2_GEN)
    rhs = pairing(proof, g2_add(tau_g2, g2_neg(g2_mul(G2_GEN, point))))
    return lhs == rhs

def kem_encapsulate(r, tau_g2, commitment, point, value, msg_len):
    """src/kem.rs:13-50 with the random r passed in (canonical int). -> (ct affine G2, key bytes, gt bytes)"""
    com_beta = g1_add(commitment, g1_neg(g1_mul(G1_GEN, value)))
    secret = pairing(g1_mul(com_beta, r), G2_GEN)
    sb = gt_serialize(secret)
    tau_alpha = g2_add(tau_g2, g2_neg(g2_mul(G2_GEN, point)))
    ct = g2_mul(tau_alpha, r)
    return ct, blake3_xof(sb, msg_len), sb

def kem_decapsulate(proof, ct, msg_len):
    """src/kem.rs:55-72."""
    sb = gt_serialize(pairing(proof, ct))
    return blake3_xof(sb, msg_len), sb

def enc_encrypt(r, tau_g2, com, point, value, msg):
    ct, key, _ = kem_encapsulate(r, tau_g2, com, point, value, len(msg))
    return ct, bytes(k ^ m for k, m in zip(key, msg))

def enc_decrypt(proof, ct):
    key, _ = kem_decapsulate(proof, ct[0], len(ct[1]))
    return bytes(k ^ c for k, c in zip(key, ct[1]))


# ----------------------------------------------------------------------------------------
# The reference's own fixture: snarkjs .ptau (coordinates stored as Montgomery LE limbs)
# ----------------------------------------------------------------------------------------
def ptau_points(blob: bytes):
    """Section layout of ptau/ppot_0080_01.ptau.test as parsed by src/kzg/ptau.rs:129-322.
    Returns dict with de-Montgomeried affine points: tau_g1[3], tau_g2[2], alpha_tau_g1[2],
    beta_tau_g1[2], beta_g2[1]."""
    assert blob[:4] == b"ptau"
    nsec = int.from_bytes(blob[8:12], "little")
    pos = 12
    secs = {}
    for _ in range(nsec):
        sid = int.from_bytes(blob[pos:pos + 4], "little")
        size = int.from_bytes(blob[pos + 4:pos + 12], "little")
        secs[sid] = (pos + 12, size)
        pos += 12 + size
    def fq(off): return (int.from_bytes(blob[off:off + 32], "little") * FQ_RINV) % P
    def g1s(sid):
        off, size = secs[sid]
        return [(fq(off + 64 * i), fq(off + 64 * i + 32)) for i in range(size // 64)]
    def g2s(sid):
        off, size = secs[sid]
        return [((fq(off + 128 * i), fq(off + 128 * i + 32)), (fq(off + 128 * i + 64), fq(off + 128 * i + 96)))
                for i in range(size // 128)]
    return {"tau_g1": g1s(2), "tau_g2": g2s(3), "alpha_tau_g1": g1s(4), "beta_tau_g1": g1s(5), "beta_g2": g2s(6)}
