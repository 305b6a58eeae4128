"""Index-exact model of the FK23 pipelines of keaki_amd/csrc/fft_g1.hip -- un-sharded (open_fk_split) and sharded over R ranks
(`keaki_hip_fk_shard_*`, ShardModel) --, TEST INFRASTRUCTURE.

The group is replaced by the additive group of Z_q (q = 998244353, a "scalar multiplication" is a product mod q), everything else --
which rank holds which element at which local index, which twiddle a butterfly takes, what the all-to-all moves where -- is the
device code's. Used by tests/test_fk_shard_model.py to pin the index maps on the CPU against the plain three-transform pipeline
of kzg::open_fk (reference src/kzg.rs:157-203), and by tests/test_gpu_fk_shard.py as the description of the exchange a caller performs.
"""
Q = 998244353
G = 3


def root(order):
    assert (Q - 1) % order == 0
    return pow(G, (Q - 1) // order, Q)


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def dft(x, w):
    n = len(x)
    return [sum(x[i] * pow(w, i * k, Q) for i in range(n)) % Q for k in range(n)]


def stage(a, half, A, B, stride, tw, dit):
    """one radix-2 stage over a LOCAL array: local span 2*half; twiddle of local offset j: tw[(j*A + B) * stride]"""
    m = len(a)
    for blk in range(m // (2 * half)):
        for j in range(half):
            i0 = blk * 2 * half + j
            i1 = i0 + half
            w = tw[(j * A + B) * stride]
            u, v = a[i0], a[i1]
            if dit:
                v = v * w % Q
                a[i0], a[i1] = (u + v) % Q, (u - v) % Q
            else:
                a[i0], a[i1] = (u + v) % Q, (u - v) * w % Q


def transpose(a, rows, cols):
    """out[c * rows + r] = in[r * cols + c]"""
    out = [0] * (rows * cols)
    for r in range(rows):
        for c in range(cols):
            out[c * rows + r] = a[r * cols + c]
    return out


def all_to_all(send):
    """send[rank] = list of R equal chunks laid end to end -> recv[rank] (chunk from source s at offset s * chunk)"""
    R = len(send)
    c = len(send[0]) // R
    return [[x for s in range(R) for x in send[s][r * c:(r + 1) * c]] for r in range(R)]


def open_fk_split(srs, p, log2d):
    """The un-sharded device pipeline (fft_g1.hip::open_fk_run). With V = hat_a o hat_s (2d products):
         h_i = (1/2) (iDFT_d(V_even)_i + w^-i iDFT_d(V_odd)_i)   for i < d        (w = omega_2d)
         proofs = DFT_d(h) = (1/2) V_even + (1/2) DFT_d(D o iDFT_d(V_odd)),  D_i = w^-i
       -- the even half needs NO transform, the odd half one inverse and one forward transform of size d with a twist between them:
       d (log2 d + 3) scalar-mults instead of d (1.5 log2 d + 3). Forward transforms run decimation-in-frequency (natural in, bit-reversed
       out), the inverse one decimation-in-time (bit-reversed in, natural out): no permutation but the final one."""
    d = 1 << log2d
    N = 2 * d
    w = root(N)
    wi = pow(w, Q - 2, Q)
    tw = [pow(w, k, Q) for k in range(d)]
    twi = [pow(wi, k, Q) for k in range(d)]
    inv_n = pow(N, Q - 2, Q)
    # hat_s: DIF over the zero-padded reversed SRS; position q < d holds hat_s[2 brev(q)], position d + q holds hat_s[2 brev(q) + 1]
    hs = [srs[d - 1 - i] for i in range(d)] + [0] * d
    half = d
    while half >= 1:
        stage(hs, half, 1, 0, N // (2 * half), tw, dit=False)
        half //= 2
    a = [x * inv_n % Q for x in dft([0] * d + list(p), w)]            # hat_a / 2d, natural order (the scalar-field side)
    e = [a[2 * bitrev(q, log2d)] * d % Q * hs[q] % Q for q in range(d)]                 # (1/2) V_even at position q
    o = [a[2 * bitrev(q, log2d) + 1] * hs[d + q] % Q for q in range(d)]                 # V_odd / 2d
    half = 1
    while 2 * half <= d:
        stage(o, half, 1, 0, 2 * (d // (2 * half)), twi, dit=True)                      # unscaled inverse transform of size d
        half *= 2
    o = [o[i] * twi[i] % Q for i in range(d)]                                           # the twist
    half = d // 2
    while half >= 1:
        stage(o, half, 1, 0, 2 * (d // (2 * half)), tw, dit=False)
        half //= 2
    out = [0] * d
    for q in range(d):
        out[bitrev(q, log2d)] = (e[q] + o[q]) % Q
    return out


class ShardModel:
    """All R ranks of one job, step by step as the C ABI runs them (keaki_hip_fk_shard_*): the split pipeline of open_fk_split with every
    transform of size d = R * Md distributed -- cyclic layout (rank r holds position k R + r at local k) for the spans 2R..d, block layout
    (rank r holds position r Md + k) for the spans 2..Md, one all-to-all per layout switch."""

    def __init__(self, srs, log2d, R):
        self.d = 1 << log2d
        self.N = 2 * self.d
        self.log2d, self.R = log2d, R
        assert R >= 2 and R & (R - 1) == 0 and self.d >= R * R
        self.Md = self.d // R
        w = root(self.N)
        wi = pow(w, Q - 2, Q)
        self.tw = [pow(w, k, Q) for k in range(self.d)]
        self.twi = [pow(wi, k, Q) for k in range(self.d)]
        self.inv_n = pow(self.N, Q - 2, Q)
        self.srs = srs
        self.hs_even = self.hs_odd = None

    def _dif_cyclic(self, a, r):
        R, Md, d = self.R, self.Md, self.d
        half = Md // 2
        while half >= 1:                      # global span 2 * half * R: d .. 2R; omega_d = omega_2d^2
            stage(a, half, R, r, 2 * (d // (2 * half * R)), self.tw, dit=False)
            half //= 2

    def _dif_block(self, a):
        R, d = self.R, self.d
        half = R // 2
        while half >= 1:                      # global span 2 * half: R .. 2
            stage(a, half, 1, 0, 2 * (d // (2 * half)), self.tw, dit=False)
            half //= 2

    # ---- hat_s: even entries = DFT_d(S), odd entries = DFT_d(S_i w^i); both at bit-reversed positions, block layout. ONE exchange:
    # the chunk for rank q is [even part | odd part]
    def setup_step0(self, r):
        R, Md, d = self.R, self.Md, self.d
        ev = [self.srs[d - 1 - (k * R + r)] for k in range(Md)]
        od = [ev[k] * self.tw[k * R + r] % Q for k in range(Md)]
        self._dif_cyclic(ev, r)
        self._dif_cyclic(od, r)
        c = Md // R
        return [x for q in range(R) for part in (ev, od) for x in part[q * c:(q + 1) * c]]

    def setup_step1(self, r, recv):
        R, Md = self.R, self.Md
        c = Md // R
        ev = transpose([x for s in range(R) for x in recv[(2 * s) * c:(2 * s + 1) * c]], R, c)
        od = transpose([x for s in range(R) for x in recv[(2 * s + 1) * c:(2 * s + 2) * c]], R, c)
        self._dif_block(ev)
        self._dif_block(od)
        return ev, od

    def setup(self):
        recv = all_to_all([self.setup_step0(r) for r in range(self.R)])
        hs = [self.setup_step1(r, recv[r]) for r in range(self.R)]
        self.hs_even = [h[0] for h in hs]
        self.hs_odd = [h[1] for h in hs]

    # ---- openings
    def hat_a(self, p):
        return [x * self.inv_n % Q for x in dft([0] * self.d + list(p), self.tw[1])]

    def open_step0(self, r, p):
        R, Md, d = self.R, self.Md, self.d
        a = self.hat_a(p)
        pos = lambda k: bitrev(r * Md + k, self.log2d)
        self.e = getattr(self, "e", {})
        self.e[r] = [a[2 * pos(k)] * d % Q * self.hs_even[r][k] % Q for k in range(Md)]
        o = [a[2 * pos(k) + 1] * self.hs_odd[r][k] % Q for k in range(Md)]
        half = 1
        while 2 * half <= Md:                 # DIT, block layout: global span 2 .. Md
            stage(o, half, 1, 0, 2 * (d // (2 * half)), self.twi, dit=True)
            half *= 2
        return transpose(o, Md // R, R)       # send[s * (Md/R) + t] = o[t * R + s]

    def open_step1(self, r, recv):
        R, Md, d = self.R, self.Md, self.d
        a = list(recv)                        # cyclic layout: local k <-> i = k R + r
        half = Md // R
        while 2 * half <= Md:                 # DIT, global span 2 * half * R: 2 Md .. d
            stage(a, half, R, r, 2 * (d // (2 * half * R)), self.twi, dit=True)
            half *= 2
        a = [a[k] * self.twi[k * R + r] % Q for k in range(Md)]       # the twist
        self._dif_cyclic(a, r)
        return a

    def open_step2(self, r, recv):
        R, Md = self.R, self.Md
        a = transpose(recv, R, Md // R)
        self._dif_block(a)
        return [(a[k] + self.e[r][k]) % Q for k in range(Md)]      # local k <-> position q = r Md + k holding proof[bitrev_d(q)]

    def open_step3(self, gathered):
        out = [0] * self.d
        for q in range(self.d):
            out[bitrev(q, self.log2d)] = gathered[q]
        return out

    def open(self, p):
        R = self.R
        recv = all_to_all([self.open_step0(r, p) for r in range(R)])
        recv = all_to_all([self.open_step1(r, recv[r]) for r in range(R)])
        parts = [self.open_step2(r, recv[r]) for r in range(R)]
        return self.open_step3([x for part in parts for x in part])       # all-gather in rank order


def open_fk_plain(srs, p, log2d):
    """the three transforms of src/kzg.rs:157-203 over Z_q"""
    d = 1 << log2d
    N = 2 * d
    w = root(N)
    s = [srs[d - 1 - i] for i in range(d)] + [0] * d
    hat_s = dft(s, w)
    hat_a = dft([0] * d + list(p), w)
    hat_h = [x * y % Q for x, y in zip(hat_s, hat_a)]
    inv = pow(N, Q - 2, Q)
    h = [x * inv % Q for x in dft(hat_h, pow(w, Q - 2, Q))][:d]
    return dft(h, w * w % Q)
