"""Index-exact model of the sharded FK23 pipeline (keaki_amd/csrc/fft_g1.hip, `keaki_hip_fk_shard_*`), TEST INFRASTRUCTURE.

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


class ShardModel:
    """All R ranks of one job, step by step as the C ABI runs them."""

    def __init__(self, srs, log2d, R):
        self.d = 1 << log2d
        self.N = 2 * self.d
        self.log2d, self.R = log2d, R
        assert R >= 2 and R & (R - 1) == 0 and self.d >= R * R
        self.M = self.N // R
        w = root(self.N)
        wi = pow(w, Q - 2, Q)
        self.tw = [pow(w, k, Q) for k in range(self.d)]
        self.twi = [pow(wi, k, Q) for k in range(self.d)]
        self.inv_n = pow(self.N, Q - 2, Q)
        self.srs = srs
        self.hat_s = None

    # ---- hat_s = DFT_2d(reversed SRS): natural order in, cyclic layout -> bit-reversed positions out, block layout
    def setup_step0(self, r):
        R, M, N, d = self.R, self.M, self.N, self.d
        a = [(self.srs[d - 1 - (k * R + r)] if k * R + r < d else 0) for k in range(M)]
        half = M // 2
        while half >= 1:                      # global span len = 2 * half * R: N .. 2R
            stage(a, half, R, r, N // (2 * half * R), self.tw, dit=False)
            half //= 2
        return a                              # = send buffer: chunk for rank r' is [r' M/R, (r'+1) M/R)

    def setup_step1(self, r, recv):
        R, M, N = self.R, self.M, self.N
        a = transpose(recv, R, M // R)        # recv[src * (M/R) + t] -> local[t * R + src]
        half = R // 2
        while half >= 1:                      # global span len = 2 * half: R .. 2
            stage(a, half, 1, 0, N // (2 * half), self.tw, dit=False)
            half //= 2
        return a                              # local k <-> position q = r M + k holding hat_S[bitrev_N(q)]

    def setup(self):
        send = [self.setup_step0(r) for r in range(self.R)]
        recv = all_to_all(send)
        self.hat_s = [self.setup_step1(r, recv[r]) for r in range(self.R)]

    # ---- openings
    def hat_a(self, p):
        a = [0] * self.d + list(p)
        return [x * self.inv_n % Q for x in dft(a, self.tw[1])]

    def open_step0(self, r, p):
        R, M, N = self.R, self.M, self.N
        ha = self.hat_a(p)
        nb = self.log2d + 1
        a = [self.hat_s[r][k] * ha[bitrev(r * M + k, nb)] % Q for k in range(M)]
        half = 1
        while 2 * half <= M:                  # DIT, block layout: global span 2 .. M
            stage(a, half, 1, 0, N // (2 * half), self.twi, dit=True)
            half *= 2
        return transpose(a, M // R, R)        # send[s * (M/R) + t] = a[t * R + s]

    def open_step1(self, r, recv):
        R, M, N, d = self.R, self.M, self.N, self.d
        a = list(recv)                        # cyclic layout: local k <-> i = k R + r
        half = M // R
        while 2 * half <= M:                  # DIT, global span 2 * half * R: 2M .. N
            stage(a, half, R, r, N // (2 * half * R), self.twi, dit=True)
            half *= 2
        h = a[:M // 2]                        # i < d  <=>  k < M/2
        md = M // 2
        half = md // 2
        while half >= 1:                      # DFT_d, DIF, cyclic layout: global span 2 * half * R: d .. 2R; omega_d = omega_N^2
            stage(h, half, R, r, 2 * (d // (2 * half * R)), self.tw, dit=False)
            half //= 2
        return h

    def open_step2(self, r, recv):
        R, d = self.R, self.d
        md = d // R
        a = transpose(recv, R, md // R)
        half = R // 2
        while half >= 1:
            stage(a, half, 1, 0, 2 * (d // (2 * half)), self.tw, dit=False)
            half //= 2
        return a                              # local k <-> position q = r md + k holding proof[bitrev_d(q)]

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
