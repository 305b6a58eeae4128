"""The width-5 NAF recoding of keaki_amd/csrc/jac29.hip.h::jac_scalar_mul_uniform_u29 (the sliding-window ladder of the wave-uniform FK23
stages), restated word for word on Python integers with the device's 32-bit word arithmetic: the digits must represent the value, be odd
and below 16 in magnitude, never sit closer than five positions, and fit the 129 positions the device reserves -- for random 127-bit
magnitudes and for the corners (0, 1, 2^127 - 1, runs of ones that carry to the top). CPU only."""
import random

M32 = (1 << 32) - 1


def recode(k):
    """(idx, neg) per position, least significant first -- the loop of the device function, u32 words a0..a3"""
    a = [(k >> (32 * i)) & M32 for i in range(4)]
    out = []
    for pos in range(129):
        byte = 0
        if a[0] & 1:
            m = a[0] & 31
            if m < 16:
                a[0] -= m
                byte = (m + 1) >> 1
            else:
                add = 32 - m
                c = a[0] + add; a[0] = c & M32; c >>= 32
                c += a[1]; a[1] = c & M32; c >>= 32
                c += a[2]; a[2] = c & M32; c >>= 32
                a[3] = (a[3] + c) & M32
                byte = ((add + 1) >> 1) | 0x80
        out.append(byte)
        a = [(a[0] >> 1) | ((a[1] << 31) & M32), (a[1] >> 1) | ((a[2] << 31) & M32), (a[2] >> 1) | ((a[3] << 31) & M32), a[3] >> 1]
    assert a == [0, 0, 0, 0], "129 positions were not enough"
    return out


def value(digits):
    v = 0
    for pos, b in enumerate(digits):
        if b:
            mag = 2 * (b & 0x7F) - 1
            v += (-mag if b & 0x80 else mag) << pos
    return v


def test_wnaf5_recoding_represents_the_scalar():
    rnd = random.Random(5)
    corners = [0, 1, 2, 15, 16, 17, 31, 32, 33, (1 << 127) - 1, (1 << 127) - 16, (1 << 127) - 17, (1 << 126) + 1, int("1" * 127, 2) ^ (1 << 60),
               sum(1 << i for i in range(0, 127, 5)), sum(31 << i for i in range(0, 125, 5))]
    weight = 0
    for k in corners + [rnd.getrandbits(127) for _ in range(3000)]:
        d = recode(k)
        assert value(d) == k
        nz = [i for i, b in enumerate(d) if b]
        assert all(1 <= (b & 0x7F) <= 8 for b in d if b)                     # |digit| = 2 idx - 1 in {1, 3, .., 15}: table entries 0..7
        assert all(j - i >= 5 for i, j in zip(nz, nz[1:]))                   # the defining property of a width-5 NAF
        weight += len(nz)
    assert weight / (len(corners) + 3000) < 23                               # ~127 / 6 non-zero digits per half: the additions the ladder saves
