"""Helpers importable from spawned worker processes (conftest.py itself is pytest-only)."""
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)


def rand_fr_ints(n, seed):
    g = SplitMix64(seed)
    out = []
    while len(out) < n:
        limbs = [g.next() for _ in range(4)]
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 2
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < R_MOD:
            out.append(v)
    return out
