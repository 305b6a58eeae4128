import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oc():
    """C oracle (ctypes) -- the checker, never the thing under test."""
    import oracle as _oc
    _oc.build()
    return _oc


@pytest.fixture(scope="session")
def py():
    import bn254_py
    return bn254_py


@pytest.fixture(scope="session")
def hip():
    from keaki_amd.hip import KeakiHip
    h = KeakiHip(0)  # raises KeakiHipError when the extension or the GPU is missing: no fallback
    yield h
    h.close()


class SplitMix64:
    """Deterministic u64 stream (SURVEY.md section 8d synthetic inputs)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)


R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


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


@pytest.fixture(scope="session")
def rand_fr():
    return rand_fr_ints
