import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs
from oracle import bn254_py as py
limbs = lambda x: np.frombuffer(int(x).to_bytes(32, "little"), np.uint64)
R256 = 1 << 256
g1 = np.concatenate([limbs(py.G1_GEN[0] * R256 % py.P), limbs(py.G1_GEN[1] * R256 % py.P)])
g2 = np.concatenate([limbs(c * R256 % py.P) for c in (py.G2_GEN[0][0], py.G2_GEN[0][1], py.G2_GEN[1][0], py.G2_GEN[1][1])])
h = KeakiHip(0)
for seed in range(4):
    n = 4096
    P = h.g1_mul_batch(g1, random_fr_limbs(n, 100 + seed)); Q = h.g2_mul_batch(g2, random_fr_limbs(n, 200 + seed))
    P[seed] = 0; Q[n - 1 - seed] = 0
    h.set_option("pair_wide_max", 0); a = h.pairing_batch(P, Q)
    h.set_option("pair_wide_max", 1 << 20); h.set_option("pair_two_waves", 0); b = h.pairing_batch(P, Q)
    h.set_option("pair_two_waves", 1); c = np.concatenate([h.pairing_batch(P[i:i + 1024], Q[i:i + 1024]) for i in range(0, n, 1024)])
    print(seed, np.array_equal(a, b), np.array_equal(a, c), flush=True)
    # tabulated lines: encap to new commitments exercises wide2<true>; compare GT path against per-item pairing path
    tau = h.g2_mul_batch(g2, random_fr_limbs(1, 6))[0]
    A, V, Rr = random_fr_limbs(700, 1 + seed), random_fr_limbs(700, 20 + seed), random_fr_limbs(700, 30 + seed)
    h.set_option("encap_gt", -1); x = h.encap_batch(P[7], tau, A, V, Rr, 32)
    h.set_option("encap_gt", 1 << 30); h.set_option("pair_wide_max", 0); y = h.encap_batch(P[7], tau, A, V, Rr, 32)
    h.set_option("encap_gt", -1); h.set_option("pair_wide_max", -1)
    print("  encap GT path == pairing path:", all(np.array_equal(u, v) for u, v in zip(x, y)), flush=True)
