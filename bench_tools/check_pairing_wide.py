"""Wide (twelve lanes per pairing) against lane-pair pairing kernel: same bytes, and the latency of each by batch size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs
from oracle import bn254_py as py
def limbs(x): return np.frombuffer(int(x).to_bytes(32, "little"), np.uint64)
R256 = 1 << 256
g1 = np.concatenate([limbs(py.G1_GEN[0] * R256 % py.P), limbs(py.G1_GEN[1] * R256 % py.P)])
g2 = np.concatenate([limbs(c * R256 % py.P) for c in (py.G2_GEN[0][0], py.G2_GEN[0][1], py.G2_GEN[1][0], py.G2_GEN[1][1])])
h = KeakiHip(0)
N = 16384
P = h.g1_mul_batch(g1, random_fr_limbs(N, 11)); Q = h.g2_mul_batch(g2, random_fr_limbs(N, 12))
P[3] = 0; Q[5] = 0                                   # identities in either slot
for n in (1, 2, 3, 4, 5, 63, 64, 65, 257, 1000):
    h.set_option("pair_wide_max", 0); a = h.pairing_batch(P[:n], Q[:n])
    h.set_option("pair_wide_max", 1 << 20); h.set_option("pair_two_waves", 0); b = h.pairing_batch(P[:n], Q[:n])
    h.set_option("pair_two_waves", 1); c = h.pairing_batch(P[:n], Q[:n])
    print("n=%5d wide == lane-pair: %s, two waves == lane-pair: %s" % (n, np.array_equal(a, b), np.array_equal(a, c)), flush=True)
    assert np.array_equal(a, b) and np.array_equal(a, c)
for n in (1, 64, 512, 1024):
    out = []
    for tw in (0, 1):
        h.set_option("pair_two_waves", tw)
        h.pairing_batch(P[:n], Q[:n])
        t = []
        for rep in range(4):
            t0 = time.perf_counter(); h.pairing_batch(P[:n], Q[:n]); t.append(time.perf_counter() - t0)
        out.append(min(t) * 1e3)
    print("n=%5d  wide, one wave %.2f ms   two waves %.2f ms" % (n, out[0], out[1]), flush=True)
for n in (1, 4, 64, 256, 1024, 2048, 4096, 8192, 16384):
    out = []
    for wide in (0, 1 << 20):
        h.set_option("pair_wide_max", wide)
        h.pairing_batch(P[:n], Q[:n])
        t = []
        for rep in range(3):
            t0 = time.perf_counter(); h.pairing_batch(P[:n], Q[:n]); t.append(time.perf_counter() - t0)
        out.append(min(t) * 1e3)
    print("n=%5d  lane-pair %.2f ms   wide %.2f ms" % (n, out[0], out[1]), flush=True)
# the parts (test hooks): Miller loop alone (lines computed on the fly), final exponentiation alone
n = 64
for wide in (0, 1 << 20):
    h.set_option("pair_wide_max", wide)
    f = h.miller_loop_batch(P[:n], Q[:n]); g = h.final_exp_batch(f)
    tm, tf = [], []
    for rep in range(3):
        t0 = time.perf_counter(); f = h.miller_loop_batch(P[:n], Q[:n]); tm.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); g = h.final_exp_batch(f); tf.append(time.perf_counter() - t0)
    print("%s: Miller loop %.2f ms, final exponentiation %.2f ms" % ("wide" if wide else "lane-pair", min(tm) * 1e3, min(tf) * 1e3), flush=True)
    if wide: assert np.array_equal(f, f0) and np.array_equal(g, g0)
    f0, g0 = f, g
