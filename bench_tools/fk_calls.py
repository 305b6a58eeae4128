"""FK23 setup (hat_s) + N calls of keaki_hip_open_fk_poly at d = 2^LOG2D, for the counter passes of collect_pmc_fk_pairing.sh (the difference of
a run with 2 calls and a run with 1 call is ONE call).   python3 bench_tools/fk_calls.py LOG2D N"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED, R_MOD
lg, calls = int(sys.argv[1]), int(sys.argv[2])
d = 1 << lg
hip = KeakiHip(0)
g1 = np.array(mont_words(1) + mont_words(2), np.uint64)
srs = hip.srs_g1_upload(hip.g1_mul_batch(g1, random_fr_limbs(d, SEED + 1)))
w2d = pow(5, (R_MOD - 1) >> (lg + 1), R_MOD)
mont_fr = lambda v: np.frombuffer(((v << 256) % R_MOD).to_bytes(32, "little"), np.uint64).copy()
om, omi, inv2d = mont_fr(w2d), mont_fr(pow(w2d, -1, R_MOD)), mont_fr(pow(2 * d, -1, R_MOD))
coeffs = random_fr_limbs(d, SEED + 4242)
for _ in range(calls):
    hip.open_fk_poly(srs, lg, coeffs, om, omi, inv2d)
print("done", calls)
