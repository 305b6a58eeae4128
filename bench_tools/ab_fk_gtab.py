"""A/B of an FK23 option on one box (same build, same inputs, proofs compared):  python3 bench_tools/ab_fk_gtab.py [LOG2D [OPTION]]
OPTION: fk_gtab (default; window tables of the per-lane-scalar ladders in the lane-contiguous global workspace vs private memory) or
fk_addsub29 (the butterflies' add + subtract in the lazy limbs with the shared products once vs two saturated additions)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED, R_MOD
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 21
opt = sys.argv[2] if len(sys.argv) > 2 else "fk_gtab"
d = 1 << lg
hip = KeakiHip(0)
g1 = np.array(mont_words(1) + mont_words(2), np.uint64)
pts = hip.g1_mul_batch(g1, random_fr_limbs(d, SEED + 1))
srs = hip.srs_g1_upload(pts)
w2d = pow(5, (R_MOD - 1) >> (lg + 1), R_MOD)
mont_fr = lambda v: np.frombuffer(((v << 256) % R_MOD).to_bytes(32, "little"), np.uint64).copy()
om, omi, inv2d = mont_fr(w2d), mont_fr(pow(w2d, -1, R_MOD)), mont_fr(pow(2 * d, -1, R_MOD))
coeffs = random_fr_limbs(d, SEED + 4242)
hip.set_timing(True)
res = {}
for gt in (1, 0, 1, 0, 1, 0):
    hip.set_option(opt, gt)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); proofs = hip.open_fk_poly(srs, lg, coeffs, om, omi, inv2d); ts.append(time.perf_counter() - t0)
    st = hip.last_fk_stats()
    print("%s=%d  d=2^%d  call %.1f ms  device %.1f ms  stages %.1f ms  pointwise %.1f ms" % (opt, gt, lg, min(ts) * 1e3, st["device_ms"], st["stages_ms"], st["pointwise_ms"]), flush=True)
    res.setdefault(gt, proofs)
    assert np.array_equal(proofs, res[gt])
assert np.array_equal(res[0], res[1]), "the two forms disagree"
print("same proofs: True")
