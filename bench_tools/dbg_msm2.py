import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as oc, bn254_py as py
from keaki_amd.hip import KeakiHip, jac_to_affine_words
hip = KeakiHip(0)
g1, _ = oc.generators()
mont = lambda v: oc.fr_to_mont(oc.ints_to_limbs(v))
pts = np.repeat(g1[None, :], 4, 0)
srs = hip.srs_g1_upload(pts)
for ks in ([1], [2], [3], [4], [5], [7], [8], [9], [64], [1, 1], [1, 2, 3], [0, 0, 0, 1]):
    got = oc.g1_to_ints(jac_to_affine_words(hip.msm_g1(srs, mont(ks))))[0]
    exp = py.g1_mul(py.G1_GEN, sum(ks))
    # find which multiple we got
    found = None
    acc = None
    for m in range(0, 200):
        if acc == got: found = m; break
        acc = py.g1_add(acc, py.G1_GEN)
    print(ks, "OK" if got == exp else "MISMATCH got multiple=%s" % found)
print("--- random point, various scalars")
from conftest_helpers import rand_fr_ints
P = hip.g1_mul_batch(g1, mont([12345]))
srs2 = hip.srs_g1_upload(P)
for k in [1, 5, 100, 500, 999, 1 << 20, (1 << 64) + 5, rand_fr_ints(1, 3)[0], py.R - 1]:
    got = oc.g1_to_ints(jac_to_affine_words(hip.msm_g1(srs2, mont([k]))))[0]
    exp = py.g1_mul(py.G1_GEN, 12345 * k)
    print(hex(k)[:20], "OK" if got == exp else "MISMATCH", hip.last_msm_stats()["window_bits"])
print("--- generator point, big scalars")
for k in [999, 1 << 20, (1 << 64) + 5, rand_fr_ints(1, 3)[0], py.R - 1]:
    got = oc.g1_to_ints(jac_to_affine_words(hip.msm_g1(srs, mont([k]))))[0]
    print(hex(k)[:20], "OK" if got == py.g1_mul(py.G1_GEN, k) else "MISMATCH")
