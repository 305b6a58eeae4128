import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as oc
from keaki_amd.hip import KeakiHip, jac_to_affine_words
from conftest_helpers import rand_fr_ints
hip = KeakiHip(0)
g1, _ = oc.generators()
mont = lambda v: oc.fr_to_mont(oc.ints_to_limbs(v))
for n in [1, 2, 3, 5, 8, 31, 32, 33, 100, 257, 1000, 1025, 4096, 20000, 1 << 16]:
    pts = hip.g1_mul_batch(g1, mont(rand_fr_ints(n, 100 + n)))
    for kind in ("small", "full"):
        sc = mont([k % 1000 for k in rand_fr_ints(n, 7)] if kind == "small" else rand_fr_ints(n, 200 + n))
        srs = hip.srs_g1_upload(pts)
        got = jac_to_affine_words(hip.msm_g1(srs, sc)); srs.free()
        ok = np.array_equal(got, oc.msm_g1(pts, sc, threads=8))
        print(n, kind, "c=%d" % hip.last_msm_stats()["window_bits"], "OK" if ok else "MISMATCH")
