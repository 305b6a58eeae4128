#!/usr/bin/env python3
"""Randomised cross-check of the MSM pipeline against the CPU oracle: random sizes (also just around tile / chunk boundaries), with and
without window tables, random / sparse / repeated scalars, some identity points.  python tests/fuzz/fuzz_msm.py [rounds [seed]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as oc
from conftest import rand_fr_ints
from keaki_amd.hip import KeakiHip, jac_to_affine_words
hip = KeakiHip(0)
mont = lambda ints: oc.fr_to_mont(oc.ints_to_limbs(ints))
g1, _ = oc.generators()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
rng = np.random.default_rng(seed)
S0 = 0 if seed == 2024 else seed * 100003
sizes = [1023, 1024, 1025, 2047, 2049, 16383, 16385, 32767, 32769, 65537] + [int(x) for x in rng.integers(1, 70000, rounds)]
bad = 0
for it, n in enumerate(sizes):
    ks = rand_fr_ints(n, S0 + 5000 + it)
    pts = hip.g1_mul_batch(g1, mont(ks))
    if n > 10:
        pts[rng.integers(0, n, 3)] = 0                      # identities
    mode = it % 4
    sc = rand_fr_ints(n, S0 + 9000 + it)
    if mode == 1:
        sc = [s if rng.random() < 0.3 else 0 for s in sc]   # sparse
    if mode == 2:
        sc = [sc[0] if rng.random() < 0.5 else sc[1] for _ in sc]   # two values: heavy buckets
    if mode == 3:
        sc = [int(x) for x in rng.integers(0, 2, n)]        # bits
    scm = mont(sc)
    exp = oc.msm_g1(pts, scm, threads=8)
    srs = hip.srs_g1_upload(pts)
    try:
        got = jac_to_affine_words(hip.msm_g1(srs, scm))
        hip.srs_g1_precompute(srs)
        got_t = jac_to_affine_words(hip.msm_g1(srs, scm))
        m = max(1, n // 3)                                   # a shorter polynomial on the same SRS / tables
        got_p = jac_to_affine_words(hip.msm_g1(srs, scm[:m]))
        exp_p = oc.msm_g1(pts[:m], scm[:m], threads=8)
    finally:
        srs.free()
    ok = np.array_equal(got, exp) and np.array_equal(got_t, exp) and np.array_equal(got_p, exp_p)
    bad += not ok
    print("n=%6d mode=%d %s" % (n, mode, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
