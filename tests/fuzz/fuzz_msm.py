#!/usr/bin/env python3
"""Randomised cross-check of the MSM pipeline against the CPU oracle: random sizes (also just around tile / chunk boundaries), with and
without window tables, random / sparse / repeated scalars, some identity points; since round 5 every call of the host-pointer entry draws a
random chunk count and growth of the chunked upload (0 = one copy in front), and `open` runs beside `commit` on the same inputs.
python tests/fuzz/fuzz_msm.py [rounds [seed]]"""
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
    def draw_pipe():
        hip.set_option("msm_pipe_chunks", int(rng.choice([0, 0, 2, 3, 4, 6, 9, 17])))
        hip.set_option("msm_pipe_growth", int(rng.choice([100, 140, 300])))
    try:
        draw_pipe()
        got = jac_to_affine_words(hip.msm_g1(srs, scm))
        hip.srs_g1_precompute(srs)
        draw_pipe()
        got_t = jac_to_affine_words(hip.msm_g1(srs, scm))
        m = max(1, n // 3)                                   # a shorter polynomial on the same SRS / tables
        draw_pipe()
        got_p = jac_to_affine_words(hip.msm_g1(srs, scm[:m]))
        exp_p = oc.msm_g1(pts[:m], scm[:m], threads=8)
        # `open` on the same coefficients: device quotient from the top, chunk by chunk, + MSM
        z = mont(rand_fr_ints(1, S0 + 777 + it))[0]
        draw_pipe()
        proof, val = hip.kzg_open(srs, scm, z)
        q, v = oc.fr_quotient(scm, z)
        ok_open = np.array_equal(val.reshape(-1), v.reshape(-1)) and (n < 2 or np.array_equal(jac_to_affine_words(proof), oc.msm_g1(pts[:n - 1], q, threads=8)))
    finally:
        srs.free()
        hip.set_option("msm_pipe_chunks", -1); hip.set_option("msm_pipe_growth", 160)
    ok = np.array_equal(got, exp) and np.array_equal(got_t, exp) and np.array_equal(got_p, exp_p) and ok_open
    bad += not ok
    print("n=%6d mode=%d %s" % (n, mode, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
