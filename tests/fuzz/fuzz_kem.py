#!/usr/bin/env python3
"""Randomised cross-check of encap_batch / decap_batch against the CPU oracle over batch sizes that hit every path (ladder < 256 <= fixed-base
tables; per-item pairing vs the GT fixed-base path, forced with the context option encap_gt).  python tests/fuzz/fuzz_kem.py [rounds [seed]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as oc
from conftest import rand_fr_ints
from keaki_amd.hip import KeakiHip
hip = KeakiHip(0)
mont = lambda ints: oc.fr_to_mont(oc.ints_to_limbs(ints))
g1, g2 = oc.generators()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
S0 = 0 if seed == 7 else seed * 100003
bad = 0
for it in range(rounds):
    hip.set_option("encap_gt", 64 if it % 2 else 1000000000)
    n = int(rng.integers(1, 700)) if it % 4 else [1, 255, 256, 257][(it // 4) % 4]
    tau, c0 = rand_fr_ints(2, S0 + 100 + it)
    com = hip.g1_mul_batch(g1, mont([c0]))[0]; tau_g2 = hip.g2_mul_batch(g2, mont([tau]))[0]
    A, V, Rr = mont(rand_fr_ints(n, S0 + 200 + it)), mont(rand_fr_ints(n, S0 + 300 + it)), mont(rand_fr_ints(n, S0 + 400 + it))
    ml = int(rng.integers(1, 100))
    ct, gt, key = hip.encap_batch(com, tau_g2, A, V, Rr, ml)
    ect, egt, ekey = oc.encap_batch(com, tau_g2, A, V, Rr, ml, threads=8)
    proofs = hip.g1_mul_batch(g1, mont(rand_fr_ints(n, S0 + 500 + it)))
    dgt, dkey = hip.decap_batch(proofs, ct, ml)
    egt2, ekey2 = oc.decap_batch(proofs, ect, ml, threads=8)
    ok = all(np.array_equal(a, b) for a, b in ((ct, ect), (gt, egt), (key, ekey), (dgt, egt2), (dkey, ekey2)))
    bad += not ok
    print("n=%4d msg_len=%3d gt_path=%s %s" % (n, ml, it % 2 == 1 and n >= 64, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
