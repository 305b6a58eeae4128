#!/usr/bin/env python3
"""Random SEQUENCES of C-ABI calls on ONE context, every result checked against the CPU oracle: MSMs on two SRSs (one with window tables),
encap / decap batches on both GT paths with commitments that repeat or change, kzg_open, kzg_verify with two setups, pairing batches.
The per-feature tests exercise each call alone; this one is after state that leaks between calls (grow-only scratch buffers, cached tables
of [tau]_2 / A / B, cached line tables).   python tests/fuzz/fuzz_mixed.py [rounds [seed]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as oc
from conftest_helpers import rand_fr_ints, R_MOD
from keaki_amd.hip import KeakiHip, jac_to_affine_words

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
S0 = seed * 1000003
hip = KeakiHip(0)
mont = lambda ints: oc.fr_to_mont(oc.ints_to_limbs(ints))
g1, g2 = oc.generators()

# two setups: powers of tau in G1 (N points), [tau]_2
N = 3000
setups = []
for k in range(2):
    tau = rand_fr_ints(1, S0 + 10 + k)[0]
    pw = [pow(tau, i, R_MOD) for i in range(N)]
    pts = hip.g1_mul_batch(g1, mont(pw))
    srs = hip.srs_g1_upload(pts)
    if k == 0:
        hip.srs_g1_precompute(srs)
    setups.append({"tau": tau, "pts": pts, "srs": srs, "tau_g2": hip.g2_mul_batch(g2, mont([tau]))[0]})
coms = [hip.g1_mul_batch(g1, mont(rand_fr_ints(1, S0 + 20 + k)))[0] for k in range(3)]
bad = 0


def poly_eval(c, z):
    acc = 0
    for x in reversed(c):
        acc = (acc * z + x) % R_MOD
    return acc


def jac_aff(j):
    return jac_to_affine_words(j)


for it in range(rounds):
    op = ["msm", "encap", "decap", "open", "verify", "pairing"][int(rng.integers(0, 6))]
    st = setups[int(rng.integers(0, 2))]
    ok = True
    if op == "msm":
        n = int(rng.integers(1, N + 1))
        sc = mont(rand_fr_ints(n, S0 + 1000 + it))
        got = jac_aff(hip.msm_g1(st["srs"], sc))
        ok = np.array_equal(got, oc.msm_g1(st["pts"][:n], sc, threads=8))
        desc = "n=%d" % n
    elif op in ("encap", "decap"):
        mode = rng.random()
        gt_opt = -1 if mode < 0.3 else (64 if mode < 0.7 else 1000000000)      # -1: the library's own policy (GT path once a commitment repeats)
        hip.set_option("encap_gt", gt_opt)
        n = int([1, 63, 100, 300, 700][int(rng.integers(0, 5))])
        com = coms[int(rng.integers(0, 3))]
        A, V, Rr = (mont(rand_fr_ints(n, S0 + 2000 + 3 * it + k)) for k in range(3))
        ml = int(rng.integers(1, 80))
        ct, gt, key = hip.encap_batch(com, st["tau_g2"], A, V, Rr, ml)
        ect, egt, ekey = oc.encap_batch(com, st["tau_g2"], A, V, Rr, ml, threads=8)
        ok = np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey)
        if op == "decap":
            proofs = hip.g1_mul_batch(g1, mont(rand_fr_ints(n, S0 + 3000 + it)))
            dgt, dkey = hip.decap_batch(proofs, ct, ml)
            egt2, ekey2 = oc.decap_batch(proofs, ect, ml, threads=8)
            ok = ok and np.array_equal(dgt, egt2) and np.array_equal(dkey, ekey2)
        desc = "n=%d gt=%s" % (n, "auto" if gt_opt < 0 else gt_opt)
    elif op in ("open", "verify"):
        n = int(rng.integers(1, 400))
        c = rand_fr_ints(n, S0 + 4000 + it)
        z = rand_fr_ints(1, S0 + 5000 + it)[0]
        proof, val = hip.kzg_open(st["srs"], mont(c), mont([z])[0])
        v = poly_eval(c, z)
        # quotient by synthetic division, committed with the oracle
        q = [0] * (n - 1)
        acc = 0
        for i in range(n - 1, 0, -1):
            acc = (c[i] + acc * z) % R_MOD
            q[i - 1] = acc
        exp = oc.msm_g1(st["pts"][:n - 1], mont(q), threads=8) if n > 1 else np.zeros(8, np.uint64)
        pa = jac_aff(proof)
        ok = oc.limbs_to_ints(oc.fr_from_mont(val.reshape(1, 4)))[0] == v and np.array_equal(pa, exp)
        desc = "n=%d" % n
        if op == "verify":
            com = oc.msm_g1(st["pts"][:n], mont(c), threads=8)
            good = hip.kzg_verify(com, st["tau_g2"], mont([z])[0], mont([v])[0], pa)
            other = setups[1 - setups.index(st)]
            wrong_v = hip.kzg_verify(com, st["tau_g2"], mont([z])[0], mont([(v + 1) % R_MOD])[0], pa)
            wrong_s = hip.kzg_verify(com, other["tau_g2"], mont([z])[0], mont([v])[0], pa)
            ok = ok and good and not wrong_v and (n == 1 or not wrong_s)
    else:
        n = int(rng.integers(1, 40))
        P = hip.g1_mul_batch(g1, mont(rand_fr_ints(n, S0 + 6000 + it)))
        Q = hip.g2_mul_batch(g2, mont(rand_fr_ints(n, S0 + 7000 + it)))
        ok = np.array_equal(hip.pairing_batch(P, Q), oc.pairing_batch(P, Q, threads=8))
        desc = "n=%d" % n
    bad += not ok
    print("%3d %-8s %-18s %s" % (it, op, desc, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
