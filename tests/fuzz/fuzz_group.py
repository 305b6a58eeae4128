#!/usr/bin/env python3
"""Random SEQUENCES of calls on ONE device group (keaki_hip_group_*, N contexts on GPU 0) with every result checked against the CPU oracle:
commit's MSM over polynomials of random length (ranges of the members cut anywhere, incl. empty ranges), `open`, encap / decap batches of
random size on both encapsulation paths with commitments that repeat or change, FK23 openings, interleaved with option changes and trims on
single members (the members' workspaces and tables are independent state).   python tests/fuzz/fuzz_group.py [rounds [seed [members]]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import oracle as oc
import bn254_py as py
from conftest_helpers import rand_fr_ints, R_MOD
from keaki_amd.hip import KeakiHip, KeakiHipGroup, jac_to_affine_words

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
members = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(seed)
S0 = seed * 1000003
NCPU = os.cpu_count() or 1
mont = lambda ints: oc.fr_to_mont(oc.ints_to_limbs(ints))
g1, g2 = oc.generators()
one = KeakiHip(0)
g = KeakiHipGroup([0] * members)
N = 1 << 12
tau = rand_fr_ints(1, S0 + 10)[0]
pw, acc = [], 1
for _ in range(N):
    pw.append(acc); acc = acc * tau % R_MOD
pts = one.g1_mul_batch(g1, mont(pw))
tau_g2 = one.g2_mul_batch(g2, mont([tau]))[0]
srs = g.srs_g1_upload(pts, precompute=True)
coms = [one.g1_mul_batch(g1, mont(rand_fr_ints(1, S0 + 20 + k)))[0] for k in range(3)]
log2d = 10
w2 = py.fr_root_of_unity(2 << log2d)
fk = g.fk_create(pts, log2d, mont([w2])[0], mont([pow(w2, -1, R_MOD)])[0], mont([pow(2 << log2d, -1, R_MOD)])[0])
bad = 0


def member_ctx(i):
    g.lib.keaki_hip_group_ctx.restype = C.c_void_p
    return C.c_void_p(g.lib.keaki_hip_group_ctx(g.g, i))


for it in range(rounds):
    op = ["msm", "msm", "open", "encap", "decap", "fk", "knob"][int(rng.integers(0, 7))]
    ok, desc = True, ""
    if op == "msm":
        n = int([0, 1, N // members, N // members + 1, N - 1, N][int(rng.integers(0, 6))]) if rng.random() < 0.5 else int(rng.integers(0, N + 1))
        sc = mont(rand_fr_ints(n, S0 + 1000 + it)) if n else np.zeros((0, 4), np.uint64)
        exp = oc.msm_g1(pts[:n], sc, threads=NCPU) if n else np.zeros(8, np.uint64)
        ok = np.array_equal(jac_to_affine_words(g.msm_g1(srs, sc)), exp)
        desc = "n=%d" % n
    elif op == "open":
        n = int(rng.integers(1, N + 2))
        c = mont(rand_fr_ints(n, S0 + 2000 + it))
        z = mont(rand_fr_ints(1, S0 + 3000 + it))[0]
        proof, val = g.kzg_open(srs, c, z)
        q, v = oc.fr_quotient(c, z)
        exp = oc.msm_g1(pts[:n - 1], q, threads=NCPU) if n > 1 else np.zeros(8, np.uint64)
        ok = np.array_equal(val, v) and np.array_equal(jac_to_affine_words(proof), exp)
        desc = "n=%d" % n
    elif op in ("encap", "decap"):
        n = int([1, members - 1, members, 63, 300, 1500][int(rng.integers(0, 6))])
        n = max(n, 1)
        com = coms[int(rng.integers(0, 3))]
        A, V, Rr = (mont(rand_fr_ints(n, S0 + 4000 + 3 * it + k)) for k in range(3))
        ml = int(rng.integers(1, 70))
        ct, gt, key = g.encap_batch(com, tau_g2, A, V, Rr, ml)
        ect, egt, ekey = oc.encap_batch(com, tau_g2, A, V, Rr, ml, threads=NCPU)
        ok = np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey)
        if op == "decap":
            proofs = pts[rng.integers(0, N, n)]
            dgt, dkey = g.decap_batch(proofs, ct, ml)
            egt2, ekey2 = oc.decap_batch(proofs, ect, ml, threads=NCPU)
            ok = ok and np.array_equal(dgt, egt2) and np.array_equal(dkey, ekey2)
        desc = "n=%d msg_len=%d" % (n, ml)
    elif op == "fk":
        d = 1 << log2d
        p = mont(rand_fr_ints(d, S0 + 5000 + it))
        got = g.fk_open(fk, log2d, p)
        w = pow(w2, 2, R_MOD)
        for i in (0, int(rng.integers(1, d)), d - 1):
            q, _ = oc.fr_quotient(p, mont([pow(w, i, R_MOD)])[0])
            ok = ok and np.array_equal(got[i], oc.msm_g1(pts[:d - 1], q, threads=NCPU))
        desc = "d=%d" % d
    else:
        # state changes on ONE member between calls: its encapsulation policy, its MSM window, or all its workspaces gone
        i = int(rng.integers(0, members))
        what = int(rng.integers(0, 3))
        if what == 0:
            g.lib.keaki_hip_ctx_set_option(member_ctx(i), b"encap_gt", int([-1, 64, 1000000000][int(rng.integers(0, 3))]))
        elif what == 1:
            g.lib.keaki_hip_ctx_set_option(member_ctx(i), b"msm_c", int([0, 7, 11][int(rng.integers(0, 3))]))
        else:
            g.lib.keaki_hip_ctx_trim.argtypes = [C.c_void_p]
            g.lib.keaki_hip_ctx_trim(member_ctx(i))
        desc = "member %d change %d" % (i, what)
    bad += not ok
    print("%3d %-6s %-24s %s" % (it, op, desc, "ok" if ok else "MISMATCH"), flush=True)
g.fk_free(fk)
srs.free()
g.close(); one.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
