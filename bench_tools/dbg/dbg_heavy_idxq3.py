import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as oc
from keaki_amd.hip import KeakiHip, jac_to_affine_words as _aff
oc.build()
hip = KeakiHip(0)
g1, _ = oc.generators()
n = 1 << 17
rng = np.random.default_rng(171)
k = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
k[:, 3] &= np.uint64((1 << 60) - 1)
pts = hip.g1_mul_batch(g1, k)
srs = hip.srs_g1_upload(pts)
hip.set_option("msm_pipe_chunks", 0)
def run(tag, vals):
    sc = oc.fr_to_mont(np.concatenate([vals[:, None].astype(np.uint64), np.zeros((n, 3), np.uint64)], 1))
    exp = oc.g1_mul_batch(g1, oc.fr_dot(sc, k).reshape(1, 4))[0]
    res = []
    for idxq in (1, 0):
        hip.set_option("acc_idxq", idxq)
        res.append(bool(np.array_equal(_aff(hip.msm_g1(srs, sc)), exp)))
    print("%-70s idxq=1: %s  idxq=0: %s" % (tag, res[0], res[1]), flush=True)
base = np.where(rng.integers(0, 4, n) != 0, 1, rng.integers(2, 300, n)).astype(np.int64)
v = base.copy(); v[: n // 4] = 1; v[n // 2: n // 2 + 30000] = rng.integers(2, 9, 30000)
run("the test's data", v)
v2 = base.copy(); v2[n // 2: n // 2 + 30000] = v[n // 2: n // 2 + 30000]
run("without the leading quarter of ones", v2)
v3 = base.copy(); v3[: n // 4] = 1
run("without the stretch", v3)
for L in (30000, 20000, 10000, 5000, 2000, 500):
    v4 = base.copy(); v4[n // 2: n // 2 + L] = rng.integers(2, 9, L)
    run("stretch of %d values 2..8" % L, v4)
v5 = base.copy(); v5[n // 2: n // 2 + 30000] = rng.integers(2, 300, 30000)
run("stretch of 30000 values 2..299", v5)
v6 = base.copy(); v6[n // 2: n // 2 + 30000] = rng.integers(2, 66, 30000)
run("stretch of 30000 values 2..65", v6)
v7 = np.ones(n, np.int64); v7[n // 2: n // 2 + 30000] = rng.integers(2, 9, 30000)
run("ones + stretch only", v7)
v8 = rng.integers(300, 1000, n); v8[n // 2: n // 2 + 30000] = rng.integers(2, 9, 30000)
run("300..999 uniform + stretch (no heavy)", v8)
