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
for nb, L in ((7, 1000), (7, 3000), (7, 4000), (7, 4090), (7, 4100), (7, 4285), (1, 4285), (1, 5000), (2, 5000), (1, 8000), (3, 8000), (64, 468), (64, 300), (30, 1000), (15, 2000)):
    v = rng.integers(300, 1000, n)
    m = nb * L
    st = np.repeat(np.arange(2, 2 + nb), L); rng.shuffle(st)
    v[n // 2: n // 2 + m] = st
    run("300..999 + stretch: %d buckets x %d entries (%d)" % (nb, L, m), v)
# the same entries NOT consecutive (spread over the vector)
v = rng.integers(300, 1000, n); idx = rng.choice(n, 30000, replace=False); v[idx] = rng.integers(2, 9, 30000)
run("300..999 + 30000 scattered values 2..8", v)
