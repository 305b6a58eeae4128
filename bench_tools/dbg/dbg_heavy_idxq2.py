import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as oc
from keaki_amd.hip import KeakiHip, jac_to_affine_words as _aff
oc.build()
hip = KeakiHip(0)
g1, _ = oc.generators()
def run(tag, n, vals):
    rng = np.random.default_rng(171)
    k = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    pts = hip.g1_mul_batch(g1, k)
    sc = oc.fr_to_mont(np.concatenate([vals[:, None].astype(np.uint64), np.zeros((n, 3), np.uint64)], 1))
    exp = oc.g1_mul_batch(g1, oc.fr_dot(sc, k).reshape(1, 4))[0]
    srs = hip.srs_g1_upload(pts)
    hip.set_option("msm_pipe_chunks", 0)
    res = []
    for idxq in (1, 0):
        hip.set_option("acc_idxq", idxq)
        res.append(bool(np.array_equal(_aff(hip.msm_g1(srs, sc)), exp)))
    print("%-60s n=%d idxq=1: %s  idxq=0: %s  window_bits %d" % (tag, n, res[0], res[1], hip.last_msm_stats()["window_bits"]), flush=True)
    srs.free()
rng = np.random.default_rng(5)
n = 1 << 17
run("all ones (one heavy bucket)", n, np.ones(n, np.int64))
run("values 2..300 uniform (no heavy, ~440 per bucket)", n, rng.integers(2, 300, n))
run("values 2..9 uniform (8 heavy buckets)", n, rng.integers(2, 10, n))
run("values 2..40 (3400 per bucket, not heavy)", n, rng.integers(2, 41, n))
run("values 2..20 (6900 per bucket, not heavy)", n, rng.integers(2, 21, n))
v = rng.integers(2, 300, n); v[: n // 2] = 1
run("half ones + 2..300", n, v)
v = rng.integers(2, 300, n); v[: 9000] = 1
run("9000 ones (just heavy) + 2..300", n, v)
v = rng.integers(2, 300, n); v[: 8000] = 1
run("8000 ones (not heavy) + 2..300", n, v)
run("values 1..2^13 uniform", n, rng.integers(1, 1 << 13, n))
for m in (1 << 15, 1 << 16, 40000, 1 << 18):
    v = rng.integers(2, 300, m); v[: m // 2] = 1
    run("half ones + 2..300", m, v)
