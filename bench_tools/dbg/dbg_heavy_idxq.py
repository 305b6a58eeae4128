import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as oc
from keaki_amd.hip import KeakiHip, jac_to_affine_words as _aff
oc.build()
hip = KeakiHip(0)
n = 1 << 17
g1, _ = oc.generators()
rng = np.random.default_rng(171)
k = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
k[:, 3] &= np.uint64((1 << 60) - 1)
pts = hip.g1_mul_batch(g1, k)
vals = np.where(rng.integers(0, 4, n) != 0, 1, rng.integers(2, 300, n)).astype(np.uint64)
vals[: n // 4] = 1
vals[n // 2: n // 2 + 30000] = rng.integers(2, 9, 30000)
sc = oc.fr_to_mont(np.concatenate([vals[:, None], np.zeros((n, 3), np.uint64)], 1))
exp = oc.g1_mul_batch(g1, oc.fr_dot(sc, k).reshape(1, 4))[0]
srs = hip.srs_g1_upload(pts)
for tables in (False, True):
    if tables:
        hip.srs_g1_precompute(srs)
    for u29 in (1, 0):
        hip.set_option("acc_u29", u29)
        for idxq in (1, 0):
            hip.set_option("acc_idxq", idxq)
            for chunks in (0, 2, 4):
                hip.set_option("msm_pipe_chunks", chunks)
                hip.set_option("msm_pipe_growth", 100)
                ok = [bool(np.array_equal(_aff(hip.msm_g1(srs, sc)), exp)) for _ in range(3)]
                print("tables", tables, "u29", u29, "idxq", idxq, "chunks", chunks, ok, flush=True)
