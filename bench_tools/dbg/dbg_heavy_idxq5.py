import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as oc
import keaki_amd.hip as KH
KH.lib_path = lambda: os.path.join(ROOT, "bench_tools", "diag", "libkeaki_hip_idxqdbg.so")
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
v = rng.integers(300, 1000, n)
st = np.repeat(np.arange(2, 9), 3000); rng.shuffle(st)
v[n // 2: n // 2 + 21000] = st
sc = oc.fr_to_mont(np.concatenate([v[:, None].astype(np.uint64), np.zeros((n, 3), np.uint64)], 1))
exp = oc.g1_mul_batch(g1, oc.fr_dot(sc, k).reshape(1, 4))[0]
print("result ok:", bool(np.array_equal(_aff(hip.msm_g1(srs, sc)), exp)), flush=True)
hip.synchronize()
