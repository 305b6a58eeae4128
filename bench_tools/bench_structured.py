"""MSM time on structured scalars (0/1 coefficients, all-equal) at 2^22: the heavy-bucket path."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, SEED
dev = torch.device("cuda", 0)
hip = KeakiHip(0, torch.cuda.current_stream(dev).cuda_stream)
n = 1 << int(sys.argv[1] if len(sys.argv) > 1 else 22)
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
w = lambda v, m: [((v << 256) % m >> (64 * i)) & (2**64 - 1) for i in range(4)]
g = torch.from_numpy(np.array(w(1, P) + w(2, P), np.uint64).view(np.int64)).to(dev)
k = torch.from_numpy(random_fr_limbs(n, SEED).view(np.int64)).to(dev)
pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
hip.g1_mul_batch_dev(g.data_ptr(), 0, k.data_ptr(), n, pts.data_ptr())
srs = hip.srs_g1_wrap_dev(pts.data_ptr(), n)
out = torch.zeros(12, dtype=torch.int64, device=dev)
one = np.array(w(1, R), np.uint64)
rng = np.random.default_rng(1)
cases = {"random": random_fr_limbs(n, SEED + 9), "bits01": np.where(rng.integers(0, 2, n)[:, None] == 1, one[None, :], 0).astype(np.uint64),
         "all_ones": np.repeat(one[None, :], n, 0), "small<2^16": np.concatenate([rng.integers(0, 1 << 16, (n, 1)).astype(np.uint64), np.zeros((n, 3), np.uint64)], 1)}
for name, sc in cases.items():
    if name == "small<2^16":   # canonical small integers -> Montgomery needs a product; use as raw residues instead (still highly structured: top limbs zero)
        pass
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to(dev)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        hip.msm_g1_dev(srs, d.data_ptr(), n, out.data_ptr()); torch.cuda.synchronize()
        dt = time.time() - t0
    print("%-12s n=2^%d  %.1f ms" % (name, n.bit_length() - 1, dt * 1e3), flush=True)
