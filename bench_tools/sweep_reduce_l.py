"""Sweep of one tuning option of the MSM (default reduce_l, the chunk length of the bucket reduction; 0 = automatic) at 2^LOG2N points with
window tables: time per MSM, result compared.   python3 bench_tools/sweep_reduce_l.py [LOG2N [v1,v2,... [OPTION]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log2n
dev = torch.device("cuda", 0)
hip = KeakiHip(0)
d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
d_k = torch.from_numpy(random_fr_limbs(n, SEED + 1).view(np.int64)).to(dev)
d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr()); hip.synchronize()
d_s = torch.from_numpy(random_fr_limbs(n, SEED + 2).view(np.int64)).to(dev)
d_out = torch.zeros(12, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
opt = sys.argv[3] if len(sys.argv) > 3 else "reduce_l"
srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
hip.srs_g1_precompute(srs)
ref = None
for L in [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 0):
    hip.set_option(opt, L)
    for _ in range(3):
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
    hip.synchronize()
    reps = 40
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
    hip.synchronize()
    t = (time.perf_counter() - t0) / reps
    got = d_out.cpu().numpy().copy()
    if ref is None:
        ref = got
    print("n=2^%d  %s=%3d  %.3f ms  same result: %s" % (log2n, opt, L, t * 1e3, bool(np.array_equal(got, ref))), flush=True)
srs.free()
