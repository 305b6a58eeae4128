"""Sweep of the window target of the SRS window tables (option msm_c_shared) at a given size: time per MSM with the scalars resident.
   python3 bench_tools/sweep_table_window.py LOG2N [c_lo c_hi]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c_lo = int(sys.argv[2]) if len(sys.argv) > 2 else 13
c_hi = int(sys.argv[3]) if len(sys.argv) > 3 else 22
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
ref = None
for c in [0] + list(range(c_lo, c_hi + 1)):
    hip.set_option("msm_c_shared", c)
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        tb = hip.srs_g1_precompute(srs)
    except Exception as e:
        print("c=%d: %s" % (c, e)); srs.free(); continue
    for _ in range(3):
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
    hip.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
    hip.synchronize()
    t = (time.perf_counter() - t0) / reps
    got = d_out.cpu().numpy().copy()
    if ref is None:
        ref = got
    print("n=2^%d  c=%2d%s  windows=%2d  tables %6.1f MB  %.3f ms  same result: %s" % (log2n, hip.last_msm_stats()["window_bits"], " (auto)" if c == 0 else "", (254 + hip.last_msm_stats()["window_bits"] - 1) // hip.last_msm_stats()["window_bits"], tb / 2**20, t * 1e3, bool(np.array_equal(got, ref))), flush=True)
    srs.free()
