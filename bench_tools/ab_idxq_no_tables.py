"""A/B of option acc_idxq (index stream by 64-byte groups through LDS) on the generic path WITHOUT window tables (per-window bucket sets), 2^LOG2N points."""
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
srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
d_s = torch.from_numpy(random_fr_limbs(n, SEED + 2).view(np.int64)).to(dev)
d_out = torch.zeros(12, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for rnd in range(2):
    for q in (1, 0):
        hip.set_option("acc_idxq", q)
        for _ in range(2): hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
        hip.synchronize()
        hip.set_timing(True)
        ms, tot = [], []
        for _ in range(5):
            hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr()); hip.synchronize()
            st = hip.last_msm_stats(); ms.append(st["bucket_ms"]); tot.append(st["total_ms"])
        hip.set_timing(False)
        print("no tables 2^%d  acc_idxq=%d  c=%d  bucket kernel %.3f ms  MSM %.3f ms" % (log2n, q, st["window_bits"], np.mean(ms), np.mean(tot)), flush=True)
