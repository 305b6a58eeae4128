"""How much of the fixed-window ladder's time is its per-lane table in scratch memory? k_mul_batch<Fq> (the ladder of the FK23 butterflies with
per-lane twiddles) over 2^20 DIFFERENT points with (a) a different scalar per lane -- every lane indexes its table differently: a wave's
table read touches up to 8 x 45 rows of 256 B -- and (b) the SAME scalar in every lane: same instruction count, every table read is one row."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED
n = 1 << 20
dev = torch.device("cuda", 0)
hip = KeakiHip(0)
d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
d_k = torch.from_numpy(random_fr_limbs(n, SEED + 1).view(np.int64)).to(dev)
d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
d_out = torch.empty((n, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr()); hip.synchronize()
s_rand = random_fr_limbs(n, SEED + 2)
s_same = np.repeat(s_rand[:1], n, 0)
for name, s in (("different scalar per lane", s_rand), ("same scalar in every lane", s_same)):
    d_s = torch.from_numpy(np.ascontiguousarray(s).view(np.int64)).to(dev)
    torch.cuda.synchronize()
    hip.g1_mul_batch_dev(d_pts.data_ptr(), 1, d_s.data_ptr(), n, d_out.data_ptr()); hip.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        hip.g1_mul_batch_dev(d_pts.data_ptr(), 1, d_s.data_ptr(), n, d_out.data_ptr())
    hip.synchronize()
    print("%-28s %.2f ms per 2^20 scalar-mults" % (name, (time.perf_counter() - t0) / 5 * 1e3))
