"""One 2^24 MSM with tables on the DIAGNOSTIC build with the gathers confined to (mask + 1) table rows (0 = the shipped addresses): the workload of the counter
passes that separate the bucket kernel's ROW traffic from its INDEX-stream traffic (bench_tools/r6_pmc_bucket_mask.sh).  argv: mask [option:value ...]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import keaki_amd.hip as KH
if os.environ.get("KEAKI_PRODUCT_LIB") != "1":
    KH.lib_path = lambda: os.path.join(ROOT, "bench_tools", "diag", "libkeaki_hip.so")
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED
dev = torch.device("cuda", 0)
hip = KeakiHip(0)
n = 1 << 24
d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
d_k = torch.from_numpy(random_fr_limbs(n, SEED + 1).view(np.int64)).to(dev)
d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr()); hip.synchronize()
srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
hip.srs_g1_precompute(srs)
d_s = torch.from_numpy(random_fr_limbs(n, SEED + 2).view(np.int64)).to(dev)
d_out = torch.zeros(12, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
mask = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
if mask:
    hip.set_option("diag_row_mask", mask)
for a in sys.argv[2:]:
    k, v = a.split(":"); hip.set_option(k, int(v))
for _ in range(2):
    hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
hip.synchronize()
print("done mask=%#x" % mask)
