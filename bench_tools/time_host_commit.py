"""kzg::commit as a host caller sees it: 2^log2n scalars in PAGEABLE host memory (a Rust Vec / numpy array) -> keaki_hip_msm_g1, against
the same call with the scalars already on the device."""
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
k = random_fr_limbs(n, SEED + 1)
d_k = torch.from_numpy(k.view(np.int64)).to(dev)
d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr()); hip.synchronize()
srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
hip.srs_g1_precompute(srs)
s = random_fr_limbs(n, SEED + 2)                       # pageable numpy memory
d_s = torch.from_numpy(s.view(np.int64)).to(dev)
d_out = torch.zeros(12, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr()); hip.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
hip.synchronize()
t_dev = (time.perf_counter() - t0) / 3
ref = d_out.cpu().numpy().view(np.uint64).copy()
out = hip.msm_g1(srs, s)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); out = hip.msm_g1(srs, s); ts.append(time.perf_counter() - t0)
assert np.array_equal(out, ref)
pinned = torch.from_numpy(s.view(np.int64)).pin_memory()
import ctypes as C
o2 = np.zeros(12, np.uint64)
tp = []
for _ in range(4):
    t0 = time.perf_counter()
    st = hip.lib.keaki_hip_msm_g1(hip.ctx, srs.handle, C.c_void_p(pinned.data_ptr()), n, o2.ctypes.data_as(C.c_void_p)); tp.append(time.perf_counter() - t0)
    assert st == 0
assert np.array_equal(o2, ref)
print("n=2^%d  scalars resident %.2f ms | pageable host scalars %.2f ms (min of %s) | pinned host scalars %.2f ms  -> %.0f MiB of scalars: pageable copy ~%.1f GB/s, pinned ~%.1f GB/s"
      % (log2n, t_dev * 1e3, min(ts) * 1e3, ["%.1f" % (x * 1e3) for x in ts], min(tp) * 1e3, n * 32 / 2**20, n * 32 / max(min(ts) - t_dev, 1e-9) / 1e9, n * 32 / max(min(tp) - t_dev, 1e-9) / 1e9))
