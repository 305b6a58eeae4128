"""Host-pointer keaki_hip_msm_g1 (kzg::commit's call): chunk count x growth of the chunked upload, pageable and pinned scalars.
usage: sweep_msm_pipe.py [log2n] [tables 0/1]"""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
tables = int(sys.argv[2]) if len(sys.argv) > 2 else 1
grid = sys.argv[3] if len(sys.argv) > 3 else "full"
n = 1 << log2n
dev = torch.device("cuda", 0)
hip = KeakiHip(0)
print(hip.version())
d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
k = random_fr_limbs(n, SEED + 1)
d_k = torch.from_numpy(k.view(np.int64)).to(dev)
d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr()); hip.synchronize()
srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
if tables:
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
pinned = torch.from_numpy(s.view(np.int64)).pin_memory()
o2 = np.zeros(12, np.uint64)


def call(ptr):
    st = hip.lib.keaki_hip_msm_g1(hip.ctx, srs.handle, C.c_void_p(ptr), n, o2.ctypes.data_as(C.c_void_p))
    assert st == 0, hip.lib.keaki_hip_last_error(hip.ctx)


def best(ptr, reps=4):
    call(ptr)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); call(ptr); ts.append(time.perf_counter() - t0)
    assert np.array_equal(o2, ref)
    return min(ts) * 1e3


print("n = 2^%d, tables %d: scalars resident %.2f ms (%.3e/s)" % (log2n, tables, t_dev * 1e3, n / t_dev))
cells = [(0, 100)]
if grid == "full":
    cells += [(c, g) for c in (2, 3, 4, 6, 8, 12, 16) for g in (100, 140, 200)]
elif grid == "fine":                                   # around the shipped 6 x 1.4
    cells += [(c, g) for c in (5, 6, 7, 8, 10) for g in (120, 130, 140, 150, 160, 170)]
elif grid == "auto":                                   # the automatic chunk count of this size against the growth
    cells += [(-1, g) for g in (120, 140, 150, 160, 170, 180, 200)]
else:
    cells += [(-1, 140)]
for c, g in cells:
    hip.set_option("msm_pipe_chunks", c)
    hip.set_option("msm_pipe_growth", g)
    tp, tg = best(pinned.data_ptr()), best(s.ctypes.data)
    print("chunks %3d growth %3d%%: pinned %.2f ms (%.3e/s) | pageable %.2f ms (%.3e/s)" % (c, g, tp, n / (tp * 1e-3), tg, n / (tg * 1e-3)), flush=True)
