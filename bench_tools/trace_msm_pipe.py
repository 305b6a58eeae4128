"""One host-pointer keaki_hip_msm_g1 call at 2^24 (window tables), three times, for `rocprofv3 --kernel-trace --memory-copy-trace`:
the timeline (bench_tools/trace_vec_timeline.py <dir>) shows the scalar chunks' copies under the kernels of the chunk before.
    python3 bench_tools/trace_msm_pipe.py [log2n] [pinned 0/1]"""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pinned = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1 << log2n
dev = torch.device("cuda", 0)
hip = KeakiHip(0)
d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
d_k = torch.from_numpy(random_fr_limbs(n, SEED + 1).view(np.int64)).to(dev)
d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr()); hip.synchronize()
srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
hip.srs_g1_precompute(srs)
s = random_fr_limbs(n, SEED + 2)
src = torch.from_numpy(s.view(np.int64)).pin_memory() if pinned else None
ptr = src.data_ptr() if pinned else s.ctypes.data
out = np.zeros(12, np.uint64)
for i in range(3):
    time.sleep(0.1)                                   # a gap in the trace between the calls
    t0 = time.perf_counter()
    st = hip.lib.keaki_hip_msm_g1(hip.ctx, srs.handle, C.c_void_p(ptr), n, out.ctypes.data_as(C.c_void_p))
    assert st == 0
    print("call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
