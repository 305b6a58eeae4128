"""Shader clock while the hot kernels run (bench_tools/clock_probe.hip: a one-wave probe on its own stream counts s_memtime ticks over a window of the
constant 100 MHz clock): idle, the 2^24 MSM, 2^16 pairings, FK23 at d = 2^20."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED, R_MOD, G2_GEN
dev = torch.device("cuda", 0)
hip = KeakiHip(0)
pr = C.CDLL(os.path.join(ROOT, "bench_tools", "libclock_probe.so"))
pr.probe_start.argtypes = [C.c_uint64]; pr.probe_wait.restype = C.c_double
def clock_during(fn, us):
    fn(); hip.synchronize()
    fn()                                  # the kernels are enqueued (asynchronous), the probe starts beside them
    time.sleep(0.0015)
    assert pr.probe_start(us) == 0
    mhz = pr.probe_wait()
    hip.synchronize()
    return mhz
print("idle: %.0f MHz" % clock_during(lambda: None, 3000))
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
def msm():
    for _ in range(3): hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
for _ in range(3):
    print("2^24 MSM with tables (window of 8 ms inside the bucket kernel of the first of three back-to-back MSMs): %.0f MHz" % clock_during(msm, 8000))
m = 1 << 16
g2w = []
for c in G2_GEN: g2w += mont_words(c)
d_g2 = torch.from_numpy(np.array(g2w, np.uint64).view(np.int64)).to(dev)
d_q = torch.empty((m, 16), dtype=torch.int64, device=dev)
hip.g2_mul_batch_dev(d_g2.data_ptr(), 0, d_k.data_ptr(), m, d_q.data_ptr()); hip.synchronize()
d_gt = torch.empty((m, 48), dtype=torch.int64, device=dev)
def pair():
    for _ in range(2): hip.pairing_batch_dev(d_pts.data_ptr(), d_q.data_ptr(), 1, m, d_gt.data_ptr())
for _ in range(2):
    print("2^16 pairings (window of 6 ms): %.0f MHz" % clock_during(pair, 6000))
# FK23 openings at d = 2^20 (the butterfly stages: ladders with per-lane window tables in memory, ~1.1 TB/s over the fabric)
import ctypes as C
lg = 20
d = 1 << lg
w2d = pow(5, (R_MOD - 1) >> (lg + 1), R_MOD)
mont_fr = lambda v: np.frombuffer(((v << 256) % R_MOD).to_bytes(32, "little"), np.uint64).copy()
om, omi, inv2d = mont_fr(w2d), mont_fr(pow(w2d, -1, R_MOD)), mont_fr(pow(2 * d, -1, R_MOD))
fsrs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), d)
coeffs = random_fr_limbs(d, SEED + 4242)
hip._ck(hip.lib.keaki_hip_srs_g1_precompute_fk(hip.ctx, fsrs.handle, lg, om.ctypes.data_as(C.c_void_p)))
hip.open_fk_poly(fsrs, lg, coeffs, om, omi, inv2d)
import threading
for _ in range(2):
    res = {}
    def probe():
        time.sleep(0.08)                       # well inside the ~0.2 s of butterfly stages
        assert pr.probe_start(8000) == 0
        res["mhz"] = pr.probe_wait()
    th = threading.Thread(target=probe); th.start()
    hip.open_fk_poly(fsrs, lg, coeffs, om, omi, inv2d)
    th.join()
    print("FK23 openings, d = 2^20 (window of 8 ms inside the butterfly stages): %.0f MHz" % res["mhz"])
