"""Board power and clocks (rocm-smi, sampled from a side thread) while one kernel family runs back to back for a few seconds:
idle, the 2^24 MSM with tables, 2^16 pairings. Says whether the shader clock under the bucket kernel (profiles/r05_ubench_clock.txt) is a power cap."""
import os, sys, time, subprocess, threading, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED, G2_GEN
dev = torch.device("cuda", 0)
hip = KeakiHip(0)

def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showtemp", "-d", "0"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return "rocm-smi failed: %r" % e
    keep = [l.strip() for l in out.split("\n") if re.search(r"Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)", l)]
    return " | ".join(re.sub(r"\s+", " ", l.split("GPU[0]")[-1]) for l in keep)

def under(name, fn, seconds=6.0):
    stop = [False]; lines = []
    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            lines.append(smi()); time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        fn(); hip.synchronize(); n += 1
    stop[0] = True; th.join()
    print("== %s (%d rounds in %.1f s)" % (name, n, time.time() - t0))
    for l in lines[:8]: print("  ", l)
    sys.stdout.flush()

print(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower", "-d", "0"], capture_output=True, text=True).stdout)
under("idle", lambda: time.sleep(0.2), 3.0)
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
    for _ in range(8): hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
under("2^24 MSM with tables, back to back", msm)
m = 1 << 16
g2w = []
for c in G2_GEN: g2w += mont_words(c)
d_g2 = torch.from_numpy(np.array(g2w, np.uint64).view(np.int64)).to(dev)
d_q = torch.empty((m, 16), dtype=torch.int64, device=dev)
hip.g2_mul_batch_dev(d_g2.data_ptr(), 0, d_k.data_ptr(), m, d_q.data_ptr()); hip.synchronize()
d_gt = torch.empty((m, 48), dtype=torch.int64, device=dev)
def pair():
    for _ in range(8): hip.pairing_batch_dev(d_pts.data_ptr(), d_q.data_ptr(), 1, m, d_gt.data_ptr())
under("2^16 pairings, back to back", pair)
