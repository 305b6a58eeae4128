"""Four encap_batch calls of 2^16 items, each to a commitment the context has not seen (the `kem.encaps_per_s` case), device-resident:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/etrace -- python3 bench_tools/trace_encap_fresh.py
then bench_tools/trace_vec_timeline.py gpurun_out/etrace shows the kernels of the last call."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs
n = 1 << 16
h = KeakiHip(0)
from oracle import bn254_py as py  # generators only (test / bench infrastructure)
def limbs(x): return np.frombuffer(int(x).to_bytes(32, "little"), np.uint64)
R256 = 1 << 256
g1 = np.concatenate([limbs(py.G1_GEN[0] * R256 % py.P), limbs(py.G1_GEN[1] * R256 % py.P)])
g2 = np.concatenate([limbs(c * R256 % py.P) for c in (py.G2_GEN[0][0], py.G2_GEN[0][1], py.G2_GEN[1][0], py.G2_GEN[1][1])])
coms = h.g1_mul_batch(g1, random_fr_limbs(6, 5))
tau = h.g2_mul_batch(g2, random_fr_limbs(1, 6))[0]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
A, V, Rr = dev(random_fr_limbs(n, 1)), dev(random_fr_limbs(n, 2)), dev(random_fr_limbs(n, 3))
d_tau = dev(tau); ct = torch.empty(n * 16, dtype=torch.int64, device="cuda"); key = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
for k in range(5):
    d_com = dev(coms[k]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.encap_batch_dev(d_com.data_ptr(), d_tau.data_ptr(), A.data_ptr(), V.data_ptr(), Rr.data_ptr(), n, ct.data_ptr(), 0, key.data_ptr(), 32)
    h.synchronize(); t = time.perf_counter() - t0
    print("fresh commitment %d: %.2f ms = %.2f M encaps/s" % (k, t * 1e3, n / t / 1e6), flush=True)
    time.sleep(0.06)
