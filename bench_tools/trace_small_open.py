"""Kernel timeline of single commit / open calls (1000 coefficients) through the host mirror; run under rocprofv3 --kernel-trace."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keaki_amd import keaki as K
rng = K.Rng(1)
s = K.KZGSetup.setup(rng.fr_rand(), 1024)
p = np.stack([rng.fr_rand() for _ in range(1000)])
z = rng.fr_rand()
for _ in range(3): K.commit(s, p); K.open(s, p, z)
time.sleep(0.1); t0 = time.perf_counter(); K.commit(s, p); print("MARK commit %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
time.sleep(0.1); t0 = time.perf_counter(); K.open(s, p, z); print("MARK open %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
time.sleep(0.1)
