"""commit / open latency against the polynomial length on a small setup (BASELINE config 1: degree 128), one burst per call under rocprofv3."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keaki_amd import keaki as K
rng = K.Rng(1)
for size in (129, 1024):
    s = K.KZGSetup.setup(rng.fr_rand(), size)
    for n in (1, 8, 64, 129, 1000):
        if n > size: continue
        p = np.stack([rng.fr_rand() for _ in range(n)])
        K.commit(s, p); K.commit(s, p)
        t0 = time.perf_counter()
        for _ in range(5): K.commit(s, p)
        t = (time.perf_counter() - t0) / 5
        print("setup %4d: commit(%4d) %.3f ms" % (size, n, t * 1e3), flush=True)
        time.sleep(0.08)
