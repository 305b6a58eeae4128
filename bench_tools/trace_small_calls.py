"""Kernel timelines of single calls through the host mirror (run under rocprofv3 --kernel-trace; each call is its own burst):
encapsulate to a repeating commitment, decapsulate, verify, encapsulate to a new commitment."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keaki_amd import keaki as K
rng = K.Rng(1)
s = K.KZGSetup.setup(rng.fr_rand(), 1024)
p = np.stack([rng.fr_rand() for _ in range(1000)])
com = K.commit(s, p); z = rng.fr_rand(); pr = K.open(s, p, z); v = K.poly_evaluate(p, z)
for _ in range(6): ct, key = K.encapsulate(rng, s, com, z, v, 32)
time.sleep(0.1); ct, key = K.encapsulate(rng, s, com, z, v, 32); print("MARK encapsulate same", flush=True)
time.sleep(0.1); K.decapsulate(s, pr, ct, 32); print("MARK decapsulate", flush=True)
time.sleep(0.1); K.verify(s, com, z, v, pr); print("MARK verify", flush=True)
com2 = K.commit(s, p[:8])
time.sleep(0.1); K.encapsulate(rng, s, com2, z, v, 32); print("MARK encapsulate new", flush=True)
time.sleep(0.1)
