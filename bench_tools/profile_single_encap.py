"""Eight single `encapsulate` calls to one commitment, for a rocprofv3 kernel trace (bench_tools/kernel_timeline.py prints the last one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keaki_amd import keaki as K
rng = K.Rng(1)
s = K.KZGSetup.setup(rng.fr_rand(), 16)
p = np.stack([rng.fr_rand() for _ in range(8)])
com = K.commit(s, p); z = rng.fr_rand(); v = K.poly_evaluate(p, z)
for _ in range(8):
    K.encapsulate(rng, s, com, z, v, 32)
