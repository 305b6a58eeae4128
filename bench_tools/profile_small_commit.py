import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keaki_amd import keaki as K
rng = K.Rng(1)
s = K.KZGSetup.setup(rng.fr_rand(), 1024)
p = np.stack([rng.fr_rand() for _ in range(1000)])
for _ in range(5): K.commit(s, p)
