import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as oc, bn254_py as py
from conftest import rand_fr_ints as rand_fr
from keaki_amd.hip import KeakiHip
hip = KeakiHip(0)
mont = lambda ints: oc.fr_to_mont(oc.ints_to_limbs(ints))
g1, g2 = oc.generators()
ks = [1, 2, 3, 255, 256, 4096, 4097, 8191, (1 << 26) - 1, (1 << 253) - 1, py.R - 1, py.R - 2, 1 << 248, 5, 7, 11, 1 << 12] + rand_fr(20, 5)
base = hip.g1_mul_batch(g1, mont([123456789]))[0]
for b, name in ((g1, "gen"), (base, "rand")):
    got = hip.g1_mul_batch(b, mont(ks)); exp = oc.g1_mul_batch(b, mont(ks))
    bad = [ks[i] for i in range(len(ks)) if not np.array_equal(got[i], exp[i])]
    print(name, "mismatching scalars:", [hex(k) for k in bad])
