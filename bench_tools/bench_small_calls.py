import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keaki_amd import keaki as K
rng = K.Rng(1)
s = K.KZGSetup.setup(rng.fr_rand(), 1024)
p = np.stack([rng.fr_rand() for _ in range(1000)])
com = K.commit(s, p); z = rng.fr_rand(); pr = K.open(s, p, z); v = K.poly_evaluate(p, z)
assert K.verify(s, com, z, v, pr)
for name, fn in (("commit(1000)", lambda: K.commit(s, p)), ("open(1000)", lambda: K.open(s, p, z)), ("verify", lambda: K.verify(s, com, z, v, pr))):
    fn(); t0 = time.perf_counter()
    for _ in range(5): fn()
    print(name, "%.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
# single KEM calls (src/kem.rs:13-50, 55-72 with n = 1)
ct, key = K.encapsulate(rng, s, com, z, v, 32)
assert K.decapsulate(s, pr, ct, 32) == key
# the third consecutive call to one commitment builds its GT tables, the fourth widens them: warm up past that
for name, fn in (("encapsulate (same commitment)", lambda: K.encapsulate(rng, s, com, z, v, 32)), ("decapsulate", lambda: K.decapsulate(s, pr, ct, 32))):
    for _ in range(4): fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    print(name, "%.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
coms = [K.commit(s, np.stack([rng.fr_rand() for _ in range(8)])) for _ in range(2)]
flip = [0]
def enc_alt():
    flip[0] += 1
    return K.encapsulate(rng, s, coms[flip[0] & 1], z, v, 32)
enc_alt(); t0 = time.perf_counter()
for _ in range(4): enc_alt()
print("encapsulate (commitment changes every call)", "%.2f ms" % ((time.perf_counter() - t0) / 4 * 1e3))
