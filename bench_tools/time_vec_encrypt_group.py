"""vec_encrypt / vec_decrypt of 2^20 items through the host mirror on ONE context against a device group of two / three contexts on the SAME GPU
(the members run on threads and streams of their own: one member's copies overlap another's kernels)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd import keaki as K
from bench import random_fr_limbs

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
el = random_fr_limbs(n, 2); vals = random_fr_limbs(n, 3)
msgs = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
ref = None
for members in ([0], [0, 0], [0, 0, 0], [0, 0, 0, 0]):
    dev = K.Device(members if len(members) > 1 else 0)
    rng = K.Rng(2024)
    s = K.KZGSetup.setup(rng.fr_rand(), 1 << 10, device=dev)
    com = K.commit(s, random_fr_limbs(100, 1))
    proofs = K.open_fk(s, random_fr_limbs(1 << 10, 4), 1 << 10)
    pr = np.ascontiguousarray(np.tile(proofs, (n // (1 << 10), 1)))
    te, td = [], []
    for rep in range(4):
        r2 = K.Rng(77)
        t0 = time.perf_counter(); g2, body = K.vec_encrypt_arrays(r2, s, com, el, vals, msgs); te.append(time.perf_counter() - t0)
    for rep in range(3):
        t0 = time.perf_counter(); out = K.vec_decrypt_arrays(s, pr, g2, body); td.append(time.perf_counter() - t0)
    if ref is None: ref = (g2.copy(), body.copy(), out.copy())
    same = np.array_equal(g2, ref[0]) and np.array_equal(body, ref[1]) and np.array_equal(out, ref[2])
    print("members %s: vec_encrypt %.1f ms (min of %d), vec_decrypt %.1f ms, bytes equal to one context: %s" % (members, min(te) * 1e3, len(te), min(td) * 1e3, same), flush=True)
    dev.close()
