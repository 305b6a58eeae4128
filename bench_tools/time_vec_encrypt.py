"""Where the wall clock of vec_encrypt / vec_decrypt at 2^20 items goes (host mirror over the host-pointer C ABI):
rng draws, the ABI call (H2D + kernels + D2H), the XOR; and the device-resident rate for comparison."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd import keaki as K
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
rng = K.Rng(2024)
s = K.KZGSetup.setup(rng.fr_rand(), 1 << 10)
com = K.commit(s, random_fr_limbs(100, 1))
el = random_fr_limbs(n, 2); vals = random_fr_limbs(n, 3)
msgs = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
t0 = time.perf_counter(); r = rng.fr_rand_many(n); t_rng = time.perf_counter() - t0
for rep in range(3):
    t0 = time.perf_counter(); g2, body = K.vec_encrypt_arrays(rng, s, com, el, vals, msgs); t_enc = time.perf_counter() - t0
proofs = K.open_fk(s, random_fr_limbs(1 << 10, 4), 1 << 10)
pr = np.ascontiguousarray(np.tile(proofs, (n // (1 << 10), 1)))
for rep in range(2):
    t0 = time.perf_counter(); out = K.vec_decrypt_arrays(s, pr, g2, body); t_dec = time.perf_counter() - t0
h = KeakiHip(0)
tau = s.tau_g2()
for rep in range(3):
    t0 = time.perf_counter(); ct, gt, key = h.encap_batch(com, tau, el, vals, r, 32); t_abi = time.perf_counter() - t0
t0 = time.perf_counter(); x = key ^ msgs; t_xor = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter(); dgt, dkey = h.decap_batch(pr, ct, 32); t_dabi = time.perf_counter() - t0
print("n=2^%d  rng draws %.1f ms | vec_encrypt (mirror) %.1f ms | encap_batch ABI call with host arrays (gt + key out) %.1f ms | numpy xor %.1f ms | "
      "vec_decrypt (mirror) %.1f ms | decap_batch ABI call (gt + key out) %.1f ms" % (log2n, t_rng * 1e3, t_enc * 1e3, t_abi * 1e3, t_xor * 1e3, t_dec * 1e3, t_dabi * 1e3))
