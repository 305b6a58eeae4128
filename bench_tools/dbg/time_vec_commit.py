"""Where Receiver::new's time goes at 2^20 choices: the pieces of vec_commit through the host mirror, one by one."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from keaki_amd import keaki as K
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
rng = K.Rng(2024)
deg = 1
while deg < n + 1: deg <<= 1
t0 = time.time(); s = K.KZGSetup.setup(rng.fr_rand(), deg); K.precompute_open_fk(s, deg); print("setup %.3f" % (time.time() - t0), flush=True)
bits = np.random.default_rng(7).integers(0, 2, n)
zero, one = K.fr(0), K.fr(1)
choices = np.where(bits[:, None] == 0, zero[None, :], one[None, :]).astype(np.uint64)
for rep in range(2):
    t0 = time.time(); com, proofs = K.vec_commit(K.Rng(5), s, choices); print("vec_commit %.3f" % (time.time() - t0), flush=True)
    t0 = time.time(); part, proofs2 = K.vec_commit_partial(K.Rng(5), s, choices, 0, 1); print("vec_commit_partial %.3f" % (time.time() - t0), flush=True)
padded = np.concatenate([choices, K.Rng(5).fr_rand()[None, :]], 0)
t0 = time.time(); coeffs = K.ifft(padded, padded.shape[0]); print("host ifft (C++ mirror, host loop) %.3f" % (time.time() - t0), flush=True)
t0 = time.time(); pr = K.open_fk(s, coeffs, deg); print("open_fk %.3f" % (time.time() - t0), flush=True)
t0 = time.time(); pr = K.open_fk(s, coeffs, deg); print("open_fk again %.3f" % (time.time() - t0), flush=True)
t0 = time.time(); c = K.commit(s, coeffs); print("commit %.3f" % (time.time() - t0), flush=True)
t0 = time.time(); c = K.commit(s, coeffs); print("commit again %.3f" % (time.time() - t0), flush=True)
