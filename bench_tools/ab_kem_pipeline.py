"""A/B of the chunk pipeline of the host-pointer KEM batches (options kem_pipe_min / kem_pipe_chunk): wall clock of keaki_hip_encap_batch
(keys only, as vec_encrypt calls it) and keaki_hip_decap_batch at 2^LOG2N items from pageable numpy arrays, one piece vs chunks.
    python3 bench_tools/ab_kem_pipeline.py [LOG2N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
h = KeakiHip(0)
g1 = np.array(mont_words(1) + mont_words(2), np.uint64)
from bench import G2_GEN
g2 = np.array(sum((mont_words(c) for c in G2_GEN), []), np.uint64)
tau = h.g2_mul_batch(g2, random_fr_limbs(1, 11))[0]
com = h.g1_mul_batch(g1, random_fr_limbs(1, 12))[0]
el, vals, r = random_fr_limbs(n, 2), random_fr_limbs(n, 3), random_fr_limbs(n, 4)
proofs = np.ascontiguousarray(np.tile(h.g1_mul_batch(g1, random_fr_limbs(1 << 10, 5)), (n >> 10, 1)))
h.encap_prepare(tau, n)
ref = None
for name, pmin, chunk in (("one piece", -1, 1 << 17), ("chunks of 2^17", 1 << 18, 1 << 17), ("chunks of 2^16", 1 << 17, 1 << 16), ("chunks of 2^18", 1 << 19, 1 << 18)):
    h.set_option("kem_pipe_min", pmin); h.set_option("kem_pipe_chunk", chunk)
    te, td = [], []
    for rep in range(4):
        t0 = time.perf_counter(); ct, key = h.encap_batch(com, tau, el, vals, r, 32, want_gt=False); te.append(time.perf_counter() - t0)
    for rep in range(3):
        t0 = time.perf_counter(); dgt, dkey = h.decap_batch(proofs, ct, 32); td.append(time.perf_counter() - t0)
    if ref is None:
        ref = (ct, key, dgt, dkey)
    same = all(np.array_equal(x, y) for x, y in zip(ref, (ct, key, dgt, dkey)))
    print("%-16s encap_batch (ct + key out) %6.1f ms (min of %d; all %s)   decap_batch (gt + key out) %6.1f ms (all %s)   same bytes: %s"
          % (name, min(te) * 1e3, len(te), ["%.1f" % (x * 1e3) for x in te], min(td) * 1e3, ["%.1f" % (x * 1e3) for x in td], same))
