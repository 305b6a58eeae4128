"""Timing of vec_commit's FK23 openings (kzg::open_fk) through the host mirror, d = 2^k."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd import keaki as K
for log2d in [int(x) for x in sys.argv[1:]] or [10, 12, 14]:
    d = 1 << log2d
    rng = K.Rng(5)
    t0 = time.time(); s = K.KZGSetup.setup(rng.fr_rand(), d); t_setup = time.time() - t0
    p = np.stack([rng.fr_rand() for _ in range(d)])
    t0 = time.time(); proofs = K.open_fk(s, p, d); t_first = time.time() - t0   # includes hat_s = DFT(SRS), cached afterwards
    t0 = time.time(); proofs = K.open_fk(s, p, d); t_fk = time.time() - t0
    el = K.domain_elements(d)
    i = d // 3
    ok = np.array_equal(proofs[i], K.open(s, p, el[i]))
    print("d=2^%d setup %.2fs open_fk first %.3fs, cached hat_s %.3fs (%.0f proofs/s) spot-check vs open: %s" % (log2d, t_setup, t_first, t_fk, d / t_fk, ok), flush=True)
    s.close()
