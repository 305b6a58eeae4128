"""Three vec_encrypt calls of 2^20 items and nothing else after the setup: run under
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/vtrace -- python3 bench_tools/trace_vec_encrypt.py
and read the timeline of the LAST call with bench_tools/trace_vec_timeline.py."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd import keaki as K
from bench import random_fr_limbs
n = 1 << 20
rng = K.Rng(2024)
s = K.KZGSetup.setup(rng.fr_rand(), 1 << 10)
com = K.commit(s, random_fr_limbs(100, 1))
el = random_fr_limbs(n, 2); vals = random_fr_limbs(n, 3)
msgs = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter(); g2, body = K.vec_encrypt_arrays(rng, s, com, el, vals, msgs); t = time.perf_counter() - t0
    print("vec_encrypt %.1f ms" % (t * 1e3), flush=True)
    time.sleep(0.05)
import ctypes as C
from keaki_amd.keaki import _lib, _p, _u64, _ck
for rep in range(2):
    t0 = time.perf_counter()
    pts = np.ascontiguousarray(_u64(el, 4)[:n]); v2 = np.ascontiguousarray(_u64(vals, 4)[:n]); m2 = np.ascontiguousarray(msgs, dtype=np.uint8)
    t1 = time.perf_counter()
    a = np.zeros((n, 16), np.uint64); b = np.zeros((n, 32), np.uint8)
    t2 = time.perf_counter()
    _ck(_lib().keaki_host_vec_encrypt(rng.h, s.h, _p(_u64(com)), _p(pts), _p(v2), _p(m2), C.c_size_t(n), C.c_size_t(32), _p(a), _p(b)))
    t3 = time.perf_counter()
    del a, b
    t4 = time.perf_counter()
    print("python: views %.2f zeros %.2f call %.2f free %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
