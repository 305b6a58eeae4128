"""What ONE rank of an N-GPU job spends in the sharded FK23 openings (kzg::open_fk, reference src/kzg.rs:157-203; keaki_hip_fk_shard_*),
measured on one GPU: rank 0's steps are run for world = 2, 4, 8 with the exchanges replaced by a device-to-device copy of the same
number of bytes (the steps' run time does not depend on the values: the butterflies use fixed signed windows), next to the un-sharded
call. The xGMI time of the real exchanges is NOT in these numbers; the bytes per rank are printed beside them.

    python bench_tools/bench_fk_shard.py [log2d ...]          (default 16 20)
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keaki_amd import keaki as K                      # noqa: E402
from keaki_amd.hip import KeakiHip, load_library     # noqa: E402

rt = C.CDLL("libamdhip64.so")
rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
rt.hipFree.argtypes = [C.c_void_p]


def dmalloc(n):
    p = C.c_void_p()
    assert rt.hipMalloc(C.byref(p), n) == 0
    return p


def main():
    hip = KeakiHip(0)
    for log2d in [int(x) for x in sys.argv[1:]] or [16, 20]:
        d = 1 << log2d
        rng = K.Rng(5)
        s = K.KZGSetup.setup(rng.fr_rand(), d)
        p = np.stack([rng.fr_rand() for _ in range(d)])
        K.open_fk(s, p, d)
        t0 = time.time(); ref = K.open_fk(s, p, d); t_plain = time.time() - t0
        # the same SRS for the raw ABI: g1 powers as the host mirror holds them
        srs = hip.srs_g1_upload(s.g1_pow())
        el = K.domain_elements(2 * d)
        inv = np.zeros(4, np.uint64)
        K._lib().keaki_host_fr_inv(K._p(K.fr(2 * d)), K._p(inv))
        dom = (el[1], el[2 * d - 1], inv)
        row = {"log2d": log2d, "unsharded_s": round(t_plain, 4), "per_rank": {}}
        for world in (2, 4, 8):
            fk = hip.fk_shard_create(srs, log2d, 0, world, *dom)
            buf, a2a_big, a2a_small, gather = fk.sizes
            send, recv = dmalloc(buf), dmalloc(buf)

            def exch(nbytes):
                hip.synchronize()
                rt.hipMemcpy(recv, send, nbytes, 3)

            t0 = time.time()
            hip.fk_shard_setup(fk, 0, send.value, 0); exch(world * a2a_big)
            hip.fk_shard_setup(fk, 1, 0, recv.value); hip.synchronize()
            t_setup = time.time() - t0
            best = None
            for _ in range(2):
                t0 = time.time()
                hip.fk_shard_open(fk, 0, send.value, 0, coeffs=p); exch(world * a2a_big)
                hip.fk_shard_open(fk, 1, send.value, recv.value); exch(world * a2a_small)
                hip.fk_shard_open(fk, 2, send.value, recv.value); hip.synchronize()
                rt.hipMemcpy(recv, send, gather, 3)
                hip.fk_shard_open(fk, 3, 0, recv.value)
                t = time.time() - t0
                best = t if best is None else min(best, t)
            row["per_rank"][str(world)] = {"hat_s_setup_s": round(t_setup, 4), "open_s": round(best, 4), "speedup_vs_unsharded": round(t_plain / best, 2),
                                           "bytes_sent_per_rank": (world - 1) * (a2a_big + a2a_small + gather)}
            fk.free(); rt.hipFree(send); rt.hipFree(recv)
        srs.free(); s.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
