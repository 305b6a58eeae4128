"""What ONE rank of an N-GPU job spends in the sharded FK23 openings (kzg::open_fk, reference src/kzg.rs:157-203; keaki_hip_fk_shard_*),
measured on one GPU: all ranks of world = 2, 4, 8 are played one after the other (every step of every rank timed on its own, the
exchanges done as device-to-device copies outside the timed regions), the result is compared with the un-sharded call, and the slowest
rank's sum of steps is reported. The xGMI time of the real exchanges is NOT in these numbers; the bytes per rank are printed beside them.

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
            fks = [hip.fk_shard_create(srs, log2d, r, world, *dom) for r in range(world)]
            buf, a2a_big, a2a_small, gather = fks[0].sizes
            send = [dmalloc(buf) for _ in range(world)]
            recv = [dmalloc(buf) for _ in range(world)]
            tt = np.zeros((world, 6))            # per rank: setup 0, setup 1, open 0, 1, 2, 3

            def timed(r, col, fn):
                hip.synchronize(); t0 = time.time(); out = fn(); hip.synchronize(); tt[r, col] = time.time() - t0
                return out

            def all_to_all(per_peer):            # the real data movement, on one GPU: chunk r of rank q's send -> chunk q of rank r's recv
                for r in range(world):
                    for q in range(world):
                        rt.hipMemcpy(C.c_void_p(recv[r].value + q * per_peer), C.c_void_p(send[q].value + r * per_peer), per_peer, 3)
                rt.hipDeviceSynchronize()        # device-to-device copies do not block the host, and the library's stream does not wait for them

            for r in range(world):
                timed(r, 0, lambda: hip.fk_shard_setup(fks[r], 0, send[r].value, 0))
            all_to_all(a2a_big)
            for r in range(world):
                timed(r, 1, lambda: hip.fk_shard_setup(fks[r], 1, 0, recv[r].value))
            for r in range(world):
                timed(r, 2, lambda: hip.fk_shard_open(fks[r], 0, send[r].value, 0, coeffs=p))
            all_to_all(a2a_small)
            for r in range(world):
                timed(r, 3, lambda: hip.fk_shard_open(fks[r], 1, send[r].value, recv[r].value))
            all_to_all(a2a_small)
            for r in range(world):
                timed(r, 4, lambda: hip.fk_shard_open(fks[r], 2, send[r].value, recv[r].value))
            for r in range(world):
                for q in range(world):
                    rt.hipMemcpy(C.c_void_p(recv[r].value + q * gather), send[q], gather, 3)
            rt.hipDeviceSynchronize()
            ok = True
            out = np.zeros((d, 8), np.uint64)         # touched once: a caller's reused buffer, not fresh pages per call
            for r in range(world):
                timed(r, 5, lambda: hip.fk_shard_open(fks[r], 3, 0, recv[r].value, out=out))
                ok = ok and np.array_equal(out, ref)
            per_rank_open = tt[:, 2:].sum(axis=1)
            row["per_rank"][str(world)] = {"hat_s_setup_s": round(float(tt[:, :2].sum(axis=1).max()), 4), "open_s": round(float(per_rank_open.max()), 4),
                                           "open_steps_s": [round(float(x), 4) for x in tt[:, 2:].max(axis=0)],
                                           "speedup_vs_unsharded": round(t_plain / float(per_rank_open.max()), 2),
                                           "bytes_sent_per_rank": (world - 1) * (2 * a2a_small + gather), "equals_unsharded": bool(ok)}
            for fk in fks:
                fk.free()
            for m in send + recv:
                rt.hipFree(m)
        srs.free(); s.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
