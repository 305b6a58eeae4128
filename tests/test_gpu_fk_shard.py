"""FK23 sharded over R ranks (keaki_hip_fk_shard_*, kzg::open_fk of the reference src/kzg.rs:157-203 on N GPUs) against the un-sharded
keaki_hip_open_fk_poly (itself pinned to the oracle) and, at d = 8, the oracle's literal restatement.

All R ranks are played by ONE process here (R handles on one context); the exchanges the caller owes between the steps -- RCCL
all-to-all / all-gather in production (keaki_amd/dist.py::ShardedFk; run across two processes by tests/test_gpu_world2.py) -- are done
through host memory exactly as tests/fk_shard_model.py::all_to_all describes them."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import make_points_g1, mont

pytestmark = pytest.mark.gpu


class DevMem:
    """raw device buffers for the test (the ABI's *_dev entries take plain device pointers)"""

    def __init__(self):
        from keaki_amd.hip import load_library
        load_library()                                  # the process's ONE HIP runtime is loaded by now (PyTorch's copy when PyTorch is installed)
        self.rt = C.CDLL("libamdhip64.so.7")            # by SONAME: the loader hands back the copy that is already in the process
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), nbytes) == 0
        self.ptrs.append(p)
        return p.value

    def get(self, ptr, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), nbytes, 2) == 0      # device to host
        return out

    def put(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        assert self.rt.hipMemcpy(C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1) == 0  # host to device

    def free(self):
        for p in self.ptrs:
            self.rt.hipFree(p)
        self.ptrs = []


def all_to_all(hip, mem, send, recv, per_peer):
    R = len(send)
    hip.synchronize()
    data = [mem.get(send[r], R * per_peer).reshape(R, per_peer) for r in range(R)]
    for r in range(R):
        mem.put(recv[r], np.concatenate([data[s][r] for s in range(R)]))


def sharded_open_all_ranks(hip, mem, srs, log2d, R, p_mont, roots, n_polys=1):
    fks = [hip.fk_shard_create(srs, log2d, r, R, *roots) for r in range(R)]
    try:
        buf, a2a_big, a2a_small, gather = fks[0].sizes
        send = [mem.alloc(buf) for _ in range(R)]
        recv = [mem.alloc(buf) for _ in range(R)]
        for r in range(R):
            hip.fk_shard_setup(fks[r], 0, send[r], 0)
        all_to_all(hip, mem, send, recv, a2a_big)
        for r in range(R):
            hip.fk_shard_setup(fks[r], 1, 0, recv[r])
        outs = []
        for p in p_mont[:n_polys]:
            for r in range(R):
                hip.fk_shard_open(fks[r], 0, send[r], 0, coeffs=p)
            all_to_all(hip, mem, send, recv, a2a_small)
            for r in range(R):
                hip.fk_shard_open(fks[r], 1, send[r], recv[r])
            all_to_all(hip, mem, send, recv, a2a_small)
            for r in range(R):
                hip.fk_shard_open(fks[r], 2, send[r], recv[r])
            hip.synchronize()
            allp = np.concatenate([mem.get(send[r], gather) for r in range(R)])       # the all-gather, in rank order
            per_rank = []
            for r in range(R):
                mem.put(recv[r], allp)
                per_rank.append(hip.fk_shard_open(fks[r], 3, 0, recv[r]))
            for r in range(1, R):
                assert np.array_equal(per_rank[r], per_rank[0])
            outs.append(per_rank[0])
        return outs
    finally:
        for fk in fks:
            fk.free()


@pytest.mark.parametrize("log2d,R", [(2, 2), (3, 2), (4, 4), (6, 8), (8, 2), (10, 4), (13, 8)])
def test_sharded_fk_equals_unsharded(oc, py, hip, rand_fr, log2d, R):
    d = 1 << log2d
    ks, pts = make_points_g1(oc, hip, d, 1800 + log2d)
    ps = [mont(oc, rand_fr(d, 1810 + 7 * log2d + t)) for t in range(2)]
    w2 = py.fr_root_of_unity(2 * d)
    roots = [mont(oc, [x])[0] for x in (w2, pow(w2, -1, py.R), pow(2 * d, -1, py.R))]
    srs = hip.srs_g1_upload(pts)
    mem = DevMem()
    try:
        got = sharded_open_all_ranks(hip, mem, srs, log2d, R, ps, roots, n_polys=2)        # the second polynomial reuses hat_s
        for p, g in zip(ps, got):
            assert np.array_equal(g, hip.open_fk_poly(srs, log2d, p, *roots))
        if d <= 8:
            g1p = oc.g1_to_ints(pts)
            p_int = rand_fr(d, 1810 + 7 * log2d)
            assert oc.g1_to_ints(got[0]) == py.kzg_open_fk(g1p[:d], p_int)
    finally:
        mem.free()
        srs.free()


def test_sharded_fk_argument_errors(oc, py, hip, rand_fr):
    from keaki_amd.hip import KeakiHipError
    d = 16
    ks, pts = make_points_g1(oc, hip, d, 1900)
    w2 = py.fr_root_of_unity(2 * d)
    roots = [mont(oc, [x])[0] for x in (w2, pow(w2, -1, py.R), pow(2 * d, -1, py.R))]
    srs = hip.srs_g1_upload(pts)
    try:
        for rank, world, log2d in [(0, 3, 4), (2, 2, 4), (0, 8, 4), (0, 1, 4), (0, 2, 5)]:     # not a power of two, rank out of range, d < world^2, world 1, d > SRS
            with pytest.raises(KeakiHipError):
                hip.fk_shard_create(srs, log2d, rank, world, *roots)
        fk = hip.fk_shard_create(srs, 4, 0, 2, *roots)
        try:
            with pytest.raises(KeakiHipError):                     # open before setup
                hip.fk_shard_open(fk, 0, 1, 0, coeffs=mont(oc, rand_fr(d, 1901)))
            with pytest.raises(KeakiHipError):                     # setup step 1 before step 0
                hip.fk_shard_setup(fk, 1, 0, 1)
        finally:
            fk.free()
    finally:
        srs.free()
