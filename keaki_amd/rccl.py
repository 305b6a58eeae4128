"""ctypes binding of libkeaki_hip_rccl.so (include/keaki_hip_rccl.h): the RCCL exchanges of the one-process-per-GPU form for callers that
are not inside PyTorch. Plumbing only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .hip import KeakiHip, KeakiHipError, load_library

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = ["keaki_hip_rccl_unique_id", "keaki_hip_rccl_create", "keaki_hip_rccl_destroy", "keaki_hip_rccl_last_error", "keaki_hip_rccl_msm_g1",
           "keaki_hip_rccl_all_to_all", "keaki_hip_rccl_all_gather", "keaki_hip_rccl_collective_status"]


def load_rccl_library():
    global _LIB
    if _LIB is None:
        load_library()
        path = os.path.join(_HERE, "libkeaki_hip_rccl.so")
        if not os.path.exists(path):
            raise KeakiHipError(-100, "%s not built (make -C keaki_amd/csrc)" % path)
        lib = C.CDLL(path)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
        lib.keaki_hip_rccl_unique_id.argtypes = [vp]
        lib.keaki_hip_rccl_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        lib.keaki_hip_rccl_destroy.argtypes = [vp]
        lib.keaki_hip_rccl_destroy.restype = None
        lib.keaki_hip_rccl_last_error.argtypes = [vp]
        lib.keaki_hip_rccl_last_error.restype = C.c_char_p
        lib.keaki_hip_rccl_msm_g1.argtypes = [vp, vp, vp, sz, vp]
        lib.keaki_hip_rccl_all_to_all.argtypes = [vp, vp, vp, sz]
        lib.keaki_hip_rccl_all_gather.argtypes = [vp, vp, vp, sz]
        lib.keaki_hip_rccl_collective_status.argtypes = [vp, C.POINTER(i32)]
        _LIB = lib
    return _LIB


def unique_id() -> bytes:
    """rank 0: 128 bytes to hand to every other rank"""
    lib = load_rccl_library()
    buf = (C.c_uint8 * 128)()
    st = lib.keaki_hip_rccl_unique_id(buf)
    if st != 0:
        raise KeakiHipError(st, lib.keaki_hip_rccl_last_error(None).decode())
    return bytes(buf)


class KeakiRccl:
    def __init__(self, hip: KeakiHip, uid: bytes, rank: int, world: int):
        self.lib, self.hip = load_rccl_library(), hip
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        st = self.lib.keaki_hip_rccl_create(hip.ctx, buf, rank, world, C.byref(h))
        if st != 0:
            raise KeakiHipError(st, self.lib.keaki_hip_rccl_last_error(None).decode())
        self.h, self.rank, self.world = h, rank, world

    def _ck(self, st):
        if st != 0:
            raise KeakiHipError(st, self.lib.keaki_hip_rccl_last_error(self.h).decode())

    def msm_g1(self, srs_chunk, d_scalars: int, n: int, d_out: int):
        self._ck(self.lib.keaki_hip_rccl_msm_g1(self.h, srs_chunk.handle, C.c_void_p(d_scalars), n, C.c_void_p(d_out)))

    def collective_status(self) -> int:
        """Waits for the stream; -1 when every rank's share of the last msm_g1 succeeded, else raises with the first failing rank."""
        bad = C.c_int32(-1)
        self._ck(self.lib.keaki_hip_rccl_collective_status(self.h, C.byref(bad)))
        return bad.value

    def all_to_all(self, d_send: int, d_recv: int, per_peer: int):
        self._ck(self.lib.keaki_hip_rccl_all_to_all(self.h, C.c_void_p(d_send), C.c_void_p(d_recv), per_peer))

    def all_gather(self, d_send: int, d_recv: int, per_rank: int):
        self._ck(self.lib.keaki_hip_rccl_all_gather(self.h, C.c_void_p(d_send), C.c_void_p(d_recv), per_rank))

    def close(self):
        if getattr(self, "h", None):
            self.lib.keaki_hip_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
