"""ctypes binding of libkeaki_hip.so (include/keaki_hip.h). Plumbing only -- no arithmetic here.

Arrays are numpy uint64 in the C-ABI layouts (Montgomery limbs, see the header). The *_dev methods
take raw device pointers (ints), e.g. torch_tensor.data_ptr(), so callers can keep data resident
in HBM; torch itself is never imported here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = [
    "keaki_hip_ctx_create", "keaki_hip_ctx_destroy", "keaki_hip_last_error", "keaki_hip_synchronize", "keaki_hip_version",
    "keaki_hip_srs_g1_upload", "keaki_hip_srs_g1_wrap_dev", "keaki_hip_srs_g1_slice", "keaki_hip_srs_g1_len", "keaki_hip_srs_g1_precompute", "keaki_hip_srs_g1_free",
    "keaki_hip_srs_g2_upload", "keaki_hip_srs_g2_wrap_dev", "keaki_hip_srs_g2_precompute", "keaki_hip_srs_g2_free",
    "keaki_hip_msm_g1", "keaki_hip_msm_g1_dev", "keaki_hip_msm_g2", "keaki_hip_msm_g2_dev",
    "keaki_hip_g1_sum_dev", "keaki_hip_g1_sum",
    "keaki_hip_g1_mul_batch", "keaki_hip_g2_mul_batch", "keaki_hip_g1_mul_batch_dev", "keaki_hip_g2_mul_batch_dev",
    "keaki_hip_pairing_batch", "keaki_hip_pairing_batch_dev",
    "keaki_hip_encap_batch", "keaki_hip_encap_batch_dev", "keaki_hip_decap_batch", "keaki_hip_decap_batch_dev",
    "keaki_hip_encrypt_batch", "keaki_hip_encrypt_batch_dev", "keaki_hip_decrypt_batch", "keaki_hip_decrypt_batch_dev",
    "keaki_hip_group_encrypt_batch", "keaki_hip_group_decrypt_batch",
    "keaki_hip_selftest_field", "keaki_hip_open_fk", "keaki_hip_open_fk_poly", "keaki_hip_srs_g1_precompute_fk", "keaki_hip_fk_shard_create", "keaki_hip_fk_shard_free", "keaki_hip_fk_shard_sizes", "keaki_hip_fk_shard_setup", "keaki_hip_fk_shard_open", "keaki_hip_fr_fft", "keaki_hip_srs_g1_check", "keaki_hip_g2_check", "keaki_hip_kzg_open", "keaki_hip_kzg_verify", "keaki_hip_final_exp_batch", "keaki_hip_miller_loop_batch", "keaki_hip_g2_prepare", "keaki_hip_set_timing", "keaki_hip_last_msm_bucket_ms", "keaki_hip_last_msm_total_ms", "keaki_hip_last_msm_window_bits",
    "keaki_hip_last_fk_ms", "keaki_hip_ctx_stream", "keaki_hip_ctx_device", "keaki_hip_ctx_set_option", "keaki_hip_debug_set_alloc_limit", "keaki_hip_ctx_memory", "keaki_hip_ctx_trim", "keaki_hip_kzg_quotient", "keaki_hip_vec_commit", "keaki_hip_encap_prepare",
    "keaki_hip_group_create", "keaki_hip_group_destroy", "keaki_hip_group_size", "keaki_hip_group_ctx", "keaki_hip_group_last_error",
    "keaki_hip_group_peer_note", "keaki_hip_group_srs_g1_upload", "keaki_hip_group_srs_g1_len", "keaki_hip_group_srs_g1_has_tables", "keaki_hip_group_srs_g1_free", "keaki_hip_group_msm_g1",
    "keaki_hip_group_kzg_open", "keaki_hip_group_encap_batch", "keaki_hip_group_decap_batch",
    "keaki_hip_group_fk_create", "keaki_hip_group_fk_open", "keaki_hip_group_fk_free",
]

KEAKI_ERR_TOO_LARGE = -5


class KeakiHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"keaki_hip error {status}: {message}")
        self.status = status
        self.message = message


def lib_path() -> str:
    return os.path.join(_HERE, "libkeaki_hip.so")


def share_torch_runtime(names=("libamdhip64.so",)):    # the HIP runtime only: loading PyTorch's librccl ahead of torch aborts at exit
    """A process must end up with ONE HIP runtime. PyTorch wheels bundle their own copy (torch/lib/libamdhip64.so, the same SONAME as the
    system's): when torch is imported FIRST, libkeaki_hip.so binds to that copy and everything shares it; the other way round the process
    would hold two runtimes and torch fails with "No HIP GPUs are available". So, when PyTorch is installed but not loaded yet, its copy
    of the named libraries is loaded here (by path, without importing torch) before ours. KEAKI_HIP_RUNTIME=system keeps the system's
    runtime (a process that never imports torch does not care)."""
    import sys
    if os.environ.get("KEAKI_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in names:
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path)        # local scope is enough: later NEEDED entries resolve to a loaded object by its SONAME
            except OSError:
                pass        # an unusable bundle: the system's runtime serves


def load_library():
    """Loads libkeaki_hip.so. Raises (never falls back) when the HIP extension is missing."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise KeakiHipError(-100, f"{path} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                                      f"or `make -C keaki_amd/csrc`. There is no CPU fallback.")
        share_torch_runtime()
        lib = C.CDLL(path)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
        lib.keaki_hip_version.restype = C.c_char_p
        lib.keaki_hip_last_error.restype = C.c_char_p
        lib.keaki_hip_last_error.argtypes = [vp]
        lib.keaki_hip_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
        lib.keaki_hip_ctx_destroy.argtypes = [vp]
        lib.keaki_hip_ctx_destroy.restype = None
        lib.keaki_hip_synchronize.argtypes = [vp]
        for g in ("g1", "g2"):
            getattr(lib, f"keaki_hip_srs_{g}_upload").argtypes = [vp, vp, sz, C.POINTER(vp)]
            getattr(lib, f"keaki_hip_srs_{g}_wrap_dev").argtypes = [vp, vp, sz, C.POINTER(vp)]
            getattr(lib, f"keaki_hip_srs_{g}_free").argtypes = [vp, vp]
            getattr(lib, f"keaki_hip_srs_{g}_free").restype = None
            getattr(lib, f"keaki_hip_msm_{g}").argtypes = [vp, vp, vp, sz, vp]
            getattr(lib, f"keaki_hip_msm_{g}_dev").argtypes = [vp, vp, vp, sz, vp]
            getattr(lib, f"keaki_hip_{g}_mul_batch").argtypes = [vp, vp, i32, vp, sz, vp]
            getattr(lib, f"keaki_hip_{g}_mul_batch_dev").argtypes = [vp, vp, i32, vp, sz, vp]
        lib.keaki_hip_srs_g1_precompute.argtypes = [vp, vp, C.POINTER(C.c_size_t)]
        lib.keaki_hip_srs_g2_precompute.argtypes = [vp, vp, C.POINTER(C.c_size_t)]
        lib.keaki_hip_srs_g1_slice.argtypes = [vp, vp, sz, sz, C.POINTER(vp)]
        lib.keaki_hip_srs_g1_len.argtypes = [vp]
        lib.keaki_hip_srs_g1_len.restype = sz
        lib.keaki_hip_g1_sum.argtypes = [vp, vp, sz, vp]
        lib.keaki_hip_g1_sum_dev.argtypes = [vp, vp, sz, vp]
        lib.keaki_hip_pairing_batch.argtypes = [vp, vp, vp, i32, sz, vp]
        lib.keaki_hip_pairing_batch_dev.argtypes = [vp, vp, vp, i32, sz, vp]
        lib.keaki_hip_encap_batch.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz]
        lib.keaki_hip_encap_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz]
        lib.keaki_hip_decap_batch.argtypes = [vp, vp, vp, sz, vp, vp, sz]
        lib.keaki_hip_decap_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, sz]
        lib.keaki_hip_selftest_field.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        lib.keaki_hip_open_fk.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp, vp]
        lib.keaki_hip_open_fk_poly.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp, vp]
        lib.keaki_hip_fr_fft.argtypes = [vp, vp, C.c_uint32, vp, vp]
        lib.keaki_hip_srs_g1_precompute_fk.argtypes = [vp, vp, C.c_uint32, vp]
        lib.keaki_hip_fk_shard_create.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, C.POINTER(vp)]
        lib.keaki_hip_fk_shard_free.argtypes = [vp, vp]
        lib.keaki_hip_fk_shard_free.restype = None
        lib.keaki_hip_fk_shard_sizes.argtypes = [vp, C.POINTER(C.c_size_t)]
        lib.keaki_hip_fk_shard_setup.argtypes = [vp, vp, C.c_int32, vp, vp]
        lib.keaki_hip_fk_shard_open.argtypes = [vp, vp, C.c_int32, vp, vp, vp, vp]
        lib.keaki_hip_srs_g1_check.argtypes = [vp, vp, vp, vp]
        lib.keaki_hip_kzg_open.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, vp]
        lib.keaki_hip_g2_check.argtypes = [vp, vp, C.c_size_t, vp, vp]
        lib.keaki_hip_kzg_verify.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]
        lib.keaki_hip_final_exp_batch.argtypes = [vp, vp, sz, vp]
        lib.keaki_hip_miller_loop_batch.argtypes = [vp, vp, vp, sz, vp]
        lib.keaki_hip_set_timing.argtypes = [vp, i32]
        lib.keaki_hip_last_msm_bucket_ms.argtypes = [vp]
        lib.keaki_hip_last_msm_bucket_ms.restype = C.c_float
        lib.keaki_hip_last_msm_total_ms.argtypes = [vp]
        lib.keaki_hip_last_msm_total_ms.restype = C.c_float
        lib.keaki_hip_last_msm_window_bits.argtypes = [vp]
        lib.keaki_hip_last_fk_ms.argtypes = [vp, C.POINTER(C.c_float)]
        lib.keaki_hip_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        lib.keaki_hip_debug_set_alloc_limit.argtypes = [vp, sz]
        lib.keaki_hip_ctx_memory.argtypes = [vp, C.POINTER(C.c_size_t)]
        lib.keaki_hip_ctx_trim.argtypes = [vp]
        lib.keaki_hip_ctx_stream.argtypes = [vp]
        lib.keaki_hip_ctx_stream.restype = vp
        lib.keaki_hip_ctx_device.argtypes = [vp]
        lib.keaki_hip_ctx_device.restype = C.c_int32
        lib.keaki_hip_kzg_quotient.argtypes = [vp, vp, sz, vp, vp, vp]
        lib.keaki_hip_encap_prepare.argtypes = [vp, vp, sz]
        lib.keaki_hip_vec_commit.argtypes = [vp, vp, vp, sz, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
        lib.keaki_hip_group_create.argtypes = [C.POINTER(i32), sz, C.POINTER(vp)]
        lib.keaki_hip_group_destroy.argtypes = [vp]
        lib.keaki_hip_group_destroy.restype = None
        lib.keaki_hip_group_size.argtypes = [vp]
        lib.keaki_hip_group_size.restype = sz
        lib.keaki_hip_group_ctx.argtypes = [vp, sz]
        lib.keaki_hip_group_ctx.restype = vp
        lib.keaki_hip_group_last_error.argtypes = [vp]
        lib.keaki_hip_group_last_error.restype = C.c_char_p
        lib.keaki_hip_group_srs_g1_upload.argtypes = [vp, vp, sz, i32, C.POINTER(vp)]
        lib.keaki_hip_group_srs_g1_len.argtypes = [vp]
        lib.keaki_hip_group_srs_g1_len.restype = sz
        lib.keaki_hip_group_peer_note.argtypes = [vp]
        lib.keaki_hip_group_peer_note.restype = C.c_char_p
        lib.keaki_hip_group_srs_g1_has_tables.argtypes = [vp]
        lib.keaki_hip_group_srs_g1_has_tables.restype = C.c_int32
        lib.keaki_hip_group_srs_g1_free.argtypes = [vp, vp]
        lib.keaki_hip_group_srs_g1_free.restype = None
        lib.keaki_hip_group_msm_g1.argtypes = [vp, vp, vp, sz, vp]
        lib.keaki_hip_group_kzg_open.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        lib.keaki_hip_group_encap_batch.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz]
        lib.keaki_hip_group_decap_batch.argtypes = [vp, vp, vp, sz, vp, vp, sz]
        lib.keaki_hip_encrypt_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, sz]
        lib.keaki_hip_encrypt_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, sz]
        lib.keaki_hip_decrypt_batch.argtypes = [vp, vp, vp, vp, sz, vp, sz]
        lib.keaki_hip_decrypt_batch_dev.argtypes = [vp, vp, vp, sz, vp, sz]
        lib.keaki_hip_group_encrypt_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, sz]
        lib.keaki_hip_group_decrypt_batch.argtypes = [vp, vp, vp, vp, sz, vp, sz]
        lib.keaki_hip_group_fk_create.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, C.POINTER(vp)]
        lib.keaki_hip_group_fk_open.argtypes = [vp, vp, vp, vp]
        lib.keaki_hip_group_fk_free.argtypes = [vp, vp]
        lib.keaki_hip_group_fk_free.restype = None
        _LIB = lib
    return _LIB


def _np(a, width=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if width is not None:
        a = a.reshape(-1, width)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class SrsG1:
    def __init__(self, owner, handle, n):
        self.owner, self.handle, self.n = owner, handle, n

    def free(self):
        if self.handle:
            self.owner.lib.keaki_hip_srs_g1_free(self.owner.ctx, self.handle)
            self.handle = None


class SrsG2(SrsG1):
    def free(self):
        if self.handle:
            self.owner.lib.keaki_hip_srs_g2_free(self.owner.ctx, self.handle)
            self.handle = None


class FkShardHandle:
    """one rank's part of a sharded FK23 (keaki_hip_fk_shard). sizes = (buffer bytes, all-to-all bytes per peer of the 2d-point
    transforms, of the d-point transform, all-gather bytes per rank)"""

    def __init__(self, owner, handle, log2d, rank, world, sizes):
        self.owner, self.handle, self.log2d, self.rank, self.world, self.sizes = owner, handle, log2d, rank, world, sizes

    def free(self):
        if self.handle:
            self.owner.lib.keaki_hip_fk_shard_free(self.owner.ctx, self.handle)
            self.handle = None


class KeakiHip:
    """One context = one GPU (one process per GPU in multi-GPU runs)."""

    STREAM_LEGACY = 1          # KEAKI_HIP_STREAM_LEGACY: the device's legacy default stream (torch's default stream has handle 0 = this one)

    def __init__(self, device: int = 0, stream: int | None = None):
        """stream: None -> private non-blocking stream (order *_dev calls with synchronize()); 0 or STREAM_LEGACY -> the legacy default
        stream (what `torch.cuda.current_stream().cuda_stream` == 0 means); else a hipStream_t handle, e.g. torch.cuda.Stream().cuda_stream."""
        self.lib = load_library()
        ctx = C.c_void_p()
        if stream is not None and stream == 0:
            stream = self.STREAM_LEGACY
        st = self.lib.keaki_hip_ctx_create(device, C.c_void_p(stream) if stream is not None else None, C.byref(ctx))
        if st != 0:
            raise KeakiHipError(st, self.lib.keaki_hip_last_error(None).decode())
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.keaki_hip_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != 0:
            raise KeakiHipError(st, self.lib.keaki_hip_last_error(self.ctx).decode())

    def synchronize(self):
        self._ck(self.lib.keaki_hip_synchronize(self.ctx))

    def set_option(self, name: str, value: int):
        """tuning / A-B switch of this context (keaki_hip_ctx_set_option); the environment is only read when a context is created"""
        self._ck(self.lib.keaki_hip_ctx_set_option(self.ctx, name.encode(), int(value)))

    def debug_set_alloc_limit(self, nbytes: int):
        """test hook: single allocations of this context above nbytes fail with KEAKI_ERR_OOM (0 = no limit)"""
        self._ck(self.lib.keaki_hip_debug_set_alloc_limit(self.ctx, int(nbytes)))

    def memory(self) -> dict:
        out = (C.c_size_t * 4)()
        self._ck(self.lib.keaki_hip_ctx_memory(self.ctx, out))
        return {"tables": int(out[0]), "workspaces": int(out[1]), "gt_tables": int(out[2]), "total": int(out[3])}

    def trim(self):
        """release every workspace / encapsulate table of the context (rebuilt on demand)"""
        self._ck(self.lib.keaki_hip_ctx_trim(self.ctx))

    def version(self) -> str:
        return self.lib.keaki_hip_version().decode()

    def kzg_quotient(self, coeffs, point):
        """-> ((n - 1) x u64[4] quotient coefficients, p(point) u64[4])"""
        c = _np(coeffs, 4); n = c.shape[0]
        q = np.zeros((max(n - 1, 0), 4), np.uint64); val = np.zeros(4, np.uint64)
        self._ck(self.lib.keaki_hip_kzg_quotient(self.ctx, _ptr(c) if n else None, n, _ptr(_np(point)), _ptr(q) if n > 1 else None, _ptr(val)))
        return q, val

    def vec_commit(self, srs: "SrsG1", values, pad, log2d: int, omega_d_inv, inv_d, omega_2d, omega_2d_inv, inv_2d):
        """the body of vec::vec_commit behind the padding draw -> (commitment as normalised Jacobian u64[12], d affine proofs)"""
        v = _np(values, 4); d = 1 << log2d
        com = np.zeros(12, np.uint64); proofs = np.zeros((d, 8), np.uint64)
        self._ck(self.lib.keaki_hip_vec_commit(self.ctx, srs.handle, _ptr(v) if v.shape[0] else None, v.shape[0], _ptr(_np(pad)) if pad is not None else None, log2d,
                                               _ptr(_np(omega_d_inv)), _ptr(_np(inv_d)), _ptr(_np(omega_2d)), _ptr(_np(omega_2d_inv)), _ptr(_np(inv_2d)),
                                               _ptr(com), _ptr(proofs)))
        return com, proofs

    def set_timing(self, on: bool):
        self._ck(self.lib.keaki_hip_set_timing(self.ctx, 1 if on else 0))

    def last_msm_stats(self):
        return {"bucket_ms": float(self.lib.keaki_hip_last_msm_bucket_ms(self.ctx)),
                "total_ms": float(self.lib.keaki_hip_last_msm_total_ms(self.ctx)),
                "window_bits": int(self.lib.keaki_hip_last_msm_window_bits(self.ctx))}

    def last_fk_stats(self):
        out = (C.c_float * 3)()
        self._ck(self.lib.keaki_hip_last_fk_ms(self.ctx, out))
        return {"pointwise_ms": float(out[0]), "stages_ms": float(out[1]), "device_ms": float(out[2])}

    def selftest_field(self, blocks: int = 1024, iters: int = 64, seed: int = 1) -> int:
        """mismatch count of the hand-scheduled Fq streams vs the portable code (must be 0)"""
        bad = C.c_uint64(1)
        self._ck(self.lib.keaki_hip_selftest_field(self.ctx, blocks, iters, seed, C.byref(bad)))
        return int(bad.value)

    # ---- SRS
    def srs_g1_upload(self, points) -> SrsG1:
        pts = _np(points, 8); h = C.c_void_p()
        self._ck(self.lib.keaki_hip_srs_g1_upload(self.ctx, _ptr(pts), pts.shape[0], C.byref(h)))
        return SrsG1(self, h, pts.shape[0])

    def srs_g1_wrap_dev(self, dptr: int, n: int) -> SrsG1:
        h = C.c_void_p()
        self._ck(self.lib.keaki_hip_srs_g1_wrap_dev(self.ctx, C.c_void_p(dptr), n, C.byref(h)))
        return SrsG1(self, h, n)

    def srs_g1_slice(self, srs: SrsG1, offset: int, n: int) -> SrsG1:
        """non-owning view of points [offset, offset + n): one rank's chunk of a sharded MSM"""
        h = C.c_void_p()
        self._ck(self.lib.keaki_hip_srs_g1_slice(self.ctx, srs.handle, offset, n, C.byref(h)))
        return SrsG1(self, h, n)

    def srs_g2_upload(self, points) -> SrsG2:
        pts = _np(points, 16); h = C.c_void_p()
        self._ck(self.lib.keaki_hip_srs_g2_upload(self.ctx, _ptr(pts), pts.shape[0], C.byref(h)))
        return SrsG2(self, h, pts.shape[0])

    def srs_g2_wrap_dev(self, dptr: int, n: int) -> SrsG2:
        h = C.c_void_p()
        self._ck(self.lib.keaki_hip_srs_g2_wrap_dev(self.ctx, C.c_void_p(dptr), n, C.byref(h)))
        return SrsG2(self, h, n)

    def srs_g1_precompute(self, srs: "SrsG1") -> int:
        """one-time window-table build for a fixed SRS; returns the table size in bytes"""
        nbytes = C.c_size_t(0)
        self._ck(self.lib.keaki_hip_srs_g1_precompute(self.ctx, srs.handle, C.byref(nbytes)))
        return int(nbytes.value)

    def srs_g2_precompute(self, srs: "SrsG2") -> int:
        nbytes = C.c_size_t(0)
        self._ck(self.lib.keaki_hip_srs_g2_precompute(self.ctx, srs.handle, C.byref(nbytes)))
        return int(nbytes.value)

    # ---- MSM
    def msm_g1(self, srs: SrsG1, scalars) -> np.ndarray:
        sc = _np(scalars, 4); out = np.zeros(12, np.uint64)
        self._ck(self.lib.keaki_hip_msm_g1(self.ctx, srs.handle, _ptr(sc), sc.shape[0], _ptr(out)))
        return out

    def msm_g1_dev(self, srs: SrsG1, d_scalars: int, n: int, d_out: int):
        self._ck(self.lib.keaki_hip_msm_g1_dev(self.ctx, srs.handle, C.c_void_p(d_scalars), n, C.c_void_p(d_out)))

    def msm_g2(self, srs: SrsG2, scalars) -> np.ndarray:
        sc = _np(scalars, 4); out = np.zeros(24, np.uint64)
        self._ck(self.lib.keaki_hip_msm_g2(self.ctx, srs.handle, _ptr(sc), sc.shape[0], _ptr(out)))
        return out

    def msm_g2_dev(self, srs: SrsG2, d_scalars: int, n: int, d_out: int):
        self._ck(self.lib.keaki_hip_msm_g2_dev(self.ctx, srs.handle, C.c_void_p(d_scalars), n, C.c_void_p(d_out)))

    def g1_sum(self, points_jac) -> np.ndarray:
        p = _np(points_jac, 12); out = np.zeros(12, np.uint64)
        self._ck(self.lib.keaki_hip_g1_sum(self.ctx, _ptr(p), p.shape[0], _ptr(out)))
        return out

    def g1_sum_dev(self, d_points: int, k: int, d_out: int):
        self._ck(self.lib.keaki_hip_g1_sum_dev(self.ctx, C.c_void_p(d_points), k, C.c_void_p(d_out)))

    # ---- batched scalar multiplication
    def g1_mul_batch(self, points, scalars) -> np.ndarray:
        pts = _np(points); sc = _np(scalars, 4); n = sc.shape[0]
        stride = 0 if pts.size == 8 else 1
        out = np.zeros((n, 8), np.uint64)
        self._ck(self.lib.keaki_hip_g1_mul_batch(self.ctx, _ptr(pts), stride, _ptr(sc), n, _ptr(out)))
        return out

    def g2_mul_batch(self, points, scalars) -> np.ndarray:
        pts = _np(points); sc = _np(scalars, 4); n = sc.shape[0]
        stride = 0 if pts.size == 16 else 1
        out = np.zeros((n, 16), np.uint64)
        self._ck(self.lib.keaki_hip_g2_mul_batch(self.ctx, _ptr(pts), stride, _ptr(sc), n, _ptr(out)))
        return out

    def g1_mul_batch_dev(self, d_points: int, stride: int, d_scalars: int, n: int, d_out: int):
        self._ck(self.lib.keaki_hip_g1_mul_batch_dev(self.ctx, C.c_void_p(d_points), stride, C.c_void_p(d_scalars), n, C.c_void_p(d_out)))

    def g2_mul_batch_dev(self, d_points: int, stride: int, d_scalars: int, n: int, d_out: int):
        self._ck(self.lib.keaki_hip_g2_mul_batch_dev(self.ctx, C.c_void_p(d_points), stride, C.c_void_p(d_scalars), n, C.c_void_p(d_out)))

    # ---- pairing
    def pairing_batch(self, g1, g2) -> np.ndarray:
        p = _np(g1, 8); q = _np(g2); n = p.shape[0]
        stride = 0 if q.size == 16 else 1
        out = np.zeros((n, 384), np.uint8)
        self._ck(self.lib.keaki_hip_pairing_batch(self.ctx, _ptr(p), _ptr(q), stride, n, _ptr(out)))
        return out

    def pairing_batch_dev(self, d_g1: int, d_g2: int, stride: int, n: int, d_gt: int):
        self._ck(self.lib.keaki_hip_pairing_batch_dev(self.ctx, C.c_void_p(d_g1), C.c_void_p(d_g2), stride, n, C.c_void_p(d_gt)))

    def g2_prepare(self, g2) -> np.ndarray:
        """line table of a fixed Q: (96, 2 parities, 3 coefficients, 4 limbs) Montgomery"""
        q = _np(g2); out = np.zeros(96 * 2 * 3 * 4, np.uint64)
        self.lib.keaki_hip_g2_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self._ck(self.lib.keaki_hip_g2_prepare(self.ctx, _ptr(q), _ptr(out), out.nbytes))
        return out.reshape(96, 2, 3, 4)

    def miller_loop_batch(self, g1, g2) -> np.ndarray:
        p = _np(g1, 8); q = _np(g2, 16); n = p.shape[0]
        out = np.zeros((n, 48), np.uint64)
        self._ck(self.lib.keaki_hip_miller_loop_batch(self.ctx, _ptr(p), _ptr(q), n, _ptr(out)))
        return out

    def final_exp_batch(self, f_mont) -> np.ndarray:
        f = _np(f_mont, 48); n = f.shape[0]
        out = np.zeros((n, 384), np.uint8)
        self._ck(self.lib.keaki_hip_final_exp_batch(self.ctx, _ptr(f), n, _ptr(out)))
        return out

    def kzg_open(self, srs: "SrsG1", coeffs, point):
        """-> (proof as normalised Jacobian u64[12], p(point) u64[4])"""
        c = _np(coeffs, 4)
        out = np.zeros(12, np.uint64); val = np.zeros(4, np.uint64)
        self._ck(self.lib.keaki_hip_kzg_open(self.ctx, srs.handle, _ptr(c) if c.shape[0] else None, c.shape[0], _ptr(_np(point)), _ptr(out), _ptr(val)))
        return out, val

    def kzg_verify(self, com_aff, tau_g2_aff, point, value, proof_aff) -> bool:
        """src/kzg.rs:127-146 on the device (affine Montgomery points, Montgomery Fr)"""
        ok = C.c_int32(0)
        self._ck(self.lib.keaki_hip_kzg_verify(self.ctx, _ptr(_np(com_aff)), _ptr(_np(tau_g2_aff)), _ptr(_np(point)), _ptr(_np(value)),
                                               _ptr(_np(proof_aff)), C.byref(ok)))
        return bool(ok.value)

    def srs_g1_check(self, srs: "SrsG1"):
        """-> (number of off-curve points, index of the first or None)"""
        bad, first = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.lib.keaki_hip_srs_g1_check(self.ctx, srs.handle, C.byref(bad), C.byref(first)))
        return bad.value, (None if bad.value == 0 else first.value)

    def g2_check(self, points):
        pts = _np(points, 16)
        bad, first = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.lib.keaki_hip_g2_check(self.ctx, _ptr(pts), pts.shape[0], C.byref(bad), C.byref(first)))
        return bad.value, (None if bad.value == 0 else first.value)

    def fr_fft(self, data, log2n: int, omega, scale=None) -> np.ndarray:
        a = _np(data, 4).copy()
        assert a.shape[0] == 1 << log2n
        self._ck(self.lib.keaki_hip_fr_fft(self.ctx, _ptr(a), log2n, _ptr(_np(omega)), _ptr(_np(scale)) if scale is not None else None))
        return a

    def open_fk(self, srs: "SrsG1", log2d: int, hat_a, tw_2d, tw_2d_inv, tw_d) -> np.ndarray:
        d = 1 << log2d
        ha = _np(hat_a, 4); t1 = _np(tw_2d, 4); t2 = _np(tw_2d_inv, 4); t3 = _np(tw_d, 4) if d >= 2 else np.zeros((1, 4), np.uint64)
        out = np.zeros((d, 8), np.uint64)
        self._ck(self.lib.keaki_hip_open_fk(self.ctx, srs.handle, log2d, _ptr(ha), _ptr(t1), _ptr(t2), _ptr(t3), _ptr(out)))
        return out

    def open_fk_poly(self, srs: "SrsG1", log2d: int, coeffs, omega_2d, omega_2d_inv, inv_2d) -> np.ndarray:
        d = 1 << log2d
        p = _np(coeffs, 4)
        assert p.shape[0] == d
        out = np.zeros((d, 8), np.uint64)
        self._ck(self.lib.keaki_hip_open_fk_poly(self.ctx, srs.handle, log2d, _ptr(p), _ptr(_np(omega_2d)), _ptr(_np(omega_2d_inv)), _ptr(_np(inv_2d)), _ptr(out)))
        return out

    # ---- KEM composites
    # FK23 sharded over `world` ranks: the steps between the caller's exchanges (keaki_amd/dist.py::ShardedFk drives them)
    def fk_shard_create(self, srs: "SrsG1", log2d: int, rank: int, world: int, omega_2d, omega_2d_inv, inv_2d) -> "FkShardHandle":
        h = C.c_void_p()
        self._ck(self.lib.keaki_hip_fk_shard_create(self.ctx, srs.handle, log2d, rank, world, _ptr(_np(omega_2d)), _ptr(_np(omega_2d_inv)),
                                                    _ptr(_np(inv_2d)), C.byref(h)))
        sizes = (C.c_size_t * 4)()
        self._ck(self.lib.keaki_hip_fk_shard_sizes(h, sizes))
        return FkShardHandle(self, h, log2d, rank, world, tuple(int(x) for x in sizes))

    def fk_shard_setup(self, fk: "FkShardHandle", step: int, d_send: int, d_recv: int):
        self._ck(self.lib.keaki_hip_fk_shard_setup(self.ctx, fk.handle, step, C.c_void_p(d_send), C.c_void_p(d_recv)))

    def fk_shard_open(self, fk: "FkShardHandle", step: int, d_send: int, d_recv: int, coeffs=None, out=None):
        """steps 0-2 return None; step 3 returns the d affine proofs (u64[d, 8]) in natural order (written into `out` if given)"""
        if step == 3 and out is None:
            out = np.empty(((1 << fk.log2d), 8), dtype=np.uint64)
        p = _np(coeffs, 4) if coeffs is not None else None
        self._ck(self.lib.keaki_hip_fk_shard_open(self.ctx, fk.handle, step, _ptr(p), C.c_void_p(d_send), C.c_void_p(d_recv), _ptr(out)))
        return out

    def encap_batch(self, com, tau_g2, points, values, rs, msg_len: int = 32, want_gt: bool = True):
        """-> (ct, gt, key), or (ct, key) with want_gt=False (the GT bytes stay on the device: only the keys come back)"""
        com = _np(com); tau = _np(tau_g2); pts = _np(points, 4); vals = _np(values, 4); rs = _np(rs, 4)
        n = pts.shape[0]
        ct = np.zeros((n, 16), np.uint64); key = np.zeros((n, max(msg_len, 1)), np.uint8)
        gt = np.zeros((n, 384), np.uint8) if want_gt else None
        self._ck(self.lib.keaki_hip_encap_batch(self.ctx, _ptr(com), _ptr(tau), _ptr(pts), _ptr(vals), _ptr(rs), n,
                                                _ptr(ct), _ptr(gt) if want_gt else None, _ptr(key) if msg_len else None, msg_len))
        return (ct, gt, key[:, :msg_len]) if want_gt else (ct, key[:, :msg_len])

    def encap_prepare(self, tau_g2, batch_hint: int):
        """setup-time: the tables of encap_batch that depend on the setup only (keaki_hip_encap_prepare)"""
        self._ck(self.lib.keaki_hip_encap_prepare(self.ctx, _ptr(_np(tau_g2)), int(batch_hint)))

    def encap_batch_dev(self, d_com, d_tau, d_points, d_values, d_r, n, d_ct, d_gt, d_key, msg_len):
        v = lambda x: C.c_void_p(x) if x else None
        self._ck(self.lib.keaki_hip_encap_batch_dev(self.ctx, v(d_com), v(d_tau), v(d_points), v(d_values), v(d_r), n, v(d_ct), v(d_gt), v(d_key), msg_len))

    def decap_batch(self, proofs, cts, msg_len: int = 32):
        p = _np(proofs, 8); c = _np(cts, 16); n = p.shape[0]
        gt = np.zeros((n, 384), np.uint8); key = np.zeros((n, max(msg_len, 1)), np.uint8)
        self._ck(self.lib.keaki_hip_decap_batch(self.ctx, _ptr(p), _ptr(c), n, _ptr(gt), _ptr(key) if msg_len else None, msg_len))
        return gt, key[:, :msg_len]

    def encrypt_batch(self, com, tau_g2, points, values, rs, msgs, in_place: bool = False):
        """enc::encrypt over a batch (src/enc.rs:19-40 in the loop of src/vec.rs:63-66): msgs is an (n, msg_len) uint8 array.
        Returns (ct points (n, 16) u64, ciphertext bodies (n, msg_len) u8); the XOR runs on the device behind the KDF."""
        p, v, r = _np(points, 4), _np(values, 4), _np(rs, 4)
        m = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, msg_len = m.shape
        ct = np.zeros((n, 16), np.uint64)
        body = m if in_place else np.zeros((n, msg_len), np.uint8)      # in_place: the message array doubles as the output (the ABI allows it)
        self._ck(self.lib.keaki_hip_encrypt_batch(self.ctx, _ptr(_np(com)), _ptr(_np(tau_g2)), _ptr(p), _ptr(v), _ptr(r), _ptr(m), n, _ptr(ct), _ptr(body), msg_len))
        return ct, body

    def decrypt_batch(self, proofs, cts, bodies):
        """enc::decrypt over a batch (src/enc.rs:44-55 in the loop of src/vec.rs:75-78): returns the (n, msg_len) messages."""
        pr, c = _np(proofs, 8), _np(cts, 16)
        b = np.ascontiguousarray(bodies, dtype=np.uint8)
        n, msg_len = b.shape
        out = np.zeros((n, msg_len), np.uint8)
        self._ck(self.lib.keaki_hip_decrypt_batch(self.ctx, _ptr(pr), _ptr(c), _ptr(b), n, _ptr(out), msg_len))
        return out

    def decap_batch_dev(self, d_proofs, d_cts, n, d_gt, d_key, msg_len):
        v = lambda x: C.c_void_p(x) if x else None
        self._ck(self.lib.keaki_hip_decap_batch_dev(self.ctx, v(d_proofs), v(d_cts), n, v(d_gt), v(d_key), msg_len))


class GroupSrsG1:
    def __init__(self, owner, handle, n):
        self.owner, self.handle, self.n = owner, handle, n

    def free(self):
        if self.handle:
            self.owner.lib.keaki_hip_group_srs_g1_free(self.owner.g, self.handle)
            self.handle = None


class KeakiHipGroup:
    """keaki_hip_group_*: one context + one host thread per entry of `devices` inside the library (in-process multi-GPU; an ordinal may
    repeat). The SRS is split into contiguous chunks, one per member, each with its own window tables."""

    def __init__(self, devices):
        self.lib = load_library()
        devs = (C.c_int32 * len(devices))(*devices)
        g = C.c_void_p()
        st = self.lib.keaki_hip_group_create(devs, len(devices), C.byref(g))
        if st != 0:
            raise KeakiHipError(st, self.lib.keaki_hip_group_last_error(None).decode())
        self.g = g
        self.size = len(devices)

    def close(self):
        if getattr(self, "g", None):
            self.lib.keaki_hip_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != 0:
            raise KeakiHipError(st, self.lib.keaki_hip_group_last_error(self.g).decode())

    def peer_note(self) -> str:
        """Empty when every pair of distinct devices of the group has direct (peer) access; otherwise the pairs whose copies the runtime stages."""
        return self.lib.keaki_hip_group_peer_note(self.g).decode()

    def member_memory(self, i: int) -> dict:
        out = (C.c_size_t * 4)()
        st = self.lib.keaki_hip_ctx_memory(self.lib.keaki_hip_group_ctx(self.g, i), out)
        if st != 0:
            raise KeakiHipError(st, "ctx_memory")
        return {"tables": int(out[0]), "workspaces": int(out[1]), "gt_tables": int(out[2]), "total": int(out[3])}

    def srs_g1_upload(self, points, precompute: bool = True) -> GroupSrsG1:
        pts = _np(points, 8); h = C.c_void_p()
        self._ck(self.lib.keaki_hip_group_srs_g1_upload(self.g, _ptr(pts), pts.shape[0], 1 if precompute else 0, C.byref(h)))
        return GroupSrsG1(self, h, pts.shape[0])

    def msm_g1(self, srs: GroupSrsG1, scalars) -> np.ndarray:
        sc = _np(scalars, 4); out = np.zeros(12, np.uint64)
        self._ck(self.lib.keaki_hip_group_msm_g1(self.g, srs.handle, _ptr(sc) if sc.shape[0] else None, sc.shape[0], _ptr(out)))
        return out

    def kzg_open(self, srs: GroupSrsG1, coeffs, point):
        c = _np(coeffs, 4)
        out = np.zeros(12, np.uint64); val = np.zeros(4, np.uint64)
        self._ck(self.lib.keaki_hip_group_kzg_open(self.g, srs.handle, _ptr(c) if c.shape[0] else None, c.shape[0], _ptr(_np(point)), _ptr(out), _ptr(val)))
        return out, val

    def fk_create(self, points, log2d: int, omega_2d, omega_2d_inv, inv_2d):
        """FK23 over the members (keaki_hip_group_fk_*): handle for domain size 2^log2d over the first 2^log2d points"""
        pts = _np(points, 8)
        assert pts.shape[0] >= 1 << log2d
        h = C.c_void_p()
        self._ck(self.lib.keaki_hip_group_fk_create(self.g, _ptr(pts), log2d, _ptr(_np(omega_2d)), _ptr(_np(omega_2d_inv)), _ptr(_np(inv_2d)), C.byref(h)))
        return h

    def fk_open(self, fk, log2d: int, coeffs) -> np.ndarray:
        p = _np(coeffs, 4)
        assert p.shape[0] == 1 << log2d
        out = np.zeros((1 << log2d, 8), np.uint64)
        self._ck(self.lib.keaki_hip_group_fk_open(self.g, fk, _ptr(p), _ptr(out)))
        return out

    def fk_free(self, fk):
        self.lib.keaki_hip_group_fk_free(self.g, fk)

    def encap_batch(self, com, tau_g2, points, values, rs, msg_len: int = 32):
        com = _np(com); tau = _np(tau_g2); pts = _np(points, 4); vals = _np(values, 4); rs = _np(rs, 4)
        n = pts.shape[0]
        ct = np.zeros((n, 16), np.uint64); gt = np.zeros((n, 384), np.uint8); key = np.zeros((n, max(msg_len, 1)), np.uint8)
        self._ck(self.lib.keaki_hip_group_encap_batch(self.g, _ptr(com), _ptr(tau), _ptr(pts), _ptr(vals), _ptr(rs), n,
                                                      _ptr(ct), _ptr(gt), _ptr(key) if msg_len else None, msg_len))
        return ct, gt, key[:, :msg_len]

    def encrypt_batch(self, com, tau_g2, points, values, rs, msgs):
        p, v, r = _np(points, 4), _np(values, 4), _np(rs, 4)
        m = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, msg_len = m.shape
        ct = np.zeros((n, 16), np.uint64); body = np.zeros((n, msg_len), np.uint8)
        self._ck(self.lib.keaki_hip_group_encrypt_batch(self.g, _ptr(_np(com)), _ptr(_np(tau_g2)), _ptr(p), _ptr(v), _ptr(r), _ptr(m), n, _ptr(ct), _ptr(body), msg_len))
        return ct, body

    def decrypt_batch(self, proofs, cts, bodies):
        pr, c = _np(proofs, 8), _np(cts, 16)
        b = np.ascontiguousarray(bodies, dtype=np.uint8)
        n, msg_len = b.shape
        out = np.zeros((n, msg_len), np.uint8)
        self._ck(self.lib.keaki_hip_group_decrypt_batch(self.g, _ptr(pr), _ptr(c), _ptr(b), n, _ptr(out), msg_len))
        return out

    def decap_batch(self, proofs, cts, msg_len: int = 32):
        p = _np(proofs, 8); c = _np(cts, 16); n = p.shape[0]
        gt = np.zeros((n, 384), np.uint8); key = np.zeros((n, max(msg_len, 1)), np.uint8)
        self._ck(self.lib.keaki_hip_group_decap_batch(self.g, _ptr(p), _ptr(c), n, _ptr(gt), _ptr(key) if msg_len else None, msg_len))
        return gt, key[:, :msg_len]


def jac_to_affine_words(jac) -> np.ndarray:
    """normalised Jacobian u64[12] (x, y, 1 | 1, 1, 0) -> affine u64[8] (identity = zeros). Pure relabelling."""
    j = _np(jac).reshape(-1)
    w = j.size // 3
    out = np.zeros(2 * w, np.uint64)
    if np.any(j[2 * w:]):
        out[:] = j[:2 * w]
    return out
