"""ctypes binding of libkeaki_host.so: the C++ mirror of keaki's public API (kzg / kem / enc / vec,
reference src/kzg.rs, src/kem.rs, src/enc.rs, src/vec.rs) for E = Bn254. Used by the parity tests
and the Laconic-OT harness so they read like the reference's own tests.

Values: Fr = numpy uint64[4] (Montgomery limbs), G1 = uint64[8], G2 = uint64[16] (affine, identity = zeros).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
PADDING_LEN = 1  # src/vec.rs:18


class KZGError(Exception):
    """KZGError::PolynomialTooLarge(degree, max_degree) -- src/kzg.rs:205-209"""

    def __init__(self, degree, max_degree):
        super().__init__(f"Can't commit to polynomial: polynomial has degree {degree} but max degree is {max_degree}")
        self.args_tuple = (degree, max_degree)

    def __eq__(self, other):
        return isinstance(other, KZGError) and self.args_tuple == other.args_tuple


class KeakiHostError(RuntimeError):
    pass


class SetupFileError(Exception):
    """SetupFileError -- src/kzg/ptau.rs:360-376 (+ OffCurve, Truncated: what the reference does not detect / panics on)"""
    KINDS = ["ElementSizeMismatch", "EmptySection", "FileError", "InvalidFileType", "InvalidNumberOfSections", "ParseError",
             "UnknownSection", "OffCurve", "Truncated"]

    def __init__(self, kind: int, a: int, b: int, text: str):
        super().__init__(text)
        self.kind, self.a, self.b = self.KINDS[kind], a, b

    def __eq__(self, other):
        return isinstance(other, SetupFileError) and (self.kind, self.a, self.b) == (other.kind, other.a, other.b)


def _lib():
    global _LIB
    if _LIB is None:
        from . import hip
        hip.load_library()  # the host mirror links against libkeaki_hip.so; fail loudly if it is missing
        path = os.path.join(_HERE, "libkeaki_host.so")
        if not os.path.exists(path):
            raise KeakiHostError(f"{path} not built (make -C keaki_amd/host)")
        lib = C.CDLL(path)
        lib.keaki_host_last_error.restype = C.c_char_p
        lib.keaki_host_rng_splitmix.restype = C.c_void_p
        lib.keaki_host_rng_splitmix.argtypes = [C.c_uint64]
        lib.keaki_host_rng_free.argtypes = [C.c_void_p]
        lib.keaki_host_domain.restype = C.c_size_t
        lib.keaki_host_domain.argtypes = [C.c_size_t, C.c_void_p]
        lib.keaki_host_setup_len.restype = C.c_size_t
        lib.keaki_host_setup_len.argtypes = [C.c_void_p]
        _LIB = lib
    return _LIB


def _u64(a, width=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, width) if width else a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ck(st, err=None):
    if st == 0:
        return
    if st == 1 and err is not None:
        raise KZGError(int(err[0]), int(err[1]))
    if st == 2 and err is not None:
        raise SetupFileError(int(err[0]), int(err[1]), int(err[2]), _lib().keaki_host_last_error().decode())
    raise KeakiHostError(f"status {st}: {_lib().keaki_host_last_error().decode()}")


class Rng:
    """Deterministic u64 stream standing in for rand::Rng (SplitMix64)."""

    def __init__(self, seed: int):
        self.h = C.c_void_p(_lib().keaki_host_rng_splitmix(seed))

    def fr_rand(self) -> np.ndarray:
        out = np.zeros(4, np.uint64)
        _lib().keaki_host_fr_rand(self.h, _p(out))
        return out

    def fr_rand_many(self, n: int) -> np.ndarray:
        """n consecutive Fr::rand draws, (n, 4)"""
        out = np.zeros((n, 4), np.uint64)
        _lib().keaki_host_fr_rand_many(self.h, C.c_size_t(n), _p(out))
        return out

    def __del__(self):
        try:
            _lib().keaki_host_rng_free(self.h)
        except Exception:
            pass


# ---- Fr helpers -------------------------------------------------------------------------------
def fr(v: int) -> np.ndarray:
    """Fr::from(i64)"""
    out = np.zeros(4, np.uint64)
    _lib().keaki_host_fr_from_i64(C.c_int64(v), _p(out))
    return out


def _binop(name, a, b):
    out = np.zeros(4, np.uint64)
    getattr(_lib(), name)(_p(_u64(a)), _p(_u64(b)), _p(out))
    return out


def fr_mul(a, b): return _binop("keaki_host_fr_mul", a, b)
def fr_add(a, b): return _binop("keaki_host_fr_add", a, b)
def fr_sub(a, b): return _binop("keaki_host_fr_sub", a, b)


def poly_evaluate(coeffs, x) -> np.ndarray:
    c = _u64(coeffs, 4); out = np.zeros(4, np.uint64)
    _lib().keaki_host_poly_eval(_p(c), C.c_size_t(c.shape[0]), _p(_u64(x)), _p(out))
    return out


def domain_elements(min_size: int) -> np.ndarray:
    """Radix2EvaluationDomain::new(min_size).elements()"""
    size = _lib().keaki_host_domain(min_size, None)
    out = np.zeros((size, 4), np.uint64)
    _lib().keaki_host_domain(min_size, _p(out))
    return out


def ifft(evals, domain_min) -> np.ndarray:
    e = _u64(evals, 4); size = _lib().keaki_host_domain(domain_min, None)
    out = np.zeros((size, 4), np.uint64)
    _lib().keaki_host_ifft(_p(e), C.c_size_t(e.shape[0]), C.c_size_t(domain_min), _p(out))
    return out


def fft(coeffs, domain_min) -> np.ndarray:
    c = _u64(coeffs, 4); size = _lib().keaki_host_domain(domain_min, None)
    out = np.zeros((size, 4), np.uint64)
    _lib().keaki_host_fft(_p(c), C.c_size_t(c.shape[0]), C.c_size_t(domain_min), _p(out))
    return out


# ---- kzg --------------------------------------------------------------------------------------
def ptau_parse(path: str) -> dict:
    """get_powers_from_file (src/kzg/ptau.rs:347-358) without a GPU: header, section table, and the limbs of sections 2 and 3."""
    lib = _lib()
    err = np.zeros(3, np.uint64)
    mod = np.zeros(64, np.uint8); mod_len = C.c_uint32(); power = C.c_uint32(); cer = C.c_uint32()
    secs = np.zeros((11, 3), np.uint64); flen = C.c_size_t(); n1 = C.c_size_t(); n2 = C.c_size_t()
    args = [os.fsencode(path), _p(mod), C.c_size_t(64), C.byref(mod_len), C.byref(power), C.byref(cer), _p(secs), C.byref(flen), C.byref(n1), C.byref(n2)]
    _ck(lib.keaki_host_ptau_parse(*args, None, None, _p(err)), err)
    g1 = np.zeros((n1.value, 8), np.uint64); g2 = np.zeros((n2.value, 16), np.uint64)
    _ck(lib.keaki_host_ptau_parse(*args, _p(g1), _p(g2), _p(err)), err)
    return {"file_len": flen.value, "field_modulus": bytes(mod[:mod_len.value]), "power": power.value, "ceremony_power": cer.value,
            "sections": [tuple(int(v) for v in r) for r in secs], "tau_g1": g1, "tau_g2": g2}


def ptau_section_info(header12: bytes, offset: int):
    """SectionInfo::new_from_data (src/kzg/ptau.rs:169-182) -> (id, size, position)"""
    out = np.zeros(3, np.uint64); err = np.zeros(3, np.uint64)
    buf = np.frombuffer(bytes(header12), np.uint8).copy()
    _ck(_lib().keaki_host_ptau_section_info(_p(buf), C.c_size_t(offset), _p(out), _p(err)), err)
    return tuple(int(v) for v in out)


def ptau_section_index(section_id: int) -> int:
    """SectionId::try_from + section_index (src/kzg/ptau.rs:54-90); -1 = UnknownSection"""
    return _lib().keaki_host_ptau_section_index(C.c_uint8(section_id))


class _CtxView:
    """a keaki_hip_ctx owned by someone else (a keaki::Device), with the option / debug / memory calls of the C ABI"""

    def __init__(self, ctx):
        from . import hip as H
        self.lib, self.ctx, self._H = H.load_library(), ctx, H

    def _ck(self, st):
        if st != 0:
            raise self._H.KeakiHipError(st, self.lib.keaki_hip_last_error(self.ctx).decode())

    def set_option(self, name: str, value: int):
        self._ck(self.lib.keaki_hip_ctx_set_option(self.ctx, name.encode(), int(value)))

    def debug_set_alloc_limit(self, nbytes: int):
        self._ck(self.lib.keaki_hip_debug_set_alloc_limit(self.ctx, int(nbytes)))

    def memory(self) -> dict:
        out = (C.c_size_t * 4)()
        self._ck(self.lib.keaki_hip_ctx_memory(self.ctx, out))
        return {"tables": int(out[0]), "workspaces": int(out[1]), "gt_tables": int(out[2]), "total": int(out[3])}


class Device:
    """keaki::Device: one GPU context shared by the setups created on it -- or, from a LIST of ordinals, several GPUs of this process
    behind one object (one context + one host thread per entry inside libkeaki_hip.so; an ordinal may repeat): commit / open then
    spread the MSM over the members by SRS range, vec_encrypt / vec_decrypt split their items."""

    def __init__(self, ordinals=0):
        lib = _lib()
        lib.keaki_host_device_ctx.restype = C.c_void_p
        lib.keaki_host_device_ctx.argtypes = [C.c_void_p, C.c_size_t]
        lib.keaki_host_device_members.restype = C.c_size_t
        lib.keaki_host_device_members.argtypes = [C.c_void_p]
        lib.keaki_host_device_free.argtypes = [C.c_void_p]
        ords = [int(ordinals)] if isinstance(ordinals, (int, np.integer)) else [int(o) for o in ordinals]
        arr = (C.c_int * len(ords))(*ords)
        h = C.c_void_p()
        _ck(lib.keaki_host_device_new(arr, C.c_size_t(len(ords)), C.byref(h)))
        self.h = h
        self.members = int(lib.keaki_host_device_members(h))

    def hip(self, member: int = 0):
        """the member's context as a non-owning view with set_option / debug_set_alloc_limit / memory (keaki_amd.hip.KeakiHip's methods)"""
        return _CtxView(C.c_void_p(_lib().keaki_host_device_ctx(self.h, member)))

    def close(self):
        if self.h:
            _lib().keaki_host_device_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KZGSetup:
    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def setup(secret, max_d: int, device=0) -> "KZGSetup":
        """device: a GPU ordinal (a context of its own is created) or a Device (shared context, or a group of GPUs)"""
        h = C.c_void_p()
        if isinstance(device, Device):
            _ck(_lib().keaki_host_setup_on(device.h, _p(_u64(secret)), C.c_size_t(max_d), C.byref(h)))
        else:
            _ck(_lib().keaki_host_setup(device, _p(_u64(secret)), C.c_size_t(max_d), C.byref(h)))
        return KZGSetup(h)

    @staticmethod
    def from_powers(g1_aff, tau_g2, device=0) -> "KZGSetup":
        pts = _u64(g1_aff, 8); h = C.c_void_p()
        if isinstance(device, Device):
            _ck(_lib().keaki_host_setup_from_powers_on(device.h, _p(pts), C.c_size_t(pts.shape[0]), _p(_u64(tau_g2)), C.byref(h)))
        else:
            _ck(_lib().keaki_host_setup_from_powers(device, _p(pts), C.c_size_t(pts.shape[0]), _p(_u64(tau_g2)), C.byref(h)))
        return KZGSetup(h)

    @staticmethod
    def new_from_file(path: str, device: int = 0) -> "KZGSetup":
        """KZGSetup::new_from_file (src/kzg.rs:33-52): sections 2 / 3 of a snarkjs .ptau uploaded verbatim, curve-checked on the GPU"""
        h = C.c_void_p(); err = np.zeros(3, np.uint64)
        _ck(_lib().keaki_host_setup_from_file(device, os.fsencode(path), C.byref(h), _p(err)), err)
        return KZGSetup(h)

    def g1_pow(self) -> np.ndarray:
        n = _lib().keaki_host_setup_len(self.h)
        out = np.zeros((n, 8), np.uint64)
        _lib().keaki_host_setup_g1_all(self.h, _p(out))
        return out

    g1_aff = g1_pow

    def tau_g2(self) -> np.ndarray:
        out = np.zeros(16, np.uint64)
        _lib().keaki_host_setup_tau_g2(self.h, _p(out))
        return out

    def has_window_tables(self) -> bool:
        """False when the optional SRS window tables did not fit in HBM (commit / open then take the generic MSM path)"""
        return bool(_lib().keaki_host_setup_has_tables(self.h))

    def close(self):
        if self.h:
            _lib().keaki_host_setup_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def commit(setup: KZGSetup, p) -> np.ndarray:
    c = _u64(p, 4); out = np.zeros(8, np.uint64); err = np.zeros(2, np.uint64)
    _ck(_lib().keaki_host_commit(setup.h, _p(c), C.c_size_t(c.shape[0]), _p(out), _p(err)), err)
    return out


def open(setup: KZGSetup, p, point) -> np.ndarray:  # noqa: A001 (mirrors kzg::open)
    c = _u64(p, 4); out = np.zeros(8, np.uint64); err = np.zeros(2, np.uint64)
    _ck(_lib().keaki_host_open(setup.h, _p(c), C.c_size_t(c.shape[0]), _p(_u64(point)), _p(out), _p(err)), err)
    return out


def verify(setup: KZGSetup, commitment, point, value, proof) -> bool:
    ok = C.c_int(0)
    _ck(_lib().keaki_host_verify(setup.h, _p(_u64(commitment)), _p(_u64(point)), _p(_u64(value)), _p(_u64(proof)), C.byref(ok)))
    return bool(ok.value)


def precompute_open_fk(setup: "KZGSetup", domain_size: int) -> None:
    """setup-time: the SRS-only transform of FK23 (hat_s) for this domain size"""
    _ck(_lib().keaki_host_precompute_open_fk(setup.h, C.c_size_t(domain_size)))


def open_fk(setup: KZGSetup, p, domain_size: int) -> np.ndarray:
    c = _u64(p, 4); size = _lib().keaki_host_domain(domain_size, None)
    out = np.zeros((size, 8), np.uint64); err = np.zeros(2, np.uint64)
    _ck(_lib().keaki_host_open_fk(setup.h, _p(c), C.c_size_t(c.shape[0]), C.c_size_t(domain_size), _p(out), _p(err)), err)
    return out


# ---- kem / enc ---------------------------------------------------------------------------------
def encapsulate(rng: Rng, setup: KZGSetup, commitment, point, value, msg_len: int):
    ct = np.zeros(16, np.uint64); key = np.zeros(max(msg_len, 1), np.uint8)
    _ck(_lib().keaki_host_encapsulate(rng.h, setup.h, _p(_u64(commitment)), _p(_u64(point)), _p(_u64(value)), C.c_size_t(msg_len), _p(ct), _p(key)))
    return ct, key[:msg_len].tobytes()


def kem_prepare(setup: KZGSetup, batch_hint: int) -> None:
    """setup-time: the tables of encapsulation that depend on the setup only, for batches of `batch_hint` items (keaki::kem::prepare)"""
    _ck(_lib().keaki_host_kem_prepare(setup.h, C.c_size_t(batch_hint)))


def decapsulate(setup: KZGSetup, proof, ciphertext, msg_len: int) -> bytes:
    key = np.zeros(max(msg_len, 1), np.uint8)
    _ck(_lib().keaki_host_decapsulate(setup.h, _p(_u64(proof)), _p(_u64(ciphertext)), C.c_size_t(msg_len), _p(key)))
    return key[:msg_len].tobytes()


def encrypt(rng: Rng, setup: KZGSetup, com, point, value, msg: bytes):
    m = np.frombuffer(msg or b"\0", np.uint8).copy(); ct = np.zeros(16, np.uint64); out = np.zeros(max(len(msg), 1), np.uint8)
    _ck(_lib().keaki_host_encrypt(rng.h, setup.h, _p(_u64(com)), _p(_u64(point)), _p(_u64(value)), _p(m), C.c_size_t(len(msg)), _p(ct), _p(out)))
    return ct, out[:len(msg)].tobytes()


def decrypt(setup: KZGSetup, proof, ct) -> bytes:
    g2, body = ct
    b = np.frombuffer(body or b"\0", np.uint8).copy(); out = np.zeros(max(len(body), 1), np.uint8)
    _ck(_lib().keaki_host_decrypt(setup.h, _p(_u64(proof)), _p(_u64(g2)), _p(b), C.c_size_t(len(body)), _p(out)))
    return out[:len(body)].tobytes()


# ---- vec ---------------------------------------------------------------------------------------
def vec_commit(rng: Rng, setup: KZGSetup, v):
    v = _u64(v, 4); n = v.shape[0]
    size = _lib().keaki_host_domain(n + PADDING_LEN, None)
    com = np.zeros(8, np.uint64); proofs = np.zeros((size, 8), np.uint64)
    _ck(_lib().keaki_host_vec_commit(rng.h, setup.h, _p(v), C.c_size_t(n), _p(com), _p(proofs)))
    return com, proofs


def vec_encrypt(rng: Rng, setup: KZGSetup, com, points, values, messages):
    n = len(messages); ml = len(messages[0]) if n else 0
    assert all(len(m) == ml for m in messages)
    pts = _u64(points, 4)[:n]; vals = _u64(values, 4)[:n]
    msgs = np.frombuffer(b"".join(messages) or b"\0", np.uint8).copy()
    g2 = np.zeros((n, 16), np.uint64); body = np.zeros(max(n * ml, 1), np.uint8)
    _ck(_lib().keaki_host_vec_encrypt(rng.h, setup.h, _p(_u64(com)), _p(np.ascontiguousarray(pts)), _p(np.ascontiguousarray(vals)), _p(msgs),
                                     C.c_size_t(n), C.c_size_t(ml), _p(g2), _p(body)))
    return [(g2[i].copy(), body[i * ml:(i + 1) * ml].tobytes()) for i in range(n)]


def vec_encrypt_arrays(rng: Rng, setup: KZGSetup, com, points, values, messages: np.ndarray):
    """vec_encrypt on contiguous arrays (n equal-length messages as an (n, len) uint8 array) -> (ct G2 points (n, 16) u64,
    ct bodies (n, len) u8): the same call without a Python object per item."""
    msgs = np.ascontiguousarray(messages, dtype=np.uint8)
    n, ml = msgs.shape
    pts = np.ascontiguousarray(_u64(points, 4)[:n]); vals = np.ascontiguousarray(_u64(values, 4)[:n])
    g2 = np.zeros((n, 16), np.uint64); body = np.zeros((n, max(ml, 1)), np.uint8)
    _ck(_lib().keaki_host_vec_encrypt(rng.h, setup.h, _p(_u64(com)), _p(pts), _p(vals), _p(msgs), C.c_size_t(n), C.c_size_t(ml), _p(g2), _p(body)))
    return g2, body[:, :ml]


def vec_decrypt_arrays(setup: KZGSetup, proofs, ct_g2: np.ndarray, ct_body: np.ndarray) -> np.ndarray:
    body = np.ascontiguousarray(ct_body, dtype=np.uint8)
    n, ml = body.shape
    pr = np.ascontiguousarray(_u64(proofs, 8)[:n]); g2 = np.ascontiguousarray(_u64(ct_g2, 16)[:n])
    out = np.zeros((n, max(ml, 1)), np.uint8)
    _ck(_lib().keaki_host_vec_decrypt(setup.h, _p(pr), _p(g2), _p(body), C.c_size_t(n), C.c_size_t(ml), _p(out)))
    return out[:, :ml]


def vec_decrypt(setup: KZGSetup, proofs, cts):
    n = len(cts); ml = len(cts[0][1]) if n else 0
    pr = np.ascontiguousarray(_u64(proofs, 8)[:n])
    g2 = np.ascontiguousarray(np.stack([_u64(c[0]) for c in cts])) if n else np.zeros((0, 16), np.uint64)
    body = np.frombuffer(b"".join(c[1] for c in cts) or b"\0", np.uint8).copy()
    out = np.zeros(max(n * ml, 1), np.uint8)
    _ck(_lib().keaki_host_vec_decrypt(setup.h, _p(pr), _p(g2), _p(body), C.c_size_t(n), C.c_size_t(ml), _p(out)))
    return [out[i * ml:(i + 1) * ml].tobytes() for i in range(n)]


# ---- keaki::dist (host/keaki.hpp): one process per GPU; the exchange of the 96-byte partials is the caller's (keaki_amd/dist.py) --------
def prepare_shard(setup: KZGSetup, rank: int, world: int) -> None:
    """setup-time: window tables of this rank's SRS chunk (commit_partial would build them on first use)"""
    _ck(_lib().keaki_host_prepare_shard(setup.h, C.c_size_t(rank), C.c_size_t(world)))


def commit_partial(setup: KZGSetup, p, rank: int, world: int) -> np.ndarray:
    """this rank's share of kzg::commit: the MSM over its contiguous range of the SRS -> normalised Jacobian u64[12]"""
    c = _u64(p, 4); out = np.zeros(12, np.uint64); err = np.zeros(2, np.uint64)
    _ck(_lib().keaki_host_commit_partial(setup.h, _p(c), C.c_size_t(c.shape[0]), C.c_size_t(rank), C.c_size_t(world), _p(out), _p(err)), err)
    return out


def vec_commit_partial(rng: Rng, setup: KZGSetup, v, rank: int, world: int):
    """vec_commit with the final commit left as this rank's partial -> (partial u64[12], proofs); see keaki::dist::vec_commit_partial"""
    v = _u64(v, 4); n = v.shape[0]
    size = _lib().keaki_host_domain(n + PADDING_LEN, None)
    part = np.zeros(12, np.uint64); proofs = np.zeros((size, 8), np.uint64)
    _ck(_lib().keaki_host_vec_commit_partial(rng.h, setup.h, _p(v), C.c_size_t(n), C.c_size_t(rank), C.c_size_t(world), _p(part), _p(proofs)))
    return part, proofs


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)    # (user, d_send, d_recv, bytes per peer / per rank) -> 0 = ok


class ShardedOpenFk:
    """keaki::dist::ShardedOpenFk: kzg::open_fk with the group FFTs sharded over the ranks. `all_to_all(d_send, d_recv, bytes_per_peer)` and
    `all_gather(d_send, d_recv, bytes_per_rank)` are the caller's collectives over device memory (they must have completed on return);
    d_send / d_recv: device pointers to buffer_bytes each (keaki_amd/dist.py::ShardedFk provides all of this over torch.distributed)."""

    def __init__(self, setup: KZGSetup, domain_size: int, rank: int, world: int, all_to_all, all_gather):
        lib = _lib()
        lib.keaki_host_fk_shard_buffer_bytes.restype = C.c_size_t
        lib.keaki_host_fk_shard_buffer_bytes.argtypes = [C.c_void_p]
        lib.keaki_host_fk_shard_free.argtypes = [C.c_void_p]
        h = C.c_void_p()
        _ck(lib.keaki_host_fk_shard_new(setup.h, C.c_size_t(domain_size), C.c_size_t(rank), C.c_size_t(world), C.byref(h)))
        self.h, self.setup, self.domain_size, self.rank, self.world = h, setup, domain_size, rank, world
        self.buffer_bytes = int(lib.keaki_host_fk_shard_buffer_bytes(h))
        self.callback_error = None

        def wrap(fn):
            def cb(user, s, r, n):
                try:                                   # an exception must not cross the C frames: report it as a status
                    fn(s, r, n)
                    return 0
                except BaseException as e:             # noqa: BLE001
                    self.callback_error = e
                    return 1
            return EXCHANGE_FN(cb)
        self._a2a, self._gather = wrap(all_to_all), wrap(all_gather)          # kept alive with the object

    @staticmethod
    def can_shard(setup: KZGSetup, domain_size: int, rank: int, world: int) -> bool:
        return bool(_lib().keaki_host_fk_shard_can(setup.h, C.c_size_t(domain_size), C.c_size_t(rank), C.c_size_t(world)))

    def _ck(self, st):
        if st != 0 and self.callback_error is not None:
            e, self.callback_error = self.callback_error, None
            raise KeakiHostError("the exchange callback failed: %r" % (e,)) from e
        _ck(st)

    def prepare(self, d_send: int, d_recv: int) -> None:
        self._ck(_lib().keaki_host_fk_shard_prepare(self.h, C.c_void_p(d_send), C.c_void_p(d_recv), self._a2a, self._gather, None))

    def open(self, p, d_send: int, d_recv: int) -> np.ndarray:
        c = _u64(p, 4); out = np.zeros((self.domain_size, 8), np.uint64)
        self._ck(_lib().keaki_host_fk_shard_open(self.h, _p(c), C.c_size_t(c.shape[0]), C.c_void_p(d_send), C.c_void_p(d_recv), self._a2a, self._gather,
                                           None, _p(out)))
        return out

    def vec_commit_partial(self, rng: Rng, v, d_send: int, d_recv: int):
        """keaki::dist::vec_commit_partial_fk -> (partial u64[12], proofs)"""
        v = _u64(v, 4); n = v.shape[0]
        part = np.zeros(12, np.uint64); proofs = np.zeros((self.domain_size, 8), np.uint64)
        self._ck(_lib().keaki_host_vec_commit_partial_fk(rng.h, self.setup.h, _p(v), C.c_size_t(n), C.c_size_t(self.rank), C.c_size_t(self.world), self.h,
                                                   C.c_void_p(d_send), C.c_void_p(d_recv), self._a2a, self._gather, None, _p(part), _p(proofs)))
        return part, proofs

    def close(self):
        if self.h:
            _lib().keaki_host_fk_shard_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def commit_combine(setup: KZGSetup, partials) -> np.ndarray:
    pr = _u64(partials, 12); out = np.zeros(8, np.uint64)
    _ck(_lib().keaki_host_commit_combine(setup.h, _p(pr), C.c_size_t(pr.shape[0]), _p(out)))
    return out


def vec_encrypt_arrays_shard(rng: Rng, setup: KZGSetup, com, points, values, messages: np.ndarray, rank: int, world: int):
    """vec_encrypt_arrays for the items of rank `rank` only: consumes the rng exactly as the full call does (one r per item of the WHOLE
    vector, in index order), returns the ciphertexts of this rank's contiguous item range"""
    from .dist import chunk_bounds
    msgs = np.ascontiguousarray(messages, dtype=np.uint8)
    n, ml = msgs.shape
    lo, hi = chunk_bounds(n, world, rank)
    pts = np.ascontiguousarray(_u64(points, 4)[:n]); vals = np.ascontiguousarray(_u64(values, 4)[:n])
    g2 = np.zeros((hi - lo, 16), np.uint64); body = np.zeros((hi - lo, max(ml, 1)), np.uint8)
    _ck(_lib().keaki_host_vec_encrypt_shard(rng.h, setup.h, _p(_u64(com)), _p(pts), _p(vals), _p(msgs), C.c_size_t(n), C.c_size_t(ml),
                                           C.c_size_t(rank), C.c_size_t(world), _p(g2), _p(body)))
    return g2, body[:, :ml]
