"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/libbn254_oracle.so (bn254_ref.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All arrays are numpy uint64 in the C-ABI layouts of include/keaki_hip.h (Montgomery limbs).
"""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbn254_oracle.so")
_lib = None

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "libbn254_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ref_window_size.restype = C.c_int
        _lib.ref_window_size.argtypes = [C.c_size_t]
        _lib.ref_g1_on_curve.restype = C.c_int
        _lib.ref_g2_on_curve.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


# ---- conversions between Python ints and limb arrays ------------------------------------------
def ints_to_limbs(vals) -> np.ndarray:
    """list of ints (< 2^256) -> (n,4) uint64, little-endian limbs"""
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def limbs_to_ints(a) -> list:
    a = _u64(a).reshape(-1, 4)
    b = a.tobytes()
    return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(a.shape[0])]


def fr_to_mont(canon: np.ndarray) -> np.ndarray:
    canon = _u64(canon).reshape(-1, 4); out = np.empty_like(canon)
    lib().ref_fr_to_mont(_p(canon), _p(out), C.c_size_t(canon.shape[0])); return out


def fr_from_mont(mont: np.ndarray) -> np.ndarray:
    mont = _u64(mont).reshape(-1, 4); out = np.empty_like(mont)
    lib().ref_fr_from_mont(_p(mont), _p(out), C.c_size_t(mont.shape[0])); return out


def fq_to_mont(canon: np.ndarray) -> np.ndarray:
    canon = _u64(canon).reshape(-1, 4); out = np.empty_like(canon)
    lib().ref_fq_to_mont(_p(canon), _p(out), C.c_size_t(canon.shape[0])); return out


def fq_from_mont(mont: np.ndarray) -> np.ndarray:
    mont = _u64(mont).reshape(-1, 4); out = np.empty_like(mont)
    lib().ref_fq_from_mont(_p(mont), _p(out), C.c_size_t(mont.shape[0])); return out


def g1_from_ints(pts) -> np.ndarray:
    """list of affine (x,y) int tuples or None -> (n,8) Montgomery limbs (identity = zeros)"""
    flat = []
    for pt in pts:
        flat += [0, 0] if pt is None else [pt[0], pt[1]]
    return fq_to_mont(ints_to_limbs(flat)).reshape(-1, 8)


def g1_to_ints(a) -> list:
    a = _u64(a).reshape(-1, 8)
    v = limbs_to_ints(fq_from_mont(a.reshape(-1, 4)))
    return [None if (v[2 * i] == 0 and v[2 * i + 1] == 0) else (v[2 * i], v[2 * i + 1]) for i in range(a.shape[0])]


def g2_from_ints(pts) -> np.ndarray:
    flat = []
    for pt in pts:
        flat += [0, 0, 0, 0] if pt is None else [pt[0][0], pt[0][1], pt[1][0], pt[1][1]]
    return fq_to_mont(ints_to_limbs(flat)).reshape(-1, 16)


def g2_to_ints(a) -> list:
    a = _u64(a).reshape(-1, 16)
    v = limbs_to_ints(fq_from_mont(a.reshape(-1, 4)))
    out = []
    for i in range(a.shape[0]):
        w = v[4 * i:4 * i + 4]
        out.append(None if not any(w) else ((w[0], w[1]), (w[2], w[3])))
    return out


def generators():
    g1 = np.zeros(8, np.uint64); g2 = np.zeros(16, np.uint64)
    lib().ref_generators(_p(g1), _p(g2)); return g1, g2


# ---- the restated algorithms ------------------------------------------------------------------
def window_size(n: int) -> int:
    return lib().ref_window_size(n)


def msm_g1(points, scalars, threads: int = 1) -> np.ndarray:
    points = _u64(points).reshape(-1, 8); scalars = _u64(scalars).reshape(-1, 4)
    n = min(points.shape[0], scalars.shape[0]); out = np.zeros(8, np.uint64)
    lib().ref_msm_g1(_p(points), _p(scalars), C.c_size_t(n), _p(out), C.c_int(threads)); return out


def msm_g2(points, scalars, threads: int = 1) -> np.ndarray:
    points = _u64(points).reshape(-1, 16); scalars = _u64(scalars).reshape(-1, 4)
    n = min(points.shape[0], scalars.shape[0]); out = np.zeros(16, np.uint64)
    lib().ref_msm_g2(_p(points), _p(scalars), C.c_size_t(n), _p(out), C.c_int(threads)); return out


def g1_mul_batch(points, scalars, threads: int = 1) -> np.ndarray:
    """points (n,8) or (8,) broadcast; scalars (n,4) Montgomery"""
    points = _u64(points); scalars = _u64(scalars).reshape(-1, 4); n = scalars.shape[0]
    stride = 0 if points.size == 8 else 8
    out = np.zeros((n, 8), np.uint64)
    lib().ref_g1_mul_batch(_p(points), C.c_int(stride), _p(scalars), C.c_size_t(n), _p(out), C.c_int(threads)); return out


def g2_mul_batch(points, scalars, threads: int = 1) -> np.ndarray:
    points = _u64(points); scalars = _u64(scalars).reshape(-1, 4); n = scalars.shape[0]
    stride = 0 if points.size == 16 else 16
    out = np.zeros((n, 16), np.uint64)
    lib().ref_g2_mul_batch(_p(points), C.c_int(stride), _p(scalars), C.c_size_t(n), _p(out), C.c_int(threads)); return out


def g1_sum(points) -> np.ndarray:
    points = _u64(points).reshape(-1, 8); out = np.zeros(8, np.uint64)
    lib().ref_g1_sum(_p(points), C.c_size_t(points.shape[0]), _p(out)); return out


def g1_on_curve(pt) -> bool:
    return bool(lib().ref_g1_on_curve(_p(_u64(pt))))


def g2_on_curve(pt) -> bool:
    return bool(lib().ref_g2_on_curve(_p(_u64(pt))))


def pairing_batch(g1, g2, threads: int = 1) -> np.ndarray:
    """g1 (n,8); g2 (n,16) or (16,) broadcast -> (n,384) uint8 GT bytes"""
    g1 = _u64(g1).reshape(-1, 8); g2 = _u64(g2); n = g1.shape[0]
    stride = 0 if g2.size == 16 else 16
    out = np.zeros((n, 384), np.uint8)
    lib().ref_pairing_batch(_p(g1), _p(g2), C.c_int(stride), C.c_size_t(n), _p(out), C.c_int(threads)); return out


def miller_loop_raw(g1, g2) -> np.ndarray:
    out = np.zeros(48, np.uint64); lib().ref_miller_loop_raw(_p(_u64(g1)), _p(_u64(g2)), _p(out)); return out


def final_exp_raw(f48) -> np.ndarray:
    out = np.zeros(48, np.uint64); lib().ref_final_exp_raw(_p(_u64(f48)), _p(out)); return out


def blake3_xof(data: bytes, out_len: int) -> bytes:
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    out = (C.c_uint8 * max(1, out_len))()
    lib().ref_blake3_xof(buf, C.c_size_t(len(data)), out, C.c_size_t(out_len)); return bytes(out)[:out_len]


def encap_batch(com, tau_g2, points, values, rs, msg_len: int = 32, threads: int = 1):
    """-> (ct (n,16) affine G2, gt (n,384) u8, key (n,msg_len) u8)"""
    points = _u64(points).reshape(-1, 4); values = _u64(values).reshape(-1, 4); rs = _u64(rs).reshape(-1, 4)
    n = points.shape[0]
    ct = np.zeros((n, 16), np.uint64); gt = np.zeros((n, 384), np.uint8); key = np.zeros((n, max(1, msg_len)), np.uint8)
    lib().ref_encap_batch(_p(_u64(com)), _p(_u64(tau_g2)), _p(points), _p(values), _p(rs), C.c_size_t(n),
                          _p(ct), _p(gt), _p(key), C.c_size_t(msg_len), C.c_int(threads))
    return ct, gt, key[:, :msg_len]


def decap_batch(proofs, cts, msg_len: int = 32, threads: int = 1):
    proofs = _u64(proofs).reshape(-1, 8); cts = _u64(cts).reshape(-1, 16); n = proofs.shape[0]
    gt = np.zeros((n, 384), np.uint8); key = np.zeros((n, max(1, msg_len)), np.uint8)
    lib().ref_decap_batch(_p(proofs), _p(cts), C.c_size_t(n), _p(gt), _p(key), C.c_size_t(msg_len), C.c_int(threads))
    return gt, key[:, :msg_len]


def fr_dot(a, b) -> np.ndarray:
    a = _u64(a).reshape(-1, 4); b = _u64(b).reshape(-1, 4); out = np.zeros(4, np.uint64)
    lib().ref_fr_dot(_p(a), _p(b), C.c_size_t(a.shape[0]), _p(out)); return out


def make_digits(scalar_canonical_int: int, w: int) -> list:
    s = ints_to_limbs([scalar_canonical_int]); nd = (254 + w - 1) // w
    d = np.zeros(nd, np.int32)
    lib().ref_make_digits(_p(s), C.c_int(w), _p(d)); return d.tolist()


def fr_fft(data, omega, scale=None) -> np.ndarray:
    """ark-poly Radix2EvaluationDomain::fft over the domain generated by `omega` (pass omega^-1 and scale = n^-1 for ifft); Montgomery"""
    a = _u64(data).reshape(-1, 4).copy(); n = a.shape[0]
    log2n = n.bit_length() - 1
    assert n == 1 << log2n
    lib().ref_fr_fft(_p(a), C.c_int(log2n), _p(_u64(omega)))
    if scale is not None:
        lib().ref_fr_scale(_p(a), C.c_size_t(n), _p(_u64(scale)))
    return a


def fr_quotient(coeffs, point):
    """src/kzg.rs:109-120 -> (quotient (n-1, 4), p(point) (4,)); Montgomery"""
    c = _u64(coeffs).reshape(-1, 4); n = c.shape[0]
    q = np.zeros((max(n - 1, 0), 4), np.uint64); v = np.zeros(4, np.uint64)
    lib().ref_fr_quotient(_p(c), C.c_size_t(n), _p(_u64(point)), _p(q), _p(v))
    return q, v
