#!/usr/bin/env python3
"""KZG `open` (src/kzg.rs:104-124) at size: keaki_hip_kzg_open = H2D of the coefficients + device quotient + MSM, against the same
call split into a host-side Horner quotient (numpy-free C++ is what the mirror used before) + keaki_hip_msm_g1.
    python bench_tools/bench_open.py --log2n 24
"""
import argparse, faulthandler, json, os, sys, time
faulthandler.enable()
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import random_fr_limbs, SEED  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--log2n", type=int, default=22); ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from keaki_amd.hip import KeakiHip
    dev = torch.device("cuda", 0)
    hip = KeakiHip(0)
    n = 1 << a.log2n
    P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    mont = lambda v: [((v << 256) % P_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    g1 = np.array(mont(1) + mont(2), np.uint64)
    ks = random_fr_limbs(n, SEED + 1)
    d_k = torch.from_numpy(ks.view(np.int64)).to(dev)
    d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
    hip.g1_mul_batch_dev(torch.from_numpy(g1.view(np.int64)).to(dev).data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr())
    torch.cuda.synchronize(dev)
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    hip.srs_g1_precompute(srs)
    coeffs = random_fr_limbs(n, SEED + 2)
    z = random_fr_limbs(1, SEED + 3)[0]
    torch.cuda.synchronize(dev)
    hip.kzg_open(srs, coeffs, z)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        proof, val = hip.kzg_open(srs, coeffs, z)
    t_open = (time.perf_counter() - t0) / a.steps
    hip.set_option("msm_pipe_chunks", 0)                 # one copy in front of the kernels (the path of rounds 1-4)
    hip.kzg_open(srs, coeffs, z)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        proof0, val0 = hip.kzg_open(srs, coeffs, z)
    t_open0 = (time.perf_counter() - t0) / a.steps
    hip.set_option("msm_pipe_chunks", -1)
    assert np.array_equal(proof0, proof) and np.array_equal(val0, val)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        hip.msm_g1(srs, coeffs[1:])
    t_msm = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"what": "KZG open, 2^%d coefficients, host pointers (PCIe included)" % a.log2n, "kzg_open_ms": t_open * 1e3, "kzg_open_one_copy_in_front_ms": t_open0 * 1e3,
                      "msm_only_same_size_ms": t_msm * 1e3, "device_quotient_overhead_ms": (t_open - t_msm) * 1e3}), flush=True)
    srs.free()
    hip.close()


if __name__ == "__main__":
    main()
