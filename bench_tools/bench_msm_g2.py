#!/usr/bin/env python3
"""G2 MSM timing (scope row K4): 2^k points generated on the device, scalars uniform. python bench_tools/bench_msm_g2.py 18"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import random_fr_limbs, SEED
import torch
torch.cuda.init()      # before the library's context
from keaki_amd.hip import KeakiHip
hip = KeakiHip(0)
P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
mont = lambda v: [((v << 256) % P_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634), (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))
g2 = np.array(mont(G2[0][0]) + mont(G2[0][1]) + mont(G2[1][0]) + mont(G2[1][1]), np.uint64)
for log2n in [int(x) for x in sys.argv[1:]] or [16, 18]:
    n = 1 << log2n
    pts = hip.g2_mul_batch(g2, random_fr_limbs(n, SEED + 1))
    sc = random_fr_limbs(n, SEED + 2)
    srs = hip.srs_g2_upload(pts)
    for tables in (False, True):
        if tables:
            t0 = time.perf_counter(); nb = hip.srs_g2_precompute(srs); tb = time.perf_counter() - t0
        hip.msm_g2(srs, sc)
        t0 = time.perf_counter()
        for _ in range(3):
            hip.msm_g2(srs, sc)
        dt = (time.perf_counter() - t0) / 3
        d_s = torch.from_numpy(sc.view(np.int64)).cuda(); d_o = torch.zeros(24, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        hip.msm_g2_dev(srs, d_s.data_ptr(), n, d_o.data_ptr()); hip.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            hip.msm_g2_dev(srs, d_s.data_ptr(), n, d_o.data_ptr())
        hip.synchronize()
        dd = (time.perf_counter() - t0) / 3
        print("G2 MSM 2^%d%s, scalars resident: %.2f ms (%.2e scalar-mults/s)" % (log2n, " with window tables" if tables else "", dd * 1e3, n / dd), flush=True)
        print("G2 MSM 2^%d%s: %.2f ms incl. %d MB scalar upload (%.2e scalar-mults/s)%s" % (
            log2n, " with window tables" if tables else "", dt * 1e3, n * 32 >> 20, n / dt, " [tables: %.0f MB built in %.2f s]" % (nb / 1e6, tb) if tables else ""), flush=True)
    srs.free()
