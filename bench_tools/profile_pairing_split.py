"""Pairing kernel split, for rocprofv3 --kernel-trace --stats: 2^14 pairings through the fused kernel (k_pairing_batch) and through the
two test entry points (k_miller_only + k_final_exp_only). Says where a pairing's time goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import random_fr_limbs, SEED
from keaki_amd.hip import KeakiHip

hip = KeakiHip(0)
n = 1 << 14
P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
mont = lambda v: [((v << 256) % P_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
g1 = np.array(mont(1) + mont(2), np.uint64)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))
g2 = np.array(mont(G2[0][0]) + mont(G2[0][1]) + mont(G2[1][0]) + mont(G2[1][1]), np.uint64)
a, b = random_fr_limbs(n, SEED + 7), random_fr_limbs(n, SEED + 8)
ps = hip.g1_mul_batch(np.tile(g1, (n, 1)), a)
qs = hip.g2_mul_batch(np.tile(g2, (n, 1)), b)
for _ in range(3):
    gt = hip.pairing_batch(ps, qs)
    f = hip.miller_loop_batch(ps, qs)
    gt2 = hip.final_exp_batch(f)
assert np.array_equal(gt, gt2)
print("ok", n)
