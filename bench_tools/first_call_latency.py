import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from bench import random_fr_limbs, SEED
from keaki_amd.hip import KeakiHip
dev = torch.device("cuda", 0)
hip = KeakiHip(0, torch.cuda.current_stream(dev).cuda_stream)
P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
mont = lambda v: [((v << 256) % P_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
g1 = np.array(mont(1) + mont(2), np.uint64)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634), (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))
g2 = np.array(mont(G2[0][0]) + mont(G2[0][1]) + mont(G2[1][0]) + mont(G2[1][1]), np.uint64)
sk = random_fr_limbs(3, SEED + 99)
com = hip.g1_mul_batch(g1, sk[0:1])[0]; tau_g2 = hip.g2_mul_batch(g2, sk[1:2])[0]
n = 1 << 16
T = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
d_com, d_tau = T(com), T(tau_g2)
d_pts, d_vals, d_rs = T(random_fr_limbs(n, 3)), T(random_fr_limbs(n, 4)), T(random_fr_limbs(n, 5))
d_ct = torch.zeros((n, 16), dtype=torch.int64, device=dev); d_gt = torch.zeros((n, 48), dtype=torch.int64, device=dev); d_key = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    hip.encap_batch_dev(d_com.data_ptr(), d_tau.data_ptr(), d_pts.data_ptr(), d_vals.data_ptr(), d_rs.data_ptr(), n, d_ct.data_ptr(), d_gt.data_ptr(), d_key.data_ptr(), 32)
    torch.cuda.synchronize()
    print("encap call %d: %.1f ms" % (k, (time.perf_counter() - t0) * 1e3))
