"""An MSM over a PREFIX of an SRS that has window tables: the tables' shared window (option msm_short_tables = 1) against the generic path
(0), by SRS size and polynomial length; the affine results must be equal."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs
from oracle import bn254_py as py
limbs = lambda x: np.frombuffer(int(x).to_bytes(32, "little"), np.uint64)
g1 = np.concatenate([limbs(py.G1_GEN[0] * (1 << 256) % py.P), limbs(py.G1_GEN[1] * (1 << 256) % py.P)])
h = KeakiHip(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
for log_srs in (10, 16, 20, 24):
    N = 1 << log_srs
    d_gen = dev(g1); d_k = dev(random_fr_limbs(N, 9)); d_pts = torch.empty(N * 8, dtype=torch.int64, device="cuda")
    h.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), N, d_pts.data_ptr())
    srs = h.srs_g1_wrap_dev(d_pts.data_ptr(), N)
    h.srs_g1_precompute(srs)
    sc = dev(random_fr_limbs(N, 3))
    out = torch.empty(12, dtype=torch.int64, device="cuda")
    line = "SRS 2^%d:" % log_srs
    for ln in sorted(set([0, 3, 7, log_srs - 8, log_srs - 4, log_srs - 2, log_srs - 1])):
        if ln < 0 or ln >= log_srs: continue
        n = (1 << ln) + (1 if ln else 0)
        res = []
        for opt in (0, 1):
            h.set_option("msm_short_tables", opt)
            h.msm_g1_dev(srs, sc.data_ptr(), n, out.data_ptr()); h.synchronize(); ref = out.cpu().numpy().view(np.uint64).copy()
            t0 = time.perf_counter()
            for _ in range(5): h.msm_g1_dev(srs, sc.data_ptr(), n, out.data_ptr())
            h.synchronize(); res.append(((time.perf_counter() - t0) / 5 * 1e3, ref))
        # Jacobian outputs of the two paths may differ by a scaling: compare X Z'^2 = X' Z^2 through the oracle's integers
        def aff(j):
            x, y, z = (int.from_bytes(j[4 * i:4 * i + 4].tobytes(), "little") * py.FQ_RINV % py.P for i in range(3))
            if z == 0: return None
            zi = pow(z, -1, py.P); return (x * zi * zi % py.P, y * zi * zi * zi % py.P)
        same = aff(res[0][1]) == aff(res[1][1])
        line += "  n=%d: generic %.3f / tables %.3f ms%s" % (n, res[0][0], res[1][0], "" if same else " RESULTS DIFFER")
    print(line, flush=True)
    srs.free()
