"""world_size-2 gloo test of the MSM sharding path (keaki_amd/dist.py) on CPU: contiguous chunks, one
all-gather of 96-byte partials, EC-add combine. The per-rank Pippenger and the combine are played by
the CPU oracle here (no GPU in this container); on the GPU box the same functions are driven by the
HIP kernels (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as oc
    from keaki_amd.dist import chunk_bounds, sharded_msm, torch_all_gather
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest_helpers import rand_fr_ints
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g1, _ = oc.generators()
    mont = lambda v: oc.fr_to_mont(oc.ints_to_limbs(v))
    pts = oc.g1_mul_batch(g1, mont(rand_fr_ints(n_total, 1)), threads=2)
    sc = mont(rand_fr_ints(n_total, 2))
    lo, hi = chunk_bounds(n_total, world, rank)
    one = oc.fq_to_mont(oc.ints_to_limbs([1]))[0]

    def to_jac(aff):  # affine words -> normalised Jacobian words (x, y, 1) / (1, 1, 0)
        j = np.zeros(12, np.uint64)
        if np.any(aff):
            j[:8] = aff; j[8:] = one
        else:
            j[:4] = one; j[4:8] = one
        return j

    def partial_fn():
        return torch.from_numpy(to_jac(oc.msm_g1(pts[lo:hi], sc[lo:hi])).view(np.int64).copy())

    all_buf = torch.zeros((world, 12), dtype=torch.int64)

    def sum_fn(allp):
        a = allp.numpy().view(np.uint64)
        aff = np.stack([a[i, :8] if np.any(a[i, 8:]) else np.zeros(8, np.uint64) for i in range(world)])
        return oc.g1_sum(aff)

    res = sharded_msm(partial_fn, torch_all_gather(dist, all_buf), sum_fn, world)
    full = oc.msm_g1(pts, sc)
    q.put((rank, bool(np.array_equal(res, full)), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [101, 256])
def test_sharded_msm_world2_gloo(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_total % 7
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    bounds = sorted(b for _, _, b in results)
    assert bounds[0][0] == 0 and bounds[0][1] == bounds[1][0] and bounds[1][1] == n_total


def test_chunk_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from keaki_amd.dist import chunk_bounds
    for n in (0, 1, 7, 8, 1 << 20, (1 << 26) + 3):
        for w in (1, 2, 4, 8):
            b = [chunk_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
