"""world_size-2 gloo tests of the multi-GPU layer (keaki_amd/dist.py) on CPU: contiguous chunks, ONE all-gather of 96-byte partials,
EC-add combine; item-sharded vec_encrypt / vec_decrypt without any collective. There is no GPU in this container, so the per-rank
group arithmetic is played by the CPU oracle behind a stub of the host mirror's `keaki::dist` calls; the sharding, the exchange
(Shard.all_gather_np / all_gather_rows) and the flow functions are the product's. On the GPU box tests/test_gpu_world2.py runs
the same layer with the HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleK:
    """stand-in for keaki_amd.keaki (libkeaki_host.so needs a GPU): the calls keaki_amd/dist.py makes, computed by the oracle"""
    PADDING_LEN = 1

    def __init__(self, oc, srs_pts):
        self.oc, self.pts = oc, srs_pts
        self.one = oc.fq_to_mont(oc.ints_to_limbs([1]))[0]

    def _jac(self, aff):
        j = np.zeros(12, np.uint64)
        if np.any(aff):
            j[:8] = aff; j[8:] = self.one
        else:
            j[:4] = self.one; j[4:8] = self.one
        return j

    def commit_partial(self, setup, p, rank, world):
        from keaki_amd.dist import chunk_bounds
        lo, hi = chunk_bounds(self.pts.shape[0], world, rank)        # ranges follow the SRS (keaki::dist::commit_partial)
        lo, hi = min(lo, p.shape[0]), min(hi, p.shape[0])
        return self._jac(self.oc.msm_g1(self.pts[lo:hi], p[lo:hi]))

    def vec_commit_partial(self, rng, setup, v, rank, world):
        pad = rng.fr_rand()
        coeffs = np.concatenate([v, pad[None, :]], 0)                # (the iFFT is irrelevant to what is tested here)
        return self.commit_partial(setup, coeffs, rank, world), np.zeros((coeffs.shape[0], 8), np.uint64)

    def commit_combine(self, setup, partials):
        a = np.asarray(partials).reshape(-1, 12)
        aff = np.stack([a[i, :8] if np.any(a[i, 8:]) else np.zeros(8, np.uint64) for i in range(a.shape[0])])
        return self.oc.g1_sum(aff)

    def vec_encrypt_arrays_shard(self, rng, setup, com, points, values, messages, rank, world):
        from keaki_amd.dist import chunk_bounds
        n = messages.shape[0]
        rs = np.stack([rng.fr_rand() for _ in range(n)])             # the WHOLE stream on every rank
        lo, hi = chunk_bounds(n, world, rank)
        ct, _, key = self.oc.encap_batch(com, setup["tau_g2"], points[lo:hi], values[lo:hi], rs[lo:hi], messages.shape[1])
        return ct, key ^ messages[lo:hi]

    def vec_decrypt_arrays(self, setup, proofs, ct_g2, ct_body):
        _, key = self.oc.decap_batch(proofs, ct_g2, ct_body.shape[1])
        return key ^ ct_body


class SeqRng:
    def __init__(self, oc, seed):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest_helpers import rand_fr_ints
        self.vals = oc.fr_to_mont(oc.ints_to_limbs(rand_fr_ints(64, seed))); self.i = 0

    def fr_rand(self):
        self.i += 1
        return self.vals[self.i - 1].copy()


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as oc
    from keaki_amd.dist import Shard, chunk_bounds, sharded_commit, sharded_vec_commit, sharded_vec_encrypt, sharded_vec_decrypt
    from conftest_helpers import rand_fr_ints
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = Shard.from_env()
    assert (shard.rank, shard.world) == (rank, world)
    g1, g2 = oc.generators()
    mont = lambda v: oc.fr_to_mont(oc.ints_to_limbs(v))
    pts = oc.g1_mul_batch(g1, mont(rand_fr_ints(n_total, 1)), threads=2)
    sc = mont(rand_fr_ints(n_total, 2))
    K = OracleK(oc, pts)
    ok = {}
    # commit: full-length polynomial, a short one (the last rank's range is empty), the empty one
    for name, m in (("full", n_total), ("short", n_total // 3), ("empty", 0)):
        ok["commit_" + name] = bool(np.array_equal(sharded_commit(K, None, sc[:m], shard), oc.msm_g1(pts[:m], sc[:m])))
    # the device-tensor exchange (what ShardedMsm.combine uses), here on CPU tensors
    row = torch.arange(12, dtype=torch.int64) + 100 * rank
    allr = shard.all_gather_rows(row)
    ok["rows"] = bool(allr.shape == (world, 12) and all(int(allr[r, 0]) == 100 * r for r in range(world)))
    # vec_commit: every rank draws the same padding; the combined commitment equals the un-sharded MSM
    v = sc[:n_total - 1]
    com, proofs = sharded_vec_commit(K, SeqRng(oc, 5), None, v, shard)
    pad = SeqRng(oc, 5).fr_rand()
    ok["vec_commit"] = bool(np.array_equal(com, oc.msm_g1(pts, np.concatenate([v, pad[None, :]], 0))))
    # vec_encrypt / vec_decrypt by item, no collective: the rank's ciphertexts are the single-process ones, its messages come back
    n_items = 7
    setup = {"tau_g2": oc.g2_mul_batch(g2, mont([12345]))[0]}
    msgs = np.frombuffer(bytes(range(n_items * 32)), np.uint8).reshape(n_items, 32).copy()
    points, values = mont(rand_fr_ints(n_items, 8)), mont(rand_fr_ints(n_items, 9))
    (lo, hi), ct, body = sharded_vec_encrypt(K, SeqRng(oc, 6), setup, pts[0], points, values, msgs, shard)
    rs = SeqRng(oc, 6).vals[:n_items]
    ect, _, ekey = oc.encap_batch(pts[0], setup["tau_g2"], points, values, rs, 32)
    ok["vec_encrypt"] = bool((lo, hi) == chunk_bounds(n_items, world, rank) and np.array_equal(ct, ect[lo:hi]) and np.array_equal(body, ekey[lo:hi] ^ msgs[lo:hi]))
    # decrypt with the matching proofs pi_i = (C - v_i g1) / (tau - p_i): any consistent (proof, ct) pair gives back the key, so use r_i-free identity:
    # e(proof, ct) == e(r (C - v g1), g2) needs a real KZG proof; here only the sharding is under test -> decap of (P, ct) on both paths
    (dlo, dhi), dec = sharded_vec_decrypt(K, None, np.repeat(pts[1:2], n_items, 0), ect, ekey ^ msgs, shard)
    _, k2 = oc.decap_batch(np.repeat(pts[1:2], n_items, 0), ect, 32)
    ok["vec_decrypt"] = bool((dlo, dhi) == (lo, hi) and np.array_equal(dec, (k2 ^ ekey ^ msgs)[lo:hi]))
    q.put((rank, ok, shard.bounds(n_total)))
    shard.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [101, 256])
def test_sharded_flow_world2_gloo(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_total % 7
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=240) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    for _, ok, _ in results:
        assert all(ok.values()), ok
    bounds = sorted(b for _, _, b in results)
    assert bounds[0][0] == 0 and bounds[0][1] == bounds[1][0] and bounds[1][1] == n_total


def test_chunk_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from keaki_amd.dist import chunk_bounds, Shard
    for n in (0, 1, 7, 8, 1 << 20, (1 << 26) + 3):
        for w in (1, 2, 4, 8):
            b = [chunk_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        Shard(0, 2, None)
    one = Shard()
    assert one.bounds(10) == (0, 10) and one.all_gather_np(np.arange(3, dtype=np.uint64)).shape == (1, 3)


# ---- the sharded FK23 pipeline across real processes: the exchange helpers of keaki_amd/dist.py (all_to_all_single / all_gather_into_tensor
# with equal splits) must move the chunks exactly as tests/fk_shard_model.py::all_to_all says; every process plays one rank of the model
def _fk_worker(rank, world, port, log2d, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random
    from fk_shard_model import Q, ShardModel, open_fk_plain
    from keaki_amd.dist import exchange_all_to_all, exchange_all_gather
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rnd = random.Random(4242 + log2d)                       # the same inputs on every rank
        d = 1 << log2d
        srs = [rnd.randrange(Q) for _ in range(d)]
        p = [rnd.randrange(Q) for _ in range(d)]
        m = ShardModel(srs, log2d, world)

        def a2a(values):
            send = torch.tensor(values, dtype=torch.int64)
            recv = torch.empty_like(send)
            exchange_all_to_all(dist, send, recv)
            return recv.tolist()
        m.hs_even, m.hs_odd = [None] * world, [None] * world
        m.hs_even[rank], m.hs_odd[rank] = m.setup_step1(rank, a2a(m.setup_step0(rank)))
        part = m.open_step2(rank, a2a(m.open_step1(rank, a2a(m.open_step0(rank, p)))))
        send = torch.tensor(part, dtype=torch.int64)
        recv = torch.empty(d, dtype=torch.int64)
        exchange_all_gather(dist, send, recv)
        got = m.open_step3(recv.tolist())
        q.put((rank, got == open_fk_plain(srs, p, log2d)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log2d", [(2, 4), (4, 5)])
def test_sharded_fk_exchanges_across_processes(world, log2d):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + world
    procs = [ctx.Process(target=_fk_worker, args=(r, world, port, log2d, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]
