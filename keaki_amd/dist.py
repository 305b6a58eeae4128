"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on a GPU node, "gloo" in the CPU tests
and wherever several ranks share one GPU). Nothing here computes: the group arithmetic is the C ABI's, this module decides WHO
computes WHAT and moves the 96-byte partial sums.

What shards how:

* kzg::commit's MSM (reference src/kzg.rs:98) -- by contiguous (scalar, point) chunk. Rank q holds SRS points [lo_q, hi_q) (and the
  window tables of that chunk only), runs a complete Pippenger on them and emits one normalised-Jacobian partial; ONE exchange step:
  an all-gather of the world_size partials (EC addition is not an RCCL reduction operator), then world_size - 1 EC additions on
  every rank (keaki_hip_g1_sum[_dev]). 96 bytes per rank: latency-bound, xGMI bandwidth is irrelevant.
* vec_encrypt / vec_decrypt (src/vec.rs:63-66, :75-78) -- by item, no collective. Every rank draws the WHOLE stream of r values so that
  the ciphertexts equal the single-process ones.
* kzg::open_fk (FK23, src/kzg.rs:157-203) inside vec_commit -- its group transforms (one inverse and one forward FFT of size d on the odd
  half of the 2d products; the even half needs none: fft_g1.hip) and the 2d scalar-mults are split over the ranks (ShardedFk below;
  keaki_hip_fk_shard_* in the C ABI): two all-to-alls of 96-byte points (d / world^2 points per peer each) and one all-gather of the
  d / world affine proofs per rank and call, one all-to-all (2d / world^2 per peer) at setup. The ONLY exchange of the path where xGMI
  bandwidth matters (d = 2^21, 8 ranks: 161 MB sent per rank and call). Needs a power-of-two world and d >= world^2; otherwise replicated.
"""
from __future__ import annotations

import numpy as np


def chunk_bounds(n_total: int, world: int, rank: int):
    """contiguous chunk [lo, hi) of rank `rank`; sizes differ by at most one (keaki::dist::Shard::bounds)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Shard:
    """This process's place in the job. world == 1 needs no process group at all -- unless `force_collectives` is set: then every
    exchange goes through torch.distributed even with one rank (how the RCCL path -- init, all-gather on the stream the kernels run on,
    barrier -- is executed on a box with a single GPU: bench.py --force-collectives)."""

    def __init__(self, rank: int = 0, world: int = 1, dist_module=None, force_collectives: bool = False):
        self.rank, self.world, self.dist = rank, world, dist_module
        if (world > 1 or force_collectives) and dist_module is None:
            raise ValueError("world > 1 needs the initialised torch.distributed module")
        self.collective = world > 1 or force_collectives

    @classmethod
    def from_env(cls):
        """under torch.distributed.run: uses the default process group if one is initialised"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return cls(dist.get_rank(), dist.get_world_size(), dist)
        return cls()

    def bounds(self, n: int):
        return chunk_bounds(n, self.world, self.rank)

    def all_gather_rows(self, row):
        """row: 1-D torch int64 tensor (on the device for RCCL, anywhere for gloo) -> (world, len) tensor holding every rank's row,
        on the device `row` lives on. THE exchange step of the sharded MSM."""
        if not self.collective:
            return row.reshape(1, -1)
        import torch
        backend = self.dist.get_backend()
        if backend == "nccl":
            out = torch.empty((self.world, row.numel()), dtype=row.dtype, device=row.device)
            self.dist.all_gather_into_tensor(out.view(-1), row.contiguous())       # ordered on the current stream
            return out
        host = row.detach().cpu().contiguous()                                     # .cpu() waits for the producer on the current stream
        out = torch.empty((self.world, host.numel()), dtype=host.dtype)
        self.dist.all_gather_into_tensor(out.view(-1), host)
        return out.to(row.device)

    def all_gather_np(self, words: np.ndarray) -> np.ndarray:
        """host variant: u64[k] per rank -> (world, k)"""
        if not self.collective:
            return words.reshape(1, -1).copy()
        import torch
        t = torch.from_numpy(np.ascontiguousarray(words).view(np.int64))
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        return self.all_gather_rows(t).cpu().numpy().view(np.uint64)

    def barrier(self):
        if self.collective:
            self.dist.barrier()


class ShardedMsm:
    """One rank's chunk of a G1 MSM whose (scalar, point) pairs live in HBM: the configuration of BASELINE configs 2-4.

        sm = ShardedMsm(hip, shard, d_points_chunk_ptr, n_chunk)    # the rank's points, already resident (SRS upload = setup time)
        sm.precompute()                                             # optional window tables of the chunk
        out = sm.run(d_scalars_chunk_ptr)                           # device tensor of 12 words: the FULL sum, on every rank

    `hip` must have been created on the stream torch uses (KeakiHip(device, torch.cuda.current_stream().cuda_stream)) so that the
    MSM, the RCCL all-gather and the final additions are ordered without host synchronisation."""

    def __init__(self, hip, shard: Shard, d_points_ptr: int, n_chunk: int, torch_device=None):
        import torch
        self.hip, self.shard, self.n = hip, shard, n_chunk
        self.srs = hip.srs_g1_wrap_dev(d_points_ptr, n_chunk)
        dev = torch_device if torch_device is not None else torch.device("cuda", torch.cuda.current_device())
        self.part = torch.zeros(12, dtype=torch.int64, device=dev)
        self.out = torch.zeros(12, dtype=torch.int64, device=dev)
        self.table_bytes = 0
        self._all = None

    def precompute(self) -> int:
        self.table_bytes = self.hip.srs_g1_precompute(self.srs)
        return self.table_bytes

    def run_partial(self, d_scalars_ptr: int, n: int | None = None):
        self.hip.msm_g1_dev(self.srs, d_scalars_ptr, self.n if n is None else n, self.part.data_ptr())
        return self.part

    def combine(self):
        """all-gather the partials, add them: afterwards self.out holds the whole MSM on every rank"""
        if not self.shard.collective:
            return self.part
        self._all = self.shard.all_gather_rows(self.part)
        self.hip.g1_sum_dev(self._all.data_ptr(), self.shard.world, self.out.data_ptr())
        return self.out

    def run(self, d_scalars_ptr: int, n: int | None = None):
        self.run_partial(d_scalars_ptr, n)
        return self.combine()

    def close(self):
        if self.srs is not None:
            self.srs.free()
            self.srs = None


# ---- the host-mirror flow (keaki_amd/keaki.py) sharded: what laconic_ot.py --gpus N runs -----------------------------------------------
def sharded_commit(K, setup, p, shard: Shard) -> np.ndarray:
    """kzg::commit with the MSM split by point range over the ranks; every rank returns the same affine commitment"""
    part = K.commit_partial(setup, p, shard.rank, shard.world)
    return K.commit_combine(setup, shard.all_gather_np(part))


def exchange_all_to_all(dist, send, recv):
    """`send`: world equal chunks laid end to end, chunk q for rank q; `recv`: the chunks received, in rank order. RCCL works on the device
    tensors themselves (ordered on torch's current stream); gloo goes through host memory when the tensors live on a GPU."""
    import torch
    if dist.get_backend() == "nccl" or not send.is_cuda:
        dist.all_to_all_single(recv, send)
    else:
        hs = send.cpu()
        hr = torch.empty_like(hs)
        dist.all_to_all_single(hr, hs)
        recv.copy_(hr)


def exchange_all_gather(dist, send, recv):
    """`recv` = every rank's `send`, in rank order"""
    import torch
    if dist.get_backend() == "nccl" or not send.is_cuda:
        dist.all_gather_into_tensor(recv, send)
    else:
        hs = send.cpu()
        hr = torch.empty(recv.numel(), dtype=recv.dtype)
        dist.all_gather_into_tensor(hr, hs)
        recv.copy_(hr)


class ShardedFk:
    """kzg::open_fk of a setup sharded over the ranks: owns the two exchange buffers (torch tensors on the setup's GPU) and performs the
    collectives keaki::dist::ShardedOpenFk asks for -- RCCL on the device buffers themselves, or through host memory for gloo.

        fk = ShardedFk(K, setup, domain_size, shard); fk.prepare()       # setup time: this rank's part of hat_s (one all-to-all)
        proofs = fk.open(coeffs)                                          # == K.open_fk(setup, coeffs, domain_size) on every rank
    """

    def __init__(self, K, setup, domain_size: int, shard: Shard, device: int = 0):
        import torch
        self.K, self.shard, self.torch = K, shard, torch
        self.dev = torch.device("cuda", device)
        self.inner = K.ShardedOpenFk(setup, domain_size, shard.rank, shard.world, self._all_to_all, self._all_gather)
        nb = self.inner.buffer_bytes
        self.send = torch.empty(nb, dtype=torch.uint8, device=self.dev)
        self.recv = torch.empty(nb, dtype=torch.uint8, device=self.dev)
        self.bytes_moved = 0

    @staticmethod
    def can_shard(K, setup, domain_size: int, shard: Shard) -> bool:
        return shard.world > 1 and K.ShardedOpenFk.can_shard(setup, domain_size, shard.rank, shard.world)

    # the library has synchronised its stream before it calls these; they return when d_recv is complete
    def _all_to_all(self, d_send, d_recv, per_peer):
        w = self.shard.world
        assert d_send == self.send.data_ptr() and d_recv == self.recv.data_ptr()
        exchange_all_to_all(self.shard.dist, self.send[:w * per_peer], self.recv[:w * per_peer])
        self.torch.cuda.synchronize(self.dev)
        self.bytes_moved += (w - 1) * per_peer

    def _all_gather(self, d_send, d_recv, per_rank):
        w = self.shard.world
        exchange_all_gather(self.shard.dist, self.send[:per_rank], self.recv[:w * per_rank])
        self.torch.cuda.synchronize(self.dev)
        self.bytes_moved += (w - 1) * per_rank

    def prepare(self):
        self.inner.prepare(self.send.data_ptr(), self.recv.data_ptr())

    def open(self, coeffs):
        return self.inner.open(coeffs, self.send.data_ptr(), self.recv.data_ptr())

    def vec_commit_partial(self, rng, v):
        return self.inner.vec_commit_partial(rng, v, self.send.data_ptr(), self.recv.data_ptr())

    def close(self):
        self.inner.close()
        self.send = self.recv = None


def sharded_vec_commit(K, rng, setup, v, shard: Shard, fk: "ShardedFk | None" = None):
    """vec_commit (src/vec.rs:22-49) on every rank: padding draw and iFFT replicated (same rng seed -> same values), the commit MSM sharded by
    point range with one all-gather of the partials, and -- given a ShardedFk -- the FK23 openings sharded as well (without one they are
    replicated). Returns (commitment, proofs) on every rank."""
    if not shard.collective and fk is None:
        return K.vec_commit(rng, setup, v)              # one rank: the un-sharded call (one device call behind the padding draw)
    if fk is not None:
        part, proofs = fk.vec_commit_partial(rng, v)
    else:
        part, proofs = K.vec_commit_partial(rng, setup, v, shard.rank, shard.world)
    return K.commit_combine(setup, shard.all_gather_np(part)), proofs


def sharded_vec_encrypt(K, rng, setup, com, points, values, messages, shard: Shard):
    """this rank's item range of vec_encrypt -> ((lo, hi), ct G2 points, ct bodies); no collective"""
    n = np.asarray(messages).shape[0]
    g2, body = K.vec_encrypt_arrays_shard(rng, setup, com, points, values, messages, shard.rank, shard.world)
    return shard.bounds(n), g2, body


def sharded_vec_decrypt(K, setup, proofs, ct_g2, ct_body, shard: Shard):
    """this rank's item range of vec_decrypt -> ((lo, hi), messages); no collective"""
    n = np.asarray(ct_body).shape[0]
    lo, hi = shard.bounds(n)
    return (lo, hi), K.vec_decrypt_arrays(setup, proofs[lo:hi], ct_g2[lo:hi], ct_body[lo:hi])
