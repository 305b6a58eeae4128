"""Multi-GPU sharding of the MSM: one process per GPU, contiguous point-chunks, ONE exchange step.

The MSM  sum_i s_i P_i  shards by (scalar, point) chunk: rank q owns pairs [lo_q, hi_q), runs a complete
Pippenger on them and produces one 96-byte normalised-Jacobian partial sum. EC addition is not an RCCL
reduction operator, so the "reduce" is an all-gather of the world_size partials (torch.distributed,
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) followed by world_size-1 EC adds
on every rank (keaki_hip_g1_sum_dev). The message is 96 B per rank: latency-bound, bandwidth-irrelevant.
Batched encapsulation / decapsulation shard by item with no collective at all.
"""
from __future__ import annotations


def chunk_bounds(n_total: int, world: int, rank: int):
    """contiguous chunk [lo, hi) of rank `rank`; sizes differ by at most one"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_msm(partial_fn, all_gather_fn, sum_fn, world: int):
    """partial_fn() -> this rank's partial (12 words); all_gather_fn(partial) -> (world, 12) on every rank;
    sum_fn(all_partials) -> the full MSM result (12 words). With world == 1 the partial is the result."""
    part = partial_fn()
    if world == 1:
        return part
    return sum_fn(all_gather_fn(part))


def torch_all_gather(dist_module, out_buf):
    """all-gather closure over a preallocated (world, 12) int64 tensor (device tensor on the GPU box)"""
    def fn(part):
        dist_module.all_gather_into_tensor(out_buf.view(-1), part)
        return out_buf
    return fn
