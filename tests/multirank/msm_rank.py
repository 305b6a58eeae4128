#!/usr/bin/env python3
"""One rank of the world-2 composition test of the sharded MSM on REAL hardware (tests/test_gpu_world2.py launches two of these as
fresh processes, before the pytest process itself touches the GPU). Both ranks share GPU 0, each with its own keaki context on its own
torch stream; the exchange is torch.distributed with the gloo backend (RCCL refuses two ranks on one device), i.e. exactly
keaki_amd/dist.py::ShardedMsm with the transport swapped:

    keaki_hip_msm_g1_dev on the rank's chunk -> all-gather of the 96-byte partials -> keaki_hip_g1_sum_dev

The global instance is seeded: n_total (scalar, point) pairs, points k_i G. Every rank generates ONLY its chunk. Three steps with three
different scalar vectors (a step that read a stale partial would give a wrong sum). Each rank writes its results to
<out>/rank<q>.npz; the parent checks them against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, n_total, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch
    import torch.distributed as dist
    from bench import random_fr_limbs, mont_words
    from keaki_amd.hip import KeakiHip
    from keaki_amd.dist import Shard, ShardedMsm, chunk_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    hip = KeakiHip(0, stream.cuda_stream)
    shard = Shard(rank, world, dist)
    lo, hi = chunk_bounds(n_total, world, rank)
    k_all = random_fr_limbs(n_total, 0xA11CE)                         # the global instance (cheap to draw; only the chunk is used)
    d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
    d_k = torch.from_numpy(k_all[lo:hi].view(np.int64).copy()).to(dev)
    d_pts = torch.empty((hi - lo, 8), dtype=torch.int64, device=dev)
    hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), hi - lo, d_pts.data_ptr())
    sm = ShardedMsm(hip, shard, d_pts.data_ptr(), hi - lo, dev)
    if rank == 0:
        sm.precompute()                                               # one rank with window tables, one without: both paths in one sum
    results, partials = [], []
    for step in range(3):
        s_all = random_fr_limbs(n_total, 0xB0B + step)
        d_s = torch.from_numpy(s_all[lo:hi].view(np.int64).copy()).to(dev)
        res = sm.run(d_s.data_ptr())
        torch.cuda.synchronize(dev)
        results.append(res.cpu().numpy().view(np.uint64).copy())
        partials.append(sm.part.cpu().numpy().view(np.uint64).copy())
    pts_host = d_pts.cpu().numpy().view(np.uint64)
    np.savez(os.path.join(out, "rank%d.npz" % rank), results=np.stack(results), partials=np.stack(partials), lo=lo, hi=hi,
             pts_head=pts_host[:64], table_bytes=sm.table_bytes)
    dist.barrier()
    dist.destroy_process_group()
    sm.close()
    hip.close()


if __name__ == "__main__":
    main()
