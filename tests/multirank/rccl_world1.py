"""RCCL on real hardware with ONE rank (the GPU box has one GPU): the exact torch.distributed calls of the N > 1 path -- process-group
init with backend "nccl" (= RCCL on ROCm), the all-gather of the 96-byte MSM partials on the stream the keaki context enqueues on
(keaki_amd/dist.py::Shard.all_gather_rows / ShardedMsm.combine), the all-to-all and all-gather of the sharded FK23 exchange buffers
(exchange_all_to_all / exchange_all_gather), barrier, teardown -- and the MSM result against the oracle. Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    import torch.distributed as dist
    from keaki_amd.hip import KeakiHip, jac_to_affine_words
    from keaki_amd.dist import Shard, ShardedMsm, exchange_all_to_all, exchange_all_gather
    from bench import random_fr_limbs, mont_words, SEED
    import oracle as oc
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    hip = KeakiHip(0, stream.cuda_stream)
    shard = Shard(0, 1, dist, force_collectives=True)
    n = 1 << 16
    k = random_fr_limbs(n, SEED + 1)
    s = random_fr_limbs(n, SEED + 2)
    d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
    d_k = torch.from_numpy(k.view(np.int64)).to(dev)
    d_s = torch.from_numpy(s.view(np.int64)).to(dev)
    d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
    hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr())
    sm = ShardedMsm(hip, shard, d_pts.data_ptr(), n, dev)
    sm.precompute()
    for _ in range(3):                       # MSM -> RCCL all-gather -> EC sum, back to back on one stream, no host sync in between
        res = sm.run(d_s.data_ptr())
    torch.cuda.synchronize(dev)
    g1, _ = oc.generators()
    exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k)[None, :])[0]
    out["msm_allgather_sum_ok"] = bool(np.array_equal(jac_to_affine_words(res.cpu().numpy().view(np.uint64)), exp))
    out["used_collective_output"] = bool(res.data_ptr() == sm.out.data_ptr())
    # the FK23 exchange shapes: all-to-all with equal splits and all-gather on byte buffers resident on the device
    send = torch.arange(96 * 1024, dtype=torch.int64, device=dev).to(torch.uint8)
    recv = torch.zeros_like(send)
    exchange_all_to_all(dist, send, recv)
    torch.cuda.synchronize(dev)
    out["all_to_all_ok"] = bool(torch.equal(send, recv))
    recv.zero_()
    exchange_all_gather(dist, send[:64 * 512], recv[:64 * 512])
    torch.cuda.synchronize(dev)
    out["all_gather_ok"] = bool(torch.equal(send[:64 * 512], recv[:64 * 512]))
    out["all_gather_np_ok"] = bool(np.array_equal(shard.all_gather_np(np.arange(12, dtype=np.uint64)), np.arange(12, dtype=np.uint64)[None, :]))
    # the torch-free companion library (include/keaki_hip_rccl.h): its own communicator on the same stream, world 1
    from keaki_amd import rccl as R
    rc = R.KeakiRccl(hip, R.unique_id(), 0, 1)
    d_out2 = torch.zeros(12, dtype=torch.int64, device=dev)
    for _ in range(2):
        rc.msm_g1(sm.srs, d_s.data_ptr(), n, d_out2.data_ptr())
    hip.synchronize()
    out["shim_msm_ok"] = bool(np.array_equal(jac_to_affine_words(d_out2.cpu().numpy().view(np.uint64)), exp))
    out["shim_status_ok"] = rc.collective_status() == -1
    # a rank-local failure (more scalars than the chunk holds) still takes part in the exchange -- with one rank: still completes --, returns its
    # own error at once, and the collective status names the rank
    from keaki_amd.hip import KeakiHipError
    try:
        rc.msm_g1(sm.srs, d_s.data_ptr(), n + 1, d_out2.data_ptr())
        out["shim_local_failure_ok"] = False
    except KeakiHipError:
        try:
            rc.collective_status()
            out["shim_local_failure_ok"] = False
        except KeakiHipError as e:
            out["shim_local_failure_ok"] = "rank 0" in str(e)
    rc.msm_g1(sm.srs, d_s.data_ptr(), n, d_out2.data_ptr())                       # and the communicator is still usable
    out["shim_after_failure_ok"] = rc.collective_status() == -1 and bool(np.array_equal(jac_to_affine_words(d_out2.cpu().numpy().view(np.uint64)), exp))
    recv.zero_()
    rc.all_to_all(send.data_ptr(), recv.data_ptr(), send.numel())
    hip.synchronize()
    out["shim_all_to_all_ok"] = bool(torch.equal(send, recv))
    recv.zero_()
    rc.all_gather(send.data_ptr(), recv.data_ptr(), 4096)
    hip.synchronize()
    out["shim_all_gather_ok"] = bool(torch.equal(send[:4096], recv[:4096]))
    rc.close()
    shard.barrier()
    sm.close()
    hip.close()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    if not all(v for k_, v in out.items() if k_.endswith("_ok")):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
