#!/usr/bin/env python3
"""Laconic OT end-to-end -- the reference's integration flow (tests/laconic_ot.rs:15-200) at any size, on 1..N GPUs.

    python laconic_ot.py --log2n 16                      # N_CHOICES = 2^16 receiver bits, 2 x 32-byte messages per bit, one GPU
    python laconic_ot.py --gpus N --log2n 20             # BASELINE config 5: one rank per GPU (RCCL); starts its own ranks as a child
                                                         # torch.distributed.run when it is not already inside one (keaki_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        laconic_ot.py --gpus N --log2n 20                # the same through the launcher directly

Receiver::new  -> vec_commit : pad with one random scalar, iFFT to coefficients (GPU Fr FFT), FK23 openings (GPU Fr + G1 FFTs), commit (GPU MSM)
Sender::send   -> 2 x vec_encrypt : one batched GPU encapsulation per message set (fixed-base GT path), XOR on host
Receiver::receive -> vec_decrypt : one batched GPU decapsulation (one pairing per item), XOR on host

With N ranks (keaki_amd/dist.py; host side keaki::dist in keaki_amd/host/keaki.hpp):
  * every rank builds the same setup (same secret) and holds the whole SRS (2^21 points = 128 MiB);
  * vec_commit: the padding draw and the scalar-field iFFT are replicated; the FK23 openings are SHARDED (every rank runs 1/N of the
    butterflies of the two size-d group FFTs and of the 2d scalar-mults; two all-to-alls of 96-byte points and one all-gather of the affine
    proofs: keaki_amd/dist.py::ShardedFk; `--fk replicated` or a world that is not a power of two falls back to every rank computing all
    proofs), the commit MSM is sharded by point range: one all-gather of 96-byte partials + N - 1 EC additions;
  * vec_encrypt / vec_decrypt: sharded by item, NO collective; every rank draws the whole stream of r so that the ciphertexts are the
    single-process ones. Rank q decrypts the items it encrypted, so nothing moves between ranks; the recovered messages are checked
    on the rank that holds them and the verdicts are combined at the end (one all-gather of a flag).
Prints the same three phase timings the reference test prints (tests/laconic_ot.rs:148,176,188) as one JSON line (rank 0; the MAX over
ranks of every phase) and checks that every decrypted message is the chosen one. Exit code 1 otherwise.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from keaki_amd.launch import self_launch, under_launcher  # noqa: E402  (standard library only: no torch, no HIP)


def run_flow(K, shard, device, log2n, value_bytes=32, fk_mode="sharded", check_single=False, backend=None):
    """The whole flow on the ranks of `shard` (one rank: the un-sharded calls). Every rank calls this; returns the report (a dict; the same on
    every rank except for `sharded_equals_single_process`, which only rank 0 computes) whose `all_messages_recovered` is the AND over all
    ranks. bench.py's `laconic` block is this function on the bench's own process group."""
    from keaki_amd.dist import ShardedFk, sharded_vec_commit, sharded_vec_encrypt
    rank, world = shard.rank, shard.world

    def phase_max(t):
        return float(shard.all_gather_np(np.array([np.float64(t)]).view(np.uint64)).view(np.float64).max())

    n = 1 << log2n
    vb = value_bytes
    rng = K.Rng(2024)                              # the same stream on every rank: setup secret, padding, r values
    t0 = time.time()
    setup_degree = 1
    while setup_degree < n + K.PADDING_LEN:
        setup_degree <<= 1
    s = K.KZGSetup.setup(rng.fr_rand(), setup_degree, device)      # SETUP_DEGREE: the domain of n+1 evaluations
    fk = None
    if fk_mode == "sharded" and ShardedFk.can_shard(K, s, setup_degree, shard):
        fk = ShardedFk(K, s, setup_degree, shard, device)
        fk.prepare()                                       # this rank's part of the SRS-only transform (one all-to-all)
    else:
        K.precompute_open_fk(s, setup_degree)              # SRS-only part of the FK23 openings, like the MSM window tables
    K.prepare_shard(s, rank, world)                        # window tables of this rank's SRS chunk (N > 1)
    K.kem_prepare(s, (n + world - 1) // world)             # the tables of encapsulation that depend on the setup only, for this rank's share
    shard.barrier()
    t_setup = time.time() - t0
    np_rng = np.random.default_rng(7)
    bits = np_rng.integers(0, 2, n)
    zero, one = K.fr(0), K.fr(1)
    choices = np.where(bits[:, None] == 0, zero[None, :], one[None, :]).astype(np.uint64)

    t0 = time.time()
    commitment, proofs = sharded_vec_commit(K, rng, s, choices, shard, fk)  # Receiver::new
    t_receiver_new = phase_max(time.time() - t0)

    sets = [np_rng.integers(0, 256, size=(n, vb), dtype=np.uint8) for _ in range(2)]
    elements = K.domain_elements(n + K.PADDING_LEN)
    t0 = time.time()
    zeros, ones = np.repeat(zero[None, :], n, 0), np.repeat(one[None, :], n, 0)
    (lo, hi), g2_0, body_0 = sharded_vec_encrypt(K, rng, s, commitment, elements, zeros, sets[0], shard)     # Sender::send: set b to "bit i == b"
    _, g2_1, body_1 = sharded_vec_encrypt(K, rng, s, commitment, elements, ones, sets[1], shard)
    t_sender_send = phase_max(time.time() - t0)

    pick = bits[lo:hi, None] == 0                                                              # Receiver::receive, this rank's items
    sel_g2, sel_body = np.where(pick, g2_0, g2_1), np.where(pick, body_0, body_1)             # the ciphertext of each bit (references in the reference's loop)
    t0 = time.time()
    got = K.vec_decrypt_arrays(s, proofs[lo:hi], sel_g2, sel_body)
    t_receive = phase_max(time.time() - t0)
    ok = bool(np.array_equal(got, np.where(pick, sets[0][lo:hi], sets[1][lo:hi])))
    # the receiver must NOT be able to read the other message: decrypting the unchosen ciphertext gives something else
    m = min(256, hi - lo)
    if m:
        other = K.vec_decrypt_arrays(s, proofs[lo:lo + m], np.where(pick, g2_1, g2_0)[:m], np.where(pick, body_1, body_0)[:m])
        ok = ok and not np.array_equal(other, np.where(pick, sets[1][lo:hi], sets[0][lo:hi])[:m])
    single = None
    if check_single and rank == 0:
        # the un-sharded calls with the same seeds: commitment, proofs and this rank's ciphertexts must be the same bytes
        rng1 = K.Rng(2024)
        rng1.fr_rand()                                                                         # the setup secret
        com1, proofs1 = K.vec_commit(rng1, s, choices)
        a0, b0 = K.vec_encrypt_arrays(rng1, s, com1, elements, zeros, sets[0])
        a1, b1 = K.vec_encrypt_arrays(rng1, s, com1, elements, ones, sets[1])
        single = bool(np.array_equal(com1, commitment) and np.array_equal(proofs1, proofs) and np.array_equal(a0[lo:hi], g2_0)
                      and np.array_equal(b0[lo:hi], body_0) and np.array_equal(a1[lo:hi], g2_1) and np.array_equal(b1[lo:hi], body_1))
        ok = ok and single
    all_ok = bool(shard.all_gather_np(np.array([1 if ok else 0], np.uint64)).min() == 1)
    report = {"flow": "laconic_ot", "n_choices": n, "value_bytes": vb, "n_gpus": world, "ranks_seen": shard.world,
              "backend": backend if world > 1 else None, "setup_s": round(t_setup, 3),
              "receiver_new_s": round(t_receiver_new, 3), "sender_send_s": round(t_sender_send, 3),
              "receiver_receive_s": round(t_receive, 3),
              "bits_per_s_end_to_end": n / (t_receiver_new + t_sender_send + t_receive), "all_messages_recovered": all_ok,
              "sharded_equals_single_process": single,
              "fk_sharded": fk is not None, "fk_exchange_bytes_sent_per_rank": fk.bytes_moved if fk is not None else 0,
              "sharding": None if world == 1 else "commit MSM by point range (1 all-gather of 96-B partials); FK23 openings %s; "
                                                  "encaps / decaps by item, no collective"
                                                  % ("sharded (2 all-to-alls of 96-B points + 1 all-gather of the proofs per call)"
                                                     if fk is not None else "replicated on every rank"),
              "note": "wall-clock through the C++ host mirror (contiguous arrays in, arrays out), max over ranks per phase; "
                      "GPU work: FK23 + MSM / 2n encaps / n decaps"}
    if fk is not None:
        fk.close()
    s.close()
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=10)
    ap.add_argument("--value-bytes", type=int, default=32)   # VALUE_BYTES (tests/laconic_ot.rs:124)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo lets several ranks share one GPU (1-GPU box)")
    ap.add_argument("--fk", default="sharded", choices=["sharded", "replicated"], help="FK23 openings of Receiver::new on N > 1 ranks")
    ap.add_argument("--check-single", action="store_true",
                    help="rank 0 also runs the un-sharded calls with the same seeds and compares commitment and ciphertexts bit for bit")
    args = ap.parse_args()
    if args.gpus > 1 and not under_launcher():
        # typed without torch.distributed.run: this process -- which has not imported torch or touched the GPU -- starts the N ranks as a
        # child job and hands back its exit code (keaki_amd/launch.py)
        raise SystemExit(self_launch(__file__, sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from keaki_amd import keaki as K
    from keaki_amd.dist import Shard
    dist = None
    device = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()          # does not initialise the GPU
        if ndev == 0:
            raise SystemExit("laconic_ot.py needs an MI355X (there is no CPU fallback)")
        if args.backend == "nccl" and world > ndev:
            raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank (use --backend gloo to share a GPU)" % (world, ndev))
        device = local_rank % ndev
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        torch.cuda.init()                         # torch first, the library second (the exchange buffers of the sharded FK23 are torch tensors)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = Shard(rank, world, dist)
    report = run_flow(K, shard, device, args.log2n, args.value_bytes, args.fk, args.check_single, args.backend)
    if rank == 0:
        print(json.dumps(report), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not report["all_messages_recovered"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
