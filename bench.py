#!/usr/bin/env python3
"""bench.py -- headline benchmark: BN254 G1 scalar-mults/s on a 2^24-point Pippenger MSM per MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full MSM (kzg::commit's msm_unchecked, reference src/kzg.rs:98) over a batch of
synthetic scalars with the SRS and the scalars already resident in HBM. With N > 1 every rank holds
its own contiguous chunk of 2^LOG2N (scalar, point) pairs (weak scaling), runs a complete Pippenger
on it, and the 96-byte partial sums are exchanged with one RCCL all-gather followed by N-1 EC adds.

Prints ONE JSON line (rank 0). `roofline` prices the dominant kernel (bucket accumulation) against
HBM with the algorithmic 96 B per scalar-mult; `alu` prices it against the measured integer-issue
rate, which is what actually bounds this path; `cpu_baseline` times the CPU restatement of the
arkworks algorithm (oracle/, the checker -- never the product) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
SEED = 0x6B65616B69  # "keaki"
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_SCALAR_MUL = 96   # 32 B scalar + 64 B affine point, each read once


def splitmix64_stream(seed, count):
    """vectorised SplitMix64: `count` u64 values of the stream seeded with `seed`."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_fr_limbs(n, seed):
    """n field elements uniform in [0, r) as (n,4) u64 limbs, the way ark-ff's Fr::rand draws them
    (4 x u64, top two bits cleared, reject >= r). The limbs are used directly as Montgomery residues."""
    r_limbs = [(R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    out = np.zeros((0, 4), np.uint64)
    round_ = 0
    while out.shape[0] < n:
        m = int((n - out.shape[0]) * 1.4) + 16
        c = splitmix64_stream(seed + 0x1000003 * round_, 4 * m).reshape(m, 4)
        c[:, 3] &= np.uint64(0xFFFFFFFFFFFFFFFF >> 2)
        lt = np.zeros(m, bool); eq = np.ones(m, bool)
        for k in (3, 2, 1, 0):
            lt |= eq & (c[:, k] < np.uint64(r_limbs[k]))
            eq &= c[:, k] == np.uint64(r_limbs[k])
        out = np.concatenate([out, c[lt]], 0)
        round_ += 1
    return np.ascontiguousarray(out[:n])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=24, help="log2 of (scalar, point) pairs PER GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kem-log2n", type=int, default=16, help="log2 of the KEM batch per GPU (second half of the BASELINE metric); 0 disables")
    ap.add_argument("--no-precompute", action="store_true", help="skip the one-time SRS window-table build (generic per-window bucket path)")
    ap.add_argument("--cpu-log2n", type=int, default=20, help="log2 of the CPU-baseline sample size")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from keaki_amd.hip import KeakiHip, jac_to_affine_words

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = 1 << args.log2n
    stream = torch.cuda.current_stream(dev)
    hip = KeakiHip(local_rank, stream.cuda_stream)

    # ---- synthetic inputs, resident in HBM -------------------------------------------------------
    # points P_i = k_i * G (valid curve points, generated on the GPU by the batched fixed-base kernel),
    # scalars s_i uniform in [0, r). Rank q uses disjoint seeds so the global instance is one MSM of N*n terms.
    t0 = time.time()
    k_host = random_fr_limbs(n, SEED + 1 + 7919 * rank)
    s_host = random_fr_limbs(n, SEED + 0 + 104729 * (rank + 1))
    d_k = torch.from_numpy(k_host.view(np.int64)).to(dev)
    d_s = torch.from_numpy(s_host.view(np.int64)).to(dev)
    d_gen = torch.zeros(8, dtype=torch.int64, device=dev)
    one_mont = (1 << 256) % 21888242871839275222246405745257275088696311157297823662689037894645226208583
    two_mont = (2 << 256) % 21888242871839275222246405745257275088696311157297823662689037894645226208583
    gen_words = [(one_mont >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] + [(two_mont >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    d_gen.copy_(torch.from_numpy(np.array(gen_words, np.uint64).view(np.int64)))
    d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
    hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr())
    torch.cuda.synchronize(dev)
    gen_s = time.time() - t0
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    table_bytes = 0
    if not args.no_precompute:
        # one-time, like uploading the SRS: KZG bases are fixed, so [2^(window offset)] P_i is tabulated once (W x n x 64 B of HBM)
        t1 = time.time()
        table_bytes = hip.srs_g1_precompute(srs)
        torch.cuda.synchronize(dev)
        gen_s += time.time() - t1
    d_part = torch.zeros(12, dtype=torch.int64, device=dev)
    d_all = torch.zeros((world, 12), dtype=torch.int64, device=dev)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)

    from keaki_amd.dist import sharded_msm, torch_all_gather

    def partial_fn():
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_part.data_ptr())
        return d_part

    def sum_fn(allp):
        hip.g1_sum_dev(allp.data_ptr(), world, d_out.data_ptr())
        return d_out

    gather_fn = torch_all_gather(dist, d_all) if world > 1 else None

    def step():
        res = sharded_msm(partial_fn, gather_fn, sum_fn, world)
        if world == 1:
            d_out.copy_(res)

    hip.set_timing(True)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    bucket_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP-event times of the dominant kernel of this launch, on the stream it ran on. Reading them
        # waits for this step's last event only (the next step cannot start earlier anyway: same stream).
        hip.synchronize()
        bucket_ms.append(hip.last_msm_stats()["bucket_ms"])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    stats = hip.last_msm_stats()

    # ---- second half of the BASELINE metric: batched KEM encapsulations / decapsulations per second -------------------
    # Items are independent: each rank processes its own batch, no data-path collective (src/vec.rs:63-66, :75-78).
    kem = None
    if args.kem_log2n > 0:
        m = 1 << args.kem_log2n
        h_a, h_v, h_r = (random_fr_limbs(m, SEED + 31 + 3 * rank), random_fr_limbs(m, SEED + 37 + 5 * rank), random_fr_limbs(m, SEED + 41 + 7 * rank))
        d_a, d_v, d_r = (torch.from_numpy(x.view(np.int64)).to(dev) for x in (h_a, h_v, h_r))
        g2_words = []
        for c in (10857046999023057135944570762232829481370756359578518086990519993285655852781,
                  11559732032986387107991004021392285783925812861821192530917403151452391805634,
                  8495653923123431417604973247489272438418190587263600148770280649306958101930,
                  4082367875863433681332203403145435568316851327593401208105741076214120093531):
            cm = (c << 256) % 21888242871839275222246405745257275088696311157297823662689037894645226208583
            g2_words += [(cm >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
        d_g2 = torch.from_numpy(np.array(g2_words, np.uint64).view(np.int64)).to(dev)
        d_tau = torch.empty(16, dtype=torch.int64, device=dev)
        hip.g2_mul_batch_dev(d_g2.data_ptr(), 0, d_k.data_ptr(), 1, d_tau.data_ptr())        # [tau]_2 = k_0 * g2
        d_com = d_pts[0].contiguous()                                                         # a commitment: any G1 point
        d_ct = torch.empty((m, 16), dtype=torch.int64, device=dev)
        d_gt = torch.empty((m, 48), dtype=torch.int64, device=dev)
        d_key = torch.empty((m, 32), dtype=torch.uint8, device=dev)
        d_gt2 = torch.empty((m, 48), dtype=torch.int64, device=dev)
        d_key2 = torch.empty((m, 32), dtype=torch.uint8, device=dev)

        def encap():
            hip.encap_batch_dev(d_com.data_ptr(), d_tau.data_ptr(), d_a.data_ptr(), d_v.data_ptr(), d_r.data_ptr(), m,
                                d_ct.data_ptr(), d_gt.data_ptr(), d_key.data_ptr(), 32)

        def decap():
            hip.decap_batch_dev(d_pts.data_ptr(), d_ct.data_ptr(), m, d_gt2.data_ptr(), d_key2.data_ptr(), 32)   # first m SRS points as "proofs"

        rates = []
        for fn in (encap, decap):
            fn(); fn()       # warm-up: the second call to one commitment fills its wider GT table (once)
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                fn()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            if world > 1:
                te = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                el = float(te.item())
            rates.append(2 * m * world / el)
        kem = {"encaps_per_s": rates[0], "decaps_per_s": rates[1], "batch_per_gpu": m, "msg_len": 32,
               "note": "whole-job aggregate over all ranks; items sharded by rank, no collective. encaps: batches >= 2^16 use two fixed-base GT "
                       "exponentiations per item (A = e(C, g2) tabulated once per commitment, reused here across calls) instead of a pairing; "
                       "decaps: one full pairing per item. bench_kem.py also reports the fresh-commitment-per-call rate.",
               "algorithmic_bytes_per_encap": 608, "algorithmic_bytes_per_decap": 576}
        kem_check = (h_a, h_v, h_r, d_com, d_tau, d_ct, d_gt, d_key, d_gt2, d_key2)

    if rank != 0:
        # wait for rank 0 (CPU baseline + JSON line) so the process group is torn down together
        dist.barrier()
        dist.destroy_process_group()
        return

    total_units = n * world * args.steps
    value = total_units / elapsed
    avg_bucket_s = float(np.mean(bucket_ms)) * 1e-3
    achieved_gbs = ALGO_BYTES_PER_SCALAR_MUL * n / avg_bucket_s / 1e9
    windows = (254 + stats["window_bits"] - 1) // stats["window_bits"]
    # integer roofline: one XYZZ mixed add = 8M + 2S = 10 Montgomery products. Measured issue rate of v_mad_u64_u32 / 32-bit
    # VALU on gfx950: 1 wave-instruction per ~4.3 cycles per SIMD (profiles/r01_ubench_int_gfx950.txt). The common path of one
    # mixed addition in k_msm_accumulate_g1_u29 (9 x 29-bit lazy limbs, fq29.cuh) is 2416 instructions in the shipped ISA
    # (1629 v_mad_u64_u32: 6 x 162 + 2 x 126 + one 243-multiply-add dual product R T + (2p - Y) PPP + U2/S2; the rest slides,
    # masks, carries, loads; the two basic blocks of the loop body, rare-path zero test included).
    ISSUES_PER_MIXED_ADD = 2416.0
    modmul_peak = 1024 * 2.4e9 / 4.3 * 64 / ISSUES_PER_MIXED_ADD * 10.0
    modmuls = 10.0 * n * windows / avg_bucket_s
    # HBM traffic of the dominant kernel from PMC counters (separate rocprofv3 --pmc passes, committed under profiles/)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_msm_2p24_hbm_traffic_pmc.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("log2n") == args.log2n:
            for kname, kv in tj["kernels"].items():
                if "k_msm_accumulate" in kname:
                    traffic = kv["fetch_bytes"] + kv["write_bytes"]
    result = {
        "metric": "BN254 G1 scalar-mults/sec on 2^%d-point MSM (per GPU)" % args.log2n,
        "value": value,
        "unit": "scalar-mults/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs, 256-bit Montgomery modular integer (9 x 29-bit lazy limbs in the bucket kernel, 8 x 32 elsewhere)",
        "data": "synthetic: scalars uniform in [0,r) (SplitMix64), points k_i*G generated on device",
        "config": {"workload": "2^%d-point BN254 G1 Pippenger MSM per GPU, SRS + scalars resident in HBM%s" % (
            args.log2n, "" if world == 1 else "; %d chunks, RCCL all-gather of 96-B partial sums + %d EC adds" % (world, world - 1)),
            "points_per_gpu": n, "window_bits": stats["window_bits"], "windows": windows, "setup_s": round(gen_s, 2),
                   "srs_window_tables_bytes": table_bytes},
        "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate_g1_u29", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": ALGO_BYTES_PER_SCALAR_MUL * n,
                     "kernel_ms": avg_bucket_s * 1e3, "msm_total_ms": stats["total_ms"]},
        "kem": kem,
        "alu": {"bound": "integer issue (v_mad_u64_u32)", "achieved": modmuls / 1e9, "peak": modmul_peak / 1e9, "unit": "G modmul/s",
                "frac": modmuls / modmul_peak, "issues_per_mixed_add": ISSUES_PER_MIXED_ADD},
    }

    # ---- correctness of what was timed + CPU baseline (oracle = checker only) ------------------------
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as oc
        ns = min(n, 1 << args.cpu_log2n)
        pts_sample = d_pts[:ns].cpu().numpy().view(np.uint64)
        t0 = time.perf_counter()
        ref = oc.msm_g1(pts_sample, s_host[:ns], threads=1)
        cpu_s = time.perf_counter() - t0
        hip.set_timing(False)
        d_chk = torch.zeros(12, dtype=torch.int64, device=dev)
        hip.msm_g1_dev(hip.srs_g1_wrap_dev(d_pts.data_ptr(), ns), d_s.data_ptr(), ns, d_chk.data_ptr())
        torch.cuda.synchronize(dev)
        got = jac_to_affine_words(d_chk.cpu().numpy().view(np.uint64))
        ncores = os.cpu_count() or 1
        nall = min(n, 1 << min(args.log2n, args.cpu_log2n + 2))
        t0 = time.perf_counter()
        oc.msm_g1(d_pts[:nall].cpu().numpy().view(np.uint64), s_host[:nall], threads=ncores)
        cpu_all_s = time.perf_counter() - t0
        # O(n) check of the full-size result of rank 0's chunk: MSM == (sum s_i k_i) * G
        if world == 1:
            dot = oc.fr_dot(s_host, k_host)           # Montgomery in -> s*k*R^-1 ... handled below
            # fr_dot multiplies Montgomery residues: mont(s)*mont(k) -> mont(s*k); sum stays Montgomery
            g1, _ = oc.generators()
            exp_full = oc.g1_mul_batch(g1, dot.reshape(1, 4))[0]
            full = jac_to_affine_words(d_out.cpu().numpy().view(np.uint64))
            result["config"]["full_size_check"] = "MSM == (sum s_i k_i) G: %s" % bool(np.array_equal(full, exp_full))
        result["cpu_baseline"] = {
            "value": ns / cpu_s, "unit": "scalar-mults/s", "cores": 1, "kind": "port",
            "sample": "first 2^%d (scalar, point) pairs of the workload, CPU restatement of ark-ec msm_bigint_wnaf (not arkworks itself); "
                      "GPU result on the same sample bit-exact: %s" % (int(np.log2(ns)), bool(np.array_equal(got, ref))),
            "all_cores": {"value": nall / cpu_all_s, "cores": ncores, "sample": "first 2^%d pairs, windows spread over threads" % int(np.log2(nall))},
        }
        if kem is not None:
            h_a, h_v, h_r, d_com, d_tau, d_ct, d_gt, d_key, d_gt2, d_key2 = kem_check
            mc = 32
            com_h, tau_h = d_com.cpu().numpy().view(np.uint64), d_tau.cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            ect, egt, ekey = oc.encap_batch(com_h, tau_h, h_a[:mc], h_v[:mc], h_r[:mc], 32, threads=1)
            ce = time.perf_counter() - t0
            t0 = time.perf_counter()
            dgt, dkey = oc.decap_batch(d_pts[:mc].cpu().numpy().view(np.uint64), ect, 32, threads=1)
            cd = time.perf_counter() - t0
            ok = (np.array_equal(d_ct[:mc].cpu().numpy().view(np.uint64), ect) and np.array_equal(d_key[:mc].cpu().numpy(), ekey)
                  and np.array_equal(d_gt[:mc].cpu().numpy().view(np.uint8).reshape(mc, 384), egt)
                  and np.array_equal(d_gt2[:mc].cpu().numpy().view(np.uint8).reshape(mc, 384), dgt) and np.array_equal(d_key2[:mc].cpu().numpy(), dkey))
            kem["cpu_baseline"] = {"encaps_per_s": mc / ce, "decaps_per_s": mc / cd, "cores": 1, "kind": "port",
                                   "sample": "first %d items, CPU restatement of src/kem.rs:13-72; GPU ct/GT/key bytes bit-exact: %s" % (mc, bool(ok))}
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
