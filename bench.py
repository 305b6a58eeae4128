#!/usr/bin/env python3
"""bench.py -- headline benchmark: BN254 G1 scalar-mults/s on a 2^24-point Pippenger MSM per MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          # N > 1 typed like this: starts its N ranks itself, as a child
                                                           # torch.distributed.run job (keaki_amd/launch.py), and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the same through the launcher directly (already inside it: no child)

A "step" = one full MSM (kzg::commit's msm_unchecked, reference src/kzg.rs:98) over a batch of synthetic scalars with the SRS
and the scalars already resident in HBM. With N > 1 every rank holds its own contiguous chunk of 2^LOG2N (scalar, point) pairs
(weak scaling), runs a complete Pippenger on it, and the 96-byte partial sums are exchanged with one RCCL all-gather followed by
N-1 EC adds (keaki_amd/dist.py::ShardedMsm). Consecutive steps use DIFFERENT scalar vectors (two resident sets, alternating), and
the result of the last step is checked at full size, so a step that read stale data would be caught.

Everything -- the MSM kernels, torch's copies and the RCCL collective -- is enqueued on ONE HIP stream (a torch.cuda.Stream whose
handle the keaki context is created on), so the K timed steps are ordered without host synchronisation; the HIP-event time of
the dominant kernel (`roofline.kernel_ms`) is read in a separate, untimed pass over the same steps.

Prints ONE JSON line (rank 0). `roofline` prices the dominant kernel (bucket accumulation) against HBM with the algorithmic 96 B
per scalar-mult; `alu` prices it against the measured integer-issue rate, which is what actually bounds this path (the shader clock is 1.86-2.35 GHz while these kernels
run, not the 2.4 GHz peak: `frac_of_power_limited_rate` compares VALU instructions per second with the pure product stream); `cpu_baseline`
times the CPU restatement of the arkworks algorithm (oracle/, the checker -- never the product) on a bounded sample. Besides
`value` the line carries `value_no_tables`, `value_incl_scalar_h2d`, a `strong` block (BASELINE config 4: 2^26 points in total)
when N > 1 or --strong is given, the `kem` block (second half of the BASELINE metric; `pairings_per_s` = BASELINE config 3), an `fk`
block (FK23 openings at d = 2^21, the kernel family that dominates config 5; rank 0's GPU) and a `laconic` block (BASELINE config 5: the
three phases of the reference's Laconic OT test at 2^20 bits, sharded over ALL ranks of the job: laconic_ot.run_flow). Exit code 1 if any
parity check fails.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from keaki_amd.launch import self_launch, under_launcher  # noqa: E402  (standard library only: no torch, no HIP call)

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
SEED = 0x6B65616B69  # "keaki"
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_SCALAR_MUL = 96   # 32 B scalar + 64 B affine point, each read once
ALGO_BYTES_PER_PAIRING = 576     # 64 B G1 + 128 B G2 in, 384 B GT out
ALGO_BYTES_PER_ENCAP = 608       # alpha, beta, r in (96 B); 128-B affine ciphertext + 384-B GT out


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


def mont_words(value, mod=P_MOD):
    v = (value << 256) % mod
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


G2_GEN = (10857046999023057135944570762232829481370756359578518086990519993285655852781,
          11559732032986387107991004021392285783925812861821192530917403151452391805634,
          8495653923123431417604973247489272438418190587263600148770280649306958101930,
          4082367875863433681332203403145435568316851327593401208105741076214120093531)


def stream_cycles_from_ubench(waves=3):
    """SIMD-cycles of one 205-instruction product stream (u29_mul, 162 v_mad_u64_u32) and of one plain dependent 32-bit VALU instruction at the
    bucket kernel's occupancy (3 waves per SIMD), from the committed micro-benchmark output (bench_tools/ubench_u29.hip)"""
    path = os.path.join(ROOT, "profiles", "r01_ubench_u29_gfx950.txt")
    mul = simple = None
    try:
        for line in open(path):
            m = re.match(r"u29_mul \(205 instr\)\s+waves/SIMD=%d\s.*?([0-9.]+) SIMD-cycles" % waves, line)
            if m:
                mul = float(m.group(1))
            m = re.match(r"v_and_b32 dependent\s+waves/SIMD=%d\s.*?([0-9.]+) SIMD-cycles" % waves, line)
            if m:
                simple = float(m.group(1))
    except OSError:
        pass
    return mul, simple, os.path.relpath(path, ROOT)


def power_limited_valu_rate(waves):
    """VALU wave-instructions per second and SIMD of the pure product stream (u29_mul: 205 instructions, 162 v_mad_u64_u32) at `waves` waves per
    SIMD, from the committed clock micro-benchmark (bench_tools/ubench_clock.hip -> profiles/r06_ubench_clock.txt): wall clock, so the power-dependent
    shader clock (1.86-2.35 GHz while these kernels run, 2.4 idle) is inside it -- this rate, not a cycle count at the peak clock, is what the arithmetic can reach"""
    try:
        for line in open(os.path.join(ROOT, "profiles", "r06_ubench_clock.txt")):
            m = re.match(r"u29_mul chain \(205 instr, 162 mads\)\s+waves/SIMD=%d\s+launch\s+([0-9.]+) ms" % waves, line)
            if m:
                return waves * 60000 * 205 / (float(m.group(1)) * 1e-3)
    except OSError:
        pass
    return None


def yardstick_valu_rate(label, waves):
    """VALU wave-instructions per second and SIMD of a synthetic stream of profiles/r06_ubench_clock.txt (bench_tools/ubench_clock.hip; wall clock, so the
    power-dependent shader clock is inside it) at `waves` waves per SIMD: `label` = "XYZZ mixed addition" (the bucket kernel's own arithmetic -- 8M + 2S, one
    dual product, the differences and carry passes -- without any memory access) or "pairing mix" (60 % v_mad_u64_u32, k_pairing's share). A line reads
    `<label> (<N> VALU, ...)  waves/SIMD=<w>  launch <ms> ms ...  <ns> ns per wave-unit and SIMD`; several runs -> the fastest (the ceiling)."""
    best = None
    try:
        for line in open(os.path.join(ROOT, "profiles", "r06_ubench_clock.txt")):
            m = re.match(r"%s \((\d+) VALU.*?waves/SIMD=%d\s.*?([0-9.]+) ns per wave-unit and SIMD" % (re.escape(label), waves), line)
            if m:
                rate = float(m.group(1)) / (float(m.group(2)) * 1e-9)
                best = rate if best is None else max(best, rate)
    except OSError:
        pass
    return best


def stamped_profile(name, files):
    """a committed measurement under profiles/ whose source stamp matches this tree, else (None, reason)"""
    from bench_tools.srchash import source_hash
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "profiles/%s absent" % name
    j = json.load(open(path))
    want = source_hash(files)
    if j.get("kernel_source_sha256") != want:
        return None, "profiles/%s was measured on other kernel sources (%s, tree has %s): refused" % (name, j.get("kernel_source_sha256"), want)
    return j, None


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
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (one rank per GPU). gloo: several ranks may share one GPU (how the world-2 path is exercised on a 1-GPU box)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise the process group and run every exchange through it even with ONE rank: executes the RCCL path "
                         "(init, all-gather on the kernels' stream, barrier) on a single-GPU box")
    ap.add_argument("--strong", action="store_true", help="also run BASELINE config 4 (2^--strong-log2n points in TOTAL, split over the ranks) at N = 1")
    ap.add_argument("--strong-log2n", type=int, default=26)
    ap.add_argument("--no-extras", action="store_true", help="skip value_no_tables / value_incl_scalar_h2d / strong / fk / laconic")
    ap.add_argument("--fk-log2d", type=int, default=21, help="log2 of the FK23 domain of the `fk` block (BASELINE config 5: 2^21); 0 disables")
    ap.add_argument("--g2-log2n", type=int, default=20, help="log2 of the points of the `msm_g2` block; 0 disables")
    ap.add_argument("--laconic-log2n", type=int, default=20, help="log2 of the receiver bits of the `laconic` block (BASELINE config 5: 2^20); 0 disables")
    args = ap.parse_args()
    if args.gpus > 1 and not under_launcher():
        # `python3 bench.py --gpus N` typed as it stands (the driver's N = 1 line has that shape): this process -- which has not imported
        # torch and never touches the GPU -- starts the N ranks as a CHILD torch.distributed.run job with the same arguments; rank 0 of the
        # child prints the one JSON line on the inherited stdout; the child's exit code is ours (keaki_amd/launch.py)
        raise SystemExit(self_launch(__file__, sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from keaki_amd.hip import KeakiHip, jac_to_affine_words
    from keaki_amd.dist import Shard, ShardedMsm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world                                  # inside a launcher the launcher's world size is the truth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank (use --backend gloo to share a GPU)" % (world, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_collectives
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = Shard(rank, world, dist if use_dist else None, force_collectives=args.force_collectives)

    # ONE stream for torch, RCCL and keaki: a real (non-default) stream, made current for the whole run
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    hip = KeakiHip(dev_index, stream.cuda_stream)

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)

    class Instance:
        """2^log2n (scalar, point) pairs of this rank, resident in HBM: points P_i = k_i G (valid curve points, generated on the GPU by
        the batched fixed-base kernel), two scalar vectors s_i uniform in [0, r). Rank q uses disjoint seeds so the global instance is
        one MSM of N * n terms."""

        def __init__(self, log2n, precompute, tag=0):
            t0 = time.time()
            self.n = n = 1 << log2n
            self.k_host = random_fr_limbs(n, SEED + 1 + 7919 * rank + 15485863 * tag)
            self.s_host = [random_fr_limbs(n, SEED + 104729 * (rank + 1) + 15485863 * tag), None]
            self.s_host[1] = np.ascontiguousarray(np.roll(self.s_host[0], 1, axis=0)[:, ::-1] >> np.uint64(3))   # a second, different vector (every limb < 2^61: value < 2^253 < r)
            d_k = torch.from_numpy(self.k_host.view(np.int64)).to(dev)
            self.d_s = [torch.from_numpy(s.view(np.int64)).to(dev) for s in self.s_host]
            self.d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
            hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, self.d_pts.data_ptr())
            torch.cuda.synchronize(dev)
            del d_k
            self.sm = ShardedMsm(hip, shard, self.d_pts.data_ptr(), n, dev)
            self.table_bytes = self.sm.precompute() if precompute else 0     # one-time, like uploading the SRS (W x n x 64 B of HBM)
            torch.cuda.synchronize(dev)
            self.setup_s = time.time() - t0
            self.count = 0

        def step(self):
            res = self.sm.run(self.d_s[self.count & 1].data_ptr())
            self.count += 1
            return res

        def timed(self, steps, warmup, per_step=None):
            for _ in range(warmup):
                self.step()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
                if per_step:
                    per_step()
            sync_all()
            return max_over_ranks(time.perf_counter() - t0)

        def last_result_affine(self):
            torch.cuda.synchronize(dev)
            out = self.sm.out if use_dist else self.sm.part
            return jac_to_affine_words(out.cpu().numpy().view(np.uint64))

        def check_last(self, oc):
            """O(n) identity at full size, over ALL ranks: MSM == (sum_q sum_i s_i k_i) G for the scalar vector of the LAST step"""
            dot = oc.fr_dot(self.s_host[(self.count - 1) & 1], self.k_host)      # Montgomery residue of this rank's sum s_i k_i
            dots = shard.all_gather_np(np.ascontiguousarray(dot.reshape(4)))
            tot = sum(int.from_bytes(dots[q].tobytes(), "little") for q in range(world)) % R_MOD
            g1, _ = oc.generators()
            tot_limbs = np.frombuffer(tot.to_bytes(32, "little"), np.uint64).reshape(1, 4)
            exp = oc.g1_mul_batch(g1, tot_limbs)[0]
            return bool(np.array_equal(self.last_result_affine(), exp))

        def close(self):
            self.sm.close()

    inst = Instance(args.log2n, not args.no_precompute)
    n = inst.n

    # ---- the timed region of the contract: W warm-up steps, then exactly K steps between barrier + synchronize -------------------
    # The K timed steps are enqueued back to back: nothing reads an event or synchronises with the host between them (for N > 1 step
    # k + 1 is enqueued while the all-gather of step k runs).
    elapsed = inst.timed(args.steps, args.warmup)
    # HIP-event time of the dominant kernel, on the stream it is launched on, from a SEPARATE untimed pass over the same steps (reading
    # an event is a host round trip, which does not belong in the timed region): one value per launch.
    hip.set_timing(True)
    bucket_ms = []
    for _ in range(max(args.steps, 5)):
        inst.step()
        hip.synchronize()
        bucket_ms.append(hip.last_msm_stats()["bucket_ms"])
    stats = hip.last_msm_stats()
    hip.set_timing(False)
    checks = {}
    # An exception inside an OPTIONAL block (not a parity mismatch: those are `checks`) must not cost the line its headline: the block reports
    # {"error": ...}, the line lists it under `blocks_failed`, everything else goes on. Rank-0-only blocks hold no collective; the `laconic` block
    # runs on every rank, where an exception every rank raises alike (an unsupported call) is survived -- a one-sided one still ends in the
    # launcher's timeout, as any desynchronised collective does.
    block_errors = {}
    g2_check = fk_check = None

    class Guard:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            return self

        def __exit__(self, et, ev, tb):
            if et is None or not issubclass(et, Exception):
                return False
            import traceback
            traceback.print_exception(et, ev, tb, file=sys.stderr)
            block_errors[self.name] = "%s: %s" % (et.__name__, ev)
            return True

    # ---- extras (each outside the contract's timed region) ------------------------------------------------------------------------
    extras = {}
    if not args.no_extras:
        # (1) without the window tables: the generic per-window bucket path on the same pairs
        if not args.no_precompute:
            gen = ShardedMsm(hip, shard, inst.d_pts.data_ptr(), n, dev)
            gen.run(inst.d_s[0].data_ptr()); gen.run(inst.d_s[1].data_ptr())          # the first call of a shape grows the workspaces
            sync_all()
            t0 = time.perf_counter()
            for i in range(4):
                gen.run(inst.d_s[i & 1].data_ptr())
            sync_all()
            el = max_over_ranks(time.perf_counter() - t0)
            extras["value_no_tables"] = n * world * 4 / el
            # both paths on the same scalar vector must give the same point
            r_gen = gen.run(inst.d_s[1].data_ptr()).clone()
            r_tab = inst.sm.run(inst.d_s[1].data_ptr())
            checks["no_tables_equals_tables"] = bool(torch.equal(r_gen, r_tab))
            inst.count = 2                                       # the last table-path step used vector 1
            gen.close()
        # (2) scalars start in HOST memory: keaki_hip_msm_g1 copies them in -- the PCIe-inclusive rate, never `value`. Since round 5 the call
        # uploads in point-range chunks on a copy stream and the kernels of chunk j run under the copy of chunk j + 1 (csrc/api.hip:
        # msm_from_host). Three figures: pinned source (as rounds 1-4 measured it), pageable source (what a Rust Vec / numpy array is), and
        # the pinned source with the chunking switched off (one copy in front of the kernels = the round-4 path).
        import ctypes as C
        h_page = inst.s_host[0]
        h_pinned = torch.from_numpy(h_page.view(np.int64)).pin_memory()
        out_host = np.zeros(12, np.uint64)

        host_calls_ms = {}

        def host_rate(ptr, tag, calls=4):
            def host_call():
                st = hip.lib.keaki_hip_msm_g1(hip.ctx, inst.sm.srs.handle, C.c_void_p(ptr), n, out_host.ctypes.data_as(C.c_void_p))
                if st != 0:
                    raise SystemExit("keaki_hip_msm_g1 failed: %s" % hip.lib.keaki_hip_last_error(hip.ctx).decode())
            host_call(); host_call()                         # two untimed calls: the first one of a kind grows the workspaces
            sync_all()
            per = []
            t0 = time.perf_counter()
            for _ in range(calls):
                t1 = time.perf_counter()
                host_call()                                  # returns when the result is in out_host
                per.append(round((time.perf_counter() - t1) * 1e3, 3))
            sync_all()
            host_calls_ms[tag] = per
            return n * world * calls / max_over_ranks(time.perf_counter() - t0), out_host.copy()
        inst.sm.run(inst.d_s[0].data_ptr())
        torch.cuda.synchronize(dev)
        want = jac_to_affine_words(inst.sm.part.cpu().numpy().view(np.uint64))      # this rank's resident-scalar result for vector 0
        inst.count = 1
        extras["value_incl_scalar_h2d"], o1 = host_rate(h_pinned.data_ptr(), "pinned")
        extras["value_incl_scalar_h2d_pageable"], o2 = host_rate(h_page.ctypes.data, "pageable")
        hip.set_option("msm_pipe_chunks", 0)
        extras["value_incl_scalar_h2d_one_copy_in_front"], o3 = host_rate(h_pinned.data_ptr(), "one_copy_in_front", 2)
        hip.set_option("msm_pipe_chunks", -1)
        checks["host_pointer_equals_resident"] = bool(all(np.array_equal(jac_to_affine_words(o), want) for o in (o1, o2, o3)))
        extras["value_incl_scalar_h2d_note"] = ("host-pointer keaki_hip_msm_g1 (kzg::commit's call, src/kzg.rs:89-101): %d MiB of scalars come from host memory inside every call, "
                                                "uploaded in growing point-range chunks under the kernels of the chunk before (first key: pinned source; _pageable: a plain "
                                                "numpy array; _one_copy_in_front: option msm_pipe_chunks = 0, the path of rounds 1-4)" % (n * 32 >> 20))
        extras["value_incl_scalar_h2d_calls_ms"] = host_calls_ms
        del h_pinned

    # ---- G2 MSM (north_star: "Pippenger variable-base MSM over BN254 G1/G2"; keaki itself only forms [tau]_2 and ciphertexts on G2) -------------
    msm_g2 = None
    if not args.no_extras and args.g2_log2n > 0 and rank == 0:
        with Guard('msm_g2'):
            n2 = 1 << min(args.g2_log2n, args.log2n)
            g2_words = []
            for c in G2_GEN:
                g2_words += mont_words(c)
            d_g2gen = torch.from_numpy(np.array(g2_words, np.uint64).view(np.int64)).to(dev)
            d_k2 = torch.from_numpy(inst.k_host[:n2].view(np.int64).copy()).to(dev)
            d_pts2 = torch.empty((n2, 16), dtype=torch.int64, device=dev)
            torch.cuda.synchronize(dev)
            hip.g2_mul_batch_dev(d_g2gen.data_ptr(), 0, d_k2.data_ptr(), n2, d_pts2.data_ptr())      # Q_i = k_i g2: valid r-torsion points
            hip.synchronize()
            del d_k2
            srs2 = hip.srs_g2_wrap_dev(d_pts2.data_ptr(), n2)
            d_out2 = torch.zeros(24, dtype=torch.int64, device=dev)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def g2_rate(reps=3):
                hip.msm_g2_dev(srs2, inst.d_s[0].data_ptr(), n2, d_out2.data_ptr())
                sync_all_local()
                ev0.record(stream)
                for i in range(reps):
                    hip.msm_g2_dev(srs2, inst.d_s[i & 1].data_ptr(), n2, d_out2.data_ptr())
                ev1.record(stream)
                sync_all_local()
                return ev0.elapsed_time(ev1) / reps
            sync_all_local = lambda: torch.cuda.synchronize(dev)
            ms_gen = g2_rate()
            hip.msm_g2_dev(srs2, inst.d_s[0].data_ptr(), n2, d_out2.data_ptr())
            sync_all_local()
            r_gen = d_out2.cpu().numpy().view(np.uint64).copy()
            t0 = time.perf_counter()
            g2_table_bytes = hip.srs_g2_precompute(srs2)
            g2_setup_s = time.perf_counter() - t0
            ms_tab = g2_rate()
            hip.msm_g2_dev(srs2, inst.d_s[0].data_ptr(), n2, d_out2.data_ptr())
            sync_all_local()
            r_tab = d_out2.cpu().numpy().view(np.uint64).copy()
            t0 = time.perf_counter()
            r_host = hip.msm_g2(srs2, inst.s_host[0][:n2])
            host_ms = (time.perf_counter() - t0) * 1e3
            checks["msm_g2.tables_equals_no_tables_equals_host_pointer"] = bool(np.array_equal(r_gen, r_tab) and np.array_equal(r_host, r_tab))
            ALGO_G2 = 160                                               # 32 B scalar + 128 B affine point
            msm_g2 = {"workload": "2^%d-point BN254 G2 Pippenger MSM, points k_i g2 generated on the device, scalars resident (the first 2^%d of the G1 vectors)" % (int(np.log2(n2)), int(np.log2(n2))),
                      "points": n2, "value": n2 / (ms_tab * 1e-3), "value_no_tables": n2 / (ms_gen * 1e-3), "unit": "scalar-mults/s",
                      "ms_per_msm": ms_tab, "ms_per_msm_no_tables": ms_gen, "host_pointer_call_ms": host_ms, "window_tables_bytes": g2_table_bytes,
                      "window_tables_setup_s": round(g2_setup_s, 3),
                      "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate_g2_u29 (whole MSM timed, stream events)", "algorithmic_bytes": ALGO_G2 * n2,
                                   "achieved": ALGO_G2 * n2 / (ms_tab * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ALGO_G2 * n2 / (ms_tab * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "note": "integer-issue bound: ~5,600 instructions per mixed addition in Fq2 (4,700 v_mad_u64_u32), one wave per SIMD at 256 VGPRs"}}
            g2_check = (r_tab, n2)
            srs2.free()
            del d_pts2
    if 'msm_g2' in block_errors:
        msm_g2 = {"error": block_errors['msm_g2']}
        g2_check = None

    # ---- BASELINE config 4: 2^26 points in TOTAL, split over the ranks (strong scaling) ----------------------------------------------
    strong = None
    if (use_dist or args.strong) and not args.no_extras:
        per = (1 << args.strong_log2n) // world
        lg = per.bit_length() - 1
        if (1 << lg) == per:
            if lg == args.log2n and not args.no_precompute:
                sinst, own = inst, False
            else:
                sinst, own = Instance(lg, not args.no_precompute, tag=1), True
            s_steps = max(2, min(args.steps, 5))
            s_el = sinst.timed(s_steps, 1)
            strong = {"total_points": per * world, "points_per_gpu": per, "value": per * world * s_steps / s_el, "unit": "scalar-mults/s",
                      "ms_per_step": s_el / s_steps * 1e3, "steps": s_steps, "scaling": "strong", "ranks_seen": shard.world,
                      "workload": "BASELINE config 4: 2^%d-point G1 MSM in total, %d chunk(s) of 2^%d, all-gather of 96-B partials + EC adds"
                                  % (args.strong_log2n, world, lg)}
            if not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle as oc
                strong["full_size_check"] = sinst.check_last(oc)
                checks["strong.full_size_check"] = strong["full_size_check"]
            if own:
                sinst.close()
                del sinst

    # time of the exchange step alone (all-gather of the partials + EC adds), N > 1
    exchange_ms = None
    if use_dist:
        inst.sm.combine()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(10):
            inst.sm.combine()
        sync_all()
        exchange_ms = max_over_ranks(time.perf_counter() - t0) / 10 * 1e3

    # ---- second half of the BASELINE metric: batched KEM encapsulations / decapsulations per second -------------------
    # Items are independent: each rank processes its own batch, no data-path collective (src/vec.rs:63-66, :75-78).
    kem = None
    if args.kem_log2n > 0:
        m = min(1 << args.kem_log2n, n)                 # the first m SRS points stand in for the proofs
        h_a, h_v, h_r = (random_fr_limbs(m, SEED + 31 + 3 * rank), random_fr_limbs(m, SEED + 37 + 5 * rank), random_fr_limbs(m, SEED + 41 + 7 * rank))
        d_a, d_v, d_r = (torch.from_numpy(x.view(np.int64)).to(dev) for x in (h_a, h_v, h_r))
        g2_words = []
        for c in G2_GEN:
            g2_words += mont_words(c)
        d_g2 = torch.from_numpy(np.array(g2_words, np.uint64).view(np.int64)).to(dev)
        d_tau = torch.empty(16, dtype=torch.int64, device=dev)
        d_k0 = torch.from_numpy(inst.k_host[:1].view(np.int64).copy()).to(dev)
        hip.g2_mul_batch_dev(d_g2.data_ptr(), 0, d_k0.data_ptr(), 1, d_tau.data_ptr())        # [tau]_2 = k_0 * g2
        d_coms = inst.d_pts[:8].contiguous()                                                  # commitments: any G1 points
        d_ct = torch.empty((m, 16), dtype=torch.int64, device=dev)
        d_gt = torch.empty((m, 48), dtype=torch.int64, device=dev)
        d_key = torch.empty((m, 32), dtype=torch.uint8, device=dev)
        d_gt2 = torch.empty((m, 48), dtype=torch.int64, device=dev)
        d_key2 = torch.empty((m, 32), dtype=torch.uint8, device=dev)

        def encap(ci=0):
            hip.encap_batch_dev(d_coms[ci].data_ptr(), d_tau.data_ptr(), d_a.data_ptr(), d_v.data_ptr(), d_r.data_ptr(), m,
                                d_ct.data_ptr(), d_gt.data_ptr(), d_key.data_ptr(), 32)

        def decap():
            hip.decap_batch_dev(inst.d_pts.data_ptr(), d_ct.data_ptr(), m, d_gt2.data_ptr(), d_key2.data_ptr(), 32)

        def rate(fn, reps, warm):
            for i in range(warm):
                fn(i)
            sync_all()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            for i in range(reps):
                fn(i)
            ev1.record(stream)
            sync_all()
            el = max_over_ranks(time.perf_counter() - t0)
            return reps * m * world / el, ev0.elapsed_time(ev1) / reps

        # steady state: the caller keeps encrypting to ONE commitment (what vec_encrypt / Laconic OT do); the second call fills the wider GT table
        enc_rate, enc_ms = rate(lambda i: encap(0), 3, 2)
        dec_rate, dec_ms = rate(lambda i: decap(), 3, 1)
        # a NEW commitment in every call: the table of A = e(C, g2) is rebuilt each time (one pairing launch + fills)
        fresh_rate, fresh_ms = rate(lambda i: encap(1 + (i % 7)), 4, 0)
        encap(0); encap(0)                      # leave the outputs of commitment 0 in place for the check below
        torch.cuda.synchronize(dev)
        kem = {"encaps_per_s": fresh_rate, "encaps_same_commitment_per_s": enc_rate, "decaps_per_s": dec_rate, "pairings_per_s": dec_rate, "batch_per_gpu": m, "msg_len": 32,
               "pairings_per_s_note": "BASELINE config 3 (\"2^16 BN254 pairings: Miller loop + final exponentiation\"): one full pairing per item = "
                                      "decaps_per_s. encaps_per_s is the rate of ONE batch to a commitment seen for the first time (its table of e(C, g2): one "
                                      "pairing launch + fills, then two fixed-base GT exponentiations per item); encaps_same_commitment_per_s is the steady "
                                      "state of a caller that keeps encrypting to one commitment (vec_encrypt / Laconic OT): no pairing per item at all",
               "note": "whole-job aggregate over all ranks; items sharded by rank, no collective. encaps: every batch uses two fixed-base GT "
                       "exponentiations per item (A = e(C, g2) tabulated once per commitment from ONE launch of the twelve-lane pairing kernel over the "
                       "tabulated multiples of g2, on a side stream beside the ciphertext kernel: rebuilt in every call in encaps_per_s, reused across "
                       "calls in encaps_same_commitment_per_s) instead of a pairing; decaps: one full pairing per item.",
               "roofline_decap": {"bound": "hbm", "kernel": "k_pairing_batch (+ k_blake3_gt_xof)", "algorithmic_bytes": ALGO_BYTES_PER_PAIRING * m,
                                  "call_ms": dec_ms, "achieved": ALGO_BYTES_PER_PAIRING * m / (dec_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ALGO_BYTES_PER_PAIRING * m / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "stream-event time of one decap_batch_dev call on this rank; integer-issue bound (SQ_INSTS_VALU per pairing under profiles/)"},
               "roofline_encap": {"bound": "hbm", "kernel": "k_encap_fixed<Fq2> + k_gt_encap_exp + k_blake3_gt_xof", "algorithmic_bytes": ALGO_BYTES_PER_ENCAP * m,
                                  "call_ms": enc_ms, "achieved": ALGO_BYTES_PER_ENCAP * m / (enc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ALGO_BYTES_PER_ENCAP * m / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
               "algorithmic_bytes_per_encap": ALGO_BYTES_PER_ENCAP, "algorithmic_bytes_per_decap": ALGO_BYTES_PER_PAIRING}
        kem_check = (h_a, h_v, h_r, d_coms[0], d_tau, d_ct, d_gt, d_key, d_gt2, d_key2)
        # integer-issue diagnostic of the throughput pairing kernel (k_pairing: a lane pair per pairing, 32 pairings per wave, 2 waves per SIMD at
        # 255 VGPRs): VALU wave-instructions per wave from the committed counter pass x the waves of this launch / the SIMD-cycles it took
        # = cycles per issued VALU instruction, against the issue rate of ITS mix (v_mad_u64_u32 share from the emitted ISA) at two waves per SIMD
        try:
            from bench_tools.srchash import library_hashes, source_hash, PAIRING_KERNEL_SOURCES
            import ctypes as C
            hip.lib.keaki_hip_version.restype = C.c_char_p
            lib_pair = library_hashes(hip.lib.keaki_hip_version().decode()).get("pairing")
            pj = json.load(open(os.path.join(ROOT, "profiles", "r05_pairing_isa.json")))
            cnt = json.load(open(os.path.join(ROOT, "profiles", "r05_pairing_pmc_sq_insts.json")))
            if pj.get("kernel_source_sha256") != lib_pair or cnt.get("hashes", {}).get("pairing") != lib_pair:
                kem["alu"] = {"note": "profiles/r05_pairing_isa.json / r05_pairing_pmc_sq_insts.json were made on other pairing kernel sources than the loaded library: refused"}
            else:
                valu_per_wave = float(cnt["k_pairing"]["valu_per_wave"])
                waves = m / 32.0
                cyc = dec_ms * 1e-3 * 2.4e9 * 1024.0 / (waves * valu_per_wave)
                rates = {}
                for line in open(os.path.join(ROOT, "profiles", "r01_ubench_u29_gfx950.txt")):
                    mm = re.match(r"(v_mad_u64_u32 dependent|v_and_b32 dependent)\s+waves/SIMD=2\s.*?([0-9.]+) SIMD-cycles", line)
                    if mm:
                        rates[mm.group(1).split()[0]] = float(mm.group(2))
                f_mad = float(pj["mad_fraction_static"])
                model = f_mad * rates["v_mad_u64_u32"] + (1.0 - f_mad) * rates["v_and_b32"]
                kem["alu"] = {"bound": "integer issue", "kernel": "k_pairing (decaps_per_s / pairings_per_s)", "valu_wave_instructions_per_pairing": valu_per_wave / 32.0,
                              "valu_per_wave_of_32_pairings": valu_per_wave, "waves_per_simd": 2, "simd_cycles_per_valu_instruction_measured": cyc,
                              "simd_cycles_per_valu_instruction_at_issue_rate": model, "frac_vs_issue_model_at_2p4ghz": model / cyc, "v_mad_u64_u32_share": f_mad,
                              "frac": (waves * valu_per_wave / (dec_ms * 1e-3) / 1024.0 / yardstick_valu_rate("pairing mix", 2)) if yardstick_valu_rate("pairing mix", 2) else None,
                              "frac_note": "= frac_of_power_limited_rate: against a measured ceiling (the synthetic stream of the kernel's own mix, wall clock). "
                                           "frac_vs_issue_model_at_2p4ghz (rounds 2-5's `frac`: a cycle model at the nominal clock) can exceed one and is kept for continuity only",
                              "issue_cycles_at_two_waves": rates,
                              "valu_wave_instructions_per_s_per_simd": waves * valu_per_wave / (dec_ms * 1e-3) / 1024.0,
                              "power_limited_stream_rate_per_simd": yardstick_valu_rate("pairing mix", 2),
                              "frac_of_power_limited_rate": (waves * valu_per_wave / (dec_ms * 1e-3) / 1024.0 / yardstick_valu_rate("pairing mix", 2)) if yardstick_valu_rate("pairing mix", 2) else None,
                              "power_limited_stream": "a synthetic stream with k_pairing's own mix (270 VALU instructions per unit, 162 of them v_mad_u64_u32 = 60 %) at two waves per "
                                                      "SIMD, wall clock (bench_tools/ubench_clock.hip -> profiles/r06_ubench_clock.txt): a ceiling for this mix -- until round 5 the yardstick "
                                                      "was the pure product stream (76 % multiply-adds), which clocks lower than the kernel and put this fraction above one",
                              "sources": ["profiles/r05_pairing_pmc_sq_insts.json", "profiles/r05_pairing_isa.json", "profiles/r01_ubench_u29_gfx950.txt"],
                              "note": "dec_ms includes the KDF kernel (k_blake3_gt_xof, < 1 %); frac = how close the launch runs to the issue rate of its own "
                                      "instruction mix at the two waves per SIMD its 255 registers allow -- a schedule diagnostic, not a claim that the stream is minimal"}
        except (OSError, ValueError, KeyError) as e:
            kem["alu"] = {"note": "no committed counter / ISA file (%s: %s)" % (type(e).__name__, e)}

    # ---- FK23 batch openings (kzg::open_fk, src/kzg.rs:157-203): the kernel family that dominates Receiver::new of BASELINE config 5 ----
    fk = None
    if not args.no_extras and args.fk_log2d > 0 and (1 << args.fk_log2d) <= n and rank == 0:
        with Guard('fk'):
            import ctypes as C
            lg = args.fk_log2d
            d = 1 << lg
            w2d = pow(5, (R_MOD - 1) >> (lg + 1), R_MOD)                       # ark-poly's group_gen of Radix2EvaluationDomain::new(2d): GENERATOR = 5
            assert pow(w2d, d, R_MOD) == R_MOD - 1
            mont_fr = lambda v: np.frombuffer(((v << 256) % R_MOD).to_bytes(32, "little"), np.uint64).copy()
            om, omi, inv2d = mont_fr(w2d), mont_fr(pow(w2d, -1, R_MOD)), mont_fr(pow(2 * d, -1, R_MOD))
            fsrs = hip.srs_g1_wrap_dev(inst.d_pts.data_ptr(), d)               # the first d points of this rank's SRS
            coeffs = random_fr_limbs(d, SEED + 4242)
            proofs = np.zeros((d, 8), np.uint64)
            t0 = time.perf_counter()
            hip._ck(hip.lib.keaki_hip_srs_g1_precompute_fk(hip.ctx, fsrs.handle, lg, om.ctypes.data_as(C.c_void_p)))   # hat_s: setup, like the window tables
            fk_setup_s = time.perf_counter() - t0
            hip.set_timing(True)
            call_s, st_ms = [], []
            for it in range(3):
                t0 = time.perf_counter()
                proofs = hip.open_fk_poly(fsrs, lg, coeffs, om, omi, inv2d)
                call_s.append(time.perf_counter() - t0)
                st_ms.append(hip.last_fk_stats())
            hip.set_timing(False)
            best = int(np.argmin(call_s))
            stages_ms = st_ms[best]["stages_ms"]
            fk_algo = d * (32 + 64)                                            # d coefficients in, d affine proofs out (checked against the oracle below)
            # integer-issue diagnostic: butterflies = d * log2(d) (two size-d transforms of d/2 * log2 d each) + 2d pointwise + d twist scalar-mults,
            # one scalar-mult ~ 129 doublings + 43..66 additions in the 29-bit ladder
            fk = {"workload": "FK23 openings (kzg::open_fk) of a degree-(2^%d - 1) polynomial at the 2^%d roots of unity, hat_s cached per SRS" % (lg, lg),
                  "n_gpus": 1, "proofs_per_s": d / call_s[best], "call_ms": call_s[best] * 1e3, "call_ms_all": [round(x * 1e3, 2) for x in call_s],
                  "call_note": "keaki_hip_open_fk_poly: coefficients from host memory in (%d MiB), affine proofs to host memory out (%d MiB)" % (d * 32 >> 20, d * 64 >> 20),
                  "device_ms": st_ms[best]["device_ms"], "pointwise_ms": st_ms[best]["pointwise_ms"], "setup_hat_s_s": round(fk_setup_s, 3),
                  "scalar_mults": d * lg + 3 * d,
                  "roofline": {"bound": "hbm", "kernel": "k_g1_fft_stage_map / k_g1_fft_stage4 (the 2 x %d butterfly stages: one butterfly = one 254-bit scalar-mult + add + sub; from span 16 on two stages per radix-4 pass)" % lg,
                               "algorithmic_bytes": fk_algo, "kernel_ms": stages_ms, "kernel_ms_stat": "all butterfly stages of one call, HIP events on the ctx stream",
                               "achieved": fk_algo / (stages_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fk_algo / (stages_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "integer-issue bound like every kernel of the path: d log2 d butterfly scalar-mults of ~129 doublings + 43..66 mixed additions each"},
                  "alu": {"scalar_mults_per_s_in_stages": d * lg / (stages_ms * 1e-3), "simd_cycles_per_butterfly": stages_ms * 1e-3 * 2.4e9 / (d * lg / 64.0 / 1024.0),
                          "note": "SIMD cycles one wave (64 butterflies) spends per butterfly stage step = scalar-mult + add + sub, at 2.4 GHz on 1024 SIMDs. The ladder: "
                                  "129 doublings + 43 mixed additions (one twiddle per wave: sliding windows) or 66 (a twiddle per lane: fixed windows) on an "
                                  "effective-affine window table; valu_per_wave = SQ_INSTS_VALU / SQ_WAVES of the stage kernels from the committed counter "
                                  "pass (DESIGN 4.2b, 7.3)"}}
            # fabric traffic of one call from the committed counter passes (bench_tools/collect_pmc_fk_pairing.sh), when they were made on this build
            fk_traffic, fk_traffic_note = None, None
            try:
                from bench_tools.srchash import library_hashes
                pj = json.load(open(os.path.join(ROOT, "profiles", "r05_fk_pairing_hbm_traffic_pmc.json")))
                hip.lib.keaki_hip_version.restype = C.c_char_p
                if pj.get("hashes", {}).get("fk") != library_hashes(hip.lib.keaki_hip_version().decode()).get("fk"):
                    fk_traffic_note = "profiles/r05_fk_pairing_hbm_traffic_pmc.json was measured on other FK23 kernel sources: refused"
                elif pj.get("fk_one_call", {}).get("log2d") != lg:
                    fk_traffic_note = "committed PMC figure is for another domain size"
                else:
                    # the r05 file holds the raw FETCH_SIZE counter: x 2 = bytes moved (every fabric request is a 128-byte line: profiles/r06_fetch_size_calibration.txt)
                    fk_traffic = 2.0 * pj["fk_one_call"]["fetch_bytes"] + pj["fk_one_call"]["write_bytes"]
                    ipw = pj.get("fk_instructions_per_wave", {})
                    pick = lambda tag: next((round(v["insts_valu"]) for k, v in ipw.items() if tag in k), None)
                    # averages over the launches of each kernel (the first stages' twiddles are short scalars: their ladders are cheaper)
                    fk["alu"]["valu_per_wave"] = {"radix4_pass_per_group_of_four_points": pick("fft_stage4<true"),
                                                  "radix2_one_twiddle_per_wave_per_butterfly": pick("stage_map<true, true, false"),
                                                  "radix2_twiddle_per_lane_per_butterfly": pick("stage_map<true, false, true")}
                    # cycles per issued VALU instruction of the stage kernels: the wave-instructions of one call (counter pass) over the SIMD-cycles of
                    # the stages (events of this run), against the plain / multiply-add issue rates at the stage kernels' occupancy (2 waves per SIMD)
                    sj = json.load(open(os.path.join(ROOT, "profiles", "r05_fk_sq_insts.json")))
                    tot_valu = sj.get("stage_kernels_valu_wave_instructions_one_call") if sj.get("hashes", {}).get("fk") == pj["hashes"]["fk"] and sj.get("log2d") == lg else None
                    if tot_valu:
                        cyc = stages_ms * 1e-3 * 2.4e9 * 1024.0 / float(tot_valu)
                        f_mad = 0.6                                  # share of v_mad_u64_u32 in the 29-bit product streams the ladders are made of (205-instruction product: 162)
                        fk["alu"].update({"valu_wave_instructions_per_s_per_simd": float(tot_valu) / (stages_ms * 1e-3) / 1024.0,
                                          "power_limited_stream_rate_per_simd": power_limited_valu_rate(3),
                                          "frac_of_power_limited_rate": (float(tot_valu) / (stages_ms * 1e-3) / 1024.0 / power_limited_valu_rate(3)) if power_limited_valu_rate(3) else None,
                                          "stage_valu_wave_instructions_per_call": float(tot_valu), "simd_cycles_per_valu_instruction_measured": cyc,
                                          "simd_cycles_per_valu_instruction_at_issue_rate": f_mad * 4.8 + (1 - f_mad) * 4.1, "frac": (f_mad * 4.8 + (1 - f_mad) * 4.1) / cyc,
                                          "frac_note": "SIMD-cycles of the butterfly stages of this run / VALU wave-instructions of the stage kernels in one call (profiles/r05_fk_sq_insts.json), "
                                                       "against the issue rates of profiles/r01_ubench_u29_gfx950.txt at two waves per SIMD (4.8 multiply-add, 4.1 plain)"})
            except (OSError, ValueError, KeyError) as e:
                fk_traffic_note = "no committed PMC figure (%s)" % type(e).__name__
            fk["roofline"]["traffic"] = fk_traffic
            fk["roofline"]["traffic_note"] = fk_traffic_note or ("FETCH_SIZE x 2 + WRITE_SIZE of ONE call (bytes moved over the fabric; counters of profiles/r05_fk_pairing_hbm_traffic_pmc.json, kernels unchanged since), all FK23 kernels: the per-lane window tables of the ladders (1 KB written, "
                                                                 "43..66 x 128 B read per scalar-mult) and the 96-byte points, not the algorithmic 96 B per opening")
            fk_check = (fsrs, coeffs, proofs, om)
    if 'fk' in block_errors:
        fk = {"error": block_errors['fk']}
        fk_check = None

    # ---- Laconic OT (BASELINE config 5): the three phases the reference's test prints (tests/laconic_ot.rs:143-188) at 2^--laconic-log2n
    # bits, on ALL ranks of this job: with N > 1 the FK23 openings of Receiver::new are sharded (ShardedFk: two all-to-alls + one all-gather
    # per call), the commit MSM is split by point range, encapsulations / decapsulations by item range -- laconic_ot.py's flow on this
    # process group (one rank: the un-sharded calls through the host mirror)
    laconic = None
    if not args.no_extras and args.laconic_log2n > 0:
        with Guard('laconic'):
            from keaki_amd import keaki as K
            import laconic_ot
            lshard = Shard(rank, world, dist if world > 1 else None)
            laconic = laconic_ot.run_flow(K, lshard, dev_index, args.laconic_log2n, 32, "sharded", False, args.backend)
            checks["laconic.all_messages_recovered"] = bool(laconic["all_messages_recovered"])
            laconic["workload"] = ("Laconic OT (tests/laconic_ot.rs:126-200), %d receiver bits, 2 x 32-byte messages per bit, %s, through the host mirror"
                                   % (1 << args.laconic_log2n, "ONE GPU" if world == 1 else "sharded over all %d ranks" % world))
            laconic["note"] = ("wall clock (max over ranks per phase) of vec_commit (iFFT + FK23 openings at d = 2^%d + commit MSM) / 2 x vec_encrypt (2^%d "
                               "encapsulations) / vec_decrypt (2^%d pairings), host arrays in and out" % (args.laconic_log2n + 1, args.laconic_log2n + 1, args.laconic_log2n))
    if 'laconic' in block_errors:
        laconic = {"error": block_errors['laconic']}

    # ---- BASELINE config 1 (degree-128 commit + open, and verify / the single KEM calls on the same setup): what ONE call costs --------------
    single = None
    if not args.no_extras and rank == 0:
        with Guard('single_calls'):
            from keaki_amd import keaki as K
            srng = K.Rng(1)
            ss = K.KZGSetup.setup(srng.fr_rand(), 129, dev_index)
            sp = np.stack([srng.fr_rand() for _ in range(129)])
            s_com = K.commit(ss, sp); sz = srng.fr_rand(); s_pr = K.open(ss, sp, sz); sv = K.poly_evaluate(sp, sz)
            ok_true = bool(K.verify(ss, s_com, sz, sv, s_pr)) and not bool(K.verify(ss, s_com, sz, K.fr_add(sv, K.fr(1)), s_pr))
            s_ct, s_key = K.encapsulate(srng, ss, s_com, sz, sv, 32)
            ok_true = ok_true and K.decapsulate(ss, s_pr, s_ct, 32) == s_key
            def ms_of(fn, reps=5):
                for _ in range(4): fn()              # past the table builds of a first call
                t0 = time.perf_counter()
                for _ in range(reps): fn()
                return round((time.perf_counter() - t0) / reps * 1e3, 3)
            single = {"workload": "BASELINE config 1: degree-128 KZG on KZGSetup::setup(secret, 129) through the host mirror, one call at a time (wall clock, host arrays in and out)",
                      "commit_ms": ms_of(lambda: K.commit(ss, sp)), "open_ms": ms_of(lambda: K.open(ss, sp, sz)),
                      "verify_ms": ms_of(lambda: K.verify(ss, s_com, sz, sv, s_pr)),
                      "encapsulate_ms": ms_of(lambda: K.encapsulate(srng, ss, s_com, sz, sv, 32)),
                      "decapsulate_ms": ms_of(lambda: K.decapsulate(ss, s_pr, s_ct, 32)),
                      "note": "a call with few pairings runs each on twelve lanes and two waves (pairing_wide.hip.h): 1.4 ms per pairing instead of 4.9 on a lane pair; "
                              "encapsulate is to a commitment the context has seen (its GT table is there)"}
            checks["single_calls_consistent"] = ok_true
            ss.close()
    if 'single_calls' in block_errors:
        single = {"error": block_errors['single_calls']}

    # ---- full-size correctness of what was timed (every rank takes part: the expected value needs every rank's dot product) --------
    oc = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as oc
        checks["full_size_check"] = inst.check_last(oc)

    if rank != 0:
        # wait for rank 0 (CPU baseline + JSON line) so the process group is torn down together
        dist.barrier()
        dist.destroy_process_group()
        return

    total_units = n * world * args.steps
    value = total_units / elapsed
    # `roofline.achieved` uses the MEAN launch duration (what the contract asks for: bytes per launch / average launch duration); the
    # minimum and the median are reported beside it (the first launches after a context switch run ~10 % long)
    avg_bucket_s = float(np.mean(bucket_ms)) * 1e-3
    achieved_gbs = ALGO_BYTES_PER_SCALAR_MUL * n / avg_bucket_s / 1e9
    windows = (254 + stats["window_bits"] - 1) // stats["window_bits"]
    from bench_tools.srchash import MSM_KERNEL_SOURCES
    # HBM traffic of the dominant kernel from PMC counters (separate rocprofv3 --pmc passes, committed under profiles/, stamped with
    # the kernel sources they were measured on)
    traffic, traffic_note = None, None
    tj, why = stamped_profile("r06_msm_2p24_hbm_traffic_pmc.json", MSM_KERNEL_SOURCES)
    if tj is None:
        traffic_note = why
    elif tj.get("log2n") != args.log2n or bool(tj.get("precompute", True)) != (not args.no_precompute):
        traffic_note = "committed PMC figure is for another configuration"
    else:
        for kname, kv in tj["kernels"].items():
            if "k_msm_accumulate" in kname:
                traffic = kv["fetch_bytes"] + kv["write_bytes"]
    # integer roofline: one XYZZ mixed add = 8M + 2S = 10 Montgomery products; instruction count of the loop body from the shipped
    # ISA (bench_tools/count_isa.py -> profiles/r06_accumulate_isa.json), issue rate from the committed micro-benchmark
    alu = None
    isa, isa_why = stamped_profile("r06_accumulate_isa.json", MSM_KERNEL_SOURCES)
    mul_cyc, simple_cyc, cyc_src = stream_cycles_from_ubench(3)
    if isa is not None and mul_cyc is not None and simple_cyc is not None:
        ipa, mads = float(isa.get("loop_valu", isa["loop_instructions"])), float(isa["loop_v_mad_u64_u32"])
        ipa_src = "static: VALU instructions on the common path of the loop in the emitted ISA"
        # the DYNAMIC count where a counter pass of this build is committed: SQ_INSTS_VALU of one launch / the wave-additions it performs
        sq, _ = stamped_profile("r06_msm_sq_insts.json", MSM_KERNEL_SOURCES)
        if sq is not None and sq.get("log2n") == args.log2n and bool(sq.get("precompute", True)) == (not args.no_precompute) and sq.get("accumulate_valu"):
            ipa = float(sq["accumulate_valu"]) / (n * windows / 64.0)
            ipa_src = "dynamic: SQ_INSTS_VALU of one launch / (n x windows / 64) wave-additions (profiles/r06_msm_sq_insts.json)"
        # model: every 162 multiply-adds are one product stream of 205 instructions at its measured rate, what is left of the loop body runs
        # at the rate of a plain 32-bit VALU instruction
        model_cycles = mads / 162.0 * mul_cyc + max(0.0, ipa - mads / 162.0 * 205.0) * simple_cyc
        wave_adds_per_simd = n * windows / 64.0 / 1024.0
        measured_cycles = avg_bucket_s * 2.4e9 / wave_adds_per_simd
        modmuls = 10.0 * n * windows / avg_bucket_s
        alu = {"bound": "integer issue (v_mad_u64_u32 streams)", "achieved": modmuls / 1e9, "peak": modmuls / 1e9 * measured_cycles / model_cycles, "unit": "G modmul/s",
               "frac": (ipa * (n * windows / 64.0) / avg_bucket_s / 1024.0 / yardstick_valu_rate("XYZZ mixed addition", 3)) if yardstick_valu_rate("XYZZ mixed addition", 3) else None,
               "frac_vs_issue_model_at_2p4ghz": model_cycles / measured_cycles, "simd_cycles_per_mixed_add_measured": measured_cycles, "simd_cycles_per_mixed_add_at_stream_rate": model_cycles,
               "issues_per_mixed_add": ipa, "issues_per_mixed_add_source": ipa_src, "v_mad_u64_u32_per_mixed_add": mads, "product_stream_cycles": mul_cyc, "plain_valu_cycles": simple_cyc,
               "valu_wave_instructions_per_s_per_simd": ipa * (n * windows / 64.0) / avg_bucket_s / 1024.0,
               "power_limited_stream_rate_per_simd": power_limited_valu_rate(3),
               "frac_of_power_limited_rate": (ipa * (n * windows / 64.0) / avg_bucket_s / 1024.0 / power_limited_valu_rate(3)) if power_limited_valu_rate(3) else None,
               "xyzz_addition_rate_per_simd": yardstick_valu_rate("XYZZ mixed addition", 3),
               "frac_of_xyzz_addition_rate": (ipa * (n * windows / 64.0) / avg_bucket_s / 1024.0 / yardstick_valu_rate("XYZZ mixed addition", 3)) if yardstick_valu_rate("XYZZ mixed addition", 3) else None,
               "frac_of_xyzz_addition_rate_note": "= `frac`, THE figure to read: VALU wave-instructions per second and SIMD of this launch against the kernel's own addition (same formulas, same "
                                                  "streams, same three waves per SIMD) run WITHOUT any memory access on this chip (bench_tools/ubench_clock.hip). What is missing from 1 is "
                                                  "shader clock that the gathers' traffic takes out of the package's power budget: HBM 7.4 %, fabric / Infinity Cache 5.3 %, L2 -> L1 1.4 % "
                                                  "of the kernel at round 5's traffic (profiles/r06_bucket_clock_diagnosis.txt). `frac_vs_issue_model_at_2p4ghz` (rounds 2-5's `frac`) and `frac_of_power_limited_rate` compare with "
                                                  "yardsticks that hide this (a nominal 2.4 GHz; the pure product stream, which itself clocks lower than this kernel's mix)",
               "sources": ["profiles/r06_accumulate_isa.json", cyc_src, "profiles/r06_ubench_clock.txt"],
               "note": "a schedule diagnostic (how close the kernel runs to the issue rate of ITS OWN instruction stream at 3 waves per SIMD and "
                       "2.4 GHz), not a claim that the stream is minimal. Round 5 corrected the count: until then the loop body included the exact-zero test that "
                       "hangs off the filter (241 instructions, one product stream, taken 18 times in 2^29 additions), which put frac at 0.94-0.96; the common "
                       "path holds 9.06 product streams, not 10.06"}
    else:
        alu = {"note": isa_why or "no micro-benchmark file"}
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
        "data": "synthetic: scalars uniform in [0,r) (SplitMix64), two vectors alternating between steps; points k_i*G generated on device",
        "config": {"workload": "2^%d-point BN254 G1 Pippenger MSM per GPU, SRS + scalars resident in HBM; `value` runs over the SRS's window tables "
                               "(built once at setup, srs_window_tables_bytes: the bases of a KZG setup never change), the strictly variable-base figure on "
                               "the same input is value_no_tables%s" % (
            args.log2n, "" if world == 1 else "; %d chunks, %s all-gather of 96-B partial sums + %d EC adds" % (world, "RCCL" if args.backend == "nccl" else "gloo", world - 1)),
            "points_per_gpu": n, "window_bits": stats["window_bits"], "windows": windows, "setup_s": round(inst.setup_s, 2),
            "srs_window_tables_bytes": inst.table_bytes, "ranks_seen": shard.world, "backend": args.backend if use_dist else None,
            "exchange_ms": exchange_ms},
        "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate_g1_u29", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_ratio": (traffic / (ALGO_BYTES_PER_SCALAR_MUL * n)) if traffic else None,
                     "traffic_note": traffic_note or ("bytes MOVED over the fabric by one launch = (FETCH_SIZE x 2) + WRITE_SIZE, separate rocprofv3 --pmc passes. Factor 2, calibrated on "
                                                      "this kernel's own access shape (profiles/r06_fetch_size_calibration.txt): FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but every request "
                                                      "moves a whole 128-byte L2 line -- a 64-byte table row costs a line (1.0 request per row, half of it unwanted: structural for "
                                                      "a bucket method, profiles/r06_bucket_clock_diagnosis.txt); 1.15 requests per row in all since the index stream goes through "
                                                      "LDS in 64-byte groups (1.44 until round 5). Rounds 1-5 reported the raw counter (18.8 GB for what was 37.2 GB)"),
                     "algorithmic_bytes": ALGO_BYTES_PER_SCALAR_MUL * n, "kernel_ms": avg_bucket_s * 1e3, "kernel_ms_stat": "mean of %d launches (HIP events, untimed pass)" % len(bucket_ms),
                     "kernel_ms_min": float(np.min(bucket_ms)), "kernel_ms_median": float(np.median(bucket_ms)), "kernel_ms_max": float(np.max(bucket_ms)),
                     "msm_total_ms": stats["total_ms"]},
        "kem": kem,
        "alu": alu,
        "fk": fk,
        "msm_g2": msm_g2,
        "laconic": laconic,
        "single_calls": single,
    }
    result.update(extras)
    if strong is not None:
        result["strong"] = strong
    if "full_size_check" in checks:
        result["config"]["full_size_check"] = "last timed step, all ranks: MSM == (sum s_i k_i) G: %s" % checks["full_size_check"]

    # ---- CPU baseline (oracle = checker only), bounded sample ----------------------------------------------------------------------
    if not args.no_cpu_baseline:
        ns = min(n, 1 << args.cpu_log2n)
        pts_sample = inst.d_pts[:ns].cpu().numpy().view(np.uint64)
        s0 = inst.s_host[0]
        t0 = time.perf_counter()
        ref = oc.msm_g1(pts_sample, s0[:ns], threads=1)
        cpu_s = time.perf_counter() - t0
        d_chk = torch.zeros(12, dtype=torch.int64, device=dev)
        sub = hip.srs_g1_wrap_dev(inst.d_pts.data_ptr(), ns)
        hip.msm_g1_dev(sub, inst.d_s[0].data_ptr(), ns, d_chk.data_ptr())
        torch.cuda.synchronize(dev)
        sub.free()
        got = jac_to_affine_words(d_chk.cpu().numpy().view(np.uint64))
        checks["sample_bit_exact"] = bool(np.array_equal(got, ref))
        ncores = os.cpu_count() or 1
        nall = min(n, 1 << min(args.log2n, args.cpu_log2n + 2))
        t0 = time.perf_counter()
        oc.msm_g1(inst.d_pts[:nall].cpu().numpy().view(np.uint64), s0[:nall], threads=ncores)
        cpu_all_s = time.perf_counter() - t0
        import shutil
        import socket
        have_rust = shutil.which("cargo") and shutil.which("rustc")
        result["cpu_baseline"] = {
            "value": ns / cpu_s, "unit": "scalar-mults/s", "cores": 1, "kind": "port",
            "reference_toolchain": ("cargo + rustc present on %s: rust/README.md's recipe can time arkworks itself" if have_rust
                                    else "absent on %s (no cargo / rustc: keaki + arkworks cannot be built here; profiles/r06_rust_toolchain_probe.txt)") % socket.gethostname(),
            "sample": "first 2^%d (scalar, point) pairs of the workload, CPU restatement of ark-ec msm_bigint_wnaf (not arkworks itself); "
                      "GPU result on the same sample bit-exact: %s" % (int(np.log2(ns)), checks["sample_bit_exact"]),
            "all_cores": {"value": nall / cpu_all_s, "cores": min(ncores, (254 + oc.window_size(nall) - 1) // oc.window_size(nall)), "host_cores": ncores,
                          "sample": "first 2^%d pairs; one thread per window (%d windows), which is how the restatement -- like ark-ec 0.4.2's `parallel` feature, "
                                    "cfg_into_iter over window_starts -- spreads an MSM: `cores` = the threads that are busy, not the %d the host has"
                                    % (int(np.log2(nall)), (254 + oc.window_size(nall) - 1) // oc.window_size(nall), ncores)},
        }
        if kem is not None:
            h_a, h_v, h_r, d_com, d_tau, d_ct, d_gt, d_key, d_gt2, d_key2 = kem_check
            mc = 32
            com_h, tau_h = d_com.cpu().numpy().view(np.uint64), d_tau.cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            ect, egt, ekey = oc.encap_batch(com_h, tau_h, h_a[:mc], h_v[:mc], h_r[:mc], 32, threads=1)
            ce = time.perf_counter() - t0
            t0 = time.perf_counter()
            dgt, dkey = oc.decap_batch(inst.d_pts[:mc].cpu().numpy().view(np.uint64), ect, 32, threads=1)
            cd = time.perf_counter() - t0
            ok = (np.array_equal(d_ct[:mc].cpu().numpy().view(np.uint64), ect) and np.array_equal(d_key[:mc].cpu().numpy(), ekey)
                  and np.array_equal(d_gt[:mc].cpu().numpy().view(np.uint8).reshape(mc, 384), egt)
                  and np.array_equal(d_gt2[:mc].cpu().numpy().view(np.uint8).reshape(mc, 384), dgt) and np.array_equal(d_key2[:mc].cpu().numpy(), dkey))
            checks["kem_bit_exact"] = bool(ok)
            kem["cpu_baseline"] = {"encaps_per_s": mc / ce, "decaps_per_s": mc / cd, "cores": 1, "kind": "port",
                                   "sample": "first %d items, CPU restatement of src/kem.rs:13-72; GPU ct/GT/key bytes bit-exact: %s" % (mc, bool(ok))}
    if msm_g2 is not None and g2_check is not None and oc is not None:
        r_tab, n2 = g2_check
        _, g2g = oc.generators()
        exp2 = oc.g2_mul_batch(g2g, oc.fr_dot(inst.s_host[0][:n2], inst.k_host[:n2]).reshape(1, 4))[0]
        from keaki_amd.hip import jac_to_affine_words as _j2a
        checks["msm_g2.full_size_check"] = bool(np.array_equal(_j2a(r_tab), exp2))
        msm_g2["checked"] = "MSM(s, k_i g2) == (sum s_i k_i) g2 at full size, tables == no tables == host-pointer call: %s" % (
            checks["msm_g2.full_size_check"] and checks["msm_g2.tables_equals_no_tables_equals_host_pointer"])
    if single is not None and "error" not in single and oc is not None:
        # the same five calls on ONE host core through the CPU restatement (the stand-in for keaki's single-threaded arkworks path)
        g1g, g2g = oc.generators()
        cp = oc.g1_mul_batch(g1g, random_fr_limbs(129, 51), threads=os.cpu_count() or 1)
        cs, cz = random_fr_limbs(129, 52), random_fr_limbs(1, 53)[0]
        cq = oc.g2_mul_batch(g2g, random_fr_limbs(2, 54))
        ca, cv, cr = random_fr_limbs(1, 55), random_fr_limbs(1, 56), random_fr_limbs(1, 57)

        def cpu_ms(fn, reps=3):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return round((time.perf_counter() - t0) / reps * 1e3, 3)

        def cpu_open():
            q, _ = oc.fr_quotient(cs, cz)
            oc.msm_g1(cp[:128], q, threads=1)

        def cpu_verify():                                        # two fixed-base mults + two pairings (src/kzg.rs:135-148)
            oc.g1_mul_batch(g1g, cv, threads=1); oc.g2_mul_batch(g2g, ca, threads=1)
            oc.pairing_batch(cp[:2], cq, threads=1)
        c_ct = oc.encap_batch(cp[0], cq[0], ca, cv, cr, 32, threads=1)[0]
        single["cpu_ms"] = {"commit": cpu_ms(lambda: oc.msm_g1(cp, cs, threads=1)), "open": cpu_ms(cpu_open), "verify": cpu_ms(cpu_verify),
                            "encapsulate": cpu_ms(lambda: oc.encap_batch(cp[0], cq[0], ca, cv, cr, 32, threads=1)),
                            "decapsulate": cpu_ms(lambda: oc.decap_batch(cp[:1], c_ct, 32, threads=1)), "cores": 1, "kind": "port",
                            "note": "the same five calls through the CPU restatement (oracle/, the checker) on one host core, same shapes (129 coefficients, 32-byte key)"}
    if fk is not None and fk_check is not None and oc is not None:
        # two proofs against the oracle's per-point opening (its quotient, its MSM over the downloaded points)
        fsrs, coeffs, proofs, om = fk_check
        dd = coeffs.shape[0]
        pts_h = inst.d_pts[:dd].cpu().numpy().view(np.uint64)
        ok = True
        for i in (1, dd // 2 + 3):
            z = np.frombuffer(((pow(w2d, 2 * i, R_MOD) << 256) % R_MOD).to_bytes(32, "little"), np.uint64).copy()
            q, _ = oc.fr_quotient(coeffs, z)
            ok = ok and bool(np.array_equal(proofs[i], oc.msm_g1(pts_h[:dd - 1], q, threads=os.cpu_count() or 1)))
        checks["fk.sample_vs_oracle"] = ok
        fk["checked"] = "proofs 1 and d/2 + 3 equal the oracle's per-point opening (its quotient + its Pippenger MSM): %s" % ok
        fsrs.free()
    result["checks"] = checks
    if block_errors:
        result["blocks_failed"] = sorted(block_errors)          # optional blocks that raised (their entries hold the message); the headline stands
    failed = [k for k, v in checks.items() if not v]
    if failed:
        result["value"] = None                         # a number whose result is wrong is not a measurement
        result["failed_checks"] = failed
    print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
