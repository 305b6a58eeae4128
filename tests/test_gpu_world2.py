"""N > 1 on real hardware with ONE GPU: two ranks share GPU 0 (each its own keaki context and stream), exchange over gloo. The jobs
are launched by the session fixture `multirank_runs` (tests/conftest.py) BEFORE anything in the pytest process touches the GPU; the
tests here only read what the ranks left behind and compare it with the oracle. Covers VERDICT r01 'next' items 1(d) and 4:
  * keaki_hip_msm_g1_dev -> all-gather of 96-byte partials -> keaki_hip_g1_sum_dev composed across two processes, three steps with
    different scalars, one rank with window tables and one without;
  * bench.py --gpus 2 (weak line + strong block + KEM block, full-size checks over both ranks);
  * laconic_ot.py --gpus 2: sharded commit, item-sharded vec_encrypt / vec_decrypt, bytes equal to the single-process calls."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _json_line(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in output:\n" + text[-2000:])


def test_world2_hip_msm_allgather_sum_vs_oracle(multirank_runs, oc):
    run = multirank_runs["msm"]
    assert run["rc"] == [0, 0], run["log"]
    from bench import random_fr_limbs
    from keaki_amd.hip import jac_to_affine_words
    n_total = run["n_total"]
    r = [np.load(os.path.join(run["dir"], "rank%d.npz" % q)) for q in range(2)]
    assert int(r[0]["lo"]) == 0 and int(r[0]["hi"]) == int(r[1]["lo"]) and int(r[1]["hi"]) == n_total
    assert int(r[0]["table_bytes"]) > 0 and int(r[1]["table_bytes"]) == 0
    k_all = random_fr_limbs(n_total, 0xA11CE)
    g1, _ = oc.generators()
    pts = oc.g1_mul_batch(g1, k_all, threads=os.cpu_count() or 1)                 # the whole instance, on the CPU
    for q in range(2):
        assert np.array_equal(r[q]["pts_head"], pts[int(r[q]["lo"]):int(r[q]["lo"]) + 64])
    for step in range(3):
        s_all = random_fr_limbs(n_total, 0xB0B + step)
        exp = oc.msm_g1(pts, s_all, threads=os.cpu_count() or 1)
        for q in range(2):
            assert np.array_equal(jac_to_affine_words(r[q]["results"][step]), exp), (step, q)
            lo, hi = int(r[q]["lo"]), int(r[q]["hi"])
            assert np.array_equal(jac_to_affine_words(r[q]["partials"][step]), oc.msm_g1(pts[lo:hi], s_all[lo:hi], threads=8)), (step, q)


def test_world2_bench_line(multirank_runs):
    run = multirank_runs["bench"]
    assert run["rc"] == 0, run["log"][-3000:]
    j = _json_line(run["log"])
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and j["scaling"] == "weak"
    assert j["value"] and j["value"] > 0
    assert j["checks"]["full_size_check"] is True and j["checks"]["sample_bit_exact"] is True and j["checks"]["kem_bit_exact"] is True
    assert j["config"]["exchange_ms"] is not None
    st = j["strong"]
    assert st["ranks_seen"] == 2 and st["points_per_gpu"] * 2 == st["total_points"] and st["full_size_check"] is True
    assert j["roofline"]["frac"] > 0 and j["cpu_baseline"]["value"] > 0


def test_world2_bench_self_launched_carries_the_whole_metric(multirank_runs):
    """`python3 bench.py --gpus 2 ...` typed WITHOUT torch.distributed.run: the parent starts the ranks as a child job
    before it imports torch; the ONE JSON line holds the weak value, the strong block (config 4's shape), the KEM aggregate and the Laconic OT
    flow (config 5's shape) run SHARDED over both ranks, every check true. Both launch forms must give the same line shape."""
    for name in ("bench_self", "bench"):
        run = multirank_runs[name]
        assert run["rc"] == 0, run["log"][-3000:]
        lines = [ln for ln in run["log"].splitlines() if ln.startswith("{")]
        assert len(lines) == 1, "exactly ONE JSON line (rank 0): %d" % len(lines)
        j = json.loads(lines[0])
        assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and j["config"]["backend"] == "gloo" and j["scaling"] == "weak"
        assert j["value"] and j["value"] > 0 and j["config"]["exchange_ms"] is not None and j["config"]["exchange_ms"] > 0
        st = j["strong"]
        assert st["ranks_seen"] == 2 and st["total_points"] == 1 << 19 and st["points_per_gpu"] == 1 << 18 and st["full_size_check"] is True
        k = j["kem"]
        assert k["encaps_per_s"] > 0 and k["decaps_per_s"] > 0 and k["batch_per_gpu"] == 1 << 10
        lo = j["laconic"]
        assert lo["n_gpus"] == 2 and lo["ranks_seen"] == 2 and lo["backend"] == "gloo" and lo["n_choices"] == 1 << 12
        assert lo["fk_sharded"] is True and lo["fk_exchange_bytes_sent_per_rank"] > 0 and lo["all_messages_recovered"] is True
        assert lo["receiver_new_s"] > 0 and lo["sender_send_s"] > 0 and lo["receiver_receive_s"] > 0 and lo["bits_per_s_end_to_end"] > 0
        assert all(j["checks"].values()), j["checks"]
        assert "absent on" in j["cpu_baseline"]["reference_toolchain"] or "present on" in j["cpu_baseline"]["reference_toolchain"]
    assert "[launch]" in multirank_runs["bench_self"]["log"] and "torch.distributed.run" in multirank_runs["bench_self"]["log"]
    run = multirank_runs["laconic_self"]
    assert run["rc"] == 0, run["log"][-3000:]
    j = _json_line(run["log"])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["all_messages_recovered"] is True and j["sharded_equals_single_process"] is True


def test_world2_laconic_ot(multirank_runs):
    run = multirank_runs["laconic"]
    assert run["rc"] == 0, run["log"][-3000:]
    j = _json_line(run["log"])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2
    assert j["all_messages_recovered"] is True and j["sharded_equals_single_process"] is True
    # the FK23 openings were computed by the sharded pipeline (one all-to-all at setup; two smaller ones + one all-gather per call)
    assert j["n_choices"] == 1 << 15                     # 2^15 receiver bits => domain d = 2^16: 2^14 points per rank and layout
    d = 1 << 16
    assert j["fk_sharded"] is True and j["fk_exchange_bytes_sent_per_rank"] == 2 * d // 4 * 96 + 2 * (d // 4 * 96) + d // 2 * 64


def test_world4_and_world3_laconic_ot(multirank_runs):
    """four ranks: two rank bits in every layout switch of the sharded FK23; three ranks: the openings fall back to the replicated path"""
    for name, world, sharded in (("laconic4", 4, True), ("laconic3", 3, False)):
        run = multirank_runs[name]
        assert run["rc"] == 0, run["log"][-3000:]
        j = _json_line(run["log"])
        assert j["n_gpus"] == world and j["ranks_seen"] == world
        assert j["all_messages_recovered"] is True and j["sharded_equals_single_process"] is True
        assert j["fk_sharded"] is sharded
        assert j["n_choices"] == (1 << 15 if world == 4 else 1 << 8)


def test_world1_under_torchrun(multirank_runs):
    """the launch line of the driver at N = 1 (torch.distributed.run with one rank): bench.py and laconic_ot.py"""
    for name in ("bench1", "laconic1"):
        run = multirank_runs[name]
        assert run["rc"] == 0, run["log"][-3000:]
        j = _json_line(run["log"])
        assert j["n_gpus"] == 1
    assert _json_line(multirank_runs["bench1"]["log"])["checks"]["full_size_check"] is True
    assert _json_line(multirank_runs["laconic1"]["log"])["all_messages_recovered"] is True


def test_rccl_executes_on_hardware_with_one_rank(multirank_runs):
    """RCCL (torch.distributed backend "nccl") had never run under this code: the box has one GPU and RCCL wants one GPU per rank. With
    ONE rank it does run: process-group init, the all-gather of the MSM partials on the stream the kernels are enqueued on followed by
    the EC sum (checked against the oracle), the all-to-all / all-gather of the FK23 exchange buffers, barrier, teardown; and bench.py's
    whole distributed branch (--force-collectives) through the driver's launch line."""
    run = multirank_runs["rccl1"]
    assert run["rc"] == 0, run["log"][-3000:]
    j = _json_line(run["log"])
    assert j["backend"] == "nccl" and j["world"] == 1
    assert j["msm_allgather_sum_ok"] and j["used_collective_output"] and j["all_to_all_ok"] and j["all_gather_ok"] and j["all_gather_np_ok"]
    # libkeaki_hip_rccl.so (no PyTorch in the exchange): MSM + ncclAllGather + EC sum on the context's stream, all-to-all, all-gather
    assert j["shim_msm_ok"] and j["shim_all_to_all_ok"] and j["shim_all_gather_ok"]
    run = multirank_runs["bench_rccl1"]
    assert run["rc"] == 0, run["log"][-3000:]
    j = _json_line(run["log"])
    assert j["n_gpus"] == 1 and j["config"]["backend"] == "nccl" and j["config"]["exchange_ms"] is not None and j["config"]["exchange_ms"] > 0
    assert j["checks"]["full_size_check"] is True and j["checks"]["sample_bit_exact"] is True
    assert j["strong"]["full_size_check"] is True and j["strong"]["ranks_seen"] == 1


def test_eight_ranks_config4_and_config5_at_full_size(multirank_runs):
    """BASELINE configs 4 and 5 at their stated sizes through the driver's 8-rank launch lines, all eight ranks on the one GPU of the box
    (gloo instead of RCCL: the transport is the only thing that differs from an 8-GPU node).
      config 4: 2^26-point G1 MSM, 2^23 points per rank with the window tables of the chunk, all-gather of eight 96-byte partials + EC sum,
                checked at full size by the O(n) identity over all ranks (every rank contributes its sum of s_i k_i);
      config 5: Laconic OT at 2^20 receiver bits: FK23 openings sharded 8 ways at d = 2^21 (three rank bits in every layout switch), commit
                by point range, 2^21 encapsulations / 2^20 decapsulations by item range; rank 0 re-runs the un-sharded calls with the same
                seeds and compares commitment, all 2^21 proofs and its ciphertexts byte for byte (those un-sharded calls are what
                tests/test_gpu_config5.py checks against the oracle)."""
    run = multirank_runs["bench8"]
    assert run["rc"] == 0, run["log"][-3000:]
    j = _json_line(run["log"])
    assert j["n_gpus"] == 8 and j["config"]["ranks_seen"] == 8 and j["scaling"] == "weak" and j["value"] > 0
    assert j["checks"]["full_size_check"] is True and j["checks"]["sample_bit_exact"] is True
    st = j["strong"]
    assert st["total_points"] == 1 << 26 and st["points_per_gpu"] == 1 << 23 and st["ranks_seen"] == 8 and st["full_size_check"] is True
    lo = j["laconic"]                                      # the bench line's own Laconic OT block, sharded over the eight ranks (small here; full size below)
    assert lo["n_gpus"] == 8 and lo["ranks_seen"] == 8 and lo["fk_sharded"] is True and lo["all_messages_recovered"] is True and lo["n_choices"] == 1 << 10
    run = multirank_runs["laconic8"]
    assert run["rc"] == 0, run["log"][-3000:]
    j = _json_line(run["log"])
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["n_choices"] == 1 << 20
    assert j["all_messages_recovered"] is True and j["sharded_equals_single_process"] is True and j["fk_sharded"] is True
    d = 1 << 21
    assert j["fk_exchange_bytes_sent_per_rank"] == 7 * (2 * (d // 64 * 96) + d // 8 * 64) + 7 * (2 * d // 64 * 96)
