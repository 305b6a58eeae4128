"""The code paths that need TWO distinct GPUs (the driver's 8-GPU node; the one-GPU test boxes skip them with that reason):
  * the in-process device group on devices [0, 1]: commit, open, open_fk and the KEM loops -- hipMemcpyPeerAsync between two devices with
    peer access enabled by keaki_hip_group_create, event-ordered on the members' streams (on one GPU every member is device 0 and only the
    same-device branch of csrc/group.hip ever runs);
  * RCCL with more than one rank: bench.py --gpus 2 --backend nccl and laconic_ot.py --gpus 2 --backend nccl through the driver's launch
    line (torch.distributed.run), i.e. ncclAllGather / all-to-all over xGMI between two processes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ndev():
    import torch
    return torch.cuda.device_count()        # does not initialise the GPU on this image


needs_two = pytest.mark.skipif(_ndev() < 2, reason="needs two GPUs (hipGetDeviceCount() = %d): runs on the multi-GPU node" % _ndev())


def _json_line(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in output:\n" + text[-2000:])


@pytest.mark.parametrize("devices", [pytest.param([0, 1], marks=needs_two), [0, 0]])
def test_group_commit_open_fk_kem(oc, py, devices):
    """devices [0, 1]: the cross-device branch; [0, 0]: the same test body on the one-GPU box (so that the body itself is known to be right)"""
    from bench import random_fr_limbs
    from keaki_amd.hip import KeakiHip, KeakiHipGroup, jac_to_affine_words
    g1, g2 = oc.generators()
    single = KeakiHip(0)
    grp = KeakiHipGroup(devices)
    try:
        note = grp.peer_note()
        print("peer access:", note or "direct between every pair")
        n = 1 << 18
        pts = single.g1_mul_batch(g1, random_fr_limbs(n, 0x2D01))
        sc = random_fr_limbs(n, 0x2D02)
        srs1 = single.srs_g1_upload(pts)
        gs = grp.srs_g1_upload(pts, precompute=True)
        exp = jac_to_affine_words(single.msm_g1(srs1, sc))
        assert np.array_equal(jac_to_affine_words(grp.msm_g1(gs, sc)), exp)
        assert np.array_equal(exp, oc.msm_g1(pts, sc, threads=os.cpu_count() or 1))
        z = random_fr_limbs(1, 0x2D03)[0]
        p1, v1 = single.kzg_open(srs1, sc[:5000], z)
        p2, v2 = grp.kzg_open(gs, sc[:5000], z)
        assert np.array_equal(jac_to_affine_words(p1), jac_to_affine_words(p2)) and np.array_equal(v1, v2)
        gs.free(); srs1.free()
        # FK23 sharded over the two devices: two all-to-alls and one gather of points BETWEEN the devices per call
        log2d = 14
        d = 1 << log2d
        om = py.fr_root_of_unity(2 * d)
        mont = lambda v: oc.fr_to_mont(oc.ints_to_limbs([v]))[0]
        om_m, omi_m, inv_m = mont(om), mont(pow(om, -1, py.R)), mont(pow(2 * d, -1, py.R))
        coeffs = random_fr_limbs(d, 0x2D04)
        srs_d = single.srs_g1_upload(pts[:d])
        ref = single.open_fk_poly(srs_d, log2d, coeffs, om_m, omi_m, inv_m)
        fk = grp.fk_create(pts[:d], log2d, om_m, omi_m, inv_m)
        for _ in range(2):
            assert np.array_equal(grp.fk_open(fk, log2d, coeffs), ref)
        grp.fk_free(fk); srs_d.free()
        # the KEM loops by item range over the two devices
        m = 5003
        com = pts[0]
        tau_g2 = single.g2_mul_batch(g2, random_fr_limbs(1, 0x2D05))[0]
        A, V, R = random_fr_limbs(m, 0x2D06), random_fr_limbs(m, 0x2D07), random_fr_limbs(m, 0x2D08)
        ct1, gt1, key1 = single.encap_batch(com, tau_g2, A, V, R, 32)
        ct2, gt2, key2 = grp.encap_batch(com, tau_g2, A, V, R, 32)
        assert np.array_equal(ct1, ct2) and np.array_equal(gt1, gt2) and np.array_equal(key1, key2)
        proofs = single.g1_mul_batch(g1, random_fr_limbs(m, 0x2D09))
        d1 = single.decap_batch(proofs, ct1, 32)
        d2 = grp.decap_batch(proofs, ct1, 32)
        assert np.array_equal(d1[0], d2[0]) and np.array_equal(d1[1], d2[1])
    finally:
        grp.close()
        single.close()


def _torchrun(script, *args, nproc=2, timeout=1500):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, script)] + list(args)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    return r.returncode, r.stdout


@needs_two
def test_bench_two_ranks_over_rccl():
    rc, log = _torchrun("bench.py", "--gpus", "2", "--backend", "nccl", "--log2n", "18", "--steps", "3", "--warmup", "1", "--kem-log2n", "12",
                        "--cpu-log2n", "14", "--fk-log2d", "0", "--laconic-log2n", "0", "--strong-log2n", "19")
    assert rc == 0, log[-3000:]
    j = _json_line(log)
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and j["config"]["backend"] == "nccl"
    assert j["checks"]["full_size_check"] is True and j["checks"]["sample_bit_exact"] is True
    assert j["config"]["exchange_ms"] is not None and j["config"]["exchange_ms"] > 0


@needs_two
def test_laconic_ot_two_ranks_over_rccl():
    rc, log = _torchrun("laconic_ot.py", "--gpus", "2", "--backend", "nccl", "--log2n", "15", "--check-single")
    assert rc == 0, log[-3000:]
    j = _json_line(log)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["all_messages_recovered"] is True and j["sharded_equals_single_process"] is True
    assert j["fk_sharded"] is True


@needs_two
def test_bench_self_launched_two_ranks_over_rccl():
    """the driver's line shape at N = 2, typed without a launcher: `python3 bench.py --gpus 2 ...` starts its own ranks (keaki_amd/launch.py),
    RCCL between two devices; the one line carries the weak value, the strong block, the KEM aggregate and the sharded Laconic OT flow"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2n", "20", "--steps", "3", "--warmup", "1", "--kem-log2n", "12",
                        "--cpu-log2n", "14", "--fk-log2d", "0", "--laconic-log2n", "14", "--strong-log2n", "21"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and j["config"]["backend"] == "nccl" and j["config"]["exchange_ms"] > 0
    assert j["strong"]["ranks_seen"] == 2 and j["strong"]["full_size_check"] is True
    assert j["laconic"]["n_gpus"] == 2 and j["laconic"]["backend"] == "nccl" and j["laconic"]["fk_sharded"] is True
    assert j["laconic"]["all_messages_recovered"] is True and all(j["checks"].values())
