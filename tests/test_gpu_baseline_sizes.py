"""GPU parity at the sizes BASELINE.json names (VERDICT r01 'next' item 1): config 2 (2^20-point G1 MSM bit-exact against the CPU
restatement of ark-ec's Pippenger, with and without window tables), config 3's kernel (2^16 all-distinct pairings / decapsulations,
64 sampled items against the oracle), config 4's per-GPU shape (one 2^23 chunk of the seeded 2^26 instance, O(n) identity)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


def _gen_points_dev(hip, torch, dev, k_host):
    from bench import mont_words
    d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
    d_k = torch.from_numpy(k_host.view(np.int64)).to(dev)
    d_pts = torch.empty((k_host.shape[0], 8), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), k_host.shape[0], d_pts.data_ptr())
    hip.synchronize()
    return d_pts


def test_config2_msm_2p20_bit_exact_vs_oracle(oc, hip):
    """BASELINE config 2: 2^20-point BN254 G1 Pippenger MSM on one MI355X, bit-exact vs the CPU path (src/kzg.rs:98)"""
    from bench import random_fr_limbs, SEED
    from keaki_amd.hip import jac_to_affine_words
    torch, dev = _torch_dev()
    n = 1 << 20
    k = random_fr_limbs(n, SEED + 1)
    s = random_fr_limbs(n, SEED + 104729)
    d_pts = _gen_points_dev(hip, torch, dev, k)
    pts = d_pts.cpu().numpy().view(np.uint64)
    exp = oc.msm_g1(pts, s, threads=os.cpu_count() or 1)                  # arkworks' signed-digit Pippenger restated, all host cores
    d_s = torch.from_numpy(s.view(np.int64)).to(dev)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())          # generic per-window path
        hip.synchronize()
        c_generic = hip.last_msm_stats()["window_bits"]
        assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), exp)
        assert hip.srs_g1_precompute(srs) > 0
        d_out.zero_()
        torch.cuda.synchronize(dev)
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())          # shared-bucket path over the window tables
        hip.synchronize()
        assert hip.last_msm_stats()["window_bits"] >= c_generic        # one shared bucket set affords wider windows
        assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), exp)
        # the host-pointer entry on the same handle
        assert np.array_equal(jac_to_affine_words(hip.msm_g1(srs, s)), exp)
    finally:
        srs.free()


def test_config3_pairings_2p16_all_distinct_sampled_vs_oracle(oc, hip):
    """BASELINE config 3's kernel at its size: 2^16 pairings with all-distinct P_i = a_i G, Q_i = b_i G2 (k_pairing_batch), as
    pairing_batch and as decap_batch (src/kem.rs:55-72); 64 sampled items against the oracle, every output row distinct."""
    from bench import random_fr_limbs
    n = 1 << 16
    g1, g2 = oc.generators()
    a, b = random_fr_limbs(n, 0xC0FFEE), random_fr_limbs(n, 0xBEEF)
    P = hip.g1_mul_batch(g1, a)
    Q = hip.g2_mul_batch(g2, b)
    assert np.unique(P, axis=0).shape[0] == n and np.unique(Q, axis=0).shape[0] == n
    gt = hip.pairing_batch(P, Q)
    dgt, dkey = hip.decap_batch(P, Q, 32)
    assert np.array_equal(gt, dgt)
    assert np.unique(gt.view(np.uint64), axis=0).shape[0] == n              # no two lanes produced the same value / overwrote a slot
    idx = np.concatenate([[0, 1, 31, 32, 33, 63, 64, n - 1, n - 2, n - 33], np.random.default_rng(3).integers(0, n, 54)])
    egt = oc.pairing_batch(P[idx], Q[idx], threads=os.cpu_count() or 1)
    assert np.array_equal(gt[idx], egt)
    _, ekey = oc.decap_batch(P[idx], Q[idx], 32, threads=os.cpu_count() or 1)
    assert np.array_equal(dkey[idx], ekey)


def test_config3_encap_2p16_all_distinct_sampled_vs_oracle(oc, hip):
    """config 3 as `encapsulate`: 2^16 items with all-distinct (point, value, r); 64 sampled against the oracle (ct, GT bytes, key)"""
    from bench import random_fr_limbs
    n = 1 << 16
    g1, g2 = oc.generators()
    com = hip.g1_mul_batch(g1, random_fr_limbs(1, 77))[0]
    tau_g2 = hip.g2_mul_batch(g2, random_fr_limbs(1, 78))[0]
    A, V, R = random_fr_limbs(n, 0x1001), random_fr_limbs(n, 0x1002), random_fr_limbs(n, 0x1003)
    ct, gt, key = hip.encap_batch(com, tau_g2, A, V, R, 32)
    idx = np.concatenate([[0, 1, 63, 64, n - 1], np.random.default_rng(4).integers(0, n, 59)])
    ect, egt, ekey = oc.encap_batch(com, tau_g2, A[idx], V[idx], R[idx], 32, threads=os.cpu_count() or 1)
    assert np.array_equal(ct[idx], ect) and np.array_equal(gt[idx], egt) and np.array_equal(key[idx], ekey)
    assert np.unique(key, axis=0).shape[0] == n


@pytest.mark.parametrize("q", [5])
def test_config4_chunk_2p23_of_2p26_identity(oc, hip, q):
    """BASELINE config 4's per-GPU shape: the q-th 2^23 chunk of the seeded 2^26 instance (bench.py's seeds for rank q of 8), with
    window tables (6 GiB for the chunk), checked by the O(n) identity MSM(s, k_i G) == (sum s_i k_i) G; and the generic path on the same chunk."""
    from bench import random_fr_limbs, SEED
    from keaki_amd.hip import jac_to_affine_words
    torch, dev = _torch_dev()
    n = 1 << 23
    k = random_fr_limbs(n, SEED + 1 + 7919 * q + 15485863)                 # Instance(tag=1) of rank q: the strong block of bench.py --gpus 8
    s = random_fr_limbs(n, SEED + 104729 * (q + 1) + 15485863)
    d_pts = _gen_points_dev(hip, torch, dev, k)
    d_s = torch.from_numpy(s.view(np.int64)).to(dev)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    g1, _ = oc.generators()
    exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k).reshape(1, 4))[0]
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
        hip.synchronize()
        assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), exp)
        assert hip.srs_g1_precompute(srs) >= 10 * n * 64
        d_out.zero_()
        torch.cuda.synchronize(dev)
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
        hip.synchronize()
        assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), exp)
    finally:
        srs.free()
        del d_pts, d_s
        torch.cuda.empty_cache()


def test_headline_msm_2p24_with_tables(oc, hip):
    """The headline configuration as a test (bench.py's own checks): 2^24-point G1 MSM over SRS window tables -- the O(n) identity
    MSM(s, k_i G) == (sum s_i k_i) G at full size, the generic path on the same input, and the first 2^20 pairs bit-exact against the
    CPU restatement of ark-ec's Pippenger (src/kzg.rs:98)."""
    from bench import random_fr_limbs, SEED
    from keaki_amd.hip import jac_to_affine_words
    torch, dev = _torch_dev()
    n = 1 << 24
    k = random_fr_limbs(n, SEED + 1)
    s = random_fr_limbs(n, SEED + 104729)
    d_pts = _gen_points_dev(hip, torch, dev, k)
    d_s = torch.from_numpy(s.view(np.int64)).to(dev)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    g1, _ = oc.generators()
    exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k).reshape(1, 4))[0]
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    sub = hip.srs_g1_wrap_dev(d_pts.data_ptr(), 1 << 20)
    try:
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())          # generic per-window path
        hip.synchronize()
        assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), exp)
        assert hip.srs_g1_precompute(srs) >= 10 * n * 64
        d_out.zero_()
        torch.cuda.synchronize(dev)
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())          # the headline path: one shared bucket set over the window tables
        hip.synchronize()
        assert hip.last_msm_stats()["window_bits"] == 22
        assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), exp)
        ns = 1 << 20
        ref = oc.msm_g1(d_pts[:ns].cpu().numpy().view(np.uint64), s[:ns], threads=os.cpu_count() or 1)
        d_out.zero_()
        torch.cuda.synchronize(dev)
        hip.msm_g1_dev(sub, d_s.data_ptr(), ns, d_out.data_ptr())
        hip.synchronize()
        assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), ref)
    finally:
        sub.free()
        srs.free()
        del d_pts, d_s
        torch.cuda.empty_cache()


def test_g2_msm_2p20_identity_with_and_without_tables(oc, hip):
    """G2 MSM at the size DESIGN quotes timings for: 2^20 points k_i G2, by the O(n) identity MSM == (sum s_i k_i) G2, on the generic path
    and over the window tables of the G2 basis."""
    from bench import random_fr_limbs
    from keaki_amd.hip import jac_to_affine_words
    n = 1 << 20
    _, g2 = oc.generators()
    k = random_fr_limbs(n, 0x62A1)
    s = random_fr_limbs(n, 0x62A2)
    s[3] = 0
    pts = hip.g2_mul_batch(g2, k)
    exp = oc.g2_mul_batch(g2, oc.fr_dot(s, k).reshape(1, 4))[0]
    srs = hip.srs_g2_upload(pts)
    try:
        assert np.array_equal(jac_to_affine_words(hip.msm_g2(srs, s)), exp)
        assert hip.srs_g2_precompute(srs) >= n * 128
        assert np.array_equal(jac_to_affine_words(hip.msm_g2(srs, s)), exp)
    finally:
        srs.free()


def test_config1_degree_128_commit_open_verify_through_the_mirror(oc, py):
    """BASELINE config 1 to the letter: KZGSetup::setup(secret, 129) -- `max_d` is a COUNT of powers, so a degree-128 polynomial needs
    129 (src/kzg.rs:55-70) --, commit / open / verify of a degree-128 polynomial through the host mirror of keaki's API, every value
    against the big-integer oracle's restatement of src/kzg.rs:89-151; 130 coefficients are PolynomialTooLarge(130, 129)."""
    from keaki_amd import keaki as K
    rng = K.Rng(128)
    secret = rng.fr_rand()
    setup = K.KZGSetup.setup(secret, 129)
    coeffs_m = np.stack([rng.fr_rand() for _ in range(129)])
    point = rng.fr_rand()
    canon = lambda a: oc.limbs_to_ints(oc.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    tau, z, coeffs = canon(secret)[0], canon(point)[0], canon(coeffs_m)
    g1p, tau_g2 = py.kzg_setup(tau, 129)
    assert oc.g1_to_ints(setup.g1_pow()) == g1p and oc.g2_to_ints(setup.tau_g2())[0] == tau_g2
    com, proof, value = K.commit(setup, coeffs_m), K.open(setup, coeffs_m, point), K.poly_evaluate(coeffs_m, point)
    assert oc.g1_to_ints(com)[0] == py.kzg_commit(g1p, coeffs)
    assert oc.g1_to_ints(proof)[0] == py.kzg_open(g1p, coeffs, z)
    assert canon(value)[0] == py.poly_eval(coeffs, z)
    assert K.verify(setup, com, point, value, proof)
    assert py.kzg_verify(tau_g2, oc.g1_to_ints(com)[0], z, canon(value)[0], oc.g1_to_ints(proof)[0])
    assert not K.verify(setup, com, point, rng.fr_rand(), proof)
    with pytest.raises(K.KZGError) as e:
        K.commit(setup, np.vstack([coeffs_m, rng.fr_rand()]))
    assert e.value == K.KZGError(130, 129)


@pytest.mark.parametrize("bits,log2n", [(11, 22), (20, 21), (34, 20)])
def test_msm_skewed_digits_exercise_the_sort_slow_paths(oc, hip, bits, log2n):
    """Scalars below 2^bits put every pair into a few bins of the bucket sort (round 4's two-pass sort, csrc/msm.hip.h):
      11 bits: 2,048 buckets of 2,048 pairs each -- cells of hundreds of entries, finished by the groups' tail loops (and chunk after chunk of one bin);
      20 bits: one pair per scalar, cells of ~3 entries -- chunks with more cells than the gather shape holds take the plain walk;
      34 bits: two windows, the second one sparse.
    With and without window tables, by the O(n) identity MSM(s, k_i G) == (sum s_i k_i) G; must not take longer than a few ordinary MSMs."""
    import time
    from bench import random_fr_limbs
    from keaki_amd.hip import jac_to_affine_words
    torch, dev = _torch_dev()
    n = 1 << log2n
    k = random_fr_limbs(n, 0x5CE0 + bits)
    rng = np.random.default_rng(bits)
    vals = rng.integers(0, 1 << bits, n, dtype=np.uint64)
    vals[::7] = 0                                                          # zero scalars in between
    canon = np.zeros((n, 4), np.uint64); canon[:, 0] = vals
    s = oc.fr_to_mont(canon)
    d_pts = _gen_points_dev(hip, torch, dev, k)
    d_s = torch.from_numpy(s.view(np.int64)).to(dev)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    g1, _ = oc.generators()
    exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k).reshape(1, 4))[0]
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        for tables in (False, True):
            if tables:
                assert hip.srs_g1_precompute(srs) > 0
            d_out.zero_()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())
            hip.synchronize()
            dt = time.perf_counter() - t0
            assert np.array_equal(jac_to_affine_words(d_out.cpu().numpy().view(np.uint64)), exp), (bits, tables)
            assert dt < 0.5, "skewed scalars took %.3f s (tables: %s)" % (dt, tables)
    finally:
        srs.free()
        del d_pts, d_s
        torch.cuda.empty_cache()
