"""In-process multi-GPU (keaki_hip_group_*, include/keaki_hip.h) and SRS handles shared between contexts.

The GPU box has ONE device, so a group of N members is N contexts on device 0 (the entry points take a list of ordinals and an
ordinal may repeat): every member still owns its chunk of the SRS, its window tables, its workspaces, its host thread -- the code
path of N GPUs, with the kernels of the members interleaving on one chip. Results must be the bytes of the single-context call and of
the CPU oracle: commit's MSM (src/kzg.rs:98) by SRS range, `open` (src/kzg.rs:104-124), the loops of vec_encrypt / vec_decrypt
(src/vec.rs:63-66, :75-78) by item range."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 1


def mont(oc, ints):
    return oc.fr_to_mont(oc.ints_to_limbs(ints))


def jac_to_aff(j):
    from keaki_amd.hip import jac_to_affine_words
    return jac_to_affine_words(j)


@pytest.fixture(scope="module")
def big(oc, hip):
    """2^20 seeded (point, scalar) pairs, the oracle's MSM of them (all host cores) and the single-context result"""
    from bench import random_fr_limbs, SEED
    n = 1 << 20
    g1, _ = oc.generators()
    pts = hip.g1_mul_batch(g1, random_fr_limbs(n, SEED + 7))
    sc = random_fr_limbs(n, SEED + 8)
    exp = oc.msm_g1(pts, sc, threads=NCPU)
    srs = hip.srs_g1_upload(pts)
    hip.srs_g1_precompute(srs)
    single = hip.msm_g1(srs, sc)
    srs.free()
    assert np.array_equal(jac_to_aff(single), exp)
    return pts, sc, exp, single


@pytest.mark.parametrize("members", [2, 3, 4])
def test_group_commit_2p20_equals_single_context_and_oracle(oc, big, members):
    """N contexts, N chunks of the SRS with their own window tables, N concurrent MSMs, partials through host memory, one sum."""
    from keaki_amd.hip import KeakiHipGroup
    pts, sc, exp, single = big
    n = pts.shape[0]
    g = KeakiHipGroup([0] * members)
    try:
        srs = g.srs_g1_upload(pts, precompute=True)
        got = g.msm_g1(srs, sc)
        assert np.array_equal(got, single), "normalised Jacobian bytes of the group differ from the single context's"
        assert np.array_equal(jac_to_aff(got), exp)
        # every member holds tables for ITS chunk only: together roughly the table of the whole SRS, none of them all of it
        mem = [g.member_memory(i) for i in range(members)]
        assert all(m["tables"] > 0 for m in mem)
        assert max(m["tables"] for m in mem) < 0.75 * sum(m["tables"] for m in mem)
        # polynomials shorter than the SRS (zip-truncation of msm_unchecked): the ranges follow the SRS, trailing members see nothing
        for m in (0, 1, 5, n // members - 1, n // members + 1, n - 1):
            e = oc.msm_g1(pts[:m], sc[:m], threads=NCPU) if m else np.zeros(8, np.uint64)
            assert np.array_equal(jac_to_aff(g.msm_g1(srs, sc[:m])), e), m
        # `open` over the group: quotient on member 0, MSM everywhere; against the oracle's quotient + MSM
        m = 70001
        z = mont(oc, [123456789123456789])[0]
        proof, val = g.kzg_open(srs, sc[:m], z)
        q, v = oc.fr_quotient(sc[:m], z)
        assert np.array_equal(val, v)
        assert np.array_equal(jac_to_aff(proof), oc.msm_g1(pts[:m - 1], q, threads=NCPU))
        # too long: the guard of src/kzg.rs:93-95
        from keaki_amd.hip import KeakiHipError
        with pytest.raises(KeakiHipError) as e:
            g.msm_g1(srs, np.concatenate([sc, sc[:1]]))
        assert e.value.status == -5 and "SRS holds" in e.value.message
        srs.free()
    finally:
        g.close()


def test_group_encap_decap_split_by_item(oc, hip, rand_fr):
    """vec_encrypt / vec_decrypt's loops over a group: item ranges per member, outputs written in place; bytes of the single context
    (both encapsulation paths) and of the oracle on sampled items"""
    from keaki_amd.hip import KeakiHipGroup
    g1, g2 = oc.generators()
    n = 5003                                           # not a multiple of the group size
    tau, c0 = rand_fr(2, 7001)
    tau_g2 = hip.g2_mul_batch(g2, mont(oc, [tau]))[0]
    com = hip.g1_mul_batch(g1, mont(oc, [c0]))[0]
    base = mont(oc, rand_fr(97, 7002))
    rng = np.random.default_rng(3)
    A, V, Rr = (base[rng.integers(0, 97, n)] for _ in range(3))
    ct1, gt1, key1 = hip.encap_batch(com, tau_g2, A, V, Rr, 32)
    proofs = hip.g1_mul_batch(g1, base[rng.integers(0, 97, n)])
    dgt1, dkey1 = hip.decap_batch(proofs, ct1, 32)
    idx = np.array([0, 1, 1667, 1668, 3335, 3336, n - 1])
    ect, egt, ekey = oc.encap_batch(com, tau_g2, A[idx], V[idx], Rr[idx], 32, threads=NCPU)
    assert np.array_equal(ct1[idx], ect) and np.array_equal(gt1[idx], egt) and np.array_equal(key1[idx], ekey)
    for members in (2, 3):
        g = KeakiHipGroup([0] * members)
        try:
            for _ in range(3):                        # third call with one commitment: every member switches to its GT fixed-base tables
                ct, gt, key = g.encap_batch(com, tau_g2, A, V, Rr, 32)
                assert np.array_equal(ct, ct1) and np.array_equal(gt, gt1) and np.array_equal(key, key1)
            dgt, dkey = g.decap_batch(proofs, ct, 32)
            assert np.array_equal(dgt, dgt1) and np.array_equal(dkey, dkey1)
            # fewer items than members
            ct, gt, key = g.encap_batch(com, tau_g2, A[:1], V[:1], Rr[:1], 32)
            assert np.array_equal(ct, ct1[:1]) and np.array_equal(key, key1[:1])
        finally:
            g.close()
    egt2, ekey2 = oc.decap_batch(proofs[idx], ct1[idx], 32, threads=NCPU)
    assert np.array_equal(dgt1[idx], egt2) and np.array_equal(dkey1[idx], ekey2)


@pytest.mark.parametrize("members,log2d", [(2, 12), (4, 12), (4, 16), (3, 10), (2, 1)])
def test_group_open_fk_equals_single_context_and_oracle(oc, py, hip, members, log2d):
    """kzg::open_fk (src/kzg.rs:157-203) over the members of a group: the sharded FK23 pipeline with the exchanges done inside the library
    (device-to-device copies between the members' buffers). Power-of-two groups shard (two rank bits at four members), three members and
    d < N^2 fall back to member 0: always the bytes of keaki_hip_open_fk_poly on one context, and the oracle's per-point openings."""
    from keaki_amd.hip import KeakiHipGroup
    from bench import random_fr_limbs
    d = 1 << log2d
    g1, _ = oc.generators()
    tau = 0x1234567890ABCDEF1234567 % oc.R_MOD
    pw, acc = [], 1
    for _ in range(d):
        pw.append(acc); acc = acc * tau % oc.R_MOD
    srs_pts = hip.g1_mul_batch(g1, mont(oc, pw))
    w2 = py.fr_root_of_unity(2 * d)
    om, omi, inv = mont(oc, [w2])[0], mont(oc, [pow(w2, -1, py.R)])[0], mont(oc, [pow(2 * d, -1, py.R)])[0]
    p = random_fr_limbs(d, 8800 + log2d)
    srs1 = hip.srs_g1_upload(srs_pts)
    single = hip.open_fk_poly(srs1, log2d, p, om, omi, inv)
    srs1.free()
    g = KeakiHipGroup([0] * members)
    try:
        fk = g.fk_create(srs_pts, log2d, om, omi, inv)
        for rep in range(2):                                   # the handle is reused: second polynomial, same SRS transform
            q = p if rep == 0 else random_fr_limbs(d, 8900 + log2d)
            got = g.fk_open(fk, log2d, q)
            if rep == 0:
                assert np.array_equal(got, single)
            w = pow(w2, 2, py.R)
            for i in sorted({0, 1, d // 2, d - 1}):
                qq, _ = oc.fr_quotient(q, mont(oc, [pow(w, i, py.R)])[0])
                assert np.array_equal(got[i], oc.msm_g1(srs_pts[:d - 1], qq, threads=NCPU)), (rep, i)
        g.fk_free(fk)
    finally:
        g.close()


def test_ctx_trim_releases_and_rebuilds(oc, rand_fr):
    from keaki_amd.hip import KeakiHip
    h = KeakiHip(0)
    try:
        g1, g2 = oc.generators()
        n = 300
        pts = h.g1_mul_batch(g1, mont(oc, rand_fr(n, 31)))
        sc = mont(oc, rand_fr(n, 32))
        srs = h.srs_g1_upload(pts)
        h.srs_g1_precompute(srs)
        r1 = h.msm_g1(srs, sc)
        tau_g2 = h.g2_mul_batch(g2, mont(oc, [77]))[0]
        A, V, Rr = (mont(oc, rand_fr(n, 33 + k)) for k in range(3))
        e1 = h.encap_batch(pts[0], tau_g2, A, V, Rr, 32)
        m = h.memory()
        assert m["workspaces"] > 0 and m["gt_tables"] > 0 and m["tables"] > 0
        h.trim()
        m2 = h.memory()
        assert m2["workspaces"] == 0 and m2["gt_tables"] == 0 and m2["tables"] == m["tables"]      # the SRS tables are the caller's
        assert np.array_equal(h.msm_g1(srs, sc), r1)
        e2 = h.encap_batch(pts[0], tau_g2, A, V, Rr, 32)
        assert all(np.array_equal(a, b) for a, b in zip(e1, e2))
        assert h.kzg_verify(pts[0], tau_g2, A[0], A[1], pts[1]) in (True, False)
        srs.free()
    finally:
        h.close()


def test_group_errors(oc):
    from keaki_amd.hip import KeakiHipGroup, KeakiHipError
    with pytest.raises(KeakiHipError) as e:
        KeakiHipGroup([0, 99])
    assert "member 1" in e.value.message and e.value.status == -1
    with pytest.raises(KeakiHipError):
        KeakiHipGroup([])
    # an SRS uploaded through a group of another size is refused, not misread
    g1, _ = oc.generators()
    a, b = KeakiHipGroup([0, 0]), KeakiHipGroup([0, 0, 0])
    try:
        pts = np.repeat(g1[None, :], 100, 0)
        srs = a.srs_g1_upload(pts, precompute=False)
        with pytest.raises(KeakiHipError) as e:
            b.msm_g1(srs, mont(oc, list(range(100))))
        assert e.value.status == -1
        exp = oc.g1_mul_batch(g1, mont(oc, [sum(range(100))]))[0]
        assert np.array_equal(jac_to_aff(a.msm_g1(srs, mont(oc, list(range(100))))), exp)
        srs.free()
    finally:
        a.close(); b.close()


def test_host_mirror_on_a_device_group(oc, py):
    """keaki::Device built from a list of ordinals: kzg::commit / open and vec::vec_encrypt / vec_decrypt spread over the members, the
    rest on member 0 -- same values as the mirror on one context (same seeds) and as the oracle."""
    from keaki_amd import keaki as K
    from bench import random_fr_limbs
    n = 1 << 14
    secret = K.Rng(91).fr_rand()
    dev = K.Device([0, 0, 0])
    assert dev.members == 3
    s3 = K.KZGSetup.setup(secret, n, device=dev)
    s1 = K.KZGSetup.setup(secret, n)
    try:
        p = random_fr_limbs(n - 5, 9100)
        srs_pts = s1.g1_pow()
        com = K.commit(s3, p)
        assert np.array_equal(com, K.commit(s1, p)) and np.array_equal(com, oc.msm_g1(srs_pts, p, threads=NCPU))
        z = K.fr(-24)
        assert np.array_equal(K.open(s3, p, z), K.open(s1, p, z))
        q, v = oc.fr_quotient(p, z)
        assert np.array_equal(K.open(s3, p, z), oc.msm_g1(srs_pts, q, threads=NCPU))
        assert K.verify(s3, com, z, v, K.open(s3, p, z))                     # member 0
        with pytest.raises(K.KZGError) as e:
            K.commit(s3, random_fr_limbs(n + 1, 1))
        assert e.value.args_tuple == (n + 1, n)
        # vec flow: same rng seed on both setups -> same commitment, proofs (FK23 on member 0: the whole SRS goes up on first use), ciphertexts
        m = 4095
        bits = np.random.default_rng(5).integers(0, 2, m)
        choices = np.where(bits[:, None] == 0, K.fr(0)[None, :], K.fr(1)[None, :]).astype(np.uint64)
        ra, rb = K.Rng(17), K.Rng(17)
        com3, proofs3 = K.vec_commit(ra, s3, choices)          # three members: FK23 un-sharded on member 0 (not a power of two)
        com1, proofs1 = K.vec_commit(rb, s1, choices)
        assert np.array_equal(com3, com1) and np.array_equal(proofs3, proofs1)
        # a power-of-two group shards the openings as well (keaki_hip_group_fk_*): same proofs
        dev4 = K.Device([0, 0, 0, 0])
        s4 = K.KZGSetup.setup(secret, n, device=dev4)
        K.precompute_open_fk(s4, m + 1)
        com4, proofs4 = K.vec_commit(K.Rng(17), s4, choices)
        assert np.array_equal(com4, com1) and np.array_equal(proofs4, proofs1)
        s4.close(); dev4.close()
        el = K.domain_elements(m + 1)
        msgs = np.random.default_rng(6).integers(0, 256, size=(m, 32), dtype=np.uint8)
        vals = np.repeat(K.fr(1)[None, :], m, 0)
        g2a, ba = K.vec_encrypt_arrays(ra, s3, com3, el, vals, msgs)          # 4095 items >= GROUP_MIN_ITEMS: split over the three members
        g2b, bb = K.vec_encrypt_arrays(rb, s1, com1, el, vals, msgs)
        assert np.array_equal(g2a, g2b) and np.array_equal(ba, bb)
        out = K.vec_decrypt_arrays(s3, proofs3[:m], g2a, ba)
        assert np.array_equal(out[bits == 1], msgs[bits == 1]) and not np.array_equal(out[bits == 0], msgs[bits == 0])
        assert np.array_equal(out, K.vec_decrypt_arrays(s1, proofs1[:m], g2b, bb))
    finally:
        s3.close(); s1.close(); dev.close()


def test_two_contexts_share_one_table_allocation(oc):
    """An SRS handle is device memory, not context state: two contexts on one device (one per host thread) run MSMs over ONE copy of the
    2^22 points and ONE set of window tables -- keaki_hip_ctx_memory shows the tables on the context that built them and nothing on
    the other; a concurrent precompute from both threads builds them once."""
    from keaki_amd.hip import KeakiHip
    from bench import random_fr_limbs, SEED
    n = 1 << 22
    g1, _ = oc.generators()
    a, b = KeakiHip(0), KeakiHip(0)
    try:
        k = random_fr_limbs(n, SEED + 21)
        pts = a.g1_mul_batch(g1, k)
        srs = a.srs_g1_upload(pts)
        sizes, errs = [], []

        def build(h):
            try:
                sizes.append(h.srs_g1_precompute(srs))
            except Exception as e:           # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=build, args=(h,)) for h in (a, b)]
        for t in ts: t.start()
        for t in ts: t.join()
        assert not errs and len(sizes) == 2 and sizes[0] == sizes[1] > 0
        ma, mb = a.memory(), b.memory()
        assert sorted([ma["tables"], mb["tables"]]) == [0, sizes[0]], (ma, mb)      # ONE table allocation for two contexts
        # both contexts use it, concurrently, on different scalars; O(n) identity: MSM(s, k_i G) = (sum s_i k_i) G
        res = {}

        def run(h, seed):
            s = random_fr_limbs(n, seed)
            res[seed] = (s, h.msm_g1(srs, s), h.last_msm_stats()["window_bits"])
        ts = [threading.Thread(target=run, args=(h, sd)) for h, sd in ((a, 501), (b, 502))]
        for t in ts: t.start()
        for t in ts: t.join()
        for sd in (501, 502):
            s, got, c = res[sd]
            assert c >= 20, "the shared-bucket path over the shared window tables was not taken (c = %d)" % c
            exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k)[None, :], threads=1)[0]
            assert np.array_equal(jac_to_aff(got), exp)
        assert b.memory()["tables"] + a.memory()["tables"] == sizes[0]
        assert a.memory()["workspaces"] > 0 and b.memory()["workspaces"] > 0 and a.memory()["total"] >= a.memory()["workspaces"]
        # a handle from another device would be refused; here: the same device, so a slice made by the other context works too
        view = b.srs_g1_slice(srs, n // 2, 1000)
        s = random_fr_limbs(1000, 77)
        assert np.array_equal(jac_to_aff(b.msm_g1(view, s)), oc.msm_g1(pts[n // 2:n // 2 + 1000], s, threads=8))
        view.free()
        srs.free()
        assert a.memory()["tables"] + b.memory()["tables"] in (0, sizes[0])       # released from the accounting of the context that freed it
    finally:
        a.close(); b.close()


def test_options_are_per_context_and_unknown_names_are_refused(oc, hip):
    from keaki_amd.hip import KeakiHip, KeakiHipError
    h = KeakiHip(0)
    try:
        with pytest.raises(KeakiHipError) as e:
            h.set_option("no_such_switch", 1)
        assert e.value.status == -1 and "unknown option" in e.value.message
        with pytest.raises(KeakiHipError):
            h.set_option("gt_wb_b", 5)
        g1, _ = oc.generators()
        n = 3000
        pts = h.g1_mul_batch(g1, mont(oc, list(range(1, n + 1))))
        sc = mont(oc, [(i * 7919 + 13) % oc.R_MOD for i in range(n)])
        exp = oc.msm_g1(pts, sc, threads=8)
        srs = h.srs_g1_upload(pts)
        for c in (0, 5, 9, 13):                       # forced window widths of the generic path, results unchanged
            h.set_option("msm_c", c)
            assert np.array_equal(jac_to_aff(h.msm_g1(srs, sc)), exp)
            if c:
                assert h.last_msm_stats()["window_bits"] == c
        # the other (session) context is untouched by this context's switches
        srs2 = hip.srs_g1_upload(pts)
        hip.msm_g1(srs2, sc)
        assert hip.last_msm_stats()["window_bits"] != 13
        srs2.free(); srs.free()
        assert "src=msm:" in h.version() and "pairing:" in h.version()
    finally:
        h.close()


def test_group_of_eight_config4_2p26_points(oc):
    """BASELINE config 4 at its size through the in-process path a Rust caller gets (KEAKI_HIP_DEVICES=8): 2^26 points in total, eight members
    with 2^23 points and the chunk's window tables each (eight contexts on the one GPU of the box), partial sums through host memory, one
    EC sum. Checked at full size by the O(n) identity MSM(s, k_i G) = (sum s_i k_i) G; the chunk sizes and tables are the 8-GPU ones."""
    from keaki_amd.hip import KeakiHip, KeakiHipGroup
    from bench import random_fr_limbs, SEED
    n = 1 << 26
    g1, _ = oc.generators()
    h = KeakiHip(0)
    k = random_fr_limbs(n, SEED + 26)
    pts = h.g1_mul_batch(g1, k)                           # 4 GiB of valid points, generated on the device
    h.close()
    s = random_fr_limbs(n, SEED + 27)
    g = KeakiHipGroup([0] * 8)
    try:
        srs = g.srs_g1_upload(pts, precompute=True)
        mem = [g.member_memory(i) for i in range(8)]
        assert all(m["tables"] >= 5 * (1 << 30) for m in mem), mem      # W x 2^23 x 64 B per member: the 6 GiB of the 8-GPU configuration
        got = g.msm_g1(srs, s)
        exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k)[None, :])[0]
        assert np.array_equal(jac_to_aff(got), exp)
        # a second polynomial, shorter than the SRS: the last members' ranges are empty or cut
        m = (5 << 23) + 12345
        exp = oc.g1_mul_batch(g1, oc.fr_dot(s[:m], k[:m])[None, :])[0]
        assert np.array_equal(jac_to_aff(g.msm_g1(srs, s[:m])), exp)
        srs.free()
    finally:
        g.close()


def test_group_of_eight_config5_2p20_bits(oc):
    """BASELINE config 5 at its size on a device group of eight (the in-process form): vec_commit with the FK23 openings sharded 8 ways inside
    the library (d = 2^21) and the commit MSM by SRS range, 2 x vec_encrypt and vec_decrypt split by item range -- commitment, all 2^21
    proofs, all ciphertexts and all recovered messages byte-equal to the single-context mirror with the same seeds (whose values
    tests/test_gpu_config5.py checks against the oracle)."""
    from keaki_amd import keaki as K
    n = 1 << 20
    d = 2 * n
    secret = K.Rng(2024).fr_rand()
    bits = np.random.default_rng(7).integers(0, 2, n)
    zero, one = K.fr(0), K.fr(1)
    choices = np.where(bits[:, None] == 0, zero[None, :], one[None, :]).astype(np.uint64)
    msgs = np.random.default_rng(8).integers(0, 256, size=(n, 32), dtype=np.uint8)
    elements = K.domain_elements(n + K.PADDING_LEN)
    ones = np.repeat(one[None, :], n, 0)
    res = []
    for dev in (None, K.Device([0] * 8)):
        s = K.KZGSetup.setup(secret, d) if dev is None else K.KZGSetup.setup(secret, d, device=dev)
        try:
            rng = K.Rng(99)
            K.precompute_open_fk(s, d)
            K.kem_prepare(s, n)
            com, proofs = K.vec_commit(rng, s, choices)
            g2, body = K.vec_encrypt_arrays(rng, s, com, elements, ones, msgs)
            out = K.vec_decrypt_arrays(s, proofs[:n], g2, body)
            res.append((com, proofs, g2, body, out))
        finally:
            s.close()
            if dev is not None:
                dev.close()
    a, b = res
    assert np.array_equal(a[0], b[0]), "commitment"
    assert np.array_equal(a[1], b[1]), "FK23 proofs"
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), "ciphertexts"
    assert np.array_equal(a[4], b[4])
    assert np.array_equal(b[4][bits == 1], msgs[bits == 1]) and not np.array_equal(b[4][bits == 0], msgs[bits == 0])
