"""BASELINE config 5 at its stated size on one GPU, against the CPU oracle: the Laconic OT flow of the reference
(tests/laconic_ot.rs:126-200) with 2^20 receiver bits => evaluation domain 2^21 (src/vec.rs:27,36), through the mirrored API.

  Receiver::new     = vec::vec_commit (src/vec.rs:22-49): padding draw, iFFT, FK23 openings (src/kzg.rs:157-203), commit (:89-101)
  Sender::send      = 2 x vec::vec_encrypt (src/vec.rs:52-69)  -> 2^21 encapsulations
  Receiver::receive = vec::vec_decrypt (src/vec.rs:72-81)      -> 2^20 decapsulations (one pairing each)

What the oracle checks (never the GPU against itself): the polynomial by its own iFFT of the same evaluations; the commitment by its
Pippenger MSM over the downloaded SRS; 8 FK23 proofs as commit((p - p(w^i)) / (x - w^i)) with ITS quotient and ITS MSM; 32 ciphertexts
(point and body) of each message set by its serial `encapsulate` with the replayed r; and the flow itself: every chosen message comes
back, no unchosen one does.

And FK23 alone at d = 2^9 .. 2^18 against the oracle's per-point opening (the wave-uniform sliding-window stages only run at >= 64
blocks per stage, i.e. at these sizes)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NCPU = os.cpu_count() or 1


@pytest.fixture(scope="module")
def K():
    from keaki_amd import keaki as K
    return K


def mont(oc, ints):
    return oc.fr_to_mont(oc.ints_to_limbs(ints))


def oracle_opening(oc, srs_pts, coeffs, z_mont):
    """`open` of the reference (src/kzg.rs:104-124) entirely on the CPU oracle: its synthetic division, its Pippenger MSM"""
    q, v = oc.fr_quotient(coeffs, z_mont)
    return oc.msm_g1(srs_pts[:q.shape[0]], q, threads=NCPU), v


@pytest.mark.parametrize("log2d", [9, 12, 15, 18])
def test_open_fk_large_domains_vs_oracle(K, oc, py, log2d):
    """kzg::open_fk (FK23: the even half of the products + one inverse and one forward G1 transform) at sizes where every branch of the
    stage kernels is taken -- stages with at least 64 blocks run the wave-uniform width-5 sliding-window ladder, the last six the fixed
    windows, trivial twiddles skip the ladder: proofs at the corners and at random positions equal the ORACLE's opening at that root of
    unity (its quotient, its MSM over the downloaded SRS), and the evaluation the quotient leaves equals p(w^i)."""
    from bench import random_fr_limbs
    d = 1 << log2d
    rng = K.Rng(3300 + log2d)
    s = K.KZGSetup.setup(rng.fr_rand(), d)
    try:
        p = random_fr_limbs(d, 4400 + log2d)                 # limbs below r: valid Montgomery representatives
        p[d // 3] = 0
        proofs = K.open_fk(s, p, d)
        srs_pts = s.g1_pow()
        w = py.fr_root_of_unity(d)
        n_rand = 20 if log2d <= 15 else 10
        pick = sorted(set(np.random.default_rng(log2d).integers(0, d, n_rand).tolist() + [0, 1, d // 2 - 1, d // 2, d - 2, d - 1]))
        zs = mont(oc, [pow(w, i, py.R) for i in pick])
        for i, z in zip(pick, zs):
            exp, _ = oracle_opening(oc, srs_pts, p, z)
            assert np.array_equal(proofs[i], exp), "proof %d of %d" % (i, d)
        # one of them through the mirrored verify as well (src/kzg.rs:470-505 at scale)
        i = pick[len(pick) // 2]
        el = K.domain_elements(d)
        assert np.array_equal(el[i], zs[len(pick) // 2])
        assert K.verify(s, K.commit(s, p), el[i], K.poly_evaluate(p, el[i]), proofs[i])
    finally:
        s.close()


def test_laconic_ot_2p20_bits_vs_oracle(K, oc, py):
    log2n = int(os.environ.get("KEAKI_TEST_CONFIG5_LOG2N", "20"))        # the BASELINE size; smaller only for a quick local run
    n = 1 << log2n
    d = 2 * n                                                           # domain of n + PADDING_LEN evaluations (src/vec.rs:27,36)
    vb = 32                                                             # VALUE_BYTES (tests/laconic_ot.rs:124)
    rng = K.Rng(2024)
    secret = rng.fr_rand()
    s = K.KZGSetup.setup(secret, d)
    try:
        K.precompute_open_fk(s, d)
        np_rng = np.random.default_rng(7)
        bits = np_rng.integers(0, 2, n)
        zero, one = K.fr(0), K.fr(1)
        choices = np.where(bits[:, None] == 0, zero[None, :], one[None, :]).astype(np.uint64)

        # ---- Receiver::new (tests/laconic_ot.rs:143-148)
        com, proofs = K.vec_commit(rng, s, choices)
        assert proofs.shape == (d, 8)

        # the oracle's view of the same call: replay the padding draw, its own iFFT, its own MSM
        replay = K.Rng(2024)
        assert np.array_equal(replay.fr_rand(), secret)
        pad = replay.fr_rand()
        evals = np.zeros((d, 4), np.uint64)
        evals[:n] = choices
        evals[n] = pad
        w = py.fr_root_of_unity(d)
        coeffs = oc.fr_fft(evals, mont(oc, [pow(w, -1, py.R)])[0], mont(oc, [pow(d, -1, py.R)])[0])      # domain.ifft (src/vec.rs:37)
        srs_pts = s.g1_pow()
        assert srs_pts.shape == (d, 8)
        assert np.array_equal(com, oc.msm_g1(srs_pts, coeffs, threads=NCPU)), "commitment (src/vec.rs:46) differs from the oracle's MSM"
        pick = sorted(set([0, 1, n - 1, n, d - 1] + np.random.default_rng(1).integers(0, d, 3).tolist()))
        while len(pick) < 8:
            pick.append(pick[-1] // 2 + 3)
        zs = mont(oc, [pow(w, i, py.R) for i in pick])
        for i, z in zip(pick, zs):
            exp, val = oracle_opening(oc, srs_pts, coeffs, z)
            assert np.array_equal(val, evals[i]), "p(w^%d) is not the committed entry" % i
            assert np.array_equal(proofs[i], exp), "FK23 proof %d differs from the oracle's per-point opening" % i

        # ---- Sender::send (tests/laconic_ot.rs:150-176): set b is encrypted to "bit i == b" at the roots of unity
        sets = [np_rng.integers(0, 256, size=(n, vb), dtype=np.uint8) for _ in range(2)]
        elements = K.domain_elements(n + K.PADDING_LEN)
        assert np.array_equal(elements[1], mont(oc, [w])[0])
        zeros, ones = np.repeat(zero[None, :], n, 0), np.repeat(one[None, :], n, 0)
        g2_0, body_0 = K.vec_encrypt_arrays(rng, s, com, elements, zeros, sets[0])
        g2_1, body_1 = K.vec_encrypt_arrays(rng, s, com, elements, ones, sets[1])
        # the oracle's serial `encrypt` on sampled items, with the r the loop drew for them (one Fr::rand per item, in index order)
        r0 = replay.fr_rand_many(n)
        r1 = replay.fr_rand_many(n)
        idx = np.array(sorted(set([0, 1, n - 1] + np.random.default_rng(2).integers(0, n, 29).tolist())))
        tau_g2 = s.tau_g2()
        for g2, body, rs, vals, msgs in ((g2_0, body_0, r0, zeros, sets[0]), (g2_1, body_1, r1, ones, sets[1])):
            ect, _, ekey = oc.encap_batch(com, tau_g2, elements[idx], vals[idx], rs[idx], vb, threads=NCPU)
            assert np.array_equal(g2[idx], ect), "ciphertext points"
            assert np.array_equal(body[idx], ekey ^ msgs[idx]), "ciphertext bodies (key XOR message, src/enc.rs:32-36)"

        # ---- Receiver::receive (tests/laconic_ot.rs:178-199): the receiver opens the ciphertext of its bit with the proof of that index
        pick0 = bits[:, None] == 0
        got = K.vec_decrypt_arrays(s, proofs[:n], np.where(pick0, g2_0, g2_1), np.where(pick0, body_0, body_1))
        chosen = np.where(pick0, sets[0], sets[1])
        assert np.array_equal(got, chosen), "%d of %d chosen messages not recovered" % (int((got != chosen).any(axis=1).sum()), n)
        # and sampled decapsulations against the oracle's pairing
        _, dkey = oc.decap_batch(proofs[idx], np.where(pick0, g2_0, g2_1)[idx], vb, threads=NCPU)
        assert np.array_equal(dkey ^ np.where(pick0, body_0, body_1)[idx], chosen[idx])
        # the unchosen message stays hidden: with the same proofs the other ciphertext decrypts to something else, for every sampled item
        m = 4096
        other = K.vec_decrypt_arrays(s, proofs[:m], np.where(pick0, g2_1, g2_0)[:m], np.where(pick0, body_1, body_0)[:m])
        unchosen = np.where(pick0, sets[1], sets[0])[:m]
        assert not (other == unchosen).all(axis=1).any()
    finally:
        s.close()


@pytest.mark.parametrize("log2d,n,with_pad", [(12, 4095, True), (12, 3000, True), (13, 8192, False), (4, 7, True)])
def test_vec_commit_one_call_vs_oracle(oc, py, hip, log2d, n, with_pad):
    """keaki_hip_vec_commit = the body of vec::vec_commit (src/vec.rs:36-46) in one device call (iFFT -> FK23 openings -> commit, the coefficients
    never return to the host): commitment and sampled proofs against the oracle's own iFFT, quotient and MSM; evaluations beyond the padded
    vector are zero (ark-poly's ifft pads), a vector that fills the domain has no pad."""
    from bench import random_fr_limbs
    from keaki_amd.hip import jac_to_affine_words
    d = 1 << log2d
    g1, _ = oc.generators()
    tau = 987654321987654321 % oc.R_MOD
    pw, acc = [], 1
    for _ in range(d):
        pw.append(acc); acc = acc * tau % oc.R_MOD
    srs_pts = hip.g1_mul_batch(g1, mont(oc, pw))
    srs = hip.srs_g1_upload(srs_pts)
    hip.srs_g1_precompute(srs)
    try:
        v = random_fr_limbs(n, 7700 + n)
        pad = random_fr_limbs(1, 7800 + n)[0] if with_pad else None
        wd, w2 = py.fr_root_of_unity(d), py.fr_root_of_unity(2 * d)
        m1 = lambda x: mont(oc, [x])[0]
        com, proofs = hip.vec_commit(srs, v, pad, log2d, m1(pow(wd, -1, py.R)), m1(pow(d, -1, py.R)), m1(w2), m1(pow(w2, -1, py.R)), m1(pow(2 * d, -1, py.R)))
        evals = np.zeros((d, 4), np.uint64)
        evals[:n] = v
        if with_pad:
            evals[n] = pad
        coeffs = oc.fr_fft(evals, m1(pow(wd, -1, py.R)), m1(pow(d, -1, py.R)))
        assert np.array_equal(jac_to_affine_words(com), oc.msm_g1(srs_pts, coeffs, threads=NCPU))
        for i in sorted({0, 1, n - 1, min(n, d - 1), d - 1}):
            exp, val = oracle_opening(oc, srs_pts, coeffs, m1(pow(wd, i, py.R)))
            assert np.array_equal(val, evals[i]) and np.array_equal(proofs[i], exp), i
    finally:
        srs.free()
