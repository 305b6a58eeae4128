"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact: integer work, so every comparison is equality of canonical limbs / bytes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def mont(oc, ints):
    return oc.fr_to_mont(oc.ints_to_limbs(ints)) if len(ints) else np.zeros((0, 4), np.uint64)


def jac_to_aff(j):
    from keaki_amd.hip import jac_to_affine_words
    return jac_to_affine_words(j)


def make_points_g1(oc, hip, n, seed):
    g1, _ = oc.generators()
    from conftest import rand_fr_ints
    ks = rand_fr_ints(n, seed)
    pts = hip.g1_mul_batch(g1, mont(oc, ks)) if n else np.zeros((0, 8), np.uint64)
    return ks, pts


def test_g1_mul_batch_vs_oracle(oc, py, hip, rand_fr):
    g1, _ = oc.generators()
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd       # GLV eigenvalue: decompositions with k1 or k2 zero / tiny / negative
    ks = rand_fr(40, 11) + [0, 1, 2, 3, py.R - 1, py.R - 2, py.R - 3, (py.R - 1) // 2, (py.R + 1) // 2,   # r - 2: the NAF ladder ends in a doubling
                            lam, lam + 1, lam - 1, py.R - lam, 2 * lam % py.R, (lam * lam) % py.R, 1 << 127, (1 << 128) - 1, 1 << 128, (1 << 253) + 5,
                            9931322734385697763, 147946756881789319010696353538189108491, 147946756881789319000765030803803410728]
    km = mont(oc, ks)
    got = hip.g1_mul_batch(g1, km)
    exp = oc.g1_mul_batch(g1, km)
    assert np.array_equal(got, exp)
    # per-item points, including the identity
    pts = got.copy()
    pts[3] = 0
    k2 = mont(oc, rand_fr(32, 12))
    assert np.array_equal(hip.g1_mul_batch(pts, k2), oc.g1_mul_batch(pts, k2))


def test_g2_mul_batch_vs_oracle(oc, py, hip, rand_fr):
    _, g2 = oc.generators()
    ks = rand_fr(13, 21) + [0, 1, py.R - 1]
    km = mont(oc, ks)
    got = hip.g2_mul_batch(g2, km)
    assert np.array_equal(got, oc.g2_mul_batch(g2, km))
    k2 = mont(oc, rand_fr(16, 22))
    assert np.array_equal(hip.g2_mul_batch(got, k2), oc.g2_mul_batch(got, k2))


@pytest.mark.parametrize("n", [0, 1, 2, 31, 32, 33, 257, 1000, 4096])
def test_msm_g1_vs_oracle(oc, hip, rand_fr, n):
    _, pts = make_points_g1(oc, hip, n, 100 + n)
    sc = mont(oc, rand_fr(n, 200 + n))
    srs = hip.srs_g1_upload(pts)
    try:
        got = jac_to_aff(hip.msm_g1(srs, sc))
    finally:
        srs.free()
    assert np.array_equal(got, oc.msm_g1(pts, sc, threads=4))


def test_msm_g1_repeated_calls_of_changing_shape(oc, hip, rand_fr):
    """Back-to-back MSMs on one handle: repeated shapes with different scalars, a shape change in between, a larger call that reallocates
    the grow-only workspaces, window tables appearing on the handle -- every call must be the MSM of ITS scalars (nothing cached from the
    previous call of the same shape may leak: counts, offsets, heavy-bucket lists, bucket contents)."""
    n_a, n_b, n_big = 129, 1000, 40000
    _, pts = make_points_g1(oc, hip, n_big, 4100)
    srs = hip.srs_g1_upload(pts)
    try:
        def check(n, seed):
            sc = mont(oc, rand_fr(n, seed))
            assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc)), oc.msm_g1(pts[:n], sc, threads=4)), (n, seed)
        for rep in range(3):
            check(n_a, 4200 + rep)
            check(n_a, 4300 + rep)
            check(n_b, 4400 + rep)
        check(n_big, 4500)
        for rep in range(2):
            check(n_a, 4600 + rep)
            check(n_b, 4700 + rep)
        hip.srs_g1_precompute(srs)
        for rep in range(2):
            check(n_b, 4800 + rep)
            check(n_a, 4900 + rep)
    finally:
        srs.free()


def test_msm_g1_edge_cases(oc, py, hip, rand_fr):
    n = 300
    _, pts = make_points_g1(oc, hip, n, 7)
    ints = rand_fr(n, 8)
    ints[0] = 0; ints[1] = py.R - 1; ints[2] = 1; ints[3] = 2**253; ints[4] = (1 << 128) - 1
    pts[5] = 0                      # identity point
    pts[6] = pts[7]                 # repeated points (forces the doubling branch when digits match)
    ints[6] = ints[7]
    pts[8] = pts[9]; ints[8] = (py.R - ints[9]) % py.R   # P and -P contributions cancel
    sc = mont(oc, ints)
    srs = hip.srs_g1_upload(pts)
    try:
        assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc)), oc.msm_g1(pts, sc))
        # all-equal scalars, all-equal points
        sc2 = mont(oc, [ints[10]] * n)
        assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc2)), oc.msm_g1(pts, sc2))
        # all-zero scalars -> identity
        z = hip.msm_g1(srs, np.zeros((n, 4), np.uint64))
        assert not np.any(jac_to_aff(z)) and not np.any(z[8:])
        # prefix (polynomial shorter than the SRS): zip-truncation like msm_unchecked
        assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc[:100])), oc.msm_g1(pts[:100], sc[:100]))
    finally:
        srs.free()
    same = np.repeat(pts[11:12], 64, 0)
    srs = hip.srs_g1_upload(same)
    try:
        sc3 = mont(oc, rand_fr(64, 9))
        assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc3)), oc.msm_g1(same, sc3))
    finally:
        srs.free()


def test_msm_too_large_is_an_error(oc, hip):
    from keaki_amd.hip import KeakiHipError, KEAKI_ERR_TOO_LARGE
    g1, _ = oc.generators()
    srs = hip.srs_g1_upload(np.repeat(g1[None, :], 4, 0))
    try:
        with pytest.raises(KeakiHipError) as e:
            hip.msm_g1(srs, np.zeros((5, 4), np.uint64))
        assert e.value.status == KEAKI_ERR_TOO_LARGE
    finally:
        srs.free()


@pytest.mark.parametrize("n", [1, 33, 200, 1 << 12, 1 << 16])
def test_msm_g2_vs_oracle(oc, hip, rand_fr, n):
    _, g2 = oc.generators()
    if n <= 200:
        pts = hip.g2_mul_batch(g2, mont(oc, rand_fr(n, 300 + n)))
        sc = mont(oc, rand_fr(n, 400 + n))
    else:
        from bench import random_fr_limbs
        pts = hip.g2_mul_batch(g2, random_fr_limbs(n, 300 + n))
        sc = random_fr_limbs(n, 400 + n)
        sc[5] = 0; pts[7] = 0                                  # a zero scalar, an identity point
    srs = hip.srs_g2_upload(pts)
    try:
        got = jac_to_aff(hip.msm_g2(srs, sc))
        assert hip.srs_g2_precompute(srs) >= n * 128                 # window tables: shared buckets, no Horner doublings
        got_t = jac_to_aff(hip.msm_g2(srs, sc))
        got_short = jac_to_aff(hip.msm_g2(srs, sc[:max(1, n // 3)]))   # a shorter polynomial on the same tables (generic path)
    finally:
        srs.free()
    import os
    th = min(32, os.cpu_count() or 1)
    exp = oc.msm_g2(pts, sc, threads=th)
    assert np.array_equal(got, exp) and np.array_equal(got_t, exp)
    assert np.array_equal(got_short, oc.msm_g2(pts[:max(1, n // 3)], sc[:max(1, n // 3)], threads=th))


def test_g1_sum_of_partials(oc, hip, rand_fr):
    g1, _ = oc.generators()
    pts = hip.g1_mul_batch(g1, mont(oc, rand_fr(8, 55)))
    one = oc.fq_to_mont(oc.ints_to_limbs([1]))[0]
    jac = np.concatenate([pts, np.tile(one, (8, 1))], axis=1)
    jac[3, :4] = one; jac[3, 4:8] = one; jac[3, 8:] = 0    # identity partial (1,1,0)
    pts[3] = 0
    assert np.array_equal(jac_to_aff(hip.g1_sum(jac)), oc.g1_sum(pts))


def test_pairing_batch_vs_oracle(oc, py, hip, rand_fr):
    g1, g2 = oc.generators()
    n = 24
    P = hip.g1_mul_batch(g1, mont(oc, rand_fr(n, 31)))
    Q = hip.g2_mul_batch(g2, mont(oc, rand_fr(n, 32)))
    P[0] = g1; Q[0] = g2
    P[1] = 0            # identity in the G1 slot -> GT one
    Q[2] = 0            # identity in the G2 slot -> GT one
    got = hip.pairing_batch(P, Q)
    exp = oc.pairing_batch(P, Q, threads=8)
    assert np.array_equal(got, exp)
    one = py.gt_serialize(py.F12_ONE)
    assert got[1].tobytes() == one and got[2].tobytes() == one
    # broadcast Q (the encapsulate shape)
    assert np.array_equal(hip.pairing_batch(P, g2), oc.pairing_batch(P, g2, threads=8))


def test_encap_decap_vs_oracle(oc, py, hip, rand_fr):
    g1, g2 = oc.generators()
    n = 16
    tau, c0 = rand_fr(2, 41)
    com = hip.g1_mul_batch(g1, mont(oc, [c0]))[0]
    tau_g2 = hip.g2_mul_batch(g2, mont(oc, [tau]))[0]
    pts = mont(oc, rand_fr(n, 42)); vals = mont(oc, rand_fr(n, 43)); rs = mont(oc, rand_fr(n, 44))
    ct, gt, key = hip.encap_batch(com, tau_g2, pts, vals, rs, 32)
    ect, egt, ekey = oc.encap_batch(com, tau_g2, pts, vals, rs, 32, threads=8)
    assert np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey)
    # long XOF output and odd length
    _, _, key2 = hip.encap_batch(com, tau_g2, pts[:3], vals[:3], rs[:3], 77)
    _, _, ekey2 = oc.encap_batch(com, tau_g2, pts[:3], vals[:3], rs[:3], 77)
    assert np.array_equal(key2, ekey2)
    proofs = hip.g1_mul_batch(g1, mont(oc, rand_fr(n, 45)))
    dgt, dkey = hip.decap_batch(proofs, ct, 32)
    egt2, ekey3 = oc.decap_batch(proofs, ct, 32, threads=8)
    assert np.array_equal(dgt, egt2) and np.array_equal(dkey, ekey3)


def test_golden_vectors_through_c_abi(oc, hip):
    """The committed fixtures (tests/golden/bn254_vectors.json) against the HIP path."""
    import json, os
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bn254_vectors.json")))
    pg1 = lambda p: None if p is None else (int(p[0]), int(p[1]))
    pg2 = lambda p: None if p is None else ((int(p[0][0]), int(p[0][1])), (int(p[1][0]), int(p[1][1])))
    g1, g2 = oc.generators()
    ks = mont(oc, [int(e["k"]) for e in vec["g1_mul"]])
    assert oc.g1_to_ints(hip.g1_mul_batch(g1, ks)) == [pg1(e["p"]) for e in vec["g1_mul"]]
    assert oc.g2_to_ints(hip.g2_mul_batch(g2, ks)) == [pg2(e["p"]) for e in vec["g2_mul"]]
    for m in vec["msm_g1"]:
        bases = hip.g1_mul_batch(g1, mont(oc, [int(k) for k in m["base_dlogs"]]))
        srs = hip.srs_g1_upload(bases)
        try:
            got = jac_to_aff(hip.msm_g1(srs, mont(oc, [int(s) for s in m["scalars"]])))
        finally:
            srs.free()
        assert oc.g1_to_ints(got)[0] == pg1(m["result"])
    P = hip.g1_mul_batch(g1, mont(oc, [int(e["a"]) for e in vec["pairing"]]))
    Q = hip.g2_mul_batch(g2, mont(oc, [int(e["b"]) for e in vec["pairing"]]))
    gt = hip.pairing_batch(P, Q)
    for i, e in enumerate(vec["pairing"]):
        assert gt[i].tobytes().hex() == e["gt_hex"]
    k = vec["kem"]
    M = lambda x: mont(oc, [int(x)])
    ct, gtb, key = hip.encap_batch(oc.g1_from_ints([pg1(k["commitment"])])[0], oc.g2_from_ints([pg2(k["tau_g2"])])[0],
                                   M(k["point"]), M(k["value"]), M(k["r"]), 32)
    assert oc.g2_to_ints(ct)[0] == pg2(k["ct"]) and gtb[0].tobytes().hex() == k["gt_hex"] and key[0].tobytes().hex() == k["key_hex"]
    _, dkey = hip.decap_batch(oc.g1_from_ints([pg1(k["proof"])]), ct, 32)
    assert dkey[0].tobytes().hex() == k["key_hex"]


def test_msm_full_size_property(oc, hip, rand_fr):
    """Size-independent property at a large size: MSM(s, k_i G) == (sum s_i k_i) G, n = 2^18."""
    n = 1 << 18
    g1, _ = oc.generators()
    rng = np.random.default_rng(5)
    def limbs(n):
        a = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)          # < 2^252 < r: valid Montgomery residues
        return a
    k, s = limbs(n), limbs(n)
    pts = hip.g1_mul_batch(g1, k)
    srs = hip.srs_g1_upload(pts)
    try:
        got = jac_to_aff(hip.msm_g1(srs, s))
    finally:
        srs.free()
    assert np.array_equal(got, oc.g1_mul_batch(g1, oc.fr_dot(s, k).reshape(1, 4))[0])


@pytest.mark.parametrize("blocks,iters", [(4, 256), (1024, 64), (8192, 16)])
def test_field_asm_streams_selftest(hip, blocks, iters):
    """Hand-scheduled Fq streams vs the portable template code, on device, at low and high occupancy
    (a wait-state hazard inside an asm string would show up as rare mismatches)."""
    for seed in (1, 2, 3):
        assert hip.selftest_field(blocks, iters, seed) == 0


def test_encap_fixed_base_tables_vs_oracle(oc, py, hip, rand_fr):
    """Batches >= 256 take the fixed-base window-table path (r C - (r beta) g1, r tau_2 - (r alpha) g2);
    it must give exactly what the ladder path / the oracle's serial loop gives, including corner scalars."""
    g1, g2 = oc.generators()
    n = 320
    tau, c0 = rand_fr(2, 61)
    com = hip.g1_mul_batch(g1, mont(oc, [c0]))[0]
    tau_g2 = hip.g2_mul_batch(g2, mont(oc, [tau]))[0]
    a, v, r = rand_fr(n, 62), rand_fr(n, 63), rand_fr(n, 64)
    a[0] = 0; v[1] = 0; r[2] = 1; r[3] = py.R - 1; a[4] = py.R - 1; v[5] = 1; r[6] = 255; r[7] = 256; r[8] = (1 << 248)
    # signed 13-bit window corners (digit 2^12, 2^12 + 1, all-ones windows, carries through every window, top digit + carry >= r / 2^247)
    r[9] = 1 << 12; r[10] = (1 << 12) + 1; r[11] = (1 << 13) - 1; r[12] = (1 << 26) - 1; r[13] = (1 << 253) - 1; a[13] = 1
    r[14] = sum(((1 << 12) + 1) << (13 * j) for j in range(19)); r[15] = py.R - 2; a[15] = 1; r[16] = (96 << 247) + (1 << 246)
    A, V, Rr = mont(oc, a), mont(oc, v), mont(oc, r)
    ct, gt, key = hip.encap_batch(com, tau_g2, A, V, Rr, 32)
    ect, egt, ekey = oc.encap_batch(com, tau_g2, A, V, Rr, 32, threads=8)
    assert np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey)
    # same batch, split below the table threshold -> ladder path; results must coincide
    ct2, gt2, key2 = hip.encap_batch(com, tau_g2, A[:100], V[:100], Rr[:100], 32)
    assert np.array_equal(ct2, ct[:100]) and np.array_equal(gt2, gt[:100])
    # identity commitment (C = O): r C vanishes, only the g1 table contributes
    ct3, gt3, _ = hip.encap_batch(np.zeros(8, np.uint64), tau_g2, A[:256], V[:256], Rr[:256], 32)
    ect3, egt3, _ = oc.encap_batch(np.zeros(8, np.uint64), tau_g2, A[:256], V[:256], Rr[:256], 32, threads=8)
    assert np.array_equal(ct3, ect3) and np.array_equal(gt3, egt3)
    # the [tau]_2 window table is cached per context: another setup's tau must rebuild it, and coming back must rebuild it again
    tau_b = hip.g2_mul_batch(g2, mont(oc, [rand_fr(1, 65)[0]]))[0]
    ct4, gt4, _ = hip.encap_batch(com, tau_b, A[:256], V[:256], Rr[:256], 32)
    ect4, egt4, _ = oc.encap_batch(com, tau_b, A[:256], V[:256], Rr[:256], 32, threads=8)
    assert np.array_equal(ct4, ect4) and np.array_equal(gt4, egt4) and not np.array_equal(ct4, ct[:256])
    ct5, _, _ = hip.encap_batch(com, tau_g2, A[:256], V[:256], Rr[:256], 32)
    assert np.array_equal(ct5, ct[:256])


@pytest.mark.parametrize("N", [300, 5000, 1 << 15])
def test_msm_precomputed_tables(oc, hip, rand_fr, N):
    """keaki_hip_srs_g1_precompute: shared-bucket path over the window tables gives the same group element as the
    generic path and the oracle, for full-length and shorter polynomials (zip-truncation) and corner scalars."""
    _, pts = make_points_g1(oc, hip, N, 900 + N)
    ints = rand_fr(N, 901 + N)
    ints[0] = 0; ints[1] = oc.R_MOD - 1; ints[2] = 1; ints[3] = (1 << 253) + 12345; ints[4] = (1 << 200) - 1
    pts[5] = 0
    sc = mont(oc, ints)
    srs = hip.srs_g1_upload(pts)
    try:
        plain = jac_to_aff(hip.msm_g1(srs, sc))
        nbytes = hip.srs_g1_precompute(srs)
        assert nbytes >= N * 64 * 2
        for n in (N, N - 1, N // 2 + 1, N // 2, 7):
            got = jac_to_aff(hip.msm_g1(srs, sc[:n]))
            assert np.array_equal(got, oc.msm_g1(pts[:n], sc[:n], threads=8)), n
        assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc)), plain)
        z = hip.msm_g1(srs, np.zeros((N, 4), np.uint64))
        assert not np.any(jac_to_aff(z))
        # lengths below half of the SRS use the tables too since round 4 (one shared window: no 250-doubling tail); the old rule behind the option
        hip.set_option("msm_short_tables", 0)
        for n in (N // 2, 7, 1):
            assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc[:n])), oc.msm_g1(pts[:n], sc[:n], threads=8)), n
        hip.set_option("msm_short_tables", -1)
        assert np.array_equal(jac_to_aff(hip.msm_g1(srs, sc[:1])), oc.msm_g1(pts[:1], sc[:1], threads=1))
    finally:
        hip.set_option("msm_short_tables", -1)
        srs.free()


def test_g2_line_table_vs_oracle(oc, py, hip):
    """Every line coefficient the lane-pair Miller loop tabulates for a fixed Q (the G2Prepared of ark-ec, but along the
    device's own NAF of 6z+2) against the big-int oracle's line formulas (bn254_py._line_double / _line_add)."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("devconst", os.path.join(root, "keaki_amd", "csrc", "gen_constants.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    naf = g.naf_6z2()
    _, g2 = oc.generators()
    for k in (1, 0xC0FFEE):
        qw = hip.g2_mul_batch(g2, mont(oc, [k]))[0]
        q = oc.g2_to_ints(qw)[0]
        r = (q[0], q[1], py.F2_ONE); exp = []
        negq = py.g2_neg(q)
        for i in range(len(naf) - 2, -1, -1):
            r, l = py._line_double(r); exp.append(l)
            if naf[i]:
                r, l = py._line_add(r, q if naf[i] == 1 else negq); exp.append(l)
        q1 = py._mul_by_char(q); q2 = py._mul_by_char(q1); q2 = (q2[0], py.f2_neg(q2[1]))
        r, l = py._line_add(r, q1); exp.append(l)
        r, l = py._line_add(r, q2); exp.append(l)
        tab = hip.g2_prepare(qw)
        vals = oc.limbs_to_ints(oc.fq_from_mont(tab.reshape(-1, 4)))
        for li, e in enumerate(exp):
            got = tuple((vals[(li * 2 + 0) * 3 + c], vals[(li * 2 + 1) * 3 + c]) for c in range(3))
            assert got == e, (k, li)


def test_miller_and_final_exp_stages(oc, hip, rand_fr):
    """Stage-wise: device final exponentiation of the ORACLE's Miller output, and oracle final exponentiation of the DEVICE's
    Miller output (the two Miller loops differ by subfield factors only, which the final exponentiation kills)."""
    g1, g2 = oc.generators()
    P = hip.g1_mul_batch(g1, mont(oc, rand_fr(5, 71)))
    Q = hip.g2_mul_batch(g2, mont(oc, rand_fr(5, 72)))
    exp = oc.pairing_batch(P, Q)
    f_or = np.stack([oc.miller_loop_raw(P[i], Q[i]) for i in range(5)])
    assert np.array_equal(hip.final_exp_batch(f_or), exp)
    f_dev = hip.miller_loop_batch(P, Q)
    for i in range(5):
        e = oc.final_exp_raw(f_dev[i])
        canon = oc.fq_from_mont(e.reshape(-1, 4)).tobytes()
        assert canon == exp[i].tobytes()


def test_encap_gt_fixed_base_path_vs_oracle(oc, py, hip, rand_fr, request):
    """Large batches compute GT_i = A^(r_i) * B^(-beta_i r_i) with A = e(C, g2), B = e(g1, g2) (fixed-base GT tables) instead of
    one pairing per item. Forced here with a low threshold; the bytes must equal the oracle's serial e(r(C - beta g1), g2)."""
    hip.set_option("encap_gt", 64)
    request.addfinalizer(lambda: hip.set_option("encap_gt", -1))      # the shared context goes back to the automatic policy
    g1, g2 = oc.generators()
    n = 300
    tau, c0, c1 = rand_fr(3, 81)
    tau_g2 = hip.g2_mul_batch(g2, mont(oc, [tau]))[0]
    a, v, r = rand_fr(n, 82), rand_fr(n, 83), rand_fr(n, 84)
    v[0] = 0; r[1] = 1; r[2] = py.R - 1; v[3] = py.R - 1; r[4] = 0; v[5] = 1; r[5] = 255; r[6] = 256; r[7] = (1 << 248) + 5
    # signed 13-bit window corners: digit exactly 2^12 (kept), 2^12 + 1 (negative, carry), all-ones windows (2^13 after the carry -> 0),
    # a run of carries through every window, the largest exponent below 2^253
    r[8] = 1 << 12; r[9] = (1 << 12) + 1; r[10] = (1 << 13) - 1; r[11] = (1 << 26) - 1; r[12] = (1 << 253) - 1; v[12] = 1
    r[13] = sum(((1 << 12) + 1) << (13 * j) for j in range(19)); r[14] = (1 << 13); r[15] = (1 << 13) + (1 << 12)
    # the same corners for the 16-bit table A gets when its commitment repeats, and for the 20-bit table of B (exponent -(r beta): r = 1)
    r[16] = 1 << 15; r[17] = (1 << 15) + 1; r[18] = (1 << 16) - 1; r[19] = sum(((1 << 15) + 1) << (16 * j) for j in range(15))
    for k, e in enumerate([1 << 19, (1 << 19) + 1, (1 << 20) - 1, sum(((1 << 19) + 1) << (20 * j) for j in range(12)), (1 << 253) + 12345]):
        r[20 + k] = 1; v[20 + k] = py.R - e
    A, V, Rr = mont(oc, a), mont(oc, v), mont(oc, r)
    for c in (c0, c1):     # two commitments, twice each: the per-commitment cache, first the 13-bit table of A, then its 16-bit one
        com = hip.g1_mul_batch(g1, mont(oc, [c]))[0]
        for _ in range(2):
            ct, gt, key = hip.encap_batch(com, tau_g2, A, V, Rr, 32)
            ect, egt, ekey = oc.encap_batch(com, tau_g2, A, V, Rr, 32, threads=8)
            assert np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey)
    # identity commitment: A = 1
    z = np.zeros(8, np.uint64)
    ct, gt, _ = hip.encap_batch(z, tau_g2, A[:128], V[:128], Rr[:128], 32)
    ect, egt, _ = oc.encap_batch(z, tau_g2, A[:128], V[:128], Rr[:128], 32, threads=8)
    assert np.array_equal(ct, ect) and np.array_equal(gt, egt)
    # and the per-item pairing path agrees with it
    hip.set_option("encap_gt", 1000000000)
    com = hip.g1_mul_batch(g1, mont(oc, [c0]))[0]
    ct2, gt2, _ = hip.encap_batch(com, tau_g2, A, V, Rr, 32)
    hip.set_option("encap_gt", 64)
    ct3, gt3, _ = hip.encap_batch(com, tau_g2, A, V, Rr, 32)
    assert np.array_equal(gt2, gt3) and np.array_equal(ct2, ct3)


def test_encap_prepare_changes_nothing_but_the_first_call(oc, rand_fr):
    """keaki_hip_encap_prepare builds the setup-only tables ahead of time: a prepared context and a fresh one give the same bytes (per-item
    pairing path at 300 items, GT fixed-base path at 2^16), the prepared one holds its tables before the first batch, and a second [tau]_2
    replaces the first one's table."""
    from keaki_amd.hip import KeakiHip
    a, b = KeakiHip(0), KeakiHip(0)
    try:
        g1, g2 = oc.generators()
        tau_g2 = a.g2_mul_batch(g2, mont(oc, [4242]))[0]
        tau2 = a.g2_mul_batch(g2, mont(oc, [4243]))[0]
        com = a.g1_mul_batch(g1, mont(oc, [77]))[0]
        a.encap_prepare(tau2, 1 << 16)
        a.encap_prepare(tau_g2, 1 << 16)                 # the setup's [tau]_2 changes: its table is rebuilt
        assert a.memory()["gt_tables"] > (1 << 30) and b.memory()["gt_tables"] == 0
        n = 300
        A, V, Rr = (mont(oc, rand_fr(n, 9900 + k)) for k in range(3))
        ra, rb = a.encap_batch(com, tau_g2, A, V, Rr, 32), b.encap_batch(com, tau_g2, A, V, Rr, 32)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
        e = oc.encap_batch(com, tau_g2, A, V, Rr, 32, threads=8)
        assert all(np.array_equal(x, y) for x, y in zip(ra, e))
        m = 1 << 16
        idx = np.random.default_rng(3).integers(0, n, m)
        ra, rb = a.encap_batch(com, tau_g2, A[idx], V[idx], Rr[idx], 32), b.encap_batch(com, tau_g2, A[idx], V[idx], Rr[idx], 32)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
        assert np.array_equal(ra[0][:64], e[0][idx[:64]]) and np.array_equal(ra[2][:64], e[2][idx[:64]])
        a.encap_prepare(tau_g2, 0)                       # a hint of zero items is a no-op
    finally:
        a.close(); b.close()


def test_encap_keys_only(oc, rand_fr):
    """encap_batch with gt_out = NULL (what vec_encrypt asks for: ciphertext points and keys, the 384 GT bytes per item stay on the device):
    same points and keys as with the GT output, and as the oracle's."""
    from keaki_amd.hip import KeakiHip
    a = KeakiHip(0)
    try:
        g1, g2 = oc.generators()
        tau_g2 = a.g2_mul_batch(g2, mont(oc, [91]))[0]
        com = a.g1_mul_batch(g1, mont(oc, [92]))[0]
        n = 700
        A, V, Rr = (mont(oc, rand_fr(n, 9950 + k)) for k in range(3))
        e = oc.encap_batch(com, tau_g2, A, V, Rr, 40, threads=8)
        ct, key = a.encap_batch(com, tau_g2, A, V, Rr, 40, want_gt=False)
        assert np.array_equal(ct, e[0]) and np.array_equal(key, e[2])
        full = a.encap_batch(com, tau_g2, A, V, Rr, 40)
        assert all(np.array_equal(x, y) for x, y in zip(full, e))
    finally:
        a.close()


def test_encap_small_calls_switch_to_gt_path_when_commitment_repeats(oc, py, rand_fr, monkeypatch):
    """Under the automatic policy (option encap_gt = -1) the third consecutive call with one commitment (any batch size) builds the tables of A = e(C, g2) and B and
    takes the GT fixed-base path; later calls reuse them, a different commitment goes back to the per-item path. Own context (the policy is
    per context); every call against the oracle."""
    from keaki_amd.hip import KeakiHip
    monkeypatch.delenv("KEAKI_ENCAP_GT", raising=False)      # the environment only matters while a context is created
    monkeypatch.delenv("KEAKI_GT_WB_B", raising=False)
    h = KeakiHip(0)
    try:
        g1, g2 = oc.generators()
        tau, c0, c1 = rand_fr(3, 281)
        tau_g2 = h.g2_mul_batch(g2, mont(oc, [tau]))[0]
        coms = [h.g1_mul_batch(g1, mont(oc, [c]))[0] for c in (c0, c1)]
        seq = [0, 0, 0, 0, 0, 1, 0, 0, 1, 1, 1, 1]          # which commitment; the GT path starts at calls 3 and (table cached) 7, 8, and 12
        for it, ci in enumerate(seq):
            n = [1, 1, 1, 2, 37, 1, 1, 300, 1, 5, 1, 1][it]
            A, V, Rr = (mont(oc, rand_fr(n, 290 + 3 * it + k)) for k in range(3))
            ct, gt, key = h.encap_batch(coms[ci], tau_g2, A, V, Rr, 32)
            ect, egt, ekey = oc.encap_batch(coms[ci], tau_g2, A, V, Rr, 32, threads=8)
            assert np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey), "call %d" % it
        # a context that built the 16-bit table of B for small calls widens it (and rebuilds A) when a large batch comes: sample against the oracle
        n = 65536
        rng = np.random.default_rng(5)
        base = mont(oc, rand_fr(64, 400))
        A, V, Rr = (base[rng.integers(0, 64, n)] for _ in range(3))
        ct, gt, key = h.encap_batch(coms[1], tau_g2, A, V, Rr, 32)
        idx = np.array([0, 1, 777, 40000, n - 1])
        ect, egt, ekey = oc.encap_batch(coms[1], tau_g2, A[idx], V[idx], Rr[idx], 32, threads=8)
        assert np.array_equal(ct[idx], ect) and np.array_equal(gt[idx], egt) and np.array_equal(key[idx], ekey)
        ct2, gt2, key2 = h.encap_batch(coms[1], tau_g2, A[:300], V[:300], Rr[:300], 32)          # and small batches keep using the wide tables
        assert np.array_equal(ct2, ct[:300]) and np.array_equal(gt2, gt[:300]) and np.array_equal(key2, key[:300])
    finally:
        h.close()


@pytest.mark.parametrize("wb", [16, 11])
def test_encap_gt_path_other_window_widths(oc, py, rand_fr, monkeypatch, wb):
    """The constant-base GT table at another window width (16 bits is what a context falls back to when the 2.6 GB 20-bit table does not
    fit; 11 leaves a 1-bit top window): own context, since the table is built once per context."""
    from keaki_amd.hip import KeakiHip
    monkeypatch.setenv("KEAKI_ENCAP_GT", "64")                 # read once, by keaki_hip_ctx_create: the initial value of option encap_gt
    h = KeakiHip(0)
    monkeypatch.setenv("KEAKI_ENCAP_GT", "1000000000")         # changing the environment afterwards must not matter
    h.set_option("gt_wb_b", wb)
    try:
        g1, g2 = oc.generators()
        n = 96
        tau, c0 = rand_fr(2, 181)
        tau_g2 = h.g2_mul_batch(g2, mont(oc, [tau]))[0]
        com = h.g1_mul_batch(g1, mont(oc, [c0]))[0]
        a, v, r = rand_fr(n, 182), rand_fr(n, 183), rand_fr(n, 184)
        r[0] = 1; v[0] = py.R - ((1 << (wb - 1)) + 1); r[1] = 1; v[1] = py.R - ((1 << wb) - 1); r[2] = 1; v[2] = 1     # exponent of B: 2^(wb-1)+1, 2^wb-1, r-1
        A, V, Rr = mont(oc, a), mont(oc, v), mont(oc, r)
        ect, egt, ekey = oc.encap_batch(com, tau_g2, A, V, Rr, 32, threads=8)
        for _ in range(2):
            ct, gt, key = h.encap_batch(com, tau_g2, A, V, Rr, 32)
            assert np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey)
    finally:
        h.close()


def test_msm_structured_scalars_heavy_buckets(oc, hip, rand_fr):
    """0/1 and repeated coefficients put tens of thousands of points into single buckets (the heavy-bucket path: sliced
    accumulation by whole workgroups). Checked via the O(n) identity MSM(s, k_i G) == (sum s_i k_i) G, with and without tables."""
    n = 1 << 17
    g1, _ = oc.generators()
    rng = np.random.default_rng(17)
    k = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    pts = hip.g1_mul_batch(g1, k)
    one = oc.fr_to_mont(oc.ints_to_limbs([1]))[0]
    seven = oc.fr_to_mont(oc.ints_to_limbs([7]))[0]
    rnd = mont(oc, rand_fr(1, 99))[0]
    cases = {"bits": np.where(rng.integers(0, 2, n)[:, None] == 1, one[None, :], np.zeros((1, 4), np.uint64)).astype(np.uint64),
             "all_ones": np.repeat(one[None, :], n, 0), "all_equal_random": np.repeat(rnd[None, :], n, 0),
             "two_values": np.where(rng.integers(0, 2, n)[:, None] == 1, seven[None, :], rnd[None, :]).astype(np.uint64)}
    srs = hip.srs_g1_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g1_precompute(srs)
            for name, sc in cases.items():
                sc = np.ascontiguousarray(sc)
                got = jac_to_aff(hip.msm_g1(srs, sc))
                exp = oc.g1_mul_batch(g1, oc.fr_dot(sc, k).reshape(1, 4))[0]
                assert np.array_equal(got, exp), (name, tables)
    finally:
        srs.free()


def test_msm_one_huge_bucket_behind_many_medium_ones(oc, hip):
    """the size sort clamps bucket counts at 1023, so with hundreds of buckets above that a HUGE bucket has no reason to stand
    among the first few of the order. Heavy buckets are now selected by count: half of 2^20 coefficients are 1 (one bucket with 2^19
    points), the rest are small values 2..401 (400 buckets of ~1300 points). Correct by the O(n) identity and finished in well under
    the seconds a single lane would need for 2^19 serial additions."""
    import time
    n = 1 << 20
    g1, _ = oc.generators()
    rng = np.random.default_rng(23)
    k = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    pts = hip.g1_mul_batch(g1, k)
    vals = np.where(rng.integers(0, 2, n) == 1, 1, rng.integers(2, 402, n))
    sc = oc.fr_to_mont(np.concatenate([vals.astype(np.uint64)[:, None], np.zeros((n, 3), np.uint64)], 1))
    exp = oc.g1_mul_batch(g1, oc.fr_dot(sc, k).reshape(1, 4))[0]
    srs = hip.srs_g1_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g1_precompute(srs)
            hip.msm_g1(srs, sc)                                   # first call sizes the workspaces
            t0 = time.perf_counter()
            got = jac_to_aff(hip.msm_g1(srs, sc))
            dt = time.perf_counter() - t0
            assert np.array_equal(got, exp), tables
            assert dt < 0.5, "heavy bucket handled by a single lane? %.3f s" % dt
    finally:
        srs.free()


def _fr_ints(oc, mont_limbs):
    return oc.limbs_to_ints(oc.fr_from_mont(np.asarray(mont_limbs).reshape(-1, 4)))


@pytest.mark.parametrize("log2n", [0, 1, 2, 5])
def test_fr_fft_vs_naive_dft(oc, py, hip, rand_fr, log2n):
    """Device scalar-field DFT (row f-4; ark-poly domain.fft / ifft over Fr) against the O(n^2) definition with ark-poly's root."""
    n = 1 << log2n
    a = rand_fr(n, 700 + log2n)
    if n > 2:
        a[1] = 0
        a[2] = py.R - 1
    w = py.fr_root_of_unity(n)
    got = _fr_ints(oc, hip.fr_fft(mont(oc, a), log2n, mont(oc, [w])[0]))
    assert got == [sum(a[i] * pow(w, i * j, py.R) for i in range(n)) % py.R for j in range(n)]
    # inverse transform with the 1/n scaling folded in returns the input
    back = hip.fr_fft(mont(oc, got), log2n, mont(oc, [pow(w, -1, py.R)])[0], mont(oc, [pow(n, -1, py.R)])[0])
    assert _fr_ints(oc, back) == a


def test_fr_fft_large_round_trip_and_spot_values(oc, py, hip, rand_fr):
    """2^16 elements: forward then inverse is the identity; X[0] = sum a_i, X[n/2] = sum (-1)^i a_i, one random bin by Horner."""
    log2n, n = 16, 1 << 16
    a = rand_fr(n, 716)
    am = mont(oc, a)
    w = py.fr_root_of_unity(n)
    X = hip.fr_fft(am, log2n, mont(oc, [w])[0])
    Xi = _fr_ints(oc, X)
    assert Xi[0] == sum(a) % py.R
    assert Xi[n // 2] == sum(v if i % 2 == 0 else -v for i, v in enumerate(a)) % py.R
    j = 12345
    wj, acc = pow(w, j, py.R), 0
    for v in reversed(a):
        acc = (acc * wj + v) % py.R
    assert Xi[j] == acc
    back = hip.fr_fft(X, log2n, mont(oc, [pow(w, -1, py.R)])[0], mont(oc, [pow(n, -1, py.R)])[0])
    assert np.array_equal(back, am)


@pytest.mark.parametrize("log2d", [0, 1, 3, 6])
def test_open_fk_poly_matches_explicit_twiddle_entry(oc, py, hip, rand_fr, log2d):
    """keaki_hip_open_fk_poly (hat_a and twiddles derived on the device) == keaki_hip_open_fk fed with host-computed
    hat_a / twiddles, and == the oracle's literal FK23 at d <= 8."""
    d = 1 << log2d
    ks, pts = make_points_g1(oc, hip, max(d, 2), 800 + log2d)      # any points serve as an "SRS" for the linear-algebra identity
    p = rand_fr(d, 810 + log2d)
    w2 = py.fr_root_of_unity(2 * d)
    w2i, inv = pow(w2, -1, py.R), pow(2 * d, -1, py.R)
    a = [0] * d + p
    hat_a = [sum(a[i] * pow(w2, i * j, py.R) for i in range(2 * d)) * inv % py.R for j in range(2 * d)]
    tw = [pow(w2, k, py.R) for k in range(d)]
    twi = [pow(w2i, k, py.R) for k in range(d)]
    twd = [tw[2 * k] for k in range(d // 2)] or [1]
    srs_a, srs_b = hip.srs_g1_upload(pts), hip.srs_g1_upload(pts)
    try:
        explicit = hip.open_fk(srs_a, log2d, mont(oc, hat_a), mont(oc, tw), mont(oc, twi), mont(oc, twd))
        poly = hip.open_fk_poly(srs_b, log2d, mont(oc, p), mont(oc, [w2])[0], mont(oc, [w2i])[0], mont(oc, [inv])[0])
        assert np.array_equal(explicit, poly)
        if d <= 8:
            g1p = oc.g1_to_ints(pts)
            assert oc.g1_to_ints(poly) == py.kzg_open_fk(g1p[:d], p)
        # second call reuses the cached hat_s
        assert np.array_equal(hip.open_fk_poly(srs_b, log2d, mont(oc, p), mont(oc, [w2])[0], mont(oc, [w2i])[0], mont(oc, [inv])[0]), poly)
    finally:
        srs_a.free(); srs_b.free()


def test_curve_checks(oc, py, hip, rand_fr):
    """keaki_hip_srs_g1_check / keaki_hip_g2_check (SRS ingest, row f-3): valid multiples of the generators and the identity pass;
    every corrupted point is counted and the first index is reported."""
    g1, g2 = oc.generators()
    n = 3000
    pts = hip.g1_mul_batch(g1, mont(oc, rand_fr(n, 901)))
    pts[17] = 0                                     # identity
    srs = hip.srs_g1_upload(pts)
    try:
        assert hip.srs_g1_check(srs) == (0, None)
    finally:
        srs.free()
    bad = pts.copy()
    bad[5, 0] ^= np.uint64(2); bad[2999, 7] ^= np.uint64(1 << 40); bad[100, 4:8] = bad[101, 4:8]     # x, y limb, y of another point
    srs = hip.srs_g1_upload(bad)
    try:
        assert hip.srs_g1_check(srs) == (3, 5)
    finally:
        srs.free()
    q = hip.g2_mul_batch(g2, mont(oc, rand_fr(500, 902)))
    assert hip.g2_check(q) == (0, None)
    qb = q.copy(); qb[499, 3] ^= np.uint64(8); qb[250, 12] ^= np.uint64(1)
    assert hip.g2_check(qb) == (2, 250)
    assert hip.g2_check(np.zeros((0, 16), np.uint64)) == (0, None)


def test_kzg_verify_abi_degenerate_left_point(oc, py, hip, rand_fr):
    """keaki_hip_kzg_verify evaluates e(C - v g1 + z proof, g2) == e(proof, [tau]_2). Inputs chosen so that the left point is the IDENTITY
    (C = v g1 - z proof): the reference's equation e(C - v g1, g2) == e(proof, [tau]_2 - z g2) then reads e(proof, [tau]_2) == 1, false for
    proof != O and true for proof == O; plus an honest opening and the oracle's verdict on the same integers."""
    g1, g2 = oc.generators()
    tau, v, z, k = rand_fr(4, 4242)
    tau_g2 = hip.g2_mul_batch(g2, mont(oc, [tau]))[0]
    G = py.G1_GEN
    proof_i = py.g1_mul(G, k)
    com_i = py.g1_add(py.g1_mul(G, v), py.g1_neg(py.g1_mul(proof_i, z)))
    com, proof = oc.g1_from_ints([com_i])[0], oc.g1_from_ints([proof_i])[0]
    zm, vm = mont(oc, [z])[0], mont(oc, [v])[0]
    assert hip.kzg_verify(com, tau_g2, zm, vm, proof) is False
    assert py.kzg_verify(py.g2_mul(py.G2_GEN, tau), com_i, z, v, proof_i) is False
    zero = np.zeros(8, np.uint64)
    com0 = oc.g1_from_ints([py.g1_mul(G, v)])[0]                     # proof = O, C = v g1: both sides are 1
    assert hip.kzg_verify(com0, tau_g2, zm, vm, zero) is True
    # an honest opening of p(x) = c0 + c1 x: proof = c1 g1, C = (c0 + c1 tau) g1, value = c0 + c1 z
    c0, c1 = rand_fr(2, 4243)
    comh = oc.g1_from_ints([py.g1_mul(G, (c0 + c1 * tau) % py.R)])[0]
    prh = oc.g1_from_ints([py.g1_mul(G, c1)])[0]
    val = (c0 + c1 * z) % py.R
    assert hip.kzg_verify(comh, tau_g2, zm, mont(oc, [val])[0], prh) is True
    assert hip.kzg_verify(comh, tau_g2, zm, mont(oc, [(val + 1) % py.R])[0], prh) is False
    assert hip.kzg_verify(comh, tau_g2, mont(oc, [(z + 1) % py.R])[0], mont(oc, [val])[0], prh) is False


@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 34, 1024, 1025, 1057, 5000, 32769, 70000])
def test_kzg_open_quotient_on_device(oc, py, hip, rand_fr, n):
    """keaki_hip_kzg_open (row f-4): p(z) and the commitment of (p - p(z)) / (x - z), the quotient made on the device by the blockwise
    Horner recurrence (block 32: sizes around one, two, three and four levels). Against the oracle's Horner quotient + MSM."""
    N = max(n, 2)
    ks, pts = make_points_g1(oc, hip, N, 950)
    c = rand_fr(n, 951 + n)
    if n > 4:
        c[1] = 0; c[n // 2] = 0
    z = rand_fr(1, 960 + n)[0]
    srs = hip.srs_g1_upload(pts)
    try:
        proof, val = hip.kzg_open(srs, mont(oc, c), mont(oc, [z])[0])
        assert _fr_ints(oc, val)[0] == (py.poly_eval(c, z) if n else 0)
        q = py.poly_quotient(c, z) if n > 1 else []
        assert len(q) == max(n - 1, 0)
        exp = oc.msm_g1(pts[:len(q)], mont(oc, q)) if q else None
        got = jac_to_aff(proof)
        if q:
            assert np.array_equal(got, exp)
        else:
            assert not np.any(got)
        # z = 0: the quotient is a shift
        proof0, val0 = hip.kzg_open(srs, mont(oc, c), mont(oc, [0])[0])
        assert _fr_ints(oc, val0)[0] == (c[0] if n else 0)
        if n > 1:
            assert np.array_equal(jac_to_aff(proof0), oc.msm_g1(pts[:n - 1], mont(oc, c[1:])))
    finally:
        srs.free()


def test_concurrent_host_threads_one_context_and_two(oc, hip, rand_fr):
    """The ABI serialises calls on one context (a mutex per context) and contexts are independent: four host threads -- two on the shared
    context, two on contexts of their own -- run MSMs, pairings and encapsulations at the same time (ctypes releases the GIL); every result
    against the oracle values computed beforehand."""
    import threading
    from keaki_amd.hip import KeakiHip
    g1, g2 = oc.generators()
    n = 3000
    pts = hip.g1_mul_batch(g1, mont(oc, rand_fr(n, 7001)))
    jobs = []
    for k in range(4):
        sc = mont(oc, rand_fr(n, 7010 + k))
        P = hip.g1_mul_batch(g1, mont(oc, rand_fr(8, 7020 + k)))
        Q = hip.g2_mul_batch(g2, mont(oc, rand_fr(8, 7030 + k)))
        com = hip.g1_mul_batch(g1, mont(oc, rand_fr(1, 7040 + k)))[0]
        tau_g2 = hip.g2_mul_batch(g2, mont(oc, rand_fr(1, 7050 + k)))[0]
        A, V, Rr = (mont(oc, rand_fr(40, 7060 + 3 * k + j)) for j in range(3))
        jobs.append({"sc": sc, "P": P, "Q": Q, "com": com, "tau": tau_g2, "A": A, "V": V, "R": Rr,
                     "msm": oc.msm_g1(pts, sc, threads=8), "gt": oc.pairing_batch(P, Q, threads=8),
                     "enc": oc.encap_batch(com, tau_g2, A, V, Rr, 32, threads=8)})
    own = [KeakiHip(0), KeakiHip(0)]
    ctxs = [hip, hip, own[0], own[1]]
    errors = []

    def work(k):
        try:
            h, j = ctxs[k], jobs[k]
            srs = h.srs_g1_upload(pts)
            try:
                for _ in range(3):
                    assert np.array_equal(jac_to_aff(h.msm_g1(srs, j["sc"])), j["msm"]), "msm"
                    assert np.array_equal(h.pairing_batch(j["P"], j["Q"]), j["gt"]), "pairing"
                    ct, gt, key = h.encap_batch(j["com"], j["tau"], j["A"], j["V"], j["R"], 32)
                    assert np.array_equal(ct, j["enc"][0]) and np.array_equal(gt, j["enc"][1]) and np.array_equal(key, j["enc"][2]), "encap"
            finally:
                srs.free()
        except Exception as e:                       # noqa: BLE001 -- reported by the main thread
            errors.append("thread %d: %r" % (k, e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for h in own:
        h.close()
    assert not errors, errors


def test_one_context_four_threads_different_sizes(oc, rand_fr):
    """VERDICT r01 'next' item 2: four host threads on ONE context, a different batch size in every call (so the shared staging buffers
    are re-reserved, i.e. freed and reallocated, while other threads wait), 200 iterations of g1_mul_batch / pairing_batch / encap_batch,
    every result against oracle values computed beforehand. With the lock released between staging, kernel and download (round 1) a
    second thread could overwrite or free the staged inputs in the gap; the ABI now holds one lock across the whole call."""
    import threading
    from keaki_amd.hip import KeakiHip
    h = KeakiHip(0)
    g1, g2 = oc.generators()
    base = h.g1_mul_batch(g1, mont(oc, rand_fr(48, 9100)))
    qbase = h.g2_mul_batch(g2, mont(oc, rand_fr(12, 9101)))
    tau_g2 = h.g2_mul_batch(g2, mont(oc, rand_fr(1, 9102)))[0]
    jobs = []
    for k in range(4):
        per = []
        for v, n in enumerate([3 + 5 * k, 40 - 7 * k, 11 + k]):          # sizes differ between threads and between consecutive calls
            sc = mont(oc, rand_fr(n, 9200 + 10 * k + v))
            P, Q = base[:min(n, 12)], qbase[:min(n, 12)]
            com = base[20 + k]
            A, V, Rr = (mont(oc, rand_fr(n, 9300 + 30 * k + 3 * v + j)) for j in range(3))
            per.append({"pts": base[:n], "sc": sc, "mul": oc.g1_mul_batch(base[:n], sc, threads=8), "P": P, "Q": Q, "gt": oc.pairing_batch(P, Q, threads=8),
                        "com": com, "A": A, "V": V, "R": Rr, "enc": oc.encap_batch(com, tau_g2, A, V, Rr, 32, threads=8)})
        jobs.append(per)
    errors = []

    def work(k):
        try:
            for it in range(200):
                j = jobs[k][it % 3]
                op = (it + k) % 3
                if op == 0:
                    assert np.array_equal(h.g1_mul_batch(j["pts"], j["sc"]), j["mul"]), "g1_mul_batch it=%d" % it
                elif op == 1:
                    assert np.array_equal(h.pairing_batch(j["P"], j["Q"]), j["gt"]), "pairing_batch it=%d" % it
                else:
                    ct, gt, key = h.encap_batch(j["com"], tau_g2, j["A"], j["V"], j["R"], 32)
                    assert np.array_equal(ct, j["enc"][0]) and np.array_equal(gt, j["enc"][1]) and np.array_equal(key, j["enc"][2]), "encap_batch it=%d" % it
        except Exception as e:                       # noqa: BLE001 -- reported by the main thread
            errors.append("thread %d: %r" % (k, e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    h.close()
    assert not errors, errors


def test_optional_memory_is_optional(oc, py, rand_fr, monkeypatch):
    """VERDICT r01 'next' item 8: when the SRS window tables do not fit (keaki_hip_debug_set_alloc_limit makes every allocation of the
    context above the limit fail with KEAKI_ERR_OOM) the ABI reports OOM and the handle keeps working through the generic path; KZGSetup survives and says so;
    the wide GT table of B falls back to the 16-bit one (201 MB) and encapsulation stays bit-exact."""
    from keaki_amd.hip import KeakiHip, KeakiHipError
    from keaki_amd import keaki as K
    h = KeakiHip(0)
    try:
        n = 5000
        g1, g2 = oc.generators()
        pts = h.g1_mul_batch(g1, mont(oc, rand_fr(n, 9400)))
        sc = mont(oc, rand_fr(n, 9401))
        exp = oc.msm_g1(pts, sc, threads=8)
        srs = h.srs_g1_upload(pts)
        h.msm_g1(srs, sc)                                               # workspaces exist before the limit is set
        h.debug_set_alloc_limit(1 << 20)                                # the table of 5000 points is ~4 MB
        with pytest.raises(KeakiHipError) as e:
            h.srs_g1_precompute(srs)
        assert e.value.status == -3 and "keaki_hip_debug_set_alloc_limit" in e.value.message
        assert np.array_equal(jac_to_aff(h.msm_g1(srs, sc)), exp)
        h.debug_set_alloc_limit(0)
        assert h.srs_g1_precompute(srs) > 0
        assert np.array_equal(jac_to_aff(h.msm_g1(srs, sc)), exp)
        srs.free()
        # the host mirror: setup survives the failed table build
        dev = K.Device(0)
        dev.hip().debug_set_alloc_limit(1 << 20)
        s = K.KZGSetup.from_powers(pts, h.g2_mul_batch(g2, mont(oc, [5]))[0], device=dev)
        dev.hip().debug_set_alloc_limit(0)
        assert s.has_window_tables() is False
        assert np.array_equal(K.commit(s, sc), exp)
        s.close()
        s2 = K.KZGSetup.from_powers(pts, h.g2_mul_batch(g2, mont(oc, [5]))[0])
        assert s2.has_window_tables() is True and np.array_equal(K.commit(s2, sc), exp)
        s2.close()
    finally:
        h.close()
    # GT table of B: 2.6 GB at 20-bit windows -> refused -> 16-bit table (201 MB); a batch of 2^16 takes the GT path
    h = KeakiHip(0)
    try:
        m = 1 << 16
        com = pts[1]
        tau_g2 = h.g2_mul_batch(g2, mont(oc, [777]))[0]
        rng = np.random.default_rng(11)
        def limbs(k):
            a = rng.integers(0, 2**63, size=(k, 4), dtype=np.int64).astype(np.uint64)
            a[:, 3] &= np.uint64((1 << 60) - 1)
            return a
        A, V, Rr = limbs(m), limbs(m), limbs(m)
        h.debug_set_alloc_limit(1 << 30)
        ct, gt, key = h.encap_batch(com, tau_g2, A, V, Rr, 32)
        h.debug_set_alloc_limit(0)
        idx = np.array([0, 1, 2, m - 1, 12345])
        ect, egt, ekey = oc.encap_batch(com, tau_g2, A[idx], V[idx], Rr[idx], 32, threads=8)
        assert np.array_equal(ct[idx], ect) and np.array_equal(gt[idx], egt) and np.array_equal(key[idx], ekey)
    finally:
        h.close()


def test_pytorch_can_start_after_the_library():
    """One HIP runtime per process whatever the import order: a fresh interpreter that creates a context FIRST and imports PyTorch
    afterwards (PyTorch wheels bundle their own libamdhip64: loaded second it would be a second runtime and torch would report
    "No HIP GPUs are available") can still allocate on the GPU, and the library's results are unchanged."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from keaki_amd.hip import KeakiHip\n"
            "h = KeakiHip(0); assert h.selftest_field(4, 2, 1) == 0\n"
            "import torch\n"
            "x = torch.arange(8, device='cuda'); assert int(x.sum().item()) == 28\n"
            "assert h.selftest_field(4, 2, 2) == 0; h.close(); print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)      # a fresh box pages PyTorch in for a minute or two
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


@pytest.mark.parametrize("msg_len", [1, 31, 32, 33, 64, 65, 200])
def test_encrypt_decrypt_batch_device_dem_vs_oracle(oc, hip, msg_len):
    """keaki_hip_encrypt_batch / _decrypt_batch = enc::encrypt / enc::decrypt per item (src/enc.rs:19-55 inside src/vec.rs:63-66, :75-78) with
    the XOR DEM on the device: ciphertext points as encap_batch, bodies == the oracle's key XOR message, and the round trip through the
    proofs' pairing relation (a valid opening decrypts, a wrong proof does not)."""
    from bench import random_fr_limbs
    n = 777
    g1, g2 = oc.generators()
    rng = np.random.default_rng(msg_len)
    com = hip.g1_mul_batch(g1, random_fr_limbs(1, 91))[0]
    tau_g2 = hip.g2_mul_batch(g2, random_fr_limbs(1, 92))[0]
    A, V, R = random_fr_limbs(n, 93), random_fr_limbs(n, 94), random_fr_limbs(n, 95)
    msgs = rng.integers(0, 256, (n, msg_len), dtype=np.uint8)
    ct, body = hip.encrypt_batch(com, tau_g2, A, V, R, msgs)
    ect, _, ekey = oc.encap_batch(com, tau_g2, A, V, R, msg_len, threads=os.cpu_count() or 1)
    assert np.array_equal(ct, ect) and np.array_equal(body, ekey ^ msgs)
    ct2, key2 = hip.encap_batch(com, tau_g2, A, V, R, msg_len, want_gt=False)
    assert np.array_equal(ct2, ct) and np.array_equal(key2 ^ msgs, body)
    # decrypt: any (proof, ct) pair decapsulates to H(e(proof, ct)); compare with the oracle's decapsulation on the same pairs
    proofs = hip.g1_mul_batch(g1, random_fr_limbs(n, 96))
    out = hip.decrypt_batch(proofs, ct, body)
    _, dkey = oc.decap_batch(proofs, ct, msg_len, threads=os.cpu_count() or 1)
    assert np.array_equal(out, dkey ^ body)
    # the in-process group splits the items by range: same bytes
    from keaki_amd.hip import KeakiHipGroup
    g = KeakiHipGroup([0, 0, 0])
    try:
        gct, gbody = g.encrypt_batch(com, tau_g2, A, V, R, msgs)
        assert np.array_equal(gct, ct) and np.array_equal(gbody, body)
        assert np.array_equal(g.decrypt_batch(proofs, ct, body), out)
    finally:
        g.close()


def test_host_array_batches_in_chunks_equal_the_unchunked_calls(oc, hip):
    """api.hip `pipelined`: host-array batches of two chunks or more (2^16 items on the KEM side, the pairing kernel's 2^17-item launch on
    the decapsulation side) overlap their copies with the neighbouring chunks' kernels. Same bytes as the same items sent in calls that are
    below the chunking threshold, a ragged last chunk included; a sample against the oracle; outputs into FRESH pages (numpy.zeros inside
    the wrapper) with the helper threads touching them."""
    from bench import random_fr_limbs
    g1, g2 = oc.generators()
    com = hip.g1_mul_batch(g1, random_fr_limbs(1, 191))[0]
    tau_g2 = hip.g2_mul_batch(g2, random_fr_limbs(1, 192))[0]
    n = 3 * 65536 + 777                      # four chunks, the last one ragged
    A, V, R = random_fr_limbs(n, 193), random_fr_limbs(n, 194), random_fr_limbs(n, 195)
    msgs = np.random.default_rng(7).integers(0, 256, (n, 48), dtype=np.uint8)
    ct, body = hip.encrypt_batch(com, tau_g2, A, V, R, msgs)
    ect, egt, ekey = hip.encap_batch(com, tau_g2, A, V, R, 48)
    assert np.array_equal(ect, ct) and np.array_equal(ekey ^ msgs, body)
    step = 100000                            # below two chunks: one upload, one launch sequence, one download
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        c1, b1 = hip.encrypt_batch(com, tau_g2, A[lo:hi], V[lo:hi], R[lo:hi], msgs[lo:hi])
        assert np.array_equal(c1, ct[lo:hi]) and np.array_equal(b1, body[lo:hi]), lo
        c2, g2_, k2 = hip.encap_batch(com, tau_g2, A[lo:hi], V[lo:hi], R[lo:hi], 48)
        assert np.array_equal(g2_, egt[lo:hi]) and np.array_equal(k2, ekey[lo:hi]), lo
    # messages in, bodies out in ONE array: the helper threads may touch a chunk's output pages only after the chunk's inputs were read
    buf = msgs.copy()
    ct_ip, body_ip = hip.encrypt_batch(com, tau_g2, A, V, R, buf, in_place=True)
    assert body_ip is buf and np.array_equal(ct_ip, ct) and np.array_equal(buf, body)
    idx = np.array([0, 1, 65535, 65536, 131071, 131072, 196607, 196608, n - 1])
    oct_, ogt, okey = oc.encap_batch(com, tau_g2, A[idx], V[idx], R[idx], 48, threads=os.cpu_count() or 1)
    assert np.array_equal(oct_, ct[idx]) and np.array_equal(ogt, egt[idx]) and np.array_equal(okey ^ msgs[idx], body[idx])
    # the pairing side: 2 * 2^17 + 5 items = three chunks
    m = 2 * 131072 + 5
    proofs = np.ascontiguousarray(np.tile(hip.g1_mul_batch(g1, random_fr_limbs(4096, 196)), (m // 4096 + 1, 1))[:m])
    cts = np.ascontiguousarray(np.tile(ct[:8192], (m // 8192 + 1, 1))[:m])
    bodies = np.ascontiguousarray(np.tile(body[:8192], (m // 8192 + 1, 1))[:m])
    out = hip.decrypt_batch(proofs, cts, bodies)
    dgt, dkey = hip.decap_batch(proofs, cts, 48)
    assert np.array_equal(dkey ^ bodies, out)
    for lo in range(0, m, 200000):
        hi = min(m, lo + 200000)
        assert np.array_equal(hip.decrypt_batch(proofs[lo:hi], cts[lo:hi], bodies[lo:hi]), out[lo:hi]), lo
        g3, k3 = hip.decap_batch(proofs[lo:hi], cts[lo:hi], 48)
        assert np.array_equal(g3, dgt[lo:hi]) and np.array_equal(k3, dkey[lo:hi]), lo
    jdx = np.array([0, 131071, 131072, 262143, 262144, m - 1])
    _, odk = oc.decap_batch(proofs[jdx], cts[jdx], 48, threads=os.cpu_count() or 1)
    assert np.array_equal(odk ^ bodies[jdx], out[jdx])


@pytest.mark.parametrize("wide_max", [0, 1 << 20])
def test_both_pairing_kernels_at_small_and_ragged_sizes(oc, py, rand_fr, wide_max):
    """pairing_launch picks between two kernels by batch size: k_pairing (one lane pair per pairing) and pw::k_pairing_wide (twelve lanes per
    pairing, a third of the latency: a single verify / decapsulate / encapsulate, the table of a new commitment). Both forms, forced by
    `pair_wide_max`, through the pairing / KEM / stage-wise tests above and at ragged sizes (partial rows and waves of the wide kernel,
    identities in either slot), against the oracle."""
    from keaki_amd.hip import KeakiHip
    h = KeakiHip(0)
    try:
        h.set_option("pair_wide_max", wide_max)
        test_pairing_batch_vs_oracle(oc, py, h, rand_fr)
        test_encap_decap_vs_oracle(oc, py, h, rand_fr)
        test_miller_and_final_exp_stages(oc, h, rand_fr)
        g1, g2 = oc.generators()
        n = 131
        P = h.g1_mul_batch(g1, mont(oc, rand_fr(n, 131)))
        Q = h.g2_mul_batch(g2, mont(oc, rand_fr(n, 132)))
        P[7] = 0; Q[64] = 0; Q[130] = 0
        exp = oc.pairing_batch(P, Q, threads=os.cpu_count() or 1)
        for two_waves in (1, 0):          # lines on the fly: the line functions on a second wave of the workgroup (pw::k_pairing_wide2), or not
            h.set_option("pair_two_waves", two_waves)
            for m in (1, 2, 3, 4, 5, 63, 64, 65, 131):
                assert np.array_equal(h.pairing_batch(P[:m], Q[:m]), exp[:m]), (two_waves, m)
        h.set_option("pair_two_waves", 1)
        # the fixed second slot (tabulated lines of g2: the encapsulation side)
        assert np.array_equal(h.pairing_batch(P[:9], g2), oc.pairing_batch(P[:9], g2, threads=8))
        # encapsulation: the automatic policy takes the GT fixed-base path for every batch (the exponentiations and the first fill levels of a
        # table in either form); option encap_gt keeps the per-item pairing path alive below its threshold: both, small and ragged
        tau_g2 = h.g2_mul_batch(g2, mont(oc, rand_fr(1, 133)))[0]
        for gt_opt in (-1, 1 << 30):
            h.set_option("encap_gt", gt_opt)
            for it, m in enumerate((1, 5, 67, 300)):
                com = h.g1_mul_batch(g1, mont(oc, rand_fr(1, 140 + it)))[0]
                A, V, Rr = (mont(oc, rand_fr(m, 150 + 3 * it + t)) for t in range(3))
                ect, egt, ekey = oc.encap_batch(com, tau_g2, A, V, Rr, 32, threads=os.cpu_count() or 1)
                for rep in range(2):                      # new commitment, then the same one again (its table is there / widened)
                    ct, gt, key = h.encap_batch(com, tau_g2, A, V, Rr, 32)
                    assert np.array_equal(ct, ect) and np.array_equal(gt, egt) and np.array_equal(key, ekey), (gt_opt, m, rep)
    finally:
        h.close()
