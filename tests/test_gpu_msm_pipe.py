"""The host-pointer MSM in point-range CHUNKS (csrc/api.hip: msm_from_host; csrc/msm_host.hip.h: MsmPipe): the call keaki makes --
kzg::commit hands over a polynomial in host memory (reference src/kzg.rs:89-101) -- uploads chunk j + 1 while the kernels of chunk j
run, every bucket pass going on from the state the earlier chunks left. Whatever the cut, the result is the same group element:
chunked == unchunked == oracle, G1 and G2, with and without window tables, in the lazy-limb and the saturated bucket kernels, with the
same / opposite point meeting a bucket's stored state across a chunk boundary, through the heavy-bucket path, and at 2^24 terms."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _aff(j):
    from keaki_amd.hip import jac_to_affine_words
    return jac_to_affine_words(j)


def _mont(oc, ints):
    return oc.fr_to_mont(oc.ints_to_limbs(ints))


@pytest.fixture()
def piped(hip):
    """the session context with the chunking options under the test's control; automatic again afterwards"""
    yield hip
    for k, v in (("msm_pipe_chunks", -1), ("msm_pipe_growth", 160), ("msm_pipe_min", 1 << 20), ("acc_u29", 1), ("acc_u29_g2", 1), ("acc_idxq", 1)):
        hip.set_option(k, v)


def _edge_instance(oc, hip, rand_fr, n, seed):
    """random pairs with the edge set of SURVEY 8d mixed in: zero scalars, r - 1, identity points, repeated points, equal scalars"""
    from conftest import R_MOD
    g1, _ = oc.generators()
    k = rand_fr(n, seed)
    s = rand_fr(n, seed + 1)
    for i in range(0, n, 11):
        s[i] = 0
    for i in range(5, n, 13):
        s[i] = R_MOD - 1
    for i in range(7, n, 17):
        k[i] = k[i - 1]                                    # repeated point ...
        s[i] = s[i - 1]                                    # ... with the same scalar: the doubling branch
    pts = hip.g1_mul_batch(g1, _mont(oc, k))
    for i in range(3, n, 19):
        pts[i] = 0                                         # identity
    return pts, _mont(oc, s)


@pytest.mark.parametrize("n", [2, 3, 33, 257, 1000, 4096, 70001])
def test_chunked_g1_equals_oracle_small(oc, piped, rand_fr, n):
    hip = piped
    pts, s = _edge_instance(oc, hip, rand_fr, n, 9000 + n)
    exp = oc.msm_g1(pts, s, threads=os.cpu_count() or 1)
    srs = hip.srs_g1_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g1_precompute(srs)
            for u29 in (1, 0):
                hip.set_option("acc_u29", u29)
                for chunks, growth in ((0, 140), (2, 100), (3, 140), (5, 100), (7, 250)):
                    hip.set_option("msm_pipe_chunks", chunks)
                    hip.set_option("msm_pipe_growth", growth)
                    assert np.array_equal(_aff(hip.msm_g1(srs, s)), exp), (n, tables, u29, chunks, growth)
    finally:
        srs.free()


@pytest.mark.parametrize("n", [2, 5, 64, 1000, 5000])
def test_chunked_g2_equals_oracle_small(oc, piped, rand_fr, n):
    hip = piped
    _, g2 = oc.generators()
    k = rand_fr(n, 9500 + n)
    s = rand_fr(n, 9501 + n)
    s[0] = 0
    if n > 3:
        k[3] = k[2]; s[3] = s[2]
    pts = hip.g2_mul_batch(g2, _mont(oc, k))
    if n > 4:
        pts[4] = 0
    sm = _mont(oc, s)
    exp = oc.msm_g2(pts, sm)
    srs = hip.srs_g2_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g2_precompute(srs)
            for u29 in (1, 0):
                hip.set_option("acc_u29_g2", u29)
                for chunks in (0, 2, 3, 5):
                    hip.set_option("msm_pipe_chunks", chunks)
                    got = hip.msm_g2(srs, sm)
                    from keaki_amd.hip import jac_to_affine_words
                    assert np.array_equal(jac_to_affine_words(got), exp), (n, tables, u29, chunks)
    finally:
        srs.free()


def test_same_and_opposite_point_meet_the_stored_bucket_across_chunks(oc, piped, rand_fr):
    """chunk 0 leaves P in bucket 4 and Q in bucket 6; chunk 1 brings P again (the doubling branch on a LOADED state), -Q (the bucket
    empties) and then more points into the emptied bucket; chunk 2 brings 2P's negative (state 2P + (-2P): empty in the last pass)."""
    hip = piped
    g1, _ = oc.generators()
    base = hip.g1_mul_batch(g1, _mont(oc, rand_fr(4, 4242)))        # P, Q, R, S
    P, Q, Rr, S = base
    P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583

    def neg(a):                                                      # (x, -y): Montgomery residue of -y is p - (y R mod p)
        y = int.from_bytes(a[4:8].tobytes(), "little")
        out = a.copy()
        out[4:8] = np.frombuffer(((P_MOD - y) % P_MOD).to_bytes(32, "little"), np.uint64)
        return out
    twoP = hip.g1_mul_batch(P.reshape(1, 8), _mont(oc, [2]))[0]
    pts = np.stack([P, Q, Rr, S,   P, neg(Q), Rr, S,   neg(twoP), Rr, S, Q])
    sc = [5, 7, 9, 11,   5, 7, 7, 7,   5, 9, 7, 7]
    s = _mont(oc, sc)
    exp = oc.msm_g1(pts, s)
    srs = hip.srs_g1_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g1_precompute(srs)
            for u29 in (1, 0):
                hip.set_option("acc_u29", u29)
                for chunks in (0, 3, 2, 12):
                    hip.set_option("msm_pipe_chunks", chunks)
                    hip.set_option("msm_pipe_growth", 100)
                    assert np.array_equal(_aff(hip.msm_g1(srs, s)), exp), (tables, u29, chunks)
    finally:
        srs.free()


def test_chunked_heavy_buckets(oc, piped):
    """structured scalars put more than HEAVY_MIN = 8192 pairs of ONE chunk into single buckets: the sliced path adds into the stored
    state (Acc29 in the lazy-limb passes, the canonical bucket in the last one and in the saturated kernel); a bucket may be heavy in
    one chunk and ordinary in the next (the first chunk holds ones only, the later ones a mix)."""
    hip = piped
    n = 1 << 17
    g1, _ = oc.generators()
    rng = np.random.default_rng(171)
    k = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    pts = hip.g1_mul_batch(g1, k)
    vals = np.where(rng.integers(0, 4, n) != 0, 1, rng.integers(2, 300, n)).astype(np.uint64)
    vals[: n // 4] = 1                                                     # chunk 0: one bucket only
    vals[n // 2: n // 2 + 30000] = rng.integers(2, 9, 30000)              # a stretch that is heavy for other buckets and light for bucket 1
    sc = oc.fr_to_mont(np.concatenate([vals[:, None], np.zeros((n, 3), np.uint64)], 1))
    exp = oc.g1_mul_batch(g1, oc.fr_dot(sc, k).reshape(1, 4))[0]
    srs = hip.srs_g1_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g1_precompute(srs)
            for u29 in (1, 0):
                hip.set_option("acc_u29", u29)
                for chunks in (0, 2, 4):
                    hip.set_option("msm_pipe_chunks", chunks)
                    hip.set_option("msm_pipe_growth", 100)
                    assert np.array_equal(_aff(hip.msm_g1(srs, sc)), exp), (tables, u29, chunks)
    finally:
        srs.free()


def _gen_points_dev(hip, torch, dev, k_host, g2=False):
    from bench import mont_words, G2_GEN
    words = mont_words(1) + mont_words(2)
    if g2:
        words = []
        for c in G2_GEN:
            words += mont_words(c)
    d_gen = torch.from_numpy(np.array(words, np.uint64).view(np.int64)).to(dev)
    d_k = torch.from_numpy(k_host.view(np.int64)).to(dev)
    d_pts = torch.empty((k_host.shape[0], 16 if g2 else 8), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    (hip.g2_mul_batch_dev if g2 else hip.g1_mul_batch_dev)(d_gen.data_ptr(), 0, d_k.data_ptr(), k_host.shape[0], d_pts.data_ptr())
    hip.synchronize()
    return d_pts


def test_config2_2p20_chunked_equals_unchunked_equals_oracle(oc, piped):
    """BASELINE config 2 through the host-pointer entry: the automatic chunking (2^20 scalars: on), forced 4 and 8 chunks, and one
    copy in front all give the oracle's point; tables and no tables"""
    import torch
    from bench import random_fr_limbs, SEED
    hip = piped
    dev = torch.device("cuda", 0)
    n = 1 << 20
    k = random_fr_limbs(n, SEED + 1)
    s = random_fr_limbs(n, SEED + 104729)
    d_pts = _gen_points_dev(hip, torch, dev, k)
    pts = d_pts.cpu().numpy().view(np.uint64)
    exp = oc.msm_g1(pts, s, threads=os.cpu_count() or 1)
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        for tables in (False, True):
            if tables:
                assert hip.srs_g1_precompute(srs) > 0
            for chunks in (-1, 0, 4, 8):
                hip.set_option("msm_pipe_chunks", chunks)
                assert np.array_equal(_aff(hip.msm_g1(srs, s)), exp), (tables, chunks)
    finally:
        srs.free()


def test_g2_2p20_chunked_identity(oc, piped):
    """G2 at 2^20 points through the host-pointer entry, chunked and not, tables and no tables: MSM(s, k_i g2) == (sum s_i k_i) g2"""
    import torch
    from bench import random_fr_limbs
    from keaki_amd.hip import jac_to_affine_words
    hip = piped
    dev = torch.device("cuda", 0)
    n = 1 << 20
    k = random_fr_limbs(n, 0x62A)
    s = random_fr_limbs(n, 0x62B)
    _, g2 = oc.generators()
    exp = oc.g2_mul_batch(g2, oc.fr_dot(s, k).reshape(1, 4))[0]
    d_pts = _gen_points_dev(hip, torch, dev, k, g2=True)
    srs = hip.srs_g2_wrap_dev(d_pts.data_ptr(), n)
    try:
        for tables in (False, True):
            if tables:
                assert hip.srs_g2_precompute(srs) > 0
            for chunks in (-1, 0, 3):
                hip.set_option("msm_pipe_chunks", chunks)
                assert np.array_equal(jac_to_affine_words(hip.msm_g2(srs, s)), exp), (tables, chunks)
    finally:
        srs.free()


def test_headline_2p24_through_the_host_pointer_entry(oc, piped):
    """the BASELINE headline size as keaki calls it: 2^24 scalars in (pageable) host memory, chunked upload under the kernels, window
    tables; by the O(n) identity. The same call with one copy in front gives the same bytes."""
    import torch
    from bench import random_fr_limbs, SEED
    hip = piped
    dev = torch.device("cuda", 0)
    n = 1 << 24
    k = random_fr_limbs(n, SEED + 1)
    s = random_fr_limbs(n, SEED + 104729)
    g1, _ = oc.generators()
    exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k).reshape(1, 4))[0]
    d_pts = _gen_points_dev(hip, torch, dev, k)
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        assert hip.srs_g1_precompute(srs) > 0
        a = hip.msm_g1(srs, s)
        assert np.array_equal(_aff(a), exp)
        hip.set_option("msm_pipe_chunks", 0)
        assert np.array_equal(hip.msm_g1(srs, s), a)
        hip.set_option("msm_pipe_chunks", 16)
        hip.set_option("msm_pipe_growth", 100)
        assert np.array_equal(hip.msm_g1(srs, s), a)
    finally:
        srs.free()
        del d_pts
        torch.cuda.empty_cache()


@pytest.mark.parametrize("bits,log2n", [(11, 22), (20, 21), (34, 20)])
def test_skewed_scalars_through_the_chunked_entry(oc, piped, bits, log2n):
    """the sort's slow paths (tests/test_gpu_baseline_sizes.py: skewed digits) chunk by chunk: 11-bit scalars at 2^22 put 1,800 pairs
    per bucket into 2,048 buckets in every chunk; with and without tables; by the O(n) identity"""
    import torch
    from bench import random_fr_limbs
    hip = piped
    dev = torch.device("cuda", 0)
    n = 1 << log2n
    k = random_fr_limbs(n, 0x5CE0 + bits)
    rng = np.random.default_rng(bits)
    vals = rng.integers(0, 1 << bits, n, dtype=np.uint64)
    vals[::7] = 0
    canon = np.zeros((n, 4), np.uint64); canon[:, 0] = vals
    s = oc.fr_to_mont(canon)
    g1, _ = oc.generators()
    exp = oc.g1_mul_batch(g1, oc.fr_dot(s, k).reshape(1, 4))[0]
    d_pts = _gen_points_dev(hip, torch, dev, k)
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        for tables in (False, True):
            if tables:
                assert hip.srs_g1_precompute(srs) > 0
            for chunks in (-1, 5):
                hip.set_option("msm_pipe_chunks", chunks)
                assert np.array_equal(_aff(hip.msm_g1(srs, s)), exp), (bits, tables, chunks)
    finally:
        srs.free()
        del d_pts
        torch.cuda.empty_cache()


@pytest.mark.parametrize("prefault,chunks", [(0, 1), (1, 0), (0, 0)])
def test_host_helper_off_switches_give_identical_results(oc, piped, prefault, chunks):
    """options host_prefault = 0 (no helper threads, no first-touch writes, no madvise on caller memory) and pipe_chunks = 0 (upload, kernels,
    download in that order): the same bytes as the defaults for a four-chunk encrypt_batch, a three-chunk decrypt_batch and a 2^20-scalar MSM
    from a host array (include/keaki_hip.h: keaki_hip_ctx_set_option)"""
    from bench import random_fr_limbs
    hip = piped
    g1, g2 = oc.generators()
    com = hip.g1_mul_batch(g1, random_fr_limbs(1, 291))[0]
    tau_g2 = hip.g2_mul_batch(g2, random_fr_limbs(1, 292))[0]
    n = 3 * 65536 + 777
    A, V, R = random_fr_limbs(n, 293), random_fr_limbs(n, 294), random_fr_limbs(n, 295)
    msgs = np.random.default_rng(9).integers(0, 256, (n, 40), dtype=np.uint8)
    m = 2 * 131072 + 5
    nm = 1 << 20
    pts = hip.g1_mul_batch(g1, random_fr_limbs(nm, 296))
    sc = random_fr_limbs(nm, 297)
    srs = hip.srs_g1_upload(pts)
    try:
        def run():
            ct, body = hip.encrypt_batch(com, tau_g2, A, V, R, msgs)
            proofs = np.ascontiguousarray(np.tile(pts[:4096], (m // 4096 + 1, 1))[:m])
            cts = np.ascontiguousarray(np.tile(ct[:8192], (m // 8192 + 1, 1))[:m])
            bodies = np.ascontiguousarray(np.tile(body[:8192], (m // 8192 + 1, 1))[:m])
            return ct, body, hip.decrypt_batch(proofs, cts, bodies), hip.msm_g1(srs, sc)
        ref = run()
        hip.set_option("host_prefault", prefault)
        hip.set_option("pipe_chunks", chunks)
        try:
            got = run()
        finally:
            hip.set_option("host_prefault", 1)
            hip.set_option("pipe_chunks", 1)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
        idx = np.array([0, 65535, 65536, n - 1])
        oct_, _, okey = oc.encap_batch(com, tau_g2, A[idx], V[idx], R[idx], 40, threads=os.cpu_count() or 1)
        assert np.array_equal(oct_, got[0][idx]) and np.array_equal(okey ^ msgs[idx], got[1][idx])
    finally:
        srs.free()


@pytest.mark.parametrize("n", [2, 3, 5, 34, 1000, 1057, 33000, 70000])
def test_chunked_open_from_the_top_equals_oracle(oc, piped, rand_fr, n):
    """keaki_hip_kzg_open with the coefficients uploaded in chunks from the TOP (api.hip: the quotient's recurrence runs downwards, every chunk
    starts from the carry the chunk above left; Horner blocks of 32): value and proof against the oracle's quotient + MSM for forced chunk
    counts, chunk sizes of one coefficient included; z = 0 and a zero top coefficient; with and without window tables"""
    hip = piped
    g1, _ = oc.generators()
    pts = hip.g1_mul_batch(g1, _mont(oc, rand_fr(max(n, 2), 8800 + n)))
    c = rand_fr(n, 8801 + n)
    if n > 4:
        c[1] = 0; c[n // 2] = 0
    zs = [rand_fr(1, 8802 + n)[0], 0]
    cm = _mont(oc, c)
    srs = hip.srs_g1_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g1_precompute(srs)
            for z in zs:
                zm = _mont(oc, [z])[0]
                q, v = oc.fr_quotient(cm, zm)
                exp = oc.msm_g1(pts[:n - 1], q)
                for chunks, growth in ((0, 140), (2, 100), (3, 140), (7, 100), (n, 100)):
                    hip.set_option("msm_pipe_chunks", chunks)
                    hip.set_option("msm_pipe_growth", growth)
                    proof, val = hip.kzg_open(srs, cm, zm)
                    assert np.array_equal(val.reshape(-1), v.reshape(-1)), (n, z != 0, tables, chunks)
                    assert np.array_equal(_aff(proof), exp), (n, z != 0, tables, chunks)
    finally:
        srs.free()


def test_open_2p22_chunked_equals_unchunked(oc, piped):
    """`open` at 2^22 coefficients from (pageable) host memory: automatic chunking (six chunks from the top) gives the bytes of the call with one
    copy in front; the proof equals the oracle's at the first 2^16 ... no: the O(n) check -- proof == MSM of the oracle's quotient by the identity
    MSM(q, k_i G) == (sum q_i k_i) G"""
    import torch
    from bench import random_fr_limbs
    hip = piped
    dev = torch.device("cuda", 0)
    n = 1 << 22
    k = random_fr_limbs(n, 0x0BE1)
    c = random_fr_limbs(n, 0x0BE2)
    z = random_fr_limbs(1, 0x0BE3)[0]
    d_pts = _gen_points_dev(hip, torch, dev, k)
    srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
    try:
        assert hip.srs_g1_precompute(srs) > 0
        q, v = oc.fr_quotient(c, z)
        g1, _ = oc.generators()
        exp = oc.g1_mul_batch(g1, oc.fr_dot(q, k[:n - 1]).reshape(1, 4))[0]
        proof, val = hip.kzg_open(srs, c, z)
        assert np.array_equal(val.reshape(-1), v.reshape(-1)) and np.array_equal(_aff(proof), exp)
        hip.set_option("msm_pipe_chunks", 0)
        proof0, val0 = hip.kzg_open(srs, c, z)
        assert np.array_equal(proof0, proof) and np.array_equal(val0, val)
    finally:
        srs.free()
        del d_pts
        torch.cuda.empty_cache()


def test_a_chunked_call_that_fails_leaves_the_context_usable(oc, rand_fr):
    """a chunked host-pointer MSM / open whose workspace cannot be had (keaki_hip_debug_set_alloc_limit) returns KEAKI_ERR_OOM before any pass
    runs, no copy is left reading the caller's array (the copy stream is drained on every exit), and the same context then gives the right
    answer -- with the limit lifted, and in chunks again"""
    from keaki_amd.hip import KeakiHip, KeakiHipError
    h = KeakiHip(0)
    try:
        n = 70000
        g1, _ = oc.generators()
        pts = h.g1_mul_batch(g1, _mont(oc, rand_fr(n, 7700)))
        s = _mont(oc, rand_fr(n, 7701))
        z = _mont(oc, rand_fr(1, 7702))[0]
        exp = oc.msm_g1(pts, s, threads=os.cpu_count() or 1)
        q, v = oc.fr_quotient(s, z)
        exp_open = oc.msm_g1(pts[:n - 1], q, threads=os.cpu_count() or 1)
        srs = h.srs_g1_upload(pts)
        h.set_option("msm_pipe_chunks", 4)
        h.debug_set_alloc_limit(1 << 16)                      # every workspace of a 70000-term MSM is larger
        for call in (lambda: h.msm_g1(srs, s), lambda: h.kzg_open(srs, s, z)):
            with pytest.raises(KeakiHipError) as e:
                call()
            assert e.value.status == -3
        h.debug_set_alloc_limit(0)
        assert np.array_equal(_aff(h.msm_g1(srs, s)), exp)
        proof, val = h.kzg_open(srs, s, z)
        assert np.array_equal(_aff(proof), exp_open) and np.array_equal(val.reshape(-1), v.reshape(-1))
        srs.free()
    finally:
        h.close()


@pytest.mark.parametrize("nb,length", [(7, 3000), (64, 468), (15, 2000), (3, 8000), (1, 5000)])
def test_index_groups_when_neighbouring_lanes_step_over_empty_segments(oc, piped, nb, length):
    """Round 6: the G1 / G2 bucket kernels read their index stream by aligned 64-byte groups through a lane-private LDS slot (csrc/msm.hip.h segq_*).
    The case that broke the first version: `nb` buckets own a stretch of nb x length entries, so in one chunk of their bin the other buckets' segments
    are EMPTY -- lanes of one wave leave the walker's segment loop after different trip counts (hipcc took a lane mask from the loop's last trip
    only; see segq_next). Every variant must give the oracle's point: groups on / off (acc_idxq), tables or none, resident and in chunks, G1 and G2."""
    hip = piped
    n = 1 << 17
    g1, g2 = oc.generators()
    rng = np.random.default_rng(1000 * nb + length)
    k = rng.integers(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    vals = rng.integers(300, 1000, n)
    st = np.repeat(np.arange(2, 2 + nb), length)
    rng.shuffle(st)
    vals[n // 2: n // 2 + nb * length] = st
    sc = oc.fr_to_mont(np.concatenate([vals[:, None].astype(np.uint64), np.zeros((n, 3), np.uint64)], 1))
    dot = oc.fr_dot(sc, k).reshape(1, 4)
    exp = oc.g1_mul_batch(g1, dot)[0]
    pts = hip.g1_mul_batch(g1, k)
    srs = hip.srs_g1_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g1_precompute(srs)
            for idxq in (1, 0):
                hip.set_option("acc_idxq", idxq)
                for chunks in (0, 3):
                    hip.set_option("msm_pipe_chunks", chunks)
                    hip.set_option("msm_pipe_growth", 100)
                    assert np.array_equal(_aff(hip.msm_g1(srs, sc)), exp), (tables, idxq, chunks)
    finally:
        srs.free()
    m = 1 << 14                                            # G2: the same shape on a smaller instance (the oracle forms the expected point by one scalar-mult)
    cut = max(1, length * m // n)
    vals2 = rng.integers(300, 1000, m)
    st2 = np.repeat(np.arange(2, 2 + nb), cut)
    rng.shuffle(st2)
    vals2[m // 2: m // 2 + nb * cut] = st2
    sc2 = oc.fr_to_mont(np.concatenate([vals2[:, None].astype(np.uint64), np.zeros((m, 3), np.uint64)], 1))
    pts2 = hip.g2_mul_batch(g2, k[:m])
    exp2 = oc.g2_mul_batch(g2, oc.fr_dot(sc2, k[:m]).reshape(1, 4))[0]
    srs2 = hip.srs_g2_upload(pts2)
    try:
        for chunks in (0, 3):
            hip.set_option("msm_pipe_chunks", chunks)
            assert np.array_equal(_aff(hip.msm_g2(srs2, sc2)), exp2), chunks
    finally:
        srs2.free()


def test_chunked_call_without_room_for_the_parked_registers_falls_back(oc, rand_fr):
    """the chunked G1 call parks the bucket kernel's registers between passes in a 144-B-per-bucket workspace (Acc29) -- optional memory.
    When that one allocation fails (keaki_hip_debug_set_alloc_limit between the canonical buckets' 128 B and the 144 B per bucket) the passes go on
    from the canonical bucket through the saturated kernel: same point as the oracle, no KEAKI_ERR_OOM; with the limit lifted the workspace appears."""
    from keaki_amd.hip import KeakiHip
    hip = KeakiHip(0)                                      # a context of its own: its workspaces start empty
    try:
        n = 70001
        pts, s = _edge_instance(oc, hip, rand_fr, n, 424242)
        exp = oc.msm_g1(pts, s, threads=os.cpu_count() or 1)
        srs = hip.srs_g1_upload(pts)
        hip.srs_g1_precompute(srs)
        hip.set_option("msm_pipe_chunks", 0)
        assert np.array_equal(_aff(hip.msm_g1(srs, s)), exp)
        nb = 1 << (hip.last_msm_stats()["window_bits"] - 1)             # shared buckets of the table path
        hip.set_option("msm_pipe_chunks", 3)
        hip.debug_set_alloc_limit(nb * 136)
        assert np.array_equal(_aff(hip.msm_g1(srs, s)), exp), "fallback to the canonical-state passes"
        held = hip.memory()["workspaces"]
        hip.debug_set_alloc_limit(0)
        assert np.array_equal(_aff(hip.msm_g1(srs, s)), exp)
        assert hip.memory()["workspaces"] - held >= nb * 144, "with room, the parked-register workspace is taken"
        srs.free()
    finally:
        hip.close()


@pytest.mark.parametrize("n", [64, 1000])
def test_g2_repeated_points_double_the_accumulator(oc, piped, rand_fr, n):
    """Round 6: when the incoming G2 point equals the bucket's accumulator the mixed addition doubles the ACCUMULATOR in the lazy limbs
    (x29g2_dbl_of_acc) instead of sending the affine point through the saturated formulas. A handful of distinct points, each many times with
    scalars from a small set: buckets see Q, Q (-> 2Q), Q (2Q + Q), Q, -Q ... in every window. Same for the fixed-base G2 sums of encapsulate,
    which share the addition (covered by the KEM parity tests)."""
    hip = piped
    _, g2 = oc.generators()
    base_k = rand_fr(3, 7700 + n)
    base_s = rand_fr(4, 7701 + n)
    from conftest import R_MOD
    k = [base_k[i % 3] for i in range(n)]
    s = [base_s[(i // 3) % 4] if i % 7 else (R_MOD - base_s[(i // 3) % 4]) for i in range(n)]      # every seventh: the opposite scalar = the opposite point in every window
    pts = hip.g2_mul_batch(g2, _mont(oc, k))
    sm = _mont(oc, s)
    exp = oc.msm_g2(pts, sm)
    srs = hip.srs_g2_upload(pts)
    try:
        for tables in (False, True):
            if tables:
                hip.srs_g2_precompute(srs)
            for chunks in (0, 3):
                hip.set_option("msm_pipe_chunks", chunks)
                assert np.array_equal(_aff(hip.msm_g2(srs, sm)), exp), (n, tables, chunks)
    finally:
        srs.free()
