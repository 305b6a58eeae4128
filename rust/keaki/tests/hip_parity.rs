//! Parity of the `hip` feature against arkworks itself -- the check that cannot run in the image the MI355X backend was built in
//! (no Rust toolchain there), and the one that closes its "parity unpinned" residual: the absolute GT value (final-exponent multiple,
//! tower basis, serialisation order) and therefore every KEM key.
//!
//!     KEAKI_HIP_LIB_DIR=/path/to/keaki_amd cargo test --release --features hip --test hip_parity -- --nocapture --test-threads=1
//!
//! No test here can pass without the GPU: every "GPU leg" runs inside `on_gpu`, which asserts that status codes came back from
//! `libkeaki_hip.so` while it ran (`hip::calls()` advanced), every "arkworks leg" inside `on_cpu` (`hip::with_disabled`: a thread-local
//! switch, no environment variable is written while tests run), which asserts the opposite. The tests also take one process-wide lock
//! (`serial()`): some of them change options of the shared `Device::global()`; `--test-threads=1` is the second belt.
//!
//! The constants below are tests/golden/bn254_vectors.json of the MI355X tree (made by its CPU oracle): if arkworks disagrees with
//! them, the ORACLE is wrong; if the GPU disagrees with arkworks, the kernels are.
#![cfg(feature = "hip")]

use ark_bn254::{Bn254, Fq, Fq2, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::{pairing::Pairing, AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::PrimeField;
use ark_poly::{univariate::DensePolynomial, DenseUVPolynomial, EvaluationDomain, Radix2EvaluationDomain};
use ark_serialize::CanonicalSerialize;
use ark_std::{ops::Mul, test_rng, UniformRand};
use keaki::{
    hip,
    kem::{decapsulate, encapsulate},
    kzg::{commit, open, open_fk, verify, KZGSetup},
    vec::{vec_commit, vec_decrypt, vec_encrypt},
};

/// serialize_uncompressed(e(G1, G2)) according to the oracle (sha-256 e109983de6d3ff0d8d4e1236dd4d91d2a313d7e7a22e3a15062b6759ad70331c)
const GT_OF_GENERATORS_HEX: &str = concat!(
    "950e879d73631f5eb5788589eb5f7ef8d63e0a28de1ba00dfe4ca9ed3f252b264a8afb8eb4349db466ed1809ea4d7c39",
    "bdab7938821f1b0a00a295c72c2de002e01dbdfd0254134efcb1ec877395d25f937719b344adb1a58d129be2d6f2a913",
    "2b16a16e8ab030b130e69c69bd20b4c45986e6744a98314b5c1a0f50faa90b04dbaf9ef8aeeee3f50be31c210b598f47",
    "52f073987f9d35be8f6770d83f2ffc0af0d18dd9d2dbcdf943825acc12a7a9ddca45e629d962c6bd64908c3930a5541c",
    "fe2924dcc5580d5cef7a4bfdec90a91b59926f850d4a7923c01a5a5dbf0f5c094a2b9fb9d415820fa6b40c59bb9eade9",
    "c953407b0fc11da350a9d872cad6d3142974ca385854afdf5f583c04231adc5957c8914b6b20dc89660ed7c3bbe7c01d",
    "972be2d53ecdb27a1bcc16ac610db95aa7d237c8ff55a898cb88645a0e32530b23d7ebf5dafdd79b0f9c2ac4ba07ce18",
    "d3d16cf36e47916c4cae5d08d3afa813972c769e8514533e380c9443b3e1ee5c96fa3a0a73f301b626454721527bf900"
);

// one KEM vector of the oracle (canonical decimal integers): tau, commitment C, point alpha, value beta, randomness r
const KEM_TAU: &str = "16108486810683800910316471678634637090532684621664899853843559589009028751094";
const KEM_COM: [&str; 2] = ["8301085058852846531532799415381687432237323807022128681561952933606273730036", "21492297623070018767944304975622711585455352824125049729415138778289051475193"];
const KEM_POINT: &str = "18421418035183923608318714001760815819998638789739981746722599328348178616250";
const KEM_VALUE: &str = "19796525744667426349873518466399876375687502633323377217699839780345057951712";
const KEM_R: &str = "9222778255153844183597133810292204042513128492245322607305955548628659646598";
const KEM_GT_HEX: &str = concat!(
    "b5951694bedc946d0cb033071ebccb74905604a0b0c5a66cda9c8ec0b2e10c0a549140177daa327d0e89b4f21beafb13",
    "b29e86fd6d4e1f2a87eff1c5cd622409115543ce297d946d44217ccb7af716346077588ce8f2c449f7e37a0bb676e31f",
    "c496512b3851851460add03bf4f0abcfbc7d63b0cafcb3b0fd90b1bf7b6ef62d430e57ebde40785394be150e2dd1ad2f",
    "94f51e4a6131bacd70d51a03f43d162db1e54bca000e2298bb3470421d549bcb6e356d30cff97fb72c80eb17972f4406",
    "4a47cee237e27b4bfa0f2104b2cdf7c36c40387931f6ef4210dcfb5f69f46c18f60fda3398158fd95ac72daf927a92ae",
    "c4344361f53c9614b9d2e3cd1605bd2e24fd3035cbddcb02f6339450936132ccf6de676e86110e5f7332d2fd8c32812f",
    "7cc69690d85e3b2294297f5e83e878d20c8088a6d5a3a057f6e2ad4948a24d27b3b8f2912973c67d62718db3ee083c6e",
    "d2b7cf4b311016dff36b1e9057fd2217042a0c386d65cf3816ee890850be5aff54630b783984b079925b4d6ffdca7100"
);
const KEM_KEY_HEX: &str = "e7f29112b1833c5614ca4457b23f10fce69eabfb19695af33781b81007d8c244";

/// One test at a time: tests change options of the process-wide `Device::global()`. A poisoned lock (an earlier test failed) is taken anyway.
static SERIAL: std::sync::Mutex<()> = std::sync::Mutex::new(());
fn serial() -> std::sync::MutexGuard<'static, ()> {
    SERIAL.lock().unwrap_or_else(|e| e.into_inner())
}
/// A GPU leg: `f` with the small-call thresholds at 0 on this thread (single calls and short vectors stay on arkworks by default,
/// `hip::active_batch`; the parity tests want them on the device) -- and the proof that it reached the library.
fn on_gpu<R>(what: &str, f: impl FnOnce() -> R) -> R {
    hip::Device::global(); // context creation + self-check happen outside the counted window
    let before = hip::calls();
    let r = hip::with_min_batch(0, f);
    assert!(hip::calls() > before, "{what}: the GPU leg never reached libkeaki_hip.so -- it ran on arkworks, the comparison would be arkworks against arkworks");
    r
}
/// An arkworks leg: `f` with the `hip` path off on this thread -- and the proof that the library was not called.
fn on_cpu<R>(what: &str, f: impl FnOnce() -> R) -> R {
    let before = hip::calls();
    let r = hip::with_disabled(f);
    assert_eq!(hip::calls(), before, "{what}: the arkworks leg called libkeaki_hip.so");
    r
}
/// puts the options a test changed back when it ends, also when it fails
struct RestoreOptions(&'static [(&'static str, i64)]);
impl Drop for RestoreOptions {
    fn drop(&mut self) {
        for &(name, value) in self.0 {
            hip::Device::global().set_option(name, value);
        }
    }
}

fn fr(s: &str) -> Fr {
    Fr::from_le_bytes_mod_order(&num_le(s))
}
fn fq(s: &str) -> Fq {
    Fq::from_le_bytes_mod_order(&num_le(s))
}
/// decimal string -> little-endian bytes (schoolbook; the test has no bignum dependency)
fn num_le(s: &str) -> Vec<u8> {
    let mut out = vec![0u8; 32];
    for ch in s.bytes() {
        let mut carry = (ch - b'0') as u32;
        for b in out.iter_mut() {
            let v = (*b as u32) * 10 + carry;
            *b = v as u8;
            carry = v >> 8;
        }
    }
    out
}
fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{x:02x}")).collect()
}
fn gt_bytes(p: G1Affine, q: G2Affine) -> Vec<u8> {
    let mut b = Vec::new();
    Bn254::pairing(p, q).serialize_uncompressed(&mut b).unwrap();
    b
}
/// a rand::Rng that replays fixed values of Fr::rand: 4 x next_u64 per draw = the limbs of the Montgomery representation
struct Replay(Vec<u64>, usize);
impl Replay {
    fn of(values: &[Fr]) -> Self {
        Replay(values.iter().flat_map(|v| v.0 .0).collect(), 0)
    }
}
impl rand::RngCore for Replay {
    fn next_u32(&mut self) -> u32 {
        self.next_u64() as u32
    }
    fn next_u64(&mut self) -> u64 {
        self.1 += 1;
        self.0[self.1 - 1]
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        for c in dest.chunks_mut(8) {
            let v = self.next_u64().to_le_bytes();
            c.copy_from_slice(&v[..c.len()]);
        }
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand::Error> {
        self.fill_bytes(dest);
        Ok(())
    }
}

#[test]
fn gt_of_generators_arkworks_oracle_gpu() {
    let _serial = serial();
    let ark = on_cpu("e(G1, G2) on arkworks", || gt_bytes(G1Affine::generator(), G2Affine::generator()));
    assert_eq!(hex(&ark), GT_OF_GENERATORS_HEX, "arkworks disagrees with the oracle's golden vector: the oracle's GT convention is wrong");
    let gpu = on_gpu("e(G1, G2) on the GPU", || hip::pairing_bytes(&[G1Affine::generator()], &[G2Affine::generator()]));
    assert_eq!(gpu, ark, "GPU pairing bytes differ from arkworks");
}

#[test]
fn pairing_batch_matches_arkworks_on_random_points() {
    let _serial = serial();
    let rng = &mut test_rng();
    let n = 64;
    let ps: Vec<G1Affine> = (0..n).map(|_| G1Affine::generator().mul(Fr::rand(rng)).into_affine()).collect();
    let mut qs: Vec<G2Affine> = (0..n).map(|_| G2Affine::generator().mul(Fr::rand(rng)).into_affine()).collect();
    qs[3] = G2Affine::identity(); // identity in a slot -> GT one
    let gpu = on_gpu("64 pairings", || hip::pairing_bytes(&ps, &qs));
    for i in 0..n {
        assert_eq!(&gpu[384 * i..384 * (i + 1)], &gt_bytes(ps[i], qs[i])[..], "item {i}");
    }
}

#[test]
fn msm_commit_open_verify_match_arkworks() {
    let _serial = serial();
    let rng = &mut test_rng();
    for &n in &[1usize, 2, 33, 129, 1 << 12] {
        let setup = KZGSetup::<Bn254>::setup(Fr::rand(rng), n);
        let p = DensePolynomial::from_coefficients_vec((0..n).map(|_| Fr::rand(rng)).collect());
        let gpu = on_gpu("commit", || commit(&setup, &p).unwrap()); // hip path
        let cpu = on_cpu("msm_unchecked", || <G1Projective as VariableBaseMSM>::msm_unchecked(setup.g1_aff(), &p.coeffs));
        assert_eq!(gpu.into_affine(), cpu.into_affine(), "commit n={n}");
        let z = Fr::rand(rng);
        let proof_cpu = on_cpu("open", || open(&setup, &p, &z).unwrap()); // arkworks path
        if n >= 2 {
            let proof = on_gpu("open", || open(&setup, &p, &z).unwrap()); // hip path: quotient + MSM on the device
            assert_eq!(proof.into_affine(), proof_cpu.into_affine(), "open n={n}");
        } else {
            // a constant polynomial: the quotient is empty, the glue may answer the identity without a device call
            assert_eq!(hip::with_min_batch(0, || open(&setup, &p, &z).unwrap()).into_affine(), proof_cpu.into_affine(), "open n={n}");
        }
        use ark_poly::Polynomial;
        let v = p.evaluate(&z);
        // arkworks says true / false ...
        assert!(on_cpu("verify", || verify(&setup, gpu, z, v, proof_cpu).unwrap()));
        assert!(!on_cpu("verify", || verify(&setup, gpu, z, v + Fr::from(1u64), proof_cpu).unwrap()));
        // ... and so does the device (keaki_hip_kzg_verify: two pairings, from the threshold of two on)
        assert!(on_gpu("verify", || verify(&setup, gpu, z, v, proof_cpu).unwrap()));
        assert!(!on_gpu("verify", || verify(&setup, gpu, z, v + Fr::from(1u64), proof_cpu).unwrap()));
    }
}

/// The host-pointer `commit` / `open` upload their scalars in point-range chunks under the kernels (round 5: `keaki_hip_msm_g1` /
/// `keaki_hip_kzg_open`, automatic from 2^20 scalars on). Forced here at a test-sized polynomial (`msm_pipe_chunks` = 5), with the library's
/// host-side helpers switched off (`host_prefault` = 0, `pipe_chunks` = 0) and on: the same points as arkworks either way.
#[test]
fn chunked_commit_and_open_and_the_helper_off_switches_match_arkworks() {
    let _serial = serial();
    let rng = &mut test_rng();
    let n = (1usize << 14) + 17; // above the 8,192 coefficients up to which the glue forms the quotient on the host: keaki_hip_kzg_open itself
    let setup = KZGSetup::<Bn254>::setup(Fr::rand(rng), n);
    let p = DensePolynomial::from_coefficients_vec((0..n).map(|_| Fr::rand(rng)).collect());
    let z = Fr::rand(rng);
    // arkworks directly (no switch involved): the MSM of the coefficients, and the proof through the reference's own body of `open`
    let cpu = on_cpu("msm_unchecked", || <G1Projective as VariableBaseMSM>::msm_unchecked(setup.g1_aff(), &p.coeffs).into_affine());
    let proof_cpu = on_cpu("open", || open(&setup, &p, &z).unwrap().into_affine());
    let dev = keaki::hip::Device::global();
    let _restore = RestoreOptions(&[("msm_pipe_chunks", -1), ("host_prefault", 1), ("pipe_chunks", 1)]);
    for &(chunks, prefault, pipe) in &[(5i64, 1i64, 1i64), (5, 0, 1), (0, 0, 0), (-1, 1, 1)] {
        dev.set_option("msm_pipe_chunks", chunks);
        dev.set_option("host_prefault", prefault);
        dev.set_option("pipe_chunks", pipe);
        assert_eq!(on_gpu("commit", || commit(&setup, &p).unwrap()).into_affine(), cpu, "commit chunks={chunks} prefault={prefault} pipe={pipe}");
        assert_eq!(on_gpu("open", || open(&setup, &p, &z).unwrap()).into_affine(), proof_cpu, "open chunks={chunks} prefault={prefault} pipe={pipe}");
    }
}

/// In-process multi-GPU (`KEAKI_HIP_DEVICES=4`, or `0,0,0` to put three contexts on one GPU): `commit` / `open` are then the sharded
/// MSM of `hip::ShardedCommit` -- every member sums its range of the SRS, the partials are added -- and must still equal arkworks.
/// Run the whole file once more with the variable set: every test above then goes through the device group as well.
#[test]
fn sharded_commit_over_a_device_group_matches_arkworks() {
    let _serial = serial();
    let rng = &mut test_rng();
    let n = 1usize << 12;
    let setup = KZGSetup::<Bn254>::setup(Fr::rand(rng), n);
    let srs = hip::HipSrs::upload(setup.g1_aff());
    let Some(sharded) = hip::ShardedCommit::new(&srs) else {
        eprintln!("single-GPU device (KEAKI_HIP_DEVICES unset): sharded commit not exercised");
        return;
    };
    assert!(sharded.members() > 1);
    for &len in &[0usize, 1, 5, n / 3, n - 1, n] {
        // shorter than the SRS: members whose range lies beyond the polynomial contribute the identity
        let coeffs: Vec<Fr> = (0..len).map(|_| Fr::rand(rng)).collect();
        let cpu = on_cpu("msm_unchecked", || <G1Projective as VariableBaseMSM>::msm_unchecked(&setup.g1_aff()[..len], &coeffs));
        let gpu = if len == 0 { sharded.commit(&coeffs) } else { on_gpu("sharded commit", || sharded.commit(&coeffs)) };
        assert_eq!(gpu.into_affine(), cpu.into_affine(), "sharded commit len={len}");
    }
    let p = DensePolynomial::from_coefficients_vec((0..n).map(|_| Fr::rand(rng)).collect());
    let z = Fr::rand(rng);
    let proof_cpu = on_cpu("open", || open(&setup, &p, &z).unwrap());
    assert_eq!(on_gpu("sharded open", || sharded.open(&p.coeffs, &z)).into_affine(), proof_cpu.into_affine(), "sharded open");
}

#[test]
fn kem_vector_of_the_oracle_arkworks_and_gpu() {
    let _serial = serial();
    let tau = fr(KEM_TAU);
    let setup = KZGSetup::<Bn254>::setup(tau, 8);
    let com: G1Projective = G1Affine::new(fq(KEM_COM[0]), fq(KEM_COM[1])).into();
    let (point, value, r) = (fr(KEM_POINT), fr(KEM_VALUE), fr(KEM_R));
    // arkworks path with r replayed
    let (ct_cpu, key_cpu) = on_cpu("encapsulate", || encapsulate::<Bn254>(&mut Replay::of(&[r]), &setup, com, point, value, 32));
    assert_eq!(hex(&key_cpu), KEM_KEY_HEX, "arkworks disagrees with the oracle's KEM key");
    let secret = on_cpu("pairing", || gt_bytes((com - G1Affine::generator().mul(value)).mul(r).into_affine(), G2Affine::generator()));
    assert_eq!(hex(&secret), KEM_GT_HEX);
    // GPU path, same r
    let (ct_gpu, key_gpu) = on_gpu("encapsulate", || encapsulate::<Bn254>(&mut Replay::of(&[r]), &setup, com, point, value, 32));
    assert_eq!(ct_gpu.into_affine(), ct_cpu.into_affine());
    assert_eq!(key_gpu, key_cpu);
    // and back: one decapsulation on the device (it stays on arkworks by default; the threshold is 0 inside on_gpu) and one on arkworks
    let proof = on_cpu("open", || {
        // any polynomial with p(point) = value committed to `com` would do; the KEM only needs e(proof, ct) == e(com - value g1, g2)^r, so
        // take proof = (com - value g1) / (tau - point)
        (com - G1Affine::generator().mul(value)).mul((tau - point).inverse().unwrap())
    });
    use ark_ff::Field;
    let key_dec_cpu = on_cpu("decapsulate", || decapsulate::<Bn254>(proof, ct_cpu, 32));
    let key_dec_gpu = on_gpu("decapsulate", || decapsulate::<Bn254>(proof, ct_cpu, 32));
    assert_eq!(key_dec_cpu, key_cpu, "arkworks decapsulation recovers the key");
    assert_eq!(key_dec_gpu, key_cpu, "GPU decapsulation recovers the key");
}

#[test]
fn vec_flow_matches_serial_arkworks_loop() {
    let _serial = serial();
    // Laconic-OT shaped: vec_commit, vec_encrypt to "value i == b", vec_decrypt; GPU path and arkworks path from the same rng seed
    let n = 15usize; // domain 16
    let flow = || {
        let rng = &mut test_rng();
        let setup = KZGSetup::<Bn254>::setup(Fr::rand(rng), 16);
        let v: Vec<Fr> = (0..n).map(|i| Fr::from((i % 2) as u64)).collect();
        let (com, proofs) = vec_commit(rng, &setup, &v).unwrap();
        let dom = Radix2EvaluationDomain::<Fr>::new(n + 1).unwrap();
        let points: Vec<Fr> = dom.elements().collect();
        let msgs: Vec<Vec<u8>> = (0..n).map(|i| vec![i as u8; 1 + (i % 5) * 7]).collect(); // ragged lengths
        let refs: Vec<&[u8]> = msgs.iter().map(|m| m.as_slice()).collect();
        let cts = vec_encrypt(rng, &setup, com, &points, &v, &refs);
        let ct_refs: Vec<_> = cts.iter().collect();
        let dec = vec_decrypt::<Bn254>(&proofs, &ct_refs);
        (com.into_affine(), G1Projective::normalize_batch(&proofs), cts.iter().map(|c| (c.0.into_affine(), c.1.clone())).collect::<Vec<_>>(), dec, msgs)
    };
    let (c1, p1, e1, d1, m1) = on_gpu("vec flow", flow);
    let (c0, p0, e0, d0, _) = on_cpu("vec flow", flow);
    assert_eq!(c1, c0);
    assert_eq!(p1, p0, "FK23 proofs");
    assert_eq!(e1, e0, "ciphertexts");
    assert_eq!(d1, d0);
    assert_eq!(d1, m1, "messages recovered");
    let _ = (open_fk::<Bn254>, decapsulate::<Bn254>, Fq2::new(Fq::from(0u64), Fq::from(0u64)), G2Projective::default());
}

#[test]
fn vec_flow_equal_length_messages_device_dem_matches_serial_arkworks_loop() {
    let _serial = serial();
    // Laconic OT's own shape (tests/laconic_ot.rs:121-124): 32-byte messages of equal length -> keaki_hip_encrypt_batch / _decrypt_batch, the XOR of
    // src/enc.rs:32-36 / :48-52 on the device behind the KDF (with KEAKI_HIP_DEVICES set: the group variants, by item range)
    let n = 15usize;
    let flow = || {
        let rng = &mut test_rng();
        let setup = KZGSetup::<Bn254>::setup(Fr::rand(rng), 16);
        let v: Vec<Fr> = (0..n).map(|i| Fr::from((i % 2) as u64)).collect();
        let (com, proofs) = vec_commit(rng, &setup, &v).unwrap();
        let dom = Radix2EvaluationDomain::<Fr>::new(n + 1).unwrap();
        let points: Vec<Fr> = dom.elements().collect();
        let msgs: Vec<Vec<u8>> = (0..n).map(|i| (0..32u8).map(|j| j.wrapping_mul(i as u8 + 3)).collect()).collect();
        let refs: Vec<&[u8]> = msgs.iter().map(|m| m.as_slice()).collect();
        let cts = vec_encrypt(rng, &setup, com, &points, &v, &refs);
        let ct_refs: Vec<_> = cts.iter().collect();
        let dec = vec_decrypt::<Bn254>(&proofs, &ct_refs);
        (cts.iter().map(|c| (c.0.into_affine(), c.1.clone())).collect::<Vec<_>>(), dec, msgs)
    };
    let (e1, d1, m1) = on_gpu("vec flow, equal lengths", flow);
    let (e0, d0, _) = on_cpu("vec flow, equal lengths", flow);
    assert_eq!(e1, e0, "ciphertext points and bodies");
    assert_eq!(d1, d0);
    assert_eq!(d1, m1, "messages recovered");
}
