//! `hip` feature of keaki: the BN254 hot path on an AMD MI355X through `libkeaki_hip.so` (crate `keaki-hip-sys`).
//!
//! keaki's public functions stay generic over `E: Pairing`. For `E = ark_bn254::Bn254` the arkworks calls that carry the cost --
//! `msm_unchecked` (src/kzg.rs:98), `E::pairing` + `serialize_uncompressed` + BLAKE3 (src/kem.rs:30-46, :58-69), the scalar
//! multiplications of `encapsulate` (src/kem.rs:22,30,36,37), the group FFTs of `open_fk` (src/kzg.rs:157-203) -- are replaced by one
//! call each into the C ABI. Every other curve keeps the arkworks path. Nothing here does arithmetic: field elements cross the boundary
//! as the Montgomery limbs ark-ff already holds (`Fp.0.0`), points as affine (x, y) with (0, 0) for the identity.
//!
//! NOT compiled in the image this was written in (no Rust toolchain there); `rust/README.md` says how to build and what to run.
#![cfg(feature = "hip")]

use ark_bn254::{Bn254, Fq, Fq2, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::{pairing::Pairing, CurveGroup};
use ark_ff::{BigInt, Zero};
use core::any::TypeId;
use keaki_hip_sys as sys;
use std::{
    cell::Cell,
    ffi::CStr,
    sync::{
        atomic::{AtomicU64, Ordering},
        Arc, OnceLock,
    },
};

// ------------------------------------------------------------------------------------------------ evidence and per-thread switches
/// Status codes that came back from `libkeaki_hip.so` since the process started: every call into the C ABI is followed by
/// `Device::check` / `check_group`, which count here. A test that compares "the GPU path" with arkworks asserts that this advanced while
/// its GPU leg ran (and stood still during its arkworks leg) -- otherwise a leg that silently fell back would compare arkworks with itself.
static LIBRARY_CALLS: AtomicU64 = AtomicU64::new(0);
pub fn calls() -> u64 {
    LIBRARY_CALLS.load(Ordering::SeqCst)
}

thread_local! {
    /// `with_disabled`: BN254 goes back to arkworks on THIS thread only (keaki's functions run on the caller's thread)
    static DISABLED: Cell<bool> = const { Cell::new(false) };
    /// `with_min_batch`: per-thread override of the small-call thresholds (`KEAKI_HIP_MIN_BATCH` is the process-wide one)
    static MIN_BATCH: Cell<Option<usize>> = const { Cell::new(None) };
}
struct Restore<T: Copy + 'static>(&'static std::thread::LocalKey<Cell<T>>, T);
impl<T: Copy + 'static> Drop for Restore<T> {
    fn drop(&mut self) {
        self.0.with(|c| c.set(self.1)); // also on unwinding: a failing assertion inside `f` must not leave the thread switched
    }
}
/// Runs `f` with the `hip` path switched off on this thread: what `KEAKI_HIP=off` does for the process, without touching the environment
/// (`set_var` while other threads call `var` is a data race, and it would switch THEIR "GPU legs" to arkworks as well).
pub fn with_disabled<R>(f: impl FnOnce() -> R) -> R {
    let _r = Restore(&DISABLED, DISABLED.with(|c| c.replace(true)));
    f()
}
/// Runs `f` with the small-call thresholds of `active_batch` / `active_encap` set to `n` on this thread (`0`: everything to the GPU).
pub fn with_min_batch<R>(n: usize, f: impl FnOnce() -> R) -> R {
    let _r = Restore(&MIN_BATCH, MIN_BATCH.with(|c| c.replace(Some(n))));
    f()
}

// ark-ff 0.4.2: Fp<MontBackend<_, 4>, 4> is `Fp(BigInt<4>, PhantomData)`, BigInt<4> is `BigInt([u64; 4])`: 32 bytes of Montgomery
// limbs. The slices of scalars are handed to the library in place; these asserts (and `Device::self_check`) guard the assumption.
const _: () = assert!(core::mem::size_of::<Fr>() == 32 && core::mem::align_of::<Fr>() == 8);
const _: () = assert!(core::mem::size_of::<Fq>() == 32);

/// GT as `serialize_uncompressed` writes it (12 x 32 canonical little-endian bytes).
pub const GT_BYTES: usize = 384;

// ------------------------------------------------------------------------------------------------ device
/// The GPU side of the process. One context (`KEAKI_HIP_DEVICE` picks the ordinal, default 0) -- or, with `KEAKI_HIP_DEVICES` set to a
/// count (`4` = ordinals 0..3) or a list (`0,1,2,3`; an ordinal may repeat), a *device group*: `libkeaki_hip.so` then keeps one
/// context and one host thread per entry, `commit` / `open` spread the MSM over the members by SRS range (every member holds its chunk
/// of the SRS and that chunk's window tables; the 96-byte partial sums come back through host memory, no collective, no RCCL) and
/// `encap_batch` / `decap_batch` split their items. `ctx` is member 0's context in that case (verify, open_fk, single calls).
/// The C ABI serialises calls on a context / group internally, so sharing this between threads is sound.
pub struct Device {
    ctx: *mut sys::keaki_hip_ctx,
    group: *mut sys::keaki_hip_group,
}

/// batches below this many items stay on member 0 of a group
const GROUP_MIN_ITEMS: usize = 1024;

fn parse_devices(v: &str) -> Vec<i32> {
    let v = v.trim();
    if v.contains(',') {
        v.split(',').filter_map(|t| t.trim().parse().ok()).collect()
    } else {
        (0..v.parse::<i32>().unwrap_or(1).max(1)).collect()
    }
}
unsafe impl Send for Device {}
unsafe impl Sync for Device {}

impl Device {
    pub fn global() -> &'static Device {
        static DEV: OnceLock<Device> = OnceLock::new();
        DEV.get_or_init(|| {
            if let Ok(list) = std::env::var("KEAKI_HIP_DEVICES") {
                let ordinals = parse_devices(&list);
                if ordinals.len() > 1 {
                    let mut group = core::ptr::null_mut();
                    let st = unsafe { sys::keaki_hip_group_create(ordinals.as_ptr(), ordinals.len(), &mut group) };
                    if st != sys::KEAKI_OK {
                        let msg = unsafe { CStr::from_ptr(sys::keaki_hip_group_last_error(core::ptr::null())) }.to_string_lossy().into_owned();
                        panic!("keaki `hip` feature: cannot create a device group on {ordinals:?}: {msg} (status {st})");
                    }
                    let dev = Device { ctx: unsafe { sys::keaki_hip_group_ctx(group, 0) }, group };
                    dev.self_check();
                    return dev;
                }
            }
            let ordinal = std::env::var("KEAKI_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0i32);
            let mut ctx = core::ptr::null_mut();
            let st = unsafe { sys::keaki_hip_ctx_create(ordinal, sys::KEAKI_HIP_STREAM_PRIVATE, &mut ctx) };
            if st != sys::KEAKI_OK {
                // the feature was asked for: fail loudly instead of silently computing on the CPU
                let msg = unsafe { CStr::from_ptr(sys::keaki_hip_last_error(core::ptr::null())) }.to_string_lossy().into_owned();
                panic!("keaki `hip` feature: cannot create a context on device {ordinal}: {msg} (status {st})");
            }
            let dev = Device { ctx, group: core::ptr::null_mut() };
            dev.self_check();
            dev
        })
    }

    /// `keaki_hip_ctx_set_option` on every context behind this device (a group: each member's). The names the header lists, e.g.
    /// `host_prefault` = 0 (the library never writes into or madvises caller memory itself), `pipe_chunks` = 0 (host-array calls run upload,
    /// kernels, download in that order), `msm_pipe_chunks` = k (a host-pointer MSM / `open` uploads its scalars in k chunks under the kernels).
    pub fn set_option(&self, name: &str, value: i64) {
        let cname = std::ffi::CString::new(name).expect("option name without NUL");
        for m in 0..self.members() {
            let ctx = if self.group.is_null() { self.ctx } else { unsafe { sys::keaki_hip_group_ctx(self.group, m) } };
            let st = unsafe { sys::keaki_hip_ctx_set_option(ctx, cname.as_ptr(), value) };
            self.check(st, "ctx_set_option");
        }
    }

    /// number of GPUs (contexts) behind this device
    pub fn members(&self) -> usize {
        if self.group.is_null() { 1 } else { unsafe { sys::keaki_hip_group_size(self.group) } }
    }

    fn check_group(&self, st: sys::keaki_status, what: &str) {
        LIBRARY_CALLS.fetch_add(1, Ordering::SeqCst);
        if st != sys::KEAKI_OK {
            let msg = unsafe { CStr::from_ptr(sys::keaki_hip_group_last_error(self.group)) }.to_string_lossy().into_owned();
            panic!("libkeaki_hip: {what} failed with status {st}: {msg}");
        }
    }

    fn check(&self, st: sys::keaki_status, what: &str) {
        LIBRARY_CALLS.fetch_add(1, Ordering::SeqCst);
        if st != sys::KEAKI_OK {
            let msg = unsafe { CStr::from_ptr(sys::keaki_hip_last_error(self.ctx)) }.to_string_lossy().into_owned();
            panic!("libkeaki_hip: {what} failed with status {st}: {msg}");
        }
    }

    /// 1 * G1 through the library must be arkworks' generator: catches a layout mismatch (limb order, Montgomery radix) at start-up.
    fn self_check(&self) {
        use ark_ec::AffineRepr;
        let g = G1Affine::generator();
        let one = Fr::from(1u64);
        let mut out = [0u64; 8];
        let st = unsafe { sys::keaki_hip_g1_mul_batch(self.ctx, g1_words(&g).as_ptr(), 0, fr_ptr(core::slice::from_ref(&one)), 1, out.as_mut_ptr()) };
        self.check(st, "self-check g1_mul_batch");
        assert_eq!(g1_from_words(&out), g, "libkeaki_hip and ark-ff disagree on the limb layout");
    }
}

impl Drop for Device {
    fn drop(&mut self) {
        if self.group.is_null() {
            unsafe { sys::keaki_hip_ctx_destroy(self.ctx) }
        } else {
            unsafe { sys::keaki_hip_group_destroy(self.group) } // owns the members' contexts
        }
    }
}

// ------------------------------------------------------------------------------------------------ layout (no arithmetic)
fn fr_ptr(s: &[Fr]) -> *const u64 {
    s.as_ptr() as *const u64
}
fn fq_words(x: &Fq) -> [u64; 4] {
    x.0 .0
}
fn fq_from_words(w: &[u64]) -> Fq {
    Fq::new_unchecked(BigInt::new([w[0], w[1], w[2], w[3]])) // limbs are already Montgomery residues
}
/// arkworks' `Affine { x, y, infinity }` is repr(Rust): write the fields explicitly; identity = all-zero words
pub fn g1_words(p: &G1Affine) -> [u64; 8] {
    let mut w = [0u64; 8];
    if !p.infinity {
        w[..4].copy_from_slice(&fq_words(&p.x));
        w[4..].copy_from_slice(&fq_words(&p.y));
    }
    w
}
pub fn g2_words(p: &G2Affine) -> [u64; 16] {
    let mut w = [0u64; 16];
    if !p.infinity {
        w[..4].copy_from_slice(&fq_words(&p.x.c0));
        w[4..8].copy_from_slice(&fq_words(&p.x.c1));
        w[8..12].copy_from_slice(&fq_words(&p.y.c0));
        w[12..].copy_from_slice(&fq_words(&p.y.c1));
    }
    w
}
pub fn g1_from_words(w: &[u64]) -> G1Affine {
    if w[..8].iter().all(|&x| x == 0) {
        return G1Affine::identity();
    }
    G1Affine::new_unchecked(fq_from_words(&w[..4]), fq_from_words(&w[4..8]))
}
pub fn g2_from_words(w: &[u64]) -> G2Affine {
    if w[..16].iter().all(|&x| x == 0) {
        return G2Affine::identity();
    }
    G2Affine::new_unchecked(
        Fq2::new(fq_from_words(&w[..4]), fq_from_words(&w[4..8])),
        Fq2::new(fq_from_words(&w[8..12]), fq_from_words(&w[12..16])),
    )
}
/// normalised Jacobian (x, y, 1) or (1, 1, 0) as the MSM entry points return it
fn g1_from_jac(w: &[u64; 12]) -> G1Projective {
    if w[8..].iter().all(|&x| x == 0) {
        return G1Projective::zero();
    }
    g1_from_words(&w[..8]).into()
}

// ------------------------------------------------------------------------------------------------ the SRS on the device
/// `KZGSetup::g1_aff` resident in HBM (uploaded once, with the window tables of the fixed bases). On a device group: one chunk of the
/// SRS (and its tables) per member for `commit` / `open` (`gsrs`), plus -- only once `open_fk` asks for it -- the whole SRS on member 0.
pub struct HipSrs {
    whole: OnceLock<WholeSrs>,
    gsrs: *mut sys::keaki_hip_group_srs_g1,
    gfk: std::sync::Mutex<Option<(u32, GroupFk)>>, // group devices: the FK23 handle of the last domain size asked for
    words: Vec<u64>, // kept only on a group (the lazy whole-SRS upload needs them); empty otherwise
    len: usize,
}
struct WholeSrs(*mut sys::keaki_hip_srs_g1);
unsafe impl Send for WholeSrs {}
unsafe impl Sync for WholeSrs {}
struct GroupFk(*mut sys::keaki_hip_group_fk);
unsafe impl Send for GroupFk {}
unsafe impl Send for HipSrs {}
unsafe impl Sync for HipSrs {}

/// `ShardedCommit`: what `commit` / `open` run on when the process has a device group -- the in-process multi-GPU form of
/// `msm_unchecked(&setup.g1_aff, p)` (src/kzg.rs:98). Exposed so that an application can also hold one explicitly.
pub struct ShardedCommit<'a> {
    srs: &'a HipSrs,
}
impl<'a> ShardedCommit<'a> {
    /// `None` on a single-GPU device
    pub fn new(srs: &'a HipSrs) -> Option<Self> {
        if srs.gsrs.is_null() { None } else { Some(ShardedCommit { srs }) }
    }
    pub fn members(&self) -> usize {
        Device::global().members()
    }
    /// Σ coeffs[i]·[τ^i]₁ over all members; bit-identical (affine) to the single-GPU result
    pub fn commit(&self, coeffs: &[Fr]) -> G1Projective {
        let dev = Device::global();
        let mut out = [0u64; 12];
        dev.check_group(unsafe { sys::keaki_hip_group_msm_g1(dev.group, self.srs.gsrs, fr_ptr(coeffs), coeffs.len(), out.as_mut_ptr()) }, "group_msm_g1");
        g1_from_jac(&out)
    }
    /// quotient on member 0, MSM on all members
    pub fn open(&self, coeffs: &[Fr], point: &Fr) -> G1Projective {
        let dev = Device::global();
        let mut out = [0u64; 12];
        dev.check_group(
            unsafe {
                sys::keaki_hip_group_kzg_open(dev.group, self.srs.gsrs, fr_ptr(coeffs), coeffs.len(), fr_ptr(core::slice::from_ref(point)), out.as_mut_ptr(), core::ptr::null_mut())
            },
            "group_kzg_open",
        );
        g1_from_jac(&out)
    }
}

impl HipSrs {
    pub fn upload(g1_aff: &[G1Affine]) -> Self {
        let dev = Device::global();
        let words: Vec<u64> = g1_aff.iter().flat_map(|p| g1_words(p)).collect();
        if !dev.group.is_null() {
            let mut gsrs = core::ptr::null_mut();
            dev.check_group(unsafe { sys::keaki_hip_group_srs_g1_upload(dev.group, words.as_ptr(), g1_aff.len(), 1, &mut gsrs) }, "group_srs_g1_upload");
            return HipSrs { whole: OnceLock::new(), gsrs, gfk: std::sync::Mutex::new(None), words, len: g1_aff.len() };
        }
        let whole = OnceLock::new();
        let _ = whole.set(WholeSrs(Self::upload_whole(&words, g1_aff.len(), true)));
        HipSrs { whole, gsrs: core::ptr::null_mut(), gfk: std::sync::Mutex::new(None), words: Vec::new(), len: g1_aff.len() }
    }
    fn upload_whole(words: &[u64], n: usize, tables: bool) -> *mut sys::keaki_hip_srs_g1 {
        let dev = Device::global();
        let mut srs = core::ptr::null_mut();
        dev.check(unsafe { sys::keaki_hip_srs_g1_upload(dev.ctx, words.as_ptr(), n, &mut srs) }, "srs_g1_upload");
        if tables {
            // optional memory: without room for the tables the handle keeps working through the generic MSM path
            let st = unsafe { sys::keaki_hip_srs_g1_precompute(dev.ctx, srs, core::ptr::null_mut()) };
            if st != sys::KEAKI_ERR_OOM {
                dev.check(st, "srs_g1_precompute");
            }
        }
        srs
    }
    /// the whole SRS on (member 0 of) the device: what `open_fk` reads. On a group it is uploaded on first use, without window tables.
    fn srs(&self) -> *mut sys::keaki_hip_srs_g1 {
        self.whole.get_or_init(|| WholeSrs(Self::upload_whole(&self.words, self.len, false))).0
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
}
impl Drop for HipSrs {
    fn drop(&mut self) {
        let dev = Device::global();
        if let Some(w) = self.whole.get() {
            unsafe { sys::keaki_hip_srs_g1_free(dev.ctx, w.0) }
        }
        if !self.gsrs.is_null() {
            unsafe { sys::keaki_hip_group_srs_g1_free(dev.group, self.gsrs) }
        }
        if let Some((_, fk)) = self.gfk.lock().unwrap().take() {
            unsafe { sys::keaki_hip_group_fk_free(dev.group, fk.0) }
        }
    }
}

/// Lazily uploaded device copy of a `KZGSetup`'s SRS. A field of `KZGSetup` under the feature; invisible to its derived traits.
#[derive(Default)]
pub struct SetupCache(OnceLock<Arc<HipSrs>>);
impl SetupCache {
    fn get(&self, g1_aff: &[G1Affine]) -> &Arc<HipSrs> {
        self.0.get_or_init(|| Arc::new(HipSrs::upload(g1_aff)))
    }
}
impl Clone for SetupCache {
    fn clone(&self) -> Self {
        let c = SetupCache::default();
        if let Some(a) = self.0.get() {
            let _ = c.0.set(a.clone());
        }
        c
    }
}
impl core::fmt::Debug for SetupCache {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "SetupCache(uploaded: {})", self.0.get().is_some())
    }
}
impl PartialEq for SetupCache {
    fn eq(&self, _: &Self) -> bool {
        true // a cache: equal setups are equal whether or not they have been uploaded
    }
}
impl Eq for SetupCache {}

// ------------------------------------------------------------------------------------------------ BN254 entry points
pub fn commit(srs: &HipSrs, coeffs: &[Fr]) -> G1Projective {
    if let Some(sharded) = ShardedCommit::new(srs) {
        return sharded.commit(coeffs); // KEAKI_HIP_DEVICES > 1: every GPU of the process works on its range of the SRS
    }
    let dev = Device::global();
    let mut out = [0u64; 12];
    dev.check(unsafe { sys::keaki_hip_msm_g1(dev.ctx, srs.srs(), fr_ptr(coeffs), coeffs.len(), out.as_mut_ptr()) }, "msm_g1");
    g1_from_jac(&out)
}

/// `open`: quotient and its commitment in one device call. `coeffs` = the polynomial with trailing zeros trimmed (DensePolynomial).
pub fn open(srs: &HipSrs, coeffs: &[Fr], point: &Fr) -> G1Projective {
    if let Some(sharded) = ShardedCommit::new(srs) {
        return sharded.open(coeffs, point);
    }
    // SHORT polynomials: the synthetic division q_(i-1) = p_i + point q_i is a chain -- tens of nanoseconds a step here, while the device's
    // blockwise recurrence waits for its three launches (0.7 ms at 1,000 coefficients); the MSM is the device's either way (host/keaki.cpp does the same)
    if coeffs.len() >= 2 && coeffs.len() - 1 <= OPEN_HOST_QUOTIENT_MAX {
        let n = coeffs.len() - 1;
        let mut q = vec![Fr::from(0u64); n];
        q[n - 1] = coeffs[n];
        for i in (1..n).rev() {
            q[i - 1] = coeffs[i] + *point * q[i];
        }
        return commit(srs, &q);
    }
    let dev = Device::global();
    let mut out = [0u64; 12];
    dev.check(
        unsafe { sys::keaki_hip_kzg_open(dev.ctx, srs.srs(), fr_ptr(coeffs), coeffs.len(), fr_ptr(core::slice::from_ref(point)), out.as_mut_ptr(), core::ptr::null_mut()) },
        "kzg_open",
    );
    g1_from_jac(&out)
}
const OPEN_HOST_QUOTIENT_MAX: usize = 8192;

pub fn verify(commitment: &G1Projective, tau_g2: &G2Projective, point: &Fr, value: &Fr, proof: &G1Projective) -> bool {
    let dev = Device::global();
    let (c, t, p) = (g1_words(&commitment.into_affine()), g2_words(&tau_g2.into_affine()), g1_words(&proof.into_affine()));
    let mut ok = 0i32;
    dev.check(
        unsafe {
            sys::keaki_hip_kzg_verify(dev.ctx, c.as_ptr(), t.as_ptr(), fr_ptr(core::slice::from_ref(point)), fr_ptr(core::slice::from_ref(value)), p.as_ptr(), &mut ok)
        },
        "kzg_verify",
    );
    ok != 0
}

/// FK23: all `d = coeffs.len()` openings at the d-th roots of unity (d a power of two, d <= srs.len()). `omega_2d`, `omega_2d_inv`,
/// `inv_2d` = `group_gen`, `group_gen_inv`, `size_inv` of `Radix2EvaluationDomain::new(2 d)`.
pub fn open_fk(srs: &HipSrs, coeffs: &[Fr], omega_2d: &Fr, omega_2d_inv: &Fr, inv_2d: &Fr) -> Vec<G1Projective> {
    let dev = Device::global();
    let d = coeffs.len();
    assert!(d.is_power_of_two() && d <= srs.len());
    let mut out = vec![0u64; 8 * d];
    if !dev.group.is_null() {
        // several GPUs in this process: the sharded FK23 pipeline with the exchanges inside the library (the group_fk entry points); the handle
        // (every member's copy of srs[0..d) and its part of the SRS-only transform) is kept for the next call with the same d
        let log2d = d.trailing_zeros();
        let mut slot = srs.gfk.lock().unwrap();
        if slot.as_ref().map(|(l, _)| *l) != Some(log2d) {
            if let Some((_, old)) = slot.take() {
                unsafe { sys::keaki_hip_group_fk_free(dev.group, old.0) }
            }
            let mut fk = core::ptr::null_mut();
            dev.check_group(
                unsafe {
                    sys::keaki_hip_group_fk_create(
                        dev.group, srs.words.as_ptr(), log2d, fr_ptr(core::slice::from_ref(omega_2d)), fr_ptr(core::slice::from_ref(omega_2d_inv)),
                        fr_ptr(core::slice::from_ref(inv_2d)), &mut fk,
                    )
                },
                "group_fk_create",
            );
            *slot = Some((log2d, GroupFk(fk)));
        }
        let fk = slot.as_ref().unwrap().1 .0;
        dev.check_group(unsafe { sys::keaki_hip_group_fk_open(dev.group, fk, fr_ptr(coeffs), out.as_mut_ptr()) }, "group_fk_open");
        return out.chunks_exact(8).map(|w| g1_from_words(w).into()).collect();
    }
    dev.check(
        unsafe {
            sys::keaki_hip_open_fk_poly(
                dev.ctx, srs.srs(), d.trailing_zeros(), fr_ptr(coeffs), fr_ptr(core::slice::from_ref(omega_2d)),
                fr_ptr(core::slice::from_ref(omega_2d_inv)), fr_ptr(core::slice::from_ref(inv_2d)), out.as_mut_ptr(),
            )
        },
        "open_fk_poly",
    );
    out.chunks_exact(8).map(|w| g1_from_words(w).into()).collect()
}

/// Whether `vec_commit` takes the one-call device path: a domain of at least 2^12 evaluations that the SRS covers, on a single-GPU device
/// (a device group runs the steps one by one: its FK23 openings and its commit are sharded over the members).
pub fn fused_vec_commit(domain_size: usize, srs_len: usize) -> bool {
    Device::global().group.is_null() && domain_size >= (1 << 12) && domain_size <= srs_len
}

/// The body of `vec_commit` (src/vec.rs:36-46) behind its padding draw, in ONE device call: `padded` = the vector and the padding scalar,
/// `size` = the domain size (`Radix2EvaluationDomain::new(padded.len())`'s). Returns the commitment and the `size` proofs.
pub fn vec_commit(srs: &HipSrs, padded: &[Fr], size: usize) -> (G1Projective, Vec<G1Projective>) {
    use ark_poly::{EvaluationDomain, Radix2EvaluationDomain};
    let dev = Device::global();
    assert!(size.is_power_of_two() && padded.len() <= size && size <= srs.len());
    let dom = Radix2EvaluationDomain::<Fr>::new(size).unwrap();
    let dom2 = Radix2EvaluationDomain::<Fr>::new(2 * size).unwrap();
    let one = |x: &Fr| fr_ptr(core::slice::from_ref(x));
    let mut com = [0u64; 12];
    let mut out = vec![0u64; 8 * size];
    dev.check(
        unsafe {
            sys::keaki_hip_vec_commit(
                dev.ctx, srs.srs(), fr_ptr(padded), padded.len(), core::ptr::null(), size.trailing_zeros(), one(&dom.group_gen_inv), one(&dom.size_inv),
                one(&dom2.group_gen), one(&dom2.group_gen_inv), one(&dom2.size_inv), com.as_mut_ptr(), out.as_mut_ptr(),
            )
        },
        "vec_commit",
    );
    (g1_from_jac(&com), out.chunks_exact(8).map(|w| g1_from_words(w).into()).collect())
}

/// The collectives a sharded `open_fk` owes between its steps: implemented by the application over its communication library (RCCL:
/// `ncclAllToAll`-style exchange with equal splits, `ncclAllGather`) on DEVICE buffers; both must have completed when they return.
pub trait FkExchange {
    /// `send`: `world` chunks of `bytes_per_peer`, chunk q for rank q; `recv`: the chunks received, in rank order
    fn all_to_all(&mut self, send: *mut core::ffi::c_void, recv: *mut core::ffi::c_void, bytes_per_peer: usize);
    /// `send`: `bytes_per_rank` of this rank; `recv`: every rank's, in rank order
    fn all_gather(&mut self, send: *mut core::ffi::c_void, recv: *mut core::ffi::c_void, bytes_per_rank: usize);
}

/// `open_fk` with its group FFTs sharded over the `world` ranks (= GPUs) of a job: every rank calls this with the same coefficients and
/// gets all `d` proofs; it runs 1/world of the butterflies and scalar-mults (the `fk_shard` entry points of include/keaki_hip.h). `send` / `recv`:
/// device buffers of `buffer_bytes()` each, owned by the caller (they are what the collectives move).
pub struct ShardedOpenFk {
    fk: *mut sys::keaki_hip_fk_shard,
    _srs: Arc<HipSrs>, // the C handle keeps a raw pointer to the SRS and reads it lazily (setup step 0 runs inside the first open)
    sizes: [usize; 4],
    d: usize,
    prepared: bool,
}
unsafe impl Send for ShardedOpenFk {}

impl ShardedOpenFk {
    /// `d`: a power of two with `world^2 <= d <= srs.len()`; `world` a power of two >= 2; roots as for `open_fk`
    pub fn new(srs: &Arc<HipSrs>, d: usize, rank: u32, world: u32, omega_2d: &Fr, omega_2d_inv: &Fr, inv_2d: &Fr) -> Self {
        let dev = Device::global();
        assert!(d.is_power_of_two() && d <= srs.len());
        let mut fk = core::ptr::null_mut();
        dev.check(
            unsafe {
                sys::keaki_hip_fk_shard_create(
                    dev.ctx, srs.srs(), d.trailing_zeros(), rank, world, fr_ptr(core::slice::from_ref(omega_2d)),
                    fr_ptr(core::slice::from_ref(omega_2d_inv)), fr_ptr(core::slice::from_ref(inv_2d)), &mut fk,
                )
            },
            "fk_shard_create",
        );
        let mut sizes = [0usize; 4];
        dev.check(unsafe { sys::keaki_hip_fk_shard_sizes(fk, sizes.as_mut_ptr()) }, "fk_shard_sizes");
        ShardedOpenFk { fk, _srs: srs.clone(), sizes, d, prepared: false }
    }
    pub fn buffer_bytes(&self) -> usize {
        self.sizes[0]
    }
    /// setup time: this rank's part of the SRS-only transform (one all-to-all)
    pub fn prepare(&mut self, send: *mut core::ffi::c_void, recv: *mut core::ffi::c_void, ex: &mut dyn FkExchange) {
        if self.prepared {
            return;
        }
        let dev = Device::global();
        dev.check(unsafe { sys::keaki_hip_fk_shard_setup(dev.ctx, self.fk, 0, send, core::ptr::null_mut()) }, "fk_shard_setup 0");
        dev.check(unsafe { sys::keaki_hip_synchronize(dev.ctx) }, "synchronize");
        ex.all_to_all(send, recv, self.sizes[1]);
        dev.check(unsafe { sys::keaki_hip_fk_shard_setup(dev.ctx, self.fk, 1, core::ptr::null_mut(), recv) }, "fk_shard_setup 1");
        dev.check(unsafe { sys::keaki_hip_synchronize(dev.ctx) }, "synchronize");
        self.prepared = true;
    }
    pub fn open(&mut self, coeffs: &[Fr], send: *mut core::ffi::c_void, recv: *mut core::ffi::c_void, ex: &mut dyn FkExchange) -> Vec<G1Projective> {
        assert!(coeffs.len() == self.d);
        self.prepare(send, recv, ex);
        let dev = Device::global();
        let null = core::ptr::null_mut::<core::ffi::c_void>();
        let step = |i: i32, c: *const u64, s, r, out: *mut u64| {
            dev.check(unsafe { sys::keaki_hip_fk_shard_open(dev.ctx, self.fk, i, c, s, r, out) }, "fk_shard_open");
            dev.check(unsafe { sys::keaki_hip_synchronize(dev.ctx) }, "synchronize");
        };
        step(0, fr_ptr(coeffs), send, null, core::ptr::null_mut());
        ex.all_to_all(send, recv, self.sizes[2]);
        step(1, core::ptr::null(), send, recv, core::ptr::null_mut());
        ex.all_to_all(send, recv, self.sizes[2]);
        step(2, core::ptr::null(), send, recv, core::ptr::null_mut());
        ex.all_gather(send, recv, self.sizes[3]);
        let mut out = vec![0u64; 8 * self.d];
        step(3, core::ptr::null(), null, recv, out.as_mut_ptr());
        out.chunks_exact(8).map(|w| g1_from_words(w).into()).collect()
    }
}
impl Drop for ShardedOpenFk {
    fn drop(&mut self) {
        unsafe { sys::keaki_hip_fk_shard_free(Device::global().ctx, self.fk) }
    }
}

/// The loop body of `vec_encrypt` (src/vec.rs:63-66 -> src/enc.rs:19-40 -> src/kem.rs:13-50) for all items at once: the SAME commitment
/// and setup for every item, `rs[i]` drawn by the caller in index order. Returns the ciphertext points and `n * msg_len` key bytes.
pub fn encap_batch(commitment: &G1Projective, tau_g2: &G2Projective, points: &[Fr], values: &[Fr], rs: &[Fr], msg_len: usize) -> (Vec<G2Projective>, Vec<u8>) {
    let dev = Device::global();
    let n = points.len();
    assert!(values.len() >= n && rs.len() == n);
    let (c, t) = (g1_words(&commitment.into_affine()), g2_words(&tau_g2.into_affine()));
    let mut ct = vec![0u64; 16 * n];
    let mut gt = vec![0u8; if msg_len == 0 { GT_BYTES * n } else { 0 }];
    let mut key = vec![0u8; n * msg_len];
    let gt_ptr = if msg_len == 0 { gt.as_mut_ptr() } else { core::ptr::null_mut() };
    let key_ptr = if msg_len == 0 { core::ptr::null_mut() } else { key.as_mut_ptr() };
    if !dev.group.is_null() && n >= GROUP_MIN_ITEMS {
        // the items of src/vec.rs:63-66 split by range over the GPUs of the process; every member writes its slice of ct / key in place
        dev.check_group(
            unsafe { sys::keaki_hip_group_encap_batch(dev.group, c.as_ptr(), t.as_ptr(), fr_ptr(points), fr_ptr(values), fr_ptr(rs), n, ct.as_mut_ptr(), gt_ptr, key_ptr, msg_len) },
            "group_encap_batch",
        );
    } else {
        dev.check(
            unsafe { sys::keaki_hip_encap_batch(dev.ctx, c.as_ptr(), t.as_ptr(), fr_ptr(points), fr_ptr(values), fr_ptr(rs), n, ct.as_mut_ptr(), gt_ptr, key_ptr, msg_len) },
            "encap_batch",
        );
    }
    (ct.chunks_exact(16).map(|w| g2_from_words(w).into()).collect(), key)
}

/// The loop body of `vec_decrypt` (src/vec.rs:75-78 -> src/kem.rs:55-72): `n * msg_len` key bytes.
pub fn decap_batch(proofs: &[G1Projective], cts: &[G2Projective], msg_len: usize) -> Vec<u8> {
    let dev = Device::global();
    let n = cts.len();
    if n == 0 || msg_len == 0 {
        return Vec::new();
    }
    let p: Vec<u64> = G1Projective::normalize_batch(&proofs[..n]).iter().flat_map(|a| g1_words(a)).collect();
    let q: Vec<u64> = G2Projective::normalize_batch(cts).iter().flat_map(|a| g2_words(a)).collect();
    let mut key = vec![0u8; n * msg_len];
    if !dev.group.is_null() && n >= GROUP_MIN_ITEMS {
        dev.check_group(
            unsafe { sys::keaki_hip_group_decap_batch(dev.group, p.as_ptr(), q.as_ptr(), n, core::ptr::null_mut(), key.as_mut_ptr(), msg_len) },
            "group_decap_batch",
        );
    } else {
        dev.check(
            unsafe { sys::keaki_hip_decap_batch(dev.ctx, p.as_ptr(), q.as_ptr(), n, core::ptr::null_mut(), key.as_mut_ptr(), msg_len) },
            "decap_batch",
        );
    }
    key
}

/// `enc::encrypt` for all items of `vec_encrypt` at once (src/vec.rs:63-66 -> src/enc.rs:19-40): equal-length messages, contiguous. The XOR of
/// src/enc.rs:32-36 runs on the device behind the KDF: neither GT bytes nor keys come back. Returns the ciphertext points and the bodies.
pub fn encrypt_batch(commitment: &G1Projective, tau_g2: &G2Projective, points: &[Fr], values: &[Fr], rs: &[Fr], msgs: &[u8], msg_len: usize) -> (Vec<G2Projective>, Vec<u8>) {
    let dev = Device::global();
    let n = points.len();
    assert!(values.len() >= n && rs.len() == n && msg_len > 0 && msgs.len() == n * msg_len);
    let (c, t) = (g1_words(&commitment.into_affine()), g2_words(&tau_g2.into_affine()));
    let mut ct = vec![0u64; 16 * n];
    let mut body = vec![0u8; n * msg_len];
    if !dev.group.is_null() && n >= GROUP_MIN_ITEMS {
        dev.check_group(
            unsafe {
                sys::keaki_hip_group_encrypt_batch(dev.group, c.as_ptr(), t.as_ptr(), fr_ptr(points), fr_ptr(values), fr_ptr(rs), msgs.as_ptr(), n, ct.as_mut_ptr(), body.as_mut_ptr(), msg_len)
            },
            "group_encrypt_batch",
        );
    } else {
        dev.check(
            unsafe { sys::keaki_hip_encrypt_batch(dev.ctx, c.as_ptr(), t.as_ptr(), fr_ptr(points), fr_ptr(values), fr_ptr(rs), msgs.as_ptr(), n, ct.as_mut_ptr(), body.as_mut_ptr(), msg_len) },
            "encrypt_batch",
        );
    }
    (ct.chunks_exact(16).map(|w| g2_from_words(w).into()).collect(), body)
}

/// `enc::decrypt` for all items of `vec_decrypt` at once (src/vec.rs:75-78 -> src/enc.rs:44-55): the messages.
pub fn decrypt_batch(proofs: &[G1Projective], cts: &[G2Projective], bodies: &[u8], msg_len: usize) -> Vec<u8> {
    let dev = Device::global();
    let n = cts.len();
    if n == 0 || msg_len == 0 {
        return Vec::new();
    }
    assert!(bodies.len() == n * msg_len);
    let p: Vec<u64> = G1Projective::normalize_batch(&proofs[..n]).iter().flat_map(|a| g1_words(a)).collect();
    let q: Vec<u64> = G2Projective::normalize_batch(cts).iter().flat_map(|a| g2_words(a)).collect();
    let mut out = vec![0u8; n * msg_len];
    if !dev.group.is_null() && n >= GROUP_MIN_ITEMS {
        dev.check_group(
            unsafe { sys::keaki_hip_group_decrypt_batch(dev.group, p.as_ptr(), q.as_ptr(), bodies.as_ptr(), n, out.as_mut_ptr(), msg_len) },
            "group_decrypt_batch",
        );
    } else {
        dev.check(unsafe { sys::keaki_hip_decrypt_batch(dev.ctx, p.as_ptr(), q.as_ptr(), bodies.as_ptr(), n, out.as_mut_ptr(), msg_len) }, "decrypt_batch");
    }
    out
}

/// `serialize_uncompressed(E::pairing(p, q))` for a batch (tests and callers that want GT itself).
pub fn pairing_bytes(ps: &[G1Affine], qs: &[G2Affine]) -> Vec<u8> {
    let dev = Device::global();
    let n = ps.len();
    assert_eq!(qs.len(), n);
    let p: Vec<u64> = ps.iter().flat_map(|a| g1_words(a)).collect();
    let q: Vec<u64> = qs.iter().flat_map(|a| g2_words(a)).collect();
    let mut gt = vec![0u8; GT_BYTES * n];
    dev.check(unsafe { sys::keaki_hip_pairing_batch(dev.ctx, p.as_ptr(), q.as_ptr(), 1, n, gt.as_mut_ptr()) }, "pairing_batch");
    gt
}

/// `domain.ifft` of src/vec.rs:37 on the device: `data.len()` = the domain size (power of two), padded by the caller.
pub fn fr_ifft_in_place(data: &mut [Fr], group_gen_inv: &Fr, size_inv: &Fr) {
    let dev = Device::global();
    assert!(data.len().is_power_of_two());
    dev.check(
        unsafe {
            sys::keaki_hip_fr_fft(dev.ctx, data.as_mut_ptr() as *mut u64, data.len().trailing_zeros(), fr_ptr(core::slice::from_ref(group_gen_inv)), fr_ptr(core::slice::from_ref(size_inv)))
        },
        "fr_fft",
    );
}

// ------------------------------------------------------------------------------------------------ dispatch from the generic functions
/// `true` when the generic parameter is BN254 and the feature is not switched off at run time (`KEAKI_HIP=off` for the process,
/// `with_disabled` for the calling thread).
pub fn active<E: Pairing>() -> bool {
    TypeId::of::<E>() == TypeId::of::<Bn254>() && !DISABLED.with(|c| c.get()) && std::env::var("KEAKI_HIP").map(|v| v != "off").unwrap_or(true)
}

/// Small calls. The thresholds follow the single-call figures of `bench.py`'s `single_calls` block (BENCH_r05: wall clock of one call through
/// the library against the same call on ONE host core, degree-128 setup, 32-byte key):
///
/// | call | GPU | one CPU core | taken from |
/// |---|---|---|---|
/// | `encapsulate` (commitment seen before / new) | 0.48 ms / ~2.0 ms | 1.7 ms | 1 item |
/// | `verify` (two pairings) | 1.68 ms | 2.2 ms | 2 pairings |
/// | `decapsulate` (one pairing with a per-item Q) | 1.54 ms | 0.9 ms | 2 items |
///
/// A call with few pairings runs each of them on twelve lanes and two waves (`pairing_wide.hip.h`, 1.4 ms per pairing whatever their
/// number up to ~1,000), so ONE decapsulation is the only call a CPU core still wins. `KEAKI_HIP_MIN_BATCH=n` overrides both thresholds
/// (`0` sends everything to the GPU -- the parity test does that; a large value keeps all pairing-shaped small calls on arkworks).
/// MSM-shaped calls (`commit`, `open`, `open_fk`) always go to the GPU (a 129-coefficient commit: 0.58 ms against 2.4 ms).
/// (Rounds 1-4 kept fewer than 8 items on arkworks: one pairing then cost 5 ms on a lane pair.)
fn min_batch_override() -> Option<usize> {
    MIN_BATCH.with(|c| c.get()).or_else(|| std::env::var("KEAKI_HIP_MIN_BATCH").ok().and_then(|v| v.parse::<usize>().ok()))
}
/// pairing-shaped work of `items` pairings with per-item second arguments: `decapsulate` (1), `verify` (2), `vec_decrypt` (its length)
pub fn active_batch<E: Pairing>(items: usize) -> bool {
    active::<E>() && items >= min_batch_override().unwrap_or(2)
}
/// encapsulation work: `encapsulate` (1), `vec_encrypt` (its length) -- fixed-base sums and GT exponentiations, no pairing per item
pub fn active_encap<E: Pairing>(items: usize) -> bool {
    active::<E>() && items >= min_batch_override().unwrap_or(1)
}

/// Reinterpret `&A` as `&B` when they are the same type. The callers have established `E == Bn254`, which makes `E::G1` and
/// `G1Projective` (etc.) one type; TypeId re-checks it, so a wrong call panics instead of transmuting.
pub fn same<A: 'static, B: 'static>(a: &A) -> &B {
    assert_eq!(TypeId::of::<A>(), TypeId::of::<B>());
    unsafe { &*(a as *const A as *const B) }
}
pub fn same_slice<A: 'static, B: 'static>(a: &[A]) -> &[B] {
    assert_eq!(TypeId::of::<A>(), TypeId::of::<B>());
    unsafe { core::slice::from_raw_parts(a.as_ptr() as *const B, a.len()) }
}
pub fn same_owned<A: 'static, B: 'static>(a: A) -> B {
    assert_eq!(TypeId::of::<A>(), TypeId::of::<B>());
    let a = core::mem::ManuallyDrop::new(a);
    unsafe { core::ptr::read(&*a as *const A as *const B) }
}
pub fn same_vec<A: 'static, B: 'static>(v: Vec<A>) -> Vec<B> {
    assert_eq!(TypeId::of::<A>(), TypeId::of::<B>());
    let mut v = core::mem::ManuallyDrop::new(v);
    unsafe { Vec::from_raw_parts(v.as_mut_ptr() as *mut B, v.len(), v.capacity()) }
}

/// The device copy of a setup's SRS (uploaded on first use). An `Arc`: `ShardedOpenFk::new` keeps a clone, because the C handle behind it
/// reads the SRS lazily.
pub fn srs_of<'a, E: Pairing>(cache: &'a SetupCache, g1_aff: &[E::G1Affine]) -> &'a Arc<HipSrs> {
    cache.get(same_slice::<E::G1Affine, G1Affine>(g1_aff))
}
