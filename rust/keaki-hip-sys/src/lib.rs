//! Raw bindings to `libkeaki_hip.so`, one declaration per export of `include/keaki_hip.h`, in the header's order.
//! `tests/test_rust_shim.py` (CPU suite of the MI355X tree) parses the `extern "C"` block below and checks every name, parameter
//! count and parameter width against the header, so the two cannot drift apart unnoticed.
//!
//! Layouts (see the header): Fr / Fq = `[u64; 4]` Montgomery limbs exactly as `ark_ff::Fp.0.0`; G1 affine `[u64; 8]`, G1 Jacobian
//! `[u64; 12]` (normalised), G2 affine `[u64; 16]`, G2 Jacobian `[u64; 24]`, GT = 384 bytes of `serialize_uncompressed`.
#![no_std]
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_void};

#[cfg(feature = "rccl")]
pub mod rccl;

pub type keaki_status = i32;
pub const KEAKI_OK: keaki_status = 0;
pub const KEAKI_ERR_BAD_ARG: keaki_status = -1;
pub const KEAKI_ERR_HIP: keaki_status = -2;
pub const KEAKI_ERR_OOM: keaki_status = -3;
pub const KEAKI_ERR_NO_DEVICE: keaki_status = -4;
pub const KEAKI_ERR_TOO_LARGE: keaki_status = -5;

/// `KEAKI_HIP_STREAM_PRIVATE`: the context creates its own non-blocking stream.
pub const KEAKI_HIP_STREAM_PRIVATE: *mut c_void = core::ptr::null_mut();
/// `KEAKI_HIP_STREAM_LEGACY` (= hipStreamLegacy): the device's legacy default stream.
pub const KEAKI_HIP_STREAM_LEGACY: *mut c_void = 1 as *mut c_void;

#[repr(C)]
pub struct keaki_hip_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct keaki_hip_srs_g1 {
    _private: [u8; 0],
}
#[repr(C)]
pub struct keaki_hip_srs_g2 {
    _private: [u8; 0],
}
#[repr(C)]
pub struct keaki_hip_fk_shard {
    _private: [u8; 0],
}
/// in-process multi-GPU: one context and one host thread per member inside the library
#[repr(C)]
pub struct keaki_hip_group {
    _private: [u8; 0],
}
#[repr(C)]
pub struct keaki_hip_group_srs_g1 {
    _private: [u8; 0],
}
#[repr(C)]
pub struct keaki_hip_group_fk {
    _private: [u8; 0],
}

extern "C" {
    // ---- context
    pub fn keaki_hip_ctx_create(device: i32, stream: *mut c_void, out: *mut *mut keaki_hip_ctx) -> keaki_status;
    pub fn keaki_hip_ctx_destroy(ctx: *mut keaki_hip_ctx);
    pub fn keaki_hip_last_error(ctx: *const keaki_hip_ctx) -> *const c_char;
    pub fn keaki_hip_synchronize(ctx: *mut keaki_hip_ctx) -> keaki_status;
    pub fn keaki_hip_ctx_stream(ctx: *const keaki_hip_ctx) -> *mut c_void;
    pub fn keaki_hip_ctx_device(ctx: *const keaki_hip_ctx) -> i32;
    pub fn keaki_hip_version() -> *const c_char;
    pub fn keaki_hip_ctx_set_option(ctx: *mut keaki_hip_ctx, name: *const c_char, value: i64) -> keaki_status;
    pub fn keaki_hip_debug_set_alloc_limit(ctx: *mut keaki_hip_ctx, bytes: usize) -> keaki_status;
    pub fn keaki_hip_ctx_memory(ctx: *mut keaki_hip_ctx, out4: *mut usize) -> keaki_status;
    pub fn keaki_hip_ctx_trim(ctx: *mut keaki_hip_ctx) -> keaki_status;

    // ---- SRS (replaces KZGSetup::g1_aff, src/kzg.rs:22-29)
    pub fn keaki_hip_srs_g1_upload(ctx: *mut keaki_hip_ctx, points_aff: *const u64, n: usize, out: *mut *mut keaki_hip_srs_g1) -> keaki_status;
    pub fn keaki_hip_srs_g1_wrap_dev(ctx: *mut keaki_hip_ctx, d_points_aff: *const c_void, n: usize, out: *mut *mut keaki_hip_srs_g1) -> keaki_status;
    pub fn keaki_hip_srs_g1_slice(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g1, offset: usize, n: usize, out: *mut *mut keaki_hip_srs_g1) -> keaki_status;
    pub fn keaki_hip_srs_g1_len(srs: *const keaki_hip_srs_g1) -> usize;
    pub fn keaki_hip_srs_g1_precompute(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g1, table_bytes_out: *mut usize) -> keaki_status;
    pub fn keaki_hip_srs_g1_free(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g1);
    pub fn keaki_hip_srs_g2_upload(ctx: *mut keaki_hip_ctx, points_aff: *const u64, n: usize, out: *mut *mut keaki_hip_srs_g2) -> keaki_status;
    pub fn keaki_hip_srs_g2_wrap_dev(ctx: *mut keaki_hip_ctx, d_points_aff: *const c_void, n: usize, out: *mut *mut keaki_hip_srs_g2) -> keaki_status;
    pub fn keaki_hip_srs_g2_precompute(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g2, table_bytes_out: *mut usize) -> keaki_status;
    pub fn keaki_hip_srs_g2_free(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g2);

    // ---- MSM (replaces msm_unchecked, src/kzg.rs:98)
    pub fn keaki_hip_msm_g1(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g1, scalars: *const u64, n: usize, out_jac: *mut u64) -> keaki_status;
    pub fn keaki_hip_msm_g1_dev(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g1, d_scalars: *const c_void, n: usize, d_out_jac: *mut c_void) -> keaki_status;
    pub fn keaki_hip_msm_g2(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g2, scalars: *const u64, n: usize, out_jac: *mut u64) -> keaki_status;
    pub fn keaki_hip_msm_g2_dev(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g2, d_scalars: *const c_void, n: usize, d_out_jac: *mut c_void) -> keaki_status;
    pub fn keaki_hip_g1_sum_dev(ctx: *mut keaki_hip_ctx, d_points_jac: *const c_void, k: usize, d_out_jac: *mut c_void) -> keaki_status;
    pub fn keaki_hip_g1_sum(ctx: *mut keaki_hip_ctx, points_jac: *const u64, k: usize, out_jac: *mut u64) -> keaki_status;

    // ---- FK23 batch openings (replaces kzg::open_fk, src/kzg.rs:157-203) and the scalar-field FFT (src/vec.rs:36-37)
    pub fn keaki_hip_open_fk(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g1, log2d: u32, hat_a: *const u64, tw_2d: *const u64,
                             tw_2d_inv: *const u64, tw_d: *const u64, proofs_out_aff: *mut u64) -> keaki_status;
    pub fn keaki_hip_open_fk_poly(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g1, log2d: u32, coeffs: *const u64, omega_2d: *const u64,
                                  omega_2d_inv: *const u64, inv_2d: *const u64, proofs_out_aff: *mut u64) -> keaki_status;
    pub fn keaki_hip_srs_g1_precompute_fk(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g1, log2d: u32, omega_2d: *const u64) -> keaki_status;
    pub fn keaki_hip_fr_fft(ctx: *mut keaki_hip_ctx, data: *mut u64, log2n: u32, omega: *const u64, scale_or_null: *const u64) -> keaki_status;
    // FK23 sharded over the ranks of a multi-GPU job: the steps between the caller's RCCL exchanges (include/keaki_hip.h)
    pub fn keaki_hip_fk_shard_create(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g1, log2d: u32, rank: u32, world: u32, omega_2d: *const u64,
                                     omega_2d_inv: *const u64, inv_2d: *const u64, out: *mut *mut keaki_hip_fk_shard) -> keaki_status;
    pub fn keaki_hip_fk_shard_free(ctx: *mut keaki_hip_ctx, fk: *mut keaki_hip_fk_shard);
    pub fn keaki_hip_fk_shard_sizes(fk: *const keaki_hip_fk_shard, sizes4: *mut usize) -> keaki_status;
    pub fn keaki_hip_fk_shard_setup(ctx: *mut keaki_hip_ctx, fk: *mut keaki_hip_fk_shard, step: i32, d_send: *mut c_void, d_recv: *mut c_void) -> keaki_status;
    pub fn keaki_hip_fk_shard_open(ctx: *mut keaki_hip_ctx, fk: *mut keaki_hip_fk_shard, step: i32, coeffs: *const u64, d_send: *mut c_void,
                                   d_recv: *mut c_void, proofs_out_aff: *mut u64) -> keaki_status;

    // vec_commit's body in one call (src/vec.rs:36-46): iFFT, FK23 openings, commit, coefficients stay on the device
    pub fn keaki_hip_vec_commit(ctx: *mut keaki_hip_ctx, srs: *mut keaki_hip_srs_g1, values: *const u64, n: usize, pad: *const u64, log2d: u32,
                                omega_d_inv: *const u64, inv_d: *const u64, omega_2d: *const u64, omega_2d_inv: *const u64, inv_2d: *const u64,
                                com_out_jac: *mut u64, proofs_out_aff: *mut u64) -> keaki_status;

    // ---- KZG open / verify in one call (src/kzg.rs:104-124, :127-151)
    pub fn keaki_hip_kzg_open(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g1, coeffs: *const u64, n: usize, point: *const u64,
                              proof_out_jac: *mut u64, value_out: *mut u64) -> keaki_status;
    pub fn keaki_hip_kzg_quotient(ctx: *mut keaki_hip_ctx, coeffs: *const u64, n: usize, point: *const u64, quotient_out: *mut u64,
                                  value_out: *mut u64) -> keaki_status;
    pub fn keaki_hip_kzg_verify(ctx: *mut keaki_hip_ctx, com_aff: *const u64, tau_g2_aff: *const u64, point: *const u64, value: *const u64,
                                proof_aff: *const u64, ok_out: *mut i32) -> keaki_status;

    // ---- SRS ingest checks (src/kzg/ptau.rs:266,314 read unchecked)
    pub fn keaki_hip_srs_g1_check(ctx: *mut keaki_hip_ctx, srs: *const keaki_hip_srs_g1, n_off_curve: *mut u64, first_off_curve: *mut u64) -> keaki_status;
    pub fn keaki_hip_g2_check(ctx: *mut keaki_hip_ctx, points_aff: *const u64, n: usize, n_off_curve: *mut u64, first_off_curve: *mut u64) -> keaki_status;

    // ---- batched scalar multiplication (replaces `.mul(scalar)`, src/kem.rs:22,30,36,37)
    pub fn keaki_hip_g1_mul_batch(ctx: *mut keaki_hip_ctx, points_aff: *const u64, point_stride: i32, scalars: *const u64, n: usize, out_aff: *mut u64) -> keaki_status;
    pub fn keaki_hip_g2_mul_batch(ctx: *mut keaki_hip_ctx, points_aff: *const u64, point_stride: i32, scalars: *const u64, n: usize, out_aff: *mut u64) -> keaki_status;
    pub fn keaki_hip_g1_mul_batch_dev(ctx: *mut keaki_hip_ctx, d_points_aff: *const c_void, point_stride: i32, d_scalars: *const c_void, n: usize, d_out_aff: *mut c_void) -> keaki_status;
    pub fn keaki_hip_g2_mul_batch_dev(ctx: *mut keaki_hip_ctx, d_points_aff: *const c_void, point_stride: i32, d_scalars: *const c_void, n: usize, d_out_aff: *mut c_void) -> keaki_status;

    // ---- batched pairing (replaces E::pairing + serialize_uncompressed, src/kem.rs:30-32,58-61)
    pub fn keaki_hip_pairing_batch(ctx: *mut keaki_hip_ctx, g1_aff: *const u64, g2_aff: *const u64, g2_stride: i32, n: usize, gt_out: *mut u8) -> keaki_status;
    pub fn keaki_hip_pairing_batch_dev(ctx: *mut keaki_hip_ctx, d_g1_aff: *const c_void, d_g2_aff: *const c_void, g2_stride: i32, n: usize, d_gt_out: *mut c_void) -> keaki_status;

    // ---- KEM composites: the loop bodies of src/vec.rs:63-66 and :75-78
    pub fn keaki_hip_encap_batch(ctx: *mut keaki_hip_ctx, com_aff: *const u64, tau_g2_aff: *const u64, points: *const u64, values: *const u64,
                                 r: *const u64, n: usize, ct_out_aff: *mut u64, gt_out: *mut u8, key_out: *mut u8, msg_len: usize) -> keaki_status;
    pub fn keaki_hip_encap_batch_dev(ctx: *mut keaki_hip_ctx, d_com_aff: *const c_void, d_tau_g2_aff: *const c_void, d_points: *const c_void,
                                     d_values: *const c_void, d_r: *const c_void, n: usize, d_ct_out_aff: *mut c_void, d_gt_out: *mut c_void,
                                     d_key_out: *mut c_void, msg_len: usize) -> keaki_status;
    pub fn keaki_hip_encap_prepare(ctx: *mut keaki_hip_ctx, tau_g2_aff: *const u64, batch_hint: usize) -> keaki_status;
    pub fn keaki_hip_decap_batch(ctx: *mut keaki_hip_ctx, proofs_aff: *const u64, cts_aff: *const u64, n: usize, gt_out: *mut u8, key_out: *mut u8,
                                 msg_len: usize) -> keaki_status;
    pub fn keaki_hip_decap_batch_dev(ctx: *mut keaki_hip_ctx, d_proofs_aff: *const c_void, d_cts_aff: *const c_void, n: usize, d_gt_out: *mut c_void,
                                     d_key_out: *mut c_void, msg_len: usize) -> keaki_status;
    // enc::encrypt / enc::decrypt over a batch: KEM + the XOR DEM on the device (src/enc.rs:19-55 inside src/vec.rs:63-66, :75-78)
    pub fn keaki_hip_encrypt_batch(ctx: *mut keaki_hip_ctx, com_aff: *const u64, tau_g2_aff: *const u64, points: *const u64, values: *const u64,
                                   r: *const u64, msgs: *const u8, n: usize, ct_out_aff: *mut u64, body_out: *mut u8, msg_len: usize) -> keaki_status;
    pub fn keaki_hip_encrypt_batch_dev(ctx: *mut keaki_hip_ctx, d_com_aff: *const c_void, d_tau_g2_aff: *const c_void, d_points: *const c_void,
                                       d_values: *const c_void, d_r: *const c_void, n: usize, d_ct_out_aff: *mut c_void, d_body_inout: *mut c_void,
                                       msg_len: usize) -> keaki_status;
    pub fn keaki_hip_decrypt_batch(ctx: *mut keaki_hip_ctx, proofs_aff: *const u64, cts_aff: *const u64, bodies: *const u8, n: usize, msgs_out: *mut u8,
                                   msg_len: usize) -> keaki_status;
    pub fn keaki_hip_decrypt_batch_dev(ctx: *mut keaki_hip_ctx, d_proofs_aff: *const c_void, d_cts_aff: *const c_void, n: usize,
                                       d_body_inout: *mut c_void, msg_len: usize) -> keaki_status;

    // ---- device group: several GPUs of ONE process (one context + one host thread per member inside the library)
    pub fn keaki_hip_group_create(devices: *const i32, n_devices: usize, out: *mut *mut keaki_hip_group) -> keaki_status;
    pub fn keaki_hip_group_destroy(g: *mut keaki_hip_group);
    pub fn keaki_hip_group_size(g: *const keaki_hip_group) -> usize;
    pub fn keaki_hip_group_ctx(g: *const keaki_hip_group, member: usize) -> *mut keaki_hip_ctx;
    pub fn keaki_hip_group_last_error(g: *const keaki_hip_group) -> *const c_char;
    pub fn keaki_hip_group_srs_g1_upload(g: *mut keaki_hip_group, points_aff: *const u64, n: usize, precompute: i32,
                                         out: *mut *mut keaki_hip_group_srs_g1) -> keaki_status;
    pub fn keaki_hip_group_peer_note(g: *const keaki_hip_group) -> *const c_char;
    pub fn keaki_hip_group_srs_g1_len(srs: *const keaki_hip_group_srs_g1) -> usize;
    pub fn keaki_hip_group_srs_g1_has_tables(srs: *const keaki_hip_group_srs_g1) -> i32;
    pub fn keaki_hip_group_srs_g1_free(g: *mut keaki_hip_group, srs: *mut keaki_hip_group_srs_g1);
    pub fn keaki_hip_group_msm_g1(g: *mut keaki_hip_group, srs: *const keaki_hip_group_srs_g1, scalars: *const u64, n: usize, out_jac: *mut u64) -> keaki_status;
    pub fn keaki_hip_group_kzg_open(g: *mut keaki_hip_group, srs: *const keaki_hip_group_srs_g1, coeffs: *const u64, n: usize, point: *const u64,
                                    proof_out_jac: *mut u64, value_out: *mut u64) -> keaki_status;
    pub fn keaki_hip_group_encap_batch(g: *mut keaki_hip_group, com_aff: *const u64, tau_g2_aff: *const u64, points: *const u64, values: *const u64,
                                       r: *const u64, n: usize, ct_out_aff: *mut u64, gt_out: *mut u8, key_out: *mut u8, msg_len: usize) -> keaki_status;
    pub fn keaki_hip_group_decap_batch(g: *mut keaki_hip_group, proofs_aff: *const u64, cts_aff: *const u64, n: usize, gt_out: *mut u8, key_out: *mut u8,
                                       msg_len: usize) -> keaki_status;
    pub fn keaki_hip_group_encrypt_batch(g: *mut keaki_hip_group, com_aff: *const u64, tau_g2_aff: *const u64, points: *const u64, values: *const u64,
                                         r: *const u64, msgs: *const u8, n: usize, ct_out_aff: *mut u64, body_out: *mut u8, msg_len: usize) -> keaki_status;
    pub fn keaki_hip_group_decrypt_batch(g: *mut keaki_hip_group, proofs_aff: *const u64, cts_aff: *const u64, bodies: *const u8, n: usize,
                                         msgs_out: *mut u8, msg_len: usize) -> keaki_status;

    pub fn keaki_hip_group_fk_create(g: *mut keaki_hip_group, points_aff: *const u64, log2d: u32, omega_2d: *const u64, omega_2d_inv: *const u64,
                                     inv_2d: *const u64, out: *mut *mut keaki_hip_group_fk) -> keaki_status;
    pub fn keaki_hip_group_fk_open(g: *mut keaki_hip_group, fk: *mut keaki_hip_group_fk, coeffs: *const u64, proofs_out_aff: *mut u64) -> keaki_status;
    pub fn keaki_hip_group_fk_free(g: *mut keaki_hip_group, fk: *mut keaki_hip_group_fk);

    // ---- instrumentation
    pub fn keaki_hip_set_timing(ctx: *mut keaki_hip_ctx, enabled: i32) -> keaki_status;
    pub fn keaki_hip_last_msm_bucket_ms(ctx: *const keaki_hip_ctx) -> f32;
    pub fn keaki_hip_last_msm_total_ms(ctx: *const keaki_hip_ctx) -> f32;
    pub fn keaki_hip_last_msm_window_bits(ctx: *const keaki_hip_ctx) -> i32;
    pub fn keaki_hip_last_fk_ms(ctx: *mut keaki_hip_ctx, out3: *mut f32) -> keaki_status;

    // ---- test hooks
    pub fn keaki_hip_g2_prepare(ctx: *mut keaki_hip_ctx, g2_aff: *const u64, lines_out: *mut u64, lines_out_bytes: usize) -> keaki_status;
    pub fn keaki_hip_miller_loop_batch(ctx: *mut keaki_hip_ctx, g1_aff: *const u64, g2_aff: *const u64, n: usize, f_mont_out: *mut u64) -> keaki_status;
    pub fn keaki_hip_final_exp_batch(ctx: *mut keaki_hip_ctx, f_mont: *const u64, n: usize, gt_out: *mut u8) -> keaki_status;
    pub fn keaki_hip_selftest_field(ctx: *mut keaki_hip_ctx, blocks: u32, iters: u32, seed: u32, mismatches_out: *mut u64) -> keaki_status;
}
