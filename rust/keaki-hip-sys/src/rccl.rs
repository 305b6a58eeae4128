//! Raw bindings to `libkeaki_hip_rccl.so` (include/keaki_hip_rccl.h): the RCCL exchanges of the one-process-per-GPU form, enqueued on the
//! context's stream. Optional (`--features rccl`; `build.rs` then links `keaki_hip_rccl` from `KEAKI_HIP_LIB_DIR` as well).
use crate::{keaki_hip_ctx, keaki_hip_srs_g1, keaki_status};
use core::ffi::{c_char, c_void};

pub const KEAKI_ERR_RCCL: keaki_status = -6;

#[repr(C)]
pub struct keaki_hip_rccl {
    _private: [u8; 0],
}

extern "C" {
    pub fn keaki_hip_rccl_unique_id(out128: *mut u8) -> keaki_status;
    pub fn keaki_hip_rccl_create(ctx: *mut keaki_hip_ctx, id128: *const u8, rank: i32, world: i32, out: *mut *mut keaki_hip_rccl) -> keaki_status;
    pub fn keaki_hip_rccl_destroy(rc: *mut keaki_hip_rccl);
    pub fn keaki_hip_rccl_last_error(rc: *const keaki_hip_rccl) -> *const c_char;
    pub fn keaki_hip_rccl_msm_g1(rc: *mut keaki_hip_rccl, srs_chunk: *const keaki_hip_srs_g1, d_scalars: *const c_void, n: usize, d_out_jac: *mut c_void) -> keaki_status;
    pub fn keaki_hip_rccl_all_to_all(rc: *mut keaki_hip_rccl, d_send: *const c_void, d_recv: *mut c_void, bytes_per_peer: usize) -> keaki_status;
    pub fn keaki_hip_rccl_all_gather(rc: *mut keaki_hip_rccl, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: usize) -> keaki_status;
    pub fn keaki_hip_rccl_collective_status(rc: *mut keaki_hip_rccl, bad_rank: *mut i32) -> keaki_status;
}
