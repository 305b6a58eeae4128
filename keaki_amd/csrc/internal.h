// Internal (not installed) declarations shared by the translation units of libkeaki_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/keaki_hip.h"

namespace keaki_internal {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace keaki_internal

namespace keaki_internal {
// Tuning / diagnostic switches of a context. The environment is read ONCE, in keaki_hip_ctx_create (tune_from_env); afterwards only
// keaki_hip_ctx_set_option changes them (under the context lock). No other code in the library calls getenv.
struct Tuning {
  int msm_c = 0;                 // KEAKI_MSM_C / "msm_c": window bits of the generic MSM, 0 = choose_window
  int msm_c_shared = 0;          // KEAKI_MSM_C_SHARED / "msm_c_shared": window target of the SRS window tables, 0 = choose_window_shared
  int msm_short_tables = -1;     // KEAKI_MSM_SHORT_TABLES / "msm_short_tables": an MSM over less than half of an SRS with window tables uses them (1, and automatic = -1) or the generic path (0)
  int reduce_l = 0;              // KEAKI_REDUCE_L / "reduce_l": chunk length of the bucket reduction, 0 = automatic
  int part_shift = -1;           // KEAKI_PART_SHIFT / "part_shift": log2 of the bucket sort's bin count, -1 = automatic
  bool acc_u29 = true;           // KEAKI_ACC_U29 / "acc_u29": G1 bucket kernel in the 29-bit lazy limbs (A/B switch for profiling)
  bool acc_u29_g2 = true;        // KEAKI_ACC_U29_G2 / "acc_u29_g2"
  bool acc_prefetch = true;      // KEAKI_ACC_PREFETCH / "acc_prefetch": the G1 bucket kernel requests the next pair's table row an iteration ahead (A/B switch)
  bool acc_idxq = true;          // KEAKI_ACC_IDXQ / "acc_idxq": the G1 bucket kernel reads its index stream by aligned 16-byte quads through a lane-private LDS slot (0: one 4-byte load per entry, as until round 5; A/B switch)
  bool cs_masked = true;         // KEAKI_CS_MASKED / "cs_masked": pass 2 of the bucket sort skips empty gather slots under the exec mask (0: every empty slot counts into a dummy word per lane, as in round 4; A/B switch: 0.84 -> 0.70 ms at 2^24)
  bool acc_nt = false;           // KEAKI_ACC_NT / "acc_nt": non-temporal loads of the table rows in the G1 bucket kernel
  bool fk_uniform = true;        // KEAKI_FK_UNIFORM / "fk_uniform": sliding-window ladder in the wave-uniform FK23 stages
  bool fk_gtab = true;           // KEAKI_FK_GTAB / "fk_gtab": window tables of the per-lane-scalar ladders in a lane-contiguous workspace (0: private memory)
  bool fk_radix4 = true;        // KEAKI_FK_RADIX4 / "fk_radix4": two wave-uniform stages in one radix-4 pass (three doubling chains instead of four)
  bool fk_addsub29 = true;      // KEAKI_FK_ADDSUB29 / "fk_addsub29": the butterflies' add + subtract in the lazy limbs, shared products once (A/B switch)
  bool fb_occ1 = false;          // KEAKI_FB_OCC1 / "fb_occ1": one wave per SIMD for the G2 fixed-base kernel at any batch size
  int pair_wide_max = -1;        // KEAKI_PAIR_WIDE_MAX / "pair_wide_max": pairing batches up to this size run the twelve-lanes-per-pairing kernel; -1 = automatic, 0 = never
  bool pair_two_waves = true;    // KEAKI_PAIR_TWO_WAVES / "pair_two_waves": up to 1,024 pairings with lines on the fly run the line functions on a second wave (A/B switch)
  int gt_wb_b = 0;               // KEAKI_GT_WB_B / "gt_wb_b": window bits of the table of e(g1, g2), 0 = automatic (20 / 16)
  long long encap_gt = -1;       // KEAKI_ENCAP_GT / "encap_gt": batch size from which encap takes the GT fixed-base path; -1 = automatic (always, since round 4)
  bool host_prefault = true;     // KEAKI_HOST_PREFAULT / "host_prefault": first-touch (write zeros into) and madvise(MADV_HUGEPAGE) the caller's OUTPUT arrays while the kernels run, on helper threads for chunked batches; 0 = the library never touches caller memory except through the device copies
  bool pipe_chunks = true;       // KEAKI_PIPE_CHUNKS / "pipe_chunks": host-pointer batches (KEM, MSM) run as chunk pipelines over a copy stream; 0 = upload, kernels, download in that order on the context's stream
  int msm_pipe_chunks = -1;      // KEAKI_MSM_PIPE_CHUNKS / "msm_pipe_chunks": chunks of a host-pointer MSM (upload under the kernels); -1 = automatic, 0 / 1 = one copy in front, k = k chunks at any length
  long long msm_pipe_min = 1 << 20;   // KEAKI_MSM_PIPE_MIN / "msm_pipe_min": automatic chunking from this many scalars on
  int msm_pipe_growth = 160;     // KEAKI_MSM_PIPE_GROWTH / "msm_pipe_growth": size of chunk j + 1 in percent of chunk j (100 = equal chunks; round 6: 160, measured best or tied at 2^20 .. 2^24, profiles/r06_msm_pipe_growth.txt; 140 until round 5)
#ifdef KEAKI_DIAG
  unsigned diag_row_mask = 0;    // "diag_row_mask" (ONLY in the diagnostic build, `make -C keaki_amd/csrc diag`; never in libkeaki_hip.so): the table-row index of every (row, sign) entry of the bucket-ordered stream is ANDed with this mask before the bucket kernel runs, so its gathers hit a table of (mask + 1) x 64 B -- the arithmetic, the instruction stream and the kernel binary stay the shipped ones, the RESULT IS WRONG by construction (bench_tools/r6_bucket_clock_diag.py)
#endif
  size_t alloc_limit = 0;        // keaki_hip_debug_set_alloc_limit: single allocations above it fail with KEAKI_ERR_OOM; 0 = none
};
}  // namespace keaki_internal

struct keaki_hip_ctx {
  int device = 0;
  uint32_t n_cu = 256;           // compute units of the device (grid size of the persistent kernels)
  keaki_internal::Tuning tune;
  // bytes this context allocated and still holds, by class (keaki_hip_ctx_memory): SRS window tables + FK23 transforms of handles built
  // through it | grow-only workspaces | GT / fixed-base tables of encapsulate
  std::atomic<size_t> mem_tables{0};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::recursive_mutex mu;   // recursive: host-pointer entry points hold it across stage -> *_dev -> download
  std::string err;
  // grow-only workspaces (all used in stream order)
  keaki_internal::DevBuf digits, hist, offsets, cursor, sorted, buckets, acc29, partials, wsums, bsums, tmp_a, tmp_b, tmp_c, io_a, io_b, io_c, io_d, io_e;
  // fixed-base window tables for encapsulate: generator tables are built once per context, the C / [tau]_2 tables per batch
  keaki_internal::DevBuf fb_bases, fb_g1_gen, fb_g2_gen, fb_com, fb_tau, perm, g2gen_lines, gt_tab_a, gt_tab_b, gt_base, heavy;
  bool gt_b_ready = false;
  bool gt_a_valid = false;
  bool gt_a_pending_aux = false;          // the A-table build on aux_stream has not been waited for by `stream` yet (api.hip: encap_impl)
  bool gt_b_fallback = false;             // the wide table of B did not fit once: stay at 16 bits
  uint64_t seen_com[8] = {};              // commitment of the last encap call and how many consecutive calls carried it
  uint32_t seen_com_runs = 0;
  uint32_t gt_a_wb = 0, gt_b_wb = 0;      // window widths of the GT tables in gt_tab_a / gt_tab_b
  // line tables of 2^s g2, s < 320 (built once per context): e(C, g2)^(2^s) = e(C, 2^s g2) -- the powers behind a commitment's GT table are
  // pairings of ONE point with fixed second arguments, no doubling chain of C in front of them
  keaki_internal::DevBuf g2pow_lines, g2pow_pts;
  bool g2pow_ready = false;
  keaki_internal::DevBuf pair_ws;                   // per-item slots of the final exponentiation (pairing.hip.h)
  keaki_internal::DevBuf fk_tab;                    // window tables of the per-lane-scalar ladders of FK23: 1 KB per lane of a launch (64 x 16 B), at most 2 GB (fft_g1.hip)
  keaki_internal::DevBuf verify_io;                 // kzg verify: small in/out block
  bool verify_ready = false;
  bool verify_tables_ready = false;       // 8-bit window tables of g1 (fbs_g1_gen) and g2 (fbs_g2_gen) for the reference-form verify              // set only after every init step of kzg verify succeeded
  bool fb_tau_valid = false;          // window table of [tau]_2 (encap ciphertext side) is for this point
  uint64_t fb_tau_pt[16] = {};
  uint64_t gt_a_com[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // commitment the cached A-table belongs to
  bool g2gen_lines_ready = false;
  bool fb_ready = false;
  keaki_internal::DevBuf fbs_g2_gen, fbs_tau, fbs_g1_gen;     // small (8-bit) tables of g2 and [tau]_2 for batches below 256 items
  bool fbs_ready = false, fbs_tau_valid = false;
  uint64_t fbs_tau_pt[16] = {};
  // instrumentation
  bool timing = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float last_bucket_ms = -1.f, last_total_ms = -1.f;
  int last_c = 0;
  bool timing_pending = false;
  // FK23 openings: [start | after the 2d pointwise products | after the two size-d group transforms | after the affine conversion]
  hipEvent_t fk_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool fk_timing_pending = false;
  float last_fk_ms[3] = {-1.f, -1.f, -1.f};      // pointwise products, butterfly stages (k_g1_fft_stage_map), whole device pipeline
  // host-pointer batches run in chunks (api.hip: pipelined): uploads and downloads of the neighbouring chunks on a stream of their own.
  // Created on first use. [in: chunk staged | done: chunk computed], one pair per buffer half
  hipStream_t aux_stream = nullptr;              // latency-bound side jobs (api.hip: the GT table of a new commitment), with [go | done]
  hipEvent_t aux_ev[2] = {nullptr, nullptr};
  hipStream_t copy_stream = nullptr;
  hipEvent_t pipe_in[2] = {nullptr, nullptr}, pipe_done[2] = {nullptr, nullptr};
  hipEvent_t open_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // chunked kzg_open: [chunk uploaded x 2 | chunk's quotient ready x 2 | start]
};

namespace keaki_internal {

// Makes `device` the calling thread's current HIP device for the scope and puts the caller's own choice back on exit: a host application
// (PyTorch, another HIP library) that shares the thread never sees its current device change behind its back.
struct DeviceScope {
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(int device) {
    if (device < 0) return;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) return;
    ok = hipSetDevice(device) == hipSuccess;
    if (ok) prev = cur;
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

keaki_status fail(keaki_hip_ctx* ctx, keaki_status code, const char* fmt, ...);
keaki_status reserve(keaki_hip_ctx* ctx, DevBuf& b, size_t bytes);
// hipMalloc behind the library's one allocation gate. keaki_hip_debug_set_alloc_limit(ctx, bytes) makes every single allocation of this
// context above that size fail with KEAKI_ERR_OOM, which is how the tests exercise the optional-memory fallbacks (SRS window tables,
// the wide GT table).
keaki_status dev_alloc(keaki_hip_ctx* ctx, void** p, size_t bytes);
keaki_status launch_check(keaki_hip_ctx* ctx, const char* what);
inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

#define HIP_TRY(ctx, call)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return keaki_internal::fail(ctx, e_ == hipErrorOutOfMemory ? KEAKI_ERR_OOM : KEAKI_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                           \
  } while (0)
#define ST_TRY(call)                 \
  do {                               \
    keaki_status s_ = (call);        \
    if (s_ != KEAKI_OK) return s_;   \
  } while (0)

// Chunked form of one MSM (the host-pointer entries, api.hip): the (scalar, point) pairs are cut into point-range chunks; every chunk
// runs its own tile sort -> chunk sort -> size order -> bucket pass, the bucket pass going on from what the earlier chunks left in the
// buckets (msm.hip.h: Acc29 / `cont`), and ONE reduction tail closes the call. `stage(j)` is called right before the kernels of chunk j
// are enqueued: the caller uploads that chunk's scalars there (on its copy stream) and makes the context's stream wait for them, so the
// upload of chunk j + 1 runs under the kernels of chunk j. The sum is the same group element whatever the cut (exact arithmetic).
struct MsmPipe {
  std::vector<std::pair<size_t, size_t>> ranges;     // chunk j = pairs [first, first + second); the ranges partition [0, n), in ANY order
  std::function<keaki_status(size_t)> stage;         // (the quotient of `open` is produced from the top coefficient down: its chunks come last-first)
};

// launchers implemented in the kernel translation units (all enqueue on ctx->stream, no sync)
keaki_status msm_g1_run(keaki_hip_ctx* ctx, const void* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac,
                        const void* d_table = nullptr, int c_table = 0, const MsmPipe* pipe = nullptr);
keaki_status msm_g1_precompute_run(keaki_hip_ctx* ctx, const void* d_points, size_t N, int* c_table_out, size_t* table_bytes_out, void** d_table_out);
keaki_status msm_g2_run(keaki_hip_ctx* ctx, const void* d_points, size_t srs_len, const void* d_scalars, size_t n, void* d_out_jac,
                        const void* d_table = nullptr, int c_table = 0, const MsmPipe* pipe = nullptr);
keaki_status msm_g2_precompute_run(keaki_hip_ctx* ctx, const void* d_points, size_t N, int* c_table_out, size_t* table_bytes_out, void** d_table_out);
keaki_status g1_sum_run(keaki_hip_ctx* ctx, const void* d_points_jac, size_t k, void* d_out_jac);
keaki_status g1_mul_batch_run(keaki_hip_ctx* ctx, const void* d_pts, int stride, const void* d_scalars, size_t n, void* d_out);
keaki_status g2_mul_batch_run(keaki_hip_ctx* ctx, const void* d_pts, int stride, const void* d_scalars, size_t n, void* d_out);
keaki_status encap_g1_run(keaki_hip_ctx* ctx, const void* d_com, const void* d_values, const void* d_r, size_t n, void* d_out);
size_t pairing_launch_items();      // items per k_pairing launch (the slots of the final exponentiation bound it)
keaki_status pairing_run(keaki_hip_ctx* ctx, const void* d_g1, const void* d_g2, int g2_stride, size_t n, void* d_gt, const void* d_fixed_lines = nullptr,
                         uint32_t lines_stride = 0);
uint32_t g2_prepared_lines();                 // Line entries of one table
// kzg verify: A = com - value g1, Q = [tau]_2 - point g2 from the wb-bit window tables of the generators (sixteen lanes per sum)
keaki_status verify_points_run(keaki_hip_ctx* ctx, const void* d_tab_g1, const void* d_tab_g2, uint32_t wb, const void* d_com, const void* d_tau_g2,
                               const void* d_value, const void* d_point, void* d_out_a, void* d_out_q);
size_t g2_prepared_bytes();
// out[i] = e(P_(i * p_stride), Q_i) as 12 Fq in the 2^261 form, Q_i given by its line table d_lines + i * lines_stride lines
keaki_status pairing_raw_fixed_run(keaki_hip_ctx* ctx, const void* d_g1, uint32_t p_stride, size_t n, const void* d_lines, uint32_t lines_stride, void* d_out);
size_t gt_table_bytes(uint32_t wb);
uint32_t gt_table_powers(uint32_t wb);      // powers of two a table needs: wb * windows
keaki_status gt_table_run(keaki_hip_ctx* ctx, const void* d_pows, void* d_table, uint32_t wb);   // d_pows: base^(2^s), 12 Fq each
// gt[i] = serialize(acc_in[i] (or one) * A^(r_i) (tab_a given) * B^(-beta_i r_i) (tab_b given)); acc_out given: the product stays raw (12 Fq per item)
keaki_status gt_encap_exp_run(keaki_hip_ctx* ctx, const void* d_tab_a, uint32_t wb_a, const void* d_tab_b, uint32_t wb_b, const void* d_betas,
                              const void* d_rs, size_t n, void* d_gt, const void* d_acc_in = nullptr, void* d_acc_out = nullptr);
keaki_status miller_only_run(keaki_hip_ctx* ctx, const void* d_g1, const void* d_g2, size_t n, void* d_out);
keaki_status final_exp_only_run(keaki_hip_ctx* ctx, const void* d_in, size_t n, void* d_gt);
keaki_status g2_prepare_run(keaki_hip_ctx* ctx, const void* d_q, void* d_lines, uint32_t n_points = 1);   // line sequence of a fixed Q (2^261 form: internal)
keaki_status lines_to256_run(keaki_hip_ctx* ctx, const void* d_lines261, void* d_lines256);   // the same table in the ABI's 2^256 form (test hook)
keaki_status blake3_gt_run(keaki_hip_ctx* ctx, const void* d_gt, size_t n, void* d_key, size_t msg_len, bool xor_into = false);   // xor_into: key ^= in place (the DEM)
keaki_status g2_generator_to(keaki_hip_ctx* ctx, void* d_dst);  // writes the affine G2 generator (128 B)
keaki_status g1_generator_to(keaki_hip_ctx* ctx, void* d_dst);  // affine G1 generator (64 B)
size_t fb_table_entries(uint32_t wb);                                                          // window-table entries per base at window width wb
keaki_status g1_fb_table_run(keaki_hip_ctx* ctx, const void* d_base, void* d_table, uint32_t wb);    // table[j * entries + d] = d 2^(wb j) base
keaki_status g2_fb_table_run(keaki_hip_ctx* ctx, const void* d_base, void* d_table, uint32_t wb);
keaki_status g2_pow2_multiples_run(keaki_hip_ctx* ctx, const void* d_base, uint32_t count, void* d_out);   // out[s] = 2^s base (affine), lane s doubles s times
keaki_status encap_g1_fixed_run(keaki_hip_ctx* ctx, const void* d_tab_a, uint32_t wb_a, const void* d_tab_b, uint32_t wb_b, const void* d_xs, const void* d_rs, size_t n, void* d_out);
keaki_status encap_g2_fixed_run(keaki_hip_ctx* ctx, const void* d_tab_a, uint32_t wb_a, const void* d_tab_b, uint32_t wb_b, const void* d_xs, const void* d_rs, size_t n, void* d_out, bool share_simds = false);
keaki_status g1_curve_check_run(keaki_hip_ctx* ctx, const void* d_pts, size_t n, void* d_bad2);   // d_bad2: u64 count, u64 first index
keaki_status g2_curve_check_run(keaki_hip_ctx* ctx, const void* d_pts, size_t n, void* d_bad2);
// top_is_carry: d_c[n - 1] is not a coefficient but the suffix value carried in from the chunk above (its quotient slot is left alone)
keaki_status open_quotient_run(keaki_hip_ctx* ctx, const void* d_c, size_t n, const uint64_t* z, void* d_q, void* d_value, void* d_work, bool top_is_carry = false);
size_t open_quotient_work_bytes(size_t n);           // size of d_work for n coefficients
keaki_status fr_fft_run(keaki_hip_ctx* ctx, void* d_data, uint32_t log2n, const uint64_t* omega, const uint64_t* scale_or_null, void* d_tw);
keaki_status open_fk_poly_run(keaki_hip_ctx* ctx, const void* d_srs, void** hat_s_cache, int* hat_s_log2d, uint32_t log2d, const void* d_p,
                              const uint64_t* omega_2d, const uint64_t* omega_2d_inv, const uint64_t* inv_2d, void* d_fr_work, void* d_g_work,
                              void* d_proofs_aff);
keaki_status fk_precompute_run(keaki_hip_ctx* ctx, const void* d_srs, void** hat_s_cache, int* hat_s_log2d, uint32_t log2d, const uint64_t* omega_2d,
                               void* d_tw_work);
keaki_status fk_hat_s_run(keaki_hip_ctx* ctx, const void* d_srs, uint32_t log2d, const void* d_tw2d, void* d_hat_s);
keaki_status open_fk_run(keaki_hip_ctx* ctx, const void* d_hat_s, uint32_t log2d, const void* d_hat_a, const void* d_tw2d, const void* d_tw2d_inv,
                         void* d_work, void* d_proofs_aff);
// FK23 sharded over 2^rho ranks (fft_g1.hip): this rank's plan and device buffers (owned by the api layer)
struct FkShard {
  uint32_t log2d = 0, rho = 0, rank = 0;
  uint64_t omega[4], omega_inv[4], inv_2d[4];
  void *tw = nullptr, *twi = nullptr;     // omega_2d^k, omega_2d^-k, k < d
  void *hat_a = nullptr, *coeffs = nullptr;   // 2d Fr, d Fr
  void *hat_s = nullptr, *work = nullptr;     // 2d / R Jacobian points each
  void *e = nullptr;                          // d / R Jacobian points: the even half of the products, from step 0 to step 2
  bool tables_ready = false, hat_s_ready = false;
};
keaki_status fk_shard_setup_run(keaki_hip_ctx* ctx, FkShard& fk, const void* d_srs, int step, void* d_send, void* d_recv);
keaki_status fk_shard_open_run(keaki_hip_ctx* ctx, FkShard& fk, int step, void* d_send, void* d_recv, void* d_out_aff);
keaki_status selftest_u29_run(keaki_hip_ctx* ctx, uint32_t blocks, uint32_t iters, uint32_t seed, void* d_mismatches);
keaki_status selftest_field_run(keaki_hip_ctx* ctx, uint32_t blocks, uint32_t iters, uint32_t seed, void* d_mismatches);

}  // namespace keaki_internal
