/* keaki_hip.h -- C ABI of libkeaki_hip.so: the MI355X (gfx950) backend for keaki's BN254 hot path.
 *
 * keaki (the reference) has no FFI of its own; its "operator API" is the generic Rust functions
 * kzg::commit/open/verify, kem::encapsulate/decapsulate and vec::vec_encrypt/vec_decrypt. Every
 * expensive step inside them is a call into arkworks. The entry points below are what a
 * feature-gated Rust shim binds in place of those arkworks calls (INTEGRATION.md shows the shim);
 * each one cites the reference line it replaces.
 *
 * Conventions
 *  - Every function returns a keaki_status (0 = ok, < 0 = error); nothing unwinds or aborts across
 *    the boundary. keaki_hip_last_error(ctx) returns a human-readable message for the last failure.
 *  - The caller owns every buffer. *_dev entry points take DEVICE pointers (memory already resident
 *    in HBM, e.g. allocated by the caller's HIP runtime) and enqueue on the ctx stream without
 *    synchronising unless stated; the plain entry points take HOST pointers, copy in/out and
 *    synchronise.
 *  - Field elements are little-endian 64-bit limbs in Montgomery form (R = 2^256) EXACTLY as ark-ff
 *    0.4.2 holds them in `Fp.0.0`, so the shim copies limbs without conversion:
 *        Fr, Fq        u64[4]
 *        G1 affine     u64[8]   (x, y);                      identity = all-zero words
 *        G1 Jacobian   u64[12]  (x, y, z) normalised: z = R (Montgomery one), identity = (R, R, 0)
 *        G2 affine     u64[16]  (x.c0, x.c1, y.c0, y.c1);    identity = all-zero words
 *        G2 Jacobian   u64[24]  normalised likewise
 *        GT            384 bytes = ark-serialize `serialize_uncompressed` of Fq12:
 *                      c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1, each canonical 32-byte LE.
 *    (arkworks' in-memory `Affine { x, y, infinity: bool }` is repr(Rust); the shim must write x,y
 *    explicitly and map `infinity` to the all-zero encoding.)
 *  - A ctx is bound to one GPU. Multi-GPU, two ways: (a) in one process, keaki_hip_group_* below (one ctx and
 *    one host thread per GPU inside the library, partial sums through host memory); (b) one process per GPU:
 *    each rank runs msm on its contiguous chunk of (scalar, point) pairs, the 96-byte partial sums are
 *    exchanged with RCCL all-gather by the caller, and keaki_hip_g1_sum_dev adds them.
 *  - Thread-safety: a ctx serialises calls internally (one recursive mutex, held from the first staging
 *    copy of a host-pointer call to its last download, so the shared staging buffers belong to one
 *    call at a time); calls from several host threads on one ctx are safe and run one after another.
 *    Use one ctx per host thread for concurrency. An SRS handle (keaki_hip_srs_*) is device memory, not
 *    context state: every ctx on the same device may pass it to msm / kzg_open / open_fk, so N threads with a
 *    ctx each share ONE copy of the points, the window tables and the FK23 transform (the handle locks its
 *    lazily built members; a handle from another device is refused with KEAKI_ERR_BAD_ARG). Handles must not
 *    be freed while another thread still uses them.
 *  - Memory: keaki_hip_ctx_memory reports what a ctx holds. Per ctx: MSM workspaces (2.4 GB at 2^24 points),
 *    pairing slots (0.48 GB), the GT tables of encapsulate (0.2 + 0.2 GB from the first encapsulation on, 2.6 + 0.2 GB once a batch >= 2^16 ran;
 *    6 MB of line tables of the multiples of g2). A ctx creates up to three non-blocking streams (its own unless one was passed in, one for
 *    the copies of chunked host-array batches, one for the table of a new commitment). Per SRS
 *    handle (shared by every ctx of the device): points (64 B each) + window tables (W x 64 B each: 12.9 GB at
 *    2^24) + the FK23 transform (192 B per opening). All optional tables fall back when they do not fit.
 *  - Current device: every call makes its ctx's GPU the calling thread's current HIP device for the duration of the
 *    call and puts the caller's own choice back before it returns; the library never calls hipDeviceReset and
 *    creates only non-blocking streams, so it can share a thread and a GPU with PyTorch or another HIP library.
 *  - The library reads the environment only inside keaki_hip_ctx_create (initial values of the tuning switches
 *    of keaki_hip_ctx_set_option); nothing on a call path calls getenv.
 */
#ifndef KEAKI_HIP_H
#define KEAKI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t keaki_status;
enum {
  KEAKI_OK = 0,
  KEAKI_ERR_BAD_ARG = -1,      /* null pointer, n out of range, n > srs length ...            */
  KEAKI_ERR_HIP = -2,          /* a HIP runtime call failed (message has hipGetErrorString)   */
  KEAKI_ERR_OOM = -3,          /* device allocation failed                                    */
  KEAKI_ERR_NO_DEVICE = -4,    /* no gfx950 device visible                                    */
  KEAKI_ERR_TOO_LARGE = -5     /* polynomial longer than the SRS: KZGError::PolynomialTooLarge */
};

typedef struct keaki_hip_ctx keaki_hip_ctx;
typedef struct keaki_hip_srs_g1 keaki_hip_srs_g1; /* device-resident [tau^i]_1, affine */
typedef struct keaki_hip_srs_g2 keaki_hip_srs_g2; /* device-resident G2 bases, affine  */

/* ---- context ------------------------------------------------------------------------------- */
/* device: HIP device ordinal. stream: the hipStream_t every call of this ctx enqueues on:
 *   KEAKI_HIP_STREAM_PRIVATE  (NULL)  the ctx creates its own NON-BLOCKING stream: unordered against the caller's streams AND against the
 *                                      legacy default stream -- order *_dev calls with keaki_hip_synchronize or events;
 *   KEAKI_HIP_STREAM_LEGACY   (= hipStreamLegacy)  the legacy default ("null") stream of the device, which a NULL handle cannot name here;
 *   any other value                    a stream the caller created (e.g. the handle of a torch.cuda.Stream): *_dev calls are then ordered
 *                                      with everything else the caller enqueues there (RCCL collectives, copies). */
#define KEAKI_HIP_STREAM_PRIVATE ((void*)0)
#define KEAKI_HIP_STREAM_LEGACY ((void*)1)
keaki_status keaki_hip_ctx_create(int32_t device, void* stream, keaki_hip_ctx** out);
void keaki_hip_ctx_destroy(keaki_hip_ctx* ctx);
const char* keaki_hip_last_error(const keaki_hip_ctx* ctx); /* ctx may be NULL: last create error */
keaki_status keaki_hip_synchronize(keaki_hip_ctx* ctx);
/* the hipStream_t every call of this ctx enqueues on (for callers that order their own work -- RCCL collectives, copies -- with it) */
void* keaki_hip_ctx_stream(const keaki_hip_ctx* ctx);
/* the HIP device ordinal the ctx is bound to (-1 for NULL): for companions that make their own HIP / RCCL calls beside it (keaki_hip_rccl.h) */
int32_t keaki_hip_ctx_device(const keaki_hip_ctx* ctx);
/* "keaki-hip <ver> (gfx950) src=msm:<hash>,pairing:<hash>,fk:<hash>": the hashes of the kernel sources the binary was built from */
const char* keaki_hip_version(void);
/* keaki_hip_last_error: the returned string is a copy private to the calling thread (valid until its next call of this function). */
/* Tuning / A-B switches of a context (profiling and tests; defaults are what ships). Initial values come from the environment variable
 * KEAKI_<NAME> at keaki_hip_ctx_create; afterwards only this call changes them. Names: "msm_c", "msm_c_shared" (window bits, 0 = automatic),
 * "reduce_l", "part_shift", "acc_u29", "acc_u29_g2", "acc_nt", "acc_prefetch", "acc_idxq" (the G1 bucket kernel reads its index stream by aligned
 * 64-byte groups through a lane-private LDS slot; 0 = one 4-byte load per entry as until round 5), "cs_masked", "fk_uniform", "fk_gtab", "fk_addsub29", "fk_radix4", "fb_occ1", "gt_wb_b" (window bits of the table of
 * e(g1, g2), 0 = automatic; a change rebuilds the table on the next use), "encap_gt" (batch size from which encap_batch takes the GT
 * fixed-base path and below which -- for a commitment that has no table yet -- it runs a pairing per item; -1 = automatic: always the GT path),
 * "pair_wide_max" (pairing batches up to this size run the twelve-lanes-per-pairing kernel; -1 = automatic (4096), 0 = never), "pair_two_waves"
 * (up to 1,024 pairings: the line side of the Miller loop on a second wave; 0 = one wave), "msm_short_tables" (an MSM over less than half of an SRS
 * with window tables: 0 = the generic path as before round 4, 1 / -1 = the tables).
 * What the library does with HOST memory of the caller, and its off switches (round 5):
 *   "host_prefault" (default 1): while the kernels of a host-array call run, the library first-touches the caller's OUTPUT arrays (writes a zero
 *       into one byte per 4 KB page, the whole range being overwritten by the download that follows; outputs >= 1 MB only) and asks for transparent
 *       huge pages on them (madvise(MADV_HUGEPAGE)); chunked batches do this on up to three helper std::threads that live for the duration of
 *       the call. 0: no helper threads, no write into and no madvise on caller memory -- every byte that appears in an output array was put there
 *       by a device-to-host copy. Results are identical; downloads into pages that do not exist yet are slower (160 MB: 30 ms instead of 3).
 *   "pipe_chunks" (default 1): host-array batches (encap / decap / encrypt / decrypt from 2 x 65,536 items on; MSMs from "msm_pipe_min" scalars on;
 *       kzg_open from 2^21 coefficients on) run as chunk pipelines over a copy stream (kzg_open: and an auxiliary stream) of the context's own.
 *       0: upload, kernels, download, in that order, on the context's stream -- for every AUTOMATIC choice. An explicit "msm_pipe_chunks" >= 2
 *       names the chunked path itself and takes precedence: a caller who wants no other stream sets "pipe_chunks" = 0 and leaves
 *       "msm_pipe_chunks" at -1 (or sets it to 0).
 *   "msm_pipe_chunks" (-1 = automatic: 6 chunks from 2^22 scalars on, 4 from 2^21, 3 from "msm_pipe_min" = 2^20 on; 0 / 1 = one copy in front; k >= 2 = k chunks at
 *       any length), "msm_pipe_growth" (size of chunk j + 1 in percent of chunk j, default 160: a short first chunk starts the device early).
 *   A host-array call that FAILS (status != KEAKI_OK) leaves its output arrays unspecified: any prefix may hold results, and with
 *   "host_prefault" = 1 pages may hold the zeros of the first touch. Input arrays are never written.
 * Unknown name -> KEAKI_ERR_BAD_ARG. */
keaki_status keaki_hip_ctx_set_option(keaki_hip_ctx* ctx, const char* name, int64_t value);
/* Test hook: every single device allocation of this ctx above `bytes` fails with KEAKI_ERR_OOM (0 = no limit). This is how the tests
 * exercise the optional-memory fallbacks (SRS window tables, the wide GT table); nothing in the library sets it. */
keaki_status keaki_hip_debug_set_alloc_limit(keaki_hip_ctx* ctx, size_t bytes);
/* Device memory this ctx holds right now, bytes: out4[0] = window tables + FK23 transforms of SRS handles built through this ctx (the
 * points themselves belong to the caller's upload), [1] = grow-only workspaces, [2] = fixed-base / GT tables of encapsulate, [3] = total. */
keaki_status keaki_hip_ctx_memory(keaki_hip_ctx* ctx, size_t* out4);
/* Gives back every workspace and every fixed-base / GT table of the context (out4[1] and out4[2] drop to 0): for a long-lived process after
 * an unusually large call. Everything is rebuilt on demand; results never change. SRS handles (and their tables) are the caller's. */
keaki_status keaki_hip_ctx_trim(keaki_hip_ctx* ctx);

/* ---- SRS: replaces KZGSetup::g1_aff (src/kzg.rs:22-29, built at :63) -------------------------- */
keaki_status keaki_hip_srs_g1_upload(keaki_hip_ctx* ctx, const uint64_t* points_aff, size_t n, keaki_hip_srs_g1** out);
/* wraps caller-owned device memory holding n affine points (not freed by _free) */
keaki_status keaki_hip_srs_g1_wrap_dev(keaki_hip_ctx* ctx, const void* d_points_aff, size_t n, keaki_hip_srs_g1** out);
/* non-owning view of points [offset, offset + n) of `srs` (which must outlive it): the contiguous chunk one rank owns when commit's MSM
 * (src/kzg.rs:98) is sharded by point range over several GPUs; free it with keaki_hip_srs_g1_free. keaki_hip_srs_g1_precompute on
 * the view tabulates the chunk only. */
keaki_status keaki_hip_srs_g1_slice(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, size_t offset, size_t n, keaki_hip_srs_g1** out);
size_t keaki_hip_srs_g1_len(const keaki_hip_srs_g1* srs);
/* One-time precomputation for a fixed SRS (KZG bases never change): builds the window tables
 * table[w][i] = 2^(bit offset of window w) * P_i (affine, W x n x 64 bytes of HBM; W = 12..14 for n >= 2^20) so that all
 * windows of an MSM share ONE bucket set and the per-window reduction / Horner doublings disappear. Later
 * keaki_hip_msm_g1* calls on this handle with n > len/2 use the tables; results are identical. Optional. */
keaki_status keaki_hip_srs_g1_precompute(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, size_t* table_bytes_out);
void keaki_hip_srs_g1_free(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs);
keaki_status keaki_hip_srs_g2_upload(keaki_hip_ctx* ctx, const uint64_t* points_aff, size_t n, keaki_hip_srs_g2** out);
keaki_status keaki_hip_srs_g2_wrap_dev(keaki_hip_ctx* ctx, const void* d_points_aff, size_t n, keaki_hip_srs_g2** out);
/* window tables of a fixed G2 basis, as keaki_hip_srs_g1_precompute (W x n x 128 bytes): removes the per-window Horner doublings of a G2 MSM */
keaki_status keaki_hip_srs_g2_precompute(keaki_hip_ctx* ctx, keaki_hip_srs_g2* srs, size_t* table_bytes_out);
void keaki_hip_srs_g2_free(keaki_hip_ctx* ctx, keaki_hip_srs_g2* srs);

/* ---- MSM: replaces <E::G1 as VariableBaseMSM>::msm_unchecked(&setup.g1_aff, p) (src/kzg.rs:98) -
 * out = sum_{i<n} scalars[i] * srs[i]; n <= len(srs) (zip-truncation as msm_unchecked).
 * n > len(srs) -> KEAKI_ERR_TOO_LARGE (the check of src/kzg.rs:93-95 lives in the caller, this is a guard).
 * out_jac: u64[12] normalised Jacobian.
 * Host-pointer form (what kzg::commit calls, src/kzg.rs:89-101: the polynomial lives in host memory): from 2^20 scalars on the vector is
 * uploaded in growing point-range chunks on the context's copy stream while the sort and bucket kernels of the chunk before run; every
 * bucket pass goes on from the state the earlier chunks left, one reduction closes the call (2^24 scalars: 18.1 ms against 28 ms with
 * the copy in front and 16.6 ms with resident scalars). The result is the same group element whatever the cut. Pageable and pinned
 * sources both work; the call returns when out_jac holds the result and no copy reads `scalars` any more. */
keaki_status keaki_hip_msm_g1(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, const uint64_t* scalars, size_t n, uint64_t* out_jac);
/* device-resident scalars and output (d_out_jac: 96 bytes of device memory); asynchronous on the ctx stream */
keaki_status keaki_hip_msm_g1_dev(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, const void* d_scalars, size_t n, void* d_out_jac);
/* same over G2 (no call site in keaki; north_star asks for it; out u64[24]) */
keaki_status keaki_hip_msm_g2(keaki_hip_ctx* ctx, const keaki_hip_srs_g2* srs, const uint64_t* scalars, size_t n, uint64_t* out_jac);
keaki_status keaki_hip_msm_g2_dev(keaki_hip_ctx* ctx, const keaki_hip_srs_g2* srs, const void* d_scalars, size_t n, void* d_out_jac);
/* out = sum of k normalised-Jacobian G1 points (combines per-GPU partial sums after the all-gather) */
keaki_status keaki_hip_g1_sum_dev(keaki_hip_ctx* ctx, const void* d_points_jac, size_t k, void* d_out_jac);
keaki_status keaki_hip_g1_sum(keaki_hip_ctx* ctx, const uint64_t* points_jac, size_t k, uint64_t* out_jac);

/* ---- FK23 batch openings: replaces kzg::open_fk (src/kzg.rs:157-203; caller src/vec.rs:40) ------------------------------
 * All d = 2^log2d opening proofs of the polynomial p (d coefficients) at the d-th roots of unity, in O(d log d) group operations
 * (three G1 FFTs + 2d scalar-mults on the GPU). The scalar-field inputs are prepared by the caller (keaki's own host-side work):
 *   hat_a[2d]     = DFT_2d(0, ..., 0, p_0, ..., p_{d-1}) * (2d)^-1        (the 1/2d of the inverse transform folded in)
 *   tw_2d[d]      = omega_2d^k,  tw_2d_inv[d] = omega_2d^-k   (k < d),    tw_d[d/2] = omega_d^k   (k < d/2; not read any more, may be NULL)
 * where omega_N is ark-poly's Radix2EvaluationDomain generator of order N. Uses srs[0..d). proofs_out_aff: d affine points.
 * The SRS-only transform hat_s = DFT_2d(reversed SRS) is computed on the first call for a given d and cached in the handle. */
keaki_status keaki_hip_open_fk(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, uint32_t log2d, const uint64_t* hat_a, const uint64_t* tw_2d,
                               const uint64_t* tw_2d_inv, const uint64_t* tw_d, uint64_t* proofs_out_aff);

/* Same, from the coefficients themselves: hat_a and all twiddle tables are derived on the device (row f-4 of the scope table), the
 * caller passes only omega_2d (the order-2d root of ark-poly's Radix2EvaluationDomain), its inverse and (2d)^-1. */
keaki_status keaki_hip_open_fk_poly(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, uint32_t log2d, const uint64_t* coeffs, const uint64_t* omega_2d,
                                    const uint64_t* omega_2d_inv, const uint64_t* inv_2d, uint64_t* proofs_out_aff);
/* Optional, setup time: tabulate hat_s = DFT_2d of the reversed SRS (src/kzg.rs:166-179) for the domain size d = 2^log2d, so that the
 * first keaki_hip_open_fk[_poly] with this d does not pay for it (it depends on the SRS only; the calls cache it on first use anyway). */
keaki_status keaki_hip_srs_g1_precompute_fk(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, uint32_t log2d, const uint64_t* omega_2d);
/* ---- FK23 sharded over `world` = 2^k ranks (one rank = one ctx = one GPU): kzg::open_fk (src/kzg.rs:157-203) of BASELINE config 5 on N GPUs.
 * The group transforms are split so that every rank does 1/world of the butterflies and of the 2d scalar-mults. The library runs NO
 * collective: a step leaves this rank's outgoing data in d_send, the CALLER exchanges (RCCL over xGMI: all-to-all with equal splits,
 * received chunks in rank order; one all-gather at the end) and hands d_recv to the next step. Steps are asynchronous on the ctx stream
 * like every *_dev entry: order the collective behind them (same stream, or keaki_hip_synchronize) and the next step behind it.
 * Needs d >= world^2 and the whole SRS (srs[0..d)) on every rank. d_send / d_recv: device buffers of sizes4[0] bytes each.
 *   setup (once per handle; the SRS-only transform, this rank's part):
 *     step 0 -> d_send | all-to-all, sizes4[1] bytes per peer | step 1 <- d_recv
 *   open (per polynomial; coeffs: d Fr on the host, the same on every rank):
 *     step 0 (coeffs) -> d_send | all-to-all, sizes4[2] per peer | step 1: d_recv -> d_send | all-to-all, sizes4[2] per peer |
 *     step 2: d_recv -> d_send = d/world affine proofs | all-gather, sizes4[3] per rank | step 3: d_recv -> proofs_out_aff (host, d affine
 *     points in natural order, the bytes keaki_hip_open_fk_poly returns).
 * A step out of order is refused (KEAKI_ERR_BAD_ARG); open step 0 may always start over. `srs` must outlive the handle. */
typedef struct keaki_hip_fk_shard keaki_hip_fk_shard;
keaki_status keaki_hip_fk_shard_create(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, uint32_t log2d, uint32_t rank, uint32_t world,
                                       const uint64_t* omega_2d, const uint64_t* omega_2d_inv, const uint64_t* inv_2d, keaki_hip_fk_shard** out);
void keaki_hip_fk_shard_free(keaki_hip_ctx* ctx, keaki_hip_fk_shard* fk);
/* sizes4[0..4): [0] = bytes of d_send / d_recv; [1] = bytes per peer of the setup's all-to-all, [2] of each of a call's two; [3] = bytes per rank of the all-gather */
keaki_status keaki_hip_fk_shard_sizes(const keaki_hip_fk_shard* fk, size_t* sizes4);
keaki_status keaki_hip_fk_shard_setup(keaki_hip_ctx* ctx, keaki_hip_fk_shard* fk, int32_t step, void* d_send, void* d_recv);
keaki_status keaki_hip_fk_shard_open(keaki_hip_ctx* ctx, keaki_hip_fk_shard* fk, int32_t step, const uint64_t* coeffs, void* d_send, void* d_recv,
                                     uint64_t* proofs_out_aff);
/* Scalar-field DFT on the device: replaces ark-poly `domain.fft` / `domain.ifft` over Fr (src/vec.rs:36-37, src/kzg.rs:185).
 * data: n = 2^log2n Fr, transformed in place: out[j] = sum_i in[i] omega^(ij), then multiplied by *scale_or_null if given
 * (pass omega^-1 and n^-1 for the inverse transform). */
keaki_status keaki_hip_fr_fft(keaki_hip_ctx* ctx, uint64_t* data, uint32_t log2n, const uint64_t* omega, const uint64_t* scale_or_null);

/* ---- vec_commit in one call: the body of vec::vec_commit (reference src/vec.rs:36-46) behind its padding draw -----------------------------
 * values: n Fr (the vector), pad: the one random scalar the caller drew (src/vec.rs:31-33; NULL: none), together the first n + 1 evaluations
 * over the domain of size d = 2^log2d (the rest are zero, as ark-poly's ifft pads). On the device, without the coefficients ever returning to
 * the host: domain.ifft (omega_d_inv = the domain's group_gen_inv, inv_d = its size_inv), the FK23 openings (omega_2d ... as for
 * keaki_hip_open_fk_poly) and the commitment (MSM over all d coefficients: trailing zeros, which DensePolynomial trims, contribute nothing).
 * com_out_jac: u64[12] normalised Jacobian; proofs_out_aff: d affine points. Needs d <= len(srs). */
keaki_status keaki_hip_vec_commit(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, const uint64_t* values, size_t n, const uint64_t* pad, uint32_t log2d,
                                  const uint64_t* omega_d_inv, const uint64_t* inv_d, const uint64_t* omega_2d, const uint64_t* omega_2d_inv,
                                  const uint64_t* inv_2d, uint64_t* com_out_jac, uint64_t* proofs_out_aff);

/* ---- KZG `open` in one call (scope row f-4) -------------------------------------------------------------------------
 * Replaces the body of `open` (reference src/kzg.rs:104-124): value = p(point) and proof = commit(q), q = (p - p(point)) / (x - point),
 * with the synthetic division done on the device (blockwise Horner recurrence) right in front of the MSM, so the n - 1 quotient
 * coefficients never exist on the host. coeffs: n Fr, low degree first, trailing zeros already trimmed as ark-poly's
 * DensePolynomial does. n - 1 must not exceed the SRS length (KEAKI_ERR_TOO_LARGE; the caller raises PolynomialTooLarge first,
 * src/kzg.rs:113-118). proof_out_jac: u64[12] normalised Jacobian; value_out: u64[4] or NULL. */
keaki_status keaki_hip_kzg_open(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, const uint64_t* coeffs, size_t n, const uint64_t* point,
                                uint64_t* proof_out_jac, uint64_t* value_out);

/* The quotient alone, for callers that run the MSM elsewhere (keaki_hip_group_kzg_open does): quotient_out = n - 1 Fr (may be NULL when
 * n <= 1), value_out = p(point) (u64[4] or NULL). Same recurrence as keaki_hip_kzg_open, host pointers in and out. */
keaki_status keaki_hip_kzg_quotient(keaki_hip_ctx* ctx, const uint64_t* coeffs, size_t n, const uint64_t* point, uint64_t* quotient_out,
                                    uint64_t* value_out);

/* ---- KZG `verify` in one call -------------------------------------------------------------------------------------
 * Replaces the body of `verify` (reference src/kzg.rs:127-146): e(com - value g1, g2) == e(proof, [tau]_2 - point g2).
 * Evaluated exactly in that form: the two inner points are fixed-base sums over 8-bit window tables of g1 and g2 (built at the first
 * call of a context), the two pairings one launch of the low-latency pairing kernel. (Until round 4: e(com - value g1 + point proof, g2)
 * == e(proof, [tau]_2) -- the same predicate by bilinearity, tabulated lines, but a variable-base ladder in front.)
 * com_aff, proof_aff: u64[8] affine G1 ((0,0) = identity);
 * tau_g2_aff: u64[16]; point, value: u64[4] Fr. *ok_out = 1 when the equation holds, else 0. */
keaki_status keaki_hip_kzg_verify(keaki_hip_ctx* ctx, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* point,
                                  const uint64_t* value, const uint64_t* proof_aff, int32_t* ok_out);

/* ---- SRS ingest (scope row f-3) -----------------------------------------------------------------------------------------
 * The reference reads .ptau points with `deserialize_uncompressed_unchecked` (src/kzg/ptau.rs:266,314): no curve check at all.
 * A snarkjs .ptau stores coordinates as Montgomery limbs, which is this ABI's point layout, so sections 2 and 3 are uploaded
 * verbatim (keaki_hip_srs_g1_upload) and checked on the device: the number of points with y^2 != x^3 + b (b = 3 on G1,
 * 3/(9+u) on the twist; (0,0) = identity passes) and the index of the first one (UINT64_MAX if none; may be NULL). */
keaki_status keaki_hip_srs_g1_check(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, uint64_t* n_off_curve, uint64_t* first_off_curve);
keaki_status keaki_hip_g2_check(keaki_hip_ctx* ctx, const uint64_t* points_aff, size_t n, uint64_t* n_off_curve, uint64_t* first_off_curve);

/* ---- batched scalar multiplication: replaces `.mul(scalar)` (src/kem.rs:22,30,36,37; src/kzg.rs:57,60,135,144)
 * out[i] = scalars[i] * points[i]   (point_stride = 1) or scalars[i] * points[0] (point_stride = 0).
 * points affine in, affine out. */
keaki_status keaki_hip_g1_mul_batch(keaki_hip_ctx* ctx, const uint64_t* points_aff, int32_t point_stride, const uint64_t* scalars, size_t n, uint64_t* out_aff);
keaki_status keaki_hip_g2_mul_batch(keaki_hip_ctx* ctx, const uint64_t* points_aff, int32_t point_stride, const uint64_t* scalars, size_t n, uint64_t* out_aff);
keaki_status keaki_hip_g1_mul_batch_dev(keaki_hip_ctx* ctx, const void* d_points_aff, int32_t point_stride, const void* d_scalars, size_t n, void* d_out_aff);
keaki_status keaki_hip_g2_mul_batch_dev(keaki_hip_ctx* ctx, const void* d_points_aff, int32_t point_stride, const void* d_scalars, size_t n, void* d_out_aff);

/* ---- batched pairing: replaces E::pairing(p, q) (src/kem.rs:30,58; src/kzg.rs:148) + serialize_uncompressed (src/kem.rs:32,61)
 * gt_out[i] = serialize_uncompressed(e(g1[i], g2[i * g2_stride])); identity in either slot -> GT one. */
keaki_status keaki_hip_pairing_batch(keaki_hip_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, int32_t g2_stride, size_t n, uint8_t* gt_out);
keaki_status keaki_hip_pairing_batch_dev(keaki_hip_ctx* ctx, const void* d_g1_aff, const void* d_g2_aff, int32_t g2_stride, size_t n, void* d_gt_out);

/* ---- KEM composites: the bodies of the loops at src/vec.rs:63-66 and :75-78 --------------------
 * encap_batch: for i < n (src/kem.rs:13-50 with the SAME com / tau_g2 for all i):
 *     ct[i]  = r[i] * (tau_g2 - points[i] * g2)                       (affine G2, u64[16])
 *     gt[i]  = serialize(e(r[i] * (com - values[i] * g1), g2))        (384 bytes)
 *     key[i] = BLAKE3-XOF(gt[i])[0..msg_len]   if key_out != NULL     (src/kem.rs:42-46)
 * r[i] are drawn by the caller (one Fr::rand per item in index order, src/kem.rs:26) so the
 * randomness stream is the caller's. gt_out may be NULL when only keys are wanted. msg_len <= 1<<16.
 * The host-array forms (no _dev suffix) of encap / decap / encrypt / decrypt_batch run batches of two chunks or more (2^16 items; the
 * pairing side: 2^17) as a pipeline -- uploads and downloads on a stream of the context's own under the neighbouring chunks' kernels --
 * and first-touch the caller's output pages from up to three short-lived helper threads ahead of the downloads (a download into pages
 * that do not exist yet runs at a tenth of the link rate). Results are those of one call per chunk; the call returns when every byte
 * has been delivered and every helper has been joined. An output array may be one of the input arrays (messages in, bodies out): a chunk's
 * output pages are touched only after its inputs have been read; arrays the HIP runtime knows (hipHostMalloc / hipHostRegister) are not touched. */
keaki_status keaki_hip_encap_batch(keaki_hip_ctx* ctx, const uint64_t* com_aff, const uint64_t* tau_g2_aff,
                                   const uint64_t* points, const uint64_t* values, const uint64_t* r, size_t n,
                                   uint64_t* ct_out_aff, uint8_t* gt_out, uint8_t* key_out, size_t msg_len);
keaki_status keaki_hip_encap_batch_dev(keaki_hip_ctx* ctx, const void* d_com_aff, const void* d_tau_g2_aff,
                                       const void* d_points, const void* d_values, const void* d_r, size_t n,
                                       void* d_ct_out_aff, void* d_gt_out, void* d_key_out, size_t msg_len);
/* Setup-time (optional): everything of encap_batch that depends on the SETUP only -- the fixed-base window tables of g1, g2 and [tau]_2, the
 * line sequence of g2 and, for batch_hint >= 65,536, the GT table of e(g1, g2) -- built now instead of inside the first large encap_batch of
 * the context (~60 ms). The KEM analogue of keaki_hip_srs_g1_precompute; results never depend on it. */
keaki_status keaki_hip_encap_prepare(keaki_hip_ctx* ctx, const uint64_t* tau_g2_aff, size_t batch_hint);
/* decap_batch (src/kem.rs:55-72): gt[i] = serialize(e(proofs[i], cts[i])), key[i] = BLAKE3-XOF(gt[i]) */
keaki_status keaki_hip_decap_batch(keaki_hip_ctx* ctx, const uint64_t* proofs_aff, const uint64_t* cts_aff, size_t n,
                                   uint8_t* gt_out, uint8_t* key_out, size_t msg_len);
keaki_status keaki_hip_decap_batch_dev(keaki_hip_ctx* ctx, const void* d_proofs_aff, const void* d_cts_aff, size_t n,
                                       void* d_gt_out, void* d_key_out, size_t msg_len);

/* ---- enc::encrypt / enc::decrypt over a batch: KEM + the XOR DEM on the device ---------------------------------------------
 * Replaces the body of the loops of vec_encrypt / vec_decrypt (src/vec.rs:63-66, :75-78), i.e. src/enc.rs:19-40 and :44-55 per item:
 * encapsulate / decapsulate as above, then `key XOR message` (src/enc.rs:32-36, :48-52) inside the KDF kernel -- neither the 384 GT bytes nor
 * the key leave the device. n messages of msg_len (1..65536) bytes each, contiguous. r[i] drawn by the caller as for encap_batch.
 * _dev: d_body_inout holds the messages on entry and the ciphertext bodies on exit (decrypt: bodies in, messages out). */
keaki_status keaki_hip_encrypt_batch(keaki_hip_ctx* ctx, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* points,
                                     const uint64_t* values, const uint64_t* r, const uint8_t* msgs, size_t n, uint64_t* ct_out_aff,
                                     uint8_t* body_out, size_t msg_len);
keaki_status keaki_hip_encrypt_batch_dev(keaki_hip_ctx* ctx, const void* d_com_aff, const void* d_tau_g2_aff, const void* d_points,
                                         const void* d_values, const void* d_r, size_t n, void* d_ct_out_aff, void* d_body_inout, size_t msg_len);
keaki_status keaki_hip_decrypt_batch(keaki_hip_ctx* ctx, const uint64_t* proofs_aff, const uint64_t* cts_aff, const uint8_t* bodies, size_t n,
                                     uint8_t* msgs_out, size_t msg_len);
keaki_status keaki_hip_decrypt_batch_dev(keaki_hip_ctx* ctx, const void* d_proofs_aff, const void* d_cts_aff, size_t n, void* d_body_inout,
                                         size_t msg_len);

/* ---- device group: in-process multi-GPU (SURVEY 8b `_create(n_devices)`, 8e "single process, one host thread per GPU") -----------------
 * What a Rust caller of keaki gets with more than one GPU and no PyTorch / RCCL: the library keeps one ctx and one host worker thread per
 * entry of `devices` (an ordinal may repeat: several contexts on one GPU, which is how the single-GPU test box exercises it), the SRS is
 * split into contiguous chunks, chunk i resident on devices[i] with ITS window tables (built in parallel at upload), and
 *   keaki_hip_group_msm_g1      = commit's MSM (src/kzg.rs:98): every member runs a complete Pippenger on its (scalar, point) range, the
 *                                 N partial sums (96 B each) come back through host memory -- no collective in-process -- and member 0
 *                                 adds them (keaki_hip_g1_sum). EC addition is exact: the affine result equals the one-GPU result.
 *   keaki_hip_group_kzg_open    = `open` (src/kzg.rs:104-124): quotient on member 0, then the group MSM.
 *   keaki_hip_group_encap_batch / _decap_batch = the loops of vec_encrypt / vec_decrypt (src/vec.rs:63-66, :75-78) split by item range,
 *                                 every member writing straight into its slice of the caller's output arrays.
 * Ranges follow the SRS (member i owns points [i*len/N ...)), not the polynomial, so the tables serve every polynomial.
 * Errors: the first failing member's status; keaki_hip_group_last_error names the member. Calls on one group are serialised. */
typedef struct keaki_hip_group keaki_hip_group;
typedef struct keaki_hip_group_srs_g1 keaki_hip_group_srs_g1;
keaki_status keaki_hip_group_create(const int32_t* devices, size_t n_devices, keaki_hip_group** out);
void keaki_hip_group_destroy(keaki_hip_group* g);
size_t keaki_hip_group_size(const keaki_hip_group* g);
keaki_hip_ctx* keaki_hip_group_ctx(const keaki_hip_group* g, size_t member);   /* member's ctx, e.g. for verify / open_fk on member 0 */
const char* keaki_hip_group_last_error(const keaki_hip_group* g);               /* g may be NULL: last create error */
/* keaki_hip_group_create enables peer access between every pair of distinct devices (the exchanges of the sharded FK23 are then device-to-device
 * copies on the members' streams, ordered by events). "" when every pair has it; otherwise the pairs whose copies the runtime stages. */
const char* keaki_hip_group_peer_note(const keaki_hip_group* g);
/* precompute != 0: also build every chunk's window tables (KEAKI_ERR_OOM of a table is tolerated: that member runs the generic MSM) */
keaki_status keaki_hip_group_srs_g1_upload(keaki_hip_group* g, const uint64_t* points_aff, size_t n, int32_t precompute, keaki_hip_group_srs_g1** out);
size_t keaki_hip_group_srs_g1_len(const keaki_hip_group_srs_g1* srs);
int32_t keaki_hip_group_srs_g1_has_tables(const keaki_hip_group_srs_g1* srs);  /* 1: every member that holds points holds their window tables */
void keaki_hip_group_srs_g1_free(keaki_hip_group* g, keaki_hip_group_srs_g1* srs);
keaki_status keaki_hip_group_msm_g1(keaki_hip_group* g, const keaki_hip_group_srs_g1* srs, const uint64_t* scalars, size_t n, uint64_t* out_jac);
keaki_status keaki_hip_group_kzg_open(keaki_hip_group* g, const keaki_hip_group_srs_g1* srs, const uint64_t* coeffs, size_t n, const uint64_t* point,
                                      uint64_t* proof_out_jac, uint64_t* value_out);
keaki_status keaki_hip_group_encap_batch(keaki_hip_group* g, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* points,
                                         const uint64_t* values, const uint64_t* r, size_t n, uint64_t* ct_out_aff, uint8_t* gt_out,
                                         uint8_t* key_out, size_t msg_len);
keaki_status keaki_hip_group_decap_batch(keaki_hip_group* g, const uint64_t* proofs_aff, const uint64_t* cts_aff, size_t n, uint8_t* gt_out,
                                         uint8_t* key_out, size_t msg_len);
/* the same split for keaki_hip_encrypt_batch / keaki_hip_decrypt_batch (KEM + XOR DEM on the members) */
keaki_status keaki_hip_group_encrypt_batch(keaki_hip_group* g, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* points,
                                           const uint64_t* values, const uint64_t* r, const uint8_t* msgs, size_t n, uint64_t* ct_out_aff,
                                           uint8_t* body_out, size_t msg_len);
keaki_status keaki_hip_group_decrypt_batch(keaki_hip_group* g, const uint64_t* proofs_aff, const uint64_t* cts_aff, const uint8_t* bodies, size_t n,
                                           uint8_t* msgs_out, size_t msg_len);

/* kzg::open_fk (src/kzg.rs:157-203) over the members of a group: the sharded FK23 pipeline (keaki_hip_fk_shard_*, member i = rank i) with the
 * exchanges done inside the library -- device-to-device copies between the members' buffers (hipMemcpyPeer), no collective, no RCCL. The
 * handle keeps srs[0..d) on every member (points_aff: the first d = 2^log2d points of the SRS) and every member's part of the SRS-only
 * transform. Needs a power-of-two group and d >= N^2; any other shape runs un-sharded on member 0. omega_2d, omega_2d_inv, inv_2d as for
 * keaki_hip_open_fk_poly. _open: coeffs = d Fr on the host, proofs_out_aff = d affine points, the bytes keaki_hip_open_fk_poly returns. */
typedef struct keaki_hip_group_fk keaki_hip_group_fk;
keaki_status keaki_hip_group_fk_create(keaki_hip_group* g, const uint64_t* points_aff, uint32_t log2d, const uint64_t* omega_2d,
                                       const uint64_t* omega_2d_inv, const uint64_t* inv_2d, keaki_hip_group_fk** out);
keaki_status keaki_hip_group_fk_open(keaki_hip_group* g, keaki_hip_group_fk* fk, const uint64_t* coeffs, uint64_t* proofs_out_aff);
void keaki_hip_group_fk_free(keaki_hip_group* g, keaki_hip_group_fk* fk);

/* ---- instrumentation (bench.py reads these; not part of the reference surface) ---------------- */
/* device time in milliseconds of the dominant kernel (bucket accumulation) of the last msm_*_dev
 * call, measured with HIP events on the ctx stream; < 0 if timing is disabled. */
keaki_status keaki_hip_set_timing(keaki_hip_ctx* ctx, int32_t enabled);
float keaki_hip_last_msm_bucket_ms(const keaki_hip_ctx* ctx);
float keaki_hip_last_msm_total_ms(const keaki_hip_ctx* ctx);
/* device time (ms) of the last keaki_hip_open_fk[_poly] call made while timing was enabled: out3 = [the 2d pointwise scalar-mults, the butterfly
 * stages of the two size-d group transforms (the dominant kernel k_g1_fft_stage_map), the whole device pipeline]; < 0 when there was none */
keaki_status keaki_hip_last_fk_ms(keaki_hip_ctx* ctx, float* out3);
/* window size (bits) the last MSM used */
int32_t keaki_hip_last_msm_window_bits(const keaki_hip_ctx* ctx);

/* Test hook: final exponentiation alone. f_mont: n x 12 Fq (Montgomery limbs, order c0.c0.c0 ... c1.c2.c1), the output of a
 * Miller loop; gt_out: n x 384 bytes, as keaki_hip_pairing_batch would emit. Lets the tests bisect Miller loop vs final exponent. */
keaki_status keaki_hip_g2_prepare(keaki_hip_ctx* ctx, const uint64_t* g2_aff, uint64_t* lines_out, size_t lines_out_bytes);
keaki_status keaki_hip_miller_loop_batch(keaki_hip_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t n, uint64_t* f_mont_out);
keaki_status keaki_hip_final_exp_batch(keaki_hip_ctx* ctx, const uint64_t* f_mont, size_t n, uint8_t* gt_out);

/* On-device self-test: runs the hand-scheduled Fq instruction streams (product, add, sub, neg, square and a
 * dependent chain) against the portable template code on `blocks` x 256 lanes x `iters` random and corner-case
 * inputs; *mismatches_out must come back 0. */
keaki_status keaki_hip_selftest_field(keaki_hip_ctx* ctx, uint32_t blocks, uint32_t iters, uint32_t seed, uint64_t* mismatches_out);

#ifdef __cplusplus
}
#endif
#endif /* KEAKI_HIP_H */
