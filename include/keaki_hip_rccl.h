/* keaki_hip_rccl.h -- C ABI of libkeaki_hip_rccl.so: the exchanges of the one-process-per-GPU form of keaki's hot path over RCCL (xGMI),
 * for callers that are NOT inside PyTorch (a Rust / C++ application launched with one process per GPU). Optional: libkeaki_hip.so itself
 * never opens a connection and has no RCCL dependency; this library adds the three collectives the sharded path needs, enqueued on the
 * stream of the caller's keaki_hip_ctx, so that they are ordered with the kernels without host synchronisation.
 *
 *   kzg::commit (src/kzg.rs:98) over N ranks   : keaki_hip_rccl_msm_g1  = this rank's MSM + all-gather of the 96-byte partials + EC sum
 *   kzg::open_fk (src/kzg.rs:157-203) over N   : keaki_hip_rccl_all_to_all / _all_gather between the keaki_hip_fk_shard_* steps
 *                                                (the two callbacks of keaki::dist::FkExchange / hip::FkExchange)
 * Rendezvous: rank 0 calls keaki_hip_rccl_unique_id and hands the 128 bytes to the other ranks by any means (file, socket, MPI, env);
 * every rank then calls keaki_hip_rccl_create. In one process per GPU only: RCCL refuses two ranks on one device. */
#ifndef KEAKI_HIP_RCCL_H
#define KEAKI_HIP_RCCL_H
#include "keaki_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#define KEAKI_ERR_RCCL (-6) /* an RCCL call failed (keaki_hip_rccl_last_error has ncclGetErrorString) */
typedef struct keaki_hip_rccl keaki_hip_rccl;
keaki_status keaki_hip_rccl_unique_id(uint8_t out128[128]);
keaki_status keaki_hip_rccl_create(keaki_hip_ctx* ctx, const uint8_t id128[128], int32_t rank, int32_t world, keaki_hip_rccl** out);
void keaki_hip_rccl_destroy(keaki_hip_rccl* rc);
const char* keaki_hip_rccl_last_error(const keaki_hip_rccl* rc); /* rc may be NULL: last create error */
/* d_out_jac (96 bytes of device memory) = sum over all ranks of MSM(srs_chunk, d_scalars[0..n)): asynchronous on the ctx stream.
 * Failure is collective-safe: a rank whose own share fails (bad handle, out of memory) returns its error at once but STILL takes part in the
 * exchange with the identity as its partial, so the other ranks never wait for it; they learn of it from keaki_hip_rccl_collective_status.
 * A caller that consumes d_out_jac MUST call keaki_hip_rccl_collective_status first (it is the only place a peer's failure surfaces: without
 * it a failed peer yields a sum that silently lacks its chunk); before the first MSM it reports KEAKI_OK.
 * After KEAKI_ERR_RCCL from any entry point the communicator is dead: destroy it. */
keaki_status keaki_hip_rccl_msm_g1(keaki_hip_rccl* rc, const keaki_hip_srs_g1* srs_chunk, const void* d_scalars, size_t n, void* d_out_jac);
/* Waits for the stream; KEAKI_OK when every rank's share of the last keaki_hip_rccl_msm_g1 succeeded, else the first failing rank's status
 * (its index in *bad_rank unless NULL; -1 when none): d_out_jac then lacks that rank's chunk. */
keaki_status keaki_hip_rccl_collective_status(keaki_hip_rccl* rc, int32_t* bad_rank);
/* d_send: world chunks of bytes_per_peer, chunk q for rank q; d_recv: the chunks received, in rank order. Asynchronous on the ctx stream. */
keaki_status keaki_hip_rccl_all_to_all(keaki_hip_rccl* rc, const void* d_send, void* d_recv, size_t bytes_per_peer);
/* d_recv = every rank's bytes_per_rank of d_send, in rank order */
keaki_status keaki_hip_rccl_all_gather(keaki_hip_rccl* rc, const void* d_send, void* d_recv, size_t bytes_per_rank);
#ifdef __cplusplus
}
#endif
#endif
