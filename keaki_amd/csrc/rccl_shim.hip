// libkeaki_hip_rccl.so (include/keaki_hip_rccl.h): the three collectives of the one-process-per-GPU form over RCCL, on the context's stream.
// Host code only. Links librccl; libkeaki_hip.so does not.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/keaki_hip_rccl.h"

struct keaki_hip_rccl {
  keaki_hip_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  void* gathered = nullptr;      // world x 96 bytes: the partials of the sharded MSM
  void* partial = nullptr;       // 96 bytes
  void* d_status = nullptr;      // world + 1 int32: [this rank's status of the collective in flight | every rank's, gathered]
  int32_t* h_status = nullptr;   // pinned: [0, world) the gathered status words of the last collective MSM (keaki_hip_rccl_collective_status)
  std::mutex mu;
  std::string err;
};

namespace {
thread_local std::string g_create_error;
// the context's GPU as the calling thread's current device for the scope (RCCL binds a communicator and its launches to the current device);
// the caller's own choice comes back on exit, as in the main library
struct DeviceScope {
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(const keaki_hip_ctx* ctx) {
    const int device = keaki_hip_ctx_device(ctx);
    int cur = -1;
    if (device < 0 || (hipGetDevice(&cur) == hipSuccess && cur == device)) return;
    ok = hipSetDevice(device) == hipSuccess;
    if (ok) prev = cur;
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
keaki_status rfail(keaki_hip_rccl* rc, keaki_status code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (rc) rc->err = buf; else g_create_error = buf;
  return code;
}
#define NCCL_TRY(rc, call)                                                                                        \
  do {                                                                                                            \
    ncclResult_t r_ = (call);                                                                                     \
    if (r_ != ncclSuccess) return rfail(rc, KEAKI_ERR_RCCL, "%s failed: %s", #call, ncclGetErrorString(r_));       \
  } while (0)
}  // namespace

extern "C" {

keaki_status keaki_hip_rccl_unique_id(uint8_t out128[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!out128) return rfail(nullptr, KEAKI_ERR_BAD_ARG, "rccl_unique_id: null pointer");
  ncclUniqueId id;
  NCCL_TRY(nullptr, ncclGetUniqueId(&id));
  memcpy(out128, &id, 128);
  return KEAKI_OK;
}

keaki_status keaki_hip_rccl_create(keaki_hip_ctx* ctx, const uint8_t id128[128], int32_t rank, int32_t world, keaki_hip_rccl** out) {
  if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return rfail(nullptr, KEAKI_ERR_BAD_ARG, "rccl_create: bad argument");
  *out = nullptr;
  DeviceScope dev_(ctx);           // the communicator and the two buffers below belong to the context's GPU
  if (!dev_.ok) return rfail(nullptr, KEAKI_ERR_HIP, "rccl_create: hipSetDevice(%d) failed", (int)keaki_hip_ctx_device(ctx));
  keaki_status st = keaki_hip_synchronize(ctx);
  if (st != KEAKI_OK) return rfail(nullptr, st, "rccl_create: %s", keaki_hip_last_error(ctx));
  auto* rc = new keaki_hip_rccl();
  rc->ctx = ctx; rc->rank = rank; rc->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = ncclCommInitRank(&rc->comm, world, id, rank);
  if (r != ncclSuccess) { rfail(nullptr, KEAKI_ERR_RCCL, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r)); delete rc; return KEAKI_ERR_RCCL; }
  if (hipMalloc(&rc->gathered, (size_t)world * 96) != hipSuccess || hipMalloc(&rc->partial, 96) != hipSuccess ||
      hipMalloc(&rc->d_status, (size_t)(world + 1) * 4) != hipSuccess || hipHostMalloc((void**)&rc->h_status, (size_t)world * 4) != hipSuccess) {
    rfail(nullptr, KEAKI_ERR_OOM, "rccl_create: hipMalloc failed");
    keaki_hip_rccl_destroy(rc);
    return KEAKI_ERR_OOM;
  }
  memset(rc->h_status, 0, (size_t)world * 4);      // collective_status before the first MSM: every rank fine
  *out = rc;
  return KEAKI_OK;
}

void keaki_hip_rccl_destroy(keaki_hip_rccl* rc) {
  if (!rc) return;
  DeviceScope dev_(rc->ctx);
  if (rc->ctx) (void)keaki_hip_synchronize(rc->ctx);
  if (rc->comm) (void)ncclCommDestroy(rc->comm);
  if (rc->gathered) (void)hipFree(rc->gathered);
  if (rc->partial) (void)hipFree(rc->partial);
  if (rc->d_status) (void)hipFree(rc->d_status);
  if (rc->h_status) (void)hipHostFree(rc->h_status);
  delete rc;
}

const char* keaki_hip_rccl_last_error(const keaki_hip_rccl* rc) {
  if (!rc) return g_create_error.c_str();
  thread_local std::string copy;
  auto* r = const_cast<keaki_hip_rccl*>(rc);
  std::lock_guard<std::mutex> lk(r->mu);
  copy = r->err;
  return copy.c_str();
}

// (A NULL buffer is a caller error on EVERY rank alike -- the buffers are allocated from the same sizes on all of them --, so returning
// before the enqueue cannot split the ranks; any other failure here comes from RCCL itself, after which the communicator is dead.)
keaki_status keaki_hip_rccl_all_gather(keaki_hip_rccl* rc, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  if (!rc) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(rc->mu);
  if (!d_send || !d_recv) return rfail(rc, KEAKI_ERR_BAD_ARG, "rccl_all_gather: null pointer");
  DeviceScope dev_(rc->ctx);
  NCCL_TRY(rc, ncclAllGather(d_send, d_recv, bytes_per_rank, ncclUint8, rc->comm, (hipStream_t)keaki_hip_ctx_stream(rc->ctx)));
  return KEAKI_OK;
}

keaki_status keaki_hip_rccl_all_to_all(keaki_hip_rccl* rc, const void* d_send, void* d_recv, size_t bytes_per_peer) {
  if (!rc) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(rc->mu);
  if (!d_send || !d_recv) return rfail(rc, KEAKI_ERR_BAD_ARG, "rccl_all_to_all: null pointer");
  DeviceScope dev_(rc->ctx);
  NCCL_TRY(rc, ncclAllToAll(d_send, d_recv, bytes_per_peer, ncclUint8, rc->comm, (hipStream_t)keaki_hip_ctx_stream(rc->ctx)));
  return KEAKI_OK;
}

// A rank-local failure must not leave the other ranks waiting in a collective this rank never enqueued: the all-gather is ALWAYS enqueued --
// a failing rank contributes the identity (96 zero bytes: z = 0) -- and a second, 4-byte all-gather carries every rank's status, which
// keaki_hip_rccl_collective_status reports after the stream has drained. The failing rank itself returns its error at once.
keaki_status keaki_hip_rccl_msm_g1(keaki_hip_rccl* rc, const keaki_hip_srs_g1* srs_chunk, const void* d_scalars, size_t n, void* d_out_jac) {
  if (!rc) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(rc->mu);
  DeviceScope dev_(rc->ctx);
  hipStream_t st_ = (hipStream_t)keaki_hip_ctx_stream(rc->ctx);
  keaki_status local = KEAKI_OK;
  if (!srs_chunk || !d_out_jac) local = rfail(rc, KEAKI_ERR_BAD_ARG, "rccl_msm_g1: null pointer");
  if (local == KEAKI_OK) {
    local = keaki_hip_msm_g1_dev(rc->ctx, srs_chunk, d_scalars, n, rc->partial);
    if (local != KEAKI_OK) rfail(rc, local, "rccl_msm_g1: %s", keaki_hip_last_error(rc->ctx));
  }
  if (local != KEAKI_OK && hipMemsetAsync(rc->partial, 0, 96, st_) != hipSuccess) local = KEAKI_ERR_HIP;   // the identity: the sum ignores it
  // this rank's status word is written by a 32-bit memset ON the stream: no host memory is read later, so nothing here waits for the stream (the call
  // stays asynchronous, as the header says) and any number of calls may be in flight (until round 5 the word was copied up from a ring of eight
  // pinned slots, which a ninth un-synchronised call would have overwritten before its copy ran). A HIP failure is folded into `local` and the
  // collectives are still enqueued: leaving before them would split the ranks.
  if (hipMemsetD32Async((hipDeviceptr_t)rc->d_status, (int)local, 1, st_) != hipSuccess && local == KEAKI_OK)
    local = rfail(rc, KEAKI_ERR_HIP, "rccl_msm_g1: status write failed");
  // EC addition is not an RCCL reduction operator: all-gather the 96-byte partials, add them on every rank.
  // (A failing RCCL call itself cannot be repaired from here: the communicator is dead on every rank alike.)
  NCCL_TRY(rc, ncclAllGather(rc->partial, rc->gathered, 96, ncclUint8, rc->comm, st_));
  NCCL_TRY(rc, ncclAllGather(rc->d_status, (char*)rc->d_status + 4, 4, ncclUint8, rc->comm, st_));
  if (hipMemcpyAsync(rc->h_status, (char*)rc->d_status + 4, (size_t)rc->world * 4, hipMemcpyDeviceToHost, st_) != hipSuccess && local == KEAKI_OK)
    local = rfail(rc, KEAKI_ERR_HIP, "rccl_msm_g1: status download failed");
  if (local != KEAKI_OK) return local;
  keaki_status st = keaki_hip_g1_sum_dev(rc->ctx, rc->gathered, (size_t)rc->world, d_out_jac);
  if (st != KEAKI_OK) return rfail(rc, st, "rccl_msm_g1: %s", keaki_hip_last_error(rc->ctx));
  return KEAKI_OK;
}

// Waits for the context's stream, then: KEAKI_OK when every rank's share of the last keaki_hip_rccl_msm_g1 succeeded, otherwise the first
// failing rank's status (its index in *bad_rank when that is not NULL): the result in d_out_jac then lacks that rank's chunk.
keaki_status keaki_hip_rccl_collective_status(keaki_hip_rccl* rc, int32_t* bad_rank) {
  if (!rc) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(rc->mu);
  DeviceScope dev_(rc->ctx);
  keaki_status st = keaki_hip_synchronize(rc->ctx);
  if (st != KEAKI_OK) return rfail(rc, st, "rccl_collective_status: %s", keaki_hip_last_error(rc->ctx));
  for (int r = 0; r < rc->world; r++)
    if (rc->h_status[r] != 0) {
      if (bad_rank) *bad_rank = r;
      return rfail(rc, (keaki_status)rc->h_status[r], "rank %d failed its share of the last collective MSM (status %d)", r, (int)rc->h_status[r]);
    }
  if (bad_rank) *bad_rank = -1;
  return KEAKI_OK;
}

}  // extern "C"
