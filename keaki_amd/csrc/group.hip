// keaki_hip_group_*: in-process multi-GPU over the single-context C ABI (include/keaki_hip.h). Host code only: one context and one
// worker thread per member, ranges of the SRS / of the batch per member, partial sums through host memory. No collective, no RCCL:
// inside one process the 96-byte partials are a memcpy away.
#include "internal.h"

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <new>
#include <thread>
#include <vector>

namespace {
thread_local std::string g_group_create_error;

// one persistent worker per member: jobs run on the member's own thread so that N members issue their calls concurrently
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<keaki_status()> job;
  bool has_job = false, done = false, quit = false;
  keaki_status result = KEAKI_OK;
  std::string what;              // set when the job ended in a C++ exception: the context's last_error knows nothing of it
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) return;
      std::function<keaki_status()> j = std::move(job);
      has_job = false;
      lk.unlock();
      keaki_status r;
      std::string w;
      try { r = j(); } catch (const std::bad_alloc&) { r = KEAKI_ERR_OOM; w = "host allocation failed in the member's worker thread"; }
      catch (const std::exception& e) { r = KEAKI_ERR_OOM; w = std::string("exception in the member's worker thread: ") + e.what(); }
      catch (...) { r = KEAKI_ERR_OOM; w = "unknown exception in the member's worker thread"; }       // no exception leaves the thread
      lk.lock();
      result = r; what = std::move(w);
      done = true;
      cv.notify_all();
    }
  }
  void submit(std::function<keaki_status()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j); has_job = true; done = false;
    cv.notify_all();
  }
  keaki_status wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    return result;
  }
};
}  // namespace

struct keaki_hip_group {
  std::vector<keaki_hip_ctx*> ctx;
  std::vector<Worker*> workers;
  std::vector<hipEvent_t> sent;  // member i's outgoing copies of the exchange in flight are done (recorded on its stream)
  std::string peer_note;         // which pairs of devices have no direct access (their copies are staged by the runtime); empty: all direct
  std::mutex mu;                 // calls on one group run one after another
  std::string err;
};
struct keaki_hip_group_srs_g1 {
  size_t n = 0;
  std::vector<keaki_hip_srs_g1*> chunk;   // member i: points [lo(i), hi(i))
  std::vector<size_t> lo, hi;
  std::vector<uint8_t> tabled;            // member i's chunk has its window tables (an out-of-memory table build is tolerated)
};

namespace {
keaki_status gfail(keaki_hip_group* g, keaki_status code, const char* fmt, ...) {
  char buf[640];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (g) g->err = buf; else g_group_create_error = buf;
  return code;
}
// contiguous range of member i out of n units; sizes differ by at most one (the rule of keaki::dist::Shard::bounds)
void range_of(size_t n, size_t world, size_t i, size_t* lo, size_t* hi) {
  const size_t base = n / world, rem = n % world;
  *lo = i * base + std::min(i, rem);
  *hi = *lo + base + (i < rem ? 1 : 0);
}
// runs fn(member) on every member's thread; the first failing member's status, its message copied into the group
keaki_status run_all(keaki_hip_group* g, const char* what, const std::function<keaki_status(size_t)>& fn) {
  const size_t N = g->ctx.size();
  for (size_t i = 0; i < N; i++) g->workers[i]->submit([&fn, i] { return fn(i); });
  keaki_status first = KEAKI_OK;
  for (size_t i = 0; i < N; i++) {
    const keaki_status st = g->workers[i]->wait();
    if (st != KEAKI_OK && first == KEAKI_OK) {
      first = st;
      const std::string w = g->workers[i]->what;     // a C++ exception in the worker: the context's own message is stale then
      gfail(g, st, "%s: member %zu (device %d): %s", what, i, g->ctx[i]->device, w.empty() ? keaki_hip_last_error(g->ctx[i]) : w.c_str());
    }
  }
  return first;
}
}  // namespace

extern "C" {

keaki_status keaki_hip_group_create(const int32_t* devices, size_t n_devices, keaki_hip_group** out) {
  if (!out) return gfail(nullptr, KEAKI_ERR_BAD_ARG, "group_create: out is null");
  *out = nullptr;
  if (!devices || n_devices == 0 || n_devices > 64) return gfail(nullptr, KEAKI_ERR_BAD_ARG, "group_create: 1..64 members wanted");
  keaki_hip_group* g = new keaki_hip_group();
  for (size_t i = 0; i < n_devices; i++) {
    keaki_hip_ctx* c = nullptr;
    const keaki_status st = keaki_hip_ctx_create(devices[i], KEAKI_HIP_STREAM_PRIVATE, &c);
    if (st != KEAKI_OK) {
      gfail(nullptr, st, "group_create: member %zu (device %d): %s", i, devices[i], keaki_hip_last_error(nullptr));
      for (keaki_hip_ctx* x : g->ctx) keaki_hip_ctx_destroy(x);
      delete g;
      return st;
    }
    g->ctx.push_back(c);
  }
  for (size_t i = 0; i < n_devices; i++) {
    Worker* w = new Worker();
    const int device = g->ctx[i]->device;
    w->th = std::thread([w, device] { (void)hipSetDevice(device); w->loop(); });     // the member's GPU is its thread's current device throughout
    g->workers.push_back(w);
  }
  // Direct access between every pair of distinct devices (xGMI on one node), so that the exchanges of the sharded FK23 are device-to-device
  // DMA on the members' own streams. A pair that cannot have it keeps working -- hipMemcpyPeerAsync then goes through host memory -- and is named
  // in keaki_hip_group_peer_note.
  g->sent.assign(n_devices, nullptr);
  for (size_t i = 0; i < n_devices; i++) {
    keaki_internal::DeviceScope dev_(g->ctx[i]->device);
    (void)hipEventCreateWithFlags(&g->sent[i], hipEventDisableTiming);
    for (size_t j = 0; j < n_devices; j++) {
      const int a = g->ctx[i]->device, b = g->ctx[j]->device;
      if (a == b) continue;
      bool seen = false;
      for (size_t k = 0; k < j; k++) seen |= g->ctx[k]->device == b;
      if (seen) continue;
      int can = 0;
      hipError_t e = hipDeviceCanAccessPeer(&can, a, b);
      if (e == hipSuccess && can) {
        e = hipDeviceEnablePeerAccess(b, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
      }
      if (e != hipSuccess || !can) {
        char buf[96];
        snprintf(buf, sizeof buf, "%sdevice %d -> %d: %s", g->peer_note.empty() ? "" : "; ", a, b, can ? hipGetErrorString(e) : "no peer access");
        g->peer_note += buf;
        (void)hipGetLastError();
      }
    }
  }
  *out = g;
  return KEAKI_OK;
}
const char* keaki_hip_group_peer_note(const keaki_hip_group* g) { return g ? g->peer_note.c_str() : ""; }

void keaki_hip_group_destroy(keaki_hip_group* g) {
  if (!g) return;
  for (Worker* w : g->workers) {
    { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_all(); }
    w->th.join();
    delete w;
  }
  for (size_t i = 0; i < g->sent.size(); i++)
    if (g->sent[i]) { keaki_internal::DeviceScope dev_(g->ctx[i]->device); (void)hipEventDestroy(g->sent[i]); }
  for (keaki_hip_ctx* c : g->ctx) keaki_hip_ctx_destroy(c);
  delete g;
}

size_t keaki_hip_group_size(const keaki_hip_group* g) { return g ? g->ctx.size() : 0; }
keaki_hip_ctx* keaki_hip_group_ctx(const keaki_hip_group* g, size_t member) { return g && member < g->ctx.size() ? g->ctx[member] : nullptr; }
const char* keaki_hip_group_last_error(const keaki_hip_group* g) {
  if (!g) return g_group_create_error.c_str();
  thread_local std::string copy;
  keaki_hip_group* gg = const_cast<keaki_hip_group*>(g);
  std::lock_guard<std::mutex> lk(gg->mu);
  copy = gg->err;
  return copy.c_str();
}

keaki_status keaki_hip_group_srs_g1_upload(keaki_hip_group* g, const uint64_t* points_aff, size_t n, int32_t precompute, keaki_hip_group_srs_g1** out) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (!out || (n && !points_aff)) return gfail(g, KEAKI_ERR_BAD_ARG, "group_srs_g1_upload: null pointer");
  *out = nullptr;
  const size_t N = g->ctx.size();
  auto* s = new keaki_hip_group_srs_g1();
  s->n = n;
  s->chunk.assign(N, nullptr); s->lo.resize(N); s->hi.resize(N); s->tabled.assign(N, 0);
  for (size_t i = 0; i < N; i++) range_of(n, N, i, &s->lo[i], &s->hi[i]);
  const keaki_status st = run_all(g, "group_srs_g1_upload", [&](size_t i) -> keaki_status {
    const size_t m = s->hi[i] - s->lo[i];
    keaki_status r = keaki_hip_srs_g1_upload(g->ctx[i], m ? points_aff + 8 * s->lo[i] : nullptr, m, &s->chunk[i]);
    if (r != KEAKI_OK || !precompute || !m) return r;
    size_t bytes = 0;
    r = keaki_hip_srs_g1_precompute(g->ctx[i], s->chunk[i], &bytes);
    s->tabled[i] = r == KEAKI_OK && bytes > 0;
    return r == KEAKI_ERR_OOM ? KEAKI_OK : r;          // tables are optional: that member keeps the generic per-window MSM
  });
  if (st != KEAKI_OK) {
    for (size_t i = 0; i < N; i++) keaki_hip_srs_g1_free(g->ctx[i], s->chunk[i]);
    delete s;
    return st;
  }
  *out = s;
  return KEAKI_OK;
}
size_t keaki_hip_group_srs_g1_len(const keaki_hip_group_srs_g1* srs) { return srs ? srs->n : 0; }
int32_t keaki_hip_group_srs_g1_has_tables(const keaki_hip_group_srs_g1* srs) {
  if (!srs) return 0;
  for (size_t i = 0; i < srs->chunk.size(); i++)
    if (srs->hi[i] > srs->lo[i] && !srs->tabled[i]) return 0;
  return 1;
}
void keaki_hip_group_srs_g1_free(keaki_hip_group* g, keaki_hip_group_srs_g1* srs) {
  if (!srs) return;
  for (size_t i = 0; i < srs->chunk.size(); i++) keaki_hip_srs_g1_free(g && i < g->ctx.size() ? g->ctx[i] : nullptr, srs->chunk[i]);
  delete srs;
}

static keaki_status group_msm_locked(keaki_hip_group* g, const keaki_hip_group_srs_g1* srs, const uint64_t* scalars, size_t n, uint64_t* out_jac) {
  const size_t N = g->ctx.size();
  if (srs->chunk.size() != N) return gfail(g, KEAKI_ERR_BAD_ARG, "group_msm_g1: the SRS was uploaded through a group of another size");
  if (n > srs->n) return gfail(g, KEAKI_ERR_TOO_LARGE, "msm: %zu scalars but the SRS holds %zu points", n, srs->n);
  std::vector<uint64_t> partials(12 * N);
  ST_TRY(run_all(g, "group_msm_g1", [&](size_t i) -> keaki_status {
    const size_t lo = std::min(srs->lo[i], n), hi = std::min(srs->hi[i], n);       // zip-truncation of msm_unchecked, per range
    return keaki_hip_msm_g1(g->ctx[i], srs->chunk[i], hi > lo ? scalars + 4 * lo : nullptr, hi - lo, &partials[12 * i]);
  }));
  const keaki_status st = keaki_hip_g1_sum(g->ctx[0], partials.data(), N, out_jac);
  if (st != KEAKI_OK) return gfail(g, st, "group_msm_g1: sum of the partials: %s", keaki_hip_last_error(g->ctx[0]));
  return KEAKI_OK;
}

keaki_status keaki_hip_group_msm_g1(keaki_hip_group* g, const keaki_hip_group_srs_g1* srs, const uint64_t* scalars, size_t n, uint64_t* out_jac) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (!srs || !out_jac || (n && !scalars)) return gfail(g, KEAKI_ERR_BAD_ARG, "group_msm_g1: null pointer");
  return group_msm_locked(g, srs, scalars, n, out_jac);
}

keaki_status keaki_hip_group_kzg_open(keaki_hip_group* g, const keaki_hip_group_srs_g1* srs, const uint64_t* coeffs, size_t n, const uint64_t* point,
                                      uint64_t* proof_out_jac, uint64_t* value_out) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (!srs || !point || !proof_out_jac || (n && !coeffs)) return gfail(g, KEAKI_ERR_BAD_ARG, "group_kzg_open: null pointer");
  const size_t nq = n ? n - 1 : 0;
  if (nq > srs->n) return gfail(g, KEAKI_ERR_TOO_LARGE, "msm: %zu scalars but the SRS holds %zu points", nq, srs->n);
  std::unique_ptr<uint64_t[]> q(new (std::nothrow) uint64_t[4 * std::max<size_t>(nq, 1)]);      // 32 bytes per coefficient: 0.5 GB at 2^24
  if (!q) return gfail(g, KEAKI_ERR_OOM, "group_kzg_open: no host memory for the %zu coefficients of the quotient", nq);
  const keaki_status st = keaki_hip_kzg_quotient(g->ctx[0], coeffs, n, point, q.get(), value_out);
  if (st != KEAKI_OK) return gfail(g, st, "group_kzg_open: quotient: %s", keaki_hip_last_error(g->ctx[0]));
  return group_msm_locked(g, srs, q.get(), nq, proof_out_jac);
}

keaki_status keaki_hip_group_encap_batch(keaki_hip_group* g, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* points,
                                         const uint64_t* values, const uint64_t* r, size_t n, uint64_t* ct_out_aff, uint8_t* gt_out,
                                         uint8_t* key_out, size_t msg_len) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (n == 0) return KEAKI_OK;
  if (!com_aff || !tau_g2_aff || !points || !values || !r || !ct_out_aff || (!gt_out && !key_out) || msg_len > 65536)
    return gfail(g, KEAKI_ERR_BAD_ARG, "group_encap_batch: bad argument");
  const size_t N = g->ctx.size();
  return run_all(g, "group_encap_batch", [&](size_t i) -> keaki_status {
    size_t lo, hi;
    range_of(n, N, i, &lo, &hi);
    if (hi == lo) return KEAKI_OK;
    return keaki_hip_encap_batch(g->ctx[i], com_aff, tau_g2_aff, points + 4 * lo, values + 4 * lo, r + 4 * lo, hi - lo, ct_out_aff + 16 * lo,
                                 gt_out ? gt_out + 384 * lo : nullptr, key_out ? key_out + msg_len * lo : nullptr, msg_len);
  });
}

keaki_status keaki_hip_group_decap_batch(keaki_hip_group* g, const uint64_t* proofs_aff, const uint64_t* cts_aff, size_t n, uint8_t* gt_out,
                                         uint8_t* key_out, size_t msg_len) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (n == 0) return KEAKI_OK;
  if (!proofs_aff || !cts_aff || (!gt_out && !key_out) || msg_len > 65536) return gfail(g, KEAKI_ERR_BAD_ARG, "group_decap_batch: bad argument");
  const size_t N = g->ctx.size();
  return run_all(g, "group_decap_batch", [&](size_t i) -> keaki_status {
    size_t lo, hi;
    range_of(n, N, i, &lo, &hi);
    if (hi == lo) return KEAKI_OK;
    return keaki_hip_decap_batch(g->ctx[i], proofs_aff + 8 * lo, cts_aff + 16 * lo, hi - lo, gt_out ? gt_out + 384 * lo : nullptr,
                                 key_out ? key_out + msg_len * lo : nullptr, msg_len);
  });
}

// enc::encrypt / enc::decrypt over the batch (KEM + XOR DEM on the devices), by item range
keaki_status keaki_hip_group_encrypt_batch(keaki_hip_group* g, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* points,
                                           const uint64_t* values, const uint64_t* r, const uint8_t* msgs, size_t n, uint64_t* ct_out_aff,
                                           uint8_t* body_out, size_t msg_len) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (n == 0) return KEAKI_OK;
  if (!com_aff || !tau_g2_aff || !points || !values || !r || !msgs || !ct_out_aff || !body_out || msg_len == 0 || msg_len > 65536)
    return gfail(g, KEAKI_ERR_BAD_ARG, "group_encrypt_batch: bad argument");
  const size_t N = g->ctx.size();
  return run_all(g, "group_encrypt_batch", [&](size_t i) -> keaki_status {
    size_t lo, hi;
    range_of(n, N, i, &lo, &hi);
    if (hi == lo) return KEAKI_OK;
    return keaki_hip_encrypt_batch(g->ctx[i], com_aff, tau_g2_aff, points + 4 * lo, values + 4 * lo, r + 4 * lo, msgs + msg_len * lo, hi - lo,
                                   ct_out_aff + 16 * lo, body_out + msg_len * lo, msg_len);
  });
}
keaki_status keaki_hip_group_decrypt_batch(keaki_hip_group* g, const uint64_t* proofs_aff, const uint64_t* cts_aff, const uint8_t* bodies, size_t n,
                                           uint8_t* msgs_out, size_t msg_len) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (n == 0) return KEAKI_OK;
  if (!proofs_aff || !cts_aff || !bodies || !msgs_out || msg_len == 0 || msg_len > 65536) return gfail(g, KEAKI_ERR_BAD_ARG, "group_decrypt_batch: bad argument");
  const size_t N = g->ctx.size();
  return run_all(g, "group_decrypt_batch", [&](size_t i) -> keaki_status {
    size_t lo, hi;
    range_of(n, N, i, &lo, &hi);
    if (hi == lo) return KEAKI_OK;
    return keaki_hip_decrypt_batch(g->ctx[i], proofs_aff + 8 * lo, cts_aff + 16 * lo, bodies + msg_len * lo, hi - lo, msgs_out + msg_len * lo, msg_len);
  });
}

// ---- FK23 (kzg::open_fk, src/kzg.rs:157-203) over the members of a group -------------------------------------------------------------------
// The sharded pipeline of keaki_hip_fk_shard_* (member i = rank i: 1/N of the butterflies of both size-d transforms and of the 2d scalar-mults)
// with the exchanges done HERE, in-process: an all-to-all is N^2 device-to-device copies between the members' buffers (hipMemcpyPeer; plain
// device copies when two members share a GPU), the final all-gather lands in member 0's buffer only, which writes the d proofs to the host.
// Needs N a power of two and d >= N^2; any other shape runs un-sharded on member 0.
}  // extern "C"

struct keaki_hip_group_fk {
  uint32_t log2d = 0;
  bool sharded = false;
  std::vector<keaki_hip_srs_g1*> srs;       // srs[0..d) on every member (sharded) or on member 0 only
  std::vector<keaki_hip_fk_shard*> fk;
  std::vector<void*> send, recv;            // sizes[0] bytes each, on the member's device
  size_t sizes[4] = {0, 0, 0, 0};
  uint64_t om[4], omi[4], inv2d[4];
};

namespace {
// One asynchronous copy on the SOURCE member's stream (behind the step that filled the buffer): a device-to-device DMA where the two devices
// have peer access (keaki_hip_group_create enabled it), a plain device copy when both members share a GPU.
keaki_status copy_between(keaki_hip_group* g, size_t dst_m, void* dst, size_t src_m, const void* src, size_t bytes) {
  const int dd = g->ctx[dst_m]->device, sd = g->ctx[src_m]->device;
  keaki_internal::DeviceScope dev_(sd);
  const hipStream_t st = g->ctx[src_m]->stream;
  const hipError_t e = dd == sd ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) : hipMemcpyPeerAsync(dst, dd, src, sd, bytes, st);
  if (e != hipSuccess) return gfail(g, KEAKI_ERR_HIP, "group_fk: copy member %zu -> member %zu failed: %s", src_m, dst_m, hipGetErrorString(e));
  return KEAKI_OK;
}
// Every member has issued its outgoing copies on its own stream: an event per source, and every destination's stream waits for the sources
// that wrote into its buffer. No host thread blocks; the next step's kernels start when their inputs have landed.
keaki_status copies_ordered(keaki_hip_group* g, const std::vector<size_t>& dests) {
  const size_t N = g->ctx.size();
  for (size_t i = 0; i < N; i++) {
    keaki_internal::DeviceScope dev_(g->ctx[i]->device);
    const hipError_t e = hipEventRecord(g->sent[i], g->ctx[i]->stream);
    if (e != hipSuccess) return gfail(g, KEAKI_ERR_HIP, "group_fk: hipEventRecord on member %zu failed: %s", i, hipGetErrorString(e));
  }
  for (size_t j : dests) {
    keaki_internal::DeviceScope dev_(g->ctx[j]->device);
    for (size_t i = 0; i < N; i++) {
      if (i == j) continue;                               // its own stream is ordered already
      const hipError_t e = hipStreamWaitEvent(g->ctx[j]->stream, g->sent[i], 0);
      if (e != hipSuccess) return gfail(g, KEAKI_ERR_HIP, "group_fk: hipStreamWaitEvent on member %zu failed: %s", j, hipGetErrorString(e));
    }
  }
  return KEAKI_OK;
}
// chunk j of member i's send buffer -> chunk i of member j's receive buffer. Every member has finished the step before (step_all waits for
// the members' streams), so the receive buffers are free to be written.
keaki_status all_to_all(keaki_hip_group* g, keaki_hip_group_fk* f, size_t per_peer) {
  const size_t N = g->ctx.size();
  std::vector<size_t> all(N);
  for (size_t i = 0; i < N; i++) {
    all[i] = i;
    for (size_t k = 0; k < N; k++) {
      const size_t j = (i + k) % N;                       // every source starts with a different destination: the links are used at once
      ST_TRY(copy_between(g, j, (char*)f->recv[j] + i * per_peer, i, (const char*)f->send[i] + j * per_peer, per_peer));
    }
  }
  return copies_ordered(g, all);
}
void group_fk_release(keaki_hip_group* g, keaki_hip_group_fk* f) {
  for (size_t i = 0; i < f->fk.size(); i++) keaki_hip_fk_shard_free(g && i < g->ctx.size() ? g->ctx[i] : nullptr, f->fk[i]);
  for (size_t i = 0; i < f->srs.size(); i++) keaki_hip_srs_g1_free(g && i < g->ctx.size() ? g->ctx[i] : nullptr, f->srs[i]);
  for (size_t i = 0; i < f->send.size(); i++) {
    keaki_internal::DeviceScope dev_(g && i < g->ctx.size() ? g->ctx[i]->device : -1);
    if (f->send[i]) (void)hipFree(f->send[i]);
    if (f->recv[i]) (void)hipFree(f->recv[i]);
  }
  delete f;
}
// a step on every member, each followed by a synchronisation of the member's stream (the copies that follow read the buffers)
keaki_status step_all(keaki_hip_group* g, const char* what, const std::function<keaki_status(size_t)>& fn) {
  return run_all(g, what, [&](size_t i) -> keaki_status {
    const keaki_status st = fn(i);
    return st != KEAKI_OK ? st : keaki_hip_synchronize(g->ctx[i]);
  });
}
}  // namespace

extern "C" {

keaki_status keaki_hip_group_fk_create(keaki_hip_group* g, const uint64_t* points_aff, uint32_t log2d, const uint64_t* omega_2d, const uint64_t* omega_2d_inv,
                                       const uint64_t* inv_2d, keaki_hip_group_fk** out) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (!points_aff || !omega_2d || !omega_2d_inv || !inv_2d || !out || log2d > 27) return gfail(g, KEAKI_ERR_BAD_ARG, "group_fk_create: bad argument");
  *out = nullptr;
  const size_t N = g->ctx.size(), d = (size_t)1 << log2d;
  auto* f = new keaki_hip_group_fk();
  f->log2d = log2d;
  memcpy(f->om, omega_2d, 32); memcpy(f->omi, omega_2d_inv, 32); memcpy(f->inv2d, inv_2d, 32);
  f->sharded = N >= 2 && (N & (N - 1)) == 0 && d >= N * N;
  keaki_status st;
  if (!f->sharded) {
    f->srs.assign(1, nullptr);
    st = keaki_hip_srs_g1_upload(g->ctx[0], points_aff, d, &f->srs[0]);
    if (st == KEAKI_OK) st = keaki_hip_srs_g1_precompute_fk(g->ctx[0], f->srs[0], log2d, omega_2d);
    if (st != KEAKI_OK) { gfail(g, st, "group_fk_create: member 0: %s", keaki_hip_last_error(g->ctx[0])); group_fk_release(g, f); return st; }
    *out = f;
    return KEAKI_OK;
  }
  f->srs.assign(N, nullptr); f->fk.assign(N, nullptr); f->send.assign(N, nullptr); f->recv.assign(N, nullptr);
  st = run_all(g, "group_fk_create", [&](size_t i) -> keaki_status {
    ST_TRY(keaki_hip_srs_g1_upload(g->ctx[i], points_aff, d, &f->srs[i]));
    ST_TRY(keaki_hip_fk_shard_create(g->ctx[i], f->srs[i], log2d, (uint32_t)i, (uint32_t)N, omega_2d, omega_2d_inv, inv_2d, &f->fk[i]));
    size_t sz[4];
    ST_TRY(keaki_hip_fk_shard_sizes(f->fk[i], sz));
    if (i == 0) memcpy(f->sizes, sz, sizeof sz);
    (void)hipSetDevice(g->ctx[i]->device);
    ST_TRY(keaki_internal::dev_alloc(g->ctx[i], &f->send[i], sz[0]));
    return keaki_internal::dev_alloc(g->ctx[i], &f->recv[i], sz[0]);
  });
  // the SRS-only transform: every member's part, one all-to-all
  if (st == KEAKI_OK) st = step_all(g, "group_fk_create (setup 0)", [&](size_t i) { return keaki_hip_fk_shard_setup(g->ctx[i], f->fk[i], 0, f->send[i], nullptr); });
  if (st == KEAKI_OK) st = all_to_all(g, f, f->sizes[1]);
  if (st == KEAKI_OK) st = step_all(g, "group_fk_create (setup 1)", [&](size_t i) { return keaki_hip_fk_shard_setup(g->ctx[i], f->fk[i], 1, nullptr, f->recv[i]); });
  if (st != KEAKI_OK) { group_fk_release(g, f); return st; }
  *out = f;
  return KEAKI_OK;
}

void keaki_hip_group_fk_free(keaki_hip_group* g, keaki_hip_group_fk* fk) {
  if (!fk) return;
  if (g) { std::lock_guard<std::mutex> lk(g->mu); group_fk_release(g, fk); }
  else group_fk_release(nullptr, fk);
}

keaki_status keaki_hip_group_fk_open(keaki_hip_group* g, keaki_hip_group_fk* f, const uint64_t* coeffs, uint64_t* proofs_out_aff) {
  if (!g) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  if (!f || !coeffs || !proofs_out_aff) return gfail(g, KEAKI_ERR_BAD_ARG, "group_fk_open: null pointer");
  if (!f->sharded) {
    const keaki_status st = keaki_hip_open_fk_poly(g->ctx[0], f->srs[0], f->log2d, coeffs, f->om, f->omi, f->inv2d, proofs_out_aff);
    if (st != KEAKI_OK) return gfail(g, st, "group_fk_open: member 0: %s", keaki_hip_last_error(g->ctx[0]));
    return KEAKI_OK;
  }
  const size_t N = g->ctx.size();
  if (f->fk.size() != N) return gfail(g, KEAKI_ERR_BAD_ARG, "group_fk_open: the handle belongs to a group of another size");
  ST_TRY(step_all(g, "group_fk_open (step 0)", [&](size_t i) { return keaki_hip_fk_shard_open(g->ctx[i], f->fk[i], 0, coeffs, f->send[i], nullptr, nullptr); }));
  ST_TRY(all_to_all(g, f, f->sizes[2]));
  ST_TRY(step_all(g, "group_fk_open (step 1)", [&](size_t i) { return keaki_hip_fk_shard_open(g->ctx[i], f->fk[i], 1, nullptr, f->send[i], f->recv[i], nullptr); }));
  ST_TRY(all_to_all(g, f, f->sizes[2]));
  ST_TRY(step_all(g, "group_fk_open (step 2)", [&](size_t i) { return keaki_hip_fk_shard_open(g->ctx[i], f->fk[i], 2, nullptr, f->send[i], f->recv[i], nullptr); }));
  // the all-gather of the d / N affine proofs of every member, into member 0 only (it alone writes the host output)
  for (size_t i = 0; i < N; i++) ST_TRY(copy_between(g, 0, (char*)f->recv[0] + i * f->sizes[3], i, f->send[i], f->sizes[3]));
  ST_TRY(copies_ordered(g, {0}));
  const keaki_status st = keaki_hip_fk_shard_open(g->ctx[0], f->fk[0], 3, nullptr, nullptr, f->recv[0], proofs_out_aff);
  if (st != KEAKI_OK) return gfail(g, st, "group_fk_open (step 3): member 0: %s", keaki_hip_last_error(g->ctx[0]));
  return KEAKI_OK;
}

}  // extern "C"
