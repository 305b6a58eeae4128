// keaki_hip_group_*: in-process multi-GPU over the single-context C ABI (include/keaki_hip.h). Host code only: one context and one
// worker thread per member, ranges of the SRS / of the batch per member, partial sums through host memory. No collective, no RCCL:
// inside one process the 96-byte partials are a memcpy away.
#include "internal.h"

#include <algorithm>
#include <condition_variable>
#include <functional>
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
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) return;
      std::function<keaki_status()> j = std::move(job);
      has_job = false;
      lk.unlock();
      keaki_status r = j();
      lk.lock();
      result = r;
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
  std::mutex mu;                 // calls on one group run one after another
  std::string err;
};
struct keaki_hip_group_srs_g1 {
  size_t n = 0;
  std::vector<keaki_hip_srs_g1*> chunk;   // member i: points [lo(i), hi(i))
  std::vector<size_t> lo, hi;
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
      gfail(g, st, "%s: member %zu (device %d): %s", what, i, g->ctx[i]->device, keaki_hip_last_error(g->ctx[i]));
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
    w->th = std::thread([w] { w->loop(); });
    g->workers.push_back(w);
  }
  *out = g;
  return KEAKI_OK;
}

void keaki_hip_group_destroy(keaki_hip_group* g) {
  if (!g) return;
  for (Worker* w : g->workers) {
    { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_all(); }
    w->th.join();
    delete w;
  }
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
  s->chunk.assign(N, nullptr); s->lo.resize(N); s->hi.resize(N);
  for (size_t i = 0; i < N; i++) range_of(n, N, i, &s->lo[i], &s->hi[i]);
  const keaki_status st = run_all(g, "group_srs_g1_upload", [&](size_t i) -> keaki_status {
    const size_t m = s->hi[i] - s->lo[i];
    keaki_status r = keaki_hip_srs_g1_upload(g->ctx[i], m ? points_aff + 8 * s->lo[i] : nullptr, m, &s->chunk[i]);
    if (r != KEAKI_OK || !precompute || !m) return r;
    r = keaki_hip_srs_g1_precompute(g->ctx[i], s->chunk[i], nullptr);
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
  std::vector<uint64_t> q(4 * std::max<size_t>(nq, 1));
  const keaki_status st = keaki_hip_kzg_quotient(g->ctx[0], coeffs, n, point, q.data(), value_out);
  if (st != KEAKI_OK) return gfail(g, st, "group_kzg_open: quotient: %s", keaki_hip_last_error(g->ctx[0]));
  return group_msm_locked(g, srs, q.data(), nq, proof_out_jac);
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

}  // extern "C"
