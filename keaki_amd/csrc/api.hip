// libkeaki_hip.so -- C ABI (include/keaki_hip.h) over the gfx950 kernels. Host side only does
// launch plumbing: workspace management, stream ordering, error codes. No arithmetic happens on
// the CPU here and there is no CPU fallback: without a gfx950 device every entry point fails.
#include "internal.h"

#include <algorithm>
#include <set>
#include <utility>
#include <vector>

using namespace keaki_internal;

// An SRS handle is device memory, not context state: every context ON THE SAME DEVICE may pass it to msm / open / open_fk (read-only use
// of the points, the window tables and the cached FK23 transform), so N host threads with a context each share ONE set of tables.
// `mu` guards the lazily built members (table, fk_hat_s); `acct` is the context whose keaki_hip_ctx_memory counts them.
struct keaki_hip_srs_g1 {
  const void* d = nullptr;
  size_t n = 0;
  bool owned = false;
  int device = -1;
  std::recursive_mutex mu;
  keaki_hip_ctx* acct = nullptr;
  size_t fk_bytes = 0;
  // precomputed window tables (keaki_hip_srs_g1_precompute): table[w * n + i] = 2^(offset_w) * P_i
  void* table = nullptr;
  size_t table_bytes = 0;
  int c_table = 0;
  // FK23: hat_s = DFT_2d(reversed SRS) for the last requested d (2d Jacobian points), reused by later keaki_hip_open_fk calls
  void* fk_hat_s = nullptr;
  int fk_log2d = -1;
};
struct keaki_hip_srs_g2 {
  const void* d = nullptr;
  size_t n = 0;
  bool owned = false;
  int device = -1;
  std::recursive_mutex mu;
  keaki_hip_ctx* acct = nullptr;
  void* table = nullptr;          // window tables (keaki_hip_srs_g2_precompute), as for G1
  size_t table_bytes = 0;
  int c_table = 0;
};

#include <dlfcn.h>
#include <sys/mman.h>
#include <memory>
#include <thread>
#include <atomic>
#include <chrono>
namespace {
// roctx ranges around the kernel families (tracing), visible to `rocprofv3 --marker-trace`. The marker library is
// looked up at run time so that the ABI has no link-time dependency on the profiler; without it the scopes are no-ops.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
    if (h) {
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!push || !pop) push = nullptr;
    }
  }
};
const Roctx& roctx() { static Roctx r; return r; }
struct RoctxScope {
  bool on;
  explicit RoctxScope(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~RoctxScope() { if (on) roctx().pop(); }
};
#define TRACE_SCOPE(name) RoctxScope roctx_scope_(name)
thread_local std::string g_create_error;
// The contexts that are alive: an SRS handle is shared by every context of its device, and whoever frees it (or rebuilds its FK23 transform)
// must be able to undo the byte accounting on the context that built the tables -- if that context still exists.
std::mutex g_live_mu;
std::set<keaki_hip_ctx*> g_live_ctx;
// runs `f(acct)` when acct is a live context. Under g_live_mu ONLY: the caller may already hold its own context's lock, and taking another
// context's lock from here would order the two locks both ways (thread A: ctx1 -> live -> ctx2, thread B: ctx2 -> live). What f touches --
// the byte count `mem_tables` -- is an atomic for that reason.
template <class Fn>
void with_live_ctx(keaki_hip_ctx* acct, Fn f) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  if (acct && g_live_ctx.count(acct)) f(acct);
}
// saturating subtraction on the byte count (a handle freed twice over, or booked before a trim, must not wrap it)
void mem_sub(std::atomic<size_t>& m, size_t v) {
  size_t cur = m.load();
  while (!m.compare_exchange_weak(cur, cur - std::min(cur, v))) {}
}
constexpr size_t G1_AFF_BYTES = 64, G2_AFF_BYTES = 128;
}  // namespace

namespace keaki_internal {

keaki_status fail(keaki_hip_ctx* ctx, keaki_status code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_create_error = buf;
  return code;
}

keaki_status dev_alloc(keaki_hip_ctx* ctx, void** p, size_t bytes) {
  if (ctx && ctx->tune.alloc_limit && bytes > ctx->tune.alloc_limit)
    return fail(ctx, KEAKI_ERR_OOM, "allocation of %zu bytes refused by keaki_hip_debug_set_alloc_limit(%zu)", bytes, ctx->tune.alloc_limit);
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, e == hipErrorOutOfMemory ? KEAKI_ERR_OOM : KEAKI_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  }
  return KEAKI_OK;
}

keaki_status reserve(keaki_hip_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return KEAKI_OK;
  if (b.p) {
    // the buffer may still be in use by enqueued work
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  ST_TRY(dev_alloc(ctx, &b.p, want));
  b.cap = want;
  return KEAKI_OK;
}

keaki_status launch_check(keaki_hip_ctx* ctx, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ctx, KEAKI_ERR_HIP, "launch %s failed: %s", what, hipGetErrorString(e));
  return KEAKI_OK;
}

}  // namespace keaki_internal

namespace {

void resolve_timing(keaki_hip_ctx* ctx) {
  if (!ctx->timing_pending) return;
  if (hipEventSynchronize(ctx->ev[3]) == hipSuccess) {
    (void)hipEventElapsedTime(&ctx->last_bucket_ms, ctx->ev[1], ctx->ev[2]);
    (void)hipEventElapsedTime(&ctx->last_total_ms, ctx->ev[0], ctx->ev[3]);
  }
  ctx->timing_pending = false;
}
void resolve_fk_timing(keaki_hip_ctx* ctx) {
  if (!ctx->fk_timing_pending) return;
  if (hipEventSynchronize(ctx->fk_ev[3]) == hipSuccess) {
    (void)hipEventElapsedTime(&ctx->last_fk_ms[0], ctx->fk_ev[0], ctx->fk_ev[1]);
    (void)hipEventElapsedTime(&ctx->last_fk_ms[1], ctx->fk_ev[1], ctx->fk_ev[2]);
    (void)hipEventElapsedTime(&ctx->last_fk_ms[2], ctx->fk_ev[0], ctx->fk_ev[3]);
  }
  ctx->fk_timing_pending = false;
}

keaki_status upload(keaki_hip_ctx* ctx, DevBuf& b, const void* host, size_t bytes) {
  ST_TRY(reserve(ctx, b, bytes ? bytes : 16));
  if (bytes) HIP_TRY(ctx, hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return KEAKI_OK;
}
keaki_status download(keaki_hip_ctx* ctx, void* host, const void* dev, size_t bytes) {
  if (bytes) HIP_TRY(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return KEAKI_OK;
}

// (table, window target) of a handle, read under its lock: another context may be building the tables right now
template <class H>
std::pair<const void*, int> srs_tables(const H* srs) {
  std::lock_guard<std::recursive_mutex> hl(const_cast<H*>(srs)->mu);
  return {srs->table, srs->c_table};
}
// bookkeeping after a call that may have (re)built the cached FK23 transform of a handle (2d Jacobian points)
void fk_account(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs) {
  const size_t now = srs->fk_hat_s && srs->fk_log2d >= 0 ? ((size_t)2 << srs->fk_log2d) * 96 : 0;
  if (now == srs->fk_bytes) return;
  if (!srs->acct) srs->acct = ctx;
  const size_t before = srs->fk_bytes;
  with_live_ctx(srs->acct, [&](keaki_hip_ctx* a) { mem_sub(a->mem_tables, before); a->mem_tables += now; });   // whichever context rebuilt it
  srs->fk_bytes = now;
}
template <class H>
H* new_srs(keaki_hip_ctx* ctx, const void* d, size_t n, bool owned) {
  H* h = new H();
  h->d = d; h->n = n; h->owned = owned; h->device = ctx->device; h->acct = ctx;
  return h;
}
// a handle may be used by any context of the device it lives on
#define SRS_CHECK(ctx, srs, what)                                                                                                     \
  if ((srs)->device != (ctx)->device)                                                                                                 \
    return fail(ctx, KEAKI_ERR_BAD_ARG, what ": the SRS handle lives on device %d, this context on device %d", (srs)->device, (ctx)->device)

// The caller's OUTPUT buffer is usually fresh memory (calloc / vec![0; n] / numpy.zeros): its pages do not exist until first touched, and a
// device-to-host copy into such pages crawls (160 MB of ciphertexts: 30 ms instead of 3). Touch one byte per page from the host WHILE the
// kernels that produce the data are still running: the faults are taken off the critical path. The whole range is overwritten by the copy
// that follows. (Resident pages cost ~2 ns each.)
// Fresh pages are also asked to be transparent HUGE pages (madvise(MADV_HUGEPAGE) on the 2 MB-aligned interior: a hint, ignored where the
// kernel does not offer it): first touch of 160 MB takes 20.5 ms in 4 KB pages and 6.5 ms in 2 MB pages on the GPU boxes
// (bench_tools/ubench_thp_touch.py). At 2^20 items the faults hide behind the kernels either way (vec_encrypt 52.6 vs 52.0 ms); the hint matters
// where the output is large against the kernel time (GT bytes out: 384 B per item).
void prefault_out(const keaki_hip_ctx* ctx, void* p, size_t bytes) {
  if (!ctx->tune.host_prefault || !p || bytes < (1u << 20)) return;    // option host_prefault = 0: the library never writes to (or madvises) caller memory itself
  {
    // memory the HIP runtime knows (hipHostMalloc / hipHostRegister: resident by construction, and copies from and to it are truly asynchronous,
    // so a write from here could overtake an upload still reading the same array): nothing to touch
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) {
      if (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeDevice) return;
    } else {
      (void)hipGetLastError();
    }
  }
  {
    const uintptr_t HP = (uintptr_t)2 << 20, a = ((uintptr_t)p + HP - 1) & ~(HP - 1), e = ((uintptr_t)p + bytes) & ~(HP - 1);
    if (e > a) (void)madvise((void*)a, e - a, MADV_HUGEPAGE);
  }
  volatile unsigned char* c = (volatile unsigned char*)p;
  for (size_t off = 0; off < bytes; off += 4096) c[off] = 0;
  c[bytes - 1] = 0;
}

// ---- host-pointer batches in CHUNKS ------------------------------------------------------------------------------------------
// A batch call that takes host arrays is upload -> kernels -> download; done in that order the device idles during both copies and the host
// thread during the kernels (vec_encrypt of 2^20 items: 8.1 ms per 2^18-item piece for 7.0 ms of kernels, profiles/r04_vec_encrypt_timeline.txt).
// Batches of two chunks or more run as a pipeline over two buffer halves: the upload of chunk k + 1 and the download of chunk k - 1 go
// through a copy stream of the context's own while the kernels of chunk k run on the context's stream. The host arrays are pageable, so a
// copy call returns when the runtime has staged (upload) or delivered (download) the bytes: one host thread is enough, and the order of its
// calls -- upload k, launch k, download k - 1 -- is what keeps the device busy. `up(lo, m, half, stream)` enqueues the uploads of items
// [lo, lo + m) into buffer half `half`, `run(lo, m, half)` the kernels (on ctx->stream), `down(lo, m, half, stream)` first-touches the
// caller's output pages and enqueues the downloads.
constexpr size_t PIPE_CHUNK = 65536;
static keaki_status pipe_ready(keaki_hip_ctx* ctx) {
  if (ctx->copy_stream) return KEAKI_OK;
  hipStream_t cs = nullptr;
  HIP_TRY(ctx, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) {
    if (!ctx->pipe_in[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_in[i], hipEventDisableTiming));
    if (!ctx->pipe_done[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_done[i], hipEventDisableTiming));
  }
  ctx->copy_stream = cs;
  return KEAKI_OK;
}
// chunk size of a batch of n items: `unit` items (PIPE_CHUNK: two rounds of the GT exponentiation kernel; the pairing path passes its own launch
// size -- 16 launches of 2^16 pairings take 3.6 ms longer than 8 of 2^17), the whole batch below two units
inline size_t pipe_chunk_items(const keaki_hip_ctx* ctx, size_t n, size_t unit = PIPE_CHUNK) { return ctx->tune.pipe_chunks && n >= 2 * unit ? unit : n; }
// `touch(lo, m)`: first-touch the caller's output ranges of items [lo, lo + m) (prefault_out). A download into pages that do not exist yet runs
// at 5 GB/s instead of 56 (bench_tools/ubench_pageable_copy_sizes.py), and touching them costs the host 40-125 us per MB: with one chunk that
// happens on the calling thread while the kernels run; with more, helper threads walk the chunks ahead of the downloads (one thread, three
// when the outputs exceed 64 MB: GT bytes out are 384 B per item) and a download waits for its chunk's flag, so no touch can land on delivered bytes.
template <class Up, class Run, class Touch, class Down>
static keaki_status pipelined(keaki_hip_ctx* ctx, size_t n, size_t ch, size_t out_bytes_per_item, Up up, Run run, Touch touch, Down down) {
  ST_TRY(pipe_ready(ctx));
  const size_t chunks = (n + ch - 1) / ch;
  hipStream_t cs = ctx->copy_stream, st = ctx->stream;
  const size_t n_helpers = chunks < 2 || !ctx->tune.host_prefault ? 0 : (n * out_bytes_per_item >= ((size_t)64 << 20) ? std::min<size_t>(3, chunks) : 1);
  std::unique_ptr<std::atomic<unsigned char>[]> touched(new std::atomic<unsigned char>[chunks]);
  for (size_t k = 0; k < chunks; k++) touched[k].store(0, std::memory_order_relaxed);
  // a helper touches (writes into) the output pages of chunk k only after chunk k's inputs have been read: a caller may pass one array as input
  // and output (messages in, bodies out). `staged` = chunks whose upload calls have returned (pageable: the source has been consumed by then);
  // `give_up` releases the helpers when the call leaves early.
  std::atomic<size_t> staged{0};
  std::atomic<bool> give_up{false};
  struct Helpers {
    std::vector<std::thread> t;
    std::atomic<bool>* give_up;
    ~Helpers() { give_up->store(true); for (auto& x : t) if (x.joinable()) x.join(); }
  } helpers{{}, &give_up};
  for (size_t h = 0; h < n_helpers; h++)
    helpers.t.emplace_back([&, h] {
      for (size_t k = h; k < chunks; k += n_helpers) {
        // yield first, not sleep: sleep_for(50 us) wakes late enough under load to cost a 2^20-item call 20 ms (measured: encap with GT out
        // 57 -> 73 ms). A wait that outlasts ~2 ms (the device is far behind: nothing to gain from a hot loop) falls back to short sleeps.
        for (unsigned spins = 0; staged.load(std::memory_order_acquire) <= k; spins++) {
          if (give_up.load(std::memory_order_relaxed)) return;
          if (spins < 20000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        touch(k * ch, std::min(ch, n - k * ch));
        touched[k].store(1, std::memory_order_release);
      }
    });
  // the copy stream starts behind whatever the context's stream holds (an earlier call's kernels may still read the buffers)
  HIP_TRY(ctx, hipEventRecord(ctx->pipe_done[0], st));
  HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->pipe_done[0], 0));
  for (size_t k = 0; k <= chunks; k++) {
    if (k < chunks) {
      const size_t lo = k * ch, m = std::min(ch, n - lo);
      const int h = (int)(k & 1);
      ST_TRY(up(lo, m, h, cs));
      staged.store(k + 1, std::memory_order_release);
      HIP_TRY(ctx, hipEventRecord(ctx->pipe_in[h], cs));
      HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->pipe_in[h], 0));
      ST_TRY(run(lo, m, h));
      HIP_TRY(ctx, hipEventRecord(ctx->pipe_done[h], st));
    }
    if (k >= 1) {
      const size_t lo = (k - 1) * ch, m = std::min(ch, n - lo);
      const int h = (int)((k - 1) & 1);
      if (!n_helpers) touch(lo, m);
      else for (unsigned spins = 0; !touched[k - 1].load(std::memory_order_acquire); spins++) {
        if (spins < 20000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
      HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->pipe_done[h], 0));
      ST_TRY(down(lo, m, h, cs));
    }
  }
  HIP_TRY(ctx, hipStreamSynchronize(cs));
  return KEAKI_OK;
}


// ---- MSM of a scalar vector in HOST memory -----------------------------------------------------------------------------------------
// kzg::commit hands over a polynomial that lives in host memory (reference src/kzg.rs:89-101): the call is upload -> MSM, and done in
// that order the 512 MiB of a 2^24-term polynomial cost 11.5 ms of copy in front of 16.7 ms of kernels (BENCH_r04: 5.96e8/s against
// 1.007e9/s resident). From `msm_pipe_min` scalars on the vector goes up in point-range chunks through the copy stream and the MSM
// runs chunk by chunk behind it (msm_host.hip.h: MsmPipe): the upload of chunk j + 1 hides under the kernels of chunk j, only the
// first chunk's copy stays in front. Chunks GROW (the copy is faster than the kernels, so a short first chunk starts the device early
// and every later copy still finishes before the device asks for it); `run(pipe)` enqueues the MSM over ctx->io_a.
// A pageable source makes every copy call return once its bytes are staged; a pinned one returns at once -- the order of the host's
// calls (copy j, kernels j, copy j + 1, ...) serves both.
static std::vector<size_t> msm_pipe_bounds(const Tuning& t, size_t n) {
  size_t k = 1;
  if (t.msm_pipe_chunks >= 2) k = (size_t)t.msm_pipe_chunks;
  else if (t.msm_pipe_chunks < 0 && t.pipe_chunks && n >= (size_t)t.msm_pipe_min) k = n >= ((size_t)1 << 22) ? 6 : n >= ((size_t)1 << 21) ? 4 : 3;      // measured: profiles/r05_msm_pipe_sweep_*.txt
  if (k > 64) k = 64;
  if (k > n) k = n ? n : 1;
  std::vector<size_t> b{0};
  if (k >= 2) {
    const double g = std::max(100, std::min(400, t.msm_pipe_growth)) / 100.0;
    double tot = 0, w = 1;
    for (size_t j = 0; j < k; j++, w *= g) tot += w;
    double acc = 0;
    w = 1;
    for (size_t j = 0; j + 1 < k; j++, w *= g) {
      acc += w;
      size_t e = (size_t)((double)n * acc / tot);
      if (n >= 65536) e &= ~(size_t)4095;                   // whole pages of scalars, whole tiles of the first sort
      if (e > b.back() && e < n) b.push_back(e);
    }
  }
  b.push_back(n);
  return b;
}
// the copy-stream side of a chunked upload: `begin` orders the copy stream behind the context's stream, `chunk` copies one piece and makes
// the context's stream wait for it; on every exit no copy reads the caller's array any more (the destructor drains the copy stream)
struct ChunkUploader {
  keaki_hip_ctx* ctx;
  hipStream_t cs = nullptr;
  explicit ChunkUploader(keaki_hip_ctx* c) : ctx(c) {}
  ~ChunkUploader() { if (cs) (void)hipStreamSynchronize(cs); }
  keaki_status begin() {
    ST_TRY(pipe_ready(ctx));
    // the copy stream starts behind whatever the context's stream holds (an earlier call's kernels may still read the destination)
    HIP_TRY(ctx, hipEventRecord(ctx->pipe_done[0], ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->pipe_done[0], 0));
    cs = ctx->copy_stream;
    return KEAKI_OK;
  }
  keaki_status chunk(size_t j, void* dst, const void* src, size_t bytes) {
    const int h = (int)(j & 1);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, hipEventRecord(ctx->pipe_in[h], cs));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_in[h], 0));
    return KEAKI_OK;
  }
};
template <class Run>
static keaki_status msm_from_host(keaki_hip_ctx* ctx, const uint64_t* scalars, size_t n, Run run) {
  ST_TRY(reserve(ctx, ctx->io_a, n ? n * 32 : 16));
  MsmPipe pipe;
  const std::vector<size_t> bounds = msm_pipe_bounds(ctx->tune, n);
  for (size_t j = 0; j + 1 < bounds.size(); j++) pipe.ranges.push_back({bounds[j], bounds[j + 1] - bounds[j]});
  if (pipe.ranges.size() <= 1) {
    if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    const keaki_status st = run(nullptr);
    // a FAILED call returns as well only when no copy reads the caller's array any more (header; with a pinned source the copy above is truly
    // asynchronous): the successful path synchronises when it downloads the result, the failing one here
    if (st != KEAKI_OK && n) (void)hipStreamSynchronize(ctx->stream);
    return st;
  }
  ChunkUploader up(ctx);
  ST_TRY(up.begin());
  pipe.stage = [&](size_t j) -> keaki_status {
    const size_t lo = pipe.ranges[j].first, m = pipe.ranges[j].second;
    return up.chunk(j, (char*)ctx->io_a.p + lo * 32, (const char*)scalars + lo * 32, m * 32);
  };
  return run(&pipe);
}

#define CTX_GUARD(ctx)                                \
  if (!(ctx)) return KEAKI_ERR_BAD_ARG;               \
  std::lock_guard<std::recursive_mutex> lock_((ctx)->mu);       \
  keaki_internal::DeviceScope dev_((ctx)->device);              \
  if (!dev_.ok) return fail(ctx, KEAKI_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device)

}  // namespace

extern "C" {

#ifndef KEAKI_SRC_HASH
#define KEAKI_SRC_HASH "unknown"
#endif
// "... src=<hash>": the hash of the kernel sources this binary was built from (csrc/Makefile, bench_tools/srchash.py); the profile
// collectors stamp their output with it and bench.py refuses figures measured on another build.
#ifdef KEAKI_DIAG
const char* keaki_hip_version(void) { return "keaki-hip 0.3 (gfx950) DIAGNOSTIC BUILD (diag_row_mask available: not the product) src=" KEAKI_SRC_HASH; }
#else
const char* keaki_hip_version(void) { return "keaki-hip 0.3 (gfx950) src=" KEAKI_SRC_HASH; }
#endif

extern "C++" {
namespace {
// The ONE place the library reads the environment: initial values of a context's tuning switches.
void tune_from_env(Tuning& t) {
  auto geti = [](const char* name, long long& out) { const char* e = getenv(name); if (!e || !*e) return false; out = atoll(e); return true; };
  long long v;
  if (geti("KEAKI_MSM_C", v)) t.msm_c = (int)v;
  if (geti("KEAKI_MSM_C_SHARED", v)) t.msm_c_shared = (int)v;
  if (geti("KEAKI_REDUCE_L", v)) t.reduce_l = (int)v;
  if (geti("KEAKI_PART_SHIFT", v)) t.part_shift = (int)v;
  if (geti("KEAKI_ACC_U29", v)) t.acc_u29 = v != 0;
  if (geti("KEAKI_ACC_U29_G2", v)) t.acc_u29_g2 = v != 0;
  if (geti("KEAKI_ACC_NT", v)) t.acc_nt = v != 0;
  if (geti("KEAKI_ACC_PREFETCH", v)) t.acc_prefetch = v != 0;
  if (geti("KEAKI_ACC_IDXQ", v)) t.acc_idxq = v != 0;
  if (geti("KEAKI_CS_MASKED", v)) t.cs_masked = v != 0;
  if (geti("KEAKI_FK_UNIFORM", v)) t.fk_uniform = v != 0;
  if (geti("KEAKI_FK_GTAB", v)) t.fk_gtab = v != 0;
  if (geti("KEAKI_FK_ADDSUB29", v)) t.fk_addsub29 = v != 0;
  if (geti("KEAKI_FK_RADIX4", v)) t.fk_radix4 = v != 0;
  if (geti("KEAKI_FB_OCC1", v)) t.fb_occ1 = v != 0;
  if (geti("KEAKI_MSM_SHORT_TABLES", v)) t.msm_short_tables = (int)v;
  if (geti("KEAKI_PAIR_WIDE_MAX", v)) t.pair_wide_max = (int)v;
  if (geti("KEAKI_PAIR_TWO_WAVES", v)) t.pair_two_waves = v != 0;
  if (geti("KEAKI_GT_WB_B", v)) t.gt_wb_b = (int)v;
  if (geti("KEAKI_ENCAP_GT", v)) t.encap_gt = v;
  if (geti("KEAKI_HOST_PREFAULT", v)) t.host_prefault = v != 0;
  if (geti("KEAKI_PIPE_CHUNKS", v)) t.pipe_chunks = v != 0;
  if (geti("KEAKI_MSM_PIPE_CHUNKS", v)) t.msm_pipe_chunks = (int)v;
  if (geti("KEAKI_MSM_PIPE_MIN", v)) t.msm_pipe_min = v;
  if (geti("KEAKI_MSM_PIPE_GROWTH", v)) t.msm_pipe_growth = (int)v;
}
struct BufClass { DevBuf* b; int cls; };   // cls: 1 = workspace, 2 = GT / fixed-base tables of encapsulate
std::vector<BufClass> all_bufs(keaki_hip_ctx* ctx) {
  std::vector<BufClass> v;
  for (DevBuf* b : {&ctx->digits, &ctx->hist, &ctx->offsets, &ctx->cursor, &ctx->sorted, &ctx->buckets, &ctx->acc29, &ctx->partials, &ctx->wsums, &ctx->bsums,
                    &ctx->tmp_a, &ctx->tmp_b, &ctx->tmp_c, &ctx->io_a, &ctx->io_b, &ctx->io_c, &ctx->io_d, &ctx->io_e, &ctx->perm, &ctx->heavy,
                    &ctx->pair_ws, &ctx->verify_io, &ctx->g2gen_lines, &ctx->fk_tab, &ctx->g2pow_lines, &ctx->g2pow_pts})
    v.push_back({b, 1});
  for (DevBuf* b : {&ctx->fb_bases, &ctx->fb_g1_gen, &ctx->fb_g2_gen, &ctx->fb_com, &ctx->fb_tau, &ctx->gt_tab_a, &ctx->gt_tab_b, &ctx->gt_base,
                    &ctx->fbs_g2_gen, &ctx->fbs_tau, &ctx->fbs_g1_gen})
    v.push_back({b, 2});
  return v;
}
}  // namespace
}  // extern "C++"

keaki_status keaki_hip_ctx_create(int32_t device, void* stream, keaki_hip_ctx** out) {
  if (!out) return fail(nullptr, KEAKI_ERR_BAD_ARG, "ctx_create: out is null");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, KEAKI_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(nullptr, KEAKI_ERR_BAD_ARG, "device %d out of range (%d visible)", device, ndev);
  keaki_internal::DeviceScope dev_(device);
  if (!dev_.ok) return fail(nullptr, KEAKI_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipDeviceProp_t prop;
  HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, KEAKI_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
  keaki_hip_ctx* ctx = new keaki_hip_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
  tune_from_env(ctx->tune);
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return fail(nullptr, KEAKI_ERR_HIP, "hipStreamCreate failed");
    }
    ctx->own_stream = true;
  }
  for (auto& e : ctx->ev) (void)hipEventCreate(&e);
  { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.insert(ctx); }
  *out = ctx;
  return KEAKI_OK;
}

void keaki_hip_ctx_destroy(keaki_hip_ctx* ctx) {
  if (!ctx) return;
  { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.erase(ctx); }
  {
  keaki_internal::DeviceScope dev_(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (const BufClass& bc : all_bufs(ctx))
    if (bc.b->p) (void)hipFree(bc.b->p);
  for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->fk_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->pipe_in) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->pipe_done) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->open_ev) if (e) (void)hipEventDestroy(e);
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  for (auto& e : ctx->aux_ev) if (e) (void)hipEventDestroy(e);
  if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  }
  delete ctx;
}

// The message is copied under the context lock into a buffer of the CALLING thread (valid until that thread's next call of this
// function), so a concurrent failing call on another thread cannot reallocate the string under the reader.
const char* keaki_hip_last_error(const keaki_hip_ctx* ctx) {
  if (!ctx) return g_create_error.c_str();
  thread_local std::string copy;
  keaki_hip_ctx* c = const_cast<keaki_hip_ctx*>(ctx);
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  copy = c->err;
  return copy.c_str();
}

keaki_status keaki_hip_ctx_set_option(keaki_hip_ctx* ctx, const char* name, int64_t value) {
  if (!ctx) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::recursive_mutex> lock_(ctx->mu);
  if (!name) return fail(ctx, KEAKI_ERR_BAD_ARG, "ctx_set_option: name is null");
  Tuning& t = ctx->tune;
  const std::string k(name);
  if (k == "msm_c") t.msm_c = (int)value;
  else if (k == "msm_c_shared") t.msm_c_shared = (int)value;
  else if (k == "reduce_l") t.reduce_l = (int)value;
  else if (k == "part_shift") t.part_shift = (int)value;
  else if (k == "acc_u29") t.acc_u29 = value != 0;
  else if (k == "acc_u29_g2") t.acc_u29_g2 = value != 0;
  else if (k == "acc_nt") t.acc_nt = value != 0;
  else if (k == "acc_prefetch") t.acc_prefetch = value != 0;
  else if (k == "acc_idxq") t.acc_idxq = value != 0;
  else if (k == "cs_masked") t.cs_masked = value != 0;
  else if (k == "fk_uniform") t.fk_uniform = value != 0;
  else if (k == "fk_gtab") t.fk_gtab = value != 0;
  else if (k == "fk_addsub29") t.fk_addsub29 = value != 0;
  else if (k == "fk_radix4") t.fk_radix4 = value != 0;
  else if (k == "fb_occ1") t.fb_occ1 = value != 0;
  else if (k == "msm_short_tables") t.msm_short_tables = (int)value;
  else if (k == "pair_wide_max") t.pair_wide_max = (int)value;
  else if (k == "pair_two_waves") t.pair_two_waves = value != 0;
  else if (k == "gt_wb_b") {
    if (value != 0 && (value < 8 || value > 22 || gt_table_powers((uint32_t)value) > 320)) return fail(ctx, KEAKI_ERR_BAD_ARG, "ctx_set_option: gt_wb_b = %lld out of range", (long long)value);
    if ((int)value != t.gt_wb_b) { ctx->gt_b_ready = false; ctx->gt_b_fallback = false; }      // the table of B is rebuilt at the new width on the next use
    t.gt_wb_b = (int)value;
  } else if (k == "encap_gt") t.encap_gt = value;
  else if (k == "host_prefault") t.host_prefault = value != 0;
  else if (k == "pipe_chunks") t.pipe_chunks = value != 0;
#ifdef KEAKI_DIAG
  else if (k == "diag_row_mask") t.diag_row_mask = (unsigned)value;
#endif
  else if (k == "msm_pipe_chunks") t.msm_pipe_chunks = (int)value;
  else if (k == "msm_pipe_min") t.msm_pipe_min = value;
  else if (k == "msm_pipe_growth") t.msm_pipe_growth = (int)value;
  else return fail(ctx, KEAKI_ERR_BAD_ARG, "ctx_set_option: unknown option '%s'", name);
  return KEAKI_OK;
}
keaki_status keaki_hip_debug_set_alloc_limit(keaki_hip_ctx* ctx, size_t bytes) {
  if (!ctx) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::recursive_mutex> lock_(ctx->mu);
  ctx->tune.alloc_limit = bytes;
  return KEAKI_OK;
}
// Releases every grow-only workspace and encapsulate table of the context (they come back on the next call that needs them; a table's
// rebuild costs what the first call cost). SRS handles are not touched.
keaki_status keaki_hip_ctx_trim(keaki_hip_ctx* ctx) {
  CTX_GUARD(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->aux_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream));      // a table build a failed call left behind
  if (ctx->copy_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
  for (const BufClass& bc : all_bufs(ctx))
    if (bc.b->p) { (void)hipFree(bc.b->p); bc.b->p = nullptr; bc.b->cap = 0; }
  ctx->gt_b_ready = ctx->gt_a_valid = ctx->gt_b_fallback = false;
  ctx->gt_a_pending_aux = false;
  ctx->seen_com_runs = 0;
  ctx->verify_ready = ctx->verify_tables_ready = false;
  ctx->fb_tau_valid = ctx->g2gen_lines_ready = ctx->fb_ready = ctx->g2pow_ready = false;
  ctx->fbs_ready = ctx->fbs_tau_valid = false;
  return KEAKI_OK;
}
keaki_status keaki_hip_ctx_memory(keaki_hip_ctx* ctx, size_t* out4) {
  if (!ctx) return KEAKI_ERR_BAD_ARG;
  std::lock_guard<std::recursive_mutex> lock_(ctx->mu);
  if (!out4) return fail(ctx, KEAKI_ERR_BAD_ARG, "ctx_memory: out4 is null");
  size_t ws = 0, gt = 0;
  for (const BufClass& bc : all_bufs(ctx)) (bc.cls == 1 ? ws : gt) += bc.b->cap;
  out4[0] = ctx->mem_tables; out4[1] = ws; out4[2] = gt; out4[3] = ctx->mem_tables + ws + gt;
  return KEAKI_OK;
}

void* keaki_hip_ctx_stream(const keaki_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int32_t keaki_hip_ctx_device(const keaki_hip_ctx* ctx) { return ctx ? ctx->device : -1; }

keaki_status keaki_hip_synchronize(keaki_hip_ctx* ctx) {
  CTX_GUARD(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  resolve_timing(ctx);
  return KEAKI_OK;
}

keaki_status keaki_hip_set_timing(keaki_hip_ctx* ctx, int32_t enabled) {
  CTX_GUARD(ctx);
  ctx->timing = enabled != 0;
  if (ctx->timing && !ctx->fk_ev[0])
    for (auto& e : ctx->fk_ev) (void)hipEventCreate(&e);
  return KEAKI_OK;
}
// device time of the last FK23 call (keaki_hip_open_fk[_poly]) with timing enabled, milliseconds: out3 = [the 2d pointwise scalar-mults,
// the butterfly stages of the two size-d transforms (k_g1_fft_stage_map + the twist), the whole device pipeline]; < 0 if none
keaki_status keaki_hip_last_fk_ms(keaki_hip_ctx* ctx, float* out3) {
  CTX_GUARD(ctx);
  if (!out3) return fail(ctx, KEAKI_ERR_BAD_ARG, "last_fk_ms: out3 is null");
  resolve_fk_timing(ctx);
  for (int i = 0; i < 3; i++) out3[i] = ctx->last_fk_ms[i];
  return KEAKI_OK;
}
float keaki_hip_last_msm_bucket_ms(const keaki_hip_ctx* ctx) { return ctx ? ctx->last_bucket_ms : -1.f; }
float keaki_hip_last_msm_total_ms(const keaki_hip_ctx* ctx) { return ctx ? ctx->last_total_ms : -1.f; }
int32_t keaki_hip_last_msm_window_bits(const keaki_hip_ctx* ctx) { return ctx ? ctx->last_c : 0; }

// ---- SRS -----------------------------------------------------------------------------------------
keaki_status keaki_hip_srs_g1_upload(keaki_hip_ctx* ctx, const uint64_t* points_aff, size_t n, keaki_hip_srs_g1** out) {
  CTX_GUARD(ctx);
  if (!out || (n && !points_aff)) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_upload: null pointer");
  void* d = nullptr;
  HIP_TRY(ctx, hipMalloc(&d, n ? n * G1_AFF_BYTES : 16));
  if (n) {
    // on the context's stream (non-blocking: not ordered behind the null stream a plain hipMemcpy uses), complete before the call returns
    hipError_t e = hipMemcpyAsync(d, points_aff, n * G1_AFF_BYTES, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d); return fail(ctx, KEAKI_ERR_HIP, "srs upload copy failed: %s", hipGetErrorString(e)); }
  }
  *out = new_srs<keaki_hip_srs_g1>(ctx, d, n, true);
  return KEAKI_OK;
}
keaki_status keaki_hip_srs_g1_wrap_dev(keaki_hip_ctx* ctx, const void* d_points_aff, size_t n, keaki_hip_srs_g1** out) {
  CTX_GUARD(ctx);
  if (!out || (n && !d_points_aff)) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_wrap_dev: null pointer");
  *out = new_srs<keaki_hip_srs_g1>(ctx, d_points_aff, n, false);
  return KEAKI_OK;
}
// non-owning view of points [offset, offset + n) of an uploaded SRS: the chunk a rank owns when an MSM is sharded by point range
keaki_status keaki_hip_srs_g1_slice(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, size_t offset, size_t n, keaki_hip_srs_g1** out) {
  CTX_GUARD(ctx);
  if (!srs || !out) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_slice: null pointer");
  if (offset > srs->n || n > srs->n - offset) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_slice: [%zu, %zu) is outside the %zu points of the SRS", offset, offset + n, srs->n);
  if (srs->device != ctx->device) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_slice: the SRS lives on device %d, this context on device %d", srs->device, ctx->device);
  *out = new_srs<keaki_hip_srs_g1>(ctx, (const char*)srs->d + offset * G1_AFF_BYTES, n, false);
  return KEAKI_OK;
}
size_t keaki_hip_srs_g1_len(const keaki_hip_srs_g1* srs) { return srs ? srs->n : 0; }
void keaki_hip_srs_g1_free(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs) {
  if (!srs) return;
  if (ctx) { keaki_internal::DeviceScope dc_(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
  keaki_internal::DeviceScope dev_(srs->device);
  if (srs->owned && srs->d) (void)hipFree((void*)srs->d);
  if (srs->table) (void)hipFree(srs->table);
  if (srs->fk_hat_s) (void)hipFree(srs->fk_hat_s);
  const size_t held = srs->table_bytes + srs->fk_bytes;          // booked on the context that built them, whichever context (or NULL) frees the handle
  with_live_ctx(srs->acct, [&](keaki_hip_ctx* a) { mem_sub(a->mem_tables, held); });
  delete srs;
}
keaki_status keaki_hip_srs_g1_precompute(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, size_t* table_bytes_out) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.srs_precompute");
  if (!srs) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_precompute: srs is null");
  SRS_CHECK(ctx, srs, "srs_g1_precompute");
  std::lock_guard<std::recursive_mutex> hl(srs->mu);       // contexts sharing the handle: the first one builds, the others find the tables
  if (!srs->table && srs->n) {
    int c = 0; size_t bytes = 0; void* t = nullptr;
    ST_TRY(msm_g1_precompute_run(ctx, srs->d, srs->n, &c, &bytes, &t));
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(t); return fail(ctx, KEAKI_ERR_HIP, "srs_g1_precompute: %s", hipGetErrorString(e)); }
    srs->c_table = c; srs->table_bytes = bytes; srs->table = t;       // published only when complete
    srs->acct = ctx; ctx->mem_tables += bytes;
  }
  if (table_bytes_out) *table_bytes_out = srs->table_bytes;
  return KEAKI_OK;
}
keaki_status keaki_hip_srs_g2_upload(keaki_hip_ctx* ctx, const uint64_t* points_aff, size_t n, keaki_hip_srs_g2** out) {
  CTX_GUARD(ctx);
  if (!out || (n && !points_aff)) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g2_upload: null pointer");
  void* d = nullptr;
  HIP_TRY(ctx, hipMalloc(&d, n ? n * G2_AFF_BYTES : 16));
  if (n) {
    hipError_t e = hipMemcpyAsync(d, points_aff, n * G2_AFF_BYTES, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d); return fail(ctx, KEAKI_ERR_HIP, "srs upload copy failed: %s", hipGetErrorString(e)); }
  }
  *out = new_srs<keaki_hip_srs_g2>(ctx, d, n, true);
  return KEAKI_OK;
}
keaki_status keaki_hip_srs_g2_wrap_dev(keaki_hip_ctx* ctx, const void* d_points_aff, size_t n, keaki_hip_srs_g2** out) {
  CTX_GUARD(ctx);
  if (!out || (n && !d_points_aff)) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g2_wrap_dev: null pointer");
  *out = new_srs<keaki_hip_srs_g2>(ctx, d_points_aff, n, false);
  return KEAKI_OK;
}
void keaki_hip_srs_g2_free(keaki_hip_ctx* ctx, keaki_hip_srs_g2* srs) {
  if (!srs) return;
  if (ctx) { keaki_internal::DeviceScope dc_(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
  keaki_internal::DeviceScope dev_(srs->device);
  if (srs->owned && srs->d) (void)hipFree((void*)srs->d);
  if (srs->table) (void)hipFree(srs->table);
  const size_t held = srs->table_bytes;
  with_live_ctx(srs->acct, [&](keaki_hip_ctx* a) { mem_sub(a->mem_tables, held); });
  delete srs;
}
keaki_status keaki_hip_srs_g2_precompute(keaki_hip_ctx* ctx, keaki_hip_srs_g2* srs, size_t* table_bytes_out) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.srs_precompute");
  if (!srs) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g2_precompute: srs is null");
  SRS_CHECK(ctx, srs, "srs_g2_precompute");
  std::lock_guard<std::recursive_mutex> hl(srs->mu);
  if (!srs->table && srs->n) {
    int c = 0; size_t bytes = 0; void* t = nullptr;
    ST_TRY(msm_g2_precompute_run(ctx, srs->d, srs->n, &c, &bytes, &t));
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(t); return fail(ctx, KEAKI_ERR_HIP, "srs_g2_precompute: %s", hipGetErrorString(e)); }
    srs->c_table = c; srs->table_bytes = bytes; srs->table = t;
    srs->acct = ctx; ctx->mem_tables += bytes;
  }
  if (table_bytes_out) *table_bytes_out = srs->table_bytes;
  return KEAKI_OK;
}

// ---- MSM -----------------------------------------------------------------------------------------
keaki_status keaki_hip_msm_g1_dev(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, const void* d_scalars, size_t n, void* d_out_jac) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.msm_g1");
  if (!srs) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm_g1: srs is null");
  SRS_CHECK(ctx, srs, "msm_g1");
  const auto tb = srs_tables(srs);
  return msm_g1_run(ctx, srs->d, srs->n, d_scalars, n, d_out_jac, tb.first, tb.second);
}
keaki_status keaki_hip_msm_g1(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, const uint64_t* scalars, size_t n, uint64_t* out_jac) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.msm_g1");
  if (!srs || !out_jac || (n && !scalars)) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm_g1: null pointer");
  SRS_CHECK(ctx, srs, "msm_g1");
  if (n > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "msm: %zu scalars but the SRS holds %zu points", n, srs->n);
  ST_TRY(reserve(ctx, ctx->io_b, 96));
  const auto tb = srs_tables(srs);
  ST_TRY(msm_from_host(ctx, scalars, n, [&](const MsmPipe* pipe) {
    return msm_g1_run(ctx, srs->d, srs->n, ctx->io_a.p, n, ctx->io_b.p, tb.first, tb.second, pipe);
  }));
  ST_TRY(download(ctx, out_jac, ctx->io_b.p, 96));
  resolve_timing(ctx);
  return KEAKI_OK;
}
keaki_status keaki_hip_msm_g2_dev(keaki_hip_ctx* ctx, const keaki_hip_srs_g2* srs, const void* d_scalars, size_t n, void* d_out_jac) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.msm_g2");
  if (!srs) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm_g2: srs is null");
  SRS_CHECK(ctx, srs, "msm_g2");
  const auto tb = srs_tables(srs);
  return msm_g2_run(ctx, srs->d, srs->n, d_scalars, n, d_out_jac, tb.first, tb.second);
}
keaki_status keaki_hip_msm_g2(keaki_hip_ctx* ctx, const keaki_hip_srs_g2* srs, const uint64_t* scalars, size_t n, uint64_t* out_jac) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.msm_g2");
  if (!srs || !out_jac || (n && !scalars)) return fail(ctx, KEAKI_ERR_BAD_ARG, "msm_g2: null pointer");
  SRS_CHECK(ctx, srs, "msm_g2");
  if (n > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "msm: %zu scalars but the SRS holds %zu points", n, srs->n);
  ST_TRY(reserve(ctx, ctx->io_b, 192));
  const auto tb = srs_tables(srs);
  ST_TRY(msm_from_host(ctx, scalars, n, [&](const MsmPipe* pipe) {
    return msm_g2_run(ctx, srs->d, srs->n, ctx->io_a.p, n, ctx->io_b.p, tb.first, tb.second, pipe);
  }));
  ST_TRY(download(ctx, out_jac, ctx->io_b.p, 192));
  resolve_timing(ctx);
  return KEAKI_OK;
}
keaki_status keaki_hip_g1_sum_dev(keaki_hip_ctx* ctx, const void* d_points_jac, size_t k, void* d_out_jac) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.g1_sum");
  if (!d_out_jac || (k && !d_points_jac)) return fail(ctx, KEAKI_ERR_BAD_ARG, "g1_sum: null pointer");
  return g1_sum_run(ctx, d_points_jac, k, d_out_jac);
}
keaki_status keaki_hip_g1_sum(keaki_hip_ctx* ctx, const uint64_t* points_jac, size_t k, uint64_t* out_jac) {
  CTX_GUARD(ctx);
  if (!out_jac || (k && !points_jac)) return fail(ctx, KEAKI_ERR_BAD_ARG, "g1_sum: null pointer");
  ST_TRY(upload(ctx, ctx->io_a, points_jac, k * 96));
  ST_TRY(reserve(ctx, ctx->io_b, 96));
  ST_TRY(g1_sum_run(ctx, ctx->io_a.p, k, ctx->io_b.p));
  return download(ctx, out_jac, ctx->io_b.p, 96);
}

// ---- batched scalar multiplication -------------------------------------------------------------------
keaki_status keaki_hip_g1_mul_batch_dev(keaki_hip_ctx* ctx, const void* d_points_aff, int32_t point_stride, const void* d_scalars, size_t n,
                                        void* d_out_aff) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.g1_mul_batch");
  if (n == 0) return KEAKI_OK;
  if (!d_points_aff || !d_scalars || !d_out_aff || (point_stride != 0 && point_stride != 1)) return fail(ctx, KEAKI_ERR_BAD_ARG, "g1_mul_batch: bad argument");
  return g1_mul_batch_run(ctx, d_points_aff, (int)point_stride, d_scalars, n, d_out_aff);
}
keaki_status keaki_hip_g2_mul_batch_dev(keaki_hip_ctx* ctx, const void* d_points_aff, int32_t point_stride, const void* d_scalars, size_t n,
                                        void* d_out_aff) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.g2_mul_batch");
  if (n == 0) return KEAKI_OK;
  if (!d_points_aff || !d_scalars || !d_out_aff || (point_stride != 0 && point_stride != 1)) return fail(ctx, KEAKI_ERR_BAD_ARG, "g2_mul_batch: bad argument");
  return g2_mul_batch_run(ctx, d_points_aff, (int)point_stride, d_scalars, n, d_out_aff);
}
keaki_status keaki_hip_g1_mul_batch(keaki_hip_ctx* ctx, const uint64_t* points_aff, int32_t point_stride, const uint64_t* scalars, size_t n,
                                    uint64_t* out_aff) {
  CTX_GUARD(ctx);                 // held across stage -> kernel -> download: io_a/io_b/io_c belong to this call until it returns
  if (n == 0) return KEAKI_OK;
  if (!points_aff || !scalars || !out_aff || (point_stride != 0 && point_stride != 1)) return fail(ctx, KEAKI_ERR_BAD_ARG, "g1_mul_batch: bad argument");
  ST_TRY(upload(ctx, ctx->io_a, points_aff, (point_stride ? n : 1) * 64));
  ST_TRY(upload(ctx, ctx->io_b, scalars, n * 32));
  ST_TRY(reserve(ctx, ctx->io_c, n * 64));
  ST_TRY(keaki_hip_g1_mul_batch_dev(ctx, ctx->io_a.p, point_stride, ctx->io_b.p, n, ctx->io_c.p));
  prefault_out(ctx, out_aff, n * 64);
  return download(ctx, out_aff, ctx->io_c.p, n * 64);
}
keaki_status keaki_hip_g2_mul_batch(keaki_hip_ctx* ctx, const uint64_t* points_aff, int32_t point_stride, const uint64_t* scalars, size_t n,
                                    uint64_t* out_aff) {
  CTX_GUARD(ctx);                 // held across stage -> kernel -> download: io_a/io_b/io_c belong to this call until it returns
  if (n == 0) return KEAKI_OK;
  if (!points_aff || !scalars || !out_aff || (point_stride != 0 && point_stride != 1)) return fail(ctx, KEAKI_ERR_BAD_ARG, "g2_mul_batch: bad argument");
  ST_TRY(upload(ctx, ctx->io_a, points_aff, (point_stride ? n : 1) * 128));
  ST_TRY(upload(ctx, ctx->io_b, scalars, n * 32));
  ST_TRY(reserve(ctx, ctx->io_c, n * 128));
  ST_TRY(keaki_hip_g2_mul_batch_dev(ctx, ctx->io_a.p, point_stride, ctx->io_b.p, n, ctx->io_c.p));
  prefault_out(ctx, out_aff, n * 128);
  return download(ctx, out_aff, ctx->io_c.p, n * 128);
}

// ---- pairing ---------------------------------------------------------------------------------------------
keaki_status keaki_hip_pairing_batch_dev(keaki_hip_ctx* ctx, const void* d_g1_aff, const void* d_g2_aff, int32_t g2_stride, size_t n,
                                         void* d_gt_out) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.pairing");
  if (n == 0) return KEAKI_OK;
  if (!d_g1_aff || !d_g2_aff || !d_gt_out || (g2_stride != 0 && g2_stride != 1)) return fail(ctx, KEAKI_ERR_BAD_ARG, "pairing_batch: bad argument");
  return pairing_run(ctx, d_g1_aff, d_g2_aff, (int)g2_stride, n, d_gt_out);
}
keaki_status keaki_hip_pairing_batch(keaki_hip_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, int32_t g2_stride, size_t n,
                                     uint8_t* gt_out) {
  CTX_GUARD(ctx);
  if (n == 0) return KEAKI_OK;
  if (!g1_aff || !g2_aff || !gt_out || (g2_stride != 0 && g2_stride != 1)) return fail(ctx, KEAKI_ERR_BAD_ARG, "pairing_batch: bad argument");
  ST_TRY(upload(ctx, ctx->io_a, g1_aff, n * 64));
  ST_TRY(upload(ctx, ctx->io_b, g2_aff, (g2_stride ? n : 1) * 128));
  ST_TRY(reserve(ctx, ctx->io_c, n * 384));
  ST_TRY(keaki_hip_pairing_batch_dev(ctx, ctx->io_a.p, ctx->io_b.p, g2_stride, n, ctx->io_c.p));
  prefault_out(ctx, gt_out, n * 384);
  return download(ctx, gt_out, ctx->io_c.p, n * 384);
}

// a second stream of the context for work that is bound by latency, not by the device (the table of a new commitment), and the means to
// send the launchers -- which all enqueue on ctx->stream -- there for a scope (the caller holds the context lock)
static keaki_status aux_ready(keaki_hip_ctx* ctx) {
  if (ctx->aux_stream) return KEAKI_OK;
  hipStream_t s = nullptr;
  HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (auto& e : ctx->aux_ev) if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  ctx->aux_stream = s;
  return KEAKI_OK;
}
struct StreamSwap {
  keaki_hip_ctx* ctx;
  hipStream_t saved;
  StreamSwap(keaki_hip_ctx* c, hipStream_t s) : ctx(c), saved(c->stream) { c->stream = s; }
  ~StreamSwap() { ctx->stream = saved; }
};
// signed-window table of e(P, g2) for a G1 point P in device memory: the powers of two e(P, g2)^(2^s) = e(P, 2^s g2) come from ONE pairing
// launch of P against the tabulated line sequences of the multiples 2^s g2 (the latency of one pairing; the tables -- 320 x 18 KB -- are built
// once per context: lane s doubles g2 s times, one k_g2_prepare workgroup per multiple). Until round 4 the chain ran on the
// G1 side (2^s P by 260 doublings one after the other: 1.4 ms with the device idle, in every call with a new commitment).
constexpr uint32_t GT_POWERS_MAX = 320;
static keaki_status g2pow_tables(keaki_hip_ctx* ctx) {
  if (ctx->g2pow_ready) return KEAKI_OK;
  ST_TRY(reserve(ctx, ctx->g2pow_lines, (size_t)GT_POWERS_MAX * g2_prepared_bytes()));
  ST_TRY(reserve(ctx, ctx->g2pow_pts, (size_t)(GT_POWERS_MAX + 1) * G2_AFF_BYTES));
  char* gen = (char*)ctx->g2pow_pts.p;
  char* pts = gen + G2_AFF_BYTES;
  ST_TRY(g2_generator_to(ctx, gen));
  ST_TRY(g2_pow2_multiples_run(ctx, gen, GT_POWERS_MAX, pts));
  ST_TRY(g2_prepare_run(ctx, pts, ctx->g2pow_lines.p, GT_POWERS_MAX));
  ctx->g2pow_ready = true;
  return KEAKI_OK;
}
static keaki_status gt_table_of(keaki_hip_ctx* ctx, const void* d_p_aff, void* d_table, uint32_t wb) {
  char* gb = (char*)ctx->gt_base.p;
  const uint32_t cnt = gt_table_powers(wb);
  if (cnt > GT_POWERS_MAX) return fail(ctx, KEAKI_ERR_BAD_ARG, "gt_table_of: %u powers", cnt);
  void* pows = gb + G1_AFF_BYTES + 320 * G1_AFF_BYTES;
  ST_TRY(g2pow_tables(ctx));
  ST_TRY(pairing_raw_fixed_run(ctx, d_p_aff, 0, cnt, ctx->g2pow_lines.p, g2_prepared_lines(), pows));
  return gt_table_run(ctx, pows, d_table, wb);
}

// ---- KEM composites ------------------------------------------------------------------------------------------
// `prep`: build what depends on the SETUP only (generator tables, the line sequence of g2, the table of [tau]_2, the GT table of e(g1, g2)) for
// batches of n items, and stop: keaki_hip_encap_prepare. Nothing per item, nothing per commitment.
// what a host-pointer entry point knows without asking the device: the two constants of the batch (no read-back, no stream synchronisation
// between the upload and the kernels), the size of the WHOLE batch this chunk belongs to, and whether it is its first chunk
struct EncapHost { const uint64_t* com; const uint64_t* tau; size_t n_batch; bool first; };
static keaki_status encap_impl(keaki_hip_ctx* ctx, bool prep, const void* d_com_aff, const void* d_tau_g2_aff, const void* d_points,
                               const void* d_values, const void* d_r, size_t n, void* d_ct_out_aff, void* d_gt_out, void* d_key_out,
                               size_t msg_len, bool xor_into = false, const EncapHost* host = nullptr) {
  // a chunk of a larger batch takes the decisions of the whole batch (table widths, the GT path) and counts as ONE call with its commitment
  const size_t n_policy = host ? host->n_batch : n;
  const bool first_of_batch = !host || host->first;
  if (!prep) ST_TRY(reserve(ctx, ctx->tmp_a, n * G1_AFF_BYTES));
  ST_TRY(reserve(ctx, ctx->tmp_c, G2_AFF_BYTES));
  void* gt = d_gt_out;
  if (!gt && !prep) { ST_TRY(reserve(ctx, ctx->tmp_b, n * 384)); gt = ctx->tmp_b.p; }
  // generator g2 in device memory for the pairing's second slot (src/kem.rs:30 pairs with E::G2Affine::generator())
  ST_TRY(g2_generator_to(ctx, ctx->tmp_c.p));
  // window widths of the fixed-base tables: 16 bits for bases that outlive a batch (generators: per context, [tau]_2: per setup),
  // 13 bits for the commitment's table (rebuilt per batch in the per-item-pairing path)
  constexpr uint32_t FB_WB_LONG = 16, FB_WB_BATCH = 13;
  const size_t FBL = fb_table_entries(FB_WB_LONG), FBS = fb_table_entries(FB_WB_BATCH);
  const bool use_tables = n_policy >= 256;
  if (use_tables && !ctx->fb_ready) {
    ST_TRY(reserve(ctx, ctx->fb_g1_gen, FBL * G1_AFF_BYTES + G1_AFF_BYTES));
    ST_TRY(reserve(ctx, ctx->fb_g2_gen, FBL * G2_AFF_BYTES));
    ST_TRY(reserve(ctx, ctx->fb_com, FBS * G1_AFF_BYTES));
    ST_TRY(reserve(ctx, ctx->fb_tau, FBL * G2_AFF_BYTES));
    void* g1pt = (char*)ctx->fb_g1_gen.p + FBL * G1_AFF_BYTES;   // scratch slot behind the table
    ST_TRY(g1_generator_to(ctx, g1pt));
    ST_TRY(g1_fb_table_run(ctx, g1pt, ctx->fb_g1_gen.p, FB_WB_LONG));
    ST_TRY(g2_fb_table_run(ctx, ctx->tmp_c.p, ctx->fb_g2_gen.p, FB_WB_LONG));
    ctx->fb_ready = true;
  }
  // the second pairing slot is the constant generator g2: its line sequence (ark-ec's G2Prepared) is built once per context
  if (!ctx->g2gen_lines_ready) {
    ST_TRY(reserve(ctx, ctx->g2gen_lines, g2_prepared_bytes()));
    ST_TRY(g2_prepare_run(ctx, ctx->tmp_c.p, ctx->g2gen_lines.p));
    ctx->g2gen_lines_ready = true;
  }
  // the two constants of the batch on the host (one read-back, one synchronisation; none when the caller's host copies came along)
  uint64_t tau_host[16], com_host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (host) {
    memcpy(tau_host, host->tau, 128);
    memcpy(com_host, host->com, 64);
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(tau_host, d_tau_g2_aff, 128, hipMemcpyDeviceToHost, ctx->stream));
    if (!prep) HIP_TRY(ctx, hipMemcpyAsync(com_host, d_com_aff, 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  // window widths of the GT tables. The constant B = e(g1, g2) is tabulated once per context: 20-bit windows (13 products per item, 2.6 GB;
  // KEAKI_GT_WB_B picks another width). A = e(C, g2) per commitment: 13 bits on first sight (20 products per item, 31.5 MB: one launch of the
  // twelve-lane pairing kernel over the tabulated multiples of g2 + the fills, 1.7 ms); when the SAME commitment comes back, its 16-bit table
  // (16 products, 201 MB) is filled from the powers of two still lying in gt_base (0.5 ms, no pairing).
  // Which calls take this path: ALL of them since the table of a new commitment costs 1.7 ms beside the ciphertext kernel (round 4; until then
  // batches of >= 65,536 items and callers that kept encrypting to one commitment, from the third consecutive call on): an item costs ~30 Fq12
  // products instead of two G1 ladders and a pairing, and a single `encapsulate` to a NEW commitment 2.0 ms instead of 4.2 (the G1 ladders alone
  // took 1.8). KEAKI_ENCAP_GT / option encap_gt = N keeps the per-item pairing path for batches below N items whose commitment has no table yet.
  constexpr uint32_t GT_WB_A_FIRST = 13, GT_WB_A_REPEAT = 16;
  const bool wbb_env = ctx->tune.gt_wb_b != 0;          // Tuning::gt_wb_b / encap_gt (the environment is read in keaki_hip_ctx_create only)
  const bool gt_env = ctx->tune.encap_gt >= 0;
  const size_t gt_threshold = gt_env ? (size_t)ctx->tune.encap_gt : (size_t)0;
  bool a_cached = false;
  if (!prep) {
    if (ctx->seen_com_runs && memcmp(com_host, ctx->seen_com, 64) == 0) {
      if (first_of_batch && ctx->seen_com_runs < 1000000) ctx->seen_com_runs++;
    } else {
      memcpy(ctx->seen_com, com_host, 64);
      ctx->seen_com_runs = 1;
    }
    a_cached = ctx->gt_b_ready && ctx->gt_a_valid && memcmp(com_host, ctx->gt_a_com, 64) == 0;
  }
  const bool use_gt = n_policy >= gt_threshold || (!prep && !gt_env && (a_cached || ctx->seen_com_runs >= 3));
  bool a_on_aux = false, b_factor_done = false;
  // the table of a new commitment is built on the aux stream; `gt_a_pending_aux` says that the context's stream has not been made to wait for
  // that build yet. It survives an early error return, so a later call that finds the table published (same commitment) or rewrites gt_base
  // orders itself behind the build first -- whatever happened in between.
  auto wait_aux = [&]() -> keaki_status {
    if (ctx->gt_a_pending_aux) {
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_ev[1], 0));
      ctx->gt_a_pending_aux = false;
    }
    return KEAKI_OK;
  };
  if (use_gt) {
    ST_TRY(wait_aux());                                   // a build an earlier (failed) call left unawaited
    // GT_i = A^(r_i) B^(-beta_i r_i) with A = e(C, g2), B = e(g1, g2) (see pairing.hip.h): no pairing per item
    ST_TRY(reserve(ctx, ctx->gt_base, G1_AFF_BYTES + 320 * (G1_AFF_BYTES + 384)));   // a point | (unused since round 4) | the powers' pairings
    char* gb = (char*)ctx->gt_base.p;
    // B: 20-bit windows for a context that runs large batches, 16-bit (201 MB) for one that only ever made small calls; widened once when a large batch comes
    const uint32_t wb_b_req = wbb_env ? (uint32_t)ctx->tune.gt_wb_b : (n_policy >= 65536 ? 20u : 16u);
    if (!ctx->gt_b_ready || (!wbb_env && wb_b_req > ctx->gt_b_wb && !ctx->gt_b_fallback)) {
      if (wb_b_req < 8 || wb_b_req > 22 || gt_table_powers(wb_b_req) > 320) return fail(ctx, KEAKI_ERR_BAD_ARG, "gt_wb_b = %u out of range", wb_b_req);
      ctx->gt_b_ready = false;
      ctx->gt_b_wb = wb_b_req;
      keaki_status st_b = reserve(ctx, ctx->gt_tab_b, gt_table_bytes(ctx->gt_b_wb));
      if (st_b == KEAKI_ERR_OOM && ctx->gt_b_wb > 16) {          // no room for the wide table: the 16-bit one is 201 MB
        (void)hipGetLastError();
        ctx->gt_b_wb = 16;
        ctx->gt_b_fallback = true;
        st_b = reserve(ctx, ctx->gt_tab_b, gt_table_bytes(ctx->gt_b_wb));
      }
      ST_TRY(st_b);
      ST_TRY(g1_generator_to(ctx, gb));
      ST_TRY(gt_table_of(ctx, gb, ctx->gt_tab_b.p, ctx->gt_b_wb));
      ctx->gt_b_ready = true;
      ctx->gt_a_valid = false;                    // gt_base now holds B's powers
    }
    // A depends on the commitment only: reuse the table while the caller keeps encrypting to the same commitment. A NEW commitment's table is
    // a latency-bound job (65 waves for 1.4 ms, then the fills): it goes to a stream of its own, IN FRONT of the ciphertext kernel below, which
    // fills the device for 0.56 ms at 2^16 items -- the two run side by side and the exponentiation waits for both.
    if (!prep && (!ctx->gt_a_valid || memcmp(com_host, ctx->gt_a_com, 64) != 0)) {
      ctx->gt_a_valid = false;
      ST_TRY(reserve(ctx, ctx->gt_tab_a, gt_table_bytes(GT_WB_A_REPEAT)));
      ST_TRY(aux_ready(ctx));
      HIP_TRY(ctx, hipEventRecord(ctx->aux_ev[0], ctx->stream));          // behind every earlier reader of the table and of gt_base
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_ev[0], 0));
      {
        StreamSwap on_aux(ctx, ctx->aux_stream);
        ST_TRY(gt_table_of(ctx, d_com_aff, ctx->gt_tab_a.p, GT_WB_A_FIRST));
      }
      HIP_TRY(ctx, hipEventRecord(ctx->aux_ev[1], ctx->aux_stream));
      ctx->gt_a_pending_aux = true;
      a_on_aux = true;
      if (n > 4096) {
        // the constant base's factor FIRST on the main stream: it fills every SIMD (two waves of 256 registers each) and must be out of the way
        // when the table's fill levels arrive; the ciphertext kernel behind it leaves half of each SIMD's registers to them
        ST_TRY(reserve(ctx, ctx->tmp_a, n * 384));
        ST_TRY(gt_encap_exp_run(ctx, nullptr, 0, ctx->gt_tab_b.p, ctx->gt_b_wb, d_values, d_r, n, nullptr, nullptr, ctx->tmp_a.p));
        b_factor_done = true;
      }
      memcpy(ctx->gt_a_com, com_host, 64);
      ctx->gt_a_wb = GT_WB_A_FIRST;
      ctx->gt_a_valid = true;
    } else if (!prep && first_of_batch && ctx->gt_a_wb != GT_WB_A_REPEAT) {
      // same commitment again IN A LATER CALL (the chunks of one host batch keep the table their first chunk found or built): the powers
      // A^(2^s), s < 260, of the first build cover the 256 the wider table needs
      ST_TRY(gt_table_run(ctx, gb + G1_AFF_BYTES + 320 * G1_AFF_BYTES, ctx->gt_tab_a.p, GT_WB_A_REPEAT));
      static_assert(GT_WB_A_REPEAT == 16 && GT_WB_A_FIRST == 13, "the power count of the first table must cover the second");
      ctx->gt_a_wb = GT_WB_A_REPEAT;
    }
  }
  // ciphertexts ct_i = r_i [tau]_2 - (r_i alpha_i) g2: two fixed-base sums. [tau]_2 belongs to the setup, not to the batch: its window table
  // is rebuilt only when the point changes. Batches of >= 256 items use (and build) the 16-bit tables; smaller ones use them when they are
  // there for this [tau]_2, else SMALL 8-bit tables (32 x 129 entries per base, 0.5 MB, built in the latency of one G2 scalar-mult):
  // 64 mixed additions per item instead of two 254-step ladders (a single `encapsulate` call: 20.5 -> 10 ms).
  {
    const bool big_has_tau = ctx->fb_ready && ctx->fb_tau_valid && memcmp(tau_host, ctx->fb_tau_pt, 128) == 0;
    if (use_tables || big_has_tau) {
      if (!big_has_tau) {
        ctx->fb_tau_valid = false;
        ST_TRY(g2_fb_table_run(ctx, d_tau_g2_aff, ctx->fb_tau.p, FB_WB_LONG));
        memcpy(ctx->fb_tau_pt, tau_host, 128);
        ctx->fb_tau_valid = true;
      }
      if (!prep) ST_TRY(encap_g2_fixed_run(ctx, ctx->fb_tau.p, FB_WB_LONG, ctx->fb_g2_gen.p, FB_WB_LONG, d_points, d_r, n, d_ct_out_aff, a_on_aux));
    } else {
      constexpr uint32_t FB_WB_SMALL = 8;
      const size_t FBX = fb_table_entries(FB_WB_SMALL);
      if (!ctx->fbs_ready) {
        ST_TRY(reserve(ctx, ctx->fbs_g2_gen, FBX * G2_AFF_BYTES));
        ST_TRY(reserve(ctx, ctx->fbs_tau, FBX * G2_AFF_BYTES));
        ST_TRY(g2_fb_table_run(ctx, ctx->tmp_c.p, ctx->fbs_g2_gen.p, FB_WB_SMALL));
        ctx->fbs_ready = true;
        ctx->fbs_tau_valid = false;
      }
      if (!ctx->fbs_tau_valid || memcmp(tau_host, ctx->fbs_tau_pt, 128) != 0) {
        ctx->fbs_tau_valid = false;
        ST_TRY(g2_fb_table_run(ctx, d_tau_g2_aff, ctx->fbs_tau.p, FB_WB_SMALL));
        memcpy(ctx->fbs_tau_pt, tau_host, 128);
        ctx->fbs_tau_valid = true;
      }
      if (!prep) ST_TRY(encap_g2_fixed_run(ctx, ctx->fbs_tau.p, FB_WB_SMALL, ctx->fbs_g2_gen.p, FB_WB_SMALL, d_points, d_r, n, d_ct_out_aff));
    }
  }
  if (prep) return KEAKI_OK;
  if (use_gt) {
    if (b_factor_done) {
      // the factor of the constant base ran while the commitment's table was on its way; the commitment's factor behind it
      ST_TRY(wait_aux());
      ST_TRY(gt_encap_exp_run(ctx, ctx->gt_tab_a.p, ctx->gt_a_wb, nullptr, 0, d_values, d_r, n, gt, ctx->tmp_a.p, nullptr));
    } else {
      ST_TRY(wait_aux());
      ST_TRY(gt_encap_exp_run(ctx, ctx->gt_tab_a.p, ctx->gt_a_wb, ctx->gt_tab_b.p, ctx->gt_b_wb, d_values, d_r, n, gt));
    }
  } else {
    // per-item pairing e(r_i (C - beta_i g1), g2) with the tabulated lines of g2
    if (use_tables) {
      ST_TRY(g1_fb_table_run(ctx, d_com_aff, ctx->fb_com.p, FB_WB_BATCH));
      ST_TRY(encap_g1_fixed_run(ctx, ctx->fb_com.p, FB_WB_BATCH, ctx->fb_g1_gen.p, FB_WB_LONG, d_values, d_r, n, ctx->tmp_a.p));
    } else {
      ST_TRY(encap_g1_run(ctx, d_com_aff, d_values, d_r, n, ctx->tmp_a.p));
    }
    ST_TRY(pairing_run(ctx, ctx->tmp_a.p, ctx->tmp_c.p, 0, n, gt, ctx->g2gen_lines.p));
  }
  if (d_key_out && msg_len) ST_TRY(blake3_gt_run(ctx, gt, n, d_key_out, msg_len, xor_into));
  return KEAKI_OK;
}
keaki_status keaki_hip_encap_batch_dev(keaki_hip_ctx* ctx, const void* d_com_aff, const void* d_tau_g2_aff, const void* d_points,
                                       const void* d_values, const void* d_r, size_t n, void* d_ct_out_aff, void* d_gt_out, void* d_key_out,
                                       size_t msg_len) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.encap");
  if (n == 0) return KEAKI_OK;
  if (!d_com_aff || !d_tau_g2_aff || !d_points || !d_values || !d_r || !d_ct_out_aff || (!d_gt_out && !d_key_out) || msg_len > 65536)
    return fail(ctx, KEAKI_ERR_BAD_ARG, "encap_batch: bad argument");
  return encap_impl(ctx, false, d_com_aff, d_tau_g2_aff, d_points, d_values, d_r, n, d_ct_out_aff, d_gt_out, d_key_out, msg_len);
}
// Setup-time: everything of encap_batch that depends on the setup only, for batches of `batch_hint` items (the KEM analogue of
// keaki_hip_srs_g1_precompute): fixed-base window tables of g1, g2 and [tau]_2, the line sequence of g2, and -- for hints >= 65,536 (or the
// threshold of option "encap_gt") -- the GT table of e(g1, g2) (2.6 GB; the 16-bit one when that does not fit). ~60 ms that the first
// large encap_batch of a context would otherwise pay. Results never depend on it.
keaki_status keaki_hip_encap_prepare(keaki_hip_ctx* ctx, const uint64_t* tau_g2_aff, size_t batch_hint) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.encap_prepare");
  if (!tau_g2_aff) return fail(ctx, KEAKI_ERR_BAD_ARG, "encap_prepare: null pointer");
  if (batch_hint == 0) return KEAKI_OK;
  ST_TRY(upload(ctx, ctx->io_e, tau_g2_aff, 128));
  ST_TRY(encap_impl(ctx, true, nullptr, ctx->io_e.p, nullptr, nullptr, nullptr, batch_hint, nullptr, nullptr, nullptr, 0));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return KEAKI_OK;
}
keaki_status keaki_hip_decap_batch_dev(keaki_hip_ctx* ctx, const void* d_proofs_aff, const void* d_cts_aff, size_t n, void* d_gt_out,
                                       void* d_key_out, size_t msg_len) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.decap");
  if (n == 0) return KEAKI_OK;
  if (!d_proofs_aff || !d_cts_aff || (!d_gt_out && !d_key_out) || msg_len > 65536) return fail(ctx, KEAKI_ERR_BAD_ARG, "decap_batch: bad argument");
  void* gt = d_gt_out;
  if (!gt) { ST_TRY(reserve(ctx, ctx->tmp_b, n * 384)); gt = ctx->tmp_b.p; }
  ST_TRY(pairing_run(ctx, d_proofs_aff, d_cts_aff, 1, n, gt));
  if (d_key_out && msg_len) ST_TRY(blake3_gt_run(ctx, gt, n, d_key_out, msg_len));
  return KEAKI_OK;
}
keaki_status keaki_hip_encap_batch(keaki_hip_ctx* ctx, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* points,
                                   const uint64_t* values, const uint64_t* r, size_t n, uint64_t* ct_out_aff, uint8_t* gt_out, uint8_t* key_out,
                                   size_t msg_len) {
  CTX_GUARD(ctx);                 // one lock from staging to the last download
  TRACE_SCOPE("keaki.encap");
  if (n == 0) return KEAKI_OK;
  if (!com_aff || !tau_g2_aff || !points || !values || !r || !ct_out_aff || (!gt_out && !key_out) || msg_len > 65536)
    return fail(ctx, KEAKI_ERR_BAD_ARG, "encap_batch: bad argument");
  const size_t ch = pipe_chunk_items(ctx, n);
  const size_t off_pts = 0, off_val = off_pts + ch * 32, off_r = off_val + ch * 32, off_ct = off_r + ch * 32, off_gt = off_ct + ch * 128,
               off_key = off_gt + ch * 384, half = (off_key + ch * msg_len + 255) & ~(size_t)255;
  ST_TRY(reserve(ctx, ctx->io_a, 256 + 2 * half));
  char* base = (char*)ctx->io_a.p;
  HIP_TRY(ctx, hipMemcpyAsync(base, com_aff, 64, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(base + 64, tau_g2_aff, 128, hipMemcpyHostToDevice, ctx->stream));
  const bool want_key = key_out && msg_len;
  return pipelined(ctx, n, ch, 128 + (gt_out ? 384 : 0) + (want_key ? msg_len : 0),
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + 256 + h * half;
      HIP_TRY(ctx, hipMemcpyAsync(b + off_pts, points + 4 * lo, m * 32, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_val, values + 4 * lo, m * 32, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_r, r + 4 * lo, m * 32, hipMemcpyHostToDevice, cs));
      return KEAKI_OK;
    },
    [&](size_t lo, size_t m, int h) -> keaki_status {
      char* b = base + 256 + h * half;
      const EncapHost eh = {com_aff, tau_g2_aff, n, lo == 0};
      return encap_impl(ctx, false, base, base + 64, b + off_pts, b + off_val, b + off_r, m, b + off_ct, b + off_gt, want_key ? b + off_key : nullptr,
                        msg_len, false, &eh);
    },
    [&](size_t lo, size_t m) {
      prefault_out(ctx, ct_out_aff + 16 * lo, m * 128);
      if (gt_out) prefault_out(ctx, gt_out + 384 * lo, m * 384);
      if (want_key) prefault_out(ctx, key_out + msg_len * lo, m * msg_len);
    },
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + 256 + h * half;
      HIP_TRY(ctx, hipMemcpyAsync(ct_out_aff + 16 * lo, b + off_ct, m * 128, hipMemcpyDeviceToHost, cs));
      if (gt_out) HIP_TRY(ctx, hipMemcpyAsync(gt_out + 384 * lo, b + off_gt, m * 384, hipMemcpyDeviceToHost, cs));
      if (want_key) HIP_TRY(ctx, hipMemcpyAsync(key_out + msg_len * lo, b + off_key, m * msg_len, hipMemcpyDeviceToHost, cs));
      return KEAKI_OK;
    });
}
keaki_status keaki_hip_decap_batch(keaki_hip_ctx* ctx, const uint64_t* proofs_aff, const uint64_t* cts_aff, size_t n, uint8_t* gt_out,
                                   uint8_t* key_out, size_t msg_len) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.decap");
  if (n == 0) return KEAKI_OK;
  if (!proofs_aff || !cts_aff || (!gt_out && !key_out) || msg_len > 65536) return fail(ctx, KEAKI_ERR_BAD_ARG, "decap_batch: bad argument");
  const size_t ch = pipe_chunk_items(ctx, n, pairing_launch_items());
  const size_t off_ct = ch * 64, off_gt = off_ct + ch * 128, off_key = off_gt + ch * 384, half = (off_key + ch * msg_len + 255) & ~(size_t)255;
  ST_TRY(reserve(ctx, ctx->io_a, 2 * half));
  char* base = (char*)ctx->io_a.p;
  const bool want_key = key_out && msg_len;
  return pipelined(ctx, n, ch, (gt_out ? 384 : 0) + (want_key ? msg_len : 0),
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + h * half;
      HIP_TRY(ctx, hipMemcpyAsync(b, proofs_aff + 8 * lo, m * 64, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_ct, cts_aff + 16 * lo, m * 128, hipMemcpyHostToDevice, cs));
      return KEAKI_OK;
    },
    [&](size_t, size_t m, int h) -> keaki_status {
      char* b = base + h * half;
      ST_TRY(pairing_run(ctx, b, b + off_ct, 1, m, b + off_gt));
      if (want_key) ST_TRY(blake3_gt_run(ctx, b + off_gt, m, b + off_key, msg_len));
      return KEAKI_OK;
    },
    [&](size_t lo, size_t m) {
      if (gt_out) prefault_out(ctx, gt_out + 384 * lo, m * 384);
      if (want_key) prefault_out(ctx, key_out + msg_len * lo, m * msg_len);
    },
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + h * half;
      if (gt_out) HIP_TRY(ctx, hipMemcpyAsync(gt_out + 384 * lo, b + off_gt, m * 384, hipMemcpyDeviceToHost, cs));
      if (want_key) HIP_TRY(ctx, hipMemcpyAsync(key_out + msg_len * lo, b + off_key, m * msg_len, hipMemcpyDeviceToHost, cs));
      return KEAKI_OK;
    });
}

// ---- enc::encrypt / enc::decrypt over a batch (src/enc.rs:19-55 inside the loops of src/vec.rs:63-66, :75-78): KEM + the XOR DEM on the device ----
// d_body_inout: n x msg_len bytes, the messages on entry and the ciphertext bodies on exit (decrypt: the other way round). The key is
// XORed in by the KDF kernel itself; neither GT bytes nor keys exist outside the device.
keaki_status keaki_hip_encrypt_batch_dev(keaki_hip_ctx* ctx, const void* d_com_aff, const void* d_tau_g2_aff, const void* d_points,
                                         const void* d_values, const void* d_r, size_t n, void* d_ct_out_aff, void* d_body_inout, size_t msg_len) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.encrypt");
  if (n == 0) return KEAKI_OK;
  if (!d_com_aff || !d_tau_g2_aff || !d_points || !d_values || !d_r || !d_ct_out_aff || !d_body_inout || msg_len == 0 || msg_len > 65536)
    return fail(ctx, KEAKI_ERR_BAD_ARG, "encrypt_batch: bad argument");
  return encap_impl(ctx, false, d_com_aff, d_tau_g2_aff, d_points, d_values, d_r, n, d_ct_out_aff, nullptr, d_body_inout, msg_len, true);
}
keaki_status keaki_hip_decrypt_batch_dev(keaki_hip_ctx* ctx, const void* d_proofs_aff, const void* d_cts_aff, size_t n, void* d_body_inout, size_t msg_len) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.decrypt");
  if (n == 0) return KEAKI_OK;
  if (!d_proofs_aff || !d_cts_aff || !d_body_inout || msg_len == 0 || msg_len > 65536) return fail(ctx, KEAKI_ERR_BAD_ARG, "decrypt_batch: bad argument");
  ST_TRY(reserve(ctx, ctx->tmp_b, n * 384));
  ST_TRY(pairing_run(ctx, d_proofs_aff, d_cts_aff, 1, n, ctx->tmp_b.p));
  return blake3_gt_run(ctx, ctx->tmp_b.p, n, d_body_inout, msg_len, true);
}
keaki_status keaki_hip_encrypt_batch(keaki_hip_ctx* ctx, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* points,
                                     const uint64_t* values, const uint64_t* r, const uint8_t* msgs, size_t n, uint64_t* ct_out_aff, uint8_t* body_out,
                                     size_t msg_len) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.encrypt");
  if (n == 0) return KEAKI_OK;
  if (!com_aff || !tau_g2_aff || !points || !values || !r || !msgs || !ct_out_aff || !body_out || msg_len == 0 || msg_len > 65536)
    return fail(ctx, KEAKI_ERR_BAD_ARG, "encrypt_batch: bad argument");
  const size_t ch = pipe_chunk_items(ctx, n);
  const size_t off_pts = 0, off_val = off_pts + ch * 32, off_r = off_val + ch * 32, off_ct = off_r + ch * 32, off_body = off_ct + ch * 128,
               half = (off_body + ch * msg_len + 255) & ~(size_t)255;
  ST_TRY(reserve(ctx, ctx->io_a, 256 + 2 * half));
  char* base = (char*)ctx->io_a.p;
  HIP_TRY(ctx, hipMemcpyAsync(base, com_aff, 64, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(base + 64, tau_g2_aff, 128, hipMemcpyHostToDevice, ctx->stream));
  return pipelined(ctx, n, ch, 128 + msg_len,
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + 256 + h * half;
      HIP_TRY(ctx, hipMemcpyAsync(b + off_pts, points + 4 * lo, m * 32, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_val, values + 4 * lo, m * 32, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_r, r + 4 * lo, m * 32, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_body, msgs + msg_len * lo, m * msg_len, hipMemcpyHostToDevice, cs));
      return KEAKI_OK;
    },
    [&](size_t lo, size_t m, int h) -> keaki_status {
      char* b = base + 256 + h * half;
      const EncapHost eh = {com_aff, tau_g2_aff, n, lo == 0};
      return encap_impl(ctx, false, base, base + 64, b + off_pts, b + off_val, b + off_r, m, b + off_ct, nullptr, b + off_body, msg_len, true, &eh);
    },
    [&](size_t lo, size_t m) {
      prefault_out(ctx, ct_out_aff + 16 * lo, m * 128);
      prefault_out(ctx, body_out + msg_len * lo, m * msg_len);
    },
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + 256 + h * half;
      HIP_TRY(ctx, hipMemcpyAsync(ct_out_aff + 16 * lo, b + off_ct, m * 128, hipMemcpyDeviceToHost, cs));
      HIP_TRY(ctx, hipMemcpyAsync(body_out + msg_len * lo, b + off_body, m * msg_len, hipMemcpyDeviceToHost, cs));
      return KEAKI_OK;
    });
}
keaki_status keaki_hip_decrypt_batch(keaki_hip_ctx* ctx, const uint64_t* proofs_aff, const uint64_t* cts_aff, const uint8_t* bodies, size_t n,
                                     uint8_t* msgs_out, size_t msg_len) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.decrypt");
  if (n == 0) return KEAKI_OK;
  if (!proofs_aff || !cts_aff || !bodies || !msgs_out || msg_len == 0 || msg_len > 65536) return fail(ctx, KEAKI_ERR_BAD_ARG, "decrypt_batch: bad argument");
  const size_t ch = pipe_chunk_items(ctx, n, pairing_launch_items());
  const size_t off_ct = ch * 64, off_body = off_ct + ch * 128, half = (off_body + ch * msg_len + 255) & ~(size_t)255;
  ST_TRY(reserve(ctx, ctx->io_a, 2 * half));
  ST_TRY(reserve(ctx, ctx->tmp_b, ch * 384));
  char* base = (char*)ctx->io_a.p;
  return pipelined(ctx, n, ch, msg_len,
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + h * half;
      HIP_TRY(ctx, hipMemcpyAsync(b, proofs_aff + 8 * lo, m * 64, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_ct, cts_aff + 16 * lo, m * 128, hipMemcpyHostToDevice, cs));
      HIP_TRY(ctx, hipMemcpyAsync(b + off_body, bodies + msg_len * lo, m * msg_len, hipMemcpyHostToDevice, cs));
      return KEAKI_OK;
    },
    [&](size_t, size_t m, int h) -> keaki_status {
      char* b = base + h * half;
      ST_TRY(pairing_run(ctx, b, b + off_ct, 1, m, ctx->tmp_b.p));
      return blake3_gt_run(ctx, ctx->tmp_b.p, m, b + off_body, msg_len, true);
    },
    [&](size_t lo, size_t m) { prefault_out(ctx, msgs_out + msg_len * lo, m * msg_len); },
    [&](size_t lo, size_t m, int h, hipStream_t cs) -> keaki_status {
      char* b = base + h * half;
      HIP_TRY(ctx, hipMemcpyAsync(msgs_out + msg_len * lo, b + off_body, m * msg_len, hipMemcpyDeviceToHost, cs));
      return KEAKI_OK;
    });
}

// ---- FK23 batch openings: replaces kzg::open_fk (src/kzg.rs:157-203) ----------------------------------------------------------
keaki_status keaki_hip_open_fk(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, uint32_t log2d, const uint64_t* hat_a, const uint64_t* tw_2d,
                               const uint64_t* tw_2d_inv, const uint64_t* tw_d, uint64_t* proofs_out_aff) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.open_fk");
  if (!srs || !hat_a || !tw_2d || !tw_2d_inv || !proofs_out_aff || log2d > 27) return fail(ctx, KEAKI_ERR_BAD_ARG, "open_fk: bad argument");
  SRS_CHECK(ctx, srs, "open_fk");
  std::lock_guard<std::recursive_mutex> hl(srs->mu);        // the cached transform hat_s belongs to the handle: one FK23 call per handle at a time
  const size_t d = (size_t)1 << log2d;
  if (d > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "open_fk: %zu coefficients but the SRS holds %zu points", d, srs->n);
  // one staging buffer: hat_a (2d Fr) | tw_2d (d) | tw_2d_inv (d) | tw_d (d/2) | work (2d Jacobian) | proofs (d affine)
  const size_t o_ha = 0, o_t1 = o_ha + 2 * d * 32, o_t2 = o_t1 + d * 32, o_t3 = o_t2 + d * 32, o_w = o_t3 + (d / 2 + 1) * 32, o_p = o_w + 2 * d * 96,
               total = o_p + d * 64;
  ST_TRY(reserve(ctx, ctx->io_d, total));
  char* b = (char*)ctx->io_d.p;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(b + o_ha, hat_a, 2 * d * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(b + o_t1, tw_2d, d * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(b + o_t2, tw_2d_inv, d * 32, hipMemcpyHostToDevice, st));
  (void)tw_d;                       // the size-d transforms take every second entry of the 2d tables
  if (srs->fk_log2d != (int)log2d) {
    if (srs->fk_hat_s) { HIP_TRY(ctx, hipStreamSynchronize(st)); (void)hipFree(srs->fk_hat_s); srs->fk_hat_s = nullptr; srs->fk_log2d = -1; }
    HIP_TRY(ctx, hipMalloc(&srs->fk_hat_s, 2 * d * 96));
    ST_TRY(fk_hat_s_run(ctx, srs->d, log2d, b + o_t1, srs->fk_hat_s));
    srs->fk_log2d = (int)log2d;
    fk_account(ctx, srs);
  }
  ST_TRY(open_fk_run(ctx, srs->fk_hat_s, log2d, b + o_ha, b + o_t1, b + o_t2, b + o_w, b + o_p));
  prefault_out(ctx, proofs_out_aff, d * 64);
  return download(ctx, proofs_out_aff, b + o_p, d * 64);
}

// FK23 from the coefficients: twiddles and hat_a are derived on the device (row f-4)
keaki_status keaki_hip_open_fk_poly(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, uint32_t log2d, const uint64_t* coeffs, const uint64_t* omega_2d,
                                    const uint64_t* omega_2d_inv, const uint64_t* inv_2d, uint64_t* proofs_out_aff) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.open_fk");
  if (!srs || !coeffs || !omega_2d || !omega_2d_inv || !inv_2d || !proofs_out_aff || log2d > 27) return fail(ctx, KEAKI_ERR_BAD_ARG, "open_fk_poly: bad argument");
  SRS_CHECK(ctx, srs, "open_fk_poly");
  std::lock_guard<std::recursive_mutex> hl(srs->mu);
  const size_t d = (size_t)1 << log2d;
  if (d > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "open_fk: %zu coefficients but the SRS holds %zu points", d, srs->n);
  const size_t o_p = 0, o_fr = o_p + d * 32, o_g = o_fr + (4 * d + d / 2 + 2) * 32, o_out = o_g + 2 * d * 96, total = o_out + d * 64;
  ST_TRY(reserve(ctx, ctx->io_d, total));
  char* b = (char*)ctx->io_d.p;
  HIP_TRY(ctx, hipMemcpyAsync(b + o_p, coeffs, d * 32, hipMemcpyHostToDevice, ctx->stream));
  const keaki_status st_fk = open_fk_poly_run(ctx, srs->d, &srs->fk_hat_s, &srs->fk_log2d, log2d, b + o_p, omega_2d, omega_2d_inv, inv_2d, b + o_fr, b + o_g, b + o_out);
  fk_account(ctx, srs);
  ST_TRY(st_fk);
  prefault_out(ctx, proofs_out_aff, d * 64);
  return download(ctx, proofs_out_aff, b + o_out, d * 64);
}
// hat_s = DFT_2d(reversed SRS) for later open_fk calls with this d: setup-time work (the FK23 analogue of keaki_hip_srs_g1_precompute)
keaki_status keaki_hip_srs_g1_precompute_fk(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, uint32_t log2d, const uint64_t* omega_2d) {
  CTX_GUARD(ctx);
  if (!srs || !omega_2d || log2d > 27) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_precompute_fk: bad argument");
  SRS_CHECK(ctx, srs, "srs_g1_precompute_fk");
  std::lock_guard<std::recursive_mutex> hl(srs->mu);
  const size_t d = (size_t)1 << log2d;
  if (d > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "open_fk: %zu coefficients but the SRS holds %zu points", d, srs->n);
  ST_TRY(reserve(ctx, ctx->io_d, d * 32));
  const keaki_status st_fk = fk_precompute_run(ctx, srs->d, &srs->fk_hat_s, &srs->fk_log2d, log2d, omega_2d, ctx->io_d.p);
  fk_account(ctx, srs);
  ST_TRY(st_fk);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // setup-time call: return when the table exists
  return KEAKI_OK;
}
// In-place scalar-field DFT of n = 2^log2n elements with the order-n root `omega`, then an optional scaling (the 1/n of an inverse transform).
keaki_status keaki_hip_fr_fft(keaki_hip_ctx* ctx, uint64_t* data, uint32_t log2n, const uint64_t* omega, const uint64_t* scale_or_null) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.fr_fft");
  if (!data || !omega || log2n > 28) return fail(ctx, KEAKI_ERR_BAD_ARG, "fr_fft: bad argument");
  const size_t n = (size_t)1 << log2n;
  ST_TRY(reserve(ctx, ctx->io_d, n * 32 + (n / 2 + 1) * 32));
  char* b = (char*)ctx->io_d.p;
  HIP_TRY(ctx, hipMemcpyAsync(b, data, n * 32, hipMemcpyHostToDevice, ctx->stream));
  ST_TRY(fr_fft_run(ctx, b, log2n, omega, scale_or_null, b + n * 32));
  return download(ctx, data, b, n * 32);
}

// ---- vec_commit in one call (src/vec.rs:22-49 behind the padding draw): iFFT -> FK23 openings -> commit, coefficients never leave the device ----
keaki_status keaki_hip_vec_commit(keaki_hip_ctx* ctx, keaki_hip_srs_g1* srs, const uint64_t* values, size_t n, const uint64_t* pad, uint32_t log2d,
                                  const uint64_t* omega_d_inv, const uint64_t* inv_d, const uint64_t* omega_2d, const uint64_t* omega_2d_inv,
                                  const uint64_t* inv_2d, uint64_t* com_out_jac, uint64_t* proofs_out_aff) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.vec_commit");
  if (!srs || (n && !values) || !omega_d_inv || !inv_d || !omega_2d || !omega_2d_inv || !inv_2d || !com_out_jac || !proofs_out_aff || log2d > 27)
    return fail(ctx, KEAKI_ERR_BAD_ARG, "vec_commit: bad argument");
  SRS_CHECK(ctx, srs, "vec_commit");
  const size_t d = (size_t)1 << log2d, m = n + (pad ? 1 : 0);
  if (m > d) return fail(ctx, KEAKI_ERR_BAD_ARG, "vec_commit: %zu evaluations do not fit the domain of %zu", m, d);
  if (d > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "open_fk: %zu coefficients but the SRS holds %zu points", d, srs->n);
  std::lock_guard<std::recursive_mutex> hl(srs->mu);
  // one staging block: coefficients (d Fr) | twiddles of the iFFT (d/2 + 1) | FK23 scalar work | FK23 point work (2d Jacobian) | proofs (d affine) | commitment
  const size_t o_c = 0, o_tw = o_c + d * 32, o_fr = o_tw + (d / 2 + 1) * 32, o_g = o_fr + (4 * d + d / 2 + 2) * 32, o_out = o_g + 2 * d * 96,
               o_com = o_out + d * 64, total = o_com + 96;
  ST_TRY(reserve(ctx, ctx->io_d, total));
  char* b = (char*)ctx->io_d.p;
  hipStream_t st = ctx->stream;
  if (m < d) HIP_TRY(ctx, hipMemsetAsync(b + o_c + m * 32, 0, (d - m) * 32, st));          // evaluations beyond the padded vector are zero (ark-poly's ifft resizes)
  if (n) HIP_TRY(ctx, hipMemcpyAsync(b + o_c, values, n * 32, hipMemcpyHostToDevice, st));
  if (pad) HIP_TRY(ctx, hipMemcpyAsync(b + o_c + n * 32, pad, 32, hipMemcpyHostToDevice, st));
  ST_TRY(fr_fft_run(ctx, b + o_c, log2d, omega_d_inv, inv_d, b + o_tw));                  // domain.ifft (src/vec.rs:37)
  const keaki_status st_fk = open_fk_poly_run(ctx, srs->d, &srs->fk_hat_s, &srs->fk_log2d, log2d, b + o_c, omega_2d, omega_2d_inv, inv_2d, b + o_fr, b + o_g, b + o_out);   // :40
  fk_account(ctx, srs);
  ST_TRY(st_fk);
  const auto tb = srs_tables(srs);
  ST_TRY(msm_g1_run(ctx, srs->d, srs->n, b + o_c, d, b + o_com, tb.first, tb.second));     // :46 (trailing zero coefficients contribute nothing)
  prefault_out(ctx, proofs_out_aff, d * 64);
  HIP_TRY(ctx, hipMemcpyAsync(com_out_jac, b + o_com, 96, hipMemcpyDeviceToHost, st));
  ST_TRY(download(ctx, proofs_out_aff, b + o_out, d * 64));
  resolve_timing(ctx);
  return KEAKI_OK;
}

// ---- FK23 sharded over `world` = 2^k ranks: one handle per rank, the caller runs the exchanges between the steps -----------------
struct keaki_hip_fk_shard {
  FkShard plan;
  const keaki_hip_srs_g1* srs = nullptr;      // must outlive the handle
  uint32_t world = 0;
  int setup_next = 0, open_next = 0;          // the step each sequence expects next (a skipped exchange cannot be detected, a skipped step can)
};
namespace {
size_t fk_shard_buffer_bytes(const keaki_hip_fk_shard* fk) {
  const size_t d = (size_t)1 << fk->plan.log2d;
  return std::max(2 * d / fk->world * 96, d * 64);
}
void fk_shard_release(keaki_hip_fk_shard* fk) {
  for (void** p : {&fk->plan.tw, &fk->plan.twi, &fk->plan.hat_a, &fk->plan.coeffs, &fk->plan.hat_s, &fk->plan.work, &fk->plan.e})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
}
}  // namespace
keaki_status keaki_hip_fk_shard_create(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, uint32_t log2d, uint32_t rank, uint32_t world,
                                       const uint64_t* omega_2d, const uint64_t* omega_2d_inv, const uint64_t* inv_2d, keaki_hip_fk_shard** out) {
  CTX_GUARD(ctx);
  if (!srs || !omega_2d || !omega_2d_inv || !inv_2d || !out || log2d > 27) return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_create: bad argument");
  *out = nullptr;
  if (world < 2 || (world & (world - 1)) || rank >= world) return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_create: world = %u must be a power of two >= 2, rank %u below it", world, rank);
  const size_t d = (size_t)1 << log2d;
  if (d < (size_t)world * world)
    return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_create: %zu openings are too few to shard over %u ranks (needs world^2); use keaki_hip_open_fk_poly", d, world);
  if (d > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "open_fk: %zu coefficients but the SRS holds %zu points", d, srs->n);
  SRS_CHECK(ctx, srs, "fk_shard_create");
  auto* fk = new keaki_hip_fk_shard();
  fk->srs = srs;
  fk->world = world;
  fk->plan.log2d = log2d;
  fk->plan.rank = rank;
  while ((1u << fk->plan.rho) < world) fk->plan.rho++;
  memcpy(fk->plan.omega, omega_2d, 32); memcpy(fk->plan.omega_inv, omega_2d_inv, 32); memcpy(fk->plan.inv_2d, inv_2d, 32);
  const size_t m = 2 * d / world;
  keaki_status st = KEAKI_OK;
  const std::pair<void**, size_t> want[] = {{&fk->plan.tw, d * 32}, {&fk->plan.twi, d * 32}, {&fk->plan.hat_a, 2 * d * 32}, {&fk->plan.coeffs, d * 32},
                                            {&fk->plan.hat_s, m * 96}, {&fk->plan.work, m * 96}, {&fk->plan.e, m / 2 * 96}};
  for (auto& w : want)
    if (st == KEAKI_OK) st = dev_alloc(ctx, w.first, w.second);
  if (st != KEAKI_OK) { fk_shard_release(fk); delete fk; return st; }
  *out = fk;
  return KEAKI_OK;
}
void keaki_hip_fk_shard_free(keaki_hip_ctx* ctx, keaki_hip_fk_shard* fk) {
  if (!fk) return;
  keaki_internal::DeviceScope dev_(ctx ? ctx->device : fk->srs ? fk->srs->device : -1);
  if (ctx) {
    std::lock_guard<std::recursive_mutex> lock_(ctx->mu);
    (void)hipStreamSynchronize(ctx->stream);
  } else if (fk->srs && fk->srs->device >= 0) {
    (void)hipDeviceSynchronize();
  }
  fk_shard_release(fk);             // hipFree needs no context: the device buffers go even when the caller's ctx is already gone
  delete fk;
}
keaki_status keaki_hip_fk_shard_sizes(const keaki_hip_fk_shard* fk, size_t* out4) {
  if (!fk || !out4) return KEAKI_ERR_BAD_ARG;
  const size_t d = (size_t)1 << fk->plan.log2d, R = fk->world;
  out4[0] = fk_shard_buffer_bytes(fk);
  out4[1] = 2 * d / R / R * 96;      // the setup's all-to-all (two d-point transforms in one exchange): bytes per peer
  out4[2] = d / R / R * 96;          // the two all-to-alls of a call (one d-point transform each)
  out4[3] = d / R * 64;              // all-gather of the affine proofs: bytes per rank
  return KEAKI_OK;
}
keaki_status keaki_hip_fk_shard_setup(keaki_hip_ctx* ctx, keaki_hip_fk_shard* fk, int32_t step, void* d_send, void* d_recv) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.fk_shard_setup");
  if (!fk || step < 0 || step > 1 || (step == 0 && !d_send) || (step == 1 && !d_recv)) return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_setup: bad argument");
  // step 0 may always start over (the caller's exchange failed after it, say): it recomputes this rank's outgoing points from the SRS
  if (step != fk->setup_next && step != 0) return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_setup: step %d out of order (step %d is next)", step, fk->setup_next);
  if (step == 0) { fk->plan.hat_s_ready = false; fk->setup_next = 0; }
  ST_TRY(fk_shard_setup_run(ctx, fk->plan, fk->srs->d, step, d_send, d_recv));
  fk->setup_next = step + 1;
  return KEAKI_OK;
}
keaki_status keaki_hip_fk_shard_open(keaki_hip_ctx* ctx, keaki_hip_fk_shard* fk, int32_t step, const uint64_t* coeffs, void* d_send, void* d_recv,
                                     uint64_t* proofs_out_aff) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.fk_shard_open");
  if (!fk || step < 0 || step > 3) return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_open: bad argument");
  if (!fk->plan.hat_s_ready) return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_open: keaki_hip_fk_shard_setup steps 0 and 1 have not run");
  const size_t d = (size_t)1 << fk->plan.log2d;
  if ((step == 0 && (!coeffs || !d_send)) || (step == 1 && (!d_send || !d_recv)) || (step == 2 && (!d_send || !d_recv)) || (step == 3 && (!d_recv || !proofs_out_aff)))
    return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_open: step %d is missing a buffer", step);
  if (step != fk->open_next && step != 0)      // step 0 may always start a new polynomial
    return fail(ctx, KEAKI_ERR_BAD_ARG, "fk_shard_open: step %d out of order (step %d is next)", step, fk->open_next);
  if (step == 0) HIP_TRY(ctx, hipMemcpyAsync(fk->plan.coeffs, coeffs, d * 32, hipMemcpyHostToDevice, ctx->stream));
  if (step < 3) {
    ST_TRY(fk_shard_open_run(ctx, fk->plan, step, d_send, d_recv, nullptr));
    fk->open_next = step + 1;
    return KEAKI_OK;
  }
  ST_TRY(reserve(ctx, ctx->io_d, d * 64));
  ST_TRY(fk_shard_open_run(ctx, fk->plan, 3, nullptr, d_recv, ctx->io_d.p));
  prefault_out(ctx, proofs_out_aff, d * 64);
  fk->open_next = 0;
  return download(ctx, proofs_out_aff, ctx->io_d.p, d * 64);
}

// ---- test hook: line table of a fixed Q (MILLER_MAX_LINES x 2 parities x 3 Fq, Montgomery)
keaki_status keaki_hip_g2_prepare(keaki_hip_ctx* ctx, const uint64_t* g2_aff, uint64_t* lines_out, size_t lines_out_bytes) {
  CTX_GUARD(ctx);
  if (!g2_aff || !lines_out || lines_out_bytes < g2_prepared_bytes()) return fail(ctx, KEAKI_ERR_BAD_ARG, "g2_prepare: bad argument");
  ST_TRY(upload(ctx, ctx->io_a, g2_aff, 128));
  ST_TRY(reserve(ctx, ctx->io_b, g2_prepared_bytes()));
  HIP_TRY(ctx, hipMemsetAsync(ctx->io_b.p, 0, g2_prepared_bytes(), ctx->stream));
  ST_TRY(g2_prepare_run(ctx, ctx->io_a.p, ctx->io_b.p));
  ST_TRY(reserve(ctx, ctx->io_c, g2_prepared_bytes()));
  HIP_TRY(ctx, hipMemsetAsync(ctx->io_c.p, 0, g2_prepared_bytes(), ctx->stream));
  ST_TRY(lines_to256_run(ctx, ctx->io_b.p, ctx->io_c.p));
  return download(ctx, lines_out, ctx->io_c.p, g2_prepared_bytes());
}

// ---- test hook: Miller loop alone (n x 12 Fq Montgomery out)
keaki_status keaki_hip_miller_loop_batch(keaki_hip_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t n, uint64_t* f_mont_out) {
  CTX_GUARD(ctx);
  if (n == 0) return KEAKI_OK;
  if (!g1_aff || !g2_aff || !f_mont_out) return fail(ctx, KEAKI_ERR_BAD_ARG, "miller_loop_batch: null pointer");
  ST_TRY(upload(ctx, ctx->io_a, g1_aff, n * 64));
  ST_TRY(upload(ctx, ctx->io_b, g2_aff, n * 128));
  ST_TRY(reserve(ctx, ctx->io_c, n * 384));
  ST_TRY(miller_only_run(ctx, ctx->io_a.p, ctx->io_b.p, n, ctx->io_c.p));
  return download(ctx, f_mont_out, ctx->io_c.p, n * 384);
}

// ---- test hook: final exponentiation of caller-supplied Miller-loop outputs (n x 12 Fq, Montgomery) -> n x 384 GT bytes
keaki_status keaki_hip_final_exp_batch(keaki_hip_ctx* ctx, const uint64_t* f_mont, size_t n, uint8_t* gt_out) {
  CTX_GUARD(ctx);
  if (n == 0) return KEAKI_OK;
  if (!f_mont || !gt_out) return fail(ctx, KEAKI_ERR_BAD_ARG, "final_exp_batch: null pointer");
  ST_TRY(upload(ctx, ctx->io_a, f_mont, n * 384));
  ST_TRY(reserve(ctx, ctx->io_b, n * 384));
  ST_TRY(final_exp_only_run(ctx, ctx->io_a.p, n, ctx->io_b.p));
  return download(ctx, gt_out, ctx->io_b.p, n * 384);
}

// ---- KZG open on the device (row f-4): value = p(z), proof = commit((p - p(z)) / (x - z)) ------------------------------------------
keaki_status keaki_hip_kzg_open(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, const uint64_t* coeffs, size_t n, const uint64_t* point,
                                uint64_t* proof_out_jac, uint64_t* value_out) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.kzg_open");
  if (!srs || !point || !proof_out_jac || (n && !coeffs)) return fail(ctx, KEAKI_ERR_BAD_ARG, "kzg_open: null pointer");
  SRS_CHECK(ctx, srs, "kzg_open");
  if (n && n - 1 > srs->n) return fail(ctx, KEAKI_ERR_TOO_LARGE, "msm: %zu scalars but the SRS holds %zu points", n - 1, srs->n);
  const size_t nq = n ? n - 1 : 0;
  const size_t o_q = 0, o_v = o_q + (nq + 1) * 32, o_w = o_v + 32, total = o_w + open_quotient_work_bytes(n + 1);
  ST_TRY(reserve(ctx, ctx->io_a, n ? n * 32 : 16));
  ST_TRY(reserve(ctx, ctx->io_c, total));
  ST_TRY(reserve(ctx, ctx->io_b, 96));
  char* b = (char*)ctx->io_c.p;
  HIP_TRY(ctx, hipMemsetAsync(b + o_v, 0, 32, ctx->stream));                     // the zero polynomial evaluates to 0
  const auto tb = srs_tables(srs);
  // Long polynomials come up in chunks FROM THE TOP (the quotient's recurrence Q_i = c_i + z Q_(i+1) runs downwards): chunk j's coefficients are
  // uploaded on the copy stream while chunk j - 1's quotient and MSM pass run; its quotient starts from the carry Q_hi the chunk above left
  // (planted as one more "coefficient" behind the chunk), and the MSM consumes the quotient chunk by chunk (msm_host.hip.h: MsmPipe).
  std::vector<std::pair<size_t, size_t>> cch;                                    // coefficient chunks [lo, hi), top first
  // (automatic from 2^21 coefficients on: the first chunk's quotient stays in front of the first pass; 2^20: 2.71 ms in three chunks against 2.64
  // with the copy in front, 2^21: 4.05 / 4.36, 2^22: 6.43 / 8.10, 2^24: 19.2 / 28.0 ms -- profiles/r05_open_chunked.txt)
  if (ctx->tune.msm_pipe_chunks >= 2 || (ctx->tune.msm_pipe_chunks < 0 && ctx->tune.pipe_chunks && n >= ((size_t)1 << 21))) {     // "pipe_chunks" = 0 or "msm_pipe_chunks" = 0 / 1: one copy in front, on the context's stream
    const std::vector<size_t> bounds = msm_pipe_bounds(ctx->tune, n);
    for (size_t j = 0; j + 1 < bounds.size(); j++) cch.push_back({n - bounds[j + 1], n - bounds[j]});
    if (cch.size() >= 2 && cch.back().second == 1) { cch[cch.size() - 2].first = 0; cch.pop_back(); }     // the lowest chunk must leave a quotient coefficient
  }
  if (cch.size() <= 1 || nq == 0) {
    if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, coeffs, n * 32, hipMemcpyHostToDevice, ctx->stream));
    // a failing call, too, returns only when no copy reads `coeffs` any more (header): the successful path synchronises in `download` below
    struct UploadFence { hipStream_t s; bool armed; ~UploadFence() { if (armed) (void)hipStreamSynchronize(s); } } fence{ctx->stream, n != 0};
    if (n) ST_TRY(open_quotient_run(ctx, ctx->io_a.p, n, point, b + o_q, b + o_v, b + o_w));
    ST_TRY(msm_g1_run(ctx, srs->d, srs->n, b + o_q, nq, ctx->io_b.p, tb.first, tb.second));
    fence.armed = false;
  } else {
    // Three streams: the copy stream brings chunk j up, the AUX stream turns it into quotient coefficients (a handful of short,
    // latency-bound kernels that depend on the chunk above only through its carry), the context's stream runs the MSM passes. The quotient of
    // chunk j + 1 therefore runs beside the MSM pass of chunk j instead of in front of its own (2^24 coefficients: 20.8 -> 18.5 ms).
    ChunkUploader up(ctx);
    ST_TRY(up.begin());
    ST_TRY(aux_ready(ctx));
    for (auto& e : ctx->open_ev) if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t ax = ctx->aux_stream, main_st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->open_ev[4], main_st));              // behind the value slot's memset and every earlier user of io_a / io_c
    HIP_TRY(ctx, hipStreamWaitEvent(ax, ctx->open_ev[4], 0));
    struct AuxFence { hipStream_t s; ~AuxFence() { (void)hipStreamSynchronize(s); } } aux_fence{ax};      // nothing of this call is left on it at any exit
    MsmPipe pipe;
    for (const auto& c : cch) pipe.ranges.push_back({c.first ? c.first - 1 : 0, c.second - 1 - (c.first ? c.first - 1 : 0)});   // q_i = Q_(i+1): chunk [lo, hi) yields q_(lo-1) .. q_(hi-2)
    char* a = (char*)ctx->io_a.p;
    pipe.stage = [&](size_t j) -> keaki_status {
      const size_t lo = cch[j].first, hi = cch[j].second;
      const int h = (int)(j & 1);
      HIP_TRY(ctx, hipMemcpyAsync(a + lo * 32, (const char*)coeffs + lo * 32, (hi - lo) * 32, hipMemcpyHostToDevice, up.cs));
      HIP_TRY(ctx, hipEventRecord(ctx->open_ev[h], up.cs));
      HIP_TRY(ctx, hipStreamWaitEvent(ax, ctx->open_ev[h], 0));
      {
        StreamSwap on_aux(ctx, ax);                                      // the launchers enqueue on ctx->stream
        // the carry: Q_hi = q_(hi-1), written by the chunk above; it takes the place of c_hi, which that chunk has consumed
        if (j) HIP_TRY(ctx, hipMemcpyAsync(a + hi * 32, b + o_q + (hi - 1) * 32, 32, hipMemcpyDeviceToDevice, ax));
        ST_TRY(open_quotient_run(ctx, a + lo * 32, hi - lo + (j ? 1 : 0), point, b + o_q + lo * 32, lo ? b + o_q + (lo - 1) * 32 : b + o_v, b + o_w, j != 0));
      }
      HIP_TRY(ctx, hipEventRecord(ctx->open_ev[2 + h], ax));
      HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->open_ev[2 + h], 0));
      return KEAKI_OK;
    };
    ST_TRY(msm_g1_run(ctx, srs->d, srs->n, b + o_q, nq, ctx->io_b.p, tb.first, tb.second, &pipe));
  }
  ST_TRY(download(ctx, proof_out_jac, ctx->io_b.p, 96));
  if (value_out) ST_TRY(download(ctx, value_out, b + o_v, 32));
  resolve_timing(ctx);
  return KEAKI_OK;
}

// the quotient alone (host in / out): what keaki_hip_group_kzg_open feeds to the MSM of all members
keaki_status keaki_hip_kzg_quotient(keaki_hip_ctx* ctx, const uint64_t* coeffs, size_t n, const uint64_t* point, uint64_t* quotient_out,
                                    uint64_t* value_out) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.kzg_quotient");
  if (!point || (n && !coeffs) || (n > 1 && !quotient_out)) return fail(ctx, KEAKI_ERR_BAD_ARG, "kzg_quotient: null pointer");
  const size_t nq = n ? n - 1 : 0;
  const size_t o_q = 0, o_v = o_q + (nq + 1) * 32, o_w = o_v + 32, total = o_w + open_quotient_work_bytes(n);
  ST_TRY(upload(ctx, ctx->io_a, coeffs, n * 32));
  ST_TRY(reserve(ctx, ctx->io_c, total));
  char* b = (char*)ctx->io_c.p;
  HIP_TRY(ctx, hipMemsetAsync(b + o_v, 0, 32, ctx->stream));
  if (n) ST_TRY(open_quotient_run(ctx, ctx->io_a.p, n, point, b + o_q, b + o_v, b + o_w));
  if (nq) HIP_TRY(ctx, hipMemcpyAsync(quotient_out, b + o_q, nq * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (value_out) HIP_TRY(ctx, hipMemcpyAsync(value_out, b + o_v, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return KEAKI_OK;
}

// ---- KZG verify -----------------------------------------------------------------------------------------------------
keaki_status keaki_hip_kzg_verify(keaki_hip_ctx* ctx, const uint64_t* com_aff, const uint64_t* tau_g2_aff, const uint64_t* point,
                                  const uint64_t* value, const uint64_t* proof_aff, int32_t* ok_out) {
  CTX_GUARD(ctx);
  TRACE_SCOPE("keaki.kzg_verify");
  if (!com_aff || !tau_g2_aff || !point || !value || !proof_aff || !ok_out) return fail(ctx, KEAKI_ERR_BAD_ARG, "kzg_verify: null pointer");
  // The predicate exactly as src/kzg.rs:135-143 writes it: e(com - value g1, g2) == e(proof, [tau]_2 - point g2). Both inner points are fixed-base
  // sums over the 8-bit tables of the generators (k_verify_points, 0.25 ms); the two pairings are ONE launch of the twelve-lane kernel with the
  // lines computed on the fly (1.44 ms). (Rounds 2-4 moved point * proof across the pairing so that both second slots were tabulated: a variable-
  // base ladder of 0.99 ms in front of 1.38 ms of pairings.)
  // io block: [qs: g2 128 | Q 128] [in: com 64 | proof 64 | value 32 | point 32 | tau_g2 128] [ps: A 64 | proof 64] [gt 768]
  constexpr size_t O_Q = 0, O_IN = 256, O_P = O_IN + 320, O_GT = O_P + 128, IO_BYTES = O_GT + 768;
  constexpr uint32_t WB = 8;
  if (!ctx->verify_ready) {
    ST_TRY(reserve(ctx, ctx->verify_io, IO_BYTES));
    ST_TRY(g2_generator_to(ctx, (char*)ctx->verify_io.p + O_Q));
    ctx->verify_ready = true;               // only after every step succeeded (a failed init is retried by the next call)
  }
  char* io = (char*)ctx->verify_io.p;
  if (!ctx->verify_tables_ready) {
    const size_t FBX = fb_table_entries(WB);
    ST_TRY(reserve(ctx, ctx->fbs_g1_gen, FBX * G1_AFF_BYTES + G1_AFF_BYTES));
    void* g1pt = (char*)ctx->fbs_g1_gen.p + FBX * G1_AFF_BYTES;
    ST_TRY(g1_generator_to(ctx, g1pt));
    ST_TRY(g1_fb_table_run(ctx, g1pt, ctx->fbs_g1_gen.p, WB));
    if (!ctx->fbs_ready) {                  // shared with the small-batch path of encapsulate (which also keeps [tau]_2's table there)
      ST_TRY(reserve(ctx, ctx->fbs_g2_gen, FBX * G2_AFF_BYTES));
      ST_TRY(reserve(ctx, ctx->fbs_tau, FBX * G2_AFF_BYTES));
      ST_TRY(g2_fb_table_run(ctx, io + O_Q, ctx->fbs_g2_gen.p, WB));
      ctx->fbs_ready = true;
      ctx->fbs_tau_valid = false;
    }
    ctx->verify_tables_ready = true;
  }
  uint64_t in[40];
  memcpy(in, com_aff, 64);
  memcpy(in + 8, proof_aff, 64);
  memcpy(in + 16, value, 32);
  memcpy(in + 20, point, 32);
  memcpy(in + 24, tau_g2_aff, 128);
  HIP_TRY(ctx, hipMemcpyAsync(io + O_IN, in, 320, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(io + O_P + 64, io + O_IN + 64, 64, hipMemcpyDeviceToDevice, ctx->stream));      // the proof into the pairing's first slot
  ST_TRY(verify_points_run(ctx, ctx->fbs_g1_gen.p, ctx->fbs_g2_gen.p, WB, io + O_IN, io + O_IN + 192, io + O_IN + 128, io + O_IN + 160, io + O_P, io + O_Q + 128));
  ST_TRY(pairing_run(ctx, io + O_P, io + O_Q, 1, 2, io + O_GT));
  uint8_t gt[768];
  ST_TRY(download(ctx, gt, io + O_GT, 768));                                       // synchronises: `in` stays alive until here
  *ok_out = memcmp(gt, gt + 384, 384) == 0 ? 1 : 0;
  return KEAKI_OK;
}

// ---- SRS ingest: on-curve check (row f-3) ------------------------------------------------------------------------
static keaki_status curve_check_common(keaki_hip_ctx* ctx, bool g2, const void* d_pts, size_t n, uint64_t* n_off_curve, uint64_t* first_off_curve) {
  ST_TRY(reserve(ctx, ctx->io_e, 16));
  const uint64_t init[2] = {0, ~0ull};
  HIP_TRY(ctx, hipMemcpyAsync(ctx->io_e.p, init, 16, hipMemcpyHostToDevice, ctx->stream));
  if (n) ST_TRY(g2 ? g2_curve_check_run(ctx, d_pts, n, ctx->io_e.p) : g1_curve_check_run(ctx, d_pts, n, ctx->io_e.p));
  uint64_t res[2];
  ST_TRY(download(ctx, res, ctx->io_e.p, 16));
  *n_off_curve = res[0];
  if (first_off_curve) *first_off_curve = res[1];
  return KEAKI_OK;
}
keaki_status keaki_hip_srs_g1_check(keaki_hip_ctx* ctx, const keaki_hip_srs_g1* srs, uint64_t* n_off_curve, uint64_t* first_off_curve) {
  CTX_GUARD(ctx);
  if (!srs || !n_off_curve) return fail(ctx, KEAKI_ERR_BAD_ARG, "srs_g1_check: null pointer");
  return curve_check_common(ctx, false, srs->d, srs->n, n_off_curve, first_off_curve);
}
keaki_status keaki_hip_g2_check(keaki_hip_ctx* ctx, const uint64_t* points_aff, size_t n, uint64_t* n_off_curve, uint64_t* first_off_curve) {
  CTX_GUARD(ctx);
  if (!n_off_curve || (n && !points_aff)) return fail(ctx, KEAKI_ERR_BAD_ARG, "g2_check: null pointer");
  ST_TRY(upload(ctx, ctx->io_a, points_aff, n * 128));
  return curve_check_common(ctx, true, ctx->io_a.p, n, n_off_curve, first_off_curve);
}

// ---- self-test --------------------------------------------------------------------------------------------
keaki_status keaki_hip_selftest_field(keaki_hip_ctx* ctx, uint32_t blocks, uint32_t iters, uint32_t seed, uint64_t* mismatches_out) {
  CTX_GUARD(ctx);
  if (!mismatches_out || blocks == 0) return fail(ctx, KEAKI_ERR_BAD_ARG, "selftest_field: bad argument");
  ST_TRY(reserve(ctx, ctx->io_e, 16));
  HIP_TRY(ctx, hipMemsetAsync(ctx->io_e.p, 0, 8, ctx->stream));
  ST_TRY(selftest_field_run(ctx, blocks, iters, seed, ctx->io_e.p));
  return download(ctx, mismatches_out, ctx->io_e.p, 8);
}

}  // extern "C"
