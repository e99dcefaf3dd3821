// Library context, error reporting, memory staging, variogram validation.
#include "gss_internal.h"

#include <atomic>
#include <mutex>

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace gss {

std::recursive_mutex& api_mutex() {
  static std::recursive_mutex mu;
  return mu;
}

// ---- profiling registry --------------------------------------------------------------------
struct ProfEntry {
  std::vector<hipEvent_t> start, stop;
  double ms = 0.0;
  int64_t launches = 0;
};
static bool g_prof_on = false;
static std::map<std::string, ProfEntry> g_prof;
static std::mutex g_prof_mu;   // the registry, the block cache and the stream chain below are shared by every host thread

bool prof_enabled() { return g_prof_on; }

void prof_begin(const char* name, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  ProfEntry& e = g_prof[name];
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess) return;
  if (hipEventCreate(&b) != hipSuccess) {
    (void)hipEventDestroy(a);
    return;
  }
  e.start.push_back(a);
  e.stop.push_back(b);
  (void)hipEventRecord(a, s);
}

void prof_end(const char* name, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  ProfEntry& e = g_prof[name];
  if (e.stop.size() == e.start.size() && !e.stop.empty()) (void)hipEventRecord(e.stop.back(), s);
}

static void prof_collect(ProfEntry& e) {
  for (size_t i = 0; i < e.start.size(); ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(e.stop[i]) == hipSuccess && hipEventElapsedTime(&ms, e.start[i], e.stop[i]) == hipSuccess) {
      e.ms += ms;
      e.launches += 1;
    }
    (void)hipEventDestroy(e.start[i]);
    (void)hipEventDestroy(e.stop[i]);
  }
  e.start.clear();
  e.stop.clear();
}

// ---- cross-stream ordering of the library's own work (see to_stream / EntryGuard in gss_internal.h) -----------
// All of this runs under the library's lock (api_mutex).
static hipStream_t g_last_stream = nullptr;    // stream of the last finished call (compared, never used)
static bool g_have_chain = false;              // g_chain_event has been recorded on it
static hipEvent_t g_chain_event = nullptr;
static hipStream_t g_call_stream = nullptr;    // stream of the call in progress
static bool g_call_has_stream = false;
static int g_entry_depth = 0;

EntryGuard::EntryGuard() : lock(api_mutex()) { ++g_entry_depth; }

EntryGuard::~EntryGuard() {
  if (--g_entry_depth != 0 || !g_call_has_stream) return;
  g_call_has_stream = false;
  if (!g_chain_event && hipEventCreateWithFlags(&g_chain_event, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    g_chain_event = nullptr;
  }
  g_have_chain = g_chain_event && hipEventRecord(g_chain_event, g_call_stream) == hipSuccess;
  if (!g_have_chain) (void)hipGetLastError();
  g_last_stream = g_call_stream;
}

hipStream_t to_stream(void* sv) {
  hipStream_t s = reinterpret_cast<hipStream_t>(sv);
  if (!g_call_has_stream && g_entry_depth > 0) {
    if (g_last_stream != s || !g_have_chain) {
      // a different stream than the call before (or no usable event): wait for what that call queued
      if (!(g_have_chain && hipStreamWaitEvent(s, g_chain_event, 0) == hipSuccess) && g_chain_event) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
      }
    }
    g_call_stream = s;
    g_call_has_stream = true;
  }
  return s;
}

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Device allocations are recycled through a small exact-size free list: handles are created and destroyed once
// per solve with identical sizes, and hipMalloc / hipFree cost 0.1-1 ms each (hipFree also synchronises the
// device, which would stop the host from queueing the next solve behind the running one).  Re-use is ordered by
// submission: every entry point goes through to_stream(), which chains a call on a new stream behind the work
// queued on the previous one, so a block released while a kernel still reads it cannot be overwritten before
// that kernel finishes -- whichever streams the caller uses (handles are still not thread-safe).
struct PoolBlock {
  void* p;
  size_t bytes;
};
static std::vector<PoolBlock> g_pool;
static size_t g_pool_bytes = 0;
static std::mutex g_pool_mu;
// Sized for 288 GB of HBM: up to 16 GiB of released blocks stay cached (GSS_POOL_MAX_MB overrides; a failing
// hipMalloc gives them all back first), single blocks up to a quarter of that -- the 1.2 GB state of a configs[3]
// LUGS handle and the 2 GiB FFTGS buffers are re-used instead of being freed (a device synchronisation) and
// allocated again on the next solve.
static size_t pool_max_bytes() {
  static const size_t v = [] {
    const char* e = std::getenv("GSS_POOL_MAX_MB");
    if (e) {
      const long long mb = atoll(e);
      return (size_t)(mb < 0 ? 0 : mb) << 20;
    }
    // default: a quarter of what is free when the library first allocates, at most 16 GiB -- a second rank on the same
    // device, or another allocator in this process, must not find the memory gone (they cannot make us give it back)
    size_t fr = 0, tot = 0;
    size_t cap = (size_t)16 << 30;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr / 4 < cap) cap = fr / 4;
    return cap;
  }();
  return v;
}
constexpr size_t POOL_MAX_COUNT = 96;

int32_t DevBuf::alloc(size_t nbytes) {
  release();
  if (nbytes == 0) nbytes = 8;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  // best fit among the cached blocks: the smallest one that holds the request and is at most a quarter larger
  // (sizes follow the problem: a point or a datum more must not cost a hipMalloc)
  size_t best = (size_t)-1;
  for (size_t i = 0; i < g_pool.size(); ++i) {
    const size_t b = g_pool[i].bytes;
    if (b >= nbytes && b - nbytes <= nbytes / 4 && (best == (size_t)-1 || b < g_pool[best].bytes)) best = i;
  }
  if (best != (size_t)-1) {
    p = g_pool[best].p;
    bytes = nbytes;
    cap = g_pool[best].bytes;
    g_pool_bytes -= cap;
    g_pool.erase(g_pool.begin() + (std::ptrdiff_t)best);   // keeps the list in order of release (oldest first)
    return GSS_OK;
  }
  hipError_t e = hipMalloc(&p, nbytes);
  if (e != hipSuccess && !g_pool.empty()) {  // give cached blocks back and retry once
    for (auto& b : g_pool) (void)hipFree(b.p);
    g_pool.clear();
    g_pool_bytes = 0;
    e = hipMalloc(&p, nbytes);
  }
  if (e != hipSuccess) {
    p = nullptr;
    set_error("hipMalloc(%zu bytes) failed: %s", nbytes, hipGetErrorString(e));
    return GSS_ERR_ALLOC;
  }
  bytes = nbytes;
  cap = nbytes;
  return GSS_OK;
}

void DevBuf::release() {
  if (p) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (cap <= pool_max_bytes() / 4) {
      // a full cache gives up its oldest blocks: sizes change from solve to solve, and the sizes of the latest
      // solves are the ones most likely to come back
      while (!g_pool.empty() && (g_pool.size() >= POOL_MAX_COUNT || g_pool_bytes + cap > pool_max_bytes())) {
        (void)hipFree(g_pool.front().p);
        g_pool_bytes -= g_pool.front().bytes;
        g_pool.erase(g_pool.begin());
      }
      g_pool.push_back(PoolBlock{p, cap});
      g_pool_bytes += cap;
    } else {
      (void)hipFree(p);
    }
  }
  p = nullptr;
  bytes = 0;
  cap = 0;
}

int32_t Staged::in(const void* src, size_t bytes, int32_t mem, hipStream_t s) {
  if (src == nullptr) {
    p = nullptr;
    return GSS_OK;
  }
  if (mem == GSS_MEM_DEVICE) {
    p = const_cast<void*>(src);
    return GSS_OK;
  }
  GSS_TRY(own.alloc(bytes));
  p = own.p;
  GSS_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s));
  return GSS_OK;
}

int32_t Staged::out(void* dst, size_t bytes, int32_t mem) {
  if (dst == nullptr) {
    p = nullptr;
    return GSS_OK;
  }
  if (mem == GSS_MEM_DEVICE) {
    p = dst;
    return GSS_OK;
  }
  GSS_TRY(own.alloc(bytes));
  p = own.p;
  return GSS_OK;
}

int32_t Staged::back(void* dst, size_t bytes, int32_t mem, hipStream_t s) {
  if (dst == nullptr || mem == GSS_MEM_DEVICE) return GSS_OK;
  GSS_HIP(hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

// ---- HostPipe (gss_internal.h) ---------------------------------------------------------------------------------
// Helper streams of the process.  They are all created together, in one fixed order, the first time any of them is
// asked for: which hardware queue a stream lands on depends on how many streams exist when it is created, and two
// streams on one hardware queue do not overlap -- with per-subsystem streams created on demand, the look-ahead of the
// LUGS preprocess lost its overlap (32.5 -> 35.5 ms) whenever an FFTGS realisation had created its slab streams
// first.  Every use is fenced by events against the caller's stream on both sides.
//   HELPER_LOOKAHEAD  low priority   look-ahead / side work of the blocked factorisations (dense_la.hip, lugs.hip)
//   HELPER_GEN0..2    normal         FFTGS slabs, split products of gss_lugs_realize, host <-> device copies
//   HELPER_FIT        high priority  asynchronous kriging fit
hipStream_t helper_stream(int which) {
  static std::mutex mu;
  static bool made = false;
  static hipStream_t st[HELPER_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::lock_guard<std::mutex> lock(mu);
  if (!made) {
    made = true;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
    const char* order = std::getenv("GSS_HELPER_ORDER");   // experiment: e.g. "LF012"
    if (!order) order = "L012F";
    for (const char* c = order; *c; ++c) {
      int w = -1, pr = 0;
      bool prio = false;
      if (*c == 'L') { w = HELPER_LOOKAHEAD; pr = lo; prio = true; }
      else if (*c == 'F') { w = HELPER_FIT; pr = hi; prio = true; }
      else if (*c >= '0' && *c <= '2') w = HELPER_GEN0 + (*c - '0');
      if (w < 0 || st[w]) continue;
      const hipError_t e = prio ? hipStreamCreateWithPriority(&st[w], hipStreamNonBlocking, pr)
                                : hipStreamCreateWithFlags(&st[w], hipStreamNonBlocking);
      if (e != hipSuccess) st[w] = nullptr;
    }
  }
  return (which >= 0 && which < HELPER_COUNT) ? st[which] : nullptr;
}

static hipStream_t host_copy_stream(int i) { return helper_stream(HELPER_GEN0 + i); }

HostPipe::~HostPipe() {
  if (on) {   // an early exit of the caller: copies may still be queued on the copy streams while the device images
    if (cin) (void)hipStreamSynchronize(cin);     // (whole-call scratch) are about to return to the pool
    if (cout) (void)hipStreamSynchronize(cout);
  }
  if (ev_in) (void)hipEventDestroy(ev_in);
  if (ev_done) (void)hipEventDestroy(ev_done);
}

int32_t HostPipe::begin(int32_t mem, int64_t m, hipStream_t s) {
  static const bool enabled = !(std::getenv("GSS_HOST_PIPELINE") && std::getenv("GSS_HOST_PIPELINE")[0] == '0');
  on = false;
  if (mem != GSS_MEM_HOST || !enabled || m <= PIECE) return GSS_OK;
  cin = host_copy_stream(0);
  cout = host_copy_stream(1);
  if (!cin || !cout) return GSS_OK;
  if (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ev_done, hipEventDisableTiming) != hipSuccess)
    return GSS_OK;
  // the scratch behind the device images may still be in use by work queued on s: the copy streams start behind it
  GSS_HIP(hipEventRecord(ev_done, s));
  GSS_HIP(hipStreamWaitEvent(cin, ev_done, 0));
  GSS_HIP(hipStreamWaitEvent(cout, ev_done, 0));
  on = true;
  return GSS_OK;
}

void HostPipe::add_in(const void* host, void* dev, size_t stride) {
  if (!host || !dev || nin >= 4) return;
  ins[nin++] = Arr{static_cast<const char*>(host), nullptr, static_cast<char*>(dev), stride};
}

void HostPipe::add_out(void* host, void* dev, size_t stride) {
  if (!host || !dev || nout >= 6) return;
  outs[nout++] = Arr{nullptr, static_cast<char*>(host), static_cast<char*>(dev), stride};
}

int32_t HostPipe::fetch(int64_t off, int64_t n, hipStream_t s) {
  if (!on) return GSS_OK;
  for (int i = 0; i < nin; ++i)
    GSS_HIP(hipMemcpyAsync(ins[i].dev + (size_t)off * ins[i].stride, ins[i].host_in + (size_t)off * ins[i].stride,
                           (size_t)n * ins[i].stride, hipMemcpyHostToDevice, cin));
  GSS_HIP(hipEventRecord(ev_in, cin));
  GSS_HIP(hipStreamWaitEvent(s, ev_in, 0));
  return GSS_OK;
}

int32_t HostPipe::deliver(int64_t off, int64_t n, hipStream_t s) {
  if (!on) return GSS_OK;
  // the piece before (its fence is already on the copy stream), behind this piece's launches
  for (int i = 0; i < nout && pend_n > 0; ++i)
    GSS_HIP(hipMemcpyAsync(outs[i].host_out + (size_t)pend_off * outs[i].stride,
                           outs[i].dev + (size_t)pend_off * outs[i].stride, (size_t)pend_n * outs[i].stride,
                           hipMemcpyDeviceToHost, cout));
  GSS_HIP(hipEventRecord(ev_done, s));
  GSS_HIP(hipStreamWaitEvent(cout, ev_done, 0));
  pend_off = off;
  pend_n = n;
  return GSS_OK;
}

int32_t HostPipe::finish(hipStream_t s) {
  if (!on) return GSS_OK;
  for (int i = 0; i < nout && pend_n > 0; ++i)
    GSS_HIP(hipMemcpyAsync(outs[i].host_out + (size_t)pend_off * outs[i].stride,
                           outs[i].dev + (size_t)pend_off * outs[i].stride, (size_t)pend_n * outs[i].stride,
                           hipMemcpyDeviceToHost, cout));
  pend_n = 0;
  GSS_HIP(hipStreamSynchronize(cout));
  GSS_HIP(hipStreamSynchronize(s));
  on = false;   // (nothing left for the destructor to join)
  return GSS_OK;
}

// ---- OutStream (gss_internal.h) ------------------------------------------------------------------------------
constexpr size_t BOUNCE_BYTES = (size_t)32 << 20;
static void* g_bounce[2] = {nullptr, nullptr};   // pinned, process-wide, freed by gss_shutdown
static std::mutex g_bounce_mu;

static void* bounce_buffer(int i) {
  std::lock_guard<std::mutex> lock(g_bounce_mu);
  if (!g_bounce[i] && hipHostMalloc(&g_bounce[i], BOUNCE_BYTES, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    g_bounce[i] = nullptr;
  }
  return g_bounce[i];
}

static void free_bounce_buffers() {
  std::lock_guard<std::mutex> lock(g_bounce_mu);
  for (int i = 0; i < 2; ++i) {
    if (g_bounce[i]) (void)hipHostFree(g_bounce[i]);
    g_bounce[i] = nullptr;
  }
}

static bool host_pointer_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();   // an ordinary (pageable) host pointer is "invalid value" to the runtime
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

struct BounceCopy {
  char* dst;
  const char* src;
  size_t bytes;
};

// runs on a runtime thread in stream order behind the transfer of the piece; no HIP calls in here
static void bounce_copy_fn(void* pv) {
  BounceCopy* c = static_cast<BounceCopy*>(pv);
  constexpr int NT = 4;
  const size_t part = (c->bytes / NT + 4095) & ~(size_t)4095;
  std::thread th[NT - 1];
  int started = 0;
  for (int t = 1; t < NT; ++t) {
    const size_t lo = (size_t)t * part;
    if (lo >= c->bytes) break;
    const size_t n = c->bytes - lo < part ? c->bytes - lo : part;
    th[started++] = std::thread([=] { std::memcpy(c->dst + lo, c->src + lo, n); });
  }
  std::memcpy(c->dst, c->src, c->bytes < part ? c->bytes : part);
  for (int t = 0; t < started; ++t) th[t].join();
  delete c;
}

static size_t out_chunk_bytes() {   // read per call: the tests shrink it to push small problems through many chunks
  const char* e = std::getenv("GSS_OUT_CHUNK_MB");
  const long long mb = e ? atoll(e) : 256;
  return (size_t)(mb < 1 ? 1 : mb) << 20;
}

static std::atomic<int64_t> g_stat_ring_bytes{0}, g_stat_chunks{0};

OutStream::~OutStream() {
  if (issued) {  // an early exit: nothing may still write into the ring or read it when its blocks return to the pool
    for (int i = 0; i < 2; ++i)
      if (cs[i]) (void)hipStreamSynchronize(cs[i]);
  }
  for (int i = 0; i < DEPTH; ++i) {
    if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
    if (ev_free[i]) (void)hipEventDestroy(ev_free[i]);
  }
  if (ev_join) (void)hipEventDestroy(ev_join);
}

size_t OutStream::staged_bytes() const {
  size_t b = 0;
  for (int i = 0; i < DEPTH; ++i) b += ring[i].bytes;
  return b;
}

int64_t OutStream::default_chunk(size_t real_bytes_, int64_t nreals_) {
  int64_t c = real_bytes_ ? (int64_t)(out_chunk_bytes() / real_bytes_) : nreals_;
  if (c < 1) c = 1;
  return c > nreals_ ? nreals_ : c;
}

int32_t OutStream::begin(void* dst_, size_t real_bytes_, int64_t nreals_, int32_t mem, hipStream_t s,
                         int64_t chunk_reals) {
  dst = static_cast<char*>(dst_);
  real_bytes = real_bytes_;
  nreals = nreals_;
  on = false;
  if (mem != GSS_MEM_HOST || dst == nullptr || nreals <= 0 || real_bytes == 0) return GSS_OK;
  chunk = chunk_reals > 0 ? (chunk_reals > nreals ? nreals : chunk_reals) : default_chunk(real_bytes, nreals);
  const int64_t nchunks = (nreals + chunk - 1) / chunk;
  const int depth = nchunks < DEPTH ? (int)nchunks : DEPTH;
  // not the streams the FFTGS slabs run on (GEN0, GEN1): a transfer queued there would hold the slab kernels behind it
  cs[0] = helper_stream(HELPER_GEN2);
  cs[1] = helper_stream(HELPER_LOOKAHEAD);
  GSS_REQUIRE(cs[0] != nullptr && cs[1] != nullptr, "no copy streams for the host outputs");
  for (int i = 0; i < depth; ++i) {
    GSS_TRY(ring[i].alloc(real_bytes * (size_t)chunk));
    GSS_HIP(hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming));
    GSS_HIP(hipEventCreateWithFlags(&ev_free[i], hipEventDisableTiming));
  }
  GSS_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  g_stat_ring_bytes.store((int64_t)staged_bytes());
  g_stat_chunks.store(0);
  pinned_dst = host_pointer_is_pinned(dst) && host_pointer_is_pinned(dst + real_bytes * (size_t)nreals - 1);
  if (!pinned_dst) GSS_REQUIRE(bounce_buffer(0) && bounce_buffer(1), "no pinned memory for the bounce buffers");
  // the blocks of the ring may still be in use by work queued earlier (the pool re-uses in submission order)
  GSS_HIP(hipEventRecord(ev_join, s));
  GSS_HIP(hipStreamWaitEvent(cs[0], ev_join, 0));
  GSS_HIP(hipStreamWaitEvent(cs[1], ev_join, 0));
  on = true;
  return GSS_OK;
}

int32_t OutStream::slot(int64_t r, hipStream_t s, double** out) {
  if (!on) {
    *out = reinterpret_cast<double*>(dst + (size_t)r * real_bytes);
    return GSS_OK;
  }
  const int64_t c = r / chunk;
  const int sl = (int)(c % DEPTH);
  if (r % chunk == 0 && c >= DEPTH) GSS_HIP(hipStreamWaitEvent(s, ev_free[sl], 0));   // chunk c - DEPTH has left
  *out = reinterpret_cast<double*>(static_cast<char*>(ring[sl].p) + (size_t)(r % chunk) * real_bytes);
  return GSS_OK;
}

int32_t OutStream::done(int64_t r, hipStream_t s) {
  if (!on) return GSS_OK;
  if ((r + 1) % chunk != 0 && r + 1 != nreals) return GSS_OK;
  const int64_t c = r / chunk;
  const int sl = (int)(c % DEPTH);
  const int64_t r0 = c * chunk;
  const size_t bytes = (size_t)(r + 1 - r0) * real_bytes;
  char* to = dst + (size_t)r0 * real_bytes;
  const char* from = static_cast<const char*>(ring[sl].p);
  GSS_HIP(hipEventRecord(ev_done[sl], s));
  issued = true;
  g_stat_chunks.fetch_add(1);
  if (pinned_dst) {
    hipStream_t q = cs[c & 1];
    GSS_HIP(hipStreamWaitEvent(q, ev_done[sl], 0));
    GSS_HIP(hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToHost, q));
    GSS_HIP(hipEventRecord(ev_free[sl], q));
    return GSS_OK;
  }
  GSS_HIP(hipStreamWaitEvent(cs[0], ev_done[sl], 0));
  GSS_HIP(hipStreamWaitEvent(cs[1], ev_done[sl], 0));
  for (size_t off = 0; off < bytes; off += BOUNCE_BYTES, ++piece_no) {
    const int b = (int)(piece_no & 1);
    const size_t n = bytes - off < BOUNCE_BYTES ? bytes - off : BOUNCE_BYTES;
    GSS_HIP(hipMemcpyAsync(g_bounce[b], from + off, n, hipMemcpyDeviceToHost, cs[b]));
    BounceCopy* job = new BounceCopy{to + off, static_cast<const char*>(g_bounce[b]), n};
    const hipError_t e = hipLaunchHostFunc(cs[b], bounce_copy_fn, job);
    if (e != hipSuccess) {
      delete job;
      set_error("hipLaunchHostFunc failed: %s", hipGetErrorString(e));
      return GSS_ERR_HIP;
    }
  }
  // the slot is free once both copy streams have passed this chunk
  GSS_HIP(hipEventRecord(ev_join, cs[0]));
  GSS_HIP(hipStreamWaitEvent(cs[1], ev_join, 0));
  GSS_HIP(hipEventRecord(ev_free[sl], cs[1]));
  return GSS_OK;
}

int32_t OutStream::finish(hipStream_t s) {
  if (!on) return GSS_OK;
  GSS_HIP(hipStreamSynchronize(cs[0]));
  GSS_HIP(hipStreamSynchronize(cs[1]));
  GSS_HIP(hipStreamSynchronize(s));
  issued = false;
  return GSS_OK;
}

static int32_t device_kind(int kind, double nu, int* out_kind, double* mscale) {
  *mscale = 0.0;
  switch (kind) {
    case GSS_VG_GAUSSIAN:
    case GSS_VG_EXPONENTIAL:
    case GSS_VG_SPHERICAL:
    case GSS_VG_CUBIC:
    case GSS_VG_PENTASPHERICAL:
    case GSS_VG_SINEHOLE:
      *out_kind = kind;
      return GSS_OK;
    case GSS_VG_MATERN:
      if (nu == 0.5) *out_kind = VG_MATERN12;
      else if (nu == 1.5) *out_kind = VG_MATERN32;
      else if (nu == 2.5) *out_kind = VG_MATERN52;
      else if (nu == 1.0) *out_kind = VG_MATERN1;
      else if (nu == 2.0) *out_kind = VG_MATERN2;
      else if (nu == 3.0) *out_kind = VG_MATERN3;
      else if (nu > 0.0 && nu <= 50.0) *out_kind = VG_MATERN_NU;   // general order: Temme's K_nu on the device
      else {
        set_error("Matern order nu=%g must lie in (0, 50]", nu);
        return GSS_ERR_INVALID;
      }
      *mscale = std::sqrt(2.0 * nu) * 3.0;
      return GSS_OK;
    default:
      set_error("unknown variogram kind %d", kind);
      return GSS_ERR_INVALID;
  }
}

__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, size_t nwords) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = 0u;
}

__global__ __launch_bounds__(256) void copy_f64_kernel(double* __restrict__ dst, const double* __restrict__ src,
                                                       int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

int32_t dev_zero_bytes(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return GSS_OK;
  const size_t nwords = bytes / 4;
  size_t blocks = (nwords + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), nwords);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

int32_t dev_copy_f64(double* dst, const double* src, int64_t n, hipStream_t s) {
  if (n <= 0) return GSS_OK;
  hipLaunchKernelGGL(copy_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

int32_t make_vgdev(const gss_variogram_t* vg, VgDev* out) {
  GSS_REQUIRE(vg != nullptr, "variogram is NULL");
  GSS_REQUIRE(vg->dim >= 1 && vg->dim <= 3, "variogram dim %d outside 1..3", vg->dim);
  GSS_REQUIRE(vg->sill > 0.0 && vg->nugget >= 0.0 && vg->nugget <= vg->sill, "invalid sill/nugget %g/%g",
              vg->sill, vg->nugget);
  GSS_REQUIRE(vg->nextra >= 0 && vg->nextra <= 3, "nested variogram with %d extra structures (at most 3)", vg->nextra);
  VgDev v;
  std::memset(&v, 0, sizeof(v));
  v.dim = vg->dim;
  v.aniso = vg->aniso ? 1 : 0;
  v.nextra = vg->nextra;
  v.sill = vg->sill;
  v.cs = vg->sill - vg->nugget;
  for (int k = 0; k < 3; ++k) v.ir[k] = 1.0;
  if (v.aniso) {
    for (int k = 0; k < vg->dim; ++k) {
      GSS_REQUIRE(vg->inv_radii[k] > 0.0, "anisotropic ball needs positive radii");
      v.ir[k] = vg->inv_radii[k];
    }
    v.inv_range = 1.0;
  } else {
    GSS_REQUIRE(vg->range > 0.0, "variogram range must be positive (got %g)", vg->range);
    v.inv_range = 1.0 / vg->range;
  }
  if (vg->kind == GSS_VG_POWER) {
    // gamma(h) = scaling h^exponent + nugget: no sill.  The device works with the pseudo-covariance
    // A - gamma(h) (A = `sill` field); with the unbiasedness constraint the estimates do not depend on A.
    GSS_REQUIRE(vg->nextra == 0, "a power variogram cannot be nested on the device");
    GSS_REQUIRE(vg->nu > 0.0 && vg->nu < 2.0, "power variogram exponent %g outside (0, 2)", vg->nu);
    GSS_REQUIRE(vg->sill > vg->nugget, "power variogram: pseudo-sill %g must exceed the nugget %g", vg->sill,
                vg->nugget);
    v.kind = GSS_VG_POWER;
    v.mscale = vg->range / v.cs;  // `range` carries the scaling factor
    v.pw = 0.5 * vg->nu;
    v.inv_range = 1.0;
    *out = v;
    return GSS_OK;
  }
  GSS_TRY(device_kind(vg->kind, vg->nu, &v.kind, &v.mscale));
  if (v.kind == VG_MATERN_NU) v.pw = vg->nu;
  for (int e = 0; e < vg->nextra; ++e) {
    VgExtra& x = v.ex[e];
    GSS_REQUIRE(vg->extra[e].sill > 0.0, "nested structure %d needs a positive sill contribution", e + 1);
    x.cs = vg->extra[e].sill;
    x.aniso = vg->extra[e].aniso ? 1 : 0;
    for (int k = 0; k < 3; ++k) x.ir[k] = 1.0;
    if (x.aniso) {
      for (int k = 0; k < vg->dim; ++k) {
        GSS_REQUIRE(vg->extra[e].inv_radii[k] > 0.0, "anisotropic ball needs positive radii");
        x.ir[k] = vg->extra[e].inv_radii[k];
      }
      x.inv_range = 1.0;
    } else {
      GSS_REQUIRE(vg->extra[e].range > 0.0, "variogram range must be positive (got %g)", vg->extra[e].range);
      x.inv_range = 1.0 / vg->extra[e].range;
    }
    GSS_TRY(device_kind(vg->extra[e].kind, vg->extra[e].nu, &x.kind, &x.mscale));
    if (x.kind == VG_MATERN_NU) x.pw = vg->extra[e].nu;
    v.sill += x.cs;
  }
  *out = v;
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_version(void) {
  GSS_ENTRY(); return GSS_VERSION; }

int32_t gss_device_count(int32_t* count) {
  GSS_ENTRY();
  GSS_REQUIRE(count != nullptr, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  *count = n;
  return GSS_OK;
}

int32_t gss_init(int32_t device) {
  GSS_ENTRY();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    set_error("no HIP device visible: the gfx950 kernels cannot run (there is no CPU fallback)");
    return GSS_ERR_NO_DEVICE;
  }
  GSS_REQUIRE(device >= 0 && device < n, "device %d outside 0..%d", device, n - 1);
  GSS_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  GSS_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; libgss_hip.so carries gfx950 code objects only", device, prop.gcnArchName);
    return GSS_ERR_NO_DEVICE;
  }
  return GSS_OK;
}

int32_t gss_shutdown(void) {
  GSS_ENTRY();
  free_bounce_buffers();
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (auto& b : g_pool) (void)hipFree(b.p);
  g_pool.clear();
  g_pool_bytes = 0;
  return GSS_OK;
}

int32_t gss_dev_to_host(void* dst, const void* src_dev, int64_t bytes, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(bytes >= 0 && (bytes == 0 || (dst != nullptr && src_dev != nullptr)), "gss_dev_to_host: bad arguments");
  if (bytes == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  if (host_pointer_is_pinned(dst) && host_pointer_is_pinned(static_cast<char*>(dst) + bytes - 1)) {
    GSS_HIP(hipMemcpyAsync(dst, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, s));
    GSS_HIP(hipStreamSynchronize(s));
    return GSS_OK;
  }
  // pageable destination: the bounce pipeline of OutStream (pieces of 32 MiB alternating over two copy streams, a host
  // function behind each transfer copies the piece home with a few threads)
  hipStream_t cs[2] = {helper_stream(HELPER_GEN2), helper_stream(HELPER_LOOKAHEAD)};
  GSS_REQUIRE(cs[0] && cs[1] && bounce_buffer(0) && bounce_buffer(1), "gss_dev_to_host: no copy streams / pinned memory");
  ScopedEvent ev;
  GSS_HIP(ev.create());
  GSS_HIP(hipEventRecord(ev, s));
  GSS_HIP(hipStreamWaitEvent(cs[0], ev, 0));
  GSS_HIP(hipStreamWaitEvent(cs[1], ev, 0));
  int32_t rc = GSS_OK;
  int64_t piece = 0;
  for (int64_t off = 0; off < bytes && rc == GSS_OK; off += (int64_t)BOUNCE_BYTES, ++piece) {
    const int b = (int)(piece & 1);
    const size_t n = (size_t)(bytes - off < (int64_t)BOUNCE_BYTES ? bytes - off : (int64_t)BOUNCE_BYTES);
    if (hipMemcpyAsync(g_bounce[b], static_cast<const char*>(src_dev) + off, n, hipMemcpyDeviceToHost, cs[b]) != hipSuccess) {
      rc = GSS_ERR_HIP;
      break;
    }
    BounceCopy* job = new BounceCopy{static_cast<char*>(dst) + off, static_cast<const char*>(g_bounce[b]), n};
    if (hipLaunchHostFunc(cs[b], bounce_copy_fn, job) != hipSuccess) {
      delete job;
      rc = GSS_ERR_HIP;
    }
  }
  (void)hipStreamSynchronize(cs[0]);   // whatever happened: nothing of this call is left in flight
  (void)hipStreamSynchronize(cs[1]);
  if (rc != GSS_OK) set_error("gss_dev_to_host: a transfer could not be queued");
  return rc;
}

int32_t gss_trim_pool(void) {
  GSS_ENTRY();
  GSS_HIP(hipDeviceSynchronize());   // a cached block may still be read by work queued before its release
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (auto& b : g_pool) (void)hipFree(b.p);
  g_pool.clear();
  g_pool_bytes = 0;
  return GSS_OK;
}

int32_t gss_stat(const char* name, int64_t* value) {
  GSS_REQUIRE(name != nullptr && value != nullptr, "gss_stat: NULL argument");
  if (!std::strcmp(name, "pool_bytes")) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    *value = (int64_t)g_pool_bytes;
  } else if (!std::strcmp(name, "out_ring_bytes")) {
    *value = g_stat_ring_bytes.load();
  } else if (!std::strcmp(name, "out_chunks")) {
    *value = g_stat_chunks.load();
  } else if (!std::strcmp(name, "panel_giveups")) {
    *value = panel_giveups();
  } else if (!std::strcmp(name, "ipc_route")) {
    *value = comm_last_ipc_route();   // of the last gss_state_ipc_import: 0 same device, 1 peer, 2 not visible, 3 refused
  } else {
    set_error("gss_stat: unknown counter '%s'", name);
    return GSS_ERR_INVALID;
  }
  return GSS_OK;
}

int32_t gss_last_error(char* buf, int32_t len) {
  if (buf == nullptr || len <= 0) return GSS_ERR_INVALID;
  std::strncpy(buf, g_err, (size_t)len - 1);
  buf[len - 1] = '\0';
  return GSS_OK;
}

int32_t gss_profile_enable(int32_t on) {
  GSS_ENTRY();
  g_prof_on = on != 0;
  return GSS_OK;
}

int32_t gss_profile_reset(void) {
  GSS_ENTRY();
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (auto& kv : g_prof) prof_collect(kv.second);
  g_prof.clear();
  return GSS_OK;
}

int32_t gss_profile_read(const char* name, double* total_ms, int64_t* launches) {
  GSS_ENTRY();
  GSS_REQUIRE(name != nullptr && total_ms != nullptr && launches != nullptr, "gss_profile_read: NULL argument");
  std::lock_guard<std::mutex> lock(g_prof_mu);
  auto it = g_prof.find(name);
  if (it == g_prof.end()) {
    *total_ms = 0.0;
    *launches = 0;
    return GSS_OK;
  }
  prof_collect(it->second);
  *total_ms = it->second.ms;
  *launches = it->second.launches;
  return GSS_OK;
}

int32_t gss_synchronize(void* stream) {
  GSS_ENTRY();
  GSS_HIP(hipStreamSynchronize(to_stream(stream)));
  return GSS_OK;
}

}  // extern "C"
