// wrhip_rt.h -- thin runtime layer under the GL state tracker.
//
// Product build (hipcc, gfx950): HIP runtime, one stream per context, pinned
// staging memory, hipEvents for GPU timing.
//
// WRHIP_HOSTSIM build (plain g++, *test infrastructure only*): the very same
// kernel sources are compiled for the host and each launch is executed as a
// serial loop over blocks/threads.  It exists so the deferred-rendering logic
// and the integer pixel pipeline can be checked against the oracle in the
// GPU-less authoring container (tests/, -m "not gpu").  It is built as a
// separate library (libwrhip_hostsim.so) that no product entry point loads;
// libwrhip.so itself has no CPU path and aborts when no HIP device is present.
#pragma once
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef WRHIP_HOSTSIM
// ---------------------------------------------------------------------------
#include <math.h>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
struct wr_dim3 { unsigned x, y, z; };
static thread_local wr_dim3 blockIdx, threadIdx, blockDim, gridDim;
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }
static inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long v) { unsigned long long o = *p; if (o == cmp) *p = v; return o; }
typedef int wr_stream_t;
typedef struct { double t; } wr_event_t;
// (WRHIP_HOSTSIM_NOEXEC=1: launches are skipped -- the host side of a frame (recording, staging, flush) can then be timed in the
// GPU-less container at the full sizes of the bench workloads: tools/host_profile.py)
static inline bool wr_hostsim_noexec() { static const bool v = getenv("WRHIP_HOSTSIM_NOEXEC") != nullptr; return v; }
#define WR_LAUNCH(kernel, grid, block, stream, ...)                     \
  do {                                                                  \
    if (wr_hostsim_noexec()) break;                                     \
    gridDim = wr_dim3{(unsigned)(grid), 1, 1};                          \
    blockDim = wr_dim3{(unsigned)(block), 1, 1};                        \
    for (unsigned _b = 0; _b < (unsigned)(grid); _b++)                  \
      for (unsigned _t = 0; _t < (unsigned)(block); _t++) {             \
        blockIdx = wr_dim3{_b, 0, 0};                                   \
        threadIdx = wr_dim3{_t, 0, 0};                                  \
        kernel(__VA_ARGS__);                                            \
      }                                                                 \
  } while (0)
namespace wrrt {
static inline int cu_count() { return 0; }
static inline bool init(int*, char* name, size_t n) { snprintf(name, n, "hostsim (CPU, tests only)"); return true; }
static inline void* dev_alloc(size_t n) { return calloc(1, n ? n : 1); }
static inline void* try_dev_alloc(size_t n) { const char* lim = getenv("WRHIP_HOSTSIM_ALLOC_LIMIT"); if (lim && n > (size_t)atoll(lim)) return nullptr; return calloc(1, n ? n : 1); }
static inline void dev_free(void* p) { free(p); }
static inline void* pinned_alloc(size_t n) { return malloc(n ? n : 1); }
static inline void pinned_free(void* p) { free(p); }
static inline void stream_create(wr_stream_t* s) { *s = 0; }
static inline void stream_destroy(wr_stream_t) {}
static inline void stream_sync(wr_stream_t) {}
static inline void h2d(void* d, const void* s, size_t n, wr_stream_t) { memcpy(d, s, n); }
static inline void d2h(void* d, const void* s, size_t n, wr_stream_t) { memcpy(d, s, n); }
static inline void d2d(void* d, const void* s, size_t n, wr_stream_t) { memmove(d, s, n); }
static inline void copy2d(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int, wr_stream_t) {
  for (size_t y = 0; y < h; y++) memmove((char*)d + y * dp, (const char*)s + y * sp, w);
}
static inline void memset8(void* d, int v, size_t n, wr_stream_t) { memset(d, v, n); }
static inline void event_create(wr_event_t* e) { e->t = 0; }
static inline void event_destroy(wr_event_t) {}
static inline void event_record(wr_event_t*, wr_stream_t) {}
static inline void event_sync(wr_event_t*) {}
static inline void stream_wait_event(wr_stream_t, wr_event_t*) {}
static inline void event_create_sync(wr_event_t* e) { e->t = 0; }
static inline float event_elapsed_ms(wr_event_t*, wr_event_t*) { return 0.f; }
}  // namespace wrrt
#else
// ---------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <tuple>
#include <new>
#include <type_traits>
#include <pthread.h>
typedef hipStream_t wr_stream_t;
typedef hipEvent_t wr_event_t;
#define WR_HIP_CHECK(expr)                                                          \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      fprintf(stderr, "libwrhip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), \
              __FILE__, __LINE__);                                                  \
      abort();                                                                      \
    }                                                                               \
  } while (0)
// ---------------------------------------------------------------------------
// Submit thread.  Every call that ENQUEUES on a stream (async copies, event records, stream waits, kernel launches) costs
// 3-8 us of host time in this runtime, and a frame issues half a dozen of them: on the thread that records the GL calls
// they were a third of a cfg2 frame's host time, with the GPU waiting.  They are handed to one helper thread instead, as
// closures in a ring (arguments captured by value), executed in program order; the recording thread only pays for the
// push.  Anything that waits for the device or frees what queued work may touch drains the ring first (wrq::drain, called by
// wrrt::stream_sync / event_sync / *_free / ...), so the order of HIP calls the runtime sees is the order the program made
// them in.  WRHIP_NO_SUBMIT_THREAD=1 runs every closure inline.
namespace wrq {
struct Cmd { void (*fn)(void*); alignas(16) unsigned char buf[496]; };
struct Queue {
  static constexpr uint64_t N = 256;
  Cmd ring[N];
  alignas(64) std::atomic<uint64_t> tail{0};        // next slot the producer writes
  alignas(64) std::atomic<uint64_t> head{0};        // next slot the consumer runs
  std::atomic<bool> sleeping{false};
  std::mutex m;
  std::condition_variable cv;
  bool enabled = false;
  int device = 0;
  int spin_limit = 500;          // pause iterations (a few microseconds) before the helper blocks: a spinning helper on the
                                 // recording thread's SMT sibling slows the staging copies down (measured: 18 -> 29 us per cfg2 frame)
  static std::atomic<bool>& forked() { static std::atomic<bool> f{false}; return f; }
  void start(int dev) {
    device = dev;
    enabled = getenv("WRHIP_NO_SUBMIT_THREAD") == nullptr;
    if (const char* e = getenv("WRHIP_SUBMIT_SPIN")) spin_limit = atoi(e);
    if (!enabled) return;
    pthread_atfork(nullptr, nullptr, +[] { forked().store(true); });      // (a fork()ed child has no helper: it runs inline)
    std::thread([this] { run(); }).detach();
  }
  void run() {
    (void)hipSetDevice(device);
    for (;;) {
      const uint64_t h = head.load(std::memory_order_relaxed);
      int spins = 0;
      while (tail.load(std::memory_order_acquire) == h) {
        if (++spins < spin_limit) { __builtin_ia32_pause(); continue; }
        std::unique_lock<std::mutex> lk(m);
        sleeping.store(true, std::memory_order_seq_cst);
        cv.wait(lk, [&] { return tail.load(std::memory_order_acquire) != h; });
        sleeping.store(false, std::memory_order_relaxed);
      }
      Cmd& c = ring[h % N];
      c.fn(c.buf);
      head.store(h + 1, std::memory_order_release);
    }
  }
  bool inline_mode() const { return !enabled || forked().load(std::memory_order_relaxed); }
  template <class F> void push(F&& f) {
    typedef typename std::decay<F>::type Fn;
    static_assert(sizeof(Fn) <= sizeof(Cmd::buf), "closure too large for a ring slot");
    static_assert(std::is_trivially_destructible<Fn>::value, "closures in the ring are not destroyed");
    if (inline_mode()) { f(); return; }
    const uint64_t t = tail.load(std::memory_order_relaxed);
    // (bounded spin, then yield: a helper that lost its core -- an oversubscribed box, a debugger -- must not pin this thread at 100 %)
    for (int spins = 0; t - head.load(std::memory_order_acquire) >= N; spins++) { if (spins < 4096) __builtin_ia32_pause(); else sched_yield(); }
    Cmd& c = ring[t % N];
    new (c.buf) Fn(static_cast<F&&>(f));
    c.fn = +[](void* p) { (*(Fn*)p)(); };
    tail.store(t + 1, std::memory_order_seq_cst);
    if (sleeping.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(m); cv.notify_one(); }
  }
  void drain() {
    if (inline_mode()) return;
    const uint64_t t = tail.load(std::memory_order_relaxed);
    for (int spins = 0; head.load(std::memory_order_acquire) != t; spins++) { if (spins < 4096) __builtin_ia32_pause(); else sched_yield(); }
  }
};
static inline Queue& q() { static Queue* p = new Queue(); return *p; }      // (never destroyed: the helper outlives static destruction)
template <class F> static inline void run(F&& f) { q().push(static_cast<F&&>(f)); }
static inline void drain() { q().drain(); }
}  // namespace wrq
#define WR_LAUNCH(kernel, grid, block, stream, ...)                                                    \
  do {                                                                                                 \
    auto wr_args_ = std::make_tuple(__VA_ARGS__);                                                      \
    const int wr_grid_ = (int)(grid), wr_block_ = (int)(block);                                        \
    hipStream_t wr_stream_ = (stream);                                                                 \
    wrq::run([wr_args_, wr_grid_, wr_block_, wr_stream_] {                                             \
      std::apply([&](auto... a) { hipLaunchKernelGGL(kernel, dim3(wr_grid_), dim3(wr_block_), 0, wr_stream_, a...); }, wr_args_); \
      WR_HIP_CHECK(hipGetLastError());                                                                 \
    });                                                                                                \
  } while (0)
namespace wrrt {
static inline int& cu_count_ref() { static int n = 0; return n; }
static inline int cu_count() { return cu_count_ref(); }
static inline bool init(int* device, char* name, size_t n) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return false;
  int dev = 0;
  const char* lr = getenv("LOCAL_RANK");
  if (lr) dev = atoi(lr) % count;
  WR_HIP_CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  WR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  snprintf(name, n, "%s (%s)", prop.name, prop.gcnArchName);
  cu_count_ref() = prop.multiProcessorCount;
  *device = dev;
  wrq::q().start(dev);
  return true;
}
static inline void* dev_alloc(size_t n) { void* p = nullptr; WR_HIP_CHECK(hipMalloc(&p, n ? n : 16)); return p; }
// storage a caller can be refused (textures, buffers): nullptr when HBM is exhausted -- the state tracker turns that into the
// sticky GL_OUT_OF_MEMORY swgl raises (gl.cc:1125-1134) instead of taking the process down
static inline void* try_dev_alloc(size_t n) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, n ? n : 16);
  if (e == hipSuccess) return p;
  (void)hipGetLastError();
  if (e != hipErrorOutOfMemory) { fprintf(stderr, "libwrhip: hipMalloc(%zu) failed: %s\n", n, hipGetErrorString(e)); abort(); }
  return nullptr;
}
static inline void dev_free(void* p) { wrq::drain(); if (p) WR_HIP_CHECK(hipFree(p)); }
static inline void* pinned_alloc(size_t n) { void* p = nullptr; WR_HIP_CHECK(hipHostMalloc(&p, n ? n : 16, hipHostMallocDefault)); return p; }
static inline void pinned_free(void* p) { wrq::drain(); if (p) WR_HIP_CHECK(hipHostFree(p)); }
static inline void stream_create(wr_stream_t* s) { WR_HIP_CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); }
static inline void stream_destroy(wr_stream_t s) { wrq::drain(); WR_HIP_CHECK(hipStreamDestroy(s)); }
static inline void stream_sync(wr_stream_t s) { wrq::drain(); WR_HIP_CHECK(hipStreamSynchronize(s)); }
// (h2d sources are the pinned staging ring or caller memory that a sync follows before it is reused)
static inline void h2d(void* d, const void* s, size_t n, wr_stream_t st) { wrq::run([=] { WR_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st)); }); }
static inline void d2h(void* d, const void* s, size_t n, wr_stream_t st) { wrq::run([=] { WR_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st)); }); }
static inline void d2d(void* d, const void* s, size_t n, wr_stream_t st) { wrq::run([=] { WR_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st)); }); }
// kind: 0 h2d, 1 d2h, 2 d2d
static inline void copy2d_now(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int kind, wr_stream_t st);
static inline void copy2d(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int kind, wr_stream_t st) {
  wrq::run([=] { copy2d_now(d, dp, s, sp, w, h, kind, st); });
}
static inline void copy2d_now(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int kind, wr_stream_t st) {
  hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (w == 0 || h == 0) return;
  // hipMemcpy2DAsync costs ~40 us of host time per call on this stack (6 us for
  // the 1-D form): use 1-D copies whenever the rows are contiguous or few.
  if (dp == w && sp == w) { WR_HIP_CHECK(hipMemcpyAsync(d, s, w * h, k, st)); return; }
  if (h <= 8) {
    for (size_t y = 0; y < h; y++) WR_HIP_CHECK(hipMemcpyAsync((char*)d + y * dp, (const char*)s + y * sp, w, k, st));
    return;
  }
  WR_HIP_CHECK(hipMemcpy2DAsync(d, dp, s, sp, w, h, k, st));
}
static inline void memset8(void* d, int v, size_t n, wr_stream_t st) { if (n) wrq::run([=] { WR_HIP_CHECK(hipMemsetAsync(d, v, n, st)); }); }
static inline void event_create(wr_event_t* e) { WR_HIP_CHECK(hipEventCreate(e)); }
static inline void event_destroy(wr_event_t e) { wrq::drain(); WR_HIP_CHECK(hipEventDestroy(e)); }
static inline void event_record(wr_event_t* e, wr_stream_t s) { const wr_event_t ev = *e; wrq::run([=] { WR_HIP_CHECK(hipEventRecord(ev, s)); }); }
static inline void event_sync(wr_event_t* e) { wrq::drain(); WR_HIP_CHECK(hipEventSynchronize(*e)); }
// ordering-only events (no timestamps) and cross-stream waits
static inline void event_create_sync(wr_event_t* e) { WR_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming)); }
static inline void stream_wait_event(wr_stream_t s, wr_event_t* e) { const wr_event_t ev = *e; wrq::run([=] { WR_HIP_CHECK(hipStreamWaitEvent(s, ev, 0)); }); }
static inline float event_elapsed_ms(wr_event_t* a, wr_event_t* b) { wrq::drain(); float ms = 0; WR_HIP_CHECK(hipEventElapsedTime(&ms, *a, *b)); return ms; }
}  // namespace wrrt
#endif
