// wrhip.hip -- libwrhip: the MI355X draw backend behind WebRender's GL-shaped
// C ABI (include/wrhip.h == the extern "C" block of swgl/src/swgl_fns.rs:23-322).
//
// Host side: a GL object store and state tracker (conceptually what
// swgl/src/gl.cc:665-2661 does, written fresh) that *records* work instead of
// executing it.  DrawElementsInstanced / Clear append draw packets to the
// pending list of their render target; everything pending is flushed together
// -- all targets in one vertex + bin + raster launch -- when a result is needed
// (a pending target gets sampled, read back, overwritten from the host, or
// Finish()).  See DESIGN.md §3-§4.
#include <vector>
#include <map>
#include <algorithm>
#include <time.h>
#include <chrono>
#include <thread>
#include <pthread.h>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>

#include "../../include/wrhip.h"
#include "wrhip_glenum.h"
#include "wrhip_types.h"
#include "wrhip_rt.h"
#include "wrhip_kernels.h"
#ifndef WRHIP_HOSTSIM
// (the raster kernel instantiations live in wrhip_inst.hip, one translation unit per group, so their device compiles run in
// parallel: here they are only launched)
#include "wrhip_inst.h"
#define WR_INST_DECLARE(K, SIG, ...) extern template __global__ void K<__VA_ARGS__> SIG;
WR_INST_ALL(WR_INST_DECLARE)
#undef WR_INST_DECLARE
#endif

namespace {
// Helper threads for the large host copies (a frame of 100 k prims stages ~24 MB of data-texture rows and instance arrays; one
// core copies that at ~45 GB/s, which had become the longest phase of such a frame).  run(f) calls f(part, parts) once per
// part, part 0 on the calling thread, and returns when all parts are done.  The helpers only ever touch the row ranges they
// are handed.  WRHIP_COPY_THREADS=0 turns them off.
struct CopyPool {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<const std::function<void(int, int)>*> job{nullptr};
  std::atomic<uint64_t> gen{0};
  std::atomic<int> left{0}, parts{1}, sleepers{0};
  int spin_iters = 0;          // how long a helper polls for the next job before it blocks (WRHIP_COPY_SPIN_US; 0: it blocks at once)
  static std::atomic<bool>& forked() { static std::atomic<bool> f{false}; return f; }
  CopyPool() {
    // a fork()ed child inherits this object but not the helper threads: it copies on its own
    pthread_atfork(nullptr, nullptr, +[] { forked().store(true); });
    const char* e = getenv("WRHIP_COPY_THREADS");
    int want = e ? atoi(e) : 7;      // (cfg5, 24 MB staged per frame: 3 helpers 1.79-1.97 k frames/s, 7: 2.17-2.19 k, 15: 1.83-2.03 k; cfg2 unchanged)
    const unsigned hc = std::thread::hardware_concurrency();
    if (hc && (int)hc - 1 < want) want = (int)hc - 1;
    n = want > 0 ? want : 0;
    const char* sp = getenv("WRHIP_COPY_SPIN_US");
    spin_iters = (sp ? atoi(sp) : 0) * 25;           // (a pause is ~40 ns; off by default: no gain measured, tools/r3_copy_ab.sh)
    for (int i = 0; i < n; i++) std::thread([this, i] { worker(i + 1); }).detach();
  }
  void worker(int part) {
    uint64_t seen = 0;
    for (;;) {
      int spins = 0;
      auto posted = [&] { const uint64_t g = gen.load(std::memory_order_seq_cst); return g != seen && !(g & 1); };
      while (!posted()) {
        if (++spins < spin_iters) { __builtin_ia32_pause(); continue; }
        std::unique_lock<std::mutex> lk(m);
        sleepers.fetch_add(1, std::memory_order_seq_cst);
        cv.wait(lk, posted);
        sleepers.fetch_sub(1, std::memory_order_relaxed);
      }
      // (only the helpers that take a part are waited for: one that sleeps through a two-part copy costs the caller nothing;
      // the job a helper reads must be the one of the generation it saw -- a bystander may look up while the next job is
      // being posted)
      uint64_t g; int np; const std::function<void(int, int)>* j;
      do {
        g = gen.load(std::memory_order_seq_cst);
        np = parts.load(std::memory_order_seq_cst); j = job.load(std::memory_order_seq_cst);
      } while ((g & 1) || gen.load(std::memory_order_seq_cst) != g);
      if (g == seen) continue;
      seen = g;
      if (part < np) { (*j)(part, np); left.fetch_sub(1, std::memory_order_release); }
    }
  }
  // f(part, parts) once per part, part 0 on the calling thread; `max_parts` bounds the fan-out of small copies
  void run(const std::function<void(int, int)>& f, int max_parts = 1 << 20) {
    if (!n || forked().load(std::memory_order_relaxed)) { f(0, 1); return; }
    const int np = std::min(n + 1, std::max(1, max_parts));
    if (np == 1) { f(0, 1); return; }
    // (a sequence lock: the generation is odd while the job is being written; helpers act on an even generation whose fields
    // they read between two equal looks at it)
    gen.fetch_add(1, std::memory_order_seq_cst);
    job.store(&f, std::memory_order_seq_cst);
    parts.store(np, std::memory_order_seq_cst);
    left.store(np - 1, std::memory_order_seq_cst);
    gen.fetch_add(1, std::memory_order_seq_cst);
    if (sleepers.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lk(m); cv.notify_all(); }
    f(0, np);
    while (left.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
  }
};
CopyPool& copy_pool() { static CopyPool* p = new CopyPool(); return *p; }     // (never destroyed: the helpers outlive static destruction)
constexpr size_t PARALLEL_COPY_MIN = 1u << 20;
// (texture uploads of WRHIP_COPY_SPLIT_MIN bytes and more could go in two parts -- the caller and one helper; off by default:
// on a 0.75 MB cfg2 frame the hand-over costs what the second core saves)
static size_t split_copy_min() { static const size_t v = getenv("WRHIP_COPY_SPLIT_MIN") ? (size_t)atoll(getenv("WRHIP_COPY_SPLIT_MIN")) : PARALLEL_COPY_MIN; return v; }
// Copy into the pinned staging ring with non-temporal stores: the ring is 96 MiB of memory the CPU never reads back (the
// DMA engine does), so ordinary stores first fetch every destination line (read-for-ownership) and then evict useful
// lines to keep it.  Streaming stores do neither; small copies keep memcpy.
#if !defined(WRHIP_HOSTSIM) && (defined(__x86_64__) || defined(__SSE2__))
#include <emmintrin.h>
static inline void stage_copy(void* dst_, const void* src_, size_t n) {
  uint8_t* d = (uint8_t*)dst_; const uint8_t* s = (const uint8_t*)src_;
  static const bool plain = getenv("WRHIP_NO_STREAM_COPY") != nullptr;       // (A/B measurements)
  if (n < 2048 || plain) { memcpy(d, s, n); return; }
  const size_t head = (64 - ((uintptr_t)d & 63)) & 63;
  if (head) { memcpy(d, s, head); d += head; s += head; n -= head; }
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 16));
    const __m128i c = _mm_loadu_si128((const __m128i*)(s + i + 32)), e = _mm_loadu_si128((const __m128i*)(s + i + 48));
    _mm_stream_si128((__m128i*)(d + i), a); _mm_stream_si128((__m128i*)(d + i + 16), b);
    _mm_stream_si128((__m128i*)(d + i + 32), c); _mm_stream_si128((__m128i*)(d + i + 48), e);
  }
  if (i < n) memcpy(d + i, s + i, n - i);
  _mm_sfence();
}
#else
static inline void stage_copy(void* dst, const void* src, size_t n) { memcpy(dst, src, n); }
#endif
void big_memcpy(void* dst, const void* src, size_t n) {
  if (n < PARALLEL_COPY_MIN) { memcpy(dst, src, n); return; }
  copy_pool().run([&](int part, int parts) {
    const size_t a = (n / 64 * part / parts) * 64, b = part + 1 == parts ? n : (n / 64 * (part + 1) / parts) * 64;
    memcpy((uint8_t*)dst + a, (const uint8_t*)src + a, b - a);
  });
}

// host-side phase timers (WrhipStats::host_*_ns).  Exclusive: a timer started inside another one (an upload that has to
// flush, a flush that has to wait) pauses the outer one, so the phases add up to the time spent in the library.
struct HostTimer {
  uint64_t* acc;
  HostTimer* outer;
  std::chrono::steady_clock::time_point t0;
  static HostTimer*& current() { static thread_local HostTimer* cur = nullptr; return cur; }
  static uint64_t since(std::chrono::steady_clock::time_point t) {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count();
  }
  explicit HostTimer(uint64_t* a) : acc(a), outer(current()), t0(std::chrono::steady_clock::now()) {
    if (outer && outer->acc) *outer->acc += since(outer->t0);
    current() = this;
  }
  ~HostTimer() {
    if (acc) *acc += since(t0);
    current() = outer;
    if (outer) outer->t0 = std::chrono::steady_clock::now();
  }
};

// ---------------------------------------------------------------------------
// Object store with swgl's id policy (first free slot >= 1; gl.cc:674-745) so
// object names match the reference's for identical call streams.
template <typename O>
struct ObjectStore {
  std::vector<O*> objects;
  size_t first_free = 1;
  ~ObjectStore() { for (O* o : objects) delete o; }
  size_t insert() {
    size_t i = first_free;
    while (i < objects.size() && objects[i]) i++;
    first_free = i;
    if (i >= objects.size()) objects.resize(i + 1, nullptr);
    objects[i] = new O();
    return i;
  }
  O& operator[](size_t i) {
    if (i >= objects.size()) objects.resize(i + 1, nullptr);
    if (!objects[i]) objects[i] = new O();
    return *objects[i];
  }
  O* find(size_t i) const { return i < objects.size() ? objects[i] : nullptr; }
  bool erase(size_t i) {
    if (i < objects.size() && objects[i]) {
      delete objects[i];
      objects[i] = nullptr;
      if (i < first_free) first_free = i;
      return true;
    }
    return false;
  }
};

struct Query { uint64_t value = 0; int slot = -1; };   // slot: WrUnsupportedCounters::samples entry a GL_SAMPLES_PASSED query counts into

struct Buffer {
  uint8_t* buf = nullptr;
  size_t size = 0, capacity = 0;
  bool allocate(size_t n) {
    if (n == size) return true;
    if (n <= capacity) { size = n; return true; }
    uint8_t* nb = (uint8_t*)realloc(buf, n);
    if (!nb) { free(buf); buf = nullptr; size = capacity = 0; return false; }
    buf = nb; size = capacity = n;
    return true;
  }
  ~Buffer() { free(buf); }
};

struct Texture {
  GLenum internal_format = 0;
  int width = 0, height = 0;
  int bpp = 0;
  int stride = 0;            // bytes; 4-byte aligned like gl.cc:268
  void* dptr = nullptr;      // HBM storage
  size_t dsize = 0;
  GLenum min_filter = GL_NEAREST, mag_filter = GL_LINEAR;
  int offx = 0, offy = 0;
  int locked = 0;
  // host mirror handed out by GetColorBuffer / GetResourceBuffer
  uint8_t* hmirror = nullptr;
  size_t hmirror_size = 0;
  // externally supplied backing store (SetTextureBuffer / InitDefaultFramebuffer(buf))
  void* ext_buf = nullptr;
  int ext_stride = 0;
  // depth attachment state (gl.cc:394-420 CLEARED flag; rasterize.h:962)
  bool depth_cleared = false;
  bool depth_materialized = false;   // dptr holds per-pixel depth (u32, width x height) that a pending / later target must load
  uint32_t depth_value = 0xFFFFFF;
  GLuint depth_owner = 0;            // colour target whose pending work last used this depth attachment
  // hazards against the pending list
  int own_y0 = 0, own_y1 = 0;   // multi-GPU: owned pixel rows (0,0 = all)
  // ... or none at all (y1 < y0, or a range that misses the texture as it is NOW: the answer follows the storage, a call made
  // before TexStorage or a later resize does not freeze it): draws and clears into such a target are dropped when recorded
  bool own_none() const { return own_y1 < own_y0 || (own_y1 > own_y0 && height > 0 && (own_y0 >= height || own_y1 <= 0)); }
  bool pending_read = false, pending_write = false;
  int pending_target = -1;   // index into Context::work when pending_write
  bool tail_ref = false;     // read or written by the deferred last raster level (Context::Tail)
  // RGBA32I data textures: does any transform id in them carry TRANSFORM_NON_AXIS_ALIGNED (bit 23,
  // transform.glsl:22-29)?  Read as sPrimitiveHeadersI (2 texels per prim, id in .z of the first) and as
  // sGpuBufferI (ps_quad header, id in .x); maintained on upload.  A draw is only declared
  // rectangle-only (WR_DF_SIMPLE) when the header texture it binds is clean.
  bool complex_ids_headers = false, complex_ids_gpubuf = false;
  // Uploads of the open batch (Context::upload_batch).  up_batch: the batch that holds rows for this texture; view_batch / view_off: a
  // WHOLE-texture upload of that batch sits at staging offset view_off in the texture's own layout (row bytes == stride), so a draw of the
  // batch's flush can read it in the staging mirror and does not wait for the scatter (fused scatter)
  uint64_t up_batch = 0, view_batch = 0;
  size_t view_off = 0;
  bool has_storage() const { return dptr != nullptr; }
};

struct VertexAttrib {
  size_t size = 0; GLenum type = 0; bool normalized = false; GLsizei stride = 0; GLuint offset = 0;
  bool enabled = false; GLuint divisor = 0; GLuint vertex_buffer = 0;
};
#define MAX_ATTRIBS 17
#define NULL_ATTRIB 16
struct VertexArray {
  VertexAttrib attribs[MAX_ATTRIBS];
  GLuint element_array_buffer_binding = 0;
};
struct Framebuffer { GLuint color_attachment = 0, depth_attachment = 0; };
struct Renderbuffer { GLuint texture = 0; };
struct Shader { GLenum type = 0; int kind = WR_SH_NONE; char name[96] = ""; };

// Static description of the programs the backend implements.
struct ShaderInfo {
  const char* key; int kind;
  const char* attribs[WR_MAX_ATTRIBS + 2];   // [0] = per-vertex aPosition, then instance attributes in shader order
  unsigned samplers;         // bit s set: program declares the sampler of slot s
  bool rect = false;         // a TEXTURE_RECT key: sColor0-2 are sampler2DRect (bound through GL_TEXTURE_RECTANGLE, unnormalised uv)
};
#define S(x) (1u << (x))
const unsigned PRIM_SAMPLERS = S(WR_S_COLOR0) | S(WR_S_GPU_CACHE) | S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS) |
                               S(WR_S_PRIM_HEADERS_F) | S(WR_S_PRIM_HEADERS_I) | S(WR_S_CLIP_MASK);
const unsigned PRIM_SAMPLERS_NO_COLOR = PRIM_SAMPLERS & ~S(WR_S_COLOR0);
const unsigned CACHED_GRADIENT_SAMPLERS = S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I) | S(WR_S_GPU_CACHE) | S(WR_S_RENDER_TASKS);
// (the sampler set of every key is the set of uniforms the reference's generated program answers GetUniformLocation for:
// tests/test_abi.py::test_uniform_locations_exist_where_the_reference_has_them compares them key by key with the oracle)
const ShaderInfo SHADERS[] = {
    {"ps_quad_textured", WR_SH_PS_QUAD_TEXTURED, {"aPosition", "aData"},
     S(WR_S_COLOR0) | S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS) | S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I)},
    {"brush_solid", WR_SH_BRUSH_SOLID, {"aPosition", "aData"}, PRIM_SAMPLERS_NO_COLOR},
    {"brush_solid ALPHA_PASS", WR_SH_BRUSH_SOLID_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS_NO_COLOR},
    {"brush_image TEXTURE_2D", WR_SH_BRUSH_IMAGE, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_image ALPHA_PASS,TEXTURE_2D", WR_SH_BRUSH_IMAGE_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_image ANTIALIASING,REPETITION,TEXTURE_2D", WR_SH_BRUSH_IMAGE_REPEAT, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_image ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D", WR_SH_BRUSH_IMAGE_REPEAT_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS},
    // (ALPHA_PASS + DUAL_SOURCE_BLENDING: what BlendMode::SubpixelDualSource / MultiplyDualSource batches are drawn with,
    // shade.rs:462-467 -- main() only, a second output colour; axis-aligned prims without swgl_antiAlias)
    {"brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D", WR_SH_BRUSH_IMAGE_DUAL, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_2D", WR_SH_BRUSH_IMAGE_REPEAT_DUAL, {"aPosition", "aData"}, PRIM_SAMPLERS},
    // (the ADVANCED_BLEND keys: the ALPHA_PASS programs with `layout(blend_support_all_equations) out`, shared.glsl:86-88 --
    // what BlendMode::Advanced batches are drawn with, shade.rs:440-468)
    {"brush_image ADVANCED_BLEND,ALPHA_PASS,TEXTURE_2D", WR_SH_BRUSH_IMAGE_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_image ADVANCED_BLEND,ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D", WR_SH_BRUSH_IMAGE_REPEAT_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_linear_gradient", WR_SH_BRUSH_LINEAR_GRADIENT, {"aPosition", "aData"}, PRIM_SAMPLERS_NO_COLOR | S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I)},
    {"brush_linear_gradient ALPHA_PASS", WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS_NO_COLOR | S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I)},
    {"ps_quad_mask", WR_SH_PS_QUAD_MASK, {"aPosition", "aData", "aClipData"},
     S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS) | S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I)},
    {"ps_quad_mask FAST_PATH", WR_SH_PS_QUAD_MASK_FAST, {"aPosition", "aData", "aClipData"},
     S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS) | S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I)},
    {"ps_quad_radial_gradient", WR_SH_PS_QUAD_RADIAL_GRADIENT, {"aPosition", "aData"},
     S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS) | S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I)},
    {"ps_quad_conic_gradient", WR_SH_PS_QUAD_CONIC_GRADIENT, {"aPosition", "aData"},
     S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS) | S(WR_S_GPU_BUFFER_F) | S(WR_S_GPU_BUFFER_I)},
    // (the split polygons of a preserve-3d context, shade.rs:1226; PRIM_INSTANCES layout)
    {"ps_split_composite", WR_SH_PS_SPLIT_COMPOSITE, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_opacity", WR_SH_BRUSH_OPACITY, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_opacity ANTIALIASING", WR_SH_BRUSH_OPACITY, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_opacity ALPHA_PASS", WR_SH_BRUSH_OPACITY_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_opacity ALPHA_PASS,ANTIALIASING", WR_SH_BRUSH_OPACITY_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_blend", WR_SH_BRUSH_BLEND, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"brush_blend ALPHA_PASS", WR_SH_BRUSH_BLEND_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS},
    // (video: planar / semi-planar YUV frames, one R8 / RG8 texture per plane: batch.rs:2301-2390, shade.rs:1012-1040)
    {"brush_yuv_image TEXTURE_2D,YUV", WR_SH_BRUSH_YUV, {"aPosition", "aData"}, PRIM_SAMPLERS | S(WR_S_COLOR1) | S(WR_S_COLOR2)},
    {"brush_yuv_image ALPHA_PASS,TEXTURE_2D,YUV", WR_SH_BRUSH_YUV_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS | S(WR_S_COLOR1) | S(WR_S_COLOR2)},
    {"brush_mix_blend", WR_SH_BRUSH_MIX_BLEND, {"aPosition", "aData"}, PRIM_SAMPLERS | S(WR_S_COLOR1)},
    {"brush_mix_blend ALPHA_PASS", WR_SH_BRUSH_MIX_BLEND_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS | S(WR_S_COLOR1)},
    {"composite TEXTURE_2D", WR_SH_COMPOSITE,
     {"aPosition", "aDeviceRect", "aDeviceClipRect", "aColor", "aParams", "aUvRect0", "aFlip"}, S(WR_S_COLOR0)},
    {"composite FAST_PATH,TEXTURE_2D", WR_SH_COMPOSITE_FAST,
     {"aPosition", "aDeviceRect", "aDeviceClipRect", "aColor", "aParams", "aUvRect0", "aFlip"}, S(WR_S_COLOR0)},
    {"composite TEXTURE_2D,YUV", WR_SH_COMPOSITE_YUV,
     {"aPosition", "aDeviceRect", "aDeviceClipRect", "aColor", "aParams", "aUvRect0", "aUvRect1", "aUvRect2", "aFlip"},
     S(WR_S_COLOR0) | S(WR_S_COLOR1) | S(WR_S_COLOR2)},
    {"ps_clear", WR_SH_PS_CLEAR, {"aPosition", "aRect", "aColor"}, 0},
    {"ps_text_run ALPHA_PASS,TEXTURE_2D", WR_SH_PS_TEXT_RUN, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D", WR_SH_PS_TEXT_RUN_DUAL, {"aPosition", "aData"}, PRIM_SAMPLERS},
    // (glyphs rasterised under the run's transform -- rotated / scaled text in screen raster space, shader_features.rs:222-226)
    {"ps_text_run ALPHA_PASS,GLYPH_TRANSFORM,TEXTURE_2D", WR_SH_PS_TEXT_RUN_GT, {"aPosition", "aData"}, PRIM_SAMPLERS},
    {"ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,GLYPH_TRANSFORM,TEXTURE_2D", WR_SH_PS_TEXT_RUN_DUAL_GT, {"aPosition", "aData"}, PRIM_SAMPLERS},
#define CLIP_RECT_ATTRIBS                                                                                             \
  {"aPosition", "aClipDeviceArea", "aClipOrigins", "aDevicePixelScale", "aTransformIds", "aClipLocalPos",           \
   "aClipLocalRect", "aClipMode", "aClipRect_TL", "aClipRadii_TL", "aClipRect_TR", "aClipRadii_TR", "aClipRect_BL", \
   "aClipRadii_BL", "aClipRect_BR", "aClipRadii_BR"}
    {"cs_clip_rectangle", WR_SH_CS_CLIP_RECT, CLIP_RECT_ATTRIBS, S(WR_S_GPU_CACHE) | S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS)},
    {"cs_clip_rectangle FAST_PATH", WR_SH_CS_CLIP_RECT_FAST, CLIP_RECT_ATTRIBS, S(WR_S_GPU_CACHE) | S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS)},
    {"cs_clip_box_shadow TEXTURE_2D", WR_SH_CS_CLIP_BOX_SHADOW,
     {"aPosition", "aClipDeviceArea", "aClipOrigins", "aDevicePixelScale", "aTransformIds", "aClipDataResourceAddress",
      "aClipSrcRectSize", "aClipMode", "aStretchMode", "aClipDestRect"},
     S(WR_S_COLOR0) | S(WR_S_GPU_CACHE) | S(WR_S_TRANSFORMS) | S(WR_S_RENDER_TASKS)},
    {"cs_scale TEXTURE_2D", WR_SH_CS_SCALE, {"aPosition", "aScaleTargetRect", "aScaleSourceRect", "aSourceRectType"},
     S(WR_S_COLOR0)},
    // (one node of a CSS / SVG filter chain, and of an SVG filter graph: render_target.rs:901-1170; blend off, colour targets)
    {"cs_svg_filter", WR_SH_CS_SVG_FILTER,
     {"aPosition", "aFilterRenderTaskAddress", "aFilterInput1TaskAddress", "aFilterInput2TaskAddress", "aFilterKind", "aFilterInputCount",
      "aFilterGenericInt", "aFilterExtraDataAddress"},
     PRIM_SAMPLERS | S(WR_S_COLOR1)},
    {"cs_svg_filter_node", WR_SH_CS_SVG_FILTER_NODE,
     {"aPosition", "aFilterTargetRect", "aFilterInput1ContentScaleAndOffset", "aFilterInput2ContentScaleAndOffset", "aFilterInput1TaskAddress",
      "aFilterInput2TaskAddress", "aFilterKind", "aFilterInputCount", "aFilterExtraDataAddress"},
     PRIM_SAMPLERS | S(WR_S_COLOR1)},
    {"ps_copy", WR_SH_PS_COPY, {"aPosition", "a_src_rect", "a_dst_rect", "a_dst_texture_size"}, S(WR_S_COLOR0)},
    {"cs_border_solid", WR_SH_CS_BORDER_SOLID,
     {"aPosition", "aTaskOrigin", "aRect", "aColor0", "aColor1", "aFlags", "aWidths", "aRadii", "aClipParams1", "aClipParams2"}, 0},
    {"cs_border_segment", WR_SH_CS_BORDER_SEGMENT,
     {"aPosition", "aTaskOrigin", "aRect", "aColor0", "aColor1", "aFlags", "aWidths", "aRadii", "aClipParams1", "aClipParams2"}, 0},
    {"cs_fast_linear_gradient", WR_SH_CS_FAST_LINEAR_GRADIENT, {"aPosition", "aTaskRect", "aColor0", "aColor1", "aAxisSelect"}, 0},
    {"cs_line_decoration", WR_SH_CS_LINE_DECORATION,
     {"aPosition", "aTaskRect", "aLocalSize", "aWavyLineThickness", "aStyle", "aAxisSelect"}, 0},
    {"cs_linear_gradient", WR_SH_CS_LINEAR_GRADIENT,
     {"aPosition", "aTaskRect", "aStartPoint", "aEndPoint", "aScale", "aExtendMode", "aGradientStopsAddress"}, CACHED_GRADIENT_SAMPLERS},
    {"cs_radial_gradient", WR_SH_CS_RADIAL_GRADIENT,
     {"aPosition", "aTaskRect", "aCenter", "aScale", "aStartRadius", "aEndRadius", "aXYRatio", "aExtendMode", "aGradientStopsAddress"},
     CACHED_GRADIENT_SAMPLERS},
    {"cs_conic_gradient", WR_SH_CS_CONIC_GRADIENT,
     {"aPosition", "aTaskRect", "aCenter", "aScale", "aStartOffset", "aEndOffset", "aAngle", "aExtendMode", "aGradientStopsAddress"},
     CACHED_GRADIENT_SAMPLERS},
    {"cs_blur ALPHA_TARGET", WR_SH_CS_BLUR_ALPHA,
     {"aPosition", "aBlurRenderTaskAddress", "aBlurSourceTaskAddress", "aBlurDirection", "aBlurParams"}, PRIM_SAMPLERS},
    {"cs_blur COLOR_TARGET", WR_SH_CS_BLUR_COLOR,
     {"aPosition", "aBlurRenderTaskAddress", "aBlurSourceTaskAddress", "aBlurDirection", "aBlurParams"}, PRIM_SAMPLERS},
    // The TEXTURE_RECT keys (shader_features.rs:136-138, 181-198; swgl compiles them under ShaderFeatureFlags::GL): the same programs with
    // sampler2DRect sColor0-2 -- unnormalised uv (samplerScale = 1, texture.h:440-443), texture_size = 1 in the vertex stages --
    // bound through GL_TEXTURE_RECTANGLE: what external / IOSurface images and native compositor surfaces are drawn with.
    {"brush_image TEXTURE_RECT", WR_SH_BRUSH_IMAGE, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_image ALPHA_PASS,TEXTURE_RECT", WR_SH_BRUSH_IMAGE_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_image ANTIALIASING,REPETITION,TEXTURE_RECT", WR_SH_BRUSH_IMAGE_REPEAT, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_image ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_RECT", WR_SH_BRUSH_IMAGE_REPEAT_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_RECT", WR_SH_BRUSH_IMAGE_DUAL, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_RECT", WR_SH_BRUSH_IMAGE_REPEAT_DUAL, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_image ADVANCED_BLEND,ALPHA_PASS,TEXTURE_RECT", WR_SH_BRUSH_IMAGE_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_image ADVANCED_BLEND,ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_RECT", WR_SH_BRUSH_IMAGE_REPEAT_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS, true},
    {"brush_yuv_image TEXTURE_RECT,YUV", WR_SH_BRUSH_YUV, {"aPosition", "aData"}, PRIM_SAMPLERS | S(WR_S_COLOR1) | S(WR_S_COLOR2), true},
    {"brush_yuv_image ALPHA_PASS,TEXTURE_RECT,YUV", WR_SH_BRUSH_YUV_ALPHA, {"aPosition", "aData"}, PRIM_SAMPLERS | S(WR_S_COLOR1) | S(WR_S_COLOR2), true},
    {"composite TEXTURE_RECT", WR_SH_COMPOSITE,
     {"aPosition", "aDeviceRect", "aDeviceClipRect", "aColor", "aParams", "aUvRect0", "aFlip"}, S(WR_S_COLOR0), true},
    {"composite FAST_PATH,TEXTURE_RECT", WR_SH_COMPOSITE_FAST,
     {"aPosition", "aDeviceRect", "aDeviceClipRect", "aColor", "aParams", "aUvRect0", "aFlip"}, S(WR_S_COLOR0), true},
    {"composite TEXTURE_RECT,YUV", WR_SH_COMPOSITE_YUV,
     {"aPosition", "aDeviceRect", "aDeviceClipRect", "aColor", "aParams", "aUvRect0", "aUvRect1", "aUvRect2", "aFlip"},
     S(WR_S_COLOR0) | S(WR_S_COLOR1) | S(WR_S_COLOR2), true},
    {"cs_scale TEXTURE_RECT", WR_SH_CS_SCALE, {"aPosition", "aScaleTargetRect", "aScaleSourceRect", "aSourceRectType"},
     S(WR_S_COLOR0), true},
};
#undef S
const char* const SAMPLER_NAMES[WR_MAX_TEX] = {
    "sColor0", "sColor1", "sColor2", "sGpuCache", "sTransformPalette", "sRenderTasks", "sDither",
    "sPrimitiveHeadersF", "sPrimitiveHeadersI", "sClipMask", "sGpuBufferF", "sGpuBufferI"};
const int UNIFORM_TRANSFORM = WR_MAX_TEX + 1;   // sampler s -> uniform index s+1

struct Program {
  const ShaderInfo* info = nullptr;
  char name[96] = "";            // the key ShaderSourceByName sent (kept for programs without an implementation)
  bool refused_once = false;     // the draw-time refusal of an unimplemented program is printed once
  bool linked = false, deleted = false;
  int attrib_loc[WR_MAX_ATTRIBS + 2];
  int sampler_unit[WR_MAX_TEX];
  float transform[16];
  Program() {
    for (int& a : attrib_loc) a = NULL_ATTRIB;
    for (int& s : sampler_unit) s = 0;
    memset(transform, 0, sizeof(transform));
  }
};

// One render target with work pending in the current batch of deferred draws.
struct TargetWork {
  GLuint tex = 0;
  GLuint depth_tex = 0;
  std::vector<WrDrawDesc> draws;   // submission order; inst_offset relative to `inst`
  std::vector<uint8_t> inst;       // instance bytes snapshotted at draw time
  std::vector<GLuint> reads;       // textures sampled by these draws
  std::vector<GLuint> rreads;      // ... the subset the RASTER stage can read (colour / mask samplers, filter tables, gradient
                                   // stops); the data textures of the vertex stage are consumed by the setup kernel of the same flush
  int prims = 0;
  int level = 0;                   // dependency depth inside the pending batch: samples targets of lower levels only
  // depth: the value the attachment held when this target's first depth-using op was recorded (the renderbuffer is shared by
  // same-size targets and direct clears change it without recording anything), and whether the caller still expects the
  // depth these draws leave behind (no InvalidateFramebuffer / full clear by another user since)
  uint32_t init_depth = 0xFFFFFF;
  bool init_depth_set = false, depth_live = false;
  // forwarded composite (set per flush by plan_forwarding): this target's finished pixels are also stored into `fwd_tex`
  GLuint fwd_tex = 0; int fwd_dx = 0, fwd_y0 = 0, fwd_ys = 1, fwd_clip[4] = {0, 0, 0, 0};
  bool forwarded_away = false;     // every draw of this target was turned into write-throughs of its sources: nothing left to rasterise
  // back to the state of a new object, the vectors keeping their storage: a frame's 72 tiles x (three 1 KB draw packets + 33 KB of
  // instance bytes) no longer cost an allocation, a growth copy and fresh pages each (cfg5: a third of the recording time)
  void recycle() {
    std::vector<WrDrawDesc> d; d.swap(draws); d.clear();
    std::vector<uint8_t> i; i.swap(inst); i.clear();
    std::vector<GLuint> r; r.swap(reads); r.clear();
    std::vector<GLuint> rr; rr.swap(rreads); rr.clear();
    *this = TargetWork();
    draws.swap(d); inst.swap(i); reads.swap(r); rreads.swap(rr);
  }
};

const size_t MAX_TEXTURE_UNITS = 16;

struct Context {
  int32_t references = 1;
  ObjectStore<Query> queries;
  ObjectStore<Buffer> buffers;
  ObjectStore<Texture> textures;
  ObjectStore<VertexArray> vertex_arrays;
  ObjectStore<Framebuffer> framebuffers;
  ObjectStore<Renderbuffer> renderbuffers;
  ObjectStore<Shader> shaders;
  ObjectStore<Program> programs;

  GLenum last_error = GL_NO_ERROR;
  int viewport[4] = {0, 0, 0, 0};  // x0,y0,x1,y1
  bool blend = false;
  GLenum blendfunc_srgb = GL_ONE, blendfunc_drgb = GL_ZERO, blendfunc_sa = GL_ONE, blendfunc_da = GL_ZERO;
  GLenum blend_equation = GL_FUNC_ADD;
  uint32_t blendcolor[2] = {0, 0};
  int blend_key = WR_BLEND_NONE;
  bool depthtest = false, depthmask = true;
  GLenum depthfunc = GL_LESS;
  bool scissortest = false;
  int scissor[4] = {0, 0, 0, 0};
  float clearcolor[4] = {0, 0, 0, 0};
  double cleardepth = 1;
  int unpack_row_length = 0;

  struct TextureUnit { GLuint texture_2d_binding = 0, texture_rectangle_binding = 0; };
  TextureUnit texture_units[MAX_TEXTURE_UNITS];
  int active_texture_unit = 0;
  GLuint current_program = 0, current_vertex_array = 0;
  GLuint pixel_pack_buffer_binding = 0, pixel_unpack_buffer_binding = 0, array_buffer_binding = 0;
  GLuint time_elapsed_query = 0, samples_passed_query = 0;
  GLuint renderbuffer_binding = 0, draw_framebuffer_binding = 0, read_framebuffer_binding = 0;
  GLuint unknown_binding = 0;

  // ---- device side ------------------------------------------------------
  wr_stream_t stream;
  // deferred work
  std::vector<TargetWork> work;       // render targets with pending draws
  std::vector<TargetWork> spare;      // flushed ones, emptied, their vectors' storage kept for the next frame (TargetWork::recycle)
  std::vector<GLuint> referenced;     // textures with pending_read/pending_write set
  // Host->HBM traffic is batched: texture uploads (TexSubImage2D ...) and the
  // per-flush frame arena are bump-allocated from one pinned staging ring and
  // moved with ONE async DMA per flush into the device mirror `dupload` at the
  // same offsets; a scatter kernel then writes texture rows to their
  // destinations.  (Individual hipMemcpy*Async calls cost 6-40 us of host time
  // and ~15 us of GPU-side latency each on this stack.)
  struct UploadSeg { uint64_t src_off; void* dst; uint32_t dst_stride, row_bytes, rows, pad; };
  std::vector<UploadSeg> useg;
  size_t upload_begin = 0;        // staging offset where the pending batch starts
  bool upload_open = false;
  uint64_t lap_ns[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t lap_n = 0;      // WRHIP_DEBUG_LAPS=1: where flush_work's time goes (printed at context destruction)
  uint64_t upload_batch = 1;      // id of the open (or next) batch: Texture::up_batch / view_batch compare with it
  bool scatter_first = false;     // a recorded draw's setup stage reads a data texture the open batch writes PARTLY: the scatter has to precede it
  struct PendingScatter { size_t seg_off = 0; int nseg = 0, parts = 0; uint64_t bytes = 0; bool valid = false; } ps;   // a batch's scatter handed to the flush's setup-carrying launch
  bool fuse_scatter = true;       // WRHIP_NO_FUSE_SCATTER=1: the scatter keeps its own launch
  uint8_t* dupload = nullptr;     // HBM mirror of the staging ring
  // Per-flush scratch, two sets used alternately (flush_seq & 1): the deferred tail of flush k
  // still reads set k & 1 while the setup stage of flush k+1 fills the other one.
  struct Scratch {
    WrPrim* prims = nullptr; size_t prims_cap = 0;
    WrRec* recs = nullptr;
    WrAux* aux = nullptr;
    float* vtab = nullptr; size_t vtab_cap = 0;   // per-row v tables of nearest-fast textured prims (WrDrawDesc::vtab_base)
    float* qtab = nullptr; size_t qtab_cap = 0;   // row tables of general quads (WrTargetDesc::qtab), handed out by the setup stage
    unsigned long long* masks = nullptr; size_t masks_cap = 0;
    unsigned* bin_ctr = nullptr; size_t bin_ctr_cap = 0;   // per-bin arrival counters of thin launches that give a bin several workgroups (WrTargetDesc::bin_ctr), zero between launches
    // mask-row store (WrMaskSlot): allocation word, slot list, row bytes
    unsigned long long* mr_ctl = nullptr; WrMaskSlot* mr_slots = nullptr; size_t mr_slots_cap = 0;
    uint8_t* mr_store = nullptr; size_t mr_store_cap = 0;
    unsigned long long mr_seen = 0;      // profiling: wr_mask_rows_kernel's byte count at the previous read-back
    uint32_t* flat = nullptr; size_t flat_cap = 0;   // flattened-depth-row tables (WrTargetDesc::flat_rows), height + 1 words per target that needs one
  } scratch[2];
  int64_t flush_seq = 0;
  int64_t held_seq = 0;            // flush_seq at the last WrhipFlushHeld
  int fb_writes_since_held = 0;    // flushes since the last WrhipFlushHeld that wrote the default framebuffer's texture ...
  bool last_flush_wrote_fb = false;   // ... and whether the most recent flush is one of them
  // The raster launches of a flush are not issued with it: they are held back, and the first of them
  // goes out fused with the setup stage of the NEXT flush (wr_setup_raster_kernel), which it does not
  // depend on and which would otherwise sit between two frames as a dozen latency-bound workgroups
  // plus a kernel boundary; the rest follow in order.  Anything that needs their results, or touches a
  // texture they read or write from outside the draw stream (host uploads, copies, readbacks,
  // deletes, Finish), drains them first (drain_tail).
  struct Held { int fmt, depth, feat, nb, off; uint64_t algo_bytes; int mr_rows = 0; int dense = 0; int row_t0 = 0, row_n = 0, row_items = 0, row_mode = 0; };   // dense: the level's R8-texture prims are glyph runs (wr_raster_dense_kernel)   // mr_rows > 0: wr_mask_rows_kernel goes first (bound on its rows)    // one raster launch: wr_raster_kernel<fmt, depth, 4, feat>, nb bins from `off`
  struct Tail {
    bool pending = false;
    std::vector<Held> held;          // the raster launches of the held-back flush, in order
    int n_targets = 0;
    const WrTargetDesc* targets = nullptr; const WrDrawDesc* draws = nullptr;
    int set = 0;
    std::vector<GLuint> refs;
  } tail;
  bool defer_tail = true;
  // Host -> HBM staging copies go through their own stream so the DMA of frame k+1's data overlaps the
  // raster work of frame k (in-stream it was a 30 us hole in every frame: 1.2 MB over PCIe); the draw
  // stream waits for the copy's event before the upload scatter.  Safe without further ordering: the copy
  // writes space of the staging mirror ring that ring_make_safe() has found dead (its last readers' fence completed).
  wr_stream_t copy_stream;
  wr_event_t ev_copy;
  bool copy_overlap = true;
  // The setup stage of flush k+1 next to the held-back raster launches of flush k WITHOUT sharing a kernel with them: on a
  // stream of its own, after this flush's upload scatter (ev_up), the draw stream waiting for it (ev_setup) behind the
  // held-back launches.  The fused kernel costs the tile pass its register budget (25.8 vs 19.9 us on cfg2: the setup
  // stage's registers put the rect loop at 4 waves per SIMD); two launches on two queues keep both budgets -- and cost five
  // more runtime calls per flush (2 records, 2 waits, 1 launch, ~5 us each on the submit thread), which is what bounds a cfg2
  // frame: MEASURED SLOWER on every workload (profiles/r04_a_setup_stream_ab.txt: cfg2 14.7 k vs 20.2-21.6 k frames/s).
  // Off; WRHIP_SETUP_STREAM=1 turns it on for A/B runs.
  wr_stream_t setup_stream;
  wr_event_t ev_up, ev_setup;
  bool setup_on_stream = false;
  WrUnsupportedCounters* dcounters = nullptr;
  WrUnsupportedCounters seen = {};
  // HBM pool for texture storage: per-frame textures (GpuBufferF/I, render
  // targets) are recycled without hipMalloc/hipFree or a device sync; reuse is
  // safe because every consumer runs on this context's single stream.
  std::multimap<size_t, void*> pool;
  size_t pool_bytes = 0;
  // upload staging ring (pinned)
  uint8_t* staging = nullptr; size_t staging_size = 0, staging_pos = 0;
  // The ring is reused lap after lap without draining the stream: `ring_base + staging_pos` is a position's VIRTUAL address
  // (ring_base grows by the ring size at every wrap), a fence is recorded on the draw stream at the end of every flush with the
  // lowest virtual address not shipped yet, and bytes below fence k-1's address are dead once fence k has completed (flush
  // k-1's held-back raster launches, which still read its arena, go out with flush k).  An allocation that would overwrite
  // virtual addresses above `ring_safe` waits for the one fence that frees them -- a frame or two back, normally long done --
  // instead of the whole pipeline (cfg5 stages ~35 MB per frame: a 96 MB ring wrapped, and drained the GPU, every third frame).
  uint64_t ring_base = 0, ring_safe = 0;
  struct RingFence { uint64_t v = 0; wr_event_t ev; bool made = false; };
  static const int RING_FENCES = 8;
  RingFence ring_fence[RING_FENCES];
  uint64_t ring_fences = 0;            // fences recorded so far (fence k lives in slot k % RING_FENCES)
  // profiling
  bool profiling = false;
  bool profiling_deferred = false;     // WrhipSetProfiling(2): launches are timed where throughput mode issues them (held-back tail, fused kernels)
  wr_event_t ev_a, ev_b;
  WrhipStats stats;
  std::vector<WrhipKernelStat> kstats;   // per kernel variant, while profiling
  int shard_rank = 0, shard_world = 1;
  int next_query_slot = 0;
  GLuint query_slot_owner[WR_QUERY_SLOTS] = {};   // the GL_SAMPLES_PASSED query that last took each device slot
  bool forward_composites = true;      // WRHIP_NO_FORWARD=1 turns the write-through of opaque 1:1 composites off
  bool profiling_no_forward = false;
  bool mask_rows = true;               // WRHIP_NO_MASK_ROWS=1: cs_clip_* prims are evaluated inside the bin raster
  bool span_rows = true;               // WRHIP_NO_SPAN_ROWS=1: cs_blur / cs_scale targets go through the bin raster like everything else
  bool tile_rows = true;               // WRHIP_NO_TILE_ROWS=1: picture targets of a few large gradient / image prims too
  bool grad_tables = true;             // WRHIP_NO_GTAB=1: gradient tables are not copied into the pool (the raster stage reads sGpuBufferF, and an upload
                                       // of that texture waits for the held-back launches that do)
  size_t gtab_pending = 0;             // words promised to recorded, not yet flushed gradient draws (WR_DF_GTAB)
  bool quad_rowtabs = true;            // WRHIP_NO_QTAB=1: no row tables of general quads (the raster stage sums every row's edge values itself)
  size_t runs_pool_words = (size_t)16 << 20;   // WRHIP_RUNS_POOL_WORDS: the share of a flush's pool (WrTargetDesc::qtab) kept for depth runs / occluder
                                       // lists that outgrow their LDS copies, 4-byte words (64 MB to start with; 0: none -- such rows are then reported).
                                       // GROWN ON DEMAND: the pool's allocation word counts what was asked for, granted or not; Finish reads the last
                                       // flush's and, if it ran out, sizes the share for what that frame asked (the frame itself stays reported:
                                       // re-running a flush whose targets load their old content is not idempotent), up to 4 GB of the 288
  unsigned long long* last_qctl = nullptr;     // the most recent flush's pool word (in the staging mirror) and the capacity it was given
  size_t last_qtab_cap = 0;
  bool cell_raster = true;             // WRHIP_NO_CELLS=1: rect-only bins always take the pixel walk
  bool thin_r8 = true;                 // WRHIP_NO_THIN=1: small R8 launches keep the 4-wave workgroup shape
  int thin_parts = 4;                  // workgroups per bin of a thin launch (WRHIP_THIN_PARTS = 1, 2, 4, 8, 16): 16 / parts waves each, so that a wave shares its SIMD with fewer others
  bool dense_text = false;             // WRHIP_DENSE_TEXT=1: text levels run the 128-VGPR build of their variant (wr_raster_dense_kernel: four waves
                                       // per SIMD; the default until the glyph walk read 32-byte glyph records -- since then the 168-VGPR
                                       // build, which does not spill, is the faster one: cfg3 97.6 vs 102.3 us, profiles/r04_g_dense_waves_ab.txt)
  int chain_grid = 0;                  // workgroups of a chained R8 launch (0: off -- the default; WRHIP_CHAIN=1 turns it on, WRHIP_CHAIN_GRID overrides)
  unsigned chain_base = 0;             // value of WrUnsupportedCounters::chain_arrive once every chained launch enqueued so far has run

  Context() {
    wrrt::stream_create(&stream);
    wrrt::stream_create(&copy_stream);
    wrrt::event_create_sync(&ev_copy);
    wrrt::stream_create(&setup_stream);
    wrrt::event_create_sync(&ev_up);
    wrrt::event_create_sync(&ev_setup);
    setup_on_stream = getenv("WRHIP_SETUP_STREAM") && atoi(getenv("WRHIP_SETUP_STREAM")) != 0;
    copy_overlap = getenv("WRHIP_NO_COPY_STREAM") == nullptr;
    forward_composites = getenv("WRHIP_NO_FORWARD") == nullptr;
    thin_r8 = getenv("WRHIP_NO_THIN") == nullptr;
    if (const char* e = getenv("WRHIP_THIN_PARTS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) thin_parts = v; }
    dense_text = getenv("WRHIP_DENSE_TEXT") != nullptr;
    // Chained thin R8 levels (wr_raster_chain_kernel) are OFF unless WRHIP_CHAIN=1: measured on cfg4 (profiles/r03_e_chain_ab.txt),
    // the five chained levels take 144 us in one launch against ~100 us + four kernel boundaries apart -- a grid barrier that has
    // to write back and invalidate the XCDs' L2s (the levels run on all eight) costs about what a kernel boundary costs.
    // (grid: half the CUs -- two contexts' chained launches may meet on the chip and every workgroup of both has to be resident)
    if (thin_r8 && getenv("WRHIP_CHAIN") && atoi(getenv("WRHIP_CHAIN")) > 0)
      chain_grid = getenv("WRHIP_CHAIN_GRID") ? atoi(getenv("WRHIP_CHAIN_GRID")) : wrrt::cu_count() / 2;
    cell_raster = getenv("WRHIP_NO_CELLS") == nullptr;
    fuse_scatter = getenv("WRHIP_NO_FUSE_SCATTER") == nullptr;
    mask_rows = getenv("WRHIP_NO_MASK_ROWS") == nullptr;
    span_rows = getenv("WRHIP_NO_SPAN_ROWS") == nullptr;
    tile_rows = getenv("WRHIP_NO_TILE_ROWS") == nullptr;
    quad_rowtabs = getenv("WRHIP_NO_QTAB") == nullptr;
    grad_tables = getenv("WRHIP_NO_GTAB") == nullptr;
    if (const char* e = getenv("WRHIP_RUNS_POOL_WORDS")) runs_pool_words = (size_t)atoll(e);
    wrrt::event_create(&ev_a); wrrt::event_create(&ev_b);
    memset(&stats, 0, sizeof(stats));
    dcounters = (WrUnsupportedCounters*)wrrt::dev_alloc(sizeof(WrUnsupportedCounters));
    wrrt::memset8(dcounters, 0, sizeof(WrUnsupportedCounters), stream);
  }
  ~Context();

  GLuint& get_binding(GLenum name) {
    switch (name) {
      case GL_PIXEL_PACK_BUFFER: return pixel_pack_buffer_binding;
      case GL_PIXEL_UNPACK_BUFFER: return pixel_unpack_buffer_binding;
      case GL_ARRAY_BUFFER: return array_buffer_binding;
      case GL_ELEMENT_ARRAY_BUFFER: return vertex_arrays[current_vertex_array].element_array_buffer_binding;
      case GL_TEXTURE_2D: return texture_units[active_texture_unit].texture_2d_binding;
      case GL_TEXTURE_RECTANGLE: return texture_units[active_texture_unit].texture_rectangle_binding;
      case GL_TIME_ELAPSED: return time_elapsed_query;
      case GL_SAMPLES_PASSED: return samples_passed_query;
      case GL_RENDERBUFFER: return renderbuffer_binding;
      case GL_DRAW_FRAMEBUFFER: return draw_framebuffer_binding;
      case GL_READ_FRAMEBUFFER: return read_framebuffer_binding;
      default: return unknown_binding;
    }
  }
};

Context* ctx = nullptr;
bool g_rt_ok = false, g_rt_tried = false;
char g_device_name[256] = "";

void ensure_runtime() {
  if (g_rt_tried) {
    if (!g_rt_ok) { fprintf(stderr, "libwrhip: no HIP device available; this backend has no CPU path\n"); abort(); }
    return;
  }
  g_rt_tried = true;
  int dev = 0;
  g_rt_ok = wrrt::init(&dev, g_device_name, sizeof(g_device_name));
  if (!g_rt_ok) { fprintf(stderr, "libwrhip: no HIP device available; this backend has no CPU path\n"); abort(); }
}

// ---------------------------------------------------------------------------
int bytes_for_internal_format(GLenum f) {
  switch (f) {
    case GL_RGBA32F: case GL_RGBA32I: return 16;
    case GL_RGBA8: case GL_BGRA8: case GL_RGBA: return 4;
    case GL_R8: case GL_RED: return 1;
    case GL_RG8: case GL_RG: return 2;
    case GL_DEPTH_COMPONENT: case GL_DEPTH_COMPONENT16: case GL_DEPTH_COMPONENT24: case GL_DEPTH_COMPONENT32: return 4;
    case GL_RGB_RAW_422_APPLE: return 2;
    case GL_R16: return 2;
    case GL_RG16: return 4;
    default: return 0;
  }
}
int aligned_stride(int row_bytes) { return (row_bytes + 3) & ~3; }
GLenum remap_internal_format(GLenum format) {
  switch (format) {
    case GL_DEPTH_COMPONENT: return GL_DEPTH_COMPONENT24;
    case GL_RGBA: return GL_RGBA8;
    case GL_RED: return GL_R8;
    case GL_RG: return GL_RG8;
    case GL_RGB_422_APPLE: return GL_RGB_RAW_422_APPLE;
    default: return format;
  }
}
int wr_format(GLenum f) {
  switch (f) {
    case GL_RGBA32F: return WR_FMT_RGBA32F;
    case GL_RGBA32I: return WR_FMT_RGBA32I;
    case GL_RGBA8: return WR_FMT_RGBA8;
    case GL_R8: return WR_FMT_R8;
    case GL_RG8: return WR_FMT_RG8;
    case GL_R16: return WR_FMT_R16;
    case GL_RG16: return WR_FMT_RG16;
    case GL_DEPTH_COMPONENT24: return WR_FMT_DEPTH24;
    default: return WR_FMT_NONE;
  }
}
GLenum internal_format_for_data(GLenum format, GLenum ty) {
  if (format == GL_RED && ty == GL_UNSIGNED_BYTE) return GL_R8;
  if ((format == GL_RGBA || format == GL_BGRA) && (ty == GL_UNSIGNED_BYTE || ty == GL_UNSIGNED_INT_8_8_8_8_REV)) return GL_RGBA8;
  if (format == GL_RGBA && ty == GL_FLOAT) return GL_RGBA32F;
  if (format == GL_RGBA_INTEGER && ty == GL_INT) return GL_RGBA32I;
  if (format == GL_RG && ty == GL_UNSIGNED_BYTE) return GL_RG8;
  if (format == GL_RGB_422_APPLE && ty == GL_UNSIGNED_SHORT_8_8_REV_APPLE) return GL_RGB_RAW_422_APPLE;
  if (format == GL_RED && ty == GL_UNSIGNED_SHORT) return GL_R16;
  if (format == GL_RG && ty == GL_UNSIGNED_SHORT) return GL_RG16;
  return 0;
}
bool format_requires_conversion(GLenum external_format, GLenum internal_format) {
  return external_format == GL_RGBA && internal_format == GL_RGBA8;
}
void out_of_memory() { ctx->last_error = GL_OUT_OF_MEMORY; }

uint64_t get_time_value() {
  struct timespec tp;
  clock_gettime(CLOCK_MONOTONIC, &tp);
  return tp.tv_sec * 1000000000ULL + tp.tv_nsec;
}

// ---------------------------------------------------------------------------
// Deferred work: flush
void flush_all();
void drain_tail();
// every host-side wait for the stream: the held-back raster level goes out first
void sync_stream();

void flush_uploads(size_t extra_end = 0, bool may_defer = false);
void prof_begin(wr_stream_t* on = nullptr);
void prof_end(int kind, int fmt, int depth, int feat, uint64_t algo_bytes, uint64_t workgroups, wr_stream_t* on = nullptr);

const size_t STAGING_BYTES_DEFAULT = size_t(96) << 20;
// WRHIP_STAGING_BYTES: another ring size (tests/test_hostsim_parity.py wraps a few-MB ring many times per run)
static size_t staging_bytes() { static const size_t v = getenv("WRHIP_STAGING_BYTES") ? std::max<size_t>(1 << 16, (size_t)atoll(getenv("WRHIP_STAGING_BYTES"))) : STAGING_BYTES_DEFAULT; return v; }
bool ring_make_safe(uint64_t need_v, bool may_drain);

// Make the virtual addresses below `need_v` of the staging ring reusable.  `may_drain`: fall back to draining the stream when no
// recorded fence frees them (false: report failure instead -- flush_uploads asks from inside a batch).
bool ring_make_safe(uint64_t need_v, bool may_drain) {
  Context* c = ctx;
  if (need_v <= c->ring_safe) return true;
  static const bool drain_only = getenv("WRHIP_RING_DRAIN") != nullptr;      // (A/B: the ring as it was -- a full drain whenever a lap ends)
  const uint64_t first = drain_only ? c->ring_fences : c->ring_fences > (uint64_t)Context::RING_FENCES ? c->ring_fences - Context::RING_FENCES + 1 : 1;
  for (uint64_t k = first; k < c->ring_fences; k++) {
    Context::RingFence& prev = c->ring_fence[(k - 1) % Context::RING_FENCES];
    if (prev.v < need_v) continue;
    HostTimer ht(&c->stats.host_wait_ns);
    wrrt::event_sync(&c->ring_fence[k % Context::RING_FENCES].ev);
    c->ring_safe = prev.v;
    return true;
  }
  if (!may_drain) return false;
  flush_uploads(771);                  // (what is staged goes out, then everything enqueued completes)
  sync_stream();
  c->ring_safe = c->ring_base + c->staging_pos;
  return true;
}
// End of a flush: the fence that (once the NEXT one has completed) frees what was shipped up to here.
void ring_record_fence() {
  Context* c = ctx;
  if (!c->staging) return;
  // (one fence per eighth of the ring, not per flush: small frames -- cfg1 stages 0.3 MB -- need no event of their own, and the
  // eight slots then always reach a full lap back.  Flush k-1's held-back launches still go out with flush k, which precedes
  // whatever flush records the next fence.)
  const uint64_t v_now = c->ring_base + (c->upload_open ? c->upload_begin : c->staging_pos);
  if (c->ring_fences && v_now - c->ring_fence[(c->ring_fences - 1) % Context::RING_FENCES].v < c->staging_size / Context::RING_FENCES) return;
  Context::RingFence& f = c->ring_fence[c->ring_fences % Context::RING_FENCES];
  if (!f.made) { wrrt::event_create_sync(&f.ev); f.made = true; }
  f.v = v_now;
  wrrt::event_record(&f.ev, c->stream);
  c->ring_fences++;
}

// Bump allocation from the pinned staging ring.  Returns the byte offset.
size_t staging_alloc(size_t n) {
  Context* c = ctx;
  n = (n + 255) & ~size_t(255);
  if (!c->staging || n > c->staging_size) {
    flush_uploads(797);
    sync_stream();
    wrrt::pinned_free(c->staging);
    wrrt::dev_free(c->dupload);
    c->staging_size = std::max(n * 2, staging_bytes());
    c->staging = (uint8_t*)wrrt::pinned_alloc(c->staging_size);
    c->dupload = (uint8_t*)wrrt::dev_alloc(c->staging_size);
    c->staging_pos = 0;
    c->ring_base = c->ring_safe = 0; c->ring_fences = 0;      // (a new ring: nothing in flight refers to it)
  }
  if (c->staging_pos + n > c->staging_size) {
    flush_uploads(808);                // the pending batch must stay contiguous
    c->ring_base += c->staging_size;   // the next lap
    c->staging_pos = 0;
  }
  // what this allocation overwrites was staged one lap ago
  const uint64_t end_v = c->ring_base + c->staging_pos + n;
  if (end_v > c->staging_size) ring_make_safe(end_v - c->staging_size, true);
  if (!c->upload_open) { c->upload_open = true; c->upload_begin = c->staging_pos; }
  size_t off = c->staging_pos;
  c->staging_pos += n;
  return off;
}

// The scatter kernel runs the segments of a batch concurrently: a write that overlaps one already queued must land after
// it (GL: last write wins), so the batch queued so far goes out first.  Called before the new rows are staged.
void order_upload(const void* dst, size_t dst_stride, size_t row_bytes, size_t rows) {
  if (!rows || !row_bytes) return;
  const uint8_t* b0 = (const uint8_t*)dst; const uint8_t* b1 = b0 + (rows - 1) * dst_stride + row_bytes;
  for (const Context::UploadSeg& q : ctx->useg) {
    const uint8_t* a0 = (const uint8_t*)q.dst; const uint8_t* a1 = a0 + (size_t)(q.rows - 1) * q.dst_stride + q.row_bytes;
    if (a0 < b1 && b0 < a1) { flush_uploads(828); return; }
  }
}

// Queue rows already written at staging offset `src_off` for texture memory.
void queue_upload(size_t src_off, void* dst, size_t dst_stride, size_t row_bytes, size_t rows) {
  Context::UploadSeg sg;
  sg.src_off = src_off; sg.dst = dst; sg.dst_stride = (uint32_t)dst_stride; sg.row_bytes = (uint32_t)row_bytes;
  sg.rows = (uint32_t)rows; sg.pad = 0;
  ctx->useg.push_back(sg);
  ctx->stats.h2d_bytes += row_bytes * rows;
}

// One DMA for everything staged since the last flush, then the scatter kernel.
void flush_uploads(size_t why, bool may_defer) {
  Context* c = ctx;
  if (!c->upload_open) return;
  size_t nseg = c->useg.size();
  size_t seg_off = 0;
  if (nseg) {
    // descriptors ride along in the same batch (cannot wrap: checked by caller paths via staging_alloc)
    size_t need = (nseg * sizeof(WrUploadSeg) + 255) & ~size_t(255);
    const uint64_t desc_end_v = c->ring_base + c->staging_pos + need;
    if (c->staging_pos + need > c->staging_size || (desc_end_v > c->staging_size && !ring_make_safe(desc_end_v - c->staging_size, false))) {
      // no room for descriptors at the tail (or their bytes of the previous lap are still in flight): ship data first with per-segment copies
      for (auto& sg : c->useg)
        wrrt::copy2d(sg.dst, sg.dst_stride, c->staging + sg.src_off, sg.row_bytes, sg.row_bytes, sg.rows, 0, c->stream);
      c->useg.clear(); nseg = 0;
    } else {
      seg_off = c->staging_pos; c->staging_pos += need;
      WrUploadSeg* d = (WrUploadSeg*)(c->staging + seg_off);
      for (size_t i = 0; i < nseg; i++) {
        d[i].src = c->dupload + c->useg[i].src_off; d[i].dst = c->useg[i].dst;
        d[i].dst_stride = c->useg[i].dst_stride; d[i].row_bytes = c->useg[i].row_bytes; d[i].rows = c->useg[i].rows; d[i].pad = 0;
      }
    }
  }
  size_t b = c->upload_begin, e = c->staging_pos;
  if (e > b) {
    if (c->copy_overlap) {
      wrrt::h2d(c->dupload + b, c->staging + b, e - b, c->copy_stream);
      wrrt::event_record(&c->ev_copy, c->copy_stream);
      wrrt::stream_wait_event(c->stream, &c->ev_copy);
    } else {
      wrrt::h2d(c->dupload + b, c->staging + b, e - b, c->stream);
    }
  }
  if (nseg) {
    uint64_t up_bytes = 0, largest = 0;
    for (auto& sg : c->useg) {
      up_bytes += 2ull * sg.row_bytes * sg.rows;
      largest = std::max<uint64_t>(largest, (uint64_t)sg.row_bytes * sg.rows);
    }
    // (a 100 k-prim frame uploads ~24 MB in half a dozen segments: at 8 workgroups per segment the scatter was a 160 us launch)
    // (... and at one workgroup per 64 KB a thread copied its sixteen 16-byte pieces one dependent round trip after the other:
    // 9.8 us for the 1.3 MB of a cfg2 frame; one workgroup per 8 KB of the largest segment = two pieces per thread)
    const int parts = (int)std::min<uint64_t>(256, std::max<uint64_t>(8, (largest + 8191) >> 13));
    if (may_defer && c->fuse_scatter && !c->scatter_first && !c->ps.valid) {
      // the caller's next launch carries the setup stage: its first workgroups run the scatter (no launch of its own)
      c->ps.seg_off = seg_off; c->ps.nseg = (int)nseg; c->ps.parts = parts; c->ps.bytes = up_bytes; c->ps.valid = true;
    } else {
      static const bool dbg = getenv("WRHIP_DEBUG_SCATTER") != nullptr;
      if (dbg) fprintf(stderr, "scatter launch of its own (flush_uploads called from line %zu): %zu segments, may_defer %d scatter_first %d pending %d\n", why, nseg, (int)may_defer, (int)c->scatter_first, (int)c->ps.valid);
      prof_begin();
      WR_LAUNCH(wr_upload_kernel, (int)nseg * parts, 256, c->stream, (const WrUploadSeg*)(c->dupload + seg_off), (int)nseg, parts);
      prof_end(0, 0, 0, 0, up_bytes, nseg * parts);
      c->stats.kernel_launches++;
    }
    c->useg.clear();
  }
  c->upload_open = false;
  c->upload_batch++;
  c->scatter_first = false;
}
// a deferred scatter nobody took along (should not happen): on its own, now
void launch_pending_scatter() {
  Context* c = ctx;
  if (!c->ps.valid) return;
  prof_begin();
  WR_LAUNCH(wr_upload_kernel, c->ps.nseg * c->ps.parts, 256, c->stream, (const WrUploadSeg*)(c->dupload + c->ps.seg_off), c->ps.nseg, c->ps.parts);
  prof_end(0, 0, 0, 0, c->ps.bytes, (uint64_t)c->ps.nseg * c->ps.parts);
  c->stats.kernel_launches++;
  c->ps.valid = false;
}

void mark_ref(GLuint id, Texture& t, bool write, int target_index = -1) {
  if (!t.pending_read && !t.pending_write) ctx->referenced.push_back(id);
  if (write) { t.pending_write = true; t.pending_target = target_index; }
  else t.pending_read = true;
}

void flush_work(const std::vector<int>& sel);

// A host- or copy-side write to texture `t` (or its deletion / reallocation)
// must not overtake pending draws that read or write it.
void sync_texture_for_write(Texture& t) { if (t.pending_read || t.pending_write) flush_all(); if (t.tail_ref) drain_tail(); }
void sync_texture_for_read(Texture& t) { if (t.pending_write) flush_all(); if (t.tail_ref) drain_tail(); flush_uploads(924); }

size_t pool_round(size_t n) {
  size_t g = n <= (1u << 20) ? 4096 : (size_t(1) << 16);
  return (n + g - 1) / g * g;
}
void* pool_alloc(size_t n, size_t* actual) {
  Context* c = ctx;
  size_t r = pool_round(n);
  auto it = c->pool.lower_bound(r);
  if (it != c->pool.end() && it->first <= r + r / 4) {
    void* p = it->second; *actual = it->first;
    c->pool_bytes -= it->first;
    c->pool.erase(it);
    return p;
  }
  *actual = r;
  void* p = wrrt::try_dev_alloc(r + 64);
  if (!p && !c->pool.empty()) {
    // HBM exhausted: give the idle pool back and try once more before reporting GL_OUT_OF_MEMORY
    sync_stream();
    for (auto& kv : c->pool) wrrt::dev_free(kv.second);
    c->pool.clear(); c->pool_bytes = 0;
    p = wrrt::try_dev_alloc(r + 64);
  }
  if (!p) *actual = 0;
  return p;
}
void pool_free(void* p, size_t n) {
  Context* c = ctx;
  c->pool.insert(std::make_pair(n, p));
  c->pool_bytes += n;
  while (c->pool_bytes > (size_t(2) << 30) && !c->pool.empty()) {   // trim: keep at most 2 GiB idle
    auto it = --c->pool.end();
    sync_stream();
    wrrt::dev_free(it->second);
    c->pool_bytes -= it->first;
    c->pool.erase(it);
  }
}

void free_texture_storage(Texture& t) {
  sync_texture_for_write(t);
  flush_uploads(967);   // queued rows may target this storage
  if (t.dptr) {
    pool_free(t.dptr, t.dsize);
    t.dptr = nullptr; t.dsize = 0;
  }
  free(t.hmirror); t.hmirror = nullptr; t.hmirror_size = 0;
  t.depth_materialized = false;
}

bool allocate_texture(Texture& t) {
  int bpp = bytes_for_internal_format(t.internal_format);
  int stride = aligned_stride(bpp * t.width);
  size_t size = (size_t)stride * t.height;
  t.bpp = bpp; t.stride = stride;
  if (t.internal_format == GL_DEPTH_COMPONENT24) {
    // depth lives in registers during a flush; HBM storage is only
    // materialised if a depth buffer must survive a flush (rare)
    return true;
  }
  if (size == 0) return true;
  if (!t.dptr || size > t.dsize) {
    if (t.dptr) { flush_uploads(988); pool_free(t.dptr, t.dsize); }
    size_t actual = 0;
    t.dptr = pool_alloc(size, &actual);
    t.dsize = actual;
    if (!t.dptr) return false;
  }
  return true;
}

void set_tex_storage(Texture& t, GLenum external_format, GLsizei width, GLsizei height, void* buf = nullptr,
                     GLsizei stride = 0) {
  GLenum internal_format = remap_internal_format(external_format);
  sync_texture_for_write(t);
  if (t.width != width || t.height != height || t.internal_format != internal_format) {
    t.internal_format = internal_format; t.width = width; t.height = height;
  }
  t.depth_cleared = false; t.depth_materialized = false;
  if (!allocate_texture(t)) out_of_memory();
  t.ext_buf = nullptr; t.ext_stride = 0;
  if (buf && t.dptr) {
    bool conv = format_requires_conversion(external_format, internal_format);
    if (!conv) { t.ext_buf = buf; t.ext_stride = stride; }
    // upload current contents of the external buffer
    size_t row = (size_t)t.bpp * width;
    order_upload(t.dptr, t.stride, row, height);
    size_t st_off = staging_alloc(row * height);
    uint8_t* st = ctx->staging + st_off;
    for (int y = 0; y < height; y++) {
      const uint8_t* s = (const uint8_t*)buf + (size_t)y * stride;
      uint8_t* d = st + (size_t)y * row;
      if (conv) {
        for (int x = 0; x < width; x++) {
          uint32_t p = ((const uint32_t*)s)[x]; uint32_t rb = p & 0x00FF00FF;
          ((uint32_t*)d)[x] = (p & 0xFF00FF00) | (rb << 16) | (rb >> 16);
        }
      } else memcpy(d, s, row);
    }
    queue_upload(st_off, t.dptr, t.stride, row, height);
    t.up_batch = ctx->upload_batch; t.view_batch = 0;
  }
}

int hash_blend_key(Context* c) {
  GLenum srgb = c->blendfunc_srgb, drgb = c->blendfunc_drgb, sa = c->blendfunc_sa, da = c->blendfunc_da;
  if (c->blend_equation != GL_FUNC_ADD) {
    switch (c->blend_equation) {
      case GL_MIN: return WR_BLEND_MIN;
      case GL_MAX: return WR_BLEND_MAX;
      case GL_MULTIPLY_KHR: return WR_BLEND_MULTIPLY_KHR;
      case GL_SCREEN_KHR: return WR_BLEND_SCREEN_KHR;
      case GL_OVERLAY_KHR: return WR_BLEND_OVERLAY_KHR;
      case GL_DARKEN_KHR: return WR_BLEND_DARKEN_KHR;
      case GL_LIGHTEN_KHR: return WR_BLEND_LIGHTEN_KHR;
      case GL_COLORDODGE_KHR: return WR_BLEND_COLORDODGE_KHR;
      case GL_COLORBURN_KHR: return WR_BLEND_COLORBURN_KHR;
      case GL_HARDLIGHT_KHR: return WR_BLEND_HARDLIGHT_KHR;
      case GL_SOFTLIGHT_KHR: return WR_BLEND_SOFTLIGHT_KHR;
      case GL_DIFFERENCE_KHR: return WR_BLEND_DIFFERENCE_KHR;
      case GL_EXCLUSION_KHR: return WR_BLEND_EXCLUSION_KHR;
      case GL_HSL_HUE_KHR: return WR_BLEND_HSL_HUE_KHR;
      case GL_HSL_SATURATION_KHR: return WR_BLEND_HSL_SATURATION_KHR;
      case GL_HSL_COLOR_KHR: return WR_BLEND_HSL_COLOR_KHR;
      case GL_HSL_LUMINOSITY_KHR: return WR_BLEND_HSL_LUMINOSITY_KHR;
      default: return WR_BLEND_UNSUPPORTED;
    }
  }
  bool sep = (srgb != sa || drgb != da);
#define K2(s, d) (!sep && srgb == (s) && drgb == (d))
#define K4(s, d, a, b) (srgb == (s) && drgb == (d) && sa == (a) && da == (b))
  if (K2(GL_ONE, GL_ZERO)) return WR_BLEND_NONE;
  if (K4(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA, GL_ONE, GL_ONE_MINUS_SRC_ALPHA)) return WR_BLEND_ALPHA;
  if (K2(GL_ONE, GL_ONE_MINUS_SRC_ALPHA)) return WR_BLEND_PREMULT;
  if (K2(GL_ZERO, GL_ONE_MINUS_SRC_COLOR)) return WR_BLEND_ZERO_INV_SRC_COLOR;
  if (K4(GL_ZERO, GL_ONE_MINUS_SRC_COLOR, GL_ZERO, GL_ONE)) return WR_BLEND_ZERO_INV_SRC_COLOR_A1;
  if (K2(GL_ZERO, GL_ONE_MINUS_SRC_ALPHA)) return WR_BLEND_DEST_OUT;
  if (K2(GL_ZERO, GL_SRC_COLOR)) return WR_BLEND_MULTIPLY;
  if (K2(GL_ONE, GL_ONE)) return WR_BLEND_ADD;
  if (K4(GL_ONE, GL_ONE, GL_ONE, GL_ONE_MINUS_SRC_ALPHA)) return WR_BLEND_ADD_A_OVER;
  if (K4(GL_ONE_MINUS_DST_ALPHA, GL_ONE, GL_ZERO, GL_ONE)) return WR_BLEND_INV_DST_A;
  if (K2(GL_CONSTANT_COLOR, GL_ONE_MINUS_SRC_COLOR)) return WR_BLEND_CONST_COLOR;
  if (K2(GL_ONE, GL_ONE_MINUS_SRC1_COLOR)) return WR_BLEND_DUAL_SRC;
#undef K2
#undef K4
  return WR_BLEND_UNSUPPORTED;
}

GLenum remap_blendfunc(GLenum rgb, GLenum a) {  // gl.cc:1240-1292
  switch (a) {
    case GL_SRC_ALPHA: if (rgb == GL_SRC_COLOR) a = GL_SRC_COLOR; break;
    case GL_ONE_MINUS_SRC_ALPHA: if (rgb == GL_ONE_MINUS_SRC_COLOR) a = GL_ONE_MINUS_SRC_COLOR; break;
    case GL_DST_ALPHA: if (rgb == GL_DST_COLOR) a = GL_DST_COLOR; break;
    case GL_ONE_MINUS_DST_ALPHA: if (rgb == GL_ONE_MINUS_DST_COLOR) a = GL_ONE_MINUS_DST_COLOR; break;
    case GL_CONSTANT_ALPHA: if (rgb == GL_CONSTANT_COLOR) a = GL_CONSTANT_COLOR; break;
    case GL_ONE_MINUS_CONSTANT_ALPHA: if (rgb == GL_ONE_MINUS_CONSTANT_COLOR) a = GL_ONE_MINUS_CONSTANT_COLOR; break;
    case GL_SRC_COLOR: if (rgb == GL_SRC_ALPHA) a = GL_SRC_ALPHA; break;
    case GL_ONE_MINUS_SRC_COLOR: if (rgb == GL_ONE_MINUS_SRC_ALPHA) a = GL_ONE_MINUS_SRC_ALPHA; break;
    case GL_DST_COLOR: if (rgb == GL_DST_ALPHA) a = GL_DST_ALPHA; break;
    case GL_ONE_MINUS_DST_COLOR: if (rgb == GL_ONE_MINUS_DST_ALPHA) a = GL_ONE_MINUS_DST_ALPHA; break;
    case GL_CONSTANT_COLOR: if (rgb == GL_CONSTANT_ALPHA) a = GL_CONSTANT_ALPHA; break;
    case GL_ONE_MINUS_CONSTANT_COLOR: if (rgb == GL_ONE_MINUS_CONSTANT_ALPHA) a = GL_ONE_MINUS_CONSTANT_ALPHA; break;
    case GL_SRC1_ALPHA: if (rgb == GL_SRC1_COLOR) a = GL_SRC1_COLOR; break;
    case GL_ONE_MINUS_SRC1_ALPHA: if (rgb == GL_ONE_MINUS_SRC1_COLOR) a = GL_ONE_MINUS_SRC1_COLOR; break;
    case GL_SRC1_COLOR: if (rgb == GL_SRC1_ALPHA) a = GL_SRC1_ALPHA; break;
    case GL_ONE_MINUS_SRC1_COLOR: if (rgb == GL_ONE_MINUS_SRC1_ALPHA) a = GL_ONE_MINUS_SRC1_ALPHA; break;
  }
  return a;
}

Framebuffer* get_framebuffer(GLenum target, bool fallback = false) {
  if (target == GL_FRAMEBUFFER) target = GL_DRAW_FRAMEBUFFER;
  Framebuffer* fb = ctx->framebuffers.find(ctx->get_binding(target));
  if (fallback && !fb) fb = &ctx->framebuffers[0];
  return fb;
}

// apply_scissor(t), gl.cc:857-864, in texture pixels
void apply_scissor(const Texture& t, int out[4]) {
  int x0 = 0, y0 = 0, x1 = t.width, y1 = t.height;
  if (ctx->scissortest) {
    x0 = std::max(x0, ctx->scissor[0] - t.offx); y0 = std::max(y0, ctx->scissor[1] - t.offy);
    x1 = std::min(x1, ctx->scissor[2] - t.offx); y1 = std::min(y1, ctx->scissor[3] - t.offy);
  }
  out[0] = x0; out[1] = y0; out[2] = x1; out[3] = y1;
}

int find_or_add_work(GLuint tex_id) {
  {
    Texture& t = ctx->textures[tex_id];
    // the target is about to be written (again): pending draws that sample it must run first
    if (t.pending_read) flush_all();
    if (t.pending_write && t.pending_target >= 0) return t.pending_target;
  }
  if (!ctx->spare.empty()) {
    ctx->work.push_back(std::move(ctx->spare.back()));
    ctx->spare.pop_back();
  } else {
    ctx->work.emplace_back();
  }
  ctx->work.back().tex = tex_id;
  int idx = (int)ctx->work.size() - 1;
  mark_ref(tex_id, ctx->textures[tex_id], true, idx);
  return idx;
}

// A colour target is about to use depth attachment `depth_tex` (a test, a write, a partial clear).  If another target's
// pending draws left depth there that the caller has neither invalidated nor cleared, that target must run first -- its
// depth is materialised by the flush and loaded here (GL lets a depth buffer be carried from one colour target to the
// next; WebRender clears it per target and never does).
void claim_depth(GLuint tex_id, GLuint depth_tex, bool full_clear) {
  Texture* dt = ctx->textures.find(depth_tex);
  if (!dt) return;
  if (dt->depth_owner && dt->depth_owner != tex_id) {
    Texture* ot = ctx->textures.find(dt->depth_owner);
    TargetWork* ow = (ot && ot->pending_write && ot->pending_target >= 0) ? &ctx->work[ot->pending_target] : nullptr;
    if (ow && ow->depth_live && ow->depth_tex == depth_tex) {
      if (full_clear || !dt->depth_cleared) ow->depth_live = false;      // overwritten / invalidated: nobody will read it
      else flush_all();
    }
  }
  dt->depth_owner = tex_id;
}

void record_clear(GLuint tex_id, bool color, uint32_t color_value, bool depth, GLuint depth_tex, uint32_t depth_value,
                  const int rect[4]) {
  {
    Texture& t = ctx->textures[tex_id];
    if (!t.has_storage() || t.own_none()) return;
  }
  if (depth) {
    const Texture& ct = ctx->textures[tex_id];
    claim_depth(tex_id, depth_tex, rect[0] <= 0 && rect[1] <= 0 && rect[2] >= ct.width && rect[3] >= ct.height);
  }
  int wi = find_or_add_work(tex_id);
  Texture& t = ctx->textures[tex_id];
  WrDrawDesc d;
  memset(&d, 0, sizeof(d));
  d.query_slot = -1;
  d.shader = WR_SH_CLEAR_OP;
  d.target = wi;
  d.count = 1;
  d.flags = (color ? WR_DF_CLEAR_COLOR : 0) | (depth ? WR_DF_CLEAR_DEPTH : 0);
  d.clip[0] = std::max(rect[0], 0); d.clip[1] = std::max(rect[1], 0);
  d.clip[2] = std::min(rect[2], t.width); d.clip[3] = std::min(rect[3], t.height);
  d.clear_color = color_value;
  d.clear_depth = depth_value;
  for (int k = 0; k < WR_MAX_ATTRIBS; k++) d.attr_off[k] = -1;
  TargetWork& w = ctx->work[wi];
  if (depth) {
    w.depth_tex = depth_tex;
    Texture& dt = ctx->textures[depth_tex];
    if (!w.init_depth_set) { w.init_depth = dt.depth_value; w.init_depth_set = true; }
    w.depth_live = true;
    dt.depth_owner = tex_id;
  }
  w.draws.push_back(d);
  w.prims += 1;
}

void download_texture(Texture& t) {
  // HBM -> host mirror (or external buffer) after a full flush
  sync_texture_for_read(t);
  if (!t.dptr) return;
  size_t row = (size_t)t.bpp * t.width;
  if (t.ext_buf) {
    wrrt::copy2d(t.ext_buf, t.ext_stride, t.dptr, t.stride, row, t.height, 1, ctx->stream);
  } else {
    size_t need = (size_t)t.stride * t.height;
    if (t.hmirror_size < need) { free(t.hmirror); t.hmirror = (uint8_t*)malloc(need + 64); t.hmirror_size = need; }
    wrrt::copy2d(t.hmirror, t.stride, t.dptr, t.stride, row, t.height, 1, ctx->stream);
  }
  ctx->stats.d2h_bytes += row * t.height;
  sync_stream();
}

Context::~Context() {
  Context* saved = ctx;
  ctx = this;
  flush_all();
  flush_uploads(1206);
  sync_stream();
  for (Texture* t : textures.objects) if (t) { if (t->dptr) wrrt::dev_free(t->dptr); t->dptr = nullptr; free(t->hmirror); t->hmirror = nullptr; }
  for (auto& kv : pool) wrrt::dev_free(kv.second);
  pool.clear();
  wrrt::dev_free(dupload); wrrt::dev_free(dcounters);
  for (Scratch& S : scratch) { wrrt::dev_free(S.prims); wrrt::dev_free(S.recs); wrrt::dev_free(S.aux); wrrt::dev_free(S.vtab); wrrt::dev_free(S.qtab); wrrt::dev_free(S.masks); wrrt::dev_free(S.bin_ctr); wrrt::dev_free(S.mr_slots); wrrt::dev_free(S.mr_store); wrrt::dev_free(S.flat); }
  wrrt::pinned_free(staging);
  wrrt::event_destroy(ev_a); wrrt::event_destroy(ev_b);
  wrrt::event_destroy(ev_copy);
  for (RingFence& f : ring_fence) if (f.made) wrrt::event_destroy(f.ev);
  wrrt::stream_destroy(copy_stream);
  wrrt::stream_destroy(stream);
  ctx = saved == this ? nullptr : saved;
}

// ---------------------------------------------------------------------------
// Execute the selected pending targets: one H2D copy of the frame arena, then
// vertex + bin + raster launches covering every selected target at once.
// While profiling, every launch is bracketed by its own event pair and waited for (measurement runs only).
void prof_begin(wr_stream_t* on) { if (ctx->profiling) wrrt::event_record(&ctx->ev_a, on ? *on : ctx->stream); }
void prof_end(int kind, int fmt, int depth, int feat, uint64_t algo_bytes, uint64_t workgroups, wr_stream_t* on) {
  Context* c = ctx;
  if (!c->profiling) return;
  wrrt::event_record(&c->ev_b, on ? *on : c->stream);
  wrrt::event_sync(&c->ev_b);
  const uint64_t ns = (uint64_t)(wrrt::event_elapsed_ms(&c->ev_a, &c->ev_b) * 1.0e6);
  WrhipKernelStat* k = nullptr;
  for (WrhipKernelStat& e : c->kstats) if (e.kind == kind && e.fmt == fmt && e.depth == depth && e.feat == feat) k = &e;
  if (!k) { c->kstats.push_back(WrhipKernelStat{kind, fmt, depth, feat, 0, 0, 0, 0}); k = &c->kstats.back(); }
  k->launches++; k->ns += ns; k->algo_bytes += algo_bytes; k->workgroups += workgroups;
  if (kind == 2 || kind == 4 || kind == 5 || kind == 6 || kind == 7 || kind == 9 || kind == 10 || kind == 11 || kind == 12 || kind == 13) c->stats.raster_ns += ns;
}

void tail_launched() {
  Context::Tail& T = ctx->tail;
  for (GLuint id : T.refs) if (Texture* t = ctx->textures.find(id)) t->tail_ref = false;
  T.refs.clear();
  T.held.clear();
  T.pending = false;
}
// The instantiated raster kernels: RGBA8 (+depth) x {0, TEX|GENERIC, +R8TEX, everything}, R8 x {0, GENERIC|BLUR, +CLIP}.
// `SA` non-null: launch the fused setup + raster variant (only for the variants can_fuse() names).
#ifndef WR_THIN_MAX_BINS
#define WR_THIN_MAX_BINS 256
#endif
bool can_fuse(const Context::Held& H) {
  if (H.row_n > 0) return H.row_mode == 2;     // (tile rows carry it: wr_setup_tile_rows_kernel; a span-rows launch is too short to hide a setup stage behind)
  if (H.mr_rows > 0) return true;          // (the mask-rows launch ahead of an R8 raster launch: wr_setup_rows_kernel)
  // a small textured colour launch is worth more as a THIN launch (four workgroups per bin: a quarter of the rows per wave, four
  // times the waves) than as the carrier of the setup stage: transforms-simple, 256 bins of eleven rotated rects, 241 us fused
  // (round 5, later: the thin launch carries it itself -- wr_setup_raster_thin_kernel; WRHIP_NO_FUSE_THIN=1: it leaves without)
  static const bool thin_first = getenv("WRHIP_FUSE_SMALL") == nullptr;
  static const bool fuse_thin = getenv("WRHIP_NO_FUSE_THIN") == nullptr;
  if (thin_first && ctx->thin_r8 && H.fmt == WR_FMT_RGBA8 && !H.depth && H.nb <= WR_THIN_MAX_BINS && H.feat == (WR_FEAT_TEX | WR_FEAT_GENERIC)) return fuse_thin;
  return H.fmt == WR_FMT_RGBA8 && (H.feat == 0 || H.feat == (WR_FEAT_TEX | WR_FEAT_GENERIC) ||
                                   H.feat == (WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX));
}
// `chain_n` >= 2: H is the first of chain_n consecutive thin R8 launches of one variant (chainable()); they go out as one
// wr_raster_chain_kernel launch.
bool chainable(const Context::Held& H) {
  return ctx->chain_grid > 0 && H.row_n == 0 && H.fmt == WR_FMT_R8 && H.nb <= WR_THIN_MAX_BINS && (H.feat == 0 || H.feat == (WR_FEAT_GENERIC | WR_FEAT_BLUR));
}
void launch_raster(const Context::Held& H, const WrTargetDesc* targets, int n_targets, const WrDrawDesc* draws, Context::Scratch& S,
                   const WrSetupArgs* SA = nullptr, int n_setup_blocks = 0, int chain_n = 1, uint64_t setup_bytes = 0) {
  Context* c = ctx;
#define WR_K(FMT, DEPTH, FEAT)                                                                                          \
  do {                                                                                                                  \
    WR_LAUNCH((wr_raster_kernel<FMT, DEPTH, 4, FEAT>), H.nb, 256, c->stream, targets, n_targets, draws,                 \
              (const WrPrim*)S.prims, (const WrRec*)S.recs, (const WrAux*)S.aux, (const float*)S.vtab, S.masks, H.off); \
  } while (0)
#define WR_KF(DEPTH, FEAT)                                                                                              \
  do {                                                                                                                  \
    WR_LAUNCH((wr_setup_raster_kernel<WR_FMT_RGBA8, DEPTH, 4, FEAT>), n_setup_blocks + SA->up_blocks + H.nb, 256, c->stream, *SA,        \
              n_setup_blocks, targets, n_targets, draws, (const WrPrim*)S.prims, (const WrRec*)S.recs,                  \
              (const WrAux*)S.aux, (const float*)S.vtab, S.masks, H.off);                                               \
  } while (0)
  if (H.row_n > 0) {
    // the span-rows targets of a level: one wave per (target row, 256-pixel piece), four waves per workgroup
    const int wgs = std::max(1, std::min((H.row_items + 3) / 4, 16384));
    static const bool dbg_rows = getenv("WRHIP_DEBUG_ROWS") != nullptr;
    if (dbg_rows) fprintf(stderr, "span rows (mode %d): targets [%d, %d) items %d workgroups %d\n", H.row_mode, H.row_t0, H.row_t0 + H.row_n, H.row_items, wgs);
    prof_begin();
    const bool rows_fused = SA != nullptr && H.row_mode == 2;
    if (rows_fused) WR_LAUNCH(wr_setup_tile_rows_kernel, n_setup_blocks + SA->up_blocks + wgs, 256, c->stream, *SA, n_setup_blocks, targets, H.row_t0, H.row_n, draws, (const WrPrim*)S.prims, (const WrAux*)S.aux);
    else if (H.row_mode == 2) WR_LAUNCH(wr_tile_rows_kernel, wgs, 256, c->stream, targets, H.row_t0, H.row_n, draws, (const WrPrim*)S.prims, (const WrAux*)S.aux);
    else WR_LAUNCH(wr_span_rows_kernel, wgs, 256, c->stream, targets, H.row_t0, H.row_n, draws, (const WrPrim*)S.prims, (const WrAux*)S.aux);
    prof_end(rows_fused ? 11 : (H.row_mode == 2 ? 10 : 9), H.fmt, 0, 0, H.algo_bytes + (rows_fused ? setup_bytes : 0), (uint64_t)wgs + (rows_fused ? n_setup_blocks : 0));
    c->stats.kernel_launches++; c->stats.raster_launches++; c->stats.row_launches++;
    return;
  }
  const int F5 = WR_FEAT_TEX | WR_FEAT_GENERIC, F7 = F5 | WR_FEAT_R8TEX, FA = F7 | WR_FEAT_BLUR | WR_FEAT_SHADE;
  if (H.mr_rows > 0 && S.mr_ctl) {
    // the cs_clip_* prims of this launch's targets, row by row (one wave per row), ahead of the bins that blend them
    const int wgs = std::max(1, std::min((H.mr_rows + 3) / 4, 4096));
    prof_begin();
    const bool rows_fused = SA != nullptr;
    if (SA) {
      WR_LAUNCH(wr_setup_rows_kernel, n_setup_blocks + SA->up_blocks + wgs, 256, c->stream, *SA, n_setup_blocks, targets, H.off, H.off + H.nb,
                (const WrPrim*)S.prims, (const WrAux*)S.aux, S.mr_ctl, (const WrMaskSlot*)S.mr_slots, S.mr_store);
      SA = nullptr;                          // (the raster launch that follows is the plain one)
    } else
    WR_LAUNCH(wr_mask_rows_kernel, wgs, 256, c->stream, targets, H.off, H.off + H.nb, (const WrPrim*)S.prims, (const WrAux*)S.aux,
              S.mr_ctl, (const WrMaskSlot*)S.mr_slots, S.mr_store);
    prof_end(rows_fused ? 8 : 3, H.fmt, 0, 0, rows_fused ? setup_bytes : 0, (uint64_t)wgs + (rows_fused ? n_setup_blocks : 0));      // (8: wr_setup_rows_kernel)
    if (c->profiling) {          // bytes this launch evaluated: the kernel's running count, read back (profiling syncs per launch anyway)
      unsigned long long parts[32], seen = 0;
      wrrt::d2h(parts, S.mr_ctl + 32, sizeof(parts), c->stream);
      wrrt::stream_sync(c->stream);
      for (unsigned long long v : parts) seen += v;
      for (WrhipKernelStat& e : c->kstats) if (e.kind == (rows_fused ? 8 : 3) && e.fmt == H.fmt) e.algo_bytes += seen - S.mr_seen;
      S.mr_seen = seen;
    }
    c->stats.kernel_launches++;
  }
  if (chain_n >= 2) {
    WrChain ch;
    memset(&ch, 0, sizeof(ch));
    ch.n = chain_n;
    int widest = 0; uint64_t bytes = 0;
    for (int i = 0; i < chain_n; i++) { ch.first[i] = (&H)[i].off; ch.count[i] = (&H)[i].nb; widest = std::max(widest, (&H)[i].nb); bytes += (&H)[i].algo_bytes; }
    const int grid = std::max(1, std::min(widest, c->chain_grid));
    // arrivals at the barrier after level l: the workgroups that have a bin at level l or later (the others have left)
    for (int l = 0; l + 1 < chain_n; l++) {
      int part = 0;
      for (int k = l; k < chain_n; k++) part = std::max(part, std::min(grid, ch.count[k]));
      c->chain_base += (unsigned)part;
      ch.want[l] = c->chain_base;
    }
    prof_begin();
    if (H.feat == 0)
      WR_LAUNCH((wr_raster_chain_kernel<0>), grid, 1024, c->stream, targets, n_targets, draws, (const WrPrim*)S.prims, (const WrRec*)S.recs,
                (const WrAux*)S.aux, (const float*)S.vtab, S.masks, ch, &c->dcounters->chain_arrive, c->dcounters);
    else
      WR_LAUNCH((wr_raster_chain_kernel<WR_FEAT_GENERIC | WR_FEAT_BLUR>), grid, 1024, c->stream, targets, n_targets, draws, (const WrPrim*)S.prims,
                (const WrRec*)S.recs, (const WrAux*)S.aux, (const float*)S.vtab, S.masks, ch, &c->dcounters->chain_arrive, c->dcounters);
    prof_end(4, H.fmt, chain_n, H.feat, bytes, (uint64_t)grid);
    c->stats.kernel_launches++; c->stats.raster_launches++;
    return;
  }
  prof_begin();
  uint64_t thin_wgs = 0;                 // a thin launch (R = 1 instantiation): its workgroups; reported as kind 12 (not mixed into the R = 4 rows)
  const bool fused = SA != nullptr;      // (SA still set: this raster launch carries the next flush's setup stage)
  static const bool thin_carrier = getenv("WRHIP_FUSE_SMALL") == nullptr;      // (can_fuse's thin_first)
  if (H.dense && H.fmt == WR_FMT_RGBA8 && H.feat == F7) {
    // (glyph levels: the 128-VGPR instantiation of the same body, plain or with the next flush's setup stage in front)
#define WR_KD(DEPTH)                                                                                                         \
  do {                                                                                                                       \
    if (SA) WR_LAUNCH((wr_setup_raster_dense_kernel<WR_FMT_RGBA8, DEPTH, 4, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX>), \
                      n_setup_blocks + SA->up_blocks + H.nb, 256, c->stream, *SA, n_setup_blocks, targets, n_targets, draws,                 \
                      (const WrPrim*)S.prims, (const WrRec*)S.recs, (const WrAux*)S.aux, (const float*)S.vtab, S.masks, H.off); \
    else WR_LAUNCH((wr_raster_dense_kernel<WR_FMT_RGBA8, DEPTH, 4, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX>), H.nb, 256, \
                   c->stream, targets, n_targets, draws, (const WrPrim*)S.prims, (const WrRec*)S.recs, (const WrAux*)S.aux,  \
                   (const float*)S.vtab, S.masks, H.off);                                                                    \
  } while (0)
    if (H.depth) WR_KD(true); else WR_KD(false);
#undef WR_KD
  }
  else if (SA && H.fmt == WR_FMT_RGBA8 && !H.depth && c->thin_r8 && H.nb <= WR_THIN_MAX_BINS && H.feat == F5 && thin_carrier) {
    // the thin colour launch with the next flush's setup stage in front (can_fuse)
    thin_wgs = (uint64_t)H.nb * 4 + (uint64_t)n_setup_blocks;
    WR_LAUNCH((wr_setup_raster_thin_kernel<WR_FMT_RGBA8, false, 1, WR_FEAT_TEX | WR_FEAT_GENERIC>), n_setup_blocks + SA->up_blocks + H.nb * 4, 256, c->stream, *SA,
              n_setup_blocks, targets, n_targets, draws, (const WrPrim*)S.prims, (const WrRec*)S.recs, (const WrAux*)S.aux, (const float*)S.vtab, S.masks, H.off);
  }
  else if (SA) {
    if (H.depth) {
      if (H.feat == 0) WR_KF(true, 0); else if (H.feat == F5) WR_KF(true, WR_FEAT_TEX | WR_FEAT_GENERIC);
      else WR_KF(true, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX);
    } else {
      if (H.feat == 0) WR_KF(false, 0); else if (H.feat == F5) WR_KF(false, WR_FEAT_TEX | WR_FEAT_GENERIC);
      else WR_KF(false, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX);
    }
  }
  else if (H.fmt == WR_FMT_RGBA8 && !H.depth && c->thin_r8 && H.nb <= WR_THIN_MAX_BINS && H.feat != 0 && H.feat != F7) {
    thin_wgs = (uint64_t)H.nb * 4;
    // Small colour launches without depth (a picture's blur chain: cs_scale halvings and cs_blur passes down to a single bin):
    // as for the thin mask launches, a bin's sixteen strips go to four workgroups of four waves of 64 x 4 pixels instead of one
    // workgroup of four waves of 64 x 16 (wrench large-blur-radius: 139 us for the one-bin blur passes)
    if (H.feat == F5)
      WR_LAUNCH((wr_raster_kernel<WR_FMT_RGBA8, false, 1, WR_FEAT_TEX | WR_FEAT_GENERIC>), H.nb * 4, 256, c->stream, targets, n_targets, draws,
                (const WrPrim*)S.prims, (const WrRec*)S.recs, (const WrAux*)S.aux, (const float*)S.vtab, S.masks, H.off);
    else
      WR_LAUNCH((wr_raster_kernel<WR_FMT_RGBA8, false, 1, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX | WR_FEAT_BLUR | WR_FEAT_SHADE>), H.nb * 4, 256, c->stream,
                targets, n_targets, draws, (const WrPrim*)S.prims, (const WrRec*)S.recs, (const WrAux*)S.aux, (const float*)S.vtab, S.masks, H.off);
  }
  else if (H.fmt == WR_FMT_RGBA8) {
    if (H.depth) {
      if (H.feat == 0) WR_K(WR_FMT_RGBA8, true, 0); else if (H.feat == F5) WR_K(WR_FMT_RGBA8, true, WR_FEAT_TEX | WR_FEAT_GENERIC);
      else if (H.feat == F7) WR_K(WR_FMT_RGBA8, true, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX);
      else WR_K(WR_FMT_RGBA8, true, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX | WR_FEAT_BLUR | WR_FEAT_SHADE);
    } else {
      if (H.feat == 0) WR_K(WR_FMT_RGBA8, false, 0); else if (H.feat == F5) WR_K(WR_FMT_RGBA8, false, WR_FEAT_TEX | WR_FEAT_GENERIC);
      else if (H.feat == F7) WR_K(WR_FMT_RGBA8, false, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX);
      else WR_K(WR_FMT_RGBA8, false, WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX | WR_FEAT_BLUR | WR_FEAT_SHADE);
    }
    (void)FA;
  } else {
    // Small mask launches (blur / down-scale passes of a few dozen bins) leave most of the chip idle and are bound by one
    // wave's critical path: there a bin is given 16 waves of 64 x 4 pixels (4 px per lane) instead of 4 waves of 64 x 16.
    const bool thin = c->thin_r8 && H.nb <= WR_THIN_MAX_BINS;
    if (thin && (H.feat == 0 || H.feat == (WR_FEAT_GENERIC | WR_FEAT_BLUR))) thin_wgs = (uint64_t)H.nb * c->thin_parts;
#define WR_K1(FEAT)                                                                                                     \
  do {                                                                                                                  \
    WR_LAUNCH((wr_raster_kernel<WR_FMT_R8, false, 1, FEAT>), H.nb * c->thin_parts, 1024 / c->thin_parts, c->stream, targets, n_targets, draws,          \
              (const WrPrim*)S.prims, (const WrRec*)S.recs, (const WrAux*)S.aux, (const float*)S.vtab, S.masks, H.off); \
  } while (0)
    if (H.feat == 0) { if (thin) WR_K1(0); else WR_K(WR_FMT_R8, false, 0); }
    else if (H.feat == (WR_FEAT_GENERIC | WR_FEAT_BLUR)) { if (thin) WR_K1(WR_FEAT_GENERIC | WR_FEAT_BLUR); else WR_K(WR_FMT_R8, false, WR_FEAT_GENERIC | WR_FEAT_BLUR); }
    else WR_K(WR_FMT_R8, false, WR_FEAT_GENERIC | WR_FEAT_BLUR | WR_FEAT_CLIP);
#undef WR_K1
  }
#undef WR_K
#undef WR_KF
  // (5: wr_raster_dense_kernel; 6 / 7: the same two with the next flush's setup stage in front, wr_setup_raster[_dense]_kernel)
  prof_end(thin_wgs ? (fused ? 13 : 12) : fused ? (H.dense ? 7 : 6) : (H.dense ? 5 : 2), H.fmt, H.depth, H.feat, H.algo_bytes + (fused ? setup_bytes : 0),
           thin_wgs ? thin_wgs : (uint64_t)H.nb + (fused ? n_setup_blocks : 0));
  c->stats.kernel_launches++; c->stats.raster_launches++;
}
// A flush's raster launches in order; runs of chainable() launches of one variant (only the first may have mask rows: its rows
// launch goes ahead of the chain) leave as one launch.  `fuse_at`: the launch that carries the next flush's setup stage.
void launch_held(const std::vector<Context::Held>& held, const WrTargetDesc* targets, int n_targets, const WrDrawDesc* draws, Context::Scratch& S,
                 int fuse_at = -1, const WrSetupArgs* SA = nullptr, int n_setup_blocks = 0, uint64_t setup_bytes = 0) {
  for (size_t i = 0; i < held.size();) {
    size_t j = i + 1;
    if (chainable(held[i]) && !(S.mr_ctl == nullptr && held[i].mr_rows > 0))
      while (j < held.size() && j - i < WR_MAX_CHAIN && chainable(held[j]) && held[j].feat == held[i].feat && held[j].mr_rows == 0 && (int)j != fuse_at) j++;
    const bool fuse = (int)i == fuse_at;
    launch_raster(held[i], targets, n_targets, draws, S, fuse ? SA : nullptr, fuse ? n_setup_blocks : 0, (int)(j - i), fuse ? setup_bytes : 0);
    i = j;
  }
}
// Launch the held-back raster launches on their own (nothing to fuse them with, or their results are needed now).
void drain_tail() {
  Context* c = ctx;
  if (!c || !c->tail.pending) return;
  Context::Tail& T = c->tail;
  launch_held(T.held, T.targets, T.n_targets, T.draws, c->scratch[T.set]);
  tail_launched();
}
void sync_stream() {
  drain_tail();
  HostTimer ht(&ctx->stats.host_wait_ns);
  wrrt::stream_sync(ctx->stream);
}

// Forwarded composites.  A target whose pending work is [full clear,] N draws of `composite FAST_PATH` that each copy a whole
// RGBA8 target of the same flush, unblended, texel for pixel (integer placement, optional y flip through the projection),
// disjoint and together covering it completely -- the window of a frame whose picture-cache tiles were all redrawn -- does
// not need a raster pass of its own: the tile pass stores each tile's pixels there as well (WrTargetDesc::fwd_*), which saves
// the composite's read of every tile (4 B per window pixel) and a launch.  Anything else falls back to the ordinary path.
static bool plan_forwarding(TargetWork& fb, const std::vector<int>& sel) {
  Context* c = ctx;
  Texture& ft = c->textures[fb.tex];
  if (ft.internal_format != GL_RGBA8 || c->profiling_no_forward) return false;
  // multi-GPU: this rank owns rows [oy0, oy1) of the window (WrhipSetTargetRows) and only those have to be covered -- by the tiles
  // that were kept because they reach into the strip, rasterised over the bin rows that do (a bin row straddling the strip's edge
  // also writes rows of the neighbour's strip: the exchange that follows on the stream overwrites them)
  const bool strip = ft.own_y1 > ft.own_y0;
  const int oy0 = strip ? std::max(0, ft.own_y0) : 0, oy1 = strip ? std::min(ft.height, ft.own_y1) : ft.height;
  if (oy1 <= oy0) return false;
  struct Plan { int src; int dx, y0, ys, clip[4]; };
  std::vector<Plan> plans;
  bool seen_draw = false;
  for (const WrDrawDesc& d : fb.draws) {
    if (d.shader == WR_SH_CLEAR_OP) {
      const bool full = d.clip[0] <= 0 && d.clip[1] <= 0 && d.clip[2] >= ft.width && d.clip[3] >= ft.height;
      if (seen_draw || !full || (d.flags & WR_DF_CLEAR_DEPTH)) return false;
      continue;
    }
    seen_draw = true;
    if (d.shader != WR_SH_COMPOSITE_FAST || d.blend != WR_BLEND_NONE || (d.flags & (WR_DF_DEPTH_TEST | WR_DF_DEPTH_WRITE | WR_DF_TEX_RECT)) || d.query_slot >= 0) return false;
    if (d.attr_off[0] < 0 || d.attr_off[1] < 0 || d.attr_off[4] < 0 || d.attr_bytes[0] < 16 || d.attr_bytes[1] < 16 || d.attr_bytes[4] < 16) return false;
    if (d.quad[0] != 0.f || d.quad[1] != 0.f || d.quad[2] != 1.f || d.quad[3] != 0.f || d.quad[4] != 1.f || d.quad[5] != 1.f || d.quad[6] != 0.f || d.quad[7] != 1.f) return false;
    const WrTexDesc& st = d.tex[WR_S_COLOR0];
    if (!st.ptr || st.format != WR_FMT_RGBA8) return false;
    int src = -1;
    // (a source that was itself forwarded away has no raster pass to store from, and one that already forwards into another
    // target has its one write-through taken: both take the ordinary composite path)
    for (int wi : sel) if (&c->work[wi] != &fb && c->textures[c->work[wi].tex].dptr == st.ptr && c->work[wi].level < fb.level && !c->work[wi].forwarded_away) src = wi;
    if (src < 0 && !strip) return false;      // (a strip: the draw may lie wholly in other ranks' rows, its source dropped when it was recorded)
    if (src >= 0 && c->work[src].fwd_tex && c->work[src].fwd_tex != fb.tex) return false;
    for (int i = 0; i < d.count; i++) {
      const uint8_t* ip = fb.inst.data() + d.inst_offset + (size_t)i * d.inst_stride;
      float rect[4], clip[4], prm[4] = {0, 0, 0, 0}, uv[4], flip[2] = {0, 0};
      memcpy(rect, ip + d.attr_off[0], 16); memcpy(clip, ip + d.attr_off[1], 16); memcpy(uv, ip + d.attr_off[4], 16);
      if (d.attr_off[3] >= 0 && d.attr_bytes[3] >= 16) memcpy(prm, ip + d.attr_off[3], 16);
      if (d.attr_off[5] >= 0 && d.attr_bytes[5] >= 8) memcpy(flip, ip + d.attr_off[5], 8);
      if (flip[0] != 0.f || flip[1] != 0.f || int(prm[1]) == 1 || uv[0] != 0.f || uv[1] != 0.f || uv[2] != 1.f || uv[3] != 1.f) return false;
      // device space -> target pixels through the draw's projection and viewport, as the vertex stage + draw_quad do it
      auto to_px = [&](float x, float y, float& sx, float& sy) {
        const float* m = d.transform;
        const float gx = m[0] * x + m[4] * y + m[12], gy = m[1] * x + m[5] * y + m[13], gw = m[3] * x + m[7] * y + m[15];
        sx = (gx / gw + 1.0f) * 0.5f * d.vp_size[0] + d.vp_origin[0];
        sy = (gy / gw + 1.0f) * 0.5f * d.vp_size[1] + d.vp_origin[1];
      };
      if (d.transform[1] != 0.f || d.transform[4] != 0.f || d.transform[3] != 0.f || d.transform[7] != 0.f) return false;   // axis-aligned, affine
      float x0, y0, x1, y1, cx0, cy0, cx1, cy1;
      to_px(rect[0], rect[1], x0, y0); to_px(rect[2], rect[3], x1, y1);
      to_px(clip[0], clip[1], cx0, cy0); to_px(clip[2], clip[3], cx1, cy1);
      auto near_int = [](float v, int& o) { o = (int)lrintf(v); return fabsf(v - (float)o) < 1.0f / 64.0f; };
      int ix0, iy0, ix1, iy1, icx0, icy0, icx1, icy1;
      if (!near_int(x0, ix0) || !near_int(y0, iy0) || !near_int(x1, ix1) || !near_int(y1, iy1) || !near_int(cx0, icx0) || !near_int(cy0, icy0) ||
          !near_int(cx1, icx1) || !near_int(cy1, icy1)) return false;
      if (strip && std::min(std::max(icy0, icy1), oy1) <= std::max(std::min(icy0, icy1), oy0)) continue;      // (none of this rank's rows)
      if (strip && std::min(std::max(iy0, iy1), oy1) <= std::max(std::min(iy0, iy1), oy0)) continue;
      if (src < 0) return false;
      const Texture& stex = c->textures[c->work[src].tex];
      if (ix1 - ix0 != stex.width || abs(iy1 - iy0) != stex.height) return false;          // texel for pixel
      for (const Plan& q : plans) if (q.src == src) return false;                            // one destination per source
      Plan P;
      P.src = src; P.dx = ix0;
      if (iy1 > iy0) { P.y0 = iy0; P.ys = 1; } else { P.y0 = iy0 - 1; P.ys = -1; }
      P.clip[0] = std::max(std::max(std::min(icx0, icx1), d.clip[0]), std::max(0, std::min(ix0, ix1)));
      P.clip[1] = std::max(std::max(std::min(icy0, icy1), d.clip[1]), std::max(0, std::min(iy0, iy1)));
      P.clip[2] = std::min(std::min(std::max(icx0, icx1), d.clip[2]), std::min(ft.width, std::max(ix0, ix1)));
      P.clip[3] = std::min(std::min(std::max(icy0, icy1), d.clip[3]), std::min(ft.height, std::max(iy0, iy1)));
      if (P.clip[2] <= P.clip[0] || P.clip[3] <= P.clip[1]) { P.clip[0] = P.clip[1] = P.clip[2] = P.clip[3] = 0; }
      plans.push_back(P);
    }
  }
  if (plans.empty()) return false;
  // disjoint and complete
  long long area = 0;
  for (size_t a = 0; a < plans.size(); a++) {
    const int* A = plans[a].clip;
    area += (long long)(A[2] - A[0]) * std::max(0, std::min(A[3], oy1) - std::max(A[1], oy0));
    for (size_t b = a + 1; b < plans.size(); b++) {
      const int* B = plans[b].clip;
      if (A[0] < B[2] && B[0] < A[2] && A[1] < B[3] && B[1] < A[3]) return false;
    }
  }
  if (area != (long long)ft.width * (oy1 - oy0)) return false;
  for (const Plan& P : plans) {
    TargetWork& sw = c->work[P.src];
    sw.fwd_tex = fb.tex; sw.fwd_dx = P.dx; sw.fwd_y0 = P.y0; sw.fwd_ys = P.ys;
    for (int k = 0; k < 4; k++) sw.fwd_clip[k] = P.clip[k];
  }
  fb.forwarded_away = true;
  return true;
}

void flush_work(const std::vector<int>& sel_in) {
  Context* c = ctx;
  if (!c || c->work.empty() || sel_in.empty()) return;
  HostTimer ht(&c->stats.host_flush_ns);
  static const bool laps = getenv("WRHIP_DEBUG_LAPS") != nullptr;
  auto lap_t = std::chrono::steady_clock::now();
  auto lap = [&](int i) { if (laps) { const auto n = std::chrono::steady_clock::now(); c->lap_ns[i] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(n - lap_t).count(); lap_t = n; } };
  if (laps) c->lap_n++;
  std::vector<int> sel(sel_in);
  std::sort(sel.begin(), sel.end());
  sel.erase(std::unique(sel.begin(), sel.end()), sel.end());
  // forwarded composites: targets that turn into write-throughs of their sources leave the launch sequence
  std::vector<GLuint> forwarded_dst;
  if (c->forward_composites) {
    for (int wi : sel) { TargetWork& w = c->work[wi]; w.fwd_tex = 0; w.forwarded_away = false; }
    for (int wi : sel) {
      TargetWork& w = c->work[wi];
      bool all_comp = !w.draws.empty();
      for (const WrDrawDesc& d : w.draws) if (d.shader != WR_SH_CLEAR_OP && d.shader != WR_SH_COMPOSITE_FAST) { all_comp = false; break; }
      if (all_comp && plan_forwarding(w, sel)) forwarded_dst.push_back(w.tex);
    }
    if (!forwarded_dst.empty()) {
      std::vector<int> keep;
      for (int wi : sel) if (!c->work[wi].forwarded_away) keep.push_back(wi);
      // (the forwarded targets stay in sel_in's hazard bookkeeping below through `forwarded_sel`)
      sel.swap(keep);
    }
  }
  std::vector<int> forwarded_sel;
  for (int wi : sel_in) if (c->work[wi].forwarded_away) forwarded_sel.push_back(wi);
  // Span-rows targets (wr_span_rows_kernel): every draw is a cs_blur / cs_scale pass or a clear, unblended, without depth --
  // the levels of a blur chain.  They get no bins; a wave per target row walks the target's few prims instead.
  auto rows_eligible = [&](const TargetWork& w) {
    if (!c->span_rows || w.fwd_tex || w.forwarded_away || w.depth_tex) return false;
    const Texture& t = c->textures[w.tex];
    if (t.internal_format != GL_R8 && t.internal_format != GL_RGBA8) return false;
    int nprim = 0; bool any = false;
    for (const WrDrawDesc& d : w.draws) {
      if (d.shader == WR_SH_CLEAR_OP) { if (d.flags & WR_DF_CLEAR_DEPTH) return false; nprim += 1; continue; }
      if (d.shader != WR_SH_CS_BLUR_ALPHA && d.shader != WR_SH_CS_BLUR_COLOR && d.shader != WR_SH_CS_SCALE) return false;
      if (d.blend != WR_BLEND_NONE || (d.flags & (WR_DF_DEPTH_TEST | WR_DF_DEPTH_WRITE | WR_DF_QUADS | WR_DF_SIMPLE | WR_DF_XFORM))) return false;
      any = true; nprim += d.count;
    }
    return any && nprim <= 48;
  };
  // Tile-rows targets (wr_tile_rows_kernel): a picture target of a few large axis-aligned prims, at least one of them a linear
  // gradient or an image -- the kinds whose bin-raster variant runs at one wave per SIMD.  Solids only where the host has verified
  // them plain (WR_DF_SIMPLE), nothing that may sit on a general quad or ask for anti-aliasing, no depth that outlives the flush.
  auto tile_rows_eligible = [&](const TargetWork& w) {
    if (!c->tile_rows || w.forwarded_away) return false;
    const Texture& t = c->textures[w.tex];
    if (t.internal_format != GL_RGBA8) return false;
    if (Texture* dt = w.depth_tex ? c->textures.find(w.depth_tex) : nullptr) {
      if (dt->depth_materialized) return false;
      if (w.depth_live && dt->depth_cleared && dt->depth_owner == w.tex) return false;      // (would have to be stored per pixel)
    }
    int nprim = 0; bool heavy = false;
    for (const WrDrawDesc& d : w.draws) {
      if (d.shader == WR_SH_CLEAR_OP) { nprim += 1; continue; }
      const bool solid = d.shader == WR_SH_BRUSH_SOLID || d.shader == WR_SH_BRUSH_SOLID_ALPHA || d.shader == WR_SH_PS_QUAD_TEXTURED;
      const bool shade = d.shader == WR_SH_BRUSH_IMAGE || d.shader == WR_SH_BRUSH_IMAGE_ALPHA || d.shader == WR_SH_BRUSH_LINEAR_GRADIENT ||
                         d.shader == WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA;
      if (!solid && !shade) return false;
      // (solids under a clip mask -- the corner segments of a rounded-rect clip, wrench large-clip-rect: 72 prims on two tiles are one
      // serial walk per wave in the bin raster -- : the host has scanned such a draw for anti-aliasing requests (WR_DF_QUADS), so its
      // prims are plain or masked solids)
      const bool masked_solid = (d.shader == WR_SH_BRUSH_SOLID || d.shader == WR_SH_BRUSH_SOLID_ALPHA) && d.tex[WR_S_CLIP_MASK].ptr != nullptr &&
                                d.tex[WR_S_CLIP_MASK].format == WR_FMT_R8;
      if (solid && !(d.flags & WR_DF_SIMPLE) && !masked_solid) return false;
      if (d.flags & (WR_DF_QUADS | WR_DF_XFORM | WR_DF_TEX_RECT)) return false;
      if (d.blend == WR_BLEND_UNSUPPORTED || d.blend == WR_BLEND_DUAL_SRC) return false;
      if (d.query_slot >= 0) return false;
      heavy = heavy || shade || (masked_solid && !(d.flags & WR_DF_SIMPLE));
      nprim += d.count;
    }
    return heavy && nprim <= 128;
  };
  std::vector<char> rows_of(c->work.size(), 0);
  for (int wi : sel) rows_of[wi] = rows_eligible(c->work[wi]) ? 1 : (tile_rows_eligible(c->work[wi]) ? 2 : 0);
  // by dependency level, RGBA8 targets first inside a level, span-rows targets last, so each launch gets a contiguous bin / target range
  std::stable_sort(sel.begin(), sel.end(), [&](int a, int b) {
    const int la = c->work[a].level, lb = c->work[b].level;
    if (la != lb) return la < lb;
    const int ka = rows_of[a] ? 1 + rows_of[a] : (c->textures[c->work[a].tex].internal_format == GL_R8 ? 1 : 0);
    const int kb = rows_of[b] ? 1 + rows_of[b] : (c->textures[c->work[b].tex].internal_format == GL_R8 ? 1 : 0);
    return ka < kb;
  });
  struct Level { int bin0 = 0, bins_rgba = 0, bins_r8 = 0, feat_rgba = 0, feat_r8 = 0; bool any_depth = false, text = false; uint64_t bytes_rgba = 0, bytes_r8 = 0, mr_rows = 0;
                 int row_t0 = -1, row_n = 0, row_items = 0, row_fmt = 0; uint64_t bytes_rows = 0;          // span-rows targets (rows_mode 1)
                 int trow_t0 = -1, trow_n = 0, trow_items = 0; uint64_t bytes_trows = 0; };                // tile-rows targets (rows_mode 2)
  int n_row_targets = 0;
  uint64_t mr_slots = 0, mr_rows = 0, mr_bytes = 0;      // bounds on what the cs_clip_* prims of this flush can reserve in the mask-row store
  std::vector<Level> levels;
  std::vector<int> target_level;
  std::vector<size_t> flat_off;          // per target: offset of its flattened-depth-row table (words), or SIZE_MAX
  size_t flat_words = 0;
  const int n_targets = (int)sel.size();
  std::vector<WrDrawDesc> draws;
  std::vector<WrTargetDesc> targets(n_targets);
  struct InstSeg { size_t base; const uint8_t* src; size_t size; };
  std::vector<InstSeg> inst_segs;      // instance bytes of the selected targets, in arena order (copied at staging time)
  size_t inst_bytes = 0;
  int prim_cursor = 0, bin_cursor = 0, word_cursor = 0;
  size_t vtab_cursor = 0;
  size_t gtab_words = 0;       // the pool's fixed head: the gradient-table copies of the WR_DF_GTAB draws (WrDrawDesc::gtab_base); the setup stage's
                               // allocations start behind it
  size_t qtab_need = 0;        // floats: rows x instances x 10 of the draws that may hold rotated / projected prims (WR_DF_XFORM)
  const bool no_qtab = !c->quad_rowtabs;
  static const bool no_run_share = getenv("WRHIP_NO_RUN_SHARE") != nullptr;
  const size_t runs_pool_words = c->runs_pool_words;
  bool runs_pool = false;
  uint64_t algo_bytes = 0, pixels = 0;
  for (int oi = 0; oi < n_targets; oi++) {
    TargetWork& w = c->work[sel[oi]];
    Texture& t = c->textures[w.tex];
    WrTargetDesc& T = targets[oi];
    memset(&T, 0, sizeof(T));
    if (levels.empty() || w.level != c->work[sel[oi - 1]].level) { levels.emplace_back(); levels.back().bin0 = bin_cursor; }
    Level& L = levels.back();
    target_level.push_back((int)levels.size() - 1);
    T.color = t.dptr; T.width = t.width; T.height = t.height; T.stride = t.stride;
    T.format = t.internal_format == GL_R8 ? WR_FMT_R8 : WR_FMT_RGBA8;
    T.bins_x = (t.width + WR_BIN_W - 1) / WR_BIN_W; T.bins_y = (t.height + WR_BIN_H - 1) / WR_BIN_H;
    // every target starts on a 64-prim boundary: a wave of the setup kernel then holds prims of one
    // target and one mask word only (the slots in between are skipped as instance >= count)
    prim_cursor = (prim_cursor + 63) & ~63;
    T.first_bin = bin_cursor; T.first_prim = prim_cursor;
    T.load_color = 1; T.init_color = 0; T.load_depth = 0; T.store_depth = 0;
    Texture* dt = w.depth_tex ? c->textures.find(w.depth_tex) : nullptr;
    T.init_depth = w.init_depth_set ? w.init_depth : (dt ? dt->depth_value : 0xFFFFFF);
    if (dt && dt->depth_materialized && dt->dptr) { T.load_depth = 1; T.depth = (uint32_t*)dt->dptr; }
    T.fwd_color = nullptr;
    if (w.fwd_tex) {
      Texture& ft = c->textures[w.fwd_tex];
      T.fwd_color = ft.dptr; T.fwd_stride = ft.stride; T.fwd_dx = w.fwd_dx; T.fwd_y0 = w.fwd_y0; T.fwd_ys = w.fwd_ys;
      for (int k = 0; k < 4; k++) T.fwd_clip[k] = w.fwd_clip[k];
    }
    T.cells = c->cell_raster ? 1 : 0;
    T.rows_mode = rows_of[sel[oi]];
    T.y_begin = 0; T.y_end = t.height;
    if (t.own_y1 > t.own_y0) {   // WrhipSetTargetRows: rows of this target owned by this process
      T.y_begin = std::max(0, t.own_y0); T.y_end = std::min(t.height, t.own_y1);
    } else if (c->shard_world > 1) {  // WrhipSetShard: contiguous strips of bin rows per rank
      T.y_begin = (int)((int64_t)T.bins_y * c->shard_rank / c->shard_world) * WR_BIN_H;
      T.y_end = std::min(t.height, (int)((int64_t)T.bins_y * (c->shard_rank + 1) / c->shard_world) * WR_BIN_H);
    }
    // (the target's instance bytes go from its snapshot straight into the staging ring when the arena is laid out: gathering
    // them in one vector first was a second pass over 2.4 MB per cfg5 frame)
    size_t inst_base = (inst_bytes + 15) & ~size_t(15);
    inst_bytes = inst_base + w.inst.size();
    if (!w.inst.empty()) inst_segs.push_back(InstSeg{inst_base, w.inst.data(), w.inst.size()});
    bool any_kept = false;
    T.dw_first = 0; T.dw_end = 0;
    for (const WrDrawDesc& d0 : w.draws) {
      WrDrawDesc d = d0;
      d.target = oi;
      d.inst_offset += inst_base;
      // Leading full-target clears initialise the bins instead of loading HBM.
      if (!any_kept && d.shader == WR_SH_CLEAR_OP && d.clip[0] <= 0 && d.clip[1] <= 0 && d.clip[2] >= t.width &&
          d.clip[3] >= t.height) {
        if (d.flags & WR_DF_CLEAR_COLOR) { T.load_color = 0; T.init_color = d.clear_color; }
        if (d.flags & WR_DF_CLEAR_DEPTH) { T.init_depth = d.clear_depth; T.load_depth = 0; }
        continue;
      }
      any_kept = true;
      if (c->mask_rows && T.format == WR_FMT_R8 && !(d.flags & WR_DF_SIMPLE) &&
          (d.shader == WR_SH_CS_CLIP_RECT || d.shader == WR_SH_CS_CLIP_RECT_FAST || d.shader == WR_SH_CS_CLIP_BOX_SHADOW)) {
        const int cw = std::max(0, std::min(d.clip[2], t.width) - std::max(d.clip[0], 0));
        const int ch = std::max(0, std::min(d.clip[3], t.height) - std::max(d.clip[1], 0));
        if (cw > 0 && ch > 0 && d.count > 0) {
          d.flags |= WR_DF_MASK_ROWS;
          mr_slots += (uint64_t)d.count; mr_rows += (uint64_t)d.count * ch * 1;      // work items: rows x parts
          mr_bytes += (uint64_t)d.count * ((uint64_t)((cw + 7) & ~3) * ch + (uint64_t)ch * 4 + 48 + sizeof(WrAccTabs));      // rows + row map + row-sum tables
          L.mr_rows += (uint64_t)d.count * ch * 1;
        }
      }
      if ((d.flags & (WR_DF_DEPTH_TEST | WR_DF_CLEAR_DEPTH)) && T.format == WR_FMT_RGBA8) L.any_depth = true;
      // A wave of the setup stage runs every vertex shader its 64 prims need one after the other (divergence), and the setup
      // stage of a small flush is one long dependent chain per wave: while the flush is small, a draw with another program
      // starts a wave of its own (cfg4: the box-shadow and the clip-out rect of a mask target, 2 x 12 us in one wave).
      if (prim_cursor < 4096 && (prim_cursor & 63) && !draws.empty() && draws.back().target == oi && draws.back().shader != d.shader)
        prim_cursor = (prim_cursor + 63) & ~63;
      d.first_prim = prim_cursor;
      if ((d.flags & WR_DF_DEPTH_WRITE) && d.shader != WR_SH_CLEAR_OP) {
        if (T.dw_end <= T.dw_first) T.dw_first = prim_cursor;
        T.dw_end = prim_cursor + d.count;
      }
      prim_cursor += d.count;
      // per-row v table budget for draws whose prims can take the nearest-fast texture path
      d.gtab_base = -1;
      if (d.flags & WR_DF_GTAB) {
        d.gtab_base = (int)gtab_words; gtab_words += (size_t)d.count * WR_GTAB_WORDS;
        // (the promise of THIS draw is redeemed; those of draws a partial flush leaves recorded stay counted: ADVICE r5)
        c->gtab_pending -= std::min(c->gtab_pending, (size_t)d.count * WR_GTAB_WORDS);
      }
      d.vtab_base = -1; d.vtab_rows = 0;
      if (T.format == WR_FMT_RGBA8 && !(d.flags & WR_DF_SIMPLE) &&
          (d.shader == WR_SH_COMPOSITE || d.shader == WR_SH_COMPOSITE_FAST || d.shader == WR_SH_CS_SCALE ||
           d.shader == WR_SH_PS_QUAD_TEXTURED || d.shader == WR_SH_BRUSH_IMAGE || d.shader == WR_SH_BRUSH_IMAGE_ALPHA)) {
        const int rows = std::max(0, std::min(d.clip[3], t.height) - std::max(d.clip[1], 0));
        const size_t need = (size_t)rows * d.count;
        if (rows > 0 && vtab_cursor + need <= ((size_t)64 << 20)) {
          d.vtab_base = (int)vtab_cursor; d.vtab_rows = rows;
          vtab_cursor += need;
        }
      }
      if (!no_qtab && T.format == WR_FMT_RGBA8 && (d.flags & WR_DF_XFORM) && !(d.flags & WR_DF_SIMPLE) && d.shader != WR_SH_CLEAR_OP) {
        const int rows = std::max(0, std::min(d.clip[3], t.height) - std::max(d.clip[1], 0));
        qtab_need += (size_t)rows * (size_t)d.count * 10;
      }
      // (the same pool takes the depth runs of a row and the occluder list of a strip that outgrow their LDS copies: a draw that is
      // depth-tested and may consume interpolants can ask for them)
      if (T.format == WR_FMT_RGBA8 && (d.flags & WR_DF_DEPTH_TEST) && !(d.flags & WR_DF_SIMPLE) && d.shader != WR_SH_CLEAR_OP) runs_pool = true;
      draws.push_back(d);
    }
    T.end_prim = prim_cursor;
    {
      // a general-quad draw with the depth test on may hold perspective prims: they flatten the depth rows they touch
      bool persp_possible = false;
      if (T.format == WR_FMT_RGBA8)
        for (const WrDrawDesc& d0 : w.draws) if ((d0.flags & WR_DF_XFORM) && (d0.flags & WR_DF_DEPTH_TEST) && !(d0.flags & WR_DF_SIMPLE)) { persp_possible = true; break; }
      flat_off.push_back(persp_possible ? flat_words : SIZE_MAX);
      if (persp_possible) flat_words += (size_t)t.height + 1;
    }
    int nrel = T.end_prim - T.first_prim;
    T.words_per_bin = (nrel + 63) / 64;
    T.word_base = word_cursor;
    if (T.rows_mode == 2) {
      T.words_per_bin = 0;
      if (L.trow_n == 0) L.trow_t0 = oi;
      L.trow_n++; n_row_targets++;
      L.trow_items += std::max(0, T.y_end - T.y_begin) * ((t.width + 255) >> 8);
    } else if (T.rows_mode) {
      // no bins, no mask words: the level's span-rows launch takes targets [row_t0, row_t0 + row_n)
      T.words_per_bin = 0;
      if (L.row_n == 0) { L.row_t0 = oi; L.row_fmt = T.format; }
      L.row_n++; n_row_targets++;
      L.row_items += std::max(0, T.y_end - T.y_begin) * WR_SPAN_PIECES(t.width);
    } else {
      word_cursor += T.words_per_bin * T.bins_x * T.bins_y;
      bin_cursor += T.bins_x * T.bins_y;
      (T.format == WR_FMT_RGBA8 ? L.bins_rgba : L.bins_r8) += T.bins_x * T.bins_y;
    }
    uint64_t owned = (uint64_t)t.width * std::max(0, T.y_end - T.y_begin);
    pixels += owned;
    {
      uint64_t tb = owned * t.bpp * (T.load_color ? 2 : 1);
      // unique source texels sampled: bounded by what a 1:1 mapping can touch
      uint64_t src = 0;
      for (GLuint id : w.reads)
        if (Texture* rt = c->textures.find(id))
          if (rt->internal_format == GL_RGBA8 || rt->internal_format == GL_R8) src += (uint64_t)rt->stride * rt->height;
      tb += std::min<uint64_t>(src, owned * t.bpp);
      if (w.fwd_tex) tb += (uint64_t)std::max(0, w.fwd_clip[2] - w.fwd_clip[0]) * std::max(0, w.fwd_clip[3] - w.fwd_clip[1]) * 4;   // the write-through
      algo_bytes += tb;
      (T.rows_mode == 2 ? L.bytes_trows : T.rows_mode ? L.bytes_rows : (T.format == WR_FMT_RGBA8 ? L.bytes_rgba : L.bytes_r8)) += tb;
    }
    if (dt && nrel > 0 && w.depth_live && dt->depth_cleared && dt->depth_owner == w.tex) {
      // The caller has not invalidated (or fully cleared) the depth these draws leave behind: it outlives the flush -- a
      // flush in the middle of a target (host upload to a sampled texture, readback of the target being drawn, a
      // TIME_ELAPSED query) -- so it is materialised: stored per pixel now, loaded by the continuation.  WebRender's own
      // pattern (clear, draw, InvalidateFramebuffer, flush later) never gets here.
      bool uses = T.load_depth != 0;
      for (const WrDrawDesc& d0 : w.draws) if (d0.flags & (WR_DF_DEPTH_WRITE | WR_DF_CLEAR_DEPTH)) uses = true;
      if (uses) {
        const size_t need = (size_t)t.width * t.height * 4;
        if (!dt->dptr || dt->dsize < need || dt->width != t.width || dt->height != t.height) {
          // (depth attachments have the size of their colour target in every caller; anything else keeps the uniform value)
          if (dt->width == t.width && dt->height == t.height) {
            if (dt->dptr) pool_free(dt->dptr, dt->dsize);
            size_t actual = 0;
            dt->dptr = pool_alloc(need, &actual); dt->dsize = actual;
          }
        }
        if (dt->dptr && dt->dsize >= need && dt->width == t.width && dt->height == t.height) {
          T.depth = (uint32_t*)dt->dptr; T.store_depth = 1;
          dt->depth_materialized = true;
          L.any_depth = true;
        }
      }
    }
    if (T.load_depth) L.any_depth = true;
  }
  const int n_prims = prim_cursor, n_bins = bin_cursor, n_words = word_cursor;
  const int nd = (int)draws.size();
  const bool any_work = n_bins > 0 || n_row_targets > 0;
  if (any_work) {
    // the prim arrays of this flush's scratch set, sized here: the address of its glyph records goes into the target descriptors
    Context::Scratch& S = c->scratch[c->flush_seq & 1];
    if (S.prims_cap < (size_t)n_prims + 1) {
      sync_stream();       // (drains the tail: nothing in flight references the buffers being replaced)
      wrrt::dev_free(S.prims); wrrt::dev_free(S.recs); wrrt::dev_free(S.aux);
      S.prims_cap = (size_t)(n_prims + 1) * 2;
      S.prims = (WrPrim*)wrrt::dev_alloc(S.prims_cap * sizeof(WrPrim));
      S.recs = (WrRec*)wrrt::dev_alloc(S.prims_cap * (sizeof(WrRec) + sizeof(WrGlyphRec)));      // recs[], then the glyph records (WrTargetDesc::grecs)
      S.aux = (WrAux*)wrrt::dev_alloc(S.prims_cap * sizeof(WrAux));
    }
    if (S.bin_ctr_cap < (size_t)n_bins || !S.bin_ctr) {
      sync_stream();
      wrrt::dev_free(S.bin_ctr);
      S.bin_ctr_cap = (size_t)std::max(n_bins, 64) * 2;
      S.bin_ctr = (unsigned*)wrrt::dev_alloc(S.bin_ctr_cap * sizeof(unsigned));
      wrrt::memset8(S.bin_ctr, 0, S.bin_ctr_cap * sizeof(unsigned), c->stream);      // (the workgroups leave them at zero)
    }
    // the flush's pool: the gradient-table copies (fixed head, <= 256 MB of promises), what the flush's WR_DF_XFORM draws could ask for in
    // row tables (<= 64 MB), and the depth runs' share (WRHIP_RUNS_POOL_WORDS, 64 MB by default)
    const size_t qtab_want = gtab_words + std::min<size_t>(qtab_need, (size_t)16 << 20) + (runs_pool ? runs_pool_words : 0);
    if (S.qtab_cap < qtab_want) {
      sync_stream();
      wrrt::dev_free(S.qtab);
      S.qtab_cap = qtab_want + qtab_want / 4;
      S.qtab = (float*)wrrt::dev_alloc(S.qtab_cap * sizeof(float));
    }
    for (WrTargetDesc& T : targets) {
      T.grecs = (const WrGlyphRec*)(S.recs + S.prims_cap); T.bin_ctr = S.bin_ctr + T.first_bin;
      T.qtab = qtab_want ? S.qtab : nullptr; T.qtab_cap = (uint32_t)std::min<size_t>(S.qtab_cap, (size_t)1 << 30); T.qtab_ctl = nullptr; T.qtab_pad = (no_qtab ? 1u : 0u) | (no_run_share ? 2u : 0u);
    }
  }
  // Mask-row store.  When the bounds fit, no reservation of the setup stage can fail and the R8 launches take the light
  // variant (the rows kernel evaluates, the bins blend bytes); otherwise the store is capped, prims that do not fit keep their
  // in-raster evaluation and the launches the variant that has it.
  bool mr_on = mr_slots > 0, mr_safe = false;
  if (mr_on) {
    Context::Scratch& S = c->scratch[c->flush_seq & 1];
    const uint64_t kCap = (uint64_t)1 << 30;
    mr_safe = mr_slots <= WR_MR_MAX_SLOTS && mr_rows <= WR_MR_MAX_ROWS && mr_bytes <= kCap;
    const size_t want_slots = (size_t)std::min<uint64_t>(mr_slots, WR_MR_MAX_SLOTS);
    const size_t want_bytes = (size_t)std::min<uint64_t>(mr_bytes, mr_safe ? kCap : ((uint64_t)256 << 20));
    if (S.mr_slots_cap < want_slots || S.mr_store_cap < want_bytes) {
      sync_stream();
      if (S.mr_slots_cap < want_slots) {
        wrrt::dev_free(S.mr_slots);
        S.mr_slots_cap = std::min<size_t>(want_slots * 2, WR_MR_MAX_SLOTS);
        S.mr_slots = (WrMaskSlot*)wrrt::dev_alloc(S.mr_slots_cap * sizeof(WrMaskSlot));
      }
      if (S.mr_store_cap < want_bytes) {
        wrrt::dev_free(S.mr_store);
        S.mr_store_cap = (want_bytes + 4095) & ~size_t(4095);
        S.mr_store = (uint8_t*)wrrt::dev_alloc(S.mr_store_cap);
      }
    }
    for (WrTargetDesc& T : targets) {
      if (T.format != WR_FMT_R8) continue;
      T.mr_ctl = nullptr; T.mr_slots = S.mr_slots; T.mr_store = S.mr_store;       // (mr_ctl: 512 bytes of this flush's arena, below)
      T.mr_cap16 = (uint32_t)std::min<uint64_t>(S.mr_store_cap >> 4, WR_MR_MAX_CAP16);
      T.mr_max_slots = (uint32_t)S.mr_slots_cap;
    }
  }
  for (WrTargetDesc& T : targets) { T.flat_rows = nullptr; T.counters = c->dcounters; }
  static const bool no_flat = getenv("WRHIP_NO_FLAT") != nullptr;      // (debugging: leave flattened depth rows unmodelled)
  if (flat_words && any_work && !no_flat) {
    Context::Scratch& S = c->scratch[c->flush_seq & 1];
    if (S.flat_cap < flat_words) {
      sync_stream();
      wrrt::dev_free(S.flat);
      S.flat_cap = flat_words * 2;
      S.flat = (uint32_t*)wrrt::dev_alloc(S.flat_cap * 4);
    }
    wrrt::memset8(S.flat, 0xFF, flat_words * 4, c->stream);          // (this set's previous user, two flushes back, has been launched)
    for (int oi = 0; oi < n_targets; oi++) if (flat_off[oi] != SIZE_MAX) targets[oi].flat_rows = S.flat + flat_off[oi];
  }
  if (any_work) {
    lap(0);      // planning: forwarding, levels, per-target descriptors
    // ---- frame arena: [draws | targets | instance bytes] -> one H2D copy ----
    size_t off_draws = 0;
    size_t off_targets = (off_draws + sizeof(WrDrawDesc) * nd + 255) & ~size_t(255);
    size_t off_inst = (off_targets + sizeof(WrTargetDesc) * n_targets + 255) & ~size_t(255);
    // first draw of every 64-prim block (wr_vertex_prim starts its draw lookup there)
    const int n_blocks = (n_prims + 63) / 64;
    size_t off_blk = (off_inst + inst_bytes + 255) & ~size_t(255);
    const size_t off_qctl = (off_blk + sizeof(WrBlock) * n_blocks + 63) & ~size_t(63);      // the row-table pool's allocation word: zero on arrival
    // ... and the mask-row store's control block ([0] allocation word, [32..63] byte counters): part of the arena as well, so that it
    // arrives zeroed with the arena's DMA instead of by a fill launch of its own per flush (4.7 us of stream time and a runtime call
    // per frame of every workload with clip masks)
    const size_t off_mrctl = off_qctl + 64;
    size_t total = off_mrctl + 512 + 256;
    size_t aoff = staging_alloc(total);
    uint8_t* h = c->staging + aoff;
    memset(h + off_qctl, 0, 64 + 512);
    if (mr_on) {
      Context::Scratch& Sm = c->scratch[c->flush_seq & 1];
      Sm.mr_ctl = (unsigned long long*)(c->dupload + aoff + off_mrctl); Sm.mr_seen = 0;
      for (WrTargetDesc& T : targets) if (T.format == WR_FMT_R8 && T.mr_slots) T.mr_ctl = Sm.mr_ctl;
    }
    *(unsigned long long*)(h + off_qctl) = (unsigned long long)gtab_words;      // (WR_GTAB_WORDS is a multiple of 4: the pieces behind stay on 16 bytes)
    for (WrTargetDesc& T : targets) T.qtab_ctl = T.qtab ? (unsigned long long*)(c->dupload + aoff + off_qctl) : nullptr;
    c->last_qctl = nullptr;
    for (const WrTargetDesc& T : targets) if (T.qtab) { c->last_qctl = T.qtab_ctl; c->last_qtab_cap = T.qtab_cap; break; }
    if (nd) stage_copy(h + off_draws, draws.data(), sizeof(WrDrawDesc) * nd);
    stage_copy(h + off_targets, targets.data(), sizeof(WrTargetDesc) * n_targets);
    if (inst_bytes >= PARALLEL_COPY_MIN && inst_segs.size() > 1) {
      // many targets' worth of instances (cfg5: 72 snapshots of ~33 KB): the helpers take whole segments
      copy_pool().run([&](int part, int parts) {
        for (size_t i = (size_t)part; i < inst_segs.size(); i += (size_t)parts) stage_copy(h + off_inst + inst_segs[i].base, inst_segs[i].src, inst_segs[i].size);
      });
    } else {
      for (const InstSeg& sg : inst_segs) {
        if (sg.size >= PARALLEL_COPY_MIN) big_memcpy(h + off_inst + sg.base, sg.src, sg.size);
        else stage_copy(h + off_inst + sg.base, sg.src, sg.size);
      }
    }
    {
      // (... and the draws and targets the block's wave stages in LDS: WrDescView)
      WrBlock* blk = (WrBlock*)(h + off_blk);
      int j = 0;
      for (int b = 0; b < n_blocks; b++) {
        while (j + 1 < nd && draws[j + 1].first_prim <= 64 * b) j++;
        WrBlock B;
        B.lo0 = j; B.nk = nd - j >= 2 ? 2 : nd - j;
        B.t0 = B.nk > 0 ? draws[j].target : -1; B.t1 = B.nk > 1 ? draws[j + 1].target : -1;
        blk[b] = B;
      }
    }
    // the launch that will carry this flush's setup stage (if any): it also runs the batch's scatter, in its first workgroups
    int fuse_at = -1;
    if (n_prims > 0 && c->tail.pending && !c->setup_on_stream) {
      // the launch that hides the setup stage best: the largest mask-rows launch if there is one, else the first raster
      // launch the fused kernel has a variant for
      int best_rows = 0;
      for (size_t hi = 0; hi < c->tail.held.size(); hi++)
        if (c->tail.held[hi].mr_rows > best_rows) { best_rows = c->tail.held[hi].mr_rows; fuse_at = (int)hi; }
      // (... else the largest tile-rows launch: a wave per row piece of a few heavy prims, the long launch of such a flush -- behind
      // the short rect pass of the same flush a setup stage would stick out)
      if (fuse_at < 0) {
        int best_items = 0;
        for (size_t hi = 0; hi < c->tail.held.size(); hi++) {
          const Context::Held& Hh = c->tail.held[hi];
          if (Hh.row_n > 0 && Hh.row_mode == 2 && Hh.row_items > best_items && can_fuse(Hh)) { best_items = Hh.row_items; fuse_at = (int)hi; }
        }
      }
      for (size_t hi = 0; hi < c->tail.held.size() && fuse_at < 0; hi++) if (can_fuse(c->tail.held[hi])) fuse_at = (int)hi;
    }
    lap(1);      // arena staged
    flush_uploads(0, fuse_at >= 0);     // one DMA: queued texture uploads + this arena; then the scatter (its own launch, or the carrier's first workgroups)
    uint8_t* darena = c->dupload + aoff;
    c->stats.h2d_bytes += total;
    algo_bytes += inst_bytes + sizeof(WrDrawDesc) * nd;
    lap(2);      // DMA + scatter queued
    // ---- scratch (two sets: the deferred tail of the previous flush still reads the other one) ----
    Context::Scratch& S = c->scratch[c->flush_seq & 1];
    if (S.vtab_cap < vtab_cursor + 1 || S.masks_cap < (size_t)n_words + 1) {      // (prims / recs / aux: sized above)
      sync_stream();       // (drains the tail: nothing in flight references the buffers being replaced)
      if (S.vtab_cap < vtab_cursor + 1) {
        wrrt::dev_free(S.vtab);
        S.vtab_cap = (vtab_cursor + 1) * 2;
        S.vtab = (float*)wrrt::dev_alloc(S.vtab_cap * sizeof(float));
      }
      if (S.masks_cap < (size_t)n_words + 1) {
        wrrt::dev_free(S.masks);
        S.masks_cap = (size_t)(n_words + 1) * 2;
        S.masks = (unsigned long long*)wrrt::dev_alloc(S.masks_cap * 8);
        wrrt::memset8(S.masks, 0, S.masks_cap * 8, c->stream);   // raster workgroups re-zero what they consume
      }
    }
#ifdef WRHIP_HOSTSIM
    wrrt::memset8(S.masks, 0, (size_t)n_words * 8, c->stream);
#endif
    const WrDrawDesc* ddraws = (const WrDrawDesc*)(darena + off_draws);
    const WrTargetDesc* dtargets = (const WrTargetDesc*)(darena + off_targets);
    const uint8_t* dinst = darena + off_inst;
    const int* dblk = (const int*)(darena + off_blk);
    if (n_prims > 0) {
#ifdef WRHIP_TIMING
      static const int setup_mode = getenv("WRHIP_SETUP_MODE") ? atoi(getenv("WRHIP_SETUP_MODE")) : 0;
      const int nd_arg = nd | (setup_mode << 24);
#else
      const int nd_arg = nd;
#endif
      const int n_setup_blocks = (n_prims + 255) / 256;
      const uint64_t setup_bytes = inst_bytes + sizeof(WrDrawDesc) * nd + (uint64_t)n_prims * (sizeof(WrPrim) + sizeof(WrRec));
      if (c->tail.pending && c->setup_on_stream) {
        wrrt::event_record(&c->ev_up, c->stream);                 // (this flush's uploads and table resets are enqueued)
        wrrt::stream_wait_event(c->setup_stream, &c->ev_up);
        prof_begin(&c->setup_stream);
        WR_LAUNCH(wr_setup_kernel, n_setup_blocks, 256, c->setup_stream, ddraws, nd_arg, dinst, S.prims, S.recs, S.aux, n_prims,
                  dtargets, S.masks, S.vtab, c->dcounters, dblk);
        prof_end(1, 0, 0, 0, setup_bytes, (uint64_t)n_setup_blocks, &c->setup_stream);
        wrrt::event_record(&c->ev_setup, c->setup_stream);
        c->stats.kernel_launches += 1;
        drain_tail();                                            // the previous flush's raster launches, concurrently
        wrrt::stream_wait_event(c->stream, &c->ev_setup);
      } else {
      if (fuse_at >= 0) {
        // the previous flush's held-back raster launches, in order; the first one the fused kernel has a
        // variant for (normally the tile pass, the longest) carries this flush's setup stage along
        Context::Tail& T = c->tail;
        WrSetupArgs SA{ddraws, nd_arg, dinst, S.prims, S.recs, S.aux, n_prims, dtargets, S.masks, S.vtab, c->dcounters, dblk, nullptr, 0, 0, 0};
        uint64_t carried = setup_bytes;
        if (c->ps.valid) {
          SA.up_segs = (const WrUploadSeg*)(c->dupload + c->ps.seg_off); SA.up_nseg = c->ps.nseg; SA.up_parts = c->ps.parts; SA.up_blocks = c->ps.nseg * c->ps.parts;
          carried += c->ps.bytes;
          c->ps.valid = false;
        }
        launch_held(T.held, T.targets, T.n_targets, T.draws, c->scratch[T.set], fuse_at, &SA, n_setup_blocks, carried);
        tail_launched();
      } else {
        prof_begin();
        WR_LAUNCH(wr_setup_kernel, n_setup_blocks, 256, c->stream, ddraws, nd_arg, dinst, S.prims, S.recs, S.aux, n_prims,
                  dtargets, S.masks, S.vtab, c->dcounters, dblk);
        prof_end(1, 0, 0, 0, setup_bytes, (uint64_t)n_setup_blocks);
        c->stats.kernel_launches += 1;
        drain_tail();        // (held-back launches the fused kernel has no variant for)
      }
      }
    } else {
      drain_tail();
    }
    launch_pending_scatter();      // (nothing is pending unless a carrier was planned and not used)
    lap(3);      // setup + held launches queued
#ifdef WRHIP_HOSTSIM
    if (getenv("WRHIP_DEBUG")) {
      fprintf(stderr, "flush: targets %d draws %d prims %d bins %d words %d\n", n_targets, nd, n_prims, n_bins, n_words);
      for (int i = 0; i < n_targets; i++)
        fprintf(stderr, "  T%d %dx%d fmt %d load %d init %08x prims [%d,%d) wpb %d\n", i, targets[i].width, targets[i].height,
                targets[i].format, targets[i].load_color, targets[i].init_color, targets[i].first_prim, targets[i].end_prim,
                targets[i].words_per_bin);
      for (int i = 0; i < nd && i < 6; i++)
        fprintf(stderr, "  D%d sh %d tgt %d first %d n %d blend %d flags %x clip %d %d %d %d vp %g %g %g %g\n", i, draws[i].shader,
                draws[i].target, draws[i].first_prim, draws[i].count, draws[i].blend, draws[i].flags, draws[i].clip[0],
                draws[i].clip[1], draws[i].clip[2], draws[i].clip[3], draws[i].vp_origin[0], draws[i].vp_origin[1],
                draws[i].vp_size[0], draws[i].vp_size[1]);
      for (int i = 0; i < nd && i < 6; i++)
        fprintf(stderr, "     inst off %llu stride %d attr_off %d %d %d %d %d %d bytes %d %d quad %g %g %g %g %g %g %g %g\n",
                (unsigned long long)draws[i].inst_offset, draws[i].inst_stride, draws[i].attr_off[0], draws[i].attr_off[1],
                draws[i].attr_off[2], draws[i].attr_off[3], draws[i].attr_off[4], draws[i].attr_off[5], draws[i].attr_bytes[0],
                draws[i].attr_bytes[1], draws[i].quad[0], draws[i].quad[1], draws[i].quad[2], draws[i].quad[3], draws[i].quad[4],
                draws[i].quad[5], draws[i].quad[6], draws[i].quad[7]);
      for (int i = 0; i < nd && i < 6; i++) {
        const float* f = (const float*)(dinst + draws[i].inst_offset);
        fprintf(stderr, "     flush inst D%d: %g %g %g %g (inst bytes %zu)\n", i, f[0], f[1], f[2], f[3], inst_bytes);
      }
      for (int i = 0; i < n_prims && i < 6; i++) {
        const WrPrim& P = S.prims[i];
        fprintf(stderr, "  P%d kind %d rect %d %d %d %d z %u blend %d flags %x color %08x %08x\n", i, P.kind, P.x0, P.y0, P.x1, P.y1,
                P.z, P.blend, P.flags, P.color[0], P.color[1]);
      }
    }
#endif
    // Four waves per 64x64 bin, each lane owning 4 x 4 pixels (R = 4; two waves x
    // 32 pixels measured slower on MI355X, profiles/r01_*).  The kernel is
    // specialised on the prim families present in the launch (FEAT) so that
    // rect-only passes do not pay the registers of the texture paths.
    for (int i = 0; i < nd; i++) {
      if (targets[draws[i].target].rows_mode) continue;          // (span-rows targets are in no bin launch)
      const bool to_r8 = targets[draws[i].target].format == WR_FMT_R8;
      Level& L = levels[target_level[draws[i].target]];
      int f = 0;
      if (!(draws[i].flags & WR_DF_SIMPLE)) {
        switch (draws[i].shader) {
          case WR_SH_PS_CLEAR: case WR_SH_CLEAR_OP: break;
          case WR_SH_PS_TEXT_RUN: case WR_SH_PS_TEXT_RUN_DUAL: case WR_SH_PS_TEXT_RUN_GT: case WR_SH_PS_TEXT_RUN_DUAL_GT: f = WR_FEAT_R8TEX | WR_FEAT_TEX | WR_FEAT_GENERIC; break;
          case WR_SH_BRUSH_SOLID: case WR_SH_BRUSH_SOLID_ALPHA: f = WR_FEAT_R8TEX | WR_FEAT_GENERIC; break;   // masked / odd blend
          case WR_SH_CS_BLUR_ALPHA: case WR_SH_CS_BLUR_COLOR: f = WR_FEAT_BLUR; break;
          case WR_SH_CS_CLIP_RECT: case WR_SH_CS_CLIP_RECT_FAST: case WR_SH_CS_CLIP_BOX_SHADOW:
            f = ((draws[i].flags & WR_DF_MASK_ROWS) && mr_safe) ? WR_FEAT_BLUR : WR_FEAT_CLIP; break;
          case WR_SH_BRUSH_LINEAR_GRADIENT: case WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA: case WR_SH_CS_LINEAR_GRADIENT: case WR_SH_CS_RADIAL_GRADIENT: case WR_SH_CS_CONIC_GRADIENT: f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          case WR_SH_BRUSH_BLEND: case WR_SH_BRUSH_BLEND_ALPHA: f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          case WR_SH_BRUSH_MIX_BLEND: case WR_SH_BRUSH_MIX_BLEND_ALPHA: f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          case WR_SH_CS_SVG_FILTER: case WR_SH_CS_SVG_FILTER_NODE: f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          case WR_SH_BRUSH_YUV: case WR_SH_BRUSH_YUV_ALPHA: case WR_SH_COMPOSITE_YUV: f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          case WR_SH_PS_QUAD_MASK: case WR_SH_PS_QUAD_MASK_FAST: case WR_SH_PS_QUAD_RADIAL_GRADIENT: case WR_SH_PS_QUAD_CONIC_GRADIENT:
            f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          case WR_SH_CS_BORDER_SOLID: case WR_SH_CS_BORDER_SEGMENT: case WR_SH_CS_FAST_LINEAR_GRADIENT: case WR_SH_CS_LINE_DECORATION:
            f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          case WR_SH_BRUSH_IMAGE_REPEAT: case WR_SH_BRUSH_IMAGE_REPEAT_ALPHA: case WR_SH_BRUSH_IMAGE_REPEAT_DUAL: f = WR_FEAT_SHADE | WR_FEAT_GENERIC; break;
          default: f = WR_FEAT_TEX | WR_FEAT_GENERIC; break;
        }
      }
      if (draws[i].flags & WR_DF_QUADS) f |= WR_FEAT_SHADE | WR_FEAT_GENERIC;
      (to_r8 ? L.feat_r8 : L.feat_rgba) |= f;
      if (!to_r8 && !(draws[i].flags & WR_DF_SIMPLE) && (draws[i].shader == WR_SH_PS_TEXT_RUN || draws[i].shader == WR_SH_PS_TEXT_RUN_DUAL || draws[i].shader == WR_SH_PS_TEXT_RUN_GT || draws[i].shader == WR_SH_PS_TEXT_RUN_DUAL_GT)) L.text = true;
    }
    // one raster launch per dependency level and target format, in level order on the one stream: the
    // smallest instantiated superset of each level's feature set.  They are held back (Context::Tail)
    // unless the flush is being profiled or deferral is off.
    std::vector<Context::Held> launches;
    for (const Level& L : levels) {
      if (L.bins_rgba > 0) {
        int f;
        if (L.feat_rgba == 0) f = 0;
        else if (!(L.feat_rgba & ~(WR_FEAT_TEX | WR_FEAT_GENERIC))) f = WR_FEAT_TEX | WR_FEAT_GENERIC;
        else if (!(L.feat_rgba & ~(WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX))) f = WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX;
        else f = WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX | WR_FEAT_BLUR | WR_FEAT_SHADE;
        launches.push_back(Context::Held{WR_FMT_RGBA8, L.any_depth ? 1 : 0, f, L.bins_rgba, L.bin0, L.bytes_rgba, 0, (L.text && c->dense_text && f == (WR_FEAT_TEX | WR_FEAT_GENERIC | WR_FEAT_R8TEX)) ? 1 : 0});
      }
      if (L.bins_r8 > 0) {
        const int f = L.feat_r8 == 0 ? 0 : (!(L.feat_r8 & WR_FEAT_CLIP) ? (WR_FEAT_GENERIC | WR_FEAT_BLUR) : (WR_FEAT_GENERIC | WR_FEAT_BLUR | WR_FEAT_CLIP));
        launches.push_back(Context::Held{WR_FMT_R8, 0, f, L.bins_r8, L.bin0 + L.bins_rgba, L.bytes_r8, (int)std::min<uint64_t>(L.mr_rows, WR_MR_MAX_ROWS)});
      }
      if (L.row_n > 0) {
        Context::Held H{L.row_fmt, 0, 0, 0, 0, L.bytes_rows};
        H.row_t0 = L.row_t0; H.row_n = L.row_n; H.row_items = L.row_items; H.row_mode = 1;
        launches.push_back(H);
      }
      if (L.trow_n > 0) {
        Context::Held H{WR_FMT_RGBA8, 0, 0, 0, 0, L.bytes_trows};
        H.row_t0 = L.trow_t0; H.row_n = L.trow_n; H.row_items = L.trow_items; H.row_mode = 2;
        launches.push_back(H);
      }
    }
    if (c->defer_tail && (!c->profiling || c->profiling_deferred) && !launches.empty()) {
      Context::Tail& T = c->tail;      // (the previous tail went out with this flush's setup launch)
      T.pending = true; T.held = launches; T.n_targets = n_targets;
      T.targets = dtargets; T.draws = ddraws; T.set = (int)(c->flush_seq & 1);
      for (int oi = 0; oi < n_targets; oi++) {
        const TargetWork& w = c->work[sel[oi]];
        T.refs.push_back(w.tex);
        if (w.fwd_tex) T.refs.push_back(w.fwd_tex);
        for (GLuint id : w.rreads) T.refs.push_back(id);
      }
      for (GLuint id : T.refs) if (Texture* t = c->textures.find(id)) t->tail_ref = true;
    } else {
      launch_held(launches, dtargets, n_targets, ddraws, S);
    }
    {
      // (for WrhipFlushHeld: which flushes wrote the window -- as a target, or as the destination of forwarded tile stores)
      const GLuint fbt = c->framebuffers.find(0) ? c->framebuffers[0].color_attachment : 0;
      bool wrote = false;
      for (int wi : sel_in) { const TargetWork& w = c->work[wi]; if (fbt && (w.tex == fbt || w.fwd_tex == fbt)) wrote = true; }
      if (wrote) c->fb_writes_since_held++;
      c->last_flush_wrote_fb = wrote;
    }
    ring_record_fence();
    c->flush_seq++;
    c->stats.flushes++;
    c->stats.prims += n_prims;
    c->stats.raster_pixels += pixels;
    c->stats.raster_algo_bytes += algo_bytes;
  }
  lap(4);        // this flush's own launches
  // ---- drop the flushed work, keep the rest, rebuild hazard flags ----------
  std::vector<char> gone(c->work.size(), 0);
  for (int i : sel) gone[i] = 1;
  for (int i : forwarded_sel) gone[i] = 1;
  std::vector<TargetWork> rest;
  for (size_t i = 0; i < c->work.size(); i++) {
    if (!gone[i]) rest.push_back(std::move(c->work[i]));
    else if (c->spare.size() < 256) { c->work[i].recycle(); c->spare.push_back(std::move(c->work[i])); }
  }
  c->work.swap(rest);
  for (GLuint id : c->referenced)
    if (Texture* t = c->textures.find(id)) { t->pending_read = t->pending_write = false; t->pending_target = -1; }
  c->referenced.clear();
  for (size_t i = 0; i < c->work.size(); i++) {
    mark_ref(c->work[i].tex, c->textures[c->work[i].tex], true, (int)i);
    for (GLuint id : c->work[i].reads) if (Texture* t = c->textures.find(id)) mark_ref(id, *t, false);
  }
  lap(5);        // bookkeeping
  if (laps && (c->lap_n % 1000) == 0) {
    fprintf(stderr, "libwrhip flush laps (us per flush, %llu flushes): plan %.2f arena %.2f uploads %.2f setup+held %.2f own launches %.2f bookkeeping %.2f\n",
            (unsigned long long)c->lap_n, c->lap_ns[0] / 1e3 / c->lap_n, c->lap_ns[1] / 1e3 / c->lap_n, c->lap_ns[2] / 1e3 / c->lap_n, c->lap_ns[3] / 1e3 / c->lap_n,
            c->lap_ns[4] / 1e3 / c->lap_n, c->lap_ns[5] / 1e3 / c->lap_n);
    memset(c->lap_ns, 0, sizeof(c->lap_ns)); c->lap_n = 0;
  }
}

void flush_all() {
  Context* c = ctx;
  if (!c || c->work.empty()) return;
  std::vector<int> sel(c->work.size());
  for (size_t i = 0; i < sel.size(); i++) sel[i] = (int)i;
  flush_work(sel);
  c->gtab_pending = 0;          // (nothing is left recorded: promises of draws that were dropped without a flush go with it)
}

}  // namespace

// ===========================================================================
// C ABI
extern "C" {

void UseProgram(GLuint program) {
  if (ctx->current_program && program != ctx->current_program) {
    Program* p = ctx->programs.find(ctx->current_program);
    if (p && p->deleted) ctx->programs.erase(ctx->current_program);
  }
  ctx->current_program = program;
}
void SetViewport(GLint x, GLint y, GLsizei width, GLsizei height) {
  ctx->viewport[0] = x; ctx->viewport[1] = y; ctx->viewport[2] = x + width; ctx->viewport[3] = y + height;
}
void Enable(GLenum cap) {
  switch (cap) { case GL_BLEND: ctx->blend = true; break; case GL_DEPTH_TEST: ctx->depthtest = true; break;
    case GL_SCISSOR_TEST: ctx->scissortest = true; break; }
}
void Disable(GLenum cap) {
  switch (cap) { case GL_BLEND: ctx->blend = false; break; case GL_DEPTH_TEST: ctx->depthtest = false; break;
    case GL_SCISSOR_TEST: ctx->scissortest = false; break; }
}
GLenum GetError(void) { GLenum e = ctx->last_error; ctx->last_error = GL_NO_ERROR; return e; }

static const char* const extensions[] = {
    "GL_ARB_blend_func_extended", "GL_ARB_clear_texture", "GL_ARB_copy_image", "GL_ARB_draw_instanced",
    "GL_ARB_explicit_attrib_location", "GL_ARB_instanced_arrays", "GL_ARB_invalidate_subdata",
    "GL_ARB_texture_storage", "GL_EXT_timer_query", "GL_KHR_blend_equation_advanced",
    "GL_KHR_blend_equation_advanced_coherent", "GL_APPLE_rgb_422"};

void GetIntegerv(GLenum pname, GLint* params) {
  switch (pname) {
    case GL_MAX_TEXTURE_UNITS: case GL_MAX_TEXTURE_IMAGE_UNITS: params[0] = MAX_TEXTURE_UNITS; break;
    case GL_MAX_TEXTURE_SIZE: params[0] = 1 << 15; break;
    case GL_MAX_ARRAY_TEXTURE_LAYERS: params[0] = 0; break;
    case GL_READ_FRAMEBUFFER_BINDING: params[0] = ctx->read_framebuffer_binding; break;
    case GL_DRAW_FRAMEBUFFER_BINDING: params[0] = ctx->draw_framebuffer_binding; break;
    case GL_PIXEL_PACK_BUFFER_BINDING: params[0] = ctx->pixel_pack_buffer_binding; break;
    case GL_PIXEL_UNPACK_BUFFER_BINDING: params[0] = ctx->pixel_unpack_buffer_binding; break;
    case GL_NUM_EXTENSIONS: params[0] = sizeof(extensions) / sizeof(extensions[0]); break;
    case GL_MAJOR_VERSION: params[0] = 3; break;
    case GL_MINOR_VERSION: params[0] = 2; break;
    case GL_MIN_PROGRAM_TEXEL_OFFSET: params[0] = 0; break;
    case GL_MAX_PROGRAM_TEXEL_OFFSET: params[0] = 8; break;
    default: break;
  }
}
void GetBooleanv(GLenum pname, GLboolean* params) { if (pname == GL_DEPTH_WRITEMASK) params[0] = ctx->depthmask; }
const char* GetString(GLenum name) {
  switch (name) {
    case GL_VENDOR: return "Mozilla Gfx";
    // Must start with "Software WebRender": webrender::Device keys its swgl
    // code paths (native clip masks / AA, immediate uploads) on it
    // (device/gl.rs:1645-1650, 1775-1779).
    case GL_RENDERER: return "Software WebRender (wrhip, AMD gfx950 HIP)";
    case GL_VERSION: return "3.2";
    case GL_SHADING_LANGUAGE_VERSION: return "1.50";
    default: return nullptr;
  }
}
const char* GetStringi(GLenum name, GLuint index) {
  if (name == GL_EXTENSIONS && index < sizeof(extensions) / sizeof(extensions[0])) return extensions[index];
  return nullptr;
}

void BlendFunc(GLenum srgb, GLenum drgb, GLenum sa, GLenum da) {
  ctx->blendfunc_srgb = srgb; ctx->blendfunc_drgb = drgb;
  ctx->blendfunc_sa = remap_blendfunc(srgb, sa); ctx->blendfunc_da = remap_blendfunc(drgb, da);
  ctx->blend_key = hash_blend_key(ctx);
}
void BlendColor(GLfloat r, GLfloat g, GLfloat b, GLfloat a) {
  // round_pixel((Float){b, g, r, a}) -> U16 (gl.cc:1336-1339)
  uint32_t bb = uint32_t(int(b * 255.0f + 0.5f)) & 0xFFFF, gg = uint32_t(int(g * 255.0f + 0.5f)) & 0xFFFF;
  uint32_t rr = uint32_t(int(r * 255.0f + 0.5f)) & 0xFFFF, aa = uint32_t(int(a * 255.0f + 0.5f)) & 0xFFFF;
  ctx->blendcolor[0] = bb | (gg << 16); ctx->blendcolor[1] = rr | (aa << 16);
}
void BlendEquation(GLenum mode) {
  if (mode != ctx->blend_equation) { ctx->blend_equation = mode; ctx->blend_key = hash_blend_key(ctx); }
}
void DepthMask(GLboolean flag) { ctx->depthmask = flag; }
void DepthFunc(GLenum func) { ctx->depthfunc = func; }
void SetScissor(GLint x, GLint y, GLsizei width, GLsizei height) {
  ctx->scissor[0] = x; ctx->scissor[1] = y; ctx->scissor[2] = x + width; ctx->scissor[3] = y + height;
}
void ClearColor(GLfloat r, GLfloat g, GLfloat b, GLfloat a) {
  ctx->clearcolor[0] = r; ctx->clearcolor[1] = g; ctx->clearcolor[2] = b; ctx->clearcolor[3] = a;
}
void ClearDepth(GLdouble depth) { ctx->cleardepth = depth; }
void ActiveTexture(GLenum texture) {
  int u = int(texture - GL_TEXTURE0);
  ctx->active_texture_unit = u < 0 ? 0 : (u > int(MAX_TEXTURE_UNITS - 1) ? int(MAX_TEXTURE_UNITS - 1) : u);
}

static inline void unlink(GLuint& binding, GLuint n) { if (binding == n) binding = 0; }

void GenQueries(GLsizei n, GLuint* result) { for (int i = 0; i < n; i++) result[i] = (GLuint)ctx->queries.insert(); }
void DeleteQuery(GLuint n) {
  if (n && ctx->queries.erase(n)) { unlink(ctx->time_elapsed_query, n); unlink(ctx->samples_passed_query, n); }
}
void GenBuffers(int32_t n, GLuint* result) { for (int i = 0; i < n; i++) result[i] = (GLuint)ctx->buffers.insert(); }
void DeleteBuffer(GLuint n) {
  if (n && ctx->buffers.erase(n)) {
    unlink(ctx->pixel_pack_buffer_binding, n); unlink(ctx->pixel_unpack_buffer_binding, n); unlink(ctx->array_buffer_binding, n);
  }
}
void GenVertexArrays(int32_t n, GLuint* result) { for (int i = 0; i < n; i++) result[i] = (GLuint)ctx->vertex_arrays.insert(); }
void DeleteVertexArray(GLuint n) { if (n && ctx->vertex_arrays.erase(n)) unlink(ctx->current_vertex_array, n); }

GLuint CreateShader(GLenum type) { GLuint id = (GLuint)ctx->shaders.insert(); ctx->shaders[id].type = type; return id; }
void ShaderSourceByName(GLuint shader, const GLchar* name) {
  Shader& s = ctx->shaders[shader];
  s.kind = WR_SH_NONE;
  snprintf(s.name, sizeof(s.name), "%s", name ? name : "");
  for (const ShaderInfo& info : SHADERS) if (!strcmp(info.key, name)) s.kind = info.kind;
}
void AttachShader(GLuint program, GLuint shader) {
  Program& p = ctx->programs[program];
  Shader& s = ctx->shaders[shader];
  if (!p.info && s.kind != WR_SH_NONE)
    for (const ShaderInfo& info : SHADERS) if (!strcmp(info.key, s.name)) p.info = &info;      // (several keys share a kind: TEXTURE_RECT, ADVANCED_BLEND)
  if (!p.name[0]) memcpy(p.name, s.name, sizeof(p.name));
}
void DeleteShader(GLuint n) { if (n) ctx->shaders.erase(n); }
GLuint CreateProgram(void) { return (GLuint)ctx->programs.insert(); }
void DeleteProgram(GLuint n) {
  if (!n) return;
  if (ctx->current_program == n) { if (Program* p = ctx->programs.find(n)) p->deleted = true; }
  else ctx->programs.erase(n);
}
void LinkProgram(GLuint program) { Program& p = ctx->programs[program]; if (p.info) p.linked = true; }
GLint GetLinkStatus(GLuint program) { Program* p = ctx->programs.find(program); return p && p->info ? 1 : 0; }
void BindAttribLocation(GLuint program, GLuint index, const GLchar* name) {
  Program& p = ctx->programs[program];
  if (!p.info) return;
  if (index >= NULL_ATTRIB) return;      // 16 attribute slots + the null attribute (gl.cc:577-580)
  for (int k = 0; k < WR_MAX_ATTRIBS + 1 && p.info->attribs[k]; k++) if (!strcmp(p.info->attribs[k], name)) { p.attrib_loc[k] = index; return; }
}
GLint GetAttribLocation(GLuint program, const GLchar* name) {
  Program& p = ctx->programs[program];
  if (!p.info) return -1;
  for (int k = 0; k < WR_MAX_ATTRIBS + 1 && p.info->attribs[k]; k++)
    if (!strcmp(p.info->attribs[k], name)) return p.attrib_loc[k] != NULL_ATTRIB ? p.attrib_loc[k] : -1;
  return -1;
}
GLint GetUniformLocation(GLuint program, const GLchar* name) {
  Program& p = ctx->programs[program];
  if (!p.info) return -1;
  if (!strcmp(name, "uTransform")) return p.info->kind == WR_SH_PS_COPY ? -1 : UNIFORM_TRANSFORM;     // (ps_copy's vertex stage has no uTransform)
  for (int s = 0; s < WR_MAX_TEX; s++)
    if (((p.info->samplers >> s) & 1) && !strcmp(SAMPLER_NAMES[s], name)) return s + 1;
  return -1;
}
void Uniform1i(GLint location, GLint v0) {
  Program* p = ctx->programs.find(ctx->current_program);
  if (p && location >= 1 && location <= WR_MAX_TEX) p->sampler_unit[location - 1] = v0;
}
void Uniform4fv(GLint, GLsizei, const GLfloat*) {}
void UniformMatrix4fv(GLint location, GLsizei, GLboolean, const GLfloat* value) {
  Program* p = ctx->programs.find(ctx->current_program);
  if (p && location == UNIFORM_TRANSFORM) memcpy(p->transform, value, sizeof(float) * 16);
}

void BeginQuery(GLenum target, GLuint id) {
  ctx->get_binding(target) = id;
  Query& q = ctx->queries[id];
  if (target == GL_SAMPLES_PASSED) {
    // shaded pixels of the draws issued inside the query (gl.cc:2784-2787), counted by the setup stage into a device slot
    q.value = 0;
    const int slot = ctx->next_query_slot;
    ctx->next_query_slot = (ctx->next_query_slot + 1) % WR_QUERY_SLOTS;
    // the slot's previous owner (64 queries ago, its result not read yet): its draws go out and its count is kept with the query
    // before the slot is zeroed for this one
    if (GLuint prev = ctx->query_slot_owner[slot]) {
      Query* pq = ctx->queries.find(prev);
      if (pq && pq->slot == slot && prev != id) {
        flush_all();
        unsigned long long v = 0;
        wrrt::d2h(&v, &ctx->dcounters->samples[slot], sizeof(v), ctx->stream);
        sync_stream();
        pq->value = v; pq->slot = -1;
      }
    }
    q.slot = slot; ctx->query_slot_owner[slot] = id;
    wrrt::memset8(&ctx->dcounters->samples[q.slot], 0, sizeof(unsigned long long), ctx->stream);
  } else if (target == GL_TIME_ELAPSED) {
    q.slot = -1;             // (an id that once counted samples: its result is the time from here on)
    // TIME_ELAPSED must cover the GPU work issued inside the query (renderer
    // GpuProfiler, device/query_gl.rs:141-163): drain what came before.
    flush_all(); sync_stream();
    q.value = get_time_value();
  }
}
void EndQuery(GLenum target) {
  Query& q = ctx->queries[ctx->get_binding(target)];
  if (target == GL_TIME_ELAPSED) {
    flush_all(); sync_stream();
    q.value = get_time_value() - q.value;
  }
  ctx->get_binding(target) = 0;
}
void GetQueryObjectui64v(GLuint id, GLenum pname, GLuint64* params) {
  if (pname != GL_QUERY_RESULT) return;
  Query& q = ctx->queries[id];
  if (q.slot >= 0) {
    flush_all();
    unsigned long long v = 0;
    wrrt::d2h(&v, &ctx->dcounters->samples[q.slot], sizeof(v), ctx->stream);
    sync_stream();
    q.value = v;
    if (ctx->samples_passed_query != id) q.slot = -1;      // ended: the slot may be reused
  }
  params[0] = q.value;
}

void BindVertexArray(GLuint vao) { ctx->current_vertex_array = vao; }
void BindTexture(GLenum target, GLuint texture) { ctx->get_binding(target) = texture; }
void BindBuffer(GLenum target, GLuint buffer) { ctx->get_binding(target) = buffer; }
void BindFramebuffer(GLenum target, GLuint fb) {
  if (target == GL_FRAMEBUFFER) { ctx->read_framebuffer_binding = fb; ctx->draw_framebuffer_binding = fb; }
  else ctx->get_binding(target) = fb;
}
void BindRenderbuffer(GLenum target, GLuint rb) { ctx->get_binding(target) = rb; }
void PixelStorei(GLenum name, GLint param) { if (name == GL_UNPACK_ROW_LENGTH) ctx->unpack_row_length = param; }

void TexStorage2D(GLenum target, GLint, GLenum internal_format, GLsizei width, GLsizei height) {
  Texture& t = ctx->textures[ctx->get_binding(target)];
  set_tex_storage(t, internal_format, width, height);
}

static void* pixel_unpack_data(const void* data) {
  if (ctx->pixel_unpack_buffer_binding) {
    Buffer& b = ctx->buffers[ctx->pixel_unpack_buffer_binding];
    return b.buf ? b.buf + (size_t)data : nullptr;
  }
  return (void*)data;
}
static void* pixel_pack_data(void* data) {
  if (ctx->pixel_pack_buffer_binding) {
    Buffer& b = ctx->buffers[ctx->pixel_pack_buffer_binding];
    return b.buf ? b.buf + (size_t)data : nullptr;
  }
  return data;
}

void TexSubImage2D(GLenum target, GLint level, GLint xoffset, GLint yoffset, GLsizei width, GLsizei height,
                   GLenum format, GLenum ty, const void* data_) {
  if (level != 0) return;
  HostTimer ht(&ctx->stats.host_upload_ns);
  const uint8_t* data = (const uint8_t*)pixel_unpack_data(data_);
  if (!data) return;
  Texture& t = ctx->textures[ctx->get_binding(target)];
  if (!t.dptr || width <= 0 || height <= 0) return;
  if (xoffset < 0 || yoffset < 0 || xoffset + width > t.width || yoffset + height > t.height) return;
  if (t.internal_format != internal_format_for_data(format, ty)) return;
  sync_texture_for_write(t);
  GLsizei row_length = ctx->unpack_row_length != 0 ? ctx->unpack_row_length : width;
  bool conv = format_requires_conversion(format, t.internal_format);
  size_t src_stride = (size_t)row_length * t.bpp;
  size_t row = (size_t)width * t.bpp;
  if (ctx->pixel_unpack_buffer_binding) {      // the rows must lie inside the bound pixel-unpack buffer
    const Buffer& pb = ctx->buffers[ctx->pixel_unpack_buffer_binding];
    const size_t off = (size_t)data_, need = (size_t)(height - 1) * src_stride + row;
    if (off > pb.size || need > pb.size - off) return;
  }
  order_upload((uint8_t*)t.dptr + (size_t)yoffset * t.stride + (size_t)xoffset * t.bpp, t.stride, row, height);
  size_t st_off = staging_alloc(row * height);
  uint8_t* st = ctx->staging + st_off;
  const bool ids = t.internal_format == GL_RGBA32I;
  std::atomic<int> id_flags{0};
  auto copy_rows = [&](int part, int parts) {
    const int ya = (int)((long long)height * part / parts), yb = (int)((long long)height * (part + 1) / parts);
    bool h = false, g = false;
    for (int y = ya; y < yb; y++) {
      const uint8_t* s = data + (size_t)y * src_stride;
      uint8_t* d = st + (size_t)y * row;
      if (conv) {  // GL_RGBA upload into BGRA storage: copy_bgra8_to_rgba8 (gl.cc:1649-1661)
        for (int x = 0; x < width; x++) {
          uint32_t p; memcpy(&p, s + 4 * x, 4);
          uint32_t rb = p & 0x00FF00FF;
          p = (p & 0xFF00FF00) | (rb << 16) | (rb >> 16);
          memcpy(d + 4 * x, &p, 4);
        }
      } else stage_copy(d, s, row);
      if (ids) {
        // (rows are whole texels of four ints, and an upload that starts mid-row keeps the two-texel phase: the uploaded
        // bytes are scanned as one array, as before -- the caller's copy: the staged one went past the cache)
        const int32_t* iv = (const int32_t*)s;
        const size_t i0 = (size_t)y * row / 4, ni = row / 4;
        for (size_t i = 0; i + 3 < ni; i += 4) {
          g |= ((uint32_t)iv[i] >> 23) != 0u;                                 // ps_quad header: [transform_id, z, pattern input]
          if (((i0 + i) & 4) == 0) h |= ((uint32_t)iv[i + 2] >> 23) != 0u;    // prim header texel 0: [z, specific, transform_id, task]
        }
      }
    }
    if (h || g) id_flags.fetch_or((h ? 1 : 0) | (g ? 2 : 0), std::memory_order_relaxed);
  };
  if (row * (size_t)height >= PARALLEL_COPY_MIN && height >= 8) copy_pool().run(copy_rows);
  else if (row * (size_t)height >= split_copy_min() && height >= 4) copy_pool().run(copy_rows, 2);
  else copy_rows(0, 1);
  if (ids) {
    if (xoffset == 0 && yoffset == 0) t.complex_ids_headers = t.complex_ids_gpubuf = false;   // (uploads start at the origin: a fresh frame)
    const int f = id_flags.load(std::memory_order_relaxed);
    t.complex_ids_headers |= (f & 1) != 0; t.complex_ids_gpubuf |= (f & 2) != 0;
  }
  queue_upload(st_off, (uint8_t*)t.dptr + (size_t)yoffset * t.stride + (size_t)xoffset * t.bpp, t.stride, row, height);
  t.up_batch = ctx->upload_batch;
  // (a data texture uploaded whole, its rows packed as the texture holds them: readable in the staging mirror)
  if (xoffset == 0 && yoffset == 0 && width == t.width && height == t.height && row == (size_t)t.stride && (st_off & 15) == 0 &&
      (t.internal_format == GL_RGBA32F || t.internal_format == GL_RGBA32I)) { t.view_batch = ctx->upload_batch; t.view_off = st_off; }
  else t.view_batch = 0;
}
void TexImage2D(GLenum target, GLint level, GLint internal_format, GLsizei width, GLsizei height, GLint, GLenum format,
                GLenum ty, const void* data) {
  if (level != 0) return;
  TexStorage2D(target, 1, internal_format, width, height);
  TexSubImage2D(target, 0, 0, 0, width, height, format, ty, data);
}
void GenerateMipmap(GLenum) {}
void SetTextureParameter(GLuint texid, GLenum pname, GLint param) {
  Texture& t = ctx->textures[texid];
  if (pname == GL_TEXTURE_MIN_FILTER) t.min_filter = param;
  else if (pname == GL_TEXTURE_MAG_FILTER) t.mag_filter = param;
}
void TexParameteri(GLenum target, GLenum pname, GLint param) { SetTextureParameter(ctx->get_binding(target), pname, param); }
void GenTextures(int32_t n, GLuint* result) { for (int i = 0; i < n; i++) result[i] = (GLuint)ctx->textures.insert(); }
void DeleteTexture(GLuint n) {
  if (!n) return;
  if (Texture* t = ctx->textures.find(n)) {
    free_texture_storage(*t);
    ctx->textures.erase(n);
    for (size_t i = 0; i < MAX_TEXTURE_UNITS; i++) {
      unlink(ctx->texture_units[i].texture_2d_binding, n);
      unlink(ctx->texture_units[i].texture_rectangle_binding, n);
    }
  }
}
void GenRenderbuffers(int32_t n, GLuint* result) { for (int i = 0; i < n; i++) result[i] = (GLuint)ctx->renderbuffers.insert(); }
void DeleteRenderbuffer(GLuint n) {
  if (!n) return;
  if (Renderbuffer* rb = ctx->renderbuffers.find(n)) {
    GLuint tex = rb->texture;
    for (Framebuffer* fb : ctx->framebuffers.objects)
      if (fb) { unlink(fb->color_attachment, tex); unlink(fb->depth_attachment, tex); }
    DeleteTexture(tex);
    ctx->renderbuffers.erase(n);
    unlink(ctx->renderbuffer_binding, n);
  }
}
void GenFramebuffers(int32_t n, GLuint* result) { for (int i = 0; i < n; i++) result[i] = (GLuint)ctx->framebuffers.insert(); }
void DeleteFramebuffer(GLuint n) {
  if (n && ctx->framebuffers.erase(n)) { unlink(ctx->read_framebuffer_binding, n); unlink(ctx->draw_framebuffer_binding, n); }
}
void RenderbufferStorage(GLenum target, GLenum internal_format, GLsizei width, GLsizei height) {
  Renderbuffer& r = ctx->renderbuffers[ctx->get_binding(target)];
  if (!r.texture) GenTextures(1, &r.texture);
  switch (internal_format) {
    case GL_DEPTH_COMPONENT: case GL_DEPTH_COMPONENT16: case GL_DEPTH_COMPONENT24: case GL_DEPTH_COMPONENT32:
      internal_format = GL_DEPTH_COMPONENT24; break;
  }
  set_tex_storage(ctx->textures[r.texture], internal_format, width, height);
}

static int bytes_per_type(GLenum type) {
  switch (type) { case GL_INT: case GL_FLOAT: return 4; case GL_UNSIGNED_SHORT: return 2; case GL_UNSIGNED_BYTE: return 1; default: return 0; }
}
void VertexAttribPointer(GLuint index, GLint size, GLenum type, GLboolean normalized, GLsizei stride, const GLvoid* offset) {
  if (index >= NULL_ATTRIB) return;
  VertexAttrib& va = ctx->vertex_arrays[ctx->current_vertex_array].attribs[index];
  va.size = size * bytes_per_type(type); va.type = type; va.normalized = normalized; va.stride = stride;
  va.offset = (GLuint)(uintptr_t)offset; va.vertex_buffer = ctx->array_buffer_binding;
}
void VertexAttribIPointer(GLuint index, GLint size, GLenum type, GLsizei stride, const GLvoid* offset) {
  if (index >= NULL_ATTRIB) return;
  VertexAttrib& va = ctx->vertex_arrays[ctx->current_vertex_array].attribs[index];
  va.size = size * bytes_per_type(type); va.type = type; va.normalized = false; va.stride = stride;
  va.offset = (GLuint)(uintptr_t)offset; va.vertex_buffer = ctx->array_buffer_binding;
}
void EnableVertexAttribArray(GLuint index) {
  if (index >= NULL_ATTRIB) return;
  ctx->vertex_arrays[ctx->current_vertex_array].attribs[index].enabled = true;
}
void VertexAttribDivisor(GLuint index, GLuint divisor) {
  if (index >= NULL_ATTRIB || divisor > 1) return;
  ctx->vertex_arrays[ctx->current_vertex_array].attribs[index].divisor = divisor;
}
void BufferData(GLenum target, GLsizeiptr size, const GLvoid* data, GLenum) {
  HostTimer ht(&ctx->stats.host_upload_ns);
  Buffer& b = ctx->buffers[ctx->get_binding(target)];
  if (size != b.size && !b.allocate(size)) out_of_memory();
  if (data && b.buf && size <= b.size) big_memcpy(b.buf, data, size);
}
void BufferSubData(GLenum target, GLintptr offset, GLsizeiptr size, const GLvoid* data) {
  HostTimer ht(&ctx->stats.host_upload_ns);
  Buffer& b = ctx->buffers[ctx->get_binding(target)];
  if (data && b.buf && offset + size <= b.size) big_memcpy(&b.buf[offset], data, size);
}
void* MapBuffer(GLenum target, GLbitfield) { return ctx->buffers[ctx->get_binding(target)].buf; }
void* MapBufferRange(GLenum target, GLintptr offset, GLsizeiptr length, GLbitfield) {
  Buffer& b = ctx->buffers[ctx->get_binding(target)];
  if (b.buf && offset >= 0 && length > 0 && offset + length <= b.size) return b.buf + offset;
  return nullptr;
}
GLboolean UnmapBuffer(GLenum target) { return ctx->buffers[ctx->get_binding(target)].buf != nullptr; }

void FramebufferTexture2D(GLenum target, GLenum attachment, GLenum, GLuint texture, GLint) {
  Framebuffer& fb = ctx->framebuffers[ctx->get_binding(target)];
  if (attachment == GL_COLOR_ATTACHMENT0) fb.color_attachment = texture;
  else if (attachment == GL_DEPTH_ATTACHMENT) fb.depth_attachment = texture;
}
void FramebufferRenderbuffer(GLenum target, GLenum attachment, GLenum, GLuint renderbuffer) {
  Framebuffer& fb = ctx->framebuffers[ctx->get_binding(target)];
  Renderbuffer& rb = ctx->renderbuffers[renderbuffer];
  if (attachment == GL_COLOR_ATTACHMENT0) fb.color_attachment = rb.texture;
  else if (attachment == GL_DEPTH_ATTACHMENT) fb.depth_attachment = rb.texture;
}
GLenum CheckFramebufferStatus(GLenum target) {
  Framebuffer* fb = get_framebuffer(target);
  if (!fb || !fb->color_attachment) return GL_FRAMEBUFFER_UNSUPPORTED;
  return GL_FRAMEBUFFER_COMPLETE;
}

void InitDefaultFramebuffer(int32_t x, int32_t y, int32_t width, int32_t height, int32_t stride, void* buf) {
  Framebuffer& fb = ctx->framebuffers[0];
  if (!fb.color_attachment) GenTextures(1, &fb.color_attachment);
  Texture& colortex = ctx->textures[fb.color_attachment];
  if (buf && stride == 0) stride = aligned_stride(4 * width);
  set_tex_storage(colortex, GL_RGBA8, width, height, buf, stride);
  colortex.offx = x; colortex.offy = y;
  if (!fb.depth_attachment) GenTextures(1, &fb.depth_attachment);
  Texture& depthtex = ctx->textures[fb.depth_attachment];
  set_tex_storage(depthtex, GL_DEPTH_COMPONENT24, width, height);
  depthtex.offx = x; depthtex.offy = y;
}
void* GetColorBuffer(GLuint fbo, GLboolean flush, int32_t* width, int32_t* height, int32_t* stride) {
  Framebuffer* fb = ctx->framebuffers.find(fbo);
  if (!fb || !fb->color_attachment) return nullptr;
  Texture& t = ctx->textures[fb->color_attachment];
  if (flush) flush_all();
  if (width) *width = t.width;
  if (height) *height = t.height;
  if (!t.dptr) return nullptr;
  download_texture(t);
  if (stride) *stride = t.ext_buf ? t.ext_stride : t.stride;
  return t.ext_buf ? t.ext_buf : t.hmirror;
}
void ResolveFramebuffer(GLuint fbo) {
  Framebuffer* fb = ctx->framebuffers.find(fbo);
  if (!fb || !fb->color_attachment) return;
  Texture& t = ctx->textures[fb->color_attachment];
  flush_all();
  if (t.ext_buf) download_texture(t);
}
void SetTextureBuffer(GLuint texid, GLenum internal_format, GLsizei width, GLsizei height, GLsizei stride, void* buf,
                      GLsizei, GLsizei) {
  set_tex_storage(ctx->textures[texid], internal_format, width, height, buf, stride);
}

void ClearTexSubImage(GLenum texture, GLint level, GLint xoffset, GLint yoffset, GLint zoffset, GLsizei width,
                      GLsizei height, GLsizei depth, GLenum format, GLenum type, const void* data) {
  if (level != 0) return;
  Texture& t = ctx->textures[texture];
  if (width <= 0 || height <= 0 || depth <= 0) return;
  (void)zoffset;
  int rect[4] = {xoffset - t.offx, yoffset - t.offy, xoffset + width - t.offx, yoffset + height - t.offy};
  if (t.internal_format == GL_DEPTH_COMPONENT24) {
    uint32_t value = 0xFFFFFF;
    if (format == GL_DEPTH_COMPONENT) {
      if (type == GL_DOUBLE) value = uint32_t(*(const GLdouble*)data * 0xFFFFFF);
      else if (type == GL_FLOAT) value = uint32_t(*(const GLfloat*)data * 0xFFFFFF);
    }
    // Depth is cleared through the colour target it is attached to (see Clear);
    // a direct clear just records the uniform value (gl.cc:2405-2415).
    bool full = rect[0] <= 0 && rect[1] <= 0 && rect[2] >= t.width && rect[3] >= t.height;
    if (!t.depth_cleared || full) {
      if (Texture* ot = t.depth_owner ? ctx->textures.find(t.depth_owner) : nullptr)
        if (ot->pending_write && ot->pending_target >= 0 && ctx->work[ot->pending_target].depth_tex == texture)
          ctx->work[ot->pending_target].depth_live = false;
      t.depth_cleared = true; t.depth_value = value; t.depth_materialized = false;
    }
    return;
  }
  uint32_t color = 0xFF000000;
  if (type == GL_FLOAT) {
    const GLfloat* f = (const GLfloat*)data;
    float v[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    switch (format) {
      case GL_RGBA: v[3] = f[3];
      case GL_RGB: v[2] = f[2];
      case GL_RG: v[1] = f[1];
      case GL_RED: v[0] = f[0]; break;
      default: break;
    }
    uint32_t c[4];
    for (int i = 0; i < 4; i++) {  // round_pixel + CONVERT(.., U8): truncating byte conversion (gl.cc:2440)
      c[i] = uint32_t(int(v[i] * 255.0f + 0.5f)) & 0xFF;
    }
    color = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
  } else if (type == GL_UNSIGNED_BYTE) {
    const GLubyte* b = (const GLubyte*)data;
    switch (format) {
      case GL_RGBA: color = (color & ~0xFF000000) | (uint32_t(b[3]) << 24);
      case GL_RGB: color = (color & ~0x00FF0000) | (uint32_t(b[2]) << 16);
      case GL_RG: color = (color & ~0x0000FF00) | (uint32_t(b[1]) << 8);
      case GL_RED: color = (color & ~0x000000FF) | uint32_t(b[0]); break;
      default: break;
    }
  }
  switch (t.internal_format) {
    case GL_RGBA8:
      record_clear(texture, true, (color & 0xFF00FF00) | ((color << 16) & 0xFF0000) | ((color >> 16) & 0xFF), false, 0, 0, rect);
      break;
    case GL_R8: record_clear(texture, true, color & 0xFF, false, 0, 0, rect); break;
    default: break;  // RG8 targets are not drawn to by WebRender
  }
}
void ClearTexImage(GLenum texture, GLint level, GLenum format, GLenum type, const void* data) {
  Texture& t = ctx->textures[texture];
  ClearTexSubImage(texture, level, t.offx, t.offy, 0, t.width, t.height, 1, format, type, data);
}
void Clear(GLbitfield mask) {
  Framebuffer& fb = *get_framebuffer(GL_DRAW_FRAMEBUFFER, true);
  bool do_depth = (mask & GL_DEPTH_BUFFER_BIT) && fb.depth_attachment;
  uint32_t dvalue = 0;
  if (do_depth) {
    Texture& dt = ctx->textures[fb.depth_attachment];
    dvalue = uint32_t(ctx->cleardepth * 0xFFFFFF);
    // gl.cc:2405-2415: a scissored clear of an uninitialised depth buffer fills all of it
    int r[4] = {0, 0, dt.width, dt.height};
    if (ctx->scissortest) {
      r[0] = std::max(0, ctx->scissor[0] - dt.offx); r[1] = std::max(0, ctx->scissor[1] - dt.offy);
      r[2] = std::min(dt.width, ctx->scissor[2] - dt.offx); r[3] = std::min(dt.height, ctx->scissor[3] - dt.offy);
    }
    bool full = r[0] <= 0 && r[1] <= 0 && r[2] >= dt.width && r[3] >= dt.height;
    if (fb.color_attachment && ctx->textures[fb.color_attachment].has_storage()) {
      int rect[4] = {r[0], r[1], r[2], r[3]};
      if (!dt.depth_cleared) { rect[0] = 0; rect[1] = 0; rect[2] = dt.width; rect[3] = dt.height; full = true; }
      record_clear(fb.color_attachment, false, 0, true, fb.depth_attachment, dvalue, rect);
    }
    if (full) { dt.depth_value = dvalue; dt.depth_materialized = false; }
    dt.depth_cleared = true;
  }
  if ((mask & GL_COLOR_BUFFER_BIT) && fb.color_attachment) {
    Texture& t = ctx->textures[fb.color_attachment];
    int x0 = t.offx, y0 = t.offy, x1 = t.offx + t.width, y1 = t.offy + t.height;
    if (ctx->scissortest) {
      x0 = std::max(x0, ctx->scissor[0]); y0 = std::max(y0, ctx->scissor[1]);
      x1 = std::min(x1, ctx->scissor[2]); y1 = std::min(y1, ctx->scissor[3]);
    }
    ClearTexSubImage(fb.color_attachment, 0, x0, y0, 0, x1 - x0, y1 - y0, 1, GL_RGBA, GL_FLOAT, ctx->clearcolor);
  }
}
void ClearColorRect(GLuint fbo, GLint xoffset, GLint yoffset, GLsizei width, GLsizei height, GLfloat r, GLfloat g,
                    GLfloat b, GLfloat a) {
  GLfloat color[] = {r, g, b, a};
  Framebuffer& fb = ctx->framebuffers[fbo];
  Texture& t = ctx->textures[fb.color_attachment];
  int x0 = std::max(xoffset, t.offx), y0 = std::max(yoffset, t.offy);
  int x1 = std::min(xoffset + width, t.offx + t.width), y1 = std::min(yoffset + height, t.offy + t.height);
  ClearTexSubImage(fb.color_attachment, 0, x0, y0, 0, x1 - x0, y1 - y0, 1, GL_RGBA, GL_FLOAT, color);
}
void InvalidateFramebuffer(GLenum target, GLsizei num_attachments, const GLenum* attachments) {
  Framebuffer* fb = get_framebuffer(target);
  if (!fb || num_attachments <= 0 || !attachments) return;
  for (GLsizei i = 0; i < num_attachments; i++) {
    if (attachments[i] == GL_DEPTH_ATTACHMENT && fb->depth_attachment) {
      Texture& t = ctx->textures[fb->depth_attachment];
      t.depth_cleared = false; t.depth_materialized = false;
      // the pending draws of the target it was attached to need not leave their depth behind
      if (Texture* ot = t.depth_owner ? ctx->textures.find(t.depth_owner) : nullptr)
        if (ot->pending_write && ot->pending_target >= 0 && ctx->work[ot->pending_target].depth_tex == fb->depth_attachment)
          ctx->work[ot->pending_target].depth_live = false;
    }
  }
}

void ReadPixels(GLint x, GLint y, GLsizei width, GLsizei height, GLenum format, GLenum type, void* data_) {
  uint8_t* data = (uint8_t*)pixel_pack_data(data_);
  if (!data) return;
  Framebuffer* fb = get_framebuffer(GL_READ_FRAMEBUFFER);
  if (!fb) return;
  Texture& t = ctx->textures[fb->color_attachment];
  if (!t.dptr) return;
  sync_texture_for_read(t);
  x -= t.offx; y -= t.offy;
  if (internal_format_for_data(format, type) != t.internal_format) return;
  uint8_t* dest = data;
  size_t destStride = (size_t)width * t.bpp;
  if (y < 0) { dest += -y * destStride; height += y; y = 0; }
  if (y + height > t.height) height = t.height - y;
  if (x < 0) { dest += -x * t.bpp; width += x; x = 0; }
  if (x + width > t.width) width = t.width - x;
  if (width <= 0 || height <= 0) return;
  size_t row = (size_t)width * t.bpp;
  wrrt::copy2d(dest, destStride, (const uint8_t*)t.dptr + (size_t)y * t.stride + (size_t)x * t.bpp, t.stride, row, height, 1,
               ctx->stream);
  sync_stream();
  ctx->stats.d2h_bytes += row * height;
  if (format_requires_conversion(format, t.internal_format)) {
    for (int yy = 0; yy < height; yy++) {
      uint32_t* p = (uint32_t*)(dest + (size_t)yy * destStride);
      for (int xx = 0; xx < width; xx++) {
        uint32_t v; memcpy(&v, &p[xx], 4);
        uint32_t rb = v & 0x00FF00FF;
        v = (v & 0xFF00FF00) | (rb << 16) | (rb >> 16);
        memcpy(&p[xx], &v, 4);
      }
    }
  }
}

void CopyImageSubData(GLuint srcName, GLenum srcTarget, GLint, GLint srcX, GLint srcY, GLint, GLuint dstName,
                      GLenum dstTarget, GLint, GLint dstX, GLint dstY, GLint, GLsizei srcWidth, GLsizei srcHeight,
                      GLsizei) {
  if (srcTarget == GL_RENDERBUFFER) srcName = ctx->renderbuffers[srcName].texture;
  if (dstTarget == GL_RENDERBUFFER) dstName = ctx->renderbuffers[dstName].texture;
  Texture& s = ctx->textures[srcName];
  Texture& d = ctx->textures[dstName];
  if (!s.dptr || !d.dptr || s.internal_format != d.internal_format) return;
  if (srcX < 0 || srcY < 0 || dstX < 0 || dstY < 0 || srcX + srcWidth > s.width || srcY + srcHeight > s.height ||
      dstX + srcWidth > d.width || dstY + srcHeight > d.height) return;
  sync_texture_for_read(s);
  sync_texture_for_write(d);
  flush_uploads(2830);
  wrrt::copy2d((uint8_t*)d.dptr + (size_t)dstY * d.stride + (size_t)dstX * d.bpp, d.stride,
               (const uint8_t*)s.dptr + (size_t)srcY * s.stride + (size_t)srcX * s.bpp, s.stride,
               (size_t)srcWidth * s.bpp, srcHeight, 2, ctx->stream);
}
void CopyTexSubImage2D(GLenum target, GLint, GLint xoffset, GLint yoffset, GLint x, GLint y, GLsizei width, GLsizei height) {
  Framebuffer* fb = get_framebuffer(GL_READ_FRAMEBUFFER);
  if (!fb) return;
  CopyImageSubData(fb->color_attachment, GL_TEXTURE_2D, 0, x, y, 0, ctx->get_binding(target), GL_TEXTURE_2D, 0, xoffset,
                   yoffset, 0, width, height, 1);
}
// scale_blit / linear_blit (composite.h:166-432) between two textures.  sr / dr: source / dest request in texture pixels;
// clip: valid dest rect RELATIVE to the dest request (nullptr: the request itself); quirk: BlitFramebuffer hands linear_blit the
// absolute dest request as its clip rect (composite.h:478) -- reproduced.
static void blit_textures(GLuint src_id, Texture& s, GLuint dst_id, Texture& d, const int sr[4], const int dr[4], bool invertX, bool invertY,
                          bool linear, bool composite, const int* clip, bool blit_quirk) {
  const int srcW = sr[2] - sr[0], srcH = sr[3] - sr[1], dstW = dr[2] - dr[0], dstH = dr[3] - dr[1];
  if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return;
  // dest bounds: dsttex.sample_bounds(dstReq) (relative to the request) ∩ clipRect
  int b[4] = {std::max(0, dr[0]) - dr[0], std::max(0, dr[1]) - dr[1], std::min(d.width, dr[2]) - dr[0], std::min(d.height, dr[3]) - dr[1]};
  b[0] = std::max(b[0], 0); b[1] = std::max(b[1], 0); b[2] = std::min(b[2], dstW); b[3] = std::min(b[3], dstH);
  if (clip) { b[0] = std::max(b[0], clip[0]); b[1] = std::max(b[1], clip[1]); b[2] = std::min(b[2], clip[2]); b[3] = std::min(b[3], clip[3]); }
  if (linear && blit_quirk) {
    b[0] = std::max(b[0], dr[0]); b[1] = std::max(b[1], dr[1]); b[2] = std::min(b[2], dr[2]); b[3] = std::min(b[3], dr[3]);
  }
  if (!linear) {
    // source texture bounds relative to the request, flipped if need be, scaled to dest space rounding inward
    int c[4] = {0 - sr[0], 0 - sr[1], s.width - sr[0], s.height - sr[1]};
    if (invertY) { const int y0 = srcH - c[1], y1 = srcH - c[3]; c[1] = y1; c[3] = y0; }
    c[0] = (c[0] * dstW + (srcW - 1)) / srcW; c[1] = (c[1] * dstH + (srcH - 1)) / srcH;
    c[2] = (c[2] * dstW) / srcW; c[3] = (c[3] * dstH) / srcH;
    b[0] = std::max(b[0], c[0]); b[1] = std::max(b[1], c[1]); b[2] = std::min(b[2], c[2]); b[3] = std::min(b[3], c[3]);
  }
  if (b[2] <= b[0] || b[3] <= b[1]) return;
  if (!composite && !linear && !invertY && srcW == dstW && srcH == dstH && s.internal_format == d.internal_format) {
    CopyImageSubData(src_id, GL_TEXTURE_2D, 0, sr[0] + b[0], sr[1] + b[1], 0, dst_id, GL_TEXTURE_2D,
                     0, dr[0] + b[0], dr[1] + b[1], 0, b[2] - b[0], b[3] - b[1], 1);
    return;
  }
  sync_texture_for_read(s);
  sync_texture_for_write(d);
  flush_uploads(2871);
  WrBlitArgs a;
  a.src = s.dptr; a.dst = d.dptr; a.src_stride = s.stride; a.dst_stride = d.stride; a.sbpp = s.bpp; a.dbpp = d.bpp;
  a.sw = s.width; a.sh = s.height;
  a.srx0 = sr[0]; a.sry0 = sr[1]; a.srw = srcW; a.srh = srcH; a.drx0 = dr[0]; a.dry0 = dr[1]; a.drw = dstW; a.drh = dstH;
  a.bx0 = b[0]; a.by0 = b[1]; a.bx1 = b[2]; a.by1 = b[3]; a.invert_y = invertY ? 1 : 0; a.linear = linear ? 1 : 0;
  a.invert_x = invertX ? 1 : 0; a.composite = composite ? 1 : 0;
  const long long n = (long long)(b[2] - b[0]) * (b[3] - b[1]);
  WR_LAUNCH(wr_blit_kernel, (int)((n + 255) / 256), 256, ctx->stream, a);
  ctx->stats.kernel_launches++;
}

void BlitFramebuffer(GLint srcX0, GLint srcY0, GLint srcX1, GLint srcY1, GLint dstX0, GLint dstY0, GLint dstX1, GLint dstY1,
                     GLbitfield mask, GLenum filter) {
  // composite.h:432-483: no scissor, Y flips forced onto the dest side, nearest stepping (scale_blit) unless a scaled
  // GL_LINEAR blit between equal renderable formats (linear_blit)
  if (!(mask & GL_COLOR_BUFFER_BIT)) return;
  Framebuffer* srcfb = get_framebuffer(GL_READ_FRAMEBUFFER);
  Framebuffer* dstfb = get_framebuffer(GL_DRAW_FRAMEBUFFER);
  if (!srcfb || !dstfb) return;
  Texture& s = ctx->textures[srcfb->color_attachment];
  Texture& d = ctx->textures[dstfb->color_attachment];
  if (!s.dptr || !d.dptr) return;
  auto renderable = [](GLenum f) { return f == GL_R8 || f == GL_RG8 || f == GL_RGBA8; };
  if (s.internal_format != d.internal_format && (!renderable(s.internal_format) || !renderable(d.internal_format))) return;
  if (srcY1 < srcY0) { std::swap(srcY0, srcY1); std::swap(dstY0, dstY1); }
  const bool invertY = dstY1 < dstY0;
  if (invertY) std::swap(dstY0, dstY1);
  const int sr[4] = {srcX0 - s.offx, srcY0 - s.offy, srcX1 - s.offx, srcY1 - s.offy};
  const int dr[4] = {dstX0 - d.offx, dstY0 - d.offy, dstX1 - d.offx, dstY1 - d.offy};
  const int srcW = sr[2] - sr[0], srcH = sr[3] - sr[1], dstW = dr[2] - dr[0], dstH = dr[3] - dr[1];
  if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return;
  const bool linear = !(srcW == dstW && srcH == dstH) && s.width >= 2 && filter == GL_LINEAR && s.internal_format == d.internal_format &&
                      (s.internal_format == GL_RGBA8 || s.internal_format == GL_R8);
  blit_textures(srcfb->color_attachment, s, dstfb->color_attachment, d, sr, dr, false, invertY, linear, false, nullptr, true);
}

// Does any instance of a batch ask for swgl_antiAlias?  The request travels in the third word of every instance (brush: flags =
// z >> 16, BRUSH_FLAG_FORCE_AA = 1024, gpu_types.rs:690-703; quad: part = (z >> 8) & 0xff in PART_LEFT..PART_BOTTOM, or PART_ALL
// with edge flags (z >> 16) & 0xff, gpu_types.rs:564-589).  A frame of cfg5 asks this of 65 k instances: eight at a time with
// AVX2 gathers where the host has them (a quarter of the recording time as a scalar loop with an early exit).
static inline uint32_t aa_request_bits(int32_t zw, bool quad) {
  if (!quad) return (uint32_t)(zw >> 16) & 1024u;
  const uint32_t part = ((uint32_t)zw >> 8) & 0xffu, edges = ((uint32_t)zw >> 16) & 0xffu;
  return (uint32_t)((part - 1u) < 4u) | (uint32_t)((part == 5u) & (edges != 0u));
}
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static bool any_aa_request_avx2(const uint8_t* ib, int n, int stride, bool quad) {
  const __m256i idx = _mm256_mullo_epi32(_mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7), _mm256_set1_epi32(stride));
  __m256i acc = _mm256_setzero_si256();
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256i z = _mm256_i32gather_epi32((const int*)(ib + (size_t)i * stride + 8), idx, 1);
    if (quad) {
      const __m256i part = _mm256_and_si256(_mm256_srli_epi32(z, 8), _mm256_set1_epi32(0xff));
      const __m256i edges = _mm256_and_si256(z, _mm256_set1_epi32(0xff0000));
      // part in 1..4: (part - 1) < 4 as signed compare on small non-negative values; part - 1 == -1 for part 0
      const __m256i pm1 = _mm256_sub_epi32(part, _mm256_set1_epi32(1));
      const __m256i in14 = _mm256_andnot_si256(_mm256_cmpgt_epi32(_mm256_setzero_si256(), pm1), _mm256_cmpgt_epi32(_mm256_set1_epi32(4), pm1));
      const __m256i all_e = _mm256_andnot_si256(_mm256_cmpeq_epi32(edges, _mm256_setzero_si256()), _mm256_cmpeq_epi32(part, _mm256_set1_epi32(5)));
      acc = _mm256_or_si256(acc, _mm256_or_si256(in14, all_e));
    } else {
      acc = _mm256_or_si256(acc, _mm256_and_si256(z, _mm256_set1_epi32(1024 << 16)));
    }
    if ((i & 1016) == 1016 && !_mm256_testz_si256(acc, acc)) return true;      // (an early exit every 1024 instances)
  }
  if (!_mm256_testz_si256(acc, acc)) return true;
  for (; i < n; i++) { int32_t zw; memcpy(&zw, ib + (size_t)i * stride + 8, 4); if (aa_request_bits(zw, quad)) return true; }
  return false;
}
#endif
static bool any_aa_request(const uint8_t* ib, int n, int stride, bool quad) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && n >= 16 && stride > 0 && (size_t)stride * 8 < (1u << 30)) return any_aa_request_avx2(ib, n, stride, quad);
#endif
  uint32_t bad = 0;
  for (int i = 0; i < n; i++) { int32_t zw; memcpy(&zw, ib + (size_t)i * stride + 8, 4); bad |= aa_request_bits(zw, quad); if ((i & 255) == 255 && bad) return true; }
  return bad != 0;
}

// ---- the hot path: record one instanced batch ------------------------------
void DrawElementsInstanced(GLenum mode, GLsizei count, GLenum type, GLintptr offset, GLsizei instancecount) {
  Context* c = ctx;
  HostTimer ht(c ? &c->stats.host_record_ns : nullptr);
  Program* prog = c->programs.find(c->current_program);
  if (offset < 0 || count <= 0 || instancecount <= 0 || !prog) return;
  // What cannot be drawn is refused HERE, when the draw is recorded (GL_INVALID_OPERATION is visible to the caller's very
  // next GetError()), not discovered by the device after the frame was presented without it.
  auto refuse = [&](const char* why) {
    c->last_error = GL_INVALID_OPERATION;
    if (!prog->refused_once) fprintf(stderr, "libwrhip: draw refused (program '%s'): %s\n", prog->name, why);
    prog->refused_once = true;
  };
  if (!prog->info) { refuse("this shader key has no implementation"); return; }
  Framebuffer& fb = *get_framebuffer(GL_DRAW_FRAMEBUFFER, true);
  if (!fb.color_attachment) return;
  GLuint color_id = fb.color_attachment;
  {
    Texture& colortex = c->textures[color_id];
    if (!colortex.dptr || colortex.own_none()) return;      // (own_none: another rank's target, WrhipSetTargetRows)
    if (colortex.internal_format != GL_RGBA8 && colortex.internal_format != GL_R8) { refuse("colour target is neither RGBA8 nor R8"); return; }
  }
  VertexArray& v = c->vertex_arrays[c->current_vertex_array];
  // Only WebRender's instanced unit quad is supported: TRIANGLES, 6 x u16,
  // indices i, i+1, i+2, i+2, i+1, i+3 (rasterize.h:1646-1659; vertex.rs:1079-1080).
  if (mode != GL_TRIANGLES || type != GL_UNSIGNED_SHORT || count != 6) {
    refuse("only WebRender's instanced unit quad (TRIANGLES, 6 x UNSIGNED_SHORT) is drawn");
    return;
  }
  Buffer& ib = c->buffers[v.element_array_buffer_binding];
  if (!ib.buf || (size_t)offset + 12 > ib.size) return;
  const uint16_t* idx = (const uint16_t*)(ib.buf + offset);
  if (!(idx[1] == idx[0] + 1 && idx[2] == idx[0] + 2 && idx[5] == idx[0] + 3)) {
    refuse("index pattern is not the unit quad's");
    return;
  }
  const ShaderInfo* info = prog->info;
  WrDrawDesc d;
  memset(&d, 0, sizeof(d));
  d.shader = info->kind;
  d.count = instancecount;
  // per-vertex aPosition of the 4 quad lanes 0,1,3,2 (load_attrib, gl.cc:1031-1039)
  {
    int loc = prog->attrib_loc[0];
    VertexAttrib& va = v.attribs[loc];
    Buffer& vb = c->buffers[va.vertex_buffer];
    static const int lane_vertex[4] = {0, 1, 3, 2};
    for (int n = 0; n < 4; n++) {
      float xy[2] = {0.f, 0.f};
      const size_t voff = (size_t)va.stride * (idx[0] + lane_vertex[n]) + va.offset;
      if (loc != NULL_ATTRIB && va.enabled && vb.buf && voff + va.size <= vb.size) {
        const uint8_t* src = vb.buf + voff;
        int comps = 2;
        for (int k = 0; k < comps; k++) {
          if (va.type == GL_UNSIGNED_BYTE) {
            if ((size_t)k < va.size) xy[k] = va.normalized ? float(src[k]) * (1.0f / 255.0f) : float(src[k]);
          } else if (va.type == GL_UNSIGNED_SHORT) {
            if ((size_t)k * 2 < va.size) { uint16_t u; memcpy(&u, src + 2 * k, 2); xy[k] = va.normalized ? float(u) * (1.0f / 65535.0f) : float(u); }
          } else if (va.type == GL_FLOAT) {
            if ((size_t)k * 4 < va.size) memcpy(&xy[k], src + 4 * k, 4);
          }
        }
      }
      d.quad[2 * n] = xy[0]; d.quad[2 * n + 1] = xy[1];
    }
  }
  // instance attributes: all must come from one interleaved instance buffer
  GLuint inst_buf = 0; int inst_stride = 0;
  for (int k = 0; k < WR_MAX_ATTRIBS; k++) { d.attr_off[k] = -1; d.attr_bytes[k] = 0; }
  d.attr_u16 = 0;
  for (int k = 1; k < WR_MAX_ATTRIBS + 1 && info->attribs[k]; k++) {
    int loc = prog->attrib_loc[k];
    if (loc == NULL_ATTRIB) continue;
    VertexAttrib& va = v.attribs[loc];
    if (!va.enabled || va.divisor != 1) continue;
    if (va.type != GL_INT && va.type != GL_FLOAT && va.type != GL_UNSIGNED_SHORT) { refuse("instance attribute type is neither INT, FLOAT nor UNSIGNED_SHORT"); return; }
    if (va.type == GL_UNSIGNED_SHORT) d.attr_u16 |= 1u << (k - 1);
    if (!inst_buf) { inst_buf = va.vertex_buffer; inst_stride = va.stride; }
    if (va.vertex_buffer != inst_buf || va.stride != inst_stride) { refuse("instance attributes come from more than one buffer / stride"); return; }
    d.attr_off[k - 1] = va.offset; d.attr_bytes[k - 1] = (int)va.size;
  }
  Buffer* instb = inst_buf ? c->buffers.find(inst_buf) : nullptr;
  size_t need = (size_t)inst_stride * instancecount;
  if (!instb || !instb->buf || need > instb->size) {
    if (inst_buf) { refuse("instance buffer holds fewer bytes than instancecount x stride"); return; }
  }
  if (c->depthtest && fb.depth_attachment) {
    Texture* dtx = c->textures.find(fb.depth_attachment);
    if (dtx && dtx->internal_format == GL_DEPTH_COMPONENT24 && dtx->depth_cleared) claim_depth(color_id, fb.depth_attachment, false);
  }
  int wi = find_or_add_work(color_id);   // may flush (target sampled by pending draws)
  // A pending target that gets sampled is not flushed on the spot: the sampling target moves one
  // dependency level above it and the whole chain (masks -> blurs -> tiles -> composite) goes
  // through ONE flush -- one arena copy, one upload scatter, one setup launch, one raster launch
  // per level -- instead of a full launch sequence per pass.
  // (sampler2DRect samplers read the unit's GL_TEXTURE_RECTANGLE binding, gl.cc:853-855)
  auto bound_tex = [&](int s) {
    const Context::TextureUnit& u = c->texture_units[prog->sampler_unit[s] & 15];
    return (info->rect && (s == WR_S_COLOR0 || s == WR_S_COLOR1 || s == WR_S_COLOR2)) ? u.texture_rectangle_binding : u.texture_2d_binding;
  };
  for (int s = 0; s < WR_MAX_TEX; s++) {
    if (!((info->samplers >> s) & 1)) continue;
    GLuint tid = bound_tex(s);
    Texture* t = tid ? c->textures.find(tid) : nullptr;
    if (!t || !t->dptr || tid == color_id) continue;
    if (t->pending_write && t->pending_target >= 0)
      c->work[wi].level = std::max(c->work[wi].level, c->work[t->pending_target].level + 1);
  }
  Texture& colortex = c->textures[color_id];
  const bool grad_prog = info->kind == WR_SH_BRUSH_LINEAR_GRADIENT || info->kind == WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA || info->kind == WR_SH_PS_QUAD_RADIAL_GRADIENT ||
                         info->kind == WR_SH_PS_QUAD_CONIC_GRADIENT || info->kind == WR_SH_CS_LINEAR_GRADIENT || info->kind == WR_SH_CS_RADIAL_GRADIENT ||
                         info->kind == WR_SH_CS_CONIC_GRADIENT;
  // a table copy per instance (WR_GTAB_WORDS words): promised here, where the raster stage's reads are booked, for draws of up to 256
  // gradients while the promises of the pending flush stay under 256 MB
  const bool gtab = grad_prog && c->grad_tables && instancecount <= 256 && c->gtab_pending + (size_t)instancecount * WR_GTAB_WORDS <= ((size_t)64 << 20);
  if (gtab) c->gtab_pending += (size_t)instancecount * WR_GTAB_WORDS;
  for (int s = 0; s < WR_MAX_TEX; s++) {
    if (!((info->samplers >> s) & 1)) continue;
    GLuint tid = bound_tex(s);
    Texture* t = tid ? c->textures.find(tid) : nullptr;
    if (!t || !t->dptr || tid == color_id) continue;
    WrTexDesc& td = d.tex[s];
    td.ptr = t->dptr; td.width = t->width; td.height = t->height;
    const bool rect_s = info->rect && (s == WR_S_COLOR0 || s == WR_S_COLOR1 || s == WR_S_COLOR2);
    td.sw = rect_s ? 1.0f : float(t->width); td.sh = rect_s ? 1.0f : float(t->height);
    td.stride = t->bpp >= 4 ? t->stride / 4 : (t->bpp == 2 ? t->stride / 2 : t->stride);
    td.format = (int16_t)wr_format(t->internal_format);
    td.linear = (t->mag_filter == GL_LINEAR || t->mag_filter == GL_LINEAR_MIPMAP_LINEAR || t->mag_filter == GL_LINEAR_MIPMAP_NEAREST) && t->width >= 2;
    if (info->kind == WR_SH_PS_COPY) td.linear = 0;      // texelFetch: the sampler's filter does not enter (ps_copy.glsl:36-38)
    mark_ref(tid, *t, false);
    std::vector<GLuint>& reads = c->work[wi].reads;
    if (std::find(reads.begin(), reads.end(), tid) == reads.end()) reads.push_back(tid);
    unsigned rmask = (1u << WR_S_COLOR0) | (1u << WR_S_COLOR1) | (1u << WR_S_COLOR2) | (1u << WR_S_CLIP_MASK);
    if (info->kind == WR_SH_BRUSH_BLEND || info->kind == WR_SH_BRUSH_BLEND_ALPHA || info->kind == WR_SH_CS_SVG_FILTER || info->kind == WR_SH_CS_SVG_FILTER_NODE) rmask |= 1u << WR_S_GPU_CACHE;      // (component-transfer tables, read by main())
    // (gradient tables: the span shaders and main() of every gradient program read the stops where the frame builder put them --
    // unless the setup stage copies them into the flush's pool for this draw, WR_DF_GTAB)
    if (grad_prog && !gtab) rmask |= 1u << WR_S_GPU_BUFFER_F;
    // Fused scatter: what the SETUP stage reads (the data textures: everything outside rmask) must not depend on the batch's scatter,
    // which runs beside it in the same launch.  A texture the open batch uploads whole is read in the staging mirror (same layout,
    // alive as long as the flush's arena); one it writes in part keeps the scatter in front of the setup stage.
    if (!((rmask >> s) & 1) && c->upload_open && t->up_batch == c->upload_batch) {
      if (t->view_batch == c->upload_batch && c->fuse_scatter) td.ptr = c->dupload + t->view_off;
      else {
        static const bool dbg = getenv("WRHIP_DEBUG_SCATTER") != nullptr;
        if (dbg && !c->scatter_first) fprintf(stderr, "scatter first: slot %d texture %u %dx%d fmt %x (the open batch writes it in part)\n", s, tid, t->width, t->height, t->internal_format);
        c->scatter_first = true;
      }
    }
    if ((rmask >> s) & 1) {
      std::vector<GLuint>& rr = c->work[wi].rreads;
      if (std::find(rr.begin(), rr.end(), tid) == rr.end()) rr.push_back(tid);
    }
  }
  d.target = wi;
  d.blend = c->blend ? c->blend_key : WR_BLEND_NONE;
  // (GL_ONE, GL_ONE_MINUS_SRC1_COLOR under the dual-source text program is fine: every prim of that
  // program replaces the key with swgl_blendSubpixelText / swgl_blendDropShadow in its vertex stage)
  const bool dual_text = d.blend == WR_BLEND_DUAL_SRC && (info->kind == WR_SH_PS_TEXT_RUN_DUAL || info->kind == WR_SH_PS_TEXT_RUN_DUAL_GT || info->kind == WR_SH_BRUSH_IMAGE_DUAL || info->kind == WR_SH_BRUSH_IMAGE_REPEAT_DUAL);      // (the image program writes the second colour itself)
  if (!dual_text && (d.blend == WR_BLEND_UNSUPPORTED || d.blend == WR_BLEND_DUAL_SRC)) {
    // not in swgl's key table either (gl.cc:614-645: the reference asserts) -- or GL_ONE, GL_ONE_MINUS_SRC1_COLOR outside the
    // dual-source text program, which needs gl_SecondaryFragColor from a shader that is not implemented
    fprintf(stderr, "libwrhip: blend state without an implementation (funcs %x %x %x %x equation %x)\n", c->blendfunc_srgb, c->blendfunc_drgb,
            c->blendfunc_sa, c->blendfunc_da, c->blend_equation);
    c->last_error = GL_INVALID_OPERATION;
  }
  d.flags = (info->rect ? WR_DF_TEX_RECT : 0) | (gtab ? WR_DF_GTAB : 0);
  Texture* depthtex = (c->depthtest && fb.depth_attachment) ? c->textures.find(fb.depth_attachment) : nullptr;
  if (depthtex && depthtex->internal_format == GL_DEPTH_COMPONENT24 && depthtex->depth_cleared) {
    d.flags |= WR_DF_DEPTH_TEST;
    if (c->depthmask) d.flags |= WR_DF_DEPTH_WRITE;
    if (c->depthfunc == GL_LESS) d.flags |= WR_DF_DEPTH_LESS;
    TargetWork& dw = c->work[wi];
    dw.depth_tex = fb.depth_attachment;
    if (!dw.init_depth_set) { dw.init_depth = depthtex->depth_value; dw.init_depth_set = true; }
    dw.depth_live = true;
  }
  d.query_slot = c->samples_passed_query ? c->queries[c->samples_passed_query].slot : -1;
  {
    // draws that can only produce solid prims with a blend the inline raster paths know (WrFeat)
    const bool plain_blend = d.blend == WR_BLEND_NONE || d.blend == WR_BLEND_PREMULT;
    const bool maskable = d.blend != WR_BLEND_NONE && d.tex[WR_S_CLIP_MASK].ptr && d.tex[WR_S_CLIP_MASK].width >= 2;
    bool simple = false;
    auto ids_clean = [&](int slot, bool headers) {
      GLuint tid = c->texture_units[prog->sampler_unit[slot] & 15].texture_2d_binding;
      Texture* ht = tid ? c->textures.find(tid) : nullptr;
      return !ht || !(headers ? ht->complex_ids_headers : ht->complex_ids_gpubuf);
    };
    if (info->kind == WR_SH_BRUSH_SOLID || info->kind == WR_SH_BRUSH_SOLID_ALPHA)
      simple = plain_blend && !maskable && ids_clean(WR_S_PRIM_HEADERS_I, true);
    else if (info->kind == WR_SH_PS_QUAD_TEXTURED)
      simple = plain_blend && d.tex[WR_S_COLOR0].width < 2 && ids_clean(WR_S_GPU_BUFFER_I, false);
    if (simple && d.blend != WR_BLEND_NONE && d.attr_off[0] >= 0 && d.attr_bytes[0] >= 12 && inst_stride >= 12) {
      // swgl_antiAlias only matters with blending on; the request travels in aData.z of every
      // instance (brush: flags = z >> 16, BRUSH_FLAG_FORCE_AA = 1024, gpu_types.rs:690-703; quad:
      // part = (z >> 8) & 0xff, edge flags = (z >> 16) & 0xff, gpu_types.rs:564-589)
      const uint8_t* ib = (const uint8_t*)instb->buf + d.attr_off[0];
      if (any_aa_request(ib, instancecount, inst_stride, info->kind == WR_SH_PS_QUAD_TEXTURED)) simple = false;
    }
    if (simple) d.flags |= WR_DF_SIMPLE;
    {
      // can a prim of this draw sit under a rotation or a projective transform?  (brushes / text: transform ids in the prim
      // headers; quads: in the GPU buffer)
      const bool quad_prog = info->kind == WR_SH_PS_QUAD_TEXTURED || info->kind == WR_SH_PS_QUAD_MASK || info->kind == WR_SH_PS_QUAD_MASK_FAST ||
                             info->kind == WR_SH_PS_QUAD_RADIAL_GRADIENT || info->kind == WR_SH_PS_QUAD_CONIC_GRADIENT;
      if (!ids_clean(quad_prog ? WR_S_GPU_BUFFER_I : WR_S_PRIM_HEADERS_I, !quad_prog)) d.flags |= WR_DF_XFORM;
      if (info->kind == WR_SH_PS_SPLIT_COMPOSITE) d.flags |= WR_DF_XFORM;      // (its vertices are any convex quad, whatever the transform)
    }
    // textured prims on rotated quads or with swgl_antiAlias need the WR_PK_TEX_QUAD path (WR_FEAT_SHADE launches): the
    // transform ids in the bound header texture and the AA requests in the instances say whether this draw can hold any
    const bool img = info->kind == WR_SH_BRUSH_IMAGE || info->kind == WR_SH_BRUSH_IMAGE_ALPHA ||
                     info->kind == WR_SH_BRUSH_OPACITY || info->kind == WR_SH_BRUSH_OPACITY_ALPHA ||
                     info->kind == WR_SH_BRUSH_IMAGE_REPEAT || info->kind == WR_SH_BRUSH_IMAGE_REPEAT_ALPHA || info->kind == WR_SH_BRUSH_IMAGE_REPEAT_DUAL ||
                     info->kind == WR_SH_BRUSH_LINEAR_GRADIENT || info->kind == WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA ||
                     info->kind == WR_SH_BRUSH_BLEND || info->kind == WR_SH_BRUSH_BLEND_ALPHA ||
                     info->kind == WR_SH_BRUSH_MIX_BLEND || info->kind == WR_SH_BRUSH_MIX_BLEND_ALPHA;
    // (glyph quads under a rotation -- local raster space -- ride on the same path; the program never asks for swgl_antiAlias,
    // and a glyph instance's third word is not a brush's flags: the transform ids alone decide)
    const bool text = info->kind == WR_SH_PS_TEXT_RUN || info->kind == WR_SH_PS_TEXT_RUN_DUAL || info->kind == WR_SH_PS_TEXT_RUN_GT || info->kind == WR_SH_PS_TEXT_RUN_DUAL_GT;
    const bool texquad = (info->kind == WR_SH_PS_QUAD_TEXTURED && d.tex[WR_S_COLOR0].width >= 2) ||
                         info->kind == WR_SH_PS_QUAD_MASK || info->kind == WR_SH_PS_QUAD_MASK_FAST ||
                         info->kind == WR_SH_PS_QUAD_RADIAL_GRADIENT || info->kind == WR_SH_PS_QUAD_CONIC_GRADIENT;
    const bool solid_masked = (info->kind == WR_SH_BRUSH_SOLID || info->kind == WR_SH_BRUSH_SOLID_ALPHA) && maskable;
    if (colortex.internal_format == GL_RGBA8 && (img || texquad || solid_masked || text)) {
      bool quads = !ids_clean(texquad ? WR_S_GPU_BUFFER_I : WR_S_PRIM_HEADERS_I, !texquad);
      if (!quads && !text && d.blend != WR_BLEND_NONE && d.attr_off[0] >= 0 && d.attr_bytes[0] >= 12 && inst_stride >= 12) {
        const uint8_t* ib = (const uint8_t*)instb->buf + d.attr_off[0];
        quads = any_aa_request(ib, instancecount, inst_stride, texquad);
      }
      if (quads) d.flags |= WR_DF_QUADS;
    }
    // (a split polygon is a general quad by construction: ps_split_composite.glsl:85-87)
    if (info->kind == WR_SH_PS_SPLIT_COMPOSITE && colortex.internal_format == GL_RGBA8) d.flags |= WR_DF_QUADS;
  }
  apply_scissor(colortex, d.clip);
  d.vp_origin[0] = float(c->viewport[0] - colortex.offx); d.vp_origin[1] = float(c->viewport[1] - colortex.offy);
  d.vp_size[0] = float(c->viewport[2] - c->viewport[0]); d.vp_size[1] = float(c->viewport[3] - c->viewport[1]);
  memcpy(d.transform, prog->transform, sizeof(d.transform));
  d.blend_color[0] = c->blendcolor[0]; d.blend_color[1] = c->blendcolor[1];
  // snapshot the instance bytes (the caller may overwrite the VBO right after)
  TargetWork& w = c->work[wi];
  size_t pos = (w.inst.size() + 15) & ~size_t(15);
  if (need >= PARALLEL_COPY_MIN) {
    w.inst.resize(pos + need);
    big_memcpy(w.inst.data() + pos, instb->buf, need);
  } else {
    // (one pass: resize() would zero the bytes the copy is about to overwrite -- 72 snapshots of ~33 KB per cfg5 frame)
    w.inst.resize(pos);
    if (need) w.inst.insert(w.inst.end(), (const uint8_t*)instb->buf, (const uint8_t*)instb->buf + need);
  }
  d.inst_offset = pos; d.inst_stride = inst_stride;
#ifdef WRHIP_HOSTSIM
  if (getenv("WRHIP_DEBUG") && need >= 16) { const float* f = (const float*)(w.inst.data() + pos); fprintf(stderr, "record: sh %d buf %u need %zu first %g %g %g %g\n", d.shader, inst_buf, need, f[0], f[1], f[2], f[3]); }
#endif
  w.draws.push_back(d);
  w.prims += instancecount;
}

void Finish(void) {
  flush_all();
  flush_uploads(3212);
  sync_stream();
  ctx->held_seq = ctx->flush_seq; ctx->fb_writes_since_held = 0;
  {
    // prims the device could not draw faithfully are reported, never dropped silently
    WrUnsupportedCounters h;
    wrrt::d2h(&h, ctx->dcounters, sizeof(h), ctx->stream);
    sync_stream();
    if (h.chain_timeout != ctx->seen.chain_timeout) {
      fprintf(stderr, "libwrhip: %u workgroup(s) of a chained mask launch gave up at a level barrier: the masks of this frame are not trustworthy\n", h.chain_timeout - ctx->seen.chain_timeout);
      ctx->last_error = GL_INVALID_OPERATION;
    }
    if (h.unsupported_prims != ctx->seen.unsupported_prims || h.perspective_prims != ctx->seen.perspective_prims) {
      fprintf(stderr, "libwrhip: %u prim(s) not reproduced exactly (no implementation: not drawn; more depth runs on a row than the tables hold: drawn from the span start), %u perspective ones (clipped by the near / far planes, or a program whose perspective inputs are not restated): not drawn\n",
              h.unsupported_prims - ctx->seen.unsupported_prims, h.perspective_prims - ctx->seen.perspective_prims);
      // visible at the ABI, not only on stderr: the frame has holes, and the caller's GetError() says so (the reference
      // itself only ever raises GL_OUT_OF_MEMORY, gl.cc:1125-1134, so any other code is unambiguous)
      ctx->last_error = GL_INVALID_OPERATION;
    }
    if (ctx->last_qctl && ctx->runs_pool_words) {
      // the flush's pool: what the last flush asked of it against what it held (Context::runs_pool_words)
      unsigned long long asked = 0;
      wrrt::d2h(&asked, ctx->last_qctl, sizeof(asked), ctx->stream);
      sync_stream();
      ctx->last_qctl = nullptr;
      if (asked > (unsigned long long)ctx->last_qtab_cap) {
        const unsigned long long extra = asked - ctx->last_qtab_cap;
        const size_t want = std::min<size_t>(((size_t)1 << 30) - ((size_t)80 << 20), ctx->runs_pool_words + (size_t)(extra + extra / 4) + ((size_t)1 << 20));
        if (want > ctx->runs_pool_words) {
          fprintf(stderr, "libwrhip: the flush's pool ran out (%llu words asked, %zu held): its share for depth runs grows from %zu to %zu words for the following flushes\n",
                  asked, ctx->last_qtab_cap, ctx->runs_pool_words, want);
          ctx->runs_pool_words = want;
        }
      }
    }
    static const bool dbgc = getenv("WRHIP_DEBUG_COUNTERS") != nullptr;
    if (dbgc) fprintf(stderr, "libwrhip dbg counters (delta): %u %u %u %u %u (max %u)\n", h.dbg[0] - ctx->seen.dbg[0], h.dbg[1] - ctx->seen.dbg[1],
                      h.dbg[2] - ctx->seen.dbg[2], h.dbg[3] - ctx->seen.dbg[3], h.dbg[4] - ctx->seen.dbg[4], h.dbg[5]);
    ctx->seen = h;
  }
  // externally backed default framebuffer: make the result visible to the host
  Framebuffer* fb = ctx->framebuffers.find(0);
  if (fb && fb->color_attachment) {
    Texture& t = ctx->textures[fb->color_attachment];
    if (t.ext_buf) download_texture(t);
  }
}

void MakeCurrent(WrhipContext* c) { ctx = (Context*)c; }
#ifdef WRHIP_TIMING
static void wr_dump_prim_times() {
  const char* path = getenv("WRHIP_PRIM_TIMES");
  if (!path) return;
  std::vector<unsigned> h(16384 * 4);
  wrq::drain();
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(wr_dbg_prim), h.size() * 4) != hipSuccess) return;
  if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
  std::vector<unsigned long long> tp(16384 * 8);
  if (hipMemcpyFromSymbol(tp.data(), HIP_SYMBOL(wr_dbg_tp), tp.size() * 8) != hipSuccess) return;
  if (FILE* f = fopen((std::string(path) + ".tp").c_str(), "wb")) { fwrite(tp.data(), 8, tp.size(), f); fclose(f); }
}
#endif
#ifdef WR_ROWS_TIMING
static void wr_dump_rows_times() {
  const char* path = getenv("WRHIP_ROWS_TIMES");
  if (!path) return;
  std::vector<unsigned long long> h(4096 * 8);
  wrq::drain();
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(wr_rows_times), h.size() * 8) != hipSuccess) return;
  if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
}
#endif
#ifdef WR_CELL_TIMING
static void wr_dump_cell_times() {
  const char* path = getenv("WRHIP_CELL_TIMES");
  if (!path) return;
  std::vector<unsigned long long> h(8192 * 16);
  wrq::drain();
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(wr_cell_times), h.size() * 8) != hipSuccess) return;
  if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
}
#endif
WrhipContext* CreateContext(void) {
#ifdef WR_CELL_TIMING
  static bool reg = false;
  if (!reg) { reg = true; atexit(wr_dump_cell_times); }
#endif
#ifdef WR_ROWS_TIMING
  static bool reg3 = false;
  if (!reg3) { reg3 = true; atexit(wr_dump_rows_times); }
#endif
#ifdef WRHIP_TIMING
  static bool reg2 = false;
  if (!reg2) { reg2 = true; atexit(wr_dump_prim_times); }
#endif
  ensure_runtime();
  return (WrhipContext*)new Context();
}
void ReferenceContext(WrhipContext* c) { if (c) ++((Context*)c)->references; }
void DestroyContext(WrhipContext* c_) {
  Context* c = (Context*)c_;
  if (!c) return;
  if (--c->references > 0) return;
#ifdef WR_CELL_TIMING
  wr_dump_cell_times();
#endif
#ifdef WR_ROWS_TIMING
  wr_dump_rows_times();
#endif
#ifdef WRHIP_TIMING
  wr_dump_prim_times();
#endif
#ifdef WRHIP_HOSTSIM
  if (getenv("WRHIP_DEBUG")) fprintf(stderr, "paths: r8fast %llu (unit %llu) generic %llu accum_loop %llu linear: fallback %llu upscale %llu fast %llu downscale %llu\n", wr_dbg_paths[0], wr_dbg_paths[3], wr_dbg_paths[1], wr_dbg_paths[2], wr_dbg_paths[4], wr_dbg_paths[5], wr_dbg_paths[6], wr_dbg_paths[7]);
#endif
  if (c->lap_n) fprintf(stderr, "libwrhip flush laps (us per flush, %llu flushes): plan %.2f arena %.2f uploads %.2f setup+held %.2f own launches %.2f bookkeeping %.2f\n",
                        (unsigned long long)c->lap_n, c->lap_ns[0] / 1e3 / c->lap_n, c->lap_ns[1] / 1e3 / c->lap_n, c->lap_ns[2] / 1e3 / c->lap_n, c->lap_ns[3] / 1e3 / c->lap_n,
                        c->lap_ns[4] / 1e3 / c->lap_n, c->lap_ns[5] / 1e3 / c->lap_n);
  if (ctx == c) { delete c; ctx = nullptr; }
  else delete c;
}
size_t ReportMemory(WrhipContext* c_, size_t (*)(const void*)) {
  Context* c = (Context*)c_;
  size_t size = 0;
  if (c) for (Texture* t : c->textures.objects) if (t && t->dptr && !t->ext_buf) size += t->dsize;
  return size;
}

// ---- locking / compositor extras (composite.h:485-593) ----------------------
// LockedTexture* is the Texture itself; locking pins a host mirror.
LockedTexture* LockFramebuffer(GLuint fbo) {
  Framebuffer* fb = ctx->framebuffers.find(fbo);
  if (!fb || !fb->color_attachment) return nullptr;
  Texture& t = ctx->textures[fb->color_attachment];
  flush_all(); download_texture(t);
  t.locked++;
  return (LockedTexture*)&t;
}
LockedTexture* LockTexture(GLuint tex) {
  Texture* t = ctx->textures.find(tex);
  if (!t || !t->dptr) return nullptr;
  flush_all(); download_texture(*t);
  t->locked++;
  return (LockedTexture*)t;
}
void LockResource(LockedTexture* r) { if (r) ((Texture*)r)->locked++; }
void UnlockResource(LockedTexture* r) { if (r && ((Texture*)r)->locked > 0) ((Texture*)r)->locked--; }
void* GetResourceBuffer(LockedTexture* r, int32_t* width, int32_t* height, int32_t* stride) {
  Texture* t = (Texture*)r;
  if (!t) return nullptr;
  if (width) *width = t->width;
  if (height) *height = t->height;
  if (stride) *stride = t->ext_buf ? t->ext_stride : t->stride;
  return t->ext_buf ? t->ext_buf : t->hmirror;
}
void Composite(LockedTexture* lockedDst, LockedTexture* lockedSrc, GLint srcX, GLint srcY, GLsizei srcWidth, GLsizei srcHeight, GLint dstX,
               GLint dstY, GLsizei dstWidth, GLsizei dstHeight, GLboolean opaque, GLboolean flipX, GLboolean flipY, GLenum filter,
               GLint clipX, GLint clipY, GLsizei clipWidth, GLsizei clipHeight) {
  // composite.h:542-589: a (scaled, flipped, clipped) copy -- or, when !opaque, a premultiplied-over blend -- of a locked RGBA8
  // texture into another one: scale_blit, or linear_blit for X flips and scaled GL_LINEAR copies.  The locked resources' host
  // views are refreshed afterwards (the caller reads the destination through GetResourceBuffer).
  if (!lockedDst || !lockedSrc || !ctx) return;
  Texture& s = *(Texture*)lockedSrc;
  Texture& d = *(Texture*)lockedDst;
  if (s.bpp != 4 || d.bpp != 4 || !s.dptr || !d.dptr) return;
  GLuint sid = 0, did = 0;
  for (size_t i = 1; i < ctx->textures.objects.size(); i++) {
    if (ctx->textures.objects[i] == &s) sid = (GLuint)i;
    if (ctx->textures.objects[i] == &d) did = (GLuint)i;
  }
  const int sr[4] = {srcX - s.offx, srcY - s.offy, srcX + srcWidth - s.offx, srcY + srcHeight - s.offy};
  const int dr[4] = {dstX - d.offx, dstY - d.offy, dstX + dstWidth - d.offx, dstY + dstHeight - d.offy};
  if (sr[2] <= sr[0] || sr[3] <= sr[1] || dr[2] <= dr[0] || dr[3] <= dr[1]) return;
  const int clip[4] = {clipX - dstX, clipY - dstY, clipX - dstX + clipWidth, clipY - dstY + clipHeight};
  const bool same = (sr[2] - sr[0]) == (dr[2] - dr[0]) && (sr[3] - sr[1]) == (dr[3] - dr[1]);
  const bool useLinear = s.width >= 2 && (flipX || (!same && filter == GL_LINEAR));
  flush_all();
  blit_textures(sid, s, did, d, sr, dr, flipX != 0, flipY != 0, useLinear, !opaque, clip, false);
  download_texture(d);
}
void CompositeYUV(LockedTexture* lockedDst, LockedTexture* lockedY, LockedTexture* lockedU, LockedTexture* lockedV, YuvRangedColorSpace colorSpace,
                  GLuint colorDepth, GLint srcX, GLint srcY, GLsizei srcWidth, GLsizei srcHeight, GLint dstX, GLint dstY, GLsizei dstWidth,
                  GLsizei dstHeight, GLboolean flipX, GLboolean flipY, GLint clipX, GLint clipY, GLsizei clipWidth, GLsizei clipHeight) {
  // composite.h:1330-1386 -> linear_convert_yuv (:1160-1208): three locked R8 (or R16) planes scaled, flipped and clipped into a
  // locked RGBA8 destination through the fixed-point colour matrix; always the linear-filter row walk (linear_row_yuv).
  if (!lockedDst || !lockedY || !lockedU || !lockedV || !ctx) return;
  if (colorSpace < 0 || colorSpace > 6) return;
  Texture& yt = *(Texture*)lockedY; Texture& ut = *(Texture*)lockedU; Texture& vt = *(Texture*)lockedV; Texture& d = *(Texture*)lockedDst;
  if (!yt.dptr || !ut.dptr || !vt.dptr || !d.dptr || d.bpp != 4) return;
  if (yt.bpp != ut.bpp || yt.bpp != vt.bpp || !((yt.bpp == 1 && colorDepth == 8) || (yt.bpp == 2 && colorDepth > 8 && colorDepth <= 16)) ||
      ut.width != vt.width || ut.height != vt.height) {
    // (the reference asserts these: planes of one format -- R8 at 8 bits, R16 above --, chroma planes of one size)
    ctx->last_error = GL_INVALID_OPERATION;
    return;
  }
  const int sr[4] = {srcX - yt.offx, srcY - yt.offy, srcX + srcWidth - yt.offx, srcY + srcHeight - yt.offy};
  const int dr[4] = {dstX - d.offx, dstY - d.offy, dstX + dstWidth - d.offx, dstY + dstHeight - d.offy};
  if (sr[2] <= sr[0] || sr[3] <= sr[1] || dr[2] <= dr[0] || dr[3] <= dr[1]) return;
  const int clip[4] = {clipX - dstX, clipY - dstY, clipX - dstX + clipWidth, clipY - dstY + clipHeight};
  // dstBounds = dsttex.sample_bounds(dstReq) (relative to the request) & clipRect
  int b[4] = {std::max(0, dr[0]) - dr[0], std::max(0, dr[1]) - dr[1], std::min(d.width, dr[2]) - dr[0], std::min(d.height, dr[3]) - dr[1]};
  b[0] = std::max(b[0], clip[0]); b[1] = std::max(b[1], clip[1]); b[2] = std::min(b[2], clip[2]); b[3] = std::min(b[3], clip[3]);
  if (b[2] <= b[0] || b[3] <= b[1]) return;
  flush_all();
  sync_texture_for_read(yt); sync_texture_for_read(ut); sync_texture_for_read(vt);
  sync_texture_for_write(d);
  flush_uploads(3399);
  WrYuvBlitArgs A;
  memset(&A, 0, sizeof(A));
  auto desc = [](const Texture& t) {
    WrTexDesc td; td.ptr = t.dptr; td.width = t.width; td.height = t.height;
    td.stride = t.bpp == 2 ? t.stride / 2 : t.stride; td.format = (int16_t)(t.bpp == 2 ? WR_FMT_R16 : WR_FMT_R8); td.linear = 1;
    td.sw = float(t.width); td.sh = float(t.height);
    return td;
  };
  A.y = desc(yt); A.u = desc(ut); A.v = desc(vt);
  A.dst = d.dptr; A.dst_stride = d.stride;
  A.dx0 = dr[0] + b[0]; A.dy0 = dr[1] + b[1]; A.span = b[2] - b[0]; A.rows = b[3] - b[1];
  A.color_depth = (int)colorDepth;
  {
    // get_ycbcr_info (composite.h:1292-1313, the colour depth forced to 8) -> YUVMatrix::From (:660-719)
    float z0 = 0.0f, z1 = 0.0f, o0 = 1.0f, o1 = 1.0f;
    const float n0 = float(16) / 255.0f, n1 = float(128) / 255.0f, n2 = float(235) / 255.0f, n3 = float(240) / 255.0f;
    if (colorSpace == 0 || colorSpace == 2 || colorSpace == 4) { z0 = n0; z1 = n1; o0 = n2; o1 = n3; }
    else if (colorSpace == 1 || colorSpace == 3 || colorSpace == 5) { z0 = 0.0f; z1 = n1; o0 = 1.0f; o1 = 1.0f; }
    static const float M601[9] = {1.00000f, 1.00000f, 1.00000f, 0.00000f, -0.17207f, 0.88600f, 0.70100f, -0.35707f, 0.00000f};
    static const float M709[9] = {1.00000f, 1.00000f, 1.00000f, 0.00000f, -0.09366f, 0.92780f, 0.78740f, -0.23406f, 0.00000f};
    static const float M2020[9] = {1.00000f, 1.00000f, 1.00000f, 0.00000f, -0.08228f, 0.94070f, 0.73730f, -0.28568f, 0.00000f};
    static const float MID[9] = {0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f};
    const float* Am = colorSpace <= 1 ? M601 : (colorSpace <= 3 ? M709 : (colorSpace <= 5 ? M2020 : MID));
    const float sx = 1.0f / (o0 - z0), sy = 1.0f / (o1 - z1);
    const float Bm[9] = {sx, 0.0f, 0.0f, 0.0f, sy, 0.0f, 0.0f, 0.0f, sy};
    float mat[9];       // rgb_from_yuv * yuv_from_debiased_ycbcr, column-major (glsl.h mat3 * mat3)
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) mat[3 * c + r] = Am[r] * Bm[3 * c] + Am[3 + r] * Bm[3 * c + 1] + Am[6 + r] * Bm[3 * c + 2];
    const double yc = double(mat[1]), rvd = double(mat[6]), gud = double(mat[4]), gvd = double(mat[7]), bud = double(mat[5]);
    A.brmask = mat[0] == 0.0f ? 0 : -1;
    A.bu = int(int16_t(bud * double(1 << 6) + 0.5)); A.rv = int(int16_t(rvd * double(1 << 6) + 0.5));
    A.gu = -int(int16_t(-gud * double(1 << 6) + 0.5)); A.gv = -int(int16_t(-gvd * double(1 << 6) + 0.5));
    A.ycoeff = int(uint16_t(yc * double(1 << 7) + 0.5));
    A.ybias = int(int16_t((double(z0 * 255.0f) * yc - 0.5) * double(1 << 6)));
    A.uvbias = int(int16_t(double(z1 * float(255)) + 0.5));
  }
  // source coordinates (linear_convert_yuv): start, step, flips, the skip to the clamped destination start, the chroma planes' scale
  float su = float(sr[0]), svv = float(sr[1]);
  float du = float(sr[2] - sr[0]) / float(dr[2] - dr[0]), dv = float(sr[3] - sr[1]) / float(dr[3] - dr[1]);
  if (flipX) { su += float(sr[2] - sr[0]); du = -du; }
  if (flipY) { svv += float(sr[3] - sr[1]); dv = -dv; }
  su += du * (float(b[0]) + 0.5f); svv += dv * (float(b[1]) + 0.5f);
  const float csx = float(ut.width) / float(yt.width), csy = float(ut.height) / float(yt.height);
  float cu = su * csx, cvv = svv * csy, cdu = du * csx, cdv = dv * csy;
  A.src_u0 = su; A.chroma_u0 = cu;
  A.nearest = (yt.width < 2 || ut.width < 2) ? 1 : 0;
  if (!A.nearest) {
    su = su * 128.0f + (0.5f - 0.5f * 128.0f); svv = svv * 128.0f + (0.5f - 0.5f * 128.0f); du *= 128.0f; dv *= 128.0f;
    cu = cu * 128.0f + (0.5f - 0.5f * 128.0f); cvv = cvv * 128.0f + (0.5f - 0.5f * 128.0f); cdu *= 128.0f; cdv *= 128.0f;
  }
  A.src_v0 = svv; A.src_dv = dv; A.chroma_v0 = cvv; A.chroma_dv = cdv;
  {
    // linear_row_yuv's row-invariant integers (composite.h:999-1011)
    const int STEP_BITS = 8;
    auto lanes = [&](float x0, float dx, int (&out)[4]) {
      const float l1 = x0 + dx, l2 = l1 + dx, l3 = l2 + dx;
      const float v[4] = {x0, l1, l2, l3};
      for (int i = 0; i < 4; i++) out[i] = (int)(v[i] * float(1 << STEP_BITS));
    };
    lanes(su, du, A.yU); lanes(cu, cdu, A.cU);
    A.yDU = (int)(float(4 << STEP_BITS) * du); A.cDU = (int)(float(4 << STEP_BITS) * cdu);
    A.fast0 = A.fast1 = 0;
    if (!A.nearest && yt.bpp == 1 && A.yDU >= A.cDU && A.cDU > 0 && A.yDU <= (4 << (STEP_BITS + 7)) && A.cDU <= (2 << (STEP_BITS + 7))) {
      // the half-resolution fast path (composite.h:1081-1122): chunks until both coordinates are positive, then as many whole chunks
      // as stay four texels inside both planes
      int span = A.span, chunk = 0;
      int32_t yx = A.yU[0], cx = A.cU[0];
      int32_t cl[4] = {A.cU[0], A.cU[1], A.cU[2], A.cU[3]};
      for (; (yx < 0 || cx < 0) && span >= 4; span -= 4) {
        yx = (int32_t)((uint32_t)yx + (uint32_t)A.yDU); cx = (int32_t)((uint32_t)cx + (uint32_t)A.cDU);
        for (int i = 0; i < 4; i++) cl[i] = (int32_t)((uint32_t)cl[i] + (uint32_t)A.cDU);
        chunk++;
      }
      const int inside = std::min(std::min((((yt.width - 4) << (STEP_BITS + 7)) - yx) / A.yDU, (((ut.width - 4) << (STEP_BITS + 7)) - cx) / A.cDU) * 4, span & ~3);
      if (inside > 0) {
        A.fast0 = chunk; A.fast1 = chunk + inside / 4;
        A.cA = (cl[0] + cl[1]) >> 1; A.cB = (cl[2] + cl[3]) >> 1;      // cU = (cU.xzxz + cU.ywyw) >> 1
      }
    }
  }
  const long long n = (long long)((A.span + 3) / 4) * A.rows;
  WR_LAUNCH(wr_composite_yuv_kernel, (int)((n + 255) / 256), 256, ctx->stream, A);
  ctx->stats.kernel_launches++;
  download_texture(d);
}

// ---- libwrhip additions ------------------------------------------------------
void WrhipGetStats(WrhipStats* out) { if (ctx && out) *out = ctx->stats; }
void WrhipResetStats(void) { if (ctx) { memset(&ctx->stats, 0, sizeof(ctx->stats)); ctx->kstats.clear(); } }
void WrhipSetProfiling(int enabled) {
  if (!ctx) return;
  if (enabled && !ctx->profiling) { flush_all(); flush_uploads(3491); sync_stream(); }   // nothing held back may go out unprofiled
  if (!enabled && ctx->profiling) { flush_all(); flush_uploads(3492); sync_stream(); }
  ctx->profiling = enabled != 0;
  ctx->profiling_deferred = enabled == 2;
}
int32_t WrhipGetKernelStats(WrhipKernelStat* out, int32_t max) {
  if (!ctx || !out) return 0;
  int32_t n = 0;
  for (const WrhipKernelStat& k : ctx->kstats) if (n < max) out[n++] = k;
  return n;
}
void WrhipSetShard(int rank, int world) { if (ctx) { flush_all(); ctx->shard_rank = rank; ctx->shard_world = world < 1 ? 1 : world; } }
void WrhipSetTargetRows(GLuint tex, int32_t y0, int32_t y1) {
  if (!ctx) return;
  flush_all();
  Texture& t = ctx->textures[tex];
  if (y0 <= 0 && y1 >= t.height && t.height > 0) y0 = y1 = 0;     // every row: no restriction (keeps forwarded composites, one rank)
  t.own_y0 = y0; t.own_y1 = y1;
}
void* WrhipGetTextureDevicePtr(GLuint tex, int32_t* width, int32_t* height, int32_t* stride) {
  Texture* t = ctx ? ctx->textures.find(tex) : nullptr;
  if (!t) return nullptr;
  sync_texture_for_read(*t);
  if (width) *width = t->width;
  if (height) *height = t->height;
  if (stride) *stride = t->stride;
  return t->dptr;
}
GLuint WrhipGetFramebufferTexture(GLuint fbo) {
  Framebuffer* fb = ctx ? ctx->framebuffers.find(fbo) : nullptr;
  return fb ? fb->color_attachment : 0;
}
const char* WrhipDeviceName(void) { return g_rt_ok ? g_device_name : nullptr; }
void WrhipFlush(void) {
  if (!ctx) return;
  flush_all();
  flush_uploads(3527);
  drain_tail();
  ctx->held_seq = ctx->flush_seq; ctx->fb_writes_since_held = 0;      // (everything is on the stream: WrhipFlushHeld counts from here)
#ifndef WRHIP_HOSTSIM
  wrq::drain();          // (the caller enqueues on the stream itself next: everything recorded so far must be on it)
#endif
}
// WrhipFlush without draining the held-back raster launches: everything recorded so far is processed -- this flush's upload and
// setup go out, and with them the raster launches the PREVIOUS flush left pending -- while this flush's own raster launches stay
// held for the next one (Context::Tail).  Returns 1 if they are held, 0 if they were launched as well (deferral off / profiling).
// The sharded frame loop moves frame k's strips right after frame k + 1's WrhipFlushHeld: frame k is complete on the stream,
// frame k + 1 has not touched a pixel yet.
// Returns 2 when the invariant the pipelined exchange rests on is broken: more than one flush since the previous call (a
// flush in the middle of the frame) and an earlier one of them already wrote the default framebuffer -- part of this frame's
// window is on the stream ahead of the previous frame's exchange.
int WrhipFlushHeld(void) {
  if (!ctx) return 0;
  const int64_t seq0 = ctx->held_seq;
  flush_all();
  flush_uploads(3546);
  const int64_t n = ctx->flush_seq - seq0;
  const bool early_fb = n > 1 && ctx->fb_writes_since_held - (ctx->last_flush_wrote_fb ? 1 : 0) > 0;
  ctx->held_seq = ctx->flush_seq;
  ctx->fb_writes_since_held = 0;
  if (n == 0 && ctx->tail.pending) drain_tail();      // nothing was recorded since: what the previous flush held back goes out now
#ifndef WRHIP_HOSTSIM
  wrq::drain();
#endif
  if (early_fb) return 2;
  return ctx->tail.pending ? 1 : 0;
}
void* WrhipGetStream(void) {
#ifdef WRHIP_HOSTSIM
  return nullptr;
#else
  return ctx ? (void*)ctx->stream : nullptr;
#endif
}

}  // extern "C"
