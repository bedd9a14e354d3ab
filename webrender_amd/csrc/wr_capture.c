/* wr_capture.c -- GL-trace capture interposer.
 *
 * A library that exports the 99 functions of the swgl ABI (swgl/src/swgl_fns.rs:23-322), records every call with its
 * payload in the WRTR trace format of webrender_amd/trace.py, and forwards it to a real backend:
 *
 *   WR_CAPTURE_BACKEND=/path/to/backend.so   the library the calls go to (the reference's gl_cc build, or libwrhip.so)
 *   WR_CAPTURE_FILE=/path/to/frame.wrtr      where the trace is written (when the last context is destroyed, at exit,
 *                                            or on WrCaptureWrite())
 *
 * Linked in place of swgl's gl_cc static library (integration/README.md), it turns a `wrench --software` run -- real
 * display lists through the real frame builder -- into a call stream that tools/ and the tests here can replay against
 * libwrhip and the oracle (csrc/wr_replay.c, trace.NativeReplayer).  Writes through mapped buffers are turned into the
 * equivalent BufferSubData at UnmapBuffer time; handles returned by Lock* / GetResourceBuffer (Gecko's compositor entry
 * points) cannot be represented and are recorded as nulls.  Single-threaded, like the ABI (gl.cc:866-869).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { TAG_INT, TAG_F32, TAG_F64, TAG_BLOB, TAG_NULL, TAG_SCRATCH, TAG_CTX, TAG_SCRATCH_INIT };
typedef struct { uint32_t tag, aux; uint64_t value; } wr_arg;
typedef struct { uint16_t id, nargs; wr_arg a[24]; } wr_rec;

#define GL_PIXEL_PACK_BUFFER 0x88EB
#define GL_PIXEL_UNPACK_BUFFER 0x88EC
#define GL_ARRAY_BUFFER 0x8892
#define GL_ELEMENT_ARRAY_BUFFER 0x8893
#define GL_UNPACK_ROW_LENGTH 0x0CF2

static struct {
  int ready;
  void* dl;
  void* fn[128];
  /* recorded stream */
  uint8_t* calls; size_t calls_len, calls_cap; uint32_t n_calls;
  uint8_t* blobs; size_t blobs_len, blobs_cap;
  size_t scratch;
  wr_rec cur;
  /* tracked GL state */
  int unpack_row_length;
  uint32_t unpack_buffer, pack_buffer, array_buffer, element_buffer;
  struct { uint32_t target; void* ptr; intptr_t offset; size_t len; int write; } map[4];
  int contexts;
  int in_synth;
} G;

static void wr_cap_write(void);
static void wr_cap_init(void);

static void grow(uint8_t** p, size_t* cap, size_t need) {
  if (need <= *cap) return;
  size_t c = *cap ? *cap * 2 : (1u << 20);
  while (c < need) c *= 2;
  *p = (uint8_t*)realloc(*p, c); *cap = c;
}
static wr_rec* wr_begin(int id, int nargs) { G.cur.id = (uint16_t)id; G.cur.nargs = (uint16_t)nargs; memset(G.cur.a, 0, sizeof(G.cur.a)); return &G.cur; }
static void wr_arg_int(wr_rec* r, int i, uint64_t v) { r->a[i].tag = TAG_INT; r->a[i].value = v; }
/* one argument more than the call has (GetUniformLocation: the location the backend returned -- backend-specific, the replayer translates) */
static void wr_arg_extra_int(wr_rec* r, int i, uint64_t v) { r->nargs = (uint16_t)(i + 1); wr_arg_int(r, i, v); }
static void wr_arg_f32(wr_rec* r, int i, float f) { uint32_t u; memcpy(&u, &f, 4); r->a[i].tag = TAG_F32; r->a[i].value = u; }
static void wr_arg_f64(wr_rec* r, int i, double d) { r->a[i].tag = TAG_F64; memcpy(&r->a[i].value, &d, 8); }
static void wr_arg_null(wr_rec* r, int i) { r->a[i].tag = TAG_NULL; }
static void wr_arg_ctx(wr_rec* r, int i, const void* p) { r->a[i].tag = p ? TAG_CTX : TAG_NULL; }
static void wr_arg_blob(wr_rec* r, int i, const void* p, size_t n) {
  if (!p) { r->a[i].tag = TAG_NULL; return; }
  size_t off = (G.blobs_len + 15) & ~(size_t)15;
  grow(&G.blobs, &G.blobs_cap, off + n + 16);
  memset(G.blobs + G.blobs_len, 0, off - G.blobs_len);
  memcpy(G.blobs + off, p, n);
  G.blobs_len = off + n;
  r->a[i].tag = TAG_BLOB; r->a[i].aux = (uint32_t)n; r->a[i].value = off;
}
static void wr_arg_scratch(wr_rec* r, int i, const void* p, size_t n) {
  if (!p) { r->a[i].tag = TAG_NULL; return; }
  size_t off = (G.scratch + 63) & ~(size_t)63;
  G.scratch = off + n;
  r->a[i].tag = TAG_SCRATCH; r->a[i].aux = (uint32_t)n; r->a[i].value = off;
}
/* caller memory the backend both reads at the call and writes later (SetTextureBuffer's backing store): the bytes it holds now
 * travel as a blob, the replayer copies them into scratch space of its own and hands that out.  value = blob offset << 32 |
 * scratch offset: both have to fit 32 bits, as the header's totals do (wr_cap_write refuses larger traces). */
static void wr_arg_scratch_init(wr_rec* r, int i, const void* p, size_t n) {
  if (!p) { r->a[i].tag = TAG_NULL; return; }
  wr_arg_blob(r, i, p, n);
  const uint64_t blob_off = r->a[i].value;
  size_t off = (G.scratch + 63) & ~(size_t)63;
  G.scratch = off + n;
  if (blob_off > 0xFFFFFFFFull || off > 0xFFFFFFFFull) { fprintf(stderr, "wr_capture: trace exceeds 4 GiB of payload: not representable in the WRTR format\n"); abort(); }
  r->a[i].tag = TAG_SCRATCH_INIT; r->a[i].aux = (uint32_t)n; r->a[i].value = (blob_off << 32) | (uint64_t)off;
}
static void wr_commit(const wr_rec* r) {
  size_t n = 4 + (size_t)r->nargs * sizeof(wr_arg);
  grow(&G.calls, &G.calls_cap, G.calls_len + n);
  memcpy(G.calls + G.calls_len, &r->id, 2); memcpy(G.calls + G.calls_len + 2, &r->nargs, 2);
  memcpy(G.calls + G.calls_len + 4, r->a, (size_t)r->nargs * sizeof(wr_arg));
  G.calls_len += n; G.n_calls++;
}

static size_t wr_pixel_bytes(uint32_t format, uint32_t type) {
  size_t comps = 4;
  switch (format) {
    case 0x1903: comps = 1; break;            /* GL_RED */
    case 0x8227: comps = 2; break;            /* GL_RG */
    case 0x1907: comps = 3; break;            /* GL_RGB */
    case 0x85BB: comps = 2; break;            /* GL_RGB_422_APPLE: 2 bytes per pixel */
    case 0x1902: comps = 1; break;            /* GL_DEPTH_COMPONENT */
    default: comps = 4; break;                /* GL_RGBA, GL_BGRA, GL_RGBA_INTEGER */
  }
  switch (type) {
    case 0x1406: case 0x1404: case 0x1405: return comps * 4;     /* FLOAT, INT, UNSIGNED_INT */
    case 0x1403: case 0x1402: return comps * 2;                  /* UNSIGNED_SHORT, SHORT */
    case 0x8367: return 4;                                       /* UNSIGNED_INT_8_8_8_8_REV */
    case 0x85BA: case 0x85BB: return 2;                          /* UNSIGNED_SHORT_8_8(_REV)_APPLE */
    default: return comps;                                       /* UNSIGNED_BYTE */
  }
}
static size_t wr_image_bytes(int32_t w, int32_t h, uint32_t format, uint32_t type) {
  if (w <= 0 || h <= 0) return 0;
  size_t bpp = wr_pixel_bytes(format, type);
  size_t row = (size_t)(G.unpack_row_length > 0 ? G.unpack_row_length : w);
  return ((size_t)(h - 1) * row + (size_t)w) * bpp;
}

static int fn_id(const char* name);
static void wr_post(int id);
/* state tracking (before the call is forwarded) */
static uint32_t* binding(uint32_t target) {
  switch (target) {
    case GL_PIXEL_PACK_BUFFER: return &G.pack_buffer;
    case GL_PIXEL_UNPACK_BUFFER: return &G.unpack_buffer;
    case GL_ARRAY_BUFFER: return &G.array_buffer;
    case GL_ELEMENT_ARRAY_BUFFER: return &G.element_buffer;
    default: return NULL;
  }
}
static void wr_pre_BindBuffer(uint32_t target, uint32_t buffer) { uint32_t* b = binding(target); if (b) *b = buffer; }
static void wr_pre_DeleteBuffer(uint32_t n) {
  if (G.pack_buffer == n) G.pack_buffer = 0;
  if (G.unpack_buffer == n) G.unpack_buffer = 0;
  if (G.array_buffer == n) G.array_buffer = 0;
}
static void wr_pre_PixelStorei(uint32_t name, int32_t v) { if (name == GL_UNPACK_ROW_LENGTH) G.unpack_row_length = v; }
static size_t* buf_size; static size_t buf_size_n;      /* by buffer id (grown on demand: ids are small integers, first-free) */
static size_t* buf_slot(uint32_t id) {
  if (id >= buf_size_n) { size_t n = buf_size_n ? buf_size_n : 4096; while (n <= id) n *= 2; buf_size = (size_t*)realloc(buf_size, n * sizeof(size_t)); memset(buf_size + buf_size_n, 0, (n - buf_size_n) * sizeof(size_t)); buf_size_n = n; }
  return &buf_size[id];
}
static void wr_pre_BufferData(uint32_t target, size_t size, void* data, uint32_t usage) {
  (void)data; (void)usage;
  uint32_t* b = binding(target);
  if (b) *buf_slot(*b) = size;
}
static size_t wr_bound_size(uint32_t target) { uint32_t* b = binding(target); return b ? *buf_slot(*b) : 0; }
static void wr_note_mapping(uint32_t target, void* ptr, intptr_t offset, size_t len, int write) {
  if (!ptr) return;
  for (int k = 0; k < 4; k++)
    if (!G.map[k].ptr) { G.map[k].target = target; G.map[k].ptr = ptr; G.map[k].offset = offset; G.map[k].len = len; G.map[k].write = write; return; }
}
/* Bytes the caller wrote through a mapping reach the backend without any call carrying them: at UnmapBuffer the mapped
 * range is snapshotted and recorded as a BufferSubData ahead of the UnmapBuffer itself. */
static void wr_pre_UnmapBuffer(uint32_t target) {
  for (int k = 0; k < 4; k++) {
    if (G.map[k].ptr && G.map[k].target == target) {
      if (G.map[k].write && G.map[k].len) {
        wr_rec saved = G.cur;
        const int id = fn_id("BufferSubData");
        if (id >= 0) {
          wr_rec* r = wr_begin(id, 4);
          wr_arg_int(r, 0, target); wr_arg_int(r, 1, (uint64_t)G.map[k].offset); wr_arg_int(r, 2, (uint64_t)G.map[k].len);
          wr_arg_blob(r, 3, G.map[k].ptr, G.map[k].len);
          wr_commit(r);
        }
        G.cur = saved;
      }
      G.map[k].ptr = NULL;
    }
  }
}

#include "wr_capture_gen.h"

static void wr_cap_init(void) {
  if (G.ready) return;
  G.ready = 1;
  const char* be = getenv("WR_CAPTURE_BACKEND");
  if (!be) { fprintf(stderr, "wr_capture: WR_CAPTURE_BACKEND is not set\n"); abort(); }
  G.dl = dlopen(be, RTLD_NOW | RTLD_LOCAL);
  if (!G.dl) { fprintf(stderr, "wr_capture: dlopen(%s): %s\n", be, dlerror()); abort(); }
  for (int i = 0; i < WR_FN_COUNT; i++) {
    G.fn[i] = dlsym(G.dl, WR_FN_NAMES[i]);
    if (!G.fn[i]) { fprintf(stderr, "wr_capture: %s lacks symbol %s\n", be, WR_FN_NAMES[i]); abort(); }
  }
  atexit(wr_cap_write);
}


static int fn_id(const char* name) { for (int i = 0; i < WR_FN_COUNT; i++) if (!strcmp(WR_FN_NAMES[i], name)) return i; return -1; }

/* after the call was forwarded: commit the record; remember mappings; count contexts */
static void wr_post(int id) {
  static int id_create = -2, id_destroy;
  if (id_create == -2) { id_create = fn_id("CreateContext"); id_destroy = fn_id("DestroyContext"); }
  wr_commit(&G.cur);
  if (id == id_create) G.contexts++;
  if (id == id_destroy && --G.contexts <= 0) wr_cap_write();
}

static void wr_cap_write(void) {
  const char* path = getenv("WR_CAPTURE_FILE");
  if (!path || !G.n_calls) return;
  if (G.blobs_len > 0xFFFFFFFFull || G.scratch + 64 > 0xFFFFFFFFull) {
    /* the WRTR header holds both totals in 32 bits: a larger session cannot be written faithfully -- say so instead of truncating */
    fprintf(stderr, "wr_capture: %zu payload bytes / %zu scratch bytes exceed the format's 4 GiB: %s NOT written\n", G.blobs_len, G.scratch, path);
    return;
  }
  FILE* f = fopen(path, "wb");
  if (!f) { fprintf(stderr, "wr_capture: cannot write %s\n", path); return; }
  uint32_t hdr[4] = {0x52545257u /* 'WRTR' */, G.n_calls, (uint32_t)G.blobs_len, (uint32_t)(G.scratch + 64)};
  fwrite(hdr, 4, 4, f);
  fwrite(G.calls, 1, G.calls_len, f);
  fwrite(G.blobs, 1, G.blobs_len, f);
  fclose(f);
}
void WrCaptureWrite(void) { wr_cap_write(); }
