/* wr_replay.c -- native replayer for GL command traces (webrender_amd/trace.py).
 *
 * Plays the role `wrench perf` plays for the reference (wrench/src/perf.rs:
 * 198-270): issue the frame's C-ABI call stream from native code and time it
 * with CLOCK_MONOTONIC, so that interpreter overhead is not part of frame time.
 * Works against any library exporting the swgl ABI (libwrhip or the oracle).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { uint32_t tag, aux; uint64_t value; } wr_arg;
typedef struct wr_replay {
  void* dl;
  void* fn[128];
  void* ctx;
  const uint8_t* blobs;
  uint8_t* scratch;
  size_t scratch_size;
  /* uniform locations are backend-specific (swgl numbers them per program in order of first use, the generated
   * get_uniform(); libwrhip by sampler slot): GetUniformLocation records carry the location the RECORDING backend
   * returned (trace.py), and Uniform1i / Uniform4fv / UniformMatrix4fv are issued with the location THIS backend gave
   * for the same (program, name). */
  uint32_t cur_program;
  struct { uint32_t program; int32_t recorded, actual; } locs[1024];
  int n_locs;
  int id_get_uniform, id_use_program, id_uniform1i, id_uniform4fv, id_uniform_matrix4fv;
} wr_replay;

enum { TAG_INT, TAG_F32, TAG_F64, TAG_BLOB, TAG_NULL, TAG_SCRATCH, TAG_CTX, TAG_SCRATCH_INIT };

static inline float wr_f32(const wr_arg* a) { float f; uint32_t u = (uint32_t)a->value; memcpy(&f, &u, 4); return f; }
static inline double wr_f64(const wr_arg* a) { double d; memcpy(&d, &a->value, 8); return d; }
static inline void* wr_ptr(wr_replay* R, const wr_arg* a) {
  switch (a->tag) {
    case TAG_BLOB: return (void*)(R->blobs + a->value);
    case TAG_SCRATCH: return (void*)(R->scratch + a->value);
    case TAG_SCRATCH_INIT: {     /* scratch the callee reads now and writes later: starts as the recorded bytes (value = blob << 32 | scratch) */
      void* p = (void*)(R->scratch + (a->value & 0xFFFFFFFFull));
      memcpy(p, R->blobs + (a->value >> 32), a->aux);
      return p;
    }
    case TAG_CTX: return R->ctx;
    case TAG_INT: return (void*)(uintptr_t)a->value;
    default: return NULL;
  }
}

#include "wr_replay_gen.h"

wr_replay* wr_replay_open(const char* path) {
  void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) { fprintf(stderr, "wr_replay: dlopen(%s): %s\n", path, dlerror()); return NULL; }
  wr_replay* R = (wr_replay*)calloc(1, sizeof(wr_replay));
  R->dl = dl;
  for (int i = 0; i < WR_FN_COUNT; i++) {
    R->fn[i] = dlsym(dl, WR_FN_NAMES[i]);
    if (!R->fn[i]) { fprintf(stderr, "wr_replay: %s lacks symbol %s\n", path, WR_FN_NAMES[i]); free(R); return NULL; }
    if (!strcmp(WR_FN_NAMES[i], "GetUniformLocation")) R->id_get_uniform = i;
    if (!strcmp(WR_FN_NAMES[i], "UseProgram")) R->id_use_program = i;
    if (!strcmp(WR_FN_NAMES[i], "Uniform1i")) R->id_uniform1i = i;
    if (!strcmp(WR_FN_NAMES[i], "Uniform4fv")) R->id_uniform4fv = i;
    if (!strcmp(WR_FN_NAMES[i], "UniformMatrix4fv")) R->id_uniform_matrix4fv = i;
  }
  return R;
}

void* wr_replay_sym(wr_replay* R, const char* name) { return dlsym(R->dl, name); }

static int run_once(wr_replay* R, const uint8_t* t, size_t len) {
  if (len < 16 || memcmp(t, "WRTR", 4) != 0) return 1;
  uint32_t n_calls, blob_bytes, scratch_bytes;
  memcpy(&n_calls, t + 4, 4); memcpy(&blob_bytes, t + 8, 4); memcpy(&scratch_bytes, t + 12, 4);
  if (R->scratch_size < scratch_bytes) {
    free(R->scratch);
    R->scratch = (uint8_t*)calloc(1, scratch_bytes + 64);
    R->scratch_size = scratch_bytes;
  }
  R->blobs = t + len - blob_bytes;
  const uint8_t* p = t + 16;
  for (uint32_t i = 0; i < n_calls; i++) {
    uint16_t id, nargs;
    memcpy(&id, p, 2); memcpy(&nargs, p + 2, 2);
    p += 4;
    wr_arg args[24];
    memcpy(args, p, (size_t)nargs * sizeof(wr_arg));
    p += (size_t)nargs * sizeof(wr_arg);
    if (id == R->id_get_uniform && nargs == 3) {
      const uint32_t prog = (uint32_t)args[0].value;
      const int32_t actual = ((int32_t(*)(uint32_t, const char*))R->fn[id])(prog, (const char*)wr_ptr(R, &args[1]));
      const int32_t recorded = (int32_t)args[2].value;
      int k = 0;
      while (k < R->n_locs && !(R->locs[k].program == prog && R->locs[k].recorded == recorded)) k++;
      if (k == R->n_locs && R->n_locs < 1024) R->n_locs++;
      if (k < 1024) { R->locs[k].program = prog; R->locs[k].recorded = recorded; R->locs[k].actual = actual; }
      continue;
    }
    if (id == R->id_use_program) R->cur_program = (uint32_t)args[0].value;
    if (id == R->id_uniform1i || id == R->id_uniform4fv || id == R->id_uniform_matrix4fv) {
      const int32_t recorded = (int32_t)args[0].value;
      for (int k = 0; k < R->n_locs; k++)
        if (R->locs[k].program == R->cur_program && R->locs[k].recorded == recorded) { args[0].value = (uint64_t)(int64_t)R->locs[k].actual; break; }
    }
    if (wr_dispatch(R, id, args) != 0) return (int)i + 1;
  }
  return 0;
}

int wr_replay_exec(wr_replay* R, const uint8_t* trace, size_t len) { return run_once(R, trace, len); }

/* Throughput loop: replays the frame trace `iters` times back to back, then calls
 * the backend's Finish() once.  Returns total wall milliseconds in *total_ms. */
int wr_replay_stream(wr_replay* R, const uint8_t* trace, size_t len, int iters, double* total_ms) {
  void (*finish)(void) = (void (*)(void))dlsym(R->dl, "Finish");
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < iters; i++) { int rc = run_once(R, trace, len); if (rc) return rc; }
  struct timespec c;
  clock_gettime(CLOCK_MONOTONIC, &c);
  if (finish) finish();
  clock_gettime(CLOCK_MONOTONIC, &b);
  if (getenv("WR_REPLAY_TIMING"))   /* host issue time vs. total: is the stream CPU- or GPU-bound? */
    fprintf(stderr, "wr_replay: %d frames issued in %.3f ms, finished after %.3f ms\n", iters,
            (c.tv_sec - a.tv_sec) * 1e3 + (c.tv_nsec - a.tv_nsec) * 1e-6, (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6);
  *total_ms = (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
  return 0;
}

int wr_replay_loop(wr_replay* R, const uint8_t* trace, size_t len, int warmup, int iters, double* ms_out) {
  for (int i = 0; i < warmup; i++) { int rc = run_once(R, trace, len); if (rc) return rc; }
  for (int i = 0; i < iters; i++) {
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    int rc = run_once(R, trace, len);
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (rc) return rc;
    ms_out[i] = (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
  }
  return 0;
}

void* wr_replay_scratch(wr_replay* R, size_t offset) { return R->scratch + offset; }

void wr_replay_close(wr_replay* R) {
  if (!R) return;
  free(R->scratch);
  /* the backend library stays loaded: contexts may own device state */
  free(R);
}
