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
#include "../../include/wrhip.h"       /* WrhipStats (the sharded loop counts a frame's flushes) */

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
#define WR_MAX_LOCS 4096
  uint32_t cur_program;
  int cur_first;
  struct { uint32_t program; int32_t recorded, actual; } locs[WR_MAX_LOCS];
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
      if (recorded < 0) continue;           /* a name the recording backend did not have: Uniform*(-1, ..) stays a no-op (GL), never translated */
      int k = 0;
      while (k < R->n_locs && !(R->locs[k].program == prog && R->locs[k].recorded == recorded)) k++;
      if (k == R->n_locs) {
        if (R->n_locs >= WR_MAX_LOCS) { fprintf(stderr, "wr_replay: more than %d uniform locations in one trace\n", WR_MAX_LOCS); return -4; }
        R->n_locs++;
      }
      R->locs[k].program = prog; R->locs[k].recorded = recorded; R->locs[k].actual = actual;
      continue;
    }
    if (id == R->id_use_program) {
      R->cur_program = (uint32_t)args[0].value;
      R->cur_first = 0;                      /* the program's entries are contiguous (queried right after linking): remember where they start */
      while (R->cur_first < R->n_locs && R->locs[R->cur_first].program != R->cur_program) R->cur_first++;
    }
    if (id == R->id_uniform1i || id == R->id_uniform4fv || id == R->id_uniform_matrix4fv) {
      const int32_t recorded = (int32_t)args[0].value;
      if (recorded >= 0) {
        int k = R->cur_first;
        while (k < R->n_locs && !(R->locs[k].program == R->cur_program && R->locs[k].recorded == recorded)) k++;
        if (k < R->n_locs) args[0].value = (uint64_t)(int64_t)R->locs[k].actual;       /* (a location the trace never queried is passed as recorded) */
      }
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

/* ---------------------------------------------------------------------------------------------------------------------
 * Sharded stream: the per-frame loop of the multi-GPU harness in native code (webrender_amd/dist.py
 * does the setup: which rows of which target this rank owns).  Every frame: replay the call stream, WrhipFlush() so that
 * everything recorded is on the backend's stream, then move the window strips -- nothing else crosses ranks.
 *
 *   RCCL transport  grouped ncclSend / ncclRecv on the backend's own stream, in place: a rank sends the rows it owns
 *                   straight out of its window, the receiver takes them straight into the same rows of ITS window (no
 *                   staging copy, no host sync; RCCL moves peer-to-peer over xGMI).  mode 0: every rank -> rank 0 (the
 *                   presenting GPU); mode 1: every rank -> every rank.  librccl is dlopen'ed; the communicator is made
 *                   from a unique id the caller distributes (dist.py: torch.distributed, once, at setup).
 *   shm transport   the CPU stand-in for the tests (world-2/3, host-simulation backend): the "window of the presenting
 *                   GPU" is a POSIX shared-memory segment, a rank's strip is memcpy'd into its rows (the peer write), an
 *                   arrival counter in the segment tells when a frame is complete.
 */
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

typedef struct { char internal[128]; } wr_nccl_id;
typedef struct wr_shard {
  wr_replay* R;
  int rank, world, mode;
  int height, row_bytes;
  int rows[2 * 64];                 /* framebuffer rows [y0, y1) of every rank's strip */
  uint8_t* fb;                      /* this rank's window storage (device pointer; host pointer with the hostsim backend) */
  void (*flush)(void);
  int (*flush_held)(void);        /* optional: WrhipFlushHeld */
  void (*get_stats)(WrhipStats*);
  int probed;                     /* the first frame of a stream has been replayed unpipelined and its flushes counted ... */
  int probe_flushes;              /* ... this many */
  int err_held;                   /* a frame that wrote the window and flushed again before its end was seen (reported at the end of the stream) */
  void (*finish)(void);
  void* (*get_stream)(void);
  /* RCCL */
  void* nccl_dl; void* comm;
  int (*group_start)(void); int (*group_end)(void);
  int (*send)(const void*, size_t, int, int, void*, void*);
  int (*recv)(void*, size_t, int, int, void*, void*);
  int (*comm_destroy)(void*);
  /* shm */
  uint8_t* shm; size_t shm_size; uint64_t frames_done;
} wr_shard;

int wr_shard_unique_id(const char* librccl, void* out128) {
  void* dl = dlopen(librccl, RTLD_NOW | RTLD_GLOBAL);
  if (!dl) { fprintf(stderr, "wr_shard: dlopen(%s): %s\n", librccl, dlerror()); return 1; }
  int (*get_id)(wr_nccl_id*) = (int (*)(wr_nccl_id*))dlsym(dl, "ncclGetUniqueId");
  if (!get_id) return 2;
  return get_id((wr_nccl_id*)out128);
}

static wr_shard* shard_new(wr_replay* R, int rank, int world, int mode) {
  if (world < 1 || world > 64) return NULL;
  wr_shard* S = (wr_shard*)calloc(1, sizeof(wr_shard));
  S->R = R; S->rank = rank; S->world = world; S->mode = mode;
  S->flush = (void (*)(void))dlsym(R->dl, "WrhipFlush");
  S->flush_held = getenv("WRHIP_SHARD_NO_PIPELINE") ? NULL : (int (*)(void))dlsym(R->dl, "WrhipFlushHeld");
  S->finish = (void (*)(void))dlsym(R->dl, "Finish");
  S->get_stats = (void (*)(WrhipStats*))dlsym(R->dl, "WrhipGetStats");
  S->get_stream = (void* (*)(void))dlsym(R->dl, "WrhipGetStream");
  if (!S->flush || !S->finish || !S->get_stream) { fprintf(stderr, "wr_shard: backend lacks WrhipFlush / WrhipGetStream\n"); free(S); return NULL; }
  return S;
}

wr_shard* wr_shard_open_rccl(wr_replay* R, const char* librccl, int rank, int world, int mode, const void* unique_id128) {
  wr_shard* S = shard_new(R, rank, world, mode);
  if (!S) return NULL;
  if (world == 1) return S;         /* nothing to exchange: the strip is the window */
  S->nccl_dl = dlopen(librccl, RTLD_NOW | RTLD_GLOBAL);
  if (!S->nccl_dl) { fprintf(stderr, "wr_shard: dlopen(%s): %s\n", librccl, dlerror()); free(S); return NULL; }
  int (*init)(void**, int, wr_nccl_id, int) = (int (*)(void**, int, wr_nccl_id, int))dlsym(S->nccl_dl, "ncclCommInitRank");
  S->group_start = (int (*)(void))dlsym(S->nccl_dl, "ncclGroupStart");
  S->group_end = (int (*)(void))dlsym(S->nccl_dl, "ncclGroupEnd");
  S->send = (int (*)(const void*, size_t, int, int, void*, void*))dlsym(S->nccl_dl, "ncclSend");
  S->recv = (int (*)(void*, size_t, int, int, void*, void*))dlsym(S->nccl_dl, "ncclRecv");
  S->comm_destroy = (int (*)(void*))dlsym(S->nccl_dl, "ncclCommDestroy");
  if (!init || !S->group_start || !S->group_end || !S->send || !S->recv) { fprintf(stderr, "wr_shard: librccl lacks the p2p API\n"); free(S); return NULL; }
  wr_nccl_id id;
  memcpy(&id, unique_id128, sizeof(id));
  int rc = init(&S->comm, world, id, rank);
  if (rc != 0) { fprintf(stderr, "wr_shard: ncclCommInitRank failed (%d)\n", rc); free(S); return NULL; }
  return S;
}

/* shm transport: rank 0 removes a segment a crashed run may have left under this name (its arrival counter would be stale);
   the caller puts a barrier between this and the ranks' wr_shard_open_shm */
void wr_shard_shm_reset(const char* name) { shm_unlink(name); }

wr_shard* wr_shard_open_shm(wr_replay* R, const char* name, int rank, int world, int mode, size_t window_bytes) {
  wr_shard* S = shard_new(R, rank, world, mode);
  if (!S) return NULL;
  S->shm_size = window_bytes + 4096;
  int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)S->shm_size) != 0) { fprintf(stderr, "wr_shard: shm_open(%s) failed\n", name); free(S); return NULL; }
  S->shm = (uint8_t*)mmap(NULL, S->shm_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (S->shm == MAP_FAILED) { free(S); return NULL; }
  return S;
}

void wr_shard_set_window(wr_shard* S, void* fb, int height, int row_bytes, const int* rows2w) {
  S->fb = (uint8_t*)fb; S->height = height; S->row_bytes = row_bytes;
  memcpy(S->rows, rows2w, sizeof(int) * 2 * (size_t)S->world);
}

static size_t strip_off(const wr_shard* S, int r) { return (size_t)S->rows[2 * r] * (size_t)S->row_bytes; }
static size_t strip_len(const wr_shard* S, int r) { int n = S->rows[2 * r + 1] - S->rows[2 * r]; return n > 0 ? (size_t)n * (size_t)S->row_bytes : 0; }

static int shard_exchange(wr_shard* S) {
  if (S->shm) {
    /* the peer write: this rank's rows into the shared window; then tell */
    const size_t n = strip_len(S, S->rank);
    if (n) memcpy(S->shm + 4096 + strip_off(S, S->rank), S->fb + strip_off(S, S->rank), n);
    __atomic_add_fetch((uint64_t*)S->shm, 1, __ATOMIC_RELEASE);
    S->frames_done++;
    return 0;
  }
  if (S->world == 1) return 0;
  void* stream = S->get_stream();
  const int me = S->rank;
  int rc = S->group_start();
  for (int p = 0; p < S->world && rc == 0; p++) {
    if (p == me) continue;
    const int to_p = S->mode == 1 || p == 0, from_p = S->mode == 1 || me == 0;
    if (to_p && strip_len(S, me)) rc = S->send(S->fb + strip_off(S, me), strip_len(S, me), 1 /* ncclUint8 */, p, S->comm, stream);
    if (rc == 0 && from_p && strip_len(S, p)) rc = S->recv(S->fb + strip_off(S, p), strip_len(S, p), 1, p, S->comm, stream);
  }
  const int rc2 = S->group_end();
  return rc ? rc : rc2;
}

/* The pipelined order of wr_shard_stream2 is only right for a frame that flushes ONCE: a flush in the middle of frame k + 1 (a
   sampled target overwritten, a query, ring pressure ...) puts its first segment on the stream ahead of frame k's exchange.  The
   first frame of a stream is therefore replayed unpipelined and its flushes counted.  Every rank has to take the SAME order (the
   collectives match by call order: a rank that dropped to the unpipelined order alone would combine frame k's strips with its
   peers' frame k - 1), and flush counts differ per rank (the targets kept per strip differ): the caller -- webrender_amd/dist.py --
   probes on every rank, takes the maximum over the ranks and tells every rank the outcome (wr_shard_set_pipelined).  A stream that
   is started without that (one process, the tests' single-rank paths) decides from its own count.  Returns the flushes of the
   probe frame, < 0 on error. */
int wr_shard_probe(wr_shard* S, const uint8_t* trace, size_t len) {
  if (S->probed) return S->probe_flushes;
  WrhipStats st0, st1;
  memset(&st0, 0, sizeof(st0)); memset(&st1, 0, sizeof(st1));
  if (S->get_stats) S->get_stats(&st0);
  int rc0 = run_once(S->R, trace, len);
  if (rc0) return rc0 > 0 ? -rc0 : rc0;
  if (S->shm) S->finish(); else S->flush();
  if (S->get_stats) S->get_stats(&st1);
  S->probed = 1;
  S->probe_flushes = S->get_stats ? (int)(st1.flushes - st0.flushes) : 2;      /* (no counter: unpipelined) */
  if (shard_exchange(S)) { fprintf(stderr, "wr_shard: exchange failed\n"); return -2; }
  return S->probe_flushes;
}
void wr_shard_set_pipelined(wr_shard* S, int on) {
  if (!on && S->flush_held) {
    fprintf(stderr, "wr_shard: a rank's frame flushes more than once: the strips are moved unpipelined on every rank\n");
    S->flush_held = NULL;
  }
}

/* `iters` frames back to back, the strips moved after every one, one Finish at the end.  Returns total wall ms in *total_ms. */
int wr_shard_stream2(wr_shard* S, const uint8_t* trace, size_t len, const uint8_t* trace_b, size_t len_b, int iters, double* total_ms);
int wr_shard_stream(wr_shard* S, const uint8_t* trace, size_t len, int iters, double* total_ms) {
  return wr_shard_stream2(S, trace, len, trace, len, iters, total_ms);
}
/* ... frames alternating between two traces of one window (even frames: `trace`, odd ones: `trace_b`): what the tests use to see
   that every exchange moves the strips of ITS frame (with one trace a late or early exchange moves identical bytes) */
int wr_shard_stream2(wr_shard* S, const uint8_t* trace_a, size_t len_a, const uint8_t* trace_b, size_t len_b, int iters, double* total_ms) {
  struct timespec a, b;
  const int shm_sync = S->shm && !getenv("WRHIP_SHARD_PIPELINE_SHM");     /* (the stand-in can take the pipelined order too: the tests do both) */
  clock_gettime(CLOCK_MONOTONIC, &a);
  /* GPU path, software-pipelined by one frame: the backend holds a flush's raster launches back until the next flush (they leave
     fused with its setup stage), so frame k's strips are moved right after frame k + 1's WrhipFlushHeld -- frame k is complete on
     the stream by then and frame k + 1 has not written a pixel -- and the host records frame k + 1 while frame k rasterises,
     exactly as the unsharded stream does. */
  int pending = 0;                                  /* the previous frame's strips are still to be moved */
  for (int i = 0; i < iters; i++) {
    const uint8_t* trace = (i & 1) ? trace_b : trace_a;
    const size_t len = (i & 1) ? len_b : len_a;
    if (!S->probed && !shm_sync && S->flush_held) {
      /* (not probed by the caller: this process decides alone -- see wr_shard_probe) */
      const int n = wr_shard_probe(S, trace, len);
      if (n < 0) return n;
      wr_shard_set_pipelined(S, n == 1);
      continue;
    }
    int rc = run_once(S->R, trace, len);
    if (rc) return rc;
    if (shm_sync || !S->flush_held) {
      if (S->shm) S->finish(); else S->flush();      /* (the stand-in copies with the host: the strip has to be there) */
      rc = shard_exchange(S);
    } else {
      const int held = S->flush_held();
      /* (held == 2: the window holds a mix of two frames.  The stream goes on -- a rank that left here would leave its peers blocked
         in their next collective -- and the error is returned at its end) */
      if (held == 2 && !S->err_held) { fprintf(stderr, "wr_shard: a frame wrote the window and flushed again before its end: not pipelinable\n"); S->err_held = 1; }
      rc = pending ? shard_exchange(S) : 0;
      pending = held;
      if (!held && rc == 0) rc = shard_exchange(S);
    }
    if (rc) { fprintf(stderr, "wr_shard: exchange failed (%d)\n", rc); return -2; }
  }
  if (pending) {
    S->flush();
    if (shard_exchange(S)) { fprintf(stderr, "wr_shard: exchange failed\n"); return -2; }
  }
  S->finish();                                      /* the backend's stream, collectives included */
  if (S->shm) {                                     /* every rank's strip of the last frame has landed */
    const uint64_t want = S->frames_done * (uint64_t)S->world;
    for (long spins = 0; __atomic_load_n((uint64_t*)S->shm, __ATOMIC_ACQUIRE) < want; spins++) {
      sched_yield();
      if (spins > 200000000L) { fprintf(stderr, "wr_shard: peers did not arrive\n"); return -3; }
    }
    /* ... in the receivers' windows, as the RCCL transport leaves them: the peers' rows of this rank's own window */
    if (S->mode == 1 || S->rank == 0)
      for (int p = 0; p < S->world; p++)
        if (p != S->rank && strip_len(S, p)) memcpy(S->fb + strip_off(S, p), S->shm + 4096 + strip_off(S, p), strip_len(S, p));
  }
  clock_gettime(CLOCK_MONOTONIC, &b);
  if (total_ms) *total_ms = (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
  if (S->err_held) { S->err_held = 0; return -6; }
  return 0;
}

/* shm transport: the assembled window (what rank 0's window holds on the GPU path) */
const uint8_t* wr_shard_shm_window(wr_shard* S) { return S->shm ? S->shm + 4096 : NULL; }

void wr_shard_close(wr_shard* S, const char* shm_name_to_unlink) {
  if (!S) return;
  if (S->comm && S->comm_destroy) S->comm_destroy(S->comm);
  if (S->shm) { munmap(S->shm, S->shm_size); if (shm_name_to_unlink) shm_unlink(shm_name_to_unlink); }
  free(S);
}
