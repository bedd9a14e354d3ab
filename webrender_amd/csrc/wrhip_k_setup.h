// wrhip_k_setup.h -- part of the gfx950 kernels of libwrhip: included by wrhip_kernels.h, in its order, and by nothing else.
// The setup stage: data-texture fetches, the shaders' vertex stages, swgl's draw_quad setup (wr_finish_prim), general-quad walks, side records, binning (wr_setup_body); the upload scatter.
#pragma once

#ifndef WR_DEVICE
#define WR_DEVICE __device__ __forceinline__
#endif

// ---------------------------------------------------------------------------
// small vector helpers
// hostsim only: which raster paths ran (printed at DestroyContext under WRHIP_DEBUG)
#ifdef WRHIP_HOSTSIM
static unsigned long long wr_dbg_paths[8];
#define WR_DBG_PATH(i) (wr_dbg_paths[i]++)
#else
#define WR_DBG_PATH(i) ((void)0)
#endif
#if defined(WRHIP_TIMING) && !defined(WRHIP_HOSTSIM)
// timing build: time points inside the setup stage of the STANDALONE setup kernel (wall_clock64: 10 ns), everything in flight waited for first
__device__ unsigned long long wr_dbg_tp[16384 * 8];
#define WR_TP(i) do { __builtin_amdgcn_s_waitcnt(0); const int g_ = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (g_ < 16384) wr_dbg_tp[g_ * 8 + (i)] = wall_clock64(); } while (0)
#else
#define WR_TP(i) ((void)0)
#endif
struct wf2 { float x, y; };
struct wf4 { float x, y, z, w; };
struct wi4 { int x, y, z, w; };

WR_DEVICE float wr_min(float a, float b) { return a < b ? a : b; }   // glsl.h:162
WR_DEVICE float wr_max(float a, float b) { return a > b ? a : b; }   // glsl.h:163
WR_DEVICE float wr_clamp(float a, float lo, float hi) { return wr_min(wr_max(a, lo), hi); }
WR_DEVICE int wr_imin(int a, int b) { return a < b ? a : b; }
WR_DEVICE int wr_imax(int a, int b) { return a > b ? a : b; }
WR_DEVICE int wr_iclamp(int a, int lo, int hi) { return wr_imin(wr_imax(a, lo), hi); }

// clampCoord, texture.h:70-72
WR_DEVICE int wr_clamp_coord(int c, int limit, int base = 0) { return wr_imin(wr_imax(c, base), limit - 1); }

// Data-texture fetches are written for the memory system, not as the reference's scalar code reads:
// the vertex stage is a chain of dependent reads (instance -> headers -> transform / task / GPU
// buffer rows) run by a handful of waves, so what it costs is round trips.  One 16-byte load per
// texel, no control flow around it (a null sampler reads a zero texel instead of branching), so
// the independent fetches of a stage are all in flight together.
#ifdef WRHIP_HOSTSIM
static const uint32_t wr_zero_texel[4] = {0, 0, 0, 0};
#else
__device__ const uint32_t wr_zero_texel[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
#endif
struct wr_u4 { uint32_t x, y, z, w; };
WR_DEVICE wr_u4 wr_load16(const void* p) {
  wr_u4 r;
#ifdef WRHIP_HOSTSIM
  __builtin_memcpy(&r, p, 16);
#else
  // (an explicit GLOBAL load: a pointer read out of a descriptor is a generic one to the compiler, and a flat load counts against
  // both wait counters -- every use then drains everything in flight, one fetch at a time)
  typedef uint32_t wr_gu4 __attribute__((ext_vector_type(4)));
  const wr_gu4 v = *(const __attribute__((address_space(1))) wr_gu4*)p;       // texel rows are 16-byte aligned (pool storage, 1024-texel rows)
  r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
#endif
  return r;
}
WR_DEVICE uint32_t wr_load4(const void* p) {
#ifdef WRHIP_HOSTSIM
  uint32_t r; __builtin_memcpy(&r, p, 4); return r;
#else
  return *(const __attribute__((address_space(1))) uint32_t*)p;
#endif
}
WR_DEVICE float wr_bits_f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
WR_DEVICE uint32_t wr_float_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
WR_DEVICE void wr_store16(float* p, float a, float b, float c, float d) {      // p: 16-byte aligned
#ifdef WRHIP_HOSTSIM
  p[0] = a; p[1] = b; p[2] = c; p[3] = d;
#else
  typedef float wr_f4 __attribute__((ext_vector_type(4)));
  *(wr_f4*)p = wr_f4{a, b, c, d};
#endif
}

// texelFetch(sampler2D RGBA32F, ivec2_scalar, 0), texture.h:283-292
WR_DEVICE wf4 wr_fetch_f(const WrTexDesc& t, int x, int y) {
  const void* base = t.ptr;
  const int w = t.width, h = t.height, stride = t.stride, fmt = t.format;
  x = wr_clamp_coord(x, w);
  y = wr_clamp_coord(y, h);
  // null_sampler: 1x1 transparent black (gl.cc:901-912)
  // ONE 16-byte load whatever the format, and NO control flow: the RGBA8 reading of the sampler (pixel_to_vec4, texture.h:101-105)
  // takes its texel out of the aligned 16 bytes around it (storage starts on 16 bytes and is a multiple of 16 long), a null sampler
  // reads the zero texel.  The offset is worked out whether or not the sampler is live and the RGBA8 conversion whether or not it
  // is used (the empty asm keeps the compiler from sinking it behind a branch again): with a conditional load, or a conditional
  // conversion, every fetch ends in a block of its own and waits for its load there -- one round trip per texel (ps_quad: 16 in a row,
  // 7 us of the setup stage's 10 us vertex stage, profiles/r06_i_setup_phases.txt) instead of one per group of independent fetches.
  const bool live = base != nullptr;
  const bool rgba8 = live && fmt != WR_FMT_RGBA32F;
  const long long off = (long long)x * (rgba8 ? 4 : 16) + (long long)y * (long long)stride * 4;
  const uintptr_t a = (live ? (uintptr_t)base : (uintptr_t)wr_zero_texel) + (uintptr_t)(live ? off : 0ll);
  const wr_u4 v = wr_load16((const void*)(a & ~(uintptr_t)15));
  const unsigned sel = (unsigned)(a >> 2) & 3u;
  const uint32_t qlo = (sel & 1u) ? v.y : v.x, qhi = (sel & 1u) ? v.w : v.z;
  const uint32_t q = (sel & 2u) ? qhi : qlo;
  wf4 r8 = {float((q >> 16) & 0xFF) * (1.0f / 255.0f), float((q >> 8) & 0xFF) * (1.0f / 255.0f), float(q & 0xFF) * (1.0f / 255.0f), float(q >> 24) * (1.0f / 255.0f)};
#ifndef WRHIP_HOSTSIM
  asm("" : "+v"(r8.x), "+v"(r8.y), "+v"(r8.z), "+v"(r8.w));
#endif
  wf4 r;
  r.x = rgba8 ? r8.x : wr_bits_f(v.x);
  r.y = rgba8 ? r8.y : wr_bits_f(v.y);
  r.z = rgba8 ? r8.z : wr_bits_f(v.z);
  r.w = rgba8 ? r8.w : wr_bits_f(v.w);
  return r;
}

WR_DEVICE wi4 wr_fetch_i(const WrTexDesc& t, int x, int y) {  // texture.h:338-343
  const void* base = t.ptr;
  const int w = t.width, h = t.height, stride = t.stride;
  x = wr_clamp_coord(x, w);
  y = wr_clamp_coord(y, h);
  const bool live = base != nullptr;
  const long long off = (long long)x * 16 + (long long)y * (long long)stride * 4;      // (unconditional: see wr_fetch_f)
  const void* p = (const void*)((live ? (uintptr_t)base : (uintptr_t)wr_zero_texel) + (uintptr_t)(live ? off : 0ll));
  const wr_u4 v = wr_load16(p);
  wi4 r = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  return r;
}

// get_fetch_uv, shared.glsl:77 ; get_gpu_cache_uv gpu_cache.glsl:16
WR_DEVICE void wr_fetch_uv(int i, unsigned vpi, int& u, int& v) {
  u = int(vpi * (unsigned(i) % (1024u / vpi)));
  v = int(unsigned(i) / (1024u / vpi));
}

struct WrMat4 { wf4 c[4]; };

// mat4_scalar * vec4, glsl.h:2582-2598 (left-to-right sums, no contraction)
WR_DEVICE wf4 wr_mul(const WrMat4& m, wf4 v) {
  wf4 u;
  u.x = m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z + m.c[3].x * v.w;
  u.y = m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z + m.c[3].y * v.w;
  u.z = m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z + m.c[3].z * v.w;
  u.w = m.c[0].w * v.x + m.c[1].w * v.y + m.c[2].w * v.z + m.c[3].w * v.w;
  return u;
}

struct WrTransform { WrMat4 m, inv_m; bool axis_aligned; };

// fetch_transform, transform.glsl:22-46
WR_DEVICE WrTransform wr_fetch_transform(const WrDrawDesc& d, int id) {
  WrTransform t;
  t.axis_aligned = (id >> 23) == 0;
  int index = id & 0x007fffff;
  int u, v;
  wr_fetch_uv(index, 8u, u, v);
  for (int k = 0; k < 4; k++) {
    t.m.c[k] = wr_fetch_f(d.tex[WR_S_TRANSFORMS], u + k, v);
    t.inv_m.c[k] = wr_fetch_f(d.tex[WR_S_TRANSFORMS], u + 4 + k, v);
  }
  return t;
}

struct WrTask { wf2 p0, p1; float dps; wf2 origin; };

// fetch_render_task_data / fetch_picture_task, render_task.glsl:17-79
WR_DEVICE WrTask wr_fetch_task(const WrDrawDesc& d, int address) {
  int u, v;
  wr_fetch_uv(address, 2u, u, v);
  wf4 t0 = wr_fetch_f(d.tex[WR_S_RENDER_TASKS], u, v);
  wf4 t1 = wr_fetch_f(d.tex[WR_S_RENDER_TASKS], u + 1, v);
  WrTask t;
  t.p0 = {t0.x, t0.y}; t.p1 = {t0.z, t0.w};
  t.dps = t1.x; t.origin = {t1.y, t1.z};
  return t;
}

// instance attribute loads (load_flat_attrib, gl.cc:1044-1062): copies
// min(sizeof(T), va.size) bytes, the rest is zero.
template <typename T>
WR_DEVICE T wr_load_attr(const WrDrawDesc& d, const uint8_t* arena, int instance, int k) {
  const int n = int(sizeof(T) / 4);
  uint32_t w[4] = {0, 0, 0, 0};
  if (d.attr_off[k] >= 0) {
    const uint8_t* src = arena + d.inst_offset + (size_t)d.inst_stride * instance + d.attr_off[k];
    if ((d.attr_u16 >> k) & 1u) {            // integer attribute stored as u16 components (load_flat_attrib converts)
      const int comps = d.attr_bytes[k] / 2;
      for (int i = 0; i < n; i++)
        if (i < comps) { uint16_t h; __builtin_memcpy(&h, src + 2 * i, 2); w[i] = h; }
    } else {
      int words = d.attr_bytes[k] / 4;
      for (int i = 0; i < n; i++)
        if (i < words) __builtin_memcpy(&w[i], src + 4 * i, 4);
    }
  }
  T out;
  __builtin_memcpy(&out, w, sizeof(T));
  return out;
}

// s0 + step + step + ... (c adds, each rounded to fp32), as swgl's span loops
// accumulate quantised UVs (`uv += uv_step`, swgl_ext.h:176).  Closed form when
// provably identical: if s0 and step are both multiples of 2^g and every partial
// sum is below 2^(g+24) in magnitude, every add is exact, so the result is the
// real number s0 + c*step (partial sums are monotone between the end points).
WR_DEVICE int wr_low_bit_exp(float x) {
  uint32_t b; __builtin_memcpy(&b, &x, 4);
  uint32_t e = (b >> 23) & 0xFF, m = b & 0x7FFFFF;
  if (e == 0) return m ? -1000 : 1000;         // denormal: force the loop; zero: no constraint
  m |= 0x800000;
  return int(e) - 150 + __builtin_ctz(m);
}
WR_DEVICE bool wr_accum_is_linear(float s0, float step, int c) {
  if (c <= 0 || step == 0.0f) return true;
  const int g0 = wr_low_bit_exp(s0), g1 = wr_low_bit_exp(step);
  const int g = g0 < g1 ? g0 : g1;
  const double end = double(s0) + double(c) * double(step);
  const double a0 = s0 < 0 ? -double(s0) : double(s0), a1 = end < 0 ? -end : end;
  const double bound = a0 > a1 ? a0 : a1;
  return g > -900 && g < 100 && bound < __builtin_ldexp(1.0, g + 24);
}
// The general case, binade by binade.  While the running sum stays inside one binade
// [2^e, 2^(e+1)) its ulp u is fixed, the sum is a multiple of u, and fl(s + step) = s + q*u with
// q = round(step / u) -- the same q every time unless step / u sits exactly on a rounding tie.  So
// all the adds that stay inside the binade are one exact fp64 multiply-add; only the adds that
// cross a binade boundary (and ties, zero / denormal sums) are executed one by one.  Verified
// against the plain loop on random, dyadic, tie-prone and binade-floor inputs (tests/test_accum.py);
// a 1840-row box-shadow mask spent ~900 dependent adds per interpolant and row in that loop.
// All in 32-bit integers, in units of u: S = the sum's 24-bit significand, q = round-half-even of the
// step's significand shifted to the sum's exponent.  (An fp64 version of the same walk cost every
// kernel that can reach the general pixel path ~20 VGPRs: with interprocedural register allocation a
// caller keeps its live values above whatever its callees clobber -- and the glyph kernel sits
// exactly at the 168-VGPR / 3-waves-per-SIMD step.)
// (out of line: 32 call sites reach it through wr_accum and it is the rare case -- inlined, its ~0.5 KB copies sat between the
// instructions that do run, and a latency-bound launch walks its code with a cold instruction cache)
__device__ __noinline__ float wr_accum_binades(float s0, float step, int c) {
  float s = s0;
  int k = c;
  uint32_t db; __builtin_memcpy(&db, &step, 4);
  const int ed = int((db >> 23) & 0xFF);
  const uint32_t md = (db & 0x7FFFFFu) | 0x800000u;
  while (k > 0) {
    uint32_t b; __builtin_memcpy(&b, &s, 4);
    const int ex = int((b >> 23) & 0xFF);
    const int sh = ex - ed;                      // step / u = md / 2^sh
    // zero / denormal / inf / nan operands, or a step that dwarfs the sum (binades fly by): plain step
    if (ex == 0 || ex == 0xFF || ed == 0 || ed == 0xFF || sh < 0) { s = s + step; k--; continue; }
    uint32_t q;
    bool tie = false;
    if (sh == 0) q = md;
    else if (sh > 25) q = 0;                     // step < u/4
    else {
      const uint32_t half = 1u << (sh - 1), rem = md & ((1u << sh) - 1u);
      q = md >> sh;
      if (rem > half) q++;
      else if (rem == half) tie = true;          // the parity of s decides: step singly
    }
    if (tie) { s = s + step; k--; continue; }
    const bool up = ((b ^ db) >> 31) == 0;       // same sign: |s| grows
    const uint32_t S = (b & 0x7FFFFFu) | 0x800000u;
    if (q == 0) {
      // |step| < u/2: s + step rounds back to s -- unless s sits on the binade floor and moves down (finer grid below)
      if (S == 0x800000u && !up) { s = s + step; k--; continue; }
      return s;
    }
    // adds that keep every exact sum inside the binade (conservative by >= 2 q: the true step is within q/2.. of q*u)
    const uint32_t dist = up ? (0x1000000u - S) : (S - 0x800000u);
    const int n = int(float(dist) / float(q)) - 3;
    if (n >= 1) {
      const int steps = n > k ? k : n;
      const uint32_t S2 = up ? S + uint32_t(steps) * q : S - uint32_t(steps) * q;
      const uint32_t nb = (b & 0x80000000u) | (uint32_t(ex) << 23) | (S2 & 0x7FFFFFu);
      __builtin_memcpy(&s, &nb, 4);
      k -= steps;
    }
    // the few adds left up to (and across) the binade boundary: plain steps, no new analysis
#pragma unroll
    for (int i = 0; i < 5; i++) if (k > 0) { s = s + step; k--; }
  }
  return s;
}
// wr_accum's closed form, when it provably equals the sequential sum (see wr_accum): true and `out` set
WR_DEVICE bool wr_accum_closed(float s0, float step, int c, float& out) {
  // (straight-line on purpose: with early returns -- a wave-uniform `c >= 2^24` test inside the divergent `step != 0` region -- the
  // build's -structurizecfg-skip-uniform-regions lost the "step == 0: closed" lanes' flag and sent them through the walk, one plain
  // add per row: 600 k cycles for row 1838 of cfg4's shadow, profiles/r06_e_acc_walk.txt)
  const bool trivial = c <= 0 || step == 0.0f;
  const int g0 = wr_low_bit_exp(s0), g1 = wr_low_bit_exp(step);
  const int g = g0 < g1 ? g0 : g1;
  const float end = fmaf(float(c), step, s0);
  const float a0 = fabsf(s0), a1 = fabsf(end);
  const float bound = a0 > a1 ? a0 : a1;
  uint32_t bb; __builtin_memcpy(&bb, &bound, 4);
  const int be = int((bb >> 23) & 0xFF) - 127;
  const bool ok = (c < (1 << 24)) & (g > -900) & (g < 100) & (be + 1 <= g + 24) & (be < 127);
  out = trivial ? s0 : end;
  return trivial | ok;
}
WR_DEVICE float wr_accum(float s0, float step, int c) {
  // closed form when provably identical: s0 and step are multiples of 2^g and every partial sum stays
  // below 2^(g+24), so every add is exact and the result is the real number s0 + c*step -- which one
  // fused multiply-add delivers (single rounding of an exactly representable value).  fp32 / integer only.
  float r;
  if (wr_accum_closed(s0, step, c, r)) return r;
  WR_DBG_PATH(2);
  return wr_accum_binades(s0, step, c);
}

// The same sum for short spans (glyph-sized prims): closed form when provable, else the plain loop.
// Used on the glyph blit path, whose kernel sits at an occupancy step and cannot afford the fp64
// temporaries of wr_accum_binades in its inline code (168 -> 188 VGPRs = 3 -> 2 waves per SIMD).
WR_DEVICE float wr_accum_short(float s0, float step, int c) {
  if (c <= 0 || step == 0.0f) return s0;
  const int g0 = wr_low_bit_exp(s0), g1 = wr_low_bit_exp(step);
  const int g = g0 < g1 ? g0 : g1;
  const double end = double(s0) + double(c) * double(step);
  const double a0 = s0 < 0 ? -double(s0) : double(s0), a1 = end < 0 ? -end : end;
  const double bound = a0 > a1 ? a0 : a1;
  if (g > -900 && g < 100 && bound < __builtin_ldexp(1.0, g + 24)) return float(end);
  float s = s0;
  for (int i = 0; i < c; i++) s += step;
  return s;
}

// row-k edge interpolant: closed form when the prim was verified linear over all its rows
WR_DEVICE float wr_row_interp(float s0, float step, int k, bool linear) {
  if (linear) return float(double(s0) + double(k) * double(step));
  return wr_accum(s0, step, k);
}

// ---- row-sum tables (WrAccTab, wrhip_types.h) ----
// The pieces of k -> s0 + step + .. + step (k adds) for k in [0, kmax]: the walk of wr_accum_binades, recorded.  A piece ends where
// the plain add leaves the line its analysis predicted (a binade boundary, a rounding tie, zero / denormal sums: those rows are
// pieces of their own), so the table is exact by construction wherever the bulk step is (tests/test_accum.py holds both to the
// plain loop).  Returns the number of pieces, 0 when they do not fit.
WR_DEVICE int wr_acctab_build(float s0, float step, int kmax, WrAccTab* T) {
  float s = s0;
  int k = 0, n = 0;
  uint32_t db; __builtin_memcpy(&db, &step, 4);
  const int ed = int((db >> 23) & 0xFF);
  const uint32_t md = (db & 0x7FFFFFu) | 0x800000u;
  while (k <= kmax) {
    if (n == WR_ACCTAB_N) return 0;
    uint32_t b; __builtin_memcpy(&b, &s, 4);
    const int ex = int((b >> 23) & 0xFF);
    const int sh = ex - ed;
    bool single = ex == 0 || ex == 0xFF || ed == 0 || ed == 0xFF || sh < 0;
    uint32_t q = 0;
    if (!single) {
      if (sh == 0) q = md;
      else if (sh > 25) q = 0;
      else {
        const uint32_t half = 1u << (sh - 1), rem = md & ((1u << sh) - 1u);
        q = md >> sh;
        if (rem > half) q++;
        else if (rem == half) single = true;
      }
    }
    const bool up = ((b ^ db) >> 31) == 0;
    const uint32_t S = (b & 0x7FFFFFu) | 0x800000u;
    if (!single && q == 0) {
      if (S == 0x800000u && !up) single = true;       // the binade floor, moving down: a finer grid below
      else { T->k[n] = k; T->s[n] = b; T->q[n] = 0; return n + 1; }      // the sum no longer moves
    }
    T->k[n] = k; T->s[n] = b; T->q[n] = single ? 0 : (up ? int32_t(q) : -int32_t(q));
    n++;
    if (single) { s = s + step; k++; continue; }
    const uint32_t dist = up ? (0x1000000u - S) : (S - 0x800000u);
    const int nb = int(float(dist) / float(q)) - 3;
    if (nb >= 1 && k < kmax) {
      const int steps = nb > kmax - k ? kmax - k : nb;
      const uint32_t S2 = up ? S + uint32_t(steps) * q : S - uint32_t(steps) * q;
      const uint32_t nbits = (b & 0xFF800000u) | (S2 & 0x7FFFFFu);
      __builtin_memcpy(&s, &nbits, 4);
      k += steps;
    }
    // plain adds for as long as they stay on this piece's line
    for (;;) {
      if (k >= kmax) return n;
      uint32_t bc; __builtin_memcpy(&bc, &s, 4);
      const float s2 = s + step;
      uint32_t b2; __builtin_memcpy(&b2, &s2, 4);
      const uint32_t Sc = (bc & 0x7FFFFFu) | 0x800000u, Sn = (b2 & 0x7FFFFFu) | 0x800000u;
      const bool on_line = (b2 >> 23) == (bc >> 23) && Sn == (up ? Sc + q : Sc - q);
      s = s2; k++;
      if (!on_line) break;
    }
  }
  return n;
}
// (every k[] is fetched, four to a 16-byte load, whatever n is -- one round trip for the scan instead of a dependent load per entry;
// the pieces are in row order, so the piece of row k is the number of valid entries at or below k, less one)
WR_DEVICE float wr_acctab_eval(const WrAccTab* T, int n, int k) {
  int cnt = 0;
#pragma unroll
  for (int g = 0; g < WR_ACCTAB_N / 4; g++) {
    const wr_u4 v = wr_load16(&T->k[4 * g]);
    cnt += ((4 * g + 0 < n) & (int(v.x) <= k)) + ((4 * g + 1 < n) & (int(v.y) <= k)) + ((4 * g + 2 < n) & (int(v.z) <= k)) + ((4 * g + 3 < n) & (int(v.w) <= k));
  }
  const int j = cnt > 0 ? cnt - 1 : 0;
  const uint32_t b = T->s[j];
  const int32_t q = T->q[j];
  const int32_t kj = T->k[j];
  if (q == 0) return wr_bits_f(b);
  const uint32_t S = ((b & 0x7FFFFFu) | 0x800000u) + uint32_t((k - kj) * q);
  return wr_bits_f((b & 0xFF800000u) | (S & 0x7FFFFFu));
}
// sum i of a mask prim at row k, from the prim's tables (mode 4 -- the sum of the other edge -- is the caller's business)
WR_DEVICE float wr_acctabs_row(const WrAccTabs* T, int i, int k) {
  // (header and k[] in one round trip: nothing below depends on a load before all of them are out)
  const float s0 = T->s0[i], st = T->st[i];
  const int n = T->n[i], mode = T->mode[i];
  const WrAccTab* tb = &T->tab[i];
  int cnt = 0;
#pragma unroll
  for (int g = 0; g < WR_ACCTAB_N / 4; g++) {
    const wr_u4 v = wr_load16(&tb->k[4 * g]);
    cnt += ((4 * g + 0 < n) & (int(v.x) <= k)) + ((4 * g + 1 < n) & (int(v.y) <= k)) + ((4 * g + 2 < n) & (int(v.z) <= k)) + ((4 * g + 3 < n) & (int(v.w) <= k));
  }
  if (mode == 0) return (k <= 0 || st == 0.0f) ? s0 : fmaf(float(k), st, s0);
  if (mode == 1) return float(double(s0) + double(k) * double(st));
  if (mode == 2) {
    const int j = cnt > 0 ? cnt - 1 : 0;
    const uint32_t b = tb->s[j];
    const int32_t q = tb->q[j], kj = tb->k[j];
    if (q == 0) return wr_bits_f(b);
    const uint32_t S = ((b & 0x7FFFFFu) | 0x800000u) + uint32_t((k - kj) * q);
    return wr_bits_f((b & 0xFF800000u) | (S & 0x7FFFFFu));
  }
  WR_DBG_PATH(2);
  return wr_accum(s0, st, k);
}
// the same for one thread that wants sum i whatever its mode (the setup stage's middle-row key, the host simulation's rows)
WR_DEVICE float wr_acctabs_row_any(const WrAccTabs* T, int i, int k) {
  return wr_acctabs_row(T, T->mode[i] == 4 ? i - 2 : i, k);
}
// the setup stage's side: the tables of a prim's `nsums` (4 or 8) interpolants over rows [0, kmax].  Sums come in left / right pairs
// (i, i + 2) that are equal on axis-aligned prims: the right one is then marked as such.  (The arrays are only ever indexed by
// constants -- a select chain picks sum i -- so they stay in registers: no scratch in the kernels that carry the setup stage.)
WR_DEVICE float wr_pick8(const float (&a)[8], int i) {
  return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : i == 3 ? a[3] : i == 4 ? a[4] : i == 5 ? a[5] : i == 6 ? a[6] : a[7];
}
WR_DEVICE void wr_acctabs_build(WrAccTabs* T, int nsums, const float (&s0)[8], const float (&st)[8], int kmax, bool uv_linear) {
  for (int p = 0; p < 4; p++) {
    const int iL = (p & 1) + 4 * (p >> 1), iR = iL + 2;
    const float aL = wr_pick8(s0, iL), dL = wr_pick8(st, iL), aR = wr_pick8(s0, iR), dR = wr_pick8(st, iR);
    T->s0[iL] = aL; T->st[iL] = dL; T->s0[iR] = aR; T->st[iR] = dR;
    T->n[iL] = 0; T->n[iR] = 0;
    int mL = 0, mR = 0;
    if (iL < nsums) {
      float r;
      if (uv_linear && iL < 4) mL = mR = 1;
      else {
        if (!wr_accum_closed(aL, dL, kmax, r)) {          // (closed at the last row: closed at every row before it)
          const int nL = wr_acctab_build(aL, dL, kmax, &T->tab[iL]);
          T->n[iL] = nL; mL = nL > 0 ? 2 : 3;
        }
        if (wr_float_bits(aL) == wr_float_bits(aR) && wr_float_bits(dL) == wr_float_bits(dR)) mR = 4;
        else if (!wr_accum_closed(aR, dR, kmax, r)) {
          const int nR = wr_acctab_build(aR, dR, kmax, &T->tab[iR]);
          T->n[iR] = nR; mR = nR > 0 ? 2 : 3;
        }
      }
    }
    T->mode[iL] = mL; T->mode[iR] = mR;
  }
}

// round_pixel (portable path): cast(v * 255 + 0.5), glsl.h:732-744
WR_DEVICE int wr_round_pixel(float v) { return int(v * 255.0f + 0.5f); }

// pack_pixels_RGBA8(vec4_scalar) -> WideRGBA8 lanes (b,g,r,a) as u16, then the
// value is what later gets `pack`ed; we keep the u16 quadruple (blend.h:41-45,
// packRGBA8 portable: CONVERT i32 -> u16 wraps).
WR_DEVICE void wr_pack_color(wf4 c, uint32_t out[2]) {
  uint32_t b = uint32_t(wr_round_pixel(c.z)) & 0xFFFF;
  uint32_t g = uint32_t(wr_round_pixel(c.y)) & 0xFFFF;
  uint32_t r = uint32_t(wr_round_pixel(c.x)) & 0xFFFF;
  uint32_t a = uint32_t(wr_round_pixel(c.w)) & 0xFFFF;
  out[0] = b | (g << 16);
  out[1] = r | (a << 16);
}

// ---------------------------------------------------------------------------
// Vertex stage outputs before draw_quad
// GradientStops::can_merge (swgl_ext.h:1318-1326) of every pair of neighbouring entries of a validated 130-entry table, as a bitmap
// (WrGradRec::merge).  Eight entries' steps are requested per round trip: the setup stage is one dependent chain per wave.
// (eight entries per trip: with sixteen in flight every kernel that carries the setup stage spilled 1.8 KB more per lane; NOT out of
// line -- a callee is compiled to its own register budget and the kernels that call it inherit its count: 128 -> 356 VGPRs)
// The gradient table of one prim, copied into the flush's pool the way main() fetches it (gradient.glsl:42-61: entry i = texels
// (u, v) and (u + 1, v) of address + 2 i) -- with the copy the raster stage does not read sGpuBufferF at all, so the next frame's
// upload of that texture need not wait for this frame's tile pass (the held-back launches keep carrying the next setup stage:
// wrench aligned- / unaligned-gradient).  One entry per trip: the setup stage's register budget (see wr_grad_merge_bits).
WR_DEVICE void wr_grad_copy_table(const WrDrawDesc& d, const WrTargetDesc& T, int inst, WrGradRec* G) {
  G->table = nullptr;
  if (!(d.flags & WR_DF_GTAB) || d.gtab_base < 0 || !T.qtab) return;
  float* dst = T.qtab + (size_t)d.gtab_base + (size_t)inst * WR_GTAB_WORDS;
  const WrTexDesc& gb = d.tex[WR_S_GPU_BUFFER_F];
  const int address = G->address;
#pragma nounroll
  for (int i = 0; i < 130; i++) {
    const int addr = address + 2 * i;
    const int u = int(unsigned(addr) % 1024u), v = int(unsigned(addr) / 1024u);
    const wf4 t0 = wr_fetch_f(gb, u, v), t1 = wr_fetch_f(gb, u + 1, v);
    wr_store16(dst + 8 * i, t0.x, t0.y, t0.z, t0.w);
    wr_store16(dst + 8 * i + 4, t1.x, t1.y, t1.z, t1.w);
  }
  G->table = dst;
  if (G->stops) G->stops = dst;
}
// The descriptors a setup wave reads go through this view.  A wave holds one 64-prim block, and a block lies in one or two draws almost
// always (the host starts targets on block boundaries; its block table names the block's first draw and the targets: WrBlock).
// WR_DESC_STAGE (off): the wave copies the block's first two draw descriptors and their target descriptors into LDS with two
// coalesced trips and reads them there (generic pointers: flat loads).  MEASURED (cfg2, profiles/r06_m_setup_points.txt): the
// descriptor reads were not what the setup stage waits for -- the compiler had already gathered them at the head of each stage;
// 0.5 us of 16 per wave -- and a flat load drains both wait counters; and the build differs from the oracle on tests/abi_surface.py
// (not understood).  Not adopted; the view and the block table stay (they cost nothing), the staging is compiled out.
struct WrBlock { int32_t lo0, nk, t0, t1; };      // first draw of the block, draws staged (<= 2), their targets
#define WR_SD_DRAW_Q  ((int)(sizeof(WrDrawDesc) / 8))
#define WR_SD_TGT_Q   ((int)(sizeof(WrTargetDesc) / 8))
#define WR_SD_WAVE_Q  (2 * (WR_SD_DRAW_Q + WR_SD_TGT_Q))      // 8-byte words of LDS per wave
struct WrDescView {
  const WrDrawDesc* draws; const WrTargetDesc* targets;
  const WrDrawDesc* ld; const WrTargetDesc* lt;      // the LDS copies (nk == 0: none)
  int lo0, nk, t0, t1;
  WR_DEVICE const WrDrawDesc& draw(int i) const { const unsigned k = unsigned(i - lo0); return k < unsigned(nk) ? ld[k] : draws[i]; }
  WR_DEVICE const WrTargetDesc& target(int t) const { return (nk > 0 && t == t0) ? lt[0] : (nk > 1 && t == t1) ? lt[1] : targets[t]; }
};
// ... for the gradient prims of one wave of the setup stage, `G` = this lane's record or nullptr.  On the device the wave shares out
// the 130 entries of every table (one thread copying 130 entries fetches them one after the other: 45 us on top of a 35 us setup
// stage, measured); the host simulation copies serially.
WR_DEVICE void wr_grad_tables_wave(const WrDescView& V, int draw, int gid, WrGradRec* G) {
#ifdef WRHIP_HOSTSIM
  if (G) { const WrDrawDesc& d = V.draw(draw); wr_grad_copy_table(d, V.target(d.target), gid - d.first_prim, G); }
#else
  const int lane = threadIdx.x & 63;
  bool want = false;
  unsigned long long dsta = 0ull;
  int address = 0;
  if (G) {
    G->table = nullptr;
    const WrDrawDesc& d = V.draw(draw);
    const WrTargetDesc& T = V.target(d.target);
    if ((d.flags & WR_DF_GTAB) && d.gtab_base >= 0 && T.qtab) {
      want = true;
      dsta = (unsigned long long)(uintptr_t)(T.qtab + (size_t)d.gtab_base + (size_t)(gid - d.first_prim) * WR_GTAB_WORDS);
      address = G->address;
    }
  }
  for (unsigned long long m = __ballot(want); m; m &= m - 1ull) {
    const int src = __builtin_ctzll(m);
    const int sdraw = __builtin_amdgcn_readlane(draw, src), saddr = __builtin_amdgcn_readlane(address, src);
    const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dsta, src), phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dsta >> 32), src);
    float* dst = (float*)(uintptr_t)((unsigned long long)plo | ((unsigned long long)phi << 32));
    const WrTexDesc& gb = V.draws[sdraw].tex[WR_S_GPU_BUFFER_F];
    for (int i = lane; i < 130; i += 64) {
      const int addr = saddr + 2 * i;
      const int u = int(unsigned(addr) % 1024u), v = int(unsigned(addr) / 1024u);
      const wf4 t0 = wr_fetch_f(gb, u, v), t1 = wr_fetch_f(gb, u + 1, v);
      wr_store16(dst + 8 * i, t0.x, t0.y, t0.z, t0.w);
      wr_store16(dst + 8 * i + 4, t1.x, t1.y, t1.z, t1.w);
    }
  }
  // (the owners read their tables next -- wr_grad_merge_bits -- : other lanes' stores first)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (want) {
    float* dst = (float*)(uintptr_t)dsta;
    G->table = dst;
    if (G->stops) G->stops = dst;
  }
#endif
}
WR_DEVICE void wr_grad_merge_bits(WrGradRec* G) {
  const float* stops = G->stops;
  uint32_t m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u, m4 = 0u;      // (in registers: the record lives in HBM)
  if (stops) {
#pragma nounroll      /* (unrolled, its 17 x 9 sixteen-byte loads in flight took the setup kernel from 166 to 474 VGPRs) */
    for (int base = 0; base < 129; base += 8) {
      wr_u4 st[9];
#pragma unroll
      for (int j = 0; j < 9; j++) st[j] = wr_load16(stops + 8 * wr_imin(base + j, 129) + 4);
      uint32_t bits = 0u;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool m = base + j < 129 && wr_bits_f(st[j].x) == wr_bits_f(st[j + 1].x) && wr_bits_f(st[j].y) == wr_bits_f(st[j + 1].y) &&
                       wr_bits_f(st[j].z) == wr_bits_f(st[j + 1].z) && wr_bits_f(st[j].w) == wr_bits_f(st[j + 1].w);
        bits |= m ? (1u << j) : 0u;
      }
      const uint32_t sh = bits << (base & 31);           // base is a multiple of 8: a batch never straddles a word
      const int w = base >> 5;
      m0 |= w == 0 ? sh : 0u; m1 |= w == 1 ? sh : 0u; m2 |= w == 2 ? sh : 0u; m3 |= w == 3 ? sh : 0u; m4 |= w == 4 ? sh : 0u;
    }
  }
  G->merge[0] = m0; G->merge[1] = m1; G->merge[2] = m2; G->merge[3] = m3; G->merge[4] = m4;
}
// first i' >= i whose merge bit is clear / last i' < i whose merge bit is clear (-1: none); bits 129 .. 159 are clear
WR_DEVICE int wr_merge_clear_from(const WrGradRec& G, int i) {
  for (int w = i >> 5; w < 5; w++) {
    const uint32_t v = ~G.merge[w] & (w == (i >> 5) ? (0xFFFFFFFFu << (i & 31)) : 0xFFFFFFFFu);
    if (v) return 32 * w + __builtin_ctz(v);
  }
  return 160;
}
WR_DEVICE int wr_merge_clear_below(const WrGradRec& G, int i) {
  if (i <= 0) return -1;
  for (int w = (i - 1) >> 5; w >= 0; w--) {
    const int top = w == ((i - 1) >> 5) ? ((i - 1) & 31) : 31;
    const uint32_t v = ~G.merge[w] & (top == 31 ? 0xFFFFFFFFu : ((1u << (top + 1)) - 1u));
    if (v) return 32 * w + 31 - __builtin_clz(v);
  }
  return -1;
}
struct WrVsOut {
  float px[4], py[4], pz[4], pw[4];  // gl_Position per SIMD lane (lane order 0,1,3,2)
  int kind;            // WrPrimKind
  wf4 color;           // flat colour (solid) / modulation colour (textured)
  int has_color;       // textured: colour != NoColor
  float u[4], v[4];    // varying uv per lane
  wf4 uv_bounds;
  int tex_slot;
  int aa_edges;        // swgl_antiAlias mask
  int has_mask;        // swgl_clipMask set
  float mask_offset[2], mask_bb[4];   // swgl_clipMask(offset, bb_origin, bb_size) arguments
  float uv_add[2];     // shader adds this to the interpolated uv before sampling (0 unless set)
  float u2[4], v2[4];  // a second interpolated vec2 varying (WR_PK_BOX_SHADOW: vLocalPos.xy)
  float u3[4], v3[4];  // a third one (WR_PK_YUV: vUv_V)
  int tail_clamp;      // fragment main(): clamps uv to uv_bounds
  int tail_modulate;   // fragment main(): multiplies texel by colour
  int blend_override;  // swgl_blendDropShadow / swgl_blendSubpixelText: WrBlend key replacing the draw's (0 = none)
  wf4 blend_color;     // ... and its constant colour (swgl_BlendColorRGBA8)
  int dual; float dual_swz;   // brush_image DUAL_SOURCE_BLENDING (WrPrim::dual / dual_swz)
  int cd[4];           // gl_ClipDistance of ps_text_run GLYPH_TRANSFORM as a box in target pixels (x0, y0, x1, y1; x1 < x0: none)
  float persp_div;     // brush_image: perspective_interpolate of `uv = v_uv * mix(gl_FragCoord.w, 1.0, .)` in main(); < 0: no such factor
};

// get_rgb_from_ycbcr_info (yuv.glsl:96-165) in strict fp32 in the shader's order + vRescaleFactor, and the coefficients swgl's
// YUVMatrix derives from them (composite.h:652-719): shared by brush_yuv_image and composite ... YUV
WR_DEVICE void wr_yuv_setup(WrYuvRec& Y, int depth, int color_space, int format) {
  WrYuvRec* Yv = &Y;
    Yv->format = format;
    Yv->rescale = (depth > 8 && format != 1) ? 16 - depth : 0;
    float channel_max = 255.0f;
    if (depth > 8) channel_max = format == 1 ? float((1 << depth) - 1) : 65535.0f;
    // yuv_channel_zero_one_{narrow_range, full_range, identity}
    const int sh = depth - 8;
    const float n0 = float(16 << sh) / channel_max, n1 = float(128 << sh) / channel_max, n2 = float(235 << sh) / channel_max, n3 = float(240 << sh) / channel_max;
    const float ones = float((1 << depth) - 1) / channel_max;
    float z0, z1, o0, o1;
    if (color_space == 0 || color_space == 2 || color_space == 4) { z0 = n0; z1 = n1; o0 = n2; o1 = n3; }
    else if (color_space == 1 || color_space == 3 || color_space == 5) { z0 = 0.0f; z1 = n1; o0 = ones; o1 = ones; }
    else { z0 = 0.0f; z1 = 0.0f; o0 = ones; o1 = ones; }
    // RgbFromYuv_* (column-major)
    float A[9];
    if (color_space <= 1) { const float m[9] = {1.00000f, 1.00000f, 1.00000f, 0.00000f, -0.17207f, 0.88600f, 0.70100f, -0.35707f, 0.00000f}; for (int i = 0; i < 9; i++) A[i] = m[i]; }
    else if (color_space <= 3) { const float m[9] = {1.00000f, 1.00000f, 1.00000f, 0.00000f, -0.09366f, 0.92780f, 0.78740f, -0.23406f, 0.00000f}; for (int i = 0; i < 9; i++) A[i] = m[i]; }
    else if (color_space <= 5) { const float m[9] = {1.00000f, 1.00000f, 1.00000f, 0.00000f, -0.08228f, 0.94070f, 0.73730f, -0.28568f, 0.00000f}; for (int i = 0; i < 9; i++) A[i] = m[i]; }
    else { const float m[9] = {0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f}; for (int i = 0; i < 9; i++) A[i] = m[i]; }
    const float sx = 1.0f / (o0 - z0), sy = 1.0f / (o1 - z1);
    Yv->bias[0] = z0; Yv->bias[1] = z1; Yv->bias[2] = z1;
    // rgb_from_yuv * mat3(scale.x, 0, 0,  0, scale.y, 0,  0, 0, scale.y)   (glsl.h mat3 * mat3: r[c] = a[0] * b[c].x + a[1] * b[c].y + a[2] * b[c].z)
    const float B[9] = {sx, 0.0f, 0.0f, 0.0f, sy, 0.0f, 0.0f, 0.0f, sy};
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) Yv->mat[3 * c + r] = A[r] * B[3 * c] + A[3 + r] * B[3 * c + 1] + A[6 + r] * B[3 * c + 2];
    // YUVMatrix::From + ctor (composite.h:652-719): the matrix in 6 (7 for y) fractional bits
    {
      const double yc = double(Yv->mat[1]), rvd = double(Yv->mat[6]), gud = double(Yv->mat[4]), gvd = double(Yv->mat[7]), bud = double(Yv->mat[5]);
      const int rs = Yv->rescale;
      Yv->brmask = Yv->mat[0] == 0.0f ? 0 : -1;
      Yv->bu = int(int16_t(bud * double(1 << (6 - rs)) + 0.5)); Yv->rv = int(int16_t(rvd * double(1 << (6 - rs)) + 0.5));
      Yv->gu = -int(int16_t(-gud * double(1 << (6 - rs)) + 0.5)); Yv->gv = -int(int16_t(-gvd * double(1 << (6 - rs)) + 0.5));
      Yv->ycoeff = int(uint16_t(yc * double(1 << (6 + 1 - rs)) + 0.5));
      Yv->ybias = int(int16_t((double(Yv->bias[0] * 255.0f) * yc - 0.5) * double(1 << 6)));
      Yv->uvbias = int(int16_t(double(Yv->bias[1] * float(255 << rs)) + 0.5));
    }
}

// The plane layouts the YUV span shader has a sampler for (sampleYUV, swgl_ext.h:1050-1150): three R8 or three R16 planes
// (YUV_FORMAT_PLANAR), R8 + RG8 (NV12) or R16 + RG16 (P010, 16-bit NV12); 16-bit planes linear-filtered.  The interleaved (YUY2)
// format and anything else is reported.
WR_DEVICE bool wr_yuv_planes_ok(const WrDrawDesc& d, int format, int depth) {
  const WrTexDesc& t0 = d.tex[WR_S_COLOR0]; const WrTexDesc& t1 = d.tex[WR_S_COLOR1]; const WrTexDesc& t2 = d.tex[WR_S_COLOR2];
  if (!t0.ptr || !t1.ptr) return false;
  if (depth != 8 && depth != 10 && depth != 12 && depth != 16) return false;
  const bool wide = t0.format == WR_FMT_R16;
  if (wide != (depth > 8)) return false;
  if (wide && !(t0.linear && t1.linear)) return false;
  if (format == 3) return t2.ptr && t1.format == t0.format && t2.format == t0.format && (t0.format == WR_FMT_R8 || (wide && t2.linear));
  if (format == 0 || format == 1) return wide ? t1.format == WR_FMT_RG16 : (t0.format == WR_FMT_R8 && t1.format == WR_FMT_RG8 && format == 0);
  return false;
}
// Under the TEXTURE_RECT keys three linear sampler2DRect planes select blendYUV's second overload (swgl_ext.h:1195-1283): the
// chunks of a span that lie inside all three clamp rects go through CompositeYUV's inner loop (linear_row_yuv: 8.8 fixed-point
// stepping, the half-resolution chroma fast path) instead of the quantised float stepping -- different roundings.  That hybrid
// is not restated in the raster stage yet: planar frames with linear samplers under a TEXTURE_RECT key are reported (the two-plane
// formats -- NV12, P010, what IOSurface video is -- and nearest samplers take the shared path and are drawn).
WR_DEVICE bool wr_yuv_rect_fast_path(const WrDrawDesc& d, int format) {
  if (!(d.flags & WR_DF_TEX_RECT) || format != 3) return false;
  if (!(d.tex[WR_S_COLOR0].linear && d.tex[WR_S_COLOR1].linear && d.tex[WR_S_COLOR2].linear)) return false;
  return d.tex[WR_S_COLOR0].width < 2 || d.tex[WR_S_COLOR1].width < 2;      // (linear_row_yuv's single-texel fill: not restated in the raster stage)
}

// ps_quad.glsl:164-418 + ps_quad_textured.glsl:13-37 (vertex stage)
// mask: 0 ps_quad_textured, 1 ps_quad_mask, 2 ps_quad_mask FAST_PATH (C = its side record)
// mask: 0 ps_quad_textured, 1 / 2 ps_quad_mask (general / FAST_PATH; C = its side record), 3 / 4 ps_quad_radial_gradient /
// ps_quad_conic_gradient (G = the gradient side record)
WR_DEVICE void wr_vs_ps_quad_textured(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, int mask = 0, WrClipRec* C = nullptr,
                                      WrGradRec* G = nullptr) {
  wi4 aData = wr_load_attr<wi4>(d, arena, inst, 0);
  int prim_address_i = aData.x, prim_address_f = aData.y;
  int quad_flags = (aData.z >> 24) & 0xff;
  int edge_flags = (aData.z >> 16) & 0xff;
  int part_index = (aData.z >> 8) & 0xff;
  int segment_index = aData.z & 0xff;
  int picture_task_address = aData.w;
  WR_TP(3);

  int hu = int(unsigned(prim_address_i) % 1024u), hv = int(unsigned(prim_address_i) / 1024u);
  wi4 header = wr_fetch_i(d.tex[WR_S_GPU_BUFFER_I], hu, hv);
  int transform_id = header.x, z_id = header.y;
  WR_TP(4);
  WrTransform transform = wr_fetch_transform(d, transform_id);
  WrTask task = wr_fetch_task(d, picture_task_address);

  int fu = int(unsigned(prim_address_f) % 1024u), fv = int(unsigned(prim_address_f) / 1024u);
  const WrTexDesc& gbf = d.tex[WR_S_GPU_BUFFER_F];
  wf4 t0 = wr_fetch_f(gbf, fu, fv), t1 = wr_fetch_f(gbf, fu + 1, fv), t2 = wr_fetch_f(gbf, fu + 2, fv);
  wf4 pso = wr_fetch_f(gbf, fu + 3, fv), prim_color = wr_fetch_f(gbf, fu + 4, fv);
  float z = float(z_id);
  WR_TP(5);

  wf4 seg_rect, seg_uv;
  if (segment_index == 0xff) { seg_rect = t0; seg_uv = t2; }
  else {
    int base = prim_address_f + 5 + segment_index * 2;
    int su = int(unsigned(base) % 1024u), sv = int(unsigned(base) / 1024u);
    seg_rect = wr_fetch_f(gbf, su, sv);
    seg_uv = wr_fetch_f(gbf, su + 1, sv);
  }
  // local_coverage_rect
  float l0x = wr_max(seg_rect.x, t1.x), l0y = wr_max(seg_rect.y, t1.y);
  float l1x = wr_min(seg_rect.z, t1.z), l1y = wr_min(seg_rect.w, t1.w);
  l1x = wr_max(l0x, l1x); l1y = wr_max(l0y, l1y);
  int aa = 0;
  switch (part_index) {
    case 1: l1x = l0x + 2.0f; aa = 1; break;
    case 2: l0x = l0x + 2.0f; l1x = l1x - 2.0f; l1y = l0y + 2.0f; aa = 2; break;
    case 3: l0x = l1x - 2.0f; aa = 4; break;
    case 4: l0x = l0x + 2.0f; l1x = l1x - 2.0f; l0y = l1y - 2.0f; aa = 8; break;
    case 0:
      l0x += (edge_flags & 1) ? 2.0f : 0.0f;
      l1x -= (edge_flags & 4) ? 2.0f : 0.0f;
      l0y += (edge_flags & 2) ? 2.0f : 0.0f;
      l1y -= (edge_flags & 8) ? 2.0f : 0.0f;
      break;
    default: aa = edge_flags; break;
  }
  o.aa_edges = aa;
  o.has_mask = 0;
  float dps = task.dps;
  if (quad_flags & 4) dps = 1.0f;
  float fox = -task.origin.x + task.p0.x, foy = -task.origin.y + task.p0.y;

  // pattern transform of the segment rect (scale_offset_map_rect)
  float sr0x = seg_rect.x * pso.x + pso.z, sr0y = seg_rect.y * pso.y + pso.w;
  float sr1x = seg_rect.z * pso.x + pso.z, sr1y = seg_rect.w * pso.y + pso.w;
  bool textured = (seg_uv.x != seg_uv.z) || (seg_uv.y != seg_uv.w);
  float tsx = 1.f, tsy = 1.f;
  if (textured) {
    tsx = float(d.tex[WR_S_COLOR0].ptr ? d.tex[WR_S_COLOR0].width : 1);
    tsy = float(d.tex[WR_S_COLOR0].ptr ? d.tex[WR_S_COLOR0].height : 1);
  }
  float mlx[4], mly[4];
  for (int n = 0; n < 4; n++) {
    float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    // mix(p0, p1, aPosition) = (p1 - p0) * a + p0
    float lx = (l1x - l0x) * ax + l0x, ly = (l1y - l0y) * ay + l0y;
    wf4 world = wr_mul(transform.m, wf4{lx, ly, 0.0f, 1.0f});
    float dx = world.x * dps, dy = world.y * dps;
    float vlx = lx, vly = ly;
    if (quad_flags & 2) {  // QF_APPLY_DEVICE_CLIP
      float c1x = task.origin.x + task.p1.x - task.p0.x, c1y = task.origin.y + task.p1.y - task.p0.y;
      dx = wr_clamp(dx, task.origin.x, c1x);
      dy = wr_clamp(dy, task.origin.y, c1y);
      wf4 lp = wr_mul(transform.inv_m, wf4{dx / dps, dy / dps, 0.0f, 1.0f});
      vlx = lp.x; vly = lp.y;
    }
    wf4 gp = wr_mul(*(const WrMat4*)d.transform,
                    wf4{dx + fox * world.w, dy + foy * world.w, z * world.w, world.w});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
    mlx[n] = vlx * pso.x + pso.z; mly[n] = vly * pso.y + pso.w;      // prim_info.local_pos (scale_offset_map_point)
    if (textured) {
      float ilx = vlx * pso.x + pso.z, ily = vly * pso.y + pso.w;
      float fx = (ilx - sr0x) / (sr1x - sr0x), fy = (ily - sr0y) / (sr1y - sr0y);
      float uvx = (seg_uv.z - seg_uv.x) * fx + seg_uv.x, uvy = (seg_uv.w - seg_uv.y) * fy + seg_uv.y;
      o.u[n] = uvx / tsx; o.v[n] = uvy / tsy;
    } else { o.u[n] = 0.f; o.v[n] = 0.f; }
  }
  if (mask == 3 || mask == 4) {
    // pattern_vertex of ps_quad_radial_gradient.glsl:37-53 / ps_quad_conic_gradient.glsl:40-56: info.pattern_input = header.zw
    const int gu = int(unsigned(header.z) % 1024u), gv = int(unsigned(header.z) / 1024u);
    const wf4 g0 = wr_fetch_f(gbf, gu, gv), g1 = wr_fetch_f(gbf, gu + 1, gv);
    const float p0x = t0.x * pso.x + pso.z, p0y = t0.y * pso.y + pso.w;      // info.local_prim_rect.p0
    G->address = header.w;
    G->repeat = g1.w;
    G->no_tile = 1;
    G->scale_dir[0] = G->scale_dir[1] = 0.0f;
    G->conic_scale = G->conic_angle = 0.0f;
    const float dd_ = g1.y - g1.x;
    const float inv = dd_ != 0.0f ? 1.0f / dd_ : 0.0f;
    G->start_offset = g1.x * inv;
    if (mask == 3) {
      G->radial = 1;
      for (int n = 0; n < 4; n++) {
        o.u[n] = ((mlx[n] - p0x) * g0.z - g0.x) * inv;
        o.v[n] = (((mly[n] - p0y) * g0.w - g0.y) * inv) * g1.z;
      }
      const int ax = int(unsigned(header.w) % 1024u), ay = int(unsigned(header.w) / 1024u);
      const bool ok = gbf.format == WR_FMT_RGBA32F && gbf.ptr && ay >= 0 && ay < gbf.height && ax >= 0 && ax < gbf.width && ax + 2 * 130 <= gbf.width;
      G->stops = ok ? (const float*)gbf.ptr + (size_t)ay * gbf.stride + (size_t)ax * 4 : nullptr;
    } else {
      G->radial = 3;
      G->conic_scale = inv;
      G->conic_angle = 3.141592653589793f / 2.0f - g1.z;
      G->stops = nullptr;
      for (int n = 0; n < 4; n++) { o.u[n] = (mlx[n] - p0x) * g0.z - g0.x; o.v[n] = (mly[n] - p0y) * g0.w - g0.y; }
    }
    // (main() multiplies the gradient by v_color and swizzles masks; the span shader does neither: only white, non-mask quads)
    const bool plain = prim_color.x == 1.0f && prim_color.y == 1.0f && prim_color.z == 1.0f && prim_color.w == 1.0f && !(quad_flags & 16);
    o.kind = plain ? WR_PK_GRADIENT : WR_PK_UNSUPPORTED;
    o.color = wf4{1.f, 1.f, 1.f, 1.f}; o.has_color = 0;
    o.tex_slot = WR_S_GPU_BUFFER_F;
    return;
  }
  if (mask) {
    // pattern_vertex (ps_quad_mask.glsl:65-152)
    const wi4 cd = wr_load_attr<wi4>(d, arena, inst, 1);          // aClipData: [clip transform id, clip address, clip space, -]
    const int cu_ = int(unsigned(cd.y) % 1024u), cv_ = int(unsigned(cd.y) / 1024u);
    const wf4 c0 = wr_fetch_f(gbf, cu_, cv_), c1 = wr_fetch_f(gbf, cu_ + 1, cv_), c2 = wr_fetch_f(gbf, cu_ + 2, cv_), c3 = wr_fetch_f(gbf, cu_ + 3, cv_);
    const WrTransform ct = wr_fetch_transform(d, cd.x);
    const bool fast = mask == 2;
    const float mode = fast ? c2.x : c3.x;
    float cw[4];
    for (int n = 0; n < 4; n++) {
      const wf4 lp = wr_mul(ct.m, wf4{mlx[n], mly[n], 0.0f, 1.0f});
      o.u[n] = lp.x; o.v[n] = lp.y; cw[n] = lp.w;
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) C->center_radius[i][j] = 0.0f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) C->plane[i][j] = 0.0f;
    for (int i = 0; i < 4; i++) C->bounds[i] = 0.0f;
    C->params[0] = C->params[1] = C->params[2] = 0.0f;
    C->fast = fast ? 1 : 0; C->mode = mode; C->w = cw[0];
    if (fast) {
      const float hx = 0.5f * (c0.z - c0.x), hy = 0.5f * (c0.w - c0.y), radius = c1.x;
      for (int n = 0; n < 4; n++) { o.u[n] = o.u[n] - (hx + c0.x) * cw[n]; o.v[n] = o.v[n] - (hy + c0.y) * cw[n]; }
      C->params[0] = hx - radius; C->params[1] = hy - radius; C->params[2] = radius;
    } else {
      if (cd.z == 0) { C->bounds[0] = c0.x; C->bounds[1] = c0.y; C->bounds[2] = c0.z; C->bounds[3] = c0.w; }   // CLIP_SPACE_RASTER
      else {       // clip rect ∩ prim_info.local_clip_rect (the prim's clip through the pattern scale-offset)
        const float pc0x = t1.x * pso.x + pso.z, pc0y = t1.y * pso.y + pso.w, pc1x = t1.z * pso.x + pso.z, pc1y = t1.w * pso.y + pso.w;
        C->bounds[0] = wr_max(c0.x, pc0x); C->bounds[1] = wr_max(c0.y, pc0y); C->bounds[2] = wr_min(c0.z, pc1x); C->bounds[3] = wr_min(c0.w, pc1y);
      }
      // corners in WrClipRec order TL, TR, BR, BL; radii_top = (tl, tr), radii_bottom = (bl, br)
      const float rx[4] = {c1.x, c1.z, c2.z, c2.x}, ry[4] = {c1.y, c1.w, c2.w, c2.y};
      const float ccx[4] = {c0.x + rx[0], c0.z - rx[1], c0.z - rx[2], c0.x + rx[3]};
      const float ccy[4] = {c0.y + ry[0], c0.y + ry[1], c0.w - ry[2], c0.w - ry[3]};
      // half-space normals and the diagonal point they pass through (:118-136)
      const float nx[4] = {-ry[0], ry[1], ry[2], -ry[3]}, ny[4] = {-rx[0], -rx[1], rx[2], rx[3]};
      const float dx[4] = {c0.x, c0.z - rx[1], c0.z, c0.x + rx[3]}, dy[4] = {c0.y + ry[0], c0.y, c0.w - ry[2], c0.w};
      for (int k = 0; k < 4; k++) {
        C->center_radius[k][0] = ccx[k]; C->center_radius[k][1] = ccy[k];
        C->center_radius[k][2] = 1.0f / wr_max(rx[k] * rx[k], 1.0e-6f);     // inverse_radii_squared, ellipse.glsl:33-35
        C->center_radius[k][3] = 1.0f / wr_max(ry[k] * ry[k], 1.0e-6f);
        C->plane[k][0] = nx[k]; C->plane[k][1] = ny[k]; C->plane[k][2] = nx[k] * dx[k] + ny[k] * dy[k];
      }
    }
    o.kind = (cw[1] != cw[0] || cw[2] != cw[0] || cw[3] != cw[0] || !(quad_flags & 16)) ? WR_PK_UNSUPPORTED : WR_PK_QUAD_MASK;   // affine clip transforms, QF_IS_MASK
    o.color = prim_color; o.has_color = 0;
    o.tex_slot = WR_S_GPU_BUFFER_F;
    return;
  }
  if (textured) {
    o.kind = (quad_flags & 16) ? WR_PK_UNSUPPORTED : WR_PK_TEX_RGBA8;
    o.color = wf4{1.f, 1.f, 1.f, 1.f};
    o.has_color = 1;   // swgl_commitTextureLinearColorRGBA8(..., v_color)
    o.tail_clamp = 1; o.tail_modulate = 1;
    o.uv_bounds = wf4{(seg_uv.x + 0.5f) / tsx, (seg_uv.y + 0.5f) / tsy, (seg_uv.z - 0.5f) / tsx, (seg_uv.w - 0.5f) / tsy};
    o.tex_slot = WR_S_COLOR0;
  } else {
    o.kind = WR_PK_SOLID;
    o.color = prim_color;
    o.has_color = 0;
  }
}

// brush.glsl:95-222 + prim_shared.glsl:54-200 + brush_solid.glsl:22-40
// `image`: 0 = brush_solid, 1 = brush_image (opaque pass), 2 = brush_image ALPHA_PASS (brush_image.glsl:54-314,
// fast variant: no REPETITION / ANTIALIASING feature)
// image: 0 brush_solid, 1 brush_image, 2 brush_image ALPHA_PASS, 3 brush_linear_gradient (G = its side record),
//        4 brush_blend (F = its side record)
// image: 0 brush_solid, 1 / 2 brush_image (opaque / ALPHA_PASS), 3 linear gradient, 4 brush_blend,
//        5 / 6 brush_image with REPETITION (opaque / ALPHA_PASS; Rp = its side record), 7 brush_opacity, 8 brush_mix_blend (Mx),
//        9 brush_image DUAL_SOURCE_BLENDING, 10 brush_yuv_image (Yv), 11 brush_image REPETITION + DUAL_SOURCE_BLENDING (Rp)
WR_DEVICE void wr_vs_brush(const WrDrawDesc& d, const uint8_t* arena, int inst, int image, WrVsOut& o, WrGradRec* G = nullptr,
                            WrFilterRec* F = nullptr, WrRepeatRec* Rp = nullptr, WrMixRec* Mx = nullptr, WrYuvRec* Yv = nullptr) {
  const bool repetition = image == 5 || image == 6 || image == 11;      // (11: the ALPHA_PASS repetition program with DUAL_SOURCE_BLENDING)
  wi4 aData = wr_load_attr<wi4>(d, arena, inst, 0);
  int prim_header_address = aData.x, clip_address = aData.y;
  int segment_index = aData.z & 0xffff, flags = aData.z >> 16;
  const int resource_address = aData.w & 0xffffff;
  const int vecs_per_brush = image == 3 ? 2 : ((image && image != 10) ? 3 : 1);   // VECS_PER_SPECIFIC_BRUSH (brush_yuv_image: 1)
  // fetch_prim_header
  int u, v;
  wr_fetch_uv(prim_header_address, 2u, u, v);
  wf4 local_rect = wr_fetch_f(d.tex[WR_S_PRIM_HEADERS_F], u, v);
  wf4 local_clip = wr_fetch_f(d.tex[WR_S_PRIM_HEADERS_F], u + 1, v);
  wi4 data0 = wr_fetch_i(d.tex[WR_S_PRIM_HEADERS_I], u, v);
  wi4 data1 = wr_fetch_i(d.tex[WR_S_PRIM_HEADERS_I], u + 1, v);
  float z = float(data0.x);
  int specific = data0.y, transform_id = data0.z, task_address = data0.w;
  WrTransform transform = wr_fetch_transform(d, transform_id);
  WrTask task = wr_fetch_task(d, task_address);
  // fetch_clip_area
  wf2 ca_p0 = {0.f, 0.f}, ca_p1 = {0.f, 0.f}, ca_origin = {0.f, 0.f};
  if (clip_address < 0x7FFFFFFF) {
    WrTask ct = wr_fetch_task(d, clip_address);
    ca_p0 = ct.p0; ca_p1 = ct.p1; ca_origin = ct.origin;
  }
  int edge_flags = (flags >> 12) & 0xf, brush_flags = flags & 0xfff;
  wf4 seg = local_rect;
  wf4 seg_data = {0.f, 0.f, 0.f, 0.f};
  if (segment_index != 0xffff) {
    int sa = specific + vecs_per_brush + segment_index * 2;
    wf4 i0 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(sa) % 1024u), int(unsigned(sa) / 1024u));
    seg = wf4{i0.x + local_rect.x, i0.y + local_rect.y, i0.z + local_rect.x, i0.w + local_rect.y};
    seg_data = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(sa) % 1024u) + 1, int(unsigned(sa) / 1024u));
  }
  wf4 adj = seg;
  int aa = 0;
  bool antialiased = !transform.axis_aligned || (brush_flags & 1024);
  if (antialiased) {
    aa = edge_flags | (local_clip.x > seg.x ? 1 : 0) | (local_clip.y > seg.y ? 2 : 0) |
         (local_clip.z < seg.z ? 4 : 0) | (local_clip.w < seg.w ? 8 : 0);
    adj.x = wr_clamp(seg.x, local_clip.x, local_clip.z); adj.y = wr_clamp(seg.y, local_clip.y, local_clip.w);
    adj.z = wr_clamp(seg.z, local_clip.x, local_clip.z); adj.w = wr_clamp(seg.w, local_clip.y, local_clip.w);
    local_clip = wf4{-1.0e16f, -1.0e16f, 1.0e16f, 1.0e16f};
  }
  o.aa_edges = aa;
  // write_clip -> swgl_clipMask: enabled iff bb_size != 0 (swgl_ext.h:1867-1876)
  o.has_mask = ((ca_p1.x - ca_p0.x) != 0.0f || (ca_p1.y - ca_p0.y) != 0.0f) ? 1 : 0;
  // write_clip (prim_shared.glsl:183-200): (task_rect.p0 - content_origin) - (area.task_rect.p0 - area.screen_origin)
  o.mask_offset[0] = (task.p0.x - task.origin.x) - (ca_p0.x - ca_origin.x);
  o.mask_offset[1] = (task.p0.y - task.origin.y) - (ca_p0.y - ca_origin.y);
  o.mask_bb[0] = ca_p0.x; o.mask_bb[1] = ca_p0.y; o.mask_bb[2] = ca_p1.x - ca_p0.x; o.mask_bb[3] = ca_p1.y - ca_p0.y;
  float fox = -task.origin.x + task.p0.x, foy = -task.origin.y + task.p0.y;
  float vlx[4], vly[4], vww[4];
  for (int n = 0; n < 4; n++) {
    float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    float lx = (adj.z - adj.x) * ax + adj.x, ly = (adj.w - adj.y) * ay + adj.y;
    lx = wr_clamp(lx, local_clip.x, local_clip.z); ly = wr_clamp(ly, local_clip.y, local_clip.w);
    wf4 world = wr_mul(transform.m, wf4{lx, ly, 0.0f, 1.0f});
    float dx = world.x * task.dps, dy = world.y * task.dps;
    wf4 gp = wr_mul(*(const WrMat4*)d.transform,
                    wf4{dx + fox * world.w, dy + foy * world.w, z * world.w, world.w});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
    o.u[n] = 0.f; o.v[n] = 0.f;
    vlx[n] = lx; vly[n] = ly; vww[n] = world.w;
  }
  wf4 color = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(specific) % 1024u), int(unsigned(specific) / 1024u));
  if (!image) {
    // brush_vs (brush_solid.glsl:24-40)
    float opacity = float(data1.x) / 65535.0f;
    o.color = wf4{color.x * opacity, color.y * opacity, color.z * opacity, color.w * opacity};
    o.kind = WR_PK_SOLID;
    o.has_color = 0;
    return;
  }
  if (image == 3) {
    // brush_vs (brush_linear_gradient.glsl:31-64) + write_gradient_vertex (gradient_shared.glsl:19-52)
    const wf4 d1 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(specific) % 1024u) + 1, int(unsigned(specific) / 1024u));
    const int extend_mode = int(d1.x);
    const float stx = d1.y, sty = d1.z;
    for (int n = 0; n < 4; n++) {
      float vx, vy;
      if (brush_flags & 2) {                   // BRUSH_FLAG_SEGMENT_RELATIVE
        vx = (vlx[n] - seg.x) / (seg.z - seg.x); vy = (vly[n] - seg.y) / (seg.w - seg.y);
        vx = vx * (seg_data.z - seg_data.x) + seg_data.x; vy = vy * (seg_data.w - seg_data.y) + seg_data.y;
        vx = vx * (local_rect.z - local_rect.x); vy = vy * (local_rect.w - local_rect.y);
      } else {
        vx = vlx[n] - local_rect.x; vy = vly[n] - local_rect.y;
      }
      o.u[n] = vx / stx; o.v[n] = vy / sty;
    }
    const float dirx = color.z - color.x, diry = color.w - color.y;   // end_point - start_point
    const float dd = dirx * dirx + diry * diry;
    float sdx = dirx / dd, sdy = diry / dd;
    G->start_offset = color.x * sdx + color.y * sdy;
    G->scale_dir[0] = sdx * stx; G->scale_dir[1] = sdy * sty;
    G->address = data1.x;
    G->repeat = extend_mode == 1 ? 1.0f : 0.0f;
    G->no_tile = 0; G->radial = 0;
    // swgl_validateGradient(sGpuBufferF, get_gpu_buffer_uv(address), 130) (swgl_ext.h:1336-1347)
    const WrTexDesc& gb = d.tex[WR_S_GPU_BUFFER_F];
    const int ax = int(unsigned(data1.x) % 1024u), ay = int(unsigned(data1.x) / 1024u);
    const bool ok = gb.format == WR_FMT_RGBA32F && gb.ptr && ay >= 0 && ay < gb.height && ax >= 0 && ax < gb.width &&
                    ax + 2 * 130 <= gb.width;
    G->stops = ok ? (const float*)gb.ptr + (size_t)ay * gb.stride + (size_t)ax * 4 : nullptr;
    o.tex_slot = WR_S_GPU_BUFFER_F;
    o.has_color = 0; o.color = wf4{1.f, 1.f, 1.f, 1.f};
    o.kind = WR_PK_GRADIENT;     // (v_pos carries no w factor: BRUSH_FLAG_PERSPECTIVE_INTERPOLATION does not enter, persp_div stays < 0)
    return;
  }
  if (image == 7) {
    // brush_vs (brush_opacity.glsl:25-63); span: swgl_commitTextureLinearColorRGBA8(sColor0, uv, v_uv_sample_bounds, v_opacity)
    const int src_address = data1.x;
    const int sau = int(unsigned(src_address) % 1024u), sav = int(unsigned(src_address) / 1024u);
    const wf4 res0 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], sau, sav);
    const int qa = src_address + 2;
    const int qu = int(unsigned(qa) % 1024u), qv = int(unsigned(qa) / 1024u);
    const wf4 st_tl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu, qv), st_tr = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 1, qv);
    const wf4 st_bl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 2, qv), st_br = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 3, qv);
    const WrTexDesc& tex = d.tex[WR_S_COLOR0];
    const float tsx = float(tex.ptr ? tex.width : 1), tsy = float(tex.ptr ? tex.height : 1);
    const float persp = (brush_flags & 1) ? 1.0f : 0.0f;
    for (int n = 0; n < 4; n++) {
      float fx = (vlx[n] - local_rect.x) / (local_rect.z - local_rect.x), fy = (vly[n] - local_rect.y) / (local_rect.w - local_rect.y);
      const float xx = (st_tr.x - st_tl.x) * fx + st_tl.x, xy = (st_tr.y - st_tl.y) * fx + st_tl.y, xw = (st_tr.w - st_tl.w) * fx + st_tl.w;
      const float yx = (st_br.x - st_bl.x) * fx + st_bl.x, yy = (st_br.y - st_bl.y) * fx + st_bl.y, yw = (st_br.w - st_bl.w) * fx + st_bl.w;
      const float zx = (yx - xx) * fy + xx, zy = (yy - xy) * fy + xy, zw = (yw - xw) * fy + xw;
      fx = zx / zw; fy = zy / zw;
      const float uu = (res0.z - res0.x) * fx + res0.x, vv = (res0.w - res0.y) * fy + res0.y;
      const float pm = (1.0f - vww[n]) * persp + vww[n];
      o.u[n] = uu / tsx * pm; o.v[n] = vv / tsy * pm;
    }
    o.uv_bounds = wf4{(res0.x + 0.5f) / tsx, (res0.y + 0.5f) / tsy, (res0.z - 0.5f) / tsx, (res0.w - 0.5f) / tsy};
    o.tex_slot = WR_S_COLOR0;
    const float opacity = wr_clamp(float(data1.y) / 65536.0f, 0.0f, 1.0f);
    o.color = wf4{opacity, opacity, opacity, opacity};
    o.has_color = 1; o.tail_clamp = 1; o.tail_modulate = 1;
    o.kind = tex.format == WR_FMT_RGBA8 ? WR_PK_TEX_RGBA8 : WR_PK_UNSUPPORTED;
    o.persp_div = persp;           // brush_opacity.glsl:68-70, as brush_image
    return;
  }
  if (image == 10) {
    // brush_vs (brush_yuv_image.glsl:40-94) + get_rgb_from_ycbcr_info (yuv.glsl:96-165), strict fp32 in the shader's order
    const int depth = int(color.x), color_space = int(color.y), format = int(color.z);      // fetch_yuv_primitive: the brush's one gpu-cache block
    wr_yuv_setup(*Yv, depth, color_space, format);
    // write_uv_rect per plane (yuv.glsl:167-183)
    const int planes = format == 3 ? 3 : ((format == 0 || format == 1) ? 2 : (format == 4 ? 1 : 0));
    for (int pl = 0; pl < 3; pl++) {
      float* ou = pl == 0 ? o.u : (pl == 1 ? o.u2 : o.u3); float* ov = pl == 0 ? o.v : (pl == 1 ? o.v2 : o.v3);
      float bnd[4] = {0.f, 0.f, 0.f, 0.f};
      if (pl < planes) {
        const int ra = pl == 0 ? data1.x : (pl == 1 ? data1.y : data1.z);
        const wf4 res = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(ra) % 1024u), int(unsigned(ra) / 1024u));      // fetch_image_source().uv_rect
        const WrTexDesc& tex = d.tex[WR_S_COLOR0 + pl];
        const float tsx = tex.ptr ? tex.sw : 1.0f, tsy = tex.ptr ? tex.sh : 1.0f;      // TEX_SIZE_YUV: 1 under TEXTURE_RECT (yuv.glsl:18-22)
        for (int n = 0; n < 4; n++) {
          const float fx = (vlx[n] - local_rect.x) / (local_rect.z - local_rect.x), fy = (vly[n] - local_rect.y) / (local_rect.w - local_rect.y);
          ou[n] = ((res.z - res.x) * fx + res.x) / tsx; ov[n] = ((res.w - res.y) * fy + res.y) / tsy;
        }
        bnd[0] = (res.x + 0.5f) / tsx; bnd[1] = (res.y + 0.5f) / tsy; bnd[2] = (res.z - 0.5f) / tsx; bnd[3] = (res.w - 0.5f) / tsy;
      } else {
        for (int n = 0; n < 4; n++) { ou[n] = 0.0f; ov[n] = 0.0f; }
      }
      if (pl == 0) o.uv_bounds = wf4{bnd[0], bnd[1], bnd[2], bnd[3]};
      else for (int k = 0; k < 4; k++) (pl == 1 ? Yv->u_bounds : Yv->v_bounds)[k] = bnd[k];
    }
    o.tex_slot = WR_S_COLOR0;
    o.tail_clamp = 1;
    o.tail_modulate = d.shader == WR_SH_BRUSH_YUV_ALPHA ? 1 : 0;      // (here: main() clamps the rgb to [0, 1] -- the ALPHA_PASS key, yuv.glsl:231-235)
    o.has_color = 0; o.color = wf4{1.f, 1.f, 1.f, 1.f};
    // (axis-aligned prims; the interleaved format and rotated video: reported)
    bool fmt_ok = planes > 0 && wr_yuv_planes_ok(d, format, depth);
    if (wr_yuv_rect_fast_path(d, format)) fmt_ok = false;
    o.kind = (fmt_ok && transform.axis_aligned && vww[0] == 1.0f && vww[1] == 1.0f && vww[2] == 1.0f && vww[3] == 1.0f) ? WR_PK_YUV : WR_PK_UNSUPPORTED;
    return;
  }
  if (image == 8) {
    // brush_vs + get_uv (brush_mix_blend.glsl:26-84): the backdrop's uv (sColor0) in o.u / o.v, the source's (sColor1) in o.u2 / o.v2
    const float persp = (brush_flags & 1) ? 1.0f : 0.0f;
    for (int pass = 0; pass < 2; pass++) {
      const int res_address = pass == 0 ? data1.y : data1.z;
      const int sau = int(unsigned(res_address) % 1024u), sav = int(unsigned(res_address) / 1024u);
      const wf4 res0 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], sau, sav);
      const int qa = res_address + 2;
      const int qu = int(unsigned(qa) % 1024u), qv = int(unsigned(qa) / 1024u);
      const wf4 st_tl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu, qv), st_tr = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 1, qv);
      const wf4 st_bl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 2, qv), st_br = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 3, qv);
      const WrTexDesc& tex = d.tex[pass == 0 ? WR_S_COLOR0 : WR_S_COLOR1];
      const float itx = 1.0f / float(tex.ptr ? tex.width : 1), ity = 1.0f / float(tex.ptr ? tex.height : 1);
      for (int n = 0; n < 4; n++) {
        float fx = (vlx[n] - local_rect.x) / (local_rect.z - local_rect.x), fy = (vly[n] - local_rect.y) / (local_rect.w - local_rect.y);
        const float xx = (st_tr.x - st_tl.x) * fx + st_tl.x, xy = (st_tr.y - st_tl.y) * fx + st_tl.y, xw = (st_tr.w - st_tl.w) * fx + st_tl.w;
        const float yx = (st_br.x - st_bl.x) * fx + st_bl.x, yy = (st_br.y - st_bl.y) * fx + st_bl.y, yw = (st_br.w - st_bl.w) * fx + st_bl.w;
        const float zx = (yx - xx) * fy + xx, zy = (yy - xy) * fy + xy, zw = (yw - xw) * fy + xw;
        fx = zx / zw; fy = zy / zw;
        const float uu = (res0.z - res0.x) * fx + res0.x, vv = (res0.w - res0.y) * fy + res0.y;
        const float pm = pass == 0 ? 1.0f : (1.0f - vww[n]) * persp + vww[n];       // get_uv(.., 1.0, ..) for the backdrop
        if (pass == 0) { o.u[n] = uu * itx * pm; o.v[n] = vv * ity * pm; }
        else { o.u2[n] = uu * itx * pm; o.v2[n] = vv * ity * pm; }
      }
      const wf4 bounds = wf4{(res0.x + 0.5f) * itx, (res0.y + 0.5f) * ity, (res0.z - 0.5f) * itx, (res0.w - 0.5f) * ity};
      if (pass == 0) o.uv_bounds = bounds;
      else { Mx->s_bounds[0] = bounds.x; Mx->s_bounds[1] = bounds.y; Mx->s_bounds[2] = bounds.z; Mx->s_bounds[3] = bounds.w; }
    }
    Mx->op = data1.x;
    o.tex_slot = WR_S_COLOR0;
    o.tail_clamp = 1; o.tail_modulate = 0;
    o.has_color = 0; o.color = wf4{1.f, 1.f, 1.f, 1.f};
    o.kind = WR_PK_MIX_BLEND;
    o.persp_div = persp;              // main(): v_src_uv * mix(gl_FragCoord.w, 1.0, v_perspective.x) (brush_mix_blend.glsl; a projective transform only)
    return;
  }
  if (image == 4) {
    // brush_vs (brush_blend.glsl:43-87)
    const int src_address = data1.x;
    const int sau = int(unsigned(src_address) % 1024u), sav = int(unsigned(src_address) / 1024u);
    const wf4 res0 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], sau, sav);                   // fetch_image_source (gpu_cache.glsl:104-109)
    const int qa = src_address + 2;                                                  // fetch_image_source_extra (:127-135)
    const int qu = int(unsigned(qa) % 1024u), qv = int(unsigned(qa) / 1024u);
    const wf4 st_tl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu, qv), st_tr = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 1, qv);
    const wf4 st_bl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 2, qv), st_br = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 3, qv);
    const WrTexDesc& tex = d.tex[WR_S_COLOR0];
    const float itx = 1.0f / float(tex.ptr ? tex.width : 1), ity = 1.0f / float(tex.ptr ? tex.height : 1);
    const float persp = (brush_flags & 1) ? 1.0f : 0.0f;
    for (int n = 0; n < 4; n++) {
      float fx = (vlx[n] - local_rect.x) / (local_rect.z - local_rect.x), fy = (vly[n] - local_rect.y) / (local_rect.w - local_rect.y);
      // get_image_quad_uv (prim_shared.glsl:204-210): mix(a, b, t) = (b - a) * t + a
      const float xx = (st_tr.x - st_tl.x) * fx + st_tl.x, xy = (st_tr.y - st_tl.y) * fx + st_tl.y, xw = (st_tr.w - st_tl.w) * fx + st_tl.w;
      const float yx = (st_br.x - st_bl.x) * fx + st_bl.x, yy = (st_br.y - st_bl.y) * fx + st_bl.y, yw = (st_br.w - st_bl.w) * fx + st_bl.w;
      const float zx = (yx - xx) * fy + xx, zy = (yy - xy) * fy + xy, zw = (yw - xw) * fy + xw;
      fx = zx / zw; fy = zy / zw;
      const float uu = (res0.z - res0.x) * fx + res0.x, vv = (res0.w - res0.y) * fy + res0.y;
      const float pm = (1.0f - vww[n]) * persp + vww[n];          // mix(world_pos.w, 1.0, perspective_interpolate)
      o.u[n] = uu * itx * pm; o.v[n] = vv * ity * pm;
    }
    o.uv_bounds = wf4{(res0.x + 0.5f) * itx, (res0.y + 0.5f) * ity, (res0.z - 0.5f) * itx, (res0.w - 0.5f) * ity};
    o.tex_slot = WR_S_COLOR0;
    o.tail_clamp = 1; o.tail_modulate = 0;
    o.has_color = 0; o.color = wf4{1.f, 1.f, 1.f, 1.f};
    o.kind = WR_PK_FILTER;
    o.persp_div = persp;           // main(): v_uv * mix(gl_FragCoord.w, 1.0, perspective_interpolate) (brush_blend.glsl:93-95)
    const float amount = float(data1.z) / 65536.0f;
    const int op = data1.y & 0xffff;
    F->op = op; F->amount = amount; F->table_address = 0;
    F->funcs[0] = float((data1.y >> 28) & 0xf); F->funcs[1] = float((data1.y >> 24) & 0xf);
    F->funcs[2] = float((data1.y >> 20) & 0xf); F->funcs[3] = float((data1.y >> 16) & 0xf);
    // SetupFilterParams (blend.glsl:27-87); color_mat is column-major: m[4 * column + row]
    float* m = F->color_mat;
    for (int i = 0; i < 16; i++) m[i] = 0.0f;
    for (int i = 0; i < 4; i++) F->color_offset[i] = 0.0f;
    const float lumR = 0.2126f, lumG = 0.7152f, lumB = 0.0722f;
    const float oneMinusLumR = 1.0f - lumR, oneMinusLumG = 1.0f - lumG, oneMinusLumB = 1.0f - lumB;
    const float invAmount = 1.0f - amount;
    if (op == 1) {          // FILTER_GRAYSCALE
      m[0] = lumR + oneMinusLumR * invAmount; m[1] = lumR - lumR * invAmount; m[2] = lumR - lumR * invAmount;
      m[4] = lumG - lumG * invAmount; m[5] = lumG + oneMinusLumG * invAmount; m[6] = lumG - lumG * invAmount;
      m[8] = lumB - lumB * invAmount; m[9] = lumB - lumB * invAmount; m[10] = lumB + oneMinusLumB * invAmount;
      m[15] = 1.0f;
    } else if (op == 2) {   // FILTER_HUE_ROTATE
      const float c = cosf(amount), sn = sinf(amount);
      m[0] = lumR + oneMinusLumR * c - lumR * sn; m[1] = lumR - lumR * c + 0.143f * sn; m[2] = lumR - lumR * c - oneMinusLumR * sn;
      m[4] = lumG - lumG * c - lumG * sn; m[5] = lumG + oneMinusLumG * c + 0.140f * sn; m[6] = lumG - lumG * c + lumG * sn;
      m[8] = lumB - lumB * c + oneMinusLumB * sn; m[9] = lumB - lumB * c - 0.283f * sn; m[10] = lumB + oneMinusLumB * c + lumB * sn;
      m[15] = 1.0f;
    } else if (op == 4) {   // FILTER_SATURATE
      m[0] = invAmount * lumR + amount; m[1] = invAmount * lumR; m[2] = invAmount * lumR;
      m[4] = invAmount * lumG; m[5] = invAmount * lumG + amount; m[6] = invAmount * lumG;
      m[8] = invAmount * lumB; m[9] = invAmount * lumB; m[10] = invAmount * lumB + amount;
      m[15] = 1.0f;
    } else if (op == 5) {   // FILTER_SEPIA
      m[0] = 0.393f + 0.607f * invAmount; m[1] = 0.349f - 0.349f * invAmount; m[2] = 0.272f - 0.272f * invAmount;
      m[4] = 0.769f - 0.769f * invAmount; m[5] = 0.686f + 0.314f * invAmount; m[6] = 0.534f - 0.534f * invAmount;
      m[8] = 0.189f - 0.189f * invAmount; m[9] = 0.168f - 0.168f * invAmount; m[10] = 0.131f + 0.869f * invAmount;
      m[15] = 1.0f;
    } else if (op == 7) {   // FILTER_COLOR_MATRIX: 4 columns + offset from the GPU cache
      const int gu = int(unsigned(data1.z) % 1024u), gv = int(unsigned(data1.z) / 1024u);
      for (int k = 0; k < 4; k++) {
        const wf4 col = wr_fetch_f(d.tex[WR_S_GPU_CACHE], gu + k, gv);
        m[4 * k] = col.x; m[4 * k + 1] = col.y; m[4 * k + 2] = col.z; m[4 * k + 3] = col.w;
      }
      const int ou = int(unsigned(data1.z + 4) % 1024u), ov = int(unsigned(data1.z + 4) / 1024u);
      const wf4 off = wr_fetch_f(d.tex[WR_S_GPU_CACHE], ou, ov);
      F->color_offset[0] = off.x; F->color_offset[1] = off.y; F->color_offset[2] = off.z; F->color_offset[3] = off.w;
    } else if (op == 11) {  // FILTER_COMPONENT_TRANSFER
      F->table_address = data1.z;
    } else if (op == 10) {  // FILTER_FLOOD
      const wf4 off = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(data1.z) % 1024u), int(unsigned(data1.z) / 1024u));
      F->color_offset[0] = off.x; F->color_offset[1] = off.y; F->color_offset[2] = off.z; F->color_offset[3] = off.w;
    }
    return;
  }
  // brush_vs (brush_image.glsl:54-314)
  const wf4 raw2 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(specific) % 1024u) + 2, int(unsigned(specific) / 1024u));
  float stx = raw2.x, sty = raw2.y;
  const WrTexDesc& tex = d.tex[WR_S_COLOR0];
  const float tsx = tex.sw, tsy = tex.sh;                // texture_size: vec2(1, 1) under TEXTURE_RECT (brush_image.glsl:68-74)
  const wf4 res0 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], int(unsigned(resource_address) % 1024u), int(unsigned(resource_address) / 1024u));
  float uv0x = res0.x, uv0y = res0.y, uv1x = res0.z, uv1y = res0.w;
  wf4 lr = local_rect;
  if (stx < 0.0f) { stx = lr.z - lr.x; sty = lr.w - lr.y; }
  if (brush_flags & 2) {                     // BRUSH_FLAG_SEGMENT_RELATIVE
    lr = seg;
    stx = lr.z - lr.x; sty = lr.w - lr.y;
    if (brush_flags & 512) {                 // BRUSH_FLAG_TEXEL_RECT
      const float usx = res0.z - res0.x, usy = res0.w - res0.y;
      uv0x = res0.x + seg_data.x * usx; uv0y = res0.y + seg_data.y * usy;
      uv1x = res0.x + seg_data.z * usx; uv1y = res0.y + seg_data.w * usy;
    }
    if (repetition) {                        // WR_FEATURE_REPETITION, brush_image.glsl:100-167
      if (brush_flags & 512) {
        float rsx = stx, rsy = sty;                                   // repeated_stretch_size
        float hux = uv1x - uv0x, huy = uv1y - uv0y;                   // horizontal_uv_size
        float vux = uv1x - uv0x, vuy = uv1y - uv0y;                   // vertical_uv_size
        if (brush_flags & 256) {             // SEGMENT_NINEPATCH_MIDDLE
          rsx = seg.x - local_rect.x; rsy = seg.y - local_rect.y;
          const float epsilon = 0.001f;
          vux = uv0x - res0.x;
          if (vux < epsilon || rsx < epsilon) { vux = res0.z - uv1x; rsx = local_rect.z - seg.z; }
          huy = uv0y - res0.y;
          if (huy < epsilon || rsy < epsilon) { huy = res0.w - uv1y; rsy = local_rect.w - seg.w; }
        }
        if (brush_flags & 4) { const float uv_ratio = hux / huy; stx = rsy * uv_ratio; }
        if (brush_flags & 8) { const float uv_ratio = vuy / vux; sty = rsx * uv_ratio; }
      } else {
        if (brush_flags & 4) stx = seg_data.z - seg_data.x;
        if (brush_flags & 8) sty = seg_data.w - seg_data.y;
      }
      if (brush_flags & 16) {                // SEGMENT_REPEAT_X_ROUND
        const float sw = seg.z - seg.x;
        const float nx = wr_max(1.0f, roundf(sw / stx));
        stx = sw / nx;
      }
      if (brush_flags & 32) {
        const float sh = seg.w - seg.y;
        const float ny = wr_max(1.0f, roundf(sh / sty));
        sty = sh / ny;
      }
    }
  }
  const bool persp = (brush_flags & 1) != 0;
  if (brush_flags & 2048) { uv0x *= tsx; uv0y *= tsy; uv1x *= tsx; uv1y *= tsy; }   // NORMALIZED_UVS
  const float mnx = wr_min(uv0x, uv1x), mny = wr_min(uv0y, uv1y), mxx = wr_max(uv0x, uv1x), mxy = wr_max(uv0y, uv1y);
  o.uv_bounds = wf4{(mnx + 0.5f) / tsx, (mny + 0.5f) / tsy, (mxx - 0.5f) / tsx, (mxy - 0.5f) / tsy};   // v_uv_sample_bounds
  const float rpx = (lr.z - lr.x) / stx, rpy = (lr.w - lr.y) / sty;
  float nox = 0.0f, noy = 0.0f;                            // normalized_offset
  if (repetition) {
    if (brush_flags & 64) { const float t = rpx * 0.5f + 0.5f; nox = 1.0f - (t - floorf(t)); }     // SEGMENT_REPEAT_X_CENTERED
    if (brush_flags & 128) { const float t = rpy * 0.5f + 0.5f; noy = 1.0f - (t - floorf(t)); }
  }
  float ubx = mnx / tsx, uby = mny / tsy, ubz = mxx / tsx, ubw = mxy / tsy;                         // v_uv_bounds
  if (d.flags & WR_DF_TEX_RECT) { ubx = 0.0f; uby = 0.0f; ubz = float(tex.width); ubw = float(tex.height); }   // vec4(0, 0, textureSize(sColor0)), brush_image.glsl:254-255
  // RASTER_SCREEN (brush_image.glsl:200-205: what blurred / drop-shadow pictures are composited with, batch.rs:1537-1542): the
  // image source's four homogeneous st corners (fetch_image_source_extra) turn the local position into the uv fraction
  const bool raster_screen = data1.y != 0;
  wf4 st_tl = {0.f, 0.f, 0.f, 1.f}, st_tr = st_tl, st_bl = st_tl, st_br = st_tl;
  if (raster_screen) {
    const int qa = resource_address + 2;
    const int qu = int(unsigned(qa) % 1024u), qv = int(unsigned(qa) / 1024u);
    st_tl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu, qv); st_tr = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 1, qv);
    st_bl = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 2, qv); st_br = wr_fetch_f(d.tex[WR_S_GPU_CACHE], qu + 3, qv);
  }
  for (int n = 0; n < 4; n++) {
    float fx = (vlx[n] - lr.x) / (lr.z - lr.x), fy = (vly[n] - lr.y) / (lr.w - lr.y);
    if (raster_screen) {      // get_image_quad_uv (prim_shared.glsl:204-210): mix(a, b, t) = (b - a) * t + a
      const float xx = (st_tr.x - st_tl.x) * fx + st_tl.x, xy = (st_tr.y - st_tl.y) * fx + st_tl.y, xw = (st_tr.w - st_tl.w) * fx + st_tl.w;
      const float yx = (st_br.x - st_bl.x) * fx + st_bl.x, yy = (st_br.y - st_bl.y) * fx + st_bl.y, yw = (st_br.w - st_bl.w) * fx + st_bl.w;
      const float zx = (yx - xx) * fy + xx, zy = (yy - xy) * fy + xy, zw = (yw - xw) * fy + xw;
      fx = zx / zw; fy = zy / zw;
    }
    float uu = ((uv1x - uv0x) * fx + uv0x) - mnx, vv = ((uv1y - uv0y) * fy + uv0y) - mny;
    uu *= rpx; vv *= rpy;
    if (repetition) { uu += nox * (mxx - mnx); vv += noy * (mxy - mny); }
    uu /= tsx; vv /= tsy;
    if (!persp) { uu *= vww[n]; vv *= vww[n]; }
    if (repetition) { uu /= (ubz - ubx); vv /= (ubw - uby); }
    o.u[n] = uu; o.v[n] = vv;
  }
  o.uv_add[0] = ubx; o.uv_add[1] = uby;                   // compute_repeated_uvs: v_uv * 1 + v_uv_bounds.xy
  o.tex_slot = WR_S_COLOR0;
  o.tail_clamp = 1;
  o.kind = tex.format == WR_FMT_RGBA8 ? WR_PK_TEX_RGBA8 : WR_PK_TEX_FS;      // swgl_isTextureRGBA8, :381
  if (repetition) {
    o.kind = WR_PK_TEX_REPEAT;
    o.uv_add[0] = o.uv_add[1] = 0.0f; o.tail_clamp = 0;
    Rp->uv_repeat[0] = ubx; Rp->uv_repeat[1] = uby; Rp->uv_repeat[2] = ubz; Rp->uv_repeat[3] = ubw;
    const bool rep_alpha = image == 6 || image == 11;
    Rp->tile_repeat[0] = rep_alpha ? rpx + nox : 0.0f; Rp->tile_repeat[1] = rep_alpha ? rpy + noy : 0.0f;
    Rp->alpha_pass = rep_alpha; Rp->no_span = tex.format != WR_FMT_RGBA8 || image == 11;      // (no span shader under the dual-source key)
    if (tex.format != WR_FMT_RGBA8 && tex.format != WR_FMT_R8) { o.kind = WR_PK_UNSUPPORTED; return; }
  }
  // BRUSH_FLAG_PERSPECTIVE_INTERPOLATION: v_uv was (not) multiplied by world_pos.w above; main() multiplies by
  // mix(gl_FragCoord.w, 1.0, perspective_interpolate) -- 1 on the 2-D path, evaluated per pixel on the perspective one
  o.persp_div = persp ? 1.0f : 0.0f;
  if (image == 1 || image == 5) {
    o.has_color = 0; o.tail_modulate = 0; o.color = wf4{1.f, 1.f, 1.f, 1.f};
    return;
  }
  const int color_mode = data1.x & 0xffff, blend_mode = data1.x >> 16;
  const float opacity = float(data1.z) / 65535.0f;
  if (blend_mode == 0) color.w *= opacity;
  else { color.x *= opacity; color.y *= opacity; color.z *= opacity; color.w *= opacity; }
  wf4 vcol;
  if (color_mode == 4) vcol = color;                                          // COLOR_MODE_IMAGE
  else if (color_mode == 3) vcol = wf4{color.w, color.w, color.w, color.w};   // COLOR_MODE_COLOR_BITMAP
  else if ((image == 9 || image == 11) && (color_mode == 1 || color_mode == 5)) vcol = color;  // COLOR_MODE_SUBPX_DUAL_SOURCE / MULTIPLY_DUAL_SOURCE
  else if ((color_mode == 0 || color_mode == 2) && image != 9 && image != 11) {
    // COLOR_MODE_ALPHA / COLOR_MODE_BITMAP_SHADOW with SWGL_BLEND (brush_image.glsl:281-291): what drop shadows of pictures are
    // drawn with (ShaderColorMode::Alpha, prim_store/picture.rs) -- swgl_blendDropShadow(image_data.color): the texel is committed
    // with v_color = 1 and the shadow colour travels in the blend stage, per primitive (blend.h:680-685), as for text
    o.color = wf4{1.f, 1.f, 1.f, 1.f};
    o.has_color = 0; o.tail_modulate = 1;
    o.blend_override = WR_BLEND_DROP_SHADOW;
    o.blend_color = color;
    return;
  }
  else { o.kind = WR_PK_UNSUPPORTED; return; }                                // (the dual-source key with a drop-shadow mode: not a combination the batcher makes)
  o.color = vcol;
  o.tail_modulate = 1;
  if (image == 9 || image == 11) {
    // ALPHA_PASS + DUAL_SOURCE_BLENDING (brush_image.glsl:296-311, 369-386): no span function, the texel is not swizzled,
    // and main() writes oFragBlend = alpha_mask * v_mask_swizzle.x + alpha_mask.aaaa * v_mask_swizzle.y beside the colour
    o.dual = color_mode == 5 ? 2 : 1;
    o.dual_swz = color_mode == 1 ? color.w : (color_mode == 5 ? -color.w : 1.0f);
    if (o.kind == WR_PK_TEX_RGBA8) o.kind = WR_PK_TEX_FS;
    o.has_color = 1;
    return;
  }
  // swgl_commitTextureColorRGBA8 unless v_color == vec4(1.0) (:404-414)
  o.has_color = (vcol.x != 1.0f || vcol.y != 1.0f || vcol.z != 1.0f || vcol.w != 1.0f) ? 1 : 0;
}

// ps_split_composite.glsl:19-113 (vertex stage): one polygon of a preserve-3d context that was split against the other planes
// (batch.rs:1985-2080; picture.rs:6571-6622 leaves its four local points in the GPU cache).  The vertices are a bilinear
// blend of those points -- any convex quad, also under an identity transform -- so the prim always takes the general-quad
// path (WR_DF_QUADS is set for every draw of this program); the fragment side is the plain image one: span =
// swgl_commitTextureRGBA8(sColor0, vUv * mix(gl_FragCoord.w, 1, vPerspective.x), vUvSampleBounds), main() = the clamped texel.
WR_DEVICE void wr_vs_ps_split_composite(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o) {
  const wi4 aData = wr_load_attr<wi4>(d, arena, inst, 0);
  const int prim_header_index = aData.x, polygons_address = aData.y, render_task_index = aData.w;
  const float ci_z = float(aData.z);
  const WrTexDesc& gc = d.tex[WR_S_GPU_CACHE];
  const int gu = int(unsigned(polygons_address) % 1024u), gv = int(unsigned(polygons_address) / 1024u);
  const wf4 g0 = wr_fetch_f(gc, gu, gv), g1 = wr_fetch_f(gc, gu + 1, gv);
  int u, v;
  wr_fetch_uv(prim_header_index, 2u, u, v);
  const wf4 local_rect = wr_fetch_f(d.tex[WR_S_PRIM_HEADERS_F], u, v);
  const wi4 data0 = wr_fetch_i(d.tex[WR_S_PRIM_HEADERS_I], u, v);
  const wi4 data1 = wr_fetch_i(d.tex[WR_S_PRIM_HEADERS_I], u + 1, v);
  const WrTransform transform = wr_fetch_transform(d, data0.z);
  const WrTask task = wr_fetch_task(d, render_task_index);
  wf2 ca_p0 = {0.f, 0.f}, ca_p1 = {0.f, 0.f}, ca_origin = {0.f, 0.f};
  if (data1.w < 0x7FFFFFFF) {
    const WrTask ct = wr_fetch_task(d, data1.w);
    ca_p0 = ct.p0; ca_p1 = ct.p1; ca_origin = ct.origin;
  }
  o.aa_edges = 0;
  o.has_mask = ((ca_p1.x - ca_p0.x) != 0.0f || (ca_p1.y - ca_p0.y) != 0.0f) ? 1 : 0;
  o.mask_offset[0] = (task.p0.x - task.origin.x) - (ca_p0.x - ca_origin.x);
  o.mask_offset[1] = (task.p0.y - task.origin.y) - (ca_p0.y - ca_origin.y);
  o.mask_bb[0] = ca_p0.x; o.mask_bb[1] = ca_p0.y; o.mask_bb[2] = ca_p1.x - ca_p0.x; o.mask_bb[3] = ca_p1.y - ca_p0.y;
  const float dox = task.p0.x - task.origin.x, doy = task.p0.y - task.origin.y;       // dest_origin
  const int src = data1.x;
  const wf4 res0 = wr_fetch_f(gc, int(unsigned(src) % 1024u), int(unsigned(src) / 1024u));     // fetch_image_source
  const int qa = src + 2;
  const int qu = int(unsigned(qa) % 1024u), qv = int(unsigned(qa) / 1024u);
  const wf4 st_tl = wr_fetch_f(gc, qu, qv), st_tr = wr_fetch_f(gc, qu + 1, qv);
  const wf4 st_bl = wr_fetch_f(gc, qu + 2, qv), st_br = wr_fetch_f(gc, qu + 3, qv);
  const WrTexDesc& tex = d.tex[WR_S_COLOR0];
  const float tsx = float(tex.ptr ? tex.width : 1), tsy = float(tex.ptr ? tex.height : 1);
  const float persp = float(data1.y);
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    // bilerp(local[0], local[1], local[3], local[2], aPosition.y, aPosition.x): mix(a, b, t) = (b - a) * t + a
    const float xx = (g0.z - g0.x) * ax + g0.x, xy = (g0.w - g0.y) * ax + g0.y;        // mix(local[0], local[1], t)
    const float yx = (g1.x - g1.z) * ax + g1.z, yy = (g1.y - g1.w) * ax + g1.w;        // mix(local[3], local[2], t)
    const float lx = (yx - xx) * ay + xx, ly = (yy - xy) * ay + xy;
    const wf4 world = wr_mul(transform.m, wf4{lx, ly, 0.0f, 1.0f});
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform,
                          wf4{dox * world.w + world.x * task.dps, doy * world.w + world.y * task.dps, world.w * ci_z, world.w});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
    float fx = (lx - local_rect.x) / (local_rect.z - local_rect.x), fy = (ly - local_rect.y) / (local_rect.w - local_rect.y);
    // get_image_quad_uv (prim_shared.glsl:204-210)
    const float hxx = (st_tr.x - st_tl.x) * fx + st_tl.x, hxy = (st_tr.y - st_tl.y) * fx + st_tl.y, hxw = (st_tr.w - st_tl.w) * fx + st_tl.w;
    const float hyx = (st_br.x - st_bl.x) * fx + st_bl.x, hyy = (st_br.y - st_bl.y) * fx + st_bl.y, hyw = (st_br.w - st_bl.w) * fx + st_bl.w;
    const float zx = (hyx - hxx) * fy + hxx, zy = (hyy - hxy) * fy + hxy, zw = (hyw - hxw) * fy + hxw;
    fx = zx / zw; fy = zy / zw;
    const float uu = (res0.z - res0.x) * fx + res0.x, vv = (res0.w - res0.y) * fy + res0.y;
    const float pm = (1.0f - gp.w) * persp + gp.w;          // mix(gl_Position.w, 1.0, perspective_interpolate)
    o.u[n] = uu / tsx * pm; o.v[n] = vv / tsy * pm;
  }
  const float mnx = wr_min(res0.x, res0.z), mny = wr_min(res0.y, res0.w), mxx = wr_max(res0.x, res0.z), mxy = wr_max(res0.y, res0.w);
  o.uv_bounds = wf4{(mnx + 0.5f) / tsx, (mny + 0.5f) / tsy, (mxx - 0.5f) / tsx, (mxy - 0.5f) / tsy};
  o.tex_slot = WR_S_COLOR0;
  o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.has_color = 0; o.tail_clamp = 1; o.tail_modulate = 0;
  o.kind = tex.format == WR_FMT_RGBA8 ? WR_PK_TEX_RGBA8 : WR_PK_UNSUPPORTED;      // (picture surfaces are RGBA8; no swgl_drawSpanR8 in the program)
  o.persp_div = persp;
}

// ps_text_run.glsl:98-268, non-GLYPH_TRANSFORM branch (vertex stage), with the
// prim_shared.glsl helpers.  Colour modes that need a blend override
// (swgl_blendDropShadow / swgl_blendSubpixelText, :229-247) are "next".
WR_DEVICE void wr_vs_ps_text_run(const WrDrawDesc& d, const uint8_t* arena, int inst, bool dual_source, WrVsOut& o, bool glyph_transform = false) {
  wi4 aData = wr_load_attr<wi4>(d, arena, inst, 0);
  int prim_header_address = aData.x, clip_address = aData.y;
  int glyph_index = aData.z & 0xffff, flags = aData.z >> 16;
  int resource_address = aData.w & 0xffffff;
  int u, v;
  wr_fetch_uv(prim_header_address, 2u, u, v);
  wf4 local_rect = wr_fetch_f(d.tex[WR_S_PRIM_HEADERS_F], u, v);
  wf4 local_clip = wr_fetch_f(d.tex[WR_S_PRIM_HEADERS_F], u + 1, v);
  wi4 data0 = wr_fetch_i(d.tex[WR_S_PRIM_HEADERS_I], u, v);
  wi4 data1 = wr_fetch_i(d.tex[WR_S_PRIM_HEADERS_I], u + 1, v);
  float z = float(data0.x);
  int specific = data0.y, transform_id = data0.z, task_address = data0.w;
  WrTransform transform = wr_fetch_transform(d, transform_id);
  wf2 ca_p0 = {0.f, 0.f}, ca_p1 = {0.f, 0.f};
  if (clip_address < 0x7FFFFFFF) {
    WrTask ct = wr_fetch_task(d, clip_address);
    ca_p0 = ct.p0; ca_p1 = ct.p1;
  }
  WrTask task = wr_fetch_task(d, task_address);
  int subpx_dir = (flags >> 8) & 0xff, color_mode = flags & 0xff;
  const WrTexDesc& gc = d.tex[WR_S_GPU_CACHE];
  wf4 text_color = wr_fetch_f(gc, int(unsigned(specific) % 1024u), int(unsigned(specific) / 1024u));
  float tox = local_rect.z, toy = local_rect.w;   // text_offset = ph.local_rect.p1
  // fetch_glyph: two glyph offsets per GPU-cache block
  int ga = specific + 1 + int(unsigned(glyph_index) / 2u);
  wf4 gdata = wr_fetch_f(gc, int(unsigned(ga) % 1024u), int(unsigned(ga) / 1024u));
  float gox = (unsigned(glyph_index) % 2u == 1u) ? gdata.z : gdata.x;
  float goy = (unsigned(glyph_index) % 2u == 1u) ? gdata.w : gdata.y;
  gox += local_rect.x; goy += local_rect.y;
  // fetch_glyph_resource
  int rx = int(unsigned(resource_address) % 1024u), ry = int(unsigned(resource_address) / 1024u);
  wf4 uvr = wr_fetch_f(gc, rx, ry);
  wf4 res1 = wr_fetch_f(gc, rx + 1, ry);
  float res_scale = res1.z;
  float sbx = (subpx_dir == 1 || subpx_dir == 3) ? 0.125f : 0.5f;   // get_snap_bias
  float sby = (subpx_dir == 2 || subpx_dir == 3) ? 0.125f : 0.5f;
  o.aa_edges = 0;
  o.has_mask = ((ca_p1.x - ca_p0.x) != 0.0f || (ca_p1.y - ca_p0.y) != 0.0f) ? 1 : 0;
  const WrTexDesc& atlas = d.tex[WR_S_COLOR0];
  float tsx = float(atlas.width), tsy = float(atlas.height);
  float st0x = uvr.x / tsx, st0y = uvr.y / tsy, st1x = uvr.z / tsx, st1y = uvr.w / tsy;
  float fox = -task.origin.x + task.p0.x, foy = -task.origin.y + task.p0.y;
  bool gt_persp = false;
  if (glyph_transform) {
    // WR_FEATURE_GLYPH_TRANSFORM (ps_text_run.glsl:24-35, 130-165, 206-216): the glyphs were rasterised under the run's 2-D
    // transform, so glyph space is device space less the translation.  mat2(transform.m) * dps, its inverse (glsl.h:2905-2908:
    // the factor is float(1. / det) computed in double), the glyph rect snapped in glyph space, its bounding rect in local space.
    const float a00 = transform.m.c[0].x * task.dps, a01 = transform.m.c[0].y * task.dps;     // column 0
    const float a10 = transform.m.c[1].x * task.dps, a11 = transform.m.c[1].y * task.dps;     // column 1
    const float gtx = transform.m.c[3].x * task.dps, gty = transform.m.c[3].y * task.dps;     // glyph_translation
    const float det = a00 * a11 - a01 * a10;
    const float idf = float(1.0 / double(det));
    const float i00 = a11 * idf, i01 = -a01 * idf, i10 = -a10 * idf, i11 = a00 * idf;
    const float rgx = floorf((a00 * gox + a10 * goy) + sbx), rgy = floorf((a01 * gox + a11 * goy) + sby);
    const float rtx = floorf(((a00 * tox + a10 * toy) + gtx) + 0.5f) - gtx, rty = floorf(((a01 * tox + a11 * toy) + gty) + 0.5f) - gty;
    const float q0x = (res1.x + rgx) + rtx, q0y = (res1.y + rgy) + rty;                       // glyph_origin
    const float q1x = (q0x + uvr.z) - uvr.x, q1y = (q0y + uvr.w) - uvr.y;
    // transform_rect(glyph_rect, glyph_transform_inv)
    const float szx = q1x - q0x, szy = q1y - q0y;
    const float hx = q0x + szx * 0.5f, hy = q0y + szy * 0.5f;
    const float cxl = i00 * hx + i10 * hy, cyl = i01 * hx + i11 * hy;
    const float rdx = fabsf(i00) * (szx * 0.5f) + fabsf(i10) * (szy * 0.5f), rdy = fabsf(i01) * (szx * 0.5f) + fabsf(i11) * (szy * 0.5f);
    const float l0x = cxl - rdx, l0y = cyl - rdy, l1x = cxl + rdx, l1y = cyl + rdy;
    const bool inside = local_clip.x <= l0x && local_clip.y <= l0y && l1x <= local_clip.z && l1y <= local_clip.w;       // rect_inside_rect
    for (int n = 0; n < 4; n++) {
      const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
      float lx = (l1x - l0x) * ax + l0x, ly = (l1y - l0y) * ay + l0y;
      if (inside) {
        const float mx = (q1x - q0x) * ax + q0x, my = (q1y - q0y) * ay + q0y;
        lx = i00 * mx + i10 * my; ly = i01 * mx + i11 * my;
      }
      lx = wr_clamp(lx, local_clip.x, local_clip.z); ly = wr_clamp(ly, local_clip.y, local_clip.w);
      const wf4 world = wr_mul(transform.m, wf4{lx, ly, 0.0f, 1.0f});
      const float dx = world.x * task.dps, dy = world.y * task.dps;
      const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{dx + fox * world.w, dy + foy * world.w, z * world.w, world.w});
      o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
      if (world.w != 1.0f) gt_persp = true;
      const float fx = ((a00 * lx + a10 * ly) - q0x) / (q1x - q0x), fy = ((a01 * lx + a11 * ly) - q0y) / (q1y - q0y);
      o.u[n] = (st1x - st0x) * fx + st0x; o.v[n] = (st1y - st0y) * fy + st0y;
    }
    // gl_ClipDistance = (f, 1 - f) (SWGL_CLIP_DIST): every span is cut to where all four are >= 0 (clip_distance_range,
    // rasterize.h:564-595).  f is affine in the device position -- glyph space is device space less the translation -- and
    // zero / one on the glyph's raster rect, whose corners are whole device pixels by construction (raster_glyph_offset and
    // raster_text_offset are floor()ed), half a pixel away from every pixel centre: the intercepts swgl computes from the edge
    // interpolants land within float noise (1e-4) of those integers and round to them, so the cut IS that rect.  It travels
    // as a box in target pixels that wr_finish_prim intersects the prim's box with; the span of a row starts at the box.
    {
      const wf4 g0 = wr_mul(*(const WrMat4*)d.transform, wf4{(q0x + gtx) + fox, (q0y + gty) + foy, 0.0f, 1.0f});
      const wf4 g1 = wr_mul(*(const WrMat4*)d.transform, wf4{(q1x + gtx) + fox, (q1y + gty) + foy, 0.0f, 1.0f});
      const float x0s = (g0.x / g0.w + 1.0f) * 0.5f * d.vp_size[0] + d.vp_origin[0], x1s = (g1.x / g1.w + 1.0f) * 0.5f * d.vp_size[0] + d.vp_origin[0];
      const float y0s = (g0.y / g0.w + 1.0f) * 0.5f * d.vp_size[1] + d.vp_origin[1], y1s = (g1.y / g1.w + 1.0f) * 0.5f * d.vp_size[1] + d.vp_origin[1];
      o.cd[0] = int(floorf(wr_min(x0s, x1s) + 0.5f)); o.cd[2] = int(floorf(wr_max(x0s, x1s) + 0.5f));
      o.cd[1] = int(floorf(wr_min(y0s, y1s) + 0.5f)); o.cd[3] = int(floorf(wr_max(y0s, y1s) + 0.5f));
    }
  } else {
  float raster_scale = float(data1.x) / 65535.0f;
  float grs = raster_scale * task.dps;
  float gsi = res_scale / grs;
  float rgx = floorf(gox * grs + sbx) / res_scale, rgy = floorf(goy * grs + sby) / res_scale;
  float g0x = gsi * (res1.x + rgx) + tox, g0y = gsi * (res1.y + rgy) + toy;
  float g1x = g0x + gsi * (uvr.z - uvr.x), g1y = g0y + gsi * (uvr.w - uvr.y);
  for (int n = 0; n < 4; n++) {
    float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    float lx = (g1x - g0x) * ax + g0x, ly = (g1y - g0y) * ay + g0y;
    lx = wr_clamp(lx, local_clip.x, local_clip.z); ly = wr_clamp(ly, local_clip.y, local_clip.w);
    wf4 world = wr_mul(transform.m, wf4{lx, ly, 0.0f, 1.0f});
    float dx = world.x * task.dps, dy = world.y * task.dps;
    wf4 gp = wr_mul(*(const WrMat4*)d.transform,
                    wf4{dx + fox * world.w, dy + foy * world.w, z * world.w, world.w});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
    float fx = (lx - g0x) / (g1x - g0x), fy = (ly - g0y) / (g1y - g0y);
    o.u[n] = (st1x - st0x) * fx + st0x; o.v[n] = (st1y - st0y) * fy + st0y;
  }
  }
  o.uv_bounds = wf4{(uvr.x + 0.5f) / tsx, (uvr.y + 0.5f) / tsy, (uvr.z + -0.5f) / tsx, (uvr.w + -0.5f) / tsy};
  o.tex_slot = WR_S_COLOR0;
  o.has_color = 1; o.tail_clamp = 1; o.tail_modulate = 1;
  const bool r8 = atlas.format == WR_FMT_R8;
  const bool rgba = atlas.format == WR_FMT_RGBA8, r8lin = r8 && atlas.width >= 2 && atlas.linear;
  // ps_text_run.glsl:219-255 with SWGL_BLEND; span shader :320-338
  //   (1, 0, 0) swizzle + v_color: swgl_commitTextureLinearColor{RGBA8, R8ToRGBA8}(v_color);
  //   DUAL_SOURCE program: swgl_commitTextureLinearRGBA8, no colour (and no swizzle in the fragment shader)
  if (color_mode == 0 && r8lin && !dual_source) {                            // COLOR_MODE_ALPHA on the R8 glyph atlas
    o.kind = WR_PK_TEX_R8; o.color = text_color;
  } else if (color_mode == 3 && rgba) {                                      // COLOR_MODE_COLOR_BITMAP
    o.kind = WR_PK_TEX_RGBA8; o.color = wf4{text_color.w, text_color.w, text_color.w, text_color.w};
  } else if ((color_mode == 1 || color_mode == 2) && rgba) {      // (subpixel masks and colour-bitmap shadows live in the BGRA8 atlas)
    // COLOR_MODE_SUBPX_DUAL_SOURCE -> swgl_blendSubpixelText(text.color), COLOR_MODE_BITMAP_SHADOW ->
    // swgl_blendDropShadow(text.color): the glyph is committed with v_color = 1 and the text colour
    // travels in the blend stage (blend.h:680-691), per primitive
    o.kind = WR_PK_TEX_RGBA8;
    o.color = wf4{1.f, 1.f, 1.f, 1.f};
    o.has_color = 0;
    o.blend_override = color_mode == 1 ? WR_BLEND_SUBPIXEL_TEXT : WR_BLEND_DROP_SHADOW;
    o.blend_color = text_color;
  } else {
    o.kind = WR_PK_UNSUPPORTED; o.color = wf4{0, 0, 0, 0};
  }
  if (dual_source && o.kind == WR_PK_TEX_RGBA8) {
    if (color_mode == 3) { o.kind = WR_PK_UNSUPPORTED; return; }   // colour bitmaps are never batched with the dual-source program
    o.has_color = 0; o.tail_modulate = 1;       // main(): v_color (1) * mask, unswizzled
  }
  // (the frame builder only selects GLYPH_TRANSFORM for transforms that reduce to 2-D, ps_text_run.glsl:139-142; under a
  // projective one the clip distances are not a device rect: reported, not drawn)
  if (gt_persp) o.kind = WR_PK_UNSUPPORTED;
}

// exp() of the vertex stage.  The reference calls libm expf (glsl.h:803), which
// is correctly rounded in practice; the device evaluates in fp64 and rounds once.
WR_DEVICE float wr_expf(float x) {
#ifdef WRHIP_HOSTSIM
  return expf(x);
#else
  return (float)exp((double)x);
#endif
}

// cs_blur.glsl:47-121 (vertex stage) + the per-instance part of blendGaussianBlur
// (swgl_ext.h:951-960: bounds) and of gaussianBlurHorizontal/Vertical
// (texture.h:1176-1214: the 8.8 fixed-point tap weights, identical for every pixel).
WR_DEVICE void wr_vs_cs_blur(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrBlurRec& B) {
  const int task_address = wr_load_attr<int>(d, arena, inst, 0);
  const int src_address = wr_load_attr<int>(d, arena, inst, 1);
  const int direction = wr_load_attr<int>(d, arena, inst, 2);
  const wf4 params = wr_load_attr<wf4>(d, arena, inst, 3);   // (std deviation, region.x, region.y)
  int u, v;
  wr_fetch_uv(task_address, 2u, u, v);
  const wf4 target = wr_fetch_f(d.tex[WR_S_RENDER_TASKS], u, v);
  wr_fetch_uv(src_address, 2u, u, v);
  const wf4 src = wr_fetch_f(d.tex[WR_S_RENDER_TASKS], u, v);
  const WrTexDesc& tex = d.tex[WR_S_COLOR0];
  const float tsx = float(tex.width), tsy = float(tex.height);
  const float sigma = params.x;
  const int support = int(ceilf(1.5f * sigma)) * 2;
  float g0 = 1.0f, g1 = 1.0f;
  if (support > 0) {
    g0 = 1.0f / (sqrtf(2.0f * 3.14159265f) * sigma);
    g1 = wr_expf(-0.5f / (sigma * sigma));
    float cx = g0, cy = g1, cz = g1 * g1;
    float total = cx;
    for (int i = 1; i <= support; i += 2) {
      cx *= cy; cy *= cz;
      float sub = cx;
      cx *= cy; cy *= cz;
      sub += cx;
      total += 2.0f * sub;
    }
    g0 /= total;
  }
  B.ptr = tex.ptr; B.stride = tex.stride; B.wh = uint32_t(tex.width) | (uint32_t(tex.height) << 16);
  B.format = tex.format; B.linear = tex.linear;
  B.coeffs[0] = g0; B.coeffs[1] = g1;
  B.offset_scale[0] = direction == 0 ? 1.0f / tsx : 0.0f;
  B.offset_scale[1] = direction == 1 ? 1.0f / tsy : 0.0f;
  B.hori = B.offset_scale[0] != 0.0f ? 1 : 0;
  B.radius = support;
  B.uv_rect[0] = (src.x + 0.5f) / tsx; B.uv_rect[1] = (src.y + 0.5f) / tsy;
  B.uv_rect[2] = (src.x + params.y - 0.5f) / tsx; B.uv_rect[3] = (src.y + params.z - 0.5f) / tsy;
  B.bounds[0] = int(B.uv_rect[0] * tsx); B.bounds[1] = int(B.uv_rect[1] * tsy);
  B.bounds[2] = int(B.uv_rect[2] * tsx); B.bounds[3] = int(B.uv_rect[3] * tsy);
  {
    float coeff = g0 * 256.0f, step = g1;
    const float step2 = g1 * g1;
    B.weights[0] = uint16_t(coeff + 0.5f);
    for (int k = 1; k <= support && k <= WR_BLUR_MAX_RADIUS; k++) {
      coeff *= step; step *= step2;
      B.weights[k] = uint16_t(coeff + 0.5f);
    }
  }
  const float u0 = src.x / tsx, v0 = src.y / tsy, u1 = src.z / tsx, v1 = src.w / tsy;
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    const float x = (target.z - target.x) * ax + target.x, y = (target.w - target.y) * ay + target.y;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{x, y, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
    o.u[n] = (u1 - u0) * ax + u0; o.v[n] = (v1 - v0) * ay + v0;
  }
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{0, 0, 0, 0};
  o.uv_bounds = wf4{B.uv_rect[0], B.uv_rect[1], B.uv_rect[2], B.uv_rect[3]};
  o.tex_slot = WR_S_COLOR0;
  o.kind = support <= WR_BLUR_MAX_RADIUS ? WR_PK_BLUR : WR_PK_UNSUPPORTED;
}

// cs_scale.glsl:24-52 (vertex stage).  swgl_drawSpanRGBA8 = swgl_commitTextureLinearRGBA8
// (:61-65); there is no R8 span function, and a format mismatch makes the
// commit draw nothing (matchTextureFormat): both cases run main() per pixel.
WR_DEVICE void wr_vs_cs_scale(const WrDrawDesc& d, const uint8_t* arena, int inst, int target_format, WrVsOut& o) {
  const wf4 tr = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf4 sr = wr_load_attr<wf4>(d, arena, inst, 1);
  const float type = wr_load_attr<float>(d, arena, inst, 2);
  const WrTexDesc& tex = d.tex[WR_S_COLOR0];
  wf4 uvr = {wr_min(sr.x, sr.z), wr_min(sr.y, sr.w), wr_max(sr.x, sr.z), wr_max(sr.y, sr.w)};
  const bool unnorm = int(type) == 1;
  const float tsx = tex.sw, tsy = tex.sh;                // (1, 1) under TEXTURE_RECT: the uv stay unnormalised (cs_scale.glsl:36-44)
  if (unnorm) {
    uvr = wf4{uvr.x + 0.5f, uvr.y + 0.5f, uvr.z - 0.5f, uvr.w - 0.5f};
    uvr = wf4{uvr.x / tsx, uvr.y / tsy, uvr.z / tsx, uvr.w / tsy};
  }
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    float u = sr.x + (sr.z - sr.x) * ax, v = sr.y + (sr.w - sr.y) * ay;
    if (unnorm) { u /= tsx; v /= tsy; }
    o.u[n] = u; o.v[n] = v;
    const float x = (tr.z - tr.x) * ax + tr.x, y = (tr.w - tr.y) * ay + tr.y;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{x, y, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = uvr;
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.tail_clamp = 1; o.tail_modulate = 0;
  o.kind = (target_format == WR_FMT_RGBA8 && tex.format == WR_FMT_RGBA8) ? WR_PK_TEX_RGBA8 : WR_PK_TEX_FS;
}

// cs_svg_filter.glsl:48-167 and cs_svg_filter_node.glsl:167-395 (vertex stages), strict fp32 in the shaders' order.  No span
// function: every pixel runs main() (wr_svg_filter_pixel).  vInput1Uv travels in the uv interpolants (unclamped: FILTER_OFFSET
// adds to it before the clamp), vInput2Uv in o.u2 / o.v2.
WR_DEVICE wf4 wr_gpu_cache_direct(const WrDrawDesc& d, int u, int v) { return wr_fetch_f(d.tex[WR_S_GPU_CACHE], u, v); }     // fetch_from_gpu_cache_1_direct
WR_DEVICE float wr_vertex_srgb_to_linear(float c) {      // vertexSrgbToLinear (cs_svg_filter_node.glsl:181-189): libm powf, mix(c1, c2, step(0.04045, c))
  const float c1 = c * (1.0f / 12.92f);
  const float c2 = powf(c * (1.0f / 1.055f) + (0.055f / 1.055f), 2.4f);
  return (c2 - c1) * (c < 0.04045f ? 0.0f : 1.0f) + c1;
}
WR_DEVICE void wr_vs_cs_svg_filter(const WrDrawDesc& d, const uint8_t* arena, int inst, bool node, WrVsOut& o, WrSvgRec& S) {
  wf4 target;                       // the rect the quad covers
  wf4 so1 = {0.f, 0.f, 0.f, 0.f}, so2 = {0.f, 0.f, 0.f, 0.f};
  int in1, in2, kind, count, generic = 0;
  wi4 extra;
  wf2 user_xy = {0.f, 0.f}; float user_x = 0.0f;
  if (node) {
    target = wr_load_attr<wf4>(d, arena, inst, 0);
    so1 = wr_load_attr<wf4>(d, arena, inst, 1); so2 = wr_load_attr<wf4>(d, arena, inst, 2);
    in1 = wr_load_attr<int>(d, arena, inst, 3); in2 = wr_load_attr<int>(d, arena, inst, 4);
    kind = wr_load_attr<int>(d, arena, inst, 5); count = wr_load_attr<int>(d, arena, inst, 6);
    extra = wr_load_attr<wi4>(d, arena, inst, 7);
  } else {
    const int task_address = wr_load_attr<int>(d, arena, inst, 0);
    in1 = wr_load_attr<int>(d, arena, inst, 1); in2 = wr_load_attr<int>(d, arena, inst, 2);
    kind = wr_load_attr<int>(d, arena, inst, 3); count = wr_load_attr<int>(d, arena, inst, 4);
    generic = wr_load_attr<int>(d, arena, inst, 5);
    extra = wr_load_attr<wi4>(d, arena, inst, 6);
    const WrTask ft = wr_fetch_task(d, task_address);       // fetch_filter_task: task_rect, user_data.xyz
    target = wf4{ft.p0.x, ft.p0.y, ft.p1.x, ft.p1.y};
    user_x = ft.dps; user_xy = ft.origin;                   // user_data.x / .yz (WrTask names them for picture tasks)
  }
  S.node = node ? 1 : 0; S.kind = kind; S.input_count = count;
  const WrTexDesc& t0 = d.tex[WR_S_COLOR0];
  const WrTexDesc& t1 = d.tex[WR_S_COLOR1];
  const float ts0x = float(t0.ptr ? t0.width : 1), ts0y = float(t0.ptr ? t0.height : 1);
  const float ts1x = float(t1.ptr ? t1.width : 1), ts1y = float(t1.ptr ? t1.height : 1);
  wf2 i1p0 = {0.f, 0.f}, i1p1 = {0.f, 0.f};
  for (int n = 0; n < 4; n++) { o.u[n] = 0.f; o.v[n] = 0.f; o.u2[n] = 0.f; o.v2[n] = 0.f; }
  for (int k = 0; k < 4; k++) { S.rect1[k] = 0.f; S.rect2[k] = 0.f; S.fdata0[k] = 0.f; S.fdata1[k] = 0.f; S.funcs[k] = 0; }
  for (int k = 0; k < 16; k++) S.color_mat[k] = 0.f;
  S.data[0] = S.data[1] = 0; S.float0 = 0.f;
  for (int pass = 0; pass < 2; pass++) {
    if (count <= pass) break;
    const float tsx = pass ? ts1x : ts0x, tsy = pass ? ts1y : ts0y;
    int u, v;
    wr_fetch_uv(pass ? in2 : in1, 2u, u, v);
    const wf4 tr = wr_fetch_f(d.tex[WR_S_RENDER_TASKS], u, v);       // fetch_render_task_rect
    if (!pass) { i1p0 = {tr.x, tr.y}; i1p1 = {tr.z, tr.w}; }
    float* rect = pass ? S.rect2 : S.rect1;
    rect[0] = (tr.x + 0.5f) / tsx; rect[1] = (tr.y + 0.5f) / tsy; rect[2] = (tr.z - 0.5f) / tsx; rect[3] = (tr.w - 0.5f) / tsy;     // compute_uv_rect
    const wf4 so = pass ? so2 : so1;
    for (int n = 0; n < 4; n++) {
      const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
      float uu, vv;
      if (node) {                   // compute_uv: (task_rect.p0 + scale_and_offset.zw + scale_and_offset.xy * aPosition.xy) / texture_size
        uu = ((tr.x + so.z) + so.x * ax) / tsx; vv = ((tr.y + so.w) + so.y * ay) / tsy;
      } else {                      // compute_uv: mix(p0 / size, floor(p1) / size, aPosition.xy)
        const float u0 = tr.x / tsx, v0 = tr.y / tsy, u1 = floorf(tr.z) / tsx, v1 = floorf(tr.w) / tsy;
        uu = (u1 - u0) * ax + u0; vv = (v1 - v0) * ay + v0;
      }
      if (pass) { o.u2[n] = uu; o.v2[n] = vv; } else { o.u[n] = uu; o.v[n] = vv; }
    }
  }
  bool known = true;
  if (!node) {
    S.funcs[0] = (generic >> 12) & 0xf; S.funcs[1] = (generic >> 8) & 0xf; S.funcs[2] = (generic >> 4) & 0xf; S.funcs[3] = generic & 0xf;
    switch (kind) {
      case 0: S.data[0] = generic; break;                                                        // FILTER_BLEND
      case 1: case 6: { const wf4 c = wr_gpu_cache_direct(d, extra.x, extra.y); S.fdata0[0] = c.x; S.fdata0[1] = c.y; S.fdata0[2] = c.z; S.fdata0[3] = c.w; break; }   // FLOOD, DROP_SHADOW
      case 4: S.float0 = user_x; break;                                                          // OPACITY
      case 5: {                                                                                  // COLOR_MATRIX
        for (int k = 0; k < 4; k++) { const wf4 c = wr_gpu_cache_direct(d, extra.x + k, extra.y); S.color_mat[4 * k] = c.x; S.color_mat[4 * k + 1] = c.y; S.color_mat[4 * k + 2] = c.z; S.color_mat[4 * k + 3] = c.w; }
        const wf4 c = wr_gpu_cache_direct(d, extra.x + 4, extra.y); S.fdata0[0] = c.x; S.fdata0[1] = c.y; S.fdata0[2] = c.z; S.fdata0[3] = c.w;
        break;
      }
      case 7:                                                                                    // OFFSET
        S.fdata0[0] = -user_x / ts0x; S.fdata0[1] = -user_xy.x / ts0y;
        S.fdata1[0] = i1p0.x / ts0x; S.fdata1[1] = i1p0.y / ts0y; S.fdata1[2] = i1p1.x / ts0x; S.fdata1[3] = i1p1.y / ts0y;
        break;
      case 8: S.data[0] = extra.x; S.data[1] = extra.y; break;                                   // COMPONENT_TRANSFER
      case 10:                                                                                   // COMPOSITE
        S.data[0] = generic;
        if (generic == 6) { const wf4 c = wr_gpu_cache_direct(d, extra.x, extra.y); S.fdata0[0] = c.x; S.fdata0[1] = c.y; S.fdata0[2] = c.z; S.fdata0[3] = c.w; }
        break;
      default: break;               // (no vertex-side data; an unknown kind leaves main() at its start colour)
    }
  } else {
    switch (kind) {
      case 2: case 3: S.float0 = so2.x; break;                                                   // OPACITY
      case 38: case 39: {                                                                        // COLOR_MATRIX
        for (int k = 0; k < 4; k++) { const wf4 c = wr_gpu_cache_direct(d, extra.x + k, extra.y); S.color_mat[4 * k] = c.x; S.color_mat[4 * k + 1] = c.y; S.color_mat[4 * k + 2] = c.z; S.color_mat[4 * k + 3] = c.w; }
        const wf4 c = wr_gpu_cache_direct(d, extra.x + 4, extra.y); S.fdata0[0] = c.x; S.fdata0[1] = c.y; S.fdata0[2] = c.z; S.fdata0[3] = c.w;
        break;
      }
      case 40: case 41: S.data[0] = extra.x; S.data[1] = extra.y; break;                         // COMPONENT_TRANSFER
      case 42: case 43: { const wf4 c = wr_gpu_cache_direct(d, extra.x, extra.y); S.fdata0[0] = c.x; S.fdata0[1] = c.y; S.fdata0[2] = c.z; S.fdata0[3] = c.w; break; }   // COMPOSITE_ARITHMETIC
      case 70: case 71: case 72: case 73: {                                                      // DROP_SHADOW / FLOOD: premultiplied colour, linearised under _CONVERTSRGB
        wf4 c = kind < 72 ? wr_gpu_cache_direct(d, extra.x, extra.y) : so2;
        if (kind & 1) { c.x = wr_vertex_srgb_to_linear(c.x); c.y = wr_vertex_srgb_to_linear(c.y); c.z = wr_vertex_srgb_to_linear(c.z); }
        S.fdata0[0] = c.x * c.w; S.fdata0[1] = c.y * c.w; S.fdata0[2] = c.z * c.w; S.fdata0[3] = c.w;
        break;
      }
      case 80: case 81: case 82: case 83: S.fdata0[0] = so2.x; S.fdata0[1] = so2.y; S.fdata0[2] = so2.z; S.fdata0[3] = so2.w; break;      // MORPHOLOGY (main() has no case for it yet)
      default: break;               // (every other kind has no vertex-side data; main() falls back to its start colour where it has no case)
    }
  }
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    const float x = (target.z - target.x) * ax + target.x, y = (target.w - target.y) * ay + target.y;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{x, y, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{S.rect1[0], S.rect1[1], S.rect1[2], S.rect1[3]};
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.tail_clamp = 0; o.tail_modulate = 0;
  const bool fmt_ok = (count < 1 || !t0.ptr || t0.format == WR_FMT_RGBA8) && (count < 2 || !t1.ptr || t1.format == WR_FMT_RGBA8);
  o.kind = (known && fmt_ok) ? WR_PK_SVG_FILTER : WR_PK_UNSUPPORTED;
}

// ps_copy.glsl:18-26 (vertex stage).  No uTransform: the destination rect maps straight to the target's pixels
// (pos / (size / 2) - 1 and back through the viewport); no span function: every pixel runs main(), a texelFetch at the
// truncated texel-space uv.  Carried as WR_PK_TEX_FS on the nearest sampler with uv / texture size: for the 1:1 copies the
// renderer issues (source and destination rect of one size) a pixel centre sits half a texel from any truncation boundary, so
// floor(u / W * W) is the texel int(u) is; any other rect pair is reported, not drawn.
WR_DEVICE void wr_vs_ps_copy(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o) {
  const wf4 sr = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf4 dr = wr_load_attr<wf4>(d, arena, inst, 1);
  const wf2 ds = wr_load_attr<wf2>(d, arena, inst, 2);
  const WrTexDesc& tex = d.tex[WR_S_COLOR0];
  const float tsx = float(tex.width), tsy = float(tex.height);
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    o.u[n] = ((sr.z - sr.x) * ax + sr.x) / tsx; o.v[n] = ((sr.w - sr.y) * ay + sr.y) / tsy;
    const float x = (dr.z - dr.x) * ax + dr.x, y = (dr.w - dr.y) * ay + dr.y;
    o.px[n] = x / (ds.x * 0.5f) - 1.0f; o.py[n] = y / (ds.y * 0.5f) - 1.0f; o.pz[n] = 0.0f; o.pw[n] = 1.0f;
  }
  o.uv_bounds = wf4{0.f, 0.f, 1.f, 1.f};
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.tail_clamp = 0; o.tail_modulate = 0;
  const bool one_to_one = (sr.z - sr.x) == (dr.z - dr.x) && (sr.w - sr.y) == (dr.w - dr.y) && sr.x == floorf(sr.x) && sr.y == floorf(sr.y) &&
                          dr.x == floorf(dr.x) && dr.y == floorf(dr.y) && (sr.z - sr.x) == floorf(sr.z - sr.x) && (sr.w - sr.y) == floorf(sr.w - sr.y);
  o.kind = (one_to_one && tex.ptr) ? WR_PK_TEX_FS : WR_PK_UNSUPPORTED;
}

// cs_border_solid.glsl:84-128 (vertex stage).  No span function: every pixel runs main() (wr_border_solid_pixel).
WR_DEVICE void wr_vs_cs_border_solid(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrBorderRec& B) {
  const wf2 origin = wr_load_attr<wf2>(d, arena, inst, 0);
  const wf4 rect = wr_load_attr<wf4>(d, arena, inst, 1);
  const wf4 color0 = wr_load_attr<wf4>(d, arena, inst, 2), color1 = wr_load_attr<wf4>(d, arena, inst, 3);
  const int flags = wr_load_attr<int>(d, arena, inst, 4);
  const wf2 widths = wr_load_attr<wf2>(d, arena, inst, 5), radii = wr_load_attr<wf2>(d, arena, inst, 6);
  const wf4 cp1 = wr_load_attr<wf4>(d, arena, inst, 7), cp2 = wr_load_attr<wf4>(d, arena, inst, 8);
  const int segment = flags & 0xff;
  const bool do_aa = ((flags >> 24) & 0xf0) != 0;
  float osx = 0.0f, osy = 0.0f;        // get_outer_corner_scale
  if (segment == 1) { osx = 1.0f; } else if (segment == 2) { osx = 1.0f; osy = 1.0f; } else if (segment == 3) { osy = 1.0f; }
  const float sx = rect.z - rect.x, sy = rect.w - rect.y;
  const float ox = osx * sx, oy = osy * sy;
  const float csx = 1.0f - 2.0f * osx, csy = 1.0f - 2.0f * osy;
  B.mix = segment < 4 ? (do_aa ? 1 : 2) : 0;
  B.color0[0] = color0.x; B.color0[1] = color0.y; B.color0[2] = color0.z; B.color0[3] = color0.w;
  B.color1[0] = color1.x; B.color1[1] = color1.y; B.color1[2] = color1.z; B.color1[3] = color1.w;
  B.clip_center_sign[0] = ox + csx * radii.x; B.clip_center_sign[1] = oy + csy * radii.y; B.clip_center_sign[2] = csx; B.clip_center_sign[3] = csy;
  B.clip_radii[0] = radii.x; B.clip_radii[1] = radii.y; B.clip_radii[2] = wr_max(radii.x - widths.x, 0.0f); B.clip_radii[3] = wr_max(radii.y - widths.y, 0.0f);
  B.color_line[0] = ox; B.color_line[1] = oy; B.color_line[2] = widths.y * -csy; B.color_line[3] = widths.x * csx;
  const float hsx = -csx, hsy = csy;
  B.h_center_sign[0] = cp1.x + hsx * cp1.z; B.h_center_sign[1] = cp1.y + hsy * cp1.w; B.h_center_sign[2] = hsx; B.h_center_sign[3] = hsy;
  B.h_radii[0] = cp1.z; B.h_radii[1] = cp1.w;
  const float vsx = csx, vsy = -csy;
  B.v_center_sign[0] = cp2.x + vsx * cp2.z; B.v_center_sign[1] = cp2.y + vsy * cp2.w; B.v_center_sign[2] = vsx; B.v_center_sign[3] = vsy;
  B.v_radii[0] = cp2.z; B.v_radii[1] = cp2.w;
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    const float vx = sx * ax, vy = sy * ay;          // vPos
    o.u[n] = vx; o.v[n] = vy;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{(origin.x + rect.x) + vx, (origin.y + rect.y) + vy, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{0, 0, 0, 0};
  o.kind = WR_PK_BORDER_SOLID;
}

// cs_border_segment.glsl:118-267 (vertex stage).  No span function: every pixel runs main() (wr_border_segment_pixel).
WR_DEVICE float wr_hypotf(float x, float y) { return float(sqrt(double(x) * double(x) + double(y) * double(y))); }   // glibc hypotf
WR_DEVICE void wr_border_side_colors(wf4 color, int style, float* r0, float* r1) {      // get_colors_for_side / mod_color
  const bool is_black = color.x == 0.0f && color.y == 0.0f && color.z == 0.0f;
  const float light_black = 0.7f, dark_black = 0.3f, dark_scale = 0.66666666f, light_scale = 1.0f;
  float lighter[4], darker[4];
  if (is_black) {
    lighter[0] = lighter[1] = lighter[2] = light_black; darker[0] = darker[1] = darker[2] = dark_black;
  } else {
    lighter[0] = color.x * light_scale; lighter[1] = color.y * light_scale; lighter[2] = color.z * light_scale;
    darker[0] = color.x * dark_scale; darker[1] = color.y * dark_scale; darker[2] = color.z * dark_scale;
  }
  lighter[3] = darker[3] = color.w;
  const float plain[4] = {color.x, color.y, color.z, color.w};
  for (int i = 0; i < 4; i++) {
    r0[i] = style == 6 ? lighter[i] : (style == 7 ? darker[i] : plain[i]);
    r1[i] = style == 6 ? darker[i] : (style == 7 ? lighter[i] : plain[i]);
  }
}
WR_DEVICE void wr_vs_cs_border_segment(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrBorderSegRec& B) {
  const wf2 origin = wr_load_attr<wf2>(d, arena, inst, 0);
  const wf4 rect = wr_load_attr<wf4>(d, arena, inst, 1);
  const wf4 color0 = wr_load_attr<wf4>(d, arena, inst, 2), color1 = wr_load_attr<wf4>(d, arena, inst, 3);
  const int flags = wr_load_attr<int>(d, arena, inst, 4);
  const wf2 widths = wr_load_attr<wf2>(d, arena, inst, 5), radii = wr_load_attr<wf2>(d, arena, inst, 6);
  const wf4 cp1 = wr_load_attr<wf4>(d, arena, inst, 7), cp2 = wr_load_attr<wf4>(d, arena, inst, 8);
  const int segment = flags & 0xff, style0 = (flags >> 8) & 0xff, style1 = (flags >> 16) & 0xff, clip_mode = (flags >> 24) & 0x0f;
  float osx = 0.0f, osy = 0.0f;
  if (segment == 1) { osx = 1.0f; } else if (segment == 2) { osx = 1.0f; osy = 1.0f; } else if (segment == 3) { osy = 1.0f; }
  const float sx = rect.z - rect.x, sy = rect.w - rect.y;
  const float ox = osx * sx, oy = osy * sy;
  const float csx = 1.0f - 2.0f * osx, csy = 1.0f - 2.0f * osy;
  int eax = 0, eay = 0;
  float erx = 0.0f, ery = 0.0f;
  switch (segment) {
    case 0: eax = 0; eay = 1; erx = ox; ery = oy; break;
    case 1: eax = 1; eay = 0; erx = ox - widths.x; ery = oy; break;
    case 2: eax = 0; eay = 1; erx = ox - widths.x; ery = oy - widths.y; break;
    case 3: eax = 1; eay = 0; erx = ox; ery = oy - widths.y; break;
    case 5: case 7: eax = 1; eay = 1; break;
    default: break;
  }
  B.segment = segment; B.clip_mode = clip_mode; B.style0 = style0; B.style1 = style1; B.edge_axis[0] = eax; B.edge_axis[1] = eay;
  B.partial_widths[0] = widths.x / 3.0f; B.partial_widths[1] = widths.y / 3.0f; B.partial_widths[2] = widths.x / 2.0f; B.partial_widths[3] = widths.y / 2.0f;
  wr_border_side_colors(color0, style0, B.color00, B.color01);
  wr_border_side_colors(color1, style1, B.color10, B.color11);
  B.clip_center_sign[0] = ox + csx * radii.x; B.clip_center_sign[1] = oy + csy * radii.y; B.clip_center_sign[2] = csx; B.clip_center_sign[3] = csy;
  B.clip_radii[0] = radii.x; B.clip_radii[1] = radii.y; B.clip_radii[2] = wr_max(radii.x - widths.x, 0.0f); B.clip_radii[3] = wr_max(radii.y - widths.y, 0.0f);
  B.color_line[0] = ox; B.color_line[1] = oy; B.color_line[2] = widths.y * -csy; B.color_line[3] = widths.x * csx;
  B.edge_reference[0] = erx; B.edge_reference[1] = ery; B.edge_reference[2] = erx + widths.x; B.edge_reference[3] = ery + widths.y;
  B.cp1[0] = cp1.x; B.cp1[1] = cp1.y; B.cp1[2] = cp1.z; B.cp1[3] = cp1.w;
  B.cp2[0] = cp2.x; B.cp2[1] = cp2.y; B.cp2[2] = cp2.z; B.cp2[3] = cp2.w;
  float dot_radius = cp1.z;
  if (dot_radius > 0.5f) dot_radius += 2.0f;
  const float cenx = (cp1.x + cp2.x) * 0.5f, ceny = (cp1.y + cp2.y) * 0.5f;
  const float dash_r = wr_max(wr_hypotf(cp1.x - cp2.x, cp1.y - cp2.y), wr_max(widths.x, widths.y)) + 2.0f;
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    float vx = sx * ax, vy = sy * ay;          // vPos
    if (clip_mode == 3) {                       // CLIP_DOT: the quad shrinks to the dot's box
      vx = cp1.x + dot_radius * (2.0f * ax - 1.0f); vy = cp1.y + dot_radius * (2.0f * ay - 1.0f);
      vx = wr_clamp(vx, 0.0f, sx); vy = wr_clamp(vy, 0.0f, sy);
    } else if (clip_mode == 1) {                // CLIP_DASH_CORNER: ... to the dash's box
      vx = wr_clamp(vx, cenx - dash_r, cenx + dash_r); vy = wr_clamp(vy, ceny - dash_r, ceny + dash_r);
    }
    o.u[n] = vx; o.v[n] = vy;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{(origin.x + rect.x) + vx, (origin.y + rect.y) + vy, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{0, 0, 0, 0};
  o.kind = WR_PK_BORDER_SEGMENT;
}

// cs_linear_gradient.glsl:26-46 (vertex stage); the fragment side is brush_linear_gradient's replay with tileRepeat off
WR_DEVICE void wr_vs_cs_linear_gradient(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrGradRec* G) {
  const wf4 task = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf2 sp = wr_load_attr<wf2>(d, arena, inst, 1), ep = wr_load_attr<wf2>(d, arena, inst, 2), scale = wr_load_attr<wf2>(d, arena, inst, 3);
  const int extend_mode = wr_load_attr<int>(d, arena, inst, 4), address = wr_load_attr<int>(d, arena, inst, 5);
  const float dirx = ep.x - sp.x, diry = ep.y - sp.y;
  const float dd = dirx * dirx + diry * diry;
  const float sdx = dirx / dd, sdy = diry / dd;
  G->start_offset = sp.x * sdx + sp.y * sdy;
  G->scale_dir[0] = sdx * (task.z - task.x); G->scale_dir[1] = sdy * (task.w - task.y);
  G->address = address;
  G->repeat = extend_mode == 1 ? 1.0f : 0.0f;
  G->no_tile = 1; G->radial = 0;
  const WrTexDesc& gb = d.tex[WR_S_GPU_BUFFER_F];
  const int ax = int(unsigned(address) % 1024u), ay = int(unsigned(address) / 1024u);
  const bool ok = gb.format == WR_FMT_RGBA32F && gb.ptr && ay >= 0 && ay < gb.height && ax >= 0 && ax < gb.width && ax + 2 * 130 <= gb.width;
  G->stops = ok ? (const float*)gb.ptr + (size_t)ay * gb.stride + (size_t)ax * 4 : nullptr;
  for (int n = 0; n < 4; n++) {
    const float ax_ = d.quad[2 * n], ay_ = d.quad[2 * n + 1];
    o.u[n] = ax_ * scale.x; o.v[n] = ay_ * scale.y;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{(task.z - task.x) * ax_ + task.x, (task.w - task.y) * ay_ + task.y, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_GPU_BUFFER_F;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.kind = WR_PK_GRADIENT;
}

// cs_radial_gradient.glsl:26-47 (vertex stage)
WR_DEVICE void wr_vs_cs_radial_gradient(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrGradRec* G) {
  const wf4 task = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf2 center = wr_load_attr<wf2>(d, arena, inst, 1), scale = wr_load_attr<wf2>(d, arena, inst, 2);
  const float r0 = wr_load_attr<float>(d, arena, inst, 3), r1 = wr_load_attr<float>(d, arena, inst, 4), ratio = wr_load_attr<float>(d, arena, inst, 5);
  const int extend_mode = wr_load_attr<int>(d, arena, inst, 6), address = wr_load_attr<int>(d, arena, inst, 7);
  const float rd = r1 - r0;
  const float radius_scale = rd != 0.0f ? 1.0f / rd : 0.0f;
  G->start_offset = r0 * radius_scale;
  G->scale_dir[0] = G->scale_dir[1] = 0.0f;
  G->address = address;
  G->repeat = extend_mode == 1 ? 1.0f : 0.0f;
  G->no_tile = 1; G->radial = 1;
  const WrTexDesc& gb = d.tex[WR_S_GPU_BUFFER_F];
  const int ax = int(unsigned(address) % 1024u), ay = int(unsigned(address) / 1024u);
  const bool ok = gb.format == WR_FMT_RGBA32F && gb.ptr && ay >= 0 && ay < gb.height && ax >= 0 && ax < gb.width && ax + 2 * 130 <= gb.width;
  G->stops = ok ? (const float*)gb.ptr + (size_t)ay * gb.stride + (size_t)ax * 4 : nullptr;
  for (int n = 0; n < 4; n++) {
    const float ax_ = d.quad[2 * n], ay_ = d.quad[2 * n + 1];
    o.u[n] = (((task.z - task.x) * ax_) * scale.x - center.x) * radius_scale;
    o.v[n] = ((((task.w - task.y) * ay_) * scale.y - center.y) * radius_scale) * ratio;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{(task.z - task.x) * ax_ + task.x, (task.w - task.y) * ay_ + task.y, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_GPU_BUFFER_F;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.kind = WR_PK_GRADIENT;
}

// cs_conic_gradient.glsl:30-48 (vertex stage); no span function: main() per pixel with libm atan2f
WR_DEVICE void wr_vs_cs_conic_gradient(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrGradRec* G) {
  const wf4 task = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf2 center = wr_load_attr<wf2>(d, arena, inst, 1), scale = wr_load_attr<wf2>(d, arena, inst, 2);
  const float o0 = wr_load_attr<float>(d, arena, inst, 3), o1 = wr_load_attr<float>(d, arena, inst, 4), angle = wr_load_attr<float>(d, arena, inst, 5);
  const int extend_mode = wr_load_attr<int>(d, arena, inst, 6), address = wr_load_attr<int>(d, arena, inst, 7);
  const float dd = o1 - o0;
  const float offset_scale = dd != 0.0f ? 1.0f / dd : 0.0f;
  G->conic_scale = offset_scale;
  G->conic_angle = 3.141592653589793f / 2.0f - angle;
  G->start_offset = o0 * offset_scale;
  G->scale_dir[0] = center.x * offset_scale; G->scale_dir[1] = center.y * offset_scale;
  G->address = address;
  G->repeat = extend_mode == 1 ? 1.0f : 0.0f;
  G->no_tile = 1; G->radial = 2;
  G->stops = nullptr;                          // no span shader
  for (int n = 0; n < 4; n++) {
    const float ax_ = d.quad[2 * n], ay_ = d.quad[2 * n + 1];
    o.u[n] = (((task.z - task.x) * ax_) * offset_scale) * scale.x;
    o.v[n] = (((task.w - task.y) * ay_) * offset_scale) * scale.y;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{(task.z - task.x) * ax_ + task.x, (task.w - task.y) * ay_ + task.y, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_GPU_BUFFER_F;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.kind = WR_PK_GRADIENT;
}

// cs_fast_linear_gradient.glsl:17-24 and cs_line_decoration.glsl:43-98 (vertex stages).  Neither program has a span function.
WR_DEVICE float wr_mixf(float x, float y, float a) { return (y - x) * a + x; }      // glsl.h:2691-2700
WR_DEVICE void wr_vs_cs_fast_linear_gradient(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrFastGradRec& G) {
  const wf4 task = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf4 c0 = wr_load_attr<wf4>(d, arena, inst, 1), c1 = wr_load_attr<wf4>(d, arena, inst, 2);
  const float axis = wr_load_attr<float>(d, arena, inst, 3);
  G.color0[0] = c0.x; G.color0[1] = c0.y; G.color0[2] = c0.z; G.color0[3] = c0.w;
  G.color1[0] = c1.x; G.color1[1] = c1.y; G.color1[2] = c1.z; G.color1[3] = c1.w;
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    o.u[n] = wr_mixf(0.0f, 1.0f, wr_mixf(ax, ay, axis)); o.v[n] = 0.0f;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{wr_mixf(task.x, task.z, ax), wr_mixf(task.y, task.w, ay), 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{0, 0, 0, 0};
  o.kind = WR_PK_FAST_GRADIENT;
}
WR_DEVICE void wr_vs_cs_line_decoration(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrLineRec& L) {
  const wf4 task = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf2 local = wr_load_attr<wf2>(d, arena, inst, 1);
  const float wavy = wr_load_attr<float>(d, arena, inst, 2);
  const int style = wr_load_attr<int>(d, arena, inst, 3);
  const float axis = wr_load_attr<float>(d, arena, inst, 4);
  const float sx = wr_mixf(local.x, local.y, axis), sy = wr_mixf(local.y, local.x, axis);
  L.style = style;
  L.params[0] = L.params[1] = L.params[2] = L.params[3] = 0.0f;
  if (style == 2) { L.params[0] = sx; L.params[1] = 0.5f * sx; }
  else if (style == 1) { const float diameter = sy; L.params[0] = diameter * 2.0f; L.params[1] = diameter / 2.0f; L.params[2] = 0.5f * sy; }
  else if (style == 3) {
    const float lt = wr_max(wavy, 1.0f);
    L.params[0] = lt / 2.0f; L.params[1] = sy - lt; L.params[2] = wr_max((lt - 1.0f) * 2.0f, 1.0f); L.params[3] = sy;
  }
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    o.u[n] = wr_mixf(ax, ay, axis) * sx; o.v[n] = wr_mixf(ay, ax, axis) * sy;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{wr_mixf(task.x, task.z, ax), wr_mixf(task.y, task.w, ay), 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{0, 0, 0, 0};
  o.kind = WR_PK_LINE_DECORATION;
}

// clip_shared.glsl:43-78 write_clip_tile_vertex + transform.glsl:48-90
// (get_node_pos / untransform / ray_plane), one corner of the quad.
WR_DEVICE wf4 wr_get_node_pos(float px, float py, const WrTransform& t) {
  const wf4 ah = t.m.c[3];   // m * (0,0,0,1)
  const float ax = ah.x / ah.w, ay = ah.y / ah.w, az = ah.z / ah.w;
  const float nx = t.inv_m.c[0].z, ny = t.inv_m.c[1].z, nz = t.inv_m.c[2].z;
  const float pz = -10000.0f;
  float tt = 0.0f;
  const float denom = nx * 0.0f + ny * 0.0f + nz * 1.0f;
  if (fabsf(denom) > 1e-6f) {
    const float dx = ax - px, dy = ay - py, dz = az - pz;
    tt = (dx * nx + dy * ny + dz * nz) / denom;
  }
  const float z = pz + 1.0f * tt;
  return wr_mul(t.inv_m, wf4{px, py, z, 1.0f});
}

// cs_clip_rectangle.glsl:81-153 (vertex stage)
WR_DEVICE void wr_vs_cs_clip_rect(const WrDrawDesc& d, const uint8_t* arena, int inst, bool fast, WrVsOut& o, WrClipRec& C) {
  const wf4 area = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf4 origins = wr_load_attr<wf4>(d, arena, inst, 1);
  const float dps = wr_load_attr<float>(d, arena, inst, 2);
  const wi4 tids = wr_load_attr<wi4>(d, arena, inst, 3);
  const wf2 lpos = wr_load_attr<wf2>(d, arena, inst, 4);
  const wf4 lrect = wr_load_attr<wf4>(d, arena, inst, 5);
  const float mode = wr_load_attr<float>(d, arena, inst, 6);
  const wf4 rad_tl = wr_load_attr<wf4>(d, arena, inst, 8), rad_tr = wr_load_attr<wf4>(d, arena, inst, 10);
  const wf4 rad_bl = wr_load_attr<wf4>(d, arena, inst, 12), rad_br = wr_load_attr<wf4>(d, arena, inst, 14);
  const WrTransform clip_t = wr_fetch_transform(d, tids.x), prim_t = wr_fetch_transform(d, tids.y);
  // local_rect.p0 = local_pos; local_rect.p1 += local_pos - p0
  const float diffx = lpos.x - lrect.x, diffy = lpos.y - lrect.y;
  const float r0x = lpos.x, r0y = lpos.y, r1x = lrect.z + diffx, r1y = lrect.w + diffy;
  float lw[4];
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    const float sx = (area.z - area.x) * ax + area.x, sy = (area.w - area.y) * ay + area.y;
    const float devx = origins.z + sx, devy = origins.w + sy;
    const float wx = devx / dps, wy = devy / dps;
    wf4 pos = wr_mul(prim_t.m, wf4{wx, wy, 0.0f, 1.0f});
    pos.x /= pos.w; pos.y /= pos.w; pos.z /= pos.w;
    const wf4 p = wr_get_node_pos(pos.x, pos.y, clip_t);
    float lx = p.x * pos.w, ly = p.y * pos.w;
    lw[n] = p.w * pos.w;
    if (fast) {
      const float hx = 0.5f * (r1x - r0x), hy = 0.5f * (r1y - r0y);
      lx -= (hx + lpos.x) * lw[n]; ly -= (hy + lpos.y) * lw[n];
    }
    o.u[n] = lx; o.v[n] = ly;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{origins.x + sx, origins.y + sy, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  C.fast = fast ? 1 : 0; C.mode = mode; C.w = lw[0];
  C.bounds[0] = r0x; C.bounds[1] = r0y; C.bounds[2] = r1x; C.bounds[3] = r1y;
  if (fast) {
    const float hx = 0.5f * (r1x - r0x), hy = 0.5f * (r1y - r0y), radius = rad_tl.x;
    C.params[0] = hx - radius; C.params[1] = hy - radius; C.params[2] = radius;
  } else {
    C.params[0] = C.params[1] = C.params[2] = 0.0f;
    const float rr[4][2] = {{rad_tl.x, rad_tl.y}, {rad_tr.x, rad_tr.y}, {rad_br.x, rad_br.y}, {rad_bl.x, rad_bl.y}};
    const float cx[4] = {r0x + rr[0][0], r1x - rr[1][0], r1x - rr[2][0], r0x + rr[3][0]};
    const float cy[4] = {r0y + rr[0][1], r0y + rr[1][1], r1y - rr[2][1], r1y - rr[3][1]};
    for (int k = 0; k < 4; k++) {
      C.center_radius[k][0] = cx[k]; C.center_radius[k][1] = cy[k];
      C.center_radius[k][2] = 1.0f / wr_max(rr[k][0] * rr[k][0], 1.0e-6f);
      C.center_radius[k][3] = 1.0f / wr_max(rr[k][1] * rr[k][1], 1.0e-6f);
    }
    // half-space normals and a point on each diagonal (:132-152)
    const float nx[4] = {-rr[0][1], rr[1][1], rr[2][1], -rr[3][1]};
    const float ny[4] = {-rr[0][0], -rr[1][0], rr[2][0], rr[3][0]};
    const float qx[4] = {r0x, r1x - rr[1][0], r1x, r0x + rr[3][0]};
    const float qy[4] = {r0y + rr[0][1], r0y, r1y - rr[2][1], r1y};
    for (int k = 0; k < 4; k++) { C.plane[k][0] = nx[k]; C.plane[k][1] = ny[k]; C.plane[k][2] = nx[k] * qx[k] + ny[k] * qy[k]; }
  }
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{0, 0, 0, 0};
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_COLOR0;
  // perspective-varying w is not handled by the span rasteriser either (:226-228 falls back to main()): "next"
  o.kind = (lw[1] == lw[0] && lw[2] == lw[0] && lw[3] == lw[0]) ? WR_PK_CLIP_RECT : WR_PK_UNSUPPORTED;
}

// cs_clip_box_shadow.glsl:59-119 (vertex stage)
WR_DEVICE void wr_vs_cs_clip_box_shadow(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrBoxRec& B) {
  const wf4 area = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf4 origins = wr_load_attr<wf4>(d, arena, inst, 1);
  const float dps = wr_load_attr<float>(d, arena, inst, 2);
  const wi4 tids = wr_load_attr<wi4>(d, arena, inst, 3);
  const wi4 res = wr_load_attr<wi4>(d, arena, inst, 4);
  const wf2 src_size = wr_load_attr<wf2>(d, arena, inst, 5);
  const int mode = wr_load_attr<int>(d, arena, inst, 6);
  const wi4 stretch = wr_load_attr<wi4>(d, arena, inst, 7);
  const wf4 dest = wr_load_attr<wf4>(d, arena, inst, 8);
  const WrTransform clip_t = wr_fetch_transform(d, tids.x), prim_t = wr_fetch_transform(d, tids.y);
  const wf4 res0 = wr_fetch_f(d.tex[WR_S_GPU_CACHE], res.x, res.y);      // fetch_image_source_direct
  const WrTexDesc& tex = d.tex[WR_S_COLOR0];
  const float tsx = float(tex.width), tsy = float(tex.height);
  const float dsx = dest.z - dest.x, dsy = dest.w - dest.y;
  float lw[4];
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    const float sx = (area.z - area.x) * ax + area.x, sy = (area.w - area.y) * ay + area.y;
    const float devx = origins.z + sx, devy = origins.w + sy;
    wf4 pos = wr_mul(prim_t.m, wf4{devx / dps, devy / dps, 0.0f, 1.0f});
    pos.x /= pos.w; pos.y /= pos.w; pos.z /= pos.w;
    const wf4 p = wr_get_node_pos(pos.x, pos.y, clip_t);
    const float lx = p.x * pos.w, ly = p.y * pos.w;
    lw[n] = p.w * pos.w;
    const float px = lx / lw[n], py = ly / lw[n];
    float ux = stretch.x == 0 ? (px - dest.x) / src_size.x : (px - dest.x) / dsx;
    float uy = stretch.y == 0 ? (py - dest.y) / src_size.y : (py - dest.y) / dsy;
    o.u[n] = ux * lw[n]; o.v[n] = uy * lw[n];
    o.u2[n] = lx; o.v2[n] = ly;
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{origins.x + sx, origins.y + sy, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  B.ptr = tex.ptr; B.stride = tex.stride; B.wh = uint32_t(tex.width) | (uint32_t(tex.height) << 16);
  B.format = tex.format; B.linear = tex.linear;
  B.mode = float(mode); B.w = lw[0];
  B.edge[0] = stretch.x == 0 ? 0.5f : 1.0f; B.edge[2] = stretch.x == 0 ? (dsx / src_size.x) - 0.5f : 1.0f;
  B.edge[1] = stretch.y == 0 ? 0.5f : 1.0f; B.edge[3] = stretch.y == 0 ? (dsy / src_size.y) - 0.5f : 1.0f;
  B.uv_bounds[0] = (res0.x + 0.5f) / tsx; B.uv_bounds[1] = (res0.y + 0.5f) / tsy;
  B.uv_bounds[2] = (res0.z - 0.5f) / tsx; B.uv_bounds[3] = (res0.w - 0.5f) / tsy;
  B.uv_noclamp[0] = res0.x / tsx; B.uv_noclamp[1] = res0.y / tsy; B.uv_noclamp[2] = res0.z / tsx; B.uv_noclamp[3] = res0.w / tsy;
  B.bounds[0] = dest.x; B.bounds[1] = dest.y; B.bounds[2] = dest.z; B.bounds[3] = dest.w;
  o.aa_edges = 0; o.has_mask = 0; o.has_color = 0;
  o.color = wf4{0, 0, 0, 0};
  o.uv_bounds = wf4{0, 0, 0, 0};
  o.tex_slot = WR_S_COLOR0;
  const bool affine = lw[1] == lw[0] && lw[2] == lw[0] && lw[3] == lw[0];
  o.kind = (affine && tex.format == WR_FMT_R8 && tex.ptr) ? WR_PK_BOX_SHADOW : WR_PK_UNSUPPORTED;
}

// composite.glsl:73-159
WR_DEVICE void wr_vs_composite(const WrDrawDesc& d, const uint8_t* arena, int inst, bool fast, WrVsOut& o) {
  wf4 aDeviceRect = wr_load_attr<wf4>(d, arena, inst, 0);
  wf4 aClip = wr_load_attr<wf4>(d, arena, inst, 1);
  wf4 aColor = wr_load_attr<wf4>(d, arena, inst, 2);
  wf4 aParams = wr_load_attr<wf4>(d, arena, inst, 3);
  wf4 aUv = wr_load_attr<wf4>(d, arena, inst, 4);
  wf2 aFlip = wr_load_attr<wf2>(d, arena, inst, 5);
  // mix(aDeviceRect.xyzw, aDeviceRect.zwxy, aFlip.xyxy)
  wf4 dr;
  dr.x = (aDeviceRect.z - aDeviceRect.x) * aFlip.x + aDeviceRect.x;
  dr.y = (aDeviceRect.w - aDeviceRect.y) * aFlip.y + aDeviceRect.y;
  dr.z = (aDeviceRect.x - aDeviceRect.z) * aFlip.x + aDeviceRect.z;
  dr.w = (aDeviceRect.y - aDeviceRect.w) * aFlip.y + aDeviceRect.w;
  wf4 bounds = {wr_min(aUv.x, aUv.z), wr_min(aUv.y, aUv.w), wr_max(aUv.x, aUv.z), wr_max(aUv.y, aUv.w)};
  bool unnorm = int(aParams.y) == 1;
#ifdef WRHIP_HOSTSIM
  if (getenv("WRHIP_DEBUG_VS")) fprintf(stderr, "   comp: rect %g %g %g %g clip %g %g %g %g flip %g %g dr %g %g %g %g\n", aDeviceRect.x, aDeviceRect.y, aDeviceRect.z, aDeviceRect.w, aClip.x, aClip.y, aClip.z, aClip.w, aFlip.x, aFlip.y, dr.x, dr.y, dr.z, dr.w);
#endif
  float tsx = 1.f, tsy = 1.f;
  if (unnorm) {
    tsx = d.tex[WR_S_COLOR0].ptr ? d.tex[WR_S_COLOR0].sw : 1.0f;        // (1 under TEXTURE_RECT, composite.glsl:136-147)
    tsy = d.tex[WR_S_COLOR0].ptr ? d.tex[WR_S_COLOR0].sh : 1.0f;
    bounds = wf4{(bounds.x + 0.5f) / tsx, (bounds.y + 0.5f) / tsy, (bounds.z + -0.5f) / tsx, (bounds.w + -0.5f) / tsy};
  }
  for (int n = 0; n < 4; n++) {
    float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    float wx = (dr.z - dr.x) * ax + dr.x, wy = (dr.w - dr.y) * ay + dr.y;
    float cx = wr_clamp(wx, aClip.x, aClip.z), cy = wr_clamp(wy, aClip.y, aClip.w);
    float ux = (cx - dr.x) / (dr.z - dr.x), uy = (cy - dr.y) / (dr.w - dr.y);
    ux = (aUv.z - aUv.x) * ux + aUv.x; uy = (aUv.w - aUv.y) * uy + aUv.y;
    if (unnorm) { ux = ux / tsx; uy = uy / tsy; }
    o.u[n] = ux; o.v[n] = uy;
    wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{cx, cy, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  o.kind = WR_PK_TEX_RGBA8;
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0;
  if (fast) {
    o.color = wf4{1.f, 1.f, 1.f, 1.f};
    o.has_color = 0;
    o.tail_clamp = 0; o.tail_modulate = 0;
    o.uv_bounds = wf4{0.f, 0.f, 1.f, 1.f};
    if (d.flags & WR_DF_TEX_RECT) o.uv_bounds = wf4{0.f, 0.f, float(d.tex[WR_S_COLOR0].width), float(d.tex[WR_S_COLOR0].height)};   // vec4(vec2(0), textureSize(sColor0)), composite.glsl:219-223
  } else {
    o.color = aColor;
    o.tail_clamp = 1; o.tail_modulate = 1;
    // swgl_drawSpanRGBA8: colour modulation only if colour != vec4(1.0) (composite.glsl:224-228)
    o.has_color = (aColor.x != 1.f || aColor.y != 1.f || aColor.z != 1.f || aColor.w != 1.f) ? 1 : 0;
    o.uv_bounds = bounds;
  }
}

// composite.glsl:75-128 with WR_FEATURE_YUV (vertex stage): the three uv rects are the planes' texel rects
WR_DEVICE void wr_vs_composite_yuv(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o, WrYuvRec& Y) {
  const wf4 aDeviceRect = wr_load_attr<wf4>(d, arena, inst, 0);
  const wf4 aClip = wr_load_attr<wf4>(d, arena, inst, 1);
  const wf4 aParams = wr_load_attr<wf4>(d, arena, inst, 3);
  const wf2 aFlip = wr_load_attr<wf2>(d, arena, inst, 7);
  const wf4 aUvR[3] = {wr_load_attr<wf4>(d, arena, inst, 4), wr_load_attr<wf4>(d, arena, inst, 5), wr_load_attr<wf4>(d, arena, inst, 6)};
  wf4 dr;
  dr.x = (aDeviceRect.z - aDeviceRect.x) * aFlip.x + aDeviceRect.x;
  dr.y = (aDeviceRect.w - aDeviceRect.y) * aFlip.y + aDeviceRect.y;
  dr.z = (aDeviceRect.x - aDeviceRect.z) * aFlip.x + aDeviceRect.z;
  dr.w = (aDeviceRect.y - aDeviceRect.w) * aFlip.y + aDeviceRect.w;
  const int color_space = int(aParams.y), format = int(aParams.z), depth = int(aParams.w);      // fetch_yuv_primitive (ExternalSurfaceDependency::Yuv)
  wr_yuv_setup(Y, depth, color_space, format);
  const int planes = format == 3 ? 3 : ((format == 0 || format == 1) ? 2 : (format == 4 ? 1 : 0));
  float fxv[4], fyv[4];
  for (int n = 0; n < 4; n++) {
    const float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    const float wx = (dr.z - dr.x) * ax + dr.x, wy = (dr.w - dr.y) * ay + dr.y;
    const float cx = wr_clamp(wx, aClip.x, aClip.z), cy = wr_clamp(wy, aClip.y, aClip.w);
    fxv[n] = (cx - dr.x) / (dr.z - dr.x); fyv[n] = (cy - dr.y) / (dr.w - dr.y);
    const wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{cx, cy, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.z; o.pw[n] = gp.w;
  }
  for (int pl = 0; pl < 3; pl++) {
    float* ou = pl == 0 ? o.u : (pl == 1 ? o.u2 : o.u3); float* ov = pl == 0 ? o.v : (pl == 1 ? o.v2 : o.v3);
    float bnd[4] = {0.f, 0.f, 0.f, 0.f};
    if (pl < planes) {
      const wf4 res = aUvR[pl];
      const WrTexDesc& tex = d.tex[WR_S_COLOR0 + pl];
      const float tsx = tex.ptr ? tex.sw : 1.0f, tsy = tex.ptr ? tex.sh : 1.0f;      // TEX_SIZE_YUV: 1 under TEXTURE_RECT (yuv.glsl:18-22)
      for (int n = 0; n < 4; n++) { ou[n] = ((res.z - res.x) * fxv[n] + res.x) / tsx; ov[n] = ((res.w - res.y) * fyv[n] + res.y) / tsy; }
      bnd[0] = (res.x + 0.5f) / tsx; bnd[1] = (res.y + 0.5f) / tsy; bnd[2] = (res.z - 0.5f) / tsx; bnd[3] = (res.w - 0.5f) / tsy;
    } else {
      for (int n = 0; n < 4; n++) { ou[n] = 0.0f; ov[n] = 0.0f; }
    }
    if (pl == 0) o.uv_bounds = wf4{bnd[0], bnd[1], bnd[2], bnd[3]};
    else for (int k = 0; k < 4; k++) (pl == 1 ? Y.u_bounds : Y.v_bounds)[k] = bnd[k];
  }
  o.tex_slot = WR_S_COLOR0;
  o.aa_edges = 0; o.has_mask = 0;
  o.tail_clamp = 1; o.tail_modulate = 0;
  o.has_color = 0; o.color = wf4{1.f, 1.f, 1.f, 1.f};
  o.kind = (planes > 0 && wr_yuv_planes_ok(d, format, depth) && !wr_yuv_rect_fast_path(d, format)) ? WR_PK_YUV : WR_PK_UNSUPPORTED;
}

// ps_clear.glsl:9-25
WR_DEVICE void wr_vs_ps_clear(const WrDrawDesc& d, const uint8_t* arena, int inst, WrVsOut& o) {
  wf4 aRect = wr_load_attr<wf4>(d, arena, inst, 0);
  wf4 aColor = wr_load_attr<wf4>(d, arena, inst, 1);
  for (int n = 0; n < 4; n++) {
    float ax = d.quad[2 * n], ay = d.quad[2 * n + 1];
    float x = (aRect.z - aRect.x) * ax + aRect.x, y = (aRect.w - aRect.y) * ay + aRect.y;
    wf4 gp = wr_mul(*(const WrMat4*)d.transform, wf4{x, y, 0.0f, 1.0f});
    o.px[n] = gp.x; o.py[n] = gp.y; o.pz[n] = gp.w /* gl_Position.z = gl_Position.w */; o.pw[n] = gp.w;
    o.u[n] = 0.f; o.v[n] = 0.f;
  }
  o.kind = WR_PK_SOLID;
  o.color = aColor;
  o.has_color = 0; o.aa_edges = 0; o.has_mask = 0;
}

// a[i] for a runtime i without making `a` addressable (a dynamically indexed local array, and
// the struct around it, would live in scratch memory)
WR_DEVICE float wr_pick4(const float (&a)[4], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
WR_DEVICE int wr_pick4i(const int (&a)[4], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
WR_DEVICE bool wr_isfinite(float x) { return (x - x) == 0.0f; }

// draw_quad_spans (rasterize.h:783-1055) for a general convex quad, walked at setup time: which vertex
// starts, which edges are the span's left and right ones, when each edge ends and is replaced
// (STEP_EDGE), the clip span of every edge pair.  Returns false for degenerate walks (nothing to draw)
// or more runs than WrQuadRec holds.  p[] in vertex-lane order (0,0) (1,0) (1,1) (0,1).
struct WrEdgeInst { float x, slope; int row, mask; float u, v, us, vs; float z, w, zs, ws; float u2, v2, u2s, v2s; };
WR_DEVICE WrEdgeInst wr_edge_init(float y, float p0x, float p0y, float p1x, float p1y, int mask, float u0, float v0, float u1,
                                  float v1, float z0, float w0, float z1, float w1, float a0 = 0.0f, float b0 = 0.0f, float a1 = 0.0f, float b1 = 0.0f) {   // Edge ctor, :850-876 (:1127-1152 for z, w)
  WrEdgeInst e;
  const float yScale = 1.0f / wr_max(p1y - p0y, 1.0f / 256.0f);
  e.slope = (p1x - p0x) * yScale;
  e.x = p0x + (y - p0y) * e.slope;
  e.us = (u1 - u0) * yScale; e.vs = (v1 - v0) * yScale;
  e.u = u0 + (y - p0y) * e.us; e.v = v0 + (y - p0y) * e.vs;
  e.zs = (z1 - z0) * yScale; e.ws = (w1 - w0) * yScale;
  e.z = z0 + (y - p0y) * e.zs; e.w = w0 + (y - p0y) * e.ws;
  e.u2s = (a1 - a0) * yScale; e.v2s = (b1 - b0) * yScale;          // (a second interpolated vec2: brush_mix_blend under perspective)
  e.u2 = a0 + (y - p0y) * e.u2s; e.v2 = b0 + (y - p0y) * e.v2s;
  e.row = int(y); e.mask = mask;
  return e;
}
// iu / iv: the shader's interpolated vec2 per vertex (zeros for solid prims)
// iz / iw: screen z and 1/w per vertex of a perspective quad (stored in Q.persp when `persp`)
WR_DEVICE bool wr_quad_walk(const float (&px)[4], const float (&py)[4], const float (&iu)[4], const float (&iv)[4], float cx0, float cy0,
                            float cx1, float cy1, bool aa, int aa_mask, WrQuadRec& Q, int& bx0, int& by0, int& bx1, int& by1,
                            const float (&iz)[4], const float (&iw)[4], bool persp, const float* iu2 = nullptr, const float* iv2 = nullptr) {
  Q.nseg = 0; Q.aa = aa ? 1 : 0; Q.rowtab = nullptr; Q.rowtab_rows = 0;
  // top-most point (:794-799)
  const int top = py[3] < py[2] ? (py[0] < py[1] ? (py[0] < py[3] ? 0 : 3) : (py[1] < py[3] ? 1 : 3))
                                : (py[0] < py[1] ? (py[0] < py[2] ? 0 : 2) : (py[1] < py[2] ? 1 : 2));
  const int next = (top + 1) & 3, prev = (top + 3) & 3;
  int l0i, l1i, r0i, r1i;
  if (wr_pick4(py, top) == wr_pick4(py, next)) { l0i = next; l1i = (next + 1) & 3; r0i = top; r1i = prev; }
  else if (wr_pick4(py, top) == wr_pick4(py, prev)) { l0i = top; l1i = next; r0i = prev; r1i = (prev + 3) & 3; }
  else { l0i = r0i = top; l1i = next; r1i = prev; }
#define WR_PX(i) wr_pick4(px, i)
#define WR_PY(i) wr_pick4(py, i)
  const float aaRound = aa ? 0.0f : 0.5f;
  float y = floorf(wr_max(wr_min(WR_PY(l0i), cy1), cy0) + aaRound) + 0.5f;
  // the l-chain walks forward through the points, the r-chain backward; `flipped` says which one is the span's left edge
#define WR_P2(arr, i) ((arr) ? ((i) == 0 ? (arr)[0] : ((i) == 1 ? (arr)[1] : ((i) == 2 ? (arr)[2] : (arr)[3]))) : 0.0f)
#define WR_EDGE(a, b, m) wr_edge_init(y, WR_PX(a), WR_PY(a), WR_PX(b), WR_PY(b), (aa_mask >> (m)) & 1, wr_pick4(iu, a), wr_pick4(iv, a), \
                                     wr_pick4(iu, b), wr_pick4(iv, b), wr_pick4(iz, a), wr_pick4(iw, a), wr_pick4(iz, b), wr_pick4(iw, b), \
                                     WR_P2(iu2, a), WR_P2(iv2, a), WR_P2(iu2, b), WR_P2(iv2, b))
  WrEdgeInst EL = WR_EDGE(l0i, l1i, l1i);
  WrEdgeInst ER = WR_EDGE(r0i, r1i, r0i);
  bool flipped;
  {   // checkIfEdgesFlipped (:766-774)
    const float l0x = WR_PX(l0i), r0x = WR_PX(r0i);
    const float ax = WR_PX(l1i) - l0x, ay = WR_PY(l1i) - WR_PY(l0i), bx = WR_PX(r1i) - r0x, by = WR_PY(r1i) - WR_PY(r0i);
    flipped = l0x > r0x || (l0x == r0x && (ax * by - ay * bx) > 0.0f);
  }
  float checkY = wr_min(wr_min(WR_PY(l1i), WR_PY(r1i)), cy1);
  float b0, b1;
#define WR_CLIPSPAN()                                                                                          \
  do {                                                                                                         \
    const float lo = wr_min(wr_min(WR_PX(l0i), WR_PX(l1i)), wr_min(WR_PX(r0i), WR_PX(r1i)));                   \
    const float hi = wr_max(wr_max(WR_PX(l0i), WR_PX(l1i)), wr_max(WR_PX(r0i), WR_PX(r1i)));                   \
    b0 = wr_clamp(lo, cx0, cx1); b1 = wr_clamp(hi, cx0, cx1);                                                  \
  } while (0)
  WR_CLIPSPAN();
  bx0 = 0x7FFFFFFF; bx1 = -0x7FFFFFFF; by0 = int(y); by1 = int(y);
  for (int guard = 0; guard < 16; guard++) {
    if (y > checkY) {
      if (y > cy1) break;
      bool done = false;
      if (y > WR_PY(l1i)) {          // STEP_EDGE(y, l0i, l0, l1i, l1, NEXT_POINT, r1i)
        do { l0i = l1i; l1i = (l1i + 1) & 3; if (l0i == r1i) { done = true; break; } } while (y > WR_PY(l1i));
        if (done) break;
        EL = WR_EDGE(l0i, l1i, l1i);
      }
      if (y > WR_PY(r1i)) {          // STEP_EDGE(y, r0i, r0, r1i, r1, PREV_POINT, l1i)
        do { r0i = r1i; r1i = (r1i + 3) & 3; if (r0i == l1i) { done = true; break; } } while (y > WR_PY(r1i));
        if (done) break;
        ER = WR_EDGE(r0i, r1i, r0i);
      }
      WR_CLIPSPAN();
      checkY = wr_min(ceilf(wr_min(WR_PY(l1i), WR_PY(r1i)) - aaRound), cy1);
    }
    // rows y, y + 1, ... up to the last one that does not exceed checkY (at least this one)
    int n = 1;
    if (checkY >= y) n = int(floor(double(checkY) - double(y))) + 1;
    if (Q.nseg >= 4) return false;
    WrQuadSeg& S = Q.seg[Q.nseg++];
    const WrEdgeInst& A = flipped ? ER : EL;
    const WrEdgeInst& B = flipped ? EL : ER;
    S.row_a = int(y); S.row_b = int(y) + n;
    S.lx = A.x; S.ls = A.slope; S.lrow = A.row; S.lmask = A.mask;
    S.rx = B.x; S.rs = B.slope; S.rrow = B.row; S.rmask = B.mask;
    S.luv[0] = A.u; S.luv[1] = A.v; S.luvs[0] = A.us; S.luvs[1] = A.vs;
    S.ruv[0] = B.u; S.ruv[1] = B.v; S.ruvs[0] = B.us; S.ruvs[1] = B.vs;
    if (persp) {
      WrPerspRec& R = Q.persp;
      const int k = Q.nseg - 1;
      R.lz[k] = A.z; R.lzs[k] = A.zs; R.lw[k] = A.w; R.lws[k] = A.ws;
      R.rz[k] = B.z; R.rzs[k] = B.zs; R.rw[k] = B.w; R.rws[k] = B.ws;
      if (iu2) {
        R.l2u[k] = A.u2; R.l2us[k] = A.u2s; R.l2v[k] = A.v2; R.l2vs[k] = A.v2s;
        R.r2u[k] = B.u2; R.r2us[k] = B.u2s; R.r2v[k] = B.v2; R.r2vs[k] = B.v2s;
      }
    }
    S.b0 = b0; S.b1 = b1;
    bx0 = wr_imin(bx0, int(floorf(b0)) - 1); bx1 = wr_imax(bx1, int(ceilf(b1)) + 1);
    by1 = S.row_b;
    y = y + float(n);
  }
#undef WR_CLIPSPAN
#undef WR_EDGE
#undef WR_P2
#undef WR_PX
#undef WR_PY
  return Q.nseg > 0;
}

// ---------------------------------------------------------------------------
// draw_perspective's clipped path (rasterize.h:1289-1430, 1490-1544): a quad with a vertex outside the near / far planes -- for
// WebRender's z = z_id * w that means a vertex at or behind the camera plane, w <= 0 -- is clipped against the view volume
// before it is projected: clip_side<Z>, and, if a clipped vertex still has w <= 0, clip_side<X> and clip_side<Y>; every pass
// can add two vertices (up to ten), and rewrites the AA edge mask.  Interpolants: the prim's one vec2 varying.
struct WrClipPt { float x, y, z, w, u, v, u2, v2; };      // (u2, v2: brush_mix_blend's second varying, clipped along with the first)
WR_DEVICE float wr_clip_sel(const WrClipPt& p, int axis) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }
WR_DEVICE WrClipPt wr_clip_lerp(const WrClipPt& a, const WrClipPt& b, float k) {       // prev + (cur - prev) * k, component by component
  WrClipPt r;
  r.x = a.x + (b.x - a.x) * k; r.y = a.y + (b.y - a.y) * k; r.z = a.z + (b.z - a.z) * k; r.w = a.w + (b.w - a.w) * k;
  r.u = a.u + (b.u - a.u) * k; r.v = a.v + (b.v - a.v) * k;
  r.u2 = a.u2 + (b.u2 - a.u2) * k; r.v2 = a.v2 + (b.v2 - a.v2) * k;
  return r;
}
__device__ __noinline__ int wr_clip_side(int axis, int nump, const WrClipPt* p, WrClipPt* out, int& edge_mask_io) {
  const int POSITIVE = 1, NEGATIVE = 2;
  int numClip = 0;
  int edgeMask = edge_mask_io;
  WrClipPt prev = p[nump - 1];
  float prevCoord = wr_clip_sel(prev, axis);
  int prevMask = (prevCoord < -prev.w ? NEGATIVE : 0) | (prevCoord > prev.w ? POSITIVE : 0);
  int outMask = 0;
  for (int i = 0; i < nump; i++, edgeMask >>= 1) {
    const WrClipPt cur = p[i];
    const float curCoord = wr_clip_sel(cur, axis);
    const int curMask = (curCoord < -cur.w ? NEGATIVE : 0) | (curCoord > cur.w ? POSITIVE : 0);
    if (!(curMask & prevMask)) {
      if (prevMask) {            // an edge that was outside crosses inside
        if (numClip >= nump + 2) return 0;
        const float prevSide = ((prevMask & NEGATIVE) && (!(prevMask & POSITIVE) || prevCoord * (cur.w - prev.w) < prev.w * (curCoord - prevCoord))) ? -1.0f : 1.0f;
        const float prevDist = prevCoord - prevSide * prev.w;
        const float curDist = curCoord - prevSide * cur.w;
        float k = prevDist / (prevDist - curDist);
        WrClipPt clipped = wr_clip_lerp(prev, cur, k);
        if (prevSide * wr_clip_sel(clipped, axis) > clipped.w) {      // (only the position is redone with the nudged weight)
          k = nextafterf(k, 1.0f);
          const WrClipPt c2 = wr_clip_lerp(prev, cur, k);
          clipped.x = c2.x; clipped.y = c2.y; clipped.z = c2.z; clipped.w = c2.w;
        }
        const WrClipPt ci = wr_clip_lerp(prev, cur, k);
        clipped.u = ci.u; clipped.v = ci.v; clipped.u2 = ci.u2; clipped.v2 = ci.v2;
        out[numClip] = clipped;
        numClip++;
      }
      if (curMask) {             // an edge that was inside crosses outside
        if (numClip >= nump + 2) return 0;
        const float curSide = ((curMask & POSITIVE) && (!(curMask & NEGATIVE) || prevCoord * (cur.w - prev.w) < prev.w * (curCoord - prevCoord))) ? 1.0f : -1.0f;
        const float prevDist = prevCoord - curSide * prev.w;
        const float curDist = curCoord - curSide * cur.w;
        float k = prevDist / (prevDist - curDist);
        WrClipPt clipped = wr_clip_lerp(prev, cur, k);
        if (curSide * wr_clip_sel(clipped, axis) > clipped.w) {
          k = nextafterf(k, 0.0f);
          const WrClipPt c2 = wr_clip_lerp(prev, cur, k);
          clipped.x = c2.x; clipped.y = c2.y; clipped.z = c2.z; clipped.w = c2.w;
        }
        const WrClipPt ci = wr_clip_lerp(prev, cur, k);
        clipped.u = ci.u; clipped.v = ci.v; clipped.u2 = ci.u2; clipped.v2 = ci.v2;
        out[numClip] = clipped;
        outMask |= (edgeMask & 1) << numClip;
        numClip++;
      }
    }
    if (!curMask) {
      if (numClip >= nump + 2) return 0;
      out[numClip] = cur;
      outMask |= (edgeMask & 1) << numClip;
      numClip++;
    }
    prev = cur; prevCoord = curCoord; prevMask = curMask;
  }
  edge_mask_io = outMask;
  return numClip;
}

// draw_perspective_spans (rasterize.h:1064-1280) for the clipped polygon, walked at setup time like wr_quad_walk: the start vertices
// of the two descending chains (:1070-1105, flat tops included), STEP_EDGE, the clip span of every edge pair.  p*: screen x, y,
// z and 1 / w per vertex; iu / iv: the varying times 1 / w.
WR_DEVICE bool wr_poly_walk(const int nump, const float* px, const float* py, const float* iu, const float* iv, const float* iz,
                                          const float* iw, float cx0, float cy0, float cx1, float cy1, bool aa, int aa_mask, WrQuadRec& Q, int& bx0,
                                          int& by0, int& bx1, int& by1, const float* iu2 = nullptr, const float* iv2 = nullptr) {
  Q.nseg = 0; Q.aa = aa ? 1 : 0; Q.rowtab = nullptr; Q.rowtab_rows = 0;
  auto NEXT = [&](int i) { return i + 1 == nump ? 0 : i + 1; };
  auto PREV = [&](int i) { return i == 0 ? nump - 1 : i - 1; };
  int top = 0;
  for (int i = 1; i < nump; i++) if (py[i] < py[top]) top = i;
  int l0i = top;
  for (int i = top + 1; i < nump && py[i] == py[top]; i++) l0i = i;
  if (l0i == nump - 1) for (int i = 0; i <= top && py[i] == py[top]; i++) l0i = i;
  int r0i = top;
  for (int i = top - 1; i >= 0 && py[i] == py[top]; i--) r0i = i;
  if (r0i == 0) for (int i = nump - 1; i >= top && py[i] == py[top]; i--) r0i = i;
  int l1i = NEXT(l0i), r1i = PREV(r0i);
  const float aaRound = aa ? 0.0f : 0.5f;
  float y = floorf(wr_max(wr_min(py[l0i], cy1), cy0) + aaRound) + 0.5f;
#define WR_PEDGE(a, b, m) wr_edge_init(y, px[a], py[a], px[b], py[b], (aa_mask >> (m)) & 1, iu[a], iv[a], iu[b], iv[b], iz[a], iw[a], iz[b], iw[b], \
                                        iu2 ? iu2[a] : 0.0f, iv2 ? iv2[a] : 0.0f, iu2 ? iu2[b] : 0.0f, iv2 ? iv2[b] : 0.0f)
  WrEdgeInst EL = WR_PEDGE(l0i, l1i, l1i);
  WrEdgeInst ER = WR_PEDGE(r0i, r1i, r0i);
  bool flipped;
  {   // checkIfEdgesFlipped (:766-774)
    const float l0x = px[l0i], r0x = px[r0i];
    const float ax = px[l1i] - l0x, ay = py[l1i] - py[l0i], bx = px[r1i] - r0x, by = py[r1i] - py[r0i];
    flipped = l0x > r0x || (l0x == r0x && (ax * by - ay * bx) > 0.0f);
  }
  float checkY = wr_min(wr_min(py[l1i], py[r1i]), cy1);
  float b0, b1;
#define WR_PCLIPSPAN()                                                                                 \
  do {                                                                                                 \
    const float lo = wr_min(wr_min(px[l0i], px[l1i]), wr_min(px[r0i], px[r1i]));                       \
    const float hi = wr_max(wr_max(px[l0i], px[l1i]), wr_max(px[r0i], px[r1i]));                       \
    b0 = wr_clamp(lo, cx0, cx1); b1 = wr_clamp(hi, cx0, cx1);                                          \
  } while (0)
  WR_PCLIPSPAN();
  bx0 = 0x7FFFFFFF; bx1 = -0x7FFFFFFF; by0 = int(y); by1 = int(y);
  for (int guard = 0; guard < 4 * WR_MAX_QSEG; guard++) {
    if (y > checkY) {
      if (y > cy1) break;
      bool done = false;
      if (y > py[l1i]) {          // STEP_EDGE(y, l0i, l0, l1i, l1, NEXT_POINT, r1i)
        do { l0i = l1i; l1i = NEXT(l1i); if (l0i == r1i) { done = true; break; } } while (y > py[l1i]);
        if (done) break;
        EL = WR_PEDGE(l0i, l1i, l1i);
      }
      if (y > py[r1i]) {          // STEP_EDGE(y, r0i, r0, r1i, r1, PREV_POINT, l1i)
        do { r0i = r1i; r1i = PREV(r1i); if (r0i == l1i) { done = true; break; } } while (y > py[r1i]);
        if (done) break;
        ER = WR_PEDGE(r0i, r1i, r0i);
      }
      WR_PCLIPSPAN();
      checkY = wr_min(ceilf(wr_min(py[l1i], py[r1i]) - aaRound), cy1);
    }
    int n = 1;
    if (checkY >= y) n = int(floor(double(checkY) - double(y))) + 1;
    if (Q.nseg >= WR_MAX_QSEG) return false;
    WrQuadSeg& S = Q.seg[Q.nseg++];
    const WrEdgeInst& A = flipped ? ER : EL;
    const WrEdgeInst& B = flipped ? EL : ER;
    S.row_a = int(y); S.row_b = int(y) + n;
    S.lx = A.x; S.ls = A.slope; S.lrow = A.row; S.lmask = A.mask;
    S.rx = B.x; S.rs = B.slope; S.rrow = B.row; S.rmask = B.mask;
    S.luv[0] = A.u; S.luv[1] = A.v; S.luvs[0] = A.us; S.luvs[1] = A.vs;
    S.ruv[0] = B.u; S.ruv[1] = B.v; S.ruvs[0] = B.us; S.ruvs[1] = B.vs;
    {
      WrPerspRec& R = Q.persp;
      const int k = Q.nseg - 1;
      R.lz[k] = A.z; R.lzs[k] = A.zs; R.lw[k] = A.w; R.lws[k] = A.ws;
      R.rz[k] = B.z; R.rzs[k] = B.zs; R.rw[k] = B.w; R.rws[k] = B.ws;
      if (iu2) {
        R.l2u[k] = A.u2; R.l2us[k] = A.u2s; R.l2v[k] = A.v2; R.l2vs[k] = A.v2s;
        R.r2u[k] = B.u2; R.r2us[k] = B.u2s; R.r2v[k] = B.v2; R.r2vs[k] = B.v2s;
      }
    }
    S.b0 = b0; S.b1 = b1;
    bx0 = wr_imin(bx0, int(floorf(b0)) - 1); bx1 = wr_imax(bx1, int(ceilf(b1)) + 1);
    by1 = S.row_b;
    y = y + float(n);
  }
#undef WR_PCLIPSPAN
#undef WR_PEDGE
  return Q.nseg > 0;
}

// What the clipped walk needs of a prim, parked in the tail of its (still unused) quad record by the vertex-stage thread: the
// walk runs after wr_finish_prim, when the vertex stage's outputs are dead -- called from inside it, the callee's registers
// came on top of the ~60 live ones and the fused setup + tile-pass kernel lost a wave per SIMD.
struct WrClipStash { float px[4], py[4], pz[4], pw[4], u[4], v[4]; float cx0, cy0, cx1, cy1; int32_t aa, aa_edges; float vp[4]; float u2[4], v2[4]; int32_t two; };      // two: u2 / v2 hold a second varying (brush_mix_blend)
static_assert(sizeof(WrClipStash) <= 5 * sizeof(WrQuadSeg), "the stash sits in seg[5..9]: the walk has copied it before it writes that far");
WR_DEVICE WrClipStash* wr_clip_stash(WrQuadRec& Q) { return (WrClipStash*)&Q.seg[WR_MAX_QSEG - 5]; }
// clip, project (rasterize.h:1518-1530), ClipRect::overlaps, and walk.  False: nothing to draw.
WR_DEVICE bool wr_persp_clipped_walk(WrQuadRec& Q, int& bx0, int& by0, int& bx1, int& by1, float& ocx0, float& ocy0, float& ocx1, float& ocy1) {
  const WrClipStash o = *wr_clip_stash(Q);
  const float cx0 = o.cx0, cy0 = o.cy0, cx1 = o.cx1, cy1 = o.cy1;
  ocx0 = cx0; ocy0 = cy0; ocx1 = cx1; ocy1 = cy1;
  const bool aa = o.aa != 0;
  WrClipPt a[WR_MAX_QSEG], b[WR_MAX_QSEG];
  for (int n = 0; n < 4; n++) { a[n].x = o.px[n]; a[n].y = o.py[n]; a[n].z = o.pz[n]; a[n].w = o.pw[n]; a[n].u = o.u[n]; a[n].v = o.v[n]; a[n].u2 = o.two ? o.u2[n] : 0.0f; a[n].v2 = o.two ? o.v2[n] : 0.0f; }
  int mask = o.aa_edges;
  int nump = wr_clip_side(2, 4, a, b, mask);
#ifdef WRHIP_HOSTSIM
  if (getenv("WRHIP_DEBUG_CLIP")) fprintf(stderr, "clip_side<Z>: %d points, w %g %g %g %g\n", nump, o.pw[0], o.pw[1], o.pw[2], o.pw[3]);
#endif
  if (nump < 3) return false;
  bool behind = false;
  for (int i = 0; i < nump; i++) if (b[i].w <= 0.0f) { behind = true; break; }
  WrClipPt* cur = b;
  if (behind) {
    nump = wr_clip_side(0, nump, b, a, mask);
    if (nump < 3) return false;
    nump = wr_clip_side(1, nump, a, b, mask);
#ifdef WRHIP_HOSTSIM
    if (getenv("WRHIP_DEBUG_CLIP")) fprintf(stderr, "  after X / Y: %d points\n", nump);
#endif
    if (nump < 3) return false;
  }
  const float scx = o.vp[2] * 0.5f, scy = o.vp[3] * 0.5f;
  const float ofx = o.vp[0] + scx, ofy = o.vp[1] + scy;
  float px[WR_MAX_QSEG], py[WR_MAX_QSEG], pz[WR_MAX_QSEG], pw[WR_MAX_QSEG], qu[WR_MAX_QSEG], qv[WR_MAX_QSEG], qu2[WR_MAX_QSEG], qv2[WR_MAX_QSEG];
  int sides = 0;
  for (int i = 0; i < nump; i++) {
    const float wn = 1.0f / cur[i].w;
    if (wr_isfinite(wn)) { px[i] = cur[i].x * wn * scx + ofx; py[i] = cur[i].y * wn * scy + ofy; pz[i] = cur[i].z * wn * 0.5f + 0.5f; pw[i] = wn; }
    else { px[i] = py[i] = pz[i] = pw[i] = 0.0f; }
    qu[i] = cur[i].u * pw[i]; qv[i] = cur[i].v * pw[i];
    qu2[i] = cur[i].u2 * pw[i]; qv2[i] = cur[i].v2 * pw[i];
    sides |= px[i] < cx1 ? (px[i] > cx0 ? 3 : 1) : 2;
    sides |= py[i] < cy1 ? (py[i] > cy0 ? 12 : 4) : 8;
  }
  if (sides != 0xF) return false;
  return wr_poly_walk(nump, px, py, qu, qv, pz, pw, cx0, cy0, cx1, cy1, aa, mask, Q, bx0, by0, bx1, by1, o.two ? qu2 : nullptr, o.two ? qv2 : nullptr);
}

// The span of row y of a general quad (aa_span, rasterize.h:520-561): [s0, s1) -- with swgl_antiAlias the rounded-out one,
// [la0, ra1).  False: the row is outside the walk.
// x of both edges of run S (the LAST run of the walk that holds row y) on row y: Edge::nextRow's row-by-row sums -- read from the
// prim's row table where the setup stage wrote one (WrQuadRec::rowtab)
WR_DEVICE const float* wr_quad_rowtab_entry(const WrQuadRec& Q, int y) {
  return (Q.rowtab && (unsigned)(y - Q.rowtab_y0) < (unsigned)Q.rowtab_rows) ? Q.rowtab + (size_t)(y - Q.rowtab_y0) * (size_t)Q.rowtab_stride : nullptr;
}
// `nwords` 4-byte words of the flush's pool (WrTargetDesc::qtab: row tables of general quads, and the depth runs / occluder lists that
// outgrow their LDS copies); nullptr: none left (the caller reports what it then cannot draw exactly)
WR_DEVICE int32_t* wr_pool_words(const WrTargetDesc& T, unsigned long long nwords) {
  if (!T.qtab || !T.qtab_ctl) return nullptr;
  nwords = (nwords + 3ull) & ~3ull;          // (every piece of the pool starts on 16 bytes)
  const unsigned long long off = atomicAdd(T.qtab_ctl, nwords);
  return off + nwords <= (unsigned long long)T.qtab_cap ? (int32_t*)(T.qtab + off) : nullptr;
}
// ... one allocation for the whole wave (every lane asks for the same words: the tile rows' sweeps are wave-uniform)
WR_DEVICE int32_t* wr_pool_words_wave(const WrTargetDesc& T, unsigned long long nwords) {
#ifdef WRHIP_HOSTSIM
  return wr_pool_words(T, nwords);
#else
  unsigned long long a = 0;
  if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(1))) a = (unsigned long long)(uintptr_t)wr_pool_words(T, nwords);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return (int32_t*)(uintptr_t)((unsigned long long)lo | ((unsigned long long)hi << 32));
#endif
}
WR_DEVICE void wr_quad_row_x(const WrQuadRec& Q, const WrQuadSeg& S, int y, float& xl, float& xr) {
  if (const float* e = wr_quad_rowtab_entry(Q, y)) { xl = e[0]; xr = e[1]; return; }
  xl = wr_accum(S.lx, S.ls, y - S.lrow); xr = wr_accum(S.rx, S.rs, y - S.rrow);
}
// The setup stage's half: every edge value of every row of the prim's box, summed ONCE, row by row as Edge::nextRow does (the first
// row of a run through wr_accum, then one add per row and value: by wr_accum's contract the same numbers the per-row calls return).
// What this replaces: in the raster stage every lane evaluates its own rows, i.e. one wr_accum per value, lane-row and prim -- for a
// full-size rotated rect 1448 rows x 16 lanes across x 2 values of a walk over binades each (wrench transforms-simple: 108 us of raster
// for eleven such rects).  A prim whose table does not fit the flush's pool keeps the per-row sums.
#define WR_QTAB_MIN_ROWS 16
WR_DEVICE void wr_quad_build_rowtab(const WrTargetDesc* Tp, const WrPrim* Pp, WrQuadRec* Qp) {
  const WrTargetDesc& T = *Tp; const WrPrim& P = *Pp; WrQuadRec& Q = *Qp;
  Q.rowtab = nullptr; Q.rowtab_rows = 0; Q.rowtab_stride = 0; Q.rowtab_y0 = 0; Q.rowtab_pad = 0;
  const int rows = P.y1 - P.y0;
  if (!T.qtab || !T.qtab_ctl || (T.qtab_pad & 1u) || rows < WR_QTAB_MIN_ROWS || Q.nseg <= 0) return;
  const bool zw = Q.pad != 0 || (P.kind == WR_PK_TEX_QUAD && Q.base_kind == WR_PK_MIX_BLEND);
  const int stride = zw ? 10 : (P.kind == WR_PK_TEX_QUAD ? 6 : 2);
  float* tab = (float*)wr_pool_words(T, (unsigned long long)rows * (unsigned long long)stride);
  if (!tab) return;
  for (int i = 0; i < Q.nseg; i++) {          // (in order: where two runs hold a row, the later one is the one the raster stage picks)
    const WrQuadSeg& S = Q.seg[i];
    const int ya = wr_imax(S.row_a, P.y0), yb = wr_imin(S.row_b, P.y1);
    if (yb <= ya) continue;
    float xl = wr_accum(S.lx, S.ls, ya - S.lrow), xr = wr_accum(S.rx, S.rs, ya - S.rrow);
    float lu = 0.0f, lv = 0.0f, ru = 0.0f, rv = 0.0f, wl = 0.0f, wr = 0.0f, zl = 0.0f, zr = 0.0f;
    if (stride >= 6) {
      lu = wr_accum(S.luv[0], S.luvs[0], ya - S.lrow); lv = wr_accum(S.luv[1], S.luvs[1], ya - S.lrow);
      ru = wr_accum(S.ruv[0], S.ruvs[0], ya - S.rrow); rv = wr_accum(S.ruv[1], S.ruvs[1], ya - S.rrow);
    }
    if (stride >= 10) {
      wl = wr_accum(Q.persp.lw[i], Q.persp.lws[i], ya - S.lrow); wr = wr_accum(Q.persp.rw[i], Q.persp.rws[i], ya - S.rrow);
      zl = wr_accum(Q.persp.lz[i], Q.persp.lzs[i], ya - S.lrow); zr = wr_accum(Q.persp.rz[i], Q.persp.rzs[i], ya - S.rrow);
    }
    // (the slopes in registers: the table's stores may alias the record as far as the compiler knows, and a reload per row and value is a
    // trip to memory on the one thread that walks the prim's rows -- 75 us for the 512 rows of a tile-high rect, measured)
    const float ls = S.ls, rs = S.rs, lus = S.luvs[0], lvs = S.luvs[1], rus = S.ruvs[0], rvs = S.ruvs[1];
    const float lws = stride >= 10 ? Q.persp.lws[i] : 0.0f, rws = stride >= 10 ? Q.persp.rws[i] : 0.0f;
    const float lzs = stride >= 10 ? Q.persp.lzs[i] : 0.0f, rzs = stride >= 10 ? Q.persp.rzs[i] : 0.0f;
    float* __restrict__ e = tab + (size_t)(ya - P.y0) * (size_t)stride;
    if (stride == 2) {
      // (two rows per 16-byte store where the pair is aligned: the one thread that walks a prim's rows is bound by the stores it may have in flight)
      int y = ya;
      if (((y - P.y0) & 1) && y < yb) { e[0] = xl; e[1] = xr; xl = xl + ls; xr = xr + rs; y++; e += 2; }
      for (; y + 2 <= yb; y += 2, e += 4) {
        const float xl1 = xl + ls, xr1 = xr + rs;
        wr_store16(e, xl, xr, xl1, xr1);
        xl = xl1 + ls; xr = xr1 + rs;
      }
      if (y < yb) { e[0] = xl; e[1] = xr; }
    } else {
      for (int y = ya; y < yb; y++, e += stride) {
        e[0] = xl; e[1] = xr; e[2] = lu; e[3] = lv; e[4] = ru; e[5] = rv;
        xl = xl + ls; xr = xr + rs; lu = lu + lus; lv = lv + lvs; ru = ru + rus; rv = rv + rvs;
        if (stride >= 10) {
          e[6] = wl; e[7] = wr; e[8] = zl; e[9] = zr;
          wl = wl + lws; wr = wr + rws; zl = zl + lzs; zr = zr + rzs;
        }
      }
    }
  }
  Q.rowtab_y0 = P.y0; Q.rowtab_rows = rows; Q.rowtab_stride = stride;
  Q.rowtab = tab;
#ifdef WRHIP_HOSTSIM
  { static const bool dbg = getenv("WRHIP_DEBUG_QTAB") != nullptr; if (dbg) fprintf(stderr, "row table: kind %d rows %d stride %d at word %lld of %u\n", (int)P.kind, rows, stride, (long long)(tab - T.qtab), T.qtab_cap); }
#endif
}
WR_DEVICE bool wr_quad_row_span(const WrQuadRec& Q, int y, int& s0, int& s1) {
  int si = -1;
  for (int i = 0; i < Q.nseg; i++) if (y >= Q.seg[i].row_a && y < Q.seg[i].row_b) si = i;
  if (si < 0) return false;
  const WrQuadSeg& S = Q.seg[si];
  float xl, xr;
  wr_quad_row_x(Q, S, y, xl, xr);
  if (!Q.aa) {
    s0 = int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f)); s1 = int(floorf(wr_clamp(xr, S.b0, S.b1) + 0.5f));
  } else {
    const float radl = 0.5f * fabsf(S.ls), radr = 0.5f * fabsf(S.rs);
    s0 = S.lmask ? int(floorf(wr_clamp(xl - radl, S.b0, S.b1))) : int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f));
    s1 = S.rmask ? int(ceilf(wr_clamp(xr + radr, S.b0, S.b1))) : int(floorf(wr_clamp(xr, S.b0, S.b1) + 0.5f));
  }
  return true;
}

// draw_quad (rasterize.h:1549-1633) + the axis-aligned closed form of
// draw_quad_spans (rasterize.h:783-1055): for a rectangle both edge slopes are
// exactly 0, so every row has the same span and rows are those whose centre
// lies in [top, bottom] after clipping.
WR_DEVICE void wr_finish_prim(const WrDrawDesc& d, int draw_index, const WrVsOut& o, WrPrim& P, WrAux* auxp,
                              WrUnsupportedCounters* cnt) {
  P.kind = WR_PK_NONE;
  P.draw = draw_index;
  P.color[0] = P.color[1] = 0; P.z = 0; P.tex_slot = 0;
  P.uv_add[0] = o.uv_add[0]; P.uv_add[1] = o.uv_add[1];
  P.dual = o.dual; P.dual_swz = o.dual_swz;
  P.blend = (int16_t)d.blend;
  P.flags = d.flags & (WR_PF_DEPTH_TEST | WR_PF_DEPTH_WRITE | WR_PF_DEPTH_LESS);
  P.x0 = P.y0 = P.x1 = P.y1 = 0;
  // draw_perspective (rasterize.h:1449-1547): any vertex w different from the others.  Implemented for solid colours
  // (no varyings to divide by w) whose vertices lie between the near and far planes; clipping against the view volume
  // (clip_side, :1286-1430), perspective-correct varyings and depth-WRITING perspective prims are counted and skipped.
  const bool persp = o.pw[1] != o.pw[0] || o.pw[2] != o.pw[0] || o.pw[3] != o.pw[0];
  float w = 1.0f / o.pw[0];
  if (!wr_isfinite(w)) w = 0.0f;
  float sx[4], sy[4], pz3[4], pw3[4];
  bool clipped = false;
  if (persp) {
    bool inside = true;
    for (int n = 0; n < 4; n++) inside = inside && (o.pz[n] > -o.pw[n]) && (o.pz[n] < o.pw[n]);
    // (textures: ps_quad_textured and the plain brush_image keys, whose main() is restated with its perspective inputs)
    // (and brush_opacity, brush_blend, brush_linear_gradient: main() on the perspective-correct varying)
    const bool ptex = ((d.shader == WR_SH_PS_QUAD_TEXTURED || o.persp_div >= 0.0f) && (o.kind == WR_PK_TEX_RGBA8 || o.kind == WR_PK_TEX_FS)) ||
                      ((d.shader == WR_SH_PS_TEXT_RUN || d.shader == WR_SH_PS_TEXT_RUN_DUAL) && (o.kind == WR_PK_TEX_R8 || o.kind == WR_PK_TEX_RGBA8)) ||      /* (the GLYPH_TRANSFORM keys never reach here with a projective transform: their vertex stage reports it) */
                      o.kind == WR_PK_FILTER || o.kind == WR_PK_MIX_BLEND || (o.kind == WR_PK_QUAD_MASK && auxp->clip.w == 1.0f) || (o.kind == WR_PK_TEX_REPEAT && o.persp_div >= 0.0f) || (o.kind == WR_PK_GRADIENT && (d.shader == WR_SH_BRUSH_LINEAR_GRADIENT || d.shader == WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA ||
                                                                              d.shader == WR_SH_PS_QUAD_RADIAL_GRADIENT || d.shader == WR_SH_PS_QUAD_CONIC_GRADIENT));
    if (!(o.kind == WR_PK_SOLID || ptex)) { atomicAdd(&cnt->perspective_prims, 1u); return; }
    clipped = !inside;           // a vertex outside the near / far planes: clip_side first (wr_persp_clipped_walk)
    if (clipped) {
      // (the clip-space vertices are parked right away: kept in registers until the walk is set up below, they cost the fused
      // setup + tile-pass kernel a wave per SIMD; the stash lies beyond any base kind's side record in the prim's WrAux)
      WrClipStash& St = *wr_clip_stash(auxp->quad);
      for (int n = 0; n < 4; n++) { St.px[n] = o.px[n]; St.py[n] = o.py[n]; St.pz[n] = o.pz[n]; St.pw[n] = o.pw[n]; St.u[n] = o.u[n]; St.v[n] = o.v[n]; }
      St.two = o.kind == WR_PK_MIX_BLEND ? 1 : 0;
      if (St.two) for (int n = 0; n < 4; n++) { St.u2[n] = o.u2[n]; St.v2[n] = o.v2[n]; }
    }
    // screen = pos.xyz * (1 / pos.w) * scale + offset, scale = (viewport size, 1) / 2, offset = (viewport origin, 0) + scale
    const float scx = d.vp_size[0] * 0.5f, scy = d.vp_size[1] * 0.5f;
    const float ofx = d.vp_origin[0] + scx, ofy = d.vp_origin[1] + scy;
    for (int n = 0; n < 4; n++) {
      const float wn = clipped ? 0.0f : 1.0f / o.pw[n];
      sx[n] = o.px[n] * wn * scx + ofx;
      sy[n] = o.py[n] * wn * scy + ofy;
      pz3[n] = o.pz[n] * wn * 0.5f + 0.5f;
      pw3[n] = wn;
    }
  } else {
    for (int n = 0; n < 4; n++) {
      sx[n] = (o.px[n] * w + 1.0f) * 0.5f * d.vp_size[0] + d.vp_origin[0];
      sy[n] = (o.py[n] * w + 1.0f) * 0.5f * d.vp_size[1] + d.vp_origin[1];
      pz3[n] = 0.f; pw3[n] = 0.f;
    }
  }
  float cx0 = float(d.clip[0]), cy0 = float(d.clip[1]), cx1 = float(d.clip[2]), cy1 = float(d.clip[3]);
  bool masked = false;
  if (o.has_mask && d.blend != WR_BLEND_NONE) {
    // ClipRect ctor (rasterize.h:408-444): clip-mask bounds constrain the draw rect
    const WrTexDesc& mt = d.tex[WR_S_CLIP_MASK];
    if (mt.format != WR_FMT_R8 || !mt.ptr || o.kind == WR_PK_BLUR || o.kind == WR_PK_CLIP_RECT || o.kind == WR_PK_BOX_SHADOW) {
      P.kind = WR_PK_UNSUPPORTED; atomicAdd(&cnt->unsupported_prims, 1u); return;
    }
    int bx0 = int(o.mask_bb[0]), by0 = int(o.mask_bb[1]);
    int bx1 = bx0 + int(o.mask_bb[2]), by1 = by0 + int(o.mask_bb[3]);
    bx0 = wr_imax(bx0, 0); by0 = wr_imax(by0, 0); bx1 = wr_imin(bx1, mt.width); by1 = wr_imin(by1, mt.height);
    const int offx = int(o.mask_offset[0]) + int(d.vp_origin[0]), offy = int(o.mask_offset[1]) + int(d.vp_origin[1]);
    cx0 = wr_max(cx0, float(bx0 + offx)); cy0 = wr_max(cy0, float(by0 + offy));
    cx1 = wr_min(cx1, float(bx1 + offx)); cy1 = wr_min(cy1, float(by1 + offy));
    P.mask_off[0] = offx; P.mask_off[1] = offy;
    masked = true;
  }
#ifdef WRHIP_HOSTSIM
  if (getenv("WRHIP_DEBUG_VS"))
    fprintf(stderr, "    vs: pos (%g %g %g %g) (%g %g) (%g %g) (%g %g) screen (%g %g) (%g %g) (%g %g) (%g %g)\n", o.px[0], o.py[0], o.pz[0],
            o.pw[0], o.px[1], o.py[1], o.px[2], o.py[2], o.px[3], o.py[3], sx[0], sy[0], sx[1], sy[1], sx[2], sy[2], sx[3], sy[3]);
#endif
  // ClipRect::overlaps, rasterize.h:465-477
  int sides = 0;
  for (int n = 0; n < 4; n++) {
    sides |= sx[n] < cx1 ? (sx[n] > cx0 ? 3 : 1) : 2;
    sides |= sy[n] < cy1 ? (sy[n] > cy0 ? 12 : 4) : 8;
  }
  if (sides != 0xF && !clipped) return;
  if (!persp) {
    float screenZ = (o.pz[0] * w + 1.0f) * 0.5f;
    if (screenZ < 0.0f || screenZ > 1.0f) return;
    P.z = uint32_t(16777215.0f * screenZ);
  }     // (perspective: depth is per pixel, P.z = 0 keeps the strip depth cap from ever rejecting the prim)

  // swgl_antiAlias only takes effect when blending is on (ClipRect ctor, rasterize.h:414-441)
  const bool aa = o.aa_edges != 0 && d.blend != WR_BLEND_NONE;
  // textured kinds that can ride on WrQuadRec (general quads, swgl_antiAlias) when the host gave the launch the path for it
  // (the dual-source programs have no general-quad / anti-aliased path: reported)
  // (... except the REPETITION key on anti-aliased prims: its main() has one path, restated on the general-quad evaluator)
  if (o.dual && ((o.aa_edges != 0 && d.blend != WR_BLEND_NONE) || persp) && o.kind != WR_PK_TEX_REPEAT) { P.kind = WR_PK_UNSUPPORTED; atomicAdd(&cnt->unsupported_prims, 1u); return; }
  const bool texq = (d.flags & WR_DF_QUADS) &&
                    (o.kind == WR_PK_TEX_RGBA8 || o.kind == WR_PK_TEX_FS || o.kind == WR_PK_TEX_R8 || o.kind == WR_PK_TEX_REPEAT ||
                     o.kind == WR_PK_GRADIENT || o.kind == WR_PK_FILTER || o.kind == WR_PK_QUAD_MASK || (o.kind == WR_PK_SOLID && masked) ||
                     o.kind == WR_PK_MIX_BLEND);
  if (aa && (o.kind != WR_PK_SOLID || masked) && !texq) {      // AA on masked solids / other shader families: "next"
    P.kind = WR_PK_UNSUPPORTED; atomicAdd(&cnt->unsupported_prims, 1u); return;
  }
  // lanes: 0=(0,0) 1=(1,0) 2=(1,1) 3=(0,1) of the unit quad
  bool typeA = sy[0] == sy[1] && sy[2] == sy[3] && sx[0] == sx[3] && sx[1] == sx[2];
  bool typeB = sx[0] == sx[1] && sx[2] == sx[3] && sy[0] == sy[3] && sy[1] == sy[2];
  if ((!typeA && !typeB) || (aa && texq) || persp) {
    // general convex quad (rotation / skew), or an anti-aliased textured one: the scanline walk is done here, per prim
    const bool solidq = o.kind == WR_PK_SOLID && !masked && !(d.flags & WR_DF_SIMPLE);
    if (persp && !solidq && !(texq && (o.kind == WR_PK_TEX_RGBA8 || o.kind == WR_PK_TEX_FS || o.kind == WR_PK_TEX_R8 || o.kind == WR_PK_TEX_REPEAT || o.kind == WR_PK_FILTER || o.kind == WR_PK_GRADIENT || o.kind == WR_PK_QUAD_MASK || (o.kind == WR_PK_SOLID && masked) || o.kind == WR_PK_MIX_BLEND))) {
      atomicAdd(&cnt->perspective_prims, 1u); return;
    }
    if (!solidq && !texq) {
      P.kind = WR_PK_UNSUPPORTED; atomicAdd(&cnt->unsupported_prims, 1u); return;
    }
    // the base kind's side record: the vertex stage left it in the (shared) WrAux slot the quad record is about to take
    union { WrRepeatRec rep; WrGradRec grad; WrFilterRec filt; WrClipRec clip; WrMixRec mix; } base;
    if (o.kind == WR_PK_MIX_BLEND) base.mix = auxp->mix;
    if (o.kind == WR_PK_TEX_REPEAT) base.rep = auxp->rep;
    else if (o.kind == WR_PK_GRADIENT) base.grad = auxp->grad;
    else if (o.kind == WR_PK_FILTER) base.filt = auxp->filt;
    else if (o.kind == WR_PK_QUAD_MASK) base.clip = auxp->clip;
    int bx0, by0, bx1, by1;
    // (perspective: the edges also carry screen z and 1/w -- Point3D edges step exactly like interpolants, rasterize.h:1127-1152 --
    // the interpolants are pre-multiplied by the vertex's 1/w (:1138-1143), and draw_perspective_spans picks the same start
    // vertex and edges as draw_quad_spans for any quad without three vertices on one row)
    float qu[4], qv[4];
    for (int n = 0; n < 4; n++) { qu[n] = persp ? o.u[n] * pw3[n] : o.u[n]; qv[n] = persp ? o.v[n] * pw3[n] : o.v[n]; }
    if (clipped) {
      // (clipped against the view volume and walked after this function: wr_finish_clipped)
      WrClipStash& St = *wr_clip_stash(auxp->quad);
      St.cx0 = cx0; St.cy0 = cy0; St.cx1 = cx1; St.cy1 = cy1; St.aa = aa ? 1 : 0; St.aa_edges = o.aa_edges;
      St.vp[0] = d.vp_origin[0]; St.vp[1] = d.vp_origin[1]; St.vp[2] = d.vp_size[0]; St.vp[3] = d.vp_size[1];
      auxp->quad.nseg = -1; auxp->quad.rowtab = nullptr; auxp->quad.rowtab_rows = 0;
      bx0 = int(cx0); by0 = int(cy0); bx1 = int(cx0) + 1; by1 = int(cy0) + 1;      // (a placeholder box: replaced by the walk's)
    }
    else if (o.kind == WR_PK_MIX_BLEND && persp) {
      // ... under a projective transform the z / w slots are taken: the second varying, divided by w like the first, on edges of its own
      float qu2[4], qv2[4];
      for (int n = 0; n < 4; n++) { qu2[n] = o.u2[n] * pw3[n]; qv2[n] = o.v2[n] * pw3[n]; }
      if (!wr_quad_walk(sx, sy, qu, qv, cx0, cy0, cx1, cy1, aa, o.aa_edges, auxp->quad, bx0, by0, bx1, by1, pz3, pw3, true, qu2, qv2)) return;
    }
    else if (o.kind == WR_PK_MIX_BLEND) {
      // brush_mix_blend's second varying (v_src_uv) rides in the walk's z / w slots
      if (!wr_quad_walk(sx, sy, qu, qv, cx0, cy0, cx1, cy1, aa, o.aa_edges, auxp->quad, bx0, by0, bx1, by1, o.u2, o.v2, true)) return;
    }
    else if (!wr_quad_walk(sx, sy, qu, qv, cx0, cy0, cx1, cy1, aa, o.aa_edges, auxp->quad, bx0, by0, bx1, by1, pz3, pw3, persp)) return;
    // 1: a program without varyings (brush_solid): glsl-to-cxx wires its perspective entry points to the plain ones, which never
    // step gl_FragCoord.z -- every chunk of a span is depth-tested with the z of the span's first four pixels; 2: a program with
    // varyings (ps_quad_textured): run_perspective / skip_perspective advance z and w chunk by chunk (lib.rs:656-659, 3627-3636)
    auxp->quad.pad = persp ? ((o.kind == WR_PK_SOLID && d.shader != WR_SH_PS_QUAD_TEXTURED) ? 1 : 2) : 0;
    if (persp) auxp->quad.persp.div = o.persp_div;
    P.x0 = wr_imax(bx0, int(cx0)); P.x1 = wr_imin(bx1, int(cx1)); P.y0 = wr_imax(by0, int(cy0)); P.y1 = wr_imin(by1, int(ceilf(cy1)));
    // gl_ClipDistance (ps_text_run GLYPH_TRANSFORM: WrVsOut::cd): the rows' spans are cut to the box after they are rounded
    // (rasterize.h:953-955) -- the raster stage intersects every span with the prim's box and starts the span there
    if (o.cd[2] >= o.cd[0]) { P.x0 = wr_imax(P.x0, o.cd[0]); P.y0 = wr_imax(P.y0, o.cd[1]); P.x1 = wr_imin(P.x1, o.cd[2]); P.y1 = wr_imin(P.y1, o.cd[3]); }
    if (P.x1 <= P.x0 || P.y1 <= P.y0) return;
    P.rows_linear = 0;
    if (solidq) {
      P.kind = WR_PK_SOLID_QUAD;
      wr_pack_color(o.color, P.color);
      return;
    }
    P.kind = WR_PK_TEX_QUAD;
    auxp->quad.base_kind = o.kind == WR_PK_SOLID ? (int)WR_PK_SOLID_MASKED : o.kind;
    if (o.kind == WR_PK_SOLID) wr_pack_color(o.color, P.color);
    if (o.kind == WR_PK_TEX_REPEAT) auxp->quad.rep = base.rep;
    else if (o.kind == WR_PK_GRADIENT) auxp->quad.grad = base.grad;
    else if (o.kind == WR_PK_FILTER) auxp->quad.filt = base.filt;
    else if (o.kind == WR_PK_QUAD_MASK) auxp->quad.clip = base.clip;
    else if (o.kind == WR_PK_MIX_BLEND) auxp->quad.mix = base.mix;
    if (masked) P.flags |= WR_PF_MASKED;
    if (o.has_color) { P.flags |= WR_PF_HAS_COLOR; wr_pack_color(o.color, P.color); }
    if (o.tail_clamp) P.flags |= WR_PF_TAIL_CLAMP;
    if (o.tail_modulate) P.flags |= WR_PF_TAIL_MODULATE;
    P.tex_slot = o.tex_slot;
    P.uv_bounds[0] = o.uv_bounds.x; P.uv_bounds[1] = o.uv_bounds.y; P.uv_bounds[2] = o.uv_bounds.z; P.uv_bounds[3] = o.uv_bounds.w;
    P.fcolor[0] = o.color.x; P.fcolor[1] = o.color.y; P.fcolor[2] = o.color.z; P.fcolor[3] = o.color.w;
    if (o.blend_override != 0 && d.blend != WR_BLEND_NONE) { P.blend = (int16_t)o.blend_override; wr_pack_color(o.blend_color, P.color); }
    return;
  }
  float xa = sx[0], xb = sx[2], ya = sy[0], yb = sy[2];
  float xmin = wr_min(xa, xb), xmax = wr_max(xa, xb), ymin = wr_min(ya, yb), ymax = wr_max(ya, yb);
  // clipSpan = clipRect.x_range().clip(edge x range); span = clipSpan.clip({left.x,right.x}).round()
  float sx0 = wr_clamp(xmin, cx0, cx1), sx1 = wr_clamp(xmax, cx0, cx1);
  int ix0 = int(floorf(sx0 + 0.5f)), ix1 = int(floorf(sx1 + 0.5f));
  float aaRound = 0.5f;
  if (aa) {
    // aa_span (rasterize.h:520-561) for vertical edges (x slope 0): an edge with its mask bit set is
    // rounded out and gets a coverage ramp, the others round to nearest.  The mask bit of an edge is
    // indexed by the vertex it ends at (Edge ctor: left edges pass l1i, right edges r0i).
    int li, ri;                       // vertex index whose bit masks the screen-left / screen-right edge
    if (typeA) { const bool flipx = sx[0] > sx[1]; li = flipx ? 2 : 0; ri = flipx ? 0 : 2; }
    else { const bool flipx = sx[0] > sx[3]; li = flipx ? 3 : 1; ri = flipx ? 1 : 3; }
    const bool ml = (o.aa_edges >> li) & 1, mr = (o.aa_edges >> ri) & 1;
    const int l_start = ml ? int(floorf(sx0)) : int(floorf(sx0 + 0.5f));
    const int l_end = ml ? int(ceilf(sx0)) : int(floorf(sx0 + 0.5f));
    const int r_end = mr ? int(ceilf(sx1)) : int(floorf(sx1 + 0.5f));
    ix0 = l_start; ix1 = r_end;
    WrAARec& A = auxp->aa;
    // aa_dist (rasterize.h:508-517): dx = dir * 256 * inversesqrt(1 + slope^2), slope == 0
    const float dxl = (-1.0f * 256.0f) * (1.0f / sqrtf(1.0f + 0.0f * 0.0f)), dxr = (1.0f * 256.0f) * (1.0f / sqrtf(1.0f + 0.0f * 0.0f));
    A.lstart = ml ? 128.0f + dxl * (xmin - 0.5f) : 256.0f; A.lend = ml ? -dxl : 0.0f;
    A.rstart = mr ? 128.0f + dxr * (xmax - 0.5f) : 256.0f; A.rend = mr ? -dxr : 0.0f;
    A.laa_end = l_end;
    aaRound = 0.0f;                   // conservative vertical round-out (rasterize.h:889-890)
  }
  // first row centre: floor(max(min(l0.y, clip.y1), clip.y0) + aaRound) + 0.5 ; rows while y <= min(bottom, clip.y1)
  float ystart = floorf(wr_max(wr_min(ymin, cy1), cy0) + aaRound) + 0.5f;
  float ylimit = wr_min(ymax, cy1);
  int iy0 = int(ystart);
  int iy1 = int(floorf(ylimit - 0.5f)) + 1;
  while ((float(iy1) + 0.5f) <= ylimit) iy1++;
  while (iy1 > iy0 && (float(iy1 - 1) + 0.5f) > ylimit) iy1--;
  // (gl_ClipDistance as a box, see above.  An axis-aligned glyph quad is the glyph's raster rect, or what the local clip rect
  // leaves of it: it lies inside the box, the cut is a no-op; x and the last row are cut all the same, the first row is not --
  // the edge interpolants start from it)
  if (o.cd[2] >= o.cd[0]) { ix0 = wr_imax(ix0, o.cd[0]); ix1 = wr_imin(ix1, o.cd[2]); iy1 = wr_imin(iy1, o.cd[3]); }
  if (ix1 <= ix0 || iy1 <= iy0) return;
  P.x0 = ix0; P.x1 = ix1; P.y0 = iy0; P.y1 = iy1;
  P.kind = (masked && o.kind == WR_PK_SOLID) ? (int16_t)WR_PK_SOLID_MASKED : (aa ? (int16_t)WR_PK_SOLID_AA : (int16_t)o.kind);
  if (masked && o.kind != WR_PK_SOLID) P.flags |= WR_PF_MASKED;
  P.rows_linear = 0;
  if ((d.flags & WR_DF_SIMPLE) && P.kind != WR_PK_SOLID) {   // the launch's kernel has no path for it: say so
    P.kind = WR_PK_UNSUPPORTED; atomicAdd(&cnt->unsupported_prims, 1u); return;
  }
  if (o.kind == WR_PK_UNSUPPORTED) { atomicAdd(&cnt->unsupported_prims, 1u); return; }
  const bool blend_override = o.blend_override != 0 && d.blend != WR_BLEND_NONE;   // only while blending is on (ClipRect ctor, rasterize.h:412-417)
  if (o.kind == WR_PK_SOLID) {
    wr_pack_color(o.color, P.color);
    if (masked) P.tex_slot = WR_S_CLIP_MASK;
  } else if (o.kind == WR_PK_TEX_RGBA8 || o.kind == WR_PK_TEX_R8 || o.kind == WR_PK_BLUR || o.kind == WR_PK_TEX_FS || o.kind == WR_PK_CLIP_RECT || o.kind == WR_PK_BOX_SHADOW || o.kind == WR_PK_GRADIENT || o.kind == WR_PK_FILTER || o.kind == WR_PK_MIX_BLEND || o.kind == WR_PK_SVG_FILTER || o.kind == WR_PK_YUV || o.kind == WR_PK_QUAD_MASK || o.kind == WR_PK_TEX_REPEAT || o.kind == WR_PK_BORDER_SOLID || o.kind == WR_PK_BORDER_SEGMENT || o.kind == WR_PK_FAST_GRADIENT || o.kind == WR_PK_LINE_DECORATION) {
    if (o.has_color) { P.flags |= WR_PF_HAS_COLOR; wr_pack_color(o.color, P.color); }
    if (o.tail_clamp) P.flags |= WR_PF_TAIL_CLAMP;
    if (o.tail_modulate) P.flags |= WR_PF_TAIL_MODULATE;
    P.tex_slot = o.tex_slot;
    P.uv_bounds[0] = o.uv_bounds.x; P.uv_bounds[1] = o.uv_bounds.y;
    P.uv_bounds[2] = o.uv_bounds.z; P.uv_bounds[3] = o.uv_bounds.w;
    P.fcolor[0] = o.color.x; P.fcolor[1] = o.color.y; P.fcolor[2] = o.color.z; P.fcolor[3] = o.color.w;
    // Edge interpolants (Edge ctor, rasterize.h:858-876). Find which lanes are
    // top-left/top-right/bottom-left/bottom-right on screen.
    int tl = 0, tr = 0, bl = 0, br = 0;
    for (int n = 0; n < 4; n++) {
      bool left = sx[n] == xmin, top = sy[n] == ymin;
      if (left && top) tl = n; else if (!left && top) tr = n; else if (left) bl = n; else br = n;
    }
    float yScale = 1.0f / wr_max(ymax - ymin, 1.0f / 256.0f);
    float dy0 = ystart - ymin;
    // left edge
    float lsu = (wr_pick4(o.u, bl) - wr_pick4(o.u, tl)) * yScale, lsv = (wr_pick4(o.v, bl) - wr_pick4(o.v, tl)) * yScale;
    float rsu = (wr_pick4(o.u, br) - wr_pick4(o.u, tr)) * yScale, rsv = (wr_pick4(o.v, br) - wr_pick4(o.v, tr)) * yScale;
    P.uvL0[0] = wr_pick4(o.u, tl) + dy0 * lsu; P.uvL0[1] = wr_pick4(o.v, tl) + dy0 * lsv;
    P.uvR0[0] = wr_pick4(o.u, tr) + dy0 * rsu; P.uvR0[1] = wr_pick4(o.v, tr) + dy0 * rsv;
    P.uvLs[0] = lsu; P.uvLs[1] = lsv; P.uvRs[0] = rsu; P.uvRs[1] = rsv;
    P.xl = xmin; P.xr = xmax;
    {
      const int rows = iy1 - iy0 - 1;
      P.rows_linear = (wr_accum_is_linear(P.uvL0[0], lsu, rows) && wr_accum_is_linear(P.uvL0[1], lsv, rows) &&
                       wr_accum_is_linear(P.uvR0[0], rsu, rows) && wr_accum_is_linear(P.uvR0[1], rsv, rows)) ? 1 : 0;
    }
    if (o.kind == WR_PK_MIX_BLEND) {
      WrMixRec& M = auxp->mix;
      const float l2u = (wr_pick4(o.u2, bl) - wr_pick4(o.u2, tl)) * yScale, l2v = (wr_pick4(o.v2, bl) - wr_pick4(o.v2, tl)) * yScale;
      const float r2u = (wr_pick4(o.u2, br) - wr_pick4(o.u2, tr)) * yScale, r2v = (wr_pick4(o.v2, br) - wr_pick4(o.v2, tr)) * yScale;
      M.sL0[0] = wr_pick4(o.u2, tl) + dy0 * l2u; M.sL0[1] = wr_pick4(o.v2, tl) + dy0 * l2v;
      M.sR0[0] = wr_pick4(o.u2, tr) + dy0 * r2u; M.sR0[1] = wr_pick4(o.v2, tr) + dy0 * r2v;
      M.sLs[0] = l2u; M.sLs[1] = l2v; M.sRs[0] = r2u; M.sRs[1] = r2v;
    }
    if (o.kind == WR_PK_SVG_FILTER) {
      WrSvgRec& M = auxp->svg;
      const float l2u = (wr_pick4(o.u2, bl) - wr_pick4(o.u2, tl)) * yScale, l2v = (wr_pick4(o.v2, bl) - wr_pick4(o.v2, tl)) * yScale;
      const float r2u = (wr_pick4(o.u2, br) - wr_pick4(o.u2, tr)) * yScale, r2v = (wr_pick4(o.v2, br) - wr_pick4(o.v2, tr)) * yScale;
      M.sL0[0] = wr_pick4(o.u2, tl) + dy0 * l2u; M.sL0[1] = wr_pick4(o.v2, tl) + dy0 * l2v;
      M.sR0[0] = wr_pick4(o.u2, tr) + dy0 * r2u; M.sR0[1] = wr_pick4(o.v2, tr) + dy0 * r2v;
      M.sLs[0] = l2u; M.sLs[1] = l2v; M.sRs[0] = r2u; M.sRs[1] = r2v;
    }
    if (o.kind == WR_PK_YUV) {
      WrYuvRec& Y = auxp->yuv;
      for (int pl = 1; pl < 3; pl++) {
        const float* pu = pl == 1 ? o.u2 : o.u3; const float* pv = pl == 1 ? o.v2 : o.v3;
        const float u4[4] = {pu[0], pu[1], pu[2], pu[3]}, v4[4] = {pv[0], pv[1], pv[2], pv[3]};
        const float l2u = (wr_pick4(u4, bl) - wr_pick4(u4, tl)) * yScale, l2v = (wr_pick4(v4, bl) - wr_pick4(v4, tl)) * yScale;
        const float r2u = (wr_pick4(u4, br) - wr_pick4(u4, tr)) * yScale, r2v = (wr_pick4(v4, br) - wr_pick4(v4, tr)) * yScale;
        float* L0 = pl == 1 ? Y.uL0 : Y.vL0; float* Ls = pl == 1 ? Y.uLs : Y.vLs; float* R0 = pl == 1 ? Y.uR0 : Y.vR0; float* Rs = pl == 1 ? Y.uRs : Y.vRs;
        L0[0] = wr_pick4(u4, tl) + dy0 * l2u; L0[1] = wr_pick4(v4, tl) + dy0 * l2v;
        R0[0] = wr_pick4(u4, tr) + dy0 * r2u; R0[1] = wr_pick4(v4, tr) + dy0 * r2v;
        Ls[0] = l2u; Ls[1] = l2v; Rs[0] = r2u; Rs[1] = r2v;
      }
    }
    if (o.kind == WR_PK_BOX_SHADOW) {
      WrBoxRec& B = auxp->box;
      const float l2u = (wr_pick4(o.u2, bl) - wr_pick4(o.u2, tl)) * yScale, l2v = (wr_pick4(o.v2, bl) - wr_pick4(o.v2, tl)) * yScale;
      const float r2u = (wr_pick4(o.u2, br) - wr_pick4(o.u2, tr)) * yScale, r2v = (wr_pick4(o.v2, br) - wr_pick4(o.v2, tr)) * yScale;
      B.lpL0[0] = wr_pick4(o.u2, tl) + dy0 * l2u; B.lpL0[1] = wr_pick4(o.v2, tl) + dy0 * l2v;
      B.lpR0[0] = wr_pick4(o.u2, tr) + dy0 * r2u; B.lpR0[1] = wr_pick4(o.v2, tr) + dy0 * r2v;
      B.lpLs[0] = l2u; B.lpLs[1] = l2v; B.lpRs[0] = r2u; B.lpRs[1] = r2v;
    }
  }
  if (blend_override) {     // the prim's colour slot carries swgl_BlendColorRGBA8 (its own colour is 1: no modulation)
    P.blend = (int16_t)o.blend_override;
    wr_pack_color(o.blend_color, P.color);
  }
}

// ---------------------------------------------------------------------------
// Blend stage (swgl/src/blend.h).  Pixels are BGRA8 in a u32; arithmetic is
// done on 4 x u16 lanes exactly as WideRGBA8 (wrapping 16-bit), then `pack`ed
// with the portable saturating pack (texture.h:14-21).

struct WrWide { uint32_t bg, ra; };   // (b | g<<16), (r | a<<16): 4 x u16

WR_DEVICE WrWide wr_unpack(uint32_t p) {
  WrWide w;
  w.bg = (p & 0xFF) | ((p & 0xFF00) << 8);
  w.ra = ((p >> 16) & 0xFF) | ((p >> 24) << 16);
  return w;
}
// genericPackWide: p = (p | (p > 255 ? 0xFFFF : 0)) + (p >> 15); low byte
WR_DEVICE uint32_t wr_pack1(uint32_t v) {  // v: one u16 lane
  uint32_t m = v > 255u ? 0xFFFFu : 0u;
  return (((v | m) + (v >> 15))) & 0xFFu;
}
WR_DEVICE uint32_t wr_pack(WrWide w) {
  return wr_pack1(w.bg & 0xFFFF) | (wr_pack1(w.bg >> 16) << 8) |
         (wr_pack1(w.ra & 0xFFFF) << 16) | (wr_pack1(w.ra >> 16) << 24);
}
// lane-wise u16 ops on a pair packed in u32 (wrapping mod 2^16 per lane)
WR_DEVICE uint32_t wr_add2(uint32_t a, uint32_t b) { return ((a & 0xFFFF) + (b & 0xFFFF) & 0xFFFF) | (((a >> 16) + (b >> 16)) << 16); }
WR_DEVICE uint32_t wr_sub2(uint32_t a, uint32_t b) { return ((a & 0xFFFF) - (b & 0xFFFF) & 0xFFFF) | (((a >> 16) - (b >> 16)) << 16); }
// muldiv255: (x*y + x) >> 8 in u16 lanes (blend.h:126-128)
WR_DEVICE uint32_t wr_muldiv255_2(uint32_t x, uint32_t y) {
  uint32_t lo = (((x & 0xFFFF) * (y & 0xFFFF) + (x & 0xFFFF)) & 0xFFFF) >> 8;
  uint32_t hi = ((((x >> 16) * (y >> 16) + (x >> 16)) & 0xFFFF) >> 8);
  return lo | (hi << 16);
}
WR_DEVICE uint32_t wr_splat_a(uint32_t ra) { uint32_t a = ra >> 16; return a | (a << 16); }

// blend_pixels for RGBA8 (blend.h:416-701): the keys WebRender's in-scope
// batches use.  `src` already has clip-mask/AA weights applied.
WR_DEVICE WrWide wr_apply_color(WrWide src, const uint32_t color[2]);

// ---- the rest of swgl's blend-key table (blend.h:490-677): constant colour, MIN / MAX, and the
// KHR_blend_equation_advanced equations on premultiplied values.  Out of line: none of it is hot.
// Lane helpers on two u16 lanes in a u32 (WideRGBA8 arithmetic wraps mod 2^16 per lane).
WR_DEVICE uint32_t wr_min2(uint32_t a, uint32_t b) {      // portable min(HalfRGBA8): unsigned lanes
  const uint32_t lo = (a & 0xFFFF) < (b & 0xFFFF) ? (a & 0xFFFF) : (b & 0xFFFF), hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
  return lo | (hi << 16);
}
WR_DEVICE uint32_t wr_max2(uint32_t a, uint32_t b) {
  const uint32_t lo = (a & 0xFFFF) > (b & 0xFFFF) ? (a & 0xFFFF) : (b & 0xFFFF), hi = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
  return lo | (hi << 16);
}
WR_DEVICE uint32_t wr_neg2(uint32_t a) { return wr_sub2(0u, a); }
// if_then_else(x * 2 <= y, t, e) per lane
WR_DEVICE uint32_t wr_sel_le2(uint32_t x, uint32_t y, uint32_t t, uint32_t e) {
  const bool lo = (((x & 0xFFFF) * 2u) & 0xFFFF) <= (y & 0xFFFF), hi = (((x >> 16) * 2u) & 0xFFFF) <= (y >> 16);
  return ((lo ? t : e) & 0xFFFF) | ((hi ? t : e) & 0xFFFF0000u);
}
WR_DEVICE uint32_t wr_addlow2(uint32_t a, uint32_t b) {   // addlow: byte-wise add (blend.h:201-205)
  return (((a & 0x00FF00FFu) + (b & 0x00FF00FFu)) & 0x00FF00FFu) | (((((a >> 8) & 0x00FF00FFu) + ((b >> 8) & 0x00FF00FFu)) & 0x00FF00FFu) << 8);
}
WR_DEVICE float wr_recip_or(float v, float f) { return v != 0.0f ? 1.0f / v : f; }
WR_DEVICE uint32_t wr_round_scaled(float v, float scale) { return uint32_t(int(v * scale + 0.5f)) & 0xFFFF; }   // round_pixel(v, scale) -> u16 lane
// clip_color / set_lum / set_lum_sat (blend.h:296-327) on one pixel
WR_DEVICE void wr_clip_color(float (&v)[3], float lum, float alpha) {
  const float mincol = wr_max(-wr_min(wr_min(v[0], v[1]), v[2]), lum);
  const float maxcol = wr_max(wr_max(wr_max(v[0], v[1]), v[2]), alpha - lum);
  const float k = lum * (alpha - lum) * wr_recip_or(mincol * maxcol, 0.0f);
  for (int i = 0; i < 3; i++) v[i] = lum + v[i] * k;
}
WR_DEVICE float wr_lumv3(const float (&v)[3]) { return v[0] * 0.30f + v[1] * 0.59f + v[2] * 0.11f; }
WR_DEVICE void wr_set_lum(float (&base)[3], const float (&ref)[3], float alpha) {
  const float lb = wr_lumv3(base);
  for (int i = 0; i < 3; i++) base[i] = base[i] - lb;
  wr_clip_color(base, wr_lumv3(ref), alpha);
}
WR_DEVICE void wr_set_lum_sat(float (&base)[3], const float (&sref)[3], const float (&lref)[3], float alpha) {
  const float mn = wr_min(wr_min(base[0], base[1]), base[2]);
  float diff[3] = {base[0] - mn, base[1] - mn, base[2] - mn};
  const float sbase = wr_max(wr_max(diff[0], diff[1]), diff[2]);
  const float ssat = wr_max(wr_max(sref[0], sref[1]), sref[2]) - wr_min(wr_min(sref[0], sref[1]), sref[2]);
  const float rs = wr_recip_or(sbase, 0.0f);
  for (int i = 0; i < 3; i++) base[i] = diff[i] * ssat * rs;
  wr_set_lum(base, lref, alpha);
}
__device__ __noinline__ uint32_t wr_blend_advanced(int key, uint32_t dstp, uint32_t sbg, uint32_t sra, uint32_t bc0, uint32_t bc1) {
  WrWide src; src.bg = sbg; src.ra = sra;
  WrWide dst = wr_unpack(dstp), r;
  const uint32_t sa = wr_splat_a(src.ra), da = wr_splat_a(dst.ra);
  const uint32_t RGB_RA = 0x0000FFFFu, A_RA = 0xFFFF0000u;      // RGB_MASK / ALPHA_MASK on the (r, a) pair; the (b, g) pair is all colour
  switch (key) {
    case WR_BLEND_CONST_COLOR:      // addlow(dst, muldiv255(src, repeat2(ctx->blendcolor) - dst))
      r.bg = wr_addlow2(dst.bg, wr_muldiv255_2(src.bg, wr_sub2(bc0, dst.bg)));
      r.ra = wr_addlow2(dst.ra, wr_muldiv255_2(src.ra, wr_sub2(bc1, dst.ra)));
      break;
    case WR_BLEND_MIN: r.bg = wr_min2(src.bg, dst.bg); r.ra = wr_min2(src.ra, dst.ra); break;
    case WR_BLEND_MAX: r.bg = wr_max2(src.bg, dst.bg); r.ra = wr_max2(src.ra, dst.ra); break;
    case WR_BLEND_MULTIPLY_KHR: {   // diff = muldiv255(alphas(src) - (src & RGB), alphas(dst) - (dst & RGB)); src + dst + (diff & RGB) - alphas(diff)
      const uint32_t dbg = wr_muldiv255_2(wr_sub2(sa, src.bg), wr_sub2(da, dst.bg));
      const uint32_t dra = wr_muldiv255_2(wr_sub2(sa, src.ra & RGB_RA), wr_sub2(da, dst.ra & RGB_RA));
      const uint32_t dal = wr_splat_a(dra);
      r.bg = wr_sub2(wr_add2(wr_add2(src.bg, dst.bg), dbg), dal);
      r.ra = wr_sub2(wr_add2(wr_add2(src.ra, dst.ra), dra & RGB_RA), dal);
    } break;
    case WR_BLEND_SCREEN_KHR:
      r.bg = wr_sub2(wr_add2(src.bg, dst.bg), wr_muldiv255_2(src.bg, dst.bg));
      r.ra = wr_sub2(wr_add2(src.ra, dst.ra), wr_muldiv255_2(src.ra, dst.ra));
      break;
    case WR_BLEND_OVERLAY_KHR:
    case WR_BLEND_HARDLIGHT_KHR: {  // diff = muldiv255(src, dst) + muldiv255(srcA - src, dstA - dst); src + dst + (c * 2 <= cA ? (diff & RGB) - alphas(diff) : -diff)
      const uint32_t dbg = wr_add2(wr_muldiv255_2(src.bg, dst.bg), wr_muldiv255_2(wr_sub2(sa, src.bg), wr_sub2(da, dst.bg)));
      const uint32_t dra = wr_add2(wr_muldiv255_2(src.ra, dst.ra), wr_muldiv255_2(wr_sub2(sa, src.ra), wr_sub2(da, dst.ra)));
      const uint32_t dal = wr_splat_a(dra);
      const bool ov = key == WR_BLEND_OVERLAY_KHR;
      const uint32_t tbg = wr_sel_le2(ov ? dst.bg : src.bg, ov ? da : sa, wr_sub2(dbg, dal), wr_neg2(dbg));
      const uint32_t tra = wr_sel_le2(ov ? dst.ra : src.ra, ov ? da : sa, wr_sub2(dra & RGB_RA, dal), wr_neg2(dra));
      r.bg = wr_add2(wr_add2(src.bg, dst.bg), tbg);
      r.ra = wr_add2(wr_add2(src.ra, dst.ra), tra);
    } break;
    case WR_BLEND_DARKEN_KHR:       // src + dst - max(muldiv255(src, alphas(dst)), muldiv255(dst, alphas(src)))
      r.bg = wr_sub2(wr_add2(src.bg, dst.bg), wr_max2(wr_muldiv255_2(src.bg, da), wr_muldiv255_2(dst.bg, sa)));
      r.ra = wr_sub2(wr_add2(src.ra, dst.ra), wr_max2(wr_muldiv255_2(src.ra, da), wr_muldiv255_2(dst.ra, sa)));
      break;
    case WR_BLEND_LIGHTEN_KHR:
      r.bg = wr_sub2(wr_add2(src.bg, dst.bg), wr_min2(wr_muldiv255_2(src.bg, da), wr_muldiv255_2(dst.bg, sa)));
      r.ra = wr_sub2(wr_add2(src.ra, dst.ra), wr_min2(wr_muldiv255_2(src.ra, da), wr_muldiv255_2(dst.ra, sa)));
      break;
    case WR_BLEND_DIFFERENCE_KHR: { // diff = min(muldiv255(dst, alphas(src)), muldiv255(src, alphas(dst))); src + dst - diff - (diff & RGB)
      const uint32_t dbg = wr_min2(wr_muldiv255_2(dst.bg, sa), wr_muldiv255_2(src.bg, da));
      const uint32_t dra = wr_min2(wr_muldiv255_2(dst.ra, sa), wr_muldiv255_2(src.ra, da));
      r.bg = wr_sub2(wr_sub2(wr_add2(src.bg, dst.bg), dbg), dbg);
      r.ra = wr_sub2(wr_sub2(wr_add2(src.ra, dst.ra), dra), dra & RGB_RA);
    } break;
    case WR_BLEND_EXCLUSION_KHR: {
      const uint32_t dbg = wr_muldiv255_2(src.bg, dst.bg), dra = wr_muldiv255_2(src.ra, dst.ra);
      r.bg = wr_sub2(wr_sub2(wr_add2(src.bg, dst.bg), dbg), dbg);
      r.ra = wr_sub2(wr_sub2(wr_add2(src.ra, dst.ra), dra), dra & RGB_RA);
    } break;
    default: {
      // the float equations: lanes b, g, r, a as floats 0..255 (CONVERT(src, WideRGBA32F))
      const float sf[4] = {float(src.bg & 0xFFFF), float(src.bg >> 16), float(src.ra & 0xFFFF), float(src.ra >> 16)};
      const float df[4] = {float(dst.bg & 0xFFFF), float(dst.bg >> 16), float(dst.ra & 0xFFFF), float(dst.ra >> 16)};
      const float sA = sf[3], dA = df[3];
      float o[4];
      const float k = 1.0f / 255.0f;
      if (key == WR_BLEND_COLORDODGE_KHR || key == WR_BLEND_COLORBURN_KHR) {
        for (int i = 0; i < 4; i++) {
          float t;     // set_alphas(<colour term>, dstF)
          if (i == 3) t = df[3];
          else if (key == WR_BLEND_COLORDODGE_KHR) t = wr_min(dA, df[i] * sA * wr_recip_or(sA - sf[i], 255.0f));
          else t = dA - wr_min(dA, (dA - df[i]) * sA * wr_recip_or(sf[i], 255.0f));
          o[i] = sA * t + sf[i] * (255.0f - dA) + df[i] * (255.0f - sA);
        }
        r.bg = wr_round_scaled(o[0], k) | (wr_round_scaled(o[1], k) << 16);
        r.ra = wr_round_scaled(o[2], k) | (wr_round_scaled(o[3], k) << 16);
      } else if (key == WR_BLEND_SOFTLIGHT_KHR) {
        const float ra_ = wr_recip_or(dA, 0.0f);
        for (int i = 0; i < 4; i++) {
          const float dstU = df[i] * ra_;
          const float scale = sf[i] + sf[i] - sA;
          float t = 0.0f;   // set_alphas(..., 0)
          if (i < 3) t = scale * (scale < 0.0f ? 1.0f - dstU : wr_min((16.0f * dstU - 12.0f) * dstU + 3.0f, (1.0f / sqrtf(dstU)) - 1.0f));
          o[i] = df[i] * (255.0f + t) + sf[i] * (255.0f - dA);
        }
        r.bg = wr_round_scaled(o[0], k) | (wr_round_scaled(o[1], k) << 16);
        r.ra = wr_round_scaled(o[2], k) | (wr_round_scaled(o[3], k) << 16);
      } else {
        // DO_HSL (blend.h:651-677): vec4 in r, g, b, a order
        const float sv[3] = {sf[2], sf[1], sf[0]}, dv[3] = {df[2], df[1], df[0]};
        const float srcA = sA * k, dstA = dA * k, srcDstA = sA * dstA;
        float sc[3], dc[3];
        for (int i = 0; i < 3; i++) { sc[i] = sv[i] * dstA; dc[i] = dv[i] * srcA; }
        float rgb[3];
        if (key == WR_BLEND_HSL_HUE_KHR) { for (int i = 0; i < 3; i++) rgb[i] = sc[i]; wr_set_lum_sat(rgb, dc, dc, srcDstA); }
        else if (key == WR_BLEND_HSL_SATURATION_KHR) { for (int i = 0; i < 3; i++) rgb[i] = dc[i]; wr_set_lum_sat(rgb, sc, dc, srcDstA); }
        else if (key == WR_BLEND_HSL_COLOR_KHR) { for (int i = 0; i < 3; i++) rgb[i] = sc[i]; wr_set_lum(rgb, dc, srcDstA); }
        else { for (int i = 0; i < 3; i++) rgb[i] = dc[i]; wr_set_lum(rgb, sc, srcDstA); }
        float res[4];
        for (int i = 0; i < 3; i++) res[i] = (((rgb[i] + sv[i]) - sc[i]) + dv[i]) - dc[i];
        res[3] = (sA + dA) - srcDstA;
        // pack_pixels_RGBA8(vec4, 1.0f): b, g, r, a lanes
        r.bg = wr_round_scaled(res[2], 1.0f) | (wr_round_scaled(res[1], 1.0f) << 16);
        r.ra = wr_round_scaled(res[0], 1.0f) | (wr_round_scaled(res[3], 1.0f) << 16);
      }
    } break;
  }
  (void)A_RA;
  return wr_pack(r);
}

WR_DEVICE uint32_t wr_blend_rgba8(int key, uint32_t dstp, WrWide src, const WrDrawDesc* d, const uint32_t* bc = nullptr) {
  if (key == WR_BLEND_CONST_COLOR || key == WR_BLEND_MIN || key == WR_BLEND_MAX || key > WR_BLEND_UNSUPPORTED)
    return wr_blend_advanced(key, dstp, src.bg, src.ra, d ? d->blend_color[0] : 0u, d ? d->blend_color[1] : 0u);
  WrWide dst = wr_unpack(dstp), r;
  switch (key) {
    default:
    case WR_BLEND_NONE: r = src; break;
    case WR_BLEND_PREMULT: {  // src + dst - muldiv255(dst, alphas(src))
      uint32_t a = wr_splat_a(src.ra);
      r.bg = wr_sub2(wr_add2(src.bg, dst.bg), wr_muldiv255_2(dst.bg, a));
      r.ra = wr_sub2(wr_add2(src.ra, dst.ra), wr_muldiv255_2(dst.ra, a));
    } break;
    case WR_BLEND_ALPHA: {  // addlow(dst, muldiv255(alphas(src), (src | ALPHA_OPAQUE) - dst))
      uint32_t a = wr_splat_a(src.ra);
      uint32_t tbg = wr_muldiv255_2(a, wr_sub2(src.bg, dst.bg));
      uint32_t tra = wr_muldiv255_2(a, wr_sub2(src.ra | (255u << 16), dst.ra));
      // addlow: byte-wise add (blend.h:201-205)
      r.bg = ((dst.bg & 0x00FF00FF) + (tbg & 0x00FF00FF)) & 0x00FF00FF;
      r.bg |= (((dst.bg >> 8) & 0x00FF00FF) + ((tbg >> 8) & 0x00FF00FF) & 0x00FF00FF) << 8;
      r.ra = ((dst.ra & 0x00FF00FF) + (tra & 0x00FF00FF)) & 0x00FF00FF;
      r.ra |= (((dst.ra >> 8) & 0x00FF00FF) + ((tra >> 8) & 0x00FF00FF) & 0x00FF00FF) << 8;
    } break;
    case WR_BLEND_ZERO_INV_SRC_COLOR:  // dst - muldiv255(dst, src)
      r.bg = wr_sub2(dst.bg, wr_muldiv255_2(dst.bg, src.bg));
      r.ra = wr_sub2(dst.ra, wr_muldiv255_2(dst.ra, src.ra));
      break;
    case WR_BLEND_ZERO_INV_SRC_COLOR_A1:  // dst - (muldiv255(dst, src) & RGB_MASK)
      r.bg = wr_sub2(dst.bg, wr_muldiv255_2(dst.bg, src.bg));
      r.ra = wr_sub2(dst.ra, wr_muldiv255_2(dst.ra, src.ra) & 0x0000FFFF);
      break;
    case WR_BLEND_DEST_OUT: {  // dst - muldiv255(dst, alphas(src))
      uint32_t a = wr_splat_a(src.ra);
      r.bg = wr_sub2(dst.bg, wr_muldiv255_2(dst.bg, a));
      r.ra = wr_sub2(dst.ra, wr_muldiv255_2(dst.ra, a));
    } break;
    case WR_BLEND_MULTIPLY:  // muldiv255(src, dst)
      r.bg = wr_muldiv255_2(src.bg, dst.bg);
      r.ra = wr_muldiv255_2(src.ra, dst.ra);
      break;
    case WR_BLEND_ADD:
      r.bg = wr_add2(src.bg, dst.bg); r.ra = wr_add2(src.ra, dst.ra);
      break;
    case WR_BLEND_ADD_A_OVER:  // src + dst - (muldiv255(dst, src) & ALPHA_MASK)
      r.bg = wr_add2(src.bg, dst.bg);
      r.ra = wr_sub2(wr_add2(src.ra, dst.ra), wr_muldiv255_2(dst.ra, src.ra) & 0xFFFF0000);
      break;
    case WR_BLEND_INV_DST_A: {  // dst + ((src - muldiv255(src, alphas(dst))) & RGB_MASK)
      uint32_t a = wr_splat_a(dst.ra);
      r.bg = wr_add2(dst.bg, wr_sub2(src.bg, wr_muldiv255_2(src.bg, a)));
      r.ra = wr_add2(dst.ra, wr_sub2(src.ra, wr_muldiv255_2(src.ra, a)) & 0x0000FFFF);
    } break;
    case WR_BLEND_DROP_SHADOW: {   // blend.h:680-685; bc = swgl_BlendColorRGBA8
      if (!bc) { r = src; break; }
      const uint32_t sa = wr_splat_a(src.ra);
      WrWide al; al.bg = sa; al.ra = sa;
      const WrWide col = wr_apply_color(al, bc);
      const uint32_t ca = wr_splat_a(col.ra);
      r.bg = wr_sub2(wr_add2(col.bg, dst.bg), wr_muldiv255_2(dst.bg, ca));
      r.ra = wr_sub2(wr_add2(col.ra, dst.ra), wr_muldiv255_2(dst.ra, ca));
    } break;
    case WR_BLEND_SUBPIXEL_TEXT: { // blend.h:687-691; swgl_BlendAlphaRGBA8 = alphas(swgl_BlendColorRGBA8)
      if (!bc) { r = src; break; }
      const uint32_t ba[2] = {wr_splat_a(bc[1]), wr_splat_a(bc[1])};
      const WrWide c1 = wr_apply_color(src, bc), c2 = wr_apply_color(src, ba);
      r.bg = wr_sub2(wr_add2(c1.bg, dst.bg), wr_muldiv255_2(dst.bg, c2.bg));
      r.ra = wr_sub2(wr_add2(c1.ra, dst.ra), wr_muldiv255_2(dst.ra, c2.ra));
    } break;
    case WR_BLEND_SCREEN:  // ONE, ONE_MINUS_SRC_COLOR is not in swgl's key table: treated as unsupported upstream
      r = src; break;
  }
  (void)d;
  return wr_pack(r);
}

// R8 targets: blend_pixels(uint8_t*) (blend.h:703-735)
WR_DEVICE uint32_t wr_blend_r8(int key, uint32_t dst, uint32_t src) {
  uint32_t r;
  switch (key) {
    default:
    case WR_BLEND_NONE: r = src; break;
    case WR_BLEND_MULTIPLY: r = (((src * dst + src) & 0xFFFF) >> 8); break;
    case WR_BLEND_ADD: r = (src + dst) & 0xFFFF; break;
  }
  return wr_pack1(r);
}

// applyColor(src, color) = muldiv255(color, src)  (blend.h:156-159)
WR_DEVICE WrWide wr_apply_color(WrWide src, const uint32_t color[2]) {
  WrWide r;
  r.bg = wr_muldiv255_2(color[0], src.bg);
  r.ra = wr_muldiv255_2(color[1], src.ra);
  return r;
}

// ---------------------------------------------------------------------------
// Texture sampling helpers used by textured prims.

// textureLinearUnpackedRGBA8 for ONE pixel (texture.h:1028-1071): 7-bit
// fixed-point bilinear on quantised coords i = uv*size*128 + (0.5 - 64).
WR_DEVICE WrWide wr_sample_linear_rgba8(const WrTexDesc& t, int qx, int qy) {
  int fx = qx, fy = qy;
  int ix = qx >> 7, iy = qy >> 7;
  // computeRow(sampler, i) with margin 1
  int cx = wr_clamp_coord(ix, t.width - 1), cy = wr_clamp_coord(iy, t.height);
  const uint32_t* buf = (const uint32_t*)t.ptr;
  size_t row0 = (size_t)cx + (size_t)cy * t.stride;
  size_t row1 = row0 + ((iy >= 0 && iy < t.height - 1) ? t.stride : 0);
  // computeFracX: ((frac.x & (i.x >= 0)) | overread) & 0x7F) - overread, overread = i.x > width-2 as all-ones
  int over = ix > t.width - 2 ? -1 : 0;
  int fracx = ((((ix >= 0) ? fx : 0) | over) & 0x7F) - over;
  int fracy = fy & 0x7F;
  uint32_t a0 = buf[row0], a0n = buf[row0 + 1], a1 = buf[row1], a1n = buf[row1 + 1];
  WrWide out;
  int ch[4];
  for (int c = 0; c < 4; c++) {
    int p00 = (a0 >> (8 * c)) & 0xFF, p01 = (a0n >> (8 * c)) & 0xFF;
    int p10 = (a1 >> (8 * c)) & 0xFF, p11 = (a1n >> (8 * c)) & 0xFF;
    // int16 arithmetic: a0 += ((a1 - a0) * fracy) >> 7 ; then columns
    int l = (int16_t)(p00 + (int16_t)(((int16_t)((p10 - p00) * fracy)) >> 7));
    int r = (int16_t)(p01 + (int16_t)(((int16_t)((p11 - p01) * fracy)) >> 7));
    ch[c] = (int16_t)(l + (int16_t)(((int16_t)((r - l) * fracx)) >> 7));
  }
  out.bg = (uint32_t(ch[0]) & 0xFFFF) | ((uint32_t(ch[1]) & 0xFFFF) << 16);
  out.ra = (uint32_t(ch[2]) & 0xFFFF) | ((uint32_t(ch[3]) & 0xFFFF) << 16);
  return out;
}

// textureLinearUnpackedR8 for ONE pixel (texture.h:543-574): same 7-bit
// bilinear on a 1-byte texel; rows first, then columns, int16 wrap-around.
WR_DEVICE int wr_sample_linear_r8(const WrTexDesc& t, int qx, int qy) {
  int ix = qx >> 7, iy = qy >> 7;
  int cx = wr_clamp_coord(ix, t.width - 1), cy = wr_clamp_coord(iy, t.height);
  const uint8_t* buf = (const uint8_t*)t.ptr;
  size_t row0 = (size_t)cx + (size_t)cy * t.stride;
  size_t row1 = row0 + ((iy >= 0 && iy < t.height - 1) ? t.stride : 0);
  int over = ix > t.width - 2 ? -1 : 0;
  int fracx = ((((ix >= 0) ? qx : 0) | over) & 0x7F) - over;
  int fracy = qy & 0x7F;
  int p00 = buf[row0], p01 = buf[row0 + 1], p10 = buf[row1], p11 = buf[row1 + 1];
  int l = (int16_t)(p00 + (int16_t)(((int16_t)((p10 - p00) * fracy)) >> 7));
  int r = (int16_t)(p01 + (int16_t)(((int16_t)((p11 - p01) * fracy)) >> 7));
  return (int16_t)(l + (int16_t)(((int16_t)((r - l) * fracx)) >> 7));
}

// ---------------------------------------------------------------------------
// blendTextureLinear (swgl_ext.h:172-455): the exact per-pixel value of a span
// committed with linear filtering, for every filter variant swgl dispatches to.
//   q[k], qy[k]   quantised (1/128 texel) coordinates of the span's first four
//                 pixels (LINEAR_QUANTIZE_UV applied to the 4 SIMD lanes)
//   filter        needsTextureLinear's choice: 1 fallback, 2 upscale, 3 fast, 4 downscale
// NCH = 4 (RGBA8 texels) or 1 (R8); out[] gets the int16 lane values.
template <int NCH>
WR_DEVICE void wr_fetch_texel(const WrTexDesc& t, size_t idx, int (&c)[4]) {
  if (NCH == 4) {
    const uint32_t p = ((const uint32_t*)t.ptr)[idx];
    c[0] = p & 0xFF; c[1] = (p >> 8) & 0xFF; c[2] = (p >> 16) & 0xFF; c[3] = p >> 24;
  } else if (NCH == 2) {      // RG8 (the chroma plane of NV12): stride and index in 2-byte texels
    const uint32_t p = ((const uint16_t*)t.ptr)[idx];
    c[0] = p & 0xFF; c[1] = p >> 8; c[2] = c[3] = 0;
  } else {
    c[0] = ((const uint8_t*)t.ptr)[idx]; c[1] = c[2] = c[3] = 0;
  }
}
// src0 + (((src1 - src0) * fracy) >> 7) for the texel column `x` of a row pair
template <int NCH>
WR_DEVICE void wr_row_lerp(const WrTexDesc& t, size_t row0, size_t row1, ptrdiff_t x, int fracy, int (&v)[4]) {
  int a[4], b[4];
  wr_fetch_texel<NCH>(t, row0 + x, a);
  wr_fetch_texel<NCH>(t, row1 + x, b);
  for (int c = 0; c < NCH; c++) v[c] = (int16_t)(a[c] + (int16_t)(((int16_t)((b[c] - a[c]) * fracy)) >> 7));
}
WR_DEVICE int wr_frac_x(const WrTexDesc& t, int ix, int q) {   // computeFracX, texture.h:466-469
  const int over = ix > t.width - 2 ? -1 : 0;
  return ((((ix >= 0) ? q : 0) | over) & 0x7F) - over;
}
template <int NCH>
WR_DEVICE void wr_bilinear(const WrTexDesc& t, int qx, int qy, int (&v)[4]) {   // textureLinearUnpacked{RGBA8,R8}
  const int ix = qx >> 7, iy = qy >> 7;
  const size_t row0 = (size_t)wr_clamp_coord(ix, t.width - 1) + (size_t)wr_clamp_coord(iy, t.height) * t.stride;
  const size_t row1 = row0 + ((iy >= 0 && iy < t.height - 1) ? t.stride : 0);
  const int fracx = wr_frac_x(t, ix, qx), fracy = qy & 0x7F;
  int l[4], r[4];
  wr_row_lerp<NCH>(t, row0, row1, 0, fracy, l);
  wr_row_lerp<NCH>(t, row0, row1, 1, fracy, r);
  for (int c = 0; c < NCH; c++) v[c] = (int16_t)(l[c] + (int16_t)(((int16_t)((r[c] - l[c]) * fracx)) >> 7));
}

// textureLinearUnpackedR16 / RG16 (texture.h:654-822, the portable path): 15-bit samples, 7-bit fractions in the high byte,
// a 32-bit multiply-high and a doubling per blend; int16 lanes
template <int NCH>
WR_DEVICE void wr_bilinear16(const WrTexDesc& t, int qx, int qy, int (&v)[4]) {
  const int ix = qx >> 7, iy = qy >> 7;
  const size_t row0 = (size_t)wr_clamp_coord(ix, t.width - 1) + (size_t)wr_clamp_coord(iy, t.height) * t.stride;
  const size_t row1 = row0 + ((iy >= 0 && iy < t.height - 1) ? t.stride : 0);
  const int fracx = ((((ix >= 0) ? qx : 0) | (ix > t.width - 2 ? -1 : 0)) & 0x7F) << 8, fracy = (qy & 0x7F) << 8;
  auto texel = [&](size_t idx, int c) -> int {
    if (NCH == 1) return int(((const uint16_t*)t.ptr)[idx]) >> 1;
    const uint32_t p = ((const uint32_t*)t.ptr)[idx];
    return int(c == 0 ? (p & 0xFFFFu) : (p >> 16)) >> 1;
  };
  auto lerp = [](int a, int b, int f) -> int { return (int)(int16_t)(a + (int)(int16_t)((int)(int16_t)(((int)(int16_t)(b - a) * f) >> 16) * 2)); };      // (* 2, not << 1: the operand can be negative)
  for (int c = 0; c < NCH; c++) {
    const int l = lerp(texel(row0, c), texel(row1, c), fracy), r = lerp(texel(row0 + 1, c), texel(row1 + 1, c), fracy);
    v[c] = lerp(l, r, fracx);
  }
}

template <int NCH>
WR_DEVICE void wr_linear_span_pixel(const WrTexDesc& t, const float (&q)[4], const float (&qy)[4], float stepx, float stepy,
                                    float minx, float maxx, float miny, float maxy, int filter, int span, int n,
                                    int (&out)[4]) {
  const int k = n & 3;
  int before = 0, inside = 0;
  float U[4] = {q[0], q[1], q[2], q[3]};        // uv.x of the four lanes as the dispatcher advances it
  if (filter != 1) {
    // blendTextureLinearDispatch (swgl_ext.h:378-440)
    const float beforeDist = wr_max(0.0f, minx) - U[0];
    if (beforeDist > 0.0f) {
      before = wr_iclamp(int(ceilf(beforeDist / stepx)) * 4, 0, span);
      const float adv = float(before / 4) * stepx;
      for (int i = 0; i < 4; i++) U[i] += adv;
    }
    const float insideDist = wr_min(maxx, float((t.width - 4) * 128)) - U[0];
    if (stepx > 0.0f && insideDist >= stepx) {
      inside = span - before;
      if (filter == 4) inside = wr_imin(int(insideDist * (0.5f / 128.0f)) & ~3, inside);
      else if (filter == 2) inside = wr_imin(int(insideDist / stepx) * 4, inside);
      else inside = wr_imin(int(insideDist * (1.0f / 128.0f)) & ~3, inside);
      if (inside < 0) inside = 0;
    }
  }
  if (filter == 1 || n < before || n >= before + inside) {
    // blendTextureLinearFallback: uv += uv_step per chunk, clamp, full bilinear
    float ux, uy;
    int c;
    if (filter == 1 || n < before) {
      c = n >> 2; ux = q[k]; uy = qy[k];
    } else {
      const float adv = float(inside / 4) * stepx;
      c = (n - before - inside) >> 2; ux = U[k] + adv; uy = qy[k];
    }
    WR_DBG_PATH(4);
    ux = wr_accum(ux, stepx, c); uy = wr_accum(uy, stepy, c);
    wr_bilinear<NCH>(t, int(wr_clamp(ux, minx, maxx)), int(wr_clamp(uy, miny, maxy)), out);
    return;
  }
  const int j = n - before;             // pixel index inside the fast run
  const int iq0y = int(wr_clamp(qy[0], miny, maxy));
  const int iy = iq0y >> 7, fracy = iq0y & 0x7F;
  const size_t rowy = (size_t)wr_clamp_coord(iy, t.height) * t.stride;
  const size_t nexty = (iy >= 0 && iy < t.height - 1) ? t.stride : 0;
  if (filter == 3 || filter == 4) {
    // blendTextureLinearFast / Downscale: constant fractions from lane 0 of the run start
    const int iq0 = int(wr_clamp(U[0], minx, maxx));
    const int ix = iq0 >> 7;
    const int fracx = wr_frac_x(t, ix, iq0);
    const size_t row0 = rowy + wr_clamp_coord(ix, t.width - 1), row1 = row0 + nexty;
    WR_DBG_PATH(filter == 3 ? 6 : 7);
    const ptrdiff_t x = filter == 3 ? j : 2 * j;
    int a[4], b[4];
    wr_row_lerp<NCH>(t, row0, row1, x, fracy, a);
    wr_row_lerp<NCH>(t, row0, row1, x + 1, fracy, b);
    for (int c = 0; c < NCH; c++) out[c] = (int16_t)(a[c] + (int16_t)(((int16_t)((b[c] - a[c]) * fracx)) >> 7));
    return;
  }
  // blendTextureLinearUpscale (swgl_ext.h:203-284): chunk m of the run
  WR_DBG_PATH(5);
  const int m = j >> 2;
  int ich[4], fch[4], ixn0;
  for (int i = 0; i < 4; i++) {
    if (m == 0) {
      const int iq = int(wr_clamp(U[i], minx, maxx));
      ich[i] = iq >> 7; fch[i] = wr_frac_x(t, ich[i], iq);
    } else {
      const int iq = int(wr_accum(U[i], stepx, m));
      ich[i] = iq >> 7; fch[i] = iq & 0x7F;
    }
  }
  ixn0 = int(wr_accum(U[0], stepx, m + 1)) >> 7;
  int S[4] = {0, 1, 2, 3}, N[4] = {1, 2, 3, 4};      // indices into src[]; 4 = the sample shifted in from srcn
  if (ich[1] == ich[0]) { S[3] = S[2]; S[2] = S[1]; S[1] = S[0]; N[3] = N[2]; N[2] = N[1]; N[1] = N[0]; }
  if (ich[2] == ich[1]) { S[3] = S[2]; S[2] = S[1]; N[3] = N[2]; N[2] = N[1]; }
  if (ich[3] == ich[2]) { S[3] = S[2]; N[3] = N[2]; }
  const size_t row0 = rowy, row1 = rowy + nexty;      // computeRow(sampler, (0, i.y.x)): absolute x indexing
  int a[4], b[4];
  wr_row_lerp<NCH>(t, row0, row1, (ptrdiff_t)ich[0] + S[k], fracy, a);
  if (N[k] < 4) wr_row_lerp<NCH>(t, row0, row1, (ptrdiff_t)ich[0] + N[k], fracy, b);
  else wr_row_lerp<NCH>(t, row0, row1, (ptrdiff_t)ixn0 + (ixn0 == ich[3] ? 1 : 0), fracy, b);
  for (int c = 0; c < NCH; c++) out[c] = (int16_t)(a[c] + (int16_t)(((int16_t)((b[c] - a[c]) * fch[k])) >> 7));
}

// blendTextureLinear (above) for ALL pixels of an R8 run of `span` pixels, dealt out to the lanes by CHUNK (four pixels): lane
// `wl` of `ws` takes chunks wl, wl + ws, ..  Where a chunk's coordinates come out of a chain of `uv += uv_step` adds -- the
// fallback's lead-in / lead-out and the upscale filter -- every lane steps the chain itself (wave-uniform values: plain adds,
// exactly the reference's loop) and keeps the values at its own chunk, instead of one wr_accum walk per pixel and interpolant:
// the mask rows of a box shadow spent 70 k of a corner-band row's 90 k cycles in those walks (profiles/r06_a_rows_times.txt).
// emit(n, v, count): the values of pixels n .. n + count - 1 of the run (count <= 4).  Same bytes as wr_linear_span_pixel<1>
// pixel by pixel (tests: the mask-row cases run both, WRHIP_NO_MASK_ROWS=1 takes the per-pixel one in the bins).
template <class Emit>
WR_DEVICE void wr_linear_span_lanes_r8(const WrTexDesc& t, const float (&q)[4], const float (&qy)[4], float stepx, float stepy,
                                       float minx, float maxx, float miny, float maxy, int filter, int span, int wl, int ws, Emit&& emit) {
  int before = 0, inside = 0;
  float U[4] = {q[0], q[1], q[2], q[3]};
  if (filter != 1) {
    const float beforeDist = wr_max(0.0f, minx) - U[0];
    if (beforeDist > 0.0f) {
      before = wr_iclamp(int(ceilf(beforeDist / stepx)) * 4, 0, span);
      const float adv = float(before / 4) * stepx;
      for (int i = 0; i < 4; i++) U[i] += adv;
    }
    const float insideDist = wr_min(maxx, float((t.width - 4) * 128)) - U[0];
    if (stepx > 0.0f && insideDist >= stepx) {
      inside = span - before;
      if (filter == 4) inside = wr_imin(int(insideDist * (0.5f / 128.0f)) & ~3, inside);
      else if (filter == 2) inside = wr_imin(int(insideDist / stepx) * 4, inside);
      else inside = wr_imin(int(insideDist * (1.0f / 128.0f)) & ~3, inside);
      if (inside < 0) inside = 0;
    }
  } else before = span;
  // blendTextureLinearFallback over pixels [n0, n1) of the run, its chains starting from (bx, qy)
  auto fallback = [&](const float (&bx)[4], int n0, int n1) {
    const int nch = (n1 - n0 + 3) >> 2;
    float u[4] = {bx[0], bx[1], bx[2], bx[3]}, v[4] = {qy[0], qy[1], qy[2], qy[3]};
    for (int c0 = 0; c0 < nch; c0 += ws) {
      const int mine = c0 + wl, end = wr_imin(c0 + ws, nch);
      float mu[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int c = c0; c < end; c++) {
        if (c == mine) {
#pragma unroll
          for (int i = 0; i < 4; i++) { mu[i] = u[i]; mv[i] = v[i]; }
        }
        if (stepx != 0.0f) {
#pragma unroll
          for (int i = 0; i < 4; i++) u[i] = u[i] + stepx;
        }
        if (stepy != 0.0f) {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = v[i] + stepy;
        }
      }
      if (mine < end) {
        int out[4] = {0, 0, 0, 0};
        const int cnt = wr_imin(4, n1 - n0 - 4 * mine);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k >= cnt) break;
          int o[4];
          wr_bilinear<1>(t, int(wr_clamp(mu[k], minx, maxx)), int(wr_clamp(mv[k], miny, maxy)), o);
          out[k] = o[0];
        }
        emit(n0 + 4 * mine, out, cnt);
      }
    }
  };
  if (before > 0) fallback(q, 0, before);
  if (inside > 0) {
    const int iq0y = int(wr_clamp(qy[0], miny, maxy));
    const int iy = iq0y >> 7, fracy = iq0y & 0x7F;
    const size_t rowy = (size_t)wr_clamp_coord(iy, t.height) * t.stride;
    const size_t nexty = (iy >= 0 && iy < t.height - 1) ? t.stride : 0;
    const int nch = inside >> 2;
    if (filter == 3 || filter == 4) {
      // blendTextureLinearFast / Downscale: constant fractions from lane 0 of the run start; no chain
      const int iq0 = int(wr_clamp(U[0], minx, maxx));
      const int ix = iq0 >> 7;
      const int fracx = wr_frac_x(t, ix, iq0);
      const size_t row0 = rowy + wr_clamp_coord(ix, t.width - 1), row1 = row0 + nexty;
      for (int m = wl; m < nch; m += ws) {
        int out[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int j = 4 * m + k;
          const ptrdiff_t x = filter == 3 ? j : 2 * j;
          int a[4], b[4];
          wr_row_lerp<1>(t, row0, row1, x, fracy, a);
          wr_row_lerp<1>(t, row0, row1, x + 1, fracy, b);
          out[k] = (int16_t)(a[0] + (int16_t)(((int16_t)((b[0] - a[0]) * fracx)) >> 7));
        }
        emit(before + 4 * m, out, 4);
      }
    } else {
      // blendTextureLinearUpscale: chunk m samples at U advanced m times (and the next chunk's first column)
      float u[4] = {U[0], U[1], U[2], U[3]};
      const size_t row0 = rowy, row1 = rowy + nexty;
      for (int c0 = 0; c0 < nch; c0 += ws) {
        const int mine = c0 + wl, end = wr_imin(c0 + ws, nch);
        float mu[4] = {0.0f, 0.0f, 0.0f, 0.0f}, nx0 = 0.0f;
        for (int c = c0; c < end; c++) {
          if (c == mine) {
#pragma unroll
            for (int i = 0; i < 4; i++) mu[i] = u[i];
          }
#pragma unroll
          for (int i = 0; i < 4; i++) u[i] = u[i] + stepx;
          if (c == mine) nx0 = u[0];
        }
        if (mine < end) {
          int ich[4], fch[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            if (mine == 0) {
              const int iq = int(wr_clamp(mu[i], minx, maxx));
              ich[i] = iq >> 7; fch[i] = wr_frac_x(t, ich[i], iq);
            } else {
              const int iq = int(mu[i]);
              ich[i] = iq >> 7; fch[i] = iq & 0x7F;
            }
          }
          const int ixn0 = int(nx0) >> 7;
          int S[4] = {0, 1, 2, 3}, N[4] = {1, 2, 3, 4};
          if (ich[1] == ich[0]) { S[3] = S[2]; S[2] = S[1]; S[1] = S[0]; N[3] = N[2]; N[2] = N[1]; N[1] = N[0]; }
          if (ich[2] == ich[1]) { S[3] = S[2]; S[2] = S[1]; N[3] = N[2]; N[2] = N[1]; }
          if (ich[3] == ich[2]) { S[3] = S[2]; N[3] = N[2]; }
          int out[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            int a[4], b[4];
            wr_row_lerp<1>(t, row0, row1, (ptrdiff_t)ich[0] + S[k], fracy, a);
            if (N[k] < 4) wr_row_lerp<1>(t, row0, row1, (ptrdiff_t)ich[0] + N[k], fracy, b);
            else wr_row_lerp<1>(t, row0, row1, (ptrdiff_t)ixn0 + (ixn0 == ich[3] ? 1 : 0), fracy, b);
            out[k] = (int16_t)(a[0] + (int16_t)(((int16_t)((b[0] - a[0]) * fch[k])) >> 7));
          }
          emit(before + 4 * mine, out, 4);
        }
      }
    }
  }
  if (before + inside < span) {
    const float adv = float(inside / 4) * stepx;
    const float bx[4] = {U[0] + adv, U[1] + adv, U[2] + adv, U[3] + adv};
    fallback(bx, before + inside, span);
  }
}

// ---------------------------------------------------------------------------
// Raster stage.  Lane l of wave w in the workgroup of bin (bx,by) owns pixels
//   x = 64*bx + 4*(l & 15) + i,   y = 64*by + 16*w + (l >> 4) + 4*j,  i,j in 0..3
// i.e. for a fixed j the wave touches 4 consecutive rows x 256 contiguous bytes.

// Per-(prim,row) setup of swgl_commitTexture*RGBA8 as seen by draw_span:
// interpolants at the span start, the filter decision of needsTextureLinear /
// needsNearestFallback, and the nearest-fast row/column clamps.
struct WrTexRow {
  float su, sv;          // per-pixel step of the interpolants (rasterize.h:1003-1017)
  float lu[4], lv[4];    // the interpolant vector (init_interp, glsl.h:3084-3089) at the start of the span shader's sub-span:
                         // lane i belongs to pixel x0 + i.  Without depth runs that is the row's span start; inside depth
                         // run k > 0 the lanes carry the chain of step_interp_inputs() calls of the runs before it
  int x0;                // first pixel of the sub-span (the row's span start, or the start of the depth run)
  int len, span;         // sub-span length; pixels [0,span) go through draw_span, the rest through main()
  int filter;            // 0 nearest-fast, 1 linear fallback, 2 upscale, 3 fast, 4 downscale, -1 unsupported
  int ix, minX, maxX;    // nearest-fast: first texel column and clamps
  int srow;              // nearest-fast: clamped source row
};

WR_DEVICE int wr_run_s(const WrRuns* R, int i) { return R->ext ? R->ext[2 * i] : R->s[i]; }
WR_DEVICE int wr_run_e(const WrRuns* R, int i) { return R->ext ? R->ext[2 * i + 1] : R->e[i]; }
WR_DEVICE int wr_find_run(const WrRuns* R, int x) {
  if (R->ext) {           // (sorted, disjoint: a bisection over the pool's pairs)
    int lo = 0, hi = R->n - 1;
    while (lo <= hi) {
      const int m = (lo + hi) >> 1;
      if (x < R->ext[2 * m]) hi = m - 1;
      else if (x >= R->ext[2 * m + 1]) lo = m + 1;
      else return m;
    }
    return -1;
  }
  for (int i = 0; i < R->n; i++) if (x >= R->s[i] && x < R->e[i]) return i;
  return -1;
}
// The interpolant lanes at the start of depth run k of a row (draw_depth_span, rasterize.h:612-664).  Run 0 is
// initialised like any span (rasterize.h:1003-1017 with span.start moved past the failed pixels); every later run is
// reached through the calls the runs before it made: step_interp_inputs(drawn) after a span shader that consumed the
// whole chunks (DISPATCH_DRAW_SPAN) or one step_interp_inputs() per chunk run by main(), one more for a partial chunk,
// then skip(skip - (4 - partial)).  All of them are `lane += interp_step * (steps * 0.25f)` with interp_step = step * 4.
WR_DEVICE void wr_run_lanes(const WrRuns* R, int k, float L, float step, float xl, bool span_shader, float (&lane)[4]) {
  const float start = float(wr_run_s(R, 0)) + 0.5f - xl;
  lane[0] = L + step * start;
#pragma unroll
  for (int i = 1; i < 4; i++) lane[i] = lane[i - 1] + step;
  const float step4 = step * 4.0f;
  for (int j = 0; j < k; j++) {
    const int n = wr_run_e(R, j) - wr_run_s(R, j), rem = n & 3, full = n >> 2;
    if (full) {
      if (span_shader) {
        const float ch = float(n & ~3) * 0.25f;
#pragma unroll
        for (int i = 0; i < 4; i++) lane[i] = lane[i] + step4 * ch;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) lane[i] = wr_accum(lane[i], step4 * 1.0f, full);
      }
    }
    if (rem) {
#pragma unroll
      for (int i = 0; i < 4; i++) lane[i] = lane[i] + step4 * 1.0f;
    }
    const int skip = wr_run_s(R, j + 1) - wr_run_e(R, j);
    const float ch = float(skip - (rem ? 4 - rem : 0)) * 0.25f;
#pragma unroll
    for (int i = 0; i < 4; i++) lane[i] = lane[i] + step4 * ch;
  }
}

// (Lu, Lv) / (Ru, Rv): the edge interpolants on this row, xl / xr the edges' x, [x0, x0 + len) the row's span.
// `runs` (with the pixel x that is being evaluated): the row's depth runs -- the setup is then that of the run holding x.
WR_DEVICE WrTexRow wr_tex_row_span(const WrPrim& P, const WrTexDesc& t, float Lu, float Lv, float Ru, float Rv, float xl, float xr,
                                   int x0, int len, const WrRuns* runs = nullptr, int x = 0, bool no_span = false) {
  WrTexRow r;
  float stepScale = 1.0f / (xr - xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  r.su = (Ru - Lu) * stepScale; r.sv = (Rv - Lv) * stepScale;
  const bool flat = runs && runs->n < 0;      // a flattened depth row: chunk by chunk through main(), from the span start
  const bool shaded = P.kind == WR_PK_TEX_FS || P.kind == WR_PK_FILTER || P.kind == WR_PK_MIX_BLEND || P.kind == WR_PK_SVG_FILTER || P.kind == WR_PK_QUAD_MASK || P.kind == WR_PK_BORDER_SOLID || P.kind == WR_PK_BORDER_SEGMENT || P.kind == WR_PK_FAST_GRADIENT || P.kind == WR_PK_LINE_DECORATION || no_span || flat;   // no draw_span for this program/target: all main()
  const int k = runs ? wr_find_run(runs, x) : -1;
  if (k >= 0) {
    r.x0 = wr_run_s(runs, k); r.len = wr_run_e(runs, k) - wr_run_s(runs, k);
    wr_run_lanes(runs, k, Lu, r.su, xl, !shaded, r.lu);
    wr_run_lanes(runs, k, Lv, r.sv, xl, !shaded, r.lv);
  } else {
    const float start = float(x0) + 0.5f - xl;
    r.x0 = x0; r.len = len;
    r.lu[0] = Lu + r.su * start; r.lv[0] = Lv + r.sv * start;
#pragma unroll
    for (int i = 1; i < 4; i++) { r.lu[i] = r.lu[i - 1] + r.su; r.lv[i] = r.lv[i - 1] + r.sv; }
  }
  r.span = r.len >= 4 ? (r.len & ~3) : 0;
  r.filter = 0; r.ix = 0; r.minX = 0; r.maxX = 0; r.srow = 0;
  if (shaded) r.span = 0;
  if (r.span == 0 || P.kind == WR_PK_TEX_REPEAT) return r;
  float W = t.sw, H = t.sh;            // samplerScale (texture.h:433-443)
  // lanes 0 and 1 of the uv vector handed to swgl_commitTexture* (shader-side offset included)
  const float p0u = r.lu[0] + P.uv_add[0], p0v = r.lv[0] + P.uv_add[1];
  const float ou1 = r.lu[1] + P.uv_add[0], ov1 = r.lv[1] + P.uv_add[1];
  if (P.kind == WR_PK_TEX_R8) {
    r.filter = 1;   // blendTextureLinearR8 (swgl_ext.h:634-650): always the quantised fallback stepping
  } else if (!t.linear) {
    // swgl_commitTextureNearest: needsNearestFallback (swgl_ext.h:876-880)
    float py0 = p0v * H, py1 = ov1 * H, px0 = p0u * W, px1 = ou1 * W;
    int sp = (r.span & ~127) + 128;
    int scaled = int(roundf((px1 - px0) * float(sp)));
    bool fallback = (py1 - py0) * float(r.span) >= 0.5f || scaled != sp;
    r.filter = fallback ? -1 : 0;   // -1: blendTextureNearestRepeat<BLEND, false>
  } else if (t.width < 2) {
    r.filter = 0;
  } else if (p0v != ov1) {
    r.filter = 1;
  } else {
    // needsTextureLinear (swgl_ext.h:553-587)
    float px0 = p0u * W, px1 = ou1 * W, py0 = p0v * H;
    int sp = (r.span & ~127) + 128;
    int scaled = int(roundf((px1 - px0) * float(sp)));
    if (scaled != sp) {
      r.filter = (px0 < px1 && px1 - px0 <= 1.0f) ? 2 : (scaled == sp * 2 ? 4 : 1);
    } else if ((int(px0 * 4.0f + 0.5f) & 3) != 2 || (int(py0 * 4.0f + 0.5f) & 3) != 2) {
      r.filter = 3;
    } else {
      r.filter = 0;
    }
  }
  if (r.filter == 0) {
    // blendTextureNearestFast (swgl_ext.h:475-537)
    r.ix = int(p0u * W);
    int iy = int(p0v * H);
    int minUx = int(P.uv_bounds[0] * W), minUy = int(P.uv_bounds[1] * H);
    int maxUx = int(P.uv_bounds[2] * W), maxUy = int(P.uv_bounds[3] * H);
    r.srow = wr_clamp_coord(wr_iclamp(iy, minUy, maxUy), t.height);
    r.minX = wr_iclamp(minUx, 0, t.width - 1);
    r.maxX = wr_iclamp(maxUx, r.minX, t.width - 1);
  }
  return r;
}
WR_DEVICE WrTexRow wr_tex_row(const WrPrim& P, const WrTexDesc& t, int y, const WrRuns* runs = nullptr, int x = 0, bool no_span = false) {
  // Edge::nextRow (rasterize.h:878-882) steps the interpolants by repeated addition
  const int k = y - P.y0;
  const bool lin = P.rows_linear != 0;
  const float Lu = wr_row_interp(P.uvL0[0], P.uvLs[0], k, lin), Lv = wr_row_interp(P.uvL0[1], P.uvLs[1], k, lin);
  const float Ru = wr_row_interp(P.uvR0[0], P.uvRs[0], k, lin), Rv = wr_row_interp(P.uvR0[1], P.uvRs[1], k, lin);
  return wr_tex_row_span(P, t, Lu, Lv, Ru, Rv, P.xl, P.xr, P.x0, P.x1 - P.x0, runs, x, no_span);
}

// Quantised (1/128 texel) sample position of a fragment-shader (tail) pixel:
// its init_interp lane (glsl.h:3084-3089), then step_interp_inputs(drawn).
WR_DEVICE void wr_tex_tail_uv(const WrPrim& P, const WrTexRow& r, int n, float& cu, float& cv) {
  // init_interp lane (glsl.h:3084-3089), step_interp_inputs(drawn) once
  // (DISPATCH_DRAW_SPAN), then one step_interp_inputs() per 4-pixel chunk run by main()
  const int lane = (n - r.span) & 3, m = (n - r.span) >> 2;
  float lu = wr_pick4(r.lu, lane), lv = wr_pick4(r.lv, lane);
  if (r.span > 0) {
    float chunks = float(r.span) * 0.25f;
    lu = lu + (r.su * 4.0f) * chunks; lv = lv + (r.sv * 4.0f) * chunks;
  }
  lu = wr_accum(lu, (r.su * 4.0f) * 1.0f, m); lv = wr_accum(lv, (r.sv * 4.0f) * 1.0f, m);
  lu = lu + P.uv_add[0]; lv = lv + P.uv_add[1];
  cu = lu; cv = lv;
  if (P.flags & WR_PF_TAIL_CLAMP) {
    cu = wr_clamp(lu, P.uv_bounds[0], P.uv_bounds[2]); cv = wr_clamp(lv, P.uv_bounds[1], P.uv_bounds[3]);
  }
}

// A fragment-shader (main()) pixel of a textured prim: texture(sColor0, (cu, cv)) -> optional colour
// modulation -> round_pixel.
WR_DEVICE WrWide wr_tex_tail_texel(const WrPrim& P, const WrTexDesc& t, float cu, float cv) {
  const float W = t.sw, H = t.sh;
  const uint32_t* buf = (const uint32_t*)t.ptr;
  float tb, tg, tr, ta;
  if (P.kind == WR_PK_TEX_R8) {
    // textureLinearR8 (texture.h:576-583) -> vec4(r,0,0,1); ps_text_run.glsl:278-283
    // swizzles it to rrrr for COLOR_MODE_ALPHA.
    int qx = int(cu * W * 128.0f + (0.5f - 64.0f)), qy = int(cv * H * 128.0f + (0.5f - 64.0f));
    float m = float(wr_sample_linear_r8(t, qx, qy)) * (1.0f / 255.0f);
    tb = tg = tr = ta = m;
  } else if (t.format == WR_FMT_R8) {
    // texture() of an R8 sampler: vec4(r, 0, 0, 1)
    float m;
    if (t.linear) m = float(wr_sample_linear_r8(t, int(cu * W * 128.0f + (0.5f - 64.0f)), int(cv * H * 128.0f + (0.5f - 64.0f)))) * (1.0f / 255.0f);
    else m = float(((const uint8_t*)t.ptr)[(size_t)wr_clamp_coord(int(cu * W), t.width) + (size_t)wr_clamp_coord(int(cv * H), t.height) * t.stride]) * (1.0f / 255.0f);
    tr = m; tg = 0.0f; tb = 0.0f; ta = 1.0f;
  } else if (t.linear) {
    int qx = int(cu * W * 128.0f + (0.5f - 64.0f)), qy = int(cv * H * 128.0f + (0.5f - 64.0f));
    WrWide s = wr_sample_linear_rgba8(t, qx, qy);
    tb = float(s.bg & 0xFFFF) * (1.0f / 255.0f); tg = float(s.bg >> 16) * (1.0f / 255.0f);
    tr = float(s.ra & 0xFFFF) * (1.0f / 255.0f); ta = float(s.ra >> 16) * (1.0f / 255.0f);
  } else {
    int tx = wr_clamp_coord(int(cu * W), t.width), ty = wr_clamp_coord(int(cv * H), t.height);
    uint32_t p = buf[(size_t)tx + (size_t)ty * t.stride];
    tb = float(p & 0xFF) * (1.0f / 255.0f); tg = float((p >> 8) & 0xFF) * (1.0f / 255.0f);
    tr = float((p >> 16) & 0xFF) * (1.0f / 255.0f); ta = float(p >> 24) * (1.0f / 255.0f);
  }
  if (P.flags & WR_PF_TAIL_MODULATE) { tr = P.fcolor[0] * tr; tg = P.fcolor[1] * tg; tb = P.fcolor[2] * tb; ta = P.fcolor[3] * ta; }
  WrWide s;
  uint32_t pc[2];
  wr_pack_color(wf4{tr, tg, tb, ta}, pc);
  s.bg = pc[0]; s.ra = pc[1];
  return s;
}

// One pixel of a WR_PK_TEX_RGBA8 prim on row y (returns the WideRGBA8 source,
// colour modulation included).
WR_DEVICE WrWide wr_tex_pixel_row(const WrPrim& P, const WrTexDesc& t, const WrTexRow& r, int n) {
  const float W = t.sw, H = t.sh;
  const uint32_t* buf = (const uint32_t*)t.ptr;
  if (n < r.span) {
    WrWide s;
    if (r.filter == 0) {
      int sx = wr_iclamp(r.ix + n, r.minX, r.maxX);
      s = wr_unpack(buf[(size_t)r.srow * t.stride + sx]);
    } else if (r.filter == -1) {
      // blendTextureNearestRepeat<BLEND, false> (swgl_ext.h:774-858): nearest sampler, clamped, any scale
      float pu[4], pv[4];
      for (int i = 0; i < 4; i++) { pu[i] = (r.lu[i] + P.uv_add[0]) * W; pv[i] = (r.lv[i] + P.uv_add[1]) * H; }
      const float stepx = 4.0f * (pu[1] - pu[0]), stepy = 4.0f * (pv[1] - pv[0]);
      const float minx = P.uv_bounds[0] * W, miny = P.uv_bounds[1] * H, maxx = P.uv_bounds[2] * W, maxy = P.uv_bounds[3] * H;
      const bool solid = (int(minx) >= int(maxx) || fabsf(stepx) * float(r.span) * 1.0f < 0.5f) &&
                         (int(miny) >= int(maxy) || fabsf(stepy) * float(r.span) * 1.0f < 0.5f);
      const int k = n & 3, c = solid ? 0 : (n >> 2);
      const float cu = wr_clamp(wr_accum(pu[k], stepx, c), minx, maxx), cv = wr_clamp(wr_accum(pv[k], stepy, c), miny, maxy);
      s = wr_unpack(buf[(size_t)wr_clamp_coord(int(cu), t.width) + (size_t)wr_clamp_coord(int(cv), t.height) * t.stride]);
    } else {
      // Linear filters: exact per-variant evaluation (wr_linear_span_pixel)
      const float qs = 128.0f, qo = 0.5f - 0.5f * qs;
      float q[4], qy[4];
      for (int i = 0; i < 4; i++) { q[i] = (r.lu[i] + P.uv_add[0]) * W * qs + qo; qy[i] = (r.lv[i] + P.uv_add[1]) * H * qs + qo; }
      const float stepx = 4.0f * (q[1] - q[0]), stepy = 4.0f * (qy[1] - qy[0]);
      const float minx = wr_max(P.uv_bounds[0] * W * qs + qo, 0.0f);
      const float miny = wr_max(P.uv_bounds[1] * H * qs + qo, 0.0f);
      const float maxx = wr_max(P.uv_bounds[2] * W * qs + qo, minx);
      const float maxy = wr_max(P.uv_bounds[3] * H * qs + qo, miny);
      int v[4];
      if (P.kind == WR_PK_TEX_R8) {   // expand_mask(buf, r): r in all four channels (blend.h)
        wr_linear_span_pixel<1>(t, q, qy, stepx, stepy, minx, maxx, miny, maxy, 1, r.span, n, v);
        const uint32_t m = uint32_t(v[0]) & 0xFFFF;
        s.bg = s.ra = m | (m << 16);
      } else {
        wr_linear_span_pixel<4>(t, q, qy, stepx, stepy, minx, maxx, miny, maxy, r.filter, r.span, n, v);
        s.bg = (uint32_t(v[0]) & 0xFFFF) | ((uint32_t(v[1]) & 0xFFFF) << 16);
        s.ra = (uint32_t(v[2]) & 0xFFFF) | ((uint32_t(v[3]) & 0xFFFF) << 16);
      }
    }
    if (P.flags & WR_PF_HAS_COLOR) s = wr_apply_color(s, P.color);
    return s;
  }
  // Tail pixels: fragment shader main() -> texture(sColor0, uv) -> round_pixel
  float cu, cv;
  wr_tex_tail_uv(P, r, n, cu, cv);
  return wr_tex_tail_texel(P, t, cu, cv);
}
WR_DEVICE WrWide wr_tex_pixel(const WrPrim& P, const WrTexDesc& t, int x, int y, const WrRuns* runs = nullptr) {
  const WrTexRow r = wr_tex_row(P, t, y, runs, x);
  return wr_tex_pixel_row(P, t, r, x - r.x0);
}

// ---------------------------------------------------------------------------
// brush_image with WR_FEATURE_REPETITION: swgl_commitTextureRepeat[Color]RGBA8 (swgl_ext.h:664-872).
// The span walk is sequential -- runs of chunks that stay inside one tile go through the ordinary
// linear dispatcher / a nearest run, the chunk that straddles a tile edge is sampled with explicit
// repeat math, and the unquantised uv is advanced by float additions in between -- so the owner of
// pixel n replays the walk of its row up to the chunk holding n.
WR_DEVICE int wr_needs_linear(const WrTexDesc& t, float p0u, float p0v, float p1u, float p1v, int span) {   // swgl_ext.h:553-587
  if (t.width < 2) return 0;
  if (p0v != p1v) return 1;
  const float px0 = p0u * t.sw, px1 = p1u * t.sw, py0 = p0v * t.sh;
  const int sp = (span & ~127) + 128;
  const int scaled = int(roundf((px1 - px0) * float(sp)));
  if (scaled != sp) return (px0 < px1 && px1 - px0 <= 1.0f) ? 2 : (scaled == sp * 2 ? 4 : 1);
  if ((int(px0 * 4.0f + 0.5f) & 3) != 2 || (int(py0 * 4.0f + 0.5f) & 3) != 2) return 3;
  return 0;
}
WR_DEVICE int wr_no_repeat_steps(const float (&uv)[4], float uv_step, float tile_repeat, int steps) {   // swgl_ext.h:679-700
  float lo = uv[0], hi = uv[3];
  if (hi < lo) { lo = uv[3]; hi = uv[0]; }
  float limit = floorf(lo) + 1.0f;
  if (tile_repeat > 0.0f) limit = wr_min(limit, tile_repeat);
  if (!(lo >= 0.0f && hi < limit)) return 0;
  return uv_step != 0.0f ? int(wr_clamp((limit - lo) / uv_step, 0.0f, float(steps))) : steps;
}
WR_DEVICE void wr_tile_repeat_uv(float u, float v, const float (&tile_repeat)[2], float& fu, float& fv) {   // tileRepeatUV, :664-673
  if (tile_repeat[0] > 0.0f) {
    u = wr_clamp(u, 0.0f, tile_repeat[0] - 1.0e-6f); v = wr_clamp(v, 0.0f, tile_repeat[1] - 1.0e-6f);
  }
  fu = u - floorf(u); fv = v - floorf(v);
}
// main() of the REPETITION keys for one pixel whose v_uv (times the perspective divisor) is (lu, lv): compute_repeated_uvs
// (brush_image.glsl:318-341), the clamp to v_uv_sample_bounds, texture()
WR_DEVICE WrWide wr_repeat_main(const WrPrim& P, const WrRepeatRec& R, const WrTexDesc& t, float lu, float lv) {
  const float usx = R.uv_repeat[2] - R.uv_repeat[0], usy = R.uv_repeat[3] - R.uv_repeat[1];
  float ru, rv;
  if (R.alpha_pass) {
    const float cu = wr_max(lu, 0.0f), cv = wr_max(lv, 0.0f);
    ru = (cu - floorf(cu)) * usx + R.uv_repeat[0]; rv = (cv - floorf(cv)) * usy + R.uv_repeat[1];
    if (cu >= R.tile_repeat[0]) ru = R.uv_repeat[2];
    if (cv >= R.tile_repeat[1]) rv = R.uv_repeat[3];
  } else {
    ru = (lu - floorf(lu)) * usx + R.uv_repeat[0]; rv = (lv - floorf(lv)) * usy + R.uv_repeat[1];
  }
  ru = wr_clamp(ru, P.uv_bounds[0], P.uv_bounds[2]); rv = wr_clamp(rv, P.uv_bounds[1], P.uv_bounds[3]);
  return wr_tex_tail_texel(P, t, ru, rv);
}
WR_DEVICE WrWide wr_repeat_pixel_row(const WrPrim& P, const WrRepeatRec& R, const WrTexDesc& t, const WrTexRow& r, int n) {
  const int span = R.no_span ? 0 : r.span;
  const float W = t.sw, H = t.sh;
  if (n >= span) {
    // main(): compute_repeated_uvs (brush_image.glsl:318-341), clamp to v_uv_sample_bounds, texture()
    const int lane = (n - span) & 3, m = (n - span) >> 2;
    float lu = wr_pick4(r.lu, lane), lv = wr_pick4(r.lv, lane);
    if (span > 0) {
      const float chunks = float(span) * 0.25f;
      lu = lu + (r.su * 4.0f) * chunks; lv = lv + (r.sv * 4.0f) * chunks;
    }
    lu = wr_accum(lu, (r.su * 4.0f) * 1.0f, m); lv = wr_accum(lv, (r.sv * 4.0f) * 1.0f, m);
    return wr_repeat_main(P, R, t, lu, lv);
  }
  const int k = n & 3, c = n >> 2, total = span >> 2;
  float ux[4], uy[4];
  for (int i = 0; i < 4; i++) { ux[i] = r.lu[i]; uy[i] = r.lv[i]; }
  const float step_x = 4.0f * (ux[1] - ux[0]), step_y = 4.0f * (uy[1] - uy[0]);
  const uint32_t* buf = (const uint32_t*)t.ptr;
  WrWide s;
  if (t.linear) {
    // blendTextureLinearRepeat (swgl_ext.h:702-753)
    const float sc_x = R.uv_repeat[2] - R.uv_repeat[0], sc_y = R.uv_repeat[3] - R.uv_repeat[1];
    int filter = wr_needs_linear(t, ux[0] * sc_x + R.uv_repeat[0], uy[0] * sc_y + R.uv_repeat[1], ux[1] * sc_x + R.uv_repeat[0],
                                 uy[1] * sc_y + R.uv_repeat[1], span);
    if (filter == 0) filter = 3;          // the dispatcher treats LINEAR_FILTER_NEAREST like FAST (swgl_ext.h:423-432)
    const float qs = 128.0f, qo = 0.5f - 0.5f * qs;
    const float qsx = sc_x * W * qs, qsy = sc_y * H * qs;
    const float qox = R.uv_repeat[0] * W * qs + qo, qoy = R.uv_repeat[1] * H * qs + qo;
    const float minx = wr_max(P.uv_bounds[0] * W * qs + qo, 0.0f), miny = wr_max(P.uv_bounds[1] * H * qs + qo, 0.0f);
    const float maxx = wr_max(P.uv_bounds[2] * W * qs + qo, minx), maxy = wr_max(P.uv_bounds[3] * H * qs + qo, miny);
    int v[4] = {0, 0, 0, 0};
    int pos = 0;
    while (pos < total) {
      int steps = total - pos;
      steps = wr_no_repeat_steps(ux, step_x, R.tile_repeat[0], steps);
      if (steps > 0) steps = wr_no_repeat_steps(uy, step_y, R.tile_repeat[1], steps);
      if (steps > 0) {
        if (c < pos + steps) {
          float q[4], qy[4];
          for (int i = 0; i < 4; i++) { q[i] = (ux[i] - floorf(ux[i])) * qsx + qox; qy[i] = (uy[i] - floorf(uy[i])) * qsy + qoy; }
          wr_linear_span_pixel<4>(t, q, qy, step_x * qsx, step_y * qsy, minx, maxx, miny, maxy, filter, steps * 4, n - pos * 4, v);
          break;
        }
        pos += steps;
        if (pos >= total) break;
        const float ax = float(steps) * step_x, ay = float(steps) * step_y;
        for (int i = 0; i < 4; i++) { ux[i] += ax; uy[i] += ay; }
      }
      if (c == pos) {
        float fu, fv;
        wr_tile_repeat_uv(ux[k], uy[k], R.tile_repeat, fu, fv);
        wr_bilinear<4>(t, int(wr_clamp(fu * qsx + qox, minx, maxx)), int(wr_clamp(fv * qsy + qoy, miny, maxy)), v);
        break;
      }
      pos += 1;
      for (int i = 0; i < 4; i++) { ux[i] += step_x; uy[i] += step_y; }
    }
    s.bg = (uint32_t(v[0]) & 0xFFFF) | ((uint32_t(v[1]) & 0xFFFF) << 16);
    s.ra = (uint32_t(v[2]) & 0xFFFF) | ((uint32_t(v[3]) & 0xFFFF) << 16);
  } else {
    // blendTextureNearestRepeat<BLEND, true> (swgl_ext.h:774-858); its uv_rect argument is v_uv_bounds
    const float minx = R.uv_repeat[0] * W, miny = R.uv_repeat[1] * H, maxx = R.uv_repeat[2] * W, maxy = R.uv_repeat[3] * H;
    const float sc_x = maxx - minx, sc_y = maxy - miny;
    float su = 0.0f, sv = 0.0f;
    const bool solid = (int(minx) + 1 >= int(maxx) || fabsf(step_x) * float(span) * sc_x < 0.5f) &&
                       (int(miny) + 1 >= int(maxy) || fabsf(step_y) * float(span) * sc_y < 0.5f);
    if (solid) {
      float fu, fv;
      wr_tile_repeat_uv(ux[k], uy[k], R.tile_repeat, fu, fv);
      su = fu * sc_x + minx; sv = fv * sc_y + miny;
    } else {
      int pos = 0;
      while (pos < total) {
        int steps = total - pos;
        steps = wr_no_repeat_steps(ux, step_x, R.tile_repeat[0], steps);
        if (steps > 0) steps = wr_no_repeat_steps(uy, step_y, R.tile_repeat[1], steps);
        if (steps > 0) {
          if (c < pos + steps) {
            const float iu = (ux[k] - floorf(ux[k])) * sc_x + minx, iv = (uy[k] - floorf(uy[k])) * sc_y + miny;
            su = wr_accum(iu, step_x * sc_x, c - pos); sv = wr_accum(iv, step_y * sc_y, c - pos);
            break;
          }
          pos += steps;
          if (pos >= total) break;
          const float ax = float(steps) * step_x, ay = float(steps) * step_y;
          for (int i = 0; i < 4; i++) { ux[i] += ax; uy[i] += ay; }
        }
        if (c == pos) {
          float fu, fv;
          wr_tile_repeat_uv(ux[k], uy[k], R.tile_repeat, fu, fv);
          su = fu * sc_x + minx; sv = fv * sc_y + miny;
          break;
        }
        pos += 1;
        for (int i = 0; i < 4; i++) { ux[i] += step_x; uy[i] += step_y; }
      }
    }
    s = wr_unpack(buf[(size_t)wr_clamp_coord(int(su), t.width) + (size_t)wr_clamp_coord(int(sv), t.height) * t.stride]);
  }
  if (P.flags & WR_PF_HAS_COLOR) s = wr_apply_color(s, P.color);
  return s;
}
__device__ __noinline__ WrWide wr_repeat_pixel(const WrPrim* Pp, const WrRepeatRec* Rp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrTexDesc& t = D->tex[Pp->tex_slot];
  const WrTexRow r = wr_tex_row(*Pp, t, y, runs, x, Rp->no_span != 0);
  return wr_repeat_pixel_row(*Pp, *Rp, t, r, x - r.x0);
}
// ... under the DUAL_SOURCE_BLENDING key (brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION): main() on every pixel --
// compute_repeated_uvs, the clamp, texture() -- and both colours into the dual-source blend; returns the blended pixel
WR_DEVICE uint32_t wr_dual_blend(const WrPrim& P, const WrDrawDesc* D, int x, int y, uint32_t dstp, const float (&tx)[4], uint32_t cov = 256u, bool aa = false);
WR_DEVICE void wr_texture_rgba_f(const WrTexDesc& t, float cu, float cv, float (&c)[4]);
__device__ __noinline__ uint32_t wr_repeat_dual_pixel(const WrPrim* Pp, const WrRepeatRec* Rp, const WrDrawDesc* D, int x, int y, uint32_t dstp,
                                                      const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp; const WrRepeatRec& R = *Rp;
  const WrTexDesc& t = D->tex[P.tex_slot];
  const WrTexRow r = wr_tex_row(P, t, y, runs, x, true);
  const int n = x - r.x0, lane = n & 3, m = n >> 2;
  float lu = wr_pick4(r.lu, lane), lv = wr_pick4(r.lv, lane);
  lu = wr_accum(lu, (r.su * 4.0f) * 1.0f, m); lv = wr_accum(lv, (r.sv * 4.0f) * 1.0f, m);
  // compute_repeated_uvs (brush_image.glsl:318-341), ALPHA_PASS branch; then the clamp to v_uv_sample_bounds (wr_repeat_main)
  const float usx = R.uv_repeat[2] - R.uv_repeat[0], usy = R.uv_repeat[3] - R.uv_repeat[1];
  const float cu = wr_max(lu, 0.0f), cv = wr_max(lv, 0.0f);
  float ru = (cu - floorf(cu)) * usx + R.uv_repeat[0], rv = (cv - floorf(cv)) * usy + R.uv_repeat[1];
  if (cu >= R.tile_repeat[0]) ru = R.uv_repeat[2];
  if (cv >= R.tile_repeat[1]) rv = R.uv_repeat[3];
  ru = wr_clamp(ru, P.uv_bounds[0], P.uv_bounds[2]); rv = wr_clamp(rv, P.uv_bounds[1], P.uv_bounds[3]);
  float tx[4];
  wr_texture_rgba_f(t, ru, rv, tx);
  return wr_dual_blend(P, D, x, y, dstp, tx);
}

// Compact raster record.  Solid prims on RGBA8 targets drawn without blending
// or with premultiplied-alpha blending of a valid premultiplied colour are
// pre-folded into (K, Clo, Chi) so the raster hot path is  new = hi_bytes(dst*K + C):
//   premultiplied:  src + dst - ((dst*(a+1)) >> 8)  ==  src + ((dst*K + 255) >> 8),  K = 255 - a
//   no blend:       K = 0, src = pack(colour)
// (fields never overflow because the result is <= 255 when b,g,r <= a).
WR_DEVICE WrRec wr_make_rec(const WrPrim& P, int target_format) {
  WrRec r;
  r.x0 = P.x0; r.y0 = P.y0; r.x1 = P.x1; r.y1 = P.y1; r.z = P.z;
  uint32_t kind = uint32_t(P.kind) & 0xFF;
  r.c0 = P.color[0]; r.c1 = P.color[1];
  uint32_t Kf = 0;
  if (kind == WR_PK_SOLID && target_format == WR_FMT_RGBA8 && (P.blend == WR_BLEND_NONE || P.blend == WR_BLEND_PREMULT)) {
    const uint32_t c0 = P.color[0], c1 = P.color[1];
    const bool bytes = ((c0 | c1) & 0xFF00FF00u) == 0;
    uint32_t b = c0 & 0xFFFF, g = c0 >> 16, rr = c1 & 0xFFFF, a = c1 >> 16;
    uint32_t K = 255u - a;
    bool folded = bytes && b <= a && g <= a && rr <= a;
    if (P.blend == WR_BLEND_NONE) { b = wr_pack1(b); g = wr_pack1(g); rr = wr_pack1(rr); a = wr_pack1(a); K = 0; folded = true; }
    if (folded) {
      kind = WR_PK_SOLID_FOLDED;
      r.c0 = 0x00FF00FFu + ((b | (rr << 16)) << 8);
      r.c1 = 0x00FF00FFu + ((g | (a << 16)) << 8);
      Kf = K & 0xFF;
    }
  }
  r.kbf = kind | ((uint32_t(P.blend) & 0xFF) << 8) | ((uint32_t(P.flags) & 0xFF) << 16) | (Kf << 24);
  return r;
}

// Sampling setup of a textured prim (see WrTexRec).
WR_DEVICE WrTexRec wr_make_texrec(const WrPrim& P, const WrTexDesc& tex) {
  WrTexRec t;
  t.ptr = tex.ptr; t.stride = tex.stride; t.wh = uint32_t(tex.width) | (uint32_t(tex.height) << 16);
  const float W = float(tex.width), H = float(tex.height);
  const float qs = 128.0f, qo = 0.5f - 0.5f * qs;
  float stepScale = 1.0f / (P.xr - P.xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  const float Lu = P.uvL0[0] + 0.0f, Ru = P.uvR0[0] + 0.0f;
  t.su = (Ru - Lu) * stepScale;
  const float start = float(P.x0) + 0.5f - P.xl;
  t.ou = Lu + t.su * start;
  const int len = P.x1 - P.x0;
  t.span = len >= 4 ? (len & ~3) : 0;
  const float q0x = t.ou * W * qs + qo, q1x = (t.ou + t.su) * W * qs + qo;
  t.stepx = 4.0f * (q1x - q0x);
  t.minx = wr_max(P.uv_bounds[0] * W * qs + qo, 0.0f);
  t.maxx = wr_max(P.uv_bounds[2] * W * qs + qo, t.minx);
  t.miny = wr_max(P.uv_bounds[1] * H * qs + qo, 0.0f);
  t.maxy = wr_max(P.uv_bounds[3] * H * qs + qo, t.miny);
  t.ub0 = P.uv_bounds[0]; t.ub1 = P.uv_bounds[1]; t.ub2 = P.uv_bounds[2]; t.ub3 = P.uv_bounds[3];
  t.lv0 = P.uvL0[1]; t.lvs = P.uvLs[1]; t.y0 = P.y0;
  const int need = WR_PF_TAIL_CLAMP | WR_PF_TAIL_MODULATE | WR_PF_HAS_COLOR;
  t.simple = (tex.ptr && P.uvLs[0] == 0.0f && P.uvRs[0] == 0.0f && P.uvL0[1] == P.uvR0[1] && P.uvLs[1] == P.uvRs[1] &&
              (P.flags & need) == need && !(P.flags & WR_PF_MASKED) && ((P.color[0] | P.color[1]) & 0xFF00FF00u) == 0 && tex.width >= 2) ? 1 : 0;
  t.fcolor[0] = P.fcolor[0]; t.fcolor[1] = P.fcolor[1]; t.fcolor[2] = P.fcolor[2]; t.fcolor[3] = P.fcolor[3];
  // Exact evaluation of every column / row coordinate of small prims (glyphs):
  // if they all land on texel centres the raster stage needs no float math.
  t.unit = 0; t.ix0 = t.iy0 = 0; t.tix[0] = t.tix[1] = t.tix[2] = 0;
  const int rows = P.y1 - P.y0;
  if (t.simple && len <= 128 && rows <= 128) {
    bool ok = true;
    int q0 = 0;
    for (int n = 0; n < len && ok; n++) {
      const bool tail = n >= t.span;
      const int lane = (tail ? n - t.span : n) & 3;
      float lu = t.ou;
      for (int i = 0; i < lane; i++) lu += t.su;
      int qx;
      if (!tail) {
        qx = int(wr_clamp(wr_accum(lu * W * qs + qo, t.stepx, n >> 2), t.minx, t.maxx));
        if (n == 0) q0 = qx;
        ok = qx == q0 + 128 * n;
      } else {
        if (t.span > 0) lu = lu + (t.su * 4.0f) * (float(t.span) * 0.25f);
        qx = int(wr_clamp(lu, t.ub0, t.ub2) * W * 128.0f + (0.5f - 64.0f));
        const int k = n - t.span;       // no dynamic index: the record must stay in registers
        if (k == 0) t.tix[0] = qx >> 7; else if (k == 1) t.tix[1] = qx >> 7; else t.tix[2] = qx >> 7;
      }
      ok = ok && (qx & 0x7F) == 0 && qx >= 0 && (qx >> 7) <= tex.width - 2;
    }
    int r0 = 0;
    float ov = t.lv0;
    for (int r = 0; r < rows && ok; r++, ov += t.lvs) {
      const int qs_ = int(wr_clamp(ov * H * qs + qo, t.miny, t.maxy));
      const int qt_ = int(wr_clamp(ov, t.ub1, t.ub3) * H * 128.0f + (0.5f - 64.0f));
      if (r == 0) r0 = qs_;
      ok = qs_ == r0 + 128 * r && qt_ == qs_ && (qs_ & 0x7F) == 0 && qs_ >= 0 && (qs_ >> 7) <= tex.height - 1;
    }
    if (ok && rows > 0) { t.unit = 1; t.ix0 = q0 >> 7; t.iy0 = r0 >> 7; }
  }
  return t;
}

// The prim's WrGlyphRec (wrhip_types.h): valid when the raster stage's lane-by-lane glyph walk may apply it -- a unit-texel
// blit with a byte colour under blend NONE / PREMULT, at least four columns wide (the walk loads four columns at once), whose
// tail columns (the < 4 pixels of a row the span shader leaves to main()) are the columns next to the span's.
WR_DEVICE void wr_write_glyph_rec(const WrTargetDesc& T, int gid, const WrPrim& P, const WrTexRec& t) {
  if (!T.grecs) return;
  WrGlyphRec g;
  g.x0 = (int16_t)P.x0; g.y0 = (int16_t)P.y0; g.x1 = (int16_t)P.x1; g.y1 = (int16_t)P.y1;
  g.c0 = P.color[0]; g.c1 = P.color[1];
  g.stride = t.stride;
  const int len = P.x1 - P.x0;
  // (a prim narrower than four columns has no span: every column is a tail column, and the record's ix0 was never set)
  const int ix0 = t.span > 0 ? t.ix0 : t.tix[0];
  g.base = (uint64_t)(unsigned long long)t.ptr + (uint64_t)((long long)(t.iy0 - t.y0) * (long long)t.stride + (long long)(ix0 - P.x0));
  const int tail_x = t.span >= len ? P.x1 : P.x0 + t.span;
  // (a prim narrower than four columns is read from its first column on: up to three bytes past its last one, which must still lie
  // inside the texture's memory)
  const long long th = (long long)(t.wh >> 16), last = (long long)(t.iy0 + (P.y1 - P.y0) - 1) * (long long)t.stride + (long long)ix0 + 3;
  bool ok = t.simple != 0 && t.unit != 0 && len >= 1 && (len >= 4 || last < th * (long long)t.stride) && (P.blend == WR_BLEND_NONE || P.blend == WR_BLEND_PREMULT) &&
            P.x0 >= -32768 && P.y0 >= -32768 && P.x1 <= 32767 && P.y1 <= 32767;
  for (int k = 0; k < 3; k++)
    if (tail_x + k < P.x1) ok = ok && (k == 0 ? t.tix[0] : (k == 1 ? t.tix[1] : t.tix[2])) == ix0 + (tail_x - P.x0) + k;
  g.fcolor[0] = t.fcolor[0]; g.fcolor[1] = t.fcolor[1]; g.fcolor[2] = t.fcolor[2]; g.fcolor[3] = t.fcolor[3];
  g.info = ok ? (1u | ((P.flags & WR_PF_DEPTH_TEST) ? 2u : 0u) | (uint32_t(P.blend & 0xFF) << 8) | (uint32_t(uint16_t(int16_t(tail_x))) << 16)) : 0u;
  ((WrGlyphRec*)T.grecs)[gid] = g;
#ifdef WRHIP_HOSTSIM
  if (getenv("WRHIP_DEBUG_GLYPHS")) {
    static int n_ok = 0, n_bad = 0, n_nounit = 0, n_narrow = 0;
    if (ok) n_ok++; else { n_bad++; if (!t.unit || !t.simple) n_nounit++; if (len < 4) n_narrow++; }
    if (((n_ok + n_bad) & 1023) == 0 || getenv("WRHIP_DEBUG_GLYPHS")[0] == '2') fprintf(stderr, "glyph recs: %d eligible, %d not (%d not unit / simple, %d narrower than 4)\n", n_ok, n_bad, n_nounit, n_narrow);
  }
#endif
}

// Vertex stage of one instance: locate its draw, run the shader's vertex
// function, then swgl's draw_quad setup.
#ifdef WRHIP_TIMING
__device__ unsigned long long wr_dbg_mid[16384];
__device__ unsigned wr_dbg_prim[16384 * 4];
#endif
WR_DEVICE void wr_vertex_prim(const WrDescView& V, int n_draws, const uint8_t* __restrict__ arena,
                              int gid, WrPrim& P, WrAux* aux, WrUnsupportedCounters* cnt) {
  P.blend = 0; P.flags = 0; P.z = 0; P.color[0] = P.color[1] = 0; P.tex_slot = 0;
  P.uv_add[0] = P.uv_add[1] = 0.0f; P.rows_linear = 0; P.dual = 0; P.dual_swz = 0.0f;
  P.uv_add[0] = P.uv_add[1] = 0.0f; P.rows_linear = 0; P.dual = 0; P.dual_swz = 0.0f;
  // the draw containing this instance: the host's per-block table gives the draw at the start of
  // the 64-prim block, the rest is a short forward scan (a binary search over the draws was
  // log2(n) dependent round trips before the first useful load)
  int lo = V.lo0;
  while (lo + 1 < n_draws && V.draw(lo + 1).first_prim <= gid) lo++;
  const WrDrawDesc& d = V.draw(lo);
  int inst = gid - d.first_prim;
  WR_TP(2);
  if (inst >= d.count) {   // padding slot between two targets (targets start on 64-prim boundaries)
    P.kind = WR_PK_NONE; P.x0 = P.x1 = P.y0 = P.y1 = 0; P.draw = lo;
    return;
  }
  if (d.shader == WR_SH_CLEAR_OP) {
    P.kind = WR_PK_CLEAR; P.blend = WR_BLEND_NONE; P.draw = lo; P.z = d.clear_depth;
    P.flags = d.flags & (WR_PF_CLEAR_COLOR | WR_PF_CLEAR_DEPTH);
    P.x0 = d.clip[0]; P.y0 = d.clip[1]; P.x1 = d.clip[2]; P.y1 = d.clip[3];
    P.color[0] = d.clear_color; P.color[1] = 0;
    if (P.x1 <= P.x0 || P.y1 <= P.y0) P.kind = WR_PK_NONE;
    return;
  }
  WrVsOut o;
  o.tex_slot = 0; o.uv_bounds = wf4{0, 0, 0, 0}; o.tail_clamp = 0; o.tail_modulate = 0;
  o.uv_add[0] = o.uv_add[1] = 0.0f;
  o.blend_override = 0; o.blend_color = wf4{0, 0, 0, 0}; o.persp_div = -1.0f; o.dual = 0; o.dual_swz = 0.0f;
  o.cd[0] = o.cd[1] = 0; o.cd[2] = o.cd[3] = -1;
  switch (d.shader) {
    case WR_SH_PS_QUAD_TEXTURED: wr_vs_ps_quad_textured(d, arena, inst, o); break;
    case WR_SH_PS_QUAD_MASK: wr_vs_ps_quad_textured(d, arena, inst, o, 1, &aux[gid].clip); break;
    case WR_SH_PS_QUAD_MASK_FAST: wr_vs_ps_quad_textured(d, arena, inst, o, 2, &aux[gid].clip); break;
    case WR_SH_PS_QUAD_RADIAL_GRADIENT: wr_vs_ps_quad_textured(d, arena, inst, o, 3, nullptr, &aux[gid].grad); break;
    case WR_SH_PS_QUAD_CONIC_GRADIENT: wr_vs_ps_quad_textured(d, arena, inst, o, 4, nullptr, &aux[gid].grad); break;
    case WR_SH_BRUSH_SOLID:
    case WR_SH_BRUSH_SOLID_ALPHA: wr_vs_brush(d, arena, inst, 0, o); break;
    case WR_SH_BRUSH_IMAGE: wr_vs_brush(d, arena, inst, 1, o); break;
    case WR_SH_BRUSH_IMAGE_ALPHA: wr_vs_brush(d, arena, inst, 2, o); break;
    case WR_SH_BRUSH_IMAGE_DUAL: wr_vs_brush(d, arena, inst, 9, o); break;
    case WR_SH_BRUSH_OPACITY:
    case WR_SH_BRUSH_OPACITY_ALPHA: wr_vs_brush(d, arena, inst, 7, o); break;
    case WR_SH_BRUSH_IMAGE_REPEAT: wr_vs_brush(d, arena, inst, 5, o, nullptr, nullptr, &aux[gid].rep); break;
    case WR_SH_BRUSH_IMAGE_REPEAT_ALPHA: wr_vs_brush(d, arena, inst, 6, o, nullptr, nullptr, &aux[gid].rep); break;
    case WR_SH_BRUSH_IMAGE_REPEAT_DUAL: wr_vs_brush(d, arena, inst, 11, o, nullptr, nullptr, &aux[gid].rep); break;
    case WR_SH_BRUSH_LINEAR_GRADIENT:
    case WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA: wr_vs_brush(d, arena, inst, 3, o, &aux[gid].grad); break;
    case WR_SH_BRUSH_BLEND:
    case WR_SH_BRUSH_BLEND_ALPHA: wr_vs_brush(d, arena, inst, 4, o, nullptr, &aux[gid].filt); break;
    case WR_SH_BRUSH_MIX_BLEND:
    case WR_SH_BRUSH_MIX_BLEND_ALPHA: wr_vs_brush(d, arena, inst, 8, o, nullptr, nullptr, nullptr, &aux[gid].mix); break;
    case WR_SH_BRUSH_YUV:
    case WR_SH_BRUSH_YUV_ALPHA: wr_vs_brush(d, arena, inst, 10, o, nullptr, nullptr, nullptr, nullptr, &aux[gid].yuv); break;
    case WR_SH_COMPOSITE_YUV: wr_vs_composite_yuv(d, arena, inst, o, aux[gid].yuv); break;
    case WR_SH_COMPOSITE: wr_vs_composite(d, arena, inst, false, o); break;
    case WR_SH_COMPOSITE_FAST: wr_vs_composite(d, arena, inst, true, o); break;
    case WR_SH_PS_CLEAR: wr_vs_ps_clear(d, arena, inst, o); break;
    case WR_SH_PS_TEXT_RUN: wr_vs_ps_text_run(d, arena, inst, false, o); break;
    case WR_SH_PS_TEXT_RUN_DUAL: wr_vs_ps_text_run(d, arena, inst, true, o); break;
    case WR_SH_PS_TEXT_RUN_GT: wr_vs_ps_text_run(d, arena, inst, false, o, true); break;
    case WR_SH_PS_TEXT_RUN_DUAL_GT: wr_vs_ps_text_run(d, arena, inst, true, o, true); break;
    case WR_SH_CS_BLUR_ALPHA: wr_vs_cs_blur(d, arena, inst, o, aux[gid].blur); break;
    case WR_SH_CS_BLUR_COLOR: wr_vs_cs_blur(d, arena, inst, o, aux[gid].blur); break;
    case WR_SH_CS_CLIP_RECT: wr_vs_cs_clip_rect(d, arena, inst, false, o, aux[gid].clip); break;
    case WR_SH_CS_CLIP_RECT_FAST: wr_vs_cs_clip_rect(d, arena, inst, true, o, aux[gid].clip); break;
    case WR_SH_CS_CLIP_BOX_SHADOW: wr_vs_cs_clip_box_shadow(d, arena, inst, o, aux[gid].box); break;
    case WR_SH_CS_SCALE: wr_vs_cs_scale(d, arena, inst, V.target(d.target).format, o); break;
    case WR_SH_PS_COPY: wr_vs_ps_copy(d, arena, inst, o); break;
    case WR_SH_CS_SVG_FILTER: wr_vs_cs_svg_filter(d, arena, inst, false, o, aux[gid].svg); break;
    case WR_SH_CS_SVG_FILTER_NODE: wr_vs_cs_svg_filter(d, arena, inst, true, o, aux[gid].svg); break;
    case WR_SH_PS_SPLIT_COMPOSITE: wr_vs_ps_split_composite(d, arena, inst, o); break;
    case WR_SH_CS_BORDER_SOLID: wr_vs_cs_border_solid(d, arena, inst, o, aux[gid].border); break;
    case WR_SH_CS_BORDER_SEGMENT: wr_vs_cs_border_segment(d, arena, inst, o, aux[gid].bseg); break;
    case WR_SH_CS_FAST_LINEAR_GRADIENT: wr_vs_cs_fast_linear_gradient(d, arena, inst, o, aux[gid].fgrad); break;
    case WR_SH_CS_LINE_DECORATION: wr_vs_cs_line_decoration(d, arena, inst, o, aux[gid].line); break;
    case WR_SH_CS_LINEAR_GRADIENT: wr_vs_cs_linear_gradient(d, arena, inst, o, &aux[gid].grad); break;
    case WR_SH_CS_RADIAL_GRADIENT: wr_vs_cs_radial_gradient(d, arena, inst, o, &aux[gid].grad); break;
    case WR_SH_CS_CONIC_GRADIENT: wr_vs_cs_conic_gradient(d, arena, inst, o, &aux[gid].grad); break;
    default:
      P.kind = WR_PK_NONE; P.x0 = P.x1 = P.y0 = P.y1 = 0; P.draw = lo; P.blend = 0; P.flags = 0; P.z = 0;
      P.color[0] = P.color[1] = 0;
      return;
  }
#ifdef WRHIP_TIMING
  if (gid < 16384) wr_dbg_mid[gid] = wall_clock64();
#endif
  WR_TP(6);
  wr_finish_prim(d, lo, o, P, &aux[gid], cnt);
  if ((P.kind == WR_PK_SOLID_QUAD || P.kind == WR_PK_TEX_QUAD) && aux[gid].quad.nseg < 0) {
    // a perspective prim that reaches the camera plane: clip_side, then the polygon's walk (wr_persp_clipped_walk)
    int bx0, by0, bx1, by1;
    float cx0, cy0, cx1, cy1;
    bool ok = wr_persp_clipped_walk(aux[gid].quad, bx0, by0, bx1, by1, cx0, cy0, cx1, cy1);
    if (ok) {
      P.x0 = wr_imax(bx0, int(cx0)); P.x1 = wr_imin(bx1, int(cx1)); P.y0 = wr_imax(by0, int(cy0)); P.y1 = wr_imin(by1, int(ceilf(cy1)));
      ok = P.x1 > P.x0 && P.y1 > P.y0;
    }
    if (!ok) { P.kind = WR_PK_NONE; P.x0 = P.x1 = P.y0 = P.y1 = 0; aux[gid].quad.nseg = 0; }
  }
  if (P.kind == WR_PK_SOLID_QUAD || P.kind == WR_PK_TEX_QUAD) wr_quad_build_rowtab(&V.target(d.target), &P, &aux[gid].quad);
  if ((P.kind == WR_PK_SOLID_QUAD || P.kind == WR_PK_TEX_QUAD) && aux[gid].quad.pad != 0 && (P.flags & WR_PF_DEPTH_TEST)) {
    // a depth-tested perspective prim: the rows its spans touch are flattened from here on (WrTargetDesc::flat_rows)
    uint32_t* fr = V.target(d.target).flat_rows;
    if (fr) {
      bool any = false;
      for (int y = P.y0; y < P.y1; y++) {
        int s0, s1;
        if (wr_quad_row_span(aux[gid].quad, y, s0, s1) && wr_imin(s1, P.x1) > wr_imax(s0, P.x0)) { atomicMin(&fr[y], (uint32_t)gid); any = true; }
      }
      if (any) atomicMin(&fr[V.target(d.target).height], (uint32_t)gid);
    }
  }
  if (d.query_slot >= 0 && P.kind != WR_PK_NONE && P.kind != WR_PK_UNSUPPORTED) {
    // GL_SAMPLES_PASSED: ctx->shaded_pixels += span.len() for every row with a non-empty span, before any depth test
    unsigned long long n = 0;
    if (P.kind == WR_PK_SOLID_QUAD || P.kind == WR_PK_TEX_QUAD) {
      for (int y = P.y0; y < P.y1; y++) { int s0, s1; if (wr_quad_row_span(aux[gid].quad, y, s0, s1)) { s0 = wr_imax(s0, P.x0); s1 = wr_imin(s1, P.x1); if (s1 > s0) n += (unsigned long long)(s1 - s0); } }
    } else {
      n = (unsigned long long)(P.x1 - P.x0) * (unsigned long long)(P.y1 - P.y0);
    }
    if (n) atomicAdd(&cnt->samples[d.query_slot & (WR_QUERY_SLOTS - 1)], n);
  }
}


// Binning: bit (p - T.first_prim) of bin b's mask row <=> prim p touches bin b.
//
// A wave holds 64 consecutive prims, i.e. (part of) one 64-bit mask word per bin, and "prim
// touches bin (bx, by)" is separable: [bx0 <= bx <= bx1] AND [by0 <= by <= by1].  So the word of
// every bin is built 64 prims at a time from ballots:  X[bx] = ballot(bx0 <= bx <= bx1) for the
// columns and Y[by] for the rows of the group's union bin box, parked one per lane,
// and the write phase lets lane l own bin l of the box: word = X[col] & Y[row]
// via two ds_bpermute pairs, one atomicOr per non-empty bin.  A tile-sized batch (16 x 8 bins)
// costs ~200 wave instructions however many bins its prims span -- the per-prim loops this
// replaces spent ~100 per prim (integer divisions, one wave pass per large prim).
// Lanes of a wave are grouped by (target, word): groups are processed one after the other.
WR_DEVICE void wr_bin_prim(const WrPrim& P, bool valid, int gid, const WrDescView& V, unsigned long long* masks) {
  const WrTargetDesc* targets = V.targets;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, tgt = -1, rel = 0;
  if (valid && P.kind != WR_PK_NONE && P.kind != WR_PK_UNSUPPORTED) {
    tgt = V.draw(P.draw).target;
    const WrTargetDesc& T = V.target(tgt);
    bx0 = wr_imax(P.x0, 0) / WR_BIN_W; bx1 = (wr_imin(P.x1, T.width) - 1) / WR_BIN_W;
    by0 = wr_imax(P.y0, 0) / WR_BIN_H; by1 = (wr_imin(P.y1, T.height) - 1) / WR_BIN_H;
    by0 = wr_imax(by0, T.y_begin / WR_BIN_H); by1 = wr_imin(by1, (T.y_end - 1) / WR_BIN_H);
    rel = gid - T.first_prim;
    if (T.rows_mode) bx1 = -1;        // a span-rows target has no bins (wr_span_rows_kernel walks its prim range per row)
  }
  const bool has = tgt >= 0 && bx1 >= bx0 && by1 >= by0;
#ifdef WRHIP_HOSTSIM
  if (has) {
    const WrTargetDesc& T = targets[tgt];
    unsigned long long* base = masks + (size_t)T.word_base + (rel >> 6);
    for (int by = by0; by <= by1; by++)
      for (int bx = bx0; bx <= bx1; bx++) atomicOr(&base[(size_t)(by * T.bins_x + bx) * T.words_per_bin], 1ull << (rel & 63));
  }
#else
  typedef short wr_s2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(has);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const int s_tgt = __builtin_amdgcn_readlane(tgt, leader), s_word = __builtin_amdgcn_readlane(rel >> 6, leader);
    const bool mine = has && tgt == s_tgt && (rel >> 6) == s_word;
    todo &= ~__ballot(mine);
    // bit of lane L is (rel & 63) = L + shift, the same shift for the whole group (consecutive prims)
    const int s_shift = __builtin_amdgcn_readlane((rel & 63) - lane, leader);
    const WrTargetDesc& T = V.target(s_tgt);
    const int bins_x = T.bins_x, wpb = T.words_per_bin;
    unsigned long long* base = masks + (size_t)T.word_base + s_word;
    // union bin box of the group: packed 16-bit min / max butterflies
    wr_s2 lo2 = {(short)(mine ? bx0 : 32767), (short)(mine ? by0 : 32767)};
    wr_s2 hi2 = {(short)(mine ? bx1 : -1), (short)(mine ? by1 : -1)};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const wr_s2 ol = __builtin_bit_cast(wr_s2, __shfl_xor(__builtin_bit_cast(int, lo2), d));
      const wr_s2 oh = __builtin_bit_cast(wr_s2, __shfl_xor(__builtin_bit_cast(int, hi2), d));
      lo2 = __builtin_elementwise_min(lo2, ol); hi2 = __builtin_elementwise_max(hi2, oh);
    }
    const int ux0 = __builtin_amdgcn_readfirstlane((int)lo2.x), uy0 = __builtin_amdgcn_readfirstlane((int)lo2.y);
    const int ux1 = __builtin_amdgcn_readfirstlane((int)hi2.x), uy1 = __builtin_amdgcn_readfirstlane((int)hi2.y);
    for (int cx0 = ux0; cx0 <= ux1; cx0 += 64) {
      const int W = wr_imin(64, ux1 - cx0 + 1);
      unsigned xlo = 0, xhi = 0;                 // lane i: X[cx0 + i]
      for (int i = 0; i < W; i++) {
        const unsigned long long m = __ballot(mine && bx0 <= cx0 + i && bx1 >= cx0 + i);
        xlo = lane == i ? (unsigned)m : xlo;
        xhi = lane == i ? (unsigned)(m >> 32) : xhi;
      }
      const float rW = 1.0f / float(W);
      for (int ry0 = uy0; ry0 <= uy1; ry0 += 64) {
        const int H = wr_imin(64, uy1 - ry0 + 1);
        unsigned ylo = 0, yhi = 0;               // lane j: Y[ry0 + j]
        for (int j = 0; j < H; j++) {
          const unsigned long long m = __ballot(mine && by0 <= ry0 + j && by1 >= ry0 + j);
          ylo = lane == j ? (unsigned)m : ylo;
          yhi = lane == j ? (unsigned)(m >> 32) : yhi;
        }
        for (int idx = lane; idx < W * H + lane; idx += 64) {   // trip count is wave-uniform (shuffles inside)
          const bool live = idx < W * H;
          const int row = live ? int((float(idx) + 0.5f) * rW) : 0;   // exact: idx < 4096, W <= 64
          const int col = live ? idx - row * W : 0;
          const unsigned wlo = (unsigned)__shfl((int)xlo, col) & (unsigned)__shfl((int)ylo, row);
          const unsigned whi = (unsigned)__shfl((int)xhi, col) & (unsigned)__shfl((int)yhi, row);
          unsigned long long w = ((unsigned long long)whi << 32) | wlo;
          w = s_shift >= 0 ? (w << s_shift) : (w >> -s_shift);
          if (live && w) atomicOr(&base[(size_t)((ry0 + row) * bins_x + cx0 + col) * wpb], w);
        }
      }
    }
  }
#endif
}

// Nearest-fast setup of an axis-aligned WR_PK_TEX_RGBA8 prim: the row-independent
// part of what blendTextureNearestFast / needsTextureLinear (swgl_ext.h:475-587)
// decide, evaluated once per prim here instead of per pixel in the raster stage.  Returns false when the x part of the decision already rules
// the nearest-fast path out (scaled or subpixel-offset sampling).
WR_DEVICE bool wr_texrow_x_setup(const WrPrim& P, const WrTexDesc& t, WrTexRec& T, WrUnsupportedCounters* cnt) {
  if (!t.ptr || P.uvLs[0] != 0.0f || P.uvRs[0] != 0.0f || P.uvL0[1] != P.uvR0[1] || P.uvLs[1] != P.uvRs[1]) return false;
  if (t.sw != float(t.width) || t.sh != float(t.height)) return false;       // (sampler2DRect: the general row setup, WrTexRec keeps no sampler scale)
  const WrTexRow r = wr_tex_row(P, t, P.y0);       // u, su, span and the x clamps do not depend on the row here
  if (r.span == 0) return false;
  const float W = float(t.width);
  bool x_ok;
  const float px0 = (r.lu[0] + P.uv_add[0]) * W, px1 = (r.lu[1] + P.uv_add[0]) * W;
  const int sp = (r.span & ~127) + 128;
  const int scaled = int(roundf((px1 - px0) * float(sp)));
  if (!t.linear) x_ok = scaled == sp;
  else if (t.width < 2) x_ok = true;
  else x_ok = scaled == sp && (int(px0 * 4.0f + 0.5f) & 3) == 2;
  if (!x_ok) return false;
  __builtin_memset(&T, 0, sizeof(T));
  T.ptr = t.ptr; T.stride = t.stride; T.wh = uint32_t(t.width) | (uint32_t(t.height) << 16);
  T.span = r.span; T.y0 = P.y0;
  T.ix0 = int((r.lu[0] + P.uv_add[0]) * W);
  const int minUx = int(P.uv_bounds[0] * W), maxUx = int(P.uv_bounds[2] * W);
  T.tix[0] = wr_iclamp(minUx, 0, t.width - 1);
  T.tix[1] = wr_iclamp(maxUx, T.tix[0], t.width - 1);
  T.lv0 = P.uvL0[1]; T.lvs = P.uvLs[1];
  T.su = P.uv_add[1];                            // v offset applied to every row before sampling
  T.ub1 = P.uv_bounds[1]; T.ub3 = P.uv_bounds[3];
  T.unit = (t.linear && t.width >= 2) ? 1 : 0;      // rows must also pass the texel-centre test
  T.simple = 2;
  // Rows that step exactly one texel per target row (tile composites): with a
  // power-of-two height, v(k) = v0 + k/H holds exactly (rows_linear), so
  // int(v(k)*H) = int(v0*H) + k and the texel-centre test gives the same answer on every row.
  const int th = t.height;
  const float H = float(th);
  const float vstep = P.uvLs[1] * H;             // exact when H is a power of two
#ifdef WRHIP_HOSTSIM
  if (getenv("WRHIP_DEBUG_ROWS")) fprintf(stderr, "texrow: lin %d th %d vstep %.9g L0*H %.9g rows %d\n", P.rows_linear, th, vstep, P.uvL0[1] * H, P.y1 - P.y0);
#endif
  if (P.rows_linear && P.uv_add[1] == 0.0f && (th & (th - 1)) == 0 && th < 65536 && (vstep == 1.0f || vstep == -1.0f) && P.y1 - P.y0 < 65536) {
    const float py0 = P.uvL0[1] * H;           // exact: scaling by a power of two
    const float pyl = fabsf(py0) + float(P.y1 - P.y0);
    const float pye = py0 + vstep * float(P.y1 - P.y0 - 1);                        // last row; every row keeps v*H >= 0
    const bool exact = pyl < 4194304.0f && (py0 * 4.0f) == floorf(py0 * 4.0f) && py0 >= 0.0f && pye >= 0.0f;   // quarter-texel grid survives the adds
    if (exact && (!T.unit || (int(py0 * 4.0f + 0.5f) & 3) == 2)) {
      const int minUy = int(P.uv_bounds[1] * H), maxUy = int(P.uv_bounds[3] * H);
      if (minUy <= maxUy) {
        const int lo = wr_iclamp(minUy, 0, th - 1), hi = wr_iclamp(maxUy, 0, th - 1);
        T.iy0 = int(py0);
        T.tix[2] = vstep > 0.0f ? 1 : -1;
        T.unit = lo | (hi << 16);
        T.simple = 3;
      }
    }
  }
  return true;
}
WR_DEVICE int wr_texrow_entry(const WrTexRec& T, float ov_raw) {
  const int th = int(T.wh >> 16);
  const float H = float(th);
  const float ov = ov_raw + T.su;
  const float py0 = ov * H;
  if (T.unit && (int(py0 * 4.0f + 0.5f) & 3) != 2) return -1;
  const int iy = int(ov * H), minUy = int(T.ub1 * H), maxUy = int(T.ub3 * H);
  return wr_clamp_coord(wr_iclamp(iy, minUy, maxUy), th);
}

WR_DEVICE float wr_step01(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
// ---- cs_clip_box_shadow row setup (used by the setup stage for the mask-row key of a prim's middle row, by the rows kernel and
// by the in-raster evaluation) ----
// row interpolants of a cs_clip_box_shadow prim: c = 0,1 vUv; 2,3 vLocalPos.xy
struct WrRowVals { float o[4], s[4]; };
WR_DEVICE WrRowVals wr_box_row_vals(const WrPrim& P, const WrBoxRec& B, int y, const WrAccTabs* tabs = nullptr) {
  WrRowVals rv;
  const int k = y - P.y0;
  const bool lin = P.rows_linear != 0;     // (the vLocalPos interpolants share the uv ones' linearity in practice; wr_accum checks)
  float stepScale = 1.0f / (P.xr - P.xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  const float start = float(P.x0) + 0.5f - P.xl;
  float L0, L1, R0, R1, L2, L3, R2, R3;
  if (tabs) {
    L0 = wr_acctabs_row_any(tabs, 0, k); L1 = wr_acctabs_row_any(tabs, 1, k); R0 = wr_acctabs_row_any(tabs, 2, k); R1 = wr_acctabs_row_any(tabs, 3, k);
    L2 = wr_acctabs_row_any(tabs, 4, k); L3 = wr_acctabs_row_any(tabs, 5, k); R2 = wr_acctabs_row_any(tabs, 6, k); R3 = wr_acctabs_row_any(tabs, 7, k);
  } else {
    L0 = wr_row_interp(P.uvL0[0], P.uvLs[0], k, lin); L1 = wr_row_interp(P.uvL0[1], P.uvLs[1], k, lin);
    R0 = wr_row_interp(P.uvR0[0], P.uvRs[0], k, lin); R1 = wr_row_interp(P.uvR0[1], P.uvRs[1], k, lin);
    L2 = wr_accum(B.lpL0[0], B.lpLs[0], k); L3 = wr_accum(B.lpL0[1], B.lpLs[1], k);
    R2 = wr_accum(B.lpR0[0], B.lpRs[0], k); R3 = wr_accum(B.lpR0[1], B.lpRs[1], k);
  }
  rv.s[0] = (R0 - L0) * stepScale; rv.o[0] = L0 + rv.s[0] * start;
  rv.s[1] = (R1 - L1) * stepScale; rv.o[1] = L1 + rv.s[1] * start;
  rv.s[2] = (R2 - L2) * stepScale; rv.o[2] = L2 + rv.s[2] * start;
  rv.s[3] = (R3 - L3) * stepScale; rv.o[3] = L3 + rv.s[3] * start;
  return rv;
}

#ifndef WRHIP_HOSTSIM
// The same for a row the whole wave works on (wr_mask_rows_body: y is wave-uniform): the eight sums -- each a walk over the
// binades its running sum passes through when the closed form does not apply, ~5 k cycles -- are taken by eight lanes at once
// instead of one after the other on values every lane holds (measured: 38 k of a box-shadow row's 118 k cycles, cfg4).
WR_DEVICE WrRowVals wr_box_row_vals_wave(const WrPrim& P, const WrBoxRec& B, int y, int lane, const WrAccTabs* tabs = nullptr) {
  WrRowVals rv;
  const int k = y - P.y0;
  const bool lin = P.rows_linear != 0;
  float stepScale = 1.0f / (P.xr - P.xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  const float start = float(P.x0) + 0.5f - P.xl;
  const int i = lane & 7;
  const float s0 = i == 0 ? P.uvL0[0] : i == 1 ? P.uvL0[1] : i == 2 ? P.uvR0[0] : i == 3 ? P.uvR0[1] : i == 4 ? B.lpL0[0] : i == 5 ? B.lpL0[1] : i == 6 ? B.lpR0[0] : B.lpR0[1];
  const float st = i == 0 ? P.uvLs[0] : i == 1 ? P.uvLs[1] : i == 2 ? P.uvRs[0] : i == 3 ? P.uvRs[1] : i == 4 ? B.lpLs[0] : i == 5 ? B.lpLs[1] : i == 6 ? B.lpRs[0] : B.lpRs[1];
  float r;
  if (tabs) {
    // (lane i reads sum i off the prim's tables; a right-edge sum marked equal to its left one takes that lane's value)
    const int m = tabs->mode[i];
    r = wr_acctabs_row(tabs, i, k);
    const float other = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane & 63) - 2) << 2, __builtin_bit_cast(int, r)));
    if (m == 4) r = other;
  } else r = wr_row_interp(s0, st, k, lin && i < 4);
  const float L0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0)), L1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 1));
  const float R0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 2)), R1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 3));
  const float L2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 4)), L3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 5));
  const float R2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 6)), R3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 7));
  rv.s[0] = (R0 - L0) * stepScale; rv.o[0] = L0 + rv.s[0] * start;
  rv.s[1] = (R1 - L1) * stepScale; rv.o[1] = L1 + rv.s[1] * start;
  rv.s[2] = (R2 - L2) * stepScale; rv.o[2] = L2 + rv.s[2] * start;
  rv.s[3] = (R3 - L3) * stepScale; rv.o[3] = L3 + rv.s[3] * start;
  return rv;
}
#endif

// Span-level setup of a cs_clip_box_shadow row (cs_clip_box_shadow.glsl:150-250): where the shadow rect and the four
// nine-patch sector boundaries fall along the row, as remaining span lengths.  Prim and row only: evaluated by the
// row-owning lanes of a wave and handed round (see WrClipRow).
struct WrBoxRow {
  int ss_se, os01, os23;       // shadow_start_len | shadow_end_len << 16, os0 | os1 << 16, os2 | os3 << 16
  int xc;                      // [pa, pb) = xc & 0xFFFF, xc >> 16: the run of the row in which u is clamped to the nine-patch's stretched
                               // middle column -- every pixel of it samples the same texel, so it has ONE value,
  uint32_t vrow;               // ... this one (wr_box_row_finish evaluates a single pixel of the run)
};
WR_DEVICE WrBoxRow wr_box_row_setup(const WrPrim& P, const WrBoxRec& B, const WrRowVals& rv) {
  WrBoxRow br;
  br.ss_se = br.os01 = br.os23 = 0; br.xc = 0; br.vrow = 0;
  const int len = P.x1 - P.x0, span = len >= 4 ? (len & ~3) : 0;
  if (span <= 0 || !(B.w > 0.0f)) return br;
  const float w = 1.0f / B.w;
  float cur[4][1], st[4];
#pragma unroll
  for (int c = 0; c < 4; c++) { cur[c][0] = rv.o[c] * w; st[c] = (rv.s[c] * 4.0f) * w; }
  const float sl = float(span), ss = 4.0f;
  int shadow_start_len, shadow_end_len, os0, os1, os2, os3;
  {
    const float p0x = cur[2][0], p0y = cur[3][0];
    const bool negx = st[2] < 0.0f, negy = st[3] < 0.0f;
    float cd0 = (negx ? B.bounds[2] : B.bounds[0]) - p0x, cd1 = (negy ? B.bounds[3] : B.bounds[1]) - p0y;
    float cd2 = (negx ? B.bounds[0] : B.bounds[2]) - p0x, cd3 = (negy ? B.bounds[1] : B.bounds[3]) - p0y;
    const float rsx = 1.0f / st[2], rsy = 1.0f / st[3];
    cd0 = st[2] != 0.0f ? cd0 * rsx : 1.0e6f * wr_step01(0.0f, cd0);
    cd1 = st[3] != 0.0f ? cd1 * rsy : 1.0e6f * wr_step01(0.0f, cd1);
    cd2 = st[2] != 0.0f ? cd2 * rsx : 1.0e6f * wr_step01(0.0f, cd2);
    cd3 = st[3] != 0.0f ? cd3 * rsy : 1.0e6f * wr_step01(0.0f, cd3);
    const float shadow_start = wr_max(cd0, cd1), shadow_end = wr_min(cd2, cd3);
    shadow_start_len = int(wr_clamp(sl - ss * floorf(shadow_start), 0.0f, sl));
    shadow_end_len = int(wr_clamp(sl - ss * ceilf(shadow_end), 0.0f, sl));
    const float u0 = cur[0][0], v0 = cur[1][0];
    const bool ngx = st[0] < 0.0f, ngy = st[1] < 0.0f;
    float od0 = (ngx ? B.edge[2] : B.edge[0]) - u0, od1 = (ngy ? B.edge[3] : B.edge[1]) - v0;
    float od2 = (ngx ? B.edge[0] : B.edge[2]) - u0, od3 = (ngy ? B.edge[1] : B.edge[3]) - v0;
    const float rux = 1.0f / st[0], ruy = 1.0f / st[1];
    od0 = st[0] != 0.0f ? od0 * rux : 1.0e6f * wr_step01(0.0f, od0);
    od1 = st[1] != 0.0f ? od1 * ruy : 1.0e6f * wr_step01(0.0f, od1);
    od2 = st[0] != 0.0f ? od2 * rux : 1.0e6f * wr_step01(0.0f, od2);
    od3 = st[1] != 0.0f ? od3 * ruy : 1.0e6f * wr_step01(0.0f, od3);
    const float sel = float(shadow_end_len);
    os0 = int(wr_clamp(sl - ss * floorf(od0), sel, sl)); os1 = int(wr_clamp(sl - ss * floorf(od1), sel, sl));
    os2 = int(wr_clamp(sl - ss * floorf(od2), sel, sl)); os3 = int(wr_clamp(sl - ss * floorf(od3), sel, sl));
  }
  br.ss_se = shadow_start_len | (shadow_end_len << 16); br.os01 = os0 | (os1 << 16); br.os23 = os2 | (os3 << 16);
  // the walk of wr_box_shadow_row4, control flow only (it is all integer): find the run with u clamped
  {
    int R = span, pos = 0;
    if (R > shadow_start_len) { const int nb = R - shadow_start_len; R -= nb; pos += nb; }
    for (int guard = 0; guard < 16 && R > 0; guard++) {
      R -= 4; pos += 4;
      if (R <= shadow_end_len) break;
      int num_inside = R - 4 - shadow_end_len;
      bool xcl = false;
      if (R >= os1) num_inside = wr_imin(num_inside, R - os1);
      else if (R >= os3) num_inside = wr_imin(num_inside, R - os3);
      if (R >= os0) num_inside = wr_imin(num_inside, R - os0);
      else if (R >= os2) { num_inside = wr_imin(num_inside, R - os2); xcl = true; }
      if (num_inside > 0) {
        if (xcl && num_inside >= 8 && br.xc == 0) br.xc = pos | ((pos + num_inside) << 16);
        R -= num_inside; pos += num_inside;
      }
    }
  }
  return br;
}

// Row key of a cs_clip_box_shadow prim on an axis-aligned transform (v and local y constant along the row: s[1] == s[3] == 0).
// Everything wr_box_shadow_row_lanes computes from the row's v / local y goes through (a) the span lengths of
// wr_box_row_setup, (b) wr_box_map_uv's v -- constant between the nine-patch's edges -- of o[1] * (1 / w) (span chunks) and of
// o[1] / w (tail pixels), (c) the in-bounds factor of wr_box_shade for o[3] * (1 / w) and o[3] / w.  Two rows with equal keys
// and equal x interpolants produce the same bytes.
WR_DEVICE float wr_box_map_v(const WrBoxRec& B, float vl) {
  float v = wr_clamp(vl, 0.0f, B.edge[1]);
  v += wr_max(0.0f, vl - B.edge[3]);
  return (B.uv_noclamp[3] - B.uv_noclamp[1]) * v + B.uv_noclamp[1];
}
WR_DEVICE WrBoxKey wr_box_row_key(const WrPrim& P, const WrBoxRec& B, const WrRowVals& r, const WrBoxRow& b) {
  WrBoxKey k;
  k.valid = (r.s[1] == 0.0f && r.s[3] == 0.0f && B.w > 0.0f) ? 1 : 0;
  k.o0 = r.o[0]; k.s0 = r.s[0]; k.o2 = r.o[2]; k.s2 = r.s[2];
  k.ss_se = b.ss_se; k.os01 = b.os01; k.os23 = b.os23;
  const float w = 1.0f / B.w;
  k.mv_mul = wr_box_map_v(B, r.o[1] * w); k.mv_div = wr_box_map_v(B, r.o[1] / B.w);
  const float l = r.o[3] * w, ld = r.o[3] / B.w;
  k.in_mul = wr_step01(B.bounds[1], l) - wr_step01(B.bounds[3], l);
  k.in_div = wr_step01(B.bounds[1], ld) - wr_step01(B.bounds[3], ld);
  return k;
}
WR_DEVICE bool wr_box_keys_equal(const WrBoxKey& a, const WrBoxKey& b) {
  return a.valid && b.valid && a.o0 == b.o0 && a.s0 == b.s0 && a.o2 == b.o2 && a.s2 == b.s2 && a.ss_se == b.ss_se && a.os01 == b.os01 &&
         a.os23 == b.os23 && a.mv_mul == b.mv_mul && a.mv_div == b.mv_div && a.in_mul == b.in_mul && a.in_div == b.in_div;
}

// Rows of the flush's mask-row store for the mask prims of one wave of the setup stage (T == nullptr: this lane has none).  The store's
// allocation word packs slots << 48 | rows << 28 | bytes / 16 and is advanced by compare-and-swap so that a request that does not fit
// is refused whole.  Lane by lane that is one round trip to the word per PRIM, one after the other -- every lane of every wave of the
// flush contends for the one address, and the lanes of a wave retry in lockstep: 44 corner masks of a rounded clip cost the setup
// stage ~50 us (wrench clip-clear, large-clip-rect).  Here the wave adds its requests up (prefix sums over the requesting lanes, in
// lane order: slots stay sorted by first row), the first requesting lane swaps ONCE for all of them, and every lane takes its share;
// only when the wave's total does not fit do the lanes ask one by one for what still does.
WR_DEVICE void wr_mask_rows_take(const WrTargetDesc& T, unsigned long long ns, unsigned long long nr, unsigned long long nb, uint32_t pitch,
                                 uint32_t tabs16, int gid, int target, WrRec* __restrict__ recs, int& slot_out) {
  WrMaskSlot sl;
  sl.prim = gid; sl.target = target; sl.row0 = uint32_t(nr); sl.pitch = pitch; sl.off16 = uint32_t(nb);
  sl.pad[0] = 1u; sl.pad[1] = tabs16; sl.pad[2] = 0;  // (pad[0]: waves sharing a row -- measured on cfg4's 1840-pixel rows: 2 or 4 lose, 99 -> 138 us)
  sl.key.valid = 0;                                   // (the caller fills the key in once the prim's row-sum tables exist)
  T.mr_slots[ns] = sl;
  slot_out = int(ns);
  const unsigned long long addr = (unsigned long long)(T.mr_store + nb * 16);
  recs[gid].kbf = (recs[gid].kbf & ~0xFFu) | WR_PK_MASK_ROWS;
  recs[gid].c0 = uint32_t(addr); recs[gid].c1 = uint32_t(addr >> 32); recs[gid].z = pitch;
}
// one request, swapped in on its own; false: it does not fit
WR_DEVICE bool wr_mask_rows_cas(const WrTargetDesc& T, unsigned long long slots, unsigned long long rows, unsigned long long n16, unsigned long long& base) {
  unsigned long long old = *(volatile unsigned long long*)T.mr_ctl;
  for (;;) {
    const unsigned long long ns = old >> 48, nr = (old >> 28) & 0xFFFFFull, nb = old & 0xFFFFFFFull;
    if (ns + slots > T.mr_max_slots || nr + rows > WR_MR_MAX_ROWS || nb + n16 > T.mr_cap16) return false;
    const unsigned long long want = ((ns + slots) << 48) | ((nr + rows) << 28) | (nb + n16);
    const unsigned long long seen = atomicCAS(T.mr_ctl, old, want);
    if (seen == old) { base = old; return true; }
    old = seen;
  }
}
WR_DEVICE void wr_mask_rows_reserve(const WrTargetDesc* T, uint32_t rows, uint32_t pitch, unsigned long long n16, uint32_t tabs16,
                                    int gid, int target, WrRec* __restrict__ recs, int& slot_out) {
  slot_out = -1;
#ifndef WRHIP_HOSTSIM
  const int lane = threadIdx.x & 63;
  const unsigned long long m = __ballot(T != nullptr);
  if (!m) return;
  // (the control block is the flush's, one for all its R8 targets: any requesting lane's descriptor names it)
  uint32_t pre_slots = 0, pre_rows = 0, tot_slots = 0, tot_rows = 0;
  unsigned long long pre_n16 = 0, tot_n16 = 0;
  for (unsigned long long mm = m; mm; mm &= mm - 1ull) {
    const int i = __builtin_ctzll(mm);
    const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)rows, i);
    const unsigned long long ni = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)n16, i) |
                                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(n16 >> 32), i) << 32);
    if (lane == i) { pre_slots = tot_slots; pre_rows = tot_rows; pre_n16 = tot_n16; }
    tot_slots += 1u; tot_rows += ri; tot_n16 += ni;
  }
  const int lead = __builtin_ctzll(m);
  unsigned long long base = 0ull;
  int ok = 0;
  if (lane == lead) ok = wr_mask_rows_cas(*T, tot_slots, tot_rows, tot_n16, base) ? 1 : 0;
  ok = __builtin_amdgcn_readlane(ok, lead);
  if (ok) {
    const unsigned long long b = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, lead) |
                                 ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), lead) << 32);
    if (T) wr_mask_rows_take(*T, (b >> 48) + pre_slots, ((b >> 28) & 0xFFFFFull) + pre_rows, (b & 0xFFFFFFFull) + pre_n16, pitch, tabs16, gid, target, recs, slot_out);
    return;
  }
#endif
  // (the host simulation's one thread at a time; a wave whose requests do not fit together)
  if (T) {
    unsigned long long base;
    if (wr_mask_rows_cas(*T, 1, rows, n16, base)) wr_mask_rows_take(*T, base >> 48, (base >> 28) & 0xFFFFFull, base & 0xFFFFFFFull, pitch, tabs16, gid, target, recs, slot_out);
  }
}

// Vertex stage + binning, one thread per instance.
WR_DEVICE void wr_setup_body(const WrDrawDesc* __restrict__ draws, int n_draws,
                                const uint8_t* __restrict__ arena, WrPrim* __restrict__ prims,
                                WrRec* __restrict__ recs, WrAux* __restrict__ aux, int n_prims,
                                const WrTargetDesc* __restrict__ targets, unsigned long long* __restrict__ masks,
                                float* __restrict__ vtab, WrUnsupportedCounters* cnt, const int* __restrict__ blk, const int bid) {
  const int gid = bid * (int)blockDim.x + (int)threadIdx.x;
  const bool valid = gid < n_prims;
  WrPrim P;
  P.kind = WR_PK_NONE; P.draw = 0; P.x0 = P.y0 = P.x1 = P.y1 = 0;
#ifdef WRHIP_TIMING
  const int dbg_mode = n_draws >> 24;          // experiment switch: 1 empty kernel, 2 vertex stage only, 3 no binning
  n_draws &= 0xFFFFFF;
  if (dbg_mode == 1) return;
  const unsigned long long tm0 = wall_clock64();
#endif
  WR_TP(0);
  // the block's descriptors: WrBlock (one 16-byte trip), then its draws and their targets into LDS together (WrDescView)
  WrDescView V;
  V.draws = draws; V.targets = targets; V.ld = nullptr; V.lt = nullptr; V.lo0 = 0; V.nk = 0; V.t0 = V.t1 = -1;
  {
    const int nblk = (n_prims + 63) >> 6;
    const int b = wr_imin(gid >> 6, nblk - 1);      // (a wave past the last block has no valid lane)
#if defined(WRHIP_HOSTSIM) || !defined(WR_DESC_STAGE)
    V.lo0 = ((const WrBlock*)blk)[b].lo0;
#else
    __shared__ __attribute__((aligned(16))) unsigned long long wr_sd_lds[4 * WR_SD_WAVE_Q];
    const wr_u4 bw = wr_load16((const void*)((const WrBlock*)blk + b));
    V.lo0 = __builtin_amdgcn_readfirstlane((int)bw.x);
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    if (blockDim.x <= 256u) {
      V.nk = __builtin_amdgcn_readfirstlane((int)bw.y); V.t0 = __builtin_amdgcn_readfirstlane((int)bw.z); V.t1 = __builtin_amdgcn_readfirstlane((int)bw.w);
      unsigned long long* L = wr_sd_lds + wave * WR_SD_WAVE_Q;
      const unsigned long long* gd = (const unsigned long long*)(draws + V.lo0);
      const unsigned long long* g0 = (const unsigned long long*)(targets + wr_imax(V.t0, 0));
      const unsigned long long* g1 = (const unsigned long long*)(targets + wr_imax(V.t1, 0));
      for (int i = lane; i < V.nk * WR_SD_DRAW_Q; i += 64) L[i] = gd[i];
      if (V.nk > 0) for (int i = lane; i < WR_SD_TGT_Q; i += 64) L[2 * WR_SD_DRAW_Q + i] = g0[i];
      if (V.nk > 1) for (int i = lane; i < WR_SD_TGT_Q; i += 64) L[2 * WR_SD_DRAW_Q + WR_SD_TGT_Q + i] = g1[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      V.ld = (const WrDrawDesc*)L; V.lt = (const WrTargetDesc*)(L + 2 * WR_SD_DRAW_Q);
    }
#endif
  }
  WR_TP(1);
  if (valid) wr_vertex_prim(V, n_draws, arena, gid, P, aux, cnt);
  WR_TP(7);
#ifdef WRHIP_TIMING
  const unsigned long long tm1 = wall_clock64();
  if (dbg_mode == 2) { if (valid && P.x0 == 12345678) prims[gid] = P; return; }
#endif
  const WrTargetDesc* mr_T = nullptr;        // a mask prim's request for rows of the flush's mask-row store (below)
  uint32_t mr_rows = 0, mr_pitch = 0;
  unsigned long long mr_n16 = 0;
  uint32_t mr_tabs16 = 0;
  if (valid) {
    // The 32-byte record is all the bin raster reads of a prim that folded into `new = hi_bytes(dst * K + C)` (or was culled): its
    // 128-byte WrPrim stays unwritten -- 80 % of what the setup stage stored for a frame of plain rects (cfg5: 41 -> 9 MB).  The row
    // kernels walk prims[] of their targets directly, so those targets keep every WrPrim.
    const WrDrawDesc& D_ = V.draw(P.draw);
    const WrTargetDesc& T_ = V.target(D_.target);
    const WrRec rec_ = wr_make_rec(P, T_.format);
    recs[gid] = rec_;
    if (T_.rows_mode || ((rec_.kbf & 0xFF) != WR_PK_SOLID_FOLDED && (rec_.kbf & 0xFF) != WR_PK_NONE)) prims[gid] = P;
    if ((P.kind == WR_PK_BOX_SHADOW || P.kind == WR_PK_CLIP_RECT) && (D_.flags & WR_DF_MASK_ROWS) && P.x1 > P.x0 && P.y1 > P.y0) {
      // this prim wants its rows in the flush's mask-row store (WrMaskSlot): the request here, the reservation after this block, once per
      // WAVE (wr_mask_rows_reserve); a prim that does not fit keeps its in-raster evaluation
      const WrTargetDesc& T = T_;
      if (T.mr_ctl) {
        mr_T = &T;
        mr_rows = uint32_t(P.y1 - P.y0); mr_pitch = uint32_t(((P.x1 + 3) & ~3) - (P.x0 & ~3));
        // [row map: rows x u32, padded to 16 B][rows x pitch bytes]
        // [row map: rows x u32, padded to 16 B][rows x pitch bytes][row-sum tables: WrAccTabs]
        const unsigned long long map16 = ((unsigned long long)mr_rows * 4 + 15) >> 4;
        mr_n16 = map16 + (((unsigned long long)mr_rows * mr_pitch + 15) >> 4);
        mr_tabs16 = uint32_t(mr_n16);
        mr_n16 += WR_ACCTABS_N16;
      }
    }
    // gradient tables' can_merge bitmap (WrGradRec::merge): here, ONCE, after the vertex stage's registers are dead (at its four
    // call sites inside the vertex stages it cost every kernel that carries the setup stage 2 KB of scratch per lane)
    // (gradient prims: table copy and merge bitmap after this block, where the wave is whole again)
    if (P.kind == WR_PK_TEX_R8) {
      aux[gid].tex = wr_make_texrec(P, D_.tex[P.tex_slot]);
      wr_write_glyph_rec(T_, gid, P, aux[gid].tex);
    }
    if (P.kind == WR_PK_SOLID_MASKED) {
      // a masked solid is a unit-texel read of the mask: reuse the glyph path's record
      const WrTexDesc& mt = D_.tex[WR_S_CLIP_MASK];
      WrTexRec t;
      __builtin_memset(&t, 0, sizeof(t));
      t.ptr = mt.ptr; t.stride = mt.stride; t.wh = uint32_t(mt.width) | (uint32_t(mt.height) << 16);
      t.span = 0x40000000; t.y0 = P.y0;
      t.simple = ((P.color[0] | P.color[1]) & 0xFF00FF00u) == 0 ? 1 : 0;
      t.unit = 1; t.ix0 = P.x0 - P.mask_off[0]; t.iy0 = P.y0 - P.mask_off[1];
      aux[gid].tex = t;
      wr_write_glyph_rec(T_, gid, P, t);
    }
    if (P.kind == WR_PK_TEX_RGBA8) {
      const WrDrawDesc& d = D_;
      WrTexRec T;
      const int rows = P.y1 - P.y0;
      if (wr_texrow_x_setup(P, d.tex[P.tex_slot], T, cnt) && (T.simple == 3 || (d.vtab_base >= 0 && rows <= d.vtab_rows))) {
        if (T.simple == 2) {
          // v of every target row, accumulated exactly like Edge::nextRow (a few hundred dependent adds)
          T.iy0 = d.vtab_base + (gid - d.first_prim) * d.vtab_rows;
          float* vp = vtab + T.iy0;
          const float st = T.lvs;
          float v = T.lv0;
          int k = 0;
          for (; k + 4 <= rows; k += 4) {
            const float a = v, b = a + st, c = b + st, e = c + st;
            v = e + st;
            vp[k] = a; vp[k + 1] = b; vp[k + 2] = c; vp[k + 3] = e;
          }
          for (; k < rows; k++, v += st) vp[k] = v;
        }
        aux[gid].tex = T;
      } else {
        aux[gid].tex.simple = 0;
      }
    }
  }
  int mr_slot = -1;
  wr_mask_rows_reserve(mr_T, mr_rows, mr_pitch, mr_n16, mr_tabs16, gid, V.draw(P.draw).target, recs, mr_slot);
  if (mr_slot >= 0) {
    // The prim has its rows: its row-sum tables (one walk over the prim's rows per interpolant that needs one, here, instead of one
    // per ROW in the rows kernel), then -- for a box shadow -- the key of its middle row, which the rows kernel compares every row's
    // key with.  (After the reservation: the wave's lanes get here together.)
    WrMaskSlot& sl = mr_T->mr_slots[mr_slot];
    WrAccTabs* tabs = (WrAccTabs*)(mr_T->mr_store + ((size_t)sl.off16 + mr_tabs16) * 16);
    const bool box = P.kind == WR_PK_BOX_SHADOW;
    const WrBoxRec& B = aux[gid].box;
    const float s0[8] = {P.uvL0[0], P.uvL0[1], P.uvR0[0], P.uvR0[1], box ? B.lpL0[0] : 0.0f, box ? B.lpL0[1] : 0.0f, box ? B.lpR0[0] : 0.0f, box ? B.lpR0[1] : 0.0f};
    const float st[8] = {P.uvLs[0], P.uvLs[1], P.uvRs[0], P.uvRs[1], box ? B.lpLs[0] : 0.0f, box ? B.lpLs[1] : 0.0f, box ? B.lpRs[0] : 0.0f, box ? B.lpRs[1] : 0.0f};
    wr_acctabs_build(tabs, box ? 8 : 4, s0, st, int(mr_rows) - 1, P.rows_linear != 0);
    if (box) {
      const int yc = P.y0 + (int(mr_rows) >> 1);
      const WrRowVals rvc = wr_box_row_vals(P, B, yc, tabs);
      sl.key = wr_box_row_key(P, B, rvc, wr_box_row_setup(P, B, rvc));
    }
  }
  {
    // gradient tables: copied into the flush's pool (wave-wide: wr_grad_tables_wave), then the can_merge bitmap (WrGradRec::merge) -- here, ONCE,
    // after the vertex stage's registers are dead (at its four call sites inside the vertex stages it cost every kernel that carries
    // the setup stage 2 KB of scratch per lane)
    WrGradRec* Gq = nullptr;
    if (valid && P.kind == WR_PK_GRADIENT) Gq = &aux[gid].grad;
    else if (valid && P.kind == WR_PK_TEX_QUAD && aux[gid].quad.base_kind == WR_PK_GRADIENT) Gq = &aux[gid].quad.grad;
    wr_grad_tables_wave(V, P.draw, gid, Gq);
    if (Gq) wr_grad_merge_bits(Gq);
  }
#ifdef WRHIP_TIMING
  const unsigned long long tm2 = wall_clock64();
  if (valid && gid < 16384) { wr_dbg_prim[gid * 4] = (unsigned)(tm1 - tm0); wr_dbg_prim[gid * 4 + 1] = (unsigned)(tm2 - tm1); wr_dbg_prim[gid * 4 + 2] = (unsigned)(wr_dbg_mid[gid] - tm0); wr_dbg_prim[gid * 4 + 3] = (unsigned)draws[P.draw].shader; }
#endif
#ifdef WRHIP_TIMING
  if (dbg_mode == 3) return;
#endif
  wr_bin_prim(P, valid, gid, V, masks);
#ifdef WRHIP_TIMING
  const unsigned long long tm3 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {   // per-wave phase times, summed (host prints per-Finish deltas)
    atomicAdd(&cnt->dbg[0], (unsigned)(tm1 - tm0)); atomicAdd(&cnt->dbg[1], (unsigned)(tm2 - tm1)); atomicAdd(&cnt->dbg[2], (unsigned)(tm3 - tm2));
    atomicAdd(&cnt->dbg[3], 1u);
    if (bid == (n_prims - 1) / (int)blockDim.x) atomicAdd(&cnt->dbg[4], (unsigned)(tm3 - tm0));   // waves of the last workgroup (composite prims)
    atomicMax(&cnt->dbg[5], (unsigned)(tm3 - tm0));
  }
#endif
}

#ifndef WR_INST_ONLY      /* (not a template: defined by wrhip.hip alone, not by the instantiation units wrhip_inst.hip) */
// (three waves per SIMD asked: left alone the allocator takes 169 VGPRs -- one past the step -- since the clip stash carries a second varying)
__global__ void __launch_bounds__(256, 3) wr_setup_kernel(const WrDrawDesc* __restrict__ draws, int n_draws,
                                const uint8_t* __restrict__ arena, WrPrim* __restrict__ prims,
                                WrRec* __restrict__ recs, WrAux* __restrict__ aux, int n_prims,
                                const WrTargetDesc* __restrict__ targets, unsigned long long* __restrict__ masks,
                                float* __restrict__ vtab, WrUnsupportedCounters* cnt, const int* __restrict__ blk) {
  wr_setup_body(draws, n_draws, arena, prims, recs, aux, n_prims, targets, masks, vtab, cnt, blk, (int)blockIdx.x);
}
#endif
