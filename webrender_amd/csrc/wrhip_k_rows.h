// wrhip_k_rows.h -- part of the gfx950 kernels of libwrhip: included by wrhip_kernels.h, in its order, and by nothing else.
// The row kernels' bodies: mask rows (cs_clip_* prims evaluated once per row), span rows (cs_blur / cs_scale targets), tile rows (picture targets of a few large prims).
#pragma once

// ---------------------------------------------------------------------------
// Mask rows (WrMaskSlot): one wave evaluates one target row of one cs_clip_* prim into the flush's mask-row store.
// dst[n] is the byte of pixel x0 + n.  Every lane walks the row's state machine (the walk is the same for all of them: no
// divergence) and the pixels of each run are dealt out to the lanes; no two lanes ever store to the same byte.
WR_DEVICE void wr_fill_lanes(uint8_t* dst, int a, int b, uint32_t v, int lane, int stride = 64) {      // lane: + 64 * part
  for (int n = a + lane; n < b; n += stride) dst[n] = (uint8_t)v;
}
// cs_clip_box_shadow (cs_clip_box_shadow.glsl:150-324): the walk of wr_box_shadow_row4 over the whole row
WR_DEVICE void wr_box_shadow_row_lanes(const WrPrim& P, const WrBoxRec& B, const WrRowVals& rv, const WrBoxRow& br, int lane, uint8_t* dst,
                                       int part = 0, int parts = 1) {
  const int wl = lane + 64 * part, ws = 64 * parts;      // this wave's share of a run: pixels wl, wl + ws, ..
  const WrTexDesc t{B.ptr, int(B.wh & 0xFFFF), int(B.wh >> 16), B.stride, (int16_t)B.format, (int16_t)B.linear, float(B.wh & 0xFFFF), float(B.wh >> 16)};
  float o4[4], s4[4];
#pragma unroll
  for (int c = 0; c < 4; c++) { o4[c] = rv.o[c]; s4[c] = rv.s[c]; }
  const int len = P.x1 - P.x0, span = len >= 4 ? (len & ~3) : 0;
  const float mode = B.mode;
  const uint32_t v_clear = uint32_t(wr_round_pixel(mode)) & 0xFFFF;
  float ln[4][4];
#pragma unroll
  for (int c = 0; c < 4; c++) { ln[c][0] = o4[c]; ln[c][1] = ln[c][0] + s4[c]; ln[c][2] = ln[c][1] + s4[c]; ln[c][3] = ln[c][2] + s4[c]; }
  // ---- tail pixels [span, len): fragment shader
  if (part == 0 && lane < len - span) {
    const int n = span + lane;
    const int sl4 = (n - span) & 3, m = (n - span) >> 2;
    float v4[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float a = wr_sel4(ln[c][0], ln[c][1], ln[c][2], ln[c][3], sl4);
      if (span > 0) a = a + (s4[c] * 4.0f) * (float(span) * 0.25f);
      v4[c] = wr_accum(a, (s4[c] * 4.0f) * 1.0f, m);
    }
    const float r = wr_box_shade(B, t, v4[0] / B.w, v4[1] / B.w, v4[2] / B.w, v4[3] / B.w);
    dst[n] = (uint8_t)(uint32_t(wr_round_pixel(B.w > 0.0f ? r : 0.0f)) & 0xFFFF);
  }
  if (span <= 0) return;
  float w = B.w;
  if (w <= 0.0f) { wr_fill_lanes(dst, 0, span, 0, wl, ws); return; }      // swgl_commitSolidR8(0.0)
  w = 1.0f / w;
  float cur[4][4], st[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int i = 0; i < 4; i++) cur[c][i] = ln[c][i] * w;
    st[c] = (s4[c] * 4.0f) * w;
  }
  const int shadow_start_len = br.ss_se & 0xFFFF, shadow_end_len = br.ss_se >> 16;
  const int os0 = br.os01 & 0xFFFF, os1 = br.os01 >> 16, os2 = br.os23 & 0xFFFF, os3 = br.os23 >> 16;
  int R = span, pos = 0;
  if (R > shadow_start_len) {                       // solid lead-in
    const int nb = R - shadow_start_len;
    const float f = float(nb / 4);
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int i = 0; i < 4; i++) cur[c][i] += f * st[c];
    }
    wr_fill_lanes(dst, 0, nb, v_clear, wl, ws);
    R -= nb; pos += nb;
  }
  // The transitional chunks (per-fragment mapping, a texture fetch each) of the walk are collected -- chunk number k goes to the four
  // lanes 4 (k mod 16) .., which keep their pixel's interpolants -- and shaded together, sixteen chunks at a time: one after the other
  // on four lanes they were a dependent fetch per chunk.
  int tchunk = 0, tn = -1;                          // transitional chunks met so far; tn: the pixel this lane holds (-1: none)
  float tv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto shade_held = [&]() {
    if (tn >= 0) dst[tn] = (uint8_t)(uint32_t(wr_round_pixel(wr_box_shade(B, t, tv[0], tv[1], tv[2], tv[3]))) & 0xFFFF);
    tn = -1;
  };
  while (R > 0) {
    if (tchunk > 0 && (tchunk & 15) == 0) shade_held();
    if (part == (tchunk >> 4) % parts && (lane >> 2) == (tchunk & 15)) {
      const int n = pos + (lane & 3);
      if (n < span) {
        const int l4 = n & 3;
        tn = n;
#pragma unroll
        for (int c = 0; c < 4; c++) tv[c] = wr_sel4(cur[c][0], cur[c][1], cur[c][2], cur[c][3], l4);
      }
    }
    tchunk++;
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int i = 0; i < 4; i++) cur[c][i] += st[c];
    }
    R -= 4; pos += 4;
    if (R <= shadow_end_len) break;
    int num_inside = R - 4 - shadow_end_len;
    float ub0 = B.uv_bounds[0], ub1 = B.uv_bounds[1], ub2 = B.uv_bounds[2], ub3 = B.uv_bounds[3];
    bool xcl = false;
    if (R >= os1) {
      num_inside = wr_imin(num_inside, R - os1);
    } else if (R >= os3) {
      num_inside = wr_imin(num_inside, R - os3);
      const float cc = wr_clamp((B.uv_noclamp[3] - B.uv_noclamp[1]) * B.edge[1] + B.uv_noclamp[1], B.uv_bounds[1], B.uv_bounds[3]);
      ub1 = cc; ub3 = cc;
    }
    if (R >= os0) {
      num_inside = wr_imin(num_inside, R - os0);
    } else if (R >= os2) {
      num_inside = wr_imin(num_inside, R - os2);
      const float cc = wr_clamp((B.uv_noclamp[2] - B.uv_noclamp[0]) * B.edge[0] + B.uv_noclamp[0], B.uv_bounds[0], B.uv_bounds[2]);
      ub0 = cc; ub2 = cc;
      xcl = true;
    }
    if (num_inside > 0) {
      float pu[4], pv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) wr_box_map_uv(B, cur[0][i], cur[1][i], pu[i], pv[i]);
      const bool centre = ub0 == ub2 && ub1 == ub3;
      const float W = float(t.width), H = float(t.height);
      int filter = 0;
      if (!centre) {   // needsTextureLinear (swgl_ext.h:553-587)
        if (t.width < 2) filter = 0;
        else if (pv[0] != pv[1]) filter = 1;
        else {
          const float px0 = pu[0] * W, px1 = pu[1] * W, py0 = pv[0] * H;
          const int sp = (num_inside & ~127) + 128;
          const int scaled = int(roundf((px1 - px0) * float(sp)));
          if (scaled != sp) filter = (px0 < px1 && px1 - px0 <= 1.0f) ? 2 : (scaled == sp * 2 ? 4 : 1);
          else if ((int(px0 * 4.0f + 0.5f) & 3) != 2 || (int(py0 * 4.0f + 0.5f) & 3) != 2) filter = 3;
          else filter = 0;
        }
      }
      // a run with u clamped to the nine-patch's stretched middle column and v not moving along the row samples one texel
      // position: every pixel of it gets what its first pixel gets (WrBoxRow::xc)
      const bool one_value = !centre && xcl && num_inside >= 8 && rv.s[1] == 0.0f;
      const int end = wr_imin(pos + num_inside, span);
      // the run, sampled like swgl_commitTextureLinear(R8, sColor0, uv, uv_bounds, NoColor/InvertColor, num_inside)
      const float qs = 128.0f, qo = 0.5f - 0.5f * qs;
      float q[4], qy[4];
#pragma unroll
      for (int a = 0; a < 4; a++) { q[a] = pu[a] * W * qs + qo; qy[a] = pv[a] * H * qs + qo; }
      const float stepx = 4.0f * (q[1] - q[0]), stepy = 4.0f * (qy[1] - qy[0]);
      const float minx = wr_max(ub0 * W * qs + qo, 0.0f), miny = wr_max(ub1 * H * qs + qo, 0.0f);
      const float maxx = wr_max(ub2 * W * qs + qo, minx), maxy = wr_max(ub3 * H * qs + qo, miny);
      // pixel j of the run (the one value of a u-clamped run; nearest runs)
      auto run_pixel = [&](int j) -> uint32_t {
        int v;
        if (filter == 0) {
          // blendTextureNearestFast (swgl_ext.h:475-537)
          const int ix = int(pu[0] * W), iy = int(pv[0] * H);
          const int minUx = int(ub0 * W), minUy = int(ub1 * H), maxUx = int(ub2 * W), maxUy = int(ub3 * H);
          const int srow = wr_clamp_coord(wr_iclamp(iy, minUy, maxUy), t.height);
          const int minX = wr_iclamp(minUx, 0, t.width - 1), maxX = wr_iclamp(maxUx, minX, t.width - 1);
          v = ((const uint8_t*)t.ptr)[(size_t)srow * t.stride + wr_iclamp(ix + j, minX, maxX)];
        } else {
          int o[4];
          wr_linear_span_pixel<1>(t, q, qy, stepx, stepy, minx, maxx, miny, maxy, filter, num_inside, j, o);
          v = o[0];
        }
        if (mode != 0.0f) v = 255 - v;               // applyColor(src, InvertColor)
        return uint32_t(v) & 0xFFFF;
      };
      if (centre) {
        // centre sector: one texel for the whole run (pattern of the 4 lanes repeated; a lane's pixels are 64 apart)
        const int l4 = (pos + lane) & 3;
        const float texel = wr_r8_texture(t, wr_clamp(wr_sel4(pu[0], pu[1], pu[2], pu[3], l4), ub0, ub2),
                                          wr_clamp(wr_sel4(pv[0], pv[1], pv[2], pv[3], l4), ub1, ub3));
        wr_fill_lanes(dst, pos, end, uint32_t(wr_round_pixel(((1.0f - texel) - texel) * mode + texel)) & 0xFFFF, lane + 64 * part, ws);
      } else if (one_value) {
        wr_fill_lanes(dst, pos, end, run_pixel(0), wl, ws);
      } else if (filter == 0) {
        for (int n = pos + wl; n < end; n += ws) dst[n] = (uint8_t)run_pixel(n - pos);
      } else {
        // a chunk per lane, the chains of `uv += uv_step` stepped by the wave (wr_linear_span_lanes_r8)
        uint8_t* const rdst = dst + pos;
        const int rlen = end - pos;
        wr_linear_span_lanes_r8(t, q, qy, stepx, stepy, minx, maxx, miny, maxy, filter, num_inside, wl, ws, [&](int n, const int (&v)[4], int cnt) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (k >= cnt || n + k >= rlen) break;
            rdst[n + k] = (uint8_t)(mode != 0.0f ? 255 - v[k] : v[k]);
          }
        });
      }
      const float f = float(num_inside / 4);
#pragma unroll
      for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int i = 0; i < 4; i++) cur[c][i] += f * st[c];
      }
      R -= num_inside; pos += num_inside;
    }
  }
  shade_held();
  wr_fill_lanes(dst, pos, span, v_clear, wl, ws);    // solid lead-out
}
// cs_clip_rectangle (cs_clip_rectangle.glsl:223-420): the row's five phases are closed forms of the chunk index, so the lanes
// take a chunk each; a chunk in a solid phase is a constant
WR_DEVICE void wr_clip_rect_row_lanes(const WrPrim* Pp, const WrClipRec* Cp, int y, int lane, uint8_t* dst, int part = 0, int parts = 1, const WrAccTabs* tabs = nullptr) {
  const int wl = lane + 64 * part, ws = 64 * parts;
  const WrPrim& P = *Pp;
#ifdef WRHIP_HOSTSIM
  const WrRowVals rv = wr_clip_row_vals(P, y, tabs);
#else
  const WrRowVals rv = wr_clip_row_vals_wave(P, y, lane, tabs);      // (y is wave-uniform here)
#endif
  const WrClipRow cr = wr_clip_row_setup(P, *Cp, rv);
  const int len = P.x1 - P.x0, span = len >= 4 ? (len & ~3) : 0, S = span >> 2;
  const int n2 = cr.n12 >> 16, n4 = cr.n34 >> 16;
  const int b1 = cr.n12 & 0xFFFF, b2 = b1 + n2, b3 = b2 + (cr.n34 & 0xFFFF), b4 = b3 + n4;
  const float mode = Cp->mode;
  if (!(Cp->w > 0.0f)) {                        // (degenerate w: every pixel through the general function)
    for (int n = wl; n < len; n += ws) dst[n] = (uint8_t)wr_clip_rect_px(P, *Cp, rv, cr, n);
    return;
  }
  // the solid phases [0, b1) clear, [b2, b3) opaque, [b4, S) clear: constants, a chunk per lane
  const uint32_t v_clear = uint32_t(wr_round_pixel(mode)) & 0xFF, v_opaque = uint32_t(wr_round_pixel(1.0f - mode)) & 0xFF;
  for (int c = wl; c < S; c += ws) {
    const int k = c < b1 ? 0 : (c < b2 ? 1 : (c < b3 ? 2 : (c < b4 ? 3 : 0)));
    if (k == 1 || k == 3) continue;
    const uint32_t v = k == 0 ? v_clear : v_opaque;
#pragma unroll
    for (int i = 0; i < 4; i++) dst[4 * c + i] = (uint8_t)v;
  }
  // the two AA phases and the tail, a PIXEL per lane
  const int na = 4 * n2, nb = 4 * n4, nt = len - span;
  for (int p = wl; p < na + nb + nt; p += ws) {
    const int n = p < na ? 4 * b1 + p : (p < na + nb ? 4 * b3 + (p - na) : span + (p - na - nb));
    dst[n] = (uint8_t)wr_clip_rect_px(P, *Cp, rv, cr, n);
  }
}
#if defined(WR_ROWS_TIMING) && !defined(WRHIP_HOSTSIM)
__device__ unsigned long long wr_rows_times[4096 * 8];      // debug build: phase timestamps of the first item of the launch's first 4096 waves
// (-DWR_ROWS_TIMING=1: rows of box-shadow prims only, =2: of clip-rectangle prims only -- the launches of a frame overwrite one another's records)
#define WR_RT(i) do { wr_rt[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WR_RT(i) ((void)0)
#endif
WR_DEVICE void wr_mask_rows_body(const WrTargetDesc* __restrict__ targets, int bin_lo, int bin_hi,
                                 const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux,
                                 unsigned long long* __restrict__ ctl,
                                 const WrMaskSlot* __restrict__ slots, uint8_t* __restrict__ store, const int block, const int nblocks) {
  const unsigned long long a = *ctl;
  const int ns = int(a >> 48), rows_total = int((a >> 28) & 0xFFFFFull);
  const int lane = threadIdx.x & 63;
  const int nwaves = int((nblocks * blockDim.x) >> 6);
#ifdef WRHIP_HOSTSIM
  const int gw = int((block * blockDim.x + threadIdx.x) >> 6);
#else
  // the wave index is wave-uniform: say so, and everything below (slot, prim, row) is read through the scalar cache
  const int gw = __builtin_amdgcn_readfirstlane(int((block * blockDim.x + threadIdx.x) >> 6));
  // the first 64 slots, one per lane, requested together with the allocation word: a launch of a few hundred rows is one
  // dependent-load chain per wave, every level of it a cold miss
  WrMaskSlot mine;
  mine.prim = mine.target = 0; mine.row0 = 0xFFFFFFFFu; mine.pitch = mine.off16 = 0; mine.pad[0] = 1; mine.pad[1] = 0;
  if (lane < ns) mine = slots[lane];
#endif
  for (int item = gw; item < rows_total; item += nwaves) {
    WrMaskSlot sl;
    int si;
#if defined(WR_ROWS_TIMING) && !defined(WRHIP_HOSTSIM)
    unsigned long long wr_rt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    WR_RT(0);
#ifndef WRHIP_HOSTSIM
    const unsigned long long le = __ballot(lane < ns && (int)mine.row0 <= item);
    if (ns <= 64 || !(le >> 63)) {
      const int idx = 63 - __builtin_clzll(le | 1ull);
      si = idx;
      sl.prim = __builtin_amdgcn_readlane(mine.prim, idx); sl.target = __builtin_amdgcn_readlane(mine.target, idx);
      sl.row0 = (uint32_t)__builtin_amdgcn_readlane((int)mine.row0, idx); sl.pitch = (uint32_t)__builtin_amdgcn_readlane((int)mine.pitch, idx);
      sl.off16 = (uint32_t)__builtin_amdgcn_readlane((int)mine.off16, idx);
      sl.pad[0] = (uint32_t)__builtin_amdgcn_readlane((int)mine.pad[0], idx);
      sl.pad[1] = (uint32_t)__builtin_amdgcn_readlane((int)mine.pad[1], idx);
    } else
#endif
    {
      int lo = 0, hi = ns - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)slots[mid].row0 <= item) lo = mid; else hi = mid - 1;
      }
      sl = slots[lo];
      si = lo;
    }
    WR_RT(1);
    const WrTargetDesc& T = targets[sl.target];
    if (T.first_bin < bin_lo || T.first_bin >= bin_hi) continue;
    const WrPrim* Pp = &prims[sl.prim];
    const int parts = (int)sl.pad[0], idx = item - (int)sl.row0;
    const int y = Pp->y0 + idx / parts, part = idx % parts;
    if (y < T.y_begin || y >= T.y_end) continue;      // rows of another rank
    WR_RT(2);
    const int nrows = Pp->y1 - Pp->y0;
    const uint32_t rows_at = uint32_t((nrows * 4 + 15) & ~15);                     // the pixel rows follow the row map
    uint8_t* pbase = store + (size_t)sl.off16 * 16;
    const WrAccTabs* tabs = sl.pad[1] ? (const WrAccTabs*)(pbase + (size_t)sl.pad[1] * 16) : nullptr;      // the prim's row-sum tables (setup stage)
    uint32_t my_off = rows_at + uint32_t(y - Pp->y0) * sl.pitch;
    WrRowVals brv;
    WrBoxRow bbr;
    if (Pp->kind == WR_PK_BOX_SHADOW) {
      // the rows of a nine-patch's middle band are identical: a row whose key equals the key of the prim's middle row (left
      // in the slot by the setup stage) points at that row's bytes instead of being evaluated (wr_box_row_key)
      const int yc = Pp->y0 + (nrows >> 1);
#ifdef WRHIP_HOSTSIM
      brv = wr_box_row_vals(*Pp, aux[sl.prim].box, y, tabs);
#else
      brv = wr_box_row_vals_wave(*Pp, aux[sl.prim].box, y, lane, tabs);
#endif
      WR_RT(3);
      bbr = wr_box_row_setup(*Pp, aux[sl.prim].box, brv);
      WR_RT(4);
      if (y != yc && wr_box_keys_equal(wr_box_row_key(*Pp, aux[sl.prim].box, brv, bbr), slots[si].key)) {
        if (lane == 0 && part == 0) ((uint32_t*)pbase)[y - Pp->y0] = rows_at + uint32_t(yc - Pp->y0) * sl.pitch;
        continue;
      }
    }
    if (lane == 0 && part == 0) {
      ((uint32_t*)pbase)[y - Pp->y0] = my_off;
      // bytes evaluated (profiling: the launch's algorithmic bytes), over 32 counters: one word took an atomic from every
      // evaluated row, and a few thousand read-modify-writes of one address are tens of microseconds the launch ends on
      atomicAdd(&ctl[32 + (gw & 31)], (unsigned long long)(Pp->x1 - Pp->x0));
    }
    uint8_t* dst = pbase + my_off + (Pp->x0 & 3);
    WR_RT(5);
    if (Pp->kind == WR_PK_BOX_SHADOW) wr_box_shadow_row_lanes(*Pp, aux[sl.prim].box, brv, bbr, lane, dst, part, parts);
    else wr_clip_rect_row_lanes(Pp, &aux[sl.prim].clip, y, lane, dst, part, parts, tabs);
    WR_RT(6);
#if defined(WR_ROWS_TIMING) && !defined(WRHIP_HOSTSIM)
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0 && gw < 4096 && item == gw && (WR_ROWS_TIMING + 0 == 0 || WR_ROWS_TIMING + 0 == (Pp->kind == WR_PK_BOX_SHADOW ? 1 : 2))) {
      for (int i = 0; i < 7; i++) wr_rows_times[gw * 8 + i] = wr_rt[i];
      wr_rows_times[gw * 8 + 7] = ((unsigned long long)(Pp->kind == WR_PK_BOX_SHADOW ? 1 : 0) << 56) | ((unsigned long long)(uint32_t)y << 32) | (uint32_t)(__builtin_readcyclecounter() & 0xFFFFFFFFu);
    }
#endif
  }
}
#ifndef WR_INST_ONLY      /* (not a template: defined by wrhip.hip alone, not by the instantiation units wrhip_inst.hip) */
__global__ void __launch_bounds__(256, 4) wr_mask_rows_kernel(const WrTargetDesc* __restrict__ targets, int bin_lo, int bin_hi,
                                                           const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux,
                                                           unsigned long long* __restrict__ ctl,
                                                           const WrMaskSlot* __restrict__ slots, uint8_t* __restrict__ store) {
  wr_mask_rows_body(targets, bin_lo, bin_hi, prims, aux, ctl, slots, store, (int)blockIdx.x, (int)gridDim.x);
}
#endif

// ---------------------------------------------------------------------------
// Span rows ("span rows").  The off-screen passes of a blur chain -- cs_scale halvings, cs_blur V / H, the
// scissored clears in front of them -- are a handful of axis-aligned prims per target whose span shaders are state machines per
// ROW (interpolants stepped along the row, the span / main() split, the filter decision); in the bin raster a lane replays that
// row setup for every one of its pixels, a level of a few bins is one wave's dependent instruction stream (profiles/r04_j), and a
// 51 x 51 task occupies 16 waves.  A span-rows target (WrTargetDesc::rows_mode, chosen by the host) has no bins: every 256-pixel
// piece of every target row gets ONE wave; lane l holds pixels 4 l .. 4 l + 3 of the piece in registers, starts them from the
// target's clear colour (or its content), applies the target's prims in submission order -- the row setup of a prim is wave-uniform
// and evaluated once, then each lane evaluates its own pixels with the very routines the bin raster uses (wr_blur_row_pixel,
// wr_tex_pixel_row: identical bytes by construction) and blends -- and stores once.  Rows of other ranks are skipped.
// (out of line and called from rolled loops: a thin level runs its code from a cold instruction cache -- kernel boundaries invalidate it --
// and what a wave pays for is the number of distinct instruction lines on its path: one copy of a pixel routine, fetched by the
// lane's first pixel and hot for the other three, not four inlined copies; profiles/r05_c: 49 -> us for a 51 x 51 blur level)
template <int FMT>
__device__ __noinline__ void wr_span_blur_setup(const WrPrim* Pp, const WrBlurRec* Bp, int y, WrBlurRow* out) { *out = wr_blur_row_setup<FMT>(*Pp, *Bp, y); }
template <int FMT>
__device__ __noinline__ uint32_t wr_span_blur_px(const WrPrim* Pp, const WrBlurRec* Bp, const WrBlurRow* Rp, int n) {
  const WrWide src = wr_blur_row_pixel<FMT>(*Pp, *Bp, *Rp, n);
  return FMT == WR_FMT_RGBA8 ? wr_pack(src) : wr_pack1(src.bg & 0xFFFF);
}
__device__ __noinline__ void wr_span_tex_setup(const WrPrim* Pp, const WrTexDesc* tp, int y, WrTexRow* out) { *out = wr_tex_row(*Pp, *tp, y); }
template <int FMT>
__device__ __noinline__ uint32_t wr_span_tex_px(const WrPrim* Pp, const WrTexDesc* tp, const WrTexRow* rp, int n) {
  const WrWide src = wr_tex_pixel_row(*Pp, *tp, *rp, n);
  return FMT == WR_FMT_RGBA8 ? wr_pack(src) : wr_pack1(src.ra & 0xFFFF);
}
// PPL pixels per lane: 4 for wide targets (16-byte stores), 1 for the narrow ones -- the small levels of a chain are latency-bound,
// and a lane that evaluates four pixels one after the other makes the level four pixel evaluations long
template <int FMT, int PPL>
WR_DEVICE void wr_span_row_piece(const WrTargetDesc& T, const int y, const int xbase, const WrDrawDesc* __restrict__ draws,
                                 const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux, const int piece_x0) {
  constexpr int BPP = FMT == WR_FMT_RGBA8 ? 4 : 1;
  uint32_t px[PPL];
  uint8_t* rowp = (uint8_t*)T.color + (size_t)y * T.stride;
  const int nvalid = wr_iclamp(T.width - xbase, 0, PPL);        // pixels of this lane inside the target
  if (T.load_color) {
#pragma unroll
    for (int i = 0; i < PPL; i++) px[i] = i < nvalid ? (BPP == 4 ? ((const uint32_t*)rowp)[xbase + i] : (uint32_t)rowp[xbase + i]) : 0u;
  } else {
#pragma unroll
    for (int i = 0; i < PPL; i++) px[i] = BPP == 4 ? T.init_color : (T.init_color & 0xFF);
  }
  for (int p = T.first_prim; p < T.end_prim; p++) {
    const WrPrim& P = prims[p];
    const int kind = P.kind;
    if (kind == WR_PK_NONE || kind == WR_PK_UNSUPPORTED) continue;
    if (y < P.y0 || y >= P.y1 || P.x1 <= piece_x0 || P.x0 >= piece_x0 + 64 * PPL) continue;       // (wave-uniform)
    const WrDrawDesc* D = &draws[P.draw];
    const int n0 = xbase - P.x0, len = P.x1 - P.x0;
    if (kind == WR_PK_CLEAR) {
      if (P.flags & WR_PF_CLEAR_COLOR) {
#pragma unroll
        for (int i = 0; i < PPL; i++) px[i] = (unsigned)(n0 + i) < (unsigned)len ? (BPP == 4 ? P.color[0] : (P.color[0] & 0xFF)) : px[i];
      }
    } else if (kind == WR_PK_BLUR && P.blend == WR_BLEND_NONE) {
      const WrBlurRec* Bp = &aux[p].blur;
      WrBlurRow R;
      wr_span_blur_setup<FMT>(&P, Bp, y, &R);
      // the lane's pixels through ONE copy of the routine: the values rotate through px[0]
#pragma nounroll
      for (int i = 0; i < PPL; i++) {
        uint32_t v = px[0];
        if ((unsigned)(n0 + i) < (unsigned)len) v = wr_span_blur_px<FMT>(&P, Bp, &R, n0 + i);
#pragma unroll
        for (int k = 0; k + 1 < PPL; k++) px[k] = px[k + 1];
        px[PPL - 1] = v;
      }
    } else if ((kind == WR_PK_TEX_RGBA8 || kind == WR_PK_TEX_FS) && P.blend == WR_BLEND_NONE && !(P.flags & WR_PF_MASKED) && !P.dual) {
      const WrTexDesc* tp = &D->tex[P.tex_slot];
      WrTexRow r;
      wr_span_tex_setup(&P, tp, y, &r);
#pragma nounroll
      for (int i = 0; i < PPL; i++) {
        uint32_t v = px[0];
        if ((unsigned)(n0 + i) < (unsigned)len) v = wr_span_tex_px<FMT>(&P, tp, &r, xbase + i - r.x0);
#pragma unroll
        for (int k = 0; k + 1 < PPL; k++) px[k] = px[k + 1];
        px[PPL - 1] = v;
      }
    } else {
      // the host's promise (only unblended row-evaluable prims in a span-rows target) does not hold for this prim: reported, not drawn
      if (T.counters && xbase == piece_x0 && y == wr_imax(P.y0, T.y_begin) && piece_x0 <= P.x0) atomicAdd(&T.counters->unsupported_prims, 1u);
    }
  }
  if (nvalid <= 0) return;
  if (BPP == 4) {
    uint32_t* dst = (uint32_t*)rowp + xbase;
#ifndef WRHIP_HOSTSIM
    if (PPL == 4 && nvalid == 4 && (((uintptr_t)dst) & 15) == 0) { *(uint4*)dst = make_uint4(px[0], px[1 % PPL], px[2 % PPL], px[3 % PPL]); return; }
#endif
#pragma unroll
    for (int i = 0; i < PPL; i++) if (i < nvalid) dst[i] = px[i];
  } else {
    uint8_t* dst = rowp + xbase;
    if (PPL == 4 && nvalid == 4) { *(uint32_t*)dst = (px[0] & 0xFF) | ((px[1 % PPL] & 0xFF) << 8) | ((px[2 % PPL] & 0xFF) << 16) | (px[3 % PPL] << 24); return; }
#pragma unroll
    for (int i = 0; i < PPL; i++) if (i < nvalid) dst[i] = (uint8_t)px[i];
  }
}
// targets [t0, t0 + nt) are the span-rows targets of one dependency level; work item = (target, row, piece of 64 lanes x PPL pixels), one wave each
WR_DEVICE void wr_span_rows_body(const WrTargetDesc* __restrict__ targets, const int t0, const int nt, const WrDrawDesc* __restrict__ draws,
                                 const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux, const int block, const int nblocks) {
  const int lane = threadIdx.x & 63;
  const int nwaves = int((nblocks * blockDim.x) >> 6);
#ifdef WRHIP_HOSTSIM
  const int gw = int((block * blockDim.x + threadIdx.x) >> 6);
#else
  const int gw = __builtin_amdgcn_readfirstlane(int((block * blockDim.x + threadIdx.x) >> 6));
#endif
  for (int item = gw;; item += nwaves) {
    int rel = item, ti = 0, pieces = 1;
    for (; ti < nt; ti++) {
      const WrTargetDesc& Tq = targets[t0 + ti];
      pieces = WR_SPAN_PIECES(Tq.width);
      const int n = wr_imax(0, Tq.y_end - Tq.y_begin) * pieces;      // (as the host counts row_items: an empty or inverted row range holds no items)
      if (rel < n) break;
      rel -= n;
    }
    if (ti >= nt) break;
    const WrTargetDesc& T = targets[t0 + ti];
    const int y = T.y_begin + rel / pieces;
    if (WR_SPAN_PPL(T.width) == 4) {
      const int piece_x0 = (rel % pieces) << 8;
      if (T.format == WR_FMT_RGBA8) wr_span_row_piece<WR_FMT_RGBA8, 4>(T, y, piece_x0 + 4 * lane, draws, prims, aux, piece_x0);
      else wr_span_row_piece<WR_FMT_R8, 4>(T, y, piece_x0 + 4 * lane, draws, prims, aux, piece_x0);
    } else {
      const int piece_x0 = (rel % pieces) << 6;
      if (T.format == WR_FMT_RGBA8) wr_span_row_piece<WR_FMT_RGBA8, 1>(T, y, piece_x0 + lane, draws, prims, aux, piece_x0);
      else wr_span_row_piece<WR_FMT_R8, 1>(T, y, piece_x0 + lane, draws, prims, aux, piece_x0);
    }
  }
}
// ---------------------------------------------------------------------------
// Tile rows (WrTargetDesc::rows_mode == 2): the same wave-per-row-piece walk for a PICTURE target that holds a few large
// axis-aligned prims with expensive span shaders -- full-size linear gradients (wrench aligned- / unaligned-gradient), a blurred
// picture composited by brush_image (large-blur-radius) -- beside solids and clears.  In the bin raster those run in the one
// variant that carries every shader replay (324 VGPRs, one wave per SIMD: every latency exposed, profiles/r05_g); here a lane
// holds four pixels and their depth, the prims are applied in submission order with swgl's depth test (LEQUAL / LESS against
// the lane's own samples), a prim nobody's pixel passes is skipped before any evaluation, and a depth-tested prim that consumes
// interpolants gets its row's depth runs (draw_depth_span: the row span minus the spans of the earlier depth writers in front
// of it -- every prim of such a target is a rect, so the sweep is a few scalar comparisons) exactly as wr_build_runs gives them
// to the bins.  Finished rows are stored once, and a second time where a forwarded composite wants them.
WR_DEVICE bool wr_kind_needs_runs(int kind);
WR_DEVICE bool wr_row_runs(const WrTargetDesc& T, const WrPrim* __restrict__ prims, const int p, const int y, WrRuns& R) {
  const WrPrim& P = prims[p];
  const bool less = (P.flags & WR_PF_DEPTH_LESS) != 0;
  const uint32_t z = P.z;
  const int end = wr_imin(T.dw_end, p);
  auto cand = [&](int i, int& lo, int& hi) -> bool {
    const WrPrim& Q = prims[i];
    lo = Q.x0; hi = Q.x1;
    return (Q.flags & WR_PF_DEPTH_WRITE) && Q.kind != WR_PK_NONE && Q.kind != WR_PK_UNSUPPORTED && Q.kind != WR_PK_CLEAR &&
           (less ? Q.z <= z : Q.z < z) && Q.x0 < P.x1 && Q.x1 > P.x0 && y >= Q.y0 && y < Q.y1;
  };
  int nc = 0;
  for (int i = T.dw_first; i < end; i++) { int lo, hi; if (cand(i, lo, hi)) nc++; }
  if (nc == 0) return false;
  // wr_sweep_runs over the candidates (a second time into the pool when the row has more runs than R holds)
  const int b = P.x1;
  auto sweep = [&](int32_t* ext) -> int {
    int n = 0, pos = P.x0;
    while (pos < b) {
      int s0 = pos;
      for (bool moved = true; moved;) {
        moved = false;
        for (int i = T.dw_first; i < end; i++) { int lo, hi; if (cand(i, lo, hi) && lo <= s0 && s0 < hi) { s0 = hi; moved = true; } }
      }
      if (s0 >= b) break;
      int e = b;
      for (int i = T.dw_first; i < end; i++) { int lo, hi; if (cand(i, lo, hi) && hi > lo && lo > s0 && lo < e) e = lo; }
      if (ext) { ext[2 * n] = s0; ext[2 * n + 1] = e; }
      else if (n < WR_MAX_RUNS) { R.s[n] = s0; R.e[n] = e; }
      n++;
      pos = e;
    }
    return n;
  };
  R.ext = nullptr; R.pad = 0;
  int n = sweep(nullptr);
  if (n > WR_MAX_RUNS) {
    int32_t* ext = wr_pool_words_wave(T, 2ull * (unsigned long long)n);
    if (ext) { sweep(ext); R.ext = ext; }
    else n = -2;                       // (the pool is exhausted: reported by the caller, drawn from the span start)
  }
  R.n = n;
  return true;
}
__device__ __noinline__ uint32_t wr_tile_blend(int key, uint32_t dstp, uint32_t sbg, uint32_t sra, const WrDrawDesc* D, const uint32_t* bc) {
  WrWide src; src.bg = sbg; src.ra = sra;
  return wr_blend_rgba8(key, dstp, src, D, bc);
}
__device__ __noinline__ void wr_tile_tex_setup(const WrPrim* Pp, const WrTexDesc* tp, int y, WrTexRow* out) { *out = wr_tex_row(*Pp, *tp, y); }
__device__ __noinline__ void wr_tile_tex_px(const WrPrim* Pp, const WrDrawDesc* D, const WrTexRow* rp, int x, int y, const WrRuns* runs, WrWide* out) {
  const WrTexDesc& t = D->tex[Pp->tex_slot];
  *out = wr_mask_src(*Pp, D, x, y, runs ? wr_tex_pixel(*Pp, t, x, y, runs) : wr_tex_pixel_row(*Pp, t, *rp, x - rp->x0));
}
WR_DEVICE void wr_tile_row_piece(const WrTargetDesc& T, const int y, const int xbase, const WrDrawDesc* __restrict__ draws,
                                 const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux, const int piece_x0) {
  uint32_t px[4], dep[4];
  uint8_t* rowp = (uint8_t*)T.color + (size_t)y * T.stride;
  const int nvalid = wr_iclamp(T.width - xbase, 0, 4);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    px[i] = T.load_color ? (i < nvalid ? ((const uint32_t*)rowp)[xbase + i] : 0u) : T.init_color;
    dep[i] = T.init_depth;
  }
  // The target's prims in submission order.  On the device the wave looks at 64 of them at a time -- lane l fetches the head of prim
  // base + l (box, z, kind, flags), one round trip for all of them -- and walks only the ones that reach this row piece (a ballot, lowest
  // bit first: still submission order); prim by prim every test was a dependent fetch of its own, ~0.4 us each whether or not the prim
  // touched the row (clip-clear: 99 segment prims a tile, a handful per row; profiles/r06_r_tile_rows_ab.txt).
  // ... and a depth cap for the piece, kept in SGPRs over the walk: every pixel of the piece holds a depth <= zcap (the clear value, then the
  // z of every depth-writing prim that spans the whole piece), so a depth-tested prim behind it is hidden on all 256 pixels and is not
  // visited at all (ten full-size opaque gradients front to back: nine of them are, for most pieces).
#ifndef WRHIP_HOSTSIM
  uint32_t zcap = T.init_depth;
  const int piece_x1 = wr_imin(piece_x0 + 256, T.width);
  for (int pbase = T.first_prim; pbase < T.end_prim; pbase += 64) {
    const int lane_ = (int)threadIdx.x & 63;
    bool reach = false;
    int qk = 0, qf = 0, qcov = 0;
    uint32_t qz = 0;
    if (pbase + lane_ < T.end_prim) {
      const WrPrim& Q = prims[pbase + lane_];
      qk = Q.kind; qf = Q.flags; qz = Q.z;
      reach = qk != WR_PK_NONE && qk != WR_PK_UNSUPPORTED && y >= Q.y0 && y < Q.y1 && Q.x1 > piece_x0 && Q.x0 < piece_x0 + 256;
      qcov = (Q.x0 <= piece_x0 && Q.x1 >= piece_x1) ? 1 : 0;
    }
  for (unsigned long long pm = __ballot(reach); pm; pm &= pm - 1ull) {
    const int pb = (int)__builtin_ctzll(pm);
    const int p = pbase + pb;
    {
      const int sk = __builtin_amdgcn_readlane(qk, pb), sf = __builtin_amdgcn_readlane(qf, pb), scov = __builtin_amdgcn_readlane(qcov, pb);
      const uint32_t sz = (uint32_t)__builtin_amdgcn_readlane((int)qz, pb);
      if (sk == WR_PK_CLEAR) {
        if (sf & WR_PF_CLEAR_DEPTH) zcap = scov ? sz : (sz > zcap ? sz : zcap);
      } else if (sf & WR_PF_DEPTH_TEST) {
        if ((sf & WR_PF_DEPTH_LESS) ? sz >= zcap : sz > zcap) continue;          // fails against every pixel's depth
        if ((sf & WR_PF_DEPTH_WRITE) && scov && sz < zcap) zcap = sz;
      }
    }
#else
  {
  for (int p = T.first_prim; p < T.end_prim; p++) {
#endif
    const WrPrim& P = prims[p];
    const int kind = P.kind;
    if (kind == WR_PK_NONE || kind == WR_PK_UNSUPPORTED) continue;
    if (y < P.y0 || y >= P.y1 || P.x1 <= piece_x0 || P.x0 >= piece_x0 + 256) continue;       // (wave-uniform)
    const WrDrawDesc* D = &draws[P.draw];
    const int n0 = xbase - P.x0, len = P.x1 - P.x0;
    const uint32_t z = P.z;
    if (kind == WR_PK_CLEAR) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const bool in = (unsigned)(n0 + i) < (unsigned)len;
        if (P.flags & WR_PF_CLEAR_COLOR) px[i] = in ? P.color[0] : px[i];
        if (P.flags & WR_PF_CLEAR_DEPTH) dep[i] = in ? z : dep[i];
      }
      continue;
    }
    const bool dtest = (P.flags & WR_PF_DEPTH_TEST) != 0, dwrite = (P.flags & WR_PF_DEPTH_WRITE) != 0, dless = (P.flags & WR_PF_DEPTH_LESS) != 0;
    bool in[4];
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      in[i] = (unsigned)(n0 + i) < (unsigned)len;
      if (dtest) {
        in[i] = in[i] && (dless ? z < dep[i] : z <= dep[i]);
        if (dwrite) dep[i] = in[i] ? z : dep[i];
      }
      any = any || in[i];
    }
#ifndef WRHIP_HOSTSIM
    if (__ballot(any) == 0ull) continue;         // hidden (or beside this piece) for every lane: no row setup, no evaluation
#else
    if (!any) continue;
#endif
    if (kind == WR_PK_SOLID) {
#pragma unroll
      for (int i = 0; i < 4; i++) if (in[i]) px[i] = wr_tile_blend(P.blend, px[i], P.color[0], P.color[1], D, P.color);
      continue;
    }
    // depth runs of this row for the kinds whose span shader restarts at every run (wr_kind_needs_runs)
    WrRuns R;
    const WrRuns* runs = nullptr;
    if (dtest && wr_kind_needs_runs(kind) && wr_row_runs(T, prims, p, y, R)) {
      if (R.n == -2) { R.n = 0; if (T.counters && xbase == wr_imax(piece_x0, P.x0 & ~3)) atomicAdd(&T.counters->unsupported_prims, 1u); }
      runs = &R;
    }
    if (kind == WR_PK_GRADIENT) {
      const WrGradRec* Gp = &aux[p].grad;
      WrGrad4 g4;
      if (!runs) g4 = wr_gradient_row4(&P, Gp, D, xbase, y);
#pragma nounroll
      for (int i = 0; i < 4; i++) {
        uint32_t v = px[0];
        if (in[0]) {
          WrWide g = g4.v[0];
          if (runs) g = wr_gradient_row4(&P, Gp, D, xbase + i, y, runs).v[0];
          const WrWide src = wr_mask_src(P, D, xbase + i, y, g);
          v = wr_tile_blend(P.blend, v, src.bg, src.ra, D, nullptr);
        }
        px[0] = px[1]; px[1] = px[2]; px[2] = px[3]; px[3] = v;
        in[0] = in[1]; in[1] = in[2]; in[2] = in[3];
        g4.v[0] = g4.v[1]; g4.v[1] = g4.v[2]; g4.v[2] = g4.v[3];
      }
    } else if (kind == WR_PK_SOLID_MASKED) {
      // a solid under a clip mask: the lane's four mask texels fetched together (1:1 at (x, y) - offset), then applyColor(expand_mask(mask),
      // colour) and the blend per pixel -- what wr_generic_pixel_rgba8 does, pixel after pixel behind a call, with a dependent fetch each
      const WrTexDesc& mt = D->tex[WR_S_CLIP_MASK];
      const uint8_t* mrow = (const uint8_t*)mt.ptr + (size_t)(y - P.mask_off[1]) * mt.stride + (xbase - P.mask_off[0]);
      uint32_t m[4];
#pragma unroll
      for (int i = 0; i < 4; i++) m[i] = in[i] ? (uint32_t)mrow[i] : 0u;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (!in[i]) continue;
        WrWide mm; mm.bg = mm.ra = m[i] | (m[i] << 16);
        const WrWide src = wr_apply_color(mm, P.color);
        px[i] = wr_tile_blend(P.blend, px[i], src.bg, src.ra, D, P.color);
      }
    } else if ((kind == WR_PK_TEX_RGBA8 || kind == WR_PK_TEX_FS) && P.dual && P.blend == WR_BLEND_DUAL_SRC) {
#pragma nounroll
      for (int i = 0; i < 4; i++) {
        uint32_t v = px[0];
        if (in[0]) v = wr_generic_pixel_rgba8(&P, D, xbase + i, y, v, runs);
        px[0] = px[1]; px[1] = px[2]; px[2] = px[3]; px[3] = v;
        in[0] = in[1]; in[1] = in[2]; in[2] = in[3];
      }
    } else if (kind == WR_PK_TEX_RGBA8 || kind == WR_PK_TEX_FS) {
      WrTexRow r;
      if (!runs) wr_tile_tex_setup(&P, &D->tex[P.tex_slot], y, &r);
#pragma nounroll
      for (int i = 0; i < 4; i++) {
        uint32_t v = px[0];
        if (in[0]) {
          WrWide src;
          wr_tile_tex_px(&P, D, &r, xbase + i, y, runs, &src);
          v = wr_tile_blend(P.blend, v, src.bg, src.ra, D, P.color);
        }
        px[0] = px[1]; px[1] = px[2]; px[2] = px[3]; px[3] = v;
        in[0] = in[1]; in[1] = in[2]; in[2] = in[3];
      }
    } else {
      // the host's promise (rect kinds of the solid / image / gradient programs only) does not hold for this prim: reported, not drawn
      if (T.counters && xbase == wr_imax(piece_x0, P.x0 & ~3) && y == wr_imax(P.y0, T.y_begin)) atomicAdd(&T.counters->unsupported_prims, 1u);
    }
  }
  }
  if (nvalid <= 0) return;
  uint32_t* dst = (uint32_t*)rowp + xbase;
#ifndef WRHIP_HOSTSIM
  if (nvalid == 4 && (((uintptr_t)dst) & 15) == 0) *(uint4*)dst = make_uint4(px[0], px[1], px[2], px[3]);
  else
#endif
  {
#pragma unroll
    for (int i = 0; i < 4; i++) if (i < nvalid) dst[i] = px[i];
  }
  if (T.fwd_color) {
    // forwarded composite (WrTargetDesc::fwd_*): the same pixels a second time, at their place in the target that would have copied them
    const int Y = T.fwd_y0 + T.fwd_ys * y, fX = xbase + T.fwd_dx;
    if (Y >= T.fwd_clip[1] && Y < T.fwd_clip[3]) {
      uint32_t* frow = (uint32_t*)((uint8_t*)T.fwd_color + (size_t)Y * T.fwd_stride);
#pragma unroll
      for (int i = 0; i < 4; i++) if (i < nvalid && fX + i >= T.fwd_clip[0] && fX + i < T.fwd_clip[2]) frow[fX + i] = px[i];
    }
  }
}
WR_DEVICE void wr_tile_rows_body(const WrTargetDesc* __restrict__ targets, const int t0, const int nt, const WrDrawDesc* __restrict__ draws,
                                 const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux, const int block, const int nblocks) {
  const int lane = threadIdx.x & 63;
  const int nwaves = int((nblocks * blockDim.x) >> 6);
#ifdef WRHIP_HOSTSIM
  const int gw = int((block * blockDim.x + threadIdx.x) >> 6);
#else
  const int gw = __builtin_amdgcn_readfirstlane(int((block * blockDim.x + threadIdx.x) >> 6));
#endif
  for (int item = gw;; item += nwaves) {
    int rel = item, ti = 0, pieces = 1;
    for (; ti < nt; ti++) {
      const WrTargetDesc& Tq = targets[t0 + ti];
      pieces = (Tq.width + 255) >> 8;
      const int n = wr_imax(0, Tq.y_end - Tq.y_begin) * pieces;      // (as the host counts row_items: an empty or inverted row range holds no items)
      if (rel < n) break;
      rel -= n;
    }
    if (ti >= nt) break;
    const WrTargetDesc& T = targets[t0 + ti];
    const int y = T.y_begin + rel / pieces, piece_x0 = (rel % pieces) << 8;
    wr_tile_row_piece(T, y, piece_x0 + 4 * lane, draws, prims, aux, piece_x0);
  }
}
#ifndef WR_INST_ONLY      /* (not a template: defined by wrhip.hip alone) */
__global__ void __launch_bounds__(256, 4) wr_tile_rows_kernel(const WrTargetDesc* __restrict__ targets, int t0, int nt, const WrDrawDesc* __restrict__ draws,
                                                              const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux) {
  wr_tile_rows_body(targets, t0, nt, draws, prims, aux, (int)blockIdx.x, (int)gridDim.x);
}
#endif
#ifndef WR_INST_ONLY      /* (not a template: defined by wrhip.hip alone) */
__global__ void __launch_bounds__(256, 2) wr_span_rows_kernel(const WrTargetDesc* __restrict__ targets, int t0, int nt, const WrDrawDesc* __restrict__ draws,
                                                              const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux) {
  wr_span_rows_body(targets, t0, nt, draws, prims, aux, (int)blockIdx.x, (int)gridDim.x);
}
#endif

// min of two 16-bit fields packed in a u32 (v_pk_min_u16)
WR_DEVICE uint32_t wr_pk_min_u16(uint32_t a, uint32_t b) {
#ifdef WRHIP_HOSTSIM
  uint32_t lo = (a & 0xFFFF) < (b & 0xFFFF) ? (a & 0xFFFF) : (b & 0xFFFF);
  uint32_t hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
  return lo | (hi << 16);
#else
  typedef unsigned short wr_us2 __attribute__((ext_vector_type(2)));
  wr_us2 r = __builtin_elementwise_min(__builtin_bit_cast(wr_us2, a), __builtin_bit_cast(wr_us2, b));
  return __builtin_bit_cast(uint32_t, r);
#endif
}
// ((u >> 8) & 0x00FF00FF) in one v_perm_b32: bytes (u.b1, 0, u.b3, 0)
WR_DEVICE uint32_t wr_hi_bytes(uint32_t u) {
#ifdef WRHIP_HOSTSIM
  return (u >> 8) & 0x00FF00FFu;
#else
  return __builtin_amdgcn_perm(0u, u, 0x0c030c01u);
#endif
}
WR_DEVICE uint32_t wr_mul24(uint32_t a, uint32_t b) {
#ifdef WRHIP_HOSTSIM
  return a * b;
#else
  return __umul24(a, b);
#endif
}

// p = hi_bytes(p * K + C), in place.  Written as tied-operand asm on the device: v_mad_u32_u24 /
// v_perm_b32 are three-address, and left to itself the compiler computes the 32 pixel registers
// of a prim into a second register set and copies them back at the loop back-edge (16 v_mov_b64
// per prim on top of the 64 useful VALU instructions -- seen in the round-1 ISA).
WR_DEVICE void wr_fold_inplace(uint32_t& p, uint32_t K, uint32_t C) {
#ifdef WRHIP_HOSTSIM
  p = ((p * K + C) >> 8) & 0x00FF00FFu;
#else
  asm("v_mad_u32_u24 %0, %0, %1, %2\n\tv_perm_b32 %0, 0, %0, %3" : "+v"(p) : "s"(K), "v"(C), "v"(0x0c030c01u));
#endif
}
// Lane masks: on the device a coverage predicate is kept as the 64-bit ballot of its compare
// (an SGPR pair straight out of v_cmp), so combining the column and row predicates of a pixel is
// one scalar AND instead of VALU selects; the host simulation keeps plain bools.
#ifdef WRHIP_HOSTSIM
typedef bool wr_lanemask;
#define WR_LANEMASK(cond) (cond)
#else
typedef unsigned long long wr_lanemask;
#define WR_LANEMASK(cond) __builtin_amdgcn_ballot_w64(cond)
#endif
// wr_fold_inplace for the lanes of `m` only (both channel pairs of one pixel): EXEC is narrowed
// to the covered lanes around the four instructions, so a partially covered strip costs the same
// four VALU instructions per pixel as a fully covered one plus two scalar instructions.
WR_DEVICE void wr_fold_masked(uint32_t& lo, uint32_t& hi, uint32_t K, uint32_t Clo, uint32_t Chi, wr_lanemask m) {
#ifdef WRHIP_HOSTSIM
  if (m) { lo = ((lo * K + Clo) >> 8) & 0x00FF00FFu; hi = ((hi * K + Chi) >> 8) & 0x00FF00FFu; }
#else
  unsigned long long saved;
  asm("s_and_saveexec_b64 %2, %3\n\t"
      "v_mad_u32_u24 %0, %0, %4, %5\n\tv_perm_b32 %0, 0, %0, %7\n\t"
      "v_mad_u32_u24 %1, %1, %4, %6\n\tv_perm_b32 %1, 0, %1, %7\n\t"
      "s_mov_b64 exec, %2"
      : "+v"(lo), "+v"(hi), "=&s"(saved)
      : "s"(m), "s"(K), "v"(Clo), "v"(Chi), "v"(0x0c030c01u)
      : "scc");
#endif
}

// d = v in the lanes of m
WR_DEVICE void wr_select_masked(uint32_t& d, uint32_t v, wr_lanemask m) {
#ifdef WRHIP_HOSTSIM
  if (m) d = v;
#else
  asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(d) : "v"(v), "s"(m));
#endif
}
