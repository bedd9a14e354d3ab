// wrhip_k_pixels.h -- part of the gfx950 kernels of libwrhip: included by wrhip_kernels.h, in its order, and by nothing else.
// Per-pixel and per-row evaluators the raster stages share: blits, texture filters, gradients, filters, mix-blend, blur, clip masks, box shadows.
#pragma once

// Scatter queued texture uploads from the staging mirror to their textures.
// 8 workgroups per segment; 16-byte lanes where the rows allow it.
// `parts` workgroups per segment (the host sizes it for the largest segment of the batch: one per 64 KB, 8 .. 256)
// (the body: also run by the first workgroups of a flush's setup-carrying launch -- WrSetupArgs::up_*, "fused scatter" --, which takes the
// scatter off the frame's critical path: the setup stage reads whole-texture uploads of data textures straight from the staging mirror)
WR_DEVICE void wr_upload_body(const WrUploadSeg* __restrict__ segs, int n_segs, int parts, int bid) {
  const int si = bid / parts, part = bid % parts;
  if (si >= n_segs) return;
  const WrUploadSeg sg = segs[si];
  const size_t total = (size_t)sg.row_bytes * sg.rows;
  const bool vec = ((sg.row_bytes & 15) == 0) && ((sg.dst_stride & 15) == 0) && (((uintptr_t)sg.src & 15) == 0) && (((uintptr_t)sg.dst & 15) == 0);
  if (vec) {
    const size_t n16 = total >> 4, per_row = sg.row_bytes >> 4;
    for (size_t i = (size_t)part * blockDim.x + threadIdx.x; i < n16; i += (size_t)parts * blockDim.x) {
      size_t r = i / per_row, c = i - r * per_row;
      ((uint4*)((uint8_t*)sg.dst + r * sg.dst_stride))[c] = ((const uint4*)sg.src)[i];
    }
  } else {
    for (size_t i = (size_t)part * blockDim.x + threadIdx.x; i < total; i += (size_t)parts * blockDim.x) {
      size_t r = i / sg.row_bytes, c = i - r * sg.row_bytes;
      ((uint8_t*)sg.dst)[r * sg.dst_stride + c] = sg.src[i];
    }
  }
}
#ifndef WR_INST_ONLY      /* (not a template: defined by wrhip.hip alone, not by the instantiation units wrhip_inst.hip) */
__global__ void wr_upload_kernel(const WrUploadSeg* __restrict__ segs, int n_segs, int parts) {
  wr_upload_body(segs, n_segs, parts, (int)blockIdx.x);
}
#endif

// BlitFramebuffer with scaling / flipping / format conversion (composite.h:62-283, 285-418); one thread per dest pixel.
#ifndef WR_INST_ONLY      /* (not a template: defined by wrhip.hip alone, not by the instantiation units wrhip_inst.hip) */
__global__ void wr_blit_kernel(WrBlitArgs a) {
  const int bw = a.bx1 - a.bx0, bh = a.by1 - a.by0;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)bw * bh) return;
  const int i = int(idx % bw), j = int(idx / bw);
  const int X = a.bx0 + i, Y = a.by0 + j;
  uint8_t* dp = (uint8_t*)a.dst + (size_t)(a.dry0 + Y) * a.dst_stride + (size_t)(a.drx0 + X) * a.dbpp;
  if (!a.linear) {
    // scale_row's stepping: the source column of dest column X is floor(srcWidth * X / dstWidth), rows likewise
    const int sx = a.srx0 + int((long long)a.srw * X / a.drw);
    const int syr = int((long long)a.srh * Y / a.drh);
    const int sy = a.invert_y ? a.sry0 + a.srh - 1 - syr : a.sry0 + syr;
    const uint8_t* sp = (const uint8_t*)a.src + (size_t)sy * a.src_stride + (size_t)sx * a.sbpp;
    if (a.composite && a.sbpp == 4 && a.dbpp == 4) {       // copy_row / scale_row<true>: src + dst - muldiv255(dst, alphas(src))
      uint32_t sv, dv;
      __builtin_memcpy(&sv, sp, 4); __builtin_memcpy(&dv, dp, 4);
      const uint32_t o = wr_blend_rgba8(WR_BLEND_PREMULT, dv, wr_unpack(sv), nullptr);
      __builtin_memcpy(dp, &o, 4);
      return;
    }
    if (a.sbpp == a.dbpp) { for (int k = 0; k < a.dbpp; k++) dp[k] = sp[k]; return; }
    // convert_pixel (composite.h:5-70)
    uint32_t v = 0;
    if (a.sbpp == 4) __builtin_memcpy(&v, sp, 4); else if (a.sbpp == 2) { uint16_t h; __builtin_memcpy(&h, sp, 2); v = h; } else v = sp[0];
    if (a.dbpp == 4) {
      const uint32_t o = a.sbpp == 1 ? ((v << 16) | 0xFF000000u) : (((v & 0x00FFu) << 16) | (v & 0xFF00u) | 0xFF000000u);
      __builtin_memcpy(dp, &o, 4);
    } else if (a.dbpp == 1) {
      dp[0] = a.sbpp == 4 ? uint8_t((v >> 16) & 0xFF) : uint8_t(v & 0xFF);
    } else {
      const uint16_t o = a.sbpp == 4 ? uint16_t(((v >> 16) & 0x00FF) | (v & 0xFF00)) : uint16_t(v);
      __builtin_memcpy(dp, &o, 2);
    }
    return;
  }
  // linear_blit: srcUV = quantize(srcReq.origin (+ size when flipped) + srcDUV * (dstBounds.origin + 0.5)), stepped by
  // srcDUV * 128 per pixel (init_interp lanes, then one add of 4 * srcDU per 4-pixel chunk) and per row
  float u0 = float(a.srx0), v0 = float(a.sry0);
  float du = float(a.srw) / float(a.drw), dv = float(a.srh) / float(a.drh);
  if (a.invert_x) { u0 += float(a.srw); du = -du; }
  if (a.invert_y) { v0 += float(a.srh); dv = -dv; }
  u0 += du * (float(a.bx0) + 0.5f); v0 += dv * (float(a.by0) + 0.5f);
  u0 = u0 * 128.0f + (0.5f - 0.5f * 128.0f); v0 = v0 * 128.0f + (0.5f - 0.5f * 128.0f);
  du *= 128.0f; dv *= 128.0f;
  float lu = u0;
  for (int k = 0; k < (i & 3); k++) lu = lu + du;
  lu = wr_accum(lu, 4.0f * du, i >> 2);
  const float lv = wr_accum(v0, dv, j);
  WrTexDesc t;
  t.ptr = a.src; t.width = a.sw; t.height = a.sh; t.stride = a.sbpp == 4 ? a.src_stride / 4 : (a.sbpp == 2 ? a.src_stride / 2 : a.src_stride);
  t.format = a.sbpp == 4 ? WR_FMT_RGBA8 : WR_FMT_R8; t.linear = 1; t.sw = float(a.sw); t.sh = float(a.sh);
  if (a.sbpp == 4) {
    const WrWide w = wr_sample_linear_rgba8(t, int(lu), int(lv));
    uint32_t o = wr_pack(w);
    if (a.composite) {                                       // linear_row_blit<true>
      uint32_t dv_;
      __builtin_memcpy(&dv_, dp, 4);
      o = wr_blend_rgba8(WR_BLEND_PREMULT, dv_, w, nullptr);
    }
    __builtin_memcpy(dp, &o, 4);
  } else {
    dp[0] = (uint8_t)wr_pack1(uint32_t(wr_sample_linear_r8(t, int(lu), int(lv))) & 0xFFFF);
  }
}
#endif

// A solid colour on a general quad, one row at a time: the row's span from the edge instances of its run (aa_span / aa_edge /
// aa_dist, rasterize.h:480-562) -- the two edge sums are what costs (Edge::nextRow, one add per row: wr_accum), so they are
// evaluated once per lane-row and shared by the row's pixels.
// (a lane's rows are 4 apart: the next row's sums are the previous row's plus four more adds when both rows lie in the same
// run of the walk -- `prev`, with prev.ok -- instead of the whole sum from the run's first row again)
struct WrQuadRowS { int ok, s0, s1, la1; float lstart, lend, rstart, rend; float xl, xr; int si, y; };
__device__ __noinline__ WrQuadRowS wr_quad_row_setup(const WrQuadRec* Qp, int y, float pxl, float pxr, int psi, int py_) {
  const WrQuadRec& Q = *Qp;
  WrQuadRowS R;
  R.ok = 0; R.s0 = R.s1 = R.la1 = 0; R.lstart = R.rstart = 256.0f; R.lend = R.rend = 0.0f;
  R.xl = R.xr = 0.0f; R.si = -1; R.y = y;
  int si = -1;
  for (int i = 0; i < Q.nseg; i++) if (y >= Q.seg[i].row_a && y < Q.seg[i].row_b) si = i;
  if (si < 0) return R;
  const WrQuadSeg& S = Q.seg[si];
  float xl, xr;
  if (si == psi && y > py_ && y - py_ <= 8) {
    xl = pxl; xr = pxr;
    for (int i = py_; i < y; i++) { xl = xl + S.ls; xr = xr + S.rs; }
  } else {
    wr_quad_row_x(Q, S, y, xl, xr);   // Edge::nextRow, one add per row
  }
  R.xl = xl; R.xr = xr; R.si = si;
  R.ok = 1;
  if (!Q.aa) {
    R.s0 = int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f)); R.s1 = int(floorf(wr_clamp(xr, S.b0, S.b1) + 0.5f));
    return R;
  }
  // aa_edge: masked edges use the row's x intercepts rounded out, the others the rounded x
  const float radl = 0.5f * fabsf(S.ls), radr = 0.5f * fabsf(S.rs);
  R.s0 = S.lmask ? int(floorf(wr_clamp(xl - radl, S.b0, S.b1))) : int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f));
  R.la1 = S.lmask ? int(ceilf(wr_clamp(xl + radl, S.b0, S.b1))) : int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f));
  R.s1 = S.rmask ? int(ceilf(wr_clamp(xr + radr, S.b0, S.b1))) : int(floorf(wr_clamp(xr, S.b0, S.b1) + 0.5f));
  // aa_dist
  if (S.lmask) { const float dx = (-1.0f * 256.0f) * (1.0f / sqrtf(1.0f + S.ls * S.ls)); R.lstart = 128.0f + dx * (xl - 0.5f); R.lend = -dx; }
  if (S.rmask) { const float dx = (1.0f * 256.0f) * (1.0f / sqrtf(1.0f + S.rs * S.rs)); R.rstart = 128.0f + dx * (xr - 0.5f); R.rend = -dx; }
  return R;
}
// One pixel of that row: its coverage (DO_AA, blend.h:433-446), the blend.  Returns the new pixel in the low word and
// 1 << 32 when the pixel is inside the row's span.
__device__ __noinline__ unsigned long long wr_quad_row_pixel_rgba8(WrQuadRowS R, int aa, const WrDrawDesc* D, int blend, uint32_t c0, uint32_t c1,
                                                                    int x, uint32_t dstp_, const WrRuns* runs = nullptr) {
  const unsigned long long dstp = dstp_;
  const unsigned long long HIT = 1ull << 32;
  if (!R.ok || x < R.s0 || x >= R.s1) return dstp;
  WrWide src; src.bg = c0; src.ra = c1;
  if (!aa) return HIT | wr_blend_rgba8(blend, dstp_, src, D);
  // the 4-pixel chunks DO_AA sees start at the span start -- with depth runs, at the start of the run holding x
  int cs = R.s0;
  if (runs) { const int k = wr_find_run(runs, x); if (k >= 0) cs = wr_run_s(runs, k); }
  const int n = x - cs, lane = n & 3, base = cs + (n & ~3);
  const float off = float(4 * (base - R.la1));
  const float dl = (R.lstart + float(R.la1 + lane) * R.lend) + (R.lend / 4.0f) * off;
  const float dr = (R.rstart + float(R.la1 + lane) * R.rend) + (R.rend / 4.0f) * off;
  const uint32_t cov = uint32_t(int(wr_clamp(wr_min(dl, dr), 0.0f, 256.0f) * 1.0f + 0.5f)) & 0xFFFF;
  src.bg = ((((c0 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c0 >> 16) * cov) & 0xFFFF) >> 8) << 16);
  src.ra = ((((c1 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c1 >> 16) * cov) & 0xFFFF) >> 8) << 16);
  return HIT | wr_blend_rgba8(blend, dstp_, src, D);
}

// The edge values of one row of a general quad -- x, the two interpolants and, under perspective, z and 1/w of the run's left
// and right edge: Edge::nextRow, one add per row, i.e. a row-by-row sum (wr_accum) per value.  They are what a pixel of such
// a prim costs, and they only depend on the row: a lane keeps the last row it evaluated (its four pixels of a row share it),
// and its next row, four rows down in the same run, continues the sums with four adds each.
struct WrQuadRowCache { int y, si; float xl, xr, lu, lv, ru, rv, wl, wr, zl, zr; };
WR_DEVICE void wr_quad_row_edges(const WrQuadRec& Q, int si, int y, WrQuadRowCache* C, WrQuadRowCache& L) {
  if (C && C->si == si && C->y == y) { L = *C; return; }
  const WrQuadSeg& S = Q.seg[si];
  if (const float* e = Q.rowtab_stride >= 6 ? wr_quad_rowtab_entry(Q, y) : nullptr) {
    // the row's edge values as the setup stage summed them (wr_quad_build_rowtab)
    L.xl = e[0]; L.xr = e[1]; L.lu = e[2]; L.lv = e[3]; L.ru = e[4]; L.rv = e[5];
    L.wl = L.wr = L.zl = L.zr = 0.0f;
    if (Q.rowtab_stride >= 10) { L.wl = e[6]; L.wr = e[7]; L.zl = e[8]; L.zr = e[9]; }
    L.y = y; L.si = si;
    if (C) *C = L;
    return;
  }
  const int dy = C ? y - C->y : 0;
  if (C && C->si == si && dy > 0 && dy <= 8) {
    L = *C;
    for (int i = 0; i < dy; i++) {
      L.xl = L.xl + S.ls; L.xr = L.xr + S.rs;
      L.lu = L.lu + S.luvs[0]; L.lv = L.lv + S.luvs[1]; L.ru = L.ru + S.ruvs[0]; L.rv = L.rv + S.ruvs[1];
      if (Q.pad || Q.base_kind == WR_PK_MIX_BLEND) {
        L.wl = L.wl + Q.persp.lws[si]; L.wr = L.wr + Q.persp.rws[si];
        L.zl = L.zl + Q.persp.lzs[si]; L.zr = L.zr + Q.persp.rzs[si];
      }
    }
  } else {
    L.xl = wr_accum(S.lx, S.ls, y - S.lrow); L.xr = wr_accum(S.rx, S.rs, y - S.rrow);
    L.lu = wr_accum(S.luv[0], S.luvs[0], y - S.lrow); L.lv = wr_accum(S.luv[1], S.luvs[1], y - S.lrow);
    L.ru = wr_accum(S.ruv[0], S.ruvs[0], y - S.rrow); L.rv = wr_accum(S.ruv[1], S.ruvs[1], y - S.rrow);
    L.wl = L.wr = L.zl = L.zr = 0.0f;
    if (Q.pad || Q.base_kind == WR_PK_MIX_BLEND) {
      L.wl = wr_accum(Q.persp.lw[si], Q.persp.lws[si], y - S.lrow); L.wr = wr_accum(Q.persp.rw[si], Q.persp.rws[si], y - S.rrow);
      L.zl = wr_accum(Q.persp.lz[si], Q.persp.lzs[si], y - S.lrow); L.zr = wr_accum(Q.persp.rz[si], Q.persp.rzs[si], y - S.rrow);
    }
  }
  L.y = y; L.si = si;
  if (C) *C = L;
}

// The packed depth of one pixel of a perspective quad (draw_perspective_spans, rasterize.h:1236-1258 + packDepth :345): the
// row's edges give z at the span's ends (Point3D edges, stepped once per row), stepZW = (right.zw - left.zw) / (right.x -
// left.x), gl_FragCoord.z = init_interp(z at the span start's pixel centre, step) -- three sequential adds -- and every
// 4-pixel chunk, drawn or skipped, adds 4 * step (step_perspective, program.h:145-148) -- in a program that has varyings;
// one without (WrQuadRec::pad == 1) runs its chunks through the plain run / skip, which leave gl_FragCoord.z alone.
__device__ __noinline__ uint32_t wr_persp_depth(const WrQuadRec* Qp, int x, int y, WrQuadRowCache* cache = nullptr) {
  const WrQuadRec& Q = *Qp;
  int si = -1;
  for (int i = 0; i < Q.nseg; i++) if (y >= Q.seg[i].row_a && y < Q.seg[i].row_b) si = i;
  if (si < 0) return 0xFFFFFFFFu;
  const WrQuadSeg& S = Q.seg[si];
  WrQuadRowCache E;
  wr_quad_row_edges(Q, si, y, cache, E);
  const float xl = E.xl, xr = E.xr;
  int s0;
  if (!Q.aa) s0 = int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f));
  else s0 = S.lmask ? int(floorf(wr_clamp(xl - 0.5f * fabsf(S.ls), S.b0, S.b1))) : int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f));
  const float zl = E.zl, zr = E.zr;
  float stepScale = 1.0f / (xr - xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  const float stepZ = (zr - zl) * stepScale;
  const float z0 = zl + stepZ * ((float(s0) + 0.5f) - xl);
  const int k = x - s0;
  if (k < 0) return 0xFFFFFFFFu;
  float zi = z0;
  for (int i = 0; i < (k & 3); i++) zi = zi + stepZ;
  const float zc = Q.pad == 2 ? wr_accum(zi, stepZ * 4.0f, k >> 2) : zi;
  return uint32_t(int(zc * 16777215.0f));
}

// One pixel of an anti-aliased solid quad (DO_AA, blend.h:433-446): src = muldiv256(src, coverage)
// ahead of the blend.  Out of line: its float math must not cost the textured kernels registers.
__device__ __noinline__ uint32_t wr_aa_pixel_rgba8(const WrAARec* Ap, const WrDrawDesc* D, int blend, uint32_t c0, uint32_t c1,
                                                    int n, int x0, uint32_t dstp) {
  const WrAARec& A = *Ap;
  const int lane = n & 3, base = x0 + (n & ~3);
  const float off = float(4 * (base - A.laa_end));
  const float dl = (A.lstart + float(A.laa_end + lane) * A.lend) + (A.lend / 4.0f) * off;
  const float dr = (A.rstart + float(A.laa_end + lane) * A.rend) + (A.rend / 4.0f) * off;
  const uint32_t cov = uint32_t(int(wr_clamp(wr_min(dl, dr), 0.0f, 256.0f) * 1.0f + 0.5f)) & 0xFFFF;
  WrWide src;
  src.bg = ((((c0 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c0 >> 16) * cov) & 0xFFFF) >> 8) << 16);
  src.ra = ((((c1 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c1 >> 16) * cov) & 0xFFFF) >> 8) << 16);
  return wr_blend_rgba8(blend, dstp, src, D);
}

// MASK_ blend keys (blend.h:458-460): src = muldiv255(src, expand_mask(clip mask texel)); the mask is
// sampled 1:1 at (x, y) - swgl_ClipMaskOffset (get_clip_mask, blend.h:357-360)
WR_DEVICE WrWide wr_mask_src(const WrPrim& P, const WrDrawDesc* D, int x, int y, WrWide src) {
  if (!(P.flags & WR_PF_MASKED)) return src;
  const WrTexDesc& mt = D->tex[WR_S_CLIP_MASK];
  const uint32_t m = ((const uint8_t*)mt.ptr)[(size_t)(y - P.mask_off[1]) * mt.stride + (x - P.mask_off[0])];
  const uint32_t mm = m | (m << 16);
  WrWide r;
  r.bg = wr_muldiv255_2(src.bg, mm); r.ra = wr_muldiv255_2(src.ra, mm);
  return r;
}

// brush_image ... DUAL_SOURCE_BLENDING under GL_ONE, GL_ONE_MINUS_SRC1_COLOR (blend.h:496-511): main() writes the colour
// v_color * texel and a second one, texel * swizzle.x + texel.aaaa * swizzle.y; dst' = src + dst - dst x second (under a clip
// mask both terms are scaled by it).  `tx`: the texel main() sampled, (r, g, b, a) floats.
WR_DEVICE uint32_t wr_dual_blend(const WrPrim& P, const WrDrawDesc* D, int x, int y, uint32_t dstp, const float (&tx)[4], uint32_t cov, bool aa) {
  const float sx = P.dual_swz, sy = P.dual == 2 ? -P.dual_swz : 0.0f;
  uint32_t pc[2], ps[2];
  wr_pack_color(wf4{P.fcolor[0] * tx[0], P.fcolor[1] * tx[1], P.fcolor[2] * tx[2], P.fcolor[3] * tx[3]}, pc);
  wr_pack_color(wf4{tx[0] * sx + tx[3] * sy, tx[1] * sx + tx[3] * sy, tx[2] * sx + tx[3] * sy, tx[3] * sx + tx[3] * sy}, ps);
  WrWide s2; s2.bg = pc[0]; s2.ra = pc[1];
  const WrWide dst = wr_unpack(dstp);
  WrWide second; second.bg = wr_muldiv255_2(ps[0], dst.bg); second.ra = wr_muldiv255_2(ps[1], dst.ra);    // applyColor(dst, secondary)
  // (anti-aliased prims of the REPETITION key, blend.h:513-530: AA_BLEND_KEY scales source AND secondary colour by the coverage,
  // AA_MASK_BLEND_KEY scales the mask by it)
  if (P.flags & WR_PF_MASKED) {
    const WrTexDesc& mt = D->tex[WR_S_CLIP_MASK];
    uint32_t m = ((const uint8_t*)mt.ptr)[(size_t)(y - P.mask_off[1]) * mt.stride + (x - P.mask_off[0])];
    if (aa) m = ((m * cov) & 0xFFFFu) >> 8;
    const uint32_t mm = m | (m << 16);
    s2.bg = wr_muldiv255_2(s2.bg, mm); s2.ra = wr_muldiv255_2(s2.ra, mm);
    second.bg = wr_muldiv255_2(second.bg, mm); second.ra = wr_muldiv255_2(second.ra, mm);
  } else if (aa) {
    auto sc = [&](uint32_t c) { return ((((c & 0xFFFFu) * cov) & 0xFFFFu) >> 8) | (((((c >> 16) * cov) & 0xFFFFu) >> 8) << 16); };
    s2.bg = sc(s2.bg); s2.ra = sc(s2.ra); second.bg = sc(second.bg); second.ra = sc(second.ra);
  }
  WrWide res;
  res.bg = wr_sub2(wr_add2(s2.bg, dst.bg), second.bg);
  res.ra = wr_sub2(wr_add2(s2.ra, dst.ra), second.ra);
  return wr_pack(res);
}
// Generic (slow-path) pixel: any prim kind / blend key, one pixel at a time.
// Kept out of line so the fast paths below stay small and the 16 pixels of a
// lane stay in registers.
WR_DEVICE void wr_texture_rgba_f(const WrTexDesc& t, float cu, float cv, float (&c)[4]);
__device__ __noinline__ uint32_t wr_generic_pixel_rgba8(const WrPrim* Pp, const WrDrawDesc* D, int x, int y, uint32_t dstp, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  WrWide src;
  if (P.kind == WR_PK_SOLID) {
    src.bg = P.color[0]; src.ra = P.color[1];
  } else if (P.kind == WR_PK_SOLID_MASKED) {
    // applyColor(expand_mask(mask), colour) (swgl_ext.h:11-23); mask texel 1:1 at (x,y) - offset
    const WrTexDesc& mt = D->tex[WR_S_CLIP_MASK];
    const uint32_t m = ((const uint8_t*)mt.ptr)[(size_t)(y - P.mask_off[1]) * mt.stride + (x - P.mask_off[0])];
    WrWide mm; mm.bg = mm.ra = m | (m << 16);
    src = wr_apply_color(mm, P.color);
  } else if (P.dual && P.kind == WR_PK_TEX_FS && P.blend == WR_BLEND_DUAL_SRC) {
    // brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING under GL_ONE, GL_ONE_MINUS_SRC1_COLOR (wr_dual_blend)
    const WrTexDesc& t = D->tex[P.tex_slot];
    const WrTexRow r = wr_tex_row(P, t, y, runs, x);
    float cu, cv;
    wr_tex_tail_uv(P, r, x - r.x0, cu, cv);
    float tx[4];
    wr_texture_rgba_f(t, cu, cv, tx);
    return wr_dual_blend(P, D, x, y, dstp, tx);
  } else {
    src = wr_mask_src(P, D, x, y, wr_tex_pixel(P, D->tex[P.tex_slot], x, y, runs));
  }
  return wr_blend_rgba8(P.blend, dstp, src, D, P.color);
}

struct WrGrad4 { WrWide v[4]; };
__device__ __noinline__ WrGrad4 wr_gradient_row4(const WrPrim* Pp, const WrGradRec* Gp, const WrDrawDesc* D, int x, int y, const WrRuns* runs);
__device__ __noinline__ WrWide wr_filter_pixel(const WrPrim* Pp, const WrFilterRec* Fp, const WrDrawDesc* D, int x, int y, const WrRuns* runs);
__device__ __noinline__ WrWide wr_filter_eval(const WrPrim* Pp, const WrFilterRec* Fp, const WrDrawDesc* D, float cu, float cv);
__device__ __noinline__ WrWide wr_gradient_main(const WrGradRec* Gp, const WrDrawDesc* D, float lu, float lv);
__device__ __noinline__ WrWide wr_quad_mask_pixel(const WrPrim* Pp, const WrClipRec* Cp, const WrDrawDesc* D, int x, int y, const WrRuns* runs);
__device__ __noinline__ WrWide wr_yuv_pixel(const WrPrim* Pp, const WrYuvRec* Yp, const WrDrawDesc* D, int x, int y, const WrRuns* runs);
__device__ __noinline__ WrWide wr_mix_blend_pixel(const WrPrim* Pp, const WrMixRec* Mp, const WrDrawDesc* D, int x, int y, const WrRuns* runs);
__device__ __noinline__ WrWide wr_mix_blend_main(const WrMixRec* Mp, const WrDrawDesc* D, float bu, float bv, float su, float sv);
__device__ __noinline__ WrWide wr_svg_filter_pixel(const WrPrim* Pp, const WrSvgRec* Sp, const WrDrawDesc* D, int x, int y, const WrRuns* runs);
WR_DEVICE float wr_r8_texture(const WrTexDesc& t, float u, float v);
WR_DEVICE WrWide wr_quad_mask_eval(const WrClipRec& C, float f0x, float f0y, float f1x, float f1y, float qx, float qy);
// One pixel of a textured prim on a general quad and / or with swgl_antiAlias (WR_PK_TEX_QUAD): this row's span and the
// pixel's coverage as in wr_quad_pixel_rgba8, the edge interpolants stepped row by row (Edge::nextRow), then the base
// kind's span shader / main() evaluation of pixel x - span.start, DO_AA ahead of the clip mask (blend.h:452-460).
__device__ __noinline__ unsigned long long wr_quad_tex_pixel_rgba8(const WrPrim* Pp, const WrQuadRec* Qp, const WrDrawDesc* D, int x, int y,
                                                                    uint32_t dstp_, const WrRuns* runs = nullptr, WrQuadRowCache* cache = nullptr) {
  const WrQuadRec& Q = *Qp;
  const unsigned long long dstp = dstp_;
  const unsigned long long HIT = 1ull << 32;
  int si = -1;
  for (int i = 0; i < Q.nseg; i++) if (y >= Q.seg[i].row_a && y < Q.seg[i].row_b) si = i;
  if (si < 0) return dstp;
  const WrQuadSeg& S = Q.seg[si];
  WrQuadRowCache E;
  wr_quad_row_edges(Q, si, y, cache, E);
  const float xl = E.xl, xr = E.xr;
  int s0, s1;
  uint32_t cov = 256;
  bool aa_skip = false;
  if (!Q.aa) {
    s0 = int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f)); s1 = int(floorf(wr_clamp(xr, S.b0, S.b1) + 0.5f));
    // (the prim's box holds every span of the walk, so this changes nothing -- except under gl_ClipDistance, whose cut the
    // setup stage folded into the box: span.intersect(clip_distance_range), rasterize.h:953-955)
    s0 = wr_imax(s0, Pp->x0); s1 = wr_imin(s1, Pp->x1);
    if (x < s0 || x >= s1) return dstp;
  } else {
    const float radl = 0.5f * fabsf(S.ls), radr = 0.5f * fabsf(S.rs);
    const int la0 = S.lmask ? int(floorf(wr_clamp(xl - radl, S.b0, S.b1))) : int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f));
    const int la1 = S.lmask ? int(ceilf(wr_clamp(xl + radl, S.b0, S.b1))) : int(floorf(wr_clamp(xl, S.b0, S.b1) + 0.5f));
    const int ra1 = S.rmask ? int(ceilf(wr_clamp(xr + radr, S.b0, S.b1))) : int(floorf(wr_clamp(xr, S.b0, S.b1) + 0.5f));
    if (x < la0 || x >= ra1) return dstp;
    float lstart = 256.0f, lend = 0.0f, rstart = 256.0f, rend = 0.0f;
    if (S.lmask) { const float dx = (-1.0f * 256.0f) * (1.0f / sqrtf(1.0f + S.ls * S.ls)); lstart = 128.0f + dx * (xl - 0.5f); lend = -dx; }
    if (S.rmask) { const float dx = (1.0f * 256.0f) * (1.0f / sqrtf(1.0f + S.rs * S.rs)); rstart = 128.0f + dx * (xr - 0.5f); rend = -dx; }
    int cs = la0;
    if (runs) { const int k = wr_find_run(runs, x); if (k >= 0) cs = wr_run_s(runs, k); }
    const int n = x - cs, lane = n & 3, base = cs + (n & ~3);
    const float off = float(4 * (base - la1));
    const float dl = (lstart + float(la1 + lane) * lend) + (lend / 4.0f) * off;
    const float dr = (rstart + float(la1 + lane) * rend) + (rend / 4.0f) * off;
    cov = uint32_t(int(wr_clamp(wr_min(dl, dr), 0.0f, 256.0f) * 1.0f + 0.5f)) & 0xFFFF;
    // chunks inside the opaque interior skip DO_AA altogether (aa_span's swgl_OpaqueStart / swgl_OpaqueSize, rasterize.h:545-546;
    // blend.h:433-436) -- not the same as multiplying by 256 when a main() output lane exceeds 255 (brush_blend, amount > 1)
    const int ra0 = S.rmask ? int(floorf(wr_clamp(xr - radr, S.b0, S.b1))) : int(floorf(wr_clamp(xr, S.b0, S.b1) + 0.5f));
    aa_skip = (unsigned)(base - la1) < (unsigned)wr_imax(ra0 - la1 - 3, 0);
    s0 = la0; s1 = ra1;
  }
  const bool flat = runs && runs->n < 0;      // a flattened depth row (WrTargetDesc::flat_rows): the whole span, chunk by chunk through main()
  if (runs && !flat) {        // the span the shader sees is the depth run holding x
    const int k = wr_find_run(runs, x);
    if (k >= 0) { s0 = wr_run_s(runs, k); s1 = wr_run_e(runs, k); } else runs = nullptr;
  }
  WrPrim Pl = *Pp;
  Pl.kind = (int16_t)Q.base_kind;
  const WrTexDesc& t = D->tex[Pl.tex_slot];
  const float Lu = E.lu, Lv = E.lv, Ru = E.ru, Rv = E.rv;
  if (Q.base_kind == WR_PK_SOLID_MASKED) {
    // flat colour under swgl_clipMask: the whole chunks of the span go through commit_masked_solid_span (colour x mask, then DO_AA
    // inside blend_span with the mask key overridden, swgl_ext.h:11-23), the < 4 leftover pixels through main() + blend_pixels
    // (DO_AA, then the mask: blend.h:452-460)
    // (perspective: no span shader, every chunk through main())
    const int len = s1 - s0, spanlen = (len >= 4 && !Q.pad && !flat) ? (len & ~3) : 0;
    WrWide src; src.bg = Pl.color[0]; src.ra = Pl.color[1];
    const bool in_span = x - s0 < spanlen;
    if (in_span) {
      const WrTexDesc& mt = D->tex[WR_S_CLIP_MASK];
      const uint32_t m = ((const uint8_t*)mt.ptr)[(size_t)(y - Pl.mask_off[1]) * mt.stride + (x - Pl.mask_off[0])];
      WrWide mm; mm.bg = mm.ra = m | (m << 16);
      src = wr_apply_color(mm, Pl.color);
    }
    if (Q.aa && !aa_skip) {
      const uint32_t c0 = src.bg, c1 = src.ra;
      src.bg = ((((c0 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c0 >> 16) * cov) & 0xFFFF) >> 8) << 16);
      src.ra = ((((c1 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c1 >> 16) * cov) & 0xFFFF) >> 8) << 16);
    }
    if (!in_span) src = wr_mask_src(Pl, D, x, y, src);
    return HIT | wr_blend_rgba8(Pl.blend, dstp_, src, D);
  }
  WrWide src;
  if (Q.pad) {
    // draw_perspective_spans (rasterize.h:1236-1258) + read_perspective_inputs / step_perspective_inputs (glsl-to-cxx
    // lib.rs:660-741): no span shader, every chunk runs main() with v_uv0 = (uv / w interpolated along the span) * (1 /
    // gl_FragCoord.w); both are init_interp lanes at the span start that every chunk, drawn or skipped, advances by 4 steps
    const float wl = E.wl, wr = E.wr;
    float stepScale = 1.0f / (xr - xl);
    if (!wr_isfinite(stepScale)) stepScale = 0.0f;
    const float su = (Ru - Lu) * stepScale, sv = (Rv - Lv) * stepScale, sw = (wr - wl) * stepScale;
    const float start = (float(s0) + 0.5f) - xl;
    const int k = x - s0;
    // the (uv / w, 1 / w) interpolants of lane kk of the span: init_interp lane, then a step of 4 per chunk
    auto lane_at = [&](int kk, float& pu_, float& pv_, float& fw_) {
      pu_ = Lu + su * start; pv_ = Lv + sv * start; fw_ = wl + sw * start;
      for (int i = 0; i < (kk & 3); i++) { pu_ = pu_ + su; pv_ = pv_ + sv; fw_ = fw_ + sw; }
      pu_ = wr_accum(pu_, (su * 4.0f) * 1.0f, kk >> 2); pv_ = wr_accum(pv_, (sv * 4.0f) * 1.0f, kk >> 2);
      fw_ = wr_accum(fw_, sw * 4.0f, kk >> 2);
    };
    if (Q.base_kind == WR_PK_QUAD_MASK) {
      // ps_quad_mask: vClipLocalPos = (xy, 0, 1) / w interpolated, times w per lane; clip_local_pos = .xy / .w of the pixel and of
      // lanes 0 / 1 of its chunk (fwidth).  The varying's w is 1 at the vertices (checked by the setup stage), so its
      // interpolant is the edges' 1 / w itself.
      float px_[3], py_[3];
      const int ks[3] = {k & ~3, (k & ~3) + 1, k};
      for (int j = 0; j < 3; j++) {
        float a_, b_, c_;
        lane_at(ks[j], a_, b_, c_);
        const float wq_ = 1.0f / c_;
        const float vx = a_ * wq_, vy = b_ * wq_, vw = c_ * wq_;
        px_[j] = vx / vw; py_[j] = vy / vw;
      }
      src = wr_quad_mask_eval(Q.clip, px_[0], py_[0], px_[1], py_[1], px_[2], py_[2]);
    } else if (Q.base_kind == WR_PK_MIX_BLEND) {
      // brush_mix_blend: v_backdrop_uv as it is, v_src_uv * mix(gl_FragCoord.w, 1.0, v_perspective.x) -- both varyings interpolated / w
      // along the span (the second one's edge values on this row: sums of its own, WrPerspRec::l2u ..), each clamped to its bounds
      float pu, pv, fw;
      lane_at(k, pu, pv, fw);
      const float wq = 1.0f / fw;
      float bu = pu * wq, bv = pv * wq;
      const float L2u = wr_accum(Q.persp.l2u[si], Q.persp.l2us[si], y - S.lrow), L2v = wr_accum(Q.persp.l2v[si], Q.persp.l2vs[si], y - S.lrow);
      const float R2u = wr_accum(Q.persp.r2u[si], Q.persp.r2us[si], y - S.rrow), R2v = wr_accum(Q.persp.r2v[si], Q.persp.r2vs[si], y - S.rrow);
      const float su2 = (R2u - L2u) * stepScale, sv2 = (R2v - L2v) * stepScale;
      float qu = L2u + su2 * start, qv = L2v + sv2 * start;
      for (int i = 0; i < (k & 3); i++) { qu = qu + su2; qv = qv + sv2; }
      qu = wr_accum(qu, (su2 * 4.0f) * 1.0f, k >> 2); qv = wr_accum(qv, (sv2 * 4.0f) * 1.0f, k >> 2);
      const float pd = (1.0f - fw) * Q.persp.div + fw;
      float su_ = (qu * wq) * pd, sv_ = (qv * wq) * pd;
      bu = wr_clamp(bu, Pl.uv_bounds[0], Pl.uv_bounds[2]); bv = wr_clamp(bv, Pl.uv_bounds[1], Pl.uv_bounds[3]);
      su_ = wr_clamp(su_, Q.mix.s_bounds[0], Q.mix.s_bounds[2]); sv_ = wr_clamp(sv_, Q.mix.s_bounds[1], Q.mix.s_bounds[3]);
      src = wr_mix_blend_main(&Q.mix, D, bu, bv, su_, sv_);
    } else {
    float pu, pv, fw;
    lane_at(k, pu, pv, fw);
    const float wq = 1.0f / fw;
    float cu = pu * wq, cv = pv * wq;
    if (Q.persp.div >= 0.0f) {       // brush_image: v_uv * mix(gl_FragCoord.w, 1.0, perspective_interpolate)
      const float pd = (1.0f - fw) * Q.persp.div + fw;
      cu = cu * pd; cv = cv * pd;
    }
    cu = cu + Pl.uv_add[0]; cv = cv + Pl.uv_add[1];
    if (Pl.flags & WR_PF_TAIL_CLAMP) { cu = wr_clamp(cu, Pl.uv_bounds[0], Pl.uv_bounds[2]); cv = wr_clamp(cv, Pl.uv_bounds[1], Pl.uv_bounds[3]); }
    // the program's main() with its varying at this pixel
    if (Q.base_kind == WR_PK_TEX_REPEAT && Pl.dual && Pl.blend == WR_BLEND_DUAL_SRC) {
      // (the dual-source REPETITION key under a projective transform: the repeated uv of wr_repeat_dual_pixel from the
      // perspective-correct v_uv, both colours and the coverage into the blend)
      const WrRepeatRec& R = Q.rep;
      const float usx = R.uv_repeat[2] - R.uv_repeat[0], usy = R.uv_repeat[3] - R.uv_repeat[1];
      const float du = wr_max(cu, 0.0f), dv = wr_max(cv, 0.0f);
      float ru = (du - floorf(du)) * usx + R.uv_repeat[0], rv = (dv - floorf(dv)) * usy + R.uv_repeat[1];
      if (du >= R.tile_repeat[0]) ru = R.uv_repeat[2];
      if (dv >= R.tile_repeat[1]) rv = R.uv_repeat[3];
      ru = wr_clamp(ru, Pl.uv_bounds[0], Pl.uv_bounds[2]); rv = wr_clamp(rv, Pl.uv_bounds[1], Pl.uv_bounds[3]);
      float tx[4];
      wr_texture_rgba_f(t, ru, rv, tx);
      return HIT | wr_dual_blend(Pl, D, x, y, dstp_, tx, cov, Q.aa && !aa_skip);
    }
    if (Q.base_kind == WR_PK_TEX_REPEAT) src = wr_repeat_main(Pl, Q.rep, t, cu, cv);
    else if (Q.base_kind == WR_PK_FILTER) src = wr_filter_eval(&Pl, &Q.filt, D, cu, cv);
    else if (Q.base_kind == WR_PK_GRADIENT) src = wr_gradient_main(&Q.grad, D, cu, cv);
    else src = wr_tex_tail_texel(Pl, t, cu, cv);
    }
  } else if (Q.base_kind == WR_PK_MIX_BLEND) {
    // brush_mix_blend on a rotated / skewed (or anti-aliased) quad: this row as a one-row axis-aligned prim, both varyings -- the
    // backdrop's uv from the walk's uv edges, the source's from its z / w slots
    Pl.uvL0[0] = Lu; Pl.uvL0[1] = Lv; Pl.uvR0[0] = Ru; Pl.uvR0[1] = Rv;
    Pl.uvLs[0] = Pl.uvLs[1] = Pl.uvRs[0] = Pl.uvRs[1] = 0.0f;
    Pl.xl = xl; Pl.xr = xr; Pl.x0 = s0; Pl.x1 = s1; Pl.y0 = y; Pl.y1 = y + 1; Pl.rows_linear = 1;
    WrMixRec M2 = Q.mix;
    M2.sL0[0] = E.zl; M2.sL0[1] = E.wl; M2.sR0[0] = E.zr; M2.sR0[1] = E.wr;
    M2.sLs[0] = M2.sLs[1] = M2.sRs[0] = M2.sRs[1] = 0.0f;
    src = wr_mix_blend_pixel(&Pl, &M2, D, x, y, runs);
  } else if (Q.base_kind == WR_PK_GRADIENT || Q.base_kind == WR_PK_FILTER || Q.base_kind == WR_PK_QUAD_MASK) {
    // shader replays that take their interpolants from the prim: hand them this row as a one-row axis-aligned prim (the span
    // [s0, s1), the edges' x and interpolants on this row, no row stepping left to do)
    Pl.uvL0[0] = Lu; Pl.uvL0[1] = Lv; Pl.uvR0[0] = Ru; Pl.uvR0[1] = Rv;
    Pl.uvLs[0] = Pl.uvLs[1] = Pl.uvRs[0] = Pl.uvRs[1] = 0.0f;
    Pl.xl = xl; Pl.xr = xr; Pl.x0 = s0; Pl.x1 = s1; Pl.y0 = y; Pl.y1 = y + 1; Pl.rows_linear = 1;
    if (Q.base_kind == WR_PK_GRADIENT) src = wr_gradient_row4(&Pl, &Q.grad, D, x, y, runs).v[0];
    else if (Q.base_kind == WR_PK_FILTER) src = wr_filter_pixel(&Pl, &Q.filt, D, x, y, runs);
    else src = wr_quad_mask_pixel(&Pl, &Q.clip, D, x, y, runs);
  } else if (Q.base_kind == WR_PK_TEX_REPEAT && Pl.dual && Pl.blend == WR_BLEND_DUAL_SRC) {
    // the dual-source REPETITION key on an anti-aliased (or rotated) prim: no span shader under this key, every pixel runs main() --
    // the repeated uv of wr_repeat_dual_pixel from this row's edge interpolants -- and the blend takes both colours and the coverage
    const WrTexRow r = wr_tex_row_span(Pl, t, Lu, Lv, Ru, Rv, xl, xr, s0, s1 - s0, runs, x, true);
    const WrRepeatRec& R = Q.rep;
    const int n = x - r.x0, lane = n & 3, m = n >> 2;
    float lu = wr_pick4(r.lu, lane), lv = wr_pick4(r.lv, lane);
    lu = wr_accum(lu, (r.su * 4.0f) * 1.0f, m); lv = wr_accum(lv, (r.sv * 4.0f) * 1.0f, m);
    const float usx = R.uv_repeat[2] - R.uv_repeat[0], usy = R.uv_repeat[3] - R.uv_repeat[1];
    const float cu = wr_max(lu, 0.0f), cv = wr_max(lv, 0.0f);
    float ru = (cu - floorf(cu)) * usx + R.uv_repeat[0], rv = (cv - floorf(cv)) * usy + R.uv_repeat[1];
    if (cu >= R.tile_repeat[0]) ru = R.uv_repeat[2];
    if (cv >= R.tile_repeat[1]) rv = R.uv_repeat[3];
    ru = wr_clamp(ru, Pl.uv_bounds[0], Pl.uv_bounds[2]); rv = wr_clamp(rv, Pl.uv_bounds[1], Pl.uv_bounds[3]);
    float tx[4];
    wr_texture_rgba_f(t, ru, rv, tx);
    return HIT | wr_dual_blend(Pl, D, x, y, dstp_, tx, cov, Q.aa && !aa_skip);
  } else {
    const WrTexRow r = wr_tex_row_span(Pl, t, Lu, Lv, Ru, Rv, xl, xr, s0, s1 - s0, runs, x, Q.base_kind == WR_PK_TEX_REPEAT && Q.rep.no_span != 0);
    src = Q.base_kind == WR_PK_TEX_REPEAT ? wr_repeat_pixel_row(Pl, Q.rep, t, r, x - r.x0) : wr_tex_pixel_row(Pl, t, r, x - r.x0);
  }
  if (Q.aa && !aa_skip) {
    const uint32_t c0 = src.bg, c1 = src.ra;
    src.bg = ((((c0 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c0 >> 16) * cov) & 0xFFFF) >> 8) << 16);
    src.ra = ((((c1 & 0xFFFF) * cov) & 0xFFFF) >> 8) | (((((c1 >> 16) * cov) & 0xFFFF) >> 8) << 16);
  }
  src = wr_mask_src(Pl, D, x, y, src);
  return HIT | wr_blend_rgba8(Pl.blend, dstp_, src, D, Pl.color);
}

// red channel of a textured prim's fragment value (R8 targets)
__device__ __noinline__ uint32_t wr_tex_pixel_r(const WrPrim* Pp, const WrDrawDesc* D, int x, int y) {
  return wr_tex_pixel(*Pp, D->tex[Pp->tex_slot], x, y).ra & 0xFFFF;
}

// ---------------------------------------------------------------------------
// brush_linear_gradient: the 4 pixels x .. x+3 of row y (WideRGBA8 sources, b,g | r,a as u16 pairs).
//   span part   commitLinearGradient (swgl_ext.h:1390-1578) walks the row from the span start:
//               [run of whole chunks inside one merged table range, colour stepped in 0..0xFF00
//               fixed point] [one per-sample table lookup chunk] ...  A pixel's value depends on
//               the run it falls in, so the walk is replayed up to the chunk(s) holding x .. x+3.
//   the rest    fragment shader: sample_gradient(dot(fract(v_pos), v_scale_dir) - v_start_offset)
//               (brush_linear_gradient.glsl:66-83, gradient.glsl:42-61), also for every pixel when
//               the table fails swgl_validateGradient or the per-chunk delta is not finite.
WR_DEVICE bool wr_stops_merge(const float* stops, int a, int b) {   // GradientStops::can_merge
  const float* sa = stops + 8 * a + 4; const float* sb = stops + 8 * b + 4;
  return sa[0] == sb[0] && sa[1] == sb[1] && sa[2] == sb[2] && sa[3] == sb[3];
}
WR_DEVICE float wr_fract(float v) { return v - floorf(v); }
WR_DEVICE uint32_t wr_u16_round1(float v) { return uint32_t(int(v * 1.0f + 0.5f)) & 0xFFFF; }   // CONVERT(round_pixel(v, 1), U16)

// sampleGradient, one lane (swgl_ext.h:1349-1373): zyxw swizzle, round_pixel, packRGBA8 (wrapping)
WR_DEVICE WrWide wr_sample_gradient(const float* stops, float entry) {
  const int index = int(entry);
  const float offset = entry - float(index);
  const float* st = stops + 8 * index;
  wf4 c = {st[0] + st[4] * offset, st[1] + st[5] * offset, st[2] + st[6] * offset, st[3] + st[7] * offset};
  uint32_t pc[2];
  wr_pack_color(c, pc);
  WrWide w; w.bg = pc[0]; w.ra = pc[1];
  return w;
}

// `runs`: the row's depth runs; only pixel x is evaluated then (out.v[0]), inside the run that holds it.
__device__ __noinline__ WrGrad4 wr_gradient_row4(const WrPrim* Pp, const WrGradRec* Gp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrGradRec& G = *Gp;
  WrGrad4 out;
#pragma unroll
  for (int i = 0; i < 4; i++) { out.v[i].bg = 0; out.v[i].ra = 0; }
  // interpolants of v_pos at the span start (rasterize.h:1003-1017), as wr_tex_row
  const int k = y - P.y0;
  const bool lin = P.rows_linear != 0;
  const float Lu = wr_row_interp(P.uvL0[0], P.uvLs[0], k, lin), Lv = wr_row_interp(P.uvL0[1], P.uvLs[1], k, lin);
  const float Ru = wr_row_interp(P.uvR0[0], P.uvRs[0], k, lin), Rv = wr_row_interp(P.uvR0[1], P.uvRs[1], k, lin);
  float stepScale = 1.0f / (P.xr - P.xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  const float su = (Ru - Lu) * stepScale, sv = (Rv - Lv) * stepScale;
  const int kr = runs ? wr_find_run(runs, x) : -1;
  const int X0 = kr >= 0 ? wr_run_s(runs, kr) : P.x0;                    // start of the sub-span the span shader sees
  const int len = kr >= 0 ? wr_run_e(runs, kr) - X0 : P.x1 - P.x0;
  const float start = float(kr >= 0 ? wr_run_s(runs, 0) : P.x0) + 0.5f - P.xl;
  const float sdx = G.scale_dir[0], sdy = G.scale_dir[1];
  int span = len >= 4 ? (len & ~3) : 0;
  // init_interp (glsl.h:3084-3089)
  float px[4], py[4];
  px[0] = Lu + su * start; py[0] = Lv + sv * start;
#pragma unroll
  for (int i = 1; i < 4; i++) { px[i] = px[i - 1] + su; py[i] = py[i - 1] + sv; }
  if (kr > 0) {
    const bool spans = G.stops && wr_isfinite(((px[1] - px[0]) * 4.0f) * sdx + ((py[1] - py[0]) * 4.0f) * sdy);
    wr_run_lanes(runs, kr, Lu, su, P.xl, spans, px);
    wr_run_lanes(runs, kr, Lv, sv, P.xl, spans, py);
  }
  const float ou = px[0], ov = py[0];
  const float lu1 = px[1], lu2 = px[2], lu3 = px[3], lv1 = py[1], lv2 = py[2], lv3 = py[3];   // the lanes at the sub-span start
  const float psx = (px[1] - px[0]) * 4.0f, psy = (py[1] - py[0]) * 4.0f;   // dFdx(pos) * 4
  const float delta = psx * sdx + psy * sdy;
  if (!G.stops || G.radial >= 2 || (!G.radial && !wr_isfinite(delta))) span = 0;
  if (runs && runs->n < 0) span = 0;        // a flattened depth row: every chunk through main()
  const int n_lo = wr_imax(x - X0, 0), n_hi = wr_imin(x + (kr >= 0 ? 0 : 3) - X0, len - 1);
  if (n_hi < n_lo) return out;
  const float size = 128.0f;
  if (G.radial == 1) {
    // commitRadialGradient (swgl_ext.h:1629-1835): the row is walked from the span start -- runs of whole chunks inside one
    // merged table range (colour = colorF + deltaColorF * length(pos) per pixel), per-sample table chunks in between -- with
    // dot(pos, pos) accumulated chunk by chunk; replayed up to the chunk(s) holding x .. x+3
    const float radius = G.start_offset;
    float ddx = px[1] - px[0], ddy = py[1] - py[0];
    float dd = ddx * ddx + ddy * ddy;
    if (!wr_isfinite(dd) || !wr_isfinite(radius)) span = 0;
    if (n_lo < span) {
      const float* stops = G.stops;
      const int c_hi = wr_imin(n_hi, span - 1) >> 2, c_lo = n_lo >> 2;
      const float fspan = float(span);
      float invDelta, middleT, middleB;
      if (dd > 0.0f) {
        invDelta = 1.0f / dd;
        middleT = -(ddx * px[0] + ddy * py[0]) * invDelta;
        middleB = middleT * middleT - (px[0] * px[0] + py[0] * py[0]) * invDelta;
      } else { invDelta = 0.0f; middleT = fspan; middleB = 0.0f; }
      const float mx_ = px[0] + ddx * middleT, my_ = py[0] + ddy * middleT, ex_ = px[0] + ddx * fspan, ey_ = py[0] + ddy * fspan;
      const float mer_x = sqrtf(wr_max(mx_ * mx_ + my_ * my_, 1.0e-12f)), mer_y = sqrtf(wr_max(ex_ * ex_ + ey_ * ey_, 1.0e-12f));
      const float middleRadius = fspan < middleT ? mer_y : mer_x, endRadius = mer_y;
      ddx *= 4.0f; ddy *= 4.0f; dd *= 16.0f;
      float dotPos[4], dotPosDelta[4], off[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { dotPos[i] = px[i] * px[i] + py[i] * py[i]; dotPosDelta[i] = 2.0f * (px[i] * ddx + py[i] * ddy) + dd; }
      const float dd2 = 2.0f * dd;
      for (int t = 0; t < span && (t >> 2) <= c_hi;) {
#pragma unroll
        for (int i = 0; i < 4; i++) off[i] = sqrtf(wr_max(dotPos[i], 1.0e-12f)) - radius;
        float startRadius = radius;
        if (G.repeat != 0.0f) {
          startRadius += off[0];
#pragma unroll
          for (int i = 0; i < 4; i++) off[i] = wr_fract(off[i]);
          startRadius -= off[0];
        }
        float intercept = -1.0f;
        int minIndex = 0, maxIndex = int(1.0f + size);
        const bool past = float(t) >= middleT;
        if (off[0] < 0.0f) {
          maxIndex = minIndex;
          if (past) intercept = radius;
        } else if (off[0] < 1.0f) {
          minIndex = int(1.0f + off[0] * size);
          maxIndex = minIndex;
          const float searchOffset = (past ? endRadius : middleRadius) - startRadius;
          const int searchIndex = int(wr_clamp(1.0f + size * searchOffset, 1.0f, size));
          if (past) {
            // (while (maxIndex + 1 <= searchIndex && can_merge(maxIndex, maxIndex + 1)) maxIndex++, by the prim's merge bitmap)
            if (maxIndex + 1 <= searchIndex) maxIndex = wr_imin(wr_merge_clear_from(G, maxIndex), searchIndex);
            intercept = float(maxIndex + 1);
          } else {
            if (minIndex - 1 >= searchIndex) minIndex = wr_imax(wr_merge_clear_below(G, minIndex) + 1, searchIndex);
            intercept = float(minIndex);
          }
          intercept = wr_clamp((intercept - 1.0f) / size, 0.0f, 1.0f) + startRadius;
        } else {
          minIndex = maxIndex;
          if (!past) intercept = radius + 1.0f;
        }
        float endT = past ? fspan : float(wr_imin(span, int(middleT)));
        if (intercept >= 0.0f) {
          float b = middleB + intercept * intercept * invDelta;
          if (b > 0.0f) { b = sqrtf(b); endT = wr_min(endT, past ? middleT + b : middleT - b); }
          else endT = wr_min(endT, middleT);
        }
        if (float(t) + 4.0f <= endT) {
          const int inside = int(endT - float(t)) & ~3;
          const float* s0 = stops + 8 * minIndex; const float* s1 = stops + 8 * maxIndex;
          const float mn[4] = {s0[2] * 255.0f, s0[1] * 255.0f, s0[0] * 255.0f, s0[3] * 255.0f};
          const float mxc[4] = {(s1[2] + s1[6]) * 255.0f, (s1[1] + s1[5]) * 255.0f, (s1[0] + s1[4]) * 255.0f, (s1[3] + s1[7]) * 255.0f};
          const float k = size / float(maxIndex + 1 - minIndex);
          const float base = startRadius + float(minIndex - 1) / size;
          float dcol[4], col[4];
#pragma unroll
          for (int q = 0; q < 4; q++) { dcol[q] = (mxc[q] - mn[q]) * k; col[q] = mn[q] - dcol[q] * base; }
          for (int c = 0; c < inside; c += 4) {
            const int chunk = (t + c) >> 2;
            if (chunk >= c_lo && chunk <= c_hi) {
#pragma unroll
              for (int i = 0; i < 4; i++) {
                const int o = X0 + t + c + i - x;
                if (o >= 0 && o < 4) {
                  const float og = sqrtf(dotPos[i]);
                  const uint32_t cb = wr_u16_round1(col[0] + dcol[0] * og), cg = wr_u16_round1(col[1] + dcol[1] * og);
                  const uint32_t cr = wr_u16_round1(col[2] + dcol[2] * og), ca = wr_u16_round1(col[3] + dcol[3] * og);
                  out.v[o].bg = cb | (cg << 16); out.v[o].ra = cr | (ca << 16);
                }
              }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { dotPos[i] += dotPosDelta[i]; dotPosDelta[i] += dd2; }
          }
          t += inside;
          if (t >= span) break;
#pragma unroll
          for (int i = 0; i < 4; i++) {
            off[i] = sqrtf(wr_max(dotPos[i], 1.0e-12f)) - radius;
            if (G.repeat != 0.0f) off[i] = wr_fract(off[i]);
          }
        }
        const int chunk = t >> 2;
        if (chunk >= c_lo && chunk <= c_hi) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int o = X0 + t + i - x;
            if (o >= 0 && o < 4) out.v[o] = wr_sample_gradient(stops, wr_clamp(off[i] * size + 1.0f, 0.0f, 1.0f + size));
          }
        }
        t += 4;
#pragma unroll
        for (int i = 0; i < 4; i++) { dotPos[i] += dotPosDelta[i]; dotPosDelta[i] += dd2; }
      }
    }
  } else if (n_lo < span) {
    const float* stops = G.stops;
    const int c_lo = n_lo >> 2, c_hi = wr_imin(n_hi, span - 1) >> 2;     // chunks wanted
    float dcxx = 0.25f * float(span), dcxy = 0.0f, dcyx = dcxx, dcyy = 0.0f;
    const bool tile = G.no_tile == 0;      // tileRepeat (swgl_ext.h:1411-1432)
    if (tile && psx != 0.0f) { const float r = 1.0f / psx; dcxx = (psx >= 0.0f ? 1.0f : 0.0f) * r; dcxy = 1.0f * r; }
    if (tile && psy != 0.0f) { const float r = 1.0f / psy; dcyx = (psy >= 0.0f ? 1.0f : 0.0f) * r; dcyy = 1.0f * r; }
    int left = span, chunk = 0;          // chunk: index of the next chunk to be produced
    while (left > 0 && chunk <= c_hi) {
      float chunks = 0.25f * float(left);
      float rx[4], ry[4], off[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { rx[i] = tile ? wr_fract(px[i]) : px[i]; ry[i] = tile ? wr_fract(py[i]) : py[i]; }
      if (tile) {
        chunks = wr_min(chunks, dcxx - rx[0] * dcxy);
        chunks = wr_min(chunks, dcyx - ry[0] * dcyy);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        off[i] = rx[i] * sdx + ry[i] * sdy - G.start_offset;
        if (G.repeat != 0.0f) off[i] = wr_fract(off[i]);
      }
      float startEntry;
      int minIndex, maxIndex;
      if (off[0] < 0.0f) {
        startEntry = 0.0f; minIndex = maxIndex = 0;
        if (delta > 0.0f) chunks = wr_min(chunks, -off[0] / delta);
      } else if (off[0] < 1.0f) {
        startEntry = 1.0f + off[0] * size;
        if (delta < 0.0f) chunks = wr_min(chunks, -off[0] / delta);
        else if (delta > 0.0f) chunks = wr_min(chunks, (1.0f - off[0]) / delta);
        const float endEntry = wr_clamp(1.0f + (off[0] + delta * float(int(chunks))) * size, 0.0f, 1.0f + size);
        minIndex = maxIndex = int(startEntry);
        if (delta > 0.0f) {
          // (while (float(maxIndex + 1) < endEntry && can_merge(maxIndex, maxIndex + 1)) maxIndex++, by the prim's merge bitmap: the walk
          // ends at the first entry whose bit is clear or at the first j with j + 1 >= endEntry, i.e. j = ceil(endEntry) - 1)
          if (float(maxIndex + 1) < endEntry) maxIndex = wr_imin(wr_merge_clear_from(G, maxIndex), wr_imax(maxIndex, int(ceilf(endEntry)) - 1));
          chunks = wr_min(chunks, (float(maxIndex + 1) - startEntry) / (delta * size));
        } else if (delta < 0.0f) {
          // (while (float(minIndex - 1) > endEntry && can_merge(minIndex - 1, minIndex)) minIndex--: down to the entry above the last
          // clear bit below, or to the first j with j - 1 <= endEntry, i.e. j = floor(endEntry) + 1)
          if (float(minIndex - 1) > endEntry) minIndex = wr_imax(wr_merge_clear_below(G, minIndex) + 1, wr_imin(minIndex, int(floorf(endEntry)) + 1));
          chunks = wr_min(chunks, (float(minIndex) - startEntry) / (delta * size));
        }
      } else {
        startEntry = 1.0f + size; minIndex = maxIndex = int(startEntry);
        if (delta < 0.0f) chunks = wr_min(chunks, (1.0f - off[0]) / delta);
      }
      if (chunks >= 1.0f) {
        const int inside = int(chunks);
        if (chunk + inside > c_lo) {
          // colours of the merged range in 0..0xFF00, BGRA order
          const float* s0 = stops + 8 * minIndex; const float* s1 = stops + 8 * maxIndex;
          const float mn[4] = {s0[2] * 65280.0f, s0[1] * 65280.0f, s0[0] * 65280.0f, s0[3] * 65280.0f};
          const float mx[4] = {(s1[2] + s1[6]) * 65280.0f, (s1[1] + s1[5]) * 65280.0f, (s1[0] + s1[4]) * 65280.0f,
                               (s1[3] + s1[7]) * 65280.0f};
          const float inv = 1.0f / float(maxIndex + 1 - minIndex);
          const float se = startEntry - float(minIndex), ds = delta * size;
          for (int c = wr_imax(c_lo, chunk); c <= c_hi && c < chunk + inside; c++) {
            const int kk = c - chunk, seg = kk >> 6, r = kk & 63;
            uint32_t ch[4][4];           // [pixel][channel]
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const float range = (mx[q] - mn[q]) * inv;
              float cf = mn[q] + range * se + 128.0f;
              const float dcf = range * ds;
              const uint32_t dc = wr_u16_round1(dcf);
              for (int t = 0; t < seg; t++) cf += dcf * 64.0f;
              ch[0][q] = ((wr_u16_round1(cf) + uint32_t(r) * dc) & 0xFFFF) >> 8;
              ch[1][q] = ((wr_u16_round1(cf + dcf * 0.25f) + uint32_t(r) * dc) & 0xFFFF) >> 8;
              ch[2][q] = ((wr_u16_round1(cf + dcf * 0.5f) + uint32_t(r) * dc) & 0xFFFF) >> 8;
              ch[3][q] = ((wr_u16_round1(cf + dcf * 0.75f) + uint32_t(r) * dc) & 0xFFFF) >> 8;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const int o = X0 + 4 * c + i - x;
              if (o >= 0 && o < 4) { out.v[o].bg = ch[i][0] | (ch[i][1] << 16); out.v[o].ra = ch[i][2] | (ch[i][3] << 16); }
            }
          }
        }
        chunk += inside;
        left -= inside * 4;
        if (left <= 0) break;
        const float fi = float(inside);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          px[i] += psx * fi; py[i] += psy * fi;
          off[i] = (tile ? wr_fract(px[i]) : px[i]) * sdx + (tile ? wr_fract(py[i]) : py[i]) * sdy - G.start_offset;
          if (G.repeat != 0.0f) off[i] = wr_fract(off[i]);
        }
      }
      if (chunk >= c_lo && chunk <= c_hi) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int o = X0 + 4 * chunk + i - x;
          if (o >= 0 && o < 4) out.v[o] = wr_sample_gradient(stops, wr_clamp(off[i] * size + 1.0f, 0.0f, 1.0f + size));
        }
      }
      chunk++;
      left -= 4;
#pragma unroll
      for (int i = 0; i < 4; i++) { px[i] += psx; py[i] += psy; }
    }
  }
  if (n_hi >= span) {
    // main() pixels: init_interp lane, step_interp_inputs(drawn) once, then one step per chunk run
    const WrTexDesc& gb = D->tex[WR_S_GPU_BUFFER_F];
    for (int n = wr_imax(n_lo, span); n <= n_hi; n++) {
      const int lane = (n - span) & 3, m = (n - span) >> 2;
      float lu = lane == 0 ? ou : (lane == 1 ? lu1 : (lane == 2 ? lu2 : lu3));
      float lv = lane == 0 ? ov : (lane == 1 ? lv1 : (lane == 2 ? lv2 : lv3));
      if (span > 0) {
        const float chunks = float(span) * 0.25f;
        lu = lu + (su * 4.0f) * chunks; lv = lv + (sv * 4.0f) * chunks;
      }
      lu = wr_accum(lu, (su * 4.0f) * 1.0f, m); lv = wr_accum(lv, (sv * 4.0f) * 1.0f, m);
      float offset = (G.no_tile ? lu : wr_fract(lu)) * sdx + (G.no_tile ? lv : wr_fract(lv)) * sdy - G.start_offset;
      if (G.radial == 1) offset = sqrtf(lu * lu + lv * lv) - G.start_offset;              // length(v_pos) - v_start_radius.x
      if (G.radial == 3) {                                                                // ps_quad_conic_gradient.glsl:60-81 (approx_atan2)
        const float ax_ = fabsf(lu), ay_ = fabsf(lv);
        const float slope = wr_min(ax_, ay_) / wr_max(ax_, ay_);
        const float s2 = slope * slope;
        float r = ((-0.0464964749f * s2 + 0.15931422f) * s2 - 0.327622764f) * s2 * slope + slope;
        r = ay_ > ax_ ? 1.57079637f - r : r;
        r = lu < 0.0f ? 3.14159274f - r : r;
        r = r * copysignf(1.0f, lv);
        offset = wr_fract((r + G.conic_angle) / (2.0f * 3.141592653589793f)) * G.conic_scale - G.start_offset;
      }
      if (G.radial == 2) {                                                                // cs_conic_gradient.glsl:52-65
        const float cur = atan2f(lv - G.scale_dir[1], lu - G.scale_dir[0]) + G.conic_angle;
        offset = wr_fract(cur / (2.0f * 3.141592653589793f)) * G.conic_scale - G.start_offset;
      }
      offset -= floorf(offset) * G.repeat;
      const float xe = wr_clamp(1.0f + offset * 128.0f, 0.0f, 1.0f + 128.0f);
      const float ei = floorf(xe), ef = xe - ei;
      const int addr = G.address + 2 * int(ei);
      wf4 t0, t1;
      if (G.table) { const float* te = G.table + 8 * int(ei); t0 = wf4{te[0], te[1], te[2], te[3]}; t1 = wf4{te[4], te[5], te[6], te[7]}; }
      else {
        t0 = wr_fetch_f(gb, int(unsigned(addr) % 1024u), int(unsigned(addr) / 1024u));
        t1 = wr_fetch_f(gb, int(unsigned(addr) % 1024u) + 1, int(unsigned(addr) / 1024u));
      }
      uint32_t pc[2];
      wr_pack_color(wf4{t0.x + t1.x * ef, t0.y + t1.y * ef, t0.z + t1.z * ef, t0.w + t1.w * ef}, pc);
      out.v[X0 + n - x].bg = pc[0]; out.v[X0 + n - x].ra = pc[1];
    }
  }
  return out;
}
// main() of the gradient programs for one pixel whose v_pos is (lu, lv) (brush_linear_gradient.glsl:66-83, gradient.glsl:42-61; the
// radial / conic quad patterns): what the loop above evaluates for the pixels the span shader leaves -- and what every pixel of
// a prim under a perspective transform runs, with the perspective-correct v_pos
__device__ __noinline__ WrWide wr_gradient_main(const WrGradRec* Gp, const WrDrawDesc* D, float lu, float lv) {
  const WrGradRec& G = *Gp;
  const WrTexDesc& gb = D->tex[WR_S_GPU_BUFFER_F];
  const float sdx = G.scale_dir[0], sdy = G.scale_dir[1];
  float offset = (G.no_tile ? lu : wr_fract(lu)) * sdx + (G.no_tile ? lv : wr_fract(lv)) * sdy - G.start_offset;
  if (G.radial == 1) offset = sqrtf(lu * lu + lv * lv) - G.start_offset;
  if (G.radial == 3) {
    const float ax_ = fabsf(lu), ay_ = fabsf(lv);
    const float slope = wr_min(ax_, ay_) / wr_max(ax_, ay_);
    const float s2 = slope * slope;
    float r = ((-0.0464964749f * s2 + 0.15931422f) * s2 - 0.327622764f) * s2 * slope + slope;
    r = ay_ > ax_ ? 1.57079637f - r : r;
    r = lu < 0.0f ? 3.14159274f - r : r;
    r = r * copysignf(1.0f, lv);
    offset = wr_fract((r + G.conic_angle) / (2.0f * 3.141592653589793f)) * G.conic_scale - G.start_offset;
  }
  if (G.radial == 2) {
    const float cur = atan2f(lv - G.scale_dir[1], lu - G.scale_dir[0]) + G.conic_angle;
    offset = wr_fract(cur / (2.0f * 3.141592653589793f)) * G.conic_scale - G.start_offset;
  }
  offset -= floorf(offset) * G.repeat;
  const float xe = wr_clamp(1.0f + offset * 128.0f, 0.0f, 1.0f + 128.0f);
  const float ei = floorf(xe), ef = xe - ei;
  const int addr = G.address + 2 * int(ei);
  wf4 t0, t1;
  if (G.table) { const float* te = G.table + 8 * int(ei); t0 = wf4{te[0], te[1], te[2], te[3]}; t1 = wf4{te[4], te[5], te[6], te[7]}; }
  else {
    t0 = wr_fetch_f(gb, int(unsigned(addr) % 1024u), int(unsigned(addr) / 1024u));
    t1 = wr_fetch_f(gb, int(unsigned(addr) % 1024u) + 1, int(unsigned(addr) / 1024u));
  }
  uint32_t pc[2];
  wr_pack_color(wf4{t0.x + t1.x * ef, t0.y + t1.y * ef, t0.z + t1.z * ef, t0.w + t1.w * ef}, pc);
  WrWide w; w.bg = pc[0]; w.ra = pc[1];
  return w;
}

// ---------------------------------------------------------------------------
// cs_blur, one destination pixel.  Returns the unpacked source value(s) that go
// to the blend stage: 4 x u16 (RGBA8 targets) or the value in .bg's low half (R8).
//   span part   blendGaussianBlur (swgl_ext.h:951-978): 4-pixel chunks from the
//               integer texel position of the span start, while they fit in
//               min(bounds.z, start + span, width); 8.8 fixed-point taps with
//               saturating adds (texture.h:1165-1308)
//   the rest    the float fragment shader (cs_blur.glsl:137-181)
// ---------------------------------------------------------------------------
// brush_blend fragment shader (brush_blend.glsl:91-120, blend.glsl:90-237), one pixel.
// swgl has no span shader for it: main() runs on every 4-pixel chunk with glsl.h's float
// vectors, so the restatement is per pixel in strict fp32, same operation order.
// pow() is glsl.h's approximation (glsl.h:776-799), not libm -- restated bit for bit.
WR_DEVICE float wr_glsl_floor(float v) {            // glsl.h:687-690
  const float roundtrip = float(int(v));
  return roundtrip - (roundtrip > v ? 1.0f : 0.0f);
}
WR_DEVICE float wr_approx_log2(float x) {           // glsl.h:776-784
  uint32_t b; __builtin_memcpy(&b, &x, 4);
  const float e = float(b) * (1.0f / (1 << 23));
  const uint32_t mb = (b & 0x007fffffu) | 0x3f000000u;
  float m; __builtin_memcpy(&m, &mb, 4);
  return e - 124.225514990f - 1.498030302f * m - 1.725879990f / (0.3520887068f + m);
}
WR_DEVICE float wr_approx_pow2(float x) {           // glsl.h:786-791; roundfast = cast(v * scale + 0.5f) (portable path)
  const float f = x - wr_glsl_floor(x);
  const float t = x + 121.274057500f - 1.490129070f * f + 27.728023300f / (4.84252568f - f);
  const int32_t r = int32_t((1.0f * (1 << 23)) * t + 0.5f);
  float o; __builtin_memcpy(&o, &r, 4);
  return o;
}
WR_DEVICE float wr_glsl_pow(float x, float y) {     // glsl.h:797-799
  return (x == 0.0f || x == 1.0f) ? x : wr_approx_pow2(wr_approx_log2(x) * y);
}

__device__ __noinline__ WrWide wr_filter_eval(const WrPrim* Pp, const WrFilterRec* Fp, const WrDrawDesc* D, float cu, float cv);
__device__ __noinline__ WrWide wr_filter_pixel(const WrPrim* Pp, const WrFilterRec* Fp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrTexDesc& t = D->tex[P.tex_slot];
  // v_uv of this pixel as the 4-wide fragment loop steps it, clamped to v_uv_sample_bounds
  const WrTexRow r = wr_tex_row(P, t, y, runs, x);
  float cu, cv;
  wr_tex_tail_uv(P, r, x - r.x0, cu, cv);
  return wr_filter_eval(Pp, Fp, D, cu, cv);
}
// ... from the (clamped) uv on: texture(), CalculateFilter, the fragment colour
__device__ __noinline__ WrWide wr_filter_eval(const WrPrim* Pp, const WrFilterRec* Fp, const WrDrawDesc* D, float cu, float cv) {
  const WrPrim& P = *Pp;
  const WrFilterRec& F = *Fp;
  const WrTexDesc& t = D->tex[P.tex_slot];
  // texture(sColor0, uv): texture.h:1028-1071 (linear RGBA8, 7-bit fractions) / nearest
  const float W = t.sw, H = t.sh;
  float cr, cg, cb, ca;
  if (!t.ptr) { cr = cg = cb = ca = 0.0f; }
  else if (t.format == WR_FMT_R8) {
    float m;
    if (t.linear) m = float(wr_sample_linear_r8(t, int(cu * W * 128.0f + (0.5f - 64.0f)), int(cv * H * 128.0f + (0.5f - 64.0f)))) * (1.0f / 255.0f);
    else m = float(((const uint8_t*)t.ptr)[(size_t)wr_clamp_coord(int(cu * W), t.width) + (size_t)wr_clamp_coord(int(cv * H), t.height) * t.stride]) * (1.0f / 255.0f);
    cr = m; cg = 0.0f; cb = 0.0f; ca = 1.0f;
  } else if (t.linear) {
    const WrWide s = wr_sample_linear_rgba8(t, int(cu * W * 128.0f + (0.5f - 64.0f)), int(cv * H * 128.0f + (0.5f - 64.0f)));
    cb = float(s.bg & 0xFFFF) * (1.0f / 255.0f); cg = float(s.bg >> 16) * (1.0f / 255.0f);
    cr = float(s.ra & 0xFFFF) * (1.0f / 255.0f); ca = float(s.ra >> 16) * (1.0f / 255.0f);
  } else {
    const uint32_t p = ((const uint32_t*)t.ptr)[(size_t)wr_clamp_coord(int(cu * W), t.width) + (size_t)wr_clamp_coord(int(cv * H), t.height) * t.stride];
    cb = float(p & 0xFF) * (1.0f / 255.0f); cg = float((p >> 8) & 0xFF) * (1.0f / 255.0f);
    cr = float((p >> 16) & 0xFF) * (1.0f / 255.0f); ca = float(p >> 24) * (1.0f / 255.0f);
  }
  // CalculateFilter (blend.glsl:190-237): un-premultiply, filter
  float alpha = ca;
  float c[3] = {alpha != 0.0f ? cr / alpha : cr, alpha != 0.0f ? cg / alpha : cg, alpha != 0.0f ? cb / alpha : cb};
  const float amount = F.amount;
  switch (F.op) {
    case 0:   // FILTER_CONTRAST
      for (int i = 0; i < 3; i++) c[i] = wr_clamp(c[i] * amount - 0.5f * amount + 0.5f, 0.0f, 1.0f);
      break;
    case 3:   // FILTER_INVERT: mix(Cs, 1 - Cs, amount)
      for (int i = 0; i < 3; i++) c[i] = ((1.0f - c[i]) - c[i]) * amount + c[i];
      break;
    case 6:   // FILTER_BRIGHTNESS
      for (int i = 0; i < 3; i++) c[i] = wr_clamp(c[i] * amount, 0.0f, 1.0f);
      break;
    case 8:   // FILTER_SRGB_TO_LINEAR
      for (int i = 0; i < 3; i++) {
        const float c1 = c[i] / 12.92f, c2 = wr_glsl_pow(c[i] / 1.055f + (0.055f / 1.055f), 2.4f);
        c[i] = c[i] <= 0.04045f ? c1 : c2;
      }
      break;
    case 9:   // FILTER_LINEAR_TO_SRGB
      for (int i = 0; i < 3; i++) {
        const float c1 = c[i] * 12.92f, c2 = 1.055f * wr_glsl_pow(c[i], 1.0f / 2.4f) - 0.055f;
        c[i] = c[i] <= 0.0031308f ? c1 : c2;
      }
      break;
    case 11: {  // FILTER_COMPONENT_TRANSFER (blend.glsl:126-188)
      float ch[4] = {c[0], c[1], c[2], alpha};
      const WrTexDesc& gc = D->tex[WR_S_GPU_CACHE];
      int offset = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int fn = int(F.funcs[i]);
        if (fn == 1 || fn == 2) {           // TABLE / DISCRETE: 256-entry lookup, 4 values per block
          const int k = int(wr_glsl_floor(ch[i] * 255.0f + 0.5f));
          const unsigned a = unsigned(F.table_address + offset + k / 4);
          const wf4 tx = wr_fetch_f(gc, int(a % 1024u), int(a / 1024u));
          const int sel = k % 4;
          const float v = sel == 0 ? tx.x : (sel == 1 ? tx.y : (sel == 2 ? tx.z : (sel == 3 ? tx.w : 0.0f)));
          ch[i] = wr_clamp(v, 0.0f, 1.0f);
          offset += 64;
        } else if (fn == 3) {               // LINEAR
          const unsigned a = unsigned(F.table_address + offset);
          const wf4 tx = wr_fetch_f(gc, int(a % 1024u), int(a / 1024u));
          ch[i] = wr_clamp(tx.x * ch[i] + tx.y, 0.0f, 1.0f);
          offset += 1;
        } else if (fn == 4) {               // GAMMA
          const unsigned a = unsigned(F.table_address + offset);
          const wf4 tx = wr_fetch_f(gc, int(a % 1024u), int(a / 1024u));
          ch[i] = wr_clamp(tx.x * wr_glsl_pow(ch[i], tx.y) + tx.z, 0.0f, 1.0f);
          offset += 1;
        }
      }
      c[0] = ch[0]; c[1] = ch[1]; c[2] = ch[2]; alpha = ch[3];
      break;
    }
    case 10:  // FILTER_FLOOD
      c[0] = F.color_offset[0]; c[1] = F.color_offset[1]; c[2] = F.color_offset[2]; alpha = F.color_offset[3];
      break;
    default: {  // colour-matrix filters: color_mat * vec4(color, alpha) + color_offset (glsl.h:2582-2598 order)
      const float* m = F.color_mat;
      float o4[4];
      for (int i = 0; i < 4; i++)
        o4[i] = wr_clamp((m[i] * c[0] + m[4 + i] * c[1] + m[8 + i] * c[2] + m[12 + i] * alpha) + F.color_offset[i], 0.0f, 1.0f);
      c[0] = o4[0]; c[1] = o4[1]; c[2] = o4[2]; alpha = o4[3];
    }
  }
  // Fragment(alpha * vec4(color, 1.0)) -> pack_pixels_RGBA8
  uint32_t pc[2];
  wr_pack_color(wf4{alpha * c[0], alpha * c[1], alpha * c[2], alpha * 1.0f}, pc);
  WrWide s; s.bg = pc[0]; s.ra = pc[1];
  return s;
}

// ---------------------------------------------------------------------------
// brush_mix_blend fragment shader (brush_mix_blend.glsl:88-330), one pixel.  No span shader in swgl: main() on every chunk;
// per pixel in strict fp32, same operation order.  The GLSL's scalar branches become selects there (every lane computes
// both sides); the selected value is the branch's.
WR_DEVICE void wr_texture_rgba_f(const WrTexDesc& t, float cu, float cv, float (&c)[4]) {      // texture(sampler, uv) -> (r, g, b, a)
  const float W = t.sw, H = t.sh;
  if (!t.ptr) { c[0] = c[1] = c[2] = c[3] = 0.0f; return; }
  if (t.format == WR_FMT_R8) {
    float m;
    if (t.linear) m = float(wr_sample_linear_r8(t, int(cu * W * 128.0f + (0.5f - 64.0f)), int(cv * H * 128.0f + (0.5f - 64.0f)))) * (1.0f / 255.0f);
    else m = float(((const uint8_t*)t.ptr)[(size_t)wr_clamp_coord(int(cu * W), t.width) + (size_t)wr_clamp_coord(int(cv * H), t.height) * t.stride]) * (1.0f / 255.0f);
    c[0] = m; c[1] = 0.0f; c[2] = 0.0f; c[3] = 1.0f;
  } else if (t.linear) {
    const WrWide s = wr_sample_linear_rgba8(t, int(cu * W * 128.0f + (0.5f - 64.0f)), int(cv * H * 128.0f + (0.5f - 64.0f)));
    c[2] = float(s.bg & 0xFFFF) * (1.0f / 255.0f); c[1] = float(s.bg >> 16) * (1.0f / 255.0f);
    c[0] = float(s.ra & 0xFFFF) * (1.0f / 255.0f); c[3] = float(s.ra >> 16) * (1.0f / 255.0f);
  } else {
    const uint32_t p = ((const uint32_t*)t.ptr)[(size_t)wr_clamp_coord(int(cu * W), t.width) + (size_t)wr_clamp_coord(int(cv * H), t.height) * t.stride];
    c[2] = float(p & 0xFF) * (1.0f / 255.0f); c[1] = float((p >> 8) & 0xFF) * (1.0f / 255.0f);
    c[0] = float((p >> 16) & 0xFF) * (1.0f / 255.0f); c[3] = float(p >> 24) * (1.0f / 255.0f);
  }
}
WR_DEVICE float wr_mix_lum(const float (&c)[3]) { return (c[0] * 0.3f + c[1] * 0.59f) + c[2] * 0.11f; }
WR_DEVICE void wr_mix_clip_color(float (&C)[3]) {
  const float L = wr_mix_lum(C);
  const float n = wr_min(C[0], wr_min(C[1], C[2])), x = wr_max(C[0], wr_max(C[1], C[2]));
  if (n < 0.0f) for (int i = 0; i < 3; i++) C[i] = L + (((C[i] - L) * L) / (L - n));
  if (x > 1.0f) for (int i = 0; i < 3; i++) C[i] = L + (((C[i] - L) * (1.0f - L)) / (x - L));
}
WR_DEVICE void wr_mix_set_lum(float (&C)[3], float l) {
  const float d = l - wr_mix_lum(C);
  for (int i = 0; i < 3; i++) C[i] = C[i] + d;
  wr_mix_clip_color(C);
}
WR_DEVICE float wr_mix_sat(const float (&c)[3]) { return wr_max(c[0], wr_max(c[1], c[2])) - wr_min(c[0], wr_min(c[1], c[2])); }
WR_DEVICE void wr_mix_set_sat_inner(float& Cmin, float& Cmid, float& Cmax, float s) {
  if (Cmax > Cmin) { Cmid = ((Cmid - Cmin) * s) / (Cmax - Cmin); Cmax = s; } else { Cmid = 0.0f; Cmax = 0.0f; }
  Cmin = 0.0f;
}
WR_DEVICE void wr_mix_set_sat(float (&C)[3], float s) {
  float& r = C[0]; float& g = C[1]; float& b = C[2];
  if (r <= g) {
    if (g <= b) wr_mix_set_sat_inner(r, g, b, s);
    else if (r <= b) wr_mix_set_sat_inner(r, b, g, s);
    else wr_mix_set_sat_inner(b, r, g, s);
  } else {
    if (r <= b) wr_mix_set_sat_inner(g, r, b, s);
    else if (g <= b) wr_mix_set_sat_inner(g, b, r, s);
    else wr_mix_set_sat_inner(b, g, r, s);
  }
}
WR_DEVICE float wr_mix_hard_light(float Cb, float Cs) {
  const float m = Cb * (2.0f * Cs);
  const float t = 2.0f * Cs - 1.0f;
  const float sc = Cb + t - (Cb * t);
  const float st = Cs < 0.5f ? 0.0f : 1.0f;         // step(edge, Cs)
  return (sc - m) * st + m;                          // mix(m, s, step)
}
__device__ __noinline__ WrWide wr_mix_blend_pixel(const WrPrim* Pp, const WrMixRec* Mp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrMixRec& M = *Mp;
  const WrTexDesc& tb = D->tex[WR_S_COLOR0];
  const WrTexDesc& ts = D->tex[WR_S_COLOR1];
  // v_backdrop_uv / v_src_uv of this pixel as the 4-wide fragment loop steps them, clamped to their sample bounds
  float bu, bv, su, sv;
  {
    const WrTexRow r = wr_tex_row(P, tb, y, runs, x);
    wr_tex_tail_uv(P, r, x - r.x0, bu, bv);
  }
  {
    WrPrim P2 = P;            // the same walk on the second varying's edges
    P2.uvL0[0] = M.sL0[0]; P2.uvL0[1] = M.sL0[1]; P2.uvLs[0] = M.sLs[0]; P2.uvLs[1] = M.sLs[1];
    P2.uvR0[0] = M.sR0[0]; P2.uvR0[1] = M.sR0[1]; P2.uvRs[0] = M.sRs[0]; P2.uvRs[1] = M.sRs[1];
    P2.uv_bounds[0] = M.s_bounds[0]; P2.uv_bounds[1] = M.s_bounds[1]; P2.uv_bounds[2] = M.s_bounds[2]; P2.uv_bounds[3] = M.s_bounds[3];
    P2.rows_linear = 0;
    const WrTexRow r = wr_tex_row(P2, ts, y, runs, x);
    wr_tex_tail_uv(P2, r, x - r.x0, su, sv);
  }
  return wr_mix_blend_main(Mp, D, bu, bv, su, sv);
}
// ... main() of brush_mix_blend with both sample positions in hand (brush_mix_blend.glsl: Cb / Cs unpremultiplied, the sixteen modes)
__device__ __noinline__ WrWide wr_mix_blend_main(const WrMixRec* Mp, const WrDrawDesc* D, float bu, float bv, float su, float sv) {
  const WrMixRec& M = *Mp;
  const WrTexDesc& tb = D->tex[WR_S_COLOR0];
  const WrTexDesc& ts = D->tex[WR_S_COLOR1];
  float Cb4[4], Cs4[4];
  wr_texture_rgba_f(tb, bu, bv, Cb4);
  wr_texture_rgba_f(ts, su, sv, Cs4);
  float Cb[3], Cs[3];
  for (int i = 0; i < 3; i++) { Cb[i] = Cb4[3] != 0.0f ? Cb4[i] / Cb4[3] : Cb4[i]; Cs[i] = Cs4[3] != 0.0f ? Cs4[i] / Cs4[3] : Cs4[i]; }
  float res[3] = {1.0f, 1.0f, 0.0f};
  switch (M.op & 0xFF) {
    case 1: for (int i = 0; i < 3; i++) res[i] = Cb[i] * Cs[i]; break;
    case 3: for (int i = 0; i < 3; i++) res[i] = wr_mix_hard_light(Cs[i], Cb[i]); break;
    case 4: for (int i = 0; i < 3; i++) res[i] = wr_min(Cs[i], Cb[i]); break;
    case 5: for (int i = 0; i < 3; i++) res[i] = wr_max(Cs[i], Cb[i]); break;
    case 6: for (int i = 0; i < 3; i++) res[i] = Cb[i] == 0.0f ? 0.0f : (Cs[i] == 1.0f ? 1.0f : wr_min(1.0f, Cb[i] / (1.0f - Cs[i]))); break;
    case 7: for (int i = 0; i < 3; i++) res[i] = Cb[i] == 1.0f ? 1.0f : (Cs[i] == 0.0f ? 0.0f : 1.0f - wr_min(1.0f, (1.0f - Cb[i]) / Cs[i])); break;
    case 8: for (int i = 0; i < 3; i++) res[i] = wr_mix_hard_light(Cb[i], Cs[i]); break;
    case 9:
      for (int i = 0; i < 3; i++) {
        if (Cs[i] <= 0.5f) res[i] = Cb[i] - (1.0f - 2.0f * Cs[i]) * Cb[i] * (1.0f - Cb[i]);
        else {
          const float Dd = Cb[i] <= 0.25f ? ((16.0f * Cb[i] - 12.0f) * Cb[i] + 4.0f) * Cb[i] : sqrtf(Cb[i]);
          res[i] = Cb[i] + (2.0f * Cs[i] - 1.0f) * (Dd - Cb[i]);
        }
      }
      break;
    case 10: for (int i = 0; i < 3; i++) res[i] = fabsf(Cb[i] - Cs[i]); break;
    case 12: { float c[3] = {Cs[0], Cs[1], Cs[2]}; wr_mix_set_sat(c, wr_mix_sat(Cb)); wr_mix_set_lum(c, wr_mix_lum(Cb)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; break; }
    case 13: { float c[3] = {Cb[0], Cb[1], Cb[2]}; wr_mix_set_sat(c, wr_mix_sat(Cs)); wr_mix_set_lum(c, wr_mix_lum(Cb)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; break; }
    case 14: { float c[3] = {Cs[0], Cs[1], Cs[2]}; wr_mix_set_lum(c, wr_mix_lum(Cb)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; break; }
    case 15: { float c[3] = {Cb[0], Cb[1], Cb[2]}; wr_mix_set_lum(c, wr_mix_lum(Cs)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; break; }
    default: break;
  }
  float rgb[3];
  for (int i = 0; i < 3; i++) rgb[i] = ((1.0f - Cb4[3]) * Cs[i] + Cb4[3] * res[i]) * Cs4[3];
  uint32_t pc[2];
  wr_pack_color(wf4{rgb[0], rgb[1], rgb[2], Cs4[3]}, pc);
  WrWide s; s.bg = pc[0]; s.ra = pc[1];
  return s;
}

// ---------------------------------------------------------------------------
// cs_svg_filter (cs_svg_filter.glsl:170-592) and cs_svg_filter_node (cs_svg_filter_node.glsl:397-857) fragment shaders, one
// pixel.  No span shader in swgl: main() on every chunk; per pixel in strict fp32, same operation order (the blend functions are
// brush_mix_blend's, pow() is glsl.h's approximation, the vector floor its int round trip).
WR_DEVICE float wr_svg_color_dodge(float Cb, float Cs) { return Cb == 0.0f ? 0.0f : (Cs == 1.0f ? 1.0f : wr_min(1.0f, Cb / (1.0f - Cs))); }
WR_DEVICE float wr_svg_color_burn(float Cb, float Cs) { return Cb == 1.0f ? 1.0f : (Cs == 0.0f ? 0.0f : 1.0f - wr_min(1.0f, (1.0f - Cb) / Cs)); }
WR_DEVICE float wr_svg_soft_light(float Cb, float Cs) {
  if (Cs <= 0.5f) return Cb - (1.0f - 2.0f * Cs) * Cb * (1.0f - Cb);
  const float Dd = Cb <= 0.25f ? ((16.0f * Cb - 12.0f) * Cb + 4.0f) * Cb : sqrtf(Cb);
  return Cb + (2.0f * Cs - 1.0f) * (Dd - Cb);
}
WR_DEVICE float wr_srgb_to_linear1(float c) { const float c1 = c / 12.92f, c2 = wr_glsl_pow(c / 1.055f + (0.055f / 1.055f), 2.4f); return c <= 0.04045f ? c1 : c2; }
WR_DEVICE float wr_linear_to_srgb1(float c) { const float c1 = c * 12.92f, c2 = 1.055f * wr_glsl_pow(c, 1.0f / 2.4f) - 0.055f; return c <= 0.0031308f ? c1 : c2; }
// the sixteen MixBlendModes on un-premultiplied colours: B(Cb, Cs) of the compositing spec; false: no such mode
WR_DEVICE bool wr_svg_blend_fn(int mode, const float (&Cb)[3], const float (&Cs)[3], float (&res)[3]) {
  switch (mode) {
    case 0: for (int i = 0; i < 3; i++) res[i] = Cs[i]; return true;
    case 1: for (int i = 0; i < 3; i++) res[i] = Cb[i] * Cs[i]; return true;
    case 2: for (int i = 0; i < 3; i++) res[i] = (Cb[i] + Cs[i]) - (Cb[i] * Cs[i]); return true;
    case 3: for (int i = 0; i < 3; i++) res[i] = wr_mix_hard_light(Cs[i], Cb[i]); return true;
    case 4: for (int i = 0; i < 3; i++) res[i] = wr_min(Cs[i], Cb[i]); return true;
    case 5: for (int i = 0; i < 3; i++) res[i] = wr_max(Cs[i], Cb[i]); return true;
    case 6: for (int i = 0; i < 3; i++) res[i] = wr_svg_color_dodge(Cb[i], Cs[i]); return true;
    case 7: for (int i = 0; i < 3; i++) res[i] = wr_svg_color_burn(Cb[i], Cs[i]); return true;
    case 8: for (int i = 0; i < 3; i++) res[i] = wr_mix_hard_light(Cb[i], Cs[i]); return true;
    case 9: for (int i = 0; i < 3; i++) res[i] = wr_svg_soft_light(Cb[i], Cs[i]); return true;
    case 10: for (int i = 0; i < 3; i++) res[i] = fabsf(Cb[i] - Cs[i]); return true;
    case 11: for (int i = 0; i < 3; i++) res[i] = (Cb[i] + Cs[i]) - ((2.0f * Cb[i]) * Cs[i]); return true;
    case 12: { float c[3] = {Cs[0], Cs[1], Cs[2]}; wr_mix_set_sat(c, wr_mix_sat(Cb)); wr_mix_set_lum(c, wr_mix_lum(Cb)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; return true; }
    case 13: { float c[3] = {Cb[0], Cb[1], Cb[2]}; wr_mix_set_sat(c, wr_mix_sat(Cs)); wr_mix_set_lum(c, wr_mix_lum(Cb)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; return true; }
    case 14: { float c[3] = {Cs[0], Cs[1], Cs[2]}; wr_mix_set_lum(c, wr_mix_lum(Cb)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; return true; }
    case 15: { float c[3] = {Cb[0], Cb[1], Cb[2]}; wr_mix_set_lum(c, wr_mix_lum(Cs)); res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; return true; }
    default: return false;
  }
}
// blend(Cs, Cb, mode) of cs_svg_filter.glsl:322-391
WR_DEVICE void wr_svg_blend(const float (&Cs)[4], const float (&Cb)[4], int mode, float (&out)[4]) {
  float res[3] = {1.0f, 0.0f, 0.0f};
  const float cb3[3] = {Cb[0], Cb[1], Cb[2]}, cs3[3] = {Cs[0], Cs[1], Cs[2]};
  wr_svg_blend_fn(mode, cb3, cs3, res);
  for (int i = 0; i < 3; i++) {
    const float rgb = ((1.0f - Cb[3]) * Cs[i]) + (Cb[3] * res[i]);
    const float x = Cb[i] * Cb[3];
    out[i] = (rgb - x) * Cs[3] + x;             // mix(vec4(Cb.rgb * Cb.a, Cb.a), vec4(rgb, 1.0), Cs.a)
  }
  out[3] = (1.0f - Cb[3]) * Cs[3] + Cb[3];
}
__device__ __noinline__ WrWide wr_svg_filter_pixel(const WrPrim* Pp, const WrSvgRec* Sp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrSvgRec& S = *Sp;
  const WrTexDesc& t0 = D->tex[WR_S_COLOR0];
  const WrTexDesc& t1 = D->tex[WR_S_COLOR1];
  // vInput1Uv / vInput2Uv of this pixel as the 4-wide fragment loop steps them
  float u1, v1, u2, v2;
  {
    const WrTexRow r = wr_tex_row(P, t0, y, runs, x);
    wr_tex_tail_uv(P, r, x - r.x0, u1, v1);
  }
  {
    WrPrim P2 = P;            // the same walk on the second varying's edges
    P2.uvL0[0] = S.sL0[0]; P2.uvL0[1] = S.sL0[1]; P2.uvLs[0] = S.sLs[0]; P2.uvLs[1] = S.sLs[1];
    P2.uvR0[0] = S.sR0[0]; P2.uvR0[1] = S.sR0[1]; P2.uvRs[0] = S.sRs[0]; P2.uvRs[1] = S.sRs[1];
    P2.rows_linear = 0;
    const WrTexRow r = wr_tex_row(P2, t1, y, runs, x);
    wr_tex_tail_uv(P2, r, x - r.x0, u2, v2);
  }
  float A[4] = {0.f, 0.f, 0.f, 0.f}, B[4] = {0.f, 0.f, 0.f, 0.f};       // sampleInUvRect of the two inputs
  if (S.input_count > 0) wr_texture_rgba_f(t0, wr_clamp(u1, S.rect1[0], S.rect1[2]), wr_clamp(v1, S.rect1[1], S.rect1[3]), A);
  if (S.input_count > 1) wr_texture_rgba_f(t1, wr_clamp(u2, S.rect2[0], S.rect2[2]), wr_clamp(v2, S.rect2[1], S.rect2[3]), B);
  float res[4] = {1.0f, 0.0f, 0.0f, 1.0f};
  const WrTexDesc& gc = D->tex[WR_S_GPU_CACHE];
  if (!S.node) {
    // cs_svg_filter: un-premultiplied inputs
    if (S.input_count > 0 && A[3] != 0.0f) for (int i = 0; i < 3; i++) A[i] = A[i] / A[3];
    if (S.input_count > 1 && B[3] != 0.0f) for (int i = 0; i < 3; i++) B[i] = B[i] / B[3];
    bool premul = true;
    switch (S.kind) {
      case 0: wr_svg_blend(A, B, S.data[0], res); premul = false; break;
      case 1: for (int i = 0; i < 4; i++) res[i] = S.fdata0[i]; premul = false; break;
      case 2: for (int i = 0; i < 3; i++) res[i] = wr_linear_to_srgb1(A[i]); res[3] = A[3]; break;
      case 3: for (int i = 0; i < 3; i++) res[i] = wr_srgb_to_linear1(A[i]); res[3] = A[3]; break;
      case 4: for (int i = 0; i < 3; i++) res[i] = A[i]; res[3] = A[3] * S.float0; break;
      case 5: {
        const float* m = S.color_mat;
        for (int i = 0; i < 4; i++) res[i] = wr_clamp((m[i] * A[0] + m[4 + i] * A[1] + m[8 + i] * A[2] + m[12 + i] * A[3]) + S.fdata0[i], 0.0f, 1.0f);
        break;
      }
      case 6: {
        const float shadow[4] = {S.fdata0[0], S.fdata0[1], S.fdata0[2], B[3] * S.fdata0[3]};
        wr_svg_blend(A, shadow, 0, res); premul = false;
        break;
      }
      case 7: {
        const float ou = u1 + S.fdata0[0], ov = v1 + S.fdata0[1];
        wr_texture_rgba_f(t0, wr_clamp(ou, S.rect1[0], S.rect1[2]), wr_clamp(ov, S.rect1[1], S.rect1[3]), res);
        // point_inside_rect: (step(p0, p) - step(p1, p)).x * .y
        const float sx = wr_step01(S.fdata1[0], ou) - wr_step01(S.fdata1[2], ou), sy = wr_step01(S.fdata1[1], ov) - wr_step01(S.fdata1[3], ov);
        const float in = sx * sy;
        for (int i = 0; i < 4; i++) res[i] = res[i] * in;
        premul = false;
        break;
      }
      case 8: {             // ComponentTransfer (cs_svg_filter.glsl:415-466)
        float ch[4] = {A[0], A[1], A[2], A[3]};
        int offset = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int fn = S.funcs[i];
          if (fn == 1 || fn == 2) {
            const int k = int(wr_glsl_floor(ch[i] * 255.0f + 0.5f));
            const wf4 tx = wr_fetch_f(gc, S.data[0] + (offset + k / 4), S.data[1]);
            const int sel = k % 4;
            const float v = sel == 0 ? tx.x : (sel == 1 ? tx.y : (sel == 2 ? tx.z : (sel == 3 ? tx.w : 0.0f)));
            ch[i] = wr_clamp(v, 0.0f, 1.0f);
            offset += 64;
          } else if (fn == 3) {
            const wf4 tx = wr_fetch_f(gc, S.data[0] + offset, S.data[1]);
            ch[i] = wr_clamp(tx.x * ch[i] + tx.y, 0.0f, 1.0f);
            offset += 1;
          } else if (fn == 4) {
            const wf4 tx = wr_fetch_f(gc, S.data[0] + offset, S.data[1]);
            ch[i] = wr_clamp(tx.x * wr_glsl_pow(ch[i], tx.y) + tx.z, 0.0f, 1.0f);
            offset += 1;
          }
        }
        for (int i = 0; i < 4; i++) res[i] = ch[i];
        break;
      }
      case 9: for (int i = 0; i < 4; i++) res[i] = A[i]; break;
      case 10: {            // composite(Cs = Ca, Cb, mode) (cs_svg_filter.glsl:470-507)
        const float* Cs = A; const float* Cb = B;
        float Cr[4] = {0.0f, 1.0f, 0.0f, 1.0f};
        switch (S.data[0]) {
          case 0: for (int i = 0; i < 3; i++) Cr[i] = (Cs[3] * Cs[i]) + ((Cb[3] * Cb[i]) * (1.0f - Cs[3])); Cr[3] = Cs[3] + (Cb[3] * (1.0f - Cs[3])); break;
          case 1: for (int i = 0; i < 3; i++) Cr[i] = (Cs[3] * Cs[i]) * Cb[3]; Cr[3] = Cs[3] * Cb[3]; break;
          case 2: for (int i = 0; i < 3; i++) Cr[i] = (Cs[3] * Cs[i]) * (1.0f - Cb[3]); Cr[3] = Cs[3] * (1.0f - Cb[3]); break;
          case 3: for (int i = 0; i < 3; i++) Cr[i] = ((Cs[3] * Cs[i]) * Cb[3]) + ((Cb[3] * Cb[i]) * (1.0f - Cs[3])); Cr[3] = (Cs[3] * Cb[3]) + (Cb[3] * (1.0f - Cs[3])); break;
          case 4: for (int i = 0; i < 3; i++) Cr[i] = ((Cs[3] * Cs[i]) * (1.0f - Cb[3])) + ((Cb[3] * Cb[i]) * (1.0f - Cs[3])); Cr[3] = (Cs[3] * (1.0f - Cb[3])) + (Cb[3] * (1.0f - Cs[3])); break;
          case 5: for (int i = 0; i < 3; i++) Cr[i] = (Cs[3] * Cs[i]) + (Cb[3] * Cb[i]); Cr[3] = Cs[3] + Cb[3]; for (int i = 0; i < 4; i++) Cr[i] = wr_clamp(Cr[i], 0.0f, 1.0f); break;
          case 6: for (int i = 0; i < 4; i++) Cr[i] = wr_clamp((((S.fdata0[0] * Cs[i]) * Cb[i]) + (S.fdata0[1] * Cs[i])) + (S.fdata0[2] * Cb[i]) + S.fdata0[3], 0.0f, 1.0f); break;
          default: break;
        }
        for (int i = 0; i < 4; i++) res[i] = Cr[i];
        premul = false;
        break;
      }
      default: break;
    }
    if (premul) for (int i = 0; i < 3; i++) res[i] = res[i] * res[3];
  } else {
    // cs_svg_filter_node: raw premultiplied colours Rs / Rb (A / B), normalised Ns / Nb; odd kinds work in linear light
    float* Rs = A; float* Rb = B;
    float Ns[4] = {0.f, 0.f, 0.f, 0.f}, Nb[4] = {0.f, 0.f, 0.f, 0.f};
    const bool lin = (S.kind & 1) != 0;
    if (S.input_count > 0) {
      const float ia = 1.0f / wr_max(0.000001f, Rs[3]);
      for (int i = 0; i < 3; i++) Ns[i] = Rs[i] * ia;
      Ns[3] = Rs[3];
      if (lin) for (int i = 0; i < 3; i++) { Ns[i] = wr_srgb_to_linear1(Ns[i]); Rs[i] = Ns[i] * Rs[3]; }
    }
    if (S.input_count > 1) {
      const float ia = 1.0f / wr_max(0.000001f, Rb[3]);
      for (int i = 0; i < 3; i++) Nb[i] = Rb[i] * ia;
      Nb[3] = Rb[3];
      if (lin) for (int i = 0; i < 3; i++) { Nb[i] = wr_srgb_to_linear1(Nb[i]); Rb[i] = Nb[i] * Rb[3]; }
    }
    const int k2 = S.kind >> 1;
    bool done = false;
    if (k2 == 0) { for (int i = 0; i < 4; i++) res[i] = Rs[i]; }                             // IDENTITY
    else if (k2 == 1) { for (int i = 0; i < 4; i++) res[i] = Rs[i] * S.float0; }             // OPACITY
    else if (k2 == 2) { res[0] = res[1] = res[2] = 0.0f; res[3] = Rs[3]; done = true; }      // TO_ALPHA: returns before the sRGB step
    else if (k2 >= 3 && k2 <= 18) {                                                          // FILTER_BLEND_* in the shader's alphabetical order
      // kind / 2 -> MixBlendMode of wr_svg_blend_fn; -1: the closed forms on premultiplied colours below
      const int modes[16] = {14, 7, 6, -4, -10, -11, 8, 12, -5, 15, -1, -100, 3, 13, -2, 9};
      const int mode = modes[k2 - 3];
      const float ra = (Rb[3] * (1.0f - Rs[3])) + Rs[3];
      if (mode >= 0) {
        const float nb3[3] = {Nb[0], Nb[1], Nb[2]}, ns3[3] = {Ns[0], Ns[1], Ns[2]};
        float r3[3] = {1.0f, 0.0f, 0.0f};
        wr_svg_blend_fn(mode, nb3, ns3, r3);
        for (int i = 0; i < 3; i++) res[i] = (((1.0f - Rb[3]) * Rs[i]) + ((1.0f - Rs[3]) * Rb[i])) + ((Rs[3] * Rb[3]) * r3[i]);
        res[3] = ra;
      } else if (mode == -4) { for (int i = 0; i < 3; i++) res[i] = (Rs[i] + Rb[i]) - wr_max(Rs[i] * Rb[3], Rb[i] * Rs[3]); res[3] = ra; }                 // DARKEN
      else if (mode == -10) { for (int i = 0; i < 3; i++) res[i] = (Rs[i] + Rb[i]) - (2.0f * wr_min(Rs[i] * Rb[3], Rb[i] * Rs[3])); res[3] = ra; }         // DIFFERENCE
      else if (mode == -11) { for (int i = 0; i < 3; i++) res[i] = (Rs[i] + Rb[i]) - (2.0f * (Rs[i] * Rb[i])); res[3] = ra; }                               // EXCLUSION
      else if (mode == -5) { for (int i = 0; i < 3; i++) res[i] = (Rs[i] + Rb[i]) - wr_min(Rs[i] * Rb[3], Rb[i] * Rs[3]); res[3] = ra; }                  // LIGHTEN
      else if (mode == -1) { for (int i = 0; i < 3; i++) res[i] = ((Rs[i] * (1.0f - Rb[3])) + (Rb[i] * (1.0f - Rs[3]))) + (Rs[i] * Rb[i]); res[3] = ra; }  // MULTIPLY
      else if (mode == -100) { for (int i = 0; i < 4; i++) res[i] = (Rb[i] * (1.0f - Rs[3])) + Rs[i]; }                                                    // NORMAL
      else { for (int i = 0; i < 3; i++) res[i] = (Rs[i] + Rb[i]) - (Rs[i] * Rb[i]); res[3] = ra; }                                                         // SCREEN
    } else if (k2 == 19) {                                                                   // COLOR_MATRIX
      const float* m = S.color_mat;
      for (int i = 0; i < 4; i++) res[i] = wr_clamp((m[i] * Ns[0] + m[4 + i] * Ns[1] + m[8 + i] * Ns[2] + m[12 + i] * Ns[3]) + S.fdata0[i], 0.0f, 1.0f);
      for (int i = 0; i < 3; i++) res[i] = res[i] * res[3];
    } else if (k2 == 20) {                                                                   // COMPONENT_TRANSFER: a [256] table of RGBA blocks
      int k[4];
      for (int i = 0; i < 4; i++) k[i] = int(wr_glsl_floor(wr_clamp(Ns[i] * 255.0f, 0.0f, 255.0f)));
      res[0] = wr_fetch_f(gc, S.data[0] + k[0], S.data[1]).x; res[1] = wr_fetch_f(gc, S.data[0] + k[1], S.data[1]).y;
      res[2] = wr_fetch_f(gc, S.data[0] + k[2], S.data[1]).z; res[3] = wr_fetch_f(gc, S.data[0] + k[3], S.data[1]).w;
      for (int i = 0; i < 3; i++) res[i] = res[i] * res[3];
    } else if (k2 == 21) { for (int i = 0; i < 4; i++) res[i] = wr_clamp(((((Rs[i] * Rb[i]) * S.fdata0[0]) + (Rs[i] * S.fdata0[1])) + (Rb[i] * S.fdata0[2])) + S.fdata0[3], 0.0f, 1.0f); }    // ARITHMETIC
    else if (k2 == 22) { for (int i = 0; i < 4; i++) res[i] = (Rs[i] * Rb[3]) + (Rb[i] * (1.0f - Rs[3])); }                       // ATOP
    else if (k2 == 23) { for (int i = 0; i < 4; i++) res[i] = Rs[i] * Rb[3]; }                                                   // IN
    else if (k2 == 24) { for (int i = 0; i < 4; i++) res[i] = wr_clamp(Rs[i] + Rb[i], 0.0f, 1.0f); }                             // LIGHTER
    else if (k2 == 25) { for (int i = 0; i < 4; i++) res[i] = Rs[i] * (1.0f - Rb[3]); }                                          // OUT
    else if (k2 == 26) { for (int i = 0; i < 4; i++) res[i] = Rs[i] + (Rb[i] * (1.0f - Rs[3])); }                                // OVER
    else if (k2 == 27) { for (int i = 0; i < 4; i++) res[i] = (Rs[i] * (1.0f - Rb[3])) + (Rb[i] * (1.0f - Rs[3])); }             // XOR
    else if (k2 == 35) { for (int i = 0; i < 4; i++) res[i] = Rs[i] + (S.fdata0[i] * (Rb[3] * (1.0f - Rs[3]))); }                // DROP_SHADOW
    else if (k2 == 36) { for (int i = 0; i < 4; i++) res[i] = S.fdata0[i]; }                                                     // FLOOD
    else if (k2 == 46) {                                                                     // TILE: rect_repeat(vInput1Uv, rect.xy, rect.zw), returns before the sRGB step
      // rect_repeat (rect.glsl): p0 + s * fract(is * r), r = p - p0, s = p1 - p0, is = 1 / max(s, 0.000001)
      const float sx = S.rect1[2] - S.rect1[0], sy = S.rect1[3] - S.rect1[1];
      const float isx = 1.0f / wr_max(sx, 0.000001f), isy = 1.0f / wr_max(sy, 0.000001f);
      const float rx = isx * (u1 - S.rect1[0]), ry = isy * (v1 - S.rect1[1]);
      const float tu = S.rect1[0] + sx * (rx - wr_glsl_floor(rx)), tv = S.rect1[1] + sy * (ry - wr_glsl_floor(ry));
      wr_texture_rgba_f(t0, wr_clamp(tu, S.rect1[0], S.rect1[2]), wr_clamp(tv, S.rect1[1], S.rect1[3]), res);
      done = true;
    }
    // (every other kind -- convolve, lighting, displacement, gaussian blur, image, morphology, turbulence -- has no case in the
    // reference's main(): the start colour (1, 0, 0, 1) goes through the sRGB step)
    if (!done && lin) {
      const float ia = 1.0f / wr_max(0.000001f, res[3]);
      for (int i = 0; i < 3; i++) res[i] = wr_linear_to_srgb1(res[i] * ia) * res[3];
    }
  }
  uint32_t pc[2];
  wr_pack_color(wf4{res[0], res[1], res[2], res[3]}, pc);
  WrWide w; w.bg = pc[0]; w.ra = pc[1];
  return w;
}

// ---------------------------------------------------------------------------
// brush_yuv_image: one pixel.  Span part: swgl_commitTextureLinearYUV (swgl_ext.h:1028-1340) -- every plane sampled with
// the quantised fallback stepping (LINEAR_QUANTIZE_UV, uv += uv_step per chunk, clamp, 7-bit bilinear), the samples put
// through the fixed-point matrix (YUVMatrix::convert, composite.h:722-768: 16-bit lanes, saturating adds).  Tail: main() ->
// sample_yuv (yuv.glsl:187-237) in float.
WR_DEVICE int wr_sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
WR_DEVICE int wr_wrap16(int v) { return (int)(int16_t)v; }
WR_DEVICE WrWide wr_yuv_convert(const WrYuvRec& Y, int y, int u, int v) {
  // yy = (u16(y) * yCoeffs) >> 1 (16-bit wrap), - yBias; uv - uvBias; br = addsat(yy & mask, coeff * uv) >> 6;
  // gg = addsat(yy, addsat(gu * u, gv * v)) >> 6; pack with unsigned saturation, alpha 255
  int yy = wr_wrap16(int((uint32_t(uint16_t(y)) * uint32_t(uint16_t(Y.ycoeff))) & 0xFFFFu) >> 1);
  yy = wr_wrap16(yy - Y.ybias);
  const int uu = wr_wrap16(u - Y.uvbias), vv = wr_wrap16(v - Y.uvbias);
  const int b = wr_sat16((yy & Y.brmask) + wr_wrap16(Y.bu * uu)) >> 6, r = wr_sat16((yy & Y.brmask) + wr_wrap16(Y.rv * vv)) >> 6;
  const int g = wr_sat16(yy + wr_sat16(wr_wrap16(Y.gu * uu) + wr_wrap16(Y.gv * vv))) >> 6;
  auto pk = [](int c) { return uint32_t(c < 0 ? 0 : (c > 255 ? 255 : c)); };
  WrWide s;
  s.bg = pk(b) | (pk(g) << 16); s.ra = pk(r) | (255u << 16);
  return s;
}
// linear_row_yuv's row-invariant integers (composite.h:999-1011, 1081-1122): the first chunk's 8.8 coordinates per lane, the steps per
// chunk, the chunk range that takes upscaleYUV42R8 and its averaged chroma coordinates at that range's start
struct WrYuvRowArgs { int yU[4], cU[4], yDU, cDU, fast0, fast1, cA, cB, color_depth; };
WR_DEVICE void wr_yuv_row_chunk(const WrTexDesc& Ty, const WrTexDesc& Tu, const WrTexDesc& Tv, const WrYuvRowArgs& R, int yV, int cV, int c,
                                int (&ys)[4], int (&us)[4], int (&vs)[4]);
// linear_row_yuv's setup for a run of `span` pixels from (su, du) on the luma plane and (cu, cdu) on the chroma planes, 1/128 texels
// (the device-side twin of what CompositeYUV's host code computes once per call)
WR_DEVICE void wr_yuv_row_setup(WrYuvRowArgs& R, const WrTexDesc& Ty, const WrTexDesc& Tu, float su, float du, float cu, float cdu, int span) {
  const int STEP_BITS = 8;
  const float yl1 = su + du, yl2 = yl1 + du, yl3 = yl2 + du, cl1 = cu + cdu, cl2 = cl1 + cdu, cl3 = cl2 + cdu;      // init_interp
  R.yU[0] = int(su * float(1 << STEP_BITS)); R.yU[1] = int(yl1 * float(1 << STEP_BITS)); R.yU[2] = int(yl2 * float(1 << STEP_BITS)); R.yU[3] = int(yl3 * float(1 << STEP_BITS));
  R.cU[0] = int(cu * float(1 << STEP_BITS)); R.cU[1] = int(cl1 * float(1 << STEP_BITS)); R.cU[2] = int(cl2 * float(1 << STEP_BITS)); R.cU[3] = int(cl3 * float(1 << STEP_BITS));
  R.yDU = int(float(4 << STEP_BITS) * du); R.cDU = int(float(4 << STEP_BITS) * cdu);
  R.fast0 = R.fast1 = 0; R.cA = R.cB = 0;
  if (Ty.format != WR_FMT_R16 && R.yDU >= R.cDU && R.cDU > 0 && R.yDU <= (4 << (STEP_BITS + 7)) && R.cDU <= (2 << (STEP_BITS + 7))) {
    // the half-resolution fast path (composite.h:1081-1122): chunks until both coordinates are positive, then as many whole chunks
    // as stay four texels inside both planes
    int left = span, chunk = 0;
    int yx = R.yU[0], cx = R.cU[0];
    int cl[4] = {R.cU[0], R.cU[1], R.cU[2], R.cU[3]};
    for (; (yx < 0 || cx < 0) && left >= 4; left -= 4) {
      yx = int(uint32_t(yx) + uint32_t(R.yDU)); cx = int(uint32_t(cx) + uint32_t(R.cDU));
      for (int i = 0; i < 4; i++) cl[i] = int(uint32_t(cl[i]) + uint32_t(R.cDU));
      chunk++;
    }
    const int inside = wr_imin(wr_imin((((Ty.width - 4) << (STEP_BITS + 7)) - yx) / R.yDU, (((Tu.width - 4) << (STEP_BITS + 7)) - cx) / R.cDU) * 4, left & ~3);
    if (inside > 0) {
      R.fast0 = chunk; R.fast1 = chunk + inside / 4;
      R.cA = (cl[0] + cl[1]) >> 1; R.cB = (cl[2] + cl[3]) >> 1;      // cU = (cU.xzxz + cU.ywyw) >> 1
    }
  }
}

// The span pixels of a PLANAR frame under a TEXTURE_RECT key: three linear sampler2DRect planes select blendYUV's second overload
// (swgl_ext.h:1195-1283).  When the planes agree (one format, chroma planes of one size sampled at the same coordinates, no change of
// row along the span, x increasing) the span is cut in three: chunks before both clamp rects are entered take the quantised float
// stepping (blendYUVFallback), the chunks inside them CompositeYUV's inner loop (linear_row_yuv from the coordinates reached by
// ONE multiply-add), the rest the float stepping again from where that run ended.  Returns false where the shared routine's value is
// the reference's (conditions not met, or the pixel lies in the first part).
WR_DEVICE bool wr_yuv_rect_span_pixel(const WrPrim* Pp, const WrYuvRec& Y, const WrDrawDesc* D, int x, int y, const WrRuns* runs, WrWide& out) {
  const WrTexDesc& T0 = D->tex[WR_S_COLOR0]; const WrTexDesc& T1 = D->tex[WR_S_COLOR1]; const WrTexDesc& T2 = D->tex[WR_S_COLOR2];
  if (!(T0.format == T1.format && T1.format == T2.format && T1.width == T2.width && T1.height == T2.height)) return false;
  float q[3][4], qy[3][4], stepx[3], stepy[3], minx[3], maxx[3], miny[3], maxy[3];
  int n = 0, span = 0;
  for (int pl = 0; pl < 3; pl++) {
    WrPrim P2 = *Pp;
    P2.kind = WR_PK_TEX_R8;
    if (pl > 0) {
      const float* L0 = pl == 1 ? Y.uL0 : Y.vL0; const float* Ls = pl == 1 ? Y.uLs : Y.vLs;
      const float* R0 = pl == 1 ? Y.uR0 : Y.vR0; const float* Rs = pl == 1 ? Y.uRs : Y.vRs; const float* Bd = pl == 1 ? Y.u_bounds : Y.v_bounds;
      P2.uvL0[0] = L0[0]; P2.uvL0[1] = L0[1]; P2.uvLs[0] = Ls[0]; P2.uvLs[1] = Ls[1];
      P2.uvR0[0] = R0[0]; P2.uvR0[1] = R0[1]; P2.uvRs[0] = Rs[0]; P2.uvRs[1] = Rs[1];
      P2.uv_bounds[0] = Bd[0]; P2.uv_bounds[1] = Bd[1]; P2.uv_bounds[2] = Bd[2]; P2.uv_bounds[3] = Bd[3];
      P2.rows_linear = 0;
    }
    const WrTexDesc& t = D->tex[WR_S_COLOR0 + pl];
    const WrTexRow r = wr_tex_row(P2, t, y, runs, x, false);
    if (pl == 0) { n = x - r.x0; span = r.span; if (n >= span) return false; }
    const float W = t.sw, H = t.sh, qs = 128.0f, qo = 0.5f - 0.5f * qs;
    for (int i = 0; i < 4; i++) { q[pl][i] = r.lu[i] * W * qs + qo; qy[pl][i] = r.lv[i] * H * qs + qo; }
    stepx[pl] = 4.0f * (q[pl][1] - q[pl][0]); stepy[pl] = 4.0f * (qy[pl][1] - qy[pl][0]);
    minx[pl] = wr_max(P2.uv_bounds[0] * W * qs + qo, 0.0f); miny[pl] = wr_max(P2.uv_bounds[1] * H * qs + qo, 0.0f);
    maxx[pl] = wr_max(P2.uv_bounds[2] * W * qs + qo, minx[pl]); maxy[pl] = wr_max(P2.uv_bounds[3] * H * qs + qo, miny[pl]);
  }
  if (!(stepy[0] == 0.0f && stepx[0] > 0.0f && stepy[1] == 0.0f && stepx[1] > 0.0f && stepx[1] == stepx[2] && stepy[1] == stepy[2] &&
        q[1][0] == q[2][0] && qy[1][0] == qy[2][0])) return false;
  const int chunks = span >> 2, c = n >> 2, k = n & 3;
  int outside = wr_imin(int(ceilf(wr_max((minx[0] - q[0][0]) / stepx[0], (minx[1] - q[1][0]) / stepx[1]))), chunks);
  if (outside < 0) outside = 0;
  if (c < outside) return false;
  float u0[4], u1[4], u2[4];
  for (int i = 0; i < 4; i++) { u0[i] = q[0][i]; u1[i] = q[1][i]; u2[i] = q[2][i]; }
  if (outside > 0) for (int i = 0; i < 4; i++) { u0[i] += float(outside) * stepx[0]; u1[i] += float(outside) * stepx[1]; u2[i] += float(outside) * stepx[2]; }
  int inside = wr_imin(int(wr_min((maxx[0] - u0[0]) / stepx[0], (maxx[1] - u1[0]) / stepx[1])), chunks - outside);
  if (inside < 0) inside = 0;
  int s3[3];
  if (c < outside + inside) {
    WrYuvRowArgs R;
    R.color_depth = (T0.format == WR_FMT_R16 ? 16 : 8) - Y.rescale;
    wr_yuv_row_setup(R, T0, T1, u0[0], stepx[0] / 4.0f, u1[0], stepx[1] / 4.0f, inside * 4);
    int ys[4], us[4], vs[4];
    wr_yuv_row_chunk(T0, T1, T2, R, int(qy[0][0]), int(qy[1][0]), c - outside, ys, us, vs);
    s3[0] = wr_pick4i(ys, k); s3[1] = wr_pick4i(us, k); s3[2] = wr_pick4i(vs, k);
  } else {
    // what is left of the span: the float stepping again, from the coordinates the inside run ended on
    if (inside > 0) for (int i = 0; i < 4; i++) { u0[i] += float(inside) * stepx[0]; u1[i] += float(inside) * stepx[1]; u2[i] += float(inside) * stepx[2]; }
    const int cc = c - outside - inside;
    const float us3[3] = {wr_pick4(u0, k), wr_pick4(u1, k), wr_pick4(u2, k)};
    for (int pl = 0; pl < 3; pl++) {
      const WrTexDesc& t = D->tex[WR_S_COLOR0 + pl];
      const int iqx = int(wr_clamp(wr_accum(us3[pl], stepx[pl], cc), minx[pl], maxx[pl]));
      const int iqy = int(wr_clamp(wr_accum(wr_pick4(qy[pl], k), stepy[pl], cc), miny[pl], maxy[pl]));
      int v4[4];
      if (t.format == WR_FMT_R16) { wr_bilinear16<1>(t, iqx, iqy, v4); s3[pl] = v4[0] >> ((16 - Y.rescale - 1) - 8); }
      else { wr_bilinear<1>(t, iqx, iqy, v4); s3[pl] = v4[0]; }
    }
  }
  out = wr_yuv_convert(Y, s3[0], s3[1], s3[2]);
  return true;
}

__device__ __noinline__ WrWide wr_yuv_pixel(const WrPrim* Pp, const WrYuvRec* Yp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrYuvRec& Y = *Yp;
  const int planes = Y.format == 3 ? 3 : 2;
  bool all_linear = true;
  for (int pl = 0; pl < planes; pl++) all_linear = all_linear && D->tex[WR_S_COLOR0 + pl].linear != 0;
  if ((D->flags & WR_DF_TEX_RECT) && planes == 3 && all_linear) {
    WrWide w;
    if (wr_yuv_rect_span_pixel(Pp, Y, D, x, y, runs, w)) return w;
  }
  int sample[3] = {0, 0, 0};          // y, u, v as the span shader's u16 lanes
  float fs[3] = {0.f, 0.f, 0.f};      // ... and as main()'s floats
  bool tail = false;
  for (int pl = 0; pl < planes; pl++) {
    WrPrim P2 = *Pp;                  // the same walk on this plane's varying (wr_mix_blend_pixel)
    P2.kind = WR_PK_TEX_R8;           // (the quantised fallback stepping of an R8 / RG8 plane: filter 1)
    if (pl > 0) {
      const float* L0 = pl == 1 ? Y.uL0 : Y.vL0; const float* Ls = pl == 1 ? Y.uLs : Y.vLs;
      const float* R0 = pl == 1 ? Y.uR0 : Y.vR0; const float* Rs = pl == 1 ? Y.uRs : Y.vRs; const float* Bd = pl == 1 ? Y.u_bounds : Y.v_bounds;
      P2.uvL0[0] = L0[0]; P2.uvL0[1] = L0[1]; P2.uvLs[0] = Ls[0]; P2.uvLs[1] = Ls[1];
      P2.uvR0[0] = R0[0]; P2.uvR0[1] = R0[1]; P2.uvRs[0] = Rs[0]; P2.uvRs[1] = Rs[1];
      P2.uv_bounds[0] = Bd[0]; P2.uv_bounds[1] = Bd[1]; P2.uv_bounds[2] = Bd[2]; P2.uv_bounds[3] = Bd[3];
      P2.rows_linear = 0;
    }
    const WrTexDesc& t = D->tex[WR_S_COLOR0 + pl];
    const WrTexRow r = wr_tex_row(P2, t, y, runs, x, !all_linear);
    const int n = x - r.x0;
    const float W = t.sw, H = t.sh;
    if (n < r.span) {
      const float qs = 128.0f, qo = 0.5f - 0.5f * qs;
      float q[4], qy[4];
      for (int i = 0; i < 4; i++) { q[i] = r.lu[i] * W * qs + qo; qy[i] = r.lv[i] * H * qs + qo; }
      const float stepx = 4.0f * (q[1] - q[0]), stepy = 4.0f * (qy[1] - qy[0]);
      const float minx = wr_max(P2.uv_bounds[0] * W * qs + qo, 0.0f), miny = wr_max(P2.uv_bounds[1] * H * qs + qo, 0.0f);
      const float maxx = wr_max(P2.uv_bounds[2] * W * qs + qo, minx), maxy = wr_max(P2.uv_bounds[3] * H * qs + qo, miny);
      int v4[4];
      if (t.format == WR_FMT_R16 || t.format == WR_FMT_RG16) {
        // (blendYUV's stepping, as wr_linear_span_pixel's fallback does it; the samples shifted down to the matrix's 8 + rescale bits)
        const int c = n >> 2, k = n & 3;
        const int iqx = int(wr_clamp(wr_accum(q[k], stepx, c), minx, maxx)), iqy = int(wr_clamp(wr_accum(qy[k], stepy, c), miny, maxy));
        const int bits = (16 - Y.rescale - 1) - 8;
        if (t.format == WR_FMT_RG16) { wr_bilinear16<2>(t, iqx, iqy, v4); sample[1] = v4[0] >> bits; sample[2] = v4[1] >> bits; }
        else { wr_bilinear16<1>(t, iqx, iqy, v4); sample[pl] = v4[0] >> bits; }
      } else if (t.format == WR_FMT_RG8) {
        wr_linear_span_pixel<2>(t, q, qy, stepx, stepy, minx, maxx, miny, maxy, 1, r.span, n, v4);
        sample[1] = v4[0]; sample[2] = v4[1];
      } else {
        wr_linear_span_pixel<1>(t, q, qy, stepx, stepy, minx, maxx, miny, maxy, 1, r.span, n, v4);
        sample[pl] = v4[0];
      }
    } else {
      tail = true;
      float cu, cv;
      wr_tex_tail_uv(P2, r, n, cu, cv);
      if (t.format == WR_FMT_R16 || t.format == WR_FMT_RG16) {      // textureLinearR16 / RG16: sample * (1 / 32767)
        int v4[4] = {0, 0, 0, 0};
        const int iqx = int(cu * W * 128.0f + (0.5f - 64.0f)), iqy = int(cv * H * 128.0f + (0.5f - 64.0f));
        if (t.format == WR_FMT_RG16) { wr_bilinear16<2>(t, iqx, iqy, v4); fs[1] = float(v4[0]) * (1.0f / 32767.0f); fs[2] = float(v4[1]) * (1.0f / 32767.0f); }
        else { wr_bilinear16<1>(t, iqx, iqy, v4); fs[pl] = float(v4[0]) * (1.0f / 32767.0f); }
      } else if (t.format == WR_FMT_RG8) {
        int v4[4] = {0, 0, 0, 0};
        if (t.linear) wr_bilinear<2>(t, int(cu * W * 128.0f + (0.5f - 64.0f)), int(cv * H * 128.0f + (0.5f - 64.0f)), v4);
        else wr_fetch_texel<2>(t, (size_t)wr_clamp_coord(int(cu * W), t.width) + (size_t)wr_clamp_coord(int(cv * H), t.height) * t.stride, v4);
        fs[1] = float(v4[0]) * (1.0f / 255.0f); fs[2] = float(v4[1]) * (1.0f / 255.0f);
      } else {
        fs[pl] = wr_r8_texture(t, cu, cv);
      }
    }
  }
  if (!tail) return wr_yuv_convert(Y, sample[0], sample[1], sample[2]);
  // rgb = vRgbFromDebiasedYcbcr * (ycbcr_sample - vYcbcrBias); ALPHA_PASS: clamp to [0, 1]; alpha 1
  const float d0 = fs[0] - Y.bias[0], d1 = fs[1] - Y.bias[1], d2 = fs[2] - Y.bias[2];
  float rgb[3];
  for (int i = 0; i < 3; i++) rgb[i] = Y.mat[i] * d0 + Y.mat[3 + i] * d1 + Y.mat[6 + i] * d2;
  if (Pp->flags & WR_PF_TAIL_MODULATE) for (int i = 0; i < 3; i++) rgb[i] = wr_clamp(rgb[i], 0.0f, 1.0f);
  uint32_t pc[2];
  wr_pack_color(wf4{rgb[0], rgb[1], rgb[2], 1.0f}, pc);
  WrWide s; s.bg = pc[0]; s.ra = pc[1];
  return s;
}

// ---------------------------------------------------------------------------
// CompositeYUV: linear_row_yuv (composite.h:993-1157) chunk by chunk.  One thread = one 4-pixel chunk of one destination row.
WR_DEVICE int wr_i16(int v) { return (int)(int16_t)v; }
// textureLinearRowR8 / textureLinearRowPairedR8 for one lane: x in 1/128 texels, the row pair and its 7-bit fraction given
WR_DEVICE int wr_yuv_row_tap(const WrTexDesc& t, int qx, long long row_off, long long row_stride, int fracv) {
  int ix = qx >> 7;
  const int fracx = (((ix >= 0) ? qx : 0) | (ix > t.width - 2 ? -1 : 0)) & 0x7F;      // (127, not 128, past the last pair: composite.h:806)
  ix = wr_clamp_coord(ix, t.width - 1);
  const uint8_t* b = (const uint8_t*)t.ptr + row_off;
  const int a0 = b[ix], a1 = b[ix + 1], b0 = b[row_stride + ix], b1 = b[row_stride + ix + 1];
  const int l = wr_i16(a0 + wr_i16(wr_i16((b0 - a0) * fracv) >> 7)), r = wr_i16(a1 + wr_i16(wr_i16((b1 - a1) * fracv) >> 7));
  return wr_i16(l + wr_i16(wr_i16((r - l) * fracx) >> 7));
}
WR_DEVICE int wr_yuv_vlerp(const uint8_t* row, long long stride, long long x, int fracv) {      // one column of a row pair
  const int a = row[x], b = row[x + stride];
  return wr_i16(a + wr_i16(wr_i16((b - a) * fracv) >> 7));
}
// ... and one 4-pixel chunk `c` of the row: the planes' samples as the colour matrix takes them (yV, cV: int32_t(srcUV.y))
WR_DEVICE void wr_yuv_row_chunk(const WrTexDesc& Ty, const WrTexDesc& Tu, const WrTexDesc& Tv, const WrYuvRowArgs& R, int yV, int cV, int c,
                                int (&ys)[4], int (&us)[4], int (&vs)[4]) {
    int yq[4], cq[4];
    for (int i = 0; i < 4; i++) { yq[i] = (int)((uint32_t)R.yU[i] + (uint32_t)c * (uint32_t)R.yDU); cq[i] = (int)((uint32_t)R.cU[i] + (uint32_t)c * (uint32_t)R.cDU); }
    if (Ty.format == WR_FMT_R16) {
      const int bits = (R.color_depth - 1) - 8;
      for (int i = 0; i < 4; i++) {
        int t4[4];
        wr_bilinear16<1>(Ty, yq[i] >> 8, yV, t4); ys[i] = t4[0] >> bits;
        wr_bilinear16<1>(Tu, cq[i] >> 8, cV, t4); us[i] = t4[0] >> bits;
        wr_bilinear16<1>(Tv, cq[i] >> 8, cV, t4); vs[i] = t4[0] >> bits;
      }
    } else {
      const int yfv = yV & 0x7F, cfv = cV & 0x7F;
      yV >>= 7; cV >>= 7;
      const long long yoff = (long long)wr_clamp_coord(yV, Ty.height) * Ty.stride, ystr = (yV >= 0 && yV < Ty.height - 1) ? Ty.stride : 0;
      const long long coff = (long long)wr_clamp_coord(cV, Tu.height) * Tu.stride, cstr = (cV >= 0 && cV < Tu.height - 1) ? Tu.stride : 0;
      if (c >= R.fast0 && c < R.fast1) {
        // upscaleYUV42R8 (composite.h:857-986): luma per lane out of a 4 + 4 texel window, chroma at the chunk's two averaged
        // coordinates, the four pixels' chroma estimated from those two samples
        const int k = c - R.fast0;
        const uint8_t* yrow = (const uint8_t*)Ty.ptr + yoff;
        const uint8_t* urow = (const uint8_t*)Tu.ptr + coff;
        const uint8_t* vrow = (const uint8_t*)Tv.ptr + coff;
        const int ca = (int)((uint32_t)R.cA + (uint32_t)k * (uint32_t)R.cDU), cb = (int)((uint32_t)R.cB + (uint32_t)k * (uint32_t)R.cDU);
        int yI[4]; for (int i = 0; i < 4; i++) yI[i] = yq[i] >> 15;
        const int cIx = ca >> 15, cIy = cb >> 15;
        const int yInx = (int)((uint32_t)yq[0] + (uint32_t)R.yDU) >> 15, cInx = (int)((uint32_t)ca + (uint32_t)R.cDU) >> 15;
        int s[4], n[4];
        for (int i = 0; i < 4; i++) { s[i] = wr_yuv_vlerp(yrow, ystr, (long long)yI[0] + i, yfv); n[i] = wr_yuv_vlerp(yrow, ystr, (long long)yInx + i, yfv); }
        int ysh[4] = {s[0], s[1], s[2], s[3]};
        int ysn[4] = {s[1], s[2], s[3], yInx == yI[3] ? n[1] : n[0]};
        if (yI[1] == yI[0]) { const int a[4] = {ysh[0], ysh[0], ysh[1], ysh[2]}, b[4] = {ysn[0], ysn[0], ysn[1], ysn[2]}; for (int i = 0; i < 4; i++) { ysh[i] = a[i]; ysn[i] = b[i]; } }
        if (yI[2] == yI[1]) { const int a[4] = {ysh[0], ysh[1], ysh[1], ysh[2]}, b[4] = {ysn[0], ysn[1], ysn[1], ysn[2]}; for (int i = 0; i < 4; i++) { ysh[i] = a[i]; ysn[i] = b[i]; } }
        if (yI[3] == yI[2]) { const int a[4] = {ysh[0], ysh[1], ysh[2], ysh[2]}, b[4] = {ysn[0], ysn[1], ysn[2], ysn[2]}; for (int i = 0; i < 4; i++) { ysh[i] = a[i]; ysn[i] = b[i]; } }
        const int u0 = wr_yuv_vlerp(urow, cstr, cIx, cfv), u1 = wr_yuv_vlerp(urow, cstr, (long long)cIx + 1, cfv);
        const int v0 = wr_yuv_vlerp(vrow, cstr, cIx, cfv), v1 = wr_yuv_vlerp(vrow, cstr, (long long)cIx + 1, cfv);
        const int nu0 = wr_yuv_vlerp(urow, cstr, cInx, cfv), nu1 = wr_yuv_vlerp(urow, cstr, (long long)cInx + 1, cfv);
        const int nv0 = wr_yuv_vlerp(vrow, cstr, cInx, cfv), nv1 = wr_yuv_vlerp(vrow, cstr, (long long)cInx + 1, cfv);
        int csh[4] = {u0, u1, v0, v1};
        int csn[4] = {u1, cInx == cIy ? nu1 : nu0, v1, cInx == cIy ? nv1 : nv0};
        if (cIy == cIx) { csh[1] = csh[0]; csh[3] = csh[2]; csn[1] = csn[0]; csn[3] = csn[2]; }
        const int fr[8] = {(yq[0] >> 8) & 0x7F, (yq[1] >> 8) & 0x7F, (yq[2] >> 8) & 0x7F, (yq[3] >> 8) & 0x7F, (ca >> 8) & 0x7F, (cb >> 8) & 0x7F, (ca >> 8) & 0x7F, (cb >> 8) & 0x7F};
        int px[8];
        for (int i = 0; i < 4; i++) px[i] = wr_i16(ysh[i] + wr_i16(wr_i16((ysn[i] - ysh[i]) * fr[i]) >> 7));
        for (int i = 0; i < 4; i++) px[4 + i] = wr_i16(csh[i] + wr_i16(wr_i16((csn[i] - csh[i]) * fr[4 + i]) >> 7));
        const int uA = px[4], uB = px[5], vA = px[6], vB = px[7];
        for (int i = 0; i < 4; i++) ys[i] = px[i];
        us[0] = wr_i16(uA + (wr_i16(uA - uB) >> 2)); us[1] = wr_i16(uA + (wr_i16(uB - uA) >> 2)); us[2] = wr_i16(uB + (wr_i16(uA - uB) >> 2)); us[3] = wr_i16(uB + (wr_i16(uB - uA) >> 2));
        vs[0] = wr_i16(vA + (wr_i16(vA - vB) >> 2)); vs[1] = wr_i16(vA + (wr_i16(vB - vA) >> 2)); vs[2] = wr_i16(vB + (wr_i16(vA - vB) >> 2)); vs[3] = wr_i16(vB + (wr_i16(vB - vA) >> 2));
      } else {
        for (int i = 0; i < 4; i++) {
          ys[i] = wr_yuv_row_tap(Ty, yq[i] >> 8, yoff, ystr, yfv);
          us[i] = wr_yuv_row_tap(Tu, cq[i] >> 8, coff, cstr, cfv);
          vs[i] = wr_yuv_row_tap(Tv, cq[i] >> 8, coff, cstr, cfv);
        }
      }
    }
}
#ifndef WR_INST_ONLY
__global__ void wr_composite_yuv_kernel(WrYuvBlitArgs A) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = (A.span + 3) >> 2;
  if (tid >= (long long)chunks * A.rows) return;
  const int row = (int)(tid / chunks), c = (int)(tid % chunks);
  const int left = A.span - 4 * c, npx = left < 4 ? left : 4;
  uint32_t* dst = (uint32_t*)((uint8_t*)A.dst + (size_t)(A.dy0 + row) * A.dst_stride) + A.dx0 + 4 * c;
  WrYuvRec M;
  M.bu = A.bu; M.rv = A.rv; M.gu = A.gu; M.gv = A.gv; M.ycoeff = A.ycoeff; M.ybias = A.ybias; M.uvbias = A.uvbias; M.brmask = A.brmask;
  // the row's v coordinates: `srcUV.y += srcDUV.y` once per row
  const float sv = wr_accum(A.src_v0, A.src_dv, row), cvf = wr_accum(A.chroma_v0, A.chroma_dv, row);
  uint32_t out[4];
  if (A.nearest) {
    // a single texel of every plane, nearest, converted once (composite.h:1014-1031)
    auto fetch = [](const WrTexDesc& t, float fu, float fv) -> float {
      const int x = wr_clamp_coord(int(fu), t.width), y = wr_clamp_coord(int(fv), t.height);
      if (t.format == WR_FMT_R16) return float(((const uint16_t*)t.ptr)[(size_t)y * t.stride + x]) * (1.0f / 65535.0f);
      return float(((const uint8_t*)t.ptr)[(size_t)y * t.stride + x]) * (1.0f / 255.0f);
    };
    float yf = fetch(A.y, A.src_u0, sv), uf = fetch(A.u, A.chroma_u0, cvf), vf = fetch(A.v, A.chroma_u0, cvf);
    if (A.color_depth > 8) { const float k = float(1 << (16 - A.color_depth)); yf *= k; uf *= k; vf *= k; }
    const WrWide w = wr_yuv_convert(M, wr_i16(wr_round_pixel(yf)), wr_i16(wr_round_pixel(uf)), wr_i16(wr_round_pixel(vf)));
    const uint32_t p = wr_pack(w);
    for (int i = 0; i < 4; i++) out[i] = p;
  } else {
    int ys[4], us[4], vs[4];
    WrYuvRowArgs R;
    for (int i = 0; i < 4; i++) { R.yU[i] = A.yU[i]; R.cU[i] = A.cU[i]; }
    R.yDU = A.yDU; R.cDU = A.cDU; R.fast0 = A.fast0; R.fast1 = A.fast1; R.cA = A.cA; R.cB = A.cB; R.color_depth = A.color_depth;
    wr_yuv_row_chunk(A.y, A.u, A.v, R, int(sv), int(cvf), c, ys, us, vs);      // (int32_t(srcUV.y): truncation)
    for (int i = 0; i < 4; i++) out[i] = wr_pack(wr_yuv_convert(M, ys[i], us[i], vs[i]));
  }
  for (int i = 0; i < npx; i++) dst[i] = out[i];
}
#endif

// The part of a cs_blur row that every pixel of the row shares: the interpolants at the span start, the integer texel position the span
// shader starts from and how many pixels it draws (the rest of the row is main()'s).
struct WrBlurRow { float ou, ov, su, sv; int startX, curY, drawn; };
template <int FMT>
WR_DEVICE WrBlurRow wr_blur_row_setup(const WrPrim& P, const WrBlurRec& B, int y) {
  WrBlurRow R;
  const int tw = int(B.wh & 0xFFFF);
  const float W = float(tw), H = float(int(B.wh >> 16));
  // interpolants at the span start (as wr_tex_row)
  const int k = y - P.y0;
  const float Lu = wr_accum(P.uvL0[0], P.uvLs[0], k), Lv = wr_accum(P.uvL0[1], P.uvLs[1], k);
  const float Ru = wr_accum(P.uvR0[0], P.uvRs[0], k), Rv = wr_accum(P.uvR0[1], P.uvRs[1], k);
  float stepScale = 1.0f / (P.xr - P.xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  R.su = (Ru - Lu) * stepScale; R.sv = (Rv - Lv) * stepScale;
  const float start = float(P.x0) + 0.5f - P.xl;
  R.ou = Lu + R.su * start; R.ov = Lv + R.sv * start;
  const int len = P.x1 - P.x0, span = len >= 4 ? (len & ~3) : 0;
  const bool fmt_ok = B.format == FMT && B.ptr != nullptr;
  R.startX = int(R.ou * W); R.curY = int(R.ov * H);
  R.drawn = 0;
  if (fmt_ok && span > 0) {
    const int endX = wr_imin(wr_imin(B.bounds[2], R.startX + span), tw);
    if (endX - R.startX >= 4) R.drawn = (endX - R.startX) & ~3;
  }
  return R;
}
// Span pixel n < drawn of the row (blendGaussianBlur -> gaussianBlurHorizontal / Vertical, texture.h:1165-1308): the chunk's texel
// position is the integer position of the span start plus whole chunks; tap o reads the texels o to either side, each side clamped to
// the blur bounds -- along the row in whole-chunk terms for the horizontal pass (taps inside the chunk's own four texels are never
// clamped), along the column for the vertical one; 8.8 fixed-point weights, 16-bit wrap of every product, saturating adds.
// One loop for both directions and both formats: this is the code a thin blur level runs from a cold instruction cache.
template <int FMT>
WR_DEVICE WrWide wr_blur_span_px(const WrBlurRec& B, const WrBlurRow& R, const int n) {
  const int tw = int(B.wh & 0xFFFF), th = int(B.wh >> 16);
  constexpr int NCH = FMT == WR_FMT_RGBA8 ? 4 : 1;
  const int kk = n & 3, ix = R.startX + (n & ~3), curY = R.curY;
  const int radius = B.radius;
  const ptrdiff_t at = (ptrdiff_t)wr_clamp_coord(ix, tw - 1) + (ptrdiff_t)wr_clamp_coord(curY, th) * B.stride + kk;
  const bool hori = B.hori != 0;
  // horizontal: offsets in texels relative to the chunk start (rb / lb: how far the chunk start is from the bounds);
  // vertical: offsets in rows (amax / bmax)
  const int rlim = hori ? wr_imin(B.bounds[2], tw - 1) - ix : wr_imax(wr_imin(B.bounds[3], th - 1) - curY, 0);
  const int llim = hori ? ix - wr_imax(B.bounds[0], 0) : wr_imax(curY - wr_imax(B.bounds[1], 0), 0);
  const ptrdiff_t unit = hori ? 1 : (ptrdiff_t)B.stride;
  auto texel = [&](ptrdiff_t i) -> uint32_t { return FMT == WR_FMT_RGBA8 ? ((const uint32_t*)B.ptr)[i] : (uint32_t)((const uint8_t*)B.ptr)[i]; };
  uint32_t sum[NCH];
  {
    const uint32_t c = texel(at), w0 = B.weights[0];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) sum[ch] = (((c >> (8 * ch)) & 0xFF) * w0) & 0xFFFF;
  }
  // taps in batches of eight: the sixteen texel loads of a batch are issued together and waited for once (a tap per round trip
  // made a thin level as long as radius x pixels-per-lane dependent L2 accesses); a tap beyond the radius re-reads the last one
  // with weight 0, which leaves the saturating sum as it is
  for (int o0 = 1; o0 <= radius; o0 += 8) {
    uint32_t rr[8], ll[8], ww[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int o = wr_imin(o0 + j, radius);
      // (hori: kk + o stays unclamped while it is inside the chunk, then min(kk + o, rb) from the chunk start; likewise to the left)
      const int ro = hori ? ((kk + o <= 3) ? o : wr_imin(kk + o, rlim) - kk) : wr_imin(o, rlim);
      const int lo = hori ? ((o <= kk) ? o : wr_imin(o - kk, llim) + kk) : wr_imin(o, llim);
      rr[j] = texel(at + (ptrdiff_t)ro * unit); ll[j] = texel(at - (ptrdiff_t)lo * unit);
      ww[j] = o0 + j <= radius ? (uint32_t)B.weights[o] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) {
        const uint32_t t = ((((rr[j] >> (8 * ch)) & 0xFF) + ((ll[j] >> (8 * ch)) & 0xFF)) * ww[j]) & 0xFFFF;
        const uint32_t a = sum[ch] + t;
        sum[ch] = a > 0xFFFF ? 0xFFFF : a;
      }
    }
  }
  WrWide out;
  if (FMT == WR_FMT_RGBA8) {
    out.bg = (sum[0] >> 8) | ((sum[1] >> 8) << 16);
    out.ra = (sum[2] >> 8) | ((sum[3] >> 8) << 16);
  } else {
    out.bg = sum[0] >> 8; out.ra = 0;
  }
  return out;
}
// A pixel the span shader leaves to main() (cs_blur.glsl:137-181): the row's last (len & 3) pixels, or every pixel of a row it did not draw
template <int FMT>
__device__ __noinline__ WrWide wr_blur_tail_px(const WrPrim* Pp, const WrBlurRec* Bp, const WrBlurRow* Rp, const int n) {
  const WrPrim& P = *Pp; const WrBlurRec& B = *Bp; const WrBlurRow& R = *Rp;
  (void)P;
  const int tw = int(B.wh & 0xFFFF), th = int(B.wh >> 16);
  const float W = float(tw), H = float(th);
  const float su = R.su, sv = R.sv, ou = R.ou, ov = R.ov;
  const int drawn = R.drawn;
  WrWide out; out.bg = out.ra = 0;
  // ---- fragment shader: uv of this pixel = its init_interp lane, stepped by
  // `drawn` at once (DISPATCH_DRAW_SPAN) and then chunk by chunk
  const int lane = (n - drawn) & 3, m = (n - drawn) >> 2;
  float lu = ou, lv = ov;
  for (int i = 0; i < lane; i++) { lu += su; lv += sv; }
  if (drawn > 0) { const float chunks = float(drawn) * 0.25f; lu = lu + (su * 4.0f) * chunks; lv = lv + (sv * 4.0f) * chunks; }
  lu = wr_accum(lu, (su * 4.0f) * 1.0f, m); lv = wr_accum(lv, (sv * 4.0f) * 1.0f, m);
  const WrTexDesc t{B.ptr, tw, th, B.stride, (int16_t)B.format, (int16_t)B.linear, W, H};
  auto sample = [&](float uu, float vv, float (&c)[4]) {
    // texture(sColor0, uv): linear (texture.h:1028-1071 / 576-583) or nearest
    if (B.format == WR_FMT_R8) {
      float r;
      if (t.linear) r = float(wr_sample_linear_r8(t, int(uu * W * 128.0f + (0.5f - 64.0f)), int(vv * H * 128.0f + (0.5f - 64.0f)))) * (1.0f / 255.0f);
      else r = float(((const uint8_t*)t.ptr)[(size_t)wr_clamp_coord(int(uu * W), tw) + (size_t)wr_clamp_coord(int(vv * H), th) * t.stride]) * (1.0f / 255.0f);
      c[0] = r; c[1] = 0.0f; c[2] = 0.0f; c[3] = 1.0f;
    } else {
      uint32_t b, g, r, a;
      if (t.linear) {
        const WrWide s = wr_sample_linear_rgba8(t, int(uu * W * 128.0f + (0.5f - 64.0f)), int(vv * H * 128.0f + (0.5f - 64.0f)));
        b = s.bg & 0xFFFF; g = s.bg >> 16; r = s.ra & 0xFFFF; a = s.ra >> 16;
      } else {
        const uint32_t p = ((const uint32_t*)t.ptr)[(size_t)wr_clamp_coord(int(uu * W), tw) + (size_t)wr_clamp_coord(int(vv * H), th) * t.stride];
        b = p & 0xFF; g = (p >> 8) & 0xFF; r = (p >> 16) & 0xFF; a = p >> 24;
      }
      c[0] = float(r) * (1.0f / 255.0f); c[1] = float(g) * (1.0f / 255.0f);
      c[2] = float(b) * (1.0f / 255.0f); c[3] = float(a) * (1.0f / 255.0f);
    }
  };
  float gx = B.coeffs[0], gy = B.coeffs[1];
  const float gz = gy * gy;
  float c0[4], avg[4];
  if (!t.ptr) { c0[0] = c0[1] = c0[2] = c0[3] = 0.0f; } else sample(lu, lv, c0);
  for (int ch = 0; ch < 4; ch++) avg[ch] = c0[ch] * gx;
  const int support = wr_imin(B.radius, 300);
  for (int i = 1; i <= support; i += 2) {
    gx *= gy; gy *= gz;
    float sub = gx;
    gx *= gy; gy *= gz;
    sub += gx;
    const float ratio = gx / sub;
    const float offx = B.offset_scale[0] * (float(i) + ratio), offy = B.offset_scale[1] * (float(i) + ratio);
    float a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (t.ptr) {
      sample(wr_max(lu - offx, B.uv_rect[0]), wr_max(lv - offy, B.uv_rect[1]), a);
      sample(wr_min(lu + offx, B.uv_rect[2]), wr_min(lv + offy, B.uv_rect[3]), b);
    }
    for (int ch = 0; ch < 4; ch++) avg[ch] += (a[ch] + b[ch]) * sub;
  }
  if (FMT == WR_FMT_RGBA8) {
    if (B.format == WR_FMT_R8) { avg[1] = avg[2] = avg[3] = avg[0]; }   // ALPHA_TARGET shader on a colour target: vec4(r)
    uint32_t pc[2];
    wr_pack_color(wf4{avg[0], avg[1], avg[2], avg[3]}, pc);
    out.bg = pc[0]; out.ra = pc[1];
  } else {
    out.bg = uint32_t(wr_round_pixel(avg[0])) & 0xFFFF;
  }
  return out;
}
// pixel n of the row (n = x - P.x0)
template <int FMT>
WR_DEVICE WrWide wr_blur_row_pixel(const WrPrim& P, const WrBlurRec& B, const WrBlurRow& R, const int n) {
  if (n < R.drawn) return wr_blur_span_px<FMT>(B, R, n);
  return wr_blur_tail_px<FMT>(&P, &B, &R, n);
}
template <int FMT>
__device__ __noinline__ WrWide wr_blur_pixel(const WrPrim* Pp, const WrBlurRec* Bp, int x, int y) {
  const WrBlurRow R = wr_blur_row_setup<FMT>(*Pp, *Bp, y);
  return wr_blur_row_pixel<FMT>(*Pp, *Bp, R, x - Pp->x0);
}

// ---------------------------------------------------------------------------
// cs_clip_rectangle, one destination pixel of an R8 mask (returns the u16 value
// handed to the blend stage).  Span part: the rounded-rectangle span rasteriser
// of cs_clip_rectangle.glsl:223-495, replayed per pixel -- the run lengths are
// closed-form in chunk units, the local position of an AA chunk follows the
// same jump-then-accumulate sequence as the reference.  Tail (< 4 pixels) and
// spans shorter than 4: the fragment shader (:170-199).
WR_DEVICE float wr_clip_dist(const WrClipRec& C, float px, float py) {
  if (C.fast) {   // sd_rounded_box
    const float dx = fabsf(px) - C.params[0], dy = fabsf(py) - C.params[1];
    const float mx = wr_max(dx, 0.0f), my = wr_max(dy, 0.0f);
    return (sqrtf(mx * mx + my * my) + wr_min(wr_max(dx, dy), 0.0f)) - C.params[2];
  }
  // distance_to_rounded_rect, ellipse.glsl:50-92
  float cx = 1.0e-6f, cy = 1.0e-6f, cz = 1.0f, cw = 1.0f;
  if (px * C.plane[0][0] + py * C.plane[0][1] > C.plane[0][2]) { cx = C.center_radius[0][0] - px; cy = C.center_radius[0][1] - py; cz = C.center_radius[0][2]; cw = C.center_radius[0][3]; }
  if (px * C.plane[1][0] + py * C.plane[1][1] > C.plane[1][2]) { cx = (C.center_radius[1][0] - px) * -1.0f; cy = (C.center_radius[1][1] - py) * 1.0f; cz = C.center_radius[1][2]; cw = C.center_radius[1][3]; }
  if (px * C.plane[2][0] + py * C.plane[2][1] > C.plane[2][2]) { cx = px - C.center_radius[2][0]; cy = py - C.center_radius[2][1]; cz = C.center_radius[2][2]; cw = C.center_radius[2][3]; }
  if (px * C.plane[3][0] + py * C.plane[3][1] > C.plane[3][2]) { cx = (C.center_radius[3][0] - px) * 1.0f; cy = (C.center_radius[3][1] - py) * -1.0f; cz = C.center_radius[3][2]; cw = C.center_radius[3][3]; }
  const float prx = cx * cz, pry = cy * cw;
  const float g = (cx * prx + cy * pry) - 1.0f;
  const float dgx = (1.0f + 1.0f) * prx, dgy = (1.0f + 1.0f) * pry;
  const float e = g * (1.0f / sqrtf(dgx * dgx + dgy * dgy));
  const float r = wr_max(wr_max(C.bounds[0] - px, px - C.bounds[2]), wr_max(C.bounds[1] - py, py - C.bounds[3]));
  return wr_max(e, r);
}

// ps_quad_mask fragment (ps_quad.glsl:399-415, ps_quad_mask.glsl:167-200): one pixel of main(), which
// runs four pixels at a time -- fwidth() of the chunk is |lane1 - lane0| in x plus in y (glsl.h:765-768).
__device__ __noinline__ WrWide wr_quad_mask_pixel(const WrPrim* Pp, const WrClipRec* Cp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrClipRec& C = *Cp;
  const WrTexRow r = wr_tex_row(P, D->tex[0], y, runs, x);   // span == 0 for this kind: interpolants only (of the depth run holding x)
  const int n = x - r.x0, n0 = n & ~3;
  float f0x, f0y, f1x, f1y, qx, qy;
  wr_tex_tail_uv(P, r, n0, f0x, f0y);
  wr_tex_tail_uv(P, r, n0 + 1, f1x, f1y);
  wr_tex_tail_uv(P, r, n, qx, qy);
  const float wv = C.w;
  f0x = f0x / wv; f0y = f0y / wv; f1x = f1x / wv; f1y = f1y / wv; qx = qx / wv; qy = qy / wv;   // vClipLocalPos.xy / vClipLocalPos.w
  return wr_quad_mask_eval(C, f0x, f0y, f1x, f1y, qx, qy);
}
// ... from clip_local_pos of the pixel (qx, qy) and of lanes 0 / 1 of its chunk (fwidth) on
WR_DEVICE WrWide wr_quad_mask_eval(const WrClipRec& C, float f0x, float f0y, float f1x, float f1y, float qx, float qy) {
  const float aa_range = 1.0f / (fabsf(f1x - f0x) + fabsf(f1y - f0y));   // recip(fwidth(pos).x), shared.glsl:145-148
  const float dist = wr_clip_dist(C, qx, qy);
  const float alpha = wr_clamp(0.5f - dist * aa_range, 0.0f, 1.0f);
  const float fin = ((1.0f - alpha) - alpha) * C.mode + alpha;
  uint32_t pc[2];
  wr_pack_color(wf4{fin, fin, fin, fin}, pc);
  WrWide s; s.bg = pc[0]; s.ra = pc[1];
  return s;
}

// cs_border_solid main() (cs_border_solid.glsl:132-177; ellipse.glsl:31-45, shared.glsl:110-189), one pixel: vPos of the pixel
// and of lanes 0 / 1 of its 4-pixel chunk (fwidth).  normalize() of the flat colour-line direction is a / hypotf(a.x, a.y)
// (glsl.h:630-640): glibc evaluates hypotf in double.
WR_DEVICE float wr_ellipse_dist(float px, float py, float rx, float ry) {      // distance_to_ellipse
  const float ix = 1.0f / wr_max(rx * rx, 1.0e-6f), iy = 1.0f / wr_max(ry * ry, 1.0e-6f);
  const float scale = (rx > 0.0f && ry > 0.0f) ? 1.0f : 0.0f;
  const float prx = px * ix, pry = py * iy;
  const float g = (px * prx + py * pry) - scale;
  const float dgx = (1.0f + scale) * prx, dgy = (1.0f + scale) * pry;
  return g * (1.0f / sqrtf(dgx * dgx + dgy * dgy));
}
__device__ __noinline__ WrWide wr_border_solid_pixel(const WrPrim* Pp, const WrBorderRec* Bp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrBorderRec& B = *Bp;
  const WrTexRow r = wr_tex_row(P, D->tex[0], y, runs, x);   // span == 0 for this kind: interpolants only
  const int n = x - r.x0, n0 = n & ~3;
  float f0x, f0y, f1x, f1y, qx, qy;
  wr_tex_tail_uv(P, r, n0, f0x, f0y);
  wr_tex_tail_uv(P, r, n0 + 1, f1x, f1y);
  wr_tex_tail_uv(P, r, n, qx, qy);
  const float aa_range = 1.0f / (fabsf(f1x - f0x) + fabsf(f1y - f0y));
  const bool do_aa = B.mix != 2;
  float mix_factor = 0.0f;
  if (B.mix != 0) {
    const float len = float(sqrt(double(B.color_line[2]) * double(B.color_line[2]) + double(B.color_line[3]) * double(B.color_line[3])));
    const float nx = B.color_line[2] / len, ny = B.color_line[3] / len;
    const float d_line = nx * (B.color_line[0] - qx) + ny * (B.color_line[1] - qy);
    if (do_aa) mix_factor = wr_clamp(0.5f - (-d_line) * aa_range, 0.0f, 1.0f);
    else mix_factor = (d_line + 0.0001f >= 0.0f) ? 1.0f : 0.0f;
  }
  float d = -1.0f;
  {
    const float rx = qx - B.clip_center_sign[0], ry = qy - B.clip_center_sign[1];
    if (B.clip_center_sign[2] * rx < 0.0f && B.clip_center_sign[3] * ry < 0.0f) {
      const float da = wr_ellipse_dist(rx, ry, B.clip_radii[0], B.clip_radii[1]), db = wr_ellipse_dist(rx, ry, B.clip_radii[2], B.clip_radii[3]);
      d = wr_max(da, -db);
    }
  }
  {
    const float rx = qx - B.h_center_sign[0], ry = qy - B.h_center_sign[1];
    if (B.h_center_sign[2] * rx < 0.0f && B.h_center_sign[3] * ry < 0.0f) d = wr_max(wr_ellipse_dist(rx, ry, B.h_radii[0], B.h_radii[1]), d);
  }
  {
    const float rx = qx - B.v_center_sign[0], ry = qy - B.v_center_sign[1];
    if (B.v_center_sign[2] * rx < 0.0f && B.v_center_sign[3] * ry < 0.0f) d = wr_max(wr_ellipse_dist(rx, ry, B.v_radii[0], B.v_radii[1]), d);
  }
  const float alpha = do_aa ? wr_clamp(0.5f - d * aa_range, 0.0f, 1.0f) : 1.0f;
  float c[4];
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = ((B.color1[i] - B.color0[i]) * mix_factor + B.color0[i]) * alpha;
  uint32_t pc[2];
  wr_pack_color(wf4{c[0], c[1], c[2], c[3]}, pc);
  WrWide s; s.bg = pc[0]; s.ra = pc[1];
  return s;
}

// cs_border_segment main() (cs_border_segment.glsl:271-449), one pixel
WR_DEVICE float wr_dist_aa(float aa_range, float sd) { return wr_clamp(0.5f - sd * aa_range, 0.0f, 1.0f); }
WR_DEVICE float wr_dist_line(float p0x, float p0y, float dx, float dy, float px, float py) {     // distance_to_line, shared.glsl:110-113
  const float len = wr_hypotf(dx, dy);
  const float nx = dx / len, ny = dy / len;
  return nx * (p0x - px) + ny * (p0y - py);
}
WR_DEVICE void wr_border_corner_color(const WrBorderSegRec& B, float rx, float ry, int style, const float* c0in, const float* c1in,
                                      float mix_factor, float aa_range, float* out) {
  float c0[4] = {c0in[0], c0in[1], c0in[2], c0in[3]};
  if (style == 2) {
    const float da = wr_ellipse_dist(rx, ry, B.clip_radii[0] - B.partial_widths[0], B.clip_radii[1] - B.partial_widths[1]);
    const float db = wr_ellipse_dist(rx, ry, B.clip_radii[0] - 2.0f * B.partial_widths[0], B.clip_radii[1] - 2.0f * B.partial_widths[1]);
    const float a = wr_dist_aa(aa_range, wr_min(-da, db));
    for (int i = 0; i < 4; i++) c0[i] *= a;
  } else if (style == 6 || style == 7) {
    const float d = wr_ellipse_dist(rx, ry, B.clip_radii[0] - B.partial_widths[2], B.clip_radii[1] - B.partial_widths[3]);
    const float alpha = wr_dist_aa(aa_range, d);
    float sf = 0.0f;
    if (B.segment == 1) sf = mix_factor; else if (B.segment == 2) sf = 1.0f; else if (B.segment == 3) sf = 1.0f - mix_factor;
    for (int i = 0; i < 4; i++) {
      const float a0 = (c0in[i] - c1in[i]) * sf + c1in[i];       // mix(color1, color0, sf)
      const float a1 = (c1in[i] - c0in[i]) * sf + c0in[i];       // mix(color0, color1, sf)
      c0[i] = (a1 - a0) * alpha + a0;
    }
  }
  for (int i = 0; i < 4; i++) out[i] = c0[i];
}
WR_DEVICE void wr_border_edge_color(const WrBorderSegRec& B, float px, float py, int style, const float* c0in, const float* c1in,
                                    float aa_range, int axis_id, float* out) {
  const float ax = axis_id != 0 ? 0.0f : 1.0f, ay = axis_id != 0 ? 1.0f : 0.0f;
  const float pos = px * ax + py * ay;
  float c0[4] = {c0in[0], c0in[1], c0in[2], c0in[3]};
  if (style == 2) {
    float d = -1.0f;
    const float pw = B.partial_widths[0] * ax + B.partial_widths[1] * ay;
    if (pw >= 1.0f) {
      const float r0 = (B.edge_reference[0] * ax + B.edge_reference[1] * ay) + pw, r1 = (B.edge_reference[2] * ax + B.edge_reference[3] * ay) - pw;
      d = wr_min(pos - r0, r1 - pos);
    }
    const float a = wr_dist_aa(aa_range, d);
    for (int i = 0; i < 4; i++) c0[i] *= a;
  } else if (style == 6 || style == 7) {
    const float ref = (B.edge_reference[0] + B.partial_widths[2]) * ax + (B.edge_reference[1] + B.partial_widths[3]) * ay;
    const float alpha = wr_dist_aa(aa_range, pos - ref);
    for (int i = 0; i < 4; i++) c0[i] = (c1in[i] - c0in[i]) * alpha + c0in[i];
  }
  for (int i = 0; i < 4; i++) out[i] = c0[i];
}
__device__ __noinline__ WrWide wr_border_segment_pixel(const WrPrim* Pp, const WrBorderSegRec* Bp, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrBorderSegRec& B = *Bp;
  const WrTexRow r = wr_tex_row(P, D->tex[0], y, runs, x);
  const int n = x - r.x0, n0 = n & ~3;
  float f0x, f0y, f1x, f1y, qx, qy;
  wr_tex_tail_uv(P, r, n0, f0x, f0y);
  wr_tex_tail_uv(P, r, n0 + 1, f1x, f1y);
  wr_tex_tail_uv(P, r, n, qx, qy);
  const float aa_range = 1.0f / (fabsf(f1x - f0x) + fabsf(f1y - f0y));
  float mix_factor = 0.0f;
  if (B.edge_axis[0] != B.edge_axis[1]) mix_factor = wr_dist_aa(aa_range, -wr_dist_line(B.color_line[0], B.color_line[1], B.color_line[2], B.color_line[3], qx, qy));
  const float rx = qx - B.clip_center_sign[0], ry = qy - B.clip_center_sign[1];
  const bool in_clip = B.clip_center_sign[2] * rx < 0.0f && B.clip_center_sign[3] * ry < 0.0f;
  float d = -1.0f;
  if (B.clip_mode == 3) {
    const float dx = B.cp1[0] - qx, dy = B.cp1[1] - qy;
    d = sqrtf(dx * dx + dy * dy) - B.cp1[2];
  } else if (B.clip_mode == 2) {
    const bool is_vertical = B.cp1[0] == 0.0f;
    const float half_dash = is_vertical ? B.cp1[1] : B.cp1[0];
    const float pos = is_vertical ? qy : qx;
    if (!(pos < half_dash || pos > 3.0f * half_dash)) d = 1.0f;
  } else if (B.clip_mode == 1) {
    const float d0 = wr_dist_line(B.cp1[0], B.cp1[1], B.cp1[2], B.cp1[3], qx, qy), d1 = wr_dist_line(B.cp2[0], B.cp2[1], B.cp2[2], B.cp2[3], qx, qy);
    d = wr_max(d0, -d1);
  }
  float c0[4], c1[4];
  if (in_clip) {
    const float da = wr_ellipse_dist(rx, ry, B.clip_radii[0], B.clip_radii[1]), db = wr_ellipse_dist(rx, ry, B.clip_radii[2], B.clip_radii[3]);
    d = wr_max(d, wr_max(da, -db));
    wr_border_corner_color(B, rx, ry, B.style0, B.color00, B.color01, mix_factor, aa_range, c0);
    wr_border_corner_color(B, rx, ry, B.style1, B.color10, B.color11, mix_factor, aa_range, c1);
  } else {
    wr_border_edge_color(B, qx, qy, B.style0, B.color00, B.color01, aa_range, B.edge_axis[0], c0);
    wr_border_edge_color(B, qx, qy, B.style1, B.color10, B.color11, aa_range, B.edge_axis[1], c1);
  }
  const float alpha = wr_dist_aa(aa_range, d);
  float c[4];
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = ((c1[i] - c0[i]) * mix_factor + c0[i]) * alpha;
  uint32_t pc[2];
  wr_pack_color(wf4{c[0], c[1], c[2], c[3]}, pc);
  WrWide s; s.bg = pc[0]; s.ra = pc[1];
  return s;
}

// cs_fast_linear_gradient main() (:28-30) and cs_line_decoration main() (:104-163), one pixel
WR_DEVICE float wr_dist_line_v(float p0x, float p0y, float dx, float dy, float px, float py) {   // distance_to_line with a per-pixel direction:
  const float len = sqrtf(dx * dx + dy * dy);                                                    // normalize() = a / sqrt(dot) (glsl.h:628, 637-640)
  const float nx = dx / len, ny = dy / len;
  return nx * (p0x - px) + ny * (p0y - py);
}
__device__ __noinline__ WrWide wr_cache_shader_pixel(const WrPrim* Pp, const WrAux* Ap, const WrDrawDesc* D, int x, int y, const WrRuns* runs = nullptr) {
  const WrPrim& P = *Pp;
  const WrTexRow r = wr_tex_row(P, D->tex[0], y, runs, x);
  const int n = x - r.x0, n0 = n & ~3;
  float qx, qy;
  wr_tex_tail_uv(P, r, n, qx, qy);
  float c[4];
  if (P.kind == WR_PK_FAST_GRADIENT) {
    const WrFastGradRec& G = Ap->fgrad;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = (G.color1[i] - G.color0[i]) * qx + G.color0[i];
  } else {
    const WrLineRec& L = Ap->line;
    float f0x, f0y, f1x, f1y;
    wr_tex_tail_uv(P, r, n0, f0x, f0y);
    wr_tex_tail_uv(P, r, n0 + 1, f1x, f1y);
    const float aa_range = 1.0f / (fabsf(f1x - f0x) + fabsf(f1y - f0y));
    float alpha = 1.0f;
    if (L.style == 2) {
      alpha = (L.params[1] >= floorf(qx + 0.5f)) ? 1.0f : 0.0f;                   // step(floor(pos.x + 0.5), vParams.y)
    } else if (L.style == 1) {
      const float dx = qx - L.params[1], dy = qy - L.params[2];
      alpha = wr_dist_aa(aa_range, sqrtf(dx * dx + dy * dy) - L.params[1]);
    } else if (L.style == 3) {
      const float half_line_thickness = L.params[0], slope_length = L.params[1], flat_length = L.params[2], vertical_bounds = L.params[3];
      const float half_period = slope_length + flat_length;
      const float mid_height = vertical_bounds / 2.0f;
      float peak_offset = mid_height - half_line_thickness;
      const float two_hp = 2.0f * half_period;
      const float m2 = qx - two_hp * floorf(qx / two_hp);                           // mod(pos.x, 2 * half_period)
      const float flip = -2.0f * (((half_period >= m2) ? 1.0f : 0.0f) - 0.5f);      // step(m2, half_period)
      peak_offset *= flip;
      const float peak_height = mid_height + peak_offset;
      const float px_ = qx - half_period * floorf(qx / half_period);                // mod(pos.x, half_period)
      const float dist1 = wr_dist_line_v(0.0f, peak_height, 1.0f, -flip, px_, qy);
      const float dist2 = wr_dist_line_v(0.0f, peak_height, 0.0f, -flip, px_, qy);
      const float dist3 = wr_dist_line_v(flat_length, peak_height, -1.0f, -flip, px_, qy);
      const float dist = fabsf(wr_max(wr_max(dist1, dist2), dist3));
      alpha = wr_dist_aa(aa_range, dist - half_line_thickness);
      if (half_line_thickness <= 1.0f) alpha = 1.0f - ((0.5f >= alpha) ? 1.0f : 0.0f);   // 1 - step(alpha, 0.5)
    }
    c[0] = c[1] = c[2] = c[3] = alpha;
  }
  uint32_t pc[2];
  wr_pack_color(wf4{c[0], c[1], c[2], c[3]}, pc);
  WrWide s; s.bg = pc[0]; s.ra = pc[1];
  return s;
}

struct WrRow4 { uint32_t v[4]; };

// Four horizontally adjacent pixels (x .. x+3) of row y: the span-level setup is
// evaluated once, each pixel then only classifies its chunk.
// Interpolants of one target row at the span start (origin o[], per-pixel step s[]): what
// Edge::nextRow has accumulated after (y - y0) rows.  They are the same for every pixel of the
// row, and a wave's strip has 16 rows, so the raster stage evaluates them once per wave with 16
// row-owning lanes and hands them round with ds_bpermute (wr_apply_prim) instead of once per lane-row.
WR_DEVICE WrRowVals wr_clip_row_vals(const WrPrim& P, int y, const WrAccTabs* tabs = nullptr) {
  WrRowVals rv;
  const int k = y - P.y0;
  const bool lin = P.rows_linear != 0;
  float Lu, Lv, Ru, Rv;
  if (tabs) { Lu = wr_acctabs_row_any(tabs, 0, k); Lv = wr_acctabs_row_any(tabs, 1, k); Ru = wr_acctabs_row_any(tabs, 2, k); Rv = wr_acctabs_row_any(tabs, 3, k); }
  else {
    Lu = wr_row_interp(P.uvL0[0], P.uvLs[0], k, lin); Lv = wr_row_interp(P.uvL0[1], P.uvLs[1], k, lin);
    Ru = wr_row_interp(P.uvR0[0], P.uvRs[0], k, lin); Rv = wr_row_interp(P.uvR0[1], P.uvRs[1], k, lin);
  }
  float stepScale = 1.0f / (P.xr - P.xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  const float start = float(P.x0) + 0.5f - P.xl;
  rv.s[0] = (Ru - Lu) * stepScale; rv.s[1] = (Rv - Lv) * stepScale;
  rv.o[0] = Lu + rv.s[0] * start; rv.o[1] = Lv + rv.s[1] * start;
  rv.s[2] = rv.s[3] = rv.o[2] = rv.o[3] = 0.0f;
  return rv;
}
#ifndef WRHIP_HOSTSIM
// (a row the whole wave works on: the four sums on four lanes at once, see wr_box_row_vals_wave)
WR_DEVICE WrRowVals wr_clip_row_vals_wave(const WrPrim& P, int y, int lane, const WrAccTabs* tabs = nullptr) {
  WrRowVals rv;
  const int k = y - P.y0;
  const bool lin = P.rows_linear != 0;
  const int i = lane & 3;
  const float s0 = i == 0 ? P.uvL0[0] : i == 1 ? P.uvL0[1] : i == 2 ? P.uvR0[0] : P.uvR0[1];
  const float st = i == 0 ? P.uvLs[0] : i == 1 ? P.uvLs[1] : i == 2 ? P.uvRs[0] : P.uvRs[1];
  float r;
  if (tabs) {
    const int m = tabs->mode[i];
    r = wr_acctabs_row(tabs, i, k);
    const float other = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane & 63) - 2) << 2, __builtin_bit_cast(int, r)));
    if (m == 4) r = other;
  } else r = wr_row_interp(s0, st, k, lin);
  const float Lu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0)), Lv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 1));
  const float Ru = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 2)), Rv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 3));
  float stepScale = 1.0f / (P.xr - P.xl);
  if (!wr_isfinite(stepScale)) stepScale = 0.0f;
  const float start = float(P.x0) + 0.5f - P.xl;
  rv.s[0] = (Ru - Lu) * stepScale; rv.s[1] = (Rv - Lv) * stepScale;
  rv.o[0] = Lu + rv.s[0] * start; rv.o[1] = Lv + rv.s[1] * start;
  rv.s[2] = rv.s[3] = rv.o[2] = rv.o[3] = 0.0f;
  return rv;
}
#endif

// Span-level setup of a cs_clip_rectangle row (cs_clip_rectangle.glsl:223-420): the lengths, in 4-pixel chunks, of the
// five phases [clear n1][AA n2][opaque n3][AA n4][clear ...] and the corners the two AA phases belong to.  It depends on
// the prim and the row only, so the raster stage evaluates it once per wave with the 16 row-owning lanes (next to the row
// interpolants) and hands it round; a 4-pixel group in a solid phase then costs a compare and a constant.
struct WrClipRow { float w, aa_range, stx, sty; int n12, n34, corners; };     // n12 = n1 | n2 << 16, n34 likewise, corners = (start + 1) | (end + 1) << 8
WR_DEVICE WrClipRow wr_clip_row_setup(const WrPrim& P, const WrClipRec& C, const WrRowVals& rv) {
  WrClipRow cr;
  const float su = rv.s[0], sv = rv.s[1];
  const float ou = rv.o[0], ov = rv.o[1];
  const int len = P.x1 - P.x0, span = len >= 4 ? (len & ~3) : 0;
  const float lx0 = ou, lx1 = lx0 + su;
  const float ly0 = ov, ly1 = ly0 + sv;
  const float wv = C.w;
  const float w = 1.0f / wv;
  const float p0x = lx0 * w, p0y = ly0 * w, p1x = lx1 * w, p1y = ly1 * w;
  const float stx = (su * 4.0f) * w, sty = (sv * 4.0f) * w;
  const float aa_range = 1.0f / (fabsf(p1x - p0x) + fabsf(p1y - p0y));
  int n1 = 0, n2 = 0, n3 = 0, n4 = 0, start_corner = -1, end_corner = -1;
  if (span > 0 && wv > 0.0f) {
    const float step_scale = wr_max(stx * stx + sty * sty, 1.0e-6f);
    float aa_margin = 1.0f / sqrtf(aa_range * aa_range * step_scale);
    float cr0, cr1, cr2, cr3;
    if (C.fast) { cr0 = -C.params[0] - C.params[2]; cr1 = -C.params[1] - C.params[2]; cr2 = C.params[0] + C.params[2]; cr3 = C.params[1] + C.params[2]; }
    else { cr0 = C.bounds[0]; cr1 = C.bounds[1]; cr2 = C.bounds[2]; cr3 = C.bounds[3]; }
    const bool negx = stx < 0.0f, negy = sty < 0.0f;
    float cd0 = (negx ? cr2 : cr0) - p0x, cd1 = (negy ? cr3 : cr1) - p0y, cd2 = (negx ? cr0 : cr2) - p0x, cd3 = (negy ? cr1 : cr3) - p0y;
    const float rsx = 1.0f / stx, rsy = 1.0f / sty;
    cd0 = stx != 0.0f ? cd0 * rsx : 1.0e6f * wr_step01(0.0f, cd0);
    cd1 = sty != 0.0f ? cd1 * rsy : 1.0e6f * wr_step01(0.0f, cd1);
    cd2 = stx != 0.0f ? cd2 * rsx : 1.0e6f * wr_step01(0.0f, cd2);
    cd3 = sty != 0.0f ? cd3 * rsy : 1.0e6f * wr_step01(0.0f, cd3);
    float opaque_start = wr_max(cd0, cd1), opaque_end = wr_min(cd2, cd3);
    float aa_start = opaque_start, aa_end = opaque_end;
    const float offset = (C.params[0] + C.params[1] + C.params[2]) * C.params[2], z = C.params[2];
#pragma unroll
    for (int c = 0; c < 4; c++) {                    // CLIP_CORNER in TL, TR, BR, BL order
      float pa, pb, pc;
      if (C.fast) { pa = (c == 0 || c == 3) ? -z : z; pb = (c < 2) ? -z : z; pc = offset; }
      else { pa = C.plane[c][0]; pb = C.plane[c][1]; pc = C.plane[c][2]; }
      const float dist = (p0x * pa + p0y * pb) - pc;
      const float scale = -(stx * pa + sty * pb);
      if (scale >= 0.0f) {
        if (dist > opaque_start * scale) {
          start_corner = c;
          const float inv_scale = 1.0f / wr_max(scale, 1.0e-6f);
          opaque_start = dist * inv_scale;
          const float apex = (0.7071f - 0.5f) * 2.0f * fabsf(pa * pb);
          aa_start = opaque_start - apex * inv_scale;
        }
      } else if (dist > opaque_end * scale) {
        end_corner = c;
        const float inv_scale = 1.0f / wr_min(scale, -1.0e-6f);
        opaque_end = dist * inv_scale;
        const float apex = (0.7071f - 0.5f) * 2.0f * fabsf(pa * pb);
        aa_end = opaque_end - apex * inv_scale;
      }
    }
    aa_margin = wr_max(aa_margin - wr_max(aa_start - aa_end, 0.0f), 0.0f);
    aa_start -= aa_margin; aa_end += aa_margin;
    const float sl = float(span), ss = 4.0f;
    const int A = int(wr_clamp(sl - ss * floorf(aa_start), 0.0f, sl)) >> 2, B = int(wr_clamp(sl - ss * ceilf(opaque_start), 0.0f, sl)) >> 2;
    const int Cc = int(wr_clamp(sl - ss * floorf(opaque_end), 0.0f, sl)) >> 2, D = int(wr_clamp(sl - ss * ceilf(aa_end), 0.0f, sl)) >> 2;
    // remaining-length bookkeeping of the five phases, in chunks
    const int S = span >> 2;
    const int R1 = S > A ? A : S;
    n1 = S - R1;
    n2 = R1 > B ? R1 - B : 0;
    const int R2 = R1 - n2;
    n3 = R2 > Cc ? R2 - Cc : 0;
    const int R3 = R2 - n3;
    n4 = R3 > D ? R3 - D : 0;
  }
  cr.w = w; cr.aa_range = aa_range; cr.stx = stx; cr.sty = sty;
  cr.n12 = n1 | (n2 << 16); cr.n34 = n3 | (n4 << 16); cr.corners = (start_corner + 1) | ((end_corner + 1) << 8);
  return cr;
}

// one pixel (n = x - P.x0) of a cs_clip_rectangle row
WR_DEVICE uint32_t wr_clip_rect_px(const WrPrim& P, const WrClipRec& C, const WrRowVals& rv, const WrClipRow& cr, int n) {
  const float su = rv.s[0], sv = rv.s[1];
  const float ou = rv.o[0], ov = rv.o[1];
  const int len = P.x1 - P.x0, span = len >= 4 ? (len & ~3) : 0;
  const float mode = C.mode;
  // the four SIMD lanes of vLocalPos.xy at the span start (init_interp)
  const float lx0 = ou, lx1 = lx0 + su, lx2 = lx1 + su, lx3 = lx2 + su;
  const float ly0 = ov, ly1 = ly0 + sv, ly2 = ly1 + sv, ly3 = ly2 + sv;
  const float wv = C.w;
  const float w = cr.w, aa_range = cr.aa_range, stx = cr.stx, sty = cr.sty;
  const int n1 = cr.n12 & 0xFFFF, n2 = cr.n12 >> 16, n3 = cr.n34 & 0xFFFF, n4 = cr.n34 >> 16;
  const int start_corner = (cr.corners & 0xFF) - 1, end_corner = (cr.corners >> 8) - 1;
  const uint32_t v_clear = uint32_t(wr_round_pixel(mode)) & 0xFFFF, v_opaque = uint32_t(wr_round_pixel(1.0f - mode)) & 0xFFFF;
  if (n < 0 || n >= len) return 0;
  const int lane = n & 3;
  const float lxl = lane == 0 ? lx0 : (lane == 1 ? lx1 : (lane == 2 ? lx2 : lx3));
  const float lyl = lane == 0 ? ly0 : (lane == 1 ? ly1 : (lane == 2 ? ly2 : ly3));
  if (n < span) {
    if (wv <= 0.0f) return 0;     // swgl_commitSolidR8(0.0)
    const int c = n >> 2;
    if (c < n1) return v_clear;
    float qx = lxl * w, qy = lyl * w;
    if (n1 > 0) { qx = qx + float(n1) * stx; qy = qy + float(n1) * sty; }
    int corner;
    if (c < n1 + n2) {
      qx = wr_accum(qx, stx, c - n1); qy = wr_accum(qy, sty, c - n1);
      corner = start_corner;
    } else if (c < n1 + n2 + n3) {
      return v_opaque;
    } else if (c < n1 + n2 + n3 + n4) {
      qx = wr_accum(qx, stx, n2); qy = wr_accum(qy, sty, n2);
      if (n3 > 0) { qx = qx + float(n3) * stx; qy = qy + float(n3) * sty; }
      qx = wr_accum(qx, stx, c - n1 - n2 - n3); qy = wr_accum(qy, sty, c - n1 - n2 - n3);
      corner = end_corner;
    } else {
      return v_clear;
    }
    float dist;
    if (C.fast) {
      dist = wr_clip_dist(C, qx, qy);
    } else {
      dist = wr_max(wr_max(C.bounds[0] - qx, qx - C.bounds[2]), wr_max(C.bounds[1] - qy, qy - C.bounds[3]));
      if (corner >= 0 && qx * C.plane[corner][0] + qy * C.plane[corner][1] > C.plane[corner][2]) {
        const float ex = qx - C.center_radius[corner][0], ey = qy - C.center_radius[corner][1];
        const float prx = ex * C.center_radius[corner][2], pry = ey * C.center_radius[corner][3];
        const float g = (ex * prx + ey * pry) - 1.0f;
        const float dgx = (1.0f + 1.0f) * prx, dgy = (1.0f + 1.0f) * pry;
        dist = g * (1.0f / sqrtf(dgx * dgx + dgy * dgy));
      }
    }
    const float alpha = wr_clamp(0.5f - dist * aa_range, 0.0f, 1.0f);
    return uint32_t(wr_round_pixel(((1.0f - alpha) - alpha) * mode + alpha)) & 0xFFFF;
  }
  // tail chunk (fragment shader): this pixel's lane and lanes 0/1 (fwidth) of its chunk, lanes stepped by `span` at once
  const float chunks = float(span) * 0.25f;
  const float tjx = (su * 4.0f) * chunks, tjy = (sv * 4.0f) * chunks;
  const int m = (n - span) >> 2;
  float t0x = lx0, t0y = ly0, t1x = lx1, t1y = ly1, tlx = lxl, tly = lyl;
  if (span > 0) { t0x = t0x + tjx; t0y = t0y + tjy; t1x = t1x + tjx; t1y = t1y + tjy; tlx = tlx + tjx; tly = tly + tjy; }
  const float sx4 = (su * 4.0f) * 1.0f, sy4 = (sv * 4.0f) * 1.0f;
  const float f0x = wr_accum(t0x, sx4, m) / wv, f0y = wr_accum(t0y, sy4, m) / wv;
  const float f1x = wr_accum(t1x, sx4, m) / wv, f1y = wr_accum(t1y, sy4, m) / wv;
  const float qx = wr_accum(tlx, sx4, m) / wv, qy = wr_accum(tly, sy4, m) / wv;
  const float far = 1.0f / (fabsf(f1x - f0x) + fabsf(f1y - f0y));
  const float dist = wr_clip_dist(C, qx, qy);
  const float alpha = wr_clamp(0.5f - dist * far, 0.0f, 1.0f);
  const float fin = ((1.0f - alpha) - alpha) * mode + alpha;
  return uint32_t(wr_round_pixel(wv > 0.0f ? fin : 0.0f)) & 0xFFFF;
}
__device__ __noinline__ WrRow4 wr_clip_rect_row4(const WrPrim* Pp, const WrClipRec* Cp, WrRowVals rv, WrClipRow cr, int x, int y) {
  WrRow4 out;
#pragma unroll
  for (int i = 0; i < 4; i++) out.v[i] = wr_clip_rect_px(*Pp, *Cp, rv, cr, x + i - Pp->x0);
  return out;
}

// ---------------------------------------------------------------------------
// cs_clip_box_shadow, one destination pixel of an R8 mask: the nine-patch span
// shader (cs_clip_box_shadow.glsl:150-324) replayed up to the pixel's chunk --
// solid lead-in, then [transitional chunk, sector run] pairs whose runs are
// swgl_commitPartialTextureLinear(Invert)R8 / solid centre fills -- or the
// fragment shader (:123-138) for the tail.
WR_DEVICE float wr_r8_texture(const WrTexDesc& t, float u, float v) {   // texture(sColor0, uv).r of an R8 sampler
  const float W = t.sw, H = t.sh;
  if (t.linear) return float(wr_sample_linear_r8(t, int(u * W * 128.0f + (0.5f - 64.0f)), int(v * H * 128.0f + (0.5f - 64.0f)))) * (1.0f / 255.0f);
  return float(((const uint8_t*)t.ptr)[(size_t)wr_clamp_coord(int(u * W), t.width) + (size_t)wr_clamp_coord(int(v * H), t.height) * t.stride]) * (1.0f / 255.0f);
}
WR_DEVICE void wr_box_map_uv(const WrBoxRec& B, float ul, float vl, float& u, float& v) {   // :124-127
  u = wr_clamp(ul, 0.0f, B.edge[0]); v = wr_clamp(vl, 0.0f, B.edge[1]);
  u += wr_max(0.0f, ul - B.edge[2]); v += wr_max(0.0f, vl - B.edge[3]);
  u = (B.uv_noclamp[2] - B.uv_noclamp[0]) * u + B.uv_noclamp[0];
  v = (B.uv_noclamp[3] - B.uv_noclamp[1]) * v + B.uv_noclamp[1];
}
WR_DEVICE float wr_box_shade(const WrBoxRec& B, const WrTexDesc& t, float ul, float vl, float lx, float ly) {
  float u, v;
  wr_box_map_uv(B, ul, vl, u, v);
  u = wr_clamp(u, B.uv_bounds[0], B.uv_bounds[2]); v = wr_clamp(v, B.uv_bounds[1], B.uv_bounds[3]);
  const float in = (wr_step01(B.bounds[0], lx) - wr_step01(B.bounds[2], lx)) * (wr_step01(B.bounds[1], ly) - wr_step01(B.bounds[3], ly));
  const float texel = wr_r8_texture(t, u, v);
  const float alpha = ((1.0f - texel) - texel) * B.mode + texel;
  return (alpha - B.mode) * in + B.mode;
}

WR_DEVICE float wr_sel4(float a0, float a1, float a2, float a3, int i) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }

// Four horizontally adjacent pixels (x .. x+3) of row y: one span-level setup and
// one walk of the nine-patch state machine serve all four.
// row interpolants of a cs_clip_box_shadow prim: c = 0,1 vUv; 2,3 vLocalPos.xy
__device__ __noinline__ WrRow4 wr_box_shadow_row4(const WrPrim* Pp, const WrBoxRec* Bp, WrRowVals rv, WrBoxRow br, int x, int y) {
  const WrPrim& P = *Pp;
  const WrBoxRec& B = *Bp;
  WrRow4 out;
  out.v[0] = out.v[1] = out.v[2] = out.v[3] = 0;
  const WrTexDesc t{B.ptr, int(B.wh & 0xFFFF), int(B.wh >> 16), B.stride, (int16_t)B.format, (int16_t)B.linear, float(B.wh & 0xFFFF), float(B.wh >> 16)};
  float o4[4], s4[4];
#pragma unroll
  for (int c = 0; c < 4; c++) { o4[c] = rv.o[c]; s4[c] = rv.s[c]; }
  const int len = P.x1 - P.x0, span = len >= 4 ? (len & ~3) : 0;
  const int n0 = x - P.x0;
  const float mode = B.mode;
  const uint32_t v_clear = uint32_t(wr_round_pixel(mode)) & 0xFFFF;
  // SIMD lanes at the span start: [component][lane]
  float ln[4][4];
#pragma unroll
  for (int c = 0; c < 4; c++) { ln[c][0] = o4[c]; ln[c][1] = ln[c][0] + s4[c]; ln[c][2] = ln[c][1] + s4[c]; ln[c][3] = ln[c][2] + s4[c]; }
  // ---- tail pixels: fragment shader
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n = n0 + i;
    if (n < span || n >= len || n < 0) continue;
    const int lane = (n - span) & 3, m = (n - span) >> 2;
    float v4[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float a = wr_sel4(ln[c][0], ln[c][1], ln[c][2], ln[c][3], lane);
      if (span > 0) a = a + (s4[c] * 4.0f) * (float(span) * 0.25f);
      v4[c] = wr_accum(a, (s4[c] * 4.0f) * 1.0f, m);
    }
    const float r = wr_box_shade(B, t, v4[0] / B.w, v4[1] / B.w, v4[2] / B.w, v4[3] / B.w);
    out.v[i] = uint32_t(wr_round_pixel(B.w > 0.0f ? r : 0.0f)) & 0xFFFF;
  }
  // ---- span pixels
  const int first = n0 < 0 ? 0 : n0, last = (n0 + 3 < span - 1) ? n0 + 3 : span - 1;    // my pixels inside [0, span)
  if (first > last) return out;
  float w = B.w;
  if (w <= 0.0f) return out;                 // swgl_commitSolidR8(0.0): zeros
  w = 1.0f / w;
  float cur[4][4], st[4];        // uv_linear.x, uv_linear.y, local_pos.x, local_pos.y lanes; per-chunk steps
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int i = 0; i < 4; i++) cur[c][i] = ln[c][i] * w;
    st[c] = (s4[c] * 4.0f) * w;
  }
  const float sl = float(span), ss = 4.0f;
  const int shadow_start_len = br.ss_se & 0xFFFF, shadow_end_len = br.ss_se >> 16;
  const int os0 = br.os01 & 0xFFFF, os1 = br.os01 >> 16, os2 = br.os23 & 0xFFFF, os3 = br.os23 >> 16;
  // everything not claimed by the walk below is the solid lead-in / lead-out
#pragma unroll
  for (int i = 0; i < 4; i++) if (n0 + i >= 0 && n0 + i < span) out.v[i] = v_clear;
  int R = span, pos = 0;
  if (R > shadow_start_len) {
    const int nb = R - shadow_start_len;
    const float f = float(nb / 4);
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int i = 0; i < 4; i++) cur[c][i] += f * st[c];
    }
    R -= nb; pos += nb;
  }
  while (R > 0 && pos <= last) {
    if (pos + 4 > first) {            // transitional chunk holds some of my pixels: per-fragment mapping
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int n = n0 + i;
        if (n < pos || n >= pos + 4 || n >= span) continue;
        const int lane = n & 3;
        out.v[i] = uint32_t(wr_round_pixel(wr_box_shade(B, t, wr_sel4(cur[0][0], cur[0][1], cur[0][2], cur[0][3], lane),
                                                         wr_sel4(cur[1][0], cur[1][1], cur[1][2], cur[1][3], lane),
                                                         wr_sel4(cur[2][0], cur[2][1], cur[2][2], cur[2][3], lane),
                                                         wr_sel4(cur[3][0], cur[3][1], cur[3][2], cur[3][3], lane)))) & 0xFFFF;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int i = 0; i < 4; i++) cur[c][i] += st[c];
    }
    R -= 4; pos += 4;
    if (R <= shadow_end_len) break;
    int num_inside = R - 4 - shadow_end_len;
    float ub0 = B.uv_bounds[0], ub1 = B.uv_bounds[1], ub2 = B.uv_bounds[2], ub3 = B.uv_bounds[3];
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
    }
    if (num_inside > 0) {
      if (pos + num_inside > first && pos <= last) {
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
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int n = n0 + i;
          if (n < pos || n >= pos + num_inside || n >= span) continue;
          const int lane = n & 3;
          if (centre) {
            // centre sector: one texel for the whole run (pattern of the 4 lanes repeated)
            const float texel = wr_r8_texture(t, wr_clamp(wr_sel4(pu[0], pu[1], pu[2], pu[3], lane), ub0, ub2),
                                              wr_clamp(wr_sel4(pv[0], pv[1], pv[2], pv[3], lane), ub1, ub3));
            out.v[i] = uint32_t(wr_round_pixel(((1.0f - texel) - texel) * mode + texel)) & 0xFFFF;
            continue;
          }
          // swgl_commitTextureLinear(R8, sColor0, uv, uv_bounds, NoColor/InvertColor, num_inside)
          const int j = n - pos;
          int v;
          if (filter == 0) {
            // blendTextureNearestFast (swgl_ext.h:475-537)
            const int ix = int(pu[0] * W), iy = int(pv[0] * H);
            const int minUx = int(ub0 * W), minUy = int(ub1 * H), maxUx = int(ub2 * W), maxUy = int(ub3 * H);
            const int srow = wr_clamp_coord(wr_iclamp(iy, minUy, maxUy), t.height);
            const int minX = wr_iclamp(minUx, 0, t.width - 1), maxX = wr_iclamp(maxUx, minX, t.width - 1);
            v = ((const uint8_t*)t.ptr)[(size_t)srow * t.stride + wr_iclamp(ix + j, minX, maxX)];
          } else {
            const float qs = 128.0f, qo = 0.5f - 0.5f * qs;
            float q[4], qy[4];
#pragma unroll
            for (int a = 0; a < 4; a++) { q[a] = pu[a] * W * qs + qo; qy[a] = pv[a] * H * qs + qo; }
            const float stepx = 4.0f * (q[1] - q[0]), stepy = 4.0f * (qy[1] - qy[0]);
            const float minx = wr_max(ub0 * W * qs + qo, 0.0f), miny = wr_max(ub1 * H * qs + qo, 0.0f);
            const float maxx = wr_max(ub2 * W * qs + qo, minx), maxy = wr_max(ub3 * H * qs + qo, miny);
            int o[4];
            wr_linear_span_pixel<1>(t, q, qy, stepx, stepy, minx, maxx, miny, maxy, filter, num_inside, j, o);
            v = o[0];
          }
          if (mode != 0.0f) v = 255 - v;               // applyColor(src, InvertColor)
          out.v[i] = uint32_t(v) & 0xFFFF;
        }
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
  return out;
}

// the one value of the row's u-clamped run (WrBoxRow::xc): stepx is exactly 0 there, so whichever filter the run takes,
// every pixel of it gets what its first pixel gets
WR_DEVICE void wr_box_row_finish(const WrPrim* Pp, const WrBoxRec* Bp, const WrRowVals& rv, WrBoxRow& br, int y) {
  if (br.xc == 0 || rv.s[1] != 0.0f) { br.xc = 0; return; }      // (v must not move along the row either: axis-aligned prims)
  br.vrow = wr_box_shadow_row4(Pp, Bp, rv, br, Pp->x0 + (br.xc & 0xFFFF), y).v[0];
}
