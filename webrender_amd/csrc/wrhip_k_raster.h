// wrhip_k_raster.h -- part of the gfx950 kernels of libwrhip: included by wrhip_kernels.h, in its order, and by nothing else.
// The bin raster: depth runs, the cell raster, wr_raster_body, and every kernel entry point (plain, fused with the next flush's setup stage, thin, dense, chained).
#pragma once

// ---------------------------------------------------------------------------
// Depth runs (draw_depth_span, rasterize.h:612-664).  A depth-tested prim whose pixels depend on where the span
// shader's sub-span starts -- interpolated varyings, 4-pixel chunk phase of an AA ramp -- is drawn by swgl one run of
// passing pixels at a time.  The depth a pixel holds when prim P arrives is min(clear value, z of every earlier
// depth-writing prim covering it) (LEQUAL / LESS only ever lower it), so the runs of a row follow from geometry: P's
// row span minus the row spans of the earlier depth writers with z below P's.  They extend across bins, whose depth
// lives in other workgroups' registers, hence geometry and not the register file.
//   phase 1   the wave scans the target's depth-writing prims that precede P (WrTargetDesc::dw_first / dw_end), 64
//             records per step; the ones that can hide part of P on this strip's rows go to an LDS list (ballot-compacted)
//   phase 2   (candidate, strip row) pairs spread over the lanes: the candidate's interval on that row -> LDS
//   phase 3   16 row-owning lanes sweep their row: the runs [s, e) of pixels no candidate covers -> LDS (WrRuns)
// The pixel evaluators then look their run up (wr_find_run) and restart there.  Returns the strip's 16 WrRuns, or
// nullptr when nothing can hide any part of P here (the common case: one scan, no LDS traffic).
WR_DEVICE bool wr_kind_needs_runs(int kind) {
  return kind == WR_PK_TEX_RGBA8 || kind == WR_PK_TEX_R8 || kind == WR_PK_TEX_FS || kind == WR_PK_GRADIENT || kind == WR_PK_FILTER || kind == WR_PK_MIX_BLEND || kind == WR_PK_SVG_FILTER || kind == WR_PK_YUV || kind == WR_PK_QUAD_MASK || kind == WR_PK_BORDER_SOLID || kind == WR_PK_BORDER_SEGMENT || kind == WR_PK_FAST_GRADIENT || kind == WR_PK_LINE_DECORATION ||
         kind == WR_PK_TEX_REPEAT || kind == WR_PK_TEX_QUAD || kind == WR_PK_SOLID_QUAD || kind == WR_PK_SOLID_AA;
}
// interval of prim `ci` (a depth writer) on row y
WR_DEVICE void wr_occ_interval(const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, int ci, int y, int& lo, int& hi) {
  const WrRec Rc = recs[ci];
  lo = hi = 0;
  if (y < Rc.y0 || y >= Rc.y1) return;
  const int kind = Rc.kbf & 0xFF;
  if (kind == WR_PK_SOLID_QUAD || kind == WR_PK_TEX_QUAD) {
    int s0, s1;
    if (wr_quad_row_span(aux[ci].quad, y, s0, s1) && s1 > s0) { lo = wr_imax(s0, Rc.x0); hi = wr_imin(s1, Rc.x1); if (hi < lo) hi = lo; }
    return;
  }
  lo = Rc.x0; hi = Rc.x1;
}
// phase 3 for one row: [a, b) minus the candidate intervals iv[c] = (lo, hi), c < nc
template <typename IV>
WR_DEVICE int wr_sweep_into(WrRuns& R, int32_t* ext, int a, int b, int nc, IV iv) {
  int n = 0, pos = a;
  while (pos < b) {
    int s = pos;
    for (bool moved = true; moved;) {
      moved = false;
      for (int c = 0; c < nc; c++) { int lo, hi; iv(c, lo, hi); if (lo <= s && s < hi) { s = hi; moved = true; } }
    }
    if (s >= b) break;
    int e = b;
    for (int c = 0; c < nc; c++) { int lo, hi; iv(c, lo, hi); if (hi > lo && lo > s && lo < e) e = lo; }
    if (ext) { ext[2 * n] = s; ext[2 * n + 1] = e; }
    else if (n < WR_MAX_RUNS) { R.s[n] = s; R.e[n] = e; }
    n++;
    pos = e;
  }
  return n;
}
// (a row with more runs than WrRuns holds inline is swept a second time, into the pool; n == -2: the pool is exhausted -- the caller
// reports it and falls back to the span start)
// (`fill` false: a longer row is left at its count, R.n > WR_MAX_RUNS with R.ext == nullptr, for wr_sweep_fill -- the bins' rows, which
// first look whether a neighbouring row of the strip has the same runs)
template <typename IV>
WR_DEVICE void wr_sweep_fill(const WrTargetDesc& T, WrRuns& R, int a, int b, int nc, IV iv) {
  int32_t* ext = wr_pool_words(T, 2ull * (unsigned long long)R.n);
  if (ext) { wr_sweep_into(R, ext, a, b, nc, iv); R.ext = ext; }
  else R.n = -2;
}
template <typename IV>
WR_DEVICE void wr_sweep_runs(const WrTargetDesc& T, WrRuns& R, int a, int b, int nc, IV iv, bool fill = true) {
  R.ext = nullptr; R.pad = 0;
  R.n = wr_sweep_into(R, nullptr, a, b, nc, iv);
  if (R.n > WR_MAX_RUNS && fill) wr_sweep_fill(T, R, a, b, nc, iv);
}
// The same for a target that continues from a materialised depth buffer (a flush in the middle of the target: the prims
// that wrote it are gone): pixel by pixel, a pixel passes when it passes against the loaded depth AND no candidate of this
// flush covers it.  A foreign call pattern (WebRender never flushes mid-target with depth live); kept simple, not fast.
template <typename IV>
WR_DEVICE void wr_scan_runs(const WrTargetDesc& T, WrRuns& R, int a, int b, int nc, IV iv, const uint32_t* __restrict__ drow, uint32_t z, bool less) {
  auto pass = [&](int px) {
    if (!(less ? z < drow[px] : z <= drow[px])) return false;
    for (int c = 0; c < nc; c++) { int lo, hi; iv(c, lo, hi); if (lo <= px && px < hi) return false; }
    return true;
  };
  auto scan = [&](int32_t* ext) -> int {
    int n = 0, x = a;
    while (x < b) {
      while (x < b && !pass(x)) x++;
      if (x >= b) break;
      const int s0 = x;
      while (x < b && pass(x)) x++;
      if (ext) { ext[2 * n] = s0; ext[2 * n + 1] = x; }
      else if (n < WR_MAX_RUNS) { R.s[n] = s0; R.e[n] = x; }
      n++;
    }
    return n;
  };
  R.ext = nullptr; R.pad = 0;
  int n = scan(nullptr);
  if (n > WR_MAX_RUNS) {
    int32_t* ext = wr_pool_words(T, 2ull * (unsigned long long)n);
    if (ext) { scan(ext); R.ext = ext; }
    else n = -2;                       // (the pool is exhausted: the caller reports it and falls back to the span start)
  }
  R.n = n;
}
template <int R4>
WR_DEVICE const WrRuns* wr_build_runs(const WrTargetDesc& T, const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, int pidx,
                                      int x0, int y0, int x1, int y1, uint32_t z, uint32_t kbf, int wy0, int lane, int wave,
                                      const unsigned long long* __restrict__ bin_words = nullptr, int bin_x0 = 0) {
  const int kind = kbf & 0xFF, flags = (kbf >> 16) & 0xFF;
  const bool less = (flags & WR_PF_DEPTH_LESS) != 0;
  const int end = wr_imin(T.dw_end, pidx);
  const int ry0 = wr_imax(y0, wy0), ry1 = wr_imin(y1, wy0 + 4 * R4);
  const bool quad = kind == WR_PK_TEX_QUAD || kind == WR_PK_SOLID_QUAD;
  const bool loaded = T.load_depth && T.depth;       // continuation of a target whose depth was materialised
#ifdef WRHIP_HOSTSIM
  // serial restatement: this thread does the whole wave's work for its own rows
  static int cidx_lds[WR_MAX_OCC];
  static WrRuns runs[4 * R4];
  const int* cidx = cidx_lds;
  int nc = 0;
  auto scan = [&](int* list, int cap) {
    nc = 0;
    for (int i = T.dw_first; i < end; i++) {
      const WrRec Rc = recs[i];
      const int ok = Rc.kbf & 0xFF, of = (Rc.kbf >> 16) & 0xFF;
      if (!(of & WR_PF_DEPTH_WRITE) || ok == WR_PK_NONE || ok == WR_PK_UNSUPPORTED || ok == WR_PK_CLEAR) continue;
      if (!(less ? Rc.z <= z : Rc.z < z)) continue;
      if (Rc.x0 >= x1 || Rc.x1 <= x0 || Rc.y0 >= ry1 || Rc.y1 <= ry0) continue;
      if (nc < cap) list[nc] = i;
      nc++;
    }
  };
  scan(cidx_lds, WR_MAX_OCC);
  bool anyflat = false;
  if (T.flat_rows) for (int y = ry0; y < ry1; y++) if (T.flat_rows[y] < (uint32_t)pidx) anyflat = true;
  if (nc == 0 && !loaded && !anyflat) return nullptr;
  if (nc > WR_MAX_OCC) {          // more occluders than the LDS list holds: the list goes to the pool (a second scan fills it)
    int* big = wr_pool_words(T, (unsigned long long)nc);
    if (big) { scan(big, nc); cidx = big; }
    else {                        // (the pool is exhausted: evaluated from the span start, as if unoccluded -- and reported)
      if (T.counters) atomicAdd(&T.counters->unsupported_prims, 1u);
      if (!anyflat) return nullptr;
      nc = 0;
    }
  }
  for (int j = 0; j < R4; j++) {
    const int r = (lane >> 4) + 4 * j, y = wy0 + r;
    WrRuns& RR = runs[r];
    RR.n = 0; RR.ext = nullptr; RR.pad = 0;
    if (y < ry0 || y >= ry1) continue;
    if (T.flat_rows && T.flat_rows[y] < (uint32_t)pidx) { RR.n = -1; continue; }
    int a = x0, b = x1;
    if (quad) { int s0, s1; if (!wr_quad_row_span(aux[pidx].quad, y, s0, s1)) continue; a = wr_imax(s0, x0); b = wr_imin(s1, x1); }
    auto iv = [&](int c, int& lo, int& hi) { wr_occ_interval(recs, aux, cidx[c], y, lo, hi); };
    if (loaded) wr_scan_runs(T, RR, a, b, nc, iv, T.depth + (size_t)y * T.width, z, less);
    else wr_sweep_runs(T, RR, a, b, nc, iv);
    if (RR.n == -2) { RR.n = 0; if (T.counters) atomicAdd(&T.counters->unsupported_prims, 1u); }
  }
  return runs;
#else
  __shared__ int cidx[4][WR_MAX_OCC];
  __shared__ short ivs[4][WR_MAX_OCC][4 * R4][2];
  __shared__ WrRuns runs[4][4 * R4];
  int nc = 0;
  int* ext = nullptr;               // a strip with more candidates than the wave's LDS list holds: its list in the pool (second scan)
  auto test = [&](int i, bool in) {
    bool hit = false;
    if (in) {
      const uint4* rp = (const uint4*)&recs[i];
      const uint4 ra = rp[0], rb = rp[1];
      const int ok = rb.y & 0xFF, of = (rb.y >> 16) & 0xFF;
      hit = (of & WR_PF_DEPTH_WRITE) && ok != WR_PK_NONE && ok != WR_PK_UNSUPPORTED && ok != WR_PK_CLEAR && (less ? rb.x <= z : rb.x < z) &&
            (int)ra.x < x1 && (int)ra.z > x0 && (int)ra.y < ry1 && (int)ra.w > ry0;
    }
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int slot = nc + __popcll(m & ((1ull << lane) - 1ull));
      if (ext) ext[slot] = i;
      else if (slot < WR_MAX_OCC) cidx[wave][slot] = i;
    }
    nc += __popcll(m);
  };
  auto scan_all = [&]() {
  nc = 0;
  if (bin_words && x0 >= bin_x0 && x1 <= bin_x0 + WR_BIN_W && end - T.dw_first > 256) {
    // A prim that lies inside this bin's columns: every depth writer that can cut its rows touches the bin too, so the
    // candidates are among the bin's own mask words (intact until the workgroup's last wave is done) -- a handful of words
    // instead of every depth writer of the target before it (many-images.yaml: 8192 opaque 8x8 images in one tile, 128
    // fetches per prim and wave: 727 us for the tile).  Same candidates in the same (submission) order.
    const int w_lo = (T.dw_first - T.first_prim) >> 6, w_hi = (end - 1 - T.first_prim) >> 6;
    for (int wb = w_lo; wb <= w_hi; wb += 64) {
      const unsigned long long mv = wb + lane <= w_hi ? bin_words[wb + lane] : 0ull;
      for (unsigned long long nz = __ballot(mv != 0ull); nz; nz &= nz - 1ull) {
        const int cw = __builtin_ctzll(nz);
        const unsigned long long m_ = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv, cw) |
                                      ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mv >> 32), cw) << 32);
        const int i = T.first_prim + (wb + cw) * 64 + lane;
        test(i, ((m_ >> lane) & 1ull) && i >= T.dw_first && i < end);
      }
    }
  } else {
    for (int b = T.dw_first; b < end; b += 64) test(b + lane, b + lane < end);
  }
  };
  scan_all();
  // rows of the strip that an earlier perspective prim has flattened (lane r looks at strip row r)
  bool myflat = false;
  if (T.flat_rows && lane < 4 * R4 && wy0 + lane >= ry0 && wy0 + lane < ry1) myflat = T.flat_rows[wy0 + lane] < (uint32_t)pidx;
  const bool anyflat = __ballot(myflat) != 0ull;
  if (nc == 0 && !loaded && !anyflat) return nullptr;
  bool big = false;                 // the candidates sit in the pool and their intervals are taken from the records as the sweeps ask for them
  if (nc > WR_MAX_OCC) {          // more occluders than the LDS list holds: the list goes to the pool, a second scan fills it
    ext = wr_pool_words_wave(T, (unsigned long long)nc);
    if (ext) {
      big = true;
      scan_all();
    } else {                        // (the pool is exhausted: evaluated from the span start, as if unoccluded -- and reported)
      if (lane == 0 && T.counters) atomicAdd(&T.counters->unsupported_prims, 1u);
      if (!anyflat) return nullptr;
      nc = 0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (!big) {
    for (int idx = lane; idx < nc * 4 * R4; idx += 64) {
      const int c = idx / (4 * R4), r = idx - c * (4 * R4);
      int lo, hi;
      wr_occ_interval(recs, aux, cidx[wave][c], wy0 + r, lo, hi);
      ivs[wave][c][r][0] = (short)lo; ivs[wave][c][r][1] = (short)hi;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 4 * R4) {
    const int r = lane, y = wy0 + r;
    WrRuns& RR = runs[wave][r];
    RR.n = 0; RR.ext = nullptr; RR.pad = 0;
    if (myflat) RR.n = -1;
    else if (y >= ry0 && y < ry1) {
      int a = x0, b = x1;
      bool ok = true;
      if (quad) { int s0, s1; ok = wr_quad_row_span(aux[pidx].quad, y, s0, s1); a = wr_imax(s0, x0); b = wr_imin(s1, x1); }
      auto iv = [&](int c, int& lo, int& hi) { lo = ivs[wave][c][r][0]; hi = ivs[wave][c][r][1]; };
      auto iv_big = [&](int c, int& lo, int& hi) { wr_occ_interval(recs, aux, ext[c], y, lo, hi); };
      if (!ok) {}
      else if (big && loaded) wr_scan_runs(T, RR, a, b, nc, iv_big, T.depth + (size_t)y * T.width, z, less);
      else if (big) wr_sweep_runs(T, RR, a, b, nc, iv_big, quad);
      else if (loaded) wr_scan_runs(T, RR, a, b, nc, iv, T.depth + (size_t)y * T.width, z, less);
      else wr_sweep_runs(T, RR, a, b, nc, iv, quad);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // Rows with more runs than the LDS copy holds (left at their count, ext == nullptr): a row whose occluders cover it exactly as they
  // cover the row above has the same runs -- behind a grid of axis-aligned rects that is fifteen of a strip's sixteen rows -- and
  // points at that row's words of the pool instead of taking its own (the pool is what such a frame runs out of).
  if (lane < 4 * R4) {
    const int r = lane, y = wy0 + r;
    WrRuns& RR = runs[wave][r];
    const bool over = RR.n > WR_MAX_RUNS && RR.ext == nullptr;
    bool same = false;
    if (over && r > 0 && runs[wave][r - 1].n == RR.n && y - 1 >= ry0 && !(T.qtab_pad & 2u)) {
      same = true;
      for (int c = 0; c < nc && same; c++) {
        int lo0, hi0, lo1, hi1;
        if (big) { wr_occ_interval(recs, aux, ext[c], y, lo0, hi0); wr_occ_interval(recs, aux, ext[c], y - 1, lo1, hi1); }
        else { lo0 = ivs[wave][c][r][0]; hi0 = ivs[wave][c][r][1]; lo1 = ivs[wave][c][r - 1][0]; hi1 = ivs[wave][c][r - 1][1]; }
        same = lo0 == lo1 && hi0 == hi1;
      }
    }
    const unsigned long long F = __ballot(same);
    if (over && !same) {
      auto iv = [&](int c, int& lo, int& hi) { lo = ivs[wave][c][r][0]; hi = ivs[wave][c][r][1]; };
      auto iv_big = [&](int c, int& lo, int& hi) { wr_occ_interval(recs, aux, ext[c], y, lo, hi); };
      if (big) wr_sweep_fill(T, RR, x0, x1, nc, iv_big); else wr_sweep_fill(T, RR, x0, x1, nc, iv);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (same) {
      const unsigned long long leaders = ~F & ((1ull << r) - 1ull);        // (row 0 is never a follower: there is one)
      const WrRuns& LR = runs[wave][63 - __builtin_clzll(leaders)];
      RR.n = LR.n; RR.ext = LR.ext;
    }
    if (RR.n == -2) { RR.n = 0; RR.ext = nullptr; if (T.counters) atomicAdd(&T.counters->unsupported_prims, 1u); }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return runs[wave];
#endif
}

// Pixels are held as two registers of 2 x 16-bit fields: lo = (B, R), hi = (G, A)
// of the BGRA8 texel -- i.e. WideRGBA8 with the channels paired so that one
// 32-bit multiply serves two channels (fields never carry into each other:
// 255 * 256 < 2^16).
#define WR_M8 0x00FF00FFu

// Apply one prim to the 4*R pixels of this lane (4 wide x R rows, rows 4 apart).
// All prim parameters are wave-uniform (SGPRs); (px,py) is the lane's first
// pixel, (wx0,wy0) the wave's 64 x 4R strip origin.
// Is the strip [wx0, wx0 + 64) x [wy0, wy0 + rows) inside the part of a (2-D) general quad where every pixel is covered
// completely?  The rows must belong to one run of the walk (one left and one right edge, straight lines), the run's clip
// span must contain the strip's columns, and both edges must stay clear of the columns by their anti-aliasing reach
// (aa_edge rounds out by |slope| / 2, aa_dist reaches full coverage within sqrt(1 + slope^2) / 2 <= (|slope| + 1) / 2 of
// the edge) plus two pixels for the difference between this straight-line estimate and the row-by-row sums.  Such a strip
// takes the flat-colour path: coverage 256 leaves the colour as it is (DO_AA, blend.h:433-446).
WR_DEVICE bool wr_strip_inside_quad(const WrQuadRec& Q, int wx0, int wy0, int rows) {
  int si = -1;
  for (int i = 0; i < Q.nseg; i++) if (wy0 >= Q.seg[i].row_a && wy0 + rows <= Q.seg[i].row_b) si = i;
  if (si < 0) return false;
  const WrQuadSeg& S = Q.seg[si];
  const float ya = float(wy0 - S.lrow), yb = float(wy0 + rows - 1 - S.lrow);
  const float la = S.lx + S.ls * ya, lb = S.lx + S.ls * yb;
  const float yc = float(wy0 - S.rrow), yd = float(wy0 + rows - 1 - S.rrow);
  const float ra = S.rx + S.rs * yc, rb = S.rx + S.rs * yd;
  const float lmax = wr_max(la, lb) + 0.5f * fabsf(S.ls) + 2.5f, rmin = wr_min(ra, rb) - 0.5f * fabsf(S.rs) - 2.5f;
  return lmax <= float(wx0) && rmin >= float(wx0 + WR_BIN_W) && S.b0 <= float(wx0) && S.b1 >= float(wx0 + WR_BIN_W);
}

template <int FMT, bool DEPTH, int R, int FEAT>
WR_DEVICE void wr_apply_prim(uint32_t (&plo)[4 * R], uint32_t (&phi)[4 * R], uint32_t (&dep)[4 * R],
                             const int x0, const int y0, const int x1, const int y1, const uint32_t z,
                             const uint32_t kbf, const uint32_t c0, const uint32_t c1,
                             const WrPrim* Pp, const WrAux* Ap, const WrDrawDesc* draws, const float* __restrict__ vtab,
                             const int px, const int py, const int wx0, const int wy0, const WrRuns* rr = nullptr) {
  constexpr int BPP = FMT == WR_FMT_RGBA8 ? 4 : 1;
  constexpr int NPX = 4 * R;
  int kind = kbf & 0xFF;
  const int blend = (kbf >> 8) & 0xFF, flags = (kbf >> 16) & 0xFF;
  // a strip in the fully covered interior of a rotated / skewed solid quad is a flat-colour strip
  // (premultiplied blend of an ordinary colour: the branch below; no blend: the colour itself; other keys keep the per-pixel path)
  bool flat_copy = false;
  if ((FEAT & WR_FEAT_GENERIC) && FMT == WR_FMT_RGBA8 && kind == WR_PK_SOLID_QUAD && Ap->quad.pad == 0 &&
      ((blend == WR_BLEND_PREMULT && ((c0 | c1) & 0xFF00FF00u) == 0) || blend == WR_BLEND_NONE) &&
      wr_strip_inside_quad(Ap->quad, wx0, wy0, 4 * R)) {
    kind = WR_PK_SOLID;
    flat_copy = blend == WR_BLEND_NONE;
  }
  const bool dtest = DEPTH && (flags & WR_PF_DEPTH_TEST);
  const bool dwrite = (flags & WR_PF_DEPTH_WRITE) != 0, dless = (flags & WR_PF_DEPTH_LESS) != 0;
  // does the prim cover this wave's whole strip?  (uniform)
  const bool full = x0 <= wx0 && x1 >= wx0 + WR_BIN_W && y0 <= wy0 && y1 >= wy0 + 4 * R;

  if (FMT == WR_FMT_RGBA8 && kind == WR_PK_SOLID_FOLDED) {
    // ---- hot path: swgl_commitSolidRGBA8, no blend or premultiplied alpha ----
    //   new = hi_bytes(dst * K + C) per channel pair; K, C prepared by wr_make_rec.
    const uint32_t K = kbf >> 24, Clo = c0, Chi = c1;
    if (full && !dtest) {
#pragma unroll
      for (int q = 0; q < NPX; q++) {
        wr_fold_inplace(plo[q], K, Clo);
        wr_fold_inplace(phi[q], K, Chi);
      }
      return;
    }
    wr_lanemask mx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) mx[i] = WR_LANEMASK((unsigned)(px + i - x0) < (unsigned)(x1 - x0));
#pragma unroll
    for (int j = 0; j < R; j++) {
      const wr_lanemask myj = WR_LANEMASK((unsigned)(py + 4 * j - y0) < (unsigned)(y1 - y0));
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        wr_lanemask in = mx[i] & myj;
        if (dtest) {
          in = in & WR_LANEMASK(dless ? (z < dep[q]) : (z <= dep[q]));
          if (dwrite) wr_select_masked(dep[q], z, in);
        }
        wr_fold_masked(plo[q], phi[q], K, Clo, Chi, in);
      }
    }
    return;
  }

  // per-lane coverage
  bool cx[4], cy[R];
#pragma unroll
  for (int i = 0; i < 4; i++) cx[i] = (unsigned)(px + i - x0) < (unsigned)(x1 - x0);
#pragma unroll
  for (int j = 0; j < R; j++) cy[j] = (unsigned)(py + 4 * j - y0) < (unsigned)(y1 - y0);

  if ((FEAT & WR_FEAT_BLUR) && FMT == WR_FMT_R8 && kind == WR_PK_MASK_ROWS) {
    // a cs_clip_* prim whose rows wr_mask_rows_kernel has evaluated: blend the stored bytes (c0/c1: address of the prim's
    // first row at column x0 & ~3, z: pitch)
    const uint8_t* base = (const uint8_t*)(((unsigned long long)c1 << 32) | c0);
    const bool anyx = cx[0] || cx[1] || cx[2] || cx[3];
    const int col = px - (x0 & ~3);
#pragma unroll
    for (int j = 0; j < R; j++) {
      if (!cy[j] || !anyx) continue;
      // (the row map: rows that came out identical to another row of the prim point at that row's bytes)
      const uint32_t ro = ((const uint32_t*)base)[py + 4 * j - y0];
      const uint32_t v = *(const uint32_t*)(base + ro + col);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        if (cx[i]) plo[q] = wr_blend_r8(blend, plo[q], (v >> (8 * i)) & 0xFF);
      }
    }
    return;
  }
  if (kind == WR_PK_CLEAR) {
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      const bool in = cx[q & 3] && cy[q >> 2];
      if (flags & WR_PF_CLEAR_COLOR) {
        uint32_t nlo = BPP == 4 ? (c0 & WR_M8) : c0, nhi = BPP == 4 ? ((c0 >> 8) & WR_M8) : 0;
        plo[q] = in ? nlo : plo[q];
        phi[q] = in ? nhi : phi[q];
      }
      if (DEPTH && (flags & WR_PF_CLEAR_DEPTH)) dep[q] = in ? z : dep[q];
    }
    return;
  }

  if ((FEAT & WR_FEAT_GENERIC) && FMT == WR_FMT_RGBA8 && flat_copy) {
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      bool in = cx[q & 3] && cy[q >> 2];
      if (dtest) {
        const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
        in = in && pass;
        if (dwrite) dep[q] = in ? z : dep[q];
      }
      plo[q] = in ? (c0 & 0xFFFF) | ((c1 & 0xFFFF) << 16) : plo[q];
      phi[q] = in ? (c0 >> 16) | (c1 & 0xFFFF0000u) : phi[q];
    }
    return;
  }
  if (FMT == WR_FMT_RGBA8 && kind == WR_PK_SOLID && blend == WR_BLEND_PREMULT && ((c0 | c1) & 0xFF00FF00u) == 0) {
    // premultiplied blend of a colour whose channels exceed its alpha: same
    // formula, but the sum can pass 255 and needs pack()'s clamp
    const uint32_t K = 255u - (c1 >> 16);
    const uint32_t ulo = (c0 & 0xFFFF) | ((c1 & 0xFFFF) << 16), uhi = (c0 >> 16) | (c1 & 0xFFFF0000u);
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      bool in = cx[q & 3] && cy[q >> 2];
      if (dtest) {
        const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
        in = in && pass;
        if (dwrite) dep[q] = in ? z : dep[q];
      }
      const uint32_t nl = wr_pk_min_u16(wr_hi_bytes(wr_mul24(plo[q], K) + WR_M8) + ulo, WR_M8);
      const uint32_t nh = wr_pk_min_u16(wr_hi_bytes(wr_mul24(phi[q], K) + WR_M8) + uhi, WR_M8);
      plo[q] = in ? nl : plo[q]; phi[q] = in ? nh : phi[q];
    }
    return;
  }
  if ((FEAT & WR_FEAT_TEX) && FMT == WR_FMT_RGBA8 && kind == WR_PK_TEX_RGBA8 && (blend == WR_BLEND_NONE || blend == WR_BLEND_PREMULT) &&
      !(flags & WR_PF_MASKED) && Ap->tex.simple >= 2 && !rr) {
    // ---- swgl_commitTexture*RGBA8, nearest-fast rows (blendTextureNearestFast,
    // swgl_ext.h:475-537): the source row of every target row was resolved by
    // the setup kernel (unit rows) or is evaluated per lane-row; a lane fetches its 4 texels of each row.
    // Rows / pixels outside that case go through the out-of-line generic path.
    const WrTexRec& T = Ap->tex;
    const uint32_t* sbuf = (const uint32_t*)T.ptr;
    const int n0 = px - x0;
    const bool has_color = (flags & WR_PF_HAS_COLOR) != 0;
    const bool inspan = n0 >= 0 && n0 + 4 <= T.span;
    const int xa = T.ix0 + n0;
    // Plain copy of a strip the prim covers completely (tile composites: unit rows, no blend, no
    // colour, no column clamp in reach, 16-byte aligned columns): one 16-byte load per lane and
    // row, two VALU instructions per pixel.  Every condition is wave-uniform.
    {
      const int xw = T.ix0 + (wx0 - x0);        // source column of the strip's first pixel
      if (full && !dtest && blend == WR_BLEND_NONE && !has_color && T.simple == 3 && wx0 - x0 >= 0 &&
          wx0 - x0 + WR_BIN_W <= T.span && xw >= T.tix[0] && xw + WR_BIN_W - 1 <= T.tix[1] && (xw & 3) == 0 &&
          (T.stride & 3) == 0 && (((uintptr_t)T.ptr) & 15) == 0) {
#pragma unroll
        for (int j = 0; j < R; j++) {
          const int y = py + 4 * j;
          const int srow = wr_iclamp(T.iy0 + T.tix[2] * (y - T.y0), T.unit & 0xFFFF, T.unit >> 16);
#ifdef WRHIP_HOSTSIM
          uint32_t v[4];
          __builtin_memcpy(v, sbuf + (size_t)srow * T.stride + xa, 16);
#else
          const uint4 vv = *(const uint4*)(sbuf + (size_t)srow * T.stride + xa);
          const uint32_t v[4] = {vv.x, vv.y, vv.z, vv.w};
#endif
#pragma unroll
          for (int i = 0; i < 4; i++) { plo[4 * j + i] = v[i] & WR_M8; phi[4 * j + i] = (v[i] >> 8) & WR_M8; }
        }
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
      const int y = py + 4 * j;
      uint32_t sp[4] = {0, 0, 0, 0};
      bool rowfast = false;
      if (cy[j] && inspan) {
        const int srow = T.simple == 3 ? wr_iclamp(T.iy0 + T.tix[2] * (y - T.y0), T.unit & 0xFFFF, T.unit >> 16)
                                        : wr_texrow_entry(T, vtab[T.iy0 + (y - T.y0)]);
        rowfast = srow >= 0;
        if (rowfast) {
          const uint32_t* rp = sbuf + (size_t)srow * T.stride;
#pragma unroll
          for (int i = 0; i < 4; i++) sp[i] = rp[wr_iclamp(xa + i, T.tix[0], T.tix[1])];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        bool in = cx[i] && cy[j];
        if (dtest) {
          const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
          in = in && pass;
          if (dwrite) dep[q] = in ? z : dep[q];
        }
        if (rowfast) {
          uint32_t sl = sp[i] & WR_M8, sh = (sp[i] >> 8) & WR_M8;
          if (has_color) {  // applyColor: muldiv255(colour, src) per channel
            const uint32_t cb = c0 & 0xFFFF, cg = c0 >> 16, cr = c1 & 0xFFFF, ca = c1 >> 16;
            const uint32_t sb = sl & 0xFFFF, sr = sl >> 16, sg = sh & 0xFFFF, sa = sh >> 16;
            sl = (((cb * sb + cb) & 0xFFFF) >> 8) | ((((cr * sr + cr) & 0xFFFF) >> 8) << 16);
            sh = (((cg * sg + cg) & 0xFFFF) >> 8) | ((((ca * sa + ca) & 0xFFFF) >> 8) << 16);
          }
          uint32_t nl = sl, nh = sh;
          if (blend == WR_BLEND_PREMULT) {
            const uint32_t K = 255u - (sh >> 16);
            nl = wr_pk_min_u16(wr_hi_bytes(wr_mul24(plo[q], K) + WR_M8) + sl, WR_M8);
            nh = wr_pk_min_u16(wr_hi_bytes(wr_mul24(phi[q], K) + WR_M8) + sh, WR_M8);
          }
          plo[q] = in ? nl : plo[q]; phi[q] = in ? nh : phi[q];
        } else if (in) {
          if (FEAT & WR_FEAT_GENERIC) {
            uint32_t r = wr_generic_pixel_rgba8(Pp, &draws[Pp->draw], px + i, y, plo[q] | (phi[q] << 8));
            plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
          }
        }
      }
    }
    return;
  }
  if ((FEAT & WR_FEAT_CLIP) && FMT == WR_FMT_R8 && kind == WR_PK_BOX_SHADOW) {
    const bool anyx = cx[0] || cx[1] || cx[2] || cx[3];
#ifndef WRHIP_HOSTSIM
    const WrRowVals mine = wr_box_row_vals(*Pp, Ap->box, wy0 + ((px - wx0) >> 2));   // lane (l & 15) owns strip row (l & 15)
    WrBoxRow mine_br = wr_box_row_setup(*Pp, Ap->box, mine);
    wr_box_row_finish(Pp, &Ap->box, mine, mine_br, wy0 + ((px - wx0) >> 2));
#endif
#pragma unroll
    for (int j = 0; j < R; j++) {
#ifdef WRHIP_HOSTSIM
      const WrRowVals rv = wr_box_row_vals(*Pp, Ap->box, py + 4 * j);
      WrBoxRow br = wr_box_row_setup(*Pp, Ap->box, rv);
      wr_box_row_finish(Pp, &Ap->box, rv, br, py + 4 * j);
#else
      WrRowVals rv;
      WrBoxRow br;
      const int src = (py - wy0) + 4 * j;      // the lane that evaluated this lane's row
#pragma unroll
      for (int c = 0; c < 4; c++) { rv.o[c] = __shfl(mine.o[c], src); rv.s[c] = __shfl(mine.s[c], src); }
      br.ss_se = __shfl(mine_br.ss_se, src); br.os01 = __shfl(mine_br.os01, src); br.os23 = __shfl(mine_br.os23, src);
      br.xc = __shfl(mine_br.xc, src); br.vrow = __shfl(mine_br.vrow, src);
#endif
      if (!cy[j] || !anyx) continue;
      WrRow4 r4;
      {
        // the solid lead-in (before the shadow rect starts) and lead-out (after it ends) of the row need no evaluation
        const int n = px - x0, len = x1 - x0, span = len >= 4 ? (len & ~3) : 0;
        const int lead_in = span - (br.ss_se & 0xFFFF), lead_out = span - (br.ss_se >> 16);
        if (n >= 0 && n + 3 < span && Ap->box.w > 0.0f && (n + 3 < lead_in || n >= wr_imax(lead_out, lead_in) + 4)) {      // (the chunk at lead_in is always evaluated)
          const uint32_t v = uint32_t(wr_round_pixel(Ap->box.mode)) & 0xFFFF;
          r4.v[0] = r4.v[1] = r4.v[2] = r4.v[3] = v;
        } else if (br.xc != 0 && n >= (br.xc & 0xFFFF) && n + 4 <= (br.xc >> 16) && n + 3 < span) {
          r4.v[0] = r4.v[1] = r4.v[2] = r4.v[3] = br.vrow;          // inside the u-clamped run: the row's one value
        } else {
          r4 = wr_box_shadow_row4(Pp, &Ap->box, rv, br, px, py + 4 * j);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        if (cx[i]) plo[q] = wr_blend_r8(blend, plo[q], r4.v[i]);
      }
    }
    return;
  }
  if ((FEAT & WR_FEAT_CLIP) && FMT == WR_FMT_R8 && kind == WR_PK_CLIP_RECT) {
    const bool anyx = cx[0] || cx[1] || cx[2] || cx[3];
#ifndef WRHIP_HOSTSIM
    const WrRowVals mine = wr_clip_row_vals(*Pp, wy0 + ((px - wx0) >> 2));
    const WrClipRow mine_cr = wr_clip_row_setup(*Pp, Ap->clip, mine);
#endif
#pragma unroll
    for (int j = 0; j < R; j++) {
#ifdef WRHIP_HOSTSIM
      const WrRowVals rv = wr_clip_row_vals(*Pp, py + 4 * j);
      const WrClipRow cr = wr_clip_row_setup(*Pp, Ap->clip, rv);
#else
      WrRowVals rv;
      WrClipRow cr;
      const int src = (py - wy0) + 4 * j;
#pragma unroll
      for (int c = 0; c < 2; c++) { rv.o[c] = __shfl(mine.o[c], src); rv.s[c] = __shfl(mine.s[c], src); }
      rv.o[2] = rv.o[3] = rv.s[2] = rv.s[3] = 0.0f;
      cr.w = __shfl(mine_cr.w, src); cr.aa_range = __shfl(mine_cr.aa_range, src); cr.stx = __shfl(mine_cr.stx, src); cr.sty = __shfl(mine_cr.sty, src);
      cr.n12 = __shfl(mine_cr.n12, src); cr.n34 = __shfl(mine_cr.n34, src); cr.corners = __shfl(mine_cr.corners, src);
#endif
      if (!cy[j] || !anyx) continue;
      WrRow4 r4;
      {
        // a group whose pixels all sit in one solid phase of the row (clear / opaque) needs no evaluation at all
        const int n = px - x0, len = x1 - x0, span = len >= 4 ? (len & ~3) : 0;
        const int b1 = cr.n12 & 0xFFFF, b2 = b1 + (cr.n12 >> 16), b3 = b2 + (cr.n34 & 0xFFFF), b4 = b3 + (cr.n34 >> 16);
        const int c0 = n >> 2, c3 = (n + 3) >> 2;
        const int k0 = c0 < b1 ? 0 : (c0 < b2 ? 1 : (c0 < b3 ? 2 : (c0 < b4 ? 3 : 0)));
        const int k3 = c3 < b1 ? 0 : (c3 < b2 ? 1 : (c3 < b3 ? 2 : (c3 < b4 ? 3 : 0)));
        if (n >= 0 && n + 3 < span && k0 == k3 && (k0 == 0 || k0 == 2) && Ap->clip.w > 0.0f) {
          const float mode = Ap->clip.mode;
          const uint32_t v = uint32_t(wr_round_pixel(k0 == 0 ? mode : 1.0f - mode)) & 0xFFFF;
          r4.v[0] = r4.v[1] = r4.v[2] = r4.v[3] = v;
        } else {
          r4 = wr_clip_rect_row4(Pp, &Ap->clip, rv, cr, px, py + 4 * j);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        if (cx[i]) plo[q] = wr_blend_r8(blend, plo[q], r4.v[i]);
      }
    }
    return;
  }
  if ((FEAT & WR_FEAT_BLUR) && kind == WR_PK_BLUR) {
    const WrDrawDesc* D = &draws[Pp->draw];
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      if (!(cx[q & 3] && cy[q >> 2])) continue;
      const WrWide src = wr_blur_pixel<FMT>(Pp, &Ap->blur, px + (q & 3), py + 4 * (q >> 2));
      if (FMT == WR_FMT_RGBA8) {
        const uint32_t r = wr_blend_rgba8(blend, plo[q] | (phi[q] << 8), src, D);
        plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
      } else {
        plo[q] = wr_blend_r8(blend, plo[q], src.bg & 0xFFFF);
      }
    }
    return;
  }
  if ((FEAT & WR_FEAT_SHADE) && FMT == WR_FMT_RGBA8 && kind == WR_PK_GRADIENT) {
    if (!(cx[0] || cx[1] || cx[2] || cx[3])) return;
    const WrDrawDesc* D = &draws[Pp->draw];
#pragma unroll
    for (int j = 0; j < R; j++) {
      if (!cy[j]) continue;
      WrGrad4 g4;
      if (!rr) g4 = wr_gradient_row4(Pp, &Ap->grad, D, px, py + 4 * j);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        bool in = cx[i];
        if (dtest) {
          const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
          in = in && pass;
          if (dwrite) dep[q] = in ? z : dep[q];
        }
        if (!in) continue;
        if (rr) g4.v[i] = wr_gradient_row4(Pp, &Ap->grad, D, px + i, py + 4 * j, &rr[py + 4 * j - wy0]).v[0];   // depth runs: pixel by pixel
        const uint32_t r = wr_blend_rgba8(blend, plo[q] | (phi[q] << 8), wr_mask_src(*Pp, D, px + i, py + 4 * j, g4.v[i]), D);
        plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
      }
    }
    return;
  }
  if ((FEAT & WR_FEAT_SHADE) && FMT == WR_FMT_RGBA8 && kind == WR_PK_QUAD_MASK) {
    const WrDrawDesc* D = &draws[Pp->draw];
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      bool in = cx[q & 3] && cy[q >> 2];
      if (dtest) {
        const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
        in = in && pass;
        if (dwrite) dep[q] = in ? z : dep[q];
      }
      if (!in) continue;
      const WrWide src = wr_quad_mask_pixel(Pp, &Ap->clip, D, px + (q & 3), py + 4 * (q >> 2), rr ? &rr[py + 4 * (q >> 2) - wy0] : nullptr);
      const uint32_t r = wr_blend_rgba8(blend, plo[q] | (phi[q] << 8), src, D);
      plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
    }
    return;
  }
  if ((FEAT & WR_FEAT_SHADE) && FMT == WR_FMT_RGBA8 && (kind == WR_PK_BORDER_SOLID || kind == WR_PK_BORDER_SEGMENT || kind == WR_PK_FAST_GRADIENT || kind == WR_PK_LINE_DECORATION)) {
    const WrDrawDesc* D = &draws[Pp->draw];
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      bool in = cx[q & 3] && cy[q >> 2];
      if (dtest) {
        const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
        in = in && pass;
        if (dwrite) dep[q] = in ? z : dep[q];
      }
      if (!in) continue;
      const WrRuns* rq = rr ? &rr[py + 4 * (q >> 2) - wy0] : nullptr;
      const WrWide src = kind == WR_PK_BORDER_SOLID ? wr_border_solid_pixel(Pp, &Ap->border, D, px + (q & 3), py + 4 * (q >> 2), rq)
                         : kind == WR_PK_BORDER_SEGMENT ? wr_border_segment_pixel(Pp, &Ap->bseg, D, px + (q & 3), py + 4 * (q >> 2), rq)
                                                        : wr_cache_shader_pixel(Pp, Ap, D, px + (q & 3), py + 4 * (q >> 2), rq);
      const uint32_t r = wr_blend_rgba8(blend, plo[q] | (phi[q] << 8), src, D);
      plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
    }
    return;
  }
  if ((FEAT & WR_FEAT_SHADE) && FMT == WR_FMT_RGBA8 && (kind == WR_PK_FILTER || kind == WR_PK_TEX_REPEAT || kind == WR_PK_MIX_BLEND || kind == WR_PK_YUV || kind == WR_PK_SVG_FILTER)) {
    const WrDrawDesc* D = &draws[Pp->draw];
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      bool in = cx[q & 3] && cy[q >> 2];
      if (dtest) {
        const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
        in = in && pass;
        if (dwrite) dep[q] = in ? z : dep[q];
      }
      if (!in) continue;
      const WrRuns* rq = rr ? &rr[py + 4 * (q >> 2) - wy0] : nullptr;
      if (kind == WR_PK_TEX_REPEAT && Pp->dual && blend == WR_BLEND_DUAL_SRC) {      // (the dual-source repetition key: the blend takes two colours)
        const uint32_t r2 = wr_repeat_dual_pixel(Pp, &Ap->rep, D, px + (q & 3), py + 4 * (q >> 2), plo[q] | (phi[q] << 8), rq);
        plo[q] = r2 & WR_M8; phi[q] = (r2 >> 8) & WR_M8;
        continue;
      }
      const WrWide raw = kind == WR_PK_FILTER ? wr_filter_pixel(Pp, &Ap->filt, D, px + (q & 3), py + 4 * (q >> 2), rq)
                         : kind == WR_PK_MIX_BLEND ? wr_mix_blend_pixel(Pp, &Ap->mix, D, px + (q & 3), py + 4 * (q >> 2), rq)
                         : kind == WR_PK_YUV ? wr_yuv_pixel(Pp, &Ap->yuv, D, px + (q & 3), py + 4 * (q >> 2), rq)
                         : kind == WR_PK_SVG_FILTER ? wr_svg_filter_pixel(Pp, &Ap->svg, D, px + (q & 3), py + 4 * (q >> 2), rq)
                                                   : wr_repeat_pixel(Pp, &Ap->rep, D, px + (q & 3), py + 4 * (q >> 2), rq);
      const WrWide src = wr_mask_src(*Pp, D, px + (q & 3), py + 4 * (q >> 2), raw);
      const uint32_t r = wr_blend_rgba8(blend, plo[q] | (phi[q] << 8), src, D);
      plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
    }
    return;
  }
  if ((FEAT & WR_FEAT_SHADE) && FMT == WR_FMT_RGBA8 && kind == WR_PK_TEX_QUAD) {
    const WrDrawDesc* D = &draws[Pp->draw];
    const bool persp = Ap->quad.pad != 0;       // (see WR_PK_SOLID_QUAD below)
    WrQuadRowCache rowc;                        // the lane's current row (wr_quad_row_edges)
    rowc.y = -0x40000000; rowc.si = -1;
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      const bool in = cx[q & 3] && cy[q >> 2];
      if (!in) continue;
      const uint32_t before = plo[q] | (phi[q] << 8);
      bool pass = true;
      uint32_t zq = z;
      if (dtest && persp) zq = wr_persp_depth(&Ap->quad, px + (q & 3), py + 4 * (q >> 2), &rowc);
      if (dtest) pass = dless ? (zq < dep[q]) : (zq <= dep[q]);
      if (!pass) continue;
      const unsigned long long hr = wr_quad_tex_pixel_rgba8(Pp, &Ap->quad, D, px + (q & 3), py + 4 * (q >> 2), before,
                                                            (rr && !persp) ? &rr[py + 4 * (q >> 2) - wy0] : nullptr, &rowc);
      if (!(hr >> 32)) continue;
      const uint32_t r = (uint32_t)hr;
      if (dtest && dwrite) dep[q] = zq;
      plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
    }
    return;
  }
  if ((FEAT & WR_FEAT_GENERIC) && FMT == WR_FMT_RGBA8 && kind == WR_PK_SOLID_QUAD) {
    const WrDrawDesc* D = &draws[Pp->draw];
    // perspective quads: depth varies per pixel, and the row is drawn chunk by chunk from the span start against the
    // flattened depth row (draw_span<.., true>, rasterize.h:667-690) -- no restarts at depth runs
    const bool persp = Ap->quad.pad != 0;
    float pxl = 0.0f, pxr = 0.0f;
    int psi = -1, pyy = 0;
#pragma unroll
    for (int j = 0; j < R; j++) {
      if (!(cy[j] && (cx[0] || cx[1] || cx[2] || cx[3]))) continue;
      const WrQuadRowS row = wr_quad_row_setup(&Ap->quad, py + 4 * j, pxl, pxr, psi, pyy);       // the lane's row: shared by its four pixels
      pxl = row.xl; pxr = row.xr; psi = row.si; pyy = row.y;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        if (!cx[i]) continue;
        const uint32_t before = plo[q] | (phi[q] << 8);
        bool pass = true;
        uint32_t zq = z;
        if (dtest && persp) zq = wr_persp_depth(&Ap->quad, px + i, py + 4 * j);
        if (dtest) pass = dless ? (zq < dep[q]) : (zq <= dep[q]);
        if (!pass) continue;
        const unsigned long long hr = wr_quad_row_pixel_rgba8(row, Ap->quad.aa, D, blend, c0, c1, px + i, before,
                                                              (rr && !persp) ? &rr[py + 4 * j - wy0] : nullptr);
        if (!(hr >> 32)) continue;
        const uint32_t r = (uint32_t)hr;
        if (dtest && dwrite) dep[q] = zq;
        plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
      }
    }
    return;
  }
  if ((FEAT & WR_FEAT_GENERIC) && FMT == WR_FMT_RGBA8 && kind == WR_PK_SOLID_AA) {
    // DO_AA (blend.h:433-446): src = muldiv256(src, coverage) ahead of the blend
    const WrAARec& A = Ap->aa;
    const WrDrawDesc* D = &draws[Pp->draw];
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      bool in = cx[q & 3] && cy[q >> 2];
      if (dtest) {
        const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
        in = in && pass;
        if (dwrite) dep[q] = in ? z : dep[q];
      }
      if (!in) continue;
      int cs = x0;             // start of the 4-pixel chunks: the span start, or the start of the depth run holding the pixel
      if (rr) { const WrRuns* rq = &rr[py + 4 * (q >> 2) - wy0]; const int k = wr_find_run(rq, px + (q & 3)); if (k >= 0) cs = wr_run_s(rq, k); }
      const uint32_t r = wr_aa_pixel_rgba8(&A, D, blend, c0, c1, px + (q & 3) - cs, cs, plo[q] | (phi[q] << 8));
      plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
    }
    return;
  }
  // ---- generic path ----
  WR_DBG_PATH(1);
  const WrDrawDesc* D = &draws[Pp->draw];
#pragma unroll
  for (int q = 0; q < NPX; q++) {
    bool in = cx[q & 3] && cy[q >> 2];
    if (dtest) {
      const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
      in = in && pass;
      if (dwrite) dep[q] = in ? z : dep[q];
    }
    if (in) {
      if (FMT == WR_FMT_RGBA8) {
        if (FEAT & WR_FEAT_GENERIC) {
          uint32_t r = wr_generic_pixel_rgba8(Pp, D, px + (q & 3), py + 4 * (q >> 2), plo[q] | (phi[q] << 8),
                                              rr ? &rr[py + 4 * (q >> 2) - wy0] : nullptr);
          plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
        }
      } else {
        // R8 target: pack_pixels_R8(gl_FragColor.x) (blend.h:67-73)
        uint32_t srcr = c1 & 0xFFFF;
        if ((FEAT & WR_FEAT_GENERIC) && (kind == WR_PK_TEX_FS || kind == WR_PK_TEX_RGBA8))
          srcr = wr_tex_pixel_r(Pp, D, px + (q & 3), py + 4 * (q >> 2));
        plo[q] = wr_blend_r8(blend, plo[q], srcr);
      }
    }
  }
}

// swgl_commitTextureLinearColorR8ToRGBA8 (glyph blits) for prims whose WrTexRec
// is `simple`: the quantised x coordinate of a column is the same on every row,
// so it is set up once per prim, the y coordinate once per row, and a pixel
// costs one atlas byte when both 7-bit fractions are zero.  `T` is wave-uniform.
template <bool DEPTH, int R>
WR_DEVICE void wr_apply_tex_r8(uint32_t (&plo)[4 * R], uint32_t (&phi)[4 * R], uint32_t (&dep)[4 * R],
                               const int x0, const int y0, const int x1, const int y1, const uint32_t z,
                               const uint32_t kbf, const uint32_t c0, const uint32_t c1,
                               const WrTexRec& T, const WrDrawDesc* draws, const WrPrim* Pp, const int px, const int py) {
  WR_DBG_PATH(0);
  const int blend = (kbf >> 8) & 0xFF, flags = (kbf >> 16) & 0xFF;
  const bool dtest = DEPTH && (flags & WR_PF_DEPTH_TEST);
  const bool dwrite = (flags & WR_PF_DEPTH_WRITE) != 0, dless = (flags & WR_PF_DEPTH_LESS) != 0;
  const uint8_t* sbuf = (const uint8_t*)T.ptr;
  const int tw = int(T.wh & 0xFFFF), th = int(T.wh >> 16);
  const float W = float(tw), H = float(th);
  const float qs = 128.0f, qo = 0.5f - 0.5f * qs;
  const int span = T.span;
  bool cx[4], cy[R];
#pragma unroll
  for (int i = 0; i < 4; i++) cx[i] = (unsigned)(px + i - x0) < (unsigned)(x1 - x0);
#pragma unroll
  for (int j = 0; j < R; j++) cy[j] = (unsigned)(py + 4 * j - y0) < (unsigned)(y1 - y0);
  const uint32_t clo = (c0 & 0xFFFF) | (c1 << 16), chi = (c0 >> 16) | (c1 & 0xFFFF0000u);
  if (T.unit) {
    // every sample is exactly one texel: m = atlas[iy0 + row][ix0 + n] (tail columns from tix[])
    WR_DBG_PATH(3);
    int colu[4]; bool tl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int n = px + i - x0, k = n - span;
      tl[i] = k >= 0;
      colu[i] = k < 0 ? T.ix0 + n : (k == 0 ? T.tix[0] : (k == 1 ? T.tix[1] : T.tix[2]));
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
      if (!cy[j]) continue;
      const uint8_t* srow = sbuf + (size_t)(T.iy0 + (py + 4 * j - T.y0)) * T.stride;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        bool in = cx[i];
        if (dtest) {
          const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
          in = in && pass;
          if (dwrite) dep[q] = in ? z : dep[q];
        }
        if (!in) continue;
        const uint32_t um = srow[colu[i]];
        if (!tl[i]) {
          const uint32_t sl = wr_hi_bytes(wr_mul24(clo, um) + clo), sh = wr_hi_bytes(wr_mul24(chi, um) + chi);
          uint32_t nl = sl, nh = sh;
          if (blend == WR_BLEND_PREMULT) {
            const uint32_t K = 255u - (sh >> 16);
            nl = wr_pk_min_u16(wr_hi_bytes(wr_mul24(plo[q], K) + WR_M8) + sl, WR_M8);
            nh = wr_pk_min_u16(wr_hi_bytes(wr_mul24(phi[q], K) + WR_M8) + sh, WR_M8);
          }
          plo[q] = nl; phi[q] = nh;
        } else {
          const float mf = float(um) * (1.0f / 255.0f);
          uint32_t pc[2];
          wr_pack_color(wf4{T.fcolor[0] * mf, T.fcolor[1] * mf, T.fcolor[2] * mf, T.fcolor[3] * mf}, pc);
          WrWide src; src.bg = pc[0]; src.ra = pc[1];
          const uint32_t r = wr_blend_rgba8(blend, plo[q] | (phi[q] << 8), src, &draws[Pp->draw]);
          plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
        }
      }
    }
    return;
  }
  int col[4], fracx[4]; bool tail[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n = px + i - x0;
    int qx = 0;
    tail[i] = n >= span;
    if (cx[i]) {
      const int lane = (tail[i] ? n - span : n) & 3;
      float lu = T.ou;
      if (lane > 0) lu += T.su;
      if (lane > 1) lu += T.su;
      if (lane > 2) lu += T.su;
      if (!tail[i]) {
        const float q = wr_accum_short(lu * W * qs + qo, T.stepx, n >> 2);
        qx = int(wr_clamp(q, T.minx, T.maxx));
      } else {
        if (span > 0) lu = lu + (T.su * 4.0f) * (float(span) * 0.25f);
        const float cu = wr_clamp(lu, T.ub0, T.ub2);
        qx = int(cu * W * 128.0f + (0.5f - 64.0f));
      }
    }
    const int ix = qx >> 7;
    const int over = ix > tw - 2 ? -1 : 0;
    col[i] = wr_clamp_coord(ix, tw - 1);
    fracx[i] = ((((ix >= 0) ? qx : 0) | over) & 0x7F) - over;
  }
#pragma unroll
  for (int j = 0; j < R; j++) {
    if (!cy[j]) continue;
    const int y = py + 4 * j;
    const float ov = wr_accum_short(T.lv0, T.lvs, y - T.y0);      // Lv == Rv on this kind of prim, so sv == 0 exactly
    const int qy_span = int(wr_clamp(ov * H * qs + qo, T.miny, T.maxy));
    const int qy_tail = int(wr_clamp(ov, T.ub1, T.ub3) * H * 128.0f + (0.5f - 64.0f));
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int q = 4 * j + i;
      bool in = cx[i];
      if (dtest) {
        const bool pass = dless ? (z < dep[q]) : (z <= dep[q]);
        in = in && pass;
        if (dwrite) dep[q] = in ? z : dep[q];
      }
      if (!in) continue;
      const int qy = tail[i] ? qy_tail : qy_span;
      const int iy = qy >> 7, fracy = qy & 0x7F;
      const size_t row0 = (size_t)col[i] + (size_t)wr_clamp_coord(iy, th) * T.stride;
      int m = sbuf[row0];
      if ((fracx[i] | fracy) != 0) {
        const size_t row1 = row0 + ((iy >= 0 && iy < th - 1) ? T.stride : 0);
        const int p01 = sbuf[row0 + 1], p10 = sbuf[row1], p11 = sbuf[row1 + 1];
        const int l = (int16_t)(m + (int16_t)(((int16_t)((p10 - m) * fracy)) >> 7));
        const int r = (int16_t)(p01 + (int16_t)(((int16_t)((p11 - p01) * fracy)) >> 7));
        m = (int16_t)(l + (int16_t)(((int16_t)((r - l) * fracx[i])) >> 7));
      }
      if (!tail[i]) {
        // applyColor(expand_mask(m), colour) = muldiv255 per channel, then the blend
        const uint32_t um = uint32_t(m);
        const uint32_t sl = wr_hi_bytes(wr_mul24(clo, um) + clo), sh = wr_hi_bytes(wr_mul24(chi, um) + chi);
        uint32_t nl = sl, nh = sh;
        if (blend == WR_BLEND_PREMULT) {
          const uint32_t K = 255u - (sh >> 16);
          nl = wr_pk_min_u16(wr_hi_bytes(wr_mul24(plo[q], K) + WR_M8) + sl, WR_M8);
          nh = wr_pk_min_u16(wr_hi_bytes(wr_mul24(phi[q], K) + WR_M8) + sh, WR_M8);
        }
        plo[q] = nl; phi[q] = nh;
      } else {
        // fragment-shader tail: texel -> float, modulate, round_pixel, generic blend
        const float mf = float(m) * (1.0f / 255.0f);
        uint32_t pc[2];
        wr_pack_color(wf4{T.fcolor[0] * mf, T.fcolor[1] * mf, T.fcolor[2] * mf, T.fcolor[3] * mf}, pc);
        WrWide src; src.bg = pc[0]; src.ra = pc[1];
        const uint32_t r = wr_blend_rgba8(blend, plo[q] | (phi[q] << 8), src, &draws[Pp->draw]);
        plo[q] = r & WR_M8; phi[q] = (r >> 8) & WR_M8;
      }
    }
  }
}

// Unit glyph blits, lane by lane (device loop of wr_raster_body).  A glyph of a text run is ~10 x 15 pixels: of the 64 lanes of
// a 64 x 16 strip it lights 12-16, and a strip of a text line holds a dozen glyphs side by side -- applied one prim at a time,
// every one of them costs the whole wave its 16 pixels per lane.  Within a run of consecutive unit-glyph prims (every sample
// exactly one atlas texel: WrTexRec::unit, blend NONE / PREMULT, no depth test) the order only matters per pixel, so each lane
// walks its OWN list -- the prims of the run whose rect reaches its 4 x 4R footprint, in submission order -- and the wave is
// done after max-over-lanes(list length) rounds instead of one round per prim.  Same arithmetic as wr_apply_tex_r8's unit path;
// the prim's record and sampling setup are per-lane values here (vector loads from recs[] / aux[]).
// (round 4) The lane reads ONE 32-byte WrGlyphRec per glyph (a dense array: neighbouring glyphs share cache lines) instead of the
// prim's WrRec and the head of its WrTexRec, and the four columns of each of its rows with one unaligned dword load -- R loads
// issued back to back at addresses that are valid whatever the lane's coverage (column clamped into [x0, max(x0, x1 - 4)], row into
// [y0, y1)), then branch-free blends under selects -- where it used to issue up to 4R byte loads, each in its own divergent
// region behind a full wait.  Tail columns (x >= WrGlyphRec::info >> 16: main()'s float path) are redone under one branch.
WR_DEVICE void wr_glyph_tail_px(uint32_t& lo, uint32_t& hi, uint32_t um, float fr, float fg, float fb, float fa, bool premult) {
  // main(): texel -> float, modulate, round_pixel (u16 lanes), then blend NONE / PREMULT (src + dst - muldiv255(dst, src.a)) and
  // pack with swgl's saturation -- wr_pack_color + wr_blend_rgba8 on the (b, r) / (g, a) pairing the pixel registers use
  const float mf = float(um) * (1.0f / 255.0f);
  const uint32_t b = uint32_t(wr_round_pixel(fb * mf)) & 0xFFFF, g = uint32_t(wr_round_pixel(fg * mf)) & 0xFFFF;
  const uint32_t r = uint32_t(wr_round_pixel(fr * mf)) & 0xFFFF, a = uint32_t(wr_round_pixel(fa * mf)) & 0xFFFF;
  uint32_t slo = b | (r << 16), shi = g | (a << 16);
  if (premult) {
    const uint32_t aa = a | (a << 16);
    slo = wr_sub2(wr_add2(slo, lo), wr_muldiv255_2(lo, aa));
    shi = wr_sub2(wr_add2(shi, hi), wr_muldiv255_2(hi, aa));
  }
  lo = wr_pack1(slo & 0xFFFF) | (wr_pack1(slo >> 16) << 16);
  hi = wr_pack1(shi & 0xFFFF) | (wr_pack1(shi >> 16) << 16);
}
template <int R>
WR_DEVICE void wr_unit_glyph_lane(uint32_t (&plo)[4 * R], uint32_t (&phi)[4 * R], const uint4 ga, const uint4 gb, const WrGlyphRec* __restrict__ Gp,
                                  const int px, const int py) {
  const int x0 = (int)(int16_t)(ga.x & 0xFFFF), y0 = (int)(int16_t)(ga.x >> 16), x1 = (int)(int16_t)(ga.y & 0xFFFF), y1 = (int)(int16_t)(ga.y >> 16);
  const uint32_t c0 = ga.z, c1 = ga.w;
  const uint8_t* base = (const uint8_t*)(((unsigned long long)gb.y << 32) | gb.x);
  const long long stride = (long long)(int)gb.z;
  const bool premult = ((gb.w >> 8) & 0xFF) == WR_BLEND_PREMULT;
  const int tail_x = (int)(int16_t)(gb.w >> 16);
  const uint32_t clo = (c0 & 0xFFFF) | (c1 << 16), chi = (c0 >> 16) | (c1 & 0xFFFF0000u);
  const int cs = wr_iclamp(px, x0, wr_imax(x1 - 4, x0));
  const int sh8 = 8 * (px - cs);              // pixel i of a covered column: byte (i + px - cs) of the row's dword
  uint32_t rowv[R];
#pragma unroll
  for (int j = 0; j < R; j++) {
    const int y = wr_iclamp(py + 4 * j, y0, y1 - 1);
#ifdef WR_GLYPH_ALIGNED_LOADS
    const unsigned long long a = (unsigned long long)(base + (long long)y * stride + cs);
    const uint2 v2 = *(const uint2*)(a & ~3ull);
    rowv[j] = (uint32_t)(((((unsigned long long)v2.y) << 32) | v2.x) >> (8u * (unsigned)(a & 3ull)));
#else
    __builtin_memcpy(&rowv[j], base + (long long)y * stride + cs, 4);
#endif
  }
  bool cx[4];
#pragma unroll
  for (int i = 0; i < 4; i++) cx[i] = (unsigned)(px + i - x0) < (unsigned)(x1 - x0);
  const uint32_t ksel = premult ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int j = 0; j < R; j++) {
    const bool rowin = (unsigned)(py + 4 * j - y0) < (unsigned)(y1 - y0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int q = 4 * j + i;
      const uint32_t um = (rowv[j] >> ((sh8 + 8 * i) & 31)) & 0xFFu;
      const uint32_t sl = wr_hi_bytes(wr_mul24(clo, um) + clo), sh = wr_hi_bytes(wr_mul24(chi, um) + chi);
      const uint32_t K = (255u - (sh >> 16)) & ksel;
      const uint32_t nl = wr_pk_min_u16(wr_hi_bytes(wr_mul24(plo[q], K) + WR_M8) + sl, WR_M8);
      const uint32_t nh = wr_pk_min_u16(wr_hi_bytes(wr_mul24(phi[q], K) + WR_M8) + sh, WR_M8);
      const bool on = rowin && cx[i] && px + i < tail_x;      // (branch-free: skipping a lane's uncovered rows was measured slower)
      plo[q] = on ? nl : plo[q]; phi[q] = on ? nh : phi[q];
    }
  }
  if (px + 3 >= tail_x && px < x1) {          // (some of this lane's columns are tail columns)
    // (the colour sits in the record's third 16 bytes -- next to what the lane has just read, not in the prim's 1 KB-strided WrAux)
    const uint4 gc = ((const uint4*)Gp)[2];
    const float fr = wr_bits_f(gc.x), fg = wr_bits_f(gc.y), fb = wr_bits_f(gc.z), fa = wr_bits_f(gc.w);
#pragma unroll
    for (int j = 0; j < R; j++) {
      const bool rowin = (unsigned)(py + 4 * j - y0) < (unsigned)(y1 - y0);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * j + i;
        if (!(rowin && cx[i] && px + i >= tail_x)) continue;
        const uint32_t um = (rowv[j] >> ((sh8 + 8 * i) & 31)) & 0xFFu;
        wr_glyph_tail_px(plo[q], phi[q], um, fr, fg, fb, fa, premult);
      }
    }
  }
}

// One workgroup per 64x64 bin; 64/(4R) waves of 64 lanes, each lane 4 x R pixels.
// Strip-level depth cap (all wave-uniform, scalar): an upper bound on every depth sample of the wave's strip.  Depth only
// ever moves towards the viewer except through a clear, so once a depth-writing rect has covered the whole strip at z, a
// later depth-tested prim at or behind z fails at every pixel of the strip and is skipped before any vector work -- what
// swgl's depth runs do for whole spans (rasterize.h:573-660), at strip granularity.  WebRender submits its opaque pass front
// to back for exactly this reason (batch.rs: opaque batches are drawn in reverse), so in a scene with real overdraw most
// prims of a strip never reach the blend code.
WR_DEVICE bool wr_zcap_rejects(uint32_t kbf, uint32_t z, uint32_t zcap) {
  const uint32_t fl = (kbf >> 16) & 0xFF;
  if (!(fl & WR_PF_DEPTH_TEST) || (kbf & 0xFF) == WR_PK_CLEAR) return false;
  return (fl & WR_PF_DEPTH_LESS) ? z >= zcap : z > zcap;
}
WR_DEVICE uint32_t wr_zcap_after(uint32_t kbf, uint32_t z, uint32_t zcap, bool full) {
  const uint32_t fl = (kbf >> 16) & 0xFF, k = kbf & 0xFF;
  if (k == WR_PK_CLEAR) return (fl & WR_PF_CLEAR_DEPTH) ? (full ? z : (z > zcap ? z : zcap)) : zcap;
  // (rect kinds whose every pixel of [x0, x1) x [y0, y1) writes depth when it passes: solids and the unmasked axis-aligned shader
  // replays -- images, gradients, filters, repeated images, video: the opaque pass of a page of stacked full-size gradients
  // (wrench aligned- / unaligned-gradient: ten, front to back) is one gradient per strip and nine scalar rejections)
  const bool rect_kind = k == WR_PK_SOLID_FOLDED || k == WR_PK_SOLID ||
                         ((k == WR_PK_TEX_RGBA8 || k == WR_PK_TEX_FS || k == WR_PK_GRADIENT || k == WR_PK_FILTER || k == WR_PK_TEX_REPEAT || k == WR_PK_YUV) && !(fl & WR_PF_MASKED));
  if (full && rect_kind && (fl & WR_PF_DEPTH_TEST) &&
      (fl & WR_PF_DEPTH_WRITE))
    return z < zcap ? z : zcap;
  return zcap;
}


#ifdef WRHIP_HOSTSIM
#define WR_CT(i) ((void)0)
#endif
#ifndef WRHIP_HOSTSIM
// ---------------------------------------------------------------------------
// Cell raster (rect-only launches, device only).  A bin that starts from a clear and receives nothing but axis-aligned flat
// colours -- solid rects, clears -- has few DISTINCT pixels: the x edges and y edges of the prims that reach it cut it into
// nx x ny cells, and every pixel of a cell sees the same prims in the same order on the same start value, so it ends up with
// the same bytes (swgl's blend is a per-pixel function of (dst, src), blend.h:416-735, and the span is all-or-nothing per pixel
// without swgl_antiAlias).  Instead of blending 4096 pixels through every prim, the workgroup
//   A  collects the edges: the lanes fetch the records of the bin's prim list (the four waves share the words out), each
//      sets the bits of its prim's column / row boundaries inside the bin, OR-combined through LDS;
//   B  gives every lane ONE cell (its first pixel) and walks the prim list in submission order: per prim one coverage test
//      and one blend per lane instead of sixteen -- only the waves that hold cells walk;
//   C  expands: a pixel's cell is (rank of its row among the row boundaries, rank of its column), two popcounts, and its
//      colour (and depth) one LDS read.
// cfg2 (1000 translucent rects over a 4K frame, ~45 prims per bin cutting it into ~100 cells): the tile pass becomes a store
// stream.  More than 256 cells (a bin crossed by dozens of small rects, cfg5) or a bin that loads its pixels: the pixel walk.
#define WR_CELL_MAX_PRIMS 160
struct WrCellShared {
  uint32_t color[256], dep[256];        // finished cells: packed BGRA8, depth
  uint8_t colflag[64], rowflag[64];     // column / row boundaries: [c] != 0 = a class starts at column (row) c of the bin
  uint8_t colstart[4][64], rowstart[4][64];   // per wave: first column / row of every class
  int pid[WR_CELL_MAX_PRIMS];           // the bin's prim list in submission order: global prim indices ..
  uint4 rec[2][WR_CELL_MAX_PRIMS + 4];  // .. and their records (+ padding the walk's last request may touch).  Depth-tested launches: (x0, y0, x1, y1), (z, kbf, c0, c1) as in WrRec;
                                        // depth-less ones: (x0, x1 - x0, y0, y1 - y0), (K, Clo, Chi, -) -- every kind that draws is a fold
  int bail;                             // a prim the cell walk has no form for (a solid whose blend needs pack()'s clamp)
};
static_assert(WR_CELL_MAX_PRIMS <= 256 && WR_CELL_MAX_PRIMS % 4 == 0, "A2 runs one thread per slot");
#ifdef WR_CELL_TIMING
__device__ unsigned long long wr_cell_times[8192 * 16];
#define WR_CT(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) wr_cell_times[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WR_CT(i) ((void)0)
#endif
// wr_fold_inplace on both channel pairs of one pixel, K and C in VGPRs (a wave-uniform value read from LDS stays a vector)
WR_DEVICE void wr_fold_masked_v(uint32_t& lo, uint32_t& hi, uint32_t K, uint32_t Clo, uint32_t Chi, wr_lanemask m) {
  unsigned long long saved;
  asm("s_and_saveexec_b64 %2, %3\n\t"
      "v_mad_u32_u24 %0, %0, %4, %5\n\tv_perm_b32 %0, 0, %0, %7\n\t"
      "v_mad_u32_u24 %1, %1, %4, %6\n\tv_perm_b32 %1, 0, %1, %7\n\t"
      "s_mov_b64 exec, %2"
      : "+v"(lo), "+v"(hi), "=&s"(saved)
      : "s"(m), "v"(K), "v"(Clo), "v"(Chi), "v"(0x0c030c01u)
      : "scc");
}
// Inclusive prefix sum over the 64 lanes of a wave in seven DPP adds (row shifts inside the rows of 16, then the row totals
// broadcast down): no LDS round trips, unlike a __shfl_up ladder.
WR_DEVICE int wr_wave_scan_incl(int x) {
  int t = x;
  t += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1
  t += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2
  t += __builtin_amdgcn_update_dpp(0, x, 0x113, 0xf, 0xf, true);      // row_shr:3
  t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xe, true);      // row_shr:4  bank_mask:0xe
  t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xc, true);      // row_shr:8  bank_mask:0xc
  t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xa, 0xf, true);      // row_bcast:15 row_mask:0xa
  t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xc, 0xf, true);      // row_bcast:31 row_mask:0xc
  return t;
}
WR_DEVICE void wr_fold_inplace_v(uint32_t& p, uint32_t K, uint32_t C) {
  asm("v_mad_u32_u24 %0, %0, %1, %2\n\tv_perm_b32 %0, 0, %0, %3" : "+v"(p) : "v"(K), "v"(C), "v"(0x0c030c01u));
}
template <bool DEPTH>
WR_DEVICE int wr_raster_cells(WrCellShared& sh, const WrTargetDesc& T, const WrRec* __restrict__ recs,
                               unsigned long long* __restrict__ mw, const int nw, const int bx0, const int by0,
                               const int wave, const int lane, uint32_t (&plo)[16], uint32_t (&phi)[16], uint32_t (&dep)[16],
                               const bool try_cells, int& list_total) {
  WR_CT(0);
  // The waves of a workgroup sit on the four SIMDs of a CU in order, and the cell walk keeps the low wave indices busy (a bin
  // has ~70 cells: one wave walks, sometimes two): the jobs rotate with the bin so that they spread over the SIMDs.
  const int role = (wave + (int)(blockIdx.x >> 3)) & 3;
  // ---- A1: the bin's prim list, compacted in submission order ---------------------------------------------------------------
  // A prim's slot is the number of set bits ahead of its own in the bin's mask words: a wave prefix sum over the words'
  // popcounts (every wave computes it, so all four agree on the total without a barrier); the lanes of a block are dealt out
  // to the four waves to write the prim indices of their words.
  if (threadIdx.x < 32) ((uint32_t*)sh.colflag)[threadIdx.x] = 0u;       // (colflag + rowflag: 128 bytes)
  if (threadIdx.x == 32) sh.bail = 0;
  int total = 0;
  for (int wb = 0; wb < nw; wb += 64) {
    const unsigned long long mv = wb + lane < nw ? mw[wb + lane] : 0ull;
    const int cnt = __popcll(mv);
    const int inc = wr_wave_scan_incl(cnt);
    const int blk_total = __builtin_amdgcn_readlane(inc, 63);
    if (total + blk_total > WR_CELL_MAX_PRIMS) return 0;      // (the same decision in every wave; nothing was modified)
    if ((lane & 3) == role) {
      int sl = total + inc - cnt;
      for (unsigned long long bts = mv; bts; bts &= bts - 1ull, sl++) sh.pid[sl] = T.first_prim + (wb + lane) * 64 + __builtin_ctzll(bts);
    }
    total += blk_total;
  }
  total = __builtin_amdgcn_readfirstlane(total);
  if (total == 0) return 2;       // an empty bin (outside every draw's clip): nothing to walk, nothing to zero; no barrier met yet
  __syncthreads();
  WR_CT(1);
  // ---- A2: one thread per prim: its record into LDS, its edges inside the bin into the flag bytes ----------------------------
  {
    const int sl = role * 64 + lane;
    bool full = false;
    if (sl < total) {
      const uint4* rp = (const uint4*)&recs[sh.pid[sl]];
      uint4 ra = rp[0], rb = rp[1];
      const int a0 = (int)ra.x - bx0, b0 = (int)ra.y - by0, a1 = (int)ra.z - bx0, b1 = (int)ra.w - by0;
      if (a1 > 0 && a0 < WR_BIN_W && b1 > 0 && b0 < WR_BIN_H && a1 > a0 && b1 > b0) {
        if (a0 > 0) sh.colflag[a0] = 1;
        if (a1 < WR_BIN_W) sh.colflag[a1] = 1;
        if (b0 > 0) sh.rowflag[b0] = 1;
        if (b1 < WR_BIN_H) sh.rowflag[b1] = 1;
        full = a0 <= 0 && a1 >= WR_BIN_W && b0 <= 0 && b1 >= WR_BIN_H;
      }
      if (!DEPTH) {
        // every kind that draws in a rect-only launch as new = hi_bytes(dst * K + C) on the rect (x0, w, y0, h)
        const uint32_t kbf = rb.y, kind = kbf & 0xFF, flags = (kbf >> 16) & 0xFF;
        uint4 qa = make_uint4(ra.x, ra.z - ra.x, ra.y, ra.w - ra.y), qb = make_uint4(kbf >> 24, rb.z, rb.w, 0);
        if (kind == WR_PK_CLEAR && (flags & WR_PF_CLEAR_COLOR)) { qb.x = 0; qb.y = (rb.z & WR_M8) << 8; qb.z = rb.z & 0xFF00FF00u; }
        else if (kind == WR_PK_SOLID && ((kbf >> 8) & 0xFF) == WR_BLEND_PREMULT && ((rb.z | rb.w) & 0xFF00FF00u) == 0) sh.bail = 1;
        else if (kind != WR_PK_SOLID_FOLDED) { qa.y = 0; full = false; }      // (draws nothing here: reported by the setup stage, as on the pixel walk)
        ra = qa; rb = qb;
      }
      sh.rec[0][sl] = ra; sh.rec[1][sl] = rb;
    } else if (!DEPTH && sl < ((total + 3) & ~3)) {
      sh.rec[0][sl] = make_uint4(0, 0, 0, 0); sh.rec[1][sl] = make_uint4(0, 0, 0, 0);      // (the walk takes four prims per trip: empty rects)
    }
    (void)full;
  }
  __syncthreads();
  WR_CT(2);
  const unsigned long long colm = __ballot(sh.colflag[lane] != 0), rowm = __ballot(sh.rowflag[lane] != 0);
  const int nx = __popcll(colm) + 1, ny = __popcll(rowm) + 1;
  const int ncell = nx * ny;
  const bool cells_ok = try_cells && ncell <= 256 && !sh.bail;
  // More cells than lanes (a bin crossed by dozens of small rects), or pixels / depth to load: the pixel walk -- over the
  // list that is in LDS now (return 3), which spares each of the four waves the scan of the bin's mask words (a target of
  // 100 k prims has 1563 of them per bin, a few dozen set bits in all) and the per-word record fetches.  Depth-tested launches
  // only: the depth-less ones hold the records in their folded form.
  if (!cells_ok && !DEPTH) return 0;
  // (self-cleaning bin masks: every wave read the bin's words before the first barrier; only the words the list names are set)
  for (int sl = threadIdx.x; sl < total; sl += (int)blockDim.x) mw[(sh.pid[sl] - T.first_prim) >> 6] = 0ull;
  if (!cells_ok) { list_total = total; return 3; }
  // first column / row of every class (a table per wave: no workgroup barrier)
  if (((colm | 1ull) >> lane) & 1ull) sh.colstart[role][lane ? __popcll(colm & ((1ull << lane) - 1ull)) + 1 : 0] = (uint8_t)lane;
  if (((rowm | 1ull) >> lane) & 1ull) sh.rowstart[role][lane ? __popcll(rowm & ((1ull << lane) - 1ull)) + 1 : 0] = (uint8_t)lane;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- B: one cell per lane, the prim list in order ----------------------------
  if (role * 64 < ncell) {
    const int id = role * 64 + lane;
    const bool active = id < ncell;
    const int iy = active ? id / nx : 0, ix = active ? id - iy * nx : 0;
    const int cx = bx0 + sh.colstart[role][ix], cy = by0 + sh.rowstart[role][iy];
    // rows of the bin this wave's cells lie in: [yb0, yb1)
    const int iy_lo = (role * 64) / nx, iy_hi = (wr_imin(role * 64 + 64, ncell) - 1) / nx;
    const int yb0 = DEPTH ? by0 + sh.rowstart[role][iy_lo] : 0, yb1 = DEPTH ? (iy_hi + 1 < ny ? by0 + sh.rowstart[role][iy_hi + 1] : by0 + WR_BIN_H) : 0;
    uint32_t lo = T.init_color & WR_M8, hi = (T.init_color >> 8) & WR_M8, dp = T.init_depth;
    if constexpr (!DEPTH) {
      // The walk: every prim of the bin, in order, four per trip, branch-free: a prim's fold constants and rect come out of
      // LDS at a wave-uniform address (the next four requested before these are applied), every lane folds, and the lanes
      // whose cell lies outside the rect keep their value.  One wave's dependent chain through ~40 prims is what a bin
      // waits for, so the trip is straight-line code: no EXEC juggling, no scalar branches but the loop's own.
      // (An idle lane's cell sits at x = INT_MAX: inside no rect.)
      const int cxi = active ? cx : 0x7fffffff;
      // (The reads are spelled out: with a provably uniform address the compiler splits each 16-byte read into one ds_read_b32
      // per used dword, each with its own address register, and sinks them to their first use.)
      typedef uint32_t wr_u32x4 __attribute__((ext_vector_type(4)));
      uint32_t va = (uint32_t)(uintptr_t)&sh.rec[0][0];          // (a flat LDS address: the low half is the LDS offset)
      constexpr int QOFF = (int)sizeof(uint4) * (WR_CELL_MAX_PRIMS + 4);       // rec[1] - rec[0]
      const int ntrip = (total + 1) >> 1;          // two prims per trip
#define WR_CELL_LOAD(S_)                                                                                        \
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:%5\n\tds_read_b128 %3, %4 offset:%6" \
                   : "=&v"(r0##S_), "=&v"(r1##S_), "=&v"(q0##S_), "=&v"(q1##S_) : "v"(va), "n"(QOFF), "n"(QOFF + 16));   \
      va += 32;
#define WR_CELL_WAIT(S_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0##S_), "+v"(r1##S_), "+v"(q0##S_), "+v"(q1##S_));
#define WR_CELL_APPLY(q_, r_)                                                                                  \
      { const bool in = (unsigned)(cxi - (int)r_.x) < r_.y && (unsigned)(cy - (int)r_.z) < r_.w;                 \
        const uint32_t nl = wr_hi_bytes(wr_mul24(lo, q_.x) + q_.y), nh = wr_hi_bytes(wr_mul24(hi, q_.x) + q_.z); \
        lo = in ? nl : lo; hi = in ? nh : hi; }
#define WR_CELL_APPLY2(S_) WR_CELL_APPLY(q0##S_, r0##S_) WR_CELL_APPLY(q1##S_, r1##S_)
      wr_u32x4 q0A, q1A, r0A, r1A, q0B, q1B, r0B, r1B;
      WR_CELL_LOAD(A)
      for (int tr = 0; tr < ntrip; tr += 2) {
        WR_CELL_WAIT(A)
        WR_CELL_LOAD(B)              // (past the end of the list: the padding behind it)
        WR_CELL_APPLY2(A)
        if (tr + 1 >= ntrip) break;
        WR_CELL_WAIT(B)
        WR_CELL_LOAD(A)
        WR_CELL_APPLY2(B)
      }
      asm volatile("s_waitcnt lgkmcnt(0)");        // (the request in flight past the end)
#undef WR_CELL_LOAD
#undef WR_CELL_WAIT
#undef WR_CELL_APPLY2
#undef WR_CELL_APPLY
    } else
    for (int c0 = 0; c0 < total; c0 += 64) {
      uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
      if (c0 + lane < total) { ra = sh.rec[0][c0 + lane]; rb = sh.rec[1][c0 + lane]; }
      // (lane b holds prim c0 + b: the ones that reach this wave's rows)
      unsigned long long live = __ballot(c0 + lane < total && (int)ra.w > yb0 && (int)ra.y < yb1);
      while (live) {
        const int b = __builtin_ctzll(live);
        live &= live - 1ull;
        const int x0 = __builtin_amdgcn_readlane((int)ra.x, b), y0 = __builtin_amdgcn_readlane((int)ra.y, b);
        const int x1 = __builtin_amdgcn_readlane((int)ra.z, b), y1 = __builtin_amdgcn_readlane((int)ra.w, b);
        const uint32_t z = (uint32_t)__builtin_amdgcn_readlane((int)rb.x, b), kbf = (uint32_t)__builtin_amdgcn_readlane((int)rb.y, b);
        const uint32_t c0_ = (uint32_t)__builtin_amdgcn_readlane((int)rb.z, b), c1_ = (uint32_t)__builtin_amdgcn_readlane((int)rb.w, b);
        const uint32_t kind = kbf & 0xFF, blend = (kbf >> 8) & 0xFF, flags = (kbf >> 16) & 0xFF;
        bool in = active && (unsigned)(cx - x0) < (unsigned)(x1 - x0) && (unsigned)(cy - y0) < (unsigned)(y1 - y0);
        if (kind == WR_PK_CLEAR) {
          if (flags & WR_PF_CLEAR_COLOR) { lo = in ? (c0_ & WR_M8) : lo; hi = in ? ((c0_ >> 8) & WR_M8) : hi; }
          if (DEPTH && (flags & WR_PF_CLEAR_DEPTH)) dp = in ? z : dp;
          continue;
        }
        if (DEPTH && (flags & WR_PF_DEPTH_TEST)) {
          in = in && ((flags & WR_PF_DEPTH_LESS) ? (z < dp) : (z <= dp));
          if (flags & WR_PF_DEPTH_WRITE) dp = in ? z : dp;
        }
        if (kind == WR_PK_SOLID_FOLDED) {
          wr_fold_masked(lo, hi, kbf >> 24, c0_, c1_, WR_LANEMASK(in));
        } else if (kind == WR_PK_SOLID && blend == WR_BLEND_PREMULT && ((c0_ | c1_) & 0xFF00FF00u) == 0) {
          const uint32_t K = 255u - (c1_ >> 16);
          const uint32_t ulo = (c0_ & 0xFFFF) | ((c1_ & 0xFFFF) << 16), uhi = (c0_ >> 16) | (c1_ & 0xFFFF0000u);
          const uint32_t nl = wr_pk_min_u16(wr_hi_bytes(wr_mul24(lo, K) + WR_M8) + ulo, WR_M8);
          const uint32_t nh = wr_pk_min_u16(wr_hi_bytes(wr_mul24(hi, K) + WR_M8) + uhi, WR_M8);
          lo = in ? nl : lo; hi = in ? nh : hi;
        }
        // (any other kind draws nothing in a rect-only launch: it was reported by the setup stage, as on the pixel walk)
      }
    }
    if (active) { sh.color[id] = lo | (hi << 8); if (DEPTH) sh.dep[id] = dp; }
  }
  WR_CT(3);
  __syncthreads();
  WR_CT(4);
  // ---- C: expand ---------------------------------------------------------------
  {
    const int lx = (lane & 15) * 4, ly = wave * 16 + (lane >> 4);
    int cxi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cxi[i] = __popcll(colm & ((2ull << (lx + i)) - 1ull));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int rc = __popcll(rowm & ((2ull << (ly + 4 * j)) - 1ull)) * nx;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int id = rc + cxi[i];
        const uint32_t v = sh.color[id];
        plo[4 * j + i] = v & WR_M8; phi[4 * j + i] = (v >> 8) & WR_M8;
        if (DEPTH) dep[4 * j + i] = sh.dep[id];
      }
    }
  }
  return 1;
}
#endif

template <int FMT, bool DEPTH, int R, int FEAT>
WR_DEVICE void wr_raster_body(const WrTargetDesc* __restrict__ targets, int n_targets,
                 const WrDrawDesc* __restrict__ draws, const WrPrim* __restrict__ prims,
                 const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, const float* __restrict__ vtab,
                 unsigned long long* __restrict__ masks, const int bin, const int part = 0, const int parts = 1) {
  constexpr int NPX = 4 * R, STRIP = 4 * R;
  WR_CT(7);
  // the target this bin belongs to: the last one whose first bin is not beyond it
  int t = 0;
#ifdef WRHIP_HOSTSIM
  {
    int lo = 0, hi = n_targets - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (targets[mid].first_bin <= bin) lo = mid; else hi = mid - 1;
    }
    t = lo;
  }
#else
  // (the lanes look at 64 targets at a time: one round trip for a frame's tiles instead of a binary search's five dependent ones)
  for (int tb = 0; tb < n_targets; tb += 64) {
    const int tl = tb + (int)(threadIdx.x & 63);
    const bool le = tl < n_targets && targets[tl].first_bin <= bin;
    const int c = __popcll(__ballot(le));
    t += c;
    if (c < 64) break;
  }
  t = t > 0 ? t - 1 : 0;
#endif
  WR_CT(8);
  const WrTargetDesc& T = targets[t];
  if (T.format != FMT) return;
  const int lb = bin - T.first_bin;
  const int bx = lb % T.bins_x, by = lb / T.bins_x;
  if ((by + 1) * WR_BIN_H <= T.y_begin || by * WR_BIN_H >= T.y_end) return;      // rows of another rank (the setup stage bins nothing there)
  // the wave index is uniform across the wave: say so, or everything derived
  // from it (strip origin, coverage class of a prim) is treated as divergent
  // (`parts` > 1, thin R8 launches only: the bin's sixteen strips are dealt out to `parts` workgroups of 16 / parts waves, so that
  // each wave has a SIMD's issue slots to itself -- a thin launch is a few dozen waves each running one long instruction stream,
  // and the four waves a 1024-thread workgroup puts on every SIMD take turns at its one vector issue port)
#ifdef WRHIP_HOSTSIM
  const int lwave = threadIdx.x >> 6;
#else
  const int lwave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
  const int wave = lwave + part * (int)(blockDim.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wx0 = bx * WR_BIN_W, wy0 = by * WR_BIN_H + wave * STRIP;
  const int px = wx0 + (lane & 15) * 4;
  const int py = wy0 + (lane >> 4);
  constexpr int BPP = FMT == WR_FMT_RGBA8 ? 4 : 1;
  const bool vec_ok = BPP == 4 && px + 4 <= T.width && ((T.stride & 15) == 0);

  // RGBA8: lo/hi channel pairs; R8: value in lo
  uint32_t plo[NPX], phi[NPX], dep[NPX];
  unsigned long long* mw = masks + (size_t)T.word_base + (size_t)lb * T.words_per_bin;
  // (rect-only bins that start from a clear: the cell raster, when the bin's prims cut it into few enough cells)
  bool cells_done = false, empty_bin = false;      // (empty_bin: the cell raster found no prim in the bin's words)
  int list_total = 0;                              // > 0: the bin's prim list sits in LDS (WrCellShared::pid / rec), its mask words are zeroed
#ifndef WRHIP_HOSTSIM
  constexpr bool CELLS = FMT == WR_FMT_RGBA8 && FEAT == 0 && R == 4;
  constexpr size_t LDS_BYTES = CELLS && sizeof(WrCellShared) > sizeof(int) * 16 * 64 ? sizeof(WrCellShared) : sizeof(int) * 16 * 64;
  __shared__ uint4 lds_raw[LDS_BYTES / 16];           // the cell raster's tables, or the pixel walk's compaction rows
  if constexpr (CELLS) {
    WrCellShared& cell_sh = *(WrCellShared*)lds_raw;
    const bool try_cells = !T.load_color && !(DEPTH && T.load_depth && T.depth);
    if (T.cells && (try_cells || DEPTH))
    {
      // (the same in every lane -- decided on LDS contents behind barriers -- but only provably so once it is said)
      int lt = 0;
      const int cr = __builtin_amdgcn_readfirstlane(wr_raster_cells<DEPTH>(cell_sh, T, recs, mw, T.words_per_bin, wx0, by * WR_BIN_H, wave, lane, plo, phi, dep, try_cells, lt));
      cells_done = cr == 1; empty_bin = cr == 2;
      if (cr == 3) list_total = __builtin_amdgcn_readfirstlane(lt);
    }
  }
#endif
  WR_CT(5);
  // ---- initial pixel state ---------------------------------------------
  if (!cells_done) {
#pragma unroll
  for (int j = 0; j < R; j++) {
    const int y = py + 4 * j;
    uint32_t c[4] = {T.init_color, T.init_color, T.init_color, T.init_color};
    if (T.load_color && y < T.height) {
      const uint8_t* rowp = (const uint8_t*)T.color + (size_t)y * T.stride;
      if (BPP == 4) {
        if (vec_ok) {
          uint4 v = *(const uint4*)(rowp + (size_t)px * 4);
          c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) if (px + i < T.width) c[i] = ((const uint32_t*)rowp)[px + i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) if (px + i < T.width) c[i] = rowp[px + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (BPP == 4) { plo[4 * j + i] = c[i] & WR_M8; phi[4 * j + i] = (c[i] >> 8) & WR_M8; }
      else { plo[4 * j + i] = c[i]; phi[4 * j + i] = 0; }
      dep[4 * j + i] = T.init_depth;
    }
    if (DEPTH && T.load_depth && T.depth && y < T.height) {
#pragma unroll
      for (int i = 0; i < 4; i++) if (px + i < T.width) dep[4 * j + i] = T.depth[(size_t)y * T.width + px + i];
    }
  }
  }
  // ---- apply every prim of this bin, in submission order -----------------
  // (first depth-tested perspective prim of the target: later depth-tested prims look their rows up in T.flat_rows)
  const uint32_t flat_first = (DEPTH && FEAT != 0 && T.flat_rows) ? T.flat_rows[T.height] : 0xFFFFFFFFu;
  const WrGlyphRec* const grecs = T.grecs;      // (read once: left to the compiler, the glyph walk re-reads the descriptor word per glyph)
  uint32_t zcap = (DEPTH && !(T.load_depth && T.depth)) ? T.init_depth : 0xFFFFFFFFu;
#ifdef WRHIP_HOSTSIM
  for (int w = 0; w < T.words_per_bin; w++) {
    // serial reference iteration (host simulation has no cross-lane ops)
    unsigned long long live = mw[w];
    const int base = T.first_prim + w * 64;
    while (live) {
      const int bit = __builtin_ctzll(live);
      live &= live - 1;
      const WrRec Rc = recs[base + bit];
      if (Rc.x1 <= wx0 || Rc.x0 >= wx0 + WR_BIN_W || Rc.y1 <= wy0 || Rc.y0 >= wy0 + STRIP) continue;
      if constexpr (DEPTH) {
        if (wr_zcap_rejects(Rc.kbf, Rc.z, zcap)) continue;
        zcap = wr_zcap_after(Rc.kbf, Rc.z, zcap, Rc.x0 <= wx0 && Rc.x1 >= wx0 + WR_BIN_W && Rc.y0 <= wy0 && Rc.y1 >= wy0 + STRIP);
      }
      const int rblend = (Rc.kbf >> 8) & 0xFF;
      const WrRuns* rr = nullptr;
      if constexpr (DEPTH && FMT == WR_FMT_RGBA8 && FEAT != 0) {
        if (((Rc.kbf >> 16) & WR_PF_DEPTH_TEST) && wr_kind_needs_runs(Rc.kbf & 0xFF) && ((T.dw_end > T.dw_first && base + bit > T.dw_first) || T.load_depth || (uint32_t)(base + bit) > flat_first))
          rr = wr_build_runs<R>(T, recs, aux, base + bit, Rc.x0, Rc.y0, Rc.x1, Rc.y1, Rc.z, Rc.kbf, wy0, lane, wave);
      }
      if ((FEAT & WR_FEAT_R8TEX) && FMT == WR_FMT_RGBA8 && ((Rc.kbf & 0xFF) == WR_PK_SOLID_MASKED || ((Rc.kbf & 0xFF) == WR_PK_TEX_R8 && !rr)) && (rblend == WR_BLEND_NONE || rblend == WR_BLEND_PREMULT) &&
          aux[base + bit].tex.simple)
        wr_apply_tex_r8<DEPTH, R>(plo, phi, dep, Rc.x0, Rc.y0, Rc.x1, Rc.y1, Rc.z, Rc.kbf, Rc.c0, Rc.c1, aux[base + bit].tex,
                                  draws, &prims[base + bit], px, py);
      else
        wr_apply_prim<FMT, DEPTH, R, FEAT>(plo, phi, dep, Rc.x0, Rc.y0, Rc.x1, Rc.y1, Rc.z, Rc.kbf, Rc.c0, Rc.c1, &prims[base + bit],
                                     &aux[base + bit], draws, vtab, px, py, wx0, wy0, rr);
    }
  }
#else
  if (!cells_done) {
  // Wave-cooperative fetch: lane l loads the record of the word's l-th prim
  // (one memory latency for up to 64 prims, the next word's records are
  // requested before the current word is processed), tests it against this
  // wave's strip, and the survivors -- a ballot mask, still in submission
  // order -- are broadcast one by one with v_readlane into SGPRs.
  // One round: up to 64 prims, lane i holding the record of prim `pid` (-1: none); in a dense round (sp false) lane i's prim is
  // dbase + i.  Prim index of the round's lane `b_` (uniform b_):
#define WR_PID(b_) (sp ? __builtin_amdgcn_readlane(pid, (b_)) : dbase + (b_))
  // (a macro: the rect-only variant expands it in place -- wrapped in a lambda its tile pass was 4 % slower, cfg2 --, the other
  // variants call it through an inlined lambda -- expanded in place the glyph variant's was 4 % slower, cfg3)
#define WR_ROUND_BODY                                                                                                                                                                                   \
    const bool has = pid >= 0;                                                                                                                                                                          \
    bool hit = has && !((int)ra.z <= wx0 || (int)ra.x >= wx0 + WR_BIN_W || (int)ra.w <= wy0 || (int)ra.y >= wy0 + STRIP);                                                                               \
    /* (prims the strip's depth cap already rejects never reach the scalar walk: 64 of them per compare; the walk re-tests */                                                                           \
    /* the survivors, since the cap may move while the round is applied) */                                                                                                                             \
    if constexpr (DEPTH) hit = hit && !wr_zcap_rejects(rb.y, rb.x, zcap);                                                                                                                               \
    unsigned long long live = __ballot(hit);                                                                                                                                                            \
    /* prims of this word that are unit glyph blits (lane i looks at prim i): runs of them are applied lane by lane */                                                                                  \
    unsigned long long glyphs = 0ull;                                                                                                                                                                   \
    if constexpr ((FEAT & WR_FEAT_R8TEX) != 0 && FMT == WR_FMT_RGBA8) {                                                                                                                                 \
      const uint32_t k8 = rb.y & 0xFF, b8 = (rb.y >> 8) & 0xFF;                                                                                                                                         \
      bool g = hit && (k8 == WR_PK_TEX_R8 || k8 == WR_PK_SOLID_MASKED) && (b8 == WR_BLEND_NONE || b8 == WR_BLEND_PREMULT) &&                                                                            \
               !(DEPTH && ((rb.y >> 16) & WR_PF_DEPTH_TEST));                                                                                                                                           \
      if (g) g = grecs != nullptr && (((const uint32_t*)&grecs[pid])[7] & 1u) != 0u;                                                                                                                      \
      glyphs = __ballot(g);                                                                                                                                                                             \
    }                                                                                                                                                                                                   \
    /* Rect-only launches (FEAT == 0): the survivors' records come back through the scalar cache, one s_load_dwordx8 per prim */                                                                        \
    /* issued a prim ahead, instead of eight v_readlane broadcasts out of the lanes that tested them -- the blend loop is */                                                                            \
    /* VALU-bound and v_readlane is VALU (cfg2 tile pass 68.6 -> 61.1 us).  The index is wave-uniform and recs[] read-only. */                                                                          \
    /* The textured variants keep the broadcasts (measured: the extra scalar round trip costs them 3 %). */                                                                                             \
    constexpr bool SCALAR_RECS = FEAT == 0;                                                                                                                                                             \
    int nbit = live ? __builtin_ctzll(live) : 0;                                                                                                                                                        \
    WrRec nrec;                                                                                                                                                                                         \
    if (SCALAR_RECS) nrec = recs[WR_PID(nbit)];                                                                                                                                                         \
    while (live) {                                                                                                                                                                                      \
      if constexpr ((FEAT & WR_FEAT_R8TEX) != 0 && FMT == WR_FMT_RGBA8) {                                                                                                                               \
        if ((glyphs >> __builtin_ctzll(live)) & 1ull) {                                                                                                                                                 \
    /* the run of surviving prims from here up to the next one that is not a unit glyph */                                                                                                              \
          const unsigned long long others = live & ~glyphs;                                                                                                                                             \
          const unsigned long long run = others ? (live & ((others & (0ull - others)) - 1ull)) : live;                                                                                                  \
          if (run & (run - 1ull)) { /* two or more: worth the per-lane lists */                                                                                                                         \
            live &= ~run;                                                                                                                                                                               \
    /* which prims of the run reach this lane's footprint (columns px .. px+3, rows py, py+4, ..) */                                                                                                    \
            unsigned mlo = 0, mhi = 0;                                                                                                                                                                  \
            for (unsigned long long rr_ = run; rr_; rr_ &= rr_ - 1ull) {                                                                                                                                \
              const int b = __builtin_ctzll(rr_);                                                                                                                                                       \
              const int gx0 = __builtin_amdgcn_readlane((int)ra.x, b), gy0 = __builtin_amdgcn_readlane((int)ra.y, b);                                                                                   \
              const int gx1 = __builtin_amdgcn_readlane((int)ra.z, b), gy1 = __builtin_amdgcn_readlane((int)ra.w, b);                                                                                   \
              const bool reach = px + 3 >= gx0 && px < gx1 && py + 4 * (R - 1) >= gy0 && py < gy1;                                                                                                      \
              if (b < 32) mlo |= reach ? (1u << b) : 0u; else mhi |= reach ? (1u << (b - 32)) : 0u;                                                                                                     \
            }                                                                                                                                                                                           \
            while (__ballot((mlo | mhi) != 0)) {                                                                                                                                                        \
              const bool act = (mlo | mhi) != 0;                                                                                                                                                        \
              int b = lane;                                                                                                                                                                             \
              if (act) {                                                                                                                                                                                \
                b = mlo ? __builtin_ctz(mlo) : 32 + __builtin_ctz(mhi);                                                                                                                                 \
                if (mlo) mlo &= mlo - 1; else mhi &= mhi - 1;                                                                                                                                           \
              }                                                                                                                                                                                         \
    /* (the shuffle runs with every lane active: a lane that has no glyph left may hold the index another one asks for) */                                                                              \
              const int gp = sp ? __shfl(pid, b) : dbase + b;                                                                                                                                           \
              if (act) {                                                                                                                                                                                \
    /* (requesting a lane's next record ahead of applying its current one was tried twice -- with the WrRec + WrTexRec pair the 16 */                                                                   \
    /* extra live VGPRs spilled, 170 -> 420 us; with the 32-byte glyph record: 110 -> 112 us, cfg3 -- and is not done; nor is */                                                                        \
    /* handing the records from lane to lane with ds_bpermutes instead of fetching them: 112 -> 118 us) */                                                                                              \
                const uint4* gr_ = (const uint4*)&grecs[gp];                                                                                                                                            \
                const uint4 ga_ = gr_[0], gb_ = gr_[1];                                                                                                                                                 \
                wr_unit_glyph_lane<R>(plo, phi, ga_, gb_, &grecs[gp], px, py);                                                                                                                               \
              }                                                                                                                                                                                         \
            }                                                                                                                                                                                           \
            continue;                                                                                                                                                                                   \
          }                                                                                                                                                                                             \
        }                                                                                                                                                                                               \
      }                                                                                                                                                                                                 \
      const int bit = SCALAR_RECS ? nbit : __builtin_ctzll(live);                                                                                                                                       \
      live &= live - 1;                                                                                                                                                                                 \
      int x0, y0, x1, y1;                                                                                                                                                                               \
      uint32_t z, kbf, c0, c1;                                                                                                                                                                          \
      if (SCALAR_RECS) {                                                                                                                                                                                \
        const WrRec Rc = nrec;                                                                                                                                                                          \
        nbit = live ? __builtin_ctzll(live) : bit;                                                                                                                                                      \
        nrec = recs[WR_PID(nbit)];                                                                                                                                                                      \
        x0 = Rc.x0; y0 = Rc.y0; x1 = Rc.x1; y1 = Rc.y1; z = Rc.z; kbf = Rc.kbf; c0 = Rc.c0; c1 = Rc.c1;                                                                                                 \
      } else {                                                                                                                                                                                          \
        x0 = __builtin_amdgcn_readlane((int)ra.x, bit); y0 = __builtin_amdgcn_readlane((int)ra.y, bit);                                                                                                 \
        x1 = __builtin_amdgcn_readlane((int)ra.z, bit); y1 = __builtin_amdgcn_readlane((int)ra.w, bit);                                                                                                 \
        z = __builtin_amdgcn_readlane((int)rb.x, bit); kbf = __builtin_amdgcn_readlane((int)rb.y, bit);                                                                                                 \
        c0 = __builtin_amdgcn_readlane((int)rb.z, bit); c1 = __builtin_amdgcn_readlane((int)rb.w, bit);                                                                                                 \
      }                                                                                                                                                                                                 \
      if constexpr (DEPTH) {                                                                                                                                                                            \
        if (wr_zcap_rejects(kbf, z, zcap)) continue;                                                                                                                                                    \
        zcap = wr_zcap_after(kbf, z, zcap, x0 <= wx0 && x1 >= wx0 + WR_BIN_W && y0 <= wy0 && y1 >= wy0 + STRIP);                                                                                        \
      }                                                                                                                                                                                                 \
      const int rblend = (kbf >> 8) & 0xFF;                                                                                                                                                             \
      const int pi = WR_PID(bit);                                                                                                                                                                       \
      const WrRuns* rr = nullptr;                                                                                                                                                                       \
      if constexpr (DEPTH && FMT == WR_FMT_RGBA8 && FEAT != 0) {                                                                                                                                        \
        if (((kbf >> 16) & WR_PF_DEPTH_TEST) && wr_kind_needs_runs(kbf & 0xFF) && ((T.dw_end > T.dw_first && pi > T.dw_first) || T.load_depth || (uint32_t)pi > flat_first))                                                         \
          rr = wr_build_runs<R>(T, recs, aux, pi, x0, y0, x1, y1, z, kbf, wy0, lane, wave, list_total > 0 ? nullptr : mw, wx0);                                                                         \
      }                                                                                                                                                                                                 \
      if ((FEAT & WR_FEAT_R8TEX) && FMT == WR_FMT_RGBA8 && ((kbf & 0xFF) == WR_PK_SOLID_MASKED || ((kbf & 0xFF) == WR_PK_TEX_R8 && !rr)) && (rblend == WR_BLEND_NONE || rblend == WR_BLEND_PREMULT) &&  \
          aux[pi].tex.simple)                                                                                                                                                                           \
        wr_apply_tex_r8<DEPTH, R>(plo, phi, dep, x0, y0, x1, y1, z, kbf, c0, c1, aux[pi].tex, draws, &prims[pi], px, py);                                                                               \
      else                                                                                                                                                                                              \
        wr_apply_prim<FMT, DEPTH, R, FEAT>(plo, phi, dep, x0, y0, x1, y1, z, kbf, c0, c1, &prims[pi], &aux[pi], draws, vtab, px, py, wx0, wy0, rr);                                                     \
    }                                                                                                                                                                                                   \
  (void)0
  auto do_round = [&](const int pid, const int dbase, const bool sp, const uint4 ra, const uint4 rb) __attribute__((always_inline)) {
    WR_ROUND_BODY;
  };
  const int nw = (empty_bin || list_total > 0) ? 0 : T.words_per_bin;      // (an empty bin / a bin whose list is in LDS: no words to walk or zero)
  if constexpr (CELLS && DEPTH) {
    if (list_total > 0) {
      // the bin's prims out of LDS, 64 per round (lane i: the list's entry c0 + i), in submission order
      const WrCellShared& L = *(const WrCellShared*)lds_raw;
      for (int c0 = 0; c0 < list_total; c0 += 64) {
        int pid = -1;
        uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
        if (c0 + lane < list_total) { pid = L.pid[c0 + lane]; ra = L.rec[0][c0 + lane]; rb = L.rec[1][c0 + lane]; }
        do_round(pid, 0, true, ra, rb);
      }
    }
  }
  if constexpr (FMT == WR_FMT_RGBA8 && FEAT == 0 && !DEPTH) {
    // The rect-only, depth-less variant runs at 8 waves per SIMD on 64 VGPRs with nothing to spare: it keeps the plain walk --
    // the bin's mask words fetched 64 at a time (lane l loads word l of the block), the non-zero ones visited one per round,
    // the next word's records requested before the current word is processed (the block walk below costs it 8 %, cfg2).
    for (int wb = 0; wb < nw; wb += 64) {
      const unsigned long long mv = wb + lane < nw ? mw[wb + lane] : 0ull;
      unsigned long long nz = __ballot(mv != 0ull);
      if (!nz) continue;
      unsigned long long m_next;
      uint4 na = make_uint4(0, 0, 0, 0), nb = make_uint4(0, 0, 0, 0);
      {
        const int cw = __builtin_ctzll(nz);
        m_next = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv, cw) |
                 ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mv >> 32), cw) << 32);
        if ((m_next >> lane) & 1ull) {
          const uint4* rp = (const uint4*)&recs[T.first_prim + (wb + cw) * 64 + lane];
          na = rp[0]; nb = rp[1];
        }
      }
      while (nz) {
        const int w = wb + __builtin_ctzll(nz);
        nz &= nz - 1ull;
        const unsigned long long m = m_next;
        const uint4 ra = na, rb = nb;
        const int base = T.first_prim + w * 64;
        if (nz) {
          const int cw = __builtin_ctzll(nz);
          m_next = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv, cw) |
                   ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mv >> 32), cw) << 32);
          if ((m_next >> lane) & 1ull) {
            const uint4* rp = (const uint4*)&recs[T.first_prim + (wb + cw) * 64 + lane];
            na = rp[0]; nb = rp[1];
          }
        }
        {
          const int pid = ((m >> lane) & 1ull) ? base + lane : -1, dbase = base;
          const bool sp = false;
          WR_ROUND_BODY;
        }
      }
    }
  } else {
  // The bin's mask words are themselves fetched 64 at a time (lane l loads word l of the block, the next block is requested
  // before this one is walked) and only the non-zero ones are visited.  A block whose set bits are thinly spread -- a target
  // with 100 k prims has ~1500 words per bin of which a few dozen hold one bit each (cfg5) -- is compacted first: a wave
  // prefix sum over the words' popcounts gives every set bit a slot, the lanes scatter their prim indices into a 64-entry
  // LDS row of the wave, and one record fetch + ballot serves up to 64 prims of up to 4096 consecutive ones instead of one
  // fetch per word.  Slots are in (word, bit) order, so submission order is kept.  Dense blocks keep the word-per-round walk.
  int (*pid_row)[64] = (int (*)[64])lds_raw;
  int wb = 0, wb_next = 0;                       // block being walked / block whose words are in mv_next
  unsigned long long mv = 0ull, mv_next = lane < nw ? mw[lane] : 0ull;
  unsigned long long nz = 0ull;                  // dense walk: non-zero words of the block still to visit
  int blk_rounds = 0, blk_round = 0, total = 0, prefix = 0;
  bool sparse = false;
  // next round: this lane's prim index (or -1), in dbase_ the index lane 0 would have in a dense round
  auto next_round = [&](int& pid_, int& dbase_, bool& sp_) -> bool {
    for (;;) {
      if (blk_rounds == 0) {
        if (wb_next >= nw) return false;
        mv = mv_next; wb = wb_next; wb_next += 64;
        mv_next = wb_next + lane < nw ? mw[wb_next + lane] : 0ull;
        nz = __ballot(mv != 0ull);
        if (!nz) continue;
        const int cnt = __popcll(mv);
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        prefix = inc - cnt;
        total = __builtin_amdgcn_readlane(inc, 63);
        const int nzw = __popcll(nz), rounds = (total + 63) >> 6;
        sparse = rounds * 2 <= nzw;
        blk_rounds = sparse ? rounds : nzw;
        blk_round = 0;
      }
      blk_rounds--;
      sp_ = sparse;
      if (sparse) {
        const int s0 = prefix - blk_round * 64;            // slot of this lane's first bit, relative to the round
        if (s0 < 64 && s0 + __popcll(mv) > 0) {
          int sl = s0;
          for (unsigned long long bts = mv; bts; bts &= bts - 1ull, sl++)
            if ((unsigned)sl < 64u) pid_row[lwave][sl] = T.first_prim + (wb + lane) * 64 + __builtin_ctzll(bts);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pid_ = lane < total - blk_round * 64 ? pid_row[lwave][lane] : -1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dbase_ = 0;
      } else {
        const int cw = __builtin_ctzll(nz);
        nz &= nz - 1ull;
        const unsigned long long m_ = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv, cw) |
                                      ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mv >> 32), cw) << 32);
        dbase_ = T.first_prim + (wb + cw) * 64;
        pid_ = ((m_ >> lane) & 1ull) ? dbase_ + lane : -1;
      }
      blk_round++;
      return true;
    }
  };
  int pid_n = -1, dbase_n = 0;
  bool sp_n = false;
  bool more = next_round(pid_n, dbase_n, sp_n);
  uint4 na = make_uint4(0, 0, 0, 0), nb = make_uint4(0, 0, 0, 0);
  if (more && pid_n >= 0) {
    const uint4* rp = (const uint4*)&recs[pid_n];
    na = rp[0]; nb = rp[1];
  }
  while (more) {
    const int pid = pid_n, dbase = dbase_n;
    const bool sp = sp_n;
    const uint4 ra = na, rb = nb;
    more = next_round(pid_n, dbase_n, sp_n);
    if (more && pid_n >= 0) {
      const uint4* rp = (const uint4*)&recs[pid_n];
      na = rp[0]; nb = rp[1];
    }
    do_round(pid, dbase, sp, ra, rb);
  }
  }
#undef WR_ROUND_BODY
#undef WR_PID
  // Self-cleaning bin masks: once every wave of the workgroup has consumed the
  // bin's words, zero them so the next flush needs no memset launch.
  __syncthreads();
  bool clean = true;
  if (parts > 1) {
    // several workgroups read this bin's words: the last one to have walked them cleans up (T.bin_ctr: zero between launches)
    __shared__ int last_part;
    if (threadIdx.x == 0) {
      const unsigned seen = atomicAdd(&T.bin_ctr[lb], 1u);
      last_part = seen == (unsigned)(parts - 1);
      if (last_part) T.bin_ctr[lb] = 0u;
    }
    __syncthreads();
    clean = last_part != 0;
  }
  if (clean)
  for (int w = threadIdx.x; w < nw; w += (int)blockDim.x) if (mw[w]) mw[w] = 0ull;      // (most words of a large target are empty already)
  }
#endif
  // ---- write back ------------------------------------------------------------
  // (forwarded composite, WrTargetDesc::fwd_*: every row is stored a second time at its place in the target that would have
  // copied this one; everything about it but the row is uniform or per-lane constant)
  uint8_t* const fwd = BPP == 4 ? (uint8_t*)T.fwd_color : nullptr;
  const int fwd_stride = T.fwd_stride, fwd_y0 = T.fwd_y0, fwd_ys = T.fwd_ys;
  const int fwd_cx0 = T.fwd_clip[0], fwd_cy0 = T.fwd_clip[1], fwd_cx1 = T.fwd_clip[2], fwd_cy1 = T.fwd_clip[3];
  const int fX = px + T.fwd_dx;
  const bool fwd_vec = fX >= fwd_cx0 && fX + 4 <= fwd_cx1 && px + 4 <= T.width && ((fX & 3) == 0) && ((fwd_stride & 15) == 0);
#pragma unroll
  for (int j = 0; j < R; j++) {
    const int y = py + 4 * j;
    if (y >= T.height || y < T.y_begin || y >= T.y_end) continue;
    uint8_t* rowp = (uint8_t*)T.color + (size_t)y * T.stride;
    if (BPP == 4) {
      uint32_t c[4];
#pragma unroll
      for (int i = 0; i < 4; i++) c[i] = plo[4 * j + i] | (phi[4 * j + i] << 8);
      if (vec_ok) {
        *(uint4*)(rowp + (size_t)px * 4) = make_uint4(c[0], c[1], c[2], c[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) if (px + i < T.width) ((uint32_t*)rowp)[px + i] = c[i];
      }
      if (fwd) {
        // forwarded composite: the same pixels, a second time, at their place in the target that would have copied them
        const int Y = fwd_y0 + fwd_ys * y;
        if (Y >= fwd_cy0 && Y < fwd_cy1) {
          uint8_t* frow = fwd + (size_t)Y * fwd_stride;
          if (fwd_vec) {
            *(uint4*)(frow + (size_t)fX * 4) = make_uint4(c[0], c[1], c[2], c[3]);
          } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (fX + i >= fwd_cx0 && fX + i < fwd_cx1 && px + i < T.width) ((uint32_t*)frow)[fX + i] = c[i];
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) if (px + i < T.width) rowp[px + i] = (uint8_t)plo[4 * j + i];
    }
    if (DEPTH && T.store_depth && T.depth) {
#pragma unroll
      for (int i = 0; i < 4; i++) if (px + i < T.width) T.depth[(size_t)y * T.width + px + i] = dep[4 * j + i];
    }
  }
  WR_CT(6);
}

// Register budget: the textured RGBA8 variants need ~155-175 VGPRs, right at the 168 that still lets 3
// waves share a SIMD, and they are latency-bound (cfg3: 240 us at 3 waves, 340 us at 2).  The compiler
// is asked to hold 3 waves per SIMD for them; that is only a win while it has to spill a handful of
// cold values (forcing a 192-VGPR build of the glyph variant down cost 2x), so what their inline paths
// and their callees need is kept small: with interprocedural register allocation a caller keeps its
// live values above whatever its callees clobber (hence the integer wr_accum_binades, wr_accum_short on
// the glyph path, wr_aa_pixel_rgba8 out of line).  The rect-only variants ask for 8 waves (64 VGPRs) and, depth-tested,
// 4 (without a request that one drifted to 129 VGPRs = 3 waves and cfg5 lost 15 %).
// The R8 clip-mask variant: the nine-patch row function alone wants 228 VGPRs (one wave per SIMD, every latency of a
// launch of a few hundred workgroups exposed); now that solid groups never reach the row functions (wr_apply_prim), they
// may spill: 2 waves per SIMD are asked for (measured: 1 wave 243 us, 2 waves 145 us, 3 waves 155 us per mask launch of cfg4).
#ifndef WR_R8_CLIP_WAVES
#define WR_R8_CLIP_WAVES 2
#endif
#ifndef WR_TEX_WAVES
#define WR_TEX_WAVES 3          // waves per SIMD asked of the textured RGBA8 variants (168 VGPRs)
#endif
#ifdef WRHIP_HOSTSIM
#define WR_RASTER_BOUNDS(R, FMT, DEPTH, FEAT) __launch_bounds__(1024 / R)
#else
#ifndef WR_RECT_WAVES
#define WR_RECT_WAVES 8
#endif
#ifndef WR_SHADE_WAVES
#define WR_SHADE_WAVES 0        // waves per SIMD asked of the RGBA8 variants that carry the shader replays (FEAT 47); 0: no request
#endif
#define WR_RASTER_BOUNDS(R, FMT, DEPTH, FEAT) __launch_bounds__(((R) == 1 && (FMT) == WR_FMT_RGBA8) ? 256 : 1024 / R, ((R) == 1 && (FMT) == WR_FMT_RGBA8) ? 0 : ((FMT) == WR_FMT_RGBA8 && (FEAT) < 16) ? ((FEAT) != 0 ? WR_TEX_WAVES : ((DEPTH) ? 4 : WR_RECT_WAVES)) : ((FMT) == WR_FMT_R8 && ((FEAT) & WR_FEAT_CLIP) ? WR_R8_CLIP_WAVES : ((FMT) == WR_FMT_RGBA8 ? WR_SHADE_WAVES : 0)))
#endif
template <int FMT, bool DEPTH, int R, int FEAT>
__global__ void WR_RASTER_BOUNDS(R, FMT, DEPTH, FEAT)
wr_raster_kernel(const WrTargetDesc* __restrict__ targets, int n_targets,
                 const WrDrawDesc* __restrict__ draws, const WrPrim* __restrict__ prims,
                 const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, const float* __restrict__ vtab,
                 unsigned long long* __restrict__ masks, int bin_offset) {
  if constexpr (R == 1) {
    // thin launches: 1024 / blockDim.x workgroups per bin (see `parts` in wr_raster_body)
    const int parts = 1024 / (int)blockDim.x;
    wr_raster_body<FMT, DEPTH, R, FEAT>(targets, n_targets, draws, prims, recs, aux, vtab, masks, (int)blockIdx.x / parts + bin_offset, (int)blockIdx.x % parts, parts);
  } else
  wr_raster_body<FMT, DEPTH, R, FEAT>(targets, n_targets, draws, prims, recs, aux, vtab, masks, (int)blockIdx.x + bin_offset);
}

// A run of consecutive thin R8 levels in ONE launch (cfg4: corner-mask bins -> 2 x cs_scale -> cs_blur V / H are five
// dependent launches of 144 / 36 / 9 / 9 / 9 workgroups): a persistent grid of at most one workgroup per CU walks level
// after level, `bin = first + blockIdx.x, += gridDim.x`, with a grid-wide barrier in between -- an arrive counter in HBM
// that is never reset (the host knows its value at launch, and how many workgroups still take part in each level:
// launches of one stream are ordered), released / acquired at agent scope so a level's stores are written back from its
// XCD's L2 and the next level's loads do not hit stale lines.  Every workgroup is resident (grid <= CUs, 1024 threads at
// <= 128 VGPRs); a workgroup that nevertheless waits longer than a few seconds gives up, counts it
// (WrUnsupportedCounters::chain_timeout -> GL_INVALID_OPERATION at Finish) and carries on rather than hang the queue.
// MEASURED (cfg4, MI355X, profiles/r03_e_chain_ab.txt): bit-exact, but NOT faster -- 144 us for the five levels against
// ~100 us + four kernel boundaries as separate launches (3.18 k vs 3.47 k frames/s): the L2 write-back + invalidate a
// cross-XCD barrier needs costs what a kernel boundary costs (~13 us per level here; 217 us with all sixteen waves issuing
// the invalidate, 280 us with acquire-polling).  Kept behind WRHIP_CHAIN=1, off by default.
struct WrChain { int n; int first[WR_MAX_CHAIN]; int count[WR_MAX_CHAIN]; unsigned want[WR_MAX_CHAIN]; };   // want[l]: the arrive counter once level l is complete
template <int FEAT>
__global__ void __launch_bounds__(1024)
wr_raster_chain_kernel(const WrTargetDesc* __restrict__ targets, int n_targets,
                       const WrDrawDesc* __restrict__ draws, const WrPrim* __restrict__ prims,
                       const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, const float* __restrict__ vtab,
                       unsigned long long* __restrict__ masks, WrChain ch, unsigned* __restrict__ arrive,
                       WrUnsupportedCounters* __restrict__ cnt) {
#ifndef WRHIP_HOSTSIM
  for (int l = 0; l < ch.n; l++) {
    for (int b = (int)blockIdx.x; b < ch.count[l]; b += (int)gridDim.x) {
      wr_raster_body<WR_FMT_R8, false, 1, FEAT>(targets, n_targets, draws, prims, recs, aux, vtab, masks, ch.first[l] + b);
      __syncthreads();                      // (the body's LDS rows are reused by the next bin)
    }
    if (l + 1 == ch.n) break;
    // (a workgroup past its last bin -- the levels of a mask chain shrink: 144, 36, 9, 9, 9 bins -- arrives and leaves: a crowd of
    // idle workgroups polling one address slows the few that still work)
    bool more = false;
    for (int k = l + 1; k < ch.n; k++) more = more || (int)blockIdx.x < ch.count[k];
    __syncthreads();                        // every wave's stores are issued and waited for ...
    if (threadIdx.x == 0) {
      const unsigned want = ch.want[l];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                                     // ... and written back before the arrival shows
      __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      // (relaxed polls: an acquire per poll would invalidate the caches again and again -- 280 us per launch; ONE acquire follows
      // the loop, from this wave only -- the invalidate it issues covers the CU's vector cache and the XCD's L2, which the
      // other fifteen waves share: 217 -> 144 us)
      while (more && (int)(__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 21)) { atomicAdd(&cnt->chain_timeout, 1u); break; }
      }
      if (more) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (!more) return;                      // (the host counts only the workgroups that still take part in a level: ch.want)
    __syncthreads();
  }
#endif
}

// The setup stage of flush k+1 and the last raster level of flush k in ONE launch: the first
// `n_setup_blocks` workgroups run wr_setup_body, the rest rasterise.  The two are independent
// (different scratch sets, see Context::Tail in wrhip.hip), the setup stage is a dozen
// latency-bound workgroups, and a kernel boundary costs ~5 us on top: fused, the setup stage
// disappears behind the composite pass of the previous frame instead of standing between two
// frames.  Setup workgroups come first so they are dispatched first.
// (the setup-stage workgroups of a fused launch are its long pole: a dozen latency-bound waves among thousands of raster waves)
#if defined(WR_SETUP_PRIORITY) && !defined(WRHIP_HOSTSIM)
#define WR_SETUP_PRIO() __builtin_amdgcn_s_setprio(WR_SETUP_PRIORITY)
#else
#define WR_SETUP_PRIO() ((void)0)
#endif
struct WrSetupArgs {
  const WrDrawDesc* draws; int n_draws; const uint8_t* arena; WrPrim* prims; WrRec* recs; WrAux* aux; int n_prims;
  const WrTargetDesc* targets; unsigned long long* masks; float* vtab; WrUnsupportedCounters* cnt; const int* blk;
  // fused scatter: the batch's upload segments, run by the launch's first up_blocks workgroups (0: none)
  const WrUploadSeg* up_segs; int up_nseg, up_parts, up_blocks;
};
// the workgroup's role in a setup-carrying launch: scatter first, then the setup stage, then the launch's own work (returns the index within it)
#define WR_FUSED_PROLOGUE(S, n_setup_blocks)                                                                                      \
  int wr_bid = (int)blockIdx.x;                                                                                                   \
  if (wr_bid < (S).up_blocks) { wr_upload_body((S).up_segs, (S).up_nseg, (S).up_parts, wr_bid); return; }                         \
  wr_bid -= (S).up_blocks;                                                                                                        \
  if (wr_bid < (n_setup_blocks)) {                                                                                                \
    WR_SETUP_PRIO();                                                                                                              \
    wr_setup_body((S).draws, (S).n_draws, (S).arena, (S).prims, (S).recs, (S).aux, (S).n_prims, (S).targets, (S).masks, (S).vtab, (S).cnt, (S).blk, wr_bid); \
    return;                                                                                                                       \
  }                                                                                                                               \
  wr_bid -= (n_setup_blocks);
// (the setup stage needs ~114 VGPRs: the fused rect variant asks for 4 waves per SIMD, not the 8 of the plain one)
#ifdef WRHIP_HOSTSIM
#define WR_FUSED_BOUNDS(R, FEAT) __launch_bounds__(1024 / R)
#else
#ifndef WR_FUSED_RECT_WAVES
#define WR_FUSED_RECT_WAVES 4
#endif
#define WR_FUSED_BOUNDS(R, FEAT) __launch_bounds__(1024 / R, (FEAT) == 0 ? WR_FUSED_RECT_WAVES : WR_TEX_WAVES)   /* (depth-tested rect variant: 128 VGPRs as well) */
#endif
template <int FMT, bool DEPTH, int R, int FEAT>
__global__ void WR_FUSED_BOUNDS(R, FEAT)
wr_setup_raster_kernel(WrSetupArgs S, int n_setup_blocks,
                       const WrTargetDesc* __restrict__ targets, int n_targets,
                       const WrDrawDesc* __restrict__ draws, const WrPrim* __restrict__ prims,
                       const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, const float* __restrict__ vtab,
                       unsigned long long* __restrict__ masks, int bin_offset) {
  WR_FUSED_PROLOGUE(S, n_setup_blocks)
  wr_raster_body<FMT, DEPTH, R, FEAT>(targets, n_targets, draws, prims, recs, aux, vtab, masks, wr_bid + bin_offset);
}
// ... and for a THIN colour launch (<= 256 bins, four 256-thread workgroups of 64 x 4 pixel strips per bin: see wr_raster_kernel's R == 1
// entry) -- wrench transforms-simple: 35 us of setup stage in line with 79 us of raster every frame, because the small launch was worth
// more thin than as the carrier of the setup stage in its 64 x 16 shape (241 us).  Here it is both.
template <int FMT, bool DEPTH, int R, int FEAT>
__global__ void __launch_bounds__(256, WR_TEX_WAVES)
wr_setup_raster_thin_kernel(WrSetupArgs S, int n_setup_blocks,
                            const WrTargetDesc* __restrict__ targets, int n_targets,
                            const WrDrawDesc* __restrict__ draws, const WrPrim* __restrict__ prims,
                            const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, const float* __restrict__ vtab,
                            unsigned long long* __restrict__ masks, int bin_offset) {
  static_assert(R == 1 && FMT == WR_FMT_RGBA8 && !DEPTH, "the thin colour shape");
  WR_FUSED_PROLOGUE(S, n_setup_blocks)
  const int b = wr_bid;
  wr_raster_body<FMT, DEPTH, R, FEAT>(targets, n_targets, draws, prims, recs, aux, vtab, masks, b / 4 + bin_offset, b % 4, 4);
}
// Text launches: the glyph walk is latency-bound (cfg3: per-lane record + atlas fetches), a fourth wave per SIMD pays for the handful
// of values a 128-VGPR build spills (tile pass 139 -> 128 us, 5.45 k -> 5.7-5.9 k frames/s; profiles/r03_e_ring_w4_ab.txt).  The
// same 128-VGPR build costs the OTHER users of this variant -- masked solids (cfg4's tile pass 22.5 -> 29 us), perspective images
// (+10..20 %) -- so it is a second instantiation of the same body that the host picks for levels whose R8-texture prims are
// glyph runs (Context::Held::dense), not a change of the variant's bounds.
// Round 4: with the glyph records (WrGlyphRec) the walk issues a fraction of the loads it used to and the 168-VGPR build, free of
// spills, is the faster one again (97.6 vs 102.3 us, profiles/r04_g_dense_waves_ab.txt): the dense instantiation is now opt-in
// (WRHIP_DENSE_TEXT=1) and stays built for that A/B.
#ifndef WR_DENSE_WAVES
#define WR_DENSE_WAVES 4
#endif
template <int FMT, bool DEPTH, int R, int FEAT>
__global__ void __launch_bounds__(1024 / R, WR_DENSE_WAVES)
wr_raster_dense_kernel(const WrTargetDesc* __restrict__ targets, int n_targets,
                       const WrDrawDesc* __restrict__ draws, const WrPrim* __restrict__ prims,
                       const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, const float* __restrict__ vtab,
                       unsigned long long* __restrict__ masks, int bin_offset) {
  wr_raster_body<FMT, DEPTH, R, FEAT>(targets, n_targets, draws, prims, recs, aux, vtab, masks, (int)blockIdx.x + bin_offset);
}
template <int FMT, bool DEPTH, int R, int FEAT>
__global__ void __launch_bounds__(1024 / R, WR_DENSE_WAVES)
wr_setup_raster_dense_kernel(WrSetupArgs S, int n_setup_blocks,
                             const WrTargetDesc* __restrict__ targets, int n_targets,
                             const WrDrawDesc* __restrict__ draws, const WrPrim* __restrict__ prims,
                             const WrRec* __restrict__ recs, const WrAux* __restrict__ aux, const float* __restrict__ vtab,
                             unsigned long long* __restrict__ masks, int bin_offset) {
  WR_FUSED_PROLOGUE(S, n_setup_blocks)
  wr_raster_body<FMT, DEPTH, R, FEAT>(targets, n_targets, draws, prims, recs, aux, vtab, masks, wr_bid + bin_offset);
}
// The same fusion for a flush whose longest held-back launch is a mask-rows launch (cfg4: the tile passes are 11-17 us, the
// setup stage of the next frame 30-50 us of dependent latency, the rows launch 50-100 us).
#ifndef WR_INST_ONLY      /* (not a template: defined by wrhip.hip alone, not by the instantiation units wrhip_inst.hip) */
__global__ void __launch_bounds__(256, 4)
wr_setup_rows_kernel(WrSetupArgs S, int n_setup_blocks, const WrTargetDesc* __restrict__ targets, int bin_lo, int bin_hi,
                     const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux, unsigned long long* __restrict__ ctl,
                     const WrMaskSlot* __restrict__ slots, uint8_t* __restrict__ store) {
  WR_FUSED_PROLOGUE(S, n_setup_blocks)
  wr_mask_rows_body(targets, bin_lo, bin_hi, prims, aux, ctl, slots, store, wr_bid, (int)gridDim.x - n_setup_blocks - S.up_blocks);
}
// ... and in front of a tile-rows launch (wr_tile_rows_kernel): a frame whose tiles all went to the row kernel has no bin launch to carry it
__global__ void __launch_bounds__(256, 4)
wr_setup_tile_rows_kernel(WrSetupArgs S, int n_setup_blocks, const WrTargetDesc* __restrict__ targets, int t0, int nt, const WrDrawDesc* __restrict__ draws,
                          const WrPrim* __restrict__ prims, const WrAux* __restrict__ aux) {
  WR_FUSED_PROLOGUE(S, n_setup_blocks)
  wr_tile_rows_body(targets, t0, nt, draws, prims, aux, wr_bid, (int)gridDim.x - n_setup_blocks - S.up_blocks);
}
#endif
