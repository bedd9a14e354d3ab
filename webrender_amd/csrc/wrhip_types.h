// wrhip_types.h -- structures shared between the host-side GL state tracker
// (wrhip_gl.cpp part of wrhip.hip) and the gfx950 kernels (wrhip_kernels.h).
//
// Data flow of one flush:
//   host: DrawDesc[] (one per DrawElementsInstanced / clear), TargetDesc[],
//         instance bytes  --(one H2D copy of the frame arena)-->  HBM
//   wr_vertex_kernel : instance -> Prim (screen rect, depth, packed colour,
//                      sampling setup); restates the shader's vertex stage
//   wr_bin_kernel    : Prim -> per-bin ordered bitmasks (bit i of word w of bin
//                      b = prim 64*w+i touches bin b); order = submission order
//   wr_raster_kernel : one workgroup per 64x64 bin; pixels stay in registers
//                      while every prim of the bin is applied in order; each
//                      destination pixel is read at most once and written once
#pragma once
#include <stdint.h>

#define WR_BIN_W 64
#define WR_BIN_H 64
#define WR_MAX_TEX 12
#define WR_MAX_ATTRIBS 16  // instance attributes of one program (CLIP_RECT has 15: vertex.rs:359-445)  // sampler slots, renderer/mod.rs:371-385

// Texture slots (TextureSampler order, renderer/mod.rs:371-385)
enum WrSlot {
  WR_S_COLOR0 = 0, WR_S_COLOR1, WR_S_COLOR2, WR_S_GPU_CACHE, WR_S_TRANSFORMS,
  WR_S_RENDER_TASKS, WR_S_DITHER, WR_S_PRIM_HEADERS_F, WR_S_PRIM_HEADERS_I,
  WR_S_CLIP_MASK, WR_S_GPU_BUFFER_F, WR_S_GPU_BUFFER_I
};

enum WrTexFormat {  // texture.h TextureFormat
  WR_FMT_NONE = 0, WR_FMT_RGBA32F, WR_FMT_RGBA32I, WR_FMT_RGBA8, WR_FMT_R8,
  WR_FMT_RG8, WR_FMT_R16, WR_FMT_RG16, WR_FMT_DEPTH24
};

// Shader programs known to the backend.  Keys are the "name FEATURES" strings
// WebRender sends through ShaderSourceByName (webrender_build/src/
// shader_features.rs:64-248).
enum WrShader {
  WR_SH_NONE = 0,
  WR_SH_PS_QUAD_TEXTURED,
  WR_SH_BRUSH_SOLID,
  WR_SH_BRUSH_SOLID_ALPHA,
  WR_SH_COMPOSITE,
  WR_SH_COMPOSITE_FAST,
  WR_SH_PS_CLEAR,
  WR_SH_PS_TEXT_RUN,
  WR_SH_PS_TEXT_RUN_DUAL,   // ALPHA_PASS,DUAL_SOURCE_BLENDING
  WR_SH_CS_BLUR_ALPHA,
  WR_SH_CS_BLUR_COLOR,
  WR_SH_CS_SCALE,
  WR_SH_CS_CLIP_RECT,
  WR_SH_CS_CLIP_RECT_FAST,
  WR_SH_CS_CLIP_BOX_SHADOW,
  WR_SH_BRUSH_IMAGE,
  WR_SH_BRUSH_IMAGE_ALPHA,
  WR_SH_BRUSH_OPACITY,             // brush_opacity [ANTIALIASING]
  WR_SH_BRUSH_OPACITY_ALPHA,       // brush_opacity ALPHA_PASS[,ANTIALIASING]
  WR_SH_BRUSH_IMAGE_REPEAT,        // brush_image ANTIALIASING,REPETITION,TEXTURE_2D
  WR_SH_BRUSH_IMAGE_REPEAT_ALPHA,  // brush_image ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D
  WR_SH_BRUSH_LINEAR_GRADIENT,
  WR_SH_BRUSH_LINEAR_GRADIENT_ALPHA,
  WR_SH_BRUSH_BLEND,
  WR_SH_BRUSH_BLEND_ALPHA,
  WR_SH_PS_QUAD_MASK,
  WR_SH_PS_QUAD_MASK_FAST,
  WR_SH_CS_BORDER_SOLID,
  WR_SH_CS_BORDER_SEGMENT,
  WR_SH_CS_FAST_LINEAR_GRADIENT,
  WR_SH_CS_LINE_DECORATION,
  WR_SH_CS_LINEAR_GRADIENT,
  WR_SH_CS_RADIAL_GRADIENT,
  WR_SH_CS_CONIC_GRADIENT,
  WR_SH_PS_QUAD_RADIAL_GRADIENT,
  WR_SH_PS_QUAD_CONIC_GRADIENT,
  WR_SH_BRUSH_IMAGE_DUAL,          // brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D
  WR_SH_BRUSH_MIX_BLEND,           // brush_mix_blend (batch.rs:1931-2003)
  WR_SH_BRUSH_MIX_BLEND_ALPHA,
  WR_SH_PS_COPY,                   // texture-cache copies, batched uploads (renderer/mod.rs:1808-1846, upload.rs:540-620)
  WR_SH_PS_TEXT_RUN_GT,            // ps_text_run ALPHA_PASS,GLYPH_TRANSFORM,TEXTURE_2D (glyphs rasterised under the run's 2-D transform)
  WR_SH_PS_TEXT_RUN_DUAL_GT,       // ... with DUAL_SOURCE_BLENDING
  WR_SH_PS_SPLIT_COMPOSITE,        // the split polygons of a preserve-3d context (batch.rs:1985-2080)
  WR_SH_BRUSH_YUV,                 // brush_yuv_image TEXTURE_2D,YUV (video frames as planar / semi-planar YUV: batch.rs:2301-2390)
  WR_SH_BRUSH_YUV_ALPHA,           // ... ALPHA_PASS
  WR_SH_COMPOSITE_YUV,             // composite TEXTURE_2D,YUV (video surfaces composited straight into the window: composite.rs ExternalSurfaceDependency::Yuv)
  WR_SH_CS_SVG_FILTER,             // cs_svg_filter: one node of a CSS / SVG filter chain (render_target.rs:901-990, renderer/mod.rs:2527-2554)
  WR_SH_BRUSH_IMAGE_REPEAT_DUAL,   // brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_2D (shader_features.rs:166-170)
  WR_SH_CS_SVG_FILTER_NODE,        // cs_svg_filter_node: one node of an SVG filter graph (render_target.rs:1000-1170, renderer/mod.rs:2556-2583)
  WR_SH_CLEAR_OP,  // internal: glClear recorded as an ordered draw
  WR_SH_COUNT
};

// Blend keys actually reachable from WebRender's BlendModes
// (device/gl.rs:3901-4025 -> gl.cc:614-660 hash_blend_key).
enum WrBlend {
  WR_BLEND_NONE = 0,               // GL_ONE, GL_ZERO / blend disabled
  WR_BLEND_ALPHA,                  // SRC_ALPHA, 1-SRC_ALPHA, ONE, 1-SRC_ALPHA
  WR_BLEND_PREMULT,                // ONE, 1-SRC_ALPHA
  WR_BLEND_ZERO_INV_SRC_COLOR,     // ZERO, 1-SRC_COLOR
  WR_BLEND_ZERO_INV_SRC_COLOR_A1,  // ZERO, 1-SRC_COLOR, ZERO, ONE
  WR_BLEND_DEST_OUT,               // ZERO, 1-SRC_ALPHA
  WR_BLEND_MULTIPLY,               // ZERO, SRC_COLOR
  WR_BLEND_ADD,                    // ONE, ONE
  WR_BLEND_ADD_A_OVER,             // ONE, ONE, ONE, 1-SRC_ALPHA
  WR_BLEND_INV_DST_A,              // 1-DST_ALPHA, ONE, ZERO, ONE
  WR_BLEND_CONST_COLOR,            // CONSTANT_COLOR, 1-SRC_COLOR
  WR_BLEND_DUAL_SRC,               // ONE, 1-SRC1_COLOR
  WR_BLEND_MIN, WR_BLEND_MAX,
  WR_BLEND_SCREEN,                 // ONE, 1-SRC_COLOR
  WR_BLEND_DROP_SHADOW, WR_BLEND_SUBPIXEL_TEXT,
  WR_BLEND_UNSUPPORTED,
  // KHR_blend_equation_advanced (BlendMode::Advanced: mix-blend-mode pictures drawn through brush_image on a backend
  // that advertises the extension, device/gl.rs:3980-4025; blend.h:565-677), in GL enum order
  WR_BLEND_MULTIPLY_KHR, WR_BLEND_SCREEN_KHR, WR_BLEND_OVERLAY_KHR, WR_BLEND_DARKEN_KHR, WR_BLEND_LIGHTEN_KHR,
  WR_BLEND_COLORDODGE_KHR, WR_BLEND_COLORBURN_KHR, WR_BLEND_HARDLIGHT_KHR, WR_BLEND_SOFTLIGHT_KHR,
  WR_BLEND_DIFFERENCE_KHR, WR_BLEND_EXCLUSION_KHR,
  WR_BLEND_HSL_HUE_KHR, WR_BLEND_HSL_SATURATION_KHR, WR_BLEND_HSL_COLOR_KHR, WR_BLEND_HSL_LUMINOSITY_KHR,
};

// Prim families a raster launch has to handle; the kernel is specialised on the
// set so that rect-only passes carry neither the registers nor the code of the
// texture / blur / clip paths.  The host derives it from the shaders and blend
// states of the launch's draws (conservatively: a draw not marked WR_DF_SIMPLE
// gets WR_FEAT_GENERIC, and a SIMPLE draw that yields anything but a solid is
// counted as unsupported instead of being drawn wrongly).
enum WrFeat {
  WR_FEAT_TEX = 1,       // WR_PK_TEX_RGBA8 row fast path (composite, images)
  WR_FEAT_R8TEX = 2,     // glyph blits / masked solids (WrTexRec)
  WR_FEAT_GENERIC = 4,   // out-of-line per-pixel path: any blend key, linear filters, fragment-shader tails
  WR_FEAT_BLUR = 8,      // cs_blur
  WR_FEAT_CLIP = 16,     // cs_clip_rectangle / cs_clip_box_shadow (R8 targets)
  WR_FEAT_SHADE = 32,    // shader replays on RGBA8 targets: gradients (per row), brush_blend filters (per pixel)
};

struct WrTexDesc {
  const void* ptr;   // HBM address (nullptr -> swgl's 1x1 transparent null sampler)
  int32_t width, height;
  int32_t stride;    // in elements: bytes/4 for bpp>=4, bytes/2 for bpp 2, bytes for bpp 1 (gl.cc:883-899)
  int16_t format;    // WrTexFormat
  int16_t linear;    // TextureFilter::LINEAR (and width >= 2, gl.cc:874-880)
  float sw, sh;      // samplerScale (texture.h:433-443): what a uv is multiplied by on its way to texels -- width / height for a
                     // sampler2D, 1 / 1 for the sampler2DRect of a TEXTURE_RECT key (unnormalised uv); also TEX_SIZE in the vertex stages
};

enum WrDrawFlags {
  WR_DF_DEPTH_TEST = 1,    // depth test on AND depth attachment present AND cleared (rasterize.h:962)
  WR_DF_DEPTH_WRITE = 2,
  WR_DF_DEPTH_LESS = 4,    // GL_LESS instead of GL_LEQUAL
  WR_DF_CLEAR_COLOR = 8,   // WR_SH_CLEAR_OP
  WR_DF_CLEAR_DEPTH = 16,
  WR_DF_QUADS = 64,        // host: this draw may hold textured prims on general (rotated) quads or with swgl_antiAlias;
                           // its launch carries WR_FEAT_SHADE, where WR_PK_TEX_QUAD lives
  WR_DF_MASK_ROWS = 128,   // host: the cs_clip_* prims of this draw may be pre-evaluated row by row (wr_mask_rows_kernel) into the
                           // flush's mask-row store; the raster stage then only blends the stored bytes (WR_PK_MASK_ROWS)
  WR_DF_SIMPLE = 32,       // host promise: every prim of this draw is a solid with blend NONE/PREMULT (see WrFeat)
  WR_DF_TEX_RECT = 512,    // the program is a TEXTURE_RECT key: sColor0-2 are sampler2DRect (WrTexDesc::sw / sh = 1)
  WR_DF_XFORM = 256,       // host: the transform ids this draw's prims can reference include non-axis-aligned ones (rotations, perspective)
  WR_DF_GTAB = 1024,       // host: the gradient tables of this draw's prims are copied into the flush's pool by the setup stage
                           // (WrDrawDesc::gtab_base) -- the raster stage then never reads sGpuBufferF for them
};
#define WR_GTAB_WORDS 1040        /* 130 entries x (start, step) x 4 floats */

struct WrDrawDesc {
  int32_t shader;        // WrShader
  int32_t target;        // index into WrTargetDesc[]
  int32_t first_prim;    // global prim index of instance 0
  int32_t count;         // instances
  int32_t blend;         // WrBlend (WR_BLEND_NONE when GL_BLEND disabled)
  int32_t flags;         // WrDrawFlags
  int32_t clip[4];       // apply_scissor(colortex) in target pixels: x0,y0,x1,y1 (gl.cc:857-864)
  float vp_origin[2];    // viewport.origin - colortex.offset   (rasterize.h:1571-1573)
  float vp_size[2];
  float transform[16];   // uTransform, column-major
  float quad[8];         // aPosition of the 4 SIMD lanes in swgl lane order 0,1,3,2 (gl.cc:1031-1039)
  uint32_t clear_color;  // WR_SH_CLEAR_OP: BGRA8 / R8 value
  uint32_t clear_depth;  // WR_SH_CLEAR_OP: 24-bit depth
  uint32_t blend_color[2];  // ctx->blendcolor as 4 x u16 (b,g,r,a)
  uint64_t inst_offset;  // byte offset of this draw's instance data in the arena
  int32_t inst_stride;
  int32_t attr_off[WR_MAX_ATTRIBS];   // byte offset of the shader's k-th instance attribute, -1 = unbound (zeros)
  int32_t attr_bytes[WR_MAX_ATTRIBS];
  int32_t vtab_base;     // first entry of this draw's per-row v table (-1: none), vtab_rows entries per instance
  int32_t vtab_rows;
  int32_t query_slot;    // GL_SAMPLES_PASSED query active around this draw: slot of WrUnsupportedCounters::samples (-1: none)
  int32_t gtab_base;     // WR_DF_GTAB: first word of this draw's gradient-table copies in the pool (WrTargetDesc::qtab), WR_GTAB_WORDS per instance; -1: none
  uint32_t attr_u16;     // bit k: attribute k is made of 16-bit unsigned integers (VertexAttributeKind::U16) // bytes provided by the VAO for that attribute (VertexAttrib::size)
  WrTexDesc tex[WR_MAX_TEX];
};

struct WrTargetDesc {
  void* color;          // HBM
  uint32_t* depth;      // HBM flat depth (u32 per pixel) or nullptr
  int32_t width, height;
  int32_t stride;       // bytes
  int32_t format;       // WR_FMT_RGBA8 | WR_FMT_R8
  int32_t load_color;   // 1: bins start from HBM content, 0: from init_color
  uint32_t init_color;
  int32_t load_depth;   // 1: depth bins start from `depth`, 0: from init_depth
  uint32_t init_depth;
  int32_t store_depth;  // 1: write depth back to `depth`
  int32_t bins_x, bins_y;
  int32_t first_bin;    // global bin index of bin (0,0)
  int32_t first_prim, end_prim;  // global prim range of this target
  int32_t word_base;    // first u64 word of this target's bin masks
  int32_t words_per_bin;
  int32_t y_begin, y_end;        // pixel rows owned by this process (multi-GPU strip sharding)
  // Write-through of an opaque 1:1 composite ("forwarded composites"): when the only thing a later target
  // of the same flush does with this target is copy it unblended, texel for pixel, the raster stage stores every finished pixel
  // row a second time at its place in that target -- pixel (x, y) goes to (x + fwd_dx, fwd_y0 + fwd_ys * y) if that lies inside
  // fwd_clip -- and the composite draw (its 4 B/pixel read and its launch) is dropped.
  void* fwd_color;               // nullptr: none
  int32_t fwd_stride;            // bytes
  int32_t fwd_dx, fwd_y0, fwd_ys;
  int32_t fwd_clip[4];           // x0, y0, x1, y1 in the destination target
  int32_t dw_first, dw_end;      // global prim range spanned by this target's depth-writing draws (dw_end <= dw_first: none).
                                 // A depth-tested prim that consumes interpolants looks there for what hides parts of its rows
                                 // (draw_depth_span's sub-spans, rasterize.h:612-664)
  // Mask-row store of the flush (R8 targets holding cs_clip_* draws, see WrMaskSlot); nullptr: none
  unsigned long long* mr_ctl;    // allocation word: slots << 48 | rows << 28 | bytes / 16
  struct WrMaskSlot* mr_slots;
  uint8_t* mr_store;
  uint32_t mr_cap16;             // capacity of mr_store in 16-byte units (< 2^28)
  uint32_t mr_max_slots;         // capacity of mr_slots (< 2^16)
  int32_t cells;                 // 1: rect-only bins that start from a clear may take the cell raster (wr_raster_cells)
  int32_t rows_mode;             // 1: a SPAN-ROWS target -- every draw is a row-evaluable off-screen pass (cs_blur, cs_scale, scissored clears)
                                 // without blending hazards: its prims are not binned and it has no bins; wr_span_rows_kernel gives every
                                 // 256-pixel piece of every target row one wave that applies the target's prims in order
  // Flattened depth rows.  A perspective span flattens the depth row it touches (rasterize.h:1222-1232), and swgl draws every
  // LATER depth-tested prim on that row chunk by chunk from the span start through main() (:1021-1031) instead of handing the
  // span shader one depth run at a time.  flat_rows[y] = index of the first depth-tested perspective prim with a non-empty
  // span on target row y (0xFFFFFFFF: none), written by the setup stage; flat_rows[height] = the smallest of them (the raster
  // stage's early-out).  nullptr: the target cannot hold such a prim (the host saw no general-quad draw with the depth test on).
  uint32_t* flat_rows;
  struct WrUnsupportedCounters* counters;      // where the raster stage reports what it could not draw exactly
  const struct WrGlyphRec* grecs;              // the flush's glyph records, one per prim (global prim index), see WrGlyphRec
  unsigned* bin_ctr;                           // one arrival counter per bin of this target (zero between launches): thin R8 launches that give
                                               // a bin several workgroups count themselves in, the last one re-zeroes the bin's mask words
  // Row tables of general quads (WrQuadRec::rowtab): a pool of floats the setup stage hands out with one atomicAdd per prim
  // (`qtab_ctl`: a word of the flush's arena, zero when the arena arrives) -- the same pool for every target of a flush;
  // nullptr: none (no draw of the flush can hold a rotated / projected prim), and a prim that does not fit keeps its per-row sums.
  float* qtab;
  unsigned long long* qtab_ctl;
  uint32_t qtab_cap;                           // floats
  uint32_t qtab_pad;                           // bit 0: no row tables (WRHIP_NO_QTAB: the pool then only takes what the depth runs spill); bit 1: rows of a strip do not share their runs (WRHIP_NO_RUN_SHARE: A/B)
};

// Pre-evaluated clip-mask prims.  A cs_clip_rectangle / cs_clip_box_shadow prim covers its rows with long solid runs and a few
// short evaluated ones; the span shaders' state machines are per row, not per pixel.  Inside the bin raster every wave of
// every bin the prim touches would replay a row's state machine up to its own 64 pixels.  Instead the setup stage reserves
// (y1 - y0) rows of pitch bytes in the flush's mask-row store for such a prim, wr_mask_rows_kernel evaluates each (prim, row)
// ONCE with one wave -- the walk along the row is wave-uniform, the lanes share out the pixels of every run -- and the raster
// stage blends the stored bytes like a 1:1 texture (WR_PK_MASK_ROWS: WrRec::c0/c1 = address of the prim's reservation, WrRec::z =
// pitch).  A reservation is [row map: one u32 byte offset per row][rows x pitch bytes, first column x0 & ~3]: rows that come
// out identical to another row of the prim (the middle band of a nine-patch) point at that row's bytes and are not evaluated.
struct WrBoxKey {                // what decides the bytes of a cs_clip_box_shadow row besides its x interpolants (wr_box_row_key)
  float o0, s0, o2, s2;
  int32_t ss_se, os01, os23;
  float mv_mul, mv_div, in_mul, in_div;
  int32_t valid;
};
// Row-sum tables of a mask prim ("row-sum tables").  swgl steps a prim's edge interpolants once per row
// (Edge::nextRow: s += step), so row k holds the k-fold sequential fp32 sum.  While that sum stays inside one binade every add moves
// it by the same whole number q of ulps, so s_k is piecewise linear in k with one piece per binade the sum passes through: the setup
// stage walks a prim's rows ONCE per interpolant and leaves the pieces here -- rows [k[j], k[j + 1]) hold significand(s[j]) + (k - k[j]) * q[j]
// at s[j]'s sign and exponent (q 0: the value s[j] itself) --, and the rows kernel reads a row's sums off them instead of walking
// the binades again for every row (20-33 k cycles per row, profiles/r06_b_rows_times.txt).
#define WR_ACCTAB_N 32
struct WrAccTab { int32_t k[WR_ACCTAB_N]; uint32_t s[WR_ACCTAB_N]; int32_t q[WR_ACCTAB_N]; };
struct WrAccTabs {
  // (the header is what a row reads first -- start, step and kind of each sum, one lane per sum -- together with its table's k[]: one round
  // trip, then the piece's s / q: a second one)
  float s0[8], st[8];            // cs_clip_box_shadow: uv L0/L1/R0/R1, local pos L0/L1/R0/R1; cs_clip_rectangle: uv L0/L1/R0/R1
  int32_t n[8];                  // pieces of tab[i] (mode 2)
  int32_t mode[8];               // 0: s0 + k * st in one fused multiply-add is the sequential sum at every row (wr_accum's closed form holds at the
                                 // last row, or the step is 0); 1: the prim's uv sums were verified linear (WrPrim::rows_linear: the fp64 form);
                                 // 2: tab[i]; 3: the walk (the pieces did not fit); 4: the sum equals sum i - 2 (left / right edges of an axis-aligned prim)
  WrAccTab tab[8];
};
#define WR_ACCTABS_N16 ((sizeof(WrAccTabs) + 15) / 16)
struct WrMaskSlot {
  int32_t prim;                  // global prim index
  int32_t target;                // its target (the rows kernel of a raster launch only evaluates that launch's targets)
  uint32_t row0;                 // first work row of this prim in the flush-wide row numbering
  uint32_t pitch;                // bytes per stored row (multiple of 4)
  uint32_t off16;                // first stored row, in 16-byte units into mr_store
  uint32_t pad[3];               // pad[0]: waves sharing a row (1: see the setup stage); pad[1]: the prim's WrAccTabs, in 16-byte units from its first byte in the store (0: none)
  WrBoxKey key;                  // cs_clip_box_shadow: key of the prim's middle row (rows equal to it are not evaluated)
};
// span rows (WrTargetDesc::rows_mode): pixels per lane and wave-sized pieces per row of a target of the given width (host and kernel agree)
#define WR_SPAN_PPL(width) ((width) > 512 ? 4 : 1)
#define WR_SPAN_PIECES(width) (((width) + 64 * WR_SPAN_PPL(width) - 1) / (64 * WR_SPAN_PPL(width)))
#define WR_MR_MAX_SLOTS 65535u
#define WR_MR_MAX_ROWS 1048575u
#define WR_MR_MAX_CAP16 268435455u

enum WrPrimKind {
  WR_PK_NONE = 0,       // culled / nothing to draw
  WR_PK_CLEAR,
  WR_PK_SOLID,          // swgl_commitSolid* / flat fragment colour
  WR_PK_TEX_RGBA8,      // swgl_commitTexture*RGBA8 family (rect, axis-aligned uv)
  WR_PK_UNSUPPORTED,
  WR_PK_SOLID_FOLDED,   // WrRec only: solid prim pre-folded for the raster hot path (wr_make_rec)
  WR_PK_SOLID_AA,       // flat colour with swgl_antiAlias edge coverage (axis-aligned quads; WrAARec)
  WR_PK_SOLID_MASKED,   // commit_masked_solid_span: flat colour x R8 clip mask sampled 1:1 (swgl_clipMask)
  WR_PK_TEX_FS,         // textured quad with no usable span shader: every pixel runs the fragment shader's main()
  WR_PK_BOX_SHADOW,     // cs_clip_box_shadow's nine-patch span shader (WrBoxRec)
  WR_PK_CLIP_RECT,      // cs_clip_rectangle's rounded-rect span rasteriser (WrClipRec)
  WR_PK_BLUR,           // swgl_commitGaussianBlur{R8,RGBA8}: one separable pass (WrBlurRec)
  WR_PK_TEX_R8,         // swgl_commitTextureLinearColorR8ToRGBA8: R8 mask expanded to RGBA8, colour-modulated
  WR_PK_GRADIENT,       // swgl_commitLinearGradientRGBA8 (WrGradRec); v_pos travels in WrPrim's uv interpolants
  WR_PK_FILTER,         // brush_blend: fragment shader only (texture() + CalculateFilter, WrFilterRec); uv as WR_PK_TEX_FS
  WR_PK_SOLID_QUAD,     // solid colour on a general (rotated / skewed) convex quad: per-row spans from WrQuadRec, optional AA
  WR_PK_TEX_REPEAT,     // swgl_commitTextureRepeat[Color]RGBA8 (brush_image REPETITION): per-row replay of the repeat walk (WrRepeatRec)
  WR_PK_TEX_QUAD,       // a textured prim (WrQuadRec::base_kind) on a general convex quad and / or with swgl_antiAlias: per-row spans and
                        // edge interpolants from WrQuadRec, then the base kind's span / main() evaluation
  WR_PK_MIX_BLEND,      // brush_mix_blend: fragment shader only, two textures; the backdrop uv travels in WrPrim's uv interpolants, the
                        // source uv in WrMixRec
  WR_PK_QUAD_MASK,      // ps_quad_mask: fragment shader only (rounded-rect coverage, WrClipRec); vClipLocalPos.xy travels in the uv interpolants
  WR_PK_BORDER_SOLID,   // cs_border_solid: fragment shader only (corner clips, edge-colour mix; WrBorderRec); vPos travels in the uv interpolants
  WR_PK_BORDER_SEGMENT, // cs_border_segment: fragment shader only (styles double / groove / ridge, dot / dash clips; WrBorderSegRec)
  WR_PK_FAST_GRADIENT,  // cs_fast_linear_gradient: main() only, mix(vColor0, vColor1, vPos) (WrFastGradRec); vPos travels in the u interpolant
  WR_PK_LINE_DECORATION,// cs_line_decoration: main() only (solid / dashed / dotted / wavy; WrLineRec); vLocalPos in the uv interpolants
  WR_PK_YUV,            // brush_yuv_image: swgl_commitTextureLinearYUV (up to three planes, fixed-point colour matrix; WrYuvRec); the Y plane's
                        // uv travels in WrPrim's uv interpolants, the chroma planes' in WrYuvRec
  WR_PK_SVG_FILTER,     // cs_svg_filter / cs_svg_filter_node: main() only, up to two inputs; vInput1Uv travels in WrPrim's uv interpolants,
                        // vInput2Uv and the flat varyings in WrSvgRec
  WR_PK_MASK_ROWS,      // WrRec only: a WR_PK_BOX_SHADOW / WR_PK_CLIP_RECT prim whose rows wr_mask_rows_kernel has evaluated (WrMaskSlot)
};

enum WrPrimFlags {
  WR_PF_DEPTH_TEST = 1, WR_PF_DEPTH_WRITE = 2, WR_PF_DEPTH_LESS = 4,
  WR_PF_CLEAR_COLOR = 8, WR_PF_CLEAR_DEPTH = 16,
  WR_PF_HAS_COLOR = 32,   // textured: modulate by colour (applyColor)
  WR_PF_TAIL_CLAMP = 64,     // main(): clamp uv to uv_bounds before sampling
  WR_PF_TAIL_MODULATE = 128, // main(): multiply texel by fcolor
  WR_PF_MASKED = 8,          // non-solid prim under swgl_clipMask: src = muldiv255(src, mask) ahead of the blend
                             // (blend.h:458-460; shares its bit with WR_PF_CLEAR_COLOR -- clears are never masked)
};

// Output of the vertex stage: everything the raster stage needs for one quad.
struct WrPrim {
  int32_t x0, y0, x1, y1;   // covered pixel rect [x0,x1) x [y0,y1) in target pixels (already clipped)
  uint32_t z;               // uint32(0xFFFFFF * screenZ)  (rasterize.h:1593)
  int16_t kind;             // WrPrimKind
  int16_t blend;            // WrBlend
  int32_t flags;            // WrPrimFlags
  int32_t draw;             // index of the WrDrawDesc (textures, clip)
  uint32_t color[2];        // packed WideRGBA8 (b,g | r,a as u16 pairs)
  // textured prims: edge interpolants (Edge, rasterize.h:850-886) of the
  // screen-left and screen-right edges at the first covered row, per-row
  // slopes, and the unclipped edge x positions
  float uvL0[2], uvLs[2], uvR0[2], uvRs[2];
  float xl, xr;
  float uv_bounds[4];       // uv_rect passed to swgl_commitTexture*
  float fcolor[4];          // float colour for the fragment-shader (tail) path
  int32_t tex_slot;         // sampler slot
  int32_t mask_off[2];      // WR_PK_SOLID_MASKED / WR_PF_MASKED: target pixel - mask texel (swgl_ClipMaskOffset)
  int32_t rows_linear;      // 1: edge interpolants at row k equal L0 + k*slope exactly (closed form of Edge::nextRow)
  float uv_add[2];          // added to the interpolated uv of every pixel before sampling (brush_image: + v_uv_bounds.xy)
  int32_t dual;             // brush_image DUAL_SOURCE_BLENDING: 1 -- main() also writes oFragBlend = mask * dual_swz[0] + mask.aaaa * dual_swz[1]
  float dual_swz;           // ... v_mask_swizzle.x (y is -x for COLOR_MODE_MULTIPLY_DUAL_SOURCE, 0 otherwise: dual == 2)
};

// Compact per-prim record the raster stage streams (32 B, dense array): the
// full WrPrim is only touched by textured prims.
struct WrRec {
  int32_t x0, y0, x1, y1;
  uint32_t z;
  uint32_t kbf;          // kind | blend << 8 | flags << 16
  uint32_t c0, c1;       // WrPrim::color
};

// A unit glyph blit (or a masked solid: one mask texel per pixel) as the lane-by-lane glyph walk of the raster stage reads it: ONE
// 48-byte record (32 bytes for the pixels of the span, 16 more for a lane that holds tail columns) in a dense array beside recs[] (WrTargetDesc::grecs), instead of the prim's WrRec plus the head of its WrTexRec in
// the 1 KB-strided aux[] -- two cache lines per (glyph, lane).  texel under target pixel (x, y) = base[y * stride + x]; the lane
// fetches its four columns of a row with one unaligned dword load at a column clamped into [x0, max(x0, x1 - 4)].
struct WrGlyphRec {
  int16_t x0, y0, x1, y1;   // covered rect in target pixels
  uint32_t c0, c1;          // WrPrim::color
  uint64_t base;            // address of the atlas / mask texel under target pixel (0, 0)
  int32_t stride;           // bytes per atlas row
  uint32_t info;            // 0: the prim is not eligible; else 1 | depth-tested << 1 | blend << 8 | (first tail column, int16) << 16
  float fcolor[4];          // WrTexRec::fcolor: the float colour main() modulates the tail columns' texels by (read by the lanes that hold any)
};

// Per-prim sampling setup of an axis-aligned textured prim, prepared by the
// setup kernel so the raster stage needs no dependent loads (prim -> draw ->
// texture descriptor) before it can address texels.  Quantised-coordinate
// arithmetic of LINEAR_QUANTIZE_UV / blendTextureLinearFallback
// (swgl_ext.h:160-183) and of the fragment-shader tail.
struct WrTexRec {
  // (what a unit glyph blit reads comes first, in 48 contiguous bytes: the lane-by-lane glyph path of the raster stage
  // fetches it with per-lane addresses, wr_unit_glyph_lane)
  const void* ptr;          // atlas / source base address
  int32_t stride;           // elements
  int32_t span;             // pixels [0,span) of a row go through draw_span
  // `unit`: every sample of the prim is exactly one texel (both 7-bit fractions zero, no
  // clamping), texel (ix0 + n, iy0 + row) for span pixel n; tix[] = columns of the <= 3 tail pixels
  int32_t unit, ix0, iy0;
  int32_t y0;
  int32_t simple;           // 1: u constant on vertical edges, v on horizontal ones, colour is bytes
  int32_t tix[3];
  // WR_PK_TEX_RGBA8 on the nearest-fast path (blendTextureNearestFast): texel column =
  // clamp(ix0 + n, tix[0], tix[1]).  `simple` = 3: rows step one texel per target row, source row =
  // clamp(iy0 + tix[2] * (y - y0), unit & 0xFFFF, unit >> 16).  `simple` = 2: v of target row y is
  // vtab[iy0 + y - y0] (the edge interpolant accumulated row by row by the setup kernel, as
  // Edge::nextRow does); the raster stage derives the source row / texel-centre test from it.
  float fcolor[4];
  uint32_t wh;              // width | height << 16
  float ou, su;             // u at the span start, per-pixel step
  float stepx;              // quantised step per 4-pixel chunk
  float minx, maxx;         // quantised clamp (uv_rect)
  float ub0, ub2;           // uv_bounds.x / .z (tail clamp)
  float lv0, lvs;           // v at row y0, per-row slope
  float miny, maxy;
  float ub1, ub3;
};

// One separable Gaussian pass (cs_blur.glsl + swgl_ext.h:947-996, texture.h:1165-1308).
#define WR_BLUR_MAX_RADIUS 32
struct WrBlurRec {
  const void* ptr;          // source render target
  int32_t stride;           // elements
  uint32_t wh;              // width | height << 16
  int32_t format, linear;   // source WrTexFormat, filter
  int32_t hori, radius;     // vOffsetScale.x != 0, vSupport.x
  int32_t bounds[4];        // make_ivec4(vUvRect * size)
  float coeffs[2];          // vGaussCoefficients
  float uv_rect[4];         // vUvRect
  float offset_scale[2];    // vOffsetScale
  uint16_t weights[WR_BLUR_MAX_RADIUS + 2];   // uint16_t(coeff_o + 0.5), 8.8 fixed point, o = 0..radius
};

// cs_clip_rectangle flat varyings (cs_clip_rectangle.glsl:7-21)
struct WrClipRec {
  int32_t fast;             // FAST_PATH program
  float mode;               // vClipMode.x
  float w;                  // vLocalPos.w (constant: affine transforms only)
  float params[3];          // vClipParams (fast path)
  float center_radius[4][4];  // vClipCenter_Radius_{TL,TR,BR,BL}
  float plane[4][3];        // vClipPlane_{TL,TR,BR,BL}
  float bounds[4];          // vTransformBounds
};

// cs_clip_box_shadow flat varyings (cs_clip_box_shadow.glsl:7-14) + the second
// interpolated varying (vLocalPos.xy; vUv lives in WrPrim's uv interpolants)
struct WrBoxRec {
  const void* ptr;          // cached blurred shadow (R8)
  int32_t stride;
  uint32_t wh;
  int32_t format, linear;
  float mode, w;            // vClipMode.x, vLocalPos.w
  float edge[4];            // vEdge
  float uv_bounds[4];       // vUvBounds
  float uv_noclamp[4];      // vUvBounds_NoClamp
  float bounds[4];          // vTransformBounds
  float lpL0[2], lpLs[2], lpR0[2], lpRs[2];   // edge interpolants of vLocalPos.xy (as WrPrim::uv*)
};

// Anti-aliased axis-aligned quad (aa_span / aa_dist, rasterize.h:480-562; DO_AA, blend.h:433-446):
// coverage of pixel X = clamp(min(L, R), 0, 256) with
//   L = (lstart + float(laa_end + lane) * lend) + (lend / bpp) * float(bpp * (chunk_base - laa_end)), R likewise,
// lane / chunk_base relative to the span start x0.
struct WrAARec {
  float lstart, lend, rstart, rend;
  int32_t laa_end;
};

// brush_linear_gradient flat varyings (brush_linear_gradient.glsl:7-10, gradient.glsl:5-11)
struct WrGradRec {
  const float* stops;       // swgl_validateGradient: first float of the 130 x (start, step) table in sGpuBufferF,
                            // or nullptr (no span shader: every pixel runs main())
  const float* table;       // the setup stage's copy of the table as main() fetches it (entry i: texels (u, v), (u + 1, v) of address + 2 i),
                            // or nullptr (main() fetches from sGpuBufferF); where stops != nullptr it is the same memory
  int32_t address;          // v_gradient_address.x
  float repeat;             // v_gradient_repeat.x
  float scale_dir[2];       // v_scale_dir
  float start_offset;       // v_start_offset.x
  int32_t no_tile;          // cs_linear_gradient: v_pos is not wrapped to [0,1) (commitLinearGradient's tileRepeat == false)
  int32_t radial;           // 1: cs_radial_gradient: offset = length(v_pos) - start_offset (= v_start_radius.x), swgl_commitRadialGradientRGBA8
                            // 3: ps_quad_conic_gradient (main() only, approx_atan2 of v_dir: scale_dir = 0)
                            // 2: cs_conic_gradient (main() only): scale_dir = v_center, start_offset = v_start_offset,
  float conic_scale, conic_angle;   //    v_offset_scale, v_angle
  uint32_t merge[5];                // stops != nullptr: bit i = GradientStops::can_merge(entry i, entry i + 1) of the 130-entry table (i = 0 .. 128),
                                    // worked out once per prim by the setup stage: the span shaders' "how far does this merged range reach"
                                    // walks (up to 128 dependent table reads per call of the raster stage) become bit scans
};

// brush_blend flat varyings (brush_blend.glsl:17-41)
struct WrFilterRec {
  int32_t op;               // v_op
  int32_t table_address;    // v_table_address (component transfer data in sGpuCache)
  float amount;             // v_amount
  float funcs[4];           // v_funcs
  float color_mat[16];      // v_color_mat, column-major
  float color_offset[4];    // v_color_offset
};

// brush_mix_blend (brush_mix_blend.glsl:9-23): v_op, and the second varying -- v_src_uv's edge interpolants, as WrPrim::uv* hold
// the first one's (v_backdrop_uv) -- with its sample bounds
struct WrMixRec {
  int32_t op;
  float sL0[2], sLs[2], sR0[2], sRs[2];
  float s_bounds[4];
};

// cs_svg_filter (cs_svg_filter.glsl:9-27) and cs_svg_filter_node (cs_svg_filter_node.glsl:45-63): the flat varyings, and the second
// input's varying as edge interpolants (as WrMixRec holds brush_mix_blend's)
struct WrSvgRec {
  int32_t node;                     // 0: cs_svg_filter, 1: cs_svg_filter_node
  int32_t kind, input_count;        // vFilterKind, vFilterInputCount
  float sL0[2], sLs[2], sR0[2], sRs[2];     // vInput2Uv
  float rect1[4], rect2[4];         // vInput1UvRect, vInput2UvRect
  int32_t data[2];                  // vData.xy
  float fdata0[4], fdata1[4];       // vFilterData0, vFilterData1
  float float0;                     // vFloat0.x
  float color_mat[16];              // vColorMat, column-major
  int32_t funcs[4];                 // vFuncs
};

// brush_yuv_image (brush_yuv_image.glsl:9-26): the flat varyings, the chroma planes' varyings as edge interpolants (as WrPrim::uv*
// hold the luma plane's), and the coefficients of swgl's fixed-point matrix (YUVMatrix, composite.h:640-720)
struct WrYuvRec {
  int32_t format, rescale;          // vFormat.x (YUV_FORMAT_*), vRescaleFactor
  float bias[3], mat[9];            // vYcbcrBias, vRgbFromDebiasedYcbcr (column-major)
  int32_t bu, rv, gu, gv, ycoeff, ybias, uvbias, brmask;      // YUVMatrix: br_uvCoeffs, gg_uvCoeffs, yCoeffs, yBias, uvBias, br_yMask
  float uL0[2], uLs[2], uR0[2], uRs[2], u_bounds[4];          // vUv_U, vUvBounds_U
  float vL0[2], vLs[2], vR0[2], vRs[2], v_bounds[4];          // vUv_V, vUvBounds_V
};

// cs_border_solid flat varyings (cs_border_solid.glsl:11-38)
struct WrBorderRec {
  float color0[4], color1[4];       // vColor0, vColor1
  float color_line[4];              // vColorLine
  int32_t mix;                      // vMixColors.x: 0 DONT_MIX, 1 MIX_AA, 2 MIX_NO_AA
  float clip_center_sign[4];        // vClipCenter_Sign
  float clip_radii[4];              // vClipRadii
  float h_center_sign[4], v_center_sign[4];   // v{Horizontal,Vertical}ClipCenter_Sign
  float h_radii[2], v_radii[2];     // v{Horizontal,Vertical}ClipRadii
};

// cs_border_segment flat varyings (cs_border_segment.glsl:9-45)
struct WrBorderSegRec {
  float color00[4], color01[4], color10[4], color11[4];
  float color_line[4];
  int32_t segment, clip_mode, style0, style1, edge_axis[2];
  float clip_center_sign[4], clip_radii[4], edge_reference[4], partial_widths[4], cp1[4], cp2[4];
};

// cs_fast_linear_gradient / cs_line_decoration flat varyings
struct WrFastGradRec { float color0[4], color1[4]; };
struct WrLineRec { int32_t style; float params[4]; };

// General convex quad (draw_quad_spans, rasterize.h:783-1055): the scanline walk cut into the runs of
// rows that share one pair of edge instances.  An Edge is (re)initialised at row `row` with x = `x`
// and then steps x += slope once per row (Edge::nextRow), so its x on row y is the (y - row)-fold
// sequential sum -- wr_accum.  `b0`,`b1`: clipSpan of the run; masks: swgl_AAEdgeMask bits of the edges.
struct WrQuadSeg {
  int32_t row_a, row_b;             // target rows [row_a, row_b)
  float lx, ls; int32_t lrow;       // span-left edge
  float rx, rs; int32_t rrow;       // span-right edge
  float b0, b1;
  int32_t lmask, rmask;
  // WR_PK_TEX_QUAD: the edges' interpolants (uv) at row lrow / rrow and their per-row slopes (Edge ctor, rasterize.h:870-874)
  float luv[2], luvs[2], ruv[2], ruvs[2];
};

// brush_image with WR_FEATURE_REPETITION (brush_image.glsl:318-341, 380-428; swgl_ext.h:664-872)
struct WrRepeatRec {
  float tile_repeat[2];             // v_tile_repeat_bounds (0,0 in the opaque pass)
  float uv_repeat[4];               // v_uv_bounds
  int32_t alpha_pass;               // compute_repeated_uvs' ALPHA_PASS branch in main()
  int32_t no_span;                  // the span shader bails out (texture not RGBA8): every pixel runs main()
};

// draw_perspective_spans (rasterize.h:1064-1280): the edges of a run also carry screen z and 1/w (Point3D edges; values at row
// lrow / rrow of the run and per-row slopes), and the uv of WrQuadSeg are uv / w
// (a quad has at most four runs; one clipped against the view volume -- clip_side, rasterize.h:1287-1430: up to ten vertices --
// at most nine)
#define WR_MAX_QSEG 10
struct WrPerspRec { float lz[WR_MAX_QSEG], lzs[WR_MAX_QSEG], lw[WR_MAX_QSEG], lws[WR_MAX_QSEG], rz[WR_MAX_QSEG], rzs[WR_MAX_QSEG], rw[WR_MAX_QSEG], rws[WR_MAX_QSEG]; float div;   // div: WrVsOut::persp_div
  // brush_mix_blend under a projective transform: its SECOND varying (v_src_uv / w) on the runs' edges -- without perspective it rides in
  // the z / w slots above, which a projective transform needs itself (every run: a prim clipped by the near plane is a polygon of up to nine)
  float l2u[WR_MAX_QSEG], l2us[WR_MAX_QSEG], l2v[WR_MAX_QSEG], l2vs[WR_MAX_QSEG], r2u[WR_MAX_QSEG], r2us[WR_MAX_QSEG], r2v[WR_MAX_QSEG], r2vs[WR_MAX_QSEG]; };

struct WrQuadRec {
  int32_t nseg;
  int32_t aa;                       // SWGL_CLIP_FLAG_AA set for this prim
  int32_t base_kind;                // WR_PK_TEX_QUAD: WR_PK_TEX_RGBA8 / TEX_FS / TEX_R8 / TEX_REPEAT / GRADIENT / FILTER / QUAD_MASK
  int32_t pad;                      // perspective: 0 no; 1 the program has no varyings (gl_FragCoord.zw never stepped); 2 it has
  // Row table (wr_quad_build_rowtab): the edge values of every target row y in [rowtab_y0, rowtab_y0 + rowtab_rows) of this prim, written
  // ONCE by the setup stage -- rowtab[(y - rowtab_y0) * rowtab_stride + ...] = x of the left and right edge (stride 2: WR_PK_SOLID_QUAD),
  // then the two interpolants of both edges (stride 6), then 1/w and z of both edges (stride 10: perspective, or brush_mix_blend's
  // second varying) -- instead of one row-by-row sum (wr_accum) per value, lane, row and prim in the raster stage.  nullptr: no table.
  const float* rowtab;
  int32_t rowtab_y0, rowtab_rows;
  int32_t rowtab_stride, rowtab_pad;
  WrQuadSeg seg[WR_MAX_QSEG];
  union {                           // the base kind's own side record (the quad record took its place in WrAux)
    WrRepeatRec rep;                // WR_PK_TEX_REPEAT
    WrGradRec grad;                 // WR_PK_GRADIENT
    WrFilterRec filt;               // WR_PK_FILTER
    WrClipRec clip;                 // WR_PK_QUAD_MASK
    WrMixRec mix;                   // WR_PK_MIX_BLEND (rotations / skews only): op and the source's sample bounds; the second varying's edges
                                    // travel in `persp`'s z / w slots (z = u, w = v of v_src_uv: Point3D edges step exactly like interpolants)
  };
  WrPerspRec persp;                 // pad != 0: screen z and 1 / w of the runs' edges (beside the base kind's record: filters and gradients
                                    // under perspective need both)
};

// Depth runs of one target row of one prim (draw_depth_span, rasterize.h:612-664): with depth testing on, swgl hands the
// span shader one sub-span per maximal run of pixels that pass the test -- the 4-pixel chunk phase, the span / main()
// split and the filter decisions restart at every run start, and the interpolants reach run k through the chain of
// step_interp_inputs() calls of runs 0 .. k-1.  Built per (wave, prim) by the raster stage from the rects of the earlier
// depth-writing prims that can hide part of the row (wr_build_runs); n == 0: nothing to restart.  Neither the runs of a row nor the
// occluders of a strip are bounded by these constants: they size the LDS / register copies, what exceeds them lives in the pool.
#define WR_MAX_RUNS 16
#define WR_MAX_OCC 64
// `ext`: a row with more runs than the inline arrays hold keeps them as (s, e) pairs in the flush's pool (WrTargetDesc::qtab, cut by
// wr_pool_words) -- n counts them all then; readers go through wr_run_s / wr_run_e.  n == -1: a flattened depth row.
struct WrRuns { int32_t n; int32_t pad; const int32_t* ext; int32_t s[WR_MAX_RUNS], e[WR_MAX_RUNS]; };

// per-prim side record, written by the setup kernel for the kinds that need one
union WrAux {
  WrRepeatRec rep;
  WrTexRec tex;
  WrBlurRec blur;
  WrClipRec clip;
  WrBoxRec box;
  WrAARec aa;
  WrGradRec grad;
  WrFilterRec filt;
  WrMixRec mix;
  WrYuvRec yuv;
  WrSvgRec svg;
  WrQuadRec quad;
  WrBorderRec border;
  WrBorderSegRec bseg;
  WrFastGradRec fgrad;
  WrLineRec line;
};

// One queued texture upload: `rows` rows of `row_bytes` packed at `src` (HBM
// mirror of the staging ring) -> `dst` with `dst_stride`.
struct WrUploadSeg { const uint8_t* src; void* dst; uint32_t dst_stride, row_bytes, rows, pad; };

struct WrFlushParams {
  int32_t n_draws, n_targets, n_prims, n_bins;
  int32_t n_words;          // total u64 mask words
  int32_t pad[3];
};

// statistics mirrored into WrhipStats (include/wrhip.h)
#define WR_QUERY_SLOTS 64
#define WR_MAX_CHAIN 8           // levels one chained R8 launch may hold
struct WrUnsupportedCounters {
  uint32_t unsupported_prims;
  uint32_t perspective_prims;
  uint32_t dbg[6];          // diagnostics (WRHIP_DEBUG_COUNTERS)
  uint32_t chain_timeout;   // workgroups of a chained launch (wr_raster_chain_kernel) that gave up waiting at a level barrier
  uint32_t chain_arrive;    // that kernel's arrive counter (never reset: the host tracks its value)
  unsigned long long samples[WR_QUERY_SLOTS];   // GL_SAMPLES_PASSED: shaded pixels = sum of span lengths (rasterize.h:957-958, gl.cc:2784-2787)
};

// BlitFramebuffer (composite.h:167-283 scale_blit, 342-418 linear_blit): dst pixel (bx0 + i, by0 + j) of the requested
// rect, sampled from the requested source rect with the reference's integer stepping (nearest) or its 1/128-texel
// quantised uv walk (linear).
struct WrBlitArgs {
  const void* src; void* dst;
  int32_t src_stride, dst_stride;      // bytes
  int32_t sbpp, dbpp;
  int32_t sw, sh;                      // source texture size
  int32_t srx0, sry0, srw, srh;        // source request (texture pixels)
  int32_t drx0, dry0, drw, drh;        // dest request
  int32_t bx0, by0, bx1, by1;          // valid dest bounds, relative to the dest request
  int32_t invert_y, linear;
  int32_t invert_x, composite;         // Composite(): X flips (linear only), premultiplied-over blend instead of a copy (RGBA8 <- RGBA8)
};

// CompositeYUV (composite.h:1160-1386): the destination rows of linear_convert_yuv, four pixels (one chunk of linear_row_yuv) per
// thread.  Everything that is constant along a row -- the planes' x coordinates in 1/128 texel x 2^8 fixed point, their steps, where
// the half-resolution fast path (upscaleYUV42R8) starts and ends -- is worked out once on the host, as the reference does per row.
struct WrYuvBlitArgs {
  WrTexDesc y, u, v;               // planes (R8, or R16 with colour depth > 8)
  void* dst; int32_t dst_stride;   // RGBA8 destination, bytes per row
  int32_t dx0, dy0, span, rows;    // destination bounds: first pixel, pixels per row, rows
  float src_v0, src_dv;            // luma v (quantised to 1/128 texel unless a plane is narrower than 2) at the first row, step per row
  float chroma_v0, chroma_dv;
  float src_u0, chroma_u0;         // unquantised: the nearest fallback of a plane narrower than 2 texels reads ivec2(srcUV)
  int32_t yU[4], cU[4];            // cast(init_interp(u, du) * 2^8) of the first chunk
  int32_t yDU, cDU;                // per chunk
  int32_t nearest;                 // a plane is narrower than two texels: one converted texel fills the row
  int32_t color_depth;
  int32_t fast0, fast1;            // chunks [fast0, fast1) of a row take upscaleYUV42R8; cA, cB: its averaged chroma coordinates at fast0
  int32_t cA, cB;
  int32_t bu, rv, gu, gv, ycoeff, ybias, uvbias, brmask;      // YUVMatrix (as WrYuvRec)
};

