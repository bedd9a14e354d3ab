// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "ps_text_run ALPHA_PASS,TEXTURE_2D" and
// "ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D"
// (webrender_build/src/shader_features.rs:210-226). Restates
// webrender/res/ps_text_run.glsl:36-338 (non-GLYPH_TRANSFORM branch) on top of
// the prim_shared helpers in brush_base.h, with SWGL defined (SWGL_BLEND,
// SWGL_CLIP_DIST, SWGL_CLIP_MASK, SWGL_DRAW_SPAN: base.glsl:38-44).

#define WRSH_PS_TEXT_RUN(NAME, KEYSTR, DUAL_SOURCE, GLYPH_TRANSFORM)                            \
  struct NAME##_vert : wrsh::prim_vert_base<NAME##_vert> {                     \
    typedef NAME##_vert Self;                                                  \
    vec4_scalar v_color;                                                       \
    vec3_scalar v_mask_swizzle;                                                \
    vec4_scalar v_uv_bounds;                                                   \
    vec2 v_uv;                                                                 \
    /* (a program that writes gl_ClipDistance: the four distances of a vertex */ \
    /* lead its interpolants, lib.rs:541-545, 570-577) */                      \
    struct InterpOutputsPlain {                                                \
      vec2_scalar v_uv;                                                        \
    };                                                                         \
    struct InterpOutputsClip {                                                 \
      Float swgl_ClipDistance;                                                 \
      vec2_scalar v_uv;                                                        \
    };                                                                         \
    typedef typename std::conditional<GLYPH_TRANSFORM, InterpOutputsClip,      \
                                      InterpOutputsPlain>::type InterpOutputs; \
    /* ps_text_run.glsl:24-35 */                                               \
    static wrsh::RectWithEndpoint transform_rect(wrsh::RectWithEndpoint rect,  \
                                                 mat2_scalar transform) {      \
      vec2_scalar size = wrsh::rect_size(rect);                                \
      vec2_scalar center = transform * (rect.p0 + size * 0.5f);                \
      vec2_scalar radius = mat2_scalar(abs(transform[0]), abs(transform[1])) * \
                           (size * 0.5f);                                      \
      return wrsh::RectWithEndpoint{center - radius, center + radius};         \
    }                                                                          \
    static bool rect_inside_rect(wrsh::RectWithEndpoint little,                \
                                 wrsh::RectWithEndpoint big) {                 \
      return big.p0.x <= little.p0.x && big.p0.y <= little.p0.y &&             \
             little.p1.x <= big.p1.x && little.p1.y <= big.p1.y;               \
    }                                                                          \
    static vec2_scalar get_snap_bias(int subpx_dir) {                          \
      switch (subpx_dir) {                                                     \
        case 0:                                                                \
        default:                                                               \
          return vec2_scalar(0.5f);                                            \
        case 1:                                                                \
          return vec2_scalar(0.125f, 0.5f);                                    \
        case 2:                                                                \
          return vec2_scalar(0.5f, 0.125f);                                    \
        case 3:                                                                \
          return vec2_scalar(0.125f);                                          \
      }                                                                        \
    }                                                                          \
    void main() {                                                              \
      using namespace wrsh;                                                    \
      /* decode_instance_attributes, prim_shared.glsl:60-72 */                 \
      int prim_header_address = aData.x;                                       \
      int clip_address = aData.y;                                              \
      int segment_index = aData.z & 0xffff;                                    \
      int flags = aData.z >> 16;                                               \
      int resource_address = aData.w & 0xffffff;                               \
      PrimitiveHeader ph = fetch_prim_header(prim_header_address);             \
      Transform transform = fetch_transform(ph.transform_id);                  \
      ClipArea clip_area = fetch_clip_area(clip_address);                      \
      PictureTask task = fetch_picture_task(ph.picture_task_address);          \
      int glyph_index = segment_index;                                         \
      int subpx_dir = (flags >> 8) & 0xff;                                     \
      int color_mode = flags & 0xff;                                           \
      vec4_scalar text_color = fetch_from_gpu_cache_1(ph.specific_prim_address); \
      vec2_scalar text_offset = ph.local_rect.p1;                              \
      /* fetch_glyph, ps_text_run.glsl:40-53 */                                \
      int glyph_address =                                                      \
          ph.specific_prim_address + 1 + int(unsigned(glyph_index) / 2u);      \
      vec4_scalar gdata = fetch_from_gpu_cache_1(glyph_address);               \
      vec2_scalar glyph_offset = (unsigned(glyph_index) % 2u == 1u)            \
                                     ? vec2_scalar(gdata.z, gdata.w)           \
                                     : vec2_scalar(gdata.x, gdata.y);          \
      glyph_offset += ph.local_rect.p0;                                        \
      /* fetch_glyph_resource, ps_text_run.glsl:61-64 */                       \
      vec4_scalar res_uv_rect = fetch_from_gpu_cache(resource_address, 0);     \
      vec4_scalar res1 = fetch_from_gpu_cache(resource_address, 1);            \
      vec2_scalar res_offset = vec2_scalar(res1.x, res1.y);                    \
      float res_scale = res1.z;                                                \
      vec2_scalar snap_bias = get_snap_bias(subpx_dir);                        \
      BrushVertexInfo vi;                                                      \
      vec2 f;                                                                  \
      if (GLYPH_TRANSFORM) {                                                   \
        /* ps_text_run.glsl:130-165, 206-216: glyphs rasterised in device */   \
        /* space, clipped to their raster rect by gl_ClipDistance */           \
        mat2_scalar glyph_transform =                                          \
            mat2_scalar(vec2_scalar(transform.m[0].x, transform.m[0].y),       \
                        vec2_scalar(transform.m[1].x, transform.m[1].y)) *     \
            task.device_pixel_scale;                                           \
        vec2_scalar glyph_translation =                                        \
            vec2_scalar(transform.m[3].x, transform.m[3].y) *                  \
            task.device_pixel_scale;                                           \
        mat2_scalar glyph_transform_inv = inverse(glyph_transform);            \
        vec2_scalar raster_glyph_offset =                                      \
            floor(glyph_transform * glyph_offset + snap_bias);                 \
        vec2_scalar raster_text_offset =                                       \
            floor(glyph_transform * text_offset + glyph_translation + 0.5f) -  \
            glyph_translation;                                                 \
        vec2_scalar glyph_origin =                                             \
            res_offset + raster_glyph_offset + raster_text_offset;             \
        RectWithEndpoint glyph_rect{                                           \
            glyph_origin,                                                      \
            glyph_origin + vec2_scalar(res_uv_rect.z, res_uv_rect.w) -         \
                vec2_scalar(res_uv_rect.x, res_uv_rect.y)};                    \
        RectWithEndpoint local_rect =                                          \
            transform_rect(glyph_rect, glyph_transform_inv);                   \
        vec2 local_pos = mix(local_rect.p0, local_rect.p1, aPosition);         \
        if (rect_inside_rect(local_rect, ph.local_clip_rect)) {                \
          local_pos = glyph_transform_inv *                                    \
                      mix(glyph_rect.p0, glyph_rect.p1, aPosition);            \
        }                                                                      \
        vi = write_vertex(local_pos, ph.local_clip_rect, ph.z, transform,      \
                          task);                                               \
        f = (glyph_transform * vi.local_pos - glyph_rect.p0) /                 \
            rect_size(glyph_rect);                                             \
        gl_ClipDistance[0] = f.x;                                              \
        gl_ClipDistance[1] = f.y;                                              \
        gl_ClipDistance[2] = 1.0f - f.x;                                       \
        gl_ClipDistance[3] = 1.0f - f.y;                                       \
      } else {                                                                 \
      /* ps_text_run.glsl:155-190 */                                           \
      float raster_scale = float(ph.user_data.x) / 65535.0f;                   \
      float glyph_raster_scale = raster_scale * task.device_pixel_scale;       \
      float glyph_scale_inv = res_scale / glyph_raster_scale;                  \
      vec2_scalar raster_glyph_offset =                                        \
          floor(glyph_offset * glyph_raster_scale + snap_bias) / res_scale;    \
      vec2_scalar glyph_origin =                                               \
          glyph_scale_inv * (res_offset + raster_glyph_offset) + text_offset;  \
      RectWithEndpoint glyph_rect{                                             \
          glyph_origin,                                                        \
          glyph_origin + glyph_scale_inv * (vec2_scalar(res_uv_rect.z,         \
                                                        res_uv_rect.w) -       \
                                            vec2_scalar(res_uv_rect.x,         \
                                                        res_uv_rect.y))};      \
      vec2 local_pos = mix(glyph_rect.p0, glyph_rect.p1, aPosition);           \
      vi = write_vertex(local_pos, ph.local_clip_rect, ph.z, transform, task); \
      f = (vi.local_pos - glyph_rect.p0) / rect_size(glyph_rect);              \
      }                                                                        \
      write_clip(clip_area, task);                                             \
      /* ps_text_run.glsl:219-255 with SWGL_BLEND */                           \
      switch (color_mode) {                                                    \
        case 0: /* COLOR_MODE_ALPHA */                                         \
          v_mask_swizzle = vec3_scalar(0.0f, 1.0f, 1.0f);                      \
          v_color = text_color;                                                \
          break;                                                               \
        case 2: /* COLOR_MODE_BITMAP_SHADOW */                                 \
          swgl_blendDropShadow(text_color);                                    \
          v_mask_swizzle = vec3_scalar(1.0f, 0.0f, 0.0f);                      \
          v_color = vec4_scalar(1.0f);                                         \
          break;                                                               \
        case 3: /* COLOR_MODE_COLOR_BITMAP */                                  \
          v_mask_swizzle = vec3_scalar(1.0f, 0.0f, 0.0f);                      \
          v_color = vec4_scalar(text_color.w);                                 \
          break;                                                               \
        case 1: /* COLOR_MODE_SUBPX_DUAL_SOURCE */                             \
          swgl_blendSubpixelText(text_color);                                  \
          v_mask_swizzle = vec3_scalar(1.0f, 0.0f, 0.0f);                      \
          v_color = vec4_scalar(1.0f);                                         \
          break;                                                               \
        default:                                                               \
          v_mask_swizzle = vec3_scalar(0.0f, 0.0f, 0.0f);                      \
          v_color = vec4_scalar(1.0f);                                         \
      }                                                                        \
      ivec2_scalar ts = textureSize(sColor0, 0);                               \
      vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));        \
      vec2_scalar st0 = vec2_scalar(res_uv_rect.x, res_uv_rect.y) / texture_size; \
      vec2_scalar st1 = vec2_scalar(res_uv_rect.z, res_uv_rect.w) / texture_size; \
      v_uv = mix(st0, st1, f);                                                 \
      v_uv_bounds = (res_uv_rect + vec4_scalar(0.5f, 0.5f, -0.5f, -0.5f)) /    \
                    vec4_scalar(texture_size.x, texture_size.y,                \
                                texture_size.x, texture_size.y);               \
    }                                                                          \
    void store_clip(InterpOutputsPlain*, int) {}                               \
    void store_clip(InterpOutputsClip* dest, int n) {                          \
      dest->swgl_ClipDistance.x = get_nth(gl_ClipDistance[0], n);              \
      dest->swgl_ClipDistance.y = get_nth(gl_ClipDistance[1], n);              \
      dest->swgl_ClipDistance.z = get_nth(gl_ClipDistance[2], n);              \
      dest->swgl_ClipDistance.w = get_nth(gl_ClipDistance[3], n);              \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {   \
      for (int n = 0; n < 4; n++) {                                            \
        auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);               \
        store_clip(dest, n);                                                   \
        dest->v_uv = get_nth(v_uv, n);                                         \
        dest_ptr += stride;                                                    \
      }                                                                        \
    }                                                                          \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() {                                                            \
      WRSH_VERT_WIRING(Self)                                                   \
      if (GLYPH_TRANSFORM) enable_clip_distance(); /* lib.rs:3645-3647 */      \
    }                                                                          \
  };                                                                           \
  struct NAME##_frag : FragmentShaderImpl, NAME##_vert {                       \
    typedef NAME##_frag Self;                                                  \
    typedef typename NAME##_vert::InterpOutputs InterpInputs;                  \
    InterpInputs interp_step;                                                  \
    static void read_interp_inputs(FragmentShaderImpl* impl,                   \
                                   const void* init_, const void* step_) {     \
      Self* self = (Self*)impl;                                                \
      const InterpInputs* init = (const InterpInputs*)init_;                   \
      const InterpInputs* step = (const InterpInputs*)step_;                   \
      self->v_uv = init_interp(init->v_uv, step->v_uv);                        \
      self->interp_step.v_uv = step->v_uv * 4.0f;                              \
    }                                                                          \
    ALWAYS_INLINE void step_interp_inputs(int steps = 4) {                     \
      float chunks = steps * 0.25f;                                            \
      v_uv += interp_step.v_uv * chunks;                                       \
    }                                                                          \
    /* the perspective entry points glsl-to-cxx emits for a program with a */  \
    /* varying (lib.rs:660-690, 716-741, 3576-3590) */                         \
    struct InterpPerspective {                                                 \
      vec2 v_uv;                                                               \
    };                                                                         \
    InterpPerspective interp_perspective;                                      \
    static void read_perspective_inputs(FragmentShaderImpl* impl,              \
                                        const void* init_, const void* step_) { \
      Self* self = (Self*)impl;                                                \
      const InterpInputs* init = (const InterpInputs*)init_;                   \
      const InterpInputs* step = (const InterpInputs*)step_;                   \
      Float w = 1.0f / self->gl_FragCoord.w;                                   \
      self->interp_perspective.v_uv = init_interp(init->v_uv, step->v_uv);     \
      self->v_uv = self->interp_perspective.v_uv * w;                          \
      self->interp_step.v_uv = step->v_uv * 4.0f;                              \
    }                                                                          \
    ALWAYS_INLINE void step_perspective_inputs(int steps = 4) {                \
      step_perspective(steps);                                                 \
      float chunks = steps * 0.25f;                                            \
      Float w = 1.0f / gl_FragCoord.w;                                         \
      interp_perspective.v_uv += interp_step.v_uv * chunks;                    \
      v_uv = w * interp_perspective.v_uv;                                      \
    }                                                                          \
    WRSH_FRAG_ABI_PERSPECTIVE(Self)                                            \
    /* text_fs + main, ps_text_run.glsl:271-318 */                             \
    void main() {                                                              \
      vec2 tc = clamp(v_uv, vec2_scalar(v_uv_bounds.x, v_uv_bounds.y),         \
                      vec2_scalar(v_uv_bounds.z, v_uv_bounds.w));              \
      vec4 mask = texture(sColor0, tc);                                        \
      if (v_mask_swizzle.z != 0.0f) mask = mask.sel(X, X, X, X);               \
      if (!(DUAL_SOURCE)) {                                                    \
        vec3 rgb = mask.sel(X, Y, Z) * v_mask_swizzle.x +                      \
                   mask.sel(W, W, W) * v_mask_swizzle.y;                       \
        mask = vec4(rgb, mask.w);                                              \
      }                                                                        \
      vec4 color = vec4(v_color) * mask;                                       \
      float clip_mask = 1.0f; /* do_clip() */                                  \
      color *= clip_mask;                                                      \
      gl_FragColor = color;                                                    \
    }                                                                          \
    /* ps_text_run.glsl:320-338 */                                             \
    void swgl_drawSpanRGBA8() {                                                \
      if (v_mask_swizzle.x != 0.0f && v_mask_swizzle.x != 1.0f) {              \
        return;                                                                \
      }                                                                        \
      if (DUAL_SOURCE) {                                                       \
        swgl_commitTextureLinearRGBA8(sColor0, v_uv, v_uv_bounds);             \
      } else if (swgl_isTextureR8(sColor0)) {                                  \
        swgl_commitTextureLinearColorR8ToRGBA8(sColor0, v_uv, v_uv_bounds,     \
                                               v_color);                       \
      } else {                                                                 \
        swgl_commitTextureLinearColorRGBA8(sColor0, v_uv, v_uv_bounds,         \
                                           v_color);                           \
      }                                                                        \
    }                                                                          \
    WRSH_FRAG_ABI(Self)                                                        \
    static int draw_span_RGBA8(FragmentShaderImpl* impl) {                     \
      Self* self = (Self*)impl;                                                \
      DISPATCH_DRAW_SPAN(self, RGBA8);                                         \
    }                                                                          \
    NAME##_frag() {                                                            \
      WRSH_FRAG_WIRING()                                                       \
      WRSH_FRAG_WIRING_PERSPECTIVE()                                           \
      draw_span_RGBA8_func = &draw_span_RGBA8;                                 \
    }                                                                          \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_PS_TEXT_RUN(ps_text_run_ALPHA_PASS_TEXTURE_2D,
                 "ps_text_run ALPHA_PASS,TEXTURE_2D", false, false)
WRSH_PS_TEXT_RUN(ps_text_run_ALPHA_PASS_DUAL_SOURCE_BLENDING_TEXTURE_2D,
                 "ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D", true, false)
/* the GLYPH_TRANSFORM keys (shader_features.rs:222-226): text whose glyphs were rasterised under the run's 2-D transform */
WRSH_PS_TEXT_RUN(ps_text_run_ALPHA_PASS_GLYPH_TRANSFORM_TEXTURE_2D,
                 "ps_text_run ALPHA_PASS,GLYPH_TRANSFORM,TEXTURE_2D", false, true)
WRSH_PS_TEXT_RUN(ps_text_run_ALPHA_PASS_DUAL_SOURCE_BLENDING_GLYPH_TRANSFORM_TEXTURE_2D,
                 "ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,GLYPH_TRANSFORM,TEXTURE_2D", true, true)
