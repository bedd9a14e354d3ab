// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "brush_image TEXTURE_2D" and "brush_image ALPHA_PASS,TEXTURE_2D"
// (the "fast" image brush: webrender_build/src/shader_features.rs:146-152,
// renderer/shade.rs:981) and of the full image brush "brush_image
// [ALPHA_PASS,]ANTIALIASING,REPETITION,TEXTURE_2D" (shader_features.rs:153-157,
// shade.rs:995). Restates webrender/res/brush_image.glsl:54-314 (VS), 318-377
// (FS), 380-428 (span) on brush_base.h. ANTIALIASING changes nothing under swgl
// (antialias_brush() is 1.0 with SWGL_ANTIALIAS; the edge mask goes through
// swgl_antiAlias in brush_base.h).
// RASTER_SCREEN sources (get_image_quad_uv): restated below.

#define WRSH_BRUSH_IMAGE(NAME, KEYSTR, ALPHA_PASS, REPETITION, DUAL)           \
  struct NAME##_vert : wrsh::brush_vert_base<NAME##_vert> {                    \
    typedef NAME##_vert Self;                                                  \
    static constexpr int VECS_PER_SPECIFIC_BRUSH = 3;                          \
    vec2 v_uv;                                                                 \
    vec4_scalar v_color;                                                       \
    vec2_scalar v_mask_swizzle;                                                \
    vec2_scalar v_tile_repeat_bounds;                                          \
    vec4_scalar v_uv_bounds, v_uv_sample_bounds;                               \
    vec2_scalar v_perspective;                                                 \
    struct InterpOutputs {                                                     \
      vec2_scalar v_uv;                                                        \
    };                                                                         \
    void brush_vs(wrsh::BrushVertexInfo vi, int prim_address,                  \
                  wrsh::RectWithEndpoint prim_rect,                            \
                  wrsh::RectWithEndpoint segment_rect,                         \
                  ivec4_scalar prim_user_data, int specific_resource_address,  \
                  mat4_scalar, wrsh::PictureTask, int brush_flags,             \
                  vec4_scalar segment_data) {                                  \
      using namespace wrsh;                                                    \
      vec4_scalar color = fetch_from_gpu_cache(prim_address, 0);               \
      vec4_scalar raw2 = fetch_from_gpu_cache(prim_address, 2);                \
      vec2_scalar stretch_size = vec2_scalar(raw2.x, raw2.y);                  \
      ivec2_scalar ts = textureSize(sColor0, 0);                               \
      vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));        \
      vec4_scalar res0 = fetch_from_gpu_cache(specific_resource_address, 0);   \
      vec2_scalar uv0 = vec2_scalar(res0.x, res0.y);                           \
      vec2_scalar uv1 = vec2_scalar(res0.z, res0.w);                           \
      RectWithEndpoint local_rect = prim_rect;                                 \
      if (stretch_size.x < 0.0f) {                                             \
        stretch_size = rect_size(local_rect);                                  \
      }                                                                        \
      if ((brush_flags & BRUSH_FLAG_SEGMENT_RELATIVE) != 0) {                  \
        local_rect = segment_rect;                                             \
        stretch_size = rect_size(local_rect);                                  \
        if ((brush_flags & BRUSH_FLAG_TEXEL_RECT) != 0) {                      \
          vec2_scalar uv_size = vec2_scalar(res0.z, res0.w) - vec2_scalar(res0.x, res0.y); \
          uv0 = vec2_scalar(res0.x, res0.y) + vec2_scalar(segment_data.x, segment_data.y) * uv_size; \
          uv1 = vec2_scalar(res0.x, res0.y) + vec2_scalar(segment_data.z, segment_data.w) * uv_size; \
        }                                                                      \
        if (REPETITION) {                                                      \
          if ((brush_flags & BRUSH_FLAG_TEXEL_RECT) != 0) {                    \
            vec2_scalar repeated_stretch_size = stretch_size;                  \
            vec2_scalar horizontal_uv_size = uv1 - uv0;                        \
            vec2_scalar vertical_uv_size = uv1 - uv0;                          \
            if ((brush_flags & BRUSH_FLAG_SEGMENT_NINEPATCH_MIDDLE) != 0) {    \
              repeated_stretch_size = segment_rect.p0 - prim_rect.p0;          \
              float epsilon = 0.001f;                                          \
              vertical_uv_size.x = uv0.x - res0.x;                             \
              if (vertical_uv_size.x < epsilon ||                              \
                  repeated_stretch_size.x < epsilon) {                         \
                vertical_uv_size.x = res0.z - uv1.x;                           \
                repeated_stretch_size.x = prim_rect.p1.x - segment_rect.p1.x;  \
              }                                                                \
              horizontal_uv_size.y = uv0.y - res0.y;                           \
              if (horizontal_uv_size.y < epsilon ||                            \
                  repeated_stretch_size.y < epsilon) {                         \
                horizontal_uv_size.y = res0.w - uv1.y;                         \
                repeated_stretch_size.y = prim_rect.p1.y - segment_rect.p1.y;  \
              }                                                                \
            }                                                                  \
            if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_X) != 0) {            \
              float uv_ratio = horizontal_uv_size.x / horizontal_uv_size.y;    \
              stretch_size.x = repeated_stretch_size.y * uv_ratio;             \
            }                                                                  \
            if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_Y) != 0) {            \
              float uv_ratio = vertical_uv_size.y / vertical_uv_size.x;        \
              stretch_size.y = repeated_stretch_size.x * uv_ratio;             \
            }                                                                  \
          } else {                                                             \
            if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_X) != 0) {            \
              stretch_size.x = segment_data.z - segment_data.x;                \
            }                                                                  \
            if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_Y) != 0) {            \
              stretch_size.y = segment_data.w - segment_data.y;                \
            }                                                                  \
          }                                                                    \
          if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_X_ROUND) != 0) {        \
            float segment_rect_width = segment_rect.p1.x - segment_rect.p0.x;  \
            float nx = max(1.0f, round(segment_rect_width / stretch_size.x));  \
            stretch_size.x = segment_rect_width / nx;                          \
          }                                                                    \
          if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_Y_ROUND) != 0) {        \
            float segment_rect_height = segment_rect.p1.y - segment_rect.p0.y; \
            float ny = max(1.0f, round(segment_rect_height / stretch_size.y)); \
            stretch_size.y = segment_rect_height / ny;                         \
          }                                                                    \
        }                                                                      \
      }                                                                        \
      float perspective_interpolate =                                          \
          (brush_flags & BRUSH_FLAG_PERSPECTIVE_INTERPOLATION) != 0 ? 1.0f : 0.0f; \
      v_perspective.x = perspective_interpolate;                               \
      if ((brush_flags & BRUSH_FLAG_NORMALIZED_UVS) != 0) {                    \
        uv0 *= texture_size;                                                   \
        uv1 *= texture_size;                                                   \
      }                                                                        \
      vec2_scalar min_uv = min(uv0, uv1);                                      \
      vec2_scalar max_uv = max(uv0, uv1);                                      \
      v_uv_sample_bounds =                                                     \
          vec4_scalar(min_uv.x + 0.5f, min_uv.y + 0.5f, max_uv.x - 0.5f,       \
                      max_uv.y - 0.5f) /                                       \
          vec4_scalar(texture_size.x, texture_size.y, texture_size.x,          \
                      texture_size.y);                                         \
      vec2 f = (vi.local_pos - local_rect.p0) / rect_size(local_rect);         \
      /* RASTER_SCREEN: get_image_quad_uv, brush_image.glsl:200-205, prim_shared.glsl:204-210 */ \
      if (prim_user_data.y != 0) {                                             \
        vec4_scalar st_tl = fetch_from_gpu_cache(specific_resource_address + 2, 0); \
        vec4_scalar st_tr = fetch_from_gpu_cache(specific_resource_address + 2, 1); \
        vec4_scalar st_bl = fetch_from_gpu_cache(specific_resource_address + 2, 2); \
        vec4_scalar st_br = fetch_from_gpu_cache(specific_resource_address + 2, 3); \
        vec4 x = mix(vec4(st_tl), vec4(st_tr), f.x);                           \
        vec4 y = mix(vec4(st_bl), vec4(st_br), f.x);                           \
        vec4 z = mix(x, y, f.y);                                               \
        f = z.sel(X, Y) / z.w;                                                 \
      }                                                                        \
      int color_mode = prim_user_data.x & 0xffff;                              \
      int blend_mode = prim_user_data.x >> 16;                                 \
      vec2_scalar repeat = rect_size(local_rect) / stretch_size;               \
      v_uv = mix(uv0, uv1, f) - min_uv;                                        \
      v_uv *= repeat;                                                          \
      vec2_scalar normalized_offset = vec2_scalar(0.0f);                       \
      if (REPETITION) {                                                        \
        if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_X_CENTERED) != 0) {       \
          normalized_offset.x = 1.0f - fract(repeat.x * 0.5f + 0.5f);          \
        }                                                                      \
        if ((brush_flags & BRUSH_FLAG_SEGMENT_REPEAT_Y_CENTERED) != 0) {       \
          normalized_offset.y = 1.0f - fract(repeat.y * 0.5f + 0.5f);          \
        }                                                                      \
        v_uv += normalized_offset * (max_uv - min_uv);                         \
      }                                                                        \
      v_uv /= texture_size;                                                    \
      if (perspective_interpolate == 0.0f) {                                   \
        v_uv *= vi.world_pos.w;                                                \
      }                                                                        \
      v_uv_bounds = vec4_scalar(min_uv.x, min_uv.y, max_uv.x, max_uv.y) /      \
                    vec4_scalar(texture_size.x, texture_size.y, texture_size.x, \
                                texture_size.y);                               \
      if (REPETITION) {                                                        \
        v_uv /= vec2_scalar(v_uv_bounds.z - v_uv_bounds.x,                     \
                            v_uv_bounds.w - v_uv_bounds.y);                    \
      }                                                                        \
      if (ALPHA_PASS) {                                                        \
        v_tile_repeat_bounds = repeat + normalized_offset;                     \
        float opacity = float(prim_user_data.z) / 65535.0f;                    \
        switch (blend_mode) {                                                  \
          case 0: /* BLEND_MODE_ALPHA */                                       \
            color.w *= opacity;                                                \
            break;                                                             \
          case 1:                                                              \
          default:                                                             \
            color *= opacity;                                                  \
            break;                                                             \
        }                                                                      \
        switch (color_mode) {                                                  \
          case 0: /* COLOR_MODE_ALPHA */                                       \
          case 2: /* COLOR_MODE_BITMAP_SHADOW */                               \
            swgl_blendDropShadow(color);                                       \
            v_mask_swizzle = vec2_scalar(1.0f, 0.0f);                          \
            v_color = vec4_scalar(1.0f);                                       \
            break;                                                             \
          case 4: /* COLOR_MODE_IMAGE */                                       \
            v_mask_swizzle = vec2_scalar(1.0f, 0.0f);                          \
            v_color = color;                                                   \
            break;                                                             \
          case 3: /* COLOR_MODE_COLOR_BITMAP */                                \
            v_mask_swizzle = vec2_scalar(1.0f, 0.0f);                          \
            v_color = vec4_scalar(color.w);                                    \
            break;                                                             \
          case 1: /* COLOR_MODE_SUBPX_DUAL_SOURCE */                           \
            v_mask_swizzle = vec2_scalar(color.w, 0.0f);                       \
            v_color = color;                                                   \
            break;                                                             \
          case 5: /* COLOR_MODE_MULTIPLY_DUAL_SOURCE */                        \
            v_mask_swizzle = vec2_scalar(-color.w, color.w);                   \
            v_color = color;                                                   \
            break;                                                             \
          default:                                                             \
            v_mask_swizzle = vec2_scalar(0.0f);                                \
            v_color = vec4_scalar(1.0f);                                       \
        }                                                                      \
      }                                                                        \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {   \
      for (int n = 0; n < 4; n++) {                                            \
        auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);               \
        dest->v_uv = get_nth(v_uv, n);                                         \
        dest_ptr += stride;                                                    \
      }                                                                        \
    }                                                                          \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() { WRSH_VERT_WIRING(Self) }                                   \
  };                                                                           \
  struct NAME##_frag : FragmentShaderImpl, NAME##_vert {                       \
    typedef NAME##_frag Self;                                                  \
    typedef NAME##_vert::InterpOutputs InterpInputs;                           \
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
    static void run_perspective(FragmentShaderImpl* impl) {                    \
      Self* self = (Self*)impl;                                                \
      self->main_w();                                                          \
      self->step_perspective_inputs();                                         \
    }                                                                          \
    static void skip_perspective(FragmentShaderImpl* impl, int steps) {        \
      Self* self = (Self*)impl;                                                \
      self->step_perspective_inputs(steps);                                    \
    }                                                                          \
    /* main() as generated, with gl_FragCoord.w read per lane (:343-377); */   \
    /* the 2-D main() below is the same function with w == 1 folded in. */     \
    void main_w() {                                                            \
      Float perspective_divisor =                                              \
          mix(gl_FragCoord.w, Float(1.0f), Float(v_perspective.x));            \
      vec2 repeated_uv = repeated_uvs(perspective_divisor);                    \
      vec2 uv = clamp(repeated_uv,                                             \
                      vec2_scalar(v_uv_sample_bounds.x, v_uv_sample_bounds.y), \
                      vec2_scalar(v_uv_sample_bounds.z, v_uv_sample_bounds.w)); \
      vec4 texel = texture(sColor0, uv);                                       \
      vec4 color;                                                              \
      if (ALPHA_PASS) {                                                        \
        float alpha = 1.0f;                                                    \
        if (!DUAL) { /* brush_image.glsl:369-371 */                            \
          vec3 rgb = texel.sel(X, Y, Z) * v_mask_swizzle.x +                   \
                     texel.sel(W, W, W) * v_mask_swizzle.y;                    \
          texel = vec4(rgb, texel.w);                                          \
        }                                                                      \
        vec4 alpha_mask = texel * alpha;                                       \
        color = vec4(v_color) * alpha_mask;                                    \
        color *= 1.0f; /* do_clip() */                                         \
        if (DUAL) { /* :376-378, brush.glsl:251-253: oFragBlend = frag.blend * clip_alpha */ \
          vec4 blend = alpha_mask * v_mask_swizzle.x +                         \
                       alpha_mask.sel(W, W, W, W) * v_mask_swizzle.y;          \
          gl_SecondaryFragColor = blend * 1.0f;                                \
        }                                                                      \
      } else {                                                                 \
        color = texel;                                                         \
      }                                                                        \
      gl_FragColor = color;                                                    \
    }                                                                          \
    /* compute_repeated_uvs: :318-341 */                                       \
    template <typename T>                                                      \
    vec2 repeated_uvs(T perspective_divisor) const {                           \
      if (REPETITION) {                                                        \
        vec2_scalar uv_size = vec2_scalar(v_uv_bounds.z - v_uv_bounds.x,       \
                                          v_uv_bounds.w - v_uv_bounds.y);      \
        if (ALPHA_PASS) {                                                      \
          vec2 local_uv = v_uv * perspective_divisor;                          \
          local_uv = max(local_uv, vec2(Float(0.0f)));                              \
          vec2 repeated_uv = fract(local_uv) * uv_size +                       \
                             vec2_scalar(v_uv_bounds.x, v_uv_bounds.y);        \
          repeated_uv.x = if_then_else(local_uv.x >= v_tile_repeat_bounds.x,   \
                                       Float(v_uv_bounds.z), repeated_uv.x);   \
          repeated_uv.y = if_then_else(local_uv.y >= v_tile_repeat_bounds.y,   \
                                       Float(v_uv_bounds.w), repeated_uv.y);   \
          return repeated_uv;                                                  \
        }                                                                      \
        return fract(v_uv * perspective_divisor) * uv_size +                   \
               vec2_scalar(v_uv_bounds.x, v_uv_bounds.y);                      \
      }                                                                        \
      return v_uv * perspective_divisor +                                      \
             vec2_scalar(v_uv_bounds.x, v_uv_bounds.y);                        \
    }                                                                          \
    /* brush_fs + main, brush_image.glsl:343-377 (2-D path: gl_FragCoord.w == 1) */ \
    void main() {                                                              \
      float perspective_divisor = mix(1.0f, 1.0f, v_perspective.x);            \
      vec2 repeated_uv = repeated_uvs(perspective_divisor);                    \
      vec2 uv = clamp(repeated_uv,                                             \
                      vec2_scalar(v_uv_sample_bounds.x, v_uv_sample_bounds.y), \
                      vec2_scalar(v_uv_sample_bounds.z, v_uv_sample_bounds.w)); \
      vec4 texel = texture(sColor0, uv);                                       \
      vec4 color;                                                              \
      if (ALPHA_PASS) {                                                        \
        float alpha = 1.0f;                                                    \
        if (!DUAL) { /* brush_image.glsl:369-371 */                            \
          vec3 rgb = texel.sel(X, Y, Z) * v_mask_swizzle.x +                   \
                     texel.sel(W, W, W) * v_mask_swizzle.y;                    \
          texel = vec4(rgb, texel.w);                                          \
        }                                                                      \
        vec4 alpha_mask = texel * alpha;                                       \
        color = vec4(v_color) * alpha_mask;                                    \
        color *= 1.0f; /* do_clip() */                                         \
        if (DUAL) { /* :376-378, brush.glsl:251-253: oFragBlend = frag.blend * clip_alpha */ \
          vec4 blend = alpha_mask * v_mask_swizzle.x +                         \
                       alpha_mask.sel(W, W, W, W) * v_mask_swizzle.y;          \
          gl_SecondaryFragColor = blend * 1.0f;                                \
        }                                                                      \
      } else {                                                                 \
        color = texel;                                                         \
      }                                                                        \
      gl_FragColor = color;                                                    \
    }                                                                          \
    void swgl_drawSpanRGBA8() {                                                \
      if (!swgl_isTextureRGBA8(sColor0)) {                                     \
        return;                                                                \
      }                                                                        \
      if (ALPHA_PASS) {                                                        \
        if (v_mask_swizzle != vec2_scalar(1.0f, 0.0f)) {                       \
          return;                                                              \
        }                                                                      \
      }                                                                        \
      float perspective_divisor = mix(1.0f, 1.0f, v_perspective.x);            \
      vec2 uv = REPETITION ? v_uv * perspective_divisor                        \
                           : repeated_uvs(perspective_divisor);                \
      if (ALPHA_PASS) {                                                        \
        if (v_color != vec4_scalar(1.0f)) {                                    \
          if (REPETITION) {                                                    \
            swgl_commitTextureRepeatColorRGBA8(sColor0, uv,                    \
                                               v_tile_repeat_bounds,           \
                                               v_uv_bounds,                    \
                                               v_uv_sample_bounds, v_color);   \
          } else {                                                             \
            swgl_commitTextureColorRGBA8(sColor0, uv, v_uv_sample_bounds,      \
                                         v_color);                             \
          }                                                                    \
          return;                                                              \
        }                                                                      \
      }                                                                        \
      if (REPETITION) {                                                        \
        if (ALPHA_PASS) {                                                      \
          swgl_commitTextureRepeatRGBA8(sColor0, uv, v_tile_repeat_bounds,     \
                                        v_uv_bounds, v_uv_sample_bounds);      \
        } else {                                                               \
          swgl_commitTextureRepeatRGBA8(sColor0, uv, vec2_scalar(0.0f),        \
                                        v_uv_bounds, v_uv_sample_bounds);      \
        }                                                                      \
      } else {                                                                 \
        swgl_commitTextureRGBA8(sColor0, uv, v_uv_sample_bounds);              \
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
      /* no span function under ALPHA_PASS + DUAL_SOURCE_BLENDING (:386) */    \
      if (!(ALPHA_PASS && DUAL)) draw_span_RGBA8_func = &draw_span_RGBA8;      \
    }                                                                          \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_BRUSH_IMAGE(brush_image_TEXTURE_2D, "brush_image TEXTURE_2D", false, false, false)
WRSH_BRUSH_IMAGE(brush_image_ALPHA_PASS_TEXTURE_2D, "brush_image ALPHA_PASS,TEXTURE_2D", true, false, false)
WRSH_BRUSH_IMAGE(brush_image_ANTIALIASING_REPETITION_TEXTURE_2D,
                 "brush_image ANTIALIASING,REPETITION,TEXTURE_2D", false, true, false)
WRSH_BRUSH_IMAGE(brush_image_ALPHA_PASS_ANTIALIASING_REPETITION_TEXTURE_2D,
                 "brush_image ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D", true, true, false)
// the DUAL_SOURCE_BLENDING keys (shader_features.rs:163-167): what BlendMode::SubpixelDualSource / MultiplyDualSource batches are
// drawn with (shade.rs:462-467); the fragment stage writes a second colour, blended with GL_ONE, GL_ONE_MINUS_SRC1_COLOR
WRSH_BRUSH_IMAGE(brush_image_ALPHA_PASS_DUAL_SOURCE_BLENDING_TEXTURE_2D,
                 "brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D", true, false, true)
WRSH_BRUSH_IMAGE(brush_image_ALPHA_PASS_ANTIALIASING_DUAL_SOURCE_BLENDING_REPETITION_TEXTURE_2D,
                 "brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_2D", true, true, true)
