// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "brush_image TEXTURE_2D" and "brush_image ALPHA_PASS,TEXTURE_2D"
// (the "fast" image brush: webrender_build/src/shader_features.rs:146-152,
// renderer/shade.rs:981). Restates webrender/res/brush_image.glsl:54-314 (VS
// without WR_FEATURE_REPETITION), 343-377 (FS), 380-428 (span) on brush_base.h.
// RASTER_SCREEN quads (get_image_quad_uv) are not restated: scenes use local raster space.

#define WRSH_BRUSH_IMAGE(NAME, KEYSTR, ALPHA_PASS)                             \
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
      int color_mode = prim_user_data.x & 0xffff;                              \
      int blend_mode = prim_user_data.x >> 16;                                 \
      vec2_scalar repeat = rect_size(local_rect) / stretch_size;               \
      v_uv = mix(uv0, uv1, f) - min_uv;                                        \
      v_uv *= repeat;                                                          \
      vec2_scalar normalized_offset = vec2_scalar(0.0f);                       \
      v_uv /= texture_size;                                                    \
      if (perspective_interpolate == 0.0f) {                                   \
        v_uv *= vi.world_pos.w;                                                \
      }                                                                        \
      v_uv_bounds = vec4_scalar(min_uv.x, min_uv.y, max_uv.x, max_uv.y) /      \
                    vec4_scalar(texture_size.x, texture_size.y, texture_size.x, \
                                texture_size.y);                               \
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
    /* compute_repeated_uvs without REPETITION: :339 */                        \
    vec2 repeated_uvs(float perspective_divisor) const {                       \
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
        vec3 rgb = texel.sel(X, Y, Z) * v_mask_swizzle.x +                     \
                   texel.sel(W, W, W) * v_mask_swizzle.y;                      \
        texel = vec4(rgb, texel.w);                                            \
        vec4 alpha_mask = texel * alpha;                                       \
        color = vec4(v_color) * alpha_mask;                                    \
        color *= 1.0f; /* do_clip() */                                         \
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
      vec2 uv = repeated_uvs(perspective_divisor);                             \
      if (ALPHA_PASS) {                                                        \
        if (v_color != vec4_scalar(1.0f)) {                                    \
          swgl_commitTextureColorRGBA8(sColor0, uv, v_uv_sample_bounds,        \
                                       v_color);                               \
          return;                                                              \
        }                                                                      \
      }                                                                        \
      swgl_commitTextureRGBA8(sColor0, uv, v_uv_sample_bounds);                \
    }                                                                          \
    WRSH_FRAG_ABI(Self)                                                        \
    static int draw_span_RGBA8(FragmentShaderImpl* impl) {                     \
      Self* self = (Self*)impl;                                                \
      DISPATCH_DRAW_SPAN(self, RGBA8);                                         \
    }                                                                          \
    NAME##_frag() {                                                            \
      WRSH_FRAG_WIRING()                                                       \
      draw_span_RGBA8_func = &draw_span_RGBA8;                                 \
    }                                                                          \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_BRUSH_IMAGE(brush_image_TEXTURE_2D, "brush_image TEXTURE_2D", false)
WRSH_BRUSH_IMAGE(brush_image_ALPHA_PASS_TEXTURE_2D, "brush_image ALPHA_PASS,TEXTURE_2D", true)
