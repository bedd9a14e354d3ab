// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated
// headers of shader keys "brush_solid" and "brush_solid ALPHA_PASS"
// (webrender_build/src/shader_features.rs:99-106). Restates
// webrender/res/brush_solid.glsl:22-58 on top of brush_base.h.
// Under SWGL both variants compute the same values: ALPHA_PASS only multiplies
// by antialias_brush() == 1.0 and do_clip() == 1.0 (brush.glsl:226-255,
// prim_shared.glsl:224-228), AA and clip masking being done natively by swgl.

#define WRSH_BRUSH_SOLID(NAME, KEYSTR, ALPHA_PASS)                             \
  struct NAME##_vert : wrsh::brush_vert_base<NAME##_vert> {                    \
    typedef NAME##_vert Self;                                                  \
    static constexpr int VECS_PER_SPECIFIC_BRUSH = 1;                          \
    vec4_scalar v_color;                                                       \
    struct InterpOutputs {};                                                   \
    void brush_vs(wrsh::BrushVertexInfo, int prim_address,                     \
                  wrsh::RectWithEndpoint, wrsh::RectWithEndpoint,              \
                  ivec4_scalar prim_user_data, int, mat4_scalar,               \
                  wrsh::PictureTask, int, vec4_scalar) {                       \
      vec4_scalar color = fetch_from_gpu_cache_1(prim_address);                \
      float opacity = float(prim_user_data.x) / 65535.0f;                      \
      v_color = color * opacity;                                               \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char*, size_t) {}                  \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() { WRSH_VERT_WIRING(Self) }                                   \
  };                                                                           \
  struct NAME##_frag : FragmentShaderImpl, NAME##_vert {                       \
    typedef NAME##_frag Self;                                                  \
    static void read_interp_inputs(FragmentShaderImpl*, const void*,           \
                                   const void*) {}                             \
    ALWAYS_INLINE void step_interp_inputs(int = 4) {}                          \
    void main() {                                                              \
      vec4 color = vec4(v_color);                                              \
      if (ALPHA_PASS) {                                                        \
        color *= 1.0f; /* antialias_brush() */                                 \
        float clip_alpha = 1.0f; /* do_clip() */                               \
        color *= clip_alpha;                                                   \
      }                                                                        \
      gl_FragColor = color;                                                    \
    }                                                                          \
    void swgl_drawSpanRGBA8() { swgl_commitSolidRGBA8(v_color); }              \
    void swgl_drawSpanR8() { swgl_commitSolidR8(v_color.x); }                  \
    WRSH_FRAG_ABI(Self)                                                        \
    static int draw_span_RGBA8(FragmentShaderImpl* impl) {                     \
      Self* self = (Self*)impl;                                                \
      DISPATCH_DRAW_SPAN(self, RGBA8);                                         \
    }                                                                          \
    static int draw_span_R8(FragmentShaderImpl* impl) {                        \
      Self* self = (Self*)impl;                                                \
      DISPATCH_DRAW_SPAN(self, R8);                                            \
    }                                                                          \
    NAME##_frag() {                                                            \
      WRSH_FRAG_WIRING()                                                       \
      draw_span_RGBA8_func = &draw_span_RGBA8;                                 \
      draw_span_R8_func = &draw_span_R8;                                       \
    }                                                                          \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_BRUSH_SOLID(brush_solid, "brush_solid", false)
WRSH_BRUSH_SOLID(brush_solid_ALPHA_PASS, "brush_solid ALPHA_PASS", true)
