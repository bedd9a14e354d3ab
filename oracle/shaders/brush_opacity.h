// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "brush_opacity", "brush_opacity ALPHA_PASS" and their
// ANTIALIASING twins (webrender_build/src/shader_features.rs:124-133). Restates
// webrender/res/brush_opacity.glsl:25-63 (VS), 67-83 (FS), 85-91 (span) on
// brush_base.h. Under SWGL antialias_brush() is 1.0 (SWGL_ANTIALIAS), so the
// ANTIALIASING feature changes nothing in the shader itself.

#define WRSH_BRUSH_OPACITY(NAME, KEYSTR, ALPHA_PASS)                           \
  struct NAME##_vert : wrsh::brush_vert_base<NAME##_vert> {                    \
    typedef NAME##_vert Self;                                                  \
    static constexpr int VECS_PER_SPECIFIC_BRUSH = 3;                          \
    vec2 v_uv;                                                                 \
    vec4_scalar v_uv_sample_bounds;                                            \
    vec2_scalar v_opacity_perspective_vec;                                     \
    struct InterpOutputs {                                                     \
      vec2_scalar v_uv;                                                        \
    };                                                                         \
    void brush_vs(wrsh::BrushVertexInfo vi, int, wrsh::RectWithEndpoint local_rect, \
                  wrsh::RectWithEndpoint, ivec4_scalar prim_user_data, int,    \
                  mat4_scalar, wrsh::PictureTask, int brush_flags,             \
                  vec4_scalar) {                                               \
      using namespace wrsh;                                                    \
      vec4_scalar res0 = fetch_from_gpu_cache(prim_user_data.x, 0);            \
      vec2_scalar uv0 = vec2_scalar(res0.x, res0.y);                           \
      vec2_scalar uv1 = vec2_scalar(res0.z, res0.w);                           \
      ivec2_scalar ts = textureSize(sColor0, 0);                               \
      vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));        \
      vec2 f = (vi.local_pos - local_rect.p0) / rect_size(local_rect);         \
      /* get_image_quad_uv, prim_shared.glsl:204-210 */                        \
      {                                                                        \
        vec4_scalar st_tl = fetch_from_gpu_cache(prim_user_data.x + 2, 0);     \
        vec4_scalar st_tr = fetch_from_gpu_cache(prim_user_data.x + 2, 1);     \
        vec4_scalar st_bl = fetch_from_gpu_cache(prim_user_data.x + 2, 2);     \
        vec4_scalar st_br = fetch_from_gpu_cache(prim_user_data.x + 2, 3);     \
        vec4 x = mix(vec4(st_tl), vec4(st_tr), f.x);                           \
        vec4 y = mix(vec4(st_bl), vec4(st_br), f.x);                           \
        vec4 z = mix(x, y, f.y);                                               \
        f = z.sel(X, Y) / z.w;                                                 \
      }                                                                        \
      vec2 uv = mix(uv0, uv1, f);                                              \
      float perspective_interpolate =                                          \
          (brush_flags & BRUSH_FLAG_PERSPECTIVE_INTERPOLATION) != 0 ? 1.0f : 0.0f; \
      v_uv = uv / texture_size *                                               \
             mix(vi.world_pos.w, Float(1.0f), Float(perspective_interpolate)); \
      v_opacity_perspective_vec.y = perspective_interpolate;                   \
      v_uv_sample_bounds =                                                     \
          vec4_scalar(uv0.x + 0.5f, uv0.y + 0.5f, uv1.x - 0.5f, uv1.y - 0.5f) / \
          vec4_scalar(texture_size.x, texture_size.y, texture_size.x,          \
                      texture_size.y);                                         \
      v_opacity_perspective_vec.x =                                            \
          clamp(float(prim_user_data.y) / 65536.0f, 0.0f, 1.0f);               \
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
      vec2 v_uv;                                                                \
    };                                                                         \
    InterpPerspective interp_perspective;                                      \
    static void read_perspective_inputs(FragmentShaderImpl* impl,              \
                                        const void* init_, const void* step_) { \
      Self* self = (Self*)impl;                                                \
      const InterpInputs* init = (const InterpInputs*)init_;                   \
      const InterpInputs* step = (const InterpInputs*)step_;                   \
      Float w = 1.0f / self->gl_FragCoord.w;                                   \
      self->interp_perspective.v_uv = init_interp(init->v_uv, step->v_uv);        \
      self->v_uv = self->interp_perspective.v_uv * w;                            \
      self->interp_step.v_uv = step->v_uv * 4.0f;                                \
    }                                                                          \
    ALWAYS_INLINE void step_perspective_inputs(int steps = 4) {                \
      step_perspective(steps);                                                 \
      float chunks = steps * 0.25f;                                            \
      Float w = 1.0f / gl_FragCoord.w;                                         \
      interp_perspective.v_uv += interp_step.v_uv * chunks;                      \
      v_uv = w * interp_perspective.v_uv;                                        \
    }                                                                          \
    static void run_perspective(FragmentShaderImpl* impl) {                    \
      Self* self = (Self*)impl;                                                \
      self->shade(mix(self->gl_FragCoord.w, Float(1.0f), Float(self->v_opacity_perspective_vec.y)));                                                              \
      self->step_perspective_inputs();                                         \
    }                                                                          \
    static void skip_perspective(FragmentShaderImpl* impl, int steps) {        \
      Self* self = (Self*)impl;                                                \
      self->step_perspective_inputs(steps);                                    \
    }                                                                          \
    /* brush_fs + main, brush_opacity.glsl:67-83 (2-D path: gl_FragCoord.w == 1) */ \
    void main() { shade(Float(mix(1.0f, 1.0f, v_opacity_perspective_vec.y))); } \
    void shade(Float perspective_divisor) {                                    \
      vec2 uv = v_uv * perspective_divisor;                                    \
      uv = clamp(uv, vec2_scalar(v_uv_sample_bounds.x, v_uv_sample_bounds.y),  \
                 vec2_scalar(v_uv_sample_bounds.z, v_uv_sample_bounds.w));     \
      vec4 color = texture(sColor0, uv);                                       \
      float alpha = v_opacity_perspective_vec.x;                               \
      if (ALPHA_PASS) {                                                        \
        alpha *= 1.0f; /* antialias_brush() */                                 \
      }                                                                        \
      vec4 frag = alpha * color;                                               \
      if (ALPHA_PASS) {                                                        \
        frag *= 1.0f; /* do_clip() */                                          \
      }                                                                        \
      gl_FragColor = frag;                                                     \
    }                                                                          \
    void swgl_drawSpanRGBA8() {                                                \
      float perspective_divisor = mix(1.0f, 1.0f, v_opacity_perspective_vec.y); \
      vec2 uv = v_uv * perspective_divisor;                                    \
      swgl_commitTextureLinearColorRGBA8(sColor0, uv, v_uv_sample_bounds,      \
                                         v_opacity_perspective_vec.x);         \
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

WRSH_BRUSH_OPACITY(brush_opacity, "brush_opacity", false)
WRSH_BRUSH_OPACITY(brush_opacity_ALPHA_PASS, "brush_opacity ALPHA_PASS", true)
WRSH_BRUSH_OPACITY(brush_opacity_ANTIALIASING, "brush_opacity ANTIALIASING", false)
WRSH_BRUSH_OPACITY(brush_opacity_ALPHA_PASS_ANTIALIASING, "brush_opacity ALPHA_PASS,ANTIALIASING", true)
