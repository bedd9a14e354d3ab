// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "brush_linear_gradient" and "brush_linear_gradient ALPHA_PASS"
// (webrender_build/src/shader_features.rs:111-121, no DITHERING). Restates
// webrender/res/brush_linear_gradient.glsl:31-93, gradient_shared.glsl:19-75
// (write_gradient_vertex, compute_repeated_pos under SWGL_ANTIALIAS) and
// gradient.glsl:30-61 (sample_gradient) on brush_base.h. The span shader hands the
// row to the reference's own swgl_commitLinearGradientRGBA8 (swgl_ext.h:1390-1607).

#define WRSH_BRUSH_LINEAR_GRADIENT(NAME, KEYSTR, ALPHA_PASS)                   \
  struct NAME##_vert : wrsh::brush_vert_base<NAME##_vert> {                    \
    typedef NAME##_vert Self;                                                  \
    static constexpr int VECS_PER_SPECIFIC_BRUSH = 2;                          \
    vec2 v_pos;                                                                \
    vec2_scalar v_repeated_size, v_tile_repeat;                                \
    ivec2_scalar v_gradient_address;                                           \
    vec2_scalar v_gradient_repeat;                                             \
    vec2_scalar v_start_offset, v_scale_dir;                                   \
    struct InterpOutputs {                                                     \
      vec2_scalar v_pos;                                                       \
    };                                                                         \
    void brush_vs(wrsh::BrushVertexInfo vi, int prim_address,                  \
                  wrsh::RectWithEndpoint local_rect,                           \
                  wrsh::RectWithEndpoint segment_rect,                         \
                  ivec4_scalar prim_user_data, int, mat4_scalar,               \
                  wrsh::PictureTask, int brush_flags,                          \
                  vec4_scalar texel_rect) {                                    \
      using namespace wrsh;                                                    \
      /* fetch_gradient, brush_linear_gradient.glsl:21-28 */                   \
      vec4_scalar start_end_point = fetch_from_gpu_cache(prim_address, 0);     \
      vec4_scalar data1 = fetch_from_gpu_cache(prim_address, 1);               \
      int extend_mode = int(data1.x);                                          \
      vec2_scalar stretch_size = vec2_scalar(data1.y, data1.z);                \
      /* write_gradient_vertex, gradient_shared.glsl:19-52 */                  \
      if ((brush_flags & BRUSH_FLAG_SEGMENT_RELATIVE) != 0) {                  \
        v_pos = (vi.local_pos - segment_rect.p0) / rect_size(segment_rect);    \
        v_pos = v_pos * (vec2_scalar(texel_rect.z, texel_rect.w) -             \
                         vec2_scalar(texel_rect.x, texel_rect.y)) +            \
                vec2_scalar(texel_rect.x, texel_rect.y);                       \
        v_pos = v_pos * rect_size(local_rect);                                 \
      } else {                                                                 \
        v_pos = vi.local_pos - local_rect.p0;                                  \
      }                                                                        \
      vec2_scalar tile_repeat = rect_size(local_rect) / stretch_size;          \
      v_repeated_size = stretch_size;                                          \
      v_pos /= v_repeated_size;                                                \
      v_gradient_address.x = prim_user_data.x;                                 \
      v_gradient_repeat.x = float(extend_mode == 1 /* EXTEND_MODE_REPEAT */);  \
      if (ALPHA_PASS) {                                                        \
        v_tile_repeat = tile_repeat;                                           \
      }                                                                        \
      /* brush_vs, brush_linear_gradient.glsl:55-63 */                         \
      vec2_scalar start_point =                                                \
          vec2_scalar(start_end_point.x, start_end_point.y);                   \
      vec2_scalar end_point =                                                  \
          vec2_scalar(start_end_point.z, start_end_point.w);                   \
      vec2_scalar dir = end_point - start_point;                               \
      v_scale_dir = dir / dot(dir, dir);                                       \
      v_start_offset.x = dot(start_point, v_scale_dir);                        \
      v_scale_dir *= v_repeated_size;                                          \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {   \
      for (int n = 0; n < 4; n++) {                                            \
        auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);               \
        dest->v_pos = get_nth(v_pos, n);                                       \
        dest_ptr += stride;                                                    \
      }                                                                        \
    }                                                                          \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() {                                                            \
      used |= 1u << wrsh::U_sGpuBufferF;                                       \
      WRSH_VERT_WIRING(Self)                                                   \
    }                                                                          \
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
      self->v_pos = init_interp(init->v_pos, step->v_pos);                     \
      self->interp_step.v_pos = step->v_pos * 4.0f;                            \
    }                                                                          \
    ALWAYS_INLINE void step_interp_inputs(int steps = 4) {                     \
      float chunks = steps * 0.25f;                                            \
      v_pos += interp_step.v_pos * chunks;                                     \
    }                                                                          \
    /* gradient.glsl:30-61 */                                                  \
    vec4 sample_gradient(Float offset) const {                                 \
      offset -= floor(offset) * v_gradient_repeat.x;                           \
      Float x = clamp(1.0f + offset * 128.0f, 0.0f, 1.0f + 128.0f);            \
      Float entry_index = floor(x);                                            \
      Float entry_fract = x - entry_index;                                     \
      I32 address = v_gradient_address.x + 2 * cast(entry_index);              \
      vec4 t0, t1;                                                             \
      for (int n = 0; n < 4; n++) {                                            \
        ivec2_scalar uv = wrsh::get_gpu_uv(address[n]);                        \
        put_nth(t0, n, texelFetch(sGpuBufferF, uv, 0));                        \
        put_nth(t1, n, texelFetch(sGpuBufferF, ivec2_scalar(uv.x + 1, uv.y), 0)); \
      }                                                                        \
      return t0 + t1 * entry_fract;                                            \
    }                                                                          \
    /* brush_fs, brush_linear_gradient.glsl:66-83 + brush.glsl main() (SWGL:   \
       antialias_brush() == 1, do_clip() == 1) */                              \
    /* the perspective entry points glsl-to-cxx emits for a program with a */  \
    /* varying (lib.rs:660-690, 716-741, 3576-3590) */                         \
    struct InterpPerspective {                                                 \
      vec2 v_pos;                                                                \
    };                                                                         \
    InterpPerspective interp_perspective;                                      \
    static void read_perspective_inputs(FragmentShaderImpl* impl,              \
                                        const void* init_, const void* step_) { \
      Self* self = (Self*)impl;                                                \
      const InterpInputs* init = (const InterpInputs*)init_;                   \
      const InterpInputs* step = (const InterpInputs*)step_;                   \
      Float w = 1.0f / self->gl_FragCoord.w;                                   \
      self->interp_perspective.v_pos = init_interp(init->v_pos, step->v_pos);        \
      self->v_pos = self->interp_perspective.v_pos * w;                            \
      self->interp_step.v_pos = step->v_pos * 4.0f;                                \
    }                                                                          \
    ALWAYS_INLINE void step_perspective_inputs(int steps = 4) {                \
      step_perspective(steps);                                                 \
      float chunks = steps * 0.25f;                                            \
      Float w = 1.0f / gl_FragCoord.w;                                         \
      interp_perspective.v_pos += interp_step.v_pos * chunks;                      \
      v_pos = w * interp_perspective.v_pos;                                        \
    }                                                                          \
    static void run_perspective(FragmentShaderImpl* impl) {                    \
      Self* self = (Self*)impl;                                                \
      self->main();                                                              \
      self->step_perspective_inputs();                                         \
    }                                                                          \
    static void skip_perspective(FragmentShaderImpl* impl, int steps) {        \
      Self* self = (Self*)impl;                                                \
      self->step_perspective_inputs(steps);                                    \
    }                                                                          \
    void main() {                                                              \
      vec2 pos = fract(v_pos);                                                 \
      Float offset = dot(pos, vec2(v_scale_dir)) - v_start_offset.x;           \
      vec4 color = sample_gradient(offset);                                    \
      if (ALPHA_PASS) {                                                        \
        color *= 1.0f;                                                         \
        color *= 1.0f;                                                         \
      }                                                                        \
      gl_FragColor = color;                                                    \
    }                                                                          \
    void swgl_drawSpanRGBA8() {                                                \
      int address = swgl_validateGradient(                                     \
          sGpuBufferF, wrsh::get_gpu_uv(v_gradient_address.x), int(128.0f + 2.0f)); \
      if (address < 0) {                                                       \
        return;                                                                \
      }                                                                        \
      swgl_commitLinearGradientRGBA8(sGpuBufferF, address, 128.0f, true,       \
                                     v_gradient_repeat.x != 0.0f, v_pos,       \
                                     v_scale_dir, v_start_offset.x);           \
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

WRSH_BRUSH_LINEAR_GRADIENT(brush_linear_gradient, "brush_linear_gradient", false)
WRSH_BRUSH_LINEAR_GRADIENT(brush_linear_gradient_ALPHA_PASS, "brush_linear_gradient ALPHA_PASS", true)
