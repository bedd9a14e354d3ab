// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "brush_mix_blend" and "brush_mix_blend ALPHA_PASS"
// (webrender_build/src/shader_features.rs:104-110). Restates
// webrender/res/brush_mix_blend.glsl:26-332 (get_uv :26-45, brush_vs :47-84, the blend
// functions :88-228, brush_fs :249-330) on brush_base.h, with swgl's glsl.h vector types.
// No swgl_drawSpan*: every pixel runs main() four at a time.  The scalar helper functions
// of the GLSL (ColorDodge, ColorBurn, SoftLight, ClipColor, SetSat) branch per fragment;
// here every lane goes through both sides and a select, which is what glsl-to-cxx emits
// for a divergent `if` (masks) -- the selected values are the same.

#define WRSH_BRUSH_MIX_BLEND(NAME, KEYSTR, ALPHA_PASS)                         \
  struct NAME##_vert : wrsh::brush_vert_base<NAME##_vert> {                    \
    typedef NAME##_vert Self;                                                  \
    static constexpr int VECS_PER_SPECIFIC_BRUSH = 3;                          \
    vec2 v_src_uv, v_backdrop_uv;                                              \
    vec4_scalar v_src_uv_sample_bounds, v_backdrop_uv_sample_bounds;           \
    vec2_scalar v_perspective;                                                 \
    ivec2_scalar v_op;                                                         \
    struct InterpOutputs {                                                     \
      vec2_scalar v_src_uv;                                                    \
      vec2_scalar v_backdrop_uv;                                               \
    };                                                                         \
    /* brush_mix_blend.glsl:26-45 */                                           \
    void get_uv(int res_address, vec2 f, ivec2_scalar texture_size, Float perspective_f, \
                vec2& out_uv, vec4_scalar& out_uv_sample_bounds) {             \
      using namespace wrsh;                                                    \
      vec4_scalar res0 = fetch_from_gpu_cache(res_address, 0);                 \
      vec2_scalar uv0 = vec2_scalar(res0.x, res0.y);                           \
      vec2_scalar uv1 = vec2_scalar(res0.z, res0.w);                           \
      vec2_scalar inv_texture_size =                                           \
          vec2_scalar(1.0f) / vec2_scalar(float(texture_size.x), float(texture_size.y)); \
      /* get_image_quad_uv, prim_shared.glsl:204-210 */                        \
      {                                                                        \
        vec4_scalar st_tl = fetch_from_gpu_cache(res_address + 2, 0);          \
        vec4_scalar st_tr = fetch_from_gpu_cache(res_address + 2, 1);          \
        vec4_scalar st_bl = fetch_from_gpu_cache(res_address + 2, 2);          \
        vec4_scalar st_br = fetch_from_gpu_cache(res_address + 2, 3);          \
        vec4 x = mix(vec4(st_tl), vec4(st_tr), f.x);                           \
        vec4 y = mix(vec4(st_bl), vec4(st_br), f.x);                           \
        vec4 z = mix(x, y, f.y);                                               \
        f = z.sel(X, Y) / z.w;                                                 \
      }                                                                        \
      vec2 uv = mix(uv0, uv1, f);                                              \
      out_uv = uv * inv_texture_size * perspective_f;                          \
      out_uv_sample_bounds =                                                   \
          vec4_scalar(uv0.x + 0.5f, uv0.y + 0.5f, uv1.x - 0.5f, uv1.y - 0.5f) * \
          vec4_scalar(inv_texture_size.x, inv_texture_size.y,                  \
                      inv_texture_size.x, inv_texture_size.y);                 \
    }                                                                          \
    /* brush_mix_blend.glsl:47-84 */                                           \
    void brush_vs(wrsh::BrushVertexInfo vi, int, wrsh::RectWithEndpoint local_rect, \
                  wrsh::RectWithEndpoint, ivec4_scalar prim_user_data, int,    \
                  mat4_scalar, wrsh::PictureTask, int brush_flags,             \
                  vec4_scalar) {                                               \
      using namespace wrsh;                                                    \
      vec2 f = (vi.local_pos - local_rect.p0) / rect_size(local_rect);         \
      float perspective_interpolate =                                          \
          (brush_flags & BRUSH_FLAG_PERSPECTIVE_INTERPOLATION) != 0 ? 1.0f : 0.0f; \
      Float perspective_f = mix(vi.world_pos.w, Float(1.0f), Float(perspective_interpolate)); \
      v_perspective.x = perspective_interpolate;                               \
      v_op.x = prim_user_data.x;                                               \
      get_uv(prim_user_data.y, f, textureSize(sColor0, 0), Float(1.0f),        \
             v_backdrop_uv, v_backdrop_uv_sample_bounds);                      \
      get_uv(prim_user_data.z, f, textureSize(sColor1, 0), perspective_f,      \
             v_src_uv, v_src_uv_sample_bounds);                                \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {   \
      for (int n = 0; n < 4; n++) {                                            \
        auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);               \
        dest->v_src_uv = get_nth(v_src_uv, n);                                 \
        dest->v_backdrop_uv = get_nth(v_backdrop_uv, n);                       \
        dest_ptr += stride;                                                    \
      }                                                                        \
    }                                                                          \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() {                                                            \
      used |= 1u << wrsh::U_sColor1;                                           \
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
      self->v_src_uv = init_interp(init->v_src_uv, step->v_src_uv);            \
      self->interp_step.v_src_uv = step->v_src_uv * 4.0f;                      \
      self->v_backdrop_uv = init_interp(init->v_backdrop_uv, step->v_backdrop_uv); \
      self->interp_step.v_backdrop_uv = step->v_backdrop_uv * 4.0f;            \
    }                                                                          \
    ALWAYS_INLINE void step_interp_inputs(int steps = 4) {                     \
      float chunks = steps * 0.25f;                                            \
      v_src_uv += interp_step.v_src_uv * chunks;                               \
      v_backdrop_uv += interp_step.v_backdrop_uv * chunks;                     \
    }                                                                          \
    /* :88-104 */                                                              \
    vec3 Multiply(vec3 Cb, vec3 Cs) { return Cb * Cs; }                        \
    vec3 Screen(vec3 Cb, vec3 Cs) { return Cb + Cs - (Cb * Cs); }              \
    vec3 HardLight(vec3 Cb, vec3 Cs) {                                         \
      vec3 m = Multiply(Cb, 2.0f * Cs);                                        \
      vec3 s = Screen(Cb, 2.0f * Cs - 1.0f);                                   \
      vec3 edge = vec3(Float(0.5f));                                           \
      return mix(m, s, step(edge, Cs));                                        \
    }                                                                          \
    /* :107-123 */                                                             \
    Float ColorDodge(Float Cb, Float Cs) {                                     \
      return if_then_else(Cb == 0.0f, Float(0.0f),                             \
                          if_then_else(Cs == 1.0f, Float(1.0f), min(Float(1.0f), Cb / (1.0f - Cs)))); \
    }                                                                          \
    Float ColorBurn(Float Cb, Float Cs) {                                      \
      return if_then_else(Cb == 1.0f, Float(1.0f),                             \
                          if_then_else(Cs == 0.0f, Float(0.0f), 1.0f - min(Float(1.0f), (1.0f - Cb) / Cs))); \
    }                                                                          \
    /* :125-139 */                                                             \
    Float SoftLight(Float Cb, Float Cs) {                                      \
      Float lo = Cb - (1.0f - 2.0f * Cs) * Cb * (1.0f - Cb);                   \
      Float D = if_then_else(Cb <= 0.25f, ((16.0f * Cb - 12.0f) * Cb + 4.0f) * Cb, sqrt(Cb)); \
      Float hi = Cb + (2.0f * Cs - 1.0f) * (D - Cb);                           \
      return if_then_else(Cs <= 0.5f, lo, hi);                                 \
    }                                                                          \
    vec3 Difference(vec3 Cb, vec3 Cs) { return abs(Cb - Cs); }                 \
    /* :148-175 */                                                             \
    Float Sat(vec3 c) { return max(c.x, max(c.y, c.z)) - min(c.x, min(c.y, c.z)); } \
    Float Lum(vec3 c) {                                                        \
      vec3 f = vec3(vec3_scalar(0.3f, 0.59f, 0.11f));                          \
      return dot(c, f);                                                        \
    }                                                                          \
    vec3 ClipColor(vec3 C) {                                                   \
      Float L = Lum(C);                                                        \
      Float n = min(C.x, min(C.y, C.z));                                       \
      Float x = max(C.x, max(C.y, C.z));                                       \
      C = if_then_else(n < 0.0f, L + (((C - L) * L) / (L - n)), C);            \
      C = if_then_else(x > 1.0f, L + (((C - L) * (1.0f - L)) / (x - L)), C);   \
      return C;                                                                \
    }                                                                          \
    vec3 SetLum(vec3 C, Float l) {                                             \
      Float d = l - Lum(C);                                                    \
      return ClipColor(C + d);                                                 \
    }                                                                          \
    /* :177-186: (Cmin, Cmid, Cmax) -> (0, mid', s) or zeros */                \
    void SetSatInner(Float& Cmin, Float& Cmid, Float& Cmax, Float s) {         \
      auto gt = Cmax > Cmin;                                                   \
      Cmid = if_then_else(gt, ((Cmid - Cmin) * s) / (Cmax - Cmin), Float(0.0f)); \
      Cmax = if_then_else(gt, s, Float(0.0f));                                 \
      Cmin = 0.0f;                                                             \
    }                                                                          \
    /* :188-214: the six orderings; every lane takes exactly one */            \
    vec3 SetSat(vec3 C, Float s) {                                             \
      vec3 out = C;                                                            \
      auto rg = C.x <= C.y, gb = C.y <= C.z, rb = C.x <= C.z;                  \
      { Float a = C.x, b = C.y, c = C.z; SetSatInner(a, b, c, s);              \
        out = if_then_else(rg & gb, vec3(a, b, c), out); }                     \
      { Float a = C.x, b = C.z, c = C.y; SetSatInner(a, b, c, s);              \
        out = if_then_else(rg & ~gb & rb, vec3(a, c, b), out); }               \
      { Float a = C.z, b = C.x, c = C.y; SetSatInner(a, b, c, s);              \
        out = if_then_else(rg & ~gb & ~rb, vec3(b, c, a), out); }              \
      { Float a = C.y, b = C.x, c = C.z; SetSatInner(a, b, c, s);              \
        out = if_then_else(~rg & rb, vec3(b, a, c), out); }                    \
      { Float a = C.y, b = C.z, c = C.x; SetSatInner(a, b, c, s);              \
        out = if_then_else(~rg & ~rb & gb, vec3(c, a, b), out); }              \
      { Float a = C.z, b = C.y, c = C.x; SetSatInner(a, b, c, s);              \
        out = if_then_else(~rg & ~rb & ~gb, vec3(c, b, a), out); }             \
      return out;                                                              \
    }                                                                          \
    vec3 Hue(vec3 Cb, vec3 Cs) { return SetLum(SetSat(Cs, Sat(Cb)), Lum(Cb)); } \
    vec3 Saturation(vec3 Cb, vec3 Cs) { return SetLum(SetSat(Cb, Sat(Cs)), Lum(Cb)); } \
    vec3 Color(vec3 Cb, vec3 Cs) { return SetLum(Cs, Lum(Cb)); }               \
    vec3 Luminosity(vec3 Cb, vec3 Cs) { return SetLum(Cb, Lum(Cs)); }          \
    /* brush_fs + main, :249-330 (2-D path: gl_FragCoord.w == 1) */            \
    void main() {                                                              \
      float perspective_divisor = mix(1.0f, 1.0f, v_perspective.x);            \
      vec2 src_uv = v_src_uv * perspective_divisor;                            \
      src_uv = clamp(src_uv, vec2_scalar(v_src_uv_sample_bounds.x, v_src_uv_sample_bounds.y), \
                     vec2_scalar(v_src_uv_sample_bounds.z, v_src_uv_sample_bounds.w)); \
      vec2 backdrop_uv = clamp(v_backdrop_uv,                                  \
                               vec2_scalar(v_backdrop_uv_sample_bounds.x, v_backdrop_uv_sample_bounds.y), \
                               vec2_scalar(v_backdrop_uv_sample_bounds.z, v_backdrop_uv_sample_bounds.w)); \
      vec4 Cb4 = texture(sColor0, backdrop_uv);                                \
      vec4 Cs4 = texture(sColor1, src_uv);                                     \
      vec3 Cb = if_then_else(Cb4.w != 0.0f, Cb4.sel(X, Y, Z) / Cb4.w, Cb4.sel(X, Y, Z)); \
      vec3 Cs = if_then_else(Cs4.w != 0.0f, Cs4.sel(X, Y, Z) / Cs4.w, Cs4.sel(X, Y, Z)); \
      vec3 rgb = vec3(vec3_scalar(1.0f, 1.0f, 0.0f));                          \
      switch (v_op.x & 0xFF) {                                                 \
        case 1: rgb = Multiply(Cb, Cs); break;                                 \
        case 3: rgb = HardLight(Cs, Cb); break;                                \
        case 4: rgb = min(Cs, Cb); break;                                      \
        case 5: rgb = max(Cs, Cb); break;                                      \
        case 6: rgb = vec3(ColorDodge(Cb.x, Cs.x), ColorDodge(Cb.y, Cs.y), ColorDodge(Cb.z, Cs.z)); break; \
        case 7: rgb = vec3(ColorBurn(Cb.x, Cs.x), ColorBurn(Cb.y, Cs.y), ColorBurn(Cb.z, Cs.z)); break; \
        case 8: rgb = HardLight(Cb, Cs); break;                                \
        case 9: rgb = vec3(SoftLight(Cb.x, Cs.x), SoftLight(Cb.y, Cs.y), SoftLight(Cb.z, Cs.z)); break; \
        case 10: rgb = Difference(Cb, Cs); break;                              \
        case 12: rgb = Hue(Cb, Cs); break;                                     \
        case 13: rgb = Saturation(Cb, Cs); break;                              \
        case 14: rgb = Color(Cb, Cs); break;                                   \
        case 15: rgb = Luminosity(Cb, Cs); break;                              \
        default: break;                                                        \
      }                                                                        \
      rgb = (1.0f - Cb4.w) * Cs + Cb4.w * rgb;                                 \
      Float a = Cs4.w;                                                         \
      rgb = rgb * a;                                                            \
      vec4 frag = vec4(rgb, a);                                                \
      if (ALPHA_PASS) {                                                        \
        frag *= 1.0f; /* antialias_brush() */                                  \
      }                                                                        \
      frag *= 1.0f; /* brush.glsl main(): do_clip() under SWGL_CLIP_MASK */    \
      gl_FragColor = frag;                                                     \
    }                                                                          \
    WRSH_FRAG_ABI(Self)                                                        \
    NAME##_frag() { WRSH_FRAG_WIRING() }                                       \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_BRUSH_MIX_BLEND(brush_mix_blend, "brush_mix_blend", false)
WRSH_BRUSH_MIX_BLEND(brush_mix_blend_ALPHA_PASS, "brush_mix_blend ALPHA_PASS", true)
