// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "cs_blur ALPHA_TARGET" and "cs_blur COLOR_TARGET"
// (webrender_build/src/shader_features.rs: cache shaders). Restates
// webrender/res/cs_blur.glsl:47-194 with SWGL defined; the span path is the
// reference's own swgl_commitGaussianBlur* (swgl_ext.h:947-996).

#define WRSH_CS_BLUR(NAME, KEYSTR, COLOR_TARGET)                               \
  struct NAME##_vert : VertexShaderImpl, wrsh::CommonState {                   \
    typedef NAME##_vert Self;                                                  \
    wrsh::AttribTable attribs;                                                 \
    int a_aPosition, a_task, a_src, a_dir, a_params;                           \
    vec2 aPosition;                                                            \
    int aBlurRenderTaskAddress, aBlurSourceTaskAddress, aBlurDirection;        \
    vec3_scalar aBlurParams;                                                   \
    vec2 vUv;                                                                  \
    vec4_scalar vUvRect;                                                       \
    vec2_scalar vOffsetScale;                                                  \
    ivec2_scalar vSupport;                                                     \
    vec2_scalar vGaussCoefficients;                                            \
    struct InterpOutputs {                                                     \
      vec2_scalar vUv;                                                         \
    };                                                                         \
    wrsh::RectWithEndpoint fetch_render_task_rect(int index) const {           \
      ivec2_scalar uv = wrsh::get_fetch_uv(index, 2u);                         \
      vec4_scalar texel0 = texelFetch(sRenderTasks, uv, 0);                    \
      return wrsh::RectWithEndpoint{vec2_scalar(texel0.x, texel0.y),           \
                                    vec2_scalar(texel0.z, texel0.w)};          \
    }                                                                          \
    /* cs_blur.glsl:47-70 */                                                   \
    void calculate_gauss_coefficients(float sigma) {                           \
      vGaussCoefficients =                                                     \
          vec2_scalar(1.0f / (sqrt(2.0f * 3.14159265f) * sigma),               \
                      exp(-0.5f / (sigma * sigma)));                           \
      vec3_scalar gauss_coefficient =                                          \
          vec3_scalar(vGaussCoefficients.x, vGaussCoefficients.y,              \
                      vGaussCoefficients.y * vGaussCoefficients.y);            \
      float gauss_coefficient_total = gauss_coefficient.x;                     \
      for (int i = 1; i <= vSupport.x; i += 2) {                               \
        gauss_coefficient.x *= gauss_coefficient.y;                            \
        gauss_coefficient.y *= gauss_coefficient.z;                            \
        float gauss_coefficient_subtotal = gauss_coefficient.x;                \
        gauss_coefficient.x *= gauss_coefficient.y;                            \
        gauss_coefficient.y *= gauss_coefficient.z;                            \
        gauss_coefficient_subtotal += gauss_coefficient.x;                     \
        gauss_coefficient_total += 2.0f * gauss_coefficient_subtotal;          \
      }                                                                        \
      vGaussCoefficients.x /= gauss_coefficient_total;                         \
    }                                                                          \
    /* cs_blur.glsl:72-121 */                                                  \
    void main() {                                                              \
      using namespace wrsh;                                                    \
      RectWithEndpoint task_rect =                                             \
          fetch_render_task_rect(aBlurRenderTaskAddress);                      \
      float blur_radius = aBlurParams.x;                                       \
      vec2_scalar blur_region = vec2_scalar(aBlurParams.y, aBlurParams.z);     \
      RectWithEndpoint src_rect =                                              \
          fetch_render_task_rect(aBlurSourceTaskAddress);                      \
      RectWithEndpoint target_rect = task_rect;                                \
      ivec2_scalar ts = textureSize(sColor0, 0);                               \
      vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));        \
      vSupport.x = int(ceil(1.5f * blur_radius)) * 2;                          \
      if (vSupport.x > 0) {                                                    \
        calculate_gauss_coefficients(blur_radius);                             \
      } else {                                                                 \
        vGaussCoefficients = vec2_scalar(1.0f, 1.0f);                          \
      }                                                                        \
      switch (aBlurDirection) {                                                \
        case 0:                                                                \
          vOffsetScale = vec2_scalar(1.0f / texture_size.x, 0.0f);             \
          break;                                                               \
        case 1:                                                                \
          vOffsetScale = vec2_scalar(0.0f, 1.0f / texture_size.y);             \
          break;                                                               \
        default:                                                               \
          vOffsetScale = vec2_scalar(0.0f);                                    \
      }                                                                        \
      vec2_scalar r0 = src_rect.p0 + vec2_scalar(0.5f);                        \
      vec2_scalar r1 = src_rect.p0 + blur_region - vec2_scalar(0.5f);          \
      vUvRect = vec4_scalar(r0.x, r0.y, r1.x, r1.y);                           \
      vUvRect /= vec4_scalar(texture_size.x, texture_size.y, texture_size.x,   \
                             texture_size.y);                                  \
      vec2 pos = mix(target_rect.p0, target_rect.p1, aPosition);               \
      vec2_scalar uv0 = src_rect.p0 / texture_size;                            \
      vec2_scalar uv1 = src_rect.p1 / texture_size;                            \
      vUv = mix(uv0, uv1, aPosition);                                          \
      gl_Position = uTransform * vec4(pos, 0.0f, 1.0f);                        \
    }                                                                          \
    static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,    \
                             uint32_t start, int instance, int count) {        \
      Self* self = (Self*)impl;                                                \
      auto& t = self->attribs;                                                 \
      load_attrib(self->aPosition, attribs[t.locs[self->a_aPosition]], start,  \
                  instance, count);                                            \
      load_flat_attrib(self->aBlurRenderTaskAddress,                           \
                       attribs[t.locs[self->a_task]], start, instance, count); \
      load_flat_attrib(self->aBlurSourceTaskAddress,                           \
                       attribs[t.locs[self->a_src]], start, instance, count);  \
      load_flat_attrib(self->aBlurDirection, attribs[t.locs[self->a_dir]],     \
                       start, instance, count);                                \
      load_flat_attrib(self->aBlurParams, attribs[t.locs[self->a_params]],     \
                       start, instance, count);                                \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {   \
      for (int n = 0; n < 4; n++) {                                            \
        auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);               \
        dest->vUv = get_nth(vUv, n);                                           \
        dest_ptr += stride;                                                    \
      }                                                                        \
    }                                                                          \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() {                                                            \
      using namespace wrsh;                                                    \
      used = (1u << U_sColor0) | (1u << U_sRenderTasks) |                      \
             (1u << U_sGpuCache) | (1u << U_sTransformPalette) |               \
             (1u << U_sPrimitiveHeadersF) | (1u << U_sPrimitiveHeadersI) |     \
             (1u << U_sClipMask) | (1u << U_uTransform);                       \
      a_aPosition = attribs.add("aPosition");                                  \
      a_task = attribs.add("aBlurRenderTaskAddress");                          \
      a_src = attribs.add("aBlurSourceTaskAddress");                           \
      a_dir = attribs.add("aBlurDirection");                                   \
      a_params = attribs.add("aBlurParams");                                   \
      vSupport = ivec2_scalar(0, 0);                                           \
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
      self->vUv = init_interp(init->vUv, step->vUv);                           \
      self->interp_step.vUv = step->vUv * 4.0f;                                \
    }                                                                          \
    ALWAYS_INLINE void step_interp_inputs(int steps = 4) {                     \
      float chunks = steps * 0.25f;                                            \
      vUv += interp_step.vUv * chunks;                                         \
    }                                                                          \
    /* cs_blur.glsl:137-181; SAMPLE_TYPE is vec4 (COLOR) or float (ALPHA) */   \
    void main() {                                                              \
      vec3_scalar gauss_coefficient =                                          \
          vec3_scalar(vGaussCoefficients.x, vGaussCoefficients.y,              \
                      vGaussCoefficients.y * vGaussCoefficients.y);            \
      int support = min(vSupport.x, 300);                                      \
      if (COLOR_TARGET) {                                                      \
        vec4 original_color = texture(sColor0, vUv);                           \
        vec4 avg_color = original_color * gauss_coefficient.x;                 \
        for (int i = 1; i <= support; i += 2) {                                \
          gauss_coefficient.x *= gauss_coefficient.y;                          \
          gauss_coefficient.y *= gauss_coefficient.z;                          \
          float gauss_coefficient_subtotal = gauss_coefficient.x;              \
          gauss_coefficient.x *= gauss_coefficient.y;                          \
          gauss_coefficient.y *= gauss_coefficient.z;                          \
          gauss_coefficient_subtotal += gauss_coefficient.x;                   \
          float gauss_ratio = gauss_coefficient.x / gauss_coefficient_subtotal; \
          vec2_scalar offset = vOffsetScale * (float(i) + gauss_ratio);        \
          vec2 st0 = max(vUv - offset, vec2_scalar(vUvRect.x, vUvRect.y));     \
          vec2 st1 = min(vUv + offset, vec2_scalar(vUvRect.z, vUvRect.w));     \
          avg_color += (texture(sColor0, st0) + texture(sColor0, st1)) *       \
                       gauss_coefficient_subtotal;                             \
        }                                                                      \
        gl_FragColor = avg_color;                                              \
      } else {                                                                 \
        Float original_color = texture(sColor0, vUv).x;                        \
        Float avg_color = original_color * gauss_coefficient.x;                \
        for (int i = 1; i <= support; i += 2) {                                \
          gauss_coefficient.x *= gauss_coefficient.y;                          \
          gauss_coefficient.y *= gauss_coefficient.z;                          \
          float gauss_coefficient_subtotal = gauss_coefficient.x;              \
          gauss_coefficient.x *= gauss_coefficient.y;                          \
          gauss_coefficient.y *= gauss_coefficient.z;                          \
          gauss_coefficient_subtotal += gauss_coefficient.x;                   \
          float gauss_ratio = gauss_coefficient.x / gauss_coefficient_subtotal; \
          vec2_scalar offset = vOffsetScale * (float(i) + gauss_ratio);        \
          vec2 st0 = max(vUv - offset, vec2_scalar(vUvRect.x, vUvRect.y));     \
          vec2 st1 = min(vUv + offset, vec2_scalar(vUvRect.z, vUvRect.w));     \
          avg_color += (texture(sColor0, st0).x + texture(sColor0, st1).x) *   \
                       gauss_coefficient_subtotal;                             \
        }                                                                      \
        gl_FragColor = vec4(avg_color);                                        \
      }                                                                        \
    }                                                                          \
    void swgl_drawSpanRGBA8() {                                                \
      if (COLOR_TARGET)                                                        \
        swgl_commitGaussianBlurRGBA8(sColor0, vUv, vUvRect,                    \
                                     vOffsetScale.x != 0.0f, vSupport.x,       \
                                     vGaussCoefficients);                      \
    }                                                                          \
    void swgl_drawSpanR8() {                                                   \
      if (!(COLOR_TARGET))                                                     \
        swgl_commitGaussianBlurR8(sColor0, vUv, vUvRect,                       \
                                  vOffsetScale.x != 0.0f, vSupport.x,          \
                                  vGaussCoefficients);                         \
    }                                                                          \
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
      if (COLOR_TARGET)                                                        \
        draw_span_RGBA8_func = &draw_span_RGBA8;                               \
      else                                                                     \
        draw_span_R8_func = &draw_span_R8;                                     \
    }                                                                          \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_CS_BLUR(cs_blur_ALPHA_TARGET, "cs_blur ALPHA_TARGET", false)
WRSH_CS_BLUR(cs_blur_COLOR_TARGET, "cs_blur COLOR_TARGET", true)
