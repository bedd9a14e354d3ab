// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated
// headers of shader keys "composite TEXTURE_2D" and
// "composite FAST_PATH,TEXTURE_2D" (shader_features.rs:170-204). Restates
// webrender/res/composite.glsl:73-159 (VS), 163-193 (FS), 195-234 (span).

#define WRSH_COMPOSITE(NAME, KEYSTR, FAST_PATH)                                \
  struct NAME##_vert : VertexShaderImpl, wrsh::CommonState {                   \
    typedef NAME##_vert Self;                                                  \
    wrsh::AttribTable attribs;                                                 \
    int a_aPosition, a_aDeviceRect, a_aDeviceClipRect, a_aColor, a_aParams,    \
        a_aFlip, a_aUvRect0;                                                   \
    vec2 aPosition;                                                            \
    vec4_scalar aDeviceRect, aDeviceClipRect, aColor, aParams, aUvRect0;       \
    vec2_scalar aFlip;                                                         \
    vec2 vUv;                                                                  \
    vec4_scalar vColor, vUVBounds;                                             \
    struct InterpOutputs {                                                     \
      vec2_scalar vUv;                                                         \
    };                                                                         \
    void main() {                                                              \
      vec4_scalar device_rect =                                                \
          mix(aDeviceRect, aDeviceRect.sel(Z, W, X, Y),                        \
              vec4_scalar(aFlip.x, aFlip.y, aFlip.x, aFlip.y));                \
      vec2 world_pos = mix(device_rect.sel(X, Y), device_rect.sel(Z, W),       \
                           aPosition);                                         \
      vec2 clipped_world_pos =                                                 \
          clamp(world_pos, vec2(aDeviceClipRect.sel(X, Y)),                    \
                vec2(aDeviceClipRect.sel(Z, W)));                              \
      vec2 uv = (clipped_world_pos - device_rect.sel(X, Y)) /                  \
                (device_rect.sel(Z, W) - device_rect.sel(X, Y));               \
      uv = mix(aUvRect0.sel(X, Y), aUvRect0.sel(Z, W), uv);                    \
      vec2_scalar bmin = min(aUvRect0.sel(X, Y), aUvRect0.sel(Z, W));          \
      vec2_scalar bmax = max(aUvRect0.sel(X, Y), aUvRect0.sel(Z, W));          \
      vec4_scalar uvBounds = vec4_scalar(bmin.x, bmin.y, bmax.x, bmax.y);      \
      if (int(aParams.y) == 1 /* UV_TYPE_UNNORMALIZED */) {                    \
        ivec2_scalar ts = textureSize(sColor0, 0);                             \
        vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));      \
        uvBounds = uvBounds + vec4_scalar(0.5f, 0.5f, -0.5f, -0.5f);           \
        uv = uv / texture_size;                                                \
        uvBounds = uvBounds / vec4_scalar(texture_size.x, texture_size.y,      \
                                          texture_size.x, texture_size.y);     \
      }                                                                        \
      vUv = uv;                                                                \
      if (!(FAST_PATH)) {                                                      \
        vUVBounds = uvBounds;                                                  \
        vColor = aColor;                                                       \
      }                                                                        \
      gl_Position = uTransform * vec4(clipped_world_pos, 0.0f, 1.0f);          \
    }                                                                          \
    static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,    \
                             uint32_t start, int instance, int count) {        \
      Self* self = (Self*)impl;                                                \
      auto& L = self->attribs.locs;                                            \
      load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start,       \
                  instance, count);                                            \
      load_flat_attrib(self->aDeviceRect, attribs[L[self->a_aDeviceRect]],     \
                       start, instance, count);                                \
      load_flat_attrib(self->aDeviceClipRect,                                  \
                       attribs[L[self->a_aDeviceClipRect]], start, instance,   \
                       count);                                                 \
      load_flat_attrib(self->aColor, attribs[L[self->a_aColor]], start,        \
                       instance, count);                                       \
      load_flat_attrib(self->aParams, attribs[L[self->a_aParams]], start,      \
                       instance, count);                                       \
      load_flat_attrib(self->aFlip, attribs[L[self->a_aFlip]], start,          \
                       instance, count);                                       \
      load_flat_attrib(self->aUvRect0, attribs[L[self->a_aUvRect0]], start,    \
                       instance, count);                                       \
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
      used = (1u << U_sColor0) | (1u << U_sColor1) | (1u << U_sColor2) |       \
             (1u << U_uTransform);                                             \
      a_aPosition = attribs.add("aPosition");                                  \
      a_aDeviceRect = attribs.add("aDeviceRect");                              \
      a_aDeviceClipRect = attribs.add("aDeviceClipRect");                      \
      a_aColor = attribs.add("aColor");                                        \
      a_aParams = attribs.add("aParams");                                      \
      a_aFlip = attribs.add("aFlip");                                          \
      a_aUvRect0 = attribs.add("aUvRect0");                                    \
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
    void main() {                                                              \
      vec4 color;                                                              \
      if (FAST_PATH) {                                                         \
        vec2 uv = vUv;                                                         \
        vec4 texel = texture(sColor0, uv);                                     \
        color = texel;                                                         \
      } else {                                                                 \
        vec2 uv = clamp(vUv, vec2_scalar(vUVBounds.x, vUVBounds.y),            \
                        vec2_scalar(vUVBounds.z, vUVBounds.w));                \
        vec4 texel = texture(sColor0, uv);                                     \
        color = vColor * texel;                                                \
      }                                                                        \
      gl_FragColor = color;                                                    \
    }                                                                          \
    void swgl_drawSpanRGBA8() {                                                \
      vec4_scalar color, uvBounds;                                             \
      if (FAST_PATH) {                                                         \
        color = vec4_scalar(1.0f);                                             \
        uvBounds = vec4_scalar(0.0f, 0.0f, 1.0f, 1.0f);                        \
      } else {                                                                 \
        color = vColor;                                                        \
        uvBounds = vUVBounds;                                                  \
      }                                                                        \
      if (color != vec4_scalar(1.0f)) {                                        \
        swgl_commitTextureColorRGBA8(sColor0, vUv, uvBounds, color);           \
      } else {                                                                 \
        swgl_commitTextureRGBA8(sColor0, vUv, uvBounds);                       \
      }                                                                        \
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

WRSH_COMPOSITE(composite_TEXTURE_2D, "composite TEXTURE_2D", false)
WRSH_COMPOSITE(composite_FAST_PATH_TEXTURE_2D, "composite FAST_PATH,TEXTURE_2D",
               true)
