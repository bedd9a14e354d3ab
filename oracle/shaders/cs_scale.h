// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_scale TEXTURE_2D" (webrender_build/src/shader_features.rs:182).
// Restates webrender/res/cs_scale.glsl:24-66 with SWGL defined.  Only an RGBA8
// span function exists (:61-65): on R8 targets every pixel runs main().

struct cs_scale_TEXTURE_2D_vert : VertexShaderImpl, wrsh::CommonState {
  typedef cs_scale_TEXTURE_2D_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_target, a_source, a_type;
  vec2 aPosition;
  vec4_scalar aScaleTargetRect, aScaleSourceRect;
  float aSourceRectType;
  vec2 vUv;
  vec4_scalar vUvRect;
  struct InterpOutputs {
    vec2_scalar vUv;
  };
  void main() {
    vec2_scalar src_offset = aScaleSourceRect.sel(X, Y);
    vec2_scalar src_size = aScaleSourceRect.sel(Z, W) - aScaleSourceRect.sel(X, Y);
    vec2_scalar bmin = min(aScaleSourceRect.sel(X, Y), aScaleSourceRect.sel(Z, W));
    vec2_scalar bmax = max(aScaleSourceRect.sel(X, Y), aScaleSourceRect.sel(Z, W));
    vUvRect = vec4_scalar(bmin.x, bmin.y, bmax.x, bmax.y);
    vUv = (src_offset + src_size * aPosition);
    if (int(aSourceRectType) == 1 /* UV_TYPE_UNNORMALIZED */) {
      vUvRect = vec4_scalar(vUvRect.x + 0.5f, vUvRect.y + 0.5f, vUvRect.z - 0.5f,
                            vUvRect.w - 0.5f);
      ivec2_scalar ts = textureSize(sColor0, 0);
      vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));
      vUvRect /= vec4_scalar(texture_size.x, texture_size.y, texture_size.x,
                             texture_size.y);
      vUv /= texture_size;
    }
    vec2 pos = mix(aScaleTargetRect.sel(X, Y), aScaleTargetRect.sel(Z, W), aPosition);
    gl_Position = uTransform * vec4(pos, 0.0f, 1.0f);
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start, instance, count);
    load_flat_attrib(self->aScaleTargetRect, attribs[L[self->a_target]], start, instance, count);
    load_flat_attrib(self->aScaleSourceRect, attribs[L[self->a_source]], start, instance, count);
    load_flat_attrib(self->aSourceRectType, attribs[L[self->a_type]], start, instance, count);
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->vUv = get_nth(vUv, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  cs_scale_TEXTURE_2D_vert() {
    using namespace wrsh;
    used = (1u << U_sColor0) | (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_target = attribs.add("aScaleTargetRect");
    a_source = attribs.add("aScaleSourceRect");
    a_type = attribs.add("aSourceRectType");
    WRSH_VERT_WIRING(Self)
  }
};

struct cs_scale_TEXTURE_2D_frag : FragmentShaderImpl, cs_scale_TEXTURE_2D_vert {
  typedef cs_scale_TEXTURE_2D_frag Self;
  typedef cs_scale_TEXTURE_2D_vert::InterpOutputs InterpInputs;
  InterpInputs interp_step;
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_,
                                 const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    self->vUv = init_interp(init->vUv, step->vUv);
    self->interp_step.vUv = step->vUv * 4.0f;
  }
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) {
    float chunks = steps * 0.25f;
    vUv += interp_step.vUv * chunks;
  }
  void main() {
    vec2 st = clamp(vUv, vec2_scalar(vUvRect.x, vUvRect.y), vec2_scalar(vUvRect.z, vUvRect.w));
    gl_FragColor = texture(sColor0, st);
  }
  void swgl_drawSpanRGBA8() { swgl_commitTextureLinearRGBA8(sColor0, vUv, vUvRect); }
  WRSH_FRAG_ABI(Self)
  static int draw_span_RGBA8(FragmentShaderImpl* impl) {
    Self* self = (Self*)impl;
    DISPATCH_DRAW_SPAN(self, RGBA8);
  }
  cs_scale_TEXTURE_2D_frag() {
    WRSH_FRAG_WIRING()
    draw_span_RGBA8_func = &draw_span_RGBA8;
  }
};

WRSH_PROGRAM(cs_scale_TEXTURE_2D, "cs_scale TEXTURE_2D")
