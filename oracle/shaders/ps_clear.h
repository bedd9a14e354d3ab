// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "ps_clear" (shader_features.rs:232). Restates
// webrender/res/ps_clear.glsl:9-25. vColor is only ever assigned a flat
// attribute, so (as the emitter's run-class inference would) it is kept scalar.

struct ps_clear_vert : VertexShaderImpl, wrsh::CommonState {
  typedef ps_clear_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aRect, a_aColor;
  vec2 aPosition;
  vec4_scalar aRect, aColor;
  vec4_scalar vColor;
  struct InterpOutputs {};

  void main() {
    vec2 pos = mix(aRect.sel(X, Y), aRect.sel(Z, W), aPosition);
    gl_Position = uTransform * vec4(pos, 0.0f, 1.0f);
    gl_Position.z = gl_Position.w;
    vColor = aColor;
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start, instance,
                count);
    load_flat_attrib(self->aRect, attribs[L[self->a_aRect]], start, instance,
                     count);
    load_flat_attrib(self->aColor, attribs[L[self->a_aColor]], start, instance,
                     count);
  }
  ALWAYS_INLINE void store_interp_outputs(char*, size_t) {}
  WRSH_VERT_ABI(Self)
  ps_clear_vert() {
    using namespace wrsh;
    used = (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_aRect = attribs.add("aRect");
    a_aColor = attribs.add("aColor");
    WRSH_VERT_WIRING(Self)
  }
};

struct ps_clear_frag : FragmentShaderImpl, ps_clear_vert {
  typedef ps_clear_frag Self;
  static void read_interp_inputs(FragmentShaderImpl*, const void*,
                                 const void*) {}
  ALWAYS_INLINE void step_interp_inputs(int = 4) {}
  void main() { gl_FragColor = vec4(vColor); }
  WRSH_FRAG_ABI(Self)
  ps_clear_frag() { WRSH_FRAG_WIRING() }
};

WRSH_PROGRAM(ps_clear, "ps_clear")
