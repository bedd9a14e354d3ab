// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_fast_linear_gradient" (webrender_build/src/shader_features.rs).
// Restates webrender/res/cs_fast_linear_gradient.glsl:7-32 with SWGL defined.
// The program has no span function: every pixel runs main().

struct cs_fast_linear_gradient_vert : VertexShaderImpl, wrsh::CommonState {
  typedef cs_fast_linear_gradient_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aTaskRect, a_aColor0, a_aColor1, a_aAxisSelect;
  vec2 aPosition;
  vec4_scalar aTaskRect, aColor0, aColor1;
  float aAxisSelect;
  vec4_scalar vColor0, vColor1;
  Float vPos;
  struct InterpOutputs {
    float vPos;
  };
  void main() {   // :17-24
    vPos = mix(Float(0.0f), Float(1.0f), mix(aPosition.x, aPosition.y, Float(aAxisSelect)));
    vColor0 = aColor0;
    vColor1 = aColor1;
    gl_Position = uTransform * vec4(mix(aTaskRect.sel(X, Y), aTaskRect.sel(Z, W), aPosition), 0.0f, 1.0f);
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start, instance, count);
    load_flat_attrib(self->aTaskRect, attribs[L[self->a_aTaskRect]], start, instance, count);
    load_flat_attrib(self->aColor0, attribs[L[self->a_aColor0]], start, instance, count);
    load_flat_attrib(self->aColor1, attribs[L[self->a_aColor1]], start, instance, count);
    load_flat_attrib(self->aAxisSelect, attribs[L[self->a_aAxisSelect]], start, instance, count);
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->vPos = get_nth(vPos, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  cs_fast_linear_gradient_vert() {
    using namespace wrsh;
    used = (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_aTaskRect = attribs.add("aTaskRect");
    a_aColor0 = attribs.add("aColor0");
    a_aColor1 = attribs.add("aColor1");
    a_aAxisSelect = attribs.add("aAxisSelect");
    WRSH_VERT_WIRING(Self)
  }
};

struct cs_fast_linear_gradient_frag : FragmentShaderImpl, cs_fast_linear_gradient_vert {
  typedef cs_fast_linear_gradient_frag Self;
  typedef cs_fast_linear_gradient_vert::InterpOutputs InterpInputs;
  InterpInputs interp_step;
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_,
                                 const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    self->vPos = init_interp(init->vPos, step->vPos);
    self->interp_step.vPos = step->vPos * 4.0f;
  }
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) {
    float chunks = steps * 0.25f;
    vPos += interp_step.vPos * chunks;
  }
  void main() {   // :28-30
    gl_FragColor = mix(vec4(vColor0), vec4(vColor1), vPos);
  }
  WRSH_FRAG_ABI(Self)
  cs_fast_linear_gradient_frag() {
    WRSH_FRAG_WIRING()
  }
};

WRSH_PROGRAM(cs_fast_linear_gradient, "cs_fast_linear_gradient")
