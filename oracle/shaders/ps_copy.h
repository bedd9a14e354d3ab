// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "ps_copy" (webrender_build/src/shader_features.rs:238).
// Restates webrender/res/ps_copy.glsl:7-39 with SWGL defined: texture-cache copies and
// batched uploads (renderer/mod.rs:1808-1846, renderer/upload.rs:540-620).  No span
// function exists: every chunk runs main(), a texelFetch at the truncated uv.

struct ps_copy_vert : VertexShaderImpl, wrsh::CommonState {
  typedef ps_copy_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_src, a_dst, a_size;
  vec2 aPosition;
  vec4_scalar a_src_rect, a_dst_rect;
  vec2_scalar a_dst_texture_size;
  vec2 v_uv;
  struct InterpOutputs {
    vec2_scalar v_uv;
  };
  void main() {
    // unnormalised device space: the fragment stage fetches texels
    v_uv = mix(a_src_rect.sel(X, Y), a_src_rect.sel(Z, W), aPosition);
    vec2 pos = mix(a_dst_rect.sel(X, Y), a_dst_rect.sel(Z, W), aPosition);
    gl_Position = vec4(pos / (a_dst_texture_size * 0.5f) - vec2_scalar(1.0f, 1.0f), 0.0f, 1.0f);
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start, instance, count);
    load_flat_attrib(self->a_src_rect, attribs[L[self->a_src]], start, instance, count);
    load_flat_attrib(self->a_dst_rect, attribs[L[self->a_dst]], start, instance, count);
    load_flat_attrib(self->a_dst_texture_size, attribs[L[self->a_size]], start, instance, count);
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->v_uv = get_nth(v_uv, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  ps_copy_vert() {
    using namespace wrsh;
    used = (1u << U_sColor0);
    a_aPosition = attribs.add("aPosition");
    a_src = attribs.add("a_src_rect");
    a_dst = attribs.add("a_dst_rect");
    a_size = attribs.add("a_dst_texture_size");
    WRSH_VERT_WIRING(Self)
  }
};

struct ps_copy_frag : FragmentShaderImpl, ps_copy_vert {
  typedef ps_copy_frag Self;
  typedef ps_copy_vert::InterpOutputs InterpInputs;
  InterpInputs interp_step;
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_,
                                 const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    self->v_uv = init_interp(init->v_uv, step->v_uv);
    self->interp_step.v_uv = step->v_uv * 4.0f;
  }
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) {
    float chunks = steps * 0.25f;
    v_uv += interp_step.v_uv * chunks;
  }
  void main() { gl_FragColor = texelFetch(sColor0, make_ivec2(v_uv), 0); }
  WRSH_FRAG_ABI(Self)
  ps_copy_frag() { WRSH_FRAG_WIRING() }
};

WRSH_PROGRAM(ps_copy, "ps_copy")
