// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_conic_gradient" (webrender_build/src/shader_features.rs).
// Restates webrender/res/cs_conic_gradient.glsl:9-67 and gradient.glsl:30-61
// (sample_gradient) with SWGL defined.  The program has no span function:
// every pixel runs main() (atan = libm atan2f, glsl.h:2838-2843).

struct cs_conic_gradient_vert : VertexShaderImpl, wrsh::CommonState {
  typedef cs_conic_gradient_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aTaskRect, a_aCenter, a_aScale, a_aStartOffset, a_aEndOffset, a_aAngle, a_aExtendMode, a_aGradientStopsAddress;
  vec2 aPosition;
  vec4_scalar aTaskRect;
  vec2_scalar aCenter, aScale;
  float aStartOffset, aEndOffset, aAngle;
  int32_t aExtendMode, aGradientStopsAddress;
  vec2 v_pos;
  vec2_scalar v_center, v_gradient_repeat;
  vec3_scalar v_start_offset_offset_scale_angle_vec;
  ivec2_scalar v_gradient_address;
  struct InterpOutputs {
    vec2_scalar v_pos;
  };
  void main() {   // :30-48
    float d = aEndOffset - aStartOffset;
    v_start_offset_offset_scale_angle_vec.y = d != 0.0f ? 1.0f / d : 0.0f;
    vec2 pos = mix(aTaskRect.sel(X, Y), aTaskRect.sel(Z, W), aPosition);
    gl_Position = uTransform * vec4(pos, 0.0f, 1.0f);
    v_start_offset_offset_scale_angle_vec.z = 3.141592653589793f / 2.0f - aAngle;
    v_start_offset_offset_scale_angle_vec.x = aStartOffset * v_start_offset_offset_scale_angle_vec.y;
    v_center = aCenter * v_start_offset_offset_scale_angle_vec.y;
    v_pos = (aTaskRect.sel(Z, W) - aTaskRect.sel(X, Y)) * aPosition * v_start_offset_offset_scale_angle_vec.y * aScale;
    v_gradient_repeat.x = float(aExtendMode == 1 /* EXTEND_MODE_REPEAT */);
    v_gradient_address.x = aGradientStopsAddress;
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start, instance, count);
    load_flat_attrib(self->aTaskRect, attribs[L[self->a_aTaskRect]], start, instance, count);
    load_flat_attrib(self->aCenter, attribs[L[self->a_aCenter]], start, instance, count);
    load_flat_attrib(self->aScale, attribs[L[self->a_aScale]], start, instance, count);
    load_flat_attrib(self->aStartOffset, attribs[L[self->a_aStartOffset]], start, instance, count);
    load_flat_attrib(self->aEndOffset, attribs[L[self->a_aEndOffset]], start, instance, count);
    load_flat_attrib(self->aAngle, attribs[L[self->a_aAngle]], start, instance, count);
    load_flat_attrib(self->aExtendMode, attribs[L[self->a_aExtendMode]], start, instance, count);
    load_flat_attrib(self->aGradientStopsAddress, attribs[L[self->a_aGradientStopsAddress]], start, instance, count);
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->v_pos = get_nth(v_pos, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  cs_conic_gradient_vert() {
    using namespace wrsh;
    used = (1u << U_uTransform) | (1u << U_sGpuBufferF);
    a_aPosition = attribs.add("aPosition");
    a_aTaskRect = attribs.add("aTaskRect");
    a_aCenter = attribs.add("aCenter");
    a_aScale = attribs.add("aScale");
    a_aStartOffset = attribs.add("aStartOffset");
    a_aEndOffset = attribs.add("aEndOffset");
    a_aAngle = attribs.add("aAngle");
    a_aExtendMode = attribs.add("aExtendMode");
    a_aGradientStopsAddress = attribs.add("aGradientStopsAddress");
    v_center = vec2_scalar(0.0f, 0.0f);
    v_start_offset_offset_scale_angle_vec = vec3_scalar(0.0f, 0.0f, 0.0f);
    v_gradient_repeat = vec2_scalar(0.0f, 0.0f);
    v_gradient_address = ivec2_scalar(0, 0);
    WRSH_VERT_WIRING(Self)
  }
};

struct cs_conic_gradient_frag : FragmentShaderImpl, cs_conic_gradient_vert {
  typedef cs_conic_gradient_frag Self;
  typedef cs_conic_gradient_vert::InterpOutputs InterpInputs;
  InterpInputs interp_step;
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_,
                                 const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    self->v_pos = init_interp(init->v_pos, step->v_pos);
    self->interp_step.v_pos = step->v_pos * 4.0f;
  }
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) {
    float chunks = steps * 0.25f;
    v_pos += interp_step.v_pos * chunks;
  }
  vec4 sample_gradient(Float offset) const {   // gradient.glsl:30-61
    offset -= floor(offset) * v_gradient_repeat.x;
    Float x = clamp(1.0f + offset * 128.0f, 0.0f, 1.0f + 128.0f);
    Float entry_index = floor(x);
    Float entry_fract = x - entry_index;
    I32 address = v_gradient_address.x + 2 * cast(entry_index);
    vec4 t0, t1;
    for (int n = 0; n < 4; n++) {
      ivec2_scalar uv = wrsh::get_gpu_uv(address[n]);
      put_nth(t0, n, texelFetch(sGpuBufferF, uv, 0));
      put_nth(t1, n, texelFetch(sGpuBufferF, ivec2_scalar(uv.x + 1, uv.y), 0));
    }
    return t0 + t1 * entry_fract;
  }
  void main() {   // :52-65
    vec2 current_dir = v_pos - v_center;
    Float current_angle = atan(current_dir.y, current_dir.x) + v_start_offset_offset_scale_angle_vec.z;
    Float offset = fract(current_angle / (2.0f * 3.141592653589793f)) * v_start_offset_offset_scale_angle_vec.y -
                   v_start_offset_offset_scale_angle_vec.x;
    gl_FragColor = sample_gradient(offset);
  }
  WRSH_FRAG_ABI(Self)
  cs_conic_gradient_frag() {
    WRSH_FRAG_WIRING()
  }
};

WRSH_PROGRAM(cs_conic_gradient, "cs_conic_gradient")
