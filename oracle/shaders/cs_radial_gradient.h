// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_radial_gradient" (webrender_build/src/shader_features.rs).
// Restates webrender/res/cs_radial_gradient.glsl:9-71 and gradient.glsl:30-61
// (sample_gradient) with SWGL defined.  The span shader hands the row to the
// reference's own swgl_commitRadialGradientRGBA8 (swgl_ext.h:1629-1858).

struct cs_radial_gradient_vert : VertexShaderImpl, wrsh::CommonState {
  typedef cs_radial_gradient_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aTaskRect, a_aCenter, a_aScale, a_aStartRadius, a_aEndRadius, a_aXYRatio, a_aExtendMode, a_aGradientStopsAddress;
  vec2 aPosition;
  vec4_scalar aTaskRect;
  vec2_scalar aCenter, aScale;
  float aStartRadius, aEndRadius, aXYRatio;
  int32_t aExtendMode, aGradientStopsAddress;
  vec2 v_pos;
  vec2_scalar v_start_radius, v_gradient_repeat;
  ivec2_scalar v_gradient_address;
  struct InterpOutputs {
    vec2_scalar v_pos;
  };
  void main() {   // :26-47
    float rd = aEndRadius - aStartRadius;
    float radius_scale = rd != 0.0f ? 1.0f / rd : 0.0f;
    vec2 pos = mix(aTaskRect.sel(X, Y), aTaskRect.sel(Z, W), aPosition);
    gl_Position = uTransform * vec4(pos, 0.0f, 1.0f);
    v_start_radius.x = aStartRadius * radius_scale;
    v_pos = ((aTaskRect.sel(Z, W) - aTaskRect.sel(X, Y)) * aPosition * aScale - aCenter) * radius_scale;
    v_pos.y *= aXYRatio;
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
    load_flat_attrib(self->aStartRadius, attribs[L[self->a_aStartRadius]], start, instance, count);
    load_flat_attrib(self->aEndRadius, attribs[L[self->a_aEndRadius]], start, instance, count);
    load_flat_attrib(self->aXYRatio, attribs[L[self->a_aXYRatio]], start, instance, count);
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
  cs_radial_gradient_vert() {
    using namespace wrsh;
    used = (1u << U_uTransform) | (1u << U_sGpuBufferF);
    a_aPosition = attribs.add("aPosition");
    a_aTaskRect = attribs.add("aTaskRect");
    a_aCenter = attribs.add("aCenter");
    a_aScale = attribs.add("aScale");
    a_aStartRadius = attribs.add("aStartRadius");
    a_aEndRadius = attribs.add("aEndRadius");
    a_aXYRatio = attribs.add("aXYRatio");
    a_aExtendMode = attribs.add("aExtendMode");
    a_aGradientStopsAddress = attribs.add("aGradientStopsAddress");
    v_start_radius = vec2_scalar(0.0f, 0.0f);
    v_gradient_repeat = vec2_scalar(0.0f, 0.0f);
    v_gradient_address = ivec2_scalar(0, 0);
    WRSH_VERT_WIRING(Self)
  }
};

struct cs_radial_gradient_frag : FragmentShaderImpl, cs_radial_gradient_vert {
  typedef cs_radial_gradient_frag Self;
  typedef cs_radial_gradient_vert::InterpOutputs InterpInputs;
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
  void main() {   // :51-56
    Float offset = length(v_pos) - v_start_radius.x;
    gl_FragColor = sample_gradient(offset);
  }
  void swgl_drawSpanRGBA8() {   // :58-69
    int address = swgl_validateGradient(sGpuBufferF, wrsh::get_gpu_uv(v_gradient_address.x), int(128.0f + 2.0f));
    if (address < 0) {
      return;
    }
    swgl_commitRadialGradientRGBA8(sGpuBufferF, address, 128.0f, v_gradient_repeat.x != 0.0f, v_pos, v_start_radius.x);
  }
  WRSH_FRAG_ABI(Self)
  static int draw_span_RGBA8(FragmentShaderImpl* impl) {
    Self* self = (Self*)impl;
    DISPATCH_DRAW_SPAN(self, RGBA8);
  }
  cs_radial_gradient_frag() {
    WRSH_FRAG_WIRING()
    draw_span_RGBA8_func = &draw_span_RGBA8;
  }
};

WRSH_PROGRAM(cs_radial_gradient, "cs_radial_gradient")
