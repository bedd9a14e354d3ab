// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_line_decoration" (webrender_build/src/shader_features.rs).
// Restates webrender/res/cs_line_decoration.glsl:14-165 (+ shared.glsl:110-189)
// with SWGL defined.  The program has no span function: every pixel runs main().

struct cs_line_decoration_vert : VertexShaderImpl, wrsh::CommonState {
  typedef cs_line_decoration_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aTaskRect, a_aLocalSize, a_aWavy, a_aStyle, a_aAxisSelect;
  vec2 aPosition;
  vec4_scalar aTaskRect;
  vec2_scalar aLocalSize;
  float aWavyLineThickness, aAxisSelect;
  int32_t aStyle;
  ivec2_scalar vStyle;
  vec4_scalar vParams;
  vec2 vLocalPos;
  struct InterpOutputs {
    vec2_scalar vLocalPos;
  };
  void main() {   // :43-98
    vec2_scalar size = mix(aLocalSize, aLocalSize.sel(Y, X), aAxisSelect);
    vStyle = ivec2_scalar(aStyle, 0);
    switch (vStyle.x) {
      case 0: break;
      case 2: {   // DASHED
        vParams = vec4_scalar(size.x, 0.5f * size.x, 0.0f, 0.0f);
        break;
      }
      case 1: {   // DOTTED
        float diameter = size.y;
        float period = diameter * 2.0f;
        float center_line = 0.5f * size.y;
        vParams = vec4_scalar(period, diameter / 2.0f, center_line, 0.0f);
        break;
      }
      case 3: {   // WAVY
        float line_thickness = max(aWavyLineThickness, 1.0f);
        float slope_length = size.y - line_thickness;
        float flat_length = max((line_thickness - 1.0f) * 2.0f, 1.0f);
        vParams = vec4_scalar(line_thickness / 2.0f, slope_length, flat_length, size.y);
        break;
      }
      default: vParams = vec4_scalar(0.0f, 0.0f, 0.0f, 0.0f);
    }
    vLocalPos = mix(aPosition, aPosition.sel(Y, X), aAxisSelect) * size;
    gl_Position = uTransform * vec4(mix(aTaskRect.sel(X, Y), aTaskRect.sel(Z, W), aPosition), 0.0f, 1.0f);
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start, instance, count);
    load_flat_attrib(self->aTaskRect, attribs[L[self->a_aTaskRect]], start, instance, count);
    load_flat_attrib(self->aLocalSize, attribs[L[self->a_aLocalSize]], start, instance, count);
    load_flat_attrib(self->aWavyLineThickness, attribs[L[self->a_aWavy]], start, instance, count);
    load_flat_attrib(self->aStyle, attribs[L[self->a_aStyle]], start, instance, count);
    load_flat_attrib(self->aAxisSelect, attribs[L[self->a_aAxisSelect]], start, instance, count);
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->vLocalPos = get_nth(vLocalPos, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  cs_line_decoration_vert() {
    using namespace wrsh;
    used = (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_aTaskRect = attribs.add("aTaskRect");
    a_aLocalSize = attribs.add("aLocalSize");
    a_aWavy = attribs.add("aWavyLineThickness");
    a_aStyle = attribs.add("aStyle");
    a_aAxisSelect = attribs.add("aAxisSelect");
    vParams = vec4_scalar(0.0f, 0.0f, 0.0f, 0.0f);
    WRSH_VERT_WIRING(Self)
  }
};

struct cs_line_decoration_frag : FragmentShaderImpl, cs_line_decoration_vert {
  typedef cs_line_decoration_frag Self;
  typedef cs_line_decoration_vert::InterpOutputs InterpInputs;
  InterpInputs interp_step;
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_,
                                 const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    self->vLocalPos = init_interp(init->vLocalPos, step->vLocalPos);
    self->interp_step.vLocalPos = step->vLocalPos * 4.0f;
  }
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) {
    float chunks = steps * 0.25f;
    vLocalPos += interp_step.vLocalPos * chunks;
  }
  static Float distance_to_line(vec2 p0, vec2 perp_dir, vec2 p) {      // shared.glsl:110-113
    vec2 dir_to_p0 = p0 - p;
    return dot(normalize(perp_dir), dir_to_p0);
  }
  static Float distance_aa(float aa_range, Float signed_distance) {
    Float dist = signed_distance * aa_range;
    return clamp(0.5f - dist, Float(0.0f), Float(1.0f));
  }
  void main() {   // :104-163
    vec2 pos = vLocalPos;
    float aa_range = recip(fwidth(pos).x);   // compute_aa_range, shared.glsl:145-148
    Float alpha = 1.0f;
    switch (vStyle.x) {
      case 0: break;
      case 2: {
        alpha = step(floor(pos.x + 0.5f), Float(vParams.y));
        break;
      }
      case 1: {
        vec2 dot_relative_pos = pos - vParams.sel(Y, Z);
        Float dot_distance = length(dot_relative_pos) - vParams.y;
        alpha = distance_aa(aa_range, dot_distance);
        break;
      }
      case 3: {
        float half_line_thickness = vParams.x;
        float slope_length = vParams.y;
        float flat_length = vParams.z;
        float vertical_bounds = vParams.w;
        float half_period = slope_length + flat_length;
        float mid_height = vertical_bounds / 2.0f;
        Float peak_offset = mid_height - half_line_thickness;
        Float flip = -2.0f * (step(mod(pos.x, Float(2.0f * half_period)), Float(half_period)) - 0.5f);
        peak_offset *= flip;
        Float peak_height = mid_height + peak_offset;
        pos.x = mod(pos.x, Float(half_period));
        Float dist1 = distance_to_line(vec2(Float(0.0f), peak_height), vec2(Float(1.0f), -flip), pos);
        Float dist2 = distance_to_line(vec2(Float(0.0f), peak_height), vec2(Float(0.0f), -flip), pos);
        Float dist3 = distance_to_line(vec2(Float(flat_length), peak_height), vec2(Float(-1.0f), -flip), pos);
        Float dist = abs(max(max(dist1, dist2), dist3));
        alpha = distance_aa(aa_range, dist - half_line_thickness);
        if (half_line_thickness <= 1.0f) {
          alpha = 1.0f - step(alpha, Float(0.5f));   // MAGIC_WAVY_LINE_AA_SNAP
        }
        break;
      }
      default: break;
    }
    gl_FragColor = vec4(alpha);
  }
  WRSH_FRAG_ABI(Self)
  cs_line_decoration_frag() {
    WRSH_FRAG_WIRING()
  }
};

WRSH_PROGRAM(cs_line_decoration, "cs_line_decoration")
