// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_border_segment" (webrender_build/src/shader_features.rs:194).
// Restates webrender/res/cs_border_segment.glsl:87-450 (+ ellipse.glsl:9-45,
// shared.glsl:110-189) with SWGL defined.  The program has no span function:
// every pixel runs main().

struct cs_border_segment_vert : VertexShaderImpl, wrsh::CommonState {
  typedef cs_border_segment_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aTaskOrigin, a_aRect, a_aColor0, a_aColor1, a_aFlags, a_aWidths, a_aRadii, a_aClipParams1, a_aClipParams2;
  vec2 aPosition;
  vec2_scalar aTaskOrigin, aWidths, aRadii;
  vec4_scalar aRect, aColor0, aColor1, aClipParams1, aClipParams2;
  int32_t aFlags;
  // flat varyings
  vec4_scalar vColor00, vColor01, vColor10, vColor11, vColorLine;
  vec2_scalar vSegmentClipMode;
  vec4_scalar vStyleEdgeAxis, vClipCenter_Sign, vClipRadii, vEdgeReference, vPartialWidths, vClipParams1, vClipParams2;
  // interpolated
  vec2 vPos;
  struct InterpOutputs {
    vec2_scalar vPos;
  };
  static vec2_scalar get_outer_corner_scale(int segment) {   // :92-116
    switch (segment) {
      case 0: return vec2_scalar(0.0f, 0.0f);
      case 1: return vec2_scalar(1.0f, 0.0f);
      case 2: return vec2_scalar(1.0f, 1.0f);
      case 3: return vec2_scalar(0.0f, 1.0f);
      default: return vec2_scalar(0.0f, 0.0f);
    }
  }
  static vec4_scalar mod_color(vec4_scalar color, bool is_black, bool lighter) {   // :121-143
    const float light_black = 0.7f, dark_black = 0.3f, dark_scale = 0.66666666f, light_scale = 1.0f;
    if (is_black) {
      if (lighter) return vec4_scalar(light_black, light_black, light_black, color.w);
      return vec4_scalar(dark_black, dark_black, dark_black, color.w);
    }
    if (lighter) return vec4_scalar(color.x * light_scale, color.y * light_scale, color.z * light_scale, color.w);
    return vec4_scalar(color.x * dark_scale, color.y * dark_scale, color.z * dark_scale, color.w);
  }
  static void get_colors_for_side(vec4_scalar color, int style, vec4_scalar& r0, vec4_scalar& r1) {   // :145-168
    bool is_black = color.x == 0.0f && color.y == 0.0f && color.z == 0.0f;
    switch (style) {
      case 6 /* GROOVE */: r0 = mod_color(color, is_black, true); r1 = mod_color(color, is_black, false); break;
      case 7 /* RIDGE */: r0 = mod_color(color, is_black, false); r1 = mod_color(color, is_black, true); break;
      default: r0 = color; r1 = color; break;
    }
  }
  void main() {   // :170-267
    int segment = aFlags & 0xff;
    int style0 = (aFlags >> 8) & 0xff;
    int style1 = (aFlags >> 16) & 0xff;
    int clip_mode = (aFlags >> 24) & 0x0f;
    vec2_scalar size = aRect.sel(Z, W) - aRect.sel(X, Y);
    vec2_scalar outer_scale = get_outer_corner_scale(segment);
    vec2_scalar outer = outer_scale * size;
    vec2_scalar clip_sign = 1.0f - 2.0f * outer_scale;
    ivec2_scalar edge_axis = ivec2_scalar(0, 0);
    vec2_scalar edge_reference = vec2_scalar(0.0f, 0.0f);
    switch (segment) {
      case 0: edge_axis = ivec2_scalar(0, 1); edge_reference = outer; break;
      case 1: edge_axis = ivec2_scalar(1, 0); edge_reference = vec2_scalar(outer.x - aWidths.x, outer.y); break;
      case 2: edge_axis = ivec2_scalar(0, 1); edge_reference = outer - aWidths; break;
      case 3: edge_axis = ivec2_scalar(1, 0); edge_reference = vec2_scalar(outer.x, outer.y - aWidths.y); break;
      case 5: case 7: edge_axis = ivec2_scalar(1, 1); break;
      default: break;
    }
    vSegmentClipMode = vec2_scalar(float(segment), float(clip_mode));
    vStyleEdgeAxis = vec4_scalar(float(style0), float(style1), float(edge_axis.x), float(edge_axis.y));
    vec2_scalar w3 = aWidths / 3.0f, w2 = aWidths / 2.0f;
    vPartialWidths = vec4_scalar(w3.x, w3.y, w2.x, w2.y);
    vPos = size * aPosition;
    get_colors_for_side(aColor0, style0, vColor00, vColor01);
    get_colors_for_side(aColor1, style1, vColor10, vColor11);
    vec2_scalar ccs = outer + clip_sign * aRadii;
    vClipCenter_Sign = vec4_scalar(ccs.x, ccs.y, clip_sign.x, clip_sign.y);
    vec2_scalar inner = max(aRadii - aWidths, 0.0f);
    vClipRadii = vec4_scalar(aRadii.x, aRadii.y, inner.x, inner.y);
    vColorLine = vec4_scalar(outer.x, outer.y, aWidths.y * -clip_sign.y, aWidths.x * clip_sign.x);
    vec2_scalar er1 = edge_reference + aWidths;
    vEdgeReference = vec4_scalar(edge_reference.x, edge_reference.y, er1.x, er1.y);
    vClipParams1 = aClipParams1;
    vClipParams2 = aClipParams2;
    if (clip_mode == 3 /* CLIP_DOT */) {
      float radius = aClipParams1.z;
      if (radius > 0.5f) radius += 2.0f;
      vPos = vClipParams1.sel(X, Y) + radius * (2.0f * aPosition - 1.0f);
      vPos = clamp(vPos, vec2_scalar(0.0f, 0.0f), size);
    } else if (clip_mode == 1 /* CLIP_DASH_CORNER */) {
      vec2_scalar center = (aClipParams1.sel(X, Y) + aClipParams2.sel(X, Y)) * 0.5f;
      float dash_length = length(aClipParams1.sel(X, Y) - aClipParams2.sel(X, Y));
      float width = max(aWidths.x, aWidths.y);
      float rr = max(dash_length, width) + 2.0f;
      vec2_scalar r = vec2_scalar(rr, rr);
      vPos = clamp(vPos, center - r, center + r);
    }
    gl_Position = uTransform * vec4(aTaskOrigin + aRect.sel(X, Y) + vPos, 0.0f, 1.0f);
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    load_attrib(self->aPosition, attribs[L[self->a_aPosition]], start, instance, count);
    load_flat_attrib(self->aTaskOrigin, attribs[L[self->a_aTaskOrigin]], start, instance, count);
    load_flat_attrib(self->aRect, attribs[L[self->a_aRect]], start, instance, count);
    load_flat_attrib(self->aColor0, attribs[L[self->a_aColor0]], start, instance, count);
    load_flat_attrib(self->aColor1, attribs[L[self->a_aColor1]], start, instance, count);
    load_flat_attrib(self->aFlags, attribs[L[self->a_aFlags]], start, instance, count);
    load_flat_attrib(self->aWidths, attribs[L[self->a_aWidths]], start, instance, count);
    load_flat_attrib(self->aRadii, attribs[L[self->a_aRadii]], start, instance, count);
    load_flat_attrib(self->aClipParams1, attribs[L[self->a_aClipParams1]], start, instance, count);
    load_flat_attrib(self->aClipParams2, attribs[L[self->a_aClipParams2]], start, instance, count);
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->vPos = get_nth(vPos, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  cs_border_segment_vert() {
    using namespace wrsh;
    used = (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_aTaskOrigin = attribs.add("aTaskOrigin");
    a_aRect = attribs.add("aRect");
    a_aColor0 = attribs.add("aColor0");
    a_aColor1 = attribs.add("aColor1");
    a_aFlags = attribs.add("aFlags");
    a_aWidths = attribs.add("aWidths");
    a_aRadii = attribs.add("aRadii");
    a_aClipParams1 = attribs.add("aClipParams1");
    a_aClipParams2 = attribs.add("aClipParams2");
    WRSH_VERT_WIRING(Self)
  }
};

struct cs_border_segment_frag : FragmentShaderImpl, cs_border_segment_vert {
  typedef cs_border_segment_frag Self;
  typedef cs_border_segment_vert::InterpOutputs InterpInputs;
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
  static vec2_scalar inverse_radii_squared(vec2_scalar radii) { return 1.0f / max(radii * radii, 1.0e-6f); }
  static Float distance_to_ellipse_approx(vec2 p, vec2_scalar inv_radii_sq, float scale) {
    vec2 p_r = p * inv_radii_sq;
    Float g = dot(p, p_r) - scale;
    vec2 dG = (1.0f + scale) * p_r;
    return g * inversesqrt(dot(dG, dG));
  }
  static Float distance_to_ellipse(vec2 p, vec2_scalar radii) {
    return distance_to_ellipse_approx(p, inverse_radii_squared(radii), float(radii.x > 0.0f && radii.y > 0.0f));
  }
  static Float distance_to_line(vec2_scalar p0, vec2_scalar perp_dir, vec2 p) {
    vec2 dir_to_p0 = p0 - p;
    return dot(vec2(normalize(perp_dir)), dir_to_p0);
  }
  static Float distance_aa(float aa_range, Float signed_distance) {
    Float dist = signed_distance * aa_range;
    return clamp(0.5f - dist, Float(0.0f), Float(1.0f));
  }
  // :271-326
  vec4 evaluate_color_for_style_in_corner(vec2 clip_relative_pos, int style, vec4_scalar color0s, vec4_scalar color1s,
                                          vec4_scalar clip_radii, Float mix_factor, int segment, float aa_range) {
    vec4 color0 = vec4(color0s), color1 = vec4(color1s);
    switch (style) {
      case 2 /* DOUBLE */: {
        Float d_radii_a = distance_to_ellipse(clip_relative_pos, clip_radii.sel(X, Y) - vPartialWidths.sel(X, Y));
        Float d_radii_b = distance_to_ellipse(clip_relative_pos, clip_radii.sel(X, Y) - 2.0f * vPartialWidths.sel(X, Y));
        Float d = min(-d_radii_a, d_radii_b);
        color0 *= distance_aa(aa_range, d);
        break;
      }
      case 6: case 7: {
        Float d = distance_to_ellipse(clip_relative_pos, clip_radii.sel(X, Y) - vPartialWidths.sel(Z, W));
        Float alpha = distance_aa(aa_range, d);
        Float swizzled_factor;
        switch (segment) {
          case 0: swizzled_factor = 0.0f; break;
          case 1: swizzled_factor = mix_factor; break;
          case 2: swizzled_factor = 1.0f; break;
          case 3: swizzled_factor = 1.0f - mix_factor; break;
          default: swizzled_factor = 0.0f; break;
        }
        vec4 c0 = mix(color1, color0, swizzled_factor);
        vec4 c1 = mix(color0, color1, swizzled_factor);
        color0 = mix(c0, c1, alpha);
        break;
      }
      default: break;
    }
    return color0;
  }
  // :328-367
  vec4 evaluate_color_for_style_in_edge(vec2 pos_vec, int style, vec4_scalar color0s, vec4_scalar color1s, float aa_range,
                                        int edge_axis_id) {
    vec4 color0 = vec4(color0s), color1 = vec4(color1s);
    vec2_scalar edge_axis = edge_axis_id != 0 ? vec2_scalar(0.0f, 1.0f) : vec2_scalar(1.0f, 0.0f);
    Float pos = dot(pos_vec, vec2(edge_axis));
    switch (style) {
      case 2: {
        Float d = -1.0f;
        float partial_width = dot(vPartialWidths.sel(X, Y), edge_axis);
        if (partial_width >= 1.0f) {
          vec2_scalar ref = vec2_scalar(dot(vEdgeReference.sel(X, Y), edge_axis) + partial_width,
                                        dot(vEdgeReference.sel(Z, W), edge_axis) - partial_width);
          d = min(pos - ref.x, ref.y - pos);
        }
        color0 *= distance_aa(aa_range, d);
        break;
      }
      case 6: case 7: {
        float ref = dot(vEdgeReference.sel(X, Y) + vPartialWidths.sel(Z, W), edge_axis);
        Float d = pos - ref;
        Float alpha = distance_aa(aa_range, d);
        color0 = mix(color0, color1, alpha);
        break;
      }
      default: break;
    }
    return color0;
  }
  static vec4 select4(I32 c, vec4 t, vec4 e) {
    return vec4(if_then_else(c, t.x, e.x), if_then_else(c, t.y, e.y), if_then_else(c, t.z, e.z), if_then_else(c, t.w, e.w));
  }
  void main() {   // :369-449
    float aa_range = recip(fwidth(vPos).x);   // compute_aa_range, shared.glsl:145-148
    int segment = int(vSegmentClipMode.x);
    int clip_mode = int(vSegmentClipMode.y);
    ivec2_scalar style = ivec2_scalar(int(vStyleEdgeAxis.x), int(vStyleEdgeAxis.y));
    ivec2_scalar edge_axis = ivec2_scalar(int(vStyleEdgeAxis.z), int(vStyleEdgeAxis.w));
    Float mix_factor = 0.0f;
    if (edge_axis.x != edge_axis.y) {
      Float d_line = distance_to_line(vColorLine.sel(X, Y), vColorLine.sel(Z, W), vPos);
      mix_factor = distance_aa(aa_range, -d_line);
    }
    vec2 clip_relative_pos = vPos - vClipCenter_Sign.sel(X, Y);
    I32 in_clip_region = (vClipCenter_Sign.z * clip_relative_pos.x < 0.0f) & (vClipCenter_Sign.w * clip_relative_pos.y < 0.0f);
    Float d = -1.0f;
    switch (clip_mode) {
      case 3 /* CLIP_DOT */: {
        d = distance(vClipParams1.sel(X, Y), vPos) - vClipParams1.z;
        break;
      }
      case 2 /* CLIP_DASH_EDGE */: {
        bool is_vertical = vClipParams1.x == 0.0f;
        float half_dash = is_vertical ? vClipParams1.y : vClipParams1.x;
        Float pos = is_vertical ? vPos.y : vPos.x;
        I32 in_dash = (pos < half_dash) | (pos > 3.0f * half_dash);
        d = if_then_else(in_dash, d, Float(1.0f));
        break;
      }
      case 1 /* CLIP_DASH_CORNER */: {
        Float d0 = distance_to_line(vClipParams1.sel(X, Y), vClipParams1.sel(Z, W), vPos);
        Float d1 = distance_to_line(vClipParams2.sel(X, Y), vClipParams2.sel(Z, W), vPos);
        d = max(d0, -d1);
        break;
      }
      default: break;
    }
    Float d_radii_a = distance_to_ellipse(clip_relative_pos, vClipRadii.sel(X, Y));
    Float d_radii_b = distance_to_ellipse(clip_relative_pos, vClipRadii.sel(Z, W));
    Float d_radii = max(d_radii_a, -d_radii_b);
    d = if_then_else(in_clip_region, max(d, d_radii), d);
    vec4 c0_corner = evaluate_color_for_style_in_corner(clip_relative_pos, style.x, vColor00, vColor01, vClipRadii, mix_factor, segment, aa_range);
    vec4 c1_corner = evaluate_color_for_style_in_corner(clip_relative_pos, style.y, vColor10, vColor11, vClipRadii, mix_factor, segment, aa_range);
    vec4 c0_edge = evaluate_color_for_style_in_edge(vPos, style.x, vColor00, vColor01, aa_range, edge_axis.x);
    vec4 c1_edge = evaluate_color_for_style_in_edge(vPos, style.y, vColor10, vColor11, aa_range, edge_axis.y);
    vec4 color0 = select4(in_clip_region, c0_corner, c0_edge);
    vec4 color1 = select4(in_clip_region, c1_corner, c1_edge);
    Float alpha = distance_aa(aa_range, d);
    vec4 color = mix(color0, color1, mix_factor);
    gl_FragColor = color * alpha;
  }
  WRSH_FRAG_ABI(Self)
  cs_border_segment_frag() {
    WRSH_FRAG_WIRING()
  }
};

WRSH_PROGRAM(cs_border_segment, "cs_border_segment")
