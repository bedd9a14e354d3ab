// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_border_solid" (webrender_build/src/shader_features.rs:193).
// Restates webrender/res/cs_border_solid.glsl:56-178 (+ ellipse.glsl:9-45,
// shared.glsl:110-189) with SWGL defined.  The program has no span function:
// every pixel runs main().

struct cs_border_solid_vert : VertexShaderImpl, wrsh::CommonState {
  typedef cs_border_solid_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aTaskOrigin, a_aRect, a_aColor0, a_aColor1, a_aFlags, a_aWidths, a_aRadii, a_aClipParams1, a_aClipParams2;
  vec2 aPosition;
  vec2_scalar aTaskOrigin, aWidths, aRadii;
  vec4_scalar aRect, aColor0, aColor1, aClipParams1, aClipParams2;
  int32_t aFlags;
  // flat varyings
  vec4_scalar vColor0, vColor1, vColorLine;
  ivec2_scalar vMixColors;
  vec4_scalar vClipCenter_Sign, vClipRadii, vHorizontalClipCenter_Sign, vVerticalClipCenter_Sign;
  vec2_scalar vHorizontalClipRadii, vVerticalClipRadii;
  // interpolated
  vec2 vPos;
  struct InterpOutputs {
    vec2_scalar vPos;
  };
  static vec2_scalar get_outer_corner_scale(int segment) {   // :58-82
    switch (segment) {
      case 0: return vec2_scalar(0.0f, 0.0f);
      case 1: return vec2_scalar(1.0f, 0.0f);
      case 2: return vec2_scalar(1.0f, 1.0f);
      case 3: return vec2_scalar(0.0f, 1.0f);
      default: return vec2_scalar(0.0f, 0.0f);
    }
  }
  void main() {   // :84-128
    int segment = aFlags & 0xff;
    bool do_aa = ((aFlags >> 24) & 0xf0) != 0;
    vec2_scalar outer_scale = get_outer_corner_scale(segment);
    vec2_scalar size = aRect.sel(Z, W) - aRect.sel(X, Y);
    vec2_scalar outer = outer_scale * size;
    vec2_scalar clip_sign = 1.0f - 2.0f * outer_scale;
    int mix_colors;
    switch (segment) {
      case 0: case 1: case 2: case 3: mix_colors = do_aa ? 1 /* MIX_AA */ : 2 /* MIX_NO_AA */; break;
      default: mix_colors = 0 /* DONT_MIX */; break;
    }
    vMixColors = ivec2_scalar(mix_colors, 0);
    vPos = size * aPosition;
    vColor0 = aColor0;
    vColor1 = aColor1;
    vec2_scalar ccs = outer + clip_sign * aRadii;
    vClipCenter_Sign = vec4_scalar(ccs.x, ccs.y, clip_sign.x, clip_sign.y);
    vec2_scalar inner = max(aRadii - aWidths, 0.0f);
    vClipRadii = vec4_scalar(aRadii.x, aRadii.y, inner.x, inner.y);
    vColorLine = vec4_scalar(outer.x, outer.y, aWidths.y * -clip_sign.y, aWidths.x * clip_sign.x);
    vec2_scalar horizontal_clip_sign = vec2_scalar(-clip_sign.x, clip_sign.y);
    vec2_scalar hc = aClipParams1.sel(X, Y) + horizontal_clip_sign * aClipParams1.sel(Z, W);
    vHorizontalClipCenter_Sign = vec4_scalar(hc.x, hc.y, horizontal_clip_sign.x, horizontal_clip_sign.y);
    vHorizontalClipRadii = aClipParams1.sel(Z, W);
    vec2_scalar vertical_clip_sign = vec2_scalar(clip_sign.x, -clip_sign.y);
    vec2_scalar vc = aClipParams2.sel(X, Y) + vertical_clip_sign * aClipParams2.sel(Z, W);
    vVerticalClipCenter_Sign = vec4_scalar(vc.x, vc.y, vertical_clip_sign.x, vertical_clip_sign.y);
    vVerticalClipRadii = aClipParams2.sel(Z, W);
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
  cs_border_solid_vert() {
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

struct cs_border_solid_frag : FragmentShaderImpl, cs_border_solid_vert {
  typedef cs_border_solid_frag Self;
  typedef cs_border_solid_vert::InterpOutputs InterpInputs;
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
  // ellipse.glsl:9-11, 31-45
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
  // shared.glsl:110-113, 184-189
  static Float distance_to_line(vec2_scalar p0, vec2_scalar perp_dir, vec2 p) {
    vec2 dir_to_p0 = p0 - p;
    return dot(vec2(normalize(perp_dir)), dir_to_p0);
  }
  static Float distance_aa(float aa_range, Float signed_distance) {
    Float dist = signed_distance * aa_range;
    return clamp(0.5f - dist, Float(0.0f), Float(1.0f));
  }
  static I32 in_region(vec4_scalar center_sign, vec2 rel) {
    return (center_sign.z * rel.x < 0.0f) & (center_sign.w * rel.y < 0.0f);
  }
  void main() {   // :132-177
    float aa_range = recip(fwidth(vPos).x);   // compute_aa_range, shared.glsl:145-148
    bool do_aa = vMixColors.x != 2 /* MIX_NO_AA */;
    Float mix_factor = 0.0f;
    if (vMixColors.x != 0 /* DONT_MIX */) {
      Float d_line = distance_to_line(vColorLine.sel(X, Y), vColorLine.sel(Z, W), vPos);
      if (do_aa) {
        mix_factor = distance_aa(aa_range, -d_line);
      } else {
        mix_factor = if_then_else(d_line + 0.0001f >= 0.0f, Float(1.0f), Float(0.0f));
      }
    }
    // main corner clip region
    vec2 clip_relative_pos = vPos - vClipCenter_Sign.sel(X, Y);
    I32 in_clip_region = in_region(vClipCenter_Sign, clip_relative_pos);
    Float d = -1.0f;
    {
      Float d_radii_a = distance_to_ellipse(clip_relative_pos, vClipRadii.sel(X, Y));
      Float d_radii_b = distance_to_ellipse(clip_relative_pos, vClipRadii.sel(Z, W));
      d = if_then_else(in_clip_region, max(d_radii_a, -d_radii_b), d);
    }
    // horizontally adjacent corner
    clip_relative_pos = vPos - vHorizontalClipCenter_Sign.sel(X, Y);
    in_clip_region = in_region(vHorizontalClipCenter_Sign, clip_relative_pos);
    {
      Float d_radii = distance_to_ellipse(clip_relative_pos, vHorizontalClipRadii);
      d = if_then_else(in_clip_region, max(d_radii, d), d);
    }
    // vertically adjacent corner
    clip_relative_pos = vPos - vVerticalClipCenter_Sign.sel(X, Y);
    in_clip_region = in_region(vVerticalClipCenter_Sign, clip_relative_pos);
    {
      Float d_radii = distance_to_ellipse(clip_relative_pos, vVerticalClipRadii);
      d = if_then_else(in_clip_region, max(d_radii, d), d);
    }
    Float alpha = do_aa ? distance_aa(aa_range, d) : Float(1.0f);
    vec4 color = mix(vec4(vColor0), vec4(vColor1), mix_factor);
    gl_FragColor = color * alpha;
  }
  WRSH_FRAG_ABI(Self)
  cs_border_solid_frag() {
    WRSH_FRAG_WIRING()
  }
};

WRSH_PROGRAM(cs_border_solid, "cs_border_solid")
