// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "cs_clip_box_shadow TEXTURE_2D"
// (webrender_build/src/shader_features.rs:69). Restates
// webrender/res/cs_clip_box_shadow.glsl:59-119 (VS), 123-138 (FS), 150-324 (the
// nine-patch span shader) on top of the clip_vert_common helpers in
// cs_clip_rectangle.h, with SWGL defined.

struct cs_clip_box_shadow_vert : wrsh::clip_vert_common {
  typedef cs_clip_box_shadow_vert Self;
  int a_res, a_src_size, a_mode, a_stretch, a_dest;
  ivec2_scalar aClipDataResourceAddress;
  vec2_scalar aClipSrcRectSize;
  int aClipMode;
  ivec2_scalar aStretchMode;
  vec4_scalar aClipDestRect;
  vec4 vLocalPos;
  vec2 vUv;
  vec4_scalar vUvBounds, vEdge, vUvBounds_NoClamp;
  vec2_scalar vClipMode;
  struct InterpOutputs {
    vec4_scalar vLocalPos;
    vec2_scalar vUv;
  };
  void main() {
    using namespace wrsh;
    Transform clip_transform = fetch_transform(aTransformIds.x);
    Transform prim_transform = fetch_transform(aTransformIds.y);
    // fetch_image_source_direct, gpu_cache.glsl:111-115
    vec4_scalar res0 = texelFetch(
        sGpuCache, ivec2_scalar(aClipDataResourceAddress.x, aClipDataResourceAddress.y), 0);
    RectWithEndpoint dest_rect{vec2_scalar(aClipDestRect.x, aClipDestRect.y),
                               vec2_scalar(aClipDestRect.z, aClipDestRect.w)};
    ClipVertexInfo vi = write_clip_tile_vertex(
        dest_rect, prim_transform, clip_transform,
        RectWithEndpoint{vec2_scalar(aClipDeviceArea.x, aClipDeviceArea.y),
                         vec2_scalar(aClipDeviceArea.z, aClipDeviceArea.w)},
        vec2_scalar(aClipOrigins.x, aClipOrigins.y),
        vec2_scalar(aClipOrigins.z, aClipOrigins.w), aDevicePixelScale);
    vClipMode.x = float(aClipMode);
    ivec2_scalar ts = textureSize(sColor0, 0);
    vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));
    vec2 local_pos = vec2(vi.local_pos.x / vi.local_pos.w, vi.local_pos.y / vi.local_pos.w);
    vLocalPos = vi.local_pos;
    vec2_scalar dest_rect_size = rect_size(dest_rect);
    switch (aStretchMode.x) {
      case 0: /* MODE_STRETCH */
        vEdge.x = 0.5f;
        vEdge.z = (dest_rect_size.x / aClipSrcRectSize.x) - 0.5f;
        vUv.x = (local_pos.x - dest_rect.p0.x) / aClipSrcRectSize.x;
        break;
      case 1: /* MODE_SIMPLE */
      default:
        vEdge.x = 1.0f;
        vEdge.z = 1.0f;
        vUv.x = (local_pos.x - dest_rect.p0.x) / dest_rect_size.x;
        break;
    }
    switch (aStretchMode.y) {
      case 0:
        vEdge.y = 0.5f;
        vEdge.w = (dest_rect_size.y / aClipSrcRectSize.y) - 0.5f;
        vUv.y = (local_pos.y - dest_rect.p0.y) / aClipSrcRectSize.y;
        break;
      case 1:
      default:
        vEdge.y = 1.0f;
        vEdge.w = 1.0f;
        vUv.y = (local_pos.y - dest_rect.p0.y) / dest_rect_size.y;
        break;
    }
    vUv.x *= vi.local_pos.w;
    vUv.y *= vi.local_pos.w;
    vec2_scalar uv0 = vec2_scalar(res0.x, res0.y);
    vec2_scalar uv1 = vec2_scalar(res0.z, res0.w);
    vUvBounds = vec4_scalar(uv0.x + 0.5f, uv0.y + 0.5f, uv1.x - 0.5f, uv1.y - 0.5f) /
                vec4_scalar(texture_size.x, texture_size.y, texture_size.x, texture_size.y);
    vUvBounds_NoClamp = vec4_scalar(uv0.x, uv0.y, uv1.x, uv1.y) /
                        vec4_scalar(texture_size.x, texture_size.y, texture_size.x, texture_size.y);
  }
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* va, uint32_t start,
                           int instance, int count) {
    Self* self = (Self*)impl;
    auto& L = self->attribs.locs;
    self->load_common(va, start, instance, count);
    load_flat_attrib(self->aClipDataResourceAddress, va[L[self->a_res]], start, instance, count);
    load_flat_attrib(self->aClipSrcRectSize, va[L[self->a_src_size]], start, instance, count);
    load_flat_attrib(self->aClipMode, va[L[self->a_mode]], start, instance, count);
    load_flat_attrib(self->aStretchMode, va[L[self->a_stretch]], start, instance, count);
    load_flat_attrib(self->aClipDestRect, va[L[self->a_dest]], start, instance, count);
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->vLocalPos = get_nth(vLocalPos, n);
      dest->vUv = get_nth(vUv, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  cs_clip_box_shadow_vert() {
    a_res = attribs.add("aClipDataResourceAddress");
    a_src_size = attribs.add("aClipSrcRectSize");
    a_mode = attribs.add("aClipMode");
    a_stretch = attribs.add("aStretchMode");
    a_dest = attribs.add("aClipDestRect");
    WRSH_VERT_WIRING(Self)
  }
};

struct cs_clip_box_shadow_frag : FragmentShaderImpl, cs_clip_box_shadow_vert {
  typedef cs_clip_box_shadow_frag Self;
  typedef cs_clip_box_shadow_vert::InterpOutputs InterpInputs;
  InterpInputs interp_step;
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_,
                                 const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    self->vLocalPos = init_interp(init->vLocalPos, step->vLocalPos);
    self->interp_step.vLocalPos = step->vLocalPos * 4.0f;
    self->vUv = init_interp(init->vUv, step->vUv);
    self->interp_step.vUv = step->vUv * 4.0f;
  }
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) {
    float chunks = steps * 0.25f;
    vLocalPos += interp_step.vLocalPos * chunks;
    vUv += interp_step.vUv * chunks;
  }
  // the per-fragment nine-patch mapping (:124-128 == :247-250)
  vec2 map_uv(vec2 uv_linear) const {
    vec2 uv = clamp(uv_linear, vec2_scalar(0.0f), vec2_scalar(vEdge.x, vEdge.y));
    uv += max(vec2_scalar(0.0f), uv_linear - vec2_scalar(vEdge.z, vEdge.w));
    uv = mix(vec2_scalar(vUvBounds_NoClamp.x, vUvBounds_NoClamp.y),
             vec2_scalar(vUvBounds_NoClamp.z, vUvBounds_NoClamp.w), uv);
    return uv;
  }
  Float shade(vec2 uv_linear, vec2 local_pos) const {
    vec2 uv = map_uv(uv_linear);
    uv = clamp(uv, vec2_scalar(vUvBounds.x, vUvBounds.y), vec2_scalar(vUvBounds.z, vUvBounds.w));
    // rectangle_aa_rough_fragment -> point_inside_rect (rect.glsl:15-18)
    vec2 s = step(vec2(vec2_scalar(vTransformBounds.x, vTransformBounds.y)), local_pos) -
             step(vec2(vec2_scalar(vTransformBounds.z, vTransformBounds.w)), local_pos);
    Float in_shadow_rect = s.x * s.y;
    Float texel = texture(sColor0, uv).x;
    Float alpha = mix(texel, 1.0f - texel, vClipMode.x);
    return mix(Float(vClipMode.x), alpha, in_shadow_rect);
  }
  // cs_clip_box_shadow.glsl:123-138
  void main() {
    vec2 uv_linear = vec2(vUv.x / vLocalPos.w, vUv.y / vLocalPos.w);
    vec2 local_pos = vec2(vLocalPos.x / vLocalPos.w, vLocalPos.y / vLocalPos.w);
    Float result = shade(uv_linear, local_pos);
    result = if_then_else(vLocalPos.w > 0.0f, result, Float(0.0f));
    gl_FragColor = vec4(result);
  }
  // cs_clip_box_shadow.glsl:150-324
  void swgl_drawSpanR8() {
    if (interp_step.vLocalPos.w != 0.0f) {
      return;
    }
    float w = vLocalPos.w.x;
    if (w <= 0.0f) {
      swgl_commitSolidR8(0.0f);
      return;
    }
    w = 1.0f / w;
    vec2 uv_linear = vec2(vUv.x * w, vUv.y * w);
    vec2_scalar uv_linear0 = vec2_scalar(uv_linear.x.x, uv_linear.y.x);
    vec2_scalar uv_linear_step = vec2_scalar(interp_step.vUv.x * w, interp_step.vUv.y * w);
    vec2 local_pos = vec2(vLocalPos.x * w, vLocalPos.y * w);
    vec2_scalar local_pos0 = vec2_scalar(local_pos.x.x, local_pos.y.x);
    vec2_scalar local_step = vec2_scalar(interp_step.vLocalPos.x * w, interp_step.vLocalPos.y * w);

    const vec4_scalar& tb = vTransformBounds;
    bool negx = local_step.x < 0.0f, negy = local_step.y < 0.0f;
    vec4_scalar clip_dist((negx ? tb.z : tb.x) - local_pos0.x, (negy ? tb.w : tb.y) - local_pos0.y,
                          (negx ? tb.x : tb.z) - local_pos0.x, (negy ? tb.y : tb.w) - local_pos0.y);
    float rsx = recip(local_step.x), rsy = recip(local_step.y);
    clip_dist = vec4_scalar(
        local_step.x != 0.0f ? clip_dist.x * rsx : 1.0e6f * step(0.0f, clip_dist.x),
        local_step.y != 0.0f ? clip_dist.y * rsy : 1.0e6f * step(0.0f, clip_dist.y),
        local_step.x != 0.0f ? clip_dist.z * rsx : 1.0e6f * step(0.0f, clip_dist.z),
        local_step.y != 0.0f ? clip_dist.w * rsy : 1.0e6f * step(0.0f, clip_dist.w));
    float shadow_start = max(clip_dist.x, clip_dist.y);
    float shadow_end = min(clip_dist.z, clip_dist.w);
    float sl = float(swgl_SpanLength), ss = float(swgl_StepSize);
    int shadow_start_len = int(clamp(sl - ss * floor(shadow_start), 0.0f, sl));
    int shadow_end_len = int(clamp(sl - ss * ceil(shadow_end), 0.0f, sl));

    bool ngx = uv_linear_step.x < 0.0f, ngy = uv_linear_step.y < 0.0f;
    vec4_scalar opaque_dist(
        (ngx ? vEdge.z : vEdge.x) - uv_linear0.x, (ngy ? vEdge.w : vEdge.y) - uv_linear0.y,
        (ngx ? vEdge.x : vEdge.z) - uv_linear0.x, (ngy ? vEdge.y : vEdge.w) - uv_linear0.y);
    float rux = recip(uv_linear_step.x), ruy = recip(uv_linear_step.y);
    opaque_dist = vec4_scalar(
        uv_linear_step.x != 0.0f ? opaque_dist.x * rux : 1.0e6f * step(0.0f, opaque_dist.x),
        uv_linear_step.y != 0.0f ? opaque_dist.y * ruy : 1.0e6f * step(0.0f, opaque_dist.y),
        uv_linear_step.x != 0.0f ? opaque_dist.z * rux : 1.0e6f * step(0.0f, opaque_dist.z),
        uv_linear_step.y != 0.0f ? opaque_dist.w * ruy : 1.0e6f * step(0.0f, opaque_dist.w));
    float sel = float(shadow_end_len);
    int opaque_steps[4] = {int(clamp(sl - ss * floor(opaque_dist.x), sel, sl)),
                           int(clamp(sl - ss * floor(opaque_dist.y), sel, sl)),
                           int(clamp(sl - ss * floor(opaque_dist.z), sel, sl)),
                           int(clamp(sl - ss * floor(opaque_dist.w), sel, sl))};

    if (swgl_SpanLength > shadow_start_len) {
      int num_before = swgl_SpanLength - shadow_start_len;
      swgl_commitPartialSolidR8(num_before, vClipMode.x);
      float steps_before = float(num_before / swgl_StepSize);
      uv_linear += steps_before * uv_linear_step;
      local_pos += steps_before * local_step;
    }

    while (swgl_SpanLength > 0) {
      {
        Float result = shade(uv_linear, local_pos);
        swgl_commitColorR8(result);
        uv_linear += uv_linear_step;
        local_pos += local_step;
      }
      if (swgl_SpanLength <= shadow_end_len) {
        break;
      }
      int num_inside = swgl_SpanLength - swgl_StepSize - shadow_end_len;
      vec4_scalar uv_bounds = vUvBounds;
      if (swgl_SpanLength >= opaque_steps[1]) {
        num_inside = min(num_inside, swgl_SpanLength - opaque_steps[1]);
      } else if (swgl_SpanLength >= opaque_steps[3]) {
        num_inside = min(num_inside, swgl_SpanLength - opaque_steps[3]);
        float c = clamp(mix(vUvBounds_NoClamp.y, vUvBounds_NoClamp.w, vEdge.y), vUvBounds.y,
                        vUvBounds.w);
        uv_bounds.y = c;
        uv_bounds.w = c;
      }
      if (swgl_SpanLength >= opaque_steps[0]) {
        num_inside = min(num_inside, swgl_SpanLength - opaque_steps[0]);
      } else if (swgl_SpanLength >= opaque_steps[2]) {
        num_inside = min(num_inside, swgl_SpanLength - opaque_steps[2]);
        float c = clamp(mix(vUvBounds_NoClamp.x, vUvBounds_NoClamp.z, vEdge.x), vUvBounds.x,
                        vUvBounds.z);
        uv_bounds.x = c;
        uv_bounds.z = c;
      }
      if (num_inside > 0) {
        vec2 uv = map_uv(uv_linear);
        if (uv_bounds.x == uv_bounds.z && uv_bounds.y == uv_bounds.w) {
          uv = clamp(uv, vec2_scalar(uv_bounds.x, uv_bounds.y),
                     vec2_scalar(uv_bounds.z, uv_bounds.w));
          Float texel = texture(sColor0, uv).x;
          Float alpha = mix(texel, 1.0f - texel, vClipMode.x);
          swgl_commitPartialSolidR8(num_inside, alpha);
        } else if (vClipMode.x != 0.0f) {
          swgl_commitPartialTextureLinearInvertR8(num_inside, sColor0, uv, uv_bounds);
        } else {
          swgl_commitPartialTextureLinearR8(num_inside, sColor0, uv, uv_bounds);
        }
        float steps_inside = float(num_inside / swgl_StepSize);
        uv_linear += steps_inside * uv_linear_step;
        local_pos += steps_inside * local_step;
      }
    }
    if (swgl_SpanLength > 0) {
      swgl_commitPartialSolidR8(swgl_SpanLength, vClipMode.x);
    }
  }
  WRSH_FRAG_ABI(Self)
  static int draw_span_R8(FragmentShaderImpl* impl) {
    Self* self = (Self*)impl;
    DISPATCH_DRAW_SPAN(self, R8);
  }
  cs_clip_box_shadow_frag() {
    WRSH_FRAG_WIRING()
    draw_span_R8_func = &draw_span_R8;
  }
};

WRSH_PROGRAM(cs_clip_box_shadow, "cs_clip_box_shadow TEXTURE_2D")
