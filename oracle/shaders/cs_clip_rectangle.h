// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "cs_clip_rectangle" and "cs_clip_rectangle FAST_PATH"
// (webrender_build/src/shader_features.rs:68). Restates
// webrender/res/cs_clip_rectangle.glsl:81-153 (VS), 170-199 (FS), 223-495 (the
// rounded-rectangle span rasteriser), clip_shared.glsl:14-78, transform.glsl:48-90
// (get_node_pos / untransform), ellipse.glsl:9-92, shared.glsl:118-189, with
// SWGL defined.  Per-pixel float expressions are evaluated lane by lane in the
// same operation order as the vector code glsl-to-cxx emits (strict IEEE, no
// contraction, so lane-wise scalar == vector).

namespace wrsh {

// clip_shared.glsl:14-78 + transform.glsl:48-90; shared by the cs_clip_* programs.
struct ClipVertexInfo {
  vec4 local_pos;
};

struct clip_vert_common : VertexShaderImpl, CommonState {
  AttribTable attribs;
  int a_aPosition, a_area, a_origins, a_dps, a_tids;
  vec2 aPosition;
  vec4_scalar aClipDeviceArea, aClipOrigins;
  float aDevicePixelScale;
  ivec2_scalar aTransformIds;
  vec4_scalar vTransformBounds;

  // transform.glsl:82-90 get_node_pos + :66-79 untransform + :51-61 ray_plane
  static vec4 get_node_pos(vec2 pos, const Transform& transform) {
    vec4_scalar ah = transform.m * vec4_scalar(0.0f, 0.0f, 0.0f, 1.0f);
    vec3_scalar a = vec3_scalar(ah.x / ah.w, ah.y / ah.w, ah.z / ah.w);
    vec3_scalar n = vec3_scalar(transform.inv_m[0].z, transform.inv_m[1].z,
                                transform.inv_m[2].z);
    Float px = pos.x, py = pos.y, pz = Float(-10000.0f);
    Float t = Float(0.0f);
    float denom = n.x * 0.0f + n.y * 0.0f + n.z * 1.0f;
    if (fabsf(denom) > 1e-6f) {
      Float dx = a.x - px, dy = a.y - py, dz = a.z - pz;
      t = (dx * n.x + dy * n.y + dz * n.z) / denom;
    }
    Float z = pz + 1.0f * t;
    return transform.inv_m * vec4(px, py, z, Float(1.0f));
  }

  ClipVertexInfo write_clip_tile_vertex(RectWithEndpoint local_clip_rect,
                                        const Transform& prim_transform,
                                        const Transform& clip_transform,
                                        RectWithEndpoint sub_rect,
                                        vec2_scalar task_origin,
                                        vec2_scalar screen_origin,
                                        float device_pixel_scale) {
    vec2 device_pos = screen_origin + mix(sub_rect.p0, sub_rect.p1, aPosition);
    vec2 world_pos = device_pos / device_pixel_scale;
    vec4 pos = prim_transform.m * vec4(world_pos, 0.0f, 1.0f);
    pos.x /= pos.w;
    pos.y /= pos.w;
    pos.z /= pos.w;
    vec4 p = get_node_pos(vec2(pos.x, pos.y), clip_transform);
    vec4 local_pos = p * pos.w;
    vec4 vertex_pos = vec4(task_origin + mix(sub_rect.p0, sub_rect.p1, aPosition),
                           0.0f, 1.0f);
    gl_Position = uTransform * vertex_pos;
    vTransformBounds =
        vec4_scalar(local_clip_rect.p0.x, local_clip_rect.p0.y,
                    local_clip_rect.p1.x, local_clip_rect.p1.y);
    return ClipVertexInfo{local_pos};
  }

  void load_common(VertexAttrib* va, uint32_t start, int instance, int count) {
    load_attrib(aPosition, va[attribs.locs[a_aPosition]], start, instance, count);
    load_flat_attrib(aClipDeviceArea, va[attribs.locs[a_area]], start, instance, count);
    load_flat_attrib(aClipOrigins, va[attribs.locs[a_origins]], start, instance, count);
    load_flat_attrib(aDevicePixelScale, va[attribs.locs[a_dps]], start, instance, count);
    load_flat_attrib(aTransformIds, va[attribs.locs[a_tids]], start, instance, count);
  }

  clip_vert_common() {
    used = (1u << U_sColor0) | (1u << U_sGpuCache) | (1u << U_sTransformPalette) |
           (1u << U_sRenderTasks) | (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_area = attribs.add("aClipDeviceArea");
    a_origins = attribs.add("aClipOrigins");
    a_dps = attribs.add("aDevicePixelScale");
    a_tids = attribs.add("aTransformIds");
  }
};

// shared.glsl:184-189
static ALWAYS_INLINE float distance_aa_1(float aa_range, float signed_distance) {
  float dist = signed_distance * aa_range;
  return clamp(0.5f - dist, 0.0f, 1.0f);
}
// ellipse.glsl:33-38
static ALWAYS_INLINE float distance_to_ellipse_approx_1(float px, float py,
                                                        float irx, float iry,
                                                        float scale) {
  float prx = px * irx, pry = py * iry;
  float g = (px * prx + py * pry) - scale;
  float dgx = (1.0f + scale) * prx, dgy = (1.0f + scale) * pry;
  return g * inversesqrt(dgx * dgx + dgy * dgy);
}
// rect.glsl:20-32
static ALWAYS_INLINE float signed_distance_rect_1(float px, float py, float p0x,
                                                  float p0y, float p1x, float p1y) {
  float dx = max(p0x - px, px - p1x), dy = max(p0y - py, py - p1y);
  return max(dx, dy);
}
// mix(x, y, a) with float a (glsl.h: (y - x) * a + x)
static ALWAYS_INLINE float mix_1(float x, float y, float a) { return (y - x) * a + x; }

}  // namespace wrsh

#define WRSH_CS_CLIP_RECTANGLE(NAME, KEYSTR, FAST_PATH)                        \
  struct NAME##_vert : wrsh::clip_vert_common {                                \
    typedef NAME##_vert Self;                                                  \
    int a_lpos, a_lrect, a_mode, a_rect[4], a_radii[4];                        \
    vec2_scalar aClipLocalPos;                                                 \
    vec4_scalar aClipLocalRect;                                                \
    float aClipMode;                                                           \
    vec4_scalar aClipRect[4], aClipRadii[4]; /* TL, TR, BL, BR */              \
    vec4 vLocalPos;                                                            \
    vec3_scalar vClipParams;                                                   \
    vec4_scalar vClipCenter_Radius[4]; /* TL, TR, BR, BL */                    \
    vec3_scalar vClipPlane[4];         /* TL, TR, BR, BL */                    \
    vec2_scalar vClipMode;                                                     \
    struct InterpOutputs {                                                     \
      vec4_scalar vLocalPos;                                                   \
    };                                                                         \
    static vec2_scalar inverse_radii_squared(vec2_scalar radii) {              \
      return vec2_scalar(1.0f / max(radii.x * radii.x, 1.0e-6f),               \
                         1.0f / max(radii.y * radii.y, 1.0e-6f));              \
    }                                                                          \
    void main() {                                                              \
      using namespace wrsh;                                                    \
      Transform clip_transform = fetch_transform(aTransformIds.x);             \
      Transform prim_transform = fetch_transform(aTransformIds.y);             \
      RectWithEndpoint local_rect{                                             \
          vec2_scalar(aClipLocalRect.x, aClipLocalRect.y),                     \
          vec2_scalar(aClipLocalRect.z, aClipLocalRect.w)};                    \
      vec2_scalar diff = aClipLocalPos - local_rect.p0;                        \
      local_rect.p0 = aClipLocalPos;                                           \
      local_rect.p1 += diff;                                                   \
      ClipVertexInfo vi = write_clip_tile_vertex(                              \
          local_rect, prim_transform, clip_transform,                          \
          RectWithEndpoint{vec2_scalar(aClipDeviceArea.x, aClipDeviceArea.y),  \
                           vec2_scalar(aClipDeviceArea.z, aClipDeviceArea.w)}, \
          vec2_scalar(aClipOrigins.x, aClipOrigins.y),                         \
          vec2_scalar(aClipOrigins.z, aClipOrigins.w), aDevicePixelScale);     \
      vClipMode.x = aClipMode;                                                 \
      vLocalPos = vi.local_pos;                                                \
      if (FAST_PATH) {                                                         \
        vec2_scalar half_size = 0.5f * rect_size(local_rect);                  \
        float radius = aClipRadii[0].x;                                        \
        vec2_scalar off = half_size + aClipLocalPos;                           \
        vLocalPos.x -= off.x * vi.local_pos.w;                                 \
        vLocalPos.y -= off.y * vi.local_pos.w;                                 \
        vClipParams = vec3_scalar(half_size.x - radius, half_size.y - radius,  \
                                  radius);                                     \
      } else {                                                                 \
        RectWithEndpoint clip_rect = local_rect;                               \
        vec2_scalar r_tl(aClipRadii[0].x, aClipRadii[0].y);                    \
        vec2_scalar r_tr(aClipRadii[1].x, aClipRadii[1].y);                    \
        vec2_scalar r_bl(aClipRadii[2].x, aClipRadii[2].y);                    \
        vec2_scalar r_br(aClipRadii[3].x, aClipRadii[3].y);                    \
        vec2_scalar i_tl = inverse_radii_squared(r_tl);                        \
        vec2_scalar i_tr = inverse_radii_squared(r_tr);                        \
        vec2_scalar i_br = inverse_radii_squared(r_br);                        \
        vec2_scalar i_bl = inverse_radii_squared(r_bl);                        \
        vClipCenter_Radius[0] = vec4_scalar(clip_rect.p0.x + r_tl.x,           \
                                            clip_rect.p0.y + r_tl.y, i_tl.x,   \
                                            i_tl.y);                           \
        vClipCenter_Radius[1] = vec4_scalar(clip_rect.p1.x - r_tr.x,           \
                                            clip_rect.p0.y + r_tr.y, i_tr.x,   \
                                            i_tr.y);                           \
        vClipCenter_Radius[2] = vec4_scalar(clip_rect.p1.x - r_br.x,           \
                                            clip_rect.p1.y - r_br.y, i_br.x,   \
                                            i_br.y);                           \
        vClipCenter_Radius[3] = vec4_scalar(clip_rect.p0.x + r_bl.x,           \
                                            clip_rect.p1.y - r_bl.y, i_bl.x,   \
                                            i_bl.y);                           \
        vec2_scalar n_tl(-r_tl.y, -r_tl.x);                                    \
        vec2_scalar n_tr(r_tr.y, -r_tr.x);                                     \
        vec2_scalar n_br(r_br.y, r_br.x);                                      \
        vec2_scalar n_bl(-r_bl.y, r_bl.x);                                     \
        vClipPlane[0] = vec3_scalar(                                           \
            n_tl.x, n_tl.y,                                                    \
            dot(n_tl, vec2_scalar(clip_rect.p0.x, clip_rect.p0.y + r_tl.y)));  \
        vClipPlane[1] = vec3_scalar(                                           \
            n_tr.x, n_tr.y,                                                    \
            dot(n_tr, vec2_scalar(clip_rect.p1.x - r_tr.x, clip_rect.p0.y)));  \
        vClipPlane[2] = vec3_scalar(                                           \
            n_br.x, n_br.y,                                                    \
            dot(n_br, vec2_scalar(clip_rect.p1.x, clip_rect.p1.y - r_br.y)));  \
        vClipPlane[3] = vec3_scalar(                                           \
            n_bl.x, n_bl.y,                                                    \
            dot(n_bl, vec2_scalar(clip_rect.p0.x + r_bl.x, clip_rect.p1.y)));  \
      }                                                                        \
    }                                                                          \
    static void load_attribs(VertexShaderImpl* impl, VertexAttrib* va,         \
                             uint32_t start, int instance, int count) {        \
      Self* self = (Self*)impl;                                                \
      auto& L = self->attribs.locs;                                            \
      self->load_common(va, start, instance, count);                           \
      load_flat_attrib(self->aClipLocalPos, va[L[self->a_lpos]], start,        \
                       instance, count);                                       \
      load_flat_attrib(self->aClipLocalRect, va[L[self->a_lrect]], start,      \
                       instance, count);                                       \
      load_flat_attrib(self->aClipMode, va[L[self->a_mode]], start, instance,  \
                       count);                                                 \
      for (int k = 0; k < 4; k++) {                                            \
        load_flat_attrib(self->aClipRect[k], va[L[self->a_rect[k]]], start,    \
                         instance, count);                                     \
        load_flat_attrib(self->aClipRadii[k], va[L[self->a_radii[k]]], start,  \
                         instance, count);                                     \
      }                                                                        \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {   \
      for (int n = 0; n < 4; n++) {                                            \
        auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);               \
        dest->vLocalPos = get_nth(vLocalPos, n);                               \
        dest_ptr += stride;                                                    \
      }                                                                        \
    }                                                                          \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() {                                                            \
      a_lpos = attribs.add("aClipLocalPos");                                   \
      a_lrect = attribs.add("aClipLocalRect");                                 \
      a_mode = attribs.add("aClipMode");                                       \
      static const char* const rn[4] = {"aClipRect_TL", "aClipRect_TR",        \
                                        "aClipRect_BL", "aClipRect_BR"};       \
      static const char* const dn[4] = {"aClipRadii_TL", "aClipRadii_TR",      \
                                        "aClipRadii_BL", "aClipRadii_BR"};     \
      for (int k = 0; k < 4; k++) {                                            \
        a_rect[k] = attribs.add(rn[k]);                                        \
        a_radii[k] = attribs.add(dn[k]);                                       \
      }                                                                        \
      WRSH_VERT_WIRING(Self)                                                   \
    }                                                                          \
  };                                                                           \
  struct NAME##_frag : FragmentShaderImpl, NAME##_vert {                       \
    typedef NAME##_frag Self;                                                  \
    typedef NAME##_vert::InterpOutputs InterpInputs;                           \
    InterpInputs interp_step;                                                  \
    static void read_interp_inputs(FragmentShaderImpl* impl,                   \
                                   const void* init_, const void* step_) {     \
      Self* self = (Self*)impl;                                                \
      const InterpInputs* init = (const InterpInputs*)init_;                   \
      const InterpInputs* step = (const InterpInputs*)step_;                   \
      self->vLocalPos = init_interp(init->vLocalPos, step->vLocalPos);         \
      self->interp_step.vLocalPos = step->vLocalPos * 4.0f;                    \
    }                                                                          \
    ALWAYS_INLINE void step_interp_inputs(int steps = 4) {                     \
      float chunks = steps * 0.25f;                                            \
      vLocalPos += interp_step.vLocalPos * chunks;                             \
    }                                                                          \
    /* sd_rounded_box / distance_to_rounded_rect for one lane */               \
    float rect_dist(float px, float py) const {                                \
      using namespace wrsh;                                                    \
      if (FAST_PATH) {                                                         \
        float dx = fabsf(px) - vClipParams.x, dy = fabsf(py) - vClipParams.y;  \
        float mx = max(dx, 0.0f), my = max(dy, 0.0f);                          \
        return (sqrt(mx * mx + my * my) + min(max(dx, dy), 0.0f)) -            \
               vClipParams.z;                                                  \
      }                                                                        \
      /* ellipse.glsl:50-92 */                                                 \
      float cx = 1.0e-6f, cy = 1.0e-6f, cz = 1.0f, cw = 1.0f;                  \
      const vec4_scalar& tl = vClipCenter_Radius[0];                           \
      const vec4_scalar& tr = vClipCenter_Radius[1];                           \
      const vec4_scalar& br = vClipCenter_Radius[2];                           \
      const vec4_scalar& bl = vClipCenter_Radius[3];                           \
      if (px * vClipPlane[0].x + py * vClipPlane[0].y > vClipPlane[0].z) {     \
        cx = tl.x - px; cy = tl.y - py; cz = tl.z; cw = tl.w;                  \
      }                                                                        \
      if (px * vClipPlane[1].x + py * vClipPlane[1].y > vClipPlane[1].z) {     \
        cx = (tr.x - px) * -1.0f; cy = (tr.y - py) * 1.0f; cz = tr.z; cw = tr.w; \
      }                                                                        \
      if (px * vClipPlane[2].x + py * vClipPlane[2].y > vClipPlane[2].z) {     \
        cx = px - br.x; cy = py - br.y; cz = br.z; cw = br.w;                  \
      }                                                                        \
      if (px * vClipPlane[3].x + py * vClipPlane[3].y > vClipPlane[3].z) {     \
        cx = (bl.x - px) * 1.0f; cy = (bl.y - py) * -1.0f; cz = bl.z; cw = bl.w; \
      }                                                                        \
      return max(distance_to_ellipse_approx_1(cx, cy, cz, cw, 1.0f),           \
                 signed_distance_rect_1(px, py, vTransformBounds.x,            \
                                        vTransformBounds.y, vTransformBounds.z, \
                                        vTransformBounds.w));                  \
    }                                                                          \
    /* cs_clip_rectangle.glsl:170-199 */                                       \
    void main() {                                                              \
      using namespace wrsh;                                                    \
      vec2 local_pos = vec2(vLocalPos.x / vLocalPos.w, vLocalPos.y / vLocalPos.w); \
      float aa_range = recip(fwidth(local_pos).x);                             \
      float out[4];                                                            \
      for (int n = 0; n < 4; n++) {                                            \
        float dist = rect_dist(get_nth(local_pos.x, n), get_nth(local_pos.y, n)); \
        float alpha = distance_aa_1(aa_range, dist);                           \
        float final_alpha = mix_1(alpha, 1.0f - alpha, vClipMode.x);           \
        out[n] = get_nth(vLocalPos.w, n) > 0.0f ? final_alpha : 0.0f;          \
      }                                                                        \
      gl_FragColor = vec4(Float{out[0], out[1], out[2], out[3]}, Float(0.0f),  \
                          Float(0.0f), Float(1.0f));                           \
    }                                                                          \
    Float aa_chunk(vec2 lp, float aa_range, bool use_plane,                    \
                   const vec3_scalar& plane, const vec4_scalar& corner) const { \
      using namespace wrsh;                                                    \
      float out[4];                                                            \
      for (int n = 0; n < 4; n++) {                                            \
        float px = get_nth(lp.x, n), py = get_nth(lp.y, n);                    \
        float d;                                                               \
        if (FAST_PATH) {                                                       \
          d = rect_dist(px, py);                                               \
        } else {                                                               \
          float rect = signed_distance_rect_1(px, py, vTransformBounds.x,      \
                                              vTransformBounds.y,              \
                                              vTransformBounds.z,              \
                                              vTransformBounds.w);             \
          if (use_plane && px * plane.x + py * plane.y > plane.z) {            \
            d = distance_to_ellipse_approx_1(px - corner.x, py - corner.y,     \
                                             corner.z, corner.w, 1.0f);        \
          } else {                                                             \
            d = rect;                                                          \
          }                                                                    \
        }                                                                      \
        float alpha = distance_aa_1(aa_range, d);                              \
        out[n] = mix_1(alpha, 1.0f - alpha, vClipMode.x);                      \
      }                                                                        \
      return Float{out[0], out[1], out[2], out[3]};                            \
    }                                                                          \
    /* cs_clip_rectangle.glsl:223-495 */                                       \
    void swgl_drawSpanR8() {                                                   \
      using namespace wrsh;                                                    \
      if (interp_step.vLocalPos.w != 0.0f) {                                   \
        return;                                                                \
      }                                                                        \
      float w = vLocalPos.w.x;                                                 \
      if (w <= 0.0f) {                                                         \
        swgl_commitSolidR8(0.0f);                                              \
        return;                                                                \
      }                                                                        \
      w = 1.0f / w;                                                            \
      vec2 local_pos = vec2(vLocalPos.x * w, vLocalPos.y * w);                 \
      vec2_scalar local_pos0 = vec2_scalar(local_pos.x.x, local_pos.y.x);      \
      vec2_scalar local_step = vec2_scalar(interp_step.vLocalPos.x * w,        \
                                           interp_step.vLocalPos.y * w);       \
      float step_scale = max(dot(local_step, local_step), 1.0e-6f);            \
      float aa_range = recip(fwidth(local_pos).x);                             \
      float aa_margin = inversesqrt(aa_range * aa_range * step_scale);         \
      vec4_scalar clip_rect;                                                   \
      if (FAST_PATH) {                                                         \
        clip_rect = vec4_scalar(-vClipParams.x - vClipParams.z,                \
                                -vClipParams.y - vClipParams.z,                \
                                vClipParams.x + vClipParams.z,                 \
                                vClipParams.y + vClipParams.z);                \
      } else {                                                                 \
        clip_rect = vTransformBounds;                                          \
      }                                                                        \
      bool negx = local_step.x < 0.0f, negy = local_step.y < 0.0f;             \
      vec4_scalar clip_dist(                                                   \
          (negx ? clip_rect.z : clip_rect.x) - local_pos0.x,                   \
          (negy ? clip_rect.w : clip_rect.y) - local_pos0.y,                   \
          (negx ? clip_rect.x : clip_rect.z) - local_pos0.x,                   \
          (negy ? clip_rect.y : clip_rect.w) - local_pos0.y);                  \
      float rsx = recip(local_step.x), rsy = recip(local_step.y);              \
      clip_dist = vec4_scalar(                                                 \
          local_step.x != 0.0f ? clip_dist.x * rsx                             \
                               : 1.0e6f * step(0.0f, clip_dist.x),             \
          local_step.y != 0.0f ? clip_dist.y * rsy                             \
                               : 1.0e6f * step(0.0f, clip_dist.y),             \
          local_step.x != 0.0f ? clip_dist.z * rsx                             \
                               : 1.0e6f * step(0.0f, clip_dist.z),             \
          local_step.y != 0.0f ? clip_dist.w * rsy                             \
                               : 1.0e6f * step(0.0f, clip_dist.w));            \
      float opaque_start = max(clip_dist.x, clip_dist.y);                      \
      float opaque_end = min(clip_dist.z, clip_dist.w);                        \
      float aa_start = opaque_start;                                           \
      float aa_end = opaque_end;                                               \
      vec3_scalar start_plane = vec3_scalar(1.0e6f);                           \
      vec3_scalar end_plane = vec3_scalar(1.0e6f);                             \
      vec4_scalar start_corner = vec4_scalar(1.0e6f, 1.0e6f, 1.0f, 1.0f);      \
      vec4_scalar end_corner = vec4_scalar(1.0e6f, 1.0e6f, 1.0f, 1.0f);        \
      vec3_scalar planes[4];                                                   \
      vec4_scalar infos[4];                                                    \
      if (FAST_PATH) {                                                         \
        float offset = (vClipParams.x + vClipParams.y + vClipParams.z) *       \
                       vClipParams.z;                                          \
        planes[0] = vec3_scalar(-vClipParams.z, -vClipParams.z, offset);       \
        planes[1] = vec3_scalar(vClipParams.z, -vClipParams.z, offset);        \
        planes[2] = vec3_scalar(vClipParams.z, vClipParams.z, offset);         \
        planes[3] = vec3_scalar(-vClipParams.z, vClipParams.z, offset);        \
        for (int k = 0; k < 4; k++) infos[k] = vec4_scalar(0.0f);              \
      } else {                                                                 \
        for (int k = 0; k < 4; k++) {                                          \
          planes[k] = vClipPlane[k];                                           \
          infos[k] = vClipCenter_Radius[k];                                    \
        }                                                                      \
      }                                                                        \
      for (int k = 0; k < 4; k++) { /* CLIP_CORNER, in TL, TR, BR, BL order */ \
        const vec3_scalar& plane = planes[k];                                  \
        float dist = (local_pos0.x * plane.x + local_pos0.y * plane.y) - plane.z; \
        float scale = -(local_step.x * plane.x + local_step.y * plane.y);      \
        if (scale >= 0.0f) {                                                   \
          if (dist > opaque_start * scale) {                                   \
            start_corner = infos[k];                                           \
            start_plane = plane;                                               \
            float inv_scale = recip(max(scale, 1.0e-6f));                      \
            opaque_start = dist * inv_scale;                                   \
            float apex = (0.7071f - 0.5f) * 2.0f * fabsf(plane.x * plane.y);   \
            aa_start = opaque_start - apex * inv_scale;                        \
          }                                                                    \
        } else if (dist > opaque_end * scale) {                                \
          end_corner = infos[k];                                               \
          end_plane = plane;                                                   \
          float inv_scale = recip(min(scale, -1.0e-6f));                       \
          opaque_end = dist * inv_scale;                                       \
          float apex = (0.7071f - 0.5f) * 2.0f * fabsf(plane.x * plane.y);     \
          aa_end = opaque_end - apex * inv_scale;                              \
        }                                                                      \
      }                                                                        \
      aa_margin = max(aa_margin - max(aa_start - aa_end, 0.0f), 0.0f);         \
      aa_start -= aa_margin;                                                   \
      aa_end += aa_margin;                                                     \
      float sl = float(swgl_SpanLength), ss = float(swgl_StepSize);            \
      int aa_start_len = int(clamp(sl - ss * floor(aa_start), 0.0f, sl));      \
      int opaque_start_len = int(clamp(sl - ss * ceil(opaque_start), 0.0f, sl)); \
      int opaque_end_len = int(clamp(sl - ss * floor(opaque_end), 0.0f, sl));  \
      int aa_end_len = int(clamp(sl - ss * ceil(aa_end), 0.0f, sl));           \
      if (swgl_SpanLength > aa_start_len) {                                    \
        int num_aa = swgl_SpanLength - aa_start_len;                           \
        swgl_commitPartialSolidR8(num_aa, vClipMode.x);                        \
        local_pos += float(num_aa / swgl_StepSize) * local_step;               \
      }                                                                        \
      if (!(FAST_PATH) && start_plane.x < 1.0e5f) {                            \
        while (swgl_SpanLength > opaque_start_len) {                           \
          swgl_commitColorR8(aa_chunk(local_pos, aa_range, true, start_plane,  \
                                      start_corner));                          \
          local_pos += local_step;                                             \
        }                                                                      \
      }                                                                        \
      while (swgl_SpanLength > opaque_start_len) {                             \
        swgl_commitColorR8(aa_chunk(local_pos, aa_range, false, start_plane,   \
                                    start_corner));                            \
        local_pos += local_step;                                               \
      }                                                                        \
      if (swgl_SpanLength > opaque_end_len) {                                  \
        int num_opaque = swgl_SpanLength - opaque_end_len;                     \
        swgl_commitPartialSolidR8(num_opaque, 1.0f - vClipMode.x);             \
        local_pos += float(num_opaque / swgl_StepSize) * local_step;           \
      }                                                                        \
      if (!(FAST_PATH) && end_plane.x < 1.0e5f) {                              \
        while (swgl_SpanLength > aa_end_len) {                                 \
          swgl_commitColorR8(aa_chunk(local_pos, aa_range, true, end_plane,    \
                                      end_corner));                            \
          local_pos += local_step;                                             \
        }                                                                      \
      }                                                                        \
      while (swgl_SpanLength > aa_end_len) {                                   \
        swgl_commitColorR8(aa_chunk(local_pos, aa_range, false, end_plane,     \
                                    end_corner));                              \
        local_pos += local_step;                                               \
      }                                                                        \
      if (swgl_SpanLength > 0) {                                               \
        swgl_commitPartialSolidR8(swgl_SpanLength, vClipMode.x);               \
      }                                                                        \
    }                                                                          \
    WRSH_FRAG_ABI(Self)                                                        \
    static int draw_span_R8(FragmentShaderImpl* impl) {                        \
      Self* self = (Self*)impl;                                                \
      DISPATCH_DRAW_SPAN(self, R8);                                            \
    }                                                                          \
    NAME##_frag() {                                                            \
      WRSH_FRAG_WIRING()                                                       \
      draw_span_R8_func = &draw_span_R8;                                       \
    }                                                                          \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_CS_CLIP_RECTANGLE(cs_clip_rectangle, "cs_clip_rectangle", false)
WRSH_CS_CLIP_RECTANGLE(cs_clip_rectangle_FAST_PATH, "cs_clip_rectangle FAST_PATH", true)
