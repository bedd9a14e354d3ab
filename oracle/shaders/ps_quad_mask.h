// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers of shader keys
// "ps_quad_mask" and "ps_quad_mask FAST_PATH" (webrender_build/src/shader_features.rs:232-240).
// Vertex stage: webrender/res/ps_quad.glsl:164-384 exactly as restated in ps_quad_textured.h
// (generated from that file by the round-1 tooling, see git history) with pattern_vertex of
// ps_quad_mask.glsl:65-152; fragment stage ps_quad.glsl:399-415 + ps_quad_mask.glsl:154-200,
// ellipse.glsl:33-92.  No swgl_drawSpan*: every pixel runs main().

#define WRSH_PS_QUAD_MASK(NAME, KEYSTR, FAST_PATH) \
struct NAME##_vert : VertexShaderImpl, wrsh::CommonState { \
  typedef NAME##_vert Self; \
  wrsh::AttribTable attribs; \
  int a_aPosition, a_aData, a_aClipData; \
 \
  /* attributes */ \
  vec2 aPosition; \
  ivec4_scalar aData; \
  ivec4_scalar aClipData; \
  /* flat varyings */ \
  vec4_scalar v_color; \
  ivec4_scalar v_flags; \
  /* ps_quad_mask.glsl:9-23 */ \
  vec3_scalar v_clip_params; \
  vec4_scalar vClipCenter_Radius_TL, vClipCenter_Radius_TR, vClipCenter_Radius_BR, vClipCenter_Radius_BL; \
  vec4_scalar vClipPlane_A, vClipPlane_B, vClipPlane_C; \
  vec4_scalar vTransformBounds; \
  vec2_scalar vClipMode; \
  /* varyings */ \
  vec4 vClipLocalPos; \
 \
  struct InterpOutputs { \
    vec4_scalar vClipLocalPos; \
  }; \
  static vec2_scalar inverse_radii_squared(vec2_scalar radii) { /* ellipse.glsl:33-35 */ \
    return vec2_scalar(1.0f) / max(radii * radii, vec2_scalar(1.0e-6f)); \
  } \
 \
  static constexpr int EDGE_AA_LEFT = 1, EDGE_AA_TOP = 2, EDGE_AA_RIGHT = 4, \
                       EDGE_AA_BOTTOM = 8; \
  static constexpr int PART_CENTER = 0, PART_LEFT = 1, PART_TOP = 2, \
                       PART_RIGHT = 3, PART_BOTTOM = 4, PART_ALL = 5; \
  static constexpr int QF_IS_OPAQUE = 1, QF_APPLY_DEVICE_CLIP = 2, \
                       QF_IGNORE_DEVICE_SCALE = 4, QF_USE_AA_SEGMENTS = 8, \
                       QF_IS_MASK = 16; \
  static constexpr int INVALID_SEGMENT_INDEX = 0xff; \
 \
  static float edge_aa_offset(int edge, int flags) { \
    return ((flags & edge) != 0) ? 2.0f : 0.0f; \
  } \
  static vec2_scalar so_map_point(vec4_scalar so, vec2_scalar p) { \
    return p * vec2_scalar(so.x, so.y) + vec2_scalar(so.z, so.w); \
  } \
  static vec2 so_map_point(vec4_scalar so, vec2 p) { \
    return p * vec2_scalar(so.x, so.y) + vec2_scalar(so.z, so.w); \
  } \
  static wrsh::RectWithEndpoint so_map_rect(vec4_scalar so, \
                                            wrsh::RectWithEndpoint r) { \
    return wrsh::RectWithEndpoint{so_map_point(so, r.p0), \
                                  so_map_point(so, r.p1)}; \
  } \
 \
  void main() { \
    using namespace wrsh; \
    /* decode_instance, ps_quad.glsl:164-178 */ \
    int prim_address_i = aData.x; \
    int prim_address_f = aData.y; \
    int quad_flags = (aData.z >> 24) & 0xff; \
    int edge_flags = (aData.z >> 16) & 0xff; \
    int part_index = (aData.z >> 8) & 0xff; \
    int segment_index = (aData.z >> 0) & 0xff; \
    int picture_task_address = aData.w; \
 \
    /* fetch_header, ps_quad.glsl:133-143 */ \
    ivec4_scalar header = fetch_from_gpu_buffer_1i(prim_address_i); \
    int transform_id = header.x; \
    int z_id = header.y; \
 \
    Transform transform = fetch_transform(transform_id); \
    PictureTask task = fetch_picture_task(picture_task_address); \
 \
    /* fetch_primitive, ps_quad.glsl:112-124 */ \
    vec4_scalar t0 = fetch_from_gpu_buffer_f(prim_address_f, 0); \
    vec4_scalar t1 = fetch_from_gpu_buffer_f(prim_address_f, 1); \
    vec4_scalar t2 = fetch_from_gpu_buffer_f(prim_address_f, 2); \
    vec4_scalar pattern_scale_offset = fetch_from_gpu_buffer_f(prim_address_f, 3); \
    vec4_scalar prim_color = fetch_from_gpu_buffer_f(prim_address_f, 4); \
    RectWithEndpoint prim_bounds{vec2_scalar(t0.x, t0.y), vec2_scalar(t0.z, t0.w)}; \
    RectWithEndpoint prim_clip{vec2_scalar(t1.x, t1.y), vec2_scalar(t1.z, t1.w)}; \
    RectWithEndpoint prim_uv_rect{vec2_scalar(t2.x, t2.y), vec2_scalar(t2.z, t2.w)}; \
    float z = float(z_id); \
 \
    RectWithEndpoint seg_rect, seg_uv_rect; \
    if (segment_index == INVALID_SEGMENT_INDEX) { \
      seg_rect = prim_bounds; \
      seg_uv_rect = prim_uv_rect; \
    } else { \
      /* fetch_segment, ps_quad.glsl:99-110 */ \
      int base = prim_address_f + 5 + segment_index * 2; \
      vec4_scalar s0 = fetch_from_gpu_buffer_f(base, 0); \
      vec4_scalar s1 = fetch_from_gpu_buffer_f(base, 1); \
      seg_rect = RectWithEndpoint{vec2_scalar(s0.x, s0.y), vec2_scalar(s0.z, s0.w)}; \
      seg_uv_rect = RectWithEndpoint{vec2_scalar(s1.x, s1.y), vec2_scalar(s1.z, s1.w)}; \
    } \
 \
    RectWithEndpoint lcr = seg_rect; \
    lcr.p0 = max(lcr.p0, prim_clip.p0); \
    lcr.p1 = min(lcr.p1, prim_clip.p1); \
    lcr.p1 = max(lcr.p0, lcr.p1); \
 \
    /* ps_quad.glsl:267-325 with SWGL_ANTIALIAS */ \
    switch (part_index) { \
      case PART_LEFT: \
        lcr.p1.x = lcr.p0.x + 2.0f; \
        swgl_antiAlias(EDGE_AA_LEFT); \
        break; \
      case PART_TOP: \
        lcr.p0.x = lcr.p0.x + 2.0f; \
        lcr.p1.x = lcr.p1.x - 2.0f; \
        lcr.p1.y = lcr.p0.y + 2.0f; \
        swgl_antiAlias(EDGE_AA_TOP); \
        break; \
      case PART_RIGHT: \
        lcr.p0.x = lcr.p1.x - 2.0f; \
        swgl_antiAlias(EDGE_AA_RIGHT); \
        break; \
      case PART_BOTTOM: \
        lcr.p0.x = lcr.p0.x + 2.0f; \
        lcr.p1.x = lcr.p1.x - 2.0f; \
        lcr.p0.y = lcr.p1.y - 2.0f; \
        swgl_antiAlias(EDGE_AA_BOTTOM); \
        break; \
      case PART_CENTER: \
        lcr.p0.x += edge_aa_offset(EDGE_AA_LEFT, edge_flags); \
        lcr.p1.x -= edge_aa_offset(EDGE_AA_RIGHT, edge_flags); \
        lcr.p0.y += edge_aa_offset(EDGE_AA_TOP, edge_flags); \
        lcr.p1.y -= edge_aa_offset(EDGE_AA_BOTTOM, edge_flags); \
        break; \
      case PART_ALL: \
      default: \
        swgl_antiAlias(edge_flags); \
        break; \
    } \
 \
    vec2 local_pos = mix(lcr.p0, lcr.p1, aPosition); \
 \
    float device_pixel_scale = task.device_pixel_scale; \
    if ((quad_flags & QF_IGNORE_DEVICE_SCALE) != 0) { \
      device_pixel_scale = 1.0f; \
    } \
 \
    /* write_vertex, ps_quad.glsl:184-221 */ \
    vec4 world_pos = transform.m * vec4(local_pos, 0.0f, 1.0f); \
    vec2 device_pos = world_pos.sel(X, Y) * device_pixel_scale; \
    vec2 vi_local_pos; \
    if ((quad_flags & QF_APPLY_DEVICE_CLIP) != 0) { \
      RectWithEndpoint device_clip_rect{ \
          task.content_origin, \
          task.content_origin + task.task_rect.p1 - task.task_rect.p0}; \
      device_pos = rect_clamp(device_clip_rect, device_pos); \
      vi_local_pos = \
          (transform.inv_m * vec4(device_pos / device_pixel_scale, 0.0f, 1.0f)) \
              .sel(X, Y); \
    } else { \
      vi_local_pos = local_pos; \
    } \
    vec2_scalar final_offset = -task.content_origin + task.task_rect.p0; \
    gl_Position = \
        uTransform * vec4(device_pos + final_offset * world_pos.w, \
                          z * world_pos.w, world_pos.w); \
 \
    v_color = prim_color; \
 \
    vec4_scalar pattern_tx = pattern_scale_offset; \
    seg_rect = so_map_rect(pattern_tx, seg_rect); \
    vec2 info_local_pos = so_map_point(pattern_tx, vi_local_pos); \
 \
    /* main(), ps_quad.glsl:373-384 */ \
    if ((quad_flags & QF_IS_MASK) != 0) { \
      v_flags.z = 1; \
    } else { \
      v_flags.z = 0; \
    } \
 \
    /* pattern_vertex, ps_quad_mask.glsl:65-152 (prim_info.local_pos = info_local_pos, */ \
    /* prim_info.local_clip_rect = scale_offset_map_rect(pattern_tx, prim.clip), ps_quad.glsl:352-360) */ \
    RectWithEndpoint info_clip_rect = so_map_rect(pattern_tx, prim_clip); \
    /* fetch_clip, ps_quad_mask.glsl:43-63 */ \
    RectWithEndpoint clip_rect; \
    vec4_scalar radii, radii_top, radii_bottom; \
    float clip_mode; \
    int clip_space = aClipData.z; \
    { \
      vec4_scalar c0 = fetch_from_gpu_buffer_f(aClipData.y, 0); \
      clip_rect = RectWithEndpoint{vec2_scalar(c0.x, c0.y), vec2_scalar(c0.z, c0.w)}; \
      if (FAST_PATH) { \
        radii = fetch_from_gpu_buffer_f(aClipData.y, 1); \
        clip_mode = fetch_from_gpu_buffer_f(aClipData.y, 2).x; \
      } else { \
        radii_top = fetch_from_gpu_buffer_f(aClipData.y, 1); \
        radii_bottom = fetch_from_gpu_buffer_f(aClipData.y, 2); \
        clip_mode = fetch_from_gpu_buffer_f(aClipData.y, 3).x; \
      } \
    } \
    Transform clip_transform = fetch_transform(aClipData.x); \
    vClipLocalPos = clip_transform.m * vec4(info_local_pos, 0.0f, 1.0f); \
    if (!(FAST_PATH)) { \
      if (clip_space == 0) { /* CLIP_SPACE_RASTER */ \
        vTransformBounds = vec4_scalar(clip_rect.p0.x, clip_rect.p0.y, clip_rect.p1.x, clip_rect.p1.y); \
      } else { \
        vec2_scalar xb0 = max(clip_rect.p0, info_clip_rect.p0); \
        vec2_scalar xb1 = min(clip_rect.p1, info_clip_rect.p1); \
        vTransformBounds = vec4_scalar(xb0.x, xb0.y, xb1.x, xb1.y); \
      } \
    } \
    vClipMode.x = clip_mode; \
    if (FAST_PATH) { \
      vec2_scalar half_size = 0.5f * (clip_rect.p1 - clip_rect.p0); \
      float radius = radii.x; \
      vec2 adj = vClipLocalPos.sel(X, Y) - (half_size + clip_rect.p0) * vClipLocalPos.w; \
      vClipLocalPos.x = adj.x; \
      vClipLocalPos.y = adj.y; \
      vec2_scalar hs = half_size - vec2_scalar(radius); \
      v_clip_params = vec3_scalar(hs.x, hs.y, radius); \
    } else { \
      vec2_scalar r_tl = vec2_scalar(radii_top.x, radii_top.y); \
      vec2_scalar r_tr = vec2_scalar(radii_top.z, radii_top.w); \
      vec2_scalar r_br = vec2_scalar(radii_bottom.z, radii_bottom.w); \
      vec2_scalar r_bl = vec2_scalar(radii_bottom.x, radii_bottom.y); \
      vec2_scalar i_tl = inverse_radii_squared(r_tl), i_tr = inverse_radii_squared(r_tr); \
      vec2_scalar i_br = inverse_radii_squared(r_br), i_bl = inverse_radii_squared(r_bl); \
      vClipCenter_Radius_TL = vec4_scalar(clip_rect.p0.x + r_tl.x, clip_rect.p0.y + r_tl.y, i_tl.x, i_tl.y); \
      vClipCenter_Radius_TR = vec4_scalar(clip_rect.p1.x - r_tr.x, clip_rect.p0.y + r_tr.y, i_tr.x, i_tr.y); \
      vClipCenter_Radius_BR = vec4_scalar(clip_rect.p1.x - r_br.x, clip_rect.p1.y - r_br.y, i_br.x, i_br.y); \
      vClipCenter_Radius_BL = vec4_scalar(clip_rect.p0.x + r_bl.x, clip_rect.p1.y - r_bl.y, i_bl.x, i_bl.y); \
      vec2_scalar n_tl = vec2_scalar(-r_tl.y, -r_tl.x); \
      vec2_scalar n_tr = vec2_scalar(r_tr.y, -r_tr.x); \
      vec2_scalar n_br = vec2_scalar(r_br.y, r_br.x); \
      vec2_scalar n_bl = vec2_scalar(-r_bl.y, r_bl.x); \
      vec3_scalar tl = vec3_scalar(n_tl.x, n_tl.y, dot(n_tl, vec2_scalar(clip_rect.p0.x, clip_rect.p0.y + r_tl.y))); \
      vec3_scalar tr = vec3_scalar(n_tr.x, n_tr.y, dot(n_tr, vec2_scalar(clip_rect.p1.x - r_tr.x, clip_rect.p0.y))); \
      vec3_scalar br = vec3_scalar(n_br.x, n_br.y, dot(n_br, vec2_scalar(clip_rect.p1.x, clip_rect.p1.y - r_br.y))); \
      vec3_scalar bl = vec3_scalar(n_bl.x, n_bl.y, dot(n_bl, vec2_scalar(clip_rect.p0.x + r_bl.x, clip_rect.p1.y))); \
      vClipPlane_A = vec4_scalar(tl.x, tl.y, tl.z, tr.x); \
      vClipPlane_B = vec4_scalar(tr.y, tr.z, br.x, br.y); \
      vClipPlane_C = vec4_scalar(br.z, bl.x, bl.y, bl.z); \
    } \
  } \
 \
  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs, \
                           uint32_t start, int instance, int count) { \
    Self* self = (Self*)impl; \
    load_attrib(self->aPosition, attribs[self->attribs.locs[self->a_aPosition]], \
                start, instance, count); \
    load_flat_attrib(self->aData, attribs[self->attribs.locs[self->a_aData]], \
                     start, instance, count); \
    load_flat_attrib(self->aClipData, attribs[self->attribs.locs[self->a_aClipData]], \
                     start, instance, count); \
  } \
 \
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) { \
    for (int n = 0; n < 4; n++) { \
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr); \
      dest->vClipLocalPos = get_nth(vClipLocalPos, n); \
      dest_ptr += stride; \
    } \
  } \
 \
  WRSH_VERT_ABI(Self) \
 \
  NAME##_vert() { \
    using namespace wrsh; \
    used = (1u << U_sTransformPalette) | \
           (1u << U_sRenderTasks) | (1u << U_sGpuBufferF) | \
           (1u << U_sGpuBufferI) | (1u << U_uTransform); \
    a_aPosition = attribs.add("aPosition"); \
    a_aData = attribs.add("aData"); \
    a_aClipData = attribs.add("aClipData"); \
    v_flags = ivec4_scalar(0, 0, 0, 0); \
    WRSH_VERT_WIRING(Self) \
  } \
}; \
 \
struct NAME##_frag : FragmentShaderImpl, NAME##_vert { \
  typedef NAME##_frag Self; \
  typedef NAME##_vert::InterpOutputs InterpInputs; \
  InterpInputs interp_step; \
 \
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_, \
                                 const void* step_) { \
    Self* self = (Self*)impl; \
    const InterpInputs* init = (const InterpInputs*)init_; \
    const InterpInputs* step = (const InterpInputs*)step_; \
    self->vClipLocalPos = init_interp(init->vClipLocalPos, step->vClipLocalPos); \
    self->interp_step.vClipLocalPos = step->vClipLocalPos * 4.0f; \
  } \
 \
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) { \
    float chunks = steps * 0.25f; \
    vClipLocalPos += interp_step.vClipLocalPos * chunks; \
  } \
 \
  /* ps_quad_mask.glsl:156-165 */ \
  static Float sd_box(vec2 pos, vec2_scalar box_size) { \
    vec2 d = abs(pos) - box_size; \
    return length(max(d, vec2(Float(0.0f)))) + min(max(d.x, d.y), Float(0.0f)); \
  } \
  /* ellipse.glsl:37-92 (as cs_clip_rectangle.h) */ \
  static Float distance_to_ellipse_approx(vec2 p, vec2 inv_radii_sq, Float scale) { \
    vec2 p_r = p * inv_radii_sq; \
    Float g = dot(p, p_r) - scale; \
    vec2 dG = (1.0f + scale) * p_r; \
    return g * inversesqrt(dot(dG, dG)); \
  } \
  Float distance_to_rounded_rect(vec2 pos, vec3_scalar plane_tl, vec4_scalar center_radius_tl, vec3_scalar plane_tr, \
                                 vec4_scalar center_radius_tr, vec3_scalar plane_br, vec4_scalar center_radius_br, \
                                 vec3_scalar plane_bl, vec4_scalar center_radius_bl, vec4_scalar rect_bounds) { \
    vec4 corner = vec4(vec2(Float(1.0e-6f)), vec2(Float(1.0f))); \
    vec4 c_tl = vec4(vec2_scalar(center_radius_tl.x, center_radius_tl.y) - pos, vec2(vec2_scalar(center_radius_tl.z, center_radius_tl.w))); \
    vec4 c_tr = vec4((vec2_scalar(center_radius_tr.x, center_radius_tr.y) - pos) * vec2_scalar(-1.0f, 1.0f), vec2(vec2_scalar(center_radius_tr.z, center_radius_tr.w))); \
    vec4 c_br = vec4(pos - vec2_scalar(center_radius_br.x, center_radius_br.y), vec2(vec2_scalar(center_radius_br.z, center_radius_br.w))); \
    vec4 c_bl = vec4((vec2_scalar(center_radius_bl.x, center_radius_bl.y) - pos) * vec2_scalar(1.0f, -1.0f), vec2(vec2_scalar(center_radius_bl.z, center_radius_bl.w))); \
    auto sel = [&](I32 c, vec4 t, vec4 e) { return vec4(if_then_else(c, t.x, e.x), if_then_else(c, t.y, e.y), if_then_else(c, t.z, e.z), if_then_else(c, t.w, e.w)); }; \
    corner = sel(dot(pos, vec2_scalar(plane_tl.x, plane_tl.y)) > plane_tl.z, c_tl, corner); \
    corner = sel(dot(pos, vec2_scalar(plane_tr.x, plane_tr.y)) > plane_tr.z, c_tr, corner); \
    corner = sel(dot(pos, vec2_scalar(plane_br.x, plane_br.y)) > plane_br.z, c_br, corner); \
    corner = sel(dot(pos, vec2_scalar(plane_bl.x, plane_bl.y)) > plane_bl.z, c_bl, corner); \
    return max(distance_to_ellipse_approx(corner.sel(X, Y), corner.sel(Z, W), Float(1.0f)), \
               signed_distance_rect(pos, vec2_scalar(rect_bounds.x, rect_bounds.y), vec2_scalar(rect_bounds.z, rect_bounds.w))); \
  } \
  static Float signed_distance_rect(vec2 pos, vec2_scalar p0, vec2_scalar p1) {   /* rect.glsl / shared.glsl */ \
    vec2 d = max(p0 - pos, pos - p1); \
    return max(d.x, d.y); \
  } \
 \
  /* ps_quad.glsl:399-415 + ps_quad_mask.glsl:167-200 */ \
  void main() { \
    vec2 clip_local_pos = vClipLocalPos.sel(X, Y) / vClipLocalPos.w; \
    float aa_range = recip(fwidth(clip_local_pos).x);        /* compute_aa_range, shared.glsl:145-148 */ \
    Float dist; \
    if (FAST_PATH) { \
      dist = sd_box(clip_local_pos, vec2_scalar(v_clip_params.x, v_clip_params.y)) - v_clip_params.z; \
    } else { \
      vec3_scalar plane_tl = vec3_scalar(vClipPlane_A.x, vClipPlane_A.y, vClipPlane_A.z); \
      vec3_scalar plane_tr = vec3_scalar(vClipPlane_A.w, vClipPlane_B.x, vClipPlane_B.y); \
      vec3_scalar plane_br = vec3_scalar(vClipPlane_B.z, vClipPlane_B.w, vClipPlane_C.x); \
      vec3_scalar plane_bl = vec3_scalar(vClipPlane_C.y, vClipPlane_C.z, vClipPlane_C.w); \
      dist = distance_to_rounded_rect(clip_local_pos, plane_tl, vClipCenter_Radius_TL, plane_tr, vClipCenter_Radius_TR, \
                                      plane_br, vClipCenter_Radius_BR, plane_bl, vClipCenter_Radius_BL, vTransformBounds); \
    } \
    Float alpha = clamp(0.5f - dist * aa_range, Float(0.0f), Float(1.0f));   /* distance_aa, shared.glsl:160-166 */ \
    Float final_alpha = mix(alpha, 1.0f - alpha, Float(vClipMode.x)); \
    vec4 output_color = vec4(final_alpha); \
    if (v_flags.z != 0) { \
      output_color = output_color.sel(X, X, X, X); \
    } \
    gl_FragColor = output_color; \
  } \
 \
  /* the perspective entry points glsl-to-cxx emits for a program with a varying (lib.rs:660-690, 716-741, 3576-3590) */ \
  struct InterpPerspective { \
    vec4 vClipLocalPos; \
  }; \
  InterpPerspective interp_perspective; \
  static void read_perspective_inputs(FragmentShaderImpl* impl, const void* init_, const void* step_) { \
    Self* self = (Self*)impl; \
    const InterpInputs* init = (const InterpInputs*)init_; \
    const InterpInputs* step = (const InterpInputs*)step_; \
    Float w = 1.0f / self->gl_FragCoord.w; \
    self->interp_perspective.vClipLocalPos = init_interp(init->vClipLocalPos, step->vClipLocalPos); \
    self->vClipLocalPos = self->interp_perspective.vClipLocalPos * w; \
    self->interp_step.vClipLocalPos = step->vClipLocalPos * 4.0f; \
  } \
  ALWAYS_INLINE void step_perspective_inputs(int steps = 4) { \
    step_perspective(steps); \
    float chunks = steps * 0.25f; \
    Float w = 1.0f / gl_FragCoord.w; \
    interp_perspective.vClipLocalPos += interp_step.vClipLocalPos * chunks; \
    vClipLocalPos = w * interp_perspective.vClipLocalPos; \
  } \
  WRSH_FRAG_ABI_PERSPECTIVE(Self) \
  WRSH_FRAG_ABI(Self) \
 \
  NAME##_frag() { \
    WRSH_FRAG_WIRING() \
    WRSH_FRAG_WIRING_PERSPECTIVE() \
  } \
}; \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_PS_QUAD_MASK(ps_quad_mask, "ps_quad_mask", false)
WRSH_PS_QUAD_MASK(ps_quad_mask_FAST_PATH, "ps_quad_mask FAST_PATH", true)
