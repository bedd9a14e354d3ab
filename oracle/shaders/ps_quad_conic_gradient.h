// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "ps_quad_conic_gradient".  The ps_quad vertex stage of ps_quad_textured.h
// (ps_quad.glsl:164-418) with webrender/res/ps_quad_conic_gradient.glsl's pattern_vertex /
// pattern_fragment and gradient.glsl:30-61 (sample_gradient), SWGL defined.

struct ps_quad_conic_gradient_vert : VertexShaderImpl, wrsh::CommonState {
  typedef ps_quad_conic_gradient_vert Self;
  wrsh::AttribTable attribs;
  int a_aPosition, a_aData;

  // attributes
  vec2 aPosition;
  ivec4_scalar aData;
  // flat varyings
  vec4_scalar v_color;
  ivec4_scalar v_flags;
  vec3_scalar v_start_offset_offset_scale_angle_vec;
  vec2_scalar v_gradient_repeat;
  ivec2_scalar v_gradient_address;
  // varyings
  vec2 v_pos;      // (v_dir)

  struct InterpOutputs {
    vec2_scalar v_pos;
  };

  static constexpr int EDGE_AA_LEFT = 1, EDGE_AA_TOP = 2, EDGE_AA_RIGHT = 4,
                       EDGE_AA_BOTTOM = 8;
  static constexpr int PART_CENTER = 0, PART_LEFT = 1, PART_TOP = 2,
                       PART_RIGHT = 3, PART_BOTTOM = 4, PART_ALL = 5;
  static constexpr int QF_IS_OPAQUE = 1, QF_APPLY_DEVICE_CLIP = 2,
                       QF_IGNORE_DEVICE_SCALE = 4, QF_USE_AA_SEGMENTS = 8,
                       QF_IS_MASK = 16;
  static constexpr int INVALID_SEGMENT_INDEX = 0xff;

  static float edge_aa_offset(int edge, int flags) {
    return ((flags & edge) != 0) ? 2.0f : 0.0f;
  }
  static vec2_scalar so_map_point(vec4_scalar so, vec2_scalar p) {
    return p * vec2_scalar(so.x, so.y) + vec2_scalar(so.z, so.w);
  }
  static vec2 so_map_point(vec4_scalar so, vec2 p) {
    return p * vec2_scalar(so.x, so.y) + vec2_scalar(so.z, so.w);
  }
  static wrsh::RectWithEndpoint so_map_rect(vec4_scalar so,
                                            wrsh::RectWithEndpoint r) {
    return wrsh::RectWithEndpoint{so_map_point(so, r.p0),
                                  so_map_point(so, r.p1)};
  }

  void main() {
    using namespace wrsh;
    // decode_instance, ps_quad.glsl:164-178
    int prim_address_i = aData.x;
    int prim_address_f = aData.y;
    int quad_flags = (aData.z >> 24) & 0xff;
    int edge_flags = (aData.z >> 16) & 0xff;
    int part_index = (aData.z >> 8) & 0xff;
    int segment_index = (aData.z >> 0) & 0xff;
    int picture_task_address = aData.w;

    // fetch_header, ps_quad.glsl:133-143
    ivec4_scalar header = fetch_from_gpu_buffer_1i(prim_address_i);
    int transform_id = header.x;
    int z_id = header.y;

    Transform transform = fetch_transform(transform_id);
    PictureTask task = fetch_picture_task(picture_task_address);

    // fetch_primitive, ps_quad.glsl:112-124
    vec4_scalar t0 = fetch_from_gpu_buffer_f(prim_address_f, 0);
    vec4_scalar t1 = fetch_from_gpu_buffer_f(prim_address_f, 1);
    vec4_scalar t2 = fetch_from_gpu_buffer_f(prim_address_f, 2);
    vec4_scalar pattern_scale_offset = fetch_from_gpu_buffer_f(prim_address_f, 3);
    vec4_scalar prim_color = fetch_from_gpu_buffer_f(prim_address_f, 4);
    RectWithEndpoint prim_bounds{vec2_scalar(t0.x, t0.y), vec2_scalar(t0.z, t0.w)};
    RectWithEndpoint prim_clip{vec2_scalar(t1.x, t1.y), vec2_scalar(t1.z, t1.w)};
    RectWithEndpoint prim_uv_rect{vec2_scalar(t2.x, t2.y), vec2_scalar(t2.z, t2.w)};
    float z = float(z_id);

    RectWithEndpoint seg_rect, seg_uv_rect;
    if (segment_index == INVALID_SEGMENT_INDEX) {
      seg_rect = prim_bounds;
      seg_uv_rect = prim_uv_rect;
    } else {
      // fetch_segment, ps_quad.glsl:99-110
      int base = prim_address_f + 5 + segment_index * 2;
      vec4_scalar s0 = fetch_from_gpu_buffer_f(base, 0);
      vec4_scalar s1 = fetch_from_gpu_buffer_f(base, 1);
      seg_rect = RectWithEndpoint{vec2_scalar(s0.x, s0.y), vec2_scalar(s0.z, s0.w)};
      seg_uv_rect = RectWithEndpoint{vec2_scalar(s1.x, s1.y), vec2_scalar(s1.z, s1.w)};
    }

    RectWithEndpoint lcr = seg_rect;
    lcr.p0 = max(lcr.p0, prim_clip.p0);
    lcr.p1 = min(lcr.p1, prim_clip.p1);
    lcr.p1 = max(lcr.p0, lcr.p1);

    // ps_quad.glsl:267-325 with SWGL_ANTIALIAS
    switch (part_index) {
      case PART_LEFT:
        lcr.p1.x = lcr.p0.x + 2.0f;
        swgl_antiAlias(EDGE_AA_LEFT);
        break;
      case PART_TOP:
        lcr.p0.x = lcr.p0.x + 2.0f;
        lcr.p1.x = lcr.p1.x - 2.0f;
        lcr.p1.y = lcr.p0.y + 2.0f;
        swgl_antiAlias(EDGE_AA_TOP);
        break;
      case PART_RIGHT:
        lcr.p0.x = lcr.p1.x - 2.0f;
        swgl_antiAlias(EDGE_AA_RIGHT);
        break;
      case PART_BOTTOM:
        lcr.p0.x = lcr.p0.x + 2.0f;
        lcr.p1.x = lcr.p1.x - 2.0f;
        lcr.p0.y = lcr.p1.y - 2.0f;
        swgl_antiAlias(EDGE_AA_BOTTOM);
        break;
      case PART_CENTER:
        lcr.p0.x += edge_aa_offset(EDGE_AA_LEFT, edge_flags);
        lcr.p1.x -= edge_aa_offset(EDGE_AA_RIGHT, edge_flags);
        lcr.p0.y += edge_aa_offset(EDGE_AA_TOP, edge_flags);
        lcr.p1.y -= edge_aa_offset(EDGE_AA_BOTTOM, edge_flags);
        break;
      case PART_ALL:
      default:
        swgl_antiAlias(edge_flags);
        break;
    }

    vec2 local_pos = mix(lcr.p0, lcr.p1, aPosition);

    float device_pixel_scale = task.device_pixel_scale;
    if ((quad_flags & QF_IGNORE_DEVICE_SCALE) != 0) {
      device_pixel_scale = 1.0f;
    }

    // write_vertex, ps_quad.glsl:184-221
    vec4 world_pos = transform.m * vec4(local_pos, 0.0f, 1.0f);
    vec2 device_pos = world_pos.sel(X, Y) * device_pixel_scale;
    vec2 vi_local_pos;
    if ((quad_flags & QF_APPLY_DEVICE_CLIP) != 0) {
      RectWithEndpoint device_clip_rect{
          task.content_origin,
          task.content_origin + task.task_rect.p1 - task.task_rect.p0};
      device_pos = rect_clamp(device_clip_rect, device_pos);
      vi_local_pos =
          (transform.inv_m * vec4(device_pos / device_pixel_scale, 0.0f, 1.0f))
              .sel(X, Y);
    } else {
      vi_local_pos = local_pos;
    }
    vec2_scalar final_offset = -task.content_origin + task.task_rect.p0;
    gl_Position =
        uTransform * vec4(device_pos + final_offset * world_pos.w,
                          z * world_pos.w, world_pos.w);

    v_color = prim_color;

    vec4_scalar pattern_tx = pattern_scale_offset;
    seg_rect = so_map_rect(pattern_tx, seg_rect);
    vec2 info_local_pos = so_map_point(pattern_tx, vi_local_pos);

    // main(), ps_quad.glsl:373-384
    if ((quad_flags & QF_IS_MASK) != 0) {
      v_flags.z = 1;
    } else {
      v_flags.z = 0;
    }

    // pattern_vertex, ps_quad_conic_gradient.glsl:40-56
    RectWithEndpoint local_prim_rect = so_map_rect(pattern_tx, prim_bounds);
    vec4_scalar g0 = fetch_from_gpu_buffer_f(header.z, 0);
    vec4_scalar g1 = fetch_from_gpu_buffer_f(header.z, 1);
    vec2_scalar g_center = vec2_scalar(g0.x, g0.y), g_scale = vec2_scalar(g0.z, g0.w);
    float g_start_offset = g1.x, g_end_offset = g1.y, g_angle = g1.z, g_repeat = g1.w;
    v_gradient_address.x = header.w;
    v_gradient_repeat.x = g_repeat;
    float d = g_end_offset - g_start_offset;
    v_start_offset_offset_scale_angle_vec.y = d != 0.0f ? 1.0f / d : 0.0f;
    v_start_offset_offset_scale_angle_vec.z = 3.141592653589793f / 2.0f - g_angle;
    v_start_offset_offset_scale_angle_vec.x = g_start_offset * v_start_offset_offset_scale_angle_vec.y;
    v_pos = ((info_local_pos - local_prim_rect.p0) * g_scale - g_center);
  }

  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Self* self = (Self*)impl;
    load_attrib(self->aPosition, attribs[self->attribs.locs[self->a_aPosition]],
                start, instance, count);
    load_flat_attrib(self->aData, attribs[self->attribs.locs[self->a_aData]],
                     start, instance, count);
  }

  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->v_pos = get_nth(v_pos, n);
      dest_ptr += stride;
    }
  }

  WRSH_VERT_ABI(Self)

  ps_quad_conic_gradient_vert() {
    using namespace wrsh;
    used = (1u << U_sColor0) | (1u << U_sTransformPalette) |
           (1u << U_sRenderTasks) | (1u << U_sGpuBufferF) |
           (1u << U_sGpuBufferI) | (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_aData = attribs.add("aData");
    v_flags = ivec4_scalar(0, 0, 0, 0);
    v_gradient_repeat = vec2_scalar(0.0f, 0.0f);
    v_gradient_address = ivec2_scalar(0, 0);
    WRSH_VERT_WIRING(Self)
  }
};

struct ps_quad_conic_gradient_frag : FragmentShaderImpl, ps_quad_conic_gradient_vert {
  typedef ps_quad_conic_gradient_frag Self;
  typedef ps_quad_conic_gradient_vert::InterpOutputs InterpInputs;
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
  // ps_quad_conic_gradient.glsl:60-70
  static Float approx_atan2(Float y, Float x) {
    vec2 a = abs(vec2(x, y));
    Float slope = min(a.x, a.y) / max(a.x, a.y);
    Float s2 = slope * slope;
    Float r = ((-0.0464964749f * s2 + 0.15931422f) * s2 - 0.327622764f) * s2 * slope + slope;
    r = if_then_else(a.y > a.x, 1.57079637f - r, r);
    r = if_then_else(x < 0.0f, 3.14159274f - r, r);
    r = r * sign(y);
    return r;
  }
  // ps_quad.glsl:399-415 + ps_quad_conic_gradient.glsl:72-81
  void main() {
    vec4 base_color = vec4(v_color);
    float alpha = 1.0f;  // antialiasing_fragment() under SWGL_ANTIALIAS
    base_color *= alpha;
    vec2 current_dir = v_pos;
    Float current_angle = approx_atan2(current_dir.y, current_dir.x) + v_start_offset_offset_scale_angle_vec.z;
    Float offset = fract(current_angle / (2.0f * 3.141592653589793f)) * v_start_offset_offset_scale_angle_vec.y -
                   v_start_offset_offset_scale_angle_vec.x;
    vec4 output_color = base_color * sample_gradient(offset);
    if (v_flags.z != 0) {
      output_color = output_color.sel(X, X, X, X);
    }
    gl_FragColor = output_color;
  }

  // the perspective entry points glsl-to-cxx emits for a program with a varying (lib.rs:660-690, 716-741, 3576-3590)
  struct InterpPerspective {
    vec2 v_pos;
  };
  InterpPerspective interp_perspective;
  static void read_perspective_inputs(FragmentShaderImpl* impl, const void* init_, const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    Float w = 1.0f / self->gl_FragCoord.w;
    self->interp_perspective.v_pos = init_interp(init->v_pos, step->v_pos);
    self->v_pos = self->interp_perspective.v_pos * w;
    self->interp_step.v_pos = step->v_pos * 4.0f;
  }
  ALWAYS_INLINE void step_perspective_inputs(int steps = 4) {
    step_perspective(steps);
    float chunks = steps * 0.25f;
    Float w = 1.0f / gl_FragCoord.w;
    interp_perspective.v_pos += interp_step.v_pos * chunks;
    v_pos = w * interp_perspective.v_pos;
  }
  WRSH_FRAG_ABI_PERSPECTIVE(Self)

  WRSH_FRAG_ABI(Self)
  ps_quad_conic_gradient_frag() {
    WRSH_FRAG_WIRING()
    WRSH_FRAG_WIRING_PERSPECTIVE()
  }
};

WRSH_PROGRAM(ps_quad_conic_gradient, "ps_quad_conic_gradient")
