// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-in for the generated header
// of shader key "ps_split_composite" (webrender_build/src/shader_features.rs:231,
// renderer/shade.rs:1226: the program BatchKind::SplitComposite batches are
// drawn with -- the polygons a preserve-3d context was split into, batch.rs:
// 1985-2080, picture.rs:6571-6622). Restates
// webrender/res/ps_split_composite.glsl:19-113 (VS), 117-133 (FS + span) on the
// prim_shared helpers in brush_base.h, with SWGL defined (SWGL_CLIP_MASK:
// write_clip -> swgl_clipMask, do_clip() == 1; SWGL_DRAW_SPAN).

struct ps_split_composite_vert : wrsh::prim_vert_base<ps_split_composite_vert> {
  typedef ps_split_composite_vert Self;
  vec2 vUv;
  vec2_scalar vPerspective;
  vec4_scalar vUvSampleBounds;
  struct InterpOutputs {
    vec2_scalar vUv;
  };
  // ps_split_composite.glsl:41-45
  static vec2 bilerp(vec2_scalar a, vec2_scalar b, vec2_scalar c, vec2_scalar d,
                     Float s, Float t) {
    vec2 x = mix(vec2(a), vec2(b), t);
    vec2 y = mix(vec2(c), vec2(d), t);
    return mix(x, y, s);
  }
  void main() {
    using namespace wrsh;
    /* fetch_composite_instance, :54-63 */
    int prim_header_index = aData.x;
    int polygons_address = aData.y;
    float ci_z = float(aData.z);
    int render_task_index = aData.w;
    /* fetch_split_geometry, :24-39 */
    vec4_scalar data0 = fetch_from_gpu_cache(polygons_address, 0);
    vec4_scalar data1 = fetch_from_gpu_cache(polygons_address, 1);
    vec2_scalar local[4] = {vec2_scalar(data0.x, data0.y), vec2_scalar(data0.z, data0.w),
                            vec2_scalar(data1.x, data1.y), vec2_scalar(data1.z, data1.w)};
    PrimitiveHeader ph = fetch_prim_header(prim_header_index);
    PictureTask dest_task = fetch_picture_task(render_task_index);
    Transform transform = fetch_transform(ph.transform_id);
    vec4_scalar res_uv_rect = fetch_from_gpu_cache(ph.user_data.x, 0);   /* fetch_image_source */
    ClipArea clip_area = fetch_clip_area(ph.user_data.w);
    vec2_scalar dest_origin = dest_task.task_rect.p0 - dest_task.content_origin;
    vec2 local_pos = bilerp(local[0], local[1], local[3], local[2], aPosition.y, aPosition.x);
    vec4 world_pos = transform.m * vec4(local_pos, 0.0f, 1.0f);
    vec4 final_pos = vec4(dest_origin * world_pos.w + world_pos.sel(X, Y) * dest_task.device_pixel_scale,
                          world_pos.w * ci_z, world_pos.w);
    write_clip(clip_area, dest_task);
    gl_Position = uTransform * final_pos;
    ivec2_scalar ts = textureSize(sColor0, 0);
    vec2_scalar texture_size = vec2_scalar(float(ts.x), float(ts.y));
    vec2_scalar uv0 = vec2_scalar(res_uv_rect.x, res_uv_rect.y);
    vec2_scalar uv1 = vec2_scalar(res_uv_rect.z, res_uv_rect.w);
    vec2_scalar min_uv = min(uv0, uv1);
    vec2_scalar max_uv = max(uv0, uv1);
    vUvSampleBounds = vec4_scalar(min_uv.x + 0.5f, min_uv.y + 0.5f, max_uv.x - 0.5f, max_uv.y - 0.5f) /
                      vec4_scalar(texture_size.x, texture_size.y, texture_size.x, texture_size.y);
    vec2 f = (local_pos - ph.local_rect.p0) / rect_size(ph.local_rect);
    /* get_image_quad_uv, prim_shared.glsl:204-210 */
    {
      vec4_scalar st_tl = fetch_from_gpu_cache(ph.user_data.x + 2, 0);
      vec4_scalar st_tr = fetch_from_gpu_cache(ph.user_data.x + 2, 1);
      vec4_scalar st_bl = fetch_from_gpu_cache(ph.user_data.x + 2, 2);
      vec4_scalar st_br = fetch_from_gpu_cache(ph.user_data.x + 2, 3);
      vec4 x = mix(vec4(st_tl), vec4(st_tr), f.x);
      vec4 y = mix(vec4(st_bl), vec4(st_br), f.x);
      vec4 z = mix(x, y, f.y);
      f = z.sel(X, Y) / z.w;
    }
    vec2 uv = mix(uv0, uv1, f);
    float perspective_interpolate = float(ph.user_data.y);
    vUv = uv / texture_size * mix(gl_Position.w, Float(1.0f), Float(perspective_interpolate));
    vPerspective.x = perspective_interpolate;
  }
  ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {
    for (int n = 0; n < 4; n++) {
      auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);
      dest->vUv = get_nth(vUv, n);
      dest_ptr += stride;
    }
  }
  WRSH_VERT_ABI(Self)
  ps_split_composite_vert() { WRSH_VERT_WIRING(Self) }
};

struct ps_split_composite_frag : FragmentShaderImpl, ps_split_composite_vert {
  typedef ps_split_composite_frag Self;
  typedef ps_split_composite_vert::InterpOutputs InterpInputs;
  InterpInputs interp_step;
  static void read_interp_inputs(FragmentShaderImpl* impl, const void* init_, const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    self->vUv = init_interp(init->vUv, step->vUv);
    self->interp_step.vUv = step->vUv * 4.0f;
  }
  ALWAYS_INLINE void step_interp_inputs(int steps = 4) {
    float chunks = steps * 0.25f;
    vUv += interp_step.vUv * chunks;
  }
  /* the perspective entry points glsl-to-cxx emits for a program with a varying (lib.rs:660-690, 716-741, 3576-3590) */
  struct InterpPerspective {
    vec2 vUv;
  };
  InterpPerspective interp_perspective;
  static void read_perspective_inputs(FragmentShaderImpl* impl, const void* init_, const void* step_) {
    Self* self = (Self*)impl;
    const InterpInputs* init = (const InterpInputs*)init_;
    const InterpInputs* step = (const InterpInputs*)step_;
    Float w = 1.0f / self->gl_FragCoord.w;
    self->interp_perspective.vUv = init_interp(init->vUv, step->vUv);
    self->vUv = self->interp_perspective.vUv * w;
    self->interp_step.vUv = step->vUv * 4.0f;
  }
  ALWAYS_INLINE void step_perspective_inputs(int steps = 4) {
    step_perspective(steps);
    float chunks = steps * 0.25f;
    Float w = 1.0f / gl_FragCoord.w;
    interp_perspective.vUv += interp_step.vUv * chunks;
    vUv = w * interp_perspective.vUv;
  }
  static void run_perspective(FragmentShaderImpl* impl) {
    Self* self = (Self*)impl;
    self->shade(mix(self->gl_FragCoord.w, Float(1.0f), Float(self->vPerspective.x)));
    self->step_perspective_inputs();
  }
  static void skip_perspective(FragmentShaderImpl* impl, int steps) {
    Self* self = (Self*)impl;
    self->step_perspective_inputs(steps);
  }
  /* main, ps_split_composite.glsl:117-122 (2-D path: gl_FragCoord.w == 1) */
  void main() { shade(Float(mix(1.0f, 1.0f, vPerspective.x))); }
  void shade(Float perspective_divisor) {
    float alpha = 1.0f; /* do_clip() */
    vec2 uv = clamp(vUv * perspective_divisor, vec2_scalar(vUvSampleBounds.x, vUvSampleBounds.y),
                    vec2_scalar(vUvSampleBounds.z, vUvSampleBounds.w));
    gl_FragColor = alpha * texture(sColor0, uv);
  }
  /* :124-131 */
  void swgl_drawSpanRGBA8() {
    float perspective_divisor = mix(1.0f, 1.0f, vPerspective.x);
    vec2 uv = vUv * perspective_divisor;
    swgl_commitTextureRGBA8(sColor0, uv, vUvSampleBounds);
  }
  WRSH_FRAG_ABI(Self)
  static int draw_span_RGBA8(FragmentShaderImpl* impl) {
    Self* self = (Self*)impl;
    DISPATCH_DRAW_SPAN(self, RGBA8);
  }
  ps_split_composite_frag() {
    WRSH_FRAG_WIRING()
    WRSH_FRAG_WIRING_PERSPECTIVE()
    draw_span_RGBA8_func = &draw_span_RGBA8;
  }
};
WRSH_PROGRAM(ps_split_composite, "ps_split_composite")
