// ORACLE / TEST INFRASTRUCTURE. Shared vertex-stage restatement for the brush_*
// family: webrender/res/brush.glsl:95-222 (main + brush_shader_main_vs) and
// prim_shared.glsl:54-200 (decode_instance_attributes, fetch_prim_header,
// write_vertex, clip_and_init_antialiasing, write_clip), with SWGL defined.
// CRTP: Derived supplies brush_vs(...) and VECS_PER_SPECIFIC_BRUSH.

namespace wrsh {

struct BrushVertexInfo {
  vec2 local_pos;
  vec4 world_pos;
};

struct PrimitiveHeader {
  RectWithEndpoint local_rect;
  RectWithEndpoint local_clip_rect;
  float z;
  int specific_prim_address;
  int transform_id;
  int picture_task_address;
  ivec4_scalar user_data;
};

// Helpers shared by every shader that includes prim_shared.glsl (brush_*, ps_text_run).
template <typename Derived>
struct prim_vert_base : VertexShaderImpl, CommonState {
  AttribTable attribs;
  int a_aPosition, a_aData;
  vec2 aPosition;
  ivec4_scalar aData;

  static constexpr int BRUSH_FLAG_PERSPECTIVE_INTERPOLATION = 1,
                       BRUSH_FLAG_SEGMENT_RELATIVE = 2,
                       BRUSH_FLAG_SEGMENT_REPEAT_X = 4,
                       BRUSH_FLAG_SEGMENT_REPEAT_Y = 8,
                       BRUSH_FLAG_SEGMENT_REPEAT_X_ROUND = 16,
                       BRUSH_FLAG_SEGMENT_REPEAT_Y_ROUND = 32,
                       BRUSH_FLAG_SEGMENT_REPEAT_X_CENTERED = 64,
                       BRUSH_FLAG_SEGMENT_REPEAT_Y_CENTERED = 128,
                       BRUSH_FLAG_SEGMENT_NINEPATCH_MIDDLE = 256,
                       BRUSH_FLAG_TEXEL_RECT = 512, BRUSH_FLAG_FORCE_AA = 1024,
                       BRUSH_FLAG_NORMALIZED_UVS = 2048;

  // prim_shared.glsl:88-110
  PrimitiveHeader fetch_prim_header(int index) const {
    PrimitiveHeader ph;
    ivec2_scalar uv_f = get_fetch_uv(index, 2u);
    vec4_scalar local_rect = texelFetch(sPrimitiveHeadersF, uv_f, 0);
    vec4_scalar local_clip_rect =
        texelFetch(sPrimitiveHeadersF, ivec2_scalar(uv_f.x + 1, uv_f.y), 0);
    ph.local_rect = RectWithEndpoint{vec2_scalar(local_rect.x, local_rect.y),
                                     vec2_scalar(local_rect.z, local_rect.w)};
    ph.local_clip_rect =
        RectWithEndpoint{vec2_scalar(local_clip_rect.x, local_clip_rect.y),
                         vec2_scalar(local_clip_rect.z, local_clip_rect.w)};
    ivec2_scalar uv_i = get_fetch_uv(index, 2u);
    ivec4_scalar data0 = texelFetch(sPrimitiveHeadersI, uv_i, 0);
    ivec4_scalar data1 =
        texelFetch(sPrimitiveHeadersI, ivec2_scalar(uv_i.x + 1, uv_i.y), 0);
    ph.z = float(data0.x);
    ph.specific_prim_address = data0.y;
    ph.transform_id = data0.z;
    ph.picture_task_address = data0.w;
    ph.user_data = data1;
    return ph;
  }

  // prim_shared.glsl:117-143
  BrushVertexInfo write_vertex(vec2 local_pos, RectWithEndpoint local_clip_rect,
                               float z, Transform transform,
                               PictureTask task) {
    vec2 clamped_local_pos = rect_clamp(local_clip_rect, local_pos);
    vec4 world_pos = transform.m * vec4(clamped_local_pos, 0.0f, 1.0f);
    vec2 device_pos = world_pos.sel(X, Y) * task.device_pixel_scale;
    vec2_scalar final_offset = -task.content_origin + task.task_rect.p0;
    gl_Position =
        uTransform * vec4(device_pos + final_offset * world_pos.w,
                          z * world_pos.w, world_pos.w);
    return BrushVertexInfo{clamped_local_pos, world_pos};
  }

  // prim_shared.glsl:145-181 (SWGL_ANTIALIAS branch)
  RectWithEndpoint clip_and_init_antialiasing(RectWithEndpoint segment_rect,
                                              RectWithEndpoint clip_rect,
                                              int edge_flags) {
    bool cx = clip_rect.p0.x > segment_rect.p0.x;
    bool cy = clip_rect.p0.y > segment_rect.p0.y;
    bool cz = clip_rect.p1.x < segment_rect.p1.x;
    bool cw = clip_rect.p1.y < segment_rect.p1.y;
    swgl_antiAlias(edge_flags | (cx ? 1 : 0) | (cy ? 2 : 0) | (cz ? 4 : 0) |
                   (cw ? 8 : 0));
    segment_rect.p0 = clamp(segment_rect.p0, clip_rect.p0, clip_rect.p1);
    segment_rect.p1 = clamp(segment_rect.p1, clip_rect.p0, clip_rect.p1);
    return segment_rect;
  }

  // prim_shared.glsl:183-200 (SWGL_CLIP_MASK branch)
  void write_clip(ClipArea area, PictureTask task) {
    swgl_clipMask(sClipMask,
                  (task.task_rect.p0 - task.content_origin) -
                      (area.task_rect.p0 - area.screen_origin),
                  area.task_rect.p0, rect_size(area.task_rect));
  }

  static void load_attribs(VertexShaderImpl* impl, VertexAttrib* attribs,
                           uint32_t start, int instance, int count) {
    Derived* self = (Derived*)impl;
    load_attrib(self->aPosition, attribs[self->attribs.locs[self->a_aPosition]],
                start, instance, count);
    load_flat_attrib(self->aData, attribs[self->attribs.locs[self->a_aData]],
                     start, instance, count);
  }

  prim_vert_base() {
    used = (1u << U_sColor0) | (1u << U_sGpuCache) |
           (1u << U_sTransformPalette) | (1u << U_sRenderTasks) |
           (1u << U_sPrimitiveHeadersF) | (1u << U_sPrimitiveHeadersI) |
           (1u << U_sClipMask) | (1u << U_uTransform);
    a_aPosition = attribs.add("aPosition");
    a_aData = attribs.add("aData");
  }
};

// brush.glsl:95-222: main() + brush_shader_main_vs.
template <typename Derived>
struct brush_vert_base : prim_vert_base<Derived> {
  using prim_vert_base<Derived>::aData;
  using prim_vert_base<Derived>::aPosition;
  using prim_vert_base<Derived>::BRUSH_FLAG_FORCE_AA;
  void main() {
    // decode_instance_attributes, prim_shared.glsl:60-72
    int prim_header_address = aData.x;
    int clip_address = aData.y;
    int segment_index = aData.z & 0xffff;
    int flags = aData.z >> 16;
    int resource_address = aData.w & 0xffffff;

    PrimitiveHeader ph = this->fetch_prim_header(prim_header_address);
    Transform transform = this->fetch_transform(ph.transform_id);
    PictureTask task = this->fetch_picture_task(ph.picture_task_address);
    ClipArea clip_area = this->fetch_clip_area(clip_address);

    // brush_shader_main_vs, brush.glsl:95-195
    int edge_flags = (flags >> 12) & 0xf;
    int brush_flags = flags & 0xfff;

    vec4_scalar segment_data;
    RectWithEndpoint segment_rect;
    if (segment_index == 0xffff) {
      segment_rect = ph.local_rect;
      segment_data = vec4_scalar(0.0f);
    } else {
      int segment_address = ph.specific_prim_address +
                            Derived::VECS_PER_SPECIFIC_BRUSH + segment_index * 2;
      vec4_scalar i0 = this->fetch_from_gpu_cache(segment_address, 0);
      vec4_scalar i1 = this->fetch_from_gpu_cache(segment_address, 1);
      segment_rect = RectWithEndpoint{vec2_scalar(i0.x, i0.y),
                                      vec2_scalar(i0.z, i0.w)};
      segment_rect.p0 += ph.local_rect.p0;
      segment_rect.p1 += ph.local_rect.p0;
      segment_data = i1;
    }

    RectWithEndpoint adjusted_segment_rect = segment_rect;
    bool antialiased = !transform.is_axis_aligned ||
                       ((brush_flags & BRUSH_FLAG_FORCE_AA) != 0);
    if (antialiased) {
      adjusted_segment_rect = this->clip_and_init_antialiasing(
          segment_rect, ph.local_clip_rect, edge_flags);
      ph.local_clip_rect.p0 = vec2_scalar(-1.0e16f);
      ph.local_clip_rect.p1 = vec2_scalar(1.0e16f);
    }

    vec2 local_pos =
        mix(adjusted_segment_rect.p0, adjusted_segment_rect.p1, aPosition);

    BrushVertexInfo vi =
        this->write_vertex(local_pos, ph.local_clip_rect, ph.z, transform, task);

    this->write_clip(clip_area, task);

    static_cast<Derived*>(this)->brush_vs(
        vi, ph.specific_prim_address, ph.local_rect, segment_rect, ph.user_data,
        resource_address, transform.m, task, brush_flags, segment_data);
  }

};

}  // namespace wrsh
