// ORACLE / TEST INFRASTRUCTURE -- not part of the product path.
//
// Hand-written equivalents of the per-shader C++ headers that the reference
// normally generates at build time with its Rust tool glsl-to-cxx (not
// buildable here: no cargo). They are compiled *into the reference's own,
// unmodified* swgl/src/gl.cc via `#include "load_shader.h"` (gl.cc:2678), so
// the rasteriser, blend stage, samplers and span-commit intrinsics that run
// are the reference's; only the shader bodies are restated here, expression by
// expression, from webrender/res/*.glsl.
//
// Shape of what the emitter produces (followed here, condensed with macros):
//   struct X_common / X_vert : VertexShaderImpl / X_frag : FragmentShaderImpl /
//   X_program : ProgramImpl           glsl-to-cxx/src/lib.rs:200-245
//   Samplers / bind_textures           lib.rs:289-353
//   uniform setters                    lib.rs:355-435
//   AttribLocations                    lib.rs:437-465
//   load_attribs                       lib.rs:514-537
//   InterpOutputs / store              lib.rs:539-594
//   read_interp_inputs / step          lib.rs:596-743
//   run/skip/draw_span wiring          lib.rs:3561-3651
//
// This file is included from inside gl.cc after blend.h/swgl_ext.h, so all of
// swgl's glsl.h types and swgl_* intrinsics are in scope.

#include <string.h>

namespace wrsh {

// Fixed uniform indices shared by all hand-written programs (the emitter
// assigns per-program indices; only name -> index -> setter consistency
// matters to gl.cc: GetUniformLocation / Uniform1i / UniformMatrix4fv).
enum Uniform {
  U_sColor0 = 1,
  U_sColor1,
  U_sColor2,
  U_sGpuCache,
  U_sTransformPalette,
  U_sRenderTasks,
  U_sDither,
  U_sPrimitiveHeadersF,
  U_sPrimitiveHeadersI,
  U_sClipMask,
  U_sGpuBufferF,
  U_sGpuBufferI,
  U_uTransform,
  U_COUNT
};

static const char* const uniform_names[U_COUNT] = {
    nullptr,           "sColor0",       "sColor1",
    "sColor2",         "sGpuCache",     "sTransformPalette",
    "sRenderTasks",    "sDither",       "sPrimitiveHeadersF",
    "sPrimitiveHeadersI", "sClipMask",  "sGpuBufferF",
    "sGpuBufferI",     "uTransform"};

#define WR_MAX_VERTEX_TEXTURE_WIDTH 1024u

// shared.glsl:77 get_fetch_uv
static ALWAYS_INLINE ivec2_scalar get_fetch_uv(int i, unsigned vpi) {
  return ivec2_scalar(
      int(vpi * (unsigned(i) % (WR_MAX_VERTEX_TEXTURE_WIDTH / vpi))),
      int(unsigned(i) / (WR_MAX_VERTEX_TEXTURE_WIDTH / vpi)));
}

// gpu_cache.glsl:16 / gpu_buffer.glsl:8
static ALWAYS_INLINE ivec2_scalar get_gpu_uv(int address) {
  return ivec2_scalar(int(unsigned(address) % WR_MAX_VERTEX_TEXTURE_WIDTH),
                      int(unsigned(address) / WR_MAX_VERTEX_TEXTURE_WIDTH));
}

struct RectWithEndpoint {
  vec2_scalar p0;
  vec2_scalar p1;
};

struct RectWithEndpointV {
  vec2 p0;
  vec2 p1;
};

// transform.glsl:16-46
struct Transform {
  mat4_scalar m;
  mat4_scalar inv_m;
  bool is_axis_aligned;
};

// render_task.glsl:60-79
struct PictureTask {
  RectWithEndpoint task_rect;
  float device_pixel_scale;
  vec2_scalar content_origin;
};

// render_task.glsl:83-101
struct ClipArea {
  RectWithEndpoint task_rect;
  float device_pixel_scale;
  vec2_scalar screen_origin;
};

// The union of every sampler/uniform any in-scope WebRender shader declares.
// `used` is the per-program mask of the ones it really declares; only those
// resolve in get_uniform() and get looked up in bind_textures().
struct CommonState {
  unsigned used = 0;

  sampler2D_impl sColor0_impl, sColor1_impl, sColor2_impl, sGpuCache_impl,
      sTransformPalette_impl, sRenderTasks_impl, sDither_impl,
      sPrimitiveHeadersF_impl, sClipMask_impl, sGpuBufferF_impl;
  isampler2D_impl sPrimitiveHeadersI_impl, sGpuBufferI_impl;
  int slots[U_COUNT] = {0};

  sampler2D sColor0 = nullptr, sColor1 = nullptr, sColor2 = nullptr,
            sGpuCache = nullptr, sTransformPalette = nullptr,
            sRenderTasks = nullptr, sDither = nullptr,
            sPrimitiveHeadersF = nullptr, sClipMask = nullptr,
            sGpuBufferF = nullptr;
  isampler2D sPrimitiveHeadersI = nullptr, sGpuBufferI = nullptr;
  mat4_scalar uTransform;

  bool uses(int u) const { return (used >> u) & 1; }

  void bind_textures() {
#define WRSH_BIND(name)  \
  if (uses(U_##name))    \
    name = lookup_sampler(&name##_impl, slots[U_##name]);
#define WRSH_IBIND(name) \
  if (uses(U_##name))    \
    name = lookup_isampler(&name##_impl, slots[U_##name]);
    WRSH_BIND(sColor0)
    WRSH_BIND(sColor1)
    WRSH_BIND(sColor2)
    WRSH_BIND(sGpuCache)
    WRSH_BIND(sTransformPalette)
    WRSH_BIND(sRenderTasks)
    WRSH_BIND(sDither)
    WRSH_BIND(sPrimitiveHeadersF)
    WRSH_IBIND(sPrimitiveHeadersI)
    WRSH_BIND(sClipMask)
    WRSH_BIND(sGpuBufferF)
    WRSH_IBIND(sGpuBufferI)
#undef WRSH_BIND
#undef WRSH_IBIND
  }

  int get_uniform_index(const char* name) const {
    for (int i = 1; i < U_COUNT; i++) {
      if (uses(i) && strcmp(uniform_names[i], name) == 0) return i;
    }
    return -1;
  }

  // --- data-texture fetch helpers (restating the GLSL fetch functions) ---

  // transform.glsl:22-46
  Transform fetch_transform(int id) const {
    Transform t;
    t.is_axis_aligned = (id >> 23) == 0;
    int index = id & 0x007fffff;
    ivec2_scalar uv = get_fetch_uv(index, 8u);
    for (int k = 0; k < 4; k++) {
      t.m[k] = texelFetch(sTransformPalette, ivec2_scalar(uv.x + k, uv.y), 0);
      t.inv_m[k] =
          texelFetch(sTransformPalette, ivec2_scalar(uv.x + 4 + k, uv.y), 0);
    }
    return t;
  }

  // render_task.glsl:17-79
  PictureTask fetch_picture_task(int address) const {
    ivec2_scalar uv = get_fetch_uv(address, 2u);
    vec4_scalar texel0 = texelFetch(sRenderTasks, uv, 0);
    vec4_scalar texel1 =
        texelFetch(sRenderTasks, ivec2_scalar(uv.x + 1, uv.y), 0);
    PictureTask task;
    task.task_rect = RectWithEndpoint{vec2_scalar(texel0.x, texel0.y),
                                      vec2_scalar(texel0.z, texel0.w)};
    task.device_pixel_scale = texel1.x;
    task.content_origin = vec2_scalar(texel1.y, texel1.z);
    return task;
  }

  // render_task.glsl:83-101
  ClipArea fetch_clip_area(int index) const {
    ClipArea area;
    if (index >= 0x7FFFFFFF) {
      area.task_rect = RectWithEndpoint{vec2_scalar(0.0f), vec2_scalar(0.0f)};
      area.device_pixel_scale = 0.0f;
      area.screen_origin = vec2_scalar(0.0f);
    } else {
      PictureTask t = fetch_picture_task(index);
      area.task_rect = t.task_rect;
      area.device_pixel_scale = t.device_pixel_scale;
      area.screen_origin = t.content_origin;
    }
    return area;
  }

  vec4_scalar fetch_from_gpu_cache_1(int address) const {
    return texelFetch(sGpuCache, get_gpu_uv(address), 0);
  }
  vec4_scalar fetch_from_gpu_cache(int address, int k) const {
    ivec2_scalar uv = get_gpu_uv(address);
    return texelFetch(sGpuCache, ivec2_scalar(uv.x + k, uv.y), 0);
  }
  vec4_scalar fetch_from_gpu_buffer_f(int address, int k) const {
    ivec2_scalar uv = get_gpu_uv(address);
    return texelFetch(sGpuBufferF, ivec2_scalar(uv.x + k, uv.y), 0);
  }
  ivec4_scalar fetch_from_gpu_buffer_1i(int address) const {
    return texelFetch(sGpuBufferI, get_gpu_uv(address), 0);
  }
};

// Generic attribute-location table (lib.rs:437-465), up to 20 named attribs.
struct AttribTable {
  static constexpr int MAX = 20;
  const char* names[MAX] = {nullptr};
  int locs[MAX];
  int count = 0;
  int add(const char* name) {
    names[count] = name;
    locs[count] = NULL_ATTRIB;
    return count++;
  }
  void bind_loc(const char* name, int index) {
    for (int i = 0; i < count; i++) {
      if (strcmp(names[i], name) == 0) {
        locs[i] = index;
        return;
      }
    }
  }
  int get_loc(const char* name) const {
    for (int i = 0; i < count; i++) {
      if (strcmp(names[i], name) == 0) {
        return locs[i] != NULL_ATTRIB ? locs[i] : -1;
      }
    }
    return -1;
  }
};

// rect.glsl helpers
static ALWAYS_INLINE vec2_scalar rect_size(const RectWithEndpoint& r) {
  return r.p1 - r.p0;
}
static ALWAYS_INLINE vec2 rect_clamp(const RectWithEndpoint& r, vec2 pt) {
  return clamp(pt, vec2(r.p0), vec2(r.p1));
}

// Uniform setters common to every program (lib.rs:355-435).
template <typename Self>
static void set_uniform_1i(VertexShaderImpl* impl, int index, int value) {
  Self* self = (Self*)impl;
  if (index > 0 && index < U_COUNT) self->slots[index] = value;
}
template <typename Self>
static void set_uniform_4fv(VertexShaderImpl*, int, const float*) {
  assert(0);
}
template <typename Self>
static void set_uniform_matrix4fv(VertexShaderImpl* impl, int index,
                                  const float* value) {
  Self* self = (Self*)impl;
  if (index == U_uTransform) {
    self->uTransform = mat4_scalar::load_from_ptr(value);
  } else {
    assert(0);
  }
}

// ProgramImpl boilerplate (lib.rs:222-240).
#define WRSH_PROGRAM(NAME, KEYSTR)                                            \
  struct NAME##_program : ProgramImpl, NAME##_frag {                          \
    int get_uniform(const char* name) const override {                        \
      return get_uniform_index(name);                                         \
    }                                                                         \
    void bind_attrib(const char* name, int index) override {                  \
      attribs.bind_loc(name, index);                                          \
    }                                                                         \
    int get_attrib(const char* name) const override {                         \
      return attribs.get_loc(name);                                           \
    }                                                                         \
    size_t interpolants_size() const override {                               \
      return sizeof(InterpOutputs);                                           \
    }                                                                         \
    VertexShaderImpl* get_vertex_shader() override { return this; }           \
    FragmentShaderImpl* get_fragment_shader() override { return this; }       \
    const char* get_name() const override { return KEYSTR; }                  \
    static ProgramImpl* loader() { return new NAME##_program; }               \
  };

// Vertex-side function-pointer wiring (lib.rs:3626-3637).
#define WRSH_VERT_WIRING(Self)                                    \
  set_uniform_1i_func = &wrsh::set_uniform_1i<Self>;              \
  set_uniform_4fv_func = &wrsh::set_uniform_4fv<Self>;            \
  set_uniform_matrix4fv_func = &wrsh::set_uniform_matrix4fv<Self>; \
  init_batch_func = &Self::init_batch;                            \
  load_attribs_func = &Self::load_attribs;                        \
  run_primitive_func = &Self::run;

#define WRSH_VERT_ABI(Self)                                                  \
  static void run(VertexShaderImpl* impl, char* interps,                     \
                  size_t interp_stride) {                                    \
    Self* self = (Self*)impl;                                                \
    self->main();                                                            \
    self->store_interp_outputs(interps, interp_stride);                      \
  }                                                                          \
  static void init_batch(VertexShaderImpl* impl) {                           \
    Self* self = (Self*)impl;                                                \
    self->bind_textures();                                                   \
  }

// Fragment-side wiring for programs WITHOUT interpolated varyings and without a read of
// gl_FragCoord.z / .w (glsl-to-cxx: use_perspective false, lib.rs:656-659): the generated
// constructor wires the W entry points of draw_perspective_spans (rasterize.h:1064-1280) to the
// plain ones (lib.rs:3632-3636) -- so gl_FragCoord.z is NOT stepped from chunk to chunk and every
// chunk of a perspective span is depth-tested with the z of the span's first four pixels.  That
// is what brush_solid does.  Programs with varyings restate read_perspective_inputs /
// step_perspective_inputs / run_perspective / skip_perspective themselves and use
// WRSH_FRAG_WIRING_PERSPECTIVE (ps_quad_textured.h); the ones that have not been restated yet keep
// this wiring, which is wrong for them under perspective -- libwrhip reports such prims as
// unsupported, so they are never compared.
#define WRSH_FRAG_ABI(Self)                                   \
  static void run(FragmentShaderImpl* impl) {                 \
    Self* self = (Self*)impl;                                 \
    self->main();                                             \
    self->step_interp_inputs();                               \
  }                                                           \
  static void skip(FragmentShaderImpl* impl, int steps) {     \
    Self* self = (Self*)impl;                                 \
    self->step_interp_inputs(steps);                          \
  }

#define WRSH_FRAG_WIRING()                    \
  init_span_func = &read_interp_inputs;       \
  run_func = &run;                            \
  skip_func = &skip;                          \
  init_span_w_func = &read_interp_inputs;     \
  run_w_func = &run;                          \
  skip_w_func = &skip;

// lib.rs:3576-3590, 3627-3631
#define WRSH_FRAG_ABI_PERSPECTIVE(Self)                                   \
  static void run_perspective(FragmentShaderImpl* impl) {                 \
    Self* self = (Self*)impl;                                             \
    self->main();                                                         \
    self->step_perspective_inputs();                                      \
  }                                                                       \
  static void skip_perspective(FragmentShaderImpl* impl, int steps) {     \
    Self* self = (Self*)impl;                                             \
    self->step_perspective_inputs(steps);                                 \
  }

#define WRSH_FRAG_WIRING_PERSPECTIVE()             \
  enable_perspective();                            \
  init_span_w_func = &read_perspective_inputs;     \
  run_w_func = &run_perspective;                   \
  skip_w_func = &skip_perspective;

}  // namespace wrsh
