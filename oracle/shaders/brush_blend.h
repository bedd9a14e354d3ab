// ORACLE / TEST INFRASTRUCTURE. Hand-written stand-ins for the generated headers
// of shader keys "brush_blend" and "brush_blend ALPHA_PASS"
// (webrender_build/src/shader_features.rs:108-112). Restates
// webrender/res/brush_blend.glsl:43-120 (VS :43-87, FS :91-120) and
// webrender/res/blend.glsl:27-237 (SetupFilterParams, CalculateFilter,
// ComponentTransfer) on brush_base.h, with swgl's glsl.h vector types.
// brush_blend has no swgl_drawSpan*: every pixel runs main() four at a time
// (draw_span_RGBA8_func stays null, rasterize.h:1003-1055).
// Under SWGL the ALPHA_PASS variant multiplies by antialias_brush() == 1.0.

#define WRSH_BRUSH_BLEND(NAME, KEYSTR, ALPHA_PASS)                             \
  struct NAME##_vert : wrsh::brush_vert_base<NAME##_vert> {                    \
    typedef NAME##_vert Self;                                                  \
    static constexpr int VECS_PER_SPECIFIC_BRUSH = 3;                          \
    static constexpr int FILTER_CONTRAST = 0, FILTER_GRAYSCALE = 1,            \
                         FILTER_HUE_ROTATE = 2, FILTER_INVERT = 3,             \
                         FILTER_SATURATE = 4, FILTER_SEPIA = 5,                \
                         FILTER_BRIGHTNESS = 6, FILTER_COLOR_MATRIX = 7,       \
                         FILTER_SRGB_TO_LINEAR = 8, FILTER_LINEAR_TO_SRGB = 9, \
                         FILTER_FLOOD = 10, FILTER_COMPONENT_TRANSFER = 11;    \
    vec2 v_uv;                                                                 \
    vec4_scalar v_uv_sample_bounds;                                            \
    vec2_scalar v_perspective_amount;                                          \
    ivec2_scalar v_op_table_address_vec;                                       \
    mat4_scalar v_color_mat;                                                   \
    vec4_scalar v_funcs;                                                       \
    vec4_scalar v_color_offset;                                                \
    struct InterpOutputs {                                                     \
      vec2_scalar v_uv;                                                        \
    };                                                                         \
    /* blend.glsl:27-87 */                                                     \
    void SetupFilterParams(int op, float amount, int gpu_data_address,         \
                           vec4_scalar& color_offset, mat4_scalar& color_mat,  \
                           int& table_address) {                               \
      float lumR = 0.2126f;                                                    \
      float lumG = 0.7152f;                                                    \
      float lumB = 0.0722f;                                                    \
      float oneMinusLumR = 1.0f - lumR;                                        \
      float oneMinusLumG = 1.0f - lumG;                                        \
      float oneMinusLumB = 1.0f - lumB;                                        \
      float invAmount = 1.0f - amount;                                         \
      if (op == FILTER_GRAYSCALE) {                                            \
        color_mat = mat4_scalar(                                               \
            vec4_scalar(lumR + oneMinusLumR * invAmount, lumR - lumR * invAmount, lumR - lumR * invAmount, 0.0f), \
            vec4_scalar(lumG - lumG * invAmount, lumG + oneMinusLumG * invAmount, lumG - lumG * invAmount, 0.0f), \
            vec4_scalar(lumB - lumB * invAmount, lumB - lumB * invAmount, lumB + oneMinusLumB * invAmount, 0.0f), \
            vec4_scalar(0.0f, 0.0f, 0.0f, 1.0f));                              \
        color_offset = vec4_scalar(0.0f);                                      \
      } else if (op == FILTER_HUE_ROTATE) {                                    \
        float c = cos(amount);                                                 \
        float s = sin(amount);                                                 \
        color_mat = mat4_scalar(                                               \
            vec4_scalar(lumR + oneMinusLumR * c - lumR * s, lumR - lumR * c + 0.143f * s, lumR - lumR * c - oneMinusLumR * s, 0.0f), \
            vec4_scalar(lumG - lumG * c - lumG * s, lumG + oneMinusLumG * c + 0.140f * s, lumG - lumG * c + lumG * s, 0.0f), \
            vec4_scalar(lumB - lumB * c + oneMinusLumB * s, lumB - lumB * c - 0.283f * s, lumB + oneMinusLumB * c + lumB * s, 0.0f), \
            vec4_scalar(0.0f, 0.0f, 0.0f, 1.0f));                              \
        color_offset = vec4_scalar(0.0f);                                      \
      } else if (op == FILTER_SATURATE) {                                      \
        color_mat = mat4_scalar(                                               \
            vec4_scalar(invAmount * lumR + amount, invAmount * lumR, invAmount * lumR, 0.0f), \
            vec4_scalar(invAmount * lumG, invAmount * lumG + amount, invAmount * lumG, 0.0f), \
            vec4_scalar(invAmount * lumB, invAmount * lumB, invAmount * lumB + amount, 0.0f), \
            vec4_scalar(0.0f, 0.0f, 0.0f, 1.0f));                              \
        color_offset = vec4_scalar(0.0f);                                      \
      } else if (op == FILTER_SEPIA) {                                         \
        color_mat = mat4_scalar(                                               \
            vec4_scalar(0.393f + 0.607f * invAmount, 0.349f - 0.349f * invAmount, 0.272f - 0.272f * invAmount, 0.0f), \
            vec4_scalar(0.769f - 0.769f * invAmount, 0.686f + 0.314f * invAmount, 0.534f - 0.534f * invAmount, 0.0f), \
            vec4_scalar(0.189f - 0.189f * invAmount, 0.168f - 0.168f * invAmount, 0.131f + 0.869f * invAmount, 0.0f), \
            vec4_scalar(0.0f, 0.0f, 0.0f, 1.0f));                              \
        color_offset = vec4_scalar(0.0f);                                      \
      } else if (op == FILTER_COLOR_MATRIX) {                                  \
        color_mat = mat4_scalar(fetch_from_gpu_cache(gpu_data_address, 0),     \
                                fetch_from_gpu_cache(gpu_data_address, 1),     \
                                fetch_from_gpu_cache(gpu_data_address, 2),     \
                                fetch_from_gpu_cache(gpu_data_address, 3));    \
        color_offset = fetch_from_gpu_cache_1(gpu_data_address + 4);           \
      } else if (op == FILTER_COMPONENT_TRANSFER) {                            \
        table_address = gpu_data_address;                                      \
      } else if (op == FILTER_FLOOD) {                                         \
        color_offset = fetch_from_gpu_cache_1(gpu_data_address);               \
      }                                                                        \
    }                                                                          \
    /* brush_blend.glsl:43-87 */                                               \
    void brush_vs(wrsh::BrushVertexInfo vi, int, wrsh::RectWithEndpoint local_rect, \
                  wrsh::RectWithEndpoint, ivec4_scalar prim_user_data, int,    \
                  mat4_scalar, wrsh::PictureTask, int brush_flags,             \
                  vec4_scalar) {                                               \
      using namespace wrsh;                                                    \
      /* fetch_image_source, gpu_cache.glsl:104-109 */                         \
      vec4_scalar res0 = fetch_from_gpu_cache(prim_user_data.x, 0);            \
      vec2_scalar uv0 = vec2_scalar(res0.x, res0.y);                           \
      vec2_scalar uv1 = vec2_scalar(res0.z, res0.w);                           \
      ivec2_scalar ts = textureSize(sColor0, 0);                               \
      vec2_scalar inv_texture_size =                                           \
          vec2_scalar(1.0f) / vec2_scalar(float(ts.x), float(ts.y));           \
      vec2 f = (vi.local_pos - local_rect.p0) / rect_size(local_rect);         \
      /* get_image_quad_uv, prim_shared.glsl:204-210 (VECS_PER_IMAGE_RESOURCE = 2) */ \
      {                                                                        \
        vec4_scalar st_tl = fetch_from_gpu_cache(prim_user_data.x + 2, 0);     \
        vec4_scalar st_tr = fetch_from_gpu_cache(prim_user_data.x + 2, 1);     \
        vec4_scalar st_bl = fetch_from_gpu_cache(prim_user_data.x + 2, 2);     \
        vec4_scalar st_br = fetch_from_gpu_cache(prim_user_data.x + 2, 3);     \
        vec4 x = mix(vec4(st_tl), vec4(st_tr), f.x);                           \
        vec4 y = mix(vec4(st_bl), vec4(st_br), f.x);                           \
        vec4 z = mix(x, y, f.y);                                               \
        f = z.sel(X, Y) / z.w;                                                 \
      }                                                                        \
      vec2 uv = mix(uv0, uv1, f);                                              \
      float perspective_interpolate =                                          \
          (brush_flags & BRUSH_FLAG_PERSPECTIVE_INTERPOLATION) != 0 ? 1.0f : 0.0f; \
      v_uv = uv * inv_texture_size *                                           \
             mix(vi.world_pos.w, Float(1.0f), Float(perspective_interpolate)); \
      v_perspective_amount.x = perspective_interpolate;                        \
      v_uv_sample_bounds =                                                     \
          vec4_scalar(uv0.x + 0.5f, uv0.y + 0.5f, uv1.x - 0.5f, uv1.y - 0.5f) * \
          vec4_scalar(inv_texture_size.x, inv_texture_size.y,                  \
                      inv_texture_size.x, inv_texture_size.y);                 \
      float amount = float(prim_user_data.z) / 65536.0f;                       \
      v_op_table_address_vec.x = prim_user_data.y & 0xffff;                    \
      v_perspective_amount.y = amount;                                         \
      v_funcs.x = float((prim_user_data.y >> 28) & 0xf);                       \
      v_funcs.y = float((prim_user_data.y >> 24) & 0xf);                       \
      v_funcs.z = float((prim_user_data.y >> 20) & 0xf);                       \
      v_funcs.w = float((prim_user_data.y >> 16) & 0xf);                       \
      SetupFilterParams(v_op_table_address_vec.x, amount, prim_user_data.z,    \
                        v_color_offset, v_color_mat,                           \
                        v_op_table_address_vec.y);                             \
    }                                                                          \
    ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {   \
      for (int n = 0; n < 4; n++) {                                            \
        auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);               \
        dest->v_uv = get_nth(v_uv, n);                                         \
        dest_ptr += stride;                                                    \
      }                                                                        \
    }                                                                          \
    WRSH_VERT_ABI(Self)                                                        \
    NAME##_vert() { WRSH_VERT_WIRING(Self) }                                   \
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
      self->v_uv = init_interp(init->v_uv, step->v_uv);                        \
      self->interp_step.v_uv = step->v_uv * 4.0f;                              \
    }                                                                          \
    ALWAYS_INLINE void step_interp_inputs(int steps = 4) {                     \
      float chunks = steps * 0.25f;                                            \
      v_uv += interp_step.v_uv * chunks;                                       \
    }                                                                          \
    /* fetch_from_gpu_cache_1 with a per-lane address (gpu_cache.glsl:16-21, 34) */ \
    vec4 fetch_from_gpu_cache_1v(I32 address) const {                          \
      U32 a = U32(address);                                                    \
      return texelFetch(sGpuCache,                                             \
                        ivec2(I32(a % 1024u), I32(a / 1024u)), 0);             \
    }                                                                          \
    /* blend.glsl:90-104 */                                                    \
    vec3 Contrast(vec3 Cs, float amount) {                                     \
      return clamp(Cs * amount - 0.5f * amount + 0.5f, Float(0.0f), Float(1.0f)); \
    }                                                                          \
    vec3 Invert(vec3 Cs, float amount) {                                       \
      return mix(Cs, vec3(Float(1.0f)) - Cs, amount);                                 \
    }                                                                          \
    vec3 Brightness(vec3 Cs, float amount) {                                   \
      return clamp(Cs * amount, vec3(Float(0.0f)), vec3(Float(1.0f)));                       \
    }                                                                          \
    /* blend.glsl:111-121 */                                                   \
    vec3 SrgbToLinear(vec3 color) {                                            \
      vec3 c1 = color / 12.92f;                                                \
      vec3 c2 = pow(color / 1.055f + vec3(Float(0.055f / 1.055f)), vec3(Float(2.4f)));       \
      return if_then_else(lessThanEqual(color, vec3(Float(0.04045f))), c1, c2);       \
    }                                                                          \
    vec3 LinearToSrgb(vec3 color) {                                            \
      vec3 c1 = color * 12.92f;                                                \
      vec3 c2 = vec3(Float(1.055f)) * pow(color, vec3(Float(1.0f / 2.4f))) - vec3(Float(0.055f));   \
      return if_then_else(lessThanEqual(color, vec3(Float(0.0031308f))), c1, c2);     \
    }                                                                          \
    /* blend.glsl:126-188 */                                                   \
    vec4 ComponentTransfer(vec4 colora, vec4_scalar vfuncs, int table_address) { \
      int offset = 0;                                                          \
      I32 k;                                                                   \
      vec4 texel;                                                              \
      int funcs[4] = {int(vfuncs.x), int(vfuncs.y), int(vfuncs.z), int(vfuncs.w)}; \
      for (int i = 0; i < 4; i++) {                                            \
        switch (funcs[i]) {                                                    \
          case 0: /* COMPONENT_TRANSFER_IDENTITY */                            \
            break;                                                             \
          case 1: /* COMPONENT_TRANSFER_TABLE */                               \
          case 2: /* COMPONENT_TRANSFER_DISCRETE */ {                          \
            k = cast(floor(colora[i] * 255.0f + 0.5f));                        \
            texel = fetch_from_gpu_cache_1v(table_address + offset + k / 4);   \
            colora[i] = clamp(texel[k % 4], Float(0.0f), Float(1.0f));         \
            offset = offset + 64;                                              \
            break;                                                             \
          }                                                                    \
          case 3: /* COMPONENT_TRANSFER_LINEAR */ {                            \
            vec4_scalar t = fetch_from_gpu_cache_1(table_address + offset);    \
            colora[i] = clamp(t.x * colora[i] + t.y, Float(0.0f), Float(1.0f)); \
            offset = offset + 1;                                               \
            break;                                                             \
          }                                                                    \
          case 4: /* COMPONENT_TRANSFER_GAMMA */ {                             \
            vec4_scalar t = fetch_from_gpu_cache_1(table_address + offset);    \
            colora[i] = clamp(t.x * pow(colora[i], Float(t.y)) + t.z,          \
                              Float(0.0f), Float(1.0f));                       \
            offset = offset + 1;                                               \
            break;                                                             \
          }                                                                    \
          default:                                                             \
            break;                                                             \
        }                                                                      \
      }                                                                        \
      return colora;                                                           \
    }                                                                          \
    /* blend.glsl:190-237 */                                                   \
    void CalculateFilter(vec4 Cs, int op, float amount, int table_address,     \
                         vec4_scalar color_offset, mat4_scalar color_mat,      \
                         vec4_scalar v_funcs_, vec3& color, Float& alpha) {    \
      alpha = Cs.w;                                                            \
      color = if_then_else(alpha != 0.0f, Cs.sel(X, Y, Z) / alpha,             \
                           Cs.sel(X, Y, Z));                                   \
      switch (op) {                                                            \
        case FILTER_CONTRAST:                                                  \
          color = Contrast(color, amount);                                     \
          break;                                                               \
        case FILTER_INVERT:                                                    \
          color = Invert(color, amount);                                       \
          break;                                                               \
        case FILTER_BRIGHTNESS:                                                \
          color = Brightness(color, amount);                                   \
          break;                                                               \
        case FILTER_SRGB_TO_LINEAR:                                            \
          color = SrgbToLinear(color);                                         \
          break;                                                               \
        case FILTER_LINEAR_TO_SRGB:                                            \
          color = LinearToSrgb(color);                                         \
          break;                                                               \
        case FILTER_COMPONENT_TRANSFER: {                                      \
          vec4 colora = vec4(color, alpha);                                    \
          colora = ComponentTransfer(colora, v_funcs_, table_address);         \
          color = colora.sel(X, Y, Z);                                         \
          alpha = colora.w;                                                    \
          break;                                                               \
        }                                                                      \
        case FILTER_FLOOD:                                                     \
          color = vec3(vec3_scalar(color_offset.x, color_offset.y, color_offset.z)); \
          alpha = color_offset.w;                                              \
          break;                                                               \
        default: {                                                             \
          vec4 result = color_mat * vec4(color, alpha) + color_offset;         \
          result = clamp(result, vec4(Float(0.0f)), vec4(Float(1.0f)));                      \
          color = result.sel(X, Y, Z);                                         \
          alpha = result.w;                                                    \
        }                                                                      \
      }                                                                        \
    }                                                                          \
    /* the perspective entry points glsl-to-cxx emits for a program with a */  \
    /* varying (lib.rs:660-690, 716-741, 3576-3590) */                         \
    struct InterpPerspective {                                                 \
      vec2 v_uv;                                                                \
    };                                                                         \
    InterpPerspective interp_perspective;                                      \
    static void read_perspective_inputs(FragmentShaderImpl* impl,              \
                                        const void* init_, const void* step_) { \
      Self* self = (Self*)impl;                                                \
      const InterpInputs* init = (const InterpInputs*)init_;                   \
      const InterpInputs* step = (const InterpInputs*)step_;                   \
      Float w = 1.0f / self->gl_FragCoord.w;                                   \
      self->interp_perspective.v_uv = init_interp(init->v_uv, step->v_uv);        \
      self->v_uv = self->interp_perspective.v_uv * w;                            \
      self->interp_step.v_uv = step->v_uv * 4.0f;                                \
    }                                                                          \
    ALWAYS_INLINE void step_perspective_inputs(int steps = 4) {                \
      step_perspective(steps);                                                 \
      float chunks = steps * 0.25f;                                            \
      Float w = 1.0f / gl_FragCoord.w;                                         \
      interp_perspective.v_uv += interp_step.v_uv * chunks;                      \
      v_uv = w * interp_perspective.v_uv;                                        \
    }                                                                          \
    static void run_perspective(FragmentShaderImpl* impl) {                    \
      Self* self = (Self*)impl;                                                \
      self->shade(mix(self->gl_FragCoord.w, Float(1.0f), Float(self->v_perspective_amount.x)));                                                              \
      self->step_perspective_inputs();                                         \
    }                                                                          \
    static void skip_perspective(FragmentShaderImpl* impl, int steps) {        \
      Self* self = (Self*)impl;                                                \
      self->step_perspective_inputs(steps);                                    \
    }                                                                          \
    /* brush_fs + main, brush_blend.glsl:91-120 (2-D path: gl_FragCoord.w == 1) */ \
    void main() { shade(Float(mix(1.0f, 1.0f, v_perspective_amount.x))); }    \
    void shade(Float perspective_divisor) {                                    \
      vec2 uv = v_uv * perspective_divisor;                                    \
      uv = clamp(uv, vec2_scalar(v_uv_sample_bounds.x, v_uv_sample_bounds.y),  \
                 vec2_scalar(v_uv_sample_bounds.z, v_uv_sample_bounds.w));     \
      vec4 Cs = texture(sColor0, uv);                                          \
      Float alpha;                                                             \
      vec3 color;                                                              \
      CalculateFilter(Cs, v_op_table_address_vec.x, v_perspective_amount.y,    \
                      v_op_table_address_vec.y, v_color_offset, v_color_mat,   \
                      v_funcs, color, alpha);                                  \
      if (ALPHA_PASS) {                                                        \
        alpha *= 1.0f; /* antialias_brush() */                                 \
      }                                                                        \
      vec4 frag = alpha * vec4(color, 1.0f);                                   \
      frag *= 1.0f; /* brush.glsl main(): do_clip() under SWGL_CLIP_MASK */    \
      gl_FragColor = frag;                                                     \
    }                                                                          \
    WRSH_FRAG_ABI(Self)                                                        \
    NAME##_frag() {                                                            \
      WRSH_FRAG_WIRING()                                                       \
      WRSH_FRAG_WIRING_PERSPECTIVE()                                           \
    }                                                                          \
  };                                                                           \
  WRSH_PROGRAM(NAME, KEYSTR)

WRSH_BRUSH_BLEND(brush_blend, "brush_blend", false)
WRSH_BRUSH_BLEND(brush_blend_ALPHA_PASS, "brush_blend ALPHA_PASS", true)
