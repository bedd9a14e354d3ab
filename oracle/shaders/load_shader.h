// ORACLE / TEST INFRASTRUCTURE. Stand-in for the file swgl/build.rs:25-37
// generates: includes every (hand-written) shader header and maps the
// "name FEATURE,FEATURE" key sent through ShaderSourceByName (gl.cc:1431) to
// its program loader.
#include "wr_common.h"
#include "brush_base.h"
#include "ps_quad_textured.h"
#include "brush_solid.h"
#include "composite.h"
#include "ps_clear.h"
#include "ps_text_run.h"
#include "cs_blur.h"
#include "cs_scale.h"
#include "cs_clip_rectangle.h"
#include "cs_clip_box_shadow.h"
#include "brush_image.h"
#include "brush_linear_gradient.h"
#include "brush_blend.h"
#include "ps_quad_mask.h"
#include "brush_opacity.h"
#include "cs_border_solid.h"
#include "cs_border_segment.h"
#include "cs_fast_linear_gradient.h"
#include "cs_line_decoration.h"
#include "cs_linear_gradient.h"
#include "cs_radial_gradient.h"
#include "cs_conic_gradient.h"
#include "ps_quad_radial_gradient.h"
#include "ps_quad_conic_gradient.h"
#include "ps_copy.h"
#include "brush_mix_blend.h"
#include "ps_split_composite.h"

ProgramLoader load_shader(const char* name) {
#define WRSH_ENTRY(KEY, SYM) \
  if (!strcmp(name, KEY)) return SYM##_program::loader;
  WRSH_ENTRY("ps_quad_textured", ps_quad_textured)
  WRSH_ENTRY("brush_solid", brush_solid)
  WRSH_ENTRY("brush_solid ALPHA_PASS", brush_solid_ALPHA_PASS)
  WRSH_ENTRY("composite TEXTURE_2D", composite_TEXTURE_2D)
  WRSH_ENTRY("composite FAST_PATH,TEXTURE_2D", composite_FAST_PATH_TEXTURE_2D)
  WRSH_ENTRY("ps_clear", ps_clear)
  WRSH_ENTRY("ps_copy", ps_copy)
  WRSH_ENTRY("ps_split_composite", ps_split_composite)
  WRSH_ENTRY("ps_text_run ALPHA_PASS,TEXTURE_2D", ps_text_run_ALPHA_PASS_TEXTURE_2D)
  WRSH_ENTRY("ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D",
             ps_text_run_ALPHA_PASS_DUAL_SOURCE_BLENDING_TEXTURE_2D)
  WRSH_ENTRY("ps_text_run ALPHA_PASS,GLYPH_TRANSFORM,TEXTURE_2D", ps_text_run_ALPHA_PASS_GLYPH_TRANSFORM_TEXTURE_2D)
  WRSH_ENTRY("ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,GLYPH_TRANSFORM,TEXTURE_2D",
             ps_text_run_ALPHA_PASS_DUAL_SOURCE_BLENDING_GLYPH_TRANSFORM_TEXTURE_2D)
  WRSH_ENTRY("cs_blur ALPHA_TARGET", cs_blur_ALPHA_TARGET)
  WRSH_ENTRY("cs_blur COLOR_TARGET", cs_blur_COLOR_TARGET)
  WRSH_ENTRY("cs_scale TEXTURE_2D", cs_scale_TEXTURE_2D)
  WRSH_ENTRY("cs_clip_rectangle", cs_clip_rectangle)
  WRSH_ENTRY("cs_clip_rectangle FAST_PATH", cs_clip_rectangle_FAST_PATH)
  WRSH_ENTRY("cs_clip_box_shadow TEXTURE_2D", cs_clip_box_shadow)
  WRSH_ENTRY("brush_image TEXTURE_2D", brush_image_TEXTURE_2D)
  WRSH_ENTRY("brush_image ALPHA_PASS,TEXTURE_2D", brush_image_ALPHA_PASS_TEXTURE_2D)
  WRSH_ENTRY("brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D", brush_image_ALPHA_PASS_DUAL_SOURCE_BLENDING_TEXTURE_2D)
  WRSH_ENTRY("brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_2D",
             brush_image_ALPHA_PASS_ANTIALIASING_DUAL_SOURCE_BLENDING_REPETITION_TEXTURE_2D)
  /* the ADVANCED_BLEND keys differ from the ALPHA_PASS ones by an output layout qualifier only (shared.glsl:86-88) */
  WRSH_ENTRY("brush_image ADVANCED_BLEND,ALPHA_PASS,TEXTURE_2D", brush_image_ALPHA_PASS_TEXTURE_2D)
  WRSH_ENTRY("brush_image ADVANCED_BLEND,ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D",
             brush_image_ALPHA_PASS_ANTIALIASING_REPETITION_TEXTURE_2D)
  WRSH_ENTRY("brush_image ANTIALIASING,REPETITION,TEXTURE_2D", brush_image_ANTIALIASING_REPETITION_TEXTURE_2D)
  WRSH_ENTRY("brush_image ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D",
             brush_image_ALPHA_PASS_ANTIALIASING_REPETITION_TEXTURE_2D)
  WRSH_ENTRY("brush_linear_gradient", brush_linear_gradient)
  WRSH_ENTRY("brush_linear_gradient ALPHA_PASS", brush_linear_gradient_ALPHA_PASS)
  WRSH_ENTRY("brush_blend", brush_blend)
  WRSH_ENTRY("brush_blend ALPHA_PASS", brush_blend_ALPHA_PASS)
  WRSH_ENTRY("brush_mix_blend", brush_mix_blend)
  WRSH_ENTRY("brush_mix_blend ALPHA_PASS", brush_mix_blend_ALPHA_PASS)
  WRSH_ENTRY("brush_opacity", brush_opacity)
  WRSH_ENTRY("brush_opacity ALPHA_PASS", brush_opacity_ALPHA_PASS)
  WRSH_ENTRY("brush_opacity ANTIALIASING", brush_opacity_ANTIALIASING)
  WRSH_ENTRY("brush_opacity ALPHA_PASS,ANTIALIASING", brush_opacity_ALPHA_PASS_ANTIALIASING)
  WRSH_ENTRY("ps_quad_mask", ps_quad_mask)
  WRSH_ENTRY("ps_quad_mask FAST_PATH", ps_quad_mask_FAST_PATH)
  WRSH_ENTRY("cs_border_solid", cs_border_solid)
  WRSH_ENTRY("cs_border_segment", cs_border_segment)
  WRSH_ENTRY("cs_fast_linear_gradient", cs_fast_linear_gradient)
  WRSH_ENTRY("cs_line_decoration", cs_line_decoration)
  WRSH_ENTRY("cs_linear_gradient", cs_linear_gradient)
  WRSH_ENTRY("cs_radial_gradient", cs_radial_gradient)
  WRSH_ENTRY("cs_conic_gradient", cs_conic_gradient)
  WRSH_ENTRY("ps_quad_radial_gradient", ps_quad_radial_gradient)
  WRSH_ENTRY("ps_quad_conic_gradient", ps_quad_conic_gradient)
#undef WRSH_ENTRY
  return nullptr;
}
