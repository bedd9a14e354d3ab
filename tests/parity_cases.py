"""Parity cases shared by the host-simulation (-m "not gpu") and the MI355X (-m gpu) suites."""
from webrender_amd import scenes

_TEXT = dict(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12)

# Depth runs (draw_depth_span, rasterize.h:612-664): every shader family behind opaque occluders -- opaque-pass
# content drawn after a nearer occluder, and the depth-tested alpha pass.  swgl restarts the span shader at every run
# of passing pixels; all of these are held to 0 differing bytes.
OCCLUDED = [
    ("occluded_images", lambda: scenes.add_occluders(scenes.image_grid(), zmax=135)),
    ("occluded_images_nearest", lambda: scenes.add_occluders(scenes.image_grid(nearest=True), zmax=135, seed=8)),
    ("occluded_images_wide", lambda: scenes.add_occluders(scenes.image_grid(width=2048, height=1024, n=300, seed=52), n=120, zmax=330, seed=21)),
    ("occluded_images_masked", lambda: scenes.add_occluders(scenes.image_grid(masked=True), zmax=135, seed=22)),
    ("occluded_gradients", lambda: scenes.add_occluders(scenes.gradient_grid(), zmax=80, seed=9)),
    ("occluded_filters", lambda: scenes.add_occluders(scenes.filter_grid(ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]), zmax=72, seed=10)),
    ("occluded_opacity", lambda: scenes.add_occluders(scenes.filter_grid(shader="opacity"), zmax=72, seed=19)),
    ("occluded_image_repeat", lambda: scenes.add_occluders(scenes.image_repeat(), zmax=80, seed=11)),
    ("occluded_image_repeat_nearest", lambda: scenes.add_occluders(scenes.image_repeat(nearest=True), zmax=80, seed=23)),
    ("occluded_rotated_rects", lambda: scenes.add_occluders(scenes.rotated_rects(), zmax=70, seed=12)),
    ("occluded_rotated_rects_quad", lambda: scenes.add_occluders(scenes.rotated_rects(encoding="quad"), zmax=70, seed=24)),
    ("rotated_rects_opaque_pass", lambda: scenes.rotated_rects(opaque_frac=0.5, seed=96)),       # rotated depth writers
    ("occluded_rotated_images", lambda: scenes.add_occluders(scenes.rotated_images(), zmax=60, seed=13)),
    ("occluded_rotated_images_repeat", lambda: scenes.add_occluders(scenes.rotated_images(repeat=True), zmax=60, seed=14)),
    ("occluded_rotated_images_masked", lambda: scenes.add_occluders(scenes.rotated_images(masked=True), zmax=60, seed=25)),
    ("occluded_rotated_images_quad", lambda: scenes.add_occluders(scenes.rotated_images(encoding="quad"), zmax=60, seed=20)),
    ("occluded_text", lambda: scenes.add_occluders(scenes.cfg3_text(**_TEXT), zmax=100, seed=15)),
    ("occluded_text_zoom", lambda: scenes.add_occluders(scenes.cfg3_text(glyph_zoom=1.25, **_TEXT), zmax=100, seed=16)),
    ("occluded_text_modes", lambda: scenes.add_occluders(scenes.cfg3_text(color_modes=(0, 1, 2, 3), **_TEXT), zmax=100, seed=26)),
    ("occluded_aa_rects", lambda: scenes.add_occluders(scenes.cfg2_overlapping_rects(width=1024, height=1024, n=120, seed=7, fractional=True,
                                                                                    encoding="brush", aa_edges=15), zmax=120, seed=17)),
    ("occluded_aa_rects_quad", lambda: scenes.add_occluders(scenes.cfg2_overlapping_rects(width=1024, height=1024, n=120, seed=7, fractional=True,
                                                                                         encoding="quad", aa_edges=15), zmax=120, seed=27)),
    ("occluded_masked_rects", lambda: scenes.add_occluders(scenes.masked_rects(), zmax=150, seed=18)),
    ("occluded_masked_rects_rotated_aa", lambda: scenes.add_occluders(scenes.masked_rects(rotate=True, force_aa=True, fractional=True, seed=14), zmax=150, seed=28)),
    # many thin occluders: rows with more depth runs, narrow runs (< 4 px: every pixel through main())
    ("occluded_images_slivers", lambda: scenes.add_occluders(scenes.image_grid(seed=53), n=160, zmax=135, seed=29, wmin=2, wmax=40)),
    ("occluded_gradients_slivers", lambda: scenes.add_occluders(scenes.gradient_grid(seed=66), n=160, zmax=80, seed=30, wmin=2, wmax=40)),
]
OCCLUDED_GOLDEN = ("occluded_images", "occluded_gradients", "occluded_rotated_images", "occluded_rotated_rects", "occluded_image_repeat",
                   "occluded_aa_rects", "occluded_images_slivers")


# One batch of overlapping solids + images per blend key of swgl's table (gl.cc:614-645): the WebRender BlendModes
# (device/gl.rs:3901-4025), MIN / MAX, the constant-colour key and the 15 KHR_blend_equation_advanced equations.
BLEND = [
    ("blend_modes", lambda: scenes.blend_modes()),
    ("blend_modes_wide", lambda: scenes.blend_modes(width=2048, height=1024, per_state=20, seed=112, clear=(0.0, 0.0, 0.0, 0.0))),
    ("blend_modes_opaque_dst", lambda: scenes.blend_modes(seed=113, clear=(0.3, 0.6, 0.1, 1.0))),
    ("blend_modes_occluded", lambda: scenes.add_occluders(scenes.blend_modes(seed=114), zmax=300, seed=31)),
]
BLEND_GOLDEN = ("blend_modes", "blend_modes_wide")


# Shader replays on general quads: gradients, brush_blend / brush_opacity filters and ps_quad_mask clips
# under rotations and skews -- per-row spans and edge interpolants from the quad walk, swgl_antiAlias edges, the base
# shader's span / main() evaluation on each row -- with and without occluders (depth runs) and clip masks.  0 differing bytes.
ROTATED = [
    ("rotated_gradients", lambda: scenes.gradient_grid(rotate=True, seed=66)),
    ("rotated_filters", lambda: scenes.filter_grid(rotate=True, seed=76, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11])),
    ("rotated_filters_masked", lambda: scenes.filter_grid(rotate=True, masked=True, seed=77, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11])),
    ("rotated_opacity", lambda: scenes.filter_grid(rotate=True, shader="opacity", seed=78)),
    ("rotated_quad_masks", lambda: scenes.quad_masks(rotate=True, seed=86)),
    ("occluded_rotated_gradients", lambda: scenes.add_occluders(scenes.gradient_grid(rotate=True, seed=67), zmax=80, seed=32)),
    ("occluded_rotated_filters", lambda: scenes.add_occluders(scenes.filter_grid(rotate=True, seed=79, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]), zmax=72, seed=33)),
    ("occluded_quad_masks", lambda: scenes.add_occluders(scenes.quad_masks(seed=88), zmax=80, seed=35)),
    ("occluded_rotated_quad_masks", lambda: scenes.add_occluders(scenes.quad_masks(rotate=True, seed=87), zmax=80, seed=34)),
]
# ps_quad_radial_gradient / ps_quad_conic_gradient (gradient patterns on the quad path): axis-aligned, rotated, occluded
ROTATED += [
    ("quad_gradients", lambda: scenes.quad_gradients()),
    ("rotated_quad_gradients", lambda: scenes.quad_gradients(rotate=True, seed=182)),
    ("occluded_quad_gradients", lambda: scenes.add_occluders(scenes.quad_gradients(seed=183), zmax=60, seed=36)),
    ("occluded_rotated_quad_gradients", lambda: scenes.add_occluders(scenes.quad_gradients(rotate=True, seed=184), zmax=60, seed=37)),
]
# draw_perspective (rasterize.h:1064-1280, 1449-1547): solid rects under projective transforms -- vertex w differs, screen
# points from pos.xyz / w, depth interpolated per pixel along every span (the occluders' depth test is per sample) -- alone
# and behind / in front of opaque occluders, both encodings.  0 differing bytes.
ROTATED += [
    ("perspective_rects", lambda: scenes.rotated_rects(perspective=True, seed=97)),
    ("perspective_rects_quad", lambda: scenes.rotated_rects(perspective=True, encoding="quad", seed=98)),
    ("occluded_perspective_rects", lambda: scenes.add_occluders(scenes.rotated_rects(perspective=True, seed=99), zmax=70, seed=38)),
    ("occluded_perspective_rects_quad", lambda: scenes.add_occluders(scenes.rotated_rects(perspective=True, encoding="quad", seed=100), zmax=70, seed=39)),
    # ps_quad_textured sampling an atlas under projective transforms: every pixel through main() with v_uv0 = (uv / w) * (1 / gl_FragCoord.w)
    ("perspective_images_quad", lambda: scenes.rotated_images(perspective="all", encoding="quad", seed=102)),
    ("occluded_perspective_images_quad", lambda: scenes.add_occluders(scenes.rotated_images(perspective="all", encoding="quad", seed=103), zmax=60, seed=40)),
    # brush_image (picture composites under 3-D transforms): screen-space-linear uv (v_uv * world w at the vertices, * gl_FragCoord.w per
    # pixel) and, with BRUSH_FLAG_PERSPECTIVE_INTERPOLATION, perspective-correct uv
    ("perspective_images", lambda: scenes.rotated_images(perspective="all", seed=104)),
    ("occluded_perspective_images", lambda: scenes.add_occluders(scenes.rotated_images(perspective="all", seed=105), zmax=60, seed=41)),
    ("perspective_images_masked", lambda: scenes.rotated_images(perspective="all", masked=True, seed=106)),
    # brush_solid under swgl_clipMask and a projective transform (rounded-corner clips on 3-D transformed content)
    ("perspective_masked_rects", lambda: scenes.masked_rects(perspective=True, force_aa=True, seed=15)),
    ("occluded_perspective_masked_rects", lambda: scenes.add_occluders(scenes.masked_rects(perspective=True, force_aa=True, fractional=True, seed=16), zmax=150, seed=42)),
]
# clip_side (rasterize.h:1287-1430, 1490-1544): prims that reach the camera plane (a vertex with w <= 0) are clipped against the
# view volume before they are projected -- near / far first, then x and y where a clipped vertex still has w <= 0 --, which
# turns the quad into a polygon of up to ten vertices with interpolated varyings and a rewritten AA edge mask: solid rects in
# both encodings, images with screen-linear and perspective-correct uv, behind occluders, writing depth, under clip masks.
# 0 differing bytes.
ROTATED += [
    ("near_clipped_rects", lambda: scenes.rotated_rects(perspective="clip", seed=297)),
    ("near_clipped_rects_quad", lambda: scenes.rotated_rects(perspective="clip", encoding="quad", seed=298)),
    ("occluded_near_clipped_rects", lambda: scenes.add_occluders(scenes.rotated_rects(perspective="clip", seed=299), zmax=70, seed=38)),
    ("near_clipped_depth_writers", lambda: scenes.rotated_rects(perspective="clip", opaque_frac=0.5, seed=396)),
    ("near_clipped_images", lambda: scenes.rotated_images(perspective="clip", seed=304)),
    ("near_clipped_images_quad", lambda: scenes.rotated_images(perspective="clip", encoding="quad", seed=302)),
    ("near_clipped_images_masked", lambda: scenes.rotated_images(perspective="clip", masked=True, seed=306)),
    ("occluded_near_clipped_images", lambda: scenes.add_occluders(scenes.rotated_images(perspective="clip", seed=305), zmax=60, seed=41)),
]
# The other programs' perspective inputs (glsl-to-cxx lib.rs:660-741): brush_blend, brush_opacity and brush_linear_gradient under
# projective transforms -- main() on every chunk with the varying divided by the interpolated 1 / w per pixel, with and without
# BRUSH_FLAG_PERSPECTIVE_INTERPOLATION, alone and behind occluders.  0 differing bytes.
ROTATED += [
    ("perspective_filters", lambda: scenes.filter_grid(rotate=True, perspective=True, seed=172)),
    ("perspective_filters_exact", lambda: scenes.filter_grid(rotate=True, perspective=True, seed=173, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11])),
    ("perspective_opacity", lambda: scenes.filter_grid(shader="opacity", rotate=True, perspective=True, seed=174)),
    ("perspective_gradients", lambda: scenes.gradient_grid(rotate=True, perspective=True, seed=162)),
    ("occluded_perspective_filters", lambda: scenes.add_occluders(scenes.filter_grid(rotate=True, perspective=True, seed=175, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]), zmax=60, seed=45)),
    ("occluded_perspective_gradients", lambda: scenes.add_occluders(scenes.gradient_grid(rotate=True, perspective=True, seed=163), zmax=60, seed=46)),
    ("perspective_filters_masked", lambda: scenes.filter_grid(rotate=True, perspective=True, masked=True, seed=176, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11])),
    # brush_image ANTIALIASING,REPETITION: compute_repeated_uvs on the perspective-correct uv (also clipped against the view volume)
    ("perspective_images_repeat", lambda: scenes.rotated_images(perspective="all", repeat=True, seed=108)),
    ("perspective_images_repeat_mixed", lambda: scenes.rotated_images(perspective=True, repeat=True, seed=107)),
    ("near_clipped_images_repeat", lambda: scenes.rotated_images(perspective="clip", repeat=True, seed=109)),
    # ps_text_run (local raster space): glyph quads under rotations / skews and projective transforms, every colour mode, the
    # dual-source program, behind occluders
    ("rotated_text", lambda: scenes.cfg3_text(rotate=True, **_TEXT)),
    ("rotated_text_modes", lambda: scenes.cfg3_text(rotate=True, color_modes=(0, 1, 2, 3), seed=4, **_TEXT)),
    ("occluded_rotated_text", lambda: scenes.add_occluders(scenes.cfg3_text(rotate=True, seed=8, **_TEXT), zmax=100, seed=17)),
    ("perspective_text", lambda: scenes.cfg3_text(perspective=True, **_TEXT)),
    ("perspective_text_modes", lambda: scenes.cfg3_text(perspective=True, color_modes=(0, 1, 2, 3), seed=4, **_TEXT)),
    ("perspective_text_dual", lambda: scenes.cfg3_text(perspective=True, color_modes=(1,), dual_source=True, seed=6, **_TEXT)),
    ("occluded_perspective_text", lambda: scenes.add_occluders(scenes.cfg3_text(perspective=True, seed=7, **_TEXT), zmax=100, seed=15)),
    # ps_quad_mask: rounded-rect clips on quads under projective transforms (vClipLocalPos / its w per lane, fwidth of the quotient)
    ("perspective_quad_masks", lambda: scenes.quad_masks(rotate=True, perspective=True, seed=83)),
    ("occluded_perspective_quad_masks", lambda: scenes.add_occluders(scenes.quad_masks(rotate=True, perspective=True, seed=84), zmax=90, seed=48)),
    # ps_quad_radial_gradient / ps_quad_conic_gradient: the quad patterns' main() on the perspective-correct v_pos
    ("perspective_quad_gradients", lambda: scenes.quad_gradients(rotate=True, perspective=True, seed=185)),
    ("occluded_perspective_quad_gradients", lambda: scenes.add_occluders(scenes.quad_gradients(rotate=True, perspective=True, seed=186), zmax=60, seed=47)),
]
# Flattened depth rows.  A perspective span flattens the depth row it touches (rasterize.h:1222-1232), and swgl then draws every
# LATER depth-tested prim on that row chunk by chunk through main(), from the span start, instead of handing the span shader one
# depth run at a time (:1021-1031).  The setup stage records, per target row, the first depth-tested perspective prim whose span
# touches it (WrTargetDesc::flat_rows); later prims on such a row skip the span shader and the depth runs.  Also what makes
# depth-WRITING perspective prims drawable: nobody needs runs against them.  All 0 differing bytes.
_U = scenes.add_perspective_underlay
FLAT = [
    ("occluded_perspective_images_mixed", lambda: scenes.add_occluders(scenes.rotated_images(perspective=True, encoding="quad", seed=103), zmax=60, seed=40)),
    ("occluded_perspective_images_mixed_brush", lambda: scenes.add_occluders(scenes.rotated_images(perspective=True, seed=203), zmax=60, seed=44)),
    ("perspective_depth_writers", lambda: scenes.rotated_rects(perspective=True, opaque_frac=0.5, seed=196)),
    ("perspective_depth_writers_quad", lambda: scenes.rotated_rects(perspective=True, opaque_frac=0.5, encoding="quad", seed=197)),
    ("perspective_depth_writers_occluded", lambda: scenes.add_occluders(scenes.rotated_rects(perspective=True, opaque_frac=0.5, seed=198), zmax=70, seed=43)),
    ("flat_images", lambda: _U(scenes.image_grid())),
    ("flat_images_nearest", lambda: _U(scenes.image_grid(nearest=True), seed=78)),
    ("flat_images_masked", lambda: _U(scenes.image_grid(masked=True), seed=79)),
    ("flat_images_occluded", lambda: _U(scenes.add_occluders(scenes.image_grid(), zmax=135), seed=80)),
    ("flat_gradients", lambda: _U(scenes.gradient_grid(), seed=81)),
    ("flat_filters", lambda: _U(scenes.filter_grid(ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]), seed=82)),
    ("flat_opacity", lambda: _U(scenes.filter_grid(shader="opacity"), seed=83)),
    ("flat_image_repeat", lambda: _U(scenes.image_repeat(), seed=84)),
    ("flat_text", lambda: _U(scenes.cfg3_text(**_TEXT), seed=85)),
    ("flat_text_modes", lambda: _U(scenes.cfg3_text(color_modes=(0, 1, 2, 3), **_TEXT), seed=86)),
    ("flat_aa_rects", lambda: _U(scenes.cfg2_overlapping_rects(width=1024, height=1024, n=120, seed=7, fractional=True, encoding="brush", aa_edges=15), seed=87)),
    ("flat_masked_rects", lambda: _U(scenes.masked_rects(), seed=88)),
    ("flat_rotated_images", lambda: _U(scenes.rotated_images(), seed=89)),
    ("flat_rotated_gradients", lambda: _U(scenes.gradient_grid(rotate=True, seed=66), seed=90)),
    ("flat_quad_masks", lambda: _U(scenes.quad_masks(seed=88), seed=91)),
    ("flat_quad_gradients", lambda: _U(scenes.quad_gradients(), seed=92)),
    ("flat_rotated_rects", lambda: _U(scenes.rotated_rects(opaque_frac=0.3), seed=93)),
]
# Many depth runs per row: thin opaque slivers at a fixed pitch in front of the image grid's opaque-pass images.
# WR_MAX_RUNS runs per row / WR_MAX_OCC occluders per strip are what the LDS copies hold; rows and strips beyond that keep theirs in the
# flush's pool (round 5: RUN_OVERFLOW below is exact too; only an exhausted pool is reported).
OCCLUDED += [
    ("occluded_images_sliver_fence", lambda: scenes.add_slivers(scenes.image_grid(), pitch=40)),
    ("occluded_images_sliver_fence_dense", lambda: scenes.add_slivers(scenes.image_grid(seed=54), pitch=17)),
]
RUN_OVERFLOW = [
    ("sliver_fence_overflow", lambda: scenes.add_slivers(scenes.image_grid(), pitch=9)),
    ("sliver_fence_overflow_4", lambda: scenes.add_slivers(scenes.image_grid(), pitch=4)),
    # > WR_MAX_OCC occluders on a strip (every image wider than 195 px sits behind 65+ slivers), > WR_MAX_RUNS runs on every row
    ("sliver_fence_overflow_3", lambda: scenes.add_slivers(scenes.image_grid(seed=55), pitch=3)),
    ("sliver_fence_overflow_gradients", lambda: scenes.add_slivers(scenes.gradient_grid(), pitch=3)),
    ("sliver_fence_overflow_text", lambda: scenes.add_slivers(scenes.cfg3_text(width=1024, height=512, lines=24, glyphs_per_line=60), pitch=5, width=1)),
    ("sliver_fence_overflow_rotated_images", lambda: scenes.add_slivers(scenes.rotated_images(), pitch=3)),
]
# wrench/benchmarks/transforms-simple.yaml (the reference's own transform benchmark): both encodings
# the dual-source REPETITION key on rotated / skewed and on projected prims, anti-aliased, with and without clip masks (round 5: the
# general-quad evaluator runs the key's main(), the blend takes both colours and the coverage -- AA_BLEND_KEY / AA_MASK_BLEND_KEY of
# GL_ONE, GL_ONE_MINUS_SRC1_COLOR, blend.h:513-530)
ROTATED += [
    ("rotated_images_repeat_dual", lambda: scenes.rotated_images(repeat=True, dual=True, seed=105)),
    ("rotated_images_repeat_dual_masked", lambda: scenes.rotated_images(repeat=True, dual=True, masked=True, seed=106)),
    ("perspective_images_repeat_dual", lambda: scenes.rotated_images(repeat=True, dual=True, perspective=True, seed=107)),
    ("perspective_images_repeat_dual_masked", lambda: scenes.rotated_images(repeat=True, dual=True, perspective="all", masked=True, seed=108)),
]
ROTATED += [
    ("transforms_simple", lambda: scenes.transforms_simple()),
    ("transforms_simple_quad", lambda: scenes.transforms_simple(encoding="quad")),
]
ROTATED_GOLDEN = ("rotated_text", "perspective_text", "perspective_quad_masks", "perspective_images_repeat", "perspective_quad_gradients", "perspective_filters_exact", "perspective_opacity", "perspective_gradients", "near_clipped_rects", "near_clipped_images", "near_clipped_images_quad", "transforms_simple", "perspective_rects", "occluded_perspective_rects", "perspective_images_quad", "perspective_images", "rotated_gradients", "rotated_filters", "rotated_quad_masks", "quad_gradients", "rotated_quad_gradients")

# ps_split_composite: the polygons a preserve-3d context is split into, composited back to front -- convex
# quads, triangles (two equal points) and both windings under rotations, skews, identity and projective transforms, with and
# without perspective-correct uv, under clip masks, behind occluders, clipped against the view volume, with a nearest sampler.
SPLIT = [
    ("split_composites", lambda: scenes.split_composites()),
    ("split_composites_wide", lambda: scenes.split_composites(width=2048, height=1024, n=140, seed=212)),
    ("split_composites_masked", lambda: scenes.split_composites(masked=True, seed=213)),
    ("split_composites_nearest", lambda: scenes.split_composites(nearest=True, seed=214)),
    ("perspective_split_composites", lambda: scenes.split_composites(perspective="all", seed=215)),
    ("perspective_split_composites_mixed", lambda: scenes.split_composites(perspective=True, seed=216)),
    ("perspective_split_composites_masked", lambda: scenes.split_composites(perspective="all", masked=True, seed=217)),
    ("near_clipped_split_composites", lambda: scenes.split_composites(perspective="clip", seed=218)),
    ("occluded_split_composites", lambda: scenes.add_occluders(scenes.split_composites(seed=219), zmax=60, seed=47)),
    ("occluded_perspective_split_composites", lambda: scenes.add_occluders(scenes.split_composites(perspective="all", seed=220), zmax=60, seed=48)),
]
# ps_text_run GLYPH_TRANSFORM: screen-raster-space text under rotations, skews and scales -- glyphs snapped
# in device space, every span cut to the glyph's raster rect by gl_ClipDistance (clip_distance_range, rasterize.h:564-595), runs
# under local clip rects that cut through their glyphs, every colour mode, the dual-source program, behind occluders.
GLYPH_TRANSFORM = [
    ("glyph_transform_text", lambda: scenes.cfg3_text(glyph_transform=True, **_TEXT)),
    ("glyph_transform_text_modes", lambda: scenes.cfg3_text(glyph_transform=True, color_modes=(0, 1, 2, 3), seed=5, **_TEXT)),
    ("glyph_transform_text_dual", lambda: scenes.cfg3_text(glyph_transform=True, color_modes=(1, 2), dual_source=True, seed=6, **_TEXT)),
    ("glyph_transform_text_dps", lambda: scenes.cfg3_text(glyph_transform=True, device_pixel_scale=1.5, seed=7, **dict(_TEXT, width=1000, height=500))),
    ("occluded_glyph_transform_text", lambda: scenes.add_occluders(scenes.cfg3_text(glyph_transform=True, seed=9, **_TEXT), zmax=100, seed=18)),
]
SPLIT_GOLDEN = ("split_composites", "perspective_split_composites", "near_clipped_split_composites")


# cs_border_solid (first family): solid border segments -- corners with elliptical outer / inner radii,
# adjacent-corner clips, two-colour corners mixed along the colour line, AA on / off, zero widths, edges -- rendered into a
# texture-cache target; the cache texture is read back.  0 differing bytes.
BORDERS = [
    ("border_solid", dict()),
    ("border_solid_many", dict(n=90, seed=132)),
    ("border_solid_small_atlas", dict(n=12, seed=133, atlas=512)),
]

# cs_border_segment: styled corners and edges (double, groove, ridge, inset / outset, black sides), dashed and dotted edges,
# corner dashes and corner dots.
BORDER_SEGMENTS = [
    ("border_segments", dict()),
    ("border_segments_many", dict(n=200, seed=142)),
]

# cs_fast_linear_gradient, cs_linear_gradient and cs_line_decoration (solid / dotted / dashed / wavy, both axes, thin wavy lines through the AA snap)
# tasks in a texture-cache target.
DECORATIONS = [
    ("cache_decorations", dict()),
    ("cache_decorations_many", dict(n_lines=200, n_grads=60, n_lgrads=20, seed=152)),
    ("cache_linear_gradients", dict(n_lines=1, n_grads=1, n_lgrads=200, seed=153)),      # cs_linear_gradient: the span shader with tileRepeat off
    ("cache_radial_gradients", dict(n_lines=1, n_grads=1, n_lgrads=1, n_rgrads=200, seed=162)),   # cs_radial_gradient: swgl_commitRadialGradientRGBA8
    ("cache_all", dict(n_lines=40, n_grads=20, n_lgrads=20, n_rgrads=30, seed=163)),
    # cs_conic_gradient: main() with libm atan2f -- bit-exact on the host build (same libm as the oracle), within 1 LSB on the GPU
    ("cache_conic_gradients", dict(n_lines=1, n_grads=1, n_lgrads=1, n_rgrads=0, n_cgrads=150, seed=171)),
]


# ps_copy: texture-cache copies / batched uploads.  The expected bytes follow from the scene alone -- a
# copy is a copy -- so these cases need no oracle: `copies_expected` applies the CopyInstances in submission order in numpy.
COPIES = [
    ("copies", dict()),
    ("copies_many_small", dict(n=200, seed=192, src_size=256, dst_size=300)),
    ("copies_unchained", dict(n=40, seed=193, chained=False)),
]


def copies_expected(frame):
    """{texture name: uint8 array as read_texture returns it} for a scenes.texture_cache_copies frame"""
    import numpy as np
    tex = {t.name: np.asarray(t.pixels) for t in frame.static_textures}
    for targets in frame.passes:
        for tg in targets:
            ref = tg.texture
            src = tex[tg.steps[0].textures[0].name]
            dst = np.zeros((ref.h, ref.w) + src.shape[2:], np.uint8)
            for e in tg.steps[0].instances:
                sx0, sy0, sx1, sy1 = [int(v) for v in e["src"]]
                dx0, dy0, dx1, dy1 = [int(v) for v in e["dst"]]
                dst[dy0:dy1, dx0:dx1] = src[sy0:sy1, sx0:sx1]
            tex[ref.name] = dst
    return {t.name: tex[t.name] for t in frame.readback}


# brush_mix_blend: the sixteen MixBlendMode values on backdrop / source picture pairs -- 1:1 swatches,
# scaled and fractionally placed prims with linear and nearest sources, sub-quads in homogeneous coordinates, clip masks
MIX_BLEND = [
    ("mix_swatches", "mix_blend_swatches", dict()),
    ("mix_grid", "mix_blend_grid", dict()),
    ("mix_grid_nearest_src", "mix_blend_grid", dict(seed=204, n=60)),
    ("mix_grid_masked", "mix_blend_grid", dict(seed=205, masked=True)),
    ("mix_grid_integer", "mix_blend_grid", dict(seed=207, fractional=False, n=50)),
    # mix-blend-mode on rotated / skewed stacking contexts (the general-quad path, anti-aliased edges) and with BRUSH_FLAG_FORCE_AA
    ("mix_grid_rotated", "mix_blend_grid", dict(seed=208, rotate=True)),
    ("mix_grid_rotated_masked", "mix_blend_grid", dict(seed=209, rotate=True, masked=True, n=60)),
    ("mix_grid_force_aa", "mix_blend_grid", dict(seed=210, force_aa=True, n=60)),
    # ... inside a 3-D context (a projective row on top of the rotation; BRUSH_FLAG_PERSPECTIVE_INTERPOLATION on every other such prim):
    # both varyings interpolated / w, the source's times mix(gl_FragCoord.w, 1, flag) -- round 5
    ("mix_grid_perspective", "mix_blend_grid", dict(seed=211, perspective=True)),
    ("mix_grid_perspective_masked", "mix_blend_grid", dict(seed=212, perspective=True, masked=True, n=60)),
    # ... cut by the near plane (every other perspective prim's projective row is strong enough for w <= 0 on part of it): clip_side
    # carries both varyings (rasterize.h:1402-1500 clips the whole Interpolants struct), the polygon's edges step the second one -- round 6
    ("mix_grid_near_clipped", "mix_blend_grid", dict(seed=213, perspective="clip")),
    ("mix_grid_near_clipped_masked", "mix_blend_grid", dict(seed=214, perspective="clip", masked=True, n=60)),
    ("mix_grid_near_clipped_force_aa", "mix_blend_grid", dict(seed=215, perspective="clip", force_aa=True, n=60)),
]


# brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING: images under GL_ONE, GL_ONE_MINUS_SRC1_COLOR with the second
# colour main() writes -- COLOR_MODE_SUBPX_DUAL_SOURCE, MULTIPLY_DUAL_SOURCE and IMAGE, translucent image colours, clip masks
DUAL_SOURCE = [
    # drop shadows of pictures: brush_image with COLOR_MODE_ALPHA / COLOR_MODE_BITMAP_SHADOW -> swgl_blendDropShadow per prim
    ("image_shadows", dict(shadows=True, seed=54)),
    ("image_shadows_masked", dict(shadows=True, masked=True, seed=55)),
    # RasterizationSpace::Screen sources (what blurred and drop-shadow pictures are composited with): get_image_quad_uv in the vertex stage
    ("image_screen", dict(screen=True, seed=56)),
    ("image_screen_shadows_masked", dict(screen=True, shadows=True, masked=True, seed=57)),
    ("image_dual", dict(dual=True)),
    ("image_dual_masked", dict(dual=True, masked=True, seed=52)),
    ("image_dual_nearest", dict(dual=True, nearest=True, seed=53)),
]
# the REPETITION variant of the dual-source key (shader_features.rs:166-170): tiled / stretched / border-image-segment images under
# the dual-source blend state
REPEAT_DUAL = [
    ("image_repeat_dual", dict(dual=True, seed=58)),
    ("image_repeat_dual_nearest", dict(dual=True, nearest=True, seed=59)),
    # ... on anti-aliased prims (BRUSH_FLAG_FORCE_AA, all four edges: blend.h's AA_BLEND_KEY(GL_ONE, GL_ONE_MINUS_SRC1_COLOR) scales both colours)
    ("image_repeat_dual_aa", dict(dual=True, aa=True, seed=58)),
    ("image_repeat_dual_aa_nearest", dict(dual=True, aa=True, nearest=True, seed=59)),
]


# The reference's own benchmark set (wrench/benchmarks/benchmarks.list), restated by webrender_amd/wrench_scenes.py from the display
# lists in webrender_amd/wrench/benchmarks.json: (name, workload, kwargs of the CPU-sized case, kwargs of the full GPU case)
WRENCH = [
    ("wrench_many_images", "many-images", dict(width=1024, height=512, count=2048), dict()),
    ("wrench_aligned_gradient", "aligned-gradient", dict(width=2048, height=1024), dict()),
    ("wrench_unaligned_gradient", "unaligned-gradient", dict(width=2048, height=1024), dict()),
    ("wrench_text_rendering", "text-rendering", dict(width=2048, height=1024), dict()),
    ("wrench_many_box_shadows", "many-box-shadows", dict(width=2048, height=1536), dict()),
    ("wrench_simple_batching_4k", "simple-batching", None, dict()),
    ("wrench_large_boxshadow_ellipse", "large-boxshadow-ellipse", dict(width=1536, height=1536), dict()),
    ("wrench_large_boxshadow_ellipse_2", "large-boxshadow-ellipse-2", dict(width=1536, height=1536), dict()),      # the inset one, benchmarks.list:5
    ("wrench_large_clip_rect", "large-clip-rect", dict(width=1536, height=1536), dict()),
    ("wrench_large_blur_radius", "large-blur-radius", dict(width=1536, height=1536), dict()),
    # in wrench/benchmarks/ but not in benchmarks.list (radial-gradient.yaml, the third one, cannot be read by the reference's own wrench)
    ("wrench_clip_clear", "clip-clear", dict(width=1024, height=512), dict()),
    ("wrench_overlapping_text_shadows", "overlapping-text-shadows", dict(width=1024, height=512), dict()),
]

# brush_yuv_image (video frames: YUV_FORMAT_PLANAR with three R8 planes, YUV_FORMAT_NV12 with R8 + RG8; the seven YuvRangedColorSpace
# values; opaque and alpha pass; 1:1, fractional, scaled; linear and nearest plane samplers; behind occluders: depth runs)
YUV = [
    ("yuv_grid", lambda: scenes.yuv_grid()),
    ("yuv_grid_nearest", lambda: scenes.yuv_grid(nearest=True, seed=302)),
    ("yuv_grid_wide", lambda: scenes.yuv_grid(width=2048, height=1024, n=150, seed=303)),
    ("occluded_yuv_grid", lambda: scenes.add_occluders(scenes.yuv_grid(seed=304, n=90), zmax=160, seed=41)),
    # 10-bit video: three R16 planes (low bits, rescaled by the span shader) and P010 (R16 + RG16, high bits)
    ("yuv_grid_10bit", lambda: scenes.yuv_grid(hdr=True, seed=305)),
    # "composite TEXTURE_2D,YUV": video surfaces composited straight into the window (scaled, clipped, flipped, blended)
    ("yuv_composites", lambda: scenes.yuv_composites()),
    ("yuv_composites_nearest", lambda: scenes.yuv_composites(nearest=True, seed=312)),
]


# The TEXTURE_RECT keys (brush_image x 8, brush_yuv_image x 2, composite x 3, cs_scale): the rectangle-texture twins of scenes of
# those families (scenes.texture_rect: sampler2DRect samplers bound through GL_TEXTURE_RECTANGLE, unnormalised uv).  Each case also
# has to differ from its TEXTURE_2D twin where the shaders differ (v_uv_bounds of brush_image, the composite fast path's bounds).
TEXTURE_RECT = [
    ("rect_image_grid", lambda: scenes.texture_rect(scenes.image_grid(seed=61))),
    ("rect_image_grid_nearest", lambda: scenes.texture_rect(scenes.image_grid(seed=62, nearest=True))),
    ("rect_image_grid_masked", lambda: scenes.texture_rect(scenes.image_grid(seed=63, masked=True))),
    ("rect_image_repeat", lambda: scenes.texture_rect(scenes.image_repeat(seed=64))),
    ("rect_image_repeat_nearest", lambda: scenes.texture_rect(scenes.image_repeat(seed=65, nearest=True))),
    ("rect_image_dual", lambda: scenes.texture_rect(scenes.image_grid(dual=True, seed=66))),
    ("rect_image_repeat_dual", lambda: scenes.texture_rect(scenes.image_repeat(dual=True, seed=67))),
    ("rect_image_shadows", lambda: scenes.texture_rect(scenes.image_grid(shadows=True, seed=68))),
    ("rect_rotated_images", lambda: scenes.texture_rect(scenes.rotated_images(seed=69))),
    ("rect_rotated_images_repeat", lambda: scenes.texture_rect(scenes.rotated_images(seed=70, repeat=True))),
    ("rect_occluded_image_grid", lambda: scenes.texture_rect(scenes.add_occluders(scenes.image_grid(seed=71), zmax=200, seed=43))),
    ("rect_yuv_grid", lambda: scenes.texture_rect(scenes.yuv_grid(seed=306))),      # planar frames: blendYUV's CompositeYUV-backed overload (swgl_ext.h:1195-1283)
    ("rect_yuv_grid_nv12", lambda: scenes.texture_rect(scenes.yuv_grid(seed=309, planar=False))),
    ("rect_yuv_grid_nearest", lambda: scenes.texture_rect(scenes.yuv_grid(seed=307, nearest=True))),
    ("rect_yuv_grid_10bit", lambda: scenes.texture_rect(scenes.yuv_grid(seed=308, hdr=True))),      # three R16 planes / P010
    ("rect_yuv_grid_wide", lambda: scenes.texture_rect(scenes.yuv_grid(width=2048, height=1024, n=150, seed=310))),
    ("rect_occluded_yuv_grid", lambda: scenes.texture_rect(scenes.add_occluders(scenes.yuv_grid(seed=311, n=90), zmax=160, seed=41))),
    ("rect_yuv_composites", lambda: scenes.texture_rect(scenes.yuv_composites(seed=313))),
    ("rect_yuv_composites_nearest", lambda: scenes.texture_rect(scenes.yuv_composites(seed=314, nearest=True))),
    ("rect_scaled_composites", lambda: scenes.texture_rect(scenes.scaled_composites(seed=22))),
    ("rect_blur_chain_scaled", lambda: scenes.texture_rect(scenes.blur_chain(fmt="rgba8", scale_steps=2, content=(150, 97), sigma=3.0), composites=False)),
    ("rect_blur_chain_r8", lambda: scenes.texture_rect(scenes.blur_chain(fmt="r8", scale_steps=2, content=(166, 140), sigma=2.5), composites=False)),
]


# cs_svg_filter / cs_svg_filter_node: every filter kind main() has a case for (and kinds it has none for), two chained colour
# targets, 1:1 / scaled / fractionally offset inputs; linear and nearest input samplers
SVG_FILTERS = [
    ("svg_filters", lambda: scenes.svg_filters()),
    ("svg_filters_nearest", lambda: scenes.svg_filters(nearest=True, seed=402)),
    ("svg_filter_nodes", lambda: scenes.svg_filters(node=True, seed=403)),
    ("svg_filter_nodes_nearest", lambda: scenes.svg_filters(node=True, nearest=True, seed=404)),
]


# BASELINE configs[2]: wrench/reftests/text -- the reftests with explicit glyph runs (webrender_amd/wrench_scenes.text_reftest) over the
# FreeType fixture of the reftests' own fonts: (name, kwargs of the CPU-sized case, kwargs of the 4K case)
from webrender_amd.wrench_scenes import TEXT_REFTESTS as _TEXT_REFTEST_NAMES      # 46 display lists of the suite (round 6: 7 -> 46)
TEXT_REFTESTS = [(n, dict(width=1024, height=1024), dict()) for n in _TEXT_REFTEST_NAMES]

# Tile rows (wr_tile_rows_kernel): picture targets of a FEW large gradient / image prims -- what the host hands to the row kernel
# instead of the bin raster (<= 24 prims per tile, no general quads) -- alone, behind opaque occluders (depth runs: the span shader
# restarts at every run), behind sliver fences (more runs than a record holds: reported), under clip masks, with a nearest sampler,
# as drop shadows.  Every case has to take that kernel (WrhipStats::row_launches) and is rendered a second time through the bins
# (WRHIP_NO_TILE_ROWS=1): both equal the oracle.
TILE_ROWS = [
    ("tile_rows_gradients", lambda: scenes.gradient_grid(n=14, seed=361)),
    ("tile_rows_gradients_int", lambda: scenes.gradient_grid(n=12, seed=362, fractional=False)),
    ("tile_rows_gradients_occluded", lambda: scenes.add_occluders(scenes.gradient_grid(n=12, seed=363), n=8, zmax=30, seed=51)),
    ("tile_rows_gradients_wide", lambda: scenes.gradient_grid(width=2048, height=1024, n=30, seed=364)),
    ("tile_rows_images", lambda: scenes.image_grid(n=16, seed=365)),
    ("tile_rows_images_occluded", lambda: scenes.add_occluders(scenes.image_grid(n=12, seed=366), n=8, zmax=30, seed=52)),
    ("tile_rows_images_masked", lambda: scenes.image_grid(n=14, seed=367, masked=True)),
    ("tile_rows_images_nearest", lambda: scenes.image_grid(n=14, seed=368, nearest=True)),
    ("tile_rows_image_shadows", lambda: scenes.image_grid(n=12, seed=369, shadows=True)),
    ("tile_rows_images_screen", lambda: scenes.image_grid(n=12, seed=370, screen=True)),
    ("tile_rows_images_slivers", lambda: scenes.add_slivers(scenes.image_grid(n=10, seed=371), pitch=40)),
    ("tile_rows_images_sliver_overflow", lambda: scenes.add_slivers(scenes.image_grid(n=10, seed=372), pitch=9)),
    # solids under clip masks (the corner segments of rounded-rect clips): plain and masked solids, depth-tested against occluders
    ("tile_rows_masked_rects", lambda: scenes.masked_rects(n=40, seed=373)),
    ("tile_rows_masked_rects_frac", lambda: scenes.masked_rects(n=60, seed=374, fractional=True)),
    ("tile_rows_masked_rects_occluded", lambda: scenes.add_occluders(scenes.masked_rects(n=40, seed=375), n=10, zmax=60, seed=53)),
    ("tile_rows_large_clip_rect", lambda: scenes.make_workload("large-clip-rect", width=1536, height=1536)),
]

# ---------------------------------------------------------------------------------------------------------------------------
# Every parity family as (family, name, make) -- what tests/test_clang_budget.py (shipping-flags swgl build) and
# tools/clang_spread.py walk.  SMALL / BLUR / CLIP / BOX live in the test modules that introduced them; they are handed in.
def family_scenes(small, blur, clip, box, wrench_small=True):
    fam = []
    fam += [("small", n, m) for n, m in small]
    fam += [("occluded", n, m) for n, m in OCCLUDED]
    fam += [("blend", n, m) for n, m in BLEND]
    fam += [("rotated", n, m) for n, m in ROTATED]
    fam += [("split", n, m) for n, m in SPLIT]
    fam += [("glyph_transform", n, m) for n, m in GLYPH_TRANSFORM]
    fam += [("flat", n, m) for n, m in FLAT]
    fam += [("blur", n, (lambda kw=kw: scenes.blur_chain(**kw))) for n, kw in blur]
    fam += [("clip", n, (lambda kw=kw: scenes.clip_masks(**kw))) for n, kw in clip]
    fam += [("box_shadow", n, (lambda kw=kw: scenes.box_shadow_masks(**kw))) for n, kw in box]
    fam += [("cfg4", "cfg4_small", lambda: scenes.cfg4_box_shadow(width=1024, height=1024))]
    fam += [("border", n, (lambda kw=kw: scenes.border_solid(**kw))) for n, kw in BORDERS]
    fam += [("border", n, (lambda kw=kw: scenes.border_segments(**kw))) for n, kw in BORDER_SEGMENTS]
    fam += [("decorations", n, (lambda kw=kw: scenes.cache_decorations(**kw))) for n, kw in DECORATIONS]
    fam += [("mix_blend", n, (lambda s=s, kw=kw: getattr(scenes, s)(**kw))) for n, s, kw in MIX_BLEND]
    fam += [("dual_source", n, (lambda kw=kw: scenes.image_grid(**kw))) for n, kw in DUAL_SOURCE]
    fam += [("dual_source", n, (lambda kw=kw: scenes.image_repeat(**kw))) for n, kw in REPEAT_DUAL]
    fam += [("yuv", n, m) for n, m in YUV]
    fam += [("texture_rect", n, m) for n, m in TEXTURE_RECT]
    fam += [("svg_filter", n, m) for n, m in SVG_FILTERS]
    for n, workload, small_kw, full_kw in WRENCH:
        kw = small_kw if wrench_small else full_kw
        if kw is None:
            continue
        fam.append(("wrench", n, (lambda w=workload, kw=kw: scenes.make_workload(w, **kw))))
    return fam


# Budgets against the SHIPPING build of the reference (clang, swgl/build.rs:150-204: -ffast-math -mrecip=none, the SSE2 paths of
# vector_type.h / glsl.h).  libwrhip equals the reference's strict-IEEE g++ build to 0 bytes (every other parity test), so what is
# bounded here is the reference's own spread between its two supported build configurations, per family, measured by
# tools/clang_spread.py (gpurun_out/ -> DESIGN section 7 has the table and the cause of every class of outlier):
#   family -> (max |diff| allowed on ANY byte or None, fraction of bytes allowed above 1 LSB, fraction allowed above 4 LSB, cause)
# Fractions are per scene (every target read back).  The bounds sit 2-3x above the measured worst scene of the family.
CLANG_BUDGET = {
    # integer / fixed-point families: the two builds agree, or differ by the rounding mode of round_pixel (cvtps2dq
    # round-to-nearest-even vs int(v + 0.5)) and the clip shaders' rcp / rsqrt approximations
    "glyph_transform": (0, 0.0, 0.0, "integer glyph blits"),
    "blur": (0, 0.0, 0.0, "8.8 fixed-point taps"),
    "cfg4": (0, 0.0, 0.0, "box-shadow chain: fixed-point sampling of an R8 mask"),
    "clip": (1, 0.0, 0.0, "float coverage of rounded corners: rsqrt approximation, <= 1 LSB"),
    "box_shadow": (1, 0.0, 0.0, "as clip"),
    "border": (1, 0.0, 0.0, "distance-to-ellipse AA in main(): <= 1 LSB"),
    "wrench": (1, 0.0, 0.0, "the reference's own benchmark display lists: <= 1 LSB"),
    # bilinear sampling: uv interpolants are set up under reassociation (fast-math) and then quantised to 1/128 texel, so a 1-ulp
    # uv difference moves a sample by one filter step (<= 4 LSB on the noise textures of these scenes); nearest samplers, repeat
    # wraps, hard gradient stops, discrete transfer tables and depth ties turn the same 1-ulp difference into a different texel /
    # table entry / winner on the locus of pixels that sit exactly on the discontinuity (any magnitude, 1-pixel-wide rows / columns)
    "small": (None, 4e-3, 4e-3, "7-bit filter fractions; tie rows of nearest / repeat sampling; hard stops; discrete tables"),
    "occluded": (None, 3e-3, 3e-3, "as small"),
    "flat": (None, 3e-4, 1.5e-4, "as small"),
    "texture_rect": (None, 2.5e-3, 2.5e-3, "as small"),
    "dual_source": (None, 3e-3, 2.5e-3, "as small (nearest samplers)"),
    "split": (None, 3e-4, 2e-5, "as small, on general quads"),
    "decorations": (None, 5e-5, 2e-5, "gradient table lookups at hard stops / repeat wraps"),
    "svg_filter": (None, 3e-4, 3e-4, "discrete / table component transfer: the table index is floor() of an unpremultiplied colour"),
    "mix_blend": (None, 1e-4, 2e-5, "non-separable blend modes: division by the luminance range"),
    "yuv": (8, 5e-3, 1e-5, "YUV matrix in 16-bit fixed point after float set-up: +-1 on most bytes, <= 5 at scaled plane edges"),
    "rotated": (None, 1.5e-2, 2e-3, "as small, plus radial-gradient offsets through v * rsqrt(v) instead of sqrt(v)"),
    # ill-conditioned blend equations stacked on each other: colour dodge / burn divide by (1 - src) resp. src, so a 1-LSB
    # difference of the destination left by an earlier state is amplified by later ones (each state alone: <= 5 LSB, <= 0.01 %)
    "blend": (None, 2.5e-2, 5e-3, "KHR advanced equations compounding over 23 stacked states"),
    # prims cut by the near plane: a clipped vertex has w -> 0+, its projection is the quotient of two nearly cancelling sums, and
    # the two builds put it pixels apart -- whole slivers of such a polygon are covered in one build and not in the other
    "near_clipped": (None, 0.35, 0.35, "near-plane clipped polygons: vertices at w -> 0 are ill-conditioned"),
    # ... and a mix-blend prim's pixels all depend on where its backdrop's and its source's polygons ended up (measured worst 0.40 + 25 %)
    "near_clipped_mix": (None, 0.5, 0.5, "near-plane clipped polygons under brush_mix_blend: two clipped pictures feed every pixel"),
}


def clang_budget(family, name):
    if family == "mix_blend" and "perspective" in name:      # (two bilinear samplers per pixel under a projective transform: the rotated family's causes)
        return CLANG_BUDGET["rotated"]
    if family == "mix_blend" and "near_clipped" in name:
        return CLANG_BUDGET["near_clipped_mix"]
    return CLANG_BUDGET["near_clipped" if "near_clipped" in name else family]
