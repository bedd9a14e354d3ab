"""The reference's own benchmark set (wrench/benchmarks/benchmarks.list) as frames: what WebRender's frame builder hands the draw
backend for each display list, restated like scenes.py restates the BASELINE configs (no Rust toolchain here: the frame
builder cannot run).  The display lists themselves -- bounds, colours, text -- come from webrender_amd/wrench/benchmarks.json,
generated from the YAML files by tests/golden/make_wrench_benchmarks.py.

  many-images          8192 opaque 8x8 images, each its own texture-cache entry, one brush_image batch in the opaque pass
                       (prepare.rs image prims -> batch.rs:2183-2300; shared texture cache: 16x16 slabs, texture_cache.rs)
  aligned-gradient     10 x full-size two-stop linear gradient.  With swgl the gradient brush is used, not a cached task
  unaligned-gradient   (scene_building.rs:3389-3396: `cached = (!is_software || is_tiled) && ...`); the axis-aligned
                       decomposition (prim_store/gradient/linear.rs:115-335) leaves this gradient as ONE two-stop segment
                       covering the prim.  Opaque stops => opaque pass, front to back: nine of the ten are depth-rejected
  text-rendering       68 text runs, sizes 8-20 pt (10.7-26.7 px), black / red / green / blue (ps_text_run, R8 glyph atlas; glyph bitmaps:
                       the FreeType fixture, like cfg3's)
  large-blur-radius    a stacking context under filter blur(100, 100): picture task -> 5 x cs_scale -> cs_blur V / H (RGBA8) -> brush_image
                       with RasterizationSpace::Screen uv (picture.rs:5872-5930, render_task.rs:1168-1260, batch.rs:1509-1557)
  many-box-shadows     9 of the 10 outset box shadows (the cards': blur radius 45, offset (0, 22.5), no corner radii, colour rgba(0,0,0,0.102),
                       each under its item clip rect: ONE cached blurred minimal shadow (BoxShadowCacheKey depends on blur
                       radius and radii only), then a cs_clip_box_shadow x cs_clip_rectangle clip-out mask and masked
                       brush_solid segments per shadow (scenes.cfg4_box_shadow generalised: box_shadow.rs, render_task.rs:1168-1194)

Window: BASELINE's 4K target (3840x2160), device pixel scale 1, the display list in the top-left corner as wrench lays it out.
"""
import json
import os
import numpy as np
from . import glconst as G
from .frame import Frame, Step, Target, TextureRef, CompositeTile, CLIP_TASK_EMPTY
from . import scenes
from .scenes import TILE_W, TILE_H, BIG, tile_grid, premultiply

# wrench's named colours (yaml_helper.rs:55-66 string_to_color: "green" is (0, 1, 0), not the CSS keyword's (0, 0.5, 0))
CSS = {"red": (255, 0, 0, 255), "green": (0, 255, 0, 255), "blue": (0, 0, 255, 255), "white": (255, 255, 255, 255), "black": (0, 0, 0, 255),
       "yellow": (255, 255, 0, 255), "cyan": (0, 255, 255, 255), "magenta": (255, 0, 255, 255), "transparent": (255, 255, 255, 0)}
_DATA = None


def display_lists():
    global _DATA
    if _DATA is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wrench", "benchmarks.json")) as f:
            _DATA = json.load(f)
    return _DATA


def _tiles(frame, width, height, tile_filter, dps=1.0):
    """picture-cache tiles of the window: yields (target, picture task address, device rect) and registers the composite"""
    out = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), dps, (float(ox), float(oy)))
        out.append((target, task, (x0, y0, x1, y1)))
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    return out


def many_images(width=3840, height=2160, tile_filter=None, count=None, **kw):
    d = display_lists()["many-images"]
    n, cols, sz = count or d["count"], d["cols"], d["size"]
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    # shared texture cache: every 8x8 image in a 16x16 slab of one 2048^2 RGBA8 texture (texture_cache.rs SlabSize::new)
    atlas, slab = 2048, 16
    per_row = atlas // slab
    pix = np.zeros((atlas, atlas, 4), np.uint8)
    i = np.arange(n)
    ax, ay = (i % per_row) * slab, (i // per_row) * slab
    for k in range(sz):
        for j in range(sz):
            pix[ay + k, ax + j] = np.stack([np.zeros(n, np.int64), i // cols, i % cols, np.full(n, 255)], axis=1).astype(np.uint8)   # B, G, R, A
    t_atlas = TextureRef("shared_cache_rgba8", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, pixels=pix, upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)
    res = [frame.gpu_cache.push([[float(ax[k]), float(ay[k]), float(ax[k] + sz), float(ay[k] + sz)], [0.0, 0.0, 0.0, 0.0]]) for k in range(n)]
    spec = frame.gpu_cache.push([[1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [-1.0, -1.0, 0.0, 0.0]])      # ImageBrushData: white, no tiling
    tiles = _tiles(frame, width, height, tile_filter)
    for target, task, (x0, y0, x1, y1) in tiles:
        op = []
        if sz * cols <= x0 or sz * ((n + cols - 1) // cols) <= y0:
            continue
        for k in range(n):
            c, r = k % cols, k // cols
            rect = (float(sz * c), float(sz * r), float(sz * c + sz), float(sz * r + sz))
            if not (rect[0] < x1 and rect[2] > x0 and rect[1] < y1 and rect[3] > y0):
                continue
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), k + 1, spec, 0, task, (4 | (1 << 16), 0, 65535, 0))
            op.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, resource_address=res[k]))
        if op:
            target.opaque.append(Step("brush_image TEXTURE_2D", "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32), None, "opaque",
                                      textures={0: t_atlas}))
    return _finish(frame, tiles)


def _finish(frame, tiles):
    frame.passes.append([t for t, _, _ in tiles])
    return frame


def linear_gradients(name, width=3840, height=2160, tile_filter=None, **kw):
    items = display_lists()[name]
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    prims = []
    for it in items:
        x, y, w, h = it["bounds"]
        stops = [(float(it["stops"][k]), tuple(v / 255.0 for v in CSS[it["stops"][k + 1]])) for k in range(0, len(it["stops"]), 2)]
        sp, ep = it["start"], it["end"]
        reverse = sp[0] > ep[0] or (sp[0] == ep[0] and sp[1] > ep[1])          # scene_building.rs:3369-3378
        if reverse:
            sp, ep = ep, sp
        lut = frame.gpu_buffer_f.push(scenes.build_gradient_lut(stops, reverse=reverse))
        spec = frame.gpu_cache.push([[sp[0], sp[1], ep[0], ep[1]], [1.0 if it["repeat"] else 0.0, w, h, 0.0]])
        prims.append(((x, y, x + w, y + h), spec, lut))
    tiles = _tiles(frame, width, height, tile_filter)
    for target, task, (x0, y0, x1, y1) in tiles:
        op = []
        for zi, (rect, spec, lut) in enumerate(prims):
            if not (rect[0] < x1 and rect[2] > x0 and rect[1] < y1 and rect[3] > y0):
                continue
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, spec, 0, task, (lut, 0, 0, 0))
            op.append(frame.brush_instance(ph, CLIP_TASK_EMPTY))
        if op:
            target.opaque.append(Step("brush_linear_gradient", "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32), None, "opaque"))
    return _finish(frame, tiles)


def text_rendering(width=3840, height=2160, tile_filter=None, **kw):
    return _text_runs(display_lists()["text-rendering"], width, height, tile_filter)


def overlapping_text_shadows(width=3840, height=2160, tile_filter=None, **kw):
    """overlapping-text-shadows.yaml (in the benchmark directory, not in benchmarks.list): 200 shadows without blur, offset (i, i),
    around one 60 px string.  A shadow whose blur is a no-op makes no picture: pop_all_shadows adds the text run once per shadow,
    in the shadow's colour at the shadow's offset, straight to the draw list, in the shadows' order, and then the run itself
    (scene_building.rs:2917-2994 `blur_is_noop` -> add_primitive_to_draw_list; create_shadow_prim) -- 201 runs of the same glyphs."""
    it = display_lists()["overlapping-text-shadows"]
    ox, oy = it["origin"]
    items = [{"text": it["text"], "origin": [ox + i, oy + i], "size": it["size"], "color": it["shadow-color"]} for i in range(int(it["shadow-count"]))]
    items.append({"text": it["text"], "origin": [ox, oy], "size": it["size"], "color": it["color"]})
    return _text_runs(items, width, height, tile_filter)


def _text_runs(items, width, height, tile_filter):
    """`text:` items of a benchmark display list: sizes are points (yaml_helper.rs:267-269 as_pt_to_f32: size * 16 / 12 pixels), the
    string is laid out as wrench does (wrench.rs:320-382 layout_simple_ascii: the pen advances by the glyph's advance, or by size / 3
    for a glyph without pixels)."""
    fx = scenes.glyph_fixture()
    font = "FreeSans.ttf"
    px_of = lambda t: float(t["size"]) * 16.0 / 12.0
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    atlas = np.zeros((scenes.ATLAS_SIZE, scenes.ATLAS_SIZE), np.uint8)
    ax = ay = 1
    shelf = 0
    res_addr = {}
    for size in sorted({px_of(t) for t in items}):
        for c in sorted({ord(ch) for t in items if px_of(t) == size for ch in t["text"]}):
            left, top, bmp, _adv = fx.char(font, size, c)
            if bmp is None:
                continue
            h, w = bmp.shape
            if ax + w + 1 > scenes.ATLAS_SIZE:
                ax, ay, shelf = 1, ay + shelf + 1, 0
            assert ay + h + 1 <= scenes.ATLAS_SIZE
            atlas[ay:ay + h, ax:ax + w] = bmp
            res_addr[(size, c)] = frame.add_glyph_resource((float(ax), float(ay), float(ax + w), float(ay + h)), (float(left), float(-top)), 1.0)
            ax += w + 1
            shelf = max(shelf, h)
    atlas_ref = TextureRef("glyph_atlas_r8", scenes.ATLAS_SIZE, scenes.ATLAS_SIZE, G.GL_R8, G.GL_LINEAR, pixels=atlas, upload_format=G.GL_RED)
    frame.static_textures.append(atlas_ref)
    runs = []
    for zi, t in enumerate(items):
        size = px_of(t)
        pts, glyphs, x = [], [], np.float32(0.0)
        for c in (ord(ch) for ch in t["text"]):
            if (size, c) in res_addr:
                pts.append((float(x), 0.0))
                glyphs.append(c)
                x = np.float32(x + np.float32(fx.char(font, size, c)[3]))
            else:                       # no pixels (space): a rough estimate, as wrench's
                x = np.float32(x + np.float32(size / 3.0))
        x = float(x)
        color = premultiply(np.array([list(CSS[t["color"] or "black"])], np.uint8))[0]
        ox, oy = t["origin"]
        bb = (ox - 2 * size, oy - 1.5 * size, ox + x + 2 * size, oy + size)
        runs.append((frame.add_text_run(color, pts), (ox, oy), size, glyphs, bb, zi + 1))
    tiles = _tiles(frame, width, height, tile_filter)
    n_glyphs = 0
    for target, task, (x0, y0, x1, y1) in tiles:
        inst = []
        for addr, origin, size, glyphs, bb, z in runs:
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            ph = frame.add_prim_header((origin[0], origin[1], 0.0, 0.0), (-BIG, -BIG, BIG, BIG), z, addr, 0, task, (65535, 0, 0, 0))
            inst += [frame.glyph_instance(ph, gi, res_addr[(size, c)]) for gi, c in enumerate(glyphs)]
        if inst:
            target.alpha.append(Step("ps_text_run ALPHA_PASS,TEXTURE_2D", "PRIM_INSTANCES", np.array(inst, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha", textures={0: atlas_ref}))
            n_glyphs += len(inst)
    frame.n_glyphs = n_glyphs
    return _finish(frame, tiles)


def many_box_shadows(width=3840, height=2160, tile_filter=None, **kw):
    items = display_lists()["many-box-shadows"]
    # (the nine cards' shadows share every parameter but the box; the tenth item -- the 1.5 px shadow of the header bar, a
    # second cache entry and chain of its own -- is left out)
    items = [it for it in items if it["blur-radius"] == items[0]["blur-radius"]]
    assert len(items) == 9 and len({(it["clip-mode"], tuple(it["offset"]), tuple(it["color"])) for it in items}) == 1
    it0 = items[0]
    boxes = [(b[0], b[1], b[0] + b[2], b[1] + b[3]) for b in (it["box-bounds"] for it in items)]
    c = it0["color"]
    color = (int(round(c[0] * 255)), int(round(c[1] * 255)), int(round(c[2] * 255)), int(round(c[3] * 255)))
    zero = ((0.0, 0.0),) * 4
    clips = [(c[0], c[1], c[0] + c[2], c[1] + c[3]) for c in (it["clip-rect"] for it in items)]
    return scenes.cfg4_box_shadow(width=width, height=height, dps=1.0, tile_filter=tile_filter, boxes=boxes, clip_rects=clips,
                                  blur_radius=it0["blur-radius"], radii=zero, shadow_color=color, offset=tuple(it0["offset"]))


def large_boxshadow_ellipse(width=3840, height=2160, tile_filter=None, **kw):
    """large-boxshadow-ellipse.yaml: one outset box shadow of a 1024^2 box, blur radius 10, elliptical corner radii -- the box-shadow
    chain of cfg4 (scenes.cfg4_box_shadow: compute_box_shadow_parameters per axis, clip.rs:1765-1856) on its parameters."""
    it = display_lists()["large-boxshadow-ellipse"]
    b, r = it["bounds"], it["border-radius"]
    radii = tuple((r[k][0], r[k][1]) for k in ("top-left", "top-right", "bottom-left", "bottom-right"))
    return scenes.cfg4_box_shadow(width=width, height=height, dps=1.0, tile_filter=tile_filter, boxes=[(b[0], b[1], b[0] + b[2], b[1] + b[3])],
                                  blur_radius=it["blur-radius"], radii=radii, shadow_color=CSS[it["color"]], offset=(0.0, 0.0))


def large_clip_rect(width=3840, height=2160, tile_filter=None, name="large-clip-rect", **kw):
    """clip-clear.yaml (name="clip-clear"; in the benchmark directory, not in benchmarks.list) has the same shape: 11 rects of 300 x 300
    at (50, 50) under a rounded clip of radius 50 -- above MIN_BRUSH_SPLIT_AREA, so segmented the same way; 44 corner masks of 50 x 50
    in a 2048^2 alpha target that is cleared whole ("large but has low usage": what that benchmark is about).
    large-clip-rect.yaml: N identical opaque rects under one rounded-rectangle clip (radius 16).  What the frame builder makes of
    each rect (prepare.rs build_segments_if_needed / update_clip_task_for_brush, segment.rs:397-470, 511-650): the clip's nine-patch
    splits the brush into 3 x 3 segments, emitted row by row; the four corner segments carry a mask -- one 16 x 16 clip-mask task
    each (cs_clip_rectangle FAST_PATH: uniform radius), per primitive, in an alpha target of the pass before -- and are drawn in
    the alpha pass (brush_solid ALPHA_PASS under swgl_clipMask); the other five are opaque and go to the opaque pass, front to
    back (batch.rs add_segmented_prim_to_batch: needs_blending = mask present)."""
    it = display_lists()[name]
    n, rad = int(it["count"]), float(it["radius"])
    rb, cb = it["rect-bounds"], it["complex-rect"]
    rect = (rb[0], rb[1], rb[0] + rb[2], rb[1] + rb[3])
    crect = (cb[0], cb[1], cb[0] + cb[2], cb[1] + cb[3])
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    # segments of the prim rect cut by the clip's inner rect (extract_inner_rect_safe: the rect less its radii)
    xs = [rect[0], max(rect[0], crect[0] + rad), min(rect[2], crect[2] - rad), rect[2]]
    ys = [rect[1], max(rect[1], crect[1] + rad), min(rect[3], crect[3] - rad), rect[3]]
    segs = [(xs[i], ys[j], xs[i + 1], ys[j + 1], (i != 1 and j != 1)) for j in range(3) for i in range(3)]
    color = premultiply(np.array([list(CSS[it["color"]])], np.uint8))[0]
    # pass 0: the corner masks, 4 per prim, packed in one R8 alpha target
    side = int(np.ceil(rad))
    per_row = 32
    atlas = 1 << int(np.ceil(np.log2(max(per_row * (side + 2) + 8, 256))))
    t_masks = TextureRef("clip_corner_masks", atlas, atlas, G.GL_R8, G.GL_LINEAR, render_target=True)
    tg_m = Target(t_masks, "alpha", clear_color=(1.0, 1.0, 1.0, 1.0))
    minst, mask_task = [], {}
    radii = ((rad, rad),) * 4
    k = 0
    for pi in range(n):
        for si, sg in enumerate(segs):
            if not sg[4]:
                continue
            x0, y0 = 4 + (k % per_row) * (side + 2), 4 + (k // per_row) * (side + 2)
            task = (float(x0), float(y0), float(x0 + int(np.ceil(sg[2]) - np.floor(sg[0]))), float(y0 + int(np.ceil(sg[3]) - np.floor(sg[1]))))
            so = (float(np.floor(sg[0])), float(np.floor(sg[1])))
            minst.append(scenes.clip_rect_instance(task, so, 1.0, (crect[0], crect[1]), (crect[2] - crect[0], crect[3] - crect[1]), radii, 0))
            mask_task[(pi, si)] = frame.add_render_task(task, 1.0, so)
            k += 1
    tg_m.steps.append(Step("cs_clip_rectangle FAST_PATH", "CLIP_RECT", np.concatenate(minst), None, "none"))
    frame.passes.append([tg_m])
    blocks = [list(color)]
    for sg in segs:
        blocks += [[sg[0] - rect[0], sg[1] - rect[1], sg[2] - rect[0], sg[3] - rect[1]], [0.0, 0.0, 0.0, 0.0]]
    spec = frame.gpu_cache.push(blocks)
    tiles = _tiles(frame, width, height, tile_filter)
    for target, task, (x0, y0, x1, y1) in tiles:
        if not (rect[0] < x1 and rect[2] > x0 and rect[1] < y1 and rect[3] > y0):
            continue
        op, al = [], []
        for pi in range(n):
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), pi + 1, spec, 0, task, (65535, 0, 0, 0))
            for si, sg in enumerate(segs):
                if sg[4]:
                    al.append(frame.brush_instance(ph, mask_task[(pi, si)], segment=si))
                else:
                    op.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, segment=si))
        target.opaque.append(Step("brush_solid", "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32), None, "opaque"))
        target.alpha.append(Step("brush_solid ALPHA_PASS", "PRIM_INSTANCES", np.array(al, dtype=np.int32), "PremultipliedAlpha", "alpha",
                                 textures={9: t_masks}))
    frame.readback = [t_masks]
    return _finish(frame, tiles)


def large_blur_radius(width=3840, height=2160, tile_filter=None, **kw):
    """large-blur-radius.yaml: a stacking context with `filters: blur(100, 100)` over one 1024^2 red rect.  What the frame builder
    makes of it:
      * the picture's surface (picture.rs get_surface_rects, the generic arm :7799-7826, and PictureCompositeMode::get_rect
        :4241-4253): the rect inflated by ceil(100) * BLUR_SAMPLE_SCALE = 300 on every side, clipped by the parent's clipping rect
        inflated the same way, and by itself again -- here the whole inflated rect, 1624^2 device pixels, off-screen parts included;
      * the picture task (:5872-5920): its size adjusted for the down-scaling passes (BlurTask::adjusted_blur_source_size,
        render_task.rs:264-278: 1624 -> ceil(1624 / 32) * 32 = 1632), UvRectKind::Quad = where the unclipped rect's corners fall in
        the adjusted task (calculate_uv_rect_kind, :7911-7941);
      * RenderTask::new_blur (render_task.rs:1168-1260): std deviation 100 > MAX_BLUR_STD_DEVIATION = 4 -> five cs_scale halvings
        (816, 408, 204, 102, 51: `blur_target_size / scale_factor` truncated), then cs_blur COLOR_TARGET vertical + horizontal at
        51^2 with std deviation 3.125 and blur_region 1624 / 32 = 50;
      * the composite (batch.rs:1509-1557): ONE brush_image ALPHA_PASS instance per tile the picture touches, premultiplied alpha,
        ImageBrushData { color_mode: Image, raster_space: Screen, opacity: 1 }, the prim rect = the picture's unclipped rect, the
        image source = the horizontal blur's task rect + the quad."""
    it = display_lists()["large-blur-radius"]
    BLUR_SAMPLE_SCALE, MAX_STD, MIN_RT = 3.0, 4.0, 8
    ox, oy = it["bounds"][0], it["bounds"][1]
    rb = it["rect-bounds"]
    rect = (ox + rb[0], oy + rb[1], ox + rb[0] + rb[2], oy + rb[1] + rb[3])           # the rect in device space (dps 1, translation only)
    std = it["blur"][0]
    assert it["blur"][0] == it["blur"][1] and std <= 300.0                           # (clamp_blur_radius: MAX_BLUR_RADIUS = 300)
    infl = float(np.ceil(std)) * BLUR_SAMPLE_SCALE
    prim = (rect[0] - infl, rect[1] - infl, rect[2] + infl, rect[3] + infl)
    # local clip = the window in the stacking context's space; get_rect(Some(prim & clip)) inflates again, & prim
    clip = (0.0, 0.0, float(width), float(height))
    nc = (max(prim[0], clip[0]) - infl, max(prim[1], clip[1]) - infl, min(prim[2], clip[2]) + infl, min(prim[3], clip[3]) + infl)
    clipped = tuple(float(v) for v in (np.floor(max(nc[0], prim[0])), np.floor(max(nc[1], prim[1])), np.ceil(min(nc[2], prim[2])), np.ceil(min(nc[3], prim[3]))))
    unclipped = prim
    orig = (clipped[2] - clipped[0], clipped[3] - clipped[1])
    # adjusted_blur_source_size
    adj, sf, sd = orig, 1.0, std
    while sd > MAX_STD:
        if adj[0] < MIN_RT or adj[1] < MIN_RT:
            break
        sd *= 0.5
        sf *= 2.0
        adj = (float(np.ceil(orig[0] / sf)), float(np.ceil(orig[1] / sf)))
    task_w, task_h = int(round(adj[0] * sf)), int(round(adj[1] * sf))
    quad = [[(cx - clipped[0]) / task_w, (cy - clipped[1]) / task_h, 0.0, 1.0]
            for cx, cy in ((unclipped[0], unclipped[1]), (unclipped[2], unclipped[1]), (unclipped[0], unclipped[3]), (unclipped[2], unclipped[3]))]
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    zero = (0.0, 0.0, 0.0, 0.0)
    pot = lambda v: 1 << int(np.ceil(np.log2(max(v, 64))))
    # -- the picture: the red rect into a transparent colour target, content origin = clipped.min
    t_pic = TextureRef("blur_picture", pot(task_w), pot(task_h), G.GL_RGBA8, G.GL_LINEAR, render_target=True)
    tg = Target(t_pic, "color", clear_color=zero)
    pic_task = frame.add_render_task((0.0, 0.0, float(task_w), float(task_h)), 1.0, (clipped[0], clipped[1]))
    col = premultiply(np.array([list(CSS[it["color"]])], np.uint8))[0]
    spec = frame.gpu_cache.push([list(col)])
    ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), 1, spec, 0, pic_task, (65535, 0, 0, 0))
    tg.steps.append(Step("brush_solid", "PRIM_INSTANCES", np.array([frame.brush_instance(ph, CLIP_TASK_EMPTY)], dtype=np.int32), None, "none"))
    frame.passes.append([tg])
    frame.readback = [t_pic]
    # -- new_blur: halvings, then vertical + horizontal
    cur_tex, cur_rect, size, sd, sf = t_pic, (0.0, 0.0, float(task_w), float(task_h)), (task_w, task_h), std, 1.0
    k = 0
    while sd > MAX_STD:
        if size[0] < MIN_RT or size[1] < MIN_RT:
            break
        sd *= 0.5
        sf *= 2.0
        size = (int(task_w / sf), int(task_h / sf))
        t_s = TextureRef(f"blur_scale_{k}", pot(size[0]), pot(size[1]), G.GL_RGBA8, G.GL_LINEAR, render_target=True)
        tgs = Target(t_s, "color", clear_color=zero)
        nr = (0.0, 0.0, float(size[0]), float(size[1]))
        inst = np.zeros(1, scenes.SCALE_DTYPE)
        inst["t"][0], inst["s"][0], inst["k"][0] = nr, cur_rect, 1.0
        tgs.steps.append(Step("cs_scale TEXTURE_2D", "SCALE", inst, None, "none", textures={0: cur_tex}))
        frame.passes.append([tgs])
        frame.readback.append(t_s)
        cur_tex, cur_rect = t_s, nr
        k += 1
    region = (int(orig[0]) // int(sf), int(orig[1]) // int(sf))
    t_v = TextureRef("blur_v", pot(size[0]), pot(size[1]), G.GL_RGBA8, G.GL_LINEAR, render_target=True)
    t_h = TextureRef("blur_h", pot(size[0]), pot(size[1]), G.GL_RGBA8, G.GL_LINEAR, render_target=True)
    a_src, a_v, a_h = frame.add_render_task(cur_rect), frame.add_render_task(cur_rect), frame.add_render_task(cur_rect)
    tg_v, tg_h = Target(t_v, "color", clear_color=zero), Target(t_h, "color", clear_color=zero)
    tg_v.steps.append(Step("cs_blur COLOR_TARGET", "BLUR", scenes.blur_instance(a_v, a_src, 1, sd, region), None, "none", textures={0: cur_tex}))
    tg_h.steps.append(Step("cs_blur COLOR_TARGET", "BLUR", scenes.blur_instance(a_h, a_v, 0, sd, region), None, "none", textures={0: t_v}))
    frame.passes.append([tg_v])
    frame.passes.append([tg_h])
    frame.readback += [t_v, t_h]
    # -- the blurred picture over the tiles
    src = frame.gpu_cache.push([[cur_rect[0], cur_rect[1], cur_rect[2], cur_rect[3]], [0.0, 0.0, 0.0, 0.0]] + quad)
    bdata = frame.gpu_cache.push([[1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [-1.0, -1.0, 0.0, 0.0]])
    tiles = _tiles(frame, width, height, tile_filter)
    for target, task, (x0, y0, x1, y1) in tiles:
        if not (unclipped[0] < x1 and unclipped[2] > x0 and unclipped[1] < y1 and unclipped[3] > y0):
            continue
        ph = frame.add_prim_header(unclipped, (-BIG, -BIG, BIG, BIG), 1, bdata, 0, task, (4 | (1 << 16), 1, 65535, 0))     # COLOR_MODE_IMAGE, premultiplied, RASTER_SCREEN
        target.alpha.append(Step("brush_image ALPHA_PASS,TEXTURE_2D", "PRIM_INSTANCES",
                                 np.array([frame.brush_instance(ph, CLIP_TASK_EMPTY, resource_address=src)], dtype=np.int32),
                                 "PremultipliedAlpha", "alpha", textures={0: t_h}))
    return _finish(frame, tiles)


# ---------------------------------------------------------------------------------------------------------------------------
# Frame-builder pieces restated for the inset box shadow of large-boxshadow-ellipse-2.yaml
def _au(v):
    """app units: the interned keys hold lengths as Au (1/60 px), api/src/units.rs Au::from_f32_px"""
    return float(np.float32(np.round(np.float32(v) * np.float32(60.0)) / np.float32(60.0)))


def ensure_no_corner_overlap(radii, size):
    """border.rs:168-215; radii = (TL, TR, BL, BR) as (w, h)"""
    tl, tr, bl, br = radii
    ratio = 1.0
    if size[0] > 0.0:
        for s in (tl[0] + tr[0], bl[0] + br[0]):
            if size[0] < s:
                ratio = min(ratio, size[0] / s)
    if size[1] > 0.0:
        for s in (tl[1] + bl[1], tr[1] + br[1]):
            if size[1] < s:
                ratio = min(ratio, size[1] / s)
    if ratio < 1.0:
        radii = tuple((float(np.float32(r[0]) * np.float32(ratio)), float(np.float32(r[1]) * np.float32(ratio))) for r in radii)
    return radii


def compute_box_shadow_parameters(fract_offset, size, radii, blur_radius):
    """clip.rs:1765-1856 -> dict(minimal rect (x0, y0, w, h), alloc size, stretch modes (0 = stretch, 1 = simple), radii, blur region)"""
    radii = ensure_no_corner_overlap(radii, size)
    fract = (abs(size[0] - np.trunc(size[0])), abs(size[1] - np.trunc(size[1])))
    cw, ch = max(r[0] for r in radii), max(r[1] for r in radii)
    region = float(np.ceil(3.0 * blur_radius))
    uw, uh = max(cw, region), max(ch, region)
    mw, mh = 2.0 * uw + region + fract[0], 2.0 * uh + region + fract[1]
    sx = sy = 0
    if size[0] < mw:
        mw, sx = size[0], 1
    if size[1] < mh:
        mh, sy = size[1], 1
    alloc = (2.0 * region + float(np.ceil(mw)), 2.0 * region + float(np.ceil(mh)))
    return dict(minimal=(region + fract_offset[0], region + fract_offset[1], mw, mh), alloc=alloc, stretch=(sx, sy), radii=radii, region=region)


def new_box_shadow_clip(shadow_rect, radii, blur_radius):
    """ClipItemKind::new_box_shadow (clip.rs:1860-1918): the minimal rect to blur; shadows whose allocation exceeds 2048 are rasterised
    downscaled (original_alloc_size keeps the unscaled size: the shader stretches by it)."""
    fo = (abs(shadow_rect[0] - np.trunc(shadow_rect[0])), abs(shadow_rect[1] - np.trunc(shadow_rect[1])))
    size = (shadow_rect[2] - shadow_rect[0], shadow_rect[3] - shadow_rect[1])
    src = compute_box_shadow_parameters(fo, size, radii, blur_radius)
    src["original_alloc"], src["blur_radius"] = src["alloc"], blur_radius
    m = max(src["alloc"])
    if m > 2048.0:
        k = float(np.float32(2048.0) / np.float32(m))
        sc = tuple((float(np.float32(r[0]) * np.float32(k)), float(np.float32(r[1]) * np.float32(k))) for r in radii)
        orig = src["alloc"]
        src = compute_box_shadow_parameters((fo[0] * k, fo[1] * k), (float(np.float32(size[0]) * np.float32(k)), float(np.float32(size[1]) * np.float32(k))), sc,
                                            float(np.float32(blur_radius) * np.float32(k)))
        src["original_alloc"], src["blur_radius"] = orig, float(np.float32(blur_radius) * np.float32(k))
    return src


def extract_inner_rect_safe(rect, radii):
    """util.rs:650-676 with k = 1"""
    tl, tr, bl, br = radii
    w, h = rect[2] - rect[0], rect[3] - rect[1]
    xl, xr = np.ceil(max(tl[0], bl[0])), np.floor(w - max(tr[0], br[0]))
    yt, yb = np.ceil(max(tl[1], tr[1])), np.floor(h - max(bl[1], br[1]))
    if xl <= xr and yt <= yb:
        return (rect[0] + float(xl), rect[1] + float(yt), rect[0] + float(xr), rect[1] + float(yb))
    return None


def build_segments(prim_rect, items):
    """SegmentBuilder::build (segment.rs:511-650) for items = [(rect, mode, has_mask)], mode in (None, "clip", "clip_out"): the sweep
    over the items' x / y events inside the prim rect -> [(x0, y0, x1, y1, has_mask)] row by row; a segment under a mask-less
    clip-out item is dropped."""
    items = [it for it in items if it[0][0] < prim_rect[2] and it[0][2] > prim_rect[0] and it[0][1] < prim_rect[3] and it[0][3] > prim_rect[1]]
    q = lambda v: int(np.round(np.float32(v) * np.float32(60.0)))
    cl = lambda v, lo, hi: min(max(v, lo), hi)
    xs = sorted({cl(q(v), q(prim_rect[0]), q(prim_rect[2])) for it in items for v in (it[0][0], it[0][2])})
    ys = sorted({cl(q(v), q(prim_rect[1]), q(prim_rect[3])) for it in items for v in (it[0][1], it[0][3])})
    out = []
    for y0, y1 in zip(ys, ys[1:]):
        for x0, x1 in zip(xs, xs[1:]):
            mask, drop = False, False
            for (r, mode, has_mask) in items:
                # active: begin <= segment start < end, in Au (EndClip sorts before BeginClip at one value)
                if q(r[0]) <= x0 < q(r[2]) and q(r[1]) <= y0 < q(r[3]):
                    mask |= has_mask
                    drop |= (mode == "clip_out" and not has_mask)
            if not drop:
                out.append((x0 / 60.0, y0 / 60.0, x1 / 60.0, y1 / 60.0, mask))
    return out


def rounded_rect_items(rect, radii, mode):
    """SegmentBuilder::push_clip_rect with a radius (segment.rs:397-470): the clip's nine-patch, corners carry the mask"""
    inner = extract_inner_rect_safe(rect, radii)
    if inner is None:
        return [(rect, mode, True)]
    p0, p1, p2, p3 = (rect[0], rect[1]), (inner[0], inner[1]), (inner[2], inner[3]), (rect[2], rect[3])
    corners = [(p0[0], p0[1], p1[0], p1[1]), (p2[0], p0[1], p3[0], p1[1]), (p2[0], p2[1], p3[0], p3[1]), (p0[0], p2[1], p1[0], p3[1])]
    others = [(p1[0], p0[1], p2[0], p1[1]), (p2[0], p1[1], p3[0], p2[1]), (p1[0], p2[1], p2[0], p3[1]), (p0[0], p1[1], p1[0], p2[1]), (p1[0], p1[1], p2[0], p2[1])]
    return [(c, mode, True) for c in corners] + [(o, mode, False) for o in others]


def large_boxshadow_ellipse_2(width=3840, height=2160, tile_filter=None, **kw):
    """large-boxshadow-ellipse-2.yaml (benchmarks.list:5): an INSET box shadow of a 1024^2 box, blur radius 10000, elliptical radii of
    400-700 px.  What the frame builder makes of it (box_shadow.rs:303-560 add_box_shadow, the "normal path"):
      * blur radius capped at MAX_BLUR_RADIUS = 300; blur offset ceil(3 * 300) = 900, dest rect = the box inflated by it;
      * the prim is a Rectangle on the BOX (inset shadows draw inside the original primitive) with two clips: the box's rounded
        rect (ClipMode::Clip; its radii scaled by ensure_no_corner_overlap: the bottom widths 600 + 600 > 1024) and the box-shadow
        clip source (BoxShadowClipMode::Inset);
      * new_box_shadow (clip.rs:1860-1918): the minimal nine-patch (2 * 900 + 900 per axis) exceeds the 1024^2 shadow rect on both
        axes -> BoxShadowStretchMode::Simple on both, the whole rect is blurred; its allocation 1800 + 1024 = 2824 > 2048 -> the
        cached shadow is rasterised downscaled by 2048 / 2824 (radii, size and blur radius scaled; alloc 2 * 653 + 743 = 2049);
      * the cached task (render_task.rs:613-700): rounded-rect mask of the minimal rect at 2049^2, new_blur with std deviation
        round(217.56 / 2) = 109 -> five cs_scale halvings (1024 .. 64), cs_blur ALPHA_TARGET V / H at 64^2, std deviation 3.40625;
      * segments (prepare.rs:1585-1660, segment.rs): the box-shadow mask region (dest rect; its inner rect -- the dest rect deflated
        by half the ORIGINAL allocation -- is empty) marks every segment as masked, the rounded rect's nine-patch has a zero-width
        centre column (xl = xr = 512) -> 2 x 3 segments, each with its own clip-mask task holding both clips
        (cs_clip_box_shadow with clip mode 1 / simple stretch, cs_clip_rectangle multiplied on top), all drawn in the alpha pass."""
    it = display_lists()["large-boxshadow-ellipse-2"]
    b, r = it["bounds"], it["border-radius"]
    box = (float(b[0]), float(b[1]), float(b[0] + b[2]), float(b[1] + b[3]))
    raw = tuple((float(r[k][0]), float(r[k][1])) for k in ("top-left", "top-right", "bottom-left", "bottom-right"))
    assert it["clip-mode"] == "inset"
    blur_radius = min(float(it["blur-radius"]), 300.0)
    size = (box[2] - box[0], box[3] - box[1])
    clip_radii = tuple((_au(a), _au(c)) for a, c in ensure_no_corner_overlap(raw, size))       # ClipItemKeyKind::rounded_rect
    blur_offset = float(np.ceil(3.0 * blur_radius))
    dest = (box[0] - blur_offset, box[1] - blur_offset, box[2] + blur_offset, box[3] + blur_offset)
    src = new_box_shadow_clip(box, raw, blur_radius)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    zero = (0.0, 0.0, 0.0, 0.0)
    pot = lambda v: 1 << int(np.ceil(np.log2(max(v, 64))))
    # -- the cached blurred shadow: mask at cache resolution (device pixel scale 1), halvings, V / H blur
    cache = (int(np.round(src["alloc"][0])), int(np.round(src["alloc"][1])))
    sigma = float(np.floor(src["blur_radius"] * 0.5 + 0.5))                  # (blur_radius_dp * content_scale).round() as i32
    t_m0 = TextureRef("bs2_corner_mask", pot(cache[0] + 8), pot(cache[1] + 8), G.GL_R8, G.GL_LINEAR, render_target=True)
    tg0 = Target(t_m0, "alpha", clear_color=zero)
    task0 = (4.0, 4.0, 4.0 + cache[0], 4.0 + cache[1])
    mn = src["minimal"]
    tg0.steps.append(Step("cs_clip_rectangle", "CLIP_RECT", scenes.clip_rect_instance(task0, (0.0, 0.0), 1.0, (mn[0], mn[1]), (mn[2], mn[3]), src["radii"], 0), None, "none"))
    frame.passes.append([tg0])
    frame.readback = [t_m0]
    cur_tex, cur_rect, sz, sf = t_m0, task0, cache, 1.0
    k = 0
    while sigma > 4.0 and min(sz) >= 8:
        sigma *= 0.5
        sf *= 2.0
        sz = (int(cache[0] / sf), int(cache[1] / sf))
        t_s = TextureRef(f"bs2_scale_{k}", pot(sz[0] + 8), pot(sz[1] + 8), G.GL_R8, G.GL_LINEAR, render_target=True)
        tgs = Target(t_s, "alpha", clear_color=zero)
        nr = (4.0, 4.0, 4.0 + sz[0], 4.0 + sz[1])
        inst = np.zeros(1, scenes.SCALE_DTYPE)
        inst["t"][0], inst["s"][0], inst["k"][0] = nr, cur_rect, 1.0
        tgs.steps.append(Step("cs_scale TEXTURE_2D", "SCALE", inst, None, "none", textures={0: cur_tex}))
        frame.passes.append([tgs])
        frame.readback.append(t_s)
        cur_tex, cur_rect = t_s, nr
        k += 1
    region = (cache[0] // int(sf), cache[1] // int(sf))
    t_v = TextureRef("bs2_blur_v", pot(sz[0] + 8), pot(sz[1] + 8), G.GL_R8, G.GL_LINEAR, render_target=True)
    t_cache = TextureRef("bs2_texture_cache", pot(sz[0] + 8), pot(sz[1] + 8), G.GL_R8, G.GL_LINEAR, render_target=True)
    a_src, a_v, a_h = (frame.add_render_task(cur_rect) for _ in range(3))
    tg_v, tg_h = Target(t_v, "alpha", clear_color=zero), Target(t_cache, "alpha", clear_color=zero)
    tg_v.steps.append(Step("cs_blur ALPHA_TARGET", "BLUR", scenes.blur_instance(a_v, a_src, 1, sigma, region), None, "none", textures={0: cur_tex}))
    tg_h.steps.append(Step("cs_blur ALPHA_TARGET", "BLUR", scenes.blur_instance(a_h, a_v, 0, sigma, region), None, "none", textures={0: t_v}))
    frame.passes += [[tg_v], [tg_h]]
    frame.readback += [t_v, t_cache]
    res = frame.gpu_cache.push([list(cur_rect), [0.0, 0.0, 0.0, 0.0]])
    # -- segments and their clip-mask tasks
    oa = src["original_alloc"]
    inner = (dest[0] + 0.5 * oa[0], dest[1] + 0.5 * oa[1], dest[2] - 0.5 * oa[0], dest[3] - 0.5 * oa[1])
    items = [(box, "clip", False), ((-BIG, -BIG, BIG, BIG), "clip", False)] + rounded_rect_items(box, clip_radii, "clip")
    if inner[0] >= inner[2] or inner[1] >= inner[3]:           # push_mask_region with an empty inner rect: the whole region needs the mask
        items.append((dest, None, True))
    else:
        raise NotImplementedError("inset shadow with a non-empty inner rect: eight mask regions + an inner clip-out")
    segs = build_segments(box, items)
    assert all(sg[4] for sg in segs)
    mw = pot(int(max(sg[2] - sg[0] for sg in segs)) * 2 + 16)
    t_masks = TextureRef("bs2_prim_masks", mw, pot(int(sum(sg[3] - sg[1] for sg in segs)) // 2 + 64), G.GL_R8, G.GL_LINEAR, render_target=True)
    tg_m = Target(t_masks, "alpha", clear_color=(1.0, 1.0, 1.0, 1.0))
    bs_inst = np.zeros(len(segs), scenes.BOX_SHADOW_DTYPE)
    rr_inst, seg_task = [], []
    col_y = [4, 4]
    for si, sg in enumerate(segs):
        so = (float(np.floor(sg[0])), float(np.floor(sg[1])))
        tw, th = int(np.ceil(sg[2]) - so[0]), int(np.ceil(sg[3]) - so[1])
        c = si % 2
        x0, y0 = 4 + c * (mw // 2), col_y[c]
        col_y[c] += th + 4
        task = (float(x0), float(y0), float(x0 + tw), float(y0 + th))
        bs_inst["area"][si] = (0.0, 0.0, tw, th)
        bs_inst["origins"][si] = (task[0], task[1], so[0], so[1])
        bs_inst["dps"][si] = 1.0
        bs_inst["res"][si] = (res % 1024, res // 1024)
        bs_inst["src_size"][si] = oa
        bs_inst["mode"][si] = 1                                   # BoxShadowClipMode::Inset
        bs_inst["stretch"][si] = src["stretch"]
        bs_inst["dest"][si] = dest
        rr_inst.append(scenes.clip_rect_instance(task, so, 1.0, (box[0], box[1]), size, clip_radii, 0))
        seg_task.append(frame.add_render_task(task, 1.0, so))
    tg_m.steps.append(Step("cs_clip_box_shadow TEXTURE_2D", "CLIP_BOX_SHADOW", bs_inst, None, "none", textures={0: t_cache}))
    tg_m.steps.append(Step("cs_clip_rectangle", "CLIP_RECT", np.concatenate(rr_inst), "Multiply", "none"))
    frame.passes.append([tg_m])
    frame.readback.append(t_masks)
    # -- the masked brush segments over the tiles
    color = premultiply(np.array([list(CSS[it["color"]])], np.uint8))[0]
    blocks = [list(color)]
    for sg in segs:
        blocks += [[sg[0] - box[0], sg[1] - box[1], sg[2] - box[0], sg[3] - box[1]], [0.0, 0.0, 0.0, 0.0]]
    spec = frame.gpu_cache.push(blocks)
    tiles = _tiles(frame, width, height, tile_filter)
    for target, task, (x0, y0, x1, y1) in tiles:
        if not (box[0] < x1 and box[2] > x0 and box[1] < y1 and box[3] > y0):
            continue
        ph = frame.add_prim_header(box, (-BIG, -BIG, BIG, BIG), 1, spec, 0, task, (65535, 0, 0, 0))
        inst = [frame.brush_instance(ph, seg_task[si], segment=si) for si in range(len(segs))]
        target.alpha.append(Step("brush_solid ALPHA_PASS", "PRIM_INSTANCES", np.array(inst, dtype=np.int32), "PremultipliedAlpha", "alpha",
                                 textures={9: t_masks}))
    return _finish(frame, tiles)


# ---------------------------------------------------------------------------------------------------------------------------
# wrench/reftests/text/*.yaml (BASELINE configs[2]: "wrench reftests/text/ suite"): the reftests whose display lists carry explicit
# glyph indices and positions (no text layout needed), over FreeType-rasterised glyphs of the reftests' own fonts (the fixture of
# tests/golden/make_glyphs.py), as the frame builder hands them to ps_text_run with `options(disable-subpixel)` (alpha glyphs):
#   * a text run is one prim header at its (snapped) reference-frame-relative offset -- here the origin of the enclosing stacking
#     contexts, snapped to device pixels (prim_store/text_run.rs:318-340) -- with the glyphs' layout positions in the run's
#     gpu-cache blocks (text_run.rs:107-132) and one GlyphInstance per glyph that has pixels;
#   * the glyph key of an instance carries the quarter-pixel x offset its device position quantises to (glyph_rasterizer
#     SubpixelOffset::quantize: < 1/8 -> 0, < 3/8 -> 1/4, < 5/8 -> 1/2, < 7/8 -> 3/4, else 0 of the next pixel), the program's
#     snap bias (ps_text_run.glsl:96-108, SubpixelDirection::Horizontal) places the bitmap accordingly;
#   * text shadows with a blur radius (1658.yaml, shadow-cover-1.yaml: scene_building.rs push_shadow -> a picture under
#     Filter::Blur(radius / 2)) are rendered as the shadow-coloured run in a colour task inflated by ceil(std) * 3, blurred V / H
#     (std <= 4: no down-scaling) and composited by brush_image ALPHA_PASS with RasterizationSpace::Screen uv, under the text.
TEXT_REFTESTS = ("text", "long-text", "negative-pos", "non-opaque", "snap-text-offset", "1658", "shadow-cover-1",
                 # round 6: `text:` strings (laid out as wrench does, by the fixture's generator), rects, local clip rects, several
                 # shadows per context with and without blur, several runs per shadow picture
                 "1658-ref", "non-opaque-notref", "shadow", "shadow-ref", "shadow-single", "shadow-cover-2", "shadow-many", "shadow-complex",
                 "two-shadows", "subtle-shadow", "subtle-shadow-ref", "snap-clip", "snap-clip-ref", "subpixel-translate-ref",
                 "shadow-partial-glyph", "shadow-partial-glyph-ref", "allow-subpixel-ref", "diacritics", "diacritics-ref", "transparent-no-aa",
                 "transparent-no-aa-ref", "subpx-bg-mask-ref", "colors", "decorations-ref", "ahem-ref", "shadow-clip-ref", "blank",
                 # solid line decorations (a LineDecoration without a cache key is drawn as a solid rect: scene_building.rs:3171-3210,
                 # get_line_decoration_size is None for LineStyle::Solid), also as members of shadow contexts
                 "decorations", "shadow-atomic", "shadow-atomic-ref", "shadow-ordering", "shadow-ordering-ref", "shadow-clip-rect",
                 "blurred-shadow-local-clip-rect",
                 # text in a reference frame whose transform is a fractional translation
                 "subpixel-translate", "snap-text-offset-ref",
                 # rectangular clip nodes: on a prim (its clip chain does not move with a shadow's offset, its clip rect does), and on a
                 # shadow -- push_shadow leaves the chain on the clip stack until pop-all-shadows (scene_building.rs:2879-2895): it clips the
                 # context's prims and, for a blurred shadow, the PICTURE (after the blur), not the prims inside it
                 "shadow-clip", "shadow-fast-clip", "shadow-fast-clip-ref")


def _css_color(c):
    if isinstance(c, str):
        parts = c.replace(",", " ").split()
        if len(parts) == 1:
            return CSS[c]
        c = [float(v) for v in parts]
    c = list(c)
    if len(c) == 3:
        c.append(1.0)
    return (int(c[0]), int(c[1]), int(c[2]), int(round(float(c[3]) * 255.0)) if float(c[3]) <= 1.0 else int(c[3]))


def _rect_of(v, off=(0.0, 0.0)):
    if isinstance(v, str):
        v = [float(t) for t in v.replace(",", " ").split()]
    x, y, w, h = (float(t) for t in v)
    return (x + off[0], y + off[1], x + w + off[0], y + h + off[1])


def _isect(a, b):
    return (max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3]))


def text_reftest(name="text", width=3840, height=2160, tile_filter=None, **kw):
    """One display list of wrench/reftests/text (the fixture holds its items: tests/golden/make_glyphs.py).  The draw list is put
    together as scene_building.rs does: items outside a shadow context in order; a context's queue of shadows and prims by
    pop_all_shadows (:2897-3047) -- a shadow takes EVERY prim that follows it in the queue: without blur each as a prim of its own in
    the shadow's colour at the shadow's offset, straight into the draw list; with blur all of them in ONE picture under
    Filter::Blur(radius / 2) --, a prim itself when its alpha is not zero."""
    fx = scenes.glyph_fixture()
    items = fx.runs[name]
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    # the glyph atlas of this test: every (font, size, glyph, quarter-pixel variant) the runs need
    atlas = np.zeros((1024, 1024), np.uint8)
    ax = ay = 1
    shelf = 0
    entries = {}

    def resource(font, size, gid, sub):
        nonlocal ax, ay, shelf
        key = (font, size, gid, sub)
        if key not in entries:
            left, top, bmp, _adv = fx.glyph(font, size, gid, sub)
            if bmp is None:
                entries[key] = None
            else:
                h, w = bmp.shape
                if ax + w + 1 > 1024:
                    ax, ay, shelf = 1, ay + shelf + 1, 0
                assert ay + h + 1 <= 1024
                atlas[ay:ay + h, ax:ax + w] = bmp
                entries[key] = frame.add_glyph_resource((float(ax), float(ay), float(ax + w), float(ay + h)), (float(left), float(-top)), 1.0)
                ax += w + 1
                shelf = max(shelf, h)
        return entries[key]

    clip_nodes = {}               # `type: clip` items with plain bounds: id -> rect

    def chain_rect(it, extra=()):
        """the rectangular clip nodes of the item's clip chain and of the enclosing shadows' (`extra`), intersected; None: none"""
        rects = [clip_nodes[i] for i in it.get("clip-chain", [])] + list(extra)
        out = None
        for r in rects:
            out = r if out is None else _isect(out, r)
        return out

    def make_run(it, offset=(0.0, 0.0), color=None, stack=()):
        """a text run prim: glyph variants chosen at its device position (`offset`: a shadow's; `stack`: clip rects of enclosing shadows)"""
        font, size = it["font"], float(it.get("size_px", it["size"]))
        off = it["offsets"]
        so = it["origin_offset"]
        # The prim header of a text run (batch.rs / ps_text_run.glsl:108-120, 170-190): local_rect.p0 = prim origin - the UNSNAPPED
        # reference-frame-relative offset -- with glyph points kept as the yaml gives them (relative to the context) that is the shadow's
        # offset, nothing else --, added to every glyph offset BEFORE the shader floors it; local_rect.p1 = the offset SNAPPED to device
        # pixels through the frame's transform (text_run.rs:318-340, SpaceSnapper::snap_point: to raster space, round, back -- for a
        # translation t: round(so + t) - t), added AFTER the floor.  The glyph key's quarter-pixel variant comes from the point plus
        # the prim offset (text_run.rs:470-477), i.e. without the context's offset and without the transform.
        t = tuple(it.get("translate", (0.0, 0.0)))
        p0 = (float(offset[0]), float(offset[1]))
        p1 = (float(np.floor(so[0] + t[0] + 0.5)) - t[0], float(np.floor(so[1] + t[1] + 0.5)) - t[1])
        ref = (p0[0] + p1[0], p0[1] + p1[1])
        pts = [(float(off[2 * i]), float(off[2 * i + 1])) for i in range(len(it["glyphs"]))]
        insts = []
        for gi, (gid, (gx, gy)) in enumerate(zip(it["glyphs"], pts)):
            fr = (p0[0] + gx) - np.floor(p0[0] + gx)
            sub = 0 if fr < 0.125 else 1 if fr < 0.375 else 2 if fr < 0.625 else 3 if fr < 0.875 else 0
            res = resource(font, size, int(gid), sub)
            if res is not None:
                insts.append((gi, res))
        xs, ys = [p[0] for p in pts] or [0.0], [p[1] for p in pts] or [0.0]
        bb = (ref[0] + t[0] + min(xs) - 2 * size, ref[1] + t[1] + min(ys) - 1.5 * size, ref[0] + t[0] + max(xs) + 2 * size, ref[1] + t[1] + max(ys) + size)
        clip = (-BIG, -BIG, BIG, BIG)
        assert t == (0.0, 0.0) or ("clip-rect" not in it and offset == (0.0, 0.0)), "no clips / shadows under a translated frame here"
        if "clip-rect" in it:
            c = _rect_of(it["clip-rect"], (so[0] + offset[0], so[1] + offset[1]))
            clip = c
            bb = (max(bb[0], c[0]), max(bb[1], c[1]), min(bb[2], c[2]), min(bb[3], c[3]))
        cr = chain_rect(it, stack)
        if cr is not None:
            assert t == (0.0, 0.0)
            clip = _isect(clip, cr)
            bb = _isect(bb, cr)
        return dict(kind="run", pts=pts, ref=ref, color=color if color is not None else _css_color(it.get("color", "black")), insts=insts, bb=bb, clip=clip, xlate=t, lrect=(p0[0], p0[1], p1[0], p1[1]))

    def make_rect(it, offset=(0.0, 0.0), color=None, stack=()):
        """a rect item, or a SOLID line decoration (yaml_frame_reader.rs:845-900: horizontal = [start, baseline, end - start, width], vertical =
        [baseline, start, width, end - start]); `offset` / `color`: a shadow's"""
        so = (it["origin_offset"][0] + offset[0], it["origin_offset"][1] + offset[1])
        if it.get("type") == "line" and "baseline" in it:
            assert it.get("style", "solid") == "solid", "solid line decorations only"
            b, a0, a1, w = float(it["baseline"]), float(it["start"]), float(it["end"]), float(it["width"])
            geom = [a0, b, a1 - a0, w] if it["orientation"] == "horizontal" else [b, a0, w, a1 - a0]
        else:
            geom = it["rect"] if "rect" in it else it["bounds"]
        r = _rect_of(geom, so)
        clip, bb = (-BIG, -BIG, BIG, BIG), r
        if "clip-rect" in it:
            clip = _rect_of(it["clip-rect"], so)
            bb = (max(r[0], clip[0]), max(r[1], clip[1]), min(r[2], clip[2]), min(r[3], clip[3]))
        cr = chain_rect(it, stack)
        if cr is not None:
            clip = _isect(clip, cr)
            bb = _isect(bb, cr)
        return dict(kind="rect", rect=r, color=color if color is not None else _css_color(it.get("color", "black")), bb=bb, clip=clip)

    def make_prim(it, offset=(0.0, 0.0), color=None, stack=()):
        return make_run(it, offset, color, stack) if "glyphs" in it else make_rect(it, offset, color, stack)

    draw, queue = [], None            # the draw list; the open shadow context's queue
    def flush_queue(q):
        # the clip chains the context's shadows pushed (push_shadow): all of them are on the stack while pop_all_shadows makes the shadow prims
        all_shadow_clips = tuple(r for r in (chain_rect(v) for k, v in q if k == "S") if r is not None)
        seen_clips = []               # ... and the ones pushed so far when a prim of the context was added
        while q:
            kind, v = q.pop(0)
            if kind == "S":
                r = chain_rect(v)
                if r is not None:
                    seen_clips.append(r)
                later = [p for k, p in q if k == "P"]
                off = tuple(float(t) for t in v.get("offset", [0, 0]))
                col = _css_color(v.get("color", "black"))
                std = float(v.get("blur-radius", 0)) * 0.5
                if std == 0.0:
                    draw.extend(make_prim(p, off, col, all_shadow_clips) for p in later)
                elif later:
                    assert std <= 4.0, "no down-scaling chain here"
                    pic_clip = None
                    for r2 in all_shadow_clips:
                        pic_clip = r2 if pic_clip is None else _isect(pic_clip, r2)
                    draw.append(dict(kind="pic", std=std, runs=[make_prim(p, off, col) for p in later], clip=pic_clip))
            else:
                r = make_prim(v, stack=tuple(v.get("_stack", ())))
                if r["color"][3] > 0:
                    draw.append(r)
    for it in items:
        t = it.get("type")
        if t == "clip":
            assert "complex" not in it and "bounds" in it, "rectangular clip nodes only"
            clip_nodes[it["id"]] = _rect_of(it["bounds"], it["origin_offset"])
        elif t == "shadow":
            queue = queue if queue is not None else []
            queue.append(("S", it))
        elif t == "pop-all-shadows":
            flush_queue(queue or [])
            queue = None
        elif "glyphs" in it or t in ("rect", "line") or "rect" in it:
            if queue is not None:
                stack = tuple(r for r in (chain_rect(v) for k, v in queue if k == "S") if r is not None)
                queue.append(("P", dict(it, _stack=stack)))
            else:
                flush_queue([("P", it)])
    assert queue is None, "unpopped shadows"

    t_atlas = TextureRef("glyph_atlas_r8", 1024, 1024, G.GL_R8, G.GL_LINEAR, pixels=atlas, upload_format=G.GL_RED)
    frame.static_textures.append(t_atlas)
    pot = lambda v: 1 << int(np.ceil(np.log2(max(v, 64))))
    zero = (0.0, 0.0, 0.0, 0.0)
    frame.readback = []
    # -- the blurred shadow pictures: the shadow-coloured runs in a colour task inflated by ceil(std) * 3, blurred V / H, composited by
    # brush_image ALPHA_PASS with RasterizationSpace::Screen uv at the picture's place in the draw list
    for si, d in enumerate(x for x in draw if x["kind"] == "pic"):
        std = d["std"]
        infl = float(np.ceil(std)) * 3.0
        bbs = [r["bb"] for r in d["runs"]]
        rect = (float(np.floor(min(b[0] for b in bbs))), float(np.floor(min(b[1] for b in bbs))), float(np.ceil(max(b[2] for b in bbs))), float(np.ceil(max(b[3] for b in bbs))))
        clipped = (rect[0] - infl, rect[1] - infl, rect[2] + infl, rect[3] + infl)
        tw, th = int(clipped[2] - clipped[0]), int(clipped[3] - clipped[1])
        t_pic = TextureRef(f"shadow_picture_{si}", pot(tw), pot(th), G.GL_RGBA8, G.GL_LINEAR, render_target=True)
        tg = Target(t_pic, "color", clear_color=zero)
        pic_task = frame.add_render_task((0.0, 0.0, float(tw), float(th)), 1.0, (clipped[0], clipped[1]))
        cur_key, inst = None, []
        def close_pic():
            nonlocal cur_key, inst
            if inst:
                tg.steps.append(Step(cur_key, "PRIM_INSTANCES", np.array(inst, dtype=np.int32), "PremultipliedAlpha", "none",
                                     **({"textures": {0: t_atlas}} if cur_key.startswith("ps_text_run") else {})))
            cur_key, inst = None, []
        for run in d["runs"]:
            scol = premultiply(np.array([list(run["color"])], np.uint8))[0]
            if run["kind"] == "rect":
                key = "brush_solid ALPHA_PASS"
                ph = frame.add_prim_header(run["rect"], run["clip"], 1, frame.gpu_cache.push([list(scol)]), 0, pic_task, (65535, 0, 0, 0))
                new_inst = [frame.brush_instance(ph, CLIP_TASK_EMPTY)]
            else:
                key = "ps_text_run ALPHA_PASS,TEXTURE_2D"
                addr = frame.add_text_run(scol, run["pts"])
                ph = frame.add_prim_header(run["lrect"], run["clip"], 1, addr, 0, pic_task, (65535, 0, 0, 0))
                new_inst = [frame.glyph_instance(ph, gi, res, subpx_dir=1) for gi, res in run["insts"]]
            if key != cur_key:
                close_pic()
                cur_key = key
            inst += new_inst
        close_pic()
        frame.passes.append([tg])
        cur_rect = (0.0, 0.0, float(tw), float(th))
        t_v = TextureRef(f"shadow_blur_v_{si}", pot(tw), pot(th), G.GL_RGBA8, G.GL_LINEAR, render_target=True)
        t_h = TextureRef(f"shadow_blur_h_{si}", pot(tw), pot(th), G.GL_RGBA8, G.GL_LINEAR, render_target=True)
        a_src, a_v, a_h = frame.add_render_task(cur_rect), frame.add_render_task(cur_rect), frame.add_render_task(cur_rect)
        tg_v, tg_h = Target(t_v, "color", clear_color=zero), Target(t_h, "color", clear_color=zero)
        tg_v.steps.append(Step("cs_blur COLOR_TARGET", "BLUR", scenes.blur_instance(a_v, a_src, 1, std, (tw, th)), None, "none", textures={0: t_pic}))
        tg_h.steps.append(Step("cs_blur COLOR_TARGET", "BLUR", scenes.blur_instance(a_h, a_v, 0, std, (tw, th)), None, "none", textures={0: t_v}))
        frame.passes += [[tg_v], [tg_h]]
        frame.readback += [t_pic, t_h]
        quad = [[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]]
        d["src"] = frame.gpu_cache.push([[cur_rect[0], cur_rect[1], cur_rect[2], cur_rect[3]], [0.0, 0.0, 0.0, 0.0]] + quad)
        d["tex"], d["bb"] = t_h, clipped
    bdata = frame.gpu_cache.push([[1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [-1.0, -1.0, 0.0, 0.0]])
    rect_color, xform_ids = {}, {}
    tiles = _tiles(frame, width, height, tile_filter)
    n_glyphs = 0
    for target, task, (x0, y0, x1, y1) in tiles:
        # the tile's alpha pass: the draw list in order, consecutive prims of one program and texture in one batch (batch.rs AlphaBatchList)
        cur_key, cur_inst, cur_tex = None, [], None
        def close():
            nonlocal cur_key, cur_inst, cur_tex
            if cur_inst:
                target.alpha.append(Step(cur_key, "PRIM_INSTANCES", np.array(cur_inst, dtype=np.int32), "PremultipliedAlpha", "alpha",
                                         **({"textures": {0: cur_tex}} if cur_tex is not None else {})))
            cur_key, cur_inst, cur_tex = None, [], None
        for zi, d in enumerate(draw):
            bb = d["bb"]
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            z = zi + 1
            if d["kind"] == "pic":
                key, tex = "brush_image ALPHA_PASS,TEXTURE_2D", d["tex"]
                ph = frame.add_prim_header(d["bb"], d.get("clip") or (-BIG, -BIG, BIG, BIG), z, bdata, 0, task, (4 | (1 << 16), 1, 65535, 0))
                inst = [frame.brush_instance(ph, CLIP_TASK_EMPTY, resource_address=d["src"])]
            elif d["kind"] == "rect":
                key, tex = "brush_solid ALPHA_PASS", None
                if d["color"] not in rect_color:
                    rect_color[d["color"]] = frame.gpu_cache.push([list(premultiply(np.array([list(d["color"])], np.uint8))[0])])
                ph = frame.add_prim_header(d["rect"], d["clip"], z, rect_color[d["color"]], 0, task, (65535, 0, 0, 0))
                inst = [frame.brush_instance(ph, CLIP_TASK_EMPTY)]
            else:
                key, tex = "ps_text_run ALPHA_PASS,TEXTURE_2D", t_atlas
                col = premultiply(np.array([list(d["color"])], np.uint8))[0]
                addr = frame.add_text_run(col, d["pts"])
                tid = 0
                if d.get("xlate", (0.0, 0.0)) != (0.0, 0.0):
                    if d["xlate"] not in xform_ids:
                        m = np.eye(4, dtype=np.float32); m[0, 3], m[1, 3] = d["xlate"]
                        xform_ids[d["xlate"]] = frame.add_transform(m.T, np.linalg.inv(m).T.astype(np.float32), axis_aligned=True)
                    tid = xform_ids[d["xlate"]]
                ph = frame.add_prim_header(d["lrect"], d["clip"], z, addr, tid, task, (65535, 0, 0, 0))
                inst = [frame.glyph_instance(ph, gi, res, subpx_dir=1) for gi, res in d["insts"]]
                n_glyphs += len(inst)
            if key != cur_key or tex is not cur_tex:
                close()
                cur_key, cur_tex = key, tex
            cur_inst += inst
        close()
    frame.n_glyphs = n_glyphs
    return _finish(frame, tiles)


WORKLOADS = {
    **{f"reftest-text-{n}": (lambda n=n, **kw: text_reftest(n, **kw)) for n in TEXT_REFTESTS},
    "large-blur-radius": large_blur_radius,
    "large-boxshadow-ellipse": large_boxshadow_ellipse,
    "large-boxshadow-ellipse-2": large_boxshadow_ellipse_2,
    "large-clip-rect": large_clip_rect,
    "clip-clear": lambda **kw: large_clip_rect(name="clip-clear", **kw),
    "overlapping-text-shadows": overlapping_text_shadows,
    "many-images": many_images,
    "aligned-gradient": lambda **kw: linear_gradients("aligned-gradient", **kw),
    "unaligned-gradient": lambda **kw: linear_gradients("unaligned-gradient", **kw),
    "text-rendering": text_rendering,
    "many-box-shadows": many_box_shadows,
}
DESCRIPTIONS = {
    **{f"reftest-text-{n}": f"wrench reftests/text/{n}.yaml (its text runs -- explicit glyph lists, or strings laid out as wrench does --, rects and shadows over FreeType-rasterised glyphs of the reftest's own font at size * 16 / 12 px, alpha glyphs: ps_text_run, brush_solid, cs_blur + brush_image for blurred shadows)" for n in TEXT_REFTESTS},
    "large-blur-radius": "wrench benchmarks/large-blur-radius.yaml: filter blur(100, 100) over a 1024x1024 rect (the picture in a 1632^2 colour task, five cs_scale halvings, cs_blur COLOR_TARGET V/H at 51^2, the result composited by brush_image ALPHA_PASS with RasterizationSpace::Screen uv)",
    "large-boxshadow-ellipse": "wrench benchmarks/large-boxshadow-ellipse.yaml: one outset box shadow of a 1024x1024 box, blur radius 10, elliptical corner radii (cached blurred corner: mask -> cs_scale -> cs_blur V/H, then the cs_clip_box_shadow x clip-out mask and masked brush_solid segments)",
    "large-boxshadow-ellipse-2": "wrench benchmarks/large-boxshadow-ellipse-2.yaml: one INSET box shadow of a 1024x1024 box, blur radius capped at 300, elliptical radii of 400-700 px (the whole shadow rect blurred downscaled: 2049^2 mask -> five cs_scale halvings -> cs_blur V/H at 64^2; six masked brush_solid segments, each mask = cs_clip_box_shadow in inset / simple-stretch mode x the box's rounded rect)",
    "large-clip-rect": "wrench benchmarks/large-clip-rect.yaml: 8 opaque 1024x1024 rects under one rounded-rectangle clip (radius 16): 3x3 brush segments per rect, 4 corner clip-mask tasks each (cs_clip_rectangle FAST_PATH), opaque + masked alpha pass",
    "clip-clear": "wrench benchmarks/clip-clear.yaml (not in benchmarks.list): 11 opaque 300x300 rects under one rounded-rectangle clip (radius 50): 3x3 brush segments per rect, 44 corner clip-mask tasks of 50x50 in a 2048^2 alpha target cleared whole",
    "overlapping-text-shadows": "wrench benchmarks/overlapping-text-shadows.yaml (not in benchmarks.list): 200 unblurred red shadows at offsets (i, i) + the 60 pt (80 px) string itself = 201 ps_text_run runs of 21 glyphs over one another (FreeType-rasterised FreeSans glyphs)",
    "many-images": "wrench benchmarks/many-images.yaml: 8192 opaque 8x8 images (one texture-cache entry each), brush_image opaque pass",
    "aligned-gradient": "wrench benchmarks/aligned-gradient.yaml: 10 x 1980x1080 two-stop linear gradient (brush_linear_gradient, opaque pass, depth-rejected overdraw)",
    "unaligned-gradient": "wrench benchmarks/unaligned-gradient.yaml: 10 x 1980x1080 two-stop linear gradient off the axis (brush_linear_gradient, opaque pass)",
    "text-rendering": "wrench benchmarks/text-rendering.yaml: 68 text runs, 8-20 pt (10.7-26.7 px: yaml_helper.rs as_pt_to_f32), four colours, strings laid out as wrench lays them out (ps_text_run, R8 glyph atlas of FreeType-rasterised FreeSans glyphs)",
    "many-box-shadows": "wrench benchmarks/many-box-shadows.yaml: its 9 card shadows, blur radius 45, rgba(0,0,0,0.1) (one cached blurred corner: mask -> 2 cs_scale -> cs_blur V/H, then cs_clip_box_shadow x clip-out masks and masked brush_solid segments)",
}
