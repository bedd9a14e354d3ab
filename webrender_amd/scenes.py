"""Synthetic scenes for BASELINE.json's configs, expressed as built Frames
(frame.py).  They restate just enough of the batcher's *output encodings* to
produce what the Rust frame builder would hand to the renderer for these
display lists (value distributions and seeds from there):

  cfg1  16x16 grid of opaque 64x64 rects @1024^2          seed 1
  cfg2  1000 overlapping translucent rects @3840x2160      seed 2
  cfg5  100k rects (50% opaque) @7680x4320                 seed 5
  simple_batching  wrench/benchmarks/simple-batching.yaml geometry

Picture-cache tiles are 1024x512 (picture.rs:266-270); every tile is its own
render target with its own picture task (task_rect at the texture origin,
content_origin = tile origin), and only primitives intersecting a tile are
batched into it (command_buffer.rs / batch.rs:813-).
"""
import os
import numpy as np
from . import glconst as G
from .frame import (Frame, Target, Step, TextureRef, CompositeTile,
                    QF_IS_OPAQUE, QF_APPLY_DEVICE_CLIP, PART_ALL, CLIP_TASK_EMPTY)

TILE_W, TILE_H = 1024, 512
BIG = 1.0e16  # "no clip" sentinel the batcher uses (LayoutRect::max_rect analog)


def tile_grid(width, height):
    tiles = []
    for ty in range((height + TILE_H - 1) // TILE_H):
        for tx in range((width + TILE_W - 1) // TILE_W):
            tiles.append((tx, ty, tx * TILE_W, ty * TILE_H))
    return tiles


def premultiply(rgba_u8):
    """ColorF::premultiplied() on u8-derived colours (quad.rs:956)."""
    c = rgba_u8.astype(np.float32) / np.float32(255.0)
    c[:, :3] *= c[:, 3:4]
    return c


def build_rect_frame(width, height, rects, colors, opaque, encoding="quad",
                     clear_color=(1.0, 1.0, 1.0, 1.0), tile_filter=None, aa_edges=0):
    """rects: float32 [N,4] (x0,y0,x1,y1) device px; colors: float32 [N,4]
    premultiplied; opaque: bool [N].  encoding: "quad" (ps_quad_textured, what
    Rectangle prims use today: prepare.rs:218-256) or "brush" (brush_solid,
    the legacy path: batch.rs:2317-2332).  tile_filter(tx,ty)->bool selects
    the tiles this process owns (multi-GPU sharding)."""
    frame = Frame(width, height, clear_color)
    rects = np.asarray(rects, np.float32)
    n = len(rects)
    z_ids = np.arange(1, n + 1, dtype=np.int32)

    # brush path: one PrimitiveHeader + one GPU-cache colour block per prim,
    # shared by all tiles (they only differ in picture task address)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        tw, th = TILE_W, TILE_H
        x0, y0, x1, y1 = ox, oy, ox + tw, oy + th
        # (a strip-sharded rank's builder also culls against the rows of the tile it draws -- tile_filter.rows, webrender_amd/dist.py
        # strip_tile_filter --, as the frame builder culls against a tile's dirty rect: a prim outside them cannot touch a pixel the rank owns)
        rows = getattr(tile_filter, "rows", None)
        hy0, hy1 = (max(y0, rows[0]), min(y1, rows[1])) if rows else (y0, y1)
        hit = np.nonzero((rects[:, 0] < x1) & (rects[:, 2] > x0) &
                         (rects[:, 1] < hy1) & (rects[:, 3] > hy0))[0]
        tex = TextureRef(f"tile_{tx}_{ty}", tw, th, G.GL_RGBA8, G.GL_LINEAR,
                         render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=clear_color, clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(tw), float(th)), 1.0,
                                     (float(ox), float(oy)))
        op_inst, al_inst = [], []
        for i in hit:
            r, c = rects[i], colors[i]
            if encoding == "quad":
                qf = QF_APPLY_DEVICE_CLIP | (QF_IS_OPAQUE if opaque[i] else 0)
                inst = frame.quad_instance(r, (-BIG, -BIG, BIG, BIG), c, int(z_ids[i]), task,
                                           quad_flags=qf, edge_flags=0 if opaque[i] else aa_edges)
            else:
                addr = frame.gpu_cache.push([list(c)])
                ph = frame.add_prim_header(r, (-BIG, -BIG, BIG, BIG), int(z_ids[i]), addr, 0,
                                           task, (65535, 0, 0, 0))
                if aa_edges and not opaque[i]:      # BRUSH_FLAG_FORCE_AA + the edges to anti-alias
                    inst = frame.brush_instance(ph, CLIP_TASK_EMPTY, brush_flags=1024, edge_flags=aa_edges)
                else:
                    inst = frame.brush_instance(ph, CLIP_TASK_EMPTY)
            (op_inst if opaque[i] else al_inst).append(inst)
        if encoding == "quad":
            sh_op = sh_al = "ps_quad_textured"
        else:
            sh_op, sh_al = "brush_solid", "brush_solid ALPHA_PASS"
        if op_inst:
            # the opaque pass is submitted front to back: OpaqueBatchList::finalize reverses the instance arrays
            # (batch.rs:487-499) and the renderer walks the opaque batches in reverse (renderer/mod.rs:2831-2836)
            target.opaque.append(Step(sh_op, "PRIM_INSTANCES",
                                      np.array(op_inst[::-1], dtype=np.int32), None, "opaque"))
        if al_inst:
            target.alpha.append(Step(sh_al, "PRIM_INSTANCES",
                                     np.array(al_inst, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha"))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
def cfg1_solid_colors(width=1024, height=1024, encoding="quad", **kw):
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, size=(256, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, np.full((256, 1), 255, np.uint8)], axis=1)
    rects = []
    for j in range(16):
        for i in range(16):
            rects.append((i * 64, j * 64, i * 64 + 64, j * 64 + 64))
    return build_rect_frame(width, height, np.array(rects, np.float32), premultiply(rgba),
                            np.ones(256, bool), encoding, **kw)


def simple_batching(width=1024, height=1024, encoding="quad", **kw):
    """wrench/benchmarks/simple-batching.yaml: 14 x rect [0,0,512,512] green."""
    rects = np.tile(np.array([[0, 0, 512, 512]], np.float32), (14, 1))
    rgba = np.tile(np.array([[0, 255, 0, 255]], np.uint8), (14, 1))      # (wrench: "green" = (0, 1, 0), yaml_helper.rs:58)
    return build_rect_frame(width, height, rects, premultiply(rgba), np.ones(14, bool),
                            encoding, **kw)


def random_rects(n, width, height, wmin, wmax, seed, fractional=False):
    rng = np.random.default_rng(seed)
    w = rng.integers(wmin, wmax + 1, size=n).astype(np.float32)
    h = rng.integers(wmin, wmax + 1, size=n).astype(np.float32)
    x = np.floor(rng.uniform(-w / 2, width - w / 2)).astype(np.float32)
    y = np.floor(rng.uniform(-h / 2, height - h / 2)).astype(np.float32)
    if fractional:
        x += rng.uniform(0, 1, size=n).astype(np.float32)
        y += rng.uniform(0, 1, size=n).astype(np.float32)
    return rng, np.stack([x, y, x + w, y + h], axis=1)


def cfg2_overlapping_rects(width=3840, height=2160, n=1000, encoding="quad",
                           fractional=False, seed=2, **kw):
    """kw: aa_edges=0..15 turns on swgl_antiAlias for the (translucent) rects."""
    rng, rects = random_rects(n, width, height, 64, 1024, seed, fractional)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    alpha = np.round(rng.uniform(0.25, 0.75, size=n) * 255).astype(np.uint8)
    rgba = np.concatenate([rgb, alpha[:, None]], axis=1)
    return build_rect_frame(width, height, rects, premultiply(rgba), np.zeros(n, bool),
                            encoding, **kw)


def cfg5_many_rects(width=7680, height=4320, n=100_000, encoding="quad", seed=5, **kw):
    rng, rects = random_rects(n, width, height, 8, 256, seed)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    opaque = rng.uniform(size=n) < 0.5
    alpha = np.round(rng.uniform(0.1, 0.9, size=n) * 255).astype(np.uint8)
    alpha[opaque] = 255
    rgba = np.concatenate([rgb, alpha[:, None]], axis=1)
    return build_rect_frame(width, height, rects, premultiply(rgba), opaque, encoding, **kw)


# ---------------------------------------------------------------------------
# Text (BASELINE config 3).  The glyph bitmaps are the reference's own reftest fonts (wrench/reftests/text/FreeSans.ttf,
# VeraBd.ttf) rasterised by FreeType the way WebRender's Linux glyph rasteriser does (wr_glyph_rasterizer/src/platform/unix/
# font.rs restated over ctypes by tests/golden/make_glyphs.py: alpha render mode, default hinting, quarter-pixel x offsets, no
# gamma preblend on this platform) and committed as a fixture -- webrender_amd/wrench/glyphs.npz -- because neither the fonts
# nor /root/reference exist on the GPU box.  (Rounds 1-4 took PIL's rendering of DejaVu Sans.)
ATLAS_SIZE = 2048       # glyph atlas: R8 2048^2 (texture_cache.rs:533-540)
_atlas_cache = {}
_glyphs = None


class GlyphFixture:
    def __init__(self):
        import json
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wrench", "glyphs.npz"))
        self.blob, self.offsets = d["blob"], d["offsets"]
        self.index = json.loads(bytes(d["index"]).decode())
        self.charmap = json.loads(bytes(d["charmap"]).decode())
        self.runs = json.loads(bytes(d["runs"]).decode())
        # FontRenderMode::Subpixel bitmaps (FT_RENDER_MODE_LCD, BGRA as rasterize_glyph packs them) of cfg3's character set
        self.lcd_blob, self.lcd_offsets = d["lcd_blob"], d["lcd_offsets"]
        self.lcd_index = json.loads(bytes(d["lcd_index"]).decode())

    def glyph_lcd(self, font, size, gid):
        """-> (left, top, bgra[h, w, 4] or None): the subpixel-mode RasterizedGlyph (whole-pixel variant)"""
        e = self.lcd_index[f"{font}|{float(size)}|{int(gid)}|0"]
        if e[0] < 0:
            return 0, 0, None
        return e[1], e[2], self.lcd_blob[self.lcd_offsets[e[0]]:self.lcd_offsets[e[0] + 1]].reshape(e[4], e[3], 4)

    def glyph(self, font, size, gid, subpx=0):
        """-> (left, top, bitmap or None, advance): RasterizedGlyph of wr_glyph_rasterizer (bitmap's top-left corner at
        (x + left, baseline - top)); bitmap None: a glyph without pixels (space)"""
        e = self.index[f"{font}|{float(size)}|{int(gid)}|{int(subpx)}"]
        if e[0] < 0:
            return 0, 0, None, e[5]
        bm = self.blob[self.offsets[e[0]]:self.offsets[e[0] + 1]].reshape(e[4], e[3])
        return e[1], e[2], bm, e[5]

    def has_char(self, font, size, ch, subpx=0):
        gid = self.charmap.get(f"{font}|{int(ch)}")
        return gid is not None and f"{font}|{float(size)}|{int(gid)}|{int(subpx)}" in self.index

    def char(self, font, size, ch, subpx=0):
        return self.glyph(font, size, self.charmap[f"{font}|{int(ch)}"], subpx)


def glyph_fixture():
    global _glyphs
    if _glyphs is None:
        _glyphs = GlyphFixture()
    return _glyphs


def build_glyph_atlas(sizes=range(12, 25), chars=range(33, 127), font="FreeSans.ttf"):
    """Shelf-pack one bitmap per (size, char) into an R8 atlas.  Returns
    (pixels[2048,2048] u8, {(size, ch): (uv_rect texels, (left, -top), advance)})."""
    key = (tuple(sizes), tuple(chars), font)
    if key in _atlas_cache:
        return _atlas_cache[key]
    fx = glyph_fixture()
    atlas = np.zeros((ATLAS_SIZE, ATLAS_SIZE), np.uint8)
    table = {}
    x = y = 1
    shelf = 0
    for size in sizes:
        for ch in chars:
            if not fx.has_char(font, size, ch):      # (a size the fixture holds for the characters of one string only)
                continue
            left, top, bmp, adv = fx.char(font, size, ch)
            if bmp is None:
                continue
            h, w = bmp.shape
            if x + w + 1 > ATLAS_SIZE:
                x, y, shelf = 1, y + shelf + 1, 0
            assert y + h + 1 <= ATLAS_SIZE
            atlas[y:y + h, x:x + w] = bmp
            table[(size, ch)] = ((float(x), float(y), float(x + w), float(y + h)), (float(left), float(-top)), float(adv))
            x += w + 1
            shelf = max(shelf, h)
    _atlas_cache[key] = (atlas, table)
    return atlas, table


def build_glyph_atlas_lcd(sizes=range(12, 25), chars=range(33, 127), font="FreeSans.ttf"):
    """The BGRA8 atlas of subpixel-mode glyphs (FreeType LCD rendering with the reference's options: tests/golden/make_glyphs.py
    rasterize_lcd), shelf-packed like build_glyph_atlas.  Returns (pixels[2048, 2048, 4] u8 in B, G, R, A order, {(size, ch): (uv rect,
    (left, -top))}): an LCD bitmap is a pixel wider either side than the alpha one, so the layout is its own."""
    key = ("lcd", tuple(sizes), tuple(chars), font)
    if key in _atlas_cache:
        return _atlas_cache[key]
    fx = glyph_fixture()
    atlas = np.zeros((ATLAS_SIZE, ATLAS_SIZE, 4), np.uint8)
    table = {}
    x = y = 1
    shelf = 0
    for size in sizes:
        for ch in chars:
            gid = fx.charmap.get(f"{font}|{int(ch)}")
            if gid is None or f"{font}|{float(size)}|{int(gid)}|0" not in fx.lcd_index:
                continue
            left, top, bmp = fx.glyph_lcd(font, size, gid)
            if bmp is None:
                continue
            h, w = bmp.shape[:2]
            if x + w + 1 > ATLAS_SIZE:
                x, y, shelf = 1, y + shelf + 1, 0
            assert y + h + 1 <= ATLAS_SIZE
            atlas[y:y + h, x:x + w] = bmp
            table[(size, ch)] = ((float(x), float(y), float(x + w), float(y + h)), (float(left), float(-top)))
            x += w + 1
            shelf = max(shelf, h)
    _atlas_cache[key] = (atlas, table)
    return atlas, table


def char_advance(size, ch, font="FreeSans.ttf"):
    return float(glyph_fixture().char(font, size, ch)[3])


def cfg3_text(width=3840, height=2160, lines=200, glyphs_per_line=250, run_len=25, seed=3,
              glyph_zoom=1.0, device_pixel_scale=1.0, tile_filter=None, color_modes=(0,), dual_source=False, rotate=False,
              perspective=False, glyph_transform=False, gt_clip=True, **kw):
    """`lines` x `glyphs_per_line` glyphs in runs of `run_len`, black-ish text
    on white, COLOR_MODE_ALPHA from an R8 atlas, PremultipliedAlpha blend
    (batch.rs:1109-1290).  glyph_zoom != 1 draws the cached bitmaps magnified
    (local raster space: raster_scale = 1/zoom) so that sampling is truly bilinear.
    color_modes other than 0 (cycled per run: 1 = COLOR_MODE_SUBPX_DUAL_SOURCE, 2 = COLOR_MODE_BITMAP_SHADOW,
    3 = COLOR_MODE_COLOR_BITMAP) sample a BGRA8 atlas of the SUBPIXEL-mode glyphs: FreeType's LCD rendering of the same
    characters with the reference's options (build_glyph_atlas_lcd; rounds 1-5 shifted the alpha coverage by a texel per
    channel); dual_source selects the DUAL_SOURCE_BLENDING program (batch.rs:1150-1180: BlendMode::SubpixelDualSource).
    rotate / perspective: two runs in three sit under a rotation (skew) about their origin, with a projective row on top for
    `perspective` -- local-raster-space text: the glyph quads are laid out in local space by the non-GLYPH_TRANSFORM program and
    transformed as quads.
    glyph_transform: screen-raster-space text under 2-D transforms (the GLYPH_TRANSFORM keys, batch.rs:1186-1200 /
    shade.rs:1190-1210: the glyphs are rasterised under the run's transform, the program snaps them in device space and cuts
    every span to the glyph's raster rect with gl_ClipDistance): four runs in five rotated / skewed, one scaled; every third
    run under a local clip rect that cuts through its glyphs (the clamped quad is then a rotated rect reaching beyond the
    glyph's raster rect)."""
    rng = np.random.default_rng(seed)
    atlas, table = build_glyph_atlas()
    sizes = sorted({k[0] for k in table})
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    atlas_ref = TextureRef("glyph_atlas_r8", ATLAS_SIZE, ATLAS_SIZE, G.GL_R8, G.GL_LINEAR,
                           pixels=atlas, upload_format=G.GL_RED)
    frame.static_textures.append(atlas_ref)
    atlas_bgra_ref = None
    res_addr_lcd = {}
    if any(m != 0 for m in color_modes):
        sub, table_lcd = build_glyph_atlas_lcd()
        atlas_bgra_ref = TextureRef("glyph_atlas_bgra8", ATLAS_SIZE, ATLAS_SIZE, G.GL_RGBA8, G.GL_LINEAR,
                                    pixels=sub, upload_format=G.GL_BGRA)
        frame.static_textures.append(atlas_bgra_ref)
        res_addr_lcd = {k: frame.add_glyph_resource(v[0], v[1], 1.0) for k, v in table_lcd.items()}
    res_addr = {k: frame.add_glyph_resource(v[0], v[1], 1.0) for k, v in table.items()}
    raster_scale = 1.0 / glyph_zoom
    dps = device_pixel_scale
    k_dev = glyph_zoom            # device px per atlas texel

    # lay the runs out in local (layout) space
    runs = []     # (origin xy, ref offset (unsnapped, snapped), colour, [(size,ch)], points, device bbox)
    pitch = height / dps / lines
    z = 1
    for li in range(lines):
        size = int(rng.choice(sizes))
        pen = float(rng.uniform(0.0, 40.0))
        base_y = float(size + li * pitch + rng.uniform(0.0, 1.0))
        chars = rng.integers(33, 127, size=glyphs_per_line)
        chars = [int(c) if (size, int(c)) in table else 65 for c in chars]
        for r0 in range(0, glyphs_per_line, run_len):
            run_chars = chars[r0:r0 + run_len]
            origin = (pen, base_y)
            pts, x = [], 0.0
            for c in run_chars:
                pts.append((x, 0.0))
                x += table[(size, c)][2] * glyph_zoom / dps
            pen += x
            rgba = np.array([[rng.integers(0, 96), rng.integers(0, 96), rng.integers(0, 96),
                              rng.integers(160, 256)]], np.uint8)
            color = premultiply(rgba)[0]
            ref = ((2.25, 1.5), (2.0, 2.0)) if (len(runs) % 7) == 3 else ((0.0, 0.0), (0.0, 0.0))
            # conservative device-space bounds of the run (for tile assignment)
            asc = size * 1.3 * k_dev
            bx0, bx1 = origin[0] * dps - 2 * size, (origin[0] + x) * dps + 2 * size
            by0, by1 = origin[1] * dps - asc, origin[1] * dps + 0.5 * size * k_dev + 2
            tid = 0
            lclip = (-BIG, -BIG, BIG, BIG)
            if glyph_transform:
                rad = float(np.hypot(x, 2.0 * size * k_dev / dps)) + 4.0
                if len(runs) % 5 == 4:       # an axis-aligned scale (is_axis_aligned stays set)
                    sx_, sy_ = float(rng.choice([1.25, 0.8, 1.0])), float(rng.choice([1.5, 0.75]))
                    m = np.eye(4)
                    m[0, 0], m[1, 1] = sx_, sy_
                    m[:2, 3] = np.array(origin) - np.array([sx_ * origin[0], sy_ * origin[1]])
                    tid = frame.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=True)
                else:
                    tid = rotation_about(frame, rng, origin[0], origin[1], len(runs), None)
                if gt_clip and len(runs) % 3 == 1:
                    lclip = (origin[0] + 0.31 * x, origin[1] - 0.55 * size, origin[0] + 0.83 * x + 0.4, origin[1] + 0.12 * size)
                rr = rad * dps * 1.6
                bx0, bx1, by0, by1 = origin[0] * dps - rr, origin[0] * dps + rr, origin[1] * dps - rr, origin[1] * dps + rr
            elif (rotate or perspective) and len(runs) % 3 != 2:
                rad = float(np.hypot(x, 2.0 * size * k_dev / dps)) + 4.0
                tid = rotation_about(frame, rng, origin[0], origin[1], len(runs), rad if perspective else None)
                rr = rad * dps * (2.2 if perspective else 1.1)
                bx0, bx1, by0, by1 = origin[0] * dps - rr, origin[0] * dps + rr, origin[1] * dps - rr, origin[1] * dps + rr
            runs.append((origin, ref, color, size, run_chars, pts, (bx0, by0, bx1, by1), z, tid, lclip))
            z += 1

    run_addr = [frame.add_text_run(r[2], r[5]) for r in runs]
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR,
                         render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), dps,
                                     (float(ox), float(oy)))
        inst, inst_bgra = [], []
        for ri, (origin, ref, color, size, run_chars, pts, bb, zid, tid, lclip) in enumerate(runs):
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            local_rect = (origin[0] - ref[0][0], origin[1] - ref[0][1], ref[1][0], ref[1][1])
            ph = frame.add_prim_header(local_rect, lclip, zid, run_addr[ri], tid, task,
                                       (int(round(raster_scale * 65535.0)), 0, 0, 0))
            mode = color_modes[ri % len(color_modes)]
            for gi, c in enumerate(run_chars):
                (inst if mode == 0 else inst_bgra).append(
                    frame.glyph_instance(ph, gi, (res_addr if mode == 0 else res_addr_lcd)[(size, c)], color_mode=mode))
        gt = "GLYPH_TRANSFORM," if glyph_transform else ""
        if inst:
            target.alpha.append(Step(f"ps_text_run ALPHA_PASS,{gt}TEXTURE_2D", "PRIM_INSTANCES",
                                     np.array(inst, dtype=np.int32), "PremultipliedAlpha", "alpha",
                                     textures={0: atlas_ref}))
        if inst_bgra:
            key = f"ps_text_run ALPHA_PASS,DUAL_SOURCE_BLENDING,{gt}TEXTURE_2D" if dual_source else f"ps_text_run ALPHA_PASS,{gt}TEXTURE_2D"
            target.alpha.append(Step(key, "PRIM_INSTANCES", np.array(inst_bgra, dtype=np.int32),
                                     "SubpixelDualSource" if dual_source else "PremultipliedAlpha", "alpha",
                                     textures={0: atlas_bgra_ref}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    frame.n_glyphs = sum(len(s.instances) for t in targets for s in t.alpha)
    return frame


# ---------------------------------------------------------------------------
# Scaled / offset textured composites: exercises every linear-filter variant
# swgl dispatches to (nearest-fast, fast, upscale, downscale, fallback and the
# clamped lead-in / lead-out of blendTextureLinearDispatch, swgl_ext.h:378-440).
def scaled_composites(width=1024, height=768, seed=21, src=192):
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (0.2, 0.3, 0.4, 1.0))
    texs = []
    for i in range(3):
        px = rng.integers(0, 256, size=(src, src, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:src, 0:src]
        px[..., 3] = np.where(((xx // 16 + yy // 16) % 2) == 0, 255, rng.integers(64, 256, size=(src, src))).astype(np.uint8)
        px[..., :3] = (px[..., :3].astype(np.uint16) * px[..., 3:4] // 255).astype(np.uint8)
        texs.append(TextureRef(f"surface_{i}", src, src, G.GL_RGBA8, G.GL_LINEAR, pixels=px, upload_format=G.GL_BGRA))
    frame.static_textures += texs
    S = float(src)
    cases = [
        # (tex, x, y, w, h, uv_rect, color)
        (0, 10.0, 10.0, S, S, None, None),                          # 1:1 aligned
        (1, 220.3, 12.6, S, S, None, None),                         # 1:1, subpixel offset  -> FAST
        (2, 430.0, 8.0, S * 1.5, S * 1.5, None, None),              # upscale
        (0, 740.5, 20.25, S * 0.5, S * 0.5, None, None),            # exact 2x downscale
        (1, 860.0, 30.0, S * 0.37, S * 0.61, None, None),           # arbitrary downscale -> FALLBACK
        (2, 12.0, 320.0, 300.0, 180.0, (20.0, 30.0, 120.0, 90.0), None),      # sub-rect upscale x3
        (0, 330.7, 330.2, 150.0, 150.0, (0.0, 0.0, 150.0, 150.0), (0.5, 0.25, 0.75, 1.0)),   # colour modulated
        (1, 500.0, 340.0, S * 3.7, S * 1.0, None, None),            # strong x upscale, partly off-screen
        (2, -50.5, 560.0, S, S, None, None),                        # hangs over the left edge
        (0, 200.0, 560.5, S * 2.0, S * 0.5, (0.0, 0.0, S, S), None),
        (1, 640.0, 560.0, S, S, (10.5, 10.5, 100.5, 100.5), None),  # fractional uv rect
    ]
    for (ti, x, y, w, h, uv, color) in cases:
        rect = (x, y, x + w, y + h)
        clip = (max(x, 0.0), max(y, 0.0), min(x + w, float(width)), min(y + h, float(height)))
        uvr = uv or (0.0, 0.0, S, S)
        frame.composite_tiles.append(CompositeTile(texs[ti], rect, clip, opaque=False, color=color, uv_rect=uvr))
    frame.passes.append([])
    return frame


# ---------------------------------------------------------------------------
# Images: brush_image (fast variant) from a texture-cache atlas -- opaque images
# in the opaque pass (front to back, depth), translucent / tinted ones in the
# alpha pass (batch.rs:2060-2150; ImageBrushData gpu_types.rs:707-724; GPU blocks
# prim_store/image.rs: [color, background_color, stretch_size]).
def image_grid(width=1024, height=1024, n=120, seed=51, atlas=1024, tile_filter=None, modes=(0, 1, 2, 3, 4), translucent=True, only=None,
               masked=False, nearest=False, dual=False, shadows=False, screen=False):
    """`shadows`: every other alpha-pass image is a picture's drop shadow (COLOR_MODE_ALPHA / COLOR_MODE_BITMAP_SHADOW:
    swgl_blendDropShadow).  `dual`: the alpha-pass images go through "brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D" under the dual-source
    blend state (BlendMode::SubpixelDualSource / MultiplyDualSource batches, batch.rs / shade.rs:462-467), colour modes
    SUBPX_DUAL_SOURCE, MULTIPLY_DUAL_SOURCE and IMAGE in turn, with a translucent image colour."""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    pix = np.zeros((atlas, atlas, 4), np.uint8)
    srcs = []
    x = y = 0
    shelf = 0
    for i in range(24):
        w, h = int(rng.integers(24, 200)), int(rng.integers(24, 160))
        if x + w > atlas:
            x, y, shelf = 0, y + shelf, 0
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 0] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
        if i % 2 == 0:
            img[..., 3] = 255                        # opaque image
        img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)
        pix[y:y + h, x:x + w] = img
        # screen: RasterizationSpace::Screen sources (blurred / drop-shadow pictures, batch.rs:1537-1542) carry the four homogeneous st
        # corners get_image_quad_uv blends: the whole task, a surface clipped by the screen (the unclipped rect's corners fall outside
        # [0, 1], calculate_uv_rect_kind) or a sub-quad with w != 1
        if i % 3 == 0:
            quad = [[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]]
        elif i % 3 == 1:
            quad = [[-0.25, -0.125, 0.0, 1.0], [1.0, -0.125, 0.0, 1.0], [-0.25, 1.0625, 0.0, 1.0], [1.0, 1.0625, 0.0, 1.0]]
        else:
            quad = [[0.125, 0.0625, 0.0, 1.0], [1.75, 0.125, 0.0, 2.0], [0.0625, 0.9375, 0.0, 1.0], [0.96875, 1.0, 0.0, 1.0]]
        addr = frame.gpu_cache.push([[x, y, x + w, y + h], [0.0, 0.0, 0.0, 0.0]] + (quad if screen else []))
        srcs.append((w, h, addr, i % 2 == 0))
        x += w
        shelf = max(shelf, h)
    t_atlas = TextureRef("image_atlas", atlas, atlas, G.GL_RGBA8, G.GL_NEAREST if nearest else G.GL_LINEAR, pixels=pix,
                         upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)
    prims = []
    # Opaque-pass images sit on a non-overlapping grid in the top band: swgl restarts a span (chunk
    # phase, span-vs-fragment-shader split) at every depth run, so a textured prim partly hidden
    # behind an earlier opaque one is not bit-reproducible by a per-pixel evaluator;
    # everything below the band is drawn in the alpha pass, where nothing fails the depth test.
    band = 230
    gx = 4.0
    opaque_srcs = [s_ for s_ in srcs if s_[3]]
    for k in range(max(1, n // 8)):
        sw, sh, addr, _ = opaque_srcs[k % len(opaque_srcs)]
        sc = (1.0, 1.0, 1.3, 0.5, 0.77)[modes[k % len(modes)]]
        w, h = sw * sc, min(sh * sc, band - 8.0)
        if gx + w + 4 > width:
            break
        off = 0.0 if modes[k % len(modes)] != 1 else 0.37
        prims.append(((gx + off, 4.0 + off, gx + off + w, 4.0 + off + h), addr, True, 1.0))
        gx += float(np.ceil(w)) + 6.0
    for k in range(n):
        sw, sh, addr, opaque_img = srcs[int(rng.integers(0, len(srcs)))]
        opaque_img = False
        mode = modes[k % len(modes)]
        if mode == 0:      # 1:1 at an integer position
            w, h = float(sw), float(sh)
            px, py = float(rng.integers(-20, width - 20)), float(rng.integers(band, height - 20))
        elif mode == 1:    # 1:1 with a subpixel offset
            w, h = float(sw), float(sh)
            px, py = float(rng.uniform(0, width - sw)), float(rng.uniform(band, height - sh))
        elif mode == 2:    # upscaled
            sc = float(rng.uniform(1.1, 3.0))
            w, h = sw * sc, sh * sc
            px, py = float(rng.integers(0, width)) - w / 2, float(rng.integers(band + 300, height + 100)) - h / 2
            py = max(py, float(band))
        elif mode == 3:    # exact 2x downscale
            w, h = sw * 0.5, sh * 0.5
            px, py = float(rng.integers(0, width - sw)), float(rng.integers(band, height - sh))
        else:              # arbitrary scale
            w, h = sw * float(rng.uniform(0.3, 1.7)), sh * float(rng.uniform(0.3, 1.7))
            px, py = float(rng.uniform(0, width - w)), float(rng.uniform(band, height - h))
        opacity = 1.0 if (k % 3 or not translucent) else float(rng.uniform(0.3, 0.9))
        prims.append(((px, py, px + w, py + h), addr, False, opacity))
    t_mask, clip_tasks = None, [None] * len(prims)
    if masked:      # alpha-pass images under swgl_clipMask (rounded-corner clips on images)
        t_mask = TextureRef("clip_mask_atlas", 1024, 1024, G.GL_R8, G.GL_LINEAR, pixels=mask_atlas(1024), upload_format=G.GL_RED)
        frame.static_textures.append(t_mask)
        clip_tasks = prim_clip_tasks(rng, [p[0] for p in prims], 1024, True)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        op, al = [], []
        for zi, (rect, addr, opaque, opacity) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            if not (rect[0] < x1 and rect[2] > x0 and rect[1] < y1 and rect[3] > y0):
                continue
            spec = frame.gpu_cache.push([[1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [-1.0, -1.0, 0.0, 0.0]])
            ud = (4 | (1 << 16), 1 if screen else 0, int(round(opacity * 65535.0)), 0)      # COLOR_MODE_IMAGE, premultiplied, RASTER_LOCAL / RASTER_SCREEN
            if shadows and not opaque and zi % 2 == 0:
                # a picture's drop shadow: ShaderColorMode::Alpha (0) or BitmapShadow (2) with the shadow colour in the brush
                # data, premultiplied or plain-alpha opacity (brush_image.glsl:268-291)
                a = (0.35, 0.6, 0.85, 1.0)[zi % 4]
                col = [((zi * 37) % 256) / 255.0 * a, ((zi * 91) % 256) / 255.0 * a, ((zi * 53) % 256) / 255.0 * a, a]
                spec = frame.gpu_cache.push([col, [0.0, 0.0, 0.0, 0.0], [-1.0, -1.0, 0.0, 0.0]])
                ud = ((0, 2)[(zi // 2) % 2] | ((zi // 4) % 2 << 16), 1 if screen else 0, int(round(opacity * 65535.0)), 0)
            if dual and not opaque:
                a = (0.35, 0.6, 0.85, 1.0)[zi % 4]
                col = [((zi * 37) % 256) / 255.0 * a, ((zi * 91) % 256) / 255.0 * a, ((zi * 53) % 256) / 255.0 * a, a]
                spec = frame.gpu_cache.push([col, [0.0, 0.0, 0.0, 0.0], [-1.0, -1.0, 0.0, 0.0]])
                ud = ((1, 5, 4)[zi % 3] | (1 << 16), 0, int(round(opacity * 65535.0)), 0)
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, spec, 0, task, ud)
            ct = None if opaque else clip_tasks[zi]
            clip_addr = CLIP_TASK_EMPTY if ct is None else frame.add_render_task(ct[0], 1.0, ct[1])
            (op if opaque else al).append(frame.brush_instance(ph, clip_addr, resource_address=addr))
        if op:
            target.opaque.append(Step("brush_image TEXTURE_2D", "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32),
                                      None, "opaque", textures={0: t_atlas}))
        if al:
            target.alpha.append(Step("brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING,TEXTURE_2D" if dual else "brush_image ALPHA_PASS,TEXTURE_2D",
                                     "PRIM_INSTANCES", np.array(al, dtype=np.int32),
                                     "SubpixelDualSource" if dual else "PremultipliedAlpha", "alpha",
                                     textures={0: t_atlas, 9: t_mask} if masked else {0: t_atlas}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# Repeated images: the full image brush ("brush_image [ALPHA_PASS,]ANTIALIASING,REPETITION,TEXTURE_2D",
# renderer/shade.rs:995) -- background tiling through ImageBrushData.stretch_size smaller (or larger)
# than the primitive, and border-image style segments (BRUSH_FLAG_SEGMENT_RELATIVE with the REPEAT_X/Y,
# *_ROUND and *_CENTERED flags, with and without a texel rect).  `nearest` switches the atlas sampler to
# GL_NEAREST (blendTextureNearestRepeat instead of blendTextureLinearRepeat).
def image_repeat(width=1024, height=1024, n=80, seed=57, atlas=512, nearest=False, tile_filter=None, only=None, translucent=True, dual=False, aa=False):
    """`dual`: the alpha-pass prims go through "brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_2D" under the
    dual-source blend state (what a repeated image in a BlendMode::SubpixelDualSource / MultiplyDualSource batch is drawn with), colour
    modes SUBPX_DUAL_SOURCE / MULTIPLY_DUAL_SOURCE / IMAGE in turn, a brush colour other than white"""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    pix = np.zeros((atlas, atlas, 4), np.uint8)
    srcs = []
    x = y = shelf = 0
    for i in range(20):
        w, h = int(rng.integers(6, 70)), int(rng.integers(6, 60))
        if i == 3:
            w = 1                                      # one-texel-wide source: the solid-span shortcut
        if i == 7:
            w, h = 1, 1
        if x + w > atlas:
            x, y, shelf = 0, y + shelf, 0
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 1] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
        if i % 2 == 0:
            img[..., 3] = 255
        img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)
        pix[y:y + h, x:x + w] = img
        addr = frame.gpu_cache.push([[x, y, x + w, y + h], [0.0, 0.0, 0.0, 0.0]])
        srcs.append((w, h, addr, i % 2 == 0))
        x += w + 2
        shelf = max(shelf, h + 2)
    t_atlas = TextureRef("image_atlas", atlas, atlas, G.GL_RGBA8, G.GL_NEAREST if nearest else G.GL_LINEAR, pixels=pix,
                         upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)
    prims = []
    band = 200
    gx = 4.0
    opaque_srcs = [s_ for s_ in srcs if s_[3]]
    # opaque pass: non-overlapping (depth-run restarts, see image_grid)
    for k in range(8):
        sw, sh, addr, _ = opaque_srcs[k % len(opaque_srcs)]
        w, h = float(rng.integers(60, 180)), float(rng.integers(60, band - 10))
        if gx + w + 4 > width:
            break
        st = (float(sw), float(sh)) if k % 2 == 0 else (sw * float(rng.uniform(0.6, 2.2)), sh * float(rng.uniform(0.6, 2.2)))
        off = 0.0 if k % 3 else 0.41
        prims.append(dict(rect=(gx + off, 4.0 + off, gx + off + w, 4.0 + off + h), addr=addr, opaque=True, opacity=1.0, stretch=st,
                          segs=None))
        gx += float(np.ceil(w)) + 7.0
    for k in range(n):
        sw, sh, addr, _ = srcs[int(rng.integers(0, len(srcs)))]
        mode = k % 6
        w, h = float(rng.integers(40, 400)), float(rng.integers(30, 300))
        px, py = float(rng.integers(-30, width - 40)), float(rng.integers(band, height - 30))
        if mode in (1, 3):
            px += float(rng.uniform(0, 1)); py += float(rng.uniform(0, 1)); w += float(rng.uniform(0, 1))
        segs = None
        if mode == 0:        # 1:1 tiles
            st = (float(sw), float(sh))
        elif mode == 1:      # scaled tiles
            st = (sw * float(rng.uniform(0.4, 3.0)), sh * float(rng.uniform(0.4, 3.0)))
        elif mode == 2:      # tile larger than the primitive (repeat < 1)
            st = (w * float(rng.uniform(1.1, 2.0)), h * float(rng.uniform(1.1, 2.0)))
        elif mode == 3:      # no repetition through the repeat shader
            st = (-1.0, -1.0)
        else:                # border-image style segments
            st = (-1.0, -1.0)
            cw, ch = min(w / 3, 24.0), min(h / 3, 20.0)
            xs, ys = (0.0, cw, w - cw, w), (0.0, ch, h - ch, h)
            us, vs = (0.0, 0.3, 0.7, 1.0), (0.0, 0.25, 0.75, 1.0)
            segs = []
            for j in range(3):
                for i in range(3):
                    flags = 2                                   # SEGMENT_RELATIVE
                    texel = mode == 4
                    if texel:
                        flags |= 512                            # TEXEL_RECT
                        data = (us[i], vs[j], us[i + 1], vs[j + 1])
                    else:
                        data = (0.0, 0.0, float(sw) * 1.5, float(sh) * 0.75)
                    if i == 1:
                        flags |= 4 | (16 if (k // 6) % 2 else 0) | (64 if (k // 12) % 2 else 0)
                    if j == 1:
                        flags |= 8 | (32 if (k // 6) % 2 else 0) | (128 if (k // 12) % 2 else 0)
                    if i == 1 and j == 1:
                        flags |= 256                            # NINEPATCH_MIDDLE
                    segs.append(((xs[i], ys[j], xs[i + 1], ys[j + 1]), data, flags))
        opacity = 1.0 if (k % 3 or not translucent) else float(rng.uniform(0.3, 0.9))
        prims.append(dict(rect=(px, py, px + w, py + h), addr=addr, opaque=False, opacity=opacity, stretch=st, segs=segs))
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        op, al = [], []
        for zi, pr in enumerate(prims):
            rect = pr["rect"]
            if only is not None and zi not in only:
                continue
            if not (rect[0] < x1 and rect[2] > x0 and rect[1] < y1 and rect[3] > y0):
                continue
            blocks = [[1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [pr["stretch"][0], pr["stretch"][1], 0.0, 0.0]]
            ud = (4 | (1 << 16), 0, int(round(pr["opacity"] * 65535.0)), 0)      # COLOR_MODE_IMAGE, premultiplied, RASTER_LOCAL
            if dual and not pr["opaque"]:
                a = (0.35, 0.6, 0.85, 1.0)[zi % 4]
                blocks[0] = [((zi * 37) % 256) / 255.0 * a, ((zi * 91) % 256) / 255.0 * a, ((zi * 53) % 256) / 255.0 * a, a]
                ud = ((1, 5, 4)[zi % 3] | (1 << 16), 0, int(round(pr["opacity"] * 65535.0)), 0)
            for (srect, sdata, _) in (pr["segs"] or []):
                blocks += [list(srect), list(sdata)]
            spec = frame.gpu_cache.push(blocks)
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, spec, 0, task, ud)
            dst = op if pr["opaque"] else al
            if pr["segs"] is None:
                # aa: BRUSH_FLAG_FORCE_AA + all four edges on the unsegmented alpha-pass prims (swgl_antiAlias)
                if aa and not pr["opaque"]:
                    dst.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, resource_address=pr["addr"], brush_flags=1024, edge_flags=15))
                else:
                    dst.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, resource_address=pr["addr"]))
            else:
                for si, (_, _, flags) in enumerate(pr["segs"]):
                    dst.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, segment=si, brush_flags=flags, resource_address=pr["addr"]))
        if op:
            target.opaque.append(Step("brush_image ANTIALIASING,REPETITION,TEXTURE_2D", "PRIM_INSTANCES",
                                      np.array(op[::-1], dtype=np.int32), None, "opaque", textures={0: t_atlas}))
        if al:
            target.alpha.append(Step("brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_2D" if dual else
                                     "brush_image ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D", "PRIM_INSTANCES",
                                     np.array(al, dtype=np.int32), "SubpixelDualSource" if dual else "PremultipliedAlpha", "alpha", textures={0: t_atlas}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# Clip-masked rectangles: brush_solid ALPHA_PASS instances whose clip task
# address points at an R8 mask region (prim_shared.glsl:183-200 write_clip ->
# swgl_clipMask).  The masks themselves are uploaded here; in a full frame they
# are what cs_clip_rectangle / cs_clip_box_shadow render in an earlier pass.
def mask_atlas(size=1024, seed=11):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    m = 0.5 + 0.5 * np.sin(xx / 9.0) * np.cos(yy / 13.0)
    m = np.clip(m * 1.4 - 0.2, 0.0, 1.0)
    m = (m * 255.0 + 0.5).astype(np.uint8)
    noise = rng.integers(0, 256, size=(size, size), dtype=np.uint8)
    return np.where(rng.uniform(size=(size, size)) < 0.1, noise, m).astype(np.uint8)


def prim_clip_tasks(rng, rects, atlas=1024, fractional=False):
    """One R8 clip-task region of the mask atlas per prim rect (every third prim unmasked): (task rect in
    the atlas, screen origin) as masked_rects builds them -- mask smaller / larger than the prim, offset
    by a few pixels, now and then hanging over the atlas edge."""
    out = []
    for i, r in enumerate(rects):
        if i % 3 == 2:
            out.append(None)
            continue
        w, h = max(8, int(r[2] - r[0])), max(8, int(r[3] - r[1]))
        mw, mh = min(atlas - 2, max(4, w + int(rng.integers(-12, 5)))), min(atlas - 2, max(4, h + int(rng.integers(-12, 5))))
        mx, my = int(rng.integers(0, atlas - mw - 1)), int(rng.integers(0, atlas - mh - 1))
        if i % 10 == 0:
            mx = atlas - mw // 2
        sx = float(np.floor(r[0])) + float(rng.integers(-6, 7))
        sy = float(np.floor(r[1])) + float(rng.integers(-6, 7))
        if fractional and i % 4 == 1:
            sx += 0.5
        out.append(((float(mx), float(my), float(mx + mw), float(my + mh)), (sx, sy)))
    return out


def masked_rects(width=1024, height=1024, n=150, seed=12, atlas=1024, fractional=False, tile_filter=None, force_aa=False, rotate=False,
                 perspective=False):
    """`force_aa`: BRUSH_FLAG_FORCE_AA + all edge flags on every prim; `rotate`: every other prim under a rotation about its
    centre (anti-aliased by brush.glsl:150-170) -- masked solids on the general-quad walk."""
    rng, rects = random_rects(n, width, height, 16, 200, seed, fractional)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    alpha = np.round(rng.uniform(0.3, 1.0, size=n) * 255).astype(np.uint8)
    colors = premultiply(np.concatenate([rgb, alpha[:, None]], axis=1))
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    t_mask = TextureRef("clip_mask_atlas", atlas, atlas, G.GL_R8, G.GL_LINEAR, pixels=mask_atlas(atlas),
                        upload_format=G.GL_RED)
    frame.static_textures.append(t_mask)
    # one clip task per prim (every third prim is unmasked)
    clip_tasks = []
    for i in range(n):
        if i % 3 == 2:
            clip_tasks.append(None)
            continue
        w, h = int(rects[i, 2] - rects[i, 0]), int(rects[i, 3] - rects[i, 1])
        mw, mh = w + int(rng.integers(-12, 5)), h + int(rng.integers(-12, 5))
        mx, my = int(rng.integers(0, atlas - mw - 1)), int(rng.integers(0, atlas - mh - 1))
        if i % 10 == 0:       # a task hanging over the atlas edge: bounds are clamped to the mask
            mx = atlas - mw // 2
        sx = float(np.floor(rects[i, 0])) + float(rng.integers(-6, 7))
        sy = float(np.floor(rects[i, 1])) + float(rng.integers(-6, 7))
        if fractional and i % 4 == 1:
            sx += 0.5
        clip_tasks.append(((float(mx), float(my), float(mx + mw), float(my + mh)), (sx, sy)))
    tids = [0] * n
    grow = np.zeros(n)
    if rotate or perspective:      # (perspective: every prim under a projective transform, see rotated_rects)
        for i in range(0, n, 1 if perspective else 2):
            cx, cy = (rects[i, 0] + rects[i, 2]) / 2, (rects[i, 1] + rects[i, 3]) / 2
            th = float(rng.uniform(0, 2 * np.pi))
            a = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            m = np.eye(4)
            m[:2, :2] = a
            m[:2, 3] = np.array([cx, cy]) - a @ np.array([cx, cy])
            hyp = float(np.hypot(rects[i, 2] - rects[i, 0], rects[i, 3] - rects[i, 1]))
            if perspective:
                m = projective_about(a, cx, cy, hyp * 0.5, rng)
            tids[i] = frame.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=False)
            grow[i] = (0.9 if perspective else 0.25) * hyp + 2
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        hit = np.nonzero((rects[:, 0] - grow < x1) & (rects[:, 2] + grow > x0) & (rects[:, 1] - grow < y1) & (rects[:, 3] + grow > y0))[0]
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        inst = []
        for i in hit:
            addr = frame.gpu_cache.push([list(colors[i])])
            ph = frame.add_prim_header(rects[i], (-BIG, -BIG, BIG, BIG), int(i + 1), addr, tids[i], task, (65535, 0, 0, 0))
            ct = clip_tasks[i]
            clip_addr = CLIP_TASK_EMPTY if ct is None else frame.add_render_task(ct[0], 1.0, ct[1])
            inst.append(frame.brush_instance(ph, clip_addr, brush_flags=1024 if force_aa else 0, edge_flags=15 if (force_aa or tids[i]) else 0))
        if inst:
            target.alpha.append(Step("brush_solid ALPHA_PASS", "PRIM_INSTANCES", np.array(inst, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha", textures={9: t_mask}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# Rounded-rectangle clip masks (cs_clip_rectangle, fast and general path).
CLIP_RECT_DTYPE = np.dtype([
    ("area", "<f4", (4,)), ("origins", "<f4", (4,)), ("dps", "<f4"), ("tids", "<i4", (2,)),
    ("lpos", "<f4", (2,)), ("lrect", "<f4", (4,)), ("mode", "<f4"),
    ("corners", "<f4", (8, 4))])      # [rect, radii] x TL, TR, BL, BR  (ClipMaskInstanceRect, gpu_types.rs:207-228)


def clip_rect_instance(task_rect, screen_origin, dps, local_pos, size, radii, mode, sub_rect=None):
    """ClipData::rounded_rect (prim_store/mod.rs:816-878): rect at the origin of
    the clip's local space, corner rects + (outer radius, inner radius 0).
    radii = ((tlw,tlh),(trw,trh),(blw,blh),(brw,brh))."""
    w, h = size
    inst = np.zeros(1, CLIP_RECT_DTYPE)
    tw, th = task_rect[2] - task_rect[0], task_rect[3] - task_rect[1]
    inst["area"][0] = sub_rect or (0.0, 0.0, tw, th)
    inst["origins"][0] = (task_rect[0], task_rect[1], screen_origin[0], screen_origin[1])
    inst["dps"][0] = dps
    inst["tids"][0] = (0, 0)
    inst["lpos"][0] = local_pos
    inst["lrect"][0] = (0.0, 0.0, w, h)
    inst["mode"][0] = float(mode)
    (tl, tr, bl, br) = radii
    c = inst["corners"][0]
    c[0] = (0.0, 0.0, tl[0], tl[1]); c[1] = (tl[0], tl[1], 0.0, 0.0)
    c[2] = (w - tr[0], 0.0, w, tr[1]); c[3] = (tr[0], tr[1], 0.0, 0.0)
    c[4] = (0.0, h - bl[1], bl[0], h); c[5] = (bl[0], bl[1], 0.0, 0.0)
    c[6] = (w - br[0], h - br[1], w, h); c[7] = (br[0], br[1], 0.0, 0.0)
    return inst


def clip_masks(n=24, atlas=1024, seed=31, dps=1.0, window=(256, 256)):
    """`n` rounded-rect mask tasks in one R8 alpha target: even tasks have a
    uniform radius (FAST_PATH program), odd ones four different elliptical
    corners (general program); a third of them clip-out; some are followed by a
    second clip multiplied on top (draw_clip_batch_list, renderer/mod.rs:3564-3640)."""
    rng = np.random.default_rng(seed)
    frame = Frame(window[0], window[1], (1.0, 1.0, 1.0, 1.0))
    t_mask = TextureRef("clip_masks", atlas, atlas, G.GL_R8, G.GL_LINEAR, render_target=True)
    tgt = Target(t_mask, "alpha", clear_color=(1.0, 1.0, 1.0, 1.0))
    fast0, slow0, fast1, slow1 = [], [], [], []
    x = y = 4
    shelf = 0
    for k in range(n):
        w, h = float(rng.integers(30, 220)), float(rng.integers(30, 160))
        if k % 5 == 0:
            w += 0.5
        lpos = (float(rng.uniform(0, 500)), float(rng.uniform(0, 500)))
        tw = int(np.ceil(w * dps)) + int(rng.integers(-12, 13))
        th = int(np.ceil(h * dps)) + int(rng.integers(-12, 13))
        if x + tw + 4 > atlas:
            x, y, shelf = 4, y + shelf + 4, 0
        task = (float(x), float(y), float(x + tw), float(y + th))
        x += tw + 4
        shelf = max(shelf, th)
        so = (float(np.floor(lpos[0] * dps)) + float(rng.integers(-8, 9)),
              float(np.floor(lpos[1] * dps)) + float(rng.integers(-8, 9)))
        mode = 1 if k % 3 == 2 else 0
        if k % 2 == 0:
            r = float(rng.integers(0, int(min(w, h) / 2)))
            radii = ((r, r),) * 4
            (fast0 if True else fast1).append(clip_rect_instance(task, so, dps, lpos, (w, h), radii, mode))
        else:
            mx, my = w / 2, h / 2
            radii = tuple((float(rng.uniform(0, mx)), float(rng.uniform(0, my))) for _ in range(4))
            if k % 7 == 1:
                radii = ((0.0, 0.0),) + radii[1:]
            slow0.append(clip_rect_instance(task, so, dps, lpos, (w, h), radii, mode))
        if k % 4 == 1:       # a second, smaller clip on the same task, multiplied in
            r2 = float(rng.integers(2, 20))
            inst2 = clip_rect_instance(task, so, dps, (lpos[0] + 7.25, lpos[1] + 5.5), (w * 0.8, h * 0.7),
                                       ((r2, r2),) * 4, 0)
            fast1.append(inst2)
    tgt.steps.append(Step("cs_clip_rectangle FAST_PATH", "CLIP_RECT", np.concatenate(fast0), None, "none"))
    tgt.steps.append(Step("cs_clip_rectangle", "CLIP_RECT", np.concatenate(slow0), None, "none"))
    if fast1:
        tgt.steps.append(Step("cs_clip_rectangle FAST_PATH", "CLIP_RECT", np.concatenate(fast1), "Multiply", "none"))
    frame.passes.append([tgt])
    frame.readback = [t_mask]
    return frame


# ---------------------------------------------------------------------------
# Box-shadow clip masks: nine-patch stretch of a cached blurred corner texture
# (cs_clip_box_shadow; ClipMaskInstanceBoxShadow, gpu_types.rs:230-247).
BOX_SHADOW_DTYPE = np.dtype([
    ("area", "<f4", (4,)), ("origins", "<f4", (4,)), ("dps", "<f4"), ("tids", "<i4", (2,)),
    ("res", "<u2", (2,)), ("src_size", "<f4", (2,)), ("mode", "<i4"), ("stretch", "<i4", (2,)),
    ("dest", "<f4", (4,))])


def blurred_shadow_tile(size, radius, sigma, rng):
    """What the blur chain leaves in the texture cache for a box shadow: a
    rounded rect blurred with std deviation `sigma` (numpy; the content only has
    to be plausible -- both backends sample the same bytes)."""
    from scipy.ndimage import gaussian_filter
    n = size
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float32)
    pad = 3.0 * sigma
    x0, y0, x1, y1 = pad, pad, n - pad, n - pad
    dx = np.maximum(np.maximum(x0 + radius - xx, xx - (x1 - radius)), 0.0)
    dy = np.maximum(np.maximum(y0 + radius - yy, yy - (y1 - radius)), 0.0)
    inside = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1) & (np.hypot(dx, dy) <= radius)
    img = gaussian_filter(inside.astype(np.float32), sigma)
    return np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)


def box_shadow_masks(n=16, atlas=1024, seed=41, dps=1.0, window=(256, 256)):
    rng = np.random.default_rng(seed)
    frame = Frame(window[0], window[1], (1.0, 1.0, 1.0, 1.0))
    cache = np.zeros((512, 512), np.uint8)
    tiles = []
    for i, (sz, rad, sg) in enumerate([(96, 12.0, 5.0), (64, 20.0, 3.0), (128, 30.0, 8.0), (48, 4.0, 2.0)]):
        x, y = 8 + (i % 2) * 200, 8 + (i // 2) * 200
        cache[y:y + sz, x:x + sz] = blurred_shadow_tile(sz, rad, sg, rng)
        addr = frame.gpu_cache.push([[x, y, x + sz, y + sz], [0.0, 0.0, 0.0, 0.0]])
        tiles.append((sz, addr))
    t_cache = TextureRef("shadow_cache", 512, 512, G.GL_R8, G.GL_LINEAR, pixels=cache, upload_format=G.GL_RED)
    frame.static_textures.append(t_cache)
    t_mask = TextureRef("box_shadow_masks", atlas, atlas, G.GL_R8, G.GL_LINEAR, render_target=True)
    tgt = Target(t_mask, "alpha", clear_color=(1.0, 1.0, 1.0, 1.0))
    inst = np.zeros(n, BOX_SHADOW_DTYPE)
    x = y = 4
    shelf = 0
    for k in range(n):
        sz, addr = tiles[k % len(tiles)]
        src_local = sz / dps                       # the cached shadow is rasterised at device scale
        stretch = (int(k % 3 == 1), int(k % 4 == 2))       # 0 = MODE_STRETCH, 1 = MODE_SIMPLE
        w = src_local if stretch[0] else src_local + float(rng.integers(10, 260))
        h = src_local if stretch[1] else src_local + float(rng.integers(10, 200))
        if k % 5 == 3:
            w += 0.5
        lpos = (float(rng.uniform(0, 400)), float(rng.uniform(0, 400)))
        if k % 2 == 0:
            lpos = (float(np.floor(lpos[0])), float(np.floor(lpos[1])))
        tw = int(np.ceil(w * dps)) + int(rng.integers(-10, 11))
        th = int(np.ceil(h * dps)) + int(rng.integers(-10, 11))
        if x + tw + 4 > atlas:
            x, y, shelf = 4, y + shelf + 4, 0
        task = (float(x), float(y), float(x + tw), float(y + th))
        x += tw + 4
        shelf = max(shelf, th)
        so = (float(np.floor(lpos[0] * dps)) + float(rng.integers(-6, 7)),
              float(np.floor(lpos[1] * dps)) + float(rng.integers(-6, 7)))
        inst["area"][k] = (0.0, 0.0, tw, th)
        inst["origins"][k] = (task[0], task[1], so[0], so[1])
        inst["dps"][k] = dps
        inst["tids"][k] = (0, 0)
        inst["res"][k] = (addr % 1024, addr // 1024)
        inst["src_size"][k] = (src_local, src_local)
        inst["mode"][k] = 1 if k % 3 == 2 else 0
        inst["stretch"][k] = stretch
        inst["dest"][k] = (lpos[0], lpos[1], lpos[0] + w, lpos[1] + h)
    tgt.steps.append(Step("cs_clip_box_shadow TEXTURE_2D", "CLIP_BOX_SHADOW", inst, None, "none",
                          textures={0: t_cache}))
    frame.passes.append([tgt])
    frame.readback = [t_mask]
    return frame


# ---------------------------------------------------------------------------
# Separable Gaussian blur chain (the off-screen half of BASELINE config 4).
BLUR_DTYPE = np.dtype([("a", "<i4", (3,)), ("p", "<f4", (3,))])   # BlurInstance, gpu_types.rs:109-118


def blur_instance(task_address, src_task_address, direction, std_dev, region):
    inst = np.zeros(1, BLUR_DTYPE)
    inst["a"][0] = (task_address, src_task_address, direction)
    inst["p"][0] = (std_dev, region[0], region[1])
    return inst


SCALE_DTYPE = np.dtype([("t", "<f4", (4,)), ("s", "<f4", (4,)), ("k", "<f4")])   # ScalingInstance, gpu_types.rs:121-143


def blur_chain(fmt="r8", content=(83, 83), sigma=2.5, atlas=256, n_tasks=1, seed=4, origin=(17, 9),
               window=(256, 256), pattern="shapes", scale_steps=0):
    """`n_tasks` independent [downscale x scale_steps -> vertical -> horizontal]
    task chains (render_task.rs:1120-1215 new_blur): source content in
    `blur_src` (standing in for the mask / picture the blur reads), `cs_scale`
    halvings into `scale_k` (used when the std deviation exceeds
    MAX_BLUR_STD_DEV = 4), vertical cs_blur into `blur_v`, horizontal into
    `blur_h`.  Rects packed on a grid in `atlas`-sized targets like the
    render-task allocator would.  Returns a Frame whose `readback` lists the
    target textures to compare."""
    rng = np.random.default_rng(seed)
    cw, ch = content
    color = fmt == "rgba8"
    glfmt = G.GL_RGBA8 if color else G.GL_R8
    if color:
        src = rng.integers(0, 256, size=(atlas, atlas, 4), dtype=np.uint8)
    else:
        src = rng.integers(0, 256, size=(atlas, atlas), dtype=np.uint8)
    if pattern == "shapes":       # rounded blobs: what a mask really looks like, plus some noise
        yy, xx = np.mgrid[0:atlas, 0:atlas]
        m = (((xx // 23 + yy // 31) % 2) * 255).astype(np.uint8)
        if color:
            src = np.where(rng.uniform(size=(atlas, atlas, 1)) < 0.7, m[..., None], src).astype(np.uint8)
            src[..., :3] = (src[..., :3].astype(np.uint16) * src[..., 3:4] // 255).astype(np.uint8)
        else:
            src = np.where(rng.uniform(size=(atlas, atlas)) < 0.7, m, src).astype(np.uint8)
    frame = Frame(window[0], window[1], (1.0, 1.0, 1.0, 1.0))
    t_src = TextureRef("blur_src", atlas, atlas, glfmt, G.GL_LINEAR, pixels=src,
                       upload_format=G.GL_BGRA if color else G.GL_RED)
    frame.static_textures.append(t_src)
    kind = "color" if color else "alpha"
    zero = (0.0, 0.0, 0.0, 0.0)
    per_row = max(1, (atlas - origin[0]) // (cw + 3))
    rects = []
    for k in range(n_tasks):
        gx, gy = k % per_row, k // per_row
        x0, y0 = origin[0] + gx * (cw + 3), origin[1] + gy * (ch + 3)
        assert y0 + ch <= atlas
        rects.append((float(x0), float(y0), float(x0 + cw), float(y0 + ch)))
    # downscale passes: each task's rect keeps its origin, its size halves (rounded up)
    cur_tex, cur_rects, size = t_src, rects, (cw, ch)
    frame.readback = []
    for step in range(scale_steps):
        size = ((size[0] + 1) // 2, (size[1] + 1) // 2)
        t_s = TextureRef(f"scale_{step}", atlas, atlas, glfmt, G.GL_LINEAR, render_target=True)
        tgt = Target(t_s, kind, clear_color=zero)
        nxt, inst = [], np.zeros(n_tasks, SCALE_DTYPE)
        for k, r in enumerate(cur_rects):
            nr = (r[0], r[1], r[0] + size[0], r[1] + size[1])
            nxt.append(nr)
            inst["t"][k], inst["s"][k], inst["k"][k] = nr, r, 1.0
        tgt.steps.append(Step("cs_scale TEXTURE_2D", "SCALE", inst, None, "none", textures={0: cur_tex}))
        frame.passes.append([tgt])
        frame.readback.append(t_s)
        cur_tex, cur_rects = t_s, nxt
    t_v = TextureRef("blur_v", atlas, atlas, glfmt, G.GL_LINEAR, render_target=True)
    t_h = TextureRef("blur_h", atlas, atlas, glfmt, G.GL_LINEAR, render_target=True)
    key = "cs_blur COLOR_TARGET" if color else "cs_blur ALPHA_TARGET"
    tgt_v = Target(t_v, kind, clear_color=zero)
    tgt_h = Target(t_h, kind, clear_color=zero)
    vi, hi = [], []
    for k, rect in enumerate(cur_rects):
        a_src = frame.add_render_task(rect)
        a_v = frame.add_render_task(rect)
        a_h = frame.add_render_task(rect)
        sg = sigma if np.isscalar(sigma) else sigma[k % len(sigma)]
        vi.append(blur_instance(a_v, a_src, 1, sg, size))
        hi.append(blur_instance(a_h, a_v, 0, sg, size))
    tgt_v.steps.append(Step(key, "BLUR", np.concatenate(vi), None, "none", textures={0: cur_tex}))
    tgt_h.steps.append(Step(key, "BLUR", np.concatenate(hi), None, "none", textures={0: t_v}))
    frame.passes.append([tgt_v])
    frame.passes.append([tgt_h])
    frame.readback += [t_v, t_h]
    return frame


# ---------------------------------------------------------------------------
# BASELINE config 4: wrench/benchmarks/box-shadow-large.yaml -- the complete
# box-shadow chain of SURVEY.md §3.3:
#   cs_clip_rectangle (minimal rounded rect, R8)  ->  cs_scale halvings while the
#   std deviation exceeds 4  ->  cs_blur V  ->  cs_blur H (the cached corner
#   texture)  ->  per-prim mask task: cs_clip_box_shadow + cs_clip_rectangle
#   clip-out of the box (multiplied)  ->  picture tiles: brush_solid ALPHA_PASS,
#   4 segments, sampling the mask through swgl_clipMask  ->  composite.
# Geometry from the yaml: bounds [100,100,800,800], blur-radius 20, radii
# TL 20 / TR 10 / BL 25 / BR 100, blue, outset.  `dps` scales the whole page
# (device_pixel_scale); sizes follow compute_box_shadow_parameters (clip.rs:1765-1856).
def cfg4_box_shadow(width=3840, height=2160, dps=1.0, n_shadows=1, tile_filter=None, boxes=None, blur_radius=20.0,
                    radii=((20.0, 20.0), (10.0, 10.0), (25.0, 25.0), (100.0, 100.0)), shadow_color=(0, 0, 255, 255), offset=(0.0, 0.0),
                    clip_rects=None, **kw):
    """`boxes`: the box-shadow items' box-bounds (x0, y0, x1, y1), default: n_shadows copies of [100, 100, 900, 900] stacked
    down the page; `offset`: the shadow's offset from its box (the clip-out stays on the box); `radii`: TL, TR, BL, BR.  Every
    shadow of one (blur radius, radii) shares ONE cached blurred minimal shadow (box_shadow.rs BoxShadowCacheKey)."""
    BLUR_SAMPLE_SCALE, MAX_BLUR_STD_DEV = 3.0, 4.0      # box_shadow.rs:48, render_task.rs:37
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    if boxes is None:
        boxes = [(100.0, 100.0 + i * 1000.0, 900.0, 900.0 + i * 1000.0) for i in range(n_shadows)]
    n_shadows = len(boxes)
    blur_region = float(np.ceil(BLUR_SAMPLE_SCALE * blur_radius))          # 60
    # (per axis: the largest corner width / height, clip.rs:1786-1806 -- elliptical radii give a non-square minimal rect)
    corner_w, corner_h = max(max(r[0] for r in radii), blur_region), max(max(r[1] for r in radii), blur_region)
    min_w, min_h = 2.0 * corner_w + blur_region, 2.0 * corner_h + blur_region          # 260
    alloc_w, alloc_h = 2.0 * blur_region + np.ceil(min_w), 2.0 * blur_region + np.ceil(min_h)      # 380 (shadow_rect_alloc_size)
    cache_w, cache_h = int(np.ceil(alloc_w * dps)), int(np.ceil(alloc_h * dps))
    cache_px = max(cache_w, cache_h)
    sigma = blur_radius * 0.5 * dps
    steps = 0
    while sigma > MAX_BLUR_STD_DEV:
        sigma *= 0.5
        steps += 1
    atlas = 1 << int(np.ceil(np.log2(max(cache_px + 8, 256))))
    zero = (0.0, 0.0, 0.0, 0.0)
    # -- pass 0: the minimal rounded rect, at cache resolution
    t_m0 = TextureRef("bs_corner_mask", atlas, atlas, G.GL_R8, G.GL_LINEAR, render_target=True)
    tgt = Target(t_m0, "alpha", clear_color=zero)
    task0 = (4.0, 4.0, 4.0 + cache_w, 4.0 + cache_h)
    tgt.steps.append(Step("cs_clip_rectangle", "CLIP_RECT",
                          clip_rect_instance(task0, (0.0, 0.0), dps, (blur_region, blur_region), (min_w, min_h),
                                             radii, 0), None, "none"))
    frame.passes.append([tgt])
    # -- downscale passes
    cur_tex, cur_rect, size = t_m0, task0, (cache_w, cache_h)
    for st in range(steps):
        size = ((size[0] + 1) // 2, (size[1] + 1) // 2)
        t_s = TextureRef(f"bs_scale_{st}", atlas, atlas, G.GL_R8, G.GL_LINEAR, render_target=True)
        tg = Target(t_s, "alpha", clear_color=zero)
        nr = (4.0, 4.0, 4.0 + size[0], 4.0 + size[1])
        inst = np.zeros(1, SCALE_DTYPE)
        inst["t"][0], inst["s"][0], inst["k"][0] = nr, cur_rect, 1.0
        tg.steps.append(Step("cs_scale TEXTURE_2D", "SCALE", inst, None, "none", textures={0: cur_tex}))
        frame.passes.append([tg])
        cur_tex, cur_rect = t_s, nr
    # -- blur passes (the horizontal one lands in the texture cache)
    t_v = TextureRef("bs_blur_v", atlas, atlas, G.GL_R8, G.GL_LINEAR, render_target=True)
    t_cache = TextureRef("bs_texture_cache", atlas, atlas, G.GL_R8, G.GL_LINEAR, render_target=True)
    a_src, a_v, a_h = (frame.add_render_task(cur_rect) for _ in range(3))
    tg_v, tg_h = Target(t_v, "alpha", clear_color=zero), Target(t_cache, "alpha", clear_color=zero)
    tg_v.steps.append(Step("cs_blur ALPHA_TARGET", "BLUR", blur_instance(a_v, a_src, 1, sigma, size), None,
                           "none", textures={0: cur_tex}))
    tg_h.steps.append(Step("cs_blur ALPHA_TARGET", "BLUR", blur_instance(a_h, a_v, 0, sigma, size), None,
                           "none", textures={0: t_v}))
    frame.passes.append([tg_v])
    frame.passes.append([tg_h])
    res = frame.gpu_cache.push([list(cur_rect), [0.0, 0.0, 0.0, 0.0]])     # ImageSource of the cached shadow
    # -- per-prim mask tasks + masked brushes (n_shadows copies stacked down the page)
    masks_w = int(np.ceil((max(max(b[2] - b[0], b[3] - b[1]) for b in boxes) + 2.0 * blur_region) * dps)) + 8
    m_atlas_w = 1 << int(np.ceil(np.log2(masks_w)))
    m_atlas_h = 1 << int(np.ceil(np.log2(masks_w * n_shadows)))
    t_masks = TextureRef("bs_prim_masks", m_atlas_w, m_atlas_h, G.GL_R8, G.GL_LINEAR, render_target=True)
    tg_m = Target(t_masks, "alpha", clear_color=(1.0, 1.0, 1.0, 1.0))
    bs_inst = np.zeros(n_shadows, BOX_SHADOW_DTYPE)
    co_inst = []
    prims = []
    for i in range(n_shadows):
        box = tuple(float(v) for v in boxes[i])
        dest = (box[0] + offset[0] - blur_region, box[1] + offset[1] - blur_region, box[2] + offset[0] + blur_region, box[3] + offset[1] + blur_region)
        dev = tuple(v * dps for v in dest)
        so = (float(np.floor(dev[0])), float(np.floor(dev[1])))
        tw, th = int(np.ceil(dev[2]) - so[0]), int(np.ceil(dev[3]) - so[1])
        task = (4.0, 4.0 + i * masks_w, 4.0 + tw, 4.0 + i * masks_w + th)
        bs_inst["area"][i] = (0.0, 0.0, tw, th)
        bs_inst["origins"][i] = (task[0], task[1], so[0], so[1])
        bs_inst["dps"][i] = dps
        bs_inst["res"][i] = (res % 1024, res // 1024)
        bs_inst["src_size"][i] = (alloc_w, alloc_h)
        bs_inst["mode"][i] = 0
        bs_inst["stretch"][i] = (0, 0)
        bs_inst["dest"][i] = dest
        co_inst.append(clip_rect_instance(task, so, dps, (box[0], box[1]), (box[2] - box[0], box[3] - box[1]), radii, 1))
        clip_task = frame.add_render_task(task, dps, so)
        prims.append((dest, box, clip_task))
    tg_m.steps.append(Step("cs_clip_box_shadow TEXTURE_2D", "CLIP_BOX_SHADOW", bs_inst, None, "none",
                           textures={0: t_cache}))
    tg_m.steps.append(Step("cs_clip_rectangle", "CLIP_RECT", np.concatenate(co_inst), "Multiply", "none"))
    frame.passes.append([tg_m])
    # -- picture tiles
    color = premultiply(np.array([list(shadow_color)], np.uint8))[0]
    addr_color = frame.gpu_cache.push([list(color)])
    in_l, in_t = max(radii[0][0], radii[2][0]), max(radii[0][1], radii[1][1])     # largest left / top radii
    in_r, in_b = max(radii[1][0], radii[3][0]), max(radii[2][1], radii[3][1])     # ... right / bottom radii
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), dps, (float(ox), float(oy)))
        inst = []
        for pi, (dest, box, clip_task) in enumerate(prims):
            dv = [v * dps for v in dest]
            if not (dv[0] < x1 and dv[2] > x0 and dv[1] < y1 and dv[3] > y0):
                continue
            # brush segments (prim_store BrushSegment): the ring around the box interior
            ix0, iy0 = box[0] + in_l, box[1] + in_t          # inset by the largest left / top radii
            ix1, iy1 = box[2] - in_r, box[3] - in_b          # ... right / bottom radii
            segs = [(dest[0], dest[1], dest[2], iy0), (dest[0], iy1, dest[2], dest[3]),
                    (dest[0], iy0, ix0, iy1), (ix1, iy0, dest[2], iy1)]
            blocks = [list(color)]
            for sg in segs:
                blocks += [[sg[0] - dest[0], sg[1] - dest[1], sg[2] - dest[0], sg[3] - dest[1]], [0.0, 0.0, 0.0, 0.0]]
            addr = frame.gpu_cache.push(blocks)
            lclip = (-BIG, -BIG, BIG, BIG) if clip_rects is None else tuple(clip_rects[pi])      # the item's clip rect
            ph = frame.add_prim_header(dest, lclip, pi + 1, addr, 0, task, (65535, 0, 0, 0))
            for si in range(4):
                inst.append(frame.brush_instance(ph, clip_task, segment=si))
        if inst:
            target.alpha.append(Step("brush_solid ALPHA_PASS", "PRIM_INSTANCES", np.array(inst, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha", textures={9: t_masks}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    frame.readback = [t_cache, t_masks]
    return frame


# ps_copy: texture-cache copies (the cache grows or is defragmented: renderer/mod.rs:1808-1846) and batched uploads (small
# uploads are packed into a staging texture and blitted into place: renderer/upload.rs:540-620).  One CopyInstance per rect
# (gpu_types.rs:169-175): source rect in source texels, destination rect in destination texels, both of one size.
COPY_DTYPE = np.dtype([("src", "<f4", (4,)), ("dst", "<f4", (4,)), ("size", "<f4", (2,))])


def texture_cache_copies(n=60, seed=191, src_size=512, dst_size=768, chained=True):
    """`n` copies out of an RGBA8 and an R8 source atlas into cache textures of another size (later copies overwrite earlier
    ones where they overlap); `chained`: a second pass copies out of the first pass' destination (a cache texture that is
    defragmented right after it was filled)."""
    rng = np.random.default_rng(seed)
    frame = Frame(64, 64, (1.0, 1.0, 1.0, 1.0))
    rgba = rng.integers(0, 256, size=(src_size, src_size, 4), dtype=np.uint8)
    r8 = rng.integers(0, 256, size=(src_size, src_size), dtype=np.uint8)
    t_src = TextureRef("copy_src_rgba", src_size, src_size, G.GL_RGBA8, G.GL_LINEAR, pixels=rgba)
    t_src8 = TextureRef("copy_src_r8", src_size, src_size, G.GL_R8, G.GL_LINEAR, pixels=r8)
    frame.static_textures += [t_src, t_src8]

    def copies(src_w, dst_w, k, seed_):
        r = np.random.default_rng(seed_)
        inst = np.zeros(k, COPY_DTYPE)
        for i in range(k):
            w, h = int(r.integers(1, 200)), int(r.integers(1, 120))
            if i % 7 == 0:
                w, h = int(r.integers(1, 4)), int(r.integers(1, 4))            # glyph-sized
            sx, sy = int(r.integers(0, src_w - w + 1)), int(r.integers(0, src_w - h + 1))
            dx, dy = int(r.integers(0, dst_w - w + 1)), int(r.integers(0, dst_w - h + 1))
            inst["src"][i] = (sx, sy, sx + w, sy + h)
            inst["dst"][i] = (dx, dy, dx + w, dy + h)
            inst["size"][i] = (dst_w, dst_w)
        return inst
    passes, out = [], []
    for name, src, fmt in (("rgba", t_src, G.GL_RGBA8), ("r8", t_src8, G.GL_R8)):
        t_dst = TextureRef(f"copy_dst_{name}", dst_size, dst_size, fmt, G.GL_LINEAR, render_target=True)
        tg = Target(t_dst, "texture_cache", clear_color=(0.0, 0.0, 0.0, 0.0))
        tg.steps.append(Step("ps_copy", "COPY", copies(src_size, dst_size, n, seed + len(out)), None, "none", textures={0: src}))
        passes.append(tg)
        out.append(t_dst)
        if chained:
            t_dst2 = TextureRef(f"copy_dst2_{name}", src_size, src_size, fmt, G.GL_LINEAR, render_target=True)
            tg2 = Target(t_dst2, "texture_cache", clear_color=(0.0, 0.0, 0.0, 0.0))
            tg2.steps.append(Step("ps_copy", "COPY", copies(dst_size, src_size, n // 2, seed + 7 + len(out)), None, "none", textures={0: t_dst}))
            passes.append(tg2)
            out.append(t_dst2)
    frame.passes.append([t for t in passes if "dst2" not in t.texture.name])
    if chained:
        frame.passes.append([t for t in passes if "dst2" in t.texture.name])
    frame.readback = out
    return frame


SCENES = {
    "cfg1": cfg1_solid_colors,
    "simple_batching": simple_batching,
    "cfg2": cfg2_overlapping_rects,
    "cfg3": cfg3_text,
    "cfg4": cfg4_box_shadow,
    "cfg5": cfg5_many_rects,
}


# ---------------------------------------------------------------------------
# Linear gradients: brush_linear_gradient instances (batch.rs:2679-2780).  The
# colour table is the 130-entry (start, step) LUT GradientGpuBlockBuilder::build
# writes into GpuBufferF (prim_store/gradient/mod.rs:108-320); the brush block in
# the GPU cache is [start_point, end_point], [extend_mode, stretch_size, 0]
# (prim_store/gradient/linear.rs:466-482).

GRADIENT_TABLE_SIZE = 128


def build_gradient_lut(stops, reverse=False):
    """stops: [(offset, (r, g, b, a) straight-alpha floats)], first offset 0, last 1.
    Returns a (260, 4) float32 array: entry i = [start_color, end_step]."""
    f32 = np.float32

    def premul(c):
        c = np.asarray(c, f32)
        return np.array([c[0] * c[3], c[1] * c[3], c[2] * c[3], c[3]], f32)

    entries = np.ones((GRADIENT_TABLE_SIZE + 2, 2, 4), f32)
    entries[:, 1, :] = 0.0
    first, begin, end, last = 0, 1, GRADIENT_TABLE_SIZE + 1, GRADIENT_TABLE_SIZE + 1

    def fill(start_idx, end_idx, c0, c1, prev_step):
        inv_steps = f32(1.0) / f32(end_idx - start_idx)
        step = ((c1 - c0) * inv_steps).astype(f32)
        if np.array_equal(step, prev_step):
            bits = step[3:4].view(np.uint32)
            bits[0] = 1 if step[3] == 0.0 else bits[0] + 1
        cur = c0.copy()
        for i in range(start_idx, end_idx):
            entries[i, 0] = cur
            cur = (cur + step).astype(f32)
            entries[i, 1] = step
        return step

    def get_index(offset):
        v = f32(min(max(f32(offset), f32(0.0)), f32(1.0))) * f32(GRADIENT_TABLE_SIZE) + f32(begin)
        return int(np.floor(v + f32(0.5)))        # f32::round (half away from zero, v > 0)

    it = iter(stops)
    cur_color = premul(next(it)[1])
    prev_step = cur_color.copy()
    if reverse:
        prev_step = fill(last, last + 1, cur_color, cur_color, prev_step)
        cur_idx = end
        for off, col in it:
            nxt, idx = premul(col), get_index(1.0 - off)
            if idx < cur_idx:
                prev_step = fill(idx, cur_idx, nxt, cur_color, prev_step)
                cur_idx = idx
            cur_color = nxt
        fill(first, first + 1, cur_color, cur_color, prev_step)
    else:
        prev_step = fill(first, first + 1, cur_color, cur_color, prev_step)
        cur_idx = begin
        for off, col in it:
            nxt, idx = premul(col), get_index(off)
            if idx > cur_idx:
                prev_step = fill(cur_idx, idx, cur_color, nxt, prev_step)
                cur_idx = idx
            cur_color = nxt
        fill(last, last + 1, cur_color, cur_color, prev_step)
    return entries.reshape(-1, 4)


def rotation_about(frame, rng, cx, cy, i, persp_radius=None, strength=(0.15, 0.6)):
    """A non-axis-aligned transform (rotation, every fourth one skewed as well) about (cx, cy) -> transform id.
    `persp_radius`: with a projective row on top (projective_about: w varies over the prim, > 0 inside that radius)."""
    th = float(rng.uniform(0, 2 * np.pi)) if i % 6 else float(rng.choice([np.pi / 4, np.pi / 2, 0.01]))
    sk = float(rng.uniform(-0.4, 0.4)) if i % 4 == 1 else 0.0
    c, sn = np.cos(th), np.sin(th)
    a = np.array([[c, -sn + sk * c], [sn, c + sk * sn]], np.float64)
    m = np.eye(4)
    m[:2, :2] = a
    m[:2, 3] = np.array([cx, cy]) - a @ np.array([cx, cy])
    if persp_radius is not None:
        m = projective_about(a, cx, cy, persp_radius * (1.0 + abs(sk)), rng, strength=strength)
    return frame.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=False)


def projective_about(a, cx, cy, rr, rng, strength=(0.15, 0.6)):
    """T(c) . P . A . T(-c): the 2x2 map `a` about (cx, cy) followed by a projective row w = 1 + k . (rotated offset from the
    centre) with |k . offset| <= 0.6 inside radius rr, so that w stays positive over the prim.  A `strength` above 1 puts part
    of the prim at or behind the camera plane (w <= 0): swgl clips it against the view volume first (clip_side,
    rasterize.h:1287-1430, 1490-1544)."""
    k = rng.uniform(-1.0, 1.0, size=2)
    k = k / max(float(np.hypot(*k)), 1e-3) * float(rng.uniform(*strength)) / rr
    tneg, tpos, rot, pm = np.eye(4), np.eye(4), np.eye(4), np.eye(4)
    tneg[:2, 3] = (-cx, -cy); tpos[:2, 3] = (cx, cy)
    rot[:2, :2] = a
    pm[3, 0], pm[3, 1] = k
    return tpos @ pm @ rot @ tneg


def rotated_bounds(rect):
    x0, y0, x1, y1 = rect
    cx, cy, rad = (x0 + x1) / 2, (y0 + y1) / 2, float(np.hypot(x1 - x0, y1 - y0)) * 0.75 + 4
    return (cx - rad, cy - rad, cx + rad, cy + rad)


def gradient_grid(width=1024, height=1024, n=60, seed=61, tile_filter=None, only=None, fractional=True, rotate=False, perspective=False):
    """Opaque-pass gradients on a disjoint grid in the top band (see image_grid on why),
    translucent / overlapping ones below it in the alpha pass."""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))

    def rand_stops(opaque, k):
        offs = [0.0] + sorted(float(o) for o in rng.uniform(0.05, 0.95, size=k - 2)) + [1.0]
        if k > 3 and rng.integers(0, 2):
            offs[2] = offs[1]                       # hard stop
        cols = []
        for _ in range(k):
            c = [float(v) / 255.0 for v in rng.integers(0, 256, size=3)]
            cols.append(tuple(c) + ((1.0,) if opaque else (float(rng.integers(40, 256)) / 255.0,)))
        return list(zip(offs, cols))

    luts = []
    for i in range(10):
        opaque = i < 4
        stops = rand_stops(opaque, int(rng.integers(2, 6)))
        addr = frame.gpu_buffer_f.push(build_gradient_lut(stops, reverse=bool(i & 1)))
        luts.append((addr, opaque))
    prims = []
    band = 200

    def make(rect, opaque_pass, k):
        x0, y0, x1, y1 = rect
        w, h = x1 - x0, y1 - y0
        lut, _ = luts[(k % 4) if opaque_pass else int(rng.integers(0, len(luts)))]
        mode = k % 8
        if mode == 0:   sp, ep = (0.0, 0.0), (w, 0.0)                    # horizontal, whole rect
        elif mode == 1: sp, ep = (0.0, 0.0), (0.0, h)                    # vertical
        elif mode == 2: sp, ep = (0.0, 0.0), (w, h)                      # diagonal
        elif mode == 3: sp, ep = (w * 0.75, h * 0.5), (w * 0.25, h * 0.4)  # reversed direction, short line
        elif mode == 4: sp, ep = (w * 0.3, 0.0), (w * 0.6, 0.0)          # clamps on both sides
        elif mode == 5: sp, ep = (float(rng.uniform(0, w)), float(rng.uniform(0, h))), (float(rng.uniform(0, w)), float(rng.uniform(0, h)))
        elif mode == 6: sp, ep = (0.0, 0.0), (w * 0.2, h * 0.1)
        else:           sp, ep = (w * 0.5, h * 0.5), (w * 0.5, h * 0.5)                  # degenerate line: delta is not finite, every pixel runs main()
        extend = 1 if (k % 3 == 1) else 0
        stretch = (w, h) if k % 5 else (max(w / 2.5, 8.0), max(h / 1.5, 8.0))      # tiled pattern
        spec = frame.gpu_cache.push([[sp[0], sp[1], ep[0], ep[1]], [float(extend), stretch[0], stretch[1], 0.0]])
        tid, bb = 0, rect
        if rotate and not opaque_pass and k % 3 != 2:       # alpha-pass gradients under a rotation / skew: spans on general quads, AA edges
            tid = rotation_about(frame, rng, (x0 + x1) / 2, (y0 + y1) / 2, k, float(np.hypot(w, h)) * 0.5 if perspective else None)
            bb = rotated_bounds(rect)
            if perspective:
                rr = float(np.hypot(w, h)) * 1.4 + 4
                bb = ((x0 + x1) / 2 - rr, (y0 + y1) / 2 - rr, (x0 + x1) / 2 + rr, (y0 + y1) / 2 + rr)
        prims.append((rect, spec, lut, opaque_pass, tid, bb))

    gx = 4.0
    k = 0
    while True:
        w, h = float(rng.integers(3, 180)), float(rng.integers(20, band - 10))
        if gx + w + 4 > width:
            break
        off = 0.41 if (fractional and k % 2) else 0.0
        make((gx + off, 4.0 + off, gx + off + w, 4.0 + off + h), True, k)
        gx += float(np.ceil(w)) + 5.0
        k += 1
    for k in range(n):
        w, h = float(rng.uniform(2, 500)), float(rng.uniform(2, 400))
        px, py = float(rng.uniform(-40, width - 20)), float(rng.uniform(band, height - 10))
        if not fractional or k % 4 == 0:
            px, py, w, h = float(int(px)), float(int(py)), float(int(w) + 1), float(int(h) + 1)
        make((px, py, px + w, py + h), False, k)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        op, al = [], []
        for zi, (rect, spec, lut, opaque, tid, bb) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, spec, tid, task, (lut, 0, 0, 0))
            (op if opaque else al).append(frame.brush_instance(ph, CLIP_TASK_EMPTY, edge_flags=15) if tid else frame.brush_instance(ph, CLIP_TASK_EMPTY))
        if op:
            target.opaque.append(Step("brush_linear_gradient", "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32),
                                      None, "opaque"))
        if al:
            target.alpha.append(Step("brush_linear_gradient ALPHA_PASS", "PRIM_INSTANCES", np.array(al, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha"))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# brush_blend: CSS filters on picture surfaces (batch.rs:1726-1860).  The "picture" each
# prim filters is a premultiplied RGBA8 image in a linear-filtered atlas (standing in for the
# picture's render task); its ImageSource carries the UvRectKind::Quad corners that
# get_image_quad_uv reads (gpu_types.rs:949-970).
FILTER_CONTRAST, FILTER_GRAYSCALE, FILTER_HUE_ROTATE, FILTER_INVERT, FILTER_SATURATE, FILTER_SEPIA = 0, 1, 2, 3, 4, 5
FILTER_BRIGHTNESS, FILTER_COLOR_MATRIX, FILTER_SRGB_TO_LINEAR, FILTER_LINEAR_TO_SRGB, FILTER_FLOOD = 6, 7, 8, 9, 10
FILTER_COMPONENT_TRANSFER = 11
CT_IDENTITY, CT_TABLE, CT_DISCRETE, CT_LINEAR, CT_GAMMA = 0, 1, 2, 3, 4


def filter_grid(width=1024, height=1024, n=72, seed=71, atlas=1024, tile_filter=None, ops=None, only=None,
                fractional=True, masked=False, shader="blend", rotate=False, perspective=False):
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    pix = np.zeros((atlas, atlas, 4), np.uint8)
    srcs = []
    x = y = shelf = 0
    for i in range(20):
        w, h = int(rng.integers(32, 180)), int(rng.integers(32, 150))
        if x + w > atlas:
            x, y, shelf = 0, y + shelf, 0
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 0] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
        img[..., 1] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
        if i % 3 == 0:
            img[..., 3] = 255
        elif i % 3 == 1:
            img[..., 3] = np.where((xx // 8 + yy // 8) % 4 == 0, 0, img[..., 3])    # holes: alpha == 0 lanes
        img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)   # premultiplied
        pix[y:y + h, x:x + w] = img
        if i % 4 == 3:     # a sub-quad in homogeneous coordinates (w != 1)
            quad = [[0.125, 0.0625, 0.0, 1.0], [1.75, 0.125, 0.0, 2.0], [0.0625, 0.9375, 0.0, 1.0], [0.96875, 1.0, 0.0, 1.0]]
        else:
            quad = [[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]]
        addr = frame.gpu_cache.push([[x, y, x + w, y + h], [0.0, 0.0, 0.0, 0.0]] + quad)
        srcs.append((w, h, addr, i % 3 == 0))
        x += w
        shelf = max(shelf, h)
    t_atlas = TextureRef("picture_atlas", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, pixels=pix, upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)

    def filter_params(k):
        """-> (filter_mode, user_data.z) as batch.rs:1727-1756 / :1817-1823 encode them"""
        op = (ops or list(range(12)))[k % len(ops or range(12))]
        if op in (FILTER_CONTRAST, FILTER_GRAYSCALE, FILTER_INVERT, FILTER_SATURATE, FILTER_SEPIA, FILTER_BRIGHTNESS):
            amount = float(rng.choice([0.0, 0.25, 0.5, 1.0, 1.5, 2.25])) if k % 2 else float(rng.uniform(0.0, 2.0))
            return op, int(np.float32(amount) * np.float32(65536.0))
        if op == FILTER_HUE_ROTATE:
            angle = float(rng.uniform(0.0, 360.0))
            return op, int(np.float32(0.01745329251) * np.float32(angle) * np.float32(65536.0))
        if op in (FILTER_SRGB_TO_LINEAR, FILTER_LINEAR_TO_SRGB):
            return op, 0
        if op == FILTER_COLOR_MATRIX:
            m = rng.uniform(-0.5, 1.2, size=(4, 4)).astype(np.float32)
            off = rng.uniform(-0.2, 0.3, size=4).astype(np.float32)
            return op, frame.gpu_cache.push([list(r) for r in m] + [list(off)])
        if op == FILTER_FLOOD:
            a = float(rng.uniform(0.2, 1.0))
            return op, frame.gpu_cache.push([[float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), a]])
        # component transfer: per channel r, g, b, a
        funcs = [int(rng.integers(0, 5)) for _ in range(4)]
        blocks = []
        for f in funcs:
            if f in (CT_TABLE, CT_DISCRETE):
                nv = int(rng.integers(2, 7))
                vals = rng.uniform(-0.1, 1.1, size=nv).astype(np.float32)
                lut = np.empty(256, np.float32)
                for i in range(256):      # filterdata.rs: the 256-entry expansion of the table / discrete function
                    c = np.float32(i) / np.float32(255.0)
                    if f == CT_TABLE:
                        kk = min(int(c * (nv - 1)), nv - 2)
                        lut[i] = vals[kk] + (c * (nv - 1) - kk) * (vals[kk + 1] - vals[kk])
                    else:
                        lut[i] = vals[min(int(c * nv), nv - 1)]
                blocks += [list(lut[4 * j:4 * j + 4]) for j in range(64)]
            elif f == CT_LINEAR:
                blocks.append([float(rng.uniform(-1.5, 2.0)), float(rng.uniform(-0.3, 0.5)), 0.0, 0.0])
            elif f == CT_GAMMA:
                blocks.append([float(rng.uniform(0.5, 1.5)), float(rng.choice([0.4, 1.0, 2.2, 3.0])), float(rng.uniform(-0.1, 0.2)), 0.0])
        addr = frame.gpu_cache.push(blocks) if blocks else 0
        mode = FILTER_COMPONENT_TRANSFER | (funcs[0] << 28) | (funcs[1] << 24) | (funcs[2] << 20) | (funcs[3] << 16)
        return int(np.int32(np.uint32(mode))), addr

    prims = []
    band = 200
    gx = 4.0
    opaque_srcs = [s_ for s_ in srcs if s_[3]]
    for k in range(max(1, n // 6)):      # opaque pass: non-overlapping grid in the top band (see image_grid)
        sw, sh, addr, _ = opaque_srcs[k % len(opaque_srcs)]
        w, h = float(sw), float(min(sh, band - 8))
        if gx + w + 4 > width:
            break
        mode, ud = filter_params(k)
        if (mode & 0xffff) in (FILTER_COLOR_MATRIX, FILTER_FLOOD, FILTER_COMPONENT_TRANSFER):
            mode, ud = FILTER_SEPIA, 65536      # these may introduce transparency: never batched opaque (batch.rs:1760)
        off = 0.0 if (k % 2 == 0 or not fractional) else 0.37
        prims.append(((gx + off, 4.0 + off, gx + off + w, 4.0 + off + h), addr, True, mode, ud))
        gx += float(np.ceil(w)) + 6.0
    for k in range(n):
        sw, sh, addr, _ = srcs[int(rng.integers(0, len(srcs)))]
        sc = (1.0, 1.0, float(rng.uniform(1.1, 2.5)), 0.5, float(rng.uniform(0.4, 1.6)))[k % 5]
        w, h = sw * sc, sh * sc
        if k % 5 == 0 or not fractional:
            px, py = float(rng.integers(-20, width - 20)), float(rng.integers(band, height - 20))
            w, h = float(round(w)), float(round(h))
        else:
            px, py = float(rng.uniform(0, width - w)), float(rng.uniform(band, height - 40))
        mode, ud = filter_params(k + 1000)
        prims.append(((px, py, px + w, py + h), addr, False, mode, ud))
    tids = [0] * len(prims)
    if rotate:       # alpha-pass filters under a rotation / skew
        for k, pr in enumerate(prims):
            if not pr[2] and k % 3 != 2:
                tids[k] = rotation_about(frame, rng, (pr[0][0] + pr[0][2]) / 2, (pr[0][1] + pr[0][3]) / 2, k,
                                         float(np.hypot(pr[0][2] - pr[0][0], pr[0][3] - pr[0][1])) * 0.5 if perspective else None)
    t_mask, clip_tasks = None, [None] * len(prims)
    if masked:
        t_mask = TextureRef("clip_mask_atlas", 1024, 1024, G.GL_R8, G.GL_LINEAR, pixels=mask_atlas(1024), upload_format=G.GL_RED)
        frame.static_textures.append(t_mask)
        clip_tasks = prim_clip_tasks(rng, [rotated_bounds(p[0]) if tids[k] else p[0] for k, p in enumerate(prims)], 1024, True)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        op, al = [], []
        for zi, (rect, addr, opaque, mode, ud) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            bb = rotated_bounds(rect) if tids[zi] else rect
            if tids[zi] and perspective:
                rr = float(np.hypot(rect[2] - rect[0], rect[3] - rect[1])) * 1.4 + 4
                cxx, cyy = (rect[0] + rect[2]) / 2, (rect[1] + rect[3]) / 2
                bb = (cxx - rr, cyy - rr, cxx + rr, cyy + rr)
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            if shader == "opacity":     # brush_opacity: user data = (image source address, opacity * 65536)
                opv = (65536, 49152, 20000, 70000, 1, 0, 32768)[zi % 7] if opaque is False else 65536
                ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, 0, tids[zi], task, (addr, opv, 0, 0))
            else:
                ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, 0, tids[zi], task, (addr, mode, ud, 0))
            ct = None if opaque else clip_tasks[zi]
            clip_addr = CLIP_TASK_EMPTY if ct is None else frame.add_render_task(ct[0], 1.0, ct[1])
            # (BRUSH_FLAG_PERSPECTIVE_INTERPOLATION on every other perspective prim: perspective-correct uv; screen-linear otherwise)
            (op if opaque else al).append(frame.brush_instance(ph, clip_addr, edge_flags=15, brush_flags=1 if (perspective and tids[zi] and zi % 2 == 0) else 0))
        keys = ("brush_opacity", "brush_opacity ALPHA_PASS,ANTIALIASING") if shader == "opacity" else ("brush_blend", "brush_blend ALPHA_PASS")
        if op:
            target.opaque.append(Step(keys[0], "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32),
                                      None, "opaque", textures={0: t_atlas}))
        if al:
            target.alpha.append(Step(keys[1], "PRIM_INSTANCES", np.array(al, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha", textures={0: t_atlas, 9: t_mask} if masked else {0: t_atlas}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# ps_quad_mask: rounded-rectangle clips applied to quads.  The quad pattern is drawn first
# (ps_quad_textured, solid colour), then one MaskInstance per clip multiplies the same pixels by the
# clip's coverage (renderer: set_blend_mode_multiply; quad.rs / gpu_types.rs:618-624).
def quad_masks(width=1024, height=1024, n=90, seed=81, tile_filter=None, fractional=True, only=None, rotate=False, perspective=False):
    from .frame import QF_APPLY_DEVICE_CLIP
    QF_IS_MASK = 16
    rng, rects = random_rects(n, width, height, 48, 360, seed, fractional)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    alpha = np.round(rng.uniform(0.4, 1.0, size=n) * 255).astype(np.uint8)
    colors = premultiply(np.concatenate([rgb, alpha[:, None]], axis=1))
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    clips = []
    for i in range(n):
        x0, y0, x1, y1 = [float(v) for v in rects[i]]
        w, h = x1 - x0, y1 - y0
        inset = float(rng.uniform(0.0, 6.0)) if i % 3 else 0.0
        cr = (x0 + inset, y0 + inset, x1 - inset * 0.5, y1 - inset)
        mode = 1.0 if i % 7 == 3 else 0.0            # clip-out now and then
        if i % 2 == 0:       # uniform radius: FAST_PATH
            r = float(rng.uniform(2.0, min(w, h) * 0.45))
            addr = frame.gpu_buffer_f.push([list(cr), [r, r, r, r], [mode, 0.0, 0.0, 0.0]])
            clips.append((addr, True))
        else:
            rad = [float(rng.uniform(0.0, min(w, h) * 0.45)) for _ in range(8)]
            if i % 5 == 0:
                rad[0] = rad[1] = 0.0                 # a square corner
            # radii_top = (tl.w, tl.h, tr.w, tr.h), radii_bottom = (bl.w, bl.h, br.w, br.h)
            addr = frame.gpu_buffer_f.push([list(cr), rad[0:4], rad[4:8], [mode, 0.0, 0.0, 0.0]])
            clips.append((addr, False))
    tids = [0] * n
    bounds = np.array(rects, np.float64)
    if rotate:       # quads (and the clips that live in their space) under a rotation / skew
        for i in range(n):
            if i % 3 != 2:
                tids[i] = rotation_about(frame, rng, (rects[i][0] + rects[i][2]) / 2, (rects[i][1] + rects[i][3]) / 2, i,
                                         float(np.hypot(rects[i][2] - rects[i][0], rects[i][3] - rects[i][1])) * 0.5 if perspective else None)
                bounds[i] = rotated_bounds(tuple(float(v) for v in rects[i]))
                if perspective:
                    rr = float(np.hypot(rects[i][2] - rects[i][0], rects[i][3] - rects[i][1])) * 1.4 + 4
                    cxx, cyy = (rects[i][0] + rects[i][2]) / 2, (rects[i][1] + rects[i][3]) / 2
                    bounds[i] = (cxx - rr, cyy - rr, cxx + rr, cyy + rr)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        hit = np.nonzero((bounds[:, 0] < x1) & (bounds[:, 2] > x0) & (bounds[:, 1] < y1) & (bounds[:, 3] > y0))[0]
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        for i in hit:
            if only is not None and i not in only:
                continue
            big = (-BIG, -BIG, BIG, BIG)
            if tids[i]:
                q = frame.quad_instance(rects[i], big, colors[i], int(i + 1), task, transform_id=tids[i], quad_flags=0, edge_flags=15)
                m = frame.quad_instance(rects[i], big, (1.0, 1.0, 1.0, 1.0), int(i + 1), task, transform_id=tids[i],
                                        quad_flags=QF_IS_MASK, edge_flags=15)
            else:
                q = frame.quad_instance(rects[i], big, colors[i], int(i + 1), task)
                m = None
            target.alpha.append(Step("ps_quad_textured", "PRIM_INSTANCES", np.array([q], dtype=np.int32),
                                     "PremultipliedAlpha", "alpha", textures={}))
            if m is None:
                m = frame.quad_instance(rects[i], big, (1.0, 1.0, 1.0, 1.0), int(i + 1), task,
                                    quad_flags=QF_APPLY_DEVICE_CLIP | QF_IS_MASK)
            addr, fast = clips[i]
            inst = np.array([m + [0, addr, int(i % 4 == 1), 0]], dtype=np.int32)     # clip transform 0 (identity), address, space
            target.alpha.append(Step("ps_quad_mask FAST_PATH" if fast else "ps_quad_mask", "MASK", inst,
                                     "Multiply", "alpha", textures={}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


def filter_swatches(seed=91):
    """One 1024x512 tile of non-overlapping 64x48 brush_blend swatches, each a 1:1, integer-aligned copy
    of its own premultiplied atlas image through one filter -- simple enough for an independent
    restatement of the filter math in the tests.  frame.swatches lists
    (x, y, image RGBA u8 premultiplied, op, parameters)."""
    rng = np.random.default_rng(seed)
    W, H = 1024, 512
    frame = Frame(W, H, (1.0, 1.0, 1.0, 1.0))
    atlas = 512
    pix = np.zeros((atlas, atlas, 4), np.uint8)
    t_atlas = TextureRef("swatch_atlas", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, pixels=pix, upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)
    quad = [[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]]
    sw, sh = 64, 48
    specs = []
    for op in range(12):
        for rep in range(3):
            specs.append(op)
    frame.swatches = []
    tex = TextureRef("tile_0_0", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
    target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
    task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (0.0, 0.0))
    inst = []
    for k, op in enumerate(specs):
        ax, ay = (k % 7) * (sw + 4) + 2, (k // 7) * (sh + 4) + 2
        img = rng.integers(0, 256, size=(sh, sw, 4), dtype=np.uint8)
        if k % 3 == 0:
            img[..., 3] = 255
        if k % 3 == 2:
            img[::5, :, 3] = 0                                        # alpha == 0 rows
        img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)
        pix[ay:ay + sh, ax:ax + sw] = img
        addr = frame.gpu_cache.push([[ax, ay, ax + sw, ay + sh], [0.0, 0.0, 0.0, 0.0]] + quad)
        params = {}
        ud = 0
        mode = op
        if op in (FILTER_CONTRAST, FILTER_GRAYSCALE, FILTER_INVERT, FILTER_SATURATE, FILTER_SEPIA, FILTER_BRIGHTNESS):
            amount = float(rng.choice([0.25, 0.5, 1.0, 1.75])) if k % 2 else float(rng.uniform(0.0, 2.0))
            ud = int(np.float32(amount) * np.float32(65536.0))
        elif op == FILTER_HUE_ROTATE:
            ud = int(np.float32(0.01745329251) * np.float32(rng.uniform(0.0, 360.0)) * np.float32(65536.0))
        elif op == FILTER_COLOR_MATRIX:
            m = rng.uniform(-0.5, 1.2, size=(4, 4)).astype(np.float32)
            off = rng.uniform(-0.2, 0.3, size=4).astype(np.float32)
            ud = frame.gpu_cache.push([list(r) for r in m] + [list(off)])
            params = {"matrix": m, "offset": off}
        elif op == FILTER_FLOOD:
            c = np.array([rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.2, 1.0)], np.float32)
            ud = frame.gpu_cache.push([list(c)])
            params = {"flood": c}
        elif op == FILTER_COMPONENT_TRANSFER:
            funcs = [int(rng.integers(0, 5)) for _ in range(4)]
            blocks, tabs = [], []
            for f in funcs:
                if f in (CT_TABLE, CT_DISCRETE):
                    lut = rng.uniform(-0.1, 1.1, size=256).astype(np.float32)
                    blocks += [list(lut[4 * j:4 * j + 4]) for j in range(64)]
                    tabs.append(lut)
                elif f == CT_LINEAR:
                    v = np.array([rng.uniform(-1.5, 2.0), rng.uniform(-0.3, 0.5), 0.0, 0.0], np.float32)
                    blocks.append(list(v)); tabs.append(v)
                elif f == CT_GAMMA:
                    v = np.array([rng.uniform(0.5, 1.5), rng.choice([0.4, 1.0, 2.2, 3.0]), rng.uniform(-0.1, 0.2), 0.0], np.float32)
                    blocks.append(list(v)); tabs.append(v)
                else:
                    tabs.append(None)
            ud = frame.gpu_cache.push(blocks) if blocks else 0
            mode = int(np.int32(np.uint32(FILTER_COMPONENT_TRANSFER | (funcs[0] << 28) | (funcs[1] << 24) | (funcs[2] << 20) | (funcs[3] << 16))))
            params = {"funcs": funcs, "tables": tabs}
        params["user_data"] = ud
        px, py = 8 + (k % 12) * (sw + 12), 8 + (k // 12) * (sh + 12)
        rect = (float(px), float(py), float(px + sw), float(py + sh))
        ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), k + 1, 0, 0, task, (addr, mode, ud, 0))
        inst.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, edge_flags=15))
        frame.swatches.append((px, py, img, op, params))
    t_atlas.pixels = pix
    target.alpha.append(Step("brush_blend ALPHA_PASS", "PRIM_INSTANCES", np.array(inst, dtype=np.int32),
                             "PremultipliedAlpha", "alpha", textures={0: t_atlas}))
    frame.passes.append([target])
    frame.composite_tiles.append(CompositeTile(tex, (0.0, 0.0, float(TILE_W), float(TILE_H)), (0.0, 0.0, float(W), float(H)), opaque=True))
    return frame


# ---------------------------------------------------------------------------
# brush_mix_blend: CSS mix-blend-mode on isolated pictures (batch.rs:1931-2003).  sColor0 = the backdrop readback, sColor1 =
# the picture; prim user data = [MixBlendMode, backdrop ImageSource address, source ImageSource address, 0]; blended onto the
# target with the premultiplied-alpha key.  Modes Screen (2), Exclusion (11) and PlusLighter (16) are GL blend states in the
# renderer and never reach the shader (brush_mix_blend.glsl:314-320): one of each is drawn anyway, the shader's answer to
# them (opaque yellow under the backdrop's alpha) is part of what it does.
def _mix_picture(rng, w, h, k):
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    if k % 4 == 0:
        img[..., 3] = 255
    elif k % 4 == 1:
        img[..., 3] = np.where((xx // 6 + yy // 5) % 5 == 0, 0, img[..., 3])       # alpha == 0 lanes
    elif k % 4 == 2:
        img[..., 0] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
        img[..., 1] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
    img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)   # premultiplied
    return img


def mix_blend_swatches(seed=201):
    """One tile of non-overlapping 64x48 brush_mix_blend swatches, backdrop and source both 1:1 and integer-aligned, three per
    mode 1..16: frame.swatches lists (x, y, backdrop RGBA u8, source RGBA u8, mode) for an independent restatement."""
    rng = np.random.default_rng(seed)
    W, H = 1024, 512
    frame = Frame(W, H, (1.0, 1.0, 1.0, 1.0))
    atlas = 1024
    pb, ps = np.zeros((atlas, atlas, 4), np.uint8), np.zeros((atlas, atlas, 4), np.uint8)
    t_b = TextureRef("mix_backdrop_atlas", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, pixels=pb, upload_format=G.GL_BGRA)
    t_s = TextureRef("mix_source_atlas", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, pixels=ps, upload_format=G.GL_BGRA)
    frame.static_textures += [t_b, t_s]
    quad = [[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]]
    sw, sh = 64, 48
    frame.swatches = []
    tex = TextureRef("tile_0_0", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
    target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
    task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (0.0, 0.0))
    inst = []
    for k in range(48):
        mode = 1 + k // 3
        ax, ay = (k % 12) * (sw + 6) + 3, (k // 12) * (sh + 6) + 3
        bx, by = (k % 10) * (sw + 9) + 5, (k // 10) * (sh + 7) + 2
        ib, isrc = _mix_picture(rng, sw, sh, k), _mix_picture(rng, sw, sh, k + 1)
        if k % 3 == 1:
            ib[::7, :, :] = 0                      # backdrop alpha == 0 rows
        pb[by:by + sh, bx:bx + sw] = ib
        ps[ay:ay + sh, ax:ax + sw] = isrc
        a_b = frame.gpu_cache.push([[bx, by, bx + sw, by + sh], [0.0, 0.0, 0.0, 0.0]] + quad)
        a_s = frame.gpu_cache.push([[ax, ay, ax + sw, ay + sh], [0.0, 0.0, 0.0, 0.0]] + quad)
        px, py = 8 + (k % 12) * (sw + 12), 8 + (k // 12) * (sh + 12)
        rect = (float(px), float(py), float(px + sw), float(py + sh))
        ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), k + 1, 0, 0, task, (mode, a_b, a_s, 0))
        inst.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, edge_flags=15))
        frame.swatches.append((px, py, ib, isrc, mode))
    target.alpha.append(Step("brush_mix_blend ALPHA_PASS", "PRIM_INSTANCES", np.array(inst, dtype=np.int32),
                             "PremultipliedAlpha", "alpha", textures={0: t_b, 1: t_s}))
    frame.passes.append([target])
    frame.composite_tiles.append(CompositeTile(tex, (0.0, 0.0, float(TILE_W), float(TILE_H)), (0.0, 0.0, float(W), float(H)), opaque=True))
    return frame


def mix_blend_grid(width=1024, height=1024, n=80, seed=203, tile_filter=None, only=None, fractional=True, masked=False, rotate=False, force_aa=False,
                   perspective=False):
    """Overlapping, scaled, fractionally placed brush_mix_blend prims over the tile grid: backdrop and source pictures of
    different sizes (linear filtering on both), sub-quads in homogeneous coordinates on some sources, every mode."""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    atlas = 1024
    pb, ps = np.zeros((atlas, atlas, 4), np.uint8), np.zeros((atlas, atlas, 4), np.uint8)

    def pack(pix, count, seed_k):
        out, x, y, shelf = [], 0, 0, 0
        for i in range(count):
            w, h = int(rng.integers(24, 170)), int(rng.integers(24, 140))
            if x + w > atlas:
                x, y, shelf = 0, y + shelf, 0
            pix[y:y + h, x:x + w] = _mix_picture(rng, w, h, i + seed_k)
            if i % 5 == 4:      # a sub-quad in homogeneous coordinates (w != 1)
                quad = [[0.125, 0.0625, 0.0, 1.0], [1.75, 0.125, 0.0, 2.0], [0.0625, 0.9375, 0.0, 1.0], [0.96875, 1.0, 0.0, 1.0]]
            else:
                quad = [[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]]
            out.append((w, h, frame.gpu_cache.push([[x, y, x + w, y + h], [0.0, 0.0, 0.0, 0.0]] + quad)))
            x += w
            shelf = max(shelf, h)
        return out
    backs, srcs = pack(pb, 24, 0), pack(ps, 24, 2)
    t_b = TextureRef("mix_backdrop_atlas", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, pixels=pb, upload_format=G.GL_BGRA)
    t_s = TextureRef("mix_source_atlas", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR if seed % 2 else G.GL_NEAREST, pixels=ps, upload_format=G.GL_BGRA)
    frame.static_textures += [t_b, t_s]
    prims = []
    for k in range(n):
        bw, bh, a_b = backs[int(rng.integers(0, len(backs)))]
        sw, sh, a_s = srcs[int(rng.integers(0, len(srcs)))]
        sc = (1.0, 1.0, float(rng.uniform(1.1, 2.2)), 0.5, float(rng.uniform(0.5, 1.5)))[k % 5]
        w, h = sw * sc, sh * sc
        if k % 5 == 0 or not fractional:
            px, py, w, h = float(rng.integers(-20, width - 20)), float(rng.integers(-10, height - 20)), float(round(w)), float(round(h))
        else:
            px, py = float(rng.uniform(0, width - w)), float(rng.uniform(0, height - 30))
        prims.append(((px, py, px + w, py + h), 1 + k % 16, a_b, a_s))
    # rotate: two prims out of three sit under a rotation / skew about their centre (mix-blend-mode on a rotated stacking context:
    # the general-quad path with swgl_antiAlias on all four edges); force_aa: BRUSH_FLAG_FORCE_AA on axis-aligned ones
    tids = [0] * len(prims)
    # perspective: ... with a projective row on top (mix-blend-mode inside a 3-D context; every other such prim asks for
    # perspective-correct interpolation of the source's uv: BRUSH_FLAG_PERSPECTIVE_INTERPOLATION)
    if rotate or perspective:
        for k, pr in enumerate(prims):
            if k % 3 != 1:
                rad = 0.5 * float(np.hypot(pr[0][2] - pr[0][0], pr[0][3] - pr[0][1]))
                # (perspective == "clip": every other such prim reaches behind the camera plane -- clip_side, then the polygon's walk)
                tids[k] = rotation_about(frame, rng, (pr[0][0] + pr[0][2]) / 2, (pr[0][1] + pr[0][3]) / 2, k, rad if perspective else None,
                                         strength=(1.1, 2.6) if (perspective == "clip" and k % 2 == 0) else (0.15, 0.6))
    t_mask, clip_tasks = None, [None] * len(prims)
    if masked:
        t_mask = TextureRef("clip_mask_atlas", 1024, 1024, G.GL_R8, G.GL_LINEAR, pixels=mask_atlas(1024), upload_format=G.GL_RED)
        frame.static_textures.append(t_mask)
        clip_tasks = prim_clip_tasks(rng, [rotated_bounds(p[0]) if tids[k] else p[0] for k, p in enumerate(prims)], 1024, True)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        al = []
        for zi, (rect, mode, a_b, a_s) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            bb = rotated_bounds(rect) if tids[zi] else rect
            if tids[zi] and perspective:
                rr = float(np.hypot(rect[2] - rect[0], rect[3] - rect[1])) * 1.4 + 4
                cxx, cyy = (rect[0] + rect[2]) / 2, (rect[1] + rect[3]) / 2
                bb = (cxx - rr, cyy - rr, cxx + rr, cyy + rr)
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, 0, tids[zi], task, (mode, a_b, a_s, 0))
            ct = clip_tasks[zi]
            clip_addr = CLIP_TASK_EMPTY if ct is None else frame.add_render_task(ct[0], 1.0, ct[1])
            bf = 1024 if (force_aa and zi % 2 == 0 and not tids[zi]) else 0
            if perspective and tids[zi] and zi % 2 == 0:
                bf |= 1                                    # BRUSH_FLAG_PERSPECTIVE_INTERPOLATION
            al.append(frame.brush_instance(ph, clip_addr, edge_flags=15, brush_flags=bf))
        if al:
            target.alpha.append(Step("brush_mix_blend ALPHA_PASS", "PRIM_INSTANCES", np.array(al, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha", textures={0: t_b, 1: t_s, 9: t_mask} if masked else {0: t_b, 1: t_s}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# Rotated / skewed solid rectangles: the general convex-quad path of draw_quad_spans (rasterize.h:783-1055)
# with swgl_antiAlias on all four edges (brush.glsl: non-axis-aligned transforms take the antialiased branch).
def rotated_rects(width=1024, height=1024, n=70, seed=95, encoding="brush", tile_filter=None, only=None, opaque_frac=0.0,
                  perspective=False):
    """perspective: every transform gets a projective row as well (w = 1 + k . (p - centre), w > 0 over the rect), so the
    vertices' w differ and swgl takes draw_perspective (rasterize.h:1449-1547; wrench/benchmarks/transforms-simple.yaml is
    the reference's own scene of this kind)."""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    prims = []
    for i in range(n):
        w, h = float(rng.uniform(20, 320)), float(rng.uniform(20, 260))
        cx, cy = float(rng.uniform(0, width)), float(rng.uniform(0, height))
        th = float(rng.uniform(0, 2 * np.pi)) if i % 6 else float(rng.choice([np.pi / 4, np.pi / 6, 0.01]))
        sk = float(rng.uniform(-0.4, 0.4)) if i % 4 == 1 else 0.0
        c, s = np.cos(th), np.sin(th)
        a = np.array([[c, -s + sk * c], [s, c + sk * s]], np.float64)          # rotation (x skew) about (cx, cy)
        m = np.eye(4)
        m[:2, :2] = a
        m[:2, 3] = np.array([cx, cy]) - a @ np.array([cx, cy])
        if perspective:      # ("clip": every other prim reaches behind the camera plane)
            m = projective_about(a, cx, cy, float(np.hypot(w, h)) * 0.5 * (1.0 + abs(sk)), rng,
                                 strength=(1.1, 2.6) if (perspective == "clip" and i % 2 == 0) else (0.15, 0.6))
        inv = np.linalg.inv(m)
        tid = frame.add_transform(m.T.astype(np.float32), inv.T.astype(np.float32), axis_aligned=False)   # blocks = columns
        rgba = np.array([[rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, 256), rng.integers(90, 256)]], np.uint8)
        opaque = rng.uniform() < opaque_frac
        if opaque:
            rgba[0, 3] = 255
        col = premultiply(rgba)[0]
        rect = (cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2)
        r = float(np.hypot(w, h)) * (1.4 if perspective else 0.75) + 4
        prims.append((rect, tid, col, (cx - r, cy - r, cx + r, cy + r), opaque))
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        op, al = [], []
        for zi, (rect, tid, col, bb, opaque) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            if encoding == "brush":
                addr = frame.gpu_cache.push([list(col)])
                ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, addr, tid, task, (65535, 0, 0, 0))
                (op if opaque else al).append(frame.brush_instance(ph, CLIP_TASK_EMPTY, edge_flags=15))
            else:
                (op if opaque else al).append(frame.quad_instance(rect, (-BIG, -BIG, BIG, BIG), col, zi + 1, task, transform_id=tid,
                                                                  quad_flags=0, edge_flags=15))
        key = ("brush_solid", "brush_solid ALPHA_PASS") if encoding == "brush" else ("ps_quad_textured", "ps_quad_textured")
        if op:
            target.opaque.append(Step(key[0], "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32), None, "opaque", textures={}))
        if al:
            target.alpha.append(Step(key[1], "PRIM_INSTANCES", np.array(al, dtype=np.int32), "PremultipliedAlpha", "alpha", textures={}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


def transforms_simple(width=1024, height=1024, n=11, encoding="brush", tile_filter=None):
    """wrench/benchmarks/transforms-simple.yaml: a stacking context under rotate(45) (about the origin of its 1024 x 1024
    bounds) holding eleven full-size rects [255, 0, 0, 0.5] -- translucent solids on one shared non-axis-aligned transform,
    anti-aliased edges, alpha pass."""
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    c = sn = float(np.sqrt(0.5))
    m = np.eye(4)
    m[:2, :2] = [[c, -sn], [sn, c]]
    tid = frame.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=False)
    col = premultiply(np.array([[255, 0, 0, 128]], np.uint8))[0]
    rect = (0.0, 0.0, 1024.0, 1024.0)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        al = []
        for zi in range(n):
            if encoding == "brush":
                addr = frame.gpu_cache.push([list(col)])
                ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, addr, tid, task, (65535, 0, 0, 0))
                al.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, edge_flags=15))
            else:
                al.append(frame.quad_instance(rect, (-BIG, -BIG, BIG, BIG), col, zi + 1, task, transform_id=tid, quad_flags=0, edge_flags=15))
        key = "brush_solid ALPHA_PASS" if encoding == "brush" else "ps_quad_textured"
        target.alpha.append(Step(key, "PRIM_INSTANCES", np.array(al, dtype=np.int32), "PremultipliedAlpha", "alpha", textures={}))
        targets.append(target)
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        frame.composite_tiles.append(CompositeTile(tex, (float(x0), float(y0), float(x1), float(y1)),
                                                   (float(x0), float(y0), float(min(x1, width)), float(min(y1, height))), opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# Images under rotations / skews, and axis-aligned ones with BRUSH_FLAG_FORCE_AA: textured prims on the
# general-quad scanline walk with swgl_antiAlias edges (brush.glsl:150-170 -> swgl_antiAlias; blend.h DO_AA).
# `repeat`: through the ANTIALIASING,REPETITION image brush with tiling stretch sizes; `masked`: under
# swgl_clipMask as well.  Alpha pass only (AA needs blending, rasterize.h:414-441).
def rotated_images(width=1024, height=1024, n=60, seed=101, atlas=512, repeat=False, nearest=False, masked=False, tile_filter=None,
                   only=None, encoding="brush", perspective=False, dual=False):
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    pix = np.zeros((atlas, atlas, 4), np.uint8)
    srcs = []
    x = y = shelf = 0
    for i in range(16):
        w, h = int(rng.integers(12, 120)), int(rng.integers(12, 100))
        if x + w > atlas:
            x, y, shelf = 0, y + shelf, 0
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 0] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
        img[..., 1] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
        if i % 2 == 0:
            img[..., 3] = 255
        img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)
        pix[y:y + h, x:x + w] = img
        srcs.append((w, h, frame.gpu_cache.push([[x, y, x + w, y + h], [0.0, 0.0, 0.0, 0.0]]), (x, y, x + w, y + h)))
        x += w + 2
        shelf = max(shelf, h + 2)
    t_atlas = TextureRef("image_atlas", atlas, atlas, G.GL_RGBA8, G.GL_NEAREST if nearest else G.GL_LINEAR, pixels=pix,
                         upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)
    prims = []
    for i in range(n):
        sw, sh, addr, texels = srcs[int(rng.integers(0, len(srcs)))]
        sc = float(rng.uniform(0.6, 3.0))
        w, h = sw * sc, sh * sc * float(rng.uniform(0.7, 1.4))
        cx, cy = float(rng.uniform(0, width)), float(rng.uniform(0, height))
        flags = 0
        if i % 6 == 5 and perspective != "all":        # axis-aligned, anti-aliased on request
            tid = 0
            flags = 1024
            cx, cy = cx + float(rng.uniform(0, 1)), cy + float(rng.uniform(0, 1))
        else:
            th = float(rng.uniform(0, 2 * np.pi)) if i % 6 else float(rng.choice([np.pi / 4, np.pi / 2, 0.01]))
            sk = float(rng.uniform(-0.4, 0.4)) if i % 4 == 1 else 0.0
            c, sn = np.cos(th), np.sin(th)
            a = np.array([[c, -sn + sk * c], [sn, c + sk * sn]], np.float64)
            m = np.eye(4)
            m[:2, :2] = a
            m[:2, 3] = np.array([cx, cy]) - a @ np.array([cx, cy])
            if perspective:      # (see rotated_rects)
                m = projective_about(a, cx, cy, float(np.hypot(w, h)) * 0.5 * (1.0 + abs(sk)), rng,
                                     strength=(1.1, 2.6) if (perspective == "clip" and i % 3 == 0) else (0.15, 0.6))
                if i % 2 == 0:
                    flags = 1         # BRUSH_FLAG_PERSPECTIVE_INTERPOLATION (brush encoding): perspective-correct uv
            tid = frame.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=False)
        rect = (cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2)
        rad = float(np.hypot(w, h)) * (1.4 if perspective else 0.75) + 4
        st = (-1.0, -1.0)
        if repeat and i % 3 != 2:
            st = (sw * float(rng.uniform(0.5, 1.5)), sh * float(rng.uniform(0.5, 1.5))) if i % 3 else (float(sw), float(sh))
        opacity = 1.0 if i % 3 else float(rng.uniform(0.3, 0.9))
        prims.append((rect, tid, addr, (cx - rad, cy - rad, cx + rad, cy + rad), flags, st, opacity, texels))
    t_mask, clip_tasks = None, [None] * len(prims)
    if masked:
        t_mask = TextureRef("clip_mask_atlas", 1024, 1024, G.GL_R8, G.GL_LINEAR, pixels=mask_atlas(1024), upload_format=G.GL_RED)
        frame.static_textures.append(t_mask)
        clip_tasks = prim_clip_tasks(rng, [p[3] for p in prims], 1024, True)
    key = "brush_image ALPHA_PASS,ANTIALIASING,REPETITION,TEXTURE_2D" if repeat else "brush_image ALPHA_PASS,TEXTURE_2D"
    if dual:        # (with repeat: the dual-source REPETITION key on rotated / projected, anti-aliased prims -- see image_repeat)
        key = "brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION,TEXTURE_2D"
    if encoding == "quad":
        key = "ps_quad_textured"
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        al = []
        for zi, (rect, tid, addr, bb, flags, st, opacity, texels) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            if encoding == "quad":      # ps_quad_textured sampling the atlas (pattern colour x texture)
                al.append(frame.quad_instance(rect, (-BIG, -BIG, BIG, BIG), (opacity, opacity, opacity, opacity), zi + 1, task,
                                              transform_id=tid, quad_flags=0, edge_flags=15 if (tid or flags) else 0,
                                              uv_rect=tuple(float(v) for v in texels)))
                continue
            col = [1.0, 1.0, 1.0, 1.0]
            ud = (4 | (1 << 16), 0, int(round(opacity * 65535.0)), 0)
            if dual:
                a = (0.35, 0.6, 0.85, 1.0)[zi % 4]
                col = [((zi * 37) % 256) / 255.0 * a, ((zi * 91) % 256) / 255.0 * a, ((zi * 53) % 256) / 255.0 * a, a]
                ud = ((1, 5, 4)[zi % 3] | (1 << 16), 0, int(round(opacity * 65535.0)), 0)
            spec = frame.gpu_cache.push([col, [0.0, 0.0, 0.0, 0.0], [st[0], st[1], 0.0, 0.0]])
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, spec, tid, task, ud)
            ct = clip_tasks[zi]
            clip_addr = CLIP_TASK_EMPTY if ct is None else frame.add_render_task(ct[0], 1.0, ct[1])
            al.append(frame.brush_instance(ph, clip_addr, brush_flags=flags, edge_flags=15, resource_address=addr))
        if al:
            target.alpha.append(Step(key, "PRIM_INSTANCES", np.array(al, dtype=np.int32), "SubpixelDualSource" if dual else "PremultipliedAlpha", "alpha",
                                     textures={0: t_atlas, 9: t_mask} if masked else {0: t_atlas}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# ps_split_composite: the planes of a preserve-3d context after plane splitting.  picture.rs:6571-6622 sorts the split polygons
# back to front, maps each polygon's four world points back into its plane's local space and leaves them in the GPU cache
# (two blocks: p0.xy p1.xy | p2.xy p3.xy); batch.rs:1985-2080 then draws each one with BatchKind::SplitComposite /
# BlendMode::PremultipliedAlpha: prim header = (the child picture's rect, its clip rect, transform of its spatial node), user
# data = (uv rect address of the child's surface, BrushFlags::PERSPECTIVE_INTERPOLATION, 0, clip task address), instance =
# SplitCompositeInstance (gpu_types.rs:531-551: header index, polygons address, z, render task address).
def split_composites(width=1024, height=1024, n=60, seed=211, atlas=512, masked=False, tile_filter=None, only=None, perspective=False,
                     nearest=False, pin=False):
    """pin: every plane faces the screen at whole device pixels and 1:1 scale, every polygon is an axis-aligned part of its
    plane (whole, a left / top part, or the whole in the other winding) -- what the tests' numpy pin of the program restates."""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    pix = np.zeros((atlas, atlas, 4), np.uint8)
    srcs = []
    x = y = shelf = 0
    unit = [[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]]
    for i in range(16):
        w, h = int(rng.integers(24, 120)), int(rng.integers(24, 100))
        if x + w > atlas:
            x, y, shelf = 0, y + shelf, 0
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 0] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
        img[..., 1] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
        if i % 2 == 0:
            img[..., 3] = 255
        img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)
        pix[y:y + h, x:x + w] = img
        # the child surface's ImageSource: uv rect, user data, then the four homogeneous corners get_image_quad_uv blends
        quad = unit if i % 5 != 4 else [[0.125, 0.0625, 0.0, 1.0], [1.75, 0.125, 0.0, 2.0], [0.0625, 0.9375, 0.0, 1.0], [0.96875, 1.0, 0.0, 1.0]]
        srcs.append((w, h, frame.gpu_cache.push([[x, y, x + w, y + h], [0.0, 0.0, 0.0, 0.0]] + quad), quad is unit))
        x += w + 2
        shelf = max(shelf, h + 2)
    t_atlas = TextureRef("split_surfaces", atlas, atlas, G.GL_RGBA8, G.GL_NEAREST if nearest else G.GL_LINEAR, pixels=pix,
                         upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)
    prims = []
    for i in range(n):
        sw, sh, addr, unitq = srcs[int(rng.integers(0, len(srcs)))]
        sc = float(rng.uniform(0.8, 3.0))
        w, h = sw * sc, sh * sc * float(rng.uniform(0.7, 1.4))
        cx, cy = float(rng.uniform(0, width)), float(rng.uniform(0, height))
        if pin:
            while not unitq:       # (surfaces whose image quad is the unit square only)
                sw, sh, addr, unitq = srcs[int(rng.integers(0, len(srcs)))]
            w, h = float(sw), float(sh)
            cx, cy = float(int(cx)) + (sw % 2) * 0.5, float(int(cy)) + (sh % 2) * 0.5
            tid = 0
        elif i % 7 == 6 and perspective != "all":
            tid = 0            # a plane that faces the screen
            if i % 2:
                cx, cy = cx + float(rng.uniform(0, 1)), cy + float(rng.uniform(0, 1))
        else:
            th = float(rng.uniform(0, 2 * np.pi)) if i % 6 else float(rng.choice([np.pi / 4, np.pi / 2, 0.01]))
            sk = float(rng.uniform(-0.4, 0.4)) if i % 4 == 1 else 0.0
            c, sn = np.cos(th), np.sin(th)
            a = np.array([[c, -sn + sk * c], [sn, c + sk * sn]], np.float64)
            m = np.eye(4)
            m[:2, :2] = a
            m[:2, 3] = np.array([cx, cy]) - a @ np.array([cx, cy])
            if perspective:
                m = projective_about(a, cx, cy, float(np.hypot(w, h)) * 0.5 * (1.0 + abs(sk)), rng,
                                     strength=(1.1, 2.6) if (perspective == "clip" and i % 3 == 0) else (0.15, 0.6))
            tid = frame.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=False)
        rect = (cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2)
        # the polygon: what the other planes left of this one -- a convex quad inside the rect, its points in order around it
        x0, y0, x1, y1 = rect
        kind = i % 5
        if pin:
            k = i % 4
            ax_, ay_ = float(int(rng.integers(4, max(5, sw - 3)))), float(int(rng.integers(4, max(5, sh - 3))))
            if k == 0:
                pts = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
            elif k == 1:       # split by a vertical plane: the left part
                pts = [(x0, y0), (x0 + ax_, y0), (x0 + ax_, y1), (x0, y1)]
            elif k == 2:       # ... by a horizontal one: the top part
                pts = [(x0, y0), (x1, y0), (x1, y0 + ay_), (x0, y0 + ay_)]
            else:              # the whole rect in the other winding, starting from another corner
                pts = [(x1, y1), (x1, y0), (x0, y0), (x0, y1)]
        elif kind == 0:      # untouched: the whole rect
            pts = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
        elif kind == 1:    # one cut: a trapezoid
            a0, a1 = float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.2, 0.8))
            pts = [(x0, y0), (x0 + w * a0, y0), (x0 + w * a1, y1), (x0, y1)]
        elif kind == 2:    # a corner cut off twice: a general quad
            pts = [(x0 + w * float(rng.uniform(0.0, 0.4)), y0), (x1, y0 + h * float(rng.uniform(0.0, 0.4))),
                   (x1 - w * float(rng.uniform(0.0, 0.4)), y1), (x0, y1 - h * float(rng.uniform(0.0, 0.4)))]
        elif kind == 3:    # a triangle: plane_split hands four points, two of them equal
            pts = [(x0, y0), (x1, y0 + h * float(rng.uniform(0.3, 1.0))), (x0 + w * float(rng.uniform(0.0, 0.7)), y1)]
            pts.append(pts[2])
        else:              # the other winding
            a0 = float(rng.uniform(0.3, 0.9))
            pts = [(x0, y0), (x0, y1), (x0 + w * a0, y1), (x1, y0)]
        poly = frame.gpu_cache.push([[pts[0][0], pts[0][1], pts[1][0], pts[1][1]], [pts[2][0], pts[2][1], pts[3][0], pts[3][1]]])
        rad = float(np.hypot(w, h)) * (1.4 if perspective else 0.75) + 4
        persp_flag = 0 if i % 4 == 3 else 1          # (WebRender always passes PERSPECTIVE_INTERPOLATION; the program reads the flag)
        prims.append((rect, tid, addr, (cx - rad, cy - rad, cx + rad, cy + rad), poly, persp_flag))
    t_mask, clip_tasks = None, [None] * len(prims)
    if masked:
        t_mask = TextureRef("clip_mask_atlas", 1024, 1024, G.GL_R8, G.GL_LINEAR, pixels=mask_atlas(1024), upload_format=G.GL_RED)
        frame.static_textures.append(t_mask)
        clip_tasks = prim_clip_tasks(rng, [p[3] for p in prims], 1024, True)
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        al = []
        for zi, (rect, tid, addr, bb, poly, persp_flag) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            ct = clip_tasks[zi]
            clip_addr = CLIP_TASK_EMPTY if ct is None else frame.add_render_task(ct[0], 1.0, ct[1])
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, 0, tid, task, (addr, persp_flag, 0, clip_addr))
            al.append([ph, poly, zi + 1, task])
        if al:
            target.alpha.append(Step("ps_split_composite", "PRIM_INSTANCES", np.array(al, dtype=np.int32), "PremultipliedAlpha", "alpha",
                                     textures={0: t_atlas, 9: t_mask} if masked else {0: t_atlas}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# Opaque occluders on top of any tile scene: axis-aligned opaque brush_solid rects in the opaque (depth-writing) pass of
# every picture-cache tile, at z ids scattered through the scene's own.  Whatever they partly hide -- opaque-pass images
# and gradients drawn after them, everything in the depth-tested alpha pass -- is drawn by swgl one depth run at a time
# (draw_depth_span, rasterize.h:612-664): the 4-pixel chunk phase, the span-shader / main() split, the filter decisions
# and the accumulated interpolants restart wherever a run of passing pixels starts.  `first`: tiles for which the
# occluder batch is drawn ahead of the scene's own opaque batches (the others draw it last, so opaque content that is
# nearer than an occluder is hidden by depth and content farther away is overwritten instead).
def add_occluders(frame, n=40, seed=7, zmax=100, wmin=12, wmax=260, first=lambda tx, ty: (tx + ty) % 2 == 0):
    rng = np.random.default_rng(seed)
    occ = []
    for i in range(n):
        w, h = float(rng.integers(wmin, wmax)), float(rng.integers(wmin, wmax))
        x, y = float(rng.integers(-20, frame.width)), float(rng.integers(-20, frame.height))
        if i % 3 == 0:
            x, y, w, h = x + float(rng.uniform(0, 1)), y + float(rng.uniform(0, 1)), w + float(rng.uniform(0, 1)), h + float(rng.uniform(0, 1))
        rgb = rng.integers(0, 256, size=3)
        col = [float(rgb[0]) / 255.0, float(rgb[1]) / 255.0, float(rgb[2]) / 255.0, 1.0]
        occ.append(((x, y, x + w, y + h), int(rng.integers(1, zmax + 1)), frame.gpu_cache.push([col])))
    for targets in frame.passes:
        for target in targets:
            name = target.texture.name
            if target.kind != "picture_tile" or not name.startswith("tile_"):
                continue
            tx, ty = (int(v) for v in name.split("_")[1:3])
            ox, oy = tx * TILE_W, ty * TILE_H
            task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
            inst = []
            for rect, z, addr in sorted(occ, key=lambda o: -o[1]):          # front to back
                if not (rect[0] < ox + TILE_W and rect[2] > ox and rect[1] < oy + TILE_H and rect[3] > oy):
                    continue
                ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), z, addr, 0, task, (65535, 0, 0, 0))
                inst.append(frame.brush_instance(ph, CLIP_TASK_EMPTY))
            if not inst:
                continue
            step = Step("brush_solid", "PRIM_INSTANCES", np.array(inst, dtype=np.int32), None, "opaque", textures={})
            if first(tx, ty):
                target.opaque.insert(0, step)
            else:
                target.opaque.append(step)
    return frame


# Thin opaque slivers at a fixed pitch across a band of every picture tile, nearer than the scene's own content and drawn
# ahead of it: a wide image behind them is cut into one depth run per gap (draw_depth_span restarts the span shader in each).
# With a small pitch a row has more runs / a strip more occluders than the backend's depth-run tables hold: such a prim must be
# REPORTED (GL_INVALID_OPERATION at Finish), never drawn from the span start as if nothing hid it.
def add_slivers(frame, pitch=40, width=2, y0=0, y1=240, z=1 << 20):
    addr = frame.gpu_cache.push([[0.1, 0.2, 0.3, 1.0]])
    for targets in frame.passes:
        for target in targets:
            name = target.texture.name
            if target.kind != "picture_tile" or not name.startswith("tile_"):
                continue
            tx, ty = (int(v) for v in name.split("_")[1:3])
            ox, oy = tx * TILE_W, ty * TILE_H
            if not (y0 < oy + TILE_H and y1 > oy):
                continue
            task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
            inst = []
            for x in range(ox + 3, ox + TILE_W, pitch):
                ph = frame.add_prim_header((float(x), float(y0), float(x + width), float(y1)), (-BIG, -BIG, BIG, BIG), z, addr, 0, task, (65535, 0, 0, 0))
                inst.append(frame.brush_instance(ph, CLIP_TASK_EMPTY))
            target.opaque.insert(0, Step("brush_solid", "PRIM_INSTANCES", np.array(inst, dtype=np.int32), None, "opaque", textures={}))
    return frame


# ---------------------------------------------------------------------------
# Flattened depth rows (rasterize.h:1222-1232, 1021-1031): a few solid rects under projective transforms drawn AHEAD of a
# scene's own content in every picture tile -- some in the opaque pass (first batch: depth-writing perspective prims),
# some at the head of the alpha pass (depth-tested only).  Every row their spans touch has its depth run array flattened,
# and swgl then draws every later depth-tested prim on those rows chunk by chunk through main(), from the span start,
# instead of handing the span shader one depth run at a time: the scene's families are exercised in that mode on the rows
# under the rects and in the ordinary one everywhere else.
def add_perspective_underlay(frame, n=10, seed=77, opaque_every=2):
    rng = np.random.default_rng(seed)
    under = []
    for i in range(n):
        w, h = float(rng.uniform(60, 360)), float(rng.uniform(30, 200))
        cx, cy = float(rng.uniform(0, frame.width)), float(rng.uniform(0, frame.height))
        th = float(rng.uniform(0, 2 * np.pi))
        c, s = np.cos(th), np.sin(th)
        a = np.array([[c, -s], [s, c]], np.float64)
        m = projective_about(a, cx, cy, float(np.hypot(w, h)) * 0.5, rng)
        tid = frame.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=False)
        opaque = i % opaque_every == 0
        rgba = np.array([[rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, 256), 255 if opaque else rng.integers(90, 230)]], np.uint8)
        col = premultiply(rgba)[0]
        r = float(np.hypot(w, h)) * 1.4 + 4
        under.append(((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), tid, frame.gpu_cache.push([list(col)]), (cx - r, cy - r, cx + r, cy + r), opaque))
    for targets in frame.passes:
        for target in targets:
            name = target.texture.name
            if target.kind != "picture_tile" or not name.startswith("tile_"):
                continue
            tx, ty = (int(v) for v in name.split("_")[1:3])
            ox, oy = tx * TILE_W, ty * TILE_H
            task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
            op, al = [], []
            for zi, (rect, tid, addr, bb, opaque) in enumerate(under):
                if not (bb[0] < ox + TILE_W and bb[2] > ox and bb[1] < oy + TILE_H and bb[3] > oy):
                    continue
                ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, addr, tid, task, (65535, 0, 0, 0))
                (op if opaque else al).append(frame.brush_instance(ph, CLIP_TASK_EMPTY, edge_flags=15))
            if op:
                target.opaque.insert(0, Step("brush_solid", "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32), None, "opaque", textures={}))
            if al:
                target.alpha.insert(0, Step("brush_solid ALPHA_PASS", "PRIM_INSTANCES", np.array(al, dtype=np.int32), "PremultipliedAlpha", "alpha", textures={}))
    return frame


# ---------------------------------------------------------------------------
# Every blend key of swgl's table (gl.cc:614-645, blend.h:466-701) over varied destination content: one batch of
# overlapping translucent prims per blend state -- solids (brush_solid ALPHA_PASS) and images (brush_image ALPHA_PASS,
# 1:1 and scaled) alternate, so both the solid-colour path and the textured paths go through the blend stage with each
# key.  BlendMode names as in Device.set_blend_mode; "Advanced:*" are the KHR_blend_equation_advanced equations the
# renderer uses for mix-blend-mode pictures on a backend that advertises the extension (device/gl.rs:3980-4025).
BLEND_STATES = ("PremultipliedAlpha", "Alpha", "PremultipliedDestOut", "Multiply", "PlusLighter", "Min", "Max",
                "SubpixelConstantTextColor:0.8,0.3,0.55,0.7",
                "Advanced:Multiply", "Advanced:Screen", "Advanced:Overlay", "Advanced:Darken", "Advanced:Lighten",
                "Advanced:ColorDodge", "Advanced:ColorBurn", "Advanced:HardLight", "Advanced:SoftLight", "Advanced:Difference",
                "Advanced:Exclusion", "Advanced:Hue", "Advanced:Saturation", "Advanced:Color", "Advanced:Luminosity")


def blend_modes(width=1024, height=1024, per_state=14, seed=111, states=BLEND_STATES, atlas=256, clear=(0.2, 0.35, 0.5, 0.75), only=None):
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    pix = rng.integers(0, 256, size=(atlas, atlas, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:atlas, 0:atlas]
    pix[..., 3] = ((xx + yy) * 255 // (2 * atlas - 2)).astype(np.uint8)          # alpha ramp 0 .. 255
    pix[..., :3] = (pix[..., :3].astype(np.uint16) * pix[..., 3:4] // 255).astype(np.uint8)
    t_atlas = TextureRef("blend_atlas", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, pixels=pix, upload_format=G.GL_BGRA)
    frame.static_textures.append(t_atlas)
    img_addr = frame.gpu_cache.push([[0, 0, atlas, atlas], [0.0, 0.0, 0.0, 0.0]])
    groups = []          # (state, [(rect, kind, colour / opacity)])
    # background content first, so the destination is not uniform: premultiplied translucent rects
    bg = []
    for i in range(30):
        w, h = float(rng.integers(80, 400)), float(rng.integers(80, 400))
        x, y = float(rng.integers(-40, width - 40)), float(rng.integers(-40, height - 40))
        rgba = np.array([[rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, 256), rng.integers(30, 256)]], np.uint8)
        bg.append(((x, y, x + w, y + h), "solid", premultiply(rgba)[0]))
    groups.append(("PremultipliedAlpha", bg))
    for st in states:
        prims = []
        for i in range(per_state):
            w, h = float(rng.integers(40, 300)), float(rng.integers(40, 300))
            x, y = float(rng.integers(-20, width - 20)), float(rng.integers(-20, height - 20))
            if i % 3 == 2:
                x, y = x + float(rng.uniform(0, 1)), y + float(rng.uniform(0, 1))
            if i % 2 == 0:
                rgba = np.array([[rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, 256)]], np.uint8)
                col = premultiply(rgba)[0] if st != "Alpha" else rgba[0].astype(np.float32) / np.float32(255.0)
                prims.append(((x, y, x + w, y + h), "solid", col))
            else:
                if i % 4 == 1:
                    w, h = float(atlas), float(atlas)                               # 1:1
                    x, y = float(int(x)), float(int(y))
                prims.append(((x, y, x + w, y + h), "image", 1.0 if i % 3 else float(rng.uniform(0.3, 0.9))))
        groups.append((st, prims))
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=clear, clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        z = 1
        for gi, (st, prims) in enumerate(groups):
            if only is not None and gi != 0 and st not in only:
                continue
            sol, img = [], []
            for rect, kind, val in prims:
                z += 1
                if not (rect[0] < x1 and rect[2] > x0 and rect[1] < y1 and rect[3] > y0):
                    continue
                if kind == "solid":
                    addr = frame.gpu_cache.push([[float(v) for v in val]])
                    ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), z, addr, 0, task, (65535, 0, 0, 0))
                    sol.append(frame.brush_instance(ph, CLIP_TASK_EMPTY))
                else:
                    spec = frame.gpu_cache.push([[1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [-1.0, -1.0, 0.0, 0.0]])
                    ud = (4 | (1 << 16), 0, int(round(val * 65535.0)), 0)
                    ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), z, spec, 0, task, ud)
                    img.append(frame.brush_instance(ph, CLIP_TASK_EMPTY, resource_address=img_addr))
            if sol:
                target.alpha.append(Step("brush_solid ALPHA_PASS", "PRIM_INSTANCES", np.array(sol, dtype=np.int32), st, "alpha", textures={}))
            if img:
                # BlendMode::Advanced batches take the ADVANCED_BLEND key of the program (shade.rs:440-468)
                key = "brush_image ADVANCED_BLEND,ALPHA_PASS,TEXTURE_2D" if st.startswith("Advanced:") else "brush_image ALPHA_PASS,TEXTURE_2D"
                target.alpha.append(Step(key, "PRIM_INSTANCES", np.array(img, dtype=np.int32), st, "alpha", textures={0: t_atlas}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        frame.composite_tiles.append(CompositeTile(tex, rect, (float(x0), float(y0), float(min(x1, width)), float(min(y1, height))), opaque=True))
    frame.passes.append(targets)
    return frame


def make_workload(workload, **kw):
    """The BASELINE.json configs by name ("cfg1" .. "cfg5"), as bench.py and the multi-GPU harness build them."""
    if workload == "cfg2":
        return cfg2_overlapping_rects(**kw)
    if workload == "cfg5":
        return cfg5_many_rects(**kw)
    if workload == "cfg1":
        return cfg1_solid_colors(**kw)
    if workload == "cfg4":
        kw.pop("encoding", None)
        return cfg4_box_shadow(dps=2.0, **kw)
    if workload == "cfg3":
        kw.pop("encoding", None)
        return cfg3_text(**kw)
    if workload == "transforms":          # wrench/benchmarks/transforms-simple.yaml (not a BASELINE config)
        return transforms_simple(**kw)
    if workload == "simple-batching":     # wrench/benchmarks/simple-batching.yaml at the 4K target
        return simple_batching(width=3840, height=2160, **kw)
    from . import wrench_scenes           # the rest of wrench/benchmarks/benchmarks.list
    if workload in wrench_scenes.WORKLOADS:
        kw.pop("encoding", None)
        return wrench_scenes.WORKLOADS[workload](**kw)
    raise SystemExit(f"unknown workload {workload}")


# ---------------------------------------------------------------------------
# cs_border_solid: the segments of solid borders (four corners, four edges) rendered into the texture cache, one render task
# each (border.rs:654-1043 create_border_segments / add_corner_segment / add_edge_segment -> BorderSegmentCacheKey ->
# build_border_instances :1380-1480 -> render_target.rs add_border_segment_task_to_target: task_origin = the task's origin in
# the cache texture).  brush_image then stretches the cached segments over the border's layout rects; that half is the
# image path and is exercised elsewhere.
BORDER_DTYPE = np.dtype([("origin", "<f4", (2,)), ("rect", "<f4", (4,)), ("c0", "<f4", (4,)), ("c1", "<f4", (4,)), ("flags", "<i4"),
                         ("widths", "<f4", (2,)), ("radii", "<f4", (2,)), ("cp", "<f4", (8,))])      # BorderInstance, gpu_types.rs:193-202
SEG_TL, SEG_TR, SEG_BR, SEG_BL, SEG_LEFT, SEG_TOP, SEG_RIGHT, SEG_BOTTOM = range(8)     # BorderSegment, border.rs:123-132
BORDER_STYLE_SOLID = 1


def border_solid(n=40, seed=131, atlas=1024):
    rng = np.random.default_rng(seed)
    frame = Frame(atlas, atlas, (1.0, 1.0, 1.0, 1.0))
    t_cache = TextureRef("border_cache", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, render_target=True)
    tgt = Target(t_cache, "texture_cache", clear_color=(0.0, 0.0, 0.0, 0.0))
    inst = []
    x = y = shelf = 2

    def place(w, h):
        nonlocal x, y, shelf
        w, h = int(np.ceil(w)), int(np.ceil(h))
        if x + w + 2 > atlas:
            x, y, shelf = 2, y + shelf + 2, 0
        if y + h + 2 > atlas:
            return None
        o = (float(x), float(y))
        x += w + 2
        shelf = max(shelf, h)
        return o

    def premul(c):
        return [c[0] * c[3], c[1] * c[3], c[2] * c[3], c[3]]

    for k in range(n):
        scale = float(rng.choice([1.0, 1.0, 1.5, 2.0]))
        W, H = float(rng.uniform(60, 260)), float(rng.uniform(50, 200))
        widths = [float(rng.choice([1.0, 2.0, 3.0, 6.5, 12.0, 20.0])) for _ in range(4)]        # left, top, right, bottom
        if k % 7 == 3:
            widths[int(rng.integers(0, 4))] = 0.0
        rad = []                                                                                # tl, tr, br, bl: (w, h)
        for c in range(4):
            if (k + c) % 5 == 0:
                rad.append((0.0, 0.0))
            elif (k + c) % 5 == 1:
                r_ = float(rng.uniform(2, min(W, H) * 0.45)); rad.append((r_, r_))
            else:
                rad.append((float(rng.uniform(1, W * 0.45)), float(rng.uniform(1, H * 0.45))))
        cols = [[float(v) for v in rng.uniform(0, 1, size=3)] + [float(rng.choice([1.0, 1.0, 0.6]))] for _ in range(4)]   # left, top, right, bottom
        if k % 3 == 0:
            cols = [cols[0]] * 4
        do_aa = k % 6 != 5
        l, t, r, b = widths
        # corner segment rects in the border's local space (border.rs:700-760): max(width, radius) on each axis
        tl = (0.0, 0.0, max(l, rad[0][0]), max(t, rad[0][1]))
        tr = (W - max(r, rad[1][0]), 0.0, W, max(t, rad[1][1]))
        br = (W - max(r, rad[2][0]), H - max(b, rad[2][1]), W, H)
        bl = (0.0, H - max(b, rad[3][1]), max(l, rad[3][0]), H)
        outer = {SEG_TL: (0.0, 0.0), SEG_TR: (W, 0.0), SEG_BR: (W, H), SEG_BL: (0.0, H)}

        def corner(seg, rect, side0, side1, wid, radius, h_seg, v_seg, h_rad, v_rad):
            # add_corner_segment: adjacent corners that do not reach this segment collapse to a point on it (border.rs:1095-1160)
            if wid[0] <= 0.0 and wid[1] <= 0.0:
                return
            ho, hr = outer[h_seg], h_rad
            vo, vr = outer[v_seg], v_rad
            if seg in (SEG_TL, SEG_BL):
                if not (ho[0] - hr[0] < rect[2]):
                    ho, hr = (rect[2], rect[1] if seg == SEG_TL else rect[3]), (0.0, 0.0)
            else:
                if not (ho[0] + hr[0] > rect[0]):
                    ho, hr = (rect[0], rect[1] if seg == SEG_TR else rect[3]), (0.0, 0.0)
            if seg in (SEG_TL, SEG_TR):
                if not (vo[1] - vr[1] < rect[3]):
                    vo, vr = (rect[0] if seg == SEG_TL else rect[2], rect[3]), (0.0, 0.0)
            else:
                if not (vo[1] + vr[1] > rect[1]):
                    vo, vr = (rect[2] if seg == SEG_BR else rect[0], rect[1]), (0.0, 0.0)
            tw, th = (rect[2] - rect[0]) * scale, (rect[3] - rect[1]) * scale
            o = place(tw, th)
            if o is None:
                return
            e = np.zeros(1, BORDER_DTYPE)
            e["origin"][0] = o
            e["rect"][0] = (0.0, 0.0, float(np.ceil(tw)), float(np.ceil(th)))
            e["c0"][0], e["c1"][0] = premul(side0), premul(side1)
            e["flags"][0] = seg | (BORDER_STYLE_SOLID << 8) | (BORDER_STYLE_SOLID << 16) | (int(do_aa) << 28)
            e["widths"][0] = (float(np.ceil(wid[0] * scale)), float(np.ceil(wid[1] * scale)))
            e["radii"][0] = (radius[0] * scale, radius[1] * scale)
            e["cp"][0] = ((ho[0] - rect[0]) * scale, (ho[1] - rect[1]) * scale, hr[0] * scale, hr[1] * scale,
                          (vo[0] - rect[0]) * scale, (vo[1] - rect[1]) * scale, vr[0] * scale, vr[1] * scale)
            inst.append(e)

        corner(SEG_TL, tl, cols[0], cols[1], (l, t), rad[0], SEG_TR, SEG_BL, rad[1], rad[3])
        corner(SEG_TR, tr, cols[1], cols[2], (r, t), rad[1], SEG_TL, SEG_BR, rad[0], rad[2])
        corner(SEG_BR, br, cols[2], cols[3], (r, b), rad[2], SEG_BL, SEG_TR, rad[3], rad[1])
        corner(SEG_BL, bl, cols[3], cols[0], (l, b), rad[3], SEG_BR, SEG_TL, rad[2], rad[0])
        # edges (add_edge_segment, border.rs:1181-1260): a solid edge is a 1-texel-long strip of the edge's width, stretched later
        for seg, wid, col, size in ((SEG_LEFT, l, cols[0], (l, 1.0)), (SEG_TOP, t, cols[1], (1.0, t)),
                                    (SEG_RIGHT, r, cols[2], (r, 1.0)), (SEG_BOTTOM, b, cols[3], (1.0, b))):
            if wid <= 0.0:
                continue
            tw, th = size[0] * scale, size[1] * scale
            o = place(tw, th)
            if o is None:
                continue
            e = np.zeros(1, BORDER_DTYPE)
            e["origin"][0] = o
            e["rect"][0] = (0.0, 0.0, float(np.ceil(tw)), float(np.ceil(th)))
            e["c0"][0] = e["c1"][0] = premul(col)
            e["flags"][0] = seg | (BORDER_STYLE_SOLID << 8) | (BORDER_STYLE_SOLID << 16) | (int(do_aa) << 28)
            e["widths"][0] = (float(np.ceil(size[0] * scale)), float(np.ceil(size[1] * scale)))
            inst.append(e)
    tgt.steps.append(Step("cs_border_solid", "BORDER", np.concatenate(inst), "PremultipliedAlpha", "none"))
    frame.passes.append([tgt])
    frame.readback = [t_cache]
    frame.n_border_segments = len(inst)
    return frame


# cs_border_segment: the styled segments -- double / groove / ridge / inset / outset corners and edges, dashed edges
# (CLIP_DASH_EDGE), dotted edges and corner dots (CLIP_DOT), corner dashes (CLIP_DASH_CORNER) -- with the instance encodings of
# border.rs:904-1043 (add_segment) and :330-560 (write_dashed / write_dotted_corner_instances).
BS_NONE, BS_SOLID, BS_DOUBLE, BS_DOTTED, BS_DASHED, BS_HIDDEN, BS_GROOVE, BS_RIDGE, BS_INSET, BS_OUTSET = range(10)
CLIP_NONE, CLIP_DASH_CORNER, CLIP_DASH_EDGE, CLIP_DOT = range(4)


def border_segments(n=60, seed=141, atlas=1024):
    rng = np.random.default_rng(seed)
    frame = Frame(atlas, atlas, (1.0, 1.0, 1.0, 1.0))
    t_cache = TextureRef("border_cache", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, render_target=True)
    tgt = Target(t_cache, "texture_cache", clear_color=(0.0, 0.0, 0.0, 0.0))
    inst = []
    x = y = shelf = 2

    def place(w, h):
        nonlocal x, y, shelf
        w, h = int(np.ceil(w)), int(np.ceil(h))
        if x + w + 2 > atlas:
            x, y, shelf = 2, y + shelf + 2, 0
        if y + h + 2 > atlas:
            return None
        o = (float(x), float(y))
        x += w + 2
        shelf = max(shelf, h)
        return o

    def color():
        c = [float(v) for v in rng.uniform(0, 1, size=3)] + [float(rng.choice([1.0, 1.0, 0.5]))]
        if rng.integers(0, 6) == 0:
            c[:3] = [0.0, 0.0, 0.0]                                   # black: mod_color's special case
        return [c[0] * c[3], c[1] * c[3], c[2] * c[3], c[3]]

    def push(o, size, c0, c1, seg, s0, s1, clip, widths, radii, cp):
        e = np.zeros(1, BORDER_DTYPE)
        e["origin"][0] = o
        e["rect"][0] = (0.0, 0.0, size[0], size[1])
        e["c0"][0], e["c1"][0] = c0, c1
        e["flags"][0] = seg | (s0 << 8) | (s1 << 16) | (clip << 24) | (1 << 28)
        e["widths"][0], e["radii"][0], e["cp"][0] = widths, radii, cp
        inst.append(e)

    corner_styles = [BS_SOLID, BS_DOUBLE, BS_GROOVE, BS_RIDGE, BS_INSET, BS_OUTSET]
    for k in range(n):
        kind = k % 6
        if kind in (0, 1, 2):                                        # corner, no clip
            seg = int(rng.integers(0, 4))
            wx, wy = float(rng.choice([3.0, 6.0, 9.0, 14.0, 21.0])), float(rng.choice([3.0, 6.0, 9.0, 14.0, 21.0]))
            rx, ry = (0.0, 0.0) if k % 5 == 0 else (float(rng.uniform(wx * 0.5, 70)), float(rng.uniform(wy * 0.5, 70)))
            size = (float(np.ceil(max(wx, rx))), float(np.ceil(max(wy, ry))))
            s0 = corner_styles[int(rng.integers(0, len(corner_styles)))]
            s1 = s0 if k % 2 else corner_styles[int(rng.integers(0, len(corner_styles)))]
            o = place(*size)
            if o is None:
                break
            push(o, size, color(), color(), seg, s0, s1, CLIP_NONE, (wx, wy), (rx, ry), (0.0,) * 8)
        elif kind == 3:                                              # corner dash: the band between two lines (point + tangent each)
            seg = int(rng.integers(0, 4))
            w_ = float(rng.choice([6.0, 10.0, 16.0]))
            rad = float(rng.uniform(30, 80))
            size = (float(np.ceil(rad)), float(np.ceil(rad)))
            a0, a1 = sorted(float(v) for v in rng.uniform(0.1, 1.4, size=2))
            ocx, ocy = (1.0 if seg in (1, 2) else 0.0), (1.0 if seg in (2, 3) else 0.0)
            cx, cy = (size[0] if ocx == 0.0 else 0.0), (size[1] if ocy == 0.0 else 0.0)     # ellipse centre: the inner corner
            sgx, sgy = (-1.0 if ocx == 0.0 else 1.0), (-1.0 if ocy == 0.0 else 1.0)
            p0 = (cx + sgx * (rad - w_ / 2) * np.cos(a0), cy + sgy * (rad - w_ / 2) * np.sin(a0))
            p1 = (cx + sgx * (rad - w_ / 2) * np.cos(a1), cy + sgy * (rad - w_ / 2) * np.sin(a1))
            t0 = (-sgx * np.sin(a0), sgy * np.cos(a0))
            t1 = (-sgx * np.sin(a1), sgy * np.cos(a1))
            o = place(*size)
            if o is None:
                break
            push(o, size, color(), color(), seg, BS_DASHED, BS_DASHED, CLIP_DASH_CORNER, (w_, w_), (rad, rad),
                 (float(p0[0]), float(p0[1]), float(t0[0]), float(t0[1]), float(p1[0]), float(p1[1]), float(t1[0]), float(t1[1])))
        elif kind == 4:                                              # dashed / dotted edges
            seg = int(rng.integers(4, 8))
            vertical = seg in (SEG_LEFT, SEG_RIGHT)
            wid = float(rng.choice([2.0, 4.0, 7.0, 12.0]))
            if k % 2:
                ln = float(rng.integers(8, 90))
                size = (wid, ln) if vertical else (ln, wid)
                cp = (0.0, ln * 0.25) if vertical else (ln * 0.25, 0.0)
                clip, st, cps = CLIP_DASH_EDGE, BS_DASHED, cp + (0.0,) * 6
            else:
                size = (wid, 2.0 * wid) if vertical else (2.0 * wid, wid)
                cp = (wid * 0.5, wid, wid * 0.5) if vertical else (wid, wid * 0.5, wid * 0.5)
                clip, st, cps = CLIP_DOT, BS_DOTTED, cp + (0.0,) * 5
            o = place(*size)
            if o is None:
                break
            c = color()
            push(o, size, c, c, seg, st, st, clip, (wid, wid), (0.0, 0.0), cps)
        else:                                                        # styled edges / corner dots
            if k % 2:
                seg = int(rng.integers(4, 8))
                vertical = seg in (SEG_LEFT, SEG_RIGHT)
                wid = float(rng.choice([3.0, 6.0, 9.0, 15.0]))
                size = (wid, float(rng.integers(1, 12))) if vertical else (float(rng.integers(1, 12)), wid)
                st = [BS_DOUBLE, BS_GROOVE, BS_RIDGE, BS_INSET][int(rng.integers(0, 4))]
                o = place(*size)
                if o is None:
                    break
                c = color()
                push(o, size, c, c, seg, st, st, CLIP_NONE, (wid, wid), (0.0, 0.0), (0.0,) * 8)
            else:
                seg = int(rng.integers(0, 4))
                w_ = float(rng.choice([4.0, 8.0, 13.0]))
                rad = float(rng.uniform(24, 70))
                size = (float(np.ceil(rad)), float(np.ceil(rad)))
                dot_r = w_ * 0.5
                dc = (float(rng.uniform(dot_r, size[0] - dot_r)), float(rng.uniform(dot_r, size[1] - dot_r)))
                o = place(*size)
                if o is None:
                    break
                push(o, size, color(), color(), seg, BS_DOTTED, BS_DOTTED, CLIP_DOT, (w_, w_), (rad, rad),
                     (dc[0], dc[1], dot_r if k % 4 else 0.4, 0.0) + (0.0,) * 4)
    tgt.steps.append(Step("cs_border_segment", "BORDER", np.concatenate(inst), "PremultipliedAlpha", "none"))
    frame.passes.append([tgt])
    frame.readback = [t_cache]
    frame.n_border_segments = len(inst)
    return frame


# ---------------------------------------------------------------------------
# cs_line_decoration and cs_fast_linear_gradient: texture-cache tasks (render_target.rs:1184-1190 LineDecorationJob, one
# period of the pattern per task, drawn with premultiplied-alpha blending; prim_store/gradient/linear.rs:689-694
# FastLinearGradientInstance, two-stop axis-aligned gradients drawn unblended).
LINE_DTYPE = np.dtype([("task", "<f4", (4,)), ("local", "<f4", (2,)), ("wavy", "<f4"), ("style", "<i4"), ("axis", "<f4")])
FASTGRAD_DTYPE = np.dtype([("task", "<f4", (4,)), ("c0", "<f4", (4,)), ("c1", "<f4", (4,)), ("axis", "<f4")])
LINE_SOLID, LINE_DOTTED, LINE_DASHED, LINE_WAVY = range(4)


LGRAD_DTYPE = np.dtype([("task", "<f4", (4,)), ("start", "<f4", (2,)), ("end", "<f4", (2,)), ("scale", "<f4", (2,)), ("extend", "<i4"), ("addr", "<i4")])


RGRAD_DTYPE = np.dtype([("task", "<f4", (4,)), ("center", "<f4", (2,)), ("scale", "<f4", (2,)), ("r0", "<f4"), ("r1", "<f4"), ("ratio", "<f4"),
                        ("extend", "<i4"), ("addr", "<i4")])


def cache_decorations(n_lines=60, n_grads=40, n_lgrads=30, n_rgrads=0, n_cgrads=0, seed=151, atlas=1024):
    rng = np.random.default_rng(seed)
    frame = Frame(atlas, atlas, (1.0, 1.0, 1.0, 1.0))
    t_cache = TextureRef("decoration_cache", atlas, atlas, G.GL_RGBA8, G.GL_LINEAR, render_target=True)
    tgt = Target(t_cache, "texture_cache", clear_color=(0.0, 0.0, 0.0, 0.0))
    x = y = shelf = 2

    def place(w, h):
        nonlocal x, y, shelf
        w, h = int(np.ceil(w)), int(np.ceil(h))
        if x + w + 2 > atlas:
            x, y, shelf = 2, y + shelf + 2, 0
        if y + h + 2 > atlas:
            return None
        o = (float(x), float(y))
        x += w + 2
        shelf = max(shelf, h)
        return o

    grads = []
    for k in range(n_grads):
        w, h = float(rng.integers(8, 200)), float(rng.integers(8, 120))
        if k % 5 == 0:
            w, h = w + 0.5, h + 0.25                                   # fractional task rects
        o = place(w, h)
        if o is None:
            break
        e = np.zeros(1, FASTGRAD_DTYPE)
        e["task"][0] = (o[0], o[1], o[0] + w, o[1] + h)
        c0 = [float(v) for v in rng.uniform(0, 1, size=4)]
        c1 = [float(v) for v in rng.uniform(0, 1, size=4)]
        e["c0"][0] = [c0[0] * c0[3], c0[1] * c0[3], c0[2] * c0[3], c0[3]]
        e["c1"][0] = [c1[0] * c1[3], c1[1] * c1[3], c1[2] * c1[3], c1[3]]
        e["axis"][0] = float(k % 2)
        grads.append(e)
    lines = []
    for k in range(n_lines):
        style = (LINE_SOLID, LINE_DOTTED, LINE_DASHED, LINE_WAVY, LINE_WAVY, LINE_DOTTED)[k % 6]
        vertical = k % 4 == 3
        thick = float(rng.choice([1.0, 2.0, 3.0, 5.0, 8.0, 12.5]))
        scale = float(rng.choice([1.0, 1.0, 2.0, 1.5]))
        wavy = 0.0
        # line_dec.rs get_line_decoration_size: one period of the pattern along the line, the line's thickness across
        if style == LINE_DASHED:
            local = (thick * 3.0 * 2.0 if k % 2 else thick * 4.0, thick)
        elif style == LINE_DOTTED:
            local = (thick * 2.0, thick)
        elif style == LINE_WAVY:
            wavy = max(1.0, float(np.floor(thick / 4.0 + 0.5))) if k % 2 else thick / 3.0
            slope = thick - wavy
            flat = max((wavy - 1.0) * 2.0, 1.0)
            local = ((slope + flat) * 2.0, thick)
        else:
            local = (float(rng.integers(8, 64)), thick)
        dev = (local[0] * scale, local[1] * scale)
        tw, th = (dev[1], dev[0]) if vertical else dev
        o = place(tw, th)
        if o is None:
            break
        e = np.zeros(1, LINE_DTYPE)
        e["task"][0] = (o[0], o[1], o[0] + float(np.ceil(tw)), o[1] + float(np.ceil(th)))
        e["local"][0] = (local[1], local[0]) if vertical else local
        e["wavy"][0], e["style"][0], e["axis"][0] = wavy, style, 1.0 if vertical else 0.0
        lines.append(e)
    # cs_linear_gradient (LinearGradientInstance, prim_store/gradient/linear.rs:727-734): multi-stop gradients through the
    # 128-entry table in sGpuBufferF; start / end points in the task's (scaled) pixel space
    lgrads = []
    for k in range(n_lgrads):
        w, h = float(rng.integers(12, 220)), float(rng.integers(8, 120))
        o = place(w, h)
        if o is None:
            break
        nst = int(rng.integers(2, 6))
        offs = [0.0] + sorted(float(v) for v in rng.uniform(0.05, 0.95, size=nst - 2)) + [1.0]
        if nst > 3 and k % 2:
            offs[2] = offs[1]                                        # hard stop
        cols = [tuple(float(v) for v in rng.uniform(0, 1, size=3)) + (float(rng.choice([1.0, 0.7])),) for _ in range(nst)]
        addr = frame.gpu_buffer_f.push(build_gradient_lut(list(zip(offs, cols)), reverse=bool(k & 1)))
        sc = (1.0, 1.0) if k % 3 else (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))
        mode = k % 7
        W, H = w * sc[0], h * sc[1]
        if mode == 0:   sp, ep = (0.0, 0.0), (W, 0.0)
        elif mode == 1: sp, ep = (0.0, 0.0), (0.0, H)
        elif mode == 2: sp, ep = (0.0, 0.0), (W, H)
        elif mode == 3: sp, ep = (W * 0.75, H * 0.5), (W * 0.25, H * 0.4)
        elif mode == 4: sp, ep = (W * 0.3, 0.0), (W * 0.6, 0.0)
        elif mode == 5: sp, ep = (float(rng.uniform(0, W)), float(rng.uniform(0, H))), (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
        else:           sp, ep = (W * 0.5, H * 0.5), (W * 0.5, H * 0.5)      # degenerate line: delta is not finite
        e = np.zeros(1, LGRAD_DTYPE)
        e["task"][0] = (o[0], o[1], o[0] + w, o[1] + h)
        e["start"][0], e["end"][0], e["scale"][0] = sp, ep, sc
        e["extend"][0], e["addr"][0] = (1 if k % 4 == 1 else 0), addr
        lgrads.append(e)
    # cs_radial_gradient (RadialGradientInstance): centre / radii in the task's (scaled) pixel space, elliptical through xy_ratio
    rgrads = []
    for k in range(n_rgrads):
        w, h = float(rng.integers(12, 220)), float(rng.integers(8, 120))
        o = place(w, h)
        if o is None:
            break
        nst = int(rng.integers(2, 6))
        offs = [0.0] + sorted(float(v) for v in rng.uniform(0.05, 0.95, size=nst - 2)) + [1.0]
        if nst > 3 and k % 2:
            offs[2] = offs[1]
        cols = [tuple(float(v) for v in rng.uniform(0, 1, size=3)) + (float(rng.choice([1.0, 0.7])),) for _ in range(nst)]
        addr = frame.gpu_buffer_f.push(build_gradient_lut(list(zip(offs, cols)), reverse=bool(k & 1)))
        sc = (1.0, 1.0) if k % 3 else (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))
        W, H = w * sc[0], h * sc[1]
        mode = k % 6
        if mode == 0:   c, r0, r1 = (W * 0.5, H * 0.5), 0.0, min(W, H) * 0.5
        elif mode == 1: c, r0, r1 = (0.0, 0.0), 0.0, float(np.hypot(W, H))
        elif mode == 2: c, r0, r1 = (W * 0.3, H * 0.7), min(W, H) * 0.1, min(W, H) * 0.4
        elif mode == 3: c, r0, r1 = (float(rng.uniform(-W, 2 * W)), float(rng.uniform(-H, 2 * H))), float(rng.uniform(0, 30)), float(rng.uniform(40, 200))
        elif mode == 4: c, r0, r1 = (W * 0.5, H * 0.5), 10.0, (10.0 if RADIAL_DEGENERATE else 14.5)   # zero-length radius range: radius_scale = 0
        else:           c, r0, r1 = (W * 0.5, -H), H * 0.9, H * 2.5                      # every row's span misses the centre row
        e = np.zeros(1, RGRAD_DTYPE)
        e["task"][0] = (o[0], o[1], o[0] + w, o[1] + h)
        e["center"][0], e["scale"][0] = c, sc
        e["r0"][0], e["r1"][0], e["ratio"][0] = r0, r1, (1.0 if k % 4 else float(rng.uniform(0.4, 2.5)))
        e["extend"][0], e["addr"][0] = (1 if k % 4 == 1 else 0), addr
        rgrads.append(e)
    # cs_conic_gradient (ConicGradientInstance): same instance layout as the radial one with (start offset, end offset, angle)
    cgrads = []
    for k in range(n_cgrads):
        w, h = float(rng.integers(12, 220)), float(rng.integers(8, 120))
        o = place(w, h)
        if o is None:
            break
        nst = int(rng.integers(2, 6))
        offs = [0.0] + sorted(float(v) for v in rng.uniform(0.05, 0.95, size=nst - 2)) + [1.0]
        cols = [tuple(float(v) for v in rng.uniform(0, 1, size=3)) + (float(rng.choice([1.0, 0.7])),) for _ in range(nst)]
        addr = frame.gpu_buffer_f.push(build_gradient_lut(list(zip(offs, cols)), reverse=bool(k & 1)))
        sc = (1.0, 1.0) if k % 3 else (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))
        W, H = w * sc[0], h * sc[1]
        e = np.zeros(1, RGRAD_DTYPE)
        e["task"][0] = (o[0], o[1], o[0] + w, o[1] + h)
        e["center"][0] = (W * 0.5, H * 0.5) if k % 2 else (float(rng.uniform(-W * 0.2, W * 1.2)), float(rng.uniform(-H * 0.2, H * 1.2)))
        e["scale"][0] = sc
        e["r0"][0], e["r1"][0] = (0.0, 1.0) if k % 3 else (float(rng.uniform(0.0, 0.3)), float(rng.uniform(0.5, 1.0)))
        e["ratio"][0] = float(rng.uniform(0.0, 2.0 * np.pi)) if k % 4 else 0.0          # the angle
        e["extend"][0], e["addr"][0] = (1 if k % 4 == 1 else 0), addr
        cgrads.append(e)
    tgt.steps.append(Step("cs_fast_linear_gradient", "FAST_LINEAR_GRADIENT", np.concatenate(grads), None, "none"))
    if cgrads:
        tgt.steps.append(Step("cs_conic_gradient", "CONIC_GRADIENT", np.concatenate(cgrads), None, "none"))
    if rgrads:
        tgt.steps.append(Step("cs_radial_gradient", "RADIAL_GRADIENT", np.concatenate(rgrads), None, "none"))
    if lgrads:
        tgt.steps.append(Step("cs_linear_gradient", "LINEAR_GRADIENT", np.concatenate(lgrads), None, "none"))
    tgt.steps.append(Step("cs_line_decoration", "LINE", np.concatenate(lines), "PremultipliedAlpha", "none"))
    frame.passes.append([tgt])
    frame.readback = [t_cache]
    frame.n_tasks = (len(grads), len(lines), len(lgrads), len(rgrads), len(cgrads))
    return frame


# ---------------------------------------------------------------------------
# ps_quad_radial_gradient / ps_quad_conic_gradient: gradient patterns on the quad path (quad.rs pattern kinds; the
# QuadHeader's pattern_input = (address of the two gradient blocks in sGpuBufferF, address of the 128-entry stop table)).
# Radial gradients with start radius == end radius (radius_scale = 0: every position maps to offset 0) are part of the parity
# scenes.  The reference's SSE2 build evaluates them through fastSqrt<false>(0) = 0 * rsqrt(0) = NaN (swgl_ext.h:1607-1617, 1789):
# tests/test_clang_budget.py turns them off for its family sweep and pins that behaviour in a case of its own.
RADIAL_DEGENERATE = True


def quad_gradients(width=1024, height=1024, n=60, seed=181, tile_filter=None, only=None, rotate=False, perspective=False):
    rng, rects = random_rects(n, width, height, 24, 380, seed, True)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    prims = []
    for i in range(n):
        x0, y0, x1, y1 = [float(v) for v in rects[i]]
        w, h = x1 - x0, y1 - y0
        nst = int(rng.integers(2, 6))
        offs = [0.0] + sorted(float(v) for v in rng.uniform(0.05, 0.95, size=nst - 2)) + [1.0]
        if nst > 3 and i % 2:
            offs[2] = offs[1]
        cols = [tuple(float(v) for v in rng.uniform(0, 1, size=3)) + (float(rng.choice([1.0, 1.0, 0.6])),) for _ in range(nst)]
        table = frame.gpu_buffer_f.push(build_gradient_lut(list(zip(offs, cols)), reverse=bool(i & 1)))
        repeat = 1.0 if i % 4 == 1 else 0.0
        conic = i % 3 == 2
        if conic:
            c = (w * 0.5, h * 0.5) if i % 2 else (float(rng.uniform(0, w)), float(rng.uniform(0, h)))
            s0, s1 = (0.0, 1.0) if i % 5 else (float(rng.uniform(0.0, 0.3)), float(rng.uniform(0.5, 1.0)))
            blocks = [[c[0], c[1], 1.0, 1.0], [s0, s1, float(rng.uniform(0, 2 * np.pi)) if i % 4 else 0.0, repeat]]
        else:
            mode = i % 5
            if mode == 0:   c, r0, r1 = (w * 0.5, h * 0.5), 0.0, min(w, h) * 0.5
            elif mode == 1: c, r0, r1 = (0.0, 0.0), 0.0, float(np.hypot(w, h))
            elif mode == 2: c, r0, r1 = (w * 0.3, h * 0.7), min(w, h) * 0.1, min(w, h) * 0.4
            elif mode == 3: c, r0, r1 = (float(rng.uniform(-w, 2 * w)), float(rng.uniform(-h, 2 * h))), float(rng.uniform(0, 30)), float(rng.uniform(40, 200))
            else:           c, r0, r1 = (w * 0.5, h * 0.5), 10.0, (10.0 if RADIAL_DEGENERATE else 14.5)
            ratio = 1.0 if i % 4 else float(rng.uniform(0.4, 2.5))
            blocks = [[c[0], c[1], 1.0, 1.0], [r0, r1, ratio, repeat]]
        params = frame.gpu_buffer_f.push(blocks)
        tid = rotation_about(frame, rng, (x0 + x1) / 2, (y0 + y1) / 2, i, float(np.hypot(x1 - x0, y1 - y0)) * 0.5 if perspective else None) if (rotate and i % 3 != 1) else 0
        bb = rotated_bounds(tuple(float(v) for v in rects[i])) if tid else tuple(rects[i])
        if tid and perspective:
            rr = float(np.hypot(x1 - x0, y1 - y0)) * 1.4 + 4
            bb = ((x0 + x1) / 2 - rr, (y0 + y1) / 2 - rr, (x0 + x1) / 2 + rr, (y0 + y1) / 2 + rr)
        prims.append((rects[i], conic, params, table, tid, bb))
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        for zi, (rect, conic, params, table, tid, bb) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            if not (bb[0] < x1 and bb[2] > x0 and bb[1] < y1 and bb[3] > y0):
                continue
            big = (-BIG, -BIG, BIG, BIG)
            if tid:
                q = frame.quad_instance(rect, big, (1.0, 1.0, 1.0, 1.0), zi + 1, task, transform_id=tid, quad_flags=0, edge_flags=15,
                                        pattern_input=(params, table))
            else:
                q = frame.quad_instance(rect, big, (1.0, 1.0, 1.0, 1.0), zi + 1, task, pattern_input=(params, table))
            target.alpha.append(Step("ps_quad_conic_gradient" if conic else "ps_quad_radial_gradient", "PRIM_INSTANCES",
                                     np.array([q], dtype=np.int32), "PremultipliedAlpha", "alpha", textures={}))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


# ---------------------------------------------------------------------------
# brush_yuv_image: video frames as YUV planes (batch.rs:2301-2390 YuvImage prims; shade.rs brush_yuv_image).  One R8 texture per
# plane for YUV_FORMAT_PLANAR (chroma at half resolution, 4:2:0), an R8 luma + an RG8 interleaved chroma texture for
# YUV_FORMAT_NV12.  The brush's gpu-cache block is YuvImageData (prim_store/image.rs write_prim_gpu_blocks): [channel bit depth,
# YuvRangedColorSpace, YuvFormat, 0]; prim_user_data = the planes' ImageSource addresses.
YUV_FORMAT_NV12, YUV_FORMAT_PLANAR = 0, 3


def yuv_grid(width=1024, height=1024, n=60, seed=301, tile_filter=None, only=None, nearest=False, hdr=False, planar=True):
    """hdr: 10-bit video -- three R16 planes holding the low 10 bits (YUV_FORMAT_PLANAR, channel bit depth 10: the span shader
    rescales by 6 bits) and R16 + RG16 planes with the sample in the high bits (YUV_FORMAT_P010)."""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (1.0, 1.0, 1.0, 1.0))
    A = 1024
    ypl, upl, vpl = np.zeros((A, A), np.uint8), np.zeros((A // 2, A // 2), np.uint8), np.zeros((A // 2, A // 2), np.uint8)
    videos = []
    x = y = shelf = 0
    for i in range(12):
        w, h = 2 * int(rng.integers(16, 120)), 2 * int(rng.integers(16, 90))
        if x + w > A:
            x, y, shelf = 0, y + shelf, 0
        yy, xx = np.mgrid[0:h, 0:w]
        ypl[y:y + h, x:x + w] = ((xx * 255 // max(w - 1, 1)) ^ rng.integers(0, 256, size=(h, w))).astype(np.uint8) if i % 3 else rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        upl[y // 2:(y + h) // 2, x // 2:(x + w) // 2] = rng.integers(0, 256, size=(h // 2, w // 2), dtype=np.uint8)
        vpl[y // 2:(y + h) // 2, x // 2:(x + w) // 2] = rng.integers(0, 256, size=(h // 2, w // 2), dtype=np.uint8)
        ry = frame.gpu_cache.push([[x, y, x + w, y + h], [0.0, 0.0, 0.0, 0.0]])
        rc = frame.gpu_cache.push([[x // 2, y // 2, (x + w) // 2, (y + h) // 2], [0.0, 0.0, 0.0, 0.0]])
        videos.append((w, h, ry, rc))
        x += w
        shelf = max(shelf, h)
    filt = G.GL_NEAREST if nearest else G.GL_LINEAR
    if hdr:
        # 10-bit samples: the 8-bit pattern extended by two random low bits
        ext = lambda p: (p.astype(np.uint16) << 2) | rng.integers(0, 4, size=p.shape).astype(np.uint16)
        y10, u10, v10 = ext(ypl), ext(upl), ext(vpl)
        US = G.GL_UNSIGNED_SHORT
        t_y = TextureRef("yuv10_plane_y", A, A, G.GL_R16, filt, pixels=y10, upload_format=G.GL_RED, upload_type=US)
        t_u = TextureRef("yuv10_plane_u", A // 2, A // 2, G.GL_R16, filt, pixels=u10, upload_format=G.GL_RED, upload_type=US)
        t_v = TextureRef("yuv10_plane_v", A // 2, A // 2, G.GL_R16, filt, pixels=v10, upload_format=G.GL_RED, upload_type=US)
        t_y_msb = TextureRef("p010_plane_y", A, A, G.GL_R16, filt, pixels=(y10 << 6).astype(np.uint16), upload_format=G.GL_RED, upload_type=US)
        t_uv = TextureRef("p010_plane_uv", A // 2, A // 2, G.GL_RG16, filt, pixels=np.ascontiguousarray(np.stack([u10 << 6, v10 << 6], axis=2).astype(np.uint16)),
                          upload_format=G.GL_RG, upload_type=US)
        frame.static_textures += [t_y, t_u, t_v, t_y_msb, t_uv]
    else:
        t_y = TextureRef("yuv_plane_y", A, A, G.GL_R8, filt, pixels=ypl, upload_format=G.GL_RED)
        t_u = TextureRef("yuv_plane_u", A // 2, A // 2, G.GL_R8, filt, pixels=upl, upload_format=G.GL_RED)
        t_v = TextureRef("yuv_plane_v", A // 2, A // 2, G.GL_R8, filt, pixels=vpl, upload_format=G.GL_RED)
        t_y_msb = t_y
        t_uv = TextureRef("yuv_plane_uv", A // 2, A // 2, G.GL_RG8, filt, pixels=np.ascontiguousarray(np.stack([upl, vpl], axis=2)), upload_format=G.GL_RG)
        frame.static_textures += [t_y, t_u, t_v, t_uv]
    depth, fmt_semi = (10.0, 1.0) if hdr else (8.0, float(YUV_FORMAT_NV12))      # (YUV_FORMAT_P010 = 1)
    prims = []         # (rect, brush data address, user data, opaque pass, nv12)
    band, gx, k = 200, 4.0, 0
    while True:        # opaque-pass videos on a disjoint grid in the top band (see image_grid)
        vw, vh, ry, rc = videos[k % len(videos)]
        sc = (1.0, 1.0, 1.4, 0.5, 0.81)[k % 5]
        w, h = vw * sc, min(vh * sc, band - 8.0)
        if gx + w + 4 > width:
            break
        off = 0.37 if k % 5 == 1 else 0.0
        semi = k % 2 == 1 or not planar          # (planar=False: two-plane frames only -- NV12 / P010)
        spec = frame.gpu_cache.push([[depth, float(k % 7), fmt_semi if semi else float(YUV_FORMAT_PLANAR), 0.0]])
        prims.append(((gx + off, 4.0 + off, gx + off + w, 4.0 + off + h), spec, (ry, rc, rc, 0), True, semi))
        gx += float(np.ceil(w)) + 6.0
        k += 1
    for k in range(n):
        vw, vh, ry, rc = videos[int(rng.integers(0, len(videos)))]
        mode = k % 5
        if mode == 0:
            w, h, px, py = float(vw), float(vh), float(rng.integers(-20, width - 20)), float(rng.integers(band, height - 20))
        elif mode == 1:
            w, h, px, py = float(vw), float(vh), float(rng.uniform(0, width - vw)), float(rng.uniform(band, height - vh))
        elif mode == 2:
            sc = float(rng.uniform(1.1, 3.0))
            w, h = vw * sc, vh * sc
            px, py = float(rng.integers(0, width)) - w / 2, max(float(rng.integers(band + 300, height + 100)) - h / 2, float(band))
        elif mode == 3:
            w, h, px, py = vw * 0.5, vh * 0.5, float(rng.integers(0, width - vw)), float(rng.integers(band, height - vh))
        else:
            w, h = vw * float(rng.uniform(0.3, 1.7)), vh * float(rng.uniform(0.3, 1.7))
            px, py = float(rng.uniform(0, width - w)), float(rng.uniform(band, height - h))
        semi = (k // 2) % 2 == 1 or not planar
        spec = frame.gpu_cache.push([[depth, float((k * 3 + 1) % 7), fmt_semi if semi else float(YUV_FORMAT_PLANAR), 0.0]])
        prims.append(((px, py, px + w, py + h), spec, (ry, rc, rc, 0), False, semi))
    targets = []
    for (tx, ty, ox, oy) in tile_grid(width, height):
        if tile_filter is not None and not tile_filter(tx, ty):
            continue
        x0, y0, x1, y1 = ox, oy, ox + TILE_W, oy + TILE_H
        tex = TextureRef(f"tile_{tx}_{ty}", TILE_W, TILE_H, G.GL_RGBA8, G.GL_LINEAR, render_target=True, with_depth=True)
        target = Target(tex, "picture_tile", clear_color=(1.0, 1.0, 1.0, 1.0), clear_depth=True)
        task = frame.add_render_task((0.0, 0.0, float(TILE_W), float(TILE_H)), 1.0, (float(ox), float(oy)))
        batches = {}       # (opaque pass, nv12) -> instances; alpha-pass batches break wherever the plane textures change
        order = []
        for zi, (rect, spec, ud, opaque, nv12) in enumerate(prims):
            if only is not None and zi not in only:
                continue
            if not (rect[0] < x1 and rect[2] > x0 and rect[1] < y1 and rect[3] > y0):
                continue
            ph = frame.add_prim_header(rect, (-BIG, -BIG, BIG, BIG), zi + 1, spec, 0, task, ud)
            inst = frame.brush_instance(ph, CLIP_TASK_EMPTY)
            if opaque:
                batches.setdefault((True, nv12), []).append(inst)
            else:
                if not order or order[-1][0] != nv12:
                    order.append((nv12, []))
                order[-1][1].append(inst)
        tex_of = lambda nv12: {0: t_y_msb, 1: t_uv} if nv12 else {0: t_y, 1: t_u, 2: t_v}
        for nv12 in (False, True):
            op = batches.get((True, nv12))
            if op:
                target.opaque.append(Step("brush_yuv_image TEXTURE_2D,YUV", "PRIM_INSTANCES", np.array(op[::-1], dtype=np.int32), None, "opaque",
                                          textures=tex_of(nv12)))
        for nv12, al in order:
            target.alpha.append(Step("brush_yuv_image ALPHA_PASS,TEXTURE_2D,YUV", "PRIM_INSTANCES", np.array(al, dtype=np.int32),
                                     "PremultipliedAlpha", "alpha", textures=tex_of(nv12)))
        targets.append(target)
        rect = (float(x0), float(y0), float(x1), float(y1))
        clip = (float(x0), float(y0), float(min(x1, width)), float(min(y1, height)))
        frame.composite_tiles.append(CompositeTile(tex, rect, clip, opaque=True))
    frame.passes.append(targets)
    return frame


def yuv_composites(width=1024, height=768, seed=311, nearest=False, planar=True):
    """Video surfaces composited straight into the window ("composite TEXTURE_2D,YUV": composite.rs ExternalSurfaceDependency::Yuv,
    renderer/mod.rs:3335-3420): planar and NV12 frames, every colour space, 1:1 / scaled / clipped / flipped, opaque and
    (premultiplied-alpha blended) on top of a picture-cache tile."""
    rng = np.random.default_rng(seed)
    frame = Frame(width, height, (0.2, 0.3, 0.4, 1.0))
    filt = G.GL_NEAREST if nearest else G.GL_LINEAR
    vids = []
    for i in range(4):
        w, h = 2 * int(rng.integers(60, 140)), 2 * int(rng.integers(40, 100))
        yy, xx = np.mgrid[0:h, 0:w]
        ypl = ((xx * 255 // (w - 1)) ^ rng.integers(0, 64, size=(h, w))).astype(np.uint8)
        upl, vpl = rng.integers(0, 256, size=(h // 2, w // 2), dtype=np.uint8), rng.integers(0, 256, size=(h // 2, w // 2), dtype=np.uint8)
        t_y = TextureRef(f"video{i}_y", w, h, G.GL_R8, filt, pixels=ypl, upload_format=G.GL_RED)
        if i % 2 == 0 and planar:
            planes = [t_y, TextureRef(f"video{i}_u", w // 2, h // 2, G.GL_R8, filt, pixels=upl, upload_format=G.GL_RED),
                      TextureRef(f"video{i}_v", w // 2, h // 2, G.GL_R8, filt, pixels=vpl, upload_format=G.GL_RED)]
            fmt = YUV_FORMAT_PLANAR
        else:
            planes = [t_y, TextureRef(f"video{i}_uv", w // 2, h // 2, G.GL_RG8, filt, pixels=np.ascontiguousarray(np.stack([upl, vpl], axis=2)),
                                      upload_format=G.GL_RG)]
            fmt = YUV_FORMAT_NV12
        frame.static_textures += planes
        vids.append((w, h, planes, fmt))
    cases = [   # (video, x, y, scale x, scale y, flip, sub-rect of the frame in luma texels or None)
        (0, 10.0, 10.0, 1.0, 1.0, (0.0, 0.0), None), (1, 300.25, 14.5, 1.0, 1.0, (0.0, 0.0), None), (2, 560.0, 8.0, 1.6, 1.6, (0.0, 0.0), None),
        (3, 20.0, 250.0, 0.5, 0.5, (0.0, 0.0), None), (0, 200.0, 260.0, 0.73, 1.31, (0.0, 1.0), None), (1, 520.5, 300.0, 2.2, 0.9, (1.0, 0.0), None),
        (2, -40.0, 520.0, 1.0, 1.0, (0.0, 0.0), None), (3, 330.0, 520.0, 1.5, 1.5, (0.0, 0.0), (16.0, 8.0, 96.0, 72.0)),
    ]
    for k, (vi, x, y, sx, sy, flip, sub) in enumerate(cases):
        w, h, planes, fmt = vids[vi]
        sr = sub or (0.0, 0.0, float(w), float(h))
        dw, dh = (sr[2] - sr[0]) * sx, (sr[3] - sr[1]) * sy
        rect = (x, y, x + dw, y + dh)
        clip = (max(x, 0.0) + (7.0 if k % 3 == 2 else 0.0), max(y, 0.0), min(x + dw, float(width)), min(y + dh, float(height)) - (5.0 if k % 3 == 1 else 0.0))
        uvs = [sr] + [(sr[0] / 2, sr[1] / 2, sr[2] / 2, sr[3] / 2)] * (len(planes) - 1)
        yuv = dict(planes=planes, uv_rects=uvs, color_space=k % 7, format=fmt, depth=8)
        frame.composite_tiles.append(CompositeTile(planes[0], rect, clip, opaque=(k % 2 == 0), flip=flip, yuv=yuv))
    frame.passes.append([])
    return frame


# ---------------------------------------------------------------------------
# cs_svg_filter / cs_svg_filter_node: the nodes of CSS / SVG filter chains and graphs, drawn into colour targets with blending
# off (renderer/mod.rs:3628-3640).  Instances as render_target.rs:901-1170 builds them; task data as render_task.rs:805-810
# (user_data = [opacity] / [offset.x, offset.y]) and :843-905 (extra GPU-cache blocks: colour matrix 5, flood / drop-shadow
# colour 1, arithmetic k 1, component-transfer tables).
SVG_FILTER_DTYPE = np.dtype([("a", "<i4", (3,)), ("k", "<u2", (4,)), ("e", "<u2", (2,))])                     # SvgFilterInstance, gpu_types.rs:148-158
SVG_NODE_DTYPE = np.dtype([("t", "<f4", (4,)), ("s1", "<f4", (4,)), ("s2", "<f4", (4,)), ("a", "<i4", (2,)),
                           ("k", "<u2", (2,)), ("e", "<u2", (2,))])                                            # SVGFEFilterInstance, gpu_types.rs:160-170
SVGF_BLEND, SVGF_FLOOD, SVGF_LINEAR_TO_SRGB, SVGF_SRGB_TO_LINEAR, SVGF_OPACITY, SVGF_COLOR_MATRIX = 0, 1, 2, 3, 4, 5
SVGF_DROP_SHADOW, SVGF_OFFSET, SVGF_COMPONENT_TRANSFER, SVGF_IDENTITY, SVGF_COMPOSITE = 6, 7, 8, 9, 10
SVG_INVALID_ADDRESS = 0xFFFF      # GpuCacheAddress::INVALID (u16::MAX, u16::MAX)


def _svg_picture(rng, w, h, k):
    """A premultiplied RGBA8 picture with the alpha patterns a filter input has: opaque, holes, soft ramps, zero rows"""
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    if k % 5 == 0:
        img[..., 3] = 255
    elif k % 5 == 1:
        img[..., 3] = np.where((xx // 5 + yy // 4) % 4 == 0, 0, img[..., 3])
    elif k % 5 == 2:
        img[..., 3] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
    elif k % 5 == 3:
        img[::6, :, 3] = 0
        img[..., 0] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
    img[..., :3] = (img[..., :3].astype(np.uint16) * img[..., 3:4] // 255).astype(np.uint8)
    return img


def svg_filters(node=False, seed=401, atlas=1024, window=(256, 256), nearest=False, only=None, chained=True):
    """A colour target full of filter tasks (swatches of 56 x 40), every kind the program has: `node` False = cs_svg_filter (blend
    x 16 modes + an unknown one, flood, both sRGB conversions, opacity, colour matrix, drop shadow, offset, component transfer
    with every function type, identity, composite x 7 operators + an unknown one), True = cs_svg_filter_node (every FILTER_*
    value with a case in main(), both colour spaces, plus kinds main() has no case for).  Inputs are two static premultiplied
    atlases (as earlier passes' colour targets would be), sampled 1:1, scaled and at fractional offsets; with `chained` a second
    target runs filters on the first target's output.  frame.readback lists the targets."""
    rng = np.random.default_rng(seed)
    frame = Frame(window[0], window[1], (1.0, 1.0, 1.0, 1.0))
    filt = G.GL_NEAREST if nearest else G.GL_LINEAR
    p1, p2 = np.zeros((atlas, atlas, 4), np.uint8), np.zeros((atlas, atlas, 4), np.uint8)
    t_in1 = TextureRef("svg_input_1", atlas, atlas, G.GL_RGBA8, filt, pixels=p1, upload_format=G.GL_BGRA)
    t_in2 = TextureRef("svg_input_2", atlas, atlas, G.GL_RGBA8, filt, pixels=p2, upload_format=G.GL_BGRA)
    frame.static_textures += [t_in1, t_in2]
    sw, sh = 56, 40
    per_row = (atlas - 8) // (sw + 6)

    def cache_address(addr):
        return (addr % 1024, addr // 1024)       # GpuCacheAddress { u, v }: MAX_VERTEX_TEXTURE_WIDTH texels per row

    def color_matrix_blocks():
        m = rng.uniform(-0.5, 1.2, size=(4, 4)).astype(np.float32)
        off = rng.uniform(-0.2, 0.3, size=4).astype(np.float32)
        return cache_address(frame.gpu_cache.push([list(r) for r in m] + [list(off)]))

    specs = []
    if not node:
        specs += [(SVGF_BLEND, m) for m in list(range(16)) + [23]]
        specs += [(SVGF_FLOOD, 0)] * 2 + [(SVGF_LINEAR_TO_SRGB, 0)] * 3 + [(SVGF_SRGB_TO_LINEAR, 0)] * 3 + [(SVGF_OPACITY, 0)] * 3
        specs += [(SVGF_COLOR_MATRIX, 0)] * 3 + [(SVGF_DROP_SHADOW, 0)] * 3 + [(SVGF_OFFSET, 0)] * 5 + [(SVGF_IDENTITY, 0)] * 3
        specs += [(SVGF_COMPONENT_TRANSFER, j) for j in range(8)]
        specs += [(SVGF_COMPOSITE, op) for op in list(range(7)) + [9]] + [(SVGF_COMPOSITE, 6)] + [(13, 0)]
    else:
        have = [0, 2, 4] + list(range(6, 38, 2)) + [38, 40, 42, 44, 46, 48, 50, 52, 54, 70, 72, 92]
        for kd in have:
            specs.append((kd, 0))
            if kd not in (2, 4):
                specs.append((kd + 1, 0))
        specs += [(40, 1), (41, 1), (38, 1), (39, 1), (56, 0), (75, 0), (80, 0), (101, 0), (0, 1), (1, 1), (28, 1), (52, 1)]
    if only is not None:
        specs = [specs[i] for i in only]

    def build_pass(name, in1, in2, in1_rects, in2_rects, seed_shift):
        """one colour target: swatch k reads rect in1_rects[k] of in1 (and in2_rects[k] of in2)"""
        tex = TextureRef(name, atlas, atlas, G.GL_RGBA8, filt, render_target=True)
        tgt = Target(tex, "color", clear_color=(0.0, 0.0, 0.0, 0.0))
        out_rects = []
        inst = np.zeros(len(specs), SVG_NODE_DTYPE if node else SVG_FILTER_DTYPE)
        for k, (kind, sub) in enumerate(specs):
            gx, gy = k % per_row, k // per_row
            x0, y0 = 5 + gx * (sw + 6), 7 + gy * (sh + 6)
            trect = (float(x0), float(y0), float(x0 + sw), float(y0 + sh))
            out_rects.append(trect)
            r1, r2 = in1_rects[k % len(in1_rects)], in2_rects[k % len(in2_rects)]
            a1, a2 = frame.add_render_task(r1), frame.add_render_task(r2)
            extra = (SVG_INVALID_ADDRESS, SVG_INVALID_ADDRESS)
            if not node:
                ud = (0.0, 0.0, 0.0)
                count = 0 if kind == SVGF_FLOOD else (2 if kind in (SVGF_BLEND, SVGF_DROP_SHADOW, SVGF_COMPOSITE) else 1)
                generic = 0
                if kind == SVGF_BLEND or kind == SVGF_COMPOSITE:
                    generic = sub
                    if kind == SVGF_COMPOSITE and sub == 6:
                        kv = rng.uniform(-0.5, 1.0, size=4).astype(np.float32)
                        extra = cache_address(frame.gpu_cache.push([list(kv)]))
                elif kind in (SVGF_FLOOD, SVGF_DROP_SHADOW):
                    c = [rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.3, 1.0)]
                    extra = cache_address(frame.gpu_cache.push([c]))
                elif kind == SVGF_OPACITY:
                    ud = (float(rng.uniform(0.1, 1.0)), 0.0, 0.0)
                elif kind == SVGF_COLOR_MATRIX:
                    extra = color_matrix_blocks()
                elif kind == SVGF_OFFSET:
                    ud = (float(rng.integers(-12, 13)) + (0.5 if k % 2 else 0.0), float(rng.integers(-9, 10)) + (0.25 if k % 3 == 0 else 0.0), 0.0)
                elif kind == SVGF_COMPONENT_TRANSFER:
                    funcs = [int(v) for v in rng.integers(0, 5, size=4)] if sub >= 2 else ([CT_TABLE, CT_DISCRETE, CT_LINEAR, CT_GAMMA] if sub else [CT_GAMMA, CT_IDENTITY, CT_TABLE, CT_LINEAR])
                    blocks = []
                    for f in funcs:
                        if f in (CT_TABLE, CT_DISCRETE):
                            lut = rng.uniform(-0.1, 1.1, size=256).astype(np.float32)
                            blocks += [list(lut[4 * j:4 * j + 4]) for j in range(64)]
                        elif f == CT_LINEAR:
                            blocks.append([rng.uniform(-1.5, 2.0), rng.uniform(-0.3, 0.5), 0.0, 0.0])
                        elif f == CT_GAMMA:
                            blocks.append([rng.uniform(0.5, 1.5), float(rng.choice([0.4, 1.0, 2.2, 3.0])), rng.uniform(-0.1, 0.2), 0.0])
                    if blocks:
                        extra = cache_address(frame.gpu_cache.push(blocks))
                    generic = funcs[0] << 12 | funcs[1] << 8 | funcs[2] << 4 | funcs[3]
                a_t = frame.add_render_task(trect, ud[0], (ud[1], ud[2]))
                inst["a"][k] = (a_t, a1 if count > 0 else 0, a2 if count > 1 else 0)
                inst["k"][k] = (kind, count, generic, 0)
                inst["e"][k] = extra
            else:
                count = 0 if kind in (72, 73) else (2 if (6 <= kind <= 37 or 42 <= kind <= 55 or kind in (70, 71)) else 1)
                # (compute_uv: (task.p0 + offset + scale * aPosition) / texture size -- 1:1 is scale = the target size, offset = inflate)
                def scale_offset(r, j):
                    zoom = (1.0, 1.0, 0.5, 1.25, 1.0)[(k + j) % 5]
                    off = ((0.0, 0.0), (1.0, 2.0), (0.5, 0.25), (-3.0, 4.0), (6.5, -2.0))[(k + 2 * j + sub) % 5]
                    return [sw * zoom, sh * zoom, off[0], off[1]]
                s1, s2 = scale_offset(r1, 0), scale_offset(r2, 1)
                if kind in (2, 3):
                    s2 = [float(rng.uniform(0.1, 1.0)), 0.0, 0.0, 0.0]
                elif kind in (72, 73):
                    s2 = [rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.3, 1.0)]
                elif kind in (80, 81, 82, 83):
                    s2 = [2.0, 3.0, 0.0, 0.0]
                elif kind in (38, 39):
                    extra = color_matrix_blocks()
                elif kind in (40, 41):
                    lut = rng.uniform(-0.1, 1.1, size=(256, 4)).astype(np.float32)
                    if sub:
                        lut = np.clip(lut, 0.0, 1.0)
                    # (a table never straddles a data-texture row: gpu_cache.rs allocates whole rows to 256-block requests)
                    extra = cache_address(frame.gpu_cache.push([list(r) for r in lut]))
                elif kind in (42, 43):
                    extra = cache_address(frame.gpu_cache.push([list(rng.uniform(-0.5, 1.0, size=4).astype(np.float32))]))
                elif kind in (70, 71):
                    extra = cache_address(frame.gpu_cache.push([[rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.3, 1.0)]]))
                inst["t"][k] = trect
                inst["s1"][k], inst["s2"][k] = s1, s2
                inst["a"][k] = (a1 if count > 0 else 0x7FFFFFFF, a2 if count > 1 else 0x7FFFFFFF)
                inst["k"][k] = (kind, count)
                inst["e"][k] = extra
        # one batch per (input 1, input 2) texture pair -- here one
        tgt.steps.append(Step("cs_svg_filter_node" if node else "cs_svg_filter", "SVG_FILTER_NODE" if node else "SVG_FILTER", inst, None, "none",
                              textures={0: in1, 1: in2}))
        return tex, tgt, out_rects

    # input pictures: one per swatch slot, some larger / smaller than the swatch (scaled sampling), some at fractional task rects
    rects1, rects2 = [], []
    for k in range(len(specs)):
        gx, gy = k % per_row, k // per_row
        for j, (pix, rects) in enumerate(((p1, rects1), (p2, rects2))):
            w, h = ((sw, sh), (sw, sh), (sw + 17, sh + 9), (sw // 2, sh // 2), (sw, sh))[(k + j) % 5]
            x0, y0 = 3 + gx * (sw + 24), 4 + gy * (sh + 14) + 5 * j
            if x0 + w + 1 >= atlas or y0 + h + 1 >= atlas:
                x0, y0 = 3, 4
            pix[y0:y0 + h, x0:x0 + w] = _svg_picture(rng, w, h, k + 3 * j)
            frac = 0.5 if (k + j) % 7 == 3 else 0.0
            rects.append((float(x0), float(y0), float(x0 + w) + frac, float(y0 + h) + frac))
    t_in1.pixels, t_in2.pixels = p1, p2
    tex_a, tgt_a, out_a = build_pass("svg_pass_a", t_in1, t_in2, rects1, rects2, 0)
    frame.passes.append([tgt_a])
    frame.readback = [tex_a]
    if chained:
        tex_b, tgt_b, _ = build_pass("svg_pass_b", tex_a, t_in2, out_a[::-1], rects2[3:] + rects2[:3], 1)
        frame.passes.append([tgt_b])
        frame.readback.append(tex_b)
    return frame


# The TEXTURE_RECT keys (shader_features.rs:136-138, 181-198): what images, video planes and compositor surfaces backed by rectangle
# textures (macOS IOSurfaces, external images) are drawn with -- the same batches, `sampler2DRect` samplers bound through
# GL_TEXTURE_RECTANGLE, texel uv left unnormalised.  The frame data is the same (uv rects are texel rects either way), only the
# program key and the texture target change, so any scene of those four families has a rectangle-texture twin.
RECT_FAMILIES = ("brush_image", "brush_yuv_image", "composite", "cs_scale")


def texture_rect(frame, composites=True):
    n = 0
    for targets in frame.passes:
        for target in targets:
            for step in list(target.opaque) + list(target.alpha) + list(target.steps):
                if step.shader.split(" ")[0] in RECT_FAMILIES and "TEXTURE_2D" in step.shader:
                    step.shader = step.shader.replace("TEXTURE_2D", "TEXTURE_RECT")
                    n += 1
    frame.texture_rect = bool(composites)
    assert n > 0 or composites
    return frame
