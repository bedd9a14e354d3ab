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
                     clear_color=(1.0, 1.0, 1.0, 1.0), tile_filter=None):
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
        hit = np.nonzero((rects[:, 0] < x1) & (rects[:, 2] > x0) &
                         (rects[:, 1] < y1) & (rects[:, 3] > y0))[0]
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
                                           quad_flags=qf)
            else:
                addr = frame.gpu_cache.push([list(c)])
                ph = frame.add_prim_header(r, (-BIG, -BIG, BIG, BIG), int(z_ids[i]), addr, 0,
                                           task, (65535, 0, 0, 0))
                inst = frame.brush_instance(ph, CLIP_TASK_EMPTY)
            (op_inst if opaque[i] else al_inst).append(inst)
        if encoding == "quad":
            sh_op = sh_al = "ps_quad_textured"
        else:
            sh_op, sh_al = "brush_solid", "brush_solid ALPHA_PASS"
        if op_inst:
            target.opaque.append(Step(sh_op, "PRIM_INSTANCES",
                                      np.array(op_inst, dtype=np.int32), None, "opaque"))
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
    rgba = np.tile(np.array([[0, 128, 0, 255]], np.uint8), (14, 1))
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


SCENES = {
    "cfg1": cfg1_solid_colors,
    "simple_batching": simple_batching,
    "cfg2": cfg2_overlapping_rects,
    "cfg5": cfg5_many_rects,
}
