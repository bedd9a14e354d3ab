"""ORACLE / TEST INFRASTRUCTURE -- numpy restatement of the reference's solid
rectangle path, independent of both the swgl build in oracle/_ref and the HIP
kernels.  Used by tests only (never by the product path) to pin the locally
built swgl oracle and to provide an oracle of last resort when oracle/_ref is
missing.

What is restated (all from swgl/src, servo/webrender @ 2024-12-20):
  span/row selection of an axis-aligned quad   rasterize.h:783-1055, 521-529
       rows/cols = [floor(v0 + 0.5), floor(v1 + 0.5)) after clipping
  colour packing  round_pixel = int(v*255 + 0.5)   glsl.h:732-744, blend.h:41-45
  premultiplied-alpha blend (GL_ONE, GL_ONE_MINUS_SRC_ALPHA):
       dst' = src + dst - ((dst*src.a + dst) >> 8)  blend.h:473-474, 126-128
  opaque pass: depth LEQUAL + write, ids increasing with paint order
       rasterize.h:41-257 ; renderer/mod.rs:2823-2865
  composite: 1:1 nearest copy of tiles into the window, y flipped by the
       window projection (renderer/mod.rs:4861-4866); ReadPixels returns
       framebuffer rows bottom-up (reftest.rs:306-319)
Parity pin: tests/test_oracle.py checks this model == oracle/_ref on seeded
scenes, and both against tests/golden/*.json.
"""
import numpy as np


def _round_half_up(v):
    return np.floor(np.float32(v) + np.float32(0.5)).astype(np.int64)


def render_rects(width, height, rects, colors, opaque, clear=(1.0, 1.0, 1.0, 1.0)):
    """rects [N,4] float32 device px, colors [N,4] premultiplied float32 RGBA,
    opaque [N] bool.  Returns uint8 [H,W,4] RGBA as ReadPixels would (rows
    bottom-up)."""
    rects = np.asarray(rects, np.float32)
    colors = np.asarray(colors, np.float32)

    def pack(c):  # RGBA float -> u8
        return (np.float32(c) * np.float32(255.0) + np.float32(0.5)).astype(np.int64)

    img = np.empty((height, width, 4), dtype=np.int64)
    img[:] = pack(np.array(clear, np.float32))
    depth = np.full((height, width), -1, dtype=np.int64)  # larger id = nearer
    n = len(rects)
    # opaque pass (order irrelevant thanks to the depth test); z id = index + 1
    for i in range(n):
        if not opaque[i]:
            continue
        x0, y0, x1, y1 = rects[i]
        cx0, cx1 = _round_half_up(np.clip(x0, 0, width)), _round_half_up(np.clip(x1, 0, width))
        cy0, cy1 = _round_half_up(np.clip(y0, 0, height)), _round_half_up(np.clip(y1, 0, height))
        if cx1 <= cx0 or cy1 <= cy0:
            continue
        sub = depth[cy0:cy1, cx0:cx1]
        m = sub <= i
        img[cy0:cy1, cx0:cx1][m] = pack(colors[i])
        sub[m] = i
    # alpha pass, painter's order, depth-tested against opaque prims in front
    for i in range(n):
        if opaque[i]:
            continue
        x0, y0, x1, y1 = rects[i]
        cx0, cx1 = _round_half_up(np.clip(x0, 0, width)), _round_half_up(np.clip(x1, 0, width))
        cy0, cy1 = _round_half_up(np.clip(y0, 0, height)), _round_half_up(np.clip(y1, 0, height))
        if cx1 <= cx0 or cy1 <= cy0:
            continue
        src = pack(colors[i])
        dst = img[cy0:cy1, cx0:cx1]
        m = depth[cy0:cy1, cx0:cx1] <= i
        out = src + dst - ((dst * src[3] + dst) >> 8)
        out = np.clip(out, 0, 255)
        dst[m] = out[m]
    return img[::-1].astype(np.uint8).copy()
