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


# ---------------------------------------------------------------------------
# brush_blend (webrender/res/blend.glsl:90-237, brush_blend.glsl:91-120): an independent float32
# restatement for 1:1 sampled swatches (scenes.filter_swatches), including swgl's own pow()
# approximation (glsl.h:776-799, portable roundfast) and the premultiplied-alpha blend over the
# tile's white clear (blend.h:474).
F = np.float32


def _approx_log2(x):
    b = x.view(np.uint32)
    e = b.astype(np.float32) * F(1.0 / (1 << 23))
    m = ((b & np.uint32(0x007fffff)) | np.uint32(0x3f000000)).view(np.float32)
    return e - F(124.225514990) - F(1.498030302) * m - F(1.725879990) / (F(0.3520887068) + m)


def _glsl_floor(v):
    rt = v.astype(np.int32).astype(np.float32)
    return rt - (rt > v).astype(np.float32)


def _approx_pow2(x):
    f = x - _glsl_floor(x)
    t = x + F(121.274057500) - F(1.490129070) * f + F(27.728023300) / (F(4.84252568) - f)
    return (F(1 << 23) * t + F(0.5)).astype(np.int32).view(np.float32)


def glsl_pow(x, y):
    x = np.ascontiguousarray(x, np.float32)
    with np.errstate(all="ignore"):
        r = _approx_pow2(_approx_log2(x) * F(y))
    return np.where((x == 0) | (x == 1), x, r).astype(np.float32)


def _clamp01(v):
    return np.minimum(np.maximum(v, F(0.0)), F(1.0))


def filter_swatch(img, op, params):
    """img: premultiplied RGBA u8 [h, w, 4] -> RGBA u8 of the filtered swatch blended over opaque white."""
    ud = params["user_data"]
    amount = F(ud) / F(65536.0)
    Cs = img.astype(np.float32) * F(1.0 / 255.0)
    alpha = Cs[..., 3].copy()
    with np.errstate(all="ignore"):
        col = np.where((alpha != 0)[..., None], Cs[..., :3] / alpha[..., None], Cs[..., :3]).astype(np.float32)
    lumR, lumG, lumB = F(0.2126), F(0.7152), F(0.0722)
    oR, oG, oB = F(1.0) - lumR, F(1.0) - lumG, F(1.0) - lumB
    inv = F(1.0) - amount
    mat = off = None
    if op == 0:
        col = _clamp01(col * amount - F(0.5) * amount + F(0.5))
    elif op == 3:
        col = ((F(1.0) - col) - col) * amount + col
    elif op == 6:
        col = _clamp01(col * amount)
    elif op == 8:
        c1 = col / F(12.92)
        c2 = glsl_pow(col / F(1.055) + F(F(0.055) / F(1.055)), F(2.4))
        col = np.where(col <= F(0.04045), c1, c2).astype(np.float32)
    elif op == 9:
        c1 = col * F(12.92)
        c2 = F(1.055) * glsl_pow(col, F(F(1.0) / F(2.4))) - F(0.055)
        col = np.where(col <= F(0.0031308), c1, c2).astype(np.float32)
    elif op == 10:
        c = params["flood"]
        col = np.broadcast_to(c[:3], col.shape).astype(np.float32)
        alpha = np.full_like(alpha, c[3])
    elif op == 11:
        ch = [col[..., 0], col[..., 1], col[..., 2], alpha]
        for i, (fn, tab) in enumerate(zip(params["funcs"], params["tables"])):
            if fn in (1, 2):
                k = _glsl_floor(ch[i] * F(255.0) + F(0.5)).astype(np.int32)
                ch[i] = _clamp01(tab[np.clip(k, 0, 255)])
            elif fn == 3:
                ch[i] = _clamp01(tab[0] * ch[i] + tab[1])
            elif fn == 4:
                ch[i] = _clamp01(tab[0] * glsl_pow(ch[i], tab[1]) + tab[2])
        col = np.stack(ch[:3], axis=-1).astype(np.float32)
        alpha = ch[3].astype(np.float32)
    else:
        if op == 1:
            mat = [[lumR + oR * inv, lumR - lumR * inv, lumR - lumR * inv, F(0)],
                   [lumG - lumG * inv, lumG + oG * inv, lumG - lumG * inv, F(0)],
                   [lumB - lumB * inv, lumB - lumB * inv, lumB + oB * inv, F(0)], [F(0), F(0), F(0), F(1)]]
        elif op == 2:
            c, s = F(np.cos(amount)), F(np.sin(amount))
            mat = [[lumR + oR * c - lumR * s, lumR - lumR * c + F(0.143) * s, lumR - lumR * c - oR * s, F(0)],
                   [lumG - lumG * c - lumG * s, lumG + oG * c + F(0.140) * s, lumG - lumG * c + lumG * s, F(0)],
                   [lumB - lumB * c + oB * s, lumB - lumB * c - F(0.283) * s, lumB + oB * c + lumB * s, F(0)], [F(0), F(0), F(0), F(1)]]
        elif op == 4:
            mat = [[inv * lumR + amount, inv * lumR, inv * lumR, F(0)], [inv * lumG, inv * lumG + amount, inv * lumG, F(0)],
                   [inv * lumB, inv * lumB, inv * lumB + amount, F(0)], [F(0), F(0), F(0), F(1)]]
        elif op == 5:
            mat = [[F(0.393) + F(0.607) * inv, F(0.349) - F(0.349) * inv, F(0.272) - F(0.272) * inv, F(0)],
                   [F(0.769) - F(0.769) * inv, F(0.686) + F(0.314) * inv, F(0.534) - F(0.534) * inv, F(0)],
                   [F(0.189) - F(0.189) * inv, F(0.168) - F(0.168) * inv, F(0.131) + F(0.869) * inv, F(0)], [F(0), F(0), F(0), F(1)]]
        else:
            mat = [list(r) for r in params["matrix"]]
        off = params["offset"] if op == 7 else np.zeros(4, np.float32)
        v = [col[..., 0], col[..., 1], col[..., 2], alpha]
        out = []
        for i in range(4):      # mat4 * vec4: columns mat[k], left-to-right sums (glsl.h:2582-2598)
            out.append(_clamp01((F(mat[0][i]) * v[0] + F(mat[1][i]) * v[1] + F(mat[2][i]) * v[2] + F(mat[3][i]) * v[3]) + F(off[i])))
        col = np.stack(out[:3], axis=-1).astype(np.float32)
        alpha = out[3].astype(np.float32)
    frag = np.concatenate([alpha[..., None] * col, (alpha * F(1.0))[..., None]], axis=-1).astype(np.float32)
    src = (frag * F(255.0) + F(0.5)).astype(np.int32).astype(np.uint16).astype(np.uint32)   # round_pixel -> u16 lanes
    dst = np.full_like(src, 255)
    a = src[..., 3:4]
    md = ((dst * a + dst) & 0xFFFF) >> 8                       # muldiv255(dst, alphas(src))
    r = (src + dst - md) & 0xFFFF
    r = np.where(r > 255, np.where(r >> 15, 0, 255), r)        # saturating pack (texture.h:14-21)
    return r.astype(np.uint8)
