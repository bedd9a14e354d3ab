#!/usr/bin/env python3
"""Generates webrender_amd/wrench/glyphs.npz: glyph bitmaps of the reference's own reftest fonts (wrench/reftests/text/*.ttf),
rasterised with FreeType exactly as WebRender's glyph rasteriser does on Linux -- a ctypes restatement of
wr_glyph_rasterizer/src/platform/unix/font.rs (load_glyph :417-596, get_bounding_box :619-658, rasterize_glyph_outline :790-838,
rasterize_glyph :884-1097) for FontRenderMode::Alpha with the platform's default options (FontHinting::LCD, which for an
alpha-mode instance leaves FT_LOAD_DEFAULT; FT_LOAD_NO_BITMAP | FT_LOAD_IGNORE_GLOBAL_ADVANCE_WIDTH; identity font transform;
FontInstanceFlags::SUBPIXEL_POSITION: four quarter-pixel x offsets; no texture padding in screen raster space).  There is no
gamma preblend on this platform (the unix backend has none).  INPUT DATA of the text workloads and parity cases, not code: the GPU
box has neither /root/reference nor these fonts, so the bitmaps travel as this fixture.

    python3 tests/golden/make_glyphs.py        (in a container that has /root/reference and libfreetype.so.6)
"""
import ctypes as C
import json
import os
import re
import numpy as np
import yaml

REF = "/root/reference/wrench/reftests/text"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "webrender_amd", "wrench", "glyphs.npz")
FT = C.CDLL("libfreetype.so.6")
L = C.c_long


class Vector(C.Structure): _fields_ = [("x", L), ("y", L)]
class BBox(C.Structure): _fields_ = [("xMin", L), ("yMin", L), ("xMax", L), ("yMax", L)]
class Generic(C.Structure): _fields_ = [("data", C.c_void_p), ("finalizer", C.c_void_p)]
class Metrics(C.Structure): _fields_ = [(n, L) for n in ("width", "height", "horiBearingX", "horiBearingY", "horiAdvance", "vertBearingX", "vertBearingY", "vertAdvance")]
class Bitmap(C.Structure):
    _fields_ = [("rows", C.c_uint), ("width", C.c_uint), ("pitch", C.c_int), ("buffer", C.POINTER(C.c_ubyte)), ("num_grays", C.c_ushort),
                ("pixel_mode", C.c_ubyte), ("palette_mode", C.c_ubyte), ("palette", C.c_void_p)]
class Outline(C.Structure):
    _fields_ = [("n_contours", C.c_short), ("n_points", C.c_short), ("points", C.c_void_p), ("tags", C.c_void_p), ("contours", C.c_void_p), ("flags", C.c_int)]
class GlyphSlot(C.Structure):
    _fields_ = [("library", C.c_void_p), ("face", C.c_void_p), ("next", C.c_void_p), ("glyph_index", C.c_uint), ("generic", Generic),
                ("metrics", Metrics), ("linearHoriAdvance", L), ("linearVertAdvance", L), ("advance", Vector), ("format", C.c_int),
                ("bitmap", Bitmap), ("bitmap_left", C.c_int), ("bitmap_top", C.c_int), ("outline", Outline)]
class Face(C.Structure):
    _fields_ = [("num_faces", L), ("face_index", L), ("face_flags", L), ("style_flags", L), ("num_glyphs", L), ("family_name", C.c_char_p),
                ("style_name", C.c_char_p), ("num_fixed_sizes", C.c_int), ("available_sizes", C.c_void_p), ("num_charmaps", C.c_int),
                ("charmaps", C.c_void_p), ("generic", Generic), ("bbox", BBox), ("units_per_EM", C.c_ushort), ("ascender", C.c_short),
                ("descender", C.c_short), ("height", C.c_short), ("max_advance_width", C.c_short), ("max_advance_height", C.c_short),
                ("underline_position", C.c_short), ("underline_thickness", C.c_short), ("glyph", C.POINTER(GlyphSlot))]


FT_LOAD_DEFAULT, FT_LOAD_NO_BITMAP, FT_LOAD_IGNORE_GLOBAL_ADVANCE_WIDTH = 0, 1 << 3, 1 << 9
FT_GLYPH_FORMAT_OUTLINE = (ord("o") << 24) | (ord("u") << 16) | (ord("t") << 8) | ord("l")
lib = C.c_void_p()
assert FT.FT_Init_FreeType(C.byref(lib)) == 0
FT.FT_Outline_Get_CBox.argtypes = [C.POINTER(Outline), C.POINTER(BBox)]
FT.FT_Outline_Translate.argtypes = [C.POINTER(Outline), L, L]
FT.FT_Set_Char_Size.argtypes = [C.c_void_p, L, L, C.c_uint, C.c_uint]
FT.FT_Load_Glyph.argtypes = [C.c_void_p, C.c_uint, C.c_int32]
FT.FT_Render_Glyph.argtypes = [C.POINTER(GlyphSlot), C.c_int]
FT.FT_Get_Char_Index.argtypes = [C.c_void_p, C.c_ulong]
FT.FT_Get_Char_Index.restype = C.c_uint
FT.FT_Set_Transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]


TEXT_REFTESTS = ("text", "long-text", "negative-pos", "non-opaque", "snap-text-offset", "1658", "shadow-cover-1",
                 # round 6
                 "1658-ref", "non-opaque-notref", "shadow", "shadow-ref", "shadow-single", "shadow-cover-2", "shadow-many", "shadow-complex",
                 "two-shadows", "subtle-shadow", "subtle-shadow-ref", "snap-clip", "snap-clip-ref", "subpixel-translate-ref",
                 "shadow-partial-glyph", "shadow-partial-glyph-ref", "allow-subpixel-ref", "diacritics", "diacritics-ref", "transparent-no-aa",
                 "transparent-no-aa-ref", "subpx-bg-mask-ref", "colors", "decorations-ref", "ahem-ref", "shadow-clip-ref", "blank",
                 "decorations", "shadow-atomic", "shadow-atomic-ref", "shadow-ordering", "shadow-ordering-ref", "shadow-clip-rect",
                 "blurred-shadow-local-clip-rect", "subpixel-translate", "snap-text-offset-ref", "shadow-clip", "shadow-fast-clip", "shadow-fast-clip-ref")


def open_face(name):
    face = C.c_void_p()
    assert FT.FT_New_Face(lib, os.path.join(REF, name).encode(), 0, C.byref(face)) == 0, name
    return face


def rasterize(face, size_px, glyph_index, subpx):
    """-> (left, top, bitmap[h, w] u8, advance px) or None for a glyph without pixels (space).  left / top as RasterizedGlyph:
    the bitmap's top-left corner is at (x + left, baseline - top)."""
    fr = C.cast(face, C.POINTER(Face)).contents
    FT.FT_Set_Transform(face, None, None)                        # (identity shape, zero delta)
    req = int(size_px * 1.0 * 64.0 + 0.5)
    assert FT.FT_Set_Char_Size(face, req, req, 0, 0) == 0
    flags = FT_LOAD_DEFAULT | FT_LOAD_NO_BITMAP | FT_LOAD_IGNORE_GLOBAL_ADVANCE_WIDTH
    assert FT.FT_Load_Glyph(face, glyph_index, flags) == 0
    slot = fr.glyph.contents
    assert slot.format == FT_GLYPH_FORMAT_OUTLINE
    advance = slot.metrics.horiAdvance / 64.0
    cbox = BBox()
    FT.FT_Outline_Get_CBox(C.byref(slot.outline), C.byref(cbox))
    if slot.outline.n_contours == 0:
        return None, advance
    dx = int(subpx * 0.25 / 1.0 * 64.0 + 0.5)                     # get_subpx_offset: quarter pixels along x; dy = 0
    dy = -int(0.0 * 64.0 + 0.5)
    bx0, by0, bx1, by1 = cbox.xMin + dx, cbox.yMin + dy, cbox.xMax + dx, cbox.yMax + dy
    bx0 &= ~63; by0 &= ~63; bx1 = (bx1 + 63) & ~63; by1 = (by1 + 63) & ~63
    left, top, width, height = bx0 >> 6, by1 >> 6, (bx1 - bx0) >> 6, (by1 - by0) >> 6
    if width == 0 or height == 0:
        return None, advance
    FT.FT_Outline_Translate(C.byref(slot.outline), dx - ((cbox.xMin + dx) & ~63), dy - ((cbox.yMin + dy) & ~63))
    assert FT.FT_Render_Glyph(fr.glyph, 0) == 0                 # FT_RENDER_MODE_NORMAL
    bm = slot.bitmap
    assert bm.pixel_mode == 2                                   # FT_PIXEL_MODE_GRAY
    rows, w = bm.rows, bm.width
    px = np.zeros((rows, w), np.uint8)
    for r in range(rows):
        px[r] = np.ctypeslib.as_array(bm.buffer, shape=(abs(bm.pitch) * rows,))[r * bm.pitch:r * bm.pitch + w]
    left += slot.bitmap_left
    top += slot.bitmap_top - height
    return (left, top, px), advance


FT_LOAD_TARGET_LCD = (3 & 15) << 16
FT_RENDER_MODE_LCD = 3
FT_PIXEL_MODE_LCD = 5


def rasterize_lcd(face, size_px, glyph_index, subpx):
    """FontRenderMode::Subpixel (font.rs: load_glyph :448-458 FT_LOAD_TARGET_LCD under the default FontHinting::LCD; pad_bounding_box
    :596-613 one pixel either side for the LCD filter; rasterize_glyph_outline :790-838 FT_RENDER_MODE_LCD; rasterize_glyph :957-971
    [b, g, r, max(b, g, r)] per pixel, RGB subpixel order; the library's LCD filter set to FT_LCD_FILTER_DEFAULT :194).
    -> (left, top, bgra[h, w, 4] u8) or None for a glyph without pixels."""
    fr = C.cast(face, C.POINTER(Face)).contents
    FT.FT_Set_Transform(face, None, None)
    req = int(size_px * 1.0 * 64.0 + 0.5)
    assert FT.FT_Set_Char_Size(face, req, req, 0, 0) == 0
    flags = FT_LOAD_TARGET_LCD | FT_LOAD_NO_BITMAP | FT_LOAD_IGNORE_GLOBAL_ADVANCE_WIDTH
    assert FT.FT_Load_Glyph(face, glyph_index, flags) == 0
    slot = fr.glyph.contents
    assert slot.format == FT_GLYPH_FORMAT_OUTLINE
    if slot.outline.n_contours == 0:
        return None
    cbox = BBox()
    FT.FT_Outline_Get_CBox(C.byref(slot.outline), C.byref(cbox))
    cbox.xMin -= 64; cbox.xMax += 64                              # pad_bounding_box
    dx = int(subpx * 0.25 / 1.0 * 64.0 + 0.5)
    dy = -int(0.0 * 64.0 + 0.5)
    bx0, by0, bx1, by1 = cbox.xMin + dx, cbox.yMin + dy, cbox.xMax + dx, cbox.yMax + dy
    bx0 &= ~63; by0 &= ~63; bx1 = (bx1 + 63) & ~63; by1 = (by1 + 63) & ~63
    left, top, width, height = bx0 >> 6, by1 >> 6, (bx1 - bx0) >> 6, (by1 - by0) >> 6
    if width == 0 or height == 0:
        return None
    FT.FT_Outline_Translate(C.byref(slot.outline), dx - ((cbox.xMin + dx) & ~63), dy - ((cbox.yMin + dy) & ~63))
    assert FT.FT_Render_Glyph(fr.glyph, FT_RENDER_MODE_LCD) == 0
    bm = slot.bitmap
    assert bm.pixel_mode == FT_PIXEL_MODE_LCD and bm.width % 3 == 0
    rows, w = bm.rows, bm.width // 3
    px = np.zeros((rows, w, 4), np.uint8)
    if rows and w:
        raw = np.ctypeslib.as_array(bm.buffer, shape=(abs(bm.pitch) * rows,))
        for r in range(rows):
            rgb = raw[r * bm.pitch:r * bm.pitch + 3 * w].reshape(w, 3)
            px[r, :, 0], px[r, :, 1], px[r, :, 2] = rgb[:, 2], rgb[:, 1], rgb[:, 0]
            px[r, :, 3] = rgb.max(axis=1)
    left += slot.bitmap_left
    top += slot.bitmap_top - height
    return left, top, px


def main():
    try:
        FT.FT_Library_SetLcdFilter.argtypes = [C.c_void_p, C.c_int]
        print("FT_Library_SetLcdFilter(FT_LCD_FILTER_DEFAULT) ->", FT.FT_Library_SetLcdFilter(lib, 1), "(7: this FreeType build has no filter of its own and renders LCD by its default method)")
    except AttributeError:
        pass
    bitmaps, index = [], {}            # index["font|size|glyph|subpx"] = [slot in the blob list or -1, left, top, w, h, advance]
    def add(font, face, size, gid, subpxs):
        for s in subpxs:
            key = f"{font}|{size}|{gid}|{s}"
            if key in index:
                continue
            g, adv = rasterize(face, size, gid, s)
            if g is None:
                index[key] = [-1, 0, 0, 0, 0, adv]
            else:
                index[key] = [len(bitmaps), g[0], g[1], g[2].shape[1], g[2].shape[0], adv]
                bitmaps.append(g[2])
    faces = {f: open_face(f) for f in ("VeraBd.ttf", "FreeSans.ttf", "Ahem.ttf")}
    charmap = {}
    # (1) the display lists of the text reftests (wrench/reftests/text/reftest.list) that need no transform, clip chain or line decoration:
    # text runs with explicit glyph lists, and `text:` strings laid out as wrench lays them out (wrench.rs:320-382 layout_simple_ascii:
    # glyph indices of the characters the font has, the pen advanced by get_glyph_dimensions' advance -- font.rs:659-700, whole-pixel
    # variant -- or by size / 3 for a glyph without pixels), rects, shadows.  Sizes are points: yaml_helper.rs:267-269 as_pt_to_f32,
    # size * 16 / 12 pixels (rounds 5 and earlier rasterised at the yaml's number).
    runs = {}
    for name in TEXT_REFTESTS:
        doc = yaml.safe_load(open(os.path.join(REF, name + ".yaml"))) or {}
        out = []
        def walk(items, origin, xlate=(0.0, 0.0)):
            for it in items or []:
                if it.get("type") == "stacking-context":
                    b = it.get("bounds", [0, 0, 0, 0])
                    t = xlate
                    if "transform" in it:               # translate(x, y) only: a reference frame whose transform moves its content by a fraction
                        m = re.fullmatch(r"\s*translate\(\s*([-0-9.]+)\s*,\s*([-0-9.]+)\s*\)\s*", str(it["transform"]))
                        assert m, it["transform"]
                        t = (xlate[0] + float(m.group(1)), xlate[1] + float(m.group(2)))
                    walk(it.get("items", []), (origin[0] + b[0], origin[1] + b[1]), t)
                elif "glyphs" in it or "text" in it:
                    font = it.get("font", "VeraBd.ttf")
                    size_px = float(it.get("size", 12.0)) * 16.0 / 12.0                 # handle_text: default 16 px
                    e = dict(it); e["font"] = font; e["origin_offset"] = list(origin); e["size_px"] = size_px; e["translate"] = list(xlate)
                    o = it.get("origin", [0.0, 0.0])
                    if isinstance(o, str):
                        o = [float(v) for v in o.replace(",", " ").split()]
                    if "glyphs" in it:
                        gids = [int(g) for g in it["glyphs"]]
                        offs = [float(v) for v in it["offsets"]]
                        e["offsets"] = [offs[i] + float(o[i & 1]) for i in range(len(offs))]
                    else:
                        gids, offs = [], []
                        cx, cy = float(o[0]), float(o[1])
                        for ch in it["text"]:
                            gid = FT.FT_Get_Char_Index(faces[font], ord(ch))
                            if gid == 0:
                                continue                                  # (get_glyph_indices: None for a character the font lacks)
                            g, adv = rasterize(faces[font], size_px, int(gid), 0)
                            gids.append(int(gid)); offs += [cx, cy]
                            cx += np.float32(adv) if g is not None else np.float32(size_px / 3.0)
                            cx = float(np.float32(cx))
                        e["glyphs"], e["offsets"] = gids, offs
                        del e["text"]
                    out.append(e)
                    for gid in set(gids):
                        add(font, faces[font], size_px, int(gid), range(4))
                else:
                    out.append(dict(it, origin_offset=list(origin), translate=list(xlate)))
        walk((doc.get("root") or {}).get("items"), (0.0, 0.0))
        runs[name] = out
    # (2) the character sets of cfg3 / text-rendering: FreeSans, sizes 8 .. 24, printable ASCII, whole-pixel variant
    for size in range(8, 25):
        for ch in range(32, 127):
            gid = FT.FT_Get_Char_Index(faces["FreeSans.ttf"], ch)
            charmap[f"FreeSans.ttf|{ch}"] = int(gid)
            add("FreeSans.ttf", faces["FreeSans.ttf"], float(size), int(gid), range(4) if size in (12, 16) else (0,))
    # (3) wrench/benchmarks/text-rendering.yaml and overlapping-text-shadows.yaml: their sizes are points too (8 .. 20 pt = 10.67 .. 26.67 px,
    # 60 pt = 80 px), whole-pixel variant
    bench = os.path.join(REF, "..", "..", "benchmarks")
    for yml in ("text-rendering.yaml", "overlapping-text-shadows.yaml"):
        for it in yaml.safe_load(open(os.path.join(bench, yml)))["root"]["items"]:
            if "text" in it:
                px = float(it.get("size", 12.0)) * 16.0 / 12.0
                for ch in sorted(set(it["text"])):
                    gid = FT.FT_Get_Char_Index(faces["FreeSans.ttf"], ord(ch))
                    charmap[f"FreeSans.ttf|{ord(ch)}"] = int(gid)
                    add("FreeSans.ttf", faces["FreeSans.ttf"], px, int(gid), (0,))
    # (4) the LCD (FontRenderMode::Subpixel) bitmaps of cfg3's character set: the BGRA atlas of its subpixel runs
    lcd, lcd_index = [], {}
    for size in range(8, 25):
        for ch in range(33, 127):
            gid = charmap[f"FreeSans.ttf|{ch}"]
            g = rasterize_lcd(faces["FreeSans.ttf"], float(size), int(gid), 0)
            key = f"FreeSans.ttf|{float(size)}|{gid}|0"
            if g is None:
                lcd_index[key] = [-1, 0, 0, 0, 0]
            else:
                lcd_index[key] = [len(lcd), g[0], g[1], g[2].shape[1], g[2].shape[0]]
                lcd.append(g[2])
    lcd_blob = np.concatenate([b.reshape(-1) for b in lcd])
    lcd_offs = np.cumsum([0] + [b.size for b in lcd]).astype(np.int64)
    blob = np.concatenate([b.reshape(-1) for b in bitmaps]) if bitmaps else np.zeros(0, np.uint8)
    offs = np.cumsum([0] + [b.size for b in bitmaps]).astype(np.int64)
    np.savez_compressed(OUT, blob=blob, offsets=offs, index=np.frombuffer(json.dumps(index).encode(), np.uint8),
                        charmap=np.frombuffer(json.dumps(charmap).encode(), np.uint8),
                        runs=np.frombuffer(json.dumps(runs).encode(), np.uint8),
                        lcd_blob=lcd_blob, lcd_offsets=lcd_offs, lcd_index=np.frombuffer(json.dumps(lcd_index).encode(), np.uint8))
    print("wrote", OUT, len(bitmaps), "bitmaps,", blob.size, "bytes of coverage,", os.path.getsize(OUT), "bytes on disk")


if __name__ == "__main__":
    main()
