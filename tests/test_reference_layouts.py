"""Independent pin of the INPUT side (VERDICT r2, next #3a): the instance / vertex layouts this repository encodes by hand
(webrender_amd/device.py descriptors, scenes.py / frame.py instance encoders, the program table in csrc/wrhip.hip) against
what the reference's RUST sources declare -- extracted mechanically by tests/golden/make_layouts.py into
tests/golden/reference_layouts.json, so a misreading of gpu_types.rs / vertex.rs / quad.rs shared by the scene generator
and the kernels' vertex stage fails here instead of passing parity silently.

When /root/reference is present (the authoring container) the committed fixture is re-derived and must be identical."""
import importlib.util
import json
import os
import re
import subprocess
import sys
import numpy as np
import pytest
from conftest import ROOT
from webrender_amd import device, scenes
from webrender_amd.frame import Frame

FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_layouts.json")))
REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "webrender", "src")), reason="reference tree not present")
def test_fixture_is_what_the_reference_tree_says():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "golden", "make_layouts.py"), REFERENCE], text=True)
    assert json.loads(out) == FIX


# ---- renderer/vertex.rs desc::* <-> webrender_amd/device.py DESC ---------------------------------------------------------
@pytest.mark.parametrize("name", sorted(device.DESC))
def test_vertex_descriptors_match_vertex_rs(name):
    want = FIX["descriptors"][name]
    got = device.DESC[name]
    assert [list(a) for a in got.vertex_attributes] == want["vertex"], want["at"]
    assert [list(a) for a in got.instance_attributes] == want["instance"], want["at"]


# ---- sizes of the #[repr(C)] instance structs: Rust struct == Rust descriptor stride == the numpy dtype the scenes write ---
_SIZES = {"f32": 4, "i32": 4, "u32": 4, "u16": 2, "u8": 1,
          # newtypes (render_task.rs:52, gpu_types.rs:30, :736, gpu_cache.rs:166-169) and euclid pairs / boxes of f32
          "RenderTaskAddress": 4, "ZBufferId": 4, "TransformPaletteId": 4, "GpuCacheAddress": 4,
          "DeviceSize": 8, "DevicePoint": 8, "LayoutPoint": 8, "LayoutSize": 8, "DeviceVector2D": 8,
          "DeviceRect": 16, "LayoutRect": 16, "TexelRect": 16, "PremultipliedColorF": 16, "(f32, f32)": 8}
_KIND = {"F32": 4, "I32": 4, "U16": 2, "U8Norm": 1, "U16Norm": 2}


def _sizeof(ty):
    m = re.fullmatch(r"\[(.+); (\d+)\]", ty)
    if m:
        return _sizeof(m.group(1)) * int(m.group(2))
    if ty in _SIZES:
        return _SIZES[ty]
    assert ty in FIX["structs"], f"no size known for Rust type {ty!r}"
    return sum(_sizeof(t) for _, t in FIX["structs"][ty]["fields"])      # (repr(C) structs of 4-byte scalars: no padding)


def _stride(desc):
    return sum(c * _KIND[k] for _, c, k in FIX["descriptors"][desc]["instance"])


_STRUCT_DESC_DTYPE = [
    ("PrimitiveInstanceData", "PRIM_INSTANCES", None),
    ("BlurInstance", "BLUR", "BLUR_DTYPE"),
    ("ScalingInstance", "SCALE", "SCALE_DTYPE"),
    ("BorderInstance", "BORDER", "BORDER_DTYPE"),
    ("ClipMaskInstanceRect", "CLIP_RECT", "CLIP_RECT_DTYPE"),
    ("ClipMaskInstanceBoxShadow", "CLIP_BOX_SHADOW", "BOX_SHADOW_DTYPE"),
    ("CompositeInstance", "COMPOSITE", None),
    ("ClearInstance", "CLEAR", None),
    ("MaskInstance", "MASK", None),
    ("CopyInstance", "COPY", "COPY_DTYPE"),
    ("SvgFilterInstance", "SVG_FILTER", "SVG_FILTER_DTYPE"),
    ("SVGFEFilterInstance", "SVG_FILTER_NODE", "SVG_NODE_DTYPE"),
]


@pytest.mark.parametrize("struct,desc,dtype", _STRUCT_DESC_DTYPE, ids=[s for s, _, _ in _STRUCT_DESC_DTYPE])
def test_instance_struct_sizes_agree(struct, desc, dtype):
    n = _sizeof(struct)
    assert n == _stride(desc), (struct, FIX["structs"][struct]["at"], desc, FIX["descriptors"][desc]["at"])
    assert n == device.DESC[desc].instance_stride()
    if dtype:
        assert getattr(scenes, dtype).itemsize == n, dtype


def test_composite_instance_layout():
    """CompositeInstance (gpu_types.rs): rect, clip_rect, colour, [padding, uv type, yuv format, bit depth], three uv rects, flip."""
    names = [f for f, _ in FIX["structs"]["CompositeInstance"]["fields"]]
    assert names == ["rect", "clip_rect", "color", "_padding", "color_space_or_uv_type", "yuv_format", "yuv_channel_bit_depth", "uv_rects", "flip"]
    inst = Frame.composite_instance((1, 2, 3, 4), (5, 6, 7, 8), color=(.1, .2, .3, .4), uv_rect=(9, 10, 11, 12), uv_type=1, flip=(1.0, 0.0))
    assert inst.nbytes == _sizeof("CompositeInstance")
    assert list(inst[0:4]) == [1, 2, 3, 4] and list(inst[4:8]) == [5, 6, 7, 8]
    assert inst[12] == 0.0 and inst[13] == 1.0            # _padding, color_space_or_uv_type
    assert list(inst[16:20]) == [9, 10, 11, 12] and list(inst[28:30]) == [1.0, 0.0]


def test_numpy_dtypes_follow_the_struct_field_order():
    """Field by field (offset and byte size) for the structs the scenes fill through numpy records."""
    def flat(struct):
        out = []
        for name, ty in FIX["structs"][struct]["fields"]:
            if ty in FIX["structs"] and not re.fullmatch(r"\[.*\]", ty):
                out += flat(ty)
            else:
                out.append((name, _sizeof(ty)))
        return out
    def offsets(dt):
        return [(dt.fields[n][1], dt.fields[n][0].itemsize) for n in dt.names]
    # BlurInstance: task, src task, direction | std deviation, region
    assert [s for _, s in flat("BlurInstance")] == [4, 4, 4, 4, 8] and [s for _, s in offsets(scenes.BLUR_DTYPE)] == [12, 12]
    # ScalingInstance
    assert [s for _, s in flat("ScalingInstance")] == [s for _, s in offsets(scenes.SCALE_DTYPE)]
    # ClipMaskInstanceCommon prefix of both clip-mask instances: sub rect, task origin + screen origin, scale, two transform ids
    common = [s for _, s in flat("ClipMaskInstanceCommon")]
    assert common == [16, 8, 8, 4, 4, 4]
    for dt in (scenes.CLIP_RECT_DTYPE, scenes.BOX_SHADOW_DTYPE):
        assert [s for _, s in offsets(dt)][:4] == [16, 16, 4, 8]
    # ClipMaskInstanceRect: + local_pos, ClipData = (rect, mode) + 4 x (rect, 4 radii)
    rect = flat("ClipMaskInstanceRect")[len(common):]
    assert [s for _, s in rect] == [8, 16, 4] + [16, 4, 4, 4, 4] * 4
    assert [s for _, s in offsets(scenes.CLIP_RECT_DTYPE)][4:] == [8, 16, 4, 128]
    # ClipMaskInstanceBoxShadow: + resource address, BoxShadowData = src size, mode, stretch x, stretch y, dest rect
    box = flat("ClipMaskInstanceBoxShadow")[len(common):]
    assert [s for _, s in box] == [4, 8, 4, 4, 4, 16]
    assert [s for _, s in offsets(scenes.BOX_SHADOW_DTYPE)][4:] == [4, 8, 4, 8, 16]
    # BorderInstance
    assert [s for _, s in flat("BorderInstance")] == [8, 16, 16, 16, 4, 8, 8, 32]
    assert [s for _, s in offsets(scenes.BORDER_DTYPE)] == [8, 16, 16, 16, 4, 8, 8, 32]


# ---- the PrimitiveInstanceData encoders: the Rust expressions, evaluated, against frame.py's packers --------------------------
class _V:
    """A value with the accessors the Rust expressions use (`.0`, `.bits()`, `.as_int()`)."""
    def __init__(self, v):
        self.v = v
    def bits(self):
        return self.v
    def as_int(self):
        return self.v


def _eval_rust(expr, env):
    e = expr
    e = re.sub(r"as(?:u32|i32|u8)", "", e)                 # the casts (whitespace was squeezed out by the extractor)
    e = re.sub(r"\.0\b", ".v", e)
    e = re.sub(r"\b(instance|self)\.", "", e)
    e = re.sub(r"\b([a-z_][a-z_0-9]*)\b(?!\()", lambda m: m.group(1) if m.group(1) in ("v", "bits", "as_int") else f"env['{m.group(1)}']", e)
    val = eval(e, {"env": env})
    return val.v if isinstance(val, _V) else val


def _i32(x):
    return int(np.int32(np.uint32(int(x) & 0xFFFFFFFF)))


def test_brush_instance_packing_matches_gpu_types_rs():
    rng = np.random.default_rng(11)
    f = Frame(64, 64, (0, 0, 0, 0))
    for _ in range(50):
        ph, clip, seg = int(rng.integers(0, 1 << 20)), int(rng.integers(0, 1 << 15)), int(rng.integers(0, 0xFFFF))
        bf, ef, res = int(rng.integers(0, 1 << 12)), int(rng.integers(0, 16)), int(rng.integers(0, 1 << 24))
        env = {"prim_header_index": _V(ph), "clip_task_address": _V(clip), "segment_index": seg, "brush_flags": _V(bf),
               "edge_flags": _V(ef), "resource_address": res}
        want = [_i32(_eval_rust(w, env)) for w in FIX["packing"]["BrushInstance"]["data"]]
        got = [_i32(v) for v in f.brush_instance(ph, clip, segment=seg, brush_flags=bf, edge_flags=ef, resource_address=res)]
        assert got == want, FIX["packing"]["BrushInstance"]["at"]


def test_glyph_instance_packing_matches_gpu_types_rs():
    rng = np.random.default_rng(12)
    for _ in range(50):
        ph, clip, gi = int(rng.integers(0, 1 << 20)), int(rng.integers(0, 1 << 15)), int(rng.integers(0, 1 << 16))
        sd, cm, uv = int(rng.integers(0, 4)), int(rng.integers(0, 10)), int(rng.integers(0, 1 << 24))
        env = {"prim_header_index": _V(ph), "clip_task": _V(clip), "subpx_dir": sd, "color_mode": cm, "glyph_index_in_text_run": gi,
               "glyph_uv_rect": _V(uv)}
        want = [_i32(_eval_rust(w, env)) for w in FIX["packing"]["GlyphInstance"]["data"]]
        got = [_i32(v) for v in Frame.glyph_instance(ph, gi, uv, clip_task=clip, subpx_dir=sd, color_mode=cm)]
        assert got == want, FIX["packing"]["GlyphInstance"]["at"]


def test_quad_instance_and_prim_blocks_match_quad_rs():
    """QuadInstance -> PrimitiveInstanceData (gpu_types.rs) and the blocks behind it: write_prim_blocks' push order and the
    QuadHeader int block (quad.rs)."""
    rng = np.random.default_rng(13)
    pb = FIX["prim_blocks"]
    assert [a for _, a in pb["blocks"]] == ["prim_rect", "clip_rect", "pattern_texture_input", "scale_offset", "pattern_base_color.premultiplied()",
                                            "segment.rect", "segment.task_id"] and pb["count"] == "5+segments.len()*2", pb["at"]
    assert FIX["quad_header"]["words"] == ["transform_id.0asi32", "z_id.0", "pattern_input.0", "pattern_input.1"], FIX["quad_header"]["at"]
    for _ in range(20):
        f = Frame(64, 64, (0, 0, 0, 0))
        bounds, clip = [float(v) for v in rng.uniform(0, 64, 4)], [float(v) for v in rng.uniform(0, 64, 4)]
        color = [float(v) for v in rng.uniform(0, 1, 4)]
        uv, so = [float(v) for v in rng.uniform(0, 1, 4)], [float(v) for v in rng.uniform(0, 2, 4)]
        tid, z, task = int(rng.integers(0, 1 << 20)), int(rng.integers(1, 1 << 20)), int(rng.integers(0, 1 << 15))
        qf, ef, part, seg = (int(rng.integers(0, 256)) for _ in range(4))
        pin = (int(rng.integers(0, 100)), int(rng.integers(0, 100)))
        nseg = int(rng.integers(0, 3))
        segs = [([float(v) for v in rng.uniform(0, 64, 4)], [float(v) for v in rng.uniform(0, 1, 4)]) for _ in range(nseg)]
        inst = f.quad_instance(bounds, clip, color, z, task, transform_id=tid, quad_flags=qf, edge_flags=ef, part=part, segment=seg,
                               uv_rect=uv, scale_offset=so, segments=segs, pattern_input=pin)
        addr_i, addr_f = inst[0], inst[1]
        env = {"prim_address_i": _V(addr_i), "prim_address_f": _V(addr_f), "quad_flags": qf, "edge_flags": ef, "part_index": part,
               "segment_index": seg, "dst_task_address": _V(task)}
        want = [_i32(_eval_rust(w, env)) for w in FIX["packing"]["QuadInstance"]["data"]]
        assert [_i32(v) for v in inst] == want, FIX["packing"]["QuadInstance"]["at"]
        # the float blocks, in write_prim_blocks' order (the render-task slots hold the uv rect the task resolves to)
        fb = np.asarray(f.gpu_buffer_f.data[addr_f:addr_f + 5 + 2 * nseg], np.float32)
        np.testing.assert_array_equal(fb[0], np.float32(bounds)); np.testing.assert_array_equal(fb[1], np.float32(clip))
        np.testing.assert_array_equal(fb[2], np.float32(uv)); np.testing.assert_array_equal(fb[3], np.float32(so))
        np.testing.assert_array_equal(fb[4], np.float32(color))
        for k, (r, u) in enumerate(segs):
            np.testing.assert_array_equal(fb[5 + 2 * k], np.float32(r)); np.testing.assert_array_equal(fb[6 + 2 * k], np.float32(u))
        ib = np.asarray(f.gpu_buffer_i.data[addr_i], np.int32)
        assert list(ib) == [tid, z, pin[0], pin[1]]
    assert FIX["quad_header"]["part_index"] == {"Center": 0, "Left": 1, "Top": 2, "Right": 3, "Bottom": 4, "All": 5}


def test_prim_headers_match_gpu_types_rs():
    """PrimitiveHeaderF = (local_rect, local_clip_rect); PrimitiveHeaderI = (z, specific address, transform id, task address, user data x 4)."""
    assert [n for n, _ in FIX["structs"]["PrimitiveHeaderF"]["fields"]] == ["local_rect", "local_clip_rect"]
    assert [n for n, _ in FIX["structs"]["PrimitiveHeaderI"]["fields"]] == ["z", "specific_prim_address", "transform_id", "render_task_address", "user_data"]
    f = Frame(64, 64, (0, 0, 0, 0))
    i = f.add_prim_header((1, 2, 3, 4), (5, 6, 7, 8), 9, 10, 11, 12, (13, 14, 15, 16))
    hf, hi = np.asarray(f.prim_headers_f.data, np.float32), np.asarray(f.prim_headers_i.data, np.int32)
    assert list(hf[2 * i]) == [1, 2, 3, 4] and list(hf[2 * i + 1]) == [5, 6, 7, 8]
    assert list(hi[2 * i]) == [9, 10, 11, 12] and list(hi[2 * i + 1]) == [13, 14, 15, 16]


# ---- sampler slots and the backend's program table ----------------------------------------------------------------------------
def _types_enum(name):
    t = open(os.path.join(ROOT, "webrender_amd", "csrc", "wrhip_types.h")).read()
    body = re.search(r"enum %s \{(.*?)\};" % name, t, re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    out, nxt = {}, 0
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, v = (s.strip() for s in item.split("="))
            nxt = int(v, 0)
        else:
            k = item
        out[k] = nxt
        nxt += 1
    return out


def test_sampler_slots_match_renderer_mod_rs():
    slots = _types_enum("WrSlot")
    want = FIX["samplers"]["slots"]
    mine = {"Color0": "WR_S_COLOR0", "Color1": "WR_S_COLOR1", "Color2": "WR_S_COLOR2", "GpuCache": "WR_S_GPU_CACHE",
            "TransformPalette": "WR_S_TRANSFORMS", "RenderTasks": "WR_S_RENDER_TASKS", "Dither": "WR_S_DITHER",
            "PrimitiveHeadersF": "WR_S_PRIM_HEADERS_F", "PrimitiveHeadersI": "WR_S_PRIM_HEADERS_I", "ClipMask": "WR_S_CLIP_MASK",
            "GpuBufferF": "WR_S_GPU_BUFFER_F", "GpuBufferI": "WR_S_GPU_BUFFER_I"}
    assert {k: slots[v] for k, v in mine.items()} == want, FIX["samplers"]["at"]


def _program_table():
    t = open(os.path.join(ROOT, "webrender_amd", "csrc", "wrhip.hip")).read()
    body = t[t.index("const ShaderInfo SHADERS[] = {"):t.index("#undef S\nconst char* const SAMPLER_NAMES")]
    macro = re.search(r"#define CLIP_RECT_ATTRIBS\s*\\\n(.*?)\}\n", body, re.S).group(1)
    body = body.replace("CLIP_RECT_ATTRIBS,", macro.replace("\\\n", " ") + "},")
    out = {}
    for m in re.finditer(r'\{"([a-z_]+)(?: ([A-Z_0-9,]+))?", (WR_SH_\w+),\s*\{((?:\s*"\w+",?)+)\s*\}', body):
        out[(m.group(1), m.group(2) or "")] = re.findall(r'"(\w+)"', m.group(4))
    return out


def test_program_table_attributes_are_the_descriptors_of_shade_rs():
    """Every program the backend knows: its attribute list is aPosition + a subsequence (same order) of the instance attributes
    of the vertex descriptor renderer/shade.rs creates that shader with."""
    table = _program_table()
    assert len(table) >= 36
    sv = FIX["shader_vaos"]
    for (name, feats), attribs in table.items():
        assert name in sv["shaders"], f"{name}: not a shader renderer/shade.rs creates"
        desc = sv["kind_to_desc"][sv["shaders"][name]]
        want = [a for a, _, _ in FIX["descriptors"][desc]["instance"]]
        assert attribs[0] == "aPosition" == FIX["descriptors"][desc]["vertex"][0][0]
        it = iter(want)
        assert all(a in it for a in attribs[1:]), (name, feats, attribs, desc, want)
