#!/usr/bin/env python3
"""Extract the instance / vertex layouts this backend depends on from the reference's RUST sources, mechanically.

  python tests/golden/make_layouts.py [/root/reference] > tests/golden/reference_layouts.json

Test infrastructure.  What comes out (tests/test_reference_layouts.py holds this repository's own encoders to it):
  descriptors   renderer/vertex.rs `desc::*`: per VertexDescriptor the (name, count, kind) lists, in order
  structs       gpu_types.rs `#[repr(C)]` instance structs: (field, type) in declaration order
  packing       the PrimitiveInstanceData encoders (BrushInstance, QuadInstance, GlyphInstance::build,
                SplitCompositeInstance, MaskInstance): the four data words as Rust expressions
  prim_blocks   quad.rs write_prim_blocks: what is pushed, in order
  quad_header   quad.rs add_to_batch: the int block (QuadHeader)
  samplers      renderer/mod.rs TextureSampler -> slot
  shader_vaos   renderer/shade.rs: shader name -> VertexArrayKind
Every item carries the file:line it was read from.
"""
import json
import os
import re
import sys


def read(root, rel):
    with open(os.path.join(root, rel)) as f:
        return f.read()


def line_of(text, pos):
    return text.count("\n", 0, pos) + 1


def parse_descriptors(root):
    rel = "webrender/src/renderer/vertex.rs"
    t = read(root, rel)
    out = {}
    for m in re.finditer(r"pub const (\w+): VertexDescriptor = VertexDescriptor \{", t):
        # the body runs to the matching closing brace of the struct literal
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(t[i], 0)
            i += 1
        body = t[m.end():i]
        parts = {}
        for key in ("vertex_attributes", "instance_attributes"):
            k = body.index(key)
            # the slice literal &[ ... ] that follows
            a = body.index("&[", k) + 2
            d, j = 1, a
            while d:
                d += {"[": 1, "]": -1}.get(body[j], 0)
                j += 1
            attrs = re.findall(r'name:\s*"(\w+)",\s*count:\s*(\d+),\s*kind:\s*VertexAttributeKind::(\w+)', body[a:j])
            parts[key] = [[n, int(c), kd] for n, c, kd in attrs]
        out[m.group(1)] = {"vertex": parts["vertex_attributes"], "instance": parts["instance_attributes"],
                           "at": f"{rel}:{line_of(t, m.start())}"}
    return out


def parse_structs(root):
    rel = "webrender/src/gpu_types.rs"
    t = read(root, rel)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct (\w+) \{(.*?)\n\}", t, re.S):
        fields = re.findall(r"^\s*(?:pub(?:\(\w+\))? )?(\w+):\s*(.+?),\s*(?://.*)?$", m.group(2), re.M)
        out[m.group(1)] = {"fields": [[n, ty.strip()] for n, ty in fields], "at": f"{rel}:{line_of(t, m.start(1))}"}
    return out


def parse_clip_data(root):
    rel = "webrender/src/prim_store/mod.rs"
    t = read(root, rel)
    out = {}
    for name in ("ClipRect", "ClipCorner", "ClipData"):
        m = re.search(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*(?:pub )?struct %s \{(.*?)\n\}" % name, t, re.S)
        fields = re.findall(r"^\s*(?:pub )?(\w+):\s*(.+?),\s*(?://.*)?$", m.group(1), re.M)
        out[name] = {"fields": [[n, ty.strip()] for n, ty in fields], "at": f"{rel}:{line_of(t, m.start())}"}
    return out


def parse_shader_vaos(root):
    """renderer/shade.rs: which VertexArrayKind (hence which desc::*) every shader is created with."""
    rel = "webrender/src/renderer/shade.rs"
    t = read(root, rel)
    kinds = dict(re.findall(r"VertexArrayKind::(\w+) => &desc::(\w+),", t))
    out = {"kind_to_desc": kinds, "shaders": {}, "at": f"{rel}:251"}
    for m in re.finditer(r"LazilyCompiledShader::new\(\s*ShaderKind::(\w+)(?:\(VertexArrayKind::(\w+)\))?,\s*\"(\w+)\"", t):
        sk, vk, name = m.groups()
        vk = vk or {"Primitive": "Primitive", "Brush": "Primitive", "Text": "Primitive", "Composite": "Composite", "Clear": "Clear",
                    "Copy": "Copy", "Resolve": "Resolve"}.get(sk, sk)
        out["shaders"].setdefault(name, vk)
    for m in re.finditer(r"(?:BrushShader|TextShader)::new\(\s*\"(\w+)\"", t):
        out["shaders"].setdefault(m.group(1), "Primitive")
    return out


def _data_words(t, start):
    """The `data: [ a, b, c, d ]` literal after position `start`: four expressions, whitespace and comments squeezed out."""
    k = t.index("data: [", start)
    a = k + len("data: [")
    d, j = 1, a
    while d:
        d += {"[": 1, "]": -1}.get(t[j], 0)
        j += 1
    body = re.sub(r"/\*.*?\*/|//[^\n]*", "", t[a:j - 1], flags=re.S)
    words, cur, depth = [], "", 0
    for ch in body:
        if ch == "," and depth == 0:
            words.append(cur)
            cur = ""
            continue
        depth += {"(": 1, ")": -1}.get(ch, 0)
        cur += ch
    if cur.strip():
        words.append(cur)
    return [re.sub(r"\s+", "", w) for w in words if w.strip()], k


def parse_packing(root):
    rel = "webrender/src/gpu_types.rs"
    t = read(root, rel)
    out = {}
    anchors = {"BrushInstance": "impl From<BrushInstance> for PrimitiveInstanceData", "QuadInstance": "impl From<QuadInstance> for PrimitiveInstanceData",
               "SplitCompositeInstance": "impl From<SplitCompositeInstance> for PrimitiveInstanceData", "GlyphInstance": "impl GlyphInstance"}
    for name, anchor in anchors.items():
        words, k = _data_words(t, t.index(anchor))
        out[name] = {"data": words, "at": f"{rel}:{line_of(t, k)}"}
    return out


def parse_quad(root):
    rel = "webrender/src/quad.rs"
    t = read(root, rel)
    k = t.index("pub fn write_prim_blocks(")
    end = t.index("writer.finish()", k)
    pushes = re.findall(r"writer\.(push_one|push_render_task)\(([^;]*?)\);?\n", t[k:end])
    blocks = [[fn, re.sub(r"\s+", "", arg)] for fn, arg in pushes]
    count = re.search(r"write_blocks\((.*?)\);", t[k:end]).group(1).replace(" ", "")
    k2 = t.index("// See QuadHeader in ps_quad.glsl")
    m = re.search(r"writer\.push_one\(\[(.*?)\]\);", t[k2:], re.S)
    header = [re.sub(r"\s+", "", w) for w in m.group(1).split(",") if w.strip()]
    parts = re.findall(r"(\w+) = (\d+),", t[t.index("enum PartIndex", k2 - 600):k2])
    return ({"blocks": blocks, "count": count, "at": f"{rel}:{line_of(t, k)}"},
            {"words": header, "part_index": {n: int(v) for n, v in parts}, "at": f"{rel}:{line_of(t, k2)}"})


def parse_samplers(root):
    rel = "webrender/src/renderer/mod.rs"
    t = read(root, rel)
    k = t.index("impl Into<TextureSlot> for TextureSampler")
    end = t.index("}\n}", k)
    pairs = re.findall(r"TextureSampler::(\w+) => TextureSlot\((\d+)\)", t[k:end])
    return {"slots": {n: int(v) for n, v in pairs}, "at": f"{rel}:{line_of(t, k)}"}


def parse_prim_header(root):
    rel = "webrender/src/gpu_types.rs"
    t = read(root, rel)
    out = {}
    for name in ("PrimitiveHeaderF", "PrimitiveHeaderI"):
        m = re.search(r"pub struct %s \{(.*?)\n\}" % name, t, re.S)
        fields = re.findall(r"^\s*pub (\w+):\s*([^,\n]+),", m.group(1), re.M)
        out[name] = {"fields": [[n, ty.strip()] for n, ty in fields], "at": f"{rel}:{line_of(t, m.start())}"}
    return out


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    prim_blocks, quad_header = parse_quad(root)
    structs = parse_structs(root)
    structs.update(parse_prim_header(root))
    structs.update(parse_clip_data(root))
    out = {"source": "servo/webrender (reference tree), extracted by tests/golden/make_layouts.py",
           "descriptors": parse_descriptors(root), "structs": structs, "packing": parse_packing(root),
           "prim_blocks": prim_blocks, "quad_header": quad_header, "samplers": parse_samplers(root),
           "shader_vaos": parse_shader_vaos(root)}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
