#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE -- never imported by the product path.

Generates the shader headers swgl is built with, mechanically, from the
reference's own GLSL where it lies under /root/reference -- what
`swgl/build.rs` does with the `glsl-to-cxx` crate (no Rust toolchain here):

  1. the key list            swgl/build.rs:107-124 + webrender_build/src/shader_features.rs:60-233
                             (flags GL | DUAL_SOURCE_BLENDING | ADVANCED_BLEND_EQUATION | DEBUG)
  2. source assembly         swgl/build.rs:40-71   (`#include` expansion, once per file; SWGL,
                             __VERSION__ 150, WR_MAX_VERTEX_TEXTURE_WIDTH, WR_FEATURE_* defines)
  3. preprocessing           swgl/build.rs:76-105  (`cc -xc -P -undef` with WR_VERTEX_SHADER /
                             WR_FRAGMENT_SHADER)
  4. translation             glsl_cxx.translate    (glsl-to-cxx/src/lib.rs, hir.rs restated)
  5. load_shader.h           swgl/build.rs:19-38

Outputs go to the directory given with --out (oracle/_ref/gen, git-ignored): they
are derived from the reference's sources and are build products, not repo content.

    python3 oracle/gen/gen_shaders.py --ref /root/reference --out oracle/_ref/gen
"""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.setrecursionlimit(10000)
import glsl_cxx  # noqa: E402

MAX_VERTEX_TEXTURE_WIDTH = 1024  # webrender_build/src/lib.rs:19


def shader_keys():
    """get_shader_features(GL | DUAL_SOURCE_BLENDING | ADVANCED_BLEND_EQUATION | DEBUG), as "name FEATURES"."""
    shaders = {}

    def finish(lst):
        return ",".join(sorted(lst))

    shaders["cs_clip_rectangle"] = ["", "FAST_PATH"]
    shaders["cs_clip_box_shadow"] = ["TEXTURE_2D"]
    shaders["cs_blur"] = ["ALPHA_TARGET", "COLOR_TARGET"]
    shaders["ps_quad_mask"] = ["", "FAST_PATH"]
    for name in ("cs_line_decoration", "cs_fast_linear_gradient", "cs_border_segment", "cs_border_solid",
                 "cs_svg_filter", "cs_svg_filter_node"):
        shaders[name] = [""]
    for name in ("cs_linear_gradient", "cs_radial_gradient", "cs_conic_gradient"):
        shaders[name] = [""]
    base = []
    alpha = base + ["ALPHA_PASS"]
    for name in ("brush_solid", "brush_blend", "brush_mix_blend"):
        shaders[name] = [finish(base), finish(alpha), "DEBUG_OVERDRAW"]
    shaders["brush_linear_gradient"] = [finish(base), finish(alpha), "DEBUG_OVERDRAW"]
    shaders["brush_opacity"] = [finish(base), finish(alpha), finish(base + ["ANTIALIASING"]),
                                finish(alpha + ["ANTIALIASING"]), "ANTIALIASING,DEBUG_OVERDRAW", "DEBUG_OVERDRAW"]
    texture_types = ["TEXTURE_2D", "TEXTURE_RECT"]
    image = []
    for tt in texture_types:
        fast = [tt]
        image.append(finish(fast + base))
        image.append(finish(fast + alpha))
        image.append(finish(fast + ["DEBUG_OVERDRAW"]))
        slow = fast + ["REPETITION", "ANTIALIASING"]
        image.append(finish(slow + base))
        image.append(finish(slow + alpha))
        image.append(finish(slow + ["DEBUG_OVERDRAW"]))
        adv = alpha + ["ADVANCED_BLEND"]
        image.append(finish(fast + adv))
        image.append(finish(slow + adv))
        dual = alpha + ["DUAL_SOURCE_BLENDING"]
        image.append(finish(fast + dual))
        image.append(finish(slow + dual))
    shaders["brush_image"] = image
    composite = list(texture_types)
    shaders["cs_scale"] = list(composite)
    yuv = []
    for tt in texture_types:
        lst = [tt, "YUV"]
        composite.append(finish(lst))
        yuv.append(finish(lst + base))
        yuv.append(finish(lst + alpha))
        yuv.append(finish(lst + ["DEBUG_OVERDRAW"]))
    shaders["brush_yuv_image"] = yuv
    for tt in texture_types:
        composite.append(finish([tt, "FAST_PATH"]))
    shaders["composite"] = composite
    text = []
    for text_type in ("", "DUAL_SOURCE_BLENDING"):
        lst = base + ["TEXTURE_2D"] + ([text_type] if text_type else [])
        alpha_list = lst + ["ALPHA_PASS"]
        text.append(finish(alpha_list))
        text.append(finish(alpha_list + ["GLYPH_TRANSFORM"]))
        text.append(finish(lst + ["DEBUG_OVERDRAW"]))
    shaders["ps_text_run"] = text
    for name in ("ps_split_composite", "ps_quad_textured", "ps_quad_radial_gradient", "ps_quad_conic_gradient",
                 "ps_clear", "ps_copy", "debug_color", "debug_font"):
        shaders[name] = [""]
    keys = []
    for name, feats in shaders.items():
        for f in feats:
            keys.append(name if not f else f"{name} {f}")
    return sorted(keys)


def shader_file(key):
    return key.replace(" ", "_").replace(",", "_")


def process_imports(shader_dir, shader, included, out):
    if shader in included:
        return
    included.add(shader)
    with open(os.path.join(shader_dir, shader + ".glsl")) as f:
        source = f.read()
    for line in source.splitlines():
        if line.startswith("#include "):
            for imp in line[len("#include "):].split(","):
                process_imports(shader_dir, imp, included, out)
        elif line.startswith("#version ") or line.startswith("#extension "):
            pass
        else:
            out.append(line)


def preprocess(key, shader_dir, out_dir, cc="gcc"):
    base, _, features = key.partition(" ")
    lines = ["#define SWGL 1", "#define __VERSION__ 150",
             f"#define WR_MAX_VERTEX_TEXTURE_WIDTH {MAX_VERTEX_TEXTURE_WIDTH}U"]
    if features:
        for feature in features.strip().split(","):
            lines.append(f"#define WR_FEATURE_{feature}")
    process_imports(shader_dir, base, set(), lines)
    fn = os.path.join(out_dir, shader_file(key) + ".c")
    with open(fn, "w") as f:
        f.write("\n".join(lines) + "\n")
    stages = []
    for define in ("WR_VERTEX_SHADER", "WR_FRAGMENT_SHADER"):
        r = subprocess.run([cc, "-E", "-xc", "-P", "-undef", f"-D{define}=1", fn], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"preprocessing {key}: {r.stderr}")
        stages.append(r.stdout)
    os.remove(fn)
    return stages


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", required=True)
    ap.add_argument("--keys", nargs="*", help="only these keys (default: all swgl builds)")
    ap.add_argument("--keep-stages", action="store_true", help="keep the preprocessed .vert/.frag next to the headers")
    args = ap.parse_args()
    shader_dir = os.path.join(args.ref, "webrender", "res")
    os.makedirs(args.out, exist_ok=True)
    keys = args.keys or shader_keys()
    done = []
    for key in keys:
        vs, fs = preprocess(key, shader_dir, args.out)
        name = shader_file(key)
        if args.keep_stages:
            open(os.path.join(args.out, name + ".vert"), "w").write(vs)
            open(os.path.join(args.out, name + ".frag"), "w").write(fs)
        try:
            text = glsl_cxx.translate(name, vs, fs)
        except Exception as exc:  # a key the translator cannot handle is an error, never skipped silently
            raise RuntimeError(f"translating {key!r}: {exc}") from exc
        with open(os.path.join(args.out, name + ".h"), "w") as f:
            f.write(f"// generated by oracle/gen/gen_shaders.py from webrender/res/{key.split(' ')[0]}.glsl "
                    f"(key \"{key}\") -- build product, do not commit\n")
            f.write(text)
        done.append(key)
    with open(os.path.join(args.out, "load_shader.h"), "w") as f:
        for key in done:
            f.write(f"#include \"{shader_file(key)}.h\"\n")
        f.write("ProgramLoader load_shader(const char* name) {\n")
        for key in done:
            f.write(f"  if (!strcmp(name, \"{key}\")) {{ return {shader_file(key)}_program::loader; }}\n")
        f.write("  return nullptr;\n}\n")
    print(f"generated {len(done)} shader headers into {args.out}")


if __name__ == "__main__":
    main()
