#!/usr/bin/env python3
"""Generates webrender_amd/wrench/benchmarks.json from the display lists of the reference's own benchmark set
(/root/reference/wrench/benchmarks/*.yaml, benchmarks.list) -- the INPUT DATA of the workloads `bench.py --workload <name>`
restates, not code: item types, bounds, colours, text strings.  The GPU box has no /root/reference, so the lists travel as
this fixture; the scenes that consume it are webrender_amd/wrench_scenes.py.

    python3 tests/golden/make_wrench_benchmarks.py        (in a container that has /root/reference)
"""
import json
import os
import re
import yaml

REF = "/root/reference/wrench/benchmarks"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "webrender_amd", "wrench", "benchmarks.json")


def nums(v):
    if isinstance(v, (list, tuple)):
        return [float(x) for x in v]
    return [float(x) for x in str(v).replace(",", " ").split()]


def main():
    out = {"source": "wrench/benchmarks/*.yaml of servo/webrender (display-list input data)"}
    # text-rendering.yaml: runs of text (wrench/src/yaml_frame_reader.rs handle_text: origin = baseline start, size in px)
    items = yaml.safe_load(open(os.path.join(REF, "text-rendering.yaml")))["root"]["items"]
    out["text-rendering"] = [{"text": it["text"], "origin": nums(it["origin"]), "size": float(it["size"]),
                              "color": it.get("color")} for it in items]
    # many-images.yaml: a regular grid of 8x8 solid-colour images; verified here, stored as its rule
    items = yaml.safe_load(open(os.path.join(REF, "many-images.yaml")))["root"]["items"]
    cols = max(int(nums(it["bounds"])[0]) for it in items) // 8 + 1
    for i, it in enumerate(items):
        c, r = i % cols, i // cols
        m = re.match(r"solid-color\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)", it["image"])
        assert [int(v) for v in m.groups()] == [c, r, 0, 255, 8, 8] and nums(it["bounds"]) == [8.0 * c, 8.0 * r, 8.0, 8.0], i
    out["many-images"] = {"count": len(items), "cols": cols, "size": 8, "rule": "image i: solid-color(i % cols, i // cols, 0, 255), bounds 8x8 at (8 (i % cols), 8 (i // cols))"}
    # aligned- / unaligned-gradient.yaml
    for name in ("aligned-gradient", "unaligned-gradient"):
        items = yaml.safe_load(open(os.path.join(REF, name + ".yaml")))["root"]["items"]
        out[name] = [{"bounds": nums(it["bounds"]), "start": nums(it["start"]), "end": nums(it["end"]), "stops": it["stops"],
                      "repeat": bool(it["repeat"])} for it in items]
    # many-box-shadows.yaml
    sc = yaml.safe_load(open(os.path.join(REF, "many-box-shadows.yaml")))["root"]["items"][0]
    out["many-box-shadows"] = [{"box-bounds": nums(it["box-bounds"]), "clip-rect": nums(it["clip-rect"]), "offset": nums(it["offset"]),
                                "color": nums(it["color"]), "blur-radius": float(it["blur-radius"]), "spread-radius": float(it["spread-radius"]),
                                "clip-mode": it["clip-mode"]} for it in sc["items"] if it["type"] == "box-shadow"]
    # large-boxshadow-ellipse.yaml: one outset box shadow with elliptical corner radii (wrench yaml_frame_reader.rs handle_box_shadow:
    # `bounds` is the box, border-radius {corner: [width, height]})
    it = yaml.safe_load(open(os.path.join(REF, "large-boxshadow-ellipse.yaml")))["root"]["items"][0]
    assert it["type"] == "box-shadow" and it["clip-mode"] == "outset"
    out["large-boxshadow-ellipse"] = {"bounds": nums(it["bounds"]), "color": it["color"], "blur-radius": float(it["blur-radius"]),
                                      "border-radius": {k: nums(v) for k, v in it["border-radius"].items()}}
    # large-boxshadow-ellipse-2.yaml (benchmarks.list:5): one INSET box shadow, blur radius 10000 (capped by the frame builder), radii of 400-700 px
    it = yaml.safe_load(open(os.path.join(REF, "large-boxshadow-ellipse-2.yaml")))["root"]["items"][0]
    assert it["type"] == "box-shadow" and it["clip-mode"] == "inset"
    out["large-boxshadow-ellipse-2"] = {"bounds": nums(it["bounds"]), "color": it["color"], "blur-radius": float(it["blur-radius"]),
                                        "clip-mode": it["clip-mode"], "border-radius": {k: nums(v) for k, v in it["border-radius"].items()}}
    # large-clip-rect.yaml: N identical opaque rects under one rounded-rectangle clip
    clip = yaml.safe_load(open(os.path.join(REF, "large-clip-rect.yaml")))["root"]["items"][0]
    assert clip["type"] == "clip" and len(clip["complex"]) == 1
    rects = [nums(r["bounds"]) for r in clip["items"]]
    assert all(r["type"] == "rect" and r["color"] == clip["items"][0]["color"] for r in clip["items"]) and all(r == rects[0] for r in rects)
    out["large-clip-rect"] = {"clip-bounds": nums(clip["bounds"]), "complex-rect": nums(clip["complex"][0]["rect"]),
                              "radius": float(clip["complex"][0]["radius"]), "rect-bounds": rects[0], "count": len(rects),
                              "color": clip["items"][0]["color"]}
    # clip-clear.yaml (in the directory, not in benchmarks.list): the same shape -- 11 rects of 300 x 300 under a rounded clip of
    # radius 50, inside a stacking context at the origin
    sc = yaml.safe_load(open(os.path.join(REF, "clip-clear.yaml")))["root"]["items"][0]
    assert sc["type"] == "stacking-context" and nums(sc["bounds"])[:2] == [0.0, 0.0] and len(sc["items"]) == 1
    clip = sc["items"][0]
    assert clip["type"] == "clip" and len(clip["complex"]) == 1
    rects = [nums(r["bounds"]) for r in clip["items"]]
    assert all(r["type"] == "rect" and r["color"] == clip["items"][0]["color"] for r in clip["items"]) and all(r == rects[0] for r in rects)
    out["clip-clear"] = {"clip-bounds": nums(clip["bounds"]), "complex-rect": nums(clip["complex"][0]["rect"]),
                         "radius": float(clip["complex"][0]["radius"]), "rect-bounds": rects[0], "count": len(rects),
                         "color": clip["items"][0]["color"]}
    # overlapping-text-shadows.yaml (not in benchmarks.list): 200 unblurred shadows (offset (i, i), red) around one text item
    items = yaml.safe_load(open(os.path.join(REF, "overlapping-text-shadows.yaml")))["root"]["items"]
    shadows = [it for it in items if it.get("type") == "shadow"]
    texts = [it for it in items if "text" in it]
    assert len(texts) == 1 and items[-1]["type"] == "pop-all-shadows" and items.index(texts[0]) == len(shadows)
    assert all("blur-radius" not in s for s in shadows)
    assert all(nums(s["offset"]) == [float(i), float(i)] and s["color"] == shadows[0]["color"] for i, s in enumerate(shadows))
    out["overlapping-text-shadows"] = {"shadow-count": len(shadows), "shadow-color": shadows[0]["color"], "rule": "shadow i: offset (i, i), no blur",
                                       "text": texts[0]["text"], "origin": nums(texts[0]["origin"]), "size": float(texts[0]["size"]),
                                       "color": texts[0].get("color")}
    # radial-gradient.yaml (not in benchmarks.list) is NOT restated: it gives `start-radius` / `end-radius`, and the reference's wrench
    # reads `radius` (wrench/src/yaml_helper.rs:1161-1163 `expect("radial gradient must have a radius")`): wrench itself panics on it
    # large-blur-radius.yaml: a stacking context with filter blur(100, 100) over one rect
    sc = yaml.safe_load(open(os.path.join(REF, "large-blur-radius.yaml")))["root"]["items"][0]
    assert sc["type"] == "stacking-context" and sc["filters"] == "blur(100, 100)" and len(sc["items"]) == 1 and sc["items"][0]["type"] == "rect"
    out["large-blur-radius"] = {"bounds": nums(sc["bounds"]), "blur": [100.0, 100.0], "rect-bounds": nums(sc["items"][0]["bounds"]),
                                "color": sc["items"][0]["color"]}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.normpath(OUT), {k: (len(v) if isinstance(v, list) else v.get("count")) for k, v in out.items() if k != "source"})


if __name__ == "__main__":
    main()
