"""The restated display lists of wrench/reftests/text (webrender_amd/wrench_scenes.text_reftest) against the reference's OWN expectations:
wrench/reftests/text/reftest.list states which display lists must render alike (`==`, some with a fuzz of (max difference, pixels)) and
which must not (`!=`).  No reference PNG or wrench binary exists here, but the relations between two yamls both restated can be checked by
drawing both with the oracle (the reference's swgl): they pin the restatement's reading of the scene builder -- shadow contexts
(scene_building.rs:2897-3047: offset-only shadows as extra prims, blurred ones as one picture per shadow), solid line decorations as rects,
local clip rects moved with a shadow's offset, the order of shadows and prims -- far more tightly than "hostsim == oracle" does, which
holds for any scene.  Pairs whose relation is stated for another configuration (`options(disable-aa)`: snap-clip, transparent-no-aa holds
anyway) or whose other side needs a transform are left out."""
import numpy as np
import pytest
from webrender_amd import wrench_scenes as ws
from webrender_amd.harness import render_direct

# (test, reference, max difference allowed, pixels allowed) -- reftest.list, the `==` lines among the restated yamls
EQUAL = [
    ("shadow", "shadow-ref", 1, 3),
    ("shadow-atomic", "shadow-atomic-ref", 1, 64),
    ("shadow-clip-rect", "shadow-atomic-ref", 1, 64),
    ("shadow-ordering", "shadow-ordering-ref", 1, 1),
    ("decorations", "decorations-ref", 0, 0),
    ("1658", "1658-ref", 0, 0),
    ("subtle-shadow", "subtle-shadow-ref", 0, 0),
    ("shadow-partial-glyph", "shadow-partial-glyph-ref", 0, 0),
    ("transparent-no-aa", "transparent-no-aa-ref", 0, 0),
    # text in a reference frame under a fractional translation: the run's offset is snapped THROUGH the transform, and the shader adds it
    # after flooring the glyph offsets (ps_text_run.glsl:170-190) -- a first reading that folded it into the glyph offsets failed both lines
    ("subpixel-translate", "subpixel-translate-ref", 1, 381),
    ("snap-text-offset", "snap-text-offset-ref", 0, 0),
    # rectangular clip nodes on prims and on shadows (a blurred shadow's chain clips the picture after the blur, not the prims inside it)
    ("shadow-clip", "shadow-clip-ref", 0, 0),
    ("shadow-fast-clip", "shadow-fast-clip-ref", 0, 0),
]
# ... and the `!=` lines
DIFFERENT = [
    ("text", "blank"), ("long-text", "blank"), ("negative-pos", "blank"), ("shadow", "text"), ("shadow-single", "blank"),
    ("shadow-cover-1", "blank"), ("shadow-cover-2", "blank"), ("shadow-cover-1", "shadow-cover-2"), ("shadow-many", "shadow"),
    ("shadow-complex", "shadow-many"), ("non-opaque", "non-opaque-notref"), ("diacritics", "diacritics-ref"),
]
_cache = {}


def _window(lib, name):
    if name not in _cache:
        out, _ = render_direct(lib, ws.text_reftest(name, width=1024, height=768))
        _cache[name] = out["window"] if isinstance(out, dict) else out
    return _cache[name]


@pytest.mark.parametrize("a,b,max_diff,pixels", EQUAL, ids=[f"{e[0]}=={e[1]}" for e in EQUAL])
def test_reftest_list_equalities_hold(oracle_gcc, a, b, max_diff, pixels):
    d = np.abs(_window(oracle_gcc, a).astype(np.int16) - _window(oracle_gcc, b).astype(np.int16)).max(axis=2)
    assert int(d.max()) <= max_diff and int((d > 0).sum()) <= pixels, (int(d.max()), int((d > 0).sum()))


@pytest.mark.parametrize("a,b", DIFFERENT, ids=[f"{e[0]}!={e[1]}" for e in DIFFERENT])
def test_reftest_list_inequalities_hold(oracle_gcc, a, b):
    assert not np.array_equal(_window(oracle_gcc, a), _window(oracle_gcc, b))
