"""Cross-oracle check: the reference's rasteriser built with the shader headers GENERATED from the reference's GLSL
(oracle/gen/gen_shaders.py -> oracle/_ref/libswgl_ref_gen.so: all 82 program keys swgl compiles) against the same
rasteriser built with the hand-written headers (oracle/shaders/*.h -> libswgl_ref_gcc.so) -- two independent derivations
of the shader side -- on every parity scene: identical bytes in every render target that is read back.

With this green, a parity test against `oracle_gcc` is a parity test against "the reference's GLSL, translated
mechanically and compiled here with the reference's own gl.cc"."""
import numpy as np
import pytest
from webrender_amd import scenes
from webrender_amd.harness import render_direct
from parity_cases import (OCCLUDED, BLEND, ROTATED, BORDERS, BORDER_SEGMENTS, DECORATIONS, FLAT, RUN_OVERFLOW, COPIES,
                          MIX_BLEND, DUAL_SOURCE, SPLIT, GLYPH_TRANSFORM)
from test_hostsim_parity import CASES, BLUR_CASES, CLIP_CASES, BOX_CASES


def _same(a, b):
    if isinstance(a, dict):
        assert set(a) == set(b)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    else:
        assert np.array_equal(a, b)


_SCENES = [(n, m) for n, m in CASES + OCCLUDED + BLEND + ROTATED + SPLIT + GLYPH_TRANSFORM + FLAT + RUN_OVERFLOW]
_SCENES += [(n, (lambda kw=kw: scenes.blur_chain(**kw))) for n, kw in BLUR_CASES]
_SCENES += [(n, (lambda kw=kw: scenes.clip_masks(**kw))) for n, kw in CLIP_CASES]
_SCENES += [(n, (lambda kw=kw: scenes.box_shadow_masks(**kw))) for n, kw in BOX_CASES]
_SCENES += [("cfg4_small", lambda: scenes.cfg4_box_shadow(width=1024, height=1024))]
_SCENES += [(n, (lambda kw=kw: scenes.border_solid(**kw))) for n, kw in BORDERS]
_SCENES += [(n, (lambda kw=kw: scenes.border_segments(**kw))) for n, kw in BORDER_SEGMENTS]
_SCENES += [(n, (lambda kw=kw: scenes.cache_decorations(**kw))) for n, kw in DECORATIONS]
_SCENES += [(n, (lambda kw=kw: scenes.texture_cache_copies(**kw))) for n, kw in COPIES]
# (the hand-written brush_mix_blend header has no perspective entry points -- it was written for 2-D transforms, gl_FragCoord.w fixed at 1 --,
# so the projective cases are the generated oracle's alone: the header glsl-to-cxx's restatement emits from the reference's GLSL)
_SCENES += [(n, (lambda s=s, kw=kw: getattr(scenes, s)(**kw))) for n, s, kw in MIX_BLEND if not kw.get("perspective")]
_SCENES += [(n, (lambda kw=kw: scenes.image_grid(**kw))) for n, kw in DUAL_SOURCE]


@pytest.mark.parametrize("name,make", _SCENES, ids=[c[0] for c in _SCENES])
def test_generated_oracle_equals_handwritten_oracle(oracle_hand, oracle_gen, name, make):
    want, _ = render_direct(oracle_hand, make())
    got, _ = render_direct(oracle_gen, make())
    _same(got, want)


def test_generated_oracle_loads_every_swgl_key(oracle_gen):
    """every "name FEATURES" key of swgl/build.rs's list links in the generated library (ShaderSourceByName + LinkProgram)"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle", "gen"))
    from gen_shaders import shader_keys
    from webrender_amd import glapi, glconst as G
    gl = glapi.GL(oracle_gen)
    ctx = gl.CreateContext()
    gl.MakeCurrent(ctx)
    keys = shader_keys()
    assert len(keys) == 82
    for key in keys:
        vs, fs = gl.CreateShader(G.GL_VERTEX_SHADER), gl.CreateShader(G.GL_FRAGMENT_SHADER)
        gl.ShaderSourceByName(vs, key.encode())
        gl.ShaderSourceByName(fs, key.encode())
        pid = gl.CreateProgram()
        gl.AttachShader(pid, vs)
        gl.AttachShader(pid, fs)
        gl.LinkProgram(pid)
        assert gl.GetLinkStatus(pid), key
    gl.DestroyContext(ctx)
