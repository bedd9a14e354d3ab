"""Parity tests proper: libwrhip (HIP kernels on the MI355X, through the C ABI)
against the reference's swgl on the same seeded inputs.  Bit-exact: the
rectangle path is integer/byte work end to end (north_star tolerance is +-1 LSB,
the tests demand 0)."""
import hashlib
import json
import os
import numpy as np
import pytest
from conftest import ROOT, wrhip_lib, oracle_ref
from webrender_amd import scenes
from webrender_amd.harness import render_direct, record_scene, ScenePlayer
from parity_cases import TEXT_REFTESTS, TILE_ROWS, WRENCH, TEXTURE_RECT, YUV, SVG_FILTERS, OCCLUDED, BLEND, ROTATED, BORDERS, BORDER_SEGMENTS, DECORATIONS, FLAT, RUN_OVERFLOW, COPIES, copies_expected, MIX_BLEND, DUAL_SOURCE, REPEAT_DUAL, SPLIT, SPLIT_GOLDEN, GLYPH_TRANSFORM

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))


def digest(px):
    return hashlib.sha256(np.ascontiguousarray(px).tobytes()).hexdigest()


def golden_applies(name):
    """Text digests only hold when PIL rasterises the very same glyph bitmaps."""
    if not name.startswith("cfg3"):
        return True
    return digest(scenes.build_glyph_atlas()[0]) == GOLDEN.get("glyph_atlas")


def test_device_is_gfx950():
    from webrender_amd.glapi import load_wrhip
    gl = load_wrhip()
    ctx = gl.CreateContext()
    name = gl.WrhipDeviceName()
    assert name and b"gfx950" in name, name
    gl.DestroyContext(ctx)


FILTER_OPS_EXACT = [0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]

SMALL = [
    ("cfg1", lambda: scenes.cfg1_solid_colors()),
    ("cfg1_brush", lambda: scenes.cfg1_solid_colors(encoding="brush")),
    ("simple_batching", lambda: scenes.simple_batching()),
    ("cfg2_small", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=200, seed=7)),
    ("cfg2_small_brush", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=200, seed=7, encoding="brush")),
    ("cfg2_small_frac", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=200, seed=7, fractional=True)),
    ("cfg2_odd_size", lambda: scenes.cfg2_overlapping_rects(width=1000, height=700, n=150, seed=3, fractional=True)),
    ("cfg5_small", lambda: scenes.cfg5_many_rects(width=2048, height=1024, n=5000)),
    ("cfg3_small", lambda: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12)),
    ("cfg3_small_zoom", lambda: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, glyph_zoom=1.25)),
    ("cfg3_small_dps", lambda: scenes.cfg3_text(width=1000, height=500, lines=20, glyphs_per_line=60, run_len=12, device_pixel_scale=1.5)),
    ("cfg3_small_modes", lambda: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, color_modes=(0, 1, 2, 3))),
    ("cfg3_small_dual", lambda: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, color_modes=(1, 2), dual_source=True)),
    ("cfg3_small_modes_zoom", lambda: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, color_modes=(1, 2, 3), glyph_zoom=1.25)),
    ("masked_rects", lambda: scenes.masked_rects()),
    ("masked_rects_frac", lambda: scenes.masked_rects(fractional=True)),
    ("masked_rects_wide", lambda: scenes.masked_rects(width=2048, height=1024, n=600, seed=5)),
    ("scaled_composites", lambda: scenes.scaled_composites()),
    ("image_grid", lambda: scenes.image_grid()),
    ("image_grid_wide", lambda: scenes.image_grid(width=2048, height=1024, n=300, seed=52)),
    ("image_grid_masked", lambda: scenes.image_grid(masked=True)),
    ("filter_grid_masked", lambda: scenes.filter_grid(masked=True, seed=75, ops=FILTER_OPS_EXACT)),
    ("rotated_rects", lambda: scenes.rotated_rects()),
    ("rotated_images", lambda: scenes.rotated_images()),
    ("rotated_images_repeat", lambda: scenes.rotated_images(repeat=True)),
    ("rotated_images_masked", lambda: scenes.rotated_images(masked=True)),
    ("rotated_images_nearest_wide", lambda: scenes.rotated_images(nearest=True, seed=103, width=2048, n=120)),
    ("rotated_images_repeat_nearest", lambda: scenes.rotated_images(repeat=True, nearest=True, seed=102)),
    ("rotated_images_repeat_masked", lambda: scenes.rotated_images(repeat=True, masked=True, seed=104)),
    ("rotated_images_quad", lambda: scenes.rotated_images(encoding="quad")),
    ("image_grid_nearest", lambda: scenes.image_grid(nearest=True)),
    ("image_grid_nearest_masked_wide", lambda: scenes.image_grid(nearest=True, masked=True, seed=52, width=2048, n=200)),
    ("opacity_grid", lambda: scenes.filter_grid(shader="opacity")),
    ("opacity_grid_masked", lambda: scenes.filter_grid(shader="opacity", masked=True, seed=74)),
    ("opacity_grid_wide", lambda: scenes.filter_grid(shader="opacity", width=2048, height=1024, n=160, seed=72)),
    ("opacity_grid_int", lambda: scenes.filter_grid(shader="opacity", width=1000, height=700, n=60, seed=73, fractional=False)),
    ("masked_rects_aa", lambda: scenes.masked_rects(force_aa=True, fractional=True)),
    ("masked_rects_rotated", lambda: scenes.masked_rects(rotate=True, seed=13)),
    ("masked_rects_rotated_aa_wide", lambda: scenes.masked_rects(rotate=True, force_aa=True, fractional=True, seed=14, width=2048, n=300)),
    ("image_repeat", lambda: scenes.image_repeat()),
    ("image_repeat_nearest", lambda: scenes.image_repeat(nearest=True)),
    ("image_repeat_wide", lambda: scenes.image_repeat(width=2048, height=1024, n=200, seed=58)),
    ("image_repeat_nearest_wide", lambda: scenes.image_repeat(width=2048, height=1024, n=200, seed=59, nearest=True)),
    ("image_repeat_opaque_int", lambda: scenes.image_repeat(width=1000, height=700, n=120, seed=60, translucent=False)),
    ("rotated_rects_quad", lambda: scenes.rotated_rects(encoding="quad")),
    ("rotated_rects_wide", lambda: scenes.rotated_rects(width=2048, height=1024, n=140, seed=97)),
    ("quad_masks", lambda: scenes.quad_masks()),
    ("quad_masks_wide", lambda: scenes.quad_masks(width=2048, height=1024, n=160, seed=82)),
    ("quad_masks_int", lambda: scenes.quad_masks(width=1000, height=700, n=70, seed=83, fractional=False)),
    ("gradient_grid", lambda: scenes.gradient_grid()),
    ("gradient_grid_wide", lambda: scenes.gradient_grid(width=2048, height=1024, n=300, seed=63)),
    ("gradient_grid_int", lambda: scenes.gradient_grid(width=1000, height=700, n=150, seed=65, fractional=False)),
    # brush_blend: every filter op except hue-rotate is bit-exact (swgl's pow() is its own float approximation,
    # restated); hue-rotate builds its matrix from libm's cosf / sinf in the vertex stage -> test below
    ("filter_grid_exact", lambda: scenes.filter_grid(ops=FILTER_OPS_EXACT)),
    ("filter_grid_exact_wide", lambda: scenes.filter_grid(width=2048, height=1024, n=160, seed=72, ops=FILTER_OPS_EXACT)),
    ("filter_grid_exact_int", lambda: scenes.filter_grid(width=1000, height=700, n=60, seed=73, fractional=False, ops=FILTER_OPS_EXACT)),
    ("aa_rects_brush", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=120, seed=7, fractional=True, encoding="brush", aa_edges=15)),
    ("aa_rects_quad", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=120, seed=7, fractional=True, encoding="quad", aa_edges=15)),
    ("aa_rects_brush_lr", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=120, seed=7, fractional=True, encoding="brush", aa_edges=5)),
    ("aa_rects_quad_tb", lambda: scenes.cfg2_overlapping_rects(width=1000, height=700, n=150, seed=3, fractional=True, encoding="quad", aa_edges=10)),
    ("empty", lambda: scenes.build_rect_frame(512, 512, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32),
                                               np.zeros(0, bool))),
]



BLUR_CASES = [
    ("blur_r8", dict(fmt="r8")),
    ("blur_rgba8", dict(fmt="rgba8")),
    ("blur_r8_sigmas", dict(fmt="r8", content=(40, 30), sigma=[0.8, 1.7, 3.2, 4.0], n_tasks=12, origin=(0, 0))),
    ("blur_rgba8_big", dict(fmt="rgba8", content=(61, 47), sigma=4.0, n_tasks=6, origin=(3, 2), atlas=512)),
    ("blur_r8_edges", dict(fmt="r8", content=(126, 62), sigma=[3.0, 0.0], n_tasks=4, origin=(0, 0), atlas=258, pattern="noise")),
    ("blur_rgba8_tiny", dict(fmt="rgba8", content=(5, 3), sigma=1.2, n_tasks=9, origin=(1, 1), atlas=64)),
    # cs_scale halvings in front of the blur (std deviation > 4 in the frame builder)
    ("blur_r8_scaled", dict(fmt="r8", scale_steps=2, content=(166, 140), sigma=2.5)),
    ("blur_rgba8_scaled", dict(fmt="rgba8", scale_steps=2, content=(150, 97), sigma=3.0)),
    ("blur_rgba8_scaled4", dict(fmt="rgba8", scale_steps=1, content=(64, 64), sigma=1.0, n_tasks=4, origin=(0, 0))),
]


@pytest.mark.parametrize("evaluation", ["span_rows", "bins"])
@pytest.mark.parametrize("name,kw", BLUR_CASES, ids=[c[0] for c in BLUR_CASES])
def test_hip_blur_matches_oracle(name, kw, evaluation, monkeypatch):
    """cs_blur vertical + horizontal passes on the GPU: 0 differing bytes (integer 8.8 taps; the coefficient's exp() is
    evaluated in fp64 and rounded once, like glibc's correctly rounded expf).  Both evaluations: a wave per target row
    (wr_span_rows_kernel, the default for cs_blur / cs_scale targets) and the bin raster (WRHIP_NO_SPAN_ROWS)."""
    if evaluation == "bins":
        monkeypatch.setenv("WRHIP_NO_SPAN_ROWS", "1")
    got, _ = render_direct(wrhip_lib(), scenes.blur_chain(**kw))
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.blur_chain(**kw))
        for k in want:
            d = np.abs(got[k].astype(int) - want[k].astype(int))
            assert d.max() == 0, (k, int(d.max()), int((d > 0).sum()))
    if name in GOLDEN:
        assert digest(got["blur_h"]) == GOLDEN[name], "golden digest mismatch"
    assert ref or name in GOLDEN


CLIP_CASES = [("clip_masks", dict()), ("clip_masks_dps", dict(dps=1.5, seed=32)), ("clip_masks_many", dict(n=60, seed=33))]


@pytest.mark.parametrize("evaluation", ["mask_rows", "in_raster"])
@pytest.mark.parametrize("name,kw", CLIP_CASES, ids=[c[0] for c in CLIP_CASES])
def test_hip_clip_rectangle_matches_oracle(name, kw, evaluation, monkeypatch):
    """cs_clip_rectangle masks on the GPU: 0 differing bytes against the oracle, and the committed digest."""
    if evaluation == "in_raster":      # the prims evaluated inside the bin raster instead of by wr_mask_rows_kernel (the fallback
        monkeypatch.setenv("WRHIP_NO_MASK_ROWS", "1")     # of a flush whose masks exceed the mask-row store)
    got, _ = render_direct(wrhip_lib(), scenes.clip_masks(**kw))
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.clip_masks(**kw))
        d = np.abs(got["clip_masks"].astype(int) - want["clip_masks"].astype(int))
        assert d.max() == 0, (int(d.max()), int((d > 0).sum()))
    if name in GOLDEN:
        assert digest(got["clip_masks"]) == GOLDEN[name]
    assert ref or name in GOLDEN


BOX_CASES = [("box_shadow_masks", dict()), ("box_shadow_masks_dps", dict(dps=1.5, seed=42)), ("box_shadow_masks_many", dict(n=40, seed=43))]


@pytest.mark.parametrize("evaluation", ["mask_rows", "in_raster"])
@pytest.mark.parametrize("name,kw", BOX_CASES, ids=[c[0] for c in BOX_CASES])
def test_hip_box_shadow_matches_oracle(name, kw, evaluation, monkeypatch):
    if evaluation == "in_raster":      # the prims evaluated inside the bin raster instead of by wr_mask_rows_kernel (the fallback
        monkeypatch.setenv("WRHIP_NO_MASK_ROWS", "1")     # of a flush whose masks exceed the mask-row store)
    got, _ = render_direct(wrhip_lib(), scenes.box_shadow_masks(**kw))
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.box_shadow_masks(**kw))
        d = np.abs(got["box_shadow_masks"].astype(int) - want["box_shadow_masks"].astype(int))
        assert d.max() == 0, (int(d.max()), int((d > 0).sum()))
    if name in GOLDEN:
        assert digest(got["box_shadow_masks"]) == GOLDEN[name]
    assert ref or name in GOLDEN


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(width=1024, height=1024), dict(dps=2.0)], ids=["small", "4k_dps2"])
def test_hip_chained_mask_levels_are_bit_exact(kw, monkeypatch):
    """WRHIP_CHAIN=1 (off by default: measured slower, profiles/r03_e_chain_ab.txt): the thin R8 levels of the box-shadow chain in
    one persistent launch with grid barriers give the bytes the separate launches give, in fewer launches, and no workgroup
    gives up at a barrier (that would raise GL_INVALID_OPERATION)."""
    monkeypatch.setenv("WRHIP_NO_SPAN_ROWS", "1")      # (the scale / blur levels as thin bin launches: what the chain is made of)
    want, st0 = render_direct(wrhip_lib(), scenes.cfg4_box_shadow(**kw))
    monkeypatch.setenv("WRHIP_CHAIN", "1")
    got, st1 = render_direct(wrhip_lib(), scenes.cfg4_box_shadow(**kw))
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert st1["gl_error"] == 0
    assert st1["kernel_launches"] < st0["kernel_launches"]


@pytest.mark.parametrize("name,kw", [("cfg4_small", dict(width=1024, height=1024)), ("cfg4_4k_dps2", dict(dps=2.0))],
                         ids=["small", "4k_dps2"])
def test_hip_cfg4_box_shadow_chain(name, kw):
    """BASELINE config 4 (box-shadow-large.yaml): the whole mask / scale / blur /
    box-shadow / masked-brush chain on the GPU."""
    got, _ = render_direct(wrhip_lib(), scenes.cfg4_box_shadow(**kw))
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.cfg4_box_shadow(**kw))
        for k in want:
            d = np.abs(got[k].astype(int) - want[k].astype(int))
            assert d.max() == 0, (k, int(d.max()), int((d > 0).sum()))
    assert digest(got["window"]) == GOLDEN[name]


@pytest.mark.parametrize("name,make", SMALL + OCCLUDED + BLEND + ROTATED, ids=[c[0] for c in SMALL + OCCLUDED + BLEND + ROTATED])
def test_hip_matches_oracle_small(name, make):
    got, stats = render_direct(wrhip_lib(), make())
    assert stats["raster_launches"] >= 1
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, make())
        assert np.array_equal(got, want)
    if name in GOLDEN and golden_applies(name):
        assert digest(got) == GOLDEN[name]
    assert ref or (name in GOLDEN and golden_applies(name))


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", SPLIT, ids=[c[0] for c in SPLIT])
def test_hip_split_composites_match_oracle(name, make):
    """(see tests/test_hostsim_parity.py::test_hostsim_split_composites_match_oracle)"""
    got, stats = render_direct(wrhip_lib(), make())
    assert stats["gl_error"] == 0
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, make())
        assert np.array_equal(got, want)
    if name in SPLIT_GOLDEN and name in GOLDEN:
        assert digest(got) == GOLDEN[name]
    assert ref or (name in SPLIT_GOLDEN and name in GOLDEN)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", GLYPH_TRANSFORM, ids=[c[0] for c in GLYPH_TRANSFORM])
def test_hip_glyph_transform_text_matches_oracle(name, make):
    """(see tests/test_hostsim_parity.py::test_hostsim_glyph_transform_text_matches_oracle)"""
    got, stats = render_direct(wrhip_lib(), make())
    assert stats["gl_error"] == 0
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, make())
        assert np.array_equal(got, want)
    if name in GOLDEN and golden_applies("cfg3"):
        assert digest(got) == GOLDEN[name]
    assert ref or (name in GOLDEN and golden_applies("cfg3"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", FLAT, ids=[c[0] for c in FLAT])
def test_hip_flattened_depth_rows_match_oracle(name, make):
    """(see tests/test_hostsim_parity.py::test_hostsim_flattened_depth_rows_match_oracle)"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, make())
    got, stats = render_direct(wrhip_lib(), make())
    assert np.array_equal(got, want)
    assert stats["gl_error"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", RUN_OVERFLOW, ids=[c[0] for c in RUN_OVERFLOW])
def test_hip_depth_run_overflow_is_exact(name, make, monkeypatch):
    """(see tests/test_hostsim_parity.py::test_hostsim_depth_run_overflow_is_exact)"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, make())
    got, stats = render_direct(wrhip_lib(), make())
    assert np.array_equal(got, want) and stats["gl_error"] == 0
    monkeypatch.setenv("WRHIP_NO_TILE_ROWS", "1")
    got, stats = render_direct(wrhip_lib(), make())
    assert np.array_equal(got, want) and stats["gl_error"] == 0
    monkeypatch.setenv("WRHIP_RUNS_POOL_WORDS", "0")
    got, stats = render_direct(wrhip_lib(), make())
    assert np.array_equal(got, want) or stats["gl_error"] == 0x0502
    assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 4


@pytest.mark.gpu
def test_hip_pool_grows_on_demand(monkeypatch):
    """(see tests/test_hostsim_parity.py::test_hostsim_pool_grows_on_demand)"""
    from test_hostsim_parity import _two_frames
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    make = RUN_OVERFLOW[2][1]
    want, _ = render_direct(ref, make())
    monkeypatch.setenv("WRHIP_RUNS_POOL_WORDS", "64")
    (px1, e1), (px2, e2) = _two_frames(wrhip_lib(), make)
    assert e1 == 0x0502
    assert e2 == 0 and np.array_equal(px2, want)


def _sweep_cases():
    """Randomised scenes that stress the prim-list walks (dense and thinly spread mask words, several 64-word blocks, the cell
    raster's and the LDS list walk's limits, the depth cap with opaque / translucent mixes) and the general-quad paths: the
    fixed-seed part of tools/sweep.py, sized for the driver's run (about 20 s on the MI355X)."""
    out = []
    for seed in range(2):
        for n, (w, h) in ((300, (512, 512)), (3000, (1024, 512)), (9000, (2048, 1024)), (30000, (2048, 2048))):
            for enc in ("quad", "brush"):
                out.append((f"rects_n{n}_{enc}_s{seed}", lambda n=n, w=w, h=h, enc=enc, seed=seed:
                            scenes.cfg5_many_rects(width=w, height=h, n=n, seed=100 + seed, encoding=enc)))
        # translucent-only and opaque-heavy mixes over a clear: cells where the bins are sparse, pixel / list walk where they are not
        out.append((f"overlap_small_s{seed}", lambda seed=seed: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=2500, seed=300 + seed)))
        out.append((f"overlap_frac_s{seed}", lambda seed=seed: scenes.cfg2_overlapping_rects(width=1536, height=1024, n=600, seed=310 + seed, fractional=True)))
        out.append((f"rotated_s{seed}", lambda seed=seed: scenes.add_occluders(scenes.rotated_rects(n=120, seed=500 + seed, opaque_frac=0.3), zmax=120, seed=seed)))
        out.append((f"persp_s{seed}", lambda seed=seed: scenes.add_occluders(scenes.rotated_rects(n=120, seed=600 + seed, perspective=True, opaque_frac=0.3), zmax=120, seed=seed)))
        out.append((f"persp_images_s{seed}", lambda seed=seed: scenes.add_occluders(scenes.rotated_images(n=80, seed=700 + seed, perspective="all"), zmax=80, seed=seed)))
        out.append((f"rot_images_s{seed}", lambda seed=seed: scenes.add_occluders(scenes.rotated_images(n=80, seed=800 + seed), zmax=80, seed=seed)))
        out.append((f"images_s{seed}", lambda seed=seed: scenes.add_occluders(scenes.image_grid(width=2048, height=1024, n=400, seed=900 + seed), n=150, zmax=430, seed=seed)))
        out.append((f"text_s{seed}", lambda seed=seed: scenes.add_occluders(scenes.cfg3_text(width=2048, height=1024, lines=50, glyphs_per_line=120, run_len=24, seed=seed), zmax=100, seed=seed)))
    return out


_SWEEP = _sweep_cases()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", _SWEEP, ids=[c[0] for c in _SWEEP])
def test_hip_randomised_sweep_matches_oracle(name, make):
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, make())
    got, stats = render_direct(wrhip_lib(), make())
    assert np.array_equal(got, want)
    assert stats["gl_error"] == 0


def _cache_key(scene):
    return "decoration_cache" if scene == "cache_decorations" else "border_cache"


_BORDER_CASES = ([(n, "border_solid", kw) for n, kw in BORDERS] + [(n, "border_segments", kw) for n, kw in BORDER_SEGMENTS] +
                 [(n, "cache_decorations", kw) for n, kw in DECORATIONS])


@pytest.mark.parametrize("name,scene,kw", _BORDER_CASES, ids=[c[0] for c in _BORDER_CASES])
def test_hip_border_solid_matches_oracle(name, scene, kw):
    """cs_border_solid / cs_border_segment tasks in the texture cache: 0 differing bytes and the committed digest.  The cached
    gradient / line-decoration scenes hold conic gradients whose angle comes from libm atan2f (OCML's on the device): a handful of
    pixels by 1 LSB there, nothing else."""
    got, _ = render_direct(wrhip_lib(), getattr(scenes, scene)(**kw))
    ref = oracle_ref()
    exact = scene != "cache_decorations"
    if ref:
        want, _ = render_direct(ref, getattr(scenes, scene)(**kw))
        d = np.abs(got[_cache_key(scene)].astype(int) - want[_cache_key(scene)].astype(int))
        if exact:
            assert d.max() == 0, (int(d.max()), int((d > 0).sum()))
        else:
            assert d.max() <= 1 and (d > 0).sum() <= 2e-5 * d.size, (int(d.max()), int((d > 0).sum()))
    if name in GOLDEN and exact:
        assert digest(got[_cache_key(scene)]) == GOLDEN[name]
    assert ref or name in GOLDEN


@pytest.mark.parametrize("encoding", ["quad", "brush"])
def test_hip_cfg2_full_4k(encoding):
    """BASELINE config 2 at full size against the committed golden digest (the
    oracle needs ~2 s per frame here, so it is also compared directly)."""
    got, _ = render_direct(wrhip_lib(), scenes.cfg2_overlapping_rects(encoding=encoding))
    assert digest(got) == GOLDEN[f"cfg2_4k_{encoding}"]
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.cfg2_overlapping_rects(encoding=encoding))
        assert np.array_equal(got, want)


def test_hip_cfg3_text_full_4k():
    """BASELINE config 3 at full size: ~67k glyph instances from the R8 atlas."""
    got, stats = render_direct(wrhip_lib(), scenes.cfg3_text())
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.cfg3_text())
        assert np.array_equal(got, want)
    if golden_applies("cfg3_4k"):
        assert digest(got) == GOLDEN["cfg3_4k"]
    assert ref or golden_applies("cfg3_4k")


def test_hip_cfg2_4k_fractional_vs_oracle():
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    got, _ = render_direct(wrhip_lib(), scenes.cfg2_overlapping_rects(fractional=True))
    want, _ = render_direct(ref, scenes.cfg2_overlapping_rects(fractional=True))
    assert np.array_equal(got, want)


def test_hip_cfg5_full_8k():
    """BASELINE config 5 at full size (100k rects, 50 % opaque, 7680x4320, 72 tiles + composite) against swgl directly (the oracle
    needs ~2.5 s for the frame) and against the committed digest; plus encoding independence (the brush encoding of the same
    display list gives the same pixels) and determinism across frames."""
    a, st = render_direct(wrhip_lib(), scenes.cfg5_many_rects(), frames=2)
    assert st["prims"] > 100_000
    assert digest(a) == GOLDEN["cfg5_8k"]
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.cfg5_many_rects())
        assert np.array_equal(a, want)
    b, _ = render_direct(wrhip_lib(), scenes.cfg5_many_rects(encoding="brush"))
    assert np.array_equal(a, b)
    assert (a[..., 3] == 255).all()


@pytest.mark.parametrize("name,workload,_small,kw", WRENCH, ids=[c[0] for c in WRENCH])
def test_hip_wrench_benchmarks_full_4k(name, workload, _small, kw):
    """wrench/benchmarks/*.yaml (wrench_scenes.py) at the 4K target, the bench workloads themselves: 0 differing bytes in every
    render target read back"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, scenes.make_workload(workload, **kw))
    got, st = render_direct(wrhip_lib(), scenes.make_workload(workload, **kw))
    assert st["gl_error"] == 0
    if isinstance(want, dict):
        for k in want:
            assert np.array_equal(got[k], want[k]), k
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name,kw", REPEAT_DUAL, ids=[c[0] for c in REPEAT_DUAL])
def test_hip_repeat_dual_source_images_match_oracle(name, kw):
    """brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION on the MI355X: 0 differing bytes"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, scenes.image_repeat(**kw))
    got, st = render_direct(wrhip_lib(), scenes.image_repeat(**kw))
    assert st["gl_error"] == 0
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,make", YUV, ids=[c[0] for c in YUV])
def test_hip_yuv_images_match_oracle(name, make):
    """brush_yuv_image (parity_cases.YUV) on the MI355X: 0 differing bytes"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, make())
    got, st = render_direct(wrhip_lib(), make())
    assert st["gl_error"] == 0 and (want != 255).any()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,make", TEXTURE_RECT, ids=[c[0] for c in TEXTURE_RECT])
def test_hip_texture_rect_keys_match_oracle(name, make):
    """The TEXTURE_RECT keys (parity_cases.TEXTURE_RECT) on the MI355X: 0 differing bytes in every target read back"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, make())
    got, st = render_direct(wrhip_lib(), make())
    assert st["gl_error"] == 0
    if isinstance(want, dict):
        for k in want:
            assert np.array_equal(got[k], want[k]), k
    else:
        assert (want != 255).any()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name,make", SVG_FILTERS, ids=[c[0] for c in SVG_FILTERS])
def test_hip_svg_filters_match_oracle(name, make):
    """cs_svg_filter / cs_svg_filter_node (parity_cases.SVG_FILTERS) on the MI355X: within 1 LSB where sqrt / division / libm powf
    enter (soft light, the un-premultiply, the node program's linearised flood colours), 0 differing bytes elsewhere"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, make())
    got, st = render_direct(wrhip_lib(), make())
    assert st["gl_error"] == 0
    for k in want:
        d = np.abs(got[k].astype(np.int32) - want[k].astype(np.int32))
        assert d.max() <= 1, (k, int(d.max()))
        assert (d > 0).sum() <= 0.001 * d.size, (k, int((d > 0).sum()))


def test_hip_staging_ring_wraps_keep_every_frame():
    """the staging ring's fence-based reuse under real asynchrony -- 40 different frames in flight through a 1 MB ring
    (wraps every 3-4 frames, held-back tail launches still reading the previous arena) equal the 96 MB ring's frames, a
    drain-per-lap run and the inline-submit run, byte for byte; the first five equal the oracle's."""
    from test_hostsim_parity import ring_wrap_digests
    base = ring_wrap_digests(wrhip_lib(), rounds=8)
    assert len(set(base)) == len(base) == 40
    assert ring_wrap_digests(wrhip_lib(), rounds=8, WRHIP_STAGING_BYTES=1 << 20) == base
    assert ring_wrap_digests(wrhip_lib(), rounds=8, WRHIP_STAGING_BYTES=1 << 20, WRHIP_RING_DRAIN=1) == base
    assert ring_wrap_digests(wrhip_lib(), rounds=8, WRHIP_STAGING_BYTES=3 << 19, WRHIP_NO_SUBMIT_THREAD=1) == base
    assert ring_wrap_digests(wrhip_lib(), rounds=8, WRHIP_STAGING_BYTES=1 << 20, WRHIP_NO_COPY_STREAM=1) == base
    ref = oracle_ref()
    if ref:
        assert ring_wrap_digests(ref, rounds=1) == base[:5]


def test_hip_vs_clang_oracle_are_a_bounded_deviation():
    ref = oracle_ref("clang")
    if not ref:
        pytest.skip("clang oracle not built")
    make = lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=300, seed=12, fractional=True)
    got, _ = render_direct(wrhip_lib(), make())
    want, _ = render_direct(ref, make())
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


def test_native_replay_and_launch_count():
    frame = scenes.cfg2_overlapping_rects(width=2048, height=1024, n=100, seed=9)
    rec, px = record_scene(wrhip_lib(), frame)
    p = ScenePlayer(wrhip_lib(), rec)
    ms = p.frames(3, 10)
    assert np.array_equal(p.read_pixels(), px)
    assert (ms > 0).all()


def test_sharded_player_single_rank_rccl():
    """The N>1 harness with world_size=1 on the real device: torch views libwrhip's
    framebuffer in HBM through __cuda_array_interface__ and RCCL all-gathers it."""
    import socket
    import torch
    import torch.distributed as dist
    from webrender_amd.dist import ShardedFramePlayer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        make = lambda: scenes.cfg2_overlapping_rects(width=2048, height=1080, n=150, seed=31)
        p = ShardedFramePlayer(wrhip_lib(), "custom", "quad", 0, 1, device="cuda", frame=make())
        p.frames(1, 3)
        p.stream(3)
        got = p.assembled()                       # raw BGRA bytes as stored in HBM
        want, _ = render_direct(wrhip_lib(), make())   # ReadPixels(GL_RGBA)
        assert np.array_equal(got[..., [2, 1, 0, 3]], want)
        # the native per-frame loop (what bench.py --gpus N runs): replay + WrhipFlush + in-place strip exchange in C
        for mode in ("root", "all"):
            pn = ShardedFramePlayer(wrhip_lib(), "custom", "quad", 0, 1, device="cuda", frame=make(), gather=mode, native="rccl")
            pn.frames(1, 2)
            pn.stream(4)
            assert np.array_equal(pn.assembled(), want)
            pn.close()
    finally:
        dist.destroy_process_group()


PIPELINED = [
    lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=200, seed=40),
    lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=220, seed=41, encoding="brush", fractional=True),
    lambda: scenes.masked_rects(),
    lambda: scenes.image_grid(),
    lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=180, seed=42),
    lambda: scenes.gradient_grid(),
    lambda: scenes.cfg3_text(width=1024, height=1024, lines=30, glyphs_per_line=60, run_len=12),
    lambda: scenes.masked_rects(fractional=True),
    lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=250, seed=43, encoding="brush"),
    # round 5, the flush's pool across frames: gradient tables copied by the setup stage while the previous frame's held-back launches
    # still read THEIR copies (two gradient frames in a row with different tables at the same GPU-buffer addresses), row tables of
    # rotated prims, depth runs that outgrow the LDS tables
    lambda: scenes.gradient_grid(seed=62),
    lambda: scenes.gradient_grid(rotate=True, seed=66),
    lambda: scenes.quad_gradients(),
    lambda: scenes.rotated_rects(opaque_frac=0.3),
    lambda: scenes.add_slivers(scenes.image_grid(), pitch=3),        # (the atlas of the image_grid frame above: the renderer mirror keeps static textures by name)
    lambda: scenes.gradient_grid(rotate=True, perspective=True, seed=67),
    lambda: scenes.gradient_grid(seed=63),
]
# (the renderer mirror keeps static textures by NAME: frames of one stream must agree on what a name holds)
def _static_names_agree():
    import hashlib
    seen = {}
    for m in PIPELINED:
        for ref in m().static_textures:
            h = hashlib.sha1(np.ascontiguousarray(ref.pixels).tobytes()).hexdigest() if ref.pixels is not None else None
            if seen.setdefault(ref.name, h) != h:
                return False
    return True



def test_hip_pipelined_frames_match_isolated_frames():
    """Different frames issued back to back with no Finish in between (host recording of frame
    k+1 overlapping the GPU work of frame k; data textures re-uploaded, per-frame textures recycled
    through the HBM pool) must each produce exactly what they produce when rendered alone."""
    from webrender_amd.harness import render_pipelined
    assert _static_names_agree()
    ref = oracle_ref()
    for rep in range(3):
        got = render_pipelined(wrhip_lib(), [m() for m in PIPELINED])
        for i, (g, m) in enumerate(zip(got, PIPELINED)):
            want, _ = render_direct(ref if (ref and rep == 0) else wrhip_lib(), m())
            assert np.array_equal(g[..., [2, 1, 0, 3]], want), f"frame {i} (rep {rep})"


def test_hip_filter_hue_rotate_are_a_bounded_deviation():
    """FILTER_HUE_ROTATE: the colour matrix comes from cos / sin of the angle (blend.glsl:49-58), libm
    on the reference's side and the device math library here -- the one brush_blend case that is
    held to the north-star tolerance (+-1 LSB) instead of 0."""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    make = lambda: scenes.filter_grid(ops=[2], n=48, seed=74)
    got, _ = render_direct(wrhip_lib(), make())
    want, _ = render_direct(ref, make())
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


def test_hip_filter_swatches_match_numpy_model():
    """brush_blend on the device against the independent numpy model of the filter math (no oracle in
    the loop): exact, except hue-rotate whose matrix depends on the device's cosf / sinf (<= 1 LSB)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle import check_filter_swatches
    fr = scenes.filter_swatches()
    px, _ = render_direct(wrhip_lib(), fr)
    check_filter_swatches(px, fr, tol_hue=1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", COPIES, ids=[c[0] for c in COPIES])
def test_hip_texture_cache_copies(name, kw):
    """ps_copy: RGBA8 and R8 texture-cache copies (one pass feeding the next) equal the copies done in numpy, and swgl's."""
    fr = scenes.texture_cache_copies(**kw)
    want = copies_expected(fr)
    got, st = render_direct(wrhip_lib(), scenes.texture_cache_copies(**kw))
    assert st["gl_error"] == 0
    ref = oracle_ref()
    for k, v in want.items():
        assert v.any() and np.array_equal(got[k], v), k
    if ref:
        sw, _ = render_direct(ref, scenes.texture_cache_copies(**kw))
        for k, v in want.items():
            assert np.array_equal(sw[k], v), k


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,kw", MIX_BLEND, ids=[c[0] for c in MIX_BLEND])
def test_hip_mix_blend_matches_oracle(name, scene, kw):
    """brush_mix_blend on the MI355X: +-1 LSB allowed where the device's sqrt / division enter (soft light, the
    non-separable modes); the swatches are also held to the numpy model of the GLSL."""
    got, st = render_direct(wrhip_lib(), getattr(scenes, scene)(**kw))
    assert st["gl_error"] == 0 and (got != 255).any()
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, getattr(scenes, scene)(**kw))
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1 and (d > 0).sum() <= 1e-4 * d.size, (int(d.max()), int((d > 0).sum()))
    assert ref or name in GOLDEN
    if not ref:
        assert digest(got) == GOLDEN[name]
    if scene == "mix_blend_swatches":
        import sys
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import np_model
        fr = scenes.mix_blend_swatches()
        H = got.shape[0]
        for (x, y, ib, isrc, mode) in fr.swatches:
            h, w = ib.shape[:2]
            want = np_model.mix_blend_swatch(ib[..., [2, 1, 0, 3]], isrc[..., [2, 1, 0, 3]], mode)
            d = np.abs(want.astype(int) - got[H - y - h:H - y][::-1, x:x + w].astype(int))
            assert d.max() <= 1, (mode, int(d.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", DUAL_SOURCE, ids=[c[0] for c in DUAL_SOURCE])
def test_hip_dual_source_images_match_oracle(name, kw):
    """brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING under the dual-source blend state on the MI355X"""
    got, st = render_direct(wrhip_lib(), scenes.image_grid(**kw))
    assert st["gl_error"] == 0 and (got != 255).any()
    ref = oracle_ref()
    if ref:
        want, _ = render_direct(ref, scenes.image_grid(**kw))
        assert np.array_equal(got, want)
    if name in GOLDEN:
        assert digest(got) == GOLDEN[name]
    assert ref or name in GOLDEN


def _same(a, b):
    if isinstance(a, dict):
        return set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
    return np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", TILE_ROWS, ids=[c[0] for c in TILE_ROWS])
def test_hip_tile_rows_match_oracle(name, make, monkeypatch):
    """picture targets of a few large gradient / image prims: the row kernel (default) and the bin raster give the oracle's bytes"""
    ref = oracle_ref()
    got, st = render_direct(wrhip_lib(), make())
    assert st["row_launches"] >= 1, "the case was meant for wr_tile_rows_kernel"
    monkeypatch.setenv("WRHIP_NO_TILE_ROWS", "1")
    got2, st2 = render_direct(wrhip_lib(), make())
    assert st2["row_launches"] == 0
    assert _same(got, got2)
    assert st["gl_error"] == st2["gl_error"]
    if ref:
        want, _ = render_direct(ref, make())
        assert _same(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name,_small,kw", TEXT_REFTESTS, ids=[c[0] for c in TEXT_REFTESTS])
def test_hip_text_reftests_full_4k(name, _small, kw):
    """BASELINE configs[2]: wrench/reftests/text/<name>.yaml at the 4K target over FreeType-rasterised glyphs of the reftest's own font"""
    ref = oracle_ref()
    if not ref:
        pytest.skip("oracle not built")
    want, _ = render_direct(ref, scenes.make_workload("reftest-text-" + name, **kw))
    got, st = render_direct(wrhip_lib(), scenes.make_workload("reftest-text-" + name, **kw))
    assert st["gl_error"] == 0
    if isinstance(want, dict):
        for k in want:
            assert np.array_equal(got[k], want[k]), k
    else:
        assert np.array_equal(got, want)
