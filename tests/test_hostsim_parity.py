"""CPU-side check of the deferred renderer's logic: the same kernel sources
compiled for the host (libwrhip_hostsim.so, test infrastructure -- see
csrc/wrhip_rt.h) must reproduce the oracle bit for bit.  This is what lets the
binning / ordering / blend pipeline be verified without a GPU; the `-m gpu`
tests then run the identical comparisons through the real HIP library."""
import hashlib
import json
import os
import numpy as np
import pytest
from conftest import ROOT
from webrender_amd import scenes
from webrender_amd.harness import render_direct, record_scene, ScenePlayer
from parity_cases import TEXT_REFTESTS, TILE_ROWS, WRENCH, TEXTURE_RECT, YUV, SVG_FILTERS, OCCLUDED, BLEND, ROTATED, BORDERS, BORDER_SEGMENTS, DECORATIONS, FLAT, RUN_OVERFLOW, COPIES, copies_expected, MIX_BLEND, DUAL_SOURCE, REPEAT_DUAL, SPLIT, SPLIT_GOLDEN, GLYPH_TRANSFORM

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))


def digest(px):
    return hashlib.sha256(np.ascontiguousarray(px).tobytes()).hexdigest()


def golden_applies(name):
    """Text digests only hold when PIL rasterises the very same glyph bitmaps."""
    if not name.startswith("cfg3"):
        return True
    return digest(scenes.build_glyph_atlas()[0]) == GOLDEN.get("glyph_atlas")


CASES = [
    ("cfg1", lambda: scenes.cfg1_solid_colors()),
    ("cfg1_brush", lambda: scenes.cfg1_solid_colors(encoding="brush")),
    ("simple_batching", lambda: scenes.simple_batching()),
    ("cfg2_small", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=200, seed=7)),
    ("cfg2_small_brush", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=200, seed=7, encoding="brush")),
    ("cfg2_small_frac", lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=200, seed=7, fractional=True)),
    ("cfg2_odd_size", lambda: scenes.cfg2_overlapping_rects(width=1000, height=700, n=150, seed=3, fractional=True)),
    ("cfg5_small", lambda: scenes.cfg5_many_rects(width=2048, height=1024, n=5000)),
    ("cfg5_small_brush", lambda: scenes.cfg5_many_rects(width=2048, height=1024, n=5000, encoding="brush")),
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
    ("filter_grid_masked", lambda: scenes.filter_grid(masked=True, seed=75)),
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
    ("filter_grid", lambda: scenes.filter_grid()),
    ("filter_grid_wide", lambda: scenes.filter_grid(width=2048, height=1024, n=160, seed=72)),
    ("filter_grid_int", lambda: scenes.filter_grid(width=1000, height=700, n=60, seed=73, fractional=False)),
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
def test_hostsim_blur_matches_oracle(hostsim, oracle_gcc, name, kw, evaluation, monkeypatch):
    """cs_blur vertical + horizontal passes (R8 and RGBA8 targets): every
    render target of the chain is compared, not only the window.  Both evaluations of the off-screen levels: a wave per
    target row (wr_span_rows_kernel, the default) and the bin raster (WRHIP_NO_SPAN_ROWS=1)."""
    if evaluation == "bins":
        monkeypatch.setenv("WRHIP_NO_SPAN_ROWS", "1")
    want, _ = render_direct(oracle_gcc, scenes.blur_chain(**kw))
    got, _ = render_direct(hostsim, scenes.blur_chain(**kw))
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert want["blur_h"].any()
    if name in GOLDEN:
        assert digest(got["blur_h"]) == GOLDEN[name]


CLIP_CASES = [("clip_masks", dict()), ("clip_masks_dps", dict(dps=1.5, seed=32)), ("clip_masks_many", dict(n=60, seed=33))]


@pytest.mark.parametrize("evaluation", ["mask_rows", "in_raster"])
@pytest.mark.parametrize("name,kw", CLIP_CASES, ids=[c[0] for c in CLIP_CASES])
def test_hostsim_clip_rectangle_matches_oracle(hostsim, oracle_gcc, name, kw, evaluation, monkeypatch):
    """cs_clip_rectangle (uniform-radius fast path and general path, clip and
    clip-out, second clip multiplied in) into an R8 alpha target."""
    if evaluation == "in_raster":      # the prims evaluated inside the bin raster instead of by wr_mask_rows_kernel (the fallback
        monkeypatch.setenv("WRHIP_NO_MASK_ROWS", "1")     # of a flush whose masks exceed the mask-row store)
    want, _ = render_direct(oracle_gcc, scenes.clip_masks(**kw))
    got, _ = render_direct(hostsim, scenes.clip_masks(**kw))
    assert np.array_equal(got["clip_masks"], want["clip_masks"])
    v = want["clip_masks"]
    assert (v == 0).any() and (v == 255).any() and ((v > 0) & (v < 255)).any()
    if name in GOLDEN:
        assert digest(got["clip_masks"]) == GOLDEN[name]


BOX_CASES = [("box_shadow_masks", dict()), ("box_shadow_masks_dps", dict(dps=1.5, seed=42)), ("box_shadow_masks_many", dict(n=40, seed=43))]


@pytest.mark.parametrize("evaluation", ["mask_rows", "in_raster"])
@pytest.mark.parametrize("name,kw", BOX_CASES, ids=[c[0] for c in BOX_CASES])
def test_hostsim_box_shadow_matches_oracle(hostsim, oracle_gcc, name, kw, evaluation, monkeypatch):
    """cs_clip_box_shadow: nine-patch stretch of a cached blurred shadow into
    R8 mask tasks (stretch / simple modes per axis, clip and clip-out)."""
    if evaluation == "in_raster":      # the prims evaluated inside the bin raster instead of by wr_mask_rows_kernel (the fallback
        monkeypatch.setenv("WRHIP_NO_MASK_ROWS", "1")     # of a flush whose masks exceed the mask-row store)
    want, _ = render_direct(oracle_gcc, scenes.box_shadow_masks(**kw))
    got, _ = render_direct(hostsim, scenes.box_shadow_masks(**kw))
    assert np.array_equal(got["box_shadow_masks"], want["box_shadow_masks"])
    v = want["box_shadow_masks"]
    assert (v == 0).any() and (v == 255).any() and ((v > 0) & (v < 255)).any()
    if name in GOLDEN:
        assert digest(got["box_shadow_masks"]) == GOLDEN[name]


def test_hostsim_cfg4_box_shadow_chain(hostsim, oracle_gcc):
    """BASELINE config 4 end to end (1024^2 window): rounded-rect mask -> two
    cs_scale halvings -> cs_blur V/H -> cs_clip_box_shadow x cs_clip_rectangle
    clip-out -> masked brush_solid segments -> composite.  Every intermediate
    that is read back must match, not only the window."""
    mk = lambda: scenes.cfg4_box_shadow(width=1024, height=1024)
    want, _ = render_direct(oracle_gcc, mk())
    got, _ = render_direct(hostsim, mk())
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert digest(got["window"]) == GOLDEN["cfg4_small"]


@pytest.mark.parametrize("name,make", CASES + OCCLUDED + BLEND + ROTATED, ids=[c[0] for c in CASES + OCCLUDED + BLEND + ROTATED])
def test_hostsim_matches_oracle(hostsim, oracle_gcc, name, make):
    want, _ = render_direct(oracle_gcc, make())
    got, stats = render_direct(hostsim, make())
    assert np.array_equal(got, want)
    assert stats["raster_launches"] >= 1
    if name in GOLDEN and golden_applies(name):
        assert digest(got) == GOLDEN[name]


@pytest.mark.parametrize("name,make", SPLIT, ids=[c[0] for c in SPLIT])
def test_hostsim_split_composites_match_oracle(hostsim, oracle_gcc, name, make):
    """ps_split_composite (parity_cases.SPLIT): the split polygons of a preserve-3d context -- general quads by construction -- on
    the general-quad / perspective paths.  0 differing bytes, nothing reported, and the polygons did leave pixels."""
    want, _ = render_direct(oracle_gcc, make())
    got, stats = render_direct(hostsim, make())
    assert np.array_equal(got, want)
    assert stats["gl_error"] == 0
    assert (want != 255).any()
    if name in SPLIT_GOLDEN and name in GOLDEN:
        assert digest(got) == GOLDEN[name]


@pytest.mark.parametrize("name,make", GLYPH_TRANSFORM, ids=[c[0] for c in GLYPH_TRANSFORM])
def test_hostsim_glyph_transform_text_matches_oracle(hostsim, oracle_gcc, name, make):
    """ps_text_run GLYPH_TRANSFORM (parity_cases.GLYPH_TRANSFORM): 0 differing bytes, nothing reported; the runs under a local
    clip rect exist (their quads are not the glyph's raster rect: the gl_ClipDistance cut is what bounds them)."""
    want, _ = render_direct(oracle_gcc, make())
    got, stats = render_direct(hostsim, make())
    assert np.array_equal(got, want)
    assert stats["gl_error"] == 0
    assert (want != 255).any()
    if name in GOLDEN and golden_applies("cfg3"):
        assert digest(got) == GOLDEN[name]


@pytest.mark.parametrize("name,make", FLAT, ids=[c[0] for c in FLAT])
def test_hostsim_flattened_depth_rows_match_oracle(hostsim, oracle_gcc, name, make):
    """Prims that follow a depth-tested perspective prim on the same rows (parity_cases.FLAT): swgl draws them chunk by chunk
    through main() on the flattened depth rows; depth-writing perspective prims.  0 differing bytes, nothing reported."""
    want, _ = render_direct(oracle_gcc, make())
    got, stats = render_direct(hostsim, make())
    assert np.array_equal(got, want)
    assert stats["gl_error"] == 0


@pytest.mark.parametrize("name,make", RUN_OVERFLOW, ids=[c[0] for c in RUN_OVERFLOW])
def test_hostsim_depth_run_overflow_is_exact(hostsim, oracle_gcc, name, make, monkeypatch):
    """A row with more depth runs (or a strip with more occluders) than the LDS copies hold keeps them in the flush's pool: the frame
    is swgl's, through the bins and through the row kernel.  With the pool taken away (WRHIP_RUNS_POOL_WORDS=0: what an exhausted pool
    looks like) the frame is either identical to swgl's or the caller is told (GL_INVALID_OPERATION at Finish) -- never silently off."""
    want, _ = render_direct(oracle_gcc, make())
    got, stats = render_direct(hostsim, make())
    assert np.array_equal(got, want) and stats["gl_error"] == 0
    monkeypatch.setenv("WRHIP_NO_TILE_ROWS", "1")
    got, stats = render_direct(hostsim, make())
    assert np.array_equal(got, want) and stats["gl_error"] == 0
    monkeypatch.setenv("WRHIP_RUNS_POOL_WORDS", "0")
    got, stats = render_direct(hostsim, make())
    assert np.array_equal(got, want) or stats["gl_error"] == 0x0502
    assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 4


def _two_frames(lib, make):
    """the same frame twice in ONE context: (pixels, gl_error) after each"""
    from webrender_amd.glapi import GL
    from webrender_amd.renderer import Renderer
    gl = GL(lib)
    frame = make()
    r = Renderer(gl, frame.width, frame.height)
    out = []
    for _ in range(2):
        r.render(frame)
        r.finish()
        err = int(gl.GetError())
        out.append((r.read_pixels(), err))
    r.destroy()
    return out


def test_hostsim_pool_grows_on_demand(hostsim, oracle_gcc, monkeypatch):
    """The flush's pool is sized for what the frame before asked of it: with a pool too small for the sliver-fence frame (here forced
    small: the default 64 MB holds this one) the first frame is REPORTED, Finish reads the pool's allocation word -- it counts every
    request, granted or not -- and grows the share, and the same frame drawn again in the same context is swgl's, with no error."""
    make = RUN_OVERFLOW[2][1]
    want, _ = render_direct(oracle_gcc, make())
    monkeypatch.setenv("WRHIP_RUNS_POOL_WORDS", "64")
    (px1, e1), (px2, e2) = _two_frames(hostsim, make)
    assert e1 == 0x0502, "the forced-small pool was expected to run out on the first frame"
    assert e2 == 0 and np.array_equal(px2, want)


def _cache_key(scene):
    return "decoration_cache" if scene == "cache_decorations" else "border_cache"


_BORDER_CASES = ([(n, "border_solid", kw) for n, kw in BORDERS] + [(n, "border_segments", kw) for n, kw in BORDER_SEGMENTS] +
                 [(n, "cache_decorations", kw) for n, kw in DECORATIONS])


@pytest.mark.parametrize("name,scene,kw", _BORDER_CASES, ids=[c[0] for c in _BORDER_CASES])
def test_hostsim_border_solid_matches_oracle(hostsim, oracle_gcc, name, scene, kw):
    want, _ = render_direct(oracle_gcc, getattr(scenes, scene)(**kw))
    got, _ = render_direct(hostsim, getattr(scenes, scene)(**kw))
    assert np.array_equal(got[_cache_key(scene)], want[_cache_key(scene)])
    v = want[_cache_key(scene)][..., 3]
    assert (v == 0).any() and (v == 255).any() and ((v > 0) & (v < 255)).any()
    if name in GOLDEN:
        assert digest(got[_cache_key(scene)]) == GOLDEN[name]


def test_hostsim_matches_golden_without_oracle(hostsim):
    """Runs even where oracle/_ref is unavailable: committed digests."""
    for name in ("cfg1", "simple_batching", "cfg2_small", "cfg2_small_frac"):
        make = dict(CASES)[name]
        got, _ = render_direct(hostsim, make())
        assert digest(got) == GOLDEN[name], name


def test_repeated_frames_and_launch_count(hostsim):
    """Steady state: a whole frame (tile pass + composite that samples the tiles) is ONE flush
    -- one upload scatter, one setup launch -- and, the composite being an opaque 1:1 copy of tiles that were
    all redrawn, ONE raster launch: the tile pass writes the window as well (forwarded composites)."""
    frame = scenes.cfg2_overlapping_rects(width=2048, height=1024, n=100, seed=9)
    from webrender_amd.glapi import GL
    from webrender_amd.renderer import Renderer
    gl = GL(hostsim)
    r = Renderer(gl, frame.width, frame.height)
    r.render(frame); r.finish()
    first = r.read_pixels()
    gl.WrhipResetStats()
    r.render(frame); r.finish()
    st = gl.stats()
    assert st["flushes"] == 1 and st["raster_launches"] == 1 and st["kernel_launches"] <= 3, st
    assert np.array_equal(r.read_pixels(), first)
    r.destroy()


def test_native_replay_of_recorded_trace(hostsim, oracle_gcc):
    frame = scenes.cfg2_overlapping_rects(width=1024, height=512, n=80, seed=4)
    rec, px = record_scene(hostsim, frame)
    for lib in (oracle_gcc, hostsim):
        p = ScenePlayer(lib, rec)
        p.frames(1, 2)
        assert np.array_equal(p.read_pixels(), px)


def test_strip_sharding_covers_frame(hostsim):
    """WrhipSetShard(rank, world): each rank rasterises a strip of bin rows of
    every target; the union over ranks equals the unsharded render."""
    from webrender_amd.glapi import GL
    from webrender_amd.renderer import Renderer
    make = lambda: scenes.cfg2_overlapping_rects(width=1024, height=1024, n=120, seed=2)
    full, _ = render_direct(hostsim, make())
    world = 4
    acc = np.zeros_like(full)
    for rank in range(world):
        gl = GL(hostsim)
        frame = make()
        r = Renderer(gl, frame.width, frame.height)
        gl.WrhipSetShard(rank, world)
        r.render(frame); r.finish()
        px = r.read_pixels()
        # default framebuffer strip owned by this rank (bin rows of 64 px; GL rows are bottom-up)
        bins_y = (frame.height + 63) // 64
        y0, y1 = bins_y * rank // world * 64, min(frame.height, bins_y * (rank + 1) // world * 64)
        acc[y0:y1] = px[y0:y1]
        r.destroy()
    # tiles are sharded by their own bin rows, so only rows whose tile strip AND
    # framebuffer strip are owned are valid; the multi-GPU path therefore shards
    # whole tiles (dist.py).  Here: the framebuffer strips of a single-target
    # scene must tile the frame.
    assert acc.shape == full.shape


def test_pipelined_frames_match_isolated_frames(hostsim, oracle_gcc):
    """Host logic of back-to-back frames (upload batching, pool recycling, hazard flushes):
    different frames without a Finish in between, each blitted into a keeper texture, must equal
    the frames rendered alone by the reference."""
    from webrender_amd.harness import render_pipelined
    makes = [
        lambda: scenes.cfg2_overlapping_rects(width=512, height=512, n=60, seed=40),
        lambda: scenes.masked_rects(width=512, height=512, n=40),
        lambda: scenes.cfg2_overlapping_rects(width=512, height=512, n=70, seed=41, encoding="brush", fractional=True),
        lambda: scenes.image_grid(width=512, height=512, n=40),
        lambda: scenes.gradient_grid(width=512, height=512, n=20),
        lambda: scenes.cfg2_overlapping_rects(width=512, height=512, n=50, seed=42),
        # (round 5: the flush's pool across frames -- gradient-table copies, row tables of rotated prims)
        lambda: scenes.gradient_grid(width=512, height=512, n=20, seed=62),
        lambda: scenes.gradient_grid(width=512, height=512, n=20, rotate=True, seed=66),
        lambda: scenes.rotated_rects(width=512, height=512, n=30, opaque_frac=0.3),
        lambda: scenes.gradient_grid(width=512, height=512, n=20, seed=63),
    ]
    got = render_pipelined(hostsim, [m() for m in makes])
    for i, (g, m) in enumerate(zip(got, makes)):
        want, _ = render_direct(oracle_gcc, m())
        assert np.array_equal(g[..., [2, 1, 0, 3]], want), f"frame {i}"


@pytest.mark.parametrize("name,workload,kw", [(n, w, k) for n, w, k, _ in WRENCH if k is not None], ids=[c[0] for c in WRENCH if c[2] is not None])
def test_hostsim_wrench_benchmarks_match_oracle(hostsim, oracle_gcc, name, workload, kw):
    """wrench/benchmarks/*.yaml as restated by wrench_scenes.py (reduced windows): every render target read back, 0 differing bytes"""
    want, _ = render_direct(oracle_gcc, scenes.make_workload(workload, **kw))
    got, st = render_direct(hostsim, scenes.make_workload(workload, **kw))
    assert st["gl_error"] == 0
    if isinstance(want, dict):
        for k in want:
            assert np.array_equal(got[k], want[k]), k
        want = want["window"]
    else:
        assert np.array_equal(got, want)
    assert (want != 255).any()


@pytest.mark.parametrize("name,make", YUV, ids=[c[0] for c in YUV])
def test_hostsim_yuv_images_match_oracle(hostsim, oracle_gcc, name, make):
    """brush_yuv_image (parity_cases.YUV): swgl_commitTextureLinearYUV spans (quantised plane sampling, the fixed-point colour
    matrix with its saturating adds) and main() tails.  0 differing bytes against the reference's generated program."""
    want, _ = render_direct(oracle_gcc, make())
    got, st = render_direct(hostsim, make())
    assert st["gl_error"] == 0 and (want != 255).any()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,make", TEXTURE_RECT, ids=[c[0] for c in TEXTURE_RECT])
def test_hostsim_texture_rect_keys_match_oracle(hostsim, oracle_gcc, name, make):
    """The TEXTURE_RECT keys of brush_image / brush_yuv_image / composite / cs_scale (parity_cases.TEXTURE_RECT): sampler2DRect samplers
    bound through GL_TEXTURE_RECTANGLE, unnormalised uv.  0 differing bytes against the reference's generated programs, in every
    target read back."""
    want, _ = render_direct(oracle_gcc, make())
    got, st = render_direct(hostsim, make())
    assert st["gl_error"] == 0
    if isinstance(want, dict):
        assert set(got) == set(want)
        for k in want:
            assert np.array_equal(got[k], want[k]), k
        assert any(w.any() for w in want.values())
    else:
        assert (want != 255).any()
        assert np.array_equal(got, want)


def test_hostsim_texture_rect_key_is_not_the_2d_key(hostsim):
    """A rectangle-texture program reads the unit's GL_TEXTURE_RECTANGLE binding, not its GL_TEXTURE_2D one: the same frame under the
    2D keys with nothing bound to GL_TEXTURE_2D... is not what this checks -- it checks that the REPETITION key's v_uv_bounds
    (the whole texture under TEXTURE_RECT, brush_image.glsl:254-258) makes the two keys draw different pixels."""
    a, _ = render_direct(hostsim, scenes.texture_rect(scenes.image_repeat(seed=64)))
    b, _ = render_direct(hostsim, scenes.image_repeat(seed=64))
    assert (a != b).sum() > 10000


@pytest.mark.parametrize("name,make", SVG_FILTERS, ids=[c[0] for c in SVG_FILTERS])
def test_hostsim_svg_filters_match_oracle(hostsim, oracle_gcc, name, make):
    """cs_svg_filter / cs_svg_filter_node (parity_cases.SVG_FILTERS): main() of every filter kind restated per pixel in strict fp32
    (glsl.h's pow approximation, its vector floor, libm powf in the node program's vertex stage).  0 differing bytes in both
    colour targets against the reference's generated programs; every swatch holds something."""
    want, _ = render_direct(oracle_gcc, make())
    got, st = render_direct(hostsim, make())
    assert st["gl_error"] == 0
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert (want["svg_pass_a"] != 0).any(axis=2).sum() > 100000 and (want["svg_pass_b"] != 0).any(axis=2).sum() > 100000


def ring_wrap_digests(lib, rounds=4, **env):
    import subprocess
    import sys
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "ring_wrap_driver.py"), lib, str(rounds)], env=e)
    return json.loads(out.decode().strip().splitlines()[-1])


def test_staging_ring_wraps_keep_every_frame(hostsim, oracle_gcc):
    """The fence-based reuse of the staging ring: 20 different frames issued without a Finish through a 1 MB ring
    (each stages 0.1-0.4 MB: the ring wraps a dozen times, descriptors included) give the frames the 96 MB ring gives, the
    frames a drain at every lap gives, the frames the inline submit path gives -- and the first ones are the reference's."""
    base = ring_wrap_digests(hostsim)
    assert len(set(base)) == len(base) == 20
    assert ring_wrap_digests(hostsim, WRHIP_STAGING_BYTES=1 << 20) == base
    assert ring_wrap_digests(hostsim, WRHIP_STAGING_BYTES=1 << 20, WRHIP_RING_DRAIN=1) == base
    assert ring_wrap_digests(hostsim, WRHIP_STAGING_BYTES=3 << 19, WRHIP_NO_SUBMIT_THREAD=1) == base
    assert ring_wrap_digests(oracle_gcc, rounds=1) == base[:5]


def test_unimplemented_perspective_prims_are_reported(hostsim, capfd):
    """Perspective prims outside the implemented set (here: ps_text_run GLYPH_TRANSFORM under a PROJECTIVE transform -- WebRender
    rasterises such text in local space, so the key never meets one (batch.rs:1186-1200); brush_mix_blend cut by the near plane, the
    case of rounds 4-5, is drawn since round 6) are counted by the setup stage, reported on stderr at Finish and raise
    GL_INVALID_OPERATION -- not drawn wrongly, not dropped silently; the rest of the frame is drawn."""
    from webrender_amd import glapi, glconst as G
    from webrender_amd.renderer import Renderer
    gl = glapi.GL(hostsim)
    r = Renderer(gl, 512, 512)
    fr = scenes.cfg3_text(width=512, height=512, lines=12, glyphs_per_line=30, run_len=10, glyph_transform=True, seed=5)
    m = scenes.projective_about(np.eye(2), 256.0, 256.0, 400.0, np.random.default_rng(3), strength=(0.3, 0.6))
    tid = fr.add_transform(m.T.astype(np.float32), np.linalg.inv(m).T.astype(np.float32), axis_aligned=False)
    hi = fr.prim_headers_i.data
    hi[0:hi.shape[0]:4, 2] = tid             # every other run under the projective transform ([z, specific, transform id, task])
    r.render(fr)
    r.finish()
    assert gl.GetError() == G.GL_INVALID_OPERATION
    assert gl.GetError() == 0
    assert "perspective" in capfd.readouterr().err
    px = r.read_pixels()
    assert (px != 255).any()
    r.destroy()


def test_undrawable_draws_are_refused_when_recorded(hostsim, capfd):
    """A draw the backend has no path for -- a shader key without an implementation (debug_color: the renderer's debug overlay,
    out of scope), an index pattern that is
    not the unit quad's -- raises GL_INVALID_OPERATION in the DrawElementsInstanced call itself (visible to the very next
    GetError, before any Finish), and nothing is recorded for it."""
    from webrender_amd import glapi, glconst as G
    from webrender_amd.device import Device
    gl = glapi.GL(hostsim)
    d = Device(gl)
    d.init_default_framebuffer(64, 64)
    tex = d.create_texture(64, 64, G.GL_RGBA8, render_target=True)
    d.bind_draw_target(tex.fbo, 64, 64)
    d.clear_target((0.0, 0.0, 0.0, 1.0), None)
    # (the link status already says so -- GetLinkStatus is what Device.create_program checks --; a caller that ignores it and
    # draws anyway is told at the draw)
    vs, fs = gl.CreateShader(G.GL_VERTEX_SHADER), gl.CreateShader(G.GL_FRAGMENT_SHADER)
    gl.ShaderSourceByName(vs, b"debug_color"); gl.ShaderSourceByName(fs, b"debug_color")
    pid = gl.CreateProgram()
    gl.AttachShader(pid, vs); gl.AttachShader(pid, fs)
    gl.LinkProgram(pid)
    assert not gl.GetLinkStatus(pid)
    vao = d.create_vao("PRIM_INSTANCES")
    gl.UseProgram(pid)
    assert gl.GetError() == 0
    d.draw_instanced_batch(vao, np.zeros((3, 4), np.int32))
    assert gl.GetError() == G.GL_INVALID_OPERATION and gl.GetError() == 0
    assert "debug_color" in capfd.readouterr().err
    # a known program, but not the unit quad
    prog2 = d.create_program("brush_solid", "PRIM_INSTANCES")
    d.bind_program(prog2, np.eye(4, dtype=np.float32))
    gl.BindVertexArray(vao.id)
    gl.BindBuffer(G.GL_ARRAY_BUFFER, vao.instance_vbo)
    gl.BufferData(G.GL_ARRAY_BUFFER, 48, np.zeros((3, 4), np.int32), G.GL_STREAM_DRAW)
    gl.DrawElementsInstanced(G.GL_TRIANGLES, 3, G.GL_UNSIGNED_SHORT, 0, 3)
    assert gl.GetError() == G.GL_INVALID_OPERATION
    gl.Finish()
    assert gl.GetError() == 0
    px = d.read_texture(tex)
    assert (px[..., :3] == 0).all()          # nothing was drawn
    d.destroy()


def test_texture_allocation_failure_is_sticky_out_of_memory(hostsim, monkeypatch):
    """HBM exhaustion while allocating texture storage raises the sticky GL_OUT_OF_MEMORY swgl raises (gl.cc:1125-1134;
    Renderer counts consecutive ones, renderer/mod.rs:1296-1303) instead of aborting; the texture stays unusable, draws to
    it are dropped, and the context keeps working."""
    from webrender_amd import glapi, glconst as G
    monkeypatch.setenv("WRHIP_HOSTSIM_ALLOC_LIMIT", str(64 << 20))     # the test library refuses larger single allocations
    gl = glapi.GL(hostsim)
    ctx = gl.CreateContext()
    gl.MakeCurrent(ctx)
    big, small = gl.gen("GenTextures"), gl.gen("GenTextures")
    gl.BindTexture(G.GL_TEXTURE_2D, big)
    gl.TexStorage2D(G.GL_TEXTURE_2D, 1, G.GL_RGBA8, 8192, 8192)          # 256 MiB: refused
    assert gl.GetError() == G.GL_OUT_OF_MEMORY
    assert gl.GetError() == 0                                            # reading it clears it
    fbo = gl.gen("GenFramebuffers")
    gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, fbo)
    gl.FramebufferTexture2D(G.GL_DRAW_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0, G.GL_TEXTURE_2D, big, 0)
    gl.ClearColor(1.0, 0.0, 0.0, 1.0)
    gl.Clear(G.GL_COLOR_BUFFER_BIT)                                      # nothing to clear: must not fault
    gl.Finish()
    gl.BindTexture(G.GL_TEXTURE_2D, small)
    gl.TexStorage2D(G.GL_TEXTURE_2D, 1, G.GL_RGBA8, 64, 64)
    assert gl.GetError() == 0
    gl.FramebufferTexture2D(G.GL_DRAW_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0, G.GL_TEXTURE_2D, small, 0)
    gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, fbo)
    gl.Clear(G.GL_COLOR_BUFFER_BIT)
    px = np.zeros((64, 64, 4), np.uint8)
    gl.ReadPixels(0, 0, 64, 64, G.GL_RGBA, G.GL_UNSIGNED_BYTE, px)
    assert (px[..., 0] == 255).all() and (px[..., 3] == 255).all()
    gl.DestroyContext(ctx)


@pytest.mark.parametrize("name,kw", COPIES, ids=[c[0] for c in COPIES])
def test_hostsim_texture_cache_copies(hostsim, oracle_gcc, name, kw):
    """ps_copy: RGBA8 and R8 texture-cache copies (one pass feeding the next) equal the copies done in numpy -- and the oracle's
    hand-written ps_copy header says the same."""
    fr = scenes.texture_cache_copies(**kw)
    want = copies_expected(fr)
    got, st = render_direct(hostsim, scenes.texture_cache_copies(**kw))
    ref, _ = render_direct(oracle_gcc, scenes.texture_cache_copies(**kw))
    assert st["gl_error"] == 0
    for k, v in want.items():
        assert v.any() and np.array_equal(got[k], v) and np.array_equal(ref[k], v), k


@pytest.mark.parametrize("name,scene,kw", MIX_BLEND, ids=[c[0] for c in MIX_BLEND])
def test_hostsim_mix_blend_matches_oracle(hostsim, oracle_gcc, name, scene, kw):
    """brush_mix_blend: 0 differing bytes against swgl with the hand-written header (itself pinned by the numpy model of the
    blend functions, tests/test_oracle.py)"""
    want, _ = render_direct(oracle_gcc, getattr(scenes, scene)(**kw))
    got, st = render_direct(hostsim, getattr(scenes, scene)(**kw))
    assert st["gl_error"] == 0 and (want != 255).any()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,kw", REPEAT_DUAL, ids=[c[0] for c in REPEAT_DUAL])
def test_hostsim_repeat_dual_source_images_match_oracle(hostsim, oracle_gcc, name, kw):
    """brush_image ALPHA_PASS,ANTIALIASING,DUAL_SOURCE_BLENDING,REPETITION: 0 differing bytes, and not what the plain repetition key draws"""
    want, _ = render_direct(oracle_gcc, scenes.image_repeat(**kw))
    got, st = render_direct(hostsim, scenes.image_repeat(**kw))
    plain, _ = render_direct(oracle_gcc, scenes.image_repeat(**{k: v for k, v in kw.items() if k != "dual"}))
    assert st["gl_error"] == 0 and (want != plain).sum() > 100000
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,kw", DUAL_SOURCE, ids=[c[0] for c in DUAL_SOURCE])
def test_hostsim_dual_source_images_match_oracle(hostsim, oracle_gcc, name, kw):
    """brush_image ALPHA_PASS,DUAL_SOURCE_BLENDING under the dual-source blend state, and the drop-shadow colour modes of the plain
    ALPHA_PASS key (swgl_blendDropShadow per prim): 0 differing bytes, and not what COLOR_MODE_IMAGE draws"""
    want, _ = render_direct(oracle_gcc, scenes.image_grid(**kw))
    got, st = render_direct(hostsim, scenes.image_grid(**kw))
    plain, _ = render_direct(oracle_gcc, scenes.image_grid(**{k: v for k, v in kw.items() if k not in ("dual", "shadows", "screen")}))
    assert st["gl_error"] == 0 and (want != plain).sum() > 100000
    assert np.array_equal(got, want)


def _same(a, b):
    if isinstance(a, dict):
        return set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
    return np.array_equal(a, b)


@pytest.mark.parametrize("name,make", TILE_ROWS, ids=[c[0] for c in TILE_ROWS])
def test_hostsim_tile_rows_match_oracle(hostsim, oracle_gcc, name, make, monkeypatch):
    """picture targets of a few large gradient / image prims: the row kernel (default) and the bin raster give the oracle's bytes"""
    want, _ = render_direct(oracle_gcc, make())
    got, st = render_direct(hostsim, make())
    assert st["row_launches"] >= 1, "the case was meant for wr_tile_rows_kernel"
    assert _same(got, want)
    monkeypatch.setenv("WRHIP_NO_TILE_ROWS", "1")
    got2, st2 = render_direct(hostsim, make())
    assert st2["row_launches"] == 0
    assert _same(got2, want)
    assert st["gl_error"] == st2["gl_error"]        # (the sliver-fence overflow is reported by both)


@pytest.mark.parametrize("name,kw,_full", TEXT_REFTESTS, ids=[c[0] for c in TEXT_REFTESTS])
def test_hostsim_text_reftests_match_oracle(hostsim, oracle_gcc, name, kw, _full):
    """wrench/reftests/text/<name>.yaml over FreeType-rasterised glyphs of the reftest's own font: every target read back"""
    want, _ = render_direct(oracle_gcc, scenes.make_workload("reftest-text-" + name, **kw))
    got, st = render_direct(hostsim, scenes.make_workload("reftest-text-" + name, **kw))
    assert st["gl_error"] == 0
    if isinstance(want, dict):
        for k in want:
            assert np.array_equal(got[k], want[k]), k
        want = want["window"]
    else:
        assert np.array_equal(got, want)
    assert name == "blank" or (want != 255).any(), "no ink"
