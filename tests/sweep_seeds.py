"""Seed sweep (test infrastructure, not collected by pytest): the round's newest paths on seeds the parity cases do not use,
hostsim library against the oracle.  python tests/sweep_seeds.py [first_seed] [n_seeds]; prints one line per (family, seed)
that differs, and a summary."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import hostsim_lib, oracle_ref
from webrender_amd import scenes
from webrender_amd.harness import render_direct

FAMILIES = {
    "rotated_images_repeat_dual": lambda s: scenes.rotated_images(repeat=True, dual=True, seed=s),
    "rotated_images_repeat_dual_masked": lambda s: scenes.rotated_images(repeat=True, dual=True, masked=True, seed=s),
    "perspective_images_repeat_dual": lambda s: scenes.rotated_images(repeat=True, dual=True, perspective=True, seed=s),
    "image_repeat_dual_aa": lambda s: scenes.image_repeat(dual=True, aa=True, seed=s),
    "mix_grid_perspective": lambda s: scenes.mix_blend_grid(seed=s, perspective=True),
    "mix_grid_perspective_masked": lambda s: scenes.mix_blend_grid(seed=s, perspective=True, masked=True, n=60),
    "mix_grid_rotated": lambda s: scenes.mix_blend_grid(seed=s, rotate=True),
    "mix_grid_near_clipped": lambda s: scenes.mix_blend_grid(seed=s, perspective="clip"),
    "mix_grid_near_clipped_masked": lambda s: scenes.mix_blend_grid(seed=s, perspective="clip", masked=True, n=60),
    "rotated_images": lambda s: scenes.rotated_images(seed=s),
    "perspective_images": lambda s: scenes.rotated_images(perspective=True, seed=s),
    "rotated_gradients": lambda s: scenes.gradient_grid(rotate=True, seed=s),
    "perspective_gradients": lambda s: scenes.gradient_grid(perspective=True, seed=s),
    "fence_images": lambda s: scenes.add_slivers(scenes.image_grid(seed=s), pitch=3 + s % 5),
    "fence_rotated_images": lambda s: scenes.add_slivers(scenes.rotated_images(seed=s), pitch=3 + s % 4),
    "occluded_rotated_images": lambda s: scenes.add_occluders(scenes.rotated_images(seed=s), zmax=80, seed=s + 1),
    "masked_rects_rotated": lambda s: scenes.masked_rects(rotate=True, seed=s),
    "gradient_grid": lambda s: scenes.gradient_grid(seed=s),
    "text_rotated": lambda s: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, rotate=True, seed=s),
    "text_perspective": lambda s: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, perspective=True, seed=s),
    "text_glyph_transform": lambda s: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, rotate=True, glyph_transform=True, seed=s),
    "quad_masks_rotated": lambda s: scenes.quad_masks(rotate=True, seed=s),
    "quad_masks_perspective": lambda s: scenes.quad_masks(perspective=True, seed=s),
    "quad_gradients_rotated": lambda s: scenes.quad_gradients(rotate=True, seed=s),
    "quad_gradients_perspective": lambda s: scenes.quad_gradients(perspective=True, seed=s),
    "split_composites": lambda s: scenes.split_composites(seed=s),
    "split_composites_perspective_masked": lambda s: scenes.split_composites(seed=s, perspective=True, masked=True),
    "filters_rotated": lambda s: scenes.filter_grid(rotate=True, seed=s, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]),
    "filters_perspective": lambda s: scenes.filter_grid(perspective=True, seed=s, ops=[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]),
    "rotated_rects_perspective_occluded": lambda s: scenes.add_occluders(scenes.rotated_rects(perspective=True, seed=s), zmax=80, seed=s + 3),
    "underlay_images": lambda s: scenes.add_perspective_underlay(scenes.image_grid(seed=s), seed=s + 5),
    "underlay_gradients": lambda s: scenes.add_perspective_underlay(scenes.gradient_grid(seed=s), seed=s + 5),
    "blend_modes": lambda s: scenes.blend_modes(seed=s),
    "yuv_grid": lambda s: scenes.yuv_grid(seed=s),
    "image_repeat": lambda s: scenes.image_repeat(seed=s),
    "masked_rects_aa": lambda s: scenes.masked_rects(force_aa=True, fractional=True, seed=s),
    # off-screen families (every read-back target is compared)
    "border_solid": lambda s: scenes.border_solid(seed=s),
    "border_segments": lambda s: scenes.border_segments(seed=s),
    "cache_decorations": lambda s: scenes.cache_decorations(seed=s, n_rgrads=10, n_cgrads=10),
    "svg_filters": lambda s: scenes.svg_filters(seed=s),
    "svg_filter_nodes": lambda s: scenes.svg_filters(node=True, seed=s),
    "texture_cache_copies": lambda s: scenes.texture_cache_copies(seed=s),
    "yuv_composites": lambda s: scenes.yuv_composites(seed=s),
    "yuv_composites_nv12": lambda s: scenes.yuv_composites(seed=s, planar=False),
    "scaled_composites": lambda s: scenes.scaled_composites(seed=s),
    "yuv_grid_hdr": lambda s: scenes.yuv_grid(seed=s, hdr=True),
    "image_grid_masked": lambda s: scenes.image_grid(seed=s, masked=True),
    "filter_grid_masked": lambda s: scenes.filter_grid(seed=s, masked=True),
    "opacity": lambda s: scenes.filter_grid(seed=s, shader="opacity"),
    "texture_rect_images": lambda s: scenes.texture_rect(scenes.image_grid(seed=s)),
    "texture_rect_composites": lambda s: scenes.texture_rect(scenes.scaled_composites(seed=s)),
    "cfg5": lambda s: scenes.cfg5_many_rects(width=1024, height=768, n=4000, seed=s),
    "cfg3_modes": lambda s: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, color_modes=(0, 1, 2, 3), seed=s),
    "cfg3_dual": lambda s: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, color_modes=(1, 2), dual_source=True, seed=s),
}


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else list(FAMILIES)
    hs, orc = hostsim_lib(), oracle_ref("gcc")
    bad = 0
    for name in only:
        for s in range(first, first + n):
            try:
                want, _ = render_direct(orc, FAMILIES[name](s))
                got, st = render_direct(hs, FAMILIES[name](s))
            except Exception as e:      # a scene builder that does not take the seed / raises
                print(f"{name} seed {s}: {type(e).__name__}: {e}", flush=True)
                continue
            if isinstance(want, dict):
                want, got = np.concatenate([want[k].ravel() for k in sorted(want)]), np.concatenate([got[k].ravel() for k in sorted(want)])
            d = int((got != want).sum())
            if d or st["gl_error"]:
                bad += 1
                print(f"{name} seed {s}: {d} differing bytes, max {int(np.abs(got.astype(np.int16) - want.astype(np.int16)).max())}, gl_error {st['gl_error']:#x}, "
                      f"unsupported {st.get('unsupported_prims')}", flush=True)
        print(f"{name}: done", flush=True)
    print("differing cases:", bad)


if __name__ == "__main__":
    main()
