"""Row-sharding sweep (test infrastructure, not collected by pytest): what a rank of the multi-GPU path renders, emulated rank by rank in
one process -- the scene recorded once, WrhipSetTargetRows per tile and for the window exactly as webrender_amd/dist.py sets them, one
frame, the rank's own framebuffer rows read back -- the strips of all ranks put together against the oracle's unsharded frame, over
scene families that use the flush's pool (row tables, depth runs, gradient tables) and the row kernels, at world sizes 2 / 3 / 5 / 8.
python tests/sweep_shards.py <first seed> <seeds per family>"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import hostsim_lib, oracle_ref
from webrender_amd import scenes
from webrender_amd.dist import target_rows_for_rank
from webrender_amd.harness import render_direct, record_scene, ScenePlayer
hs, orc = hostsim_lib(), oracle_ref("gcc")
F = {
    "rects": lambda s: scenes.cfg2_overlapping_rects(width=1024, height=1000, n=120, seed=s, fractional=True),
    "rotated_images": lambda s: scenes.rotated_images(seed=s),
    "perspective_images": lambda s: scenes.rotated_images(seed=s, perspective=True),
    "rotated_images_repeat_dual": lambda s: scenes.rotated_images(seed=s, repeat=True, dual=True),
    "gradients": lambda s: scenes.gradient_grid(seed=s),
    "rotated_gradients": lambda s: scenes.gradient_grid(seed=s, rotate=True),
    "fence_images": lambda s: scenes.add_slivers(scenes.image_grid(seed=s), pitch=3),
    "occluded_gradients": lambda s: scenes.add_occluders(scenes.gradient_grid(seed=s), zmax=80, seed=s + 1),
    "text": lambda s: scenes.cfg3_text(width=1024, height=512, lines=20, glyphs_per_line=60, run_len=12, seed=s),
    "mix_perspective": lambda s: scenes.mix_blend_grid(seed=s, perspective=True),
    "masked_rects_rotated": lambda s: scenes.masked_rects(seed=s, rotate=True),
    "cfg5": lambda s: scenes.cfg5_many_rects(width=1024, height=768, n=4000, seed=s),
}


def rank_strip(make, rank, world, lib=None):
    lib = lib or hs
    rec, _ = record_scene(lib, make())
    p = ScenePlayer(lib, rec)
    sym = p.symbol
    set_rows = C.CFUNCTYPE(None, C.c_uint32, C.c_int32, C.c_int32)(sym("WrhipSetTargetRows"))
    fb_tex = C.CFUNCTYPE(C.c_uint32, C.c_uint32)(sym("WrhipGetFramebufferTexture"))(0)
    rows, fb = target_rows_for_rank(rec, rank, world)
    for name, (y0, y1) in rows.items():
        set_rows(rec.texture_ids[name], *((y0, y1) if y1 > y0 else (1, 0)))
    set_rows(fb_tex, *(fb if fb[1] > fb[0] else (1, 0)))
    p.rp.exec(rec.frame)
    px = p.read_pixels()
    err = C.CFUNCTYPE(C.c_uint32)(sym("GetError"))()
    return fb, px, err


def main():
    first, n = int(sys.argv[1]), int(sys.argv[2])
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else list(F)
    bad = 0
    for name in only:
        for s in range(first, first + n):
            mk = lambda: F[name](s)
            want, _ = render_direct(orc, mk())         # [H, W, 4] RGBA, framebuffer row order of read_pixels?
            full = None
            for world in (2, 3, 5, 8):
                out = np.zeros_like(want)
                errs = 0
                for rank in range(world):
                    (f0, f1), px, err = rank_strip(mk, rank, world)
                    errs |= err
                    if f1 > f0:
                        out[f0:f1] = px[f0:f1]
                if full is None:                           # the unsharded frame through the same player (row order / channel order of read_pixels)
                    rec, _ = record_scene(hs, mk()); p = ScenePlayer(hs, rec); p.rp.exec(rec.frame); full = p.read_pixels()
                if not np.array_equal(out, full) or errs:
                    bad += 1
                    rows = np.nonzero((out != full).any(axis=(1, 2)))[0]
                    print(f"{name} seed {s} world {world}: {int((out != full).sum())} differing bytes, rows {rows[:1]}..{rows[-1:]}, gl_error {errs:#x}", flush=True)
            # and the player's unsharded frame is the oracle's (render_direct flips / swizzles: compare as a multiset of rows both ways)
            if not (np.array_equal(full, want) or np.array_equal(full[::-1], want) or np.array_equal(full[::-1][..., [2, 1, 0, 3]], want) or np.array_equal(full[..., [2, 1, 0, 3]], want)):
                bad += 1
                print(f"{name} seed {s}: the unsharded player frame is not the oracle's", flush=True)
        print(name, "done", flush=True)
    print("bad", bad)


if __name__ == "__main__":
    main()
