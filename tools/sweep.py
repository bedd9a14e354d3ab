"""usage: python tools/sweep.py  -- on the GPU box: libwrhip against the oracle on randomised scenes that stress the prim-list walk
(dense and thinly spread mask words, several 64-word blocks, the depth cap with opaque / translucent mixes) and the general-quad
paths; prints the scenes that differ."""
import sys
import numpy as np
sys.path.insert(0, ".")
from webrender_amd import scenes
from webrender_amd.glapi import wrhip_path
from webrender_amd.harness import render_direct
import __graft_entry__ as ge
ref, lib = ge.oracle_path("gcc"), wrhip_path()
bad = total = 0
def check(name, mk):
    global bad, total
    want, _ = render_direct(ref, mk()); got, _ = render_direct(lib, mk())
    total += 1
    if not np.array_equal(got, want):
        bad += 1
        print("DIFF", name, int((got != want).any(axis=-1).sum()))
for seed in range(6):
    for n, (w, h) in ((300, (512, 512)), (3000, (1024, 512)), (9000, (2048, 1024)), (30000, (2048, 2048))):
        for enc in ("quad", "brush"):
            check(f"rects n={n} {enc} seed={seed}", lambda: scenes.cfg5_many_rects(width=w, height=h, n=n, seed=100 + seed, encoding=enc))
    check(f"rotated seed={seed}", lambda: scenes.add_occluders(scenes.rotated_rects(n=120, seed=500 + seed, opaque_frac=0.3), zmax=120, seed=seed))
    check(f"persp seed={seed}", lambda: scenes.add_occluders(scenes.rotated_rects(n=120, seed=600 + seed, perspective=True), zmax=120, seed=seed))
    check(f"persp images seed={seed}", lambda: scenes.add_occluders(scenes.rotated_images(n=80, seed=700 + seed, perspective="all"), zmax=80, seed=seed))
    check(f"rot images seed={seed}", lambda: scenes.add_occluders(scenes.rotated_images(n=80, seed=800 + seed), zmax=80, seed=seed))
    check(f"images seed={seed}", lambda: scenes.add_occluders(scenes.image_grid(width=2048, height=1024, n=400, seed=900 + seed), n=150, zmax=430, seed=seed))
    check(f"text seed={seed}", lambda: scenes.add_occluders(scenes.cfg3_text(width=2048, height=1024, lines=50, glyphs_per_line=120, run_len=24, seed=seed), zmax=100, seed=seed))
print("scenes", total, "differing", bad)
