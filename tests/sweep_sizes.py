"""Target-size sweep (test infrastructure, not collected by pytest): twelve families at random odd target sizes (520..1500 x 520..1100) and
random seeds, hostsim library against the oracle.  python tests/sweep_sizes.py <rng seed> <iterations>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import hostsim_lib, oracle_ref
from webrender_amd import scenes
from webrender_amd.harness import render_direct
hs, orc = hostsim_lib(), oracle_ref("gcc")
F = {
 "rot_img": lambda s,w,h: scenes.rotated_images(seed=s, width=w, height=h),
 "rot_img_dual": lambda s,w,h: scenes.rotated_images(seed=s, width=w, height=h, repeat=True, dual=True),
 "persp_img": lambda s,w,h: scenes.rotated_images(seed=s, width=w, height=h, perspective=True),
 "grad": lambda s,w,h: scenes.gradient_grid(seed=s, width=w, height=h),
 "grad_rot": lambda s,w,h: scenes.gradient_grid(seed=s, width=w, height=h, rotate=True),
 "img": lambda s,w,h: scenes.image_grid(seed=s, width=w, height=h),
 "fence": lambda s,w,h: scenes.add_slivers(scenes.image_grid(seed=s, width=w, height=h), pitch=3+s%4, y1=min(240,h)),
 "masked": lambda s,w,h: scenes.masked_rects(seed=s, width=w, height=h, fractional=True),
 "mix_persp": lambda s,w,h: scenes.mix_blend_grid(seed=s, width=w, height=h, perspective=True),
 "rects": lambda s,w,h: scenes.cfg2_overlapping_rects(width=w, height=h, n=150, seed=s, fractional=True),
 "rot_rects": lambda s,w,h: scenes.rotated_rects(seed=s, width=w, height=h),
 "quad_masks": lambda s,w,h: scenes.quad_masks(seed=s, width=w, height=h, rotate=True),
}
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
for it in range(int(sys.argv[2])):
    w = int(rng.integers(520, 1500)); h = int(rng.integers(520, 1100)); s = int(rng.integers(1, 1 << 20))
    for name, mk in F.items():
        try:
            want, _ = render_direct(orc, mk(s, w, h)); got, st = render_direct(hs, mk(s, w, h))
        except Exception as e:
            print(name, s, w, h, type(e).__name__, str(e)[:100], flush=True); continue
        d = int((got != want).sum())
        if d or st["gl_error"]:
            bad += 1; print(f"{name} seed {s} {w}x{h}: {d} bytes differ, gl_error {st['gl_error']:#x}", flush=True)
print("bad", bad)
