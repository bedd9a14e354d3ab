"""Pipelined-sequence sweep (test infrastructure, not collected by pytest): random sequences of 6-15 frames from a menu of fifteen scenes
streamed without a Finish in between (harness.render_pipelined), every frame against the oracle's render of that scene alone.
python tests/sweep_pipelined.py <rng seed> <iterations>"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import hostsim_lib, oracle_ref
from webrender_amd import scenes
from webrender_amd.harness import render_direct, render_pipelined
hs, orc = hostsim_lib(), oracle_ref("gcc")
W = dict(width=512, height=512)
MENU = [
 lambda: scenes.cfg2_overlapping_rects(n=60, seed=40, **W),
 lambda: scenes.masked_rects(n=40, **W),
 lambda: scenes.cfg2_overlapping_rects(n=70, seed=41, encoding="brush", fractional=True, **W),
 lambda: scenes.image_grid(n=40, **W),
 lambda: scenes.gradient_grid(n=20, **W),
 lambda: scenes.gradient_grid(n=20, seed=62, **W),
 lambda: scenes.gradient_grid(n=20, rotate=True, seed=66, **W),
 lambda: scenes.rotated_rects(n=30, opaque_frac=0.3, **W),
 lambda: scenes.add_slivers(scenes.image_grid(n=40, **W), pitch=3),
 lambda: scenes.gradient_grid(n=20, perspective=True, seed=67, **W),
 lambda: scenes.masked_rects(n=40, rotate=True, seed=13, **W),
 lambda: scenes.add_occluders(scenes.gradient_grid(n=20, seed=64, **W), n=20, zmax=40, seed=9),
 lambda: scenes.cfg5_many_rects(n=1500, **W),
 lambda: scenes.quad_masks(n=30, rotate=True, seed=86, **W),
 lambda: scenes.filter_grid(n=30, seed=71, **W),
]
# static textures agree by name?
seen = {}
ok = []
for i, m in enumerate(MENU):
    try:
        f = m()
    except Exception as e:
        print("menu", i, type(e).__name__, e); continue
    good = True
    for ref in f.static_textures:
        h = hashlib.sha1(np.ascontiguousarray(ref.pixels).tobytes()).hexdigest() if ref.pixels is not None else None
        if seen.setdefault(ref.name, h) != h:
            good = False
    if good: ok.append(i)
    else: print("menu", i, "dropped (static texture name clash)")
want = {i: render_direct(orc, MENU[i]())[0] for i in ok}
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
for it in range(int(sys.argv[2])):
    seq = [int(rng.choice(ok)) for _ in range(int(rng.integers(6, 16)))]
    got = render_pipelined(hs, [MENU[i]() for i in seq])
    for k, (g, i) in enumerate(zip(got, seq)):
        if not np.array_equal(g[..., [2, 1, 0, 3]], want[i]):
            bad += 1; print("iter", it, "seq", seq, "frame", k, "menu", i, "differs", flush=True)
print("bad", bad)
