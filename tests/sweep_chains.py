"""Off-screen chain sweep (test infrastructure, not collected by pytest): cs_blur / cs_scale chains, clip and box-shadow mask targets and
the whole box-shadow chain (cfg4's shape) at random parameters, every read-back target of the hostsim library against the oracle's.
python tests/sweep_chains.py <rng seed> <iterations>"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import hostsim_lib, oracle_ref
from webrender_amd import scenes
from webrender_amd.harness import render_direct
GENERATORS = None      # (blur, clip, box, chain), set below


def blur(r):
    fmt = ["r8", "rgba8"][int(r.integers(2))]
    steps = int(r.integers(0, 3))
    cw, ch = int(r.integers(5, 170)), int(r.integers(3, 150))
    n = int(r.integers(1, 8)) if steps == 0 else 1
    atlas = 512 if (steps or max(cw, ch) > 100 or n > 4) else 256
    return "blur_chain", dict(fmt=fmt, content=(cw, ch), sigma=float(np.round(r.uniform(0.5, 4.0), 2)), n_tasks=n, seed=int(r.integers(1, 9999)),
                              origin=(int(r.integers(0, 20)), int(r.integers(0, 12))), atlas=atlas, scale_steps=steps,
                              pattern=["shapes", "noise"][int(r.integers(2))])


def clip(r):
    return "clip_masks", dict(n=int(r.integers(8, 60)), seed=int(r.integers(1, 9999)), dps=[1.0, 1.5, 2.0, 1.25][int(r.integers(4))])


def box(r):
    return "box_shadow_masks", dict(n=int(r.integers(6, 40)), seed=int(r.integers(1, 9999)), dps=[1.0, 1.5, 2.0][int(r.integers(3))])


def chain(r):
    rad = lambda: (float(r.integers(0, 60)), float(r.integers(0, 60)))
    x0, y0 = float(r.integers(20, 200)), float(r.integers(20, 200))
    return "cfg4_box_shadow", dict(width=1024, height=1024, blur_radius=float(r.integers(2, 30)), radii=(rad(), rad(), rad(), rad()),
                                   boxes=[(x0, y0, x0 + float(r.integers(200, 700)), y0 + float(r.integers(200, 700)))],
                                   offset=(float(r.integers(-10, 11)), float(r.integers(-10, 11))), dps=[1.0, 1.0, 1.5][int(r.integers(3))])


GENERATORS = (blur, clip, box, chain)


def main():
    hs, orc = hostsim_lib(), oracle_ref("gcc")
    rng = np.random.default_rng(int(sys.argv[1]))
    bad = 0
    for it in range(int(sys.argv[2])):
        for gen in GENERATORS:
            name, kw = gen(rng)
            for env in ({}, {"WRHIP_NO_SPAN_ROWS": "1", "WRHIP_NO_MASK_ROWS": "1"}) if gen is not chain else ({},):
                os.environ.update(env)
                try:
                    want, _ = render_direct(orc, getattr(scenes, name)(**kw))
                    got, st = render_direct(hs, getattr(scenes, name)(**kw))
                except Exception as e:
                    print(name, kw, type(e).__name__, str(e)[:120], flush=True)
                    break
                finally:
                    for k in env:
                        os.environ.pop(k, None)
                if isinstance(want, dict):
                    diff = [k for k in want if not np.array_equal(got[k], want[k])]
                else:
                    diff = [] if np.array_equal(got, want) else ["window"]
                if diff or st["gl_error"]:
                    bad += 1
                    print(name, kw, env, "differs in", diff, "gl_error", hex(st["gl_error"]), flush=True)
    print("bad", bad)


if __name__ == "__main__":
    main()
