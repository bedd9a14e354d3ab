#!/usr/bin/env python3
"""usage: python tools/cfg4_probe.py  -- per-kernel times of cfg4's mask pass with one of its two prims removed (GPU box)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from webrender_amd import scenes, glapi
from webrender_amd.harness import record_scene, ScenePlayer


def run(tag, edit):
    fr = scenes.cfg4_box_shadow(dps=2.0)
    edit(fr)
    rec, _ = record_scene(glapi.wrhip_path(), fr)
    p = ScenePlayer(glapi.wrhip_path(), rec)
    get_k = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)(p.symbol("WrhipGetKernelStats"))
    reset = C.CFUNCTYPE(None)(p.symbol("WrhipResetStats"))
    prof = C.CFUNCTYPE(None, C.c_int)(p.symbol("WrhipSetProfiling"))
    p.frames(3, 0)
    prof(1); p.frames(2, 0); reset(); p.frames(0, 10)
    ks = (glapi.WrhipKernelStat * 32)(); n = get_k(ks, 32); prof(0)
    print(tag, [(f"k{k.kind}<{k.fmt},{k.feat}>", round(k.launches / 10, 1), round(k.ns / k.launches / 1e3, 1)) for k in list(ks)[:n]])


def mask_target(fr):
    for targets in fr.passes:
        for t in targets:
            if t.texture.name == "bs_prim_masks":
                return t

run("full      ", lambda fr: None)
run("box only  ", lambda fr: mask_target(fr).steps.__delitem__(1))
run("clip only ", lambda fr: mask_target(fr).steps.__delitem__(0))
run("no masks  ", lambda fr: (mask_target(fr).steps.__delitem__(0), mask_target(fr).steps.__delitem__(0)))
