#!/usr/bin/env python3
"""Generates tests/golden/clang_spread.json: for every scene of the parity families, the MEASURED deviation of the reference's strict-IEEE build
(g++: what libwrhip equals byte for byte) from the build the reference ships (clang, swgl/build.rs:150-204): [maximum difference, bytes
differing by more than 1 LSB, bytes differing by more than 4 LSB, bytes compared].  Both builds are deterministic, so these are exact numbers,
not budgets: tests/test_clang_budget.py demands them unchanged on the CPU (a scene or an oracle build that drifts shows) and holds libwrhip on the
MI355X to them.  Run in the authoring container: python tests/golden/make_clang_spread.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_ref
from webrender_amd import scenes
from webrender_amd.harness import render_direct
from parity_cases import family_scenes
from test_hostsim_parity import CASES, BLUR_CASES, CLIP_CASES, BOX_CASES


def counts(a, b):
    if not isinstance(a, dict):
        a, b = {"window": a}, {"window": b}
    tot = n1 = n4 = mx = 0
    for k in a:
        d = np.abs(a[k].astype(np.int16) - b[k].astype(np.int16))
        tot += d.size; n1 += int((d > 1).sum()); n4 += int((d > 4).sum()); mx = max(mx, int(d.max()) if d.size else 0)
    return [mx, n1, n4, tot]


def main():
    scenes.RADIAL_DEGENERATE = False          # (start radius == end radius: NaN in the shipping build, pinned by its own test)
    gcc, clang = oracle_ref("gcc"), oracle_ref("clang")
    out = {}
    for fam, name, make in family_scenes(CASES, BLUR_CASES, CLIP_CASES, BOX_CASES):
        out[f"{fam}-{name}"] = counts(render_direct(gcc, make())[0], render_direct(clang, make())[0])
        print(fam, name, out[f"{fam}-{name}"], flush=True)
    # The shipping build uses SSE's rsqrtps / rcpps approximations (fastSqrt, recip: swgl/src/glsl.h), whose results differ between CPU
    # vendors: the SAME binary gives another spread on an AMD host than on an Intel one (radial gradients, the KHR blend equations; found
    # in round 6 when the Intel container's numbers failed on the GPU box's EPYC).  So the pins are kept per vendor; the strict build, and
    # libwrhip, do not depend on the host.
    vendor = next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("vendor_id")), "unknown")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "clang_spread.json")
    allp = json.load(open(path)) if os.path.exists(path) else {}
    if allp and not all(isinstance(v, dict) for v in allp.values()):
        allp = {}
    allp[vendor] = out
    json.dump(allp, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, "for", vendor)


if __name__ == "__main__":
    main()
