"""libwrhip against the build of swgl that SHIPS (VERDICT r4, next #1a).

The bit-exact parity target of every other test is the reference's strict-IEEE configuration (g++, no fast-math, the
portable vector_type.h path: swgl/build.rs:186-194).  What Firefox links is the clang configuration of swgl/build.rs:150-204
(-ffast-math -mrecip=none -fno-finite-math-only, SSE2 intrinsics: oracle/_ref/libswgl_ref_gen_clang.so here).  The two builds of
the reference do not agree with each other; these tests hold the deviation of libwrhip (== the strict build) from the shipping
build inside a stated budget for EVERY parity family (parity_cases.CLANG_BUDGET; causes in DESIGN section 7), and pin the one
case where the shipping build computes NaN.

  not gpu:  strict build vs shipping build (the reference's own spread: what the budgets are derived from), every scene
  gpu:      libwrhip on the MI355X vs the shipping build, every scene + the wrench workloads at 4K
"""
import numpy as np
import pytest
from conftest import wrhip_lib, oracle_ref
from webrender_amd import scenes
from webrender_amd.harness import render_direct
from parity_cases import family_scenes, clang_budget, WRENCH
from test_hostsim_parity import CASES, BLUR_CASES, CLIP_CASES, BOX_CASES

_FAM = family_scenes(CASES, BLUR_CASES, CLIP_CASES, BOX_CASES)
_IDS = [f"{f}-{n}" for f, n, _ in _FAM]


@pytest.fixture(autouse=True)
def _no_degenerate_radii(monkeypatch):
    # start radius == end radius: NaN in the shipping build (test below)
    monkeypatch.setattr(scenes, "RADIAL_DEGENERATE", False)


def _spread(a, b):
    if not isinstance(a, dict):
        a, b = {"window": a}, {"window": b}
    assert set(a) == set(b)
    tot = n1 = n4 = mx = 0
    for k in a:
        d = np.abs(a[k].astype(np.int16) - b[k].astype(np.int16))
        tot += d.size
        n1 += int((d > 1).sum())
        n4 += int((d > 4).sum())
        mx = max(mx, int(d.max()) if d.size else 0)
    return mx, n1 / max(tot, 1), n4 / max(tot, 1)


import json
import os
# The MEASURED spread of every scene (tests/golden/make_clang_spread.py): [max, bytes > 1 LSB, bytes > 4 LSB, bytes].  Both builds of the
# reference are deterministic, so on the CPU the numbers must come out UNCHANGED -- a family that improves shows as much as one that gets
# worse ("a budget 2-3 x above the worst scene is not a bound anyone can act on") --; on the MI355X libwrhip equals the strict
# build except where the device's math library enters (<= 1 LSB on a few bytes: DESIGN section 7), so it is held to the same numbers with
# a margin of max(16 bytes, 2 %) and one LSB on the maximum.  The per-family budgets (parity_cases.CLANG_BUDGET) remain as the table of causes.
# PER CPU VENDOR: the shipping build's rsqrtps / rcpps approximations differ between AMD and Intel, so the same oracle binary has another spread
# on the GPU box's EPYC than in the Intel authoring container (radial gradients, KHR blend equations); a vendor without pins keeps the budgets only.
_VENDOR = next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("vendor_id")), "unknown")
_PINS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clang_spread.json"))).get(_VENDOR, {})


def _counts(a, b):
    if not isinstance(a, dict):
        a, b = {"window": a}, {"window": b}
    assert set(a) == set(b)
    tot = n1 = n4 = mx = 0
    for k in a:
        d = np.abs(a[k].astype(np.int16) - b[k].astype(np.int16))
        tot += d.size; n1 += int((d > 1).sum()); n4 += int((d > 4).sum()); mx = max(mx, int(d.max()) if d.size else 0)
    return [mx, n1, n4, tot]


def _check(family, name, got, want, exact=False, pinned=True):
    mx, f1, f4 = _spread(got, want)
    cap, b1, b4, cause = clang_budget(family, name)
    assert cap is None or mx <= cap, (name, mx, cause)
    assert f1 <= b1, (name, "bytes above 1 LSB", f1, b1, cause)
    assert f4 <= b4, (name, "bytes above 4 LSB", f4, b4, cause)
    pin = _PINS.get(f"{family}-{name}") if pinned else None
    if pin is None:
        return
    c = _counts(got, want)
    if exact:
        assert c == pin, (name, "measured spread changed: regenerate tests/golden/clang_spread.json if the scene or the oracle was meant to change", c, pin)
    else:
        assert c[3] == pin[3] and c[0] <= pin[0] + 1, (name, c, pin)
        for i in (1, 2):
            assert abs(c[i] - pin[i]) <= max(16, 0.02 * pin[i]), (name, "bytes above %d LSB" % (1 if i == 1 else 4), c, pin)


@pytest.mark.parametrize("family,name,make", _FAM, ids=_IDS)
def test_reference_builds_spread_is_inside_the_budget(family, name, make, oracle_gcc, oracle_clang):
    """strict-IEEE g++ build vs shipping clang build of the reference itself"""
    a, _ = render_direct(oracle_gcc, make())
    b, _ = render_direct(oracle_clang, make())
    _check(family, name, a, b, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("family,name,make", _FAM, ids=_IDS)
def test_hip_vs_shipping_build_is_inside_the_budget(family, name, make):
    ref = oracle_ref("clang")
    if not ref:
        pytest.skip("clang oracle not built")
    got, st = render_direct(wrhip_lib(), make())
    want, _ = render_direct(ref, make())
    _check(family, name, got, want)


_WRENCH_4K = [(n, w, kw) for n, w, _small, kw in WRENCH]


@pytest.mark.gpu
@pytest.mark.parametrize("name,workload,kw", _WRENCH_4K, ids=[c[0] for c in _WRENCH_4K])
def test_hip_wrench_4k_vs_shipping_build(name, workload, kw):
    ref = oracle_ref("clang")
    if not ref:
        pytest.skip("clang oracle not built")
    got, _ = render_direct(wrhip_lib(), scenes.make_workload(workload, **kw))
    want, _ = render_direct(ref, scenes.make_workload(workload, **kw))
    _check("wrench", name, got, want, pinned=False)      # (the pins are the CPU-sized scenes')


def _degenerate_only():
    fr = scenes.quad_gradients(only=[4, 19, 24])       # the r0 == r1 prims of the scene without repeat (mode 4)
    return fr


def test_degenerate_radial_gradient_is_nan_in_the_shipping_build(monkeypatch, oracle_gcc, oracle_clang, hostsim):
    """start radius == end radius: radius_scale = 0, every position is (0, 0) and the merged-gradient loop evaluates
    fastSqrt<false>(dot(pos, pos)) = fastSqrt(0) (swgl_ext.h:1789).  Portable build: sqrt(0) = 0 -> the first stop's colour.
    SSE2 build: 0 * rsqrt(0) = 0 * inf = NaN -> round_pixel(NaN) -> cvtps2dq's integer indefinite, packed with saturation: pixels
    that depend on nothing in the gradient.  libwrhip follows the portable build (what the GLSL says: offset 0)."""
    monkeypatch.setattr(scenes, "RADIAL_DEGENERATE", True)
    a, _ = render_direct(oracle_gcc, _degenerate_only())
    b, _ = render_direct(oracle_clang, _degenerate_only())
    h, _ = render_direct(hostsim, _degenerate_only())
    assert np.array_equal(h, a)
    drawn = (a != 255).any(axis=2)
    differ = (np.abs(a.astype(int) - b.astype(int)).max(axis=2) > 4) & drawn
    assert differ.sum() > 0.5 * drawn.sum()          # most of every such prim
    monkeypatch.setattr(scenes, "RADIAL_DEGENERATE", False)
    a2, _ = render_direct(oracle_gcc, _degenerate_only())
    b2, _ = render_direct(oracle_clang, _degenerate_only())
    assert (np.abs(a2.astype(int) - b2.astype(int)) > 4).mean() < 2e-3   # the same prims with r1 = r0 + 4.5
