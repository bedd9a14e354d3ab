"""The ABI surface outside the frame scenes (queries, PBOs, buffer mapping, texture copies / clears, scaled and flipped
blits, externally backed textures, locked resources, a depth buffer surviving a mid-target flush, overlapping uploads):
the same call sequence (tests/abi_surface.py) on the reference's swgl and on libwrhip, every observable compared."""
import json
import os
import pytest
from conftest import ROOT, wrhip_lib, oracle_ref
import abi_surface

GOLDEN = os.path.join(ROOT, "tests", "golden", "abi_surface.json")


def test_hostsim_abi_surface_matches_oracle(hostsim, oracle_gcc):
    want = abi_surface.run(oracle_gcc)
    got = abi_surface.run(hostsim)
    assert not abi_surface.compare(got, want)
    # the committed digests (what the GPU box compares with when the oracle is absent) are the oracle's
    gold = json.load(open(GOLDEN))
    dig = abi_surface.digest_of(want)
    assert {k: dig[k] for k in gold} == gold
    assert want["samples_rects"] > 0 and want["samples_rotated"] > 0 and want["time_elapsed_positive"] == 1


@pytest.mark.gpu
def test_hip_abi_surface_matches_oracle():
    got = abi_surface.run(wrhip_lib())
    ref = oracle_ref()
    if ref:
        assert not abi_surface.compare(got, abi_surface.run(ref))
    gold = json.load(open(GOLDEN))
    dig = abi_surface.digest_of(got)
    bad = [k for k in gold if dig[k] != gold[k] and k not in abi_surface.BACKEND_SPECIFIC]
    assert not bad, bad
