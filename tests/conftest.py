import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch bundles its own HIP runtime; if it is going to be used in this process
    # (RCCL test) it has to initialise the device before libwrhip does.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def oracle_lib(kind="gcc"):
    p = os.path.join(ROOT, "oracle", "_ref", f"libswgl_ref_{kind}.so")
    return p if os.path.exists(p) else None


def oracle_ref(kind="gcc"):
    """The oracle: the reference's gl.cc built with the shader headers generated from the reference's GLSL (oracle/gen) --
    g++ strict IEEE ("gcc") or clang with swgl/build.rs's flags ("clang"); the build with the hand-written headers where
    the generated one is absent."""
    return oracle_lib("gen" if kind == "gcc" else "gen_clang") or oracle_lib(kind)


def hostsim_lib():
    # (WRHIP_HOSTSIM_LIB: another build of the host simulation, e.g. the AddressSanitizer / UBSan one of tools/asan.sh)
    p = os.environ.get("WRHIP_HOSTSIM_LIB") or os.path.join(ROOT, "webrender_amd", "csrc", "libwrhip_hostsim.so")
    return p if os.path.exists(p) else None


def wrhip_lib():
    return os.path.join(ROOT, "webrender_amd", "csrc", "libwrhip.so")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build native pieces once per session if they are missing (CPU-only: hipcc
    cross-compiles gfx950 without a GPU)."""
    need = [wrhip_lib(), os.path.join(ROOT, "webrender_amd", "csrc", "libwr_replay.so")]
    if not all(os.path.exists(p) for p in need) or hostsim_lib() is None:
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session")
def oracle_gcc():
    """THE oracle of the parity tests: the reference's gl.cc built (g++, strict IEEE) with the shader headers generated
    from the reference's GLSL (oracle/gen); the hand-written headers only where that build is absent."""
    p = oracle_ref("gcc")
    if p is None:
        pytest.skip("oracle/_ref/libswgl_ref_gen.so not built (needs /root/reference)")
    return p


@pytest.fixture(scope="session")
def oracle_hand():
    """gl.cc with the hand-written headers of oracle/shaders/: the second derivation (tests/test_oracle_gen.py)"""
    p = oracle_lib("gcc")
    if p is None:
        pytest.skip("oracle/_ref/libswgl_ref_gcc.so not built (needs /root/reference)")
    return p


@pytest.fixture(scope="session")
def oracle_clang():
    p = oracle_ref("clang")
    if p is None:
        pytest.skip("oracle/_ref/libswgl_ref_gen_clang.so not built (needs /root/reference)")
    return p


@pytest.fixture(scope="session")
def oracle_gen():
    p = oracle_lib("gen")
    if p is None:
        pytest.skip("oracle/_ref/libswgl_ref_gen.so not built (needs /root/reference)")
    return p


@pytest.fixture(scope="session")
def hostsim():
    p = hostsim_lib()
    if p is None:
        pytest.skip("hostsim library not built")
    return p
