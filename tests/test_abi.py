"""The C-ABI boundary: include/wrhip.h declares exactly the reference's 99
`extern "C"` functions (swgl/src/swgl_fns.rs:23-322) and libwrhip.so exports
every one of them (no compute calls here: loading the library needs no GPU)."""
import os
import re
import subprocess
from conftest import ROOT, wrhip_lib
from webrender_amd.glapi import SIGNATURES, EXTRA_SIGNATURES

REFERENCE_FNS = "/root/reference/swgl/src/swgl_fns.rs"


def header_functions():
    src = open(os.path.join(ROOT, "include", "wrhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^[A-Za-z_][\w \*]*?\b([A-Z]\w+)\s*\(", src, flags=re.M)
    return [n for n in names if n not in ("WRHIP_H",)]


def test_header_declares_reference_abi():
    names = header_functions()
    core = [n for n in names if not n.startswith("Wrhip")]
    assert len(core) == 99
    assert set(core) == set(SIGNATURES)
    assert set(n for n in names if n.startswith("Wrhip")) == set(EXTRA_SIGNATURES)


def test_signature_table_matches_reference_extern_block():
    if not os.path.exists(REFERENCE_FNS):
        import pytest
        pytest.skip("reference tree not present")
    src = open(REFERENCE_FNS).read()
    block = src[src.index('extern "C" {'):src.index("#[derive(Clone, Copy)]")]
    ref = re.findall(r"fn (\w+)\(", block)
    assert len(ref) == 99
    assert ref == list(SIGNATURES), "same functions, same order as swgl_fns.rs:23-322"
    # argument counts agree too
    for m in re.finditer(r"fn (\w+)\((.*?)\)(?: -> [^;]+)?;", block, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if not args else len([a for a in args.split(",") if a.strip()])
        assert n == len(SIGNATURES[name][1]), name


def test_libwrhip_exports_every_symbol():
    lib = wrhip_lib()
    assert os.path.exists(lib), "libwrhip.so not built"
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    missing = [n for n in list(SIGNATURES) + list(EXTRA_SIGNATURES) if n not in exported]
    assert not missing, missing


def test_libwrhip_loads_and_has_gfx950_code():
    import ctypes
    lib = ctypes.CDLL(wrhip_lib())
    for n in SIGNATURES:
        getattr(lib, n)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", wrhip_lib()],
                         capture_output=True, text=True).stdout
    if out:
        assert "gfx950" in out


def test_product_path_does_not_touch_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "webrender_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/_ref" not in txt and "libswgl_ref" not in txt, f
                assert "np_model" not in txt, f


def test_uniform_locations_exist_where_the_reference_has_them(hostsim, oracle_gen):
    """GetUniformLocation answers -1 / not -1 for the same names as the reference's generated programs, for every key the
    backend links: WebRender's Device only binds the samplers a program reports (device/gl.rs bind_samplers), and a call
    stream recorded over one backend replays on the other only if both report the same set."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle", "gen"))
    from gen_shaders import shader_keys
    from webrender_amd import glapi, glconst as G
    from webrender_amd.device import SAMPLER_SLOTS
    names = list(SAMPLER_SLOTS) + ["uTransform", "uMode"]

    def locs(lib):
        gl = glapi.GL(lib)
        ctx = gl.CreateContext()
        gl.MakeCurrent(ctx)
        out = {}
        for key in shader_keys():
            vs, fs = gl.CreateShader(G.GL_VERTEX_SHADER), gl.CreateShader(G.GL_FRAGMENT_SHADER)
            gl.ShaderSourceByName(vs, key.encode())
            gl.ShaderSourceByName(fs, key.encode())
            pid = gl.CreateProgram()
            gl.AttachShader(pid, vs)
            gl.AttachShader(pid, fs)
            gl.LinkProgram(pid)
            if gl.GetLinkStatus(pid):
                out[key] = {n for n in names if gl.GetUniformLocation(pid, n.encode()) != -1}
        gl.DestroyContext(ctx)
        return out
    ref, got = locs(oracle_gen), locs(hostsim)
    assert len(ref) == 82 and len(got) >= 46
    for key, have in got.items():
        assert have == ref[key], key
