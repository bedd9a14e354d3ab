"""Register budgets of the raster kernels (gfx950 cross-compile, no GPU needed).

The kernels are latency- or VALU-bound and their speed tracks occupancy in steps: the rect-only
variant must stay at 64 VGPRs (8 waves per SIMD), the textured RGBA8 variants at 168 (3 waves; at 170
they drop to 2 and the cfg3 glyph pass goes from 240 us to 340 us -- that happened twice during round
1 when an innocent-looking code path grew).  This test compiles each variant on its own and checks
the allocation the compiler reports."""
import os
import re
import shutil
import subprocess
import pytest
from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "webrender_amd", "csrc")
SRC = '''
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "wrhip_types.h"
#include "wrhip_kernels.h"
template __global__ void wr_raster_kernel<WR_FMT_RGBA8, false, 4, 0>(const WrTargetDesc*, int, const WrDrawDesc*, const WrPrim*,
    const WrRec*, const WrAux*, const float*, unsigned long long*, int);
template __global__ void wr_raster_kernel<WR_FMT_RGBA8, true, 4, 0>(const WrTargetDesc*, int, const WrDrawDesc*, const WrPrim*,
    const WrRec*, const WrAux*, const float*, unsigned long long*, int);
'''
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result",
         "-w", "--cuda-device-only", "-S", "-mllvm", "-structurizecfg-skip-uniform-regions"]


_LIB_NOTES = None


def lib_kernels(tmp_path):
    """{demangled kernel name: (vgpr_count, scratch bytes)} of the built library's gfx950 code objects -- what ships, and
    seconds instead of the minute a stand-alone compile of a textured variant takes."""
    global _LIB_NOTES
    if _LIB_NOTES is not None:
        return _LIB_NOTES
    import glob
    lib = os.path.join(CSRC, "libwrhip.so")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(lib) or not os.path.exists(os.path.join(llvm, "llvm-readelf")):
        pytest.skip("libwrhip.so / llvm-readelf not present")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.check_call([f"{llvm}/llvm-objdump", "--offloading", str(tmp_path / "lib.so")], stdout=subprocess.DEVNULL, cwd=tmp_path)
    cos = glob.glob(str(tmp_path / "lib.so.*gfx950*"))      # (one code object per translation unit: wrhip.hip + the instantiation groups)
    if not cos:
        pytest.skip("the library holds no gfx950 code object")
    raw = {}
    for co in cos:
        txt = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
        for m in re.finditer(r"\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
            raw[m.group(1)] = (int(m.group(3)), int(m.group(2)))
    dem = subprocess.run(["c++filt"], input="\n".join(raw), capture_output=True, text=True).stdout.split("\n")
    _LIB_NOTES = {re.sub(r"\(.*", "", d.replace("void ", "")): raw[n] for n, d in zip(raw, dem)}
    return _LIB_NOTES


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_rect_only_kernel_code_shape(tmp_path):
    """The rect-only variants (both in one compile, under a minute): their allocation and the shape of their assembly."""
    src = tmp_path / "one.hip"
    src.write_text(SRC)
    out = tmp_path / "one.s"
    subprocess.check_call([HIPCC] + FLAGS + ["-I", CSRC, str(src), "-o", str(out)])
    asm = out.read_text()
    for feat, depth, max_vgpr in ((0, 0, 64), (0, 1, 128)):
        _check_rect_only(asm, feat, depth, max_vgpr)


def _check_rect_only(asm, feat, depth, max_vgpr):
    m = re.search(r"\.amdhsa_kernel _Z16wr_raster_kernelILi3ELb%dELi4ELi%dE.*?\.end_amdhsa_kernel" % (depth, feat), asm, re.S)
    assert m, "kernel not found in the assembly"
    vgpr = int(re.search(r"next_free_vgpr (\d+)", m.group(0)).group(1))
    assert vgpr <= max_vgpr, f"FEAT={feat}: {vgpr} VGPRs > {max_vgpr}: the kernel lost a wave per SIMD"
    if feat == 0:
        # no spills in the prim loop: at most a few cold values (row pointers for the write-back) parked across it
        assert int(re.search(r"private_segment_fixed_size (\d+)", m.group(0)).group(1)) <= 16
        fn = asm[asm.index("\n_Z16wr_raster_kernelILi3ELb%dELi4ELi0E" % depth):]
        fn = fn[:fn.index(".Lfunc_end")].split("\n")
        inner = [i for i, l in enumerate(fn) if "Loop Header: Depth=2" in l]
        head = max(i for i, l in enumerate(fn) if "=>This Loop Header: Depth=1" in l and i < min(inner))      # the round loop
        tail = min(i for i, l in enumerate(fn) if "Loop Header: Depth=1" in l and i > max(inner))             # the mask-clearing loop
        spills = [i for i, l in enumerate(fn) if "scratch_" in l]
        assert all(i < head or i > tail for i in spills), (spills, head, tail)
    if feat == 0 and not depth:
        body = asm[asm.index("\n_Z16wr_raster_kernelILi3ELb0ELi4ELi0E"):]
        body = body[:body.index(".Lfunc_end")]
        # the in-place loop: no register-file copies of the pixel state at the loop back-edge
        assert body.count("v_mov_b64") < 80


def test_shipped_kernel_budgets(tmp_path):
    """The allocations of every latency-bound kernel in the built library.  Textured variants: 168 VGPRs = 3 waves per
    SIMD; the fused setup + raster variants run the raster body at the occupancy the setup stage's registers leave (4 waves
    for the rect-only ones, the setup stage needs ~110 VGPRs; 3 for the textured ones); glyph levels run a second
    instantiation of the textured variant at 4 waves per SIMD (cfg3's tile pass gains 8 % from the fourth wave as long as
    the build spills no more than a few hundred bytes); the row kernels of round 5 are sized for 4 waves (tile rows) and 7
    (span rows); the setup stage's scratch is the one number that regressed twice this round (an unrolled table walk took it
    from 560 B to 2.9 KB)."""
    k = lib_kernels(tmp_path)
    budget = {                                            # name: (max VGPRs, max scratch bytes)
        "wr_raster_kernel<3, false, 4, 0>": (64, 16), "wr_raster_kernel<3, true, 4, 0>": (128, 16),
        "wr_raster_kernel<3, false, 4, 5>": (168, 256), "wr_raster_kernel<3, false, 4, 7>": (168, 256),
        "wr_raster_kernel<3, true, 4, 5>": (168, 512), "wr_raster_kernel<3, true, 4, 7>": (168, 512),
        # (every kernel that carries the setup stage: + 272 B since the near-plane clip carries brush_mix_blend's second varying -- two
        # ten-point polygons of eight floats instead of six, two more ten-float arrays for the walk; touched by clipped prims only:
        # cfg2's fused launch is unchanged, transforms' + 1 %, profiles/r06_zz_mixclip_ab.txt)
        "wr_setup_raster_kernel<3, false, 4, 0>": (128, 1280), "wr_setup_raster_kernel<3, true, 4, 0>": (128, 1280),
        "wr_setup_raster_kernel<3, false, 4, 5>": (168, 1280), "wr_setup_raster_kernel<3, false, 4, 7>": (168, 1280),
        "wr_setup_raster_kernel<3, true, 4, 5>": (168, 1536), "wr_setup_raster_kernel<3, true, 4, 7>": (168, 1536),
        "wr_raster_dense_kernel<3, false, 4, 7>": (128, 640), "wr_raster_dense_kernel<3, true, 4, 7>": (128, 640),
        "wr_setup_raster_dense_kernel<3, false, 4, 7>": (128, 1792), "wr_setup_raster_dense_kernel<3, true, 4, 7>": (128, 1792),
        "wr_setup_kernel": (168, 1024), "wr_setup_rows_kernel": (128, 1280), "wr_mask_rows_kernel": (128, 64),
        "wr_span_rows_kernel": (72, 192), "wr_tile_rows_kernel": (128, 512), "wr_setup_tile_rows_kernel": (128, 1280),
    }
    missing = [n for n in budget if n not in k]
    assert not missing, missing
    over = {n: k[n] for n, (v, sc) in budget.items() if k[n][0] > v or k[n][1] > sc}
    assert not over, f"(VGPRs, scratch) over budget: {over}"
