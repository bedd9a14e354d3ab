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
#ifndef KDEPTH
#define KDEPTH false
#endif
template __global__ void wr_raster_kernel<WR_FMT_RGBA8, KDEPTH, 4, KFEAT>(const WrTargetDesc*, int, const WrDrawDesc*, const WrPrim*,
    const WrRec*, const WrAux*, const float*, unsigned long long*, int);
'''
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result",
         "-w", "--cuda-device-only", "-S", "-mllvm", "-structurizecfg-skip-uniform-regions"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("feat,depth,max_vgpr", [(0, 0, 64), (0, 1, 128), (5, 0, 168), (7, 0, 168)])
def test_raster_kernel_vgpr_budget(tmp_path, feat, depth, max_vgpr):
    src = tmp_path / "one.hip"
    src.write_text(SRC)
    out = tmp_path / "one.s"
    subprocess.check_call([HIPCC] + FLAGS + [f"-DKFEAT={feat}", f"-DKDEPTH={'true' if depth else 'false'}", "-I", CSRC, str(src), "-o", str(out)])
    asm = out.read_text()
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


FUSED_SRC = SRC.split("template __global__")[0] + '''
template __global__ void wr_setup_raster_kernel<WR_FMT_RGBA8, KDEPTH, 4, KFEAT>(WrSetupArgs, int, const WrTargetDesc*, int, const WrDrawDesc*,
    const WrPrim*, const WrRec*, const WrAux*, const float*, unsigned long long*, int);
'''


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("feat,depth,max_vgpr", [(0, 0, 128), (0, 1, 128), (5, 0, 168), (7, 0, 168)])
def test_fused_setup_raster_kernel_vgpr_budget(tmp_path, feat, depth, max_vgpr):
    """The fused setup + raster variants (the tile pass of a frame carries the next frame's setup stage) run the raster
    body at the occupancy the setup stage's registers leave: 4 waves per SIMD for the rect-only ones (the setup stage
    needs ~110 VGPRs), 3 for the textured ones -- the same steps as the plain variants."""
    src = tmp_path / "fused.hip"
    src.write_text(FUSED_SRC)
    out = tmp_path / "fused.s"
    subprocess.check_call([HIPCC] + FLAGS + [f"-DKFEAT={feat}", f"-DKDEPTH={'true' if depth else 'false'}", "-I", CSRC, str(src), "-o", str(out)])
    asm = out.read_text()
    m = re.search(r"\.amdhsa_kernel _Z22wr_setup_raster_kernelILi3ELb%dELi4ELi%dE.*?\.end_amdhsa_kernel" % (depth, feat), asm, re.S)
    assert m, "kernel not found in the assembly"
    vgpr = int(re.search(r"next_free_vgpr (\d+)", m.group(0)).group(1))
    assert vgpr <= max_vgpr, f"fused FEAT={feat}: {vgpr} VGPRs > {max_vgpr}: the kernel lost a wave per SIMD"


def test_dense_text_kernel_vgpr_budget(tmp_path):
    """Glyph levels run a second instantiation of the textured variant at 4 waves per SIMD (128 VGPRs): cfg3's tile pass is
    latency-bound and gains 8 % from the fourth wave as long as the build spills no more than a few hundred bytes.  Read from
    the built library's code object (the other budgets compile their variant on its own; this one would cost another minute)."""
    lib = os.path.join(CSRC, "libwrhip.so")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(lib) or not os.path.exists(os.path.join(llvm, "llvm-readelf")):
        pytest.skip("libwrhip.so / llvm-readelf not present")
    import glob
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.check_call([f"{llvm}/llvm-objdump", "--offloading", str(tmp_path / "lib.so")], stdout=subprocess.DEVNULL, cwd=tmp_path)
    cos = glob.glob(str(tmp_path / "lib.so.*gfx950*"))      # (one code object per translation unit: wrhip.hip + the instantiation groups)
    if not cos:
        pytest.skip("the library holds no gfx950 code object")
    found = set()
    for co in cos:
        txt = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
        for m in re.finditer(r"\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
            name, scratch, vgpr = m.group(1), int(m.group(2)), int(m.group(3))
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
            k = re.match(r"void (wr_(?:setup_)?raster_dense_kernel)<\d+, (true|false), 4, \d+>", dem)
            if k:
                found.add((k.group(1), k.group(2)))
                # (the plain variants may spill a few hundred bytes; the fused ones also carry the setup stage's spills)
                limit = 640 if k.group(1) == "wr_raster_dense_kernel" else 1536
                assert vgpr <= 128 and scratch <= limit, (dem.strip(), vgpr, scratch)
    assert found == {("wr_raster_dense_kernel", "false"), ("wr_raster_dense_kernel", "true"),
                     ("wr_setup_raster_dense_kernel", "false"), ("wr_setup_raster_dense_kernel", "true")}
