#!/usr/bin/env python3
"""usage: tools/kernel_regs.py [libwrhip.so] [name filter]  -- VGPR / SGPR / LDS / scratch of every kernel in the gfx950 code objects
(one per translation unit: wrhip.hip and the instantiation groups of wrhip_inst.hip)."""
import os, re, subprocess, sys, tempfile, glob, shutil
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "webrender_amd", "csrc", "libwrhip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    shutil.copy(lib, os.path.join(td, "lib.so"))
    subprocess.check_call([f"{llvm}/llvm-objdump", "--offloading", os.path.join(td, "lib.so")], stdout=subprocess.DEVNULL)
    for co in sorted(glob.glob(os.path.join(td, "lib.so.*gfx950*"))):
        txt = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
        for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
            lds, name, scratch, sgpr, vgpr = m.groups()
            dem = re.sub(r"\(.*", "", subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip())
            if flt in dem:
                print(f"{dem[:70]:70s} vgpr {vgpr:>4s} sgpr {sgpr:>4s} lds {lds:>6s} scratch {scratch:>5s}")
