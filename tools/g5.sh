cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_l
python3 - ab/libwrhip_nostage.so <<'PY'
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import abi_surface
from conftest import wrhip_lib, oracle_ref
for lib in [wrhip_lib()] + [os.path.abspath(p) for p in sys.argv[1:]]:
    got = abi_surface.run(lib)
    print(os.path.basename(lib), abi_surface.compare(got, abi_surface.run(oracle_ref()))[:3])
PY
WRHIP_LIB_PATH=$PWD/ab/libwrhip_timing.so WRHIP_DEBUG_COUNTERS=1 WRHIP_PRIM_TIMES=$PWD/gpurun_out/r06_l/prim_times_cfg2.bin python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r06_l/timing_cfg2.err | grep '"metric"' > gpurun_out/r06_l/timing_cfg2.json
grep "dbg counters" gpurun_out/r06_l/timing_cfg2.err | tail -4
python3 - <<'PY'
import numpy as np
a = np.fromfile('gpurun_out/r06_l/prim_times_cfg2.bin', dtype=np.uint32).reshape(-1, 4)
n = 1024
print('vertex (10ns): median', np.median(a[:n,0]), 'max', a[:n,0].max(), ' post-vertex:', np.median(a[:n,1]), a[:n,1].max(), ' to mid:', np.median(a[:n,2]))
PY
bash tools/ab.sh cfg2 webrender_amd/csrc/libwrhip.so ab/libwrhip_nostage.so
