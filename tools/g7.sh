cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_n
python3 - <<'PY'
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import abi_surface
from conftest import wrhip_lib, oracle_ref
for lib in [wrhip_lib()] + [os.path.abspath(p) for p in sys.argv[1:]]:
    got = abi_surface.run(lib)
    print(os.path.basename(lib), abi_surface.compare(got, abi_surface.run(oracle_ref()))[:3])
PY
bash tools/ab.sh cfg2 webrender_amd/csrc/libwrhip.so ab/libwrhip_prio3.so
(time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8) 2>&1
