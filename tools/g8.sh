cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_o
python3 - <<'PY'
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import abi_surface
from conftest import wrhip_lib, oracle_ref
got = abi_surface.run(wrhip_lib())
print('abi', abi_surface.compare(got, abi_surface.run(oracle_ref()))[:3])
PY
bash tools/ab_env.sh r06_o "cfg2 cfg3 cfg5 cfg4" "WRHIP_ANYORDER_SETUP=1" 2>&1 | cut -c1-260
WRHIP_ANYORDER_SETUP=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
