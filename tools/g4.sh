cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_k
python3 - <<'PY'
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import abi_surface
from conftest import wrhip_lib, oracle_ref
for lib in [wrhip_lib()] + [os.path.abspath(p) for p in sys.argv[1:]]:
    got = abi_surface.run(lib)
    print(os.path.basename(lib), abi_surface.compare(got, abi_surface.run(oracle_ref())))
PY
