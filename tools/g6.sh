cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_m
for w in cfg2 cfg4; do
WRHIP_LIB_PATH=$PWD/ab/libwrhip_timing.so WRHIP_PRIM_TIMES=$PWD/gpurun_out/r06_m/pt_$w.bin python tools/setup_tp.py $w 2>&1 | grep -v amdgpu.ids | tail -3
python3 - $w <<'PY'
import numpy as np, sys
w = sys.argv[1]
a = np.fromfile(f'gpurun_out/r06_m/pt_{w}.bin', dtype=np.uint32).reshape(-1, 4)
t = np.fromfile(f'gpurun_out/r06_m/pt_{w}.bin.tp', dtype=np.uint64).reshape(-1, 8).astype(np.int64)
n = 4096
ok = (t[:n, 0] > 0) & (t[:n, 7] > t[:n, 0])
print(w, 'lanes with points', int(ok.sum()))
for sh in np.unique(a[:n, 3][ok]):
    m = ok & (a[:n, 3] == sh)
    d = t[:n][m]
    base = d[:, 0:1]
    rel = d - base
    rel[rel < 0] = 0
    print(' shader', sh, 'lanes', int(m.sum()), 'median time points (10 ns) 0..7:', np.median(rel, axis=0).astype(int), ' vertex/post/mid:', np.median(a[:n][m][:, :3], axis=0).astype(int))
PY
done
