cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_i
# setup-stage phase times (wall_clock64: 10 ns units) of cfg2, fused (throughput mode) -- WRHIP_TIMING build
WRHIP_LIB_PATH=$PWD/ab/libwrhip_timing.so WRHIP_DEBUG_COUNTERS=1 WRHIP_PRIM_TIMES=$PWD/gpurun_out/r06_i/prim_times_cfg2.bin python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r06_i/timing_cfg2.err | grep '"metric"' > gpurun_out/r06_i/timing_cfg2.json
grep "dbg counters" gpurun_out/r06_i/timing_cfg2.err | tail -12
for m in 1 2 3; do
echo "setup mode $m"
WRHIP_SETUP_MODE=$m WRHIP_LIB_PATH=$PWD/ab/libwrhip_timing.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('fps', d['value'], ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
python3 - <<'PY'
import numpy as np
a = np.fromfile('gpurun_out/r06_i/prim_times_cfg2.bin', dtype=np.uint32).reshape(-1, 4)
n = 1024
print('vertex (10ns): median', np.median(a[:n,0]), 'max', a[:n,0].max(), ' post-vertex:', np.median(a[:n,1]), a[:n,1].max(), ' to mid:', np.median(a[:n,2]), a[:n,2].max(), 'shaders', np.unique(a[:n,3]))
for w in range(0, n, 64): print(w, a[w:w+64,0].max(), a[w:w+64,1].max(), a[w:w+64,2].max(), a[w,3])
PY
