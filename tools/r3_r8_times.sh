#!/bin/bash
# usage: tools/r3_r8_times.sh <tag> <timing lib>  -- per-wave phase timestamps of the R8 launches of a cfg4 frame (debug build, -DWR_R8_TIMING)
tag=$1; lib=$2
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
WRHIP_R8_TIMES=$PWD/gpurun_out/$tag/r8.bin WRHIP_LIB_PATH=$PWD/$lib timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
python3 - <<PY
import numpy as np
t = np.fromfile("gpurun_out/$tag/r8.bin", dtype=np.uint64).reshape(32, 256, 16, 8).astype(np.int64)
for ti in range(32):
    a = t[ti]
    ok = (a[..., 0] > 0) & (a[..., 4] > 0)
    if not ok.any(): continue
    s0, s1, s2, s3, s4 = [a[..., i][ok] for i in range(5)]
    had = s2 > 0
    span = s4.max() - s0.min()
    print("target %2d: waves %5d with prims %5d | first stamp .. last end %7d cyc | init %6.0f | end-start median %7.0f max %7d" % (ti, ok.sum(), had.sum(), span, np.median(s1 - s0), np.median(s4 - s0), (s4 - s0).max()))
    if had.any():
        print("            with prims: init->first apply %7.0f | last apply %7.0f (max %7d) | after %6.0f | start spread %7d" % (
            np.median((s2 - s1)[had]), np.median((s3 - s2)[had]), (s3 - s2)[had].max(), np.median((s4 - s3)[had]), s0.max() - s0.min()))
PY
