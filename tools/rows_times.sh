#!/bin/bash
# usage: tools/rows_times.sh <tag> <timing lib> <workload>  -- per-wave phase timestamps of the mask-rows launch (debug build of the
# "host" group with -DWR_ROWS_TIMING: tools/build_variant.sh rt "-DWR_ROWS_TIMING" host), on the GPU box
tag=$1; lib=$2; wl=${3:-cfg4}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
WRHIP_ROWS_TIMES=$PWD/gpurun_out/$tag/rows_$wl.bin WRHIP_LIB_PATH=$PWD/$lib timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
python3 - <<PY
import numpy as np
t = np.fromfile("gpurun_out/$tag/rows_$wl.bin", dtype=np.uint64).reshape(4096, 8)
ok = (t[:, 0] > 0) & (t[:, 6] > 0)
print("waves with a full record:", int(ok.sum()))
kind = (t[:, 7] >> 56).astype(int); y = ((t[:, 7] >> 32) & 0xFFFFFF).astype(int)
names = ["slot", "target/prim", "rowvals", "rowsetup", "key..dst", "lanes"]
for k in (0, 1):
    m = ok & (kind == k) & (t[:, 3] > 0 if k == 1 else ok)
    if not m.any(): continue
    T = t[m].astype(np.int64)
    if k == 1: d = np.stack([T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 4] - T[:, 3], T[:, 5] - T[:, 4], T[:, 6] - T[:, 5]], 1)
    else: d = np.stack([T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], 0 * T[:, 0], 0 * T[:, 0], T[:, 5] - T[:, 2], T[:, 6] - T[:, 5]], 1)
    tot = T[:, 6] - T[:, 0]
    print("kind", "box" if k else "clip", "rows", int(m.sum()), "ticks (100 MHz): median / p90 / max per phase")
    for i, n in enumerate(names): print("   %-12s %8.0f %8.0f %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
    print("   %-12s %8.0f %8.0f %8.0f" % ("total", np.median(tot), np.percentile(tot, 90), tot.max()))
    span = T[:, 6].max() - T[:, 0].min()
    print("   first start .. last end:", span, "ticks; starts spread:", T[:, 0].max() - T[:, 0].min())
    o = np.argsort(-tot)[:8]
    for j in o: print("   slow row y=%d phases %s" % (y[m][j], d[j].tolist()))
    # histogram of totals
    print("   total histogram (ticks):", np.histogram(tot, bins=[0, 100, 200, 400, 800, 1600, 3200, 6400, 12800, 25600, 1 << 30])[0].tolist())
PY
