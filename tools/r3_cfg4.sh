#!/bin/bash
# usage: tools/r3_cfg4.sh <tag>  -- kernel timeline (start / end timestamps) of steady cfg4 frames: durations and the gaps between the launches of the chain
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o kt -- python bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/bench.log 2>&1
grep metric gpurun_out/$tag/bench.log | cut -c1-200
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$tag/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 60 launches = a few steady frames
tail = rows[-75:-10]
t0 = tail[0][0]
prev = None
for s, e, n, g, w in tail:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%9.1f  dur %7.1f  gap %6.1f  %-60s grid %s wg %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n, g, w))
    prev = e
PY
