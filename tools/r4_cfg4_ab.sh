#!/bin/bash
# usage: tools/r4_cfg4_ab.sh <tag> <lib>...  -- one steady cfg4 frame's kernel durations (rocprofv3 kernel trace) per library build, on one box
tag=$1; shift
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
for lib in "$@"; do
  n=$(basename $lib .so)
  WRHIP_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/$n -o kt -- python bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/$n.log 2>&1
  WRHIP_LIB_PATH=$PWD/$lib python bench.py --workload cfg4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "
import sys, json
for l in sys.stdin: print('$n fps', json.loads(l)['value'])"
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$tag/$n/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))]
rows.sort()
# the last complete frame: from the last-but-one wr_upload_kernel to the last one
ups = [i for i, r in enumerate(rows) if r[2].startswith("wr_upload")]
a, b = ups[-3], ups[-2]
short = lambda n: n.replace("void ", "").replace("wr_", "").replace("_kernel", "").replace("raster", "r").replace(", false", "F").replace(", true", "T").replace("__amd_rocclr_", "")
print("$n", " ".join("%s:%.1f" % (short(n), (e - s) / 1e3) for s, e, n in rows[a:b]), "| sum %.1f" % sum((e - s) / 1e3 for s, e, n in rows[a:b]))
PY
done 2>&1 | tee gpurun_out/$tag/summary.txt
