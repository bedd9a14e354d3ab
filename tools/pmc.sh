#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args]  -- PMC pass (kernel-trace only, no other trace domains)
tag=$1; ctr=$2; shift; shift
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/$tag -o r -- python bench.py --no-cpu-baseline "$@" > gpurun_out/$tag/bench.log 2>&1
ls gpurun_out/$tag
python - <<PY
import csv, collections, glob
f = glob.glob("gpurun_out/$tag/*counter_collection.csv")
if not f: raise SystemExit("no counter csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    k = (row["Kernel_Name"][:40], row["Grid_Size"])
    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x)/len(x), 1) for c, x in v.items()}, "n", len(next(iter(v.values()))))
PY
