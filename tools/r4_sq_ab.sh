#!/bin/bash
# usage: tools/r4_sq_ab.sh <workload> <lib>... -- SQ counters of one workload per library build (per-launch means)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
w=$1; shift
for lib in "$@"; do
  n=$(basename $lib .so); d=gpurun_out/sqab_${w}_$n; mkdir -p $d
  WRHIP_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $d -o r -- python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 > $d/bench.log 2>&1
  python3 - <<PY
import csv, collections, glob
f = glob.glob("$d/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    agg[row["Kernel_Name"].split("(")[0][:60] + " grid=" + row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    if "dense" in k or "raster" in k: print("$n", k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n", len(next(iter(v.values()))))
PY
  rm -rf $d
done
