#!/bin/bash
# usage: tools/sq.sh <tag> -- SQ counters (waves, cycles, instruction mix, wait) of cfg2 / cfg3 / cfg4, one json per workload
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for w in ${WORKLOADS:-cfg2 cfg3 cfg4}; do
  mkdir -p gpurun_out/${tag}_sq_$w
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/${tag}_sq_$w -o r -- python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 > gpurun_out/${tag}_sq_$w/bench.log 2>&1
  python3 - <<PY
import csv, collections, glob, json
f = glob.glob("gpurun_out/${tag}_sq_$w/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    agg[row["Kernel_Name"].split("(")[0] + " grid=" + row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"workload": "$w (bench.py --workload $w --steps 10 --warmup 3), rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_LDS; per-launch means; SQ cycle counters are quad-cycles", "kernels": {}}
for k, v in agg.items():
    out["kernels"][k] = {c: round(sum(x) / len(x), 1) for c, x in v.items()}
    out["kernels"][k]["launches"] = len(next(iter(v.values())))
json.dump(out, open("gpurun_out/${tag}_pmc_sq_$w.json", "w"), indent=1)
PY
done
ls gpurun_out/${tag}_pmc_sq_*.json
