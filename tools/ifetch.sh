#!/bin/bash
# usage: tools/ifetch.sh <tag> [workload] -- instruction-fetch counters per kernel: is a short dependent launch waiting for its own code?
tag=$1; w=${2:-cfg4}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  mkdir -p gpurun_out/${tag}_if$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/${tag}_if$i -o r -- python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 > gpurun_out/${tag}_if$i/bench.log 2>&1
done
python3 - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    f = glob.glob("gpurun_out/${tag}_if%d/*counter_collection.csv" % i)
    if not f: continue
    for row in csv.DictReader(open(f[0])):
        agg[row["Kernel_Name"].split("(")[0][:50] + " grid=" + row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, v in agg.items():
    out[k] = {c: round(sum(x) / len(x), 1) for c, x in v.items()}
    out[k]["launches"] = len(next(iter(v.values())))
json.dump(out, open("gpurun_out/${tag}_ifetch_$w.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
