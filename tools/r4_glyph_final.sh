#!/bin/bash
# the GPU parity suite with the glyph-record lane walk, then cfg3 / text-rendering bench lines, cfg3 kernel stats, HBM traffic and SQ counters
tag=r04_f
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
bash tools/r4_tests.sh 2>&1 | tail -6
for w in cfg3 text-rendering; do python bench.py --workload $w --steps 50 --warmup 5 2>/dev/null | grep metric > gpurun_out/$tag/bench_$w.json; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -o r -- python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp gpurun_out/$tag/prof/r_kernel_stats.csv gpurun_out/$tag/cfg3_kernel_stats.csv; rm -rf gpurun_out/$tag/prof
bash tools/pmc_hbm.sh ${tag}_pmc_hbm_cfg3 cfg3 2>&1 | tail -6
WORKLOADS="cfg3" bash tools/round2_sq.sh ${tag} > /dev/null 2>&1
python3 -c "
import json
for w in ('cfg3','text-rendering'):
    d=json.loads(open('gpurun_out/$tag/bench_%s.json'%w).read()); r=d['roofline']; print(w, d['value'], 'fps', r['kernel'], r['avg_launch_us'], r['frac'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))"
head -5 gpurun_out/$tag/cfg3_kernel_stats.csv
