#!/bin/bash
# usage: tools/round4.sh <tag>  -- the round's measurement pass on the GPU box: parity suite, bench lines of every BASELINE config with the
# swgl CPU baseline beside them, one line per wrench benchmark workload, the N > 1 path's world-1 self-test, rocprofv3 kernel stats,
# HBM traffic (separate FETCH / WRITE passes) and SQ counters of cfg2 / cfg3 / cfg5
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
cat gpurun_out/$tag/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_cfg2.json 2> gpurun_out/$tag/bench_cfg2.err
for w in cfg1 cfg3 cfg4 cfg5; do
  python bench.py --workload $w --steps 50 --warmup 5 2>/dev/null | grep metric > gpurun_out/$tag/bench_$w.json
done
python bench.py --workload transforms --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric > gpurun_out/$tag/bench_transforms.json
python bench.py --sharded --steps 20 --warmup 3 2>/dev/null | grep metric > gpurun_out/$tag/bench_sharded_world1_cfg5.json
for wl in large-blur-radius large-clip-rect large-boxshadow-ellipse many-images aligned-gradient unaligned-gradient text-rendering many-box-shadows simple-batching; do
  python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_${wl}.json
done
for w in cfg2 cfg3 cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  cp gpurun_out/$tag/prof_$w/r_kernel_stats.csv gpurun_out/$tag/${w}_kernel_stats.csv
  rm -rf gpurun_out/$tag/prof_$w
done
for w in cfg2 cfg3 cfg5; do bash tools/pmc_hbm.sh ${tag}_pmc_hbm_$w $w > gpurun_out/$tag/pmc_hbm_$w.log 2>&1; done
WORKLOADS="cfg2 cfg3 cfg5" bash tools/round2_sq.sh ${tag} > gpurun_out/$tag/sq.log 2>&1
grep -h metric gpurun_out/$tag/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print(d['config']['workload'][:24], 'fps', d['value'], 'lat_ms', d.get('frame_latency_ms'), 'dom', r.get('kernel'), r.get('avg_launch_us'), 'frac', r.get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), ((d.get('cpu_baseline') or {}).get('multi_process') or {}).get('value'))"
