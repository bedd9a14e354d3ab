#!/bin/bash
# usage: tools/round.sh <tag>  -- on the GPU box: bench lines of every BASELINE config + rocprof kernel stats of cfg2/3/4
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench_cfg2.json 2> gpurun_out/$tag/bench_cfg2.err
for w in cfg1 cfg3 cfg4 cfg5; do
  python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric > gpurun_out/$tag/bench_$w.json
done
python bench.py --encoding brush --no-cpu-baseline 2>/dev/null | grep metric > gpurun_out/$tag/bench_cfg2_brush.json
for w in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  cp gpurun_out/$tag/prof_$w/r_kernel_stats.csv gpurun_out/$tag/${w}_kernel_stats.csv
  rm -rf gpurun_out/$tag/prof_$w
done
grep -h metric gpurun_out/$tag/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print(d['config']['workload'][:12], d['config']['encoding'], 'fps', d['value'], 'lat_ms', d['frame_latency_ms'], 'raster_us/frame', r.get('raster_us_per_frame'), 'frac', r.get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"
