#!/bin/bash
# usage: tools/r4_wrench.sh <tag>  -- GPU parity of the wrench benchmark workloads at 4K, then one bench line each (profiles/<tag>_bench_<workload>.json)
tag=${1:-r04_b}
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "wrench" 2>&1 | tail -3 | tee gpurun_out/${tag}_wrench_tests.log
for wl in large-blur-radius large-clip-rect large-boxshadow-ellipse many-images aligned-gradient unaligned-gradient text-rendering many-box-shadows simple-batching; do
  python bench.py --workload $wl --steps 40 --warmup 5 2>/dev/null | grep '"metric"' > gpurun_out/${tag}_bench_${wl}.json
  python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_${wl}.json')); r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}
print('$wl', d['value'], 'fps', 'kern', r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), 'cpu', c.get('value'), c.get('unit'))"
done
