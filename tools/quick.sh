#!/bin/bash
# usage: tools/quick5.sh <tag> "<workloads>" -- GPU suite + one bench line per workload (no CPU baseline), summary on stdout
tag=$1; wls=$2
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) > gpurun_out/$tag/gpu_tests.log 2>&1
cat gpurun_out/$tag/gpu_tests.log
for w in $wls; do
  python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_$w.json
  python3 -c "
import json
d = json.load(open('gpurun_out/$tag/bench_$w.json')); r = d.get('roofline') or {}
print('$w', 'fps', d['value'], 'kernel_us', r.get('kernel_us_per_frame'), 'dom', r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), '|', ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
