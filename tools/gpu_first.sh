#!/bin/bash
# usage: tools/gpu_first.sh <tag> -- round 5, first GPU contact: the new tests + the new workload's bench line
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 900 python -m pytest tests/test_clang_budget.py tests/test_gpu_parity.py -m gpu -q -k "shipping or wrench" 2>&1 | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
cat gpurun_out/$tag/gpu_tests.log
python bench.py --workload large-boxshadow-ellipse-2 --steps 40 --warmup 5 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_large-boxshadow-ellipse-2.json
python3 -c "
import json
d=json.load(open('gpurun_out/$tag/bench_large-boxshadow-ellipse-2.json')); r=d['roofline']
print(d['value'], r['kernel'], r['avg_launch_us'], r['frac'], d['cpu_baseline']['value'])
for k in r['per_kernel']: print(k['name'], k['launches_per_frame'], k['us'], k['workgroups'], k['algo_bytes'])"
