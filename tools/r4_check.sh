#!/bin/bash
# usage: tools/r4_check.sh <tag>  -- on the GPU box: the GPU parity suite, then one bench line per BASELINE config (cfg2 with the swgl baseline beside it)
tag=${1:-r04_e}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
cat gpurun_out/$tag/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_cfg2.json 2> gpurun_out/$tag/bench_cfg2.err
for w in cfg3 cfg4 cfg5; do
  python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric > gpurun_out/$tag/bench_$w.json
done
grep -h metric gpurun_out/$tag/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print(d['config']['workload'][:12], 'fps', d['value'], 'lat_ms', d.get('frame_latency_ms'), 'dom', r.get('kernel'), r.get('avg_launch_us'), 'frac', r.get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    for k in r.get('per_kernel', []): print('    ', k['name'], k['launches_per_frame'], 'x', k['us'], 'us  algo', k['algo_bytes'], 'frac', k['frac'])"
