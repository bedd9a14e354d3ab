# usage: tools/g3.sh <tag> "<workloads>" [tests]  -- bench lines (no CPU baseline) of the workloads, optionally the GPU suite first
tag=$1; wls=$2; tests=$3
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
if [ -n "$tests" ]; then
(time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8) > gpurun_out/$tag/gpu_tests.log 2>&1
cat gpurun_out/$tag/gpu_tests.log
fi
for w in $wls; do
  steps=50; [ $w = cfg2 ] && steps=20
  python bench.py --workload $w --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_$w.json
  python3 -c "
import json
d = json.load(open('gpurun_out/$tag/bench_$w.json')); r = d.get('roofline') or {}
print('$w', 'fps', d['value'], 'host', (d.get('host') or {}).get('wall'), 'busy', d.get('gpu_busy_frac'), 'kernel_us', r.get('kernel_us_per_frame'), 'frac', r.get('frac'), '|', ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
