cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_g
(time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8) > gpurun_out/r06_g/gpu_tests.log 2>&1
cat gpurun_out/r06_g/gpu_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/r06_g/bench_cfg2_$i.json
python3 -c "
import json
d = json.load(open('gpurun_out/r06_g/bench_cfg2_$i.json')); r = d.get('roofline') or {}
print('cfg2', 'fps', d['value'], 'host', d.get('host'), 'busy', d.get('gpu_busy_frac'), 'host_bound', d.get('host_bound'), 'kernel_us', r.get('kernel_us_per_frame'), '|', ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
for w in cfg1 cfg3 cfg5; do
  python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/r06_g/bench_$w.json
  python3 -c "
import json
d = json.load(open('gpurun_out/r06_g/bench_$w.json')); r = d.get('roofline') or {}
print('$w', 'fps', d['value'], 'host', d.get('host'), 'busy', d.get('gpu_busy_frac'), 'kernel_us', r.get('kernel_us_per_frame'), '|', ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
