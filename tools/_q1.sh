cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r05_q1
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8) > gpurun_out/r05_q1/gpu_tests.log 2>&1
cat gpurun_out/r05_q1/gpu_tests.log
B=webrender_amd/csrc/ab/libwrhip_base.so
bash tools/ab_mix.sh r05_q1 "transforms" "base:lib=$B" "new" "new_fuse:WRHIP_FUSE_SMALL=1" "new_noqtab:WRHIP_NO_QTAB=1"
bash tools/ab_mix.sh r05_q1 "cfg2 cfg3 cfg5 many-images" "base:lib=$B" "new"
for w in clip-clear overlapping-text-shadows; do
  python bench.py --workload $w --steps 40 --warmup 5 2>/dev/null | grep '"metric"' > gpurun_out/r05_q1/bench_$w.json
  python3 -c "
import json
d = json.load(open('gpurun_out/r05_q1/bench_$w.json')); r = d.get('roofline') or {}
print('$w', 'fps', d['value'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'dom', r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), '|', ' '.join('%s:%gx%.1f' % (k['name'], k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
