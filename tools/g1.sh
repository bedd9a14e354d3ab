cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_f
bash tools/rows_times.sh r06_f ab/libwrhip_rt.so many-box-shadows 2>&1 | grep -v 'slow row' | tail -12
bash tools/rows_times.sh r06_f ab/libwrhip_rt.so cfg4 2>&1 | grep -v 'slow row' | tail -12
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "box or shadow or cfg4 or mask or clip" 2>&1 | tail -3
for w in cfg4 many-box-shadows large-boxshadow-ellipse large-boxshadow-ellipse-2 large-clip-rect clip-clear; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/r06_f/bench_$w.json
  python3 -c "
import json
d = json.load(open('gpurun_out/r06_f/bench_$w.json')); r = d.get('roofline') or {}
print('$w', 'fps', d['value'], 'kernel_us', r.get('kernel_us_per_frame'), '|', ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
