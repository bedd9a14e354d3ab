cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_b
bash tools/rows_times.sh r06_b ab/libwrhip_rt.so many-box-shadows > gpurun_out/r06_b/rows_mbs.txt 2>&1; cat gpurun_out/r06_b/rows_mbs.txt
bash tools/rows_times.sh r06_b ab/libwrhip_rt.so cfg4 > gpurun_out/r06_b/rows_cfg4.txt 2>&1; cat gpurun_out/r06_b/rows_cfg4.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "box or shadow or cfg4 or mask" 2>&1 | tail -3
for w in cfg4 many-box-shadows large-boxshadow-ellipse large-boxshadow-ellipse-2; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/r06_b/bench_$w.json
  python3 -c "
import json
d = json.load(open('gpurun_out/r06_b/bench_$w.json')); r = d.get('roofline') or {}
print('$w', 'fps', d['value'], 'kernel_us', r.get('kernel_us_per_frame'), '|', ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
