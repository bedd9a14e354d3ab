cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_h
for pin in 0 1 0 1; do
if [ $pin = 0 ]; then export WRHIP_BENCH_NO_PIN=1; else unset WRHIP_BENCH_NO_PIN; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/r06_h/b.json
python3 -c "
import json
d = json.load(open('gpurun_out/r06_h/b.json')); r = d.get('roofline') or {}
print('cfg2 pin $pin', d.get('host_affinity'), 'fps', d['value'], 'regions', d['timed_regions_ms'], 'host', d.get('host'), 'busy', d.get('gpu_busy_frac'))"
done
uptime
