#!/bin/bash
# usage: tools/r3_misc.sh <tag>  -- GPU suite, the N > 1 bench path at world 1, submit-thread spin variants, default bench line
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
tail -5 gpurun_out/$tag/gpu_tests.log
echo "--- sharded self-test (cfg5, world 1)"
timeout 600 python bench.py --sharded --steps 20 --warmup 3 2> gpurun_out/$tag/sharded.err | grep metric | tee gpurun_out/$tag/bench_sharded.json | cut -c1-1500
tail -3 gpurun_out/$tag/sharded.err
summ() { python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}; h = d.get('host') or {}
    print('$1', d['config']['workload'][:5], 'fps', d['value'], 'lat', d['frame_latency_ms'], 'host wall %s rec %s stage %s flush %s blocked %s other %s |' % (h.get('wall'), h.get('record_draws'), h.get('stage_uploads'), h.get('flush_and_launch'), h.get('blocked_on_stream'), h.get('other_calls_and_replayer')), ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"; }
for round in 1 2; do
  for spin in 500 20000 50; do
    WRHIP_SUBMIT_SPIN=$spin timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | summ spin$spin
  done
done
python bench.py --steps 20 --warmup 5 2>/dev/null | grep metric > gpurun_out/$tag/bench_default.json; cut -c1-600 gpurun_out/$tag/bench_default.json
