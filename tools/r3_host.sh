#!/bin/bash
# usage: tools/r3_host.sh <tag>  -- GPU parity suite, then every workload with the submit thread on / off (fps + host phase timers)
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
tail -5 gpurun_out/$tag/gpu_tests.log
summ() { python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}; h = d.get('host') or {}
    print('$1', d['config']['workload'][:5], 'fps', d['value'], 'lat', d['frame_latency_ms'], 'host wall %s rec %s stage %s flush %s blocked %s other %s |' % (h.get('wall'), h.get('record_draws'), h.get('stage_uploads'), h.get('flush_and_launch'), h.get('blocked_on_stream'), h.get('other_calls_and_replayer')), ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"; }
for round in 1 2; do
for w in ${WORKLOADS:-cfg2 cfg1 cfg3 cfg4 cfg5}; do
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | tee -a gpurun_out/$tag/bench_${w}.json | summ thread
  WRHIP_NO_SUBMIT_THREAD=1 timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | tee -a gpurun_out/$tag/bench_${w}_inline.json | summ inline
done
done
