#!/bin/bash
# usage: tools/r3_env_ab.sh <tag> <ENVVAR> <workloads...>  -- bench lines with and without ENVVAR=1, interleaved, three rounds
tag=$1; var=$2; shift; shift
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
summ() { python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}; h = d.get('host') or {}
    print('$1', d['config']['workload'][:5], 'fps', d['value'], 'lat', d['frame_latency_ms'], 'host wall %s rec %s stage %s flush %s blocked %s other %s |' % (h.get('wall'), h.get('record_draws'), h.get('stage_uploads'), h.get('flush_and_launch'), h.get('blocked_on_stream'), h.get('other_calls_and_replayer')), ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"; }
for round in 1 2 3; do
  for w in "$@"; do
    timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | summ "default "
    env $var=1 timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | summ "$var"
  done
done
