#!/bin/bash
# usage: tools/quick.sh  -- on the GPU box: one bench line per BASELINE config (no CPU baseline), compact
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for w in cfg1 cfg2 cfg3 cfg4 cfg5; do
  python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print(d['config']['workload'][:12], 'fps', d['value'], 'lat_ms', d['frame_latency_ms'], ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done
