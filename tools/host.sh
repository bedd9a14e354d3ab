#!/bin/bash
# usage: tools/host.sh  -- on the GPU box: the host-side phase breakdown of every BASELINE config (bench.py "host" object)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for w in cfg1 cfg2 cfg3 cfg4 cfg5; do
  python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$w', 'fps', d['value'], 'us/frame', round(1e3 * d['ms_per_step'], 1), d.get('host'))"
done
