#!/bin/bash
# usage: tools/ab.sh <workload> <libA> <libB> ...  -- on ONE GPU box: the same bench line with each library build, interleaved, three rounds
# (latency-bound launches differ by tens of percent between boxes: variants are only comparable within one call)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
w=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    WRHIP_LIB_PATH=$PWD/$lib python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('$lib'.split('/')[-1], 'fps', d['value'], ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
  done
done
