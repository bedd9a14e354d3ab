#!/bin/bash
# usage: tools/r4_setup_modes.sh [workload] -- what the setup stage's duration is made of: the timing build's experiment switch
# (WRHIP_SETUP_MODE: 0 whole stage, 1 empty kernel, 2 vertex stage only, 3 no binning), standalone setup launches event-timed
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out
w=${1:-cfg2}
for m in 0 1 2 3 0; do
  WRHIP_SETUP_MODE=$m WRHIP_LIB_PATH=$PWD/ab/libwrhip_tm.so python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('mode $m', ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel',''), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', []) if 'setup' in k['name']))"
done
