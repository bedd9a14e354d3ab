#!/bin/bash
# usage: tools/r3_copy_ab.sh  -- A/B of the split staging copies on one box: cfg2 / cfg3 / cfg5 with the split off and on, interleaved
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for rep in 1 2 3; do
  for mode in off on; do
    if [ $mode = off ]; then export WRHIP_COPY_SPLIT_MIN=99999999; else unset WRHIP_COPY_SPLIT_MIN; fi
    for w in cfg2 cfg3; do
      python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); h = d['host']
print('$mode $w rep$rep fps %8.0f  wall %6.1f stage %5.1f record %5.1f flush %5.1f blocked %5.1f other %5.1f' % (d['value'], h['wall'], h['stage_uploads'], h['record_draws'], h['flush_and_launch'], h['blocked_on_stream'], h['other_calls_and_replayer']))"
    done
  done
done
