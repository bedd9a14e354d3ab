#!/bin/bash
# usage: tools/r4_ab.sh  -- A/B of the setup stage on its own stream (default) vs the fused kernels (WRHIP_SETUP_STREAM=0), every workload
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4_ab.txt; : > $out
python -m pytest tests/test_gpu_parity.py -q -x -k "staging_ring or cfg2 or pipelined" 2>&1 | tail -3 | tee -a $out
for wl in cfg2 cfg3 cfg4 cfg5; do
  for mode in 1 0; do
    for rep in 1 2; do
      WRHIP_SETUP_STREAM=$mode python bench.py --no-cpu-baseline --workload $wl --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline') or {}
        print('$wl setup_stream=$mode', d['value'], 'fps', 'host', d['host']['wall'], 'kern', r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), 'gpu_us', r.get('kernel_us_per_frame'))
" | tee -a $out
    done
  done
done
