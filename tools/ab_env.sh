#!/bin/bash
# usage: tools/ab_env.sh <tag> "<workloads>" "<ENV=1 ...>"  -- on ONE GPU box: every workload's bench line with the library's defaults and with
# the given environment switches (an optimisation turned off), interleaved, three rounds: the A/B figures of DESIGN section 5
tag=$1; wls=$2; envs=$3
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
for w in $wls; do
  for round in 1 2 3; do
    for variant in default off; do
      if [ $variant = off ]; then pre="env $envs"; else pre=""; fi
      $pre python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('$w', '$variant', 'fps', d['value'], 'kernel_us', r.get('kernel_us_per_frame'), ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))" | tee -a gpurun_out/$tag/ab.txt
    done
  done
done
