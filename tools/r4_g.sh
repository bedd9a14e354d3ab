#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r04_g
L=webrender_amd/csrc/libwrhip.so
bash tools/r4_tests.sh 2>&1 | tail -5
bash tools/ab.sh cfg3 $L ab/libwrhip_d3.so 2>&1 | tee gpurun_out/r04_g/dense_waves_ab.txt
for w in cfg5 cfg2; do
  for k in 1 2; do python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric > gpurun_out/r04_g/bench_${w}_$k.json
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_g/bench_${w}_$k.json').read()); print('$w', d['value'], 'fps host', d['host'])"; done
done
