#!/bin/bash
# experiment: setup-kernel duration per WRHIP_SETUP_MODE (timing build), on the GPU box
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
cp webrender_amd/csrc/libwrhip_timing.so webrender_amd/csrc/libwrhip.so
for m in 0 1 2 3; do
  mkdir -p gpurun_out/modes_$m
  WRHIP_SETUP_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/modes_$m -o r -- python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/modes_$m/bench.log 2>&1
  python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/modes_$m/r_kernel_stats.csv')):
    if 'setup' in r['Name'] or 'upload' in r['Name']: print('mode $m', r['Name'][:20], '%9.1f us avg' % (float(r['AverageNs'])/1e3), '%9.1f min' % (float(r['MinNs'])/1e3))"
done
