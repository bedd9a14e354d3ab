#!/bin/bash
# usage: tools/stats.sh <tag> [bench args]  -- kernel-trace stats of a bench run, compact (on the GPU box via gpurun)
tag=$1
bash tools/prof.sh "$@" > /dev/null
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/$tag/r_kernel_stats.csv')): print(r['Name'][:44].ljust(44), r['Calls'].rjust(6), '%9.1f us avg' % (float(r['AverageNs'])/1e3), '%9.1f min' % (float(r['MinNs'])/1e3))"
grep metric gpurun_out/$tag/bench.log | cut -c1-330
