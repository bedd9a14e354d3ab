#!/bin/bash
# usage: tools/lds_sweep.sh <tag> <workload> -- bench the workload with different dynamic-LDS residency caps of the rect kernel
tag=$1; wl=$2
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag
for lds in 0 20000 26000 40000 53000 65536; do
  WRHIP_RECT_LDS=$lds python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep metric > gpurun_out/$tag/lds_$lds.json
  python3 -c "
import json; d=json.load(open('gpurun_out/$tag/lds_$lds.json')); r=d['roofline']
print('lds', $lds, 'fps', d['value'], 'dom', r['kernel'], r['avg_launch_us'], [ (k['name'][-12:],k['us']) for k in r['per_kernel']])"
done
