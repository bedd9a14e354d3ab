#!/bin/bash
# usage: tools/r3_cells.sh <tag>  -- on the GPU box: parity suite with the cell raster on, then cfg1/2/5 bench lines with it on / off
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
cat gpurun_out/$tag/gpu_tests.log
summ() { python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('$1', d['config']['workload'][:5], 'fps', d['value'], 'lat', d['frame_latency_ms'], 'host', d.get('host'), ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"; }
for round in 1 2; do
for w in cfg2 cfg1 cfg5; do
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | tee -a gpurun_out/$tag/bench_${w}_cells.json | summ cells
  WRHIP_NO_CELLS=1 timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | tee -a gpurun_out/$tag/bench_${w}_nocells.json | summ pixel
done
done
timeout 300 python bench.py --encoding brush --no-cpu-baseline 2>/dev/null | grep metric | tee gpurun_out/$tag/bench_cfg2_brush.json | summ brush
