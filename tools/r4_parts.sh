#!/bin/bash
# usage: tools/r4_parts.sh <tag>  -- cfg4 (and the blur / box-shadow GPU tests) with 1 / 2 / 4 / 8 / 16 workgroups per bin in the thin R8 launches
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
for round in 1 2; do
for p in 1 4 16 2 8; do
  WRHIP_THIN_PARTS=$p python bench.py --workload cfg4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('parts $p fps', d['value'], ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))"
done; done 2>&1 | tee gpurun_out/$tag/parts.txt
for p in 4 16; do WRHIP_THIN_PARTS=$p timeout 600 python -m pytest tests -m gpu -q -k "blur or cfg4 or box_shadow or many_box or scale or clip" 2>&1 | tail -2; done | tee -a gpurun_out/$tag/parts.txt
