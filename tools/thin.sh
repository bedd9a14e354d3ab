#!/bin/bash
# usage: tools/thin.sh  -- on the GPU box: cfg4 with and without the thin (16-wave) R8 workgroups, parity of the mask scenes
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/thin
(timeout 900 python -m pytest tests -m gpu -q -k "cfg4 or blur or shadow or clip or mask" 2>&1 | tail -5) > gpurun_out/thin/tests.log 2>&1
cat gpurun_out/thin/tests.log
for v in "" "WRHIP_NO_MASK_ROWS=1"; do
  echo "== $v"
  env $v python bench.py --workload cfg4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('fps', d['value'], 'lat_ms', d['frame_latency_ms'])
    for k in r.get('per_kernel', []): print('    ', k['name'], k['launches_per_frame'], 'x', k['us'], 'us')"
done
