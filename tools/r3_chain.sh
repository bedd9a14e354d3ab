#!/bin/bash
# usage: tools/r3_chain.sh <tag>  -- chained thin R8 levels (wr_raster_chain_kernel) A/B on cfg4: parity cases that run mask chains, then
# frames/s and the kernel timeline with the chain on, off (WRHIP_CHAIN unset) and at other grid sizes
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(WRHIP_CHAIN=1 timeout 600 python -m pytest tests -m gpu -q -x -k "cfg4 or blur or box_shadow or clip_rect or scale or mask" 2>&1 | tail -5) > gpurun_out/$tag/tests.log 2>&1
cat gpurun_out/$tag/tests.log
for i in 1 2; do
  for v in on off; do
    if [ $v = on ]; then export WRHIP_CHAIN=1; else unset WRHIP_CHAIN; fi
    echo "chain $v: $(timeout 300 python bench.py --workload cfg4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['frame_latency_ms'])")"
  done
done
export WRHIP_CHAIN=1
for g in 64 192; do
  echo "chain grid $g: $(WRHIP_CHAIN_GRID=$g timeout 300 python bench.py --workload cfg4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['frame_latency_ms'])")"
done
bash tools/r3_cfg4.sh $tag/kt_on | tail -40
