#!/bin/bash
# usage: tools/r3_ring.sh <tag>  -- staging ring reuse by fences vs a full drain per lap (WRHIP_RING_DRAIN=1), on the workloads that
# stage the most per frame; the cfg5 / upload-heavy parity cases first; then the 4-waves-per-SIMD build of the textured variants on cfg3
tag=$1
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(timeout 900 python -m pytest tests -m gpu -q -x -k "cfg5 or shadows or dual or abi_surface or capture or sweep or dist" 2>&1 | tail -4) > gpurun_out/$tag/tests.log 2>&1
cat gpurun_out/$tag/tests.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host']; print(d['value'], d['ms_per_step'], 'blocked', h['blocked_on_stream'], 'stage', h['stage_uploads'], 'flush', h['flush_and_launch'])"; }
for w in cfg5 cfg3 cfg2; do
  for i in 1 2; do
    for v in fences drain; do
      if [ $v = drain ]; then export WRHIP_RING_DRAIN=1; else unset WRHIP_RING_DRAIN; fi
      echo "$w ring $v: $(timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | line)"
    done
  done
done
unset WRHIP_RING_DRAIN
bash tools/ab.sh cfg3 ab/libwrhip_base.so ab/libwrhip_w4.so 2>&1 | cut -c1-200
