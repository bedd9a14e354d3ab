#!/bin/bash
# tests/test_gpu_sweep.py with 100 more unpinned seeds on the round's final library
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_zz
: > gpurun_out/r06_zz/gpu_sweep2.txt
for s in $(seq 9001 9100); do
  WRHIP_SWEEP_SEED=$s WRHIP_SWEEP_SECONDS=240 timeout 600 python -m pytest tests/test_gpu_sweep.py -m gpu -q -s 2>&1 | grep "WRHIP_SWEEP_SEED\|failed\|differ\|skipped" >> gpurun_out/r06_zz/gpu_sweep2.txt
done
grep -c "0 failures" gpurun_out/r06_zz/gpu_sweep2.txt; grep -v "0 failures" gpurun_out/r06_zz/gpu_sweep2.txt | head -20
