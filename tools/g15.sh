#!/bin/bash
# the round's closing pass: tools/round.sh r06_zz cpu, then tests/test_gpu_sweep.py with 30 more unpinned seeds (the near-plane
# mix-blend families are in the draw since this build)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
bash tools/round.sh r06_zz cpu > gpurun_out/r06_zz_round.log 2>&1
: > gpurun_out/r06_zz/gpu_sweep.txt
for s in $(seq 8001 8030); do
  WRHIP_SWEEP_SEED=$s WRHIP_SWEEP_SECONDS=240 timeout 600 python -m pytest tests/test_gpu_sweep.py -m gpu -q -s 2>&1 | grep "WRHIP_SWEEP_SEED\|failed\|differ\|skipped" | tee -a gpurun_out/r06_zz/gpu_sweep.txt
done
grep -c "0 failures" gpurun_out/r06_zz/gpu_sweep.txt
tail -32 gpurun_out/r06_zz_round.log
