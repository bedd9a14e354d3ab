cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_s
: > gpurun_out/r06_s/gpu_sweep.txt
for s in $(seq 7001 7040); do
  WRHIP_SWEEP_SEED=$s WRHIP_SWEEP_SECONDS=240 timeout 600 python -m pytest tests/test_gpu_sweep.py -m gpu -q -s 2>&1 | grep "WRHIP_SWEEP_SEED\|failed\|differ\|skipped" | tee -a gpurun_out/r06_s/gpu_sweep.txt
done
grep -c "0 failures" gpurun_out/r06_s/gpu_sweep.txt
