#!/bin/bash
# usage: tools/gpu_check.sh [bench args]  -- GPU parity suite, then the bench line (on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8 > gpurun_out/gpu_tests.log
cat gpurun_out/gpu_tests.log
WR_REPLAY_TIMING=1 python bench.py --no-cpu-baseline "$@" 2>&1 | grep "wr_replay\|metric" | tee gpurun_out/bench_last.log
