#!/bin/bash
# usage: tools/r4_tests.sh [pytest -k expression] -- the GPU parity suite (or a subset) on the GPU box
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
(time timeout 1500 python -m pytest tests -m gpu -q "${K[@]}" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -25) 2>&1 | tee gpurun_out/gpu_tests_last.log
