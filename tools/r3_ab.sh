#!/bin/bash
# usage: tools/r3_ab.sh <tag> <libs...>  -- GPU parity suite with the in-tree library, then cfg2 / cfg5 / cfg1 A/B of the given libraries
tag=$1; shift
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
tail -5 gpurun_out/$tag/gpu_tests.log
for w in ${WORKLOADS:-cfg2 cfg5 cfg1}; do bash tools/ab.sh $w "$@" 2>&1 | tee -a gpurun_out/$tag/ab_$w.log; done
