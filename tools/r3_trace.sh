#!/bin/bash
# usage: tools/r3_trace.sh <tag> <workload>  -- rocprofv3 kernel trace (timestamps) of a short streamed run
tag=$1; w=$2
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/t_$w -o r -- python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 5 > gpurun_out/$tag/trace_$w.log 2>&1
cp gpurun_out/$tag/t_$w/r_kernel_trace.csv gpurun_out/$tag/${w}_kernel_trace.csv; rm -rf gpurun_out/$tag/t_$w
ls -la gpurun_out/$tag
