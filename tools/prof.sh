#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   -- runs on the GPU box via gpurun
tag=$1; shift
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o r -- python bench.py --no-cpu-baseline "$@" > gpurun_out/$tag/bench.log 2>&1
grep '"metric"' gpurun_out/$tag/bench.log
find gpurun_out/$tag -name "*kernel_stats.csv" | head -1 | xargs cat
