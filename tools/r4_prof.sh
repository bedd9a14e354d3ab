#!/bin/bash
# usage: tools/r4_prof.sh <tag> <prof lib> [workload]  -- region timers of a -DWR_PROF build (WRHIP_PROF=1) over a short bench run
tag=$1; lib=$2; w=${3:-cfg4}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
WRHIP_PROF=1 WRHIP_LIB_PATH=$PWD/$lib python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "prof region|metric" | cut -c1-200 | tee gpurun_out/$tag/prof_$w.txt
