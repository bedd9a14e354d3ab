#!/bin/bash
# usage: tools/r3_times.sh <tag> <timing lib> [workload]  -- per-workgroup phase timestamps of the cell raster (debug build, -DWR_CELL_TIMING)
tag=$1; lib=$2; w=${3:-cfg2}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
WRHIP_CELL_TIMES=$PWD/gpurun_out/$tag/times_$w.bin WRHIP_LIB_PATH=$PWD/$lib timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c1-300
ls -la gpurun_out/$tag/
