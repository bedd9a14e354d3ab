#!/bin/bash
# usage: tools/round2_pmc.sh <tag> -- HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of cfg2..cfg5, one json per workload
tag=$1
for w in cfg2 cfg3 cfg4 cfg5; do bash tools/pmc_hbm.sh ${tag}_pmc_hbm_$w $w > /dev/null 2>&1; done
ls gpurun_out/${tag}_pmc_hbm_*.json
