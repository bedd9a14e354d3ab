#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r04_h
L=webrender_amd/csrc/libwrhip.so
(bash tools/ab.sh cfg2 $L ab/libwrhip_p3.so ab/libwrhip_r4.so; bash tools/ab.sh cfg5 $L ab/libwrhip_p3.so; bash tools/ab.sh cfg3 $L ab/libwrhip_p3.so) 2>&1 | tee gpurun_out/r04_h/setup_prio_ab.txt
