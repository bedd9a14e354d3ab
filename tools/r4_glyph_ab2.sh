#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out
bash tools/r4_tests.sh 2>&1 | tail -6
L=webrender_amd/csrc/libwrhip.so
(bash tools/ab.sh cfg3 ab/libwrhip_base.so $L; bash tools/ab.sh text-rendering ab/libwrhip_base.so $L) 2>&1 | tee gpurun_out/r04_f_glyph_ab2.txt
