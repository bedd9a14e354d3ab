#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out
L=webrender_amd/csrc/libwrhip.so
bash tools/r4_tests.sh "text or cfg3 or glyph or masked or cfg4 or wrench" 2>&1 | tail -4
(bash tools/ab.sh text-rendering ab/libwrhip_base.so ab/libwrhip_g2c.so $L; bash tools/ab.sh cfg3 ab/libwrhip_base.so ab/libwrhip_g2c.so $L) 2>&1 | tee gpurun_out/r04_f_glyph_ab9.txt
