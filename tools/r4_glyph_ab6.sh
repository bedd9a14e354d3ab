#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out
L=webrender_amd/csrc/libwrhip.so
(bash tools/ab.sh text-rendering ab/libwrhip_base.so $L ab/libwrhip_al.so; bash tools/ab.sh cfg3 ab/libwrhip_base.so $L ab/libwrhip_al.so) 2>&1 | tee gpurun_out/r04_f_glyph_ab6.txt
