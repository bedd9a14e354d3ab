#!/bin/bash
# usage: tools/r4_glyph_ab.sh -- GPU parity suite with the glyph-record lane walk, then A/B against the previous build (cfg3, cfg4, text-rendering)
# and the fused rect kernel at 4 / 5 / 6 waves per SIMD (cfg2)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out
bash tools/r4_tests.sh 2>&1 | tail -8
L=webrender_amd/csrc/libwrhip.so
(bash tools/ab.sh cfg3 ab/libwrhip_base.so $L; bash tools/ab.sh text-rendering ab/libwrhip_base.so $L; bash tools/ab.sh cfg4 ab/libwrhip_base.so $L; bash tools/ab.sh cfg2 $L ab/libwrhip_w5.so ab/libwrhip_w6.so) 2>&1 | tee gpurun_out/r04_f_glyph_ab.txt
