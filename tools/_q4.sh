cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r05_q4
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -4) > gpurun_out/r05_q4/gpu_tests.log 2>&1
cat gpurun_out/r05_q4/gpu_tests.log
B=webrender_amd/csrc/ab/libwrhip_base.so
bash tools/ab_mix.sh r05_q4 "aligned-gradient unaligned-gradient" "base:lib=$B" "new" "new_nogtab:WRHIP_NO_GTAB=1"
bash tools/ab_mix.sh r05_q4 "large-blur-radius large-clip-rect" "base:lib=$B" "new"
