cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r05_q6
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -4) > gpurun_out/r05_q6/gpu_tests.log 2>&1
cat gpurun_out/r05_q6/gpu_tests.log
B=webrender_amd/csrc/ab/libwrhip_base.so
bash tools/ab_mix.sh r05_q6 "transforms" "base:lib=$B" "new" "new_fuse:WRHIP_FUSE_SMALL=1"
bash tools/ab_mix.sh r05_q6 "cfg3 many-images" "base:lib=$B" "new"
