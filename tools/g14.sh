cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_v
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -2
bash tools/ab_env.sh r06_v "clip-clear large-clip-rect cfg4 many-box-shadows large-boxshadow-ellipse aligned-gradient large-blur-radius" "WRHIP_NO_TILE_ROWS_LIGHT=1" 2>&1 | cut -c1-300
