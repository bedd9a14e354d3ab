cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_r
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -4)
for w in aligned-gradient unaligned-gradient clip-clear large-clip-rect many-box-shadows cfg4; do
  bash tools/ab.sh $w ab/libwrhip_base.so webrender_amd/csrc/libwrhip.so 2>&1 | sed "s/^/$w /" | cut -c1-330 | tee -a gpurun_out/r06_r/tile_rows_ab2.txt
done
