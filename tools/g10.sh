cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_r
for w in clip-clear large-clip-rect many-box-shadows cfg4 aligned-gradient many-images; do
  WLS="$w" bash tools/ab.sh $w ab/libwrhip_base.so webrender_amd/csrc/libwrhip.so 2>&1 | sed "s/^/$w /" | cut -c1-330 | tee -a gpurun_out/r06_r/tile_rows_ab.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -x 2>&1 | tail -3
