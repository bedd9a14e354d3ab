cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_t
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "blur or chain or cfg4 or wrench or shadow" 2>&1 | tail -2
for w in cfg4 large-blur-radius large-boxshadow-ellipse-2 many-box-shadows large-boxshadow-ellipse; do
  bash tools/ab.sh $w ab/libwrhip_base.so webrender_amd/csrc/libwrhip.so 2>&1 | sed "s/^/$w /" | cut -c1-330 | tee -a gpurun_out/r06_t/span_px4_ab.txt
done
