#!/bin/bash
# usage: tools/glyph.sh  -- on the GPU box: text parity cases, then cfg3 kernel stats
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -k "cfg3 or text or masked_rects or image_grid_masked or glyph" 2>&1 | tail -6)
bash tools/stats.sh g3 --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline | cut -c1-220
