#!/bin/bash
# usage: tools/r4_shard.sh <tag> -- world-1 self-test of the N>1 path (native loop) + sharded vs unsharded cfg5 on one GPU
tag=${1:-r04_d}
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "sharded_player" 2>&1 | tail -3
python bench.py --sharded --gpus 1 --steps 40 --warmup 5 2>&1 | grep '"metric"' > gpurun_out/${tag}_bench_sharded_world1_cfg5.json
python bench.py --workload cfg5 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' > gpurun_out/${tag}_bench_cfg5.json
python -c "
import json
a=json.load(open('gpurun_out/${tag}_bench_sharded_world1_cfg5.json')); b=json.load(open('gpurun_out/${tag}_bench_cfg5.json'))
print('sharded world-1', a['value'], 'same-run unsharded', a['single_gpu_same_workload']['value'], 'ratio', a['speedup_vs_single_gpu'], '| separate unsharded run', b['value'], '| all-gather mode', a['multi_gpu']['all_gather'])"
