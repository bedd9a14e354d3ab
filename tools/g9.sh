cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r06_p
run() { # label, env...
  lbl=$1; shift
  env "$@" python bench.py --workload cfg5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lbl', 'fps', d['value'], 'host', d.get('host'), 'single', (d.get('single_gpu_same_workload') or {}).get('value'))" | tee -a gpurun_out/r06_p/dist_ab.txt
}
for round in 1 2; do
run plain A=1
run gloo_init WRHIP_BENCH_INIT_DIST=gloo
run nccl_init WRHIP_BENCH_INIT_DIST=nccl
run nccl_init_warm WRHIP_BENCH_INIT_DIST=nccl WRHIP_BENCH_DIST_WARM=1
run nccl_warm_nopin WRHIP_BENCH_INIT_DIST=nccl WRHIP_BENCH_DIST_WARM=1 WRHIP_BENCH_NO_PIN=1
done
python bench.py --sharded --steps 20 --warmup 3 2>/dev/null | grep '"metric"' > gpurun_out/r06_p/bench_sharded_world1_cfg5.json
python3 -c "
import json; d = json.load(open('gpurun_out/r06_p/bench_sharded_world1_cfg5.json')); print('sharded world1', d['value'], 'single', d['single_gpu_same_workload'])" | tee -a gpurun_out/r06_p/dist_ab.txt
for W in 2 4 8; do
python bench.py --emulate-world $W --steps 20 --warmup 3 2>gpurun_out/r06_p/emu_$W.err | grep '"metric"' > gpurun_out/r06_p/bench_emulate_world${W}_cfg5.json
python3 -c "
import json; d = json.load(open('gpurun_out/r06_p/bench_emulate_world${W}_cfg5.json')); print('emulate', $W, 'projected fps', d['value'], 'speedup', d['projected_speedup_vs_single_gpu'], 'single', d['single_gpu'], 'model', d['model']['slowest_rank_us'], d['model']['gather_us'], 'h2d frac', d['max_rank_h2d_frac']); [print('   ', r) for r in d['per_rank']]"
tail -2 gpurun_out/r06_p/emu_$W.err | grep -v amdgpu.ids
done
