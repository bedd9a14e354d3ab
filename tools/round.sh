#!/bin/bash
# usage: tools/round.sh <tag> [cpu]  -- the round's measurement pass on the GPU box (gpurun): the GPU parity suite; the bench line of
# every BASELINE config and of every wrench benchmark workload (with "cpu": the swgl CPU baseline beside EVERY one of them, one core
# and one process per core); the N > 1 path's world-1 self-test; rocprofv3 kernel stats of cfg2-5; HBM traffic (separate FETCH / WRITE
# passes) and SQ counters of cfg2 / cfg3 / cfg4 / cfg5.  Everything lands in gpurun_out/<tag>/; copy what is to be judged to profiles/.
tag=$1; cpu=$2
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
(time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15) > gpurun_out/$tag/gpu_tests.log 2>&1
cat gpurun_out/$tag/gpu_tests.log
# (the HBM counter passes first, and their summaries into profiles/ ON THE BOX: bench.py takes `traffic` from the newest summary there that was
# measured with these very kernel sources -- after a kernel edit the committed ones no longer match, and the lines below would carry null)
for w in cfg2 cfg3 cfg4 cfg5; do bash tools/pmc_hbm.sh ${tag}_pmc_hbm_$w $w > gpurun_out/$tag/pmc_hbm_$w.log 2>&1; cp gpurun_out/${tag}_pmc_hbm_$w.json profiles/ 2>/dev/null; done
nocpu="--no-cpu-baseline"; [ "$cpu" = cpu ] && nocpu=""
python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_cfg2.json 2> gpurun_out/$tag/bench_cfg2.err
for w in cfg1 cfg3 cfg4 cfg5; do
  python bench.py --workload $w --steps 50 --warmup 5 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_$w.json
done
python bench.py --sharded --steps 20 --warmup 3 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_sharded_world1_cfg5.json
for W in 2 4 8; do python bench.py --emulate-world $W --steps 20 --warmup 3 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_emulate_world${W}_cfg5.json; done
for wl in aligned-gradient unaligned-gradient simple-batching large-boxshadow-ellipse large-boxshadow-ellipse-2 large-clip-rect transforms text-rendering many-images large-blur-radius many-box-shadows clip-clear overlapping-text-shadows; do
  python bench.py --workload $wl --steps 40 --warmup 5 $nocpu 2>/dev/null | grep '"metric"' > gpurun_out/$tag/bench_${wl}.json
done
for w in cfg2 cfg3 cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  cp gpurun_out/$tag/prof_$w/r_kernel_stats.csv gpurun_out/$tag/${w}_kernel_stats.csv
  rm -rf gpurun_out/$tag/prof_$w
done
WORKLOADS="cfg2 cfg3 cfg4 cfg5" bash tools/sq.sh ${tag} > gpurun_out/$tag/sq.log 2>&1
grep -h '"metric"' gpurun_out/$tag/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print(d['config']['workload'][:26], 'fps', d['value'], 'lat_ms', d.get('frame_latency_ms'), 'dom', r.get('kernel'), r.get('avg_launch_us'), 'frac', r.get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), ((d.get('cpu_baseline') or {}).get('multi_process') or {}).get('value'))"
