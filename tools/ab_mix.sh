#!/bin/bash
# usage: tools/ab_mix.sh <tag> "<workloads>" "<variant> ..."  -- on ONE GPU box, interleaved, three rounds: every workload's bench line under
# each variant; a variant is  name[:lib=<path under the repo>][:ENV=v[,ENV=v...]]  (e.g. "base:lib=webrender_amd/csrc/ab/libwrhip_base.so"
# "new" "noqtab:WRHIP_NO_QTAB=1").  Library builds and environment switches in one comparison (tools/ab.sh, tools/ab_env.sh do one kind each).
tag=$1; wls=$2; shift 2
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$tag
for w in $wls; do
  for round in 1 2 3; do
    for v in "$@"; do
      name=${v%%:*}; rest=${v#*:}; [ "$rest" = "$v" ] && rest=""
      pre="env"
      IFS=':' read -ra parts <<< "$rest"
      for p in "${parts[@]}"; do
        case "$p" in
          lib=*) pre="$pre WRHIP_LIB_PATH=$PWD/${p#lib=}";;
          "") ;;
          *) for kv in ${p//,/ }; do pre="$pre $kv"; done;;
        esac
      done
      $pre python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print('$w', '$name', 'fps', d['value'], 'kernel_us', r.get('kernel_us_per_frame'), ' '.join('%s:%gx%.1f' % (k['name'].replace('wr_','').replace('_kernel','').replace(', false','F').replace(', true','T'), k['launches_per_frame'], k['us']) for k in r.get('per_kernel', [])))" | tee -a gpurun_out/$tag/ab.txt
    done
  done
done
