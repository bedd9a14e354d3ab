#!/bin/bash
# usage: tools/trace.sh <tag> [bench args] -- per-launch kernel durations of one steady-state frame (on the GPU box)
tag=$1
bash tools/prof.sh "$@" > /dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/$tag/r_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)
# find a frame boundary: copyBuffer following a raster
start=n//2
while start<n and 'copyBuffer' not in rows[start]['Kernel_Name']: start+=1
base=int(rows[start]['Start_Timestamp'])
k=start
seen=0
while k<n:
    r=rows[k]; s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if 'copyBuffer' in r['Kernel_Name'] and k>start and 'raster' in rows[k-1]['Kernel_Name'] and (s-base)>50000: seen+=1
    if seen>=1: break
    print(f"{(s-base)/1e3:9.1f} {(e-s)/1e3:8.1f} {r['Kernel_Name'][:44]} grid={r.get('Grid_Size_X','')}")
    k+=1
PY
