#!/usr/bin/env python3
"""usage: tools/results_table.py <tag> [dir]  -- the markdown table of a measurement pass (tools/round.sh <tag> cpu): one row per bench line
found as <dir>/<tag>_bench_*.json (default dir: profiles/) or gpurun_out/<tag>/bench_*.json -- frames/s, the swgl CPU baseline beside it
(one core / one process per core), the dominant kernel with its mean launch time and roofline fractions, host and GPU time per frame."""
import glob, json, os, sys
tag = sys.argv[1]
d = sys.argv[2] if len(sys.argv) > 2 else "profiles"
files = sorted(glob.glob(os.path.join(d, f"{tag}_bench_*.json"))) or sorted(glob.glob(os.path.join("gpurun_out", tag, "bench_*.json")))
order = ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]
rows = []
for f in files:
    try:
        j = json.loads([l for l in open(f) if l.strip().startswith("{")][-1])
    except Exception:
        continue
    if j.get("projected") or j.get("n_gpus", 1) != 1 or "multi_gpu" in j:
        continue
    w = j["config"]["workload"].split(":")[0]
    r = j.get("roofline") or {}
    cb = j.get("cpu_baseline") or {}
    mp = (cb.get("multi_process") or {})
    k = (r.get("kernel") or "").replace("wr_", "").replace("_kernel", "")
    pm = (r.get("peak_measured") or {})
    rows.append((order.index(w) if w in order else 99, w,
                 f"| {w} | {cb.get('value', '--')} / {mp.get('value', '--')} ({mp.get('cores', '--')} proc.) | **{j['value']:.0f}** | {j.get('host', {}).get('wall', '--')} / {r.get('kernel_us_per_frame', '--')} | "
                 f"`{k}` {r.get('avg_launch_us', '--')} us | {r.get('frac', '--')}" + (f" ({r.get('frac_of_measured_store_only')} of the measured write-only ceiling)" if r.get("frac_of_measured_store_only") else "") +
                 f" | {('%.1f MB / %.1f MB' % (r['traffic'] / 1e6, r['algo_bytes_per_launch'] / 1e6)) if r.get('traffic') else '--'} |"))
print("| workload | swgl frames/s: 1 core / N processes | MI355X frames/s | host wall / kernel us per frame | dominant kernel (HIP events) | `frac` of 8 TB/s | HBM traffic / algorithmic per launch |")
print("|---|---|---|---|---|---|---|")
for _, _, line in sorted(rows):
    print(line)
