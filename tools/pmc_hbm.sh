#!/bin/bash
# usage: tools/pmc_hbm.sh <tag> <workload>  -- HBM traffic per launch: separate --pmc passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md: one counter per pass; FETCH_SIZE in KB is doubled for 16 B/lane streaming reads on gfx950)
tag=$1; wl=$2
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/${tag}_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/${tag}_$ctr -o r -- python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_$ctr/bench.log 2>&1
done
python3 - <<PY
import csv, collections, json, glob, hashlib
src_hash = hashlib.sha256(b"".join(open("webrender_amd/csrc/" + f, "rb").read() for f in ("wrhip_kernels.h", "wrhip_k_setup.h", "wrhip_k_pixels.h", "wrhip_k_rows.h", "wrhip_k_raster.h", "wrhip_types.h"))).hexdigest()[:16]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/${tag}_%s/*counter_collection.csv" % ctr)
    for row in csv.DictReader(open(f[0])):
        name = row["Kernel_Name"].split("(")[0]
        agg[f"{name} grid={row['Grid_Size']}"][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"workload": "$wl quad (bench.py --workload $wl --steps 20 --warmup 5), separate --pmc passes for FETCH_SIZE and WRITE_SIZE; "
                   "hbm_bytes_per_launch = 1024 * (2 * FETCH_SIZE_KB + WRITE_SIZE_KB)", "kernel_sources_sha256_16": src_hash, "kernels": {}}
for k, v in agg.items():
    fk = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0])))
    wk = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
    out["kernels"][k] = {"FETCH_SIZE_KB_mean": round(fk, 1), "WRITE_SIZE_KB_mean": round(wk, 1),
                         "hbm_bytes_per_launch": int(1024 * (2 * fk + wk)), "launches": len(v.get("FETCH_SIZE", []))}
json.dump(out, open("gpurun_out/${tag}.json", "w"), indent=1)
for k, v in out["kernels"].items():
    if "raster" in k or "setup" in k or "upload" in k: print(k[:70], v)
PY
