#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_score.sh <tag>
# HBM traffic of the landmark scan from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (with --kernel-trace only), gfx950 read correction x2 for 16 B/lane streams.
tag=$1
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -o pmc -- python3 tools/time_score.py > gpurun_out/pmc_${tag}_$c.log 2>&1 || exit 1
done
python3 - $tag <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "skv_score_tile_kernel" in k and r["Counter_Name"] == c:
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    out[c] = {k: (sum(v) / len(v), len(v)) for k, v in per.items()}
res = {}
for k in out["FETCH_SIZE"]:
    fkb, n = out["FETCH_SIZE"][k]; wkb, _ = out["WRITE_SIZE"].get(k, (0.0, 0))
    fetch = fkb * 1024 * 2; write = wkb * 1024          # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 x2 read correction
    res[k] = dict(launches=n, FETCH_SIZE_KB_raw=round(fkb, 1), WRITE_SIZE_KB_raw=round(wkb, 1), fetch_bytes_corrected=int(fetch),
                  write_bytes=int(write), hbm_bytes_per_launch=int(fetch + write))
json.dump(res, open(f"gpurun_out/{tag}_score_kernel_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
