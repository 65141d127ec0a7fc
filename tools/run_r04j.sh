#!/bin/bash
tag=$1
run() { # workload E extra...
  wl=$1; e=$2; shift 2
  timeout -k 10 300 python bench.py --workload $wl --steps 24 --warmup 6 --no-extras --no-cpu-baseline --early-fetch $e "$@" > gpurun_out/${tag}_tmp.json 2> gpurun_out/${tag}_tmp.err || { echo "$wl E=$e failed"; tail -3 gpurun_out/${tag}_tmp.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_tmp.json").read().strip().splitlines()[-1])
print("$wl E=$e $*", d["value"], "tok/s", d["ms_per_step"], "ms hit", d["chunk_hit_rate"], "scan", d["roofline"]["us_per_launch"])
PY
}
for e in 48 64 80 96; do run glm4_200k $e || exit 1; done
run glm4_200k 64 --fused-select 0 || exit 1
for e in 28 40 56 72; do run llama31_244k_b4096 $e || exit 1; done
run llama31_244k_b4096 28 --fused-select 0 || exit 1
for e in 20 28 40; do run llama31_60k_b1024 $e || exit 1; done
for e in 48 64 80; do run yi9b_122k $e || exit 1; done
for f in 1 0; do run llama31_122k 0 --batch 8 --fused-select $f || exit 1; done
