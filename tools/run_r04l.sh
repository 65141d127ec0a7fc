#!/bin/bash
tag=$1
run() { # workload E delay
  wl=$1; e=$2; d=$3; shift 3
  SKV_PULL_DELAY_US=$d timeout -k 10 300 python bench.py --workload $wl --steps 24 --warmup 6 --no-extras --no-cpu-baseline --early-fetch $e "$@" > gpurun_out/${tag}_tmp.json 2> gpurun_out/${tag}_tmp.err || { echo "$wl E=$e failed"; tail -3 gpurun_out/${tag}_tmp.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_tmp.json").read().strip().splitlines()[-1])
print("$wl E=$e delay=$d $*", d["value"], "tok/s", d["ms_per_step"], "ms")
PY
}
for d in 0 3 5 7 9; do run llama31_122k 32 $d || exit 1; done
for d in 5 7; do run llama31_122k 40 $d || exit 1; done
for d in 0 4 7 10 13; do run glm4_200k 64 $d || exit 1; done
for d in 0 5 8; do run llama31_244k_b4096 56 $d || exit 1; done
run llama31_244k_b4096 56 0 --fused-select 0 || exit 1
run llama31_244k_b4096 72 0 --fused-select 0 || exit 1
