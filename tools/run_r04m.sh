#!/bin/bash
tag=$1
( cd /tmp; for cfg in "4 8 25544 256"; do set -- $cfg; echo "== B=$1 G=$2 N=$3 S=$4"; TOPK_PROBE_B=$1 TOPK_PROBE_G=$2 TOPK_PROBE_N=$3 TOPK_PROBE_S=$4 timeout -k 5 60 $GRAFT_REPO_ROOT/tools/bin/topk3_probe_stamps; done ) 2>&1 | cut -c1-200
timeout -k 10 300 python -m pytest tests/test_gpu_fused_parity.py -m gpu -q -x --timeout 120 -k "fused_selection" > gpurun_out/${tag}_fusedtests.txt 2>&1
rc=$?; echo rc=$rc >> gpurun_out/${tag}_fusedtests.txt; tail -5 gpurun_out/${tag}_fusedtests.txt
if [ $rc -ne 0 ]; then echo "fused tests failed: stop"; exit 1; fi
run() { # workload E delay
  wl=$1; e=$2; d=$3; shift 3
  SKV_PULL_DELAY_US=$d timeout -k 10 300 python bench.py --workload $wl --steps 24 --warmup 6 --no-extras --no-cpu-baseline --early-fetch $e "$@" > gpurun_out/${tag}_tmp.json 2> gpurun_out/${tag}_tmp.err || { echo "$wl E=$e failed"; tail -3 gpurun_out/${tag}_tmp.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_tmp.json").read().strip().splitlines()[-1])
print("$wl E=$e delay=$d $*", d["value"], "tok/s", d["ms_per_step"], "ms")
PY
}
run llama31_122k 28 0 --fused-select 0 || exit 1
for e in 32 40; do run llama31_122k $e 0 || exit 1; done
run llama31_122k 40 5 || exit 1
run glm4_200k 64 0 --fused-select 0 || exit 1
for e in 64 80; do run glm4_200k $e 0 || exit 1; done
run glm4_200k 64 7 || exit 1
run llama31_244k_b4096 56 0 || exit 1
run yi9b_122k 64 0 || exit 1
run yi9b_122k 64 0 --fused-select 0 || exit 1
