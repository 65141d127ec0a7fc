#!/bin/bash
tag=$1
timeout -k 10 600 python -m pytest tests/test_gpu_kv_cache.py tests/test_gpu_fullsize.py -m gpu -q --timeout 300 -k "early" > gpurun_out/${tag}_earlytests.txt 2>&1
rc=$?; echo rc=$rc >> gpurun_out/${tag}_earlytests.txt; tail -8 gpurun_out/${tag}_earlytests.txt
if [ $rc -ne 0 ]; then echo "early tests failed: stop"; exit 1; fi
run() { wl=$1; shift
  timeout -k 10 300 python bench.py --workload $wl --steps 32 --warmup 8 --no-extras --no-cpu-baseline "$@" > gpurun_out/${tag}_tmp.json 2> gpurun_out/${tag}_tmp.err || { echo "$wl failed"; tail -3 gpurun_out/${tag}_tmp.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_tmp.json").read().strip().splitlines()[-1])
print("$wl $*", d["value"], "tok/s", d["ms_per_step"], "ms E", d["early_fetch_chunks_per_head"])
PY
}
run llama31_122k || exit 1
run llama31_122k --early-fetch 40 || exit 1
run llama31_122k --fused-select 0 || exit 1
run glm4_200k || exit 1
run llama31_244k_b4096 || exit 1
