#!/bin/bash
tag=$1
( cd /tmp; for cfg in "8 4 15560 256" "4 8 25544 256"; do set -- $cfg; echo "== B=$1 G=$2 N=$3 S=$4"; TOPK_PROBE_B=$1 TOPK_PROBE_G=$2 TOPK_PROBE_N=$3 TOPK_PROBE_S=$4 timeout -k 5 60 $GRAFT_REPO_ROOT/tools/bin/topk3_probe_stamps; TOPK_PROBE_B=$1 TOPK_PROBE_G=$2 TOPK_PROBE_N=$3 TOPK_PROBE_S=$4 timeout -k 5 60 $GRAFT_REPO_ROOT/tools/bin/topk3_probe; done ) > gpurun_out/${tag}_topk3_probe.txt 2>&1
cat gpurun_out/${tag}_topk3_probe.txt
timeout -k 10 300 python -m pytest tests/test_gpu_fused_parity.py -m gpu -q -x --timeout 120 -k "fused_selection" > gpurun_out/${tag}_fusedtests.txt 2>&1
rc=$?; echo rc=$rc >> gpurun_out/${tag}_fusedtests.txt; tail -12 gpurun_out/${tag}_fusedtests.txt
if [ $rc -ne 0 ]; then echo "fused tests failed: stop"; exit 1; fi
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 500 > gpurun_out/${tag}_gputests.txt 2>&1
rc=$?; echo rc=$rc >> gpurun_out/${tag}_gputests.txt; tail -6 gpurun_out/${tag}_gputests.txt
if [ $rc -ge 124 ]; then echo "tests killed: no further GPU step"; exit $rc; fi
for e in 28 40 56; do
  timeout -k 10 200 python bench.py --steps 32 --warmup 8 --no-extras --no-cpu-baseline --early-fetch $e > gpurun_out/${tag}_E$e.json 2> gpurun_out/${tag}_E$e.err || { echo "E=$e failed"; tail -3 gpurun_out/${tag}_E$e.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_E$e.json").read().strip().splitlines()[-1])
print("E=$e", d["value"], "tok/s", d["ms_per_step"], "ms")
PY
done
