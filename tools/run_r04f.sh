#!/bin/bash
# usage (GPU box): tools/run_r04f.sh <tag>: fused-selection tests first (short timeout), then the whole GPU suite, then bench A/B
tag=$1
timeout -k 10 300 python -m pytest tests/test_gpu_fused_parity.py -m gpu -q -x --timeout 120 -k "fused_selection" > gpurun_out/${tag}_fusedtests.txt 2>&1
rc=$?; echo rc=$rc >> gpurun_out/${tag}_fusedtests.txt; tail -15 gpurun_out/${tag}_fusedtests.txt
if [ $rc -ne 0 ]; then echo "fused tests failed: stop"; exit 1; fi
rm -f gpurun_out/attention_parity.txt gpurun_out/rebuild_parity.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 500 > gpurun_out/${tag}_gputests.txt 2>&1
rc=$?; echo rc=$rc >> gpurun_out/${tag}_gputests.txt; tail -8 gpurun_out/${tag}_gputests.txt
if [ $rc -ge 124 ]; then echo "tests killed: no further GPU step"; exit $rc; fi
for f in 1 0; do
  timeout -k 10 300 python bench.py --steps 32 --warmup 8 --no-extras --no-cpu-baseline --fused-select $f > gpurun_out/${tag}_bench_f$f.json 2> gpurun_out/${tag}_bench_f$f.err || { echo "bench fused=$f failed"; tail -5 gpurun_out/${tag}_bench_f$f.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench_f$f.json").read().strip().splitlines()[-1])
print("fused=$f", d["value"], "tok/s", d["ms_per_step"], "ms hit", d["chunk_hit_rate"], "scan", d["roofline"]["us_per_launch"], "us")
PY
done
