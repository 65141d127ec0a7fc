#!/bin/bash
# usage (GPU box, repo root): tools/run_r04.sh <tag>  - GPU tests, then short bench lines of the round-4 workloads (skipped if the tests were killed)
tag=$1
rm -f gpurun_out/attention_parity.txt gpurun_out/rebuild_parity.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 500 > gpurun_out/${tag}_gputests.txt 2>&1
rc=$?
echo rc=$rc >> gpurun_out/${tag}_gputests.txt
tail -8 gpurun_out/${tag}_gputests.txt
if [ $rc -ge 124 ]; then echo "tests killed: no further GPU step"; exit $rc; fi
for wl in llama31_244k_b4096 llama31_60k_b1024 yi9b_122k; do
  timeout -k 10 400 python bench.py --workload $wl --steps 24 --warmup 6 --no-extras --no-cpu-baseline > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err || { echo "bench $wl rc=$?"; tail -5 gpurun_out/${tag}_bench_$wl.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", d["value"], "tok/s", d["ms_per_step"], "ms", "hit", d["chunk_hit_rate"], "scan", d["roofline"]["us_per_launch"], "us", d["roofline"]["frac"], "E", d["early_fetch_chunks_per_head"])
PY
done
