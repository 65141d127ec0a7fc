#!/bin/bash
# usage (GPU box): tools/run_r04g.sh <tag>: rocprof kernel stats of the fused / three-launch step, then an E sweep of the fused step
tag=$1
tools/prof.sh ${tag}_fused --steps 24 --warmup 6 --fused-select 1 > /dev/null 2>&1
tools/prof.sh ${tag}_three --steps 24 --warmup 6 --fused-select 0 > /dev/null 2>&1
for t in fused three; do echo "== $t"; grep -E "skv_(score|normalize|topk2|rebuild|attn_merge)" gpurun_out/${tag}_${t}_kernel_stats.txt | cut -c1-150; done
for e in 20 28 40 56 72 96; do
  timeout -k 10 200 python bench.py --steps 32 --warmup 8 --no-extras --no-cpu-baseline --early-fetch $e > gpurun_out/${tag}_E$e.json 2> gpurun_out/${tag}_E$e.err || { echo "E=$e failed"; tail -3 gpurun_out/${tag}_E$e.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_E$e.json").read().strip().splitlines()[-1])
print("E=$e", d["value"], "tok/s", d["ms_per_step"], "ms")
PY
done
