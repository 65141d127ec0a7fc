#!/bin/bash
# usage (GPU box): tools/ab_prof.sh <tag> <kernel regex>  - rocprof kernel stats of a short bench run with the previous library
# (shadowkv_amd/libshadowkv_hip_prev.so) and the current one, on the same box
tag=$1; pat=$2
export TMPDIR=/tmp
for v in prev new prev new; do
  out=gpurun_out/prof_${tag}_$v; rm -rf $out; mkdir -p $out
  if [ $v = prev ]; then export SKV_LIB_PATH=$PWD/shadowkv_amd/libshadowkv_hip_prev.so; else unset SKV_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-extras --no-cpu-baseline --steps 24 --warmup 6 > gpurun_out/${tag}_${v}_bench.json 2> gpurun_out/${tag}_${v}.log
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  echo "== $v: $(python3 -c "import json;print(json.loads(open('gpurun_out/${tag}_${v}_bench.json').read().strip().splitlines()[-1])['value'])")"
  python3 tools/summarize_rocprof.py "$f" 45 | grep -E "$pat" | cut -c1-150
done
