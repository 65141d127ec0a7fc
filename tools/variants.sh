#!/bin/bash
# usage (on the GPU box, from the repo root): tools/variants.sh <tag>
# the non-headline lines of DESIGN.md section 5 (one short bench run each) -> gpurun_out/<tag>_variants.txt
tag=$1
out=gpurun_out/${tag}_variants.txt
: > $out
run() { echo "# python bench.py --no-extras --no-cpu-baseline $*" >> $out; python3 bench.py --no-extras --no-cpu-baseline "$@" >> $out 2>> gpurun_out/${tag}_variants.err || exit 1; }
for b in 2 4 8 16 24; do run --batch $b --steps 24 --warmup 4; done
run --mode eager --steps 32 --warmup 4
run --v-table hbm --steps 32 --warmup 4
run --attn full --steps 32 --warmup 4
run --overlap-attention 0 --steps 32 --warmup 4
run --workload llama31_4k --steps 32 --warmup 4
run --workload llama3_1048k_full --steps 16 --warmup 4
python3 - $out <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("#"): cmd = line[2:].strip(); continue
    d = json.loads(line); print(f"{d['value']:9.2f} tok/s {d['ms_per_step']:8.3f} ms/step  hit {d.get('chunk_hit_rate')}  | {cmd}")
PY
