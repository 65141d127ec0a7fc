#!/bin/bash
# usage (GPU box): tools/sweep_early.sh <tag> <workload> <E values...>  - headline value per early-fetch chunks per head, two rounds
tag=$1; wl=$2; shift 2
for round in 1 2; do
  for e in "$@"; do
    python bench.py --workload $wl --steps 32 --warmup 6 --no-extras --no-cpu-baseline --no-secondary --early-fetch $e > gpurun_out/${tag}_e$e.json 2> gpurun_out/${tag}_e$e.err
    echo "round $round E=$e: $(python3 -c "import json;d=json.loads(open('gpurun_out/${tag}_e$e.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
  done
done
