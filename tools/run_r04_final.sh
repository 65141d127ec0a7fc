#!/bin/bash
# usage (GPU box, repo root): tools/run_r04_final.sh <tag>  - the round's evidence set: GPU suite, default bench, kernel stats, PMC
tag=$1
rm -f gpurun_out/attention_parity.txt gpurun_out/rebuild_parity.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/${tag}_tests.txt 2>&1; rc=$?
echo "rc=$rc" >> gpurun_out/${tag}_tests.txt; tail -3 gpurun_out/${tag}_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
python tools/show_bench.py gpurun_out/${tag}_bench.json 2>/dev/null | cut -c1-400
tools/prof.sh ${tag} --steps 24 --warmup 6 | cut -c1-150 || exit 1
tools/prof.sh ${tag}_glm --workload glm4_200k --steps 16 --warmup 6 | cut -c1-150 || exit 1
tools/pmc_score.sh ${tag} | tail -30
