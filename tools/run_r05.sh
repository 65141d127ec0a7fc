#!/bin/bash
# usage (GPU box, repo root): tools/run_r05.sh tests|bench <tag>  - the round's evidence set in two gpurun calls
what=$1; tag=$2
if [ "$what" = tests ]; then
  rm -f gpurun_out/attention_parity.txt gpurun_out/rebuild_parity.txt gpurun_out/decode_trace_factors.txt
  timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/${tag}_tests.txt 2>&1; rc=$?
  echo "rc=$rc" >> gpurun_out/${tag}_tests.txt; tail -5 gpurun_out/${tag}_tests.txt
  exit $rc
fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
python tools/show_bench.py gpurun_out/${tag}_bench.json 2>/dev/null | cut -c1-400
tools/prof.sh ${tag} --steps 24 --warmup 6 | cut -c1-170 || exit 1
tools/prof.sh ${tag}_glm --workload glm4_200k --steps 16 --warmup 6 | cut -c1-170 || exit 1
tools/pmc_score.sh ${tag} | tail -24
