#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# rocprofv3 kernel trace of a short bench run; condensed per-kernel table -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/${tag}_prof_bench.json 2> gpurun_out/${tag}_prof.log
f=$(find $out -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --no-cpu-baseline $*"; python3 tools/summarize_rocprof.py "$f" 45; echo "# bench line under the profiler:"; cat gpurun_out/${tag}_prof_bench.json; } > gpurun_out/${tag}_kernel_stats.txt
# steady state only (the timed hipGraph replays): per-kernel table of the last K steps + the in-step duration of the scan
steps=64; wl=llama31_122k; prev=""
for a in "$@"; do [ "$prev" = "--steps" ] && steps=$a; [ "$prev" = "--workload" ] && wl=$a; prev=$a; done
t=$(find $out -name '*kernel_trace.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --no-cpu-baseline $*"; python3 tools/summarize_rocprof.py --graph-replays-only "$t" --steps $steps --json gpurun_out/${tag}_score_in_step.json 30; } > gpurun_out/${tag}_step_kernel_stats.txt
python3 - <<PY
import json
p = "gpurun_out/${tag}_score_in_step.json"
z = json.load(open(p)); z["workload"] = "$wl"; z["command"] = "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --no-cpu-baseline $*"
json.dump(z, open(p, "w"), indent=1)
PY
grep -E "skv_|^# " gpurun_out/${tag}_step_kernel_stats.txt
grep -E "skv_|^# total" gpurun_out/${tag}_kernel_stats.txt
