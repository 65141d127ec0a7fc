#!/bin/bash
tag=$1
tools/prof.sh ${tag}_glm_fused --workload glm4_200k --steps 16 --warmup 6 --fused-select 1 > /dev/null 2>&1
tools/prof.sh ${tag}_glm_three --workload glm4_200k --steps 16 --warmup 6 --fused-select 0 > /dev/null 2>&1
tools/prof.sh ${tag}_244k_fused --workload llama31_244k_b4096 --steps 16 --warmup 6 --fused-select 1 > /dev/null 2>&1
tools/prof.sh ${tag}_244k_three --workload llama31_244k_b4096 --steps 16 --warmup 6 --fused-select 0 > /dev/null 2>&1
for t in glm_fused glm_three 244k_fused 244k_three; do echo "== $t"; grep -E "skv_(score|normalize|topk2|rebuild|attn_merge)" gpurun_out/${tag}_${t}_kernel_stats.txt | cut -c1-150; done
