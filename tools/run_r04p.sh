#!/bin/bash
tag=$1
tools/prof.sh ${tag}_fused --steps 24 --warmup 6 > /dev/null 2>&1
for t in fused; do echo "== $t"; grep -E "skv_|^# total" gpurun_out/${tag}_${t}_kernel_stats.txt | cut -c1-150; done
tools/prof.sh ${tag}_glm --workload glm4_200k --steps 16 --warmup 6 > /dev/null 2>&1
echo "== glm"; grep -E "skv_(score|normalize|topk2|rebuild|attn_merge)" gpurun_out/${tag}_glm_kernel_stats.txt | cut -c1-150
