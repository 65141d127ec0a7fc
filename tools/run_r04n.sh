#!/bin/bash
tag=$1
rm -f gpurun_out/attention_parity.txt gpurun_out/rebuild_parity.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 500 > gpurun_out/${tag}_gputests.txt 2>&1
rc=$?; echo rc=$rc >> gpurun_out/${tag}_gputests.txt; tail -6 gpurun_out/${tag}_gputests.txt
if [ $rc -ge 124 ]; then echo "tests killed: no further GPU step"; exit $rc; fi
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo bench rc=$?
python tools/show_bench.py gpurun_out/${tag}_bench.json 2>/dev/null | head -40
