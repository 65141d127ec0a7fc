#!/bin/bash
# usage (GPU box, repo root): tools/run_steps.sh <tag> "<seconds> <command...>" ...
# Runs the steps one after the other, each under its own `timeout -k 10`; a step that merely FAILS (non-zero exit) does
# not stop the sequence, a step that was KILLED (timeout 124/137, or a signal >= 128) does: no further GPU step after it.
tag=$1; shift
mkdir -p gpurun_out
log=gpurun_out/${tag}_steps.log
: > $log
for step in "$@"; do
    secs=${step%% *}; cmd=${step#* }
    echo "=== [$(date +%T)] timeout $secs: $cmd" | tee -a $log
    timeout -k 10 $secs bash -o pipefail -c "$cmd"
    rc=$?
    echo "=== rc $rc" | tee -a $log
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "=== step killed: stopping" | tee -a $log; exit $rc; fi
done
exit 0
