#!/bin/bash
# tools/steps.sh "<limit_s> <cmd>" ...: run GPU steps one after another on the box, each under its own `timeout -k 10`.
# A step that fails (tests red) does not stop the sequence; a step that hits its limit or is killed DOES -- nothing else
# touches the card after a hung kernel.  One line per step goes to gpurun_out/steps.log.
mkdir -p gpurun_out
for step in "$@"; do
    limit=${step%% *}
    cmd=${step#* }
    start=$(date +%s)
    timeout -k 10 "$limit" bash -c "$cmd"
    rc=$?
    echo "$(date +%H:%M:%S) rc=$rc $(( $(date +%s) - start ))s :: $cmd" | tee -a gpurun_out/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "step hit its limit or was killed: stopping" | tee -a gpurun_out/steps.log
        exit $rc
    fi
done
exit 0
