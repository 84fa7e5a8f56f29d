#!/bin/bash
# Local helper: submit ONE gpurun call; while the pod has no free GPU slot (exit 3: nothing ran, nothing charged) wait and
# ask again.  Any other outcome -- success, a failing command, a refusal -- is final: a command that ran is never repeated.
#   scripts/gpurun_wait.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 90
done
exit 3
