#!/bin/bash
# gpurun with retries ONLY for "no box / slot free" (exit code 3: nothing ran, nothing was charged).  Any other outcome is returned as is.
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    echo "[gpurun_retry] no slot (attempt $i), sleeping 150 s"
    sleep 150
done
exit 3
