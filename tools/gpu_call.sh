#!/bin/bash
# usage: tools/gpu_call.sh <name> <timeout-seconds> <command...>   (several triples separated by ---)
# Runs the steps in order inside one gpurun call, each under its own timeout, logging to gpurun_out/<name>.log.  A step that times out or is
# killed ends the call: no further GPU step is started after a kill.
set -o pipefail
mkdir -p gpurun_out
while [ $# -gt 0 ]; do
    name=$1; t=$2; shift 2
    cmd=()
    while [ $# -gt 0 ] && [ "$1" != "---" ]; do cmd+=("$1"); shift; done
    [ "$1" == "---" ] && shift
    echo "== $name: $(date +%T)"
    timeout -k 10 "$t" "${cmd[@]}" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "== $name rc=$rc $(date +%T)"; tail -n 4 "gpurun_out/$name.log" | cut -c1-240
    if [ $rc -ge 124 ]; then echo "timeout/kill in $name: stopping"; exit $rc; fi
done
exit 0
