#!/bin/bash
# A/B and timing variants of the split-precision attention kernel, one process per variant (the switches are read once per process by the
# DIAGNOSTIC library).  Usage on the GPU box:  bash tools/attn_ab.sh "PF=0" "PF=1" "PF=1 DBG=1" ...   (default list below)
cd "$(dirname "$0")/.."
if [ $# -eq 0 ]; then
    set -- "PF=0" "PF=1" "PF=0" "PF=1" "PF=1 DBG=1" "PF=1 DBG=32" "PF=1 DBG=4" "PF=1 DBG=16" "PF=1 DBG=36" "PF=1 DBG=5" "PF=1 DBG=2" "PF=1 DBG=8" "PF=1 DBG=10" "RING=2"
fi
for v in "$@"; do
    envs=""
    for kv in $v; do envs="$envs MMEE_ATTN_${kv}"; done
    echo "== $v"
    env $envs timeout -k 10 300 python tools/attn_variants.py 2>&1 | grep -v amdgpu.ids || exit 1
done
