#!/bin/bash
# A/B and timing variants of the split-precision attention kernel, one process per variant (the switches are read once per process by the
# DIAGNOSTIC library).  Usage on the GPU box:  bash tools/attn_ab.sh "DBG=0" "DBG=1" "LIB=prev" ...   (default list below).
# DBG bits: 1 no bias (index loads + lookups), 2 no softmax VALU, 4 no K / V DMA, 8 no P V, 16 L2-resident DMA source, 64 no barrier,
# 128 no Q K^T MFMAs.  LIB=<tag> runs tools/bin/libmmee_hip_diag_<tag>.so instead (copy a build there before changing the kernel).
cd "$(dirname "$0")/.."
if [ $# -eq 0 ]; then
    set -- "DBG=0" "LIB=prev" "DBG=0" "LIB=prev" "DBG=1" "DBG=4" "DBG=5" "DBG=2" "DBG=8" "DBG=10" "DBG=64" "DBG=128" "DBG=236"
fi
for v in "$@"; do
    envs=""
    for kv in $v; do
        case $kv in
            WORDS=*|B=*) envs="$envs $kv";;
            LIB=*) envs="$envs MMEE_LIB=$PWD/tools/bin/libmmee_hip_diag_${kv#LIB=}.so";;      # an earlier build kept under tools/bin (same-box A/B)
            *) envs="$envs MMEE_ATTN_${kv}";;
        esac
    done
    echo "== $v"
    env $envs timeout -k 10 300 python tools/attn_variants.py 2>&1 | grep -v amdgpu.ids || exit 1
done
