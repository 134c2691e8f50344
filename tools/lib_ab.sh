#!/bin/bash
# Same-box A/B of whole-step throughput between builds of the RELEASE library: bash tools/lib_ab.sh "" al5 al6 ...   ("" = the in-tree
# libmmee_hip.so, <tag> = tools/bin/libmmee_hip_<tag>.so; build variants with make OBJDIR=build_<tag> TARGET=../../tools/bin/libmmee_hip_<tag>.so EXTRA=...)
cd "$(dirname "$0")/.."
for tag in "$@"; do
    lib="$PWD/multi-modal-early-exit_amd/libmmee_hip.so"
    [ -n "$tag" ] && lib="$PWD/tools/bin/libmmee_hip_${tag}.so"
    echo "== ${tag:-in-tree}"
    MMEE_LIB=$lib timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-traffic --cpu-docs 0 --stream-docs 0 --no-small-batch --no-extra-rates 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],1),'docs/s  clock',round(d['clock_ghz_timed_region'],4),' per GHz',round(d['docs_per_sec_per_ghz'],1),' gemm',round(d['gemm_class_tflops'],1),' attn',round(d['attention_tflops'],1),' ffn_up',round(d['roofline']['achieved'],1), {k:v for k,v in d['kernel_time_share'].items() if v>0.03})" || exit 1
done
