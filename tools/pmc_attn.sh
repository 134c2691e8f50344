#!/bin/bash
# Which instructions of attention_idx_kernel cause its LDS bank conflicts: rocprofv3 counter passes over
# tools/attn_variants.py for timing variants of the DIAGNOSTIC library (MMEE_ATTN_DBG bits: 1 no index loads / lookups, 8 no P V and its
# transposed V reads, 128 no Q K^T MFMAs, 2 no softmax VALU), per-kernel sums of the attention kernel only.
# Usage (GPU box): bash tools/pmc_attn.sh <outfile> ["DBG=0" "DBG=1" ...]
out=$GRAFT_REPO_ROOT/$1; shift
if [ $# -eq 0 ]; then set -- "DBG=0" "DBG=1" "DBG=8" "DBG=9"; fi
cd /tmp && export TMPDIR=/tmp
export B=${B:-64}
for v in "$@"; do
    for kv in $v; do export MMEE_ATTN_${kv}; done
    echo "== $v  (B=$B)" >> "$out"
    i=0
    for ctrs in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
                "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
                "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        rm -rf /tmp/pmc_a_$i
        timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_a_$i -o t -- python3 "$GRAFT_REPO_ROOT/tools/attn_variants.py" > /tmp/pmc_a_$i.log 2>&1 || { echo "pass $i ($ctrs) failed" >> "$out"; tail -n 3 /tmp/pmc_a_$i.log >> "$out"; }
        f=$(find /tmp/pmc_a_$i -name '*counter_collection.csv' | head -1)
        [ -n "$f" ] && python3 - "$f" >> "$out" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "attention_idx_kernel" in n:
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
for c, (v, k) in sorted(acc.items()):
    print(f"  {c:28s} per launch {v / k:16.1f}  launches {k}")
PY
    done
    for kv in $v; do unset MMEE_ATTN_${kv%%=*}; done
done
