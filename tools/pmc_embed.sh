#!/bin/bash
# rocprofv3 counter passes over tools/embed_variants.py for the library in MMEE_LIB; prints per-kernel sums for kernels matching "embed".
# Usage (GPU box): MMEE_LIB=... bash tools/pmc_embed.sh <outdir>
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_e_$i -o t -- python3 "$GRAFT_REPO_ROOT/tools/embed_variants.py" > /dev/null 2>&1 || echo "pass $i ($ctrs) failed"
    f=$(find /tmp/pmc_e_$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - "$f" >> "$GRAFT_REPO_ROOT/$out/pmc.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "embed" in n:
        k = (n.split("(")[0][:60], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (n, c), (v, k) in sorted(acc.items()):
    print(f"{n:60s} {c:24s} per launch {v / k:16.1f}  launches {k}")
PY
done
