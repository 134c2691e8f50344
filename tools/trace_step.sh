#!/bin/bash
# per-launch kernel durations of ONE pinned-threshold bench step, in launch order (GPU box):
#   bash tools/trace_step.sh tag   -> gpurun_out/<tag>_trace.csv  (kernel, start_ns, dur_us)
set -o pipefail
TAG=${1:-trace}
THR=${THR:-0.524019,0.542159,0.507206,0.430526,0.883812}      # thresholds of the default bench (round 5: 2 x 1024 documents, release 0.2, calibrated on seed 501234)
OUT=$PWD/gpurun_out
ROOT=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_kt
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_kt -o t -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-docs 0 --stream-docs 0 --no-traffic --no-profile --serial-slices --distinct-batches 1 --no-small-batch --thresholds $THR --probe-layers ${PLAN:-1,3,5,7,9} $EXTRA > $OUT/${TAG}_kt.log 2>&1
python3 - "$(find $OUT/${TAG}_kt -name '*kernel_trace.csv' | head -1)" $OUT/${TAG}_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last forward = everything after the last doc_prep launch
last = max(i for i, r in enumerate(rows) if "doc_prep" in r["Kernel_Name"] or "prep_uniform" in r["Kernel_Name"])
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "start_us", "dur_us", "grid", "wg"])
    t0 = int(rows[last]["Start_Timestamp"])
    for r in rows[last:]:
        w.writerow([r["Kernel_Name"][:110], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                    r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", "")])
PY
rm -rf $OUT/${TAG}_kt
