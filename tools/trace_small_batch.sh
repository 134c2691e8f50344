#!/bin/bash
# per-kernel durations of B = 1 forwards (GPU box): rocprofv3 --kernel-trace over tools/small_batch_probe.py with N forwards; prints, per kernel name, calls, mean
# duration, share of the summed kernel time, and the summed kernel time against the wall time of the forwards (the rest is dispatch gaps).
OUT=$PWD/gpurun_out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
export N=${N:-40} B=${B:-1}
rm -rf $OUT/sb_kt
rocprofv3 --kernel-trace --output-format csv -d $OUT/sb_kt -o t -- python3 $ROOT/tools/small_batch_probe.py > $OUT/sb_kt.log 2>&1
python3 - "$(find $OUT/sb_kt -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 forwards of the synchronised whole-layers leg = the tail of the trace: take the kernels behind the last 40 doc_prep launches
idx = [i for i, r in enumerate(rows) if "doc_prep" in r["Kernel_Name"]]
first = idx[-40]
sel = rows[first:]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r["Kernel_Name"].replace("void mmee::", "")[:90]
    acc[n][0] += 1; acc[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
print(f"last 40 forwards (whole layers, synchronised): {len(sel)} kernels, {len(sel) / 40:.1f} per forward; summed kernel time {tot / 40:.1f} us per forward; span {span / 40:.1f} us per forward (incl. host gaps between forwards)")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{c / 40:6.1f} x {t / c:7.1f} us = {t / 40:7.1f} us/fwd ({100 * t / tot:4.1f} %)  {n}")
PY
rm -rf $OUT/sb_kt
