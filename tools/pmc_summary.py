"""Per-kernel sums of rocprofv3 counter-collection / kernel-trace CSVs (several --pmc passes) -> one summary CSV.
usage: python tools/pmc_summary.py OUT.csv DIR [DIR ...]   (every *counter_collection.csv under the DIRs is read)"""
import csv, os, re, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(float)
launches = defaultdict(set)
for d in dirs:
    for dp, _, fs in os.walk(d):
        for f in fs:
            if not f.endswith("counter_collection.csv"):
                continue
            for row in csv.DictReader(open(os.path.join(dp, f))):
                k = re.sub(r"^void\s+|mmee::|\(.*$", "", row["Kernel_Name"]).strip()
                if k.startswith("at::") or k.startswith("__amd"):
                    continue
                acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
                launches[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches", "counter", "sum_over_launches"])
    for (k, c) in sorted(acc):
        w.writerow([k, len(launches[(k, c)]), c, f"{acc[(k, c)]:.6g}"])
print("wrote", out, len(acc), "rows")
