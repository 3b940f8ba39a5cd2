"""Where the DBoF cluster kernel's LDS bank conflicts come from: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the three forms
scripts/dbof_bench.py fwd launches in this order, 21 launches each - training (tape + statistics + selection), statistics + selection without
the tape, evaluation (selection only).  The main loop is the same in all three; only the first has the tape's LDS transpose.

    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d <dir> -o run -- python3 scripts/dbof_bench.py fwd
    python scripts/dbof_lds_conflicts.py <dir>
"""
import csv
import glob
import os
import sys
from collections import defaultdict

fs = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
if not fs:
    sys.exit("no counter_collection.csv under %s" % sys.argv[1])
per = defaultdict(dict)
order = {}
for r in csv.DictReader(open(fs[0])):
    if "dbof_cluster_pool" not in r["Kernel_Name"]:
        continue
    d = int(r["Dispatch_Id"])
    per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order[d] = (int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0])
ds = sorted(per, key=lambda d: order[d][0])
n = len(ds) // 3
for name, group in (("training: tape + statistics + selection", ds[:n]), ("statistics + selection, no tape", ds[n:2 * n]), ("evaluation: selection only", ds[2 * n:])):
    c = sum(per[d].get("SQ_LDS_BANK_CONFLICT", 0.0) for d in group) / max(1, len(group))
    a = sum(per[d].get("SQ_LDS_IDX_ACTIVE", 0.0) for d in group) / max(1, len(group))
    print("%-42s %2d launches of %s: SQ_LDS_BANK_CONFLICT %.3e  SQ_LDS_IDX_ACTIVE %.3e  ratio %.4f" % (name, len(group), order[group[0]][1] if group else "-", c, a, c / a if a else 0.0))
