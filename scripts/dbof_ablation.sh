#!/bin/bash
# cfg 4's cluster kernel (dbof_cluster_pool_walk_kernel), round 6: (a) timing ablations of its epilogue - the bf16 tape store (runtime: act = NULL), the
# column statistics (part = NULL / -DEVC_ABLATE_DBOF_STATS), the per-video arg-max selection (-DEVC_ABLATE_DBOF_SELECT), the whole epilogue
# (-DEVC_ABLATE_DBOF_EPILOGUE: the walk's main loops alone); (b) XCD pinning by ROW tile instead of by column panel (EVC_DBOF_PIN=rows), kernel alone,
# inside the cfg-4 training step, and its HBM-side FETCH_SIZE.  The ablation libraries are build_ab/libevc_dbof_no_*.so (csrc/build.sh -DEVC_ABLATE_DBOF_*).
#   bash scripts/dbof_ablation.sh [out-file]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/dbof_ablation.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for round in 1 2; do
  echo "== shipped library, column panels pinned to XCDs (round $round)" >> "$OUT"
  timeout 200 python3 scripts/dbof_bench.py fwd plain 2>/dev/null >> "$OUT"
  echo "== shipped library, ROW tiles pinned to XCDs: EVC_DBOF_PIN=rows (round $round)" >> "$OUT"
  EVC_DBOF_PIN=rows timeout 200 python3 scripts/dbof_bench.py fwd 2>/dev/null >> "$OUT"
  for v in STATS SELECT EPILOGUE; do
    echo "== -DEVC_ABLATE_DBOF_$v (round $round)" >> "$OUT"
    EVC_LIB=$PWD/build_ab/libevc_dbof_no_$v.so timeout 200 python3 scripts/dbof_bench.py fwd 2>/dev/null | head -1 >> "$OUT"
  done
done
echo "== cfg 4 training step (bench.py --config dbof --steps 30), alternating" >> "$OUT"
for round in 1 2 3; do
  for pin in cols rows; do
    EVC_DBOF_PIN=$pin timeout 300 python3 bench.py --config dbof --no_cpu_baseline --steps 30 --warmup 6 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pin %-5s %.3f ms/step (median %.3f)  cluster kernel %.1f us = %.3f of peak' % ('$pin', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'] * 1e3, d['roofline']['frac']))" >> "$OUT"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pin in cols rows; do
  rm -rf gpurun_out/pmc_dbof_pin_$pin
  EVC_DBOF_PIN=$pin timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_dbof_pin_$pin -o run -- python3 scripts/dbof_bench.py fwd > /dev/null 2>&1
  python3 - gpurun_out/pmc_dbof_pin_$pin $pin >> "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dbof_cluster_pool" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
if rows:
    # the first 21 launches = the training form (tape + statistics + selection) of scripts/dbof_bench.py fwd.  FETCH_SIZE in KB, doubled per the gfx950
    # note of MI355X_MICROARCH.md (scripts/pmc_kernels.py uses the same correction); includes Infinity-Cache hits
    v = sorted(x for _, x in sorted(rows)[:21])
    print("pin %-5s FETCH_SIZE per launch (median of %d training-form launches): raw %.0f KB -> %.0f MB HBM-side (operands: 38 MB frames + 19 MB W_c)" % (
        sys.argv[2], len(v), v[len(v) // 2], v[len(v) // 2] * 2.0 * 1024 / 1e6))
else:
    print("pin %s: no counter rows found" % sys.argv[2])
PY
  rm -rf gpurun_out/pmc_dbof_pin_$pin
done
cat "$OUT"
