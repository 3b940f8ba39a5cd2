"""Per-launch HBM bytes and MFMA busy fraction of the fused LSTM forward step from the three --pmc passes of
scripts/pmc_collect.sh.  Writes profiles/<name>.json (bench.py reports it as roofline.traffic / mfma_busy_pmc)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_pmc_traffic.json"
MATCH = ("lstm_fwd_step_kernel<TileCfg2<", "lstm_fwd_step_kernel<TileCfg3<")     # the ring tiles (v2 / v3 stages), one tile per workgroup
WALK = ("lstm_fwd_walk2_kernel<TileCfg3<",)                                        # round 5: two tiles per workgroup (layer 0 step s + layer 1 step s-1)


def _match():
    """The walk kernel if the trace has it (then the dominant forward kernel: the one-tile launches left are a level's first and last step), else the one-tile kernels."""
    f = glob.glob(os.path.join(root, "fetch", "*counter_collection.csv"))[0]
    names = {r["Kernel_Name"] for r in csv.DictReader(open(f))}
    return WALK if any(any(m in k for m in WALK) for k in names) else MATCH


MATCH = _match()
IS_WALK = MATCH is WALK


def load(sub):
    f = glob.glob(os.path.join(root, sub, "*counter_collection.csv"))[0]
    per = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values
    dur = defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(m in k for m in MATCH):
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return per, dur


fetch, _ = load("fetch")
write, _ = load("write")
mfma, dur = load("mfma")
kernels = sorted(fetch, key=lambda k: -len(fetch[k]["FETCH_SIZE"]))
n = {k: len(fetch[k]["FETCH_SIZE"]) for k in kernels}
tot = sum(n.values())
avg = lambda d, c: sum(sum(d[k][c]) for k in kernels if c in d[k]) / max(1, sum(len(d[k][c]) for k in kernels if c in d[k]))
fetch_kb, write_kb = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE")
busy, gui = avg(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(mfma, "GRBM_GUI_ACTIVE")
res = {
    "source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; each with "
              "--kernel-trace only) of `python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline` on MI355X (scripts/pmc_collect.sh)",
    "kernel": ("lstm_fwd_walk2_kernel<TileCfg3<BM,4,64,2,4,2>> (two tiles per workgroup and launch: layer 0's step s + layer 1's step s-1; bytes and cycles are per LAUNCH = "
               "two step tiles) - the ring tile heights the row plans select, launches per variant: " if IS_WALK else
               "lstm_fwd_step_kernel<TileCfg3<BM,4,64,2,4,2>> (TileCfg2 for 288/320 rows) - the ring tile heights the row plans select, launches per variant: ")
              + ", ".join("%s x%d" % (k.split("TileCfg")[1].split(">")[0].replace(" ", ""), n[k]) for k in kernels),
    "launches_sampled": tot,
    "FETCH_SIZE_KB_avg_raw": fetch_kb,
    "WRITE_SIZE_KB_avg_raw": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide (16 B/lane) coalesced reads -> doubled "
                  "(MI355X_MICROARCH.md 'HBM'); WRITE_SIZE exact for 16-B stores.",
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "mfma": {
        "SQ_VALU_MFMA_BUSY_CYCLES_avg": busy,
        "GRBM_GUI_ACTIVE_avg_sum_over_8_xcd": gui,
        "avg_kernel_ns_in_this_pass": sum(sum(v) for v in dur.values()) / max(1, sum(len(v) for v in dur.values())),
        "mfma_busy_fraction": busy / ((gui / 8.0) * 1024.0),
        "note": "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE/8) * 1024 SIMDs); gfx950 has no derived-counter XML "
                "(guide), so this is the raw ratio",
    },
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
