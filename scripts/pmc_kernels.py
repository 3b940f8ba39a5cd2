"""Per-kernel summary of the four --pmc passes of scripts/pmc_collect.sh: HBM-side bytes per launch (with the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md), MFMA busy fraction and the wave-state split (SQ_WAIT_ANY = parked on
s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three are disjoint shares of
SQ_WAVE_CYCLES).    python scripts/pmc_kernels.py <pmc dir> <out.json> [name=substring ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
KERNELS = dict(a.split("=", 1) for a in sys.argv[3:]) or {
    "lstm_fwd_walk2 (teacher L1, two step tiles per launch: layer 0 step s + layer 1 step s-1)": "lstm_fwd_walk2_kernel<TileCfg3<",
    "lstm_fwd_step (one tile per launch: a level's first / last step, other tile heights)": "lstm_fwd_step_kernel<TileCfg3<",
    "lstm_bwd_step (L1 BPTT, 128x128 ring tile)": "lstm_bwd_step_kernel<TileCfg3<128",
    "gemm_tn 256x256 (weight gradients)": "gemm_tn_kernel<TileCfg2<256",
    "gemm_nt 256x256 / 224x256 (dX, hoisted projections)": "gemm_nt_kernel<TileCfg3<2",
    "moe_update pass 2 (fused clip + Adam of the MoE weights)": "moe_update_kernel<TileCfg2<128, 1, 128, 2, 4, 5, true>, 2>",
    "dbof_cluster_pool (DBoF cluster GEMM + statistics + selection; round 5: the tile walk)": "dbof_cluster_pool_",
    "dbof_dact": "dbof_dact_kernel",
}


def load(sub):
    fs = glob.glob(os.path.join(root, sub, "*counter_collection.csv"))
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    if not fs:
        return per, dur
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        for name, sub_ in KERNELS.items():
            if sub_ in k:
                per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (name, r.get("Dispatch_Id"))
                if key not in seen:
                    seen.add(key)
                    dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return per, dur


fetch, _ = load("fetch")
write, _ = load("write")
mfma, dur = load("mfma")
waves, _ = load("waves")
mean = lambda v: sum(v) / len(v) if v else None
res = {"source": "rocprofv3 --kernel-trace --pmc passes of `%s` on one MI355X (scripts/pmc_collect.sh): FETCH_SIZE | WRITE_SIZE | "
                 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" % open(os.path.join(root, "command.txt")).read().strip(),
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> doubled; WRITE_SIZE exact for 16-B "
                     "stores; FETCH_SIZE includes Infinity-Cache hits (MI355X_MICROARCH.md 'HBM')",
       "kernels": {}}
for name in KERNELS:
    if not fetch[name] and not mfma[name] and not waves[name]:
        continue
    e = {"launches_sampled": len(dur[name]) or len(fetch[name].get("FETCH_SIZE", []))}
    if dur[name]:
        e["avg_kernel_us_in_pmc_pass"] = round(mean(dur[name]) / 1e3, 2)
    f, w = mean(fetch[name].get("FETCH_SIZE", [])), mean(write[name].get("WRITE_SIZE", []))
    if f is not None and w is not None:
        e["FETCH_SIZE_KB_raw"], e["WRITE_SIZE_KB_raw"] = round(f, 1), round(w, 1)
        e["hbm_side_MB_per_launch"] = round((2.0 * f + w) * 1024 / 1e6, 2)
        if dur[name]:
            e["hbm_side_TB_per_s"] = round((2.0 * f + w) * 1024 / (mean(dur[name]) * 1e-9) / 1e12, 2)
    b, g = mean(mfma[name].get("SQ_VALU_MFMA_BUSY_CYCLES", [])), mean(mfma[name].get("GRBM_GUI_ACTIVE", []))
    if b is not None and g:
        e["mfma_busy_fraction"] = round(b / ((g / 8.0) * 1024.0), 4)
    wc = mean(waves[name].get("SQ_WAVE_CYCLES", []))
    if wc:
        for c, key in (("SQ_WAIT_ANY", "wave_cycles_parked_on_waitcnt_or_barrier"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stalled"),
                       ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing"), ("SQ_WAIT_INST_LDS", "wave_cycles_lds_issue_stall")):
            v = mean(waves[name].get(c, []))
            if v is not None:
                e[key] = round(v / wc, 4)
        lc, li = mean(waves[name].get("SQ_LDS_BANK_CONFLICT", [])), mean(waves[name].get("SQ_LDS_IDX_ACTIVE", []))
        if lc is not None and li:
            e["lds_bank_conflict_cycles_over_lds_active"] = round(lc / li, 4)
    res["kernels"][name] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
