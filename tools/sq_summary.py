#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of SQ counters into per-kernel totals.

usage: sq_summary.py out.json <dir of pass 1> [<dir of pass 2> ...]

Every pass is its own run of the same command (8 SQ counter slots per pass on
gfx950, MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Per kernel (the size-class
launches of k_components_lds / k_walk_tasks are added up) the counters are
summed over the dispatches of the run; derived figures:
  cycles_per_wave      SQ_WAVE_CYCLES * 4 / SQ_WAVES   (the counter ticks quad-cycles)
  insts_per_wave       (VALU + SALU + LDS + SMEM + VMEM ...) / SQ_WAVES
  wait_any_frac        SQ_WAIT_ANY / SQ_WAVE_CYCLES     (wave parked at s_waitcnt / barrier)
  issue_stall_frac     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  active_frac          SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    # a counter collected in several passes (SQ_WAVES, SQ_WAVE_CYCLES go into every
    # pass as the denominators) is averaged over the passes that hold it
    per_pass = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    launches = defaultdict(lambda: defaultdict(lambda: defaultdict(int)))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                per_pass[k][c][d] += float(row["Counter_Value"])
                launches[k][c][d] += 1
    res = {}
    for k, byc in per_pass.items():
        cs = {c: sum(v.values()) / len(v) for c, v in byc.items()}
        r = dict(cs)
        r["launches"] = max(max(v.values()) for v in launches[k].values())
        w, wc = cs.get("SQ_WAVES"), cs.get("SQ_WAVE_CYCLES")
        if w and wc:
            r["cycles_per_wave"] = wc * 4 / w
        insts = sum(v for c, v in cs.items() if c.startswith("SQ_INSTS_"))
        if w and insts:
            r["insts_per_wave"] = insts / w
        for name, c in (("wait_any_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"),
                        ("active_frac", "SQ_ACTIVE_INST_ANY"), ("lds_issue_stall_frac", "SQ_WAIT_INST_LDS")):
            if wc and c in cs:
                r[name] = cs[c] / wc
        if cs.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_bank_conflict_frac"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
        res[k] = r
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, r in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:10]:
        print("%-28s waves %10.0f  cyc/wave %10.0f  inst/wave %9.0f  wait_any %.2f  issue_stall %.2f  active %.2f"
              % (k, r.get("SQ_WAVES", 0), r.get("cycles_per_wave", 0), r.get("insts_per_wave", 0),
                 r.get("wait_any_frac", 0), r.get("issue_stall_frac", 0), r.get("active_frac", 0)))


if __name__ == "__main__":
    main()
