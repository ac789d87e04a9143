#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE collected in two
separate passes, as MI355X_MICROARCH.md prescribes) into per-kernel HBM
traffic per launch.

usage: pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> out.json

gfx950 notes from the guide: FETCH_SIZE / WRITE_SIZE are reported in KiB-like
units of the L2's memory-side requests (FETCH_SIZE = TCC_EA0_RDREQ x 64 B);
FETCH_SIZE reads exactly 1/2 of the bytes of a wide (16 B / lane) coalesced
stream; other access widths are uncalibrated.  Both the raw value and the
x2-corrected read figure are recorded; the engine's kernels use 1-8 B / lane
accesses, so the truth lies between them.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

def short(name, lds):
    base = name.split("(")[0].replace("void ", "").strip()
    if base in ("k_components_lds", "k_walk_tasks"):
        return "%s[lds%dk]" % (base, int(lds) // 1024)
    return base


def load(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"], row.get("LDS_Block_Size", "0") or 0)
            per[k][0] += 1
            per[k][1] += float(row["Counter_Value"])
    return per


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        res[k] = dict(launches=n,
                      fetch_bytes_per_launch_raw=f * 1024 / max(nf, 1),
                      fetch_bytes_per_launch_x2=2 * f * 1024 / max(nf, 1),
                      write_bytes_per_launch=w * 1024 / max(nw, 1))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch_raw"] * kv[1]["launches"]))[:12]
    for k, v in top:
        print("%-40s launches %3d  fetch raw %8.1f MB (x2 %8.1f)  write %8.1f MB" % (
            k, v["launches"], v["fetch_bytes_per_launch_raw"] / 1e6,
            v["fetch_bytes_per_launch_x2"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
