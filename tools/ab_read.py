"""Prints the A/B runs written by tools/ab.sh: step, pool kernel, wave time sums per run."""
import json, sys, glob
for f in sorted(glob.glob("gpurun_out/%s_*_?.json" % sys.argv[1])):
    try:
        d = json.load(open(f))
    except Exception as ex:
        print(f, "unreadable", ex); continue
    ck = d.get("component_kernel", {})
    k = d["kernels_ms_per_step"]
    bs = ck.get("by_size", {})
    print("%-28s step %7.2f pool %6.2f tasks %s run_s %5.1f rc %5.2f walks %5.2f | walk_us <=64 %s >64 %s" % (
        f.split("/")[-1][:-5], d["ms_per_step"], k.get("k_components_pool", 0), k.get("k_walk_tasks"),
        ck.get("pool", {}).get("us_sum_run", 0) / 1e6, ck.get("us_sum_removecycles", 0) / 1e6,
        ck.get("us_sum_walks_fast", 0) / 1e6,
        sum(v["walk_us"] for kk, v in bs.items() if kk != "<=inf") / 1e6 if bs else None,
        bs.get("<=inf", {}).get("walk_us", 0) / 1e6 if bs else None))
