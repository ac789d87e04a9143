"""Table of the A/B runs of tools/opt_ab.sh / tools/ab.sh: python tools/ab_table.py <tag>"""
import glob, json, sys, os
tag = sys.argv[1]
rows = {}
for f in sorted(glob.glob("gpurun_out/%s_*.json" % tag)):
    name = os.path.basename(f)[len(tag) + 1:-5]
    try:
        d = json.load(open(f))
    except Exception as ex:
        print(name, "unreadable", ex); continue
    k = d["kernels_ms_per_step"]; ck = d["component_kernel"]; fa = ck.get("fast", {})
    rows[name] = (d["ms_per_step"], k.get("k_components_fast", 0), k.get("k_components_pool", 0),
                  d["spans_ms_per_step"].get("span_components_makescaffold", 0),
                  fa.get("fast_components_handed_over", -1), fa.get("fast_us_sum_run", -1), fa.get("fast_us_sum_wait_pages", -1),
                  fa.get("fast_us_sum_wave_life", -1), fa.get("fast_us_first_exit", -1), fa.get("fast_us_last_exit", -1),
                  fa.get("cold_us_last_exit_after_fast_start", -1), ck["pool"]["us_sum_run"], ck["pool"]["us_sum_wait_pages"])
print("%-28s %8s %7s %7s %7s %6s %9s %9s %9s %6s %6s %6s %9s %9s" % ("run", "step", "fast", "pool", "span", "hand", "f_run", "f_wait", "f_life", "f_1st", "f_last", "c_last", "p_run", "p_wait"))
for n, r in rows.items():
    print("%-28s %8.2f %7.2f %7.2f %7.2f %6d %9d %9d %9d %6d %6d %6d %9d %9d" % ((n,) + r))
