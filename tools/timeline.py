"""Timeline of one step from a rocprofv3 --kernel-trace CSV: kernels in start
order with start offset, duration and the idle gap before each."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = the tail starting at the last histogram pass of the pair sort (the build's first kernel)
starts = [i for i, r in enumerate(rows) if "k_onesweep_hist<unsigned long" in r["Kernel_Name"] or "k_pair_keys" in r["Kernel_Name"]]
rows = rows[starts[-1]:] if starts else rows
t0 = int(rows[0]["Start_Timestamp"]); end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - end
    name = r["Kernel_Name"].split("(")[0][:48]
    if (e - s) > 100000 or gap > 100000:
        print("%9.3f ms  dur %8.3f  gap %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap / 1e6, name))
    end = max(end, e)
print("total %.3f ms" % ((end - t0) / 1e6))
