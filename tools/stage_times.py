import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
import bench
pkg = load_package()
g = bench.make_inputs(pkg, 10_000_000, 1234, "cuda:0", bench.WORKLOAD["gen"])
g["num_pairs"] = g["num_pairs"].to(torch.int64)
eng = pkg.engine.Engine(0)
eng.set_option("profile", 1)
C = bench.CUTS
for it in range(3):
    eng.reset_kernel_times()
    ts = []
    def T(f, *a):
        torch.cuda.synchronize(); t = time.perf_counter(); f(*a); torch.cuda.synchronize(); ts.append((f.__name__, (time.perf_counter() - t) * 1e3))
    T(eng.set_contigs, g["seq_len"], g["astat"], g["copy_num"])
    T(eng.build_from_records, g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"], g["flags"])
    T(eng.mark_repeats, True, C["copy_num_cutoff"], C["astat_cutoff"])
    T(eng.filter, C["pcutoff"], C["cncutoff"], C["ocutoff"])
    T(eng.makescaffold)
    kt = eng.kernel_times()
    def ksum(prefixes): return sum(v[1] for k, v in kt.items() if any(k.startswith(p) for p in prefixes))
    print("iter", it, " ".join("%s=%.1f" % x for x in ts), "| kernel sums: build %.1f repeat %.1f filter %.1f comp(non-class) %.1f" % (ksum(["build", "iota"]), ksum(["repeat"]), ksum(["filter"]), ksum(["comp_"])))
