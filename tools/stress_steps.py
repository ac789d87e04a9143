#!/usr/bin/env python3
"""Repeats the pipeline on one graph and says where it is (GPU box): a watchdog thread
writes the iteration and the stage in progress to the log every few seconds, so a step
that never ends can be told from one that is slow.
usage: python tools/stress_steps.py [iterations] [inversions 0/1] [name=value engine options ...]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
inv = int(sys.argv[2]) if len(sys.argv) > 2 else 1
opts = [a.split("=") for a in sys.argv[3:]]
pkg = load_package()
gen = dict(bench.WORKLOADS["10M"]["gen"])
if inv:
    gen.update(p_inversion=0.1, unique_pairs=False)
g = pkg.synth.make_graph(10_000_000, seed=1234, device="cuda:0", **gen)
g["num_pairs"] = g["num_pairs"].to(torch.int64)
eng = pkg.engine.Engine(0)
for k, v in opts:
    eng.set_option(k, int(v))
state = {"it": -1, "stage": "start", "t": time.time()}


def watchdog():
    while True:
        time.sleep(5)
        print("watchdog: iteration %d stage %s for %.0f s" % (state["it"], state["stage"], time.time() - state["t"]), flush=True)


threading.Thread(target=watchdog, daemon=True).start()
C = bench.CUTS
for it in range(iters):
    def at(s):
        state.update(it=it, stage=s, t=time.time())
    at("set_contigs"); eng.set_contigs(g["seq_len"], g["astat"], g["copy_num"])
    at("build"); eng.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"], g["flags"])
    at("mark_repeats"); eng.mark_repeats(True, C["copy_num_cutoff"], C["astat_cutoff"])
    at("filter"); eng.filter(C["pcutoff"], C["cncutoff"], C["ocutoff"])
    at("makescaffold"); eng.makescaffold()
    at("sync"); torch.cuda.synchronize()
    if it % 10 == 0:
        print("iteration %d done, digest %s" % (it, eng.digest()), flush=True)
print("all %d iterations done" % iters, flush=True)
