#!/usr/bin/env python3
"""Benchmark of the scaffold-graph hot path on MI355X.

metric : scaffold-graph edges processed/sec (build + mark_repeats + filter +
         makescaffold), BASELINE.json.
step   : one pass of the hot path over one synthetic batch whose inputs
         (contig table + DistEst records in file order) are already resident
         in HBM when the timed region starts.
N > 1  : one process per GPU.  Default (--mode partition, BASELINE configs[3]):
         ONE graph, its connected components sharded over the ranks -- label /
         plan / route over RCCL and one all-reduce inside the filter
         (gt-scaffold_amd/dist.py) -> "scaling": "strong".  --mode shards: every
         rank scaffolds its own graph, RCCL only for the barrier and the
         max-over-ranks of the time -> "scaling": "weak".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# The engine overlaps the launches of its LDS size classes on side streams; the
# HIP runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4; the engine's main stream takes one).  Eight let every class
# launch that fits run (measured 88.4 -> 85.0 ms per step); has to be set
# before the runtime starts, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402

# the configuration BASELINE.json's metric is quoted on (configs[2], the
# north_star's 10M-contig / 100M-edge graph; it fits one GPU)
WORKLOAD = dict(name="synthetic 10M-contig / 100M-edge scaffold graph",
                n_contigs=10_000_000,
                gen=dict(links_per_side=5, p_repeat=0.03, repeat_degree=43, p_inversion=0.0,
                         unique_pairs=True))
# BASELINE configs[4]: the human-scale, repeat-rich graph (--workload 50M; one
# GPU holds it: the CSR, the sort buffers and all scratch stay in HBM).  1.5 % of
# the contigs are repeats with 100 links on average (exponential, up to ~1500),
# and one repeat in ~33000 (about 22 contigs) looks unique to mark_repeats: its
# links reach the filter's hub path and tie hundreds of scaffolds into one
# component, which runs from global memory on one wavefront.  The reference's
# semantics make such a component quadratic (every best walk that ends in the
# hub revives an arc out of it, algorithms.c:842-845, so later walks fan out
# into the scaffolds marked before): at 1.5e-4 (112 hubs, components of up to
# 55 000 contigs) two components alone take 3.2 s of a 3.4 s step
# (profiles/r02d_bench_50M_112hubs.json).
WORKLOAD_50M = dict(name="synthetic 50M-contig / 500M-edge repeat-rich scaffold graph",
                    n_contigs=50_000_000,
                    gen=dict(links_per_side=4, p_repeat=0.015, repeat_degree=100, p_inversion=0.0,
                             unique_pairs=True, p_repeat_unmarked=3e-5))
WORKLOADS = {"10M": WORKLOAD, "50M": WORKLOAD_50M}
CUTS = dict(copy_num_cutoff=0.3, astat_cutoff=20.0, pcutoff=0.01, cncutoff=1.5, ocutoff=400)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)


# the engine's LDS size classes (gts_engine.hip): up to eleven, read back as statistics
MAX_LDS_CLASSES = 11


def algorithmic_bytes(name, n, m, nrec, eng):
    """Algorithmic HBM bytes of ONE launch of a kernel ('Kernels'):
    every input element read once, every output element written once."""
    vb = max(1, (int(n - 1).bit_length() + 7) // 8)         # 8-bit digits per vertex id
    nce = max(eng.stat("compact_edges"), 0)
    ns = max(eng.stat("slots"), 0)
    pp = max(eng.stat("pair_sort_passes"), 1)
    table = {
        # records bucketed on pp digits of the pair key (engine statistic; 3 for 100 M
        # records): the two contig ids read for the histograms and again by the first
        # pass (which makes the 64-bit key and the record number), then 12 B per pair
        # written / read per pass
        "build_sort_pairs": nrec * (8 + 8 + 12 + (pp - 1) * 24),
        # start vertex keys; the edge ids are made by the first pass
        "build_sort_csr": m * (4 + 4 + 8 + (vb - 1) * 16),
        "build_pair_segments": nrec * (12 + 4 + 2),
        "build_emit_edges": nrec * 8 + m * (29 + 32),
        "build_gather_csr": m * (4 + 32 + 30),
        "build_twins": m * 12,
        "build_row_offsets": m * 4 + n * 4,
        "repeat_edges": m * (8 + 1) + n * 8,
        "filter_pairs": m * (4 + 8 + 4 + 1 + 1) + n * (4 + 4 + 8 + 2),
        "filter_final": m * (4 + 4 + 1 + 1 + 1) + n * (4 + 4 + 1 + 8),
        "comp_live_union": m * (4 + 4 + 1 + 1) + n * 6,
        "comp_compact_fill": m * (4 + 1 + 4) + nce * 26 + ns * 8,
    }
    # component programs: the compact graph + vertex records of the components
    # the launch handles, read once; states and marks written once
    if name == "k_components":
        return max(eng.stat("bytes_components_global_mem"), 0)
    if name in ("k_components_lds", "k_components_pool", "k_components_fast"):   # all size classes: bytes of ONE step's launches
        total = sum(max(eng.stat("bytes_components_lds_class%d" % i), 0) for i in range(MAX_LDS_CLASSES))
        if eng.stat("fast_kernel") > 0:
            # the clean program stages every component of its slice once (the ones it
            # hands over as well); the full program next to it stages the rest
            done, handed = max(eng.stat("bytes_fast_finished"), 0), max(eng.stat("bytes_fast_handed_over"), 0)
            return done + handed if name == "k_components_fast" else total - done
        return total
    if name == "k_walk_tasks":            # every task stages its component once; one step's launches
        return max(eng.stat("bytes_walk_tasks"), 0)
    return table.get(name)


def kernel_groups(kt):
    """hipEvent entries -> kernels as rocprof names them: the eleven size-class
    launches of k_components_lds are ONE kernel (the engine times every launch
    under its own event name); `span_*` entries (fork -> last join of
    overlapped launches) are kept apart."""
    groups, spans = {}, {}
    for name, (calls, ms) in kt.items():
        if name.startswith("span_"):
            spans[name] = (calls, ms)
            continue
        if name in ("components_makescaffold_pool", "components_removecycles_pool",
                    "components_makescaffold_cold", "components_removecycles_cold"):
            key = "k_components_pool"
        elif name in ("components_makescaffold_fast", "components_removecycles_fast"):
            key = "k_components_fast"
        elif name.startswith("components_makescaffold_lds") or name.startswith("components_removecycles_lds"):
            key = "k_components_lds"
        elif name == "components_walk_tasks":
            key = "k_walk_tasks"
        elif name in ("components_makescaffold", "components_removecycles"):
            key = "k_components"
        else:
            key = name
        c, m = groups.get(key, (0, 0.0))
        groups[key] = (c + calls, m + ms)
    return groups, spans


# bench-event name -> kernel names in the rocprofv3 --pmc passes
PMC_NAMES = {
    "build_sort_pairs": ["k_onesweep_hist<unsigned long", "k_radix_scatter<unsigned long"],
    "build_pair_segments": ["k_pair_segments"],
    "build_emit_edges": ["k_emit_edges"], "build_gather_csr": ["k_gather_csr"],
    "build_twins": ["k_twins"], "repeat_edges": ["k_repeat_edges"],
    "filter_pairs": ["k_filter_pairs"], "filter_ovf_init": ["k_filter_ovf_init"],
    "filter_final": ["k_filter_final"], "filter_tpoly": ["k_filter_tpoly"],
    "filter_lasthit": ["k_filter_lasthit"], "comp_live_union": ["k_live_union"],
    "comp_compact_fill": ["k_compact_fill"],
}


def recorded_traffic(name):
    """HBM bytes per launch of a kernel from the committed PMC passes
    (profiles/pmc_traffic_latest.json: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE
    in separate runs of this bench at the BASELINE configuration; FETCH_SIZE is
    taken as reported -- the guide's x2 correction is calibrated for 16 B/lane
    streams only, the engine's kernels read 1-8 B per lane -- so this is a lower
    bound; the second value of the pair applies the x2).  A recording, not a live
    measurement: None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    if name in ("k_components_lds", "k_components_pool", "k_components_fast", "k_walk_tasks", "k_components"):
        ks = [k for k in d if k.split("(")[0].split("<")[0].split("[")[0] == name]
        tot = sum((d[k]["fetch_bytes_per_launch_raw"] + d[k]["write_bytes_per_launch"]) * d[k]["launches"] for k in ks)
        tot2 = sum((d[k]["fetch_bytes_per_launch_x2"] + d[k]["write_bytes_per_launch"]) * d[k]["launches"] for k in ks)
        n = sum(d[k]["launches"] for k in ks)
        return (tot / n, tot2 / n) if n else None      # average over the launches of the recorded step
    ks = [k for k in d if any(k.startswith(pre) for pre in PMC_NAMES.get(name, []))]   # (template arguments vary)
    if not ks:
        return None
    per_step = sum((d[k]["fetch_bytes_per_launch_raw"] + d[k]["write_bytes_per_launch"]) * d[k]["launches"] for k in ks)
    per_step2 = sum((d[k]["fetch_bytes_per_launch_x2"] + d[k]["write_bytes_per_launch"]) * d[k]["launches"] for k in ks)
    return per_step, per_step2      # the PMC passes ran one step: bytes of the whole (composite) kernel


def make_inputs(pkg, n_contigs, seed, device, gen):
    g = pkg.synth.make_graph(n_contigs, seed=seed, device=device, **gen)
    return g


def run_step(eng, g):
    eng.set_contigs(g["seq_len"], g["astat"], g["copy_num"])
    eng.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"], g["flags"])
    eng.mark_repeats(True, CUTS["copy_num_cutoff"], CUTS["astat_cutoff"])
    eng.filter(CUTS["pcutoff"], CUTS["cncutoff"], CUTS["ocutoff"])
    eng.makescaffold()
    return eng.ne


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pkg, n_sample, gen, seed):
    """The oracle (a single-threaded port of the reference's algorithms) on a
    bounded sample of the same workload, on this host's cores."""
    from oracle.oracle_py import OracleGraph
    g = pkg.synth.to_numpy(pkg.synth.make_graph(n_sample, seed=seed, device="cpu", **gen))
    t0 = time.perf_counter()
    og = OracleGraph.from_records(g["seq_len"], g["astat"], g["copy_num"], g["root"], g["ctg"],
                                  g["dist"], g["std_dev"], g["num_pairs"], g["flags"])
    og.mark_repeats(True, CUTS["copy_num_cutoff"], CUTS["astat_cutoff"])
    og.filter(CUTS["pcutoff"], CUTS["cncutoff"], CUTS["ocutoff"])
    og.makescaffold(True)
    dt = time.perf_counter() - t0
    return dict(value=og.ne / dt, unit="edges/s", cores=1, kind="port", host_cpu=host_cpu_model(),
                host_cores=os.cpu_count(),
                sample="%d-contig / %d-edge graph from the same generator, %.1f s; oracle with "
                       "epoch-stamped distance maps (the reference's per-walk O(|V|) map "
                       "initialisation, algorithms.c:648-650, would be slower still)"
                       % (n_sample, og.ne, dt)), og, g


def cpu_baseline_full_size():
    """The oracle on the WHOLE 10 M-contig / 100 M-edge configuration: a recording
    (tests/golden/full_size_digest.json, made by tools/make_full_size_digest.py
    in the build container -- the run takes 36 minutes, the bench must not), next
    to the sample timed live on this host."""
    path = os.path.join(ROOT, "tests", "golden", "full_size_digest.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    secs = {k: v for k, v in d["oracle_seconds"].items() if k != "generate"}   # build, mark_repeats + filter, makescaffold
    total = float(sum(secs.values()))
    return dict(value=d["n_edges"] / total, unit="edges/s", cores=1, kind="port", seconds=round(total, 1),
                stages_s={k: round(v, 1) for k, v in secs.items()}, edges=d["n_edges"], contigs=d["n_contigs"],
                host="build container (no GPU), one core", recorded=True,
                source="tests/golden/full_size_digest.json (tools/make_full_size_digest.py: portable generator, "
                       "seed %d; the GPU test test_full_size_against_oracle_digest compares the engine's "
                       "states on the same inputs with this run's digests)" % d["seed"])


def secondary_workload(pkg, eng, dev, torch, n_contigs, steps=2):
    """The slow path next to the headline, outside its timed region: the same
    generator with 10 % of the false links drawn as inversions and repeated
    contig pairs kept -- components where a contig is walked in both directions
    leave the linear-time walks for the reference's FIFO search."""
    gen = dict(WORKLOADS["10M"]["gen"], p_inversion=0.1, unique_pairs=False)
    g = pkg.synth.make_graph(n_contigs, seed=1234, device=dev, **gen)
    g["num_pairs"] = g["num_pairs"].to(torch.int64)
    torch.cuda.empty_cache()
    run_step(eng, g)                                   # warm-up (workspace growth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for _ in range(steps):
        edges += run_step(eng, g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(workload="synthetic %d-contig scaffold graph, 10 %% of the false links inversions, repeated "
                         "contig pairs kept" % n_contigs,
                steps=steps, ms_per_step=dt / steps * 1e3, value=edges / dt, unit="edges/s",
                edges=eng.ne, walks_reference=eng.stat("slow_walks"), walks_fast=eng.stat("fast_walks"),
                walk_tasks=eng.stat("walk_tasks"), walk_task_rounds=eng.stat("walk_task_rounds"),
                deferred_components=eng.stat("deferred_components"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="10M",
                    help="10M: BASELINE configs[2], the configuration the metric is quoted on "
                         "(default); 50M: configs[4], the repeat-rich human-scale graph")
    ap.add_argument("--contigs", type=int, default=None,
                    help="contigs per GPU (default: the workload's own size)")
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary (inversions) workload reported next to the headline")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with hipEvents")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="engine option (gtsg_set_option), e.g. defer_min_contigs=256")
    ap.add_argument("--gen", action="append", default=[], metavar="NAME=VALUE",
                    help="generator option (synth.make_graph), e.g. permute_ids=0: a measurement aid, "
                         "the line then names another workload")
    ap.add_argument("--inversions", type=float, default=None,
                    help="fraction of the chimeric links that are inversions (default 0: the "
                         "reference's walk search is exponential on components holding one)")
    ap.add_argument("--duplicate-pairs", action="store_true",
                    help="keep false links that repeat a contig pair (two estimates with other "
                         "geometry for one pair; the headline workload drops them)")
    ap.add_argument("--mode", choices=["shards", "partition"], default=None,
                    help="partition (default for N > 1, BASELINE configs[3]): ONE graph of --contigs "
                         "contigs over all ranks, records split by file chunk, component-partition "
                         "step over RCCL inside the timed region, strong scaling; shards (default "
                         "for N = 1): every rank scaffolds its own graph of --contigs contigs, no "
                         "data-path collective, weak scaling")
    ap.add_argument("--verify", action="store_true",
                    help="also run the oracle on the FULL workload and compare digests (slow)")
    args = ap.parse_args()
    global WORKLOAD
    WORKLOAD = WORKLOADS[args.workload]
    if args.contigs is None:
        args.contigs = WORKLOAD["n_contigs"]

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the engine has no CPU path")
    # rehearsal on a box with fewer GPUs than ranks (not a measurement):
    # GTS_BENCH_BACKEND=gloo maps the ranks onto the devices there are
    backend = os.environ.get("GTS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    pkg = load_package()
    dev = "cuda:%d" % local_rank

    for kv in args.gen:
        name, value = kv.split("=")
        WORKLOAD["gen"][name] = float(value) if "." in value else int(value)
        WORKLOAD["name"] += " [%s]" % kv
    if args.inversions is not None:
        WORKLOAD["gen"]["p_inversion"] = args.inversions
    if args.duplicate_pairs:
        WORKLOAD["gen"]["unique_pairs"] = False
    mode = args.mode or ("partition" if world > 1 else "shards")
    # the engine runs on a stream of its own (a blocking stream: HIP orders it
    # with the null stream the inputs are generated on; every engine call
    # drains it before returning)
    eng = pkg.engine.Engine(local_rank)
    eng.set_option("profile", 0 if args.no_profile else 1)
    for kv in args.opt:
        name, value = kv.split("=")
        eng.set_option(name, int(value))

    def generate(seed):
        gg = make_inputs(pkg, args.contigs, seed, dev, WORKLOAD["gen"])
        gg["num_pairs"] = gg["num_pairs"].to(torch.int64)
        torch.cuda.empty_cache()   # the generator's scratch goes back to HIP: the engine allocates for itself
        return gg

    comm = contigs = rec = None
    stage_s = {}
    if mode == "partition":
        # ONE graph: every rank draws the same graph (same seed) and keeps the
        # contig table and ITS chunk of the record file, as if the ranks had
        # read consecutive chunks of one .de file
        g = generate(1234)
        nrec_all = g["root"].numel()
        lo, hi = nrec_all * rank // world, nrec_all * (rank + 1) // world
        comm = pkg.dist.TorchComm() if world > 1 else pkg.dist.ThreadComm(pkg.dist.ThreadComm.Shared(1), 0)
        contigs = dict(seq_len=g["seq_len"], astat=g["astat"], copy_num=g["copy_num"])
        rec = {name: g[name][lo:hi].clone() for name in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")}
        rec["k"] = torch.arange(lo, hi, dtype=torch.int64, device=dev)
        g = {**contigs, **rec}
        torch.cuda.empty_cache()
    else:
        g = generate(1234 + rank)
    nrec = g["root"].numel()

    def step():
        if mode == "partition":
            pkg.dist.scaffold_sharded(comm, eng, contigs, rec, CUTS,
                                      timers=None if args.no_profile else stage_s)
            return eng.ne
        return run_step(eng, g)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.reset_kernel_times()
    stage_s.clear()
    barrier()
    t0 = time.perf_counter()
    edges = 0
    for _ in range(args.steps):
        edges += step()
    barrier()
    dt = time.perf_counter() - t0
    rdev = dev if backend == "nccl" else "cpu"
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    etot = torch.tensor([edges], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(etot, op=dist.ReduceOp.SUM)
    dt_max, edges_all = float(tmax.item()), float(etot.item())

    if rank == 0:
        n, m = eng.nv, eng.ne
        kt = eng.kernel_times()
        if not args.no_profile and mode == "shards":
            # one more, untimed, step that also copies the per-component clocks back
            eng.set_option("profile", 2)
            step()
            eng.set_option("profile", 1)
        if not kt:
            kt = {"(profiling off)": (1, 0.0)}
        groups, spans = kernel_groups(kt)
        steps = max(args.steps, 1)

        def roofline_of(name):
            calls, ms = groups[name]
            avg_ms = ms / max(calls, 1)
            ab_step = algorithmic_bytes(name, n, m, nrec, eng)      # bytes of one step's launches
            per_launch = ab_step * steps / max(calls, 1) if ab_step else None
            tr = recorded_traffic(name)
            r = dict(bound="hbm", kernel=name, launches=calls, launches_per_step=calls / steps,
                     avg_ms=avg_ms, algorithmic_bytes=per_launch,
                     achieved=(per_launch / (avg_ms * 1e-3) / 1e9) if per_launch and avg_ms else None,
                     peak=HBM_PEAK_GBS, unit="GB/s", frac=None,
                     traffic=tr[0] if tr else None,
                     traffic_is="raw FETCH_SIZE + WRITE_SIZE per launch, recorded (profiles/pmc_traffic_latest.json); "
                                "a lower bound: gfx950 reports half the bytes of wide coalesced reads",
                     traffic_with_fetch_x2=tr[1] if tr else None)
            if r["achieved"] is not None:
                r["frac"] = r["achieved"] / HBM_PEAK_GBS
            return r

        # the dominant kernel: largest sum of launch durations, as rocprofv3 --stats ranks them
        dname = max(groups.items(), key=lambda kv: kv[1][1])[0]
        roof = roofline_of(dname)
        if dname in ("k_components_lds", "k_components_pool", "k_components_fast"):
            sp = spans.get("span_components_makescaffold")
            if sp and roof["algorithmic_bytes"]:
                span_ms = sp[1] / max(sp[0], 1)
                step_bytes = roof["algorithmic_bytes"] * roof["launches_per_step"]
                roof.update(span_ms=span_ms, achieved_over_span=step_bytes / (span_ms * 1e-3) / 1e9,
                            frac_over_span=step_bytes / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
            roof["note"] = ("one wavefront per connected component, graph staged in LDS: bound by "
                            "instruction issue and LDS latency of the resident waves (DESIGN.md, SQ "
                            "counters in profiles/), not by HBM; span_ms = fork to join of the component "
                            "launches")
        sq = os.path.join(ROOT, "profiles", "sq_counters_latest.json")
        roof_issue = None
        if os.path.exists(sq):
            rec = json.load(open(sq)).get(dname)
            roof["sq_counters_recorded"] = rec
            if rec and rec.get("SQ_WAVES") and rec.get("SQ_BUSY_CYCLES"):
                # what does bound the component kernel: instructions issued per SIMD and
                # cycle against one per cycle (MI355X_MICROARCH.md: a wave64 VALU instruction
                # takes the SIMD two cycles, scalar and LDS issue beside it)
                insts = sum(v for k, v in rec.items() if k.startswith("SQ_INSTS_"))
                launches = max(rec.get("launches", 1), 1)
                cyc = rec["SQ_WAVE_CYCLES"] * 4 / rec["SQ_WAVES"]          # cycles a wavefront lives
                n_simd = 256 * 4
                waves_per_simd = rec["SQ_WAVES"] / launches / n_simd
                ipc = insts / rec["SQ_WAVES"] / cyc * waves_per_simd
                roof_issue = dict(kernel=dname, insts_per_simd_cycle=ipc, peak=1.0, frac=ipc,
                                  waves_per_simd=waves_per_simd, wave_executing_frac=rec.get("active_frac"),
                                  wave_waiting_frac=rec.get("wait_any_frac"),
                                  source="profiles/sq_counters_latest.json (rocprofv3 --pmc SQ passes, recorded)")
                roof["limited_by"] = ("LDS capacity x time and the latency of dependent LDS reads at ~14 running "
                                      "wavefronts per CU (DESIGN.md); HBM carries 2 % of its peak")
        # for reference, the largest HBM-streaming kernel of the step
        stream = [k for k in groups if not k.startswith("k_components") and k != "k_walk_tasks"
                  and algorithmic_bytes(k, n, m, nrec, eng)]
        roof_stream = roofline_of(max(stream, key=lambda k: groups[k][1])) if stream else None
        out = dict(metric="scaffold-graph edges processed/sec (build+filter+makescaffold)",
                   value=edges_all / dt_max, unit="edges/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=dt_max / args.steps * 1e3,
                   higher_is_better=True, scaling="strong" if mode == "partition" else "weak",
                   vs_baseline=None, dtype="int64/u8",
                   data="synthetic",
                   config=dict(workload=(WORKLOAD["name"] + (", %.0f %% of the false links are inversions"
                                                             % (100 * args.inversions) if args.inversions else ""))
                               if args.contigs == WORKLOAD["n_contigs"] else
                               "synthetic scaffold graph, %d contigs (same generator)" % args.contigs,
                               contigs_total=args.contigs * (1 if mode == "partition" else world),
                               edges_total=int(edges_all / max(args.steps, 1)),
                               contigs_rank0=n, edges_rank0=m, records_rank0=nrec,
                               components_rank0=eng.stat("components"),
                               max_component_rank0=eng.stat("max_component"),
                               parallelism=("one graph, its connected components sharded over %d GPU(s): "
                                            "label / plan / route over RCCL + one all-reduce inside the "
                                            "filter" % world) if mode == "partition" else
                                           ("%d independent graph(s), one per GPU, no data-path collective" % world),
                               mode=mode,
                               hip_hw_queues=int(os.environ["GPU_MAX_HW_QUEUES"])),
                   roofline=roof, roofline_issue=roof_issue, roofline_largest_streaming_kernel=roof_stream,
                   component_kernel=dict(
                       walks_fast=eng.stat("fast_walks"), walks_reference=eng.stat("slow_walks"),
                       clean_components=eng.stat("clean_components"),
                       components_per_lds_class={"%dk" % eng.stat("lds_class%d_kb" % i):
                                                 eng.stat("components_lds_class%d" % i)
                                                 for i in range(MAX_LDS_CLASSES)
                                                 if eng.stat("lds_class%d_kb" % i) > 0},
                       components_global_mem=eng.stat("components_global_mem"),
                       pool={k[5:]: eng.stat(k) for k in
                             ("pool_us_sum_run", "pool_us_sum_wait_pages", "pool_us_sum_wave_life",
                              "pool_us_first_exit", "pool_us_last_exit", "pool_helper_joins", "pool_us_sum_helping")},
                       fast={k: eng.stat(k) for k in
                             ("fast_kernel", "fast_wavefronts", "fast_components_done", "fast_components_handed_over",
                              "fast_us_sum_run", "fast_us_sum_wait_pages", "fast_us_sum_claim", "fast_us_sum_wave_life",
                              "fast_us_first_exit", "fast_us_last_exit", "cold_us_last_exit_after_fast_start")},
                       wave_us_per_lds_class={"%dk" % eng.stat("lds_class%d_kb" % i):
                                              dict(wave_us=eng.stat("lds_class%d_wave_us" % i),
                                                   walk_us=eng.stat("lds_class%d_walk_us" % i))
                                              for i in range(MAX_LDS_CLASSES)
                                              if eng.stat("lds_class%d_kb" % i) > 0},
                       **{k: eng.stat(k) for k in
                          ("us_sum_removecycles", "us_max_removecycles", "us_sum_makescaffold_other",
                           "us_max_makescaffold_other", "us_sum_walks_fast", "us_max_walks_fast",
                           "us_sum_walks_reference", "us_max_walks_reference")},
                       left_linear_walk_because={k[4:]: eng.stat(k) for k in
                                                 ("why_mixed_start", "why_self_arc", "why_back_at_start",
                                                  "why_marked_end", "why_two_directions",
                                                  "why_inexact_tie", "why_cycle",
                                                  "why_inexact_length_tie")},
                       unclean_batches={k[14:]: eng.stat(k) for k in
                                        ("unclean_batch_walks", "unclean_batch_given_up", "unclean_batch_cycle")},
                       small={k[6:]: eng.stat(k) for k in
                              ("small_all_live_components", "small_other_components",
                               "small_all_live_removecycles_us", "small_other_removecycles_us")},
                       team={k[5:]: eng.stat(k) for k in
                             ("team_components", "team_ccs", "team_batches", "team_sweep_steps", "team_ccs_with_tie",
                              "team_us_clear", "team_us_sweep", "team_us_paths", "team_us_wave0_barriers")},
                       walk_tasks=eng.stat("walk_tasks"), walk_task_rounds=eng.stat("walk_task_rounds"),
                       walk_task_runs=eng.stat("walk_task_runs"),
                       walk_rounds=[dict(us=eng.stat("walk_round%d_us" % r), walks=eng.stat("walk_round%d_walks" % r))
                                    for r in range(min(8, max(0, eng.stat("walk_task_rounds"))))],
                       deferred_components=eng.stat("deferred_components"),
                       by_size={"<=%s" % b: dict(components=eng.stat("size_band%d_components" % i),
                                                 wave_us=eng.stat("size_band%d_us" % i),
                                                 walk_us=eng.stat("size_band%d_walk_us" % i))
                                for i, b in enumerate(("2", "3", "4", "8", "16", "32", "64", "inf"))},
                       last_to_finish=[
                           {k: eng.stat("last%d_%s" % (r, k)) for k in
                            ("size", "terminals", "clean", "start_us", "end_us", "removecycles_us", "walks_us")}
                           for r in range(8)],
                       slowest_components=[
                           {k: eng.stat("top%d_%s" % (r, k)) for k in
                            ("size", "edges", "terminals", "ccs", "clean", "deferred", "why_not_deferred", "walks",
                             "ref_walks", "removecycles_us", "other_us", "walks_us", "ref_us", "ref_pops")}
                           for r in range(12)]),
                   hbm_resident=dict(graph_bytes=eng.stat("bytes_graph"),
                                     workspace_bytes=eng.stat("bytes_workspace"),
                                     input_bytes=int(sum(t.numel() * t.element_size() for t in g.values()))),
                   kernels_ms_per_step={k: round(v[1] / args.steps, 3) for k, v in
                                        sorted(groups.items(), key=lambda kv: -kv[1][1])},
                   spans_ms_per_step={k: round(v[1] / args.steps, 3) for k, v in spans.items()},
                   events_ms_per_step={k: round(v[1] / args.steps, 3) for k, v in
                                       sorted(kt.items(), key=lambda kv: -kv[1][1])})
        # the longest component program of the step: a component runs on one wavefront
        # (or one workgroup), so no number of GPUs takes the component launch below this
        top = [sum(max(eng.stat("top%d_%s" % (r, k)), 0) for k in ("removecycles_us", "other_us", "walks_us", "ref_us"))
               for r in range(12)]
        if max(top) > 0:   # (the per-component clocks are copied back in "shards" mode only)
          out["critical_path_ms"] = dict(longest_component_program=round(max(top) / 1e3, 3),
                                         what="removecycles + makescaffold of the slowest component on its one "
                                              "wavefront / workgroup: the floor of the component launch whatever "
                                              "the number of GPUs (the stages before it divide by the rank count)")
        if mode == "partition" and stage_s:
            out["partition_stages_ms_per_step_rank0"] = {k: round(v / args.steps * 1e3, 3)
                                                         for k, v in stage_s.items()}
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only
            cb, og, gs = cpu_baseline(pkg, args.cpu_sample, WORKLOAD["gen"], 99)
            if args.workload == "10M":
                cb["full_size"] = cpu_baseline_full_size()
            out["cpu_baseline"] = cb
        if args.verify and mode == "shards":
            # (before the secondary workload: the engine still holds the headline graph's states)
            from oracle.oracle_py import OracleGraph
            gn = pkg.synth.to_numpy(g)
            og = OracleGraph.from_records(gn["seq_len"], gn["astat"], gn["copy_num"], gn["root"],
                                          gn["ctg"], gn["dist"], gn["std_dev"], gn["num_pairs"],
                                          gn["flags"])
            og.mark_repeats(True, CUTS["copy_num_cutoff"], CUTS["astat_cutoff"])
            og.filter(CUTS["pcutoff"], CUTS["cncutoff"], CUTS["ocutoff"])
            og.makescaffold(True)
            out["verified_against_oracle"] = (eng.digest() == pkg.engine.state_digest_host(
                og.vertex_states(), og.edge_states()))
        if (not args.no_secondary and world == 1 and mode == "shards" and args.workload == "10M"
                and args.inversions is None and not args.duplicate_pairs and not args.gen):
            # outside the timed region of the headline: its inputs are released first
            g = None
            torch.cuda.empty_cache()
            eng.set_option("profile", 0)
            out["secondary"] = [secondary_workload(pkg, eng, dev, torch, args.contigs)]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
