#!/usr/bin/env python3
"""Where the time of the drop-in file API goes: every step of
src/test.c:118-199 (module scaffold) timed on generated .fa / .de / .astat
files.

    python tools/file_api_times.py [--contigs N] [--host-parser]

Prints one JSON line (seconds per step).  --host-parser: the distance file
through the host restatement of parser.c instead of the GPU parser.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from __graft_entry__ import load_package
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=300_000)
    ap.add_argument("--host-parser", action="store_true")
    args = ap.parse_args()
    pkg = load_package()
    engine = pkg.engine
    L = engine.lib()
    t = {}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        g = pkg.synth.to_numpy(pkg.synth.make_graph(args.contigs, seed=5, device="cpu"))
        pkg.synth.write_files(g, os.path.join(d, "syn"))
        t["generate_and_write_files"] = time.perf_counter() - t0
        fa, de, astat = [os.path.join(d, "syn" + x).encode() for x in (".fa", ".de", ".astat")]
        sizes = {k: os.path.getsize(v) for k, v in (("fa", fa), ("de", de), ("astat", astat))}
        L.gt_scaffolder_set_distance_parser(1 if args.host_parser else 0)
        err = C.create_string_buffer(512)
        nc, nd = C.c_uint64(), C.c_uint64()

        def step(name, fn):
            t0 = time.perf_counter()
            rc = fn()
            t[name] = time.perf_counter() - t0
            assert rc == 0 or rc is None, (name, err.value)

        step("count_contigs", lambda: L.gt_scaffolder_parser_count_contigs(fa, 200, C.byref(nc), err, 512))
        h = C.c_void_p(L.gt_scaffolder_graph_new(nc.value, 0))
        step("read_contigs", lambda: L.gt_scaffolder_parser_read_contigs(h, fa, 200, False, err, 512))
        step("count_distances", lambda: L.gt_scaffolder_parser_count_distances(h, de, C.byref(nd), err, 512))
        step("read_distances", lambda: L.gt_scaffolder_parser_read_distances(de, h, False, err, 512))
        step("mark_repeats", lambda: L.gt_scaffolder_graph_mark_repeats(astat, h, 0.3, 20.0, err, 512))
        step("filter", lambda: L.gt_scaffolder_graph_filter(h, 0.01, 1.5, 400))
        step("removecycles", lambda: L.gt_scaffolder_removecycles(h))
        step("makescaffold", lambda: L.gt_scaffolder_makescaffold(h))
        dot = os.path.join(d, "out.dot").encode()
        step("print_dot", lambda: L.gt_scaffolder_graph_print(h, dot, err, 512))
        recs = [None]

        def it():
            recs[0] = C.c_void_p(L.gt_scaffolder_graph_iterate_scaffolds(h, None))
        step("iterate_scaffolds", it)
        scaf = os.path.join(d, "out.scaf").encode()
        step("write_scaffold", lambda: L.gt_scaffolder_graph_write_scaffold(recs[0], scaf, err, 512))
        ne = int(L.gt_scaffolder_graph_nof_edges(h))
        L.gt_scaffolder_graph_records_delete(recs[0])
        L.gt_scaffolder_graph_delete(h)
    L.gt_scaffolder_set_distance_parser(0)
    print(json.dumps(dict(contigs=int(nc.value), edges=ne, file_bytes=sizes,
                          distance_parser="host" if args.host_parser else "gpu",
                          seconds={k: round(v, 4) for k, v in t.items()})))


if __name__ == "__main__":
    main()
