#!/usr/bin/env python3
"""Build-stage kernel times of bench.py records: ab_build.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    ls = [l for l in open(f).read().splitlines() if l.startswith('{')]
    if not ls:
        print(f, 'no json'); continue
    d = json.loads(ls[-1]); k = d['kernels_ms_per_step']
    b = sum(v for kk, v in k.items() if kk.startswith('build'))
    names = ['build_sort_pairs', 'build_pair_segments', 'build_scan_creators', 'build_emit_edges', 'build_sort_csr',
             'build_row_offsets', 'build_gather_csr', 'build_twins', 'comp_live_union', 'comp_compact_fill', 'k_components_pool']
    print("%-40s step %.2f build %.2f | " % (f.split('/')[-1][:-5], d['ms_per_step'], b) +
          " ".join("%s %.2f" % (n.replace('build_', '').replace('comp_', 'c.'), k.get(n, 0)) for n in names))
