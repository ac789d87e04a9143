"""smoke(): one small scaffolding job on cuda:0, checked against the CPU oracle."""
import os
import sys

import numpy as np

from . import engine, synth


def smoke(n_contigs=3000, seed=3):
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root_dir not in sys.path:
        sys.path.insert(0, root_dir)
    from oracle.oracle_py import OracleGraph  # the checker, never the product path

    g = synth.to_numpy(synth.make_graph(n_contigs, seed=seed, device="cpu"))
    og = OracleGraph.from_records(g["seq_len"], g["astat"], g["copy_num"], g["root"], g["ctg"],
                                  g["dist"], g["std_dev"], g["num_pairs"], g["flags"])
    og.mark_repeats(); og.filter(); og.makescaffold(True)
    eng = engine.Engine(0)
    eng.set_contigs(g["seq_len"].astype(np.int64), g["astat"], g["copy_num"])
    eng.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"],
                           g["num_pairs"].astype(np.int64), g["flags"])
    eng.mark_repeats(); eng.filter(); eng.makescaffold()
    assert eng.ne == og.ne, (eng.ne, og.ne)
    assert np.array_equal(eng.vertex_states(), og.vertex_states()), "vertex states differ"
    assert np.array_equal(eng.edge_states(), og.edge_states()), "edge states differ"
    print("smoke ok: %d contigs, %d edges, %d components" % (eng.nv, eng.ne, eng.stat("components")))
    eng.close()
