"""Hand-derived fixtures (tests/golden/handmade/README.md): tiny inputs in the
reference's file formats whose expected .dot after every stage was derived by
hand from the reference's source -- POLYMORPHIC, INCONSISTENT, CYCLIC, VISITED,
last-writer-wins, LIFO / strict tie-breaks.  The oracle is checked on CPU, the
engine through the drop-in file API on the GPU."""
import filecmp
import os

import pytest

from helpers import DEFAULTS, pkg
from oracle.oracle_py import OracleGraph

NAMES = ["polymorphic", "inconsistent", "cycle", "equal_walks", "diamond_tie", "diamond_improve",
         "overwrite_polymorphic"]
STAGES = ["mark_repeats", "filter", "removecycles", "makescaffold"]
COLOUR_OF = {"gray80": "POLYMORPHIC", "gainsboro": "INCONSISTENT", "blue": "CYCLIC", "red": "VISITED",
             "magenta": "SCAFFOLD", "black": "UNVISITED"}


@pytest.fixture(scope="module")
def hm_dir(golden_dir):
    return os.path.join(golden_dir, "handmade")


def check(tmp_path, hm_dir, name, graph, steps):
    for stage, fn in zip(STAGES, steps):
        if fn:
            fn()
        out = str(tmp_path / ("%s_%s.dot" % (name, stage)))
        graph.print_dot(out)
        want = "%s/%s_%s_expected.dot" % (hm_dir, name, stage)
        assert filecmp.cmp(out, want, shallow=False), "%s after %s:\n%s\nexpected:\n%s" % (
            name, stage, open(out).read(), open(want).read())


def test_fixtures_cover_every_final_state(hm_dir):
    seen = set()
    for name in NAMES:
        for stage in STAGES:
            text = open("%s/%s_%s_expected.dot" % (hm_dir, name, stage)).read()
            seen |= {s for c, s in COLOUR_OF.items() if 'color="%s"' % c in text}
    assert seen == set(COLOUR_OF.values())   # REPEAT is in the reference's own vectors


@pytest.mark.parametrize("name", NAMES)
def test_oracle_on_handmade_fixture(tmp_path, hm_dir, name):
    og = OracleGraph.from_files("%s/%s.fa" % (hm_dir, name), "%s/%s.de" % (hm_dir, name),
                                DEFAULTS["min_ctg_len"])
    og.mark_repeats_file("%s/%s.astat" % (hm_dir, name), DEFAULTS["copy_num_cutoff"],
                         DEFAULTS["astat_cutoff"])
    check(tmp_path, hm_dir, name, og, [None, og.filter, og.removecycles, lambda: og.makescaffold(False)])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_on_handmade_fixture(tmp_path, hm_dir, name):
    G = pkg.engine.ScaffolderGraph.from_files("%s/%s.fa" % (hm_dir, name), "%s/%s.de" % (hm_dir, name),
                                              DEFAULTS["min_ctg_len"])
    G.mark_repeats("%s/%s.astat" % (hm_dir, name), DEFAULTS["copy_num_cutoff"], DEFAULTS["astat_cutoff"])
    check(tmp_path, hm_dir, name, G, [None, G.filter, G.removecycles, G.makescaffold])


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(fast_walks=0), dict(lds_components=0), dict(defer_min_contigs=2, defer_min_work=0),
                                  dict(pool_components=0)])
def test_engine_variants_on_handmade_fixtures(tmp_path, hm_dir, opts):
    """the same final states from the reference-search walks, the global-memory
    component programs and the deferred walk tasks"""
    import numpy as np
    for name in NAMES:
        og = OracleGraph.from_files("%s/%s.fa" % (hm_dir, name), "%s/%s.de" % (hm_dir, name))
        v, e = og.vertices(), og.edges()
        og.mark_repeats_file("%s/%s.astat" % (hm_dir, name))
        v2 = og.vertices()
        eng = pkg.engine.Engine(0)
        for k, val in opts.items():
            eng.set_option(k, val)
        eng.set_contigs(v["seq_len"].astype(np.int64), v2["astat"], v2["copy_num"])
        sel = np.arange(0, og.ne, 2)
        eng.build_from_records(e["start"][sel], e["end"][sel], e["dist"][sel], e["std_dev"][sel],
                               e["num_pairs"][sel].astype(np.int64), e["flags"][sel])
        eng.mark_repeats(True, DEFAULTS["copy_num_cutoff"], DEFAULTS["astat_cutoff"])
        eng.filter(); eng.makescaffold()
        # expected final states from the hand-derived .dot
        col = {c: i for i, c in enumerate(["black", "gray80", "gainsboro", "ivory3", "red", "green",
                                           "magenta", "blue"])}
        vs, es = [], []
        for line in open("%s/%s_makescaffold_expected.dot" % (hm_dir, name)):
            if "->" in line:
                es.append(col[line.split('color="')[1].split('"')[0]])
            elif "label" in line:
                vs.append(col[line.split('color="')[1].split('"')[0]])
        assert list(eng.vertex_states()) == vs, (name, opts)
        assert list(eng.edge_states()) == es, (name, opts)
