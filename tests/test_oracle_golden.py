"""Pins the oracle to every golden vector the reference's test-suite holds for
this path (ref testsuite/scaffolder_include.rb: graph / parser / scaffold
modules; the fixtures under tests/golden/ are the reference's testdata/)."""
import filecmp
import os
import re
import subprocess

import pytest

from helpers import ROOT
from oracle import oracle_py

CLI = os.path.join(ROOT, "oracle", "gts_oracle_cli")


@pytest.fixture(scope="module", autouse=True)
def built():
    oracle_py.build()


def run_cli(args, cwd):
    return subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True)


@pytest.mark.parametrize("stage", ["mark_repeats", "filter", "removecycles", "makescaffold"])
@pytest.mark.parametrize("mode", ["false", "lazy"])
def test_scaffold_stages_match_reference_dot(tmp_path, golden_dir, stage, mode):
    # ref testsuite/scaffolder_include.rb:88-121
    r = run_cli(["scaffold", golden_dir + "/primary-contigs.fa", golden_dir + "/libPE.de",
                 golden_dir + "/libPE.astat", mode], tmp_path)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(tmp_path / ("gt_scaffolder_algorithms_test_%s.dot" % stage),
                       "%s/gt_scaffolder_algorithms_test_%s_expected.dot" % (golden_dir, stage),
                       shallow=False)


def test_graph_module_toy_graph(tmp_path, golden_dir):
    # ref testsuite/scaffolder_include.rb:1-56
    assert run_cli("graph 5 8 0 0 0 0 0".split(), tmp_path).returncode == 0
    assert run_cli("graph 5 8 1 5 1 8 0".split(), tmp_path).returncode == 0
    assert run_cli("graph 5 8 1 6 0 0 0".split(), tmp_path).returncode == 2
    assert run_cli("graph 5 8 1 5 1 9 0".split(), tmp_path).returncode == 2
    assert run_cli("graph 5 8 1 5 1 8 1".split(), tmp_path).returncode == 0
    assert filecmp.cmp(tmp_path / "gt_scaffolder_graph_test.dot",
                       golden_dir + "/gt_scaffolder_graph_test_expected.dot", shallow=False)


def test_parser_module_roundtrip(tmp_path, golden_dir):
    # ref testsuite/scaffolder_include.rb:58-80
    for f in ["wrong_libPE_1.de", "wrong_libPE_2.de", "libPE.de"]:
        assert run_cli(["parser", os.path.join(golden_dir, f)], tmp_path).returncode == 0
    assert filecmp.cmp(tmp_path / "gt_scaffolder_parser_test_read_distances.de",
                       golden_dir + "/libPE.de", shallow=False)


def test_erroneous_de_files_are_rejected_by_graph_construction(golden_dir):
    for f in ["wrong_libPE_1.de", "wrong_libPE_2.de"]:
        with pytest.raises(RuntimeError, match="Invalid record in dist file"):
            oracle_py.OracleGraph.from_files(golden_dir + "/primary-contigs.fa",
                                             os.path.join(golden_dir, f))


def _sga_graph(path):
    tab = {}
    for line in open(path):
        m = re.match(r'\s*"(.+)" -> "(.+)" \[.+\];', line)
        if m:
            tab[m.group(1)].append(m.group(2)); continue
        m = re.match(r'\s*"(.+)" \[.+\];', line)
        if m:
            tab[m.group(1)] = []
    return tab


def _gt_graph(path):
    visible = {"black", "magenta", "red", "green"}
    tab, name = {}, {}
    for line in open(path):
        m = re.match(r'(\d+) -> (\d+) \[color="(.+?)".+\];', line)
        if m and m.group(3) in visible:
            tab[name[m.group(1)]].append(name[m.group(2)]); continue
        m = re.match(r'(\d+) \[color="(.+)" label="(.+)"\];', line)
        if m and m.group(2) in visible:
            tab[m.group(3)] = []; name[m.group(1)] = m.group(3)
    return tab


def test_same_scaffold_graph_as_sga(tmp_path, golden_dir):
    # ref testsuite/diff_graph_files.rb against testdata/sga_makeScaffolds.dot
    r = run_cli(["scaffold", golden_dir + "/primary-contigs.fa", golden_dir + "/libPE.de",
                 golden_dir + "/libPE.astat", "false"], tmp_path)
    assert r.returncode == 0
    a = _sga_graph(golden_dir + "/sga_makeScaffolds.dot")
    b = _gt_graph(tmp_path / "gt_scaffolder_algorithms_test_makescaffold.dot")
    assert set(a) == set(b)
    for k in a:
        assert sorted(a[k]) == sorted(b[k])


def test_lazy_and_faithful_distance_maps_agree():
    import numpy as np
    from helpers import make_inputs, oracle_from_inputs
    for seed in range(3):
        g = make_inputs(1500, 40 + seed, p_chimeric=0.05)
        res = []
        for lazy in (False, True):
            og = oracle_from_inputs(g)
            og.mark_repeats(); og.filter(); og.makescaffold(lazy)
            res.append((og.vertex_states(), og.edge_states()))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_an_unmarked_edge_has_unmarked_ends():
    """mark_vertex marks a vertex' edges and their twins (algorithms.c:76-87) and
    no edge goes back from a marked state: the engine's live-edge test relies on
    it (k_live_union reads the edge state only)"""
    import numpy as np
    from helpers import make_inputs, oracle_from_inputs
    marked_v = np.zeros(8, bool); marked_v[[1, 3, 7]] = True
    marked_e = np.zeros(8, bool); marked_e[[1, 2, 3, 7]] = True
    for seed in range(4):
        g = make_inputs(3000, 300 + seed, p_chimeric=0.05, p_bubble=0.05, p_relist=0.03,
                        p_inversion=0.5 * (seed % 2))
        og = oracle_from_inputs(g)
        stages = [og.mark_repeats, og.filter, og.removecycles, lambda: og.makescaffold(True)]
        for stage in stages:
            stage()
            e = og.edges()
            vs, es = og.vertex_states(), og.edge_states()
            bad = ~marked_e[es] & (marked_v[vs[e["start"]]] | marked_v[vs[e["end"]]])
            assert not bad.any()
