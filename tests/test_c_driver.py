"""A plain-C caller of the drop-in API: tests/c/scaffold_driver.c is the
"scaffold" module of the reference's test driver (ref src/test.c:118-199)
restated on include/gt_scaffolder_host.h, compiled with gcc -std=gnu11 against
include/ and linked to libgtscaffold_hip.so -- what "host code stays C and
drops in" means."""
import filecmp
import os
import subprocess

import pytest

from helpers import ROOT, pkg

LIBDIR = os.path.join(ROOT, "gt-scaffold_amd", "csrc")
STAGES = ("mark_repeats", "filter", "removecycles", "makescaffold")


def build_driver(tmp_path):
    pkg.engine.lib()   # the library has to be there
    exe = str(tmp_path / "scaffold_driver")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-O1",
                    "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "scaffold_driver.c"),
                    "-L", LIBDIR, "-lgtscaffold_hip", "-Wl,-rpath," + LIBDIR, "-o", exe], check=True)
    return exe


def run_driver(exe, cwd, golden_dir, *extra):
    return subprocess.run([exe, golden_dir + "/primary-contigs.fa", golden_dir + "/libPE.de",
                           golden_dir + "/libPE.astat", *extra], cwd=cwd, capture_output=True,
                          text=True, timeout=600)


def test_c_driver_compiles_links_and_fails_loudly_without_gpu(tmp_path, golden_dir):
    exe = build_driver(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = run_driver(exe, tmp_path, golden_dir)
    assert r.returncode != 0
    assert "no CPU path" in r.stderr


def test_c_driver_edge_and_vertex_accessors_on_a_handbuilt_graph(tmp_path):
    # find_edge / get_vertex / get_vertex_id / alter_edge (ref gt_scaffolder_graph.h:127-146)
    # on a graph built with add_vertex / add_edge: host only, no GPU needed
    exe = build_driver(tmp_path)
    r = subprocess.run([exe, "handbuilt"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "handbuilt ok" in r.stdout, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("stepwise",), ("api",)])
def test_c_driver_reproduces_reference_dot_files(tmp_path, golden_dir, mode):
    exe = build_driver(tmp_path)
    r = run_driver(exe, tmp_path, golden_dir, *mode)
    assert r.returncode == 0, r.stderr
    for name in STAGES:
        out = tmp_path / ("gt_scaffolder_algorithms_test_%s.dot" % name)
        assert filecmp.cmp(out, "%s/gt_scaffolder_algorithms_test_%s_expected.dot" % (golden_dir, name),
                           shallow=False), name
    assert (tmp_path / "gt_scaffolder_new_write.scaf").stat().st_size > 0
    if mode == ("api",):
        assert "api ok" in r.stdout, r.stdout + r.stderr
        # gt_scaffolder_graph_print_generic (stream) = gt_scaffolder_graph_print (file name)
        assert "print_generic ok" in r.stdout
        assert filecmp.cmp(tmp_path / "gt_scaffolder_print_generic.dot",
                           tmp_path / "gt_scaffolder_algorithms_test_makescaffold.dot", shallow=False)
    if mode == ("stepwise",):
        # count_contigs tests >= min_ctg_len, read_contigs > (ref parser.c:408, :481)
        assert "contigs counted 50, distances counted" in r.stdout, r.stdout


@pytest.mark.gpu
def test_c_driver_matepair_switch(tmp_path, golden_dir):
    """ismatepair = true (ref parser.c:362): a pair listed again never alters
    its edges.  libPE.de lists every pair from both contigs, so the twin record
    would otherwise replace the backward estimate when its std_dev is larger;
    the oracle reads the same file with the same switch."""
    from oracle.oracle_py import OracleGraph
    exe = build_driver(tmp_path)
    de = tmp_path / "relisted.de"
    lines = open(golden_dir + "/libPE.de").read().splitlines()
    # re-list the first record of the first line with a larger std_dev
    head = lines[0].split(" ")
    rec = head[1].split(",")
    rec[1] = str(int(rec[1]) + 7)
    rec[3] = "%.1f" % (float(rec[3]) + 50.0)
    lines.append(" ".join([head[0], ",".join(rec), ";"]))
    de.write_text("\n".join(lines) + "\n")
    for mp in (False, True):
        d = tmp_path / ("mp%d" % mp)
        d.mkdir()
        r = subprocess.run([exe, golden_dir + "/primary-contigs.fa", str(de), golden_dir + "/libPE.astat",
                            "stepwise"] + (["matepair"] if mp else []), cwd=d, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        og = OracleGraph.from_files(golden_dir + "/primary-contigs.fa", str(de), ismatepair=mp)
        og.mark_repeats_file(golden_dir + "/libPE.astat")
        og.filter(); og.removecycles(); og.makescaffold(False)
        og.print_dot(str(d / "oracle.dot"))
        assert filecmp.cmp(d / "oracle.dot", d / "gt_scaffolder_algorithms_test_makescaffold.dot",
                           shallow=False)
    # the switch changes the graph: the re-listed estimate shows up only without it
    a = (tmp_path / "mp0" / "gt_scaffolder_algorithms_test_mark_repeats.dot").read_text()
    b = (tmp_path / "mp1" / "gt_scaffolder_algorithms_test_mark_repeats.dot").read_text()
    assert a != b
