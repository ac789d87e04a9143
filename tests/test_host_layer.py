"""Host layer of the product library (include/gt_scaffolder_host.h): the parts
that need no GPU -- library loads with every declared symbol, hand-built
graphs, .de normalisation, error behaviour of the parsers -- mirroring
ref testsuite/scaffolder_include.rb."""
import ctypes as C
import filecmp
import os
import re

import pytest

from helpers import ROOT, pkg

engine = pkg.engine


def test_library_exports_every_declared_symbol():
    L = engine.lib()
    for hdr in ("gt_scaffold_hip.h", "gt_scaffolder_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names = set(re.findall(r"\b(gtsg_\w+|gt_scaffolder_\w+)\s*\(", text))
        assert len(names) >= 15
        for n in names:
            assert hasattr(L, n), "%s declared in %s but not exported" % (n, hdr)


def test_no_gpu_means_loud_failure(golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError):
        engine.Engine(0)
    with pytest.raises(engine.EngineError, match="no CPU path"):
        engine.ScaffolderGraph.from_files(golden_dir + "/primary-contigs.fa", golden_dir + "/libPE.de")


def test_gpu_steps_of_the_file_api_fail_cleanly_without_a_gpu():
    """header sort, FASTA record table and the distance parser report an error
    (the host layer then runs its own code) instead of crashing"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError):
        engine.sort_names([b"b", b"a"])
    with pytest.raises(engine.EngineError):
        engine.fasta_records(b">a\nACGT\n")
    with pytest.raises(engine.EngineError):
        engine.DeParser([b"a", b"b"])


def test_graph_module_toy_graph(tmp_path, golden_dir):
    # ref testsuite/scaffolder_include.rb:1-56 (exit status 2 = failed assertion)
    L = engine.lib()
    err = C.create_string_buffer(256)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert L.gt_scaffolder_graph_test(5, 8, False, 0, False, 0, False, err, 256) == 0
        assert L.gt_scaffolder_graph_test(5, 8, True, 5, False, 0, False, err, 256) == 0
        assert L.gt_scaffolder_graph_test(5, 8, True, 6, False, 0, False, err, 256) == 2
        assert L.gt_scaffolder_graph_test(5, 8, True, 5, True, 8, False, err, 256) == 0
        assert L.gt_scaffolder_graph_test(5, 8, True, 5, True, 9, False, err, 256) == 2
        assert L.gt_scaffolder_graph_test(5, 8, True, 5, True, 8, True, err, 256) == 0
        assert filecmp.cmp("gt_scaffolder_graph_test.dot",
                           golden_dir + "/gt_scaffolder_graph_test_expected.dot", shallow=False)
    finally:
        os.chdir(cwd)


def test_parser_module_roundtrip(tmp_path, golden_dir):
    # ref testsuite/scaffolder_include.rb:58-80
    L = engine.lib()
    err = C.create_string_buffer(256)
    out = str(tmp_path / "norm.de").encode()
    for f in ("wrong_libPE_1.de", "wrong_libPE_2.de", "libPE.de"):
        assert L.gt_scaffolder_parser_read_distances_test((golden_dir + "/" + f).encode(), out, err, 256) == 0
    assert filecmp.cmp(tmp_path / "norm.de", golden_dir + "/libPE.de", shallow=False)


@pytest.mark.parametrize("de", ["wrong_libPE_1.de", "wrong_libPE_2.de"])
def test_erroneous_de_files_are_rejected(golden_dir, de):
    with pytest.raises(engine.EngineError, match="Invalid record in dist file"):
        engine.ScaffolderGraph.from_files(golden_dir + "/primary-contigs.fa", golden_dir + "/" + de)


def test_missing_files_and_empty_de(tmp_path, golden_dir):
    with pytest.raises(engine.EngineError, match="cannot open"):
        engine.ScaffolderGraph.from_files(str(tmp_path / "nope.fa"), golden_dir + "/libPE.de")
    with pytest.raises(engine.EngineError, match="can not read distance file"):
        engine.ScaffolderGraph.from_files(golden_dir + "/primary-contigs.fa", str(tmp_path / "nope.de"))
    empty = tmp_path / "only_unknown.de"
    empty.write_text("contig-x contig-y+,1,2,3.0 ;\n")
    with pytest.raises(engine.EngineError, match="is empty"):
        engine.ScaffolderGraph.from_files(golden_dir + "/primary-contigs.fa", str(empty))
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n")
    with pytest.raises(engine.EngineError, match="has to be '>'"):
        engine.ScaffolderGraph.from_files(str(bad), golden_dir + "/libPE.de")


def test_scaffold_records_of_a_hand_built_graph(tmp_path):
    """ref algorithms.c:901-1040 on a graph built with add_vertex / add_edge (host
    only): nothing is SCAFFOLD there, so every contig is a record of its own and
    the .scaf file holds one header per line"""
    L = engine.lib()
    g = L.gt_scaffolder_graph_new(3, 2)
    for name, ln in ((b"ctg_a", 100), (b"ctg_b", 200), (b"ctg_c", 300)):
        assert L.gt_scaffolder_graph_add_vertex(g, name, ln, 1.0, 1.0) == 0
    assert L.gt_scaffolder_graph_add_edge(g, 0, 1, 10, 1.5, 3, True, True) == 0
    assert L.gt_scaffolder_graph_add_edge(g, 1, 0, 10, 1.5, 3, False, True) == 0
    seqlen = C.POINTER(C.c_uint64)()
    recs = L.gt_scaffolder_graph_iterate_scaffolds(g, C.byref(seqlen))
    assert recs and L.gt_scaffolder_graph_records_size(recs) == 3
    assert [seqlen[i] for i in range(3)] == [100, 200, 300]
    err = C.create_string_buffer(256)
    out = tmp_path / "hand.scaf"
    assert L.gt_scaffolder_graph_write_scaffold(recs, str(out).encode(), err, 256) == 0
    assert out.read_text() == "ctg_a\nctg_b\nctg_c\n"
    L.gt_scaffolder_graph_records_delete(recs)
    L.gt_scaffolder_graph_delete(g)


def test_large_contig_file_is_read_in_parallel_pieces(tmp_path):
    """a contig file above the size from which it is read by several threads
    (gt_scaffolder_host.c, io_parallel: pieces of at least 32 MB): every contig
    is counted, the ones below min_ctg_len are not (ref parser.c:399-415)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("host table only: with a GPU the FASTA table is made there")
    n, width = 45000, 70
    long_seq = ("ACGT" * 18)[:width] + "\n"
    fa = tmp_path / "big.fa"
    with open(fa, "w") as f:
        for i in range(n):
            lines = 50 if i % 3 else 2            # 3500 or 140 bases
            f.write(">ctg%07d some description\n" % i)
            f.write(long_seq * lines)
    assert fa.stat().st_size > 3 * (32 << 20)
    L = engine.lib()
    err = C.create_string_buffer(256)
    cnt = C.c_uint64()
    assert L.gt_scaffolder_parser_count_contigs(str(fa).encode(), 200, C.byref(cnt), err, 256) == 0
    assert cnt.value == n - (n + 2) // 3
    assert L.gt_scaffolder_parser_count_contigs(str(fa).encode(), 100, C.byref(cnt), err, 256) == 0
    assert cnt.value == n


def test_counting_pass_is_reused_only_for_the_same_file(tmp_path):
    """gt_scaffolder_parser_count_contigs keeps its scan of the contig file for
    ..._read_contigs (the reference reads the file twice, parser.c:399-415 and
    :417-493); a file that changed in between is read again"""
    import time
    L = engine.lib()
    err = C.create_string_buffer(256)
    cnt = C.c_uint64()

    def write(path, n, length):
        with open(path, "w") as f:
            for i in range(n):
                f.write(">c%04d x\n%s\n" % (i, "ACGT" * (length // 4)))

    def read(path):
        g = L.gt_scaffolder_graph_new(0, 0)
        assert L.gt_scaffolder_parser_read_contigs(g, str(path).encode(), 200, False, err, 256) == 0, err.value
        n = L.gt_scaffolder_graph_nof_vertices(g)
        L.gt_scaffolder_graph_delete(g)
        return n

    fa = tmp_path / "a.fa"
    write(fa, 50, 400)
    assert L.gt_scaffolder_parser_count_contigs(str(fa).encode(), 200, C.byref(cnt), err, 256) == 0
    assert cnt.value == 50 and read(fa) == 50          # the kept scan
    assert read(fa) == 50                              # nothing kept any more: a scan of its own
    assert L.gt_scaffolder_parser_count_contigs(str(fa).encode(), 200, C.byref(cnt), err, 256) == 0
    time.sleep(0.01)
    write(fa, 70, 400)                                 # same path, other content
    assert read(fa) == 70
    # same size and path, newer: still not the kept scan
    assert L.gt_scaffolder_parser_count_contigs(str(fa).encode(), 200, C.byref(cnt), err, 256) == 0
    time.sleep(0.01)
    write(fa, 70, 400)
    with open(fa, "r+") as f:
        f.write(">X")                                  # first header renamed in place
    assert read(fa) == 70
    other = tmp_path / "b.fa"
    write(other, 30, 400)
    assert L.gt_scaffolder_parser_count_contigs(str(fa).encode(), 200, C.byref(cnt), err, 256) == 0
    assert read(other) == 30


@pytest.mark.gpu
def test_distance_records_of_the_counting_pass_are_reused_only_for_the_same_file(tmp_path, golden_dir):
    """gt_scaffolder_parser_count_distances leaves the parsed records on the
    device for ..._read_distances (the reference reads the file twice,
    parser.c:150 and :295); a file that changed in between is parsed again"""
    import shutil
    import time
    L = engine.lib()
    err = C.create_string_buffer(256)
    fa = (golden_dir + "/primary-contigs.fa").encode()
    lines = open(golden_dir + "/libPE.de").read().splitlines(True)

    def edges(de_for_count, de_for_read, change=None):
        cnt, nd = C.c_uint64(), C.c_uint64()
        assert L.gt_scaffolder_parser_count_contigs(fa, 200, C.byref(cnt), err, 256) == 0
        g = L.gt_scaffolder_graph_new(cnt.value, 0)
        assert L.gt_scaffolder_parser_read_contigs(g, fa, 200, False, err, 256) == 0
        assert L.gt_scaffolder_parser_count_distances(g, str(de_for_count).encode(), C.byref(nd), err, 256) == 0, err.value
        if change:
            change()
        assert L.gt_scaffolder_parser_read_distances(str(de_for_read).encode(), g, False, err, 256) == 0, err.value
        n = L.gt_scaffolder_graph_nof_edges(g)
        L.gt_scaffolder_graph_delete(g)
        return n

    full = tmp_path / "full.de"
    half = tmp_path / "half.de"
    full.write_text("".join(lines))
    # the file without its first pair (listed from both contigs)
    few = [ln for ln in lines if "contig-4616" not in ln]
    half.write_text("".join(few))
    n_full, n_half = edges(full, full), edges(half, half)
    assert 0 < n_half < n_full
    work = tmp_path / "work.de"
    shutil.copy(full, work)

    def shrink():
        time.sleep(0.01)
        work.write_text("".join(few))

    assert edges(work, work, shrink) == n_half     # counted on the full file, read after it shrank
    assert edges(full, half) == n_half             # another file
