"""The GPU DistEst parser (gts_deparse.hip) against the host restatement of
gt_scaffolder_parser.c (gt_scaffolder_host.c: sscanf / strtok / bsearch, line by
line as the reference) and against records written by the test itself."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from helpers import DEFAULTS, pkg

pytestmark = pytest.mark.gpu
engine = pkg.engine


def host_mode(mode):
    engine.lib().gt_scaffolder_set_distance_parser(mode)


@pytest.fixture(autouse=True)
def default_parser_afterwards():
    yield
    host_mode(0)


def graph_arrays(fa, de, mode):
    host_mode(mode)
    G = engine.ScaffolderGraph.from_files(fa, de, DEFAULTS["min_ctg_len"])
    e = G.edges()
    G.close()
    return e


def same_edges(a, b):
    assert len(a["start"]) == len(b["start"])
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.dtype.kind == "f":
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), k
        else:
            assert np.array_equal(x, y), k


def write_fasta(path, names, length=300):
    with open(path, "w") as f:
        for n in names:
            f.write(">%s\n%s\n" % (n, "A" * length))


def test_reference_files_parse_on_the_gpu(golden_dir):
    fa, de = golden_dir + "/primary-contigs.fa", golden_dir + "/libPE.de"
    same_edges(graph_arrays(fa, de, 2), graph_arrays(fa, de, 1))


@pytest.mark.parametrize("name", ["wrong_libPE_1.de", "wrong_libPE_2.de"])
def test_reference_error_files(golden_dir, name):
    """the reference's two broken distance files: same message from both parsers"""
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip("fixture not present")
    msgs = []
    for mode in (1, 2):
        host_mode(mode)
        with pytest.raises(engine.EngineError) as ei:
            engine.ScaffolderGraph.from_files(golden_dir + "/primary-contigs.fa", path, DEFAULTS["min_ctg_len"])
        msgs.append(str(ei.value))
    assert msgs[0] == msgs[1]


def random_de(rng, names, known_frac=0.97, lines=400):
    """a DistEst file in the regular form + what it says, as the test reads it"""
    out, recs = [], []
    ids = {n: i for i, n in enumerate(sorted(names))}
    pool = list(names) + ["ghost%d" % i for i in range(max(1, int(len(names) * (1 - known_frac))))]
    for _ in range(lines):
        root = rng.choice(pool)
        toks, sense = [root], True
        parts = []
        for side in range(2):
            for _ in range(rng.randrange(0, 6)):
                ctg = rng.choice(pool)
                sign = rng.choice("+-")
                dist = rng.randrange(-5000, 50000)
                npairs = rng.randrange(0, 2000)
                digits = rng.randrange(1, 8)
                sd = "%.*f" % (rng.randrange(0, 7), rng.uniform(0, 10 ** rng.randrange(0, digits)))
                parts.append((side == 0, ctg, sign, dist, npairs, sd))
                toks.append("%s%s,%d,%d,%s" % (ctg, sign, dist, npairs, sd))
            if side == 0:
                toks.append(";")
        if len(toks) == 2 and toks[1] == ";":
            pass   # "root ;" is a legal line without records
        out.append((" " * rng.randrange(0, 2)) + (" " * rng.randrange(1, 3)).join(toks) + "\n")
        if root in ids:
            for sense, ctg, sign, dist, npairs, sd in parts:
                if ctg in ids:
                    recs.append((ids[root], ids[ctg], dist, npairs, np.float32(float(sd)),
                                 (1 if sense else 0) | (2 if sign == "+" else 0)))
    return "".join(out).encode(), recs


def test_records_of_a_regular_file():
    """record by record: ids, distances, pairs, deviations, sense flipped at
    ';', records of unknown contigs and of lines with an unknown root dropped
    (the compaction), repeated blanks"""
    rng = random.Random(5)
    names = ["ctg%d" % i for i in range(300)] + ["k%d_x" % i for i in range(50)]
    text, recs = random_de(rng, names, lines=3000)
    p = engine.DeParser(sorted(names))
    res = p.parse(text)
    assert not res.irregular and res.error == 0
    assert res.n_records == len(recs) and res.n_candidates > res.n_records
    got = p.records()
    exp = np.array([(r[0], r[1], r[2], r[3], r[5]) for r in recs], dtype=np.int64)
    assert np.array_equal(got["root"], exp[:, 0]) and np.array_equal(got["ctg"], exp[:, 1])
    assert np.array_equal(got["dist"], exp[:, 2]) and np.array_equal(got["num_pairs"], exp[:, 3])
    assert np.array_equal(got["flags"], exp[:, 4])
    assert np.array_equal(got["std_dev"].view(np.uint32),
                          np.array([r[4] for r in recs], dtype=np.float32).view(np.uint32))
    # all contigs known: no compaction, same records from a device-resident text
    text, recs = random_de(rng, names, known_frac=1.0, lines=2000)
    import torch
    dev = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    res = p.parse(dev)
    # (ghost pool is never empty: one unknown name stays; just compare the counts)
    assert not res.irregular and res.error == 0 and res.n_records == len(recs)
    p.close()


@pytest.mark.parametrize("n_contigs", [3000, 100000])
def test_synthetic_files_both_parsers(tmp_path, n_contigs):
    """the file API on generated .fa / .de files: the GPU parser's graph is the
    host parser's graph, bit for bit (std_dev as sscanf reads it)"""
    from helpers import make_inputs
    g = make_inputs(n_contigs, 77)
    pkg.synth.write_files(g, str(tmp_path / "syn"))
    fa, de = str(tmp_path / "syn.fa"), str(tmp_path / "syn.de")
    same_edges(graph_arrays(fa, de, 2), graph_arrays(fa, de, 1))


IRREGULAR = {
    "exponent_without_digits": "c0 c1+,10,5,1.5e ;\n",
    "hex_float": "c0 c1+,10,5,0x1p3 ;\n",
    "tab": "c0 c1+,10,5,1.5\t;\n",
    "crlf": "c0 c1+,10,5,1.5 ;\r\n",
    "no_final_newline": "c0 c1+,10,5,1.5 ;",
    "gt_in_header": "c0 c>1+,10,5,1.5 ;\n",
    "nineteen_digits": "c0 c1+,1234567890123456789,5,1.5 ;\n",
    "blank_before_newline": "c0 c1+,10,5,1.5 \n",
    "inf": "c0 c1+,10,5,inf ;\n",
    "twenty_significant_digits": "c0 c1+,10,5,1.2345678901234567891 ;\n",
    "semicolon_token": "c0 ;c1+,10,5,1.5\n",
    # the reference's token loop starts at the root field (parser.c:338): a root that
    # looks like a separator or like a record is scanned as one there
    "root_is_a_separator": "; c1+,10,5,1.5 ;\n",
    "root_with_commas": "c0,1,2,3.5 c1+,10,5,1.5 ;\n",
    "long_line": "c0 " + " ".join("c1+,10,5,1.5" for _ in range(90)) + " ;\n",
}


@pytest.mark.parametrize("what", sorted(IRREGULAR))
def test_irregular_files_go_to_the_host_parser(tmp_path, what):
    names = ["c0", "c1", "c2"]
    fa, de = str(tmp_path / "x.fa"), str(tmp_path / "x.de")
    write_fasta(fa, names)
    with open(de, "w", newline="") as f:
        f.write("c2 c0-,7,3,2.25 ;\n" + IRREGULAR[what])
    p = engine.DeParser(sorted(names))
    assert p.parse(open(de, "rb").read()).irregular == 1
    p.close()
    outcomes = []
    for mode in (1, 0):
        host_mode(mode)
        try:
            G = engine.ScaffolderGraph.from_files(fa, de, 200)
            outcomes.append(("ok", G.edges()))
            G.close()
        except engine.EngineError as e:
            outcomes.append(("error", str(e)))
    assert outcomes[0][0] == outcomes[1][0]
    if outcomes[0][0] == "ok":
        same_edges(outcomes[0][1], outcomes[1][1])
    else:
        assert outcomes[0][1] == outcomes[1][1]
    host_mode(2)
    with pytest.raises(engine.EngineError, match="regular form"):
        engine.ScaffolderGraph.from_files(fa, de, 200)


ERRONEOUS = {
    "one_token_line": "c0\n",
    "empty_line": "\n",
    "bad_record": "c0 c1+,10,x,1.5 ;\n",
    "missing_field": "c0 c1+,10,5 ;\n",
    "negative_pairs": "c0 c1+,10,-5,1.5 ;\n",
    "no_sign": "c0 c1,10,5,1.5 ;\n",
    "unknown_root_one_token": "zz\n",
    "error_on_unknown_root_line_is_none": "zz c1+,10,x,1.5 ;\n",
}


@pytest.mark.parametrize("what", sorted(ERRONEOUS))
def test_integrity_errors_are_the_reference_messages(tmp_path, what):
    """first error in file order, the reference's message (parser.c:205-236);
    errors on a line whose root is unknown do not count"""
    names = ["c0", "c1", "c2"]
    fa, de = str(tmp_path / "x.fa"), str(tmp_path / "x.de")
    write_fasta(fa, names)
    with open(de, "w") as f:
        f.write("c2 c0-,7,3,2.25 ;\n" + ERRONEOUS[what] + "c1 c2+,5,5,5 ; c0+,1,1,1\n")
    out = []
    for mode in (1, 2):
        host_mode(mode)
        try:
            G = engine.ScaffolderGraph.from_files(fa, de, 200)
            out.append(("ok", G.ne))
            G.close()
        except engine.EngineError as e:
            out.append(("error", str(e)))
    assert out[0] == out[1], out


def test_decimal_fractions_round_like_strtof(tmp_path):
    """std_dev strings with up to 19 significant digits, tiny and huge, the
    shortest round-trip forms of floats and values at float midpoints: the GPU
    parser's float is sscanf's (or the file is handed to the host)"""
    rng = random.Random(11)
    names = ["c%d" % i for i in range(4)]
    fa, de = str(tmp_path / "x.fa"), str(tmp_path / "x.de")
    write_fasta(fa, names)
    vals = ["0", "0.0", "-0", "1", "16777217", "16777216.5", "0.1", "0.5", "123456789012345", "0.000000000000001",
            "33554433", "8388608.5", "8388609.5", "1.0000000596046448", "340282346638528", ".5", "5.", "+2.5"]
    assert len(vals) == 18
    for _ in range(500):   # exponent forms
        vals.append("%s%de%s%d" % (rng.choice(["", "0.", "1."]), rng.randrange(0, 10 ** rng.randrange(1, 9)),
                                   rng.choice(["", "+", "-"]), rng.randrange(0, 12)))
        vals.append("%.*E" % (rng.randrange(0, 9), rng.uniform(1e-9, 1e9)))
    for _ in range(1000):   # repr of a float32 (up to 17 digits), as synth.write_files writes them
        vals.append(repr(float(np.float32(rng.uniform(0.001, 10 ** rng.randrange(-2, 6))))))   # no exponent form
    for _ in range(3000):
        nd = rng.randrange(1, 20)
        digits = "".join(rng.choice("0123456789") for _ in range(nd))
        k = rng.randrange(0, nd + 1)
        v = (digits[:k] or "0") + ("." + digits[k:] if k < nd else "")
        vals.append(v)
    with open(de, "w") as f:
        for i, v in enumerate(vals):
            f.write("c%d c%d+,%d,1,%s ;\n" % (i % 4, (i + 1) % 4, i, v))
    same_edges(graph_arrays(fa, de, 0), graph_arrays(fa, de, 1))
    p = engine.DeParser(sorted(names))
    res = p.parse(open(de, "rb").read())
    p.close()
    # "1.0000000596046448" is 2e-17 above the midpoint of two floats, closer than
    # the double in between can tell: such a value is the host's, the whole file irregular
    assert res.irregular == 1
    with open(de, "w") as f:
        for i, v in enumerate(vals[18:]):
            f.write("c%d c%d+,%d,1,%s ;\n" % (i % 4, (i + 1) % 4, i, v))
    same_edges(graph_arrays(fa, de, 2), graph_arrays(fa, de, 1))


def astat_states(tmp_path, fa, de, astat, mode):
    """the .dot after mark_repeats with the A-statistic file read by `mode`"""
    host_mode(mode)
    G = engine.ScaffolderGraph.from_files(fa, de, DEFAULTS["min_ctg_len"])
    G.mark_repeats(astat, DEFAULTS["copy_num_cutoff"], DEFAULTS["astat_cutoff"])
    out = str(tmp_path / ("m%d.dot" % mode))
    G.print_dot(out)
    G.close()
    return open(out, "rb").read()


def test_reference_astat_file_on_the_gpu(tmp_path, golden_dir):
    fa, de, astat = [golden_dir + x for x in ("/primary-contigs.fa", "/libPE.de", "/libPE.astat")]
    a, b = astat_states(tmp_path, fa, de, astat, 2), astat_states(tmp_path, fa, de, astat, 1)
    assert a == b and b"ivory3" in a
    assert a == open(golden_dir + "/gt_scaffolder_algorithms_test_mark_repeats_expected.dot", "rb").read()


def test_astat_values_record_by_record():
    """contigs named in the file take copy number and A-statistic, the others
    keep what they had; exponent forms, blanks or tabs between the fields"""
    rng = random.Random(3)
    names = sorted("c%05d" % i for i in range(5000))
    lines, exp_as, exp_cn = [], {}, {}
    for nm in rng.sample(names, 3500) + ["ghost1", "ghost2"]:
        cn = rng.choice(["%.3f" % rng.uniform(0, 40), "%.6e" % rng.uniform(1e-4, 100), "%d" % rng.randrange(0, 50)])
        a = rng.choice(["%.4f" % rng.uniform(-300, 300), "%.5E" % rng.uniform(-1e3, 1e3), "-0.0"])
        sep = rng.choice(["\t", " ", "\t "])
        lines.append(sep.join([nm, str(rng.randrange(100, 10 ** 6)), str(rng.randrange(0, 10 ** 4)),
                               str(rng.randrange(0, 50)), cn, a]) + "\n")
        exp_as[nm], exp_cn[nm] = np.float32(float(a)), np.float32(float(cn))
    text = "".join(lines).encode()
    p = engine.DeParser(names)
    astat = np.full(len(names), 7.5, np.float32)
    cnum = np.full(len(names), -1.25, np.float32)
    res = p.parse_astat(text, astat, cnum)
    assert not res.irregular and res.error == 0 and res.n_records == 3500
    for i, nm in enumerate(names):
        ea, ec = exp_as.get(nm, np.float32(7.5)), exp_cn.get(nm, np.float32(-1.25))
        assert astat[i].view(np.uint32) == ea.view(np.uint32) and cnum[i].view(np.uint32) == ec.view(np.uint32), nm
    # a contig named twice: the reference lets the last line win -- the host's business
    a2, c2 = astat.copy(), cnum.copy()
    res = p.parse_astat(text + lines[0].encode(), a2, c2)
    assert res.irregular == 1 and np.array_equal(a2, astat) and np.array_equal(c2, cnum)
    # five fields: the reference's "Invalid record"
    res = p.parse_astat(b"c00001\t5\t5\t5\t1.0\n", a2, c2)
    assert res.error == 1 and not res.irregular
    # an integer field that is not one
    assert p.parse_astat(b"c00001\t5x\t5\t5\t1.0\t2.0\n", a2, c2).irregular == 1
    assert p.parse_astat(b"c00001\tx\t5\t5\t1.0\t2.0\n", a2, c2).error == 1
    p.close()


@pytest.mark.parametrize("n_contigs", [3000, 100000])
def test_synthetic_astat_files_both_parsers(tmp_path, n_contigs):
    from helpers import make_inputs
    g = make_inputs(n_contigs, 78, repeat_degree=12)
    pkg.synth.write_files(g, str(tmp_path / "syn"))
    fa, de, astat = [str(tmp_path / ("syn" + x)) for x in (".fa", ".de", ".astat")]
    a, b = astat_states(tmp_path, fa, de, astat, 2), astat_states(tmp_path, fa, de, astat, 1)
    assert a == b and b"ivory3" in a


def test_contig_headers_sorted_on_the_gpu():
    """ids are the ranks of the headers in strcmp order (ref parser.c:172):
    14 bytes on the GPU, runs that agree in them ordered by the caller"""
    rng = random.Random(9)
    names = set()
    while len(names) < 120000:
        k = rng.randrange(6)
        if k == 0:
            names.add(b"contig-%d" % rng.randrange(10 ** 7))
        elif k == 1:
            names.add(b"scaffold_with_a_long_prefix_%d" % rng.randrange(10 ** 6))     # agree in 14 bytes
        elif k == 2:
            names.add(bytes(rng.randrange(33, 256) for _ in range(rng.randrange(1, 20))).replace(b" ", b"_"))
        elif k == 3:
            names.add(b"k99_%d_%d" % (rng.randrange(1000), rng.randrange(1000)))
        elif k == 4:
            names.add(b"a" * rng.randrange(1, 30))
        else:
            names.add(b"%d" % rng.randrange(10 ** 9))
    names = list(names)
    rng.shuffle(names)
    names += names[:50]   # repeated headers: a run of ties
    perm, tie = engine.sort_names(names)
    assert sorted(perm.tolist()) == list(range(len(names)))
    got = [names[i] for i in perm]
    assert tie[0] == 0
    i = 0
    while i < len(got):
        j = i + 1
        while j < len(got) and tie[j]:
            j += 1
        assert all(x[:14] == got[i][:14] for x in got[i:j])
        got[i:j] = sorted(got[i:j])
        i = j
    assert got == sorted(names)
    assert int(tie.sum()) >= 50


def fasta_table_by_hand(text):
    """the record table as the reference's callbacks see the file (a '>' starts
    a record unless it is in a description; blanks and line ends do not count)"""
    ds, de, sl = [], [], []
    i, n = 0, len(text)
    while i < n and text[i:i + 1] == b">":
        i += 1
        ds.append(i)
        j = text.find(b"\n", i)
        j = n if j < 0 else j
        de.append(j)
        i = min(j + 1, n)
        k = text.find(b">", i)
        k = n if k < 0 else k
        seq = text[i:k]
        sl.append(len(seq) - seq.count(b"\n") - seq.count(b"\r") - seq.count(b" "))
        i = k
    return np.array(ds, np.uint64), np.array(de, np.uint64), np.array(sl, np.uint64)


def test_fasta_record_table_on_the_gpu():
    """lines of any length, '>' inside descriptions and in the middle of
    sequence lines, CRLF, blanks, records of a few bytes and of megabytes,
    a file that ends inside a description"""
    rng = random.Random(13)
    parts = []
    for r in range(3000):
        desc = b"ctg%d some text" % r + (b" with > inside" if rng.random() < 0.1 else b"")
        L = rng.choice([0, 1, 5, 60, 61, 300, 5000, 70000]) if r % 50 else 3_000_000
        seq = bytes(rng.choice(b"ACGT") for _ in range(min(L, 2000))) * (L // 2000 + 1)
        seq = seq[:L]
        w = rng.choice([60, 70, 10 ** 9])
        eol = b"\r\n" if rng.random() < 0.2 else b"\n"
        lines = [seq[k:k + w] for k in range(0, len(seq), w)]
        if rng.random() < 0.1 and lines:
            lines[0] = lines[0][:3] + b" " + lines[0][3:]
        body = eol.join(lines) + (eol if rng.random() < 0.9 else b"")   # the next '>' may follow sequence bytes directly
        parts.append(b">" + desc + eol + body)
    text = b"".join(parts)
    for t in (text, text + b">last record without newline"):
        exp = fasta_table_by_hand(t)
        got = engine.fasta_records(t)
        for a, b in zip(got, exp):
            assert np.array_equal(a, b)
    assert len(text) > 60_000_000


def test_file_api_with_the_gpu_fasta_scan(tmp_path):
    """a contig file large enough for the GPU scan (and the GPU sort of the
    headers): the graph is the one the host scan gives"""
    from helpers import make_inputs
    g = make_inputs(60000, 91)
    pkg.synth.write_files(g, str(tmp_path / "syn"))
    fa, de = str(tmp_path / "syn.fa"), str(tmp_path / "syn.de")
    assert os.path.getsize(fa) > (32 << 20)
    same_edges(graph_arrays(fa, de, 0), graph_arrays(fa, de, 1))


def test_distance_file_in_pieces(tmp_path, monkeypatch):
    """a file above the parser's 4 GB goes over in pieces that end at line ends
    and the records are collected on the device; here with pieces of 1 MB"""
    from helpers import make_inputs
    g = make_inputs(60000, 92)
    pkg.synth.write_files(g, str(tmp_path / "syn"))
    fa, de = str(tmp_path / "syn.fa"), str(tmp_path / "syn.de")
    assert os.path.getsize(de) > (8 << 20)
    whole = graph_arrays(fa, de, 2)
    monkeypatch.setenv("GTS_DE_CHUNK", str(1 << 20))
    pieces = graph_arrays(fa, de, 2)
    monkeypatch.delenv("GTS_DE_CHUNK")
    same_edges(pieces, whole)
    same_edges(pieces, graph_arrays(fa, de, 1))


def test_duplicate_headers_go_to_the_host_parser(tmp_path):
    """two contigs with one header: the host's binary search has a fixed answer,
    the GPU name table would return whichever was inserted first -- such a contig
    file sends the distance file to the host code (mode 2: an error)"""
    fa, de = str(tmp_path / "d.fa"), str(tmp_path / "d.de")
    write_fasta(fa, ["c0", "c1", "c1", "c2"])
    with open(de, "w") as f:
        f.write("c0 c1+,10,5,1.5 ; c2-,7,3,2.25\nc1 c2+,4,2,0.5 ;\n")
    same_edges(graph_arrays(fa, de, 0), graph_arrays(fa, de, 1))
    host_mode(2)
    with pytest.raises(engine.EngineError, match="duplicate contig headers"):
        engine.ScaffolderGraph.from_files(fa, de, 200)


def test_contigs_read_after_the_name_table_was_uploaded(tmp_path):
    """read_contigs, count_distances (sorts the vertices, uploads the GPU parser's
    name table), read_contigs again: the vertex set has changed, the table is
    stale and must be rebuilt -- same edges as with the host parser"""
    L = engine.lib()
    fa1, fa2, de = str(tmp_path / "a.fa"), str(tmp_path / "b.fa"), str(tmp_path / "x.de")
    write_fasta(fa1, ["c%02d" % i for i in range(0, 40, 2)])
    write_fasta(fa2, ["c%02d" % i for i in range(1, 40, 2)])
    with open(de, "w") as f:
        for i in range(0, 38):      # neighbours (one of the two is missing at first) and next-but-one
            f.write("c%02d c%02d+,%d,5,1.5 ; c%02d-,%d,4,2.5\n" % (i, i + 1, 10 + i, i + 2, 50 + i))
    out = []
    for mode in (1, 0):
        host_mode(mode)
        err = C.create_string_buffer(512)
        h = C.c_void_p(L.gt_scaffolder_graph_new(40, 0))
        nd = C.c_uint64()
        assert L.gt_scaffolder_parser_read_contigs(h, fa1.encode(), 200, False, err, 512) == 0, err.value
        assert L.gt_scaffolder_parser_count_distances(h, de.encode(), C.byref(nd), err, 512) == 0, err.value
        assert L.gt_scaffolder_parser_read_contigs(h, fa2.encode(), 200, False, err, 512) == 0, err.value
        assert L.gt_scaffolder_parser_count_distances(h, de.encode(), C.byref(nd), err, 512) == 0, err.value
        assert L.gt_scaffolder_parser_read_distances(de.encode(), h, False, err, 512) == 0, err.value
        m = int(L.gt_scaffolder_graph_nof_edges(h))
        assert m == 4 * 38
        a = {k: np.zeros(m, dt) for k, dt in (("start", np.uint32), ("end", np.uint32), ("dist", np.int64),
                                               ("std_dev", np.float32), ("num_pairs", np.int64), ("flags", np.uint8))}
        assert L.gt_scaffolder_graph_get_edges(h, *[a[k].ctypes.data_as(C.c_void_p) for k in a]) == 0
        out.append(a)
        L.gt_scaffolder_graph_delete(h)
    same_edges(out[0], out[1])
    assert list(out[0]["start"][:4]) == [0, 1, 0, 2]      # ids are the ranks among all forty headers
