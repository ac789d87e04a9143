#!/usr/bin/env python3
"""Writer of the hand-derived fixtures in this directory (test data).

Every state below was derived BY HAND from the reference's source
(/root/reference/src/gt_scaffolder_algorithms.c, gt_scaffolder_parser.c); the
derivations are in README.md.  This script only formats them: it calls neither
the oracle nor the engine.  State letters: U unvisited (black), P polymorphic
(gray80), I inconsistent (gainsboro), R repeat (ivory3), V visited (red),
S scaffold (magenta), C cyclic (blue) -- the colours of ref
gt_scaffolder_graph.c:269-307.
"""
import os

os.chdir(os.path.dirname(os.path.abspath(__file__)))
COL = dict(U="black", P="gray80", I="gainsboro", R="ivory3", V="red", G="green", S="magenta", C="blue")


def fasta(name, contigs):
    with open(name + ".fa", "w") as f:
        for c, L in contigs:
            f.write(">%s %d 0\n" % (c, L))
            s = ("ACGT" * (L // 4 + 1))[:L]
            for o in range(0, L, 60):
                f.write(s[o:o + 60] + "\n")


def astat(name, rows):
    with open(name + ".astat", "w") as f:
        for c, L, cn, a in rows:
            f.write("%s\t%d\t0\t0\t%f\t%f\n" % (c, L, cn, a))


def dot(path, vertices, vcol, edges, ecol):
    with open(path, "w") as f:
        f.write("digraph {\n")
        for i, (v, c) in enumerate(zip(vertices, vcol)):
            f.write('%d [color="%s" label="%s"];\n' % (i, COL[c], v))
        for (s, e, d, sense), c in zip(edges, ecol):
            f.write('%d -> %d [color="%s" label="%d" arrowhead="%s"];\n'
                    % (vertices.index(s), vertices.index(e), COL[c], d, "normal" if sense else "inv"))
        f.write("}\n")


def stages(name, vertices, edges, table):
    for stage, (vc, ec) in table.items():
        assert len(vc) == len(vertices) and len(ec) == len(edges), (name, stage)
        dot("%s_%s_expected.dot" % (name, stage), vertices, vc, edges, ec)


# ---- 1: polymorphic pair ------------------------------------------------
V = ["ctgA", "ctgB", "ctgC"]
fasta("polymorphic", [("ctgC", 300), ("ctgA", 300), ("ctgB", 300)])
open("polymorphic.de", "w").write("ctgA ctgB+,100,10,5.0 ctgC+,110,10,5.0 ;\nctgB ;\nctgC ;\n")
astat("polymorphic", [("ctgA", 300, 1.0, 50.0), ("ctgB", 300, 0.4, 50.0), ("ctgC", 300, 0.5, 50.0)])
E = [("ctgA", "ctgB", 100, 1), ("ctgB", "ctgA", 100, 0), ("ctgA", "ctgC", 110, 1), ("ctgC", "ctgA", 110, 0)]
stages("polymorphic", V, E, dict(mark_repeats=("UUU", "UUUU"), filter=("UPU", "PPUU"),
                                 removecycles=("UPU", "PPUU"), makescaffold=("SPS", "PPSS")))

# ---- 2: inconsistent overlap, twin direction, last writer wins ----------
V = ["A", "B", "C", "D", "E", "F", "G"]
fasta("inconsistent", [(v, 500) for v in ["D", "A", "G", "C", "B", "F", "E"]])
open("inconsistent.de", "w").write(
    "A B+,100,10,5.0 C+,150,10,5.0 ;\n"
    "B E+,50,10,5.0 ; D+,60,10,5.0\n"
    "D F+,70,10,5.0 ;\n"
    "C ; G+,80,10,5.0\n"
    "E ;\nF ;\nG ;\n")
astat("inconsistent", [("A", 500, 1.0, 50.0), ("B", 500, 0.4, 50.0), ("C", 500, 1.2, 50.0),
                       ("D", 500, 1.0, 50.0), ("E", 500, 1.0, 50.0), ("F", 500, 0.5, 50.0),
                       ("G", 500, 1.0, 50.0)])
E = [("A", "B", 100, 1), ("B", "A", 100, 0), ("A", "C", 150, 1), ("C", "A", 150, 0),
     ("B", "E", 50, 1), ("E", "B", 50, 0), ("B", "D", 60, 0), ("D", "B", 60, 1),
     ("D", "F", 70, 1), ("F", "D", 70, 0), ("C", "G", 80, 0), ("G", "C", 80, 1)]
stages("inconsistent", V, E, dict(
    mark_repeats=("UUUUUUU", "UUUUUUUUUUUU"),
    filter=("UPUUUUU", "PPIIPPPPUUIU"),
    removecycles=("UPUUUUU", "PPIIPPPPUUIU"),
    makescaffold=("SPSSSSV", "PPIIPPPPSSIU")))

# ---- 3: directed 3-cycle behind a terminal ------------------------------
V = ["T", "X", "Y", "Z"]
fasta("cycle", [(v, 300) for v in ["Z", "Y", "X", "T"]])
open("cycle.de", "w").write("T X+,10,10,5.0 ;\nX Y+,20,10,5.0 ;\nY Z+,30,10,5.0 ;\nZ X+,40,10,5.0 ;\n")
astat("cycle", [(v, 300, 1.0, 50.0) for v in V])
E = [("T", "X", 10, 1), ("X", "T", 10, 0), ("X", "Y", 20, 1), ("Y", "X", 20, 0),
     ("Y", "Z", 30, 1), ("Z", "Y", 30, 0), ("Z", "X", 40, 1), ("X", "Z", 40, 0)]
stages("cycle", V, E, dict(mark_repeats=("UUUU", "UUUUUUUU"), filter=("UUUU", "UUUUUUUU"),
                           removecycles=("UCUC", "CCCCCCCC"), makescaffold=("SCSC", "CCCCCCCC")))

# ---- 4: two walks of equal length ---------------------------------------
V = ["M", "S", "T1", "T2"]
fasta("equal_walks", [(v, 300) for v in ["S", "T2", "M", "T1"]])
open("equal_walks.de", "w").write("S M+,10,10,5.0 ;\nM T1+,20,10,5.0 T2+,500,10,5.0 ;\nT1 ;\nT2 ;\n")
astat("equal_walks", [(v, 300, 1.0, 50.0) for v in V])
E = [("S", "M", 10, 1), ("M", "S", 10, 0), ("M", "T1", 20, 1), ("T1", "M", 20, 0),
     ("M", "T2", 500, 1), ("T2", "M", 500, 0)]
stages("equal_walks", V, E, dict(mark_repeats=("UUUU", "UUUUUU"), filter=("UUUU", "UUUUUU"),
                                 removecycles=("UUUU", "UUUUUU"), makescaffold=("SSVS", "SSUUSS")))

# ---- 5 / 6: diamond, equal labels (first setter keeps the edge map) and a
#      label that is improved later (the vertex is queued again) ------------
V = ["A", "B", "S", "T"]
for name, dbt, vfin, efin in (("diamond_tie", 5, "SVSS", "SSUUSSUU"),
                              ("diamond_improve", 4, "VSSS", "UUSSUUSS")):
    fasta(name, [(v, 300) for v in ["T", "S", "B", "A"]])
    open(name + ".de", "w").write("S A+,10,10,5.0 B+,10,10,5.0 ;\nA T+,5,10,5.0 ;\nB T+,%d,10,5.0 ;\nT ;\n" % dbt)
    astat(name, [(v, 300, 1.0, 50.0) for v in V])
    E = [("S", "A", 10, 1), ("A", "S", 10, 0), ("S", "B", 10, 1), ("B", "S", 10, 0),
         ("A", "T", 5, 1), ("T", "A", 5, 0), ("B", "T", dbt, 1), ("T", "B", dbt, 0)]
    stages(name, V, E, dict(mark_repeats=("UUUU", "UUUUUUUU"), filter=("UUUU", "UUUUUUUU"),
                            removecycles=("UUUU", "UUUUUUUU"), makescaffold=(vfin, efin)))

# ---- 7: a polymorphic edge overwritten by a later inconsistency mark ------
V = ["A", "B", "C", "D", "E", "F"]
fasta("overwrite_polymorphic", [(v, 500) for v in ["F", "E", "D", "C", "B", "A"]])
open("overwrite_polymorphic.de", "w").write(
    "A B+,100,10,5.0 C+,110,10,5.0 ;\n"
    "B D+,30,10,5.0 ;\n"
    "F D+,100,10,5.0 E+,150,10,5.0 ;\n"
    "C ;\nD ;\nE ;\n")
astat("overwrite_polymorphic", [("A", 500, 1.0, 50.0), ("B", 500, 0.4, 50.0), ("C", 500, 0.5, 50.0),
                                ("D", 500, 1.0, 50.0), ("E", 500, 1.0, 50.0), ("F", 500, 1.0, 50.0)])
E = [("A", "B", 100, 1), ("B", "A", 100, 0), ("A", "C", 110, 1), ("C", "A", 110, 0),
     ("B", "D", 30, 1), ("D", "B", 30, 0), ("F", "D", 100, 1), ("D", "F", 100, 0),
     ("F", "E", 150, 1), ("E", "F", 150, 0)]
stages("overwrite_polymorphic", V, E, dict(
    mark_repeats=("UUUUUU", "UUUUUUUUUU"),
    filter=("UPUUUU", "PPUUPIIIII"),
    removecycles=("UPUUUU", "PPUUPIIIII"),
    makescaffold=("SPSSSS", "PPSSPIIIII")))
