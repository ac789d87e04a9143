"""ctypes binding of the CPU oracle (oracle/libgts_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATE_NAMES = ["UNVISITED", "POLYMORPHIC", "INCONSISTENT", "REPEAT", "VISITED",
               "PROCESSED", "SCAFFOLD", "CYCLIC"]


def build():
    """Compile the oracle with gcc (oracle/Makefile)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgts_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, u64, i64, f32, ci = C.c_void_p, C.c_uint64, C.c_int64, C.c_float, C.c_int
        L.ora_graph_new.restype = vp
        L.ora_graph_new.argtypes = [u64, u64]
        L.ora_graph_delete.argtypes = [vp]
        L.ora_graph_add_vertex.argtypes = [vp, C.c_char_p, u64, f32, f32]
        L.ora_graph_add_edge.argtypes = [vp, u64, u64, i64, f32, u64, C.c_bool, C.c_bool]
        L.ora_graph_add_record.argtypes = [vp, u64, u64, i64, f32, u64, C.c_bool, C.c_bool]
        L.ora_graph_add_records.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
        L.ora_graph_add_records_mp.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, C.c_bool]
        L.ora_graph_new_from_file.argtypes = [C.POINTER(vp), C.c_char_p, u64, C.c_char_p,
                                              C.c_bool, C.c_char_p, C.c_size_t]
        L.ora_graph_new_from_file.restype = ci
        L.ora_graph_new_from_file_mp.argtypes = [C.POINTER(vp), C.c_char_p, u64, C.c_char_p,
                                                 C.c_bool, C.c_bool, C.c_char_p, C.c_size_t]
        L.ora_graph_new_from_file_mp.restype = ci
        L.ora_graph_test.argtypes = [u64, u64, C.c_bool, u64, C.c_bool, u64, C.c_char_p]
        L.ora_graph_test.restype = ci
        L.ora_parser_read_distances_test.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.ora_parser_read_distances_test.restype = ci
        L.ora_mark_repeats.argtypes = [C.c_char_p, vp, f32, f32, C.c_char_p, C.c_size_t]
        L.ora_mark_repeats.restype = ci
        L.ora_mark_repeats_loaded.argtypes = [vp, C.c_bool, f32, f32]
        L.ora_filter.argtypes = [vp, f32, f32, i64]
        L.ora_removecycles.argtypes = [vp]
        L.ora_makescaffold.argtypes = [vp, ci]
        L.ora_iterate_scaffolds.argtypes = [vp]
        L.ora_iterate_scaffolds.restype = vp
        L.ora_records_delete.argtypes = [vp]
        L.ora_write_scaffold.argtypes = [vp, vp, C.c_char_p]
        L.ora_write_scaffold.restype = ci
        L.ora_graph_print.argtypes = [vp, C.c_char_p]
        L.ora_graph_print.restype = ci
        L.ora_ambiguousorder.argtypes = [i64, f32, i64, f32, f32]
        L.ora_ambiguousorder.restype = C.c_bool
        L.ora_ambiguous_from_interval.argtypes = [f32, f32]
        L.ora_ambiguous_from_interval.restype = C.c_bool
        L.ora_nv.argtypes = [vp]
        L.ora_nv.restype = u64
        L.ora_ne.argtypes = [vp]
        L.ora_ne.restype = u64
        L.ora_get_vertex_states.argtypes = [vp, vp]
        L.ora_get_edge_states.argtypes = [vp, vp]
        L.ora_get_edges.argtypes = [vp] + [vp] * 6
        L.ora_get_vertices.argtypes = [vp, vp, vp, vp]
        L.ora_vertex_header.argtypes = [vp, u64]
        L.ora_vertex_header.restype = C.c_char_p
        L.ora_set_vertex_attrs.argtypes = [vp, vp, vp]
        L.ora_records_n.argtypes = [vp]
        L.ora_records_n.restype = u64
        L.ora_records_total_edges.argtypes = [vp]
        L.ora_records_total_edges.restype = u64
        L.ora_records_flatten.argtypes = [vp, vp, vp, vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleGraph:
    """Mirror of the reference's GtScaffolderGraph life cycle on the oracle."""

    def __init__(self, handle):
        self.h = handle

    # ---- constructors -------------------------------------------------
    @classmethod
    def from_files(cls, fasta, dist, min_ctg_len=200, astat_is_annotated=False, ismatepair=False):
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.ora_graph_new_from_file_mp(C.byref(h), fasta.encode(), min_ctg_len, dist.encode(),
                                          astat_is_annotated, ismatepair, err, 512)
        if rc != 0:
            raise RuntimeError(err.value.decode())
        return cls(h)

    @classmethod
    def from_records(cls, seq_len, astat, copy_num, root, ctg, dist, std_dev, num_pairs, flags,
                     headers=None, ismatepair=False):
        L = lib()
        n = len(seq_len)
        h = C.c_void_p(L.ora_graph_new(max(n, 1), max(2 * len(root), 1)))
        g = cls(h)
        for i in range(n):
            hdr = headers[i] if headers is not None else "contig-%09d" % i
            L.ora_graph_add_vertex(h, hdr.encode(), int(seq_len[i]), float(astat[i]),
                                   float(copy_num[i]))
        root = np.ascontiguousarray(root, dtype=np.uint32)
        ctg = np.ascontiguousarray(ctg, dtype=np.uint32)
        dist = np.ascontiguousarray(dist, dtype=np.int64)
        std_dev = np.ascontiguousarray(std_dev, dtype=np.float32)
        num_pairs = np.ascontiguousarray(num_pairs, dtype=np.uint64)
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        L.ora_graph_add_records_mp(h, len(root), _p(root), _p(ctg), _p(dist), _p(std_dev),
                                   _p(num_pairs), _p(flags), bool(ismatepair))
        return g

    def __del__(self):
        try:
            if self.h:
                lib().ora_graph_delete(self.h)
                self.h = None
        except Exception:
            pass

    # ---- algorithms ---------------------------------------------------
    def mark_repeats_file(self, astat_file, copy_num_cutoff=0.3, astat_cutoff=20.0):
        err = C.create_string_buffer(512)
        rc = lib().ora_mark_repeats(astat_file.encode(), self.h, copy_num_cutoff, astat_cutoff,
                                    err, 512)
        if rc != 0:
            raise RuntimeError(err.value.decode())

    def mark_repeats(self, have_file=True, copy_num_cutoff=0.3, astat_cutoff=20.0):
        lib().ora_mark_repeats_loaded(self.h, have_file, copy_num_cutoff, astat_cutoff)

    def filter(self, pcutoff=0.01, cncutoff=1.5, ocutoff=400):
        lib().ora_filter(self.h, pcutoff, cncutoff, ocutoff)

    def removecycles(self):
        lib().ora_removecycles(self.h)

    def makescaffold(self, lazy_maps=True):
        lib().ora_makescaffold(self.h, 1 if lazy_maps else 0)

    def scaffolds(self):
        L = lib()
        r = C.c_void_p(L.ora_iterate_scaffolds(self.h))
        n = L.ora_records_n(r)
        t = L.ora_records_total_edges(r)
        roots = np.zeros(n, np.uint64)
        off = np.zeros(n + 1, np.uint64)
        edges = np.zeros(max(t, 1), np.uint64)
        seqlen = np.zeros(n, np.uint64)
        L.ora_records_flatten(r, _p(roots), _p(off), _p(edges), _p(seqlen))
        L.ora_records_delete(r)
        return roots, off, edges[:t], seqlen

    def write_scaffold(self, path):
        L = lib()
        r = C.c_void_p(L.ora_iterate_scaffolds(self.h))
        rc = L.ora_write_scaffold(self.h, r, path.encode())
        L.ora_records_delete(r)
        if rc != 0:
            raise RuntimeError("cannot write " + path)

    def print_dot(self, path):
        if lib().ora_graph_print(self.h, path.encode()) != 0:
            raise RuntimeError("cannot write " + path)

    # ---- accessors ----------------------------------------------------
    @property
    def nv(self):
        return int(lib().ora_nv(self.h))

    @property
    def ne(self):
        return int(lib().ora_ne(self.h))

    def vertex_states(self):
        out = np.zeros(self.nv, np.uint8)
        lib().ora_get_vertex_states(self.h, _p(out))
        return out

    def edge_states(self):
        out = np.zeros(max(self.ne, 1), np.uint8)
        lib().ora_get_edge_states(self.h, _p(out))
        return out[:self.ne]

    def edges(self):
        m = self.ne
        n = max(m, 1)
        start = np.zeros(n, np.uint32)
        end = np.zeros(n, np.uint32)
        dist = np.zeros(n, np.int64)
        sd = np.zeros(n, np.float32)
        npairs = np.zeros(n, np.uint64)
        flags = np.zeros(n, np.uint8)
        lib().ora_get_edges(self.h, _p(start), _p(end), _p(dist), _p(sd), _p(npairs), _p(flags))
        return dict(start=start[:m], end=end[:m], dist=dist[:m], std_dev=sd[:m],
                    num_pairs=npairs[:m], flags=flags[:m])

    def vertices(self):
        n = self.nv
        seq_len = np.zeros(n, np.uint64)
        astat = np.zeros(n, np.float32)
        cn = np.zeros(n, np.float32)
        lib().ora_get_vertices(self.h, _p(seq_len), _p(astat), _p(cn))
        return dict(seq_len=seq_len, astat=astat, copy_num=cn)

    def headers(self):
        L = lib()
        return [L.ora_vertex_header(self.h, i).decode() for i in range(self.nv)]
