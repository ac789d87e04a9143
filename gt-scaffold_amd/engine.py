"""ctypes binding of the engine's C ABI (include/gt_scaffold_hip.h).

There is no CPU path: loading fails if csrc/libgtscaffold_hip.so has not been
built, and Engine() raises if no GPU is visible.  Arrays cross the boundary as
raw pointers: numpy arrays (host) or anything with a CUDA/HIP data_ptr()
(torch tensors on the engine's device)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GTS_ENGINE_LIB: another build of the same library, for A/B measurements)
LIB_PATH = os.environ.get("GTS_ENGINE_LIB") or os.path.join(_HERE, "csrc", "libgtscaffold_hip.so")
_LIB = None

SYMBOLS = [
    "gtsg_create", "gtsg_destroy", "gtsg_last_error", "gtsg_set_contigs",
    "gtsg_build_from_records", "gtsg_build_from_records_ex", "gtsg_set_vertex_times", "gtsg_set_astat", "gtsg_mark_repeats", "gtsg_filter",
    "gtsg_removecycles", "gtsg_makescaffold", "gtsg_num_vertices", "gtsg_num_edges",
    "gtsg_get_vertex_states", "gtsg_get_edge_states", "gtsg_get_edges", "gtsg_get_csr", "gtsg_state_digest",
    "gtsg_set_option", "gtsg_selftest_ambiguous", "gtsg_filter_begin", "gtsg_filter_end",
    "gtsg_filter_get_lasthit", "gtsg_filter_set_lasthit", "gtsg_label_components",
    "gtsg_route_pack", "gtsg_route_unpack", "gtsg_get_kernel_times", "gtsg_reset_kernel_times", "gtsg_get_stat",
    "gtsg_deparser_create", "gtsg_deparser_destroy", "gtsg_deparser_last_error", "gtsg_deparser_set_names",
    "gtsg_deparser_parse", "gtsg_deparser_records", "gtsg_deparser_download", "gtsg_deparser_parse_astat",
    "gtsg_deparser_trim", "gtsg_sort_names", "gtsg_fasta_records", "gtsg_deparser_accumulate",
    "gtsg_find_edge", "gtsg_alter_edge", "gtsg_plan_weights", "gtsg_plan_deal", "gtsg_route_unpack_ex",
    "gtsg_get_scaffold_edges", "gtsg_format_dot_edges", "gtsg_format_dot_edges_pinned",
    "gtsg_scaffold_records", "gtsg_scaffold_records_fetch",
]


HOST_SYMBOLS = [
    "gt_scaffolder_graph_new", "gt_scaffolder_graph_delete", "gt_scaffolder_graph_add_vertex",
    "gt_scaffolder_graph_add_edge", "gt_scaffolder_graph_new_from_file",
    "gt_scaffolder_graph_print", "gt_scaffolder_graph_test",
    "gt_scaffolder_parser_read_distances_test", "gt_scaffolder_graph_mark_repeats",
    "gt_scaffolder_graph_filter", "gt_scaffolder_removecycles", "gt_scaffolder_makescaffold",
    "gt_scaffolder_graph_iterate_scaffolds", "gt_scaffolder_graph_records_size",
    "gt_scaffolder_graph_records_delete", "gt_scaffolder_graph_write_scaffold",
    "gt_scaffolder_graph_nof_vertices", "gt_scaffolder_graph_nof_edges",
    "gt_scaffolder_graph_last_error", "gt_scaffolder_set_device",
    "gt_scaffolder_parser_count_contigs", "gt_scaffolder_parser_read_contigs",
    "gt_scaffolder_parser_count_distances", "gt_scaffolder_parser_read_distances",
    "gt_scaffolder_set_distance_parser", "gt_scaffolder_graph_get_edges",
    "gt_scaffolder_graph_find_edge", "gt_scaffolder_graph_get_vertex_id",
    "gt_scaffolder_graph_get_vertex", "gt_scaffolder_graph_alter_edge", "gt_scaffolder_set_dot_writer",
    "gt_scaffolder_set_record_walk", "gt_scaffolder_last_record_walk",
]


class DeParseResult(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_candidates", C.c_uint64), ("error_pos", C.c_uint64),
                ("error", C.c_int), ("irregular", C.c_int)]


class RecordCounts(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_edges", C.c_uint64), ("n_open_roots", C.c_uint64),
                ("n_open_edges", C.c_uint64)]


_REC_FIELDS = ("root", "off", "seqlen", "eid", "end", "dist", "std_dev", "flags", "open_root", "open_start",
               "open_eid", "open_end", "open_dist", "open_std_dev", "open_flags")


class RecordArrays(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in _REC_FIELDS]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("calls", C.c_uint64), ("ms", C.c_double)]


def lib():
    """dlopen the engine; raises if it is missing (build with __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `python __graft_entry__.py` (hipcc, gfx950); "
                              "the engine has no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for s in SYMBOLS + HOST_SYMBOLS:
            getattr(L, s)  # AttributeError if a declared entry point is missing
        vp, u64, i64, f32, ci = C.c_void_p, C.c_uint64, C.c_int64, C.c_float, C.c_int
        L.gtsg_create.argtypes = [C.POINTER(vp), ci, vp]
        L.gtsg_destroy.argtypes = [vp]
        L.gtsg_last_error.argtypes = [vp]
        L.gtsg_last_error.restype = C.c_char_p
        L.gtsg_set_contigs.argtypes = [vp, u64, vp, vp, vp, ci]
        L.gtsg_build_from_records.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, ci]
        L.gtsg_build_from_records_ex.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, ci, ci]
        L.gtsg_set_astat.argtypes = [vp, vp, vp, ci]
        L.gtsg_set_vertex_times.argtypes = [vp, vp, ci]
        L.gtsg_mark_repeats.argtypes = [vp, ci, f32, f32]
        L.gtsg_filter.argtypes = [vp, f32, f32, i64]
        L.gtsg_filter_begin.argtypes = [vp, f32, f32, i64]
        L.gtsg_filter_end.argtypes = [vp]
        L.gtsg_filter_get_lasthit.argtypes = [vp, vp, ci]
        L.gtsg_filter_set_lasthit.argtypes = [vp, vp, ci]
        L.gtsg_label_components.argtypes = [vp, u64, u64, vp, vp, vp, vp, ci]
        L.gtsg_route_pack.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, u64, u64, vp, C.c_uint32, vp,
                                      C.POINTER(u64)]
        L.gtsg_route_unpack.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.gtsg_route_unpack_ex.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(ci)]
        L.gtsg_plan_weights.argtypes = [vp, u64, u64, vp, vp, vp, vp, vp]
        L.gtsg_plan_deal.argtypes = [vp, u64, vp, vp, vp, C.c_uint32, vp, vp]
        L.gtsg_removecycles.argtypes = [vp]
        L.gtsg_makescaffold.argtypes = [vp]
        L.gtsg_num_vertices.argtypes = [vp]
        L.gtsg_num_vertices.restype = u64
        L.gtsg_num_edges.argtypes = [vp]
        L.gtsg_num_edges.restype = u64
        L.gtsg_get_vertex_states.argtypes = [vp, vp]
        L.gtsg_get_edge_states.argtypes = [vp, vp]
        L.gtsg_get_edges.argtypes = [vp] + [vp] * 6
        L.gtsg_get_csr.argtypes = [vp, vp, vp]
        L.gtsg_find_edge.argtypes = [vp, u64, u64, C.POINTER(u64)]
        L.gtsg_alter_edge.argtypes = [vp, u64, i64, f32, u64, ci, ci]
        L.gtsg_scaffold_records.argtypes = [vp, C.POINTER(RecordCounts)]
        L.gtsg_scaffold_records_fetch.argtypes = [vp, C.POINTER(RecordArrays)]
        L.gtsg_state_digest.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.gtsg_selftest_ambiguous.argtypes = [vp, u64, vp, vp, vp, vp, f32, vp]
        L.gtsg_set_option.argtypes = [vp, C.c_char_p, i64]
        L.gtsg_get_kernel_times.argtypes = [vp, C.POINTER(KernelTime), ci]
        L.gtsg_reset_kernel_times.argtypes = [vp]
        L.gtsg_get_stat.argtypes = [vp, C.c_char_p]
        L.gtsg_get_stat.restype = i64
        cp, sz, b = C.c_char_p, C.c_size_t, C.c_bool
        L.gt_scaffolder_graph_new.argtypes = [u64, u64]
        L.gt_scaffolder_graph_new.restype = vp
        L.gt_scaffolder_graph_delete.argtypes = [vp]
        L.gt_scaffolder_graph_add_vertex.argtypes = [vp, cp, u64, f32, f32]
        L.gt_scaffolder_graph_add_edge.argtypes = [vp, u64, u64, i64, f32, u64, b, b]
        L.gt_scaffolder_graph_new_from_file.argtypes = [C.POINTER(vp), cp, u64, cp, b, cp, sz]
        L.gt_scaffolder_graph_print.argtypes = [vp, cp, cp, sz]
        L.gt_scaffolder_graph_test.argtypes = [u64, u64, b, u64, b, u64, b, cp, sz]
        L.gt_scaffolder_parser_read_distances_test.argtypes = [cp, cp, cp, sz]
        L.gt_scaffolder_graph_mark_repeats.argtypes = [cp, vp, f32, f32, cp, sz]
        L.gt_scaffolder_graph_filter.argtypes = [vp, f32, f32, i64]
        L.gt_scaffolder_removecycles.argtypes = [vp]
        L.gt_scaffolder_makescaffold.argtypes = [vp]
        L.gt_scaffolder_graph_iterate_scaffolds.argtypes = [vp, C.POINTER(C.POINTER(u64))]
        L.gt_scaffolder_graph_iterate_scaffolds.restype = vp
        L.gt_scaffolder_graph_records_size.argtypes = [vp]
        L.gt_scaffolder_graph_records_size.restype = u64
        L.gt_scaffolder_graph_records_delete.argtypes = [vp]
        L.gt_scaffolder_graph_write_scaffold.argtypes = [vp, cp, cp, sz]
        L.gt_scaffolder_graph_nof_vertices.argtypes = [vp]
        L.gt_scaffolder_graph_nof_vertices.restype = u64
        L.gt_scaffolder_graph_nof_edges.argtypes = [vp]
        L.gt_scaffolder_graph_nof_edges.restype = u64
        L.gt_scaffolder_graph_last_error.argtypes = [vp]
        L.gt_scaffolder_graph_last_error.restype = cp
        L.gt_scaffolder_set_device.argtypes = [ci]
        L.gt_scaffolder_parser_count_contigs.argtypes = [cp, u64, C.POINTER(u64), cp, sz]
        L.gt_scaffolder_parser_read_contigs.argtypes = [vp, cp, u64, b, cp, sz]
        L.gt_scaffolder_parser_count_distances.argtypes = [vp, cp, C.POINTER(u64), cp, sz]
        L.gt_scaffolder_parser_read_distances.argtypes = [cp, vp, b, cp, sz]
        L.gt_scaffolder_set_distance_parser.argtypes = [ci]
        L.gt_scaffolder_set_dot_writer.argtypes = [ci]
        L.gt_scaffolder_set_record_walk.argtypes = [ci]
        L.gt_scaffolder_last_record_walk.argtypes = []
        L.gt_scaffolder_last_record_walk.restype = ci
        L.gt_scaffolder_graph_get_edges.argtypes = [vp] * 7
        L.gt_scaffolder_graph_find_edge.argtypes = [vp, u64, u64]
        L.gt_scaffolder_graph_find_edge.restype = u64
        L.gt_scaffolder_graph_get_vertex_id.argtypes = [vp, u64]
        L.gt_scaffolder_graph_get_vertex_id.restype = u64
        L.gt_scaffolder_graph_get_vertex.argtypes = [vp, C.POINTER(u64), cp]
        L.gt_scaffolder_graph_get_vertex.restype = b
        L.gt_scaffolder_graph_alter_edge.argtypes = [vp, u64, i64, f32, u64, b, b]
        L.gtsg_deparser_create.argtypes = [C.POINTER(vp), ci, vp]
        L.gtsg_deparser_destroy.argtypes = [vp]
        L.gtsg_deparser_last_error.argtypes = [vp]
        L.gtsg_deparser_last_error.restype = cp
        L.gtsg_deparser_set_names.argtypes = [vp, vp, vp, u64]
        L.gtsg_deparser_parse.argtypes = [vp, vp, u64, ci, C.POINTER(DeParseResult)]
        L.gtsg_deparser_records.argtypes = [vp, C.POINTER(u64)] + [C.POINTER(vp)] * 6
        L.gtsg_deparser_download.argtypes = [vp] + [vp] * 6
        L.gtsg_deparser_parse_astat.argtypes = [vp, vp, u64, ci, vp, vp, ci, C.POINTER(DeParseResult)]
        L.gtsg_deparser_trim.argtypes = [vp]
        L.gtsg_deparser_accumulate.argtypes = [vp, ci]
        L.gtsg_sort_names.argtypes = [ci, vp, vp, u64, vp, vp]
        L.gtsg_fasta_records.argtypes = [ci, vp, u64, C.POINTER(u64)] + [C.POINTER(C.POINTER(u64))] * 3
        _LIB = L
    return _LIB


class EngineError(RuntimeError):
    pass


def fasta_records(text, device=0):
    """record table of a FASTA text on the GPU (gtsg_fasta_records):
    (desc_start, desc_end, seq_len) numpy arrays, one entry per record"""
    L = lib()
    n = C.c_uint64()
    ptrs = [C.POINTER(C.c_uint64)() for _ in range(3)]
    rc = L.gtsg_fasta_records(device, bytes(text), len(text), C.byref(n), *[C.byref(p) for p in ptrs])
    if rc != 0:
        raise EngineError("gtsg_fasta_records failed (code %d)" % rc)
    out = [np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64) for p in ptrs]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for p in ptrs:
        if n.value:
            libc.free(C.cast(p, C.c_void_p))
    return out


def sort_names(names, device=0):
    """strcmp order of contig headers by their first 14 bytes on the GPU
    (gtsg_sort_names): (perm, tie) -- tie[j]: name perm[j] agrees with name
    perm[j-1] in those bytes, the caller orders such runs."""
    L = lib()
    enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in enc], out=off[1:])
    blob = b"".join(enc)
    perm = np.zeros(max(len(enc), 1), np.uint32)
    tie = np.zeros(max(len(enc), 1), np.uint8)
    rc = L.gtsg_sort_names(device, blob, off.ctypes.data, len(enc), perm.ctypes.data, tie.ctypes.data)
    if rc != 0:
        raise EngineError("gtsg_sort_names failed (code %d)" % rc)
    return perm[:len(enc)], tie[:len(enc)]


class DeParser:
    """DistEst text -> records on the GPU (include/gt_scaffold_hip.h, gtsg_deparser_*).
    names: the contig headers in id order (sorted)."""

    ERRORS = {1: "Invalid record", 2: "Invalid value for number of pairs", 3: "Invalid composition sign"}

    def __init__(self, names, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        if self._L.gtsg_deparser_create(C.byref(self._h), device, None) != 0:
            self._h = None
            raise EngineError("no MI355X available: the distance parser has no CPU path")
        enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
        off = np.zeros(len(enc) + 1, dtype=np.uint64)
        np.cumsum([len(x) for x in enc], out=off[1:])
        blob = b"".join(enc)
        self._chk(self._L.gtsg_deparser_set_names(self._h, blob, off.ctypes.data, len(enc)))

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(self._L.gtsg_deparser_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.gtsg_deparser_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parse(self, text):
        """text: bytes (host) or a uint8 device tensor.  Returns the result
        structure (n_records, n_candidates, error, error_pos, irregular)."""
        res = DeParseResult()
        if isinstance(text, (bytes, bytearray)):
            self._chk(self._L.gtsg_deparser_parse(self._h, bytes(text), len(text), 0, C.byref(res)))
        else:
            import torch
            torch.cuda.current_stream().synchronize()   # the text must be complete before the call
            self._chk(self._L.gtsg_deparser_parse(self._h, text.data_ptr(), text.numel(), 1, C.byref(res)))
        return res

    def parse_astat(self, text, astat, copy_num):
        """A-statistic file: astat / copy_num (float32 numpy arrays over the
        names) are updated in place for the contigs the file names"""
        res = DeParseResult()
        assert astat.dtype == np.float32 and copy_num.dtype == np.float32
        self._chk(self._L.gtsg_deparser_parse_astat(self._h, bytes(text), len(text), 0, astat.ctypes.data,
                                                    copy_num.ctypes.data, 0, C.byref(res)))
        return res

    def records(self):
        """the records of the last parse as numpy arrays"""
        n = C.c_uint64()
        self._chk(self._L.gtsg_deparser_records(self._h, C.byref(n), None, None, None, None, None, None))
        n = n.value
        out = dict(root=np.zeros(n, np.uint32), ctg=np.zeros(n, np.uint32), dist=np.zeros(n, np.int64),
                   std_dev=np.zeros(n, np.float32), num_pairs=np.zeros(n, np.int64), flags=np.zeros(n, np.uint8))
        self._chk(self._L.gtsg_deparser_download(self._h, *[out[k].ctypes.data for k in
                                                            ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")]))
        return out


HIP_STREAM_LEGACY = 1   # hip_runtime_api.h: #define hipStreamLegacy ((hipStream_t)1)


def _ptr(a, dtype):
    """(pointer, on_device, keepalive) of a numpy array or a device tensor."""
    if a is None:
        return None, None, None
    if isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a, dtype=dtype)
        return a.ctypes.data_as(C.c_void_p), 0, a
    # torch tensor (or anything exposing data_ptr / is_cuda)
    if not a.is_contiguous():
        a = a.contiguous()
    if a.element_size() != np.dtype(dtype).itemsize:
        raise TypeError("tensor element size %d does not match %s" % (a.element_size(), dtype))
    return C.c_void_p(a.data_ptr()), (1 if a.is_cuda else 0), a


class Engine:
    """One scaffold graph resident on one GPU.  Method names follow the
    reference's API (gt_scaffolder_graph_* / gt_scaffolder_*)."""

    def __init__(self, device=0, stream=None):
        """stream: None = the engine creates its own (blocking) stream; an int =
        a hipStream_t handle, 0 being the legacy null stream (torch's default
        stream), which the C ABI spells hipStreamLegacy."""
        self._L = lib()
        h = C.c_void_p()
        self._stream = stream
        arg = None if stream is None else C.c_void_p(HIP_STREAM_LEGACY if stream == 0 else stream)
        rc = self._L.gtsg_create(C.byref(h), int(device), arg)
        if rc != 0:
            raise EngineError("gtsg_create failed (%d): no usable HIP device; the engine has no "
                              "CPU fallback" % rc)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.gtsg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError("%s (code %d)" % (self._L.gtsg_last_error(self._h).decode(), rc))

    def _sync_producer(self, ptrs):
        """Stream contract of gt_scaffold_hip.h: device inputs must be complete
        before the call.  Tensors made on torch's null stream are ordered with
        the engine's own blocking stream by HIP; if torch's current stream is
        any other stream than the engine's, it is drained here."""
        if not any(p[1] for p in ptrs if p[1] is not None):
            return
        import torch
        cur = torch.cuda.current_stream().cuda_stream
        if cur != 0 and cur != self._stream:
            torch.cuda.current_stream().synchronize()

    def _same_side(self, sides):
        sides = [s for s in sides if s is not None]
        if len(set(sides)) > 1:
            raise TypeError("mixing host and device arrays in one call")
        return sides[0] if sides else 0

    # ---- construction ----
    def set_contigs(self, seq_len, astat=None, copy_num=None):
        p0, d0, k0 = _ptr(seq_len, np.int64)
        p1, d1, k1 = _ptr(astat, np.float32)
        p2, d2, k2 = _ptr(copy_num, np.float32)
        self._sync_producer([(p0, d0), (p1, d1), (p2, d2)])
        self._chk(self._L.gtsg_set_contigs(self._h, len(seq_len), p0, p1, p2,
                                           self._same_side([d0, d1, d2])))

    def build_from_records(self, root, ctg, dist, std_dev, num_pairs, flags, ismatepair=False):
        a = [_ptr(root, np.uint32), _ptr(ctg, np.uint32), _ptr(dist, np.int64),
             _ptr(std_dev, np.float32), _ptr(num_pairs, np.int64), _ptr(flags, np.uint8)]
        self._sync_producer(a)
        self._chk(self._L.gtsg_build_from_records_ex(self._h, len(root), *[x[0] for x in a],
                                                     self._same_side([x[1] for x in a]),
                                                     int(bool(ismatepair))))

    def set_vertex_times(self, times):
        """shard of a larger graph: times[v] = id of local vertex v in the whole graph"""
        p, d, k = _ptr(times, np.uint32)
        self._sync_producer([(p, d)])
        self._chk(self._L.gtsg_set_vertex_times(self._h, p, d if d is not None else 0))

    def set_astat(self, astat, copy_num):
        p1, d1, k1 = _ptr(astat, np.float32)
        p2, d2, k2 = _ptr(copy_num, np.float32)
        self._sync_producer([(p1, d1), (p2, d2)])
        self._chk(self._L.gtsg_set_astat(self._h, p1, p2, self._same_side([d1, d2])))

    # ---- algorithms ----
    def mark_repeats(self, have_file=True, copy_num_cutoff=0.3, astat_cutoff=20.0):
        self._chk(self._L.gtsg_mark_repeats(self._h, int(have_file), copy_num_cutoff, astat_cutoff))

    def filter(self, pcutoff=0.01, cncutoff=1.5, ocutoff=400):
        self._chk(self._L.gtsg_filter(self._h, pcutoff, cncutoff, int(ocutoff)))

    def removecycles(self):
        self._chk(self._L.gtsg_removecycles(self._h))

    def makescaffold(self):
        self._chk(self._L.gtsg_makescaffold(self._h))

    # ---- multi-GPU hooks ----
    def filter_begin(self, pcutoff=0.01, cncutoff=1.5, ocutoff=400):
        self._chk(self._L.gtsg_filter_begin(self._h, pcutoff, cncutoff, int(ocutoff)))

    def filter_end(self):
        self._chk(self._L.gtsg_filter_end(self._h))

    def filter_get_lasthit(self, dst):
        p, d, k = _ptr(dst, np.int32)
        self._chk(self._L.gtsg_filter_get_lasthit(self._h, p, d))

    def filter_set_lasthit(self, src):
        p, d, k = _ptr(src, np.int32)
        self._sync_producer([(p, d)])
        self._chk(self._L.gtsg_filter_set_lasthit(self._h, p, d))

    def label_components(self, n, root, ctg, skip, labels):
        a = [_ptr(root, np.uint32), _ptr(ctg, np.uint32), _ptr(skip, np.uint8), _ptr(labels, np.uint32)]
        self._sync_producer(a)
        self._chk(self._L.gtsg_label_components(self._h, int(n), len(root), a[0][0], a[1][0], a[2][0],
                                                a[3][0], self._same_side([x[1] for x in a])))

    def plan_weights(self, root, ctg, skip8, labels):
        """records of this shard per component label (device tensors: root / ctg /
        labels int32, skip8 uint8) -> int32 [n] on the device (gtsg_plan_weights)"""
        import torch
        n = labels.numel()
        w = torch.empty(n, dtype=torch.int32, device=labels.device)
        a = [_ptr(root, np.uint32), _ptr(ctg, np.uint32), _ptr(skip8, np.uint8), _ptr(labels, np.uint32),
             _ptr(w, np.int32)]
        if not all(x[1] for x in a):
            raise TypeError("plan_weights takes device tensors")
        self._sync_producer(a)
        self._chk(self._L.gtsg_plan_weights(self._h, n, root.numel(), *[x[0] for x in a]))
        return w

    def plan_deal(self, skip8, labels, weights, world):
        """components to ranks, heaviest first in serpentine order (gtsg_plan_deal):
        returns (owner int8 [n], -1 for skipped contigs; load int64 [world])"""
        import torch
        n = labels.numel()
        owner = torch.empty(n, dtype=torch.int8, device=labels.device)
        load = torch.empty(world, dtype=torch.int64, device=labels.device)
        a = [_ptr(skip8, np.uint8), _ptr(labels, np.uint32), _ptr(weights, np.int32)]
        b = [_ptr(owner, np.int8), _ptr(load, np.int64)]
        if not all(x[1] for x in a):
            raise TypeError("plan_deal takes device tensors")
        self._sync_producer(a)
        self._chk(self._L.gtsg_plan_deal(self._h, n, a[0][0], a[1][0], a[2][0], int(world), b[0][0], b[1][0]))
        return owner, load

    def route_pack(self, rec, first_index, owner8, world):
        """rec: device tensors root / ctg (int32), dist, num_pairs (int64), std_dev
        (float32), flags (uint8); owner8: int8 per contig (negative = shared).
        Returns (rows [n, 4] int64 grouped by destination, counts per rank)."""
        import torch
        n = rec["root"].numel()
        rows = torch.empty((n, 4), dtype=torch.int64, device=rec["root"].device)
        counts = (C.c_uint64 * world)()
        a = [_ptr(rec[k], dt) for k, dt in (("root", np.uint32), ("ctg", np.uint32), ("dist", np.int64),
                                            ("std_dev", np.float32), ("num_pairs", np.int64),
                                            ("flags", np.uint8))]
        po, do, ko = _ptr(owner8, np.int8)
        pr, dr, kr = _ptr(rows, np.int64)
        self._sync_producer(a + [(po, do), (pr, dr)])
        self._chk(self._L.gtsg_route_pack(self._h, n, *[x[0] for x in a], int(first_index),
                                          owner8.numel(), po, int(world), pr, counts))
        return rows, [int(c) for c in counts]

    def route_unpack(self, rows, loc_of=None):
        """rows [n, 4] int64 (device) -> dict of device tensors; loc_of: int32 per
        contig, whole-graph id -> local number (or None)."""
        import torch
        n, dev = rows.shape[0], rows.device
        out = dict(root=torch.empty(n, dtype=torch.int32, device=dev), ctg=torch.empty(n, dtype=torch.int32, device=dev),
                   dist=torch.empty(n, dtype=torch.int64, device=dev),
                   std_dev=torch.empty(n, dtype=torch.float32, device=dev),
                   num_pairs=torch.empty(n, dtype=torch.int64, device=dev),
                   flags=torch.empty(n, dtype=torch.uint8, device=dev),
                   k=torch.empty(n, dtype=torch.int64, device=dev))
        pr, dr, kr = _ptr(rows, np.int64)
        pl = _ptr(loc_of, np.uint32)[0] if loc_of is not None else None
        self._sync_producer([(pr, dr)])
        ooo = C.c_int(0)
        self._chk(self._L.gtsg_route_unpack_ex(self._h, n, pr, pl, *[C.c_void_p(out[k].data_ptr()) for k in
                                                                      ("root", "ctg", "dist", "std_dev",
                                                                       "num_pairs", "flags", "k")], C.byref(ooo)))
        out["out_of_order"] = bool(ooo.value)
        return out

    # ---- results ----
    @property
    def nv(self):
        return int(self._L.gtsg_num_vertices(self._h))

    @property
    def ne(self):
        return int(self._L.gtsg_num_edges(self._h))

    def vertex_states(self):
        out = np.zeros(max(self.nv, 1), np.uint8)
        self._chk(self._L.gtsg_get_vertex_states(self._h, out.ctypes.data_as(C.c_void_p)))
        return out[:self.nv]

    def edge_states(self):
        out = np.zeros(max(self.ne, 1), np.uint8)
        self._chk(self._L.gtsg_get_edge_states(self._h, out.ctypes.data_as(C.c_void_p)))
        return out[:self.ne]

    def edges(self):
        m = self.ne
        n = max(m, 1)
        o = dict(start=np.zeros(n, np.uint32), end=np.zeros(n, np.uint32),
                 dist=np.zeros(n, np.int64), std_dev=np.zeros(n, np.float32),
                 num_pairs=np.zeros(n, np.int64), flags=np.zeros(n, np.uint8))
        self._chk(self._L.gtsg_get_edges(self._h, *[o[k].ctypes.data_as(C.c_void_p) for k in
                                                    ("start", "end", "dist", "std_dev",
                                                     "num_pairs", "flags")]))
        return {k: v[:m] for k, v in o.items()}

    def find_edge(self, vertex_1, vertex_2):
        """id of the first edge of vertex_1's list that ends in vertex_2, or None
        (ref gt_scaffolder_graph.c:174-193)"""
        eid = C.c_uint64()
        self._chk(self._L.gtsg_find_edge(self._h, int(vertex_1), int(vertex_2), C.byref(eid)))
        return None if eid.value == 2 ** 64 - 1 else eid.value

    def alter_edge(self, eid, dist, std_dev, num_pairs, sense, same):
        """ref gt_scaffolder_graph.c:219-235"""
        self._chk(self._L.gtsg_alter_edge(self._h, int(eid), int(dist), float(std_dev), int(num_pairs),
                                          int(bool(sense)), int(bool(same))))

    def scaffold_records(self):
        """gtsg_scaffold_records (ref gt_scaffolder_algorithms.c:901-997): the records of the clean
        SCAFFOLD paths ranked on the device -- root, off [n+1], seqlen per record; eid, end, dist,
        std_dev, flags per edge -- and the open part left to the caller: open_root, and per open
        edge open_start / _eid / _end / _dist / _std_dev / _flags."""
        c = RecordCounts()
        self._chk(self._L.gtsg_scaffold_records(self._h, C.byref(c)))
        nr, ne, pr, pe = c.n_records, c.n_edges, c.n_open_roots, c.n_open_edges
        size = dict(root=nr, off=nr, seqlen=nr, open_root=pr)
        dt = dict(seqlen=np.uint64, dist=np.int64, open_dist=np.int64, std_dev=np.float32,
                  open_std_dev=np.float32, flags=np.uint8, open_flags=np.uint8)
        o = {}
        for k in _REC_FIELDS:
            o[k] = np.zeros(size.get(k, pe if k.startswith("open_") else ne) + 1, dt.get(k, np.uint32))
        a = RecordArrays(**{k: o[k].ctypes.data for k in _REC_FIELDS})
        self._chk(self._L.gtsg_scaffold_records_fetch(self._h, C.byref(a)))
        o["off"][nr] = ne
        out = {k: v[:-1] for k, v in o.items() if k != "off"}
        out["off"] = o["off"]
        return out

    def digest(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self._L.gtsg_state_digest(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def selftest_ambiguous(self, d1, s1, d2, s2, pcutoff):
        d1 = np.ascontiguousarray(d1, np.int64); d2 = np.ascontiguousarray(d2, np.int64)
        s1 = np.ascontiguousarray(s1, np.float32); s2 = np.ascontiguousarray(s2, np.float32)
        out = np.zeros(len(d1), np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self._L.gtsg_selftest_ambiguous(self._h, len(d1), p(d1), p(s1), p(d2), p(s2),
                                                  pcutoff, p(out)))
        return out

    # ---- tuning / measurement ----
    def set_option(self, name, value):
        self._chk(self._L.gtsg_set_option(self._h, name.encode(), int(value)))

    def stat(self, name):
        return int(self._L.gtsg_get_stat(self._h, name.encode()))

    def kernel_times(self):
        n = self._L.gtsg_get_kernel_times(self._h, None, 0)
        buf = (KernelTime * max(n, 1))()
        n = self._L.gtsg_get_kernel_times(self._h, buf, n)
        return {buf[i].name.decode(): (int(buf[i].calls), float(buf[i].ms)) for i in range(n)}

    def reset_kernel_times(self):
        self._L.gtsg_reset_kernel_times(self._h)


class ScaffolderGraph:
    """The reference's GtScaffolderGraph life cycle through the C host layer
    (include/gt_scaffolder_host.h): files in, .dot / .scaf out."""

    def __init__(self, handle):
        self._L = lib()
        self._h = handle

    @classmethod
    def from_files(cls, fasta, dist, min_ctg_len=200, astat_is_annotated=False, device=0):
        L = lib()
        L.gt_scaffolder_set_device(device)
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.gt_scaffolder_graph_new_from_file(C.byref(h), fasta.encode(), min_ctg_len,
                                                 dist.encode(), astat_is_annotated, err, 512)
        if rc != 0:
            raise EngineError(err.value.decode())
        return cls(h)

    @classmethod
    def from_files_stepwise(cls, fasta, dist, min_ctg_len=200, astat_is_annotated=False,
                            ismatepair=False, device=0):
        """The reference's parser.h entry points one by one (what
        gt_scaffolder_graph_new_from_file does, ref graph.c:346-419), with the
        `ismatepair` switch of read_distances exposed.  Returns the graph and
        (nof_contigs, nof_distances) as counted."""
        L = lib()
        L.gt_scaffolder_set_device(device)
        err = C.create_string_buffer(512)
        nc, nd = C.c_uint64(), C.c_uint64()
        if L.gt_scaffolder_parser_count_contigs(fasta.encode(), min_ctg_len, C.byref(nc), err, 512):
            raise EngineError(err.value.decode())
        h = C.c_void_p(L.gt_scaffolder_graph_new(nc.value, 0))
        g = cls(h)
        if L.gt_scaffolder_parser_read_contigs(h, fasta.encode(), min_ctg_len, astat_is_annotated, err, 512) or \
                L.gt_scaffolder_parser_count_distances(h, dist.encode(), C.byref(nd), err, 512) or \
                L.gt_scaffolder_parser_read_distances(dist.encode(), h, ismatepair, err, 512):
            raise EngineError(err.value.decode())
        return g, (nc.value, nd.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.gt_scaffolder_graph_delete(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, err=None):
        if rc != 0:
            msg = err.value.decode() if err is not None and err.value else \
                self._L.gt_scaffolder_graph_last_error(self._h).decode()
            raise EngineError(msg)

    @property
    def nv(self):
        return int(self._L.gt_scaffolder_graph_nof_vertices(self._h))

    @property
    def ne(self):
        return int(self._L.gt_scaffolder_graph_nof_edges(self._h))

    def edges(self):
        m = self.ne
        n = max(m, 1)
        o = dict(start=np.zeros(n, np.uint32), end=np.zeros(n, np.uint32),
                 dist=np.zeros(n, np.int64), std_dev=np.zeros(n, np.float32),
                 num_pairs=np.zeros(n, np.int64), flags=np.zeros(n, np.uint8))
        self._chk(self._L.gt_scaffolder_graph_get_edges(self._h, *[o[k].ctypes.data for k in
                                                                    ("start", "end", "dist", "std_dev",
                                                                     "num_pairs", "flags")]))
        return {k: v[:m] for k, v in o.items()}

    def mark_repeats(self, astat_file, copy_num_cutoff=0.3, astat_cutoff=20.0):
        err = C.create_string_buffer(512)
        self._chk(self._L.gt_scaffolder_graph_mark_repeats(astat_file.encode(), self._h,
                                                           copy_num_cutoff, astat_cutoff, err, 512), err)

    def filter(self, pcutoff=0.01, cncutoff=1.5, ocutoff=400):
        self._chk(self._L.gt_scaffolder_graph_filter(self._h, pcutoff, cncutoff, int(ocutoff)))

    def removecycles(self):
        self._chk(self._L.gt_scaffolder_removecycles(self._h))

    def makescaffold(self):
        self._chk(self._L.gt_scaffolder_makescaffold(self._h))

    def print_dot(self, path):
        err = C.create_string_buffer(512)
        self._chk(self._L.gt_scaffolder_graph_print(self._h, path.encode(), err, 512), err)

    def write_scaffold(self, path):
        """iterate_scaffolds + write_scaffold; returns the scaffold lengths."""
        sl = C.POINTER(C.c_uint64)()
        r = C.c_void_p(self._L.gt_scaffolder_graph_iterate_scaffolds(self._h, C.byref(sl)))
        if not r:
            raise EngineError(self._L.gt_scaffolder_graph_last_error(self._h).decode())
        n = int(self._L.gt_scaffolder_graph_records_size(r))
        lens = np.array([sl[i] for i in range(n)], np.uint64)
        C.CDLL(None).free(sl)
        err = C.create_string_buffer(512)
        rc = self._L.gt_scaffolder_graph_write_scaffold(r, path.encode(), err, 512)
        self._L.gt_scaffolder_graph_records_delete(r)
        self._chk(rc, err)
        return lens


def state_digest_host(vstates, estates):
    """Host restatement of gtsg_state_digest (for comparing with oracle states)."""
    def mix(x):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
        return x

    with np.errstate(over="ignore"):
        def dig(s):
            ids = np.arange(len(s), dtype=np.uint64)
            return int(mix((ids << np.uint64(8)) | s.astype(np.uint64)).sum(dtype=np.uint64))
        return dig(vstates), dig(estates)
