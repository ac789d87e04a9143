"""Shared helpers of the test-suite (test infrastructure)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402
from oracle.oracle_py import OracleGraph  # noqa: E402

pkg = load_package()
synth = pkg.synth

DEFAULTS = dict(min_ctg_len=200, copy_num_cutoff=0.3, astat_cutoff=20.0, pcutoff=0.01,
                cncutoff=1.5, ocutoff=400)  # ref src/test.c:35-42


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_inputs(n, seed, **kw):
    return synth.to_numpy(synth.make_graph(n, seed=seed, device="cpu", **kw))


def oracle_from_inputs(g):
    return OracleGraph.from_records(g["seq_len"], g["astat"], g["copy_num"], g["root"], g["ctg"],
                                    g["dist"], g["std_dev"], g["num_pairs"], g["flags"])


def csr_from_oracle(og):
    """CSR arrays in the engine's layout from the oracle's edge list."""
    e = og.edges()
    v = og.vertices()
    n, m = og.nv, og.ne
    order = np.argsort(e["start"], kind="stable").astype(np.uint32)   # pos -> eid
    pos_of = np.empty(max(m, 1), np.uint32)
    pos_of[order] = np.arange(m, dtype=np.uint32)
    row = np.zeros(n + 1, np.uint32)
    np.cumsum(np.bincount(e["start"], minlength=n), out=row[1:])
    twin = pos_of[order ^ 1] if m else np.zeros(1, np.uint32)
    return dict(n=n, m=m, row=row, eid=order, pos_of=pos_of[:m],
                end=np.ascontiguousarray(e["end"][order]),
                dist=np.ascontiguousarray(e["dist"][order]),
                sd=np.ascontiguousarray(e["std_dev"][order]),
                flags=np.ascontiguousarray(e["flags"][order]),
                twin=np.ascontiguousarray(twin.astype(np.uint32)),
                seq_len=v["seq_len"].astype(np.int64), astat=v["astat"], copy_num=v["copy_num"])


_HS = None


def hostsim():
    """tests/hostsim/libhostsim.so: the engine's algorithm bodies compiled for the host."""
    global _HS
    if _HS is None:
        d = os.path.join(ROOT, "tests", "hostsim")
        subprocess.run(["make", "-s", "-C", d], check=True)
        L = C.CDLL(os.path.join(d, "libhostsim.so"))
        vp = C.c_void_p
        L.hs_amb_thresholds.argtypes = [C.c_float, vp, vp]
        L.hs_ambiguous.argtypes = [C.c_int64, C.c_float, C.c_int64, C.c_float, C.c_float, C.c_float]
        L.hs_ambiguous.restype = C.c_int
        L.hs_mark_repeats.argtypes = [C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float]
        L.hs_filter.argtypes = [C.c_uint32, C.c_uint32] + [vp] * 12 + [C.c_float, C.c_float, C.c_int64]
        L.hs_filter.restype = C.c_uint32
        L.hs_components.argtypes = [C.c_uint32, C.c_uint32] + [vp] * 8 + [C.c_int, C.c_uint32,
                                                                         C.c_uint64, vp, vp, C.c_int, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32]
        L.hs_components.restype = C.c_uint32
        _HS = L
    return _HS


class HostSimGraph:
    """Drives the engine's algorithm bodies on the host over a CSR graph."""

    def __init__(self, csr):
        self.g = csr
        self.vstate = np.zeros(max(csr["n"], 1), np.uint8)
        self.state = np.zeros(max(csr["m"], 1), np.uint8)
        self.rounds = None
        self.ncomp = self.maxcomp = 0

    def mark_repeats(self, have_file=True, copy_num_cutoff=0.3, astat_cutoff=20.0):
        g = self.g
        hostsim().hs_mark_repeats(g["n"], _p(g["row"]), _p(g["end"]), _p(g["astat"]),
                                  _p(g["copy_num"]), _p(self.vstate), _p(self.state),
                                  int(have_file), copy_num_cutoff, astat_cutoff)

    def filter(self, pcutoff=0.01, cncutoff=1.5, ocutoff=400):
        g = self.g
        r = hostsim().hs_filter(g["n"], g["m"], _p(g["row"]), _p(g["seq_len"]), _p(g["astat"]),
                                _p(g["copy_num"]), _p(self.vstate), _p(g["end"]), _p(g["dist"]),
                                _p(g["sd"]), _p(g["flags"]), _p(self.state), _p(g["twin"]),
                                _p(g["eid"]), pcutoff, cncutoff, ocutoff)
        self.rounds = (r >> 16, r & 0xFFFF)

    def _components(self, mode, wq_factor=8, max_pops=1 << 40, fast_walks=1, defer_min_nv=0,
                    defer_ref_min_nv=0):
        g = self.g
        nc = C.c_uint32()
        mc = C.c_uint32()
        nf = C.c_uint64()
        ns = C.c_uint64()
        ncl = C.c_uint64()
        ndf = C.c_uint64()
        nrd = C.c_uint64()
        nerr = hostsim().hs_components(g["n"], g["m"], _p(g["row"]), _p(g["seq_len"]),
                                       _p(self.vstate), _p(g["end"]), _p(g["dist"]), _p(g["flags"]),
                                       _p(self.state), _p(g["twin"]), mode, wq_factor, max_pops,
                                       C.byref(nc), C.byref(mc), fast_walks, C.byref(nf),
                                       C.byref(ns), C.byref(ncl), defer_min_nv, C.byref(ndf), C.byref(nrd),
                                       defer_ref_min_nv)
        self.ncomp, self.maxcomp = nc.value, mc.value
        self.fast_walks, self.slow_walks, self.clean_components = nf.value, ns.value, ncl.value
        self.deferred_components = ndf.value
        self.walk_task_rounds = nrd.value
        return nerr

    def removecycles(self, **kw):
        return self._components(0, **kw)

    def makescaffold(self, **kw):
        return self._components(1, **kw)

    def vertex_states(self):
        return self.vstate[:self.g["n"]].copy()

    def edge_states(self):
        """in the reference's edge-id order"""
        return self.state[:self.g["m"]][self.g["pos_of"]].copy() if self.g["m"] else self.state[:0]
