/*
  hostsim.cpp -- TEST INFRASTRUCTURE.  Compiles the engine's per-vertex and
  per-component algorithm bodies (gt-scaffold_amd/csrc/gts_filter.hpp,
  gts_component.hpp) for the host with g++ and drives them serially, so that
  the `-m "not gpu"` tests can check the data-parallel reformulation against
  the oracle without a GPU.  Not part of the product: the product path runs
  these bodies inside gfx950 kernels only.
*/
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "gts_defs.h"
#include "gts_amb_host.h"
#include "gts_filter.hpp"
#include "gts_component.hpp"

extern "C" {

void hs_amb_thresholds(float pcutoff, float *tpos, float *tneg)
{
  GtsAmbThresholds t = gts_amb_thresholds(pcutoff);
  *tpos = t.tpos; *tneg = t.tneg;
}

int hs_ambiguous(int64_t d1, float s1, int64_t d2, float s2, float tpos, float tneg)
{
  GtsAmbThresholds t = {tpos, tneg};
  return gts_ambiguous(d1, s1, d2, s2, t);
}

/* mark_repeats on a CSR graph: ref algorithms.c:155-167 restated as
   "an edge is REPEAT iff its start or its end is a repeat vertex" */
void hs_mark_repeats(uint32_t n, const uint32_t *row, const uint32_t *end,
                     const float *astat, const float *cn, uint8_t *vstate,
                     uint8_t *state, int have_file, float cncut, float acut)
{
  for (uint32_t v = 0; v < n; v++)
    if (gts_is_repeat(astat[v], cn[v], have_file, cncut, acut))
      vstate[v] = GIS_REPEAT;
  for (uint32_t v = 0; v < n; v++)
    for (uint32_t p = row[v]; p < row[v + 1]; p++)
      if (gts_is_repeat(astat[v], cn[v], have_file, cncut, acut) ||
          gts_is_repeat(astat[end[p]], cn[end[p]], have_file, cncut, acut))
        state[p] = GIS_REPEAT;
}

/* returns the number of rounds used by the two fixpoints (P << 16 | I) */
uint32_t hs_filter(uint32_t n, uint32_t m, const uint32_t *row,
                   const int64_t *seq_len, const float *astat, const float *cn,
                   uint8_t *vstate, const uint32_t *end, const int64_t *dist,
                   const float *sd, const uint8_t *flags, uint8_t *state,
                   const uint32_t *twin, const uint32_t *eid, float pcutoff,
                   float cncutoff, int64_t ocutoff)
{
  GtsGraphView G = {n, m, row, seq_len, astat, cn, vstate, end, dist, sd, flags,
                    state, twin, eid};
  GtsFilterParams P;
  P.amb = gts_amb_thresholds(pcutoff);
  P.cncutoff = cncutoff;
  P.ocutoff = ocutoff;
  const bool zero_ovf = 0 > ocutoff;
  std::vector<uint8_t> prop(m ? m : 1, 0), vinfo(n ? n : 1, 0), ovf(n ? n : 1, 0);
  std::vector<uint32_t> tpoly(n ? n : 1, GTS_NONE), lasthit(2 * (size_t)n + 2, GTS_NONE);
  uint32_t rounds_p = 0, rounds_i = 0;
  for (uint32_t v = 0; v < n; v++) {
    if (gts_vertex_is_marked(vstate[v])) { vinfo[v] = GTS_VI_INACTIVE; continue; }
    vinfo[v] = (uint8_t)gts_filter_pairs(G, P, v, 0, 1, prop.data());
  }
  for (;;) {
    bool pending = false;
    std::vector<uint8_t> next(vinfo);   /* Jacobi rounds, as on the device */
    for (uint32_t v = 0; v < n; v++) {
      if (vinfo[v] & (GTS_VI_ACTIVE0 | GTS_VI_INACTIVE)) continue;
      uint32_t r = gts_filter_active_round(G, v, prop.data(), vinfo.data());
      if (r) next[v] |= (uint8_t)r; else pending = true;
    }
    vinfo.swap(next);
    rounds_p++;
    if (!pending) break;
  }
  for (uint32_t v = 0; v < n; v++)
    if (!gts_vertex_is_marked(vstate[v]))
      tpoly[v] = gts_filter_tpoly(G, v, prop.data(), vinfo.data());
  for (uint32_t v = 0; v < n; v++) {
    if (!(vinfo[v] & GTS_VI_ACTIVE0) || tpoly[v] == v) { ovf[v] = GTS_OV_KNOWN; continue; }
    uint32_t o = GTS_OV_ACTIVE1;
    if (zero_ovf) o |= GTS_OV0_A | GTS_OV0_S;
    else if (vinfo[v] & (GTS_VI_OVALL_A | GTS_VI_OVALL_S))
      o |= gts_filter_ovf0(G, P, v, 0, 1, tpoly.data());
    if (!zero_ovf && !(o & (GTS_OV0_A | GTS_OV0_S))) o |= GTS_OV_KNOWN;
    ovf[v] = (uint8_t)o;
  }
  for (;;) {
    bool pending = false;
    std::vector<uint8_t> next(ovf);
    for (uint32_t v = 0; v < n; v++) {
      if (ovf[v] & GTS_OV_KNOWN) continue;
      uint32_t r = gts_filter_hit_round(G, v, ovf.data(), zero_ovf);
      if (r) next[v] = (uint8_t)r; else pending = true;
    }
    ovf.swap(next);
    rounds_i++;
    if (!pending) break;
  }
  for (uint32_t a = 0; a < n; a++)
    gts_filter_lasthit(G, a, ovf.data(), &lasthit[2 * (size_t)a]);
  std::vector<uint8_t> fin(m ? m : 1);
  for (uint32_t a = 0; a < n; a++)
    for (uint32_t p = row[a]; p < row[a + 1]; p++)
      fin[p] = gts_filter_final_edge(G, a, p, tpoly.data(), ovf.data(), lasthit.data());
  memcpy(state, fin.data(), m);
  for (uint32_t v = 0; v < n; v++)
    if (tpoly[v] != GTS_NONE) vstate[v] = GIS_POLYMORPHIC;
  return (rounds_p << 16) | rounds_i;
}


/* removecycles / makescaffold through the per-component program.  The host
   harness prepares what the device pipeline prepares with kernels: weak
   components over live edges, slots sorted by (component, vertex), compact CSR
   of live edges, scratch.  Returns the number of components with an error. */
uint32_t hs_components(uint32_t n, uint32_t m, const uint32_t *row,
                       const int64_t *seq_len, uint8_t *vstate,
                       const uint32_t *end, const int64_t *dist,
                       const uint8_t *flags, uint8_t *state,
                       const uint32_t *twin, int mode, uint32_t wq_factor,
                       uint64_t max_pops, uint32_t *out_ncomp,
                       uint32_t *out_maxcomp, int fast_walks,
                       uint64_t *out_fast, uint64_t *out_slow, uint64_t *out_clean,
                       uint32_t defer_min_nv, uint64_t *out_deferred, uint64_t *out_rounds,
                       uint32_t defer_ref_min_nv)
{
  GtsGraphView G = {n, m, row, seq_len, nullptr, nullptr, vstate, end, dist,
                    nullptr, flags, state, twin, nullptr};
  std::vector<uint32_t> start_of(m ? m : 1);
  for (uint32_t v = 0; v < n; v++)
    for (uint32_t p = row[v]; p < row[v + 1]; p++) start_of[p] = v;
  /* live edges, touched vertices, union-find with min-root */
  std::vector<uint32_t> parent(n ? n : 1);
  std::vector<uint8_t> touched(n ? n : 1, 0);
  for (uint32_t v = 0; v < n; v++) parent[v] = v;
  auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
  auto live = [&](uint32_t p) {
    return !gts_edge_is_marked(state[p]) && !gts_vertex_is_marked(vstate[start_of[p]]) &&
           !gts_vertex_is_marked(vstate[end[p]]);
  };
  for (uint32_t p = 0; p < m; p++) {
    if (!live(p)) continue;
    uint32_t a = find(start_of[p]), b = find(end[p]);
    touched[start_of[p]] = 1; touched[end[p]] = 1;
    if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }
  }
  /* trivial vertices: unmarked, no live edge in or out */
  for (uint32_t v = 0; v < n; v++) {
    if (gts_vertex_is_marked(vstate[v]) || touched[v]) continue;
    vstate[v] = mode == GTS_MODE_MAKESCAFFOLD ? GIS_SCAFFOLD : GIS_UNVISITED;
  }
  /* slots sorted by (label, vertex): labels are min vertex ids, so a stable
     counting pass over vertices in index order per label does it */
  std::vector<uint32_t> label(n ? n : 1, GTS_NONE), slot_of(n ? n : 1, GTS_NONE);
  std::vector<uint32_t> cnt(n + 1, 0);
  uint32_t nslots = 0;
  for (uint32_t v = 0; v < n; v++)
    if (touched[v] && !gts_vertex_is_marked(vstate[v])) { label[v] = find(v); cnt[label[v]]++; nslots++; }
  std::vector<uint32_t> comp_off, lab_first(n + 1, GTS_NONE);
  uint32_t acc = 0;
  for (uint32_t l = 0; l < n; l++)
    if (cnt[l]) { lab_first[l] = acc; comp_off.push_back(acc); acc += cnt[l]; }
  comp_off.push_back(acc);
  const uint32_t ncomp = (uint32_t)comp_off.size() - 1;
  std::vector<uint32_t> slot_v(nslots ? nslots : 1), fill(n + 1, 0);
  for (uint32_t v = 0; v < n; v++)
    if (label[v] != GTS_NONE) { uint32_t s = lab_first[label[v]] + fill[label[v]]++; slot_v[s] = v; slot_of[v] = s; }
  /* compact CSR: an edge is kept if it is live or its twin is (marking a walk
     edge's twin SCAFFOLD, algorithms.c:842-845, makes a marked twin usable
     again for later walks) */
  auto incl = [&](uint32_t p) { return live(p) || live(twin[p]); };
  std::vector<uint32_t> coff(nslots + 1, 0), cmap(m ? m : 1, GTS_NONE);
  for (uint32_t s = 0; s < nslots; s++) {
    uint32_t v = slot_v[s], k = 0;
    for (uint32_t p = row[v]; p < row[v + 1]; p++) if (incl(p)) k++;
    coff[s + 1] = coff[s] + k;
  }
  const uint32_t nce = coff[nslots];
  std::vector<uint32_t> cstart(nce ? nce : 1), cend(nce ? nce : 1), cgpos(nce ? nce : 1);
  std::vector<int64_t> cdist(nce ? nce : 1), cseq(nslots ? nslots : 1);
  std::vector<uint8_t> cflags(nce ? nce : 1), cstate(nce ? nce : 1), vst(nslots ? nslots : 1);
  for (uint32_t s = 0; s < nslots; s++) {
    uint32_t v = slot_v[s], k = coff[s];
    cseq[s] = seq_len[v]; vst[s] = vstate[v];
    for (uint32_t p = row[v]; p < row[v + 1]; p++)
      if (incl(p)) {
        cstart[k] = s - lab_first[label[v]]; cend[k] = slot_of[end[p]] - lab_first[label[v]];
        cdist[k] = dist[p];
        const bool uturn = ((flags[twin[p]] & GTS_F_SENSE) != 0) == gts_next_dir(flags[p]);
        cflags[k] = (uint8_t)((flags[p] & 3u) | (uturn ? GTS_F_UTURN : 0u) |
                              (live(twin[p]) ? GTS_F_TWINLIVE : 0u));
        cstate[k] = state[p]; cgpos[k] = p; cmap[p] = k; k++;
      }
  }
  uint32_t maxcomp = 0;
  for (uint32_t c = 0; c < ncomp; c++)
    if (comp_off[c + 1] - comp_off[c] > maxcomp) maxcomp = comp_off[c + 1] - comp_off[c];
  const uint64_t wq_pool = 2 * ((uint64_t)wq_factor * nce + 64ull * ncomp) + 64;  /* rings are powers of two */
  unsigned long long wq_used = 0;
  const size_t S = nslots ? nslots : 1;
  std::vector<uint32_t> queue(S), term(S), visited(S), st_v(S), st_par(S), st_cur(S),
      edgemap(S), lastpop(S, 0), wterm(S), touchedl(S), cc_best(S), ccoff(S + ncomp + 1),
      wq_edge(wq_pool), cerr(ncomp ? ncomp : 1, 0);
  std::vector<uint8_t> st_dir(S);
  std::vector<float> distmap(S, GTS_DIST_UNSET);
  std::vector<int64_t> wq_dist(wq_pool);
  GtsCompView C;
  C.G = G; C.cmap = cmap.data(); C.ncomp = ncomp; C.comp_off = comp_off.data();
  C.slot_v = slot_v.data(); C.cseq = cseq.data(); C.coff = coff.data();
  C.cstart = cstart.data(); C.cend = cend.data(); C.cdist = cdist.data();
  C.cflags = cflags.data(); C.cgpos = cgpos.data(); C.cstate = cstate.data();
  C.vst = vst.data(); C.queue = queue.data(); C.term = term.data();
  C.visited = visited.data(); C.st_v = st_v.data(); C.st_par = st_par.data();
  C.st_cur = st_cur.data(); C.edgemap = edgemap.data(); std::vector<uint32_t> par_v(S); C.par = par_v.data(); C.lastpop = lastpop.data();
  C.wterm = wterm.data(); C.touched = touchedl.data(); C.cc_best = cc_best.data();
  C.st_dir = st_dir.data(); C.distmap = distmap.data(); C.ccoff = ccoff.data();
  C.wq_edge = wq_edge.data(); C.wq_dist = wq_dist.data(); C.wq_used = &wq_used;
  C.wq_pool = wq_pool; C.wq_factor = wq_factor;
  C.cerr = cerr.data(); C.max_pops = max_pops;
  std::vector<int64_t> nd(S); std::vector<uint64_t> plen(S); std::vector<uint8_t> tight(S, 0);
  std::vector<uint32_t> sf(ncomp ? ncomp : 1, 0), ss(ncomp ? ncomp : 1, 0);
  C.fast_walks = fast_walks; C.batch_walks = 0; C.small_masks = 0; C.timing_skip_writeback = 0; C.local_marks = 0; C.help_walks = 0; C.comp_d32 = nullptr; C.team_slab = nullptr; C.team_used = nullptr; C.team_cap = 0; C.team_stat = nullptr; C.small_stat = nullptr; C.tspan = nullptr; C.nd = nd.data(); C.plen = plen.data(); C.tight = tight.data();
  std::vector<uint64_t> tstat(5 * (size_t)(ncomp ? ncomp : 1), 0);
  unsigned long long why[8] = {0};
  std::vector<uint8_t> gorient(S); std::vector<uint32_t> topo(S), tpos(S), sclean(ncomp ? ncomp : 1, 0);
  C.gorient = gorient.data(); C.topo = topo.data(); C.tpos = tpos.data(); C.stat_clean = sclean.data();
  std::vector<uint32_t> sncc(ncomp ? ncomp : 1, 0);
  C.stat_ncc = sncc.data();
  C.stat_fast = sf.data(); C.stat_slow = ss.data(); C.tstat = tstat.data(); C.why = why;
  /* deferred walks (as the engine's k_walk_tasks / k_select_walks) */
  std::vector<uint8_t> defer_flag(ncomp ? ncomp : 1, 0), task_skip(S, 0);
  std::vector<uint32_t> comp_task0(ncomp ? ncomp : 1, 0), comp_ncc(ncomp ? ncomp : 1, 0),
      comp_nterm(ncomp ? ncomp : 1, 0), task_comp(S), task_start(S), task_n(S);
  std::vector<uint64_t> task_len(S), task_poff(S);
  uint64_t path_cap = 1;
  for (uint32_t c = 0; c < ncomp; c++) {
    const uint64_t k = comp_off[c + 1] - comp_off[c];
    path_cap += k * (k + (k + 31) / 32);
  }
  std::vector<uint32_t> paths(path_cap);
  unsigned long long ntasks = 0, path_used = 0;
  C.defer_min_nv = defer_min_nv; C.defer_min_work = 0; C.defer_unclean_work = 0; C.defer_flag = defer_flag.data();
  C.defer_ref_min_nv = defer_ref_min_nv; C.task_reference = defer_ref_min_nv != 0;
  C.comp_task0 = comp_task0.data(); C.comp_ncc = comp_ncc.data(); C.comp_nterm = comp_nterm.data();
  unsigned long long task_bytes = 0;
  C.task_bytes = &task_bytes;
  C.ntasks = &ntasks; C.path_used = &path_used; C.task_cap = S; C.path_cap = path_cap;
  C.task_comp = task_comp.data(); C.task_start = task_start.data(); C.task_n = task_n.data();
  C.task_skip = task_skip.data(); C.task_len = task_len.data(); C.task_poff = task_poff.data();
  C.paths = paths.data();
  std::vector<uint32_t> comp_next_cc(ncomp ? ncomp : 1, 0), wbits(S / 32 + ncomp + 2, 0);
  C.comp_next_cc = comp_next_cc.data(); C.wbits = wbits.data();
  /* one class in the harness */
  std::vector<uint64_t> task_roff(S), comp_ring(2 * (size_t)(ncomp ? ncomp : 1), 0);
  std::vector<uint8_t> comp_klass(ncomp ? ncomp : 1, 0);
  std::vector<uint32_t> tq(S + 1), defer_list(ncomp ? ncomp : 1);
  uint32_t tq_base[1] = {0};
  unsigned long long tq_cnt[1] = {0}, ndeferred = 0;
  C.task_roff = task_roff.data(); C.comp_ring = comp_ring.data(); C.comp_klass = comp_klass.data();
  C.tq = tq.data(); C.tq_base = tq_base; C.tq_cnt = tq_cnt; C.defer_list = defer_list.data();
  C.ndeferred = &ndeferred;
  uint32_t nerr = 0;
  for (uint32_t c = 0; c < ncomp; c++) {
    GtsCompMem mem = GtsComponent<GtsWave1>::global_mem(C, c);
    GtsComponent<GtsWave1> prog(C, mem, c);
    prog.run(mode);
  }
  uint64_t ndefer = ndeferred, rounds = 0;
  for (;; rounds++) {
    if (!tq_cnt[0]) break;
    for (uint64_t q = 0; q < tq_cnt[0]; q++) {
      const uint32_t t = tq[q], c = task_comp[t];
      GtsCompMem mem = GtsComponent<GtsWave1>::global_mem(C, c);
      GtsComponent<GtsWave1> prog(C, mem, c);
      /* the harness runs the tasks of a component one after the other: they
         share the component's ring (on the device every task carves its own) */
      prog.qbase = comp_ring[2 * (size_t)c]; prog.qcap = comp_ring[2 * (size_t)c + 1];
      prog.walk_task(t);
      comp_ring[2 * (size_t)c] = prog.qbase; comp_ring[2 * (size_t)c + 1] = prog.qcap;
    }
    tq_cnt[0] = 0;
    for (uint64_t d = 0; d < ndeferred; d++) {
      const uint32_t c = defer_list[d];
      if (defer_flag[c]) GtsComponent<GtsWave1>::select_walks(C, c, wbits.data() + comp_off[c] / 32 + c);
    }
  }
  for (uint32_t c = 0; c < ncomp; c++) if (cerr[c]) nerr++;
  if (out_rounds) *out_rounds = rounds;
  if (out_deferred) *out_deferred = ndefer;
  uint64_t tf = 0, ts = 0;
  for (uint32_t c = 0; c < ncomp; c++) { tf += sf[c]; ts += ss[c]; }
  uint64_t ncl = 0;
  for (uint32_t c = 0; c < ncomp; c++) ncl += sclean[c] & 1u;
  if (out_clean) *out_clean = ncl;
  if (out_fast) *out_fast = tf;
  if (out_slow) *out_slow = ts;
  if (out_ncomp) *out_ncomp = ncomp;
  if (out_maxcomp) *out_maxcomp = maxcomp;
  return nerr;
}

} /* extern "C" */
