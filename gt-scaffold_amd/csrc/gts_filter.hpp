/*
  gts_filter.hpp -- data-parallel restatement of gt_scaffolder_graph_filter
  (ref src/gt_scaffolder_algorithms.c:261-343) and of mark_repeats' marking
  loop (ref algorithms.c:155-167).

  The reference walks the vertices in index order and lets every vertex see
  the marks its predecessors left (polymorphic vertices are skipped, marked
  edges drop out of the overlap test, later writes overwrite earlier edge
  states).  Here every write gets the logical time of the reference's loop:

     time 2u    : vertex u's polymorphic pass   (algorithms.c:283-295)
     time 2u+1  : vertex u's inconsistency pass (algorithms.c:301-341)

  and the result is rebuilt from three facts that hold for that loop:
   (P) "u proposes w" (a same-sense pair of u's edges is ambiguous, the two
       ends' copy numbers sum below the cut-off and w has the smaller one)
       depends on static data only.  w turns polymorphic at the first ACTIVE
       proposer; u is active iff it was not pre-marked and no active proposer
       of u has a smaller index.  This is a lexicographically-first fixpoint
       over the proposal arcs, solved in rounds (every round settles at least
       the smallest unsettled vertex).
   (I) When u (active, not polymorphic by time 2u+1) finds an overlap above
       the cut-off in direction d, it marks all its d-edges and, on the end
       vertex of each, ALL edges in the twin direction (algorithms.c:249-258).
       So if an earlier neighbour hit (v, d'), every d'-edge of v is already
       marked when v runs and v's maximum overlap in d' is 0; otherwise v sees
       only pre-marked edges and edges whose end turned polymorphic earlier.
       Again a lexicographically-first fixpoint, on (vertex, direction).
   (F) The final state of an edge is the state written last: POLYMORPHIC at
       the time either end vertex turned polymorphic, INCONSISTENT at 2a+1 if
       its start a overflowed in its direction, or at 2y+1 for the latest
       neighbour y of a whose overflow hit that direction of a.

  All functions take (lane, nlanes): the caller gives one lane (thread per
  vertex) or a wavefront's 64 lanes (hub vertices) and reduces the partial
  results itself.
*/
#ifndef GTS_FILTER_HPP
#define GTS_FILTER_HPP

#include "gts_defs.h"

/* vertex flags kept in vinfo[] during the filter */
#define GTS_VI_ACTIVE0 0x01u  /* runs its polymorphic pass */
#define GTS_VI_INACTIVE 0x02u /* pre-marked, or polymorphic before its turn */
#define GTS_VI_OVALL_A 0x04u  /* some antisense pair overlaps > cutoff (no marks) */
#define GTS_VI_OVALL_S 0x08u  /* some sense pair overlaps > cutoff (no marks) */
/* overflow flags kept in ovf[] */
#define GTS_OV_A 0x01u        /* overflow in antisense direction (d = 0) */
#define GTS_OV_S 0x02u        /* overflow in sense direction (d = 1) */
#define GTS_OV_KNOWN 0x04u
#define GTS_OV_ACTIVE1 0x08u  /* runs its inconsistency pass */
#define GTS_OV0_A 0x10u       /* overflow if not hit, antisense */
#define GTS_OV0_S 0x20u       /* overflow if not hit, sense */

/* ref algorithms.c:184-187 with the erf pipeline folded into thresholds.
   Must be compiled with -ffp-contract=off (the reference is built without
   FMA contraction: src/Makefile:7, plain x86-64 gcc). */
GTS_HD bool gts_ambiguous(int64_t d1, float s1, int64_t d2, float s2,
                          GtsAmbThresholds t)
{
  float expval = (float)(d1 - d2);
  float variance = 2.0f * ((s1 * s1) + (s2 * s2));
  float interval = (float)((double)(0.0f - expval) / sqrt((double)variance));
  return interval >= 0.0f ? interval <= t.tpos : -interval <= t.tneg;
}

/* ref algorithms.c:197-220; seq_len enters as in the reference's
   GtWord + GtUword - 1 arithmetic (two's complement) */
GTS_HD int64_t gts_overlap(int64_t dist1, int64_t len1, int64_t dist2,
                           int64_t len2)
{
  int64_t end1 = (int64_t)((uint64_t)dist1 + (uint64_t)len1 - 1u);
  int64_t end2 = (int64_t)((uint64_t)dist2 + (uint64_t)len2 - 1u);
  if (dist2 <= end1 && dist1 <= end2) {
    int64_t is = dist1 > dist2 ? dist1 : dist2;
    int64_t ie = end1 < end2 ? end1 : end2;
    return ie - is + 1;
  }
  return 0;
}

/* mark_repeats, ref algorithms.c:160-166: the vertex test */
GTS_HD bool gts_is_repeat(float astat, float copy_num, bool have_file,
                          float copy_num_cutoff, float astat_cutoff)
{
  return astat <= astat_cutoff || (have_file && copy_num < copy_num_cutoff);
}

/* Edge attributes as the pair passes see them.  GtsEdgeAccGlobal reads the CSR
   arrays (and gathers the end vertex' attributes per access); the kernels
   stage a block's edges in LDS once and use their own accessor. */
struct GtsEdgeAccGlobal {
  const GtsGraphView &G;
  GTS_HD GtsEdgeAccGlobal(const GtsGraphView &g) : G(g) {}
  GTS_HD uint8_t sense(uint32_t p) const { return G.flags[p] & GTS_F_SENSE; }
  GTS_HD int64_t dist(uint32_t p) const { return G.dist[p]; }
  GTS_HD float sd(uint32_t p) const { return G.sd[p]; }
  GTS_HD float cn(uint32_t p) const { return G.copy_num[G.end[p]]; }
  GTS_HD int64_t len(uint32_t p) const { return G.seq_len[G.end[p]]; }
};

template <class Acc>
GTS_HD uint32_t gts_filter_pairs_acc(const Acc &A, const GtsFilterParams &P,
                                     uint32_t b, uint32_t e, uint32_t lane,
                                     uint32_t nlanes, uint8_t *prop, uint32_t prop_base)
{
  uint32_t ovall = 0;
  for (uint32_t i = b + lane; i < e; i += nlanes) {
    const uint8_t fi = A.sense(i);
    const int64_t di = A.dist(i);
    const float si = A.sd(i);
    const float cni = A.cn(i);
    const int64_t li = A.len(i);
    for (uint32_t j = i + 1; j < e; ++j) {
      if (A.sense(j) != fi) continue;
      const int64_t dj = A.dist(j);
      const float cnj = A.cn(j);
      if ((cni + cnj) < P.cncutoff && gts_ambiguous(di, si, dj, A.sd(j), P.amb))
        prop[prop_base + (cni < cnj ? i : j)] = 1;
      if (gts_overlap(di, li, dj, A.len(j)) > P.ocutoff)
        ovall |= fi ? GTS_VI_OVALL_S : GTS_VI_OVALL_A;
    }
  }
  return ovall;
}

/* (I) static part over an accessor that also knows which edges are marked */
template <class Acc>
GTS_HD uint32_t gts_filter_ovf0_acc(const Acc &A, const GtsFilterParams &P,
                                    uint32_t b, uint32_t e, uint32_t lane,
                                    uint32_t nlanes)
{
  uint32_t bits = 0;
  for (uint32_t i = b + lane; i < e; i += nlanes) {
    if (A.marked(i)) continue;
    const uint8_t fi = A.sense(i);
    const int64_t di = A.dist(i), li = A.len(i);
    for (uint32_t j = i + 1; j < e; ++j) {
      if (A.sense(j) != fi || A.marked(j)) continue;
      if (gts_overlap(di, li, A.dist(j), A.len(j)) > P.ocutoff)
        bits |= fi ? GTS_OV0_S : GTS_OV0_A;
    }
  }
  return bits;
}

/* (P) static part: proposal flags prop[p] = "start(p) proposes end(p)" and
   the mark-free overflow pre-test.  Returns the GTS_VI_OVALL_* bits seen by
   this lane.  Only called for vertices that are not pre-marked. */
GTS_HD uint32_t gts_filter_pairs(const GtsGraphView &G, const GtsFilterParams &P,
                                 uint32_t v, uint32_t lane, uint32_t nlanes,
                                 uint8_t *prop)
{
  const uint32_t b = G.row[v], e = G.row[v + 1];
  uint32_t ovall = 0;
  for (uint32_t i = b + lane; i < e; i += nlanes) {
    const uint8_t fi = G.flags[i] & GTS_F_SENSE;
    const int64_t di = G.dist[i];
    const float si = G.sd[i];
    const uint32_t xi = G.end[i];
    const float cni = G.copy_num[xi];
    const int64_t li = G.seq_len[xi];
    for (uint32_t j = i + 1; j < e; ++j) {
      if ((G.flags[j] & GTS_F_SENSE) != fi) continue;
      const uint32_t xj = G.end[j];
      const int64_t dj = G.dist[j];
      const float cnj = G.copy_num[xj];
      if ((cni + cnj) < P.cncutoff &&
          gts_ambiguous(di, si, dj, G.sd[j], P.amb))
        prop[cni < cnj ? i : j] = 1;
      if (gts_overlap(di, li, dj, G.seq_len[xj]) > P.ocutoff)
        ovall |= fi ? GTS_VI_OVALL_S : GTS_VI_OVALL_A;
    }
  }
  return ovall;
}

/* (P) one round for vertex v.  Returns the new ACTIVE0/INACTIVE bits (0 =
   still unsettled).  vinfo[] is read racily: a stale "unsettled" only defers
   v to the next round. */
GTS_HD uint32_t gts_filter_active_round(const GtsGraphView &G, uint32_t v,
                                        const uint8_t *prop,
                                        const uint8_t *vinfo)
{
  const uint32_t b = G.row[v], e = G.row[v + 1];
  bool pending = false;
  for (uint32_t p = b; p < e; ++p) {
    const uint32_t u = G.end[p];
    if (u >= v || !prop[G.twin[p]]) continue;
    const uint32_t su = vinfo[u] & (GTS_VI_ACTIVE0 | GTS_VI_INACTIVE);
    if (su & GTS_VI_ACTIVE0) return GTS_VI_INACTIVE;
    if (su == 0) pending = true;
  }
  return pending ? 0u : GTS_VI_ACTIVE0;
}

/* (P) first active proposer of v (GTS_NONE if none); v is not pre-marked */
GTS_HD uint32_t gts_filter_tpoly(const GtsGraphView &G, uint32_t v,
                                 const uint8_t *prop, const uint8_t *vinfo)
{
  const uint32_t b = G.row[v], e = G.row[v + 1];
  uint32_t t = GTS_NONE;
  for (uint32_t p = b; p < e; ++p) {
    const uint32_t u = G.end[p];
    if (prop[G.twin[p]] && (vinfo[u] & GTS_VI_ACTIVE0) && u < t) t = u;
  }
  return t;
}

/* (I) static part for an ACTIVE1 vertex: overflow per direction given only the
   marks that do not depend on other vertices' overflow (pre-marked edges and
   ends that turned polymorphic at a time < 2v+1).  Returns GTS_OV0_* bits. */
GTS_HD uint32_t gts_filter_ovf0(const GtsGraphView &G, const GtsFilterParams &P,
                                uint32_t v, uint32_t lane, uint32_t nlanes,
                                const uint32_t *tpoly)
{
  const uint32_t b = G.row[v], e = G.row[v + 1];
  uint32_t bits = 0;
  for (uint32_t i = b + lane; i < e; i += nlanes) {
    const uint32_t xi = G.end[i];
    if (gts_edge_is_marked(G.state[i]) || tpoly[xi] <= v) continue;
    const uint8_t fi = G.flags[i] & GTS_F_SENSE;
    const int64_t di = G.dist[i], li = G.seq_len[xi];
    for (uint32_t j = i + 1; j < e; ++j) {
      if ((G.flags[j] & GTS_F_SENSE) != fi) continue;
      const uint32_t xj = G.end[j];
      if (gts_edge_is_marked(G.state[j]) || tpoly[xj] <= v) continue;
      if (gts_overlap(di, li, G.dist[j], G.seq_len[xj]) > P.ocutoff)
        bits |= fi ? GTS_OV0_S : GTS_OV0_A;
    }
  }
  return bits;
}

/* (I) one round for vertex v (ACTIVE1, not KNOWN).  Returns the new ovf byte
   or 0 if v still waits for a smaller neighbour. */
GTS_HD uint32_t gts_filter_hit_round(const GtsGraphView &G, uint32_t v,
                                     const uint8_t *ovf, bool zero_ovf)
{
  const uint32_t b = G.row[v], e = G.row[v + 1];
  const uint32_t mine = ovf[v];
  uint32_t hit = 0;
  for (uint32_t p = b; p < e; ++p) {
    const uint32_t y = G.end[p];
    if (y >= v) continue;
    const uint32_t oy = ovf[y];
    if (!(oy & GTS_OV_ACTIVE1)) continue;
    const uint8_t ff = G.flags[G.twin[p]];      /* the edge y -> v */
    const bool fs = ff & GTS_F_SENSE;
    if (!(oy & GTS_OV_KNOWN)) {
      /* y can only overflow in direction fs if its static test says so */
      if (!zero_ovf && !(oy & (fs ? GTS_OV0_S : GTS_OV0_A))) continue;
      return 0;
    }
    if (oy & (fs ? GTS_OV_S : GTS_OV_A))
      hit |= gts_twin_dir(ff) ? GTS_OV_S : GTS_OV_A;
  }
  uint32_t r = GTS_OV_KNOWN | GTS_OV_ACTIVE1 | (mine & (GTS_OV0_A | GTS_OV0_S));
  if ((hit & GTS_OV_A) ? zero_ovf : (mine & GTS_OV0_A) != 0) r |= GTS_OV_A;
  if ((hit & GTS_OV_S) ? zero_ovf : (mine & GTS_OV0_S) != 0) r |= GTS_OV_S;
  return r;
}

/* (F) latest neighbour whose overflow marks direction d of vertex a;
   out[0] antisense, out[1] sense */
GTS_HD void gts_filter_lasthit(const GtsGraphView &G, uint32_t a,
                               const uint8_t *ovf, uint32_t out[2])
{
  const uint32_t b = G.row[a], e = G.row[a + 1];
  int64_t h0 = -1, h1 = -1;
  for (uint32_t q = b; q < e; ++q) {
    const uint32_t y = G.end[q];
    const uint32_t oy = ovf[y];
    if (!(oy & GTS_OV_ACTIVE1)) continue;
    const uint8_t ff = G.flags[G.twin[q]];
    if (!(oy & ((ff & GTS_F_SENSE) ? GTS_OV_S : GTS_OV_A))) continue;
    if (gts_twin_dir(ff)) { if ((int64_t)y > h1) h1 = y; }
    else { if ((int64_t)y > h0) h0 = y; }
  }
  out[0] = h0 < 0 ? GTS_NONE : (uint32_t)h0;
  out[1] = h1 < 0 ? GTS_NONE : (uint32_t)h1;
}

/* (F) final state of edge position p of vertex a */
/* with_end = false leaves out what the END vertex contributes (its time stamp
   of the polymorphic pass): the engine adds it in a second pass over the few
   edges that end in a polymorphic vertex */
/* vtime (may be null = identity): the time stamp of a vertex when the engine
   holds a SHARD of a larger graph with order-preserving local vertex numbers --
   the vertex' number in the whole graph.  lasthit[] then holds such numbers
   too (the shards combine the table), and only here are times of this shard
   compared with times that may come from another one. */
GTS_HD uint8_t gts_filter_final_edge(const GtsGraphView &G, uint32_t a,
                                     uint32_t p, const uint32_t *tpoly,
                                     const uint8_t *ovf, const uint32_t *lasthit,
                                     bool with_end = true, const uint32_t *vtime = nullptr)
{
  const bool s = G.flags[p] & GTS_F_SENSE;
  int64_t tp = -1, ti = -1;
  uint32_t ta = tpoly[a], tb = with_end ? tpoly[G.end[p]] : GTS_NONE;
  if (vtime) { if (ta != GTS_NONE) ta = vtime[ta]; if (tb != GTS_NONE) tb = vtime[tb]; }
  if (ta != GTS_NONE) tp = 2 * (int64_t)ta;
  if (tb != GTS_NONE && 2 * (int64_t)tb > tp) tp = 2 * (int64_t)tb;
  const uint32_t oa = ovf[a];
  if ((oa & GTS_OV_ACTIVE1) && (oa & (s ? GTS_OV_S : GTS_OV_A)))
    ti = 2 * (int64_t)(vtime ? vtime[a] : a) + 1;
  const uint32_t lh = lasthit[2 * (uint64_t)a + (s ? 1 : 0)];
  if (lh != GTS_NONE && 2 * (int64_t)lh + 1 > ti) ti = 2 * (int64_t)lh + 1;
  if (tp < 0 && ti < 0) return G.state[p];
  return tp > ti ? GIS_POLYMORPHIC : GIS_INCONSISTENT;
}

#endif
