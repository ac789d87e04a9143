/*
  gts_component.hpp -- cycle removal and scaffold construction for ONE weakly
  connected component of the filtered scaffold graph, executed by ONE
  wavefront.

  ref src/gt_scaffolder_algorithms.c:346-868 (isterminal, calc_cc_and_terminals,
  detect_cycle_recursive, removecycles, create_walk, makescaffold).

  Why components: every traversal of those functions follows unmarked edges
  between unmarked vertices, and every mark they set lands on edges incident
  to the vertex being marked.  Edge states are NOT symmetric after the filter
  (algorithms.c:249-258 marks edges whose twins stay unmarked), so the unit of
  independence is the WEAKLY connected component (an unmarked edge in either
  direction joins its end vertices).  Inside a component the reference's
  order-dependent semantics are kept exactly: its "connected components" are
  directed BFS reachability sets taken in vertex-index order, terminals are in
  BFS order, the DFS visits adjacency lists in insertion order, walks use the
  FIFO label-correcting search with float distance maps, strict-greater
  tie-breaks and LIFO evaluation of reached terminals.

  Execution model: the wavefront runs the component's sequential program with
  wave-uniform control flow; the 64 lanes share each adjacency-list scan
  (one coalesced 64-entry chunk per step, __ballot to pick / order the
  qualifying edges, popcount prefix sums to append to the BFS / walk queues in
  list order) and the back-tracking of the reached terminals.  W is the wave
  policy: 64 lanes on gfx950; the test harness instantiates it with 1 lane on
  the host.

  Component-local ("compact") graph: vertices of all non-trivial components
  are renumbered into slots (sorted by component, then vertex index) and their
  live edges (unmarked when the component phase starts) are copied into a
  compact CSR in list order.  Marks set here (CYCLIC, SCAFFOLD) go to the
  global graph AND to the compact copy.

  Everything the program touches is addressed with COMPONENT-LOCAL indices
  (slot 0..nv-1, compact edge 0..ne-1) through GtsCompMem, a bundle of base
  pointers.  The bases either point into the global arrays (any size) or into
  the workgroup's LDS, where the launcher has staged the component's graph and
  scratch (gts_engine.hip: one launch per LDS size class): the dependent
  pointer chasing of BFS / DFS / walks then runs at LDS latency.
*/
#ifndef GTS_COMPONENT_HPP
#define GTS_COMPONENT_HPP

#include "gts_defs.h"

enum { GTS_MODE_REMOVECYCLES = 0, GTS_MODE_MAKESCAFFOLD = 1 };
enum { GTS_CERR_NONE = 0, GTS_CERR_WALKQ_OVERFLOW = 1, GTS_CERR_WALK_LOOP = 2, GTS_CERR_PATH_OVERFLOW = 3 };

struct GtsCompView {
  GtsGraphView G;            /* global graph (marks are mirrored there) */
  const uint32_t *cmap;      /* m: global position -> compact edge or NONE */
  uint32_t ncomp;
  const uint32_t *comp_off;  /* ncomp+1: slot range of a component */
  const uint32_t *slot_v;    /* slot -> global vertex */
  const int64_t *cseq;       /* slot -> seq_len */
  const uint32_t *coff;      /* nslots+1: compact edge range of a slot */
  const uint32_t *cstart;    /* compact edge -> start slot, component-local */
  const uint32_t *cend;      /* compact edge -> end slot, component-local */
  const int64_t *cdist;
  uint8_t *cflags;
  const uint32_t *cgpos;     /* compact edge -> global position */
  uint8_t *cstate;           /* compact copy of the edge state */
  uint8_t *vst;              /* slot -> vertex state */
  /* per-slot scratch (a component uses its own slot range) */
  uint32_t *queue, *term, *visited, *st_v, *st_par, *st_cur, *edgemap,
      *lastpop, *wterm, *touched, *cc_best;
  uint8_t *st_dir;
  float *distmap;            /* all GTS_DIST_UNSET between walks */
  uint32_t *ccoff;           /* nslots + ncomp entries; comp c at comp_off[c]+c */
  /* walk FIFO of the reference search: a ring carved out of one pool the first
     time a component needs it (few components do) */
  uint32_t *wq_edge;
  int64_t *wq_dist;
  unsigned long long *wq_used;  /* pool entries handed out */
  uint64_t wq_pool;          /* pool entries */
  uint64_t wq_factor;        /* ring = factor * compact edges of the component + 64 */
  uint32_t *cerr;            /* ncomp: error code per component */
  uint64_t max_pops;         /* bound on queue pops of one walk */
  /* linear-time walk (create_walk_fast) */
  int fast_walks;            /* 0: always run the reference's search */
  int batch_walks;           /* LDS-resident clean components: the walks of a cc side by side */
  int small_masks;           /* LDS-resident components of at most 64 contigs: peel_small() */
  int team_coff;             /* k_components_team: the list offsets in LDS during the walks of a cc */
  int timing_skip_writeback; /* timing aid: run_fast() does not write its results to the global graph */
  int local_marks;           /* the LDS programs keep their marks in the working copy (GtsComponent::local_marks) */
  int help_walks;            /* k_components_pool: jobs for the walks that are made one by one (GtsHelpJob) */
  char *team_slab;           /* k_components_team: walk slots and path buffers of the workgroups */
  unsigned long long *team_used;   /* bytes handed out */
  uint64_t *tspan;           /* per component x2 (or null): clock at the start and the end of its program */
  unsigned long long *small_stat;  /* [4] statistics: components of at most 64 contigs whose compact edges are
                                      all live / all components of that size, and their cycle-removal ticks */
  unsigned long long *team_stat;   /* [8] statistics: ccs posted, batches, sweep steps, ccs with a tie,
                                      ticks clearing slots / sweeping / extracting paths / wavefront 0 at the barriers */
  uint64_t team_cap;
  int64_t *nd;               /* slot -> integer label pushed with the node */
  uint64_t *plen;            /* slot -> contig length of the tree path */
  uint8_t *tight;            /* slot -> number of tight in-arcs (saturating) */
  uint32_t *par;             /* nslots: parent vertex of the linear walks */
  uint8_t *gorient;          /* slot -> strand + 1 of the whole-component analysis */
  uint32_t *topo, *tpos;     /* topological order of the forward sheet, inverse */
  uint32_t *stat_clean;      /* per component: bit0 the analysis succeeded, bit1 walks deferred,
                                bits2-3 why not (1 revivable twin, 2 one terminal, 3 pool full), nterm << 8 */
  /* walks of large clean components fan out: the component program emits one
     task per terminal, k_walk_tasks runs every walk on its own wavefront,
     the select pass keeps the best walk of every cc (gts_engine.hip) */
  uint32_t defer_min_nv;     /* 0: never defer */
  uint64_t defer_unclean_work;   /* a component that is not clean (its walks are made one by one, ten
                                    times the time per contig) defers from this many terminals x
                                    contigs on -- if some component of the launch has deferred already,
                                    so that rounds of walk tasks follow anyway (on their own they cost
                                    more than the launch gains); 0: the rules above only */
  uint64_t defer_min_work;   /* defer only if terminals x contigs reaches this: the walks of a
                                component cost about that many vertex steps when made in place */
  uint32_t defer_ref_min_nv; /* 0: never.  A component of at least this many contigs that meets a walk
                                the linear-time walks cannot make (it needs the reference's search,
                                which can take 10^5 .. 10^6 queue pops) stops there and hands the
                                walks of the ccs it has not decided yet to tasks: the walks of a cc
                                are independent (ref algorithms.c:809-832), each task replays the
                                search on a wavefront of its own with a ring of its own */
  int task_reference;        /* walk tasks replay the reference's search themselves (else select_walks
                                does, one after the other) */
  uint8_t *defer_flag;       /* ncomp */
  uint32_t *comp_task0, *comp_ncc, *comp_nterm;   /* ncomp */
  unsigned long long *ntasks, *path_used;         /* device counters */
  unsigned long long *task_bytes;                 /* statistics: bytes the walk tasks staged */
  uint64_t task_cap, path_cap;
  uint32_t *task_comp, *task_start, *task_n;      /* task_cap */
  uint8_t *task_skip;
  uint64_t *task_len, *task_poff;
  uint32_t *paths;           /* path_cap: pool for the tasks' bitmaps of labelled vertices
                                (fixed at defer time) and their walks (component-local edges) */
  uint32_t *comp_next_cc;    /* per deferred component: first cc not yet decided */
  uint64_t *task_roff;       /* task_cap: offset of the task's bitmap in paths */
  uint64_t *comp_ring;       /* per component: 2 x u64, ring of select_walks' reference searches */
  const uint8_t *comp_klass; /* per component: LDS size class (launch group of its tasks) */
  const uint8_t *comp_d32;   /* per component: a distance does not fit 16 bits (packed layout: int32 distances) */
  uint32_t *tq;              /* task_cap: pending tasks, one segment per class */
  const uint32_t *tq_base;   /* per class: start of its segment */
  unsigned long long *tq_cnt;/* per class: pending tasks */
  uint32_t *defer_list;      /* ncomp: the deferred components */
  unsigned long long *ndeferred;
  uint32_t *wbits;           /* nslots / 32 + ncomp + 1 words: select_walks' bitmap of component c
                                starts at comp_off[c] / 32 + c */
  uint32_t *stat_fast, *stat_slow;  /* per component: walks by path taken */
  uint32_t *stat_ncc;        /* per component: ccs of the last terminal search */
  unsigned long long *why;   /* [8] why walks left the linear path: mixed start,
                                self arc, back at start, marked end, two
                                directions, inexact tie, cycle, inexact length tie */
  uint64_t *tstat;           /* per component x5: ticks in removecycles, makescaffold
                                outside walks, fast walks, reference walks; pops of
                                the reference walks */
};

/* A component that runs from global memory with a whole workgroup ("team",
   k_components_team): wavefront 0 runs the component program; when it comes to
   the walks of a cc -- independent of each other, ref algorithms.c:809-832 -- it
   posts the cc here, every wavefront of the workgroup sweeps batches of eight of
   its terminals (walks_clean_batch_global) and leaves the best walk it has seen. */
#define GTS_TEAM_WAVES_MAX 16
struct GtsTeamCtl {
  uint32_t kind;             /* 1: walks of the cc [tb, te); 2: the topological order (peel_team); 0: done, the helpers leave */
  uint32_t tb, te;
  uint32_t pq_lvl[3];        /* peel_team: contigs appended in level l, at [l % 3] */
  uint32_t slab_ok;
  unsigned long long slab;   /* byte offset of the workgroup's slab */
  unsigned long long len[GTS_TEAM_WAVES_MAX];   /* per wavefront: its longest walk, */
  uint32_t j[GTS_TEAM_WAVES_MAX];               /* the first terminal (index) that attains it, */
  uint32_t n[GTS_TEAM_WAVES_MAX];               /* its number of edges (the path is in the slab) */
  uint32_t bad[GTS_TEAM_WAVES_MAX];             /* a walk of its batches met a tie */
};

/* Pointer type of the component's working set: generic pointers into the
   global arrays, or address-space-3 pointers when the component is staged in
   LDS.  LDS pointers matter: accesses become ds_read / ds_write (lgkmcnt only);
   through generic pointers they would be flat_* instructions, whose results
   also wait for every outstanding global store (the walk-queue pushes). */
template <class T, bool LDS> struct GtsPtrSel { typedef T *type; };
#if defined(__HIPCC__)
template <class T> struct GtsPtrSel<T, true> {
  typedef T __attribute__((address_space(3))) *type;
};
#endif
#define GTS_P(T) typename GtsPtrSel<T, LDS>::type

/* Per-edge arrays of the working set.  In the global arrays the start vertex,
   the flags and the state of a compact edge are arrays of their own.  The LDS
   copy spends 7 instead of 10 bytes per edge: flags (low nibble) and state
   (high nibble) share a byte, and the start vertex is not stored -- the few
   places that need it outside the walks' parent array search the offsets. */
template <bool LDS> struct GtsEdgeArrays {
  typedef uint8_t *flags_t; typedef uint8_t *state_t; typedef const uint32_t *start_t;
};
#if defined(__HIPCC__)
typedef uint8_t __attribute__((address_space(3))) *gts_lds_u8p;
struct GtsLdsStateRef {
  gts_lds_u8p p;
  /* bit 7 of the byte: the state is a marked one (gts_edge_is_marked) -- the
     test every traversal makes per edge, one AND instead of shift, mask, shift,
     mask, compare */
  __device__ __forceinline__ operator uint8_t() const { return (uint8_t)((*p >> 4) & 7u); }
  __device__ __forceinline__ void operator=(uint8_t st) const
  { *p = (uint8_t)((*p & 15u) | (st << 4) | (gts_edge_is_marked(st) ? 0x80u : 0u)); }
  __device__ __forceinline__ void operator=(const GtsLdsStateRef &o) const { *this = (uint8_t)o; }
};
struct GtsLdsFlagsRef {
  gts_lds_u8p p;
  __device__ __forceinline__ operator uint8_t() const { return (uint8_t)(*p & 15u); }
  __device__ __forceinline__ void operator=(uint8_t f) const { *p = (uint8_t)((*p & 0xF0u) | (f & 15u)); }
  __device__ __forceinline__ void operator=(const GtsLdsFlagsRef &o) const { *this = (uint8_t)o; }
};
struct GtsLdsStateArr {
  gts_lds_u8p b;
  __device__ __forceinline__ GtsLdsStateRef operator[](uint32_t i) const { return GtsLdsStateRef{b + i}; }
};
struct GtsLdsFlagsArr {
  gts_lds_u8p b;
  __device__ __forceinline__ GtsLdsFlagsRef operator[](uint32_t i) const { return GtsLdsFlagsRef{b + i}; }
};
struct GtsLdsStartArr {
  const uint16_t __attribute__((address_space(3))) *coff;   /* local offsets, nv + 1 */
  uint32_t nv;
  __device__ __forceinline__ uint32_t operator[](uint32_t ce) const
  {
    uint32_t lo = 0, hi = nv;          /* last vertex whose list starts at or before ce */
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (coff[mid] <= ce) lo = mid; else hi = mid;
    }
    return lo;
  }
};
template <> struct GtsEdgeArrays<true> {
  typedef GtsLdsFlagsArr flags_t; typedef GtsLdsStateArr state_t; typedef GtsLdsStartArr start_t;
};
#endif

/* Element types of the working set.  In the global arrays everything is
   32 / 64 bit.  The LDS copy is packed: component-local slot and edge numbers
   fit 16 bits (a component that fits 160 KB of LDS has far fewer than 65535
   vertices or edges), distances and contig lengths are staged as int32 (a
   component holding a wider value is run from global memory instead). */
/* nd_t: integer labels of the linear-time walks; len_t: contig bases along a
   walk.  The packed LDS layout is used only for components whose sums fit
   (k_comp_lds_keys: fewer than 4096 contigs, |distance| < 2^19, contig
   lengths adding up to less than 2^32), so 32 bits hold them and the float
   conversions are single instructions. */
template <bool LDS> struct GtsCompTypes {
  typedef uint32_t idx_t; typedef int64_t dist_t; typedef int64_t seq_t;
  typedef int64_t nd_t; typedef uint64_t len_t;
};
template <> struct GtsCompTypes<true> {
  typedef uint16_t idx_t; typedef int32_t dist_t; typedef int32_t seq_t;
  typedef int32_t nd_t; typedef uint32_t len_t;
};

/* base pointers of ONE component, component-local indices */
template <bool LDS>
struct GtsCompMemT {
  typedef typename GtsCompTypes<LDS>::idx_t idx_t;
  typedef typename GtsCompTypes<LDS>::dist_t dist_t;
  typedef typename GtsCompTypes<LDS>::seq_t seq_t;
  typedef typename GtsCompTypes<LDS>::nd_t nd_t;
  typedef typename GtsCompTypes<LDS>::len_t len_t;
  uint32_t nv, ne;           /* slots, compact edges of the component */
  uint32_t e0;               /* value to subtract from coff[] entries */
  GTS_P(const idx_t) coff;   /* nv+1 */
  typename GtsEdgeArrays<LDS>::start_t cstart;
  GTS_P(const idx_t) cend;
  GTS_P(const dist_t) cdist;
  /* packed LDS layout: a component whose distances all fit 16 bits (comp_d32 = 0)
     keeps them as int16 -- 5 instead of 7 bytes per edge; the launch is bound
     by LDS x time, so a seventh less footprint is a seventh more components in
     flight (read through GtsComponent::dist_of) */
  GTS_P(const int16_t) cdist16;
  bool d16;
  typename GtsEdgeArrays<LDS>::flags_t cflags;   /* GTS_F_TWINLIVE is cleared when a twin dies */
  GTS_P(const seq_t) cseq;
  typename GtsEdgeArrays<LDS>::state_t cstate;
  GTS_P(uint8_t) vst;
  GTS_P(idx_t) queue;
  GTS_P(idx_t) term;
  GTS_P(idx_t) visited;
  GTS_P(idx_t) st_v;
  GTS_P(idx_t) st_par;
  GTS_P(idx_t) st_cur;
  GTS_P(idx_t) edgemap;
  GTS_P(idx_t) par;          /* linear walks: start vertex of edgemap[v] */
  GTS_P(uint32_t) lastpop;
  GTS_P(idx_t) wterm;
  GTS_P(idx_t) touched;
  GTS_P(idx_t) cc_best;
  GTS_P(idx_t) ccoff;
  GTS_P(uint8_t) st_dir;
  GTS_P(uint8_t) tight;
  GTS_P(float) distmap;
  GTS_P(nd_t) nd;
  GTS_P(len_t) plen;
  /* whole-component analysis (orient + peel): strand of every vertex + 1, a
     topological order of the forward sheet and its inverse */
  GTS_P(uint8_t) gorient;
  GTS_P(idx_t) topo;
  GTS_P(idx_t) tpos;
  /* packed LDS layout only: the walk scratch (distmap ... st_par) is one
     contiguous region that the batched clean walks re-view as `wslots` walk
     slots of gts_walk_slot_bytes(nv) each (walks_clean_batch); slot 0 starts
     with distmap.  0 slots: no batched walks (global arrays). */
  GTS_P(char) wbase;
  uint32_t wslots;
  /* walk tasks only (or null): the arcs of every vertex split by sense -- sarc
     holds the compact edges of vertex v with sense 1 in list order at
     [coff[v], smid[v]) and those with sense 0 at [smid[v], coff[v+1]).  The
     reference's search relaxes only the arcs that leave a vertex in the node's
     direction (algorithms.c:699): with the split view half as many lanes carry
     an arc that cannot be relaxed (create_walk_reference) */
  GTS_P(const idx_t) sarc;
  GTS_P(const idx_t) smid;
};
typedef GtsCompMemT<false> GtsCompMem;

/* ---- walks of a cc over the wavefronts of the workgroup (round 4) ------------------
   The walks of a cc are independent (ref algorithms.c:809-832).  Where they cannot be
   swept side by side on one wavefront -- a component that is not clean and whose
   reachable states hold a cycle (walk_cyclic: four traversals a walk), a tie without a
   closed form -- they were made one after the other: 45 walks of a 380-contig component
   took 8 of the launch's 10.4 ms, a few components of 50 - 170 contigs claimed late set
   its last millisecond.  In k_components_pool the wavefronts of a workgroup share the
   component's LDS, so the owner opens a JOB in the workgroup's control block: wavefronts
   that are between components (or have none left to claim) take terminals off a counter,
   walk with scratch of their own (pages of the pool) on the owner's graph, and leave
   their best walk; the owner takes the first strictly longest in terminal order
   (algorithms.c:826-832), copies its path and closes the job.  One job per workgroup at a
   time; an owner that finds the slot taken, or nobody to help, walks alone as before. */
#define GTS_HUB_WAVES 16
#if defined(__HIPCC__)
struct GtsHelpJob {
  uint32_t seq;            /* odd: a job is open; changes when it closes */
  uint32_t ready;          /* = seq once the fields below are filled */
  uint32_t comp, tb, te, clean, nv;
  uint32_t next;           /* next terminal (index into term[]) */
  uint32_t active;         /* participants that may still write results */
  uint32_t need_ref;       /* a walk needs the reference's search: the owner starts over, alone */
  uint32_t walks;          /* statistics */
  uint32_t n_running;      /* wavefronts of the workgroup inside a component program (they may open jobs) */
  unsigned long long best_len[GTS_HUB_WAVES];
  uint32_t best_j[GTS_HUB_WAVES], best_n[GTS_HUB_WAVES], best_start[GTS_HUB_WAVES];
  uint32_t best_path[GTS_HUB_WAVES];   /* LDS address of the participant's cc_best */
  GtsCompMemT<true> M;     /* the owner's view of the component */
};
#else
struct GtsHelpJob {      /* (host harness: never used, the fields are there for the parser) */
  uint32_t seq, ready, comp, tb, te, clean, nv, next, active, need_ref, walks, n_running;
  unsigned long long best_len[GTS_HUB_WAVES];
  uint32_t best_j[GTS_HUB_WAVES], best_n[GTS_HUB_WAVES], best_start[GTS_HUB_WAVES], best_path[GTS_HUB_WAVES];
  int M;
};
#endif

/* LDS bytes needed to stage a component in the packed layout (every array
   16-byte aligned) */
GTS_HD uint32_t gts_comp_lds_bytes(uint32_t nv, uint32_t ne, bool d32 = true)
{
  const uint32_t a = 16;
  uint32_t b = 0;
  b += (((nv + 1) * 2 + a - 1) / a) * a * 2;          /* coff, ccoff */
  /* scratch that is never live at the same time shares storage: st_cur (cycle
     search) with cc_best (walks), touched (reference search) with visited,
     lastpop (reference search, zeroed on entry) with nd (linear walks).
     The walk scratch -- distmap, plen, edgemap, par | nd, queue, visited,
     st_v, wterm, st_par -- comes last and in one piece: two walk slots
     (gts_walk_slot_bytes), more if the launcher has room behind it */
  b += ((nv * 2 + a - 1) / a) * a * 11;               /* term, cc_best, topo, tpos | edgemap, par, queue, visited, st_v, wterm, st_par */
  b += ((nv * 4 + a - 1) / a) * a * 4;                /* cseq | distmap, plen, nd */
  b += ((nv + a - 1) / a) * a * 4;                    /* vst, st_dir, tight, gorient */
  b += ((ne * 2 + a - 1) / a) * a;                    /* cend */
  b += ((ne * (d32 ? 4u : 2u) + a - 1) / a) * a;      /* cdist: int32, or int16 when every distance fits */
  b += ((ne + a - 1) / a) * a;                        /* flags + state */
  return b;
}
/* One walk slot of the batched clean walks: label (f32), tree-path length
   (u32), edgemap and parent (u16) per contig.  The scratch region of the
   layout above (3 arrays of 4 bytes, 7 of 2 bytes per contig) holds two. */
GTS_HD uint32_t gts_walk_slot_bytes(uint32_t nv)
{
  const uint32_t p4 = ((nv * 4 + 15) / 16) * 16, p2 = ((nv * 2 + 15) / 16) * 16;
  return 2 * p4 + 2 * p2;
}
#define GTS_WALK_SLOTS_MAX 8u
#ifndef GTS_WALK_LANES
#define GTS_WALK_LANES 8u   /* lanes of a walk in a batch: 8 walks per wavefront */
#endif
/* LDS bytes a component asks for: its footprint plus, from `big_nv` contigs
   on, room for up to `big_slots` walk slots (as many as fit `limit`): the
   walks of a large component are the critical path of the launch, and a cc's
   walks are independent (ref algorithms.c:809-832), so they are swept side by
   side (walks_clean_batch).  Smaller components use what their last page has
   left. */
/* Second tier (round 4): from `huge_nv` contigs on, `huge_slots`.  The launch ends
   with its longest programs, the few components of several hundred contigs whose
   ccs hold 10 - 20 terminals each: with eight slots a wavefront sweeps a cc in two
   or three batches instead of six (they are too few for their LDS to matter). */
GTS_HD uint32_t gts_comp_lds_want(uint32_t nv, uint32_t ne, uint32_t big_nv, uint32_t big_slots,
                                  uint32_t limit, bool d32 = true, uint32_t huge_nv = 0,
                                  uint32_t huge_slots = 0)
{
  uint32_t need = gts_comp_lds_bytes(nv, ne, d32);
  if (big_nv && nv >= big_nv && need <= limit) {
    const uint32_t sb = gts_walk_slot_bytes(nv);
    if (huge_nv && nv >= huge_nv && huge_slots > big_slots) big_slots = huge_slots;
    uint32_t extra = big_slots > 2 ? big_slots - 2 : 0;
    while (extra && need + extra * sb > limit) --extra;
    need += extra * sb;
  }
  return need;
}
/* the packed layout addresses at most this many slots / edges */
#define GTS_LDS_MAX_INDEX 65000u

/* (float)GT_WORD_MAX, ref algorithms.c:650 */
#define GTS_DIST_UNSET 9223372036854775808.0f

template <class W, bool LDS = false, bool SPLIT = false>
struct GtsComponent {
  const GtsCompView &C;
  const GtsCompMemT<LDS> &M;
  uint32_t c, s0, e0g;  /* component, its first slot and first compact edge */
  uint32_t nv;
  uint32_t nterm, ncc;  /* filled by calc_cc */
  uint32_t err;
  /* walk-queue state of the walk in flight */
  uint64_t qbase, qcap, qh, qn;
  uint32_t ntouch;
  uint32_t nfast, nslow;
  uint64_t tfast, tslow, npops;
  bool clean;           /* oriented, D acyclic (orient + peel) and nothing changed since */
  bool reuse_cc;        /* makescaffold may use the ccs run() computed */
  uint32_t nodefer;     /* statistics: why try_defer declined */
  uint32_t *reach_bits; /* walk_task: bitmap of the vertices the walk labels */
  uint64_t ubases;      /* all_bases(), ~0 = not computed yet */
  uint32_t walk_from;   /* reference search: the walk's start vertex */
  bool no_reference;    /* do not run the reference search in place: walk_task leaves it to
                           select_walks (task_reference = 0), makescaffold hands it to a task */
  bool needs_reference;
  bool deferred_late;   /* makescaffold stopped at a cc and published the rest as tasks */
  bool was_all_live;    /* statistics */
  /* round 4: the CYCLIC and SCAFFOLD marks of the program stay in the working copy
     (mark_vertex_cyclic_lds, mark_best_lds) and reach the global graph in one go
     (flush_local_marks) when the component is done or hands its walks to tasks:
     no chain of dependent global reads per marked edge in the middle of the program */
  bool local_marks, any_scaffold_marks;
  bool run_clean, run_deferred;   /* run(): the component was clean after cycle removal / handed its walks to tasks */
  bool lean_stats;      /* no clocks, no per-component statistics (k_components_pool outside the detailed
                           profile: the kernel adds nfast / nslow / clean / deferred up itself) */
  GtsHelpJob *hub;      /* k_components_pool: the workgroup's job slot (or null) */
  uint32_t hub_me;      /* this wavefront's index in the workgroup (its result entry of the job) */
  /* k_components_team */
  GtsTeamCtl *team;     /* null: no team */
  char *team_base;      /* this workgroup's slab */
  uint32_t team_wave, team_waves;
  /* LDS addresses of the team's vertex states, BFS queue and edge scratch (GTS_NONE:
     not in LDS): calc_cc_team reaches them with ds_* instructions */
  uint32_t tl_vst, tl_queue, tl_scratch, tl_stv;
  /* LDS the team has no other use for while it makes the walks of a cc (the
     search's scratch, its queue, the degrees of peel): the position bitmaps of
     the walks in flight (walks_clean_batch_global) */
  uint32_t tl_pbits, tl_pbits_bytes;

  GTS_HD GtsComponent(const GtsCompView &cv, const GtsCompMemT<LDS> &mem, uint32_t comp)
      : C(cv), M(mem), c(comp), s0(cv.comp_off[comp]), e0g(cv.coff[cv.comp_off[comp]]),
        nv(mem.nv), nterm(0), ncc(0), err(0), qbase(0), qcap(0), qh(0), qn(0),
        ntouch(0), nfast(0), nslow(0), tfast(0), tslow(0), npops(0), clean(false), reuse_cc(false), nodefer(0), reach_bits(nullptr), ubases(~0ull), walk_from(0), no_reference(false), needs_reference(false), deferred_late(false), was_all_live(false), local_marks(false), any_scaffold_marks(false), run_clean(false), run_deferred(false), lean_stats(false), hub(nullptr), hub_me(0), team(nullptr), team_base(nullptr), team_wave(0), team_waves(1), tl_vst(GTS_NONE), tl_queue(GTS_NONE), tl_scratch(GTS_NONE), tl_stv(GTS_NONE), tl_pbits(GTS_NONE), tl_pbits_bytes(0) {}

  /* a clock read costs a wait for every LDS operation in flight: only where somebody looks */
  GTS_HD uint64_t tick() const { return lean_stats ? 0 : W::clock(); }

  /* bases into the global arrays */
  static GTS_HD GtsCompMem global_mem(const GtsCompView &C, uint32_t comp)
  {
    GtsCompMem m;
    const uint32_t s0 = C.comp_off[comp], s1 = C.comp_off[comp + 1];
    const uint32_t e0 = C.coff[s0];
    m.nv = s1 - s0; m.ne = C.coff[s1] - e0; m.e0 = e0;
    m.coff = C.coff + s0; m.cstart = C.cstart + e0; m.cend = C.cend + e0;
    m.cdist = C.cdist + e0; m.cdist16 = nullptr; m.d16 = false; m.cflags = C.cflags + e0; m.cseq = C.cseq + s0;
    m.cstate = C.cstate + e0; m.vst = C.vst + s0;
    m.queue = C.queue + s0; m.term = C.term + s0; m.visited = C.visited + s0;
    m.st_v = C.st_v + s0; m.st_par = C.st_par + s0; m.st_cur = C.st_cur + s0;
    m.edgemap = C.edgemap + s0; m.par = C.par + s0; m.lastpop = C.lastpop + s0; m.wterm = C.wterm + s0;
    m.touched = C.touched + s0; m.cc_best = C.cc_best + s0;
    m.ccoff = C.ccoff + s0 + comp; m.st_dir = C.st_dir + s0; m.tight = C.tight + s0;
    m.distmap = C.distmap + s0; m.nd = C.nd + s0; m.plen = C.plen + s0;
    m.gorient = C.gorient + s0; m.topo = C.topo + s0; m.tpos = C.tpos + s0;
    m.wbase = nullptr; m.wslots = 0; m.sarc = nullptr; m.smid = nullptr;
    return m;
  }

  typedef typename GtsCompTypes<LDS>::nd_t nd_t;
  typedef typename GtsCompTypes<LDS>::len_t len_t;
  static GTS_HD int32_t uni_t(int32_t v) { return (int32_t)W::uni((uint32_t)v); }
  static GTS_HD uint32_t uni_t(uint32_t v) { return W::uni(v); }
  static GTS_HD int64_t uni_t(int64_t v) { return W::uni64(v); }
  static GTS_HD uint64_t uni_t(uint64_t v) { return (uint64_t)W::uni64((int64_t)v); }
  GTS_HD uint32_t eoff(uint32_t ls) const { return W::uni(M.coff[ls]) - M.e0; }
  /* distance of a compact edge (GtsCompMemT::cdist16) */
  GTS_HD typename GtsCompMemT<LDS>::dist_t dist_of(uint32_t ce) const
  {
    if constexpr (LDS) { if (M.d16) return (typename GtsCompMemT<LDS>::dist_t)M.cdist16[ce]; }
    return M.cdist[ce];
  }
  /* the same with the width known at compile time (the sweep loops: no branch
     on M.d16 per step) */
  template <bool D16>
  GTS_HD int32_t dist_w(uint32_t ce) const
  {
    if constexpr (LDS) { if constexpr (D16) return (int32_t)M.cdist16[ce]; else return (int32_t)M.cdist[ce]; }
    else return (int32_t)M.cdist[ce];
  }
  /* flags (low nibble) and state (high nibble) of a compact edge; the packed
     LDS layout keeps them in one byte */
  /* marked state, from edge_bits() / of edge ce */
  static GTS_HD bool bits_marked(uint32_t fs)
  {
    if constexpr (LDS) return (fs & 0x80u) != 0;
    else return gts_edge_is_marked((uint8_t)(fs >> 4));
  }
  GTS_HD bool edge_marked(uint32_t ce) const
  {
    if constexpr (LDS) return (*(M.cflags.b + ce) & 0x80u) != 0;
    else return gts_edge_is_marked(M.cstate[ce]);
  }
  GTS_HD uint32_t edge_bits(uint32_t ce) const
  {
    if constexpr (LDS) return *(M.cflags.b + ce);
    else return (uint32_t)(M.cflags[ce] & 15u) | ((uint32_t)M.cstate[ce] << 4);
  }
  /* walk_task: the vertices a search labels (see try_defer) */
  GTS_HD void note_labelled(uint32_t v) const
  {
    if (reach_bits) W::or_bits(reach_bits + (v >> 5), 1u << (v & 31));
  }

  /* ---- ref algorithms.c:379-436 (with isterminal, :346-373, fused) ---- */
  GTS_HD void calc_cc()
  {
#if defined(__HIPCC__)
    if constexpr (W::TEAM && !LDS) {
      if (tl_vst != GTS_NONE && tl_queue != GTS_NONE && tl_scratch != GTS_NONE && tl_stv != GTS_NONE) { calc_cc_team(); return; }
    }
#endif
    bool unused;
    calc_cc_t<false>(unused);
  }

#if defined(__HIPCC__)
  /* calc_cc_t<false> for a component in global memory whose vertex states and
     queue the team keeps in LDS (k_components_team).  The search visits one
     vertex at a time in queue order (ref algorithms.c:398-430: that order is the
     order of the terminals, which decides between walks of equal length), and
     through generic pointers a vertex cost a round trip to L2 for its edges
     plus the wait for the global stores in flight that a flat_load implies:
     ~1.1 us a vertex, ten passes over the 8247 contigs of the 50 M workload's
     largest component = 90 ms.  Here the states and the queue are reached with
     ds_read / ds_write, and the list bounds and the first GTS_TCC_K edges of the 64
     queue vertices a chunk holds are fetched by all lanes at once into an LDS
     scratch: a vertex then costs a few LDS round trips (longer lists read the
     rest from L2 as before). */
#define GTS_TCC_K 8u
  GTS_HD void calc_cc_team()
  {
    typedef uint8_t __attribute__((address_space(3))) *l8p;
    typedef uint32_t __attribute__((address_space(3))) *l32p;
    typedef uint64_t __attribute__((address_space(3))) *l64p;
    const l8p vst = (l8p)(uintptr_t)tl_vst;
    const l32p queue = (l32p)(uintptr_t)tl_queue;
    const l64p scratch = (l64p)(uintptr_t)tl_scratch;   /* [64][GTS_TCC_K]: end vertex | edge bits << 32 */
    const l32p claim = (l32p)(uintptr_t)tl_stv;         /* a word per vertex: the lowest lane that wants to append it */
    const uint32_t lane = W::lane();
    /* the other functions reach the same bytes with flat_* instructions, which
       are not ordered with ds_* ones: everything in flight lands first (and
       the ds_writes below before this returns) */
    __builtin_amdgcn_s_waitcnt(0);
    for (uint32_t s = lane; s < nv; s += W::WIDTH) {
      if (!gts_vertex_is_marked(vst[s])) vst[s] = GIS_UNVISITED;
      claim[s] = 0xFFFFFFFFu;   /* (a vertex is appended once, so its word is used once) */
    }
    W::fence();
    auto ccoff = M.ccoff;
    nterm = 0; ncc = 0;
    for (uint32_t base0 = 0; base0 < nv; base0 += W::WIDTH) {
      uint32_t next_lane = 0;
      for (;;) {
        bool cand = false;
        if (base0 + lane < nv && lane >= next_lane) {
          const uint8_t st = vst[base0 + lane];
          cand = !gts_vertex_is_marked(st) && st != GIS_VISITED;
        }
        const uint64_t cm = W::ballot(cand);
        if (!cm) break;
        const uint32_t l0 = W::ctz(cm);
        next_lane = l0 + 1;
        const uint32_t s = base0 + l0;
        if (lane == 0) { vst[s] = GIS_PROCESSED; queue[0] = s; }
        ccoff[ncc++] = nterm;
        W::fence();
        uint32_t bh = 0, bn = 1;
        while (bh < bn) {
          const uint32_t cnt = bn - bh < W::WIDTH ? bn - bh : W::WIDTH;
          uint32_t my_v = 0, my_eb = 0, my_ee = 0;
          if (lane < cnt) {
            my_v = queue[bh + lane];
            my_eb = M.coff[my_v] - M.e0; my_ee = M.coff[my_v + 1] - M.e0;
          }
          {
            uint32_t nb[GTS_TCC_K], fl[GTS_TCC_K];
#pragma unroll
            for (uint32_t k = 0; k < GTS_TCC_K; ++k) {
              const uint32_t ce = my_eb + k < my_ee ? my_eb + k : my_eb;
              const bool in = lane < cnt && my_eb + k < my_ee;
              nb[k] = in ? (uint32_t)M.cend[ce] : 0u;
              fl[k] = in ? edge_bits(ce) : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < GTS_TCC_K; ++k)
              scratch[lane * GTS_TCC_K + k] = (uint64_t)nb[k] | (uint64_t)fl[k] << 32;
          }
          W::fence();
          /* which of the chunk's vertices have at most GTS_TCC_K entries: up to eight
             of those in a row are visited in ONE step, eight lanes each.  The queue
             order stays the reference's: the lanes are in (queue position, list
             position) order, a neighbour two of them see unvisited goes to the
             lowest lane (an LDS minimum on a word per vertex), and the winners are
             appended in lane order. */
          const uint64_t shortm = W::ballot(lane < cnt && my_ee - my_eb <= GTS_TCC_K);
          for (uint32_t i = 0; i < cnt;) {
            const uint64_t longs = ~shortm >> i;
            uint32_t nbat = longs ? W::ctz(longs) : W::WIDTH;
            if (nbat > cnt - i) nbat = cnt - i;
            if (nbat > W::WIDTH / GTS_TCC_K) nbat = W::WIDTH / GTS_TCC_K;
            if (nbat >= 2) {
              const uint32_t g = lane / GTS_TCC_K, a = lane % GTS_TCC_K;
              const uint32_t vi = i + (g < nbat ? g : 0u);
              const uint32_t cur = W::shfl(my_v, vi), eb = W::shfl(my_eb, vi), ee = W::shfl(my_ee, vi);
              const bool in = g < nbat && eb + a < ee;
              uint32_t fl = 0, nb = 0;
              if (in) {
                const uint64_t x = scratch[vi * GTS_TCC_K + a];
                nb = (uint32_t)x; fl = (uint32_t)(x >> 32);
              }
              const bool live = in && !bits_marked(fl);
              const bool sense = (fl & GTS_F_SENSE) != 0;
              const bool unv = live && vst[nb] == GIS_UNVISITED;
              if (unv) __hip_atomic_fetch_min(claim + nb, lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              const bool win = unv && claim[nb] == lane;
              const uint64_t mask = W::ballot(win);
              if (win) {
                queue[bn + W::popc_below(mask, lane)] = nb;
                vst[nb] = GIS_PROCESSED;
              }
              bn += W::popc(mask);
              const uint64_t bs = W::ballot(live && sense), ba = W::ballot(live && !sense);
              const uint64_t gmask = ((1ull << GTS_TCC_K) - 1ull) << (g * GTS_TCC_K);
              const bool is_term = a == 0 && g < nbat && !((bs & gmask) && (ba & gmask));
              const uint64_t tm = W::ballot(is_term);
              if (is_term) M.term[nterm + W::popc_below(tm, lane)] = cur;
              nterm += W::popc(tm);
              if (a == 0 && g < nbat) vst[cur] = GIS_VISITED;
              W::fence();
              i += nbat;
              continue;
            }
            const uint32_t cur = W::bcast(my_v, i), eb = W::bcast(my_eb, i), ee = W::bcast(my_ee, i);
            bool has_s = false, has_a = false;
            for (uint32_t base = eb; base < ee; base += W::WIDTH) {
              const uint32_t ce = base + lane;
              const bool in = ce < ee;
              uint32_t fl = 0, nb = 0;
              if (base == eb && lane < GTS_TCC_K) {
                const uint64_t x = scratch[i * GTS_TCC_K + lane];
                nb = (uint32_t)x; fl = (uint32_t)(x >> 32);
              } else if (in) {   /* (a list of more than GTS_TCC_K entries) */
                fl = edge_bits(ce); nb = M.cend[ce];
              }
              const bool live = in && !bits_marked(fl);
              const bool sense = (fl & GTS_F_SENSE) != 0;
              const bool unv = live && vst[nb] == GIS_UNVISITED;
              has_s |= W::ballot(live && sense) != 0;
              has_a |= W::ballot(live && !sense) != 0;
              const uint64_t mask = W::ballot(unv);
              if (unv) {
                queue[bn + W::popc_below(mask, lane)] = nb;
                vst[nb] = GIS_PROCESSED;
              }
              bn += W::popc(mask);
              W::fence();
            }
            if (!(has_s && has_a)) M.term[nterm++] = cur;
            if (lane == 0) vst[cur] = GIS_VISITED;
            W::fence();
            ++i;
          }
          bh += cnt;
        }
      }
    }
    ccoff[ncc] = nterm;
    W::fence();
    __builtin_amdgcn_s_waitcnt(0);
  }
#endif

  /* ORIENT: also assigns the strands as orient() does.  Only for a component
     whose compact edges are all live: then the terminal search and orient()
     are the same traversal (one cc, every edge followed), and one pass does
     for both.  oriented = false on a contradiction. */
  template <bool ORIENT>
  GTS_HD void calc_cc_t(bool &oriented)
  {
    const uint32_t lane = W::lane();
    oriented = ORIENT;
    for (uint32_t s = lane; s < nv; s += W::WIDTH) {
      if (!gts_vertex_is_marked(M.vst[s])) M.vst[s] = GIS_UNVISITED;
      if (ORIENT) M.gorient[s] = 0;
    }
    W::fence();
    auto ccoff = M.ccoff;
    nterm = 0; ncc = 0;
    /* Start vertices in index order: a chunk of states is looked at by all
       lanes at once and again after every search (which visits vertices of
       the chunk).  Inside a search the known part of the queue is read a
       chunk at a time together with the list bounds of its vertices, so the
       per-vertex chain is "edges, neighbour states" only. */
    for (uint32_t base0 = 0; base0 < nv; base0 += W::WIDTH) {
      uint32_t next_lane = 0;
      for (;;) {
        bool cand = false;
        if (base0 + lane < nv && lane >= next_lane) {
          const uint8_t st = M.vst[base0 + lane];
          cand = !gts_vertex_is_marked(st) && st != GIS_VISITED;
        }
        const uint64_t cm = W::ballot(cand);
        if (!cm) break;
        const uint32_t l0 = W::ctz(cm);
        next_lane = l0 + 1;
        const uint32_t s = base0 + l0;
        M.vst[s] = GIS_PROCESSED;
        M.queue[0] = s;
        if (ORIENT) {
          if (ncc > 0) oriented = false;   /* a second cc: not the case this is for */
          M.gorient[s] = 2;
        }
        ccoff[ncc++] = nterm;
        W::fence();
        uint32_t bh = 0, bn = 1;
        while (bh < bn) {
          const uint32_t cnt = bn - bh < W::WIDTH ? bn - bh : W::WIDTH;
          uint32_t my_v = 0, my_eb = 0, my_ee = 0, my_o = 0;
          if (lane < cnt) {
            my_v = M.queue[bh + lane];
            my_eb = M.coff[my_v] - M.e0; my_ee = M.coff[my_v + 1] - M.e0;
            if (ORIENT) my_o = M.gorient[my_v];
          }
          for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t cur = W::bcast(my_v, i), eb = W::bcast(my_eb, i), ee = W::bcast(my_ee, i);
            const bool ou = ORIENT && W::bcast(my_o, i) == 2;
            bool has_s = false, has_a = false;
            for (uint32_t base = eb; base < ee; base += W::WIDTH) {
              /* straight-line (see create_walk_clean): every lane loads, a lane
                 past the list reads the list's first edge */
              const uint32_t ce = base + lane;
              const bool in = ce < ee;
              const uint32_t cec = in ? ce : eb;
              const uint32_t fl = edge_bits(cec);
              const bool live = in && !bits_marked(fl);
              const bool sense = (fl & GTS_F_SENSE) != 0;
              const uint32_t nb = M.cend[cec];
              const bool unv = live && M.vst[nb] == GIS_UNVISITED;
              bool clash = false;
              if (ORIENT) {   /* as orient(): every compact edge, whatever the end's state */
                const uint32_t ov = (gts_next_dir((uint8_t)fl) != (sense != ou)) ? 2u : 1u;
                const uint32_t co = M.gorient[nb];
                clash = in && ((fl & GTS_F_UTURN) || nb == cur || (co != 0 && co != ov));
                if (in && co == 0) M.gorient[nb] = (uint8_t)ov;
              }
              if (ORIENT && W::ballot(clash)) oriented = false;
              has_s |= W::ballot(live && sense) != 0;
              has_a |= W::ballot(live && !sense) != 0;
              const uint64_t mask = W::ballot(unv);
              if (unv) {
                M.queue[bn + W::popc_below(mask, lane)] = nb;
                M.vst[nb] = GIS_PROCESSED;
              }
              bn += W::popc(mask);
              W::fence();
            }
            if (!(has_s && has_a)) M.term[nterm++] = cur;
            M.vst[cur] = GIS_VISITED;
            W::fence();
          }
          bh += cnt;
        }
      }
    }
    ccoff[ncc] = nterm;
    W::fence();
  }

  /* ---- ref algorithms.c:448-492, explicit stack.  Returns the compact edge
     that closes a cycle or GTS_NONE. ---- */
  GTS_HD uint32_t detect_cycle(uint32_t start, bool dir0, uint32_t &nvis)
  {
    const uint32_t lane = W::lane();
    uint32_t sp = 1;
    nvis = 0;
    M.st_v[0] = start; M.st_cur[0] = eoff(start);
    M.st_par[0] = (typename GtsCompMemT<LDS>::idx_t)GTS_NONE;   /* truncated in the packed layout */
    M.st_dir[0] = dir0 ? 1 : 0;
    M.visited[nvis++] = start;
    M.vst[start] = GIS_VISITED;
    W::fence();
    while (sp > 0) {
      const uint32_t f = sp - 1;
      const uint32_t v = W::uni(M.st_v[f]);
      /* the root's parent is GTS_NONE truncated to the index type: never a slot */
      const uint32_t par = W::uni(M.st_par[f]);
      const bool dir = W::uni((uint32_t)M.st_dir[f]) != 0;
      uint32_t cur = W::uni(M.st_cur[f]);
      const uint32_t ee = eoff(v + 1);
      bool descended = false;
      while (cur < ee) {
        const uint32_t ce = cur + lane;
        bool cand = false;
        uint32_t nb = 0, vs = 0, fl = 0;
        if (ce < ee) {
          fl = M.cflags[ce];
          nb = M.cend[ce];
          if (((fl & GTS_F_SENSE) != 0) == dir &&
              !edge_marked(ce) && nb != par) {
            vs = M.vst[nb];
            cand = !gts_vertex_is_marked((uint8_t)vs) && vs != GIS_PROCESSED;
          }
        }
        const uint64_t mask = W::ballot(cand);
        if (mask) {
          const uint32_t l = W::ctz(mask);
          const uint32_t nb_l = W::bcast(nb, l), vs_l = W::bcast(vs, l);
          const uint32_t fl_l = W::bcast(fl, l);
          if (vs_l == GIS_VISITED) return cur + l;      /* back edge */
          /* GIS_UNVISITED: descend */
          M.st_cur[f] = cur + l + 1;
          const uint32_t g = sp;
          M.st_v[g] = nb_l; M.st_par[g] = v; M.st_cur[g] = eoff(nb_l);
          M.st_dir[g] = gts_next_dir((uint8_t)fl_l) ? 1 : 0;
          ++sp;
          M.visited[nvis++] = nb_l;
          M.vst[nb_l] = GIS_VISITED;
          W::fence();
          descended = true;
          break;
        }
        cur += W::WIDTH;
      }
      if (!descended) {
        M.vst[v] = GIS_PROCESSED;
        --sp;
        W::fence();
      }
    }
    return GTS_NONE;
  }

  /* ---- ref algorithms.c:61-87 mark_vertex(v, GIS_CYCLIC) ---- */
  GTS_HD void mark_vertex_cyclic(uint32_t s)
  {
    const uint32_t lane = W::lane();
    const uint32_t v = C.slot_v[s0 + s];
    M.vst[s] = GIS_CYCLIC;
    C.G.vstate[v] = GIS_CYCLIC;
    const uint32_t b = C.G.row[v], e = C.G.row[v + 1];
    for (uint32_t p = b + lane; p < e; p += W::WIDTH) {
      const uint32_t t = C.G.twin[p];
      C.G.state[p] = GIS_CYCLIC;
      C.G.state[t] = GIS_CYCLIC;
      const uint32_t cp = C.cmap[p], ct = C.cmap[t];
      if (cp != GTS_NONE) {
        M.cstate[cp - e0g] = GIS_CYCLIC;
        M.cflags[cp - e0g] = (uint8_t)(M.cflags[cp - e0g] & ~GTS_F_TWINLIVE);
      }
      if (ct != GTS_NONE) {
        M.cstate[ct - e0g] = GIS_CYCLIC;
        M.cflags[ct - e0g] = (uint8_t)(M.cflags[ct - e0g] & ~GTS_F_TWINLIVE);
      }
    }
    W::fence();
  }

  /* the same on the working copy alone: every compact edge of the vertex and its
     twin -- the one edge of the end vertex' list that comes back (an edge is in
     the compact graph iff its twin is: k_live_union) -- turn CYCLIC.  Edges that
     are not in the compact graph change nothing the program looks at; they are
     marked in the global graph when the component is done (cyclic_marks_global) */
  GTS_HD void mark_vertex_cyclic_lds(uint32_t s)
  {
    const uint32_t lane = W::lane();
    M.vst[s] = GIS_CYCLIC;
    const uint32_t le = M.coff[s + 1] - M.e0;
    for (uint32_t ce = M.coff[s] - M.e0 + lane; ce < le; ce += W::WIDTH) {
      M.cstate[ce] = GIS_CYCLIC;
      M.cflags[ce] = (uint8_t)(M.cflags[ce] & ~GTS_F_TWINLIVE);
      const uint32_t w = M.cend[ce];
      const uint32_t we = M.coff[w + 1] - M.e0;
      for (uint32_t t = M.coff[w] - M.e0; t < we; ++t)
        if (M.cend[t] == s) {
          M.cstate[t] = GIS_CYCLIC;
          M.cflags[t] = (uint8_t)(M.cflags[t] & ~GTS_F_TWINLIVE);
          break;
        }
    }
    W::fence();
  }
  /* the global part of mark_vertex(v, GIS_CYCLIC), ref algorithms.c:61-87, for
     every vertex the working copy holds as CYCLIC (a vertex that was CYCLIC before
     the call is in no component) */
  GTS_HD void cyclic_marks_global()
  {
    const uint32_t lane = W::lane();
    for (uint32_t base = 0; base < nv; base += W::WIDTH) {
      const uint32_t s = base + lane;
      uint64_t cm = W::ballot(s < nv && M.vst[s < nv ? s : 0] == GIS_CYCLIC);
      while (cm) {
        const uint32_t l = W::ctz(cm);
        cm &= cm - 1;
        const uint32_t v = C.slot_v[s0 + base + l];
        const uint32_t b = C.G.row[v], e = C.G.row[v + 1];
        for (uint32_t p = b + lane; p < e; p += W::WIDTH) {
          C.G.state[p] = GIS_CYCLIC;
          C.G.state[C.G.twin[p]] = GIS_CYCLIC;
        }
      }
    }
  }

  /* the working copy's CYCLIC and SCAFFOLD marks into the global graph (vertex states
     go with run()'s write-back).  Only ever adds marks: calling it twice is harmless */
  GTS_HD void flush_local_marks()
  {
    const uint32_t lane = W::lane();
    cyclic_marks_global();
    if (any_scaffold_marks) {
#pragma unroll 1
      for (uint32_t k = lane; k < M.ne; k += W::WIDTH)
        if ((uint8_t)M.cstate[k] == GIS_SCAFFOLD) C.G.state[C.cgpos[e0g + k]] = GIS_SCAFFOLD;
    }
    W::fence();
  }

  /* ---- ref algorithms.c:495-578 ----
     keep_cc: the caller goes on with makescaffold, whose terminal search
     (algorithms.c:784) would repeat the one of the last pass.
     With strands assigned (orient) a DFS that starts where no cycle of D can
     be reached is futile -- it marks nothing and restores every state it
     touched (algorithms.c:551-554) -- and is skipped; D is peeled again after
     every mark.  Once D is acyclic the pass ends there and the component is
     clean for the walks (topo / tpos are those of the last peeling). */
  GTS_HD void removecycles(bool keep_cc) { removecycles_t<false>(keep_cc); }
  /* LOCAL: CYCLIC marks stay in the working copy (mark_vertex_cyclic_lds); the
     caller writes them to the global graph when the component is done (run_fast) */
  template <bool LOCAL>
  GTS_HD void removecycles_t(bool keep_cc)
  {
    const uint32_t lane = W::lane();
    /* every compact edge live (no marked edge kept for its live twin): the first
       terminal search assigns the strands on its way */
    bool all_live = C.fast_walks && nv > 1;
    for (uint32_t base = 0; base < M.ne && all_live; base += W::WIDTH) {
      const uint32_t ce = base + lane;
      all_live = W::ballot(ce < M.ne && edge_marked(ce)) == 0;
    }
    was_all_live = all_live;
    bool oriented = false, first = true;
    if (!all_live) oriented = C.fast_walks && nv > 1 && orient();
    bool found = true;
    clean = false;
    while (found) {
      found = false;
      if (first && all_live) calc_cc_t<true>(oriented);
      else calc_cc();
      first = false;
      if (oriented && peel()) { clean = true; break; }
      for (uint32_t s = lane; s < nv; s += W::WIDTH)
        if (!gts_vertex_is_marked(M.vst[s])) M.vst[s] = GIS_UNVISITED;
      W::fence();
      /* the ccs are visited in order; their boundaries do not matter here */
      for (uint32_t j = 0; j < nterm; ++j) {
        const uint32_t start = W::uni(M.term[j]);
        /* direction of the LAST unmarked edge, algorithms.c:533-538 */
        const uint32_t eb = eoff(start), ee = eoff(start + 1);
        bool set_dir = false, dir = true;
        for (uint32_t base = eb; base < ee; base += W::WIDTH) {
          const uint32_t ce = base + lane;
          bool live = false;
          uint32_t fl = 0;
          if (ce < ee) {
            live = !edge_marked(ce);
            fl = M.cflags[ce];
          }
          const uint64_t mask = W::ballot(live);
          if (mask) {
            dir = (W::bcast(fl, W::msb(mask)) & GTS_F_SENSE) != 0;
            set_dir = true;
          }
        }
        if (!set_dir) continue;
        if (gts_vertex_is_marked((uint8_t)W::uni(M.vst[start]))) continue;
        if (oriented) {
          const uint32_t g = W::uni((uint32_t)M.gorient[start]);
          const bool forward = dir == ((g & 3u) == 2);
          if (!(g & (forward ? 8u : 4u))) continue;   /* no cycle ahead */
        }
        uint32_t nvis = 0;
        const uint32_t back = detect_cycle(start, dir, nvis);
        for (uint32_t k = lane; k < nvis; k += W::WIDTH)
          M.vst[M.visited[k]] = GIS_UNVISITED;
        W::fence();
        if (back != GTS_NONE) {
          found = true;
          const uint32_t a = W::uni(M.cstart[back]), b = W::uni(M.cend[back]);
          if constexpr (LOCAL) { mark_vertex_cyclic_lds(a); mark_vertex_cyclic_lds(b); }
          else { mark_vertex_cyclic(a); mark_vertex_cyclic(b); }
          if (oriented) peel();
        }
      }
    }
    /* clean exit: the states are those of calc_cc; the reference ends with
       every unmarked vertex UNVISITED (algorithms.c:513-517, 551-554) */
    if (clean && !keep_cc) {
      for (uint32_t s = lane; s < nv; s += W::WIDTH)
        if (!gts_vertex_is_marked(M.vst[s])) M.vst[s] = GIS_UNVISITED;
      W::fence();
    }
  }

  /* one lane's relaxation of edge ce (start's seeding when unconditional,
     ref algorithms.c:671-677; otherwise algorithms.c:706-723).  q selects the
     participating lanes; they must have distinct end slots nb. */
  /* a queue entry: the edge; in the packed layout (edges below 2^16, vertices
     below 2^12) also its start vertex, which the pop would otherwise have to
     search for in the offsets (GtsLdsStartArr: nine dependent LDS reads) */
  GTS_HD uint32_t q_entry(uint32_t ce, uint32_t from) const { return LDS ? (ce | from << 16) : ce; }

  /* (called for the edges of the walk's start vertex: walk_from) */
  GTS_HD bool relax_distinct(bool q, uint32_t nb, uint32_t ce, float distance,
                             int64_t pushd, bool unconditional)
  {
    const uint32_t lane = W::lane();
    float old = 0.0f;
    if (q) old = M.distmap[nb];
    const bool imp = q && (unconditional || old == GTS_DIST_UNSET || old > distance);
    const bool fresh = imp && old == GTS_DIST_UNSET;
    const uint64_t im = W::ballot(imp), fm = W::ballot(fresh);
    const uint32_t ni = W::popc(im);
    if (qn + ni - qh > qcap) { err = GTS_CERR_WALKQ_OVERFLOW; return false; }
    if (imp) {
      M.distmap[nb] = distance;
      M.edgemap[nb] = ce;
      const uint64_t slot = qbase + ((qn + W::popc_below(im, lane)) & (qcap - 1));
      C.wq_edge[slot] = q_entry(ce, walk_from);
      C.wq_dist[slot] = pushd;
    }
    if (fresh) M.touched[ntouch + W::popc_below(fm, lane)] = nb;
    qn += ni;
    ntouch += W::popc(fm);
    W::fence();
    return true;
  }

  /* the same for lanes that may share an end slot (self loops only): one
     lane at a time, in list order */
  GTS_HD bool relax_ordered(bool q, uint32_t nb, uint32_t ce, float distance,
                            int64_t pushd, bool unconditional)
  {
    const uint32_t lane = W::lane();
    uint64_t qm = W::ballot(q);
    while (qm) {
      const uint32_t l = W::ctz(qm);
      qm &= qm - 1;
      if (!relax_distinct(q && lane == l, nb, ce, distance, pushd, unconditional))
        return false;
    }
    return true;
  }

  /* ---- ref algorithms.c:620-763.  Runs the label-correcting search from
     terminal `start`, evaluates the reached terminals and, if the best walk is
     longer than cc_len, stores it in cc_best (edge order as the reference:
     from the far terminal back to start).  Returns false on error. ---- */
  /* The ring of the walk's FIFO is carved from the pool on first use (rings are
     powers of two: positions are masked, not divided).  A walk that overflows
     its ring leaves the maps clean, takes a ring eight times as large and
     starts over; only when the pool has none left does the error reach the
     host, which runs the whole call again with a larger pool. */
  GTS_HD bool create_walk_reference(uint32_t start, uint64_t &cc_len, uint32_t &cc_n)
  {
    for (;;) {
      if (qcap == 0) {
        uint64_t need = 64;
        while (need < C.wq_factor * (uint64_t)M.ne + 64) need <<= 1;
        const uint64_t off = W::alloc(C.wq_used, need);
        if (off + need > C.wq_pool) { err = GTS_CERR_WALKQ_OVERFLOW; return false; }
        qbase = off; qcap = need;
      }
      const uint64_t len0 = cc_len;
      const uint32_t n0 = cc_n;
      bool done;
      if constexpr (SPLIT) done = M.sarc ? create_walk_reference_once<true>(start, cc_len, cc_n)
                                         : create_walk_reference_once<false>(start, cc_len, cc_n);
      else done = create_walk_reference_once<false>(start, cc_len, cc_n);
      if (done) return true;
      if (err != GTS_CERR_WALKQ_OVERFLOW) return false;
      const uint64_t need = qcap * 8;
      const uint64_t off = W::alloc(C.wq_used, need);
      if (off + need > C.wq_pool) return false;
      qbase = off; qcap = need; err = 0;
      cc_len = len0; cc_n = n0;
    }
  }

  /* SV: the split view of the arcs (GtsCompMemT::sarc) is there */
  template <bool SV>
  GTS_HD bool create_walk_reference_once(uint32_t start, uint64_t &cc_len, uint32_t &cc_n)
  {
    const uint32_t lane = W::lane();
    qh = 0; qn = 0; ntouch = 0;
    walk_from = start;
    uint32_t nwt = 0;
    bool ok = true;
    /* lastpop may share its storage with the labels of the linear walks */
    for (uint32_t s = lane; s < nv; s += W::WIDTH) M.lastpop[s] = 0;
    if constexpr (SV) {
      /* which vertices are terminals (algorithms.c:694: no live edge in one of
         the two senses) does not change during a walk: once per vertex here
         instead of once per popped node from the node's own arcs */
      for (uint32_t s = lane; s < nv; s += W::WIDTH) {
        const uint32_t b = M.coff[s], m = M.smid[s], e = M.coff[s + 1];
        bool hs = false, ha = false;
        for (uint32_t k = b; k < m; ++k) hs |= !gts_edge_is_marked(M.cstate[M.sarc[k]]);
        for (uint32_t k = m; k < e; ++k) ha |= !gts_edge_is_marked(M.cstate[M.sarc[k]]);
        M.tight[s] = (hs && ha) ? 0 : 1;
      }
    }
    W::fence();
    /* seed with the start's live edges, algorithms.c:661-679 */
    {
      const uint32_t eb = eoff(start), ee = eoff(start + 1);
      for (uint32_t base = eb; base < ee && ok; base += W::WIDTH) {
        const uint32_t ce = base + lane;
        bool live = false;
        uint32_t nb = 0;
        int64_t d = 0;
        if (ce < ee) {
          live = !edge_marked(ce);
          nb = M.cend[ce];
          d = dist_of(ce);
        }
        if (W::popc(W::ballot(live && nb == start)) >= 2)
          ok = relax_ordered(live, nb, ce, (float)d, d, true);
        else
          ok = relax_distinct(live, nb, ce, (float)d, d, true);
      }
    }
    /* Main loop, algorithms.c:681-728, vectorised without changing its
       sequential meaning.  The FIFO is consumed in order; the out-arcs of
       consecutive queued nodes are laid out across the lanes (node after node,
       arc after arc: exactly the order in which the reference relaxes them).
       A candidate succeeds iff it beats the label its target had before the
       step AND every earlier candidate of the step for the same target (a
       per-target prefix minimum over the lanes holding that target); every
       success pushes a node, in lane order, and the last success per target
       leaves label and edgemap.  Nodes pushed during a step lie behind the
       nodes it consumes, as in the reference. */
    uint64_t pops = 0;
    uint64_t wbase = 0, wend = 0;      /* queue window held in registers */
    uint32_t w_edge = 0;
    int64_t w_dist = 0;
    uint32_t cur_off = 0;              /* arcs of node qh already relaxed */
    bool carry_s = false, carry_a = false;
    const uint32_t tbits = 32u - W::clz32(nv > 1 ? nv - 1 : 1);
    while (ok && qh < qn) {
      if (qh == wend || (wend - qh < W::WIDTH / 4 && qn > wend)) {
        const uint64_t cnt = qn - qh < W::WIDTH ? qn - qh : W::WIDTH;
        if (lane < cnt) {
          const uint64_t slot = qbase + ((qh + lane) & (qcap - 1));
          w_edge = C.wq_edge[slot];
          w_dist = C.wq_dist[slot];
        }
        wbase = qh; wend = qh + cnt;
      }
      /* node l of this step = queue entry qh + l */
      const uint32_t shift = (uint32_t)(qh - wbase);
      const uint32_t navail = (uint32_t)(wend - qh);
      const uint32_t src = (lane + shift) & (W::WIDTH - 1);
      const uint32_t pw = W::shfl(w_edge, src);
      const uint32_t pe = LDS ? (pw & 0xFFFFu) : pw;
      const int64_t nd = (int64_t)W::shfl64((uint64_t)w_dist, src);
      uint32_t endv = 0, from = 0, eb = 0, deg = 0;
      bool dir = false;
      if (lane < navail) {
        endv = M.cend[pe];
        if constexpr (LDS) from = pw >> 16; else from = M.cstart[pe];
        dir = gts_next_dir(M.cflags[pe]);
        eb = M.coff[endv] - M.e0;
        uint32_t ee = M.coff[endv + 1] - M.e0;
        if constexpr (SV) {      /* only the arcs that leave in the node's direction */
          const uint32_t mid = M.smid[endv];
          if (dir) ee = mid; else eb = mid;
        }
        if (lane == 0) eb += cur_off;
        deg = ee - eb;
      }
      /* prefix sums of the arc counts.  A step takes at most WIDTH arcs, so
         counts are clamped to WIDTH + 1 (keeps "does not fit" visible) and the
         sum is built from one ballot per bit */
      const uint32_t degc = deg > W::WIDTH ? W::WIDTH + 1 : deg;
      const uint32_t incl = W::scan_incl_small(degc), excl = incl - degc;
      const uint32_t total = W::bcast(incl, W::WIDTH - 1);
      const uint32_t take = total < W::WIDTH ? total : W::WIDTH;
      /* arc lane a -> node r: the last node that starts at or before a */
      const uint32_t nstart = W::popc(W::ballot(lane < navail && excl < take));
      uint32_t r = 0;
      for (uint32_t k = 1; k < nstart; ++k)
        if (lane >= W::bcast(excl, k)) r = k;
      const bool act = lane < take;
      const uint32_t rr = act ? r : 0;
      const uint32_t r_eb = W::shfl(eb, rr), r_excl = W::shfl(excl, rr);
      const uint32_t r_from = W::shfl(from, rr);
      const uint32_t r_endv = W::shfl(endv, rr);
      const bool r_dir = W::shfl((uint32_t)dir, rr) != 0;
      const int64_t r_nd = (int64_t)W::shfl64((uint64_t)nd, rr);
      bool live = false, sense = false, q = false;
      uint32_t nb = 0, ce = 0;
      float distance = 0.0f, old = 0.0f;
      if (act) {
        ce = r_eb + (lane - r_excl);
        if constexpr (SV) ce = M.sarc[ce];
        live = !edge_marked(ce);
        sense = SV ? r_dir : (M.cflags[ce] & GTS_F_SENSE) != 0;
        nb = M.cend[ce];
        q = live && sense == r_dir && nb != r_from && !gts_vertex_is_marked(M.vst[nb]);
        distance = (float)(r_nd + dist_of(ce));
        if (q) old = M.distmap[nb];
      }
      const uint64_t bs = W::ballot(act && live && sense);
      const uint64_t ba = W::ballot(act && live && !sense);
      /* same-target lanes */
      uint64_t peers = W::ballot(q);
      for (uint32_t bit = 0; bit < tbits; ++bit) {
        const bool one = (nb >> bit) & 1u;
        const uint64_t bm = W::ballot(q && one);
        peers &= one ? bm : ~bm;
      }
      const uint64_t lt = W::lanemask_lt(lane);
      float minprev = GTS_DIST_UNSET;
      {
        uint64_t lower = q ? (peers & lt) : 0;
        /* every lane walks its own (short) list of earlier same-target lanes;
           the shuffles are executed by all lanes */
        uint64_t any = W::ballot(lower != 0);
        while (any) {
          const uint32_t m = lower ? W::ctz(lower) : 0;
          const float cm = W::shflf(distance, m);
          if (lower) { if (cm < minprev) minprev = cm; lower &= lower - 1; }
          any = W::ballot(lower != 0);
        }
      }
      const bool imp = q && (old == GTS_DIST_UNSET || old > distance) && distance < minprev;
      const uint64_t im = W::ballot(imp);
      const uint32_t ni = W::popc(im);
      if (qn + ni - qh > qcap) { err = GTS_CERR_WALKQ_OVERFLOW; ok = false; break; }
      const bool last = imp && (peers & im & ~lt & ~(1ull << lane)) == 0;
      const bool fresh = imp && old == GTS_DIST_UNSET && (peers & im & lt) == 0;
      const uint64_t fm = W::ballot(fresh);
      if (imp) {
        const uint64_t slot = qbase + ((qn + W::popc_below(im, lane)) & (qcap - 1));
        C.wq_edge[slot] = q_entry(ce, r_endv);
        C.wq_dist[slot] = (int64_t)distance;
      }
      if (last) { M.distmap[nb] = distance; M.edgemap[nb] = ce; }
      if (fresh) M.touched[ntouch + W::popc_below(fm, lane)] = nb;
      qn += ni;
      ntouch += W::popc(fm);
      /* nodes whose arcs were all relaxed in this step are popped now */
      const bool done = lane < navail && incl <= take && deg == degc;
      const uint32_t ndone = W::popc(W::ballot(done));
      bool term = false;
      if (done) {
        if constexpr (SV) term = M.tight[endv] != 0;
        else {
          const uint64_t mine = W::range_mask(excl, incl);
          const bool hs = (bs & mine) != 0 || (lane == 0 && carry_s);
          const bool ha = (ba & mine) != 0 || (lane == 0 && carry_a);
          term = !(hs && ha);          /* algorithms.c:694 */
        }
      }
      uint32_t prev = 1;
      if (term)                      /* remember the LAST pop of a terminal */
        prev = W::atomic_max(&M.lastpop[endv], (uint32_t)(pops + lane + 1));
      const uint64_t nm = W::ballot(term && prev == 0);
      if (term && prev == 0) M.wterm[nwt + W::popc_below(nm, lane)] = endv;
      nwt += W::popc(nm);
      if (ndone < navail && take > W::bcast(excl, ndone < W::WIDTH ? ndone : 0)) {
        /* the next node is partly relaxed: keep its terminal flags and cursor */
        const uint32_t pex = W::bcast(excl, ndone);
        const uint64_t part = W::range_mask(pex, take);
        const bool first = ndone == 0;
        carry_s = (bs & part) != 0 || (first && carry_s);
        carry_a = (ba & part) != 0 || (first && carry_a);
        cur_off = (first ? cur_off : 0) + (take - pex);
      } else {
        carry_s = carry_a = false;
        cur_off = 0;
      }
      qh += ndone;
      pops += ndone;
      if (pops > C.max_pops) { err = GTS_CERR_WALK_LOOP; ok = false; break; }
      W::fence();
    }
    npops += pops;
    /* evaluate the reached terminals, algorithms.c:732-756: the reference pops
       them from the back and keeps a strictly longer walk, i.e. it returns the
       longest walk and, among equals, the terminal popped last.  Each lane
       back-tracks its own terminals. */
    uint64_t best_len = 0;
    uint32_t best_pop = 0, best_t = GTS_NONE;
    if (ok) {
      const uint32_t limit = nv + 1;
      bool loop_err = false;
      for (uint32_t k = lane; k < nwt; k += W::WIDTH) {
        const uint32_t t = M.wterm[k];
        uint64_t len = (uint64_t)M.cseq[start];
        uint32_t cv = t, steps = 0;
        while (cv != start) {
          const uint32_t re = M.edgemap[cv];
          len += (uint64_t)M.cseq[cv];
          cv = M.cstart[re];
          if (++steps > limit) { loop_err = true; break; }
        }
        const uint32_t lp = M.lastpop[t];
        if (len > best_len || (len == best_len && len > 0 && lp > best_pop)) {
          best_len = len; best_pop = lp; best_t = t;
        }
      }
      if (W::ballot(loop_err)) { err = GTS_CERR_WALK_LOOP; ok = false; }
      /* wave arg-max on (len, lastpop) */
      for (uint32_t off = W::WIDTH / 2; off > 0; off >>= 1) {
        const uint64_t ol = W::shfl64(best_len, lane ^ off);
        const uint32_t op = W::shfl(best_pop, lane ^ off);
        const uint32_t ot = W::shfl(best_t, lane ^ off);
        if (ol > best_len || (ol == best_len && op > best_pop)) {
          best_len = ol; best_pop = op; best_t = ot;
        }
      }
      best_len = W::uni64((int64_t)best_len);
      best_t = W::uni(best_t);
    }
    /* makescaffold keeps the first strictly longer walk, algorithms.c:826-832 */
    if (ok && best_t != GTS_NONE && best_len > cc_len) {
      uint32_t cv = best_t, n = 0;
      while (cv != start) {
        const uint32_t re = W::uni(M.edgemap[cv]);
        M.cc_best[n++] = re;
        cv = W::uni(M.cstart[re]);
      }
      cc_len = best_len;
      cc_n = n;
      W::fence();
    }
    /* leave the maps clean for the next walk */
    for (uint32_t k = lane; k < ntouch; k += W::WIDTH) {
      const uint32_t v = M.touched[k];
      M.distmap[v] = GTS_DIST_UNSET;
      note_labelled(v);
    }
    for (uint32_t k = lane; k < nwt; k += W::WIDTH)
      M.lastpop[M.wterm[k]] = 0;
    if constexpr (SV)
      for (uint32_t s = lane; s < nv; s += W::WIDTH) M.tight[s] = 0;
    W::fence();
    return ok;
  }


  /* ---- linear-time equivalent of create_walk for the common case ----
     The reference's search (algorithms.c:681-728) is a FIFO label-correcting
     relaxation: on scaffolds with multi-hop links every vertex is re-queued
     once per hop count, O(L^2) pops for a chain of L contigs.  Its RESULT is
     order-independent whenever
       (a) the (vertex, leaving direction) states reachable from the start form
           a DAG in which every vertex occurs with ONE direction (then the twin
           exclusion of algorithms.c:702 never applies either),
       (b) ties -- several in-arcs attaining a vertex' final label (edgemap
           keeps the FIRST arc that sets it), several reached terminals with
           the longest tree path (the reference keeps the one popped last) --
           are resolved by the FIFO's push order of final nodes, which is
           known in closed form (pushed_after) as long as every label stays
           below 2^24 in magnitude, where the float sums are exact.
     The final label is min over paths of the nested float roundings
     fl(int(label) + dist); fl is monotone, so a relaxation in topological
     order computes it with the very same operations.  Any violated condition
     returns false with the scratch restored, and the caller runs the
     reference's search instead. */
  /* Order in which the reference pushes the FINAL queue nodes of two reached
     vertices a != b (exact arithmetic, labels final, tree = first-arrival
     tree given by edgemap/depth): the FIFO handles whole generations one after
     the other, a final node's generation is its tree depth, and inside a
     generation nodes keep the order of their parents and, under one parent,
     the order of the parent's adjacency list.  Returns true if a's final node
     is pushed (hence popped) after b's. */
  GTS_HD bool pushed_after(uint32_t a, uint32_t b, uint32_t start) const
  {
    const uint32_t da = a == start ? 0 : W::uni(M.st_par[a]);
    const uint32_t db = b == start ? 0 : W::uni(M.st_par[b]);
    if (da != db) return da > db;
    uint32_t ea = W::uni(M.edgemap[a]), eb = W::uni(M.edgemap[b]);
    uint32_t pa = W::uni(M.par[a]), pb = W::uni(M.par[b]);
    while (pa != pb) {
      ea = W::uni(M.edgemap[pa]); eb = W::uni(M.edgemap[pb]);
      pa = W::uni(M.par[pa]); pb = W::uni(M.par[pb]);
    }
    return ea > eb;
  }


  /* Reachable states with a cycle (single direction per vertex, no u-turn
     arcs, start not re-entered: checked by pass 1 of create_walk_fast).
     In exact arithmetic the reference's search ends with the shortest-path
     labels whatever the order, only a value equal to the final label of v --
     it can only come from a node carrying the final label of its source --
     can fix edgemap[v], and the final node of v is pushed when the final node
     of its first tight source is popped.  Hence edgemap and the pop order of
     the final nodes are those of a FIFO search over the tight arcs, adjacency
     lists in list order.  Labels: queue relaxation with an in-queue flag
     (bounded pops, gives up on negative cycles). */
  GTS_HD bool walk_cyclic(uint32_t start, uint32_t nr, uint32_t npeeled, uint64_t &best_len,
                          uint32_t &best_t)
  {
    const uint32_t lane = W::lane();
    auto R = M.queue;
    auto TQ = M.visited;
    auto BQ = M.wterm;
    auto orient = M.st_dir;
    auto dirty = M.tight;
    auto indeg = M.st_v;
    /* Relaxation order: the topological prefix create_walk_fast peeled, then
       the states it got stuck on in the order they were reached.  The labels
       are the fixpoint whatever the order; with this one a sweep settles
       everything up to the next arc that points backwards. */
    for (uint32_t base = 0; base < nr; base += W::WIDTH) {
      const uint32_t k = base + lane;
      uint32_t v = 0;
      bool stuck = false;
      if (k < nr) { v = R[k]; stuck = indeg[v] != 0; }
      const uint64_t sm = W::ballot(stuck);
      if (stuck) TQ[npeeled + W::popc_below(sm, lane)] = v;
      npeeled += W::popc(sm);
    }
    for (uint32_t k = lane; k < nr; k += W::WIDTH) {
      const uint32_t v = R[k];
      M.distmap[v] = GTS_DIST_UNSET;
      dirty[v] = 0;
    }
    W::fence();
    if (npeeled != nr) return false;
    bool inexact = false, bad = false, changed = true;
    uint64_t pops = 0;
    const uint64_t max_pops = 64ull * nr + 64;
    dirty[start] = 1;
    M.nd[start] = 0;
    W::fence();
    while (changed && !bad) {
      changed = false;
      for (uint32_t base = 0; base < nr && !bad; base += W::WIDTH) {
        const uint32_t k = base + lane;
        uint32_t cv = 0;
        if (k < nr) cv = TQ[k];
        /* a vertex handled in this chunk may label a later one of the chunk */
        uint32_t next_lane = 0;
        while (!bad) {
          const uint64_t dm = W::ballot(k < nr && lane >= next_lane && dirty[cv] != 0);
          if (!dm) break;
          const uint32_t l = W::ctz(dm);
          next_lane = l + 1;
          const uint32_t u = W::bcast(cv, l);
          if (++pops > max_pops) { bad = true; break; }
          dirty[u] = 0;
          const bool du = (W::uni((uint32_t)orient[u]) & 3u) == 2;
          const nd_t ndu = uni_t(M.nd[u]);   /* nd[start] = 0 */
          const uint32_t eb = eoff(u), ee = eoff(u + 1);
          for (uint32_t eb2 = eb; eb2 < ee; eb2 += W::WIDTH) {
            const uint32_t ce = eb2 + lane;
            if (ce < ee && !edge_marked(ce) &&
                ((M.cflags[ce] & GTS_F_SENSE) != 0) == du) {
              const uint32_t v = M.cend[ce];
              const nd_t w = (nd_t)dist_of(ce);
              const float cand = (float)(ndu + w);
              const float old = M.distmap[v];
              if (!(cand > -16777216.0f && cand < 16777216.0f)) inexact = true;
              if (old == GTS_DIST_UNSET || old > cand) {
                M.distmap[v] = cand;
                M.nd[v] = u == start ? w : (nd_t)cand;
                dirty[v] = 1;
              }
            }
            W::fence();
          }
        }
      }
      /* labels that moved behind the sweep: once more */
      for (uint32_t base = 0; base < nr && !changed; base += W::WIDTH) {
        const uint32_t k = base + lane;
        changed = W::ballot(k < nr && dirty[TQ[k]] != 0) != 0;
      }
    }
    npops += pops;
    if (W::ballot(inexact)) bad = true;
    if (bad) return false;
    /* FIFO search over the tight arcs */
    uint32_t bh = 0, bt = 1;
    best_len = 0; best_t = GTS_NONE;
    BQ[0] = start;
    M.plen[start] = (len_t)M.cseq[start];
    W::fence();
    while (bh < bt) {
      const uint32_t u = W::uni(BQ[bh]);
      ++bh;
      const bool du = (W::uni((uint32_t)orient[u]) & 3u) == 2;
      const nd_t ndu = uni_t(M.nd[u]);   /* nd[start] = 0 */
      const len_t plu = uni_t(M.plen[u]);
      const uint32_t eb = eoff(u), ee = eoff(u + 1);
      bool us = false, ua = false;
      for (uint32_t base = eb; base < ee; base += W::WIDTH) {
        const uint32_t ce = base + lane;
        bool live = false, sense = false, take = false;
        uint32_t v = 0;
        if (ce < ee) {
          live = !edge_marked(ce);
          sense = (M.cflags[ce] & GTS_F_SENSE) != 0;
          if (live && sense == du) {
            v = M.cend[ce];
            const nd_t w = (nd_t)dist_of(ce);
            const float cand = (float)(ndu + w);
            take = cand == M.distmap[v] && !(orient[v] & 4u);
          }
        }
        us |= W::ballot(live && sense) != 0;
        ua |= W::ballot(live && !sense) != 0;
        const uint64_t tm = W::ballot(take);
        if (take) {
          orient[v] = (uint8_t)(orient[v] | 4u);
          M.edgemap[v] = ce;
          M.par[v] = u;
          M.plen[v] = (len_t)(plu + (len_t)M.cseq[v]);
          BQ[bt + W::popc_below(tm, lane)] = v;
        }
        bt += W::popc(tm);
        W::fence();
      }
      /* later final node = later pop: on equal length the later terminal wins */
      if (u != start && !(us && ua) && plu >= best_len && plu > 0) { best_len = plu; best_t = u; }
    }
    return bt == nr;
  }

  GTS_HD bool create_walk_fast(uint32_t start, uint64_t &cc_len, uint32_t &cc_n)
  {
    const uint32_t lane = W::lane();
    auto R = M.queue;
    auto TQ = M.visited;
    auto indeg = M.st_v;
    auto orient = M.st_dir;
    auto depth = M.st_par;
    /* all live edges of the start must leave in one direction */
    bool has_s = false, has_a = false;
    {
      const uint32_t eb = eoff(start), ee = eoff(start + 1);
      for (uint32_t base = eb; base < ee; base += W::WIDTH) {
        const uint32_t ce = base + lane;
        bool live = false, sense = false;
        if (ce < ee) {
          live = !edge_marked(ce);
          sense = (M.cflags[ce] & GTS_F_SENSE) != 0;
        }
        has_s |= W::ballot(live && sense) != 0;
        has_a |= W::ballot(live && !sense) != 0;
      }
    }
    if (has_s && has_a) { W::count(C.why + 0); return false; }
    if (!has_s && !has_a) return true;          /* nothing reachable: empty walk */
    /* pass 1: reachable states, in-degrees */
    uint32_t nr = 1, rh = 0;
    bool bad = false;
    R[0] = start;
    orient[start] = has_s ? 2 : 1;              /* direction + 1 */
    indeg[start] = 0;
    W::fence();
    while (rh < nr && !bad) {
      const uint32_t u = W::uni(R[rh]);
      ++rh;
      const bool du = W::uni((uint32_t)orient[u]) == 2;
      const uint32_t eb = eoff(u), ee = eoff(u + 1);
      for (uint32_t base = eb; base < ee && !bad; base += W::WIDTH) {
        const uint32_t ce = base + lane;
        bool arc = false, fresh = false, clash = false;
        uint32_t v = 0, od = 0;
        if (ce < ee) {
          const uint32_t fl = M.cflags[ce];
          arc = !edge_marked(ce) && ((fl & GTS_F_SENSE) != 0) == du;
          if (arc) {
            v = M.cend[ce];
            od = gts_next_dir((uint8_t)fl) ? 2u : 1u;
            const uint32_t ov = orient[v];
            clash = v == u || v == start || gts_vertex_is_marked(M.vst[v]) ||
                    (ov != 0 && ov != od) || (fl & GTS_F_UTURN);
            fresh = ov == 0;
          }
        }
        const uint64_t cm = W::ballot(clash);
        if (cm) {
          const uint32_t l = W::ctz(cm), cv = W::bcast(v, l);
          const bool ut = (W::bcast((uint32_t)(ce < ee ? M.cflags[ce] : 0), l) & GTS_F_UTURN) != 0;
          W::count(C.why + (cv == u || ut ? 1 : cv == start ? 2
                            : gts_vertex_is_marked((uint8_t)W::uni(M.vst[cv])) ? 3 : 4));
          bad = true; break;
        }
        const uint64_t fm = W::ballot(arc && fresh);
        if (arc) {
          if (fresh) {
            orient[v] = (uint8_t)od;
            indeg[v] = 1;
            R[nr + W::popc_below(fm, lane)] = v;
          } else
            indeg[v] = indeg[v] + 1;
        }
        nr += W::popc(fm);
        W::fence();
      }
    }
    /* pass 2: relaxation in topological order */
    uint32_t nq = 1, qh2 = 0, processed = 0, best_t = GTS_NONE;
    uint64_t best_len = 0;
    bool inexact = false;     /* a label left the range where floats are exact */
    if (!bad) {
      TQ[0] = start;
      M.plen[start] = (len_t)M.cseq[start];
      M.nd[start] = 0;      /* no conditional loads in the loop: they would be fetched one by one */
      depth[start] = 0;
      W::fence();
      while (qh2 < nq && !bad) {
        const uint32_t u = W::uni(TQ[qh2]);
        ++qh2; ++processed;
        const bool du = W::uni((uint32_t)orient[u]) == 2;
        const nd_t ndu = uni_t(M.nd[u]);   /* nd[start] = 0 */
        const len_t plu = uni_t(M.plen[u]);
        const uint32_t dpu = W::uni(depth[u]);   /* depth[start] = 0 */
        const uint32_t eb = eoff(u), ee = eoff(u + 1);
        bool us = false, ua = false;
        for (uint32_t base = eb; base < ee && !bad; base += W::WIDTH) {
          const uint32_t ce = base + lane;
          bool live = false, sense = false, arc = false, ready = false, tie = false;
          uint32_t v = 0;
          if (ce < ee) {
            live = !edge_marked(ce);
            sense = (M.cflags[ce] & GTS_F_SENSE) != 0;
            arc = live && sense == du;
            if (arc) {
              v = M.cend[ce];
              const nd_t w = (nd_t)dist_of(ce);
              const float cand = (float)(ndu + w);
              const float old = M.distmap[v];
              if (!(cand > -16777216.0f && cand < 16777216.0f)) inexact = true;
              if (old == GTS_DIST_UNSET || old > cand) {
                M.distmap[v] = cand;
                M.edgemap[v] = ce;
                M.par[v] = u;
                M.nd[v] = u == start ? w : (nd_t)cand;
                M.plen[v] = (len_t)(plu + (len_t)M.cseq[v]);
                depth[v] = dpu + 1;
              } else if (old == cand)
                tie = true;
              const uint32_t d = indeg[v] - 1;
              indeg[v] = d;
              ready = d == 0;
            }
          }
          us |= W::ballot(live && sense) != 0;
          ua |= W::ballot(live && !sense) != 0;
          /* two in-arcs attain the label of v: edgemap keeps the one whose
             value arrived first (algorithms.c:711-717), i.e. the arc whose
             source's final node is pushed first */
          uint64_t tm = W::ballot(tie);
          /* (inexact is kept per lane; the wave is asked only where it matters) */
          if (tm && W::ballot(inexact)) { W::count(C.why + 5); bad = true; break; }
          while (tm) {
            const uint32_t l = W::ctz(tm);
            tm &= tm - 1;
            const uint32_t tv = W::bcast(v, l), tce = W::bcast(ce, l);
            const uint32_t up = W::uni(M.par[tv]);
            if (pushed_after(up, u, start)) {       /* u's value came first */
              M.edgemap[tv] = tce;
              M.par[tv] = u;
              M.plen[tv] = (len_t)(plu + (len_t)uni_t(M.cseq[tv]));
              depth[tv] = dpu + 1;
              W::fence();
            }
          }
          const uint64_t rm = W::ballot(ready);
          if (ready) TQ[nq + W::popc_below(rm, lane)] = v;
          nq += W::popc(rm);
          W::fence();
        }
        /* reached terminal (algorithms.c:694): candidate end of the walk; among
           equally long walks the reference keeps the terminal popped last
           (algorithms.c:732-756), whose last pop is its final node */
        if (!bad && u != start && !(us && ua)) {
          if (plu > best_len) { best_len = plu; best_t = u; }
          else if (plu == best_len && best_t != GTS_NONE) {
            if (W::ballot(inexact)) { W::count(C.why + 7); bad = true; }
            else if (pushed_after(u, best_t, start)) best_t = u;
          }
        }
      }
      if (!bad && processed != nr) {
        /* the reachable states hold a cycle: labels by a queue-based
           relaxation, tie-breaks by a FIFO search over the tight arcs */
        if (!walk_cyclic(start, nr, processed, best_len, best_t)) { W::count(C.why + 6); bad = true; }
      }
    }
    if (!bad && best_t != GTS_NONE && best_len > cc_len) {
      uint32_t cv = best_t, n = 0;
      while (cv != start) {
        const uint32_t re = W::uni(M.edgemap[cv]);
        M.cc_best[n++] = re;
        cv = W::uni(M.par[cv]);
      }
      cc_len = best_len;
      cc_n = n;
    }
    /* restore the scratch */
    for (uint32_t k = lane; k < nr; k += W::WIDTH) {
      const uint32_t v = R[k];
      orient[v] = 0;
      M.distmap[v] = GTS_DIST_UNSET;
      M.tight[v] = 0;
      note_labelled(v);
    }
    W::fence();
    return !bad;
  }


  /* ---- whole-component analysis ------------------------------------------
     A scaffold graph is bidirected: a walk enters a contig at one end and
     leaves at the other.  If the component has a consistent strand assignment
     o(v) -- every compact edge (u -> v) relates the two strands by
     o(v) = next_dir(e) xor (sense(e) != o(u)) without contradiction -- then
     every traversal the reference starts (DFS of removecycles, search of
     create_walk) stays on ONE sheet: the forward sheet (arcs with
     sense == o(u)) or its mirror image.  If in addition no edge is a u-turn
     and the forward sheet is acyclic, then
       * no DFS of removecycles can meet a vertex on its stack: cycle removal
         marks nothing and is reduced to its terminal search;
       * one order serves every walk, which becomes a single sweep: edge
         states are not symmetric, so the mirror sheet (live arcs with
         sense != o(u)) is not the transpose of the forward sheet; the order
         is a topological order of D = forward arcs + reversed mirror arcs
         (from x's list: edges with sense == o(x) that are live or whose twin
         is), swept upwards on the forward sheet and downwards on the mirror.
     The analysis costs two passes over the component.  It survives the
     SCAFFOLD marks of makescaffold: a marked twin that turns SCAFFOLD is a new
     live arc of one sheet, but D already holds it (its twin was live). */
  /* strands by a search over all compact edges, whatever their state.  False
     on a contradiction, a u-turn arc or a self loop. */
  GTS_HD bool orient()
  {
    const uint32_t lane = W::lane();
    auto Q = M.queue;
    for (uint32_t s = lane; s < nv; s += W::WIDTH) M.gorient[s] = 0;
    W::fence();
    uint32_t qh2 = 0, qn2 = 1;
    bool bad = false;
    Q[0] = 0;
    M.gorient[0] = 2;
    W::fence();
    while (qh2 < qn2 && !bad) {
      /* the known part of the queue, a chunk at a time (see calc_cc) */
      const uint32_t cnt = qn2 - qh2 < W::WIDTH ? qn2 - qh2 : W::WIDTH;
      uint32_t my_u = 0, my_eb = 0, my_ee = 0, my_o = 0;
      if (lane < cnt) {
        my_u = Q[qh2 + lane];
        my_eb = M.coff[my_u] - M.e0; my_ee = M.coff[my_u + 1] - M.e0;
        my_o = M.gorient[my_u];
      }
      for (uint32_t i = 0; i < cnt && !bad; ++i) {
        const uint32_t u = W::bcast(my_u, i), eb = W::bcast(my_eb, i), ee = W::bcast(my_ee, i);
        const bool ou = W::bcast(my_o, i) == 2;
        for (uint32_t base = eb; base < ee; base += W::WIDTH) {
          const uint32_t ce = base + lane;
          const bool in = ce < ee;
          const uint32_t cec = in ? ce : eb;
          const uint32_t fl = edge_bits(cec);
          const bool sense = (fl & GTS_F_SENSE) != 0;
          const uint32_t v = M.cend[cec];
          const uint32_t ov = (gts_next_dir((uint8_t)fl) != (sense != ou)) ? 2u : 1u;
          const uint32_t cur = M.gorient[v];
          const bool clash = in && ((fl & GTS_F_UTURN) || v == u || (cur != 0 && cur != ov));
          const bool fresh = in && cur == 0;
          if (W::ballot(clash)) { bad = true; break; }
          const uint64_t fm = W::ballot(fresh);
          if (fresh) {
            M.gorient[v] = (uint8_t)ov;
            Q[qn2 + W::popc_below(fm, lane)] = v;
          }
          qn2 += W::popc(fm);
          W::fence();
        }
      }
      qh2 += cnt;
    }
    return !bad && qn2 == nv;
  }

  /* arc of D seen from its tail x (forward) or from its head (!forward) */
  GTS_HD bool d_arc(uint32_t ce, bool ox, bool forward) const
  {
    const uint32_t fl = edge_bits(ce);
    return (((fl & GTS_F_SENSE) != 0) == ox) == forward &&
           (!bits_marked(fl) || (fl & GTS_F_TWINLIVE));
  }

  /* Topological order of D for an oriented component (topo / tpos); true if
     D is acyclic.  Otherwise gorient gets, next to the strand, bit 2 for the
     vertices some cycle of D reaches (left over by the peeling of sources)
     and bit 3 for those that reach a cycle (left over by the peeling of
     sinks): a traversal on the forward sheet can close a cycle only from a
     vertex with bit 3, one on the mirror sheet only from a vertex with bit 2. */
  /* peel() for a component of at most 64 contigs, a lane per contig: the arcs
     of D as two 64-bit sets per lane -- the tails of the arcs into the contig
     (read off its own list and, transposed, off the lists of the others: both
     views, so the order holds for the forward and for the mirror sheet whatever
     the twins' flags say) -- and Kahn's algorithm by levels: the contigs whose
     predecessors are all done get the next positions together.  Any topological
     order serves the sweeps.  ~10 instructions a level instead of ~90 a contig.
     False if D (both views) has a cycle: peel() then decides by its own rules. */
  GTS_HD bool peel_small()
  {
    const uint32_t lane = W::lane();
    const bool in = lane < nv;
    const uint32_t v = in ? lane : 0u;
    const uint32_t g = M.gorient[v];
    const bool os = (g & 3u) == 2;
    const uint32_t eb = in ? (uint32_t)M.coff[v] : 0u, ee = in ? (uint32_t)M.coff[v + 1] : 0u;
    uint32_t maxdeg = ee - eb;
    for (uint32_t off = W::WIDTH / 2; off > 0; off >>= 1) {
      const uint32_t o = W::shfl(maxdeg, lane ^ off);
      maxdeg = o > maxdeg ? o : maxdeg;
    }
    maxdeg = W::uni(maxdeg);
    uint64_t dout = 0, din = 0;
    for (uint32_t k = 0; k < maxdeg; ++k) {
      const bool has = eb + k < ee;
      const uint32_t ce = has ? eb + k : 0u;
      const uint32_t fl = edge_bits(ce);
      const uint32_t w = M.cend[ce];
      const bool darc = has && (!bits_marked(fl) || (fl & GTS_F_TWINLIVE));
      const bool fwd = ((fl & GTS_F_SENSE) != 0) == os;
      if (darc && fwd) dout |= 1ull << w;
      if (darc && !fwd) din |= 1ull << w;
    }
    /* the tail view, transposed: u -> v in D if v is in dout[u] */
    for (uint32_t u = 0; u < nv; ++u) {
      const uint64_t m = (uint64_t)W::bcast((uint32_t)dout, u) | (uint64_t)W::bcast((uint32_t)(dout >> 32), u) << 32;
      if ((m >> lane) & 1ull) din |= 1ull << u;
    }
    const uint64_t all = nv >= 64 ? ~0ull : (1ull << nv) - 1ull;
    uint64_t done = 0;
    uint32_t base = 0;
    while (done != all) {
      const uint64_t ready = W::ballot(in && !((done >> lane) & 1ull) && (din & ~done) == 0);
      if (!ready) return false;
      if ((ready >> lane) & 1ull) {
        const uint32_t tp = base + W::popc_below(ready, lane);
        M.tpos[v] = (typename GtsCompMemT<LDS>::idx_t)tp;
        M.topo[tp] = (typename GtsCompMemT<LDS>::idx_t)v;
      }
      base += W::popc(ready);
      done |= ready;
    }
    if (in) M.gorient[v] = (uint8_t)(g & 3u);
    W::fence();
    return true;
  }

  /* peel() of a component in global memory over the team's wavefronts, by levels:
     the contigs whose in-arcs are all used up take the next positions together,
     a lane each, and use up their out-arcs with atomic decrements; whoever takes
     a contig's last in-arc appends it.  Any topological order serves the sweeps
     (see peel_small), so the order inside a level is left to the atomics.  On one
     wavefront a pass over the 8247 contigs of the 50 M workload's largest
     component took 7 ms, and the cycle removal makes one after every cycle it
     marks: 79 of its 103 ms.  Every wavefront of the team calls this; all return
     the same.  One barrier a level (the order of that component is ~1000 levels
     deep, so the barriers are what a pass costs): the number of contigs a level
     appends is counted in one of three words taken in turn -- level l adds to
     [l % 3], reads it after the barrier and zeroes [(l + 1) % 3], which the slowest
     lane stopped reading a barrier ago. */
  GTS_HD bool peel_team_run()
  {
#if defined(__HIPCC__)
    if constexpr (W::TEAM && !LDS) {
      const uint32_t lane = W::lane();
      const uint32_t T = team_waves * W::WIDTH, t = team_wave * W::WIDTH + lane;
      auto deg = M.st_v;
      for (int pass = 0; pass < 2; ++pass) {
        const bool fwd = pass == 0;     /* pass 0 peels sources, pass 1 sinks */
        const uint32_t bit = fwd ? 4u : 8u;
        auto Qp = M.topo;
        if (!fwd) Qp = M.visited;
        for (uint32_t s = t; s < nv; s += T) {
          const uint32_t g = M.gorient[s];
          const bool os = (g & 3u) == 2;
          uint32_t d = 0;
          const uint32_t le = M.coff[s + 1] - M.e0;
          for (uint32_t ce = M.coff[s] - M.e0; ce < le; ++ce)
            d += d_arc(ce, os, !fwd) ? 1u : 0u;   /* in-arcs when peeling sources */
          deg[s] = d;
          M.gorient[s] = (uint8_t)((g & (fwd ? 3u : 7u)) | bit);
        }
        if (t == 0) { team->pq_lvl[0] = 0; team->pq_lvl[1] = 0; }
        W::team_barrier();
        /* level 0: the sources */
        for (uint32_t s = t; s < nv; s += T)
          if (deg[s] == 0) Qp[W::team_add(&team->pq_lvl[0], 1u)] = s;
        uint32_t th = 0, tn = 0, l = 0;
        for (;;) {
          W::team_barrier();
          const uint32_t added = *(volatile uint32_t *)&team->pq_lvl[l % 3u];
          th = tn; tn += added; ++l;
          if (added == 0) break;
          if (t == 0) team->pq_lvl[(l + 1u) % 3u] = 0;
          uint32_t *cnt = &team->pq_lvl[l % 3u];
          for (uint32_t idx = th + t; idx < tn; idx += T) {
            const uint32_t u = Qp[idx];
            const uint32_t g = M.gorient[u];
            if (fwd) M.tpos[u] = idx;
            M.gorient[u] = (uint8_t)(g & ~bit);
            const bool ou = (g & 3u) == 2;
            const uint32_t le = M.coff[u + 1] - M.e0;
            for (uint32_t ce = M.coff[u] - M.e0; ce < le; ++ce) {
              if (!d_arc(ce, ou, fwd)) continue;   /* out-arcs when peeling sources */
              const uint32_t v = M.cend[ce];
              if (W::team_add(&deg[v], 0xFFFFFFFFu) == 1u) Qp[tn + W::team_add(cnt, 1u)] = v;
            }
          }
        }
        W::team_barrier();   /* (the words are set again by the next pass) */
        if (fwd && tn == nv) return true;   /* acyclic: flags all cleared */
      }
    }
#endif
    return false;
  }

  GTS_HD bool peel()
  {
    const uint32_t lane = W::lane();
    if constexpr (LDS) {
      if (C.small_masks && nv <= W::WIDTH && peel_small()) return true;
    }
#if defined(__HIPCC__)
    if constexpr (W::TEAM && !LDS) {
      if (team && team_waves > 1) {
        if (lane == 0) team->kind = 2;
        W::team_barrier();
        const bool r = peel_team_run();
        W::team_barrier();
        return r;
      }
    }
#endif
    auto deg = M.st_v;
    for (int pass = 0; pass < 2; ++pass) {
      const bool fwd = pass == 0;     /* pass 0 peels sources, pass 1 sinks */
      const uint32_t bit = fwd ? 4u : 8u;
      auto Qp = M.topo;
      if (!fwd) Qp = M.visited;
      for (uint32_t s = lane; s < nv; s += W::WIDTH) {
        const bool os = (M.gorient[s] & 3u) == 2;
        uint32_t d = 0;
        const uint32_t le = M.coff[s + 1] - M.e0;
        for (uint32_t ce = M.coff[s] - M.e0; ce < le; ++ce)
          d += d_arc(ce, os, !fwd) ? 1u : 0u;   /* in-arcs when peeling sources */
        deg[s] = d;
        M.gorient[s] = (uint8_t)((M.gorient[s] & (fwd ? 3u : 7u)) | bit);
      }
      W::fence();
      uint32_t th = 0, tn = 0;
      for (uint32_t base = 0; base < nv; base += W::WIDTH) {
        const uint32_t v = base + lane;
        const bool src = v < nv && deg[v] == 0;
        const uint64_t sm = W::ballot(src);
        if (src) Qp[tn + W::popc_below(sm, lane)] = v;
        tn += W::popc(sm);
      }
      W::fence();
      while (th < tn) {
        /* the known part of the queue, a chunk at a time (see calc_cc); the
           lanes retire their own vertices' bookkeeping */
        const uint32_t cnt = tn - th < W::WIDTH ? tn - th : W::WIDTH;
        uint32_t my_u = 0, my_eb = 0, my_ee = 0, my_g = 0;
        if (lane < cnt) {
          my_u = Qp[th + lane];
          my_eb = M.coff[my_u] - M.e0; my_ee = M.coff[my_u + 1] - M.e0;
          my_g = M.gorient[my_u];
          if (fwd) M.tpos[my_u] = th + lane;
          M.gorient[my_u] = (uint8_t)(my_g & ~bit);
        }
        for (uint32_t i = 0; i < cnt; ++i) {
          const uint32_t eb = W::bcast(my_eb, i), ee = W::bcast(my_ee, i);
          const bool ou = (W::bcast(my_g, i) & 3u) == 2;
          for (uint32_t base = eb; base < ee; base += W::WIDTH) {
            const uint32_t ce = base + lane;
            const bool in = ce < ee;
            const uint32_t cec = in ? ce : eb;
            const bool arc = in && d_arc(cec, ou, fwd);   /* out-arcs when peeling sources */
            const uint32_t v = M.cend[cec];
            const uint32_t d = deg[v] - 1;
            if (arc) deg[v] = d;
            const bool ready = arc && d == 0;
            const uint64_t rm = W::ballot(ready);
            if (ready) Qp[tn + W::popc_below(rm, lane)] = v;
            tn += W::popc(rm);
            W::fence();
          }
        }
        th += cnt;
      }
      if (fwd && tn == nv) return true;   /* acyclic: flags all cleared */
    }
    return false;
  }

  /* create_walk on a clean component: one sweep over the precomputed order
     (see orient / peel and create_walk_fast for why the result is the reference's).
     The reached vertices are taken in the order of their topological position
     (upwards on the forward sheet, downwards on the mirror); every arc points
     onwards in that order, so a vertex is final when its turn comes.
     Two ways to find the next reached vertex:
       * LDS-resident components (below 4096 contigs): the order is read a
         chunk of 64 positions at a time and the lanes look up the labels;
       * components in global memory (any size): a bitmap over the positions
         (bit tpos[v] set when v gets its first label), scanned 64 words = 2048
         positions per load.  A walk then costs its reachable set, not the
         component: the 3000 walks of a 34 000-contig component (a scaffold each,
         tied together by one unmarked hub) swept 34 000 / 64 chunks of dependent
         gathers each before. */
  GTS_HD bool create_walk_clean(uint32_t start, uint64_t &cc_len, uint32_t &cc_n)
  {
    const uint32_t lane = W::lane();
    auto R = M.queue;
    auto depth = M.st_par;
    /* the start's live edges pick the sheet */
    bool has_s = false, has_a = false;
    {
      const uint32_t eb = eoff(start), ee = eoff(start + 1);
      for (uint32_t base = eb; base < ee; base += W::WIDTH) {
        const uint32_t ce = base + lane;
        bool live = false, sense = false;
        if (ce < ee) {
          live = !edge_marked(ce);
          sense = (M.cflags[ce] & GTS_F_SENSE) != 0;
        }
        has_s |= W::ballot(live && sense) != 0;
        has_a |= W::ballot(live && !sense) != 0;
      }
    }
    if (has_s && has_a) { W::count(C.why + 0); return false; }
    if (!has_s && !has_a) return true;
    const bool forward = has_s == ((W::uni((uint32_t)M.gorient[start]) & 3u) == 2);
    uint32_t nr = 0, pending = 0, best_t = GTS_NONE;
    uint64_t best_len = 0;
    bool inexact = false, bad = false;
    M.plen[start] = (len_t)M.cseq[start];
    depth[start] = 0;
    M.nd[start] = 0;
    W::fence();
    const uint32_t spos = W::uni(M.tpos[start]);
    /* sparse sweep: position bitmap in st_cur (zeroed by makescaffold, left
       zero by every walk; nothing else writes it while the component is clean) */
    uint32_t *pbits = nullptr;
    uint32_t nwords = 0;
    if constexpr (!LDS) { pbits = (uint32_t *)&M.st_cur[0]; nwords = (nv + 31) / 32; }

    /* relaxes the out-arcs of u (its label is final) */
    auto process = [&](uint32_t u) {
      const bool du = ((W::uni((uint32_t)M.gorient[u]) & 3u) == 2) == forward;
      const nd_t ndu = uni_t(M.nd[u]);   /* nd[start] = 0 */
      const len_t plu = uni_t(M.plen[u]);
      const uint32_t dpu = W::uni(depth[u]);   /* depth[start] = 0 */
      const uint32_t eb = eoff(u), ee = eoff(u + 1);
      bool us = false, ua = false;
      for (uint32_t base = eb; base < ee && !bad; base += W::WIDTH) {
        /* straight-line: every lane loads (a lane past the list reads the
           list's first edge) and the predicates are values, so there is one
           predicated region -- the stores -- instead of three nested ones */
        const uint32_t ce = base + lane;
        const bool in = ce < ee;
        const uint32_t cec = in ? ce : eb;
        const uint32_t fs = edge_bits(cec);
        const bool live = in && !bits_marked(fs);
        const bool sense = (fs & GTS_F_SENSE) != 0;
        const bool arc = live && sense == du;
        const uint32_t v = M.cend[cec];
        const nd_t w = (nd_t)dist_of(cec);
        const float cand = (float)(ndu + w);
        const float old = M.distmap[v];
        const bool imp = arc && (old == GTS_DIST_UNSET || old > cand);
        const bool tie = arc && !imp && old == cand;
        const bool fresh = imp && old == GTS_DIST_UNSET;
        inexact |= arc && !(cand > -16777216.0f && cand < 16777216.0f);
        if (imp) {
          M.distmap[v] = cand;
          M.edgemap[v] = ce;
          M.par[v] = u;
          M.nd[v] = u == start ? w : (nd_t)cand;
          M.plen[v] = (len_t)(plu + (len_t)M.cseq[v]);
          depth[v] = dpu + 1;
        }
        us |= W::ballot(live && sense) != 0;
        ua |= W::ballot(live && !sense) != 0;
        const uint64_t fm = W::ballot(fresh);
        if (fresh) {
          R[nr + W::popc_below(fm, lane)] = v;
          if constexpr (!LDS) { const uint32_t tp = M.tpos[v]; W::or_bits(pbits + (tp >> 5), 1u << (tp & 31)); }
        }
        nr += W::popc(fm);
        pending += W::popc(fm);
        uint64_t tm = W::ballot(tie);
        /* (inexact is kept per lane; the wave is asked only where it matters) */
        if (tm && W::ballot(inexact)) { W::count(C.why + 5); bad = true; break; }
        while (tm) {
          const uint32_t tl = W::ctz(tm);
          tm &= tm - 1;
          const uint32_t tv = W::bcast(v, tl), tce = W::bcast(ce, tl);
          const uint32_t up = W::uni(M.par[tv]);
          if (pushed_after(up, u, start)) {
            M.edgemap[tv] = tce;
            M.par[tv] = u;
            M.plen[tv] = (len_t)(plu + (len_t)uni_t(M.cseq[tv]));
            depth[tv] = dpu + 1;
            W::fence();
          }
        }
        W::fence();
      }
      if (!bad && u != start && !(us && ua)) {
        if (plu > best_len) { best_len = plu; best_t = u; }
        else if (plu == best_len && best_t != GTS_NONE) {
          if (W::ballot(inexact)) { W::count(C.why + 7); bad = true; }
          else if (pushed_after(u, best_t, start)) best_t = u;
        }
      }
    };

    process(start);
    if (LDS) {
      int64_t pos = forward ? (int64_t)spos + 1 : (int64_t)spos - 1;
      while (pending > 0 && !bad && pos >= 0 && pos < (int64_t)nv) {
        /* next reached vertices in sweep order */
        const int64_t mypos = forward ? pos + (int64_t)lane : pos - (int64_t)lane;
        uint32_t cv = 0;
        const bool inrange = mypos >= 0 && mypos < (int64_t)nv;
        if (inrange) cv = M.topo[(uint32_t)mypos];
        /* a vertex handled in this chunk may label a later vertex of the same
           chunk: look again after every vertex */
        uint32_t next_lane = 0;
        while (!bad) {
          const bool reached = inrange && lane >= next_lane && M.distmap[cv] != GTS_DIST_UNSET;
          const uint64_t rm = W::ballot(reached);
          if (!rm) break;
          const uint32_t l = W::ctz(rm);
          next_lane = l + 1;
          --pending;
          process(W::bcast(cv, l));
        }
        pos = forward ? pos + (int64_t)W::WIDTH : pos - (int64_t)W::WIDTH;
      }
    } else {
      /* cw: word of the position handled last; bits at or before it (in sweep
         direction) are never set again */
      int64_t cw = (int64_t)(spos >> 5);
      while (pending > 0 && !bad && cw >= 0 && cw < (int64_t)nwords) {
        const int64_t myw = forward ? cw + (int64_t)lane : cw - (int64_t)lane;
        uint32_t word = 0;
        if (myw >= 0 && myw < (int64_t)nwords) word = pbits[myw];
        const uint64_t wm = W::ballot(word != 0);
        if (!wm) { cw = forward ? cw + (int64_t)W::WIDTH : cw - (int64_t)W::WIDTH; continue; }
        const uint32_t l = W::ctz(wm);
        const uint32_t wv = W::bcast(word, l);
        const uint32_t bit = forward ? (uint32_t)__builtin_ctz(wv) : 31u - (uint32_t)__builtin_clz(wv);
        cw = forward ? cw + (int64_t)l : cw - (int64_t)l;
        if (lane == 0) pbits[cw] = wv & ~(1u << bit);
        W::fence();
        --pending;
        process(W::uni(M.topo[(uint32_t)cw * 32u + bit]));
      }
    }
    if (!bad && best_t != GTS_NONE && best_len > cc_len) {
      uint32_t cv = best_t, n = 0;
      while (cv != start) {
        const uint32_t re = W::uni(M.edgemap[cv]);
        M.cc_best[n++] = re;
        cv = W::uni(M.par[cv]);
      }
      cc_len = best_len;
      cc_n = n;
    }
    for (uint32_t k = lane; k < nr; k += W::WIDTH) {
      const uint32_t v = R[k];
      M.distmap[v] = GTS_DIST_UNSET;
      if (!LDS && bad) pbits[M.tpos[v] >> 5] = 0;   /* a sweep that gave up leaves bits behind (pbits is null with LDS) */
      note_labelled(v);
    }
    W::fence();
    return !bad;
  }


  /* ---- the walks of one cc side by side (LDS-resident clean components) ----
     The reference makes every walk of a cc before it marks anything
     (algorithms.c:809-832), so they are independent.  On a clean component a
     walk is one sweep over the topological order (create_walk_clean); here the
     wavefront is split into groups of L lanes, group g sweeps for terminal
     j0 + g with labels, tree lengths, edgemap and parents in walk slot g
     (GtsCompMemT::wbase), all groups in lock step: an iteration looks at the
     next L positions of a group's sweep or relaxes up to L arcs of its current
     vertex, lane = (walk, arc).  The groups share the graph in LDS and nothing
     else; the forward and the mirror sheet are swept at the same time.
     A tie (two in-arcs attaining a label, two terminals with the longest
     walk) or a start with live edges in both senses sets `bad`: the caller
     then makes the walks of this cc one by one (create_walk), which resolves
     them (pushed_after) or runs the reference's search.
     Slots must be clean (labels unset) on entry; clear_walk_slots() after. */
  /* pushed_after() on a walk slot, for the lanes that met a tie (round 4: a tie
     used to end the batch, and the cc's walks were made one by one -- 7 of the
     8.3 ms of a 716-contig component).  The slot keeps no depths: the generation
     of a vertex is the length of its parent chain, so the two chains are climbed
     in lock step -- the one that reaches the start first is the shallower --,
     then, at equal depth, again up to the common parent (as pushed_after).  Only
     valid while the labels are exact (the caller asks). */
  template <class P>
  static GTS_HD bool pushed_after_slot(P par, P emap, uint32_t a, uint32_t b, uint32_t start)
  {
    uint32_t xa = a, xb = b;
    while (xa != start && xb != start) { xa = par[xa]; xb = par[xb]; }
    if (xa != start || xb != start) return xb == start;
    uint32_t ea = emap[a], eb = emap[b], pa = par[a], pb = par[b];
    while (pa != pb) { ea = emap[pa]; eb = emap[pb]; pa = par[pa]; pb = par[pb]; }
    return ea > eb;
  }

  template <uint32_t L, bool D16>
  GTS_HD void walks_clean_batch(uint32_t j0, uint32_t nb, uint32_t &r_len, uint32_t &r_t, bool &r_bad)
  {
    typedef typename GtsCompMemT<LDS>::idx_t idx_t;
    static_assert(L == 8 || W::WIDTH == 1, "the group reductions are written for eight lanes");
    const uint32_t lane = W::lane(), g = lane / L, a = lane % L;
    const uint32_t gsh = g * L;
    const uint64_t gm = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
    const uint32_t p4 = ((nv * 4 + 15) / 16) * 16, p2 = ((nv * 2 + 15) / 16) * 16;
    bool active = g < nb;
    auto sbase = M.wbase + (active ? g : 0u) * (2 * p4 + 2 * p2);
    auto dist = (GTS_P(float))sbase;
    auto plen = (GTS_P(uint32_t))(sbase + p4);
    auto emap = (GTS_P(idx_t))(sbase + 2 * p4);
    auto par = (GTS_P(idx_t))(sbase + 2 * p4 + p2);
    const uint32_t start = active ? (uint32_t)M.term[j0 + g] : 0u;
    /* the start's live edges pick the sheet */
    const uint32_t sb0 = M.coff[start], se0 = M.coff[start + 1];
    bool hs = false, ha = false;
    for (uint32_t cur0 = sb0; W::ballot(active && cur0 < se0); cur0 += L) {
      const uint32_t ce = cur0 + a;
      const bool in = active && ce < se0;
      const uint32_t fs = edge_bits(in ? ce : sb0);
      const bool live = in && !bits_marked(fs);
      const bool sense = (fs & GTS_F_SENSE) != 0;
      const uint64_t bs = W::ballot(live && sense), ba = W::ballot(live && !sense);
      hs |= ((bs >> gsh) & gm) != 0;
      ha |= ((ba >> gsh) & gm) != 0;
    }
    uint32_t bad = active && hs && ha ? 1u : 0u;
    if (bad || !(hs || ha)) active = false;          /* (nothing reachable: empty walk) */
    const bool forward = hs == ((M.gorient[start] & 3u) == 2);
    const int32_t step = forward ? 1 : -1;
    /* The loop is written for the instruction issue it is bound by (see
       walks_clean_batch_small): the state of a group is the number of labelled
       positions it has not handled yet -- none = done, and a walk that met a tie
       drops them --, the start is labelled in its slot like any other vertex,
       one pass of the outer loop is one vertex, and what the lanes of a group
       found is combined over the group once per vertex. */
    if (active && a == 0) { dist[start] = 0.0f; plen[start] = (uint32_t)M.cseq[start]; }
    int32_t pos = (int32_t)M.tpos[start];        /* the sweep has handled everything before pos */
    uint32_t pending = active ? 1u : 0u;
    uint32_t best_len = 0, best_t = GTS_NONE;
    uint32_t inex = 0;       /* this lane has seen a label outside the range where floats are exact */
    W::fence();
    while (W::ballot(pending != 0)) {
      bool on = pending != 0;
      /* (1) the next labelled position at or after pos, L positions a step */
      uint32_t u = 0;
      bool found = !on;
      while (W::ballot(!found)) {
        const int32_t p = pos + (int32_t)a * step;
        const bool inr = !found && p >= 0 && p < (int32_t)nv;
        const uint32_t cv = M.topo[inr ? (uint32_t)p : 0u];
        const float lbl = dist[cv];
        const uint64_t rb = (W::ballot(inr && lbl != GTS_DIST_UNSET) >> gsh) & gm;
        const uint32_t k = rb ? W::ctz(rb) : 0u;
        const uint32_t uk = W::shfl(cv, gsh + k);
        if (!found) {
          if (rb) { u = uk; pos += (int32_t)(k + 1u) * step; found = true; }
          else {
            pos += (int32_t)L * step;
            /* (a labelled position is ahead while pending != 0; never past the range) */
            if (pos < 0 || pos >= (int32_t)nv) { found = true; on = false; pending = 0; }
          }
        }
      }
      const uint32_t ub = M.coff[u], ue = M.coff[u + 1];
      const bool du = ((M.gorient[u] & 3u) == 2) == forward;
      const int32_t ndu = (int32_t)dist[u];                 /* the integer the reference pushes with the node */
      const uint32_t plu = plen[u];
      uint32_t mine = 0;       /* vertices this lane labelled first */
      uint32_t seen = 0;       /* 1: live sense edge, 2: live antisense edge, 4: a tie that has no closed form */
      for (uint32_t cur = ub; W::ballot(on && cur < ue); cur += L) {
        const uint32_t ce = cur + a;
        const bool in = on && ce < ue;
        const uint32_t cec = in ? ce : ub;
        const uint32_t fs = edge_bits(cec);
        const uint32_t v = M.cend[cec];
        const float cand = (float)(ndu + dist_w<D16>(cec));
        const float old = dist[v];
        const uint32_t sv = (uint32_t)M.cseq[v];
        const bool live = in && !bits_marked(fs);
        const bool sense = (fs & GTS_F_SENSE) != 0;
        const bool arc = live && sense == du;
        const bool unset = old == GTS_DIST_UNSET;
        const bool imp = arc && (unset || old > cand);
        const bool tie = arc && !imp && old == cand;
        if (imp) {
          dist[v] = cand;
          emap[v] = (idx_t)ce;
          par[v] = (idx_t)u;
          plen[v] = plu + sv;
        }
        mine += imp && unset ? 1u : 0u;
        seen |= live ? (sense ? 1u : 2u) : 0u;
        inex |= arc && !(cand > -16777216.0f && cand < 16777216.0f) ? 1u : 0u;
        W::fence();
        if (W::ballot(tie)) {
          /* two in-arcs attain the label of v: edgemap keeps the one whose value
             arrived first (create_walk_clean) */
          const uint32_t gi = W::group8_or32(inex);
          if (tie) {
            if (gi) seen |= 4u;
            else if (pushed_after_slot(par, emap, (uint32_t)par[v], u, start)) {
              emap[v] = (idx_t)ce;
              par[v] = (idx_t)u;
              plen[v] = plu + sv;
            }
          }
          W::fence();
        }
      }
      mine = W::group8_add32(mine);
      seen = W::group8_or32(seen);
      bool tie_t = false;
      if (on) {
        pending += mine - 1u;
        /* reached terminal (algorithms.c:694): candidate end of the walk */
        if (u != start && (seen & 3u) != 3u) {
          if (plu > best_len) { best_len = plu; best_t = u; }
          else if (plu == best_len && best_t != GTS_NONE) tie_t = true;
        }
        bad |= seen >> 2;
      }
      if (W::ballot(tie_t)) {
        /* two terminals with the longest walk: the reference keeps the one popped last */
        const uint32_t gi = W::group8_or32(inex);
        if (tie_t) {
          if (gi) bad = 1u;
          else if (pushed_after_slot(par, emap, u, best_t, start)) best_t = u;
        }
      }
      if (bad) pending = 0;
    }
    r_len = best_len; r_t = best_t; r_bad = bad != 0;
  }


  /* walks_clean_batch for a component of at most 64 contigs.  The labelled
     positions a sweep has not handled yet are a 64-bit set in the registers of
     the group (bit tpos[v] joins when v gets its first label), so "the next
     labelled vertex in sweep order" is a count of leading or trailing zeros.
     The loop is written for the instruction issue it is bound by (a step was
     ~190 instructions, half of them scalar mask bookkeeping for a dozen
     loop-carried conditions): the state of a group is its pending set -- empty
     = done, and a walk that met a tie empties it --, the start is labelled in
     its slot like any other vertex, one pass of the outer loop is one vertex
     (its arcs in an inner loop that runs once unless a list has more than eight
     entries), and what the lanes of a group found -- new positions, senses
     seen, ties -- is OR-ed over the group once per vertex (DPP), not per step. */
  template <bool D16>
  GTS_HD void walks_clean_batch_small(uint32_t j0, uint32_t nb, uint32_t &r_len, uint32_t &r_t, bool &r_bad)
  {
    typedef typename GtsCompMemT<LDS>::idx_t idx_t;
    constexpr uint32_t L = 8;
    const uint32_t lane = W::lane(), g = lane / L, a = lane % L;
    const uint32_t gsh = g * L;
    const uint64_t gm = (1ull << L) - 1ull;
    const uint32_t p4 = ((nv * 4 + 15) / 16) * 16, p2 = ((nv * 2 + 15) / 16) * 16;
    bool active = g < nb;
    auto sbase = M.wbase + (active ? g : 0u) * (2 * p4 + 2 * p2);
    auto dist = (GTS_P(float))sbase;
    auto plen = (GTS_P(uint32_t))(sbase + p4);
    auto emap = (GTS_P(idx_t))(sbase + 2 * p4);
    auto par = (GTS_P(idx_t))(sbase + 2 * p4 + p2);
    const uint32_t start = active ? (uint32_t)M.term[j0 + g] : 0u;
    const uint32_t sb0 = M.coff[start], se0 = M.coff[start + 1];
    bool hs = false, ha = false;
    for (uint32_t cur0 = sb0; W::ballot(active && cur0 < se0); cur0 += L) {
      const uint32_t ce = cur0 + a;
      const bool in = active && ce < se0;
      const uint32_t fs = edge_bits(in ? ce : sb0);
      const bool live = in && !bits_marked(fs);
      const bool sense = (fs & GTS_F_SENSE) != 0;
      const uint64_t bs = W::ballot(live && sense), ba = W::ballot(live && !sense);
      hs |= ((bs >> gsh) & gm) != 0;
      ha |= ((ba >> gsh) & gm) != 0;
    }
    uint32_t bad = active && hs && ha ? 1u : 0u;
    if (bad || !(hs || ha)) active = false;
    const bool forward = hs == ((M.gorient[start] & 3u) == 2);
    /* the start: label 0, its own length, position pending (its sheet is the
       sense of its live edges: (strand == 2) == forward is hs) */
    if (active && a == 0) { dist[start] = 0.0f; plen[start] = (uint32_t)M.cseq[start]; }
    uint64_t pend = active ? 1ull << (uint32_t)M.tpos[start] : 0ull;
    uint32_t best_len = 0, best_t = GTS_NONE;
    uint32_t inex = 0;
    W::fence();
    while (W::ballot(pend != 0)) {
      const bool on = pend != 0;
      const uint32_t p = on ? (forward ? W::ctz(pend) : 63u - W::clz64(pend)) : 0u;
      pend &= ~(1ull << p);
      const uint32_t u = M.topo[p];
      const uint32_t ub = M.coff[u], ue = M.coff[u + 1];
      const bool du = ((M.gorient[u] & 3u) == 2) == forward;
      const int32_t ndu = (int32_t)dist[u];                 /* the integer the reference pushes with the node */
      const uint32_t plu = plen[u];
      uint64_t mine = 0;       /* positions this lane labelled first */
      uint32_t seen = 0;       /* 1: live sense edge, 2: live antisense edge, 4: tie */
      for (uint32_t cur = ub; W::ballot(on && cur < ue); cur += L) {
        const uint32_t ce = cur + a;
        const bool in = on && ce < ue;
        const uint32_t cec = in ? ce : ub;
        const uint32_t fs = edge_bits(cec);
        const uint32_t v = M.cend[cec];
        const float cand = (float)(ndu + dist_w<D16>(cec));
        const float old = dist[v];
        const uint32_t sv = (uint32_t)M.cseq[v];
        const uint32_t tv = M.tpos[v];
        const bool live = in && !bits_marked(fs);
        const bool sense = (fs & GTS_F_SENSE) != 0;
        const bool arc = live && sense == du;
        const bool unset = old == GTS_DIST_UNSET;
        const bool imp = arc && (unset || old > cand);
        const bool tie = arc && !imp && old == cand;
        if (imp) {
          dist[v] = cand;
          emap[v] = (idx_t)ce;
          par[v] = (idx_t)u;
          plen[v] = plu + sv;
        }
        if (imp && unset) mine |= 1ull << tv;
        seen |= live ? (sense ? 1u : 2u) : 0u;
        inex |= arc && !(cand > -16777216.0f && cand < 16777216.0f) ? 1u : 0u;
        W::fence();
        if (W::ballot(tie)) {      /* (as walks_clean_batch) */
          const uint32_t gi = W::group8_or32(inex);
          if (tie) {
            if (gi) seen |= 4u;
            else if (pushed_after_slot(par, emap, (uint32_t)par[v], u, start)) {
              emap[v] = (idx_t)ce;
              par[v] = (idx_t)u;
              plen[v] = plu + sv;
            }
          }
          W::fence();
        }
      }
      pend |= W::group8_or(mine);
      seen = W::group8_or32(seen);
      bool tie_t = false;
      if (on) {
        /* reached terminal (algorithms.c:694): candidate end of the walk */
        if (u != start && (seen & 3u) != 3u) {
          if (plu > best_len) { best_len = plu; best_t = u; }
          else if (plu == best_len && best_t != GTS_NONE) tie_t = true;
        }
        bad |= seen >> 2;
      }
      if (W::ballot(tie_t)) {
        const uint32_t gi = W::group8_or32(inex);
        if (tie_t) {
          if (gi) bad = 1u;
          else if (pushed_after_slot(par, emap, u, best_t, start)) best_t = u;
        }
      }
      if (bad) pend = 0;
    }
    r_len = best_len; r_t = best_t; r_bad = bad != 0;
  }

  /* ---- the walks of one cc of a component that is NOT clean, side by side (round 4) ----
     Without a topological order of the whole component every walk needs its own:
     create_walk_fast() collects the states reachable from the start (pass 1: strands,
     in-degrees) and relaxes them in Kahn order (pass 2).  One by one that is ~0.2 us a
     vertex step and walk -- the 45 walks of a 380-contig component took 8 of the
     launch's 10.6 ms, an 80-contig one claimed late 2.2 ms.  Here a group of eight lanes
     makes a walk, as in walks_clean_batch, with its own strands, in-degrees and queue
     next to labels, tree lengths, edgemap and parents: a slot of 16 bytes a contig
     (gts_uwalk_slot_bytes), as many as the component's walk scratch holds.  Anything
     the closed form does not cover -- a clash of strands, a self arc or u-turn, a start
     with edges in both senses, a cycle among the reachable states, an inexact tie --
     sets `bad`; the caller then makes the walks of the cc one by one. */
  static GTS_HD uint32_t gts_uwalk_slot_bytes(uint32_t nv)
  {
    const uint32_t p4 = ((nv * 4 + 15) / 16) * 16, p2 = ((nv * 2 + 15) / 16) * 16;
    return 2 * p4 + 4 * p2;
  }
  template <bool D16>
  GTS_HD void walks_fast_batch(uint32_t j0, uint32_t nb, uint32_t &r_len, uint32_t &r_t, bool &r_bad)
  {
    typedef typename GtsCompMemT<LDS>::idx_t idx_t;
    constexpr uint32_t L = 8;
    const uint32_t lane = W::lane(), g = lane / L, a = lane % L;
    const uint32_t gsh = g * L;
    const uint32_t below = (1u << a) - 1u;
    const uint32_t p4 = ((nv * 4 + 15) / 16) * 16, p2 = ((nv * 2 + 15) / 16) * 16;
    bool active = g < nb;
    auto sbase = M.wbase + (active ? g : 0u) * (2 * p4 + 4 * p2);
    auto dist = (GTS_P(float))sbase;
    auto plen = (GTS_P(uint32_t))(sbase + p4);
    auto emap = (GTS_P(idx_t))(sbase + 2 * p4);
    auto par = (GTS_P(idx_t))(sbase + 2 * p4 + p2);
    auto deg = (GTS_P(uint16_t))(sbase + 2 * p4 + 2 * p2);   /* in-degree | (strand + 1) << 14 */
    auto Q = (GTS_P(idx_t))(sbase + 2 * p4 + 3 * p2);
    const uint32_t start = active ? (uint32_t)M.term[j0 + g] : 0u;
    const uint32_t sb0 = M.coff[start], se0 = M.coff[start + 1];
    bool hs = false, ha = false;
    for (uint32_t cur0 = sb0; W::ballot(active && cur0 < se0); cur0 += L) {
      const uint32_t ce = cur0 + a;
      const bool in = active && ce < se0;
      const uint32_t fs = edge_bits(in ? ce : sb0);
      const bool live = in && !bits_marked(fs);
      const bool sense = (fs & GTS_F_SENSE) != 0;
      const uint64_t bs = W::ballot(live && sense), ba = W::ballot(live && !sense);
      hs |= ((bs >> gsh) & 0xFFu) != 0;
      ha |= ((ba >> gsh) & 0xFFu) != 0;
    }
    uint32_t bad = active && hs && ha ? 1u : 0u;
    if (bad || !(hs || ha)) active = false;          /* (nothing reachable: empty walk) */
    /* pass 1: the reachable states, their strands and in-degrees (create_walk_fast) */
    uint32_t qh = 0, nr = active ? 1u : 0u;
    if (active && a == 0) { Q[0] = (idx_t)start; deg[start] = (uint16_t)((hs ? 2u : 1u) << 14); }
    W::fence();
    while (W::ballot(active && qh < nr)) {
      const bool on = active && qh < nr;
      const uint32_t u = Q[on ? qh : 0u];
      const bool du = ((uint32_t)deg[u] >> 14) == 2u;
      const uint32_t ub = M.coff[u], ue = M.coff[u + 1];
      uint32_t clash = 0;
      for (uint32_t cur = ub; W::ballot(on && cur < ue); cur += L) {
        const uint32_t ce = cur + a;
        const bool in = on && ce < ue;
        const uint32_t cec = in ? ce : ub;
        const uint32_t fs = edge_bits(cec);
        const uint32_t v = M.cend[cec];
        const uint32_t dv = deg[v];
        const bool arc = in && !bits_marked(fs) && ((fs & GTS_F_SENSE) != 0) == du;
        const uint32_t od = gts_next_dir((uint8_t)fs) ? 2u : 1u, ov = dv >> 14;
        clash |= arc && (v == u || v == start || gts_vertex_is_marked(M.vst[v]) || (ov != 0 && ov != od) ||
                         (fs & GTS_F_UTURN)) ? 1u : 0u;
        const bool fresh = arc && ov == 0;
        const uint32_t fm = (uint32_t)(W::ballot(fresh) >> gsh) & 0xFFu;
        if (arc) {
          if (fresh) { deg[v] = (uint16_t)((od << 14) | 1u); Q[nr + W::popc((uint64_t)(fm & below))] = (idx_t)v; }
          else deg[v] = (uint16_t)(dv + 1u);
        }
        nr += W::popc((uint64_t)fm);
        W::fence();
      }
      if (W::group8_or32(clash)) { bad = 1u; active = false; }
      if (on) ++qh;
    }
    /* pass 2: relaxation in Kahn order; the queue takes the place of pass 1's */
    uint32_t qh2 = 0, nq = active ? 1u : 0u, best_len = 0, best_t = GTS_NONE, inex = 0;
    if (active && a == 0) { Q[0] = (idx_t)start; dist[start] = 0.0f; plen[start] = (uint32_t)M.cseq[start]; }
    W::fence();
    while (W::ballot(active && qh2 < nq)) {
      const bool on = active && qh2 < nq;
      const uint32_t u = Q[on ? qh2 : 0u];
      const bool du = ((uint32_t)deg[u] >> 14) == 2u;
      const int32_t ndu = (int32_t)dist[u];
      const uint32_t plu = plen[u];
      const uint32_t ub = M.coff[u], ue = M.coff[u + 1];
      uint32_t seen = 0;       /* 1: live sense edge, 2: live antisense edge, 4: a tie without closed form */
      for (uint32_t cur = ub; W::ballot(on && cur < ue); cur += L) {
        const uint32_t ce = cur + a;
        const bool in = on && ce < ue;
        const uint32_t cec = in ? ce : ub;
        const uint32_t fs = edge_bits(cec);
        const uint32_t v = M.cend[cec];
        const float cand = (float)(ndu + dist_w<D16>(cec));
        const float old = dist[v];
        const uint32_t sv = (uint32_t)M.cseq[v];
        const uint32_t dv = deg[v];
        const bool live = in && !bits_marked(fs);
        const bool sense = (fs & GTS_F_SENSE) != 0;
        const bool arc = live && sense == du;
        const bool unset = old == GTS_DIST_UNSET;
        const bool imp = arc && (unset || old > cand);
        const bool tie = arc && !imp && old == cand;
        if (imp) {
          dist[v] = cand;
          emap[v] = (idx_t)ce;
          par[v] = (idx_t)u;
          plen[v] = plu + sv;
        }
        const bool ready = arc && ((dv - 1u) & 0x3FFFu) == 0;
        const uint32_t rm = (uint32_t)(W::ballot(ready) >> gsh) & 0xFFu;
        if (arc) deg[v] = (uint16_t)(dv - 1u);
        if (ready) Q[nq + W::popc((uint64_t)(rm & below))] = (idx_t)v;
        nq += W::popc((uint64_t)rm);
        seen |= live ? (sense ? 1u : 2u) : 0u;
        inex |= arc && !(cand > -16777216.0f && cand < 16777216.0f) ? 1u : 0u;
        W::fence();
        if (W::ballot(tie)) {      /* (as walks_clean_batch) */
          const uint32_t gi = W::group8_or32(inex);
          if (tie) {
            if (gi) seen |= 4u;
            else if (pushed_after_slot(par, emap, (uint32_t)par[v], u, start)) {
              emap[v] = (idx_t)ce;
              par[v] = (idx_t)u;
              plen[v] = plu + sv;
            }
          }
          W::fence();
        }
      }
      seen = W::group8_or32(seen);
      bool tie_t = false;
      if (on) {
        ++qh2;
        if (u != start && (seen & 3u) != 3u) {
          if (plu > best_len) { best_len = plu; best_t = u; }
          else if (plu == best_len && best_t != GTS_NONE) tie_t = true;
        }
        bad |= seen >> 2;
      }
      if (W::ballot(tie_t)) {
        const uint32_t gi = W::group8_or32(inex);
        if (tie_t) {
          if (gi) bad = 1u;
          else if (pushed_after_slot(par, emap, u, best_t, start)) best_t = u;
        }
      }
      if (bad) active = false;
    }
    /* a cycle among the reachable states (walk_cyclic's case): not here */
    const bool cyc = active && qh2 != nr;
    if (cyc) bad = 1u;
    if (C.why) {   /* statistics: walks of the batch by outcome */
      const uint32_t ncyc = W::popc(W::ballot(cyc && a == 0)), nbad = W::popc(W::ballot(bad != 0 && a == 0 && g < nb));
      if (lane == 0) {
        W::add64((uint64_t *)C.why + 8, nb); W::add64((uint64_t *)C.why + 9, nbad); W::add64((uint64_t *)C.why + 10, ncyc);
      }
    }
    r_len = best_len; r_t = best_t; r_bad = bad != 0;
  }

  /* the walks of the cc [tb, te) of a component that is not clean in batches; false:
     nothing is kept, the caller makes them one by one */
  GTS_HD bool cc_walks_batched_unclean(uint32_t tb, uint32_t te, uint64_t &cc_len, uint32_t &cc_n,
                                       uint32_t *best_start)
  {
    typedef typename GtsCompMemT<LDS>::idx_t idx_t;
    const uint32_t lane = W::lane();
    const uint32_t ubytes = gts_uwalk_slot_bytes(nv);
    uint32_t per = (uint32_t)(((uint64_t)M.wslots * gts_walk_slot_bytes(nv)) / ubytes);
    if (per > W::WIDTH / 8) per = W::WIDTH / 8;
    if (per < 2) return false;
    const uint32_t p4 = ((nv * 4 + 15) / 16) * 16, p2 = ((nv * 2 + 15) / 16) * 16;
    bool ok = true;
    const uint32_t nfast0 = nfast;
    for (uint32_t j0 = tb; j0 < te && ok; j0 += per) {
      if (cc_len == all_bases()) break;
      const uint32_t nb = te - j0 < per ? te - j0 : per;
      for (uint32_t k = 0; k < nb; ++k) {      /* labels unset, strands and in-degrees zero */
        auto dist = (GTS_P(float))(M.wbase + k * ubytes);
        auto deg = (GTS_P(uint16_t))(M.wbase + k * ubytes + 2 * p4 + 2 * p2);
        for (uint32_t s = lane; s < nv; s += W::WIDTH) { dist[s] = GTS_DIST_UNSET; deg[s] = 0; }
      }
      W::fence();
      uint32_t r_len, r_t;
      bool r_bad;
      if (LDS && M.d16) walks_fast_batch<true>(j0, nb, r_len, r_t, r_bad);
      else walks_fast_batch<false>(j0, nb, r_len, r_t, r_bad);
      if (W::ballot(r_bad)) { ok = false; break; }
      uint32_t wg = GTS_NONE;
      for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t len = W::bcast(r_len, k * 8);
        if ((uint64_t)len > cc_len) { cc_len = len; wg = k; }
      }
      if (wg != GTS_NONE) {
        auto sbase = M.wbase + wg * ubytes;
        auto emap = (GTS_P(idx_t))(sbase + 2 * p4);
        auto par = (GTS_P(idx_t))(sbase + 2 * p4 + p2);
        const uint32_t start = W::uni((uint32_t)M.term[j0 + wg]);
        uint32_t cv = W::bcast(r_t, wg * 8), n = 0;
        while (cv != start) {
          const uint32_t re = W::uni((uint32_t)emap[cv]);
          M.cc_best[n++] = (idx_t)re;
          cv = W::uni((uint32_t)par[cv]);
        }
        cc_n = n;
        if (best_start) *best_start = start;
      }
      nfast += nb;
    }
    clear_walk_slots(1);   /* distmap: unset between walks */
    if (!ok) nfast = nfast0;
    return ok;
  }

  /* labels of the walk slots back to "unset" (slot 0's are distmap) */
  GTS_HD void clear_walk_slots(uint32_t nslots)
  {
    const uint32_t lane = W::lane();
    const uint32_t sbytes = gts_walk_slot_bytes(nv);
    for (uint32_t k = 0; k < nslots; ++k) {
      auto dist = (GTS_P(float))(M.wbase + k * sbytes);
      for (uint32_t s = lane; s < nv; s += W::WIDTH) dist[s] = GTS_DIST_UNSET;
    }
    W::fence();
  }

  /* the walks of the cc with terminals [tb, te) in batches; false if a walk
     met a tie (nothing is kept then, the caller makes the walks one by one) */
  template <uint32_t L>
  GTS_HD bool cc_walks_batched(uint32_t tb, uint32_t te, uint64_t &cc_len, uint32_t &cc_n,
                               uint32_t *best_start = nullptr)
  {
    typedef typename GtsCompMemT<LDS>::idx_t idx_t;
    const uint32_t G = W::WIDTH / L;
    const uint32_t per = M.wslots < G ? M.wslots : G;
    const uint32_t sbytes = gts_walk_slot_bytes(nv);
    const uint32_t p4 = ((nv * 4 + 15) / 16) * 16, p2 = ((nv * 2 + 15) / 16) * 16;
    bool ok = true;
    uint32_t used = 0;
    const uint32_t nfast0 = nfast;
    for (uint32_t j0 = tb; j0 < te && ok; j0 += per) {
      if (cc_len == all_bases()) break;             /* as makescaffold(): nothing can be strictly longer */
      const uint32_t nb = te - j0 < per ? te - j0 : per;
      clear_walk_slots(nb);       /* between ccs the slots behind the first hold other scratch */
      used = nb;
      uint32_t r_len, r_t;
      bool r_bad;
      const bool d16 = LDS && M.d16;
      if (L == 8 && nv <= 64 && C.small_masks) {
        if (d16) walks_clean_batch_small<true>(j0, nb, r_len, r_t, r_bad);
        else walks_clean_batch_small<false>(j0, nb, r_len, r_t, r_bad);
      } else if (d16) walks_clean_batch<L, true>(j0, nb, r_len, r_t, r_bad);
      else walks_clean_batch<L, false>(j0, nb, r_len, r_t, r_bad);
      if (W::ballot(r_bad)) { ok = false; break; }
      /* first strictly longest walk in terminal order, algorithms.c:826-832 */
      uint32_t wg = GTS_NONE;
      for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t len = W::bcast(r_len, k * L);
        if ((uint64_t)len > cc_len) { cc_len = len; wg = k; }
      }
      if (wg != GTS_NONE) {
        auto sbase = M.wbase + wg * sbytes;
        auto emap = (GTS_P(idx_t))(sbase + 2 * p4);
        auto par = (GTS_P(idx_t))(sbase + 2 * p4 + p2);
        const uint32_t start = W::uni((uint32_t)M.term[j0 + wg]);
        uint32_t cv = W::bcast(r_t, wg * L), n = 0;
        while (cv != start) {
          const uint32_t re = W::uni((uint32_t)emap[cv]);
          M.cc_best[n++] = (idx_t)re;
          cv = W::uni((uint32_t)par[cv]);
        }
        cc_n = n;
        if (best_start) *best_start = start;
      }
      nfast += nb;
    }
    if (used) clear_walk_slots(1);   /* distmap: unset between walks */
    if (!ok) nfast = nfast0;
    return ok;
  }


  /* ---- the walks of a cc on a team of wavefronts (global memory) ---------------
     Slab of a workgroup: per wavefront GTS_WALK_SLOTS_MAX walk slots -- label
     (f32), tree-path length (u64), edgemap, parent (u32) per contig and a bitmap
     over the sweep positions -- and a buffer for the best path it has seen. */
  static GTS_HD uint64_t team_slot_bytes(uint32_t nv)
  {
    const uint64_t a = 16, n = nv;
    return ((n * 4 + a - 1) / a) * a * 3 + ((n * 8 + a - 1) / a) * a + ((((n + 31) / 32) * 4 + a - 1) / a) * a;
  }
  static GTS_HD uint64_t team_wave_bytes(uint32_t nv)
  {
    return GTS_WALK_SLOTS_MAX * team_slot_bytes(nv) + (((uint64_t)nv * 4 + 15) / 16) * 16;
  }

  /* walks_clean_batch for a component in global memory: the next labelled
     vertex of a sweep comes from the slot's bitmap over the positions (a walk
     costs its reachable set, not the component: create_walk_clean), eight words
     a step */
  /* pb_lds: LDS address of the batch's position bitmaps (pb_stride bytes a walk)
     or GTS_NONE: the bitmaps of the slots in global memory.  A sweep step is a
     chain of dependent accesses -- the bitmap word, the contig's label and list
     bounds, its arcs, the labels at their ends, the bit of a new label (an atomic
     the next step's read waits for): with the bitmap in LDS two of its five round
     trips to L2 are gone. */
  template <uint32_t L>
  GTS_HD void walks_clean_batch_global(uint32_t j0, uint32_t nb, char *slots, uint32_t pb_lds, uint32_t pb_stride,
                                       uint32_t coff_lds, uint64_t &r_len, uint32_t &r_t, bool &r_bad)
  {
    const uint32_t lane = W::lane(), g = lane / L, a = lane % L;
    const uint32_t gsh = g * L;
    const uint64_t gm = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
    const uint64_t a16 = 16, p4 = (((uint64_t)nv * 4 + a16 - 1) / a16) * a16, p8 = (((uint64_t)nv * 8 + a16 - 1) / a16) * a16;
    const uint32_t nw = (nv + 31) / 32;
    bool active = g < nb;
    char *sbase = slots + (uint64_t)(active ? g : 0u) * team_slot_bytes(nv);
    float *dist = (float *)sbase;
    uint64_t *plen = (uint64_t *)(sbase + p4);
    uint32_t *emap = (uint32_t *)(sbase + p4 + p8);
    uint32_t *par = (uint32_t *)(sbase + 2 * p4 + p8);
    uint32_t *pbits = (uint32_t *)(sbase + 3 * p4 + p8);
#if defined(__HIPCC__)
    typedef uint32_t __attribute__((address_space(3))) *l32p;
    const bool pl = pb_lds != GTS_NONE;
    const l32p lbits = (l32p)(uintptr_t)(pl ? pb_lds + (active ? g : 0u) * pb_stride : 0u);
    /* the list bounds from the team's copy in LDS (coff_lds; team_share_t) */
    const bool cl = coff_lds != GTS_NONE;
    const l32p lcoff = (l32p)(uintptr_t)(cl ? coff_lds : 0u);
#define GTS_TCOFF(i) (cl ? (uint32_t)lcoff[i] : (uint32_t)M.coff[i])
#else
#define GTS_TCOFF(i) ((uint32_t)M.coff[i])
#endif
    const uint32_t start = active ? (uint32_t)M.term[j0 + g] : 0u;
    const uint32_t sb0 = GTS_TCOFF(start) - M.e0, se0 = GTS_TCOFF(start + 1) - M.e0;
    bool hs = false, ha = false;
    for (uint32_t cur0 = sb0; W::ballot(active && cur0 < se0); cur0 += L) {
      const uint32_t ce = cur0 + a;
      const bool in = active && ce < se0;
      const uint32_t fs = edge_bits(in ? ce : sb0);
      const bool live = in && !bits_marked(fs);
      const bool sense = (fs & GTS_F_SENSE) != 0;
      const uint64_t bs = W::ballot(live && sense), ba = W::ballot(live && !sense);
      hs |= ((bs >> gsh) & gm) != 0;
      ha |= ((ba >> gsh) & gm) != 0;
    }
    bool bad = active && hs && ha;
    if (bad || !(hs || ha)) active = false;
    const bool forward = hs == ((M.gorient[start] & 3u) == 2);
    const int32_t step = forward ? 1 : -1;
    int32_t wcur = (int32_t)(M.tpos[start] >> 5);
    bool have_u = active, is_start = true;
    uint32_t u = start, cur = sb0, ub = sb0, ue = se0;
    bool du = hs, us = false, ua = false;
    int64_t ndu = 0;
    uint64_t plu = (uint64_t)M.cseq[start], best_len = 0;
    uint32_t pending = 0, best_t = GTS_NONE, steps = 0;
    while (W::ballot(active)) {
      ++steps;
      /* (1) the next labelled position: eight words of the bitmap a step */
      const bool scan = active && !have_u;
      const int32_t wi = wcur + (int32_t)a * step;
      const bool inr = scan && wi >= 0 && wi < (int32_t)nw;
      uint32_t word = 0;
#if defined(__HIPCC__)
      if (pl) { if (inr) word = lbits[wi]; } else
#endif
      if (inr) word = pbits[wi];
      const uint64_t rb = (W::ballot(word != 0) >> gsh) & gm;
      const uint32_t k = rb ? W::ctz(rb) : 0u;
      const uint32_t wk = W::shfl(word, gsh + k);          /* the word of the group's lane k */
      if (scan) {
        if (rb) {
          const uint32_t bit = forward ? W::ctz((uint64_t)wk) : 31u - W::clz32(wk);
          wcur += (int32_t)k * step;
#if defined(__HIPCC__)
          if (pl) { if (a == 0) __hip_atomic_fetch_and(lbits + wcur, ~(1u << bit), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); } else
#endif
          if (a == 0) W::and_bits(pbits + wcur, ~(1u << bit));
          u = M.topo[(uint32_t)wcur * 32u + bit];
          have_u = true; is_start = false; us = ua = false;
          --pending;
          ub = cur = GTS_TCOFF(u) - M.e0; ue = GTS_TCOFF(u + 1) - M.e0;
          du = ((M.gorient[u] & 3u) == 2) == forward;
          ndu = (int64_t)dist[u];
          plu = plen[u];
        } else {
          wcur += (int32_t)L * step;
          if (wcur < 0 || wcur >= (int32_t)nw) active = false;
        }
      }
      /* (2) up to L arcs of the current vertex */
      const bool proc = active && have_u;
      const uint32_t ce = cur + a;
      const bool in = proc && ce < ue;
      const uint32_t cec = in ? ce : ub;
      const uint32_t fs = edge_bits(cec);
      const bool live = in && !bits_marked(fs);
      const bool sense = (fs & GTS_F_SENSE) != 0;
      const bool arc = live && sense == du;
      const uint32_t v = M.cend[cec];
      const int64_t w = (int64_t)dist_of(cec);
      const float cand = (float)(ndu + w);
      const float old = dist[v];
      const bool imp = arc && (old == GTS_DIST_UNSET || old > cand);
      const bool tie = arc && !imp && old == cand;
      const bool fresh = imp && old == GTS_DIST_UNSET;
      /* the node of a child of the start carries the distance itself, not its
         float image (create_walk_clean: nd[v] = w): the same below 2^24 */
      const bool wide = arc && is_start && !(w > -16777216 && w < 16777216);
      if (imp) {
        dist[v] = cand;
        emap[v] = ce;
        par[v] = u;
        plen[v] = plu + (uint64_t)M.cseq[v];
      }
      if (fresh) {
        const uint32_t tp = M.tpos[v];
#if defined(__HIPCC__)
        if (pl) __hip_atomic_fetch_or(lbits + (tp >> 5), 1u << (tp & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); else
#endif
        W::or_bits(pbits + (tp >> 5), 1u << (tp & 31));
      }
      const uint64_t bs = W::ballot(live && sense), ba = W::ballot(live && !sense);
      const uint64_t fm = W::ballot(fresh), tm = W::ballot(tie || wide);
      us |= ((bs >> gsh) & gm) != 0;
      ua |= ((ba >> gsh) & gm) != 0;
      pending += W::popc((fm >> gsh) & gm);
      if ((tm >> gsh) & gm) bad = true;
      if (proc) {
        cur += L;
        if (cur >= ue) {
          if (!is_start && !(us && ua)) {
            if (plu > best_len) { best_len = plu; best_t = u; }
            else if (plu == best_len && best_t != GTS_NONE) bad = true;
          }
          have_u = false;
          if (pending == 0) active = false;
        }
      }
      if (bad) active = false;
      W::fence();
    }
    if (lane == 0) W::add64((uint64_t *)C.team_stat + 2, steps);
    r_len = best_len; r_t = best_t; r_bad = bad;
#undef GTS_TCOFF
  }

  /* this wavefront's share of the walks of the cc [tb, te): batches
     team_wave, team_wave + team_waves, ...; result in the team's control block */
  /* few terminals: more lanes per walk (the lists of a hub vertex and the
     bitmap are swept L entries a step) */
  GTS_HD void team_share(uint32_t tb, uint32_t te)
  {
    if (te - tb <= 2) team_share_t<32>(tb, te);
    else if (te - tb <= 4) team_share_t<16>(tb, te);
    else team_share_t<8>(tb, te);
  }
  template <uint32_t L>
  GTS_HD void team_share_t(uint32_t tb, uint32_t te)
  {
    const uint32_t lane = W::lane();
    constexpr uint32_t G = W::WIDTH / L;
    const uint64_t sbytes = team_slot_bytes(nv);
    char *wbase = team_base + (uint64_t)team_wave * team_wave_bytes(nv);
    uint32_t *path = (uint32_t *)(wbase + GTS_WALK_SLOTS_MAX * sbytes);
    const uint64_t a16 = 16, p4 = (((uint64_t)nv * 4 + a16 - 1) / a16) * a16, p8 = (((uint64_t)nv * 8 + a16 - 1) / a16) * a16;
    const uint32_t nw = (nv + 31) / 32;
    /* this wavefront's G bitmaps in the team's spare LDS, if all of them fit */
    const uint32_t pb_stride = ((nw * 4u + 15u) / 16u) * 16u;
    const uint32_t pb_lds = tl_pbits != GTS_NONE && (uint64_t)team_waves * G * pb_stride <= tl_pbits_bytes
                                ? tl_pbits + team_wave * G * pb_stride : GTS_NONE;
    /* behind the bitmaps the component's list offsets, if they fit as well: copied
       by all wavefronts for every cc (33 KB, ~2 us: what else uses this LDS between
       two ccs is not tracked); one more barrier */
    uint32_t coff_lds = GTS_NONE;
#if defined(__HIPCC__)
    if (C.team_coff && pb_lds != GTS_NONE &&
        (uint64_t)team_waves * G * pb_stride + ((uint64_t)nv + 1) * 4 <= tl_pbits_bytes) {
      coff_lds = tl_pbits + team_waves * G * pb_stride;
      uint32_t __attribute__((address_space(3))) *lc =
          (uint32_t __attribute__((address_space(3))) *)(uintptr_t)coff_lds;
      for (uint32_t i = team_wave * W::WIDTH + lane; i <= nv; i += team_waves * W::WIDTH) lc[i] = M.coff[i];
      W::team_barrier();
    }
#endif
    uint64_t best_len = 0;
    uint32_t best_j = GTS_NONE, best_n = 0;
    bool bad = false;
    for (uint32_t b = team_wave; b * G < te - tb && !bad; b += team_waves) {
      const uint32_t j0 = tb + b * G, nb = te - j0 < G ? te - j0 : G;
      const uint64_t tc0 = W::clock();
      for (uint32_t k = 0; k < nb; ++k) {
        float *dist = (float *)(wbase + k * sbytes);
        uint32_t *pb = (uint32_t *)(wbase + k * sbytes + 3 * p4 + p8);
        for (uint32_t s = lane; s < nv; s += W::WIDTH) dist[s] = GTS_DIST_UNSET;
#if defined(__HIPCC__)
        if (pb_lds != GTS_NONE) {
          uint32_t __attribute__((address_space(3))) *lb =
              (uint32_t __attribute__((address_space(3))) *)(uintptr_t)(pb_lds + k * pb_stride);
          for (uint32_t s = lane; s < nw; s += W::WIDTH) lb[s] = 0;
        } else
#endif
        for (uint32_t s = lane; s < nw; s += W::WIDTH) pb[s] = 0;
      }
      W::fence();
      const uint64_t tc1 = W::clock();
      uint64_t r_len;
      uint32_t r_t;
      bool r_bad;
      walks_clean_batch_global<L>(j0, nb, wbase, pb_lds, pb_stride, coff_lds, r_len, r_t, r_bad);
      const uint64_t tc2 = W::clock();
      if (lane == 0) {
        W::add64((uint64_t *)C.team_stat + 1, 1); W::add64((uint64_t *)C.team_stat + 4, tc1 - tc0);
        W::add64((uint64_t *)C.team_stat + 5, tc2 - tc1);
      }
      if (W::ballot(r_bad)) { bad = true; break; }
      uint32_t wg = GTS_NONE;
      for (uint32_t k = 0; k < nb; ++k) {
        const uint64_t len = (uint64_t)W::bcast((uint32_t)r_len, k * L) |
                             (uint64_t)W::bcast((uint32_t)(r_len >> 32), k * L) << 32;
        if (len > best_len) { best_len = len; wg = k; }
      }
      if (wg != GTS_NONE) {
        const uint32_t *emap = (const uint32_t *)(wbase + wg * sbytes + p4 + p8);
        const uint32_t *par = (const uint32_t *)(wbase + wg * sbytes + 2 * p4 + p8);
        const uint32_t start = W::uni((uint32_t)M.term[j0 + wg]);
        uint32_t cv = W::bcast(r_t, wg * L), n = 0;
        while (cv != start) {
          const uint32_t re = W::uni(emap[cv]);
          if (lane == 0) path[n] = re;
          ++n;
          cv = W::uni(par[cv]);
        }
        best_j = j0 + wg; best_n = n;
      }
      if (lane == 0) W::add64((uint64_t *)C.team_stat + 6, W::clock() - tc2);
      nfast += nb;
    }
    if (lane == 0) {
      team->len[team_wave] = best_len; team->j[team_wave] = best_j; team->n[team_wave] = best_n;
      team->bad[team_wave] = bad ? 1u : 0u;
    }
    W::fence();
  }

  /* wavefront 0: posts the cc, takes its share, picks the first strictly
     longest walk in terminal order (algorithms.c:826-832).  False if a walk
     met a tie: the caller makes the walks of this cc one by one. */
  GTS_HD bool cc_walks_team(uint32_t tb, uint32_t te, uint64_t &cc_len, uint32_t &cc_n)
  {
    const uint32_t lane = W::lane();
    if (lane == 0) { team->kind = 1; team->tb = tb; team->te = te; }
    const uint64_t tb0 = W::clock();
    W::team_barrier();
    const uint64_t tb1 = W::clock();
    team_share(tb, te);
    const uint64_t tb2 = W::clock();
    W::team_barrier();
    if (lane == 0) { W::add64((uint64_t *)C.team_stat, 1); W::add64((uint64_t *)C.team_stat + 7, (tb1 - tb0) + (W::clock() - tb2)); }
    uint64_t best = 0;
    uint32_t bj = GTS_NONE, bw = 0;
    bool bad = false;
    for (uint32_t w = 0; w < team_waves; ++w) {
      bad |= team->bad[w] != 0;
      const uint64_t len = team->len[w];
      const uint32_t j = team->j[w];
      if (j != GTS_NONE && (len > best || (len == best && j < bj))) { best = len; bj = j; bw = w; }
    }
    if (bad) { if (lane == 0) W::add64((uint64_t *)C.team_stat + 3, 1); return false; }
    if (bj != GTS_NONE && best > cc_len) {
      const uint32_t *path = (const uint32_t *)(team_base + (uint64_t)bw * team_wave_bytes(nv) +
                                                GTS_WALK_SLOTS_MAX * team_slot_bytes(nv));
      const uint32_t n = team->n[bw];
      for (uint32_t k = lane; k < n; k += W::WIDTH) M.cc_best[k] = path[k];
      cc_len = best; cc_n = n;
      W::fence();
    }
    return true;
  }

  /* a participant's share of the open job: terminals off the counter, walked one by one
     on this program's scratch; its best walk (first strictly longest of its own, in the
     order it took them = terminal order) into the job */
  GTS_HD void hub_take_walks(GtsHelpJob *J, uint32_t te)
  {
    if constexpr (LDS && W::WIDTH == 64) {
      const uint32_t lane = W::lane(), me = hub_me;
      uint64_t my_len = 0;
      uint32_t my_n = 0, my_j = GTS_NONE, my_start = 0, made = 0;
      for (;;) {
        uint32_t j = 0;
        if (lane == 0) j = W::hub_add(&J->next, 1u);
        j = W::uni(j);
        if (j >= te) break;
        if (my_len == all_bases()) continue;        /* nothing can be strictly longer */
        const uint64_t len0 = my_len;
        const uint32_t start = W::uni(M.term[j]);
        no_reference = true;
        create_walk(start, my_len, my_n);
        ++made;
        if (needs_reference) { needs_reference = false; if (lane == 0) W::hub_store(&J->need_ref, 1u); break; }
        if (my_len != len0) { my_j = j; my_start = start; }
      }
      if (lane == 0) {
        J->best_len[me] = my_len; J->best_j[me] = my_j; J->best_n[me] = my_n; J->best_start[me] = my_start;
        J->best_path[me] = W::lds_addr(&M.cc_best[0]);
        if (made) W::hub_add(&J->walks, made);
      }
      W::fence();
    }
  }

  /* the owner's side: opens a job for the cc [tb, te) if the slot is free (0: it is not) */
  GTS_HD uint32_t hub_open(uint32_t tb, uint32_t te)
  {
    if constexpr (LDS && W::WIDTH == 64) {
      GtsHelpJob *J = hub;
      uint32_t s = 0;
      if (W::lane() == 0) {
        s = W::hub_load(&J->seq);
        if (!(s & 1u) && W::hub_cas(&J->seq, s, s + 1u) == s) {
          J->comp = c; J->tb = tb; J->te = te; J->clean = clean ? 1u : 0u; J->nv = nv;
          /* (J->active is NOT set here: it counts helpers between their increment and
             their decrement, and one of them may still be on its way out of the last job --
             a store of 0 under its feet left the count at -1 after its decrement, and
             the owner of this job then waited for 0 for ever: one makescaffold call in
             ~70 on the inversions workload never returned) */
          J->next = tb; J->need_ref = 0;
          for (uint32_t w = 0; w < GTS_HUB_WAVES; ++w) { J->best_j[w] = GTS_NONE; J->best_len[w] = 0; J->best_n[w] = 0; }
          J->M = M;
          W::hub_release();
          W::hub_store(&J->ready, s + 1u);
          s = s + 1u;
        } else
          s = 0;
      }
      return W::uni(s);
    } else
      return 0;
  }
  /* ... and closes it: waits for the participants (they finish: a walk waits for
     nothing), takes the first strictly longest walk in terminal order over its own
     (cc_len, cc_n, cc_start, cc_j: in M.cc_best) and theirs, copies a helper's path.
     False: a walk needs the reference's search -- nothing is kept, the caller starts the
     cc over, alone */
  GTS_HD bool hub_finish(uint32_t s, uint64_t &cc_len, uint32_t &cc_n, uint32_t &cc_start, uint32_t cc_j)
  {
    if constexpr (LDS && W::WIDTH == 64) {
      typedef typename GtsCompMemT<LDS>::idx_t idx_t;
      GtsHelpJob *J = hub;
      const uint32_t lane = W::lane();
      /* the gate first, then the wait: a helper counts itself in and THEN looks at
         `ready` again (try_help), this side closes `ready` and THEN reads the count --
         one of the two sees the other, so nobody joins a job whose results are being
         read (it would walk on a graph that is being marked, or on the next job's) */
      if (lane == 0) { W::hub_store(&J->ready, 0u); W::hub_fence(); while (W::hub_load(&J->active) != 0) W::nap(); }
      W::hub_acquire();
      W::fence();
      uint64_t best = cc_len;
      uint32_t bj = cc_n ? cc_j : GTS_NONE, bw = GTS_NONE;
      for (uint32_t w = 0; w < GTS_HUB_WAVES; ++w) {
        const uint32_t j = W::uni(W::hub_load(&J->best_j[w]));
        const uint64_t len = (uint64_t)W::uni64((int64_t)J->best_len[w]);
        if (j != GTS_NONE && len != 0 && (len > best || (len == best && j < bj))) { best = len; bj = j; bw = w; }
      }
      const bool redo = W::uni(W::hub_load(&J->need_ref)) != 0;
      if (!redo && bw != GTS_NONE) {
        const uint32_t n = W::uni(W::hub_load(&J->best_n[bw]));
        const idx_t __attribute__((address_space(3))) *src =
            (const idx_t __attribute__((address_space(3))) *)(uintptr_t)W::uni(W::hub_load(&J->best_path[bw]));
        for (uint32_t k = lane; k < n; k += W::WIDTH) M.cc_best[k] = src[k];
        cc_len = best; cc_n = n; cc_start = W::uni(W::hub_load(&J->best_start[bw]));
      }
      nfast += W::uni(W::hub_load(&J->walks));
      W::fence();
      /* close: the helpers give their pages back (the winner's path has been copied) */
      if (lane == 0) { W::hub_store(&J->walks, 0u); W::hub_release(); W::hub_store(&J->seq, s + 1u); }
      return !redo;
    } else
      return true;
  }

  GTS_HD bool create_walk(uint32_t start, uint64_t &cc_len, uint32_t &cc_n)
  {
    /* (the reference's test for a start without any edge, algorithms.c:655,
       cannot fire: a vertex is in a component because it has a live edge) */
    const uint64_t t0 = tick();
    if (C.fast_walks && (clean ? create_walk_clean(start, cc_len, cc_n)
                               : create_walk_fast(start, cc_len, cc_n))) {
      ++nfast; tfast += tick() - t0; return true;
    }
    const uint64_t t1 = tick();
    tfast += t1 - t0;
    if (no_reference) { needs_reference = true; return true; }
    ++nslow;
    const bool ok = create_walk_reference(start, cc_len, cc_n);
    tslow += tick() - t1;
    return ok;
  }

  /* bases of all contigs of the component (computed once) */
  GTS_HD uint64_t all_bases()
  {
    if (ubases == ~0ull) {
      uint64_t sum = 0;
      for (uint32_t s = W::lane(); s < nv; s += W::WIDTH) sum += (uint64_t)M.cseq[s];
      for (uint32_t off = W::WIDTH / 2; off > 0; off >>= 1) sum += W::shfl64(sum, W::lane() ^ off);
      ubases = (uint64_t)W::uni64((int64_t)sum);
    }
    return ubases;
  }

  /* ---- ref algorithms.c:767-868 (after its removecycles call) ---- */
  GTS_HD void makescaffold()
  {
    const uint32_t lane = W::lane();
    if (!reuse_cc) calc_cc();   /* else run() just computed the same ccs */
    for (uint32_t s = lane; s < nv; s += W::WIDTH) { M.st_dir[s] = 0; M.tight[s] = 0; }
    if constexpr (!LDS)   /* position bitmap of create_walk_clean's sparse sweep */
      for (uint32_t k = lane; k < (nv + 31) / 32; k += W::WIDTH) ((uint32_t *)&M.st_cur[0])[k] = 0;
    W::fence();
    auto ccoff = M.ccoff;
    /* walks fan out (try_defer): a large component from the start, any component
       of defer_ref_min_nv contigs from the first cc with a walk that needs the
       reference's search.  (One call site: the body is inlined.) */
    bool want_defer = C.defer_min_nv && nv >= C.defer_min_nv, forced = false;
    if (!clean && C.defer_unclean_work && (C.defer_min_nv || C.defer_ref_min_nv) && nterm >= 4 &&
        (uint64_t)nterm * nv >= C.defer_unclean_work && W::peek(C.ndeferred) != 0) {
      want_defer = true; forced = true;
    }
    bool may_late = C.defer_ref_min_nv && nv >= C.defer_ref_min_nv;
    uint32_t i0 = 0;
    for (;;) {
      if (want_defer) {
        if (try_defer(i0, forced)) { deferred_late = true; return; }
        /* (the task tables are full, or the component is too small to gain:) here after all */
        want_defer = false;
        if (forced) may_late = false;
      }
      for (uint32_t i = i0; i < ncc && !err && !want_defer; ++i) {
        const uint32_t tb = W::uni(ccoff[i]), te = W::uni(ccoff[i + 1]);
        if (te - tb == 1) lonesome(W::uni(M.term[tb]));
        if (te - tb > 1) {
          uint64_t cc_len = 0;
          uint32_t cc_n = 0, cc_start = 0;
          bool batched = false;
          if constexpr (LDS) {
            if (C.fast_walks && C.batch_walks && M.wslots >= 2 && (clean || C.batch_walks >= 2)) {
              const uint64_t tw0 = tick();
              batched = clean ? cc_walks_batched<GTS_WALK_LANES>(tb, te, cc_len, cc_n, &cc_start)
                              : cc_walks_batched_unclean(tb, te, cc_len, cc_n, &cc_start);
              tfast += tick() - tw0;
              if (!batched) { cc_len = 0; cc_n = 0; }
            }
          }
          if constexpr (W::TEAM) {
            if (clean && C.fast_walks && team) {
              const uint64_t tw0 = tick();
              batched = cc_walks_team(tb, te, cc_len, cc_n);
              tfast += tick() - tw0;
              if (!batched) { cc_len = 0; cc_n = 0; }
            }
          }
          /* the walks one by one; over the workgroup's wavefronts when there is a job slot
             (GtsHelpJob): the terminals then come off the job's counter, whoever passes
             by takes some, and hub_finish picks the cc's walk over all of them.  (One
             call of create_walk for both ways: its body is the bulk of the kernel.) */
          uint32_t job = 0;
          if constexpr (LDS && W::WIDTH == 64) {
            if (!batched && hub && C.fast_walks && local_marks && te - tb >= 3) job = hub_open(tb, te);
          }
          for (int attempt = 0; attempt < 2 && !batched; ++attempt) {
            uint32_t jn = tb, cc_j = GTS_NONE;
            bool job_ref = false;
            for (;;) {
              uint32_t j = jn++;
              if constexpr (LDS && W::WIDTH == 64) {
                if (job) { j = 0; if (W::lane() == 0) j = W::hub_add(&hub->next, 1u); j = W::uni(j); }
              }
              if (j >= te) break;
              /* a walk is a simple path inside the component: none can be STRICTLY
                 longer (algorithms.c:826) than one that holds every contig of it --
                 the usual outcome on a clean chain, whose other end needs no walk */
              if (cc_len == all_bases()) { if (job) continue; break; }
              no_reference = may_late || job != 0;
              const uint64_t len0 = cc_len;
              const uint32_t wstart = W::uni(M.term[j]);
              if (!create_walk(wstart, cc_len, cc_n)) break;
              if (cc_len != len0) { cc_start = wstart; cc_j = j; }
              no_reference = false;
              if (needs_reference) {
                needs_reference = false;
                if (job) { job_ref = true; break; }
                /* this walk needs the reference's search: the walks of this cc and
                   of the ccs after it become tasks */
                want_defer = true; forced = true; i0 = i;
                break;
              }
            }
            if (!job) break;
            if constexpr (LDS && W::WIDTH == 64) {
              if (job_ref && W::lane() == 0) W::hub_store(&hub->need_ref, 1u);
              const bool kept = hub_finish(job, cc_len, cc_n, cc_start, cc_j);
              job = 0;
              if (kept) break;
              cc_len = 0; cc_n = 0;      /* once more, alone: the second pass handles the reference search */
            }
          }
          if (err || want_defer) break;
          if (local_marks) { if (cc_n) { mark_best_lds(cc_n, cc_start); any_scaffold_marks = true; } }
          else mark_best(M.cc_best, cc_n);
        }
      }
      if (!want_defer) break;
    }
  }

  /* marks a cc's best walk, algorithms.c:835-848 (a walk without edges is
     undefined behaviour there and is left unmarked here) */
  template <class BestPtr>
  GTS_HD void mark_best(BestPtr best, uint32_t cc_n)
  {
    const uint32_t lane = W::lane();
    if (cc_n == 0) return;
    for (uint32_t k = lane; k < cc_n; k += W::WIDTH) {
      const uint32_t ce = best[k];
      const uint32_t p = C.cgpos[e0g + ce], t = C.G.twin[p];
      M.cstate[ce] = GIS_SCAFFOLD;
      C.G.state[p] = GIS_SCAFFOLD;
      C.G.state[t] = GIS_SCAFFOLD;
      const uint32_t ct = C.cmap[t];
      if (ct != GTS_NONE) {
        M.cstate[ct - e0g] = GIS_SCAFFOLD;
      }
      M.vst[M.cend[ce]] = GIS_SCAFFOLD;
      M.vst[M.cstart[ce]] = GIS_SCAFFOLD;
    }
    /* a marked twin that turns SCAFFOLD is a new live arc, but not a new
       arc of D (d_arc counts an edge whose twin is live): the component
       stays clean, its sweep order holds */
    W::fence();
  }

  /* lonesome test of a cc with one terminal, algorithms.c:790-807 */
  GTS_HD void lonesome(uint32_t v)
  {
    const uint32_t lane = W::lane();
    const uint32_t eb = eoff(v), ee = eoff(v + 1);
    bool any_live = false;
    for (uint32_t base = eb; base < ee; base += W::WIDTH) {
      const uint32_t ce = base + lane;
      const bool live = ce < ee && !edge_marked(ce);
      any_live |= W::ballot(live) != 0;
    }
    if (!any_live) M.vst[v] = GIS_SCAFFOLD;
    W::fence();
  }

  /* ---- fan-out of the walks of a large component -------------------------
     The terminals of every cc are found before the first walk
     (algorithms.c:784) and SCAFFOLD is an unmarked state, so marking the
     best walk of a cc changes what a later walk sees only where it revives a
     marked twin: a new arc in the list of the walk edge's end vertex.  A
     search inspects the lists of the vertices it labels and nothing else, so
     a walk computed BEFORE such marks is still the reference's walk as long
     as no revived arc starts at a vertex it labelled.
     The component program therefore stops after its terminal search and
     publishes one task per terminal.  Rounds of (all pending walks in
     parallel, one wave each) + (select_walks: the ccs in order, one wave per
     component) follow: select_walks keeps a bitmap of the vertices that
     gained an arc in this pass, accepts a cc only if none of its walks
     labelled such a vertex, and otherwise hands the ccs from there on to the
     next round -- only the walks that did touch such a vertex run again.  The
     first pending cc of a pass is always accepted, so the rounds end. */
  /* first_cc: the ccs before it are decided already (their marks are in the
     working copy); forced: no size test (makescaffold met a walk that needs the
     reference's search) */
  GTS_HD bool try_defer(uint32_t first_cc = 0, bool forced = false)
  {
    const uint32_t lane = W::lane();
    if (!forced) {
      if (!C.defer_min_nv || nv < C.defer_min_nv) return false;
      if (nterm < 2) { nodefer = 2; return false; }
      if ((uint64_t)nterm * nv < C.defer_min_work) { nodefer = 1; return false; }
    }
    auto ccoff = M.ccoff;
    uint32_t npend = 0;
    for (uint32_t i = first_cc; i < ncc; ++i) {
      const uint32_t tb = W::uni(ccoff[i]), te = W::uni(ccoff[i + 1]);
      if (te - tb >= 2) npend += te - tb;
    }
    /* no cc with two terminals: no walk to fan out, and nothing would make the
       host run the select pass that does the lonesome test of a deferred
       component (found by tests/golden/handmade/cycle) */
    if (npend == 0) { nodefer = 2; return false; }
    const uint32_t nw = (nv + 31) / 32;   /* labelled-vertex bitmap of a task */
    const uint64_t t0 = W::alloc(C.ntasks, nterm);
    if (t0 + nterm > C.task_cap) return false;
    const uint64_t p0 = W::alloc(C.path_used, (uint64_t)nterm * nw);
    if (p0 + (uint64_t)nterm * nw > C.path_cap) { nodefer = 3; return false; }   /* walk in place */
    const uint32_t kl = W::uni((uint32_t)C.comp_klass[c]);
    const uint64_t q0 = C.tq_base[kl] + W::alloc(C.tq_cnt + kl, npend);
    npend = 0;
    for (uint32_t i = first_cc; i < ncc; ++i) {
      const uint32_t tb = W::uni(ccoff[i]), te = W::uni(ccoff[i + 1]);
      const bool skip = te - tb < 2;
      for (uint32_t j = tb + lane; j < te; j += W::WIDTH) {
        const uint64_t t = t0 + j;
        C.task_comp[t] = c;
        C.task_start[t] = M.term[j];
        C.task_skip[t] = skip ? 1 : 0;
        C.task_roff[t] = p0 + (uint64_t)j * nw;
        C.task_poff[t] = 0;
        C.task_len[t] = 0;
        C.task_n[t] = 0;
        if (!skip) C.tq[q0 + npend + (j - tb)] = (uint32_t)t;
      }
      if (!skip) npend += te - tb;
    }
    /* the tasks and the select pass work on the global graph */
    if (local_marks) flush_local_marks();
    /* what the tasks and the select pass read (no-op copies when M already
       points into the global arrays) */
    for (uint32_t s = lane; s < nv; s += W::WIDTH) {
      C.gorient[s0 + s] = M.gorient[s];
      C.topo[s0 + s] = M.topo[s];
      C.tpos[s0 + s] = M.tpos[s];
      C.term[s0 + s] = M.term[s];
    }
    for (uint32_t i = lane; i <= ncc; i += W::WIDTH) C.ccoff[s0 + c + i] = ccoff[i];
    /* marks of removecycles live in the working copy: publish them */
    for (uint32_t s = lane; s < nv; s += W::WIDTH) C.vst[s0 + s] = M.vst[s];
    for (uint32_t k = lane; k < M.ne; k += W::WIDTH) {
      C.cstate[e0g + k] = M.cstate[k];
      C.cflags[e0g + k] = M.cflags[k];
    }
    if (lane == 0) {
      C.defer_flag[c] = clean ? 2 : 1;
      C.comp_task0[c] = (uint32_t)t0;
      C.comp_ncc[c] = ncc;
      C.comp_nterm[c] = nterm;
      C.comp_next_cc[c] = first_cc;
      /* (comp_ring[2c], [2c + 1] are zero: the host clears the table before the launch) */
    }
    const uint64_t dl = W::alloc(C.ndeferred, 1);
    if (lane == 0) C.defer_list[dl] = c;
    W::fence();
    return true;
  }

  /* one deferred walk: the program object is constructed on a staged copy of
     the component's published state (with gorient / topo / tpos while the
     component is clean) */
  GTS_HD void walk_task(uint64_t t)
  {
    const uint32_t lane = W::lane();
    clean = W::uni((uint32_t)C.defer_flag[c]) == 2;   /* revived twins do not change D */
    const uint32_t nw = (nv + 31) / 32;
    reach_bits = C.paths + C.task_roff[t];
    no_reference = !C.task_reference;
    for (uint32_t k = lane; k < nw; k += W::WIDTH) reach_bits[k] = 0;
    for (uint32_t s = lane; s < nv; s += W::WIDTH) { M.st_dir[s] = 0; M.tight[s] = 0; }   /* as makescaffold */
    if constexpr (!LDS)   /* (host harness: tasks run on the global arrays) */
      for (uint32_t k = lane; k < (nv + 31) / 32; k += W::WIDTH) ((uint32_t *)&M.st_cur[0])[k] = 0;
    W::fence();
    uint64_t len = 0;
    uint32_t n = 0;
    const uint32_t start = W::uni(C.task_start[t]);
    create_walk(start, len, n);
    if (lane == 0) W::or_bits(reach_bits + (start >> 5), 1u << (start & 31));
    uint64_t po = 0;
    if (needs_reference) { len = 0; n = GTS_NONE; }   /* select_walks runs the reference search */
    else if (n) {
      po = W::alloc(C.path_used, n);
      if (po + n > C.path_cap) { err = GTS_CERR_PATH_OVERFLOW; n = 0; len = 0; }
      for (uint32_t k = lane; k < n; k += W::WIDTH) C.paths[po + k] = M.cc_best[k];
    }
    if (lane == 0) {
      C.task_len[t] = len;
      C.task_n[t] = n;
      C.task_poff[t] = po;
      C.task_skip[t] = 1;
      if (err) C.cerr[c] = err;
      if (nfast) W::count_n(C.stat_fast + c, nfast);
      if (nslow) W::count_n(C.stat_slow + c, nslow);
      W::add64(C.tstat + 5 * (uint64_t)c + 2, tfast);
      if (tslow) W::add64(C.tstat + 5 * (uint64_t)c + 3, tslow);
      W::add64(C.tstat + 5 * (uint64_t)c + 4, npops);
      W::add64((uint64_t *)C.task_bytes, (uint64_t)M.ne * 19 + (uint64_t)nv * 27);
    }
    W::fence();
  }

  /* true if the walk of task t labelled a vertex of the bitmap wb */
  static GTS_HD bool task_touches(const GtsCompView &C, uint64_t t, uint32_t nv, const uint32_t *wb)
  {
    const uint32_t lane = W::lane(), nw = (nv + 31) / 32;
    const uint32_t *rb = C.paths + C.task_roff[t];
    bool hit = false;
    for (uint32_t base = 0; base < nw && !hit; base += W::WIDTH) {
      const uint32_t k = base + lane;
      hit = W::ballot(k < nw && (rb[k] & wb[k]) != 0) != 0;
    }
    return hit;
  }

  /* One pass over the pending ccs of a deferred component, in order: the
     lonesome test of a cc with one terminal (algorithms.c:790-807), the first
     strictly longest walk in terminal order (algorithms.c:823-832) and its
     marks (algorithms.c:835-848).  Works on the global arrays only; wb is
     the component's scratch bitmap.  Returns true when ccs are left for
     another round. */
  static GTS_HD bool select_walks(const GtsCompView &C, uint32_t c, uint32_t *wb)
  {
    const uint32_t lane = W::lane();
    const uint32_t s0 = C.comp_off[c], e0g = C.coff[s0];
    const uint32_t nv = C.comp_off[c + 1] - s0, nw = (nv + 31) / 32;
    const uint32_t ncc = C.comp_ncc[c], t0 = C.comp_task0[c];
    const uint32_t *ccoff = C.ccoff + s0 + c;
    for (uint32_t k = lane; k < nw; k += W::WIDTH) wb[k] = 0;
    W::fence();
    bool revived_any = false;
    uint32_t i = W::uni(C.comp_next_cc[c]);
    for (; i < ncc; ++i) {
      const uint32_t tb = W::uni(ccoff[i]), te = W::uni(ccoff[i + 1]);
      if (te - tb == 1) {
        const uint32_t v = W::uni(C.term[s0 + tb]);
        const uint32_t eb = W::uni(C.coff[s0 + v]), ee = W::uni(C.coff[s0 + v + 1]);
        bool any_live = false;
        for (uint32_t base = eb; base < ee; base += W::WIDTH) {
          const uint32_t ce = base + lane;
          any_live |= W::ballot(ce < ee && !gts_edge_is_marked(C.cstate[ce])) != 0;
        }
        if (!any_live && lane == 0) C.G.vstate[C.slot_v[s0 + v]] = GIS_SCAFFOLD;
      }
      if (te - tb < 2) continue;
      bool stale = false;
      if (revived_any)
        for (uint32_t j = tb; j < te && !stale; ++j)   /* (a walk left to the reference search is made below) */
          stale = W::uni(C.task_n[t0 + j]) != GTS_NONE && task_touches(C, t0 + j, nv, wb);
      if (stale) break;
      /* walks the tasks left to the reference search: here, in terminal
         order, on the global arrays, with one ring per component */
      for (uint32_t j = tb; j < te; ++j) {
        if (W::uni(C.task_n[t0 + j]) != GTS_NONE) continue;
        const GtsCompMem gm = GtsComponent<W, false>::global_mem(C, c);
        GtsComponent<W, false> prog(C, gm, c);
        prog.qbase = C.comp_ring[2 * (uint64_t)c]; prog.qcap = C.comp_ring[2 * (uint64_t)c + 1];
        uint64_t len = 0;
        uint32_t n = 0;
        prog.create_walk_reference(W::uni(C.task_start[t0 + j]), len, n);
        uint64_t po = 0;
        if (n) {
          po = W::alloc(C.path_used, n);
          if (po + n > C.path_cap) { prog.err = GTS_CERR_PATH_OVERFLOW; n = 0; len = 0; }
          for (uint32_t k = lane; k < n; k += W::WIDTH) C.paths[po + k] = prog.M.cc_best[k];
        }
        if (lane == 0) {
          C.comp_ring[2 * (uint64_t)c] = prog.qbase; C.comp_ring[2 * (uint64_t)c + 1] = prog.qcap;
          C.task_len[t0 + j] = len; C.task_n[t0 + j] = n; C.task_poff[t0 + j] = po;
          if (prog.err) C.cerr[c] = prog.err;
          W::count_n(C.stat_slow + c, 1);
        }
        W::fence();
      }
      uint64_t best = 0;
      uint32_t bj = GTS_NONE;
      for (uint32_t j = tb; j < te; ++j) {
        const uint64_t len = (uint64_t)W::uni64((int64_t)C.task_len[t0 + j]);
        if (len > best) { best = len; bj = j; }
      }
      if (bj == GTS_NONE) continue;
      const uint32_t n = W::uni(C.task_n[t0 + bj]);
      const uint64_t po = C.task_poff[t0 + bj];
      bool revived = false;
      for (uint32_t base = 0; base < n; base += W::WIDTH) {
        const uint32_t k = base + lane;
        bool rv = false;
        if (k < n) {
          const uint32_t ce = C.paths[po + k];
          const uint32_t p = C.cgpos[e0g + ce], t = C.G.twin[p];
          C.G.state[p] = GIS_SCAFFOLD;
          C.G.state[t] = GIS_SCAFFOLD;
          C.cstate[e0g + ce] = GIS_SCAFFOLD;
          const uint32_t ct = C.cmap[t];
          const uint32_t ve = C.cend[e0g + ce];
          if (ct != GTS_NONE) {
            if (gts_edge_is_marked(C.cstate[ct])) {   /* new arc out of ve */
              rv = true;
              W::or_bits(wb + (ve >> 5), 1u << (ve & 31));
            }
            C.cstate[ct] = GIS_SCAFFOLD;
          }
          C.G.vstate[C.slot_v[s0 + ve]] = GIS_SCAFFOLD;
          C.G.vstate[C.slot_v[s0 + C.cstart[e0g + ce]]] = GIS_SCAFFOLD;
        }
        revived |= W::ballot(rv) != 0;
      }
      revived_any |= revived;
      W::fence();
    }
    /* walks of the ccs left that labelled a vertex with a new arc run again */
    for (uint32_t k = i; k < ncc; ++k) {
      const uint32_t tb = W::uni(ccoff[k]), te = W::uni(ccoff[k + 1]);
      if (te - tb < 2) continue;
      for (uint32_t j = tb; j < te; ++j) {
        /* (a walk left to the reference search is made when its cc is due) */
        if (W::uni(C.task_n[t0 + j]) == GTS_NONE || !task_touches(C, t0 + j, nv, wb)) continue;
        const uint32_t kl = W::uni((uint32_t)C.comp_klass[c]);
        const uint64_t q = C.tq_base[kl] + W::alloc(C.tq_cnt + kl, 1);
        if (lane == 0) { C.task_skip[t0 + j] = 0; C.tq[q] = t0 + j; }
      }
    }
    if (lane == 0) {
      C.comp_next_cc[c] = i;
      if (i == ncc) C.defer_flag[c] = 0;
    }
    W::fence();
    return i < ncc;
  }

  /* ---- the program of a clean component (round 4: k_components_fast) ----------
     99.95 % of the components of a scaffold graph are clean (orient + peel): no
     DFS of removecycles can close a cycle, every walk is one sweep of the
     topological order and the walks of a cc are swept side by side.  This is
     that program and nothing else: it touches LDS only until its last lines, so
     when it meets something it does not hold -- a contradiction of strands, a
     cycle of D, a tie in a batch, a component that wants to hand its walks to
     tasks -- it returns false, nothing has left LDS, and the caller hands the
     component to the full program (run(), k_components_pool's cold list).  Kept
     apart from run() so that the kernel built around it holds none of the cycle
     search, the walks one by one, the reference's search and the task tables:
     half the registers, no scratch, twice the wavefronts per CU. */

  /* mark_best() without the global graph: the twin of a walk edge u -> v is the
     one edge of v's list that ends in u (a pair of contigs has one edge per
     direction, ref gt_scaffolder_parser.c:357-378), and it is in the compact
     graph because its twin is live (k_live_union).  best[] lists the walk from
     its far end back to `start`, so the start vertex of best[k] is the end of
     best[k + 1].  ref algorithms.c:835-848 */
  GTS_HD void mark_best_lds(uint32_t cc_n, uint32_t start)
  {
    const uint32_t lane = W::lane();
    for (uint32_t k = lane; k < cc_n; k += W::WIDTH) {
      const uint32_t ce = M.cc_best[k];
      const uint32_t v = M.cend[ce];
      const uint32_t u = k + 1 < cc_n ? (uint32_t)M.cend[M.cc_best[k + 1]] : start;
      M.cstate[ce] = GIS_SCAFFOLD;
      const uint32_t le = M.coff[v + 1] - M.e0;
      for (uint32_t t = M.coff[v] - M.e0; t < le; ++t)
        if (M.cend[t] == u) { M.cstate[t] = GIS_SCAFFOLD; break; }
      M.vst[v] = GIS_SCAFFOLD;
      M.vst[u] = GIS_SCAFFOLD;
    }
    W::fence();
  }

  /* false: not a component for this program; nothing outside LDS has been
     written.  walks: the walks it made (statistics) */
  GTS_HD bool run_fast(int mode)
  {
    const uint32_t lane = W::lane();
    if (nv < 2) return false;
    const bool timed = C.tspan != nullptr;     /* detailed profile only: no clock reads otherwise */
    uint64_t t0 = 0, t1 = 0;
    if (timed) t0 = W::clock();
    /* cycle removal with its marks kept in LDS (1 % of the components have a
       cycle to remove; all but a few are clean afterwards) */
    removecycles_t<true>(mode == GTS_MODE_MAKESCAFFOLD);
    reuse_cc = clean;
    if (timed) t1 = W::clock();
    bool marks = false;
    if (mode == GTS_MODE_MAKESCAFFOLD) {
      if (!reuse_cc) calc_cc();
      /* a component that would hand its walks to tasks (try_defer) */
      if (C.defer_min_nv && nv >= C.defer_min_nv && nterm >= 2 &&
          (uint64_t)nterm * nv >= C.defer_min_work) return false;
      if (!clean && C.defer_unclean_work && (C.defer_min_nv || C.defer_ref_min_nv) && nterm >= 4 &&
          (uint64_t)nterm * nv >= C.defer_unclean_work && W::peek(C.ndeferred) != 0) return false;
      for (uint32_t s = lane; s < nv; s += W::WIDTH) { M.st_dir[s] = 0; M.tight[s] = 0; }   /* as makescaffold() */
      W::fence();
      auto ccoff = M.ccoff;
      for (uint32_t i = 0; i < ncc; ++i) {
        const uint32_t tb = W::uni(ccoff[i]), te = W::uni(ccoff[i + 1]);
        if (te - tb == 1) lonesome(W::uni(M.term[tb]));
        if (te - tb > 1) {
          uint64_t cc_len = 0;
          uint32_t cc_n = 0, cc_start = 0;
          bool batched = false;
          if constexpr (LDS) {
            batched = clean ? cc_walks_batched<GTS_WALK_LANES>(tb, te, cc_len, cc_n, &cc_start)
                            : (C.batch_walks >= 2 && cc_walks_batched_unclean(tb, te, cc_len, cc_n, &cc_start));
          }
          if (!batched) {
            /* a tie in the batch, or a component that is not clean: the walks of
               this cc one by one (create_walk_clean / create_walk_fast: the
               reference's tie-breaks in closed form); a walk that needs the
               reference's search sends the component to the full program */
            cc_len = 0; cc_n = 0;
            for (uint32_t j = tb; j < te; ++j) {
              if (cc_len == all_bases()) break;
              const uint64_t len0 = cc_len;
              const uint32_t start = W::uni(M.term[j]);
              no_reference = true;
              create_walk(start, cc_len, cc_n);
              if (needs_reference) return false;
              if (cc_len != len0) cc_start = start;
            }
          }
          if (cc_n) { mark_best_lds(cc_n, cc_start); marks = true; }
        }
      }
    }
    /* (removecycles alone ends with every unmarked vertex UNVISITED: removecycles_t) */
    if (C.timing_skip_writeback) return true;
    any_scaffold_marks = marks;
    flush_local_marks();
#pragma unroll 1
    for (uint32_t s = lane; s < nv; s += W::WIDTH) C.G.vstate[C.slot_v[s0 + s]] = M.vst[s];
    if (timed && lane == 0) {
      const uint64_t t2 = W::clock();
      C.stat_fast[c] = nfast; C.stat_ncc[c] = ncc; C.stat_clean[c] = (clean ? 1u : 0u) | (nterm << 8);
      C.tstat[5 * (uint64_t)c] = t1 - t0;
      C.tstat[5 * (uint64_t)c + 2] = t2 - t1;
      C.tspan[2 * (uint64_t)c] = t0; C.tspan[2 * (uint64_t)c + 1] = t2;
    }
    return true;
  }

  GTS_HD void run(int mode)
  {
    const uint32_t lane = W::lane();
    const uint64_t t0 = tick();
    if (local_marks) removecycles_t<true>(mode == GTS_MODE_MAKESCAFFOLD);
    else removecycles(mode == GTS_MODE_MAKESCAFFOLD);
    reuse_cc = clean;   /* the terminal search of the last pass is makescaffold's */
    const uint64_t t1 = tick();
    if (C.small_stat && nv <= 64 && lane == 0) {
      W::add64((uint64_t *)C.small_stat + (was_all_live && clean ? 0 : 1), 1);
      W::add64((uint64_t *)C.small_stat + (was_all_live && clean ? 2 : 3), t1 - t0);
    }
    const bool was_clean = clean;
    bool deferred = false;
    if (mode == GTS_MODE_MAKESCAFFOLD) {
      makescaffold();
      deferred = deferred_late;
    }
    const uint64_t t2 = tick();
    if (local_marks) flush_local_marks();
    /* (not unrolled: an eight-fold copy hipcc 7.2 made of this loop inside
       k_components_pool -- lane-dependent trip count, address registers spilled
       around it -- left the states of the slots past the first 64 of a large
       component unwritten in one build of this kernel; it showed in
       tests/test_gpu_parity.py::test_small_graphs_stage_by_stage) */
#pragma unroll 1
    for (uint32_t s = lane; s < nv; s += W::WIDTH) {
      const uint8_t st = M.vst[s];
      /* removecycles leaves every unmarked vertex UNVISITED (algorithms.c:
         513-517, 550-553); makescaffold leaves VISITED or SCAFFOLD */
      C.G.vstate[C.slot_v[s0 + s]] = st;
    }
    run_clean = was_clean; run_deferred = deferred;
    if (lean_stats) { if (lane == 0 && err) C.cerr[c] = err; }
    else if (lane == 0) {
      C.cerr[c] = err; C.stat_fast[c] = nfast; C.stat_slow[c] = nslow; C.stat_ncc[c] = ncc;
      C.stat_clean[c] = (was_clean ? 1u : 0u) | (deferred ? 2u : 0u) | (nodefer << 2) | (nterm << 8);
      C.tstat[5 * (uint64_t)c] = t1 - t0;
      C.tstat[5 * (uint64_t)c + 1] = t2 - t1 - tfast - tslow;
      C.tstat[5 * (uint64_t)c + 2] = tfast;
      C.tstat[5 * (uint64_t)c + 3] = tslow;
      C.tstat[5 * (uint64_t)c + 4] = npops;
      if (C.tspan) { C.tspan[2 * (uint64_t)c] = t0; C.tspan[2 * (uint64_t)c + 1] = tick(); }
    }
    W::fence();
  }
};

/* host / test wave policy: one lane */
struct GtsWave1 {
  static const uint32_t WIDTH = 1;
  static const bool TEAM = false;
  static GTS_HD void and_bits(uint32_t *p, uint32_t m) { *p &= m; }
  static GTS_HD uint64_t group8_or(uint64_t x) { return x; }
  static GTS_HD uint32_t group8_or32(uint32_t x) { return x; }
  static GTS_HD uint32_t group8_add32(uint32_t x) { return x; }
  static GTS_HD uint64_t peek(const unsigned long long *p) { return *p; }
  static GTS_HD uint32_t clz64(uint64_t v) { uint32_t n = 0; while (n < 64 && !(v & (0x8000000000000000ull >> n))) ++n; return n; }
  static GTS_HD uint32_t lane() { return 0; }
  static GTS_HD uint64_t ballot(bool p) { return p ? 1u : 0u; }
  static GTS_HD uint32_t popc(uint64_t m) { return (uint32_t)(m & 1u); }
  static GTS_HD uint32_t popc_below(uint64_t, uint32_t) { return 0; }
  static GTS_HD uint32_t ctz(uint64_t) { return 0; }
  static GTS_HD uint32_t msb(uint64_t) { return 0; }
  static GTS_HD uint32_t shfl(uint32_t v, uint32_t) { return v; }
  static GTS_HD uint64_t shfl64(uint64_t v, uint32_t) { return v; }
  static GTS_HD uint32_t uni(uint32_t v) { return v; }
  static GTS_HD int64_t uni64(int64_t v) { return v; }
  static GTS_HD void fence() {}
  static GTS_HD uint64_t clock() { return 0; }
  static GTS_HD void count(unsigned long long *p) { ++*p; }
  static GTS_HD void count_n(uint32_t *p, uint32_t n) { *p += n; }
  static GTS_HD void or_bits(uint32_t *p, uint32_t m) { *p |= m; }
  static GTS_HD void add64(uint64_t *p, uint64_t n) { *p += n; }
  static GTS_HD uint64_t alloc(unsigned long long *used, uint64_t n)
  { const uint64_t o = *used; *used += n; return o; }
  static GTS_HD uint32_t clz32(uint32_t v) { uint32_t n = 0; while (n < 32 && !(v & (0x80000000u >> n))) ++n; return n; }
  static GTS_HD uint32_t scan_incl_small(uint32_t v) { return v; }
  static GTS_HD uint32_t bcast(uint32_t v, uint32_t) { return v; }
  static GTS_HD float shflf(float v, uint32_t) { return v; }
  static GTS_HD uint64_t lanemask_lt(uint32_t) { return 0; }
  static GTS_HD uint64_t range_mask(uint32_t lo, uint32_t hi) { return lo < hi ? 1u : 0u; }
  template <class P> static GTS_HD uint32_t atomic_max(P p, uint32_t v)
  { const uint32_t o = *p; if (v > o) *p = v; return o; }
};

#endif
