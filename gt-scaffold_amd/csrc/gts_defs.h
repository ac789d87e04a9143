/*
  gts_defs.h -- shared definitions of the MI355X scaffold-graph engine.

  The per-vertex / per-component algorithm bodies (gts_filter.hpp,
  gts_component.hpp) are written against these plain-pointer views so that the
  same source is compiled into gfx950 kernels by hipcc (the product) and into a
  serial host harness by g++ (tests/hostsim, test infrastructure that checks
  the parallel reformulation against the oracle without a GPU).
*/
#ifndef GTS_DEFS_H
#define GTS_DEFS_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GTS_HD __host__ __device__ __forceinline__
#else
#define GTS_HD inline
#endif

/* GraphItemState, ref src/gt_scaffolder_graph.h:29-31 */
enum : uint8_t {
  GIS_UNVISITED = 0, GIS_POLYMORPHIC = 1, GIS_INCONSISTENT = 2, GIS_REPEAT = 3,
  GIS_VISITED = 4, GIS_PROCESSED = 5, GIS_SCAFFOLD = 6, GIS_CYCLIC = 7
};

#define GTS_NONE 0xFFFFFFFFu
#define GTS_F_SENSE 1u   /* edge flag bit0: sense, bit1: same (graph.h:63-69) */
#define GTS_F_SAME 2u
/* compact edges only: the twin would be followed right after this edge (its
   sense equals the direction in which this edge leaves its end vertex), i.e.
   the pair is geometrically inconsistent and the reference's twin exclusion
   (algorithms.c:702) becomes path dependent */
#define GTS_F_UTURN 4u
/* compact edges only: the twin was live when the component was compacted */
#define GTS_F_TWINLIVE 8u

/* ref gt_scaffolder_algorithms.c:38-47 */
GTS_HD bool gts_vertex_is_marked(uint8_t s)
{
  /* POLYMORPHIC, REPEAT, CYCLIC as a bit set: one shift instead of a chain of
     compares and branches in the per-edge loops */
  return ((0x8Au >> (s & 7u)) & 1u) != 0;
}
/* ref gt_scaffolder_algorithms.c:50-58 */
GTS_HD bool gts_edge_is_marked(uint8_t s)
{
  return ((0x8Eu >> (s & 7u)) & 1u) != 0;   /* POLYMORPHIC, INCONSISTENT, REPEAT, CYCLIC */
}
/* direction in which a walk leaves the end vertex of an edge,
   ref gt_scaffolder_algorithms.c:475-478, 688-691, 968-971 */
GTS_HD bool gts_next_dir(uint8_t flags)
{
  bool sense = flags & GTS_F_SENSE, same = flags & GTS_F_SAME;
  return same ? sense : !sense;
}
/* direction marked on the end vertex of an inconsistent edge,
   ref gt_scaffolder_algorithms.c:331,336: sense ? !same : same */
GTS_HD bool gts_twin_dir(uint8_t flags)
{
  bool sense = flags & GTS_F_SENSE, same = flags & GTS_F_SAME;
  return sense ? !same : same;
}

/* Scaffold graph resident in HBM, CSR by start vertex.  Edge attributes are
   stored ONCE, in adjacency order ("position" p), so that a vertex's list is a
   contiguous, coalesced segment; eid[p] is the reference's edge id (creation
   order, ref gt_scaffolder_graph.c:137-170) and twin[p] the position of the
   edge created with it in the opposite direction (ref parser.c:374-377). */
struct GtsGraphView {
  uint32_t n;              /* vertices (contigs) */
  uint32_t m;              /* directed edges */
  const uint32_t *row;     /* n+1 */
  const int64_t *seq_len;  /* n */
  const float *astat;      /* n */
  const float *copy_num;   /* n */
  uint8_t *vstate;         /* n */
  const uint32_t *end;     /* m */
  const int64_t *dist;     /* m */
  const float *sd;         /* m */
  const uint8_t *flags;    /* m */
  uint8_t *state;          /* m */
  const uint32_t *twin;    /* m */
  const uint32_t *eid;     /* m */
};

/* Thresholds that replace erf() in the ambiguous-order test: the reference's
   float pipeline p_wrong(interval) > cutoff (ref algorithms.c:187-192) is a
   monotone step function of |interval| on each sign, so the host finds by
   bisection (with the host libm the reference itself would call) the largest
   float for which it holds; the device compares the correctly-rounded interval
   against it and never evaluates erf. t < 0 means "never ambiguous". */
struct GtsAmbThresholds {
  float tpos;  /* interval >= 0:  ambiguous <=> interval <= tpos */
  float tneg;  /* interval <  0:  ambiguous <=> -interval <= tneg */
};

struct GtsFilterParams {
  GtsAmbThresholds amb;
  float cncutoff;
  int64_t ocutoff;
};

#endif
