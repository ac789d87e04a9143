/*
  gts_oracle.h -- CPU oracle for the gt-scaffold hot path.

  TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
  scaffold-graph algorithms, used as the checker for the HIP engine.  Only
  tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
  The product (gt-scaffold_amd/) never links, imports or calls anything here.

  Parity pinning: the oracle is checked against every golden vector the
  reference's own test-suite holds for this path (the .dot files under tests/golden/, taken
  from reference testdata/, see tests/test_oracle_golden.py).  The reference
  itself cannot be built in this image: it needs GenomeTools (libgenometools,
  its core/ and extended/ headers), which is not vendored in /root/reference and is not
  installed; no stand-in was written.

  Every function cites the reference file:line it restates.
*/
#ifndef GTS_ORACLE_H
#define GTS_ORACLE_H

#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/gt_scaffolder_graph.h:29-31 (GraphItemState, same order) */
enum {
  ORA_UNVISITED = 0, ORA_POLYMORPHIC = 1, ORA_INCONSISTENT = 2, ORA_REPEAT = 3,
  ORA_VISITED = 4, ORA_PROCESSED = 5, ORA_SCAFFOLD = 6, ORA_CYCLIC = 7
};

/* reference src/gt_scaffolder_graph.h:34-47 (vertex), indices instead of
   pointers */
typedef struct {
  char *header;
  uint64_t seq_len;
  float astat;
  float copy_num;
  uint64_t nof_edges;
  uint64_t cap_edges;
  uint64_t *edges;  /* edge ids in insertion order */
  uint8_t state;
} OraVertex;

/* reference src/gt_scaffolder_graph.h:50-70 (edge) */
typedef struct {
  uint64_t start, end;
  int64_t dist;
  float std_dev;
  uint64_t num_pairs;
  uint8_t state;
  bool sense, same;
} OraEdge;

typedef struct {
  OraVertex *v;
  uint64_t nv, cap_v;
  OraEdge *e;
  uint64_t ne, cap_e;
} OraGraph;

/* scaffold records: reference src/gt_scaffolder_graph.h:97-100 */
typedef struct {
  uint64_t root;
  uint64_t nof_edges;
  uint64_t cap;
  uint64_t *edges;
  uint64_t seqlen; /* value handed to the assembly-stats calculator */
} OraRecord;

typedef struct {
  OraRecord *r;
  uint64_t n, cap;
} OraRecords;

/* ---- graph construction ---- */
OraGraph *ora_graph_new(uint64_t max_v, uint64_t max_e);
void ora_graph_delete(OraGraph *g);
/* gt_scaffolder_graph.c:105 */
void ora_graph_add_vertex(OraGraph *g, const char *header, uint64_t seq_len,
                          float astat, float copy_num);
/* gt_scaffolder_graph.c:137 */
void ora_graph_add_edge(OraGraph *g, uint64_t vstart, uint64_t vend,
                        int64_t dist, float std_dev, uint64_t num_pairs,
                        bool dir, bool same);
/* gt_scaffolder_parser.c:340-378 for ONE distance record whose two contigs
   are already resolved to vertex ids */
void ora_graph_add_record(OraGraph *g, uint64_t root, uint64_t ctg,
                          int64_t dist, float std_dev, uint64_t num_pairs,
                          bool sense, bool same);
/* the same with read_distances' ismatepair argument (parser.c:297, :362:
   true = an existing edge is never altered) */
void ora_graph_add_record_mp(OraGraph *g, uint64_t root, uint64_t ctg,
                             int64_t dist, float std_dev, uint64_t num_pairs,
                             bool sense, bool same, bool ismatepair);
void ora_graph_add_records_mp(OraGraph *g, uint64_t n, const uint32_t *root,
                              const uint32_t *ctg, const int64_t *dist,
                              const float *std_dev, const uint64_t *num_pairs,
                              const uint8_t *flags, bool ismatepair);
/* bulk form of the above for synthetic inputs; flags bit0 = sense, bit1 = same */
void ora_graph_add_records(OraGraph *g, uint64_t n, const uint32_t *root,
                           const uint32_t *ctg, const int64_t *dist,
                           const float *std_dev, const uint64_t *num_pairs,
                           const uint8_t *flags);
/* gt_scaffolder_graph.c:346 */
int ora_graph_new_from_file(OraGraph **out, const char *ctg_filename,
                            uint64_t min_ctg_len, const char *dist_filename,
                            bool astat_is_annotated, char *err, size_t errlen);
int ora_graph_new_from_file_mp(OraGraph **out, const char *ctg_filename,
                               uint64_t min_ctg_len, const char *dist_filename,
                               bool astat_is_annotated, bool ismatepair,
                               char *err, size_t errlen);
/* gt_scaffolder_graph.c:421 */
int ora_graph_test(uint64_t max_v, uint64_t max_e, bool init_v, uint64_t nv,
                   bool init_e, uint64_t ne, const char *dot_out);
/* gt_scaffolder_parser.c:55 */
int ora_parser_read_distances_test(const char *filename, const char *out,
                                   char *err, size_t errlen);

/* ---- algorithms ---- */
/* gt_scaffolder_algorithms.c:90 */
int ora_mark_repeats(const char *filename, OraGraph *g, float copy_num_cutoff,
                     float astat_cutoff, char *err, size_t errlen);
/* same, astat/copy_num already loaded into the vertices; have_file mirrors
   strlen(filename) != 0 */
void ora_mark_repeats_loaded(OraGraph *g, bool have_file, float copy_num_cutoff,
                             float astat_cutoff);
/* gt_scaffolder_algorithms.c:261 */
void ora_filter(OraGraph *g, float pcutoff, float cncutoff, int64_t ocutoff);
/* gt_scaffolder_algorithms.c:495 */
void ora_removecycles(OraGraph *g);
/* gt_scaffolder_algorithms.c:767.  lazy_maps != 0 replaces the per-walk
   O(|V|) distance-map initialisation (gt_scaffolder_algorithms.c:648-650) by an
   epoch-stamped map; results are identical (tests/test_oracle_modes.py). */
void ora_makescaffold(OraGraph *g, int lazy_maps);
/* gt_scaffolder_algorithms.c:901 */
OraRecords *ora_iterate_scaffolds(OraGraph *g);
void ora_records_delete(OraRecords *r);
/* gt_scaffolder_algorithms.c:1000 */
int ora_write_scaffold(const OraGraph *g, const OraRecords *r, const char *fn);
/* gt_scaffolder_graph.c:247/269 */
int ora_graph_print(const OraGraph *g, const char *filename);
/* gt_scaffolder_algorithms.c:175 (exposed for threshold tests) */
bool ora_ambiguousorder(int64_t dist1, float sd1, int64_t dist2, float sd2,
                        float cutoff);
/* the float pipeline of gt_scaffolder_algorithms.c:187-192 on a given interval */
bool ora_ambiguous_from_interval(float interval, float cutoff);

/* ---- flat accessors for ctypes ---- */
uint64_t ora_nv(const OraGraph *g);
uint64_t ora_ne(const OraGraph *g);
void ora_get_vertex_states(const OraGraph *g, uint8_t *out);
void ora_get_edge_states(const OraGraph *g, uint8_t *out);
void ora_get_edges(const OraGraph *g, uint32_t *start, uint32_t *end,
                   int64_t *dist, float *std_dev, uint64_t *num_pairs,
                   uint8_t *flags);
void ora_get_vertices(const OraGraph *g, uint64_t *seq_len, float *astat,
                      float *copy_num);
const char *ora_vertex_header(const OraGraph *g, uint64_t i);
void ora_set_vertex_attrs(OraGraph *g, const float *astat,
                          const float *copy_num);
uint64_t ora_records_n(const OraRecords *r);
uint64_t ora_records_total_edges(const OraRecords *r);
/* roots[n], offsets[n+1], edges[total], seqlen[n] */
void ora_records_flatten(const OraRecords *r, uint64_t *roots,
                         uint64_t *offsets, uint64_t *edges, uint64_t *seqlen);

#ifdef __cplusplus
}
#endif
#endif
