/*
  gt_scaffolder_host.h -- C host layer of libgtscaffold_hip.so.

  Same function names, argument meaning and error behaviour as the reference's
  public API (src/gt_scaffolder_graph.h, src/gt_scaffolder_parser.h,
  src/gt_scaffolder_algorithms.h), with GenomeTools' types replaced by plain C:
      GtStr *      -> const char *
      GtError *    -> char *err, size_t errlen   (message, may be NULL)
      GtUword/GtWord -> uint64_t / int64_t
      GtArray * of GtScaffolderGraphRecord * -> GtScaffolderGraphRecords *
  Text parsing (.fa / .de / .astat), .dot / .scaf writing and the scaffold
  record walk run on the host; graph construction and all algorithms run on
  the GPU through gt_scaffold_hip.h.  Without a GPU every call that needs the
  engine fails with an error message.
*/
#ifndef GT_SCAFFOLDER_HOST_H
#define GT_SCAFFOLDER_HOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GtScaffolderGraph GtScaffolderGraph;
typedef struct GtScaffolderGraphRecords GtScaffolderGraphRecords;

/* ref gt_scaffolder_graph.c:60 / :75 */
GtScaffolderGraph *gt_scaffolder_graph_new(uint64_t max_nof_vertices,
                                           uint64_t max_nof_edges);
void gt_scaffolder_graph_delete(GtScaffolderGraph *graph);
/* ref gt_scaffolder_graph.c:105, :137 (hand-built graphs live on the host and
   can be printed; they return -1 when they exceed the announced capacity,
   where the reference asserts) */
int gt_scaffolder_graph_add_vertex(GtScaffolderGraph *graph, const char *header_seq,
                                   uint64_t seq_len, float astat, float copy_num);
int gt_scaffolder_graph_add_edge(GtScaffolderGraph *graph, uint64_t vstart,
                                 uint64_t vend, int64_t dist, float std_dev,
                                 uint64_t num_pairs, bool dir, bool same);
/* ref gt_scaffolder_graph.h:127-146, gt_scaffolder_graph.c:174-244.  Vertices
   and edges are named by their ids (the reference hands out pointers into its
   two arrays; an id is that pointer minus the array base):
     find_edge      the first edge of vertex_1's list that ends in vertex_2,
                    GT_SCAFFOLDER_NO_EDGE where the reference returns NULL
     get_vertex_id  the id of a vertex (the identity here; GT_SCAFFOLDER_NO_VERTEX
                    out of range)
     get_vertex     the vertex with this header (binary search: the vertices
                    are in header order after count_distances / read_distances)
     alter_edge     new distance, deviation, pair count, sense and same for one
                    edge, before or after the graph has moved to the GPU; -1 for
                    an edge that does not exist (the reference asserts) */
#define GT_SCAFFOLDER_NO_EDGE UINT64_MAX
#define GT_SCAFFOLDER_NO_VERTEX UINT64_MAX
uint64_t gt_scaffolder_graph_find_edge(GtScaffolderGraph *graph, uint64_t vertex_1,
                                       uint64_t vertex_2);
uint64_t gt_scaffolder_graph_get_vertex_id(const GtScaffolderGraph *graph, uint64_t vertex);
bool gt_scaffolder_graph_get_vertex(const GtScaffolderGraph *graph, uint64_t *vertex,
                                    const char *header_seq);
int gt_scaffolder_graph_alter_edge(GtScaffolderGraph *graph, uint64_t edge, int64_t dist,
                                   float std_dev, uint64_t num_pairs, bool sense, bool same);
/* ref gt_scaffolder_graph.c:346 */
int gt_scaffolder_graph_new_from_file(GtScaffolderGraph **graph_par,
                                      const char *ctg_filename,
                                      uint64_t min_ctg_len,
                                      const char *dist_filename,
                                      bool astat_is_annotated, char *err,
                                      size_t errlen);
/* The four steps gt_scaffolder_graph_new_from_file is made of, ref
   src/gt_scaffolder_parser.h (same names and argument order):
     count_contigs   parser.c:495  contigs of at least min_ctg_len in a FASTA file
     read_contigs    parser.c:524  contigs longer than min_ctg_len become the
                                   vertices of `graph` (from gt_scaffolder_graph_new)
     count_distances parser.c:150  integrity check of the .de file, sorts the
                                   vertices by header; *nof_distances = upper
                                   bound of the edges (2 per valid record)
     read_distances  parser.c:295  the records become edges: this is where the
                                   graph moves to the GPU (one call per graph).
                                   ismatepair = true: a record of a contig pair
                                   that already has its edges never alters
                                   them (parser.c:362) */
int gt_scaffolder_parser_count_contigs(const char *filename, uint64_t min_ctg_len,
                                       uint64_t *nof_contigs, char *err, size_t errlen);
int gt_scaffolder_parser_read_contigs(GtScaffolderGraph *graph, const char *filename,
                                      uint64_t min_ctg_len, bool astat_is_annotated,
                                      char *err, size_t errlen);
int gt_scaffolder_parser_count_distances(const GtScaffolderGraph *graph,
                                         const char *file_name,
                                         uint64_t *nof_distances, char *err,
                                         size_t errlen);
int gt_scaffolder_parser_read_distances(const char *filename,
                                        GtScaffolderGraph *graph, bool ismatepair,
                                        char *err, size_t errlen);
/* ref gt_scaffolder_graph.c:247 */
int gt_scaffolder_graph_print(const GtScaffolderGraph *g, const char *filename,
                              char *err, size_t errlen);
/* ref gt_scaffolder_graph.h:151-153, gt_scaffolder_graph.c:269-307: the dot
   representation into an open stream (the reference's GtFile is a stdio stream
   here; it returns nothing there -- a failed write ends the program --, here
   0 / -1, the message is gt_scaffolder_graph_last_error's) */
int gt_scaffolder_graph_print_generic(const GtScaffolderGraph *g, FILE *f);
/* ref gt_scaffolder_graph.c:421 */
int gt_scaffolder_graph_test(uint64_t max_nof_vertices, uint64_t max_nof_edges,
                             bool init_vertices, uint64_t nof_vertices,
                             bool init_edges, uint64_t nof_edges,
                             bool print_graph, char *err, size_t errlen);
/* ref gt_scaffolder_parser.c:55 */
int gt_scaffolder_parser_read_distances_test(const char *filename,
                                             const char *output_filename,
                                             char *err, size_t errlen);

/* ref gt_scaffolder_algorithms.c:90 */
int gt_scaffolder_graph_mark_repeats(const char *filename,
                                     GtScaffolderGraph *graph,
                                     float copy_num_cutoff, float astat_cutoff,
                                     char *err, size_t errlen);
/* ref gt_scaffolder_algorithms.c:261, :495, :767 (void in the reference; here
   they return the engine's status, 0 = ok) */
int gt_scaffolder_graph_filter(GtScaffolderGraph *graph, float pcutoff,
                               float cncutoff, int64_t ocutoff);
int gt_scaffolder_removecycles(GtScaffolderGraph *graph);
int gt_scaffolder_makescaffold(GtScaffolderGraph *graph);

/* ref gt_scaffolder_algorithms.c:901; scaf_seqlen (may be NULL) receives a
   malloc'ed array of the lengths the reference hands to its assembly-stats
   calculator, one per record */
GtScaffolderGraphRecords *
gt_scaffolder_graph_iterate_scaffolds(GtScaffolderGraph *graph,
                                      uint64_t **scaf_seqlen);
uint64_t gt_scaffolder_graph_records_size(const GtScaffolderGraphRecords *r);
void gt_scaffolder_graph_records_delete(GtScaffolderGraphRecords *r);
/* ref gt_scaffolder_algorithms.c:1000 */
int gt_scaffolder_graph_write_scaffold(const GtScaffolderGraphRecords *records,
                                       const char *file_name, char *err,
                                       size_t errlen);

/* accessors used by bindings and tests */
uint64_t gt_scaffolder_graph_nof_vertices(const GtScaffolderGraph *g);
uint64_t gt_scaffolder_graph_nof_edges(const GtScaffolderGraph *g);
const char *gt_scaffolder_graph_last_error(const GtScaffolderGraph *g);
/* the edges in id order, arrays of gt_scaffolder_graph_nof_edges elements (any
   may be NULL); flags: bit 0 sense, bit 1 same */
int gt_scaffolder_graph_get_edges(GtScaffolderGraph *g, uint32_t *start, uint32_t *end,
                                  int64_t *dist, float *std_dev, int64_t *num_pairs,
                                  uint8_t *flags);
/* which GPU new_from_file / the algorithms use (default 0) */
void gt_scaffolder_set_device(int device);
/* who reads distance and A-statistic files: 0 (default) the GPU parser
   (gts_deparse.hip), the host restatement of parser.c / algorithms.c:108-153
   for a file outside its regular form; 1 the host code only; 2 the GPU parser
   or an error */
void gt_scaffolder_set_distance_parser(int mode);
/* who formats the edge lines of gt_scaffolder_graph_print for a graph on the
   GPU: 0 (default) the GPU (gtsg_format_dot_edges), 1 the host */
void gt_scaffolder_set_dot_writer(int mode);
/* who walks the scaffold records (gt_scaffolder_graph_iterate_scaffolds) of a
   graph on the GPU: 0 (default) the GPU ranks the records of the SCAFFOLD edges
   that form vertex-disjoint simple paths (gtsg_scaffold_records) and the host
   walks the few paths that do not, in the reference's order of visits; 1 the
   host walks everything.  _last_record_walk: which of the two the last call was */
void gt_scaffolder_set_record_walk(int mode);
int gt_scaffolder_last_record_walk(void);

#ifdef __cplusplus
}
#endif
#endif
