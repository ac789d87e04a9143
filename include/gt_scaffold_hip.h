/*
  gt_scaffold_hip.h -- C ABI of the MI355X scaffold-graph engine
  (libgtscaffold_hip.so).  Plain pointers and sizes only.

  It replaces, function by function, the compute path of the reference's
  GtScaffolderGraph C API.  A GenomeTools build keeps gt_scaffolder_graph.h /
  gt_scaffolder_algorithms.h unchanged and routes their bodies here
  (INTEGRATION.md shows the shim); the host layer in gt_scaffolder_host.h does
  the same with plain C types in place of GtStr / GtError / GtFile.

  Conventions
    * every call returns 0 on success, a negative GTSG_E* code on failure and
      leaves a message retrievable with gtsg_last_error();
    * vertex ids are the reference's: contigs sorted by header
      (ref src/gt_scaffolder_parser.c:172), edge ids are creation order
      (ref src/gt_scaffolder_graph.c:137-170);
    * states are GraphItemState values (ref src/gt_scaffolder_graph.h:29-31);
    * record/edge flags: bit0 = sense, bit1 = same (ref graph.h:63-69);
    * `on_device` != 0: the array arguments are device pointers valid on the
      engine's GPU (e.g. torch tensors' data_ptr()), consumed on the engine's
      stream; 0: host pointers, copied with hipMemcpyAsync;
    * stream contract: every call enqueues on the engine's stream and returns
      after that stream has drained, so results are visible to any stream
      afterwards.  Device INPUTS must be complete before the call: either they
      were produced on the engine's stream, or on the legacy null stream while
      the engine runs on a stream it created itself (a blocking stream, ordered
      with the null stream by HIP), or the producer stream was synchronised by
      the caller.  To run on the null stream itself pass hipStreamLegacy
      ((void *)1) to gtsg_create; NULL always means "create a stream".
*/
#ifndef GT_SCAFFOLD_HIP_H
#define GT_SCAFFOLD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GtsgEngine GtsgEngine;

enum {
  GTSG_OK = 0,
  GTSG_EINVAL = -1,    /* bad argument / call order */
  GTSG_EHIP = -2,      /* HIP runtime error */
  GTSG_ENOMEM = -3,
  GTSG_EWALK = -4,     /* a walk exceeded its pop bound (cyclic distance maps) */
  GTSG_ELIMIT = -5,    /* 2^31 contigs, 2^29 records or 2^30 edges, or more */
  GTSG_EINTERNAL = -6  /* a device-side wait ran into its bound (a bug: please report) */
};

/* Engine bound to HIP device `device`.  `stream` is a hipStream_t (NULL: the
   engine creates its own).  Fails loudly if no gfx950-capable GPU is present:
   there is no CPU fallback. */
int gtsg_create(GtsgEngine **out, int device, void *stream);
void gtsg_destroy(GtsgEngine *e);
const char *gtsg_last_error(const GtsgEngine *e);

/* Contigs = vertices.  replaces gt_scaffolder_graph_add_vertex in the loop of
   gt_scaffolder_parser_read_contigs (ref parser.c:524-549, graph.c:105-131).
   astat / copy_num may be NULL (0.0, as for FASTA headers without
   annotation, ref parser.c:434-435). */
int gtsg_set_contigs(GtsgEngine *e, uint64_t n, const int64_t *seq_len,
                     const float *astat, const float *copy_num, int on_device);

/* For a SHARD of a larger graph (multi-GPU): the engine's vertex
   numbers are then local -- any numbering that keeps the order of the whole
   graph's vertex ids -- and times[v] is the id of local vertex v in the whole
   graph (strictly increasing).  The filter is the only stage that compares
   "times" across shards (the latest-hit table of gtsg_filter_get/set_lasthit
   then holds whole-graph ids).  NULL goes back to the identity; a call to
   gtsg_set_contigs does the same. */
int gtsg_set_vertex_times(GtsgEngine *e, const uint32_t *times, int on_device);

/* DistEst records in file order -> edges + CSR.  replaces
   gt_scaffolder_parser_read_distances' per-record graph updates
   (ref parser.c:357-378: find_edge / alter_edge / two add_edge calls,
   graph.c:137-235).  num_pairs may be NULL. */
int gtsg_build_from_records(GtsgEngine *e, uint64_t n_records,
                            const uint32_t *root, const uint32_t *ctg,
                            const int64_t *dist, const float *std_dev,
                            const int64_t *num_pairs, const uint8_t *flags,
                            int on_device);
/* the same with the reference's `ismatepair` argument of
   gt_scaffolder_parser_read_distances (ref parser.c:297, :362): non-zero = a
   record of a pair that already has its edges never alters them (the creating
   record's estimate stays in both directions); 0 = gtsg_build_from_records.
   Contig ids >= the number of contigs are rejected with GTSG_EINVAL before
   anything is indexed with them. */
int gtsg_build_from_records_ex(GtsgEngine *e, uint64_t n_records,
                               const uint32_t *root, const uint32_t *ctg,
                               const int64_t *dist, const float *std_dev,
                               const int64_t *num_pairs, const uint8_t *flags,
                               int on_device, int ismatepair);

/* A-statistics / copy numbers read from the .astat file, per vertex
   (ref algorithms.c:118-149, the file part of mark_repeats). */
int gtsg_set_astat(GtsgEngine *e, const float *astat, const float *copy_num,
                   int on_device);

/* ref gt_scaffolder_algorithms.h: gt_scaffolder_graph_mark_repeats (marking
   loop algorithms.c:155-167; have_file = strlen(filename) != 0) */
int gtsg_mark_repeats(GtsgEngine *e, int have_file, float copy_num_cutoff,
                      float astat_cutoff);
/* ref gt_scaffolder_graph_filter, algorithms.c:261-343 */
int gtsg_filter(GtsgEngine *e, float pcutoff, float cncutoff, int64_t ocutoff);
/* The filter in two halves, for graphs sharded over several GPUs by connected
   component (multi-GPU): marked (repeat) contigs are shared between
   shards and the only cross-shard effect of the filter is the time of the
   latest inconsistency hit on their edges (algorithms.c:249-258).  Between
   _begin and _end the shards combine the table (2 x u32 per contig, as int32:
   -1 = none) with an element-wise MAX; no other engine call may intervene. */
int gtsg_filter_begin(GtsgEngine *e, float pcutoff, float cncutoff, int64_t ocutoff);
int gtsg_filter_get_lasthit(GtsgEngine *e, uint32_t *dst, int on_device);
int gtsg_filter_set_lasthit(GtsgEngine *e, const uint32_t *src, int on_device);
int gtsg_filter_end(GtsgEngine *e);

/* Component-partition step: joins the contigs of every record (root, ctg) of a
   slice whose contigs are both not skipped; labels[] (n entries, labels[v] <= v,
   identity at first) are parent pointers on entry and the smallest contig of
   each tree on return.  Shards iterate {label, all-reduce MIN} to a fixpoint. */
int gtsg_label_components(GtsgEngine *e, uint64_t n, uint64_t n_records,
                          const uint32_t *root, const uint32_t *ctg,
                          const uint8_t *skip, uint32_t *labels, int on_device);
/* Step 2 of that partition, on the device (all pointers are device pointers).
   gtsg_plan_weights: weights[c] = records of THIS shard that belong to the
   component with label c (a record counts for its first contig that is not
   skipped; labels from gtsg_label_components after the shards agreed).  The
   caller sums the weights over the shards (all_reduce), then
   gtsg_plan_deal: the components go to the ranks heaviest first (ties: smaller
   label first) in serpentine order 0..world-1, world-1..0, ...; owner[v] = the
   rank of v's component, -1 for skipped (repeat) contigs, which every shard
   holds; load[r] = records of rank r's components. */
int gtsg_plan_weights(GtsgEngine *e, uint64_t n, uint64_t nrec, const uint32_t *root,
                      const uint32_t *ctg, const uint8_t *skip, const uint32_t *labels,
                      int32_t *weights);
int gtsg_plan_deal(GtsgEngine *e, uint64_t n, const uint8_t *skip, const uint32_t *labels,
                   const int32_t *weights, uint32_t world, int8_t *owner, int64_t *load);

/* Component-partition step, routing: the records of this shard, packed into
   four 64-bit words each and grouped by the rank that owns their component
   (file order kept inside a group), ready for one all_to_all; and back.  All
   array arguments are DEVICE pointers; counts[world] is a host array.
     owner[c]  rank of contig c, negative for a repeat contig (shared: the record
               follows its other contig)
     first_index  global index (= position in the .de file) of this shard's
               first record
   unpack: loc_of (may be NULL) maps whole-graph contig ids to the shard's
   local numbers; index[] receives the global record indices. */
int gtsg_route_pack(GtsgEngine *e, uint64_t n_records, const uint32_t *root,
                    const uint32_t *ctg, const int64_t *dist, const float *std_dev,
                    const int64_t *num_pairs, const uint8_t *flags,
                    uint64_t first_index, uint64_t n_contigs, const int8_t *owner,
                    uint32_t world, uint64_t *rows, uint64_t *counts);
int gtsg_route_unpack(GtsgEngine *e, uint64_t n_rows, const uint64_t *rows,
                      const uint32_t *loc_of, uint32_t *root, uint32_t *ctg,
                      int64_t *dist, float *std_dev, int64_t *num_pairs,
                      uint8_t *flags, uint64_t *index);
/* the same; *out_of_order = 1 if the rows' record indices are not ascending
   (rows dealt in file-order chunks arrive in file order: the caller sorts only
   when told to) */
int gtsg_route_unpack_ex(GtsgEngine *e, uint64_t n_rows, const uint64_t *rows,
                         const uint32_t *loc_of, uint32_t *root, uint32_t *ctg,
                         int64_t *dist, float *std_dev, int64_t *num_pairs,
                         uint8_t *flags, uint64_t *index, int *out_of_order);

/* ref gt_scaffolder_removecycles, algorithms.c:495-578 */
int gtsg_removecycles(GtsgEngine *e);
/* ref gt_scaffolder_makescaffold, algorithms.c:767-868 */
int gtsg_makescaffold(GtsgEngine *e);

uint64_t gtsg_num_vertices(const GtsgEngine *e);
uint64_t gtsg_num_edges(const GtsgEngine *e);
/* results to HOST buffers; edges in the reference's edge-id order */
int gtsg_get_vertex_states(GtsgEngine *e, uint8_t *out);
int gtsg_get_edge_states(GtsgEngine *e, uint8_t *out);
int gtsg_get_edges(GtsgEngine *e, uint32_t *start, uint32_t *end, int64_t *dist,
                   float *std_dev, int64_t *num_pairs, uint8_t *flags);
/* the adjacency lists: row has num_vertices + 1 offsets into adj, adj holds the
   edge ids of a vertex in creation order (the order of the reference's
   vertex->edges array, graph.c:137-160) */
int gtsg_get_csr(GtsgEngine *e, uint32_t *row, uint32_t *adj);
/* The .dot lines of the edges first .. first + count - 1 (edge-id order) as
   gt_scaffolder_graph_print_generic writes them (ref gt_scaffolder_graph.c:
   288-300), formatted on the device and copied to host_buf (cap bytes; 107 per
   edge always suffice); *nbytes = their length.  At most 2^25 edges a call. */
int gtsg_format_dot_edges(GtsgEngine *e, uint64_t first, uint64_t count, char *host_buf,
                          uint64_t cap, uint64_t *nbytes);
/* the same into a page-locked buffer owned by the engine: *text points at the
   lines until the next call on this engine */
int gtsg_format_dot_edges_pinned(GtsgEngine *e, uint64_t first, uint64_t count,
                                 const char **text, uint64_t *nbytes);
/* The SCAFFOLD edges only, as a compact CSR in adjacency order: what the
   scaffold record walk (ref gt_scaffolder_algorithms.c:901-997) and the .scaf
   writer (:1000-1040) read of the graph after makescaffold.  *count = their
   number; with row == NULL nothing else is written (size the arrays, call
   again).  row: num_vertices + 1 offsets; per edge its id, end vertex,
   distance, deviation and flags (bit 0 sense, bit 1 same).  Host pointers. */
int gtsg_get_scaffold_edges(GtsgEngine *e, uint64_t *count, uint32_t *row, uint32_t *eid,
                            uint32_t *end, int64_t *dist, float *std_dev, uint8_t *flags);
/* ref gt_scaffolder_algorithms.c:901-997 gt_scaffolder_graph_iterate_scaffolds
   on the device.  The SCAFFOLD edges are nearly all vertex-disjoint simple
   paths with a twin per edge; the records of those ("clean" paths, and of the
   unmarked contigs without SCAFFOLD edges) are a list ranking: root vertices in
   index order, the edges of each in walk order, the sequence length of each =
   lengths of its contigs plus the distances between them.  What is not such a
   path -- a walk of makescaffold that passed a contig twice, a hand-built or
   altered graph -- comes back as the "open" part: the vertices that could open
   a record there (index order) and the SCAFFOLD edges of those paths (adjacency
   order, start vertex per edge), closed under SCAFFOLD edges, for the caller to
   walk in the reference's order of visits; the records of both parts merge by
   root index.  gtsg_scaffold_records computes and returns the counts; the
   arrays wait on the device for gtsg_scaffold_records_fetch (host pointers):
   root / off (first edge of a record) / seqlen per record; id, end vertex,
   distance, deviation and flags (bit 0 sense, bit 1 same) per edge. */
typedef struct {
  uint64_t n_records, n_edges;          /* ranked on the device */
  uint64_t n_open_roots, n_open_edges;  /* left to the caller */
} GtsgRecordCounts;
typedef struct {
  uint32_t *root, *off;
  uint64_t *seqlen;
  uint32_t *eid, *end;
  int64_t *dist;
  float *std_dev;
  uint8_t *flags;
  uint32_t *open_root, *open_start, *open_eid, *open_end;
  int64_t *open_dist;
  float *open_std_dev;
  uint8_t *open_flags;
} GtsgRecordArrays;
int gtsg_scaffold_records(GtsgEngine *e, GtsgRecordCounts *counts);
int gtsg_scaffold_records_fetch(GtsgEngine *e, const GtsgRecordArrays *arrays);
/* ref gt_scaffolder_graph.c:174-193 gt_scaffolder_graph_find_edge: the id of
   the first edge in vertex_1's list (creation order) that ends in vertex_2,
   GTSG_NO_EDGE if there is none */
#define GTSG_NO_EDGE UINT64_MAX
int gtsg_find_edge(GtsgEngine *e, uint64_t vertex_1, uint64_t vertex_2, uint64_t *eid);
/* ref gt_scaffolder_graph.c:219-235 gt_scaffolder_graph_alter_edge: new
   distance, deviation, pair count and sense / same for edge `eid`; its state
   and the edge created with it in the other direction are left alone */
int gtsg_alter_edge(GtsgEngine *e, uint64_t eid, int64_t dist, float std_dev, uint64_t num_pairs,
                    int sense, int same);
/* order-independent 64-bit digest of (vertex states, edge states by id),
   computed on the device: used to compare full-size runs */
int gtsg_state_digest(GtsgEngine *e, uint64_t *vertex_digest,
                      uint64_t *edge_digest);

/* diagnostic: evaluates the engine's ambiguous-order test (ref algorithms.c:
   175-193) on n host-side pairs (dist1, std_dev1, dist2, std_dev2) on the
   device; out[i] = 1 if ambiguous.  Used by the parity tests to pin the
   device's int64->float / double sqrt / double divide roundings. */
int gtsg_selftest_ambiguous(GtsgEngine *e, uint64_t n, const int64_t *d1,
                            const float *s1, const int64_t *d2, const float *s2,
                            float pcutoff, uint8_t *out);

/* tuning */
int gtsg_set_option(GtsgEngine *e, const char *name, int64_t value);
/*   Names and defaults (a name that is not known or a value out of range:
     GTSG_EINVAL).  The results do not depend on any of them.
     reference search:
       "fast_walks" (1; 0 forces the reference's label-correcting search for
         every walk), "walk_queue_factor" (8: ring of a search = factor x the
         compact edges of its component to begin with; a search that overflows
         its ring takes one eight times the size from the pool),
       "walk_pool_entries" (2^26: the pool the rings are carved from; if it
         runs out the call restores the states and runs again with eight times
         the pool), "max_walk_pops" (2^32: a search that pops more nodes ends
         the call with GTSG_EWALK, states restored)
     where a component runs:
       "lds_components" (1; 0 runs every component from global memory),
       "pool_components" (1: the LDS-resident components run in ONE launch of a
         workgroup per CU whose "pool_waves" (16, the maximum) wavefronts claim
         components and share the CU's LDS in 1.5 KB pages; 0: one launch per
         LDS size class on "class_streams" (6) side streams),
       "pool_fill_kb" (4: the pool's second claim cursor starts at the components
         of at most this footprint), "pool_wait_limit_us" (10^7: bound of every
         wait inside that launch; a wait that gives up ends the call with
         GTSG_EINTERNAL, states restored),
       "lds_int16_distances" (1: int16 distances in LDS for components whose
         distances all fit), "team_components" (1: a component too large for LDS
         gets a workgroup of eight wavefronts, if there are at most
         "team_max_components" (512) of them and their slabs fit "team_pool_mb"
         (4096))
     the walks of a component:
       "batch_walks" (1: the walks of a cc of a clean component side by side,
         eight lanes each), "batch_big_contigs" (64) / "batch_big_slots" (3):
         LDS for that many walks from that many contigs on,
       "small_masks" (1: sweep order and pending positions of components of at
         most 64 contigs on 64-bit masks)
     walks as tasks (one workgroup per walk, rounds with an in-order select pass):
       "defer_min_contigs" (256) and "defer_min_work" (2^17 terminals x contigs):
         a component that large hands out all its walks (0 = never),
       "defer_ref_min_contigs" (48): a component of that many contigs hands out
         the walks from the first one that needs the reference's search,
       "defer_unclean_work" (2048 terminals x contigs): a component that is not
         clean hands out its walks once another component of the launch has
         deferred, "task_reference_walks" (1: tasks replay the reference's search
         themselves; 0: the select pass does), "mixed_task_limit" (256: a round
         with at most that many pending walks is one launch for all classes),
       "walk_path_entries" (2^24, pool of the tasks' walks and bitmaps; grows by
         itself like the walk queues), "defer_global_components" (0; 1: the
         components in global memory hand their walks out, too, with
         "global_task_pool_mb" (2048) of scratch slabs)
     the build:
       "pair_sort_full" (0: the records are bucketed on ~log2(records) bits of
         their contig pair, three radix passes for 10^8 records; 1: sorted on the
         whole pair, the fallback taken by itself when a thread of the fold
         would look at more than "pair_bucket_limit" (2048) records of a bucket;
         statistic "pair_sort_fallback" counts those),
       "gather_unroll" (4: edges a thread of the gather-shaped build kernels),
       "gather_nt" (1: nontemporal stores for their coalesced outputs)
     other:
       "hub_degree" (32), "team_coff" (0: 1 keeps the list offsets of a team's
       component in LDS while the walks of a cc are made), "team_lds_bytes" (0;
       test aid: a cap on the team kernel's LDS), "lds_poison" (-1; test aid: a byte to fill a component's
       LDS pages with before staging), "profile" (0, 1 = hipEvents around every
       kernel, 2 = also the per-component clocks: "us_sum_*" / "us_max_*", the
       slowest components as "top<r>_*", the size bands as "size_band<b>_*" and
       the rounds of walk tasks as "walk_round<r>_*" statistics) */

/* ---- DistEst text on the GPU (gts_deparse.hip) --------------------------------
   Replaces the two passes of gt_scaffolder_parser.c over the .de file
   (count_distances: integrity check, ref parser.c:150-291; read_distances:
   records in file order, parser.c:295-394) for files in the regular form
   DistanceEst writes.  The records stay on the device and go to
   gtsg_build_from_records_ex with on_device = 1.
     set_names   the contig headers in id order (sorted, ref parser.c:172):
                 `offsets` has n + 1 entries into `blob`
     parse       text: the file's bytes (host or device pointer).  On return
                 irregular != 0: the file is outside the form this parser
                                 reproduces exactly -- parse it on the host;
                 error != 0    : the reference's first error in file order
                                 (1 "Invalid record", 2 "Invalid value for
                                 number of pairs", 3 "Invalid composition
                                 sign") at byte error_pos;
                 otherwise     : n_records records between known contigs
                                 (count_distances' *nof_distances is twice that)
     records     device pointers to the records of the last parse, valid until
                 the next parse or destroy */
typedef struct GtsgDeParser GtsgDeParser;
typedef struct {
  uint64_t n_records, n_candidates, error_pos;
  int error, irregular;
} GtsgDeParseResult;
int gtsg_deparser_create(GtsgDeParser **out, int device, void *hip_stream);
void gtsg_deparser_destroy(GtsgDeParser *p);
const char *gtsg_deparser_last_error(const GtsgDeParser *p);
int gtsg_deparser_set_names(GtsgDeParser *p, const char *blob, const uint64_t *offsets, uint64_t n);
int gtsg_deparser_parse(GtsgDeParser *p, const char *text, uint64_t len, int on_device,
                        GtsgDeParseResult *res);
int gtsg_deparser_records(const GtsgDeParser *p, uint64_t *n, const uint32_t **root,
                          const uint32_t **ctg, const int64_t **dist, const float **std_dev,
                          const int64_t **num_pairs, const uint8_t **flags);
/* the same copied to host arrays of n_records elements (any may be NULL) */
int gtsg_deparser_download(GtsgDeParser *p, uint32_t *root, uint32_t *ctg, int64_t *dist,
                           float *std_dev, int64_t *num_pairs, uint8_t *flags);
/* A-statistic file (ref algorithms.c:108-153: "%s\t%ld\t%ld\t%ld\t%f\t%f" per
   line): astat / copy_num -- arrays over the names, in and out, host pointers or
   (arrays_on_device) device pointers -- take the two values of every line whose
   contig is known.  error 1 = "Invalid record in A-statistic file"; irregular:
   the arrays are untouched, parse the file on the host. */
int gtsg_deparser_parse_astat(GtsgDeParser *p, const char *text, uint64_t len, int on_device,
                              float *astat, float *copy_num, int arrays_on_device,
                              GtsgDeParseResult *res);
/* on != 0: the records of every later parse are appended to those before (a
   file of 4 GB or more handed over in pieces that end at line ends), `records`
   returns all of them; on == 0: back to "the last parse" */
int gtsg_deparser_accumulate(GtsgDeParser *p, int on);
/* frees the text and the records of the last parse; the name table stays */
void gtsg_deparser_trim(GtsgDeParser *p);
/* FASTA record table (ref parser.c:399-494, the description / sequence-length
   callbacks of the reference's two passes over the contig file): for every
   record in file order the offsets of its description (after the '>' up to its
   newline, `len` if the file ends first) and the number of sequence characters
   (everything up to the next record start but newline, carriage return and
   blank; a record starts at the first '>' since the last newline).  The three
   arrays are malloc'ed by the call, the caller frees them. */
int gtsg_fasta_records(int device, const char *text, uint64_t len, uint64_t *n_records,
                       uint64_t **desc_start, uint64_t **desc_end, uint64_t **seq_len);
/* Contig headers into strcmp order (ref parser.c:172, the qsort that makes the
   vertex ids): perm[j] = index of the j-th name by its first 14 bytes, tie[j]
   != 0 where name perm[j] agrees with name perm[j-1] in those bytes -- the caller
   orders such runs with strcmp.  blob / offsets as for set_names; perm and tie
   are host arrays of n elements. */
int gtsg_sort_names(int device, const char *blob, const uint64_t *offsets, uint64_t n,
                    uint32_t *perm, uint8_t *tie);

/* per-kernel timing collected with hipEvents on the engine's stream while
   option "profile" is 1.  Fills up to cap entries, returns the number of
   distinct kernels. */
typedef struct {
  char name[48];
  uint64_t calls;
  double ms;
} GtsgKernelTime;
int gtsg_get_kernel_times(GtsgEngine *e, GtsgKernelTime *out, int cap);
void gtsg_reset_kernel_times(GtsgEngine *e);
/* counters of the last calls: "filter_rounds_p", "filter_rounds_i",
   "components", "max_component", "slots", "compact_edges", "hubs",
   "walk_retries", "fast_walks", "slow_walks", "clean_components",
   "deferred_components", "walk_tasks", "walk_task_rounds", "walk_task_runs";
   "bytes_graph" (HBM held by the contig and CSR arrays) and "bytes_workspace"
   (the scratch region, grown to the largest stage run so far) */
int64_t gtsg_get_stat(const GtsgEngine *e, const char *name);

#ifdef __cplusplus
}
#endif
#endif
