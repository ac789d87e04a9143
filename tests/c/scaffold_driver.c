/*
  scaffold_driver.c -- a plain-C caller of the drop-in API (test code).

  The "scaffold" module of the reference's test driver (ref src/test.c:118-199)
  restated on include/gt_scaffolder_host.h: same call sequence, same cut-offs
  (ref src/test.c:35-42), same output file names, GtError replaced by a
  message buffer.  Compiled by tests/test_c_driver.py with
      gcc -std=gnu11 -Iinclude tests/c/scaffold_driver.c -lgtscaffold_hip
  and run on the reference's test data; its four .dot files are compared
  byte for byte with the reference's *_expected.dot.

  usage: scaffold_driver <contigs.fa> <DistEst file> <astat file> [stepwise [matepair] | api]
    stepwise : build the graph through the four parser.h entry points instead
               of gt_scaffolder_graph_new_from_file (ref graph.c:346-419)
    matepair : ... with read_distances' ismatepair = true
    api      : also call find_edge / get_vertex / get_vertex_id / alter_edge
               (ref gt_scaffolder_graph.h:127-146) on the graph once it is on
               the GPU: every edge is found from its ends, an edge is altered
               and altered back, so the .dot files still have to be the
               reference's
  usage: scaffold_driver handbuilt
    the same four calls on a hand-built graph (no GPU needed)
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gt_scaffolder_host.h"

#define MIN_CONTIG_LEN 200
#define COPY_NUM_CUTOFF 0.3f
#define ASTAT_NUM_CUTOFF 20.0f
#define PROBABILITY_CUTOFF 0.01f
#define COPY_NUM_CUTOFF_2 1.5f
#define OVERLAP_CUTOFF 400
#define ASTAT_IS_ANNOTATED false

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "api check failed: %s (line %d)\n", #c, __LINE__); \
                                  return -1; } } while (0)

/* find_edge / get_vertex_id / alter_edge against the edge list itself */
static int check_edge_api(GtScaffolderGraph *graph)
{
  uint64_t m = gt_scaffolder_graph_nof_edges(graph), n = gt_scaffolder_graph_nof_vertices(graph), i;
  uint32_t *start = malloc(4 * (m + 1)), *end = malloc(4 * (m + 1));
  int64_t *dist = malloc(8 * (m + 1)), *np = malloc(8 * (m + 1));
  float *sd = malloc(4 * (m + 1));
  uint8_t *fl = malloc(m + 1);
  CHECK(start && end && dist && np && sd && fl);
  CHECK(gt_scaffolder_graph_get_edges(graph, start, end, dist, sd, np, fl) == 0);
  for (i = 0; i < m; i++) {
    /* one edge per (start, end): the first edge of start's list that ends in end is this one */
    CHECK(gt_scaffolder_graph_find_edge(graph, start[i], end[i]) == i);
  }
  CHECK(gt_scaffolder_graph_find_edge(graph, n, 0) == GT_SCAFFOLDER_NO_EDGE);
  CHECK(gt_scaffolder_graph_get_vertex_id(graph, n - 1) == n - 1);
  CHECK(gt_scaffolder_graph_get_vertex_id(graph, n) == GT_SCAFFOLDER_NO_VERTEX);
  CHECK(gt_scaffolder_graph_alter_edge(graph, m, 0, 0.0f, 0, false, false) == -1);
  if (m) {
    uint64_t e = m / 2;
    int64_t d2;
    float s2;
    uint8_t f2;
    CHECK(gt_scaffolder_graph_alter_edge(graph, e, dist[e] + 11, sd[e] + 2.5f, (uint64_t)np[e] + 1,
                                         !(fl[e] & 1), !(fl[e] & 2)) == 0);
    {
      uint32_t *s3 = malloc(4 * m), *e3 = malloc(4 * m);
      int64_t *d3 = malloc(8 * m), *n3 = malloc(8 * m);
      float *sd3 = malloc(4 * m);
      uint8_t *f3 = malloc(m);
      CHECK(gt_scaffolder_graph_get_edges(graph, s3, e3, d3, sd3, n3, f3) == 0);
      d2 = d3[e]; s2 = sd3[e]; f2 = f3[e];
      CHECK(n3[e] == np[e] + 1 && s3[e] == start[e] && e3[e] == end[e]);
      /* nothing else moved */
      for (i = 0; i < m; i++)
        if (i != e) CHECK(d3[i] == dist[i] && sd3[i] == sd[i] && n3[i] == np[i] && f3[i] == fl[i]);
      free(s3); free(e3); free(d3); free(n3); free(sd3); free(f3);
    }
    CHECK(d2 == dist[e] + 11 && s2 == sd[e] + 2.5f && f2 == (uint8_t)((fl[e] ^ 3) & 3));
    CHECK(gt_scaffolder_graph_alter_edge(graph, e, dist[e], sd[e], (uint64_t)np[e], fl[e] & 1,
                                         (fl[e] & 2) != 0) == 0);
  }
  free(start); free(end); free(dist); free(np); free(sd); free(fl);
  return 0;
}

static int handbuilt(void)
{
  GtScaffolderGraph *g = gt_scaffolder_graph_new(4, 6);
  uint64_t v = 99;
  CHECK(g != NULL);
  CHECK(gt_scaffolder_graph_add_vertex(g, "ctg_a", 100, 1.0f, 1.0f) == 0);
  CHECK(gt_scaffolder_graph_add_vertex(g, "ctg_b", 200, 1.0f, 1.0f) == 0);
  CHECK(gt_scaffolder_graph_add_vertex(g, "ctg_c", 300, 1.0f, 1.0f) == 0);
  CHECK(gt_scaffolder_graph_add_vertex(g, "ctg_d", 400, 1.0f, 1.0f) == 0);
  CHECK(gt_scaffolder_graph_add_vertex(g, "ctg_e", 500, 1.0f, 1.0f) == -1);   /* over capacity */
  CHECK(gt_scaffolder_graph_add_edge(g, 0, 1, 10, 1.5f, 3, true, true) == 0);
  CHECK(gt_scaffolder_graph_add_edge(g, 1, 0, 10, 1.5f, 3, false, true) == 0);
  CHECK(gt_scaffolder_graph_add_edge(g, 0, 2, -5, 2.5f, 4, true, false) == 0);
  CHECK(gt_scaffolder_graph_add_edge(g, 2, 0, -5, 2.5f, 4, true, false) == 0);
  CHECK(gt_scaffolder_graph_get_vertex(g, &v, "ctg_c") && v == 2);
  CHECK(!gt_scaffolder_graph_get_vertex(g, &v, "ctg_x") && v == 2);
  CHECK(gt_scaffolder_graph_find_edge(g, 0, 2) == 2);
  CHECK(gt_scaffolder_graph_find_edge(g, 2, 1) == GT_SCAFFOLDER_NO_EDGE);
  CHECK(check_edge_api(g) == 0);
  gt_scaffolder_graph_delete(g);
  printf("handbuilt ok\n");
  return 0;
}

int main(int argc, char **argv)
{
  GtScaffolderGraph *graph = NULL;
  GtScaffolderGraphRecords *recs;
  uint64_t *scaf_seqlen = NULL;
  char err[512] = "";
  int had_err;
  if (argc == 2 && strcmp(argv[1], "handbuilt") == 0) return handbuilt() ? EXIT_FAILURE : EXIT_SUCCESS;
  if (argc < 4 || argc > 6) {
    fprintf(stderr, "Usage: <FASTA-file with contigs> <DistEst file> <astat file> "
                    "[stepwise [matepair]]\n");
    return EXIT_FAILURE;
  }
  if (argc >= 5 && strcmp(argv[4], "stepwise") == 0) {
    uint64_t nof_contigs = 0, nof_distances = 0;
    bool ismatepair = argc == 6 && strcmp(argv[5], "matepair") == 0;
    had_err = gt_scaffolder_parser_count_contigs(argv[1], MIN_CONTIG_LEN, &nof_contigs,
                                                 err, sizeof err);
    if (!had_err) {
      graph = gt_scaffolder_graph_new(nof_contigs, 0);
      had_err = gt_scaffolder_parser_read_contigs(graph, argv[1], MIN_CONTIG_LEN,
                                                  ASTAT_IS_ANNOTATED, err, sizeof err);
    }
    if (!had_err)
      had_err = gt_scaffolder_parser_count_distances(graph, argv[2], &nof_distances,
                                                     err, sizeof err);
    if (!had_err)
      had_err = gt_scaffolder_parser_read_distances(argv[2], graph, ismatepair, err, sizeof err);
    if (!had_err)
      printf("contigs counted %lu, distances counted %lu, vertices %lu, edges %lu\n",
             (unsigned long)nof_contigs, (unsigned long)nof_distances,
             (unsigned long)gt_scaffolder_graph_nof_vertices(graph),
             (unsigned long)gt_scaffolder_graph_nof_edges(graph));
  } else
    had_err = gt_scaffolder_graph_new_from_file(&graph, argv[1], MIN_CONTIG_LEN, argv[2],
                                                ASTAT_IS_ANNOTATED, err, sizeof err);

  if (had_err == 0 && argc == 5 && strcmp(argv[4], "api") == 0) {
    uint64_t v = 0;
    had_err = check_edge_api(graph);
    /* the vertices are in header order: the reference's binary search finds them */
    if (!had_err && !(gt_scaffolder_graph_get_vertex(graph, &v, "contig-4616") &&
                      gt_scaffolder_graph_get_vertex_id(graph, v) == v &&
                      !gt_scaffolder_graph_get_vertex(graph, &v, "no such contig"))) {
      fprintf(stderr, "api check failed: get_vertex\n");
      had_err = -1;
    }
    if (had_err) snprintf(err, sizeof err, "edge / vertex accessors disagree with the edge list");
    else printf("api ok\n");
  }
  if (!ASTAT_IS_ANNOTATED && had_err == 0)
    had_err = gt_scaffolder_graph_mark_repeats(argv[3], graph, COPY_NUM_CUTOFF,
                                               ASTAT_NUM_CUTOFF, err, sizeof err);
  if (had_err == 0) {
    had_err = gt_scaffolder_graph_print(graph, "gt_scaffolder_algorithms_test_mark_repeats.dot",
                                        err, sizeof err);
    if (!had_err) had_err = gt_scaffolder_graph_filter(graph, PROBABILITY_CUTOFF,
                                                       COPY_NUM_CUTOFF_2, OVERLAP_CUTOFF);
    if (!had_err) had_err = gt_scaffolder_graph_print(graph,
                               "gt_scaffolder_algorithms_test_filter.dot", err, sizeof err);
    if (!had_err) had_err = gt_scaffolder_removecycles(graph);
    if (!had_err) had_err = gt_scaffolder_graph_print(graph,
                               "gt_scaffolder_algorithms_test_removecycles.dot", err, sizeof err);
    if (!had_err) had_err = gt_scaffolder_makescaffold(graph);
    if (!had_err) had_err = gt_scaffolder_graph_print(graph,
                               "gt_scaffolder_algorithms_test_makescaffold.dot", err, sizeof err);
    if (!had_err && argc > 4 && strcmp(argv[4], "api") == 0) {
      /* the stream variant (ref gt_scaffolder_graph.h:151-153) writes the same bytes */
      FILE *f = fopen("gt_scaffolder_print_generic.dot", "w");
      had_err = f ? gt_scaffolder_graph_print_generic(graph, f) : -1;
      if (f && fclose(f) != 0) had_err = -1;
      if (had_err) snprintf(err, sizeof err, "print_generic failed");
      else printf("print_generic ok\n");
    }
    if (had_err && !err[0]) snprintf(err, sizeof err, "%s", gt_scaffolder_graph_last_error(graph));
  }
  if (had_err == 0) {
    recs = gt_scaffolder_graph_iterate_scaffolds(graph, &scaf_seqlen);
    if (!recs) {
      had_err = -1;
      snprintf(err, sizeof err, "%s", gt_scaffolder_graph_last_error(graph));
    } else {
      had_err = gt_scaffolder_graph_write_scaffold(recs, "gt_scaffolder_new_write.scaf",
                                                   err, sizeof err);
      printf("scaffolds %lu\n", (unsigned long)gt_scaffolder_graph_records_size(recs));
      gt_scaffolder_graph_records_delete(recs);
      free(scaf_seqlen);
    }
  }
  if (had_err != 0) fprintf(stderr, "ERROR: %s\n", err);
  gt_scaffolder_graph_delete(graph);
  return had_err ? EXIT_FAILURE : EXIT_SUCCESS;
}
