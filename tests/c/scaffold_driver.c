/*
  scaffold_driver.c -- a plain-C caller of the drop-in API (test code).

  The "scaffold" module of the reference's test driver (ref src/test.c:118-199)
  restated on include/gt_scaffolder_host.h: same call sequence, same cut-offs
  (ref src/test.c:35-42), same output file names, GtError replaced by a
  message buffer.  Compiled by tests/test_c_driver.py with
      gcc -std=gnu11 -Iinclude tests/c/scaffold_driver.c -lgtscaffold_hip
  and run on the reference's test data; its four .dot files are compared
  byte for byte with the reference's *_expected.dot.

  usage: scaffold_driver <contigs.fa> <DistEst file> <astat file> [stepwise [matepair]]
    stepwise : build the graph through the four parser.h entry points instead
               of gt_scaffolder_graph_new_from_file (ref graph.c:346-419)
    matepair : ... with read_distances' ismatepair = true
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

int main(int argc, char **argv)
{
  GtScaffolderGraph *graph = NULL;
  GtScaffolderGraphRecords *recs;
  uint64_t *scaf_seqlen = NULL;
  char err[512] = "";
  int had_err;
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
