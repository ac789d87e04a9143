/*
  gts_oracle_cli.c -- command-line driver for the CPU oracle (TEST
  INFRASTRUCTURE ONLY).  Mirrors the module dispatch of the reference's test
  driver (ref: src/test.c:56-233: modules graph / parser / scaffold, same
  cut-offs src/test.c:35-42 and the same output file names), so the golden
  .dot files of the reference's test-suite can be diffed directly.
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gts_oracle.h"

int main(int argc, char **argv)
{
  char err[512] = "";
  if (argc >= 2 && strcmp(argv[1], "graph") == 0 && argc == 9) {
    return ora_graph_test(strtoull(argv[2], 0, 10), strtoull(argv[3], 0, 10),
                          atoi(argv[4]), strtoull(argv[5], 0, 10),
                          atoi(argv[6]), strtoull(argv[7], 0, 10),
                          atoi(argv[8]) ? "gt_scaffolder_graph_test.dot" : NULL);
  } else if (argc == 3 && strcmp(argv[1], "parser") == 0) {
    int rc = ora_parser_read_distances_test(argv[2],
               "gt_scaffolder_parser_test_read_distances.de", err, sizeof err);
    if (rc) fprintf(stderr, "ERROR: %s\n", err);
    return rc;
  } else if (argc >= 5 && strcmp(argv[1], "scaffold") == 0) {
    OraGraph *g = NULL;
    OraRecords *recs;
    int lazy = argc >= 6 && strcmp(argv[5], "lazy") == 0;
    int rc = ora_graph_new_from_file(&g, argv[2], 200, argv[3], false, err,
                                     sizeof err);
    if (!rc) rc = ora_mark_repeats(argv[4], g, 0.3f, 20.0f, err, sizeof err);
    if (rc) { fprintf(stderr, "ERROR: %s\n", err); ora_graph_delete(g); return 1; }
    ora_graph_print(g, "gt_scaffolder_algorithms_test_mark_repeats.dot");
    ora_filter(g, 0.01f, 1.5f, 400);
    ora_graph_print(g, "gt_scaffolder_algorithms_test_filter.dot");
    ora_removecycles(g);
    ora_graph_print(g, "gt_scaffolder_algorithms_test_removecycles.dot");
    ora_makescaffold(g, lazy);
    ora_graph_print(g, "gt_scaffolder_algorithms_test_makescaffold.dot");
    recs = ora_iterate_scaffolds(g);
    ora_write_scaffold(g, recs, "gt_scaffolder_new_write.scaf");
    ora_records_delete(recs);
    ora_graph_delete(g);
    return 0;
  }
  fprintf(stderr, "Usage: %s graph|parser|scaffold <arguments>\n", argv[0]);
  return EXIT_FAILURE;
}
