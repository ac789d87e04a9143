/*
  gts_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see gts_oracle.h).

  A restatement, with vertex/edge indices instead of pointers, of the
  reference's single-threaded algorithms.  "ref:" comments give the reference
  file:line each block follows.  Quirks of the reference that influence results
  (order of marking, last-writer-wins edge states, float distance maps, LIFO
  terminal evaluation, strict-greater tie-breaks) are kept on purpose.
*/
#define _GNU_SOURCE
#include "gts_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORA_BUFSIZE 1024 /* ref: gt_scaffolder_parser.c:30, algorithms.c:96 */

static void *xmalloc(size_t n)
{
  void *p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}
static void *xrealloc(void *q, size_t n)
{
  void *p = realloc(q, n ? n : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}
static void set_err(char *err, size_t n, const char *msg, const char *arg)
{
  if (err && n) snprintf(err, n, msg, arg ? arg : "");
}

/* ------------------------------------------------------------------ */
/* graph container: ref gt_scaffolder_graph.c:32-131                   */

OraGraph *ora_graph_new(uint64_t max_v, uint64_t max_e)
{
  OraGraph *g = xmalloc(sizeof *g);
  g->nv = 0; g->cap_v = max_v;
  g->ne = 0; g->cap_e = max_e;
  g->v = xmalloc(sizeof *g->v * (max_v ? max_v : 1));
  g->e = xmalloc(sizeof *g->e * (max_e ? max_e : 1));
  return g;
}

void ora_graph_delete(OraGraph *g)
{
  uint64_t i;
  if (!g) return;
  for (i = 0; i < g->nv; i++) { free(g->v[i].header); free(g->v[i].edges); }
  free(g->v); free(g->e); free(g);
}

void ora_graph_add_vertex(OraGraph *g, const char *header, uint64_t seq_len,
                          float astat, float copy_num)
{
  OraVertex *v;
  if (g->nv == g->cap_v) {  /* the reference asserts; the oracle grows */
    g->cap_v = g->cap_v ? 2 * g->cap_v : 16;
    g->v = xrealloc(g->v, sizeof *g->v * g->cap_v);
  }
  v = g->v + g->nv++;
  v->header = strdup(header ? header : "");
  v->seq_len = seq_len; v->astat = astat; v->copy_num = copy_num;
  v->nof_edges = 0; v->cap_edges = 0; v->edges = NULL;
  v->state = ORA_UNVISITED;
}

void ora_graph_add_edge(OraGraph *g, uint64_t vstart, uint64_t vend,
                        int64_t dist, float std_dev, uint64_t num_pairs,
                        bool dir, bool same)
{
  OraEdge *e;
  OraVertex *s;
  if (g->ne == g->cap_e) {
    g->cap_e = g->cap_e ? 2 * g->cap_e : 16;
    g->e = xrealloc(g->e, sizeof *g->e * g->cap_e);
  }
  e = g->e + g->ne;
  e->start = vstart; e->end = vend; e->dist = dist; e->std_dev = std_dev;
  e->num_pairs = num_pairs; e->sense = dir; e->same = same;
  e->state = ORA_UNVISITED;
  s = g->v + vstart;  /* ref graph.c:164-167: append to the start's list */
  if (s->nof_edges == s->cap_edges) {
    s->cap_edges = s->cap_edges ? 2 * s->cap_edges : 4;
    s->edges = xrealloc(s->edges, sizeof *s->edges * s->cap_edges);
  }
  s->edges[s->nof_edges++] = g->ne;
  g->ne++;
}

/* ref graph.c:173-184; returns edge id or UINT64_MAX */
static uint64_t find_edge(const OraGraph *g, uint64_t v1, uint64_t v2)
{
  const OraVertex *a = g->v + v1;
  uint64_t k;
  for (k = 0; k < a->nof_edges; k++)
    if (g->e[a->edges[k]].end == v2) return a->edges[k];
  return UINT64_MAX;
}

/* ref graph.c:187-216 (binary search over the header-sorted vertices;
   gt_str_cmp on headers without embedded NULs behaves as strcmp) */
static bool get_vertex(const OraGraph *g, uint64_t *out, const char *header)
{
  int64_t lo = 0, hi = (int64_t)g->nv - 1;
  while (hi >= lo) {
    int64_t mid = lo + (hi - lo) / 2;
    int c = strcmp(g->v[mid].header, header);
    if (c == 0) { *out = (uint64_t)mid; return true; }
    if (c < 0) lo = mid + 1; else hi = mid - 1;
  }
  return false;
}

/* ref parser.c:357-378; ismatepair is the argument of
   gt_scaffolder_parser_read_distances (parser.c:297; new_from_file passes
   false, graph.c:399) */
void ora_graph_add_record_mp(OraGraph *g, uint64_t root, uint64_t ctg,
                             int64_t dist, float std_dev, uint64_t num_pairs,
                             bool sense, bool same, bool ismatepair)
{
  uint64_t eid = find_edge(g, root, ctg);
  if (eid != UINT64_MAX) {
    OraEdge *e = g->e + eid;
    if (!ismatepair && e->std_dev < std_dev) {    /* parser.c:362, graph.c:219-235 alter_edge */
      e->dist = dist; e->std_dev = std_dev; e->num_pairs = num_pairs;
      e->sense = sense; e->same = same;
    }
  } else {
    bool twin_dir = same ? !sense : sense;
    ora_graph_add_edge(g, root, ctg, dist, std_dev, num_pairs, sense, same);
    ora_graph_add_edge(g, ctg, root, dist, std_dev, num_pairs, twin_dir, same);
  }
}

void ora_graph_add_record(OraGraph *g, uint64_t root, uint64_t ctg,
                          int64_t dist, float std_dev, uint64_t num_pairs,
                          bool sense, bool same)
{
  ora_graph_add_record_mp(g, root, ctg, dist, std_dev, num_pairs, sense, same, false);
}

void ora_graph_add_records_mp(OraGraph *g, uint64_t n, const uint32_t *root,
                              const uint32_t *ctg, const int64_t *dist,
                              const float *std_dev, const uint64_t *num_pairs,
                              const uint8_t *flags, bool ismatepair)
{
  uint64_t k;
  for (k = 0; k < n; k++)
    ora_graph_add_record_mp(g, root[k], ctg[k], dist[k], std_dev[k],
                            num_pairs ? num_pairs[k] : 0,
                            (flags[k] & 1) != 0, (flags[k] & 2) != 0, ismatepair);
}

void ora_graph_add_records(OraGraph *g, uint64_t n, const uint32_t *root,
                           const uint32_t *ctg, const int64_t *dist,
                           const float *std_dev, const uint64_t *num_pairs,
                           const uint8_t *flags)
{
  ora_graph_add_records_mp(g, n, root, ctg, dist, std_dev, num_pairs, flags, false);
}

/* ------------------------------------------------------------------ */
/* printing: ref gt_scaffolder_graph.c:269-307                         */

static const char *const color_array[] = {"black", "gray80", "gainsboro",
  "ivory3", "red", "green", "magenta", "blue"};

int ora_graph_print(const OraGraph *g, const char *filename)
{
  FILE *f = fopen(filename, "w");
  uint64_t i;
  if (!f) return -1;
  fprintf(f, "digraph {\n");
  for (i = 0; i < g->nv; i++)
    fprintf(f, "%lu [color=\"%s\" label=\"%s\"];\n", (unsigned long)i,
            color_array[g->v[i].state], g->v[i].header);
  for (i = 0; i < g->ne; i++) {
    const OraEdge *e = g->e + i;
    fprintf(f, "%lu -> %lu [color=\"%s\" label=\"%ld\" arrowhead=\"%s\"];\n",
            (unsigned long)e->start, (unsigned long)e->end,
            color_array[e->state], (long)e->dist, e->sense ? "normal" : "inv");
  }
  fprintf(f, "}\n");
  fclose(f);
  return 0;
}

/* ref graph.c:421-500 */
int ora_graph_test(uint64_t max_v, uint64_t max_e, bool init_v, uint64_t nv,
                   bool init_e, uint64_t ne, const char *dot_out)
{
  OraGraph *g = ora_graph_new(max_v, max_e);
  uint64_t i;
  if (init_v) {
    if (nv > max_v) { ora_graph_delete(g); return 2; }  /* gt_assert -> abort */
    for (i = 0; i < nv; i++) ora_graph_add_vertex(g, "foobar", 100, 20, 40);
  }
  if (init_e) {
    uint64_t v1 = 0, v2 = 0;
    if (ne > max_e) { ora_graph_delete(g); return 2; }
    for (i = 0; i < ne; i++) {
      if ((int64_t)v2 < (int64_t)nv - 1) v2++;
      else if ((int64_t)v1 < (int64_t)nv - 2) { v1++; v2 = v1 + 1; }
      ora_graph_add_edge(g, v1, v2, 2, 1.5f, 4, true, true);
    }
  }
  if (dot_out) ora_graph_print(g, dot_out);
  ora_graph_delete(g);
  return 0;
}

/* ------------------------------------------------------------------ */
/* FASTA reading.  The reference drives GenomeTools' recursive-descent FASTA
   reader (core/fasta_reader_rec.h; GenomeTools is a dependency NOT present in
   /root/reference, its version is not pinned by the reference: README.md:26-30
   asks for branch gt_scaffolder of dorleosterode/genometools).  Restated from
   its documented behaviour: a record starts with '>', the description is the
   rest of that line, the sequence is every following character up to the next
   '>' except newlines, carriage returns and blanks; callbacks
   (description, length) then (sequence length). */

typedef int (*FastaCb)(const char *desc, uint64_t desc_len, uint64_t seq_len,
                       void *data);

static int fasta_run(const char *filename, FastaCb cb, void *data, char *err,
                     size_t errlen)
{
  FILE *f = fopen(filename, "rb");
  char *desc = NULL;
  size_t dcap = 0;
  int c, had_err = 0;
  if (!f) { set_err(err, errlen, "cannot open file %s", filename); return -1; }
  c = fgetc(f);
  if (c == EOF) { fclose(f); set_err(err, errlen, "sequence file %s is empty",
                                      filename); return -1; }
  if (c != '>') {
    fclose(f);
    set_err(err, errlen, "the first character of fasta file %s has to be '>'",
            filename);
    return -1;
  }
  while (!had_err && c == '>') {
    size_t dl = 0;
    uint64_t seq_len = 0;
    while ((c = fgetc(f)) != EOF && c != '\n') {
      if (dl + 2 > dcap) { dcap = dcap ? 2 * dcap : 256;
                           desc = xrealloc(desc, dcap); }
      desc[dl++] = (char)c;
    }
    if (dl && desc[dl - 1] == '\r') dl--;
    if (dl + 1 > dcap) { dcap = dl + 16; desc = xrealloc(desc, dcap); }
    desc[dl] = '\0';
    while ((c = fgetc(f)) != EOF && c != '>')
      if (c != '\n' && c != '\r' && c != ' ') seq_len++;
    had_err = cb(desc, dl, seq_len, data);
    if (had_err) set_err(err, errlen, "invalid FASTA record in %s", filename);
  }
  free(desc);
  fclose(f);
  return had_err;
}

typedef struct {
  uint64_t nof_valid_ctg, min_ctg_len;
  OraGraph *graph;
  bool astat_is_annotated;
  bool count_only;
} FastaData;

/* ref parser.c:399-494: count (>=), save header (cut at first blank, optional
   annotated astat), save contig (>) */
static int fasta_cb(const char *desc, uint64_t desc_len, uint64_t seq_len,
                    void *data)
{
  FastaData *d = data;
  if (d->count_only) {
    if (seq_len >= d->min_ctg_len) d->nof_valid_ctg++;  /* parser.c:408 */
    return seq_len == 0 ? -1 : 0;
  } else {
    float astat = 0.0f, copynum = 0.0f;
    char *hdr, *sp;
    if (d->astat_is_annotated) {      /* parser.c:438-450 */
      char part1[ORA_BUFSIZE];
      long n1, n2;
      if (sscanf(desc, "%1023s length=%ld depth=%ld k=%f astat=%f", part1, &n1,
                 &n2, &copynum, &astat) != 5)
        return -1;
    }
    if (desc_len == 0 || seq_len == 0) return -1;
    hdr = strdup(desc);
    sp = strchr(hdr, ' ');
    if (sp) *sp = '\0';
    if (seq_len > d->min_ctg_len)     /* parser.c:481 (strict) */
      ora_graph_add_vertex(d->graph, hdr, seq_len, astat, copynum);
    free(hdr);
    return 0;
  }
}

static int vertex_cmp(const void *a, const void *b)
{
  return strcmp(((const OraVertex *)a)->header, ((const OraVertex *)b)->header);
}

/* ref parser.c:150-291: integrity check of the abyss-dist format.  The edge
   counting there only sizes allocations (the oracle's arrays grow), the error
   conditions are kept. */
static int count_distances(const OraGraph *g, const char *file_name,
                           uint64_t *nof_distances, char *err, size_t errlen)
{
  FILE *file = fopen(file_name, "rb");
  char line[ORA_BUFSIZE + 1], ctg_header[ORA_BUFSIZE + 1], *field;
  uint64_t record_counter = 0, root, ctg;
  long dist, num_pairs;
  float std_dev;
  int had_err = 0;
  if (!file) {
    set_err(err, errlen, "can not read distance file %s", file_name);
    return -1;
  }
  while (fgets(line, ORA_BUFSIZE, file) != NULL) {
    bool valid;
    uint64_t line_records = 0;
    field = strtok(line, " ");
    valid = field && get_vertex(g, &root, field);
    field = strtok(NULL, " ");
    if (field == NULL) {
      set_err(err, errlen, "Invalid record in dist file %s", file_name);
      had_err = -1; break;
    }
    if (!valid) continue;
    while (field != NULL) {
      if (sscanf(field, "%[^>,],%ld,%ld,%f", ctg_header, &dist, &num_pairs,
                 &std_dev) == 4) {
        char sign;
        if (num_pairs < 0) {
          set_err(err, errlen, "Invalid value for number of pairs in dist "
                  "file %s", file_name);
          had_err = -1; break;
        }
        sign = ctg_header[strlen(ctg_header) - 1];
        if (sign != '+' && sign != '-') {
          set_err(err, errlen, "Invalid composition sign in dist file %s",
                  file_name);
          had_err = -1; break;
        }
        ctg_header[strlen(ctg_header) - 1] = '\0';
        if (get_vertex(g, &ctg, ctg_header)) line_records += 2;
      } else if (*field != ';') {
        set_err(err, errlen, "Invalid record in dist file %s", file_name);
        had_err = -1; break;
      }
      field = strtok(NULL, " ");
    }
    if (had_err) break;
    record_counter += line_records;
  }
  fclose(file);
  if (record_counter == 0 && !had_err) {
    set_err(err, errlen, "distance file %s is empty", file_name);
    had_err = -1;
  }
  *nof_distances = record_counter;
  return had_err;
}

/* ref parser.c:295-394 */
static int read_distances(const char *filename, OraGraph *g, bool ismatepair,
                          char *err, size_t errlen)
{
  FILE *file = fopen(filename, "rb");
  char line[ORA_BUFSIZE + 1], ctg_header[ORA_BUFSIZE + 1], *field;
  long dist, num_pairs;
  float std_dev;
  uint64_t root, ctg;
  if (!file) {
    set_err(err, errlen, " can not read distance file %s ", filename);
    return -1;
  }
  while (fgets(line, ORA_BUFSIZE, file) != NULL) {
    bool sense = true;
    line[strlen(line) - 1] = '\0';      /* parser.c:325: drops the last char */
    field = strtok(line, " ");
    if (!field || !get_vertex(g, &root, field)) continue;
    while (field != NULL) {
      if (sscanf(field, "%[^>,],%ld,%ld,%f", ctg_header, &dist, &num_pairs,
                 &std_dev) == 4) {
        size_t len = strlen(ctg_header);
        bool same = ctg_header[len - 1] == '+';
        ctg_header[len - 1] = '\0';
        if (get_vertex(g, &ctg, ctg_header))
          ora_graph_add_record_mp(g, root, ctg, dist, std_dev,
                                  (uint64_t)num_pairs, sense, same, ismatepair);
      } else if (*field == ';')
        sense = !sense;
      field = strtok(NULL, " ");
    }
  }
  fclose(file);
  return 0;
}

int ora_graph_new_from_file(OraGraph **out, const char *ctg_filename,
                            uint64_t min_ctg_len, const char *dist_filename,
                            bool astat_is_annotated, char *err, size_t errlen)
{
  return ora_graph_new_from_file_mp(out, ctg_filename, min_ctg_len, dist_filename,
                                    astat_is_annotated, false, err, errlen);
}

/* the same five steps (graph.c:346-419) with read_distances' ismatepair
   argument chosen by the caller (graph.c:399 passes false) */
int ora_graph_new_from_file_mp(OraGraph **out, const char *ctg_filename,
                               uint64_t min_ctg_len, const char *dist_filename,
                               bool astat_is_annotated, bool ismatepair,
                               char *err, size_t errlen)
{
  FastaData d;
  OraGraph *g = NULL;
  uint64_t nof_distances = 0;
  int had_err;
  *out = NULL;
  d.nof_valid_ctg = 0; d.min_ctg_len = min_ctg_len; d.graph = NULL;
  d.astat_is_annotated = astat_is_annotated; d.count_only = true;
  had_err = fasta_run(ctg_filename, fasta_cb, &d, err, errlen);
  if (!had_err) {
    g = ora_graph_new(d.nof_valid_ctg ? d.nof_valid_ctg : 1, 16);
    d.graph = g; d.count_only = false;
    had_err = fasta_run(ctg_filename, fasta_cb, &d, err, errlen);
  }
  if (!had_err) {
    /* parser.c:172 sorts before validating the distance file */
    qsort(g->v, g->nv, sizeof *g->v, vertex_cmp);
    had_err = count_distances(g, dist_filename, &nof_distances, err, errlen);
  }
  if (!had_err) had_err = read_distances(dist_filename, g, ismatepair, err, errlen);
  if (had_err) { ora_graph_delete(g); g = NULL; }
  *out = g;
  return had_err;
}

/* ref parser.c:55-147 */
int ora_parser_read_distances_test(const char *filename, const char *out,
                                   char *err, size_t errlen)
{
  FILE *file = fopen(filename, "rb"), *f;
  char line[ORA_BUFSIZE + 1], ctg_header[ORA_BUFSIZE + 1], *field;
  long dist, num_pairs;
  float std_dev;
  int had_err = 0;
  if (!file) {
    set_err(err, errlen, "can not read distance file %s", filename);
    return -1;
  }
  f = fopen(out, "w");
  if (!f) { fclose(file); return -1; }
  while (fgets(line, ORA_BUFSIZE, file) != NULL) {
    bool sense = true, first_antisense = true;
    line[strlen(line) - 1] = '\0';
    field = strtok(line, " ");
    fprintf(f, "%s", field ? field : "(null)");
    while (field != NULL) {
      if (sscanf(field, "%[^>,],%ld,%ld,%f", ctg_header, &dist, &num_pairs,
                 &std_dev) == 4) {
        size_t len;
        bool same;
        if (num_pairs < 0) {
          set_err(err, errlen, "Invalid value for number of pairs%s", NULL);
          had_err = -1; break;
        }
        len = strlen(ctg_header);
        same = ctg_header[len - 1] == '+';
        ctg_header[len - 1] = '\0';
        fprintf(f, " %s%c,%ld,%ld,%.1f", ctg_header, same ? '+' : '-', dist,
                num_pairs, std_dev);
      } else if (*field == ';')
        sense = !sense;
      field = strtok(NULL, " ");
      if (!sense && first_antisense) { fprintf(f, " ;"); first_antisense = false; }
    }
    if (had_err) break;
    if (sense) fprintf(f, " ;");
    fprintf(f, "\n");
  }
  fclose(f);
  fclose(file);
  return had_err;
}

/* ------------------------------------------------------------------ */
/* marking helpers: ref gt_scaffolder_algorithms.c:38-87               */

static bool vertex_is_marked(const OraGraph *g, uint64_t v)
{
  uint8_t s = g->v[v].state;
  return s == ORA_POLYMORPHIC || s == ORA_REPEAT || s == ORA_CYCLIC;
}
static bool edge_is_marked(const OraGraph *g, uint64_t e)
{
  uint8_t s = g->e[e].state;
  return s == ORA_INCONSISTENT || s == ORA_POLYMORPHIC || s == ORA_CYCLIC ||
         s == ORA_REPEAT;
}
static void mark_edge(OraGraph *g, uint64_t e, uint8_t state)
{
  const OraVertex *end = g->v + g->e[e].end;
  uint64_t k;
  g->e[e].state = state;
  for (k = 0; k < end->nof_edges; k++)
    if (g->e[end->edges[k]].end == g->e[e].start)
      g->e[end->edges[k]].state = state;
}
static void mark_vertex(OraGraph *g, uint64_t v, uint8_t state)
{
  uint64_t k;
  g->v[v].state = state;
  for (k = 0; k < g->v[v].nof_edges; k++) mark_edge(g, g->v[v].edges[k], state);
}

/* ref algorithms.c:155-167 */
void ora_mark_repeats_loaded(OraGraph *g, bool have_file, float copy_num_cutoff,
                             float astat_cutoff)
{
  uint64_t v;
  for (v = 0; v < g->nv; v++)
    if (g->v[v].astat <= astat_cutoff ||
        (have_file && g->v[v].copy_num < copy_num_cutoff))
      mark_vertex(g, v, ORA_REPEAT);
}

/* ref algorithms.c:90-170 */
int ora_mark_repeats(const char *filename, OraGraph *g, float copy_num_cutoff,
                     float astat_cutoff, char *err, size_t errlen)
{
  int had_err = 0;
  if (strlen(filename) != 0) {
    FILE *file = fopen(filename, "rb");
    char line[ORA_BUFSIZE + 1], hdr[ORA_BUFSIZE + 1];
    if (!file) {
      set_err(err, errlen, "can not read A-statistic file %s", filename);
      return -1;
    }
    while (fgets(line, ORA_BUFSIZE, file) != NULL) {
      long n1 = 0, n2 = 0, n3 = 0;
      float copy_num = 0.0f, astat = 0.0f;
      uint64_t ctg;
      line[strlen(line) - 1] = '\0';
      if (sscanf(line, "%s\t%ld\t%ld\t%ld\t%f\t%f", hdr, &n1, &n2, &n3,
                 &copy_num, &astat) == 6) {
        if (get_vertex(g, &ctg, hdr)) {
          g->v[ctg].astat = astat;
          g->v[ctg].copy_num = copy_num;
        }
      } else {
        set_err(err, errlen, "Invalid record in A-statistic file %s", filename);
        had_err = -1;
        break;
      }
    }
    fclose(file);
  }
  if (!had_err)
    ora_mark_repeats_loaded(g, strlen(filename) != 0, copy_num_cutoff,
                            astat_cutoff);
  return had_err;
}

/* ------------------------------------------------------------------ */
/* filter: ref gt_scaffolder_algorithms.c:172-343                      */

bool ora_ambiguous_from_interval(float interval, float cutoff)
{
  /* ref algorithms.c:188-192, float variables, double libm erf */
  float prob12, prob21, p_wrong;
  prob12 = 0.5 * (1 + erf(interval));
  prob21 = 1.0 - prob12;
  p_wrong = 1.0 - (prob12 > prob21 ? prob12 : prob21);
  return p_wrong > cutoff;
}

bool ora_ambiguousorder(int64_t dist1, float sd1, int64_t dist2, float sd2,
                        float cutoff)
{
  /* ref algorithms.c:184-187 */
  float expval, variance, interval;
  expval = dist1 - dist2;
  variance = 2 * ((sd1 * sd1) + (sd2 * sd2));
  interval = (0 - expval) / sqrt(variance);
  return ora_ambiguous_from_interval(interval, cutoff);
}

/* ref algorithms.c:197-220 */
static int64_t calculate_overlap(const OraGraph *g, const OraEdge *e1,
                                 const OraEdge *e2)
{
  int64_t overlap = 0;
  int64_t start1 = e1->dist, start2 = e2->dist;
  int64_t end1 = e1->dist + (int64_t)g->v[e1->end].seq_len - 1;
  int64_t end2 = e2->dist + (int64_t)g->v[e2->end].seq_len - 1;
  if (start2 <= end1 && start1 <= end2) {
    int64_t is = start1 > start2 ? start1 : start2;
    int64_t ie = end1 < end2 ? end1 : end2;
    overlap = ie - is + 1;
  }
  return overlap;
}

/* ref algorithms.c:223-246 */
static void check_mark_polymorphic(OraGraph *g, const OraEdge *e1,
                                   const OraEdge *e2, float pcutoff,
                                   float cncutoff)
{
  if (ora_ambiguousorder(e1->dist, e1->std_dev, e2->dist, e2->std_dev,
                         pcutoff) &&
      (g->v[e1->end].copy_num + g->v[e2->end].copy_num) < cncutoff) {
    uint64_t poly = g->v[e1->end].copy_num < g->v[e2->end].copy_num
                    ? e1->end : e2->end;
    if (!vertex_is_marked(g, poly)) mark_vertex(g, poly, ORA_POLYMORPHIC);
  }
}

/* ref algorithms.c:249-258 */
static void mark_edges_in_twin_dir(OraGraph *g, uint64_t v, bool sense)
{
  uint64_t k;
  for (k = 0; k < g->v[v].nof_edges; k++)
    if (g->e[g->v[v].edges[k]].sense == sense)
      g->e[g->v[v].edges[k]].state = ORA_INCONSISTENT;
}

void ora_filter(OraGraph *g, float pcutoff, float cncutoff, int64_t ocutoff)
{
  uint64_t v, i, j;
  for (v = 0; v < g->nv; v++) {
    OraVertex *vx = g->v + v;
    int64_t sense_max = 0, anti_max = 0;
    if (vertex_is_marked(g, v)) continue;
    for (i = 0; i < vx->nof_edges; i++)
      for (j = i + 1; j < vx->nof_edges; j++) {
        const OraEdge *e1 = g->e + vx->edges[i], *e2 = g->e + vx->edges[j];
        if (e1->sense == e2->sense)
          check_mark_polymorphic(g, e1, e2, pcutoff, cncutoff);
      }
    if (vertex_is_marked(g, v)) continue;
    for (i = 0; i < vx->nof_edges; i++)
      for (j = i + 1; j < vx->nof_edges; j++) {
        uint64_t a = vx->edges[i], b = vx->edges[j];
        const OraEdge *e1 = g->e + a, *e2 = g->e + b;
        if (e1->sense == e2->sense && !edge_is_marked(g, a) &&
            !edge_is_marked(g, b)) {
          int64_t ov = calculate_overlap(g, e1, e2);
          if (e1->sense && ov > sense_max) sense_max = ov;
          if (!e1->sense && ov > anti_max) anti_max = ov;
        }
      }
    if (sense_max > ocutoff || anti_max > ocutoff) {
      for (i = 0; i < vx->nof_edges; i++) {
        OraEdge *e = g->e + vx->edges[i];
        if (sense_max > ocutoff && e->sense) {
          e->state = ORA_INCONSISTENT;
          mark_edges_in_twin_dir(g, e->end, !e->same);
        }
        if (anti_max > ocutoff && !e->sense) {
          e->state = ORA_INCONSISTENT;
          mark_edges_in_twin_dir(g, e->end, e->same);
        }
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* connected components and terminals: ref algorithms.c:346-436        */

static bool isterminal(const OraGraph *g, uint64_t v)
{
  const OraVertex *vx = g->v + v;
  bool dir = false, set_dir = false;
  uint64_t k;
  if (vx->nof_edges == 0) return true;
  for (k = 0; k < vx->nof_edges; k++) {
    uint64_t e = vx->edges[k];
    if (set_dir) {
      if (g->e[e].sense != dir && !edge_is_marked(g, e)) return false;
    } else if (!edge_is_marked(g, e)) {
      dir = g->e[e].sense;
      set_dir = true;
    }
  }
  return true;
}

/* a growable list of vertex ids / ccs as (offsets into one flat list) */
typedef struct { uint64_t *a; uint64_t n, cap; } U64Vec;
static void vec_push(U64Vec *v, uint64_t x)
{
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 64;
                        v->a = xrealloc(v->a, sizeof *v->a * v->cap); }
  v->a[v->n++] = x;
}

/* terminals of all ccs flattened into term, cc i = term[ccoff[i]..ccoff[i+1]) */
static void calc_cc_and_terminals(OraGraph *g, U64Vec *term, U64Vec *ccoff)
{
  uint64_t v, k, qh;
  U64Vec queue = {0};
  term->n = 0; ccoff->n = 0;
  for (v = 0; v < g->nv; v++)
    if (!vertex_is_marked(g, v)) g->v[v].state = ORA_UNVISITED;
  for (v = 0; v < g->nv; v++) {
    if (vertex_is_marked(g, v) || g->v[v].state == ORA_VISITED) continue;
    g->v[v].state = ORA_PROCESSED;
    queue.n = 0; qh = 0;
    vec_push(&queue, v);
    vec_push(ccoff, term->n);
    while (qh < queue.n) {
      uint64_t cur = queue.a[qh++];
      if (isterminal(g, cur)) vec_push(term, cur);
      g->v[cur].state = ORA_VISITED;
      for (k = 0; k < g->v[cur].nof_edges; k++) {
        uint64_t e = g->v[cur].edges[k];
        if (!edge_is_marked(g, e)) {
          uint64_t nx = g->e[e].end;
          if (vertex_is_marked(g, nx)) continue;
          if (g->v[nx].state == ORA_UNVISITED) {
            g->v[nx].state = ORA_PROCESSED;
            vec_push(&queue, nx);
          }
        }
      }
    }
  }
  vec_push(ccoff, term->n);
  free(queue.a);
}

/* ------------------------------------------------------------------ */
/* cycle detection: ref algorithms.c:438-578.  The reference recurses; the
   oracle keeps an explicit stack with the same visiting order. */

typedef struct { uint64_t v, p, eid; bool dir; } DfsFrame;

static uint64_t detect_cycle(OraGraph *g, uint64_t start, bool dir,
                             U64Vec *visited, DfsFrame **stk, uint64_t *cap)
{
  uint64_t sp = 0;
#define PUSH(V, P, D) do { \
    if (sp == *cap) { *cap = *cap ? 2 * *cap : 64; \
                      *stk = xrealloc(*stk, sizeof **stk * *cap); } \
    (*stk)[sp].v = (V); (*stk)[sp].p = (P); (*stk)[sp].dir = (D); \
    (*stk)[sp].eid = 0; sp++; \
    vec_push(visited, (V)); g->v[(V)].state = ORA_VISITED; } while (0)
  PUSH(start, UINT64_MAX, dir);
  while (sp > 0) {
    DfsFrame *fr = *stk + (sp - 1);
    const OraVertex *vx = g->v + fr->v;
    bool descended = false;
    while (fr->eid < vx->nof_edges) {
      uint64_t b = vx->edges[fr->eid++];
      const OraEdge *back = g->e + b;
      if (back->sense == fr->dir && !edge_is_marked(g, b) &&
          back->end != fr->p && !vertex_is_marked(g, back->end)) {
        if (g->v[back->end].state == ORA_VISITED) return b;
        if (g->v[back->end].state == ORA_UNVISITED) {
          bool next_dir = back->same ? back->sense : !back->sense;
          uint64_t cur = fr->v;
          PUSH(back->end, cur, next_dir);
          descended = true;
          break;
        }
      }
    }
    if (!descended) { g->v[(*stk)[sp - 1].v].state = ORA_PROCESSED; sp--; }
  }
#undef PUSH
  return UINT64_MAX;
}

void ora_removecycles(OraGraph *g)
{
  bool found_cycle = true;
  U64Vec term = {0}, ccoff = {0}, visited = {0};
  DfsFrame *stk = NULL;
  uint64_t cap = 0, v, i, j, k;
  while (found_cycle) {
    found_cycle = false;
    calc_cc_and_terminals(g, &term, &ccoff);
    for (v = 0; v < g->nv; v++)
      if (!vertex_is_marked(g, v)) g->v[v].state = ORA_UNVISITED;
    for (i = 0; i + 1 < ccoff.n; i++)
      for (j = ccoff.a[i]; j < ccoff.a[i + 1]; j++) {
        uint64_t start = term.a[j], back;
        bool dir = true, set_dir = false;
        if (g->v[start].nof_edges == 0) continue;
        for (k = 0; k < g->v[start].nof_edges; k++) {
          uint64_t e = g->v[start].edges[k];
          if (!edge_is_marked(g, e)) { dir = g->e[e].sense; set_dir = true; }
        }
        if (!set_dir) continue;
        if (vertex_is_marked(g, start)) continue;
        visited.n = 0;
        back = detect_cycle(g, start, dir, &visited, &stk, &cap);
        for (k = 0; k < visited.n; k++) g->v[visited.a[k]].state = ORA_UNVISITED;
        if (back != UINT64_MAX) {
          found_cycle = true;
          mark_vertex(g, g->e[back].start, ORA_CYCLIC);
          mark_vertex(g, g->e[back].end, ORA_CYCLIC);
        }
      }
  }
  free(term.a); free(ccoff.a); free(visited.a); free(stk);
}

/* ------------------------------------------------------------------ */
/* walks and scaffolds: ref algorithms.c:580-868                       */

typedef struct { uint64_t *edges; uint64_t n, cap, total_contig_len; } Walk;

static bool is_twin(const OraEdge *a, const OraEdge *b)
{
  return a->start == b->end && a->end == b->start;
}

typedef struct { uint64_t edge; int64_t dist; } WalkNode;

typedef struct {
  float *distmap;       /* ref algorithms.c:648 */
  uint64_t *edgemap;    /* ref algorithms.c:652 */
  uint32_t *stamp;      /* lazy mode only */
  uint32_t epoch;
  int lazy;
  WalkNode *q; uint64_t qcap;
  U64Vec term;
} WalkScratch;

static const float ORA_DIST_UNSET = (float)INT64_MAX; /* GT_WORD_MAX as float */

static float dm_get(const WalkScratch *s, uint64_t i)
{
  if (s->lazy && s->stamp[i] != s->epoch) return ORA_DIST_UNSET;
  return s->distmap[i];
}
static void dm_set(WalkScratch *s, uint64_t i, float d)
{
  s->distmap[i] = d;
  if (s->lazy) s->stamp[i] = s->epoch;
}

/* ref algorithms.c:620-763.  Returns false for the reference's NULL walk. */
static bool create_walk(OraGraph *g, uint64_t start, WalkScratch *s, Walk *best)
{
  uint64_t qh = 0, qn = 0, k, lengthbest = 0;
  const OraVertex *sv = g->v + start;
  Walk cur = {0};
  best->n = 0; best->total_contig_len = 0;
  if (s->lazy) {
    if (++s->epoch == 0) { memset(s->stamp, 0, sizeof *s->stamp * g->nv);
                           s->epoch = 1; }
  } else
    for (k = 0; k < g->nv; k++) s->distmap[k] = ORA_DIST_UNSET;
  if (sv->nof_edges == 0) return false;
#define QPUSH(E, D) do { \
    if (qn == s->qcap) { s->qcap = s->qcap ? 2 * s->qcap : 256; \
                         s->q = xrealloc(s->q, sizeof *s->q * s->qcap); } \
    s->q[qn].edge = (E); s->q[qn].dist = (D); qn++; } while (0)
  for (k = 0; k < sv->nof_edges; k++) {
    uint64_t e = sv->edges[k];
    if (!edge_is_marked(g, e) && !vertex_is_marked(g, g->e[e].end)) {
      dm_set(s, g->e[e].end, (float)g->e[e].dist);
      s->edgemap[g->e[e].end] = e;
      QPUSH(e, g->e[e].dist);
    }
  }
  s->term.n = 0;
  while (qh < qn) {
    WalkNode node = s->q[qh++];
    const OraEdge *edge = g->e + node.edge;
    uint64_t endv = edge->end;
    bool dir = edge->same ? edge->sense : !edge->sense;
    if (isterminal(g, endv)) vec_push(&s->term, endv);
    for (k = 0; k < g->v[endv].nof_edges; k++) {
      uint64_t ne = g->v[endv].edges[k];
      const OraEdge *nx = g->e + ne;
      if (nx->sense == dir && !edge_is_marked(g, ne) &&
          !vertex_is_marked(g, nx->end) && !is_twin(edge, nx)) {
        float distance = node.dist + nx->dist;   /* int64 sum -> float */
        float old = dm_get(s, nx->end);
        if (old == ORA_DIST_UNSET || old > distance) {
          dm_set(s, nx->end, distance);
          s->edgemap[nx->end] = ne;
          QPUSH(ne, (int64_t)distance);          /* float -> GtWord */
        }
      }
    }
  }
#undef QPUSH
  /* ref algorithms.c:732-756: terminals are popped from the back */
  while (s->term.n != 0) {
    uint64_t cv = s->term.a[--s->term.n];
    cur.n = 0; cur.total_contig_len = 0;
    while (cv != start) {
      uint64_t re = s->edgemap[cv];
      cv = g->e[re].start;
      if (cur.n == cur.cap) { cur.cap = cur.cap ? 2 * cur.cap : 32;
                              cur.edges = xrealloc(cur.edges,
                                                   sizeof *cur.edges * cur.cap); }
      cur.edges[cur.n++] = re;
      cur.total_contig_len += g->v[g->e[re].end].seq_len;
    }
    cur.total_contig_len += sv->seq_len;
    if (cur.total_contig_len > lengthbest) {
      if (best->cap < cur.n) { best->cap = cur.n;
                               best->edges = xrealloc(best->edges,
                                                sizeof *best->edges * best->cap); }
      memcpy(best->edges, cur.edges, sizeof *cur.edges * cur.n);
      best->n = cur.n;
      best->total_contig_len = cur.total_contig_len;
      lengthbest = cur.total_contig_len;
    }
  }
  free(cur.edges);
  return true;
}

void ora_makescaffold(OraGraph *g, int lazy_maps)
{
  U64Vec term = {0}, ccoff = {0};
  WalkScratch s;
  Walk walk = {0}, ccbest = {0};
  uint64_t i, j, k;
  ora_removecycles(g);
  memset(&s, 0, sizeof s);
  s.lazy = lazy_maps;
  s.distmap = xmalloc(sizeof *s.distmap * (g->nv ? g->nv : 1));
  s.edgemap = xmalloc(sizeof *s.edgemap * (g->nv ? g->nv : 1));
  if (lazy_maps) s.stamp = calloc(g->nv ? g->nv : 1, sizeof *s.stamp);
  calc_cc_and_terminals(g, &term, &ccoff);
  for (i = 0; i + 1 < ccoff.n; i++) {
    uint64_t nterm = ccoff.a[i + 1] - ccoff.a[i], max_num_bases = 0;
    bool have_best = false;
    if (nterm == 1) {                       /* ref algorithms.c:790-807 */
      uint64_t v = term.a[ccoff.a[i]];
      bool lonesome = true;
      for (k = 0; k < g->v[v].nof_edges; k++)
        if (!edge_is_marked(g, g->v[v].edges[k])) { lonesome = false; break; }
      if (lonesome) g->v[v].state = ORA_SCAFFOLD;
    }
    if (nterm > 1)
      for (j = ccoff.a[i]; j < ccoff.a[i + 1]; j++) {
        /* ref algorithms.c:823-832: first walk with the strictly largest
           total contig length over the cc */
        if (create_walk(g, term.a[j], &s, &walk) &&
            walk.total_contig_len > max_num_bases) {
          if (ccbest.cap < walk.n) { ccbest.cap = walk.n;
            ccbest.edges = xrealloc(ccbest.edges,
                                    sizeof *ccbest.edges * ccbest.cap); }
          memcpy(ccbest.edges, walk.edges, sizeof *walk.edges * walk.n);
          ccbest.n = walk.n;
          max_num_bases = walk.total_contig_len;
          have_best = true;
        }
      }
    if (have_best && ccbest.n > 0) {         /* ref algorithms.c:835-848 */
      int64_t id;
      g->v[g->e[ccbest.edges[ccbest.n - 1]].start].state = ORA_SCAFFOLD;
      for (id = (int64_t)ccbest.n - 1; id >= 0; id--) {
        OraEdge *e = g->e + ccbest.edges[id];
        const OraVertex *end = g->v + e->end;
        e->state = ORA_SCAFFOLD;
        for (k = 0; k < end->nof_edges; k++)
          if (is_twin(e, g->e + end->edges[k]))
            g->e[end->edges[k]].state = ORA_SCAFFOLD;
        g->v[e->end].state = ORA_SCAFFOLD;
      }
    }
  }
  free(term.a); free(ccoff.a); free(walk.edges); free(ccbest.edges);
  free(s.distmap); free(s.edgemap); free(s.stamp); free(s.q); free(s.term.a);
}

/* ------------------------------------------------------------------ */
/* scaffold records: ref algorithms.c:870-1042                         */

static void rec_add_edge(OraRecord *r, uint64_t e)
{
  if (r->nof_edges == r->cap) { r->cap = r->cap ? 2 * r->cap : 8;
    r->edges = xrealloc(r->edges, sizeof *r->edges * r->cap); }
  r->edges[r->nof_edges++] = e;
}

OraRecords *ora_iterate_scaffolds(OraGraph *g)
{
  OraRecords *recs = xmalloc(sizeof *recs);
  uint64_t v, k, unmarked_edge = 0;
  recs->r = NULL; recs->n = 0; recs->cap = 0;
  for (v = 0; v < g->nv; v++)
    if (!vertex_is_marked(g, v) && g->v[v].state != ORA_SCAFFOLD)
      g->v[v].state = ORA_UNVISITED;
  for (v = 0; v < g->nv; v++) {
    uint64_t nof_scaffold_edges = 0;
    if (g->v[v].state == ORA_VISITED || vertex_is_marked(g, v)) continue;
    for (k = 0; k < g->v[v].nof_edges; k++) {
      uint64_t e = g->v[v].edges[k];
      if (!edge_is_marked(g, e) && g->e[e].state == ORA_SCAFFOLD) {
        nof_scaffold_edges++;
        unmarked_edge = e;
      }
    }
    if (nof_scaffold_edges <= 1) {
      OraRecord *rec;
      if (recs->n == recs->cap) { recs->cap = recs->cap ? 2 * recs->cap : 64;
        recs->r = xrealloc(recs->r, sizeof *recs->r * recs->cap); }
      rec = recs->r + recs->n++;
      rec->root = v; rec->nof_edges = 0; rec->cap = 0; rec->edges = NULL;
      rec->seqlen = g->v[v].seq_len;
      g->v[v].state = ORA_VISITED;
      if (nof_scaffold_edges == 1) {
        uint64_t next_edge = unmarked_edge;
        while (1) {
          const OraEdge *ne = g->e + next_edge;
          uint64_t nend = ne->end, in_dir = 0;
          bool dir;
          rec_add_edge(rec, next_edge);
          rec->seqlen += (uint64_t)ne->dist;
          rec->seqlen += g->v[nend].seq_len;
          if (g->v[nend].state == ORA_VISITED) break;
          g->v[nend].state = ORA_VISITED;
          dir = ne->same ? ne->sense : !ne->sense;
          for (k = 0; k < g->v[nend].nof_edges; k++) {
            uint64_t e = g->v[nend].edges[k];
            if (g->e[e].sense == dir && !edge_is_marked(g, e) &&
                !is_twin(ne, g->e + e) && g->e[e].state == ORA_SCAFFOLD) {
              in_dir++;
              unmarked_edge = e;
            }
          }
          if (in_dir == 1) next_edge = unmarked_edge; else break;
        }
      }
    }
  }
  return recs;
}

void ora_records_delete(OraRecords *r)
{
  uint64_t i;
  if (!r) return;
  for (i = 0; i < r->n; i++) free(r->r[i].edges);
  free(r->r); free(r);
}

int ora_write_scaffold(const OraGraph *g, const OraRecords *r, const char *fn)
{
  FILE *f = fopen(fn, "w");
  uint64_t i, j;
  if (!f) return -1;
  for (i = 0; i < r->n; i++) {
    const OraRecord *rec = r->r + i;
    fprintf(f, "%s", g->v[rec->root].header);
    for (j = 0; j < rec->nof_edges; j++) {
      const OraEdge *e = g->e + rec->edges[j];
      fprintf(f, "\t%s,%ld,%f,%d,%d,", g->v[e->end].header, (long)e->dist,
              e->std_dev, e->sense, e->same);
    }
    fprintf(f, "\n");
  }
  fclose(f);
  return 0;
}

/* ------------------------------------------------------------------ */
/* flat accessors                                                      */

uint64_t ora_nv(const OraGraph *g) { return g->nv; }
uint64_t ora_ne(const OraGraph *g) { return g->ne; }
void ora_get_vertex_states(const OraGraph *g, uint8_t *out)
{ uint64_t i; for (i = 0; i < g->nv; i++) out[i] = g->v[i].state; }
void ora_get_edge_states(const OraGraph *g, uint8_t *out)
{ uint64_t i; for (i = 0; i < g->ne; i++) out[i] = g->e[i].state; }
void ora_get_edges(const OraGraph *g, uint32_t *start, uint32_t *end,
                   int64_t *dist, float *std_dev, uint64_t *num_pairs,
                   uint8_t *flags)
{
  uint64_t i;
  for (i = 0; i < g->ne; i++) {
    const OraEdge *e = g->e + i;
    start[i] = (uint32_t)e->start; end[i] = (uint32_t)e->end;
    dist[i] = e->dist; std_dev[i] = e->std_dev; num_pairs[i] = e->num_pairs;
    flags[i] = (uint8_t)((e->sense ? 1 : 0) | (e->same ? 2 : 0));
  }
}
void ora_get_vertices(const OraGraph *g, uint64_t *seq_len, float *astat,
                      float *copy_num)
{
  uint64_t i;
  for (i = 0; i < g->nv; i++) {
    seq_len[i] = g->v[i].seq_len; astat[i] = g->v[i].astat;
    copy_num[i] = g->v[i].copy_num;
  }
}
const char *ora_vertex_header(const OraGraph *g, uint64_t i)
{ return g->v[i].header; }
void ora_set_vertex_attrs(OraGraph *g, const float *astat, const float *copy_num)
{
  uint64_t i;
  for (i = 0; i < g->nv; i++) { g->v[i].astat = astat[i];
                                g->v[i].copy_num = copy_num[i]; }
}
uint64_t ora_records_n(const OraRecords *r) { return r->n; }
uint64_t ora_records_total_edges(const OraRecords *r)
{ uint64_t i, t = 0; for (i = 0; i < r->n; i++) t += r->r[i].nof_edges; return t; }
void ora_records_flatten(const OraRecords *r, uint64_t *roots,
                         uint64_t *offsets, uint64_t *edges, uint64_t *seqlen)
{
  uint64_t i, t = 0;
  for (i = 0; i < r->n; i++) {
    roots[i] = r->r[i].root; offsets[i] = t; seqlen[i] = r->r[i].seqlen;
    memcpy(edges + t, r->r[i].edges, sizeof *edges * r->r[i].nof_edges);
    t += r->r[i].nof_edges;
  }
  offsets[r->n] = t;
}
