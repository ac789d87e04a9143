/*
  gt_scaffolder_host.c -- C host layer: the reference's public API
  (include/gt_scaffolder_host.h) on top of the GPU engine
  (include/gt_scaffold_hip.h).

  Host work: parsing contigs (.fa), distance estimates (.de) and A-statistics
  (.astat) into flat record arrays, writing .dot / .scaf, and the scaffold
  record walk.  Everything between -- edge construction, repeat marking,
  filtering, cycle removal, scaffold construction -- is a call into the engine.
  There is no host implementation of those steps.
*/
#define _GNU_SOURCE
#include "gt_scaffolder_host.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "gt_scaffold_hip.h"

#define LINE_MAX_REF 1023 /* the reference reads lines with fgets(line, 1024) */

/* GT_SCAFFOLDER_TIMING=1: the steps of the output functions on stderr */
static double now_s(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static void lap(const char *what, double *t0)
{
  static int on = -1;
  double t;
  if (on < 0) { const char *e = getenv("GT_SCAFFOLDER_TIMING"); on = e && *e && *e != '0'; }
  if (!on) return;
  t = now_s();
  fprintf(stderr, "[gt_scaffolder] %-28s %8.1f ms\n", what, 1e3 * (t - *t0));
  *t0 = t;
}

static int g_device = 0;
void gt_scaffolder_set_device(int device) { g_device = device; }

typedef struct {
  char *name;
  uint64_t seq_len;
  float astat, copy_num;
} Contig;

typedef struct {   /* hand-built or downloaded edge, id order */
  uint32_t start, end;
  int64_t dist;
  float std_dev;
  int64_t num_pairs;
  uint8_t flags;
} HEdge;

struct GtScaffolderGraph {
  Contig *ctg;
  uint64_t nof_vertices, max_nof_vertices;
  HEdge *edges;              /* host copy (hand-built, or cached download) */
  uint64_t nof_edges, max_nof_edges;
  bool edges_cached;
  uint8_t *vstate, *estate;  /* host copies, refreshed from the engine */
  GtsgEngine *eng;           /* NULL for hand-built graphs */
  GtsgDeParser *dp;          /* GPU parser of the distance file (holds the name table) */
  bool dp_names;             /* its name table is the sorted headers */
  bool dp_parsed;            /* it holds the records of the distance file below (count_distances), */
  dev_t dp_dev; ino_t dp_ino; off_t dp_size; struct timespec dp_mtime;   /* ... as that file was then */
  GtsgDeParseResult dp_res;
  bool dup_names;            /* two contigs share a header: the GPU name table would pick either */
  bool sorted;               /* contigs in header order (ids are final) */
  char err[512];
};

typedef struct {             /* an edge of a scaffold record: what the .scaf line prints of it */
  uint32_t eid, end;
  int64_t dist;
  float std_dev;
  uint8_t flags;
} REdge;

struct GtScaffolderGraphRecords {
  const GtScaffolderGraph *g;
  uint64_t n, cap;
  uint64_t *root;
  uint64_t *off;             /* n+1 offsets into edge */
  REdge *edge;
  uint64_t nedge, capedge;
};

static int seterr(char *err, size_t n, const char *fmt, ...)
{
  if (err && n) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, n, fmt, ap);
    va_end(ap);
  }
  return -1;
}

static void *xcalloc(size_t n, size_t sz)
{
  void *p = calloc(n ? n : 1, sz);
  if (!p) { fprintf(stderr, "gt_scaffolder: out of memory\n"); abort(); }
  return p;
}
static void *xrealloc(void *q, size_t sz)
{
  void *p = realloc(q, sz ? sz : 1);
  if (!p) { fprintf(stderr, "gt_scaffolder: out of memory\n"); abort(); }
  return p;
}

/* A large transfer between a file and memory in pieces side by side: one
   thread copies to or from the page cache at 2 - 3 GB/s (and takes the page
   faults of a fresh buffer one by one); the 3.9 GB contig file of the 3 M-contig
   example was 1 s of every pass over it. */
#define IO_PIECE_MIN ((size_t)32 << 20)
#define IO_THREADS_MAX 8
typedef struct { int fd, wr, bad; char *buf; size_t len; off_t off; } IoJob;
static void *io_worker(void *p)
{
  IoJob *j = p;
  size_t done = 0;
  while (done < j->len) {
    ssize_t k = j->wr ? pwrite(j->fd, j->buf + done, j->len - done, j->off + (off_t)done)
                      : pread(j->fd, j->buf + done, j->len - done, j->off + (off_t)done);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) { j->bad = 1; break; }
    done += (size_t)k;
  }
  return NULL;
}
static int io_parallel(int fd, char *buf, size_t len, off_t off, int wr)
{
  IoJob job[IO_THREADS_MAX];
  pthread_t th[IO_THREADS_MAX];
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  size_t n = len / IO_PIECE_MIN, i, piece;
  int bad = 0, started[IO_THREADS_MAX];
  if (n > IO_THREADS_MAX) n = IO_THREADS_MAX;
  if (ncpu > 0 && n > (size_t)ncpu) n = (size_t)ncpu;
  if (n < 1) n = 1;
  piece = (len / n + 4095) & ~(size_t)4095;
  for (i = 0; i < n; i++) {
    size_t b = i * piece, e = i + 1 == n ? len : (i + 1) * piece;
    if (b > len) b = len;
    if (e > len) e = len;
    job[i].fd = fd; job[i].wr = wr; job[i].bad = 0; job[i].buf = buf + b; job[i].len = e - b;
    job[i].off = off + (off_t)b;
    started[i] = i > 0 && pthread_create(&th[i], NULL, io_worker, &job[i]) == 0;
  }
  io_worker(&job[0]);
  for (i = 1; i < n; i++) {
    if (started[i]) pthread_join(th[i], NULL);
    else io_worker(&job[i]);     /* no thread to be had: here */
  }
  for (i = 0; i < n; i++) bad |= job[i].bad;
  return bad;
}

/* free() of a multi-gigabyte buffer is a quarter of a second of page-table work
   (256 ms for the 3.9 GB contig file): a thread of its own does it */
static void *free_worker(void *p) { free(p); return NULL; }
static void free_large(void *p, size_t bytes)
{
  pthread_t th;
  pthread_attr_t at;
  if (p && bytes >= ((size_t)256 << 20) && pthread_attr_init(&at) == 0) {
    int ok = pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED) == 0 &&
             pthread_create(&th, &at, free_worker, p) == 0;
    pthread_attr_destroy(&at);
    if (ok) return;
  }
  free(p);
}

/* whole file into memory, NUL-terminated */
static char *slurp(const char *path, size_t *len)
{
  FILE *f;
  char *buf;
  long sz;
  {
    /* a regular file: sized by fstat, read in parallel pieces */
    struct stat st;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return NULL;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 0) {
      buf = malloc((size_t)st.st_size + 1);
      if (!buf || io_parallel(fd, buf, (size_t)st.st_size, 0, 0)) { free(buf); close(fd); return NULL; }
      buf[st.st_size] = '\0';
      close(fd);
      *len = (size_t)st.st_size;
      return buf;
    }
    close(fd);
  }
  f = fopen(path, "rb");
  if (!f) return NULL;
  if (fseek(f, 0, SEEK_END) != 0 || (sz = ftell(f)) < 0) { fclose(f); return NULL; }
  rewind(f);
  buf = malloc((size_t)sz + 1);
  if (!buf || fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); fclose(f); return NULL; }
  buf[sz] = '\0';
  fclose(f);
  *len = (size_t)sz;
  return buf;
}

/* ------------------------------------------------------------------ */
GtScaffolderGraph *gt_scaffolder_graph_new(uint64_t max_v, uint64_t max_e)
{
  GtScaffolderGraph *g = xcalloc(1, sizeof *g);
  g->max_nof_vertices = max_v;
  g->max_nof_edges = max_e;
  g->ctg = xcalloc(max_v, sizeof *g->ctg);
  g->edges = xcalloc(max_e, sizeof *g->edges);
  g->vstate = xcalloc(max_v, 1);
  g->estate = xcalloc(max_e, 1);
  return g;
}

static void fascan_drop(void);
void gt_scaffolder_graph_delete(GtScaffolderGraph *g)
{
  uint64_t i;
  if (!g) return;
  fascan_drop();   /* a scan count_contigs kept for a read_contigs that never came (up to the file's size) */
  for (i = 0; i < g->nof_vertices; i++) free(g->ctg[i].name);
  free(g->ctg); free(g->edges); free(g->vstate); free(g->estate);
  if (g->eng) gtsg_destroy(g->eng);
  if (g->dp) gtsg_deparser_destroy(g->dp);
  free(g);
}

int gt_scaffolder_graph_add_vertex(GtScaffolderGraph *g, const char *header,
                                   uint64_t seq_len, float astat, float copy_num)
{
  Contig *c;
  if (!g || g->nof_vertices >= g->max_nof_vertices) return -1;
  c = g->ctg + g->nof_vertices;
  c->name = strdup(header ? header : "");
  c->seq_len = seq_len; c->astat = astat; c->copy_num = copy_num;
  g->vstate[g->nof_vertices++] = 0;
  g->sorted = false;       /* ids are final only after the next sort (parser.c:172) */
  g->dp_names = false;
  return 0;
}

int gt_scaffolder_graph_add_edge(GtScaffolderGraph *g, uint64_t vstart,
                                 uint64_t vend, int64_t dist, float std_dev,
                                 uint64_t num_pairs, bool dir, bool same)
{
  HEdge *e;
  if (!g || g->eng || g->nof_edges >= g->max_nof_edges ||
      vstart >= g->nof_vertices || vend >= g->nof_vertices)
    return -1;
  e = g->edges + g->nof_edges;
  e->start = (uint32_t)vstart; e->end = (uint32_t)vend; e->dist = dist;
  e->std_dev = std_dev; e->num_pairs = (int64_t)num_pairs;
  e->flags = (uint8_t)((dir ? 1 : 0) | (same ? 2 : 0));
  g->estate[g->nof_edges++] = 0;
  g->edges_cached = true;
  return 0;
}

/* ref gt_scaffolder_graph.c:174-193: the first edge of vertex_1's list that
   ends in vertex_2 (an edge id; GT_SCAFFOLDER_NO_EDGE where the reference
   returns NULL) */
uint64_t gt_scaffolder_graph_find_edge(GtScaffolderGraph *g, uint64_t vertex_1, uint64_t vertex_2)
{
  uint64_t i, eid = GT_SCAFFOLDER_NO_EDGE;
  if (!g || vertex_1 >= g->nof_vertices || vertex_2 >= g->nof_vertices) return GT_SCAFFOLDER_NO_EDGE;
  if (g->eng) {
    if (gtsg_find_edge(g->eng, vertex_1, vertex_2, &eid) != 0) {
      snprintf(g->err, sizeof g->err, "%s", gtsg_last_error(g->eng));
      return GT_SCAFFOLDER_NO_EDGE;
    }
    return eid;
  }
  /* hand-built graph: a vertex' list is its edges in creation order */
  for (i = 0; i < g->nof_edges; i++)
    if (g->edges[i].start == vertex_1 && g->edges[i].end == vertex_2) return i;
  return GT_SCAFFOLDER_NO_EDGE;
}

/* ref gt_scaffolder_graph.c:237-244: a vertex IS its id here (the reference
   subtracts the base of the vertex array); out of range: GT_SCAFFOLDER_NO_VERTEX */
uint64_t gt_scaffolder_graph_get_vertex_id(const GtScaffolderGraph *g, uint64_t vertex)
{
  return g && vertex < g->nof_vertices ? vertex : GT_SCAFFOLDER_NO_VERTEX;
}

static bool find_contig64(const GtScaffolderGraph *g, const char *name, uint64_t *id);
/* ref gt_scaffolder_graph.c:196-216: binary search over the vertices, which
   are in header order once the distance file has been counted or read
   (parser.c:172); like the reference it assumes that order */
bool gt_scaffolder_graph_get_vertex(const GtScaffolderGraph *g, uint64_t *vertex,
                                    const char *header_seq)
{
  if (!g || !vertex || !header_seq) return false;
  return find_contig64(g, header_seq, vertex);
}

/* ref gt_scaffolder_graph.c:219-235 */
int gt_scaffolder_graph_alter_edge(GtScaffolderGraph *g, uint64_t edge, int64_t dist,
                                   float std_dev, uint64_t num_pairs, bool sense, bool same)
{
  if (!g || edge >= g->nof_edges) return -1;
  if (g->eng) {
    int rc = gtsg_alter_edge(g->eng, edge, dist, std_dev, num_pairs, sense, same);
    if (rc) { snprintf(g->err, sizeof g->err, "%s", gtsg_last_error(g->eng)); return -1; }
    if (!g->edges_cached) return 0;   /* the next download sees the new values */
  }
  g->edges[edge].dist = dist; g->edges[edge].std_dev = std_dev;
  g->edges[edge].num_pairs = (int64_t)num_pairs;
  g->edges[edge].flags = (uint8_t)((sense ? 1 : 0) | (same ? 2 : 0));
  return 0;
}

uint64_t gt_scaffolder_graph_nof_vertices(const GtScaffolderGraph *g) { return g ? g->nof_vertices : 0; }
uint64_t gt_scaffolder_graph_nof_edges(const GtScaffolderGraph *g) { return g ? g->nof_edges : 0; }
const char *gt_scaffolder_graph_last_error(const GtScaffolderGraph *g) { return g ? g->err : ""; }

/* ------------------------------------------------------------------ */
/* contigs                                                             */

static int contig_cmp(const void *a, const void *b)
{
  return strcmp(((const Contig *)a)->name, ((const Contig *)b)->name);
}

/* the reference's probing sequence (gt_scaffolder_graph.c:196-216: inclusive
   bounds, mid = min + (max - min) / 2): with equal headers it decides which of
   them is found -- sorted [A, A] gives the first there */
static bool find_contig64(const GtScaffolderGraph *g, const char *name, uint64_t *id)
{
  int64_t lo = 0, hi = (int64_t)g->nof_vertices - 1;
  while (hi >= lo) {
    const int64_t mid = lo + (hi - lo) / 2;
    const int c = strcmp(g->ctg[mid].name, name);
    if (c == 0) { *id = (uint64_t)mid; return true; }
    if (c < 0) lo = mid + 1; else hi = mid - 1;
  }
  return false;
}
static bool find_contig(const GtScaffolderGraph *g, const char *name, uint32_t *id)
{
  uint64_t v;
  if (!find_contig64(g, name, &v)) return false;
  *id = (uint32_t)v;
  return true;
}

/* the records of a FASTA text: description [ds, de) (after the '>', up to the
   newline) and the number of sequence characters (blanks and line ends do not
   count), in file order */
typedef struct { uint64_t n, *ds, *de, *sl; } FaTable;
static void fa_free(FaTable *t) { free(t->ds); free(t->de); free(t->sl); memset(t, 0, sizeof *t); }

static void fasta_table_host(const char *buf, size_t len, FaTable *t)
{
  size_t i = 0;
  uint64_t cap = 0;
  memset(t, 0, sizeof *t);
  while (i < len && buf[i] == '>') {
    size_t ds = ++i, de;
    uint64_t slen = 0;
    while (i < len && buf[i] != '\n') i++;
    de = i;
    if (i < len) i++;
    while (i < len && buf[i] != '>') {
      char c = buf[i++];
      if (c != '\n' && c != '\r' && c != ' ') slen++;
    }
    if (t->n == cap) {
      cap = cap ? 2 * cap : 1024;
      t->ds = xrealloc(t->ds, cap * sizeof *t->ds);
      t->de = xrealloc(t->de, cap * sizeof *t->de);
      t->sl = xrealloc(t->sl, cap * sizeof *t->sl);
    }
    t->ds[t->n] = ds; t->de[t->n] = de; t->sl[t->n] = slen;
    t->n++;
  }
}

static int g_host_parser;   /* set by gt_scaffolder_set_distance_parser, below */

/* the same table from the GPU (gtsg_fasta_records) for large files */
#define GPU_FASTA_MIN (32u << 20)
static void fasta_table(const char *buf, size_t len, FaTable *t)
{
  if (len >= GPU_FASTA_MIN && g_host_parser != 1) {
    memset(t, 0, sizeof *t);
    if (gtsg_fasta_records(g_device, buf, len, &t->n, &t->ds, &t->de, &t->sl) == 0) return;
    memset(t, 0, sizeof *t);   /* no GPU, or the file is above its limits: the host loop */
  }
  fasta_table_host(buf, len, t);
}

/* What the parsers need of a contig file: per contig its description (the
   header line without '>' and line end, NUL-terminated in one blob) and the
   number of sequence characters.  The reference's API reads the file twice --
   gt_scaffolder_parser_count_contigs sizes the graph, ..._read_contigs fills it
   -- and at 3 M contigs a pass over the 3.9 GB file is 1.2 s; the scan of the
   counting pass is kept (180 MB at 3 M contigs) and handed to the reading pass
   if the file is still the same (device, inode, size, modification time).  */
typedef struct {
  uint64_t n, *sl, *doff;
  char *blob;
  dev_t dev; ino_t ino; off_t size; struct timespec mtime;
} FaScan;
static void fascan_free(FaScan *sc)
{
  if (!sc) return;
  free(sc->sl); free(sc->doff); free(sc->blob); free(sc);
}
static FaScan *g_fa_kept;
static pthread_mutex_t g_fa_lock = PTHREAD_MUTEX_INITIALIZER;

static bool fa_same_file(const FaScan *sc, const struct stat *st)
{
  return sc->dev == st->st_dev && sc->ino == st->st_ino && sc->size == st->st_size &&
         sc->mtime.tv_sec == st->st_mtim.tv_sec && sc->mtime.tv_nsec == st->st_mtim.tv_nsec;
}

/* the scan of `path`: the kept one if it is of this file as it is now, else a
   fresh one.  NULL + message on error.  The caller owns the result. */
static FaScan *fascan_get(const char *path, char *err, size_t errlen)
{
  struct stat st;
  FaScan *sc = NULL;
  FaTable t;
  size_t len;
  char *buf;
  uint64_t r, total = 0;
  const bool have_stat = stat(path, &st) == 0;
  pthread_mutex_lock(&g_fa_lock);
  if (g_fa_kept && have_stat && fa_same_file(g_fa_kept, &st)) { sc = g_fa_kept; g_fa_kept = NULL; }
  pthread_mutex_unlock(&g_fa_lock);
  if (sc) return sc;
  double t0 = now_s();
  buf = slurp(path, &len);
  lap("contigs: file read", &t0);
  if (!buf) { seterr(err, errlen, "cannot open file %s", path); return NULL; }
  if (len == 0) { free(buf); seterr(err, errlen, "sequence file %s is empty", path); return NULL; }
  if (buf[0] != '>') {
    free(buf);
    seterr(err, errlen, "the first character of fasta file %s has to be '>'", path);
    return NULL;
  }
  fasta_table(buf, len, &t);
  lap("contigs: record table", &t0);
  sc = xcalloc(1, sizeof *sc);
  sc->n = t.n;
  sc->sl = t.sl; t.sl = NULL;
  sc->doff = xcalloc(t.n + 1, sizeof *sc->doff);
  for (r = 0; r < t.n; r++) {
    size_t ds = t.ds[r], de = t.de[r];
    if (de > ds && buf[de - 1] == '\r') de--;
    t.de[r] = de;
    sc->doff[r] = total;
    total += de - ds + 1;
  }
  sc->doff[t.n] = total;
  sc->blob = xcalloc(total + 1, 1);
  for (r = 0; r < t.n; r++) memcpy(sc->blob + sc->doff[r], buf + t.ds[r], t.de[r] - t.ds[r]);
  lap("contigs: descriptions", &t0);
  free_large(buf, len); fa_free(&t);
  lap("contigs: buffers freed", &t0);
  if (have_stat && stat(path, &st) == 0) {
    sc->dev = st.st_dev; sc->ino = st.st_ino; sc->size = st.st_size; sc->mtime = st.st_mtim;
  } else
    sc->size = -1;      /* never matches */
  return sc;
}
/* keeps sc for a later pass over the same file (replaces what was kept) */
static void fascan_keep(FaScan *sc)
{
  FaScan *old;
  pthread_mutex_lock(&g_fa_lock);
  old = g_fa_kept; g_fa_kept = sc;
  pthread_mutex_unlock(&g_fa_lock);
  fascan_free(old);
}

static void fascan_drop(void) { fascan_keep(NULL); }

/* FASTA: '>' description newline, then sequence characters up to the next
   '>' (blanks and line ends do not count).  With a graph: keeps contigs longer
   than min_ctg_len (ref parser.c:481), header cut at the first blank
   (parser.c:452-458), optional "length= depth= k= astat=" annotation
   (parser.c:438-450).  Without (g == NULL): counts the contigs of at least
   min_ctg_len into *count (ref parser.c:399-415, which tests >=). */
static int scan_contigs(GtScaffolderGraph *g, const char *path, uint64_t min_len,
                        bool annotated, uint64_t *count, char *err, size_t errlen)
{
  uint64_t cap = g ? g->max_nof_vertices : 0, r;
  FaScan *sc = fascan_get(path, err, errlen);
  if (!sc) return -1;
  for (r = 0; r < sc->n; r++) {
    char *desc = sc->blob + sc->doff[r], *sp;
    const uint64_t dlen = sc->doff[r + 1] - sc->doff[r] - 1;
    const uint64_t slen = sc->sl[r];
    float astat = 0.0f, copynum = 0.0f;
    if (g && annotated) {
      char part[1024];
      long n1, n2;
      if (sscanf(desc, "%1023s length=%ld depth=%ld k=%f astat=%f", part, &n1, &n2,
                 &copynum, &astat) != 5) {
        fascan_free(sc);
        return seterr(err, errlen, "No A-statistic/copy number was found in header");
      }
    }
    if (g && dlen == 0) { fascan_free(sc); return seterr(err, errlen, "Invalid header length"); }
    if (slen == 0) { fascan_free(sc); return seterr(err, errlen, "Invalid sequence length"); }
    sp = strchr(desc, ' ');
    if (sp) *sp = '\0';
    if (!g) {
      if (slen >= min_len) ++*count;
    } else if (slen > min_len) {
      if (g->nof_vertices == cap) {
        cap = cap ? 2 * cap : 1024;
        g->ctg = xrealloc(g->ctg, cap * sizeof *g->ctg);
        g->vstate = xrealloc(g->vstate, cap);
      }
      g->ctg[g->nof_vertices].name = strdup(desc);
      g->ctg[g->nof_vertices].seq_len = slen;
      g->ctg[g->nof_vertices].astat = astat;
      g->ctg[g->nof_vertices].copy_num = copynum;
      g->vstate[g->nof_vertices] = 0;
      g->nof_vertices++;
    }
    if (sp) *sp = ' ';
  }
  if (g) {
    fascan_free(sc);
    g->max_nof_vertices = cap;
    g->sorted = false;
    g->dp_names = false;   /* the GPU parser's name table is that of the old vertex set */
    g->dp_parsed = false;
  } else if ((sc->doff[sc->n] + 16 * sc->n) >> 30 == 0)
    fascan_keep(sc);          /* (up to 1 GB: 15 M contigs with 50-byte headers) */
  else
    fascan_free(sc);
  return 0;
}

/* vertex ids = rank of the header, ref parser.c:172.  Many headers: the first
   14 bytes are sorted on the GPU (gtsg_sort_names, two radix sorts), runs that
   agree in them with qsort / strcmp here. */
#define GPU_SORT_MIN 50000u
static void sort_contigs_inner(GtScaffolderGraph *g);
static void sort_contigs(GtScaffolderGraph *g)
{
  uint64_t i;
  if (g->sorted) return;
  sort_contigs_inner(g);
  g->dup_names = false;
  for (i = 1; i < g->nof_vertices && !g->dup_names; i++)
    g->dup_names = strcmp(g->ctg[i - 1].name, g->ctg[i].name) == 0;
}
static void sort_contigs_inner(GtScaffolderGraph *g)
{
  uint64_t n = g->nof_vertices;
  g->sorted = true;
  if (n >= GPU_SORT_MIN && g_host_parser != 1) {
    uint64_t i, total = 0, *off = xcalloc(n + 1, sizeof *off);
    uint32_t *perm = xcalloc(n, sizeof *perm);
    uint8_t *tie = xcalloc(n, 1);
    char *blob;
    for (i = 0; i < n; i++) { off[i] = total; total += strlen(g->ctg[i].name); }
    off[n] = total;
    blob = xcalloc(total + 1, 1);
    for (i = 0; i < n; i++) memcpy(blob + off[i], g->ctg[i].name, off[i + 1] - off[i]);
    if (gtsg_sort_names(g_device, blob, off, n, perm, tie) == 0) {
      Contig *sorted = xcalloc(n, sizeof *sorted);
      for (i = 0; i < n; i++) sorted[i] = g->ctg[perm[i]];
      for (i = 0; i < n;) {
        uint64_t j = i + 1;
        while (j < n && tie[j]) j++;
        if (j - i > 1) qsort(sorted + i, j - i, sizeof *sorted, contig_cmp);
        i = j;
      }
      memcpy(g->ctg, sorted, n * sizeof *sorted);
      free(sorted); free(blob); free(off); free(perm); free(tie);
      return;
    }
    free(blob); free(off); free(perm); free(tie);   /* no GPU: the host sort */
  }
  qsort(g->ctg, n, sizeof *g->ctg, contig_cmp);
}

/* ref gt_scaffolder_parser.c:495 */
int gt_scaffolder_parser_count_contigs(const char *filename, uint64_t min_ctg_len,
                                       uint64_t *nof_contigs, char *err, size_t errlen)
{
  uint64_t n = 0;
  int rc;
  if (!filename || !nof_contigs) return seterr(err, errlen, "invalid argument");
  rc = scan_contigs(NULL, filename, min_ctg_len, false, &n, err, errlen);
  *nof_contigs = n;
  return rc;
}

/* ref gt_scaffolder_parser.c:524 */
int gt_scaffolder_parser_read_contigs(GtScaffolderGraph *graph, const char *filename,
                                      uint64_t min_ctg_len, bool astat_is_annotated,
                                      char *err, size_t errlen)
{
  if (!graph || !filename) return seterr(err, errlen, "invalid argument");
  if (graph->eng) return seterr(err, errlen, "the graph has been built already");
  return scan_contigs(graph, filename, min_ctg_len, astat_is_annotated, NULL, err, errlen);
}

/* ------------------------------------------------------------------ */
/* distance estimates                                                  */

typedef struct {
  uint32_t *root, *ctg;
  int64_t *dist, *np;
  float *sd;
  uint8_t *flags;
  uint64_t n, cap;
} Records;

static void rec_push(Records *r, uint32_t root, uint32_t ctg, int64_t dist,
                     int64_t np, float sd, uint8_t flags)
{
  if (r->n == r->cap) {
    r->cap = r->cap ? 2 * r->cap : 4096;
    r->root = xrealloc(r->root, r->cap * sizeof *r->root);
    r->ctg = xrealloc(r->ctg, r->cap * sizeof *r->ctg);
    r->dist = xrealloc(r->dist, r->cap * sizeof *r->dist);
    r->np = xrealloc(r->np, r->cap * sizeof *r->np);
    r->sd = xrealloc(r->sd, r->cap * sizeof *r->sd);
    r->flags = xrealloc(r->flags, r->cap);
  }
  r->root[r->n] = root; r->ctg[r->n] = ctg; r->dist[r->n] = dist;
  r->np[r->n] = np; r->sd[r->n] = sd; r->flags[r->n] = flags;
  r->n++;
}
static void rec_free(Records *r)
{
  free(r->root); free(r->ctg); free(r->dist); free(r->np); free(r->sd); free(r->flags);
}

/* one "header{+,-},dist,pairs,std" record as the reference scans it with
   "%[^>,],%ld,%ld,%f" (ref parser.c:212, :340) */
static bool scan_record(const char *field, char *hdr, long *dist, long *np, float *sd)
{
  return sscanf(field, "%1023[^>,],%ld,%ld,%f", hdr, dist, np, sd) == 4;
}

/* Both passes of the reference over the .de file: the integrity check of
   parser.c:150-291 (its error messages and conditions) and the record loop of
   parser.c:295-394, which drops the last character of every line and flips
   the direction at ';'.  Lines the reference's 1024-byte fgets buffer would
   split are rejected instead of being mis-parsed. */
static int read_distance_records(const GtScaffolderGraph *g, const char *path,
                                 int pass, uint64_t *nof_valid, Records *out,
                                 char *err, size_t errlen)
{
  size_t len, pos;
  char *buf = slurp(path, &len), *line = NULL, hdr[1024];
  size_t linecap = 0;
  uint64_t valid_records = 0;
  if (!buf) return seterr(err, errlen, "can not read distance file %s", path);
  {
    pos = 0;
    while (pos < len) {
      size_t ls = pos, ll;
      char *save = NULL, *field;
      uint32_t root = 0, ctg = 0;
      bool valid, sense = true;
      long dist, np;
      float sd;
      while (pos < len && buf[pos] != '\n') pos++;
      if (pos < len) pos++;               /* the line keeps its newline, as fgets */
      ll = pos - ls;
      if (ll > LINE_MAX_REF) {
        free(buf); free(line);
        return seterr(err, errlen, "line longer than %d characters in dist file %s "
                      "(the reference's line buffer would split it)", LINE_MAX_REF, path);
      }
      if (ll + 1 > linecap) { linecap = ll + 64; line = xrealloc(line, linecap); }
      memcpy(line, buf + ls, ll);
      line[ll] = '\0';
      if (pass == 1) line[ll - 1] = '\0';  /* parser.c:325 */
      field = strtok_r(line, " ", &save);
      valid = field && find_contig(g, field, &root);
      if (pass == 0) {
        field = strtok_r(NULL, " ", &save);
        if (!field) {
          free(buf); free(line);
          return seterr(err, errlen, "Invalid record in dist file %s", path);
        }
        if (!valid) continue;
        for (; field; field = strtok_r(NULL, " ", &save)) {
          if (scan_record(field, hdr, &dist, &np, &sd)) {
            size_t hl = strlen(hdr);
            if (np < 0) {
              free(buf); free(line);
              return seterr(err, errlen, "Invalid value for number of pairs in dist file %s", path);
            }
            if (hdr[hl - 1] != '+' && hdr[hl - 1] != '-') {
              free(buf); free(line);
              return seterr(err, errlen, "Invalid composition sign in dist file %s", path);
            }
            hdr[hl - 1] = '\0';
            if (find_contig(g, hdr, &ctg)) valid_records++;
          } else if (*field != ';') {
            free(buf); free(line);
            return seterr(err, errlen, "Invalid record in dist file %s", path);
          }
        }
      } else {
        if (!valid) continue;
        for (; field; field = strtok_r(NULL, " ", &save)) {
          if (scan_record(field, hdr, &dist, &np, &sd)) {
            size_t hl = strlen(hdr);
            bool same = hdr[hl - 1] == '+';
            hdr[hl - 1] = '\0';
            if (find_contig(g, hdr, &ctg))
              rec_push(out, root, ctg, dist, np, sd,
                       (uint8_t)((sense ? 1 : 0) | (same ? 2 : 0)));
          } else if (*field == ';')
            sense = !sense;
        }
      }
    }
    if (pass == 0 && valid_records == 0) {
      free(buf); free(line);
      return seterr(err, errlen, "distance file %s is empty", path);
    }
  }
  free(buf); free(line);
  if (nof_valid) *nof_valid = valid_records;
  return 0;
}

/* Which parser reads distance files: 0 the GPU parser with the host code as
   the fallback for files outside its regular form (default), 1 the host code
   only, 2 the GPU parser or an error (tests). */
void gt_scaffolder_set_distance_parser(int mode) { g_host_parser = mode; }
/* who formats the edge lines of gt_scaffolder_graph_print: 0 (default) the GPU
   for a graph that lives there, 1 the host (the reference's loop by hand) */
static int g_host_dot;
void gt_scaffolder_set_dot_writer(int mode) { g_host_dot = mode; }
/* who walks the scaffold records of a graph on the GPU: 0 (default) the GPU ranks
   the clean SCAFFOLD paths (gtsg_scaffold_records) and the host walks the open
   part; 1 the host walks everything (the checker).  gt_scaffolder_last_record_walk:
   which of the two the last call was. */
static int g_host_records, g_last_record_walk;
void gt_scaffolder_set_record_walk(int mode) { g_host_records = mode; }
int gt_scaffolder_last_record_walk(void) { return g_last_record_walk; }

/* The GPU parser of the graph with the sorted headers as its name table.
   *have = 0: there is none (host-only mode, no GPU, too many names). */
static int ensure_parser(GtScaffolderGraph *g, int *have, char *err, size_t errlen)
{
  int rc;
  *have = 0;
  if (g_host_parser == 1) return 0;
  sort_contigs(g);
  if (g->dup_names) {
    /* which of two equal headers a look-up returns is fixed for the host's
       binary search and a matter of insertion order for the GPU table */
    if (g_host_parser == 2) return seterr(err, errlen, "duplicate contig headers: the GPU distance parser does not take them");
    return 0;
  }
  if (!g->dp) {
    if (gtsg_deparser_create(&g->dp, g_device, NULL) != 0) {
      g->dp = NULL;
      if (g_host_parser == 2) return seterr(err, errlen, "no MI355X available for the distance parser");
      return 0;
    }
    g->dp_names = false;
  }
  if (!g->dp_names) {
    uint64_t i, total = 0, *off = xcalloc(g->nof_vertices + 1, sizeof *off);
    char *blob;
    for (i = 0; i < g->nof_vertices; i++) { off[i] = total; total += strlen(g->ctg[i].name); }
    off[g->nof_vertices] = total;
    blob = xcalloc(total + 1, 1);
    for (i = 0; i < g->nof_vertices; i++) memcpy(blob + off[i], g->ctg[i].name, off[i + 1] - off[i]);
    rc = gtsg_deparser_set_names(g->dp, blob, off, g->nof_vertices);
    free(blob); free(off);
    if (rc == GTSG_ELIMIT && g_host_parser != 2) return 0;
    if (rc) return seterr(err, errlen, "distance parser: %s", gtsg_deparser_last_error(g->dp));
    g->dp_names = true;
  }
  *have = 1;
  return 0;
}

/* The distance file through the GPU parser (gts_deparse.hip).  *used = 0: not
   parsed there (no GPU, file outside the regular form or above its limits) --
   the caller runs the host passes instead.  Otherwise res holds the outcome
   and, without an error, the records are on the device. */
static int gpu_parse_distances(GtScaffolderGraph *g, const char *path, GtsgDeParseResult *res,
                               int *used, char *err, size_t errlen)
{
  size_t len = 0;
  char *buf;
  int rc, have = 0;
  struct stat st;
  const bool have_stat = stat(path, &st) == 0;
  *used = 0;
  /* (the file is taken to be the same if device, inode, size and modification time
     agree: a file rewritten in place with the same size inside one tick of the
     clock would not be seen; host-parser mode never takes the kept records) */
  if (g_host_parser == 1) g->dp_parsed = false;
  if (g->dp && g->dp_names && g->dp_parsed && have_stat && g->dp_dev == st.st_dev && g->dp_ino == st.st_ino &&
      g->dp_size == st.st_size && g->dp_mtime.tv_sec == st.st_mtim.tv_sec &&
      g->dp_mtime.tv_nsec == st.st_mtim.tv_nsec) {
    /* the integrity check (count_distances) has parsed this very file for this
       contig set: its records are still on the device (the reference reads the
       file twice, parser.c:150 and :295) */
    *res = g->dp_res;
    *used = 1;
    return 0;
  }
  g->dp_parsed = false;
  if (ensure_parser(g, &have, err, errlen)) return -1;
  if (!have) return 0;
  buf = slurp(path, &len);
  if (!buf) return seterr(err, errlen, "can not read distance file %s", path);
  {
    /* the parser takes less than 4 GB at a time: a larger file goes over in
       pieces that end at line ends, the records are collected on the device
       (GTS_DE_CHUNK, bytes: a smaller piece size, for tests) */
    size_t chunk = (size_t)3 << 30, start = 0;
    const char *env = getenv("GTS_DE_CHUNK");
    uint64_t total = 0, cand = 0;
    if (env && atol(env) > 0) chunk = (size_t)atol(env);
    memset(res, 0, sizeof *res);
    rc = gtsg_deparser_accumulate(g->dp, len > chunk);
    while (!rc && start < len) {
      size_t end = len - start > chunk ? start + chunk : len;
      GtsgDeParseResult part;
      if (end < len) {                       /* up to and including the next newline */
        const char *nl = memchr(buf + end, '\n', len - end);
        end = nl ? (size_t)(nl - buf) + 1 : len;
      }
      rc = gtsg_deparser_parse(g->dp, buf + start, end - start, 0, &part);
      if (rc) break;
      total += part.n_records; cand += part.n_candidates;
      if (part.irregular) { res->irregular = 1; break; }
      if (part.error) { res->error = part.error; res->error_pos = start + part.error_pos; break; }
      start = end;
    }
    res->n_records = total; res->n_candidates = cand;
    if (len == 0 && !rc) rc = gtsg_deparser_parse(g->dp, buf, 0, 0, res);
  }
  free_large(buf, len);
  if (rc == GTSG_ELIMIT && g_host_parser != 2) return 0;
  if (rc) return seterr(err, errlen, "distance parser: %s", gtsg_deparser_last_error(g->dp));
  if (res->irregular) {
    if (g_host_parser == 2)
      return seterr(err, errlen, "distance file %s is outside the GPU parser's regular form", path);
    return 0;
  }
  *used = 1;
  if (have_stat && stat(path, &st) == 0) {
    g->dp_parsed = true; g->dp_res = *res;
    g->dp_dev = st.st_dev; g->dp_ino = st.st_ino; g->dp_size = st.st_size; g->dp_mtime = st.st_mtim;
  }
  return 0;
}

static int engine_err(GtScaffolderGraph *g, int rc, char *err, size_t errlen)
{
  if (rc == 0) return 0;
  snprintf(g->err, sizeof g->err, "%s", g->eng ? gtsg_last_error(g->eng) : "no engine");
  seterr(err, errlen, "%s", g->err);
  return -1;
}

/* ref gt_scaffolder_parser.c:150: integrity check of the .de file; sorts the
   contigs by header (parser.c:172: ids are final from here on) and returns in
   *nof_distances the number of edges the records can create at most (two per
   record between known contigs, parser.c:246-250) */
int gt_scaffolder_parser_count_distances(const GtScaffolderGraph *graph, const char *file_name,
                                         uint64_t *nof_distances, char *err, size_t errlen)
{
  uint64_t valid = 0;
  int rc, used = 0;
  GtsgDeParseResult res;
  if (!graph || !file_name || !nof_distances) return seterr(err, errlen, "invalid argument");
  sort_contigs((GtScaffolderGraph *)graph);   /* the reference sorts through the const, too */
  if (gpu_parse_distances((GtScaffolderGraph *)graph, file_name, &res, &used, err, errlen)) return -1;
  if (used) {
    /* the reference's messages, parser.c:205-236, :286 */
    if (res.error == 1) return seterr(err, errlen, "Invalid record in dist file %s", file_name);
    if (res.error == 2)
      return seterr(err, errlen, "Invalid value for number of pairs in dist file %s", file_name);
    if (res.error == 3) return seterr(err, errlen, "Invalid composition sign in dist file %s", file_name);
    if (res.n_records == 0) return seterr(err, errlen, "distance file %s is empty", file_name);
    *nof_distances = 2 * res.n_records;
    return 0;
  }
  rc = read_distance_records(graph, file_name, 0, &valid, NULL, err, errlen);
  if (!rc) *nof_distances = 2 * valid;
  return rc;
}

/* ref gt_scaffolder_parser.c:295: the records of the file, in file order,
   become the edges of the graph -- on the GPU (gtsg_build_from_records_ex
   restates the per-record find_edge / alter_edge / add_edge updates of
   parser.c:357-378).  One call per graph: the reference sizes its edge array
   for one file (graph.c:391-396). */
int gt_scaffolder_parser_read_distances(const char *filename, GtScaffolderGraph *g,
                                        bool ismatepair, char *err, size_t errlen)
{
  Records r;
  int64_t *seq;
  float *as, *cn;
  uint64_t i;
  int rc, used = 0;
  GtsgDeParseResult res;
  if (!g || !filename) return seterr(err, errlen, "invalid argument");
  if (g->eng) return seterr(err, errlen, "distances have been read into this graph already");
  if (g->nof_edges) return seterr(err, errlen, "the graph holds hand-built edges");
  memset(&r, 0, sizeof r);
  sort_contigs(g);
  if (gpu_parse_distances(g, filename, &res, &used, err, errlen)) return -1;
  /* a file the integrity check would refuse: the reference's second pass
     skips what it cannot scan and takes the rest as it is (parser.c:340-378);
     that is the host code's business */
  if (used && res.error) used = 0;
  if (!used) {
    rc = read_distance_records(g, filename, 1, NULL, &r, err, errlen);
    if (rc) { rec_free(&r); return -1; }
  }
  if (gtsg_create(&g->eng, g_device, NULL) != 0) {
    rec_free(&r);
    g->eng = NULL;
    return seterr(err, errlen, "no MI355X available: the scaffolder engine has no CPU path");
  }
  seq = xcalloc(g->nof_vertices, sizeof *seq);
  as = xcalloc(g->nof_vertices, sizeof *as);
  cn = xcalloc(g->nof_vertices, sizeof *cn);
  for (i = 0; i < g->nof_vertices; i++) {
    seq[i] = (int64_t)g->ctg[i].seq_len; as[i] = g->ctg[i].astat; cn[i] = g->ctg[i].copy_num;
  }
  rc = gtsg_set_contigs(g->eng, g->nof_vertices, seq, as, cn, 0);
  if (!rc && used) {
    /* the records never leave the device */
    uint64_t n = 0;
    const uint32_t *d_root, *d_ctg;
    const int64_t *d_dist, *d_np;
    const float *d_sd;
    const uint8_t *d_flags;
    gtsg_deparser_records(g->dp, &n, &d_root, &d_ctg, &d_dist, &d_sd, &d_np, &d_flags);
    rc = gtsg_build_from_records_ex(g->eng, n, d_root, d_ctg, d_dist, d_sd, d_np, d_flags, 1,
                                    ismatepair ? 1 : 0);
  } else if (!rc)
    rc = gtsg_build_from_records_ex(g->eng, r.n, r.root, r.ctg, r.dist, r.sd, r.np, r.flags, 0,
                                    ismatepair ? 1 : 0);
  free(seq); free(as); free(cn); rec_free(&r);
  if (g->dp) gtsg_deparser_trim(g->dp);   /* the name table stays for the A-statistic file */
  g->dp_parsed = false;
  if (rc) {
    engine_err(g, rc, err, errlen);
    gtsg_destroy(g->eng);
    g->eng = NULL;
    return -1;
  }
  g->nof_edges = g->max_nof_edges = gtsg_num_edges(g->eng);
  free(g->vstate); free(g->estate); free(g->edges);
  g->vstate = xcalloc(g->nof_vertices, 1);
  g->estate = xcalloc(g->nof_edges, 1);
  g->edges = xcalloc(g->nof_edges, sizeof *g->edges);
  g->edges_cached = false;
  return 0;
}

/* ref gt_scaffolder_graph.c:346-419: the same five steps */
int gt_scaffolder_graph_new_from_file(GtScaffolderGraph **out, const char *ctg_filename,
                                      uint64_t min_ctg_len, const char *dist_filename,
                                      bool astat_is_annotated, char *err, size_t errlen)
{
  GtScaffolderGraph *g = NULL;
  uint64_t nof_contigs = 0, nof_distances = 0;
  int rc;
  *out = NULL;
  rc = gt_scaffolder_parser_count_contigs(ctg_filename, min_ctg_len, &nof_contigs, err, errlen);
  if (!rc) {
    g = gt_scaffolder_graph_new(nof_contigs, 0);
    rc = gt_scaffolder_parser_read_contigs(g, ctg_filename, min_ctg_len, astat_is_annotated,
                                           err, errlen);
  }
  if (!rc) rc = gt_scaffolder_parser_count_distances(g, dist_filename, &nof_distances, err, errlen);
  if (!rc) rc = gt_scaffolder_parser_read_distances(dist_filename, g, false, err, errlen);
  if (rc) { gt_scaffolder_graph_delete(g); return -1; }
  *out = g;
  return 0;
}

/* ------------------------------------------------------------------ */
/* algorithms: thin calls into the engine                              */

int gt_scaffolder_graph_mark_repeats(const char *filename, GtScaffolderGraph *g,
                                     float copy_num_cutoff, float astat_cutoff,
                                     char *err, size_t errlen)
{
  bool have_file = filename && strlen(filename) != 0;
  if (!g || !g->eng) return seterr(err, errlen, "graph is not on the GPU");
  if (have_file) {
    /* ref algorithms.c:108-153: one record per line, six tab separated fields */
    size_t len, pos = 0;
    char *buf = slurp(filename, &len), hdr[1025], line[1025];
    float *as, *cn;
    uint64_t i;
    int rc, have = 0, on_gpu = 0;
    if (!buf) return seterr(err, errlen, "can not read A-statistic file %s", filename);
    if (ensure_parser(g, &have, err, errlen)) { free(buf); return -1; }
    if (have) {
      /* the same scan on the GPU (gts_deparse.hip, k_dp_astat) */
      GtsgDeParseResult res;
      as = xcalloc(g->nof_vertices, sizeof *as);
      cn = xcalloc(g->nof_vertices, sizeof *cn);
      for (i = 0; i < g->nof_vertices; i++) { as[i] = g->ctg[i].astat; cn[i] = g->ctg[i].copy_num; }
      rc = gtsg_deparser_parse_astat(g->dp, buf, len, 0, as, cn, 0, &res);
      if (rc && !(rc == GTSG_ELIMIT && g_host_parser != 2)) {
        free(as); free(cn); free(buf);
        return seterr(err, errlen, "A-statistic parser: %s", gtsg_deparser_last_error(g->dp));
      }
      if (!rc && res.irregular && g_host_parser == 2) {
        free(as); free(cn); free(buf);
        return seterr(err, errlen, "A-statistic file %s is outside the GPU parser's regular form", filename);
      }
      if (!rc && !res.irregular) {
        if (res.error) {
          free(as); free(cn); free(buf);
          return seterr(err, errlen, "Invalid record in A-statistic file %s", filename);
        }
        for (i = 0; i < g->nof_vertices; i++) { g->ctg[i].astat = as[i]; g->ctg[i].copy_num = cn[i]; }
        on_gpu = 1;
      }
      free(as); free(cn);
      gtsg_deparser_trim(g->dp);
      g->dp_parsed = false;
    }
    while (!on_gpu && pos < len) {
      size_t ls = pos, ll;
      long n1, n2, n3;
      float copy_num = 0.0f, astat = 0.0f;
      uint32_t id;
      while (pos < len && buf[pos] != '\n') pos++;
      if (pos < len) pos++;
      ll = pos - ls;
      if (ll > LINE_MAX_REF) {
        free(buf);
        return seterr(err, errlen, "line longer than %d characters in A-statistic file %s",
                      LINE_MAX_REF, filename);
      }
      memcpy(line, buf + ls, ll);
      line[ll - 1] = '\0';                 /* algorithms.c:121 */
      if (sscanf(line, "%1024s\t%ld\t%ld\t%ld\t%f\t%f", hdr, &n1, &n2, &n3, &copy_num,
                 &astat) != 6) {
        free(buf);
        return seterr(err, errlen, "Invalid record in A-statistic file %s", filename);
      }
      if (find_contig(g, hdr, &id)) { g->ctg[id].astat = astat; g->ctg[id].copy_num = copy_num; }
    }
    free(buf);
    as = xcalloc(g->nof_vertices, sizeof *as);
    cn = xcalloc(g->nof_vertices, sizeof *cn);
    for (i = 0; i < g->nof_vertices; i++) { as[i] = g->ctg[i].astat; cn[i] = g->ctg[i].copy_num; }
    rc = gtsg_set_astat(g->eng, as, cn, 0);
    free(as); free(cn);
    if (rc) return engine_err(g, rc, err, errlen);
  }
  return engine_err(g, gtsg_mark_repeats(g->eng, have_file, copy_num_cutoff, astat_cutoff),
                    err, errlen);
}

int gt_scaffolder_graph_filter(GtScaffolderGraph *g, float pcutoff, float cncutoff,
                               int64_t ocutoff)
{
  if (!g || !g->eng) return -1;
  return engine_err(g, gtsg_filter(g->eng, pcutoff, cncutoff, ocutoff), NULL, 0);
}
int gt_scaffolder_removecycles(GtScaffolderGraph *g)
{
  if (!g || !g->eng) return -1;
  return engine_err(g, gtsg_removecycles(g->eng), NULL, 0);
}
int gt_scaffolder_makescaffold(GtScaffolderGraph *g)
{
  if (!g || !g->eng) return -1;
  return engine_err(g, gtsg_makescaffold(g->eng), NULL, 0);
}

/* ------------------------------------------------------------------ */
/* output                                                              */

static int refresh(GtScaffolderGraph *g)
{
  int rc = 0;
  if (!g->eng) return 0;
  if (!g->edges_cached && g->nof_edges) {
    uint64_t m = g->nof_edges, i;
    uint32_t *s = xcalloc(m, 4), *e = xcalloc(m, 4);
    int64_t *d = xcalloc(m, 8), *np = xcalloc(m, 8);
    float *sd = xcalloc(m, 4);
    uint8_t *fl = xcalloc(m, 1);
    rc = gtsg_get_edges(g->eng, s, e, d, sd, np, fl);
    for (i = 0; i < m && !rc; i++) {
      g->edges[i].start = s[i]; g->edges[i].end = e[i]; g->edges[i].dist = d[i];
      g->edges[i].std_dev = sd[i]; g->edges[i].num_pairs = np[i]; g->edges[i].flags = fl[i];
    }
    free(s); free(e); free(d); free(np); free(sd); free(fl);
    if (rc) return engine_err(g, rc, NULL, 0);
    g->edges_cached = true;
  }
  rc = gtsg_get_vertex_states(g->eng, g->vstate);
  if (!rc && g->nof_edges) rc = gtsg_get_edge_states(g->eng, g->estate);
  return engine_err(g, rc, NULL, 0);
}

/* buffered text output without stdio's formatting: a 1 MB buffer, pieces are
   copied (or digits written) straight into it while there is room */
#define OB_CAP (1u << 20)
typedef struct { FILE *f; size_t n; int bad; char *buf; } OutBuf;
static void ob_init(OutBuf *o, FILE *f) { o->f = f; o->n = 0; o->bad = 0; o->buf = xcalloc(OB_CAP, 1); }
static int ob_flush(OutBuf *o)
{
  if (o->n && fwrite(o->buf, 1, o->n, o->f) != o->n) o->bad = 1;
  o->n = 0;
  return o->bad;
}
static int ob_close(OutBuf *o) { int bad = ob_flush(o); free(o->buf); o->buf = NULL; return bad; }
static inline void ob_mem(OutBuf *o, const char *p, size_t len)
{
  if (len <= OB_CAP - o->n) { memcpy(o->buf + o->n, p, len); o->n += len; return; }
  while (len) {
    size_t room = OB_CAP - o->n, k = len < room ? len : room;
    memcpy(o->buf + o->n, p, k);
    o->n += k; p += k; len -= k;
    if (o->n == OB_CAP) ob_flush(o);
  }
}
static inline void ob_str(OutBuf *o, const char *s) { ob_mem(o, s, strlen(s)); }
static inline void ob_u64(OutBuf *o, uint64_t v)
{
  char t[24];
  int k = 24;
  do { t[--k] = (char)('0' + v % 10); v /= 10; } while (v);
  ob_mem(o, t + k, (size_t)(24 - k));
}

/* the edges in id order (accessor for bindings and tests; arrays of
   gt_scaffolder_graph_nof_edges elements, any may be NULL) */
int gt_scaffolder_graph_get_edges(GtScaffolderGraph *g, uint32_t *start, uint32_t *end, int64_t *dist,
                                  float *std_dev, int64_t *num_pairs, uint8_t *flags)
{
  uint64_t i;
  if (!g) return -1;
  if (refresh(g)) return -1;
  for (i = 0; i < g->nof_edges; i++) {
    if (start) start[i] = g->edges[i].start;
    if (end) end[i] = g->edges[i].end;
    if (dist) dist[i] = g->edges[i].dist;
    if (std_dev) std_dev[i] = g->edges[i].std_dev;
    if (num_pairs) num_pairs[i] = g->edges[i].num_pairs;
    if (flags) flags[i] = g->edges[i].flags;
  }
  return 0;
}

/* ref gt_scaffolder_graph.c:247-266: opens the file and prints into it */
int gt_scaffolder_graph_print(const GtScaffolderGraph *cg, const char *filename,
                              char *err, size_t errlen)
{
  GtScaffolderGraph *g = (GtScaffolderGraph *)cg;
  FILE *f = fopen(filename, "w");
  int rc;
  if (!f) return seterr(err, errlen, "cannot open %s for writing", filename);
  g->err[0] = 0;
  rc = gt_scaffolder_graph_print_generic(cg, f);
  if (fclose(f) != 0) rc = rc ? rc : -1;
  if (rc) return seterr(err, errlen, g->err[0] ? "%s" : "cannot write %s", g->err[0] ? g->err : filename);
  return 0;
}

/* ref gt_scaffolder_graph.c:269-307 (the reference takes a GtFile and returns
   nothing: a failed write ends the program there; here it is the return value,
   the message is left in the graph: gt_scaffolder_graph_last_error) */
int gt_scaffolder_graph_print_generic(const GtScaffolderGraph *cg, FILE *f)
{
  static const char *const color[] = {"black", "gray80", "gainsboro", "ivory3",
                                      "red", "green", "magenta", "blue"};
  GtScaffolderGraph *g = (GtScaffolderGraph *)cg;
  OutBuf ob;
  uint64_t i;
  double t0 = now_s();
  if (!g || !f) return -1;
  if (g->eng && !g_host_dot) {
    /* a graph on the GPU: its edge lines (all but a few per cent of the file)
       are formatted there, 2^23 edges at a time; only the vertex states cross
       the bus besides the text */
    const uint64_t chunk = 1ull << 23;
    int rc = gtsg_get_vertex_states(g->eng, g->vstate);
    if (rc) { engine_err(g, rc, NULL, 0); return -1; }
    ob_init(&ob, f);
    ob_str(&ob, "digraph {\n");
    for (i = 0; i < g->nof_vertices; i++) {
      /* (the headers are strings of their own all over the heap) */
      if (i + 16 < g->nof_vertices) __builtin_prefetch(g->ctg[i + 16].name);
      ob_u64(&ob, i);
      ob_str(&ob, " [color=\""); ob_str(&ob, color[g->vstate[i] & 7]);
      ob_str(&ob, "\" label=\""); ob_str(&ob, g->ctg[i].name);
      ob_str(&ob, "\"];\n");
    }
    rc = ob_close(&ob);
    lap("dot: vertex lines", &t0);
    for (i = 0; i < g->nof_edges && !rc; i += chunk) {
      uint64_t cnt = g->nof_edges - i < chunk ? g->nof_edges - i : chunk, nb = 0;
      const char *text = NULL;
      rc = gtsg_format_dot_edges_pinned(g->eng, i, cnt, &text, &nb);
      if (rc) { engine_err(g, rc, NULL, 0); break; }
      lap("dot: chunk formatted", &t0);
      {
        /* past stdio: the chunk (half a gigabyte) goes out in parallel pieces; a
           stream that cannot seek (a pipe) gets it through stdio */
        off_t at;
        if (fflush(f) != 0) rc = -1;
        else if ((at = ftello(f)) < 0) { if (fwrite(text, 1, nb, f) != nb) rc = -1; }
        else if (io_parallel(fileno(f), (char *)(uintptr_t)text, nb, at, 1) ||
                 fseeko(f, at + (off_t)nb, SEEK_SET) != 0)
          rc = -1;
      }
      lap("dot: chunk written", &t0);
    }
    if (!rc && fwrite("}\n", 1, 2, f) != 2) rc = -1;
    if (!rc && fflush(f) != 0) rc = -1;
    return rc ? -1 : 0;
  }
  if (refresh(g)) return -1;
  /* the lines of gt_scaffolder_graph_print_generic (graph.c:269-307), put
     together by hand: fprintf costs more than everything the GPU does */
  ob_init(&ob, f);
  ob_str(&ob, "digraph {\n");
  for (i = 0; i < g->nof_vertices; i++) {
    ob_u64(&ob, i);
    ob_str(&ob, " [color=\""); ob_str(&ob, color[g->vstate[i] & 7]);
    ob_str(&ob, "\" label=\""); ob_str(&ob, g->ctg[i].name);
    ob_str(&ob, "\"];\n");
  }
  for (i = 0; i < g->nof_edges; i++) {
    const HEdge *e = g->edges + i;
    ob_u64(&ob, e->start); ob_str(&ob, " -> "); ob_u64(&ob, e->end);
    ob_str(&ob, " [color=\""); ob_str(&ob, color[g->estate[i] & 7]);
    ob_str(&ob, "\" label=\"");
    if (e->dist < 0) { ob_str(&ob, "-"); ob_u64(&ob, (uint64_t)0 - (uint64_t)e->dist); }
    else ob_u64(&ob, (uint64_t)e->dist);
    ob_str(&ob, (e->flags & 1) ? "\" arrowhead=\"normal\"];\n" : "\" arrowhead=\"inv\"];\n");
  }
  ob_str(&ob, "}\n");
  if (ob_close(&ob) || fflush(f) != 0) return -1;
  return 0;
}


/* ref gt_scaffolder_graph.c:421-500; exit status 2 stands for the reference's
   failing assertion */
int gt_scaffolder_graph_test(uint64_t max_v, uint64_t max_e, bool init_v,
                             uint64_t nv, bool init_e, uint64_t ne, bool print_graph,
                             char *err, size_t errlen)
{
  GtScaffolderGraph *g = gt_scaffolder_graph_new(max_v, max_e);
  uint64_t i, v1 = 0, v2 = 0;
  int rc = 0;
  if (init_v)
    for (i = 0; i < nv && !rc; i++)
      if (gt_scaffolder_graph_add_vertex(g, "foobar", 100, 20, 40)) rc = 2;
  if (init_e && !rc)
    for (i = 0; i < ne && !rc; i++) {
      if (v2 + 1 < nv) v2++;
      else if (v1 + 2 < nv) { v1++; v2 = v1 + 1; }
      if (gt_scaffolder_graph_add_edge(g, v1, v2, 2, 1.5f, 4, true, true)) rc = 2;
    }
  if (!rc && print_graph)
    rc = gt_scaffolder_graph_print(g, "gt_scaffolder_graph_test.dot", err, errlen);
  gt_scaffolder_graph_delete(g);
  return rc;
}

/* ref gt_scaffolder_parser.c:55-147: normalised copy of a .de file */
int gt_scaffolder_parser_read_distances_test(const char *filename, const char *out,
                                             char *err, size_t errlen)
{
  size_t len, pos = 0;
  char *buf = slurp(filename, &len), line[1025], hdr[1024];
  FILE *f;
  if (!buf) return seterr(err, errlen, "can not read distance file %s", filename);
  f = fopen(out, "w");
  if (!f) { free(buf); return seterr(err, errlen, "cannot open %s for writing", out); }
  while (pos < len) {
    size_t ls = pos, ll;
    char *save = NULL, *field;
    bool sense = true, first_antisense = true;
    long dist, np;
    float sd;
    while (pos < len && buf[pos] != '\n') pos++;
    if (pos < len) pos++;
    ll = pos - ls;
    if (ll > LINE_MAX_REF) ll = LINE_MAX_REF;
    memcpy(line, buf + ls, ll);
    line[ll - 1] = '\0';
    field = strtok_r(line, " ", &save);
    fprintf(f, "%s", field ? field : "(null)");
    while (field) {
      if (scan_record(field, hdr, &dist, &np, &sd)) {
        size_t hl = strlen(hdr);
        bool same = hdr[hl - 1] == '+';
        if (np < 0) {
          fclose(f); free(buf);
          return seterr(err, errlen, "Invalid value for number of pairs");
        }
        hdr[hl - 1] = '\0';
        fprintf(f, " %s%c,%ld,%ld,%.1f", hdr, same ? '+' : '-', dist, np, sd);
      } else if (*field == ';')
        sense = !sense;
      field = strtok_r(NULL, " ", &save);
      if (!sense && first_antisense) { fputs(" ;", f); first_antisense = false; }
    }
    if (sense) fputs(" ;", f);
    fputc('\n', f);
  }
  fclose(f);
  free(buf);
  return 0;
}

/* ------------------------------------------------------------------ */
/* scaffold records: ref gt_scaffolder_algorithms.c:901-997 on the host copy
   of the final states (adjacency = edges grouped by start in id order)      */

static bool v_marked(uint8_t s) { return s == 1 || s == 3 || s == 7; }

/* The SCAFFOLD edges of the graph as a compact CSR in adjacency order (creation
   order per vertex): the record walk looks at nothing else.  From the engine
   (gtsg_get_scaffold_edges: a few per cent of the edge list cross the bus) or,
   for a hand-built graph, from the host's own edge list. */
typedef struct { uint32_t *row; REdge *e; uint64_t cnt; } ScafCsr;

static void scaf_free(ScafCsr *c) { free(c->row); free(c->e); memset(c, 0, sizeof *c); }

static int scaf_csr(GtScaffolderGraph *g, ScafCsr *c)
{
  uint64_t n = g->nof_vertices, m = g->nof_edges, k, v;
  memset(c, 0, sizeof *c);
  c->row = xcalloc(n + 1, sizeof *c->row);
  if (g->eng) {
    uint64_t cnt = 0;
    uint32_t *eid, *end;
    int64_t *dist;
    float *sd;
    uint8_t *fl;
    int rc = gtsg_get_scaffold_edges(g->eng, &cnt, NULL, NULL, NULL, NULL, NULL, NULL);
    if (rc) { scaf_free(c); return rc; }
    eid = xcalloc(cnt, 4); end = xcalloc(cnt, 4); dist = xcalloc(cnt, 8); sd = xcalloc(cnt, 4);
    fl = xcalloc(cnt, 1);
    rc = gtsg_get_scaffold_edges(g->eng, &cnt, c->row, eid, end, dist, sd, fl);
    if (!rc) {
      c->e = xcalloc(cnt, sizeof *c->e);
      c->cnt = cnt;
      for (k = 0; k < cnt; k++) {
        c->e[k].eid = eid[k]; c->e[k].end = end[k]; c->e[k].dist = dist[k];
        c->e[k].std_dev = sd[k]; c->e[k].flags = fl[k];
      }
    }
    free(eid); free(end); free(dist); free(sd); free(fl);
    if (rc) { scaf_free(c); return rc; }
    return 0;
  }
  /* hand-built: a vertex' list is its edges in id order */
  for (k = 0; k < m; k++) if (g->estate[k] == 6) c->row[g->edges[k].start + 1]++;
  for (v = 0; v < n; v++) c->row[v + 1] += c->row[v];
  c->cnt = c->row[n];
  c->e = xcalloc(c->cnt, sizeof *c->e);
  {
    uint32_t *fill = xcalloc(n, sizeof *fill);
    for (k = 0; k < m; k++)
      if (g->estate[k] == 6) {
        const HEdge *x = g->edges + k;
        REdge *o = c->e + c->row[x->start] + fill[x->start]++;
        o->eid = (uint32_t)k; o->end = x->end; o->dist = x->dist; o->std_dev = x->std_dev; o->flags = x->flags;
      }
    free(fill);
  }
  return 0;
}

/* The records of a graph on the GPU (gtsg_scaffold_records): those of the clean
   SCAFFOLD paths come ranked -- roots in index order, the edges of each in walk
   order, the lengths summed.  The open part (paths through a contig a walk of
   makescaffold passed twice: a few per cent of the edges at most) is walked here
   in the reference's order of visits, ref algorithms.c:925-995, on its own short
   edge list; the two parts share no vertex and merge by root index. */
typedef struct {
  uint32_t *root, *off, *eid, *end, *p_root, *p_start, *p_eid, *p_end;
  uint64_t *seqlen;
  int64_t *dist, *p_dist;
  float *sd, *p_sd;
  uint8_t *fl, *p_fl;
} RecDl;

static void recdl_free(RecDl *d)
{
  free(d->root); free(d->off); free(d->eid); free(d->end); free(d->p_root); free(d->p_start); free(d->p_eid);
  free(d->p_end); free(d->seqlen); free(d->dist); free(d->p_dist); free(d->sd); free(d->p_sd); free(d->fl);
  free(d->p_fl);
}

/* first position of the open edge list whose start vertex is >= v */
static uint64_t open_lower(const uint32_t *start, uint64_t n, uint32_t v)
{
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (start[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

static GtScaffolderGraphRecords *records_from_device(GtScaffolderGraph *g, uint64_t **seqlen_out, double *t0)
{
  GtsgRecordCounts c;
  GtsgRecordArrays a;
  RecDl d;
  GtScaffolderGraphRecords *r;
  uint64_t nr, ne, pr, pe, i, j, nb = 0, nbe = 0, *seqlen, *b_len = NULL, *b_off = NULL;
  uint32_t *b_root = NULL, *took = NULL;
  uint64_t capt = 0;
  int rc = gtsg_scaffold_records(g->eng, &c);
  if (rc) { engine_err(g, rc, NULL, 0); return NULL; }
  lap("records: ranked on the device", t0);
  nr = c.n_records; ne = c.n_edges; pr = c.n_open_roots; pe = c.n_open_edges;
  memset(&d, 0, sizeof d);
  d.root = xcalloc(nr + 1, 4); d.off = xcalloc(nr + 1, 4); d.seqlen = xcalloc(nr + 1, 8);
  d.eid = xcalloc(ne + 1, 4); d.end = xcalloc(ne + 1, 4); d.dist = xcalloc(ne + 1, 8); d.sd = xcalloc(ne + 1, 4);
  d.fl = xcalloc(ne + 1, 1);
  d.p_root = xcalloc(pr + 1, 4); d.p_start = xcalloc(pe + 1, 4); d.p_eid = xcalloc(pe + 1, 4);
  d.p_end = xcalloc(pe + 1, 4); d.p_dist = xcalloc(pe + 1, 8); d.p_sd = xcalloc(pe + 1, 4); d.p_fl = xcalloc(pe + 1, 1);
  a.root = d.root; a.off = d.off; a.seqlen = d.seqlen; a.eid = d.eid; a.end = d.end; a.dist = d.dist;
  a.std_dev = d.sd; a.flags = d.fl; a.open_root = d.p_root; a.open_start = d.p_start; a.open_eid = d.p_eid;
  a.open_end = d.p_end; a.open_dist = d.p_dist; a.open_std_dev = d.p_sd; a.open_flags = d.p_fl;
  rc = gtsg_scaffold_records_fetch(g->eng, &a);
  if (rc) { recdl_free(&d); engine_err(g, rc, NULL, 0); return NULL; }
  d.off[nr] = (uint32_t)ne;
  lap("records: download", t0);
  if (pr) {
    /* the open part in the reference's order of visits */
    uint8_t *visited = xcalloc(g->nof_vertices / 8 + 1, 1);
    b_root = xcalloc(pr, 4); b_len = xcalloc(pr, 8); b_off = xcalloc(pr + 1, 8);
    capt = pe + 16;
    took = xrealloc(NULL, capt * sizeof *took);
    for (i = 0; i < pr; i++) {
      const uint32_t v = d.p_root[i];
      uint64_t lo = open_lower(d.p_start, pe, v), hi = lo, len;
      if (visited[v >> 3] >> (v & 7) & 1) continue;
      while (hi < pe && d.p_start[hi] == v) hi++;
      if (hi - lo > 1) continue;
      b_root[nb] = v; b_off[nb] = nbe;
      len = g->ctg[v].seq_len;
      visited[v >> 3] |= (uint8_t)(1u << (v & 7));
      if (hi - lo == 1) {
        uint32_t from = v;
        uint64_t k = lo;
        for (;;) {
          const uint32_t w = d.p_end[k];
          const uint8_t fl = d.p_fl[k];
          const bool sense = fl & 1, same = fl & 2, dir = same ? sense : !sense;
          uint64_t cnt = 0, nk = 0, q;
          if (nbe == capt) { capt *= 2; took = xrealloc(took, capt * sizeof *took); }
          took[nbe++] = (uint32_t)k;
          len += g->ctg[w].seq_len + (uint64_t)d.p_dist[k];
          if (visited[w >> 3] >> (w & 7) & 1) break;
          visited[w >> 3] |= (uint8_t)(1u << (w & 7));
          for (q = open_lower(d.p_start, pe, w); q < pe && d.p_start[q] == w; q++)
            /* (not the edge back to where the walk came from: the twin) */
            if (((d.p_fl[q] & 1) != 0) == dir && d.p_end[q] != from) { cnt++; nk = q; }
          if (cnt != 1) break;
          from = w;
          k = nk;
        }
      }
      b_len[nb] = len;
      nb++;
    }
    b_off[nb] = nbe;
    free(visited);
    lap("records: open part walked", t0);
  }
  r = xcalloc(1, sizeof *r);
  r->g = g;
  r->n = r->cap = nr + nb;
  r->nedge = r->capedge = ne + nbe;
  r->root = xrealloc(NULL, (r->n + 1) * sizeof *r->root);
  r->off = xrealloc(NULL, (r->n + 1) * sizeof *r->off);
  r->edge = xrealloc(NULL, (r->nedge + 1) * sizeof *r->edge);
  seqlen = xrealloc(NULL, (r->n + 1) * sizeof *seqlen);
  {
    uint64_t ia = 0, ib = 0, o = 0, pos = 0;
    while (ia < nr || ib < nb) {
      if (ib == nb || (ia < nr && d.root[ia] < b_root[ib])) {
        r->root[o] = d.root[ia]; r->off[o] = pos; seqlen[o] = d.seqlen[ia];
        for (j = d.off[ia]; j < d.off[ia + 1]; j++) {
          REdge *x = r->edge + pos++;
          x->eid = d.eid[j]; x->end = d.end[j]; x->dist = d.dist[j]; x->std_dev = d.sd[j]; x->flags = d.fl[j];
        }
        ia++;
      } else {
        r->root[o] = b_root[ib]; r->off[o] = pos; seqlen[o] = b_len[ib];
        for (j = b_off[ib]; j < b_off[ib + 1]; j++) {
          const uint32_t k = took[j];
          REdge *x = r->edge + pos++;
          x->eid = d.p_eid[k]; x->end = d.p_end[k]; x->dist = d.p_dist[k]; x->std_dev = d.p_sd[k]; x->flags = d.p_fl[k];
        }
        ib++;
      }
      o++;
    }
    r->off[o] = pos;
  }
  free(b_root); free(b_len); free(b_off); free(took);
  recdl_free(&d);
  lap("records: edges", t0);
  *seqlen_out = seqlen;
  return r;
}

/* ref gt_scaffolder_algorithms.c:901-997.  Vertices in index order; a record
   starts at an unvisited vertex with at most one SCAFFOLD edge and follows the
   SCAFFOLD edges while the way on is unique.  Every test of the reference's
   loop over a vertex' list asks for state SCAFFOLD, so the walk runs on the
   compact CSR of those edges. */
GtScaffolderGraphRecords *
gt_scaffolder_graph_iterate_scaffolds(GtScaffolderGraph *g, uint64_t **scaf_seqlen)
{
  GtScaffolderGraphRecords *r;
  uint64_t n, v, k, *seqlen = NULL;
  uint8_t *vs;
  ScafCsr c;
  double t0 = now_s();
  if (!g) return NULL;
  g_last_record_walk = 1;
  if (g->eng && g_host_records != 1) {
    r = records_from_device(g, &seqlen, &t0);
    if (!r) return NULL;
    g_last_record_walk = 0;
    if (scaf_seqlen) *scaf_seqlen = seqlen; else free(seqlen);
    return r;
  }
  if (g->eng) {
    int rc = gtsg_get_vertex_states(g->eng, g->vstate);
    if (rc) { engine_err(g, rc, NULL, 0); return NULL; }
  }
  lap("records: vertex states", &t0);
  if (scaf_csr(g, &c) != 0) {
    if (g->eng) engine_err(g, -1, NULL, 0);
    return NULL;
  }
  lap("records: SCAFFOLD sub-CSR", &t0);
  n = g->nof_vertices;
  r = xcalloc(1, sizeof *r);
  r->g = g;
  /* at most a record per contig and (but for revisits) an entry per SCAFFOLD
     edge: sized once, not grown by doubling */
  r->cap = n ? n : 1;
  r->root = xrealloc(NULL, r->cap * sizeof *r->root);
  r->off = xrealloc(NULL, (r->cap + 1) * sizeof *r->off);
  seqlen = xrealloc(NULL, r->cap * sizeof *seqlen);
  r->capedge = c.cnt + 16;
  r->edge = xrealloc(NULL, r->capedge * sizeof *r->edge);
  vs = g->vstate;
  for (v = 0; v < n; v++)
    if (!v_marked(vs[v]) && vs[v] != 6) vs[v] = 0;
  {
    /* The walk hops from contig to contig all over the graph: what a step asks
       of the contig it arrives at -- its length, its first two SCAFFOLD edges
       (end, flags; nearly every contig has at most two) and where its list
       starts -- sits in one 32-byte record, one cache miss a step instead of
       three dependent ones (contig table, row offsets, edge list).  The walk
       keeps the positions of the edges it takes; the edges themselves and the
       distances along a record are fetched afterwards, in order. */
    typedef struct { uint64_t seq_len; uint32_t k0, end[2]; uint8_t ns, fl[2]; } VRec;
    /* (on huge pages if the system hands them out: a hop is then a cache miss,
       not a cache miss and a page-table walk) */
    const size_t vr_bytes = (((n ? n : 1) * sizeof(VRec)) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    VRec *vr = aligned_alloc((size_t)2 << 20, vr_bytes);
    if (!vr) { fprintf(stderr, "gt_scaffolder: out of memory\n"); abort(); }
    madvise(vr, vr_bytes, MADV_HUGEPAGE);
    uint32_t *took = xrealloc(NULL, r->capedge * sizeof *took);
    uint64_t capt = r->capedge, i;
    for (v = 0; v < n; v++) {
      const uint32_t b = c.row[v], e = c.row[v + 1];
      VRec *x = vr + v;
      x->seq_len = g->ctg[v].seq_len;
      x->k0 = b;
      x->ns = (uint8_t)(e - b > 255 ? 255 : e - b);
      for (k = 0; k < 2; k++) {
        x->end[k] = b + k < e ? c.e[b + k].end : 0;
        x->fl[k] = b + k < e ? c.e[b + k].flags : 0;
      }
    }
    lap("records: arrays", &t0);
    for (v = 0; v < n; v++) {
      uint64_t len;
      if (vs[v] == 4 || v_marked(vs[v])) continue;
      if (vr[v].ns > 1) continue;
      r->root[r->n] = v;
      r->off[r->n] = r->nedge;
      len = vr[v].seq_len;
      vs[v] = 4;
      if (vr[v].ns == 1) {
        uint32_t from = (uint32_t)v, kk = vr[v].k0, w = vr[v].end[0];
        uint8_t fl = vr[v].fl[0];
        for (;;) {
          const VRec *x = vr + w;
          uint64_t cnt = 0;
          uint32_t nk = 0, nw = 0;
          uint8_t nf = 0;
          bool sense = fl & 1, same = fl & 2, dir;
          if (r->nedge == capt) {
            capt = capt ? 2 * capt : 1024;
            took = xrealloc(took, capt * sizeof *took);
          }
          took[r->nedge++] = kk;
          len += x->seq_len;
          if (vs[w] == 4) break;
          vs[w] = 4;
          dir = same ? sense : !sense;
          if (x->ns <= 2) {
            for (k = 0; k < x->ns; k++)
              /* (not the edge back to where the walk came from: the twin) */
              if (((x->fl[k] & 1) != 0) == dir && x->end[k] != from) {
                cnt++; nk = x->k0 + (uint32_t)k; nw = x->end[k]; nf = x->fl[k];
              }
          } else
            for (k = c.row[w]; k < c.row[w + 1]; k++) {
              const REdge *y = c.e + k;
              if (((y->flags & 1) != 0) == dir && y->end != from) {
                cnt++; nk = (uint32_t)k; nw = y->end; nf = y->flags;
              }
            }
          if (cnt != 1) break;
          from = w;
          kk = nk; w = nw; fl = nf;
        }
      }
      seqlen[r->n] = len;
      r->n++;
    }
    r->off[r->n] = r->nedge;
    lap("records: walk", &t0);
    if (r->nedge > r->capedge) {
      r->capedge = r->nedge;
      r->edge = xrealloc(r->edge, r->capedge * sizeof *r->edge);
    }
    for (i = 0; i < r->n; i++) {
      uint64_t j, d = 0;
      for (j = r->off[i]; j < r->off[i + 1]; j++) {
        r->edge[j] = c.e[took[j]];
        d += (uint64_t)r->edge[j].dist;
      }
      seqlen[i] += d;
    }
    free(vr); free(took);
  }
  if (r->off || r->n == 0) r->off = xrealloc(r->off, (r->n + 1) * sizeof *r->off);
  lap("records: edges", &t0);
  scaf_free(&c);
  if (scaf_seqlen) *scaf_seqlen = seqlen; else free(seqlen);
  return r;
}

uint64_t gt_scaffolder_graph_records_size(const GtScaffolderGraphRecords *r) { return r ? r->n : 0; }

void gt_scaffolder_graph_records_delete(GtScaffolderGraphRecords *r)
{
  if (!r) return;
  free(r->root); free(r->off); free(r->edge); free(r);
}

/* ref gt_scaffolder_algorithms.c:1000-1042 */
/* printf("%f") of a float, by hand: the value is a double with at most 24
   significant bits, so the integer part and six decimals -- rounded to nearest,
   ties to even, as glibc does on the exact value -- come out of exact double
   arithmetic (fraction * 10^6 has at most 44 bits).  Anything else (1e15 and
   above, infinities, NaN) goes through snprintf. */
static void ob_f6(OutBuf *o, float x)
{
  double d = (double)x, a = d < 0 ? -d : d, ip, fr, y;
  char t[32];
  if (!(a < 1e15)) {
    char big[400];
    int k = snprintf(big, sizeof big, "%f", d);
    ob_mem(o, big, (size_t)(k > 0 ? k : 0));
    return;
  }
  ip = floor(a);
  fr = a - ip;                       /* exact */
  y = nearbyint(fr * 1e6);           /* exact product, round half to even */
  if (y >= 1e6) { y -= 1e6; ip += 1.0; }
  if (signbit(d)) ob_mem(o, "-", 1);
  ob_u64(o, (uint64_t)ip);
  {
    uint64_t q = (uint64_t)y;
    int k;
    t[0] = '.';
    for (k = 6; k >= 1; k--) { t[k] = (char)('0' + q % 10); q /= 10; }
    ob_mem(o, t, 7);
  }
}

/* ref gt_scaffolder_algorithms.c:1000-1040: one line per record,
   "root\tcontig,dist,std_dev,sense,same,\t..." */
int gt_scaffolder_graph_write_scaffold(const GtScaffolderGraphRecords *r,
                                       const char *file_name, char *err, size_t errlen)
{
  FILE *f;
  OutBuf obuf, *ob = &obuf;
  uint64_t i, j;
  int bad;
  if (!r) return seterr(err, errlen, "no records");
  f = fopen(file_name, "w");
  if (!f) return seterr(err, errlen, "can not create file %s", file_name);
  ob_init(ob, f);
  for (i = 0; i < r->n; i++) {
    ob_str(ob, r->g->ctg[r->root[i]].name);
    for (j = r->off[i]; j < r->off[i + 1]; j++) {
      const REdge *e = r->edge + j;
      /* the headers of the end contigs are strings of their own all over the
         heap: two cache misses an edge unless they are asked for ahead of time */
      if (j + 16 < r->nedge) __builtin_prefetch(&r->g->ctg[r->edge[j + 16].end]);
      if (j + 8 < r->nedge) __builtin_prefetch(r->g->ctg[r->edge[j + 8].end].name);
      ob_mem(ob, "\t", 1);
      ob_str(ob, r->g->ctg[e->end].name);
      ob_mem(ob, ",", 1);
      if (e->dist < 0) { ob_mem(ob, "-", 1); ob_u64(ob, (uint64_t)0 - (uint64_t)e->dist); }
      else ob_u64(ob, (uint64_t)e->dist);
      ob_mem(ob, ",", 1);
      ob_f6(ob, e->std_dev);
      ob_mem(ob, (e->flags & 1) ? ",1" : ",0", 2);
      ob_mem(ob, (e->flags & 2) ? ",1," : ",0,", 3);
    }
    ob_mem(ob, "\n", 1);
  }
  bad = ob_close(ob);
  if (fclose(f) != 0 || bad) return seterr(err, errlen, "can not write file %s", file_name);
  return 0;
}
