/*
  gts_engine.hip -- MI355X (gfx950) scaffold-graph engine behind the C ABI of
  include/gt_scaffold_hip.h.

  Pipeline (all on one HIP stream, graph resident in HBM):
    build      records -> (pair key, record#) radix sort -> segment scan picks
               the creating record and the surviving estimate per direction
               (ref parser.c:357-378) -> edge ids by prefix sum -> stable sort
               by start vertex -> CSR in adjacency order, twin positions.
    repeats    ref algorithms.c:155-167 as one vertex pass + one edge pass.
    filter     gts_filter.hpp: pair pass, two lexicographic fixpoints solved in
               rounds, last-writer edge states.
    components weak components (atomic min-root hooking), slots sorted by
               (component, vertex), compact CSR of live edges, then one
               wavefront per component runs gts_component.hpp.
*/
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <vector>

#define GTS_NKLASS 11
/* LDS size classes of the component launches (bytes of dynamic LDS).  (Coarser
   tables measured worse: nine classes -- 48 K with 64 K, 96 K with 160 K, whose
   few hundred components have room either way -- 59.3 against 55.5 ms per step:
   a workgroup that asks for a CU's whole LDS waits until one has drained
   completely, which the small classes' workgroups prevent for milliseconds;
   seven classes 60.8 ms.) */
static const uint32_t gts_klass_bytes[GTS_NKLASS] = {4096, 6144, 8192, 12288, 16384, 24576, 32768,
                                                    49152, 65536, 98304, 159744};
/* profile-event names of the class launches: string literals, so the pointers
   kept in the pending-event table are valid in every thread for the life of
   the library ([0] removecycles, [1] makescaffold) */
static const char *const gts_klass_event[2][GTS_NKLASS] = {
  {"components_removecycles_lds4k", "components_removecycles_lds6k", "components_removecycles_lds8k",
   "components_removecycles_lds12k", "components_removecycles_lds16k", "components_removecycles_lds24k",
   "components_removecycles_lds32k", "components_removecycles_lds48k", "components_removecycles_lds64k",
   "components_removecycles_lds96k", "components_removecycles_lds160k"},
  {"components_makescaffold_lds4k", "components_makescaffold_lds6k", "components_makescaffold_lds8k",
   "components_makescaffold_lds12k", "components_makescaffold_lds16k", "components_makescaffold_lds24k",
   "components_makescaffold_lds32k", "components_makescaffold_lds48k", "components_makescaffold_lds64k",
   "components_makescaffold_lds96k", "components_makescaffold_lds160k"}};
#define GTS_NSTREAMS 6
static const char *klass_event(int makescaffold, uint32_t bytes);
/* device scalars of the class bookkeeping (u32 index into d_scalars, 16 entries each) */
#define GTS_S_KSIZE 256
#define GTS_S_KCOUNT 272
#define GTS_S_KBYTES 288   /* u64 */
#define GTS_S_KSLOTS 320
#define GTS_S_TQBASE 336
#define GTS_S_TQCNT 352    /* u64 */
#define GTS_S_NDEF 384     /* u64 */
#define GTS_POOL_WAVES 16u /* wavefronts of a k_components_pool workgroup (one per CU) */
#define GTS_FAST_WAVES 10 /* of a k_components_fast workgroup (two per CU: 20 wavefronts, 96 registers a lane) */
#define GTS_S_POOLCUR 392  /* u64: claim counter of k_components_pool */
#define GTS_S_POOLSTAT 400 /* 10 x u64: clocks, give-up and overrun counts of k_components_pool */
#define GTS_S_TEAMUSED 424 /* u64: bytes of the team slab handed out */
#define GTS_S_TEAMSTAT 432 /* 8 x u64: statistics of k_components_team */
#define GTS_S_SMALLSTAT 448 /* 4 x u64: small components by "all edges live" */
#define GTS_S_COLD 464      /* 8 x u64: the cold list's words (GTS_COLD_*) */
#define GTS_S_POOLTOT 512   /* 3 x u64: totals of the pool's programs that keep no per-component statistics */
#define GTS_S_VIEW 2048     /* byte offset 8192: the GtsCompView of the pool kernels (d_scalars holds 16 KB) */
#define GTS_S_FASTSTAT 480  /* 16 x u64: statistics of k_components_fast (as GTS_S_POOLSTAT, + [10..12]) */

static const char *klass_event(int makescaffold, uint32_t bytes)
{
  for (int k = 0; k < GTS_NKLASS; ++k)
    if (gts_klass_bytes[k] == bytes) return gts_klass_event[makescaffold ? 1 : 0][k];
  return makescaffold ? "components_makescaffold_lds" : "components_removecycles_lds";
}

#include "../../include/gt_scaffold_hip.h"
#include "gts_amb_host.h"
#include "gts_component.hpp"
#include "gts_defs.h"
#include "gts_filter.hpp"
#include "gts_prims.hpp"

/* ------------------------------------------------------------------ */
/* engine object                                                       */

struct GtsgEngine {
  int device = 0;
  hipStream_t st = nullptr;
  bool own_stream = false;
  hipStream_t side[GTS_NSTREAMS] = {};  /* class launches */
  hipEvent_t ev_fork = nullptr, ev_join[GTS_NKLASS] = {};
  hipStream_t team_st = nullptr;        /* k_components_team: next to the class launches AND the rounds of walk tasks */
  hipEvent_t ev_team = nullptr;
  std::string err;
  /* vertices */
  uint32_t n = 0;
  int64_t *seq_len = nullptr;
  float *astat = nullptr, *copy_num = nullptr;
  uint8_t *vstate = nullptr;
  uint32_t *vtime = nullptr;   /* time stamps of a shard's vertices (gtsg_set_vertex_times), or null */
  bool have_vtime = false;
  /* edges, CSR / adjacency order */
  uint32_t m = 0;
  uint32_t *row = nullptr, *estart = nullptr, *eend = nullptr, *twin = nullptr,
           *eid = nullptr, *pos_of_eid = nullptr;
  int64_t *dist = nullptr, *npairs = nullptr;
  float *sd = nullptr;
  uint8_t *flags = nullptr, *state = nullptr;
  uint32_t nhub = 0;
  uint32_t *hubs = nullptr;
  uint32_t built_hub_degree = 32;   /* the hub list holds the vertices above THIS degree */
  bool built = false;
  /* scratch of an open gtsg_filter_begin / gtsg_filter_end pair */
  bool filter_open = false;
  uint32_t *f_tpoly = nullptr, *f_lasthit = nullptr;
  uint8_t *f_ovf = nullptr, *f_newstate = nullptr;
  /* workspace */
  char *pool = nullptr;
  size_t pool_cap = 0, pool_used = 0;
  uint32_t *d_scalars = nullptr;   /* 16 x u64 device scalars */
  /* options */
  /* ring of a reference search: walk_queue_factor x compact edges of its
     component to begin with (a walk that overflows it takes a larger one) */
  int64_t walk_queue_factor = 8, max_walk_pops = 1ll << 32, hub_degree = 32;
  int64_t walk_pool_entries = 1ll << 26;
  int64_t class_streams = GTS_NSTREAMS;

  int64_t mixed_task_limit = 256;
  int64_t defer_min_contigs = 256, walk_path_entries = 1ll << 24;
  /* terminals x contigs from which a component's walks fan out.  Made in place
     they take about 0.12 us per vertex step; a round of tasks costs ~0.5 ms of
     launches and host round trips, and a component's walks are all its
     wavefront has to do while the other classes keep the GPU busy.  On the
     10 M workload (largest component 1012 contigs, at most 72 terminals) no
     deferral at all is fastest: 55.8 ms per step against 60.0 with every
     component of 256 contigs deferred (gpurun_out/r02m) */
  int64_t defer_min_work = 1ll << 17;
  int64_t defer_unclean_work = 2048; /* components that are not clean: walks as tasks from this many terminals x contigs on,
                                        once another component of the launch has deferred (0: off) */
  /* a component of at least this many contigs hands a walk that needs the
     reference's search (and the walks of the ccs behind it) to tasks instead of
     replaying the search in line: 15 such walks of one 248-contig component
     took 273 ms one after the other on its wavefront (profiles/r02x_*) */
  int64_t defer_ref_min_contigs = 48;
  int64_t task_reference_walks = 1;
  int64_t pool_components = 1;        /* all LDS components in one launch (k_components_pool) */
  int64_t pool_waves = GTS_POOL_WAVES; /* wavefronts per workgroup of that launch */
  int64_t pool_fill_kb = 4;            /* the pool's fill cursor starts at the components of at most this footprint */
  /* round 4, measured and not the default: a lean program (no reference search, no
     task tables: 89 VGPRs, or 96 on two workgroups of fast_waves wavefronts per CU
     with half a pool each -- fast_split) on k_components_fast, and what it cannot
     finish handed to cold_cus workgroups of the full program next to it.  With the
     same walk code the full program on one workgroup per CU is as fast on the
     headline graph (10.2 ms either way: the launch is bound by LDS x time, not by
     registers or wavefronts) and faster wherever components are handed over (a
     component that needs the reference's search is found out late and starts
     again: 124 against 117 ms on the inversions workload) */
  int64_t fast_components = 0, fast_waves = GTS_FAST_WAVES, cold_cus = 8;
  int64_t fast_split = 0;
  int64_t help_walks = 1;    /* k_components_pool: walks of a cc that have to be made one by one go over the workgroup's wavefronts (GtsHelpJob) */
  int64_t local_marks = 1;   /* LDS programs keep their CYCLIC / SCAFFOLD marks in the working copy until they are done */
  int64_t timing_skip_writeback = 0;   /* timing aid (results are wrong): what the scattered write-back of the fast program costs */   /* two workgroups with half a pool each per CU; 0: one with the whole pool */
  bool pair_sort_full = false;         /* build: sort the records on (larger, smaller) contig, not by the larger one only */
  int64_t pair_bucket_limit = 2048;    /* build: records a thread of k_pair_segments_bucket looks at one way before the full sort is asked for */
  bool gather_nt = true;               /* build: nontemporal stores for the coalesced outputs of the gather kernels */
  int64_t gather_unroll = 4;           /* edges a thread of the gather-shaped build kernels (1: A/B measurements) */
  int64_t lds_poison = -1;             /* test aid: fill a component's pages with this byte before staging */
  int64_t pool_wait_limit_us = 10000000; /* bound of every wait inside that launch (0: test aid, a wait gives up at once) */
  int n_cus = 256;
  int64_t fast_walks = 1, lds_components = 1;
  /* clean LDS-resident components sweep the walks of a cc side by side
     (walks_clean_batch); from batch_big_contigs contigs on a component asks for
     LDS for batch_big_slots walk slots */
  int64_t batch_walks = 2, batch_big_contigs = 64, batch_big_slots = 3;
  int64_t batch_huge_contigs = 256, batch_huge_slots = 5;   /* second tier: the launch's longest programs */
  int64_t lds_int16_distances = 1;   /* packed layout: int16 distances for components whose distances all fit */
  int64_t team_lds_bytes = 0;          /* test aid: cap of the team kernel's dynamic LDS (0: what the largest component asks for) */
  int64_t team_coff = 0;     /* k_components_team: list offsets in LDS while the walks of a cc are made (measured: walks
                                185 -> 180 ms, cycle removal 51 -> 58 ms on the 50 M workload's largest component: off) */
  int64_t small_masks = 1;   /* topological order of components of at most 64 contigs on bit masks (peel_small) */
  /* walks of global-memory components fan out only on request: the components
     that end up there on the 50 M workload are scaffolds tied together by an
     unmarked hub, where every accepted cc revives an arc out of the hub and
     makes the later ccs' walks stale -- 125 rounds instead of 24, 895 instead
     of 732 ms per step (gpurun_out/r02l) */
  int64_t defer_global_components = 0;
  int64_t global_task_pool_mb = 2048;   /* scratch slabs of the walks deferred from global-memory components */
  char *gtask_pool = nullptr;
  /* components that run from global memory get a workgroup each (k_components_team)
     when there are few of them: the walks of a cc over its wavefronts */
  int64_t team_components = 1, team_max_components = 512, team_pool_mb = 4096;
  char *team_pool = nullptr;
  char *text_buf = nullptr;   /* .dot text of a chunk of edges (gtsg_format_dot_edges) */
  char *rec_buf = nullptr;    /* the scaffold records between gtsg_scaffold_records and ..._fetch */
  size_t rec_counts[4] = {0, 0, 0, 0};
  bool rec_ready = false;
  char *text_host = nullptr;  /* its page-locked host copy (gtsg_format_dot_edges_pinned) */
  size_t text_host_cap = 0;
  int profile = 0;        /* 1: hipEvents around kernels, 2: also per-component clocks */
  /* profiling */
  struct Pending { const char *name; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  std::map<std::string, std::pair<uint64_t, double>> ktimes;
  std::map<std::string, int64_t> stats;
  std::map<void *, size_t> alloc_bytes;
};

static int fail(GtsgEngine *e, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  return code;
}

#define HIPCHK(call)                                                          \
  do {                                                                        \
    hipError_t _r = (call);                                                   \
    if (_r != hipSuccess) {                                                   \
      (void)hipGetLastError();   /* reported here: not left with the thread */ \
      return fail(e, GTSG_EHIP, "%s failed: %s (%s:%d)", #call,               \
                  hipGetErrorString(_r), __FILE__, __LINE__);                 \
    }                                                                         \
  } while (0)

static hipEvent_t get_event(GtsgEngine *e)
{
  if (!e->free_events.empty()) {
    hipEvent_t ev = e->free_events.back();
    e->free_events.pop_back();
    return ev;
  }
  hipEvent_t ev;
  hipEventCreate(&ev);
  return ev;
}

/* launch with optional hipEvent bracketing on the engine's stream */
#define LAUNCH(name, kern, grid, block, ...)                                  \
  do {                                                                        \
    hipEvent_t _a = nullptr, _b = nullptr;                                    \
    if (e->profile) { _a = get_event(e); _b = get_event(e);                   \
                      hipEventRecord(_a, e->st); }                            \
    kern<<<(grid), (block), 0, e->st>>>(__VA_ARGS__);                         \
    if (e->profile) { hipEventRecord(_b, e->st);                              \
                      e->pending.push_back({name, _a, _b}); }                 \
  } while (0)

/* bracket a host-side composite (scan / sort) as one entry */
struct ProfScope {
  GtsgEngine *e; const char *name; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(GtsgEngine *en, const char *nm) : e(en), name(nm)
  { if (e->profile) { a = get_event(e); b = get_event(e); hipEventRecord(a, e->st); } }
  ~ProfScope()
  { if (e->profile) { hipEventRecord(b, e->st); e->pending.push_back({name, a, b}); } }
};

static void collect_times(GtsgEngine *e)
{
  for (auto &p : e->pending) {
    float ms = 0;
    hipEventSynchronize(p.b);
    hipEventElapsedTime(&ms, p.a, p.b);
    auto &k = e->ktimes[p.name];
    k.first++; k.second += ms;
    e->free_events.push_back(p.a);
    e->free_events.push_back(p.b);
  }
  e->pending.clear();
}

static int sync_stream(GtsgEngine *e)
{
  HIPCHK(hipStreamSynchronize(e->st));
  HIPCHK(hipGetLastError());
  collect_times(e);
  return 0;
}

/* ---- workspace: one region, bump-allocated per call, grown between calls */
static int pool_reserve(GtsgEngine *e, size_t bytes)
{
  e->pool_used = 0;
  e->filter_open = false;   /* its scratch lives in the pool */
  if (bytes <= e->pool_cap) return 0;
  HIPCHK(hipStreamSynchronize(e->st));
  if (e->pool) HIPCHK(hipFree(e->pool));
  e->pool = nullptr; e->pool_cap = 0;
  bytes += bytes / 8 + (1u << 20);
  hipError_t r = hipMalloc((void **)&e->pool, bytes);
  if (r != hipSuccess) {
    (void)hipGetLastError();   /* the error stays with the thread otherwise and fails the next, unrelated call */
    e->pool = nullptr;
    return fail(e, GTSG_ENOMEM, "workspace of %zu bytes: %s", bytes,
                hipGetErrorString(r));
  }
  e->pool_cap = bytes;
  return 0;
}
template <typename T>
static T *pool_alloc(GtsgEngine *e, size_t count)
{
  size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  if (e->pool_used + bytes > e->pool_cap) {
    fail(e, GTSG_ENOMEM, "workspace exhausted (%zu + %zu > %zu)", e->pool_used,
         bytes, e->pool_cap);
    return nullptr;
  }
  T *p = (T *)(e->pool + e->pool_used);
  e->pool_used += bytes;
  return p;
}
#define PALLOC(var, T, count)                                                 \
  T *var = pool_alloc<T>(e, (count));                                         \
  if (!var) return GTSG_ENOMEM

/* persistent arrays keep their allocation between calls when it is large
   enough (hipFree/hipMalloc of multi-GB arrays costs milliseconds) */
template <typename T>
static int dev_alloc(GtsgEngine *e, T **p, size_t count)
{
  const size_t bytes = (count ? count : 1) * sizeof(T);
  auto it = e->alloc_bytes.find((void *)*p);
  if (*p && it != e->alloc_bytes.end() && it->second >= bytes) return 0;
  if (*p) { e->alloc_bytes.erase((void *)*p); hipFree(*p); *p = nullptr; }
  hipError_t r = hipMalloc((void **)p, bytes + bytes / 16);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    *p = nullptr;
    return fail(e, GTSG_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(r));
  }
  e->alloc_bytes[(void *)*p] = bytes + bytes / 16;
  return 0;
}

static int read_u32(GtsgEngine *e, const uint32_t *d, uint32_t *h)
{
  HIPCHK(hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}
static int read_u64(GtsgEngine *e, const uint64_t *d, uint64_t *h)
{
  HIPCHK(hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

static inline uint32_t nblk(uint64_t n, uint32_t per = GTS_BLOCK)
{
  uint64_t b = (n + per - 1) / per;
  return (uint32_t)(b ? b : 1);
}
static int bits_for(uint64_t n)
{
  int b = 1;
  while (b < 32 && (1ull << b) < n) ++b;
  return b;
}

/* ------------------------------------------------------------------ */
/* small kernels                                                       */

template <typename T>
__global__ void k_fill(T *p, T v, uint64_t n)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_iota(uint32_t *p, uint64_t n)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}

/* ---- build ---- */
/* One thread per sorted position.  Records of one contig pair are adjacent,
   in file order: the first creates both edges, a later record listed from the
   same root replaces the estimate of "its" direction when its std_dev is
   strictly larger (ref parser.c:359-366, graph.c:219-235).  The per-record
   results start out as "creator, both directions from this record"
   (is_creator = 1, winners = GTS_NONE, set by the caller); only later records
   of a pair and creators that lose a direction are written here, and which
   root a record was listed from is read off bit 63 of its key. */
__global__ void k_pair_segments(const uint64_t *keys, const uint32_t *recs,
                                const float *sd, uint8_t *is_creator, uint8_t *replaced,
                                uint32_t *fwd_win, uint32_t *bwd_win, uint64_t nrec,
                                int never_replace)
{
  const uint64_t PAIR = ~(1ull << 63);
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrec) return;
  const uint64_t key = keys[i];
  if (i > 0 && ((keys[i - 1] ^ key) & PAIR) == 0) { is_creator[recs[i]] = 0; return; }
  if (never_replace) return;   /* ismatepair: an existing edge is never altered (parser.c:362) */
  if (i + 1 >= nrec || ((keys[i + 1] ^ key) & PAIR) != 0) return;
  const uint32_t k0 = recs[i];
  const bool selfloop = (uint32_t)(key >> 32 & 0x7FFFFFFFu) == (uint32_t)key;
  uint32_t fw = k0, bw = k0;
  /* (the usual pair has two records, one from each contig: both deviations are
     fetched together, not one after the other) */
  const uint32_t k1 = recs[i + 1];
  const float s1 = sd[k1];
  float fsd = sd[k0], bsd = fsd;
  for (uint64_t j = i + 1; j < nrec; ++j) {
    const uint64_t kj = keys[j];
    if (((kj ^ key) & PAIR) != 0) break;
    const uint32_t k = j == i + 1 ? k1 : recs[j];
    const float s = j == i + 1 ? s1 : sd[k];
    if (selfloop || ((kj ^ key) >> 63) == 0) { if (fsd < s) { fsd = s; fw = k; } }   /* same root */
    else { if (bsd < s) { bsd = s; bw = k; } }
  }
  /* a direction was taken over by a later record (rare): only then are the two
     winner entries written -- and read (replaced[] is one byte per record) */
  if (fw != k0 || bw != k0) { fwd_win[k0] = fw; bwd_win[k0] = bw; replaced[k0] = 1; }
}

/* The same fold on records that are only BUCKETED: sorted on a few digits of
   the pair key (bmask: the key bits the sorting passes looked at, about log2 of
   the number of records of them, so a bucket holds a handful of records of a
   few pairs).  All records of a pair share a bucket and keep their file order
   there, which is all the fold needs: a record is the creator iff no earlier
   record of its bucket has the same pair; a creator folds the later ones of
   its pair as above.  Half the sorting passes of the full order (three instead
   of six for 100 M records of 10 M contigs) -- and the digits are taken from
   the low bits of BOTH contigs, so a contig with thousands of links does not
   make a long bucket.  The workgroup's 256 keys and 32 either side are staged
   in LDS and looked at eight at a time (independent reads: a loop that ends on
   the key it has just read costs a latency per record); records and
   deviations are fetched by all lanes at once after the search.  A thread
   whose bucket reaches beyond the staged keys goes on in global memory, and
   one that would have to look at more than `limit` records one way raises
   *over: the host then sorts on the full key (ids that agree in their low
   bits, e.g. all multiples of 256; quadratic work here). */
#define GTS_SEG_HALO 32
__global__ void __launch_bounds__(GTS_BLOCK)
k_pair_segments_bucket(const uint64_t *keys, const uint32_t *recs,
                       const float *sd, uint8_t *is_creator, uint8_t *replaced,
                       uint32_t *fwd_win, uint32_t *bwd_win, uint64_t nrec,
                       int never_replace, uint64_t bmask, uint32_t limit, uint32_t *over)
{
  __shared__ uint64_t s_key[GTS_BLOCK + 2 * GTS_SEG_HALO];
  const uint64_t PAIR = ~(1ull << 63);
  const uint64_t base = (uint64_t)blockIdx.x * GTS_BLOCK;
  for (uint32_t t = threadIdx.x; t < GTS_BLOCK + 2 * GTS_SEG_HALO; t += GTS_BLOCK) {
    const uint64_t g = base + t;   /* index + halo */
    s_key[t] = (g >= GTS_SEG_HALO && g - GTS_SEG_HALO < nrec) ? keys[g - GTS_SEG_HALO] : 0ull;
  }
  __syncthreads();
  const uint64_t i = base + threadIdx.x;
  if (i >= nrec) return;
  const uint32_t at = threadIdx.x + GTS_SEG_HALO;
  const uint64_t key = s_key[at];
  {
    const uint32_t avail = i < GTS_SEG_HALO ? (uint32_t)i : GTS_SEG_HALO;   /* staged records before this one */
    bool edge = false, found = false;
    for (uint32_t c = 0; c < GTS_SEG_HALO && !edge && !found; c += 8) {
      uint64_t kj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) kj[u] = s_key[at - c - 1 - u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint64_t x = kj[u] ^ key;
        if (edge || found) continue;
        if (c + 1 + u > avail || (x & bmask) != 0) edge = true;
        else if ((x & PAIR) == 0) found = true;
      }
    }
    if (!edge && !found) {   /* the bucket starts before the staged keys (i >= GTS_SEG_HALO here) */
      const uint64_t jmin = i > limit ? i - limit : 0;
      uint64_t j = i - GTS_SEG_HALO;
      while (j-- > jmin) {
        const uint64_t x = keys[j] ^ key;
        if ((x & bmask) != 0) { edge = true; break; }
        if ((x & PAIR) == 0) { found = true; break; }
      }
      if (!edge && !found && jmin > 0) { *over = 1; return; }
    }
    if (found) { is_creator[recs[i]] = 0; return; }
  }
  if (never_replace) return;   /* ismatepair: an existing edge is never altered (parser.c:362) */
  /* where the pair's next record is, from the staged keys */
  const bool selfloop = ((uint32_t)(key >> 32) & 0x7FFFFFFFu) == (uint32_t)key;
  const uint32_t avail = nrec - 1 - i < GTS_SEG_HALO ? (uint32_t)(nrec - 1 - i) : GTS_SEG_HALO;
  uint32_t first = 0, nmatch = 0;
  bool edge = false;
  for (uint32_t c = 0; c < GTS_SEG_HALO && !edge; c += 8) {
    uint64_t kj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kj[u] = s_key[at + c + 1 + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint64_t x = kj[u] ^ key;
      if (edge) continue;
      if (c + 1 + u > avail || (x & bmask) != 0) { edge = true; continue; }
      if ((x & PAIR) == 0 && nmatch++ == 0) first = c + 1 + u;
    }
  }
  if (nmatch == 0 && edge) return;   /* the pair's only record */
  uint32_t k0 = recs[i], fw = k0, bw = k0;
  float fsd = 0.f, bsd = 0.f;
  if (nmatch) {
    const uint64_t kj = s_key[at + first];
    const uint32_t k = recs[i + first];
    const float s0 = sd[k0], s = sd[k];
    fsd = bsd = s0;
    if (selfloop || ((kj ^ key) >> 63) == 0) { if (fsd < s) { fsd = s; fw = k; } }   /* same root */
    else { if (bsd < s) { bsd = s; bw = k; } }
  }
  if (nmatch > 1 || !edge) {   /* a pair listed three times or more, or a bucket that goes on: one by one */
    if (!nmatch) fsd = bsd = sd[k0];
    const uint64_t jmax = nrec - i - 1 > limit ? i + 1 + limit : nrec;
    uint64_t j = i + (nmatch ? first : GTS_SEG_HALO) + 1;
    for (; j < jmax; ++j) {
      const uint64_t kj = keys[j];
      if (((kj ^ key) & bmask) != 0) break;
      if (((kj ^ key) & PAIR) != 0) continue;
      const uint32_t k = recs[j];
      const float s = sd[k];
      if (selfloop || ((kj ^ key) >> 63) == 0) { if (fsd < s) { fsd = s; fw = k; } }
      else { if (bsd < s) { bsd = s; bw = k; } }
    }
    if (j >= jmax && jmax < nrec) { *over = 1; return; }
  }
  if (fw != k0 || bw != k0) { fwd_win[k0] = fw; bwd_win[k0] = bw; replaced[k0] = 1; }
}

struct __attribute__((aligned(32))) GtsEdgeRec {
  int64_t dist;
  int64_t npairs;
  uint32_t end;
  float sd;
  uint32_t flags;
  uint32_t pad;
};

__global__ void k_emit_edges(const uint8_t *is_creator, const uint8_t *replaced, const uint32_t *jidx,
                             const uint32_t *fwd_win, const uint32_t *bwd_win,
                             const uint32_t *root, const uint32_t *ctg,
                             const int64_t *dist, const float *sd,
                             const int64_t *npairs, const uint8_t *flags,
                             uint32_t *estart, GtsEdgeRec *rec, uint64_t nrec)
{
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nrec) return;
  if (!is_creator[k]) return;
  const uint64_t e0 = 2ull * jidx[k];
  const uint32_t r = root[k], c = ctg[k];
  uint32_t fw = (uint32_t)k, bw = (uint32_t)k;
  if (replaced[k]) { fw = fwd_win[k]; bw = bwd_win[k]; }
  GtsEdgeRec a, b;
  a.dist = dist[fw]; a.npairs = npairs ? npairs[fw] : 0; a.end = c;
  a.sd = sd[fw]; a.flags = flags[fw] & 3u; a.pad = 0;
  b.dist = dist[bw]; b.npairs = npairs ? npairs[bw] : 0; b.end = r;
  b.sd = sd[bw]; b.pad = 0;
  if (bw == k) {  /* twin of the creating record, ref parser.c:369-377 */
    const uint32_t f = flags[k];
    const bool sense = f & GTS_F_SENSE, same = f & GTS_F_SAME;
    const bool twin_dir = same ? !sense : sense;
    b.flags = (twin_dir ? GTS_F_SENSE : 0u) | (same ? GTS_F_SAME : 0u);
  } else
    b.flags = flags[bw] & 3u;
  /* both edges of the pair are in hand here: the "u-turn" mark of an edge (its
     twin would be followed right after it, gts_defs.h) is a function of the two
     flag pairs and rides along as bit 2 of the stored flags, so the component
     compaction does not have to fetch the twin's flags per edge */
  if (((b.flags & GTS_F_SENSE) != 0) == gts_next_dir((uint8_t)a.flags)) a.flags |= GTS_F_UTURN;
  if (((a.flags & GTS_F_SENSE) != 0) == gts_next_dir((uint8_t)b.flags)) b.flags |= GTS_F_UTURN;
  estart[e0] = r; estart[e0 + 1] = c;   /* (the values of the CSR sort, the edge ids, are made by its first pass) */
  rec[e0] = a; rec[e0 + 1] = b;
}

__global__ void k_row_offsets(const uint32_t *sorted_start, uint32_t *row,
                              uint32_t n, uint32_t m)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > m) return;
  const uint32_t lo = i == 0 ? 0u : sorted_start[i - 1] + 1u;
  const uint32_t hi = i == m ? n : sorted_start[i];
  for (uint32_t v = lo; v <= hi; ++v) row[v] = (uint32_t)i;
}

/* coalesced outputs that are not read again soon go past the caches */
template <bool NT, typename T>
__device__ __forceinline__ void gts_store(T *p, T v)
{
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <int GTS_U, bool NT>
__global__ void k_gather_csr(const uint32_t *perm, const GtsEdgeRec *rec,
                             uint32_t *eend, int64_t *dist, int64_t *npairs,
                             float *sd, uint8_t *flags, uint8_t *state,
                             uint32_t *pos_of_eid, uint32_t m)
{
  /* several edges a thread, their random 32-byte records in flight together:
     one dependent gather per thread leaves the memory system idle (1.4 TB/s) */
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x) * GTS_U + threadIdx.x;
  uint32_t id[GTS_U];
  GtsEdgeRec r[GTS_U];
#pragma unroll
  for (int k = 0; k < GTS_U; ++k) {
    const uint64_t p = base + (uint64_t)k * blockDim.x;
    id[k] = p < m ? perm[p] : 0u;
  }
#pragma unroll
  for (int k = 0; k < GTS_U; ++k) r[k] = rec[id[k]];
#pragma unroll
  for (int k = 0; k < GTS_U; ++k) {
    const uint64_t p = base + (uint64_t)k * blockDim.x;
    if (p >= m) continue;
    gts_store<NT>(eend + p, r[k].end); gts_store<NT>(dist + p, r[k].dist);
    gts_store<NT>(npairs + p, r[k].npairs); gts_store<NT>(sd + p, r[k].sd);
    gts_store<NT>(flags + p, (uint8_t)r[k].flags); gts_store<NT>(state + p, (uint8_t)GIS_UNVISITED);
    pos_of_eid[id[k]] = (uint32_t)p;
  }
}
template <int GTS_U, bool NT>
__global__ void k_twins(const uint32_t *eid, const uint32_t *pos_of_eid,
                        uint32_t *twin, uint32_t m)
{
  /* several gathers a thread in flight (see k_gather_csr) */
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x) * GTS_U + threadIdx.x;
  uint32_t id[GTS_U], t[GTS_U];
#pragma unroll
  for (int k = 0; k < GTS_U; ++k) {
    const uint64_t p = base + (uint64_t)k * blockDim.x;
    id[k] = p < m ? eid[p] ^ 1u : 0u;
  }
#pragma unroll
  for (int k = 0; k < GTS_U; ++k) t[k] = pos_of_eid[id[k]];
#pragma unroll
  for (int k = 0; k < GTS_U; ++k) {
    const uint64_t p = base + (uint64_t)k * blockDim.x;
    if (p < m) gts_store<NT>(twin + p, t[k]);
  }
}
__global__ void k_hub_flags(const uint32_t *row, uint32_t *flag, uint32_t n,
                            uint32_t hub_degree)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) flag[v] = (row[v + 1] - row[v]) > hub_degree ? 1u : 0u;
}
__global__ void k_compact_ids(const uint32_t *flag, const uint32_t *idx,
                              uint32_t *out, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n && flag[v]) out[idx[v]] = (uint32_t)v;
}

/* ---- mark_repeats ---- */
__global__ void k_repeat_vertices(const float *astat, const float *cn,
                                  uint8_t *vstate, uint8_t *isrep, uint32_t n,
                                  int have_file, float cncut, float acut)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const bool r = gts_is_repeat(astat[v], cn[v], have_file, cncut, acut);
  isrep[v] = r ? 1 : 0;
  if (r) vstate[v] = GIS_REPEAT;
}
/* an edge turns REPEAT iff one of its ends is marked by THIS call
   (mark_vertex marks the vertex' edges and their twins, algorithms.c:61-87) */
__global__ void k_repeat_edges(const uint32_t *estart, const uint32_t *twin,
                               const uint8_t *isrep, uint8_t *state, uint32_t m)
{
  /* from the repeat's side: its own edges and their twins; an edge of another
     contig reads its start vertex' flag and nothing else */
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m || !isrep[estart[p]]) return;
  state[p] = GIS_REPEAT;
  state[twin[p]] = GIS_REPEAT;
}

/* ---- filter ---- */
/* The pair passes are the only O(degree^2) work of the filter.  A workgroup
   takes GTS_BLOCK consecutive vertices; their adjacency lists are ONE
   contiguous CSR range, which is staged in LDS with coalesced loads (plus one
   gather per edge for the end vertex' copy number and length) and then paired
   from LDS.  Without staging every pair re-gathers from HBM: 58 GB fetched for
   1.9 GB of edges (profiles/r01b_pmc_traffic.json). */
/* staged as 32-bit values (17 B per edge, 3072 edges = 51 KB of LDS: three
   workgroups a CU; with 3584 edges -- two workgroups, eight wavefronts a CU --
   k_filter_pairs took 3.24 instead of 2.44 ms, with 2432 the edges that do not
   fit and are paired from global memory make it 5.8 ms: same-call A/B, round 4);
   a block holding a distance or length outside int32 pairs from global memory */
#ifndef GTS_FP_CAP
#define GTS_FP_CAP 3072
#endif
struct GtsEdgeAccLds {
  const int32_t *d, *l;
  const float *s, *c;
  const uint8_t *f;
  __device__ __forceinline__ uint8_t sense(uint32_t i) const { return f[i] & GTS_F_SENSE; }
  __device__ __forceinline__ int64_t dist(uint32_t i) const { return d[i]; }
  __device__ __forceinline__ float sd(uint32_t i) const { return s[i]; }
  __device__ __forceinline__ float cn(uint32_t i) const { return c[i]; }
  __device__ __forceinline__ int64_t len(uint32_t i) const { return l[i]; }
  __device__ __forceinline__ bool marked(uint32_t i) const { return (f[i] & 0x80u) != 0; }
};

/* what the pair passes gather per edge from the end vertex, in one 16-byte
   record (two separate arrays cost two sectors per edge) */
struct __attribute__((aligned(16))) GtsVAttr { int64_t len; float cn; uint32_t tpoly; };
__global__ void k_pack_vattr(const int64_t *seq_len, const float *copy_num, GtsVAttr *va, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  GtsVAttr a; a.len = seq_len[v]; a.cn = copy_num[v]; a.tpoly = GTS_NONE;
  va[v] = a;
}
/* bitmap of the vertices the polymorphic pass stamped: k_filter_ovf_init asks
   it per edge (n / 8 bytes, L2 resident) and reaches for tpoly[] only for the
   few ends that are polymorphic */
__global__ void k_poly_bitmap(const uint32_t *tpoly, uint32_t *bits, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t b = __builtin_amdgcn_ballot_w64(v < n && tpoly[v] != GTS_NONE);
  const uint32_t lane = threadIdx.x & 63u;
  if (lane == 0 && v < n) bits[v >> 5] = (uint32_t)b;
  if (lane == 32 && v < n) bits[v >> 5] = (uint32_t)(b >> 32);
}
/* "wide" mark of an end vertex' length in elen[] (does not fit the 32-bit staging) */
#define GTS_ELEN_WIDE INT32_MIN

__global__ void __launch_bounds__(GTS_BLOCK)
k_filter_pairs(GtsGraphView G, GtsFilterParams P, const GtsVAttr *va, uint8_t *prop,
               uint8_t *vinfo, uint32_t hub_degree, int32_t *elen)
{
  __shared__ int32_t s_d[GTS_FP_CAP], s_l[GTS_FP_CAP];
  __shared__ float s_s[GTS_FP_CAP], s_c[GTS_FP_CAP];
  __shared__ uint8_t s_f[GTS_FP_CAP];
  __shared__ uint32_t s_wide;
  const uint32_t v0 = blockIdx.x * GTS_BLOCK;
  const uint32_t v1 = v0 + GTS_BLOCK < G.n ? v0 + GTS_BLOCK : G.n;
  const uint32_t e0 = G.row[v0], e1 = G.row[v1];
  uint32_t ns = e1 - e0 < GTS_FP_CAP ? e1 - e0 : GTS_FP_CAP;
  if (threadIdx.x == 0) s_wide = 0;
  __syncthreads();
  /* every edge of the block: the end vertex' length also goes to elen[] in
     edge order, so that the overlap pass (k_filter_ovf_init) reads it back
     coalesced instead of gathering the vertex record a second time */
  /* four edges per thread and trip: the four dependent gathers of the end
     vertices' records are in flight together (a 60 KB workgroup leaves a CU with
     eight wavefronts, one gather each is too little to hide their latency) */
  const uint32_t ne_blk = e1 - e0;
  for (uint32_t i0 = threadIdx.x; i0 < ne_blk; i0 += 4 * GTS_BLOCK) {
    uint32_t xs[4];
    GtsVAttr as[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i = i0 + j * GTS_BLOCK;
      xs[j] = i < ne_blk ? G.end[e0 + i] : G.end[e0];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) as[j] = va[xs[j]];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i = i0 + j * GTS_BLOCK;
      if (i >= ne_blk) continue;
      const uint32_t p = e0 + i;
      const int64_t l = as[j].len;
      const bool lw = l != (int32_t)l || (int32_t)l == GTS_ELEN_WIDE;
      elen[p] = lw ? GTS_ELEN_WIDE : (int32_t)l;
      if (i < ns) {
        const int64_t d = G.dist[p];
        if (d != (int32_t)d || lw) s_wide = 1;
        s_d[i] = (int32_t)d; s_s[i] = G.sd[p]; s_f[i] = G.flags[p];
        s_c[i] = as[j].cn; s_l[i] = (int32_t)l;
      }
    }
  }
  __syncthreads();
  if (s_wide) ns = 0;
  const uint32_t v = v0 + threadIdx.x;
  if (v >= G.n) return;
  if (gts_vertex_is_marked(G.vstate[v])) { vinfo[v] = GTS_VI_INACTIVE; return; }
  const uint32_t b = G.row[v], e = G.row[v + 1];
  if (e - b > hub_degree) { vinfo[v] = 0; return; }
  if (e - e0 <= ns) {
    GtsEdgeAccLds A = {s_d, s_l, s_s, s_c, s_f};
    vinfo[v] = (uint8_t)gts_filter_pairs_acc(A, P, b - e0, e - e0, 0, 1, prop, e0);
  } else {
    GtsEdgeAccGlobal A(G);
    vinfo[v] = (uint8_t)gts_filter_pairs_acc(A, P, b, e, 0, 1, prop, 0);
  }
}
/* hub vertices: one wavefront per vertex, the outer pair index strided over
   the lanes */
__global__ void __launch_bounds__(GTS_BLOCK)
k_filter_pairs_hub(GtsGraphView G, GtsFilterParams P, uint8_t *prop,
                   uint8_t *vinfo, const uint32_t *hubs, uint32_t nhub)
{
  const uint32_t w = (blockIdx.x * GTS_BLOCK + threadIdx.x) >> 6;
  if (w >= nhub) return;
  const uint32_t v = hubs[w];
  if (gts_vertex_is_marked(G.vstate[v])) return;
  uint32_t bits = gts_filter_pairs(G, P, v, gts_lane(), GTS_WAVE, prop);
  const bool s = __ballot(bits & GTS_VI_OVALL_S) != 0;
  const bool a = __ballot(bits & GTS_VI_OVALL_A) != 0;
  if (gts_lane() == 0)
    vinfo[v] = (uint8_t)((s ? GTS_VI_OVALL_S : 0u) | (a ? GTS_VI_OVALL_A : 0u));
}
/* vertices that a smaller vertex proposes, found from the proposing edges
   (few): the others are active at once, without looking at their lists */
__global__ void k_filter_proposed(GtsGraphView G, const uint32_t *estart, const uint8_t *prop,
                                  uint8_t *proposed)
{
  uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= G.m || !prop[q]) return;
  const uint32_t u = estart[q], v = G.end[q];
  if (u < v) proposed[v] = 1;
}
__global__ void k_filter_active_round(GtsGraphView G, const uint8_t *prop, const uint8_t *proposed,
                                      uint8_t *vinfo, uint32_t *pending)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= G.n) return;
  const uint8_t cur = vinfo[v];
  if (cur & (GTS_VI_ACTIVE0 | GTS_VI_INACTIVE)) return;
  const uint32_t r = proposed[v] ? gts_filter_active_round(G, (uint32_t)v, prop, vinfo)
                                 : (uint32_t)GTS_VI_ACTIVE0;
  if (r) vinfo[v] = (uint8_t)(cur | r);
  else *pending = 1;
}
/* first active proposer of every vertex, one lane per edge: an edge
   q = (u -> v) that carries a proposal counts for v if u is active.  Few edges
   do, and the test reads the lane's own byte: nothing is gathered for the
   rest. */
__global__ void k_filter_tpoly(GtsGraphView G, const uint32_t *estart,
                               const uint8_t *prop, const uint8_t *vinfo,
                               uint32_t *tpoly)
{
  uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= G.m) return;
  if (!prop[q]) return;
  const uint32_t u = estart[q], v = G.end[q];
  if (!(vinfo[u] & GTS_VI_ACTIVE0) || gts_vertex_is_marked(G.vstate[v])) return;
  atomicMin(&tpoly[v], u);
}
__global__ void __launch_bounds__(GTS_BLOCK)
k_filter_ovf_init(GtsGraphView G, GtsFilterParams P, const int32_t *elen, const uint32_t *ispoly,
                  const uint32_t *estart, const uint8_t *vinfo, const uint32_t *tpoly, uint8_t *ovf,
                  int zero_ovf, uint32_t hub_degree)
{
  __shared__ int32_t s_d[GTS_FP_CAP], s_l[GTS_FP_CAP];
  __shared__ uint8_t s_f[GTS_FP_CAP];
  __shared__ uint32_t s_need, s_wide;
  const uint32_t v0 = blockIdx.x * GTS_BLOCK;
  const uint32_t v1 = v0 + GTS_BLOCK < G.n ? v0 + GTS_BLOCK : G.n;
  const uint32_t v = v0 + threadIdx.x;
  if (threadIdx.x == 0) { s_need = 0; s_wide = 0; }
  __syncthreads();
  /* who has to pair at all: ACTIVE1 vertices whose mark-free pre-test fired */
  uint32_t o = 0;
  bool pairing = false;
  uint32_t b = 0, e = 0;
  if (v < G.n) {
    const uint8_t vi = vinfo[v];
    b = G.row[v]; e = G.row[v + 1];
    if (!(vi & GTS_VI_ACTIVE0) || tpoly[v] == v) o = GTS_OV_KNOWN;
    else {
      o = GTS_OV_ACTIVE1;
      if (zero_ovf) o |= GTS_OV0_A | GTS_OV0_S;
      else if ((vi & (GTS_VI_OVALL_A | GTS_VI_OVALL_S)) && e - b <= hub_degree) pairing = true;
      /* hubs with the pre-test set are finished by k_filter_ovf_init_hub */
    }
  }
  if (pairing) s_need = 1;
  __syncthreads();
  if (s_need) {
    const uint32_t e0 = G.row[v0], e1 = G.row[v1];
    uint32_t ns = e1 - e0 < GTS_FP_CAP ? e1 - e0 : GTS_FP_CAP;
    for (uint32_t i = threadIdx.x; i < ns; i += GTS_BLOCK) {
      const uint32_t p = e0 + i, x = G.end[p];
      /* polymorphic end that was stamped no later than this vertex' turn */
      const bool pe = ((ispoly[x >> 5] >> (x & 31u)) & 1u) && tpoly[x] <= estart[p];
      const bool mk = gts_edge_is_marked(G.state[p]) || pe;
      const int64_t d = G.dist[p];
      const int32_t l = elen[p];
      if (d != (int32_t)d || l == GTS_ELEN_WIDE) s_wide = 1;
      s_d[i] = (int32_t)d; s_l[i] = l;
      s_f[i] = (uint8_t)((G.flags[p] & 3u) | (mk ? 0x80u : 0u));
    }
    __syncthreads();
    if (s_wide) ns = 0;
    if (pairing) {
      if (e - e0 <= ns) {
        GtsEdgeAccLds A = {s_d, s_l, nullptr, nullptr, s_f};
        o |= gts_filter_ovf0_acc(A, P, b - e0, e - e0, 0, 1);
      } else
        o |= gts_filter_ovf0(G, P, v, 0, 1, tpoly);
    }
  }
  if (v >= G.n) return;
  const bool hub_pending = (o & GTS_OV_ACTIVE1) && !zero_ovf && e - b > hub_degree &&
                           (vinfo[v] & (GTS_VI_OVALL_A | GTS_VI_OVALL_S));
  if ((o & GTS_OV_ACTIVE1) && !zero_ovf && !hub_pending && !(o & (GTS_OV0_A | GTS_OV0_S)))
    o |= GTS_OV_KNOWN;
  ovf[v] = (uint8_t)o;
}
__global__ void __launch_bounds__(GTS_BLOCK)
k_filter_ovf_init_hub(GtsGraphView G, GtsFilterParams P, const uint8_t *vinfo,
                      const uint32_t *tpoly, uint8_t *ovf, const uint32_t *hubs,
                      uint32_t nhub)
{
  const uint32_t w = (blockIdx.x * GTS_BLOCK + threadIdx.x) >> 6;
  if (w >= nhub) return;
  const uint32_t v = hubs[w];
  const uint8_t vi = vinfo[v];
  if (!(vi & GTS_VI_ACTIVE0) || tpoly[v] == v) return;
  if (!(vi & (GTS_VI_OVALL_A | GTS_VI_OVALL_S))) return;
  uint32_t bits = gts_filter_ovf0(G, P, v, gts_lane(), GTS_WAVE, tpoly);
  uint32_t o = GTS_OV_ACTIVE1;
  if (__ballot(bits & GTS_OV0_S)) o |= GTS_OV0_S;
  if (__ballot(bits & GTS_OV0_A)) o |= GTS_OV0_A;
  if (!(o & (GTS_OV0_A | GTS_OV0_S))) o |= GTS_OV_KNOWN;
  if (gts_lane() == 0) ovf[v] = (uint8_t)o;
}
__global__ void k_filter_hit_round(GtsGraphView G, uint8_t *ovf, int zero_ovf,
                                   uint32_t *pending)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= G.n) return;
  if (ovf[v] & GTS_OV_KNOWN) return;
  const uint32_t r = gts_filter_hit_round(G, (uint32_t)v, ovf, zero_ovf != 0);
  if (r) ovf[v] = (uint8_t)r;
  else *pending = 1;
}
/* latest neighbour whose overflow marks direction d of a vertex a, one lane
   per edge t = (y -> a) seen from the overflowing side: the test reads the
   lane's own start vertex and flags, only a hit touches a (lasthit[] is
   pre-set to GTS_NONE = -1 as int32) */
__global__ void k_filter_lasthit(GtsGraphView G, const uint32_t *estart,
                                 const uint8_t *ovf, uint32_t *lasthit, const uint32_t *vtime)
{
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.m) return;
  const uint32_t y = estart[t];
  const uint32_t oy = ovf[y];
  if (!(oy & GTS_OV_ACTIVE1) || !(oy & (GTS_OV_A | GTS_OV_S))) return;
  const uint8_t ff = G.flags[t];
  if (!(oy & ((ff & GTS_F_SENSE) ? GTS_OV_S : GTS_OV_A))) return;
  atomicMax((int *)&lasthit[2 * (uint64_t)G.end[t] + (gts_twin_dir(ff) ? 1 : 0)],
            (int)(vtime ? vtime[y] : y));
}
/* final edge states in two passes: every edge from what its own start vertex
   holds (coalesced), then the edges that END in a polymorphic vertex once more
   with that vertex' time stamp -- found from the polymorphic side, through
   the twin, so that the common edge gathers nothing */
__global__ void k_filter_final(GtsGraphView G, const uint32_t *estart,
                               const uint32_t *tpoly, const uint8_t *ovf,
                               const uint32_t *lasthit, uint8_t *newstate, const uint32_t *vtime)
{
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.m) return;
  newstate[p] = gts_filter_final_edge(G, estart[p], (uint32_t)p, tpoly, ovf, lasthit, false, vtime);
}
__global__ void k_filter_final_poly_ends(GtsGraphView G, const uint32_t *estart,
                                         const uint32_t *tpoly, const uint8_t *ovf,
                                         const uint32_t *lasthit, uint8_t *newstate,
                                         const uint32_t *vtime)
{
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.m || tpoly[estart[t]] == GTS_NONE) return;
  const uint32_t p = G.twin[t];   /* ends in the polymorphic vertex estart[t] */
  newstate[p] = gts_filter_final_edge(G, G.end[t], p, tpoly, ovf, lasthit, true, vtime);
}
__global__ void k_filter_final_vertices(uint8_t *vstate, const uint32_t *tpoly,
                                        uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n && tpoly[v] != GTS_NONE) vstate[v] = GIS_POLYMORPHIC;
}

/* ---- components ---- */
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
  uint32_t p = parent[x];
  while (p != x) {
    const uint32_t gp = parent[p];
    if (gp != p) parent[x] = gp;   /* path halving, benign race */
    x = p; p = gp;
  }
  return x;
}
/* live edge: unmarked edge between unmarked vertices.  Hooks the larger root
   under the smaller one, so a component's label is its smallest vertex. */
__global__ void k_live_union(GtsGraphView G, const uint32_t *estart,
                             uint8_t *incl, uint32_t *parent)
{
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.m) return;
  const uint32_t a = estart[p], b = G.end[p];
  /* an unmarked edge has unmarked ends: whatever marks a vertex marks its
     edges and their twins (mark_vertex, algorithms.c:76-87), and no edge ever
     goes back from a marked state */
  const bool lv = !gts_edge_is_marked(G.state[p]);
  /* an edge enters the compact graph if it or its twin is live: marking a walk
     edge's twin SCAFFOLD (algorithms.c:842-845) revives a marked twin.  The
     twin is looked up only for an edge that is not live itself.  (Round 4, tried
     the other way round -- a live edge sets its own flag and its twin's, no
     look-up by the others: 2.31 -> 3.1 ms; a random byte store costs a sector
     read and a write-back, and the live edges' stores cost more than the dead
     edges' reads.  Also tried: of two live twins only the one that starts at the
     smaller contig joins the pair, the other looks at its twin's state instead of
     searching the forest twice: 2.29 -> 2.39 ms.) */
  incl[p] = lv ? 1 : !gts_edge_is_marked(G.state[G.twin[p]]) ? 1 : 0;
  if (!lv) return;
  /* (which vertices have a live edge at either end is read off the forest
     afterwards, k_component_roots: two random byte writes per live edge cost
     more than the union itself) */
  uint32_t x = a, y = b;
  for (;;) {
    x = uf_find(parent, x); y = uf_find(parent, y);
    if (x == y) break;
    if (x < y) { const uint32_t t = x; x = y; y = t; }   /* x > y */
    const uint32_t old = atomicCAS(&parent[x], x, y);
    if (old == x) break;
  }
}
/* a vertex is joined to another one iff it is not its own root or it is the
   root of some other vertex (has_child); the finds shorten the paths for
   k_slot_keys on their way */
__global__ void k_component_roots(uint32_t *parent, uint8_t *has_child, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const uint32_t r = uf_find(parent, (uint32_t)v);
  if (r != v) has_child[r] = 1;
}
__global__ void k_component_vertices(const uint8_t *has_child, uint8_t *vstate,
                                     const uint32_t *parent, uint32_t *flag,
                                     uint32_t n, int mode)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const bool marked = gts_vertex_is_marked(vstate[v]);
  /* only unmarked vertices are ever joined (a live edge has unmarked ends) */
  const bool touched = parent[v] != v || has_child[v];
  const bool in = !marked && touched;
  flag[v] = in ? 1u : 0u;
  if (!marked && !touched)   /* lonesome vertex: algorithms.c:790-806 */
    vstate[v] = mode == GTS_MODE_MAKESCAFFOLD ? GIS_SCAFFOLD : GIS_UNVISITED;
}
__global__ void k_slot_keys(const uint32_t *flag, const uint32_t *idx,
                            uint32_t *parent, uint32_t *keys, uint32_t *vals,
                            uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n || !flag[v]) return;
  keys[idx[v]] = uf_find(parent, (uint32_t)v);
  vals[idx[v]] = (uint32_t)v;
}
__global__ void k_slot_heads(const uint32_t *labels, uint32_t *head,
                             uint32_t nslots)
{
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nslots) head[s] = (s == 0 || labels[s] != labels[s - 1]) ? 1u : 0u;
}
__global__ void k_slot_finish(const uint32_t *labels, const uint32_t *cidx,
                              const uint32_t *slot_v, const int64_t *seq_len,
                              const uint8_t *vstate, uint32_t *comp_off,
                              uint32_t *slot_of, int64_t *cseq, uint8_t *vst,
                              uint32_t nslots)
{
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  if (s == 0 || labels[s] != labels[s - 1]) comp_off[cidx[s]] = (uint32_t)s;
  const uint32_t v = slot_v[s];
  slot_of[v] = (uint32_t)s;
  cseq[s] = seq_len[v];
  vst[s] = vstate[v];
}
/* first slot of the component of every slot */
/* first slot of the component of every slot; flags components that hold a
   contig length the packed LDS layout (int32) cannot carry */
__global__ void k_slot_bases(const uint32_t *head, const uint32_t *cidx,
                             const uint32_t *comp_off, const int64_t *cseq,
                             uint32_t *slot_base, uint32_t *slot_comp,
                             uint8_t *comp_wide, unsigned long long *comp_len,
                             uint32_t nslots)
{
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  const uint32_t c = cidx[s] + head[s] - 1;
  slot_base[s] = comp_off[c];
  slot_comp[s] = c;
  const int64_t l = cseq[s];
  if (l != (int32_t)l) comp_wide[c] = 1;
  /* bases of the component (len_t of the LDS layout).  Slots are sorted by
     component: the lanes of a run add up first (segmented wave scan), the last
     lane of the run does the atomic */
  unsigned long long sum = (unsigned long long)l;
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(sum, off);
    const uint32_t oc = __shfl_up(c, off);
    if (lane >= (uint32_t)off && oc == c) sum += o;
  }
  const uint32_t nc = __shfl_down(c, 1);
  if (lane == 63u || nc != c || s + 1 == nslots) atomicAdd(&comp_len[c], sum);
}
/* (which edges enter the compact graph: k_live_union; the rank of an edge
   among the included edges of its vertex comes from one prefix sum over all
   positions) */
__global__ void k_compact_count(const uint32_t *row, const uint32_t *ipos,
                                const uint32_t *slot_v, uint32_t *cnt,
                                uint32_t nslots)
{
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  const uint32_t v = slot_v[s];
  cnt[s] = ipos[row[v + 1]] - ipos[row[v]];
}
__global__ void k_compact_fill(GtsGraphView G, const uint32_t *estart,
                               const uint8_t *incl, const uint32_t *ipos,
                               const uint32_t *slot_of, const uint32_t *slot_base,
                               const uint32_t *coff, const uint32_t *slot_comp,
                               uint8_t *comp_wide, uint8_t *comp_d32, uint32_t *cstart,
                               uint32_t *cend, int64_t *cdist, uint8_t *cflags,
                               uint32_t *cgpos, uint8_t *cstate, uint32_t *cmap)
{
  /* two edges a thread: the random reads of an edge (the slot of its end
     vertex, the record of its start vertex' slot) are a chain of three; two
     chains in flight hide half of it */
  constexpr int U = 2;
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x) * U + threadIdx.x;
  uint64_t p[U];
  bool in[U];
  uint32_t a[U], b[U], s[U], sb[U], ip[U], r0[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    p[k] = base + (uint64_t)k * blockDim.x;
    in[k] = p[k] < G.m && incl[p[k]];
    if (p[k] < G.m && !in[k]) cmap[p[k]] = GTS_NONE;
    a[k] = in[k] ? estart[p[k]] : 0u;
    b[k] = in[k] ? G.end[p[k]] : 0u;
    ip[k] = in[k] ? ipos[p[k]] : 0u;
  }
#pragma unroll
  for (int k = 0; k < U; ++k) { s[k] = slot_of[a[k]]; sb[k] = slot_of[b[k]]; r0[k] = G.row[a[k]]; }
#pragma unroll
  for (int k = 0; k < U; ++k) {
    if (!in[k]) continue;
    const uint32_t base_s = slot_base[s[k]];
    const uint32_t kk = coff[s[k]] + (ip[k] - ipos[r0[k]]);
    /* flags with the u-turn bit from the build.  An included edge's twin is live
       or the edge is live itself; GTS_F_TWINLIVE matters only while the edge is
       marked (d_arc) and is cleared with the twin's CYCLIC mark, the only way a
       live edge dies (mark_vertex_cyclic takes both): set for every edge. */
    const uint8_t f = G.flags[p[k]];
    cstart[kk] = s[k] - base_s; cend[kk] = sb[k] - base_s;
    const int64_t d = G.dist[p[k]];
    /* the LDS layout adds up to 4095 distances in 32 bits (nd_t) */
    if (d >= (1 << 19) || d <= -(1 << 19)) comp_wide[slot_comp[s[k]]] = 1;
    if (d > 32767 || d < -32768) comp_d32[slot_comp[s[k]]] = 1;   /* int16 distances in LDS otherwise */
    /* a u-turn pair or a self loop: the component has no strand assignment, its
       program takes the slow paths (ten times the time per contig) -- it is
       claimed before the others of the launch so that it cannot set the tail */
    cdist[kk] = d;
    cflags[kk] = (uint8_t)((f & 7u) | GTS_F_TWINLIVE);
    cgpos[kk] = (uint32_t)p[k]; cstate[kk] = G.state[p[k]];
    cmap[p[k]] = kk;
  }
}

/* gfx950 wave policy of gts_component.hpp */
struct GtsWave64 {
  static const uint32_t WIDTH = 64;
  static const bool TEAM = false;
  static __device__ __forceinline__ void and_bits(uint32_t *p, uint32_t m) { atomicAnd(p, m); }
  static __device__ __forceinline__ uint32_t clz64(uint64_t v) { return (uint32_t)__clzll((long long)v); }
  /* OR over the eight lanes of a group (lanes 8k .. 8k+7), result in all of
     them: two quad permutations and a half-row mirror, no LDS */
  static __device__ __forceinline__ uint32_t group8_or32(uint32_t x)
  {
    x |= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);    /* quad_perm [1,0,3,2] */
    x |= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);    /* quad_perm [2,3,0,1] */
    x |= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);   /* row_half_mirror */
    return x;
  }
  /* a counter other workgroups add to, as it is now (same value in every lane) */
  static __device__ __forceinline__ uint64_t peek(const unsigned long long *p)
  {
    const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint64_t)uni64((int64_t)v);
  }
  static __device__ __forceinline__ uint32_t group8_add32(uint32_t x)
  {
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);
    return x;
  }
  static __device__ __forceinline__ uint64_t group8_or(uint64_t x)
  {
    return (uint64_t)group8_or32((uint32_t)x) | (uint64_t)group8_or32((uint32_t)(x >> 32)) << 32;
  }
  /* opaque to the optimiser: everything a function derives from the lane number
     (group masks, lane-indexed addresses into a dozen arrays) would otherwise be
     hoisted out of the kernel's component loop and held in registers for the
     whole launch -- twenty of them in k_components_fast */
  static __device__ __forceinline__ uint32_t lane()
  {
    /* from the hardware (set bits of an all-ones mask below this lane), not from
       threadIdx.x: that register would have to stay live for the whole kernel */
    uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
  }
  static __device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
  static __device__ __forceinline__ uint32_t popc(uint64_t m) { return (uint32_t)__popcll(m); }
  /* set bits of m below the calling lane (l is always lane()): v_mbcnt_lo/hi */
  static __device__ __forceinline__ uint32_t popc_below(uint64_t m, uint32_t)
  { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
  static __device__ __forceinline__ uint32_t ctz(uint64_t m) { return (uint32_t)__ffsll((long long)m) - 1u; }
  static __device__ __forceinline__ uint32_t msb(uint64_t m) { return 63u - (uint32_t)__clzll((long long)m); }
  static __device__ __forceinline__ uint32_t shfl(uint32_t v, uint32_t l) { return (uint32_t)__shfl((int)v, (int)l); }
  static __device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t l)
  {
    const uint32_t lo = shfl((uint32_t)v, l), hi = shfl((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
  }
  static __device__ __forceinline__ uint32_t uni(uint32_t v)
  { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
  static __device__ __forceinline__ int64_t uni64(int64_t v)
  {
    const uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
  }
  static __device__ __forceinline__ void fence()
  { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  /* constant 100 MHz counter (s_memrealtime) */
  static __device__ __forceinline__ uint64_t clock() { return wall_clock64(); }
  static __device__ __forceinline__ void count(unsigned long long *p)
  { if (lane() == 0) atomicAdd(p, 1ull); }
  static __device__ __forceinline__ void count_n(uint32_t *p, uint32_t n) { atomicAdd(p, n); }
  static __device__ __forceinline__ void or_bits(uint32_t *p, uint32_t m) { atomicOr(p, m); }
  static __device__ __forceinline__ void add64(uint64_t *p, uint64_t n)
  { atomicAdd((unsigned long long *)p, (unsigned long long)n); }
  static __device__ __forceinline__ uint32_t clz32(uint32_t v) { return (uint32_t)__clz((int)v); }
  /* inclusive prefix sum of values < 128: one ballot per bit, the lower-lane
     population count of each ballot weighted by the bit */
  static __device__ __forceinline__ uint32_t scan_incl_small(uint32_t v)
  {
    const uint32_t l = lane();
    const uint64_t le = l == 63 ? ~0ull : (2ull << l) - 1ull;
    uint32_t s = 0;
#pragma unroll
    for (int b = 0; b < 7; ++b)
      s += (uint32_t)__popcll(__ballot((v >> b) & 1u) & le) << b;
    return s;
  }
  /* value of lane l, l wave-uniform: v_readlane instead of ds_bpermute */
  static __device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t l)
  { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
  static __device__ __forceinline__ float shflf(float v, uint32_t l) { return __shfl(v, (int)l); }
  static __device__ __forceinline__ uint64_t lanemask_lt(uint32_t l) { return (1ull << l) - 1ull; }
  static __device__ __forceinline__ uint64_t range_mask(uint32_t lo, uint32_t hi)
  {
    const uint64_t h = hi >= 64 ? ~0ull : (1ull << hi) - 1ull;
    const uint64_t l = lo >= 64 ? ~0ull : (1ull << lo) - 1ull;
    return h & ~l;
  }
  template <class P> static __device__ __forceinline__ uint32_t atomic_max(P p, uint32_t v)
  { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  /* the workgroup's job slot (GtsHelpJob, in LDS): words other wavefronts of the
     workgroup read and write */
  static __device__ __forceinline__ uint32_t hub_add(uint32_t *p, uint32_t v)
  { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  static __device__ __forceinline__ uint32_t hub_cas(uint32_t *p, uint32_t expect, uint32_t v)
  { __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return expect; }
  static __device__ __forceinline__ uint32_t hub_load(const uint32_t *p)
  { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  static __device__ __forceinline__ void hub_store(uint32_t *p, uint32_t v)
  { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  static __device__ __forceinline__ void hub_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
  static __device__ __forceinline__ void hub_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
  /* between a store and a load that must not change places (the gate / count handshake of a job) */
  static __device__ __forceinline__ void hub_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
  static __device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(8); }
  template <class T> static __device__ __forceinline__ uint32_t lds_addr(T __attribute__((address_space(3))) *p)
  { return (uint32_t)(uintptr_t)p; }
  /* lane 0 takes n entries from the pool, every lane gets the offset */
  static __device__ __forceinline__ uint64_t alloc(unsigned long long *used, uint64_t n)
  {
    unsigned long long o = 0;
    if (lane() == 0) o = atomicAdd(used, (unsigned long long)n);
    return (uint64_t)uni64((int64_t)o);
  }
};

/* one wavefront per component; `order` lists the components by decreasing
   LDS footprint, [first, first + count) is the slice of this launch.
   Global-memory variant: any component size. */
__global__ void __launch_bounds__(GTS_WAVE)
k_components(GtsCompView C, const uint32_t *order, uint32_t first, uint32_t count, int mode,
             int defer_global)
{
  if (blockIdx.x >= count) return;
  const uint32_t c = order[first + blockIdx.x];
  /* a component of this class defers its walks only on request (option
     "defer_global_components"): its tasks run from global memory,
     k_walk_tasks_global */
  if (!defer_global) { C.defer_min_nv = 0; C.defer_ref_min_nv = 0; }
  const GtsCompMem M = GtsComponent<GtsWave64>::global_mem(C, c);
  GtsComponent<GtsWave64> prog(C, M, c);
  prog.run(mode);
}

/* Global-memory variant with a team: a workgroup of GTS_TEAM_WAVES wavefronts
   per component.  Wavefront 0 runs the component program; the others wait for
   the ccs it posts (GtsTeamCtl) and sweep their share of the cc's walks.  Every
   wavefront meets every barrier: two per posted cc, one for the end. */
#define GTS_TEAM_WAVES 8u
/* LDS address (byte offset) of a generic pointer into the workgroup's LDS */
static __device__ __forceinline__ uint32_t lds_offset(const void *p)
{
  return (uint32_t)(uintptr_t)(const char __attribute__((address_space(3))) *)p;
}
struct GtsWave64Team : GtsWave64 {
  static const bool TEAM = true;
  static __device__ __forceinline__ void team_barrier() { __syncthreads(); }
  /* returning add on a word of the workgroup (LDS or global memory) */
  static __device__ __forceinline__ uint32_t team_add(uint32_t *p, uint32_t v)
  { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};
__global__ void __launch_bounds__(GTS_TEAM_WAVES * GTS_WAVE)
k_components_team(GtsCompView C, const uint32_t *order, uint32_t first, uint32_t count, int mode,
                  uint32_t lds_bytes)
{
  __shared__ GtsTeamCtl ctl;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.x >= count) return;
  const uint32_t c = order[first + blockIdx.x];
  const uint32_t wv = threadIdx.x / GTS_WAVE;
  C.defer_min_nv = 0; C.defer_ref_min_nv = 0;
  GtsCompMem M = GtsComponent<GtsWave64Team>::global_mem(C, c);
  uint32_t tl_vst = GTS_NONE, tl_queue = GTS_NONE, tl_scratch = GTS_NONE, tl_stv = GTS_NONE, tl_pbits = GTS_NONE, tl_pbits_bytes = 0;
  {
    /* the vertex-indexed arrays the traversals chase -- states, strands, the
       queue, degrees, the sweep order -- move to the workgroup's LDS as far as
       they fit (generic pointers: the program is the global-memory one); the
       lists stay in global memory (L2) */
    const uint32_t nv = M.nv, nv4 = ((nv * 4u + 15u) / 16u) * 16u, nv1 = ((nv + 15u) / 16u) * 16u;
    uint32_t off = 0;
    if (off + nv1 <= lds_bytes) {
      uint8_t *p = (uint8_t *)(smem + off); off += nv1;
      for (uint32_t s = threadIdx.x; s < nv; s += blockDim.x) p[s] = M.vst[s];
      M.vst = p; tl_vst = lds_offset(p);
    }
    if (off + nv1 <= lds_bytes) { M.gorient = (uint8_t *)(smem + off); off += nv1; }
    /* the edge scratch of the terminal search (calc_cc_team), its queue and the
       degrees of peel in one piece: free while the walks of a cc are made, when it
       holds the walks' position bitmaps */
    const uint32_t spare0 = off;
    if (off + 64u * GTS_TCC_K * 8u <= lds_bytes) { tl_scratch = lds_offset(smem + off); off += 64u * GTS_TCC_K * 8u; }
    if (off + nv4 <= lds_bytes) { M.queue = (uint32_t *)(smem + off); tl_queue = lds_offset(M.queue); off += nv4; }
    if (off + nv4 <= lds_bytes) { M.st_v = (uint32_t *)(smem + off); tl_stv = lds_offset(M.st_v); off += nv4; }
    tl_pbits = lds_offset(smem + spare0); tl_pbits_bytes = off - spare0;
    if (off + nv4 <= lds_bytes) { M.topo = (uint32_t *)(smem + off); off += nv4; }
    if (off + nv4 <= lds_bytes) { M.tpos = (uint32_t *)(smem + off); off += nv4; }
  }
  if (threadIdx.x == 0) {
    const uint64_t need = (uint64_t)GTS_TEAM_WAVES * GtsComponent<GtsWave64Team>::team_wave_bytes(M.nv);
    const unsigned long long off = atomicAdd(C.team_used, (unsigned long long)need);
    ctl.slab = off; ctl.slab_ok = off + need <= C.team_cap ? 1u : 0u;
    ctl.kind = 0;
  }
  __syncthreads();
  const bool ok = ctl.slab_ok != 0;
  GtsComponent<GtsWave64Team> prog(C, M, c);
  if (ok) {
    prog.team = &ctl; prog.team_base = C.team_slab + ctl.slab;
    prog.team_wave = wv; prog.team_waves = GTS_TEAM_WAVES;
  }
  prog.tl_vst = tl_vst; prog.tl_queue = tl_queue; prog.tl_scratch = tl_scratch; prog.tl_stv = tl_stv;
  prog.tl_pbits = tl_pbits; prog.tl_pbits_bytes = tl_pbits_bytes;
  if (wv == 0) {
    prog.run(mode);
    if (ok) {
      if (threadIdx.x == 0) ctl.kind = 0;
      __syncthreads();
    }
  } else if (ok) {
    prog.clean = true;     /* only clean components post walks */
    for (;;) {
      __syncthreads();
      if (ctl.kind == 0) break;
      if (ctl.kind == 2) (void)prog.peel_team_run();
      else prog.team_share(ctl.tb, ctl.te);
      __syncthreads();
    }
  }
}

/* LDS-resident variant: the launcher guarantees gts_comp_lds_bytes(nv, ne) <=
   the dynamic LDS size of the launch.  The wavefront stages the component's
   compact graph and vertex states with coalesced loads, initialises the
   scratch in LDS and runs the same program on LDS base pointers; marks still
   go to the global graph as they are set, vertex states are written back by
   run(). */
typedef char __attribute__((address_space(3))) *gts_lds_cursor;
template <typename T>
__device__ __forceinline__ T __attribute__((address_space(3))) *lds_carve(gts_lds_cursor &p, uint32_t count)
{
  T __attribute__((address_space(3))) *r = (T __attribute__((address_space(3))) *)p;
  p += ((count * (uint32_t)sizeof(T) + 15u) / 16u) * 16u;
  return r;
}
/* stages component c into the workgroup's LDS (packed layout) and points M at
   it; with_analysis also loads the strands / sweep order a deferred walk needs */
__device__ __forceinline__ void stage_component(const GtsCompView &C, uint32_t c, char *smem,
                                                GtsCompMemT<true> &M, bool with_analysis, uint32_t avail)
{
  const GtsCompMem G0 = GtsComponent<GtsWave64>::global_mem(C, c);
  const uint32_t nv = G0.nv, ne = G0.ne, lane = GtsWave64::lane();
  gts_lds_cursor p = (gts_lds_cursor)smem;
  typedef GtsCompMemT<true>::idx_t idx_t;
  M.nv = nv; M.ne = ne; M.e0 = 0;
  auto coff = lds_carve<idx_t>(p, nv + 1);
  M.ccoff = lds_carve<idx_t>(p, nv + 1);
  M.term = lds_carve<idx_t>(p, nv); M.cc_best = lds_carve<idx_t>(p, nv);
  M.topo = lds_carve<idx_t>(p, nv); M.tpos = lds_carve<idx_t>(p, nv);
  auto cseq = lds_carve<int32_t>(p, nv);
  M.vst = lds_carve<uint8_t>(p, nv); M.st_dir = lds_carve<uint8_t>(p, nv);
  M.tight = lds_carve<uint8_t>(p, nv); M.gorient = lds_carve<uint8_t>(p, nv);
  auto cend = lds_carve<idx_t>(p, ne);
  const bool d32 = C.comp_d32[c] != 0;
  int32_t __attribute__((address_space(3))) *cdist = nullptr;
  int16_t __attribute__((address_space(3))) *cdist16 = nullptr;
  if (d32) cdist = lds_carve<int32_t>(p, ne); else cdist16 = lds_carve<int16_t>(p, ne);
  auto cfs = lds_carve<uint8_t>(p, ne);   /* flags | state << 4 | marked << 7 */
  /* the walk scratch, in one piece (gts_comp_lds_bytes): slot 0 of the batched
     walks is distmap, plen, edgemap, par */
  M.wbase = (char __attribute__((address_space(3))) *)p;
  M.distmap = lds_carve<float>(p, nv); M.plen = lds_carve<uint32_t>(p, nv);
  M.edgemap = lds_carve<idx_t>(p, nv); M.par = lds_carve<idx_t>(p, nv);
  M.nd = lds_carve<int32_t>(p, nv);
  M.queue = lds_carve<idx_t>(p, nv); M.visited = lds_carve<idx_t>(p, nv);
  M.st_v = lds_carve<idx_t>(p, nv); M.wterm = lds_carve<idx_t>(p, nv);
  M.st_par = lds_carve<idx_t>(p, nv);
  M.sarc = nullptr; M.smid = nullptr;
  idx_t __attribute__((address_space(3))) *sarc = nullptr, *smid = nullptr;
  {
    const uint32_t need = (uint32_t)(p - (gts_lds_cursor)smem);
    if (with_analysis) {
      /* a walk task: what the launch has behind the footprint holds the arcs
         split by sense for the reference's search (GtsCompMemT::sarc) */
      const uint32_t vb = ((ne * 2u + 15u) / 16u) * 16u + (((nv + 1u) * 2u + 15u) / 16u) * 16u;
      M.wslots = 2u;
      if (avail >= need + vb) { sarc = lds_carve<idx_t>(p, ne); smid = lds_carve<idx_t>(p, nv + 1); }
    } else {
      const uint32_t more = avail > need ? (avail - need) / gts_walk_slot_bytes(nv) : 0u;
      M.wslots = 2u + more < GTS_WALK_SLOTS_MAX ? 2u + more : GTS_WALK_SLOTS_MAX;
    }
  }
  /* never live at the same time (gts_comp_lds_bytes) */
  M.st_cur = M.cc_best; M.touched = M.visited;
  M.lastpop = (uint32_t __attribute__((address_space(3))) *)M.nd;
  M.cflags.b = cfs; M.cstate.b = cfs;
  M.coff = coff; M.cseq = cseq; M.cend = cend; M.cdist = cdist; M.cdist16 = cdist16; M.d16 = !d32;
  M.cstart.coff = coff; M.cstart.nv = nv;
  for (uint32_t i = lane; i <= nv; i += GTS_WAVE) coff[i] = (idx_t)(G0.coff[i] - G0.e0);
  for (uint32_t i = lane; i < nv; i += GTS_WAVE) {
    cseq[i] = (int32_t)G0.cseq[i]; M.vst[i] = G0.vst[i];
    M.distmap[i] = GTS_DIST_UNSET; M.st_dir[i] = 0; M.tight[i] = 0;
    if (with_analysis) {
      M.gorient[i] = G0.gorient[i]; M.topo[i] = (idx_t)G0.topo[i]; M.tpos[i] = (idx_t)G0.tpos[i];
    }
  }
  for (uint32_t i = lane; i < ne; i += GTS_WAVE) {
    cend[i] = (idx_t)G0.cend[i];
    if (d32) cdist[i] = (int32_t)G0.cdist[i]; else cdist16[i] = (int16_t)G0.cdist[i];
    const uint8_t st = G0.cstate[i];
    cfs[i] = (uint8_t)((G0.cflags[i] & 15u) | (st << 4) | (gts_edge_is_marked(st) ? 0x80u : 0u));   /* GtsLdsStateRef */
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (sarc) {
    /* stable partition of every list by sense, a lane per vertex */
    for (uint32_t v = lane; v < nv; v += GTS_WAVE) {
      const uint32_t b = coff[v], e = coff[v + 1];
      uint32_t k = b;
      for (uint32_t ce = b; ce < e; ++ce) if (cfs[ce] & GTS_F_SENSE) sarc[k++] = (idx_t)ce;
      smid[v] = (idx_t)k;
      for (uint32_t ce = b; ce < e; ++ce) if (!(cfs[ce] & GTS_F_SENSE)) sarc[k++] = (idx_t)ce;
    }
    M.sarc = sarc; M.smid = smid;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}
__global__ void __launch_bounds__(GTS_WAVE)
k_components_lds(GtsCompView C, const uint32_t *order, uint32_t first, uint32_t count, int mode,
                 uint32_t lds_bytes)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.x >= count) return;
  const uint32_t c = order[first + blockIdx.x];
  GtsCompMemT<true> M;
  stage_component(C, c, smem, M, false, lds_bytes);
  GtsComponent<GtsWave64, true> prog(C, M, c);
  prog.run(mode);
}

/* ---- all LDS components in ONE launch: a pool of wavefronts per CU -----------
   One workgroup of GTS_POOL_WAVES wavefronts per CU owns (almost) all of the
   CU's LDS as a pool of GTS_POOL_PAGE-byte pages.  Every wavefront works on its
   own: it claims the next component from the list sorted by decreasing
   footprint -- from the FRONT (largest first) or, while another wavefront of
   the workgroup is waiting for room for a front component, from the BACK
   (smallest first) --, allocates the component's pages, stages it, runs the
   component program, frees the pages and claims again.  Large components
   start first (their wave time is the critical path), the LDS they leave and
   the wave slots they do not use are filled with small ones: the CU is bound
   by LDS at the start and by wave slots at the end, not by one of them after
   the other as a launch per size class is.

   Claims: one 64-bit counter, a front claim of B adds B, a fill ("back") claim
   B * 2^32; the value before the add (h, t) gives the indices h .. h+B-1 below
   g0 resp. g0+t .. g0+t+B-1 below count, g0 = the first component of at most
   two pages -- every index is handed out once.  Both cursors move towards the
   smaller components (round 2: the back cursor came up from the smallest and
   the two met among components of 50 - 80 contigs, where one that is not clean
   takes milliseconds: they set a tail of 1.4 ms).  The workgroup claims
   for its wavefronts and keeps the indices in stock (a single counter hit
   once per component by 4096 wavefronts is what the launch would wait for:
   same-address atomics across the XCDs run at a few tens per microsecond):
   one at a time among the `nbig` largest, GTS_POOL_BATCH at a time after that
   and from the back.  A wavefront that finds stock and both ends empty leaves
   (every wavefront gets there: the exit condition).
   A waiting wavefront holds no pages, and what it waits for is released by
   wavefronts that run to completion, so it gets its turn; while it waits, back
   components may only take pages above the ones it needs. */
#define GTS_POOL_PAGE 1536u
#define GTS_POOL_PAGES 104u                     /* 156 KB: one workgroup per CU (2 KB pages: 25 % more of a small component's last page wasted) */
#define GTS_POOL_BYTES (GTS_POOL_PAGES * GTS_POOL_PAGE)
#define GTS_POOL_BATCH 8u
/* k_components_fast (round 4): TWO workgroups per CU, half the pages each */
#define GTS_FAST_PAGES 52u
#define GTS_FAST_BYTES (GTS_FAST_PAGES * GTS_POOL_PAGE)
/* the cold list (device words, u64): components the fast program hands to the full one */
enum { GTS_COLD_HEAD = 0, GTS_COLD_TAIL = 1, GTS_COLD_DONE = 2, GTS_COLD_PRODUCERS = 3, GTS_COLD_NSEED = 4,
       GTS_COLD_WORDS = 8 };
struct GtsPoolCtl {
  uint32_t used[4];        /* page bitmap (16-byte aligned: read in one piece) */
  uint32_t exited;         /* k_components_fast: wavefronts of the workgroup that have left */
  uint32_t seed_round;     /* cold mode: seeded components this workgroup has taken */
  uint32_t lock;
  uint32_t front_busy;     /* a wavefront holds the front role: a front claim it has no pages for yet */
  uint32_t wait_pages;     /* pages that wavefront needs (0: it is not waiting) */
  uint32_t f_done, b_done; /* the cursor has nothing more to give */
  uint32_t f_last_end;     /* end of the last front refill (claims among the nbig largest come one at a time) */
  /* stock of indices: next | end << 32.  A claim is one returning 64-bit LDS add
     (round 4: claims under the workgroup's lock -- a dozen dependent LDS round
     trips each, twice per component -- kept the lock busy 80 % of the time and
     cost a wavefront 11 us per component); the lock is taken to refill */
  unsigned long long f_stock, b_stock;
  /* statistics, summed here and added to the launch's words by the workgroup's last
     wavefront (kept out of the wavefronts' registers: they would be live across
     every component program) */
  unsigned long long t_run, t_wait, t_life, t_claim;
  unsigned long long n_done, n_walks, n_cold, b_done_bytes, b_cold_bytes;
  unsigned long long t_begin[GTS_POOL_WAVES];
  unsigned long long n_helped, t_helping;   /* statistics: joins of an open job, ticks spent there */
  GtsHelpJob job;          /* walks of a cc over the workgroup's wavefronts (gts_component.hpp) */
};
/* the page bitmap as one integer (bit q = page q in use) */
typedef unsigned __int128 gts_pool_bits;
__device__ __forceinline__ gts_pool_bits pool_bits_load(const volatile uint32_t *used)
{
  typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
  const u32x4 w = *(const volatile u32x4 *)used;     /* first member of the control block: 16-byte aligned */
  return (gts_pool_bits)w.x | (gts_pool_bits)w.y << 32 | (gts_pool_bits)w.z << 64 | (gts_pool_bits)w.w << 96;
}
/* pages are taken under the lock and given back without it: the bits of a run
   are set / cleared with LDS atomics, so a free never collides with the search
   of the wavefront that holds the lock (it may miss pages freed meanwhile) */
__device__ __forceinline__ void pool_bits_set(uint32_t *used, gts_pool_bits m)
{
#pragma unroll
  for (int k = 0; k < 4; ++k) { const uint32_t w = (uint32_t)(m >> (32 * k)); if (w) atomicOr(&used[k], w); }
}
__device__ __forceinline__ void pool_bits_clear(uint32_t *used, gts_pool_bits m)
{
#pragma unroll
  for (int k = 0; k < 4; ++k) { const uint32_t w = (uint32_t)(m >> (32 * k)); if (w) atomicAnd(&used[k], ~w); }
}
__device__ __forceinline__ gts_pool_bits pool_run_mask(uint32_t pos, uint32_t n)
{
  return (((gts_pool_bits)1 << n) - 1) << pos;    /* n <= 104 */
}
/* first fit from below (front) or from above (fill, not below `floor`); the
   search runs on registers: lane 0, lock held.  The starts of the free runs of
   n pages by doubling -- r &= r >> t leaves the bits whose next t neighbours are
   free as well --: seven steps instead of a walk over the positions */
__device__ __forceinline__ uint32_t pool_find(gts_pool_bits used, uint32_t n, bool from_below, uint32_t floor,
                                              uint32_t pages)
{
  if (floor > pages || n > pages - floor || n == 0) return GTS_NONE;
  gts_pool_bits r = ~used & pool_run_mask(0, pages);
  for (uint32_t have = 1; have < n;) {
    const uint32_t t = have < n - have ? have : n - have;
    r &= r >> t;
    have += t;
  }
  if (floor) r &= ~pool_run_mask(0, floor);
  if (!r) return GTS_NONE;
  const uint64_t lo = (uint64_t)r, hi = (uint64_t)(r >> 64);
  if (from_below) return lo ? (uint32_t)__ffsll((long long)lo) - 1u : 64u + (uint32_t)__ffsll((long long)hi) - 1u;
  return hi ? 127u - (uint32_t)__clzll((long long)hi) : 63u - (uint32_t)__clzll((long long)lo);
}
/* Every wait of the pool is bounded by the wall clock (100 MHz counter;
   `limit` ticks, ten seconds by default -- a count of spins would depend on the
   clock the chip happens to run at): a wavefront that runs into the bound counts
   the place in pstat[6 + where] and does NOT enter what it was waiting for -- no
   claim, no pages, a lock not taken means a critical section not run --, so the
   control block stays consistent, the launch ends, and the host restores the
   states from its snapshot and reports GTSG_EINTERNAL. */
__device__ __forceinline__ bool pool_lock(GtsPoolCtl *ctl, unsigned long long *pstat, uint64_t limit)
{
  uint32_t spins = 0;
  uint64_t t0 = 0;
  bool got = true;
  while (atomicCAS(&ctl->lock, 0u, 1u) != 0u) {
    if (spins == 0) t0 = GtsWave64::clock();
    if ((spins++ & 255u) == 0 && GtsWave64::clock() - t0 >= limit) { atomicAdd(pstat + 6, 1ull); got = false; break; }
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return got;
}
__device__ __forceinline__ void pool_unlock(GtsPoolCtl *ctl)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  atomicExch(&ctl->lock, 0u);
}
/* The launch's arguments besides the view (one struct: the fast and the full
   kernel take the same) */
struct GtsPoolArgs {
  const uint32_t *order, *order_key;   /* components by decreasing footprint; ~footprint */
  uint32_t first, count;               /* the launch's slice of them */
  int mode;
  unsigned long long *cursor, *pstat;
  unsigned long long *ptot;            /* 3 words: totals of the full program's lean statistics */
  uint32_t nbig, g0;
  int poison;
  uint64_t wait_limit;
  /* cold list (or null): u64 words GTS_COLD_*, entries (footprint << 32 | component + 1).
     The fast kernel appends; the full kernel in cold mode takes its components
     from it, in order, until every producer has left and the list is empty */
  unsigned long long *cold, *cold_list;
};
/* the leading components of the slice -- those whose footprint does not fit a
   fast workgroup's pool -- go to the cold list before the launches; the fast
   kernel's slice starts behind them */
__global__ void k_cold_seed(const uint32_t *order, const uint32_t *order_key, uint32_t first, uint32_t count,
                            unsigned long long *cold, unsigned long long *cold_list, uint32_t producers,
                            uint32_t fast_bytes)
{
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) {
    uint32_t lo = 0, hi = count;     /* first index whose footprint fits (keys ascending = footprints descending) */
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (~order_key[first + mid] > fast_bytes - 16u) lo = mid + 1; else hi = mid;
    }
    s_n = lo;
    cold[GTS_COLD_HEAD] = lo; cold[GTS_COLD_TAIL] = lo; cold[GTS_COLD_DONE] = 0;   /* (the seeds are dealt by position) */
    cold[GTS_COLD_PRODUCERS] = producers; cold[GTS_COLD_NSEED] = lo;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < s_n; i += blockDim.x)
    cold_list[i] = (unsigned long long)(~order_key[first + i]) << 32 | (unsigned long long)(order[first + i] + 1u);
}

/* The control block's address, opaque to the optimiser: its fields are then
   reached as base + immediate offset.  With the address known (a static
   __shared__ object) every field's address is a constant of its own in a vector
   register, hoisted out of the kernel's loop: a dozen registers held for the
   whole launch. */
__device__ __forceinline__ GtsPoolCtl *pool_ctl_opaque(GtsPoolCtl *p)
{
  typedef GtsPoolCtl __attribute__((address_space(3))) *lds_ctl;
  uint32_t a = (uint32_t)(uintptr_t)(lds_ctl)p;
  asm volatile("" : "+v"(a));
  return (GtsPoolCtl *)(lds_ctl)(uintptr_t)a;
}

/* scratch of a wavefront that walks for another one's component (GtsHelpJob): what
   create_walk_clean / create_walk_fast / walk_cyclic write -- labels, tree lengths,
   edgemap, parents, integer labels, three queues, degrees, depths, the best path,
   strands and dirty flags: 30 bytes a contig */
__device__ __forceinline__ uint32_t help_scratch_bytes(uint32_t nv)
{
  const uint32_t p4 = ((nv * 4u + 15u) / 16u) * 16u, p2 = ((nv * 2u + 15u) / 16u) * 16u, p1 = ((nv + 15u) / 16u) * 16u;
  return 3u * p4 + 8u * p2 + 2u * p1;
}
/* the owner's view with the scratch arrays replaced by the helper's own */
__device__ __forceinline__ void help_view(const GtsHelpJob *J, char *scratch, GtsCompMemT<true> &M)
{
  typedef GtsCompMemT<true>::idx_t idx_t;
  M = J->M;
  /* wave-uniform again (the copy came through vector loads) */
#define GTS_UNI_PTR(f) M.f = (decltype(M.f))(uintptr_t)GtsWave64::uni((uint32_t)(uintptr_t)M.f)
  GTS_UNI_PTR(coff); GTS_UNI_PTR(cend); GTS_UNI_PTR(cdist); GTS_UNI_PTR(cdist16); GTS_UNI_PTR(cseq); GTS_UNI_PTR(vst);
  GTS_UNI_PTR(term); GTS_UNI_PTR(ccoff); GTS_UNI_PTR(gorient); GTS_UNI_PTR(topo); GTS_UNI_PTR(tpos);
  GTS_UNI_PTR(cflags.b); GTS_UNI_PTR(cstate.b); GTS_UNI_PTR(cstart.coff);
#undef GTS_UNI_PTR
  M.nv = GtsWave64::uni(M.nv); M.ne = GtsWave64::uni(M.ne); M.e0 = GtsWave64::uni(M.e0);
  M.cstart.nv = M.nv; M.d16 = GtsWave64::uni((uint32_t)M.d16) != 0;
  const uint32_t nv = M.nv;
  gts_lds_cursor p = (gts_lds_cursor)scratch;
  M.distmap = lds_carve<float>(p, nv); M.plen = lds_carve<uint32_t>(p, nv); M.nd = lds_carve<int32_t>(p, nv);
  M.edgemap = lds_carve<idx_t>(p, nv); M.par = lds_carve<idx_t>(p, nv); M.queue = lds_carve<idx_t>(p, nv);
  M.visited = lds_carve<idx_t>(p, nv); M.st_v = lds_carve<idx_t>(p, nv); M.wterm = lds_carve<idx_t>(p, nv);
  M.st_par = lds_carve<idx_t>(p, nv); M.cc_best = lds_carve<idx_t>(p, nv);
  M.st_dir = lds_carve<uint8_t>(p, nv); M.tight = lds_carve<uint8_t>(p, nv);
  M.st_cur = M.cc_best; M.touched = M.visited;
  M.lastpop = (uint32_t __attribute__((address_space(3))) *)M.nd;
  M.wbase = nullptr; M.wslots = 0; M.sarc = nullptr; M.smid = nullptr;
  const uint32_t lane = GtsWave64::lane();
  for (uint32_t i = lane; i < nv; i += GTS_WAVE) { M.distmap[i] = GTS_DIST_UNSET; M.st_dir[i] = 0; M.tight[i] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

/* The pool kernels take their view as a pointer to a copy in device memory, and read it
   through a pointer the optimiser cannot see through: the ~100 pointers of a GtsCompView
   are then loaded where they are used (scalar loads, the scalar cache holds them).  As
   a by-value kernel argument they are loop invariants of the component loop:
   they are all loaded at the kernel's entry and kept for the whole launch -- 200 scalar
   registers more than there are, spilled into the lanes of nine vector registers that
   every wavefront then carries through every component program. */
template <class T>
__device__ __forceinline__ const T &opaque_const(const T *q)
{
  typedef const T __attribute__((address_space(4))) *const_ptr;
  const_ptr p = (const_ptr)q;
  asm volatile("" : "+s"(p));
  return *(const T *)p;
}
__device__ __forceinline__ const GtsCompView &opaque_view(const GtsCompView *C)
{
  typedef const GtsCompView __attribute__((address_space(4))) *const_ptr;   /* not written during the launch */
  const_ptr p = (const_ptr)C;
  asm volatile("" : "+s"(p));
  return *(const GtsCompView *)p;
}

/* FAST: the clean program on half a CU's pool (k_components_fast); else the full
   program (k_components_pool), which in cold mode (A.cold) claims from the cold
   list instead of the sorted one */
template <bool FAST, uint32_t PAGES>
__device__ __forceinline__ void pool_body(const GtsCompView *C0, const GtsPoolArgs *A0, char *smem, GtsPoolCtl *ctl0)
{
  GtsPoolCtl *ctl = ctl0;
  const GtsPoolArgs &A = opaque_const(A0);
  /* pstat (100 MHz ticks, summed over the wavefronts): [0] staging + program,
     [1] waiting for pages, [2] whole life of the wavefront; [3] first exit,
     [4] last exit (since the first wavefront's start, [5]); [6..8] waits that
     ran into their bound (lock, claim, pages); [9] programs that wrote past
     their footprint; fast kernel: [10] components it finished, [11] their walks,
     [12] components it handed to the cold list, [13] / [14] bytes of the former /
     the latter */
  unsigned long long *const pstat = A.pstat;
  const uint64_t wait_limit = A.wait_limit;
  if (threadIdx.x == 0) {
    ctl->exited = 0; ctl->seed_round = 0;
    ctl->t_run = ctl->t_wait = ctl->t_life = ctl->t_claim = 0;
    ctl->n_done = ctl->n_walks = ctl->n_cold = ctl->b_done_bytes = ctl->b_cold_bytes = 0;
    ctl->lock = 0; ctl->front_busy = 0; ctl->wait_pages = 0;
    ctl->used[0] = ctl->used[1] = ctl->used[2] = ctl->used[3] = 0;
    ctl->f_done = ctl->b_done = 0; ctl->f_last_end = 0; ctl->f_stock = 0; ctl->b_stock = 0;
    ctl->n_helped = ctl->t_helping = 0;
    ctl->job.seq = 0; ctl->job.ready = 0; ctl->job.active = 0; ctl->job.n_running = 0; ctl->job.walks = 0;
  }
  __syncthreads();
  const uint32_t my_wave = GtsWave64::uni(threadIdx.x / GTS_WAVE);   /* (scalar: threadIdx.x itself is not needed again) */
  if (GtsWave64::lane() == 0) ctl->t_begin[my_wave] = GtsWave64::clock();
  const bool cold_mode = !FAST && A.cold != nullptr;
  /* the fast kernel's slice starts behind the components seeded to the cold list */
  uint32_t first = A.first, count = A.count, nbig = A.nbig, g0 = A.g0;
  if (A.cold) {
    const uint32_t ns = GtsWave64::uni((uint32_t)A.cold[GTS_COLD_NSEED]);   /* (uniform: kept in scalar registers) */
    first += ns; count -= ns;
    nbig = nbig > ns ? nbig - ns : 0u;
    g0 = g0 > ns ? g0 - ns : 0u;
  }
  /* joins the workgroup's open job, if there is one with terminals left and the pool has
     pages for this wavefront's scratch: walks for it, leaves its result, waits for the
     owner to close (it may copy the path out of these pages) and gives the pages back */
  auto try_help = [&]() -> bool {
    if constexpr (FAST) return false;
    else {
      const GtsCompView &C = opaque_view(C0);
      GtsPoolCtl *const ctl = pool_ctl_opaque(ctl0);   /* (its own: through the captured one the accesses turn generic) */
      GtsHelpJob *J = &ctl->job;
      /* (the slot's words as LDS words: through the generic pointer every look at them
         would go through the flat aperture, whose base the compiler then keeps in a
         vector register for the whole kernel) */
      typedef GtsHelpJob __attribute__((address_space(3))) *lds_job;
      const lds_job J3 = (lds_job)J;
      auto ld = [](const uint32_t __attribute__((address_space(3))) *p) -> uint32_t {
        return GtsWave64::uni(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); };
      const uint32_t s = ld(&J3->seq);
      if (!(s & 1u) || ld(&J3->ready) != s) return false;
      GtsWave64::hub_acquire();
      const uint32_t te = ld(&J3->te);
      if (ld(&J3->next) >= te) return false;
      const uint32_t hl = GtsWave64::lane();
      const uint64_t th0 = GtsWave64::clock();
      const uint32_t hp = (help_scratch_bytes(ld(&J3->nv)) + GTS_POOL_PAGE - 1u) / GTS_POOL_PAGE;
      uint32_t hpos = GTS_NONE;
      if (hl == 0 && pool_lock(ctl, pstat, wait_limit)) {      /* one try: no waiting for pages */
        volatile GtsPoolCtl *v = (volatile GtsPoolCtl *)ctl;
        hpos = pool_find(pool_bits_load(v->used), hp, false, v->wait_pages, PAGES);
        if (hpos != GTS_NONE) pool_bits_set(ctl->used, pool_run_mask(hpos, hp));
        pool_unlock(ctl);
        if (hpos != GTS_NONE) {
          __hip_atomic_fetch_add(&J3->active, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          GtsWave64::hub_fence();
          /* (closing -- the owner has shut `ready` and is waiting for the count --, closed,
             or closed and opened again meanwhile: not this one) */
          if (__hip_atomic_load(&J3->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != s ||
              __hip_atomic_load(&J3->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != s) {
            __hip_atomic_fetch_add(&J3->active, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pool_bits_clear(ctl->used, pool_run_mask(hpos, hp));
            hpos = GTS_NONE;
          }
        }
      }
      hpos = GtsWave64::uni(hpos);
      if (hpos == GTS_NONE) return false;
      {
        GtsCompMemT<true> Mh;
        help_view(J, smem + hpos * GTS_POOL_PAGE, Mh);
        GtsComponent<GtsWave64, true> hprog(C, Mh, ld(&J3->comp));
        hprog.clean = ld(&J3->clean) != 0;
        hprog.hub_me = my_wave;
        hprog.hub_take_walks(J, te);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (hl == 0) {
        GtsWave64::hub_release();
        __hip_atomic_fetch_add(&J3->active, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&J3->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == s) GtsWave64::nap();
        pool_bits_clear(ctl->used, pool_run_mask(hpos, hp));
        atomicAdd(&ctl->n_helped, 1ull); atomicAdd(&ctl->t_helping, (unsigned long long)(GtsWave64::clock() - th0));
      }
      return true;
    }
  };
  uint32_t lingered = 0;
  for (;;) {
    ctl = pool_ctl_opaque(ctl0);
    const uint32_t lane = GtsWave64::lane();   /* (inside the loop: nothing derived from it is held across components) */
    const GtsCompView &C = opaque_view(C0);
    const GtsPoolArgs &A = opaque_const(A0);   /* (shadows the one the set-up above read) */
    if (C.help_walks) try_help();
    /* role, claim, pages: lane 0; the rest of the wavefront waits at the broadcast */
    uint32_t idx = GTS_NONE, pos = 0, npages = 0, need = 0, comp = GTS_NONE;
    if (lane == 0) {
      bool front = false;
      const uint64_t tc0 = GtsWave64::clock();
      /* a claim from the sorted list.  0: idx; 1: nothing for this wavefront now, but
         the front (held by the one that waits for pages) has; 2: nothing left;
         3: the lock was not to be had (this wavefront leaves) */
      auto claim_sorted = [&](bool may_front) -> int {
        volatile GtsPoolCtl *v = (volatile GtsPoolCtl *)ctl;
        /* one index off a stock, or GTS_NONE when it is empty */
        auto pop = [&](unsigned long long *stock) -> uint32_t {
          const unsigned long long old = atomicAdd(stock, 1ull);
          const uint32_t next = (uint32_t)old, end = (uint32_t)(old >> 32);
          return next < end ? next : GTS_NONE;
        };
        auto empty = [&](const unsigned long long *stock) -> bool {
          const unsigned long long cur = *(volatile const unsigned long long *)stock;
          return (uint32_t)cur >= (uint32_t)(cur >> 32);
        };
        /* the front role: one wavefront of the workgroup at a time, from its claim
           until it has its pages */
        if (may_front && !(v->f_done && empty(&ctl->f_stock)) && atomicCAS(&ctl->front_busy, 0u, 1u) == 0u) {
          for (;;) {
            idx = pop(&ctl->f_stock);
            if (idx != GTS_NONE) { front = true; return 0; }
            if (v->f_done) break;
            if (!pool_lock(ctl, pstat, wait_limit)) { atomicExch(&ctl->front_busy, 0u); return 3; }
            if (empty(&ctl->f_stock) && !v->f_done) {
              const uint32_t want = v->f_last_end < nbig ? 1u : GTS_POOL_BATCH;
              const unsigned long long old = atomicAdd(A.cursor, (unsigned long long)want);
              const uint64_t h = old & 0xFFFFFFFFull;
              const uint64_t lim = g0;                                              /* the fill cursor holds [g0, count) */
              if (h < lim) {
                const uint32_t fe = (uint32_t)(h + want < lim ? h + want : lim);
                v->f_last_end = fe;
                atomicExch(&ctl->f_stock, (unsigned long long)fe << 32 | (unsigned long long)h);
              } else v->f_done = 1;
            }
            pool_unlock(ctl);
          }
          atomicExch(&ctl->front_busy, 0u);   /* nothing left at the front */
        }
        for (;;) {
          idx = pop(&ctl->b_stock);
          if (idx != GTS_NONE) return 0;
          if (v->b_done) break;
          if (!pool_lock(ctl, pstat, wait_limit)) return 3;
          if (empty(&ctl->b_stock) && !v->b_done) {
            const unsigned long long old = atomicAdd(A.cursor, (unsigned long long)GTS_POOL_BATCH << 32);
            const uint64_t t = old >> 32, at = (uint64_t)g0 + t;
            const uint64_t avail = at < (uint64_t)count ? (uint64_t)count - at : 0;
            if (avail) {
              const uint64_t be = at + (avail < GTS_POOL_BATCH ? avail : GTS_POOL_BATCH);
              atomicExch(&ctl->b_stock, be << 32 | at);
            } else v->b_done = 1;
          }
          pool_unlock(ctl);
        }
        /* nothing for this wavefront now; 1: but the front (held by another one) has */
        return (may_front && !(v->f_done && empty(&ctl->f_stock))) ? 1 : 2;
      };
      if (cold_mode) {
        /* in this order: the workgroup's share of the seeded components (the
           largest of the launch: dealt round-robin, so that every workgroup's LDS
           starts with one of them), what the fast kernel has handed over, and --
           rather than idle -- small components from the back of the sorted list */
        const uint32_t nseed = (uint32_t)A.cold[GTS_COLD_NSEED];
        for (uint32_t spins = 0;; ++spins) {
          unsigned long long ent = 0;
          if (pool_lock(ctl, pstat, wait_limit)) {
            volatile GtsPoolCtl *v = (volatile GtsPoolCtl *)ctl;
            const uint32_t s = blockIdx.x + v->seed_round * gridDim.x;
            if (s < nseed) { v->seed_round = v->seed_round + 1u; ent = A.cold_list[s]; }
            pool_unlock(ctl);
          } else break;
          if (!ent) {
            const unsigned long long h = __hip_atomic_load(&A.cold[GTS_COLD_HEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t = __hip_atomic_load(&A.cold[GTS_COLD_TAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h < t) {
              if (atomicCAS(&A.cold[GTS_COLD_HEAD], h, h + 1ull) != h) continue;
              /* the entry is stored right after the tail moved: a few polls at most */
              for (uint32_t k = 0; !ent; ++k) {
                ent = __hip_atomic_load(&A.cold_list[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!ent && (k & 255u) == 255u && GtsWave64::clock() - tc0 >= wait_limit) break;
              }
              if (!ent) { atomicAdd(pstat + 7, 1ull); break; }
            }
          }
          if (ent) { comp = (uint32_t)ent - 1u; need = (uint32_t)(ent >> 32); idx = 0; break; }
          const int r = claim_sorted(false);
          if (r == 0) { need = ~A.order_key[first + idx]; comp = A.order[first + idx]; break; }
          if (r == 3) break;
          /* nothing anywhere: done when every producer has left and the list is empty */
          if (atomicAdd(&A.cold[GTS_COLD_DONE], 0ull) >= A.cold[GTS_COLD_PRODUCERS] &&
              atomicAdd(&A.cold[GTS_COLD_TAIL], 0ull) <= atomicAdd(&A.cold[GTS_COLD_HEAD], 0ull))
            break;
          if ((spins & 63u) == 63u && GtsWave64::clock() - tc0 >= wait_limit) { atomicAdd(pstat + 7, 1ull); break; }
          __builtin_amdgcn_s_sleep(32);
        }
      } else
      for (uint32_t spins = 0;;) {
        const int r = claim_sorted(true);
        if (r != 1) break;
        /* nothing for this wavefront now, but the front (held by the one that
           waits for pages) has: look again later */
        if ((spins++ & 63u) == 0 && GtsWave64::clock() - tc0 >= wait_limit) { atomicAdd(pstat + 7, 1ull); break; }
        __builtin_amdgcn_s_sleep(16);
      }
      if (idx != GTS_NONE) {
        if (!cold_mode) {
          need = ~A.order_key[first + idx];   /* the sort key: one load instead of three dependent ones */
          comp = A.order[first + idx];
        }
        atomicAdd(&ctl->t_claim, (unsigned long long)(GtsWave64::clock() - tc0));
        npages = (need + GTS_POOL_PAGE - 1u) / GTS_POOL_PAGE;
        bool waiting = false;
        const uint64_t tw0 = GtsWave64::clock();
        for (uint32_t spins = 0;;) {
          /* a lock not taken: the component is not run (its state stays as it is);
             its front claim stays with this wavefront, the others of the
             workgroup run into their own bounds: the launch ends */
          if (!pool_lock(ctl, pstat, wait_limit)) { idx = GTS_NONE; break; }
          volatile GtsPoolCtl *v = (volatile GtsPoolCtl *)ctl;
          const gts_pool_bits bits = pool_bits_load(v->used);
          if (cold_mode) {
            /* no roles: a wavefront takes pages from above and stays clear of the
               pages a waiting one needs; the first that finds no room becomes that
               waiting one, takes from below, and what the others free stays free
               for it */
            pos = pool_find(bits, npages, waiting, waiting ? 0u : v->wait_pages, PAGES);
            if (pos != GTS_NONE) {
              pool_bits_set(ctl->used, pool_run_mask(pos, npages));
              if (waiting) v->wait_pages = 0;
            } else if (!waiting && v->wait_pages == 0) { v->wait_pages = npages; waiting = true; }
          } else {
            const uint32_t floor = front ? 0u : v->wait_pages;
            pos = pool_find(bits, npages, front, floor, PAGES);
            if (pos != GTS_NONE) {
              pool_bits_set(ctl->used, pool_run_mask(pos, npages));
              if (front) { v->front_busy = 0; v->wait_pages = 0; }
            } else if (front && !waiting) {
              v->wait_pages = npages; waiting = true;
            }
          }
          if (pos == GTS_NONE && (spins++ & 63u) == 0 && GtsWave64::clock() - tw0 >= wait_limit) {
            /* give up: the component is not run, the host sees the count */
            atomicAdd(pstat + 8, 1ull);
            if (cold_mode ? waiting : front) { v->front_busy = 0; v->wait_pages = 0; }
            pool_unlock(ctl);
            idx = GTS_NONE;
            break;
          }
          pool_unlock(ctl);
          if (pos != GTS_NONE) break;
          __builtin_amdgcn_s_sleep(8);
        }
        atomicAdd(&ctl->t_wait, (unsigned long long)(GtsWave64::clock() - tw0));
      }
    }
    idx = GtsWave64::uni(idx);
    if (idx == GTS_NONE) {
      /* nothing left to claim: while a wavefront of the workgroup is still inside a
         program it may open a job -- stay for those (the launch used to end with a few
         components walking their terminals one by one next to 4000 idle wavefronts).
         Round the loop again: try_help() is at its top */
      if constexpr (!FAST) {
        if (C.help_walks && GtsWave64::uni(GtsWave64::hub_load(&ctl->job.n_running)) != 0 &&
            (uint64_t)(++lingered) * 64ull < wait_limit) {
          __builtin_amdgcn_s_sleep(32);
          continue;
        }
      }
      break;
    }
    pos = GtsWave64::uni(pos); npages = GtsWave64::uni(npages); need = GtsWave64::uni(need);
    const uint32_t c = GtsWave64::uni(comp);
    /* the components of a workgroup are neighbours in LDS: a word at the end of
       the last page (when the footprint leaves room) shows a program that wrote
       past its pages' arrays */
    volatile uint32_t *canary = need + 16u <= npages * GTS_POOL_PAGE
                                    ? (volatile uint32_t *)(smem + (pos + npages) * GTS_POOL_PAGE - 4u) : nullptr;
    if (A.poison >= 0) {
      /* test aid: the pages hold this byte instead of what the last component
         left there -- a program that reads scratch it has not written shows */
      uint32_t *pw = (uint32_t *)(smem + pos * GTS_POOL_PAGE);
      const uint32_t word = (uint32_t)(A.poison & 0xFF) * 0x01010101u;
      for (uint32_t i = lane; i < npages * (GTS_POOL_PAGE / 4u); i += GTS_WAVE) pw[i] = word;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0 && canary) *canary = 0x5CAFF01Du;
    {
      const uint64_t tr0 = GtsWave64::clock();
      GtsCompMemT<true> M;
      /* what the last page has left behind the footprint: more walk slots */
      stage_component(C, c, smem + pos * GTS_POOL_PAGE, M, false, npages * GTS_POOL_PAGE - (canary ? 16u : 0u));
      GtsComponent<GtsWave64, true> prog(C, M, c);
      if constexpr (FAST) {
        const uint64_t cb = (uint64_t)M.ne * 19ull + (uint64_t)M.nv * 18ull;   /* as k_comp_lds_keys counts them */
        if (prog.run_fast(A.mode)) {
          if (lane == 0) {
            atomicAdd(&ctl->n_done, prog.clean ? 0x100000001ull : 1ull);   /* finished | clean << 32 */
            atomicAdd(&ctl->n_walks, (unsigned long long)prog.nfast);
            atomicAdd(&ctl->b_done_bytes, (unsigned long long)cb);
          }
        } else {
          /* not for this program (nothing has left LDS): to the cold list */
          if (lane == 0) {
            atomicAdd(&ctl->n_cold, 1ull); atomicAdd(&ctl->b_cold_bytes, (unsigned long long)cb);
            const unsigned long long at = atomicAdd(&A.cold[GTS_COLD_TAIL], 1ull);
            __hip_atomic_store(&A.cold_list[at], (unsigned long long)need << 32 | (unsigned long long)(c + 1u),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      } else {
        prog.local_marks = C.local_marks != 0;
        if (C.help_walks) { prog.hub = &ctl->job; prog.hub_me = my_wave; }
        prog.lean_stats = C.tspan == nullptr;   /* per-component tables for the detailed profile only */
        if (lane == 0) GtsWave64::hub_add(&ctl->job.n_running, 1u);
        prog.run(A.mode);
        if (lane == 0) {
          GtsWave64::hub_add(&ctl->job.n_running, 0xFFFFFFFFu);
          if (prog.lean_stats) {
            atomicAdd(&ctl->n_done, prog.run_clean ? 0x100000001ull : 1ull);
            atomicAdd(&ctl->n_walks, (unsigned long long)prog.nfast | (unsigned long long)prog.nslow << 32);
            if (prog.run_deferred) atomicAdd(&ctl->n_cold, 1ull);
          }
        }
      }
      if (lane == 0) atomicAdd(&ctl->t_run, (unsigned long long)(GtsWave64::clock() - tr0));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) {
      /* test aid (lds_poison = 256): the word is overwritten as a program that ran
         past its arrays would -- the report and the restore are what is tested */
      if (A.poison == 256 && canary) *canary = 0u;
      if (canary && *canary != 0x5CAFF01Du) atomicAdd(pstat + 9, 1ull);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      pool_bits_clear(ctl->used, pool_run_mask(pos, npages));
    }
  }
  ctl = pool_ctl_opaque(ctl0);
  if (GtsWave64::lane() == 0) {
    const uint64_t t_end = GtsWave64::clock();
    const uint64_t t_begin = ctl->t_begin[my_wave];
    atomicAdd(&ctl->t_life, (unsigned long long)(t_end - t_begin));
    atomicMin(pstat + 5, t_begin);
    atomicMin(pstat + 3, t_end); atomicMax(pstat + 4, t_end);
    /* (its entries of the cold list are complete before the count below moves) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (atomicAdd(&ctl->exited, 1u) + 1u == blockDim.x / GTS_WAVE) {
      /* the workgroup's last wavefront: its sums, and -- fast kernel -- the word
         that tells the cold workgroups that this producer has left */
      volatile GtsPoolCtl *v = (volatile GtsPoolCtl *)ctl;
      atomicAdd(pstat + 0, v->t_run); atomicAdd(pstat + 1, v->t_wait); atomicAdd(pstat + 2, v->t_life);
      if constexpr (FAST) atomicAdd(pstat + 15, v->t_claim);
      else {
        if (v->n_helped) { atomicAdd(pstat + 10, v->n_helped); atomicAdd(pstat + 11, v->t_helping); }
        /* totals of the programs that kept no per-component statistics (lean_stats):
           finished | clean << 32, linear walks | reference walks << 32, deferred */
        if (v->n_done) { atomicAdd(A.ptot + 0, v->n_done); atomicAdd(A.ptot + 1, v->n_walks); atomicAdd(A.ptot + 2, v->n_cold); }
      }
      if constexpr (FAST) {
        if (v->n_done) { atomicAdd(pstat + 10, v->n_done); atomicAdd(pstat + 11, v->n_walks); atomicAdd(pstat + 13, v->b_done_bytes); }
        if (v->n_cold) { atomicAdd(pstat + 12, v->n_cold); atomicAdd(pstat + 14, v->b_cold_bytes); }
        if (A.cold) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          atomicAdd(&A.cold[GTS_COLD_DONE], 1ull);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(GTS_POOL_WAVES * GTS_WAVE)
k_components_pool(const GtsCompView *C, const GtsPoolArgs *A)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ GtsPoolCtl ctl_s;
  pool_body<false, GTS_POOL_PAGES>(C, A, smem, &ctl_s);
}
/* two workgroups per CU (GTS_FAST_BYTES of dynamic LDS each) of WAVES wavefronts:
   the register budget follows from the launch bounds (2 x WAVES / 4 wavefronts
   per SIMD: 96 registers a lane for 10, 80 for 12, 64 for 16).  Measured in round 4
   with 12 / 14 / 16 wavefronts: slower than one workgroup with the whole pool --
   the launch is bound by LDS x time, the page waits triple */
template <int WAVES>
__global__ void __launch_bounds__(WAVES * GTS_WAVE, (2 * WAVES + 3) / 4)
k_components_fast2(const GtsCompView *C, const GtsPoolArgs *A)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ GtsPoolCtl ctl_s;
  pool_body<true, GTS_FAST_PAGES>(C, A, smem, &ctl_s);
}
/* the same program on one workgroup per CU with the whole pool (the launch is
   bound by LDS x time: one pool packs better than two halves) */
__global__ void __launch_bounds__(GTS_POOL_WAVES * GTS_WAVE)
k_components_fast(const GtsCompView *C, const GtsPoolArgs *A)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ GtsPoolCtl ctl_s;
  pool_body<true, GTS_POOL_PAGES>(C, A, smem, &ctl_s);
}

/* one deferred walk per workgroup (gts_component.hpp, try_defer / walk_task);
   launched once per LDS size class, a workgroup whose task belongs to another
   class leaves at once */
__global__ void __launch_bounds__(GTS_WAVE)
k_walk_tasks(GtsCompView C, uint32_t klass, uint32_t count, uint32_t lds_bytes)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.x >= count) return;
  const uint32_t t = C.tq[C.tq_base[klass] + blockIdx.x];
  const uint32_t c = C.task_comp[t];
  GtsCompMemT<true> M;
  stage_component(C, c, smem, M, true, lds_bytes);
  GtsComponent<GtsWave64, true, true> prog(C, M, c);
  prog.walk_task(t);
}
/* the same for the pending tasks of all classes in one launch (few tasks: the
   rounds after the first), with the LDS of the largest class among them */
struct GtsTaskPrefix { uint32_t pre[GTS_NKLASS + 1]; };
__global__ void __launch_bounds__(GTS_WAVE)
k_walk_tasks_mixed(GtsCompView C, GtsTaskPrefix P, uint32_t lds_bytes)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.x >= P.pre[GTS_NKLASS]) return;
  uint32_t k = 0;
  while (blockIdx.x >= P.pre[k + 1]) ++k;
  const uint32_t t = C.tq[C.tq_base[k] + blockIdx.x - P.pre[k]];
  const uint32_t c = C.task_comp[t];
  GtsCompMemT<true> M;
  stage_component(C, c, smem, M, true, lds_bytes);
  GtsComponent<GtsWave64, true, true> prog(C, M, c);
  prog.walk_task(t);
}
/* Deferred walks of the components that run from global memory (too large or
   too wide for the LDS layout): one wavefront per pending walk, the
   component's graph, vertex states and analysis read in place (nothing writes
   them during a round), the walk's scratch in a slab of its own -- the
   component's scratch arrays in the workspace are one set, and the walks of a
   component run side by side.  bytes per vertex of a slab: GTS_SLAB_BYTES. */
#define GTS_SLAB_BYTES 72u
__global__ void __launch_bounds__(GTS_WAVE)
k_walk_tasks_global(GtsCompView C, uint32_t klass, uint32_t first, uint32_t count, char *slabs,
                    uint64_t slab_stride)
{
  if (blockIdx.x >= count) return;
  const uint32_t t = C.tq[C.tq_base[klass] + first + blockIdx.x];
  const uint32_t c = C.task_comp[t];
  GtsCompMem M = GtsComponent<GtsWave64>::global_mem(C, c);
  const uint32_t nv = M.nv, lane = threadIdx.x;
  const uint64_t nva = ((uint64_t)nv + 3) & ~3ull;   /* keeps every array 8-byte aligned */
  char *p = slabs + (uint64_t)blockIdx.x * slab_stride;
  M.nd = (int64_t *)p; p += nva * 8;
  M.plen = (uint64_t *)p; p += nva * 8;
  uint32_t *u = (uint32_t *)p;
  M.queue = u; u += nva; M.visited = u; u += nva; M.st_v = u; u += nva; M.st_par = u; u += nva;
  M.st_cur = u; u += nva; M.edgemap = u; u += nva; M.par = u; u += nva; M.lastpop = u; u += nva;
  M.wterm = u; u += nva; M.touched = u; u += nva; M.cc_best = u; u += nva;
  M.distmap = (float *)u; u += nva;
  uint8_t *b = (uint8_t *)u;
  M.st_dir = b; b += nva; M.tight = b;
  /* what the component's own arrays guarantee between walks (the rest is
     written before it is read; walk_task clears st_dir, tight and the position
     bitmap in st_cur) */
  for (uint32_t s = lane; s < nv; s += GTS_WAVE) { M.distmap[s] = GTS_DIST_UNSET; M.lastpop[s] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  GtsComponent<GtsWave64> prog(C, M, c);
  prog.walk_task(t);
}

__global__ void __launch_bounds__(GTS_WAVE)
k_select_walks(GtsCompView C, uint32_t ndeferred)
{
  if (blockIdx.x >= ndeferred) return;
  const uint32_t c = C.defer_list[blockIdx.x];
  if (!C.defer_flag[c]) return;
  GtsComponent<GtsWave64, false>::select_walks(C, c, C.wbits + C.comp_off[c] / 32 + c);
}
/* LDS footprint of every component as a descending sort key, and how many
   components fit each size class */
__global__ void k_comp_lds_keys(const uint32_t *comp_off, const uint32_t *coff,
                                uint32_t *keys, uint32_t *vals, uint32_t ncomp,
                                const uint8_t *comp_wide, const uint8_t *comp_d32,
                                const unsigned long long *comp_len,
                                uint8_t *comp_klass, const uint32_t *klass, uint32_t nklass,
                                uint32_t *klass_count, unsigned long long *klass_bytes,
                                uint32_t *klass_slots, uint32_t big_nv, uint32_t big_slots,
                                uint32_t huge_nv, uint32_t huge_slots, uint32_t fast_bytes)
{
  /* counters are summed per workgroup in LDS first: seven global counters hit by
     every component serialise */
  __shared__ uint32_t s_cnt[GTS_NKLASS + 1], s_slots[GTS_NKLASS + 1];
  __shared__ unsigned long long s_bytes[GTS_NKLASS + 1];
  if (threadIdx.x <= GTS_NKLASS) { s_cnt[threadIdx.x] = 0; s_bytes[threadIdx.x] = 0; s_slots[threadIdx.x] = 0; }
  __syncthreads();
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncomp) {
    const uint32_t s0 = comp_off[c], s1 = comp_off[c + 1];
    const uint32_t cnv = s1 - s0, cne = coff[s1] - coff[s0];
    /* footprint, and room for walk slots from big_nv contigs on (gts_comp_lds_want) */
    uint32_t need = gts_comp_lds_want(cnv, cne, big_nv, big_slots, GTS_POOL_BYTES - 16u, comp_d32[c] != 0,
                                      huge_nv, huge_slots);
    /* k_components_fast on half pools: a component whose footprint fits one asks
       for the slots that fit it too (the others run on the cold workgroups' pools) */
    if (fast_bytes && need > fast_bytes - 16u && gts_comp_lds_bytes(cnv, cne, comp_d32[c] != 0) <= fast_bytes - 16u)
      need = gts_comp_lds_want(cnv, cne, big_nv, big_slots, fast_bytes - 16u, comp_d32[c] != 0, huge_nv, huge_slots);
    /* not representable in the packed LDS layout: run from global memory */
    if (comp_wide[c] || cnv >= 4096u || cne > GTS_LDS_MAX_INDEX || comp_len[c] >= (1ull << 32))
      need = 0x7FFFFFFFu;
    keys[c] = ~need;   /* ascending sort = largest first */
    vals[c] = (uint32_t)c;
    uint32_t k = 0;
    while (k < nklass && need > klass[k]) ++k;   /* klass ascending; nklass = global */
    comp_klass[c] = (uint8_t)k;
    atomicAdd(&s_slots[k], cnv);
    atomicAdd(&s_cnt[k], 1u);
    /* bytes the component's program has to touch once: its compact graph and
       vertex records in, vertex states and edge marks out */
    atomicAdd(&s_bytes[k], (unsigned long long)(coff[s1] - coff[s0]) * 19ull +
                               (unsigned long long)(s1 - s0) * 18ull);
  }
  __syncthreads();
  if (threadIdx.x <= GTS_NKLASS && s_cnt[threadIdx.x]) {
    atomicAdd(&klass_count[threadIdx.x], s_cnt[threadIdx.x]);
    atomicAdd(&klass_bytes[threadIdx.x], s_bytes[threadIdx.x]);
    atomicAdd(&klass_slots[threadIdx.x], s_slots[threadIdx.x]);
  }
}
/* start of every class' segment of the pending-task array: a class has at most
   as many tasks as its components have vertices */
__global__ void k_task_queue_bases(const uint32_t *klass_slots, uint32_t *tq_base)
{
  uint32_t acc = 0;
  for (int k = 0; k <= GTS_NKLASS; ++k) { tq_base[k] = acc; acc += klass_slots[k]; }
}
__global__ void k_count_errors(const uint32_t *cerr, uint32_t ncomp,
                               uint32_t *out /* [0]=ring pool, [1]=loop, [3]=path pool */)
{
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncomp) return;
  /* one atomic in the source: written as an if / else-if chain of three, hipcc 7.2
     merged them and sent a wavefront's cerr == 2 lanes to out[0] whenever it held
     no lane with cerr == 0 (profiles/r04_isa_k_count_errors_miscompiled.txt): a
     walk error was taken for an exhausted ring pool */
  const uint32_t ce = cerr[c];
  const uint32_t k = ce == GTS_CERR_WALKQ_OVERFLOW ? 0u : ce == GTS_CERR_PATH_OVERFLOW ? 3u : 1u;
  if (ce != 0) atomicAdd(out + k, 1u);
}
/* statistics: the lanes of a wave add up (or take the maximum) first, one
   atomic per wave */
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}
/* out[k] = sum, out[4+k] = max of column k of the per-component tick table */
__global__ void k_tstat_reduce(const uint64_t *t, uint32_t ncomp, unsigned long long *out)
{
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < 4; ++k) {
    const unsigned long long v = c < ncomp ? t[5 * c + k] : 0ull;
    const unsigned long long sm = wave_sum(v), mx = wave_max(v);
    if ((threadIdx.x & 63u) == 0 && sm) { atomicAdd(&out[k], sm); atomicMax(&out[4 + k], mx); }
  }
}
__global__ void k_sum_u32(const uint32_t *a, uint32_t n, unsigned long long *out)
{
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long sm = wave_sum(c < n ? (unsigned long long)a[c] : 0ull);
  if ((threadIdx.x & 63u) == 0 && sm) atomicAdd(out, sm);
}
__global__ void k_sum_bit(const uint32_t *a, uint32_t n, uint32_t bit, unsigned long long *out)
{
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long sm = wave_sum(c < n ? (unsigned long long)(a[c] >> bit & 1u) : 0ull);
  if ((threadIdx.x & 63u) == 0 && sm) atomicAdd(out, sm);
}
__global__ void k_max_u32_diff(const uint32_t *off, uint32_t n, uint32_t *out)
{
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) atomicMax(out, off[c + 1] - off[c]);
}

/* ---- results ---- */
__global__ void k_states_by_id(const uint8_t *state, const uint32_t *eid,
                               uint8_t *out, uint32_t m)
{
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < m) out[eid[p]] = state[p];
}
__global__ void k_edges_by_id(const uint32_t *eid, const uint32_t *estart,
                              const uint32_t *eend, const int64_t *dist,
                              const float *sd, const int64_t *np,
                              const uint8_t *flags, uint32_t *ostart,
                              uint32_t *oend, int64_t *odist, float *osd,
                              int64_t *onp, uint8_t *oflags, uint32_t m)
{
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const uint32_t id = eid[p];
  ostart[id] = estart[p]; oend[id] = eend[p]; odist[id] = dist[p];
  osd[id] = sd[p]; onp[id] = np[p]; oflags[id] = flags[p] & 3u;   /* (bit 2 is internal) */
}
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
/* sum over items of mix(id, state): independent of storage order */
__global__ void k_digest(const uint8_t *state, const uint32_t *ids, uint64_t n,
                         unsigned long long *out)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t h = 0;
  if (i < n) h = mix64(((uint64_t)(ids ? ids[i] : (uint32_t)i) << 8) | state[i]);
  for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)h);
}

/* diagnostic: the device's ambiguous-order test on caller-supplied pairs */
__global__ void k_amb_test(const int64_t *d1, const float *s1, const int64_t *d2,
                           const float *s2, uint8_t *out, uint64_t n,
                           GtsAmbThresholds t)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = gts_ambiguous(d1[i], s1[i], d2[i], s2[i], t) ? 1 : 0;
}

/* ------------------------------------------------------------------ */
/* host side of the C ABI                                              */

static GtsGraphView view_of(GtsgEngine *e)
{
  GtsGraphView G;
  G.n = e->n; G.m = e->m; G.row = e->row; G.seq_len = e->seq_len;
  G.astat = e->astat; G.copy_num = e->copy_num; G.vstate = e->vstate;
  G.end = e->eend; G.dist = e->dist; G.sd = e->sd; G.flags = e->flags;
  G.state = e->state; G.twin = e->twin; G.eid = e->eid;
  return G;
}

template <typename T>
static int upload(GtsgEngine *e, T *dst, const T *src, size_t count, int on_device)
{
  if (!count) return 0;
  HIPCHK(hipMemcpyAsync(dst, src, count * sizeof(T),
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                        e->st));
  return 0;
}

extern "C" {

int gtsg_create(GtsgEngine **out, int device, void *stream)
{
  if (!out) return GTSG_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    fprintf(stderr, "gtsg_create: no HIP device %d (found %d); the engine has "
                    "no CPU path\n", device, ndev);
    return GTSG_EHIP;
  }
  GtsgEngine *e = new GtsgEngine();
  e->device = device;
  /* gtsg_destroy releases whatever has been created so far */
  if (hipSetDevice(device) != hipSuccess) { delete e; return GTSG_EHIP; }
  if (stream) e->st = (hipStream_t)stream;
  else {
    if (hipStreamCreate(&e->st) != hipSuccess) { delete e; return GTSG_EHIP; }
    e->own_stream = true;
  }
  int rc = 0;
  if (hipMalloc((void **)&e->d_scalars, 16384) != hipSuccess) rc = GTSG_ENOMEM;
  for (int k = 0; k < GTS_NSTREAMS && !rc; ++k)
    if (hipStreamCreateWithFlags(&e->side[k], hipStreamNonBlocking) != hipSuccess) rc = GTSG_EHIP;
  for (int k = 0; k < GTS_NKLASS && !rc; ++k)
    if (hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming) != hipSuccess) rc = GTSG_EHIP;
  if (!rc && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) rc = GTSG_EHIP;
  if (!rc) {
    /* at the lowest priority: the runtime keeps a pool of hardware queues per
       priority, so this stream does not share a queue with the streams of the
       rounds (measured with a default-priority stream: the rounds of walk tasks
       queued up behind the team kernel, 18 -> 200 ms) */
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&e->team_st, hipStreamNonBlocking, least) != hipSuccess) rc = GTSG_EHIP;
  }
  if (!rc && hipEventCreateWithFlags(&e->ev_team, hipEventDisableTiming) != hipSuccess) rc = GTSG_EHIP;
  const void *big_lds[] = {(const void *)k_walk_tasks_mixed, (const void *)k_walk_tasks,
                           (const void *)k_components_lds,
                           (const void *)k_components_pool, (const void *)k_components_fast2<GTS_FAST_WAVES>,
                           (const void *)k_components_fast,
                           (const void *)k_components_team};
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      e->n_cus = cus;
  }
  for (const void *f : big_lds)
    if (!rc && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159744) != hipSuccess) {
      fprintf(stderr, "gtsg_create: cannot raise the dynamic LDS limit to 160 KiB\n");
      rc = GTSG_EHIP;
    }
  if (rc) { gtsg_destroy(e); return rc; }
  *out = e;
  return 0;
}

static void free_graph(GtsgEngine *e, bool release)
{
  if (release) {
    void *ptrs[] = {e->row, e->estart, e->eend, e->twin, e->eid, e->pos_of_eid,
                    e->dist, e->npairs, e->sd, e->flags, e->state, e->hubs};
    for (void *p : ptrs) if (p) { e->alloc_bytes.erase(p); hipFree(p); }
    e->row = e->estart = e->eend = e->twin = e->eid = e->pos_of_eid = e->hubs = nullptr;
    e->dist = e->npairs = nullptr; e->sd = nullptr; e->flags = e->state = nullptr;
  }
  e->m = 0; e->nhub = 0; e->built = false;
}

void gtsg_destroy(GtsgEngine *e)
{
  if (!e) return;
  hipSetDevice(e->device);
  hipStreamSynchronize(e->st);
  collect_times(e);
  free_graph(e, true);
  void *ptrs[] = {e->seq_len, e->astat, e->copy_num, e->vstate, e->vtime, e->pool, e->d_scalars,
                  e->gtask_pool, e->team_pool, e->text_buf, e->rec_buf};
  for (void *p : ptrs) if (p) hipFree(p);
  if (e->text_host) hipHostFree(e->text_host);
  for (auto ev : e->free_events) hipEventDestroy(ev);
  for (int k = 0; k < GTS_NSTREAMS; ++k) if (e->side[k]) hipStreamDestroy(e->side[k]);
  for (int k = 0; k < GTS_NKLASS; ++k) if (e->ev_join[k]) hipEventDestroy(e->ev_join[k]);
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  if (e->ev_team) hipEventDestroy(e->ev_team);
  if (e->team_st) hipStreamDestroy(e->team_st);
  if (e->own_stream) hipStreamDestroy(e->st);
  delete e;
}

const char *gtsg_last_error(const GtsgEngine *e) { return e ? e->err.c_str() : "no engine"; }

int gtsg_set_option(GtsgEngine *e, const char *name, int64_t value)
{
  if (!e || !name) return GTSG_EINVAL;
  if (!strcmp(name, "walk_queue_factor") && value >= 1) e->walk_queue_factor = value;
  else if (!strcmp(name, "walk_pool_entries") && value >= 1) e->walk_pool_entries = value;
  else if (!strcmp(name, "max_walk_pops") && value >= 1) e->max_walk_pops = value;
  else if (!strcmp(name, "hub_degree") && value >= 1) e->hub_degree = value;
  else if (!strcmp(name, "fast_walks")) e->fast_walks = value != 0;
  else if (!strcmp(name, "lds_components")) e->lds_components = value != 0;
  else if (!strcmp(name, "batch_walks") && value >= 0 && value <= 2) e->batch_walks = value;   /* 2: components that are not clean too */
  else if (!strcmp(name, "small_masks")) e->small_masks = value != 0;
  else if (!strcmp(name, "team_coff")) e->team_coff = value != 0;
  else if (!strcmp(name, "team_lds_bytes") && value >= 0) e->team_lds_bytes = value;
  else if (!strcmp(name, "lds_int16_distances")) e->lds_int16_distances = value != 0;
  else if (!strcmp(name, "batch_big_contigs") && value >= 0) e->batch_big_contigs = value;
  else if (!strcmp(name, "batch_big_slots") && value >= 2 && value <= GTS_WALK_SLOTS_MAX) e->batch_big_slots = value;
  else if (!strcmp(name, "batch_huge_contigs") && value >= 0) e->batch_huge_contigs = value;
  else if (!strcmp(name, "batch_huge_slots") && value >= 2 && value <= GTS_WALK_SLOTS_MAX) e->batch_huge_slots = value;
  else if (!strcmp(name, "defer_min_contigs") && value >= 0) e->defer_min_contigs = value;
  else if (!strcmp(name, "defer_min_work") && value >= 0) e->defer_min_work = value;
  else if (!strcmp(name, "defer_unclean_work") && value >= 0) e->defer_unclean_work = value;
  else if (!strcmp(name, "defer_ref_min_contigs") && value >= 0) e->defer_ref_min_contigs = value;
  else if (!strcmp(name, "task_reference_walks")) e->task_reference_walks = value != 0;
  else if (!strcmp(name, "pool_components")) e->pool_components = value != 0;
  else if (!strcmp(name, "pool_waves") && value >= 1 && value <= GTS_POOL_WAVES) e->pool_waves = value;
  else if (!strcmp(name, "lds_poison") && value >= -1 && value <= 256) e->lds_poison = value;   /* 256: also clobbers the canaries */
  else if (!strcmp(name, "gather_unroll") && value >= 1) e->gather_unroll = value;
  else if (!strcmp(name, "gather_nt")) e->gather_nt = value != 0;
  else if (!strcmp(name, "pair_sort_full")) e->pair_sort_full = value != 0;
  else if (!strcmp(name, "pair_bucket_limit") && value >= 1 && value <= (1 << 20)) e->pair_bucket_limit = value;
  else if (!strcmp(name, "pool_fill_kb") && value >= 0) e->pool_fill_kb = value;
  else if (!strcmp(name, "fast_components")) e->fast_components = value != 0;
  else if (!strcmp(name, "fast_split")) e->fast_split = value != 0;
  else if (!strcmp(name, "local_marks")) e->local_marks = value != 0;
  else if (!strcmp(name, "help_walks")) e->help_walks = value != 0;
  else if (!strcmp(name, "timing_skip_writeback")) e->timing_skip_writeback = value != 0;
  else if (!strcmp(name, "fast_waves") && value >= 1 && value <= GTS_FAST_WAVES) e->fast_waves = value;
  else if (!strcmp(name, "cold_cus") && value >= 1 && value <= 255) e->cold_cus = value;
  else if (!strcmp(name, "pool_wait_limit_us") && value >= 0) e->pool_wait_limit_us = value;
  else if (!strcmp(name, "class_streams") && value >= 1 && value <= GTS_NSTREAMS) e->class_streams = value;
  else if (!strcmp(name, "mixed_task_limit") && value >= 0) e->mixed_task_limit = value;

  else if (!strcmp(name, "walk_path_entries") && value >= 1) e->walk_path_entries = value;
  else if (!strcmp(name, "global_task_pool_mb") && value >= 0) e->global_task_pool_mb = value;
  else if (!strcmp(name, "defer_global_components")) e->defer_global_components = value != 0;
  else if (!strcmp(name, "team_components")) e->team_components = value != 0;
  else if (!strcmp(name, "team_max_components") && value >= 0) e->team_max_components = value;
  else if (!strcmp(name, "team_pool_mb") && value >= 0) e->team_pool_mb = value;
  else if (!strcmp(name, "profile")) e->profile = (int)value;
  else return fail(e, GTSG_EINVAL, "unknown option %s", name);
  return 0;
}

int gtsg_set_contigs(GtsgEngine *e, uint64_t n, const int64_t *seq_len,
                     const float *astat, const float *copy_num, int on_device)
{
  if (!e || (n && !seq_len)) return GTSG_EINVAL;
  if (n >= (1ull << 31)) return fail(e, GTSG_ELIMIT, "more than 2^31-1 contigs");
  HIPCHK(hipSetDevice(e->device));
  free_graph(e, false);
  e->n = (uint32_t)n;
  int rc;
  if ((rc = dev_alloc(e, &e->seq_len, n))) return rc;
  if ((rc = dev_alloc(e, &e->astat, n))) return rc;
  if ((rc = dev_alloc(e, &e->copy_num, n))) return rc;
  if ((rc = dev_alloc(e, &e->vstate, n))) return rc;
  if ((rc = upload(e, e->seq_len, seq_len, n, on_device))) return rc;
  if (astat) { if ((rc = upload(e, e->astat, astat, n, on_device))) return rc; }
  else HIPCHK(hipMemsetAsync(e->astat, 0, n * 4, e->st));
  if (copy_num) { if ((rc = upload(e, e->copy_num, copy_num, n, on_device))) return rc; }
  else HIPCHK(hipMemsetAsync(e->copy_num, 0, n * 4, e->st));
  HIPCHK(hipMemsetAsync(e->vstate, GIS_UNVISITED, n ? n : 1, e->st));
  e->have_vtime = false;
  return sync_stream(e);
}

int gtsg_set_vertex_times(GtsgEngine *e, const uint32_t *times, int on_device)
{
  if (!e) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (!times) { e->have_vtime = false; return 0; }
  int rc;
  if ((rc = dev_alloc(e, &e->vtime, e->n))) return rc;
  if ((rc = upload(e, e->vtime, times, e->n, on_device))) return rc;
  e->have_vtime = true;
  return sync_stream(e);
}

int gtsg_set_astat(GtsgEngine *e, const float *astat, const float *copy_num,
                   int on_device)
{
  if (!e || !astat || !copy_num) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if ((rc = upload(e, e->astat, astat, e->n, on_device))) return rc;
  if ((rc = upload(e, e->copy_num, copy_num, e->n, on_device))) return rc;
  return sync_stream(e);
}

int gtsg_build_from_records(GtsgEngine *e, uint64_t nrec, const uint32_t *root,
                            const uint32_t *ctg, const int64_t *dist,
                            const float *std_dev, const int64_t *num_pairs,
                            const uint8_t *flags, int on_device)
{
  return gtsg_build_from_records_ex(e, nrec, root, ctg, dist, std_dev, num_pairs, flags,
                                    on_device, 0);
}

int gtsg_build_from_records_ex(GtsgEngine *e, uint64_t nrec, const uint32_t *root,
                               const uint32_t *ctg, const int64_t *dist,
                               const float *std_dev, const int64_t *num_pairs,
                               const uint8_t *flags, int on_device, int ismatepair)
{
  if (!e || (nrec && (!root || !ctg || !dist || !std_dev || !flags))) return GTSG_EINVAL;
  /* the one-sweep sort keeps counts in 30 bits (gts_prims.hpp) */
  if (nrec >= GTS_ONESWEEP_MAX_N / 2) return fail(e, GTSG_ELIMIT, "2^29 records or more");
  HIPCHK(hipSetDevice(e->device));
  free_graph(e, false);
  const uint32_t n = e->n;
  int rc;
  /* workspace estimate (bytes per record / edge, generous) */
  const size_t ws = nrec * (on_device ? 64 : 96) + (size_t)n * 16 +
                    (gts_sort_tmp_elems(2 * nrec + 16) + gts_scan_tmp_elems(2 * nrec + n + 16)) * 8 +
                    2 * nrec * (sizeof(GtsEdgeRec) + 24) + (64u << 20);
  if ((rc = pool_reserve(e, ws))) return rc;
  /* records on the device */
  const uint32_t *d_root = root, *d_ctg = ctg;
  const int64_t *d_dist = dist, *d_np = num_pairs;
  const float *d_sd = std_dev;
  const uint8_t *d_flags = flags;
  if (!on_device && nrec) {
    PALLOC(t_root, uint32_t, nrec); PALLOC(t_ctg, uint32_t, nrec);
    PALLOC(t_dist, int64_t, nrec); PALLOC(t_sd, float, nrec);
    PALLOC(t_flags, uint8_t, nrec);
    if ((rc = upload(e, t_root, root, nrec, 0)) || (rc = upload(e, t_ctg, ctg, nrec, 0)) ||
        (rc = upload(e, t_dist, dist, nrec, 0)) || (rc = upload(e, t_sd, std_dev, nrec, 0)) ||
        (rc = upload(e, t_flags, flags, nrec, 0)))
      return rc;
    d_root = t_root; d_ctg = t_ctg; d_dist = t_dist; d_sd = t_sd; d_flags = t_flags;
    if (num_pairs) {
      PALLOC(t_np, int64_t, nrec);
      if ((rc = upload(e, t_np, num_pairs, nrec, 0))) return rc;
      d_np = t_np;
    }
  }
  uint32_t npairs_created = 0;
  uint8_t *is_creator = nullptr, *replaced = nullptr;   /* one byte per record */
  uint32_t *jidx = nullptr, *fwd = nullptr, *bwd = nullptr;
  if (nrec) {
    PALLOC(k0, uint64_t, nrec); PALLOC(k1, uint64_t, nrec);
    PALLOC(v0, uint32_t, nrec); PALLOC(v1, uint32_t, nrec);
    PALLOC(stmp, uint32_t, gts_sort_tmp_elems(nrec));
    PALLOC(t_isc, uint8_t, nrec); PALLOC(t_rep, uint8_t, nrec);
    is_creator = t_isc; replaced = t_rep;
    PALLOC(t_fwd, uint32_t, nrec); PALLOC(t_bwd, uint32_t, nrec);
    PALLOC(t_jidx, uint32_t, nrec);
    PALLOC(sctmp, uint32_t, gts_scan_tmp_elems(nrec));
    fwd = t_fwd; bwd = t_bwd; jidx = t_jidx;
    HIPCHK(hipMemsetAsync(e->d_scalars + 6, 0, 8, e->st));
    e->stats["pair_sort_fallback"] = 0;
    const int vb = bits_for(n);
    /* the pair keys are made inside the sort's first pass (and its histogram
       pass) from the records themselves: no key / value arrays are written first */
    const GtsSortSrc src = {GTS_SRC_RECORDS, d_root, d_ctg, n, e->d_scalars + 6};
    /* first the short way: the records bucketed on about log2(nrec) bits taken from
       the low end of both contig ids (k_pair_segments_bucket); the full order on
       (larger, smaller) when a bucket is too long for that, or when asked for
       (pair_sort_full) */
    for (int full = e->pair_sort_full ? 1 : 0; full < 2; ++full) {
      int shifts[8], np = 0;
      uint64_t bmask = 0;
      if (full) {
        for (int s = 0; s < vb; s += 8) shifts[np++] = s;
        for (int s = 0; s < vb; s += 8) shifts[np++] = 32 + s;
      } else {
        /* digits alternately from the smaller and the larger contig, lowest first,
           until the buckets hold ~8 records on average or the ids are used up */
        const int want = bits_for(nrec) > 4 ? bits_for(nrec) - 3 : 1;
        for (int s = 0; s < vb && 8 * np < want; s += 8) {
          shifts[np++] = s;
          if (8 * np < want) shifts[np++] = 32 + s;
        }
        for (int q = 0; q < np; ++q) {
          const int width = vb - (shifts[q] & 31) < 8 ? vb - (shifts[q] & 31) : 8;
          bmask |= ((1ull << width) - 1) << shifts[q];
        }
      }
      int where;
      e->stats["pair_sort_passes"] = np;
      { ProfScope ps(e, "build_sort_pairs");
        where = gts_radix_sort<uint64_t>(k0, v0, k1, v1, nrec, shifts, np, stmp, e->st, &src); }
      if (where < 0) return fail(e, GTSG_ELIMIT, "too many records for the pair sort");
      uint64_t *ks = where ? k1 : k0;
      uint32_t *vs = where ? v1 : v0;
      HIPCHK(hipMemsetAsync(replaced, 0, nrec, e->st));
      HIPCHK(hipMemsetAsync(is_creator, 1, nrec, e->st));
      if (full) {
        LAUNCH("build_pair_segments", k_pair_segments, nblk(nrec), GTS_BLOCK, ks, vs, d_sd,
               is_creator, replaced, fwd, bwd, nrec, ismatepair ? 1 : 0);
        break;
      }
      LAUNCH("build_pair_segments", k_pair_segments_bucket, nblk(nrec), GTS_BLOCK, ks, vs, d_sd,
             is_creator, replaced, fwd, bwd, nrec, ismatepair ? 1 : 0, bmask, (uint32_t)e->pair_bucket_limit,
             e->d_scalars + 7);
      uint32_t over = 0;
      if ((rc = read_u32(e, e->d_scalars + 7, &over))) return rc;
      if (!over) break;
      e->stats["pair_sort_fallback"] += 1;
    }
    { ProfScope ps(e, "build_scan_creators");
      gts_exscan<uint8_t, uint32_t>(is_creator, jidx, nrec, sctmp, e->d_scalars, e->st); }
    uint32_t bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, e->d_scalars + 6, 4, hipMemcpyDeviceToHost, e->st));
    if ((rc = read_u32(e, e->d_scalars, &npairs_created))) return rc;
    if (bad) return fail(e, GTSG_EINVAL, "contig id out of range in the records (%u contigs)", n);
  }
  const uint64_t m64 = 2ull * npairs_created;
  if (m64 >= GTS_ONESWEEP_MAX_N) return fail(e, GTSG_ELIMIT, "2^30 edges or more");
  const uint32_t m = (uint32_t)m64;
  e->m = m;
  if ((rc = dev_alloc(e, &e->row, (size_t)n + 1))) return rc;
  if ((rc = dev_alloc(e, &e->estart, m))) return rc;
  if ((rc = dev_alloc(e, &e->eend, m))) return rc;
  if ((rc = dev_alloc(e, &e->twin, m))) return rc;
  if ((rc = dev_alloc(e, &e->eid, m))) return rc;
  if ((rc = dev_alloc(e, &e->pos_of_eid, m))) return rc;
  if ((rc = dev_alloc(e, &e->dist, m))) return rc;
  if ((rc = dev_alloc(e, &e->npairs, m))) return rc;
  if ((rc = dev_alloc(e, &e->sd, m))) return rc;
  if ((rc = dev_alloc(e, &e->flags, m))) return rc;
  if ((rc = dev_alloc(e, &e->state, m))) return rc;
  HIPCHK(hipMemsetAsync(e->vstate, GIS_UNVISITED, n ? n : 1, e->st));
  if (m) {
    const int vb = bits_for(n);
    int shifts[4], np = 0;
    for (int s = 0; s < vb; s += 8) shifts[np++] = s;
    /* the sort ping-pongs between two pairs of buffers; the persistent arrays
       are the pair its last pass writes to, so nothing is copied afterwards */
    PALLOC(es_t, uint32_t, m); PALLOC(id_t, uint32_t, m);
    uint32_t *es0 = (np & 1) ? es_t : e->estart, *id0 = (np & 1) ? id_t : e->eid;
    uint32_t *es1 = (np & 1) ? e->estart : es_t, *id1 = (np & 1) ? e->eid : id_t;
    PALLOC(rec, GtsEdgeRec, m);
    PALLOC(stmp2, uint32_t, gts_sort_tmp_elems(m));
    LAUNCH("build_emit_edges", k_emit_edges, nblk(nrec), GTS_BLOCK, is_creator, replaced, jidx, fwd,
           bwd, d_root, d_ctg, d_dist, d_sd, d_np, d_flags, es0, rec, nrec);
    int where;
    const GtsSortSrc iota = {GTS_SRC_IOTA, nullptr, nullptr, 0, nullptr};
    { ProfScope ps(e, "build_sort_csr");
      where = gts_radix_sort<uint32_t>(es0, id0, es1, id1, m, shifts, np, stmp2, e->st, &iota); }
    if (where < 0 || (where ? es1 : es0) != e->estart)
      return fail(e, GTSG_EHIP, "CSR sort ended in the wrong buffer");
    LAUNCH("build_row_offsets", k_row_offsets, nblk((uint64_t)m + 1), GTS_BLOCK, e->estart,
           e->row, n, m);
#define GTS_GATHER_LAUNCH(U, NT)                                                                          \
  do {                                                                                                   \
    LAUNCH("build_gather_csr", (k_gather_csr<U, NT>), nblk(m, GTS_BLOCK * U), GTS_BLOCK, e->eid, rec,    \
           e->eend, e->dist, e->npairs, e->sd, e->flags, e->state, e->pos_of_eid, m);                    \
    LAUNCH("build_twins", (k_twins<U, NT>), nblk(m, GTS_BLOCK * U), GTS_BLOCK, e->eid, e->pos_of_eid,    \
           e->twin, m);                                                                                  \
  } while (0)
    const int gu = (int)e->gather_unroll;
    if (e->gather_nt) {
      if (gu >= 8) GTS_GATHER_LAUNCH(8, true); else if (gu >= 4) GTS_GATHER_LAUNCH(4, true);
      else if (gu >= 2) GTS_GATHER_LAUNCH(2, true); else GTS_GATHER_LAUNCH(1, true);
    } else {
      if (gu >= 8) GTS_GATHER_LAUNCH(8, false); else if (gu >= 4) GTS_GATHER_LAUNCH(4, false);
      else if (gu >= 2) GTS_GATHER_LAUNCH(2, false); else GTS_GATHER_LAUNCH(1, false);
    }
  } else
    HIPCHK(hipMemsetAsync(e->row, 0, ((size_t)n + 1) * 4, e->st));
  /* hub list */
  if (n) {
    PALLOC(hflag, uint32_t, n); PALLOC(hidx, uint32_t, n);
    PALLOC(htmp, uint32_t, gts_scan_tmp_elems(n));
    e->built_hub_degree = (uint32_t)e->hub_degree;
    LAUNCH("build_hub_flags", k_hub_flags, nblk(n), GTS_BLOCK, e->row, hflag, n,
           e->built_hub_degree);
    gts_exscan<uint32_t, uint32_t>(hflag, hidx, n, htmp, e->d_scalars + 2, e->st);
    uint32_t nh = 0;
    if ((rc = read_u32(e, e->d_scalars + 2, &nh))) return rc;
    e->nhub = nh;
    if ((rc = dev_alloc(e, &e->hubs, nh))) return rc;
    if (nh) LAUNCH("build_hub_list", k_compact_ids, nblk(n), GTS_BLOCK, hflag, hidx, e->hubs, n);
  }
  e->stats["hubs"] = e->nhub;
  e->built = true;
  return sync_stream(e);
}

int gtsg_mark_repeats(GtsgEngine *e, int have_file, float copy_num_cutoff,
                      float astat_cutoff)
{
  if (!e || !e->built) return e ? fail(e, GTSG_EINVAL, "graph not built") : GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if (e->pool_cap < (size_t)e->n + 4096) { if ((rc = pool_reserve(e, (size_t)e->n + 4096))) return rc; }
  else pool_reserve(e, 0);
  PALLOC(isrep, uint8_t, (size_t)e->n + 1);
  if (e->n)
    LAUNCH("repeat_vertices", k_repeat_vertices, nblk(e->n), GTS_BLOCK, e->astat,
           e->copy_num, e->vstate, isrep, e->n, have_file, copy_num_cutoff, astat_cutoff);
  if (e->m)
    LAUNCH("repeat_edges", k_repeat_edges, nblk(e->m), GTS_BLOCK, e->estart, e->twin, isrep,
           e->state, e->m);
  return sync_stream(e);
}

int gtsg_filter_begin(GtsgEngine *e, float pcutoff, float cncutoff, int64_t ocutoff)
{
  if (!e || !e->built) return e ? fail(e, GTSG_EINVAL, "graph not built") : GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  const uint32_t n = e->n, m = e->m;
  e->filter_open = false;
  if (!n) return 0;
  int rc;
  if ((rc = pool_reserve(e, (size_t)m * 6 + (size_t)n * 43 + (8u << 20)))) return rc;
  GtsGraphView G = view_of(e);
  GtsFilterParams P;
  P.amb = gts_amb_thresholds(pcutoff);
  P.cncutoff = cncutoff;
  P.ocutoff = ocutoff;
  const int zero_ovf = 0 > ocutoff;
  PALLOC(prop, uint8_t, (size_t)m + 1); PALLOC(newstate, uint8_t, (size_t)m + 1);
  PALLOC(vinfo, uint8_t, n); PALLOC(ovf, uint8_t, n);
  PALLOC(tpoly, uint32_t, n); PALLOC(lasthit, uint32_t, 2 * (size_t)n);
  uint32_t *pending = e->d_scalars + 4;
  PALLOC(vattr, GtsVAttr, n);
  PALLOC(elen, int32_t, (size_t)m + 1); PALLOC(ispoly, uint32_t, (size_t)n / 32 + 2);
  HIPCHK(hipMemsetAsync(prop, 0, (size_t)m + 1, e->st));
  LAUNCH("filter_pack_vattr", k_pack_vattr, nblk(n), GTS_BLOCK, e->seq_len, e->copy_num, vattr, n);
  /* the degree the hub list was built with: option "hub_degree" set after the
     build takes effect at the next build */
  LAUNCH("filter_pairs", k_filter_pairs, nblk(n), GTS_BLOCK, G, P, vattr, prop, vinfo,
         e->built_hub_degree, elen);
  if (e->nhub)
    LAUNCH("filter_pairs_hub", k_filter_pairs_hub, nblk((uint64_t)e->nhub * GTS_WAVE),
           GTS_BLOCK, G, P, prop, vinfo, e->hubs, e->nhub);
  int64_t rounds_p = 0, rounds_i = 0;
  PALLOC(proposed, uint8_t, (size_t)n + 1);
  HIPCHK(hipMemsetAsync(proposed, 0, (size_t)n + 1, e->st));
  if (m) LAUNCH("filter_active_round", k_filter_proposed, nblk(m), GTS_BLOCK, G, e->estart, prop, proposed);
  /* The rounds of a fixpoint are idempotent once it is reached (a settled
     vertex returns at once), and a host round trip costs more than a round
     that finds nothing to do: several rounds per look at the flag. */
  for (;;) {
    uint32_t h = 0;
    for (int r = 0; r < 3; ++r) {
      HIPCHK(hipMemsetAsync(pending, 0, 4, e->st));
      LAUNCH("filter_active_round", k_filter_active_round, nblk(n), GTS_BLOCK, G, prop, proposed, vinfo,
             pending);
      ++rounds_p;
    }
    if ((rc = read_u32(e, pending, &h))) return rc;
    if (!h) break;
  }
  HIPCHK(hipMemsetAsync(tpoly, 0xFF, (size_t)n * 4, e->st));
  if (m) LAUNCH("filter_tpoly", k_filter_tpoly, nblk(m), GTS_BLOCK, G, e->estart, prop, vinfo, tpoly);
  LAUNCH("filter_pack_vattr", k_poly_bitmap, nblk(n), GTS_BLOCK, tpoly, ispoly, n);
  LAUNCH("filter_ovf_init", k_filter_ovf_init, nblk(n), GTS_BLOCK, G, P, elen, ispoly, e->estart, vinfo,
         tpoly, ovf, zero_ovf, e->built_hub_degree);
  if (e->nhub && !zero_ovf)
    LAUNCH("filter_ovf_init_hub", k_filter_ovf_init_hub,
           nblk((uint64_t)e->nhub * GTS_WAVE), GTS_BLOCK, G, P, vinfo, tpoly, ovf, e->hubs,
           e->nhub);
  for (;;) {
    uint32_t h = 0;
    for (int r = 0; r < 4; ++r) {
      HIPCHK(hipMemsetAsync(pending, 0, 4, e->st));
      LAUNCH("filter_hit_round", k_filter_hit_round, nblk(n), GTS_BLOCK, G, ovf, zero_ovf,
             pending);
      ++rounds_i;
    }
    if ((rc = read_u32(e, pending, &h))) return rc;
    if (!h) break;
  }
  HIPCHK(hipMemsetAsync(lasthit, 0xFF, (size_t)n * 8, e->st));
  if (m) LAUNCH("filter_lasthit", k_filter_lasthit, nblk(m), GTS_BLOCK, G, e->estart, ovf, lasthit,
                e->have_vtime ? e->vtime : (const uint32_t *)nullptr);
  e->stats["filter_rounds_p"] = rounds_p;
  e->stats["filter_rounds_i"] = rounds_i;
  e->f_tpoly = tpoly; e->f_ovf = ovf; e->f_lasthit = lasthit; e->f_newstate = newstate;
  e->filter_open = true;
  return sync_stream(e);
}

/* the "latest hit" table (2 x u32 per contig, GTS_NONE = none) between the two
   halves of the filter; as int32 it is ordered (-1 = none), so shards that
   share marked (repeat) vertices combine it with an all-reduce MAX */
int gtsg_filter_get_lasthit(GtsgEngine *e, uint32_t *dst, int on_device)
{
  if (!e || !dst || !e->filter_open) return e ? fail(e, GTSG_EINVAL, "no open filter call") : GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(dst, e->f_lasthit, 8ull * e->n,
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}
int gtsg_filter_set_lasthit(GtsgEngine *e, const uint32_t *src, int on_device)
{
  if (!e || !src || !e->filter_open) return e ? fail(e, GTSG_EINVAL, "no open filter call") : GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(e->f_lasthit, src, 8ull * e->n,
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->st));
  return sync_stream(e);
}

int gtsg_filter_end(GtsgEngine *e)
{
  if (!e || !e->filter_open) return e ? fail(e, GTSG_EINVAL, "no open filter call") : GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  const uint32_t n = e->n, m = e->m;
  GtsGraphView G = view_of(e);
  e->filter_open = false;
  if (m) {
    const uint32_t *vt = e->have_vtime ? e->vtime : (const uint32_t *)nullptr;
    LAUNCH("filter_final", k_filter_final, nblk(m), GTS_BLOCK, G, e->estart, e->f_tpoly,
           e->f_ovf, e->f_lasthit, e->f_newstate, vt);
    LAUNCH("filter_final", k_filter_final_poly_ends, nblk(m), GTS_BLOCK, G, e->estart, e->f_tpoly,
           e->f_ovf, e->f_lasthit, e->f_newstate, vt);
    HIPCHK(hipMemcpyAsync(e->state, e->f_newstate, m, hipMemcpyDeviceToDevice, e->st));
  }
  LAUNCH("filter_final_vertices", k_filter_final_vertices, nblk(n), GTS_BLOCK, e->vstate,
         e->f_tpoly, n);
  return sync_stream(e);
}

int gtsg_filter(GtsgEngine *e, float pcutoff, float cncutoff, int64_t ocutoff)
{
  int rc = gtsg_filter_begin(e, pcutoff, cncutoff, ocutoff);
  if (rc || !e->filter_open) return rc;
  return gtsg_filter_end(e);
}

/* Component labels of the contigs under a slice of records (multi-GPU
   partition step): labels[] holds parent pointers with labels[v] <= v (identity
   at first, or the element-wise minimum over the shards' previous results);
   every record whose two contigs are not skipped joins their trees (larger
   root under the smaller); on return labels[v] is the root = smallest contig
   of v's tree. */
__global__ void k_label_union(const uint32_t *root, const uint32_t *ctg,
                              const uint8_t *skip, uint32_t *parent, uint64_t nrec,
                              uint32_t n, uint32_t *bad)
{
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nrec) return;
  uint32_t x = root[k], y = ctg[k];
  if (x >= n || y >= n) { *bad = 1; return; }
  if (skip && (skip[x] || skip[y])) return;
  for (;;) {
    x = uf_find(parent, x); y = uf_find(parent, y);
    if (x == y) break;
    if (x < y) { const uint32_t t = x; x = y; y = t; }
    if (atomicCAS(&parent[x], x, y) == x) break;
  }
}
/* every entry its root.  The climb only reads: the path halving of uf_find
   stores grandparents, and such a store landing after another thread's final
   store would leave that entry pointing at an inner vertex (seen as contigs
   without an owner in the plan of the partition).  What the other threads of
   this kernel store are roots, so any mix of old and new entries leads to the
   root. */
__global__ void k_label_flatten(uint32_t *parent, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  uint32_t r = (uint32_t)v, p = parent[r];
  while (p != r) { r = p; p = parent[r]; }
  parent[v] = r;
}
/* ---- routing of records to the shard that owns their component (multi-GPU) ----
   A record travels as four 64-bit words (gt-scaffold_amd/dist.py, pack_records):
     word 0: root | sense << 31 | ctg << 32 | same << 63
     word 1: dist        word 2: num_pairs
     word 3: global record index | bits(std_dev) << 32 */
__global__ void k_route_dest(const uint32_t *root, const uint32_t *ctg, const int8_t *owner,
                             uint32_t world, uint32_t *dest, uint32_t *idx, uint64_t nrec,
                             uint32_t n, uint32_t *bad)
{
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nrec) return;
  const uint32_t a = root[k], b = ctg[k];
  if (a >= n || b >= n) { *bad = 1; dest[k] = 0; idx[k] = (uint32_t)k; return; }
  /* owner < 0: a repeat contig, shared by the shards; the record follows its
     other contig; both repeats: any rank, the same for every record of the pair */
  const int oa = owner[a], ob = owner[b];
  dest[k] = oa >= 0 ? (uint32_t)oa : ob >= 0 ? (uint32_t)ob : (a < b ? a : b) % world;
  idx[k] = (uint32_t)k;
}
__global__ void k_route_rows(const uint32_t *perm, const uint32_t *root, const uint32_t *ctg,
                             const int64_t *dist, const float *sd, const int64_t *np,
                             const uint8_t *flags, uint64_t k0, uint64_t *rows, uint64_t nrec)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrec) return;
  const uint32_t k = perm[i];
  const uint64_t f = flags[k];
  ulonglong2 lo, hi;
  lo.x = (uint64_t)root[k] | ((f & 1ull) << 31) | ((uint64_t)ctg[k] << 32) | (((f >> 1) & 1ull) << 63);
  lo.y = (uint64_t)dist[k];
  hi.x = np ? (uint64_t)np[k] : 0ull;
  hi.y = (k0 + k) | ((uint64_t)__float_as_uint(sd[k]) << 32);
  ((ulonglong2 *)rows)[2 * i] = lo;
  ((ulonglong2 *)rows)[2 * i + 1] = hi;
}
/* rows -> record arrays; loc_of (may be null) turns whole-graph contig ids into
   the shard's local numbers on the way */
__global__ void k_route_unpack(const uint64_t *rows, const uint32_t *loc_of, uint32_t *root,
                               uint32_t *ctg, int64_t *dist, float *sd, int64_t *np,
                               uint8_t *flags, uint64_t *kidx, uint64_t nrec, uint32_t *unsorted)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrec) return;
  const ulonglong2 lo = ((const ulonglong2 *)rows)[2 * i], hi = ((const ulonglong2 *)rows)[2 * i + 1];
  /* rows dealt in file-order chunks arrive in file order; say so if they did not */
  if (i && (rows[4 * (i - 1) + 3] & 0xFFFFFFFFull) > (hi.y & 0xFFFFFFFFull)) *unsorted = 1;
  const uint32_t a = (uint32_t)lo.x & 0x7FFFFFFFu, b = (uint32_t)(lo.x >> 32) & 0x7FFFFFFFu;
  root[i] = loc_of ? loc_of[a] : a;
  ctg[i] = loc_of ? loc_of[b] : b;
  flags[i] = (uint8_t)(((lo.x >> 31) & 1ull) | (((lo.x >> 63) & 1ull) << 1));
  dist[i] = (int64_t)lo.y; np[i] = (int64_t)hi.x;
  sd[i] = __uint_as_float((uint32_t)(hi.y >> 32));
  kidx[i] = hi.y & 0xFFFFFFFFull;
}

int gtsg_route_pack(GtsgEngine *e, uint64_t nrec, const uint32_t *root, const uint32_t *ctg,
                    const int64_t *dist, const float *std_dev, const int64_t *num_pairs,
                    const uint8_t *flags, uint64_t first_index, uint64_t n_contigs,
                    const int8_t *owner, uint32_t world, uint64_t *rows, uint64_t *counts)
{
  if (!e || !counts || world < 1 || world > 127 ||
      (nrec && (!root || !ctg || !dist || !std_dev || !flags || !owner || !rows)))
    return GTSG_EINVAL;
  if (nrec >= GTS_ONESWEEP_MAX_N || first_index + nrec >= (1ull << 32) || n_contigs >= (1ull << 31))
    return fail(e, GTSG_ELIMIT, "too many records for one routing call");
  HIPCHK(hipSetDevice(e->device));
  for (uint32_t r = 0; r < world; ++r) counts[r] = 0;
  if (!nrec) return 0;
  int rc;
  if ((rc = pool_reserve(e, nrec * 16 + gts_sort_tmp_elems(nrec) * 4 + (1u << 20)))) return rc;
  PALLOC(d0, uint32_t, nrec); PALLOC(d1, uint32_t, nrec);
  PALLOC(i0, uint32_t, nrec); PALLOC(i1, uint32_t, nrec);
  PALLOC(stmp, uint32_t, gts_sort_tmp_elems(nrec));
  HIPCHK(hipMemsetAsync(e->d_scalars + 6, 0, 4, e->st));
  LAUNCH("route_dest", k_route_dest, nblk(nrec), GTS_BLOCK, root, ctg, owner, world, d0, i0, nrec,
         (uint32_t)n_contigs, e->d_scalars + 6);
  /* one stable 8-bit pass by destination keeps the file order inside a destination;
     its digit histogram (exclusive prefix after the sort) gives the counts */
  const int shift0 = 0;
  int where;
  { ProfScope ps(e, "route_sort");
    where = gts_radix_sort<uint32_t>(d0, i0, d1, i1, nrec, &shift0, 1, stmp, e->st); }
  if (where < 0) return fail(e, GTSG_ELIMIT, "too many records for one routing call");
  LAUNCH("route_rows", k_route_rows, nblk(nrec), GTS_BLOCK, where ? i1 : i0, root, ctg, dist, std_dev,
         num_pairs, flags, first_index, rows, nrec);
  uint32_t base[256], bad = 0;
  HIPCHK(hipMemcpyAsync(base, stmp, sizeof base, hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipMemcpyAsync(&bad, e->d_scalars + 6, 4, hipMemcpyDeviceToHost, e->st));
  if ((rc = sync_stream(e))) return rc;
  if (bad) return fail(e, GTSG_EINVAL, "contig id out of range in the records");
  for (uint32_t r = 0; r < world; ++r)
    counts[r] = (r + 1 < 256 ? base[r + 1] : (uint32_t)nrec) - base[r];
  counts[world - 1] = nrec - base[world - 1];
  return 0;
}

int gtsg_route_unpack_ex(GtsgEngine *e, uint64_t nrec, const uint64_t *rows, const uint32_t *loc_of,
                         uint32_t *root, uint32_t *ctg, int64_t *dist, float *std_dev,
                         int64_t *num_pairs, uint8_t *flags, uint64_t *index, int *out_of_order)
{
  if (!e || (nrec && (!rows || !root || !ctg || !dist || !std_dev || !num_pairs || !flags || !index)))
    return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (out_of_order) *out_of_order = 0;
  HIPCHK(hipMemsetAsync(e->d_scalars + 6, 0, 4, e->st));
  if (nrec)
    LAUNCH("route_unpack", k_route_unpack, nblk(nrec), GTS_BLOCK, rows, loc_of, root, ctg, dist, std_dev,
           num_pairs, flags, index, nrec, e->d_scalars + 6);
  uint32_t uns = 0;
  int rc;
  if ((rc = read_u32(e, e->d_scalars + 6, &uns))) return rc;
  if (out_of_order) *out_of_order = uns ? 1 : 0;
  return 0;
}
int gtsg_route_unpack(GtsgEngine *e, uint64_t nrec, const uint64_t *rows, const uint32_t *loc_of,
                      uint32_t *root, uint32_t *ctg, int64_t *dist, float *std_dev,
                      int64_t *num_pairs, uint8_t *flags, uint64_t *index)
{
  return gtsg_route_unpack_ex(e, nrec, rows, loc_of, root, ctg, dist, std_dev, num_pairs, flags, index, nullptr);
}

int gtsg_label_components(GtsgEngine *e, uint64_t n, uint64_t nrec, const uint32_t *root,
                          const uint32_t *ctg, const uint8_t *skip, uint32_t *labels,
                          int on_device)
{
  if (!e || !labels || (nrec && (!root || !ctg))) return GTSG_EINVAL;
  if (n >= (1ull << 31)) return fail(e, GTSG_ELIMIT, "more than 2^31-1 contigs");
  HIPCHK(hipSetDevice(e->device));
  int rc;
  const uint32_t *d_root = root, *d_ctg = ctg;
  const uint8_t *d_skip = skip;
  uint32_t *d_lab = labels;
  if (!on_device) {
    if ((rc = pool_reserve(e, nrec * 8 + n * 5 + (1u << 20)))) return rc;
    PALLOC(tr, uint32_t, nrec + 1); PALLOC(tc, uint32_t, nrec + 1);
    PALLOC(tl, uint32_t, n + 1); PALLOC(ts, uint8_t, n + 1);
    if ((rc = upload(e, tr, root, nrec, 0)) || (rc = upload(e, tc, ctg, nrec, 0)) ||
        (rc = upload(e, tl, labels, n, 0)) || (skip && (rc = upload(e, ts, skip, n, 0))))
      return rc;
    d_root = tr; d_ctg = tc; d_lab = tl; d_skip = skip ? ts : nullptr;
  }
  HIPCHK(hipMemsetAsync(e->d_scalars + 6, 0, 4, e->st));
  if (nrec)
    LAUNCH("label_union", k_label_union, nblk(nrec), GTS_BLOCK, d_root, d_ctg, d_skip, d_lab, nrec,
           (uint32_t)n, e->d_scalars + 6);
  if (n) LAUNCH("label_flatten", k_label_flatten, nblk(n), GTS_BLOCK, d_lab, (uint32_t)n);
  uint32_t bad = 0;
  if ((rc = read_u32(e, e->d_scalars + 6, &bad))) return rc;
  if (bad) return fail(e, GTSG_EINVAL, "contig id out of range in the records (%llu contigs)",
                       (unsigned long long)n);
  if (!on_device)
    HIPCHK(hipMemcpyAsync(labels, d_lab, n * 4, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

/* ---- plan of the component partition (multi-GPU, dist.py step 2) on the device ----
   weights: records per component of THIS shard (a record counts for the
   component of its first non-repeat contig); after the shards have summed them
   (all_reduce), deal: the components -- their labels are their smallest contigs
   -- go to the ranks largest first, ties by label, in serpentine order. */
__global__ void k_plan_count(const uint32_t *root, const uint32_t *ctg, const uint8_t *skip,
                             uint32_t *cnt, uint64_t nrec, uint32_t n, uint32_t *bad)
{
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t anchor = GTS_NONE;
  if (k < nrec) {
    const uint32_t a = root[k], b = ctg[k];
    if (a >= n || b >= n) *bad = 1;
    else anchor = !skip[a] ? a : !skip[b] ? b : GTS_NONE;
  }
  /* a root's records are neighbours in the file: one atomic per run of equal
     anchors inside the wavefront instead of one per record */
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t prev = (uint32_t)__shfl_up((int)anchor, 1);
  const uint64_t heads = __builtin_amdgcn_ballot_w64(lane == 0 || prev != anchor);
  if ((heads >> lane) & 1ull) {
    const uint64_t later = lane == 63 ? 0ull : heads >> (lane + 1);
    const uint32_t run = later ? (uint32_t)__ffsll((long long)later) : 64u - lane;
    if (anchor != GTS_NONE) atomicAdd(&cnt[anchor], run);
  }
}
__global__ void k_plan_fold(const uint32_t *cnt, const uint32_t *labels, int32_t *w, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n && cnt[v]) atomicAdd(&w[labels[v]], (int32_t)cnt[v]);
}
__global__ void k_plan_keys(const uint32_t *labels, const uint8_t *skip, const int32_t *w,
                            uint32_t *key, uint32_t *val, uint32_t n, uint32_t *bad)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const bool is_root = labels[v] == (uint32_t)v && !skip[v];
  /* the weights are int32 sums over the ranks: a component of 2^31 records or
     more wrapped on the way, and its key would collide with the sentinel */
  if (is_root && w[v] < 0) atomicAdd(bad, 1u);
  key[v] = is_root ? 0xFFFFFFFEu - (uint32_t)w[v] : 0xFFFFFFFFu;   /* ascending = heaviest first, then by id */
  val[v] = (uint32_t)v;
}
__device__ __forceinline__ uint32_t plan_rank(uint64_t pos, uint32_t world)
{
  const uint64_t lap = pos / world;
  const uint32_t col = (uint32_t)(pos % world);
  return (lap & 1ull) ? world - 1u - col : col;
}
__global__ void k_plan_scatter(const uint32_t *key, const uint32_t *val, int8_t *owner_of_root,
                               unsigned long long *load, uint32_t world, uint32_t n)
{
  __shared__ unsigned long long s_load[128];
  if (threadIdx.x < 128) s_load[threadIdx.x] = 0;
  __syncthreads();
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && key[p] != 0xFFFFFFFFu) {
    const uint32_t r = plan_rank(p, world);
    owner_of_root[val[p]] = (int8_t)r;
    atomicAdd(&s_load[r], (unsigned long long)(0xFFFFFFFEu - key[p]));
  }
  __syncthreads();
  if (threadIdx.x < world && s_load[threadIdx.x]) atomicAdd(&load[threadIdx.x], s_load[threadIdx.x]);
}
__global__ void k_plan_owner(const uint32_t *labels, const uint8_t *skip, const int8_t *owner_of_root,
                             int8_t *owner, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) owner[v] = skip[v] ? (int8_t)-1 : owner_of_root[labels[v]];
}

int gtsg_plan_weights(GtsgEngine *e, uint64_t n, uint64_t nrec, const uint32_t *root,
                      const uint32_t *ctg, const uint8_t *skip, const uint32_t *labels,
                      int32_t *weights)
{
  if (!e || !skip || !labels || !weights || (nrec && (!root || !ctg))) return GTSG_EINVAL;
  if (n >= (1ull << 31) || nrec >= (1ull << 31)) return fail(e, GTSG_ELIMIT, "too many contigs or records");
  HIPCHK(hipSetDevice(e->device));
  if (!n) return 0;
  int rc;
  if ((rc = pool_reserve(e, n * 4 + (1u << 20)))) return rc;
  PALLOC(cnt, uint32_t, n);
  HIPCHK(hipMemsetAsync(cnt, 0, n * 4, e->st));
  HIPCHK(hipMemsetAsync(weights, 0, n * 4, e->st));
  HIPCHK(hipMemsetAsync(e->d_scalars + 6, 0, 4, e->st));
  if (nrec)
    LAUNCH("plan_count", k_plan_count, nblk(nrec), GTS_BLOCK, root, ctg, skip, cnt, nrec, (uint32_t)n,
           e->d_scalars + 6);
  LAUNCH("plan_fold", k_plan_fold, nblk(n), GTS_BLOCK, cnt, labels, weights, (uint32_t)n);
  uint32_t bad = 0;
  if ((rc = read_u32(e, e->d_scalars + 6, &bad))) return rc;
  if (bad) return fail(e, GTSG_EINVAL, "contig id out of range in the records");
  return 0;
}

int gtsg_plan_deal(GtsgEngine *e, uint64_t n, const uint8_t *skip, const uint32_t *labels,
                   const int32_t *weights, uint32_t world, int8_t *owner, int64_t *load)
{
  if (!e || !skip || !labels || !weights || !owner || !load || world < 1 || world > 127) return GTSG_EINVAL;
  if (n >= GTS_ONESWEEP_MAX_N) return fail(e, GTSG_ELIMIT, "too many contigs for one plan");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemsetAsync(load, 0, (size_t)world * 8, e->st));
  if (!n) return sync_stream(e);
  int rc;
  if ((rc = pool_reserve(e, n * 17 + gts_sort_tmp_elems(n) * 4 + (1u << 20)))) return rc;
  PALLOC(k0, uint32_t, n); PALLOC(k1, uint32_t, n); PALLOC(v0, uint32_t, n); PALLOC(v1, uint32_t, n);
  PALLOC(oor, int8_t, n);
  PALLOC(stmp, uint32_t, gts_sort_tmp_elems(n));
  HIPCHK(hipMemsetAsync(e->d_scalars + 6, 0, 4, e->st));
  LAUNCH("plan_keys", k_plan_keys, nblk(n), GTS_BLOCK, labels, skip, weights, k0, v0, (uint32_t)n, e->d_scalars + 6);
  {
    uint32_t bad = 0;
    if ((rc = read_u32(e, e->d_scalars + 6, &bad))) return rc;
    if (bad) return fail(e, GTSG_ELIMIT, "%u component(s) of 2^31 records or more: the plan's weights are 32 bit", bad);
  }
  int shifts[4] = {0, 8, 16, 24};
  int where;
  { ProfScope ps(e, "plan_sort");
    where = gts_radix_sort<uint32_t>(k0, v0, k1, v1, n, shifts, 4, stmp, e->st); }
  if (where < 0) return fail(e, GTSG_ELIMIT, "too many contigs for one plan");
  HIPCHK(hipMemsetAsync(oor, 0xFF, n, e->st));
  LAUNCH("plan_scatter", k_plan_scatter, nblk(n), GTS_BLOCK, where ? k1 : k0, where ? v1 : v0, oor,
         (unsigned long long *)load, world, (uint32_t)n);
  LAUNCH("plan_owner", k_plan_owner, nblk(n), GTS_BLOCK, labels, skip, oor, owner, (uint32_t)n);
  return sync_stream(e);
}

static int run_components(GtsgEngine *e, int mode)
{
  if (!e || !e->built) return e ? fail(e, GTSG_EINVAL, "graph not built") : GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  const uint32_t n = e->n, m = e->m;
  if (!n) return 0;
  int rc;
  int64_t factor = e->walk_queue_factor, pool_entries = e->walk_pool_entries;
  int64_t path_entries = e->walk_path_entries;
  int64_t retries = 0;
  /* snapshot for the (rare) walk-queue retry */
  uint8_t *snap_v = nullptr, *snap_e = nullptr;
  for (;;) {
    /* phase A: sizes are data dependent, so the workspace is reserved in two
       steps (the pool cannot grow while pointers into it are live) */
    const size_t wsA = (size_t)m * (2 + 4 + 8) + (size_t)n * 40 +
                       (gts_sort_tmp_elems(n) + 2 * gts_scan_tmp_elems((uint64_t)n + m)) * 8 +
                       (size_t)n + m + (16u << 20);
    /* upper bounds for phase B: slots <= n, compact edges <= m */
    const size_t wsB = (size_t)n * (4 * 16 + 8 + 2 + 4 + 8 + 64 + 64 + 8 + 16 + 48) + (size_t)path_entries * 4 + (size_t)n * 32 + (size_t)m * (4 * 3 + 8 + 2) +
                       (size_t)pool_entries * 12 + (size_t)n * 32 + (16u << 20);
    if (!e->pool || e->pool_cap < wsA + wsB) {
      if ((rc = pool_reserve(e, wsA + wsB))) return rc;
    } else
      e->pool_used = 0;
    GtsGraphView G = view_of(e);
    PALLOC(sv, uint8_t, (size_t)n + 1); PALLOC(se, uint8_t, (size_t)m + 1);
    snap_v = sv; snap_e = se;
    HIPCHK(hipMemcpyAsync(snap_v, e->vstate, n, hipMemcpyDeviceToDevice, e->st));
    if (m) HIPCHK(hipMemcpyAsync(snap_e, e->state, m, hipMemcpyDeviceToDevice, e->st));
    PALLOC(incl, uint8_t, (size_t)m + 1); PALLOC(touched, uint8_t, n);
    PALLOC(parent, uint32_t, n); PALLOC(flag, uint32_t, n); PALLOC(idx, uint32_t, n);
    PALLOC(sctmp, uint32_t, gts_scan_tmp_elems((uint64_t)n + m + 16));
    HIPCHK(hipMemsetAsync(touched, 0, n, e->st));
    LAUNCH("iota", k_iota, nblk(n), GTS_BLOCK, parent, (uint64_t)n);
    if (m)
      LAUNCH("comp_live_union", k_live_union, nblk(m), GTS_BLOCK, G, e->estart, incl, parent);
    LAUNCH("comp_vertices", k_component_roots, nblk(n), GTS_BLOCK, parent, touched, n);
    LAUNCH("comp_vertices", k_component_vertices, nblk(n), GTS_BLOCK, touched, e->vstate,
           parent, flag, n, mode);
    gts_exscan<uint32_t, uint32_t>(flag, idx, n, sctmp, e->d_scalars, e->st);
    uint32_t nslots = 0;
    if ((rc = read_u32(e, e->d_scalars, &nslots))) return rc;
    e->stats["slots"] = nslots;
    e->stats["components"] = 0; e->stats["max_component"] = 0;
    e->stats["compact_edges"] = 0;
    if (!nslots) break;
    PALLOC(lk0, uint32_t, nslots); PALLOC(lk1, uint32_t, nslots);
    PALLOC(sv0, uint32_t, nslots); PALLOC(sv1, uint32_t, nslots);
    PALLOC(stmp, uint32_t, gts_sort_tmp_elems(nslots));
    LAUNCH("comp_slot_keys", k_slot_keys, nblk(n), GTS_BLOCK, flag, idx, parent, lk0, sv0, n);
    {
      const int vb = bits_for(n);
      int shifts[4], np = 0;
      for (int s = 0; s < vb; s += 8) shifts[np++] = s;
      ProfScope ps(e, "comp_sort_slots");
      const int where = gts_radix_sort<uint32_t>(lk0, sv0, lk1, sv1, nslots, shifts, np,
                                                 stmp, e->st);
      if (where < 0) return fail(e, GTSG_ELIMIT, "too many contigs for the slot sort");
      if (where) { uint32_t *t = lk0; lk0 = lk1; lk1 = t; t = sv0; sv0 = sv1; sv1 = t; }
    }
    uint32_t *labels = lk0, *slot_v = sv0, *head = lk1, *cidx = sv1;
    LAUNCH("comp_slot_heads", k_slot_heads, nblk(nslots), GTS_BLOCK, labels, head, nslots);
    gts_exscan<uint32_t, uint32_t>(head, cidx, nslots, sctmp, e->d_scalars, e->st);
    uint32_t ncomp = 0;
    if ((rc = read_u32(e, e->d_scalars, &ncomp))) return rc;
    PALLOC(comp_off, uint32_t, (size_t)ncomp + 1);
    PALLOC(slot_of, uint32_t, n); PALLOC(cseq, int64_t, nslots); PALLOC(vst, uint8_t, nslots);
    LAUNCH("comp_slot_finish", k_slot_finish, nblk(nslots), GTS_BLOCK, labels, cidx, slot_v,
           e->seq_len, e->vstate, comp_off, slot_of, cseq, vst, nslots);
    LAUNCH("fill", k_fill<uint32_t>, 1, 1, comp_off + ncomp, nslots, (uint64_t)1);
    PALLOC(slot_base, uint32_t, nslots);
    PALLOC(slot_comp, uint32_t, nslots); PALLOC(comp_wide, uint8_t, (size_t)ncomp + 1);
    HIPCHK(hipMemsetAsync(comp_wide, 0, (size_t)ncomp + 1, e->st));
    PALLOC(comp_d32, uint8_t, (size_t)ncomp + 1);
    HIPCHK(hipMemsetAsync(comp_d32, e->lds_int16_distances ? 0 : 1, (size_t)ncomp + 1, e->st));
    PALLOC(comp_len, unsigned long long, (size_t)ncomp + 1);
    HIPCHK(hipMemsetAsync(comp_len, 0, ((size_t)ncomp + 1) * 8, e->st));
    LAUNCH("comp_slot_bases", k_slot_bases, nblk(nslots), GTS_BLOCK, head, cidx, comp_off, cseq,
           slot_base, slot_comp, comp_wide, comp_len, nslots);
    PALLOC(coff, uint32_t, (size_t)nslots + 1);
    PALLOC(ipos, uint32_t, (size_t)m + 2);
    gts_exscan<uint8_t, uint32_t>(incl, ipos, m, sctmp, ipos + m, e->st);
    LAUNCH("comp_compact_count", k_compact_count, nblk(nslots), GTS_BLOCK, e->row, ipos, slot_v,
           coff, nslots);
    gts_exscan<uint32_t, uint32_t>(coff, coff, nslots, sctmp, e->d_scalars, e->st);
    uint32_t nce = 0;
    if ((rc = read_u32(e, e->d_scalars, &nce))) return rc;
    LAUNCH("fill", k_fill<uint32_t>, 1, 1, coff + nslots, nce, (uint64_t)1);
    PALLOC(cstart, uint32_t, (size_t)nce + 1); PALLOC(cend, uint32_t, (size_t)nce + 1);
    PALLOC(cgpos, uint32_t, (size_t)nce + 1); PALLOC(cdist, int64_t, (size_t)nce + 1);
    PALLOC(cflags, uint8_t, (size_t)nce + 1); PALLOC(cstate, uint8_t, (size_t)nce + 1);
    PALLOC(cmap, uint32_t, (size_t)m + 1);
    if (m)
      LAUNCH("comp_compact_fill", k_compact_fill, nblk(m, GTS_BLOCK * 2), GTS_BLOCK, G, e->estart, incl, ipos,
             slot_of, slot_base, coff, slot_comp, comp_wide, comp_d32, cstart, cend, cdist, cflags, cgpos,
             cstate, cmap);
    /* walk queue pool of the reference search */
    const uint64_t wq_pool = (uint64_t)pool_entries;
    PALLOC(wq_edge, uint32_t, wq_pool + 1); PALLOC(wq_dist, int64_t, wq_pool + 1);
    unsigned long long *wq_used = (unsigned long long *)(e->d_scalars + 48);
    HIPCHK(hipMemsetAsync(wq_used, 0, 8, e->st));
    /* per-slot scratch */
    PALLOC(s_queue, uint32_t, nslots); PALLOC(s_term, uint32_t, nslots);
    PALLOC(s_visited, uint32_t, nslots); PALLOC(s_stv, uint32_t, nslots);
    PALLOC(s_stpar, uint32_t, nslots); PALLOC(s_stcur, uint32_t, nslots);
    PALLOC(s_edgemap, uint32_t, nslots); PALLOC(s_lastpop, uint32_t, nslots); PALLOC(s_par, uint32_t, nslots);
    PALLOC(s_wterm, uint32_t, nslots); PALLOC(s_touched, uint32_t, nslots);
    PALLOC(s_ccbest, uint32_t, nslots); PALLOC(s_stdir, uint8_t, nslots);
    PALLOC(s_distmap, float, nslots); PALLOC(s_ccoff, uint32_t, (size_t)nslots + ncomp + 1);
    PALLOC(cerr, uint32_t, ncomp);
    PALLOC(s_nd, int64_t, nslots); PALLOC(s_plen, uint64_t, nslots); PALLOC(s_tight, uint8_t, nslots);
    PALLOC(stat_fast, uint32_t, ncomp); PALLOC(stat_slow, uint32_t, ncomp);
    PALLOC(stat_clean, uint32_t, ncomp); PALLOC(stat_ncc, uint32_t, ncomp);
    PALLOC(comp_klass, uint8_t, (size_t)ncomp + 1); PALLOC(defer_flag, uint8_t, (size_t)ncomp + 1);
    PALLOC(comp_task0, uint32_t, ncomp); PALLOC(comp_ncc, uint32_t, ncomp); PALLOC(comp_nterm, uint32_t, ncomp);
    const uint64_t task_cap = nslots, path_cap = (uint64_t)path_entries;
    PALLOC(task_comp, uint32_t, task_cap); PALLOC(task_start, uint32_t, task_cap);
    PALLOC(task_n, uint32_t, task_cap); PALLOC(task_skip, uint8_t, task_cap);
    PALLOC(task_len, uint64_t, task_cap); PALLOC(task_poff, uint64_t, task_cap);
    PALLOC(task_paths, uint32_t, path_cap + 1);
    PALLOC(comp_next_cc, uint32_t, ncomp); PALLOC(wbits, uint32_t, (size_t)nslots / 32 + ncomp + 2);
    PALLOC(task_roff, uint64_t, task_cap); PALLOC(comp_ring, uint64_t, 2 * (size_t)ncomp);
    PALLOC(tq, uint32_t, task_cap + 1); PALLOC(defer_list, uint32_t, (size_t)ncomp + 1);
    HIPCHK(hipMemsetAsync(defer_flag, 0, (size_t)ncomp + 1, e->st));
    HIPCHK(hipMemsetAsync(comp_ring, 0, 16 * (size_t)ncomp, e->st));   /* no ring yet (try_defer used to store the zeros) */
    HIPCHK(hipMemsetAsync(e->d_scalars + 128, 0, 16, e->st));
    PALLOC(s_gorient, uint8_t, nslots); PALLOC(s_topo, uint32_t, nslots); PALLOC(s_tpos, uint32_t, nslots);
    PALLOC(tstat, uint64_t, 5 * (size_t)ncomp);
    PALLOC(ok0, uint32_t, ncomp); PALLOC(ok1, uint32_t, ncomp);
    PALLOC(ov0, uint32_t, ncomp); PALLOC(ov1, uint32_t, ncomp);
    PALLOC(otmp, uint32_t, gts_sort_tmp_elems(ncomp));
    HIPCHK(hipMemsetAsync(s_lastpop, 0, (size_t)nslots * 4, e->st));
    HIPCHK(hipMemsetAsync(cerr, 0, (size_t)ncomp * 4, e->st));
    LAUNCH("fill", k_fill<float>, nblk(nslots), GTS_BLOCK, s_distmap, GTS_DIST_UNSET,
           (uint64_t)nslots);
    /* components by decreasing LDS footprint; size classes of the LDS launches */
    const bool use_fast = e->fast_components && e->pool_components && e->lds_components && e->fast_walks &&
                          e->batch_walks && e->n_cus > (int)e->cold_cus && e->cold_cus >= 1;
    const uint32_t *klass_h = gts_klass_bytes;
    const uint32_t nklass = GTS_NKLASS;
    uint32_t *klass_d = e->d_scalars + GTS_S_KSIZE, *klass_count = e->d_scalars + GTS_S_KCOUNT;
    HIPCHK(hipMemsetAsync(e->d_scalars + GTS_S_KSIZE, 0, (GTS_S_NDEF + 4 - GTS_S_KSIZE) * 4, e->st));
    HIPCHK(hipMemcpyAsync(klass_d, klass_h, nklass * sizeof(uint32_t), hipMemcpyHostToDevice, e->st));
    LAUNCH("comp_lds_keys", k_comp_lds_keys, nblk(ncomp), GTS_BLOCK, comp_off, coff, ok0, ov0,
           ncomp, comp_wide, comp_d32,
           comp_len, comp_klass, klass_d, (uint32_t)(e->lds_components ? nklass : 0), klass_count,
           (unsigned long long *)(e->d_scalars + GTS_S_KBYTES), e->d_scalars + GTS_S_KSLOTS,
           (uint32_t)(e->batch_walks && mode == GTS_MODE_MAKESCAFFOLD ? e->batch_big_contigs : 0),
           (uint32_t)e->batch_big_slots, (uint32_t)e->batch_huge_contigs, (uint32_t)e->batch_huge_slots,
           (uint32_t)(use_fast && e->fast_split ? GTS_FAST_BYTES : 0u));
    LAUNCH("comp_lds_keys", k_task_queue_bases, 1, 1, e->d_scalars + GTS_S_KSLOTS, e->d_scalars + GTS_S_TQBASE);
    const uint32_t *order, *order_key;   /* order_key[i] = ~footprint of component order[i] */
    {
      int shifts[4] = {0, 8, 16, 24};
      const int where = gts_radix_sort<uint32_t>(ok0, ov0, ok1, ov1, ncomp, shifts, 4, otmp, e->st);
      if (where < 0) return fail(e, GTSG_ELIMIT, "too many components for the footprint sort");
      order = where ? ov1 : ov0;
      order_key = where ? ok1 : ok0;
    }
    uint32_t kcount[GTS_NKLASS + 1], kslots_h[GTS_NKLASS + 1];
    uint64_t kbytes[GTS_NKLASS + 1];
    HIPCHK(hipMemcpyAsync(kcount, klass_count, sizeof kcount, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(kslots_h, e->d_scalars + GTS_S_KSLOTS, sizeof kslots_h, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(kbytes, e->d_scalars + GTS_S_KBYTES, sizeof kbytes, hipMemcpyDeviceToHost, e->st));
    /* largest component: sizes the scratch slabs of walks deferred from global memory */
    uint32_t maxcomp = 0;
    HIPCHK(hipMemsetAsync(e->d_scalars + 7, 0, 4, e->st));
    LAUNCH("comp_max_size", k_max_u32_diff, nblk(ncomp), GTS_BLOCK, comp_off, ncomp, e->d_scalars + 7);
    HIPCHK(hipMemcpyAsync(&maxcomp, e->d_scalars + 7, 4, hipMemcpyDeviceToHost, e->st));
    if ((rc = sync_stream(e))) return rc;
    HIPCHK(hipMemsetAsync(e->d_scalars + 12, 0, 16, e->st));
    LAUNCH("comp_max_size", k_max_u32_diff, nblk(ncomp), GTS_BLOCK, comp_off, ncomp,
           e->d_scalars + 14);
    bool pool_ran = false, fast_ran = false, lean_pool = false;
    /* an entry per component and one per ticket a cold wavefront can hold beyond them */
    const size_t cold_entries = (size_t)ncomp + 1 + 256u * GTS_POOL_WAVES;
    PALLOC(cold_list, unsigned long long, cold_entries);
    GtsCompView C;
    C.G = G; C.cmap = cmap; C.ncomp = ncomp; C.comp_off = comp_off; C.slot_v = slot_v;
    C.cseq = cseq; C.coff = coff; C.cstart = cstart; C.cend = cend; C.cdist = cdist;
    C.cflags = cflags; C.cgpos = cgpos; C.cstate = cstate; C.vst = vst; C.comp_d32 = comp_d32;
    C.queue = s_queue; C.term = s_term; C.visited = s_visited; C.st_v = s_stv;
    C.st_par = s_stpar; C.st_cur = s_stcur; C.edgemap = s_edgemap; C.par = s_par; C.lastpop = s_lastpop;
    C.wterm = s_wterm; C.touched = s_touched; C.cc_best = s_ccbest; C.st_dir = s_stdir;
    C.distmap = s_distmap; C.ccoff = s_ccoff; C.wq_edge = wq_edge; C.wq_used = wq_used;
    C.wq_pool = wq_pool; C.wq_factor = (uint64_t)factor;
    C.wq_dist = wq_dist; C.cerr = cerr; C.max_pops = (uint64_t)e->max_walk_pops;
    C.fast_walks = (int)e->fast_walks; C.batch_walks = (int)e->batch_walks; C.small_masks = (int)e->small_masks; C.team_coff = (int)e->team_coff;
    C.timing_skip_writeback = (int)e->timing_skip_writeback; C.local_marks = (int)e->local_marks;
    C.help_walks = (int)e->help_walks;
    C.nd = s_nd;
    C.team_slab = nullptr; C.team_used = nullptr; C.team_cap = 0; C.team_stat = nullptr;
    C.small_stat = nullptr; C.tspan = nullptr;
    if (e->profile >= 2) {
      PALLOC(tspan, uint64_t, 2 * (size_t)ncomp);
      HIPCHK(hipMemsetAsync(tspan, 0, 16 * (size_t)ncomp, e->st));
      C.tspan = tspan;
      C.small_stat = (unsigned long long *)(e->d_scalars + GTS_S_SMALLSTAT);
      HIPCHK(hipMemsetAsync(C.small_stat, 0, 32, e->st));
    }
    bool team_ran = false; C.plen = s_plen; C.tight = s_tight;
    /* (a return before the join below must not leave the team kernel running on
       the workspace) */
    struct TeamJoin {
      GtsgEngine *e; bool armed;
      ~TeamJoin() { if (armed) hipStreamSynchronize(e->team_st); }
    } team_join = {e, false};
    C.stat_fast = stat_fast; C.stat_slow = stat_slow; C.tstat = tstat; C.stat_clean = stat_clean;
    C.stat_ncc = stat_ncc;
    C.gorient = s_gorient; C.topo = s_topo; C.tpos = s_tpos;
    C.defer_min_nv = mode == GTS_MODE_MAKESCAFFOLD && e->lds_components ? (uint32_t)e->defer_min_contigs : 0u;
    C.defer_min_work = (uint64_t)e->defer_min_work;
    C.defer_unclean_work = mode == GTS_MODE_MAKESCAFFOLD && e->lds_components ? (uint64_t)e->defer_unclean_work : 0;
    C.defer_ref_min_nv = mode == GTS_MODE_MAKESCAFFOLD && e->lds_components ? (uint32_t)e->defer_ref_min_contigs : 0u;
    C.task_reference = (int)e->task_reference_walks;
    C.defer_flag = defer_flag; C.comp_task0 = comp_task0; C.comp_ncc = comp_ncc; C.comp_nterm = comp_nterm;
    C.ntasks = (unsigned long long *)(e->d_scalars + 128); C.path_used = (unsigned long long *)(e->d_scalars + 130);
    C.task_bytes = (unsigned long long *)(e->d_scalars + GTS_S_NDEF + 2);
    C.task_cap = task_cap; C.path_cap = path_cap;
    C.task_comp = task_comp; C.task_start = task_start; C.task_n = task_n; C.task_skip = task_skip;
    C.task_len = task_len; C.task_poff = task_poff; C.paths = task_paths;
    C.comp_next_cc = comp_next_cc; C.wbits = wbits;
    C.task_roff = task_roff; C.comp_ring = comp_ring; C.comp_klass = comp_klass; C.tq = tq;
    C.tq_base = e->d_scalars + GTS_S_TQBASE; C.tq_cnt = (unsigned long long *)(e->d_scalars + GTS_S_TQCNT);
    C.defer_list = defer_list; C.ndeferred = (unsigned long long *)(e->d_scalars + GTS_S_NDEF);
    C.why = (unsigned long long *)(e->d_scalars + 96);
    HIPCHK(hipMemsetAsync(C.why, 0, 96, e->st));   /* [8..10]: walks_fast_batch: walks, given up, of them for a cycle */
    {
      /* order[] is sorted by decreasing footprint: the global-memory class
         (larger than every LDS class, or all if LDS is disabled) comes first,
         then the LDS classes from the largest to the smallest */
      const char *kname = mode == GTS_MODE_MAKESCAFFOLD ? "components_makescaffold"
                                                        : "components_removecycles";
      uint32_t first = 0;
      const uint32_t nk = e->lds_components ? nklass : 0;
      for (int k = 0; k < GTS_NKLASS; ++k) {
        e->stats["components_lds_class" + std::to_string(k)] = 0;
        e->stats["bytes_components_lds_class" + std::to_string(k)] = 0;
        e->stats["lds_class" + std::to_string(k) + "_kb"] = k < (int)nk ? (int64_t)(klass_h[k] / 1024) : -1;
      }
      /* profile: the span of the overlapped class launches, fork -> last join,
         as one entry next to the per-launch entries */
      hipEvent_t span_a = nullptr, span_b = nullptr;
      if (e->profile) { span_a = get_event(e); span_b = get_event(e); hipEventRecord(span_a, e->st); }
      HIPCHK(hipEventRecord(e->ev_fork, e->st));
      if (kcount[nk]) {
        /* few of them (the components too large for LDS): a workgroup each, the
           walks of a cc over its wavefronts (k_components_team) */
        const uint64_t team_bytes = (uint64_t)GTS_TEAM_WAVES * (172ull * kslots_h[nk] + 1024ull * kcount[nk]);
        const bool team = mode == GTS_MODE_MAKESCAFFOLD && e->team_components && e->fast_walks &&
                          !e->defer_global_components && kcount[nk] <= (uint32_t)e->team_max_components &&
                          team_bytes <= ((uint64_t)e->team_pool_mb << 20);
        if (team) {
          if ((rc = dev_alloc(e, &e->team_pool, (size_t)team_bytes))) return rc;
          C.team_slab = e->team_pool; C.team_cap = team_bytes;
          C.team_used = (unsigned long long *)(e->d_scalars + GTS_S_TEAMUSED);
          C.team_stat = (unsigned long long *)(e->d_scalars + GTS_S_TEAMSTAT);
          /* on a stream of its own, joined only after the rounds of walk tasks: the
             launch is as long as its largest component's program (223 ms on the 50 M
             workload) and occupies a handful of CUs; team components do not defer
             walks, so the rounds (18 ms there) have nothing to wait for in it */
          hipStream_t ts = e->team_st;
          HIPCHK(hipStreamWaitEvent(ts, e->ev_fork, 0));
          HIPCHK(hipMemsetAsync(C.team_used, 0, 8, ts));
          HIPCHK(hipMemsetAsync(C.team_stat, 0, 64, ts));
          team_ran = true;
          team_join.armed = true;
          uint64_t lds = 18ull * maxcomp + 128 + 64ull * GTS_TCC_K * 8ull;
          if (lds > 159744u - 1024u) lds = 159744u - 1024u;
          if (e->team_lds_bytes > 0 && (uint64_t)e->team_lds_bytes < lds) lds = (uint64_t)e->team_lds_bytes;   /* test aid */
          hipEvent_t _a = nullptr, _b = nullptr;
          if (e->profile) { _a = get_event(e); _b = get_event(e); hipEventRecord(_a, ts); }
          k_components_team<<<kcount[nk], GTS_TEAM_WAVES * GTS_WAVE, (size_t)lds, ts>>>(
              C, order, first, kcount[nk], mode, (uint32_t)lds);
          if (e->profile) { hipEventRecord(_b, ts); e->pending.push_back({kname, _a, _b}); }
          HIPCHK(hipEventRecord(e->ev_team, ts));
        } else
          LAUNCH(kname, k_components, kcount[nk], GTS_WAVE, C, order, first, kcount[nk], mode,
                 (int)e->defer_global_components);
        e->stats["team_components"] = team ? (int64_t)kcount[nk] : 0;
        first += kcount[nk];
      }
      uint32_t pooled = 0;
      if (e->pool_components)
        for (uint32_t k = 0; k < nk; ++k) {
          pooled += kcount[k];
          e->stats["components_lds_class" + std::to_string(k)] = kcount[k];
        }
      pool_ran = pooled != 0;
      if (pooled) {
        /* one launch for all of them: a workgroup per CU, its wavefronts claim
           components until none is left (k_components_pool) */
        hipStream_t ss = e->side[0];
        uint32_t nbig = 0;   /* components above 8 KB: claimed one at a time */
        for (uint32_t k = 0; k < nk; ++k) if (klass_h[k] > 8192u) nbig += kcount[k];
        /* the fill cursor starts where the footprints fit two pages and moves on
           towards the smallest: the last components claimed are the smallest, so
           a component that turns out slow (not clean: ten times the time per
           contig) cannot start late and set the launch's tail */
        uint32_t g0 = 0;
        for (uint32_t k = 0; k < nk; ++k) if (klass_h[k] > (uint32_t)e->pool_fill_kb * 1024u) g0 += kcount[k];
        if (g0 > pooled) g0 = pooled;
        unsigned long long *cursor = (unsigned long long *)(e->d_scalars + GTS_S_POOLCUR);
        HIPCHK(hipStreamWaitEvent(ss, e->ev_fork, 0));
        HIPCHK(hipMemsetAsync(cursor, 0, 8, ss));
        unsigned long long *pstat = (unsigned long long *)(e->d_scalars + GTS_S_POOLSTAT);
        HIPCHK(hipMemsetAsync(pstat, 0, 96, ss));
        HIPCHK(hipMemsetAsync(pstat + 3, 0xFF, 8, ss));
        HIPCHK(hipMemsetAsync(pstat + 5, 0xFF, 8, ss));
        hipEvent_t _a = nullptr, _b = nullptr;
        if (e->profile) { _a = get_event(e); _b = get_event(e); hipEventRecord(_a, ss); }
        const uint32_t pw = (uint32_t)e->pool_waves;
        /* the view as the pool kernels read it: a copy in device memory (opaque_view) */
        GtsCompView *Cdev = (GtsCompView *)(e->d_scalars + GTS_S_VIEW);
        HIPCHK(hipMemcpyAsync(Cdev, &C, sizeof(GtsCompView), hipMemcpyHostToDevice, ss));
        GtsPoolArgs *PAdev = (GtsPoolArgs *)(e->d_scalars + GTS_S_VIEW + 256), *CAdev = PAdev + 1, *FAdev = PAdev + 2;
        GtsPoolArgs PA;
        PA.order = order; PA.order_key = order_key; PA.first = first; PA.count = pooled; PA.mode = mode;
        PA.ptot = (unsigned long long *)(e->d_scalars + GTS_S_POOLTOT);
        HIPCHK(hipMemsetAsync(PA.ptot, 0, 24, ss));
        lean_pool = e->profile < 2;
        if (lean_pool) {   /* (the programs leave these alone; the walk tasks add to them) */
          HIPCHK(hipMemsetAsync(stat_fast, 0, (size_t)ncomp * 4, ss));
          HIPCHK(hipMemsetAsync(stat_slow, 0, (size_t)ncomp * 4, ss));
          HIPCHK(hipMemsetAsync(stat_clean, 0, (size_t)ncomp * 4, ss));
          HIPCHK(hipMemsetAsync(stat_ncc, 0, (size_t)ncomp * 4, ss));
        }
        PA.cursor = cursor; PA.pstat = pstat; PA.nbig = nbig; PA.g0 = g0; PA.poison = (int)e->lds_poison;
        PA.wait_limit = (uint64_t)e->pool_wait_limit_us * 100ull; PA.cold = nullptr; PA.cold_list = nullptr;
        /* the clean program on two workgroups per CU, the full program next to it on
           `cold_cus` CUs for what the clean one hands over (k_components_fast) */
        fast_ran = use_fast;
        if (fast_ran) {
          /* the fast program writes a component's statistics for the detailed
             profile only; its totals come from its own counters */
          HIPCHK(hipMemsetAsync(stat_fast, 0, (size_t)ncomp * 4, ss));
          HIPCHK(hipMemsetAsync(stat_slow, 0, (size_t)ncomp * 4, ss));
          HIPCHK(hipMemsetAsync(stat_clean, 0, (size_t)ncomp * 4, ss));
          HIPCHK(hipMemsetAsync(stat_ncc, 0, (size_t)ncomp * 4, ss));
          HIPCHK(hipMemsetAsync(tstat, 0, (size_t)ncomp * 40, ss));
          hipStream_t cs = e->side[1];
          const uint32_t cold_wgs = (uint32_t)e->cold_cus;
          const uint32_t fast_wgs = (e->fast_split ? 2u : 1u) * (uint32_t)(e->n_cus - (int)e->cold_cus);
          const uint32_t fast_bytes = e->fast_split ? GTS_FAST_BYTES : GTS_POOL_BYTES;
          unsigned long long *cold = (unsigned long long *)(e->d_scalars + GTS_S_COLD);
          unsigned long long *fstat = (unsigned long long *)(e->d_scalars + GTS_S_FASTSTAT);
          HIPCHK(hipMemsetAsync(cold_list, 0, cold_entries * 8, ss));
          HIPCHK(hipMemsetAsync(fstat, 0, 16 * 8, ss));
          HIPCHK(hipMemsetAsync(fstat + 3, 0xFF, 8, ss));
          HIPCHK(hipMemsetAsync(fstat + 5, 0xFF, 8, ss));
          k_cold_seed<<<1, GTS_BLOCK, 0, ss>>>(order, order_key, first, pooled, cold, cold_list, fast_wgs, fast_bytes);
          HIPCHK(hipEventRecord(e->ev_join[1], ss));
          HIPCHK(hipStreamWaitEvent(cs, e->ev_join[1], 0));
          /* the full program first: its workgroups take their CUs (a CU's whole LDS
             each) before the fast ones fill the others */
          GtsPoolArgs CA = PA;
          CA.cold = cold; CA.cold_list = cold_list;
          hipEvent_t _c = nullptr, _d = nullptr;
          if (e->profile) { _c = get_event(e); _d = get_event(e); hipEventRecord(_c, cs); }
          HIPCHK(hipMemcpyAsync(CAdev, &CA, sizeof CA, hipMemcpyHostToDevice, cs));
          k_components_pool<<<cold_wgs, pw * GTS_WAVE, GTS_POOL_BYTES, cs>>>(Cdev, CAdev);
          if (e->profile) { hipEventRecord(_d, cs);
                            e->pending.push_back({mode == GTS_MODE_MAKESCAFFOLD ? "components_makescaffold_cold"
                                                                                : "components_removecycles_cold",
                                                  _c, _d}); }
          HIPCHK(hipEventRecord(e->ev_join[2], cs));
          GtsPoolArgs FA = CA;
          FA.pstat = fstat;
          if (e->profile) hipEventRecord(_a, ss);
          HIPCHK(hipMemcpyAsync(FAdev, &FA, sizeof FA, hipMemcpyHostToDevice, ss));
          if (!e->fast_split) k_components_fast<<<fast_wgs, (uint32_t)e->pool_waves * GTS_WAVE, GTS_POOL_BYTES, ss>>>(Cdev, FAdev);
          else k_components_fast2<GTS_FAST_WAVES><<<fast_wgs, (uint32_t)e->fast_waves * GTS_WAVE, GTS_FAST_BYTES, ss>>>(Cdev, FAdev);
          if (e->profile) { hipEventRecord(_b, ss);
                            e->pending.push_back({mode == GTS_MODE_MAKESCAFFOLD ? "components_makescaffold_fast"
                                                                                : "components_removecycles_fast",
                                                  _a, _b}); }
          HIPCHK(hipEventRecord(e->ev_join[0], ss));
          HIPCHK(hipStreamWaitEvent(e->st, e->ev_join[0], 0));
          HIPCHK(hipStreamWaitEvent(e->st, e->ev_join[2], 0));
        } else {
        HIPCHK(hipMemcpyAsync(PAdev, &PA, sizeof PA, hipMemcpyHostToDevice, ss));
        k_components_pool<<<e->n_cus, pw * GTS_WAVE, GTS_POOL_BYTES, ss>>>(Cdev, PAdev);
        if (e->profile) { hipEventRecord(_b, ss);
                          e->pending.push_back({mode == GTS_MODE_MAKESCAFFOLD ? "components_makescaffold_pool"
                                                                              : "components_removecycles_pool",
                                                _a, _b}); }
        HIPCHK(hipEventRecord(e->ev_join[0], ss));
        HIPCHK(hipStreamWaitEvent(e->st, e->ev_join[0], 0));
        }
        first += pooled;
      }
      /* pool_components = 0: a launch per size class.  The classes are
         independent: fork them onto side streams so the launches overlap (each
         has its own tail and, being bound by its LDS footprint, leaves room for
         workgroups of the other classes); join before the statistics.  The
         runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware
         queues (default 4, the main stream included): that many launches are
         in flight at a time -- the small classes, most of the wave time, start
         when the large ones are done (17.4 ms against the pool's 14). */
      for (int k = (int)nk - 1; k >= 0 && !pooled; --k) {
        if (!kcount[k]) continue;
        hipStream_t ss = e->side[((int)nk - 1 - k) % (int)e->class_streams];
        HIPCHK(hipStreamWaitEvent(ss, e->ev_fork, 0));
        hipEvent_t _a = nullptr, _b = nullptr;
        if (e->profile) { _a = get_event(e); _b = get_event(e); hipEventRecord(_a, ss); }
        k_components_lds<<<kcount[k], GTS_WAVE, klass_h[k], ss>>>(C, order, first, kcount[k], mode, klass_h[k]);
        if (e->profile) { hipEventRecord(_b, ss);
                          e->pending.push_back({klass_event(mode == GTS_MODE_MAKESCAFFOLD, klass_h[k]), _a, _b}); }
        HIPCHK(hipEventRecord(e->ev_join[k], ss));

        e->stats["components_lds_class" + std::to_string(k)] = kcount[k];
        first += kcount[k];
      }
      /* joins after the last launch: a wait queued earlier would hold back a
         side stream that shares its hardware queue with this stream */
      for (uint32_t k = 0; k < nk && !pooled; ++k)
        if (kcount[k]) HIPCHK(hipStreamWaitEvent(e->st, e->ev_join[k], 0));
      if (e->profile) {
        hipEventRecord(span_b, e->st);
        e->pending.push_back({mode == GTS_MODE_MAKESCAFFOLD ? "span_components_makescaffold"
                                                            : "span_components_removecycles", span_a, span_b});
      }
      hipEvent_t rounds_a = nullptr, rounds_b = nullptr;
      if (e->profile) { rounds_a = get_event(e); rounds_b = get_event(e); hipEventRecord(rounds_a, e->st); }
      /* deferred walks (gts_component.hpp, try_defer): rounds of one workgroup
         per pending walk, grouped by LDS class, and an in-order select pass */
      uint64_t ntasks = 0, walks_run = 0, task_launches = 0;
      uint32_t rounds = 0;
      if (C.defer_min_nv || C.defer_ref_min_nv) {
        uint64_t pend[GTS_NKLASS + 1], ndef = 0;
        HIPCHK(hipMemcpyAsync(pend, e->d_scalars + GTS_S_TQCNT, sizeof pend, hipMemcpyDeviceToHost, e->st));
        HIPCHK(hipMemcpyAsync(&ndef, e->d_scalars + GTS_S_NDEF, 8, hipMemcpyDeviceToHost, e->st));
        if ((rc = read_u64(e, (uint64_t *)(e->d_scalars + 128), &ntasks))) return rc;
        const uint64_t slab_stride = ((((uint64_t)maxcomp + 3) & ~3ull) * GTS_SLAB_BYTES + 255) & ~255ull;
        for (;; ++rounds) {
          const auto round_t0 = std::chrono::steady_clock::now();
          uint64_t total = 0;
          for (uint32_t k = 0; k < nk; ++k) total += pend[k];
          /* pending walks of components that run from global memory (class nk) */
          const uint64_t gpend = e->lds_components ? pend[nk] : 0;
          if (!total && !gpend) break;
          walks_run += total + gpend;
          if (total && total <= (uint64_t)e->mixed_task_limit) {
            /* few walks: one launch, no waiting for a queue per class */
            GtsTaskPrefix P;
            uint32_t acc = 0, kmax = 0;
            for (uint32_t k = 0; k < GTS_NKLASS; ++k) {
              P.pre[k] = acc;
              if (k < nk && pend[k]) { acc += (uint32_t)pend[k]; kmax = k; }
            }
            P.pre[GTS_NKLASS] = acc;
            ++task_launches;
            hipEvent_t _a = nullptr, _b = nullptr;
            if (e->profile) { _a = get_event(e); _b = get_event(e); hipEventRecord(_a, e->st); }
            const uint32_t lds_m = klass_h[kmax + 1 < nk ? kmax + 1 : kmax];   /* a class up: room for the split arcs */
            k_walk_tasks_mixed<<<acc, GTS_WAVE, lds_m, e->st>>>(C, P, lds_m);
            if (e->profile) { hipEventRecord(_b, e->st); e->pending.push_back({"components_walk_tasks", _a, _b}); }
            for (uint32_t k = 0; k < nk; ++k) pend[k] = 0;   /* nothing to join */
          }
          for (uint32_t k = 0; k < nk; ++k) task_launches += pend[k] != 0;
          HIPCHK(hipEventRecord(e->ev_fork, e->st));
          if (gpend) {
            /* on the main stream, next to the class launches on the side streams;
               as many walks at a time as the slab pool holds */
            uint64_t cap = ((uint64_t)e->global_task_pool_mb << 20) / slab_stride;
            if (cap < 1) cap = 1;
            if (cap > gpend) cap = gpend;
            if ((rc = dev_alloc(e, &e->gtask_pool, (size_t)(cap * slab_stride)))) return rc;
            for (uint64_t first = 0; first < gpend; first += cap) {
              const uint32_t cnt = (uint32_t)(gpend - first < cap ? gpend - first : cap);
              ++task_launches;
              LAUNCH("components_walk_tasks_global", k_walk_tasks_global, cnt, GTS_WAVE, C, (uint32_t)nk,
                     (uint32_t)first, cnt, e->gtask_pool, slab_stride);
            }
          }
          for (int k = (int)nk - 1; k >= 0; --k) {
            if (!pend[k]) continue;
            hipStream_t ss = e->side[((int)nk - 1 - k) % (int)e->class_streams];
            HIPCHK(hipStreamWaitEvent(ss, e->ev_fork, 0));
            hipEvent_t _a = nullptr, _b = nullptr;
            if (e->profile) { _a = get_event(e); _b = get_event(e); hipEventRecord(_a, ss); }
            const uint32_t lds_k = klass_h[(uint32_t)k + 1 < nk ? k + 1 : k];
            k_walk_tasks<<<(uint32_t)pend[k], GTS_WAVE, lds_k, ss>>>(C, (uint32_t)k, (uint32_t)pend[k], lds_k);
            if (e->profile) { hipEventRecord(_b, ss); e->pending.push_back({"components_walk_tasks", _a, _b}); }
            HIPCHK(hipEventRecord(e->ev_join[k], ss));
          }
          for (uint32_t k = 0; k < nk; ++k)
            if (pend[k]) HIPCHK(hipStreamWaitEvent(e->st, e->ev_join[k], 0));
          HIPCHK(hipMemsetAsync(e->d_scalars + GTS_S_TQCNT, 0, sizeof pend, e->st));
          LAUNCH("components_select_walks", k_select_walks, (uint32_t)ndef, GTS_WAVE, C, (uint32_t)ndef);
          HIPCHK(hipMemcpyAsync(pend, e->d_scalars + GTS_S_TQCNT, sizeof pend, hipMemcpyDeviceToHost, e->st));
          if ((rc = sync_stream(e))) return rc;
          if (e->profile >= 2 && rounds < 8) {   /* host clock: a round ends with a look at the queues anyway */
            e->stats["walk_round" + std::to_string(rounds) + "_us"] = (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(
                std::chrono::steady_clock::now() - round_t0).count();
            e->stats["walk_round" + std::to_string(rounds) + "_walks"] = (int64_t)(total + gpend);
          }
        }
      }
      if (e->profile) {
        hipEventRecord(rounds_b, e->st);
        e->pending.push_back({"span_walk_rounds", rounds_a, rounds_b});
      }
      e->stats["walk_task_rounds"] = rounds;
      e->stats["walk_task_launches"] = (int64_t)task_launches;
      {
        uint64_t tb = 0;
        if ((rc = read_u64(e, (uint64_t *)(e->d_scalars + GTS_S_NDEF + 2), &tb))) return rc;
        e->stats["bytes_walk_tasks"] = (int64_t)tb;
      }
      e->stats["walk_task_runs"] = (int64_t)walks_run;
      e->stats["walk_tasks"] = (int64_t)ntasks;
      e->stats["components_global_mem"] = kcount[nk];
      e->stats["bytes_components_global_mem"] = (int64_t)kbytes[nk];
      for (uint32_t k = 0; k < nk; ++k)
        e->stats["bytes_components_lds_class" + std::to_string(k)] = (int64_t)kbytes[k];
    }
    if (team_ran) HIPCHK(hipStreamWaitEvent(e->st, e->ev_team, 0));
    LAUNCH("comp_count_errors", k_count_errors, nblk(ncomp), GTS_BLOCK, cerr, ncomp,
           e->d_scalars + 12);
    HIPCHK(hipMemsetAsync(e->d_scalars + 16, 0, 16, e->st));
    LAUNCH("comp_walk_stats", k_sum_u32, nblk(ncomp), GTS_BLOCK, stat_fast, ncomp,
           (unsigned long long *)(e->d_scalars + 16));
    LAUNCH("comp_walk_stats", k_sum_u32, nblk(ncomp), GTS_BLOCK, stat_slow, ncomp,
           (unsigned long long *)(e->d_scalars + 18));
    HIPCHK(hipMemsetAsync(e->d_scalars + 20, 0, 16, e->st));
    LAUNCH("comp_walk_stats", k_sum_bit, nblk(ncomp), GTS_BLOCK, stat_clean, ncomp, 0u,
           (unsigned long long *)(e->d_scalars + 20));
    LAUNCH("comp_walk_stats", k_sum_bit, nblk(ncomp), GTS_BLOCK, stat_clean, ncomp, 1u,
           (unsigned long long *)(e->d_scalars + 22));
    HIPCHK(hipMemsetAsync(e->d_scalars + 32, 0, 64, e->st));
    if (e->profile >= 2)   /* the per-component clocks are reduced for the detailed profile only */
      LAUNCH("comp_walk_stats", k_tstat_reduce, nblk(ncomp), GTS_BLOCK, tstat, ncomp,
             (unsigned long long *)(e->d_scalars + 32));
    uint64_t ts[8];
    HIPCHK(hipMemcpyAsync(ts, e->d_scalars + 32, 64, hipMemcpyDeviceToHost, e->st));
    uint64_t why[12];
    HIPCHK(hipMemcpyAsync(why, e->d_scalars + 96, 96, hipMemcpyDeviceToHost, e->st));
    uint64_t pst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (pool_ran) HIPCHK(hipMemcpyAsync(pst, e->d_scalars + GTS_S_POOLSTAT, 96, hipMemcpyDeviceToHost, e->st));
    uint64_t ptot[3] = {0, 0, 0};
    if (pool_ran) HIPCHK(hipMemcpyAsync(ptot, e->d_scalars + GTS_S_POOLTOT, 24, hipMemcpyDeviceToHost, e->st));
    uint64_t fst[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fast_ran) HIPCHK(hipMemcpyAsync(fst, e->d_scalars + GTS_S_FASTSTAT, 128, hipMemcpyDeviceToHost, e->st));
    uint64_t tst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (team_ran) HIPCHK(hipMemcpyAsync(tst, e->d_scalars + GTS_S_TEAMSTAT, 64, hipMemcpyDeviceToHost, e->st));
    uint32_t res[4] = {0, 0, 0, 0};
    uint64_t wstat[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(res, e->d_scalars + 12, 16, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(wstat, e->d_scalars + 16, 32, hipMemcpyDeviceToHost, e->st));
    if ((rc = sync_stream(e))) return rc;
    for (int k = 6; k < 10; ++k) pst[k] += fst[k];   /* either kernel's waits and overruns */
    e->stats["components_ring_overflow"] = res[0];
    e->stats["components_walk_error"] = res[1];
    e->stats["components_path_overflow"] = res[3];
    if (pst[6] | pst[7] | pst[8] | pst[9]) {
      /* some components have written their marks, others have not run: the graph
         goes back to its state before the call (as for GTSG_EWALK), the caller
         may call again */
      e->stats["pool_lds_overruns"] = (int64_t)pst[9];
      e->stats["pool_gave_up_lock"] = (int64_t)pst[6];
      e->stats["pool_gave_up_claim"] = (int64_t)pst[7];
      e->stats["pool_gave_up_pages"] = (int64_t)pst[8];
      HIPCHK(hipMemcpyAsync(e->vstate, snap_v, n, hipMemcpyDeviceToDevice, e->st));
      if (m) HIPCHK(hipMemcpyAsync(e->state, snap_e, m, hipMemcpyDeviceToDevice, e->st));
      HIPCHK(hipStreamSynchronize(e->st));
      if (pst[9])
        return fail(e, GTSG_EINTERNAL, "component pool: %llu program(s) wrote past their LDS arrays; "
                    "states restored", (unsigned long long)pst[9]);
      return fail(e, GTSG_EINTERNAL, "component pool: a wavefront gave up waiting (lock %llu, claim %llu, "
                  "pages %llu); states restored", (unsigned long long)pst[6], (unsigned long long)pst[7],
                  (unsigned long long)pst[8]);
    }
    if (C.small_stat) {
      uint64_t ss[4] = {0, 0, 0, 0};
      HIPCHK(hipMemcpy(ss, C.small_stat, 32, hipMemcpyDeviceToHost));
      e->stats["small_all_live_components"] = (int64_t)ss[0];
      e->stats["small_other_components"] = (int64_t)ss[1];
      e->stats["small_all_live_removecycles_us"] = (int64_t)(ss[2] / 100);
      e->stats["small_other_removecycles_us"] = (int64_t)(ss[3] / 100);
    }
    if (team_ran) {
      static const char *tn[8] = {"team_ccs", "team_batches", "team_sweep_steps", "team_ccs_with_tie",
                                  "team_us_clear", "team_us_sweep", "team_us_paths", "team_us_wave0_barriers"};
      for (int k = 0; k < 8; ++k) e->stats[tn[k]] = (int64_t)(k < 4 ? tst[k] : tst[k] / 100);
    }
    if (pool_ran) {   /* 100 MHz ticks -> microseconds */
      e->stats["pool_us_sum_run"] = (int64_t)(pst[0] / 100);
      e->stats["pool_us_sum_wait_pages"] = (int64_t)(pst[1] / 100);
      e->stats["pool_us_sum_wave_life"] = (int64_t)(pst[2] / 100);
      e->stats["pool_us_first_exit"] = (int64_t)((pst[3] - pst[5]) / 100);
      e->stats["pool_us_last_exit"] = (int64_t)((pst[4] - pst[5]) / 100);
      e->stats["pool_helper_joins"] = (int64_t)pst[10];
      e->stats["pool_us_sum_helping"] = (int64_t)(pst[11] / 100);
    }
    e->stats["fast_kernel"] = fast_ran ? 1 : 0;
    if (fast_ran) {
      e->stats["fast_us_sum_run"] = (int64_t)(fst[0] / 100);
      e->stats["fast_us_sum_wait_pages"] = (int64_t)(fst[1] / 100);
      e->stats["fast_us_sum_claim"] = (int64_t)(fst[15] / 100);
      e->stats["fast_us_sum_wave_life"] = (int64_t)(fst[2] / 100);
      e->stats["fast_us_first_exit"] = (int64_t)((fst[3] - fst[5]) / 100);
      e->stats["fast_us_last_exit"] = (int64_t)((fst[4] - fst[5]) / 100);
      e->stats["fast_components_done"] = (int64_t)(fst[10] & 0xFFFFFFFFull);
      e->stats["fast_components_handed_over"] = (int64_t)fst[12];
      e->stats["bytes_fast_finished"] = (int64_t)fst[13];
      e->stats["bytes_fast_handed_over"] = (int64_t)fst[14];
      /* the cold workgroups start first; their clocks against the fast kernel's first start */
      e->stats["cold_us_last_exit_after_fast_start"] = (int64_t)(((int64_t)pst[4] - (int64_t)fst[5]) / 100);
      e->stats["fast_wavefronts"] = (int64_t)(e->n_cus - (int)e->cold_cus) *
                                    (e->fast_split ? 2 * e->fast_waves : e->pool_waves);
    }
    {
      static const char *nm[4] = {"removecycles", "makescaffold_other", "walks_fast", "walks_reference"};
      for (int k = 0; k < 4; ++k) {   /* 100 MHz ticks -> microseconds */
        e->stats[std::string("us_sum_") + nm[k]] = (int64_t)(ts[k] / 100);
        e->stats[std::string("us_max_") + nm[k]] = (int64_t)(ts[4 + k] / 100);
      }
    }
    if (e->profile >= 2) {   /* the three components that took longest */
      std::vector<uint64_t> ht(5 * (size_t)ncomp);
      std::vector<uint32_t> hs(ncomp), hf(ncomp), hc(ncomp), ho((size_t)ncomp + 1), hco((size_t)nslots + 1), hn(ncomp);
      HIPCHK(hipMemcpy(hn.data(), stat_ncc, (size_t)ncomp * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(ht.data(), tstat, ht.size() * 8, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(hs.data(), stat_slow, (size_t)ncomp * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(hf.data(), stat_fast, (size_t)ncomp * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(hc.data(), stat_clean, (size_t)ncomp * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(ho.data(), comp_off, ((size_t)ncomp + 1) * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(hco.data(), coff, ((size_t)nslots + 1) * 4, hipMemcpyDeviceToHost));
      if (const char *dump = getenv("GTS_DUMP_COMPONENTS")) {   /* input of tools/sim/pack_sim.py: contigs, edges, clocks per component */
        FILE *f = fopen(dump, "wb");
        if (f) {
          for (uint32_t c2 = 0; c2 < ncomp; ++c2) {
            uint64_t rec[6] = {ho[c2 + 1] - ho[c2], hco[ho[c2 + 1]] - hco[ho[c2]], ht[5 * (size_t)c2],
                               ht[5 * (size_t)c2 + 1], ht[5 * (size_t)c2 + 2], ht[5 * (size_t)c2 + 3]};
            fwrite(rec, 8, 6, f);
          }
          fclose(f);
        }
      }
      {   /* component sizes and the time spent per size band */
        static const uint32_t band[8] = {2, 3, 4, 8, 16, 32, 64, 0xFFFFFFFFu};
        uint64_t bc[8] = {0}, bt[8] = {0}, bw[8] = {0};
        for (uint32_t c2 = 0; c2 < ncomp; ++c2) {
          const uint32_t sz = ho[c2 + 1] - ho[c2];
          int b = 0;
          while (sz > band[b]) ++b;
          ++bc[b];
          const uint64_t *t = &ht[5 * (size_t)c2];
          bt[b] += t[0] + t[1] + t[2] + t[3];
          bw[b] += t[2] + t[3];
        }
        {   /* the same per LDS size class ([GTS_NKLASS] = global memory) */
          std::vector<uint8_t> hk((size_t)ncomp);
          HIPCHK(hipMemcpy(hk.data(), comp_klass, (size_t)ncomp, hipMemcpyDeviceToHost));
          uint64_t kt[GTS_NKLASS + 1] = {0}, kw[GTS_NKLASS + 1] = {0};
          for (uint32_t c2 = 0; c2 < ncomp; ++c2) {
            const uint64_t *t = &ht[5 * (size_t)c2];
            const int k = hk[c2] <= GTS_NKLASS ? hk[c2] : GTS_NKLASS;
            kt[k] += t[0] + t[1] + t[2] + t[3];
            kw[k] += t[2] + t[3];
          }
          for (int k = 0; k <= GTS_NKLASS; ++k) {
            e->stats["lds_class" + std::to_string(k) + "_wave_us"] = (int64_t)(kt[k] / 100);
            e->stats["lds_class" + std::to_string(k) + "_walk_us"] = (int64_t)(kw[k] / 100);
          }
        }
        for (int b = 0; b < 8; ++b) {
          e->stats["size_band" + std::to_string(b) + "_components"] = (int64_t)bc[b];
          e->stats["size_band" + std::to_string(b) + "_us"] = (int64_t)(bt[b] / 100);
          e->stats["size_band" + std::to_string(b) + "_walk_us"] = (int64_t)(bw[b] / 100);
        }
      }
      if (C.tspan) {   /* who finishes last: start and end of the component programs since the first start */
        std::vector<uint64_t> sp(2 * (size_t)ncomp);
        HIPCHK(hipMemcpy(sp.data(), C.tspan, sp.size() * 8, hipMemcpyDeviceToHost));
        uint64_t t_min = ~0ull;
        for (uint32_t c2 = 0; c2 < ncomp; ++c2) if (sp[2 * (size_t)c2] && sp[2 * (size_t)c2] < t_min) t_min = sp[2 * (size_t)c2];
        std::vector<uint32_t> idx(ncomp);
        for (uint32_t c2 = 0; c2 < ncomp; ++c2) idx[c2] = c2;
        const size_t top = ncomp < 8 ? ncomp : 8;
        std::partial_sort(idx.begin(), idx.begin() + top, idx.end(),
                          [&](uint32_t a, uint32_t b) { return sp[2 * (size_t)a + 1] > sp[2 * (size_t)b + 1]; });
        for (size_t r = 0; r < top; ++r) {
          const uint32_t c2 = idx[r];
          const std::string pre = "last" + std::to_string(r) + "_";
          e->stats[pre + "size"] = ho[c2 + 1] - ho[c2];
          e->stats[pre + "terminals"] = hc[c2] >> 8;
          e->stats[pre + "clean"] = hc[c2] & 1;
          e->stats[pre + "start_us"] = (int64_t)((sp[2 * (size_t)c2] - t_min) / 100);
          e->stats[pre + "end_us"] = (int64_t)((sp[2 * (size_t)c2 + 1] - t_min) / 100);
          e->stats[pre + "removecycles_us"] = (int64_t)(ht[5 * (size_t)c2] / 100);
          e->stats[pre + "walks_us"] = (int64_t)((ht[5 * (size_t)c2 + 2] + ht[5 * (size_t)c2 + 3]) / 100);
        }
      }
      auto total = [&](uint32_t c2) { const uint64_t *t = &ht[5 * (size_t)c2]; return t[0] + t[1] + t[2] + t[3]; };
      for (int r = 0; r < 12 && r < (int)ncomp; ++r) {
        uint32_t best = 0;
        for (uint32_t c2 = 1; c2 < ncomp; ++c2) if (total(c2) > total(best)) best = c2;
        const std::string pre = "top" + std::to_string(r) + "_";
        e->stats[pre + "size"] = ho[best + 1] - ho[best];
        e->stats[pre + "edges"] = hco[ho[best + 1]] - hco[ho[best]];
        e->stats[pre + "terminals"] = hc[best] >> 8;
        e->stats[pre + "ccs"] = hn[best];
        e->stats[pre + "clean"] = hc[best] & 1;
        e->stats[pre + "deferred"] = hc[best] >> 1 & 1;
        e->stats[pre + "why_not_deferred"] = hc[best] >> 2 & 3;
        e->stats[pre + "walks"] = hf[best] + hs[best];
        e->stats[pre + "ref_walks"] = hs[best];
        e->stats[pre + "removecycles_us"] = (int64_t)(ht[5 * (size_t)best] / 100);
        e->stats[pre + "other_us"] = (int64_t)(ht[5 * (size_t)best + 1] / 100);
        e->stats[pre + "walks_us"] = (int64_t)(ht[5 * (size_t)best + 2] / 100);
        e->stats[pre + "ref_us"] = (int64_t)(ht[5 * (size_t)best + 3] / 100);
        e->stats[pre + "ref_pops"] = (int64_t)ht[5 * (size_t)best + 4];
        for (int k = 0; k < 4; ++k) ht[5 * (size_t)best + k] = 0;
      }
    }
    {
      static const char *wn[8] = {"why_mixed_start", "why_self_arc", "why_back_at_start",
                                  "why_marked_end", "why_two_directions", "why_inexact_tie",
                                  "why_cycle", "why_inexact_length_tie"};
      for (int k = 0; k < 8; ++k) e->stats[wn[k]] = (int64_t)why[k];
      e->stats["unclean_batch_walks"] = (int64_t)why[8];
      e->stats["unclean_batch_given_up"] = (int64_t)why[9];
      e->stats["unclean_batch_cycle"] = (int64_t)why[10];
    }
    /* (the fast program counts in its own words unless the detailed profile made
       it write the per-component tables the sums above are taken from) */
    if (fast_ran && e->profile < 2) { wstat[0] += fst[11]; wstat[2] += fst[10] >> 32; }
    if (pool_ran && lean_pool) {
      wstat[0] += ptot[1] & 0xFFFFFFFFull; wstat[1] += ptot[1] >> 32; wstat[2] += ptot[0] >> 32; wstat[3] += ptot[2];
    }
    e->stats["fast_walks"] = (int64_t)wstat[0];
    e->stats["slow_walks"] = (int64_t)wstat[1];
    e->stats["clean_components"] = (int64_t)wstat[2];
    e->stats["deferred_components"] = (int64_t)wstat[3];
    e->stats["components"] = ncomp;
    e->stats["max_component"] = res[2];
    e->stats["compact_edges"] = nce;
    if (res[1]) {
      /* the graph goes back to its state before the call: some components have
         written their marks, others have not */
      HIPCHK(hipMemcpyAsync(e->vstate, snap_v, n, hipMemcpyDeviceToDevice, e->st));
      if (m) HIPCHK(hipMemcpyAsync(e->state, snap_e, m, hipMemcpyDeviceToDevice, e->st));
      HIPCHK(hipStreamSynchronize(e->st));
      return fail(e, GTSG_EWALK, "%u components exceeded max_walk_pops=%lld or hold a "
                  "cyclic distance map; states restored", res[1], (long long)e->max_walk_pops);
    }
    if (!res[0] && !res[3]) break;
    /* the pool of walk rings or the pool of task paths was too small somewhere:
       restore and run again with more of what ran out (a single ring that is
       too small is replaced on the device, create_walk_reference) */
    HIPCHK(hipMemcpyAsync(e->vstate, snap_v, n, hipMemcpyDeviceToDevice, e->st));
    if (m) HIPCHK(hipMemcpyAsync(e->state, snap_e, m, hipMemcpyDeviceToDevice, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    if (res[0]) pool_entries *= 8;
    if (res[3]) path_entries *= 4;
    if (++retries > 6) return fail(e, GTSG_EWALK, "walk queues keep overflowing");
  }
  e->stats["walk_retries"] = retries;
  return sync_stream(e);
}

int gtsg_removecycles(GtsgEngine *e) { return run_components(e, GTS_MODE_REMOVECYCLES); }
int gtsg_makescaffold(GtsgEngine *e) { return run_components(e, GTS_MODE_MAKESCAFFOLD); }

uint64_t gtsg_num_vertices(const GtsgEngine *e) { return e ? e->n : 0; }
uint64_t gtsg_num_edges(const GtsgEngine *e) { return e ? e->m : 0; }

int gtsg_get_vertex_states(GtsgEngine *e, uint8_t *out)
{
  if (!e || !out) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (e->n) HIPCHK(hipMemcpyAsync(out, e->vstate, e->n, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

int gtsg_get_edge_states(GtsgEngine *e, uint8_t *out)
{
  if (!e || !out) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (!e->m) return 0;
  int rc;
  if (e->pool_cap < (size_t)e->m + 4096) { if ((rc = pool_reserve(e, (size_t)e->m + 4096))) return rc; }
  else e->pool_used = 0;
  PALLOC(tmp, uint8_t, e->m);
  LAUNCH("states_by_id", k_states_by_id, nblk(e->m), GTS_BLOCK, e->state, e->eid, tmp, e->m);
  HIPCHK(hipMemcpyAsync(out, tmp, e->m, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

int gtsg_get_edges(GtsgEngine *e, uint32_t *start, uint32_t *end, int64_t *dist,
                   float *std_dev, int64_t *num_pairs, uint8_t *flags)
{
  if (!e || !start || !end || !dist || !std_dev || !num_pairs || !flags) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  const uint32_t m = e->m;
  if (!m) return 0;
  int rc;
  const size_t need = (size_t)m * 32 + (1u << 20);
  if (e->pool_cap < need) { if ((rc = pool_reserve(e, need))) return rc; }
  else e->pool_used = 0;
  PALLOC(os, uint32_t, m); PALLOC(oe, uint32_t, m); PALLOC(od, int64_t, m);
  PALLOC(osd, float, m); PALLOC(onp, int64_t, m); PALLOC(of, uint8_t, m);
  LAUNCH("edges_by_id", k_edges_by_id, nblk(m), GTS_BLOCK, e->eid, e->estart, e->eend,
         e->dist, e->sd, e->npairs, e->flags, os, oe, od, osd, onp, of, m);
  HIPCHK(hipMemcpyAsync(start, os, (size_t)m * 4, hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipMemcpyAsync(end, oe, (size_t)m * 4, hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipMemcpyAsync(dist, od, (size_t)m * 8, hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipMemcpyAsync(std_dev, osd, (size_t)m * 4, hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipMemcpyAsync(num_pairs, onp, (size_t)m * 8, hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipMemcpyAsync(flags, of, (size_t)m, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

/* adjacency lists as the reference keeps them: row[v] .. row[v + 1] index adj[],
   adj[] holds the edge ids of vertex v in creation order */
int gtsg_get_csr(GtsgEngine *e, uint32_t *row, uint32_t *adj)
{
  if (!e || !row || !adj) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (!e->row) return fail(e, GTSG_EINVAL, "no graph has been built");
  HIPCHK(hipMemcpyAsync(row, e->row, ((size_t)e->n + 1) * 4, hipMemcpyDeviceToHost, e->st));
  if (e->m) HIPCHK(hipMemcpyAsync(adj, e->eid, (size_t)e->m * 4, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

/* ref gt_scaffolder_graph.c:174-193 (find_edge): the first edge of vertex_1's
   list -- creation order -- that ends in vertex_2.  Reads the one list back. */
int gtsg_find_edge(GtsgEngine *e, uint64_t vertex_1, uint64_t vertex_2, uint64_t *eid)
{
  if (!e || !eid) return GTSG_EINVAL;
  *eid = GTSG_NO_EDGE;
  HIPCHK(hipSetDevice(e->device));
  if (!e->built) return fail(e, GTSG_EINVAL, "graph not built");
  if (vertex_1 >= e->n || vertex_2 >= e->n) return fail(e, GTSG_EINVAL, "vertex out of range");
  uint32_t be[2];
  HIPCHK(hipMemcpyAsync(be, e->row + vertex_1, 8, hipMemcpyDeviceToHost, e->st));
  int rc;
  if ((rc = sync_stream(e))) return rc;
  const size_t deg = be[1] - be[0];
  if (!deg) return 0;
  std::vector<uint32_t> ends(deg), ids(deg);
  HIPCHK(hipMemcpyAsync(ends.data(), e->eend + be[0], deg * 4, hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipMemcpyAsync(ids.data(), e->eid + be[0], deg * 4, hipMemcpyDeviceToHost, e->st));
  if ((rc = sync_stream(e))) return rc;
  for (size_t k = 0; k < deg; ++k)
    if (ends[k] == vertex_2) { *eid = ids[k]; break; }
  return 0;
}

/* ref gt_scaffolder_graph.c:219-235 (alter_edge): new attributes for one edge;
   its state and its twin's attributes stay.  The stored flags carry a bit
   derived from the senses of an edge and its twin (k_emit_edges), which is
   redone for both. */
int gtsg_alter_edge(GtsgEngine *e, uint64_t eid, int64_t dist, float std_dev, uint64_t num_pairs,
                    int sense, int same)
{
  if (!e) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (!e->built) return fail(e, GTSG_EINVAL, "graph not built");
  if (eid >= e->m) return fail(e, GTSG_EINVAL, "edge out of range");
  if (e->filter_open) return fail(e, GTSG_EINVAL, "gtsg_filter_begin without gtsg_filter_end");
  int rc;
  uint32_t p = 0, t = 0;
  uint8_t fp = 0, ft = 0;
  HIPCHK(hipMemcpyAsync(&p, e->pos_of_eid + eid, 4, hipMemcpyDeviceToHost, e->st));
  if ((rc = sync_stream(e))) return rc;
  HIPCHK(hipMemcpyAsync(&t, e->twin + p, 4, hipMemcpyDeviceToHost, e->st));
  if ((rc = sync_stream(e))) return rc;
  HIPCHK(hipMemcpyAsync(&ft, e->flags + t, 1, hipMemcpyDeviceToHost, e->st));
  if ((rc = sync_stream(e))) return rc;
  fp = (uint8_t)((sense ? GTS_F_SENSE : 0u) | (same ? GTS_F_SAME : 0u));
  ft &= 3u;
  if (((ft & GTS_F_SENSE) != 0) == gts_next_dir(fp)) fp |= GTS_F_UTURN;
  if (((fp & GTS_F_SENSE) != 0) == gts_next_dir(ft)) ft |= GTS_F_UTURN;
  const int64_t np = (int64_t)num_pairs;
  HIPCHK(hipMemcpyAsync(e->dist + p, &dist, 8, hipMemcpyHostToDevice, e->st));
  HIPCHK(hipMemcpyAsync(e->sd + p, &std_dev, 4, hipMemcpyHostToDevice, e->st));
  HIPCHK(hipMemcpyAsync(e->npairs + p, &np, 8, hipMemcpyHostToDevice, e->st));
  HIPCHK(hipMemcpyAsync(e->flags + p, &fp, 1, hipMemcpyHostToDevice, e->st));
  if (t != p) HIPCHK(hipMemcpyAsync(e->flags + t, &ft, 1, hipMemcpyHostToDevice, e->st));
  return sync_stream(e);
}

/* ---- .dot text of the edges, formatted on the device ---------------------------
   ref gt_scaffolder_graph.c:288-300: one line per edge, in edge-id order,
     <start> -> <end> [color="<colour of the state>" label="<dist>" arrowhead="normal"|"inv"];
   A thread per edge: the length of its line, a prefix sum, then the bytes. */
__device__ __forceinline__ uint32_t dec_len(uint64_t v)
{
  uint32_t k = 1;
  while (v >= 10) { v /= 10; ++k; }
  return k;
}
__device__ __forceinline__ char *dec_put(char *p, uint64_t v)
{
  const uint32_t k = dec_len(v);
  for (uint32_t i = k; i-- > 0;) { p[i] = (char)('0' + v % 10); v /= 10; }
  return p + k;
}
__device__ __forceinline__ char *str_put(char *p, const char *s, uint32_t n)
{
  for (uint32_t i = 0; i < n; ++i) p[i] = s[i];
  return p + n;
}
/* colours of the states 0..7 (graph.c:277-278) and their lengths */
__constant__ char gts_dot_color[8][12] = {"black", "gray80", "gainsboro", "ivory3", "red", "green", "magenta", "blue"};
__constant__ uint8_t gts_dot_clen[8] = {5, 6, 9, 6, 3, 5, 7, 4};
__global__ void k_dot_len(const uint32_t *pos_of_eid, const uint32_t *estart, const uint32_t *eend,
                          const int64_t *dist, const uint8_t *flags, const uint8_t *state,
                          uint64_t first, uint32_t count, uint32_t *len)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t p = pos_of_eid[first + i];
  const int64_t d = dist[p];
  const uint64_t ad = d < 0 ? (uint64_t)0 - (uint64_t)d : (uint64_t)d;
  len[i] = dec_len(estart[p]) + 4u + dec_len(eend[p]) + 9u + gts_dot_clen[state[p] & 7u] + 9u +
           (d < 0 ? 1u : 0u) + dec_len(ad) + ((flags[p] & 1u) ? 23u : 20u);
}
__global__ void k_dot_write(const uint32_t *pos_of_eid, const uint32_t *estart, const uint32_t *eend,
                            const int64_t *dist, const uint8_t *flags, const uint8_t *state,
                            uint64_t first, uint32_t count, const uint32_t *off, char *text)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t p = pos_of_eid[first + i];
  char *o = text + off[i];
  const int64_t d = dist[p];
  const uint32_t st = state[p] & 7u;
  o = dec_put(o, estart[p]);
  o = str_put(o, " -> ", 4);
  o = dec_put(o, eend[p]);
  o = str_put(o, " [color=\"", 9);
  o = str_put(o, gts_dot_color[st], gts_dot_clen[st]);
  o = str_put(o, "\" label=\"", 9);
  if (d < 0) *o++ = '-';
  o = dec_put(o, d < 0 ? (uint64_t)0 - (uint64_t)d : (uint64_t)d);
  if (flags[p] & 1u) str_put(o, "\" arrowhead=\"normal\"];\n", 23);
  else str_put(o, "\" arrowhead=\"inv\"];\n", 20);
}
static int format_dot_edges(GtsgEngine *e, uint64_t first, uint64_t count, char *host_buf, uint64_t cap,
                            uint64_t *nbytes, const char **pinned);
int gtsg_format_dot_edges(GtsgEngine *e, uint64_t first, uint64_t count, char *host_buf, uint64_t cap,
                          uint64_t *nbytes)
{
  if (!e || !nbytes || (count && !host_buf)) return GTSG_EINVAL;
  return format_dot_edges(e, first, count, host_buf, cap, nbytes, nullptr);
}
/* the same into a page-locked buffer of the engine's (valid until the next
   call): the copy from the device runs at the bus' rate and the caller writes
   the file straight from it */
int gtsg_format_dot_edges_pinned(GtsgEngine *e, uint64_t first, uint64_t count, const char **text,
                                 uint64_t *nbytes)
{
  if (!e || !nbytes || !text) return GTSG_EINVAL;
  *text = nullptr;
  return format_dot_edges(e, first, count, nullptr, 0, nbytes, text);
}
static int format_dot_edges(GtsgEngine *e, uint64_t first, uint64_t count, char *host_buf, uint64_t cap,
                            uint64_t *nbytes, const char **pinned)
{
  *nbytes = 0;
  HIPCHK(hipSetDevice(e->device));
  if (!e->built) return fail(e, GTSG_EINVAL, "graph not built");
  if (first > e->m || count > e->m - first) return fail(e, GTSG_EINVAL, "edge range out of bounds");
  if (!count) return 0;
  if (count > (1u << 25)) return fail(e, GTSG_ELIMIT, "at most 2^25 edges a call (32-bit text offsets)");
  int rc;
  const uint32_t cnt = (uint32_t)count;
  /* (the workspace is the open filter call's too: pool_reserve would drop it) */
  if (e->filter_open) return fail(e, GTSG_EINVAL, "gtsg_filter_begin without gtsg_filter_end");
  if ((rc = pool_reserve(e, (size_t)cnt * 8 + gts_scan_tmp_elems((uint64_t)cnt + 16) * 4 + (1u << 20)))) return rc;
  PALLOC(len, uint32_t, (size_t)cnt + 1); PALLOC(off, uint32_t, (size_t)cnt + 2);
  PALLOC(sctmp, uint32_t, gts_scan_tmp_elems((uint64_t)cnt + 16));
  LAUNCH("dot_len", k_dot_len, nblk(cnt), GTS_BLOCK, e->pos_of_eid, e->estart, e->eend, e->dist, e->flags,
         e->state, first, cnt, len);
  gts_exscan<uint32_t, uint32_t>(len, off, cnt, sctmp, off + cnt, e->st);
  uint32_t total = 0;
  if ((rc = read_u32(e, off + cnt, &total))) return rc;
  *nbytes = total;
  if (pinned) {
    if (e->text_host_cap < (size_t)total + 64) {
      if (e->text_host) hipHostFree(e->text_host);
      e->text_host = nullptr; e->text_host_cap = 0;
      const size_t want = (size_t)total + total / 8 + 4096;
      if (hipHostMalloc((void **)&e->text_host, want, hipHostMallocDefault) != hipSuccess)
        return fail(e, GTSG_ENOMEM, "page-locked text buffer of %zu bytes", want);
      e->text_host_cap = want;
    }
    host_buf = e->text_host;
    *pinned = e->text_host;
  } else if (total > cap)
    return fail(e, GTSG_EINVAL, "text buffer too small: %u bytes needed", total);
  if ((rc = dev_alloc(e, &e->text_buf, (size_t)total + 64))) return rc;
  LAUNCH("dot_write", k_dot_write, nblk(cnt), GTS_BLOCK, e->pos_of_eid, e->estart, e->eend, e->dist, e->flags,
         e->state, first, cnt, off, e->text_buf);
  HIPCHK(hipMemcpyAsync(host_buf, e->text_buf, total, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

/* ---- the SCAFFOLD edges only (output side of the file API) ------------------
   The scaffold record walk (ref algorithms.c:901-997) and the .scaf writer look
   at SCAFFOLD edges and nothing else: a compact CSR of them -- adjacency order
   kept, a few per cent of the edges -- instead of the whole edge list. */
__global__ void k_scaf_flags(const uint8_t *state, uint8_t *flag, uint32_t m)
{
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < m) flag[p] = state[p] == GIS_SCAFFOLD ? 1 : 0;
}
__global__ void k_scaf_rows(const uint32_t *row, const uint32_t *ipos, uint32_t *srow, uint32_t n)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v <= n) srow[v] = ipos[row[v]];
}
__global__ void k_scaf_fill(const uint8_t *flag, const uint32_t *ipos, const uint32_t *eid,
                            const uint32_t *eend, const int64_t *dist, const float *sd,
                            const uint8_t *flags, uint32_t *o_eid, uint32_t *o_end, int64_t *o_dist,
                            float *o_sd, uint8_t *o_flags, uint32_t m)
{
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m || !flag[p]) return;
  const uint32_t k = ipos[p];
  o_eid[k] = eid[p]; o_end[k] = eend[p]; o_dist[k] = dist[p]; o_sd[k] = sd[p];
  o_flags[k] = flags[p] & 3u;
}
int gtsg_get_scaffold_edges(GtsgEngine *e, uint64_t *count, uint32_t *row, uint32_t *eid,
                            uint32_t *end, int64_t *dist, float *std_dev, uint8_t *flags)
{
  if (!e || !count) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (!e->built) return fail(e, GTSG_EINVAL, "graph not built");
  const uint32_t n = e->n, m = e->m;
  int rc;
  const size_t need = (size_t)m * 5 + (size_t)n * 4 + gts_scan_tmp_elems((uint64_t)m + 16) * 4 + (1u << 20);
  if (e->filter_open) return fail(e, GTSG_EINVAL, "gtsg_filter_begin without gtsg_filter_end");
  if ((rc = pool_reserve(e, need))) return rc;
  PALLOC(flag, uint8_t, (size_t)m + 1); PALLOC(ipos, uint32_t, (size_t)m + 2);
  PALLOC(srow, uint32_t, (size_t)n + 1);
  PALLOC(sctmp, uint32_t, gts_scan_tmp_elems((uint64_t)m + 16));
  uint32_t cnt = 0;
  if (m) {
    LAUNCH("scaf_flags", k_scaf_flags, nblk(m), GTS_BLOCK, e->state, flag, m);
    gts_exscan<uint8_t, uint32_t>(flag, ipos, m, sctmp, ipos + m, e->st);
    if ((rc = read_u32(e, ipos + m, &cnt))) return rc;
  }
  *count = cnt;
  if (!row) return 0;            /* first call: the caller sizes its arrays */
  if (!eid || !end || !dist || !std_dev || !flags) return GTSG_EINVAL;
  if (m) LAUNCH("scaf_rows", k_scaf_rows, nblk((uint64_t)n + 1), GTS_BLOCK, e->row, ipos, srow, n);
  else HIPCHK(hipMemsetAsync(srow, 0, ((size_t)n + 1) * 4, e->st));
  HIPCHK(hipMemcpyAsync(row, srow, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, e->st));
  if (cnt) {
    /* (a second reservation would move the pool: the outputs come from one of their own) */
    uint32_t *o_eid = nullptr, *o_end = nullptr;
    int64_t *o_dist = nullptr;
    float *o_sd = nullptr;
    uint8_t *o_fl = nullptr;
    char *buf = nullptr;
    if (hipMalloc((void **)&buf, (size_t)cnt * 21 + 64) != hipSuccess) return fail(e, GTSG_ENOMEM, "scaffold edge buffer");
    o_dist = (int64_t *)buf; o_eid = (uint32_t *)(o_dist + cnt); o_end = o_eid + cnt;
    o_sd = (float *)(o_end + cnt); o_fl = (uint8_t *)(o_sd + cnt);
    LAUNCH("scaf_fill", k_scaf_fill, nblk(m), GTS_BLOCK, flag, ipos, e->eid, e->eend, e->dist, e->sd,
           e->flags, o_eid, o_end, o_dist, o_sd, o_fl, m);
    hipError_t h1 = hipMemcpyAsync(eid, o_eid, (size_t)cnt * 4, hipMemcpyDeviceToHost, e->st);
    hipError_t h2 = hipMemcpyAsync(end, o_end, (size_t)cnt * 4, hipMemcpyDeviceToHost, e->st);
    hipError_t h3 = hipMemcpyAsync(dist, o_dist, (size_t)cnt * 8, hipMemcpyDeviceToHost, e->st);
    hipError_t h4 = hipMemcpyAsync(std_dev, o_sd, (size_t)cnt * 4, hipMemcpyDeviceToHost, e->st);
    hipError_t h5 = hipMemcpyAsync(flags, o_fl, (size_t)cnt, hipMemcpyDeviceToHost, e->st);
    rc = sync_stream(e);
    hipFree(buf);
    if (h1 != hipSuccess || h2 != hipSuccess || h3 != hipSuccess || h4 != hipSuccess || h5 != hipSuccess)
      return fail(e, GTSG_EHIP, "copying the scaffold edges");
    return rc;
  }
  return sync_stream(e);
}

/* ---- the scaffold records on the device (ref algorithms.c:901-997) -----------
   The reference visits the vertices in index order; an unvisited, unmarked vertex
   with at most one SCAFFOLD edge opens a record, and the record follows the
   SCAFFOLD edges -- marking the vertices it arrives at VISITED, stopping at one
   that is already -- while exactly one of them, not the way back, leaves the
   vertex in the direction the arriving edge asks for.

   makescaffold marks the edges of a walk and their twins.  Nearly all of these
   walks are simple paths: every edge u -> w has exactly one edge w -> u, w has at
   most two SCAFFOLD edges and is not marked, and the other edge of w (if there is
   one) points in the asked direction.  Call such an edge regular and a path of
   regular edges between untainted vertices clean (a vertex is tainted by any
   irregular edge at it, in or out).  Nothing leads into a clean path: an edge from
   outside would find no edge back at the vertex it ends in and taint it.  The
   record of a clean path therefore depends on nothing else: it starts at the end
   of lower index (the first the loop meets; the other end is visited by then) and
   runs to the other end, and no vertex is met twice -- list ranking.  For a
   directed edge k = (u -> w), next[k] is the other edge of w; pointer jumping
   gives last[k], the final edge of the chain from k, and hops[k], the edges after
   k, and whether an irregular edge lies ahead.  With t = twin(k): the chain from t
   ends at the path end the walk through k started from (start = end[last[t]]),
   and k is edge number hops[t] of that walk.  Clean cycles have no end and no
   record, as in the reference (none of their vertices has fewer than two edges).

   A walk of the reference's search that passes a contig twice (a contig pushed
   again with a shorter distance, create_walk :700-730) leaves a vertex with three
   or four SCAFFOLD edges: about one contig in a thousand on the synthetic
   graphs.  The paths such vertices lie on, their edges and the vertices that
   could open a record there are handed to the caller as the "open" part -- a
   compact list in adjacency order -- to be walked in the reference's order of
   visits; being closed under SCAFFOLD edges as well, the two parts do not see
   each other, and the records of both merge by root index. */
#define GTS_REC_NONE 0xFFFFFFFFu
#define GTS_REC_OPEN 0x80000000u
__global__ void k_scaf_fill_start(const uint8_t *flag, const uint32_t *ipos, const uint32_t *estart,
                                  uint32_t *o_start, uint32_t m)
{
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < m && flag[p]) o_start[ipos[p]] = estart[p];
}
__device__ __forceinline__ bool rec_marked(uint8_t s)
{
  return s == GIS_POLYMORPHIC || s == GIS_REPEAT || s == GIS_CYCLIC;
}
/* per directed SCAFFOLD edge: its twin and the edge the walk takes after it, or what is
   irregular about it (gtsg_get_stat "records_irregular": 1 self loop, 2 more than two
   SCAFFOLD edges at the end vertex, 4 a marked vertex, 8 two edges back, 16 none, 32 the
   edge on points the other way) */
__global__ void k_rec_links(const uint32_t *srow, const uint32_t *su, const uint32_t *en, const uint8_t *fl,
                            const uint8_t *vstate, uint32_t *tw, uint32_t *other_of, uint8_t *taint,
                            uint32_t cnt, uint32_t *why)
{
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const uint32_t u = su[k], w = en[k];
  const uint32_t b = srow[w], ns = srow[w + 1] - b;
  const bool sense = fl[k] & 1, same = fl[k] & 2, dir = same ? sense : !sense;
  uint32_t back = GTS_REC_NONE, other = GTS_REC_NONE;
  uint32_t bad = (u == w ? 1u : 0u) | (ns > 2 ? 2u : 0u) | (rec_marked(vstate[w]) || rec_marked(vstate[u]) ? 4u : 0u);
  if (!bad)
    for (uint32_t j = b; j < b + ns; ++j) {
      if (en[j] == u) { if (back != GTS_REC_NONE) bad |= 8u; back = j; }
      else other = j;
    }
  if (!bad && back == GTS_REC_NONE) bad |= 16u;
  if (!bad && other != GTS_REC_NONE && ((fl[other] & 1) != 0) != dir) bad |= 32u;
  if (bad) {
    atomicOr(why, bad);
    taint[u] = 1; taint[w] = 1;
    tw[k] = GTS_REC_NONE; other_of[k] = GTS_REC_NONE;
    return;
  }
  tw[k] = back; other_of[k] = other;
}
/* (pointer, hops) to start the jumping from; an edge at a tainted vertex ends its chain */
__global__ void k_rec_start(const uint32_t *su, const uint32_t *en, const uint32_t *tw, const uint32_t *other_of,
                            const uint8_t *taint, uint2 *link, uint32_t cnt)
{
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const uint32_t o = other_of[k];
  if (tw[k] == GTS_REC_NONE || taint[su[k]] || taint[en[k]]) link[k] = make_uint2((uint32_t)k, GTS_REC_OPEN);
  else link[k] = o == GTS_REC_NONE ? make_uint2((uint32_t)k, 0u) : make_uint2(o, 1u);
}
/* one round of pointer jumping: (pointer, hops) -> (pointer of pointer, hops + its hops) */
__global__ void k_rec_jump(const uint2 *in, uint2 *out, uint32_t cnt, uint32_t *changed)
{
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const uint2 a = in[k];
  const uint2 b = in[a.x];
  out[k] = make_uint2(b.x, (((a.y & ~GTS_REC_OPEN) + (b.y & ~GTS_REC_OPEN)) & ~GTS_REC_OPEN) | ((a.y | b.y) & GTS_REC_OPEN));
  if (b.x != a.x) *changed = 1;
}
/* per vertex: does it open a record of a clean path (and with how many edges), or could it
   open one in the open part */
__global__ void k_rec_roots(const uint32_t *srow, const uint32_t *en, const uint2 *link, const uint8_t *vstate,
                            const uint8_t *taint, uint8_t *isroot, uint32_t *nedge, uint8_t *isopen, uint32_t n)
{
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const uint32_t b = srow[v], ns = srow[v + 1] - b;
  uint8_t r = 0, o = 0;
  uint32_t c = 0;
  if (!rec_marked(vstate[v])) {
    if (ns == 0) { if (taint[v]) o = 1; else r = 1; }
    else if (ns == 1) {
      const uint2 l = link[b];
      if ((l.y & GTS_REC_OPEN) || taint[v]) o = 1;
      else if ((uint32_t)v < en[l.x]) { r = 1; c = l.y + 1; }
    }
  }
  isroot[v] = r; nedge[v] = c; isopen[v] = o;
}
/* per directed SCAFFOLD edge: is it part of the open part (something irregular ahead of it
   or behind it on its path) */
__global__ void k_rec_open_edges(const uint32_t *tw, const uint2 *link, uint8_t *eopen, uint32_t cnt)
{
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const uint32_t t = tw[k];
  const uint2 a = link[k];
  uint8_t o = 0;
  if (t == GTS_REC_NONE || (a.y & GTS_REC_OPEN)) o = 1;
  else if (link[t].y & GTS_REC_OPEN) o = 1;
  /* (the chains of a clean cycle never end: neither open nor anybody's) */
  eopen[k] = o;
}
__global__ void k_rec_heads(const uint8_t *isroot, const uint32_t *ridx, const uint32_t *eoff,
                            const uint8_t *isopen, const uint32_t *oidx, const int64_t *seq_len,
                            uint32_t *root, uint32_t *off, unsigned long long *seqlen, uint32_t *open_root,
                            uint32_t n)
{
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  if (isroot[v]) {
    const uint32_t r = ridx[v];
    root[r] = (uint32_t)v; off[r] = eoff[v]; seqlen[r] = (unsigned long long)seq_len[v];
  }
  if (isopen[v]) open_root[oidx[v]] = (uint32_t)v;
}
/* per directed SCAFFOLD edge: the record it belongs to (if it points away from the root of
   its clean path) and its place in it; or its place in the list of the open part */
__global__ void k_rec_place(const uint32_t *tw, const uint2 *link, const uint32_t *su, const uint32_t *en,
                            const uint8_t *isroot, const uint32_t *ridx, const uint32_t *eoff,
                            const uint8_t *eopen, const uint32_t *eoidx, const int64_t *seq_len,
                            const uint32_t *i_eid, const int64_t *i_dist, const float *i_sd, const uint8_t *i_fl,
                            uint32_t *o_eid, uint32_t *o_end, int64_t *o_dist, float *o_sd, uint8_t *o_fl,
                            unsigned long long *seqlen,
                            uint32_t *p_start, uint32_t *p_eid, uint32_t *p_end, int64_t *p_dist, float *p_sd,
                            uint8_t *p_fl, uint32_t cnt)
{
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const uint32_t w = en[k];
  const int64_t d = i_dist[k];
  if (eopen[k]) {
    const uint32_t pos = eoidx[k];
    p_start[pos] = su[k]; p_eid[pos] = i_eid[k]; p_end[pos] = w; p_dist[pos] = d; p_sd[pos] = i_sd[k];
    p_fl[pos] = i_fl[k];
    return;
  }
  const uint2 l = link[tw[k]];
  if (l.y & GTS_REC_OPEN) return;
  const uint32_t s = en[l.x];
  if (!isroot[s]) return;
  const uint32_t pos = eoff[s] + l.y;
  o_eid[pos] = i_eid[k]; o_end[pos] = w; o_dist[pos] = d; o_sd[pos] = i_sd[k]; o_fl[pos] = i_fl[k];
  atomicAdd(seqlen + ridx[s], (unsigned long long)seq_len[w] + (unsigned long long)d);
}

struct RecLayout {
  size_t nr, ne, pr, pe;
  char *seqlen, *dist, *p_dist, *root, *off, *eid, *end, *sd, *p_root, *p_start, *p_eid, *p_end, *p_sd, *fl, *p_fl;
  size_t bytes;
};
static RecLayout rec_layout(char *base, size_t nr, size_t ne, size_t pr, size_t pe)
{
  RecLayout L;
  char *kp = base;
  auto carve = [&](size_t bytes) { char *r = kp; kp += (bytes + 15) & ~(size_t)15; return r; };
  L.nr = nr; L.ne = ne; L.pr = pr; L.pe = pe;
  L.seqlen = carve(nr * 8); L.dist = carve(ne * 8); L.p_dist = carve(pe * 8);
  L.root = carve(nr * 4); L.off = carve(nr * 4); L.eid = carve(ne * 4); L.end = carve(ne * 4); L.sd = carve(ne * 4);
  L.p_root = carve(pr * 4); L.p_start = carve(pe * 4); L.p_eid = carve(pe * 4); L.p_end = carve(pe * 4);
  L.p_sd = carve(pe * 4);
  L.fl = carve(ne); L.p_fl = carve(pe);
  L.bytes = (size_t)(kp - base) + 64;
  return L;
}

int gtsg_scaffold_records(GtsgEngine *e, GtsgRecordCounts *counts)
{
  if (!e || !counts) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (!e->built) return fail(e, GTSG_EINVAL, "graph not built");
  if (e->filter_open) return fail(e, GTSG_EINVAL, "gtsg_filter_begin without gtsg_filter_end");
  const uint32_t n = e->n, m = e->m;
  int rc;
  memset(counts, 0, sizeof *counts);
  e->rec_counts[0] = e->rec_counts[1] = e->rec_counts[2] = e->rec_counts[3] = 0;
  e->rec_ready = false;
  e->stats["records_irregular"] = 0;
  uint32_t cnt = 0;
  {
    const size_t need = (size_t)m * 5 + (size_t)n * 4 + gts_scan_tmp_elems((uint64_t)m + 16) * 4 + (1u << 20);
    if ((rc = pool_reserve(e, need))) return rc;
  }
  uint8_t *flag = nullptr;
  uint32_t *ipos = nullptr;
  if (m) {
    flag = pool_alloc<uint8_t>(e, (size_t)m + 1); ipos = pool_alloc<uint32_t>(e, (size_t)m + 2);
    uint32_t *sctmp = pool_alloc<uint32_t>(e, gts_scan_tmp_elems((uint64_t)m + 16));
    if (!flag || !ipos || !sctmp) return GTSG_ENOMEM;
    LAUNCH("scaf_flags", k_scaf_flags, nblk(m), GTS_BLOCK, e->state, flag, m);
    gts_exscan<uint8_t, uint32_t>(flag, ipos, m, sctmp, ipos + m, e->st);
    if ((rc = read_u32(e, ipos + m, &cnt))) return rc;
  }
  /* the compact CSR and everything the ranking needs, in an allocation of its own
     (the pool holds flag / ipos until the fill is done) */
  const size_t c1 = (size_t)cnt + 16, n1 = (size_t)n + 16;
  const size_t scan_elems = gts_scan_tmp_elems((uint64_t)(cnt > n ? cnt : n) + 16);
  const size_t work = c1 * (8 + 8 + 8 + 4 * 7 + 2) + n1 * (4 * 5 + 3) + scan_elems * 4 + 64 * 32;
  char *wbuf = nullptr;
  if (hipMalloc((void **)&wbuf, work) != hipSuccess) {
    (void)hipGetLastError();
    return fail(e, GTSG_ENOMEM, "scaffold record workspace of %zu bytes", work);
  }
  char *wp = wbuf;
  auto carve = [&](size_t bytes) { char *r = wp; wp += (bytes + 15) & ~(size_t)15; return r; };
  int64_t *c_dist = (int64_t *)carve(c1 * 8);
  uint2 *linkA = (uint2 *)carve(c1 * 8), *linkB = (uint2 *)carve(c1 * 8);
  uint32_t *su = (uint32_t *)carve(c1 * 4), *en = (uint32_t *)carve(c1 * 4);
  float *c_sd = (float *)carve(c1 * 4);
  uint32_t *c_eid = (uint32_t *)carve(c1 * 4), *tw = (uint32_t *)carve(c1 * 4), *other_of = (uint32_t *)carve(c1 * 4);
  uint32_t *eoidx = (uint32_t *)carve(c1 * 4);
  uint32_t *srow = (uint32_t *)carve(n1 * 4), *nedge = (uint32_t *)carve(n1 * 4), *ridx = (uint32_t *)carve(n1 * 4);
  uint32_t *eoff = (uint32_t *)carve(n1 * 4), *oidx = (uint32_t *)carve(n1 * 4);
  uint32_t *sctmp2 = (uint32_t *)carve(scan_elems * 4);
  uint8_t *c_fl = (uint8_t *)carve(c1), *eopen = (uint8_t *)carve(c1);
  uint8_t *isroot = (uint8_t *)carve(n1), *isopen = (uint8_t *)carve(n1), *taint = (uint8_t *)carve(n1);
  if ((size_t)(wp - wbuf) > work) { hipFree(wbuf); return fail(e, GTSG_EINTERNAL, "scaffold record workspace layout"); }
  /* [0] why irregular, [1] changed, [2] records, [3] their edges, [4] open roots, [5] open edges */
  uint32_t *d_flag = (uint32_t *)(e->d_scalars + 8);
  auto bail = [&](int r) { hipStreamSynchronize(e->st); hipFree(wbuf); return r; };
#define RECCHK(x) do { if ((x) != hipSuccess) return bail(fail(e, GTSG_EHIP, "scaffold records: %s", #x)); } while (0)
  RECCHK(hipMemsetAsync(d_flag, 0, 32, e->st));
  RECCHK(hipMemsetAsync(taint, 0, n1, e->st));
  if (m) LAUNCH("scaf_rows", k_scaf_rows, nblk((uint64_t)n + 1), GTS_BLOCK, e->row, ipos, srow, n);
  else RECCHK(hipMemsetAsync(srow, 0, n1 * 4, e->st));
  uint2 *link = linkA;
  if (cnt) {
    LAUNCH("scaf_fill", k_scaf_fill, nblk(m), GTS_BLOCK, flag, ipos, e->eid, e->eend, e->dist, e->sd,
           e->flags, c_eid, en, c_dist, c_sd, c_fl, m);
    LAUNCH("scaf_fill", k_scaf_fill_start, nblk(m), GTS_BLOCK, flag, ipos, e->estart, su, m);
    LAUNCH("rec_links", k_rec_links, nblk(cnt), GTS_BLOCK, srow, su, en, c_fl, e->vstate, tw, other_of, taint, cnt,
           d_flag);
    LAUNCH("rec_links", k_rec_start, nblk(cnt), GTS_BLOCK, su, en, tw, other_of, taint, linkA, cnt);
    uint32_t h[2] = {0, 0};
    /* a path of L edges is ranked after ceil(log2 L) rounds; the chains of a cycle
       never settle and are nobody's: 32 rounds at most */
    for (int round = 0; round < 32; ++round) {
      RECCHK(hipMemsetAsync(d_flag + 1, 0, 4, e->st));
      uint2 *out = link == linkA ? linkB : linkA;
      LAUNCH("rec_jump", k_rec_jump, nblk(cnt), GTS_BLOCK, link, out, cnt, d_flag + 1);
      link = out;
      RECCHK(hipMemcpyAsync(h, d_flag, 8, hipMemcpyDeviceToHost, e->st));
      if ((rc = sync_stream(e))) return bail(rc);
      e->stats["records_jump_rounds"] = round + 1;
      if (!h[1]) break;
    }
    e->stats["records_irregular"] = h[0];
    LAUNCH("rec_open", k_rec_open_edges, nblk(cnt), GTS_BLOCK, tw, link, eopen, cnt);
    gts_exscan<uint8_t, uint32_t>(eopen, eoidx, cnt, sctmp2, d_flag + 5, e->st);
  }
  if (n) {
    LAUNCH("rec_roots", k_rec_roots, nblk(n), GTS_BLOCK, srow, en, link, e->vstate, taint, isroot, nedge, isopen, n);
    gts_exscan<uint8_t, uint32_t>(isroot, ridx, n, sctmp2, d_flag + 2, e->st);
    gts_exscan<uint32_t, uint32_t>(nedge, eoff, n, sctmp2, d_flag + 3, e->st);
    gts_exscan<uint8_t, uint32_t>(isopen, oidx, n, sctmp2, d_flag + 4, e->st);
  }
  uint32_t tot[4] = {0, 0, 0, 0};
  RECCHK(hipMemcpyAsync(tot, d_flag + 2, 16, hipMemcpyDeviceToHost, e->st));
  if ((rc = sync_stream(e))) return bail(rc);
  const size_t nr = tot[0], ne = tot[1], pr = tot[2], pe = tot[3];
  {
    RecLayout L0 = rec_layout(nullptr, nr, ne, pr, pe);
    if ((rc = dev_alloc(e, &e->rec_buf, L0.bytes))) return bail(rc);
  }
  const RecLayout L = rec_layout(e->rec_buf, nr, ne, pr, pe);
  if (n) LAUNCH("rec_heads", k_rec_heads, nblk(n), GTS_BLOCK, isroot, ridx, eoff, isopen, oidx, e->seq_len,
                (uint32_t *)L.root, (uint32_t *)L.off, (unsigned long long *)L.seqlen, (uint32_t *)L.p_root, n);
  if (cnt) LAUNCH("rec_place", k_rec_place, nblk(cnt), GTS_BLOCK, tw, link, su, en, isroot, ridx, eoff, eopen, eoidx,
                  e->seq_len, c_eid, c_dist, c_sd, c_fl, (uint32_t *)L.eid, (uint32_t *)L.end, (int64_t *)L.dist,
                  (float *)L.sd, (uint8_t *)L.fl, (unsigned long long *)L.seqlen, (uint32_t *)L.p_start,
                  (uint32_t *)L.p_eid, (uint32_t *)L.p_end, (int64_t *)L.p_dist, (float *)L.p_sd, (uint8_t *)L.p_fl,
                  cnt);
  if ((rc = sync_stream(e))) return bail(rc);
#undef RECCHK
  hipFree(wbuf);
  e->rec_counts[0] = nr; e->rec_counts[1] = ne; e->rec_counts[2] = pr; e->rec_counts[3] = pe;
  e->rec_ready = true;
  counts->n_records = nr; counts->n_edges = ne; counts->n_open_roots = pr; counts->n_open_edges = pe;
  e->stats["records_ranked"] = (int64_t)nr;
  e->stats["records_open_roots"] = (int64_t)pr;
  e->stats["records_open_edges"] = (int64_t)pe;
  return 0;
}

int gtsg_scaffold_records_fetch(GtsgEngine *e, const GtsgRecordArrays *a)
{
  if (!e || !a) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  if (!e->rec_ready || !e->rec_buf) return fail(e, GTSG_EINVAL, "gtsg_scaffold_records first");
  const size_t nr = e->rec_counts[0], ne = e->rec_counts[1], pr = e->rec_counts[2], pe = e->rec_counts[3];
  if ((nr && (!a->root || !a->off || !a->seqlen)) ||
      (ne && (!a->eid || !a->end || !a->dist || !a->std_dev || !a->flags)) || (pr && !a->open_root) ||
      (pe && (!a->open_start || !a->open_eid || !a->open_end || !a->open_dist || !a->open_std_dev || !a->open_flags)))
    return GTSG_EINVAL;
  const RecLayout L = rec_layout(e->rec_buf, nr, ne, pr, pe);
#define RECGET(dst, src, bytes) do { if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->st)); } while (0)
  RECGET(a->seqlen, L.seqlen, nr * 8); RECGET(a->root, L.root, nr * 4); RECGET(a->off, L.off, nr * 4);
  RECGET(a->dist, L.dist, ne * 8); RECGET(a->eid, L.eid, ne * 4); RECGET(a->end, L.end, ne * 4);
  RECGET(a->std_dev, L.sd, ne * 4); RECGET(a->flags, L.fl, ne);
  RECGET(a->open_root, L.p_root, pr * 4);
  RECGET(a->open_dist, L.p_dist, pe * 8); RECGET(a->open_start, L.p_start, pe * 4); RECGET(a->open_eid, L.p_eid, pe * 4);
  RECGET(a->open_end, L.p_end, pe * 4); RECGET(a->open_std_dev, L.p_sd, pe * 4); RECGET(a->open_flags, L.p_fl, pe);
#undef RECGET
  return sync_stream(e);
}

int gtsg_state_digest(GtsgEngine *e, uint64_t *vd, uint64_t *ed)
{
  if (!e || !vd || !ed) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  unsigned long long *acc = (unsigned long long *)(e->d_scalars + 8);
  HIPCHK(hipMemsetAsync(acc, 0, 16, e->st));
  if (e->n)
    LAUNCH("digest", k_digest, nblk(e->n), GTS_BLOCK, e->vstate, (const uint32_t *)nullptr,
           (uint64_t)e->n, acc);
  if (e->m)
    LAUNCH("digest", k_digest, nblk(e->m), GTS_BLOCK, e->state, e->eid, (uint64_t)e->m, acc + 1);
  uint64_t h[2];
  HIPCHK(hipMemcpyAsync(h, acc, 16, hipMemcpyDeviceToHost, e->st));
  int rc = sync_stream(e);
  *vd = h[0]; *ed = h[1];
  return rc;
}

int gtsg_selftest_ambiguous(GtsgEngine *e, uint64_t n, const int64_t *d1,
                            const float *s1, const int64_t *d2, const float *s2,
                            float pcutoff, uint8_t *out)
{
  if (!e || !d1 || !s1 || !d2 || !s2 || !out) return GTSG_EINVAL;
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if ((rc = pool_reserve(e, n * 32 + (1u << 20)))) return rc;
  PALLOC(a, int64_t, n); PALLOC(b, float, n); PALLOC(c, int64_t, n);
  PALLOC(d, float, n); PALLOC(o, uint8_t, n);
  if ((rc = upload(e, a, d1, n, 0)) || (rc = upload(e, b, s1, n, 0)) || (rc = upload(e, c, d2, n, 0)) ||
      (rc = upload(e, d, s2, n, 0)))
    return rc;
  LAUNCH("amb_test", k_amb_test, nblk(n), GTS_BLOCK, a, b, c, d, o, n, gts_amb_thresholds(pcutoff));
  HIPCHK(hipMemcpyAsync(out, o, n, hipMemcpyDeviceToHost, e->st));
  return sync_stream(e);
}

int gtsg_get_kernel_times(GtsgEngine *e, GtsgKernelTime *out, int cap)
{
  if (!e) return GTSG_EINVAL;
  collect_times(e);
  int i = 0;
  for (auto &kv : e->ktimes) {
    if (out && i < cap) {
      memset(&out[i], 0, sizeof out[i]);
      strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
      out[i].calls = kv.second.first;
      out[i].ms = kv.second.second;
    }
    ++i;
  }
  return i;
}
void gtsg_reset_kernel_times(GtsgEngine *e) { if (e) { collect_times(e); e->ktimes.clear(); } }

int64_t gtsg_get_stat(const GtsgEngine *e, const char *name)
{
  if (!e || !name) return -1;
  /* HBM held by the engine: the graph and contig arrays, and the workspace
     (grown to the largest stage so far) */
  if (!strcmp(name, "bytes_workspace")) return (int64_t)e->pool_cap;
  if (!strcmp(name, "bytes_graph")) {
    size_t b = 0;
    for (auto &kv : e->alloc_bytes) b += kv.second;
    return (int64_t)b;
  }
  auto it = e->stats.find(name);
  return it == e->stats.end() ? -1 : it->second;
}

} /* extern "C" */
