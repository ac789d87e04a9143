/*
  gts_deparse.hip -- DistEst (.de) text -> records on the GPU (gfx950).

  Replaces the two passes of gt_scaffolder_parser.c over the distance file for
  files in the regular form ABySS' DistanceEst writes:

      <root> <ctg>{+,-},<dist>,<pairs>,<std> ... ; <ctg>{+,-},... \n

    pass 0  gt_scaffolder_parser_count_distances  (ref parser.c:150-291):
            integrity check, first error in file order, number of records
            between known contigs;
    pass 1  gt_scaffolder_parser_read_distances   (ref parser.c:295-394):
            the records in file order, sense flipped at ';'.

  The reference tokenises with strtok(" "), scans a record with
  sscanf("%[^>,],%ld,%ld,%f") and finds contigs with bsearch over the sorted
  headers.  Here: the text is cut into strides of 16 KB, a workgroup takes the
  lines that START in its stride (a line is at most 1023 bytes, so a stride
  plus one line fits LDS), stages them with coalesced loads, finds the line
  starts with a ballot-free scan in LDS and parses one line per thread from
  LDS.  Contig names are looked up in an open-addressing hash table built from
  the sorted header list (entry = upper half of a 64-bit FNV-1a + id, the name
  itself is compared on a hit).  Records go to candidate slots numbered by a
  scan over the strides (every record token of the file has one), a compaction
  follows only if a token did not become a record.

  Exactness: the parser accepts exactly the inputs for which it provably
  computes what sscanf / strtol / strtof compute (plain decimal integers of at
  most 18 digits; decimal fractions of at most 19 significant digits, computed
  in double and refused if the double is within its error of a float midpoint,
  so rounding it to float cannot differ from strtof's single rounding of the
  decimal).  Anything else -- control characters, exponents, hex floats,
  inf/nan, '>' in a header, a blank before the line end, a last line without
  newline, lines above 1023 bytes -- sets `irregular`, and the caller parses
  the file with the host code instead (gt_scaffolder_host.c, which restates
  the reference line by line).  irregular never changes a result, only who
  computes it.
*/
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gt_scaffold_hip.h"
#include "gts_prims.hpp"
#include "gts_deparse_tok.hpp"

#define DP_STRIDE 16384u
#define DP_MAXLINE 1023u            /* fgets(line, 1024): longest line with its newline */
#define DP_THREADS 256u
#define DP_MAXLINES 1024u           /* lines starting in one stride (more: irregular) */
#define DP_MAXTOK 2048u             /* tokens in one stride (more: irregular) */
#define DP_BUF (DP_STRIDE + 1024u + 32u)

#define DP_ERR_RECORD 1u            /* "Invalid record in dist file"                 parser.c:205,236 */
#define DP_ERR_PAIRS 2u             /* "Invalid value for number of pairs ..."       parser.c:218 */
#define DP_ERR_SIGN 3u              /* "Invalid composition sign ..."                parser.c:226 */

struct GtsgDeParser {
  int device;
  hipStream_t st;
  bool own_stream;
  /* contig names: blob + n+1 offsets, hash table */
  char *names;
  uint32_t *name_off;
  uint64_t *table;
  uint64_t table_mask;
  uint64_t n_names;
  /* text staged on the device when the caller hands a host buffer */
  char *text;
  uint64_t text_cap;
  /* candidates / records of the last parse */
  uint32_t *root, *ctg;
  int64_t *dist, *np;
  float *sd;
  uint8_t *flags, *valid;
  uint32_t *cand_cnt;               /* per stride, then its exclusive scan */
  uint32_t *scan_tmp;               /* scratch of the prefix sums, grown like the rest (scan_cap u32) */
  uint64_t scan_cap;
  uint32_t *cpos;                   /* positions of the compaction (cpos_cap u32) */
  uint64_t cpos_cap;
  uint64_t cap_cand, cap_blocks;
  /* compacted copies (only when a candidate is not a record) */
  uint32_t *root2, *ctg2;
  int64_t *dist2, *np2;
  float *sd2;
  uint8_t *flags2;
  uint64_t cap2;
  bool compacted;
  uint64_t n_records;
  /* records of several parses back to back (a file in pieces) */
  bool accumulate;
  uint32_t *a_root, *a_ctg;
  int64_t *a_dist, *a_np;
  float *a_sd;
  uint8_t *a_flags;
  uint64_t a_n, a_cap;
  unsigned long long *d_res;        /* [0] first error, [1] irregular, [2] valid, [3] candidates */
  char err[256];
};

static int dp_fail(GtsgDeParser *p, int code, const char *msg)
{
  snprintf(p->err, sizeof p->err, "%s", msg);
  return code;
}
#define DPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { \
  (void)hipGetLastError();   /* (not left with the thread for the next, unrelated call) */ \
  snprintf(p->err, sizeof p->err, "%s: %s", #x, hipGetErrorString(_e)); \
  return _e == hipErrorOutOfMemory ? GTSG_ENOMEM : GTSG_EHIP; } } while (0)

/* ---- names ---------------------------------------------------------------- */
__device__ __forceinline__ uint64_t dp_hash_step(uint64_t h, uint8_t c)
{
  return (h ^ c) * 1099511628211ull;    /* FNV-1a */
}
#define DP_HASH_INIT 14695981039346656037ull

__global__ void k_dp_insert(const char *names, const uint32_t *off, uint64_t n, uint64_t *table,
                            uint64_t mask)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t h = DP_HASH_INIT;
  for (uint32_t k = off[i]; k < off[i + 1]; ++k) h = dp_hash_step(h, (uint8_t)names[k]);
  const unsigned long long entry = (h & 0xFFFFFFFF00000000ull) | (unsigned long long)(i + 1);
  uint64_t slot = h & mask;
  while (atomicCAS((unsigned long long *)&table[slot], 0ull, entry) != 0ull) slot = (slot + 1) & mask;
}

/* id of the name buf[a, b) (LDS), or GTS_NONE */
#define DP_NONE 0xFFFFFFFFu
__device__ __forceinline__ uint32_t dp_lookup(DpText &buf, uint32_t a, uint32_t b, const char *names,
                                              const uint32_t *off, const uint64_t *table, uint64_t mask)
{
  if (a >= b) return DP_NONE;
  uint64_t h = DP_HASH_INIT;
  for (uint32_t k = a; k < b; ++k) h = dp_hash_step(h, buf[k]);
  for (uint64_t slot = h & mask;; slot = (slot + 1) & mask) {
    const uint64_t e = table[slot];
    if (e == 0) return DP_NONE;
    if ((e ^ h) >> 32) continue;
    const uint32_t id = (uint32_t)e - 1u;
    const uint32_t o = off[id], len = off[id + 1] - o;
    if (len != b - a) continue;
    /* the name itself, eight bytes per step from aligned words (the blob has
       16 bytes of slack behind it); no early exit, so the loads of all steps
       are in flight together */
    uint64_t diff = 0;
    for (uint32_t k = 0; k < len; k += 8) {
      const uint32_t at = o + k, al = at & ~7u, sh = (at & 7u) * 8u;
      const uint64_t lo = *(const uint64_t *)(names + al), hi = *(const uint64_t *)(names + al + 8);
      const uint64_t wn = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
      uint64_t wt = 0;
      const uint32_t m = len - k < 8 ? len - k : 8;
      for (uint32_t j = 0; j < m; ++j) wt |= (uint64_t)buf[a + k + j] << (8u * j);
      const uint64_t bytes_m = m < 8 ? (1ull << (8u * m)) - 1ull : ~0ull;
      diff |= (wn ^ wt) & bytes_m;
    }
    if (diff == 0) return id;
  }
}

/* ---- a stride of text in LDS -------------------------------------------------
   Workgroup b owns the lines that start in [b * DP_STRIDE, (b + 1) * DP_STRIDE):
   from the first line start at or after the one bound to the first at or after
   the other (one past the first newline from position x - 1 on; a line has at
   most DP_MAXLINE bytes, or the file is the host's).  Stages them in sbuf with
   16-byte loads (text position 0 of the range at offset start % 16).  false:
   nothing starts here, or the file is irregular (flagged in res[1]). */
__device__ __forceinline__ bool dp_stage_stride(const char *text, uint64_t len, uint8_t *sbuf, uint32_t *s_first,
                                                unsigned long long *res, uint64_t &start, uint32_t &n)
{
  const uint32_t tid = threadIdx.x;
  const uint64_t lo = (uint64_t)blockIdx.x * DP_STRIDE;
  const uint64_t hi = lo + DP_STRIDE < len ? lo + DP_STRIDE : len;
  if (tid < 2) s_first[tid] = 0xFFFFFFFFu;
  __syncthreads();
  for (int w = 0; w < 2; ++w) {
    const uint64_t x = w ? hi : lo;
    if (x == 0 || x >= len) continue;
    for (uint32_t k = tid; k < DP_MAXLINE + 1; k += DP_THREADS) {
      const uint64_t q = x - 1 + k;
      if (q < len && text[q] == '\n') atomicMin(&s_first[w], k);
    }
  }
  __syncthreads();
  uint64_t end = hi;
  start = lo;
  n = 0;
  bool bad = false;
  if (lo != 0) { if (s_first[0] == 0xFFFFFFFFu) bad = true; else start = lo + s_first[0]; }
  if (hi < len) { if (s_first[1] == 0xFFFFFFFFu) bad = true; else end = hi + s_first[1]; }
  if (bad) {
    /* a line above the reference's buffer (or no newline up to the end of the
       file): the host reports it */
    if (tid == 0) atomicOr(res + 1, 1ull);
    return false;
  }
  if (start >= end) return false;
  n = (uint32_t)(end - start);     /* <= DP_STRIDE + DP_MAXLINE */
  const uint64_t g0 = start & ~15ull;
  const uint32_t shift = (uint32_t)(start - g0);
  const uint32_t chunks = (shift + n + 15u) / 16u;
  for (uint32_t c = tid; c < chunks; c += DP_THREADS) {
    const uint64_t g = g0 + (uint64_t)c * 16u;
    if (g >= start && g + 16u <= end && ((uintptr_t)(text + g) & 15u) == 0) {
      *(uint4 *)(sbuf + (size_t)c * 16u) = *(const uint4 *)(text + g);
    } else {
      for (uint32_t k = 0; k < 16u; ++k) {
        const uint64_t q = g + k;
        sbuf[(size_t)c * 16u + k] = (q >= start && q < end) ? (uint8_t)text[q] : (uint8_t)'\n';
      }
    }
  }
  __syncthreads();
  return true;
}

/* ---- the stride kernel ------------------------------------------------------
   EMIT = false: candidate slots per stride (cand_cnt[b]), irregular flags.
   EMIT = true : parses, looks the names up, writes the candidates of stride b
                 from cand_base[b] on, first error, number of valid records.

   Every step has ALL lanes on the same code:
     1. each thread scans a slice of the staged bytes for line starts and token
        starts (a token: a run of bytes other than ' ' and '\n'), two block
        scans number them, a second walk writes line_start[] / tok_start[] /
        tok_line[];
     2. one thread per token: separator or not, first of its line or not; an
        inclusive scan of "is a candidate" (neither) gives every record token
        its slot, and the separators between a line's first token and a token
        (its sense) are (tokens in between) - (candidates in between);
     3. one thread per line: the root's id; a line with one token or none is the
        reference's "Invalid record";
     4. one thread per record token: conversion, look-up, the record.
   (The first version parsed a line per thread: lanes of a wavefront were at
   different places of the token loop most of the time, 72.8 k instructions per
   wavefront of 15 busy lanes.) */
template <bool EMIT>
__global__ void __launch_bounds__(DP_THREADS)
k_dp_stride(const char *text, uint64_t len, uint32_t *cand_cnt, const uint32_t *cand_base,
            const char *names, const uint32_t *name_off, const uint64_t *table, uint64_t mask,
            uint32_t *o_root, uint32_t *o_ctg, int64_t *o_dist, int64_t *o_np, float *o_sd,
            uint8_t *o_flags, uint8_t *o_valid, unsigned long long *res)
{
  __shared__ __attribute__((aligned(16))) uint8_t sbuf[DP_BUF];
  __shared__ uint16_t line_start[DP_MAXLINES + 1];
  __shared__ uint16_t first_tok[DP_MAXLINES];
  __shared__ uint32_t root_of[EMIT ? DP_MAXLINES : 1];
  __shared__ uint16_t tok_start[DP_MAXTOK], tok_line[DP_MAXTOK], tok_cand[DP_MAXTOK];
  __shared__ uint32_t s_scan[DP_THREADS], s_scan2[DP_THREADS];
  __shared__ uint32_t s_first[2], s_irregular, s_carry;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) { s_irregular = 0; s_carry = 0; }
  uint64_t start;
  uint32_t n;
  if (!dp_stage_stride(text, len, sbuf, s_first, res, start, n)) {
    if (!EMIT && tid == 0) cand_cnt[blockIdx.x] = 0;
    return;
  }
  DpText txt(sbuf, (uint32_t)(start - (start & ~15ull)));
  /* ---- 1. lines and tokens: count per slice, scan, write ---- */
  const uint32_t q = (n + DP_THREADS - 1) / DP_THREADS;
  const uint32_t b0 = tid * q < n ? tid * q : n, b1 = b0 + q < n ? b0 + q : n;
  uint32_t nl = 0, nt = 0;
  bool irregular = false;
  {
    uint8_t prev = b0 ? txt[b0 - 1] : (uint8_t)'\n';
    for (uint32_t i = b0; i < b1; ++i) {
      const uint8_t c = txt[i];
      if (prev == '\n') ++nl;
      if (c != ' ' && c != '\n' && (prev == ' ' || prev == '\n')) ++nt;
      if ((c < 0x20 && c != '\n') || c == 0x7F) irregular = true;   /* tabs, CR, NUL, ... */
      if (c == '\n' && prev == ' ') irregular = true;               /* pass 0 and pass 1 see different tokens */
      prev = c;
    }
  }
  if (start + n == len && tid == DP_THREADS - 1 && txt[n - 1] != '\n') irregular = true;
  s_scan[tid] = nl; s_scan2[tid] = nt;
  __syncthreads();
  for (uint32_t off = 1; off < DP_THREADS; off <<= 1) {
    const uint32_t v = tid >= off ? s_scan[tid - off] : 0, v2 = tid >= off ? s_scan2[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v; s_scan2[tid] += v2;
    __syncthreads();
  }
  const uint32_t nlines = s_scan[DP_THREADS - 1], ntok = s_scan2[DP_THREADS - 1];
  if (nlines > DP_MAXLINES || ntok > DP_MAXTOK) irregular = true;   /* (degenerate text: the host's) */
  if (irregular) atomicOr(&s_irregular, 1u);
  if (nlines <= DP_MAXLINES && ntok <= DP_MAXTOK) {
    uint32_t wl = s_scan[tid] - nl, wt = s_scan2[tid] - nt;
    uint8_t prev = b0 ? txt[b0 - 1] : (uint8_t)'\n';
    for (uint32_t i = b0; i < b1; ++i) {
      const uint8_t c = txt[i];
      if (prev == '\n') line_start[wl++] = (uint16_t)i;
      if (c != ' ' && c != '\n' && (prev == ' ' || prev == '\n')) {
        tok_start[wt] = (uint16_t)i;
        tok_line[wt] = (uint16_t)(wl - 1u);   /* a line start precedes every token: wl >= 1 */
        ++wt;
      }
      prev = c;
    }
  }
  for (uint32_t j = tid; j < DP_MAXLINES; j += DP_THREADS) first_tok[j] = 0xFFFFu;
  __syncthreads();
  if (s_irregular) {
    if (tid == 0) atomicOr(res + 1, 1ull);
    if (!EMIT && tid == 0) cand_cnt[blockIdx.x] = 0;
    return;
  }
  /* ---- 2. per token: separator / first of its line / candidate; scan ---- */
  for (uint32_t base = 0; base < ntok; base += DP_THREADS) {
    const uint32_t t = base + tid;
    uint32_t cand = 0;
    if (t < ntok) {
      const uint32_t s0 = tok_start[t];
      const bool first = t == 0 || tok_line[t - 1] != tok_line[t];
      const uint8_t c1 = s0 + 1u < n ? txt[s0 + 1u] : (uint8_t)'\n';
      const bool semi = txt[s0] == ';' && (c1 == ' ' || c1 == '\n');
      if (first) first_tok[tok_line[t]] = (uint16_t)t;
      cand = (!first && !semi) ? 1u : 0u;
    }
    s_scan[tid] = cand;
    __syncthreads();
    for (uint32_t off = 1; off < DP_THREADS; off <<= 1) {
      const uint32_t v = tid >= off ? s_scan[tid - off] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    if (t < ntok) tok_cand[t] = (uint16_t)(s_carry + s_scan[tid]);   /* inclusive */
    __syncthreads();
    if (tid == DP_THREADS - 1) s_carry += s_scan[tid];
    __syncthreads();
  }
  if (!EMIT) {
    /* a line above the reference's line buffer is the host's */
    for (uint32_t j = tid; j < nlines; j += DP_THREADS)
      if ((j + 1 < nlines ? line_start[j + 1] : n) - line_start[j] > DP_MAXLINE) atomicOr(res + 1, 1ull);
    if (tid == 0) cand_cnt[blockIdx.x] = s_carry;
    return;
  }
  /* ---- 3. per line: the root; lines without a second token ---- */
  unsigned long long first_err = ~0ull;
  for (uint32_t j = tid; j < nlines; j += DP_THREADS) {
    const uint32_t f = first_tok[j];
    uint32_t root = DP_NONE, ntl = 0;
    if (f != 0xFFFFu) {
      uint32_t e = tok_start[f];
      /* the reference's token loop starts AT the root field (parser.c:338): a
         first token that looks like a separator or like a record is scanned as
         one there -- such a line is the host's */
      bool odd = txt[e] == ';';
      while (e < n && txt[e] != ' ' && txt[e] != '\n') { odd |= txt[e] == ','; ++e; }
      if (odd) atomicOr(res + 1, 1ull);
      root = dp_lookup(txt, tok_start[f], e, names, name_off, table, mask);
      /* tokens of the line: up to the next line that has one */
      uint32_t g = f + 1;
      ntl = 1;
      if (g < ntok && tok_line[g] == j) ntl = 2;
    }
    root_of[j] = root;
    /* a line without a second token, whatever its first one (parser.c:203-208) */
    if (ntl < 2) {
      const unsigned long long at = ((unsigned long long)(start + line_start[j]) << 4) | DP_ERR_RECORD;
      if (at < first_err) first_err = at;
    }
  }
  __syncthreads();
  /* ---- 4. per record token ---- */
  const uint64_t base_slot = cand_base[blockIdx.x];
  uint32_t nvalid = 0;
  bool irr = false;
  for (uint32_t t = tid; t < ntok; t += DP_THREADS) {
    const uint32_t line = tok_line[t], f = first_tok[line];
    if (t == f) continue;
    const uint32_t s0 = tok_start[t];
    uint32_t e = s0;
    while (e < n && txt[e] != ' ' && txt[e] != '\n') ++e;
    DpRecord r;
    const int kind = dp_token(txt, s0, e, r);
    if (kind == DP_TOK_SEMI) continue;
    const uint64_t slot = base_slot + tok_cand[t] - 1u;
    const uint32_t root = root_of[line];
    /* separators between the line's first token and this one */
    const uint32_t nsemi = (t - f) - ((uint32_t)tok_cand[t] - (uint32_t)tok_cand[f]);
    const bool sense = (nsemi & 1u) == 0;
    bool ok = false;
    if (kind == DP_TOK_IRREGULAR) irr = true;
    else if (root != DP_NONE) {
      const unsigned long long at = (unsigned long long)(start + s0) << 4;
      if (kind == DP_TOK_FAIL) { if ((at | DP_ERR_RECORD) < first_err) first_err = at | DP_ERR_RECORD; }
      else if (r.np < 0) { if ((at | DP_ERR_PAIRS) < first_err) first_err = at | DP_ERR_PAIRS; }
      else if (r.last != '+' && r.last != '-') { if ((at | DP_ERR_SIGN) < first_err) first_err = at | DP_ERR_SIGN; }
      else {
        const uint32_t ctg = dp_lookup(txt, r.h0, r.h1, names, name_off, table, mask);
        if (ctg != DP_NONE) {
          ok = true;
          o_root[slot] = root; o_ctg[slot] = ctg; o_dist[slot] = r.dist; o_np[slot] = r.np;
          o_sd[slot] = r.sd;
          o_flags[slot] = (uint8_t)((sense ? 1u : 0u) | (r.last == '+' ? 2u : 0u));
        }
      }
    }
    o_valid[slot] = ok ? 1 : 0;
    nvalid += ok ? 1u : 0u;
  }
  if (irr) atomicOr(res + 1, 1ull);
  if (first_err != ~0ull) atomicMin(res + 0, first_err);
  if (nvalid) atomicAdd(res + 2, (unsigned long long)nvalid);
}

/* ---- A-statistic file (ref algorithms.c:108-153) --------------------------------
   One record per line, scanned with "%s\t%ld\t%ld\t%ld\t%f\t%f" after the
   line's last character is dropped (:121): header, three integers, copy number,
   A-statistic; a contig that is found takes the two values (:139-144), a line
   that does not scan is "Invalid record".  A thread per line (six tokens each:
   the lanes stay together).  Regular: fields separated by blanks or tabs, the
   integers plain, the two numbers as dp_float takes them, every contig named
   once (the reference lets the last line win; a second mention goes to the host). */
__global__ void __launch_bounds__(DP_THREADS)
k_dp_astat(const char *text, uint64_t len, const char *names, const uint32_t *name_off, const uint64_t *table,
           uint64_t mask, float *astat, float *copy_num, uint32_t *seen, unsigned long long *res)
{
  __shared__ __attribute__((aligned(16))) uint8_t sbuf[DP_BUF];
  __shared__ uint16_t line_start[DP_MAXLINES + 1];
  __shared__ uint32_t s_scan[DP_THREADS];
  __shared__ uint32_t s_first[2], s_irregular;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_irregular = 0;
  uint64_t start;
  uint32_t n;
  if (!dp_stage_stride(text, len, sbuf, s_first, res, start, n)) return;
  DpText txt(sbuf, (uint32_t)(start - (start & ~15ull)));
  const uint32_t q = (n + DP_THREADS - 1) / DP_THREADS;
  const uint32_t b0 = tid * q < n ? tid * q : n, b1 = b0 + q < n ? b0 + q : n;
  uint32_t nl = 0;
  bool irregular = false;
  {
    uint8_t prev = b0 ? txt[b0 - 1] : (uint8_t)'\n';
    for (uint32_t i = b0; i < b1; ++i) {
      const uint8_t c = txt[i];
      if (prev == '\n') ++nl;
      if ((c < 0x20 && c != '\n' && c != '\t') || c == 0x7F) irregular = true;
      prev = c;
    }
  }
  if (start + n == len && tid == DP_THREADS - 1 && txt[n - 1] != '\n') irregular = true;
  s_scan[tid] = nl;
  __syncthreads();
  for (uint32_t off = 1; off < DP_THREADS; off <<= 1) {
    const uint32_t v = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  const uint32_t nlines = s_scan[DP_THREADS - 1];
  if (nlines > DP_MAXLINES) irregular = true;
  if (irregular) atomicOr(&s_irregular, 1u);
  if (nlines <= DP_MAXLINES) {
    uint32_t wl = s_scan[tid] - nl;
    uint8_t prev = b0 ? txt[b0 - 1] : (uint8_t)'\n';
    for (uint32_t i = b0; i < b1; ++i) {
      if (prev == '\n') line_start[wl++] = (uint16_t)i;
      prev = txt[i];
    }
  }
  __syncthreads();
  if (s_irregular) {
    if (tid == 0) atomicOr(res + 1, 1ull);
    return;
  }
  unsigned long long first_err = ~0ull;
  uint32_t nvalid = 0;
  bool irr = false;
  for (uint32_t j = tid; j < nlines; j += DP_THREADS) {
    const uint32_t ls = line_start[j], le = (j + 1 < nlines ? line_start[j + 1] : n) - 1u;
    if (le + 1u - ls > DP_MAXLINE) { irr = true; continue; }
    uint32_t i = ls, h0 = 0, h1 = 0;
    int64_t iv;
    float cn = 0.0f, as = 0.0f;
    int state = 0;   /* 0 ok, 1 the line does not scan, 2 irregular */
    for (uint32_t f = 0; f < 6 && state == 0; ++f) {
      while (i < le && (txt[i] == ' ' || txt[i] == '\t')) ++i;
      if (i >= le) { state = 1; break; }
      const uint32_t s0 = i;
      while (i < le && txt[i] != ' ' && txt[i] != '\t') ++i;
      uint32_t k = s0;
      if (f == 0) { h0 = s0; h1 = i; }
      else if (f < 4) {
        const int rc = dp_int(txt, k, i, iv);
        if (rc == 2 || (rc == 0 && k != i)) state = 2;   /* "12x": the next %ld starts inside the token */
        else if (rc == 1) state = 1;
      } else {
        const int rc = dp_float(txt, k, i, f == 4 ? cn : as);
        if (rc) state = rc;
      }
    }
    if (state == 2) { irr = true; continue; }
    if (state == 1) {
      const unsigned long long at = ((unsigned long long)(start + ls) << 4) | DP_ERR_RECORD;
      if (at < first_err) first_err = at;
      continue;
    }
    const uint32_t id = dp_lookup(txt, h0, h1, names, name_off, table, mask);
    if (id == DP_NONE) continue;
    if (atomicAdd(&seen[id], 1u) != 0u) { irr = true; continue; }
    astat[id] = as; copy_num[id] = cn;
    ++nvalid;
  }
  if (irr) atomicOr(res + 1, 1ull);
  if (first_err != ~0ull) atomicMin(res + 0, first_err);
  if (nvalid) atomicAdd(res + 2, (unsigned long long)nvalid);
}

/* ---- contig headers in strcmp order (ref parser.c:172: qsort of the vertices) ----
   Keys: the first 14 bytes of a name, big endian, seven to a 64-bit word (the
   sort keeps bit 63 out of the order), zero padded -- a shorter name sorts
   before the names it is a prefix of, as with strcmp.  Two stable 7-pass
   sorts (low word, then high word); names that agree in all 14 bytes are
   flagged and put in order by the caller. */
__global__ void k_dp_name_keys(const char *names, const uint32_t *off, uint64_t n, uint64_t *khi, uint64_t *klo,
                               uint32_t *idx)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = off[i], len = off[i + 1] - o;
  uint64_t hi = 0, lo = 0;
  for (uint32_t k = 0; k < 7; ++k) hi = (hi << 8) | (k < len ? (uint8_t)names[o + k] : 0u);
  for (uint32_t k = 7; k < 14; ++k) lo = (lo << 8) | (k < len ? (uint8_t)names[o + k] : 0u);
  khi[i] = hi; klo[i] = lo; idx[i] = (uint32_t)i;
}
__global__ void k_dp_gather_u64(const uint64_t *src, const uint32_t *idx, uint64_t n, uint64_t *dst)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_dp_name_ties(const uint64_t *khi_sorted, const uint64_t *klo, const uint32_t *perm, uint64_t n,
                               uint8_t *tie)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  tie[i] = i > 0 && khi_sorted[i] == khi_sorted[i - 1] && klo[perm[i]] == klo[perm[i - 1]] ? 1 : 0;
}

/* ---- FASTA: the record table (ref parser.c:399-494, the callback loops) -----------
   A record starts at a '>' that is not inside a description line, i.e. at the
   first '>' since the last newline; its description runs to the next newline;
   its sequence is every later byte up to the next record start other than
   newline, carriage return and blank.  Lines can be of any length, so the state
   "a '>' has been seen since the last newline" is carried over the strides:
   k_fa_summary reduces every stride to (has a newline, '>' after its last
   newline or anywhere if none, '>' before its first newline, lines after the
   first newline that hold a '>'), one thread chains the strides, k_fa_records
   walks every stride with the state it starts in and writes, per record, the
   offsets of its description and, summed in LDS first, its sequence length. */
#define FA_STRIDE 16384u
#define FA_THREADS 256u
#define FA_SLICE (FA_STRIDE / FA_THREADS)
#define FA_LREC 1024u

struct FaSum { uint32_t n, a, f, s; };   /* see above; s counts lines */
__device__ __forceinline__ FaSum fa_join(const FaSum &x, const FaSum &y)
{
  /* x then y */
  FaSum r;
  r.n = x.n | y.n;
  r.a = y.n ? y.a : (x.a | y.a);
  r.f = x.n ? x.f : (x.f | y.f);
  /* lines of the join after its first newline that hold a '>': those of x,
     those of y, and the line x ends / y begins with if x has a newline */
  r.s = x.s + y.s + ((x.n && !x.a && y.f) ? 1u : 0u);
  return r;
}
__device__ __forceinline__ FaSum fa_slice(const char *text, uint64_t lo, uint64_t hi)
{
  FaSum r = {0, 0, 0, 0};
  bool gt = false;      /* '>' since the last newline of the slice (or its start) */
  /* eight bytes a load (lo is a multiple of FA_SLICE, the buffer has 16 bytes of slack) */
  for (uint64_t q = lo; q < hi; q += 8) {
    uint64_t w = *(const uint64_t *)(text + q);
    const uint32_t m = hi - q < 8 ? (uint32_t)(hi - q) : 8u;
    for (uint32_t j = 0; j < m; ++j, w >>= 8) {
      const char c = (char)(w & 0xFFu);
      if (c == '\n') { r.n = 1; gt = false; }
      else if (c == '>' && !gt) { gt = true; if (r.n) ++r.s; else r.f = 1; }
    }
  }
  r.a = gt ? 1u : 0u;
  return r;
}
/* block reduction of the slices' summaries in order; every thread gets the
   summary of the slices before its own (sh[tid] exclusive) */
__device__ __forceinline__ FaSum fa_block_scan(FaSum mine, FaSum *sh, FaSum *total)
{
  const uint32_t tid = threadIdx.x;
  sh[tid] = mine;
  __syncthreads();
  for (uint32_t off = 1; off < FA_THREADS; off <<= 1) {
    FaSum v = sh[tid];
    if (tid >= off) v = fa_join(sh[tid - off], sh[tid]);
    __syncthreads();
    sh[tid] = v;
    __syncthreads();
  }
  *total = sh[FA_THREADS - 1];
  FaSum before = {0, 0, 0, 0};
  if (tid) before = sh[tid - 1];
  __syncthreads();
  return before;
}
__global__ void __launch_bounds__(FA_THREADS)
k_fa_summary(const char *text, uint64_t len, FaSum *sums)
{
  __shared__ FaSum sh[FA_THREADS];
  const uint64_t lo = (uint64_t)blockIdx.x * FA_STRIDE + (uint64_t)threadIdx.x * FA_SLICE;
  const uint64_t hi = lo + FA_SLICE < len ? lo + FA_SLICE : len;
  FaSum total;
  fa_block_scan(lo < len ? fa_slice(text, lo, hi) : FaSum{0, 0, 0, 0}, sh, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
/* carry[b] = '>' seen since the last newline at the start of stride b;
   base[b] = records that start before stride b.  One thread: a few hundred
   thousand strides for a file of gigabytes. */
__global__ void k_fa_chain(const FaSum *sums, uint64_t nblocks, uint8_t *carry, uint64_t *base, uint64_t *nrec)
{
  if (threadIdx.x || blockIdx.x) return;
  uint32_t c = 0;
  uint64_t r = 0;
  for (uint64_t b = 0; b < nblocks; ++b) {
    const FaSum s = sums[b];
    carry[b] = (uint8_t)c;
    base[b] = r;
    r += s.s + ((s.f && !c) ? 1u : 0u);
    c = s.n ? s.a : (c | s.a);
  }
  *nrec = r;
}
__global__ void __launch_bounds__(FA_THREADS)
k_fa_records(const char *text, uint64_t len, const uint8_t *carry, const uint64_t *base, uint64_t *hs,
             uint64_t *he, unsigned long long *slen)
{
  __shared__ FaSum sh[FA_THREADS];
  __shared__ unsigned long long lsum[FA_LREC];
  const uint32_t tid = threadIdx.x;
  const uint64_t lo = (uint64_t)blockIdx.x * FA_STRIDE + (uint64_t)tid * FA_SLICE;
  const uint64_t hi = lo + FA_SLICE < len ? lo + FA_SLICE : len;
  for (uint32_t k = tid; k < FA_LREC; k += FA_THREADS) lsum[k] = 0;
  FaSum total;
  const FaSum before = fa_block_scan(lo < len ? fa_slice(text, lo, hi) : FaSum{0, 0, 0, 0}, sh, &total);
  const uint32_t c0 = carry[blockIdx.x];
  const uint64_t b0 = base[blockIdx.x];
  /* state at the slice's first byte */
  bool gt = before.n ? before.a != 0 : (c0 | before.a) != 0;
  /* records started before the slice: those before the stride + in the slices before */
  uint64_t rec = b0 + before.s + ((before.f && !c0) ? 1u : 0u);   /* index of the NEXT record */
  /* a description is open iff a '>' has been seen since the last newline */
  unsigned long long cnt = 0;
  for (uint64_t q = lo; q < hi; q += 8) {
    uint64_t w = *(const uint64_t *)(text + q);
    const uint32_t m = hi - q < 8 ? (uint32_t)(hi - q) : 8u;
    for (uint32_t j = 0; j < m; ++j, w >>= 8) {
      const char c = (char)(w & 0xFFu);
      const uint64_t p = q + j;
      if (c == '\n') {
        if (gt && rec) he[rec - 1] = p;   /* (he is preset to "no newline": all ones) */
        gt = false;
      } else if (c == '>' && !gt) {
        if (cnt && rec) {
          const uint64_t l = rec - b0;   /* slot 0: the record open at the stride's start */
          if (l < FA_LREC) atomicAdd(&lsum[l], cnt); else atomicAdd(&slen[rec - 1], cnt);
        }
        cnt = 0;
        gt = true;
        hs[rec] = p + 1;
        ++rec;
      } else if (!gt && c != '\r' && c != ' ') ++cnt;
    }
  }
  if (cnt && rec) {
    const uint64_t l = rec - b0;
    if (l < FA_LREC) atomicAdd(&lsum[l], cnt); else atomicAdd(&slen[rec - 1], cnt);
  }
  __syncthreads();
  for (uint32_t k = tid; k < FA_LREC; k += FA_THREADS)
    if (lsum[k] && b0 + k >= 1) atomicAdd(&slen[b0 + k - 1], lsum[k]);
}

__global__ void k_dp_compact(const uint8_t *valid, const uint32_t *pos, uint64_t n, const uint32_t *root,
                             const uint32_t *ctg, const int64_t *dist, const int64_t *np, const float *sd,
                             const uint8_t *flags, uint32_t *root2, uint32_t *ctg2, int64_t *dist2,
                             int64_t *np2, float *sd2, uint8_t *flags2)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !valid[i]) return;
  const uint32_t o = pos[i];
  root2[o] = root[i]; ctg2[o] = ctg[i]; dist2[o] = dist[i]; np2[o] = np[i]; sd2[o] = sd[i];
  flags2[o] = flags[i];
}

/* ---- C ABI -------------------------------------------------------------------- */
extern "C" {

int gtsg_deparser_create(GtsgDeParser **out, int device, void *stream)
{
  if (!out) return GTSG_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return GTSG_EHIP;
  GtsgDeParser *p = new GtsgDeParser();
  memset(p, 0, sizeof *p);
  p->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete p; return GTSG_EHIP; }
  if (stream) p->st = (hipStream_t)stream;
  else {
    if (hipStreamCreate(&p->st) != hipSuccess) { delete p; return GTSG_EHIP; }
    p->own_stream = true;
  }
  if (hipMalloc((void **)&p->d_res, 64) != hipSuccess) {
    if (p->own_stream) hipStreamDestroy(p->st);
    delete p;
    return GTSG_ENOMEM;
  }
  *out = p;
  return 0;
}

/* scratch of the parser's prefix sums: kept between calls (a file in pieces is
   one call per piece), grown when a call needs more */
static int dp_scan_scratch(GtsgDeParser *p, uint64_t elems)
{
  if (p->scan_cap >= elems) return 0;
  if (p->scan_tmp) hipFree(p->scan_tmp);
  p->scan_tmp = nullptr; p->scan_cap = 0;
  if (hipMalloc((void **)&p->scan_tmp, elems * sizeof(uint32_t)) != hipSuccess) {
    snprintf(p->err, sizeof p->err, "out of device memory (scan scratch)");
    return GTSG_ENOMEM;
  }
  p->scan_cap = elems;
  return 0;
}
static void dp_free_acc(GtsgDeParser *p)
{
  void *ptrs[] = {p->a_root, p->a_ctg, p->a_dist, p->a_np, p->a_sd, p->a_flags};
  for (void *q : ptrs) if (q) hipFree(q);
  p->a_root = p->a_ctg = nullptr; p->a_dist = p->a_np = nullptr; p->a_sd = nullptr; p->a_flags = nullptr;
  p->a_n = p->a_cap = 0;
}

static void dp_free_parse(GtsgDeParser *p)
{
  void *ptrs[] = {p->root, p->ctg, p->dist, p->np, p->sd, p->flags, p->valid, p->cand_cnt, p->scan_tmp,
                  p->cpos, p->root2, p->ctg2, p->dist2, p->np2, p->sd2, p->flags2};
  for (void *q : ptrs) if (q) hipFree(q);
  p->root = p->ctg = p->root2 = p->ctg2 = p->cand_cnt = p->scan_tmp = p->cpos = nullptr;
  p->scan_cap = p->cpos_cap = 0;
  p->dist = p->np = p->dist2 = p->np2 = nullptr;
  p->sd = p->sd2 = nullptr;
  p->flags = p->valid = p->flags2 = nullptr;
  p->cap_cand = p->cap_blocks = p->cap2 = 0;
}

void gtsg_deparser_destroy(GtsgDeParser *p)
{
  if (!p) return;
  hipSetDevice(p->device);
  dp_free_parse(p);
  dp_free_acc(p);
  if (p->names) hipFree(p->names);
  if (p->name_off) hipFree(p->name_off);
  if (p->table) hipFree(p->table);
  if (p->text) hipFree(p->text);
  if (p->d_res) hipFree(p->d_res);
  if (p->own_stream) hipStreamDestroy(p->st);
  delete p;
}

const char *gtsg_deparser_last_error(const GtsgDeParser *p) { return p ? p->err : "no parser"; }

/* on != 0: from now on the records of every gtsg_deparser_parse are appended to
   those of the parses before (a file handed over in pieces that end at line
   ends); gtsg_deparser_records then returns all of them.  on == 0: back to
   "the last parse", the collected records are dropped. */
int gtsg_deparser_accumulate(GtsgDeParser *p, int on)
{
  if (!p) return GTSG_EINVAL;
  hipSetDevice(p->device);
  dp_free_acc(p);
  p->accumulate = on != 0;
  return 0;
}

void gtsg_deparser_trim(GtsgDeParser *p)
{
  if (!p) return;
  hipSetDevice(p->device);
  dp_free_acc(p);
  p->accumulate = false;
  dp_free_parse(p);
  if (p->text) hipFree(p->text);
  p->text = nullptr; p->text_cap = 0;
  p->n_records = 0; p->compacted = false;
}

int gtsg_deparser_set_names(GtsgDeParser *p, const char *blob, const uint64_t *offsets, uint64_t n)
{
  if (!p || (n && (!blob || !offsets))) return GTSG_EINVAL;
  DPCHK(hipSetDevice(p->device));
  if (p->names) { hipFree(p->names); p->names = nullptr; }
  if (p->name_off) { hipFree(p->name_off); p->name_off = nullptr; }
  if (p->table) { hipFree(p->table); p->table = nullptr; }
  p->n_names = 0;
  const uint64_t bytes = n ? offsets[n] : 0;
  if (n >= 0x7FFFFFFFull || bytes >= 0xFFFFFFFFull) return dp_fail(p, GTSG_ELIMIT, "too many contig names");
  uint64_t size = 16;
  while (size < 2 * n + 2) size <<= 1;
  DPCHK(hipMalloc((void **)&p->names, bytes + 16));
  DPCHK(hipMalloc((void **)&p->name_off, (n + 1) * sizeof(uint32_t)));
  DPCHK(hipMalloc((void **)&p->table, size * sizeof(uint64_t)));
  uint32_t *off32 = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
  if (!off32) return dp_fail(p, GTSG_ENOMEM, "out of host memory");
  for (uint64_t i = 0; i <= n; ++i) off32[i] = (uint32_t)(n ? offsets[i] : 0);
  hipError_t e1 = hipMemcpyAsync(p->name_off, off32, (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, p->st);
  hipError_t e2 = bytes ? hipMemcpyAsync(p->names, blob, bytes, hipMemcpyHostToDevice, p->st) : hipSuccess;
  hipError_t e3 = hipMemsetAsync(p->table, 0, size * sizeof(uint64_t), p->st);
  if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && n)
    k_dp_insert<<<(uint32_t)((n + 255) / 256), 256, 0, p->st>>>(p->names, p->name_off, n, p->table, size - 1);
  hipError_t e4 = hipStreamSynchronize(p->st);
  free(off32);
  DPCHK(e1); DPCHK(e2); DPCHK(e3); DPCHK(e4);
  DPCHK(hipGetLastError());
  p->table_mask = size - 1;
  p->n_names = n;
  return 0;
}

int gtsg_deparser_parse(GtsgDeParser *p, const char *text, uint64_t len, int on_device, GtsgDeParseResult *res)
{
  if (!p || !res || (len && !text)) return GTSG_EINVAL;
  memset(res, 0, sizeof *res);
  DPCHK(hipSetDevice(p->device));
  if (!p->table) return dp_fail(p, GTSG_EINVAL, "gtsg_deparser_set_names has not been called");
  p->n_records = 0;
  p->compacted = false;
  if (len == 0) return 0;
  /* candidate slots are numbered in 32 bits, a stride has fewer than 2^14 */
  if (len >= (1ull << 32)) return dp_fail(p, GTSG_ELIMIT, "distance file of 4 GB or more");
  const char *d_text = text;
  if (!on_device) {
    if (p->text_cap < len + 16) {
      if (p->text) hipFree(p->text);
      p->text = nullptr; p->text_cap = 0;
      DPCHK(hipMalloc((void **)&p->text, len + 16));
      p->text_cap = len + 16;
    }
    DPCHK(hipMemcpyAsync(p->text, text, len, hipMemcpyHostToDevice, p->st));
    d_text = p->text;
  }
  const uint64_t nblocks = (len + DP_STRIDE - 1) / DP_STRIDE;
  if (nblocks >= 0x7FFFFFFFull) return dp_fail(p, GTSG_ELIMIT, "distance file too large");
  if (p->cap_blocks < nblocks + 1) {
    if (p->cand_cnt) hipFree(p->cand_cnt);
    p->cand_cnt = nullptr; p->cap_blocks = 0;
    DPCHK(hipMalloc((void **)&p->cand_cnt, (nblocks + 1) * sizeof(uint32_t)));
    p->cap_blocks = nblocks + 1;
  }
  unsigned long long init[4] = {~0ull, 0ull, 0ull, 0ull};
  DPCHK(hipMemcpyAsync(p->d_res, init, sizeof init, hipMemcpyHostToDevice, p->st));
  k_dp_stride<false><<<(uint32_t)nblocks, DP_THREADS, 0, p->st>>>(
      d_text, len, p->cand_cnt, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
      nullptr, nullptr, nullptr, p->d_res);
  /* candidates per stride -> bases; a text of n bytes holds fewer than n / 2 tokens */
  {
    const int rcs = dp_scan_scratch(p, gts_scan_tmp_elems(nblocks + 1) + 2);
    if (rcs) return rcs;
    gts_exscan<uint32_t, uint32_t>(p->cand_cnt, p->cand_cnt, nblocks, p->scan_tmp, (uint32_t *)(p->d_res + 3), p->st);
  }
  unsigned long long h[4];
  DPCHK(hipMemcpyAsync(h, p->d_res, sizeof h, hipMemcpyDeviceToHost, p->st));
  DPCHK(hipStreamSynchronize(p->st));
  DPCHK(hipGetLastError());
  const uint64_t ncand = (uint32_t)h[3];
  res->n_candidates = ncand;
  if (h[1]) { res->irregular = 1; return 0; }
  if (ncand >= (1ull << 31)) return dp_fail(p, GTSG_ELIMIT, "2^31 records or more in the distance file");
  if (p->cap_cand < ncand + 1) {
    void *ptrs[] = {p->root, p->ctg, p->dist, p->np, p->sd, p->flags, p->valid};
    for (void *q : ptrs) if (q) hipFree(q);
    p->root = p->ctg = nullptr; p->dist = p->np = nullptr; p->sd = nullptr; p->flags = p->valid = nullptr;
    p->cap_cand = 0;
    const uint64_t c = ncand + 1;
    DPCHK(hipMalloc((void **)&p->root, c * 4)); DPCHK(hipMalloc((void **)&p->ctg, c * 4));
    DPCHK(hipMalloc((void **)&p->dist, c * 8)); DPCHK(hipMalloc((void **)&p->np, c * 8));
    DPCHK(hipMalloc((void **)&p->sd, c * 4)); DPCHK(hipMalloc((void **)&p->flags, c));
    DPCHK(hipMalloc((void **)&p->valid, c));
    p->cap_cand = c;
  }
  DPCHK(hipMemsetAsync(p->d_res + 3, 0, 8, p->st));
  k_dp_stride<true><<<(uint32_t)nblocks, DP_THREADS, 0, p->st>>>(
      d_text, len, nullptr, p->cand_cnt, p->names, p->name_off, p->table, p->table_mask, p->root, p->ctg,
      p->dist, p->np, p->sd, p->flags, p->valid, p->d_res);
  DPCHK(hipMemcpyAsync(h, p->d_res, sizeof h, hipMemcpyDeviceToHost, p->st));
  DPCHK(hipStreamSynchronize(p->st));
  DPCHK(hipGetLastError());
  if (h[1]) { res->irregular = 1; return 0; }
  if (h[0] != ~0ull) {
    res->error = (int)(h[0] & 15u);
    res->error_pos = h[0] >> 4;
    return 0;
  }
  const uint64_t nvalid = h[2];
  res->n_records = nvalid;
  p->n_records = nvalid;
  if (nvalid != ncand) {
    /* some token is not a record (unknown contig): close the gaps, order kept */
    if (p->cap2 < nvalid + 1) {
      void *ptrs[] = {p->root2, p->ctg2, p->dist2, p->np2, p->sd2, p->flags2};
      for (void *q : ptrs) if (q) hipFree(q);
      p->root2 = p->ctg2 = nullptr; p->dist2 = p->np2 = nullptr; p->sd2 = nullptr; p->flags2 = nullptr;
      p->cap2 = 0;
      const uint64_t c = nvalid + 1;
      DPCHK(hipMalloc((void **)&p->root2, c * 4)); DPCHK(hipMalloc((void **)&p->ctg2, c * 4));
      DPCHK(hipMalloc((void **)&p->dist2, c * 8)); DPCHK(hipMalloc((void **)&p->np2, c * 8));
      DPCHK(hipMalloc((void **)&p->sd2, c * 4)); DPCHK(hipMalloc((void **)&p->flags2, c));
      p->cap2 = c;
    }
    if (p->cpos_cap < ncand + 1) {
      if (p->cpos) hipFree(p->cpos);
      p->cpos = nullptr; p->cpos_cap = 0;
      DPCHK(hipMalloc((void **)&p->cpos, (ncand + 1) * 4));
      p->cpos_cap = ncand + 1;
    }
    {
      const int rcs = dp_scan_scratch(p, gts_scan_tmp_elems(ncand) + 2);
      if (rcs) return rcs;
    }
    gts_exscan<uint8_t, uint32_t>(p->valid, p->cpos, ncand, p->scan_tmp, (uint32_t *)nullptr, p->st);
    k_dp_compact<<<(uint32_t)((ncand + 255) / 256), 256, 0, p->st>>>(
        p->valid, p->cpos, ncand, p->root, p->ctg, p->dist, p->np, p->sd, p->flags, p->root2, p->ctg2, p->dist2,
        p->np2, p->sd2, p->flags2);
    DPCHK(hipStreamSynchronize(p->st));
    DPCHK(hipGetLastError());
    p->compacted = true;
  }
  if (p->accumulate && nvalid) {
    const uint64_t need = p->a_n + nvalid;
    if (need > p->a_cap) {
      uint64_t cap = p->a_cap ? p->a_cap : 1u << 20;
      while (cap < need) cap *= 2;
      uint32_t *r2 = nullptr, *c2 = nullptr; int64_t *d2 = nullptr, *n2 = nullptr; float *s2 = nullptr; uint8_t *f2 = nullptr;
      if (hipMalloc((void **)&r2, cap * 4) != hipSuccess || hipMalloc((void **)&c2, cap * 4) != hipSuccess ||
          hipMalloc((void **)&d2, cap * 8) != hipSuccess || hipMalloc((void **)&n2, cap * 8) != hipSuccess ||
          hipMalloc((void **)&s2, cap * 4) != hipSuccess || hipMalloc((void **)&f2, cap) != hipSuccess) {
        void *tmpv[] = {r2, c2, d2, n2, s2, f2};
        for (void *q : tmpv) if (q) hipFree(q);
        return dp_fail(p, GTSG_ENOMEM, "out of device memory (record accumulator)");
      }
      if (p->a_n) {
        DPCHK(hipMemcpyAsync(r2, p->a_root, p->a_n * 4, hipMemcpyDeviceToDevice, p->st));
        DPCHK(hipMemcpyAsync(c2, p->a_ctg, p->a_n * 4, hipMemcpyDeviceToDevice, p->st));
        DPCHK(hipMemcpyAsync(d2, p->a_dist, p->a_n * 8, hipMemcpyDeviceToDevice, p->st));
        DPCHK(hipMemcpyAsync(n2, p->a_np, p->a_n * 8, hipMemcpyDeviceToDevice, p->st));
        DPCHK(hipMemcpyAsync(s2, p->a_sd, p->a_n * 4, hipMemcpyDeviceToDevice, p->st));
        DPCHK(hipMemcpyAsync(f2, p->a_flags, p->a_n, hipMemcpyDeviceToDevice, p->st));
        DPCHK(hipStreamSynchronize(p->st));
      }
      const uint64_t keep = p->a_n;
      dp_free_acc(p);
      p->a_root = r2; p->a_ctg = c2; p->a_dist = d2; p->a_np = n2; p->a_sd = s2; p->a_flags = f2;
      p->a_n = keep; p->a_cap = cap;
    }
    const bool c = p->compacted;
    DPCHK(hipMemcpyAsync(p->a_root + p->a_n, c ? p->root2 : p->root, nvalid * 4, hipMemcpyDeviceToDevice, p->st));
    DPCHK(hipMemcpyAsync(p->a_ctg + p->a_n, c ? p->ctg2 : p->ctg, nvalid * 4, hipMemcpyDeviceToDevice, p->st));
    DPCHK(hipMemcpyAsync(p->a_dist + p->a_n, c ? p->dist2 : p->dist, nvalid * 8, hipMemcpyDeviceToDevice, p->st));
    DPCHK(hipMemcpyAsync(p->a_np + p->a_n, c ? p->np2 : p->np, nvalid * 8, hipMemcpyDeviceToDevice, p->st));
    DPCHK(hipMemcpyAsync(p->a_sd + p->a_n, c ? p->sd2 : p->sd, nvalid * 4, hipMemcpyDeviceToDevice, p->st));
    DPCHK(hipMemcpyAsync(p->a_flags + p->a_n, c ? p->flags2 : p->flags, nvalid, hipMemcpyDeviceToDevice, p->st));
    DPCHK(hipStreamSynchronize(p->st));
    p->a_n += nvalid;
  }
  return 0;
}

int gtsg_deparser_records(const GtsgDeParser *p, uint64_t *n, const uint32_t **root, const uint32_t **ctg,
                          const int64_t **dist, const float **std_dev, const int64_t **num_pairs,
                          const uint8_t **flags)
{
  if (!p || !n) return GTSG_EINVAL;
  if (p->accumulate) {
    *n = p->a_n;
    if (root) *root = p->a_root;
    if (ctg) *ctg = p->a_ctg;
    if (dist) *dist = p->a_dist;
    if (std_dev) *std_dev = p->a_sd;
    if (num_pairs) *num_pairs = p->a_np;
    if (flags) *flags = p->a_flags;
    return 0;
  }
  *n = p->n_records;
  const bool c = p->compacted;
  if (root) *root = c ? p->root2 : p->root;
  if (ctg) *ctg = c ? p->ctg2 : p->ctg;
  if (dist) *dist = c ? p->dist2 : p->dist;
  if (std_dev) *std_dev = c ? p->sd2 : p->sd;
  if (num_pairs) *num_pairs = c ? p->np2 : p->np;
  if (flags) *flags = c ? p->flags2 : p->flags;
  return 0;
}

/* A-statistic file: astat / copy_num (n names, in and out; host pointers or,
   with arrays_on_device, device pointers) take the values of the lines whose
   contig is known.  res->error 1 = "Invalid record in A-statistic file",
   res->n_records = contigs set; res->irregular: nothing was written that the
   host code would not write as well, parse the file there. */
int gtsg_deparser_parse_astat(GtsgDeParser *p, const char *text, uint64_t len, int on_device, float *astat,
                              float *copy_num, int arrays_on_device, GtsgDeParseResult *res)
{
  if (!p || !res || (len && !text) || !astat || !copy_num) return GTSG_EINVAL;
  memset(res, 0, sizeof *res);
  DPCHK(hipSetDevice(p->device));
  if (!p->table) return dp_fail(p, GTSG_EINVAL, "gtsg_deparser_set_names has not been called");
  if (len == 0 || p->n_names == 0) return 0;
  if (len >= (1ull << 32)) return dp_fail(p, GTSG_ELIMIT, "A-statistic file of 4 GB or more");
  const char *d_text = text;
  if (!on_device) {
    if (p->text_cap < len + 16) {
      if (p->text) hipFree(p->text);
      p->text = nullptr; p->text_cap = 0;
      DPCHK(hipMalloc((void **)&p->text, len + 16));
      p->text_cap = len + 16;
    }
    DPCHK(hipMemcpyAsync(p->text, text, len, hipMemcpyHostToDevice, p->st));
    d_text = p->text;
  }
  const uint64_t n = p->n_names;
  float *d_as = astat, *d_cn = copy_num, *work = nullptr;
  uint32_t *seen = nullptr;
  /* the kernel writes to a working copy: an irregular file must leave the
     caller's arrays as they were */
  DPCHK(hipMalloc((void **)&work, 2 * n * sizeof(float)));
  if (hipMalloc((void **)&seen, n * sizeof(uint32_t)) != hipSuccess) {
    hipFree(work);
    return dp_fail(p, GTSG_ENOMEM, "out of device memory");
  }
  const hipMemcpyKind in = arrays_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const hipMemcpyKind back = arrays_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  hipError_t e1 = hipMemcpyAsync(work, d_as, n * sizeof(float), in, p->st);
  hipError_t e2 = hipMemcpyAsync(work + n, d_cn, n * sizeof(float), in, p->st);
  hipError_t e3 = hipMemsetAsync(seen, 0, n * sizeof(uint32_t), p->st);
  unsigned long long init[4] = {~0ull, 0ull, 0ull, 0ull}, h[4] = {0, 0, 0, 0};
  hipError_t e4 = hipMemcpyAsync(p->d_res, init, sizeof init, hipMemcpyHostToDevice, p->st);
  const uint64_t nblocks = (len + DP_STRIDE - 1) / DP_STRIDE;
  if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess)
    k_dp_astat<<<(uint32_t)nblocks, DP_THREADS, 0, p->st>>>(d_text, len, p->names, p->name_off, p->table,
                                                           p->table_mask, work, work + n, seen, p->d_res);
  hipError_t e5 = hipMemcpyAsync(h, p->d_res, sizeof h, hipMemcpyDeviceToHost, p->st);
  hipError_t e6 = hipStreamSynchronize(p->st);
  hipError_t e7 = hipGetLastError();
  int rc = 0;
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess ||
      e6 != hipSuccess || e7 != hipSuccess)
    rc = dp_fail(p, GTSG_EHIP, "HIP error in the A-statistic parser");
  else if (h[1]) res->irregular = 1;
  else if (h[0] != ~0ull) { res->error = (int)(h[0] & 15u); res->error_pos = h[0] >> 4; }
  else {
    res->n_records = h[2];
    if (hipMemcpy(d_as, work, n * sizeof(float), back) != hipSuccess ||
        hipMemcpy(d_cn, work + n, n * sizeof(float), back) != hipSuccess)
      rc = dp_fail(p, GTSG_EHIP, "HIP error in the A-statistic parser");
  }
  hipFree(work); hipFree(seen);
  return rc;
}

/* perm[j] = index of the name that is j-th in strcmp order by its first 14
   bytes; tie[j] != 0: name perm[j] agrees with name perm[j - 1] in those bytes
   (the caller orders such runs itself).  Host arrays of n elements. */
int gtsg_sort_names(int device, const char *blob, const uint64_t *offsets, uint64_t n, uint32_t *perm,
                    uint8_t *tie)
{
  if (n == 0) return 0;
  if (!blob || !offsets || !perm || !tie) return GTSG_EINVAL;
  if (n >= (1ull << 30) || offsets[n] >= 0xFFFFFFFFull) return GTSG_ELIMIT;
  if (hipSetDevice(device) != hipSuccess) return GTSG_EHIP;
  const uint64_t bytes = offsets[n];
  char *d_names = nullptr;
  uint32_t *d_off = nullptr, *v0 = nullptr, *v1 = nullptr, *tmp = nullptr;
  uint64_t *khi = nullptr, *klo = nullptr, *k0 = nullptr, *k1 = nullptr;
  uint8_t *d_tie = nullptr;
  uint32_t *off32 = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
  int rc = 0;
  hipStream_t st = nullptr;
  if (!off32) return GTSG_ENOMEM;
  for (uint64_t i = 0; i <= n; ++i) off32[i] = (uint32_t)offsets[i];
#define SN(x) do { if (!rc && (x) != hipSuccess) rc = GTSG_EHIP; } while (0)
  SN(hipMalloc((void **)&d_names, bytes + 16)); SN(hipMalloc((void **)&d_off, (n + 1) * 4));
  SN(hipMalloc((void **)&khi, n * 8)); SN(hipMalloc((void **)&klo, n * 8));
  SN(hipMalloc((void **)&k0, n * 8)); SN(hipMalloc((void **)&k1, n * 8));
  SN(hipMalloc((void **)&v0, n * 4)); SN(hipMalloc((void **)&v1, n * 4));
  SN(hipMalloc((void **)&tmp, gts_sort_tmp_elems(n) * 4)); SN(hipMalloc((void **)&d_tie, n));
  SN(hipStreamCreate(&st));
  if (!rc) {
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    int shifts[7] = {0, 8, 16, 24, 32, 40, 48};
    SN(hipMemcpyAsync(d_names, blob, bytes, hipMemcpyHostToDevice, st));
    SN(hipMemcpyAsync(d_off, off32, (n + 1) * 4, hipMemcpyHostToDevice, st));
    k_dp_name_keys<<<grid, 256, 0, st>>>(d_names, d_off, n, khi, klo, v0);
    SN(hipMemcpyAsync(k0, klo, n * 8, hipMemcpyDeviceToDevice, st));
    int w = gts_radix_sort<uint64_t>(k0, v0, k1, v1, n, shifts, 7, tmp, st);
    if (w < 0) rc = GTSG_ELIMIT;
    if (!rc) {
      uint32_t *va = w ? v1 : v0, *vb = w ? v0 : v1;
      uint64_t *ka = w ? k1 : k0, *kb = w ? k0 : k1;
      k_dp_gather_u64<<<grid, 256, 0, st>>>(khi, va, n, ka);
      w = gts_radix_sort<uint64_t>(ka, va, kb, vb, n, shifts, 7, tmp, st);
      if (w < 0) rc = GTSG_ELIMIT;
      if (!rc) {
        const uint32_t *vf = w ? vb : va;
        const uint64_t *kf = w ? kb : ka;
        k_dp_name_ties<<<grid, 256, 0, st>>>(kf, klo, vf, n, d_tie);
        SN(hipMemcpyAsync(perm, vf, n * 4, hipMemcpyDeviceToHost, st));
        SN(hipMemcpyAsync(tie, d_tie, n, hipMemcpyDeviceToHost, st));
      }
    }
    SN(hipStreamSynchronize(st));
    SN(hipGetLastError());
  }
#undef SN
  void *ptrs[] = {d_names, d_off, khi, klo, k0, k1, v0, v1, tmp, d_tie};
  for (void *q : ptrs) if (q) hipFree(q);
  if (st) hipStreamDestroy(st);
  free(off32);
  return rc;
}

/* FASTA record table: for every record in file order the offsets of its
   description (after the '>', up to its newline -- `len` if the file ends
   first) and the number of sequence characters (everything up to the next
   record start but newline, carriage return and blank).  The arrays are
   malloc'ed here; the caller frees them. */
int gtsg_fasta_records(int device, const char *text, uint64_t len, uint64_t *n_records, uint64_t **desc_start,
                       uint64_t **desc_end, uint64_t **seq_len)
{
  if (!n_records || !desc_start || !desc_end || !seq_len || (len && !text)) return GTSG_EINVAL;
  *n_records = 0; *desc_start = *desc_end = *seq_len = nullptr;
  if (len == 0) return 0;
  if (hipSetDevice(device) != hipSuccess) return GTSG_EHIP;
  const uint64_t nblocks = (len + FA_STRIDE - 1) / FA_STRIDE;
  if (nblocks >= 0x7FFFFFFFull) return GTSG_ELIMIT;
  char *d_text = nullptr;
  FaSum *sums = nullptr;
  uint8_t *carry = nullptr;
  uint64_t *base = nullptr, *d_n = nullptr, *hs = nullptr, *he = nullptr, *sl = nullptr;
  hipStream_t st = nullptr;
  int rc = 0;
  uint64_t nrec = 0;
#define FR(x) do { if (!rc && (x) != hipSuccess) rc = GTSG_EHIP; } while (0)
  FR(hipStreamCreate(&st));
  FR(hipMalloc((void **)&d_text, len + 16));
  FR(hipMalloc((void **)&sums, nblocks * sizeof(FaSum)));
  FR(hipMalloc((void **)&carry, nblocks)); FR(hipMalloc((void **)&base, nblocks * 8));
  FR(hipMalloc((void **)&d_n, 8));
  if (!rc) {
    FR(hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, st));
    k_fa_summary<<<(uint32_t)nblocks, FA_THREADS, 0, st>>>(d_text, len, sums);
    k_fa_chain<<<1, 1, 0, st>>>(sums, nblocks, carry, base, d_n);
    FR(hipMemcpyAsync(&nrec, d_n, 8, hipMemcpyDeviceToHost, st));
    FR(hipStreamSynchronize(st));
    FR(hipGetLastError());
  }
  if (!rc && nrec) {
    FR(hipMalloc((void **)&hs, nrec * 8)); FR(hipMalloc((void **)&he, nrec * 8)); FR(hipMalloc((void **)&sl, nrec * 8));
    if (!rc) {
      FR(hipMemsetAsync(sl, 0, nrec * 8, st));
      FR(hipMemsetAsync(he, 0xFF, nrec * 8, st));
      k_fa_records<<<(uint32_t)nblocks, FA_THREADS, 0, st>>>(d_text, len, carry, base, hs, he,
                                                            (unsigned long long *)sl);
      *desc_start = (uint64_t *)malloc(nrec * 8); *desc_end = (uint64_t *)malloc(nrec * 8);
      *seq_len = (uint64_t *)malloc(nrec * 8);
      if (!*desc_start || !*desc_end || !*seq_len) rc = GTSG_ENOMEM;
      if (!rc) {
        FR(hipMemcpyAsync(*desc_start, hs, nrec * 8, hipMemcpyDeviceToHost, st));
        FR(hipMemcpyAsync(*desc_end, he, nrec * 8, hipMemcpyDeviceToHost, st));
        FR(hipMemcpyAsync(*seq_len, sl, nrec * 8, hipMemcpyDeviceToHost, st));
      }
      FR(hipStreamSynchronize(st));
      FR(hipGetLastError());
      if (!rc)
        for (uint64_t i = 0; i < nrec; ++i)
          if ((*desc_end)[i] == ~0ull) (*desc_end)[i] = len;   /* the file ends inside the description */
    }
  }
#undef FR
  void *ptrs[] = {d_text, sums, carry, base, d_n, hs, he, sl};
  for (void *q : ptrs) if (q) hipFree(q);
  if (st) hipStreamDestroy(st);
  if (rc) { free(*desc_start); free(*desc_end); free(*seq_len); *desc_start = *desc_end = *seq_len = nullptr; return rc; }
  *n_records = nrec;
  return 0;
}

/* the records of the last parse copied to host arrays (tests, bindings) */
int gtsg_deparser_download(GtsgDeParser *p, uint32_t *root, uint32_t *ctg, int64_t *dist, float *std_dev,
                           int64_t *num_pairs, uint8_t *flags)
{
  if (!p) return GTSG_EINVAL;
  /* the records gtsg_deparser_records hands out: all pieces in accumulate mode,
     else those of the last parse */
  const bool acc = p->accumulate;
  const uint64_t n = acc ? p->a_n : p->n_records;
  if (!n) return 0;
  DPCHK(hipSetDevice(p->device));
  const bool c = p->compacted;
  if (root) DPCHK(hipMemcpy(root, acc ? p->a_root : c ? p->root2 : p->root, n * 4, hipMemcpyDeviceToHost));
  if (ctg) DPCHK(hipMemcpy(ctg, acc ? p->a_ctg : c ? p->ctg2 : p->ctg, n * 4, hipMemcpyDeviceToHost));
  if (dist) DPCHK(hipMemcpy(dist, acc ? p->a_dist : c ? p->dist2 : p->dist, n * 8, hipMemcpyDeviceToHost));
  if (std_dev) DPCHK(hipMemcpy(std_dev, acc ? p->a_sd : c ? p->sd2 : p->sd, n * 4, hipMemcpyDeviceToHost));
  if (num_pairs) DPCHK(hipMemcpy(num_pairs, acc ? p->a_np : c ? p->np2 : p->np, n * 8, hipMemcpyDeviceToHost));
  if (flags) DPCHK(hipMemcpy(flags, acc ? p->a_flags : c ? p->flags2 : p->flags, n, hipMemcpyDeviceToHost));
  return 0;
}

}  /* extern "C" */
