/*
  deparse_fuzz.cpp -- test infrastructure: the token level of the GPU DistEst
  parser (gt-scaffold_amd/csrc/gts_deparse_tok.hpp, the same source the device
  compiles) against the libc calls the reference makes
  (sscanf("%[^>,],%ld,%ld,%f"), ref gt_scaffolder_parser.c:212, :340, and
  sscanf("%f") for the A-statistic fields, ref algorithms.c:126).

    deparse_fuzz <seed> <count>     prints "<checked> <records> <fails> <irregular> <mismatches>"

  A token the parser calls a record must scan with exactly its values (floats
  bit for bit), a token it calls a failure must not scan, a token it hands to
  the host (irregular) may be anything.
*/
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gts_deparse_tok.hpp"

static uint64_t rng_state;
static uint64_t rnd()
{
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}
static uint32_t below(uint32_t n) { return (uint32_t)(rnd() % n); }

static void digits(char *&p, uint32_t n) { for (uint32_t i = 0; i < n; ++i) *p++ = (char)('0' + below(10)); }

/* a number the way the fields are usually written, and less usual ways */
static void number(char *&p, bool integer)
{
  const uint32_t kind = below(100);
  if (below(4) == 0) *p++ = below(2) ? '-' : '+';
  if (integer) {
    if (kind < 80) digits(p, 1 + below(6));
    else if (kind < 90) digits(p, 1 + below(19));
    else if (kind < 95) { digits(p, below(3)); *p++ = "x.e-"[below(4)]; digits(p, below(3)); }
    else { /* empty */ }
    return;
  }
  if (kind < 50) { digits(p, 1 + below(4)); *p++ = '.'; digits(p, 1 + below(2)); }
  else if (kind < 65) { digits(p, below(8)); *p++ = '.'; digits(p, below(12)); }
  else if (kind < 75) { digits(p, 1 + below(19)); }
  else if (kind < 85) { digits(p, 1 + below(3)); if (below(2)) { *p++ = '.'; digits(p, below(9)); }
                        *p++ = below(2) ? 'e' : 'E'; if (below(2)) *p++ = below(2) ? '-' : '+'; digits(p, below(3)); }
  else if (kind < 90) { /* near float midpoints: 2^24 + 1 + tiny, x.5 of large integers */
    const uint64_t base = 16777216ull + 2ull * below(1000000) + 1ull;
    p += sprintf(p, "%llu", (unsigned long long)base);
    if (below(2)) { *p++ = '.'; digits(p, below(3)); if (below(2)) { memset(p, '0', 10); p += below(10); *p++ = '1'; } } }
  else if (kind < 94) { const char *w[] = {"inf", "nan", "0x1p3", "INF", "infinity", ".", "e5", "1e", "--1"};
                        p += sprintf(p, "%s", w[below(9)]); }
  else if (kind < 97) { digits(p, 1 + below(3)); *p++ = "xyz,>;"[below(6)]; digits(p, below(3)); }
  else { /* empty */ }
}

static uint32_t make_token(char *tok)
{
  char *p = tok;
  const uint32_t kind = below(100);
  if (kind < 2) { *p++ = ';'; if (below(3) == 0) { *p++ = 'a'; } *p = 0; return (uint32_t)(p - tok); }
  const uint32_t hl = kind < 4 ? 0 : 1 + below(12);
  for (uint32_t i = 0; i < hl; ++i) {
    const uint32_t c = below(200);
    *p++ = c < 120 ? (char)('a' + c % 26) : c < 180 ? (char)('0' + c % 10) : "_-+.:;>"[c % 7];
  }
  if (below(20) != 0) *p++ = "+-+-+-x"[below(7)];
  if (below(25) != 0) *p++ = ',';
  number(p, true);
  if (below(25) != 0) *p++ = ',';
  number(p, true);
  if (below(25) != 0) *p++ = ',';
  number(p, false);
  *p = 0;
  return (uint32_t)(p - tok);
}

int main(int argc, char **argv)
{
  const uint64_t seed = argc > 1 ? strtoull(argv[1], NULL, 10) : 1;
  const uint64_t count = argc > 2 ? strtoull(argv[2], NULL, 10) : 1000000;
  rng_state = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  static char tok[512] __attribute__((aligned(16)));
  uint64_t nrec = 0, nfail = 0, nirr = 0, bad = 0, checked = 0;
  for (uint64_t it = 0; it < count; ++it) {
    uint32_t len = make_token(tok + 16);
    char *t = tok + 16;
    /* blanks and control characters never reach a token (the parser splits on
       blanks and hands files with control characters to the host) */
    bool skip = len == 0;
    for (uint32_t i = 0; i < len; ++i) if ((unsigned char)t[i] <= ' ' || t[i] == 0x7F) skip = true;
    if (skip) continue;
    ++checked;
    DpText txt((const uint8_t *)tok, 16);
    DpRecord r;
    memset(&r, 0, sizeof r);
    const int kind = dp_token(txt, 0, len, r);
    char hdr[1024];
    long d = 0, np = 0;
    float sd = 0;
    const int got = sscanf(t, "%1023[^>,],%ld,%ld,%f", hdr, &d, &np, &sd);
    if (kind == DP_TOK_IRREGULAR) { ++nirr; continue; }
    if (kind == DP_TOK_SEMI) {
      if (!(len == 1 && t[0] == ';')) { ++bad; fprintf(stderr, "SEMI for '%s'\n", t); }
      continue;
    }
    if (kind == DP_TOK_FAIL) {
      ++nfail;
      if (got == 4) { ++bad; fprintf(stderr, "FAIL but sscanf scans '%s'\n", t); }
      continue;
    }
    ++nrec;
    const size_t hl = got >= 1 ? strlen(hdr) : 0;
    uint32_t a, b;
    memcpy(&a, &sd, 4); memcpy(&b, &r.sd, 4);
    if (got != 4 || d != (long)r.dist || np != (long)r.np || a != b || hl != (size_t)(r.h1 - r.h0) + 1 ||
        memcmp(hdr, t + r.h0, hl - 1) != 0 || (uint8_t)hdr[hl - 1] != r.last) {
      ++bad;
      fprintf(stderr, "REC mismatch '%s': sscanf %d (%ld %ld %a) parser (%lld %lld %a)\n", t, got, d, np, sd,
              (long long)r.dist, (long long)r.np, r.sd);
    }
    /* the A-statistic fields are scanned with a bare %f */
  }
  /* bare floats (A-statistic fields) */
  for (uint64_t it = 0; it < count; ++it) {
    char *t = tok + 16, *p = t;
    number(p, false);
    *p = 0;
    const uint32_t len = (uint32_t)(p - t);
    bool skip = len == 0;
    for (uint32_t i = 0; i < len; ++i) if ((unsigned char)t[i] <= ' ') skip = true;
    if (skip) continue;
    ++checked;
    DpText txt((const uint8_t *)tok, 16);
    uint32_t i = 0;
    float v = 0;
    const int rc = dp_float(txt, i, len, v);
    float sv = 0;
    const int got = sscanf(t, "%f", &sv);
    if (rc == 2) { ++nirr; continue; }
    if (rc == 1) { ++nfail; if (got == 1) { ++bad; fprintf(stderr, "float FAIL but scans '%s'\n", t); } continue; }
    ++nrec;
    uint32_t a, b;
    memcpy(&a, &sv, 4); memcpy(&b, &v, 4);
    if (got != 1 || a != b) { ++bad; fprintf(stderr, "float mismatch '%s': %a vs %a\n", t, sv, v); }
  }
  printf("%llu %llu %llu %llu %llu\n", (unsigned long long)checked, (unsigned long long)nrec,
         (unsigned long long)nfail, (unsigned long long)nirr, (unsigned long long)bad);
  return bad ? 1 : 0;
}
