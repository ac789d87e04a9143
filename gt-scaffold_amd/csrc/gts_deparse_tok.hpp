/*
  gts_deparse_tok.hpp -- the token level of the DistEst / A-statistic parser:
  what sscanf("%[^>,],%ld,%ld,%f") and strtol / strtof make of one token, for
  the tokens of the regular form (gts_deparse.hip).  Compiled for the device by
  gts_deparse.hip and for the host by tests/hostsim/deparse_fuzz.cpp, which
  compares it with the libc calls on random tokens.
*/
#ifndef GTS_DEPARSE_TOK_HPP
#define GTS_DEPARSE_TOK_HPP
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define DP_HD __device__ __forceinline__
#define DP_DOUBLE_BITS(d) ((uint64_t)__double_as_longlong(d))
#define DP_FMA(a, b, c) __fma_rn(a, b, c)
#define DP_HD_MEMBER __device__ __forceinline__
#else
#define DP_HD_MEMBER inline
#include <math.h>
#define DP_HD static inline
static inline uint64_t dp_double_bits_host(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
#define DP_DOUBLE_BITS(d) dp_double_bits_host(d)
#define DP_FMA(a, b, c) fma(a, b, c)
#endif

/* The staged text, read eight bytes at a time: a byte load from LDS costs a
   round trip of ~100 cycles and the loops below walk the text byte by byte;
   the cursor keeps the aligned 8-byte word of the last access in a register
   pair (the walks go forward, so seven of eight accesses hit it). */
struct DpText {
  const uint8_t *base;    /* 8-byte aligned start of the LDS buffer */
  uint32_t shift;         /* offset of text position 0 in it */
  uint32_t blk;
  uint64_t w;
  DP_HD_MEMBER DpText(const uint8_t *b, uint32_t sh) : base(b), shift(sh), blk(0xFFFFFFFFu), w(0) {}
  DP_HD_MEMBER uint8_t operator[](uint32_t i)
  {
    const uint32_t j = i + shift, b = j >> 3;
    if (b != blk) { blk = b; w = *(const uint64_t *)(base + ((size_t)b << 3)); }
    return (uint8_t)(w >> ((j & 7u) * 8u));
  }
};

/* ---- tokens --------------------------------------------------------------- */
enum { DP_TOK_REC = 0, DP_TOK_FAIL = 1, DP_TOK_SEMI = 2, DP_TOK_IRREGULAR = 3 };

struct DpRecord {
  uint32_t h0, h1;    /* header without its last character: buf[h0, h1) */
  uint8_t last;       /* the last character of the header (the sign) */
  int64_t dist, np;
  float sd;
};

DP_HD bool dp_digit(uint8_t c) { return c >= '0' && c <= '9'; }

/* [+-]?digits, at most 18 digits.  0 ok, 1 sscanf would fail, 2 out of the regular form */
DP_HD int dp_int(DpText &buf, uint32_t &i, uint32_t e, int64_t &out)
{
  bool neg = false;
  if (i < e && (buf[i] == '-' || buf[i] == '+')) { neg = buf[i] == '-'; ++i; }
  if (i >= e || !dp_digit(buf[i])) return 1;
  uint64_t v = 0;
  uint32_t nd = 0;
  while (i < e && dp_digit(buf[i])) { v = v * 10u + (uint64_t)(buf[i] - '0'); ++i; ++nd; }
  if (nd > 18) return 2;
  out = neg ? -(int64_t)v : (int64_t)v;
  return 0;
}

/* %f on buf[i, e): sign, digits [. digits] | . digits, optional exponent, then
   the end of the token.  0 ok, 1 sscanf would fail, 2 out of the regular form
   (inf, nan, hex, more than 19 digits, trailing text, a value the double in
   between cannot round for certain). */
DP_HD int dp_float(DpText &buf, uint32_t &i, uint32_t e, float &out)
{
  bool neg = false;
  if (i < e && (buf[i] == '-' || buf[i] == '+')) { neg = buf[i] == '-'; ++i; }
  if (i >= e) return 1;
  if (!dp_digit(buf[i]) && buf[i] != '.') return 2;   /* inf, nan, ... or a failure */
  uint64_t m = 0;
  uint32_t sig = 0, frac = 0, nd = 0;
  while (i < e && dp_digit(buf[i])) {
    if (m || buf[i] != '0') ++sig;
    m = m * 10u + (uint64_t)(buf[i] - '0'); ++i; ++nd;
    if (sig > 19) return 2;
  }
  if (i < e && buf[i] == '.') {
    ++i;
    while (i < e && dp_digit(buf[i])) {
      if (m || buf[i] != '0') ++sig;
      m = m * 10u + (uint64_t)(buf[i] - '0'); ++i; ++nd; ++frac;
      if (sig > 19 || frac > 40) return 2;
    }
  }
  if (nd == 0) return 2;            /* "." alone: a failure; rare enough for the host */
  int32_t e10 = -(int32_t)frac;
  if (i < e && (buf[i] == 'e' || buf[i] == 'E')) {
    ++i;
    bool eneg = false;
    if (i < e && (buf[i] == '-' || buf[i] == '+')) { eneg = buf[i] == '-'; ++i; }
    uint32_t ex = 0, ed = 0;
    while (i < e && dp_digit(buf[i])) { ex = ex * 10u + (uint32_t)(buf[i] - '0'); ++i; if (++ed > 3) return 2; }
    if (ed == 0) return 2;          /* "1.5e": the 'e' is not part of the number */
    e10 += eneg ? -(int32_t)ex : (int32_t)ex;
  }
  if (i != e) return 2;             /* hex, trailing text */
  if (e10 < -22 || e10 > 22) return 2;
  /* m < 10^19 < 2^64; 10^|e10| (<= 10^22) is a double.  (double)m and the
     quotient / product round once each: d is within 2^-52 of the decimal,
     relatively (about one unit of its last place).  strtof rounds the decimal
     to 24 bits; (float)d rounds d.  The two agree unless a float midpoint lies
     between the decimal and d, i.e. unless d is within that distance of a
     midpoint: the low 29 bits of d's fraction are then within a few units of
     2^28.  Four units of margin; such a value goes to the host. */
  double p10 = 1.0;
  for (int32_t k = 0; k < (e10 < 0 ? -e10 : e10); ++k) p10 *= 10.0;
  const double md = (double)m;
  const double d = e10 < 0 ? md / p10 : md * p10;
  const uint64_t bits = DP_DOUBLE_BITS(d);
  const int64_t low = (int64_t)(bits & 0x1FFFFFFFull) - 0x10000000ll;
  /* (an exact d -- 99936252, 8388608.5 -- may be a midpoint: both roundings
     then break the tie to even) */
  const bool exact = m < (1ull << 53) &&
                     (e10 < 0 ? DP_FMA(d, p10, -md) == 0.0 : DP_FMA(md, p10, -d) == 0.0);
  if (!exact && low >= -4 && low <= 4) return 2;
  if (d != 0.0 && (d < 1.2e-38 || d > 3.4e38)) return 2;   /* float subnormals round on another grid; overflow */
  out = neg ? -(float)d : (float)d;
  return 0;
}

/* the token buf[s, e) as sscanf("%1023[^>,],%ld,%ld,%f") sees it */
DP_HD int dp_token(DpText &buf, uint32_t s, uint32_t e, DpRecord &r)
{
  if (e - s == 1 && buf[s] == ';') return DP_TOK_SEMI;
  const bool semi = buf[s] == ';';
  /* a token that starts with ';' and is longer: a record named ";..." if it
     scans, a separator if not -- left to the host */
  if (semi) return DP_TOK_IRREGULAR;
  uint32_t i = s;
  while (i < e && buf[i] != ',' && buf[i] != '>') ++i;
  if (i == s) return DP_TOK_FAIL;                 /* %[ matched nothing */
  if (i < e && buf[i] == '>') return DP_TOK_IRREGULAR;
  if (i >= e) return DP_TOK_FAIL;                 /* no ',' after the header */
  r.h0 = s; r.h1 = i - 1; r.last = buf[i - 1];
  ++i;
  int rc = dp_int(buf, i, e, r.dist);
  if (rc) return rc == 1 ? DP_TOK_FAIL : DP_TOK_IRREGULAR;
  if (i >= e || buf[i] != ',') return DP_TOK_FAIL;
  ++i;
  rc = dp_int(buf, i, e, r.np);
  if (rc) return rc == 1 ? DP_TOK_FAIL : DP_TOK_IRREGULAR;
  if (i >= e || buf[i] != ',') return DP_TOK_FAIL;
  ++i;
  rc = dp_float(buf, i, e, r.sd);
  if (rc) return rc == 1 ? DP_TOK_FAIL : DP_TOK_IRREGULAR;
  return DP_TOK_REC;
}

#endif
