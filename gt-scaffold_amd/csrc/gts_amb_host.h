/*
  gts_amb_host.h -- host-side derivation of the ambiguous-order thresholds.

  ref src/gt_scaffolder_algorithms.c:187-192:
      prob12 = 0.5 * (1 + erf(interval));  prob21 = 1.0 - prob12;
      p_wrong = 1.0 - MAX(prob12, prob21);  return p_wrong > cutoff;
  with float variables and the host libm's double erf.  For interval >= 0
  (resp. < 0) p_wrong is non-increasing in |interval|, so the set of floats on
  which the test holds is a prefix [0, t] of the non-negative floats; t is
  found by bisection over float bit patterns with the SAME libm call the
  reference makes on this host.  Runs once per filter call (~64 erf calls).
*/
#ifndef GTS_AMB_HOST_H
#define GTS_AMB_HOST_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "gts_defs.h"

static inline bool gts_amb_pipeline(float interval, float cutoff)
{
  float prob12, prob21, p_wrong;
  /* the reference is C: erf() takes the double overload.  Spell the
     promotion out, this header is compiled as C++ (erf(float) would bind to
     erff there). */
  prob12 = (float)(0.5 * (1.0 + erf((double)interval)));
  prob21 = (float)(1.0 - (double)prob12);
  p_wrong = (float)(1.0 - (double)(prob12 > prob21 ? prob12 : prob21));
  return p_wrong > cutoff;
}

static inline float gts_amb_bisect(float cutoff, bool negative)
{
  /* non-negative float bit patterns 0 .. 0x7F800000 (+inf) are ordered */
  uint32_t lo = 0, hi = 0x7F800000u;
  float x;
  if (!gts_amb_pipeline(negative ? -0.0f : 0.0f, cutoff)) return -1.0f;
  while (lo < hi) {       /* largest pattern for which the test holds */
    uint32_t mid = lo + (hi - lo + 1) / 2;
    memcpy(&x, &mid, 4);
    if (gts_amb_pipeline(negative ? -x : x, cutoff)) lo = mid; else hi = mid - 1;
  }
  memcpy(&x, &lo, 4);
  return x;
}

static inline GtsAmbThresholds gts_amb_thresholds(float pcutoff)
{
  GtsAmbThresholds t;
  t.tpos = gts_amb_bisect(pcutoff, false);
  t.tneg = gts_amb_bisect(pcutoff, true);
  return t;
}

#endif
