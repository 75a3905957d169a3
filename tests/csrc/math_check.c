/* Test shim: compiles envutil_amd/csrc/eu_math.h for the host and counts
 * disagreements with the live libm (the library the reference's portable
 * back-end calls). gcc -O2 -ffp-contract=off -fopenmp -shared. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../../envutil_amd/csrc/eu_math.h"

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static int same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

/* every float bit pattern in [first, last] */
long check_atanf_range(uint32_t first, uint32_t last, uint32_t *first_bad)
{
  long bad = 0;
  uint32_t fb = 0;
#pragma omp parallel for reduction(+:bad) schedule(static)
  for (long long u = first; u <= (long long)last; u++) {
    float x;
    uint32_t uu = (uint32_t)u;
    memcpy(&x, &uu, 4);
    if (!same(eu_atanf(x), atanf(x))) {
      bad++;
#pragma omp critical
      if (!fb) fb = uu;
    }
  }
  if (first_bad) *first_bad = fb;
  return bad;
}

static inline uint64_t mix(uint64_t *s)
{
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

/* n pseudo-random pairs; mode 0: arbitrary bit patterns, mode 1: unit-scale
 * rays (the values the render path feeds it), mode 2: special values grid */
long check_atan2f_random(long n, uint64_t seed, int mode, float *bad_y, float *bad_x)
{
  long bad = 0;
#pragma omp parallel reduction(+:bad)
  {
    uint64_t s = seed;
#ifdef _OPENMP
    extern int omp_get_thread_num(void);
    s += 0x1234567ull * (uint64_t)omp_get_thread_num();
#endif
#pragma omp for schedule(static)
    for (long i = 0; i < n; i++) {
      uint64_t r = mix(&s);
      float y, x;
      if (mode == 0) {
        uint32_t a = (uint32_t)r, b = (uint32_t)(r >> 32);
        memcpy(&y, &a, 4); memcpy(&x, &b, 4);
      } else {
        y = (float)((double)(uint32_t)r / 2147483648.0 - 1.0);
        x = (float)((double)(uint32_t)(r >> 32) / 2147483648.0 - 1.0);
        if (mode == 2) { y *= 1e-3f; }
      }
      if (!same(eu_atan2f(y, x), atan2f(y, x))) {
        bad++;
#pragma omp critical
        { if (bad_y) *bad_y = y; if (bad_x) *bad_x = x; }
      }
    }
  }
  return bad;
}

long check_atan2f_pairs(const float *y, const float *x, long n)
{
  long bad = 0;
  for (long i = 0; i < n; i++)
    if (!same(eu_atan2f(y[i], x[i]), atan2f(y[i], x[i]))) bad++;
  return bad;
}

#ifdef EU_HAVE_SINCOSF
/* every float bit pattern in [first, last]: which = 0 sinf, 1 cosf */
long check_sincosf_range(uint32_t first, uint32_t last, int which, uint32_t *first_bad)
{
  long bad = 0;
  uint32_t fb = 0;
#pragma omp parallel for reduction(+:bad) schedule(static)
  for (long long u = first; u <= (long long)last; u++) {
    float x;
    uint32_t uu = (uint32_t)u;
    memcpy(&x, &uu, 4);
    float a = which ? eu_cosf(x) : eu_sinf(x), b = which ? cosf(x) : sinf(x);
    if (!same(a, b)) {
      bad++;
#pragma omp critical
      if (!fb) fb = uu;
    }
  }
  if (first_bad) *first_bad = fb;
  return bad;
}
/* eu_sincosf_120 (one reduction, both polynomials, no branch) against the live libm's sinf and cosf:
 * every float with |y| < 120, both signs */
long check_sincosf120(uint32_t *first_bad)
{
  long bad = 0;
  uint32_t fb = 0;
  const float lim = 120.0f;
  uint32_t last;
  memcpy(&last, &lim, 4);
#pragma omp parallel for reduction(+:bad) schedule(static)
  for (long long u = 0; u < (long long)last; u++) {
    for (int sg = 0; sg < 2; sg++) {
      float x, sn, cs;
      uint32_t uu = (uint32_t)u | (sg ? 0x80000000u : 0u);
      memcpy(&x, &uu, 4);
      eu_sincosf_120(x, &sn, &cs);
      if (!same(sn, sinf(x)) || !same(cs, cosf(x))) {
        bad++;
#pragma omp critical
        if (!fb) fb = uu;
      }
    }
  }
  if (first_bad) *first_bad = fb;
  return bad;
}
int have_sincosf(void) { return 1; }
#else
int have_sincosf(void) { return 0; }
#endif

/* stereographic stepper angle (stepper.h:1146) for every float norm in
 * [first, last] (bit patterns of non-negative floats) against the live libm */
long check_ster_angle_range(uint32_t first, uint32_t last, uint32_t *first_bad)
{
  long bad = 0;
  uint32_t fb = 0;
#pragma omp parallel for reduction(+:bad) schedule(static)
  for (long long u = first; u <= (long long)last; u++) {
    float x;
    uint32_t uu = (uint32_t)u;
    memcpy(&x, &uu, 4);
    float ref = (float)(M_PI_2 - 2.0 * atan((double)x / 2.0));
    if (!same(eu_ster_angle(x), ref)) {
      bad++;
#pragma omp critical
      if (!fb) fb = uu;
    }
  }
  if (first_bad) *first_bad = fb;
  return bad;
}

/* eu_tanf against libm's tanf for every float with |x| <= the float whose bits are `last` (both signs) */
long check_tanf_range(uint32_t last, uint32_t *first_bad)
{
  long bad = 0;
  uint32_t fb = 0;
#pragma omp parallel for reduction(+:bad) schedule(static)
  for (long long u = 0; u <= (long long)last; u++)
    for (int sg = 0; sg < 2; sg++) {
      float x;
      uint32_t uu = (uint32_t)u | (sg ? 0x80000000u : 0u);
      memcpy(&x, &uu, 4);
      if (!same(eu_tanf(x), tanf(x))) {
        bad++;
#pragma omp critical
        if (!fb) fb = uu;
      }
    }
  if (first_bad) *first_bad = fb;
  return bad;
}
