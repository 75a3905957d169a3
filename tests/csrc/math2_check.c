/* Test shim (clang): host compilation of envutil_amd/csrc/eu_math2.h - the
 * two-lane atanf/atan2f used by the packed render kernel - against the live
 * libm. The safe division/sqrt helpers compile to `/` and sqrtf here; their
 * device sequences are checked on the GPU (tests/test_gpu_math.py). */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../../envutil_amd/csrc/eu_math2.h"

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static int same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

/* all non-negative floats t (incl. +inf) in [first,last], two per call */
long check_atanf_pos2_range(uint32_t first, uint32_t last)
{
  long bad = 0;
#pragma omp parallel for reduction(+:bad) schedule(static)
  for (long long u = first; u <= (long long)last; u += 2) {
    uint32_t a = (uint32_t)u, b = (uint32_t)(u + 1 <= last ? u + 1 : u);
    eu_f2 t;
    float ta, tb;
    memcpy(&ta, &a, 4); memcpy(&tb, &b, 4);
    t.x = ta; t.y = tb;
    eu_f2 r = eu_atanf_pos2(t);
    if (!same(r.x, atanf(ta))) bad++;
    if (!same(r.y, atanf(tb))) bad++;
  }
  return bad;
}

static inline uint64_t mix(uint64_t *s)
{
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

long check_atan2f_2_random(long n, uint64_t seed, int mode)
{
  long bad = 0;
#pragma omp parallel reduction(+:bad)
  {
    uint64_t s = seed;
#ifdef _OPENMP
    extern int omp_get_thread_num(void);
    s += 0x7654321ull * (uint64_t)omp_get_thread_num();
#endif
#pragma omp for schedule(static)
    for (long i = 0; i < n; i++) {
      uint64_t r0 = mix(&s), r1 = mix(&s);
      eu_f2 y, x;
      float v[4];
      if (mode == 0) {       /* arbitrary bit patterns: exercises the slow path too */
        uint32_t w[4] = { (uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32) };
        memcpy(v, w, 16);
      } else {               /* ray-like magnitudes, all quadrants */
        v[0] = (float)((double)(uint32_t)r0 / 2147483648.0 - 1.0);
        v[1] = (float)((double)(uint32_t)(r0 >> 32) / 2147483648.0 - 1.0);
        v[2] = (float)((double)(uint32_t)r1 / 2147483648.0 - 1.0) * (mode == 2 ? 1e-4f : 6.0f);
        v[3] = (float)((double)(uint32_t)(r1 >> 32) / 2147483648.0 - 1.0);
      }
      y.x = v[0]; x.x = v[1]; y.y = v[2]; x.y = v[3];
      eu_f2 r = eu_atan2f_2(y, x);
      if (!same(r.x, atan2f(v[0], v[1]))) bad++;
      if (!same(r.y, atan2f(v[2], v[3]))) bad++;
      /* the x > 0 form */
      eu_f2 xp = { fabsf(v[1]), fabsf(v[3]) };
      if (xp.x > 0.0f && xp.y > 0.0f && xp.x == xp.x && xp.y == xp.y) {
        r = eu_atan2f_2_xpos(y, xp);
        if (!same(r.x, atan2f(v[0], xp.x))) bad++;
        if (!same(r.y, atan2f(v[2], xp.y))) bad++;
      }
    }
  }
  return bad;
}

long check_atan2f_2_pairs(const float *y, const float *x, long n)
{
  long bad = 0;
  for (long i = 0; i + 1 < n; i += 2) {
    eu_f2 yy = { y[i], y[i + 1] }, xx = { x[i], x[i + 1] };
    eu_f2 r = eu_atan2f_2(yy, xx);
    if (!same(r.x, atan2f(y[i], x[i]))) bad++;
    if (!same(r.y, atan2f(y[i + 1], x[i + 1]))) bad++;
  }
  return bad;
}

/* table-driven variants (the ones the render kernel uses) */
static float g_tab[EU_ATAN_TAB_FLOATS];
static void fill_tab(void) { for (int i = 0; i < EU_ATAN_TAB_ENTRIES; i++) eu_atan_tab_entry(i, g_tab + 8 * i); }

long check_atanf_pos2_tab_range(uint32_t first, uint32_t last)
{
  long bad = 0;
  fill_tab();
#pragma omp parallel for reduction(+:bad) schedule(static)
  for (long long u = first; u <= (long long)last; u += 2) {
    uint32_t a = (uint32_t)u, b = (uint32_t)(u + 1 <= last ? u + 1 : u);
    float ta, tb;
    memcpy(&ta, &a, 4); memcpy(&tb, &b, 4);
    eu_f2 t = { ta, tb };
    eu_f2 r = eu_atanf_pos2_tab(t, g_tab);
    if (!same(r.x, atanf(ta))) bad++;
    if (!same(r.y, atanf(tb))) bad++;
  }
  return bad;
}

long check_atan2f_2_tab_random(long n, uint64_t seed, int mode)
{
  long bad = 0;
  fill_tab();
#pragma omp parallel reduction(+:bad)
  {
    uint64_t s = seed;
#ifdef _OPENMP
    extern int omp_get_thread_num(void);
    s += 0x7654321ull * (uint64_t)omp_get_thread_num();
#endif
#pragma omp for schedule(static)
    for (long i = 0; i < n; i++) {
      uint64_t r0 = mix(&s), r1 = mix(&s);
      float v[4];
      if (mode == 0) {
        uint32_t w[4] = { (uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32) };
        memcpy(v, w, 16);
      } else {
        v[0] = (float)((double)(uint32_t)r0 / 2147483648.0 - 1.0);
        v[1] = (float)((double)(uint32_t)(r0 >> 32) / 2147483648.0 - 1.0);
        v[2] = (float)((double)(uint32_t)r1 / 2147483648.0 - 1.0) * (mode == 2 ? 1e-4f : 6.0f);
        v[3] = (float)((double)(uint32_t)(r1 >> 32) / 2147483648.0 - 1.0);
      }
      eu_f2 y = { v[0], v[2] }, x = { v[1], v[3] };
      eu_f2 r = eu_atan2f_2_tab(y, x, g_tab, 0);
      if (!same(r.x, atan2f(v[0], v[1]))) bad++;
      if (!same(r.y, atan2f(v[2], v[3]))) bad++;
      eu_f2 xp = { fabsf(v[1]), fabsf(v[3]) };
      if (xp.x > 0.0f && xp.y > 0.0f && xp.x == xp.x && xp.y == xp.y) {
        r = eu_atan2f_2_tab(y, xp, g_tab, 1);
        if (!same(r.x, atan2f(v[0], xp.x))) bad++;
        if (!same(r.y, atan2f(v[2], xp.y))) bad++;
      }
    }
  }
  return bad;
}
