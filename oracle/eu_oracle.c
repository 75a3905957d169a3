/* TEST INFRASTRUCTURE - not product code. See eu_oracle.h for scope and the
 * pinning status of every stage. Each function names the reference lines it
 * restates (paths relative to /root/reference).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fopenmp -fPIC -shared
 * (no FMA contraction, no -ffast-math: every float operation below is one
 * IEEE operation in the order the reference's goading back-end performs it).
 */
#define _GNU_SOURCE
#include "eu_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <float.h>

/* ------------------------------------------------------------------------ */
/* set-up: extents (envutil_basic.cc:49-229)                                 */
/* ------------------------------------------------------------------------ */

double euo_get_vfov(int projection, int width, int height, double hfov)
{
  double vfov = 0.0;
  switch (projection) {
    case EUO_RECTILINEAR:
      vfov = 2.0 * atan(height * tan(hfov / 2.0) / width);
      break;
    case EUO_CYLINDRICAL: {
      double pixels_per_rad = width / hfov;
      double h_rad = height / pixels_per_rad;
      vfov = 2.0 * atan(h_rad / 2.0);
      break;
    }
    case EUO_STEREOGRAPHIC: {
      double w_rad = 2.0 * tan(hfov / 4.0);
      double pixels_per_rad = width / w_rad;
      double h_rad = height / pixels_per_rad;
      vfov = 4.0 * atan(h_rad / 2.0);
      break;
    }
    case EUO_SPHERICAL:
    case EUO_FISHEYE:
      vfov = hfov * height / width;
      break;
    default:
      /* envutil_basic.cc:91-101: the cubemap case falls through to this */
      vfov = hfov;
      break;
  }
  return vfov;
}

double euo_get_step(int projection, int width, int height, double hfov)
{
  (void)height;
  switch (projection) {
    case EUO_RECTILINEAR:
    case EUO_CUBEMAP:
      return atan(2.0 * tan(hfov / 2.0) / width);
    case EUO_BIATAN6:
    case EUO_SPHERICAL:
    case EUO_CYLINDRICAL:
    case EUO_FISHEYE:
      return hfov / width;
    case EUO_STEREOGRAPHIC:
      return atan(4.0 * tan(hfov / 4.0) / width);
  }
  return 0.0;
}

void euo_get_extent(int projection, int width, int height, double hfov,
                    double *e)
{
  double x0, x1, y0, y1;
  double alpha_x = -hfov / 2.0;
  double beta_x = hfov / 2.0;
  double beta_y = euo_get_vfov(projection, width, height, hfov) / 2.0;
  double alpha_y = -beta_y;
  switch (projection) {
    case EUO_SPHERICAL:
    case EUO_FISHEYE:
      x0 = alpha_x; x1 = beta_x; y0 = alpha_y; y1 = beta_y;
      break;
    case EUO_CYLINDRICAL:
      x0 = alpha_x; x1 = beta_x; y0 = tan(alpha_y); y1 = tan(beta_y);
      break;
    case EUO_RECTILINEAR:
      x0 = tan(alpha_x); x1 = tan(beta_x);
      y0 = tan(alpha_y); y1 = tan(beta_y);
      break;
    case EUO_STEREOGRAPHIC:
      x0 = 2.0 * tan(alpha_x / 2.0); x1 = 2.0 * tan(beta_x / 2.0);
      y0 = 2.0 * tan(alpha_y / 2.0); y1 = 2.0 * tan(beta_y / 2.0);
      break;
    case EUO_CUBEMAP:
    case EUO_BIATAN6:
      x0 = tan(alpha_x); x1 = tan(beta_x);
      y0 = 6 * x0; y1 = 6 * x1;
      break;
    default:
      x0 = x1 = y0 = y1 = 0.0;
  }
  e[0] = x0; e[1] = x1; e[2] = y0; e[3] = y1;
}

/* ------------------------------------------------------------------------ */
/* set-up: rotations (envutil_payload.cc:136-218, geometry.h:74-97)          */
/* Imath is not under /root/reference: Euler<float>(roll,pitch,yaw,ZXY)      */
/* .toQuat(), Quat::invert and Vec3*Quat follow Imath 3's published          */
/* ImathEuler.h / ImathQuat.h (static frame, even parity, initial axis Z,    */
/* no repetition: i=2, j=0, k=1). PARITY UNPINNED.                           */
/* ------------------------------------------------------------------------ */

static void cross3(const double *a, const double *b, double *o)
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

void euo_make_r3(double roll, double pitch, double yaw, int inverse, double *m)
{
  /* Eulerf: the three angles narrow to float (payload.cc:152) */
  float ax = (float)roll, ay = (float)pitch, az = (float)yaw;
  float ti = (float)(ax * 0.5), tj = (float)(ay * 0.5), th = (float)(az * 0.5);
  float ci = cosf(ti), cj = cosf(tj), ch = cosf(th);
  float si = sinf(ti), sj = sinf(tj), sh = sinf(th);
  float cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
  float a[3];
  a[2] = cj * sc - sj * cs;              /* i = 2 */
  a[0] = (float)((cj * ss + sj * cc) * 1.0); /* j = 0, parity +1 */
  a[1] = cj * cs - sj * sc;              /* k = 1 */
  float qr = cj * cc + sj * ss;
  double r = qr, v[3] = { a[0], a[1], a[2] };
  if (inverse) {
    double qdot = r * r + (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    r /= qdot;
    v[0] = -v[0] / qdot; v[1] = -v[1] / qdot; v[2] = -v[2] / qdot;
  }
  for (int e = 0; e < 3; e++) {
    double in[3] = { 0.0, 0.0, 0.0 }, A[3], B[3];
    in[e] = 1.0;
    cross3(v, in, A);
    cross3(v, A, B);
    for (int i = 0; i < 3; i++)
      m[3 * e + i] = in[i] + 2.0 * (r * A[i] + B[i]);
  }
}

/* geometry.h:80-97: every row of lhs as a linear combination of rhs rows */
void euo_rotate_r3(const double *l, const double *r, double *o)
{
  double t[9];
  for (int row = 0; row < 3; row++)
    for (int i = 0; i < 3; i++)
      t[3 * row + i] = (l[3 * row + 0] * r[0 + i] + l[3 * row + 1] * r[3 + i])
                       + l[3 * row + 2] * r[6 + i];
  memcpy(o, t, sizeof t);
}

/* ------------------------------------------------------------------------ */
/* set-up: twining tap table (envutil_main.cc:1253-1355)                     */
/* ------------------------------------------------------------------------ */

int euo_make_spread(int w, int h, float d, float sigma, float threshold,
                    float *out, int max_taps)
{
  if (w <= 2) w = 2;
  if (h <= 0) h = w;
  if (w * h > max_taps) return -1;
  float wgt = (float)(1.0 / (w * h));
  double x0 = -(w - 1.0) / (2.0 * w);
  double dx = 1.0 / w;
  double y0 = -(h - 1.0) / (2.0 * h);
  double dy = 1.0 / h;
  sigma = (float)(sigma * -x0);
  double sum = 0.0;
  int n = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float wf = 1.0f;
      if (sigma > 0.0) {
        double wx = (x0 + x * dx) / sigma;
        double wy = (y0 + y * dy) / sigma;
        wf = (float)exp(-sqrt(wx * wx + wy * wy));
      }
      out[3 * n + 0] = (float)(d * (x0 + x * dx));
      out[3 * n + 1] = (float)(d * (y0 + y * dy));
      out[3 * n + 2] = wf * wgt;
      sum += wf * wgt;
      n++;
    }
  if (sigma != 0.0) {
    double th_sum = 0.0;
    int renormalize = 0;
    for (int i = 0; i < n; i++) {
      out[3 * i + 2] = (float)(out[3 * i + 2] / sum);
      if (out[3 * i + 2] >= threshold) th_sum += out[3 * i + 2];
      else { renormalize = 1; out[3 * i + 2] = 0.0f; }
    }
    if (renormalize) {
      int m = 0;
      for (int i = 0; i < n; i++) {
        out[3 * i + 2] = (float)(out[3 * i + 2] / th_sum);
        if (out[3 * i + 2] > 0.0f) {
          out[3 * m + 0] = out[3 * i + 0];
          out[3 * m + 1] = out[3 * i + 1];
          out[3 * m + 2] = out[3 * i + 2];
          m++;
        }
      }
      n = m;
    }
  }
  return n;
}

/* ------------------------------------------------------------------------ */
/* b-spline basis: weight matrix (zimt/basis.h:419-545) and poles            */
/* ------------------------------------------------------------------------ */

/* value of the centred B-spline of degree n at x2/2, as the exact rational
 * (1/n!) * sum_k (-1)^k C(n+1,k) (x - k + (n+1)/2)_+^n evaluated in
 * long double on integers scaled by 2^n. The reference reads the same values
 * from a table of long double literals (zimt/poles.h K0..K45). */
static long double basis_half(int x2, int n)
{
  if (n == 0) return (x2 == -1 || x2 == 0) ? 1.0L : 0.0L;
  int ax = x2 < 0 ? -x2 : x2;
  if (ax > n) return 0.0L;
  long double acc = 0.0L, binom = 1.0L;
  for (int k = 0; k <= n + 1; k++) {
    /* t = 2*(x - k) + (n+1), i.e. twice the argument */
    long double t = (long double)(ax - 2 * k + (n + 1));
    if (t > 0) {
      long double p = 1.0L;
      for (int i = 0; i < n; i++) p *= t;
      acc += ((k & 1) ? -binom : binom) * p;
    }
    binom = binom * (long double)(n + 1 - k) / (long double)(k + 1);
  }
  long double den = 1.0L;
  for (int i = 2; i <= n; i++) den *= i;
  for (int i = 0; i < n; i++) den *= 2.0L;
  return acc / den;
}

/* basis.h:419-485 restated: Taylor coefficients of each polynomial piece,
 * built by repeated differencing of lower-degree basis values. */
void euo_weight_matrix(int degree, float *m)
{
  int order = degree + 1;
  long double line[EUO_MAX_DEGREE + 2];
  long double faculty = 1.0L;
  for (int row = 0; row < order; row++) {
    if (row > 1) faculty *= row;
    long double *first = line, *end = line + degree + 1;
    int mm = degree - row;
    if (mm == 0) { line[0] = 1.0L; first++; }
    else if (degree & 1)
      for (int x2 = -mm + 1; x2 <= mm - 1; x2 += 2) *first++ = basis_half(x2, mm);
    else
      for (int x2 = -mm; x2 <= mm; x2 += 2) *first++ = basis_half(x2, mm);
    for (long double *p = first; p < end; p++) *p = 0.0L;
    for (int d = mm; d < degree; d++) {
      long double *put = first, *pick = put - 1;
      while (pick >= line) { *put = *pick - *put; --put; --pick; }
      *put = -*put;
      first++;
    }
    for (int k = 0; k <= degree; k++)
      m[k * order + row] = (float)(line[k] / faculty);
  }
}

/* basis.h:650-690: weights for one delta; power accumulates, no Horner */
static void weights_from_matrix(const float *m, int degree, float delta, float *w)
{
  int order = degree + 1;
  for (int c = 0; c <= degree; c++) w[c] = m[c * order];
  if (!degree) return;
  float power = delta;
  for (int row = 1;; row++) {
    for (int c = 0; c <= degree; c++) w[c] += power * m[c * order + row];
    if (row == degree) break;
    power *= delta;
  }
}

void euo_basis_weights(int degree, float delta, float *w)
{
  float m[(EUO_MAX_DEGREE + 1) * (EUO_MAX_DEGREE + 1)];
  euo_weight_matrix(degree, m);
  weights_from_matrix(m, degree, delta, w);
}

/* Prefilter poles: roots inside the unit circle of sum_k beta^n(k) z^(k+n/2)
 * (zimt/poles.h tabulates them as long double literals). Computed here by
 * Newton iteration in long double from the same polynomial; only their
 * float value, the float gain and the horizon enter the arithmetic. */
static long double poly_eval(const long double *c, int len, long double z, long double *dp)
{
  long double p = c[len - 1], d = 0.0L;
  for (int i = len - 2; i >= 0; i--) { d = d * z + p; p = p * z + c[i]; }
  if (dp) *dp = d;
  return p;
}

int euo_poles(int degree, long double *poles)
{
  int np = degree / 2;
  if (np == 0) return 0;
  int n = degree, len = 2 * np + 1;
  long double c[2 * EUO_MAX_DEGREE + 3];
  for (int k = -np; k <= np; k++) c[k + np] = basis_half(2 * k, n);
  /* all roots are real, negative and simple; np of them lie in (-1, 0).
   * Walk a geometric grid from -1 towards 0, bisect every sign change,
   * polish with Newton steps. Results come out most negative first, the
   * order of the reference's tables. */
  int found = 0;
  long double za = -1.0L, pa = poly_eval(c, len, za, 0);
  while (found < np && za < -1e-12L) {
    long double zb = za * 0.9L, pb = poly_eval(c, len, zb, 0);
    if ((pa < 0) != (pb < 0)) {
      long double lo = za, hi = zb, plo = pa;
      for (int it = 0; it < 90; it++) {
        long double mid = 0.5L * (lo + hi), pm = poly_eval(c, len, mid, 0);
        if ((pm < 0) == (plo < 0)) { lo = mid; plo = pm; } else hi = mid;
      }
      long double z = 0.5L * (lo + hi);
      for (int it = 0; it < 4; it++) {
        long double dp, p = poly_eval(c, len, z, &dp);
        z -= p / dp;
      }
      poles[found++] = z;
    }
    za = zb; pa = pb;
  }
  return found;
}

/* ------------------------------------------------------------------------ */
/* b-spline container (zimt/bspline.h:233-450, :759-820)                     */
/* ------------------------------------------------------------------------ */

static long left_brace(int degree, int bc)
{
  long n = degree / 2;
  if (bc == EUO_REFLECT) n++;
  else if (degree & 1) n++;
  if (bc == EUO_PERIODIC && !(degree & 1)) n++;
  return n;
}

static long right_brace(int degree, int bc)
{
  long n = degree / 2;
  if (bc == EUO_REFLECT) { if (!(degree & 1)) n++; }
  if (degree & 1) n++;
  if (bc == EUO_PERIODIC) n++;
  return n;
}

void euo_spline_geometry(int degree, int bc0, int bc1, long w, long h, long *o)
{
  o[2] = left_brace(degree, bc0);  o[3] = left_brace(degree, bc1);
  o[4] = right_brace(degree, bc0); o[5] = right_brace(degree, bc1);
  o[0] = w + o[2] + o[4];
  o[1] = h + o[3] + o[5];
}

int euo_spline_init(euo_spline *s, float *container, long w, long h, int nch,
                    int degree, int bc0, int bc1)
{
  long g[6];
  euo_spline_geometry(degree, bc0, bc1, w, h, g);
  s->data = container;
  s->shape[0] = g[0]; s->shape[1] = g[1];
  s->stride[0] = 1;   s->stride[1] = g[0];
  s->left[0] = g[2];  s->left[1] = g[3];
  s->right[0] = g[4]; s->right[1] = g[5];
  s->core[0] = w;     s->core[1] = h;
  s->bc[0] = bc0;     s->bc[1] = bc1;
  s->degree = degree;
  s->nch = nch;
  return 0;
}

static float *core_px(const euo_spline *s, long x, long y)
{
  return s->data + ((s->left[1] + y) * s->stride[1] + (s->left[0] + x) * s->stride[0]) * s->nch;
}

void euo_spline_set_core(euo_spline *s, const float *core)
{
  for (long y = 0; y < s->core[1]; y++)
    memcpy(core_px(s, 0, y), core + y * s->core[0] * s->nch,
           sizeof(float) * s->core[0] * s->nch);
}

/* zimt/brace.h:134-330, one axis, slices span the whole container */
static void copy_slice(euo_spline *s, int axis, long to, long from)
{
  int other = 1 - axis;
  long n = s->shape[other];
  for (long i = 0; i < n; i++) {
    float *d = s->data + (to * s->stride[axis] + i * s->stride[other]) * s->nch;
    const float *p = s->data + (from * s->stride[axis] + i * s->stride[other]) * s->nch;
    for (int c = 0; c < s->nch; c++) d[c] = p[c];
  }
}

static void natural_slice(euo_spline *s, int axis, long to, long pivot, long from)
{
  int other = 1 - axis;
  long n = s->shape[other];
  for (long i = 0; i < n; i++) {
    float *d = s->data + (to * s->stride[axis] + i * s->stride[other]) * s->nch;
    const float *a = s->data + (pivot * s->stride[axis] + i * s->stride[other]) * s->nch;
    const float *b = s->data + (from * s->stride[axis] + i * s->stride[other]) * s->nch;
    for (int c = 0; c < s->nch; c++) d[c] = a[c] + a[c] - b[c];
  }
}

static void zero_slice(euo_spline *s, int axis, long to)
{
  int other = 1 - axis;
  for (long i = 0; i < s->shape[other]; i++) {
    float *d = s->data + (to * s->stride[axis] + i * s->stride[other]) * s->nch;
    for (int c = 0; c < s->nch; c++) d[c] = 0.0f;
  }
}

static void brace_axis(euo_spline *s, int axis)
{
  int bc = s->bc[axis];
  long lsz = s->left[axis], rsz = s->right[axis];
  long w = s->shape[axis], m = w - (lsz + rsz);
  if (m == 1) {
    for (long i = 0; i < w; i++) if (i != lsz) copy_slice(s, axis, i, lsz);
    return;
  }
  long l0 = lsz - 1, r0 = lsz + m, lp = l0 + 1, rp = r0 - 1, l1 = -1, r1 = w;
  long lt = l0, rt = r0, ls = 0, rs = 0, ds = 1;
  switch (bc) {
    case EUO_PERIODIC: ls = l0 + m; rs = r0 - m; ds = -1; break;
    case EUO_NATURAL:
    case EUO_MIRROR:   ls = l0 + 2; rs = r0 - 2; break;
    case EUO_CONSTANT:
    case EUO_REFLECT:  ls = l0 + 1; rs = r0 - 1; break;
    default: break;
  }
  for (long i = lsz > rsz ? lsz : rsz; i > 0; --i) {
    if (lt > l1) {
      switch (bc) {
        case EUO_PERIODIC: case EUO_MIRROR: case EUO_REFLECT:
          copy_slice(s, axis, lt, ls); break;
        case EUO_NATURAL:  natural_slice(s, axis, lt, lp, ls); break;
        case EUO_CONSTANT: copy_slice(s, axis, lt, lp); break;
        case EUO_ZEROPAD:  zero_slice(s, axis, lt); break;
        default: break;
      }
      --lt; ls += ds;
    }
    if (rt < r1) {
      switch (bc) {
        case EUO_PERIODIC: case EUO_MIRROR: case EUO_REFLECT:
          copy_slice(s, axis, rt, rs); break;
        case EUO_NATURAL:  natural_slice(s, axis, rt, rp, rs); break;
        case EUO_CONSTANT: copy_slice(s, axis, rt, rp); break;
        case EUO_ZEROPAD:  zero_slice(s, axis, rt); break;
        default: break;
      }
      ++rt; rs -= ds;
    }
  }
}

void euo_brace(euo_spline *s, int axis)
{
  if (axis < 0) { brace_axis(s, 0); brace_axis(s, 1); }
  else brace_axis(s, axis);
}

/* ------------------------------------------------------------------------ */
/* recursive prefilter (zimt/recursive.h:93-104, :385-620, :631-733, :790-860) */
/* one line of floats, element stride 'es'; math type float                  */
/* ------------------------------------------------------------------------ */

typedef struct {
  int bc, npoles;
  float pole[EUO_MAX_DEGREE / 2 + 1];
  long double lpole[EUO_MAX_DEGREE / 2 + 1];
  int horizon[EUO_MAX_DEGREE / 2 + 1];
  float gain;
} iir_t;

static void iir_init(iir_t *f, int bc, int degree, long double tolerance)
{
  f->bc = bc;
  f->npoles = euo_poles(degree, f->lpole);
  long double gain = 1.0L;
  for (int k = 0; k < f->npoles; k++) {
    f->pole[k] = (float)f->lpole[k];
    if (tolerance > 0)
      f->horizon[k] = (int)ceill(logl(tolerance) / logl(fabsl(f->lpole[k])));
    else
      f->horizon[k] = INT_MAX;
    gain *= (1.0L - f->lpole[k]) * (1.0L - 1.0L / f->lpole[k]);
  }
  f->gain = (float)gain;
}

static float icc(const iir_t *f, const float *c, long es, int M, int k)
{
  float z = f->pole[k], zn, z2n, iz, Sum;
  int n, hz = f->horizon[k];
  switch (f->bc) {
    case EUO_NATURAL:
      if (hz < M) {
        float c02 = c[0] + c[0];
        zn = z; Sum = c[0];
        for (n = 1; n < hz; n++) { Sum += zn * (c02 - c[n * es]); zn *= z; }
        return Sum;
      }
      zn = z; iz = 1.0f / z;
      z2n = (float)powl(f->lpole[k], (long double)(M - 1));
      Sum = ((1.0f + z) / (1.0f - z)) * (c[0] - z2n * c[(M - 1) * es]);
      z2n *= z2n * iz;
      for (n = 1; n <= M - 2; n++) { Sum -= (zn - z2n) * c[n * es]; zn *= z; z2n *= iz; }
      return Sum / (1.0f - zn * zn);
    case EUO_REFLECT:
      if (hz < M) {
        zn = z; Sum = c[0];
        for (n = 0; n < hz; n++) { Sum += zn * c[n * es]; zn *= z; }
        return Sum;
      }
      zn = z; iz = 1.0f / z;
      z2n = (float)powl(f->lpole[k], (long double)(2 * M));
      Sum = 0;
      for (n = 0; n < M - 1; n++) { Sum += (zn + z2n) * c[n * es]; zn *= z; z2n *= iz; }
      Sum += (zn + z2n) * c[n * es];
      return c[0] + Sum / (1.0f - zn * zn);
    case EUO_PERIODIC:
      if (hz < M) {
        zn = z; Sum = c[0];
        for (n = M - 1; n > (M - hz); n--) { Sum += zn * c[n * es]; zn *= z; }
      } else {
        zn = z; Sum = c[0];
        for (n = M - 1; n > 0; n--) { Sum += zn * c[n * es]; zn *= z; }
        Sum /= (1.0f - zn);
      }
      return Sum;
    case EUO_MIRROR:
      /* recursive.h:321-360 */
      if (hz < M) {
        zn = z; Sum = c[0];
        for (n = 1; n < hz; n++) { Sum += zn * c[n * es]; zn *= z; }
        return Sum;
      }
      zn = z; iz = 1.0f / z;
      z2n = (float)powl(f->lpole[k], (long double)(M - 1));
      Sum = c[0] + z2n * c[(M - 1) * es];
      z2n *= z2n * iz;
      for (n = 1; n <= M - 2; n++) { Sum += (zn + z2n) * c[n * es]; zn *= z; z2n *= iz; }
      return Sum / (1.0f - zn * zn);
    case EUO_GUESS:
      return c[0] * (float)(1.0 / (1.0 - (double)f->lpole[k]));
    default: /* ZEROPAD: identity */
      return c[0];
  }
}

static float iacc(const iir_t *f, const float *c, long es, int M, int k)
{
  float z = f->pole[k], zn, Sum;
  int hz = f->horizon[k];
  switch (f->bc) {
    case EUO_NATURAL:
      return -(z / ((1.0f - z) * (1.0f - z))) * (c[(M - 1) * es] - z * c[(M - 2) * es]);
    case EUO_REFLECT:
      return c[(M - 1) * es] / (1.0f - 1.0f / z);
    case EUO_PERIODIC:
      if (hz < M) {
        zn = z; Sum = c[(M - 1) * es] * z;
        for (int n = 0; n < hz; n++) { zn *= z; Sum += zn * c[n * es]; }
        Sum = -Sum;
      } else {
        zn = z; Sum = c[(M - 1) * es];
        for (int n = 0; n < M - 1; n++) { Sum += zn * c[n * es]; zn *= z; }
        Sum = z * Sum / (zn - 1.0f);
      }
      return Sum;
    case EUO_MIRROR:
    case EUO_GUESS:
      /* recursive.h:362-372 */
      return (z / (z * z - 1.0f)) * (c[(M - 1) * es] + z * c[(M - 2) * es]);
    default:
      return c[(M - 1) * es];
  }
}

static void solve_line(const iir_t *f, float *x, long es, int M)
{
  if (M == 1 || f->npoles < 1) return;
  float p = f->pole[0], g = f->gain, X;
  /* in place: every input value is read before its slot is overwritten */
  X = g * icc(f, x, es, M, 0);
  float prev_in;
  x[0] = X;
  for (int n = 1; n < M; n++) { prev_in = x[n * es]; X = g * prev_in + p * X; x[n * es] = X; }
  X = iacc(f, x, es, M, 0);
  x[(M - 1) * es] = X;
  for (int n = M - 2; n >= 0; n--) { X = p * (X - x[n * es]); x[n * es] = X; }
  for (int k = 1; k < f->npoles; k++) {
    p = f->pole[k];
    X = icc(f, x, es, M, k);
    x[0] = X;
    for (int n = 1; n < M; n++) { X = x[n * es] + p * X; x[n * es] = X; }
    X = iacc(f, x, es, M, k);
    x[(M - 1) * es] = X;
    for (int n = M - 2; n >= 0; n--) { X = p * (X - x[n * es]); x[n * es] = X; }
  }
}

void euo_filter_lines(float *base, long n_lines, long line_stride, long len,
                      long ele_stride, int bc, int degree, double tolerance)
{
  iir_t f;
  iir_init(&f, bc, degree, (long double)tolerance);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n_lines; i++)
    solve_line(&f, base + i * line_stride, ele_stride, (int)len);
}

/* zimt/prefilter.h:133-190 + bspline.h:1017-1041: both axes over the core,
 * tolerance = float epsilon, then brace all axes */
void euo_prefilter(euo_spline *s, int prefilter_degree)
{
  if (prefilter_degree > 1) {
    long W = s->core[0], H = s->core[1];
    int nch = s->nch;
    long double tol = FLT_EPSILON;
    for (int axis = 0; axis < 2; axis++) {
      iir_t f;
      iir_init(&f, s->bc[axis], prefilter_degree, tol);
      long nl = axis == 0 ? H : W, len = axis == 0 ? W : H;
#pragma omp parallel for schedule(static)
      for (long i = 0; i < nl; i++)
        for (int c = 0; c < nch; c++) {
          float *p = axis == 0 ? core_px(s, 0, i) + c : core_px(s, i, 0) + c;
          solve_line(&f, p, s->stride[axis] * nch, (int)len);
        }
    }
  }
  euo_brace(s, -1);
}

/* environment.h:356-522: full 360x180 lat/lon image */
void euo_spherical_prefilter(euo_spline *s, int degree)
{
  long W = s->core[0], H = s->core[1];
  int nch = s->nch;
  long half = W / 2;
  if (degree > 1) {
    iir_t f;
    iir_init(&f, EUO_PERIODIC, degree, 0.0001L);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < H; y++)
      for (int c = 0; c < nch; c++)
        solve_line(&f, core_px(s, 0, y) + c, nch, (int)W);
    /* left-half column top->bottom followed by the opposite column
     * bottom->top, filtered as one periodic line of length 2H */
#pragma omp parallel
    {
      float *line = (float *)malloc(sizeof(float) * 2 * (size_t)H);
#pragma omp for schedule(static)
      for (long x = 0; x < half; x++)
        for (int c = 0; c < nch; c++) {
          for (long y = 0; y < H; y++) {
            line[y] = core_px(s, x, y)[c];
            line[H + y] = core_px(s, x + half, H - 1 - y)[c];
          }
          solve_line(&f, line, 1, (int)(2 * H));
          for (long y = 0; y < H; y++) {
            core_px(s, x, y)[c] = line[y];
            core_px(s, x + half, H - 1 - y)[c] = line[H + y];
          }
        }
      free(line);
    }
  }
  /* over-the-pole frame rows: left half <-> right half, in the reference's order (environment.h:455-516): one
   * row above and one row below per round, each from a source row that only has to lie inside the CONTAINER - for
   * an image lower than its frame the sources run on into frame rows written in earlier rounds. For every other
   * image this is the plain "row -1 - k from row k, row H + k from row H - 1 - k". */
  {
    long y0 = -s->left[1], y1 = H + s->right[1];
    long us = 0, ut = -1, ls = H - 1, lt = H;
    while (1) {
      int c1 = us >= y0 && us < y1, c2 = ut >= y0 && ut < y1, c3 = ls >= y0 && ls < y1, c4 = lt >= y0 && lt < y1;
      if (!c2 && !c4) break;
      if (c2) {
        if (c1) {
          for (long x = 0; x < half; x++) {
            memcpy(core_px(s, x, ut), core_px(s, x + half, us), sizeof(float) * nch);
            memcpy(core_px(s, x + half, ut), core_px(s, x, us), sizeof(float) * nch);
          }
          us++;
        }
        ut--;
      }
      if (c4) {
        if (c3) {
          for (long x = 0; x < half; x++) {
            memcpy(core_px(s, x, lt), core_px(s, x + half, ls), sizeof(float) * nch);
            memcpy(core_px(s, x + half, lt), core_px(s, x, ls), sizeof(float) * nch);
          }
          ls--;
        }
        lt++;
      }
    }
  }
  brace_axis(s, 0);
}

/* ------------------------------------------------------------------------ */
/* evaluator: gates (zimt/map.h), split (basis.h:102-146), offsets and the   */
/* weighted sum (eval.h:904-1059, :1237-1300)                                */
/* ------------------------------------------------------------------------ */

typedef struct {
  const float *base;      /* core origin */
  int es[2];              /* strides in float elements */
  int degree, nch;
  int gate[2];            /* 0 clamp, 1 mirror, 2 periodic */
  float lower[2], upper[2];
  float m[(EUO_MAX_DEGREE + 1) * (EUO_MAX_DEGREE + 1)];
} ev_t;

static void ev_init(ev_t *e, const euo_spline *s, int degree)
{
  e->base = s->data + (s->left[1] * s->stride[1] + s->left[0] * s->stride[0]) * s->nch;
  e->es[0] = (int)(s->nch * s->stride[0]);
  e->es[1] = (int)(s->nch * s->stride[1]);
  e->degree = degree;
  e->nch = s->nch;
  for (int a = 0; a < 2; a++) {
    int bc = s->bc[a];
    /* bspline.h:233-270 */
    long double lo = 0.0L, up = (long double)(s->core[a] - 1);
    if (bc == EUO_REFLECT || bc == EUO_PERIODIC) { lo = -0.5L; up += 0.5L; }
    e->lower[a] = (float)lo;
    e->upper[a] = (float)up;
    if (s->core[a] == 1) { bc = EUO_CONSTANT; e->lower[a] = e->upper[a] = 0.0f; }
    e->gate[a] = bc == EUO_PERIODIC ? 2 : (bc == EUO_MIRROR || bc == EUO_REFLECT) ? 1 : 0;
  }
  euo_weight_matrix(degree, e->m);
}

/* map.h:268-279 */
static float v_fmod(float lhs, float rhs)
{
  float help = lhs;
  help /= rhs;
  help = truncf(help);
  help *= rhs;
  lhs -= help;
  if (fabsf(lhs) >= fabsf(rhs)) lhs = 0.0f;
  return lhs;
}

static float gate(const ev_t *e, int a, float c)
{
  float lower = e->lower[a], upper = e->upper[a];
  if (e->gate[a] == 2) {          /* map.h:423-440 */
    float cc = c - lower;
    float w = upper - lower;
    int below = cc < 0.0f, above = cc >= w;
    if (below || above) {
      float cm = v_fmod(cc, w);
      if (below) cm = cm + w;
      if (cm >= w) cm = 0.0f;
      cc = cm;
    }
    return cc + lower;
  }
  if (e->gate[a] == 1) {          /* map.h:341-357 */
    float cc = c - lower;
    float w = upper - lower;
    cc = fabsf(cc);
    if (cc >= w) {
      float cm = v_fmod(cc, 2 * w);
      cm -= w;
      cm = fabsf(cm);
      cm = w - cm;
      cc = cm;
    }
    return cc + lower;
  }
  /* clamp, map.h:231-236 */
  float r = c;
  if (c < lower) r = lower;
  if (c > upper) r = upper;
  return r;
}

static void ev_eval(const ev_t *e, float cx, float cy, float *out)
{
  int d = e->degree, nch = e->nch, order = d + 1;
  float g[2] = { gate(e, 0, cx), gate(e, 1, cy) };
  float fl[2], tune[2];
  int sel[2];
  for (int a = 0; a < 2; a++) {
    fl[a] = (d & 1) ? floorf(g[a]) : roundf(g[a]);
    tune[a] = g[a] - fl[a];
    sel[a] = (int)fl[a];
  }
  /* eval.h:1292-1294: int32 offset in float elements */
  int origin = sel[0] * e->es[0];
  origin += sel[1] * e->es[1];
  const float *p = e->base + origin;
  if (d == 0) {
    for (int c = 0; c < nch; c++) out[c] = p[c];
    return;
  }
  if (d == 1) {
    float wl0 = 1.0f - tune[0], wr0 = tune[0];
    float wl1 = 1.0f - tune[1], wr1 = tune[1];
    for (int c = 0; c < nch; c++) {
      float sum = p[c];
      sum *= wl0;
      sum += p[e->es[0] + c] * wr0;
      sum *= wl1;
      float sub = p[e->es[1] + c];
      sub *= wl0;
      sub += p[e->es[1] + e->es[0] + c] * wr0;
      sum += sub * wr1;
      out[c] = sum;
    }
    return;
  }
  float wx[EUO_MAX_DEGREE + 1], wy[EUO_MAX_DEGREE + 1];
  weights_from_matrix(e->m, d, tune[0], wx);
  weights_from_matrix(e->m, d, tune[1], wy);
  int off0 = -(d / 2);
  for (int c = 0; c < nch; c++) {
    float sum = 0.0f;
    for (int j = 0; j < order; j++) {
      const float *row = p + (j + off0) * e->es[1] + off0 * e->es[0] + c;
      float r = row[0];
      r *= wx[0];
      for (int i = 1; i < order; i++) r += wx[i] * row[i * e->es[0]];
      if (j == 0) { sum = r; sum *= wy[0]; }
      else sum += r * wy[j];
    }
    out[c] = sum;
  }
}

void euo_eval(const euo_spline *s, const float *crd, long n, float *out)
{
  ev_t e;
  ev_init(&e, s, s->degree);
  for (long i = 0; i < n; i++)
    ev_eval(&e, crd[2 * i], crd[2 * i + 1], out + i * s->nch);
}

void euo_eval_shifted(const euo_spline *s, int degree, const float *crd, long n,
                      float *out)
{
  ev_t e;
  ev_init(&e, s, degree);
  for (long i = 0; i < n; i++)
    ev_eval(&e, crd[2 * i], crd[2 * i + 1], out + i * s->nch);
}

/* ------------------------------------------------------------------------ */
/* geometry.h:1178-1289 cube face + in-face coordinate                       */
/* ------------------------------------------------------------------------ */

static void ray_to_cubeface(const float *c, int *face, float *in_face)
{
  float ax = fabsf(c[0]), ay = fabsf(c[1]), az = fabsf(c[2]);
  int m1 = ax >= ay, m2 = ax >= az, m3 = ay >= az;
  if (m1 && m2) {
    *face = c[0] < 0.0f ? 0 : 1;          /* CM_LEFT : CM_RIGHT */
    in_face[0] = -c[2] / c[0];
    in_face[1] = c[1] / ax;
  } else if (!m2 && !m3) {
    *face = c[2] < 0.0f ? 5 : 4;          /* CM_BACK : CM_FRONT */
    in_face[0] = c[0] / c[2];
    in_face[1] = c[1] / az;
  } else {
    *face = c[1] < 0.0f ? 2 : 3;          /* CM_TOP : CM_BOTTOM */
    in_face[0] = -c[0] / ay;
    in_face[1] = c[2] / c[1];
  }
}

/* ------------------------------------------------------------------------ */
/* cubemap source set-up (cubemap.h:233-400, :576-946, :1143-1233)           */
/* ------------------------------------------------------------------------ */

void euo_metrics_init(euo_metrics *m, long face_px, double face_fov,
                      long support_min_px, long tile_px)
{
  double overscan_md = 0.0, diameter_md = 2.0;
  m->face_px = face_px;
  m->radius_md = 1.0;
  m->inherent_support_px = 0;
  if (face_fov > M_PI_2) {
    m->radius_md = tan(face_fov / 2.0);
    diameter_md = 2.0 * m->radius_md;
    overscan_md = m->radius_md - 1.0;
  }
  m->model_to_px = (double)face_px / diameter_md;
  m->px_to_model = diameter_md / (double)face_px;
  double px_overscan = m->model_to_px * overscan_md;
  m->inherent_support_px = (long)trunc(px_overscan);
  m->discrete90 = (px_overscan - trunc(px_overscan) < 0.0000001);
  long additional = 0;
  if (m->inherent_support_px < support_min_px)
    additional = support_min_px - m->inherent_support_px;
  long px_min = face_px + 2 * additional;
  m->n_tiles = px_min / tile_px;
  if (m->n_tiles * tile_px < px_min) m->n_tiles++;
  m->section_px = m->n_tiles * tile_px;
  long frame_total = m->section_px - face_px;
  m->left_frame_px = frame_total / 2;
  m->right_frame_px = frame_total - m->left_frame_px;
  m->section_md = m->px_to_model * m->section_px;
  double refc_px = (double)m->left_frame_px + (double)face_px / 2.0;
  m->refc_md = m->px_to_model * refc_px;
}

/* cubemap.h:724-809 for one frame pixel; crd2 are the doubled integer
 * coordinates of linspace_t<int>(-(section_px-1), 2) */
static void fill_frame_px(const euo_metrics *m, const ev_t *bil, int face,
                          int ix, int iy, int ithird, float *px)
{
  float c3[3];
  switch (face) {
    case 4: c3[0] = (float)ix;  c3[1] = (float)iy; c3[2] = (float)ithird; break;
    case 5: c3[0] = (float)-ix; c3[1] = (float)iy; c3[2] = (float)-ithird; break;
    case 1: c3[0] = (float)ithird;  c3[1] = (float)iy; c3[2] = (float)-ix; break;
    case 0: c3[0] = (float)-ithird; c3[1] = (float)iy; c3[2] = (float)ix; break;
    case 3: c3[0] = (float)-ix; c3[1] = (float)ithird;  c3[2] = (float)iy; break;
    default: c3[0] = (float)-ix; c3[1] = (float)-ithird; c3[2] = (float)-iy; break;
  }
  int fv;
  float inf[2], pick[2];
  ray_to_cubeface(c3, &fv, inf);
  /* metrics_t::get_pickup_coordinate_px, cubemap.h:452-464: the sum is
   * formed in double (A.0), the compound operations narrow their operand */
  pick[0] = (float)((double)inf[0] + m->refc_md);
  pick[1] = (float)((double)inf[1] + m->refc_md);
  pick[0] *= (float)m->model_to_px;
  pick[1] *= (float)m->model_to_px;
  pick[1] += (float)(fv * (int)m->section_px);
  pick[0] -= .5f;
  pick[1] -= .5f;
  ev_eval(bil, pick[0], pick[1], px);
}

void euo_cubemap_build(const euo_metrics *m, const float *faces, int nch,
                       int spline_degree, int prefilter_degree, float *ir)
{
  long S = m->section_px, F = m->face_px, lf = m->left_frame_px, rf = m->right_frame_px;
  (void)spline_degree;
  euo_spline s;
  memset(&s, 0, sizeof s);
  s.data = ir; s.shape[0] = S; s.shape[1] = 6 * S; s.stride[0] = 1; s.stride[1] = S;
  s.core[0] = S; s.core[1] = 6 * S; s.bc[0] = s.bc[1] = EUO_REFLECT;
  s.degree = 1; s.nch = nch;
  /* faces into their slots (cubemap.h:1147-1151, :1219) */
  for (int f = 0; f < 6; f++)
    for (long y = 0; y < F; y++)
      memcpy(ir + (((f * S) + lf + y) * S + lf) * nch,
             faces + ((f * F + y) * F) * nch, sizeof(float) * F * nch);
  if (lf || rf) {
    /* mirror_around, cubemap.h:607-659 */
    for (int f = 0; f < 6; f++) {
      float *cf = ir + ((f * S + lf) * S + lf) * nch;
#define CF(x, y) (cf + ((long)(y) * S + (long)(x)) * nch)
      int cmin = lf > 0 ? -1 : 0;
      int cmax = rf > 0 ? (int)F : (int)F - 1;
      for (int x = cmin; x <= cmax; x++) {
        if (lf) memcpy(CF(x, -1), CF(x, 0), sizeof(float) * nch);
        if (rf) memcpy(CF(x, F), CF(x, F - 1), sizeof(float) * nch);
      }
      for (int y = cmin; y <= cmax; y++) {
        if (lf) memcpy(CF(-1, y), CF(0, y), sizeof(float) * nch);
        if (rf) memcpy(CF(F, y), CF(F - 1, y), sizeof(float) * nch);
      }
#undef CF
    }
    /* fill_support, cubemap.h:819-911 */
    ev_t bil;
    ev_init(&bil, &s, 1);
    int ithird = (int)(m->model_to_px * 2);
    int ishift = (int)S - 1;
    for (int f = 0; f < 6; f++) {
      float *sec = ir + (f * S) * S * nch;
      long win[4][4] = {
        { 0, 0, S, lf },
        { 0, S - rf, S, S },
        { 0, lf, lf, S - rf },
        { lf + F, lf, S, S - rf } };
      int on[4] = { lf > 0, rf > 0, lf > 0, rf > 0 };
      /* The order is part of the result. With an odd face size (right frame =
       * left frame + 1) the frame pixels in the row / column next to the face tie
       * in the dominant-axis test and map onto their OWN face's edge: they read
       * frame pixels of this section, among them the one to their left / above.
       * The restated order is the reference's with one worker thread: stripes in
       * the order above, below, left, right; rows top to bottom; a row in vectors
       * of 16 lanes from the window's first column, every vector evaluated from
       * the array as it stands and then stored (zimt::process, wielding.h:337-463).
       * (With several threads the reference races on exactly these pixels.) */
      for (int k = 0; k < 4; k++) {
        if (!on[k]) continue;
        for (long y = win[k][1]; y < win[k][3]; y++)
          for (long x0 = win[k][0]; x0 < win[k][2]; x0 += EUO_LANES) {
            float tmp[EUO_LANES][4];
            long n = win[k][2] - x0 < EUO_LANES ? win[k][2] - x0 : EUO_LANES;
            for (long l = 0; l < n; l++)
              fill_frame_px(m, &bil, f, (int)(2 * (x0 + l) - ishift), (int)(2 * y - ishift),
                            ithird, tmp[l]);
            for (long l = 0; l < n; l++)
              memcpy(sec + (y * S + x0 + l) * nch, tmp[l], sizeof(float) * nch);
          }
      }
    }
  }
  /* per-face prefilter, NATURAL x NATURAL, float epsilon (cubemap.h:921-946) */
  if (prefilter_degree > 1) {
    for (int f = 0; f < 6; f++) {
      float *sec = ir + (f * S) * S * nch;
      for (int c = 0; c < nch; c++)
        euo_filter_lines(sec + c, S, S * nch, S, nch, EUO_NATURAL,
                         prefilter_degree, FLT_EPSILON);
      for (int c = 0; c < nch; c++)
        euo_filter_lines(sec + c, S, nch, S, S * nch, EUO_NATURAL,
                         prefilter_degree, FLT_EPSILON);
    }
  }
}

/* ------------------------------------------------------------------------ */
/* inverse_lcp (lens_correction.h:236-301): the inverse of the PTO radial     */
/* factor as a cubic b-spline over sqrt-spaced knots, built by Newton          */
/* iteration in double (eu_polynomial::inverse, :112-139). Pinned against the  */
/* reference's own class through oracle/ref_zimt.cc (ref_inverse_lcp).         */
/* ------------------------------------------------------------------------ */

#define EUO_INV_LCP_MAX 136
typedef struct {
  int nk;
  long left;
  double rr_max;
  float coef[EUO_INV_LCP_MAX];   /* braced, prefiltered; core starts at coef[left] */
  float m[16];                   /* weight matrix, degree 3 */
} inv_lcp_t;

static double poly4_function(const double *cf, double x)
{
  double sum = 0.0, power = 1.0;
  for (int i = 0; i <= 4; i++) { sum += cf[4 - i] * power; power *= x; }
  return sum;
}
static double poly4_derivative(const double *dcf, double x)
{
  double sum = 0.0, power = 1.0;
  for (int i = 0; i < 4; i++) { sum += dcf[4 - i - 1] * power; power *= x; }
  return sum;
}

static int inv_lcp_init(inv_lcp_t *q, double a, double b, double c, double r_max_in, int sz)
{
  double cf[5] = { a, b, c, 1.0 - (a + b + c), 0.0 }, dcf[5];
  { int power = 4; for (int i = 0; i <= 4; i++) { dcf[i] = cf[i] * power; --power; } }
  int nk = sz + 4;
  long left = left_brace(3, EUO_NATURAL), right = right_brace(3, EUO_NATURAL);
  if (nk + left + right > EUO_INV_LCP_MAX) return 0;
  double r_max = r_max_in * ((sz + 3.0) / sz);
  q->rr_max = poly4_function(cf, r_max);
  q->nk = nk;
  q->left = left;
  float *core = q->coef + left;
  for (int i = 0; i < nk; i++) {
    double notch = (double)i / (nk - 1);
    notch *= notch;
    notch *= q->rr_max;
    double out = i * r_max / sz;
    {                                           /* eu_polynomial::inverse, tolerance 100 eps */
      const double tolerance = 100 * DBL_EPSILON;
      double current = out, result, difference = 0.0, last_difference = DBL_MAX;
      for (int count = 0; count < 16; count++) {
        result = poly4_function(cf, current);
        difference = notch - result;
        if (last_difference == difference) break;
        if (fabs(difference) <= tolerance) break;
        last_difference = difference;
        current = current + difference / poly4_derivative(dcf, current);
      }
      if (fabs(difference) < tolerance) out = current;
    }
    core[i] = (float)(notch == 0.0 ? 1.0 / poly4_derivative(dcf, 0.0) : (out / notch) - 1);
  }
  /* bspline<float,1>::prefilter: one line, then the NATURAL brace (point mirror, brace.h:134+) */
  euo_filter_lines(core, 1, nk, nk, 1, EUO_NATURAL, 3, FLT_EPSILON);
  for (long k = 1; k <= left; k++) core[-k] = core[0] + core[0] - core[k];
  for (long k = 1; k <= right; k++) core[nk - 1 + k] = core[nk - 1] + core[nk - 1] - core[nk - 1 - k];
  euo_weight_matrix(3, q->m);
  return 1;
}

/* inverse_lcp::eval (:289-299) for one lane: the argument arrives as double (norm / s, A.0) */
static float inv_lcp_eval(const inv_lcp_t *q, double x)
{
  double t = x / q->rr_max;
  t = sqrt(t);
  t *= (q->nk - 1);
  float c = (float)t;
  /* make_safe_evaluator on a NATURAL spline: clamp gate [0, nk - 1] (map.h:231-236) */
  float lower = 0.0f, upper = (float)(q->nk - 1);
  float g = c;
  if (c < lower) g = lower;
  if (c > upper) g = upper;
  float fl = floorf(g), delta = g - fl;
  int sel = (int)fl;
  float w[4];
  weights_from_matrix(q->m, 3, delta, w);
  const float *p = q->coef + q->left + sel - 1;
  float sum = p[0];
  sum *= w[0];
  for (int i = 1; i < 4; i++) sum += w[i] * p[i];
  return sum + 1.0f;
}

int euo_inverse_lcp(double a, double b, double c, double r_max, int sz, const float *x, long n,
                    float *out, float *knots, int max_knots)
{
  inv_lcp_t q;
  if (!inv_lcp_init(&q, a, b, c, r_max, sz)) return 0;
  for (long i = 0; i < n; i++) out[i] = inv_lcp_eval(&q, (double)x[i]);
  for (int i = 0; knots && i < q.nk && i < max_knots; i++) knots[i] = q.coef[q.left + i];
  return q.nk;
}

/* ------------------------------------------------------------------------ */
/* steppers (stepper.h) - one instance per row segment                       */
/* ------------------------------------------------------------------------ */

typedef struct {
  int projection, normalize;
  int width, height;
  float fx0, fx1, fy0, fy1, bias_x, bias_y, delta;
  float xx[3], yy[3], zz[3];
  float section_md, refc_md;     /* cubemap/biatan6 */
  int x_off, y_off;              /* bill.get_offset (wielding.h:215-224)   */
  /* generic_stepper (stepper.h:353-490) over tf_ex_facet (envutil_payload.cc:1841-1885): used for
   * a facet with translation parameters. tf33 = generic_r3, a tf3d_t (geometry.h:1850-1941) */
  int generic, ntf;
  struct tf3d { int has_shift; float trg_to_md[9], md_to_src[9], trg_to_src[9], shift[3], dcp; } tf[2];   /* r3_t<float>: m[3 * i + c] = r[i][c] */
  /* tf22 = pto_planar<T, L, true> of the target (environment.h:285-307), --single only */
  int inv_shear, inv_shift, inv_lcp;
  double shear_g, shear_t, inv_s;
  float inv_h, inv_v;
  const inv_lcp_t *inv_model;
} stepper_t;

/* rotate(xel_t<U,3>, r3_t<T>), geometry.h:80-87: (lhs[0] * rhs[0]) + (lhs[1] * rhs[1]) + (lhs[2] * rhs[2]) */
static void rotate_f(const float *in, const float *m, float *out)
{
  float t[3];
  for (int c = 0; c < 3; c++) {
    float v = in[0] * m[c] + in[1] * m[3 + c];
    t[c] = v + in[2] * m[6 + c];
  }
  out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}
static void rotate_m_f(const float *lhs, const float *rhs, float *out9)   /* r3_t<float> x r3_t<float> */
{
  float t[9];
  for (int i = 0; i < 3; i++) rotate_f(lhs + 3 * i, rhs, t + 3 * i);
  memcpy(out9, t, sizeof t);
}
static void make_r3_f(double roll, double pitch, double yaw, int inverse, float *m9)
{
  double d[9];
  euo_make_r3(roll, pitch, yaw, inverse, d);     /* r3_t<double> narrowed into r3_t<float> */
  for (int i = 0; i < 9; i++) m9[i] = (float)d[i];
}

/* generic_r3(ft, fs), envutil_payload.cc:1636-1755: ft the target in camera position, fs the source
 * facet; ft6/fs6 = { tr_x, tr_y, tr_z, tp_y, tp_p, tp_r } */
static void generic_r3_init(stepper_t *s, const double *ft_rpy, const double *ft6, const double *fs_rpy,
                            const double *fs6)
{
  float r_camera[9], rt_tp[9], rt_tpi[9], rs_tp[9], rs_tpi[9], r_facet[9];
  make_r3_f(ft_rpy[0], ft_rpy[1], ft_rpy[2], 0, r_camera);
  make_r3_f(ft6[5], ft6[4], ft6[3], 1, rt_tp);
  make_r3_f(ft6[5], ft6[4], ft6[3], 0, rt_tpi);
  make_r3_f(fs6[5], fs6[4], fs6[3], 1, rs_tp);
  make_r3_f(fs6[5], fs6[4], fs6[3], 0, rs_tpi);
  make_r3_f(fs_rpy[0], fs_rpy[1], fs_rpy[2], 1, r_facet);
  int have_ttp = ft6[0] != 0 || ft6[1] != 0 || ft6[2] != 0;
  int have_stp = fs6[0] != 0 || fs6[1] != 0 || fs6[2] != 0;
  float shift_t[3] = { (float)ft6[0], (float)ft6[1], (float)ft6[2] };
  if (ft6[3] != 0 || ft6[4] != 0 || ft6[5] != 0) {
    /* rotate(xel_t<double,3>(shift_t), rt_tp): double vector, float matrix, narrowed on assignment */
    double v[3] = { shift_t[0], shift_t[1], shift_t[2] }, o[3];
    for (int c = 0; c < 3; c++) o[c] = (v[0] * (double)rt_tp[c] + v[1] * (double)rt_tp[3 + c]) + v[2] * (double)rt_tp[6 + c];
    for (int c = 0; c < 3; c++) shift_t[c] = (float)o[c];
  }
  float dcp = (float)(1.0 - (double)shift_t[2]);
  for (int c = 0; c < 3; c++) shift_t[c] = -shift_t[c];
  float shift_s[3] = { (float)fs6[0], (float)fs6[1], (float)fs6[2] };
  if (fs6[3] != 0 || fs6[4] != 0 || fs6[5] != 0) {
    double v[3] = { shift_s[0], shift_s[1], shift_s[2] }, o[3];
    for (int c = 0; c < 3; c++) o[c] = (v[0] * (double)rs_tp[c] + v[1] * (double)rs_tp[3 + c]) + v[2] * (double)rs_tp[6 + c];
    for (int c = 0; c < 3; c++) shift_s[c] = (float)o[c];
  }
  /* tf3d_t ctor (geometry.h:1866-1879) */
#define TF3D_SET(T, A, B, SH, DCP) do { \
    memcpy((T)->trg_to_md, (A), 9 * sizeof(float)); memcpy((T)->md_to_src, (B), 9 * sizeof(float)); \
    rotate_m_f((A), (B), (T)->trg_to_src); \
    for (int c_ = 0; c_ < 3; c_++) (T)->shift[c_] = (SH)[c_]; \
    (T)->has_shift = (T)->shift[0] != 0 || (T)->shift[1] != 0 || (T)->shift[2] != 0; \
    (T)->dcp = (DCP); } while (0)
  float m1[9], m2[9];
  const float zero3[3] = { 0.0f, 0.0f, 0.0f };
  s->ntf = 1;
  if (have_ttp && have_stp) {          /* case 1: tf3d1 + tf3d2 */
    rotate_m_f(r_camera, rt_tp, m1);
    TF3D_SET(&s->tf[0], m1, rt_tpi, shift_t, dcp);
    rotate_m_f(rs_tpi, r_facet, m2);
    TF3D_SET(&s->tf[1], rs_tp, m2, shift_s, 1.0f);
    s->ntf = 2;
  } else if (have_ttp) {               /* case 2 */
    rotate_m_f(r_camera, rt_tp, m1);
    rotate_m_f(rt_tpi, r_facet, m2);
    TF3D_SET(&s->tf[0], m1, m2, shift_t, dcp);
  } else if (have_stp) {               /* case 3 */
    rotate_m_f(r_camera, rs_tp, m1);
    rotate_m_f(rs_tpi, r_facet, m2);
    TF3D_SET(&s->tf[0], m1, m2, shift_s, 1.0f);
  } else {                             /* case 4: rotate_t(rotate(r_camera, r_facet)) */
    TF3D_SET(&s->tf[0], r_camera, r_facet, zero3, 1.0f);
  }
#undef TF3D_SET
  s->generic = 1;
}

/* roll_out_23 (geometry.h:1800-1837): planar -> ray by the target's projection, float */
static int planar_to_ray_f(int projection, const float *in, float *out)
{
  switch (projection) {
    case EUO_SPHERICAL: {        /* ll_to_ray_t, geometry.h:152-211 */
      float sinlat = sinf(in[1]), coslat = cosf(in[1]), sinlon = sinf(in[0]), coslon = cosf(in[0]);
      out[0] = sinlon * coslat; out[2] = coslon * coslat; out[1] = sinlat;
      return 1;
    }
    case EUO_CYLINDRICAL:        /* cyl_to_ray_t, geometry.h:417-446 */
      out[2] = cosf(in[0]); out[0] = sinf(in[0]); out[1] = in[1];
      return 1;
    case EUO_RECTILINEAR:        /* rect_to_ray_t, geometry.h:363-387 */
      out[0] = in[0]; out[1] = in[1]; out[2] = 1.0f;
      return 1;
    case EUO_STEREOGRAPHIC: {    /* ster_to_ray_t, geometry.h:481-510 */
      float r = sqrtf(in[0] * in[0] + in[1] * in[1]);
      float theta = atanf(r / 2.0f) * 2.0f;
      float phi = atan2f(in[0], -in[1]);
      out[2] = cosf(theta); out[1] = -sinf(theta) * cosf(phi); out[0] = sinf(theta) * sinf(phi);
      return 1;
    }
    case EUO_FISHEYE: {          /* fish_to_ray_t, geometry.h:539-566 */
      float r = sqrtf(in[0] * in[0] + in[1] * in[1]);
      float phi = atan2f(in[0], -in[1]);
      out[2] = cosf(r); out[1] = -sinf(r) * cosf(phi); out[0] = sinf(r) * sinf(phi);
      return 1;
    }
    case EUO_CUBEMAP:
    case EUO_BIATAN6: {          /* ir_to_ray_t (geometry.h:663-770) / ba6_to_ray_t (:855-1000), default-constructed
                                  * by roll_out_23: section_md 2.0, refc_md 1.0 (a cubemap of 90-degree faces) */
      float c0 = in[0] + 1.0f, c1 = in[1] + 6.0f;          /* crd2 += ul2c = { refc_md, 3 * section_md } */
      int section = (int)((double)c1 / 2.0);
      c1 = (float)((double)c1 - (double)section * 2.0);
      c0 -= 1.0f; c1 -= 1.0f;                              /* crd2 -= refc_md */
      if (projection == EUO_BIATAN6) {                     /* crd2 = tan(crd2 * T(M_PI / 4)) */
        c0 = tanf(c0 * (float)(M_PI / 4)); c1 = tanf(c1 * (float)(M_PI / 4));
      }
      switch (section) {                                  /* CM_LEFT 0 .. CM_BACK 5, envutil_basic.h:56-64 */
        case 1: out[0] = 1.0f;  out[1] = c1;    out[2] = -c0; break;
        case 0: out[0] = -1.0f; out[1] = c1;    out[2] = c0;  break;
        case 3: out[0] = -c0;   out[1] = 1.0f;  out[2] = c1;  break;
        case 2: out[0] = -c0;   out[1] = -1.0f; out[2] = -c1; break;
        case 4: out[0] = c0;    out[1] = c1;    out[2] = 1.0f; break;
        case 5: out[0] = -c0;   out[1] = c1;    out[2] = -1.0f; break;
        default: out[0] = out[1] = out[2] = 0.0f;         /* no lane of crd3 is written */
      }
      return 1;
    }
  }
  return 0;
}

/* tf3d_t::eval, geometry.h:1896-1941; the all_of / any_of tests only skip work */
static void tf3d_eval(const struct tf3d *s, const float *in, float *out)
{
  if (!s->has_shift) { rotate_f(in, s->trg_to_src, out); return; }
  float t[3];
  rotate_f(in, s->trg_to_md, t);
  int mask = t[2] <= 0.0f;
  t[0] /= t[2]; t[1] /= t[2]; t[2] = 1.0f;
  for (int c = 0; c < 3; c++) t[c] *= s->dcp;
  for (int c = 0; c < 3; c++) t[c] -= s->shift[c];
  rotate_f(t, s->md_to_src, out);
  if (mask) { out[0] = 0.0f; out[1] = 0.0f; out[2] = -INFINITY; }
}

/* stepper.h:294-307 */
static void stepper_init(stepper_t *s, int projection, int normalize,
                         const double *basis9, int width, int height,
                         double a0d, double a1d, double b0d, double b1d,
                         float bias_x, float bias_y)
{
  float a0 = (float)a0d, a1 = (float)a1d, b0 = (float)b0d, b1 = (float)b1d;
  s->projection = projection;
  s->normalize = normalize;
  s->width = width;
  s->height = height;
  s->fx1 = (float)(a1 / (2.0 * width));
  s->fx0 = (float)(a0 / (2.0 * width));
  s->fy1 = (float)(b1 / (2.0 * height));
  s->fy0 = (float)(b0 / (2.0 * height));
  s->bias_x = bias_x * (a1 - a0) / (float)width;
  s->bias_y = bias_y * (b1 - b0) / (float)height;
  s->delta = (float)EUO_LANES * (a1 - a0) / (float)width;
  for (int i = 0; i < 3; i++) {
    s->xx[i] = (float)basis9[i];
    s->yy[i] = (float)basis9[3 + i];
    s->zz[i] = (float)basis9[6 + i];
  }
  s->section_md = a1 - a0;
  s->refc_md = (float)((a1 - a0) / 2.0);
  s->x_off = s->y_off = 0;
  s->generic = 0;
}

/* fuse() (envutil_payload.cc:2095-2110, :2145-2158): a facet with translation parameters gets
 * generic_stepper(tf_ex_facet(args, fct)); returns 0 when this build does not restate the case */
static int stepper_make_generic(stepper_t *s, const euo_job *job, const euo_source *src, const inv_lcp_t *inv_model)
{
  double ft_rpy[3] = { job->roll, job->pitch, job->yaw }, ft6[6] = { 0, 0, 0, 0, 0, 0 };
  s->inv_shear = s->inv_shift = s->inv_lcp = 0;
  s->inv_model = inv_model;
  if (job->single) {
    const euo_source *t = job->single;
    ft6[0] = t->tr_x; ft6[1] = t->tr_y; ft6[2] = t->tr_z; ft6[3] = t->tp_y; ft6[4] = t->tp_p; ft6[5] = t->tp_r;
    s->inv_shear = t->shear_g != 0.0 || t->shear_t != 0.0;
    s->inv_shift = t->h != 0.0 || t->v != 0.0;
    s->inv_lcp = t->a != 0.0 || t->b != 0.0 || t->c != 0.0;
    s->shear_g = t->shear_g; s->shear_t = t->shear_t;
    s->inv_h = (float)t->h; s->inv_v = (float)t->v;
    { /* facet_spec::process_geometry: the reference radius is half the smaller edge (envutil_basic.h:508-513) */
      double dv = fabs(job->y1 - job->y0) / 2.0, dh = fabs(job->x1 - job->x0) / 2.0;
      s->inv_s = (dh < dv) ? dh : dv; }
  }
  double fs_rpy[3] = { src->roll, src->pitch, src->yaw };
  double fs6[6] = { src->tr_x, src->tr_y, src->tr_z, src->tp_y, src->tp_p, src->tp_r };
  float probe[2] = { 0.0f, 0.0f }, r[3];
  if (!planar_to_ray_f(s->projection, probe, r)) return 0;
  generic_r3_init(s, ft_rpy, ft6, fs_rpy, fs6);
  return s->generic == 1;
}
static int has_translation(const euo_source *src) { return src->tr_x != 0 || src->tr_y != 0 || src->tr_z != 0; }
/* fuse(), envutil_payload.cc:2058-2069: the single facet has lens correction or translation */
static int generic_target(const euo_job *job)
{
  const euo_source *t = job->single;
  if (!t) return 0;
  int has_2d_tf = t->h != 0.0 || t->v != 0.0 || t->a != 0.0 || t->b != 0.0 || t->c != 0.0 || t->shear_g != 0.0 || t->shear_t != 0.0;
  return has_2d_tf || has_translation(t);
}
/* the inverse lens model of the single facet: r_max from the facet's extent (facet_spec::process_geometry,
 * envutil_basic.h:508-520), sz = 100 (environment.h:251) */
static int single_inv_model(const euo_job *job, inv_lcp_t *q)
{
  const euo_source *t = job->single;
  if (!t || !(t->a != 0.0 || t->b != 0.0 || t->c != 0.0)) return 1;
  double dv = fabs(job->y1 - job->y0) / 2.0, dh = fabs(job->x1 - job->x0) / 2.0;
  double aspect = (dh >= dv) ? dh / dv : dv / dh;
  return inv_lcp_init(q, t->a, t->b, t->c, sqrt(1 + aspect * aspect), 100);
}

/* planar coordinate of pixel (x, y): stepper.h:324-350. The x value is the
 * segment-start value of the pixel's lane plus k additions of delta. */
static void planar_at(const stepper_t *s, int x, int y, float *p)
{
  int seg_start = (x / EUO_SEGMENT) * EUO_SEGMENT;
  int in_seg = x - seg_start;
  int lane = in_seg % EUO_LANES, k = in_seg / EUO_LANES;
  /* segments start at multiples of 512 of the PROCESSED shape; the offset is
   * added to the coordinate handed to init (wielding.h:215-224) */
  float ll0 = (float)(2 * lane) + (float)((seg_start + s->x_off) * 2 + 1);
  float p0 = s->bias_x + ll0 * s->fx1 + ((float)(2 * s->width) - ll0) * s->fx0;
  for (int i = 0; i < k; i++) p0 += s->delta;
  int ll1 = (y + s->y_off) * 2 + 1;
  float p1 = s->bias_y + ll1 * s->fy1 + (float)(2 * s->height - ll1) * s->fy0;
  p[0] = p0;
  p[1] = p1;
}

static void normalize3(float *t)
{
  /* xel.h:752-765: sqn = v0*v0; sqn += v1*v1; sqn += v2*v2 */
  float sqn = t[0] * t[0];
  sqn += t[1] * t[1];
  sqn += t[2] * t[2];
  float n = sqrtf(sqn);
  t[0] /= n; t[1] /= n; t[2] /= n;
}

/* ray for pixel (x,y) */
static void stepper_ray(const stepper_t *s, int x, int y, float *trg)
{
  float pl[2];
  planar_at(s, x, y, pl);
  if (s->generic) {             /* generic_stepper::init / increase, stepper.h:422-473 */
    float r[3];
    /* tf_ex_facet::eval (envutil_payload.cc:1869-1884): tf22 (if has_2d_tf), tf23, tf33 */
    if (s->inv_shear) {           /* pto_planar<T, L, true>, environment.h:290-295: doubles, narrowed on assignment */
      pl[1] = (float)(((double)pl[1] - s->shear_t * (double)pl[0]) / (1 - s->shear_t * s->shear_g));
      pl[0] = (float)((double)pl[0] - s->shear_g * (double)pl[1]);
    }
    if (s->inv_shift) { pl[0] -= s->inv_h; pl[1] -= s->inv_v; }
    if (s->inv_lcp) {
      float sqn = pl[0] * pl[0];
      sqn += pl[1] * pl[1];
      float factor = inv_lcp_eval(s->inv_model, (double)sqrtf(sqn) / s->inv_s);
      pl[0] *= factor; pl[1] *= factor;
    }
    planar_to_ray_f(s->projection, pl, r);
    tf3d_eval(&s->tf[0], r, trg);
    if (s->ntf == 2) { float t[3] = { trg[0], trg[1], trg[2] }; tf3d_eval(&s->tf[1], t, trg); }
    if (s->normalize) normalize3(trg);
    return;
  }
  const float *xx = s->xx, *yy = s->yy, *zz = s->zz;
  switch (s->projection) {
    case EUO_SPHERICAL: {       /* stepper.h:605-667 */
      float sy = sinf(pl[1]), r = cosf(pl[1]);
      float sx = sinf(pl[0]), z = cosf(pl[0]);
      for (int i = 0; i < 3; i++) {
        float xxx = xx[i] * r, yyy = yy[i] * sy, zzz = zz[i] * r;
        trg[i] = xxx * sx + zzz * z + yyy;
      }
      break;
    }
    case EUO_CYLINDRICAL: {     /* stepper.h:760-800 */
      float sx = sinf(pl[0]), z = cosf(pl[0]);
      /* the reciprocal length is taken once, from the first vector of the
       * segment, and reused (stepper.h:771-775, :786) */
      for (int i = 0; i < 3; i++) {
        float yyy = yy[i] * pl[1];
        trg[i] = xx[i] * sx + zz[i] * z + yyy;
      }
      if (s->normalize) {
        /* rcp_length is a per-lane value computed at init() from that
         * lane's first pixel in the segment */
        int seg_start = (x / EUO_SEGMENT) * EUO_SEGMENT;
        int lane = (x - seg_start) % EUO_LANES;
        float p0[2], t0[3];
        planar_at(s, seg_start + lane, y, p0);
        float sx0 = sinf(p0[0]), z0 = cosf(p0[0]);
        for (int i = 0; i < 3; i++) {
          float yyy = yy[i] * p0[1];
          t0[i] = xx[i] * sx0 + zz[i] * z0 + yyy;
        }
        float sqn = t0[0] * t0[0];
        sqn += t0[1] * t0[1];
        sqn += t0[2] * t0[2];
        float rcp = 1.0f / sqrtf(sqn);
        trg[0] *= rcp; trg[1] *= rcp; trg[2] *= rcp;
      }
      break;
    }
    case EUO_RECTILINEAR: {     /* stepper.h:895-932 */
      for (int i = 0; i < 3; i++) {
        float ddd = yy[i] * pl[1] + zz[i];
        trg[i] = xx[i] * pl[0] + ddd;
      }
      if (s->normalize) normalize3(trg);
      break;
    }
    case EUO_FISHEYE:
    case EUO_STEREOGRAPHIC: {   /* stepper.h:1019-1030, :1146-1157 */
      float sqn = pl[0] * pl[0];
      sqn += pl[1] * pl[1];
      float nrm = sqrtf(sqn);
      double ad;
      if (s->projection == EUO_FISHEYE)
        ad = M_PI_2 - (double)nrm;
      else
        ad = M_PI_2 - 2.0 * atan((double)nrm / 2.0);
      /* sincos(vec<double>, vec<float>&, vec<float>&) resolves to the float
       * overload: the angle narrows to float first */
      float a = (float)ad;
      float b = atan2f(pl[0], pl[1]);
      float z = sinf(a), r = cosf(a);
      float sx = sinf(b), sy = cosf(b);
      for (int i = 0; i < 3; i++)
        trg[i] = xx[i] * r * sx + zz[i] * z + yy[i] * r * sy;
      break;
    }
    case EUO_CUBEMAP:
    case EUO_BIATAN6: {         /* stepper.h:1274-1358, :1449-1560 */
      int face = (y + s->y_off) / s->width;
      float p1 = pl[1] + (float)(3 - face) * s->section_md - s->refc_md;
      float p0 = pl[0];
      if (s->projection == EUO_BIATAN6) {
        p1 = tanf(p1 * (float)(M_PI / 4.0));
        p0 = tanf(p0 * (float)(M_PI / 4.0));
      }
      float ccc[3], vvv[3];
      for (int i = 0; i < 3; i++) {
        switch (face) {
          case 0: ccc[i] = (float)(-1.0 * (double)xx[i] + (double)(p1 * yy[i])); vvv[i] = zz[i]; break;
          case 1: ccc[i] = (float)(1.0 * (double)xx[i] + (double)(p1 * yy[i])); vvv[i] = -zz[i]; break;
          case 2: ccc[i] = (float)(-1.0 * (double)yy[i] - (double)(p1 * zz[i])); vvv[i] = -xx[i]; break;
          case 3: ccc[i] = (float)(1.0 * (double)yy[i] + (double)(p1 * zz[i])); vvv[i] = -xx[i]; break;
          case 4: ccc[i] = (float)((double)(p1 * yy[i]) + 1.0 * (double)zz[i]); vvv[i] = xx[i]; break;
          default: ccc[i] = (float)((double)(p1 * yy[i]) - 1.0 * (double)zz[i]); vvv[i] = -xx[i]; break;
        }
        trg[i] = ccc[i] + p0 * vvv[i];
      }
      if (s->normalize) normalize3(trg);
      break;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* source lookup: mount_t / source_t / cubemap_view_t (environment.h)        */
/* ------------------------------------------------------------------------ */

typedef struct {
  const euo_source *src;
  ev_t ev;
  double tex[4], wex[4];       /* total_extent, window_extent */
  float wexf[4];               /* window extent narrowed for the compares */
  float brighten;
} mount_t;

static void mount_init(mount_t *m, const euo_source *src)
{
  m->src = src;
  ev_init(&m->ev, &src->spl, src->spl.degree);
  m->brighten = (float)src->brighten;
  if (src->projection == EUO_CUBEMAP || src->projection == EUO_BIATAN6) return;
  /* environment.h:617-633 (the y terms divide by total_width and add
   * window_width, as the reference does) */
  euo_get_extent(src->projection, src->width, src->height, src->hfov, m->tex);
  double wx = m->tex[1] - m->tex[0];
  double wy = m->tex[3] - m->tex[2];
  double px = (double)src->window_x_offset / src->width;
  double py = (double)src->window_y_offset / src->width;
  m->wex[0] = m->tex[0] + px * wx;
  m->wex[2] = m->tex[2] + py * wy;
  px = (double)(src->window_x_offset + src->window_width) / src->width;
  py = (double)(src->window_y_offset + src->window_width) / src->width;
  m->wex[1] = m->tex[0] + px * wx;
  m->wex[3] = m->tex[2] + py * wy;
  for (int i = 0; i < 4; i++) m->wexf[i] = (float)m->wex[i];
}

/* ray -> source planar coordinate, geometry.h:278-540 */
static void ray_to_planar(int projection, const float *ray, float *crd)
{
  float right = ray[0], down = ray[1], forward = ray[2];
  switch (projection) {
    case EUO_SPHERICAL: {
      float s = sqrtf(right * right + forward * forward);
      crd[1] = atan2f(down, s);
      crd[0] = atan2f(right, forward);
      break;
    }
    case EUO_CYLINDRICAL: {
      float s = sqrtf(right * right + forward * forward);
      crd[1] = down / s;
      crd[0] = atan2f(right, forward);
      break;
    }
    case EUO_RECTILINEAR:
      crd[0] = right / forward;
      crd[1] = down / forward;
      break;
    case EUO_STEREOGRAPHIC: {
      float rn = 1.0f / sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
      float r = right * rn, d = down * rn, f = forward * rn;
      float factor = 2.0f / (f + 1.0f);
      crd[0] = r * factor;
      crd[1] = d * factor;
      break;
    }
    case EUO_FISHEYE: {
      float s = sqrtf(right * right + down * down);
      float r = (float)M_PI_2 - atan2f(forward, s);
      float phi = atan2f(down, right);
      crd[0] = r * cosf(phi);
      crd[1] = r * sinf(phi);
      break;
    }
  }
}

/* pto_planar forward direction, environment.h:240-340 + lens_correction.h */
static void planar_lens(const euo_source *src, float *crd);
static int mount_mask(const mount_t *m, const float *ray);

/* the coordinate half of mount_t::eval: crd3 = { source x, source y, cube face | 0 }; returns
 * the hit mask (crd3 = { 0, 0, -1 } for a miss) */
static int mount_coordinate(const mount_t *m, const float *ray, float *crd3)
{
  const euo_source *src = m->src;
  if (src->projection == EUO_CUBEMAP || src->projection == EUO_BIATAN6) {
    int face;
    float inf[2], pick[2];
    ray_to_cubeface(ray, &face, inf);
    if (src->projection == EUO_BIATAN6) {
      inf[0] = (float)(4.0 / M_PI) * atanf(inf[0]);
      inf[1] = (float)(4.0 / M_PI) * atanf(inf[1]);
    }
    /* environment.h:1452-1460 */
    pick[0] = inf[0] + src->refc_md;
    pick[1] = inf[1] + src->refc_md;
    pick[0] *= src->model_to_px;
    pick[1] *= src->model_to_px;
    pick[1] += (float)(face * src->section_px);
    pick[0] -= .5f;
    pick[1] -= .5f;
    crd3[0] = pick[0]; crd3[1] = pick[1]; crd3[2] = (float)face;
    return 1;
  }
  float crd[2] = { 0.0f, 0.0f };
  ray_to_planar(src->projection, ray, crd);
  if (src->has_lcp) planar_lens(src, crd);
  /* source_t::test_crd, environment.h:970-977: float compares */
  int mask = crd[0] >= m->wexf[0] && crd[0] <= m->wexf[1]
          && crd[1] >= m->wexf[2] && crd[1] <= m->wexf[3];
  if (src->projection == EUO_RECTILINEAR) mask = mask && (ray[2] > 0.0f);
  if (!mask) {
    crd3[0] = crd3[1] = 0.0f; crd3[2] = -1.0f;
    return 0;
  }
  /* md_to_spline, environment.h:988-1006 */
  float ic0 = (float)((double)crd[0] - m->tex[0]);
  ic0 /= (float)(m->tex[1] - m->tex[0]);
  ic0 *= (float)src->width;
  ic0 -= .5f;
  float ic1 = (float)((double)crd[1] - m->tex[2]);
  ic1 /= (float)(m->tex[3] - m->tex[2]);
  ic1 *= (float)src->height;
  ic1 -= .5f;
  crd3[0] = ic0 - (float)src->window_x_offset;
  crd3[1] = ic1 - (float)src->window_y_offset;
  crd3[2] = 0.0f;
  return 1;
}

/* --mask_for: what the facet's evaluator yields instead of the interpolated pixel px (nch floats, in place).
 * masking_t (1 or 3 channels: the paint, unconditionally) or alpha_masking_t (2 or 4: colour = paint * alpha,
 * alpha kept), masking.h:70-135. mask_paint 1: painted black, 2: painted white. Pinned against the reference's
 * own functors through oracle/ref_zimt.cc (ref_masking / ref_alpha_masking). */
void euo_mask_paint(int nch, int mask_paint, float *px)
{
  const float paint = mask_paint == 2 ? 1.0f : 0.0f;
  if (nch == 1 || nch == 3) for (int c = 0; c < nch; c++) px[c] = paint;
  else {
    px[0] = paint * px[nch - 1];
    if (nch == 4) px[1] = px[2] = px[0];
  }
}

/* returns the hit mask; px gets nch floats */
static int mount_eval(const mount_t *m, const float *ray, float *px, float *dbg)
{
  float crd3[3];
  const int mask = mount_coordinate(m, ray, crd3);
  if (dbg) { dbg[0] = crd3[0]; dbg[1] = crd3[1]; dbg[2] = crd3[2]; }
  if (!mask) {
    for (int c = 0; c < m->src->spl.nch; c++) px[c] = 0.0f;
    return 0;
  }
  ev_eval(&m->ev, crd3[0], crd3[1], px);
  if (m->src->mask_paint) euo_mask_paint(m->src->spl.nch, m->src->mask_paint, px);
  return 1;
}

/* repix_t, environment.h:1205-1309: channel-count adaption */
static void repix(int in_n, int out_n, const float *in, float *out)
{
  if (in_n == out_n) { for (int c = 0; c < in_n; c++) out[c] = in[c]; return; }
  if (in_n == 1) {
    if (out_n == 3) { out[0] = out[1] = out[2] = in[0]; }
    else if (out_n == 2) { out[0] = in[0]; out[1] = 1.0f; }
    else { out[0] = out[1] = out[2] = in[0]; out[3] = 1.0f; }
  } else if (in_n == 2) {
    if (out_n == 1 || out_n == 3) {
      float v = in[0] / in[1];
      if (in[1] == 0.0f) v = 0.0f;
      for (int c = 0; c < out_n; c++) out[c] = v;
    } else { out[0] = out[1] = out[2] = in[0]; out[3] = in[1]; }
  } else if (in_n == 3) {
    float sum = in[0]; sum += in[1]; sum += in[2];
    if (out_n == 1) out[0] = sum / 3.0f;
    else if (out_n == 2) { out[0] = sum / 3.0f; out[1] = 1.0f; }
    else { out[0] = in[0]; out[1] = in[1]; out[2] = in[2]; out[3] = 1.0f; }
  } else {
    if (out_n == 1) {
      float v = (in[0] + in[1] + in[2]) / 3.0f;
      v /= in[3];
      if (in[3] == 0.0f) v = 0.0f;
      out[0] = v;
    } else if (out_n == 2) { out[0] = (in[0] + in[1] + in[2]) / 3.0f; out[1] = in[3]; }
    else {
      for (int c = 0; c < 3; c++) out[c] = in[c] / in[3];
      if (in[3] == 0.0f) out[0] = out[1] = out[2] = 0.0f;
    }
  }
}

/* mono_t, environment.h:1325-1383: the channel adaption of masking jobs (1 or 2 output channels:
 * the colour channels of a mask are equal, so one stands for all; alpha stays associated).
 * Returns 0 for the combinations the reference asserts on. */
static int mono(int in_n, int out_n, const float *in, float *out)
{
  if (in_n == out_n) { for (int c = 0; c < in_n; c++) out[c] = in[c]; return 1; }
  if (out_n != 1 && out_n != 2) return 0;
  if (in_n == 1) { out[0] = in[0]; out[1] = 1.0f; }
  else if (in_n == 2) { out[0] = in[0] / in[1]; if (in[1] == 0.0f) out[0] = 0.0f; }
  else if (in_n == 3) { out[0] = in[0]; if (out_n == 2) out[1] = 1.0f; }
  else {
    if (out_n == 1) { out[0] = in[0]; out[0] /= in[3]; if (in[3] == 0.0f) out[0] = 0.0f; }
    else { out[0] = in[0]; out[1] = in[3]; }
  }
  return 1;
}

/* environment::eval (environment.h:1821-1842) for a target with out_n channels:
 * inner evaluation, channel adaption, then brighten on the OUTPUT layout */
static void env_eval(const mount_t *m, const float *ray, int out_n, float *px, float *dbg)
{
  float raw[4];
  int in_n = m->src->spl.nch;
  mount_eval(m, ray, raw, dbg);
  if (m->src->mask_paint) mono(in_n, out_n, raw, px);      /* environment.h:1909-1957 */
  else repix(in_n, out_n, raw, px);
  if (m->brighten != 1.0f) {
    int ncol = (out_n == 2 || out_n == 4) ? out_n - 1 : out_n;
    for (int c = 0; c < ncol; c++) px[c] *= m->brighten;
  }
}

/* lcp<float>::eval (lens_correction.h:224-235 on eu_polynomial::function, :93-105):
 * coefficients {a, b, c, 1 - (a + b + c)} formed in float from the facet's doubles, the sum
 * runs from the constant term up with `power *= x` behind every term. Pinned against the
 * reference's own lcp (oracle/ref_zimt.cc: ref_lcp_factor; tests/golden). */
static float lens_factor(double da, double db, double dc, float x)
{
  float a = (float)da, b = (float)db, c = (float)dc;
  float d = 1.0f - (a + b + c);
  float sum = 0.0f, power = 1.0f;
  sum += d * power; power *= x;
  sum += c * power; power *= x;
  sum += b * power; power *= x;
  sum += a * power; power *= x;
  return sum;
}

void euo_lens_factor(double a, double b, double c, const float *x, long n, float *out)
{
  for (long i = 0; i < n; i++) out[i] = lens_factor(a, b, c, x[i]);
}

/* lens polynomial + shift + shear, PTO forward direction
 * (environment.h:254-284; lens_correction.h:93-105, :224-235). The flags and
 * the scale s come from process_geometry (envutil_basic.h:499-521). */
static void planar_lens(const euo_source *src, float *crd)
{
  double ext[4];
  euo_get_extent(src->projection, src->width, src->height, src->hfov, ext);
  double dv = fabs(ext[3] - ext[2]) / 2.0, dh = fabs(ext[1] - ext[0]) / 2.0;
  double sd = (dh < dv) ? dh : dv;
  int has_lcp = (src->a != 0.0 || src->b != 0.0 || src->c != 0.0);
  int has_shift = (src->h != 0.0 || src->v != 0.0);
  int has_shear = (src->shear_g != 0.0 || src->shear_t != 0.0);
  float o0 = crd[0], o1 = crd[1];
  if (has_lcp) {
    float sqn = crd[0] * crd[0];
    sqn += crd[1] * crd[1];
    float x = sqrtf(sqn) / (float)sd;
    float sum = lens_factor(src->a, src->b, src->c, x);
    o0 *= sum; o1 *= sum;
  }
  if (has_shift) { o0 += (float)src->h; o1 += (float)src->v; }
  if (has_shear) {
    /* vec<float> * double -> double (A.0) */
    float h0 = (float)((double)o0 + ((double)o1 * src->shear_g));
    float h1 = (float)((double)o1 + ((double)o0 * src->shear_t));
    o0 = h0; o1 = h1;
  }
  crd[0] = o0; crd[1] = o1;
}

/* mount_t::get_mask (environment.h:1151-1159): does the ray hit the facet? */
static int mount_mask(const mount_t *m, const float *ray)
{
  const euo_source *src = m->src;
  if (src->projection == EUO_CUBEMAP || src->projection == EUO_BIATAN6) return 1;
  if (src->projection == EUO_FISHEYE && src->hfov >= M_PI * 2.0) return 1;   /* environment.h:1747-1749 */
  float crd[2] = { 0.0f, 0.0f };
  ray_to_planar(src->projection, ray, crd);
  if (src->has_lcp) planar_lens(src, crd);
  int mask = crd[0] >= m->wexf[0] && crd[0] <= m->wexf[1]
          && crd[1] >= m->wexf[2] && crd[1] <= m->wexf[3];
  if (src->projection == EUO_RECTILINEAR) mask = mask && (ray[2] > 0.0f);
  return mask;
}

/* ------------------------------------------------------------------------ */
/* the hot loop: zimt::process(shape, stepper, environment | twine_t, storer) */
/* (envutil_payload.cc:425-579, :2118-2232; zimt/wielding.h:151-463;         */
/*  twining.h:128-263)                                                       */
/* ------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------ */
/* multi-facet rendering: fusion_t (zimt/get.h:1181-1242) + synopsis          */
/* (envutil_payload.cc:587-691 synopsis_t, :762-957 _voronoi_syn,             */
/*  :964-1233 _voronoi_syn_plus). The synopsis objects take some decisions per */
/* 16-lane VECTOR (any_of / all_of); the oracle therefore works on groups of   */
/* EUO_LANES pixels that start at segment boundaries, like zimt::process.      */
/* ------------------------------------------------------------------------ */

#define EUO_MAX_FACETS 256

typedef struct {
  int nfct, nch, plus;
  int hdr, hdr_low, hdr_high;          /* _hdr_merge_syn: facet that rules the shadows / the highlights */
  mount_t mnt[EUO_MAX_FACETS];
  float recip_step[EUO_MAX_FACETS];
  float optimum[EUO_MAX_FACETS];       /* 0.5f * brighten, envutil_payload.cc:1364 */
} syn_t;

/* _hdr_merge_syn::get_quality for a grey value (envutil_payload.cc:1388-1446): kind 0 LOW, 1 MIDDLE, 2 HIGH */
static float hdr_quality(float grey, float optimum, int kind)
{
  int grey_is_large = grey > optimum;
  float distance = fabsf(optimum - grey);
  if (kind == 0 && !grey_is_large) distance = 0.0f;
  if (kind == 2 && grey_is_large) distance = 0.0f;
  float proximity = optimum - distance;
  return proximity / (optimum * optimum);
}
static float std_max(float a, float b) { return a < b ? b : a; }   /* std::max, broadcast per lane */

/* _hdr_merge_syn::operator() (envutil_payload.cc:1500-1622): EVERY facet is evaluated (a miss is a
 * zero pixel and takes part with the quality of a zero pixel), quality-weighted sum, normalised */
static void synopsis_hdr(const syn_t *sy, float rays[][EUO_LANES][3], int n, float px[][4])
{
  int nf = sy->nfct, nch = sy->nch;
  float qsum[EUO_LANES];
  for (int l = 0; l < n; l++) { qsum[l] = 0.0f; for (int c = 0; c < 4; c++) px[l][c] = 0.0f; }
  for (int f = 0; f < nf; f++) {
    float v[EUO_LANES][4];
    int kind = f == sy->hdr_low ? 0 : (f == sy->hdr_high ? 2 : 1);
    float opt = sy->optimum[f];
    for (int l = 0; l < n; l++) env_eval(&sy->mnt[f], rays[f][l], nch, v[l], NULL);
    int all_transparent = 1;
    if (nch == 2 || nch == 4)
      for (int l = 0; l < n; l++) if (!(v[l][nch - 1] == 0.0f)) all_transparent = 0;
    for (int l = 0; l < n; l++) {
      float q;
      if (nch == 1) q = hdr_quality(v[l][0], opt, kind);
      else if (nch == 3) q = hdr_quality(std_max(v[l][0], std_max(v[l][1], v[l][2])), opt, kind);
      else {
        float grey = nch == 2 ? v[l][0] : std_max(v[l][0], std_max(v[l][1], v[l][2]));
        q = all_transparent ? 0.0f : v[l][nch - 1] * hdr_quality(grey, opt, kind);
      }
      qsum[l] += q;
      if (nch == 1 || nch == 3) {
        for (int c = 0; c < nch; c++) px[l][c] += v[l][c] * q;
      } else {
        float a = v[l][nch - 1];
        for (int c = 0; c < nch - 1; c++) {
          float d = 0.0f;
          if (a > 0.000001f) d = v[l][c] / a;
          px[l][c] += d * q;
        }
        px[l][nch - 1] = std_max(px[l][nch - 1], a);
      }
    }
  }
  for (int l = 0; l < n; l++) {
    int ncol = (nch == 2 || nch == 4) ? nch - 1 : nch;
    for (int c = 0; c < ncol; c++) {
      px[l][c] /= qsum[l];
      if (!(qsum[l] > 0.0f)) px[l][c] = 0.0f;
      if (ncol != nch) px[l][c] *= px[l][nch - 1];
    }
  }
}

/* rays[f][lane][3] -> px[lane][nch] for n valid lanes */
static void synopsis_group(const syn_t *sy, float rays[][EUO_LANES][3], int n, float px[][4])
{
  int nf = sy->nfct, nch = sy->nch;
  if (sy->hdr) { synopsis_hdr(sy, rays, n, px); return; }
  if (!sy->plus) {
    /* _voronoi_syn: champion = facet with the largest z * recip_step among the
     * facets the ray hits; strict '>' keeps the earlier facet on ties */
    for (int l = 0; l < n; l++) {
      int champ = -1;
      float max_z = -FLT_MAX;
      for (int f = 0; f < nf; f++) {
        if (!mount_mask(&sy->mnt[f], rays[f][l])) continue;
        float z = rays[f][l][2] * sy->recip_step[f];
        if (champ == -1 && f == 0) { champ = 0; max_z = z; continue; }
        float cur = z;
        if (cur > max_z) { max_z = cur; champ = f; }
      }
      for (int c = 0; c < nch; c++) px[l][c] = 0.0f;
      if (champ >= 0) {
        env_eval(&sy->mnt[champ], rays[champ][l], nch, px[l], NULL);
      }
    }
    return;
  }
  /* _voronoi_syn_plus: per lane a list of (z, facet) sorted by z, one slot per
   * facet that is valid for ANY lane of the vector */
  static _Thread_local float maxz[EUO_MAX_FACETS][EUO_LANES];
  static _Thread_local int champ[EUO_MAX_FACETS][EUO_LANES];
  static _Thread_local float lv[EUO_MAX_FACETS][EUO_LANES][4];
  int layers = 0, next_best = -1;
  for (int f = 0; f < nf; f++) {
    int valid[EUO_LANES], any = 0;
    for (int l = 0; l < n; l++) { valid[l] = mount_mask(&sy->mnt[f], rays[f][l]); any |= valid[l]; }
    if (f == 0) {
      for (int l = 0; l < n; l++) { maxz[0][l] = -FLT_MAX; champ[0][l] = -1; }
      if (any) {
        next_best = 0; layers = 1;
        for (int l = 0; l < n; l++) if (valid[l]) { champ[0][l] = 0; maxz[0][l] = rays[0][l][2] * sy->recip_step[0]; }
      }
      continue;
    }
    if (!any) continue;
    next_best = f;
    for (int l = 0; l < n; l++) {
      maxz[layers][l] = valid[l] ? rays[f][l][2] * sy->recip_step[f] : -FLT_MAX;
      champ[layers][l] = valid[l] ? f : -1;
    }
    for (int k = layers; k > 0; --k) {
      int swapped = 0;
      for (int l = 0; l < n; l++)
        if (maxz[k][l] > maxz[k - 1][l]) {
          float tz = maxz[k][l]; maxz[k][l] = maxz[k - 1][l]; maxz[k - 1][l] = tz;
          int tc = champ[k][l]; champ[k][l] = champ[k - 1][l]; champ[k - 1][l] = tc;
          swapped = 1;
        }
      if (!swapped) break;
    }
    ++layers;
  }
  for (int l = 0; l < n; l++) for (int c = 0; c < nch; c++) px[l][c] = 0.0f;
  if (layers == 0) return;
  int all_top = 1;
  for (int l = 0; l < n; l++) if (champ[0][l] != next_best) all_top = 0;
  if (all_top) {
    /* one facet on top everywhere and fully opaque: take it as is */
    float help[EUO_LANES][4];
    int opaque = 1;
    for (int l = 0; l < n; l++) {
      env_eval(&sy->mnt[next_best], rays[next_best][l], nch, help[l], NULL);
      if (!(help[l][nch - 1] >= 1.0f)) opaque = 0;
    }
    if (opaque) {
      for (int l = 0; l < n; l++) for (int c = 0; c < nch; c++) px[l][c] = help[l][c];
      return;
    }
  }
  for (int f = 0; f < nf; f++)
    for (int l = 0; l < n; l++) {
      env_eval(&sy->mnt[f], rays[f][l], nch, lv[f][l], NULL);
    }
  for (int i = 0; i < layers; i++)
    for (int l = 0; l < n; l++) {
      int ci = champ[i][l];
      if (ci == -1) continue;
      const float *help = lv[ci][l];
      if (i == 0) for (int c = 0; c < nch; c++) px[l][c] = help[c];
      else for (int c = 0; c < nch; c++) px[l][c] += (1.0f - px[l][nch - 1]) * help[c];
    }
}

/* ------------------------------------------------------------------------ */
/* to_screen_t (envutil_payload.cc:225-413): lut_based_tf = degree-1          */
/* bspline<float,1>(256, NATURAL) over 255 * RGB2sRGB(i / 255.0), evaluated   */
/* through make_safe_evaluator (clamp gate [0, 255], eval.h:2096-2104) at     */
/* in * 255.0f; the float results convert to uint32 by truncation and pack.   */
/* ------------------------------------------------------------------------ */

void euo_screen_lut(float *lut)
{
  for (int i = 0; i < 256; i++) {
    double x = i / (double)(256 - 1);
    double r = 1.055 * pow(x, 0.41666666666666667) - 0.055;
    if (x <= 0.0031308) r = 12.92 * x;
    double y = r * 255.0;
    lut[i] = (float)y;
  }
  /* NATURAL brace, one coefficient to the right (brace.h:134+): point mirror */
  lut[256] = lut[255] + lut[255] - lut[254];
}

float euo_lut_eval(const float *lut, float in)
{
  float c = in * (float)(256 - 1);
  if (c < 0.0f) c = 0.0f;            /* clamp_gate, map.h:184-231 */
  else if (c > 255.0f) c = 255.0f;
  if (c != c) return 0.0f;           /* NaN: undefined in the reference */
  float fl = floorf(c);
  float t = c - fl;
  int i = (int)fl;
  float wl = 1.0f - t, wr = t;       /* _eval_linear<0>, eval.h:1040-1059 */
  float sum = lut[i];
  sum *= wl;
  sum += lut[i + 1] * wr;
  return sum;
}

static unsigned screen_channel(const float *lut, float in)
{
  return (unsigned)euo_lut_eval(lut, in);
}

unsigned euo_to_screen(const float *lut, int nch, const float *px)
{
  unsigned out;
  if (nch == 1) {
    unsigned ch = screen_channel(lut, px[0]);
    out = 0xFF00; out |= ch; out <<= 8; out |= ch; out <<= 8; out |= ch;
  } else if (nch == 2) {
    unsigned c1 = screen_channel(lut, px[0]), c2 = screen_channel(lut, px[1]);
    out = c2; out <<= 8; out |= c1; out <<= 8; out |= c1; out <<= 8; out |= c1;
  } else if (nch == 3) {
    unsigned c1 = screen_channel(lut, px[0]), c2 = screen_channel(lut, px[1]),
             c3 = screen_channel(lut, px[2]);
    out = 0xFF00; out |= c3; out <<= 8; out |= c2; out <<= 8; out |= c1;
  } else {
    unsigned c1 = screen_channel(lut, px[0]), c2 = screen_channel(lut, px[1]),
             c3 = screen_channel(lut, px[2]), c4 = screen_channel(lut, px[3]);
    out = c4; out <<= 8; out |= c3; out <<= 8; out |= c2; out <<= 8; out |= c1;
  }
  return out;
}

static int euo_render_multi(const euo_job *job, const euo_source *srcs, int nsrc,
                            float *out, long ors)
{
  int nch = job->nch, W = job->crop_w > 0 ? job->crop_w : job->width;
  float lut[257];
  euo_screen_lut(lut);
  if (nsrc > EUO_MAX_FACETS) return -2;
  syn_t *sy = (syn_t *)calloc(1, sizeof(syn_t));
  stepper_t *st = (stepper_t *)calloc(3 * (size_t)nsrc, sizeof(stepper_t));
  sy->nfct = nsrc; sy->nch = nch; sy->plus = (nch == 2 || nch == 4);
  static _Thread_local inv_lcp_t inv_model;
  if (!single_inv_model(job, &inv_model)) { free(sy); free(st); return -5; }
  double r_cam[9];
  euo_make_r3(job->roll, job->pitch, job->yaw, 0, r_cam);
  for (int f = 0; f < nsrc; f++) {
    /* a facet with another channel count goes through repix_t inside its
     * environment object (environment.h:1846-1900) */
    double r_fct[9], basis[9];
    euo_make_r3(srcs[f].roll, srcs[f].pitch, srcs[f].yaw, 1, r_fct);
    euo_rotate_r3(r_cam, r_fct, basis);
    /* multi-facet steppers are built with normalize = true (payload.cc:2152) */
    stepper_init(&st[3 * f], job->projection, 1, basis, job->width, job->height, job->x0, job->x1, job->y0, job->y1, 0.0f, 0.0f);
    stepper_init(&st[3 * f + 1], job->projection, 1, basis, job->width, job->height, job->x0, job->x1, job->y0, job->y1, 0.25f, 0.0f);
    stepper_init(&st[3 * f + 2], job->projection, 1, basis, job->width, job->height, job->x0, job->x1, job->y0, job->y1, 0.0f, 0.25f);
    if (job->crop_w > 0)
      for (int v = 0; v < 3; v++) { st[3 * f + v].x_off = job->crop_x0; st[3 * f + v].y_off = job->crop_y0; }
    if (has_translation(&srcs[f]) || generic_target(job))
      for (int v = 0; v < 3; v++)
        if (!stepper_make_generic(&st[3 * f + v], job, &srcs[f], &inv_model)) { free(sy); free(st); return -4; }
    mount_init(&sy->mnt[f], &srcs[f]);
    sy->recip_step[f] = (float)(1.0 / srcs[f].step);
    sy->optimum[f] = 0.5f * (float)srcs[f].brighten;
  }
  sy->hdr = job->synopsis == 1;
  {
    /* _hdr_merge_syn ctor (envutil_payload.cc:1346-1376): first strict minimum / maximum of brighten */
    float lowest = 100000.0f, highest = -1.0f;
    sy->hdr_low = sy->hdr_high = -1;
    for (int f = 0; f < nsrc; f++) {
      float b = (float)srcs[f].brighten;
      if (b < lowest) { lowest = b; sy->hdr_low = f; }
      if (b > highest) { highest = b; sy->hdr_high = f; }
    }
  }
  int twining = job->ntaps > 0;
  float *taps = NULL;
  if (twining) {
    taps = (float *)malloc(sizeof(float) * 3 * (size_t)job->ntaps);
    for (int k = 0; k < job->ntaps; k++) {
      taps[3 * k] = job->taps[3 * k] * 4.0f;
      taps[3 * k + 1] = job->taps[3 * k + 1] * 4.0f;
      taps[3 * k + 2] = job->taps[3 * k + 2];
    }
  }
  int nthreads = job->nthreads > 0 ? job->nthreads : 1;
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads)
  for (int y = job->row_begin; y < job->row_end; y++) {
    float *row = out + (long)(y - job->row_begin) * ors;
    float (*rays)[EUO_LANES][3] = malloc(sizeof(float) * (size_t)nsrc * EUO_LANES * 3);
    float (*p0)[EUO_LANES][3] = malloc(sizeof(float) * (size_t)nsrc * EUO_LANES * 3);
    float (*du)[EUO_LANES][3] = malloc(sizeof(float) * (size_t)nsrc * EUO_LANES * 3);
    float (*dv)[EUO_LANES][3] = malloc(sizeof(float) * (size_t)nsrc * EUO_LANES * 3);
    for (int seg = 0; seg < W; seg += EUO_SEGMENT)
      for (int x0 = seg; x0 < W && x0 < seg + EUO_SEGMENT; x0 += EUO_LANES) {
        int n = W - x0 < EUO_LANES ? W - x0 : EUO_LANES;
        float px[EUO_LANES][4];
        if (!twining) {
          for (int f = 0; f < nsrc; f++)
            for (int l = 0; l < n; l++) stepper_ray(&st[3 * f], x0 + l, y, rays[f][l]);
          synopsis_group(sy, rays, n, px);
        } else {
          /* synopsis_t::operator() for ninepacks (payload.cc:647-690) */
          for (int f = 0; f < nsrc; f++)
            for (int l = 0; l < n; l++) {
              float r10[3], r01[3];
              stepper_ray(&st[3 * f], x0 + l, y, p0[f][l]);
              stepper_ray(&st[3 * f + 1], x0 + l, y, r10);
              stepper_ray(&st[3 * f + 2], x0 + l, y, r01);
              for (int i = 0; i < 3; i++) { du[f][l][i] = r10[i] - p0[f][l][i]; dv[f][l][i] = r01[i] - p0[f][l][i]; }
            }
          float acc[EUO_LANES][4];
          memset(acc, 0, sizeof acc);
          for (int k = 0; k < job->ntaps; k++) {
            for (int f = 0; f < nsrc; f++)
              for (int l = 0; l < n; l++)
                for (int i = 0; i < 3; i++)
                  rays[f][l][i] = p0[f][l][i] + taps[3 * k] * du[f][l][i] + taps[3 * k + 1] * dv[f][l][i];
            float help[EUO_LANES][4];
            synopsis_group(sy, rays, n, help);
            for (int l = 0; l < n; l++)
              for (int c = 0; c < nch; c++) acc[l][c] += taps[3 * k + 2] * help[l][c];
          }
          memcpy(px, acc, sizeof acc);
        }
        for (int l = 0; l < n; l++) {
          if (job->screen) { ((unsigned *)row)[x0 + l] = euo_to_screen(lut, nch, px[l]); continue; }
          for (int c = 0; c < nch; c++) row[(long)(x0 + l) * nch + c] = px[l][c];
        }
      }
    free(rays); free(p0); free(du); free(dv);
  }
  free(taps); free(sy); free(st);
  return 0;
}

/* mount_t::get_coordinate / cubemap_view_t pickup for caller-supplied rays (tests of the edge
 * cases of ray_to_cubeface and the mounts: exact ties, signed zeros, axis-aligned rays):
 * out3 = { source x, source y, cube face | 0 } and { 0, 0, -1 } for a miss */
void euo_source_coordinates(const euo_source *src, const float *rays, long n, float *out3)
{
  mount_t mnt;
  mount_init(&mnt, src);
  for (long i = 0; i < n; i++) mount_coordinate(&mnt, rays + 3 * i, out3 + 3 * i);
}

int euo_render(const euo_job *job, const euo_source *srcs, int nsrc,
               float *out, long ors)
{
  if (nsrc > 1) return euo_render_multi(job, srcs, nsrc, out, ors);
  if (nsrc != 1) return -2;
  const euo_source *src = &srcs[0];
  int nch = job->nch;
  double r_cam[9], r_fct[9], basis[9];
  euo_make_r3(job->roll, job->pitch, job->yaw, 0, r_cam);
  euo_make_r3(src->roll, src->pitch, src->yaw, 1, r_fct);
  euo_rotate_r3(r_cam, r_fct, basis);
  int twining = job->ntaps > 0;
  stepper_t st00, st10, st01;
  /* single facet, no twining: normalize = false (payload.cc:2118);
   * deriv_stepper: normalize = true, bias .25 (payload.cc:2227, stepper.h:1617) */
  stepper_init(&st00, job->projection, twining, basis, job->width, job->height,
               job->x0, job->x1, job->y0, job->y1, 0.0f, 0.0f);
  stepper_init(&st10, job->projection, 1, basis, job->width, job->height,
               job->x0, job->x1, job->y0, job->y1, 0.25f, 0.0f);
  stepper_init(&st01, job->projection, 1, basis, job->width, job->height,
               job->x0, job->x1, job->y0, job->y1, 0.0f, 0.25f);
  if (job->crop_w > 0) {
    st00.x_off = st10.x_off = st01.x_off = job->crop_x0;
    st00.y_off = st10.y_off = st01.y_off = job->crop_y0;
  }
  inv_lcp_t inv_model;
  if (!single_inv_model(job, &inv_model)) return -5;
  if (has_translation(src) || generic_target(job)) {
    /* generic_stepper<float, LANES, false> / deriv_stepper<..., generic_stepper, true> (payload.cc:2095-2110, :2214-2224) */
    if (!stepper_make_generic(&st00, job, src, &inv_model) || !stepper_make_generic(&st10, job, src, &inv_model) ||
        !stepper_make_generic(&st01, job, src, &inv_model)) return -4;
  }
  float lut[257];
  euo_screen_lut(lut);
  mount_t mnt;
  mount_init(&mnt, src);
  /* twine_t ctor: x and y of every tap pre-multiplied by 4 (twining.h:106-121) */
  float *taps = NULL;
  if (twining) {
    taps = (float *)malloc(sizeof(float) * 3 * (size_t)job->ntaps);
    for (int k = 0; k < job->ntaps; k++) {
      taps[3 * k + 0] = job->taps[3 * k + 0] * 4.0f;
      taps[3 * k + 1] = job->taps[3 * k + 1] * 4.0f;
      taps[3 * k + 2] = job->taps[3 * k + 2];
    }
  }
  int nthreads = job->nthreads > 0 ? job->nthreads : 1;
  int W = job->crop_w > 0 ? job->crop_w : job->width;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
  for (int y = job->row_begin; y < job->row_end; y++) {
    float *row = out + (long)(y - job->row_begin) * ors;
    for (int x = 0; x < W; x++) {
      float ray[3], px[4], dbg[3];
      stepper_ray(&st00, x, y, ray);
      if (job->stage == 1) {
        row[3 * x] = ray[0]; row[3 * x + 1] = ray[1]; row[3 * x + 2] = ray[2];
        continue;
      }
      if (!twining) {
        env_eval(&mnt, ray, nch, px, dbg);
        if (job->stage == 2) {
          row[3 * x] = dbg[0]; row[3 * x + 1] = dbg[1]; row[3 * x + 2] = dbg[2];
          continue;
        }
        if (job->screen) { ((unsigned *)row)[x] = euo_to_screen(lut, nch, px); continue; }
        for (int c = 0; c < nch; c++) row[(long)x * nch + c] = px[c];
      } else {
        float r10[3], r01[3], dx[3], dy[3], acc[4] = { 0, 0, 0, 0 };
        stepper_ray(&st10, x, y, r10);
        stepper_ray(&st01, x, y, r01);
        for (int i = 0; i < 3; i++) { dx[i] = r10[i] - ray[i]; dy[i] = r01[i] - ray[i]; }
        for (int k = 0; k < job->ntaps; k++) {
          float in_k[3];
          for (int i = 0; i < 3; i++)
            in_k[i] = ray[i] + taps[3 * k] * dx[i] + taps[3 * k + 1] * dy[i];
          env_eval(&mnt, in_k, nch, px, NULL);
          for (int c = 0; c < nch; c++) acc[c] += taps[3 * k + 2] * px[c];
        }
        if (job->screen) { ((unsigned *)row)[x] = euo_to_screen(lut, nch, acc); continue; }
        for (int c = 0; c < nch; c++) row[(long)x * nch + c] = acc[c];
      }
    }
  }
  free(taps);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* double-precision projection functors for the reference's own property     */
/* tests (geometry.h:152-560; geometry.cc:283-420)                           */
/* ------------------------------------------------------------------------ */

void euo_prj_to_ray_d(int projection, const double *in, double *out)
{
  switch (projection) {
    case EUO_SPHERICAL: {
      double sl = sin(in[1]), cl = cos(in[1]), so = sin(in[0]), co = cos(in[0]);
      out[0] = so * cl; out[2] = co * cl; out[1] = sl;
      break;
    }
    case EUO_CYLINDRICAL:
      out[2] = cos(in[0]); out[0] = sin(in[0]); out[1] = in[1];
      break;
    case EUO_RECTILINEAR:
      out[0] = in[0]; out[1] = in[1]; out[2] = 1.0;
      break;
    case EUO_STEREOGRAPHIC: {
      double r = sqrt(in[0] * in[0] + in[1] * in[1]);
      double theta = atan(r / 2.0) * 2.0;
      double phi = atan2(in[0], -in[1]);
      out[2] = cos(theta);
      out[1] = -sin(theta) * cos(phi);
      out[0] = sin(theta) * sin(phi);
      break;
    }
    case EUO_FISHEYE: {
      double r = sqrt(in[0] * in[0] + in[1] * in[1]);
      double phi = atan2(in[0], -in[1]);
      out[2] = cos(r);
      out[1] = -sin(r) * cos(phi);
      out[0] = sin(r) * sin(phi);
      break;
    }
  }
}

void euo_ray_to_prj_d(int projection, const double *in, double *out)
{
  double right = in[0], down = in[1], forward = in[2];
  switch (projection) {
    case EUO_SPHERICAL: {
      double s = sqrt(right * right + forward * forward);
      out[1] = atan2(down, s); out[0] = atan2(right, forward);
      break;
    }
    case EUO_CYLINDRICAL: {
      double s = sqrt(right * right + forward * forward);
      out[1] = down / s; out[0] = atan2(right, forward);
      break;
    }
    case EUO_RECTILINEAR:
      out[0] = right / forward; out[1] = down / forward;
      break;
    case EUO_STEREOGRAPHIC: {
      double rn = 1.0 / sqrt(in[0] * in[0] + in[1] * in[1] + in[2] * in[2]);
      double r = right * rn, d = down * rn, f = forward * rn;
      double factor = 2.0 / (f + 1.0);
      out[0] = r * factor; out[1] = d * factor;
      break;
    }
    case EUO_FISHEYE: {
      double s = sqrt(right * right + down * down);
      double r = M_PI_2 - atan2(forward, s);
      double phi = atan2(down, right);
      out[0] = r * cos(phi); out[1] = r * sin(phi);
      break;
    }
  }
}


/* ---------------------------------------------------------------------------
 * PTO exclude masks and lens crops: the alpha plane of a facet
 * (environment.h:727-843). Restated loop for loop.
 * ------------------------------------------------------------------------- */

/* zimt/extrapolate.h:141-155 */
static float fir_reflect(const float *buf, long stride, int n, int i)
{
  if (i < 0) i = -1 - i;
  if (i >= n) {
    i %= 2 * n;
    if (i >= n) i = 2 * n - 1 - i;
  }
  return buf[(long)i * stride];
}

/* fir_filter::solve, zimt/convolve.h:240-383, for the 5-tap kernel with headroom 2: a
 * circular buffer of the last five samples, the kernel stored twice so that a pointer into
 * it lines the weights up with the slots; the slots are summed in slot order. `in` is a
 * private copy of the line (the filter driver buffers lines), `out` strided. */
static void fir_line(const float *in, int n, float *out, long ostride)
{
  enum { K = 5, HEAD = 2 };
  const double kd[K] = { 1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0 };
  float circ[K], kern[2 * K], tail[K];
  for (int i = 0; i < K; i++) kern[i] = kern[i + K] = (float)(long double)kd[i];
  int si = -HEAD, ti = 0;
  for (int i = 0; i < K; i++, si++) circ[i] = fir_reflect(in, 1, n, si);
  for (int i = 0, z = n; i < K - HEAD; i++, z++) tail[i] = fir_reflect(in, 1, n, z);
  while (ti < n) {
    const float *pk = kern + K;
    float *pd = circ;
    for (int i = 0; i < K && ti < n; i++) {
      float result = circ[0] * pk[0];
      for (int j = 1; j < K; j++) result += circ[j] * pk[j];
      out[(long)ti * ostride] = result;
      *pd = si < n ? in[si] : tail[si - n];
      ++si; ++ti; ++pd; --pk;
    }
  }
}

void euo_binomial_plane(float *plane, int w, int h)
{
  int m = w > h ? w : h;
  float *line = (float *)malloc((size_t)m * sizeof(float));
  for (int y = 0; y < h; y++) {                 /* axis 0 */
    memcpy(line, plane + (long)y * w, (size_t)w * sizeof(float));
    fir_line(line, w, plane + (long)y * w, 1);
  }
  for (int x = 0; x < w; x++) {                 /* axis 1 */
    for (int y = 0; y < h; y++) line[y] = plane[(long)y * w + x];
    fir_line(line, h, plane + x, w);
  }
  free(line);
}

/* fill_polygon, envutil_basic.cc:236-320 */
static void fill_polygon_clear(float *alpha, int w, const float *px, const float *py, int N,
                               int LEFT, int TOP, int RIGHT, int BOT)
{
  int *nodeX = (int *)malloc((size_t)(N + 1) * sizeof(int)), *dir = (int *)malloc((size_t)(N + 1) * sizeof(int));
  for (int pixelY = TOP; pixelY < BOT; pixelY++) {
    int nodes = 0, j = N - 1, i, swap;
    for (i = 0; i < N; i++) {
      int cross = 0;
      if (py[i] < (float)pixelY && py[j] >= (float)pixelY) cross = 1;
      else if (py[j] < (float)pixelY && py[i] >= (float)pixelY) cross = -1;
      if (cross) {
        nodeX[nodes] = (int)(px[i] + (pixelY - py[i]) / (py[j] - py[i]) * (px[j] - px[i]));
        dir[nodes++] = cross;
      }
      j = i;
    }
    i = 0;
    while (i < nodes - 1) {
      if (nodeX[i] > nodeX[i + 1]) {
        swap = nodeX[i]; nodeX[i] = nodeX[i + 1]; nodeX[i + 1] = swap;
        swap = dir[i]; dir[i] = dir[i + 1]; dir[i + 1] = swap;
        if (i) i--;
      } else i++;
    }
    int w_ord = 0;
    for (i = 0; i < nodes; i++) {
      w_ord += dir[i];
      if (!w_ord) continue;
      if (i + 1 >= nodes) break;
      if (nodeX[i] >= RIGHT) break;
      if (nodeX[i + 1] > LEFT) {
        if (nodeX[i] < LEFT) nodeX[i] = LEFT;
        if (nodeX[i + 1] > RIGHT) nodeX[i + 1] = RIGHT;
        for (int pixelX = nodeX[i]; pixelX < nodeX[i + 1]; pixelX++) alpha[(long)pixelY * w + pixelX] = 0.0f;
      }
    }
  }
  free(nodeX); free(dir);
}

void euo_facet_alpha(float *alpha, int w, int h, int npolys, const int *counts, const float *xs,
                     const float *ys, int crop_kind, int cx0, int cx1, int cy0, int cy1, int stage)
{
  for (long i = 0; i < (long)w * h; i++) alpha[i] = 1.0f;
  long off = 0;
  for (int p = 0; p < npolys; p++) {
    fill_polygon_clear(alpha, w, xs + off, ys + off, counts[p], 0, 0, w, h);
    off += counts[p];
  }
  if (crop_kind == 2) {
    float a = (float)(fabs((double)(cx1 - cx0)) / 2.0);
    float b = (float)(fabs((double)(cy1 - cy0)) / 2.0);
    float mx = (float)((cx0 + cx1) / 2.0);
    float my = (float)((cy0 + cy1) / 2.0);
    for (int y = 0; y < h; y++) {
      float dy = fabsf(y - my);
      if (dy > b) {
        for (int x = 0; x < w; x++) alpha[(long)y * w + x] = 0;
        continue;
      }
      float xmargin = (float)sqrt((a * a) * (1.0 - (dy * dy) / (b * b)));
      for (int x = 0; x < w; x++) {
        float dx = fabsf(x - mx);
        if (dx > xmargin) alpha[(long)y * w + x] = 0;
      }
    }
  } else if (crop_kind == 1) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        if (x < cx0 || x >= cx1 || y < cy0 || y >= cy1) alpha[(long)y * w + x] = 0;
  }
  if (stage >= 1) euo_binomial_plane(alpha, w, h);
}
