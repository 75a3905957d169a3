// Host-side set-up arithmetic of the render path: everything envutil computes
// once per job on the CPU before zimt::process runs. Plain C++, no HIP.
// Each function names the reference lines it stands in for.
#ifndef EU_SETUP_MATH_H
#define EU_SETUP_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "eu_device.h"

namespace eu {

// ---- extents: envutil_basic.cc:49-229 -------------------------------------

inline double get_vfov(int prj, int width, int height, double hfov)
{
  switch (prj) {
    case EU_RECTILINEAR:
      return 2.0 * std::atan(height * std::tan(hfov / 2.0) / width);
    case EU_CYLINDRICAL: {
      double pixels_per_rad = width / hfov;
      double h_rad = height / pixels_per_rad;
      return 2.0 * std::atan(h_rad / 2.0);
    }
    case EU_STEREOGRAPHIC: {
      double w_rad = 2.0 * std::tan(hfov / 4.0);
      double pixels_per_rad = width / w_rad;
      double h_rad = height / pixels_per_rad;
      return 4.0 * std::atan(h_rad / 2.0);
    }
    case EU_SPHERICAL:
    case EU_FISHEYE:
      return hfov * height / width;
    default:
      return hfov;   // cubemap/biatan6 fall through to this in the reference
  }
}

inline double get_step(int prj, int width, int /*height*/, double hfov)
{
  switch (prj) {
    case EU_RECTILINEAR:
    case EU_CUBEMAP:
      return std::atan(2.0 * std::tan(hfov / 2.0) / width);
    case EU_BIATAN6:
    case EU_SPHERICAL:
    case EU_CYLINDRICAL:
    case EU_FISHEYE:
      return hfov / width;
    case EU_STEREOGRAPHIC:
      return std::atan(4.0 * std::tan(hfov / 4.0) / width);
  }
  return 0.0;
}

inline void get_extent(int prj, int width, int height, double hfov, double *e)
{
  double ax = -hfov / 2.0, bx = hfov / 2.0;
  double by = get_vfov(prj, width, height, hfov) / 2.0, ay = -by;
  double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
  switch (prj) {
    case EU_SPHERICAL:
    case EU_FISHEYE: x0 = ax; x1 = bx; y0 = ay; y1 = by; break;
    case EU_CYLINDRICAL: x0 = ax; x1 = bx; y0 = std::tan(ay); y1 = std::tan(by); break;
    case EU_RECTILINEAR:
      x0 = std::tan(ax); x1 = std::tan(bx); y0 = std::tan(ay); y1 = std::tan(by); break;
    case EU_STEREOGRAPHIC:
      x0 = 2.0 * std::tan(ax / 2.0); x1 = 2.0 * std::tan(bx / 2.0);
      y0 = 2.0 * std::tan(ay / 2.0); y1 = 2.0 * std::tan(by / 2.0); break;
    case EU_CUBEMAP:
    case EU_BIATAN6:
      x0 = std::tan(ax); x1 = std::tan(bx); y0 = 6 * x0; y1 = 6 * x1; break;
  }
  e[0] = x0; e[1] = x1; e[2] = y0; e[3] = y1;
}

// ---- orientation: envutil_payload.cc:136-218, geometry.h:74-97 ------------
// Imath (not part of the reference tree) supplies Eulerf(roll, pitch, yaw,
// ZXY).toQuat(), Quat::invert and Vec3 * Quat; this follows Imath 3's
// ImathEuler.h / ImathQuat.h for that order (static frame, even parity,
// first axis Z: i = 2, j = 0, k = 1).

struct mat3 { double m[9]; };

inline mat3 make_r3(double roll, double pitch, double yaw, bool inverse)
{
  float ax = (float)roll, ay = (float)pitch, az = (float)yaw;
  float ti = (float)(ax * 0.5), tj = (float)(ay * 0.5), th = (float)(az * 0.5);
  float ci = std::cos(ti), cj = std::cos(tj), ch = std::cos(th);
  float si = std::sin(ti), sj = std::sin(tj), sh = std::sin(th);
  float cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
  float a[3];
  a[2] = cj * sc - sj * cs;
  a[0] = (float)((cj * ss + sj * cc) * 1.0);
  a[1] = cj * cs - sj * sc;
  float qr = cj * cc + sj * ss;
  double r = qr, v[3] = { a[0], a[1], a[2] };
  if (inverse) {
    double qdot = r * r + (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    r /= qdot;
    for (int i = 0; i < 3; i++) v[i] = -v[i] / qdot;
  }
  auto cross = [](const double *p, const double *q, double *o) {
    o[0] = p[1] * q[2] - p[2] * q[1];
    o[1] = p[2] * q[0] - p[0] * q[2];
    o[2] = p[0] * q[1] - p[1] * q[0];
  };
  mat3 out;
  for (int e = 0; e < 3; e++) {
    double in[3] = { 0, 0, 0 }, A[3], B[3];
    in[e] = 1.0;
    cross(v, in, A);
    cross(v, A, B);
    for (int i = 0; i < 3; i++) out.m[3 * e + i] = in[i] + 2.0 * (r * A[i] + B[i]);
  }
  return out;
}

inline mat3 rotate(const mat3 &l, const mat3 &r)
{
  mat3 o;
  for (int row = 0; row < 3; row++)
    for (int i = 0; i < 3; i++)
      o.m[3 * row + i] = (l.m[3 * row] * r.m[i] + l.m[3 * row + 1] * r.m[3 + i])
                         + l.m[3 * row + 2] * r.m[6 + i];
  return o;
}

// ---- twining taps: envutil_main.cc:1253-1355 -------------------------------

inline int make_spread(int w, int h, float d, float sigma, float threshold,
                       std::vector<float> &out)
{
  if (w <= 2) w = 2;
  if (h <= 0) h = w;
  float wgt = (float)(1.0 / (w * h));
  double x0 = -(w - 1.0) / (2.0 * w), dx = 1.0 / w;
  double y0 = -(h - 1.0) / (2.0 * h), dy = 1.0 / h;
  out.clear();
  sigma = (float)(sigma * -x0);
  double sum = 0.0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float wf = 1.0f;
      if (sigma > 0.0) {
        double wx = (x0 + x * dx) / sigma, wy = (y0 + y * dy) / sigma;
        wf = (float)std::exp(-std::sqrt(wx * wx + wy * wy));
      }
      out.push_back((float)(d * (x0 + x * dx)));
      out.push_back((float)(d * (y0 + y * dy)));
      out.push_back(wf * wgt);
      sum += wf * wgt;
    }
  if (sigma != 0.0) {
    double th_sum = 0.0;
    bool renorm = false;
    for (size_t i = 2; i < out.size(); i += 3) {
      out[i] = (float)(out[i] / sum);
      if (out[i] >= threshold) th_sum += out[i];
      else { renorm = true; out[i] = 0.0f; }
    }
    if (renorm) {
      std::vector<float> keep;
      for (size_t i = 0; i < out.size(); i += 3) {
        float wv = (float)(out[i + 2] / th_sum);
        if (wv > 0.0f) { keep.push_back(out[i]); keep.push_back(out[i + 1]); keep.push_back(wv); }
      }
      out.swap(keep);
    }
  }
  return (int)(out.size() / 3);
}

// ---- cubemap geometry: metrics_t, cubemap.h:233-400 ------------------------

struct metrics {
  long face_px, section_px, left_frame_px, right_frame_px;
  long inherent_px;          // pixels of the face image beyond the 90-degree face proper, per side (hfov > 90 degrees)
  double model_to_px, px_to_model, section_md, refc_md;
};

inline metrics make_metrics(long face_px, double face_fov, long support_min, long tile_px)
{
  metrics m;
  double overscan_md = 0.0, diameter_md = 2.0;
  m.face_px = face_px;
  if (face_fov > M_PI_2) {
    double radius_md = std::tan(face_fov / 2.0);
    diameter_md = 2.0 * radius_md;
    overscan_md = radius_md - 1.0;
  }
  m.model_to_px = double(face_px) / diameter_md;
  m.px_to_model = diameter_md / double(face_px);
  long inherent = (long)std::trunc(m.model_to_px * overscan_md);
  m.inherent_px = inherent;
  long additional = inherent < support_min ? support_min - inherent : 0;
  long px_min = face_px + 2 * additional;
  long n_tiles = px_min / tile_px;
  if (n_tiles * tile_px < px_min) n_tiles++;
  m.section_px = n_tiles * tile_px;
  long frame_total = m.section_px - face_px;
  m.left_frame_px = frame_total / 2;
  m.right_frame_px = frame_total - m.left_frame_px;
  m.section_md = m.px_to_model * m.section_px;
  m.refc_md = m.px_to_model * (double(m.left_frame_px) + double(face_px) / 2.0);
  return m;
}

// ---- b-spline container geometry: zimt/bspline.h:305-450 -------------------

inline long left_brace(int degree, int bc)
{
  long n = degree / 2;
  if (bc == EU_BC_REFLECT) n++;
  else if (degree & 1) n++;
  if (bc == EU_BC_PERIODIC && !(degree & 1)) n++;
  return n;
}

inline long right_brace(int degree, int bc)
{
  long n = degree / 2;
  if (bc == EU_BC_REFLECT && !(degree & 1)) n++;
  if (degree & 1) n++;
  if (bc == EU_BC_PERIODIC) n++;
  return n;
}

inline void container_geometry(int degree, int bc0, int bc1, int64_t w, int64_t h,
                               eu_container *g)
{
  g->left[0] = left_brace(degree, bc0);  g->left[1] = left_brace(degree, bc1);
  g->right[0] = right_brace(degree, bc0); g->right[1] = right_brace(degree, bc1);
  g->core[0] = w; g->core[1] = h;
  g->shape[0] = w + g->left[0] + g->right[0];
  g->shape[1] = h + g->left[1] + g->right[1];
}

// ---- basis: weight matrix (zimt/basis.h:419-545), poles (zimt/poles.h) -----

// centred B-spline of degree n at x2/2, exact rational evaluated in long double
inline long double basis_half(int x2, int n)
{
  if (n == 0) return (x2 == -1 || x2 == 0) ? 1.0L : 0.0L;
  int ax = x2 < 0 ? -x2 : x2;
  if (ax > n) return 0.0L;
  long double acc = 0.0L, binom = 1.0L;
  for (int k = 0; k <= n + 1; k++) {
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

// m[c * (degree+1) + row]: Taylor coefficient 'row' of the polynomial piece
// that weights tap c
inline void weight_matrix(int degree, float *m)
{
  const int order = degree + 1;
  long double line[EU_MAX_DEGREE + 2];
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
    for (int k = 0; k <= degree; k++) m[k * order + row] = (float)(line[k] / faculty);
  }
}

// prefilter poles: the degree/2 roots in (-1, 0) of sum_k beta^n(k) z^(k+n/2),
// most negative first
inline int poles(int degree, long double *out)
{
  int np = degree / 2;
  if (np == 0) return 0;
  int len = 2 * np + 1;
  long double c[2 * EU_MAX_DEGREE + 3];
  for (int k = -np; k <= np; k++) c[k + np] = basis_half(2 * k, degree);
  auto ev = [&](long double z, long double *dp) {
    long double p = c[len - 1], d = 0.0L;
    for (int i = len - 2; i >= 0; i--) { d = d * z + p; p = p * z + c[i]; }
    if (dp) *dp = d;
    return p;
  };
  int found = 0;
  long double za = -1.0L, pa = ev(za, nullptr);
  while (found < np && za < -1e-12L) {
    long double zb = za * 0.9L, pb = ev(zb, nullptr);
    if ((pa < 0) != (pb < 0)) {
      long double lo = za, hi = zb, plo = pa;
      for (int it = 0; it < 90; it++) {
        long double mid = 0.5L * (lo + hi), pm = ev(mid, nullptr);
        if ((pm < 0) == (plo < 0)) { lo = mid; plo = pm; } else hi = mid;
      }
      long double z = 0.5L * (lo + hi);
      for (int it = 0; it < 4; it++) { long double dp, p = ev(z, &dp); z -= p / dp; }
      out[found++] = z;
    }
    za = zb; pa = pb;
  }
  return found;
}

// ---- stepper tables: stepper.h:294-350 and the per-projection init() --------

struct stepper_tables {
  int form, norm_mode;
  std::vector<float> col;     // [4][W]
  std::vector<float> row;     // [H][EU_ROW_FLOATS]
};

// planar x of every column, with the accumulation the reference performs:
// segment-start value of the pixel's lane plus k additions of delta
// W: width of the frame; the OW columns written start at discrete coordinate
// x_off (bill.get_offset, wielding.h:215-224): segments of 512 count from the
// first PROCESSED column, the offset is added to the coordinate init() receives
inline void planar_columns(int W, float a0, float a1, float bias, float *out, int x_off = 0,
                           int OW = -1)
{
  if (OW < 0) OW = W;
  float fx1 = (float)(a1 / (2.0 * W));
  float fx0 = (float)(a0 / (2.0 * W));
  float bias_x = bias * (a1 - a0) / (float)W;
  float delta = (float)EU_LANES * (a1 - a0) / (float)W;
  for (int seg = 0; seg < OW; seg += EU_SEGMENT)
    for (int lane = 0; lane < EU_LANES && seg + lane < OW; lane++) {
      float ll0 = (float)(2 * lane) + (float)((seg + x_off) * 2 + 1);
      float p = bias_x + ll0 * fx1 + ((float)(2 * W) - ll0) * fx0;
      for (int x = seg + lane; x < OW && x < seg + EU_SEGMENT; x += EU_LANES) {
        out[x] = p;
        p += delta;
      }
    }
}

inline float planar_row(int H, float b0, float b1, float bias, int y)
{
  float fy1 = (float)(b1 / (2.0 * H));
  float fy0 = (float)(b0 / (2.0 * H));
  float bias_y = bias * (b1 - b0) / (float)H;
  int ll1 = y * 2 + 1;
  return bias_y + ll1 * fy1 + (float)(2 * H - ll1) * fy0;
}

// to_screen_t's LUT (envutil_payload.cc:251-287, :330-334): 256 knots of
// 255 * RGB2sRGB(i / 255.0) evaluated in double with libm's pow, narrowed to
// float, plus the NATURAL brace coefficient on the right (never weighted)
inline void screen_lut(float *lut257)
{
  for (int i = 0; i < 256; i++) {
    double x = i / double(256 - 1);
    double r = 1.055 * std::pow(x, 0.41666666666666667) - 0.055;
    if (x <= 0.0031308) r = 12.92 * x;
    lut257[i] = (float)(r * 255.0);
  }
  lut257[256] = lut257[255] + lut257[255] - lut257[254];
}

// Fills the tables for one target. `normalize` is the stepper template flag:
// false for single-facet rendering without twining, true otherwise
// (envutil_payload.cc:2118, :2227). Returns false for an unknown projection.
inline bool build_stepper_tables(const eu_target &t, const mat3 &basis, bool normalize,
                                 bool twine, stepper_tables &tb)
{
  const int W = t.width, H = t.height, prj = t.projection;
  // store_cropped: OW x OH processed pixels starting at (X0, Y0) of the W x H frame
  const bool crop = t.crop_w > 0;
  const int OW = crop ? t.crop_w : W, OH = crop ? t.crop_h : H;
  const int X0 = crop ? t.crop_x0 : 0, Y0 = crop ? t.crop_y0 : 0;
  const float a0 = (float)t.x0, a1 = (float)t.x1, b0 = (float)t.y0, b1 = (float)t.y1;
  float xx[3], yy[3], zz[3];
  for (int i = 0; i < 3; i++) { xx[i] = (float)basis.m[i]; yy[i] = (float)basis.m[3 + i]; zz[i] = (float)basis.m[6 + i]; }
  tb.col.assign((size_t)6 * OW, 0.0f);
  tb.row.assign((size_t)OH * EU_ROW_FLOATS, 0.0f);
  std::vector<float> p0((size_t)OW), p0b((size_t)OW);
  planar_columns(W, a0, a1, 0.0f, p0.data(), X0, OW);
  if (twine) planar_columns(W, a0, a1, 0.25f, p0b.data(), X0, OW);
  // the raw planar x of every column (generic_stepper facets, eu_generic)
  for (int x = 0; x < OW; x++) {
    tb.col[(size_t)4 * OW + x] = p0[x];
    if (twine) tb.col[(size_t)5 * OW + x] = p0b[x];
  }
  const float section_md = a1 - a0, refc_md = (float)((a1 - a0) / 2.0);
  const float q = (float)(M_PI / 4.0);
  tb.norm_mode = EU_NORM_NONE;
  switch (prj) {
    case EU_SPHERICAL:
    case EU_CYLINDRICAL:
      tb.form = EU_FORM_BCA;
      if (prj == EU_CYLINDRICAL && normalize) tb.norm_mode = EU_NORM_CYL;
      for (int x = 0; x < OW; x++) {
        tb.col[x] = std::sin(p0[x]);
        tb.col[(size_t)OW + x] = std::cos(p0[x]);
        if (twine) {
          tb.col[(size_t)2 * OW + x] = std::sin(p0b[x]);
          tb.col[(size_t)3 * OW + x] = std::cos(p0b[x]);
        }
      }
      break;
    case EU_RECTILINEAR:
    case EU_CUBEMAP:
      tb.form = EU_FORM_BA;
      if (normalize) tb.norm_mode = EU_NORM_DIV;
      for (int x = 0; x < OW; x++) {
        tb.col[x] = p0[x];
        if (twine) tb.col[(size_t)2 * OW + x] = p0b[x];
      }
      break;
    case EU_BIATAN6:
      tb.form = EU_FORM_BA;
      if (normalize) tb.norm_mode = EU_NORM_DIV;
      for (int x = 0; x < OW; x++) {
        tb.col[x] = std::tan(p0[x] * q);
        if (twine) tb.col[(size_t)2 * OW + x] = std::tan(p0b[x] * q);
      }
      break;
    case EU_FISHEYE:
    case EU_STEREOGRAPHIC:
      // fisheye_stepper::work (stepper.h:1019-1030) and stereographic_stepper::work
      // (:1146-1157) have no per-row or per-column invariant beyond the planar
      // coordinates themselves
      tb.form = prj == EU_FISHEYE ? EU_FORM_FISH : EU_FORM_STER;
      for (int x = 0; x < OW; x++) {
        tb.col[x] = p0[x];
        if (twine) tb.col[(size_t)2 * OW + x] = p0b[x];
      }
      break;
    default:
      return false;
  }
  for (int y = 0; y < OH; y++) {
    for (int v = 0; v < (twine ? 2 : 1); v++) {
      float p1 = planar_row(H, b0, b1, v ? 0.25f : 0.0f, y + Y0);
      float *r = &tb.row[(size_t)y * EU_ROW_FLOATS + EU_ROW_VARIANT * v];   // A, B, C, planar y
      r[9] = p1;
      switch (prj) {
        case EU_SPHERICAL: {     // stepper.h:605-667
          float sy = std::sin(p1), rr = std::cos(p1);
          for (int i = 0; i < 3; i++) { r[3 + i] = xx[i] * rr; r[i] = yy[i] * sy; r[6 + i] = zz[i] * rr; }
          break;
        }
        case EU_CYLINDRICAL:     // stepper.h:760-800
          for (int i = 0; i < 3; i++) { r[3 + i] = xx[i]; r[i] = yy[i] * p1; r[6 + i] = zz[i]; }
          break;
        case EU_RECTILINEAR:     // stepper.h:895-932
          for (int i = 0; i < 3; i++) { r[3 + i] = xx[i]; r[i] = yy[i] * p1 + zz[i]; }
          break;
        case EU_FISHEYE:
        case EU_STEREOGRAPHIC:
          for (int i = 0; i < 3; i++) { r[i] = xx[i]; r[3 + i] = yy[i]; r[6 + i] = zz[i]; }
          break;
        default: {               // cubemap, biatan6: stepper.h:1274-1358, :1449-1560
          int face = (y + Y0) / W;
          float pp = p1 + (float)(3 - face) * section_md - refc_md;
          if (prj == EU_BIATAN6) pp = std::tan(pp * q);
          for (int i = 0; i < 3; i++) {
            float ccc, vvv;
            switch (face) {
              case 0: ccc = (float)(-1.0 * (double)xx[i] + (double)(pp * yy[i])); vvv = zz[i]; break;
              case 1: ccc = (float)(1.0 * (double)xx[i] + (double)(pp * yy[i])); vvv = -zz[i]; break;
              case 2: ccc = (float)(-1.0 * (double)yy[i] - (double)(pp * zz[i])); vvv = -xx[i]; break;
              case 3: ccc = (float)(1.0 * (double)yy[i] + (double)(pp * zz[i])); vvv = -xx[i]; break;
              case 4: ccc = (float)((double)(pp * yy[i]) + 1.0 * (double)zz[i]); vvv = xx[i]; break;
              default: ccc = (float)((double)(pp * yy[i]) - 1.0 * (double)zz[i]); vvv = -xx[i]; break;
            }
            r[i] = ccc;
            r[3 + i] = vvv;
          }
        }
      }
    }
  }
  return true;
}

// generic_r3(ft, fs) for ft = the job's target (no translation of its own), envutil_payload.cc:
// 1757-1810: the facet's tf3d_t. Float matrix products as rotate(r3_t<float>, r3_t<float>)
// (geometry.h:80-97) forms them. false: target projections whose planar -> ray functor
// (ir_to_ray_t, ba6_to_ray_t) is not built.
inline void r3f(double roll, double pitch, double yaw, bool inverse, float *m)
{
  const mat3 d = make_r3(roll, pitch, yaw, inverse);
  for (int i = 0; i < 9; i++) m[i] = (float)d.m[i];
}
inline void rotate_f(const float *in, const float *m, float *out)
{
  float t[3];
  for (int c = 0; c < 3; c++) { float v = in[0] * m[c] + in[1] * m[3 + c]; t[c] = v + in[2] * m[6 + c]; }
  out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}
inline void rotate_mf(const float *l, const float *r, float *o)
{
  float t[9];
  for (int i = 0; i < 3; i++) rotate_f(l + 3 * i, r, t + 3 * i);
  std::memcpy(o, t, sizeof t);
}
inline bool has_translation(const eu_facet &f) { return f.tr_x != 0 || f.tr_y != 0 || f.tr_z != 0; }

inline bool has_2d_tf(const eu_facet &f)
{
  return f.h != 0.0 || f.v != 0.0 || f.a != 0.0 || f.b != 0.0 || f.c != 0.0 || f.shear_g != 0.0 || f.shear_t != 0.0;
}
// fuse(), envutil_payload.cc:2058-2069: the facet a --single job recreates has lens correction or translation
inline bool generic_target(const eu_target &t) { return t.single && (has_2d_tf(*t.single) || has_translation(*t.single)); }

inline void set_tf3d(eu_tf3d &q, const float *a, const float *b, const float *sh, float dcp)
{
  std::memcpy(q.trg_to_md, a, 9 * sizeof(float));
  std::memcpy(q.md_to_src, b, 9 * sizeof(float));
  rotate_mf(a, b, q.trg_to_src);
  for (int c = 0; c < 3; c++) q.shift[c] = sh[c];
  q.has_shift = q.shift[0] != 0 || q.shift[1] != 0 || q.shift[2] != 0;
  q.dcp = dcp;
}

// generic_r3(ft, fs), envutil_payload.cc:1636-1755: ft = the job's target (with the translation of the facet
// a --single job recreates), fs = the source facet
inline bool make_generic(const eu_target &t, const eu_facet &f, eu_generic &g)
{
  std::memset(&g, 0, sizeof g);
  switch (t.projection) {
    case EU_SPHERICAL: case EU_CYLINDRICAL: case EU_RECTILINEAR: case EU_STEREOGRAPHIC: case EU_FISHEYE:
    case EU_CUBEMAP: case EU_BIATAN6: break;
    default: return false;
  }
  double ft6[6] = { 0, 0, 0, 0, 0, 0 };
  if (t.single) { ft6[0] = t.single->tr_x; ft6[1] = t.single->tr_y; ft6[2] = t.single->tr_z;
                  ft6[3] = t.single->tp_y; ft6[4] = t.single->tp_p; ft6[5] = t.single->tp_r; }
  float r_camera[9], rt_tp[9], rt_tpi[9], rs_tp[9], rs_tpi[9], r_facet[9];
  r3f(t.roll, t.pitch, t.yaw, false, r_camera);
  r3f(ft6[5], ft6[4], ft6[3], true, rt_tp);
  r3f(ft6[5], ft6[4], ft6[3], false, rt_tpi);
  r3f(f.tp_r, f.tp_p, f.tp_y, true, rs_tp);
  r3f(f.tp_r, f.tp_p, f.tp_y, false, rs_tpi);
  r3f(f.roll, f.pitch, f.yaw, true, r_facet);
  const bool have_ttp = ft6[0] != 0 || ft6[1] != 0 || ft6[2] != 0, have_stp = has_translation(f);
  // rotate(xel_t<double,3>(shift), r_tp): double vector, float matrix, narrowed on assignment
  auto to_plane = [](float *sh, const float *m) {
    const double v[3] = { sh[0], sh[1], sh[2] };
    for (int c = 0; c < 3; c++)
      sh[c] = (float)((v[0] * (double)m[c] + v[1] * (double)m[3 + c]) + v[2] * (double)m[6 + c]);
  };
  float shift_t[3] = { (float)ft6[0], (float)ft6[1], (float)ft6[2] };
  if (ft6[3] != 0 || ft6[4] != 0 || ft6[5] != 0) to_plane(shift_t, rt_tp);
  const float dcp = (float)(1.0 - (double)shift_t[2]);
  for (int c = 0; c < 3; c++) shift_t[c] = -shift_t[c];
  float shift_s[3] = { (float)f.tr_x, (float)f.tr_y, (float)f.tr_z };
  if (f.tp_y != 0 || f.tp_p != 0 || f.tp_r != 0) to_plane(shift_s, rs_tp);
  const float zero3[3] = { 0.0f, 0.0f, 0.0f };
  float m1[9], m2[9];
  g.ntf = 1;
  if (have_ttp && have_stp) {
    rotate_mf(r_camera, rt_tp, m1);
    set_tf3d(g.tf[0], m1, rt_tpi, shift_t, dcp);
    rotate_mf(rs_tpi, r_facet, m2);
    set_tf3d(g.tf[1], rs_tp, m2, shift_s, 1.0f);
    g.ntf = 2;
  } else if (have_ttp) {
    rotate_mf(r_camera, rt_tp, m1);
    rotate_mf(rt_tpi, r_facet, m2);
    set_tf3d(g.tf[0], m1, m2, shift_t, dcp);
  } else if (have_stp) {
    rotate_mf(r_camera, rs_tp, m1);
    rotate_mf(rs_tpi, r_facet, m2);
    set_tf3d(g.tf[0], m1, m2, shift_s, 1.0f);
  } else {
    set_tf3d(g.tf[0], r_camera, r_facet, zero3, 1.0f);
  }
  g.prj = t.projection;
  g.on = 1;
  return true;
}

// inverse_lcp (lens_correction.h:236-301) as pto_planar<T, L, true> builds it for the facet of a --single
// job (environment.h:247-252: sz = 100; r_max from the facet's extent, envutil_basic.h:508-520): Newton
// iteration per knot in double (eu_polynomial::inverse, :112-139), the knots prefiltered as a cubic NATURAL
// b-spline (recursive.h:631-733 for one pole) and braced. coef: nk + 4 floats, the core starts at coef + 2.
// false: the polynomial has no inverse there (the reference asserts).
inline bool make_inverse_lcp(double a, double b, double c, double r_max_in, int sz, std::vector<float> &coef,
                             double &rr_max)
{
  const double cf[5] = { a, b, c, 1.0 - (a + b + c), 0.0 };
  double dcf[5];
  { int power = 4; for (int i = 0; i <= 4; i++) { dcf[i] = cf[i] * power; --power; } }
  auto fn = [&](double x) { double sum = 0.0, power = 1.0; for (int i = 0; i <= 4; i++) { sum += cf[4 - i] * power; power *= x; } return sum; };
  auto dfn = [&](double x) { double sum = 0.0, power = 1.0; for (int i = 0; i < 4; i++) { sum += dcf[4 - i - 1] * power; power *= x; } return sum; };
  const int nk = sz + 4;
  const double r_max = r_max_in * ((sz + 3.0) / sz);
  rr_max = fn(r_max);
  coef.assign((size_t)nk + 4, 0.0f);
  float *core = coef.data() + 2;
  for (int i = 0; i < nk; i++) {
    double notch = (double)i / (nk - 1);
    notch *= notch;
    notch *= rr_max;
    double out = i * r_max / sz;
    const double tolerance = 100 * 2.220446049250313e-16;
    double current = out, result, difference = 0.0, last_difference = 1.7976931348623157e308;
    for (int count = 0; count < 16; count++) {
      result = fn(current);
      difference = notch - result;
      if (last_difference == difference) break;
      if (std::fabs(difference) <= tolerance) break;
      last_difference = difference;
      current = current + difference / dfn(current);
    }
    if (!(std::fabs(difference) < tolerance)) return false;
    out = current;
    core[i] = (float)(notch == 0.0 ? 1.0 / dfn(0.0) : (out / notch) - 1);
  }
  // prefilter: degree 3, one pole, NATURAL, tolerance = float epsilon (bspline.h:1017-1041)
  long double lp[8];
  poles(3, lp);
  const float p = (float)lp[0];
  const float g = (float)((1.0L - lp[0]) * (1.0L - 1.0L / lp[0]));
  const int hz = (int)std::ceil(std::log((long double)1.1920928955078125e-07L) / std::log(std::fabs(lp[0])));
  const int M = nk;
  float X;
  {
    // icc NATURAL (recursive.h:321-360)
    if (hz < M) {
      const float c02 = core[0] + core[0];
      float zn = p, Sum = core[0];
      for (int n = 1; n < hz; n++) { Sum = Sum + zn * (c02 - core[n]); zn = zn * p; }
      X = Sum;
    } else {
      float zn = p, iz = 1.0f / p, z2n = (float)std::pow(lp[0], (long double)(M - 1));
      float Sum = ((1.0f + p) / (1.0f - p)) * (core[0] - z2n * core[M - 1]);
      z2n = z2n * (z2n * iz);
      for (int n = 1; n <= M - 2; n++) { Sum = Sum - (zn - z2n) * core[n]; zn = zn * p; z2n = z2n * iz; }
      X = Sum / (1.0f - zn * zn);
    }
  }
  X = g * X;
  core[0] = X;
  for (int n = 1; n < M; n++) { X = g * core[n] + p * X; core[n] = X; }
  X = -(p / ((1.0f - p) * (1.0f - p))) * (core[M - 1] - p * core[M - 2]);   // iacc NATURAL
  core[M - 1] = X;
  for (int n = M - 2; n >= 0; n--) { X = p * (X - core[n]); core[n] = X; }
  // NATURAL brace (brace.h:134+): point mirror on the first / last core value
  for (int k = 1; k <= 2; k++) {
    core[-k] = core[0] + core[0] - core[k];
    core[nk - 1 + k] = core[nk - 1] + core[nk - 1] - core[nk - 1 - k];
  }
  return true;
}

}  // namespace eu
#endif
