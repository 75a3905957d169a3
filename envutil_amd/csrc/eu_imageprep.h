// Load-time edits of a facet's pixels that PTO scripts ask for: exclude masks (k-lines) and lens
// crops (the S clause of an i-line). Host code, run once per image before the pixels go to the
// device (source_t's constructor, environment.h:700-890, does the same on the CPU before it
// prefilters): an alpha plane starts at 1, polygons and the outside of the crop clear it, a 5-tap
// binomial softens it along both axes, every channel of the image is multiplied by it.
//
// Arithmetic follows the reference operation for operation (float products and sums, no
// contraction), including the ORDER of the five products in the binomial: zimt's fir_filter
// (zimt/convolve.h:240-383) keeps the last five samples in a circular buffer and sums the slots
// in slot order, so the order of the terms rotates with the position along the line.
#ifndef EU_IMAGEPREP_H
#define EU_IMAGEPREP_H

#include <cmath>
#include <cstddef>
#include <thread>
#include <vector>

namespace eu {

// fill_polygon (envutil_basic.cc:236-320): scan lines, crossings with their direction, fill
// where the winding number is not zero. clear(x, y) is called for every pixel inside.
template <class F>
inline void fill_polygon(const float *px, const float *py, int n, int left, int top, int right, int bot, F clear)
{
  std::vector<int> node_x(size_t(n > 0 ? n : 1)), dir(size_t(n > 0 ? n : 1));
  for (int y = top; y < bot; y++) {
    int nodes = 0, j = n - 1;
    for (int i = 0; i < n; i++) {
      int cross = 0;
      if (py[i] < float(y) && py[j] >= float(y)) cross = 1;
      else if (py[j] < float(y) && py[i] >= float(y)) cross = -1;
      if (cross) {
        node_x[size_t(nodes)] = int(px[i] + (y - py[i]) / (py[j] - py[i]) * (px[j] - px[i]));
        dir[size_t(nodes++)] = cross;
      }
      j = i;
    }
    // the reference's exchange sort; stable for equal keys, as this insertion sort is
    for (int i = 1; i < nodes; i++) {
      const int kx = node_x[size_t(i)], kd = dir[size_t(i)];
      int k = i - 1;
      while (k >= 0 && node_x[size_t(k)] > kx) {
        node_x[size_t(k + 1)] = node_x[size_t(k)]; dir[size_t(k + 1)] = dir[size_t(k)];
        k--;
      }
      node_x[size_t(k + 1)] = kx; dir[size_t(k + 1)] = kd;
    }
    int winding = 0;
    for (int i = 0; i < nodes; i++) {
      winding += dir[size_t(i)];
      if (!winding) continue;
      if (i + 1 >= nodes) break;          // an open winding at the last node has no partner
      if (node_x[size_t(i)] >= right) break;
      if (node_x[size_t(i + 1)] > left) {
        if (node_x[size_t(i)] < left) node_x[size_t(i)] = left;
        if (node_x[size_t(i + 1)] > right) node_x[size_t(i + 1)] = right;
        for (int x = node_x[size_t(i)]; x < node_x[size_t(i + 1)]; x++) clear(x, y);
      }
    }
  }
}

// zimt's REFLECT extrapolation (zimt/extrapolate.h:141-155)
inline int reflect_index(int i, int w)
{
  if (i < 0) i = -1 - i;
  if (i >= w) {
    i %= 2 * w;
    if (i >= w) i = 2 * w - 1 - i;
  }
  return i;
}

// one line of the binomial (1 4 6 4 1) / 16, headroom 2, REFLECT at both ends
inline void binomial_line(const float *in, float *out, int n, ptrdiff_t stride)
{
  static const float kf[5] = { float(1.0 / 16.0), float(4.0 / 16.0), float(6.0 / 16.0), float(4.0 / 16.0),
                               float(1.0 / 16.0) };
  std::vector<float> line(size_t(n) + 4);
  for (int m = -2; m < n + 2; m++) line[size_t(m + 2)] = in[ptrdiff_t(reflect_index(m, n)) * stride];
  for (int t = 0; t < n; t++) {
    // slot s of the circular buffer holds sample t - 2 + ((s - t) mod 5), weighted kf[(s - t) mod 5]
    int k = ((0 - t) % 5 + 5) % 5;
    float r = line[size_t(t + k)] * kf[k];
    for (int s = 1; s < 5; s++) {
      k = k == 4 ? 0 : k + 1;
      r += line[size_t(t + k)] * kf[k];
    }
    out[ptrdiff_t(t) * stride] = r;
  }
}

struct mask_polygon { int n; const float *x, *y; };

// fn(y0, y1) over [0, n) on up to 16 host threads (rows are independent in every pass below)
template <class F>
inline void parallel_rows(int n, F fn)
{
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : nt > 16 ? 16 : nt;
  if (n < 256 || nt == 1) { fn(0, n); return; }
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt; t++) {
    const int y0 = int((long long)n * t / nt), y1 = int((long long)n * (t + 1) / nt);
    if (y1 > y0) pool.emplace_back([=] { fn(y0, y1); });
  }
  for (auto &th : pool) th.join();
}

// axis 1 of the binomial, row by row: output row t is the slot-ordered sum of the five rows
// t - 2 + ((s - t) mod 5) (REFLECT beyond the ends) - per element the same operations in the same
// order as binomial_line down a column, but along contiguous memory
inline void binomial_rows(const float *in, float *out, int w, int h, int y0, int y1)
{
  static const float kf[5] = { float(1.0 / 16.0), float(4.0 / 16.0), float(6.0 / 16.0), float(4.0 / 16.0),
                               float(1.0 / 16.0) };
  for (int t = y0; t < y1; t++) {
    int k = ((0 - t) % 5 + 5) % 5;
    float *o = out + size_t(t) * w;
    {
      const float *r = in + size_t(reflect_index(t - 2 + k, h)) * w;
      const float kk = kf[k];
      for (int x = 0; x < w; x++) o[x] = r[x] * kk;
    }
    for (int s = 1; s < 5; s++) {
      k = k == 4 ? 0 : k + 1;
      const float *r = in + size_t(reflect_index(t - 2 + k, h)) * w;
      const float kk = kf[k];
      for (int x = 0; x < w; x++) o[x] += r[x] * kk;
    }
  }
}

// alpha plane of a facet (w x h floats): environment.h:727-843
inline void facet_alpha(float *alpha, int w, int h, const mask_polygon *polys, int npolys, int crop_kind,
                        int cx0, int cx1, int cy0, int cy1)
{
  parallel_rows(h, [&](int y0, int y1) { for (size_t i = size_t(y0) * w; i < size_t(y1) * w; i++) alpha[i] = 1.0f; });
  for (int p = 0; p < npolys; p++)
    fill_polygon(polys[p].x, polys[p].y, polys[p].n, 0, 0, w, h,
                 [&](int x, int y) { alpha[size_t(y) * w + x] = 0.0f; });
  if (crop_kind == 2) {
    // elliptic crop of a fisheye image
    const float a = float(std::fabs(double(cx1 - cx0)) / 2.0), b = float(std::fabs(double(cy1 - cy0)) / 2.0);
    const float mx = float((cx0 + cx1) / 2.0), my = float((cy0 + cy1) / 2.0);
    parallel_rows(h, [&](int y0, int y1) {
    for (int y = y0; y < y1; y++) {
      const float dy = std::fabs(float(y) - my);
      if (dy > b) {
        for (int x = 0; x < w; x++) alpha[size_t(y) * w + x] = 0.0f;
        continue;
      }
      // (dy * dy) / (b * b) is a float quotient in the reference, the rest runs in double
      const float xmargin = float(std::sqrt(double(a * a) * (1.0 - double((dy * dy) / (b * b)))));
      for (int x = 0; x < w; x++) {
        const float dx = std::fabs(float(x) - mx);
        if (dx > xmargin) alpha[size_t(y) * w + x] = 0.0f;
      }
    }
    });
  } else if (crop_kind == 1) {
    parallel_rows(h, [&](int y0, int y1) {
      for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++)
          if (x < cx0 || x >= cx1 || y < cy0 || y >= cy1) alpha[size_t(y) * w + x] = 0.0f;
    });
  }
  // convolve(alpha, alpha, {REFLECT, REFLECT}, binomial, 2): axis 0, then axis 1 on the result
  std::vector<float> tmp(size_t(w) * h);
  parallel_rows(h, [&](int y0, int y1) {
    for (int y = y0; y < y1; y++) binomial_line(alpha + size_t(y) * w, tmp.data() + size_t(y) * w, w, 1);
  });
  parallel_rows(h, [&](int y0, int y1) { binomial_rows(tmp.data(), alpha, w, h, y0, y1); });
}

}  // namespace eu
#endif
