// TEST INFRASTRUCTURE - not product code.
//
// Thin C-ABI harness around the reference's own zimt headers, compiled IN PLACE
// from /root/reference (see oracle/Makefile, target _ref/libref_zimt.so). Nothing
// from the reference is copied here: this file only *calls* zimt's public
// templates so that the plain-C restatement in oracle/eu_oracle.c can be
// checked bit-for-bit against the reference implementation of
//   * b-spline container layout + bracing   (zimt/bspline.h, zimt/brace.h)
//   * recursive prefilter                   (zimt/prefilter.h, recursive.h, filter.h)
//   * gates + split + weights + weighted sum (zimt/map.h, basis.h, eval.h)
//   * the strip-mining driver               (zimt/wielding.h, get.h, put.h)
// Back-end: zimt "goading" (no USE_HWY/USE_VC/USE_STDSIMD), LANES = 16,
// segment 512, compiled with -ffp-contract=off: the pinned oracle definition
// of SURVEY.md §8(c).
//
//   * the binomial that softens a masked facet's alpha plane (zimt/convolve.h, called with the
//     arguments of environment.h:833-843)
//   * the PTO lens polynomial lcp            (lens_correction.h: it includes nothing
//     but zimt/eval.h, so it compiles in place)
//   * masking_t / alpha_masking_t            (masking.h: zimt/bspline.h + zimt/eval.h only), round 3
//   * pto_parser_type                        (pto.h: standard library only, behind <iostream> as in
//     envutil_main.cc), round 3
// envutil's other headers (geometry.h, stepper.h, environment.h, cubemap.h,
// twining.h) cannot be compiled here: every one of them reaches
// envutil_basic.h:198-200, which includes OpenImageIO (absent from the image).
// Those stages are restated from the source text and are NOT pinned by this
// library ("Oracle").

#include <cstring>
#include <vector>
#include <memory>
#include "zimt/zimt.h"
#include "zimt/bspline.h"
#include "zimt/prefilter.h"
#include "zimt/eval.h"
#include "zimt/convolve.h"
#include "lens_correction.h"
#include "masking.h"        // masking_t / alpha_masking_t: includes nothing but zimt/bspline.h and zimt/eval.h
#include <iostream>         // pto.h uses std::cerr / std::cout and leaves the include to envutil_main.cc
#include "pto.h"            // pto_parser_type: <map>, <vector>, <fstream>, <regex>
#include <string>

namespace {

static const std::size_t L = zimt::simd_traits<float>::default_size;

struct handle_base {
  int nch;
  virtual ~handle_base() {}
  virtual void prefilter(int prefilter_degree) = 0;
  virtual void spherical(int prefilter_degree) = 0;
  virtual void brace(int axis) = 0;
  virtual void geometry(long *out) = 0;
  virtual float *container_data() = 0;
  virtual void eval(const float *crd, long n, float *out) = 0;
  virtual void process_affine(long w, long h, const float *aff, float *out) = 0;
};

template <int NCH>
struct handle : handle_base {
  typedef zimt::xel_t<float, NCH> px_t;
  typedef zimt::bspline<px_t, 2> spl_t;
  typedef zimt::xel_t<float, 2> crd_t;
  typedef zimt::simdized_type<crd_t, L> crd_v;
  typedef zimt::simdized_type<px_t, L> px_v;
  std::unique_ptr<spl_t> sp;

  handle(const float *core, long W, long H, int degree, int bc0, int bc1) {
    nch = NCH;
    sp.reset(new spl_t({std::size_t(W), std::size_t(H)}, degree,
                       {zimt::bc_code(bc0), zimt::bc_code(bc1)}));
    for (long y = 0; y < H; y++)
      for (long x = 0; x < W; x++) {
        px_t p;
        for (int c = 0; c < NCH; c++) p[c] = core[(y * W + x) * NCH + c];
        sp->core[{x, y}] = p;
      }
  }

  // source_t's ordinary branch: the spline is filtered with prefilter_degree but
  // keeps the frame it was allocated with (environment.h:933-936)
  void prefilter(int prefilter_degree) override {
    int keep = sp->spline_degree;
    sp->spline_degree = prefilter_degree;
    sp->prefilter();
    sp->spline_degree = keep;
  }

  // The sequence of zimt calls envutil makes for a full 360x180 degree lat/lon
  // image (environment.h:356-522): periodic rows, then every left-half column
  // stacked on the flipped opposite column as ONE periodic line, tolerance 1e-4,
  // over-the-pole frame rows, periodic brace along x.
  void spherical(int prefilter_degree) override {
    typedef zimt::view_t<2, px_t> view_type;
    typedef zimt::recursive_filter<zimt::simdized_type, float> stripe_t;
    int degree = prefilter_degree;
    auto out = sp->core;
    zimt::iir_filter_specs specs(zimt::PERIODIC, degree / 2,
                                 zimt_constants::precomputed_poles[degree],
                                 0.0001);
    if (degree > 1)
      zimt::detail::separable_filter<view_type, view_type, stripe_t>()
        (out, out, 0, specs);
    auto shape = out.shape;
    shape[0] /= 2;
    auto neg = out.strides;
    neg[1] = -neg[1];
    view_type upper(shape, out.strides, out.data());
    view_type lower(shape, neg,
                    out.data() + (shape[1] - 1) * out.strides[1]
                               + shape[0] * out.strides[0]);
    std::vector<view_type> stack{upper, lower};
    if (degree > 1)
      zimt::detail::separable_filter<view_type, view_type, stripe_t>()
        (stack, stack, 1, specs, zimt::default_njobs);
    view_type left(shape, out.strides, out.data());
    view_type right(shape, out.strides, out.data() + shape[0]);
    long H = shape[1];
    for (long k = 0; k < long(sp->left_frame[1]); k++) {
      left.slice(1, -1 - k).copy_data(right.slice(1, k));
      right.slice(1, -1 - k).copy_data(left.slice(1, k));
    }
    for (long k = 0; k < long(sp->right_frame[1]); k++) {
      left.slice(1, H + k).copy_data(right.slice(1, H - 1 - k));
      right.slice(1, H + k).copy_data(left.slice(1, H - 1 - k));
    }
    sp->brace(0);
  }

  void brace(int axis) override { sp->brace(axis); }

  void geometry(long *o) override {
    o[0] = sp->container.shape[0];
    o[1] = sp->container.shape[1];
    o[2] = sp->container.strides[0];
    o[3] = sp->container.strides[1];
    o[4] = sp->left_frame[0];
    o[5] = sp->left_frame[1];
    o[6] = sp->right_frame[0];
    o[7] = sp->right_frame[1];
    o[8] = sp->core.shape[0];
    o[9] = sp->core.shape[1];
  }

  float *container_data() override { return (float *)sp->container.data(); }

  void eval(const float *crd, long n, float *out) override {
    auto ev = zimt::make_safe_evaluator<spl_t, float, L>(*sp);
    for (long i0 = 0; i0 < n; i0 += L) {
      crd_v c;
      px_v p;
      for (std::size_t l = 0; l < L; l++) {
        long i = i0 + l < n ? i0 + l : n - 1;
        c[0][l] = crd[2 * i];
        c[1][l] = crd[2 * i + 1];
      }
      ev.eval(c, p);
      for (std::size_t l = 0; l < L && i0 + long(l) < n; l++)
        for (int ch = 0; ch < NCH; ch++) out[(i0 + l) * NCH + ch] = p[ch][l];
    }
  }

  // zimt::process over a w x h raster: discrete coordinate -> affine map
  // (x' = a0 + a1*x, y' = a2 + a3*y) -> safe evaluator -> storer. Exercises
  // the driver's segmentation and the leftover ("cap") paths.
  struct affine_t : public zimt::unary_functor<zimt::xel_t<float, 2>, crd_t, L> {
    float a[4];
    template <typename I, typename O>
    void eval(const I &in, O &out) const {
      out[0] = in[0] * a[1] + a[0];
      out[1] = in[1] * a[3] + a[2];
    }
  };

  void process_affine(long w, long h, const float *aff, float *out) override {
    auto ev = zimt::make_safe_evaluator<spl_t, float, L>(*sp);
    affine_t af;
    for (int i = 0; i < 4; i++) af.a[i] = aff[i];
    zimt::view_t<2, px_t> trg((px_t *)out, {1L, w},
                              {std::size_t(w), std::size_t(h)});
    zimt::storer<float, NCH, 2, L> st(trg);
    zimt::get_crd<float, 2, 2, L> gc;
    zimt::process(trg.shape, gc, af + ev, st);
  }
};

}  // namespace

extern "C" {

int ref_lanes() { return int(L); }
int ref_segment() { return int(zimt::bill_t().segment_size); }

// bc codes are passed as zimt numbers them (zimt/common.h:82-91):
// MIRROR 0, PERIODIC 1, REFLECT 2, NATURAL 3, CONSTANT 4, ZEROPAD 5, GUESS 6

void *ref_bspline_new(const float *core, long W, long H, int nch, int degree,
                      int bc0, int bc1) {
  switch (nch) {
    case 1: return new handle<1>(core, W, H, degree, bc0, bc1);
    case 2: return new handle<2>(core, W, H, degree, bc0, bc1);
    case 3: return new handle<3>(core, W, H, degree, bc0, bc1);
    case 4: return new handle<4>(core, W, H, degree, bc0, bc1);
  }
  return nullptr;
}
void ref_bspline_free(void *h) { delete (handle_base *)h; }
void ref_bspline_prefilter(void *h, int d) { ((handle_base *)h)->prefilter(d); }
void ref_bspline_spherical(void *h, int d) { ((handle_base *)h)->spherical(d); }
void ref_bspline_brace(void *h, int axis) { ((handle_base *)h)->brace(axis); }
void ref_bspline_geometry(void *h, long *o) { ((handle_base *)h)->geometry(o); }
void ref_bspline_container(void *h, float *out) {
  auto *b = (handle_base *)h;
  long g[10];
  b->geometry(g);
  std::memcpy(out, b->container_data(), sizeof(float) * g[0] * g[1] * b->nch);
}
void ref_bspline_eval(void *h, const float *crd, long n, float *out) {
  ((handle_base *)h)->eval(crd, n, out);
}
void ref_bspline_process_affine(void *h, long w, long hh, const float *aff,
                                float *out) {
  ((handle_base *)h)->process_affine(w, hh, aff, out);
}

// b-spline basis weights for one delta (basis.h:650-690), math type float
void ref_basis_weights(int degree, float delta, float *w) {
  zimt::basis_functor<float> bf(degree);
  bf(w, delta);
}

// prefilter poles and the weight matrix are long double in the reference
void ref_poles(int degree, long double *out) {
  for (int i = 0; i < degree / 2; i++)
    out[i] = zimt_constants::precomputed_poles[degree][i];
}

// zimt::prefilter over a plain (frameless) w x h image, both axes, in place:
// what cubemap_t::prefilter does per IR section with NATURAL x NATURAL
// (cubemap.h:921-946), default tolerance
void ref_filter_2d(float *data, long w, long h, int nch, int degree, int bc0,
                   int bc1) {
#define REF_F2D(N)                                                          \
  {                                                                         \
    typedef zimt::xel_t<float, N> px_t;                                     \
    zimt::view_t<2, px_t> v((px_t *)data, {1L, w},                          \
                            {std::size_t(w), std::size_t(h)});              \
    zimt::prefilter(v, v, {zimt::bc_code(bc0), zimt::bc_code(bc1)}, degree); \
  }
  switch (nch) {
    case 1: REF_F2D(1) break;
    case 2: REF_F2D(2) break;
    case 3: REF_F2D(3) break;
    case 4: REF_F2D(4) break;
  }
#undef REF_F2D
}

// The zimt calls lut_based_tf makes (envutil_payload.cc:251-287): a 1-D
// degree-1 bspline<float,1> with NATURAL boundary over `size` knots,
// prefilter(), make_safe_evaluator, evaluated at in * float(size - 1) on
// 16-lane vectors. `knots` come from the caller (the sRGB curve needs no zimt).
void ref_lut_eval(const float *knots, long size, const float *in, long n, float *out) {
  zimt::bspline<float, 1> lut(size, 1, zimt::NATURAL);
  for (long i = 0; i < size; i++) lut.core[i] = knots[i];
  lut.prefilter();
  auto gev = zimt::make_safe_evaluator<zimt::bspline<float, 1>, float, L>(lut);
  typedef zimt::simdized_type<float, L> f_v;
  for (long i = 0; i < n; i += L) {
    f_v v(0.0f), r;
    for (std::size_t l = 0; l < L && i + long(l) < n; l++) v[l] = in[i + l];
    gev.eval(v * float(size - 1), r);
    for (std::size_t l = 0; l < L && i + long(l) < n; l++) out[i + l] = r[l];
  }
}

}  // extern "C"

// lcp<float, L>::eval (lens_correction.h:224-235), constructed the way pto_planar does
// (environment.h:247-252: the facet's double a, b, c narrow at the call), evaluated on
// zimt's 16-lane vectors as the pixel pipeline does; n is padded internally
extern "C" void ref_lcp_factor(double a, double b, double c, const float *x, long n, float *out)
{
  project::lcp<float, L> f(a, b, c, 0.0);
  typedef zimt::simdized_type<float, L> f_v;
  for (long i0 = 0; i0 < n; i0 += (long)L) {
    f_v v(0.0f), r;
    for (std::size_t l = 0; l < L && i0 + (long)l < n; l++) v[l] = x[i0 + l];
    f.eval(v, r);
    for (std::size_t l = 0; l < L && i0 + (long)l < n; l++) out[i0 + l] = r[l];
  }
}

// inverse_lcp<float, L> (lens_correction.h:236-301) as pto_planar<T, L, true> builds it
// (environment.h:247-252: sz = 100): the spline model of the inverse of the radial factor,
// evaluated on 16-lane vectors; knots = the prefiltered coefficients of the model's core
extern "C" void ref_inverse_lcp(double a, double b, double c, double r_max, int sz, const float *x,
                                long n, float *out, float *knots, int max_knots)
{
  project::inverse_lcp<float, L> f(a, b, c, r_max, sz);
  typedef zimt::simdized_type<float, L> f_v;
  for (long i0 = 0; i0 < n; i0 += (long)L) {
    f_v v(0.0f), r;
    for (std::size_t l = 0; l < L && i0 + (long)l < n; l++) v[l] = x[i0 + l];
    f.eval(v, r);
    for (std::size_t l = 0; l < L && i0 + (long)l < n; l++) out[i0 + l] = r[l];
  }
  for (int i = 0; knots && i < f.nk && i < max_knots; i++) knots[i] = f.inv_model.core[i];
}

// zimt::convolve on a w x h float plane, in place, with the call of environment.h:833-843:
// binomial 1 4 6 4 1 / 16, headroom 2, REFLECT on both axes, all axes
extern "C" void ref_binomial_alpha(float *plane, long w, long h)
{
  zimt::view_t<2, float> alpha(plane, {1L, w}, {std::size_t(w), std::size_t(h)});
  zimt::convolve(alpha, alpha, {zimt::REFLECT, zimt::REFLECT},
                 {1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0}, 2);
}

// ---------------------------------------------------------------------------------------------------
// round 3: masking.h and pto.h compile in place as well
// ---------------------------------------------------------------------------------------------------

// masking_t<2, C, L>: the paint, unconditionally (masking.h:74-93); n pixels of C channels
template <int C> static void masking_run(float paint, long n, float *out)
{
  project::masking_t<2, C, L> f(paint);
  typedef zimt::simdized_type<zimt::xel_t<float, 2>, L> in_v;
  typedef zimt::simdized_type<zimt::xel_t<float, C>, L> out_v;
  for (long i0 = 0; i0 < n; i0 += long(L)) {
    in_v in; out_v o;
    for (int d = 0; d < 2; d++) for (std::size_t l = 0; l < L; l++) in[d][l] = 0.0f;
    f.eval(in, o);
    for (std::size_t l = 0; l < L && i0 + long(l) < n; l++)
      for (int c = 0; c < C; c++) out[(i0 + l) * C + c] = o[c][l];
  }
}
extern "C" void ref_masking(int nch, float paint, long n, float *out)
{
  switch (nch) {
    case 1: masking_run<1>(paint, n, out); break;
    case 2: masking_run<2>(paint, n, out); break;
    case 3: masking_run<3>(paint, n, out); break;
    case 4: masking_run<4>(paint, n, out); break;
  }
}

// alpha_masking_t<C, L> (masking.h:95-135) over the spline of a handle made by ref_bspline_new, evaluated at
// n spline coordinates (x, y): colour = paint * alpha, alpha kept
template <int C> static void alpha_masking_run(handle_base *hb, float paint, const float *crd, long n, float *out)
{
  auto *h = static_cast<handle<C> *>(hb);
  typedef typename handle<C>::spl_t spl_t;
  std::shared_ptr<spl_t> sp(h->sp.get(), [](spl_t *) {});      // the handle keeps owning the spline
  project::alpha_masking_t<C, L> f(paint, sp);
  typedef zimt::simdized_type<zimt::xel_t<float, 2>, L> in_v;
  typedef zimt::simdized_type<zimt::xel_t<float, C>, L> out_v;
  for (long i0 = 0; i0 < n; i0 += long(L)) {
    in_v in; out_v o;
    for (std::size_t l = 0; l < L; l++) {
      const long i = i0 + long(l) < n ? i0 + long(l) : n - 1;
      in[0][l] = crd[2 * i]; in[1][l] = crd[2 * i + 1];
    }
    f.eval(in, o);
    for (std::size_t l = 0; l < L && i0 + long(l) < n; l++)
      for (int c = 0; c < C; c++) out[(i0 + l) * C + c] = o[c][l];
  }
}
extern "C" void ref_alpha_masking(void *h, float paint, const float *crd, long n, float *out)
{
  handle_base *hb = (handle_base *)h;
  if (hb->nch == 2) alpha_masking_run<2>(hb, paint, crd, n, out);
  else if (hb->nch == 4) alpha_masking_run<4>(hb, paint, crd, n, out);
}

// pto_parser_type (pto.h:72-180): parse the lines of `text` (one PTO line per '\n') and write the line groups
// in a canonical form - "head<TAB>index<TAB>field=value<TAB>field=value...\n", groups and fields in the order of
// the parser's own std::map - into out (at most cap bytes incl. the terminator). Returns the length needed.
extern "C" long ref_pto_parse(const char *text, char *out, long cap)
{
  pto_parser_type parser;
  std::string all(text), line;
  std::size_t pos = 0;
  while (pos <= all.size()) {
    std::size_t e = all.find('\n', pos);
    if (e == std::string::npos) e = all.size();
    line = all.substr(pos, e - pos);
    parser.parse_pto_line(line);
    pos = e + 1;
  }
  std::string res;
  for (const auto &g : parser.line_group) {
    int idx = 0;
    for (const auto &ln : g.second) {
      res += g.first + "\t" + std::to_string(idx++);
      for (const auto &f : ln.field_map) res += "\t" + f.first + "=" + f.second;
      res += "\n";
    }
  }
  if (out && cap > 0) {
    const long n = (long)res.size() < cap - 1 ? (long)res.size() : cap - 1;
    std::memcpy(out, res.data(), (size_t)n);
    out[n] = 0;
  }
  return (long)res.size() + 1;
}
