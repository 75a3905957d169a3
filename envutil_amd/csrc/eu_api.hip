// C ABI (include/eu_hip.h): host glue between the reference-shaped job
// description and the HIP kernels. No CPU rendering path exists in this
// library: without a HIP device every render/load call fails with
// EU_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <new>
#include "eu_device.h"
#include "eu_setup_math.h"
#include "eu_imageprep.h"
#include "eu_math2.h"

extern "C" int eu_launch_render(const eu_render_params *p, void *stream);
extern "C" int eu_launch_diag(const eu_render_params *p, unsigned long long *stamps_dev, void *stream);
extern "C" int eu_launch_render_multi(const void *p, int degree, void *stream);
extern "C" int eu_launch_render2(const eu_render_params *p, void *stream);
extern "C" int eu_launch_diag_coords(const eu_src_dev *s, const float *rays_dev, long n, int variant,
                                     float *out_dev, void *stream);
extern "C" int eu_launch_render4(const eu_render_params *p, const float *h_row, size_t h_row_floats,
                                 unsigned long long plan_gen, int only_if_worth, void *stream);
extern "C" size_t eu_render4_worklist_ints(size_t ntiles);
extern "C" size_t eu_render4_worklist_header_ints(void);
extern "C" int eu_launch_to_screen(const float *in, long long in_stride, unsigned *out,
                                   long long out_stride, int w, int rows, int nch, const float *lut,
                                   void *stream);

extern "C" int eu_verify_const_div(float c, float limit, void *stream);
extern "C" int eu_launch_selftest(unsigned long long seed, int blocks, int iters,
                                  unsigned long long *bad_dev, void *stream);
extern "C" int eu_launch_prefilter(float *container, const eu_container *g, int nch,
                                   int bc0, int bc1, int prefilter_degree, int spherical,
                                   void *stream);
extern "C" int eu_launch_cubemap_build(const float *faces_dev, float *ir_dev, int nch,
                                       long face_px, long section_px, long left_frame,
                                       long right_frame, double refc_md, double model_to_px,
                                       int prefilter_degree, void *stream);

struct eu_source {
  eu_facet fct;
  eu_container geom;
  int bc[2];
  int degree;
  int nch;
  float *dev;            // braced container in HBM
  size_t nfloats;
  eu_src_dev sd;
  // several devices in one process (eu_hip_init_devices): the device slot this container lives on and its
  // copies on the other slots (made on first use by eu_hip_render_devices, freed with the source)
  int slot = 0;
  eu_source *replica[EU_MAX_SLOTS] = {};
  bool is_replica = false;
};

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

struct context {
  int device = -1;
  hipStream_t stream = nullptr;
  float *col = nullptr, *row = nullptr, *taps = nullptr;
  size_t col_cap = 0, row_cap = 0, taps_cap = 0;
  float *lut = nullptr;        // to_screen_t's sRGB LUT, 257 floats
  float *scr = nullptr; size_t scr_cap = 0;       // float frame of a tethered job
  // host copies of the plan's stepper tables and, from them, the layout the packed
  // kernel should use per segment of EU_SEG_ROWS frame rows (launch-level hybrid)
  std::vector<float> h_col, h_row;
  std::vector<unsigned char> seg_flags;
  bool seg_valid = false, seg_mixed = false;
  unsigned long long plan_gen = 0;                // bumped whenever the stepper tables change
  int tab_finite = 0;                             // every entry of the plan's stepper tables is finite
  unsigned long long launches = 0;                // render kernel launches so far
  eu_src_dev seg_sd;
  float *stage = nullptr; size_t stage_cap = 0;   // host-output staging
  hipStream_t last_user = nullptr;                // caller's stream of the last render (eu_hip_sync waits on it too)
  eu_generic *mgen = nullptr; size_t mgen_cap = 0; // multi-facet jobs: the translated facets' transformations
  float *inv_coef = nullptr;                      // --single: the inverse lens model's coefficients (eu_inv_planar)
  hipEvent_t wl_done = nullptr;                   // behind the last staged launch pair (its work list is free again)
  hipStream_t wl_stream = nullptr; bool wl_stream_set = false;
  hipStream_t copy = nullptr;                     // D2H of a host-output frame, chunk by chunk
  hipEvent_t chunk_done[4] = { nullptr, nullptr, nullptr, nullptr };
  int *wl = nullptr; size_t wl_cap = 0;           // eu_render4.hip work list (count, done, tile ids)
  // the tables of the last target stay valid while (target geometry,
  // orientation, taps) repeat: streaming / tethered jobs re-render the same
  // target many times (envutil_main.cc:1948-1982)
  std::vector<unsigned char> plan_key;
  int plan_form = 0, plan_norm = 0;
  // multi-facet jobs keep their own tables
  float *mcol = nullptr, *mrow = nullptr, *mtaps = nullptr;
  size_t mcol_cap = 0, mrow_cap = 0, mtaps_cap = 0;
  eu_src_dev *msrc = nullptr; size_t msrc_cap = 0;
  float *mrej = nullptr; size_t mrej_cap = 0;      // early-miss tables of a multi-facet job
  std::vector<unsigned char> mplan_key;
  int mplan_form = 0, mplan_norm = 0;
  float *strip = nullptr; size_t strip_cap = 0;   // eu_hip_render_devices: this slot's rows before they are gathered
};
// One context per device SLOT. A process that never calls eu_hip_init_devices has one slot (one process per
// GPU, the set-up bench.py's multi-rank runs use); eu_hip_init_devices makes a slot per listed device, and every
// entry point works on the current one (`g`).
context ctx_[EU_MAX_SLOTS];
int nslots_ = 1, cur_slot_ = 0;
#define g (ctx_[cur_slot_])

#define HIPCHK(call)                                                          \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
      return fail(EU_ERR_NO_DEVICE, std::string(#call ": ") + hipGetErrorString(e_)); \
  } while (0)

int set_slot(int k)
{
  cur_slot_ = k;
  if (ctx_[k].device >= 0) HIPCHK(hipSetDevice(ctx_[k].device));
  return EU_OK;
}

// per-device state: the library's stream and to_screen_t's sRGB LUT. Used by the implicit
// initialisation and by eu_hip_init; switching devices while sources or tables of the old
// device are alive is refused by eu_hip_init.
int init_device(int dev)
{
  HIPCHK(hipSetDevice(dev));
  if (!g.stream) HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  if (!g.lut) {
    float lut[257];
    eu::screen_lut(lut);
    HIPCHK(hipMalloc((void **)&g.lut, sizeof lut));
    HIPCHK(hipMemcpy(g.lut, lut, sizeof lut, hipMemcpyHostToDevice));
  }
  g.device = dev;
  return EU_OK;
}

int ensure_init()
{
  if (g.device >= 0) return EU_OK;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(EU_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  int dev = 0;
  const char *lr = getenv("LOCAL_RANK");
  if (lr) dev = atoi(lr) % n;
  return init_device(dev);
}

int grow(float **p, size_t *cap, size_t need)
{
  if (*cap >= need) return EU_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  HIPCHK(hipMalloc((void **)p, need * sizeof(float)));
  *cap = need;
  return EU_OK;
}

bool is_cube(int prj) { return prj == EU_CUBEMAP || prj == EU_BIATAN6; }

// evaluator + mount parameters (eval.h:2039-2164, environment.h:594-633)
void fill_src_dev(eu_source *s)
{
  eu_src_dev &d = s->sd;
  const eu_facet &f = s->fct;
  memset(&d, 0, sizeof d);
  const eu_container &g0 = s->geom;
  d.base = s->dev + ((size_t)g0.left[1] * g0.shape[0] + g0.left[0]) * s->nch;
  d.es0 = s->nch;
  d.es1 = (long long)s->nch * g0.shape[0];
  d.prj = f.projection; d.nch = s->nch; d.degree = s->degree;
  float lo[2], up[2]; int gt[2];
  for (int a = 0; a < 2; a++) {
    int bc = s->bc[a];
    long double l = 0.0L, u = (long double)(g0.core[a] - 1);
    if (bc == EU_BC_REFLECT || bc == EU_BC_PERIODIC) { l = -0.5L; u += 0.5L; }
    lo[a] = (float)l; up[a] = (float)u;
    if (g0.core[a] == 1) { bc = EU_BC_CONSTANT; lo[a] = up[a] = 0.0f; }
    gt[a] = bc == EU_BC_PERIODIC ? 2 : (bc == EU_BC_MIRROR || bc == EU_BC_REFLECT) ? 1 : 0;
  }
  d.gate0 = gt[0]; d.gate1 = gt[1];
  d.lower0 = lo[0]; d.upper0 = up[0]; d.lower1 = lo[1]; d.upper1 = up[1];
  d.brighten = (float)f.brighten;
  d.mask_paint = f.mask_paint;
  d.recip_step = (float)(1.0 / f.step);
  d.mask_all = is_cube(f.projection) || (f.projection == EU_FISHEYE && f.hfov >= M_PI * 2.0);
  eu::weight_matrix(s->degree, d.wm);
  if (is_cube(f.projection)) return;
  double te[4], we[4];
  eu::get_extent(f.projection, f.width, f.height, f.hfov, te);
  double wx = te[1] - te[0], wy = te[3] - te[2];
  // environment.h:617-633, including its use of total_width / window_width
  // in the y terms
  double px = double(f.window_x_offset) / f.width;
  double py = double(f.window_y_offset) / f.width;
  we[0] = te[0] + px * wx; we[2] = te[2] + py * wy;
  px = double(f.window_x_offset + f.window_width) / f.width;
  py = double(f.window_y_offset + f.window_width) / f.width;
  we[1] = te[0] + px * wx; we[3] = te[2] + py * wy;
  {
    // process_geometry, envutil_basic.h:499-521; the planar functor is only
    // installed when the radial polynomial is present (environment.h:1692-1695)
    double dv = std::fabs(te[3] - te[2]) / 2.0, dh = std::fabs(te[1] - te[0]) / 2.0;
    d.has_lcp = (f.a != 0.0 || f.b != 0.0 || f.c != 0.0);
    d.has_shift = d.has_lcp && (f.h != 0.0 || f.v != 0.0);
    d.has_shear = d.has_lcp && (f.shear_g != 0.0 || f.shear_t != 0.0);
    d.lens_a = (float)f.a; d.lens_b = (float)f.b; d.lens_c = (float)f.c;
    d.lens_d = 1.0f - (d.lens_a + d.lens_b + d.lens_c);
    d.lens_s = (float)((dh < dv) ? dh : dv);
    d.lens_h = (float)f.h; d.lens_v = (float)f.v;
    d.shear_g = f.shear_g; d.shear_t = f.shear_t;
  }
  d.rej_cos = -2.0f;
  if (f.projection == EU_RECTILINEAR) {
    d.rej_cos = 0.0f;                      // the mask includes rz > 0 (environment.h:1135-1137)
  } else if (f.projection == EU_FISHEYE && !d.has_shear) {
    // radius after the lens polynomial: |c'| >= r * |sum(r / s)| - |shift|; outside the
    // window for sure when its larger component (>= |c'| / sqrt 2) exceeds every edge
    const double W = 1.001 * std::max(std::max(std::fabs(we[0]), std::fabs(we[1])),
                                      std::max(std::fabs(we[2]), std::fabs(we[3])));
    const double shift = d.has_shift ? std::hypot((double)d.lens_h, (double)d.lens_v) : 0.0;
    auto outside = [&](double r) {
      double sum = 1.0;
      if (d.has_lcp) {
        const double x = r / d.lens_s;
        sum = d.lens_d + d.lens_c * x + d.lens_b * x * x + d.lens_a * x * x * x;
      }
      return (r * std::fabs(sum) - shift) / std::sqrt(2.0) > W;
    };
    // the smallest angle from which EVERY larger angle is outside
    double r0 = -1.0;
    for (double r = M_PI; r >= 0.0; r -= 1e-4) {
      if (!outside(r)) break;
      r0 = r;
    }
    if (r0 >= 0.0) d.rej_cos = (float)(std::cos(r0) - 1e-4);
  }
  d.tex_x0 = te[0]; d.tex_y0 = te[2];
  d.ext_w = (float)(te[1] - te[0]); d.ext_h = (float)(te[3] - te[2]);
  d.total_w = (float)f.width; d.total_h = (float)f.height;
  d.win_x_off = (float)f.window_x_offset; d.win_y_off = (float)f.window_y_offset;
  d.wex0 = (float)we[0]; d.wex1 = (float)we[1]; d.wex2 = (float)we[2]; d.wex3 = (float)we[3];
  // hits have 0 <= coordinate - extent.x0 <= extent width: verify the cheap
  // constant division over that whole range (twice the width for slack)
  // atan2f returns values in [-0x1.921fb6p+1, 0x1.921fb6p+1]; lat = atan2f(d, s >= 0) in
  // [-0x1.921fb6p+0, 0x1.921fb6p+0]
  d.always_hit = f.projection == EU_SPHERICAL && d.wex0 <= -0x1.921fb6p+1f && d.wex1 >= 0x1.921fb6p+1f
              && d.wex2 <= -0x1.921fb6p+0f && d.wex3 >= 0x1.921fb6p+0f;
  d.rcp_ext_w = 1.0f / d.ext_w; d.rcp_ext_h = 1.0f / d.ext_h;
  d.cdiv_ok = eu_verify_const_div(d.ext_w, 2.0f * d.ext_w, g.stream)
           && eu_verify_const_div(d.ext_h, 2.0f * d.ext_h, g.stream);
}

int check_facet(const eu_facet *f)
{
  if (!f) return fail(EU_ERR_ARGUMENT, "null facet");
  if (f->nchannels < 1 || f->nchannels > 4) return fail(EU_ERR_ARGUMENT, "nchannels must be 1..4");
  if (f->mask_paint < 0 || f->mask_paint > 2) return fail(EU_ERR_ARGUMENT, "mask_paint must be 0, 1 or 2");
  if (f->projection < 0 || f->projection > EU_BIATAN6) return fail(EU_ERR_ARGUMENT, "unknown source projection");
  if (f->width <= 0 || f->height <= 0) return fail(EU_ERR_ARGUMENT, "empty source image");
  return EU_OK;
}

// allocates the eu_source and its container for the facet
int new_source(const eu_facet *fct, int spline_degree, int bc0, int bc1, int support_min,
               int tile_size, eu_source **out)
{
  if (spline_degree < 0 || spline_degree > EU_MAX_DEGREE)
    return fail(EU_ERR_ARGUMENT, "spline degree out of range");
  eu_source *s = new (std::nothrow) eu_source;
  if (!s) return fail(EU_ERR_MEMORY, "host allocation failed");
  memset(s, 0, sizeof *s);
  s->fct = *fct;
  s->degree = spline_degree;
  s->nch = fct->nchannels;
  if (is_cube(fct->projection)) {
    // IR image: container == core, REFLECT x REFLECT (cubemap.h:576-583)
    eu::metrics m = eu::make_metrics(fct->width, fct->hfov, support_min, tile_size);
    // The support frame is all the margin the IR has: a ray at a face's edge picks up at left_frame - 0.5, and a
    // spline of degree d reaches d / 2 + 1 texels beyond that. With less frame than that (--support_min below its
    // default of 8 AND a small --tile_size) the reference reads outside its IR array (README.md:1553: the frame
    // is there "so that interpolators needing support can operate without special-casing"); here the job is refused.
    // Found by the randomised set-up test at seed 1033: biatan6, 45-pixel faces, degree 4, support 1, tile 16 - frame
    // 1 / 2 - where the oracle and the device each read their own memory in front of the array.
    const long need = spline_degree / 2 + 1;
    if (m.left_frame_px + m.inherent_px < need || m.right_frame_px + m.inherent_px < need) {
      delete s;
      return fail(EU_ERR_ARGUMENT, "cubemap support frame (--support_min / --tile_size) too small for the spline degree");
    }
    s->geom.shape[0] = s->geom.core[0] = m.section_px;
    s->geom.shape[1] = s->geom.core[1] = 6 * m.section_px;
    s->geom.left[0] = s->geom.left[1] = s->geom.right[0] = s->geom.right[1] = 0;
    s->bc[0] = s->bc[1] = EU_BC_REFLECT;
  } else {
    eu::container_geometry(spline_degree, bc0, bc1, fct->window_width, fct->window_height, &s->geom);
    s->bc[0] = bc0; s->bc[1] = bc1;
  }
  s->nfloats = (size_t)s->geom.shape[0] * s->geom.shape[1] * s->nch;
  // slack behind the container: the LDS-staging kernel (eu_render4.hip) fetches whole
  // 64-texel instructions, up to 3 rows and 63 texels past a tile's box (never evaluated)
  const size_t slack = (size_t)4 * s->geom.shape[0] * s->nch + 64 * 4;
  hipError_t e = hipMalloc((void **)&s->dev, (s->nfloats + slack) * sizeof(float));
  if (e == hipSuccess) e = hipMemset(s->dev + s->nfloats, 0, slack * sizeof(float));
  if (e != hipSuccess) { delete s; return fail(EU_ERR_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  fill_src_dev(s);
  if (is_cube(fct->projection)) {
    eu::metrics m = eu::make_metrics(fct->width, fct->hfov, support_min, tile_size);
    s->sd.refc_md = (float)m.refc_md;
    s->sd.model_to_px = (float)m.model_to_px;
    s->sd.section_px = (int)m.section_px;
  }
  *out = s;
  return EU_OK;
}

// boundary conditions source_t picks (environment.h:638-644)
void source_bcs(const eu_facet *f, int *bc0, int *bc1)
{
  *bc0 = EU_BC_REFLECT; *bc1 = EU_BC_REFLECT;
  if ((f->projection == EU_SPHERICAL || f->projection == EU_CYLINDRICAL)
      && std::fabs(f->hfov - 2.0 * M_PI) < .000001)
    *bc0 = EU_BC_PERIODIC;
}

// the processed frame: the whole target or its crop window (store_cropped)
inline int frame_w(const eu_target *t) { return t->crop_w > 0 ? t->crop_w : t->width; }
inline int frame_h(const eu_target *t) { return t->crop_w > 0 ? t->crop_h : t->height; }

// rows of the processed frame that belong to this call (interleaved bands)
int local_rows(int height, int band_rows, int band_count, int band_index)
{
  if (band_count <= 1) return height;
  const int nb = (height + band_rows - 1) / band_rows;       // bands of the frame
  int rows = 0;
  for (int b = band_index; b < nb; b += band_count)
    rows += std::min(band_rows, height - b * band_rows);
  return rows;
}

int band_shift_of(int band_rows)
{
  int sh = 0;
  while ((1 << sh) < band_rows) sh++;
  return sh;
}

int check_target(const eu_target *t)
{
  if (t->nchannels < 1 || t->nchannels > 4) return fail(EU_ERR_ARGUMENT, "target channels must be 1..4");
  if (t->width <= 0 || t->height <= 0) return fail(EU_ERR_ARGUMENT, "empty target");
  if (t->crop_w < 0 || (t->crop_w > 0 && (t->crop_h <= 0 || t->crop_x0 < 0 || t->crop_y0 < 0 ||
                        (long long)t->crop_x0 + t->crop_w > t->width ||
                        (long long)t->crop_y0 + t->crop_h > t->height)))
    return fail(EU_ERR_ARGUMENT, "crop window outside the target");
  if (t->band_count > 1) {
    if (t->band_rows < 4 || (t->band_rows & (t->band_rows - 1)))
      return fail(EU_ERR_ARGUMENT, "band_rows must be a power of two >= 4");
    if (t->band_index < 0 || t->band_index >= t->band_count)
      return fail(EU_ERR_ARGUMENT, "band_index outside [0, band_count)");
  }
  if (t->row_begin < 0 || t->row_begin > t->row_end ||
      t->row_end > local_rows(frame_h(t), t->band_rows, t->band_count, t->band_index))
    return fail(EU_ERR_ARGUMENT, "row range outside the target");
  if (t->ntaps < 0 || t->ntaps > EU_MAX_TAPS || (t->ntaps > 0 && !t->taps))
    return fail(EU_ERR_ARGUMENT, "bad twining tap table");
  if ((t->projection == EU_CUBEMAP || t->projection == EU_BIATAN6) && t->height != 6 * t->width)
    return fail(EU_ERR_ARGUMENT, "cubemap targets are 1:6");
  if (t->synopsis != EU_SYN_PANORAMA && t->synopsis != EU_SYN_HDR_MERGE)
    return fail(EU_ERR_ARGUMENT, "unknown synopsis");
  if (t->out_format != EU_OUT_FLOAT && t->out_format != EU_OUT_SRGBA8)
    return fail(EU_ERR_ARGUMENT, "unknown output format");
  if (t->out_format == EU_OUT_SRGBA8 && t->stage)
    return fail(EU_ERR_ARGUMENT, "stage outputs are float only");
  return EU_OK;
}

// tf22 of a --single job: the inverse planar transformation of the facet the target recreates
// (environment.h:285-307); the inverse lens model goes to device memory (g.inv_coef)
int build_inv_planar(const eu_target *t, eu_inv_planar *q)
{
  memset(q, 0, sizeof *q);
  const eu_facet *f = t->single;
  if (!f || !eu::has_2d_tf(*f)) return EU_OK;
  q->shear = f->shear_g != 0.0 || f->shear_t != 0.0;
  q->shift = f->h != 0.0 || f->v != 0.0;
  q->lcp = f->a != 0.0 || f->b != 0.0 || f->c != 0.0;
  q->shear_g = f->shear_g; q->shear_t = f->shear_t;
  { // the reference radius: half the smaller edge of the facet's extent (envutil_basic.h:508-513)
    const double dv0 = std::fabs(t->y1 - t->y0) / 2.0, dh0 = std::fabs(t->x1 - t->x0) / 2.0;
    q->s = (dh0 < dv0) ? dh0 : dv0; }
  q->h = (float)f->h; q->v = (float)f->v;
  if (q->lcp) {
    // r_max of facet_spec::process_geometry (envutil_basic.h:508-520) from the target's (= the facet's) extent
    const double dv = std::fabs(t->y1 - t->y0) / 2.0, dh = std::fabs(t->x1 - t->x0) / 2.0;
    const double aspect = (dh >= dv) ? dh / dv : dv / dh;
    std::vector<float> coef;
    if (!eu::make_inverse_lcp(f->a, f->b, f->c, std::sqrt(1 + aspect * aspect), 100, coef, q->rr_max))
      return fail(EU_ERR_ARGUMENT, "--single: the lens polynomial has no inverse over the facet (the reference asserts here)");
    if (g.last_user) HIPCHK(hipStreamSynchronize(g.last_user));
    if (!g.inv_coef) HIPCHK(hipMalloc((void **)&g.inv_coef, 128 * sizeof(float)));
    HIPCHK(hipMemcpyAsync(g.inv_coef, coef.data(), coef.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    q->nk = (int)coef.size() - 4;
    q->coef = g.inv_coef + 2;
    eu::weight_matrix(3, q->m);
  }
  return EU_OK;
}

int build_params(const eu_target *t, eu_source *const *srcs, int nsrc, float *out_dev,
                 size_t row_stride_bytes, eu_render_params *p)
{
  if (!t || !srcs || !out_dev) return fail(EU_ERR_ARGUMENT, "null argument");
  if (nsrc != 1)
    return fail(EU_ERR_ARGUMENT, "build_params takes one source (several facets go through build_multi)");
  eu_source *s = srcs[0];
  if (!s) return fail(EU_ERR_HANDLE, "null source");
  { int rc0 = check_target(t); if (rc0) return rc0; }
  if (row_stride_bytes % sizeof(float))
    return fail(EU_ERR_ARGUMENT, "row stride must be a multiple of 4 bytes");
  const bool twine = t->ntaps > 0;
  // orientation: envutil_payload.cc:1923-1948
  eu::mat3 r_cam = eu::make_r3(t->roll, t->pitch, t->yaw, false);
  eu::mat3 r_fct = eu::make_r3(s->fct.roll, s->fct.pitch, s->fct.yaw, true);
  eu::mat3 basis = eu::rotate(r_cam, r_fct);
  // plan key: everything the tables depend on
  // a facet with translation parameters: generic_stepper over tf_ex_facet (envutil_payload.cc:2095-2110,
  // :2214-2224), normalised only under twining (deriv_stepper<..., generic_stepper, true>)
  eu_generic gen;
  memset(&gen, 0, sizeof gen);
  if ((eu::has_translation(s->fct) || eu::generic_target(*t)) && !eu::make_generic(*t, s->fct, gen))
    return fail(EU_ERR_UNSUPPORTED, "generic stepper (translation, --single): no planar-to-ray functor for this target projection");
  eu_inv_planar inv;
  { int rci = build_inv_planar(t, &inv); if (rci) return rci; }
  std::vector<unsigned char> key(sizeof(eu_target) + 3 * sizeof(double) + 3 * sizeof(float) * (size_t)t->ntaps);
  {
    eu_target tk = *t;
    tk.taps = nullptr; tk.single = nullptr; tk.row_begin = 0; tk.row_end = 0; tk.stage = 0; tk.nchannels = 0; tk.out_format = 0;
    tk.band_rows = 0; tk.band_count = 0; tk.band_index = 0;     // the tables cover the whole frame
    unsigned char *q = key.data();
    memcpy(q, &tk, sizeof tk); q += sizeof tk;
    double fo[3] = { s->fct.yaw, s->fct.pitch, s->fct.roll };
    memcpy(q, fo, sizeof fo); q += sizeof fo;
    if (twine) memcpy(q, t->taps, 3 * sizeof(float) * (size_t)t->ntaps);
  }
  int rc;
  int form = g.plan_form, norm_mode = g.plan_norm;
  if (key != g.plan_key) {
    eu::stepper_tables tb;
    if (!eu::build_stepper_tables(*t, basis, twine, twine, tb))
      return fail(EU_ERR_UNSUPPORTED, "no stepper for this target projection");
    // a kernel of the previous job may still read the tables on the caller's stream
    if (g.last_user) HIPCHK(hipStreamSynchronize(g.last_user));
    if ((rc = grow(&g.col, &g.col_cap, tb.col.size()))) return rc;
    if ((rc = grow(&g.row, &g.row_cap, tb.row.size()))) return rc;
    HIPCHK(hipMemcpyAsync(g.col, tb.col.data(), tb.col.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.row, tb.row.data(), tb.row.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    std::vector<float> taps;
    if (twine) {
      // twine_t ctor: x, y pre-multiplied by the bias 4.0 (twining.h:106-121)
      taps.assign(t->taps, t->taps + 3 * (size_t)t->ntaps);
      for (int k = 0; k < t->ntaps; k++) { taps[3 * k] *= 4.0f; taps[3 * k + 1] *= 4.0f; }
      if ((rc = grow(&g.taps, &g.taps_cap, taps.size()))) return rc;
      HIPCHK(hipMemcpyAsync(g.taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    }
    // the host vectors die at the end of this block: the copies must have left them
    HIPCHK(hipStreamSynchronize(g.stream));
    form = g.plan_form = tb.form;
    norm_mode = g.plan_norm = tb.norm_mode;
    g.plan_key.swap(key);
    g.h_col.swap(tb.col);
    g.h_row.swap(tb.row);
    g.seg_valid = false;
    g.plan_gen++;
    g.tab_finite = 1;
    for (float v : g.h_col) if (!std::isfinite(v)) g.tab_finite = 0;
    for (float v : g.h_row) if (!std::isfinite(v)) g.tab_finite = 0;
  }
  memset(p, 0, sizeof *p);
  p->tab_finite = g.tab_finite;
  p->width = frame_w(t); p->height = frame_h(t);
  p->row_begin = t->row_begin; p->row_end = t->row_end;
  if (t->band_count > 1) {
    p->band_shift = band_shift_of(t->band_rows); p->band_count = t->band_count; p->band_index = t->band_index;
  }
  p->form = form; p->norm_mode = norm_mode;
  if (gen.on) {                // the tables (planar x per column, planar y per row) are the same
    p->form = EU_FORM_GENERIC;
    p->norm_mode = twine ? EU_NORM_DIV : EU_NORM_NONE;
    p->gen = gen;
    p->inv = inv;
  }
  p->twine = twine; p->ntaps = t->ntaps; p->stage = t->stage; p->nch = s->nch;
  p->nch_out = t->nchannels;
  p->col = g.col; p->row = g.row; p->taps = g.taps;
  p->out = out_dev;
  p->out_stride = (long long)(row_stride_bytes / sizeof(float));
  p->src = s->sd;
  { const char *e = getenv("EU_HIP_DIRECT"); p->direct = (e && e[0] == '1') ? 1 : 0; }
  return EU_OK;
}

// mirror of eu_multi_params (eu_render_multi.hip)
struct multi_params {
  int width, height, row_begin, row_end;
  int form, norm_mode, twine, ntaps, nch, nfct, plus;
  const float *col, *row, *taps;
  const eu_src_dev *srcs;
  float *out;
  long long out_stride;
  int tiles_x, tiles_y;
  int band_shift, band_count, band_index;
  int hdr, hdr_low, hdr_high;
  const eu_generic *gen;
  eu_inv_planar inv;
  const float *rej;
};

// The multi-facet kernels' second early-miss stage (eu_render_multi.hip: eu_multi_maybe): for a fisheye facet
// (no shear) a table over u = cos(angle to the facet's axis), u in [rej_cos, 1], of a lower bound of the
// radius R(theta) = theta * lens polynomial(theta / s) a ray of that angle maps to (environment.h:254-284,
// geometry.h:513-531), and the window's edges moved out by 0.1 %. tab: EU_REJ_STRIDE floats; false: no table.
#define EU_REJ_N 1024
#define EU_REJ_HDR 16
#define EU_REJ_STRIDE (EU_REJ_HDR + EU_REJ_N)
static bool build_reject_table(const eu_src_dev &d, float *tab, bool analytic)
{
  for (int i = 0; i < EU_REJ_STRIDE; i++) tab[i] = 0.0f;
  if (d.prj != EU_FISHEYE || d.has_shear || d.mask_all || !(d.rej_cos > -1.5f) || !(d.rej_cos < 0.999f)) return false;
  const double u0 = d.rej_cos, du = (1.0 - u0) / EU_REJ_N;
  auto radius = [&](double th, bool &okk) {
    double sum = 1.0;
    if (d.has_lcp) {
      const double x = th / d.lens_s;
      sum = d.lens_d + d.lens_c * x + d.lens_b * x * x + d.lens_a * x * x * x;
    }
    if (!(sum > 0.0)) okk = false;
    return th * sum;
  };
  std::vector<double> raw(EU_REJ_N);
  bool okk = true;
  for (int k = 0; k < EU_REJ_N; k++) {
    const double ua = std::min(1.0, std::max(-1.0, u0 + k * du)), ub = std::min(1.0, std::max(-1.0, u0 + (k + 1) * du));
    const double t_hi = std::acos(ua), t_lo = std::acos(ub);
    double m = 1e300;
    for (int j = 0; j <= 16; j++) m = std::min(m, radius(t_lo + (t_hi - t_lo) * j / 16.0, okk));
    raw[k] = m;
  }
  if (!okk) return false;
  for (int k = 0; k < EU_REJ_N; k++) {
    double m = raw[k];
    if (k > 0) m = std::min(m, raw[k - 1]);
    if (k + 1 < EU_REJ_N) m = std::min(m, raw[k + 1]);
    tab[EU_REJ_HDR + k] = (float)(m * (1.0 - 2e-3));
  }
  const double W = std::max(std::max(std::fabs((double)d.wex0), std::fabs((double)d.wex1)),
                            std::max(std::fabs((double)d.wex2), std::fabs((double)d.wex3)));
  const double mg = 1e-3 * W;
  tab[0] = (float)u0; tab[1] = (float)(1.0 / du); tab[2] = 1.0f;
  // the table-free form (EU_HIP_REJ=2): needs R increasing over the cone
  {
    bool mono = true;
    double prev = -1.0;
    const double tmax = std::acos(std::max(-1.0, u0));
    for (int i = 0; i <= 4096 && mono; i++) {
      bool k2 = true;
      const double r = radius(tmax * i / 4096.0, k2);
      mono = k2 && r > prev;
      prev = r;
    }
    if (mono && analytic) {
      tab[2] = 2.0f;
      const double f = 1.0 - 2e-3;
      tab[10] = d.has_lcp ? 1.0f / d.lens_s : 0.0f;
      tab[11] = (float)(f * (d.has_lcp ? d.lens_d : 1.0)); tab[12] = d.has_lcp ? (float)(f * d.lens_c) : 0.0f;
      tab[13] = d.has_lcp ? (float)(f * d.lens_b) : 0.0f; tab[14] = d.has_lcp ? (float)(f * d.lens_a) : 0.0f;
    }
  }
  tab[4] = d.has_shift ? d.lens_h : 0.0f; tab[5] = d.has_shift ? d.lens_v : 0.0f;
  tab[6] = (float)(d.wex0 - mg); tab[7] = (float)(d.wex1 + mg); tab[8] = (float)(d.wex2 - mg); tab[9] = (float)(d.wex3 + mg);
  return true;
}

// fuse() for several facets (envutil_payload.cc:2139-2180, :2240-2281): one
// stepper per facet, all with normalize = true, synopsis by channel count
int build_multi(const eu_target *t, eu_source *const *srcs, int nsrc, float *out_dev,
                size_t row_stride_bytes, multi_params *p, int *degree)
{
  { int rc0 = check_target(t); if (rc0) return rc0; }
  const eu_source *s0 = srcs[0];
  for (int f = 0; f < nsrc; f++) {
    if (!srcs[f]) return fail(EU_ERR_HANDLE, "null source");
    if (srcs[f]->degree != s0->degree) return fail(EU_ERR_ARGUMENT, "facets must share the spline degree");
  }
  const bool twine = t->ntaps > 0;
  std::vector<unsigned char> key(sizeof(eu_target) + (size_t)nsrc * 3 * sizeof(double)
                                 + 3 * sizeof(float) * (size_t)t->ntaps + sizeof(int));
  {
    eu_target tk = *t;
    tk.taps = nullptr; tk.single = nullptr; tk.row_begin = 0; tk.row_end = 0; tk.stage = 0; tk.nchannels = 0; tk.out_format = 0;
    tk.band_rows = 0; tk.band_count = 0; tk.band_index = 0;     // the tables cover the whole frame
    unsigned char *q = key.data();
    memcpy(q, &tk, sizeof tk); q += sizeof tk;
    memcpy(q, &nsrc, sizeof(int)); q += sizeof(int);
    for (int f = 0; f < nsrc; f++) {
      double fo[3] = { srcs[f]->fct.yaw, srcs[f]->fct.pitch, srcs[f]->fct.roll };
      memcpy(q, fo, sizeof fo); q += sizeof fo;
    }
    if (twine) memcpy(q, t->taps, 3 * sizeof(float) * (size_t)t->ntaps);
  }
  int rc;
  if (g.last_user) HIPCHK(hipStreamSynchronize(g.last_user));   // g.msrc and the tables are rewritten below
  if (key != g.mplan_key) {
    eu::mat3 r_cam = eu::make_r3(t->roll, t->pitch, t->yaw, false);
    std::vector<float> rows;
    eu::stepper_tables tb;
    for (int f = 0; f < nsrc; f++) {
      eu::mat3 r_fct = eu::make_r3(srcs[f]->fct.roll, srcs[f]->fct.pitch, srcs[f]->fct.yaw, true);
      eu::mat3 basis = eu::rotate(r_cam, r_fct);
      if (!eu::build_stepper_tables(*t, basis, true, twine, tb))
        return fail(EU_ERR_UNSUPPORTED, "no stepper for this target projection");
      rows.insert(rows.end(), tb.row.begin(), tb.row.end());
    }
    if ((rc = grow(&g.mcol, &g.mcol_cap, tb.col.size()))) return rc;
    if ((rc = grow(&g.mrow, &g.mrow_cap, rows.size()))) return rc;
    HIPCHK(hipMemcpyAsync(g.mcol, tb.col.data(), tb.col.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.mrow, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    std::vector<float> taps;
    if (twine) {
      taps.assign(t->taps, t->taps + 3 * (size_t)t->ntaps);
      for (int k = 0; k < t->ntaps; k++) { taps[3 * k] *= 4.0f; taps[3 * k + 1] *= 4.0f; }
      if ((rc = grow(&g.mtaps, &g.mtaps_cap, taps.size()))) return rc;
      HIPCHK(hipMemcpyAsync(g.mtaps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    g.mplan_form = tb.form; g.mplan_norm = tb.norm_mode;
    g.mplan_key.swap(key);
  }
  // the facets' evaluator parameters (they can change between jobs: always refreshed)
  std::vector<eu_src_dev> sd((size_t)nsrc);
  for (int f = 0; f < nsrc; f++) sd[f] = srcs[f]->sd;
  if (g.msrc_cap < (size_t)nsrc) {
    if (g.msrc) (void)hipFree(g.msrc);
    HIPCHK(hipMalloc((void **)&g.msrc, sizeof(eu_src_dev) * (size_t)nsrc));
    g.msrc_cap = (size_t)nsrc;
  }
  HIPCHK(hipMemcpyAsync(g.msrc, sd.data(), sizeof(eu_src_dev) * (size_t)nsrc, hipMemcpyHostToDevice, g.stream));
  // the early-miss tables of the fisheye facets - OFF unless EU_HIP_REJ=1: measured on config 5 the second
  // stage drops a third of the exact hit tests and the step takes 7.67 instead of 7.30 ms (the table read is
  // one more round trip on a path that waits for memory already)
  bool any_rej = false;
  std::vector<float> rej;
  {
    const char *rje = getenv("EU_HIP_REJ");                  // read on every job
    const bool rej_on = rje && (rje[0] == '1' || rje[0] == '2');       // 2: the table-free form where it applies
    if (rej_on) {
      rej.resize((size_t)nsrc * EU_REJ_STRIDE);
      for (int f = 0; f < nsrc; f++) any_rej |= build_reject_table(sd[f], rej.data() + (size_t)f * EU_REJ_STRIDE, rje[0] == '2');
    }
    if (any_rej) {
      if ((rc = grow(&g.mrej, &g.mrej_cap, rej.size()))) return rc;
      HIPCHK(hipMemcpyAsync(g.mrej, rej.data(), rej.size() * sizeof(float), hipMemcpyHostToDevice, g.stream));
    }
  }
  // facets with translation parameters step through generic_stepper (envutil_payload.cc:2145-2158,
  // :2246-2258); like the evaluator parameters these are refreshed on every job
  std::vector<eu_generic> gv((size_t)nsrc);
  bool any_generic = false;
  for (int f = 0; f < nsrc; f++) {
    memset(&gv[f], 0, sizeof(eu_generic));
    if (!eu::has_translation(srcs[f]->fct) && !eu::generic_target(*t)) continue;
    if (!eu::make_generic(*t, srcs[f]->fct, gv[f]))
      return fail(EU_ERR_UNSUPPORTED, "generic stepper (translation, --single): no planar-to-ray functor for this target projection");
    any_generic = true;
  }
  if (any_generic) {
    if (g.mgen_cap < (size_t)nsrc) {
      if (g.mgen) (void)hipFree(g.mgen);
      g.mgen = nullptr; g.mgen_cap = 0;
      HIPCHK(hipMalloc((void **)&g.mgen, sizeof(eu_generic) * (size_t)nsrc));
      g.mgen_cap = (size_t)nsrc;
    }
    HIPCHK(hipMemcpyAsync(g.mgen, gv.data(), sizeof(eu_generic) * (size_t)nsrc, hipMemcpyHostToDevice, g.stream));
  }
  HIPCHK(hipStreamSynchronize(g.stream));
  eu_inv_planar inv;
  { int rci = build_inv_planar(t, &inv); if (rci) return rci; }
  memset(p, 0, sizeof *p);
  p->gen = any_generic ? g.mgen : nullptr;
  p->rej = any_rej ? g.mrej : nullptr;
  p->inv = inv;
  p->width = frame_w(t); p->height = frame_h(t); p->row_begin = t->row_begin; p->row_end = t->row_end;
  if (t->band_count > 1) {
    p->band_shift = band_shift_of(t->band_rows); p->band_count = t->band_count; p->band_index = t->band_index;
  }
  p->form = g.mplan_form; p->norm_mode = g.mplan_norm; p->twine = twine; p->ntaps = t->ntaps;
  p->nch = t->nchannels; p->nfct = nsrc; p->plus = (t->nchannels == 2 || t->nchannels == 4);
  p->hdr = t->synopsis == EU_SYN_HDR_MERGE;
  {
    // _hdr_merge_syn ctor (envutil_payload.cc:1346-1376): the first strict minimum / maximum of brighten
    float lowest = 100000.0f, highest = -1.0f;
    p->hdr_low = p->hdr_high = -1;
    for (int f = 0; f < nsrc; f++) {
      const float b = srcs[f]->sd.brighten;
      if (b < lowest) { lowest = b; p->hdr_low = f; }
      if (b > highest) { highest = b; p->hdr_high = f; }
    }
  }
  p->col = g.mcol; p->row = g.mrow; p->taps = g.mtaps; p->srcs = g.msrc;
  p->out = out_dev; p->out_stride = (long long)(row_stride_bytes / sizeof(float));
  *degree = s0->degree;
  return EU_OK;
}

// Launch-level hybrid of the packed kernel's two work layouts. Where source rows run
// ACROSS target rows (the inner half of the polar faces of a cubemap made from a lat/lon
// image: 0.098 -> 0.071 ms per 1024 rows) 32x16 tiles beat the 128x4 row strips; everywhere
// else the strips win (0.045 vs 0.055 ms). The frame is cut into segments of EU_SEG_ROWS
// rows; 16 probe pixels per segment, evaluated on the host from the plan's stepper
// tables, say how the source rows run there. Lat/lon sources only.
#define EU_SEG_ROWS 512

void compute_seg_flags(const eu_render_params *p)
{
  const int W = p->width, H = p->height;
  const int nseg = (H + EU_SEG_ROWS - 1) / EU_SEG_ROWS;
  g.seg_flags.assign((size_t)nseg, 0);
  g.seg_mixed = false;
  const eu_src_dev &s = p->src;
  if (s.prj != EU_SPHERICAL || W < 64 || g.h_col.size() < (size_t)2 * W || g.h_row.size() < (size_t)H * EU_ROW_FLOATS)
    return;
  const double kx = (double)s.total_w / s.ext_w, ky = (double)s.total_h / s.ext_h;   // pixels per radian
  auto ray = [&](int x, int y, double *r) {
    const float *rt = &g.h_row[(size_t)y * EU_ROW_FLOATS];
    const float c0 = g.h_col[(size_t)x], c1 = g.h_col[(size_t)W + x];
    for (int i = 0; i < 3; i++)
      r[i] = p->form == EU_FORM_BCA ? (double)rt[3 + i] * c0 + (double)rt[6 + i] * c1 + rt[i]
                                    : (double)rt[3 + i] * c0 + rt[i];
  };
  int any = 0;
  for (int k = 0; k < nseg; k++) {
    const int yc = std::min(k * EU_SEG_ROWS + EU_SEG_ROWS / 2, H - 1);
    int across = 0;
    for (int j = 0; j < 16; j++) {
      const int xc = std::min((int)((j + 0.5) * W / 16), W - 2);
      double a[3], b[3];
      ray(xc, yc, a);
      ray(xc + 1, yc, b);
      const double lon0 = std::atan2(a[0], a[2]), lon1 = std::atan2(b[0], b[2]);
      const double lat0 = std::atan2(a[1], std::hypot(a[0], a[2])), lat1 = std::atan2(b[1], std::hypot(b[0], b[2]));
      double dlon = std::fabs(lon1 - lon0);
      if (dlon > M_PI) dlon = 2.0 * M_PI - dlon;
      if (std::fabs(lat1 - lat0) * ky > 0.5 * dlon * kx) across++;
    }
    g.seg_flags[(size_t)k] = across > 8;
    any += across > 8;
  }
  g.seg_mixed = any > 0;
}

// the packed two-pixel kernel where it applies, the general kernel otherwise
// (EU_HIP_KERNEL=1 forces the general kernel: A/B switch)
int launch_render(const eu_render_params *p, void *st)
{
  static const int force_v1 = [] { const char *e = getenv("EU_HIP_KERNEL"); return e && e[0] == '1'; }();
  static const bool hybrid = [] { const char *e = getenv("EU_HIP_HYBRID"); return !(e && e[0] == '0'); }();
  // The LDS-staging kernel (eu_render4.hip). Measured: it wins where the taps
  // dominate and the tile boxes are small - cubic / quadratic jobs on cubemap sources (config 3:
  // 1.13 -> 0.99 ms) - and loses to the direct-gather kernels on lat/lon sources (headline 1.22
  // vs 1.57 ms: the polar faces' boxes do not fit, the equatorial faces tie) and on bilinear
  // jobs. EU_HIP_R4: 0 never, 1 wherever it applies (tests, A/B runs); read on every call.
  const char *r4env = getenv("EU_HIP_R4");
  const int r4mode = r4env ? atoi(r4env) : -1;
  // round 3: also cubic / quadratic lat/lon jobs whose target has column plans (the headline: an upright cubemap),
  // with the persistent form of the staged kernel (eu_render5_kernel) - eu_launch_render4 declines the others
  const bool fast5 = p->src.prj == EU_SPHERICAL && p->src.degree >= 2 && p->form == EU_FORM_BA && p->norm_mode == EU_NORM_NONE &&
                     p->band_count <= 1 && p->src.brighten == 1.0f && p->src.always_hit && !p->twine;
  const bool use_r4 = r4mode == 1 || (r4mode != 0 && p->src.degree >= 2 && (is_cube(p->src.prj) || fast5));
  // a --mask_for job paints the facet at the inner evaluation: only the general kernels do that
  if (p->src.mask_paint) {
    eu_render_params q = *p;
    q.direct = 1;                       // not the LDS-staged variant: it evaluates inline
    g.launches++;
    return eu_launch_render(&q, st);
  }
  if (!force_v1 && use_r4) {
    // work list of the staged kernel (eu_render4.hip: chunk counters of the persistent kernel,
    // lists of the tiles left to the direct-gather kernel that follows it on the same stream)
    const size_t ntiles = (size_t)((p->width + 15) / 16) * (size_t)((p->row_end - p->row_begin + 7) / 8);
    const size_t need = eu_render4_worklist_ints(ntiles);
    if (g.wl_cap < need) {
      if (g.wl) (void)hipFree(g.wl);
      g.wl = nullptr; g.wl_cap = 0;
      if (hipMalloc((void **)&g.wl, need * sizeof(int)) != hipSuccess) return -1;
      if (hipMemsetAsync(g.wl, 0, eu_render4_worklist_header_ints() * sizeof(int), (hipStream_t)st) != hipSuccess) return -1;
      g.wl_cap = need;
    }
    eu_render_params q = *p;
    q.wl = g.wl;
    // the work list and the persistent kernel's queues belong to ONE launch pair at a time: a job on another
    // stream than the last staged job's waits for that job's event (same stream: stream order does it)
    if (g.wl_stream_set && g.wl_stream != (hipStream_t)st && g.wl_done)
      if (hipStreamWaitEvent((hipStream_t)st, g.wl_done, 0) != hipSuccess) return -1;
    g.launches += 2;
    const int rc = eu_launch_render4(&q, g.h_row.data(), g.h_row.size(), g.plan_gen, r4mode != 1, st);
    if (rc == 0) {
      if (!g.wl_done && hipEventCreateWithFlags(&g.wl_done, hipEventDisableTiming) != hipSuccess) return -1;
      if (hipEventRecord(g.wl_done, (hipStream_t)st) != hipSuccess) return -1;
      g.wl_stream = (hipStream_t)st; g.wl_stream_set = true;
    }
    if (rc <= 0) return rc;
    g.launches -= 2;
  }
  if (!force_v1) {
    // worth it for cubic / quadratic jobs whose rows fall into a few long runs: every run
    // is a launch of its own (a rank's share of a band-interleaved split has many short
    // runs, a 0.2 ms bilinear job little to gain: both lose more to the extra launches
    // than the layout gains); EU_HIP_HYBRID=2 lifts the limits (tests)
    static const bool hybrid_any = [] { const char *e = getenv("EU_HIP_HYBRID"); return e && e[0] == '2'; }();
    const bool worth = hybrid_any || p->src.degree >= 2;
    if (hybrid && worth && !p->twine && p->stage == 0 && p->norm_mode == EU_NORM_NONE && p->src.prj == EU_SPHERICAL) {
      eu_src_dev cmp = p->src;
      cmp.base = nullptr;
      if (!g.seg_valid || memcmp(&cmp, &g.seg_sd, sizeof cmp)) {
        compute_seg_flags(p);
        g.seg_sd = cmp;
        g.seg_valid = true;
      }
      // runs of local rows whose segments want the same layout, in chunks of 64 rows
      auto flag_of = [&](int r) {
        const int fy = eu_frame_row(std::min(r, p->row_end - 1), p->band_shift, p->band_count, p->band_index);
        return (int)g.seg_flags[(size_t)std::min(fy / EU_SEG_ROWS, (int)g.seg_flags.size() - 1)];
      };
      auto run_end = [&](int a, int fl) {
        int b = std::min((a / 64 + 1) * 64, p->row_end);
        while (b < p->row_end && flag_of(b) == fl) b = std::min(b + 64, p->row_end);
        return b;
      };
      int nruns = 0, tiled = 0, shortest = INT_MAX;
      if (g.seg_mixed)
        for (int a = p->row_begin; a < p->row_end; nruns++) {
          const int fl = flag_of(a);
          tiled += fl;
          const int b = run_end(a, fl);
          shortest = std::min(shortest, b - a);
          a = b;
        }
      // a few long runs (the whole frame: 5; a contiguous strip of a split: 1-3), not the
      // many short ones of a band-interleaved share (0.18 -> 0.21 ms when split up)
      if (tiled > 0 && (hybrid_any || nruns <= 3 || (nruns <= 5 && shortest >= EU_SEG_ROWS))) {
        int a = p->row_begin;
        while (a < p->row_end) {
          const int fl = flag_of(a);
          const int b = run_end(a, fl);
          eu_render_params q = *p;
          q.row_begin = a; q.row_end = b;
          q.out = p->out + (long long)(a - p->row_begin) * p->out_stride;
          q.layout = fl ? 2 : 1;
          g.launches++;
          const int rc = eu_launch_render2(&q, st);
          if (rc > 0) {                 // not a packed-kernel job after all: one ordinary launch
            if (a != p->row_begin) return -1;
            return eu_launch_render(p, st);
          }
          if (rc < 0) return rc;
          a = b;
        }
        return 0;
      }
    }
    g.launches++;
    int rc = eu_launch_render2(p, st);
    if (rc <= 0) return rc;
    g.launches--;
  }
  g.launches++;
  return eu_launch_render(p, st);
}

}  // namespace

extern "C" {

const char *eu_hip_last_error(void) { return g_err.c_str(); }

int eu_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int eu_hip_init(int device)
{
  int n = eu_hip_device_count();
  if (n <= 0) return fail(EU_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= n) return fail(EU_ERR_ARGUMENT, "device index out of range");
  if (g.device == device) return EU_OK;
  // one device per process (one process per GPU): the stream, the LUT, the plan tables and
  // every resident source live on the device chosen first
  if (g.device >= 0)
    return fail(EU_ERR_ARGUMENT, "the library is already initialised on another device");
  return init_device(device);
}

int eu_hip_get_extent(int prj, int w, int h, double hfov, double *e)
{
  if (!e || prj < 0 || prj > EU_BIATAN6) return fail(EU_ERR_ARGUMENT, "bad projection");
  eu::get_extent(prj, w, h, hfov, e);
  return EU_OK;
}

double eu_hip_get_step(int prj, int w, int h, double hfov) { return eu::get_step(prj, w, h, hfov); }

int eu_hip_make_spread(int w, int h, float d, float sigma, float threshold, float *taps, int max_taps)
{
  std::vector<float> v;
  int n = eu::make_spread(w, h, d, sigma, threshold, v);
  if (n > max_taps) return fail(EU_ERR_ARGUMENT, "tap buffer too small");
  memcpy(taps, v.data(), v.size() * sizeof(float));
  return n;
}

int eu_hip_facet_alpha(float *pixels, int width, int height, int nchannels, const eu_mask_polygon *polygons,
                       int npolygons, int crop_kind, int crop_x0, int crop_x1, int crop_y0, int crop_y1,
                       float *alpha_out)
{
  if (width <= 0 || height <= 0 || (nchannels != 2 && nchannels != 4) || (!pixels && !alpha_out))
    return fail(EU_ERR_ARGUMENT, "facet_alpha: width x height x {2, 4} channels");
  if (npolygons < 0 || (npolygons > 0 && !polygons) || crop_kind < 0 || crop_kind > 2)
    return fail(EU_ERR_ARGUMENT, "facet_alpha: polygons / crop kind");
  std::vector<eu::mask_polygon> ps;
  for (int i = 0; i < npolygons; i++) {
    if (polygons[i].n < 0 || (polygons[i].n > 0 && (!polygons[i].x || !polygons[i].y)))
      return fail(EU_ERR_ARGUMENT, "facet_alpha: polygon without vertices");
    ps.push_back({ polygons[i].n, polygons[i].x, polygons[i].y });
  }
  std::vector<float> alpha;
  try { alpha.resize(size_t(width) * height); } catch (...) { return fail(EU_ERR_MEMORY, "facet_alpha: host memory"); }
  eu::facet_alpha(alpha.data(), width, height, ps.data(), int(ps.size()), crop_kind, crop_x0, crop_x1, crop_y0, crop_y1);
  if (pixels)
    eu::parallel_rows(height, [&](int y0, int y1) {
      for (size_t i = size_t(y0) * width; i < size_t(y1) * width; i++)
        for (int c = 0; c < nchannels; c++) pixels[i * nchannels + c] = pixels[i * nchannels + c] * alpha[i];
    });
  if (alpha_out) memcpy(alpha_out, alpha.data(), alpha.size() * sizeof(float));
  return EU_OK;
}

int eu_hip_cubemap_metrics(int face_px, double face_fov, int support_min, int tile_px,
                           int64_t *section_px, int64_t *left_frame_px, double *refc_md,
                           double *model_to_px)
{
  if (face_px <= 0 || tile_px <= 0 || (tile_px & (tile_px - 1)) || face_fov < M_PI_2 - 1e-12)
    return fail(EU_ERR_ARGUMENT, "bad cubemap metrics arguments");
  eu::metrics m = eu::make_metrics(face_px, face_fov, support_min, tile_px);
  if (section_px) *section_px = m.section_px;
  if (left_frame_px) *left_frame_px = m.left_frame_px;
  if (refc_md) *refc_md = m.refc_md;
  if (model_to_px) *model_to_px = m.model_to_px;
  return EU_OK;
}

int eu_hip_container_geometry(int degree, int bc0, int bc1, int64_t w, int64_t h, eu_container *out)
{
  if (!out || degree < 0 || degree > EU_MAX_DEGREE || w <= 0 || h <= 0)
    return fail(EU_ERR_ARGUMENT, "bad container geometry arguments");
  eu::container_geometry(degree, bc0, bc1, w, h, out);
  return EU_OK;
}

int eu_hip_source_adopt(const eu_facet *fct, const float *container, int spline_degree,
                        int bc0, int bc1, int support_min, int tile_size, eu_source **out)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if ((rc = check_facet(fct))) return rc;
  if (!container || !out) return fail(EU_ERR_ARGUMENT, "null argument");
  eu_source *s = nullptr;
  if ((rc = new_source(fct, spline_degree, bc0, bc1, support_min, tile_size, &s))) return rc;
  hipError_t e = hipMemcpy(s->dev, container, s->nfloats * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(s->dev); delete s; return fail(EU_ERR_NO_DEVICE, hipGetErrorString(e)); }
  *out = s;
  return EU_OK;
}

int eu_hip_source_load(const eu_facet *fct, const float *pixels, int spline_degree,
                       int prefilter_degree, int support_min, int tile_size, eu_source **out)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if ((rc = check_facet(fct))) return rc;
  if (!pixels || !out) return fail(EU_ERR_ARGUMENT, "null argument");
  if (prefilter_degree < 0 || prefilter_degree > EU_MAX_DEGREE)
    return fail(EU_ERR_ARGUMENT, "prefilter degree out of range");
  int bc0, bc1;
  source_bcs(fct, &bc0, &bc1);
  eu_source *s = nullptr;
  if ((rc = new_source(fct, spline_degree, bc0, bc1, support_min, tile_size, &s))) return rc;
  // (a core narrower than the spline's frame on an axis is braced slice by slice in zimt's order,
  // eu_setup.hip: brace_seq_kernel)
  const int nch = s->nch;
  hipError_t e = hipSuccess;
  if (is_cube(fct->projection)) {
    eu::metrics m = eu::make_metrics(fct->width, fct->hfov, support_min, tile_size);
    size_t nface = (size_t)6 * m.face_px * m.face_px * nch;
    float *faces = nullptr;
    e = hipMalloc((void **)&faces, nface * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(faces, pixels, nface * sizeof(float), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) {
      rc = eu_launch_cubemap_build(faces, s->dev, nch, m.face_px, m.section_px, m.left_frame_px,
                                   m.right_frame_px, m.refc_md, m.model_to_px, prefilter_degree,
                                   g.stream);
      e = hipStreamSynchronize(g.stream);
    }
    if (faces) (void)hipFree(faces);
  } else {
    // pixels -> core of the container (2-D strided copy), then prefilter + brace
    const eu_container &gm = s->geom;
    e = hipMemsetAsync(s->dev, 0, s->nfloats * sizeof(float), g.stream);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(s->dev + ((size_t)gm.left[1] * gm.shape[0] + gm.left[0]) * nch,
                           (size_t)gm.shape[0] * nch * sizeof(float), pixels,
                           (size_t)gm.core[0] * nch * sizeof(float),
                           (size_t)gm.core[0] * nch * sizeof(float), (size_t)gm.core[1],
                           hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) {
      // source_t ctor, environment.h:905-936: full spherical images get the
      // two-axis periodic scheme, everything else bspline::prefilter()
      int spherical = fct->projection == EU_SPHERICAL && std::fabs(fct->hfov - 2.0 * M_PI) < .000001
                      && fct->width == 2 * fct->height;
      // (a full-sphere image smaller than its frame - 2 x 1, 4 x 2, 6 x 3 for degree 3 - takes the sequential forms of
      // the pole rows and of the horizontal bracing: eu_setup.hip, pole_rows_seq_kernel / brace_seq_kernel)
      rc = eu_launch_prefilter(s->dev, &s->geom, nch, bc0, bc1, prefilter_degree, spherical, g.stream);
      e = hipStreamSynchronize(g.stream);
    }
  }
  if (e != hipSuccess || rc) {
    (void)hipFree(s->dev);
    delete s;
    if (e != hipSuccess) return fail(EU_ERR_NO_DEVICE, hipGetErrorString(e));
    return fail(rc, "device set-up stage failed");
  }
  *out = s;
  return EU_OK;
}

int eu_hip_source_alloc(const eu_facet *fct, int spline_degree, int support_min, int tile_size,
                        eu_source **out)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if ((rc = check_facet(fct))) return rc;
  if (!out) return fail(EU_ERR_ARGUMENT, "null argument");
  int bc0, bc1;
  source_bcs(fct, &bc0, &bc1);
  return new_source(fct, spline_degree, bc0, bc1, support_min, tile_size, out);
}

int eu_hip_source_update_facet(eu_source *src, const eu_facet *fct)
{
  int rc;
  if (!src) return fail(EU_ERR_HANDLE, "null source");
  if ((rc = check_facet(fct))) return rc;
  const eu_facet &o = src->fct;
  // what the resident container was built from stays as it is
  if (fct->projection != o.projection || fct->nchannels != o.nchannels || fct->width != o.width ||
      fct->height != o.height || fct->window_width != o.window_width || fct->window_height != o.window_height)
    return fail(EU_ERR_ARGUMENT, "the facet's image (projection, size, channels) differs from the resident one");
  if (is_cube(o.projection) && fct->hfov != o.hfov)
    return fail(EU_ERR_ARGUMENT, "a cubemap's field of view is part of its resident image");
  const eu_src_dev keep = src->sd;
  src->fct = *fct;
  fill_src_dev(src);
  if (is_cube(o.projection)) {
    src->sd.refc_md = keep.refc_md; src->sd.model_to_px = keep.model_to_px; src->sd.section_px = keep.section_px;
  }
  return EU_OK;
}

int eu_hip_source_device_ptr(const eu_source *src, void **dev_ptr, size_t *nfloats)
{
  if (!src) return fail(EU_ERR_HANDLE, "null source");
  if (dev_ptr) *dev_ptr = src->dev;
  if (nfloats) *nfloats = src->nfloats;
  return EU_OK;
}

int eu_hip_source_download(const eu_source *src, float *container, size_t nfloats)
{
  if (!src || !container) return fail(EU_ERR_HANDLE, "null argument");
  if (nfloats != src->nfloats) return fail(EU_ERR_ARGUMENT, "container size mismatch");
  HIPCHK(hipMemcpy(container, src->dev, nfloats * sizeof(float), hipMemcpyDeviceToHost));
  return EU_OK;
}

int eu_hip_source_info(const eu_source *src, eu_container *geom, int *nch)
{
  if (!src) return fail(EU_ERR_HANDLE, "null source");
  if (geom) *geom = src->geom;
  if (nch) *nch = src->nch;
  return EU_OK;
}

int eu_hip_source_release(eu_source *src)
{
  if (!src) return EU_OK;
  for (int k = 0; k < EU_MAX_SLOTS; k++)
    if (src->replica[k]) {
      if (src->replica[k]->dev) (void)hipFree(src->replica[k]->dev);
      delete src->replica[k];
    }
  if (src->dev) (void)hipFree(src->dev);
  delete src;
  return EU_OK;
}

// one job, everything on the device: float pixels straight into out_dev, or -
// tethered - float pixels into the library's frame buffer followed by the
// to_screen_t pass that writes the packed words to out_dev
static int render_on_device(const eu_target *trg, eu_source *const *srcs, int nsrc, float *out_dev,
                            size_t stride_bytes, hipStream_t st)
{
  int rc;
  const bool multi = nsrc > 1;
  const bool screen = trg->out_format == EU_OUT_SRGBA8;
  eu_target tf = *trg;
  float *fout = out_dev;
  size_t fstride = stride_bytes;
  const size_t rows = (size_t)(trg->row_end - trg->row_begin);
  if (screen) {
    if (!g.lut) return fail(EU_ERR_NO_DEVICE, "sRGB table missing: library not initialised");
    tf.out_format = EU_OUT_FLOAT;
    fstride = (size_t)frame_w(trg) * trg->nchannels * sizeof(float);
    if ((rc = grow(&g.scr, &g.scr_cap, rows * frame_w(trg) * trg->nchannels))) return rc;
    fout = g.scr;
  }
  if (multi) {
    multi_params mp;
    int mdeg = 0;
    if ((rc = build_multi(&tf, srcs, nsrc, fout, fstride, &mp, &mdeg))) return rc;
    if (eu_launch_render_multi(&mp, mdeg, st)) return fail(EU_ERR_NO_DEVICE, "kernel launch failed");
  } else {
    eu_render_params p;
    if ((rc = build_params(&tf, srcs, nsrc, fout, fstride, &p))) return rc;
    if (launch_render(&p, st)) return fail(EU_ERR_NO_DEVICE, "kernel launch failed");
  }
  if (screen &&
      eu_launch_to_screen(fout, (long long)(fstride / sizeof(float)), (unsigned *)out_dev,
                          (long long)(stride_bytes / sizeof(unsigned)), frame_w(trg), (int)rows,
                          trg->nchannels, g.lut, st))
    return fail(EU_ERR_NO_DEVICE, "kernel launch failed");
  return EU_OK;
}

int eu_hip_render(const eu_target *trg, eu_source *const *srcs, int nsrc, float *out,
                  size_t out_row_stride_bytes, int out_on_device, void *stream)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if (!trg) return fail(EU_ERR_ARGUMENT, "null target");
  if (!srcs || nsrc < 1 || !out) return fail(EU_ERR_ARGUMENT, "no source / no output");
  if ((rc = check_target(trg))) return rc;
  // words per pixel: a packed sRGBA8 word, 3 floats of a stage output, or the channels
  const int och = trg->out_format == EU_OUT_SRGBA8 ? 1 : trg->stage ? 3 : trg->nchannels;
  const size_t min_stride = (size_t)frame_w(trg) * och * sizeof(float);
  if (out_row_stride_bytes < min_stride) return fail(EU_ERR_ARGUMENT, "row stride smaller than a row");
  if (out_row_stride_bytes % sizeof(float)) return fail(EU_ERR_ARGUMENT, "row stride must be a multiple of 4 bytes");
  if (nsrc > 1 && trg->stage) return fail(EU_ERR_ARGUMENT, "stage outputs exist for single-facet jobs only");
  for (int f = 0; f < nsrc; f++) {
    if (!srcs[f]) return fail(EU_ERR_HANDLE, "null source");
    // a masking job adapts channel counts with mono_t, which knows 1 and 2 output channels only
    // (environment.h:1339: the reference asserts)
    if (srcs[f]->fct.mask_paint && srcs[f]->nch != trg->nchannels && trg->nchannels > 2 && !trg->stage)
      return fail(EU_ERR_ARGUMENT, "--mask_for: facets whose channel count differs from the target's need a 1- or 2-channel target");
  }
  hipStream_t st = stream ? (hipStream_t)stream : g.stream;
  // g.last_user still names the PREVIOUS job's stream while this job is set up: build_params / build_multi wait
  // for it before they rewrite tables that job may be reading (it used to be overwritten here, so a job on another
  // stream rewrote the tables under the previous one: tests/test_gpu_round3_switches.py)
  struct note_stream { hipStream_t s; ~note_stream() { g.last_user = s; } } note_on_exit{ stream ? (hipStream_t)stream : nullptr };
  if (out_on_device) return render_on_device(trg, srcs, nsrc, out, out_row_stride_bytes, st);
  const size_t rows = (size_t)(trg->row_end - trg->row_begin);
  if (!rows) return EU_OK;
  if ((rc = grow(&g.stage, &g.stage_cap, rows * frame_w(trg) * och))) return rc;
  // The frame goes to the host in up to four row chunks: every chunk is a launch of its own
  // on `st`, and its copy (second stream, behind the chunk's event) runs while the later
  // chunks render - the link (57 GB/s pinned, 21 ms for the 1.2 GB headline frame) is the
  // whole cost, the kernels hide under the first copy.
  const size_t nchunk = rows >= 1024 ? 4 : 1;
  if (!g.copy) HIPCHK(hipStreamCreateWithFlags(&g.copy, hipStreamNonBlocking));
  for (size_t c = 0; c < 4; c++)
    if (!g.chunk_done[c]) HIPCHK(hipEventCreateWithFlags(&g.chunk_done[c], hipEventDisableTiming));
  const size_t per = ((rows + nchunk - 1) / nchunk + 7) / 8 * 8;
  // whatever happens below, nothing of this call may still write into `out` or read g.stage when it returns
  struct drain { hipStream_t a, b; ~drain() { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); } } drain_on_exit{ g.copy, st };
  for (size_t c = 0; c < nchunk; c++) {
    const size_t a = std::min(rows, c * per), b = std::min(rows, (c + 1) * per);
    if (a >= b) break;
    eu_target tc = *trg;
    tc.row_begin = trg->row_begin + (int)a;
    tc.row_end = trg->row_begin + (int)b;
    float *dst = g.stage + a * frame_w(trg) * och;
    if ((rc = render_on_device(&tc, srcs, nsrc, dst, min_stride, st))) return rc;
    HIPCHK(hipEventRecord(g.chunk_done[c], st));
    HIPCHK(hipStreamWaitEvent(g.copy, g.chunk_done[c], 0));
    HIPCHK(hipMemcpy2DAsync((char *)out + a * out_row_stride_bytes, out_row_stride_bytes, dst, min_stride,
                            min_stride, b - a, hipMemcpyDeviceToHost, g.copy));
  }
  HIPCHK(hipStreamSynchronize(g.copy));
  HIPCHK(hipStreamSynchronize(st));
  return EU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// one process, several devices (include/eu_hip.h)
// ---------------------------------------------------------------------------------------------------------
int eu_hip_device_slots(void) { return nslots_; }
int eu_current_slot(void) { return cur_slot_; }

int eu_hip_init_devices(const int *devices, int ndevices)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(EU_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (!devices || ndevices < 1 || ndevices > EU_MAX_SLOTS) return fail(EU_ERR_ARGUMENT, "1 .. EU_MAX_SLOTS devices");
  for (int k = 0; k < ndevices; k++)
    if (devices[k] < 0 || devices[k] >= n) return fail(EU_ERR_ARGUMENT, "device index out of range");
  // slot 0 may already be initialised (sources exist on it): it keeps its device
  if (ctx_[0].device >= 0 && ctx_[0].device != devices[0])
    return fail(EU_ERR_ARGUMENT, "the library is already initialised on another device than devices[0]");
  for (int k = 1; k < nslots_; k++)
    if (k >= ndevices || ctx_[k].device != devices[k])
      return fail(EU_ERR_ARGUMENT, "device slots cannot be re-assigned once made");
  int rc = EU_OK;
  for (int k = 0; k < ndevices && !rc; k++) {
    cur_slot_ = k;
    if (ctx_[k].device < 0) rc = init_device(devices[k]);
    // peers: slot k reads slot 0's containers and slot 0 receives the strips (a no-op for the same device)
    if (!rc && devices[k] != devices[0]) {
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, devices[k], devices[0]);
      if (can) { hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0); if (e != hipSuccess) (void)hipGetLastError(); }
    }
  }
  nslots_ = std::max(nslots_, ndevices);
  int rc2 = set_slot(0);
  return rc ? rc : rc2;
}

namespace {

// the source's copy on slot k (made on first use: one peer copy of the braced container)
int replica_of(eu_source *src, int k, eu_source **out)
{
  if (k == src->slot) { *out = src; return EU_OK; }
  if (!src->replica[k]) {
    eu_source *r = new (std::nothrow) eu_source(*src);
    if (!r) return fail(EU_ERR_NO_DEVICE, "out of host memory");
    for (int j = 0; j < EU_MAX_SLOTS; j++) r->replica[j] = nullptr;
    r->is_replica = true; r->slot = k; r->dev = nullptr;
    int rc = set_slot(k);
    if (rc) { delete r; return rc; }
    hipError_t e = hipMalloc((void **)&r->dev, src->nfloats * sizeof(float));
    if (e == hipSuccess) {
      if (ctx_[k].device == ctx_[src->slot].device)
        e = hipMemcpyAsync(r->dev, src->dev, src->nfloats * sizeof(float), hipMemcpyDeviceToDevice, g.stream);
      else
        e = hipMemcpyPeerAsync(r->dev, ctx_[k].device, src->dev, ctx_[src->slot].device, src->nfloats * sizeof(float), g.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) { if (r->dev) (void)hipFree(r->dev); delete r; return fail(EU_ERR_NO_DEVICE, hipGetErrorString(e)); }
    // the evaluator parameters are the original's (same geometry, same verified constants) on another base
    r->sd.base = r->dev + (src->sd.base - src->dev);
    src->replica[k] = r;
  } else if (memcmp(&src->replica[k]->fct, &src->fct, sizeof(eu_facet))) {
    // the facet's geometry changed since (eu_hip_source_update_facet): same container, new mount
    eu_source *r = src->replica[k];
    const float *base = r->sd.base;
    r->fct = src->fct; r->sd = src->sd; r->sd.base = base;
  }
  *out = src->replica[k];
  return EU_OK;
}

// contiguous strips of equal estimated cost: the segments the layout probe marks (source rows running across
// target rows: the polar faces of a cubemap made from a lat/lon image) cost ~1.6x the others with every kernel
void cost_strips(const eu_target *trg, eu_source *const *srcs, int nsrc, int n, int *begin, int *end)
{
  const int H = frame_h(trg);
  std::vector<unsigned char> flags((size_t)(H / EU_SEG_ROWS + 2), 0);
  int seg = EU_SEG_ROWS, nseg = 0;
  if (nsrc == 1) {
    nseg = eu_hip_layout_segments(trg, srcs, 1, flags.data(), (int)flags.size(), &seg);
    if (nseg < 0) nseg = 0;
  }
  auto cost_upto = [&](int y) {          // cost of rows [0, y)
    double c = 0.0;
    for (int k = 0; k * seg < y; k++) {
      const int a = k * seg, b = std::min(y, a + seg);
      c += (b - a) * ((k < nseg && flags[(size_t)k]) ? 1.6 : 1.0);
    }
    return c;
  };
  const double total = cost_upto(H);
  int prev = 0;
  for (int k = 0; k < n; k++) {
    int y = H;
    if (k + 1 < n) {
      const double want = total * (k + 1) / n;
      int lo = prev, hi = H;
      while (lo < hi) { const int mid = (lo + hi) / 2; if (cost_upto(mid) < want) lo = mid + 1; else hi = mid; }
      y = std::min(H, (lo + 15) / 16 * 16);          // whole 16-row tile rows (the staged kernel's pairs)
    }
    begin[k] = prev; end[k] = std::max(prev, y);
    prev = end[k];
  }
}

}  // namespace

int eu_hip_device_strips(const eu_target *trg, eu_source *const *srcs, int nsrc, int *begin, int *end)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if (!trg || !srcs || nsrc < 1 || !begin || !end) return fail(EU_ERR_ARGUMENT, "null argument");
  if ((rc = check_target(trg))) return rc;
  cost_strips(trg, srcs, nsrc, nslots_, begin, end);
  return EU_OK;
}

int eu_hip_render_devices(const eu_target *trg, eu_source *const *srcs, int nsrc, float *out,
                          size_t out_row_stride_bytes, int out_on_device)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if (nslots_ <= 1) return eu_hip_render(trg, srcs, nsrc, out, out_row_stride_bytes, out_on_device, nullptr);
  if (!trg || !srcs || nsrc < 1 || !out) return fail(EU_ERR_ARGUMENT, "no source / no output");
  if ((rc = check_target(trg))) return rc;
  if (trg->band_count > 1 || trg->row_begin != 0 || trg->row_end != frame_h(trg))
    return fail(EU_ERR_ARGUMENT, "eu_hip_render_devices renders whole frames (it does the tiling itself)");
  if (nsrc > 64) return fail(EU_ERR_UNSUPPORTED, "more than 64 facets over several devices");
  const int och = trg->out_format == EU_OUT_SRGBA8 ? 1 : trg->stage ? 3 : trg->nchannels;
  const size_t min_stride = (size_t)frame_w(trg) * och * sizeof(float);
  if (out_row_stride_bytes < min_stride || out_row_stride_bytes % sizeof(float))
    return fail(EU_ERR_ARGUMENT, "row stride smaller than a row / not a multiple of 4 bytes");
  int begin[EU_MAX_SLOTS], end[EU_MAX_SLOTS];
  cost_strips(trg, srcs, nsrc, nslots_, begin, end);
  // every slot: its replicas, its strip into its own buffer, the strip's way to `out` behind it on the
  // slot's stream; the slots run concurrently, one host thread feeds them
  struct guard { ~guard() { for (int k = 0; k < nslots_; k++) { cur_slot_ = k; if (ctx_[k].device >= 0 && hipSetDevice(ctx_[k].device) == hipSuccess && ctx_[k].stream) (void)hipStreamSynchronize(ctx_[k].stream); } cur_slot_ = 0; if (ctx_[0].device >= 0) (void)hipSetDevice(ctx_[0].device); } } sync_all_on_exit;
  for (int k = 0; k < nslots_; k++) {
    if (end[k] <= begin[k]) continue;
    eu_source *reps[64];
    for (int f = 0; f < nsrc; f++) {
      if (!srcs[f]) return fail(EU_ERR_HANDLE, "null source");
      if ((rc = replica_of(srcs[f], k, &reps[f]))) return rc;
    }
    if ((rc = set_slot(k))) return rc;
    eu_target t = *trg;
    t.row_begin = begin[k]; t.row_end = end[k];
    const size_t rows = (size_t)(end[k] - begin[k]);
    const bool direct = out_on_device && ctx_[k].device == ctx_[0].device && out_row_stride_bytes == min_stride;
    float *dst = nullptr;
    if (direct) dst = out + (size_t)begin[k] * (min_stride / sizeof(float));
    else {
      if ((rc = grow(&g.strip, &g.strip_cap, rows * (min_stride / sizeof(float))))) return rc;
      dst = g.strip;
    }
    if ((rc = render_on_device(&t, reps, nsrc, dst, min_stride, g.stream))) return rc;
    if (!direct) {
      char *o = (char *)out + (size_t)begin[k] * out_row_stride_bytes;
      if (!out_on_device)
        HIPCHK(hipMemcpy2DAsync(o, out_row_stride_bytes, dst, min_stride, min_stride, rows, hipMemcpyDeviceToHost, g.stream));
      else if (ctx_[k].device == ctx_[0].device)
        HIPCHK(hipMemcpy2DAsync(o, out_row_stride_bytes, dst, min_stride, min_stride, rows, hipMemcpyDeviceToDevice, g.stream));
      else if (out_row_stride_bytes == min_stride)
        HIPCHK(hipMemcpyPeerAsync(o, ctx_[0].device, dst, ctx_[k].device, rows * min_stride, g.stream));
      else
        for (size_t r = 0; r < rows; r++)
          HIPCHK(hipMemcpyPeerAsync(o + r * out_row_stride_bytes, ctx_[0].device, (char *)dst + r * min_stride, ctx_[k].device, min_stride, g.stream));
    }
  }
  return EU_OK;      // the guard waits for every slot
}

// the layout choice per segment of the (cropped) frame for this job: flags[k] = 1 where
// rows [k * seg_rows, (k + 1) * seg_rows) render faster - and cost about 1.55x the others -
// with the tile layout; returns the number of segments (0: no lat/lon source / no table)
int eu_hip_layout_segments(const eu_target *trg, eu_source *const *srcs, int nsrc,
                           unsigned char *flags, int max_flags, int *seg_rows)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if (!trg || !srcs || nsrc != 1 || !srcs[0] || !flags) return fail(EU_ERR_ARGUMENT, "one source, flags buffer");
  eu_render_params p;
  eu_target t = *trg;
  t.band_rows = 0; t.band_count = 0; t.band_index = 0;
  t.row_begin = 0; t.row_end = frame_h(trg);
  float dummy;
  if ((rc = build_params(&t, srcs, 1, &dummy, (size_t)frame_w(trg) * t.nchannels * sizeof(float), &p))) return rc;
  if (seg_rows) *seg_rows = EU_SEG_ROWS;
  if (p.twine || p.norm_mode != EU_NORM_NONE || p.src.prj != EU_SPHERICAL) return 0;
  eu_src_dev cmp = p.src;
  cmp.base = nullptr;
  if (!g.seg_valid || memcmp(&cmp, &g.seg_sd, sizeof cmp)) {
    compute_seg_flags(&p);
    g.seg_sd = cmp;
    g.seg_valid = true;
  }
  const int n = (int)g.seg_flags.size();
  if (n > max_flags) return fail(EU_ERR_ARGUMENT, "flags buffer too small");
  memcpy(flags, g.seg_flags.data(), (size_t)n);
  return n;
}

// render kernel launches of this process so far (a render step of a big cubic job is
// several: launch-level layout choice); lets a benchmark report launches per step
unsigned long long eu_hip_launch_count(void) { return g.launches; }

int eu_hip_band_rows(int height, int band_rows, int band_count, int band_index)
{
  if (height < 0 || (band_count > 1 && (band_rows < 1 || band_index < 0 || band_index >= band_count))) return 0;
  return local_rows(height, band_rows, band_count, band_index);
}

int eu_hip_sync(void)
{
  if (g.device < 0) return EU_OK;
  HIPCHK(hipStreamSynchronize(g.stream));
  if (g.last_user) HIPCHK(hipStreamSynchronize(g.last_user));
  return EU_OK;
}

int eu_hip_render_timed(const eu_target *trg, eu_source *const *srcs, int nsrc, float *out_dev,
                        size_t out_row_stride_bytes, int iters, float *mean_ms)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if (iters <= 0 || !mean_ms) return fail(EU_ERR_ARGUMENT, "bad iteration count");
  if (!srcs || nsrc < 1 || !trg || !out_dev) return fail(EU_ERR_ARGUMENT, "no source / no output");
  if ((rc = check_target(trg))) return rc;
  // one untimed launch builds the plan (stepper tables, derived copies)
  if ((rc = render_on_device(trg, srcs, nsrc, out_dev, out_row_stride_bytes, g.stream))) return rc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct guard {
    hipEvent_t &a, &b;
    ~guard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } events { e0, e1 };
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, g.stream));
  for (int i = 0; i < iters; i++)
    if ((rc = render_on_device(trg, srcs, nsrc, out_dev, out_row_stride_bytes, g.stream))) return rc;
  HIPCHK(hipEventRecord(e1, g.stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.0f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *mean_ms = ms / iters;
  return EU_OK;
}

// DIAGNOSTIC (not declared in eu_hip.h): phase stamps of the headline path,
// 8 x uint64 per wave: t0..t5, XCC id, tile index
int eu_hip_diag_stamps(const eu_target *trg, eu_source *const *srcs, int nsrc, float *out_dev,
                       size_t out_row_stride_bytes, unsigned long long *host_stamps,
                       size_t nwaves)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  eu_render_params p;
  if ((rc = build_params(trg, srcs, nsrc, out_dev, out_row_stride_bytes, &p))) return rc;
  if (p.nch != 3 || p.src.degree != 3 || p.twine) return fail(EU_ERR_ARGUMENT, "diag: NCH 3, degree 3, no twining");
  unsigned long long *d = nullptr;
  struct guard { unsigned long long *&q; ~guard() { if (q) (void)hipFree(q); } } buf { d };
  HIPCHK(hipMalloc((void **)&d, nwaves * 8 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(d, 0, nwaves * 8 * sizeof(unsigned long long), g.stream));
  for (int i = 0; i < 3; i++)
    if (eu_launch_diag(&p, d, g.stream)) return fail(EU_ERR_NO_DEVICE, "diag launch failed");
  HIPCHK(hipStreamSynchronize(g.stream));
  HIPCHK(hipMemcpy(host_stamps, d, nwaves * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return EU_OK;
}

// DIAGNOSTIC (not in eu_hip.h): the ray -> source coordinate stage of the kernels on
// caller-supplied rays (n x 3 floats, host): variant 0 eu_source_coordinate, 1 eu_coord2,
// 2 eu_coord2_ok (eu_diag.hip). out = n x 3 floats: x, y, cube face | 0; a miss is 0, 0, -1
int eu_hip_diag_source_coordinates(const eu_source *src, const float *rays, long n, int variant, float *out)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  if (!src || !rays || !out || n < 0) return fail(EU_ERR_ARGUMENT, "null argument");
  if (n == 0) return EU_OK;
  float *d = nullptr;
  struct guard { float *&q; ~guard() { if (q) (void)hipFree(q); } } buf { d };
  HIPCHK(hipMalloc((void **)&d, (size_t)n * 6 * sizeof(float)));
  HIPCHK(hipMemcpy(d, rays, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  const int lrc = eu_launch_diag_coords(&src->sd, d, n, variant, d + (size_t)n * 3, g.stream);
  if (lrc > 0) return fail(EU_ERR_UNSUPPORTED, "the packed forms cover lat/lon, cubemap and biatan6 sources");
  if (lrc < 0) return fail(EU_ERR_NO_DEVICE, "diag launch failed");
  HIPCHK(hipStreamSynchronize(g.stream));
  HIPCHK(hipMemcpy(out, d + (size_t)n * 3, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return EU_OK;
}

// DIAGNOSTIC (not in eu_hip.h): mismatch counts {div, sqrt, atan2, const div}
int eu_hip_selftest_math(unsigned long long seed, int blocks, int iters, unsigned long long *bad4)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  unsigned long long *d = nullptr;
  struct guard { unsigned long long *&q; ~guard() { if (q) (void)hipFree(q); } } buf { d };
  HIPCHK(hipMalloc((void **)&d, 4 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), g.stream));
  if (eu_launch_selftest(seed, blocks, iters, d, g.stream)) return fail(EU_ERR_NO_DEVICE, "selftest launch failed");
  HIPCHK(hipStreamSynchronize(g.stream));
  HIPCHK(hipMemcpy(bad4, d, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return EU_OK;
}

int eu_hip_malloc(void **p, size_t bytes)
{
  int rc;
  if ((rc = ensure_init())) return rc;
  HIPCHK(hipMalloc(p, bytes));
  return EU_OK;
}
int eu_hip_free(void *p) { if (p) HIPCHK(hipFree(p)); return EU_OK; }
int eu_hip_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return EU_OK;
}
int eu_hip_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return EU_OK;
}

}  // extern "C"
