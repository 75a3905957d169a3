/* TEST INFRASTRUCTURE - not product code.
 *
 * CPU restatement, in plain scalar C, of envutil's per-output-pixel
 * reprojection path and of the set-up stages that fix the coefficients it
 * reads. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; nothing under envutil_amd/ links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - zimt stages (container layout, bracing, recursive prefilter, gates,
 *     split, weights, weighted sum, strip-mining) are checked bit-for-bit
 *     against the reference's own zimt headers compiled in place
 *     (oracle/_ref/libref_zimt.so, tests/test_oracle_vs_ref.py) and against
 *     committed fixtures generated from them (tests/golden/).
 *   - envutil stages (steppers, geometry.h projections, mounts, cubemap
 *     pickup, IR build, twining, rotation set-up) are restated from the
 *     source text: those headers need OpenImageIO/Imath, which the image
 *     lacks, so the reference cannot be built for them here. They are pinned
 *     only by the properties the reference's geometry.cc asserts
 *     (round trips, stepper == unrotated stepper + rotation): PARITY UNPINNED
 *     for their float32 results.
 *
 * Arithmetic model: zimt "goading" back-end, LANES = 16, segment 512, no FP
 * contraction, glibc libm.
 */
#ifndef EU_ORACLE_H
#define EU_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* projection_t, envutil_basic.h:68-78 */
enum { EUO_SPHERICAL, EUO_CYLINDRICAL, EUO_RECTILINEAR, EUO_STEREOGRAPHIC,
       EUO_FISHEYE, EUO_CUBEMAP, EUO_BIATAN6 };
/* zimt bc_code, zimt/common.h:82-91 */
enum { EUO_MIRROR, EUO_PERIODIC, EUO_REFLECT, EUO_NATURAL, EUO_CONSTANT,
       EUO_ZEROPAD, EUO_GUESS };

#define EUO_MAX_DEGREE 9
#define EUO_LANES 16
#define EUO_SEGMENT 512

/* a b-spline: braced coefficient container + what the evaluator needs */
typedef struct {
  float *data;          /* container origin, interleaved channels          */
  long shape[2];        /* container shape, pixels                         */
  long stride[2];       /* container strides, pixels (x stride is 1)       */
  long left[2], right[2]; /* frame widths                                  */
  long core[2];         /* core shape                                      */
  int bc[2];
  int degree;           /* spline_degree: frame + evaluator                */
  int nch;
} euo_spline;

/* one source image ("facet"), envutil_basic.h:432-520, after set-up */
typedef struct {
  int projection;
  double hfov;                 /* radians */
  int width, height;           /* total size; cubemap: face width */
  int window_width, window_height, window_x_offset, window_y_offset;
  double yaw, pitch, roll;     /* radians */
  double brighten;
  double step;
  int has_lcp;                 /* lens polynomial + shift + shear present */
  double a, b, c, h, v, s, shear_g, shear_t;
  /* PTO translation (facet_base, envutil_basic.h:446-447): position of the virtual camera in model
   * space units, orientation of the translation plane (radians) */
  double tr_x, tr_y, tr_z, tp_y, tp_p, tp_r;
  euo_spline spl;
  /* cubemap sources: cubemap_view_t members, environment.h:1425-1436 */
  float refc_md, model_to_px;
  int section_px;
  /* --mask_for (envutil_main.cc:1077-1092): 0 ordinary pixels, 1 the facet is painted black
   * (fct.masked == 0), 2 white (fct.masked == 1): masking_t / alpha_masking_t, masking.h:70-135 */
  int mask_paint;
} euo_source;

/* the target + job parameters travelling in envutil's global 'args' */
typedef struct {
  int projection;
  int width, height;
  double x0, x1, y0, y1;       /* extent */
  double yaw, pitch, roll;     /* radians */
  int nch;
  int ntaps;                   /* 0: ninputs 3; >0: ninputs 9 (twining) */
  const float *taps;           /* ntaps x {x, y, weight}, make_spread output */
  int row_begin, row_end;      /* rows [row_begin,row_end) are rendered    */
  int stage;                   /* 0 pixels, 1 rays, 2 source coordinate    */
  int nthreads;
  /* store_cropped (envutil_payload.cc:440-474): the output is crop_w x crop_h
   * and the discrete coordinates fed to the stepper are raised by
   * (crop_x0, crop_y0); crop_w == 0: no cropping. rows and row_begin/row_end
   * count in the cropped frame. */
  int crop_x0, crop_y0, crop_w, crop_h;
  /* 1: 'act + to_screen_t' (envutil_payload.cc:251-413, :524-530): one packed
   * sRGBA8 uint32 per pixel; out is then a uint32 buffer, stride in words */
  int screen;
  /* args.synopsis for several facets: 0 "panorama" (voronoi_syn / voronoi_syn_plus by channel
   * count), 1 "hdr_merge" (_hdr_merge_syn, envutil_payload.cc:1325-1626) */
  int synopsis;
  /* args.single (envutil_main.cc:1161-1180, envutil_payload.cc:2058-2069): the target recreates this
   * facet - the job's projection, size, extent and orientation are the facet's, and when it has lens
   * parameters (a, b, c, h, v, shear) or translation every facet is stepped by generic_stepper over
   * tf_ex_facet with the INVERSE planar transformation and the inverse translation. NULL: none. */
  const euo_source *single;
} euo_job;

/* to_screen_t's LUT (256 knots of 255 * sRGB(i / 255), float) and one pixel */
void   euo_screen_lut(float *lut257);
float  euo_lut_eval(const float *lut257, float in);   /* lut_based_tf::eval, one lane */
unsigned euo_to_screen(const float *lut257, int nch, const float *px);

/* set-up arithmetic */
double euo_get_vfov(int projection, int width, int height, double hfov);
double euo_get_step(int projection, int width, int height, double hfov);
void   euo_get_extent(int projection, int width, int height, double hfov,
                      double *ext4);
void   euo_make_r3(double roll, double pitch, double yaw, int inverse,
                   double *m9);
void   euo_rotate_r3(const double *lhs9, const double *rhs9, double *out9);
int    euo_make_spread(int w, int h, float d, float sigma, float threshold,
                       float *taps_out, int max_taps);
void   euo_weight_matrix(int degree, float *m);  /* m[c*(degree+1)+row] */
int    euo_poles(int degree, long double *poles);
void   euo_basis_weights(int degree, float delta, float *w);

/* b-spline container, bracing, prefilter */
void   euo_spline_geometry(int degree, int bc0, int bc1, long w, long h,
                           long *shape2_left2_right2);
int    euo_spline_init(euo_spline *s, float *container, long w, long h,
                       int nch, int degree, int bc0, int bc1);
void   euo_spline_set_core(euo_spline *s, const float *core);
void   euo_brace(euo_spline *s, int axis);      /* -1: all axes */
void   euo_prefilter(euo_spline *s, int prefilter_degree);
void   euo_spherical_prefilter(euo_spline *s, int prefilter_degree);
void   euo_filter_lines(float *base, long n_lines, long line_stride,
                        long len, long ele_stride, int bc, int degree,
                        double tolerance);

/* cubemap source set-up: metrics_t + IR image */
typedef struct {
  long face_px, section_px, left_frame_px, right_frame_px, n_tiles;
  long inherent_support_px;
  double model_to_px, px_to_model, section_md, refc_md, radius_md;
  int discrete90;
} euo_metrics;
void   euo_metrics_init(euo_metrics *m, long face_px, double face_fov,
                        long support_min_px, long tile_px);
/* faces: 6 stacked face_px x face_px images; ir: section_px x 6*section_px */
void   euo_cubemap_build(const euo_metrics *m, const float *faces, int nch,
                         int spline_degree, int prefilter_degree, float *ir);

/* evaluation */
void   euo_eval(const euo_spline *s, const float *crd2, long n, float *out);
void   euo_eval_shifted(const euo_spline *s, int degree, const float *crd2,
                        long n, float *out);

/* the hot path: zimt::process(shape, stepper, environment|twine_t, storer) */
int    euo_render(const euo_job *job, const euo_source *src, int nsrc,
                  float *out, long out_row_stride /* floats */);

/* geometry functors, double precision, for the reference's own property
 * tests (geometry.cc:283-420) */
void   euo_lens_factor(double a, double b, double c, const float *x, long n, float *out);
/* inverse_lcp<float, LANES>(a, b, c, r_max, sz).eval (lens_correction.h:236-301); returns the number of
 * knots of the model (0: too many), knots = its prefiltered core */
int    euo_inverse_lcp(double a, double b, double c, double r_max, int sz, const float *x, long n,
                       float *out, float *knots, int max_knots);
/* a masked / cropped facet's alpha plane (environment.h:727-843): 1 everywhere, PTO exclude
 * polygons (fill_polygon, envutil_basic.cc:236-320) and the outside of the lens crop cleared,
 * then zimt::convolve with the binomial 1 4 6 4 1 / 16, REFLECT, along both axes.
 * polygons: counts[npolys] vertices each, coordinates concatenated in xs / ys.
 * crop_kind 0 none, 1 rectangular, 2 elliptic. stage 0: before the convolution, 1: after. */
void   euo_facet_alpha(float *alpha, int w, int h, int npolys, const int *counts, const float *xs,
                       const float *ys, int crop_kind, int cx0, int cx1, int cy0, int cy1, int stage);
/* the binomial alone, in place (pinned against zimt::convolve through oracle/_ref) */
void   euo_binomial_plane(float *plane, int w, int h);
void   euo_mask_paint(int nch, int mask_paint, float *px);   /* masking_t / alpha_masking_t, masking.h:70-135 */
void   euo_source_coordinates(const euo_source *src, const float *rays, long n, float *out3);
void   euo_prj_to_ray_d(int projection, const double *in2, double *out3);
void   euo_ray_to_prj_d(int projection, const double *in3, double *out2);

#ifdef __cplusplus
}
#endif
#endif
