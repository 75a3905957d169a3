// Structures shared by the host set-up code and the HIP kernels.
#ifndef EU_DEVICE_H
#define EU_DEVICE_H

#include <stdint.h>
#include "../../include/eu_hip.h"

#define EU_LANES 16          // zimt vector width of the pinned reference build
#define EU_SEGMENT 512       // WIELDING_SEGMENT_SIZE, zimt/bill.h:67-69

// ray = f(column table, row table): the steppers' per-segment invariants
// (stepper.h) hoisted into tables that the host fills once per target.
enum { EU_FORM_BCA = 0,      // (B*c0 + C*c1) + A   spherical, cylindrical
       EU_FORM_BA = 1,       //  B*c0 + A           rectilinear, cubemap, biatan6
       EU_FORM_FISH = 2,     //  per-pixel polar form of the fisheye stepper
       EU_FORM_STER = 3,     //  the same with the stereographic latitude
       EU_FORM_GENERIC = 4 };//  generic_stepper over tf_ex_facet (a facet with translation): eu_generic
                             //  (c0 = planar x, row: xx, yy, zz, planar y)
enum { EU_NORM_NONE = 0, EU_NORM_DIV = 1, EU_NORM_CYL = 2 };

// floats per row-table entry: {A, B, C, planar y, pad, pad} for the unbiased
// and the y-biased stepper
#define EU_ROW_VARIANT 12
#define EU_ROW_FLOATS 24

// evaluator + mount parameters of one source, device side
struct eu_src_dev {
  const float *base;         // core origin inside the braced container
  long long es0, es1;        // strides in float elements (eval.h:1865)
  int prj, nch, degree;
  int gate0, gate1;          // 0 clamp, 1 mirror, 2 periodic (eval.h:2039-2164)
  float lower0, upper0, lower1, upper1;
  // mount_t / source_t (environment.h:970-1006, :1117-1149)
  double tex_x0, tex_y0;     // total_extent.x0 / .y0 stay double (A.0)
  float ext_w, ext_h;        // float(x1 - x0), float(y1 - y0)
  float rcp_ext_w, rcp_ext_h; // RN(1/ext_w), RN(1/ext_h)
  int always_hit;            // the mask of mount_t::get_coordinate is true for every finite ray
  int cdiv_ok;               // x/ext_* == the 3-op constant division for every x (verified on the device)
  float total_w, total_h;    // float(total_width), float(total_height)
  float win_x_off, win_y_off;
  float wex0, wex1, wex2, wex3;  // window extent narrowed for the compares
  float brighten;
  int mask_paint;            // --mask_for: 0 pixels, 1 painted black, 2 painted white (masking.h:70-135)
  // pto_planar (environment.h:240-340), flags as process_geometry sets them
  int has_lcp, has_shift, has_shear;
  float lens_a, lens_b, lens_c, lens_d, lens_s, lens_h, lens_v;
  double shear_g, shear_t;
  float recip_step;          // float(1.0 / facet.step): z-score weight of the synopsis
  int mask_all;              // get_mask is constant true (cubemaps, fisheye with hfov >= 2 pi)
  // conservative early miss for the multi-facet mask pass: a ray with
  // rz < rej_cos * |ray| cannot land in the facet's window (fisheye: the angle
  // from the facet's axis maps, through the lens polynomial, to a radius whose
  // larger component already exceeds the window by 0.1 %; rectilinear: rays from
  // behind). -2: no such bound. Only whole wavefronts skip the exact test.
  float rej_cos;
  // cubemap_view_t (environment.h:1425-1460)
  float refc_md, model_to_px;
  int section_px;
  float wm[(EU_MAX_DEGREE + 1) * (EU_MAX_DEGREE + 1)];  // weight matrix [c][row]
};

// frame row of local row yl when the frame is dealt out in bands of
// (1 << shift) rows, band b going to part b % count
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int eu_frame_row(int yl, int shift, int count, int index)
{
  if (count <= 1) return yl;
  return ((((yl >> shift) * count) + index) << shift) | (yl & ((1 << shift) - 1));
}

// generic_stepper (stepper.h:353-490) over tf_ex_facet (envutil_payload.cc:1841-1885) for a facet with
// PTO translation parameters: planar -> ray by the TARGET's projection (roll_out_23, geometry.h:
// 1800-1837) -> tf3d_t (geometry.h:1850-1941). Matrices are r3_t<float>, m[3 * i + c] = r[i][c].
struct eu_tf3d {             // tf3d_t, geometry.h:1850-1941
  int has_shift;
  float dcp;
  float shift[3];
  float trg_to_md[9], md_to_src[9], trg_to_src[9];
};
struct eu_generic {
  int on, prj, ntf;          // ntf 2: tf3d1 + tf3d2 (target AND source translated, --single only)
  eu_tf3d tf[2];
};
// tf22 of tf_ex_facet: pto_planar<T, L, true> of the facet a --single job recreates (environment.h:
// 285-307) - inverse shear, inverse shift, inverse lens polynomial (inverse_lcp, lens_correction.h:
// 236-301: a cubic b-spline model, nk knots, braced + prefiltered on the host, core at coef)
struct eu_inv_planar {
  int shear, shift, lcp, nk;
  double shear_g, shear_t, s, rr_max;
  float h, v;
  const float *coef;         // device memory; coef[-1 .. nk + 1] are readable
  float m[16];               // weight matrix of degree 3
};

struct eu_render_params {
  int width, height, row_begin, row_end;
  int form, norm_mode, twine, ntaps, stage, nch;   // nch: channels of the source
  int nch_out;               // channels of the target (repix_t when they differ)
  const float *col;          // [6][width]: c0, c1, c0 (x-biased), c1 (x-biased), planar x, planar x (x-biased)
  const float *row;          // [height][EU_ROW_FLOATS]
  const float *taps;         // [ntaps][3], x and y already scaled by 4
  float *out;
  long long out_stride;      // floats per output row
  int tiles_x, tiles_y;      // grid of 64x4 tiles
  int unit_rows;             // tile rows per XCD unit (eu_render2.hip)
  int direct;                // 1: never stage through LDS (A/B switch, EU_HIP_DIRECT=1)
  // interleaved row bands (multi-GPU tiling, eu_target.band_*): local row yl of this
  // call is frame row eu_frame_row(yl, ...); band_count <= 1: identity
  int band_shift, band_count, band_index;
  int *wl;                   // eu_render4.hip: chunk counters of the persistent kernel and the lists of
                             // the tiles left to the direct-gather kernel (layout: EU4_WL_*)
  int layout;                // packed kernel: 0 by environment (default row strips), 1 row strips,
                             // 2 32x16 tiles with direct gathers (eu_render2.hip)
  int tab_finite;            // every entry of the stepper tables is finite (checked on the host when the
                             // plan is built): rays are then finite or +-inf, never NaN (eu_render5's range test)
  eu_src_dev src;
  eu_generic gen;            // form == EU_FORM_GENERIC
  eu_inv_planar inv;         // ... of a --single job (all zero otherwise)
};

#endif
