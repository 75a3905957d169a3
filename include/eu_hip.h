/* eu_hip.h - C ABI of the MI355X-native envutil reprojection path.
 *
 * This is the drop-in boundary for ONE path of kfjahnke/envutil: the
 * per-output-pixel reprojection that the reference runs as
 *   zimt::process(trg.shape, get, act, cstor, bill)      envutil_payload.cc:541
 * behind
 *   dispatch_base::payload(nchannels, ninputs, projection) envutil_dispatch.h:49-65
 * Everything here is plain C: pointers, sizes, PODs. No torch types, no C++.
 * A reference maintainer binds it exactly where the per-ISA `dispatch` structs
 * are built today (envutil_payload.cc:2390-2442); INTEGRATION.md shows the stub.
 *
 * Conventions: axes RIGHT, DOWN, FORWARD; cube faces
 * LEFT0 RIGHT1 TOP2 BOTTOM3 FRONT4 BACK5 stacked vertically; angles in radians;
 * pixels are interleaved float32 channels, x fastest, rows contiguous.
 *
 * All functions return 0 on success or a negative eu_status code; none of them
 * aborts. eu_hip_last_error() gives the text for the calling thread.
 *
 * Threading: like the reference's payload (called on the main thread, global `args`), the
 * library is NOT re-entrant: one host thread at a time, one device per process (the first
 * eu_hip_init or the first call fixes it). A caller-supplied stream orders the KERNELS of a
 * render; the stepper tables of a job are uploaded synchronously before its launch, and the
 * library waits for the previous job's stream before it rewrites them, so jobs on different
 * streams never race on the library's own buffers. eu_hip_sync() waits for the library's
 * stream and for the stream of the last render.
 */
#ifndef EU_HIP_H
#define EU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors projection_t, envutil_basic.h:68-78 */
typedef enum {
  EU_SPHERICAL = 0, EU_CYLINDRICAL, EU_RECTILINEAR, EU_STEREOGRAPHIC,
  EU_FISHEYE, EU_CUBEMAP, EU_BIATAN6, EU_PRJ_NONE
} eu_projection;

/* mirrors zimt::bc_code, zimt/common.h:82-91 */
typedef enum {
  EU_BC_MIRROR = 0, EU_BC_PERIODIC, EU_BC_REFLECT, EU_BC_NATURAL,
  EU_BC_CONSTANT, EU_BC_ZEROPAD, EU_BC_GUESS
} eu_bc;

typedef enum {
  EU_OK = 0,
  EU_ERR_NO_DEVICE = -1,     /* no HIP device / HIP runtime error            */
  EU_ERR_ARGUMENT = -2,      /* inconsistent or out-of-range argument        */
  EU_ERR_UNSUPPORTED = -3,   /* valid for the reference, not built yet here  */
  EU_ERR_MEMORY = -4,
  EU_ERR_HANDLE = -5
} eu_status;

#define EU_MAX_DEGREE 9
#define EU_MAX_TAPS 1024
#define EU_MAX_FACETS 64

/* One source image. Mirrors the fields of facet_spec / facet_base that the
 * render path reads (envutil_basic.h:432-520). `step`, `brighten` as there. */
typedef struct eu_facet {
  int32_t projection;            /* eu_projection                              */
  int32_t nchannels;             /* 1..4                                       */
  double  hfov;                  /* radians                                    */
  int32_t width, height;         /* total size; cubemaps: face width, height=6w*/
  int32_t window_width, window_height, window_x_offset, window_y_offset;
  double  yaw, pitch, roll;      /* radians                                    */
  double  brighten;              /* facet_spec::brighten                       */
  double  step;                  /* facet_base::step (get_step)                */
  int32_t has_lcp;               /* lens polynomial / shift / shear present    */
  double  a, b, c, h, v, s, shear_g, shear_t;
  /* PTO translation (facet_base::tr_*, tp_*, envutil_basic.h:446-447): position of the virtual camera
   * in model space units and the orientation of the translation plane (radians). A facet with
   * tr_x/y/z != 0 is stepped by generic_stepper over tf_ex_facet (envutil_payload.cc:2095-2110,
   * :2145-2158; geometry.h:1850-1941) instead of the target projection's own stepper.            */
  double  tr_x, tr_y, tr_z, tp_y, tp_p, tp_r;
  /* --mask_for (facet_spec::masked, envutil_main.cc:1077-1092; masking.h:70-135): 0 ordinary
   * pixels (masked == -1), 1 the facet is painted black (masked == 0), 2 white (masked == 1).
   * Facets of 1 / 3 channels yield the paint, facets with alpha paint * alpha and their alpha;
   * a target with another channel count takes them through mono_t (1 or 2 channels only).   */
  int32_t mask_paint;
} eu_facet;

typedef struct eu_source eu_source;   /* opaque: coefficients resident in HBM */

/* Geometry of a braced b-spline container as zimt::bspline lays it out
 * (zimt/bspline.h:305-450, :759-820); filled by eu_hip_container_geometry. */
typedef struct eu_container {
  int64_t shape[2];              /* container shape, pixels                    */
  int64_t left[2], right[2];     /* frame widths                               */
  int64_t core[2];               /* core shape                                 */
} eu_container;

/* The target image and job parameters that travel in envutil's global `args`
 * (envutil_basic.h:633-705): extent as computed by get_extent
 * (envutil_basic.cc:156-229), camera orientation, twining tap table as
 * produced by make_spread (envutil_main.cc:1253-1355; ntaps = 0: ninputs 3). */
typedef struct eu_target {
  int32_t projection;
  int32_t width, height;
  double  x0, x1, y0, y1;
  double  yaw, pitch, roll;
  int32_t nchannels;
  int32_t ntaps;
  const float *taps;             /* host pointer, ntaps x {x, y, weight}       */
  int32_t row_begin, row_end;    /* rows [row_begin,row_end): multi-GPU tiling */
  int32_t stage;                 /* 0 pixels; 1 rays; 2 source coordinates
                                    (stages 1/2 write 3 floats per pixel; used
                                    by the stage-wise parity tests)            */
  /* args.store_cropped + p_crop_* (envutil_basic.h:684-687; applied at
   * envutil_payload.cc:440-474): the job renders crop_w x crop_h pixels whose
   * discrete coordinates start at (crop_x0, crop_y0) of the width x height
   * frame (zimt's bill.get_offset). crop_w == 0: the whole frame. Rows,
   * row_begin/row_end and the output buffer are those of the cropped frame.  */
  int32_t crop_x0, crop_y0, crop_w, crop_h;
  /* EU_OUT_FLOAT: nchannels floats per pixel (zimt::storer).
   * EU_OUT_SRGBA8: one packed uint32 per pixel, the tethered 'act + to_screen_t'
   * pipeline writing args.p_screen_data (envutil_payload.cc:251-413, :524-530);
   * the output buffer is then uint32 and strides still count bytes.          */
  int32_t out_format;
  /* Multi-GPU tiling by INTERLEAVED row bands (load balance: rows differ in
   * cost, e.g. the polar faces of a cubemap target take 1.7x the others): the
   * frame's rows are cut into bands of band_rows rows (a power of two >= 4),
   * band b belongs to part b % band_count, and this call renders the bands of
   * part band_index, compacted in order: local row l is frame row
   * ((l / band_rows) * band_count + band_index) * band_rows + l % band_rows.
   * row_begin/row_end and the output buffer then count LOCAL rows
   * (eu_hip_band_rows tells how many there are). band_count <= 1: off.       */
  int32_t band_rows, band_count, band_index;
  /* args.synopsis (envutil_main.cc:232, dispatched at envutil_payload.cc:2302-2318): how a job
   * with several facets composes them. EU_SYN_PANORAMA: voronoi_syn (1/3 channels) or
   * voronoi_syn_plus (2/4 channels); EU_SYN_HDR_MERGE: _hdr_merge_syn (envutil_payload.cc:
   * 1325-1626), the quality-weighted sum of ALL facets. Ignored for single-facet jobs.        */
  int32_t synopsis;
  /* args.single (envutil_main.cc:1161-1180; envutil_payload.cc:2058-2069): the target RECREATES this facet.
   * Projection, size, extent and orientation above are then the facet's own ((facet_base&) args = fspec);
   * this pointer hands over what a target otherwise does not have - lens polynomial, shift, shear and
   * translation - and when any of them is set every facet of the job is stepped by generic_stepper over
   * tf_ex_facet with the inverse planar transformation (pto_planar<T, L, true>, inverse_lcp) and the
   * inverse translation. Host pointer, read during the call; NULL: an ordinary target.            */
  const eu_facet *single;
} eu_target;

/* How the library would lay the job's rows out (eu_api.hip: launch-level choice between
 * row strips and 32x16 tiles): flags[k] = 1 for segment k of *seg_rows frame rows where
 * source rows run across target rows - those rows take about 1.55x the time of the others,
 * about 2x when they are part of a strip they break into several launches. For hosts that
 * split a frame into CONTIGUOUS strips of equal cost. Returns the number of segments
 * (0 when the job has no such structure) or a negative eu_status. */
int  eu_hip_layout_segments(const eu_target *trg, eu_source *const *srcs, int nsrc,
                            unsigned char *flags, int max_flags, int *seg_rows);

/* render kernel launches issued so far by this process (single-facet path) */
unsigned long long eu_hip_launch_count(void);

/* number of local rows of part band_index (see eu_target.band_*) in a frame of
 * `height` rows; height itself when band_count <= 1 */
int  eu_hip_band_rows(int height, int band_rows, int band_count, int band_index);

enum { EU_OUT_FLOAT = 0, EU_OUT_SRGBA8 = 1 };
enum { EU_SYN_PANORAMA = 0, EU_SYN_HDR_MERGE = 1 };

/* ---- device / lifecycle ------------------------------------------------- */
int  eu_hip_device_count(void);
int  eu_hip_init(int device);                 /* selects the device for this process */
const char *eu_hip_last_error(void);

/* ---- one process, several devices (round 3; SURVEY 8e: the output rows tiled across the GPUs of a node;
 * the reference has no counterpart - its payload() runs on the host's cores) --------------------------------
 * eu_hip_init_devices makes one device SLOT per entry of `devices` (at most EU_MAX_SLOTS; the same device may
 * be listed more than once: separate streams, tables and buffers - how a one-GPU box tests this). Sources are
 * created on slot 0 as before; eu_hip_render_devices replicates them onto the other slots on first use
 * (hipMemcpyPeerAsync over xGMI: the "broadcast of the source environment"), cuts the frame into contiguous
 * strips of equal estimated cost, renders strip k on slot k - all slots concurrently - and gathers the strips
 * into `out` (a host buffer, or device memory of slot 0: peer copies). Same pixels as eu_hip_render. */
#define EU_MAX_SLOTS 16
int  eu_hip_init_devices(const int *devices, int ndevices);
int  eu_hip_device_slots(void);               /* 1 unless eu_hip_init_devices was called */
int  eu_hip_render_devices(const eu_target *trg, eu_source *const *srcs, int nsrc, float *out,
                           size_t out_row_stride_bytes, int out_on_device);
/* the rows [begin[k], end[k]) eu_hip_render_devices gives slot k for this job (nslots entries each) */
int  eu_hip_device_strips(const eu_target *trg, eu_source *const *srcs, int nsrc, int *begin, int *end);

/* ---- set-up arithmetic shared with the reference's host code ------------- */
/* get_extent / get_vfov / get_step, envutil_basic.cc:49-229 */
int  eu_hip_get_extent(int projection, int width, int height, double hfov,
                       double *x0x1y0y1);
double eu_hip_get_step(int projection, int width, int height, double hfov);
/* make_spread, envutil_main.cc:1253-1355; returns tap count or <0 */
int  eu_hip_make_spread(int w, int h, float d, float sigma, float threshold,
                        float *taps, int max_taps);
/* metrics_t, cubemap.h:233-400: section_px, left/right frame, refc_md, model_to_px */
/* PTO exclude masks and lens crops: the edit of a facet's loaded pixels that source_t's
 * constructor makes before it prefilters (environment.h:700-890; fill_polygon,
 * envutil_basic.cc:236-320; zimt::convolve with the binomial 1 4 6 4 1 / 16, REFLECT).
 * HOST function, no device needed: `pixels` (width x height x nchannels, nchannels 2 or 4,
 * alpha last) is multiplied in place, every channel, by the softened alpha plane; `alpha_out`
 * (width x height floats) receives that plane when it is not NULL.
 * crop_kind: 0 none, 1 rectangular [x0, x1) x [y0, y1), 2 elliptic (fisheye images). */
typedef struct eu_mask_polygon { int n; const float *x; const float *y; } eu_mask_polygon;
int  eu_hip_facet_alpha(float *pixels, int width, int height, int nchannels,
                        const eu_mask_polygon *polygons, int npolygons, int crop_kind,
                        int crop_x0, int crop_x1, int crop_y0, int crop_y1, float *alpha_out);
int  eu_hip_cubemap_metrics(int face_px, double face_fov, int support_min,
                            int tile_px, int64_t *section_px,
                            int64_t *left_frame_px, double *refc_md,
                            double *model_to_px);
int  eu_hip_container_geometry(int degree, int bc0, int bc1, int64_t w,
                               int64_t h, eu_container *out);

/* ---- sources: the asset_handler residency (environment.h:84-227) --------- */
/* Load pixel data (host pointer, window_width x window_height interleaved
 * floats; cubemaps: 6 stacked faces), build the braced coefficient container
 * in HBM and prefilter it ON THE DEVICE, as source_t's constructor
 * (environment.h:594-962) or cubemap_t::load (cubemap.h:1147-1233) do on the
 * CPU. */
int  eu_hip_source_load(const eu_facet *fct, const float *pixels,
                        int spline_degree, int prefilter_degree,
                        int support_min, int tile_size, eu_source **out);
/* Adopt an already prefiltered and braced container (host pointer, shape as
 * eu_hip_container_geometry reports; cubemaps: the IR image section_px x
 * 6*section_px). This is what a bound reference hands over when it keeps its
 * own set-up stage. */
int  eu_hip_source_adopt(const eu_facet *fct, const float *container,
                         int spline_degree, int bc0, int bc1,
                         int support_min, int tile_size, eu_source **out);
/* Allocate the container for a facet without filling it, and expose its device
 * address: the multi-GPU host broadcasts rank 0's prefiltered coefficients
 * straight into it over RCCL/xGMI - one broadcast per source,
 * after which the source stays resident like any other. */
int  eu_hip_source_alloc(const eu_facet *fct, int spline_degree,
                         int support_min, int tile_size, eu_source **out);
/* The asset_handler caches the b-spline only; the facet's geometry (yaw / pitch / roll,
 * hfov, window offsets, lens a/b/c/h/v, shear, brighten, step) is read fresh from facet_spec
 * on every job (environment.h:84-227). A host that keeps sources resident calls this before a
 * job whose facet_spec changed: the evaluator and mount parameters are rebuilt from `fct`, the
 * coefficients stay. The image itself (projection, size, channels) must be the resident one. */
int  eu_hip_source_update_facet(eu_source *src, const eu_facet *fct);
int  eu_hip_source_device_ptr(const eu_source *src, void **dev_ptr,
                              size_t *nfloats);
/* copy the device-resident container back (tests, checkpointing) */
int  eu_hip_source_download(const eu_source *src, float *container,
                            size_t nfloats);
int  eu_hip_source_info(const eu_source *src, eu_container *geom, int *nch);
int  eu_hip_source_release(eu_source *src);

/* ---- the hot path --------------------------------------------------------- */
/* zimt::process(shape, stepper, environment|twine_t, storer): renders rows
 * [row_begin,row_end) of the target into `out` (row stride in bytes).
 * out_on_device != 0: `out` is a device pointer, the call is asynchronous on
 * `stream` (a hipStream_t, NULL = the library's own stream) until
 * eu_hip_sync(); otherwise `out` is host memory and the call returns when the
 * rows are there. With trg->out_format == EU_OUT_SRGBA8 `out` points to uint32
 * words (one per pixel, cast the pointer); with a crop window the rows are
 * crop_w pixels wide. */
int  eu_hip_render(const eu_target *trg, eu_source *const *srcs, int nsrc,
                   float *out, size_t out_row_stride_bytes, int out_on_device,
                   void *stream);
int  eu_hip_sync(void);

/* Time the render kernel alone with HIP events on its own stream: runs the
 * launch `iters` times back to back, returns the mean kernel duration. */
int  eu_hip_render_timed(const eu_target *trg, eu_source *const *srcs, int nsrc,
                         float *out_dev, size_t out_row_stride_bytes,
                         int iters, float *mean_ms);

/* device memory helpers for hosts without a HIP binding of their own */
int  eu_hip_malloc(void **p, size_t bytes);
int  eu_hip_free(void *p);
int  eu_hip_memcpy_d2h(void *dst, const void *src, size_t bytes);
int  eu_hip_memcpy_h2d(void *dst, const void *src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
