// eu_dispatch.hpp - C++ host mirror of the envutil surface this path plugs into.
//
// Same names and meaning as the reference (all in namespace project):
//   projection_t                    envutil_basic.h:68-78
//   extent_type / facet_base / facet_spec / arguments (the fields the render
//   path reads)                     envutil_basic.h:432-705
//   arguments::twine_setup          envutil_main.cc:1405-1616 (explicit twine)
//   dispatch_base::payload(nchannels, ninputs, projection)
//                                   envutil_dispatch.h:49-65
//   get_dispatch()                  envutil_dispatch.h:70-73
// Differences, all at the I/O edge (file I/O is out of scope): a facet carries
// a pointer to its pixels instead of a file name, and the job's output goes to
// a caller-provided float buffer instead of an image file. payload() returns 0
// on success like the reference, or the negative eu_status of the C ABI
// (the reference aborts instead).
//
// Header-only; link with -leu_hip.
#ifndef EU_DISPATCH_HPP
#define EU_DISPATCH_HPP

#include <array>
#include <cmath>
#include <fstream>
#include <map>
#include <string>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "eu_hip.h"

namespace project {

typedef enum { SPHERICAL, CYLINDRICAL, RECTILINEAR, STEREOGRAPHIC, FISHEYE, CUBEMAP, BIATAN6,
               PRJ_NONE } projection_t;

struct extent_type { double x0 = 0, x1 = 0, y0 = 0, y1 = 0; };

struct facet_base : public extent_type
{
  projection_t projection = SPHERICAL;
  double hfov = 0.0;              // radians
  double step = 0.0;
  double yaw = 0.0, pitch = 0.0, roll = 0.0;   // radians
  int width = 0, height = 0;
  int window_width = 0, window_height = 0, window_x_offset = 0, window_y_offset = 0;
  double shear_g = 0.0, shear_t = 0.0;
  double s = 0.0, a = 0.0, b = 0.0, c = 0.0, d = 0.0, h = 0.0, v = 0.0;
  bool has_shift = false, has_lcp = false, has_shear = false;
  // PTO translation and reprojection plane (envutil_basic.h:441-446): parsed by the front end and handed
  // to the library (eu_facet.tr_*, tp_*), which renders such facets with the generic stepper
  double tr_x = 0.0, tr_y = 0.0, tr_z = 0.0, tp_y = 0.0, tp_p = 0.0, tp_r = 0.0;

  // facet_base::process_geometry, envutil_basic.h:499-521
  void process_geometry()
  {
    double e[4];
    eu_hip_get_extent(projection, width, height, hfov, e);
    x0 = e[0]; x1 = e[1]; y0 = e[2]; y1 = e[3];
    step = eu_hip_get_step(projection, width, height, hfov);
    has_shift = (h != 0.0 || v != 0.0);
    has_lcp = (a != 0.0 || b != 0.0 || c != 0.0);
    has_shear = (shear_g != 0.0 || shear_t != 0.0);
    double dv = std::fabs(y1 - y0) / 2.0, dh = std::fabs(x1 - x0) / 2.0;
    s = (dh < dv) ? dh : dv;
    if (window_width == 0) { window_width = width; window_height = height; }
  }
};

// one PTO k-line (envutil_basic.h:387-411): variant 0 is an exclude mask of that image
struct pto_mask_type
{
  int image = -1, variant = -1;
  std::vector<float> vx, vy;
};

struct facet_spec : public facet_base
{
  int facet_no = 0;
  int nchannels = 3;
  std::string filename, projection_str;
  std::string colour_space;       // what the file's samples are in: the PTO i-line's Csp clause, else --input_colour_space,
                                  // else (empty) what the file itself says (envutil_main.cc:640-670, envutil_basic.h:950-977)
  std::string asset_key;          // residency key (environment.h:84-227)
  float brighten = 1.0f;
  int masked = -1;                // --mask_for (envutil_main.cc:1080-1092)
  bool has_lens_crop = false, has_pto_mask = false;
  int crop_x0 = 0, crop_x1 = 0, crop_y0 = 0, crop_y1 = 0;
  std::vector<pto_mask_type> pto_mask_v;
  // pixels_prepared: the caller has applied masks and crop to `pixels` (prepare_facet_pixels,
  // eu_imageprep.hpp - what source_t's constructor does after read_image_data)
  bool pixels_prepared = false;
  const float *pixels = nullptr;  // window_width x window_height x nchannels (cubemaps: 6 faces)
};

struct arguments : public facet_base
{
  int spline_degree = 1, prefilter_degree = -1;
  int twine = 0;
  float twine_width = 1.0f, twine_sigma = 0.0f, twine_threshold = 0.0f;
  float twine_density = 1.0f;
  int twine_max = 8;
  bool twine_normalize = false, twine_precise = false;
  float brighten = 1.0f;
  std::string output, split, synopsis = "panorama", pto_file, twf_file, projection_str;
  // envutil_main.cc:400-437: the working space and the output's default to "Linear", the input's to "" (the file's own)
  std::string input_colour_space, working_colour_space = "Linear", colour_space = "Linear";
  int solo = -1, single = -1, mask_for = -1;
  std::vector<std::array<float, 3>> twine_spread;
  int support_min = 8, tile_size = 64;
  int nchannels = 3;
  int nfacets = 0;
  std::vector<facet_spec> facet_spec_v;
  float *p_output = nullptr;      // width x height x nchannels floats (crop size if store_cropped)
  // PTO p-line crop (envutil_basic.h:684-687; envutil_payload.cc:440-474)
  bool store_cropped = false;
  int p_crop_x0 = 0, p_crop_x1 = 0, p_crop_y0 = 0, p_crop_y1 = 0;
  // tethered rendering (envutil_basic.h; envutil_payload.cc:524-530): packed
  // sRGBA8 words into caller-owned memory instead of float pixels
  bool tethered = false;
  void *p_screen_data = nullptr;
  bool verbose = false;

  // target extent and step, envutil_main.cc:1203-1232
  void target_setup()
  {
    double e[4];
    eu_hip_get_extent(projection, width, height, hfov, e);
    x0 = e[0]; x1 = e[1]; y0 = e[2]; y1 = e[3];
    step = (x1 - x0) / width;
    if (prefilter_degree < 0) prefilter_degree = spline_degree;
    nfacets = int(facet_spec_v.size());
  }

  // arguments::twine_setup, envutil_main.cc:1405-1616: --twine N (N >= 0) is taken as given;
  // -1 (the command line's default) derives twine and twine_width from the magnification
  // smallest facet step / target step; then twine_density, then make_spread
  // (envutil_main.cc:1253-1355) - or, with --twf_file, the tap table read from that text file
  // (read_twf_file, envutil_main.cc:1357-1403: x y weight triples, x and y scaled by twine_width,
  // the weights divided by their sum under --twine_normalize). Returns false when the file
  // cannot be read (the reference asserts).
  bool twine_setup()
  {
    twine_spread.clear();
    if (!twf_file.empty()) twine = 1;
    if (twine != -1) {
      if (twine < 0) twine = 0;
    } else {
      double smallest_step = 1e300;
      if (nfacets == 1 || solo > 0) smallest_step = facet_spec_v[size_t(solo < 0 ? 0 : solo)].step;
      else for (const auto &f : facet_spec_v) smallest_step = f.step < smallest_step ? f.step : smallest_step;
      const double mag = smallest_step / step;
      if (mag > 1.0) {
        if (spline_degree > 1) twine = nfacets > 1 ? 3 : (mag < 2.0 ? 2 : 1);
        else {
          twine = int(1.0 + mag) < 5 ? int(1.0 + mag) : 5;
          twine_width = float(mag);
        }
      } else {
        twine = int(1.0 + 1.0 / mag);
        if (twine > twine_max) twine = twine_max;
        twine_width = 1.0f;
      }
    }
    if (twine_density != 1.0f) twine = int(std::round(twine * twine_density));
    if (!twf_file.empty()) {
      std::ifstream ifs(twf_file);
      if (!ifs.good()) return false;
      double sum = 0.0;
      std::array<float, 3> c;
      while (ifs.good()) {
        ifs >> c[0] >> c[1] >> c[2];
        if (ifs.eof()) break;
        if (ifs.fail()) return false;              // not a number: the reference would loop forever
        twine_spread.push_back(c);
        sum += c[2];
      }
      for (auto &t : twine_spread) {
        t[0] *= twine_width; t[1] *= twine_width;
        if (twine_normalize) t[2] /= sum;
      }
      return !twine || !twine_spread.empty();
    }
    if (twine <= 0) return true;
    std::vector<float> t(3 * size_t(twine < 2 ? 4 : twine * twine));
    int n = eu_hip_make_spread(twine, twine, twine_width, twine_sigma, twine_threshold, t.data(),
                               int(t.size() / 3));
    for (int i = 0; i < n; i++) twine_spread.push_back({ t[3 * i], t[3 * i + 1], t[3 * i + 2] });
    return true;
  }
};

inline arguments args;            // the reference's global (envutil_basic.h:705)

struct dispatch_base
{
  // envutil_dispatch.h:55-57; there is no highway target here: 0, named for the GPU
  std::size_t hwy_target = 0;
  std::string hwy_target_name = "gfx950", hwy_target_str = "HIP/CDNA4";
  virtual int payload(int nchannels, int ninputs, projection_t projection) const = 0;
  virtual ~dispatch_base() {}
};

struct hip_dispatch : public dispatch_base
{
  mutable std::map<std::string, eu_source *> resident;   // asset_handler

  static eu_facet to_eu(const facet_spec &f)
  {
    eu_facet e {};
    e.projection = f.projection; e.nchannels = f.nchannels; e.hfov = f.hfov;
    e.width = f.width; e.height = f.height;
    e.window_width = f.window_width; e.window_height = f.window_height;
    e.window_x_offset = f.window_x_offset; e.window_y_offset = f.window_y_offset;
    e.yaw = f.yaw; e.pitch = f.pitch; e.roll = f.roll;
    e.brighten = f.brighten; e.step = f.step; e.has_lcp = f.has_lcp;
    e.a = f.a; e.b = f.b; e.c = f.c; e.h = f.h; e.v = f.v; e.s = f.s;
    e.shear_g = f.shear_g; e.shear_t = f.shear_t;
    // PTO translation: such a facet is stepped by generic_stepper (envutil_payload.cc:2095-2110, :2145-2158)
    e.tr_x = f.tr_x; e.tr_y = f.tr_y; e.tr_z = f.tr_z; e.tp_y = f.tp_y; e.tp_p = f.tp_p; e.tp_r = f.tp_r;
    e.mask_paint = f.masked + 1;     // --mask_for: -1 ordinary, 0 painted black, 1 painted white
    return e;
  }

  int payload(int nchannels, int ninputs, projection_t projection) const override
  {
    if (projection != args.projection) return EU_ERR_ARGUMENT;
    if ((ninputs == 9) != !args.twine_spread.empty()) return EU_ERR_ARGUMENT;
    if (args.tethered ? !args.p_screen_data : !args.p_output) return EU_ERR_ARGUMENT;
    // PTO masks and lens crops edit the pixels at load time (prepare_facet_pixels) and must have
    // been applied by the caller; an unknown synopsis is the reference's assert(false)
    // (envutil_payload.cc:2316-2318)
    if (args.synopsis != "panorama" && args.synopsis != "hdr_merge") return EU_ERR_ARGUMENT;
    // --split is the caller's loop over --single jobs (core(), envutil_main.cc:1676-1722)
    if (args.single >= int(args.facet_spec_v.size()) || args.solo >= int(args.facet_spec_v.size())) return EU_ERR_ARGUMENT;
    for (const auto &fct : args.facet_spec_v)
      if ((fct.has_pto_mask || fct.has_lens_crop) && !fct.pixels_prepared && !resident.count(fct.asset_key))
        return EU_ERR_UNSUPPORTED;
    std::vector<eu_source *> srcs;
    for (size_t fi = 0; fi < args.facet_spec_v.size(); fi++) {
      // --solo: only that facet takes part (fuse(), envutil_payload.cc:2085-2127)
      if (args.solo >= 0 && int(fi) != args.solo) continue;
      const auto &fct = args.facet_spec_v[fi];
      auto it = resident.find(fct.asset_key);
      if (it == resident.end()) {
        eu_facet e = to_eu(fct);
        eu_source *s = nullptr;
        int rc = eu_hip_source_load(&e, fct.pixels, args.spline_degree, args.prefilter_degree,
                                    args.support_min, args.tile_size, &s);
        if (rc != EU_OK) return rc;
        it = resident.emplace(fct.asset_key, s).first;
      } else {
        // the asset is resident; its facet_spec may have changed since (orientation, hfov,
        // lens, brighten): the reference reads it fresh on every job
        eu_facet e = to_eu(fct);
        int rc = eu_hip_source_update_facet(it->second, &e);
        if (rc != EU_OK) return rc;
      }
      srcs.push_back(it->second);
    }
    eu_target t {};
    t.projection = projection; t.width = args.width; t.height = args.height;
    t.x0 = args.x0; t.x1 = args.x1; t.y0 = args.y0; t.y1 = args.y1;
    t.yaw = args.yaw; t.pitch = args.pitch; t.roll = args.roll;
    t.nchannels = nchannels;
    t.ntaps = ninputs == 9 ? int(args.twine_spread.size()) : 0;
    t.taps = ninputs == 9 ? args.twine_spread[0].data() : nullptr;
    int w = args.width, h = args.height;
    if (args.store_cropped) {
      w = args.p_crop_x1 - args.p_crop_x0; h = args.p_crop_y1 - args.p_crop_y0;
      t.crop_x0 = args.p_crop_x0; t.crop_y0 = args.p_crop_y0; t.crop_w = w; t.crop_h = h;
    }
    t.row_begin = 0; t.row_end = h; t.stage = 0;
    t.synopsis = args.synopsis == "hdr_merge" ? EU_SYN_HDR_MERGE : EU_SYN_PANORAMA;
    // --single: args carries the facet's geometry ((facet_base&) args = fspec); its lens and translation
    // parameters reach the library through eu_target.single
    eu_facet single_fct {};
    if (args.single >= 0) {
      single_fct = to_eu(args.facet_spec_v[size_t(args.single)]);
      t.single = &single_fct;
    }
    // a node with several GPUs: the output rows tiled over all of them behind the same call (one device slot
    // per GPU, the sources replicated by peer copies on first use; eu_hip.h: eu_hip_render_devices).
    // EU_HIP_DEVICES=n limits the slots, =1 keeps the single-device path.
    static const int slots = [] {
      int n = eu_hip_device_count();
      if (const char *e = std::getenv("EU_HIP_DEVICES")) n = std::min(n, std::max(1, std::atoi(e)));
      n = std::min(n, int(EU_MAX_SLOTS));
      if (n > 1) {
        std::vector<int> dev(size_t(n), 0);
        for (int k = 0; k < n; k++) dev[size_t(k)] = k;
        if (eu_hip_init_devices(dev.data(), n) != 0) n = 1;
      }
      return n;
    }();
    if (args.tethered) {
      t.out_format = EU_OUT_SRGBA8;
      if (slots > 1)
        return eu_hip_render_devices(&t, srcs.data(), int(srcs.size()), (float *)args.p_screen_data,
                                     size_t(w) * sizeof(uint32_t), 0);
      return eu_hip_render(&t, srcs.data(), int(srcs.size()), (float *)args.p_screen_data,
                           size_t(w) * sizeof(uint32_t), 0, nullptr);
    }
    if (slots > 1)
      return eu_hip_render_devices(&t, srcs.data(), int(srcs.size()), args.p_output,
                                   size_t(w) * nchannels * sizeof(float), 0);
    return eu_hip_render(&t, srcs.data(), int(srcs.size()), args.p_output,
                         size_t(w) * nchannels * sizeof(float), 0, nullptr);
  }

  // conclude_cycle / asset_handler.cycle (environment.h:202-227): drop everything
  void clear() const
  {
    for (auto &kv : resident) eu_hip_source_release(kv.second);
    resident.clear();
  }
  ~hip_dispatch() { clear(); }
};

inline const dispatch_base *get_dispatch()
{
  static hip_dispatch d;
  return &d;
}

}  // namespace project
#endif
