// eu_imageprep.hpp - what source_t's constructor does to a facet's pixels between
// read_image_data and the prefilter (environment.h:700-890), for hosts that load the images
// themselves (tools/envutil_hip.cc): the alpha channel a masked or cropped facet gains
// (nchannels 1 -> 2, 3 -> 4, set to 1), the PTO exclude masks (k-lines, variant 0) and the lens
// crop (S clause: elliptic for fisheye images, rectangular otherwise) cleared in an alpha
// plane, that plane softened by a 5-tap binomial and multiplied into every channel. The
// arithmetic is the library's host function eu_hip_facet_alpha (include/eu_hip.h).
#ifndef EU_IMAGEPREP_HPP
#define EU_IMAGEPREP_HPP

#include <string>
#include <vector>
#include "eu_dispatch.hpp"

namespace project {

// pixels: window_width x window_height x native_nchannels on entry, x f.nchannels on return
inline bool prepare_facet_pixels(facet_spec &f, std::vector<float> &pixels, int native_nchannels, std::string &err)
{
  const bool cube = f.projection == CUBEMAP || f.projection == BIATAN6;
  const int w = f.window_width, h = cube ? 6 * f.window_width : f.window_height;
  const size_t npix = size_t(w) * size_t(h);
  if (pixels.size() != npix * size_t(native_nchannels)) { err = "pixel buffer does not match the facet's window"; return false; }
  if (native_nchannels != f.nchannels) {
    // only masks and crops raise a facet's channel count (envutil_main.cc:1062-1075)
    if (!(f.has_lens_crop || f.has_pto_mask) || f.nchannels != native_nchannels + 1 ||
        (f.nchannels != 2 && f.nchannels != 4)) {
      err = "the image has " + std::to_string(native_nchannels) + " channels, the facet " + std::to_string(f.nchannels);
      return false;
    }
    std::vector<float> wide(npix * size_t(f.nchannels));
    for (size_t i = 0; i < npix; i++) {
      for (int c = 0; c < native_nchannels; c++) wide[i * f.nchannels + c] = pixels[i * native_nchannels + c];
      wide[i * f.nchannels + native_nchannels] = 1.0f;
    }
    pixels.swap(wide);
  }
  if (f.has_lens_crop || f.has_pto_mask) {
    if (f.nchannels != 2 && f.nchannels != 4) { err = "a masked or cropped facet needs an alpha channel"; return false; }
    std::vector<eu_mask_polygon> polys;
    for (const auto &m : f.pto_mask_v)
      if (m.variant == 0) polys.push_back({ int(m.vx.size()), m.vx.data(), m.vy.data() });   // other variants: ignored, as there
    const int kind = !f.has_lens_crop ? 0 : f.projection == FISHEYE ? 2 : 1;
    const int rc = eu_hip_facet_alpha(pixels.data(), w, h, f.nchannels, polys.data(), int(polys.size()), kind,
                                      f.crop_x0, f.crop_x1, f.crop_y0, f.crop_y1, nullptr);
    if (rc != EU_OK) { err = eu_hip_last_error(); return false; }
  }
  f.pixels_prepared = true;
  return true;
}

}  // namespace project
#endif
