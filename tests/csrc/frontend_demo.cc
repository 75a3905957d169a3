// Parses an envutil command line with include/eu_frontend.hpp and prints the resulting
// project::args as JSON (TEST CODE). Image sizes come from the environment:
//   EU_TEST_IMAGES="pano.tif=4000x2000x3;img0.jpg=3000x2000x3"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "eu_frontend.hpp"

static std::string esc(const std::string &s)
{
  std::string o;
  for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; }
  return o;
}

int main(int argc, char **argv)
{
  using namespace project;
  std::map<std::string, image_info> table;
  if (const char *e = std::getenv("EU_TEST_IMAGES")) {
    std::string s(e);
    size_t p = 0;
    while (p < s.size()) {
      size_t q = s.find(';', p);
      if (q == std::string::npos) q = s.size();
      const std::string item = s.substr(p, q - p);
      const size_t eq = item.find('=');
      image_info info;
      if (eq != std::string::npos &&
          std::sscanf(item.c_str() + eq + 1, "%dx%dx%d", &info.width, &info.height, &info.nchannels) == 3)
        table[item.substr(0, eq)] = info;
      p = q + 1;
    }
  }
  image_probe probe = [&](const std::string &name, image_info &info) {
    auto it = table.find(name);
    if (it == table.end()) return false;
    info = it->second;
    return true;
  };
  std::string err;
  if (!init_arguments(argc, argv, probe, err)) {
    std::printf("{\"ok\": false, \"error\": \"%s\"}\n", err.c_str());
    return 2;
  }
  args.twine_setup();
  const arguments &a = args;
  std::printf("{\"ok\": true, \"output\": \"%s\", \"projection\": %d, \"width\": %d, \"height\": %d, "
              "\"hfov\": %.17g, \"x0\": %.17g, \"x1\": %.17g, \"y0\": %.17g, \"y1\": %.17g, \"step\": %.17g, "
              "\"yaw\": %.17g, \"pitch\": %.17g, \"roll\": %.17g, \"degree\": %d, \"prefilter\": %d, "
              "\"twine\": %d, \"twine_width\": %.9g, \"nchannels\": %d, \"nfacets\": %d, \"solo\": %d, "
              "\"single\": %d, \"store_cropped\": %d, \"crop\": [%d, %d, %d, %d], \"support_min\": %d, "
              "\"tile_size\": %d, \"synopsis\": \"%s\",\n \"spread\": [",
              esc(a.output).c_str(), int(a.projection), a.width, a.height, a.hfov, a.x0, a.x1, a.y0, a.y1, a.step,
              a.yaw, a.pitch, a.roll, a.spline_degree, a.prefilter_degree, a.twine, double(a.twine_width),
              a.nchannels, a.nfacets, a.solo, a.single, int(a.store_cropped), a.p_crop_x0, a.p_crop_x1,
              a.p_crop_y0, a.p_crop_y1, a.support_min, a.tile_size, a.synopsis.c_str());
  for (size_t i = 0; i < a.twine_spread.size(); i++)
    std::printf("%s[%.9g, %.9g, %.9g]", i ? ", " : "", double(a.twine_spread[i][0]), double(a.twine_spread[i][1]),
                double(a.twine_spread[i][2]));
  std::printf("],\n \"facets\": [");
  for (size_t i = 0; i < a.facet_spec_v.size(); i++) {
    const facet_spec &f = a.facet_spec_v[i];
    std::printf("%s{\"filename\": \"%s\", \"asset_key\": \"%s\", \"projection\": %d, \"hfov\": %.17g, "
                "\"width\": %d, \"height\": %d, \"window\": [%d, %d, %d, %d], \"nchannels\": %d, "
                "\"yaw\": %.17g, \"pitch\": %.17g, \"roll\": %.17g, \"x0\": %.17g, \"x1\": %.17g, \"y0\": %.17g, "
                "\"y1\": %.17g, \"step\": %.17g, \"brighten\": %.9g, \"a\": %.17g, \"b\": %.17g, \"c\": %.17g, "
                "\"h\": %.17g, \"v\": %.17g, \"s\": %.17g, \"shear_g\": %.17g, \"shear_t\": %.17g, "
                "\"has_lcp\": %d, \"has_shift\": %d, \"has_shear\": %d, \"tr\": [%.17g, %.17g, %.17g], "
                "\"has_lens_crop\": %d, \"has_pto_mask\": %d, \"masked\": %d, \"lens_crop\": [%d, %d, %d, %d], "
                "\"masks\": [",
                i ? ",\n  " : "", esc(f.filename).c_str(), esc(f.asset_key).c_str(), int(f.projection), f.hfov, f.width,
                f.height, f.window_width, f.window_height, f.window_x_offset, f.window_y_offset, f.nchannels,
                f.yaw, f.pitch, f.roll, f.x0, f.x1, f.y0, f.y1, f.step, double(f.brighten), f.a, f.b, f.c, f.h,
                f.v, f.s, f.shear_g, f.shear_t, int(f.has_lcp), int(f.has_shift), int(f.has_shear), f.tr_x,
                f.tr_y, f.tr_z, int(f.has_lens_crop), int(f.has_pto_mask), f.masked, f.crop_x0, f.crop_x1, f.crop_y0,
                f.crop_y1);
    for (size_t m = 0; m < f.pto_mask_v.size(); m++) {
      std::printf("%s{\"variant\": %d, \"xy\": [", m ? ", " : "", f.pto_mask_v[m].variant);
      for (size_t v = 0; v < f.pto_mask_v[m].vx.size(); v++)
        std::printf("%s%.9g, %.9g", v ? ", " : "", double(f.pto_mask_v[m].vx[v]), double(f.pto_mask_v[m].vy[v]));
      std::printf("]}");
    }
    std::printf("]}");
  }
  std::printf("]}\n");
  if (std::getenv("EU_TEST_RENDER")) {
    // front end -> dispatch -> HIP: synthetic pixels for every facet, payload(), a checksum
    std::vector<std::vector<float>> px(args.facet_spec_v.size());
    for (size_t k = 0; k < args.facet_spec_v.size(); k++) {
      facet_spec &f = args.facet_spec_v[k];
      const int w = f.window_width, h = (f.projection == CUBEMAP || f.projection == BIATAN6) ? 6 * f.width : f.window_height;
      px[k].resize(size_t(w) * h * f.nchannels);
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
          for (int c = 0; c < f.nchannels; c++)
            px[k][(size_t(y) * w + x) * f.nchannels + c] =
              (c == f.nchannels - 1 && (f.nchannels == 2 || f.nchannels == 4))
                ? 1.0f : 0.5f + 0.25f * float((x * 7 + y * 13 + c * 29 + int(k) * 5) % 97) / 97.0f;
      f.pixels = px[k].data();
    }
    int ow = args.width, oh = args.height;
    if (args.store_cropped) { ow = args.p_crop_x1 - args.p_crop_x0; oh = args.p_crop_y1 - args.p_crop_y0; }
    std::vector<float> out(size_t(ow) * oh * args.nchannels);
    args.p_output = out.data();
    const int rc = get_dispatch()->payload(args.nchannels, args.twine ? 9 : 3, args.projection);
    std::printf("rc %d\n", rc);
    if (rc != 0) { std::printf("error: %s\n", eu_hip_last_error()); return rc == EU_ERR_NO_DEVICE ? 3 : 1; }
    unsigned long long hsum = 1469598103934665603ull;
    for (float v : out) { unsigned u; std::memcpy(&u, &v, 4); hsum = (hsum ^ u) * 1099511628211ull; }
    std::printf("fnv1a %016llx\n", hsum);
  }
  return 0;
}
