// Host code written against the reference's surface: fill the global `args`,
// fetch the dispatch object, call payload() - exactly what envutil's core()
// does (envutil_main.cc:1655-1727) - through include/eu_dispatch.hpp.
// Prints "rc <code>" and, on success, a checksum of the output.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "eu_dispatch.hpp"

int main(int argc, char **argv)
{
  using namespace project;
  const int sw = 256, sh = 128, tw = 64;
  std::vector<float> img(size_t(sw) * sh * 3);
  for (int y = 0; y < sh; y++)
    for (int x = 0; x < sw; x++)
      for (int c = 0; c < 3; c++)
        img[(size_t(y) * sw + x) * 3 + c] = 0.5f + 0.25f * float((x * 7 + y * 13 + c * 29) % 97) / 97.0f;
  facet_spec f;
  f.projection = SPHERICAL; f.width = sw; f.height = sh; f.hfov = 2.0 * M_PI;
  f.nchannels = 3; f.asset_key = "demo"; f.pixels = img.data();
  f.process_geometry();
  args.projection = CUBEMAP; args.width = tw; args.height = 6 * tw; args.hfov = M_PI / 2.0;
  args.yaw = 0.3; args.pitch = -0.2; args.roll = 0.1;
  bool twine = false, screen = false, crop = false;
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "twine")) twine = true;
    if (!std::strcmp(argv[i], "screen")) screen = true;     // tethered: packed sRGBA8 words
    if (!std::strcmp(argv[i], "crop")) crop = true;         // PTO p-line crop
  }
  args.spline_degree = 3; args.twine = twine ? 2 : 0;
  args.facet_spec_v = { f };
  args.target_setup();
  args.twine_setup();
  int ow = args.width, oh = args.height;
  if (crop) {
    args.store_cropped = true;
    args.p_crop_x0 = 5; args.p_crop_x1 = 60; args.p_crop_y0 = 30; args.p_crop_y1 = 301;
    ow = args.p_crop_x1 - args.p_crop_x0; oh = args.p_crop_y1 - args.p_crop_y0;
  }
  // one 32-bit word per channel value (float) or per pixel (tethered)
  std::vector<float> out(size_t(ow) * oh * (screen ? 1 : 3));
  if (screen) { args.tethered = true; args.p_screen_data = out.data(); }
  else args.p_output = out.data();
  const dispatch_base *dp = get_dispatch();
  int rc = dp->payload(3, args.twine ? 9 : 3, args.projection);
  std::printf("rc %d\n", rc);
  if (rc != 0) { std::printf("error: %s\n", eu_hip_last_error()); return rc == EU_ERR_NO_DEVICE ? 3 : 1; }
  uint64_t hsum = 1469598103934665603ull;
  for (float v : out) { uint32_t u; std::memcpy(&u, &v, 4); hsum = (hsum ^ u) * 1099511628211ull; }
  std::printf("fnv1a %016llx\n", (unsigned long long)hsum);
  return 0;
}
