// envutil_hip - envutil's command line on the MI355X path.
//
//   envutil_hip --facet pano.pfm spherical 360 0 0 0 --projection cubemap --hfov 90
//               --width 1024 --degree 3 --output cube.pfm
//   envutil_hip --pto project.pto --output pano.pfm
//   envutil_hip <fixed options> -          (pipe mode: one job per line of stdin)
//
// The same options, defaults and PTO handling as the reference's main() / core()
// (envutil_main.cc:1634-1727, :1948-1982): include/eu_frontend.hpp fills project::args,
// get_dispatch()->payload() renders (include/eu_dispatch.hpp -> libeu_hip.so), and the images
// go through include/eu_image_io.hpp (PFM / PNM / PAM / Radiance instead of OpenImageIO; the sRGB / Rec709 /
// linear transfer pairs of OpenImageIO's built-in colour configuration, --input / --working / --output_colour_space
// and the PTO's Csp clause honoured, any other colour space an error message). Sources stay resident in HBM between the jobs of a pipe-mode session, as the
// reference's asset_handler keeps them in RAM. There is no CPU rendering path: without an
// MI355X every job fails with the library's error.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>
#include "eu_frontend.hpp"
#include "eu_image_io.hpp"
#include "eu_imageprep.hpp"

using namespace project;

// tokenize (envutil_basic.cc:323-420): blanks separate, single or double quotes group, a
// backslash inside quotes carries the quote sign
static std::vector<std::string> tokenize(const std::string &s)
{
  std::vector<std::string> out;
  std::string tok;
  char quote = 0;
  bool in_tok = false;
  for (size_t i = 0; i < s.size(); i++) {
    const char c = s[i];
    if (quote) {
      if (c == '\\' && i + 1 < s.size() && s[i + 1] == quote) { tok += quote; i++; }
      else if (c == quote) { out.push_back(tok); tok.clear(); quote = 0; in_tok = false; }
      else tok += c;
    } else if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
      if (in_tok) { out.push_back(tok); tok.clear(); in_tok = false; }
    } else if (!in_tok && (c == '"' || c == '\'')) { quote = c; in_tok = true; }
    else { tok += c; in_tok = true; }
  }
  if (in_tok) out.push_back(tok);
  return out;
}

// image_series (envutil_basic.h:207-262): a format string with one %-sequence taking an integer
// The reference formats whenever the string holds exactly one '%' and returns it verbatim otherwise; a
// conversion other than an integer one ("%s", "%n") is undefined behaviour there. Here the one '%' has to
// introduce "%[flags][width]d" (or i / u / x / X / o) - anything else is returned verbatim, never handed to printf
// (the string arrives from the command line and, in pipe mode, from stdin).
static std::string series_name(const std::string &fmt, int index)
{
  size_t npct = 0, pos = 0;
  for (size_t i = 0; i < fmt.size(); i++) if (fmt[i] == '%') { npct++; pos = i; }
  if (npct != 1) return fmt;
  size_t k = pos + 1;
  while (k < fmt.size() && (fmt[k] == '0' || fmt[k] == '-' || fmt[k] == '+' || fmt[k] == ' ')) k++;
  while (k < fmt.size() && fmt[k] >= '0' && fmt[k] <= '9') k++;
  if (k >= fmt.size() || !std::strchr("diuxXo", fmt[k]) || k - pos > 8) return fmt;
  std::vector<char> buf(fmt.size() + 32);
  std::snprintf(buf.data(), buf.size(), fmt.c_str(), index);
  return buf.data();
}

static int run_payload(const std::string &output)
{
  int ow = args.width, oh = args.height;
  if (args.store_cropped) { ow = args.p_crop_x1 - args.p_crop_x0; oh = args.p_crop_y1 - args.p_crop_y0; }
  std::vector<float> out(size_t(ow) * oh * args.nchannels);
  args.p_output = out.data();
  const int rc = get_dispatch()->payload(args.nchannels, args.twine ? 9 : 3, args.projection);
  args.p_output = nullptr;
  if (rc != 0) {
    std::fprintf(stderr, "envutil_hip: render failed (%d): %s\n", rc, eu_hip_last_error());
    return 1;
  }
  std::string err;
  const bool cube = (args.projection == CUBEMAP || args.projection == BIATAN6) && !args.store_cropped;
  // save_array attaches the target's projection and hfov (degrees) to the file (envutil_basic.h:770-772)
  io::metadata meta;
  meta.projection = projection_name[args.projection];
  meta.hfov = (180.0 / M_PI) * args.hfov;
  // save_array: from the working colour space to the output's where they differ (envutil_basic.h:786-812)
  if (!io::convert_colour(out.data(), size_t(ow) * oh, args.nchannels, args.working_colour_space, args.colour_space, err)) {
    std::fprintf(stderr, "envutil_hip: %s\n", err.c_str());
    return 1;
  }
  if (!io::write_image(output, out.data(), ow, oh, args.nchannels, cube, err, &meta)) {
    std::fprintf(stderr, "envutil_hip: %s\n", err.c_str());
    return 1;
  }
  if (args.verbose) std::printf("saved %s (%d x %d x %d)\n", output.c_str(), ow, oh, args.nchannels);
  return 0;
}

// core(), envutil_main.cc:1634-1727
static int core(int argc, const char *const *argv)
{
  std::string err;
  image_probe probe = [&](const std::string &name, image_info &info) {
    std::string e;
    io::metadata m;
    if (!io::probe(name, info.width, info.height, info.nchannels, e, &m)) return false;
    info.projection = m.projection; info.hfov = m.hfov;
    return true;
  };
  if (!init_arguments(argc, argv, probe, err)) {
    std::fprintf(stderr, "envutil_hip: %s\n", err.c_str());
    return 2;
  }
  const auto *dp = static_cast<const hip_dispatch *>(get_dispatch());
  if (args.verbose) std::printf("using %s (%s)\n", dp->hwy_target_name.c_str(), dp->hwy_target_str.c_str());
  if (!args.twine_setup()) {
    std::fprintf(stderr, "envutil_hip: cannot read the tap table %s\n", args.twf_file.c_str());
    return 2;
  }

  // pixels of the facets that are not resident yet (asset_handler, environment.h:84-227)
  std::vector<std::vector<float>> pixels(args.facet_spec_v.size());
  for (size_t k = 0; k < args.facet_spec_v.size(); k++) {
    facet_spec &f = args.facet_spec_v[k];
    if (args.solo >= 0 && int(k) != args.solo) continue;
    if (dp->resident.count(f.asset_key)) {
      if (args.verbose) std::printf("asset %s is already resident\n", f.asset_key.c_str());
      continue;
    }
    int w = 0, h = 0, nch = 0;
    if (!io::read_image(f.filename, pixels[k], w, h, nch, err)) {
      std::fprintf(stderr, "envutil_hip: %s\n", err.c_str());
      return 2;
    }
    // read_image_data: from the facet's colour space - Csp clause / --input_colour_space, else what the file's
    // format says - to the working one (envutil_basic.h:950-977)
    {
      const std::string csp = f.colour_space.empty() ? io::file_colour_space(f.filename) : f.colour_space;
      if (args.verbose && csp != args.working_colour_space)
        std::printf("converting %s from %s to %s\n", f.filename.c_str(), csp.c_str(), args.working_colour_space.c_str());
      if (!io::convert_colour(pixels[k].data(), size_t(w) * h, nch, csp, args.working_colour_space, err)) {
        std::fprintf(stderr, "envutil_hip: %s: %s\n", f.filename.c_str(), err.c_str());
        return 2;
      }
    }
    const bool cube = f.projection == CUBEMAP || f.projection == BIATAN6;
    if (w != f.window_width || (cube ? h != 6 * w : h != f.window_height)) {
      std::fprintf(stderr, "envutil_hip: %s is %d x %d, expected %d x %d\n", f.filename.c_str(), w, h,
                   f.window_width, cube ? 6 * f.window_width : f.window_height);
      return 2;
    }
    // PTO masks and lens crops edit the loaded pixels (environment.h:700-890)
    if (!prepare_facet_pixels(f, pixels[k], nch, err)) {
      std::fprintf(stderr, "envutil_hip: %s: %s\n", f.filename.c_str(), err.c_str());
      return 2;
    }
    f.pixels = pixels[k].data();
  }

  if (!args.split.empty()) {
    int rc = 0;
    const int nfacets = args.nfacets;
    for (int i = 0; i < nfacets && rc == 0; i++) {
      if (i == args.solo) continue;
      static_cast<facet_base &>(args) = args.facet_spec_v[size_t(i)];
      args.single = i;
      args.store_cropped = false;
      args.output = series_name(args.split, i);
      rc = run_payload(args.output);
    }
    return rc;
  }
  if (args.single != -1) args.store_cropped = false;
  return run_payload(args.output);
}

int main(int argc, const char **argv)
{
  if (argc < 2) {
    std::fprintf(stderr, "usage: envutil_hip <envutil options> (see README.md); a trailing '-' reads jobs from stdin\n");
    return 2;
  }
  for (int i = 1; i < argc; i++)
    if (std::string(argv[i]) == "--help" || std::string(argv[i]) == "-h") {
      std::puts(
        "envutil_hip - envutil's reprojection on an MI355X (options as in envutil, envutil_main.cc:190-372)\n"
        "  source:   --facet IMAGE PROJECTION HFOV YAW PITCH ROLL (repeatable) | --photo IMAGE | --pto FILE [--pto_line LINE]\n"
        "            projections: spherical cylindrical rectilinear stereographic fisheye cubemap biatan6\n"
        "            --solo N  --single N  --split FORMAT(%d)  --mask_for N  --synopsis panorama|hdr_merge\n"
        "  target:   --output FILE  --projection P  --hfov DEG  --width W  --height H  --yaw --pitch --roll DEG\n"
        "            --x0 --x1 --y0 --y1 (extent instead of hfov)  --nchannels N  --brighten F\n"
        "  spline:   --degree D  --prefilter D  --support_min PX  --tile_size PX (cubemap sources)\n"
        "  twining:  --twine N (-1: automatic)  --twine_width F  --twine_density F  --twine_max N\n"
        "            --twine_sigma F  --twine_threshold F  --twine_normalize  --twine_precise  --twf_file FILE\n"
        "  images:   .pfm .hdr (float), .pgm .ppm .pnm (8 / 16 bit), .pam (8 / 16 bit, alpha); a name with one %s\n"
        "            stands for six cube faces: left right top bottom front back\n"
        "  -v verbose; a trailing '-' reads one job per line from stdin (sources stay resident in HBM)");
      return 0;
    }
  if (std::string(argv[argc - 1]) != "-") return core(argc, argv);
  argc--;
  int rc = 0;
  std::string line;
  while (std::getline(std::cin, line)) {
    const std::vector<std::string> sv = tokenize(line);
    if (sv.empty()) continue;
    std::vector<const char *> av(argv, argv + argc);
    for (const auto &t : sv) av.push_back(t.c_str());
    const int r = core(int(av.size()), av.data());
    if (r) rc = r;
  }
  return rc;
}
