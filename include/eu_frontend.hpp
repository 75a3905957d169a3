// eu_frontend.hpp - OpenImageIO-free front end of the reprojection path: envutil's command
// line and PTO scripts -> the global project::args that hip_dispatch::payload() reads.
//
// Mirrors, option for option and default for default:
//   arguments::init            envutil_main.cc:178-1251   (project::init_arguments)
//   pto_parser_type            pto.h:63-200                (project::pto_script)
//   facet_spec::init           envutil_basic.h:552-629     (--facet IMAGE PRJ HFOV YAW PITCH ROLL)
// The automatic twining of arguments::twine_setup (envutil_main.cc:1405-1616) is in
// eu_dispatch.hpp. What the reference gets from OpenImageIO here is an image file's width,
// height and channel count (facet_base::get_image_metrics, envutil_basic.h:546-589): the host
// passes a callback for that (`image_probe`). Everything that is file I/O or colour
// management stays out: the colour-space options are parsed and stored (arguments::*_colour_space,
// facet_spec::colour_space) for the program that owns the files (tools/envutil_hip.cc converts), the
// --oiio pass-through options. --photo images take projection and hfov from the metadata the probe reports.
//
// Errors: the reference asserts or exits; this returns false and a message.
// Header-only; link with -leu_hip (get_extent / get_step are the library's).
#ifndef EU_FRONTEND_HPP
#define EU_FRONTEND_HPP

#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <regex>
#include <string>
#include <vector>
#include "eu_dispatch.hpp"

namespace project {

// projection / hfov: the image's "Projection" and "Hfov" (degrees) metadata where the file carries them
// (envutil writes both into its output, save_array, envutil_basic.h:770-772); empty / negative: absent
struct image_info { int width = 0, height = 0, nchannels = 0; std::string projection; double hfov = -1.0; };
typedef std::function<bool(const std::string &filename, image_info &info)> image_probe;

static const char *const projection_name[] = { "spherical", "cylindrical", "rectilinear", "stereographic",
                                               "fisheye", "cubemap", "biatan6" };

// ---- pto.h:63-200 ---------------------------------------------------------------------
struct pto_line_type
{
  std::string original, head;
  std::map<std::string, std::string> field_map;
};

struct pto_script
{
  std::map<std::string, std::vector<pto_line_type>> line_group;

  // one line: "<letter> <item> <item> ...", an item is <name><value> with the value a quoted
  // string or a run of non-blanks; "=N" takes the value of the same field of i-line N
  // (pto.h:98-160). Lines that do not look like that are ignored, as there.
  bool parse_pto_line(const std::string &s, std::string &err)
  {
    static const std::regex line_re("([a-zA-Z])\\s(.+)[\n\r]*");
    static const std::regex item_re("([A-Za-z]+)((\"[^\"]+\")|(\\S*))");
    std::smatch parts;
    if (!std::regex_match(s, parts, line_re)) return true;
    pto_line_type line;
    line.head = parts[1].str();
    line.original = s;
    const std::string tail = parts[2].str();
    for (auto i = std::sregex_iterator(tail.begin(), tail.end(), item_re); i != std::sregex_iterator(); ++i) {
      const std::string item = i->str();
      std::smatch p2;
      if (!std::regex_match(item, p2, item_re)) continue;
      const std::string name = p2[1].str();
      std::string value = p2[2].str();
      if (!value.empty() && value[0] == '=') {
        if (name != "j") {
          int ref = 0;
          try { ref = std::stoi(value.substr(1)); } catch (...) { err = "bad back reference in PTO line: " + s; return false; }
          auto &il = line_group["i"];
          if (ref < 0 || size_t(ref) >= il.size()) { err = "PTO back reference to a missing i-line: " + s; return false; }
          value = il[size_t(ref)].field_map[name];
        }
      }
      line.field_map[name] = value;
    }
    line_group[line.head].push_back(line);
    return true;
  }

  bool read_pto_file(const std::string &filename, const std::vector<std::string> &addenda, std::string &err)
  {
    if (!filename.empty()) {
      std::ifstream str(filename);
      if (!str) { err = "could not open pto file " + filename; return false; }
      std::string buffer;
      while (std::getline(str, buffer))
        if (!parse_pto_line(buffer, err)) return false;
    }
    for (const auto &l : addenda)
      if (!parse_pto_line(l, err)) return false;
    return true;
  }
};

namespace detail {

inline double glean(const std::string &s) { return s.empty() ? 0.0 : std::stod(s); }
inline int iglean(const std::string &s) { return s.empty() ? 0 : std::stoi(s); }
inline std::string unquote(const std::string &s)
{
  return (!s.empty() && s[0] == '"' && s.size() >= 2) ? s.substr(1, s.size() - 2) : s;
}
inline bool four_ints(const std::string &s, int *v)
{
  static const std::regex re("([0-9]+),([0-9]+),([0-9]+),([0-9]+)");
  std::smatch p;
  if (!std::regex_match(s, p, re)) return false;
  for (int i = 0; i < 4; i++) v[i] = std::stoi(p[size_t(i) + 1].str());
  return true;
}
inline int projection_index(const std::string &s)
{
  for (int i = 0; i < 7; i++) if (s == projection_name[i]) return i;
  return 7;
}
// the extent and step of a facet from its projection, size and hfov (get_extent / get_step)
inline void facet_extent(facet_spec &f)
{
  double e[4];
  eu_hip_get_extent(f.projection, f.width, f.height, f.hfov, e);
  f.x0 = e[0]; f.x1 = e[1]; f.y0 = e[2]; f.y1 = e[3];
  f.step = eu_hip_get_step(f.projection, f.width, f.height, f.hfov);
}

}  // namespace detail

// arguments::init (envutil_main.cc:178-1251): fills project::args. argv[0] is the program name.
inline bool init_arguments(int argc, const char *const *argv, const image_probe &probe, std::string &err)
{
  using namespace detail;
  arguments &a = args;
  a = arguments();
  // ---- the option table (envutil_main.cc:190-372): name -> number of values; repeatable lists
  static const std::map<std::string, int> nvalues = {
    { "-v", 0 }, { "--output", 1 }, { "--projection", 1 }, { "--hfov", 1 }, { "--width", 1 }, { "--height", 1 },
    { "--support_min", 1 }, { "--tile_size", 1 }, { "--synopsis", 1 }, { "--working_colour_space", 1 },
    { "--output_colour_space", 1 }, { "--single", 1 }, { "--split", 1 }, { "--yaw", 1 }, { "--pitch", 1 },
    { "--roll", 1 }, { "--x0", 1 }, { "--x1", 1 }, { "--y0", 1 }, { "--y1", 1 }, { "--brighten", 1 },
    { "--prefilter", 1 }, { "--degree", 1 }, { "--twine", 1 }, { "--twf_file", 1 }, { "--twine_normalize", 0 },
    { "--twine_precise", 0 }, { "--twine_width", 1 }, { "--twine_density", 1 }, { "--twine_sigma", 1 },
    { "--twine_threshold", 1 }, { "--twine_max", 1 }, { "--photo", 1 }, { "--facet", 6 }, { "--oiio", 1 },
    { "--input_colour_space", 1 }, { "--pto", 1 }, { "--pto_line", 1 }, { "--solo", 1 }, { "--mask_for", 1 },
    { "--nchannels", 1 } };
  std::map<std::string, std::string> opt;
  std::vector<std::vector<std::string>> facets;
  std::vector<std::string> photos, addenda;
  for (int i = 1; i < argc; i++) {
    const std::string key = argv[i];
    auto it = nvalues.find(key);
    if (it == nvalues.end()) { err = "unknown option " + key; return false; }
    if (i + it->second >= argc) { err = "option " + key + " needs " + std::to_string(it->second) + " value(s)"; return false; }
    std::vector<std::string> v(argv + i + 1, argv + i + 1 + it->second);
    i += it->second;
    if (key == "--facet") facets.push_back(v);
    else if (key == "--photo") photos.push_back(v[0]);
    else if (key == "--pto_line") addenda.push_back(v[0]);
    else if (key == "--oiio") continue;
    else opt[key] = it->second ? v[0] : "1";
  }
  auto str = [&](const char *k, const std::string &d) { auto it = opt.find(k); return it == opt.end() ? d : it->second; };
  auto has = [&](const char *k) { return opt.count(k) != 0; };
  bool bad = false;
  // ArgParse's get<float> / get<int>: the value passes through that type
  auto fl = [&](const char *k, float d) -> float {
    if (!has(k)) return d;
    try { return std::stof(opt[k]); } catch (...) { bad = true; err = std::string("bad number for ") + k; return d; }
  };
  auto in = [&](const char *k, int d) -> int {
    if (!has(k)) return d;
    try { return std::stoi(opt[k]); } catch (...) { bad = true; err = std::string("bad number for ") + k; return d; }
  };
  a.verbose = has("-v");
  a.twine_normalize = has("--twine_normalize");
  a.twine_precise = has("--twine_precise");
  a.output = str("--output", "");
  a.pto_file = str("--pto", "");
  a.twf_file = str("--twf_file", "");
  a.split = str("--split", "");
  a.synopsis = str("--synopsis", "panorama");
  a.input_colour_space = str("--input_colour_space", "");
  a.working_colour_space = str("--working_colour_space", "Linear");
  a.colour_space = str("--output_colour_space", "Linear");
  a.prefilter_degree = in("--prefilter", -1);
  a.spline_degree = in("--degree", 1);
  a.twine = in("--twine", -1);
  a.twine_width = fl("--twine_width", 1.0f);
  a.twine_density = fl("--twine_density", 1.0f);
  a.twine_sigma = fl("--twine_sigma", 0.0f);
  a.twine_threshold = fl("--twine_threshold", 0.0f);
  a.twine_max = in("--twine_max", 8);
  a.x0 = fl("--x0", 0.0f); a.x1 = fl("--x1", 0.0f); a.y0 = fl("--y0", 0.0f); a.y1 = fl("--y1", 0.0f);
  a.width = in("--width", 0);
  a.height = in("--height", 0);
  a.hfov = fl("--hfov", 90.0f);
  a.tile_size = in("--tile_size", 64);
  a.support_min = in("--support_min", 8);
  if (a.hfov != 0.0) a.x0 = a.x1 = a.y0 = a.y1 = 0;
  a.yaw = fl("--yaw", 0.0f); a.pitch = fl("--pitch", 0.0f); a.roll = fl("--roll", 0.0f);
  a.brighten = fl("--brighten", 1.0f);
  a.projection_str = str("--projection", "rectilinear");
  if (bad) return false;
  if (a.prefilter_degree < 0) a.prefilter_degree = a.spline_degree;
  a.projection = projection_t(projection_index(a.projection_str));
  if (a.projection == PRJ_NONE) { err = "unknown projection " + a.projection_str; return false; }
  if (a.pto_file.empty() && addenda.empty() && facets.empty() && photos.empty()) { err = "no facet, photo or PTO input"; return false; }
  if (a.output.empty() && a.split.empty()) { err = "no --output (or --split)"; return false; }
  bool ignore_p_line = false;
  a.solo = -1;
  if (a.width == 0) a.width = 1024;
  else ignore_p_line = true;
  if (a.projection == CUBEMAP || a.projection == BIATAN6) {
    a.height = 6 * a.width;
    if (a.hfov < 90.0) { err = "cubemap targets need hfov >= 90"; return false; }
  }
  if (a.projection == SPHERICAL && a.height == 0) {
    if (a.width & 1) ++a.width;
    a.height = a.width / 2;
  }
  if (a.height == 0) a.height = a.width;

  bool p_line_present = false;
  projection_t p_prj = PRJ_NONE;
  int p_w = 0, p_h = 0;
  double p_hfov = 0.0, p_eev = 0.0;
  float eev_sum = 0.0f;
  int eev_count = 0;
  image_info info;      // of the image probed last (its metadata serve --photo / "metadata" facets)
  auto metrics = [&](facet_spec &f) -> bool {
    info = image_info();
    if (!probe || !probe(f.filename, info) || info.width <= 0 || info.height <= 0 || info.nchannels <= 0) {
      err = "failed to open facet image '" + f.filename + "'";
      return false;
    }
    f.width = f.window_width = info.width;
    f.height = f.window_height = info.height;
    f.window_x_offset = f.window_y_offset = 0;
    f.nchannels = info.nchannels;
    return true;
  };

  if (!a.pto_file.empty() || !addenda.empty()) {
    pto_script parser;
    if (!parser.read_pto_file(a.pto_file, addenda, err)) return false;
    try {
      if (!ignore_p_line) {
        auto &pl = parser.line_group["p"];
        if (!pl.empty()) {
          p_line_present = true;
          auto &dir = pl[0].field_map;                   // further p-lines are ignored
          static const projection_t pmap[5] = { RECTILINEAR, CYLINDRICAL, SPHERICAL, FISHEYE, STEREOGRAPHIC };
          const int prj = std::stoi(dir["f"]);
          p_prj = (prj >= 0 && prj <= 4) ? pmap[prj] : PRJ_NONE;
          if (p_prj == PRJ_NONE) { err = "can't handle PTO projection code " + dir["f"] + " in p-line"; return false; }
          p_w = iglean(dir["w"]); p_h = iglean(dir["h"]);
          p_hfov = (M_PI / 180.0) * glean(dir["v"]);
          p_eev = glean(dir["Eev"]);
          if (!dir["S"].empty()) {
            int v[4];
            if (!four_ints(dir["S"], v)) { err = "malformed S clause in p-line"; return false; }
            a.store_cropped = true;
            a.p_crop_x0 = v[0]; a.p_crop_x1 = v[1]; a.p_crop_y0 = v[2]; a.p_crop_y1 = v[3];
          }
        }
      }
      for (auto &il : parser.line_group["i"]) {
        auto &dir = il.field_map;
        facet_spec f;
        f.facet_no = a.nfacets++;
        if (!dir["Pano"].empty()) {
          // envutil's extension for unstitching: the facet IS the panorama of the p-line
          if (!p_line_present) { err = "a Pano clause needs a p-line"; return false; }
          f.filename = unquote(dir["Pano"]);
          f.asset_key = dir["Pano"];
          f.projection = p_prj;
          f.hfov = p_hfov;
          if (!metrics(f)) return false;
          if (a.store_cropped) {
            if (a.p_crop_x1 - a.p_crop_x0 != f.width || a.p_crop_y1 - a.p_crop_y0 != f.height) {
              err = "the Pano image does not have the p-line's crop size";
              return false;
            }
            f.width = p_w; f.height = p_h;
            f.window_x_offset = a.p_crop_x0; f.window_y_offset = a.p_crop_y0;
            if (f.width < f.window_x_offset + f.window_width || f.height < f.window_y_offset + f.window_height) {
              err = "the p-line's crop window lies outside its frame";
              return false;
            }
          }
          a.solo = f.facet_no;
        } else {
          f.filename = unquote(dir["n"]);
          f.asset_key = f.filename;
          // envutil's extension of the i-line: Csp"name", else the blanket --input_colour_space (envutil_main.cc:642-670)
          f.colour_space = dir.count("Csp") && !dir["Csp"].empty() ? unquote(dir["Csp"]) : a.input_colour_space;
          const int prj = std::stoi(dir["f"]);
          if (prj == 0) f.projection = RECTILINEAR;
          else if (prj == 1) f.projection = CYLINDRICAL;
          else if (prj == 2 || prj == 3) f.projection = FISHEYE;
          else if (prj == 4) f.projection = SPHERICAL;
          else if (prj == 10) f.projection = STEREOGRAPHIC;
          else { err = "can't handle PTO projection code " + dir["f"] + " in i-line"; return false; }
          if (!metrics(f)) return false;
          f.hfov = (M_PI / 180.0) * std::stod(dir["v"]);
          if (!dir["W"].empty()) {
            // envutil's extension for cropped input: W<x0>,<x1>,<y0>,<y1> + w, h of the whole image
            int v[4];
            if (!four_ints(dir["W"], v)) { err = "malformed W clause in i-line"; return false; }
            f.window_x_offset = v[0]; f.window_y_offset = v[2];
            f.window_width = v[1] - v[0]; f.window_height = v[3] - v[2];
            if (f.window_width != f.width || f.window_height != f.height) { err = "the W window does not have the image's size"; return false; }
            f.width = iglean(dir["w"]); f.height = iglean(dir["h"]);
            if (f.width == 0 || f.height == 0) { err = "a W clause needs w and h"; return false; }
          }
        }
        f.projection_str = projection_name[f.projection];
        f.yaw = (M_PI / 180.0) * glean(dir["y"]);
        f.pitch = (M_PI / 180.0) * glean(dir["p"]);
        f.roll = (M_PI / 180.0) * glean(dir["r"]);
        f.tr_x = glean(dir["TrX"]); f.tr_y = glean(dir["TrY"]); f.tr_z = -glean(dir["TrZ"]);
        f.tp_y = (M_PI / 180.0) * glean(dir["Tpy"]);
        f.tp_p = (M_PI / 180.0) * glean(dir["Tpp"]);
        f.tp_r = 0.0;
        f.shear_g = glean(dir["g"]) / f.height;
        f.shear_t = glean(dir["t"]) / f.width;
        f.a = glean(dir["a"]); f.b = glean(dir["b"]); f.c = glean(dir["c"]);
        f.h = glean(dir["d"]); f.v = glean(dir["e"]);
        {
          const int ww = f.window_width, wh = f.window_height;   // process_geometry keeps a set window
          f.process_geometry();
          f.window_width = ww; f.window_height = wh;
        }
        f.brighten = float(glean(dir["Eev"]));
        if (f.brighten != 0.0f) { eev_sum += f.brighten; eev_count++; }
        if (!dir["S"].empty()) {
          int v[4];
          if (!four_ints(dir["S"], v)) { err = "malformed S clause in i-line"; return false; }
          f.has_lens_crop = true;
          f.crop_x0 = v[0]; f.crop_x1 = v[1]; f.crop_y0 = v[2]; f.crop_y1 = v[3];
        }
        a.facet_spec_v.push_back(f);
      }
      int mask_no = 0;
      for (auto &kl : parser.line_group["k"]) {
        auto &dir = kl.field_map;
        const int image = std::stoi(dir["i"]);
        if (image < 0 || size_t(image) >= a.facet_spec_v.size()) { err = "k-line refers to a missing image"; return false; }
        auto &fct = a.facet_spec_v[size_t(image)];
        std::string suffix(".");
        if (fct.filename == fct.asset_key) suffix += a.pto_file + ".";
        fct.has_pto_mask = true;
        fct.asset_key += suffix + std::to_string(mask_no++);
        // the polygon: pairs of numbers in the p field (envutil_main.cc:827-866)
        pto_mask_type mask;
        mask.image = image;
        mask.variant = std::stoi(dir["t"]);
        static const std::regex corner_re("([+-]?[0-9.]+)\\s([+-]?[0-9.]+)");
        const std::string &vl = dir["p"];
        for (auto ci = std::sregex_iterator(vl.begin(), vl.end(), corner_re); ci != std::sregex_iterator(); ++ci) {
          mask.vx.push_back(float(std::stod((*ci)[1].str())));
          mask.vy.push_back(float(std::stod((*ci)[2].str())));
        }
        fct.pto_mask_v.push_back(mask);
      }
    } catch (const std::exception &e) {
      err = std::string("malformed number in PTO script: ") + e.what();
      return false;
    }
  }

  for (const auto &ph : photos) facets.push_back({ ph, "metadata", "-1", "0", "0", "0" });
  // free facets come behind the PTO's, whatever the order on the command line
  for (const auto &v : facets) {
    facet_spec f;
    f.filename = v[0];
    f.projection_str = v[1];
    f.colour_space = a.input_colour_space;
    // facet_spec::init (envutil_main.cc:104-176): hfov -1 and the projection "metadata" (what --photo
    // passes for both) are read from the image's metadata, with 65 degrees / rectilinear where absent
    // (get_image_metrics, envutil_basic.h:589-625)
    double hfov_deg = 0.0;
    try {
      hfov_deg = std::stod(v[2]);
      f.yaw = std::stod(v[3]) * (M_PI / 180.0);
      f.pitch = std::stod(v[4]) * (M_PI / 180.0);
      f.roll = std::stod(v[5]) * (M_PI / 180.0);
    } catch (...) { err = "parse of 'facet' argument failed: " + v[0]; return false; }
    const bool read_hfov = hfov_deg == -1.0;
    if (hfov_deg <= 0.0 && !read_hfov) { err = "facet hfov invalid: " + v[2]; return false; }
    if (!metrics(f)) return false;
    if (read_hfov) hfov_deg = info.hfov >= 0.0 ? double(float(info.hfov)) : 65.0;
    if (f.projection_str == "metadata") f.projection_str = info.projection.empty() ? "rectilinear" : info.projection;
    f.hfov = hfov_deg * (M_PI / 180.0);
    f.projection = projection_t(projection_index(f.projection_str));
    if (f.projection == PRJ_NONE) { err = "unknown facet projection " + f.projection_str; return false; }
    f.facet_no = a.nfacets++;
    f.process_geometry();
    f.asset_key = f.filename;
    f.brighten = 0.0f;
    a.facet_spec_v.push_back(f);
  }
  if (a.nfacets == 0) { err = "no facets"; return false; }
  if (a.solo == -1) a.solo = in("--solo", -1);
  a.single = in("--single", -1);
  if (a.solo >= a.nfacets || a.single >= a.nfacets) { err = "--solo / --single beyond the last facet"; return false; }
  if (a.nfacets == 1) a.solo = 0;
  a.mask_for = in("--mask_for", -1);
  if (a.mask_for >= a.nfacets) { err = "--mask_for beyond the last facet"; return false; }

  // brightness from the Eev values (envutil_main.cc:1003-1060), channel counts (:1062-1157)
  a.nchannels = 1;
  bool alpha_seen = false;
  if (eev_count > 0) eev_sum /= eev_count;
  if (p_eev != 0.0) eev_sum = float(p_eev);
  for (auto &m : a.facet_spec_v) {
    if (eev_count) m.brighten = m.brighten == 0.0f ? 1.0f : float(std::pow(2.0, m.brighten - eev_sum));
    else m.brighten = 1.0f;
    if (a.brighten != 1.0f) m.brighten *= a.brighten;
    if ((m.has_pto_mask || m.has_lens_crop) && (m.nchannels == 1 || m.nchannels == 3)) m.nchannels++;
    if (m.nchannels == 2 || m.nchannels == 4) alpha_seen = true;
    if (m.nchannels > a.nchannels) a.nchannels = m.nchannels;
    m.masked = a.mask_for == -1 ? -1 : (m.facet_no == a.mask_for ? 1 : 0);
  }
  if (alpha_seen && a.nchannels == 3) a.nchannels = 4;
  const int nch = in("--nchannels", 0);
  if (nch > 0) a.nchannels = nch;
  if (bad) return false;

  if (a.single >= 0) {
    // the facet's geometry becomes the target's (envutil_main.cc:1170-1182)
    static_cast<facet_base &>(a) = a.facet_spec_v[size_t(a.single)];
  } else if (p_line_present) {
    a.hfov = p_hfov;
    a.projection = p_prj;
    a.projection_str = projection_name[p_prj];
    a.width = p_w;
    a.height = p_h;
  } else {
    a.hfov *= M_PI / 180.0; a.yaw *= M_PI / 180.0; a.pitch *= M_PI / 180.0; a.roll *= M_PI / 180.0;
  }
  a.step = 0.0;
  if (a.hfov != 0.0) {
    double e[4];
    eu_hip_get_extent(a.projection, a.width, a.height, a.hfov, e);
    a.x0 = e[0]; a.x1 = e[1]; a.y0 = e[2]; a.y1 = e[3];
  }
  if (!(a.x0 <= a.x1) || !(a.y0 <= a.y1)) { err = "empty target extent"; return false; }
  a.step = (a.x1 - a.x0) / a.width;
  return true;
}

}  // namespace project
#endif
