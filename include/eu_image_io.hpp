// eu_image_io.hpp - image files for the stand-alone command line (tools/envutil_hip.cc).
//
// The reference reads and writes every format through OpenImageIO (read_image_data /
// save_array, envutil_basic.h:710-986), which this image does not have. What the path needs
// from a file is what those two functions hand over: interleaved float pixels, x fastest, the
// file's own channel count - and the colour space the samples are in, because the reference converts
// every image whose colour space differs from the working one (envutil_basic.h:950-977) and the output
// when the output's differs (:786-812). OpenImageIO labels a file by its format (oiio:ColorSpace):
// float formats (PFM, Radiance) are linear, 8/16-bit integer formats are display-referred. Here:
// file_colour_space() says "Linear" for .pfm / .hdr / .pic and "sRGB" for PNM / PAM, and convert_colour()
// knows the transfer pairs of OpenImageIO's built-in configuration (no OpenColorIO): sRGB <-> linear (IEC
// 61966-2-1 piecewise curve), Rec709 <-> linear (BT.709 OETF) - the alpha channel is not touched - and the
// names "Linear" / "linear" / "scene_linear" / "lin_srgb" / "lin_rec709", "sRGB" / "srgb" / "sRGB - Texture" /
// "srgb_tx", "Rec709" / "rec709". Any other name is an error message, not silence. DIFFERENCE from the
// reference, stated in INTEGRATION.md: which label OpenImageIO gives a PNM file (sRGB or Rec709, by version)
// cannot be checked here; --input_colour_space / the PTO's Csp clause override it as in the reference.
// Formats, chosen because they need no library and hold linear float or plain integer samples:
//   .pfm            Portable Float Map: "Pf" 1 channel, "PF" 3 channels, "PF4" 4 channels (the
//                   extension several tools use for RGBA); float32, either byte order, rows
//                   bottom to top
//   .pgm .ppm .pnm  binary PNM (P5 / P6), 8 or 16 bit: value / maxval
//   .pam            P7 (GRAYSCALE, GRAYSCALE_ALPHA, RGB, RGB_ALPHA), 8 or 16 bit: the integer
//                   format with an alpha channel
//   .hdr .pic       Radiance RGBE pictures (32-bit_rle_rgbe, -Y h +X w; flat or run-length encoded
//                   scanlines on input, flat on output), the usual container of lat/lon environment maps;
//                   mantissa * 2^(exponent - 136) as in the widely used rgbe.c that OpenImageIO follows
// Writing integer formats follows OpenImageIO's float -> unsigned conversion: clamp to [0, 1],
// scale by maxval, add 0.5, truncate. Cubemaps: one image of aspect 1:6, or six files named by a
// format string with one %s, filled with left, right, top, bottom, front, back (cubeface_series,
// envutil_basic.h:267-340).
//
// Header-only, host code, no dependency on the HIP library.
#ifndef EU_IMAGE_IO_HPP
#define EU_IMAGE_IO_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace project {
namespace io {

struct header
{
  int width = 0, height = 0, nchannels = 0;
  int maxval = 0;          // 0: float samples
  bool little_endian = true;   // PFM
  bool bottom_up = false;
  bool rgbe = false;           // Radiance picture: 4-byte RGBE pixels, flat or run-length encoded scanlines
  long data_offset = 0;
  // "Projection" / "Hfov" (degrees) as envutil's save_array attaches them to its output
  // (envutil_basic.h:770-772): header lines Projection=... / Hfov=... of a Radiance picture, comment
  // lines "# Projection: ..." / "# Hfov: ..." of a PNM / PAM header; PFM has no room for them
  std::string projection;
  double hfov = -1.0;
};

struct metadata { std::string projection; double hfov = -1.0; };

inline std::string lower_ext(const std::string &name)
{
  const size_t dot = name.find_last_of('.');
  if (dot == std::string::npos) return std::string();
  std::string e = name.substr(dot + 1);
  for (auto &c : e) c = char(c >= 'A' && c <= 'Z' ? c - 'A' + 'a' : c);
  return e;
}

// next whitespace-separated token of a PNM header; '#' starts a comment up to the line's end
inline void note_metadata(const std::string &line, char sep, header &h)
{
  const size_t p = line.find(sep);
  if (p == std::string::npos) return;
  std::string key = line.substr(0, p), val = line.substr(p + 1);
  while (!key.empty() && (key[0] == ' ' || key[0] == '#')) key.erase(0, 1);
  while (!val.empty() && (val[0] == ' ')) val.erase(0, 1);
  while (!val.empty() && (val.back() == '\n' || val.back() == '\r' || val.back() == ' ')) val.pop_back();
  if (key == "Projection") h.projection = val;
  else if (key == "Hfov") { try { h.hfov = std::stod(val); } catch (...) {} }
}

inline bool token(FILE *f, std::string &t, header *meta = nullptr)
{
  t.clear();
  int c;
  for (;;) {
    c = std::fgetc(f);
    if (c == EOF) return false;
    if (c == '#') {
      std::string line;
      while ((c = std::fgetc(f)) != EOF && c != '\n') line += char(c);
      if (meta) note_metadata(line, ':', *meta);
      continue;
    }
    if (c != ' ' && c != '\t' && c != '\n' && c != '\r') break;
  }
  while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r') { t += char(c); c = std::fgetc(f); }
  return true;   // exactly one whitespace character behind the token has been consumed
}

// Radiance header: "#?RADIANCE" (or "#?RGBE"), lines of NAME=value, an empty line, then "-Y h +X w"
inline bool read_rgbe_header(FILE *f, header &h, std::string &err)
{
  char line[512];
  bool format_ok = false;
  for (;;) {
    if (!std::fgets(line, sizeof line, f)) { err = "truncated Radiance header"; return false; }
    if (line[0] == '\n' || (line[0] == '\r' && line[1] == '\n')) break;
    if (!std::strncmp(line, "FORMAT=", 7)) format_ok = !std::strncmp(line + 7, "32-bit_rle_rgbe", 15);
    else note_metadata(line, '=', h);
  }
  if (!format_ok) { err = "Radiance picture without FORMAT=32-bit_rle_rgbe"; return false; }
  if (!std::fgets(line, sizeof line, f)) { err = "truncated Radiance header"; return false; }
  if (std::sscanf(line, "-Y %d +X %d", &h.height, &h.width) != 2) { err = "only the standard orientation -Y h +X w is read"; return false; }
  h.nchannels = 3; h.maxval = 0; h.rgbe = true;
  if (h.width <= 0 || h.height <= 0) { err = "unsupported image geometry"; return false; }
  h.data_offset = std::ftell(f);
  return true;
}

inline bool read_header(FILE *f, header &h, std::string &err)
{
  std::string magic, t;
  {
    const int c0 = std::fgetc(f), c1 = std::fgetc(f);
    if (c0 == '#' && c1 == '?') return read_rgbe_header(f, h, err);
    std::rewind(f);
  }
  if (!token(f, magic, &h)) { err = "empty file"; return false; }
  try {
    if (magic == "PF" || magic == "Pf" || magic == "PF4") {
      h.nchannels = magic == "Pf" ? 1 : magic == "PF" ? 3 : 4;
      if (!token(f, t, &h)) { err = "truncated header"; return false; }
      h.width = std::stoi(t);
      if (!token(f, t, &h)) { err = "truncated header"; return false; }
      h.height = std::stoi(t);
      if (!token(f, t, &h)) { err = "truncated header"; return false; }
      h.little_endian = std::stod(t) < 0.0;
      h.maxval = 0;
      h.bottom_up = true;
    } else if (magic == "P5" || magic == "P6") {
      h.nchannels = magic == "P5" ? 1 : 3;
      if (!token(f, t, &h)) { err = "truncated header"; return false; }
      h.width = std::stoi(t);
      if (!token(f, t, &h)) { err = "truncated header"; return false; }
      h.height = std::stoi(t);
      if (!token(f, t, &h)) { err = "truncated header"; return false; }
      h.maxval = std::stoi(t);
    } else if (magic == "P7") {
      for (;;) {
        if (!token(f, t, &h)) { err = "truncated header"; return false; }
        if (t == "ENDHDR") break;
        std::string v;
        if (!token(f, v, &h)) { err = "truncated header"; return false; }
        if (t == "WIDTH") h.width = std::stoi(v);
        else if (t == "HEIGHT") h.height = std::stoi(v);
        else if (t == "DEPTH") h.nchannels = std::stoi(v);
        else if (t == "MAXVAL") h.maxval = std::stoi(v);
      }
    } else {
      err = "not a PFM / PNM / PAM / Radiance file (magic '" + magic + "')";
      return false;
    }
  } catch (...) { err = "malformed header"; return false; }
  if (h.width <= 0 || h.height <= 0 || h.nchannels < 1 || h.nchannels > 4 || h.maxval < 0 || h.maxval > 65535) {
    err = "unsupported image geometry";
    return false;
  }
  // integer formats carry their MAXVAL (1 .. 65535); without one the samples would be read as raw floats
  if (magic[0] == 'P' && magic[1] >= '5' && magic[1] <= '7' && magic.size() == 2 && h.maxval < 1) {
    err = "PNM / PAM header without a MAXVAL between 1 and 65535";
    return false;
  }
  h.data_offset = std::ftell(f);
  return true;
}

inline bool probe_one(const std::string &name, header &h, std::string &err)
{
  FILE *f = std::fopen(name.c_str(), "rb");
  if (!f) { err = "cannot open " + name; return false; }
  const bool ok = read_header(f, h, err);
  std::fclose(f);
  if (!ok) err = name + ": " + err;
  return ok;
}

// cubeface_series (envutil_basic.h:267-340): six names from a format string with ONE percent sign
inline bool cubeface_names(const std::string &fmt, std::vector<std::string> &names)
{
  size_t count = 0;
  for (char c : fmt) count += c == '%';
  const size_t p = fmt.find("%s");
  if (count != 1 || p == std::string::npos) return false;
  static const char *const face[6] = { "left", "right", "top", "bottom", "front", "back" };
  names.clear();
  for (int i = 0; i < 6; i++) names.push_back(fmt.substr(0, p) + face[i] + fmt.substr(p + 2));
  return true;
}

// facet_base::get_image_metrics (envutil_basic.h:546-589): a name with a percent sign is a set
// of six cube faces and reports the metrics of the first
inline bool probe(const std::string &name, int &width, int &height, int &nchannels, std::string &err,
                  metadata *meta = nullptr)
{
  header h;
  std::vector<std::string> faces;
  const bool series = name.find('%') != std::string::npos;
  if (series && !cubeface_names(name, faces)) { err = "a format string needs exactly one %s: " + name; return false; }
  if (!probe_one(series ? faces[0] : name, h, err)) return false;
  width = h.width; height = h.height; nchannels = h.nchannels;
  if (meta) { meta->projection = h.projection; meta->hfov = h.hfov; }
  return true;
}

// rows top to bottom, interleaved, the file's own channel count
inline bool read_one(const std::string &name, header &h, float *dst, std::string &err)
{
  FILE *f = std::fopen(name.c_str(), "rb");
  if (!f) { err = "cannot open " + name; return false; }
  header g;
  if (!read_header(f, g, err)) { std::fclose(f); err = name + ": " + err; return false; }
  if (h.width && (g.width != h.width || g.height != h.height || g.nchannels != h.nchannels)) {
    std::fclose(f);
    err = name + ": size differs from the first image of the set";
    return false;
  }
  h = g;
  const size_t row = size_t(h.width) * h.nchannels;
  bool ok = true;
  if (h.rgbe) {
    std::vector<uint8_t> sl(size_t(h.width) * 4);
    for (int y = 0; y < h.height && ok; y++) {
      uint8_t b4[4];
      ok = std::fread(b4, 1, 4, f) == 4;
      if (!ok) break;
      if (b4[0] == 2 && b4[1] == 2 && !(b4[2] & 0x80) && ((int(b4[2]) << 8) | b4[3]) == h.width && h.width >= 8 && h.width < 32768) {
        // new run-length encoding: the four components one after the other
        for (int c = 0; c < 4 && ok; c++) {
          int x = 0;
          while (x < h.width && ok) {
            int n = std::fgetc(f);
            if (n == EOF) { ok = false; break; }
            if (n > 128) {
              n -= 128;
              const int v = std::fgetc(f);
              if (v == EOF || x + n > h.width) { ok = false; break; }
              for (int i = 0; i < n; i++) sl[size_t(x++) * 4 + c] = uint8_t(v);
            } else {
              if (n == 0 || x + n > h.width) { ok = false; break; }
              for (int i = 0; i < n; i++) {
                const int v = std::fgetc(f);
                if (v == EOF) { ok = false; break; }
                sl[size_t(x++) * 4 + c] = uint8_t(v);
              }
            }
          }
        }
      } else {
        // flat scanline (the first pixel is already read)
        std::memcpy(sl.data(), b4, 4);
        ok = std::fread(sl.data() + 4, 1, sl.size() - 4, f) == sl.size() - 4;
      }
      float *d = dst + size_t(y) * row;
      for (int x = 0; x < h.width; x++) {
        const uint8_t *p = sl.data() + size_t(x) * 4;
        if (p[3]) {
          const float fexp = std::ldexp(1.0f, int(p[3]) - (128 + 8));
          d[3 * x] = p[0] * fexp; d[3 * x + 1] = p[1] * fexp; d[3 * x + 2] = p[2] * fexp;
        } else d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = 0.0f;
      }
    }
  } else if (h.maxval == 0) {
    const uint16_t one = 1;
    const bool host_little = *reinterpret_cast<const uint8_t *>(&one) == 1;
    for (int y = 0; y < h.height && ok; y++) {
      float *d = dst + size_t(h.bottom_up ? h.height - 1 - y : y) * row;
      ok = std::fread(d, 4, row, f) == row;
      if (ok && host_little != h.little_endian)
        for (size_t i = 0; i < row; i++) {
          uint32_t u;
          std::memcpy(&u, d + i, 4);
          u = (u >> 24) | ((u >> 8) & 0xff00u) | ((u << 8) & 0xff0000u) | (u << 24);
          std::memcpy(d + i, &u, 4);
        }
    }
  } else {
    const int bytes = h.maxval > 255 ? 2 : 1;
    std::vector<uint8_t> buf(row * bytes);
    const float mv = float(h.maxval);
    for (int y = 0; y < h.height && ok; y++) {
      ok = std::fread(buf.data(), 1, buf.size(), f) == buf.size();
      float *d = dst + size_t(y) * row;
      if (bytes == 1) for (size_t i = 0; i < row; i++) d[i] = float(buf[i]) / mv;
      else for (size_t i = 0; i < row; i++) d[i] = float((unsigned(buf[2 * i]) << 8) | buf[2 * i + 1]) / mv;   // big endian
    }
  }
  std::fclose(f);
  if (!ok) err = name + ": truncated pixel data";
  return ok;
}

// a facet's pixels: a single image, or six cube faces stacked to the 1:6 image
// ---- colour spaces (envutil_basic.h:786-812, :950-977; OpenImageIO's built-in ColorConfig) -----------------
enum colour_class { CSP_UNKNOWN = 0, CSP_LINEAR = 1, CSP_SRGB = 2, CSP_REC709 = 3 };
inline colour_class classify_colour_space(const std::string &name)
{
  std::string n;
  for (char c : name) n += char(c >= 'A' && c <= 'Z' ? c - 'A' + 'a' : c);
  if (n == "linear" || n == "scene_linear" || n == "lin_srgb" || n == "lin_rec709" || n == "linear rec.709 (srgb)") return CSP_LINEAR;
  if (n == "srgb" || n == "srgb - texture" || n == "srgb_tx" || n == "srgb_texture" || n == "srgb encoded rec.709 (srgb)") return CSP_SRGB;
  if (n == "rec709" || n == "rec.709") return CSP_REC709;
  return CSP_UNKNOWN;
}
// what the format says about its samples (OpenImageIO: oiio:ColorSpace)
inline std::string file_colour_space(const std::string &name)
{
  const std::string e = lower_ext(name);
  return (e == "pfm" || e == "hdr" || e == "pic") ? "Linear" : "sRGB";
}
inline float to_linear(colour_class c, float v)
{
  if (c == CSP_SRGB) return v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);
  if (c == CSP_REC709) return v < 0.081f ? v / 4.5f : std::pow((v + 0.099f) / 1.099f, 1.0f / 0.45f);
  return v;
}
inline float from_linear(colour_class c, float v)
{
  if (c == CSP_SRGB) return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
  if (c == CSP_REC709) return v < 0.018f ? 4.5f * v : 1.099f * std::pow(v, 0.45f) - 0.099f;
  return v;
}
// npix pixels of nch interleaved channels from colour space `from` to `to`, in place; the alpha channel (the
// last of 2 or 4) stays. False with a message for a name this table does not know.
inline bool convert_colour(float *px, size_t npix, int nch, const std::string &from, const std::string &to, std::string &err)
{
  if (from == to) return true;
  const colour_class a = classify_colour_space(from), b = classify_colour_space(to);
  if (a == CSP_UNKNOWN || b == CSP_UNKNOWN) {
    err = "colour space '" + (a == CSP_UNKNOWN ? from : to) + "' is not known here (known: Linear / scene_linear / lin_srgb, sRGB, Rec709; OpenColorIO configurations are not read)";
    return false;
  }
  if (a == b) return true;
  const int ncol = (nch == 2 || nch == 4) ? nch - 1 : nch;
  for (size_t i = 0; i < npix; i++)
    for (int c = 0; c < ncol; c++) {
      float &v = px[i * size_t(nch) + c];
      v = from_linear(b, to_linear(a, v));
    }
  return true;
}

inline bool read_image(const std::string &name, std::vector<float> &pixels, int &width, int &height,
                       int &nchannels, std::string &err)
{
  std::vector<std::string> faces;
  if (name.find('%') != std::string::npos) {
    if (!cubeface_names(name, faces)) { err = "a format string needs exactly one %s: " + name; return false; }
  } else faces.push_back(name);
  header h;
  if (!probe_one(faces[0], h, err)) return false;
  const size_t per = size_t(h.width) * h.height * h.nchannels;
  try { pixels.resize(per * faces.size()); }
  catch (const std::exception &) { err = name + ": not enough host memory for " + std::to_string(h.width) + " x " + std::to_string(h.height) + " pixels"; return false; }
  for (size_t i = 0; i < faces.size(); i++)
    if (!read_one(faces[i], h, pixels.data() + i * per, err)) return false;
  width = h.width; height = h.height * int(faces.size()); nchannels = h.nchannels;
  return true;
}

inline bool write_one(const std::string &name, const float *src, int width, int height, int nch, std::string &err,
                      const metadata *meta = nullptr)
{
  const std::string ext = lower_ext(name);
  FILE *f = std::fopen(name.c_str(), "wb");
  if (!f) { err = "cannot create " + name; return false; }
  const size_t row = size_t(width) * nch;
  bool ok = true;
  if (ext == "pfm") {
    if (nch == 2) { std::fclose(f); err = name + ": PFM holds 1, 3 or 4 channels; use .pam for 2"; return false; }
    const uint16_t one = 1;
    const bool host_little = *reinterpret_cast<const uint8_t *>(&one) == 1;
    std::fprintf(f, "%s\n%d %d\n%s\n", nch == 1 ? "Pf" : nch == 3 ? "PF" : "PF4", width, height, host_little ? "-1.0" : "1.0");
    for (int y = height - 1; y >= 0 && ok; y--) ok = std::fwrite(src + size_t(y) * row, 4, row, f) == row;
  } else if (ext == "hdr" || ext == "pic") {
    if (nch != 3) { std::fclose(f); err = name + ": a Radiance picture holds 3 channels"; return false; }
    std::fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n");
    if (meta && !meta->projection.empty()) std::fprintf(f, "Projection=%s\nHfov=%.9g\n", meta->projection.c_str(), meta->hfov);
    std::fprintf(f, "\n-Y %d +X %d\n", height, width);
    std::vector<uint8_t> sl(size_t(width) * 4);
    for (int y = 0; y < height && ok; y++) {
      const float *sp = src + size_t(y) * row;
      for (int x = 0; x < width; x++) {
        const float r = sp[3 * x], g = sp[3 * x + 1], b = sp[3 * x + 2];
        float v = r > g ? r : g;
        v = b > v ? b : v;
        uint8_t *p = sl.data() + size_t(x) * 4;
        if (!(v >= 1e-32f)) { p[0] = p[1] = p[2] = p[3] = 0; continue; }     // also NaN and negatives
        int e;
        const float m = std::frexp(v, &e) * 256.0f / v;
        p[0] = uint8_t(r > 0.0f ? r * m : 0.0f); p[1] = uint8_t(g > 0.0f ? g * m : 0.0f);
        p[2] = uint8_t(b > 0.0f ? b * m : 0.0f); p[3] = uint8_t(e + 128);
      }
      ok = std::fwrite(sl.data(), 1, sl.size(), f) == sl.size();
    }
  } else if (ext == "pgm" || ext == "ppm" || ext == "pnm" || ext == "pam") {
    const bool pam = ext == "pam";
    if (!pam && nch != 1 && nch != 3) { std::fclose(f); err = name + ": PNM holds 1 or 3 channels; use .pam or .pfm"; return false; }
    const int maxval = pam ? 65535 : 255;
    if (pam) {
      static const char *const tt[5] = { "", "GRAYSCALE", "GRAYSCALE_ALPHA", "RGB", "RGB_ALPHA" };
      std::fprintf(f, "P7\n");
      if (meta && !meta->projection.empty()) std::fprintf(f, "# Projection: %s\n# Hfov: %.9g\n", meta->projection.c_str(), meta->hfov);
      std::fprintf(f, "WIDTH %d\nHEIGHT %d\nDEPTH %d\nMAXVAL %d\nTUPLTYPE %s\nENDHDR\n", width, height, nch, maxval, tt[nch]);
    } else {
      std::fprintf(f, "%s\n", nch == 1 ? "P5" : "P6");
      if (meta && !meta->projection.empty()) std::fprintf(f, "# Projection: %s\n# Hfov: %.9g\n", meta->projection.c_str(), meta->hfov);
      std::fprintf(f, "%d %d\n%d\n", width, height, maxval);
    }
    std::vector<uint8_t> buf(row * (pam ? 2 : 1));
    for (int y = 0; y < height && ok; y++) {
      const float *s = src + size_t(y) * row;
      for (size_t i = 0; i < row; i++) {
        float v = s[i];
        v = v < 0.0f ? 0.0f : v > 1.0f ? 1.0f : v;      // NaN fails both tests and is written as 0 below
        const unsigned q = v == v ? unsigned(v * float(maxval) + 0.5f) : 0u;
        if (pam) { buf[2 * i] = uint8_t(q >> 8); buf[2 * i + 1] = uint8_t(q & 255u); }
        else buf[i] = uint8_t(q);
      }
      ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    }
  } else {
    std::fclose(f);
    err = name + ": output formats are .pfm, .hdr, .pgm, .ppm, .pnm, .pam";
    return false;
  }
  ok = std::fclose(f) == 0 && ok;
  if (!ok) err = name + ": write failed";
  return ok;
}

// save_array (envutil_basic.h:710-815): a cubemap target whose output name is a format string
// goes to six face images, everything else to one file
inline bool write_image(const std::string &name, const float *src, int width, int height, int nch,
                        bool cubemap, std::string &err, const metadata *meta = nullptr)
{
  std::vector<std::string> faces;
  if (cubemap && name.find('%') != std::string::npos && cubeface_names(name, faces)) {
    if (height != 6 * width) { err = "a cubemap is 1:6"; return false; }
    // save_array writes the six faces as rectilinear images (envutil_basic.h:741-757)
    metadata face_meta;
    if (meta) { face_meta.projection = "rectilinear"; face_meta.hfov = meta->hfov; }
    for (int i = 0; i < 6; i++)
      if (!write_one(faces[size_t(i)], src + size_t(i) * width * width * nch, width, width, nch, err, meta ? &face_meta : nullptr)) return false;
    return true;
  }
  return write_one(name, src, width, height, nch, err, meta);
}

}  // namespace io
}  // namespace project
#endif
