// TEST CODE: reads an image with include/eu_image_io.hpp and writes it again (format by extension).
//   io_demo IN OUT        prints "w h nchannels"
#include <cstdio>
#include <string>
#include <vector>
#include "eu_image_io.hpp"

int main(int argc, char **argv)
{
  if (argc != 3) return 2;
  std::vector<float> px;
  int w = 0, h = 0, n = 0;
  std::string err;
  if (!project::io::read_image(argv[1], px, w, h, n, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
  std::printf("%d %d %d\n", w, h, n);
  if (!project::io::write_image(argv[2], px.data(), w, h, n, false, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
  return 0;
}
