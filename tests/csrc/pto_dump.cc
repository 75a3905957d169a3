// Test helper: the product's PTO reader (include/eu_frontend.hpp, pto_script) on the lines of stdin, its line
// groups written in the canonical form of oracle/ref_zimt.cc's ref_pto_parse:
//   head<TAB>index<TAB>field=value<TAB>field=value...   (groups and fields in std::map order)
// so that tests/test_pto_pinned.py can compare the two parsers field by field.
#include <cstdio>
#include <iostream>
#include <string>
#include "eu_frontend.hpp"

int main()
{
  project::pto_script script;
  std::string line, err;
  while (std::getline(std::cin, line))
    if (!script.parse_pto_line(line, err)) { std::printf("ERROR\t%s\n", err.c_str()); return 2; }
  for (const auto &g : script.line_group) {
    int idx = 0;
    for (const auto &ln : g.second) {
      std::printf("%s\t%d", g.first.c_str(), idx++);
      for (const auto &f : ln.field_map) std::printf("\t%s=%s", f.first.c_str(), f.second.c_str());
      std::printf("\n");
    }
  }
  return 0;
}
