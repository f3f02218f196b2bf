// vcf_dump.cpp — prints what vcf_reader.hpp extracts from a VCF (test helper for the reader):
//   C <contig>                       one line per header contig
//   R chrom pos svtype svend chr2 strand1 strand2
//   E <message>                      the first unreadable record
#include <cstdio>

#include "vcf_reader.hpp"

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: vcf_dump <file.vcf[.gz]> <nls|delly>\n");
    return 2;
  }
  try {
    sv2nl::VcfFile f(argv[1], argv[2]);
    for (auto const &c : f.chroms()) std::printf("C %s\n", c.c_str());
    std::string err;
    for (auto const &r : f.read_all(&err))
      std::printf("R %s %u %s %u %s %d %d\n", r.chrom.c_str(), r.pos, r.svtype.c_str(), r.svend,
                  r.chr2.empty() ? "." : r.chr2.c_str(), r.strand1 ? 1 : 0, r.strand2 ? 1 : 0);
    if (!err.empty()) std::printf("E %s\n", err.c_str());
  } catch (const binary::VcfReaderError &e) {
    std::printf("E %s\n", e.what());
  }
  return 0;
}
