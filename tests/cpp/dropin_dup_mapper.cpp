// tests/cpp/dropin_dup_mapper.cpp — a caller written the way the reference's sv2nl mapper is written
// (standalone/sv2nl/include/mapper.hpp:147-162 build_tree, :194-236 map_impl; source/mapper.cpp:50-55 DupMapper::
// check_condition; include/helper.hpp:16-63,84-91; source/writer.cpp:21-28), against NOTHING but the drop-in headers:
// binary/algorithm/interval_tree.hpp, binary/parser/vcf.hpp and sv2nl's vcf_info.hpp. One tree per chromosome from a
// filter | transform view of the SV file, one find_overlaps(record) per NL record, the post-filter, the duplicate-key
// rule, the line format. It shows that per-record caller code needs no change to run on the MI355X engine (slowly: one
// device round trip per record — INTEGRATION.md has the batched form); tests/test_vcf_facade_cpp.py compares its
// lines with the expected TSV of the authored pair fixture (the same lines the batched tool prints).
//
//   usage: dropin_dup_mapper <sv.vcf> <nl.vcf> [max distance = 1000000]
#include <algorithm>
#include <cstdio>
#include <iterator>
#include <ranges>
#include <set>
#include <string>
#include <vector>

#include "../../binary_amd/sv2nl/vcf_info.hpp"

using namespace sv2nl;

// pos <= svend, so that the record can be a node or a query of the tree
static Sv2nlVcfRecord ordered(const Sv2nlVcfRecord &r) {
  Sv2nlVcfRecord o = r;
  if (o.pos > o.info->svend) std::swap(o.pos, o.info->svend);
  return o;
}
static vcf::pos_t gap(vcf::pos_t a, vcf::pos_t b) { return a >= b ? a - b : b - a; }
// the SV duplication contains the NL tandem duplication and both ends lie within `d` of each other
static bool dup_condition(const Sv2nlVcfRecord &nl, const Sv2nlVcfRecord &sv, vcf::pos_t d) {
  const bool contains = sv.pos <= nl.pos && sv.info->svend >= nl.info->svend;
  return contains && gap(nl.pos, sv.pos) <= d && gap(nl.info->svend, sv.info->svend) <= d;
}
static std::string columns(const Sv2nlVcfRecord &r) {  // chrom, 1-based pos, end, svtype
  return r.chrom + "\t" + std::to_string(r.pos + 1) + "\t" + std::to_string(r.info->svend) + "\t" + r.info->svtype;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const std::string sv_path = argv[1], nl_path = argv[2];
  const vcf::pos_t dis = argc > 3 ? static_cast<vcf::pos_t>(std::stoul(argv[3])) : 1000000u;
  std::puts("chrom\tpos\tend\tsvtype\tchrom\tpos\tend\tsvtype");
  const auto chroms = Sv2nlVcfRanges(nl_path, "nls").chroms();
  for (auto const &chrom : chroms) {
    if (chrom.find('_') != std::string::npos) continue;  // primary contigs only
    const Sv2nlVcfRanges sv_records(sv_path, "delly");
    const Sv2nlVcfRanges nl_records(nl_path, "nls");
    // the chromosome's tree: its DUP records, ordered, in file order
    Sv2nlVcfIntervalTree tree{};
    auto of_chrom = sv_records | std::views::filter([&](auto const &r) { return r.chrom == chrom && r.info->svtype == "DUP"; })
                    | std::views::transform([](auto const &r) { return ordered(r); });
    tree.insert_node(of_chrom);
    // one query per NL record
    std::set<std::string> seen;  // an NL key is reported once
    auto queries = nl_records | std::views::filter([&](auto const &r) { return r.chrom == chrom && r.info->svtype == "TDUP"; });
    for (auto nl : queries) {
      const Sv2nlVcfRecord q = ordered(nl);
      const std::string key = nl.chrom + "-" + std::to_string(nl.pos) + "-" + std::to_string(nl.info->svend);
      if (seen.count(key)) continue;
      std::vector<Sv2nlVcfRecord> kept;
      for (auto const &hit : tree.find_overlaps(q))
        if (dup_condition(q, hit.record, dis)) kept.push_back(hit.record);
      if (kept.empty()) continue;
      seen.insert(key);
      for (auto const &sv : kept) std::printf("%s\t%s\n", columns(nl).c_str(), columns(sv).c_str());
    }
  }
  return 0;
}
