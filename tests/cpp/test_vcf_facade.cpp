// tests/cpp/test_vcf_facade.cpp — the reference's VCF -> tree glue tests (test/source/test_parser/test_vcf.cpp:47-187),
// re-expressed against the drop-in headers include/binary/parser/vcf.hpp and binary_amd/sv2nl/vcf_info.hpp. Same
// fixtures (tests/golden/vcf/debug.vcf.gz(.tbi), debug_uncom.vcf: the reference's test/data files, kept as data), same
// expected values; doctest is not in this image, so a small CHECK harness stands in. The tree cases run their overlap
// queries on the GPU through libbivx.so.
//
//   usage: test_vcf_facade <dir with the fixtures> [--no-gpu]
//          test_vcf_facade <dir> --overlaps <queries.txt>    prints, per query line "low high", the hits of the
//              6-record tree as "low-high:chrom:svtype" lists: "<reference pre-order> | <default order>"
#include <algorithm>
#include <binary/algorithm/all.hpp>
#include <binary/parser/all.hpp>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <ranges>
#include <sstream>
#include <string>

#include "../../binary_amd/sv2nl/vcf_info.hpp"

using namespace binary::parser::vcf;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                               \
  do {                                                            \
    ++g_checks;                                                   \
    if (!(cond)) {                                                \
      ++g_fail;                                                   \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
    }                                                             \
  } while (0)
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_THROWS(expr)                  \
  do {                                      \
    bool threw_ = false;                    \
    try {                                   \
      (void)(expr);                         \
    } catch (...) {                         \
      threw_ = true;                        \
    }                                       \
    CHECK(threw_);                          \
  } while (0)
#define CHECK_NOTHROW(expr)                 \
  do {                                      \
    bool threw_ = false;                    \
    try {                                   \
      (void)(expr);                         \
    } catch (...) {                         \
      threw_ = true;                        \
    }                                       \
    CHECK(!threw_);                         \
  } while (0)

static std::size_t run_query(const std::string &path) {  // test_vcf.cpp:18-36
  auto ranges = VcfRanges<VcfRecord>{path};
  CHECK_EQ(ranges.has_read_index(), false);
  std::size_t seen = 0;
  for (auto i = ranges.query("chr17", 7707250, 7798250); i != ranges.end(); i = ranges.iter_query_record()) {
    CHECK_EQ((*i).chrom, "chr17");
    ++seen;
  }
  for (auto i = ranges.query("chr10"); i != ranges.end(); i = ranges.iter_query_record()) {
    CHECK_EQ(i->chrom, "chr10");
    ++seen;
  }
  CHECK_EQ(ranges.has_read_index(), true);
  return seen;
}

static void parser_cases(const std::string &gz, const std::string &plain) {
  VcfRanges<VcfRecord> vcf_ranges(gz);
  {  // chroms (test_vcf.cpp:54-62): SURVEY §4: 455 contigs, 25 of them without '_'
    const auto chroms = vcf_ranges.chroms();
    CHECK_EQ(chroms.size(), 455u);
    CHECK_EQ(std::ranges::count_if(chroms, [](auto const &c) { return c.find('_') == std::string::npos; }), 25);
    CHECK_EQ(chroms.front(), "chr1");
  }
  {  // copy and move (:64-69)
    VcfRanges<VcfRecord> a(gz);
    VcfRanges<VcfRecord> copy(a);
    VcfRanges<VcfRecord> moved(std::move(a));
    CHECK_EQ(copy.file_path(), moved.file_path());
  }
  {  // queries need the index file (:71-86); five of the six records lie on chr10 / in the chr17 window
    CHECK_EQ(vcf_ranges.file_path(), gz);
    CHECK(vcf_ranges.has_index_file());
    CHECK_EQ(run_query(gz), 5u);
    VcfRanges<VcfRecord> r3(plain);
    CHECK_EQ(r3.has_index_file(), false);
    std::size_t n = 0;
    for (auto rec : r3) n += rec.chrom.empty() ? 0 : 1;
    CHECK_EQ(n, 6u);
    CHECK_THROWS(run_query(plain));
    CHECK_THROWS(VcfRanges<VcfRecord>(gz).query("chrNotThere"));
  }
  {  // the first record (:88-101): chr10 93567288 <TRA> ... SVTYPE=TRA;CHR2=chr17;SVEND=7705262 — pos is 0-based
    auto it = vcf_ranges.begin();
    CHECK_EQ(it->chrom, "chr10");
    CHECK_EQ(it->info->svtype, "TRA");
    CHECK_EQ(it->pos, 93567288u - 1u);
    CHECK_EQ(it->info->svend, 7705262u);
  }
  {  // a copied record is DETACHED, as the reference's clone() leaves it (vcf.hpp:246-250,297-303): the five value fields,
     // no file binding ("past the end"; next() throws); assigning to a bound record leaves its binding alone
    auto it = vcf_ranges.begin();
    VcfRecord copy(*it);
    CHECK_EQ(copy.chrom, "chr10");
    CHECK_EQ(copy.pos, 93567288u - 1u);
    CHECK_EQ(copy.info->svtype, "TRA");
    CHECK(copy == *it);
    CHECK(copy.eof_);
    CHECK_THROWS(copy.next());
    auto it2 = vcf_ranges.begin();
    VcfRecord bound(std::move(const_cast<VcfRecord &>(*it2.operator->())));  // a move keeps the binding
    CHECK(!bound.eof_);
    VcfRecord other;
    other.chrom = "chrZ";
    bound = other;
    CHECK_EQ(bound.chrom, "chrZ");
    CHECK(!bound.eof_);
    bound.next();  // still reads the file it was bound to: the second record
    CHECK_EQ(bound.chrom, "chr10");
    CHECK_EQ(bound.pos, 93567289u - 1u);
  }
  {  // ranges / views (:103-128)
    static_assert(std::forward_iterator<VcfRanges<VcfRecord>::iterator>);
    static_assert(std::ranges::input_range<VcfRanges<VcfRecord>>);
    CHECK_EQ(std::ranges::count_if(vcf_ranges, [](auto rec) { return rec.chrom == "chr10"; }), 3);
    std::size_t tdup = 0;
    for (auto const &rec : vcf_ranges | std::views::filter([](auto const &r) { return r.info->svtype == "TDUP"; })) {
      CHECK_EQ(rec.info->svtype, "TDUP");
      ++tdup;
    }
    CHECK_EQ(tdup, 3u);
  }
  {  // info factory (:129) and a lookup through it
    InfoFieldFactory<char, pos_t> f("SVTYPE", "SVEND");
    auto data = std::make_shared<details::DataImpl>(plain);
    CHECK(data->read());
    f.update(data, "");
    CHECK_EQ(std::get<0>(f.data_tuple), "TRA");
    CHECK_EQ(std::get<1>(f.data_tuple), 7705262u);
    CHECK_THROWS(get_info_field<pos_t>("SVTYPE", data->header.get(), data->record.get()));   // declared String
    CHECK_THROWS(get_info_field<char>("NOT_DECLARED", data->header.get(), data->record.get()));
  }
  {  // node from a record (:131-137) and from (end, start, record) (:139-147)
    auto begin = vcf_ranges.begin();
    auto node = VcfIntervalNode{*begin};
    CHECK_EQ(node.interval.record.chrom, "chr10");
    CHECK_EQ(node.interval.record.pos, 93567288u - 1u);
    CHECK_EQ(node.interval.record.info->svtype, "TRA");
    CHECK_EQ(node.interval.low, 93567287u);   // low = record.pos, high = record.info->svend (vcf.hpp:606-618):
    CHECK_EQ(node.interval.high, 7705262u);   // this record has POS > SVEND — a low > high node
    CHECK_EQ(node.key, 93567287u);
    CHECK_EQ(node.max, 7705262u);
    const auto start = begin->pos;
    const auto end = begin->info->svend;
    auto node2 = VcfIntervalNode{end, start, *begin};
    CHECK_EQ(node2.interval.record.chrom, "chr10");
    CHECK_EQ(node2.interval.record.pos, 93567288u - 1u);
    CHECK_EQ(node2.interval.record.info->svtype, "TRA");
    CHECK_EQ(node2.interval.low, 7705262u);
    CHECK_EQ(node2.interval.high, 93567287u);
    std::ostringstream os;
    os << node2.interval;
    CHECK_EQ(os.str(), "[VcfInterval: 7705262-93567287 [BaseVcfRecord chrom: chr10 pos: 93567287 rlen: 1 info: svtype: TRA "
                       "svend: 7705262]]");
  }
  {  // sv2nl's types (vcf_info.hpp:42-46): the same file read as source "nls"
    sv2nl::Sv2nlVcfRanges nl(gz, "nls");
    auto it = nl.begin();
    CHECK_EQ(it->info->chr2, "chr17");
    CHECK_EQ(it->info->svend, 7705262u);
    sv2nl::Sv2nlVcfIntervalNode n{*it};
    CHECK_EQ(n.interval.record.info->svtype, "TRA");
    static_assert(std::same_as<sv2nl::Sv2nlVcfIntervalTree::interval_type, sv2nl::Sv2nlVcfInterval>);
    // read as a delly file the first record has no END: the reader fails like the reference (vcf_info.cpp:39-41)
    sv2nl::Sv2nlVcfRanges wrong(gz, "delly");
    CHECK_THROWS(wrong.begin());
  }
}

static void tree_cases(const std::string &gz) {
  using namespace binary::algorithm::tree;
  VcfRanges<VcfRecord> vcf_ranges(gz);
  {  // one record (:149-156)
    auto t = IntervalTree<VcfIntervalNode>{};
    auto begin = vcf_ranges.begin();
    t.insert_node(*begin);
    CHECK_EQ(t.size(), 1u);
  }
  {  // the whole range (:158-164)
    auto t = IntervalTree<VcfIntervalNode>{};
    t.insert_node(vcf_ranges);
    CHECK_EQ(t.size(), 6u);
    CHECK_EQ(t.root()->interval.record.chrom.substr(0, 3), "chr");
    // a query built from a record, as sv2nl does (mapper.hpp:218): the INS record [93567288, 93567289] meets itself
    // only (the two TRA records start at 93567287 / 93567288 but END at 77052xx: low > high nodes hit nothing here)
    std::vector<VcfRecord> recs;
    for (auto r : vcf_ranges) recs.push_back(r);
    const auto hits = t.find_overlaps(recs[2]);
    CHECK_EQ(hits.size(), 1u);
    if (!hits.empty()) CHECK_EQ(hits[0].record.info->svtype, "INS");
    const auto one = t.find_overlap(recs[3]);   // chr14 TDUP [29927247, 29929790]
    CHECK(one.has_value());
    if (one) CHECK_EQ(one->record.chrom, "chr14");
  }
  {  // a filtered view (:166-174)
    auto t = IntervalTree<VcfIntervalNode>{};
    auto v = vcf_ranges | std::views::filter([](auto const &r) { return r.info->svtype == "TRA"; });
    t.insert_node(v);
    CHECK_EQ(t.size(), 2u);
  }
  {  // a range-for of moved records (:176-186)
    auto t = IntervalTree<VcfIntervalNode>{};
    for (auto r : vcf_ranges) t.insert_node(std::move(r));
    CHECK_EQ(t.size(), 6u);
  }
  {  // sv2nl's tree type over sv2nl's records
    sv2nl::Sv2nlVcfIntervalTree t{};
    t.insert_node(sv2nl::Sv2nlVcfRanges(gz, "nls"));
    CHECK_EQ(t.size(), 6u);
  }
}

static int overlaps_mode(const std::string &gz, const char *qfile) {
  using namespace binary::algorithm::tree;
  auto t = IntervalTree<VcfIntervalNode>{};
  t.insert_node(VcfRanges<VcfRecord>(gz));
  std::ifstream in(qfile);
  std::uint32_t lo = 0, hi = 0;
  while (in >> lo >> hi) {
    VcfInterval q{};
    q.low = lo;
    q.high = hi;
    for (auto order : {HitOrder::ReferencePreorder, HitOrder::Insertion}) {
      t.set_hit_order(order);
      for (auto const &h : t.find_overlaps(q))
        std::printf("%u-%u:%s:%s ", h.low, h.high, h.record.chrom.c_str(), h.record.info->svtype.c_str());
      if (order == HitOrder::ReferencePreorder) std::printf("| ");
    }
    std::printf("\n");
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  const std::string gz = dir + "/debug.vcf.gz", plain = dir + "/debug_uncom.vcf";
  if (argc >= 4 && std::strcmp(argv[2], "--overlaps") == 0) return overlaps_mode(gz, argv[3]);
  const bool gpu = !(argc >= 3 && std::strcmp(argv[2], "--no-gpu") == 0);
  try {
    parser_cases(gz, plain);
    if (gpu) tree_cases(gz);
  } catch (const std::exception &e) {
    std::printf("FAIL uncaught exception: %s\n", e.what());
    ++g_fail;
  }
  std::printf("%d checks, %d failed\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
