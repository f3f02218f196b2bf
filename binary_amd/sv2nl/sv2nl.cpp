// sv2nl.cpp — the `sv2nl` command line tool of ylab-hi/BINARY on top of the MI355X interval index.
//
// Same command line, defaults, validation, output files and line format as the reference
// (standalone/sv2nl/source/main.cpp:83-165, run() :46-81), same mapping rules
// (include/mapper.hpp:147-246, source/mapper.cpp:50-170, include/helper.hpp:16-91, source/writer.cpp:21-28).
// What changed is the engine: instead of one red-black interval tree per (mapper, chromosome) task, ALL SV records
// go into ONE device index as (chromosome id, start, end, svtype) columns (bivx_append_typed: DUP = 1, INV = 2,
// BND = 3; the index is partitioned by chromosome and type, so the svtype filter of mapper.hpp:153-156 costs
// nothing per candidate), each mapper asks all its NL records in one batch through the C ABI (include/bivx.h) with
// its type selected, and the mapper's check_condition runs on the device, fused into the overlap enumeration
// (bivx_filter). --index-per-mapper keeps round 1's three separate indexes for A/B. The reference's per-chromosome
// thread pool (-t) has nothing left to do; the flag is parsed and validated for compatibility.
//
// Output line ORDER: the reference's tasks append concurrently under a mutex (writer.hpp:33-38), so its order is
// not deterministic; here lines come out in NL-contig order, NL file order, SV file order.
#include <bivx.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "vcf_reader.hpp"

namespace sv2nl {

  constexpr std::string_view HEADER = "chrom\tpos\tend\tsvtype\tchrom\tpos\tend\tsvtype";  // mapper.hpp:29
  constexpr int NUM_THREADS = 4;                                                             // main.cpp:28
  constexpr const char *VERSION = "0.1.0-mi355x";

  // ---- helper.hpp -------------------------------------------------------------------------------------------
  inline auto is_tra(const Record &r) -> bool { return r.svtype == "TRA" || r.svtype == "BND"; }

  inline auto validate_record(const Record &record) -> Record {  // helper.hpp:52-63
    Record res = record;
    if (res.pos > res.svend) {
      std::swap(res.pos, res.svend);
      if (is_tra(record)) std::swap(res.chrom, res.chr2);
    }
    return res;
  }

  struct TwoChroms {
    std::string c1, c2;
    pos_t p1, p2;
    bool swapped;
  };
  inline auto two_chroms_with_pos(const Record &r) -> TwoChroms {  // helper.hpp:76-82
    if (r.chrom > r.chr2) return {r.chr2, r.chrom, r.svend, r.pos, true};
    return {r.chrom, r.chr2, r.pos, r.svend, false};
  }

  inline auto format_map_key(const Record &r) -> std::string {  // helper.hpp:84-91
    if (is_tra(r)) {
      auto t = two_chroms_with_pos(r);
      return t.c1 + "-" + t.c2 + "-" + std::to_string(t.p1) + "-" + std::to_string(t.p2);
    }
    return r.chrom + "-" + std::to_string(r.pos) + "-" + std::to_string(r.svend);
  }

  inline auto format_keys(const Record &r) -> std::string {  // writer.cpp:21-28
    std::string s = r.chrom;
    if (is_tra(r)) s += "," + r.chr2;
    s += "\t" + std::to_string(static_cast<std::uint64_t>(r.pos) + 1) + "\t" + std::to_string(r.svend) + "\t" + r.svtype;
    return s;
  }

  // ---- host restatement of the three check_condition predicates (used with --host-filter and for A/B) -------
  inline auto absdiff(pos_t a, pos_t b) -> pos_t { return a >= b ? a - b : b - a; }
  inline auto is_contained(const Record &t, const Record &s) -> bool { return t.pos <= s.pos && t.svend >= s.svend; }
  inline auto distance_less(const Record &a, const Record &b, pos_t d) -> bool {
    return absdiff(a.pos, b.pos) <= d && absdiff(a.svend, b.svend) <= d;
  }
  inline auto check_dup(const Record &nl, const Record &sv, pos_t d) -> bool {  // mapper.cpp:50-55
    return is_contained(sv, nl) && distance_less(nl, sv, d);
  }
  inline auto check_inv(const Record &nl, const Record &sv, pos_t d, bool use_strand) -> bool {  // mapper.cpp:57-79
    if (is_contained(sv, nl) || is_contained(nl, sv) || !distance_less(nl, sv, d)) return false;
    if (!use_strand) return true;
    if (nl.pos <= sv.pos) return nl.strand1 && !nl.strand2;
    return !nl.strand1 && nl.strand2;
  }
  inline auto check_tra(const Record &nl, const Record &sv, pos_t d) -> bool {  // mapper.cpp:144-156
    auto n = two_chroms_with_pos(nl), s = two_chroms_with_pos(sv);
    return n.c1 == s.c1 && n.c2 == s.c2 && absdiff(n.p1, s.p1) <= d && absdiff(n.p2, s.p2) <= d;
  }

  struct Options {
    std::string sv_path, nl_path, output{"output.tsv"};
    std::uint32_t dis{1000000};
    int threads{NUM_THREADS};
    bool short_reads{false}, merge{false}, debug{false}, host_filter{false}, index_per_mapper{false};
    int device{0};
    std::vector<int> devices;  // --devices: shard the index by chromosome over these GPUs (bivx_create_sharded)
  };

  enum class Kind { Dup, Inv, Tra };
  inline auto type_code(Kind k) -> std::uint8_t { return k == Kind::Dup ? 1 : k == Kind::Inv ? 2 : 3; }

  // The tree side of all three mappers in ONE device index. DUP / INV records enter validated under their chromosome
  // (mapper.hpp:153-156); BND records enter as read, all under chromosome 0 (TraMapper builds one tree of every
  // chromosome's records, un-validated, mapper.cpp:158-170); the svtype column keeps the three apart.
  struct SharedIndex {
    bivx_index *ix = nullptr;
    std::vector<Record> items;             // stored record of each interval id
    std::vector<std::uint32_t> iaux;       // TRA: ordered chromosome pair << 1 | swapped (0 for DUP / INV)
    std::unordered_map<std::string, std::uint32_t> chrom_id;
    std::map<std::pair<std::string, std::string>, std::uint32_t> pair_id;
    ~SharedIndex() {
      if (ix) bivx_destroy(ix);
    }
  };

  inline void die_bivx(const char *what);

  inline void build_shared(SharedIndex &sh, const std::vector<Record> &sv, const Options &opt) {
    std::vector<std::uint32_t> ic, ilo, ihi;
    std::vector<std::uint8_t> ity;
    auto id_of = [&](const std::string &c) {
      auto it = sh.chrom_id.find(c);
      if (it != sh.chrom_id.end()) return it->second;
      auto v = static_cast<std::uint32_t>(sh.chrom_id.size());
      sh.chrom_id.emplace(c, v);
      return v;
    };
    for (auto const &r : sv) {
      Kind k;
      if (r.svtype == "DUP") k = Kind::Dup;
      else if (r.svtype == "INV") k = Kind::Inv;
      else if (r.svtype == "BND") k = Kind::Tra;
      else continue;
      Record s = k == Kind::Tra ? r : validate_record(r);
      ic.push_back(k == Kind::Tra ? 0u : id_of(s.chrom));
      ilo.push_back(s.pos);
      ihi.push_back(s.svend);
      ity.push_back(type_code(k));
      if (k == Kind::Tra) {
        auto t = two_chroms_with_pos(s);
        auto key = std::make_pair(t.c1, t.c2);
        auto it = sh.pair_id.find(key);
        if (it == sh.pair_id.end()) it = sh.pair_id.emplace(key, static_cast<std::uint32_t>(sh.pair_id.size())).first;
        sh.iaux.push_back((it->second << 1) | (t.swapped ? 1u : 0u));
      } else {
        sh.iaux.push_back(0u);
      }
      sh.items.push_back(std::move(s));
    }
    if (sh.items.empty()) return;
    if (opt.devices.size() > 1) {
      if (bivx_create_sharded(&sh.ix, opt.devices.data(), static_cast<int>(opt.devices.size())) != 0)
        die_bivx("bivx_create_sharded");
    } else if (bivx_create(&sh.ix, opt.device) != 0) {
      die_bivx("bivx_create");
    }
    if (bivx_append_typed(sh.ix, ic.data(), ilo.data(), ihi.data(), ity.data(), sh.items.size()) != 0)
      die_bivx("bivx_append_typed");
    if (bivx_build(sh.ix) != 0) die_bivx("bivx_build");
  }

  inline void die_bivx(const char *what) {
    std::fprintf(stderr, "[sv2nl] %s: %s\n", what, bivx_last_error());
    std::exit(2);
  }

  // One mapper = one batch of queries against its type's partition of the shared index (or, with
  // --index-per-mapper, against an index of its own).
  class Mapper {
  public:
    Mapper(Kind kind, std::string nl_type, std::string sv_type, const Options &opt)
        : kind_(kind), nl_type_(std::move(nl_type)), sv_type_(std::move(sv_type)), opt_(opt) {}

    // sv: all SV records (file order). nl: NL records before the first unreadable one. chroms: NL header contigs.
    auto map(const std::vector<Record> &sv, const std::vector<Record> &nl, const std::vector<std::string> &chroms,
             const SharedIndex *shared) -> std::vector<std::string> {
      if (shared != nullptr) return map_shared(*shared, nl, chroms);
      std::unordered_map<std::string, std::uint32_t> chrom_id;
      auto id_of = [&](const std::string &c) {
        auto it = chrom_id.find(c);
        if (it != chrom_id.end()) return it->second;
        auto v = static_cast<std::uint32_t>(chrom_id.size());
        chrom_id.emplace(c, v);
        return v;
      };
      std::map<std::pair<std::string, std::string>, std::uint32_t> pair_id;  // ordered chromosome pair -> id (TRA)
      auto pair_of = [&](const TwoChroms &t) {
        auto key = std::make_pair(t.c1, t.c2);
        auto it = pair_id.find(key);
        if (it != pair_id.end()) return it->second;
        auto v = static_cast<std::uint32_t>(pair_id.size());
        pair_id.emplace(key, v);
        return v;
      };

      // tree side: DUP/INV keep (chrom, validated record) (mapper.hpp:153-156); TRA keeps raw BND records of every
      // chromosome in ONE tree (mapper.cpp:158-170)
      std::vector<Record> items;
      std::vector<std::uint32_t> ic, ilo, ihi, iaux;
      for (auto const &r : sv) {
        if (r.svtype != sv_type_) continue;
        Record s = kind_ == Kind::Tra ? r : validate_record(r);
        ic.push_back(kind_ == Kind::Tra ? 0u : id_of(s.chrom));
        ilo.push_back(s.pos);
        ihi.push_back(s.svend);
        if (kind_ == Kind::Tra) {
          auto t = two_chroms_with_pos(s);
          iaux.push_back((pair_of(t) << 1) | (t.swapped ? 1u : 0u));
        } else {
          iaux.push_back(0u);
        }
        items.push_back(std::move(s));
      }

      // query side: per primary chromosome of the NL header (names with '_' skipped, mapper.hpp:241-243), NL
      // records of nl_type in file order; the first record of each key wins once it has a surviving hit
      std::vector<const Record *> qrec;
      std::vector<Record> qval;
      std::vector<std::uint32_t> qc, qlo, qhi, qaux;
      for (auto const &chrom : chroms) {
        if (chrom.find('_') != std::string::npos) continue;
        for (auto const &r : nl) {
          if (r.chrom != chrom || r.svtype != nl_type_) continue;
          Record q = validate_record(r);
          std::uint32_t aux = 0, c = 0;
          if (kind_ == Kind::Tra) {
            auto t = two_chroms_with_pos(q);
            auto key = std::make_pair(t.c1, t.c2);
            auto it = pair_id.find(key);
            aux = ((it == pair_id.end() ? 0x7FFFFFFFu : it->second) << 1) | (t.swapped ? 1u : 0u);
          } else {
            auto it = chrom_id.find(q.chrom);
            c = it == chrom_id.end() ? 0xFFFFFFFFu : it->second;  // no SV record on this chromosome: empty tree
            if (kind_ == Kind::Inv) aux = (q.strand1 ? 1u : 0u) | (q.strand2 ? 2u : 0u);
          }
          qrec.push_back(&r);
          qc.push_back(c);
          qlo.push_back(q.pos);
          qhi.push_back(q.svend);
          qaux.push_back(aux);
          qval.push_back(std::move(q));
        }
      }

      std::vector<std::string> lines;
      if (qrec.empty()) return lines;
      std::vector<std::uint64_t> off(qrec.size() + 1, 0);
      std::vector<std::uint32_t> hits;
      if (!items.empty()) {
        bivx_index *ix = nullptr;
        if (bivx_create(&ix, opt_.device) != 0) die_bivx("bivx_create");
        if (bivx_append(ix, ic.data(), ilo.data(), ihi.data(), items.size()) != 0) die_bivx("bivx_append");
        if (bivx_build(ix) != 0) die_bivx("bivx_build");
        bivx_filter flt{};
        flt.kind = opt_.host_filter ? BIVX_FILTER_NONE
                   : kind_ == Kind::Dup ? BIVX_FILTER_SV2NL_DUP
                   : kind_ == Kind::Inv ? BIVX_FILTER_SV2NL_INV
                                        : BIVX_FILTER_SV2NL_TRA;
        flt.max_dist = opt_.dis;
        flt.use_strand = opt_.short_reads ? 0u : 1u;
        flt.query_aux = qaux.data();
        flt.interval_aux = iaux.data();
        const std::uint32_t *qcp = kind_ == Kind::Tra ? nullptr : qc.data();
        if (bivx_count_f(ix, qcp, qlo.data(), qhi.data(), qrec.size(), &flt, off.data()) != 0) die_bivx("bivx_count_f");
        hits.resize(static_cast<std::size_t>(off.back()));
        if (bivx_fill_f(ix, qcp, qlo.data(), qhi.data(), qrec.size(), &flt, off.data(), hits.data(), 1) != 0)
          die_bivx("bivx_fill_f");
        bivx_destroy(ix);
      }

      return emit(qrec, qval, off, hits, items);
    }

    // the same against the shared typed index: only the query side is assembled here
    auto map_shared(const SharedIndex &sh, const std::vector<Record> &nl, const std::vector<std::string> &chroms)
        -> std::vector<std::string> {
      std::vector<const Record *> qrec;
      std::vector<Record> qval;
      std::vector<std::uint32_t> qc, qlo, qhi, qaux;
      for (auto const &chrom : chroms) {
        if (chrom.find('_') != std::string::npos) continue;
        for (auto const &r : nl) {
          if (r.chrom != chrom || r.svtype != nl_type_) continue;
          Record q = validate_record(r);
          std::uint32_t aux = 0, c = 0;
          if (kind_ == Kind::Tra) {
            auto t = two_chroms_with_pos(q);
            auto it = sh.pair_id.find(std::make_pair(t.c1, t.c2));
            aux = ((it == sh.pair_id.end() ? 0x7FFFFFFFu : it->second) << 1) | (t.swapped ? 1u : 0u);
          } else {
            auto it = sh.chrom_id.find(q.chrom);
            c = it == sh.chrom_id.end() ? 0xFFFFFFFFu : it->second;  // no SV record on this chromosome: empty tree
            if (kind_ == Kind::Inv) aux = (q.strand1 ? 1u : 0u) | (q.strand2 ? 2u : 0u);
          }
          qrec.push_back(&r);
          qc.push_back(c);
          qlo.push_back(q.pos);
          qhi.push_back(q.svend);
          qaux.push_back(aux);
          qval.push_back(std::move(q));
        }
      }
      std::vector<std::string> lines;
      if (qrec.empty()) return lines;
      std::vector<std::uint64_t> off(qrec.size() + 1, 0);
      std::vector<std::uint32_t> hits;
      if (sh.ix != nullptr) {
        bivx_filter flt{};
        flt.kind = opt_.host_filter ? BIVX_FILTER_NONE
                   : kind_ == Kind::Dup ? BIVX_FILTER_SV2NL_DUP
                   : kind_ == Kind::Inv ? BIVX_FILTER_SV2NL_INV
                                        : BIVX_FILTER_SV2NL_TRA;
        flt.max_dist = opt_.dis;
        flt.use_strand = opt_.short_reads ? 0u : 1u;
        flt.svtype = type_code(kind_);  // this mapper's partition of the index (mapper.hpp:153-156)
        flt.query_aux = qaux.data();
        flt.interval_aux = sh.iaux.data();
        if (bivx_count_f(sh.ix, qc.data(), qlo.data(), qhi.data(), qrec.size(), &flt, off.data()) != 0)
          die_bivx("bivx_count_f");
        hits.resize(static_cast<std::size_t>(off.back()));
        if (bivx_fill_f(sh.ix, qc.data(), qlo.data(), qhi.data(), qrec.size(), &flt, off.data(), hits.data(), 1) != 0)
          die_bivx("bivx_fill_f");
      }
      return emit(qrec, qval, off, hits, sh.items);
    }

    // hits -> output lines: the first NL record of each key with a surviving hit wins (mapper.hpp:213,228-232)
    auto emit(const std::vector<const Record *> &qrec, const std::vector<Record> &qval,
              const std::vector<std::uint64_t> &off, const std::vector<std::uint32_t> &hits,
              const std::vector<Record> &items) const -> std::vector<std::string> {
      std::vector<std::string> lines;
      std::unordered_set<std::string> cache;  // ThreadSafeMap of the reference: keys that already produced output
      for (std::size_t i = 0; i < qrec.size(); ++i) {
        const Record &orig = *qrec[i];
        std::string key = format_map_key(orig);
        if (cache.count(key)) continue;  // mapper.hpp:213
        std::string key_line;
        bool any = false;
        for (std::uint64_t k = off[i]; k < off[i + 1]; ++k) {
          const Record &s = items[hits[k]];
          if (opt_.host_filter) {
            const bool ok = kind_ == Kind::Dup   ? check_dup(qval[i], s, opt_.dis)
                            : kind_ == Kind::Inv ? check_inv(qval[i], s, opt_.dis, !opt_.short_reads)
                                                 : check_tra(qval[i], s, opt_.dis);
            if (!ok) continue;
          }
          if (!any) key_line = format_keys(orig);  // the NL record is printed as read, not validated (mapper.hpp:215)
          any = true;
          lines.push_back(key_line + "\t" + format_keys(s));
        }
        if (any) cache.insert(std::move(key));  // only keys with output are cached (mapper.hpp:228-232)
      }
      return lines;
    }

  private:
    Kind kind_;
    std::string nl_type_, sv_type_;
    const Options &opt_;
  };

  inline void write_part(const std::string &path, const std::vector<std::string> &lines) {  // writer.hpp:22-39
    std::ofstream out(path);
    out << HEADER << '\n';
    for (auto const &l : lines) out << l << '\n';
  }

  inline void merge_files(const std::vector<std::string> &files, const std::string &output) {  // utils.hpp:43-68
    std::ofstream out(output);
    out << HEADER << '\n';
    for (auto const &f : files) {
      if (!std::filesystem::exists(f)) continue;
      std::ifstream in(f);
      std::string first;
      std::getline(in, first);  // skip the part's header
      out << in.rdbuf();
      in.close();
      std::filesystem::remove(f);
    }
  }

  inline auto usage() -> std::string {
    return "Map structural Variation to Non-Linear Transcription\n"
           "Usage:\n  sv2nl [OPTION...] [sv non-linear]\n\n"
           "      --sv arg          The file path of segment information from delly\n"
           "      --non-linear arg  The file path of non-linear information from scannls\n"
           "      --dis arg         The distance threshold for trans mapper (default: 1000000)\n"
           "  -o, --output arg      The file path of output (default: output.tsv)\n"
           "  -t, --thread arg      The number of thread program use (default: 4)\n"
           "  -s, --short           If running in short read and do not use strand\n"
           "  -m, --merge           If provided only merge outputs into one file\n"
           "  -d, --debug           Print debug info\n"
           "  -h, --help            Print help\n"
           "  -v, --version         Print the current version number\n"
           "      --device arg      HIP device ordinal (default: 0 or $BIVX_DEVICE)\n"
           "      --devices arg     Comma-separated HIP device ordinals: one index sharded by chromosome over them\n"
           "      --host-filter     Evaluate check_condition on the host instead of on the device\n"
           "      --index-per-mapper  One device index per mapper instead of one typed index for all three\n";
  }

}  // namespace sv2nl

int main(int argc, char **argv) {
  using namespace sv2nl;
  Options opt;
  if (const char *e = std::getenv("BIVX_DEVICE")) opt.device = std::atoi(e);
  std::vector<std::string> positional;
  bool have_sv = false, have_nl = false;
  auto need = [&](int &i, const char *name) -> std::string {
    if (i + 1 >= argc) {
      std::fprintf(stderr, "[sv2nl] error parsing options: Option '%s' is missing an argument\n%s\n", name, usage().c_str());
      std::exit(1);
    }
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], val;
    auto eq = a.find('=');
    bool has_val = a.rfind("--", 0) == 0 && eq != std::string::npos;
    if (has_val) {
      val = a.substr(eq + 1);
      a = a.substr(0, eq);
    }
    auto value = [&](const char *name) { return has_val ? val : need(i, name); };
    if (a == "-h" || a == "--help") {
      std::cout << usage() << "\n";
      return 0;
    } else if (a == "-v" || a == "--version") {
      std::fprintf(stderr, "[sv2nl] version %s\n", VERSION);
      return 0;
    } else if (a == "--sv") {
      opt.sv_path = value("sv");
      have_sv = true;
    } else if (a == "--non-linear") {
      opt.nl_path = value("non-linear");
      have_nl = true;
    } else if (a == "--dis") {
      opt.dis = static_cast<std::uint32_t>(std::stoul(value("dis")));
    } else if (a == "-o" || a == "--output") {
      opt.output = value("output");
    } else if (a == "-t" || a == "--thread") {
      opt.threads = std::stoi(value("thread"));
    } else if (a == "-s" || a == "--short") {
      opt.short_reads = true;
    } else if (a == "-m" || a == "--merge") {
      opt.merge = true;
    } else if (a == "-d" || a == "--debug") {
      opt.debug = true;
    } else if (a == "--device") {
      opt.device = std::stoi(value("device"));
    } else if (a == "--devices") {
      std::string list = value("devices");
      for (std::size_t b = 0; b <= list.size();) {
        const std::size_t e = list.find(',', b);
        const std::string tok = list.substr(b, e == std::string::npos ? std::string::npos : e - b);
        if (!tok.empty()) opt.devices.push_back(std::stoi(tok));
        if (e == std::string::npos) break;
        b = e + 1;
      }
      if (!opt.devices.empty()) opt.device = opt.devices[0];
    } else if (a == "--host-filter") {
      opt.host_filter = true;
    } else if (a == "--index-per-mapper") {
      opt.index_per_mapper = true;
    } else if (a.size() > 1 && a[0] == '-') {
      std::fprintf(stderr, "[sv2nl] error parsing options: Option '%s' does not exist\n%s\n", a.c_str(), usage().c_str());
      return 1;
    } else {
      positional.push_back(argv[i]);
    }
  }
  for (auto &p : positional) {  // positional order: sv, non-linear (main.cpp:101)
    if (!have_sv) {
      opt.sv_path = p;
      have_sv = true;
    } else if (!have_nl) {
      opt.nl_path = p;
      have_nl = true;
    }
  }
  if (!have_sv || !have_nl) {  // cxxopts::option_has_no_value_exception -> help, exit 1 (main.cpp:158-162)
    std::fprintf(stderr, "[sv2nl] error parsing options: Option '%s' has no value\n", have_sv ? "non-linear" : "sv");
    std::cout << usage() << "\n";
    return 1;
  }
  for (auto const &p : {opt.sv_path, opt.nl_path})
    if (!std::filesystem::exists(p)) {  // check_file_path -> exit(1) (main.cpp:128-130)
      std::fprintf(stderr, "[sv2nl] %s does not exist\n", p.c_str());
      return 1;
    }
  if (opt.threads < 0 || opt.threads > static_cast<int>(std::thread::hardware_concurrency())) {  // main.cpp:132-137
    std::fprintf(stderr, "[sv2nl] The number of threads %d is invalid, default value %d will be used\n", opt.threads, NUM_THREADS);
    opt.threads = NUM_THREADS;
  }
  std::fprintf(stderr, "[sv2nl] non-linear file path: %s\n[sv2nl] struct variation file path: %s\n[sv2nl] distance threshold: %u bp\n"
                       "[sv2nl] use strand: %s\n",
               opt.nl_path.c_str(), opt.sv_path.c_str(), opt.dis, opt.short_reads ? "false" : "true");

  const std::vector<std::string> parts{opt.output + ".dup", opt.output + ".inv", opt.output + ".tra"};
  try {
    // the two files are read side by side, and the three mappers run side by side (-t 1 keeps everything on the
    // calling thread); the reference spreads its per-chromosome tasks over a pool of -t threads (mapper.hpp:238-246)
    const bool threaded = opt.threads != 1;
    VcfFile nl_file(opt.nl_path, "nls");
    VcfFile sv_file(opt.sv_path, "delly");
    std::string nl_err, sv_err;
    std::vector<Record> nl, sv;
    {
      auto read_sv = [&] { sv = sv_file.read_all(&sv_err); };
      std::future<void> f;
      if (threaded) f = std::async(std::launch::async, read_sv);
      nl = nl_file.read_all(&nl_err);
      if (threaded) f.get(); else read_sv();
    }
    const auto chroms = nl_file.chroms();
    if (!nl_err.empty()) std::fprintf(stderr, "[sv2nl] non-linear file: %s (records after it are not mapped)\n", nl_err.c_str());
    if (!sv_err.empty()) {
      // reference: every chromosome task dies reading the SV file (exception swallowed, header-only .dup/.inv) and
      // TraMapper::build_sv_tree throws on the main thread (mapper.cpp:130, uncaught). Here: same files, exit 1.
      std::fprintf(stderr, "[sv2nl] struct variation file: %s\n", sv_err.c_str());
      for (auto const &p : parts) write_part(p, {});
      return 1;
    }
    SharedIndex shared;
    if (!opt.index_per_mapper) build_shared(shared, sv, opt);
    const SharedIndex *sh = opt.index_per_mapper ? nullptr : &shared;
    auto run = [&](int part, Kind kind, const char *nl_type, const char *sv_type) {
      write_part(parts[part], Mapper(kind, nl_type, sv_type, opt).map(sv, nl, chroms, sh));
    };
    if (threaded) {
      auto dup = std::async(std::launch::async, run, 0, Kind::Dup, "TDUP", "DUP");  // main.cpp:52-58
      auto inv = std::async(std::launch::async, run, 1, Kind::Inv, "INV", "INV");   // main.cpp:60-67
      run(2, Kind::Tra, "TRA", "BND");                                              // main.cpp:69-75
      dup.get();
      inv.get();
    } else {
      run(0, Kind::Dup, "TDUP", "DUP");
      run(1, Kind::Inv, "INV", "INV");
      run(2, Kind::Tra, "TRA", "BND");
    }
  } catch (const binary::VcfReaderError &e) {
    std::fprintf(stderr, "[sv2nl] %s\n", e.what());
    return 1;
  }
  if (opt.merge) {
    merge_files(parts, opt.output);
    std::fprintf(stderr, "[sv2nl] result file path: %s\n", opt.output.c_str());
  } else {
    std::fprintf(stderr, "[sv2nl] result file path: %s[.dup|.inv|.tra]\n", opt.output.c_str());
  }
  return 0;
}
