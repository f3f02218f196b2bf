// vcf_reader.hpp — minimal VCF record reader for sv2nl (plain text, gzip and BGZF through zlib; no htslib).
//
// Replaces, for the five fields sv2nl needs, the reference's VcfParser path
//   VcfRanges / BaseVcfRecord::next/update      library/include/binary/parser/vcf.hpp:263-275,305-310
//   header contig list (chroms())               vcf.hpp:577-589
//   typed INFO lookup (bcf_get_info_values)     vcf.hpp:119-130
//   Sv2nlInfoField::update                      standalone/sv2nl/source/vcf_info.cpp:9-43
// The parsing arithmetic of the reference lives in htslib 1.15.1 (absent here); this file restates the
// behaviour the reference relies on: POS is returned 0-based, INFO values are typed by the header's ##INFO
// lines, and asking for a tag that is undeclared, of another type, or missing from the record is an error
// ("Failed to get info <TAG>", binary::VcfReaderError).
#ifndef BINARY_AMD_SV2NL_VCF_READER_HPP_
#define BINARY_AMD_SV2NL_VCF_READER_HPP_

#include <zlib.h>

#include <charconv>
#include <cstdint>
#include <exception>
#include <string>
#include <string_view>
#include <unordered_map>
#include <utility>
#include <vector>

namespace binary {
  class VcfReaderError : public std::exception {  // reference: library/include/binary/exception.hpp:13-20
  public:
    explicit VcfReaderError(std::string m) : msg_(std::move(m)) {}
    [[nodiscard]] auto what() const noexcept -> const char * override { return msg_.c_str(); }

  private:
    std::string msg_;
  };
}  // namespace binary

namespace sv2nl {

  using pos_t = std::uint32_t;

  /// What sv2nl keeps of one VCF line (reference Sv2nlVcfRecord = BaseVcfRecord<Sv2nlInfoField>).
  struct Record {
    std::string chrom;
    pos_t pos{0};   // 0-based (htslib bcf1_t::pos)
    pos_t rlen{0};
    std::string svtype;
    pos_t svend{0};  // as written in the file (1-based coordinate, not shifted)
    std::string chr2;
    bool strand1{true};  // true = '+'
    bool strand2{true};
  };

  enum class InfoType { Flag, Integer, Float, String, Other };

  class VcfFile {
  public:
    /// source: "nls" (ScanNLS: end coordinate in SVEND) or "delly" (END); BND records use POS2 either way.
    VcfFile(const std::string &path, std::string source) : path_(path), source_(std::move(source)) {
      fp_ = gzopen(path.c_str(), "rb");
      if (fp_ == nullptr) throw binary::VcfReaderError("Failed to open " + path);
      gzbuffer(fp_, 1 << 18);
      read_header();
    }
    VcfFile(const VcfFile &) = delete;
    auto operator=(const VcfFile &) -> VcfFile & = delete;
    ~VcfFile() {
      if (fp_ != nullptr) gzclose(fp_);
    }

    /// Header contigs in header order (what VcfRanges::chroms() returns).
    [[nodiscard]] auto chroms() const -> const std::vector<std::string> & { return contigs_; }
    [[nodiscard]] auto file_path() const -> const std::string & { return path_; }

    /// Reads the next record; false at end of file. Throws binary::VcfReaderError like the reference.
    auto next(Record &out) -> bool {
      std::string line;
      while (true) {
        if (!pending_.empty()) {
          line.swap(pending_);
          pending_.clear();
        } else if (!getline(line)) {
          return false;
        }
        if (line.empty()) continue;
        break;
      }
      parse_line(line, out);
      return true;
    }

    /// Convenience: all records up to the first error. Returns the error text ("" if none) in *error.
    auto read_all(std::string *error) -> std::vector<Record> {
      std::vector<Record> v;
      if (error) error->clear();
      try {
        Record r;
        while (next(r)) v.push_back(r);
      } catch (const binary::VcfReaderError &e) {
        if (error) *error = e.what();
      }
      return v;
    }

  private:
    auto getline(std::string &line) -> bool {
      line.clear();
      char buf[1 << 16];
      while (true) {
        if (gzgets(fp_, buf, sizeof(buf)) == nullptr) return !line.empty();
        line.append(buf);
        if (!line.empty() && line.back() == '\n') {
          line.pop_back();
          if (!line.empty() && line.back() == '\r') line.pop_back();
          return true;
        }
        if (gzeof(fp_)) return !line.empty();
      }
    }

    static auto attr(std::string_view body, std::string_view key) -> std::string {
      // value of key= inside <...>; values may be quoted
      std::size_t p = 0;
      while (p < body.size()) {
        std::size_t eq = body.find('=', p);
        if (eq == std::string_view::npos) break;
        std::string_view k = body.substr(p, eq - p);
        std::size_t vbeg = eq + 1, vend;
        if (vbeg < body.size() && body[vbeg] == '"') {
          vend = body.find('"', vbeg + 1);
          if (vend == std::string_view::npos) vend = body.size();
          std::string_view val = body.substr(vbeg + 1, vend - vbeg - 1);
          if (k == key) return std::string(val);
          p = body.find(',', vend);
          p = p == std::string_view::npos ? body.size() : p + 1;
        } else {
          vend = body.find(',', vbeg);
          if (vend == std::string_view::npos) vend = body.size();
          if (k == key) return std::string(body.substr(vbeg, vend - vbeg));
          p = vend + 1;
        }
      }
      return {};
    }

    void read_header() {
      std::string line;
      while (getline(line)) {
        if (line.rfind("##", 0) == 0) {
          auto lt = line.find('<'), gt = line.rfind('>');
          if (lt == std::string::npos || gt == std::string::npos || gt < lt) continue;
          std::string_view body(line.data() + lt + 1, gt - lt - 1);
          if (line.rfind("##contig=", 0) == 0) {
            auto id = attr(body, "ID");
            if (!id.empty() && !contig_seen_.count(id)) {
              contig_seen_.emplace(id, contigs_.size());
              contigs_.push_back(id);
            }
          } else if (line.rfind("##INFO=", 0) == 0) {
            auto id = attr(body, "ID");
            auto ty = attr(body, "Type");
            InfoType t = ty == "Integer" ? InfoType::Integer
                         : ty == "String" ? InfoType::String
                         : ty == "Flag"   ? InfoType::Flag
                         : ty == "Float"  ? InfoType::Float
                                          : InfoType::Other;
            if (!id.empty()) info_types_.emplace(id, t);
          }
        } else if (line.rfind("#", 0) == 0) {
          return;  // the #CHROM line ends the header
        } else {
          pending_ = line;  // headerless file: first data line
          return;
        }
      }
    }

    // typed lookup with the failure modes of bcf_get_info_values (undeclared / wrong type / absent)
    auto info_value(const std::unordered_map<std::string_view, std::string_view> &kv, const char *tag, InfoType want) const
        -> std::string_view {
      auto ht = info_types_.find(tag);
      if (ht == info_types_.end() || ht->second != want) throw binary::VcfReaderError(std::string("Failed to get info ") + tag);
      auto it = kv.find(tag);
      if (it == kv.end() || it->second.empty() || it->second == ".")
        throw binary::VcfReaderError(std::string("Failed to get info ") + tag);
      return it->second;
    }
    auto info_string(const std::unordered_map<std::string_view, std::string_view> &kv, const char *tag) const -> std::string {
      return std::string(info_value(kv, tag, InfoType::String));
    }
    auto info_pos(const std::unordered_map<std::string_view, std::string_view> &kv, const char *tag) const -> pos_t {
      std::string_view v = info_value(kv, tag, InfoType::Integer);
      v = v.substr(0, v.find(','));  // first value of a vector
      long long x = 0;
      auto [p, ec] = std::from_chars(v.data(), v.data() + v.size(), x);
      if (ec != std::errc() || p != v.data() + v.size()) throw binary::VcfReaderError(std::string("Failed to get info ") + tag);
      return static_cast<pos_t>(static_cast<std::int32_t>(x));  // htslib stores int32, the reference reads it as pos_t
    }

    void parse_line(const std::string &line, Record &r) const {
      std::string_view f[8];
      std::size_t p = 0;
      int nf = 0;
      for (; nf < 8; ++nf) {
        std::size_t t = line.find('\t', p);
        if (t == std::string::npos) {
          f[nf++] = std::string_view(line).substr(p);
          break;
        }
        f[nf] = std::string_view(line).substr(p, t - p);
        p = t + 1;
      }
      if (nf < 8) throw binary::VcfReaderError("Failed to read line in vcf ");
      r = Record{};
      r.chrom = std::string(f[0]);
      long long pos1 = 0;
      {
        auto [q, ec] = std::from_chars(f[1].data(), f[1].data() + f[1].size(), pos1);
        if (ec != std::errc() || q != f[1].data() + f[1].size()) throw binary::VcfReaderError("Failed to read line in vcf ");
      }
      r.pos = static_cast<pos_t>(pos1 - 1);
      r.rlen = static_cast<pos_t>(f[3].size());

      std::unordered_map<std::string_view, std::string_view> kv;
      std::string_view info = f[7];
      for (std::size_t s = 0; s <= info.size();) {
        std::size_t e = info.find(';', s);
        if (e == std::string_view::npos) e = info.size();
        std::string_view item = info.substr(s, e - s);
        if (!item.empty() && item != ".") {
          std::size_t eq = item.find('=');
          if (eq == std::string_view::npos) kv.emplace(item, std::string_view{});
          else kv.emplace(item.substr(0, eq), item.substr(eq + 1));
        }
        s = e + 1;
      }
      if (auto it = kv.find("END"); it != kv.end()) {  // htslib: rlen follows INFO/END when it is an Integer tag
        auto ht = info_types_.find("END");
        long long e1 = 0;
        if (ht != info_types_.end() && ht->second == InfoType::Integer &&
            std::from_chars(it->second.data(), it->second.data() + it->second.size(), e1).ec == std::errc() && e1 > 0 &&
            e1 - 1 >= static_cast<long long>(r.pos))
          r.rlen = static_cast<pos_t>(e1 - static_cast<long long>(r.pos));
      }

      // Sv2nlInfoField::update, vcf_info.cpp:9-43
      r.svtype = info_string(kv, "SVTYPE");
      if (r.svtype == "TRA" || r.svtype == "BND") r.chr2 = info_string(kv, "CHR2");
      if (r.svtype == "INV") {
        try {  // both lookups sit in one try block: if STRAND1 is unavailable STRAND2 is not read either
          r.strand1 = info_string(kv, "STRAND1") == "+";
          r.strand2 = info_string(kv, "STRAND2") == "+";
        } catch (...) {
        }
      }
      if (r.svtype == "BND")
        r.svend = info_pos(kv, "POS2");
      else if (source_ == "nls")
        r.svend = info_pos(kv, "SVEND");
      else
        r.svend = info_pos(kv, "END");
    }

    std::string path_, source_;
    gzFile fp_{nullptr};
    std::vector<std::string> contigs_;
    std::unordered_map<std::string, std::size_t> contig_seen_;
    std::unordered_map<std::string, InfoType> info_types_;
    std::string pending_;
  };

}  // namespace sv2nl

#endif  // BINARY_AMD_SV2NL_VCF_READER_HPP_
