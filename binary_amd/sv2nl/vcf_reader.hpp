// vcf_reader.hpp — the sv2nl tool's flat view of a VCF file.
//
// The tool holds its records as plain structs (a million of them at a time; the reference's record type owns a heap
// info field per record), read by the SAME reader and the SAME extraction rules as the drop-in types:
// binary::parser::vcf::details::DataImpl (include/binary/parser/vcf.hpp — lines, header contigs, typed INFO lookup
// with htslib's failure modes) and sv2nl::Sv2nlInfoField::update (vcf_info.hpp — which tag holds the end coordinate).
// Reference path replaced: VcfRanges / BaseVcfRecord::next/update (library/include/binary/parser/vcf.hpp:263-275,
// 305-310), chroms() (:577-589), Sv2nlInfoField::update (standalone/sv2nl/source/vcf_info.cpp:9-43).
#ifndef BINARY_AMD_SV2NL_VCF_READER_HPP_
#define BINARY_AMD_SV2NL_VCF_READER_HPP_

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "vcf_info.hpp"

namespace sv2nl {

  using pos_t = vcf::pos_t;

  /// What sv2nl keeps of one VCF line (the fields of the reference's Sv2nlVcfRecord, flattened).
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

  class VcfFile {
  public:
    /// source: "nls" (ScanNLS: end coordinate in SVEND) or "delly" (END); BND records use POS2 either way.
    VcfFile(const std::string &path, std::string source)
        : path_(path), source_(std::move(source)), data_(std::make_shared<vcf::details::DataImpl>(path)) {}

    /// Header contigs in header order (what VcfRanges::chroms() returns).
    [[nodiscard]] auto chroms() const -> const std::vector<std::string> & { return data_->header->contigs; }
    [[nodiscard]] auto file_path() const -> const std::string & { return path_; }

    /// Reads the next record; false at end of file. Throws binary::VcfReaderError like the reference.
    auto next(Record &out) -> bool {
      if (!data_->read()) return false;
      info_.update(data_, source_);
      const auto &line = *data_->record;
      out.chrom.assign(line.chrom);
      out.pos = line.pos;
      out.rlen = line.rlen;
      out.svtype = info_.svtype;
      out.svend = info_.svend;
      out.chr2 = info_.chr2;
      out.strand1 = info_.strand1;
      out.strand2 = info_.strand2;
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
    std::string path_, source_;
    std::shared_ptr<vcf::details::DataImpl> data_;
    Sv2nlInfoField info_;
  };

}  // namespace sv2nl

#endif  // BINARY_AMD_SV2NL_VCF_READER_HPP_
