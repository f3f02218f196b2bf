// vcf_info.hpp — sv2nl's info field and the record / interval / tree types built on it.
//
// Drop-in for the reference's standalone/sv2nl/include/vcf_info.hpp + source/vcf_info.cpp: same namespace, type names
// and members (Sv2nlInfoField: svtype, svend, chr2, strand1, strand2; the aliases Sv2nlVcfRecord, Sv2nlVcfRanges,
// Sv2nlVcfInterval, Sv2nlVcfIntervalNode, Sv2nlVcfIntervalTree, vcf_info.hpp:42-46), so mapper code written against
// them compiles. An Sv2nlVcfIntervalTree keeps its records on the host and answers find_overlaps on the GPU
// (binary/algorithm/interval_tree.hpp).
//
// Where the END coordinate comes from (vcf_info.cpp:9-43): POS2 for BND records (delly translocations); SVEND when the
// file's source is "nls" (ScanNLS); END otherwise (delly). CHR2 is read for TRA / BND only, the strands for INV only —
// both inside ONE try block, so a missing STRAND1 leaves STRAND2 unread (default '+') — and every other missing tag
// ends the reading with binary::VcfReaderError.
#ifndef BINARY_AMD_SV2NL_VCF_INFO_HPP_
#define BINARY_AMD_SV2NL_VCF_INFO_HPP_

#include <binary/algorithm/interval_tree.hpp>
#include <binary/parser/vcf.hpp>
#include <memory>
#include <ostream>
#include <string>
#include <string_view>

namespace sv2nl {
  namespace tree = binary::algorithm::tree;
  namespace vcf = binary::parser::vcf;

  struct Sv2nlInfoField final : vcf::BaseInfoField {
    // what sv2nl keeps of a record's INFO column
    std::string svtype;
    vcf::pos_t svend = 0;  // as written in the file (a 1-based coordinate, not shifted)
    std::string chr2;      // TRA / BND records only
    bool strand1 = true;   // INV records only; true: '+'
    bool strand2 = true;

    void update(std::shared_ptr<vcf::details::DataImpl> const &data, std::string_view source) override {
      const auto *hdr = data->header.get();
      const auto *rec = data->record.get();
      auto text = [&](const char *tag) { return vcf::get_info_field<char>(tag, hdr, rec); };
      svtype = text("SVTYPE");
      const bool bnd = svtype == "BND";
      // As in the reference (vcf_info.cpp:14-31) the object is updated in place, record after record: chr2 and the strands
      // keep what the record before left there when this one does not set them (an INV line without STRAND tags).
      if (bnd || svtype == "TRA") chr2 = text("CHR2");
      if (svtype == "INV") {
        try {  // ONE block for both: without STRAND1 the second tag is not looked at
          strand1 = text("STRAND1") == "+";
          strand2 = text("STRAND2") == "+";
        } catch (...) {  // (files without strands: both stay '+' from the first record on)
        }
      }
      const char *end_tag = bnd ? "POS2" : source == "nls" ? "SVEND" : "END";
      svend = vcf::get_info_field<vcf::pos_t>(end_tag, hdr, rec);
    }
  };

  inline auto operator<<(std::ostream &os, Sv2nlInfoField const &i) -> std::ostream & {
    return os << "svtype: " << i.svtype << " svend: " << i.svend;
  }
  inline auto operator==(Sv2nlInfoField const &a, Sv2nlInfoField const &b) -> bool {
    return a.svend == b.svend && a.svtype == b.svtype;
  }

  using Sv2nlVcfRecord = vcf::BaseVcfRecord<Sv2nlInfoField>;
  using Sv2nlVcfRanges = vcf::VcfRanges<Sv2nlVcfRecord>;
  using Sv2nlVcfInterval = vcf::BaseVcfInterval<Sv2nlVcfRecord>;
  using Sv2nlVcfIntervalNode = tree::IntervalNode<Sv2nlVcfInterval>;
  using Sv2nlVcfIntervalTree = tree::IntervalTree<Sv2nlVcfIntervalNode>;

}  // namespace sv2nl

#endif  // BINARY_AMD_SV2NL_VCF_INFO_HPP_
