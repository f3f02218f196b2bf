// binary/exception.hpp — the one exception type of the drop-in headers.
//
// binary::VcfReaderError is what the reference's VcfParser throws (library/include/binary/exception.hpp:13-20) and what
// code written against it catches; binary/parser/vcf.hpp here raises it in the same situations (unreadable file, a
// line that is not a VCF record, an INFO tag that is undeclared, of another type, or absent from the record).
#ifndef BINARY_AMD_EXCEPTION_HPP_
#define BINARY_AMD_EXCEPTION_HPP_

#include <exception>
#include <string>
#include <utility>

namespace binary {

  class VcfReaderError : public std::exception {
  public:
    explicit VcfReaderError(std::string text) : text_(std::move(text)) {}
    [[nodiscard]] auto what() const noexcept -> const char * override { return text_.c_str(); }

  private:
    std::string text_;
  };

}  // namespace binary

#endif  // BINARY_AMD_EXCEPTION_HPP_
