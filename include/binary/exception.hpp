// binary/exception.hpp — the one exception type of the drop-in headers.
//
// binary::VcfReaderError is what the reference's VcfParser throws (library/include/binary/exception.hpp:13-20) and what
// code written against it catches; binary/parser/vcf.hpp here raises it in the same situations (unreadable file, a
// line that is not a VCF record, an INFO tag that is undeclared, of another type, or absent from the record).
// A std::runtime_error underneath (the reference derives from std::exception directly): what() is the message either
// way, and a handler for std::exception or for VcfReaderError catches both forms.
#ifndef BINARY_AMD_EXCEPTION_HPP_
#define BINARY_AMD_EXCEPTION_HPP_

#include <stdexcept>
#include <string>

namespace binary {

  struct VcfReaderError : std::runtime_error {
    using std::runtime_error::runtime_error;
  };

}  // namespace binary

#endif  // BINARY_AMD_EXCEPTION_HPP_
